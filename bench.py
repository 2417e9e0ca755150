#!/usr/bin/env python3
"""Benchmark of the LEAF text-encoder adversarial fine-tuning step on MI355X (BASELINE.json metric).

One "step" = anchor forward [B,77] + 2k x score_candidates([B*rho,77]) + train forward/backward of the TextFARE
loss on [B,77] + (flat RCCL all-reduce) + AdamW + weight re-pack, on synthetic token ids resident in HBM.
value = B * world / step_time  (adversarial text samples/s; the reference's own formula, utils_AT.py:390).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (fields: see the task contract; plus `roofline` for the dominant GEMM kernel timed
live with HIP events around every launch on its stream, `cpu_baseline` = the numpy oracle timed on the host cores
for a bounded sample of the same workload, and `step_mfma_frac` = whole-step algorithmic FLOP/s / dense peak).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS_16BIT = 2500.0  # MI355X dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
FAMILY = {0: "gemm_nt_kernel", 1: "gemm_nt256_kernel", 2: "gemm_nt256_ring_kernel", 3: "gemm_nt256_persist_kernel",
          4: "gemm_nt256_half_kernel", 5: "gemm_nt256_halfp_kernel",
          6: "gemm_nt64_ring_kernel"}
EPI_NAMES = {0: "store16(qkv)", 1: "act16(c_fc)", 2: "resid32(out_proj+c_proj)", 3: "store32(wgrad/dgrad)", 4: "actgrad16"}


def fwd_flops_per_seq(cfg, L=77):
    """F = layers * (24 d^2 L + 4 L^2 d), dense, full-square attention."""
    return cfg.layers * (24 * cfg.width ** 2 * L + 4 * L * L * cfg.width)


def cpu_baseline(model_name, rho, k, b_cpu, seed):
    """The numpy oracle (kind "port") running the same step on b_cpu captions; all host cores via OpenBLAS."""
    import numpy as np
    from oracle import text_oracle as O
    cfg = O.CONFIGS[model_name]
    w = O.init_weights(cfg, seed=1)
    base = O.synthetic_tokens(b_cpu, seed=seed)
    t0 = time.time()
    anchor = O.encode_text(w, cfg, base)
    cur = base
    for _ in range(k):
        cand = O.synthetic_candidates(cur, rho, seed=seed + 1)
        idx, _, _ = O.score_candidates(w, cfg, cand, anchor, chunk=50)
        pos = np.array([int(np.nonzero(cand[b, idx[b]] != cur[b])[0][0]) if np.any(cand[b, idx[b]] != cur[b]) else 1
                        for b in range(b_cpu)])
        cand = O.synthetic_candidates(cur, rho, seed=seed + 2, fixed_pos=pos)
        idx, _, _ = O.score_candidates(w, cfg, cand, anchor, chunk=50)
        cur = cand[np.arange(b_cpu), idx]
    loss, _, g = O.encode_text_backward(w, cfg, cur, anchor)
    m = {kk: np.zeros_like(v) for kk, v in w.items()}
    v = {kk: np.zeros_like(vv) for kk, vv in w.items()}
    O.adamw_step(w, g, m, v, 1, 1e-5, 1e-4)
    dt = time.time() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": b_cpu / dt, "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": f"numpy fp32 oracle, one full step on B={b_cpu} captions (rho={rho}, k={k}: "
                      f"{2 * rho * k * b_cpu} candidate forwards + anchor + train fwd/bwd + AdamW), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default=os.environ.get("LEAF_BENCH_MODEL", "ViT-L-14-quickgelu"))
    ap.add_argument("--batch", type=int, default=int(os.environ.get("LEAF_BENCH_BATCH", "128")))
    ap.add_argument("--rho", type=int, default=50)
    ap.add_argument("--k-adv", type=int, default=int(os.environ.get("LEAF_BENCH_K", "1")))
    ap.add_argument("--dtype", default=os.environ.get("LEAF_DTYPE", "fp16"), choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense", action="store_true", help="compute all 77 rows per sequence (no EOT trimming)")
    ap.add_argument("--no-prefix-reuse", action="store_true", help="recompute every kept row of every candidate")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--attack", default="leaf", choices=["leaf", "pgd"],
                    help="leaf = the reference's character search (the BASELINE.json metric); pgd = the OPTIONAL embedding-space "
                         "PGD mode of SURVEY.md 8a row a12 (k-adv inner steps), which the reference's text trainer does not run")
    ap.add_argument("--pgd-eps", type=float, default=0.05)
    ap.add_argument("--pgd-alpha", type=float, default=0.02)
    ap.add_argument("--pgd-norm", default="linf", choices=["linf", "l2"])
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from leaf_amd import _lib
    from leaf_amd.model import LeafCLIPText, create_model, get_config
    from leaf_amd.step import StepConfig, train_step_tokens

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("LEAF_BENCH_FORCE_DIST") == "1"   # the latter rehearses the RCCL path on 1 GPU
    if use_dist:
        # the image exports NCCL_DEBUG=VERSION, which makes RCCL print a version banner on STDOUT in front of the one JSON
        # line this script owes its caller; keep stdout clean (any other NCCL_DEBUG setting is respected)
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = get_config(args.model)
    model = create_model(args.model, device=dev, dtype=args.dtype, seed=1, trainable=True)
    frozen = LeafCLIPText(cfg, device=dev, dtype=args.dtype).copy_from(model)
    frozen.pack()
    model.pack()
    sc = StepConfig(rho=args.rho, k_adv=args.k_adv, lr=1e-5, wd=1e-4, attack=args.attack, pgd_eps=args.pgd_eps,
                    pgd_alpha=args.pgd_alpha, pgd_norm=args.pgd_norm)
    # synthetic captions: SOT, U{8..40} ids, EOT, zero pad; a different shard per rank (seed + rank)
    g = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    base = torch.zeros(B, cfg.context_length, dtype=torch.int32)
    lens = torch.randint(8, 41, (B,), generator=g)
    for i in range(B):
        n = int(lens[i])
        base[i, 0] = cfg.vocab_size - 2
        base[i, 1:1 + n] = torch.randint(1, cfg.vocab_size - 2, (n,), generator=g, dtype=torch.int32)
        base[i, 1 + n] = cfg.vocab_size - 1
    base_lens = None if args.dense else (lens.numpy().astype(np.int32) + 2)   # SOT + n ids + EOT rows are kept
    base = base.to(dev)
    base_ready = torch.cuda.Event()
    base_ready.record()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step_id = 0
    for _ in range(args.warmup):
        train_step_tokens(model, frozen, base, sc, seed=step_id, base_lens=base_lens, prefix_reuse=not args.no_prefix_reuse,
                          base_ready=base_ready)
        step_id += 1
    lib = _lib.lib()
    barrier()
    if rank == 0:
        lib.leaf_prof_begin()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = train_step_tokens(model, frozen, base, sc, seed=step_id, base_lens=base_lens, prefix_reuse=not args.no_prefix_reuse,
                          base_ready=base_ready)
        step_id += 1
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        nk = 128
        ms, fl, by, cnt = (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_int64 * nk)()
        _lib.check(lib.leaf_prof_end(ms, fl, by, cnt, nk), "leaf_prof_end")
        kinds = [(ms[i], fl[i], cnt[i], i) for i in range(nk) if cnt[i] > 0]
        kinds.sort(reverse=True)
        dom_ms, dom_fl, dom_cnt, dom_key = kinds[0]
        dom_bytes = by[dom_key]
        dt_code = 1 if args.dtype == "fp16" else 0
        achieved = dom_fl / (dom_ms * 1e-3) / 1e12
        gemm_total_ms = sum(k[0] for k in kinds)
        gemm_total_fl = sum(k[1] for k in kinds)
        F = fwd_flops_per_seq(cfg)
        # algorithmic cost per sample: LEAF search (2 rho k candidate forwards + anchor + train fwd + 2x bwd), or for the
        # optional embedding-space PGD mode k x (fwd + input-gradient bwd ~ 1 + 1) on top of anchor + start fwd + train bwd
        flops_per_sample = (2 * args.rho * args.k_adv + 4) * F if args.attack == "leaf" else (2 * args.k_adv + 4) * F
        value = B * world * args.steps / dt
        traffic = None     # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                tj = json.load(f)
            if tj["kernel"] == f"{FAMILY[dom_key // 16]}<{'F16' if (dom_key // 8) % 2 == 1 else 'BF16'},{dom_key % 8}>" and not args.dense \
                    and not args.no_prefix_reuse and args.model == "ViT-L-14-quickgelu" and B == 128:
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "adversarial text samples/sec", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype + " MFMA operands, f32 accumulate", "data": "synthetic",
            "config": {"workload": (f"CLIP {args.model} text encoder, LEAF step k={args.k_adv} rho={args.rho}, "
                                    f"B={B} per GPU, seq=77 (BASELINE.json configs[1])") if args.attack == "leaf" else
                                   (f"CLIP {args.model} text encoder, OPTIONAL embedding-space PGD mode (SURVEY 8a row a12, NOT "
                                    f"the reference's text attack): k={args.k_adv} steps, {args.pgd_norm} eps={args.pgd_eps} "
                                    f"alpha={args.pgd_alpha}, B={B} per GPU, seq=77"),
                       "attack": args.attack,
                       "global_batch": B * world, "seq_len": cfg.context_length, "rho": args.rho, "k": args.k_adv,
                       "parallelism": f"dp{world}", "candidate_forwards_per_step_per_gpu": 2 * args.rho * args.k_adv * B},
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS_16BIT, "unit": "TFLOP/s",
                "frac": achieved / PEAK_TFLOPS_16BIT, "traffic": traffic,
                "traffic_note": "HBM bytes per launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) * 1024, profiles/r01_traffic.json",
                "kernel": f"{FAMILY[dom_key // 16]}<{'F16' if (dom_key // 8) % 2 == 1 else 'BF16'},{dom_key % 8}> {EPI_NAMES.get(dom_key % 8)}",
                "algorithmic_bytes_per_launch": dom_bytes / dom_cnt if dom_bytes else None,
                "launches": int(dom_cnt), "avg_launch_ms": dom_ms / dom_cnt,
                "algorithmic_gflop_per_launch": dom_fl / dom_cnt / 1e9,
                "all_gemm_tflops": gemm_total_fl / (gemm_total_ms * 1e-3) / 1e12,
                "gemm_share_of_step": gemm_total_ms * 1e-3 / dt,
            },
            "step_mfma_frac": value * flops_per_sample / (world * PEAK_TFLOPS_16BIT * 1e12),
            "algorithmic_tflop_per_sample": flops_per_sample / 1e12,
            "exact_work_skipping": "none (dense, 77 rows per sequence)" if args.dense else
                                   "EOT trimming: rows after EOT are not computed (bit-identical outputs); "
                                   f"mean kept rows {float(base_lens.mean()):.1f} of 77" +
                                   ("" if args.no_prefix_reuse else "; prefix reuse: rows before the edited token come "
                                    "from the clean caption's per-layer K/V cache (bit-identical outputs)"),
            "executed_gemm_tflop_per_step": gemm_total_fl / args.steps / 1e12,
            "loss": float(loss),
        }
        if world == 1 and not args.no_cpu_baseline and args.attack == "leaf":
            oracle_name = args.model if args.model in ("ViT-L-14", "ViT-L-14-quickgelu", "ViT-H-14", "ViT-g-14", "ViT-bigG-14") else "ViT-L-14"
            out["cpu_baseline"] = cpu_baseline(oracle_name, args.rho, args.k_adv, args.cpu_batch, seed=1234)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
