#!/usr/bin/env python3
"""Benchmark of the LEAF text-encoder adversarial fine-tuning step on MI355X (BASELINE.json metric).

One "step" = anchor forward [B,77] + 2k x score_candidates([B*rho,77]) + train forward/backward of the TextFARE
loss on [B,77] + (flat RCCL all-reduce) + AdamW + weight re-pack, on synthetic token ids resident in HBM.
value = B * world / step_time  (adversarial text samples/s; the reference's own formula, utils_AT.py:390).

    python bench.py                               # 1 GPU, default steps / warmup
    python bench.py --gpus 2 --steps 50           # starts 2 ranks itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W    # the driver's form: runs as launched

Prints ONE JSON line on rank 0: the task contract's fields, plus
  roofline      the dominant GEMM kernel, timed live with HIP events around every launch on its own stream
                (leaf_prof_*), and `shapes`: the same figures per GEMM shape (N, K) of every kernel family;
  cpu_baseline  a plain PyTorch-CPU fp32 harness of the same step (oracle/torch_cpu_harness.py) on BASELINE.json
                configs[0] (B = 8), bounded in time, all usable host cores (rank 0, N = 1 only);
  dense         samples/s of the same step with every exact work-skipping switched off (a few steps, after the timed region);
  step_exec_frac    executed GEMM FLOP/s of the whole step / dense 16-bit MFMA peak  (the honest step-level fraction);
  dense_equiv_frac  samples/s x dense algorithmic FLOPs per sample / peak (SURVEY.md 8d's prescription; exceeds the
                    executed fraction by the factor exact work skipping removes);
  ms_per_step_first_tenth / _last_tenth   the step is data dependent (the rows the second stage computes follow the search's
                    winners, which move as the model trains on the synthetic batch): both ends of the timed region.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from leaf_amd import configure_runtime  # noqa: E402

configure_runtime()   # HIP_FORCE_DEV_KERNARG=1, before the HIP runtime initialises (leaf_amd/__init__.py says why)

PEAK_TFLOPS_16BIT = 2500.0  # MI355X dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak (same table; ~6.3 TB/s achievable)
TRAFFIC_GLOB = "r06_traffic*.json"   # PMC summaries, one per workload (tools/pmc_passes.sh, tools/pmc_summary.py)
FAMILY = {0: "gemm_nt_kernel", 1: "gemm_nt256_kernel", 4: "gemm_nt256_half_kernel", 6: "gemm_nt64_ring_kernel", 7: "gemm_nt128pp_kernel",
          8: "qkv_attn_kernel"}
EPI_NAMES = {0: "store16", 1: "act16", 2: "resid32", 3: "store32", 4: "actgrad16", 5: "lnfold16", 6: "lnfold_act16", 7: "resid32+x16+stats", 8: "resid16+8+stats"}


def kernel_sources_hash():
    """sha256 over the GEMM kernel sources: a traffic summary under profiles/ is only attached to a line produced by the
    same kernels."""
    h = hashlib.sha256()
    for f in ("gemm256h.hip", "gemm64.hip", "gemm.hip", "qkv_attn.hip", "gemm_epilogue.h", "common.h", "lnfold.h"):
        p = os.path.join(ROOT, "leaf_amd", "csrc", f)
        if os.path.exists(p):
            with open(p, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def fwd_flops_per_seq(cfg, L=77):
    """F = layers * (24 d^2 L + 4 L^2 d), dense, full-square attention."""
    return cfg.layers * (24 * cfg.width ** 2 * L + 4 * L * L * cfg.width)


def cpu_baseline(model_name, rho, k, b_cpu, seed, budget_s, gpu_encode=None, parity_rows=512):
    """Plain PyTorch-CPU fp32 harness (own code, oracle/torch_cpu_harness.py) of the same step, dense, all usable cores.
    ``gpu_encode`` (tokens -> features of the benchmark's start weights on the GPU): the embeddings of ``parity_rows`` rows (synthetic
    captions + single-edit candidates of them, what the search scores) are compared with the harness's fp32 ones, so that every
    performance line carries its accuracy (``parity_rel_l2``: batch figure, row quantiles, rows above 1e-3; the oracle is the checker
    here, outside the timed region; the 12,928-row census is profiles/r06_row_error_census*.txt)."""
    import numpy as np
    from oracle import text_oracle as O
    from oracle import torch_cpu_harness as H
    cfg = O.CONFIGS[model_name]
    w = O.init_weights(cfg, seed=1)
    base = O.synthetic_tokens(b_cpu, seed=seed)
    parity = None
    if gpu_encode is not None:
        import torch
        torch.set_num_threads(H.usable_cores())
        nb = max(parity_rows // 8, 1)
        pb = O.synthetic_tokens(nb, seed=seed)
        rows_t = np.concatenate([pb, O.synthetic_candidates(pb, 7, seed=seed + 1).reshape(-1, pb.shape[1])])
        L = int(rows_t.argmax(-1).max()) + 1                  # causal attention: cutting behind the longest EOT is exact
        tower = H.TorchTextTower(w, cfg)
        with torch.no_grad():
            ref = np.concatenate([tower.encode_text(torch.from_numpy(rows_t[s:s + 128, :L].astype(np.int64))).numpy()
                                  for s in range(0, rows_t.shape[0], 128)])
        got = gpu_encode(rows_t)
        rows = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
        parity = {"captions": int(rows_t.shape[0]), "global": float(np.linalg.norm(got - ref) / np.linalg.norm(ref)), "row_max": float(rows.max()),
                  "row_median": float(np.median(rows)), "row_p99": float(np.quantile(rows, 0.99)), "rows_above_1e-3": int((rows > 1e-3).sum()),
                  "tolerance": 1e-3,
                  "rows": f"{nb} synthetic captions + 7 single-edit candidates each",
                  "against": "plain PyTorch CPU fp32 forward of the same start weights (oracle/torch_cpu_harness.py)"}
    r = H.time_step(w, cfg, base, lambda cur, rho_, s, pos: O.synthetic_candidates(cur, rho_, s, fixed_pos=pos), rho, k,
                    budget_s=budget_s)
    return {"value": r["samples_per_s"], "unit": "samples/s", "cores": r["cores"], "kind": "port", "parity_rel_l2": parity,
            "impl": "own harness: plain PyTorch CPU fp32 (torch.nn.functional ops), dense 77-row sequences",
            "sample": (f"BASELINE.json configs[0] shape: one step on B={b_cpu} captions, rho={rho}, k={k}: anchor + "
                       f"{r['total_candidate_forwards']} candidate forwards ({r['measured_candidate_forwards']} measured at "
                       f"{r['seq_per_s']:.1f} seq/s{', rest extrapolated' if r['extrapolated'] else ''}) + train fwd/bwd + AdamW "
                       f"({r['t_train_s']:.1f} s); {r['wall_s']:.1f} s of CPU time spent")}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def step_trend(marks):
    """ms per step over the first and the last tenth of the timed steps (events on the step's stream).  The exact work skipping
    makes the step DATA dependent: as the model trains on the synthetic batch the search's winners move and with them the rows
    the second stage has to compute (lr = 0 keeps the time flat), so a 20-step run and a 200-step run average differently."""
    n = len(marks)
    if n < 20:
        return {}
    k = max(n // 10, 2)
    first = marks[0].elapsed_time(marks[k]) / k
    last = marks[n - 1 - k].elapsed_time(marks[n - 1]) / k
    return {"ms_per_step_first_tenth": first, "ms_per_step_last_tenth": last,
            "step_time_note": "data dependent (exact work skipping follows the search's winners as the model trains); value = all timed steps"}


def self_launch(args, argv):
    """--gpus N > 1 from a bare shell: start N fresh ranks with torch.distributed.run.  This parent has not touched the
    GPU (no torch.cuda call, nothing imported that initialises HIP); it only waits and forwards the children's output."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def per_rank_summary(gathered):
    """Per-rank diagnostics of a multi-rank run (rank order): what the first scaling run needs in order to say WHY it is 0.8 or
    0.95.  ``ms_per_step`` is each rank's own step time (its stream's events; the ranks meet at every gradient reduction, so they
    agree closely), ``exposed_collective_ms_per_step`` the time its stream waited inside the reduction -- small on the rank whose
    data-dependent search took longest, large on the ranks that waited for it -- and ``scored_rows_per_step`` the rows its search
    computed (exact work skipping makes that differ by rank)."""
    import statistics
    cols = list(zip(*[[float(x) for x in g.tolist()] for g in gathered]))
    names = ("ms_per_step", "exposed_collective_ms_per_step", "scored_rows_per_step")
    out = {n: list(c) for n, c in zip(names, cols)}
    out.update({n + "_min_median_max": [min(c), statistics.median(c), max(c)] for n, c in zip(names, cols)})
    ms, ex = cols[0], cols[1]
    out["compute_ms_per_step"] = [m - e for m, e in zip(ms, ex)]        # step time without the wait: the slowest rank sets the pace
    return out


def dry_run(args):
    """Launcher / rendezvous rehearsal without a GPU (tests/test_bench_launch.py): gloo process group, one all-reduce,
    the same one-JSON-line contract with n_ranks_seen."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    seen = 1
    mine = torch.tensor([10.0 + rank, 0.5 * rank, 1000.0 * (rank + 1)], dtype=torch.float64)   # stand-ins for the per-rank diagnostics
    gathered = [mine]
    if world > 1:
        dist.all_reduce(t)
        seen = dist.get_world_size()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "adversarial text samples/sec", "value": 0.0, "unit": "samples/s", "n_gpus": args.gpus,
                          "n_ranks_seen": seen, "dry": True, "backend": args.backend, "allreduce_sum": float(t.item()),
                          "steps": args.steps, "warmup": args.warmup, "host_threads_per_rank": args.host_threads,
                          "per_rank": per_rank_summary(gathered)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def dp_bucket_summary(model):
    """The gradient buckets GradReducer all-reduces, in issue order (leaf_amd/step.py:bucket_plan): which backward event releases
    each one and its bytes.  Only the LAST one (embedding tables + every vector) cannot overlap the backward."""
    from leaf_amd.step import bucket_plan
    plan = bucket_plan(model.layout, model.n_params, model.cfg.layers)
    return [{"released_by_block": int(ev) if ev < model.cfg.layers else "end of backward", "bytes": int(sum(n for _, n in ranges) * 4),
             "collectives": int(sum(1 for _, n in ranges if n))} for ev, ranges in plan]


XGMI_LINK_GBS_PER_DIRECTION = 76.8      # MI355X: 7 xGMI links per GPU x ~153 GB/s bidirectional each (the task's figures) = 76.8 GB/s per direction
XGMI_LINKS = 7
XGMI_PROTOCOL_EFF = 0.75                # ASSUMED share of the link rate a large RCCL ring sustains (no multi-GPU box to measure it)


def exposed_collective_model(model, dev, R):
    """What of the gradient reduction stays exposed on R GPUs: the last bucket.  MEASURED here: the time a one-rank RCCL group
    needs for exactly that bucket's all-reduce calls (launch + RCCL's own kernel, no link traffic).  MODELLED: the link time of a
    ring all-reduce over fully connected point-to-point xGMI -- RCCL builds min(R - 1, 7) link-disjoint rings, each bound by ONE
    link direction: t = 2 (R - 1) / R * bytes / (rings * 76.8 GB/s * 0.75)."""
    import torch
    import torch.distributed as dist
    from leaf_amd.step import bucket_plan
    last = bucket_plan(model.layout, model.n_params, model.cfg.layers)[-1][1]
    nbytes = sum(n for _, n in last) * 4
    launch_ms, how, created = None, "not measured", False
    try:
        if not dist.is_initialized():
            created = True
            if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
                del os.environ["NCCL_DEBUG"]
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        if dist.get_backend() == "nccl" and model.grads is not None:
            cur = torch.cuda.current_stream(dev)
            saved = [model.grads[o:o + n].clone() for o, n in last]

            def once():
                for o, n in last:
                    if n:
                        dist.all_reduce(model.grads[o:o + n], op=dist.ReduceOp.SUM)
            for _ in range(3):
                once()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for _ in range(20):
                once()
            e1.record(cur)
            e1.synchronize()
            launch_ms, how = e0.elapsed_time(e1) / 20, "one-rank RCCL group on this GPU, 20 repetitions of the bucket's all-reduce calls"
            for (o, n), v in zip(last, saved):
                model.grads[o:o + n].copy_(v)
    except Exception as e:      # the model is still printed, without the measured part
        how = f"not measured ({type(e).__name__}: {e})"
    if created and dist.is_initialized():
        dist.destroy_process_group()
    rings = max(1, min(R - 1, XGMI_LINKS))
    link_ms = 0.0 if R < 2 else 2 * (R - 1) / R * nbytes / (rings * XGMI_LINK_GBS_PER_DIRECTION * 1e9 * XGMI_PROTOCOL_EFF) * 1e3
    return {"bucket_bytes": int(nbytes), "one_rank_rccl_ms_measured": launch_ms, "measured_how": how, "ranks": R, "rings_assumed": rings,
            "link_GBs_per_direction": XGMI_LINK_GBS_PER_DIRECTION, "protocol_efficiency_assumed": XGMI_PROTOCOL_EFF,
            "link_ms_model": link_ms, "exposed_ms_model": link_ms + (launch_ms or 0.0)}


def rank_sim(args, model, frozen, sc, batch_for, dev, B):
    """the R-GPU step predicted from one GPU.  Data parallelism here = every rank searches and back-propagates its
    own B captions on the SAME weights, the gradients are summed, one AdamW.  That is what this loop executes, one rank after the
    other (micro-batches with accum_scale 1/R = the mean over ranks), with an event behind every rank's backward: t_r.  On R GPUs
    the ranks meet at the gradient reduction, so a step costs max_r t_r (+ the optimizer, + what of the all-reduce stays exposed);
    on one GPU per rank it costs t_r.  pred_eff = E_step[mean_r t_r + t_opt] / E_step[max_r t_r + t_opt + exposed]."""
    import statistics
    import torch
    from leaf_amd.step import StepConfig, get_reducer, train_step_tokens
    R = args.rank_sim
    sc = StepConfig(**{**sc.__dict__, "accum_freq": R})
    cur = torch.cuda.current_stream(dev)
    per_step = []

    def one_step(step, timed):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(R + 2)]
        rows = []
        evs[0].record(cur)
        for r in range(R):
            base, lens_np, ready = batch_for(step, r)
            rows0 = model.rows_scored
            train_step_tokens(model, frozen, base, sc, seed=step * R + r, base_lens=None if args.dense else lens_np,
                              prefix_reuse=not args.no_prefix_reuse, base_ready=ready, micro_index=r, optimizer_step=False)
            evs[r + 1].record(cur)
            rows.append(model.rows_scored - rows0)
        scale = get_reducer(model).finish()         # 1.0 here: one process (main() refuses --rank-sim under a multi-rank launch)
        model.adamw_step(sc.lr, (sc.beta1, sc.beta2), sc.eps, sc.wd, grad_scale=scale)
        model.pack()
        evs[R + 1].record(cur)
        if timed:
            per_step.append((evs, rows))

    for i in range(args.warmup):
        one_step(i, False)
    torch.cuda.synchronize()
    for i in range(args.steps):
        one_step(args.warmup + i, True)
    torch.cuda.synchronize()
    t = [[e[r].elapsed_time(e[r + 1]) for r in range(R)] for e, _ in per_step]
    t_opt = [e[R].elapsed_time(e[R + 1]) for e, _ in per_step]
    mean_r = [statistics.mean(x) for x in t]
    max_r = [max(x) for x in t]
    opt = statistics.mean(t_opt)
    one_gpu = statistics.mean(mean_r) + opt
    n_gpu = statistics.mean(max_r) + opt
    coll = exposed_collective_model(model, dev, R)
    exposed = float(os.environ["LEAF_RANKSIM_EXPOSED_MS"]) if "LEAF_RANKSIM_EXPOSED_MS" in os.environ else coll["exposed_ms_model"]
    out = {"rank_sim": R, "steps": args.steps, "warmup": args.warmup, "B_per_rank": B, "model": args.model, "k": args.k_adv, "rho": args.rho,
           "batches": "fixed per rank" if args.fixed_batch else "new per (rank, step)",
           "ms_per_rank_step_mean": statistics.mean(mean_r), "ms_per_rank_step_by_rank": [statistics.mean(x[r] for x in t) for r in range(R)],
           "ms_max_over_ranks_mean": statistics.mean(max_r), "ms_optimizer": opt,
           "skew_ratio_max_over_mean": statistics.mean(max_r) / statistics.mean(mean_r),
           "rank_time_cv": statistics.mean(statistics.pstdev(x) / statistics.mean(x) for x in t),
           "scored_rows_per_rank_step_min_mean_max": [min(min(r_) for _, r_ in per_step), statistics.mean(statistics.mean(r_) for _, r_ in per_step),
                                                      max(max(r_) for _, r_ in per_step)],
           "pred_eff_skew_only": one_gpu / n_gpu,                       # MEASURED on this GPU: the headline of this mode
           "exposed_collective": coll,
           "exposed_collective_ms_used": exposed,
           "pred_eff_with_collective_ESTIMATE": one_gpu / (n_gpu + exposed),
           "pred_samples_per_s_R_gpus_ESTIMATE": R * B / ((n_gpu + exposed) * 1e-3),
           "samples_per_s_one_gpu_same_loop": B / (one_gpu * 1e-3),
           "note": "one GPU, ranks run one after another on the same weights (gradient sum, one AdamW = the R-rank step's arithmetic); "
                   "t_r = events around rank r's anchor + search + forward + backward; the anchor of rank r + 1 overlaps the tail of rank r "
                   "as in the real step.  pred_eff_skew_only is measured; the collective term is a MODEL (measured one-rank RCCL launch "
                   "time of the exposed bucket + its bytes over the stated xGMI link figures), never a measurement: "
                   "override with LEAF_RANKSIM_EXPOSED_MS"}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default=os.environ.get("LEAF_BENCH_MODEL", "ViT-L-14-quickgelu"))
    ap.add_argument("--batch", type=int, default=int(os.environ.get("LEAF_BENCH_BATCH", "128")))
    ap.add_argument("--rho", type=int, default=50)
    ap.add_argument("--k-adv", type=int, default=int(os.environ.get("LEAF_BENCH_K", "1")))
    ap.add_argument("--accum-freq", type=int, default=1, help="micro-batches per optimizer step (configs[3]: --batch 32 --accum-freq 4)")
    ap.add_argument("--dtype", default=os.environ.get("LEAF_DTYPE", "fp16"), choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-leg", action="store_true")
    ap.add_argument("--dense", action="store_true", help="compute all 77 rows per sequence (no EOT trimming)")
    ap.add_argument("--no-prefix-reuse", action="store_true", help="recompute every kept row of every candidate")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--parity-rows", type=int, default=512, help="rows of the line's parity_rel_l2 (GPU embeddings vs the PyTorch CPU fp32 forward)")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="host threads per rank (torch CPU ops, the native tokenizer: LEAF_HOST_THREADS); default = usable cores // ranks on this node")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--prof-every", type=int, default=4,
                    help="take the per-launch HIP events of the live roofline on every N-th timed step (an event pair makes the queue wait for the "
                         "launch before it: 1.2 ms per 50-ms step when taken on all of them)")
    ap.add_argument("--dry", action="store_true", help="rendezvous rehearsal only (no GPU work); with --backend gloo runs on CPU")
    ap.add_argument("--precision", default=None, choices=["rowsafe", "fast"],
                    help="arithmetic of the forward-only passes (leaf_amd.model.PRECISION_MODES).  rowsafe (DEFAULT, the metric): hi + lo "
                         "operand splits in the GEMMs of the leading blocks that the census prices -- every row of the 12,928-row "
                         "census within 1e-3 of fp32 (profiles/r06_row_error_census*.txt); fast: no split GEMM (rounds 1-5: batch 8.7e-4, "
                         "1.1 %% of the rows above 1e-3)")
    ap.add_argument("--split-blocks", type=int, default=None,
                    help="A/B: hi + lo operand splits in ALL FOUR GEMMs of the first N blocks (the round-5 escape hatch) instead of the "
                         "--precision mode")
    ap.add_argument("--split-masks", default=None,
                    help="A/B: comma-separated split mask per leading block (bit 0 QKV, 1 out_proj weights, 2 c_fc, 3 c_proj weights), "
                         "e.g. 3,1 -- instead of the --precision mode")
    ap.add_argument("--precise-anchor", action="store_true",
                    help="A/B: the frozen model's anchor pass in the fp32-grade arithmetic (leaf_text_forward_precise) instead of the 16-bit one")
    ap.add_argument("--fp32-residual", action="store_true",
                    help="A/B: keep the residual stream of the forward-only passes as fp32 rows beside their 16-bit copy (option compact_resid = 0) "
                         "instead of the 16-bit copy + a remainder byte per element")
    ap.add_argument("--fixed-batch", action="store_true",
                    help="train on ONE synthetic batch for all steps (the rounds 1-4 form; A/B work).  Default: a NEW batch per step, seeded "
                         "by (rank, step), built on a side stream one step ahead -- the step is data dependent and must not depend on how far "
                         "the model has over-fitted one batch")
    ap.add_argument("--rank-sim", type=int, default=0, metavar="R",
                    help="predict the R-GPU data-parallel step from ONE GPU: per step the R ranks' batches run one after another on the same "
                         "weights (gradients summed, ONE AdamW: exactly the R-rank step's arithmetic), each rank's time is taken with events; "
                         "prints pred_eff = (mean_r t_r + t_opt) / (max_r t_r + t_opt + exposed collective) instead of the metric line")
    ap.add_argument("--attack", default="leaf", choices=["leaf", "pgd"],
                    help="leaf = the reference's character search (the BASELINE.json metric); pgd = the OPTIONAL embedding-space "
                         "PGD mode of SURVEY.md 8a row a12 (k-adv inner steps), which the reference's text trainer does not run")
    ap.add_argument("--pgd-eps", type=float, default=0.05)
    ap.add_argument("--pgd-alpha", type=float, default=0.02)
    ap.add_argument("--pgd-norm", default="linf", choices=["linf", "l2"])
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="BASELINE.json configs[i] verbatim (model, k, per-GPU batch, accum): 1 = ViT-L k=1 B=128 (the default), "
                         "2 = ViT-L k=5 B=128, 3 = ViT-H k=2 B=128 as accum 4 x 32, 4 = ViT-bigG k=1 B=256; explicit flags still win")
    args = ap.parse_args()
    if args.config:
        preset = {1: dict(model="ViT-L-14-quickgelu", k_adv=1, batch=128, accum_freq=1),
                  2: dict(model="ViT-L-14-quickgelu", k_adv=5, batch=128, accum_freq=1),
                  3: dict(model="ViT-H-14", k_adv=2, batch=32, accum_freq=4),
                  4: dict(model="ViT-bigG-14", k_adv=1, batch=256, accum_freq=1)}[args.config]
        given = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
        for k, v in preset.items():
            if "--" + k.replace("_", "-") not in given:
                setattr(args, k, v)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # host threads: N ranks on one node share its cores (a default of min(32, cores) per rank would start 8 x 32 threads)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    args.host_threads = args.host_threads or max(1, usable // max(local_world, 1))
    os.environ["LEAF_HOST_THREADS"] = str(args.host_threads)
    if args.dry:
        return dry_run(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from leaf_amd import _lib
    from leaf_amd.model import LeafCLIPText, create_model, get_config
    from leaf_amd.step import StepConfig, train_step_tokens

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    # --backend gloo without --dry: REHEARSAL of the multi-rank step with real GPU work and gloo as the transport (RCCL wants one
    # device per rank); ranks then share the devices there are (local rank % device count).  Never a metric: the line says so.
    rehearsal = args.backend == "gloo"
    if local >= ndev and not rehearsal:
        raise SystemExit(f"rank {rank}: local rank {local} but only {ndev} GPU(s) visible (use --backend gloo to rehearse on fewer)")
    local = local % max(ndev, 1)
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
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_ranks_seen = dist.get_world_size() if use_dist else 1

    cfg = get_config(args.model)
    model = create_model(args.model, device=dev, dtype=args.dtype, seed=1, trainable=True)
    frozen = LeafCLIPText(cfg, device=dev, dtype=args.dtype).copy_from(model)
    frozen.pack()
    model.pack()
    if args.precision:
        model.set_precision(args.precision)
    if args.split_blocks is not None:
        model.set_split_blocks(args.split_blocks)
    if args.split_masks is not None:
        model.set_split_masks([int(x) for x in args.split_masks.split(",") if x != ""])
    if args.fp32_residual:
        model.set_option("compact_resid", 0)
    frozen.copy_from(model)          # weights AND arithmetic (split blocks / policy, residual format)
    frozen.pack()
    compact = bool(model.arithmetic_tag()[0])
    arith = model.precision_name()
    sc = StepConfig(rho=args.rho, k_adv=args.k_adv, lr=1e-5, wd=1e-4, attack=args.attack, pgd_eps=args.pgd_eps,
                    pgd_alpha=args.pgd_alpha, pgd_norm=args.pgd_norm, accum_freq=args.accum_freq, precise_anchor=args.precise_anchor)
    # synthetic captions: SOT, U{8..40} ids, EOT, zero pad; a different shard per rank (seed + rank) and -- unless
    # --fixed-batch -- a different batch per step: the lengths are drawn on the host (the row plan needs them), the ids on the device
    # on a stream of their own, one step ahead of the step that consumes them
    B = args.batch
    ctx, V = cfg.context_length, cfg.vocab_size
    data_stream = torch.cuda.Stream(device=dev)
    dgen = torch.Generator(device=dev)
    col = torch.arange(ctx, device=dev, dtype=torch.int64)[None, :]
    data_stream.wait_stream(torch.cuda.current_stream(dev))       # ``col`` was made on the current stream
    keep_alive = []
    lens_seen = []

    def make_batch(step, rank_=None):
        r_ = rank if rank_ is None else rank_
        seed_ = 1234 + r_ if args.fixed_batch else 1234 + r_ + 1000003 * (step + 1)
        n = torch.randint(8, 41, (B,), generator=torch.Generator().manual_seed(seed_))
        n_pin = n.pin_memory()
        with torch.cuda.stream(data_stream):
            dgen.manual_seed(seed_)
            ids = torch.randint(1, V - 2, (B, ctx), device=dev, generator=dgen, dtype=torch.int32)
            nd = n_pin.to(dev, non_blocking=True)[:, None]
            b = torch.where((col >= 1) & (col <= nd), ids, torch.zeros_like(ids))
            b[:, 0] = V - 2
            b.scatter_(1, nd + 1, torch.full((B, 1), V - 1, device=dev, dtype=torch.int32))
            ev = torch.cuda.Event()
            ev.record(data_stream)
        keep_alive.append((b, n_pin))                  # allocated on data_stream, read on two other streams: keep a few alive
        del keep_alive[:-6]
        ln = n.numpy().astype(np.int32) + 2             # SOT + n ids + EOT rows are kept
        lens_seen.append(float(ln.mean()))
        return b, ln, ev

    fixed = make_batch(0) if args.fixed_batch else None
    pending = {}

    def batch_for(step, rank_=None):
        """(ids, kept rows per caption, ready event) of a step; the NEXT step's batch is requested right away"""
        if fixed is not None and rank_ is None:
            return fixed
        key = (step, rank_)
        cur_ = pending.pop(key, None) or make_batch(step, rank_)
        if rank_ is None:
            pending[(step + 1, None)] = make_batch(step + 1)
        return cur_

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step_id = [0]

    step_marks = []     # one event per timed step (current stream, no synchronisation): the step time is DATA dependent

    prof = {"on": False, "steps": 0}

    def run_steps(n, lens_arg, prefix, mark=False):
        loss = None
        for j in range(n):
            if prof["on"]:                      # the live roofline samples every --prof-every-th timed step
                take = j % max(args.prof_every, 1) == 0
                lib.leaf_prof_pause(0 if take else 1)
                prof["steps"] += int(take)
            base, lens_np, base_ready = batch_for(step_id[0])
            loss = train_step_tokens(model, frozen, base, sc, seed=step_id[0], base_lens=None if lens_arg is None else lens_np,
                                     prefix_reuse=prefix, base_ready=base_ready, micro_index=step_id[0] % args.accum_freq)
            step_id[0] += 1
            if mark:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                step_marks.append(e)
        return loss

    base_lens = None if args.dense else True          # run_steps: None = dense rows, else the batch's own kept-row counts
    lib = _lib.lib()
    if args.rank_sim:
        if world > 1:
            raise SystemExit("--rank-sim predicts the R-GPU step from ONE process; it refuses a multi-rank launch (measure instead: --gpus N)")
        return rank_sim(args, model, frozen, sc, batch_for, dev, B)
    run_steps(args.warmup, base_lens, not args.no_prefix_reuse)
    from leaf_amd.step import get_reducer
    reducer = get_reducer(model)
    reducer.timing = use_dist
    reducer.exposed_ms()          # drop the warm-up's pairs
    rows0 = model.rows_scored
    barrier()
    if rank == 0 and os.environ.get("LEAF_BENCH_NO_PROF") != "1":     # (=1: overhead probe of the live per-launch events; the line then has no roofline)
        lib.leaf_prof_begin()
        prof["on"] = True
    t0 = time.perf_counter()
    loss = run_steps(args.steps, base_lens, not args.no_prefix_reuse, mark=True)
    prof["on"] = False
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    gathered = None
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # per-rank diagnostics: own step time (events on this rank's stream), exposed reduction wait, rows its search computed
        own_ms = step_marks[0].elapsed_time(step_marks[-1]) / max(len(step_marks) - 1, 1) if len(step_marks) > 1 else dt / args.steps * 1e3
        mine = torch.tensor([own_ms, reducer.exposed_ms() / args.steps, (model.rows_scored - rows0) / args.steps],
                            dtype=torch.float64, device="cpu" if rehearsal else dev)
        gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, mine)
    dt = float(tmax.item())
    if os.environ.get("LEAF_BENCH_NO_PROF") == "1":      # overhead probe only: no per-launch events were taken, so no roofline
        if rank == 0:
            print(json.dumps({"probe": "no per-launch events", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps}), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    if rank == 0:
        ng = 256
        ms, fl, by = (C.c_double * ng)(), (C.c_double * ng)(), (C.c_double * ng)()
        rows, info, n_out = (C.c_int64 * ng)(), (C.c_int32 * (4 * ng))(), C.c_int(0)
        _lib.check(lib.leaf_prof_end_shapes(ms, fl, by, rows, info, ng, C.byref(n_out)), "leaf_prof_end_shapes")
        shapes, per_key = [], {}
        psteps = max(prof["steps"], 1)          # timed steps whose launches carry events
        for i in range(n_out.value):
            key, N, K, cnt = info[4 * i], info[4 * i + 1], info[4 * i + 2], info[4 * i + 3]
            big, key = key >= 1024, key % 1024        # launches of >= 16,384 rows (the scoring passes) are grouped apart
            if ms[i] <= 0:
                continue
            kname = f"{FAMILY.get(key // 32, 'gemm?')}<{'F16' if (key // 16) % 2 == 1 else 'BF16'},{key % 16}>"
            tf = fl[i] / (ms[i] * 1e-3) / 1e12
            gbs = by[i] / (ms[i] * 1e-3) / 1e9
            # which roofline bounds this shape: algorithmic FLOP per algorithmic byte against the machine balance
            intensity = fl[i] / by[i]
            bound = "hbm" if intensity < PEAK_TFLOPS_16BIT * 1e12 / (PEAK_HBM_GBS * 1e9) * 0.5 else "mfma"
            shapes.append({"kernel": kname, "epilogue": EPI_NAMES.get(key % 16), "N": N, "K": K, "launches": cnt, "big_launches": big,
                           "rows_per_launch": rows[i] / cnt, "ms_per_step": ms[i] / psteps, "tflops": tf,
                           "mfma_frac": tf / PEAK_TFLOPS_16BIT, "algorithmic_gbs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS,
                           "flop_per_byte": intensity, "bound": bound, "algorithmic_bytes_per_launch": by[i] / cnt})
            a = per_key.setdefault(key, [0.0, 0.0, 0.0, 0])
            a[0] += ms[i]; a[1] += fl[i]; a[2] += by[i]; a[3] += cnt
        shapes.sort(key=lambda s: -s["ms_per_step"])
        dom_key = max(per_key, key=lambda k: per_key[k][0])
        dom_ms, dom_fl, dom_bytes, dom_cnt = per_key[dom_key]
        dom_name = f"{FAMILY.get(dom_key // 32, 'gemm?')}<{'F16' if (dom_key // 16) % 2 == 1 else 'BF16'},{dom_key % 16}>"
        achieved = dom_fl / (dom_ms * 1e-3) / 1e12
        gemm_total_ms = sum(v[0] for v in per_key.values())
        gemm_total_fl = sum(v[1] for v in per_key.values())
        F = fwd_flops_per_seq(cfg)
        # algorithmic cost per sample: LEAF search (2 rho k candidate forwards + anchor + train fwd + 2x bwd), or for the
        # optional embedding-space PGD mode k x (fwd + input-gradient bwd ~ 1 + 1) on top of anchor + start fwd + train bwd
        flops_per_sample = (2 * args.rho * args.k_adv + 4) * F if args.attack == "leaf" else (2 * args.k_adv + 4) * F
        value = B * world * args.steps / dt
        traffic, traffic_note, traffic_kernels = None, "no PMC summary under profiles/ for this kernel build and workload", None
        # workload a PMC summary belongs to: tools/pmc_summary.py copies this key from the line of its own FETCH pass
        wkey = (f"{args.model}|B{B}|accum{args.accum_freq}|k{args.k_adv}|rho{args.rho}|{args.attack}|"
                f"{'dense' if args.dense else 'trimmed'}|{'noprefix' if args.no_prefix_reuse else 'prefix'}|{'fixedbatch' if args.fixed_batch else 'freshbatch'}"
                + ("" if arith == "rowsafe" else "|" + arith) + ("" if compact else "|fp32resid") + ("|preciseanchor" if args.precise_anchor else ""))
        default_key = "ViT-L-14-quickgelu|B128|accum1|k1|rho50|leaf|trimmed|prefix|freshbatch"
        try:    # HBM bytes per launch of the dominant kernel from the committed PMC passes; only for the build AND workload they were taken on
            import glob
            stale = False
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", TRAFFIC_GLOB))):
                with open(path) as f:
                    tj = json.load(f)
                if tj.get("workload_key", default_key) != wkey:
                    continue
                name = "profiles/" + os.path.basename(path)
                if tj.get("kernel_sources_sha16") != kernel_sources_hash():
                    stale = True
                    traffic_note = name + " was taken on different kernel sources: not attached"
                    continue
                traffic_kernels = tj.get("kernels")        # every big kernel: measured bytes / algorithmic bytes of the PMC run itself
                if tj.get("kernel") == dom_name:
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_note = ("HBM bytes per launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) * 1024, separate passes, "
                                    + name + " (same kernel sources, sha16 " + tj["kernel_sources_sha16"] + ", same workload; that run's own "
                                    f"algorithmic bytes per launch: {tj.get('algorithmic_bytes_per_launch')})")
                else:
                    traffic_note = name + " names " + str(tj.get("kernel")) + " as dominant, this line " + dom_name + ": per-kernel list attached only"
                break
            else:
                if not stale:
                    traffic_note = ("no PMC summary under profiles/ for this workload (" + wkey + "): tools/pmc_passes.sh + "
                                    "tools/pmc_summary.py produce it")
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "adversarial text samples/sec", "value": value, "unit": "samples/s",
            "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "host_threads_per_rank": args.host_threads,
            "dp_gradient_buckets": {"overlap_with_backward": os.environ.get("LEAF_DP_OVERLAP", "1") != "0", "active": bool(use_dist),
                                    "buckets_in_issue_order": dp_bucket_summary(model)},
            **({"per_rank": per_rank_summary(gathered)} if gathered is not None else {}),
            **({"rehearsal": f"gloo transport, {world} rank(s) on {ndev} device(s): functional rehearsal of the multi-rank step, "
                             "NOT the metric (RCCL, one device per rank, is)"} if rehearsal else {}),
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype + " MFMA operands, f32 accumulate",
            "data": "synthetic" + (" (one fixed batch for all steps: --fixed-batch)" if args.fixed_batch else " (a new batch per step, seeded by rank and step)"),
            "config": {"workload": (f"CLIP {args.model} text encoder, LEAF step k={args.k_adv} rho={args.rho}, "
                                    f"B={B} per GPU" + (f" x accum {args.accum_freq}" if args.accum_freq > 1 else "") +
                                    f", seq=77 (BASELINE.json configs[1]; {args.dtype} operands run the MFMA at the bf16 rate -- "
                                    "fp16's 11-bit significand is what meets the 1e-3 embedding tolerance)") if args.attack == "leaf" else
                                   (f"CLIP {args.model} text encoder, OPTIONAL embedding-space PGD mode (SURVEY 8a row a12, NOT "
                                    f"the reference's text attack): k={args.k_adv} steps, {args.pgd_norm} eps={args.pgd_eps} "
                                    f"alpha={args.pgd_alpha}, B={B} per GPU, seq=77"),
                       "attack": args.attack, "precision": arith, "split_masks": list(model.split_masks),
                       "residual_stream": ("16-bit copy + block-scaled e4m3 remainder byte (2^-16 per store) in the forward-only passes; fp32 in the training forward / backward"
                                           if compact else "fp32"), "accum_freq": args.accum_freq, "baseline_config_index": args.config or None, "workload_key": wkey,
                       "global_batch": B * world, "seq_len": cfg.context_length, "rho": args.rho, "k": args.k_adv,
                       "parallelism": f"dp{world}", "candidate_forwards_per_step_per_gpu": 2 * args.rho * args.k_adv * B},
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS_16BIT, "unit": "TFLOP/s",
                "frac": achieved / PEAK_TFLOPS_16BIT, "traffic": traffic, "traffic_note": traffic_note, "traffic_kernels": traffic_kernels,
                "kernel": f"{dom_name} {EPI_NAMES.get(dom_key % 16)}",
                "algorithmic_bytes_per_launch": dom_bytes / dom_cnt if dom_bytes else None,
                "launches": int(dom_cnt), "avg_launch_ms": dom_ms / dom_cnt,
                "algorithmic_gflop_per_launch": dom_fl / dom_cnt / 1e9,
                "all_gemm_tflops": gemm_total_fl / (gemm_total_ms * 1e-3) / 1e12,
                "gemm_share_of_step": gemm_total_ms * 1e-3 / (dt * psteps / args.steps),
                "sampled_steps": psteps,
                "shapes": shapes[:16],
            },
            "step_exec_frac": gemm_total_fl / (dt * psteps / args.steps) / (PEAK_TFLOPS_16BIT * 1e12),
            **step_trend(step_marks),
            "dense_equiv_frac": value * flops_per_sample / (world * PEAK_TFLOPS_16BIT * 1e12),
            "algorithmic_tflop_per_sample": flops_per_sample / 1e12,
            "exact_work_skipping": "none (dense, 77 rows per sequence)" if args.dense else
                                   "EOT trimming: rows after EOT are not computed (bit-identical outputs); "
                                   f"mean kept rows {sum(lens_seen) / len(lens_seen):.1f} of 77" +
                                   ("" if args.no_prefix_reuse else "; prefix reuse: rows before the edited token come "
                                    "from the clean caption's per-layer K/V cache (bit-identical outputs)"),
            "executed_gemm_tflop_per_step": gemm_total_fl / psteps / 1e12,
            "kernel_sources_sha16": kernel_sources_hash(),
            "loss": float(loss),
        }
    # ---- dense leg: the same step with every exact work-skipping switched off (outside the timed region, all ranks)
    if not args.no_dense_leg and not args.dense and args.attack == "leaf" and world == 1:
        n_dense = 3
        run_steps(1, None, False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(n_dense, None, False)
        torch.cuda.synchronize()
        d_dt = time.perf_counter() - t1
        if rank == 0:
            out["dense"] = {"value": B * n_dense / d_dt, "unit": "samples/s", "steps": n_dense, "ms_per_step": d_dt / n_dense * 1e3,
                            "note": "every sequence 77 rows, no prefix reuse (the reference's own amount of arithmetic)"}
    # ---- fast-arithmetic leg: the same step without the split GEMMs (the rounds 1-5 arithmetic), same box, outside the timed region --
    # what the default's per-row parity costs, readable from one line
    if not args.no_dense_leg and not args.dense and args.attack == "leaf" and world == 1 and model.split_masks:
        keep = model.split_masks
        model.set_precision("fast")
        frozen.set_precision("fast")          # (arithmetic only: the frozen model keeps its start weights -- the parity check below reads them)
        n_fast = 10
        run_steps(2, base_lens, not args.no_prefix_reuse)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(n_fast, base_lens, not args.no_prefix_reuse)
        torch.cuda.synchronize()
        f_dt = time.perf_counter() - t1
        model.set_split_masks(keep)
        frozen.set_split_masks(keep)
        if rank == 0:
            out["fast_arithmetic"] = {"value": B * n_fast / f_dt, "unit": "samples/s", "steps": n_fast, "ms_per_step": f_dt / n_fast * 1e3,
                                      "note": "no split GEMMs (--precision fast; weights as trained so far): batch rel-L2 8.7e-4 but 1.1 % of the "
                                              "embedding rows above 1e-3 (profiles/r06_row_error_census.txt); NOT the metric"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.attack == "leaf":
            oracle_name = args.model if args.model in ("ViT-L-14", "ViT-L-14-quickgelu", "ViT-H-14", "ViT-g-14", "ViT-bigG-14") else "ViT-L-14"
            same_weights = oracle_name == args.model          # the frozen model still holds the seed-1 start weights
            out["cpu_baseline"] = cpu_baseline(oracle_name, args.rho, args.k_adv, args.cpu_batch, seed=1234, budget_s=args.cpu_budget_s,
                                               gpu_encode=(lambda t: frozen.encode_text(t).cpu().numpy()) if same_weights else None,
                                               parity_rows=args.parity_rows)
            out["parity_rel_l2"] = out["cpu_baseline"].pop("parity_rel_l2")
        print(json.dumps(out), flush=True)
        if os.environ.get("LEAF_BENCH_JSON_OUT"):     # the PMC passes keep the line of THEIR run (tools/pmc_passes.sh): own algorithmic bytes
            with open(os.environ["LEAF_BENCH_JSON_OUT"], "w") as f:
                json.dump(out, f)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
