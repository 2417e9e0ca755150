"""One outer step of LEAF text-encoder adversarial fine-tuning on device tensors.

The step of utils_AT.py:282-366 with the host string work replaced by token-id tensors (the benchmark's
synthetic mode, SURVEY.md section 8d, and the inner engine of ``train_one_epoch_text_only``):

    anchor = frozen.encode_text(base)                                   utils_AT.py:296
    for _ in range(k):                                                  utils_attacks.py:310
        stage 1: score rho candidates per caption (random positions)    :316-348
        stage 2: score rho candidates at the winning position           :355-389
    feat = model.encode_text(adv)  (train mode, stash)                  utils_AT.py:317-319
    loss = mse(anchor, feat).sum(-1).mean(); backward                   :321-337
    [one flat RCCL all-reduce of the gradient buffer]                   SURVEY.md section 8e
    AdamW, zero_grad                                                    :339-362

Data parallelism: every rank holds a full replica and its own B captions; the only exchange is the gradient sum
(RCCL all-reduce) per optimizer step, issued per transformer block behind the backward so that it overlaps with it
(``GradReducer``; one flat all-reduce with LEAF_DP_OVERLAP=0); the 1/world factor is folded into the AdamW kernel's
``grad_scale``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch


@dataclass
class StepConfig:
    rho: int = 50
    k_adv: int = 1
    lr: float = 1e-5
    wd: float = 1e-4
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    accum_freq: int = 1
    # attack = "leaf": the reference's character search (default, the BASELINE.json metric).  attack = "pgd": the OPTIONAL
    # embedding-space PGD mode (not what the reference's text trainer runs): k_adv steps of
    # delta <- project(delta + pgd_alpha * normalize_grad(grad)) within the pgd_norm ball of radius pgd_eps.
    attack: str = "leaf"
    pgd_eps: float = 0.05
    pgd_alpha: float = 0.02
    pgd_norm: str = "linf"
    # frozen model's anchor pass (utils_AT.py:296) in the fp32-grade arithmetic of precise.hip instead of the 16-bit arithmetic the
    # candidates are scored in.  Off by default: the reference computes anchor and candidates in the SAME reduced arithmetic (TF32),
    # and at equal weights the operand-rounding error common to both sides cancels in the search's ||f_cand - anchor||^2
    # (profiles/r06_precise_anchor_ab.txt prices both)
    precise_anchor: bool = False


class SyntheticCandidates:
    """Stand-in for the host string mutation on token ids: candidates are copies of the caption's token row with one
    position resampled (stage 1: rho random positions; stage 2: rho random ids at the stage-1 winner's position).
    As in the real search the HOST draws the positions (it needs them to plan the rows to compute) and learns the
    stage-1 winners through one small device-to-host copy per stage; the replacement ids are drawn on the device."""

    def __init__(self, base: torch.Tensor, base_lens, rho: int, vocab: int, seed: int):
        import numpy as np
        self.rho, self.vocab = rho, vocab
        self.gen = torch.Generator(device=base.device)
        self.gen.manual_seed(seed)
        self.rng = np.random.default_rng(seed)
        self.B, self.ctx = base.shape
        if base_lens is None:
            base_lens = (base.argmax(-1) + 1).cpu().numpy()
        self.inner = np.asarray(base_lens, dtype=np.int64) - 2          # ids between SOT and EOT

    def _draw(self, cur):
        """Everything of a stage that does not depend on the positions: the rho copies of the caption rows and the
        replacement ids (drawn on the device).  Stage 2 calls this BEFORE the host waits for the stage-1 winners, so the
        device has work queued while the host plans the rows."""
        B, rho = self.B, self.rho
        ids = torch.randint(1, self.vocab - 2, (B, rho), device=cur.device, generator=self.gen, dtype=torch.int32)
        return cur[:, None, :].repeat(1, rho, 1), ids

    def stage1(self, cur: torch.Tensor):
        import numpy as np
        pos = 1 + (self.rng.random((self.B, self.rho)) * self.inner[:, None]).astype(np.int64)
        cand, ids = self._draw(cur)
        self.pos1_dev = torch.from_numpy(pos).pin_memory().to(cand.device, non_blocking=True)
        cand.scatter_(2, self.pos1_dev[:, :, None], ids[:, :, None])
        return cand, pos

    def stage2_device(self, cur: torch.Tensor, best1: torch.Tensor):
        """The stage-2 candidate ids, built ON THE DEVICE from the stage-1 arg-max (no host round trip): queued behind stage 1
        BEFORE the host waits for the winners, so after that wait only the row plan is left to do."""
        cand, ids = self._draw(cur)
        p = self.pos1_dev.gather(1, best1.to(torch.int64)[:, None])          # [B, 1] winning positions
        cand.scatter_(2, p[:, :, None].expand(self.B, self.rho, 1), ids[:, :, None])
        return cand

    def stage2_positions(self, pos, best1_host):
        """Host copy of the stage-2 edit positions (the prefix lengths of the row plan)."""
        import numpy as np
        p = pos[np.arange(self.B), best1_host]
        return np.repeat(p[:, None], self.rho, axis=1)


def search_synthetic(model, anchor: torch.Tensor, base: torch.Tensor, cfg: StepConfig, seed: int,
                     base_lens=None, prefix_reuse: bool = True, anchor_ready=None) -> torch.Tensor:
    """2k calls of score_candidates on [B*rho, ctx] synthetic candidates; returns the adversarial ids [B, ctx].
    ``base_lens`` (host int array, EOT position + 1 per caption) enables EOT trimming (the synthetic edits never move
    EOT); with ``prefix_reuse`` the clean captions' per-layer K/V are cached once per edit and every candidate only
    recomputes the positions from its edited token on (the edit position is the prefix length)."""
    import numpy as np
    cand_lens = None if base_lens is None else np.repeat(np.asarray(base_lens, dtype=np.int32), cfg.rho)
    gen = SyntheticCandidates(base, base_lens, cfg.rho, model.cfg.vocab_size, seed)
    reuse = prefix_reuse and base_lens is not None and getattr(model, "trim_rows", False)
    cur = base
    B = base.shape[0]
    ar = torch.arange(B, device=base.device)
    fuse = reuse and os.environ.get("LEAF_FUSE_KV", "1") != "0"
    for _ in range(cfg.k_adv):
        cand, pos = gen.stage1(cur)
        if anchor_ready is not None:
            torch.cuda.current_stream().wait_event(anchor_ready)
            anchor_ready = None
        # stage 1 with the clean captions' K/V pass fused in (their rows ride in the candidates' launches); falls back to the
        # separate pass when the rows do not fit one chunk
        fused = model.score_candidates_fused(cur, base_lens, cand.view(B * cfg.rho, -1), anchor, cfg.rho, cand_lens,
                                             pos.reshape(-1)) if fuse else None
        if fused is not None:
            best1, _, kv = fused
        else:
            kv = model.encode_text_kv(cur, seq_lens=base_lens) if reuse else None
            best1, _ = model.score_candidates(cand.view(B * cfg.rho, -1), anchor, cfg.rho, "l2", want_features=False,
                                              seq_lens=cand_lens, prefix_lens=pos.reshape(-1) if reuse else None, kv=kv)
        if os.environ.get("LEAF_DIAG_FIXED_WINNER") == "1":
            best1 = torch.zeros_like(best1)      # diagnostic A/B runs only: the same row plan whatever the (garbage) scores say
        cand = gen.stage2_device(cur, best1)                             # queued behind stage 1, before the host waits
        # ... and everything of the second stage's call that does not need the winners (the device has nothing to run between their
        # arrival on the host and that call's first launch)
        plan2 = model.score_candidates_prepare(cand.view(B * cfg.rho, -1), anchor, cfg.rho, "l2", want_features=False,
                                               seq_lens=cand_lens, kv=kv) if reuse and kv is not None else None
        pos2 = gen.stage2_positions(pos, best1.cpu().numpy())           # the search's device->host sync (B indices)
        if plan2 is not None:
            best2, _ = model.score_candidates_run(plan2, pos2)
        else:
            best2, _ = model.score_candidates(cand.view(B * cfg.rho, -1), anchor, cfg.rho, "l2", want_features=False,
                                              seq_lens=cand_lens, prefix_lens=pos2.reshape(-1) if reuse else None, kv=kv)
        cur = cand[ar, best2.to(torch.int64)]
    return cur


_SIDE = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def _dp_active() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("LEAF_BENCH_FORCE_DIST") == "1")


def allreduce_grads(model) -> float:
    """ONE flat all-reduce (sum) of the gradient buffer over RCCL; returns the factor AdamW must apply."""
    import torch.distributed as dist
    if _dp_active():
        dist.all_reduce(model.grads, op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()
    return 1.0


def bucket_plan(layout, n_params: int, layers: int, min_bytes: int = 64 << 20):
    """Gradient buckets in the order the backward finishes them.  The four GEMM weights of a transformer block are one
    contiguous 12 d^2 run of the flat buffer (28 MB for ViT-L) and consecutive blocks are adjacent, so a bucket is the run of
    one or more consecutive blocks, grown downwards from block L-1 until it holds at least ``min_bytes`` (fewer, larger
    collectives: ViT-L -> 3 blocks = 85 MB per bucket, bigG -> 1 block = 79 MB); its event is the LOWEST block's, the one
    that finishes last.  ONE last bucket list takes everything else (embedding tables + text_projection in front of the block
    weights; every bias / LayerNorm vector + ln_final behind them), complete only after the backward's last kernel.
    Returns [(event index, [(offset, numel), ...]), ...]; the ranges partition [0, n_params) (tests/test_dp_gloo.py)."""
    def span(l):
        p = f"transformer.resblocks.{l}."
        names = [p + "attn.in_proj_weight", p + "attn.out_proj.weight", p + "mlp.c_fc.weight", p + "mlp.c_proj.weight"]
        lo = min(layout[k][0] for k in names)
        hi = max(layout[k][0] + int(np.prod(layout[k][1])) for k in names)
        return lo, hi
    plan = []
    w_lo, w_hi = span(0)[0], span(layers - 1)[1]
    hi = None
    for l in reversed(range(layers)):
        lo, h_ = span(l)
        if hi is None:
            hi = h_
        if (hi - lo) * 4 >= min_bytes or l == 0:
            plan.append((l, [(lo, hi - lo)]))
            hi = None
        else:
            assert span(l - 1)[1] == lo, "block weights are expected back to back in the flat layout"
    plan.append((layers, [(0, w_lo), (w_hi, n_params - w_hi)]))
    return plan


def reduce_token_rows(table: torch.Tensor, my_tokens: torch.Tensor, cap: int, group=None) -> None:
    """Data-parallel SUM of the token-embedding gradient ``table`` [vocab, d] that moves only the rows some rank touched
    (LEAF_DP_SPARSE_EMBED=1).  A rank's gradient is non-zero in the rows of ITS captions' token ids only (a few
    thousand of 49,408), yet the dense all-reduce ships the whole 152-MB table over point-to-point xGMI -- the one collective of
    the step that cannot hide behind the backward.  Here every rank
      1. flags its ids in a [vocab] int32 vector and all-reduces it with MAX (198 KB): the union of touched rows, identical everywhere;
      2. takes the first ``cap`` ids of a stable descending sort of the flags -- the touched ids in ascending order, then untouched ones
         (whose rows are zero on every rank: harmless filler) -- a FIXED-size, host-known selection (no device-to-host sync);
      3. gathers those rows, all-reduces the [cap, d] block (SUM), scatters the sums back.
    ``cap`` must bound the size of the union (the caller derives it from an all-reduced MAX of the per-rank token counts); row 0 is
    always part of the selection (the fp16 backward's saturation poison lives in element [0, 0] and must reach every rank).
    Same result as ``all_reduce(table)`` on the touched rows, untouched rows stay zero; standard collectives only."""
    import torch.distributed as dist
    vocab = table.shape[0]
    cap = int(min(max(cap, 1), vocab))
    flag = torch.zeros(vocab, dtype=torch.int32, device=table.device)
    flag[my_tokens.reshape(-1).to(torch.int64).clamp_(0, vocab - 1)] = 1
    flag[0] = 1
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    sel = torch.argsort(flag, descending=True, stable=True)[:cap]
    block = table.index_select(0, sel)
    dist.all_reduce(block, op=dist.ReduceOp.SUM, group=group)
    table.index_copy_(0, sel, block)


class GradReducer:
    """Data-parallel gradient mean overlapped with the backward.  The backward records one
    event per transformer block as soon as that block's gradients are final (leaf_textfare_backward_events); the weight
    buckets (runs of consecutive blocks, >= 64 MB each: bucket_plan) are all-reduced (RCCL, sum) on a side stream while the
    backward of the earlier blocks is still running; only the last bucket (embedding tables + vectors) is exposed.  Every rank issues the same collectives in the
    same order.  LEAF_DP_OVERLAP=0 (or world size 1) falls back to ONE flat all-reduce after the backward.
    The 1/world factor is folded into the AdamW kernel (``finish()`` returns it)."""

    def __init__(self, model):
        self.model = model
        self.plan = bucket_plan(model.layout, model.n_params, model.cfg.layers)
        self.overlap = os.environ.get("LEAF_DP_OVERLAP", "1") != "0"
        # LEAF_DP_SPARSE_EMBED=1 (opt-in: no multi-GPU box has run it): the token-embedding table of the exposed last bucket is
        # reduced by touched rows (reduce_token_rows) instead of densely; needs the overlapped form and one micro-batch per step
        self.sparse_embed = os.environ.get("LEAF_DP_SPARSE_EMBED", "0") == "1"
        self._cap_host = None      # pinned [1] int64: all-reduced MAX of the per-rank token counts of the current step
        self._cap_ev = None
        self.events = None
        self.comm = None
        self._launched = False
        # timing = True (bench.py at world > 1): an event pair around every finish(), i.e. around what the step's stream really
        # waits for -- the EXPOSED part of the reduction, arrival skew of the other ranks included
        self.timing = False
        self._pairs = []

    def _setup(self):
        dev = self.model.device
        self.comm = torch.cuda.Stream(device=dev)
        self.events = [torch.cuda.Event() for _ in range(self.model.cfg.layers + 1)]
        for e in self.events:          # an event's handle exists only once it has been recorded
            e.record(torch.cuda.current_stream(dev))

    def begin_step(self, n_tokens: int):
        """Sparse embedding reduction only: all-reduce (MAX) this rank's token count of the step on the comm stream NOW -- it is long
        finished when the backward's reduction needs it on the host to size its fixed [cap, d] block (no device-to-host wait then)."""
        import torch.distributed as dist
        if not (self.sparse_embed and _dp_active() and self.overlap):
            return
        if self.events is None:
            self._setup()
        dev = self.model.device
        if self._cap_host is None:
            self._cap_host = torch.zeros(1, dtype=torch.int64).pin_memory()
            self._cap_ev = torch.cuda.Event()
        src = torch.tensor([int(n_tokens)], dtype=torch.int64).pin_memory()     # (pinned + non_blocking: the host must not wait for the comm stream)
        with torch.cuda.stream(self.comm):
            t = src.to(dev, non_blocking=True)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            self._cap_host.copy_(t, non_blocking=True)
            self._cap_ev.record(self.comm)
        self._cap_pending = True

    def backward(self, feat, anchor, accum_scale: float = 1.0, last_micro: bool = True):
        """model.backward + (on the last micro-batch of an optimizer step) the bucketed reduction queued behind it."""
        import torch.distributed as dist
        m = self.model
        if not (_dp_active() and self.overlap and last_micro):
            return m.backward(feat, anchor, accum_scale=accum_scale)
        if self.events is None:
            self._setup()
        loss = m.backward(feat, anchor, accum_scale=accum_scale, layer_events=self.events)
        sparse = self.sparse_embed and getattr(self, "_cap_pending", False) and accum_scale == 1.0
        for ev_idx, ranges in self.plan:
            self.comm.wait_event(self.events[ev_idx])
            with torch.cuda.stream(self.comm):
                if sparse and ev_idx == m.cfg.layers:
                    # the exposed bucket: the token-embedding table by touched rows, the rest of the bucket densely
                    off_t, shape_t = m.layout["token_embedding.weight"]
                    n_t = shape_t[0] * shape_t[1]
                    self._cap_ev.synchronize()
                    self._cap_pending = False
                    cap = dist.get_world_size() * (int(self._cap_host[0]) + 1)
                    reduce_token_rows(m.grads[off_t:off_t + n_t].view(shape_t), m._train_tokens, cap)
                    ranges = [(o, c) for (o, c) in _minus(ranges, off_t, n_t)]
                for off, numel in ranges:
                    if numel:
                        dist.all_reduce(m.grads[off:off + numel], op=dist.ReduceOp.SUM)
        self._launched = True
        return loss

    def finish(self) -> float:
        """Make the current stream wait for the reduction (launching the flat one if none is in flight); returns the
        factor AdamW must apply to the summed gradients."""
        import torch.distributed as dist
        cur = torch.cuda.current_stream(self.model.device)
        e0 = None
        if self.timing and _dp_active():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        if self._launched:
            cur.wait_stream(self.comm)
            self._launched = False
            scale = 1.0 / dist.get_world_size()
        else:
            scale = allreduce_grads(self.model)
        if e0 is not None:
            e1.record(cur)
            self._pairs.append((e0, e1))
        return scale

    def exposed_ms(self) -> float:
        """Sum over the finish() calls since the last call of the time the step's stream spent waiting for the gradient
        reduction (synchronises on the recorded events)."""
        tot = 0.0
        for e0, e1 in self._pairs:
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        self._pairs = []
        return tot


def _minus(ranges, off, n):
    """``ranges`` ([(offset, numel)]) without the span [off, off + n)"""
    out = []
    for o, c in ranges:
        lo, hi = o, o + c
        if hi <= off or lo >= off + n:
            out.append((o, c))
            continue
        if lo < off:
            out.append((lo, off - lo))
        if hi > off + n:
            out.append((off + n, hi - (off + n)))
    return out


def reduce_buckets(flat: torch.Tensor, plan) -> None:
    """The collectives of GradReducer without streams / events (CPU rehearsal under gloo, tests/test_dp_gloo.py)."""
    import torch.distributed as dist
    for _, ranges in plan:
        for off, numel in ranges:
            if numel:
                dist.all_reduce(flat[off:off + numel], op=dist.ReduceOp.SUM)


def get_reducer(model) -> GradReducer:
    """The model's reducer, created on first use and stored ON the model, so that it (its comm stream, events) dies with the
    model instead of pinning every model ever trained in a module-level table."""
    r = getattr(model, "_grad_reducer", None)
    if r is None:
        r = model._grad_reducer = GradReducer(model)
    return r


def train_step_tokens(model, frozen, base: torch.Tensor, cfg: StepConfig, seed: int, lr: Optional[float] = None,
                      micro_index: int = 0, base_lens=None, prefix_reuse: bool = True,
                      base_ready: Optional["torch.cuda.Event"] = None, optimizer_step: Optional[bool] = None) -> torch.Tensor:
    """Full outer step on token ids (int32 [B, ctx] on the model's device).  Returns the TextFARE loss (0-d).
    ``base_ready``: event after which ``base`` is valid.  The anchor forward depends only on ``base`` and the FROZEN
    weights, so with this event its side stream need not wait for the previous step's backward / all-reduce / AdamW
    still queued on the current stream and the anchor of step i+1 overlaps the tail of step i; without it the side
    stream waits for everything queued so far (always safe).  ``optimizer_step``: None = after the last micro-batch of
    ``cfg.accum_freq`` (the trainer's rule); False = never (the caller reduces / steps itself: bench.py --rank-sim)."""
    model.eval()
    if _dp_active():      # (sparse embedding reduction: the per-rank token count travels now, section 6; a no-op otherwise)
        n_tok = int(np.asarray(base_lens).sum()) if base_lens is not None else base.shape[0] * base.shape[1]
        get_reducer(model).begin_step(n_tok)
    # The frozen model's anchor forward depends on nothing but the captions and the FROZEN weights: it runs on a side stream and the
    # search waits for it right before its first scoring call.  The side stream also waits for the PREVIOUS step's search
    # (LEAF_ANCHOR_AT=tail, the default): the host runs a stage ahead of the device, and without that ordering the anchor's ~85 small
    # launches land in the middle of the previous step's second stage, beside its big persistent GEMMs, instead of beside that
    # step's training pass (small launches, AdamW).  Round 4, same-box A/B (profiles/r04_anchor_ab.txt): 50.41 against 50.54 ms per
    # step; queueing the NEXT step's anchor into the hole at the stage boundary, where the device waits for the host to read the
    # winners, was tried too and is slower (50.9: it then runs beside the start of stage 2).  LEAF_ANCHOR_AT=free: no ordering.
    cur = torch.cuda.current_stream()
    side = _side_stream(base.device)
    if base_ready is not None:
        side.wait_event(base_ready)
        cur.wait_event(base_ready)      # the search reads ``base`` on this stream (a batch built on a third stream, one step ahead)
    else:
        side.wait_stream(cur)
    prev = getattr(model, "_search_done", None)
    if prev is not None and os.environ.get("LEAF_ANCHOR_AT", "tail") == "tail":
        side.wait_event(prev)
    with torch.cuda.stream(side):
        anchor = frozen.encode_text(base, seq_lens=base_lens, precise=cfg.precise_anchor)
        ready = torch.cuda.Event()
        ready.record(side)
    anchor.record_stream(cur)
    if cfg.attack == "pgd":
        from .attacks import attack_embedding_pgd
        cur.wait_event(ready)
        # the attack's last forward is the training forward of the perturbed captions: its stash feeds the backward
        feat, _ = attack_embedding_pgd(model, base, anchor, cfg.pgd_eps, cfg.pgd_alpha, k=cfg.k_adv, norm=cfg.pgd_norm,
                                       seq_lens=base_lens if base_lens is not None else np.full(base.shape[0], base.shape[1]),
                                       seed=seed)
        model.train()
    else:
        adv = search_synthetic(model, anchor, base, cfg, seed, base_lens=base_lens, prefix_reuse=prefix_reuse,
                               anchor_ready=ready)
        done = torch.cuda.Event()
        done.record(cur)
        model._search_done = done
        model.train()
        feat = model.forward_train(adv, seq_lens=base_lens)
    if micro_index % cfg.accum_freq == 0:
        model.zero_grad()
    last = (micro_index + 1) % cfg.accum_freq == 0 if optimizer_step is None else bool(optimizer_step)
    red = get_reducer(model)
    loss = red.backward(feat, anchor, accum_scale=1.0 / cfg.accum_freq, last_micro=last)   # + bucketed all-reduce behind it
    if last:
        scale = red.finish()
        model.adamw_step(lr if lr is not None else cfg.lr, (cfg.beta1, cfg.beta2), cfg.eps, cfg.wd, grad_scale=scale)
        model.pack()
    return loss
