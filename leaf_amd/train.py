"""Epoch loop, schedule, optimizer wrapper, text-only data readers and checkpointing for the LEAF text trainer.

Mirrors the reference's host orchestration around the hot path:

* ``train_one_epoch_text_only``  utils_AT.py:262-426 (same argument list; ``loss`` and ``scaler`` are accepted and
  unused -- the reference never uses ``loss`` either; the fp16 gradient path carries its own device-side power-of-two loss scale)
* ``cosine_lr`` / ``const_lr``   src/open_clip_train/scheduler.py:4-53
* ``LeafAdamW``                  torch.optim.AdamW with the two groups of train_AT_text_only.py:323-341, executed
  by ONE fused HIP kernel over the flat parameter buffer after ONE flat RCCL all-reduce
* ``get_text_data``              captions only; the reference decodes, augments and then discards every image
  (data_AT.py:499-503, utils_AT.py:290)
* checkpoint / resume            train_AT_text_only.py:351-372,516-525
"""
from __future__ import annotations

import csv
import glob
import io
import logging
import math
import os
import random
import re
import tarfile
import time
from typing import Iterator, List, Optional

import numpy as np
import torch

from .attacks import DEFAULT_V, attack_text
from .step import allreduce_grads, get_reducer

LATEST_CHECKPOINT_NAME = "epoch_latest.pt"


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def is_master(args, local=False):
    return (getattr(args, "local_rank", 0) if local else getattr(args, "rank", 0)) == 0


def unwrap_model(model):
    return model.module if hasattr(model, "module") else model


# ----------------------------------------------------------------------------- schedule
def _assign_lr(optimizer, lr):
    for g in optimizer.param_groups:
        g["lr"] = lr


def cosine_lr(optimizer, base_lr, warmup_length, steps):
    def _lr_adjuster(step):
        if step < warmup_length:
            lr = base_lr * (step + 1) / warmup_length
        else:
            e, es = step - warmup_length, steps - warmup_length
            lr = 0.5 * (1 + np.cos(np.pi * e / es)) * base_lr
        _assign_lr(optimizer, lr)
        return lr
    return _lr_adjuster


def const_lr(optimizer, base_lr, warmup_length, steps):
    def _lr_adjuster(step):
        lr = base_lr * (step + 1) / warmup_length if step < warmup_length else base_lr
        _assign_lr(optimizer, lr)
        return lr
    return _lr_adjuster


def const_lr_cooldown(optimizer, base_lr, warmup_length, steps, cooldown_steps, cooldown_power=1.0, cooldown_end_lr=0.):
    """--lr-scheduler const-cooldown (open_clip_train/scheduler.py:24-41): warm-up, constant, then over the last ``cooldown_steps``
    a polynomial decay (1 - e / es) ** power from base_lr to cooldown_end_lr."""
    def _lr_adjuster(step):
        start = steps - cooldown_steps
        if step < warmup_length:
            lr = base_lr * (step + 1) / warmup_length
        elif step < start:
            lr = base_lr
        else:
            e, es = step - start, steps - start
            lr = (1 - (e / es)) ** cooldown_power * (base_lr - cooldown_end_lr) + cooldown_end_lr
        _assign_lr(optimizer, lr)
        return lr
    return _lr_adjuster


def make_scheduler(args, optimizer, total_steps, num_batches):
    """train_AT_text_only.py:384-401: --lr-scheduler cosine | const | const-cooldown (the latter needs --epochs-cooldown)."""
    if args.lr_scheduler == "cosine":
        return cosine_lr(optimizer, args.lr, args.warmup, total_steps)
    if args.lr_scheduler == "const":
        return const_lr(optimizer, args.lr, args.warmup, total_steps)
    if args.lr_scheduler == "const-cooldown":
        assert args.epochs_cooldown is not None, "Please specify the number of cooldown epochs for this lr schedule."
        cooldown_steps = (num_batches // args.accum_freq) * args.epochs_cooldown
        return const_lr_cooldown(optimizer, args.lr, args.warmup, total_steps, cooldown_steps, args.lr_cooldown_power, args.lr_cooldown_end)
    raise ValueError(f"Unknown scheduler, {args.lr_scheduler}. Available options are: cosine, const, const-cooldown.")


# ----------------------------------------------------------------------------- optimizer
class LeafAdamW:
    """AdamW over the engine's flat buffers.  ``param_groups`` mirrors the reference's two groups (excluded tensors
    first with weight_decay 0, the rest with ``--wd``) so schedulers and loggers that poke ``param_groups[0]['lr']``
    work unchanged."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lock_image=False):
        # lock_image: --lock-image froze the image tower BEFORE the reference built its optimizer (train_AT_text_only.py:286-290),
        # so that run's optimizer state has text-only groups; otherwise the groups span the whole CLIP (checkpoint.reference_param_groups)
        self.lock_image = bool(lock_image)
        self.model = model.enable_training()
        nd = model.n_decay
        excl = [k for k, (off, _) in model.layout.items() if off >= nd]
        rest = [k for k, (off, _) in model.layout.items() if off < nd]
        self.param_groups = [
            {"params": excl, "weight_decay": 0.0, "lr": lr, "betas": tuple(betas), "eps": eps},
            {"params": rest, "weight_decay": weight_decay, "lr": lr, "betas": tuple(betas), "eps": eps}]

    def step(self, max_norm=None, reduced=False):
        """``max_norm``: --grad-clip-norm (utils_AT.py:348-357), applied to the all-reduced, averaged gradients like
        clip_grad_norm_ on DDP-averaged .grad; returns the total norm (0-d tensor) when clipping.  ``reduced``: the gradient
        buffer already holds the clipped mean over ranks (micro-batch clipping, ``MicroClip`` below)."""
        g = self.param_groups[1]
        # waits for the bucketed reduction queued behind the backward (or runs the flat one)
        scale = 1.0 if reduced else get_reducer(self.model).finish()
        total = self.model.adamw_step(g["lr"], g["betas"], g["eps"], g["weight_decay"], grad_scale=scale, max_norm=max_norm)
        self.model.pack()
        return total

    def zero_grad(self):
        self.model.zero_grad()

    def state_dict(self):
        """torch.optim.AdamW.state_dict() of the reference's optimizer (two groups in its parameter order, per-parameter
        step / exp_avg / exp_avg_sq views of the flat moments): what --resume of the reference loads (train_AT_text_only.py:366)."""
        from .checkpoint import image_tower_is_known, non_text_parameters, optimizer_state_to_torch
        m, g = self.model, self.param_groups[1]
        extra = None if self.lock_image else m.extra_state
        if extra and not image_tower_is_known(extra):
            # e.g. a timm trunk: which of its tensors are parameters cannot be told from the names, and a wrong guess would
            # SHIFT the parameter ids of the full-CLIP layout silently
            unknown = [k for k in extra if not image_tower_is_known({k: 0})][:3]
            logging.warning(f"optimizer state_dict: the start checkpoint's image tower is not one of open_clip's ViT / ModifiedResNet "
                            f"(e.g. {unknown}); writing the TEXT-ONLY groups, which the reference resumes only with --lock-image")
            extra = None
        n_vis = len(non_text_parameters(extra))
        logging.info("optimizer state_dict: " + (f"the reference's groups over the whole CLIP ({n_vis} carried-through non-text parameters "
                                                 "hold ids without state)" if n_vis else
                                                 "TEXT-ONLY groups (" + ("--lock-image" if self.lock_image else "the start checkpoint "
                                                 "names no image tower") + "): the reference resumes it only with --lock-image"))
        return optimizer_state_to_torch(m.layout, m.cfg.layers, m.exp_avg, m.exp_avg_sq, m.applied_steps(), g["lr"], g["betas"], g["eps"],
                                        g["weight_decay"], lrs=(self.param_groups[0]["lr"], g["lr"]), extra=extra)

    def load_state_dict(self, sd):
        """Accepts the torch AdamW layout (written here or by the reference) and the flat layout of round-1 checkpoints."""
        m = self.model
        if "state" in sd and "param_groups" in sd:
            from .checkpoint import optimizer_state_from_torch, optimizer_state_layout
            # m.extra_state = the non-text tensors of the checkpoint just loaded: they name the visual.* ids of a full-CLIP layout
            kind = optimizer_state_layout(sd, m.cfg.layers, m.extra_state)
            logging.info(f"optimizer state_dict read in the {kind} layout")
            m.set_applied_steps(optimizer_state_from_torch(sd, m.layout, m.cfg.layers, m.exp_avg, m.exp_avg_sq, extra=m.extra_state))
            saved = sd["param_groups"]
        elif "exp_avg" in sd:
            m.set_applied_steps(int(sd["step"]))
            m.exp_avg.copy_(sd["exp_avg"])
            m.exp_avg_sq.copy_(sd["exp_avg_sq"])
            saved = sd["param_groups"]
        else:
            raise ValueError("unrecognised optimizer state (neither torch.optim.AdamW.state_dict() nor the flat layout)")
        for g, s in zip(self.param_groups, saved):
            g.update({k: v for k, v in s.items() if k in ("lr", "betas", "eps", "weight_decay")})
            g["betas"] = tuple(g["betas"])


# ----------------------------------------------------------------------------- data (captions only)
def _expand_braces(pattern: str) -> List[str]:
    """'{00000000..00001287}.tar' style ranges and '::'-separated sources (what braceexpand does for the scripts)."""
    out = []
    for part in pattern.split("::"):
        m = re.search(r"\{(\d+)\.\.(\d+)\}", part)
        if not m:
            out += sorted(glob.glob(part)) or [part]
            continue
        a, b, width = int(m.group(1)), int(m.group(2)), len(m.group(1))
        for i in range(a, b + 1):
            out += _expand_braces(part[:m.start()] + str(i).zfill(width) + part[m.end():])
    return out


def _scan_tar_captions(path: str) -> Iterator[str]:
    """The .txt members of an uncompressed ustar archive, read WITHOUT materialising the other members: a 512-byte header at a
    time, image / json payloads skipped with one seek each (the reference decodes every image and throws it away,
    data_AT.py:499-503).  Anything this scan does not understand -- compression, GNU long names, pax records, base-256 sizes --
    raises _NotPlainTar before a single caption has been yielded and the shard goes through ``tarfile`` instead."""
    with open(path, "rb") as f:
        first = True
        while True:
            hdr = f.read(512)
            if len(hdr) < 512 or hdr.count(0) == 512:
                if first and len(hdr) < 512:
                    raise _NotPlainTar(path)
                return
            if first and hdr[257:262] != b"ustar":
                raise _NotPlainTar(path)
            typeflag, size_field = hdr[156:157], hdr[124:136]
            if typeflag not in (b"0", b"\0", b"5") or size_field[0] & 0x80:
                if first:
                    raise _NotPlainTar(path)
                raise tarfile.TarError(f"unsupported member type {typeflag!r} in the middle of a plain shard")
            first = False
            size = int(size_field.rstrip(b"\0 ") or b"0", 8)
            padded = (size + 511) & ~511
            if typeflag != b"5" and hdr[:100].rstrip(b"\0").endswith(b".txt"):
                data = f.read(padded)
                if len(data) < size:
                    raise tarfile.TarError("unexpected end of data")
                yield data[:size].decode("utf-8", errors="replace").strip()
            else:
                f.seek(padded, 1)


class _NotPlainTar(Exception):
    pass


def _iter_tar_captions(paths: List[str]) -> Iterator[str]:
    for path in paths:
        try:
            try:
                yield from _scan_tar_captions(path)
                continue
            except _NotPlainTar:
                pass
            with tarfile.open(path) as tf:
                for member in tf:
                    if member.isfile() and member.name.endswith(".txt"):
                        yield tf.extractfile(member).read().decode("utf-8", errors="replace").strip()
        except (tarfile.TarError, OSError, ValueError) as e:   # data_AT.py:285-288 log_and_continue
            logging.warning(f"skipping shard {path}: {e}")


class TextLoader:
    """Iterable of ``(None, texts)`` batches (the reference's batches are ``(images, texts)`` with the images
    discarded at utils_AT.py:290).  Sharded across ranks by stride; reshuffled per epoch with seed + epoch."""

    def __init__(self, captions: Optional[List[str]], shards: Optional[List[str]], batch_size, num_samples, rank, world,
                 seed, prefetch: int = 4):
        self.captions, self.shards = captions, shards
        self.batch_size, self.rank, self.world, self.seed = batch_size, rank, world, seed
        self.epoch = 0
        self.num_samples = num_samples
        self.num_batches = max(1, math.ceil(num_samples / (batch_size * world)))
        self.prefetch = prefetch          # batches read ahead by the background reader (0: synchronous)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _stream(self) -> Iterator[str]:
        """Endless stream of this rank's captions.  A full pass that yields nothing (missing / unreadable shards, no .txt
        members, fewer captions than ranks) raises instead of spinning forever -- under data parallelism a rank that hangs
        here would deadlock the others' all-reduce."""
        rng = random.Random(self.seed + self.epoch)
        if self.captions is not None:
            idx = list(range(len(self.captions)))
            rng.shuffle(idx)
            mine = idx[self.rank::self.world] or idx
            if not mine:
                raise RuntimeError("the caption source is empty")
            while True:
                for i in mine:
                    yield self.captions[i]
        else:
            shards = list(self.shards)
            rng.shuffle(shards)
            mine = shards[self.rank::self.world] or shards
            while True:
                n = 0
                for cap in _iter_tar_captions(mine):
                    n += 1
                    yield cap
                if n == 0:
                    raise RuntimeError(f"rank {self.rank}: no caption (.txt member) found in a full pass over {len(mine)} shard(s), "
                                       f"first: {mine[0] if mine else '<none>'} -- check --train-data")

    def _batches(self):
        it = self._stream()
        for _ in range(self.num_batches):
            yield None, [next(it) for _ in range(self.batch_size)]

    def __iter__(self):
        """The epoch's batches, read AHEAD of the training step by one background thread through a bounded queue (the
        reference overlaps loading with worker processes: data_AT.py:455-503, --workers).  The training thread spends its time
        inside ctypes calls and device waits, which release the interpreter lock, so one thread keeps up (tools/loader_bench.py:
        captions/s out of image-sized shards).  ``prefetch = 0`` reads synchronously.  The consumer may stop early: the
        finally block stops and joins the reader."""
        if self.prefetch <= 0:
            yield from self._batches()
            return
        import queue
        import threading
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        END = object()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def reader():
            try:
                for b in self._batches():
                    if not put(b):
                        return
                put(END)
            except BaseException as e:     # surfaces in the training thread (an empty pass must not hang the other ranks)
                put(e)

        t = threading.Thread(target=reader, name="leaf-text-loader", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            t.join(timeout=10)

    def __len__(self):
        return self.num_batches


class DataInfo:
    def __init__(self, loader: TextLoader):
        self.dataloader = loader

    def set_epoch(self, epoch):
        self.dataloader.set_epoch(epoch)


_SYN_WORDS = ("a photo of the small red car on wet street with two people near old house in sunny park at night "
              "dog cat bird tree river bridge mountain snow beach city table chair").split()


def get_text_data(args, epoch=0):
    rank, world = getattr(args, "rank", 0), getattr(args, "world_size", 1)
    kind, src = args.dataset_type, args.train_data
    if kind == "auto":
        kind = "synthetic" if not src else ("webdataset" if src.endswith(".tar") or "{" in src else
                                             ("csv" if src.endswith((".csv", ".tsv")) else "text"))
    captions = shards = None
    if kind == "synthetic":
        rng = random.Random(args.seed)
        n = args.train_num_samples or 1024
        captions = [" ".join(rng.choice(_SYN_WORDS) for _ in range(rng.randint(3, 14))) for _ in range(n)]
    elif kind == "text":
        with open(src) as f:
            captions = [l.strip() for l in f if l.strip()]
    elif kind == "csv":
        with open(src, newline="") as f:
            captions = [row[args.csv_caption_key] for row in csv.DictReader(f, delimiter=getattr(args, "csv_separator", "\t"))]
    else:
        shards = _expand_braces(src)
    n = args.train_num_samples or (len(captions) if captions is not None else None)
    if n is None:
        raise ValueError("--train-num-samples is required for webdataset shards (data_AT.py:455-465)")
    # --workers 0 is the reference's "load in the training process" (data_AT.py:487-494): no read-ahead thread then
    loader = TextLoader(captions, shards, args.batch_size, n, rank, world, args.seed,
                        prefetch=4 if getattr(args, "workers", 1) > 0 else 0)
    loader.set_epoch(epoch)
    return {"train": DataInfo(loader)}


class MicroClip:
    """--grad-clip-norm together with --accum-freq > 1 (utils_AT.py:348-357): the reference calls clip_grad_norm_ after EVERY
    micro-batch's backward, i.e. on the running sum of the micro-batch gradients (under DDP: of their means over ranks, every
    backward all-reduces).  With a GradScaler (--precision amp, train_AT_text_only.py:347) the second ``scaler.unscale_`` of an
    optimizer step raises in torch -- that combination never trains in the reference, and ``check`` raises the same error.
    Otherwise, per micro-batch j of an optimizer step, with G the replicated running sum (mean units):
        G <- G / world (j > 0, data parallel only);  G += g_local / (accum * world);  all-reduce(sum);  G <- clip(G)
    so that the optimizer step finds the gradient the reference's AdamW finds; one gradient reduction per MICRO-batch is what
    these semantics cost (the default path reduces once per optimizer step)."""

    def __init__(self, model, args):
        self.model = unwrap_model(model)
        self.max_norm = args.grad_clip_norm
        self.accum = args.accum_freq
        self.active = self.max_norm is not None and self.accum > 1

    @staticmethod
    def check(args, scaler=None):
        if args.grad_clip_norm is not None and args.accum_freq > 1 and (scaler is not None or getattr(args, "precision", "amp") == "amp"):
            raise RuntimeError("unscale_() has already been called on this optimizer since the last update().  (--grad-clip-norm "
                               "with --accum-freq > 1 under --precision amp: the reference un-scales the gradients once per "
                               "micro-batch, utils_AT.py:348-351, which torch.cuda.amp.GradScaler refuses; use amp_bf16)")

    def backward(self, feat, anchor, micro: int):
        from .step import _dp_active
        import torch.distributed as dist
        red = get_reducer(self.model)
        world = dist.get_world_size() if _dp_active() else 1
        if world > 1 and micro > 0:
            self.model.clip_grads_(None, pre_scale=1.0 / world)
        loss = red.backward(feat, anchor, accum_scale=1.0 / (self.accum * world), last_micro=True)
        if world > 1:
            red.finish()
        self.model.clip_grads_(self.max_norm)
        return loss


# ----------------------------------------------------------------------------- the epoch loop
def train_one_epoch_text_only(model, model_frozen, tokenizer, V, data, loss, epoch, optimizer, scaler, scheduler, args,
                              tb_writer=None):
    if getattr(args, "use_charmer", False):
        raise NotImplementedError("--use_charmer (per-sentence Charmer attack) is outside the accelerated path")
    if getattr(args, "horovod", False):
        raise NotImplementedError("horovod is not supported; launch with torch.distributed.run (RCCL)")
    normalize_fare = bool(getattr(args, "normalize_fare", False))   # utils_AT.py:296,319
    MicroClip.check(args, scaler)
    micro_clip = MicroClip(model, args)
    device = torch.device(args.device)
    model.train()
    data['train'].set_epoch(epoch)
    dataloader = data['train'].dataloader
    num_batches_per_epoch = dataloader.num_batches // args.accum_freq
    sample_digits = math.ceil(math.log(dataloader.num_samples + 1, 10))
    losses_m, losses_accum, times = {}, {}, []
    batch_time_m, data_time_m = AverageMeter(), AverageMeter()
    log_data = {}
    end = time.time()
    load_time_m = AverageMeter()       # time this loop WAITED for the loader's next batch (the reference's "Data (t)" also
    batches = iter(dataloader)         # spans the attack and the forward, utils_AT.py:324, and is kept as it is)
    i = -1
    while True:
        t_load = time.time()
        try:
            batch = next(batches)
        except StopIteration:
            break
        load_time_m.update(time.time() - t_load)
        i += 1
        i_accum = i // args.accum_freq
        step = num_batches_per_epoch * epoch + i_accum
        if not args.skip_scheduler:
            scheduler(step)
        _, texts = batch
        model.eval()
        # the anchor forward needs only the captions and the FROZEN weights: it runs on a side stream beside the tail of the
        # previous step (backward / AdamW / re-pack still queued on this stream) and beside the search's clean-caption K/V pass;
        # the search's first scoring launch waits for its event (as leaf_amd/step.py does for the token-id step)
        from .step import _side_stream
        cur_stream, side = torch.cuda.current_stream(device), _side_stream(device)
        if i == 0:
            # once per epoch: whatever the caller queued on this stream before the loop (copy_from / pack of the frozen model's
            # 16-bit weights, a checkpoint load) is finished before the side stream first reads it; later steps need no such
            # ordering (the frozen weights never change), which is what lets the anchor overlap the previous step's tail
            side.wait_stream(cur_stream)
        with torch.cuda.stream(side):
            anchor = model_frozen.encode_text(tokenizer.encode_batch(texts), normalize=normalize_fare,
                                              precise=bool(getattr(args, "precise_anchor", False)))
            anchor_ready = torch.cuda.Event()
            anchor_ready.record(side)
        anchor.record_stream(cur_stream)
        t0 = time.time()
        _, adv_texts = attack_text(model, tokenizer, texts, anchor, device, objective='l2', n=args.rho, k=args.k_adv,
                                   V=V, constrain=args.constrain, debug=False, anchor_ready=anchor_ready)
        times.append(time.time() - t0)
        if i == 0 and os.environ.get("LEAF_DEBUG_DUMP_ADV"):   # test hook: this rank's first adversarial batch of the epoch
            with open(os.path.join(os.environ["LEAF_DEBUG_DUMP_ADV"], f"adv_epoch{epoch}_rank{getattr(args, 'rank', 0)}.txt"), "w") as f:
                f.write("\n".join(f"{a}\t{b}" for a, b in zip(texts, adv_texts)) + "\n")
        adv_tokens = tokenizer.encode_batch(adv_texts)
        model.train()
        if args.accum_freq == 1 and not micro_clip.active:
            # (opt-in sparse reduction of the token-embedding gradient, LEAF_DP_SPARSE_EMBED=1: this rank's token count travels now;
            # a no-op otherwise -- leaf_amd/step.py:GradReducer.begin_step)
            get_reducer(model).begin_step(int(np.count_nonzero(np.asarray(adv_tokens))))
        feat = model.forward_train(adv_tokens, normalize=normalize_fare)
        data_time_m.update(time.time() - end)
        if micro_clip.active:
            loss_fare = micro_clip.backward(feat, anchor, i % args.accum_freq)                # device scalar, no sync
        else:
            loss_fare = get_reducer(model).backward(feat, anchor, accum_scale=1.0 / args.accum_freq,
                                                    last_micro=(i + 1) % args.accum_freq == 0)
        for key in ("loss", "loss_FARE_text"):
            losses_accum[key] = losses_accum.get(key, 0) + loss_fare / args.accum_freq
        if (i + 1) % args.accum_freq == 0:
            # --grad-clip-norm (utils_AT.py:348-357): with one micro-batch per step the clip is fused into the AdamW kernel; with
            # --accum-freq > 1 the running sum was clipped after every micro-batch's backward (MicroClip), the last one included
            if micro_clip.active:
                optimizer.step(max_norm=None, reduced=True)
            else:
                optimizer.step(max_norm=args.grad_clip_norm)
            optimizer.zero_grad()
        with torch.no_grad():
            unwrap_model(model).logit_scale.clamp_(0, math.log(100))
        batch_time_m.update(time.time() - end)
        end = time.time()
        batch_count = i_accum + 1
        if is_master(args) and (i + 1) % args.accum_freq == 0 and (
                batch_count % args.log_every_n_steps == 0 or batch_count == num_batches_per_epoch):
            batch_size = len(texts)
            num_samples = batch_count * batch_size * args.accum_freq * args.world_size
            percent_complete = 100.0 * batch_count / num_batches_per_epoch
            for key, val in losses_accum.items():
                losses_m.setdefault(key, AverageMeter()).update(float(val), batch_size)
            gs = unwrap_model(model).grad_scaler_state()      # float(val) above already synchronised
            skipped = gs["skipped"]
            if skipped != getattr(model, "_skipped_seen", 0):      # per model, not per process
                logging.warning(f"non-finite gradient norm: {skipped} optimizer step(s) skipped so far ({gs['skipped_saturated']} of them "
                                f"because a 16-bit gradient tensor saturated at the current loss scale); weights and AdamW moments "
                                f"left untouched for those steps and the loss scale halved, as torch.cuda.amp.GradScaler does "
                                f"(persistent loss-scale factor now {gs['loss_scale_factor']:g})")
                model._skipped_seen = skipped
            loss_log = " ".join(f"{n.capitalize()}: {m.val:#.5g} ({m.avg:#.5g})" for n, m in losses_m.items())
            sps = args.accum_freq * args.batch_size * args.world_size / batch_time_m.val
            sps_gpu = args.accum_freq * args.batch_size / batch_time_m.val
            logging.info(
                f"Train Epoch: {epoch} [{num_samples:>{sample_digits}}/{dataloader.num_samples} ({percent_complete:.0f}%)] "
                f"Data (t): {data_time_m.avg:.3f} Batch (t): {batch_time_m.avg:.3f}, {sps:#g}/s, {sps_gpu:#g}/s/gpu "
                f"LR: {optimizer.param_groups[0]['lr']:5f} " + loss_log + f" Load (t): {load_time_m.avg:.4f}")
            log_data = {"data_time": data_time_m.val, "batch_time": batch_time_m.val, "load_time": load_time_m.avg, "samples_per_second": sps,
                        "samples_per_second_per_gpu": sps_gpu, "lr": optimizer.param_groups[0]["lr"]}
            log_data.update({name: val.val for name, val in losses_m.items()})
            log_data = {"train/" + name: val for name, val in log_data.items()}
            if tb_writer is not None:
                for name, val in log_data.items():
                    tb_writer.add_scalar(name, val, step)
            batch_time_m.reset()
            data_time_m.reset()
            load_time_m.reset()
        if (i + 1) % args.accum_freq == 0:
            losses_accum = {}
    if is_master(args):  # the reference rewrites this file every step (utils_AT.py:311); same content at epoch end
        with open(f"times_{getattr(args, 'use_charmer', False)}.csv", "w") as f:
            f.write("0\n" + "\n".join(str(t) for t in times) + "\n")
    return log_data


# ----------------------------------------------------------------------------- checkpoints
def natural_key(s):
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s.lower())]


def get_latest_checkpoint(path: str):
    cks = sorted(glob.glob(os.path.join(path, '**', '*.pt'), recursive=True), key=natural_key)
    return cks[-1] if cks else None


def save_checkpoint(path, epoch, name, model, optimizer):
    from .checkpoint import save_training_checkpoint
    save_training_checkpoint(path, epoch, name, model, optimizer.state_dict())


def load_checkpoint(path, model, optimizer=None):
    """train_AT_text_only.py:351-372: a training checkpoint ({'epoch', 'name', 'state_dict', 'optimizer'}, written here or by
    the reference) restores weights, AdamW moments and step; a bare state_dict restores the weights only.  Returns the epoch."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    if "epoch" in ck:
        model.load_state_dict(ck["state_dict"])
        if optimizer is not None:
            if "optimizer" in ck:
                optimizer.load_state_dict(ck["optimizer"])
                logging.info(f"=> optimizer state restored (step {model.opt_step})")
            else:
                logging.warning(f"=> checkpoint '{path}' carries no optimizer state: AdamW moments and step start from zero")
        model.pack()
        return ck["epoch"]
    model.load_state_dict(ck)
    if optimizer is not None:
        logging.warning(f"=> '{path}' is a bare state_dict: AdamW moments and step start from zero")
    model.pack()
    return 0
