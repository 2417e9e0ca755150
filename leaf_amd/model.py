"""Text tower of CLIP on the HIP engine, behind the reference's ``model.encode_text`` seam.

``LeafCLIPText`` owns ONE flat fp32 parameter tensor on the GPU (layout defined by libleaf_hip, decay
group first) plus 16-bit MFMA operand copies, and exposes

* ``encode_text(text, normalize=False)``   -- src/open_clip/model.py:269-284
* ``score_candidates(tokens, anchor, rho, objective)`` -- one stage of utils_attacks.py:297-393
* ``forward_train`` / ``backward`` / ``adamw_step``    -- utils_AT.py:317-362
* ``state_dict`` / ``load_state_dict`` with OpenCLIP keys and HF ``CLIPTextModel`` keys
  (conversion/convert_2.py:37-99 key map).

PyTorch is used for device memory and streams only; every computation is a call into the C ABI.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib


@dataclass(frozen=True)
class TextConfig:
    width: int = 768
    heads: int = 12
    layers: int = 12
    embed_dim: int = 768
    context_length: int = 77
    vocab_size: int = 49408
    quick_gelu: bool = False
    ln_eps: float = 1e-5


# src/open_clip/model_configs/ViT-{L,H,g,bigG}-14.json (text_cfg + embed_dim); the hf-hub ids are the ones the
# reference's launch scripts pass as --model (scripts/train_leaf_vit*.sh)
MODEL_CONFIGS: Dict[str, TextConfig] = {
    "ViT-L-14": TextConfig(768, 12, 12, 768),
    "ViT-L-14-quickgelu": TextConfig(768, 12, 12, 768, quick_gelu=True),
    "ViT-H-14": TextConfig(1024, 16, 24, 1024),
    "ViT-g-14": TextConfig(1024, 16, 24, 1024),
    "ViT-bigG-14": TextConfig(1280, 20, 32, 1280),
    "tiny-test": TextConfig(128, 2, 2, 64),
    "tiny-test-quickgelu": TextConfig(128, 2, 2, 64, quick_gelu=True),
}
_HUB_ALIASES = {
    "hf-hub:chs20/fare2-clip": "ViT-L-14-quickgelu",
    "hf-hub:chs20/tecoa2-clip": "ViT-L-14-quickgelu",
    "hf-hub:laion/CLIP-ViT-H-14-laion2B-s32B-b79K": "ViT-H-14",
    "hf-hub:laion/CLIP-ViT-g-14-laion2B-s12B-b42K": "ViT-g-14",
    "hf-hub:laion/CLIP-ViT-bigG-14-laion2B-39B-b160k": "ViT-bigG-14",
}


def get_config(name: str) -> TextConfig:
    name = _HUB_ALIASES.get(name, name)
    if name not in MODEL_CONFIGS:
        raise KeyError(f"unknown text tower '{name}'; known: {sorted(MODEL_CONFIGS)}")
    return MODEL_CONFIGS[name]


_DTYPES = {"bf16": 0, "fp16": 1}
# arithmetic of the forward-only passes: split masks per leading block (LeafCLIPText.set_split_masks: bit 0 QKV, 1 out_proj weights,
# 2 c_fc, 3 c_proj weights)
PRECISION_MODES = {"rowsafe": (3, 2, 2, 2), "fast": ()}
PRECISION_DEFAULT = "rowsafe"
_OBJ = {"l2": 0, "negl2": 1, "dissim": 2, "sim": 3}


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class LeafCLIPText:
    """Text tower on the HIP engine.  Not an ``nn.Module``: parameters live in ``self.flat`` (fp32, CUDA);
    ``self.params[name]`` are views into it with the OpenCLIP names."""

    def __init__(self, cfg: TextConfig, device="cuda:0", dtype: str = None, chunk: int = None, trainable: bool = False):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LeafHipError("LeafCLIPText needs a CUDA(HIP) device; there is no CPU path")
        dtype = dtype or os.environ.get("LEAF_DTYPE", "fp16")
        self.dtype_name = dtype
        self._lib = _lib.lib()
        c = _lib.TextCfgC(cfg.layers, cfg.width, cfg.heads, cfg.embed_dim, cfg.context_length, cfg.vocab_size,
                          1 if cfg.quick_gelu else 0, cfg.ln_eps)
        h = C.c_void_p()
        _lib.check(self._lib.leaf_text_create(C.byref(c), _DTYPES[dtype], C.byref(h)), "leaf_text_create")
        self._h = h
        chunk = chunk or int(os.environ.get("LEAF_CHUNK", "4096"))   # sequences-worth of rows per pass (x 77 rows)
        _lib.check(self._lib.leaf_text_set_chunk(h, chunk), "leaf_text_set_chunk")
        self.n_params = self._lib.leaf_text_param_count(h)
        self.n_decay = self._lib.leaf_text_decay_count(h)
        with torch.cuda.device(self.device):
            self.flat = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.w16 = torch.empty(self._lib.leaf_text_w16_bytes(h), dtype=torch.uint8, device=self.device)
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.params: Dict[str, torch.Tensor] = {}
        name = C.create_string_buffer(128)
        off, rows, cols = C.c_size_t(), C.c_int64(), C.c_int64()
        for i in range(self._lib.leaf_text_num_tensors(h)):
            _lib.check(self._lib.leaf_text_param_info(h, i, name, 128, C.byref(off), C.byref(rows), C.byref(cols)), "param_info")
            shape = (rows.value, cols.value) if cols.value else (rows.value,)
            self.layout[name.value.decode()] = (off.value, shape)
            self.params[name.value.decode()] = self.flat[off.value: off.value + int(np.prod(shape))].view(shape)
        self.logit_scale = torch.tensor(math.log(1 / 0.07), device=self.device)  # carried for checkpoints only
        # tensors of a loaded checkpoint outside the text tower (visual.*, ...): kept on the host and written back by
        # state_dict(), so that epoch_latest.pt stays a FULL CLIP state_dict the reference's --resume / converters can load
        self.extra_state: Dict[str, torch.Tensor] = {}
        self._ws: Dict[int, torch.Tensor] = {}
        self._packed = False
        # exact work skipping: compute only rows up to EOT (include/leaf_hip.h, "EOT trimming"); LEAF_PACK=0 disables
        self.trim_rows = os.environ.get("LEAF_PACK", "1") != "0"
        self.training = False
        # training state (allocated on demand)
        self.grads = self.exp_avg = self.exp_avg_sq = self.w16_bwd = self._stash = None
        self.opt_step = 0
        self.rows_scored = 0      # packed rows handed to the scoring passes so far (host-side count; bench.py reports it per rank)
        if trainable:
            self.enable_training()
        # default arithmetic of the forward-only passes: LEAF_PRECISION=rowsafe (default) | fast
        self.split_masks = ()
        self._split_buf = None
        mode = os.environ.get("LEAF_PRECISION", PRECISION_DEFAULT)
        if mode not in PRECISION_MODES:
            raise ValueError(f"LEAF_PRECISION={mode}: one of {sorted(PRECISION_MODES)}")
        if PRECISION_MODES[mode] and self._lib.leaf_text_get_option(self._h, b"ln_fold"):
            self.set_precision(mode)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.leaf_text_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        """Engine switches (leaf_text_set_option): 'chunk', 'last_layer_trim', 'streams' (1 | 2: two-stream
        chunk pipeline of the forward-only passes), 'normalize_fare' (training forward / backward on normalised features),
        'compact_resid' (1 default: the residual stream of the forward-only passes as the 16-bit copy + an 8-bit remainder per
        element instead of fp32 rows; 0 = fp32 rows)."""
        if name == "ln_fold" and not value and getattr(self, "split_masks", ()):
            self.set_split_masks(())          # the split GEMMs exist in the LN-folded forward only: switching folding off switches them off
        _lib.check(self._lib.leaf_text_set_option(self._h, name.encode(), int(value)), "leaf_text_set_option")
        return self

    # ------------------------------------------------------------------ nn.Module look-alikes
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def named_parameters(self):
        return list(self.params.items())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, mode: int, n_seq: int) -> torch.Tensor:
        need = self._lib.leaf_text_workspace_bytes(self._h, n_seq, mode)
        ws = self._ws.get(mode)
        if ws is None or ws.numel() < need:
            ws = None
            self._ws[mode] = None
            with torch.cuda.device(self.device):
                ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[mode] = ws
        return ws

    # ------------------------------------------------------------------ weights
    def init_random(self, seed: int = 1):
        """Same generator and distributions as oracle.text_oracle.init_weights (numpy PCG64), restated here so
        the product does not import the oracle: transformer.py:731-752 std's, perturbed LN affine / biases."""
        cfg = self.cfg
        rng = np.random.default_rng(seed)
        d, L, D = cfg.width, cfg.layers, cfg.embed_dim
        n = lambda shape, std: (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)
        sd = {"token_embedding.weight": n((cfg.vocab_size, d), 0.02), "positional_embedding": n((cfg.context_length, d), 0.01)}
        proj_std, attn_std, fc_std = (d ** -0.5) * ((2 * L) ** -0.5), d ** -0.5, (2 * d) ** -0.5
        for i in range(L):
            p = f"transformer.resblocks.{i}."
            sd[p + "ln_1.weight"] = (1.0 + n((d,), 0.05)).astype(np.float32)
            sd[p + "ln_1.bias"] = n((d,), 0.02)
            sd[p + "attn.in_proj_weight"] = n((3 * d, d), attn_std)
            sd[p + "attn.in_proj_bias"] = n((3 * d,), 0.02)
            sd[p + "attn.out_proj.weight"] = n((d, d), proj_std)
            sd[p + "attn.out_proj.bias"] = n((d,), 0.02)
            sd[p + "ln_2.weight"] = (1.0 + n((d,), 0.05)).astype(np.float32)
            sd[p + "ln_2.bias"] = n((d,), 0.02)
            sd[p + "mlp.c_fc.weight"] = n((4 * d, d), fc_std)
            sd[p + "mlp.c_fc.bias"] = n((4 * d,), 0.02)
            sd[p + "mlp.c_proj.weight"] = n((d, 4 * d), proj_std)
            sd[p + "mlp.c_proj.bias"] = n((d,), 0.02)
        sd["ln_final.weight"] = (1.0 + n((d,), 0.05)).astype(np.float32)
        sd["ln_final.bias"] = n((d,), 0.02)
        sd["text_projection"] = n((d, D), d ** -0.5)
        self.load_state_dict(sd)
        return self

    def load_state_dict(self, sd, strict: bool = True):
        """OpenCLIP text-tower keys (a full CLIP state_dict is accepted: visual.* is ignored, 'module.' stripped,
        factory.py:138-139) or HF CLIPTextModel / CLIPModel keys (conversion/convert_to_openclip.py:78-121)."""
        from .checkpoint import non_text_tensors, to_openclip_text_keys
        extra = non_text_tensors(sd)
        if extra:
            self.extra_state = extra
        sd = to_openclip_text_keys(sd, self.cfg)
        missing = [k for k in self.params if k not in sd]
        if missing and strict:
            raise KeyError(f"missing text-tower keys: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        for k, p in self.params.items():
            if k in sd:
                v = sd[k]
                v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
                if tuple(v.shape) != tuple(p.shape):
                    raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(p.shape)}")
                p.copy_(v.to(device=self.device, dtype=torch.float32))
        if "logit_scale" in sd:
            self.logit_scale = torch.as_tensor(sd["logit_scale"], dtype=torch.float32).to(self.device)
        self._packed = False
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """OpenCLIP keys: the text tower (device tensors), ``logit_scale`` and every non-text tensor of the checkpoint this
        model was loaded from (``visual.*`` etc., host tensors) -- a full CLIP ``state_dict`` when it started from one."""
        out = {k: v.detach().clone() for k, v in self.params.items()}
        out["logit_scale"] = self.logit_scale.detach().clone()
        out.update(self.extra_state)
        return out

    def set_split_masks(self, masks):
        """Higher-precision GEMMs in the leading blocks of the forward-only passes (include/leaf_hip.h, leaf_text_split_pack_masks):
        ``masks[l]`` says which GEMMs of block l multiply hi + lo 16-bit splits (more MFMA passes through the unchanged kernels):
        bit 0 = QKV (both operands), 1 = out_proj (weights), 2 = c_fc (both operands), 3 = c_proj (weights).  ``PRECISION_MODES``
        names the lists that matter: 'rowsafe' (the DEFAULT: every one of the census's 12,928 ViT-L search rows within 1e-3 of the
        fp32 reference, profiles/r06_row_error_census*.txt) and 'fast' (no split: rounds 1-5, batch 8.7e-4, 1.1 % of the rows above
        1e-3).  ``pack()`` refreshes the split weight copies after every optimizer step."""
        masks = [int(x) for x in masks]
        while masks and masks[-1] == 0:
            masks.pop()
        if len(masks) > self.cfg.layers - 1 or any(x < 0 or x > 15 for x in masks):
            raise ValueError(f"split masks {masks}: at most {self.cfg.layers - 1} blocks, values 0..15")
        self.split_masks = tuple(masks)
        self._split_buf = None
        if masks:
            with torch.cuda.device(self.device):
                self._split_buf = torch.empty(self._lib.leaf_text_split_bytes(self._h, len(masks)), dtype=torch.uint8, device=self.device)
        self._split_pack()
        return self

    def set_split_blocks(self, blocks: int):
        """All four GEMMs of the first ``blocks`` blocks on splits (rounds 4-5's escape hatch; 0 = none): ``set_split_masks([15] * blocks)``."""
        blocks = int(blocks)
        if blocks < 0 or blocks > self.cfg.layers - 1:
            raise ValueError(f"split blocks {blocks} out of range 0..{self.cfg.layers - 1}")
        return self.set_split_masks([15] * blocks)

    @property
    def split_blocks(self) -> int:
        return len(getattr(self, "split_masks", ()))

    def set_precision(self, mode: str):
        """'rowsafe' (default) | 'fast' (no split GEMMs: the round-5 arithmetic) -- see ``set_split_masks``."""
        if mode not in PRECISION_MODES:
            raise ValueError(f"precision mode '{mode}': one of {sorted(PRECISION_MODES)}")
        return self.set_split_masks(PRECISION_MODES[mode][:self.cfg.layers - 1])

    def precision_name(self) -> str:
        for k, v in PRECISION_MODES.items():
            vv = list(v[:self.cfg.layers - 1])
            while vv and vv[-1] == 0:
                vv.pop()
            if tuple(vv) == tuple(getattr(self, "split_masks", ())):
                return k
        return "masks" + "-".join(str(x) for x in self.split_masks)

    def arithmetic_tag(self):
        """What a K/V cache depends on besides the weights: a cache built under one setting must not be consumed under another
        (the bit-exact prefix reuse would silently break)."""
        return (int(self._lib.leaf_text_get_option(self._h, b"compact_resid")), tuple(getattr(self, "split_masks", ())))

    def _split_pack(self):
        masks = getattr(self, "split_masks", ())
        arr = (C.c_int32 * max(len(masks), 1))(*masks)
        _lib.check(self._lib.leaf_text_split_pack_masks(self._h, _ptr(self.flat), arr, len(masks), _ptr(getattr(self, "_split_buf", None)),
                                                        self._stream()), "leaf_text_split_pack_masks")

    def copy_from(self, other: "LeafCLIPText"):
        """Weights AND arithmetic of ``other`` (split blocks / policy, residual-stream format): a frozen copy must embed like its source."""
        self.flat.copy_(other.flat)
        if tuple(getattr(other, "split_masks", ())) != tuple(getattr(self, "split_masks", ())):
            self.set_split_masks(getattr(other, "split_masks", ()))
        self.set_option("compact_resid", other.arithmetic_tag()[0])
        self.logit_scale = other.logit_scale.clone()
        self.extra_state = dict(other.extra_state)
        self._packed = False
        return self

    def pack(self):
        """fp32 masters -> 16-bit MFMA operand copies (forward dtype; + transposed bf16 when training)."""
        _lib.check(self._lib.leaf_text_pack_weights(self._h, _ptr(self.flat), _ptr(self.w16), _ptr(self.w16_bwd),
                                                    self._stream()), "leaf_text_pack_weights")
        if getattr(self, "split_masks", ()):
            self._split_pack()
        self._packed = True

    # ------------------------------------------------------------------ inference
    def _tokens(self, text) -> torch.Tensor:
        if isinstance(text, np.ndarray):
            text = torch.from_numpy(text)
        t = text.to(device=self.device, dtype=torch.int32, non_blocking=True).contiguous()
        if t.dim() != 2 or t.shape[1] != self.cfg.context_length:
            raise ValueError(f"tokens must be [N,{self.cfg.context_length}], got {tuple(t.shape)}")
        return t

    def _row_plan(self, text, seq_lens):
        """(host lens pointer, device cu tensor, keep-alive) for EOT trimming, or NULLs for the dense layout.
        Lengths come from the caller (host array) or, for host-resident tokens, from argmax(tokens) + 1."""
        if not self.trim_rows:
            return C.c_void_p(0), None, None
        if seq_lens is None:
            if isinstance(text, np.ndarray):
                seq_lens = text.reshape(-1, text.shape[-1]).argmax(-1) + 1
            elif isinstance(text, torch.Tensor) and text.device.type == "cpu":
                seq_lens = (text.reshape(-1, text.shape[-1]).argmax(-1) + 1).numpy()
            else:
                return C.c_void_p(0), None, None       # device-resident tokens, lengths unknown on the host
        lens = np.ascontiguousarray(np.asarray(seq_lens).reshape(-1), dtype=np.int32)
        cu = np.zeros(lens.size + 1, dtype=np.int32)
        np.cumsum(lens, out=cu[1:])
        # pinned + non_blocking: a pageable H2D copy would make the host wait for all queued GPU work
        return C.c_void_p(lens.ctypes.data), torch.from_numpy(cu).pin_memory().to(self.device, non_blocking=True), lens

    def encode_text(self, text, normalize: bool = False, seq_lens=None, precise: Optional[bool] = None) -> torch.Tensor:
        """``precise`` (default: the model's ``precise_encode`` attribute, False): the fp32-grade forward of precise.hip -- fp32 stored
        intermediates, fp32 master weights through three MFMA passes of fp16 hi / lo splits -- for embeddings that leave the engine
        (export, eval_textfare) or the frozen model's anchors; ~1e-6 of the fp32 reference per row instead of ~9e-4, about 8x the
        time per row.  The search's scoring passes always run the 16-bit arithmetic."""
        if len(text) == 0:      # an empty batch encodes to an empty [0, embed_dim] tensor, as the torch module's (model.py:269-284)
            return torch.empty(0, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        if precise is None:
            precise = getattr(self, "precise_encode", False)
        if precise:
            lens_p, cu, keep = self._row_plan(text, seq_lens)
            t = self._tokens(text)
            n = t.shape[0]
            out = torch.empty(n, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
            need = self._lib.leaf_text_precise_workspace_bytes(self._h, n)
            ws = self._ws.get("precise")
            if ws is None or ws.numel() < need:
                self._ws["precise"] = ws = None
                with torch.cuda.device(self.device):
                    self._ws["precise"] = ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            _lib.check(self._lib.leaf_text_forward_precise(self._h, _ptr(self.flat), _ptr(t), lens_p, _ptr(cu), n, _ptr(out),
                                                           int(bool(normalize)), _ptr(ws), ws.numel(), self._stream()),
                       "leaf_text_forward_precise")
            return out
        if not self._packed:
            self.pack()
        lens_p, cu, keep = self._row_plan(text, seq_lens)
        t = self._tokens(text)
        n = t.shape[0]
        out = torch.empty(n, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        ws = self._workspace(0, n)
        _lib.check(self._lib.leaf_text_forward(self._h, _ptr(self.flat), _ptr(self.w16), _ptr(t), lens_p, _ptr(cu), n,
                                               _ptr(out), int(bool(normalize)), _ptr(ws), ws.numel(), self._stream()),
                   "leaf_text_forward")
        return out

    def encode_text_kv(self, text, seq_lens=None, want_features: bool = False, slot: int = 0):
        """Forward of the clean captions that keeps every layer's q|k|v rows for prefix reuse by the search
        (include/leaf_hip.h, "prefix reuse").  Returns an opaque cache (and the features when asked).  ``slot``: which of the
        model's cache buffers to fill -- caches of different slots can be alive at the same time (the string search pipelines
        groups of captions, leaf_amd/attacks.py)."""
        if not self._packed:
            self.pack()
        if seq_lens is None:
            arr = text if isinstance(text, np.ndarray) else text.cpu().numpy()
            seq_lens = arr.reshape(-1, arr.shape[-1]).argmax(-1) + 1
        lens = np.ascontiguousarray(np.asarray(seq_lens).reshape(-1), dtype=np.int32)
        cu = np.zeros(lens.size + 1, dtype=np.int32)
        np.cumsum(lens, out=cu[1:])
        cu_dev = torch.from_numpy(cu).pin_memory().to(self.device, non_blocking=True)
        t = self._tokens(text)
        n = t.shape[0]
        need = int(cu[-1]) * 3 * self.cfg.width * 2 * self.cfg.layers
        if slot == 0:
            if getattr(self, "_kv", None) is None or self._kv.numel() < need:
                self._kv = None
                with torch.cuda.device(self.device):
                    self._kv = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
            kvbuf = self._kv
        else:
            slots = self.__dict__.setdefault("_kv_slots", {})
            if slots.get(slot) is None or slots[slot].numel() < need:
                slots[slot] = None
                with torch.cuda.device(self.device):
                    slots[slot] = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
            kvbuf = slots[slot]
        out = torch.empty(n, self.cfg.embed_dim, dtype=torch.float32, device=self.device) if want_features else None
        ws = self._workspace(0, n)
        _lib.check(self._lib.leaf_text_forward_kv(self._h, _ptr(self.flat), _ptr(self.w16), _ptr(t),
                                                  C.c_void_p(lens.ctypes.data), _ptr(cu_dev), n, _ptr(out), 0,
                                                  _ptr(kvbuf), kvbuf.numel(), _ptr(ws), ws.numel(), self._stream()),
                   "leaf_text_forward_kv")
        cache = {"kv": kvbuf, "base_cu": cu_dev, "base_rows": int(cu[-1]), "lens": lens, "n": n, "arith": self.arithmetic_tag()}
        return (cache, out) if want_features else cache

    def score_candidates(self, tokens, anchor: torch.Tensor, rho: int, objective: str = "l2", want_features=True,
                         want_loss=False, seq_lens=None, prefix_lens=None, kv=None):
        """``prefix_lens`` (host int array [B*rho], number of leading positions equal to the clean caption) together
        with ``kv`` (from ``encode_text_kv`` of the clean captions) enables prefix reuse: only positions from the
        first changed token on are recomputed.  ``seq_lens`` must then be the candidates' full lengths."""
        if prefix_lens is not None and kv is not None and self.trim_rows:
            return self._score_prefix(tokens, anchor, rho, objective, want_features, want_loss, seq_lens, prefix_lens, kv)
        """tokens [B*rho, ctx] (or [B,rho,ctx]); anchor [B,D] fp32 CUDA.  Returns (best_idx int32[B],
        best_feat [B,D] or None) (+ loss [B,rho] when want_loss)."""
        if not self._packed:
            self.pack()
        lens_p, cu, keep = self._row_plan(tokens, seq_lens)
        if isinstance(tokens, np.ndarray):
            tokens = torch.from_numpy(tokens)
        t = self._tokens(tokens.reshape(-1, tokens.shape[-1]))
        B = anchor.shape[0]
        if t.shape[0] != B * rho:
            raise ValueError(f"{t.shape[0]} candidate rows != B*rho = {B}*{rho}")
        anchor = anchor.to(device=self.device, dtype=torch.float32).contiguous()
        idx = torch.empty(B, dtype=torch.int32, device=self.device)
        feat = torch.empty(B, self.cfg.embed_dim, dtype=torch.float32, device=self.device) if want_features else None
        loss = torch.empty(B, rho, dtype=torch.float32, device=self.device) if want_loss else None
        ws = self._workspace(1, B * rho)
        self.rows_scored += int(keep.sum()) if keep is not None else B * rho * self.cfg.context_length
        _lib.check(self._lib.leaf_score_candidates(self._h, _ptr(self.flat), _ptr(self.w16), _ptr(t), lens_p, _ptr(cu),
                                                   _ptr(anchor), B, rho, _OBJ[objective], _ptr(idx), _ptr(feat),
                                                   _ptr(loss), _ptr(ws), ws.numel(), self._stream()),
                   "leaf_score_candidates")
        return (idx, feat, loss) if want_loss else (idx, feat)

    def _score_prefix(self, tokens, anchor, rho, objective, want_features, want_loss, seq_lens, prefix_lens, kv):
        return self.score_candidates_run(
            self.score_candidates_prepare(tokens, anchor, rho, objective, want_features, want_loss, seq_lens, kv), prefix_lens)

    def score_candidates_prepare(self, tokens, anchor, rho, objective="l2", want_features=True, want_loss=False, seq_lens=None, kv=None):
        """Everything of a prefix-reuse ``score_candidates`` call that does NOT depend on the prefix lengths: token / anchor placement,
        output buffers, the length limits and a pinned staging buffer for the row plan.  The search calls this BEFORE it waits for the
        previous stage's winners and ``score_candidates_run(plan, prefix_lens)`` right after them: the device idles at that boundary
        for as long as the host needs between the two (round 4, tools/stage_hole_probe.py: 270 -> 94 us at ViT-L B = 128, 181 -> 123 us at ViT-H
        B = 32)."""
        if not self._packed:
            self.pack()
        if seq_lens is None:
            arr = tokens if isinstance(tokens, np.ndarray) else tokens.cpu().numpy()
            seq_lens = arr.reshape(-1, arr.shape[-1]).argmax(-1) + 1
        full = np.ascontiguousarray(np.asarray(seq_lens).reshape(-1), dtype=np.int32)
        n = full.size
        # prefix <= full - 1 (keep at least the EOT row) and <= the rows the cache really holds
        limit = np.minimum(full - 1, np.repeat(kv["lens"].astype(np.int32), rho))
        host = torch.empty(2 * n + 1, dtype=torch.int32, pin_memory=True)      # [cu (n + 1) | prefix (n)]
        if isinstance(tokens, np.ndarray):
            tokens = torch.from_numpy(tokens)
        t = self._tokens(tokens.reshape(-1, tokens.shape[-1]))
        B = anchor.shape[0]
        if t.shape[0] != B * rho or kv["n"] != B or n != B * rho:
            raise ValueError("candidate rows / kv cache do not match B*rho")
        if kv.get("arith", self.arithmetic_tag()) != self.arithmetic_tag():
            raise ValueError(f"K/V cache built under arithmetic {kv['arith']} (compact_resid, split masks), "
                             f"the model now runs {self.arithmetic_tag()}: rebuild the cache (encode_text_kv)")
        anchor = anchor.to(device=self.device, dtype=torch.float32).contiguous()
        idx = torch.empty(B, dtype=torch.int32, device=self.device)
        feat = torch.empty(B, self.cfg.embed_dim, dtype=torch.float32, device=self.device) if want_features else None
        loss = torch.empty(B, rho, dtype=torch.float32, device=self.device) if want_loss else None
        return {"full": full, "limit": limit, "host": host, "t": t, "anchor": anchor, "idx": idx, "feat": feat, "loss": loss, "B": B,
                "rho": rho, "obj": _OBJ[objective], "kv": kv, "ws": self._workspace(1, B * rho), "max_len": int(full.max()),
                "want_loss": want_loss}

    def score_candidates_run(self, plan, prefix_lens):
        """The prefix-dependent rest of ``score_candidates_prepare``: row plan (suffix rows per candidate, their running sum), one
        host-to-device copy, the launches.  Returns what ``score_candidates`` returns."""
        full, n, kv = plan["full"], plan["full"].size, plan["kv"]
        hv = plan["host"].numpy()
        pfx = hv[n + 1:]
        np.minimum(np.asarray(prefix_lens).reshape(-1), plan["limit"], out=pfx, casting="unsafe")
        suf = full - pfx                                                       # rows to compute per candidate (int32, contiguous)
        hv[0] = 0
        np.cumsum(suf, out=hv[1:n + 1])
        dev = plan["host"].to(self.device, non_blocking=True)
        cu_dev, pfx_dev = dev[:n + 1], dev[n + 1:]
        self.rows_scored += int(hv[n])
        _lib.check(self._lib.leaf_score_candidates_prefix(
            self._h, _ptr(self.flat), _ptr(self.w16), _ptr(plan["t"]), C.c_void_p(suf.ctypes.data), _ptr(cu_dev), _ptr(pfx_dev),
            _ptr(kv["base_cu"]), _ptr(kv["kv"]), kv["base_rows"], plan["max_len"], _ptr(plan["anchor"]), plan["B"], plan["rho"], plan["obj"],
            _ptr(plan["idx"]), _ptr(plan["feat"]), _ptr(plan["loss"]), _ptr(plan["ws"]), plan["ws"].numel(), self._stream()),
            "leaf_score_candidates_prefix")
        return (plan["idx"], plan["feat"], plan["loss"]) if plan["want_loss"] else (plan["idx"], plan["feat"])

    def score_candidates_fused(self, base_tokens, base_lens, tokens, anchor: torch.Tensor, rho: int, seq_lens, prefix_lens,
                               objective: str = "l2", want_features: bool = False, want_loss: bool = False):
        """``encode_text_kv(base_tokens)`` and the first stage's ``score_candidates(..., prefix_lens, kv)`` in ONE pass
        (leaf_score_candidates_prefix_fused: the B clean captions ride in the launches of their B*rho candidates).
        Returns (best_idx, best_feat or None, kv cache for the later stages[, loss [B, rho] when want_loss]), or None when
        the rows do not fit one chunk (the caller then uses the two separate calls).  Bit-identical to the separate calls."""
        if not self._packed:
            self.pack()
        B = anchor.shape[0]
        blens = np.ascontiguousarray(np.asarray(base_lens).reshape(-1), dtype=np.int32)
        full = np.asarray(seq_lens, dtype=np.int64).reshape(-1)
        pfx = np.minimum(np.asarray(prefix_lens, dtype=np.int64).reshape(-1), full - 1)      # keep at least the EOT row
        pfx = np.minimum(pfx, np.repeat(blens.astype(np.int64), rho))                        # rows the captions really have
        lens = np.ascontiguousarray(np.concatenate([blens, full - pfx]), dtype=np.int32)     # rows to compute per sequence
        cu = np.zeros(lens.size + 1, dtype=np.int32)
        np.cumsum(lens, out=cu[1:])
        host = np.concatenate([cu, np.zeros(B, dtype=np.int32), pfx.astype(np.int32)])
        dev = torch.from_numpy(host).pin_memory().to(self.device, non_blocking=True)
        cu_dev, pfx_dev = dev[:lens.size + 1], dev[lens.size + 1:]
        bt = self._tokens(base_tokens)
        ct = self._tokens(tokens.reshape(-1, tokens.shape[-1]))
        if bt.shape[0] != B or ct.shape[0] != B * rho:
            raise ValueError("captions / candidate rows do not match B, B*rho")
        t = torch.cat([bt, ct], 0)
        base_rows = int(cu[B])
        need = base_rows * 3 * self.cfg.width * 2 * self.cfg.layers
        if getattr(self, "_kv", None) is None or self._kv.numel() < need:
            self._kv = None
            with torch.cuda.device(self.device):
                self._kv = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        anchor = anchor.to(device=self.device, dtype=torch.float32).contiguous()
        idx = torch.empty(B, dtype=torch.int32, device=self.device)
        feat = torch.empty(B, self.cfg.embed_dim, dtype=torch.float32, device=self.device) if want_features else None
        loss = torch.empty(B, rho, dtype=torch.float32, device=self.device) if want_loss else None
        ws = self._workspace(1, B + B * rho)
        max_len = int(max(int(full.max()), int(blens.max())))
        rc = self._lib.leaf_score_candidates_prefix_fused(
            self._h, _ptr(self.flat), _ptr(self.w16), _ptr(t), C.c_void_p(lens.ctypes.data), _ptr(cu_dev), _ptr(pfx_dev), max_len,
            _ptr(anchor), B, rho, _OBJ[objective], _ptr(idx), _ptr(feat), _ptr(loss), _ptr(self._kv), self._kv.numel(),
            _ptr(ws), ws.numel(), self._stream())
        if rc == 2:
            return None
        _lib.check(rc, "leaf_score_candidates_prefix_fused")
        self.rows_scored += int(cu[-1])
        cache = {"kv": self._kv, "base_cu": cu_dev[:B + 1], "base_rows": base_rows, "lens": blens, "n": B, "arith": self.arithmetic_tag()}
        return (idx, feat, cache, loss) if want_loss else (idx, feat, cache)

    # ------------------------------------------------------------------ training
    def enable_training(self):
        if self.grads is None:
            with torch.cuda.device(self.device):
                self.grads = torch.zeros_like(self.flat)
                self.exp_avg = torch.zeros_like(self.flat)
                self.exp_avg_sq = torch.zeros_like(self.flat)
                self.w16_bwd = torch.empty_like(self.w16)
                # gradient-scaler state + scratch of the guarded optimizer step (include/leaf_hip.h "gradient scaler"): attached to
                # the handle, so the fp16 backward checks its 16-bit gradient tensors for saturation (LEAF_GRAD_SCALER=0 detaches)
                self._clip_ws = torch.zeros(_lib.SC_WORDS + 2048, dtype=torch.float32, device=self.device)
            self._scaler_attached = os.environ.get("LEAF_GRAD_SCALER", "1") != "0"
            if self._scaler_attached:
                _lib.check(self._lib.leaf_text_set_grad_scaler(self._h, _ptr(self._clip_ws)), "leaf_text_set_grad_scaler")
            self._packed = False
        return self

    def zero_grad(self):
        self.grads.zero_()

    def forward_train(self, text, seq_lens=None, delta: Optional[torch.Tensor] = None, normalize: bool = False) -> torch.Tensor:
        """Training-mode forward that keeps the activation stash.  ``delta`` (optional embedding-space PGD mode, SURVEY 8a
        row a12): fp32 CUDA tensor [rows, width] in the PACKED row layout of this call (``rows_of(seq_lens)`` rows; dense
        [N * ctx, width] when rows are not trimmed), added to the token embeddings.  ``normalize`` = --normalize_fare
        (utils_AT.py:319): returns F.normalize(features); the following ``backward`` differentiates through it."""
        self.enable_training()
        if bool(normalize) != getattr(self, "_normalize_fare", False):
            self.set_option("normalize_fare", int(bool(normalize)))
            self._normalize_fare = bool(normalize)
        if not self._packed:
            self.pack()
        self._train_plan = self._row_plan(text, seq_lens)
        lens_p, cu, keep = self._train_plan
        t = self._tokens(text)
        n = t.shape[0]
        need = self._lib.leaf_text_stash_bytes(self._h, n)
        if self._stash is None or self._stash.numel() < need:
            self._stash = None
            with torch.cuda.device(self.device):
                self._stash = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(n, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        ws = self._workspace(2, n)
        if delta is not None:
            rows = int(keep.sum()) if keep is not None else n * self.cfg.context_length
            if delta.dtype != torch.float32 or not delta.is_contiguous() or delta.numel() != rows * self.cfg.width:
                raise ValueError(f"delta must be contiguous fp32 with {rows} x {self.cfg.width} elements")
        _lib.check(self._lib.leaf_text_forward_train_delta(self._h, _ptr(self.flat), _ptr(self.w16), _ptr(t), lens_p,
                                                           _ptr(cu), n, _ptr(delta), _ptr(out), _ptr(self._stash),
                                                           self._stash.numel(), _ptr(ws), ws.numel(), self._stream()),
                   "leaf_text_forward_train_delta")
        self._train_tokens = t
        return out

    def input_grad(self, feat: torch.Tensor, anchor: torch.Tensor):
        """TextFARE loss of (anchor, feat) and its gradient with respect to the token embeddings (= the gradient of
        ``delta``) through the stash of the last ``forward_train``: (loss 0-d, d_embed fp32 [rows, width] packed).
        No parameter gradient is touched (the inner step of the optional embedding-space PGD mode)."""
        t = self._train_tokens
        n = t.shape[0]
        lens_p, cu, keep = self._train_plan
        rows = int(keep.sum()) if keep is not None else n * self.cfg.context_length
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        g = torch.empty(rows, self.cfg.width, dtype=torch.float32, device=self.device)
        anchor = anchor.to(device=self.device, dtype=torch.float32).contiguous()
        ws = self._workspace(2, n)
        _lib.check(self._lib.leaf_textfare_input_grad(self._h, _ptr(self.flat), _ptr(self.w16_bwd), _ptr(t), lens_p,
                                                      _ptr(cu), n, _ptr(feat.contiguous()), _ptr(anchor),
                                                      _ptr(self._stash), _ptr(g), _ptr(loss), _ptr(ws), ws.numel(),
                                                      self._stream()), "leaf_textfare_input_grad")
        return loss, g

    def pgd_step(self, delta: torch.Tensor, grad: torch.Tensor, alpha: float, eps: float, norm: str = "linf"):
        """In-place fused update of the resident perturbation for the rows of the last ``forward_train``:
        'linf': clamp(delta + alpha sign(grad), -eps, eps); 'l2': per-sequence normalised step + L2-ball projection."""
        lens_p, cu, keep = self._train_plan
        n = self._train_tokens.shape[0]
        _lib.check(self._lib.leaf_pgd_step(_ptr(delta), _ptr(grad), _ptr(cu), n, self.cfg.context_length, self.cfg.width,
                                           float(alpha), float(eps), {"linf": 0, "l2": 2}[norm], self._stream()),
                   "leaf_pgd_step")
        return delta

    def backward(self, feat: torch.Tensor, anchor: torch.Tensor, accum_scale: float = 1.0, layer_events=None) -> torch.Tensor:
        """TextFARE loss of (anchor, feat) + backward through the stash of the last ``forward_train``.
        Accumulates into ``self.grads``; returns the (unscaled) loss as a 0-d CUDA tensor.
        ``layer_events``: optional list of cfg.layers + 1 ``torch.cuda.Event`` (already created, i.e. recorded once): event l
        is recorded when every gradient of block l is final, the last one when everything is (gradient-bucket overlap)."""
        t = self._train_tokens
        n = t.shape[0]
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        anchor = anchor.to(device=self.device, dtype=torch.float32).contiguous()
        ws = self._workspace(2, n)
        lens_p, cu, keep = self._train_plan
        if layer_events is not None:
            if len(layer_events) != self.cfg.layers + 1:
                raise ValueError(f"layer_events must hold {self.cfg.layers + 1} events")
            evs = (C.c_void_p * len(layer_events))(*[C.c_void_p(e.cuda_event) for e in layer_events])
            _lib.check(self._lib.leaf_textfare_backward_events(self._h, _ptr(self.flat), _ptr(self.w16_bwd), _ptr(t), lens_p,
                                                               _ptr(cu), n, _ptr(feat.contiguous()), _ptr(anchor), float(accum_scale),
                                                               _ptr(self._stash), _ptr(self.grads), _ptr(loss), _ptr(ws),
                                                               ws.numel(), self._stream(), evs), "leaf_textfare_backward_events")
            return loss
        _lib.check(self._lib.leaf_textfare_backward(self._h, _ptr(self.flat), _ptr(self.w16_bwd), _ptr(t), lens_p,
                                                    _ptr(cu), n, _ptr(feat.contiguous()), _ptr(anchor), float(accum_scale),
                                                    _ptr(self._stash), _ptr(self.grads), _ptr(loss), _ptr(ws),
                                                    ws.numel(), self._stream()), "leaf_textfare_backward")
        return loss

    def clip_grads_(self, max_norm: Optional[float], pre_scale: float = 1.0) -> torch.Tensor:
        """``torch.nn.utils.clip_grad_norm_`` on the accumulated gradient buffer, in place (utils_AT.py:348-357 with
        --accum-freq > 1: the running sum is clipped after every micro-batch's backward).  ``pre_scale`` is applied in the same
        pass; ``max_norm=None`` only applies it.  Returns the 0-d norm tensor (of pre_scale * grads, before clipping)."""
        if getattr(self, "_clipi_ws", None) is None:
            self._clipi_ws = torch.zeros(2 + 2048, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.leaf_clip_grads_inplace(_ptr(self.grads), self.n_params, float(pre_scale),
                                                     float(max_norm) if max_norm is not None else float("inf"),
                                                     _ptr(self._clipi_ws), self._stream()), "leaf_clip_grads_inplace")
        return self._clipi_ws[1]

    def adamw_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                   grad_scale: float = 1.0, max_norm: Optional[float] = None, guard: bool = True):
        """Fused AdamW over the flat buffers.  ``max_norm`` (--grad-clip-norm, utils_AT.py:348-357): clip the global L2
        norm of grad_scale * grads first (torch.nn.utils.clip_grad_norm_ semantics).  ``guard`` (default): one extra read of
        the gradients computes their global norm and SKIPS the whole step when it is inf / NaN (what GradScaler.step does in
        the reference's fp16 regime).  The 16-bit conversions of the gradient path saturate instead of producing inf; the backward
        therefore checks its 16-bit gradient tensors and poisons the gradient with a NaN when one saturated, so this guard skips
        that step too -- on every rank, the NaN travels through the all-reduce -- and the persistent loss-scale back-off halves
        (GradScaler.update; ``loss_scale_backoff()``).  ``opt_step`` counts ATTEMPTED steps; the bias corrections use the steps
        really applied (kept on the device: a skipped step does not advance them, as torch's per-parameter ``step``).
        Returns the 0-d total-norm tensor when clipping or guarding, else None; ``skipped_steps()`` counts skipped steps."""
        self.opt_step += 1
        # with the gradient scaler attached a saturated backward has poisoned the gradient with a NaN that only the guard stops
        guard = guard or getattr(self, "_scaler_attached", False)
        if max_norm is not None or guard:
            if getattr(self, "_clip_ws", None) is None:
                self._clip_ws = torch.zeros(_lib.SC_WORDS + 2048, dtype=torch.float32, device=self.device)
            _lib.check(self._lib.leaf_adamw_step_clip(_ptr(self.flat), _ptr(self.grads), _ptr(self.exp_avg),
                                                      _ptr(self.exp_avg_sq), self.n_params, self.n_decay, float(lr),
                                                      float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                                      self.opt_step, float(grad_scale),
                                                      float(max_norm) if max_norm is not None else float("inf"),
                                                      _ptr(self._clip_ws), self._stream()), "leaf_adamw_step_clip")
            self._packed = False
            return self._clip_ws[_lib.SC_NORM]
        _lib.check(self._lib.leaf_adamw_step(_ptr(self.flat), _ptr(self.grads), _ptr(self.exp_avg),
                                             _ptr(self.exp_avg_sq), self.n_params, self.n_decay, float(lr),
                                             float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                             self.opt_step, float(grad_scale), self._stream()), "leaf_adamw_step")
        self._packed = False
        return None

    def skipped_steps(self) -> int:
        """Optimizer steps skipped by the non-finite guard so far (synchronises: call it where the loop already does)."""
        ws = getattr(self, "_clip_ws", None)
        return int(ws[_lib.SC_SKIPPED].item()) if ws is not None else 0

    def applied_steps(self) -> int:
        """AdamW steps really taken = attempted - skipped: the ``step`` of the bias corrections and of a checkpoint (synchronises)."""
        return self.opt_step - self.skipped_steps()

    def set_applied_steps(self, step: int):
        """After loading optimizer state: ``step`` applied steps so far, none skipped."""
        self.opt_step = int(step)
        ws = getattr(self, "_clip_ws", None)
        if ws is not None:
            ws[_lib.SC_SKIPPED] = 0.0
            ws[_lib.SC_APPLIED] = float(step)

    def grad_scaler_state(self) -> dict:
        """GradScaler-style bookkeeping of the fp16 gradient path (synchronises): the persistent loss-scale factor 2^backoff,
        steps skipped in all / because a 16-bit gradient tensor saturated."""
        ws = getattr(self, "_clip_ws", None)
        if ws is None:
            return {"loss_scale_factor": 1.0, "skipped": 0, "skipped_saturated": 0}
        h = ws[:_lib.SC_WORDS].cpu()
        return {"loss_scale_factor": float(2.0 ** float(h[_lib.SC_BACKOFF])), "skipped": int(h[_lib.SC_SKIPPED]),
                "skipped_saturated": int(h[_lib.SC_SAT_STEPS]), "good_steps": int(h[_lib.SC_GOOD])}

def create_model(name: str, device="cuda:0", dtype: str = None, pretrained: Optional[str] = None,
                 trainable: bool = False, seed: int = 1) -> LeafCLIPText:
    """open_clip.create_model equivalent for the text tower: random init (seeded) or a local checkpoint
    (OpenCLIP .bin/.pt or HF safetensors / directory).  ``hf-hub:`` ids map to their architecture; weights must be
    given as a local path (there is no network on the build or GPU boxes).  An ``open_clip_config.json`` next to the
    checkpoint (the hub layout, src/open_clip/factory.py:200-207) defines the architecture when present."""
    cfg = None
    if pretrained:
        import dataclasses
        import logging
        from .checkpoint import read_open_clip_config
        cfg = read_open_clip_config(pretrained)
        try:
            named = get_config(name)
        except (KeyError, ValueError):
            named = None              # an unknown name is fine when the json defines the architecture
        if cfg is not None and named is not None:
            # the name (or --force-quick-gelu's '-quickgelu' suffix) can only ADD QuickGELU: hub configs of the OpenAI-lineage
            # towers omit the flag and rely on the model name (src/open_clip/factory.py:219-222)
            if named.quick_gelu and not cfg.quick_gelu:
                cfg = dataclasses.replace(cfg, quick_gelu=True)
            shape = lambda c: (c.width, c.heads, c.layers, c.embed_dim, c.context_length, c.vocab_size)
            if shape(cfg) != shape(named):
                if not name.startswith("hf-hub:"):
                    raise ValueError(f"open_clip_config.json next to '{pretrained}' describes a text tower (width, heads, layers, embed_dim, "
                                     f"ctx, vocab) = {shape(cfg)}, but --model {name} is {shape(named)}: remove the stray json or name the "
                                     "model the checkpoint belongs to")
                logging.info(f"architecture taken from the open_clip_config.json next to '{pretrained}': {shape(cfg)}")
            else:
                logging.info(f"open_clip_config.json next to '{pretrained}' agrees with --model {name}"
                             + (" (QuickGELU from the model name)" if named.quick_gelu else ""))
    m = LeafCLIPText(cfg or get_config(name), device=device, dtype=dtype, trainable=trainable)
    if pretrained:
        from .checkpoint import load_checkpoint_file
        m.load_state_dict(load_checkpoint_file(pretrained))
    else:
        m.init_random(seed)
    return m
