"""CPU oracle for the LEAF text hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain numpy fp32 restatement of what the reference computes on the
text path.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``leaf_amd/`` does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference itself in
the build container (stubs for absent third-party packages) and
stores its outputs as fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function below against them.

Reference lines each function follows (paths relative to the reference checkout):

* ``encode_text``            src/open_clip/model.py:269-284 (CLIP.encode_text)
* ``_block``                 src/open_clip/transformer.py:254-265 (ResidualAttentionBlock.forward)
* ``_mha``                   src/open_clip/transformer.py:225,239-252 (nn.MultiheadAttention,
                             packed in_proj, additive causal mask :758-764, scale 1/sqrt(head_dim))
* ``layer_norm``             src/open_clip/transformer.py:15-30 (eps 1e-5, biased variance)
* ``quick_gelu`` / ``gelu``  src/open_clip/transformer.py:33-36 ; nn.GELU (erf form)
* EOT pooling                src/open_clip/transformer.py:653-665 (argmax of the token ids)
* ``score_candidates``       utils_attacks.py:330-348 and :368-386 (objective 'l2' and friends)
* ``textfare_loss``          utils_AT.py:321-322
* ``encode_text_backward``   what ``total_loss.backward()`` (utils_AT.py:79-83,337) produces
* ``adamw_step``             torch.optim.AdamW as configured at train_AT_text_only.py:326-341
* ``cosine_lr``              src/open_clip_train/scheduler.py:4-10,43-53
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32


@dataclass
class TextCfg:
    """Text-tower shape (src/open_clip/model.py:57-83 CLIPTextCfg + embed_dim)."""
    width: int = 768
    heads: int = 12
    layers: int = 12
    embed_dim: int = 768
    context_length: int = 77
    vocab_size: int = 49408
    quick_gelu: bool = False
    eps: float = 1e-5

    @property
    def head_dim(self) -> int:
        return self.width // self.heads


CONFIGS = {
    # src/open_clip/model_configs/ViT-L-14.json etc. (text_cfg + embed_dim)
    "ViT-L-14": TextCfg(768, 12, 12, 768),
    "ViT-L-14-quickgelu": TextCfg(768, 12, 12, 768, quick_gelu=True),
    "ViT-H-14": TextCfg(1024, 16, 24, 1024),
    "ViT-g-14": TextCfg(1024, 16, 24, 1024),
    "ViT-bigG-14": TextCfg(1280, 20, 32, 1280),
}


# --------------------------------------------------------------------------- init
def init_weights(cfg: TextCfg, seed: int = 1) -> Dict[str, np.ndarray]:
    """Random text-tower weights with the distributions of
    src/open_clip/transformer.py:731-752 (init_parameters).  Uses numpy's PCG64 so the
    same seed gives the same weights on every machine.  Biases / LN affine get small
    random values too (the reference leaves them 0 / 1) so that parity tests exercise them."""
    rng = np.random.default_rng(seed)
    d, L, D = cfg.width, cfg.layers, cfg.embed_dim

    def n(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * F32(std)).astype(F32)

    w: Dict[str, np.ndarray] = {}
    w["token_embedding.weight"] = n((cfg.vocab_size, d), 0.02)
    w["positional_embedding"] = n((cfg.context_length, d), 0.01)
    proj_std = (d ** -0.5) * ((2 * L) ** -0.5)
    attn_std = d ** -0.5
    fc_std = (2 * d) ** -0.5
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        w[p + "ln_1.weight"] = (1.0 + n((d,), 0.05)).astype(F32)
        w[p + "ln_1.bias"] = n((d,), 0.02)
        w[p + "attn.in_proj_weight"] = n((3 * d, d), attn_std)
        w[p + "attn.in_proj_bias"] = n((3 * d,), 0.02)
        w[p + "attn.out_proj.weight"] = n((d, d), proj_std)
        w[p + "attn.out_proj.bias"] = n((d,), 0.02)
        w[p + "ln_2.weight"] = (1.0 + n((d,), 0.05)).astype(F32)
        w[p + "ln_2.bias"] = n((d,), 0.02)
        w[p + "mlp.c_fc.weight"] = n((4 * d, d), fc_std)
        w[p + "mlp.c_fc.bias"] = n((4 * d,), 0.02)
        w[p + "mlp.c_proj.weight"] = n((d, 4 * d), proj_std)
        w[p + "mlp.c_proj.bias"] = n((d,), 0.02)
    w["ln_final.weight"] = (1.0 + n((d,), 0.05)).astype(F32)
    w["ln_final.bias"] = n((d,), 0.02)
    w["text_projection"] = n((d, D), d ** -0.5)
    return w


def synthetic_tokens(batch: int, seed: int = 0, ctx: int = 77, min_len: int = 8, max_len: int = 40,
                     vocab: int = 49408) -> np.ndarray:
    """Base token ids of SURVEY.md section 8d: SOT=vocab-2, len~U{min..max} ids in
    1..vocab-3, EOT=vocab-1 (row maximum, so argmax pooling finds it), zero padding."""
    rng = np.random.default_rng(seed)
    out = np.zeros((batch, ctx), dtype=np.int64)
    sot, eot = vocab - 2, vocab - 1
    for i in range(batch):
        n = int(rng.integers(min_len, max_len + 1))
        out[i, 0] = sot
        out[i, 1:1 + n] = rng.integers(1, vocab - 2, size=n)
        out[i, 1 + n] = eot
    return out


def synthetic_candidates(base: np.ndarray, rho: int, seed: int, vocab: int = 49408,
                         fixed_pos: Optional[np.ndarray] = None) -> np.ndarray:
    """[B,rho,ctx] candidates: copy of the base row with ONE position in 1..len resampled
    (SURVEY.md section 8d).  ``fixed_pos`` (int[B]) pins the position (stage 2)."""
    rng = np.random.default_rng(seed)
    B, ctx = base.shape
    eot = base.argmax(-1)  # position of EOT
    cand = np.repeat(base[:, None, :], rho, axis=1).copy()
    for i in range(B):
        n = int(eot[i]) - 1
        if fixed_pos is None:
            pos = rng.integers(1, n + 1, size=rho)
        else:
            pos = np.full(rho, int(fixed_pos[i]))
        cand[i, np.arange(rho), pos] = rng.integers(1, vocab - 2, size=rho)
    return cand


# --------------------------------------------------------------------------- ops
def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=F32)
    rstd = (1.0 / np.sqrt(var + F32(eps))).astype(F32)
    return (xc * rstd * g + b).astype(F32), mu, rstd


def quick_gelu(x):
    return (x / (1.0 + np.exp(-1.702 * x))).astype(F32)


def gelu(x):
    return (0.5 * x * (1.0 + _erf(x * F32(1.0 / math.sqrt(2.0))))).astype(F32)


def _act(cfg):
    return quick_gelu if cfg.quick_gelu else gelu


def _act_grad(cfg, x):
    if cfg.quick_gelu:
        s = 1.0 / (1.0 + np.exp(-1.702 * x))
        return (s * (1.0 + 1.702 * x * (1.0 - s))).astype(F32)
    cdf = 0.5 * (1.0 + _erf(x * F32(1.0 / math.sqrt(2.0))))
    pdf = np.exp(-0.5 * x * x) * F32(1.0 / math.sqrt(2.0 * math.pi))
    return (cdf + x * pdf).astype(F32)


Round = Optional[Callable[[np.ndarray], np.ndarray]]


class RoundPolicy:
    """Operand rounding that depends on WHERE a value is used: ``default`` everywhere except the (layer, site, what) triples
    ``exact`` answers True for -- site in {"qkv", "attn", "out", "fc", "proj", "final"}, what in {"a" (activation operand),
    "w" (weight operand), "out" (a stored 16-bit result: q|k|v, probabilities, attention output, hidden activations)}.
    Used by tests/test_precision_budget.py to attribute the 16-bit forward's error to layers and sites and to price targeted
    spends (a site kept in fp32 / split into hi + lo halves behaves as exact here)."""

    def __init__(self, default, exact=lambda layer, site, what: False):
        self.default, self.exact = default, exact

    def pick(self, layer, site, what):
        return None if self.exact(layer, site, what) else self.default


def _r(rnd, layer, site, what):
    """the rounding function in force at (layer, site, what): a plain callable applies everywhere"""
    if rnd is None:
        return None
    return rnd.pick(layer, site, what) if isinstance(rnd, RoundPolicy) else rnd


def _mm(a, b_t, rnd: Round, layer=-1, site=""):
    """a[M,K] @ b_t[N,K]^T with optional operand rounding (emulates 16-bit MFMA inputs,
    fp32 accumulate).  rnd=None is the fp32 oracle."""
    ra, rw = _r(rnd, layer, site, "a"), _r(rnd, layer, site, "w")
    if ra is not None:
        a = ra(a)
    if rw is not None:
        b_t = rw(b_t)
    return (a @ b_t.T).astype(F32)


def _mha(cfg, xn, wqkv, bqkv, wo, bo, rnd: Round, stash=None, layer=-1):
    N, L, d = xn.shape
    H, hd = cfg.heads, cfg.head_dim
    qkv = _mm(xn.reshape(N * L, d), wqkv, rnd, layer, "qkv") + bqkv
    if _r(rnd, layer, "qkv", "out") is not None:
        qkv = _r(rnd, layer, "qkv", "out")(qkv)
    qkv = qkv.reshape(N, L, 3, H, hd)
    q = qkv[:, :, 0].transpose(0, 2, 1, 3)  # [N,H,L,hd]
    k = qkv[:, :, 1].transpose(0, 2, 1, 3)
    v = qkv[:, :, 2].transpose(0, 2, 1, 3)
    s = np.matmul(q, k.transpose(0, 1, 3, 2)) * F32(1.0 / math.sqrt(hd))
    mask = np.triu(np.full((L, L), -np.inf, dtype=F32), 1)
    s = s + mask
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = (p / p.sum(-1, keepdims=True)).astype(F32)
    rp = _r(rnd, layer, "attn", "out")
    pv = rp(p) if rp is not None else p
    o = np.matmul(pv, v).transpose(0, 2, 1, 3).reshape(N * L, d).astype(F32)
    if rp is not None:
        o = rp(o)
    if stash is not None:
        stash["q"], stash["k"], stash["v"], stash["p"], stash["o"] = q, k, v, p, o
    return (_mm(o, wo, rnd, layer, "out") + bo).reshape(N, L, d)


def encode_text(w: Dict[str, np.ndarray], cfg: TextCfg, tokens: np.ndarray, normalize: bool = False,
                rnd: Round = None, stash: Optional[list] = None, delta: Optional[np.ndarray] = None,
                resid: Round = None) -> np.ndarray:
    """[N,ctx] int ids -> [N,embed_dim] fp32 (src/open_clip/model.py:269-284).  ``delta`` [N,ctx,width]: additive
    perturbation of the token embeddings (the embedding-input forward of src/pez/open_clip_pez/model.py:210-228),
    SURVEY.md 8a row a12.  ``resid``: emulation only -- a storage rounding applied to the residual stream every time it is
    written (the engine's 16 + 8-bit stream: ``lambda x: resid_unpack(*resid_pack(x))``); None = the reference's fp32 stream."""
    tokens = np.asarray(tokens)
    N, L = tokens.shape
    d = cfg.width
    act = _act(cfg)
    x = w["token_embedding.weight"][tokens]
    if delta is not None:
        x = x + np.asarray(delta, dtype=F32)
    x = (x + w["positional_embedding"][:L]).astype(F32)
    if resid is not None:
        x = resid(x)
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        st = {} if stash is not None else None
        xn, mu1, rs1 = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.eps)
        a = _mha(cfg, xn, w[p + "attn.in_proj_weight"], w[p + "attn.in_proj_bias"],
                 w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"], rnd, st, layer=i)
        x1 = (x + a).astype(F32)
        if resid is not None:
            x1 = resid(x1)
        xn2, mu2, rs2 = layer_norm(x1, w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.eps)
        pre = _mm(xn2.reshape(N * L, d), w[p + "mlp.c_fc.weight"], rnd, i, "fc") + w[p + "mlp.c_fc.bias"]
        h = act(pre)
        if _r(rnd, i, "fc", "out") is not None:
            h = _r(rnd, i, "fc", "out")(h)
        m = _mm(h, w[p + "mlp.c_proj.weight"], rnd, i, "proj") + w[p + "mlp.c_proj.bias"]
        x2 = (x1 + m.reshape(N, L, d)).astype(F32)
        if resid is not None:
            x2 = resid(x2)
        if stash is not None:
            st.update(x0=x, xn1=xn, mu1=mu1, rs1=rs1, x1=x1, xn2=xn2, mu2=mu2, rs2=rs2, pre=pre, h=h)
            stash.append(st)
        x = x2
    xf, muf, rsf = layer_norm(x, w["ln_final.weight"], w["ln_final.bias"], cfg.eps)
    eot = tokens.argmax(-1)
    pooled = xf[np.arange(N), eot]
    out = _mm(pooled, np.ascontiguousarray(w["text_projection"].T), rnd, cfg.layers, "final")
    if stash is not None:
        stash.append(dict(x=x, muf=muf, rsf=rsf, eot=eot, pooled=pooled, out=out))
    if normalize:
        out = out / np.linalg.norm(out, axis=-1, keepdims=True)
    return out.astype(F32)


# --------------------------------------------------------------------------- search
def candidate_loss(feat: np.ndarray, anchor: np.ndarray, rho: int, objective: str = "l2") -> np.ndarray:
    """loss[B,rho] of utils_attacks.py:332-346.  feat [B*rho,D], anchor [B,D]."""
    B = anchor.shape[0]
    f = feat.reshape(B, rho, -1)
    a = anchor.reshape(B, 1, -1)
    if objective == "l2":
        return ((f - a) ** 2).sum(-1).astype(F32)
    if objective == "negl2":
        return (-((f - a) ** 2).sum(-1)).astype(F32)
    if objective == "dissim":
        return (-(f * a).sum(-1)).astype(F32)
    if objective == "sim":
        return ((f * a).sum(-1)).astype(F32)
    raise ValueError(objective)


def score_candidates(w, cfg, cand_tokens: np.ndarray, anchor: np.ndarray, objective: str = "l2",
                     rnd: Round = None, chunk: int = 512):
    """cand_tokens [B,rho,ctx] -> (best_idx int64[B], best_feat [B,D], loss [B,rho]).
    torch.argmax semantics: first maximum wins (utils_attacks.py:348,386)."""
    B, rho, L = cand_tokens.shape
    flat = cand_tokens.reshape(B * rho, L)
    norm = objective in ("sim", "dissim")
    feats = np.concatenate([encode_text(w, cfg, flat[i:i + chunk], normalize=norm, rnd=rnd)
                            for i in range(0, B * rho, chunk)], 0)
    loss = candidate_loss(feats, anchor, rho, objective)
    idx = loss.argmax(-1)
    best = feats.reshape(B, rho, -1)[np.arange(B), idx]
    return idx.astype(np.int64), best.astype(F32), loss


# --------------------------------------------------------------------------- training
def textfare_loss(anchor: np.ndarray, feat: np.ndarray) -> float:
    """F.mse_loss(anchor, feat, reduction='none').sum(-1).mean()  (utils_AT.py:321-322)."""
    return float(((anchor.astype(F32) - feat.astype(F32)) ** 2).sum(-1).mean(dtype=np.float64))


def _ln_bwd(dy, x, mu, rstd, g):
    d = x.shape[-1]
    xhat = (x - mu) * rstd
    dg = (dy * xhat).reshape(-1, d).sum(0)
    db = dy.reshape(-1, d).sum(0)
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xhat * (dxh * xhat).mean(-1, keepdims=True))
    return dx.astype(F32), dg.astype(F32), db.astype(F32)


def encode_text_backward(w, cfg: TextCfg, tokens: np.ndarray, anchor: np.ndarray, accum_scale: float = 1.0,
                         delta: Optional[np.ndarray] = None, normalize: bool = False):
    """Forward + TextFARE loss + full backward in fp32.  Returns (loss, feat, grads) where
    grads has the same keys/shapes as ``w`` (utils_AT.py:317-337; loss / accum_freq is
    what gets back-propagated, ``accum_scale`` = 1/accum_freq); when ``delta`` is given also ``grads["d_embed"]``
    [N,ctx,width], the gradient with respect to the perturbed token embeddings = the gradient of ``delta``.
    ``normalize`` = --normalize_fare (utils_AT.py:296,319): the loss is taken on F.normalize(feat) (the anchor is expected
    normalised by the caller); the returned ``feat`` is then the normalised one."""
    tokens = np.asarray(tokens)
    N, L = tokens.shape
    d, H, hd = cfg.width, cfg.heads, cfg.head_dim
    stash: list = []
    feat = encode_text(w, cfg, tokens, stash=stash, delta=delta)
    if normalize:       # F.normalize(dim=-1, eps=1e-12) and its backward: d f = (d n - n (n . d n)) / ||f||
        nrm = np.maximum(np.linalg.norm(feat.astype(np.float64), axis=-1, keepdims=True), 1e-12)
        feat_n = (feat / nrm).astype(F32)
        loss = textfare_loss(anchor, feat_n)
        dn = 2.0 * (feat_n - anchor) / N * accum_scale
        dout = ((dn - feat_n * (feat_n * dn).sum(-1, keepdims=True)) / nrm).astype(F32)
        feat = feat_n
    else:
        loss = textfare_loss(anchor, feat)
        dout = (2.0 * (feat - anchor) / N * accum_scale).astype(F32)          # [N,D]
    g = {k: np.zeros_like(v) for k, v in w.items()}
    top = stash[-1]
    g["text_projection"] = (top["pooled"].T @ dout).astype(F32)
    dpooled = dout @ w["text_projection"].T                                # [N,d]
    dxf = np.zeros((N, L, d), dtype=F32)
    dxf[np.arange(N), top["eot"]] = dpooled
    dx, g["ln_final.weight"], g["ln_final.bias"] = _ln_bwd(dxf, top["x"], top["muf"], top["rsf"], w["ln_final.weight"])
    for i in reversed(range(cfg.layers)):
        p = f"transformer.resblocks.{i}."
        st = stash[i]
        dx2 = dx.reshape(N * L, d)
        # mlp
        g[p + "mlp.c_proj.weight"] = (dx2.T @ st["h"]).astype(F32)
        g[p + "mlp.c_proj.bias"] = dx2.sum(0)
        dh = dx2 @ w[p + "mlp.c_proj.weight"]
        dpre = (dh * _act_grad(cfg, st["pre"])).astype(F32)
        g[p + "mlp.c_fc.weight"] = (dpre.T @ st["xn2"].reshape(N * L, d)).astype(F32)
        g[p + "mlp.c_fc.bias"] = dpre.sum(0)
        dxn2 = (dpre @ w[p + "mlp.c_fc.weight"]).reshape(N, L, d)
        dl, g[p + "ln_2.weight"], g[p + "ln_2.bias"] = _ln_bwd(dxn2, st["x1"], st["mu2"], st["rs2"], w[p + "ln_2.weight"])
        dx1 = (dx + dl).astype(F32)
        # attention
        dx1f = dx1.reshape(N * L, d)
        g[p + "attn.out_proj.weight"] = (dx1f.T @ st["o"]).astype(F32)
        g[p + "attn.out_proj.bias"] = dx1f.sum(0)
        do = (dx1f @ w[p + "attn.out_proj.weight"]).reshape(N, L, H, hd).transpose(0, 2, 1, 3)
        q, k, v, pr = st["q"], st["k"], st["v"], st["p"]
        dv = np.matmul(pr.transpose(0, 1, 3, 2), do)
        dp = np.matmul(do, v.transpose(0, 1, 3, 2))
        ds = pr * (dp - (dp * pr).sum(-1, keepdims=True))
        ds = ds * F32(1.0 / math.sqrt(hd))
        dq = np.matmul(ds, k)
        dk = np.matmul(ds.transpose(0, 1, 3, 2), q)
        dqkv = np.stack([dq, dk, dv], 0).transpose(1, 3, 0, 2, 4).reshape(N * L, 3 * d).astype(F32)
        g[p + "attn.in_proj_weight"] = (dqkv.T @ st["xn1"].reshape(N * L, d)).astype(F32)
        g[p + "attn.in_proj_bias"] = dqkv.sum(0)
        dxn1 = (dqkv @ w[p + "attn.in_proj_weight"]).reshape(N, L, d)
        dl, g[p + "ln_1.weight"], g[p + "ln_1.bias"] = _ln_bwd(dxn1, st["x0"], st["mu1"], st["rs1"], w[p + "ln_1.weight"])
        dx = (dx1 + dl).astype(F32)
    g["positional_embedding"][:L] = dx.sum(0)
    np.add.at(g["token_embedding.weight"], tokens.reshape(-1), dx.reshape(N * L, d))
    if delta is not None:      # only in the embedding-space mode, so that g keeps exactly the keys of w otherwise
        g["d_embed"] = dx.astype(F32)
    return loss, feat, g


# --------------------------------------------------------------------------- optional embedding-space PGD (a12)
def pgd_normalize_grad(grad: np.ndarray, norm: str) -> np.ndarray:
    """src/robust_vlm/train/utils.py:107-114: 'linf' -> sign; 'l2' -> per-sample F.normalize of the flattened gradient
    (eps 1e-12)."""
    if norm == "linf":
        return np.sign(grad).astype(F32)
    flat = grad.reshape(grad.shape[0], -1)
    n = np.maximum(np.sqrt((flat.astype(np.float64) ** 2).sum(1)), 1e-12)
    return (flat / n[:, None]).reshape(grad.shape).astype(F32)


def pgd_project(delta: np.ndarray, eps: float, norm: str) -> np.ndarray:
    """src/robust_vlm/train/utils.py:96-104: 'linf' -> clamp(-eps, eps); 'l2' -> torch.renorm(p=2, dim=0, maxnorm=eps):
    a sample whose L2 norm exceeds eps is scaled by eps / (norm + 1e-7)."""
    if norm == "linf":
        return np.clip(delta, -eps, eps).astype(F32)
    flat = delta.reshape(delta.shape[0], -1)
    n = np.sqrt((flat.astype(np.float64) ** 2).sum(1))
    f = np.where(n > eps, eps / (n + 1e-7), 1.0)
    return (flat * f[:, None]).reshape(delta.shape).astype(F32)


def pgd_step(delta: np.ndarray, grad: np.ndarray, alpha: float, eps: float, norm: str) -> np.ndarray:
    """delta <- project(delta + alpha * normalize_grad(grad)): the update of the continuous attack loop
    (utils_attacks.py:680-697 in its 'linf' form) generalised by the two functions above."""
    return pgd_project((delta + F32(alpha) * pgd_normalize_grad(grad, norm)).astype(F32), eps, norm)


def exclude_from_decay(name: str, p: np.ndarray) -> bool:
    """train_AT_text_only.py:323-324: p.ndim < 2 or 'bn'/'ln'/'bias'/'logit_scale' in name."""
    return p.ndim < 2 or "bn" in name or "ln" in name or "bias" in name or "logit_scale" in name


def adamw_step(w, g, m, v, step: int, lr: float, wd: float, beta1: float = 0.9, beta2: float = 0.999,
               eps: float = 1e-8):
    """One torch.optim.AdamW step (decoupled decay) in place; ``step`` counts from 1.
    Two groups as train_AT_text_only.py:326-341: excluded tensors get weight_decay 0."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k in w:
        decay = 0.0 if exclude_from_decay(k, w[k]) else wd
        w[k] *= F32(1.0 - lr * decay)
        m[k] = (beta1 * m[k] + (1.0 - beta1) * g[k]).astype(F32)
        v[k] = (beta2 * v[k] + (1.0 - beta2) * g[k] * g[k]).astype(F32)
        denom = np.sqrt(v[k]) / F32(math.sqrt(bc2)) + F32(eps)
        w[k] -= (F32(lr / bc1) * m[k] / denom).astype(F32)


def clip_grad_norm(g: Dict[str, np.ndarray], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm, norm_type=2.0) in place (utils_AT.py:348-357): total norm over
    all tensors, coefficient max_norm / (total + 1e-6) clamped to 1.  Returns the total norm."""
    total = math.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g.values()))
    coef = min(1.0, max_norm / (total + 1e-6))
    for k in g:
        g[k] = (g[k] * F32(coef)).astype(F32)
    return total


def accumulate_micro_clipped(w, cfg: TextCfg, micro_batches, max_norm: float, world_batches=None):
    """--grad-clip-norm with --accum-freq > 1, no GradScaler (utils_AT.py:327-362): per micro-batch ``backward(loss / accum)``
    into the running sum, then clip_grad_norm_ ON that sum.  ``micro_batches``: list of (tokens, anchor).  Returns (grads the
    optimizer step sees, [total norm found at each micro-batch])."""
    accum = len(micro_batches)
    run: Dict[str, np.ndarray] = {}
    norms = []
    for toks, anchor in micro_batches:
        _, _, g = encode_text_backward(w, cfg, toks, anchor, accum_scale=1.0 / accum)
        for k, v in g.items():
            run[k] = (run[k] + v).astype(F32) if k in run else v.astype(F32)
        norms.append(clip_grad_norm(run, max_norm))
    return run, norms


def cosine_lr(base_lr: float, warmup: int, total_steps: int, step: int) -> float:
    """src/open_clip_train/scheduler.py:4-10,43-53."""
    if step < warmup:
        return base_lr * (step + 1) / warmup
    e, es = step - warmup, total_steps - warmup
    return 0.5 * (1 + math.cos(math.pi * e / es)) * base_lr


# --------------------------------------------------------------------------- rounding emulators
def round_bf16(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a, dtype=F32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(F32)


def round_fp16(a: np.ndarray) -> np.ndarray:
    return np.asarray(a, dtype=F32).astype(np.float16).astype(F32)


# ---- the 16 + 8-bit residual format of the engine's forward-only passes (leaf_amd/csrc/common.h resid_lo4 / resid_decode4; no
# counterpart in the reference, whose residual stream is fp32, src/open_clip/transformer.py:254-265).  Test infrastructure: the
# GPU suite holds the kernels to these byte-exact definitions.
def e4m3_encode(v: np.ndarray) -> np.ndarray:
    """fp32 -> OCP e4m3fn bytes (bias 7, three mantissa bits, subnormal step 2^-9, no inf), round-to-nearest-even; |v| <= 448."""
    x = np.ascontiguousarray(v, dtype=F32)
    u = x.view(np.uint32)
    sign = ((u >> 24) & 0x80).astype(np.uint8)
    mag = np.abs(x)
    r = (u & np.uint32(0x7FFFFFFF)) + np.uint32(0x7FFFF) + ((u >> 20) & np.uint32(1))
    normal = (((r >> 20) - np.uint32((127 - 7) << 3)) & np.uint32(0x7F)).astype(np.uint8)
    sub = np.rint(np.minimum(mag, F32(2.0 ** -6)).astype(np.float64) * 512.0).astype(np.uint8)
    return (np.where(mag < 2.0 ** -6, sub, normal) | sign).astype(np.uint8)


def e4m3_decode(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint8)
    e, m = ((b >> 3) & 15).astype(np.int32), (b & 7).astype(np.float64)
    mag = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * np.exp2((e - 7).astype(np.float64)))
    return np.where(b & 0x80, -mag, mag).astype(F32)


def _resid_chunk_scale(hi: np.ndarray, mant: int) -> np.ndarray:
    """the power of two the four values of a chunk share: 2^floor(log2(max(|hi| of the chunk, smallest normal)))"""
    floor = F32(2.0 ** -14) if mant == 10 else F32(2.0 ** -126)
    sc = np.maximum(np.abs(hi).reshape(-1, 4).max(-1), floor).astype(F32)
    return (sc.view(np.uint32) & np.uint32(0x7F800000)).view(F32)


def resid_pack(x: np.ndarray, mant: int = 10):
    """x (fp32, size % 4 == 0, chunks of 4 consecutive values) -> (hi: fp16(x) saturated -- bf16(x) for mant = 7 --, as fp32 values;
    lo8: e4m3((x - hi) * 2^(mant + 7) / chunk scale), the block-scaled remainder of common.h resid_lo4)."""
    x = np.ascontiguousarray(x, dtype=F32)
    shape = x.shape
    hi = (np.clip(x, -65504.0, 65504.0).astype(np.float16).astype(F32) if mant == 10 else np.ascontiguousarray(round_bf16(x))).reshape(-1)
    xmax = F32(65520.0) if mant == 10 else F32(3.0e38)
    d = ((np.clip(x.reshape(-1), -xmax, xmax) - hi).astype(F32) * F32(2.0 ** (mant + 7))).astype(F32)
    pow2 = np.repeat(_resid_chunk_scale(hi, mant), 4)
    with np.errstate(over="ignore", invalid="ignore"):
        lo8 = e4m3_encode((d / pow2).astype(F32))
    return hi.reshape(shape), lo8.reshape(shape)


def resid_unpack(hi: np.ndarray, lo8: np.ndarray, mant: int = 10) -> np.ndarray:
    hi = np.ascontiguousarray(hi, dtype=F32)
    shape = hi.shape
    pow2 = np.repeat(_resid_chunk_scale(hi.reshape(-1), mant), 4).astype(np.float64)
    r = e4m3_decode(np.asarray(lo8).reshape(-1).view(np.uint8)).astype(np.float64) * pow2
    return (r * 2.0 ** -(mant + 7) + hi.reshape(-1).astype(np.float64)).astype(F32).reshape(shape)
