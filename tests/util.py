import ctypes as C

import numpy as np


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def row_rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=-1) / (np.linalg.norm(b, axis=-1) + 1e-30)


def to16(x_np, dtype, device):
    """numpy fp32 -> CUDA 16-bit tensor of the given operand type ('fp16' | 'bf16')."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x_np, dtype=np.float32)).to(device)
    return t.to(torch.float16 if dtype == "fp16" else torch.bfloat16).contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# north_star's embedding gate: rel-L2 <= 1e-3 PER ROW against the fp32 reference / oracle on the BASELINE.json towers (ViT-L, ViT-H, bigG).
# Round 4: tightened from 1.25e-3 after measuring every row these tests look at (tests/row_error_survey.py,
# profiles/r04_row_error_survey.txt: maxima 9.3e-4 ... 9.9e-4); the forward is deterministic, so a test that passes, passes again.
TOL_ROW = 1.0e-3


def trained_like_weights(cfg, seed=0, alpha=0.7, gain_sigma=0.5):
    """A text tower whose weights LOOK like a trained CLIP's as far as that can be synthesised offline (pretrained
    weights need the network): every GEMM weight gets a power-law singular spectrum s_i ~ i^-alpha (random near-orthogonal factors,
    Frobenius norm of the random init kept, so activations keep their scale), LayerNorm gains are log-normal per channel with a few
    large outlier channels, LayerNorm / linear biases are non-zero, token embeddings have a heavy-tailed norm distribution.  Starts
    from ``oracle.init_weights(cfg, seed=1)`` (shapes, names)."""
    from oracle import text_oracle as O
    rng = np.random.default_rng(seed)
    w = {k: v.copy() for k, v in O.init_weights(cfg, seed=1).items()}
    for k, v in w.items():
        if v.ndim == 2 and k.endswith(("in_proj_weight", "out_proj.weight", "c_fc.weight", "c_proj.weight")) or k == "text_projection":
            n, m = v.shape
            r = min(n, m)
            a = rng.standard_normal((n, r)).astype(np.float32) / np.sqrt(n)
            b = rng.standard_normal((m, r)).astype(np.float32) / np.sqrt(m)
            s = (np.arange(1, r + 1, dtype=np.float32) ** -alpha)
            new = (a * s[None, :]) @ b.T
            w[k] = (new * (np.linalg.norm(v) / np.linalg.norm(new))).astype(np.float32)
        elif k.endswith(("ln_1.weight", "ln_2.weight")) or k == "ln_final.weight":
            g = np.exp(gain_sigma * rng.standard_normal(v.shape)).astype(np.float32)
            g[rng.choice(v.size, 4, replace=False)] *= 6.0            # outlier channels
            w[k] = g
        elif k.endswith(("ln_1.bias", "ln_2.bias")) or k == "ln_final.bias":
            w[k] = (0.1 * rng.standard_normal(v.shape)).astype(np.float32)
        elif k.endswith("bias"):
            w[k] = (0.02 * rng.standard_normal(v.shape)).astype(np.float32)
        elif k == "token_embedding.weight":
            scale = np.exp(0.4 * rng.standard_normal((v.shape[0], 1))).astype(np.float32)      # heavy-tailed token norms
            w[k] = (v * scale).astype(np.float32)
    return w
