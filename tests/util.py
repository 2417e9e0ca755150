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
