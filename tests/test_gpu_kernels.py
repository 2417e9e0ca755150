"""Per-kernel parity: each HIP kernel through its C-ABI hook vs a numpy restatement on the same inputs."""
import math

import numpy as np
import pytest

from oracle import text_oracle as O
from tests.util import ptr, rel_l2, stream, to16

pytestmark = pytest.mark.gpu
DT = {"bf16": 0, "fp16": 1}


@pytest.fixture(scope="module")
def env():
    import torch
    from leaf_amd import _lib
    assert torch.cuda.is_available()
    return _lib.lib(), torch, torch.device("cuda:0")


def _round(x, dtype):
    return O.round_fp16(x) if dtype == "fp16" else O.round_bf16(x)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(77 * 3, 128, 64), (300, 384, 128), (1000, 256, 512), (2100, 256, 128), (4096 + 77, 512, 192), (2048, 256, 768), (2500, 768, 3072),
                                   (11085, 768, 256), (8300, 1024, 192)])   # >= 128 tiles of 256^2: the ring kernel
def test_gemm_epilogues(env, dtype, M, N, K):
    lib, torch, dev = env
    from leaf_amd import _lib
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K), dtype=np.float32)
    # asymmetric B so that a transposed C-write cannot pass
    B = (rng.standard_normal((N, K), dtype=np.float32) * 0.1 + np.arange(N, dtype=np.float32)[:, None] * 1e-3)
    bias = rng.standard_normal(N).astype(np.float32)
    Ar, Br = _round(A, dtype), _round(B, dtype)
    ref = (Ar.astype(np.float64) @ Br.astype(np.float64).T)
    a16, b16 = to16(A, dtype, dev), to16(B, dtype, dev)
    tbias = torch.from_numpy(bias).to(dev)
    t16 = torch.float16 if dtype == "fp16" else torch.bfloat16
    # epi 0: 16-bit store with bias
    c = torch.zeros(M, N, dtype=t16, device=dev)
    _lib.check(lib.leaf_op_gemm(DT[dtype], 0, ptr(a16), ptr(b16), ptr(c), ptr(tbias), None, M, N, K, 0, 0.0, 0, stream()), "gemm0")
    torch.cuda.synchronize()
    got = c.float().cpu().numpy()
    assert rel_l2(got, ref + bias) < (2e-3 if dtype == "fp16" else 1e-2)
    # epi 3: fp32 store, beta = 0 then beta = 1 (accumulate)
    c32 = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
    _lib.check(lib.leaf_op_gemm(DT[dtype], 3, ptr(a16), ptr(b16), ptr(c32), None, None, M, N, K, 0, 0.0, 0, stream()), "gemm3")
    torch.cuda.synchronize()
    got = c32.cpu().numpy()
    assert rel_l2(got, ref) < 1e-5, "fp32-accumulated product of the rounded operands must match to fp32 noise"
    _lib.check(lib.leaf_op_gemm(DT[dtype], 3, ptr(a16), ptr(b16), ptr(c32), None, None, M, N, K, 0, 1.0, 0, stream()), "gemm3b")
    torch.cuda.synchronize()
    assert rel_l2(c32.cpu().numpy(), 2 * ref) < 1e-5
    # epi 2: residual add in place
    x0 = rng.standard_normal((M, N)).astype(np.float32)
    x = torch.from_numpy(x0).to(dev)
    _lib.check(lib.leaf_op_gemm(DT[dtype], 2, ptr(a16), ptr(b16), ptr(x), ptr(tbias), None, M, N, K, 0, 0.0, 0, stream()), "gemm2")
    torch.cuda.synchronize()
    assert rel_l2(x.cpu().numpy(), x0 + ref + bias) < 1e-5
    # epi 1: activation (+ pre-activation stash), both activations
    for act, fn in ((0, O.gelu), (1, O.quick_gelu)):
        c = torch.zeros(M, N, dtype=t16, device=dev)
        pre = torch.zeros(M, N, dtype=t16, device=dev)
        _lib.check(lib.leaf_op_gemm(DT[dtype], 1, ptr(a16), ptr(b16), ptr(c), ptr(tbias), ptr(pre), M, N, K, act, 0.0, 0, stream()), "gemm1")
        torch.cuda.synchronize()
        want = fn((ref + bias).astype(np.float32))
        assert rel_l2(c.float().cpu().numpy(), want) < (2e-3 if dtype == "fp16" else 1e-2)
        assert rel_l2(pre.float().cpu().numpy(), ref + bias) < (2e-3 if dtype == "fp16" else 1e-2)
    # epi 4: acc * act'(aux)
    for act in (0, 1):
        cfg = O.TextCfg(quick_gelu=bool(act))
        prev = rng.standard_normal((M, N)).astype(np.float32)
        aux = to16(prev, "fp16", dev)
        c = torch.zeros(M, N, dtype=t16, device=dev)
        _lib.check(lib.leaf_op_gemm(DT[dtype], 4, ptr(a16), ptr(b16), ptr(c), None, ptr(aux), M, N, K, act, 0.0, 1, stream()), "gemm4")
        torch.cuda.synchronize()
        want = ref * O._act_grad(cfg, O.round_fp16(prev))
        assert rel_l2(c.float().cpu().numpy(), want) < (2e-3 if dtype == "fp16" else 1e-2)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("d", [128, 768, 1280])
def test_layernorm(env, dtype, d):
    lib, torch, dev = env
    from leaf_amd import _lib
    rng = np.random.default_rng(d)
    rows = 77 * 2 + 3
    x = (rng.standard_normal((rows, d)) * 3 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(d)).astype(np.float32)
    want, _, _ = O.layer_norm(x, g, b, 1e-5)
    out = torch.zeros(rows, d, dtype=torch.float16 if dtype == "fp16" else torch.bfloat16, device=dev)
    tx, tg, tb = (torch.from_numpy(v).to(dev) for v in (x, g, b))
    _lib.check(lib.leaf_op_layernorm(ptr(tx), ptr(tg), ptr(tb), 1e-5, ptr(out), rows, d, DT[dtype], stream()), "ln")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.array_equal(got, _round(want, dtype)) or rel_l2(got, want) < (5e-4 if dtype == "fp16" else 4e-3)


def _attention_ref(qkv, n, L, H):
    d = H * 64
    q, k, v = (qkv.reshape(n, L, 3, H, 64)[:, :, i].transpose(0, 2, 1, 3).astype(np.float64) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / 8.0 + np.triu(np.full((L, L), -np.inf), 1)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(n * L, d), p, q, k, v


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("L,H,n", [(77, 2, 3), (77, 12, 5), (16, 2, 2), (33, 2, 1), (96, 2, 2)])
def test_attention_fwd(env, dtype, L, H, n):
    lib, torch, dev = env
    from leaf_amd import _lib
    d = H * 64
    rng = np.random.default_rng(L * 100 + H)
    qkv = rng.standard_normal((n * L, 3 * d)).astype(np.float32)
    qkv[:, :d] *= 2.0   # sharper softmax
    qr = _round(qkv, dtype)
    want, _, _, _, _ = _attention_ref(qr, n, L, H)
    t = to16(qkv, dtype, dev)
    out = torch.zeros(n * L, d, dtype=t.dtype, device=dev)
    _lib.check(lib.leaf_op_attention_fwd(ptr(t), ptr(out), n, L, H, d, DT[dtype], stream()), "attn")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_l2(got, want) < (2e-3 if dtype == "fp16" else 1.2e-2)


@pytest.mark.parametrize("L,H,n", [(77, 2, 2), (20, 2, 1)])
def test_attention_bwd(env, L, H, n):
    lib, torch, dev = env
    from leaf_amd import _lib
    d = H * 64
    rng = np.random.default_rng(7)
    qkv = rng.standard_normal((n * L, 3 * d)).astype(np.float32)
    do = rng.standard_normal((n * L, d)).astype(np.float32)
    qr, dor = O.round_fp16(qkv), O.round_bf16(do)
    _, p, q, k, v = _attention_ref(qr, n, L, H)
    dO = dor.reshape(n, L, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    dv = p.transpose(0, 1, 3, 2) @ dO
    dp = dO @ v.transpose(0, 1, 3, 2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / 8.0
    dq, dk = ds @ k, ds.transpose(0, 1, 3, 2) @ q
    want = np.stack([dq, dk, dv], 0).transpose(1, 3, 0, 2, 4).reshape(n * L, 3 * d)
    tq = to16(qkv, "fp16", dev)
    tdo = to16(do, "bf16", dev)
    out = torch.zeros(n * L, 3 * d, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.leaf_op_attention_bwd(ptr(tq), 1, ptr(tdo), ptr(out), n, L, H, d, stream()), "attn_bwd")
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu().numpy(), want) < 5e-3
