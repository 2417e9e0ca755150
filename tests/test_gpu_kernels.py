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


@pytest.mark.parametrize("fdt,gdt", [("fp16", "fp16"), ("fp16", "bf16"), ("bf16", "bf16")])
@pytest.mark.parametrize("L,H,n", [(77, 2, 2), (20, 12, 3), (48, 2, 1), (96, 2, 1)])
def test_attention_bwd_mfma(env, fdt, gdt, L, H, n):
    """MFMA attention backward (attention_bwd.hip) for every forward / gradient dtype pair."""
    lib, torch, dev = env
    from leaf_amd import _lib
    d = H * 64
    rng = np.random.default_rng(L + H)
    qkv = rng.standard_normal((n * L, 3 * d)).astype(np.float32)
    do = rng.standard_normal((n * L, d)).astype(np.float32)
    qr, dor = _round(qkv, fdt), _round(do, gdt)
    _, p, q, k, v = _attention_ref(qr, n, L, H)
    dO = dor.reshape(n, L, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    dv = p.transpose(0, 1, 3, 2) @ dO
    dp = dO @ v.transpose(0, 1, 3, 2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / 8.0
    dq, dk = ds @ k, ds.transpose(0, 1, 3, 2) @ q
    want = np.stack([dq, dk, dv], 0).transpose(1, 3, 0, 2, 4).reshape(n * L, 3 * d)
    tq, tdo = to16(qkv, fdt, dev), to16(do, gdt, dev)
    out = torch.zeros(n * L, 3 * d, dtype=tdo.dtype, device=dev)
    _lib.check(lib.leaf_op_attention_bwd_t(ptr(tq), DT[fdt], ptr(tdo), ptr(out), DT[gdt], n, L, H, d, stream()), "attn_bwd")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_l2(got, want) < (2e-3 if gdt == "fp16" else 1.2e-2)


@pytest.mark.parametrize("xdt,gdt", [("fp16", "fp16"), ("fp16", "bf16"), ("bf16", "fp16"), ("bf16", "bf16")])
@pytest.mark.parametrize("rows,Nw,Kw", [(77 * 3, 256, 128), (3200, 768, 3072), (1000, 3072, 768), (33, 128, 128), (3219, 2304, 768)])
def test_wgrad_group(env, xdt, gdt, rows, Nw, Kw):
    """dW += alpha dY^T X and db += alpha colsum(dY) (wgrad.hip) against float64 numpy on the rounded operands."""
    lib, torch, dev = env
    from leaf_amd import _lib
    rng = np.random.default_rng(rows + Nw)
    dy = rng.standard_normal((rows, Nw)).astype(np.float32)
    x = rng.standard_normal((rows, Kw)).astype(np.float32)
    w0 = rng.standard_normal((Nw, Kw)).astype(np.float32)
    b0 = rng.standard_normal(Nw).astype(np.float32)
    alpha = 0.25
    dyr = _round(dy, gdt).astype(np.float64)
    xr = _round(_round(x, xdt), gdt).astype(np.float64)     # X is converted to the gradient type when they differ
    want_w = w0 + alpha * (dyr.T @ xr)
    want_b = b0 + alpha * dyr.sum(0)
    tdy, tx = to16(dy, gdt, dev), to16(x, xdt, dev)
    tw, tb = torch.from_numpy(w0).to(dev), torch.from_numpy(b0).to(dev)
    ta = torch.tensor([alpha], dtype=torch.float32, device=dev)
    _lib.check(lib.leaf_op_wgrad(ptr(tdy), ptr(tx), ptr(tw), ptr(tb), rows, Nw, Kw, DT[xdt], DT[gdt], ptr(ta), stream()), "wgrad")
    torch.cuda.synchronize()
    assert rel_l2(tw.cpu().numpy(), want_w) < 2e-6
    assert rel_l2(tb.cpu().numpy(), want_b) < 2e-6


@pytest.mark.parametrize("with_params", [True, False])
@pytest.mark.parametrize("rows,d", [(3219, 768), (800, 1024), (130, 1280), (7, 128), (40, 2048)])
def test_layernorm_bwd(env, with_params, rows, d):
    """LayerNorm backward (train.hip ln_bwd_kernel: dx += dLN/dx, 16-bit copy, dg / db += column sums / S) against float64
    numpy; every tower width has its own instantiation (float4 chunks per lane, waves per workgroup)."""
    lib, torch, dev = env
    from leaf_amd import _lib
    rng = np.random.default_rng(rows + d)
    x = (rng.standard_normal((rows, d)) * 2 + 0.3).astype(np.float32)
    dy = rng.standard_normal((rows, d)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    dx0 = rng.standard_normal((rows, d)).astype(np.float32)
    dg0, db0 = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
    S = 8.0
    x64, dy64 = x.astype(np.float64), dy.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(x64.var(-1, keepdims=True) + 1e-5)
    xhat = (x64 - mu) * rstd
    dxhat = dy64 * g
    want_dx = dx0 + rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    want_dg = dg0 + (dy64 * xhat).sum(0) / S
    want_db = db0 + dy64.sum(0) / S
    tx, tdy, tg, tdx, tdg, tdb = (torch.from_numpy(a.copy()).to(dev) for a in (x, dy, g, dx0, dg0, db0))
    t16 = torch.zeros(rows, d, dtype=torch.float16, device=dev)
    gs = torch.tensor([S, 1.0 / S], dtype=torch.float32, device=dev)
    ws = torch.empty(lib.leaf_op_layernorm_bwd_ws_bytes(rows, d), dtype=torch.uint8, device=dev)
    _lib.check(lib.leaf_op_layernorm_bwd(ptr(tdy), ptr(tx), ptr(tg), 1e-5, ptr(tdx), ptr(t16), DT["fp16"], ptr(gs),
                                         ptr(tdg) if with_params else None, ptr(tdb) if with_params else None, rows, d,
                                         ptr(ws), ws.numel(), stream()), "layernorm_bwd")
    torch.cuda.synchronize()
    got = tdx.cpu().numpy()
    assert rel_l2(got, want_dx) < 2e-6
    assert np.array_equal(t16.cpu().numpy(), np.clip(got, -65504, 65504).astype(np.float16))
    if with_params:
        assert rel_l2(tdg.cpu().numpy(), want_dg) < 5e-6 and rel_l2(tdb.cpu().numpy(), want_db) < 5e-6
    else:
        assert np.array_equal(tdg.cpu().numpy(), dg0) and np.array_equal(tdb.cpu().numpy(), db0)


@pytest.mark.parametrize("normalize", [0, 1])
@pytest.mark.parametrize("M,d,D", [(300, 768, 768), (64, 1024, 1024), (1, 1280, 1280), (130, 128, 128)])
def test_project_rows(env, normalize, M, d, D):
    """ln_final + text projection of pooled rows on the fp32 matrix cores (project.hip) against float64 numpy."""
    lib, torch, dev = env
    from leaf_amd import _lib
    rng = np.random.default_rng(M + d)
    x = (rng.standard_normal((M, d)) * 3 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(d)).astype(np.float32)
    proj = (rng.standard_normal((d, D)) * d ** -0.5).astype(np.float32)
    x64 = x.astype(np.float64)
    xn = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + b
    want = xn @ proj.astype(np.float64)
    if normalize:
        want /= np.maximum(np.linalg.norm(want, axis=-1, keepdims=True), 1e-12)
    tx, tg, tb, tp = (torch.from_numpy(a).to(dev) for a in (x, g, b, proj))
    scratch = torch.empty(M, d, dtype=torch.float32, device=dev)
    out = torch.zeros(M, D, dtype=torch.float32, device=dev)
    _lib.check(lib.leaf_op_project_rows(ptr(tx), ptr(tg), ptr(tb), 1e-5, ptr(tp), ptr(scratch), ptr(out), M, d, D, normalize,
                                        stream()), "project_rows")
    torch.cuda.synchronize()
    assert rel_l2(out.cpu().numpy(), want) < 3e-6


@pytest.mark.parametrize("N,K", [(768, 768), (768, 3072), (3072, 768)])
def test_gemm_rows_do_not_depend_on_the_kernel(env, N, K):
    """The tile-count dispatch picks the 256^2 LDS-DMA kernel for big launches and the 128^2 kernel for small ones.
    Every kernel accumulates k in ascending 32-steps into one fp32 accumulator per element and applies the same epilogue
    arithmetic, so a row's result must be BIT-identical whichever kernel computed it: the exactness of the prefix reuse
    (cache rows from a small launch, candidate rows from a big one) and of chunked passes rests on this."""
    lib, torch, dev = env
    from leaf_amd import _lib
    M, CH = 12000, 2000                       # 47 x N/256 >= 128 tiles in one launch; 8 x N/256 < 128 per chunk
    rng = np.random.default_rng(N + K)
    a16 = to16(rng.standard_normal((M, K), dtype=np.float32), "fp16", dev)
    b16 = to16(rng.standard_normal((N, K), dtype=np.float32) * 0.05, "fp16", dev)
    bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    x0 = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(dev)

    def run(epi, rows, act=0):
        outs = []
        for r0 in range(0, M, rows):
            r1 = min(M, r0 + rows)
            a = a16[r0:r1]
            if epi in (0, 1):
                c = torch.zeros(r1 - r0, N, dtype=torch.float16, device=dev)
            else:
                c = x0[r0:r1].clone()
            _lib.check(lib.leaf_op_gemm(1, epi, ptr(a), ptr(b16), ptr(c), ptr(bias), None, r1 - r0, N, K, act,
                                        1.0 if epi == 3 else 0.0, 0, stream()), "gemm")
            outs.append(c)
        torch.cuda.synchronize()
        return torch.cat(outs)

    for epi, act in ((0, 0), (1, 0), (1, 1), (2, 0), (3, 0)):
        assert torch.equal(run(epi, M, act), run(epi, CH, act)), f"epilogue {epi} act {act}"


@pytest.mark.parametrize("Ka,N", [(768, 768), (1024, 1024)])
def test_gemm_that_rereads_its_a_panel_equals_the_launch_on_a_duplicated_panel(env, Ka, N):
    """Round 6 (GemmArgs::a_wrap): the out-projection of a split block is x += [A | A] [W_hi | W_lo]^T in ONE launch; the half-stage
    ring kernel restarts A's k offset after Ka / 64 tiles instead of reading a materialised [M, 2 Ka] copy.  Same operands, same k
    order: BIT-identical to the plain launch on the duplicated panel -- whichever kernel that one runs in (big launch: the ring
    kernel; chunks of 2,000 rows: the small-launch kernel, which is what the B-caption passes use)."""
    lib, torch, dev = env
    from leaf_amd import _lib
    M, CH = 12000, 2000
    rng = np.random.default_rng(Ka)
    a16 = to16(rng.standard_normal((M, Ka), dtype=np.float32), "fp16", dev)
    b16 = to16(rng.standard_normal((N, 2 * Ka), dtype=np.float32) * 0.05, "fp16", dev)
    bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    x0 = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(dev)
    wrap = x0.clone()
    _lib.check(lib.leaf_op_gemm_awrap(1, ptr(a16), ptr(b16), ptr(wrap), ptr(bias), M, N, Ka, stream()), "gemm_awrap")
    a2 = torch.cat([a16, a16], 1).contiguous()
    for rows in (M, CH):
        outs = []
        for r0 in range(0, M, rows):
            c = x0[r0:r0 + rows].clone()
            _lib.check(lib.leaf_op_gemm(1, 2, ptr(a2[r0:r0 + rows]), ptr(b16), ptr(c), ptr(bias), None, c.shape[0], N, 2 * Ka, 0, 0.0, 0, stream()), "gemm")
            outs.append(c)
        torch.cuda.synchronize()
        assert torch.equal(wrap, torch.cat(outs)), rows
    # fp32 check of the sum itself
    ref = x0.double() + a2.double() @ b16.double().T + bias.double()
    assert float((wrap.double() - ref).norm() / ref.norm()) < 1e-5
    small = x0[:CH].clone()           # a launch the ring kernel does not take must be refused, not mis-computed
    assert lib.leaf_op_gemm_awrap(1, ptr(a16), ptr(b16), ptr(small), ptr(bias), CH, N, Ka, stream()) != 0


@pytest.mark.parametrize("M,K", [(12000, 768), (2000, 768), (500, 128)])
def test_forward_gemm_outputs_saturate_at_fp16_max_in_every_kernel_family(env, M, K):
    """F16::pack2 relies on MODE.FP16_OVFL (leaf_fp16_sat_mode at kernel entry) for the +-65504 saturation of every
    16-bit store.  A product far beyond fp16's range must come out as +-65504 (0x7bff / 0xfbff), never inf / NaN, from the
    half-stage ring kernel (12,000 rows), the small-launch ring kernel (2,000 rows) and the register-staged kernel (K = 128),
    with the plain 16-bit-store epilogue and with the activation epilogue."""
    lib, torch, dev = env
    from leaf_amd import _lib
    N = 768
    a = torch.full((M, K), 200.0, dtype=torch.float16, device=dev)
    a[1::2] = -200.0
    b = torch.full((N, K), 200.0, dtype=torch.float16, device=dev)
    bias = torch.zeros(N, dtype=torch.float32, device=dev)
    for epi, act in ((0, 0), (1, 0)):
        c = torch.zeros(M, N, dtype=torch.float16, device=dev)
        _lib.check(lib.leaf_op_gemm(1, epi, ptr(a), ptr(b), ptr(c), ptr(bias), None, M, N, K, act, 0.0, 0, stream()), "gemm")
        torch.cuda.synchronize()
        bits = c.view(torch.int16)
        assert bool(torch.isfinite(c).all()), (epi, M, K)
        assert bool((bits[0::2] == 0x7bff).all()), (epi, M, K)
        if epi == 0:
            assert bool((bits[1::2] == -0x0401).all()), (epi, M, K)       # 0xfbff = -65504
        else:
            assert bool((c[1::2].float().abs() < 1e-3).all())             # GELU(-1e7) = -0


def _fold_problem(rng, M, d, N, dev, torch):
    """A residual GEMM that PRODUCES a row (out_proj shape d x d) followed by the LN-folded consumer (N x d)."""
    A = rng.standard_normal((M, d), dtype=np.float32)
    Wo = (rng.standard_normal((d, d), dtype=np.float32) * 0.05).astype(np.float32)
    bo = rng.standard_normal(d).astype(np.float32)
    x0 = (rng.standard_normal((M, d)) * 2.0 + rng.standard_normal((M, 1)) * 3.0).astype(np.float32)   # rows with a sizeable mean
    x0[:, 5] += 300.0                                                                                  # and one outlier channel
    g = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(d)).astype(np.float32)
    W = (rng.standard_normal((N, d)) * 0.05 + np.arange(N)[:, None] * 1e-4).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    return A, Wo, bo, x0, g, beta, W, bias


@pytest.mark.parametrize("M,d,N", [(300, 128, 256), (3000, 768, 2304), (40000, 768, 3072), (9000, 1280, 1280)])
def test_lnfold_gemm_pair_vs_numpy(env, M, d, N):
    """lnfold.h: producer epilogue (fp32 residual + 16-bit copy + per-group statistics) and consumer epilogue
    (rstd (acc - mean s) + c [+ activation]) against a float64 LayerNorm + linear of the same rows; small launches run the
    register-direct kernels, the 40,000-row one the 256^2 half-stage kernel."""
    lib, torch, dev = env
    from leaf_amd import _lib
    rng = np.random.default_rng(M + d + N)
    A, Wo, bo, x0, g, beta, W, bias = _fold_problem(rng, M, d, N, dev, torch)
    a16, wo16 = to16(A, "fp16", dev), to16(Wo, "fp16", dev)
    x = torch.from_numpy(x0).to(dev)
    x16 = torch.zeros(M, d, dtype=torch.float16, device=dev)
    stat = torch.zeros(d // 64, M, 2, dtype=torch.float32, device=dev)
    _lib.check(lib.leaf_op_gemm_resid_ln(1, ptr(a16), ptr(wo16), ptr(x), ptr(torch.from_numpy(bo).to(dev)), ptr(x16), ptr(stat),
                                         M, d, d, stream()), "resid_ln")
    torch.cuda.synchronize()
    xr = x.cpu().numpy()
    want_x = x0 + O.round_fp16(A).astype(np.float64) @ O.round_fp16(Wo).astype(np.float64).T + bo
    assert rel_l2(xr, want_x) < 1e-5
    assert torch.equal(x16, x.half()), "x16 must be the 16-bit rounding of the fp32 result"
    grp = xr.astype(np.float64).reshape(M, d // 64, 64)
    st = stat.cpu().numpy()
    assert np.allclose(st[:, :, 0].T, grp.sum(-1), rtol=1e-5, atol=1e-3)
    assert np.allclose(st[:, :, 1].T, ((grp - grp.mean(-1, keepdims=True)) ** 2).sum(-1), rtol=1e-4)
    # consumer: gamma-scaled weights, s, c as the pack kernel builds them
    wp = O.round_fp16((W * g[None, :]).astype(np.float32))
    s_vec = wp.sum(-1, dtype=np.float64).astype(np.float32)
    c_vec = (W.astype(np.float64) @ beta + bias).astype(np.float32)
    wp16 = to16(wp, "fp16", dev)
    ts, tc = torch.from_numpy(s_vec).to(dev), torch.from_numpy(c_vec).to(dev)
    x64 = xr.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(x64.var(-1, keepdims=True) + 1e-5)
    ln16 = ((x16.float().cpu().numpy().astype(np.float64) - mu) * rstd)                # what the folded GEMM effectively multiplies
    want = ln16 @ wp.astype(np.float64).T + c_vec
    exact = ((x64 - mu) * rstd * g + beta) @ W.astype(np.float64).T + bias             # the unrounded LayerNorm + linear
    rowstat = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    _lib.check(lib.leaf_op_ln_finalize(ptr(stat), M, M, d // 64, 1e-5, ptr(rowstat), stream()), "ln_finalize")
    torch.cuda.synchronize()
    rsn = rowstat.cpu().numpy()
    assert np.allclose(rsn[:, 0], mu[:, 0], rtol=1e-5, atol=1e-5) and np.allclose(rsn[:, 1], rstd[:, 0], rtol=2e-6)
    for act, fn in ((-1, None), (0, O.gelu), (1, O.quick_gelu)):
        y = torch.zeros(M, N, dtype=torch.float16, device=dev)
        _lib.check(lib.leaf_op_gemm_lnfold(1, act, ptr(x16), ptr(wp16), ptr(y), ptr(tc), ptr(ts), ptr(rowstat), M, N, d, stream()),
                   "lnfold")
        torch.cuda.synchronize()
        w_ = want if fn is None else fn(want.astype(np.float32))
        e_ = exact if fn is None else fn(exact.astype(np.float32))
        got = y.float().cpu().numpy()
        assert rel_l2(got, w_) < 1e-3, act          # same operands: fp32 accumulation + one fp16 rounding of the output
        assert rel_l2(got, e_) < 3e-3, act          # against the exact LN + linear: operand rounding only


@pytest.mark.parametrize("d,N", [(768, 2304), (768, 3072), (1024, 4096)])
def test_lnfold_rows_do_not_depend_on_the_kernel(env, d, N):
    """The folded pair must give BIT-identical rows whichever kernel family the tile-count dispatch picks for the producer and
    for the consumer (one big launch: 256^2 half-stage ring; chunks of 1,500 rows: the 64 x 128 ring; both orders mixed)."""
    lib, torch, dev = env
    from leaf_amd import _lib
    M, CH = 24000, 1500
    rng = np.random.default_rng(d + N)
    A, Wo, bo, x0, g, beta, W, bias = _fold_problem(rng, M, d, N, dev, torch)
    a16, wo16 = to16(A, "fp16", dev), to16(Wo, "fp16", dev)
    tbo = torch.from_numpy(bo).to(dev)
    wp = O.round_fp16((W * g[None, :]).astype(np.float32))
    wp16 = to16(wp, "fp16", dev)
    ts = torch.from_numpy(wp.sum(-1, dtype=np.float64).astype(np.float32)).to(dev)
    tc = torch.from_numpy((W.astype(np.float64) @ beta + bias).astype(np.float32)).to(dev)

    def run(rows_p, rows_c):
        x = torch.from_numpy(x0).to(dev)
        x16 = torch.zeros(M, d, dtype=torch.float16, device=dev)
        stat_chunks, ys = [], []
        for r0 in range(0, M, rows_p):
            r1 = min(M, r0 + rows_p)
            st = torch.zeros(d // 64, r1 - r0, 2, dtype=torch.float32, device=dev)
            _lib.check(lib.leaf_op_gemm_resid_ln(1, ptr(a16[r0:r1]), ptr(wo16), ptr(x[r0:r1]), ptr(tbo), ptr(x16[r0:r1]), ptr(st),
                                                 r1 - r0, d, d, stream()), "resid_ln")
            stat_chunks.append(st)
        stat = torch.cat(stat_chunks, 1).contiguous()
        rowstat = torch.zeros(M, 2, dtype=torch.float32, device=dev)
        _lib.check(lib.leaf_op_ln_finalize(ptr(stat), M, M, d // 64, 1e-5, ptr(rowstat), stream()), "ln_finalize")
        for r0 in range(0, M, rows_c):
            r1 = min(M, r0 + rows_c)
            y = torch.zeros(r1 - r0, N, dtype=torch.float16, device=dev)
            _lib.check(lib.leaf_op_gemm_lnfold(1, 1, ptr(x16[r0:r1]), ptr(wp16), ptr(y), ptr(tc), ptr(ts), ptr(rowstat[r0:r1]), r1 - r0, N, d,
                                               stream()), "lnfold")
            ys.append(y)
        torch.cuda.synchronize()
        return x, x16, stat, torch.cat(ys)

    ref = run(M, M)
    for rp, rc in ((CH, CH), (M, CH), (CH, M)):
        got = run(rp, rc)
        for name, a, b in zip(("x", "x16", "stat", "y"), ref, got):
            assert torch.equal(a, b), f"{name} differs between launch sizes producer={rp} consumer={rc}"


# ---------------------------------------------------------------- the 16 + 8-bit residual stream (round 5)
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_resid16_8_format_is_the_oracles_byte_for_byte(env, dtype):
    """common.h resid_lo4 / resid_decode4 against oracle.resid_pack / resid_unpack: the 16-bit half is fp16(x) (saturating), the
    remainder byte is e4m3((x - hi) * 2^(mant + 7) / the chunk's power of two) (four consecutive values share the exponent of their
    largest |hi|; CDNA4's scaled MX conversions, nearest-even), decoding is exact arithmetic -- over twenty decades of magnitudes,
    ties, binade boundaries, fp16 subnormals, zero, the largest fp16 and beyond it (finite, never inf / NaN)."""
    lib, torch, dev = env
    from leaf_amd import _lib
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(1 << 18) * np.exp(rng.uniform(-25, 11, 1 << 18))).astype(np.float32)
    hi_grid = rng.standard_normal(4096).astype(np.float16).astype(np.float32)
    ties = hi_grid + np.abs(hi_grid) * np.float32(2.0 ** -11) * rng.choice(np.float32([0.125, 0.375, 0.5, 0.625, 0.875, -0.375]), 4096)
    x = np.concatenate([x, ties.astype(np.float32), np.float32([0, -0.0, 65504, -65504, 65519.9, -65519.9, 6e-8, 2e-8, 6.1e-5, 1, -1, 0.5,
                                                                1e9, -1e9, 0.99999994, 1.0000001, 2047.9999, 2048.5])])
    x = x[: x.size // 4 * 4].copy()
    f16 = dtype == "fp16"
    mant, did = (10, 1) if f16 else (7, 0)
    tx = torch.from_numpy(x).to(dev)
    x16 = torch.zeros(x.size, dtype=torch.float16 if f16 else torch.bfloat16, device=dev)
    lo8 = torch.zeros(x.size, dtype=torch.uint8, device=dev)
    back = torch.zeros(x.size, dtype=torch.float32, device=dev)
    _lib.check(lib.leaf_op_resid_pack(did, ptr(tx), ptr(x16), ptr(lo8), x.size, stream()), "resid_pack")
    _lib.check(lib.leaf_op_resid_unpack(did, ptr(x16), ptr(lo8), ptr(back), x.size, stream()), "resid_unpack")
    torch.cuda.synchronize()
    hi, lo = O.resid_pack(x, mant)
    assert np.array_equal(x16.float().cpu().numpy(), hi)
    got_lo = lo8.cpu().numpy()
    same = (got_lo == lo) | (((got_lo | lo) & 0x7F) == 0)          # (+0 and -0 remainders decode alike)
    assert same.all(), (x[~same][:8], got_lo[~same][:8], lo[~same][:8])
    want = O.resid_unpack(hi, got_lo, mant)
    assert np.array_equal(back.cpu().numpy(), want) and np.isfinite(want).all()
    # every value of a chunk of four to 2^-(mant + 6) of the chunk's largest magnitude (inside the 16-bit type's range)
    cmax = np.repeat(np.abs(x).reshape(-1, 4).max(-1), 4)
    ok = (cmax > 1e-3) & (cmax < 65504)
    assert np.max(np.abs(want[ok].astype(np.float64) - x[ok]) / cmax[ok]) <= 2.0 ** -(mant + 6) * 1.0001


@pytest.mark.parametrize("M,d,K", [(300, 128, 128), (3000, 768, 768), (3000, 768, 3072), (40000, 768, 768), (33000, 768, 3072)])
def test_resid16_8_gemm_vs_numpy(env, M, d, K):
    """EPI_RESID_LN8 (the residual GEMMs of the scoring passes on the 16 + 8-bit stream): decode(x16, lo8) + A W^T + bias in float64,
    re-encoded -- the stored pair must be the oracle's packing of the kernel's own fp32 value, i.e. the 16-bit half the rounding of
    something within fp32 accumulation noise of the float64 result and the pair within 2^-16 of its chunk's largest value; statistics as for EPI_RESID_LN.
    Small launches run the register-direct kernels, the large ones the 256^2 half-stage kernel."""
    lib, torch, dev = env
    from leaf_amd import _lib
    rng = np.random.default_rng(M + d + K)
    A = rng.standard_normal((M, K), dtype=np.float32)
    Wo = (rng.standard_normal((d, K), dtype=np.float32) * 0.05).astype(np.float32)
    bo = rng.standard_normal(d).astype(np.float32)
    x0 = (rng.standard_normal((M, d)) * 2.0 + rng.standard_normal((M, 1)) * 3.0).astype(np.float32)
    x0[:, 5] += 300.0
    hi0, lo0 = O.resid_pack(x0)
    a16, wo16 = to16(A, "fp16", dev), to16(Wo, "fp16", dev)
    x16 = torch.from_numpy(hi0).to(dev).half()
    lo8 = torch.from_numpy(lo0).to(dev)
    stat = torch.zeros(d // 64, M, 2, dtype=torch.float32, device=dev)
    _lib.check(lib.leaf_op_gemm_resid_ln8(1, ptr(a16), ptr(wo16), ptr(lo8), ptr(torch.from_numpy(bo).to(dev)), ptr(x16), ptr(stat),
                                          M, d, K, stream()), "resid_ln8")
    torch.cuda.synchronize()
    want = O.resid_unpack(hi0, lo0).astype(np.float64) + O.round_fp16(A).astype(np.float64) @ O.round_fp16(Wo).astype(np.float64).T + bo
    got_hi, got_lo = x16.float().cpu().numpy(), lo8.cpu().numpy()
    got = O.resid_unpack(got_hi, got_lo).astype(np.float64)
    assert rel_l2(got, want) < 1e-5
    # per element: the block-scaled remainder leaves at most 2^-16 of the largest value of its chunk of four columns, plus the fp32
    # accumulation noise of the sum itself (terms of size ~10 that may cancel)
    noise = 1e-5
    cmax = np.repeat(np.abs(want).reshape(M, d // 4, 4).max(-1), 4, axis=-1).reshape(M, d)
    assert np.all(np.abs(got - want) <= 2.0 ** -16 * cmax + noise)
    assert np.all(np.abs(got_hi - want) <= 2.0 ** -11 * np.abs(want) + noise)          # the 16-bit half alone: an fp16 rounding
    grp = want.reshape(M, d // 64, 64)
    st = stat.cpu().numpy()
    assert np.allclose(st[:, :, 0].T, grp.sum(-1), rtol=1e-5, atol=2e-3)
    assert np.allclose(st[:, :, 1].T, ((grp - grp.mean(-1, keepdims=True)) ** 2).sum(-1), rtol=1e-4)


def test_resid16_8_rows_do_not_depend_on_the_kernel(env):
    """The 16 + 8-bit producer gives BIT-identical rows (both halves and the statistics) from the 256^2 half-stage kernel (one launch)
    and from the 64 x 128 ring (chunks of 1,500 rows), for both residual shapes -- what lets candidates and their captions' cached
    prefixes come from different launches."""
    lib, torch, dev = env
    from leaf_amd import _lib
    M, CH, d = 24000, 1500, 768
    for K in (768, 3072):
        rng = np.random.default_rng(K)
        A = rng.standard_normal((M, K), dtype=np.float32)
        Wo = (rng.standard_normal((d, K), dtype=np.float32) * 0.05).astype(np.float32)
        tbo = torch.from_numpy(rng.standard_normal(d).astype(np.float32)).to(dev)
        hi0, lo0 = O.resid_pack((rng.standard_normal((M, d)) * 2.0).astype(np.float32))
        a16, wo16 = to16(A, "fp16", dev), to16(Wo, "fp16", dev)

        def run(rows):
            x16 = torch.from_numpy(hi0).to(dev).half()
            lo8 = torch.from_numpy(lo0).to(dev)
            stats = []
            for r0 in range(0, M, rows):
                r1 = min(M, r0 + rows)
                st = torch.zeros(d // 64, r1 - r0, 2, dtype=torch.float32, device=dev)
                _lib.check(lib.leaf_op_gemm_resid_ln8(1, ptr(a16[r0:r1]), ptr(wo16), ptr(lo8[r0:r1]), ptr(tbo), ptr(x16[r0:r1]), ptr(st),
                                                      r1 - r0, d, K, stream()), "resid_ln8")
                stats.append(st)
            torch.cuda.synchronize()
            return x16, lo8, torch.cat(stats, 1)

        ref, got = run(M), run(CH)
        for name, a, b in zip(("x16", "lo8", "stat"), ref, got):
            assert torch.equal(a, b), f"K={K}: {name} differs between the two kernels"
