// fp32-grade CLIP.encode_text ("precise" mode, leaf_text_forward_precise; src/open_clip/model.py:269-284, transformer.py:210-265).
//
// The forward-only passes of the search run 16-bit MFMA operands: 9e-4 rel-L2 per embedding row against the
// fp32 reference at random init, more on a trained-like tower.  Embeddings a user exports, eval_textfare's columns and -- optionally --
// the frozen model's anchor pass are B-caption sized and do not need that speed, so this file offers the same function in
// arithmetic that is fp32-grade end to end:
//   * every stored intermediate (residual stream, LayerNorm output, q|k|v, attention output, MLP hidden) is an fp32 row;
//   * every GEMM multiplies the fp32 MASTER weights (no 16-bit pack is read) with fp32 activations through three MFMA passes of
//     fp16 splits made on the fly: v = hi + 2^-11 lo', hi = fp16(v), lo' = fp16(2^11 (v - hi)) (the scaling keeps the remainder out of
//     fp16's subnormals: hi + 2^-11 lo' carries ~22 significand bits); acc = hi*hi + 2^-11 (lo'*hi + hi*lo'), both sums in fp32
//     accumulators, the dropped lo*lo term is 2^-22 relative;
//   * LayerNorm, softmax, the activation and the residual adds are fp32 VALU code (ocml expf / erff, two-pass variance).
// ln_final + text_projection reuse the engine's fp32 matrix-core projection (project.hip).
#include "engine.h"

namespace {

constexpr int PBM = 128, PBN = 64, PBK = 32;     // workgroup tile; 4 waves of 32 x 64
constexpr float LO_UP = 2048.f, LO_DOWN = 1.f / 2048.f;

// byte offset of 16-B chunk c (0..3) of row r inside a [rows][32 halfs] LDS tile: chunk index XOR (row / 4) % 4, so that the 16
// lanes of one ds_read_b128 pass (16 rows, one chunk column) land in 16 distinct bank groups
__device__ __forceinline__ int p_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
    const f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    const f16x4 l = {(_Float16)((v.x - (float)h[0]) * LO_UP), (_Float16)((v.y - (float)h[1]) * LO_UP),
                     (_Float16)((v.z - (float)h[2]) * LO_UP), (_Float16)((v.w - (float)h[3]) * LO_UP)};
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

__device__ __forceinline__ float p_act(float x, int act) {
    if (act == 1) return x / (1.f + expf(-1.702f * x));                 // QuickGELU (transformer.py:33-36)
    return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));         // nn.GELU (erf form)
}

// C[m, n] = epi(sum_k A[m, k] W[n, k]): EPI 0: + bias; 1: act(+ bias); 2: C += (+ bias)   (A, W, C fp32, row-major, ld = K / K / N)
template <int EPI>
__global__ __launch_bounds__(256) void pgemm_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                    const float* __restrict__ bias, int M, int N, int K, int act) {
    leaf_fp16_sat_mode();
    __shared__ __attribute__((aligned(16))) char smem[2][24576];      // per stage: A_hi 8 K | A_lo 8 K | W_hi 4 K | W_lo 4 K
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = N / PBN;
    const int m0 = (blockIdx.x / tiles_n) * PBM, n0 = (blockIdx.x % tiles_n) * PBN;
    // staging: thread -> row (tid / 8) + 32 i, float4 column tid % 8
    const int sr = tid >> 3, sc = tid & 7;
    const float* ap[4];
    const float* wp[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { int r = m0 + sr + 32 * i; r = r < M ? r : M - 1; ap[i] = A + (size_t)r * K + 4 * sc; }
#pragma unroll
    for (int i = 0; i < 2; ++i) wp[i] = W + (size_t)(n0 + sr + 32 * i) * K + 4 * sc;
    float4 ga[4], gw[2];
    auto g_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ga[i] = *(const float4*)(ap[i] + k0);
#pragma unroll
        for (int i = 0; i < 2; ++i) gw[i] = *(const float4*)(wp[i] + k0);
    };
    auto s_store = [&](int buf) {
        char* b = smem[buf];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 hi, lo;
            split4(ga[i], hi, lo);
            const int o = p_off(sr + 32 * i, sc >> 1) + (sc & 1) * 8;
            *(uint2*)(b + o) = hi;
            *(uint2*)(b + 8192 + o) = lo;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint2 hi, lo;
            split4(gw[i], hi, lo);
            const int o = p_off(sr + 32 * i, sc >> 1) + (sc & 1) * 8;
            *(uint2*)(b + 16384 + o) = hi;
            *(uint2*)(b + 20480 + o) = lo;
        }
    };
    f32x4 acc[2][4], cor[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; cor[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int frow = lane & 15, fkc = lane >> 4;
    int ao[2], wo[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) ao[i] = p_off(wid * 32 + i * 16 + frow, fkc);
#pragma unroll
    for (int j = 0; j < 4; ++j) wo[j] = p_off(j * 16 + frow, fkc);
    auto compute = [&](int buf) {
        const char* b = smem[buf];
        f16x8 ah[2], al[2], wh[4], wl[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[i] = *(const f16x8*)(b + ao[i]); al[i] = *(const f16x8*)(b + 8192 + ao[i]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { wh[j] = *(const f16x8*)(b + 16384 + wo[j]); wl[j] = *(const f16x8*)(b + 20480 + wo[j]); }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // W fragment as the MFMA's first operand: the accumulator then holds C^T -- lane l: m = l % 16, n = 4 (l / 16) + e
                acc[i][j] = F16::mfma(wh[j], ah[i], acc[i][j]);
                cor[i][j] = F16::mfma(wh[j], al[i], cor[i][j]);
                cor[i][j] = F16::mfma(wl[j], ah[i], cor[i][j]);
            }
    };
    const int nt = K / PBK;
    g_load(0);
    s_store(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) g_load((t + 1) * PBK);
        compute(cur);
        if (t + 1 < nt) s_store(cur ^ 1);      // the other buffer: its last readers passed the barrier of the previous iteration
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wid * 32 + i * 16 + frow;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + j * 16 + fkc * 4;
            const float4 bv = *(const float4*)(bias + n);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + cor[i][j][e] * LO_DOWN;
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            float4* cp = (float4*)(C + (size_t)m * N + n);
            if (EPI == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = p_act(v[e], act);
            }
            if (EPI == 2) { const float4 c = *cp; v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w; }
            *cp = float4{v[0], v[1], v[2], v[3]};
        }
    }
}

// one wave per row: y = (x - mean) / sqrt(var + eps) * g + b, mean and the variance of the CENTERED values in fp32 (d <= 2048)
__global__ __launch_bounds__(256) void pln_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                  float eps, float* __restrict__ y, int rows, int d) {
    leaf_fp16_sat_mode();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * d;
    float v[32];
    const int n = d >> 6;           // elements per lane (d % 64 == 0)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) if (i < n) { v[i] = xr[lane + 64 * i]; s += v[i]; }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) if (i < n) { const float c = v[i] - mean; q += c * c; }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)d + eps);
    float* yr = y + (size_t)r * d;
#pragma unroll
    for (int i = 0; i < 32; ++i) if (i < n) yr[lane + 64 * i] = (v[i] - mean) * rstd * g[lane + 64 * i] + b[lane + 64 * i];
}

// x[row of (sequence, position)] = token_embedding[token] + positional_embedding[position]; one wave per row
__global__ __launch_bounds__(256) void pembed_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ tok_emb,
                                                     const float* __restrict__ pos_emb, float* __restrict__ x, int rows, int n_seq, RowMap map,
                                                     int d, int vocab) {
    leaf_fp16_sat_mode();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const int sg = seq_of_row(map, r, n_seq), pos = r - seq_row(map, sg);
    int t = tokens[(size_t)sg * map.ctx + pos];
    t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    const float4* te = (const float4*)(tok_emb + (size_t)t * d);
    const float4* pe = (const float4*)(pos_emb + (size_t)pos * d);
    float4* xr = (float4*)(x + (size_t)r * d);
    for (int c = lane; c < (d >> 2); c += 64) {
        const float4 a = te[c], p = pe[c];
        xr[c] = float4{a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w};
    }
}

// causal attention of one (sequence, head) per workgroup, fp32 throughout: K, V of the head in LDS (row stride 65 floats), one
// wave per query row (rows w, w + 4, ...), lane j scores keys j and j + 64 (ctx <= 96), softmax over the wave, lane = output column
constexpr int PCTX = 96;
__global__ __launch_bounds__(256) void pattn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n_seq, RowMap map, int d) {
    leaf_fp16_sat_mode();
    __shared__ float Ks[PCTX * 65], Vs[PCTX * 65], qs[4][64], ps[4][PCTX];
    const int sg = map.s0 + blockIdx.x, head = blockIdx.y;
    const int row0 = seq_row(map, sg), L = seq_len(map, sg);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const size_t ld = 3 * (size_t)d;
    for (int i = tid; i < L * 64; i += 256) {
        const int j = i >> 6, c = i & 63;
        const float* rowp = qkv + (size_t)(row0 + j) * ld + head * 64 + c;
        Ks[j * 65 + c] = rowp[d];
        Vs[j * 65 + c] = rowp[2 * d];
    }
    __syncthreads();
    for (int r = wid; r < L; r += 4) {
        qs[wid][lane] = qkv[(size_t)(row0 + r) * ld + head * 64 + lane] * 0.125f;      // 1 / sqrt(64), exact
        __builtin_amdgcn_wave_barrier();
        float s0 = -INFINITY, s1 = -INFINITY;
        if (lane <= r) {
            float a = 0.f;
#pragma unroll 8
            for (int c = 0; c < 64; ++c) a += qs[wid][c] * Ks[lane * 65 + c];
            s0 = a;
        }
        if (lane + 64 <= r) {
            float a = 0.f;
#pragma unroll 8
            for (int c = 0; c < 64; ++c) a += qs[wid][c] * Ks[(lane + 64) * 65 + c];
            s1 = a;
        }
        float mx = fmaxf(s0, s1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float p0 = lane <= r ? expf(s0 - mx) : 0.f, p1 = lane + 64 <= r ? expf(s1 - mx) : 0.f;
        const float inv = 1.f / wave_sum(p0 + p1);
        ps[wid][lane] = p0 * inv;
        if (lane + 64 < PCTX) ps[wid][lane + 64] = p1 * inv;
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int j = 0; j <= r; ++j) o += ps[wid][j] * Vs[j * 65 + lane];
        out[(size_t)(row0 + r) * d + head * 64 + lane] = o;
        __builtin_amdgcn_wave_barrier();
    }
}

struct PBuf { float *x, *xn, *qkv, *hid, *xg, *scr; int32_t* eot; };

size_t precise_chunk_bytes(const leaf_text* h, size_t rows, size_t cs) {
    const size_t d = h->cfg.width;
    Carver c(nullptr, 0);
    c.take(rows * d * 4); c.take(rows * d * 4); c.take(rows * 3 * d * 4); c.take(rows * 4 * d * 4);
    c.take(cs * d * 4); c.take(cs * d * 4); c.take(cs * 4);
    return align_up(c.off, 256) + 256;
}

int pgemm(int epi, const float* A, const float* W, float* C, const float* bias, int M, int N, int K, int act, hipStream_t s) {
    if (M < 1 || N % PBN || K % PBK) { leaf_set_error("precise gemm: shape %d x %d x %d", M, N, K); return 1; }
    const dim3 grid((unsigned)(((size_t)M + PBM - 1) / PBM * (N / PBN)));
    if (epi == 0) hipLaunchKernelGGL((pgemm_kernel<0>), grid, dim3(256), 0, s, A, W, C, bias, M, N, K, act);
    else if (epi == 1) hipLaunchKernelGGL((pgemm_kernel<1>), grid, dim3(256), 0, s, A, W, C, bias, M, N, K, act);
    else hipLaunchKernelGGL((pgemm_kernel<2>), grid, dim3(256), 0, s, A, W, C, bias, M, N, K, act);
    return leaf_check(hipGetLastError(), "precise gemm");
}

int precise_chunk(const leaf_text* h, const float* P, const int32_t* tokens, int cs, int rows, RowMap map, float* out, int normalize,
                  const PBuf& b, hipStream_t s) {
    const leaf_text_cfg& c = h->cfg;
    const int d = c.width;
    const dim3 rgrid((rows + 3) / 4);
    hipLaunchKernelGGL(pembed_kernel, rgrid, dim3(256), 0, s, tokens, P + h->tok_emb, P + h->pos_emb, b.x, rows, cs, map, d, c.vocab_size);
    LEAF_TRY(hipGetLastError());
    for (int l = 0; l < c.layers; ++l) {
        const LayerOff& o = h->layer[l];
        hipLaunchKernelGGL(pln_kernel, rgrid, dim3(256), 0, s, b.x, P + o.ln1_w, P + o.ln1_b, c.ln_eps, b.xn, rows, d);
        if (pgemm(0, b.xn, P + o.qkv_w, b.qkv, P + o.qkv_b, rows, 3 * d, d, 0, s)) return 1;
        hipLaunchKernelGGL(pattn_kernel, dim3(cs, c.heads), dim3(256), 0, s, b.qkv, b.xn, cs, map, d);
        if (pgemm(2, b.xn, P + o.out_w, b.x, P + o.out_b, rows, d, d, 0, s)) return 1;
        hipLaunchKernelGGL(pln_kernel, rgrid, dim3(256), 0, s, b.x, P + o.ln2_w, P + o.ln2_b, c.ln_eps, b.xn, rows, d);
        if (pgemm(1, b.xn, P + o.fc_w, b.hid, P + o.fc_b, rows, 4 * d, d, c.activation, s)) return 1;
        if (pgemm(2, b.hid, P + o.proj_w, b.x, P + o.proj_b, rows, d, 4 * d, 0, s)) return 1;
    }
    LEAF_TRY(hipGetLastError());
    if (!leaf_project_rows_ok(d, c.embed_dim)) {       // widths the fp32 matrix-core projection does not take (the d = 128 test config): fp32 VALU
        LEAF_TRY(leaf_launch_pool_project(b.x, tokens, P + h->lnf_w, P + h->lnf_b, c.ln_eps, P + h->text_proj, out, nullptr, nullptr, cs, map,
                                          d, c.embed_dim, normalize, s));
        return 0;
    }
    LEAF_TRY(leaf_launch_eot_positions(tokens, b.eot, cs, map, s));
    LEAF_TRY(leaf_launch_gather_rows(b.x, b.eot, b.xg, cs, map, d, s));
    LEAF_TRY(leaf_launch_project_rows(b.xg, P + h->lnf_w, P + h->lnf_b, c.ln_eps, P + h->text_proj, b.scr, out, cs, d, c.embed_dim, normalize, s));
    return 0;
}

constexpr int PRECISE_CHUNK_SEQS = 512;      // 512 x 77 rows x 36 d bytes: 1.1 GB of workspace at d = 768

}  // namespace

extern "C" size_t leaf_text_precise_workspace_bytes(leaf_text_t h, int n_seq) {
    if (!h || n_seq < 1) return 0;
    const size_t cs = n_seq < PRECISE_CHUNK_SEQS ? n_seq : PRECISE_CHUNK_SEQS;
    return precise_chunk_bytes(h, cs * h->cfg.context_length, cs);
}

extern "C" int leaf_text_forward_precise(leaf_text_t h, const float* params, const int32_t* tokens, const int32_t* seq_lens,
                                         const int32_t* cu_rows, int n_seq, float* out, int normalize, void* ws, size_t ws_bytes,
                                         leaf_stream_t s_) {
    if (!h || !params || !tokens || !out || !ws || n_seq < 1) { leaf_set_error("null/invalid argument"); return 1; }
    if ((seq_lens == nullptr) != (cu_rows == nullptr)) { leaf_set_error("seq_lens (host) and cu_rows (device) go together"); return 1; }
    const leaf_text_cfg& c = h->cfg;
    const int ctx = c.context_length, d = c.width;
    if (ctx > PCTX || d % 64 || d > 2048) {
        leaf_set_error("precise forward: unsupported shape (ctx <= 96, width %% 64 == 0, width <= 2048)");
        return 1;
    }
    hipStream_t s = (hipStream_t)s_;
    const size_t cs_max = n_seq < PRECISE_CHUNK_SEQS ? n_seq : PRECISE_CHUNK_SEQS, budget = cs_max * ctx;
    Carver cv(ws, ws_bytes);
    PBuf b;
    b.x = (float*)cv.take(budget * d * 4); b.xn = (float*)cv.take(budget * d * 4);
    b.qkv = (float*)cv.take(budget * 3 * d * 4); b.hid = (float*)cv.take(budget * 4 * d * 4);
    b.xg = (float*)cv.take(cs_max * d * 4); b.scr = (float*)cv.take(cs_max * d * 4); b.eot = (int32_t*)cv.take(cs_max * 4);
    if (!cv.ok()) { leaf_set_error("precise forward: workspace too small: need %zu bytes, have %zu", cv.off, cv.cap); return 1; }
    int s0 = 0;
    size_t row0 = 0;
    while (s0 < n_seq) {
        int s1 = s0;
        size_t rows = 0;
        while (s1 < n_seq && s1 - s0 < (int)cs_max) {
            const int L = seq_lens ? seq_lens[s1] : ctx;
            if (L < 1 || L > ctx) { leaf_set_error("seq_lens[%d] = %d out of range 1..%d", s1, L, ctx); return 1; }
            if (rows + L > budget) break;
            rows += L;
            ++s1;
        }
        RowMap map{cu_rows, s0, (int)row0, ctx, nullptr, nullptr, 1, 0};
        if (precise_chunk(h, params, tokens, s1 - s0, (int)rows, map, out + (size_t)s0 * c.embed_dim, normalize, b, s)) return 1;
        s0 = s1;
        row0 += rows;
    }
    return 0;
}
