// HBM-bound forward kernels of the text path: token gather + positional add + LayerNorm, LayerNorm,
// EOT pooling + final LayerNorm + projection, and the LEAF selection arithmetic (squared-L2 / cosine
// loss per candidate + first-index arg-max over rho + winner gather).
//
// Reference semantics: src/open_clip/model.py:269-284 (encode_text), transformer.py:15-30 (LayerNorm,
// eps 1e-5, fp32 statistics), transformer.py:653-665 (argmax pooling), utils_attacks.py:332-348.
// All rows are handled by ONE wave (64 lanes x float4), so every reduction is a wave shuffle and
// every global access is a coalesced 16-B-per-lane load/store.
#include "common.h"
#include "kernels.h"
#include "lnfold.h"

namespace {

constexpr int MAXCH = 8;  // float4 chunks per lane -> d <= 2048

template <class TT>
__device__ __forceinline__ void ln_row_store(const float4 (&v)[MAXCH], int nq, int lane, const float* __restrict__ g,
                                             const float* __restrict__ b, float eps, int d, u16* __restrict__ xn_row) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        int c = lane + 64 * i;
        if (c < nq) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        int c = lane + 64 * i;
        if (c < nq) {
            float a0 = v[i].x - mu, a1 = v[i].y - mu, a2 = v[i].z - mu, a3 = v[i].w - mu;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        int c = lane + 64 * i;
        if (c < nq) {
            float4 gg = *(const float4*)(g + 4 * c), bb = *(const float4*)(b + 4 * c);
            *(uint2*)(xn_row + 4 * c) = pack4<TT>((v[i].x - mu) * rstd * gg.x + bb.x, (v[i].y - mu) * rstd * gg.y + bb.y,
                                                  (v[i].z - mu) * rstd * gg.z + bb.z, (v[i].w - mu) * rstd * gg.w + bb.w);
        }
    }
}

template <class TT, bool EMBED>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x_in, const int32_t* __restrict__ tokens,
                                                 const float* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                                                 const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                 float* __restrict__ x_out, u16* __restrict__ xn, int rows, int n_seq,
                                                 RowMap map, int d, int vocab) {
    leaf_fp16_sat_mode();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nq = d >> 2;
    float4 v[MAXCH];
    if constexpr (EMBED) {
        const int sq = seq_of_row(map, row, n_seq);
        const int pos = row - seq_row(map, sq) + seq_prefix(map, sq);
        int tok = tokens[(size_t)sq * map.ctx + pos];
        tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
        const float* te = tok_emb + (size_t)tok * d;
        const float* pe = pos_emb + (size_t)pos * d;
        float* xo = x_out + (size_t)row * d;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            int c = lane + 64 * i;
            if (c < nq) {
                float4 a = *(const float4*)(te + 4 * c), p = *(const float4*)(pe + 4 * c);
                if (x_in) {   // EMBED mode: x_in = optional additive embedding perturbation delta [rows, d] (row a12)
                    const float4 dl = *(const float4*)(x_in + (size_t)row * d + 4 * c);
                    a = float4{a.x + dl.x, a.y + dl.y, a.z + dl.z, a.w + dl.w};
                }
                v[i] = float4{a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w};
                *(float4*)(xo + 4 * c) = v[i];
            }
        }
    } else {
        const float* xi = x_in + (size_t)row * d;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            int c = lane + 64 * i;
            if (c < nq) v[i] = *(const float4*)(xi + 4 * c);
        }
    }
    ln_row_store<TT>(v, nq, lane, g, b, eps, d, xn + (size_t)row * d);
}

// Token gather + positional add for the LN-folded forward (lnfold.h): writes the fp32 residual row, its 16-bit copy (the A
// operand of the first QKV GEMM) and the (sum, M2) statistics of each 64-column group.  One wave per row; a lane holds the
// 4-element chunks lane, lane + 64, ...: a group is 16 consecutive lanes of one iteration, reduced in the same tree as the
// GEMM epilogues (row16_sum).
// LO8: the residual stream in 16 + 8 bits (common.h resid_lo4): x_out is the [rows, d] BYTE matrix of remainders, no fp32 row is written
template <class TT, bool LO8>
__global__ __launch_bounds__(256) void embed_fold_kernel(const float* __restrict__ delta, const int32_t* __restrict__ tokens,
                                                         const float* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                                                         float* __restrict__ x_out, u16* __restrict__ x16,
                                                         float2* __restrict__ stat, int stat_ld, int rows, int n_seq, RowMap map,
                                                         int d, int vocab, u16* __restrict__ split3) {
    leaf_fp16_sat_mode();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nq = d >> 2;
    const int sq = seq_of_row(map, row, n_seq);
    const int pos = row - seq_row(map, sq) + seq_prefix(map, sq);
    int tok = tokens[(size_t)sq * map.ctx + pos];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    const float* te = tok_emb + (size_t)tok * d;
    const float* pe = pos_emb + (size_t)pos * d;
    float* xo = x_out + (size_t)row * d;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (64 * i >= nq) break;                       // wave-uniform
        float4 v = float4{0.f, 0.f, 0.f, 0.f};
        if (c < nq) {
            float4 a = *(const float4*)(te + 4 * c), p = *(const float4*)(pe + 4 * c);
            if (delta) {
                const float4 dl = *(const float4*)(delta + (size_t)row * d + 4 * c);
                a = float4{a.x + dl.x, a.y + dl.y, a.z + dl.z, a.w + dl.w};
            }
            v = float4{a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w};
            const uint2 hi = pack4<TT>(v.x, v.y, v.z, v.w);
            if constexpr (LO8) *(unsigned*)((unsigned char*)x_out + (size_t)row * d + 4 * c) = resid_lo4<TT>(v.x, v.y, v.z, v.w, hi);
            else *(float4*)(xo + 4 * c) = v;
            *(uint2*)(x16 + (size_t)row * d + 4 * c) = hi;
            if (split3) {       // [hi | lo | hi] of the exact fp32 row: the A operand of block 0's three-pass QKV GEMM (api.hip, split blocks)
                float hf[4];
                unpack4<TT>(hi, hf);
                const uint2 lo = pack4<TT>(v.x - hf[0], v.y - hf[1], v.z - hf[2], v.w - hf[3]);
                uint2* o = (uint2*)(split3 + (size_t)row * (3 * d)) + c;
                o[0] = hi; o[nq] = lo; o[2 * nq] = hi;
            }
        }
        const float gs = row16_sum(lnfold_sum4(v.x, v.y, v.z, v.w));
        const float gq = row16_sum(lnfold_dev4(v.x, v.y, v.z, v.w, gs * (1.0f / 64.0f)));
        if (c < nq && (lane & 15) == 0) stat[(size_t)(c >> 4) * stat_ld + row] = float2{gs, gq};
    }
}

__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* __restrict__ stat, int ld, int rows, int ngroups, float eps,
                                                          float2* __restrict__ rowstat) {
    leaf_fp16_sat_mode();
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < rows) rowstat[m] = lnfold_row_stat(stat, ld, m, ngroups, eps);
}

// Weight side of the LN folding, one wave per output row of W [N, K = d] (fp32): W'[n,k] = 16-bit(g[k] W[n,k]),
// s[n] = sum_k W'[n,k] (of the ROUNDED values), c[n] = sum_k b[k] W[n,k] + bias[n].  ONE launch for all layers:
// blockIdx.y = 2 * layer + (0: QKV with ln_1, 1: c_fc with ln_2); layers are regularly strided in the flat parameter buffer.
struct FoldPackArgs {
    const float* W[2]; const float* g[2]; const float* b[2]; const float* bias[2];   // layer 0
    size_t w_stride, v_stride;       // floats between consecutive layers (weights / vectors)
    u16* Wp[2]; size_t wp_stride;    // 16-bit outputs of layer 0, elements between layers
    float* s[2]; float* c[2]; size_t aux_stride;
    int N[2], d;
};
template <class TT>
__global__ __launch_bounds__(256) void fold_pack_kernel(FoldPackArgs a) {
    leaf_fp16_sat_mode();
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.y >> 1, w = blockIdx.y & 1;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N[w]) return;
    const int d = a.d;
    const float* wr = a.W[w] + l * a.w_stride + (size_t)n * d;
    const float* g = a.g[w] + l * a.v_stride;
    const float* b = a.b[w] + l * a.v_stride;
    u16* Wp = a.Wp[w] + l * a.wp_stride + (size_t)n * d;
    float ss = 0.f, cc = 0.f;
    for (int c = lane; c < (d >> 2); c += 64) {
        const float4 wv = *(const float4*)(wr + 4 * c), gv = *(const float4*)(g + 4 * c), bv = *(const float4*)(b + 4 * c);
        const uint2 pk = pack4<TT>(gv.x * wv.x, gv.y * wv.y, gv.z * wv.z, gv.w * wv.w);
        *(uint2*)(Wp + 4 * c) = pk;
        float r[4];
        unpack4<TT>(pk, r);
        ss += (r[0] + r[1]) + (r[2] + r[3]);
        cc += (bv.x * wv.x + bv.y * wv.y) + (bv.z * wv.z + bv.w * wv.w);
    }
    ss = wave_sum(ss);
    cc = wave_sum(cc);
    if (lane == 0) {
        a.s[w][l * a.aux_stride + n] = ss;
        a.c[w][l * a.aux_stride + n] = cc + (a.bias[w] + l * a.v_stride)[n];
    }
}

// ---------------------------------------------------------------- pooling + final LN + projection
constexpr int PR = 8;      // sequences per block
constexpr int JJMAX = 8;   // output columns per thread -> D <= 2048

__global__ __launch_bounds__(256) void pool_project_kernel(const float* __restrict__ x, const int32_t* __restrict__ tokens,
                                                           const float* __restrict__ g, const float* __restrict__ b,
                                                           float eps, const float* __restrict__ proj,
                                                           float* __restrict__ out, float* __restrict__ pooled,
                                                           int32_t* __restrict__ eot_idx, int n_seq, RowMap map, int d,
                                                           int D, int normalize, int rows_are_pooled) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;               // [PR][d]
    float* red = xs + PR * d;               // [4][PR]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int s0 = blockIdx.x * PR;
    const int nq = d >> 2;
    // phase 1: each wave pools + normalises two sequences
    for (int rr = wid; rr < PR; rr += 4) {
        const int n = s0 + rr;
        float* xr = xs + rr * d;
        if (n >= n_seq) {
            for (int c = lane; c < d; c += 64) xr[c] = 0.f;
            continue;
        }
        // first index of the maximum token id (torch argmax) among the rows this sequence keeps
        const int sg = map.s0 + n, pfx = seq_prefix(map, sg), ctx = pfx + seq_len(map, sg);
        int bv = -2147483647 - 1, bi = pfx;
        for (int p = pfx + lane; p < ctx; p += 64) {   // only kept positions; EOT (the row maximum) is among them
            int t = tokens[(size_t)sg * map.ctx + p];
            if (t > bv) { bv = t; bi = p; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int ov = __shfl_xor(bv, o, 64), oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (eot_idx && lane == 0) eot_idx[n] = bi;
        // rows_are_pooled: x already holds ONE row per sequence (the EOT row, last-layer trimming)
        const float* xi = rows_are_pooled ? x + (size_t)n * d : x + ((size_t)seq_row(map, sg) + bi - pfx) * d;
        float4 v[MAXCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            int c = lane + 64 * i;
            if (c < nq) { v[i] = *(const float4*)(xi + 4 * c); s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
        }
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            int c = lane + 64 * i;
            if (c < nq) {
                float a0 = v[i].x - mu, a1 = v[i].y - mu, a2 = v[i].z - mu, a3 = v[i].w - mu;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            int c = lane + 64 * i;
            if (c < nq) {
                float4 gg = *(const float4*)(g + 4 * c), bb = *(const float4*)(b + 4 * c);
                float4 y = float4{(v[i].x - mu) * rstd * gg.x + bb.x, (v[i].y - mu) * rstd * gg.y + bb.y,
                                  (v[i].z - mu) * rstd * gg.z + bb.z, (v[i].w - mu) * rstd * gg.w + bb.w};
                *(float4*)(xr + 4 * c) = y;
                if (pooled) *(float4*)(pooled + (size_t)n * d + 4 * c) = y;
            }
        }
    }
    __syncthreads();
    // phase 2: out[r][j] = sum_k xs[r][k] * proj[k][j]; thread owns columns tid + 256*jj.  k is walked in blocks of
    // KU with all KU x JJ projection loads issued before the FMAs (otherwise every k pays one L2 round trip).
    constexpr int KU = 8;
    float acc[JJMAX][PR];
#pragma unroll
    for (int jj = 0; jj < JJMAX; ++jj)
#pragma unroll
        for (int r = 0; r < PR; ++r) acc[jj][r] = 0.f;
    const int njj = (D + 255) >> 8;
    for (int k0 = 0; k0 < d; k0 += KU) {
        float pv[KU][JJMAX];
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int jj = 0; jj < JJMAX; ++jj) {
                const int j = tid + 256 * jj;
                pv[u][jj] = (jj < njj && j < D) ? proj[(size_t)(k0 + u) * D + j] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            float xv[PR];
#pragma unroll
            for (int r = 0; r < PR; ++r) xv[r] = xs[r * d + k0 + u];
#pragma unroll
            for (int jj = 0; jj < JJMAX; ++jj)
                if (jj < njj)
#pragma unroll
                    for (int r = 0; r < PR; ++r) acc[jj][r] = fmaf(xv[r], pv[u][jj], acc[jj][r]);
        }
    }
    if (normalize) {
        float ss[PR];
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < JJMAX; ++jj) s += acc[jj][r] * acc[jj][r];
            ss[r] = wave_sum(s);
        }
        if (lane == 0)
#pragma unroll
            for (int r = 0; r < PR; ++r) red[wid * PR + r] = ss[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            float tot = red[r] + red[PR + r] + red[2 * PR + r] + red[3 * PR + r];
            float inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);   // F.normalize eps
#pragma unroll
            for (int jj = 0; jj < JJMAX; ++jj) acc[jj][r] *= inv;
        }
    }
#pragma unroll
    for (int jj = 0; jj < JJMAX; ++jj) {
        int j = tid + 256 * jj;
        if (j < D)
#pragma unroll
            for (int r = 0; r < PR; ++r)
                if (s0 + r < n_seq) out[(size_t)(s0 + r) * D + j] = acc[jj][r];
    }
}

// ---------------------------------------------------------------- candidate loss + arg-max
__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ feat, const float* __restrict__ anchor,
                                                    int rho, int D, int objective, int32_t* __restrict__ best_idx,
                                                    float* __restrict__ best_feat, float* __restrict__ loss_out) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ls = (float*)smem;  // [rho]
    __shared__ int s_best;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* a = anchor + (size_t)b * D;
    // one wave per candidate, four candidates per wave in flight (their loads are independent: one memory round trip per group instead of
    // one per candidate and per 64 columns -- the launch sits between a stage's last GEMM and the host's read of the winners)
    auto partial = [&](const float* f) {
        float s = 0.f;
        if ((D & 3) == 0) {
            for (int j = 4 * lane; j < D; j += 256) {
                const float4 fv = *(const float4*)(f + j), av = *(const float4*)(a + j);
                if (objective <= 1) {
                    const float t0 = fv.x - av.x, t1 = fv.y - av.y, t2 = fv.z - av.z, t3 = fv.w - av.w;
                    s = fmaf(t0, t0, s); s = fmaf(t1, t1, s); s = fmaf(t2, t2, s); s = fmaf(t3, t3, s);
                } else {
                    s = fmaf(fv.x, av.x, s); s = fmaf(fv.y, av.y, s); s = fmaf(fv.z, av.z, s); s = fmaf(fv.w, av.w, s);
                }
            }
        } else if (objective <= 1) {
            for (int j = lane; j < D; j += 64) { float t = f[j] - a[j]; s = fmaf(t, t, s); }
        } else {
            for (int j = lane; j < D; j += 64) s = fmaf(f[j], a[j], s);
        }
        return s;
    };
    for (int r0 = 4 * wid; r0 < rho; r0 += 16) {
        float s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = r0 + q < rho ? partial(feat + ((size_t)b * rho + r0 + q) * D) : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = wave_sum(s[q]);
            if (objective == 1 || objective == 2) v = -v;  // 0 l2, 1 negl2, 2 dissim, 3 sim
            if (lane == 0 && r0 + q < rho) { ls[r0 + q] = v; if (loss_out) loss_out[(size_t)b * rho + r0 + q] = v; }
        }
    }
    __syncthreads();
    if (wid == 0) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int r = lane; r < rho; r += 64) {
            float v = ls[r];
            if (v > bv || bi == 0x7fffffff) { if (v > bv || bi == 0x7fffffff) { bv = v; bi = r; } }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_best = bi; best_idx[b] = bi; }
    }
    __syncthreads();
    const float* f = feat + ((size_t)b * rho + s_best) * D;
    if (best_feat)
        for (int j = tid; j < D; j += 256) best_feat[(size_t)b * D + j] = f[j];
}

__global__ __launch_bounds__(256) void eot_positions_kernel(const int32_t* __restrict__ tokens, int32_t* __restrict__ eot_pos,
                                                            int n_seq, RowMap map) {
    leaf_fp16_sat_mode();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= n_seq) return;
    const int sg = map.s0 + n, pfx = seq_prefix(map, sg), end = pfx + seq_len(map, sg);
    int bv = -2147483647 - 1, bi = pfx;
    for (int p = pfx + lane; p < end; p += 64) {
        int t = tokens[(size_t)sg * map.ctx + p];
        if (t > bv) { bv = t; bi = p; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int ov = __shfl_xor(bv, o, 64), oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) eot_pos[n] = bi;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ eot_pos,
                                                          float* __restrict__ out, int n_seq, RowMap map, int d) {
    leaf_fp16_sat_mode();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= n_seq) return;
    const int sg = map.s0 + n;
    const float* src = x + ((size_t)seq_row(map, sg) + eot_pos[n] - seq_prefix(map, sg)) * d;
    for (int c = lane; c < (d >> 2); c += 64) *(float4*)(out + (size_t)n * d + 4 * c) = *(const float4*)(src + 4 * c);
}

// the same out of the 16 + 8-bit residual stream: fp32 rows = decode(x16, lo8)
template <class TT>
__global__ __launch_bounds__(256) void gather_rows_lo8_kernel(const u16* __restrict__ x16, const unsigned char* __restrict__ lo8,
                                                              const int32_t* __restrict__ eot_pos, float* __restrict__ out, int n_seq,
                                                              RowMap map, int d) {
    leaf_fp16_sat_mode();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= n_seq) return;
    const int sg = map.s0 + n;
    const size_t r = (size_t)seq_row(map, sg) + eot_pos[n] - seq_prefix(map, sg);
    for (int c = lane; c < (d >> 2); c += 64)
        *(float4*)(out + (size_t)n * d + 4 * c) = resid_decode4<TT>(*(const uint2*)(x16 + r * d + 4 * c), *(const unsigned*)(lo8 + r * d + 4 * c));
}

// the pooled rows of the 16 + 8-bit stream as they are (both halves copied): the last block then runs on them in the same format
__global__ __launch_bounds__(256) void gather_rows_pair_kernel(const u16* __restrict__ x16, const unsigned char* __restrict__ lo8,
                                                               const int32_t* __restrict__ eot_pos, u16* __restrict__ o16,
                                                               unsigned char* __restrict__ o8, int n_seq, RowMap map, int d) {
    leaf_fp16_sat_mode();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= n_seq) return;
    const int sg = map.s0 + n;
    const size_t r = (size_t)seq_row(map, sg) + eot_pos[n] - seq_prefix(map, sg);
    for (int c = lane; c < (d >> 2); c += 64) {
        *(uint2*)(o16 + (size_t)n * d + 4 * c) = *(const uint2*)(x16 + r * d + 4 * c);
        *(unsigned*)(o8 + (size_t)n * d + 4 * c) = *(const unsigned*)(lo8 + r * d + 4 * c);
    }
}

// kernel hooks of the 16 + 8-bit residual format itself (tests): fp32 -> (x16, lo8) and back
template <class TT>
__global__ __launch_bounds__(256) void resid_pack_kernel(const float* __restrict__ x, u16* __restrict__ x16, unsigned char* __restrict__ lo8, size_t n4) {
    leaf_fp16_sat_mode();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = *(const float4*)(x + 4 * i);
        const uint2 hi = pack4<TT>(v.x, v.y, v.z, v.w);
        *(uint2*)(x16 + 4 * i) = hi;
        *(unsigned*)(lo8 + 4 * i) = resid_lo4<TT>(v.x, v.y, v.z, v.w, hi);
    }
}
template <class TT>
__global__ __launch_bounds__(256) void resid_unpack_kernel(const u16* __restrict__ x16, const unsigned char* __restrict__ lo8, float* __restrict__ x, size_t n4) {
    leaf_fp16_sat_mode();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        *(float4*)(x + 4 * i) = resid_decode4<TT>(*(const uint2*)(x16 + 4 * i), *(const unsigned*)(lo8 + 4 * i));
}

template <class TT>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, u16* __restrict__ dst, size_t n4) {
    leaf_fp16_sat_mode();
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) {
        float4 v = *(const float4*)(src + 4 * i);
        *(uint2*)(dst + 4 * i) = pack4<TT>(v.x, v.y, v.z, v.w);
    }
}

}  // namespace

hipError_t leaf_launch_embed_ln(const int32_t* tokens, const float* tok_emb, const float* pos_emb, const float* g,
                                const float* b, float eps, float* x, void* xn, int rows, int n_seq, RowMap map, int d,
                                int vocab, int dtype, hipStream_t s, const float* delta) {
    if (d % 4 || d > 256 * MAXCH) return hipErrorInvalidValue;
    dim3 grid((rows + 3) / 4), blk(256);
    if (dtype == LEAF_F16)
        hipLaunchKernelGGL((ln_kernel<F16, true>), grid, blk, 0, s, delta, tokens, tok_emb, pos_emb, g, b, eps, x,
                           (u16*)xn, rows, n_seq, map, d, vocab);
    else
        hipLaunchKernelGGL((ln_kernel<BF16, true>), grid, blk, 0, s, delta, tokens, tok_emb, pos_emb, g, b, eps, x,
                           (u16*)xn, rows, n_seq, map, d, vocab);
    return hipGetLastError();
}

hipError_t leaf_launch_embed_fold(const int32_t* tokens, const float* tok_emb, const float* pos_emb, float* x, void* x16,
                                  float2* stat, int stat_ld, int rows, int n_seq, RowMap map, int d, int vocab, int dtype,
                                  hipStream_t s, const float* delta, bool lo8, void* split3) {
    if (d % 64 || d > 256 * MAXCH) return hipErrorInvalidValue;
    dim3 grid((rows + 3) / 4), blk(256);
#define LEAF_EF(TT, L8) hipLaunchKernelGGL((embed_fold_kernel<TT, L8>), grid, blk, 0, s, delta, tokens, tok_emb, pos_emb, x, (u16*)x16, stat, stat_ld, rows, n_seq, map, d, vocab, (u16*)split3)
    if (dtype == LEAF_F16) { if (lo8) LEAF_EF(F16, true); else LEAF_EF(F16, false); }
    else { if (lo8) LEAF_EF(BF16, true); else LEAF_EF(BF16, false); }
#undef LEAF_EF
    return hipGetLastError();
}

hipError_t leaf_launch_ln_finalize(const float2* stat, int ld, int rows, int ngroups, float eps, float2* rowstat, hipStream_t s) {
    if (ngroups < 1 || ngroups > LNFOLD_MAXG || rows < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, stat, ld, rows, ngroups, eps, rowstat);
    return hipGetLastError();
}

hipError_t leaf_launch_fold_pack(const float* qkv_w, const float* fc_w, size_t w_stride, const float* ln1_w, const float* ln1_b,
                                 const float* qkv_b, const float* ln2_w, const float* ln2_b, const float* fc_b, size_t v_stride,
                                 void* qkv_p, void* fc_p, size_t wp_stride, float* aux, size_t aux_stride, int d, int layers,
                                 int dtype, hipStream_t s) {
    if (d % 4) return hipErrorInvalidValue;
    FoldPackArgs a;
    a.W[0] = qkv_w; a.W[1] = fc_w; a.g[0] = ln1_w; a.g[1] = ln2_w; a.b[0] = ln1_b; a.b[1] = ln2_b; a.bias[0] = qkv_b; a.bias[1] = fc_b;
    a.w_stride = w_stride; a.v_stride = v_stride;
    a.Wp[0] = (u16*)qkv_p; a.Wp[1] = (u16*)fc_p; a.wp_stride = wp_stride;
    a.s[0] = aux; a.c[0] = aux + 3 * d; a.s[1] = aux + 6 * d; a.c[1] = aux + 10 * d; a.aux_stride = aux_stride;
    a.N[0] = 3 * d; a.N[1] = 4 * d; a.d = d;
    dim3 grid((4 * d + 3) / 4, 2 * layers), blk(256);
    if (dtype == LEAF_F16) hipLaunchKernelGGL((fold_pack_kernel<F16>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((fold_pack_kernel<BF16>), grid, blk, 0, s, a);
    return hipGetLastError();
}

hipError_t leaf_launch_layernorm(const float* x, const float* g, const float* b, float eps, void* xn, int rows, int d,
                                 int dtype, hipStream_t s) {
    if (d % 4 || d > 256 * MAXCH) return hipErrorInvalidValue;
    dim3 grid((rows + 3) / 4), blk(256);
    if (dtype == LEAF_F16)
        hipLaunchKernelGGL((ln_kernel<F16, false>), grid, blk, 0, s, x, nullptr, nullptr, nullptr, g, b, eps, nullptr,
                           (u16*)xn, rows, 0, RowMap{nullptr, 0, 0, 1, nullptr, nullptr, 1}, d, 0);
    else
        hipLaunchKernelGGL((ln_kernel<BF16, false>), grid, blk, 0, s, x, nullptr, nullptr, nullptr, g, b, eps, nullptr,
                           (u16*)xn, rows, 0, RowMap{nullptr, 0, 0, 1, nullptr, nullptr, 1}, d, 0);
    return hipGetLastError();
}

hipError_t leaf_launch_pool_project(const float* x, const int32_t* tokens, const float* g, const float* b, float eps,
                                    const float* proj, float* out, float* pooled, int32_t* eot_idx, int n_seq, RowMap map,
                                    int d, int D, int normalize, hipStream_t s, int rows_are_pooled) {
    if (d % 4 || d > 256 * MAXCH || D > 256 * JJMAX) return hipErrorInvalidValue;
    size_t lds = (size_t)(PR * d + 4 * PR) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)pool_project_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(pool_project_kernel, dim3((n_seq + PR - 1) / PR), dim3(256), lds, s, x, tokens, g, b, eps, proj,
                       out, pooled, eot_idx, n_seq, map, d, D, normalize, rows_are_pooled);
    return hipGetLastError();
}

hipError_t leaf_launch_eot_positions(const int32_t* tokens, int32_t* eot_pos, int n_seq, RowMap map, hipStream_t s) {
    hipLaunchKernelGGL(eot_positions_kernel, dim3((n_seq + 3) / 4), dim3(256), 0, s, tokens, eot_pos, n_seq, map);
    return hipGetLastError();
}

hipError_t leaf_launch_gather_rows(const float* x, const int32_t* eot_pos, float* out, int n_seq, RowMap map, int d,
                                   hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n_seq + 3) / 4), dim3(256), 0, s, x, eot_pos, out, n_seq, map, d);
    return hipGetLastError();
}

hipError_t leaf_launch_gather_rows_pair(const void* x16, const void* lo8, const int32_t* eot_pos, void* o16, void* o8, int n_seq, RowMap map,
                                        int d, hipStream_t s) {
    if (d % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_pair_kernel, dim3((n_seq + 3) / 4), dim3(256), 0, s, (const u16*)x16, (const unsigned char*)lo8, eot_pos,
                       (u16*)o16, (unsigned char*)o8, n_seq, map, d);
    return hipGetLastError();
}

hipError_t leaf_launch_resid_pack(const float* x, void* x16, void* lo8, size_t n, int dtype, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096));
    if (dtype == LEAF_F16) hipLaunchKernelGGL((resid_pack_kernel<F16>), grid, dim3(256), 0, s, x, (u16*)x16, (unsigned char*)lo8, n4);
    else hipLaunchKernelGGL((resid_pack_kernel<BF16>), grid, dim3(256), 0, s, x, (u16*)x16, (unsigned char*)lo8, n4);
    return hipGetLastError();
}
hipError_t leaf_launch_resid_unpack(const void* x16, const void* lo8, float* x, size_t n, int dtype, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096));
    if (dtype == LEAF_F16) hipLaunchKernelGGL((resid_unpack_kernel<F16>), grid, dim3(256), 0, s, (const u16*)x16, (const unsigned char*)lo8, x, n4);
    else hipLaunchKernelGGL((resid_unpack_kernel<BF16>), grid, dim3(256), 0, s, (const u16*)x16, (const unsigned char*)lo8, x, n4);
    return hipGetLastError();
}

hipError_t leaf_launch_gather_rows_lo8(const void* x16, const void* lo8, const int32_t* eot_pos, float* out, int n_seq, RowMap map, int d,
                                       int dtype, hipStream_t s) {
    if (d % 4) return hipErrorInvalidValue;
    if (dtype == LEAF_F16)
        hipLaunchKernelGGL((gather_rows_lo8_kernel<F16>), dim3((n_seq + 3) / 4), dim3(256), 0, s, (const u16*)x16, (const unsigned char*)lo8, eot_pos, out, n_seq, map, d);
    else
        hipLaunchKernelGGL((gather_rows_lo8_kernel<BF16>), dim3((n_seq + 3) / 4), dim3(256), 0, s, (const u16*)x16, (const unsigned char*)lo8, eot_pos, out, n_seq, map, d);
    return hipGetLastError();
}

hipError_t leaf_launch_score(const float* feat, const float* anchor, int B, int rho, int D, int objective,
                             int32_t* best_idx, float* best_feat, float* loss, hipStream_t s) {
    if (rho <= 0 || rho > 8192 || objective < 0 || objective > 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(score_kernel, dim3(B), dim3(256), rho * sizeof(float), s, feat, anchor, rho, D, objective,
                       best_idx, best_feat, loss);
    return hipGetLastError();
}

// device-to-device copy in ONE launch (hipMemcpyAsync splits a 15-MB copy into three dispatches: bulk + two remainders)
namespace {
__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                     const unsigned char* __restrict__ tsrc, unsigned char* __restrict__ tdst, int tail) {
    leaf_fp16_sat_mode();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) tdst[threadIdx.x] = tsrc[threadIdx.x];
}
}  // namespace
hipError_t leaf_launch_copy_bytes(const void* src, void* dst, size_t bytes, hipStream_t s) {
    if (!bytes) return hipSuccess;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
    const size_t n16 = bytes / 16;
    const int tail = (int)(bytes - n16 * 16);
    size_t nb = (n16 + 255) / 256;
    nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, n16,
                       (const unsigned char*)src + n16 * 16, (unsigned char*)dst + n16 * 16, tail);
    return hipGetLastError();
}

// ---- operand splits of the optional higher-precision blocks (leaf_text_split_pack / option "split blocks"):
// a fp32 value v as hi = 16-bit(v), lo = 16-bit(v - hi); hi + lo carries ~22 significand bits.  The GEMM then multiplies
// [x_hi | x_lo | x_hi] by [W_hi | W_hi | W_lo] over a three times longer K: the fp32 accumulator receives x_hi W_hi + x_lo W_hi +
// x_hi W_lo (the missing x_lo W_lo is 2^-22 relative) -- the unchanged GEMM kernels, no new arithmetic.
namespace {
template <class TT>
__global__ __launch_bounds__(256) void split16_rows_kernel(const float* __restrict__ x, u16* __restrict__ out, size_t n4, int d4) {
    leaf_fp16_sat_mode();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t r = i / d4;
        const int c = (int)(i - r * d4);
        const float4 v = ((const float4*)x)[i];
        const uint2 hi = pack4<TT>(v.x, v.y, v.z, v.w);
        float h[4];
        unpack4<TT>(hi, h);
        const uint2 lo = pack4<TT>(v.x - h[0], v.y - h[1], v.z - h[2], v.w - h[3]);
        uint2* o = (uint2*)(out + r * (size_t)(12 * d4)) + c;        // row stride 3 d elements = 12 d4
        o[0] = hi; o[d4] = lo; o[2 * d4] = hi;
    }
}

// the same from the 16 + 8-bit residual stream (common.h resid_lo4): hi = the 16-bit copy as it is, lo = 16-bit(decoded remainder)
template <class TT>
__global__ __launch_bounds__(256) void split16_rows_lo8_kernel(const u16* __restrict__ x16, const unsigned char* __restrict__ lo8,
                                                               u16* __restrict__ out, size_t n4, int d4) {
    leaf_fp16_sat_mode();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t r = i / d4;
        const int c = (int)(i - r * d4);
        const uint2 hi = ((const uint2*)x16)[i];
        const float4 v = resid_decode4<TT>(hi, ((const unsigned*)lo8)[i]);
        float h[4];
        unpack4<TT>(hi, h);
        const uint2 lo = pack4<TT>(v.x - h[0], v.y - h[1], v.z - h[2], v.w - h[3]);
        uint2* o = (uint2*)(out + r * (size_t)(12 * d4)) + c;
        o[0] = hi; o[d4] = lo; o[2 * d4] = hi;
    }
}

// one workgroup per weight row n: W'[n, k] = g[k] * W[n, k] (g == nullptr: W itself) split into hi / lo;
// triple == 1: out row = [hi | hi | lo] (3 K elements) and s[n] = sum_k (hi + lo) in fp32; triple == 0: out row = lo only (K elements);
// triple == 2: out row = [hi | lo] (2 K elements: the B operand of an [A | A] product, GemmArgs::a_wrap)
template <class TT>
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ W, const float* __restrict__ g, u16* __restrict__ out,
                                                         float* __restrict__ srow, int K, int triple) {
    leaf_fp16_sat_mode();
    __shared__ float red[256];
    const int n = blockIdx.x;
    const float* w = W + (size_t)n * K;
    u16* o = out + (size_t)n * (triple == 1 ? 3 * K : triple == 2 ? 2 * K : K);
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = g ? g[k] * w[k] : w[k];
        const typename TT::elem hi = TT::from_f32(v);
        const float hf = TT::to_f32(hi);
        const typename TT::elem lo = TT::from_f32(v - hf);
        const u16 hb = __builtin_bit_cast(u16, hi), lb = __builtin_bit_cast(u16, lo);
        if (triple == 1) { o[k] = hb; o[K + k] = hb; o[2 * K + k] = lb; acc += hf + TT::to_f32(lo); }
        else if (triple == 2) { o[k] = hb; o[K + k] = lb; }
        else o[k] = lb;
    }
    if (triple == 1) {
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) srow[n] = red[0];
    }
}
}  // namespace

hipError_t leaf_launch_split16_rows(const float* x, void* out, int rows, int d, int dtype, hipStream_t s) {
    if (d % 4 || rows < 1) return hipErrorInvalidValue;
    const size_t n4 = (size_t)rows * (d / 4);
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (dtype == LEAF_F16) hipLaunchKernelGGL((split16_rows_kernel<F16>), dim3(grid), dim3(256), 0, s, x, (u16*)out, n4, d / 4);
    else hipLaunchKernelGGL((split16_rows_kernel<BF16>), dim3(grid), dim3(256), 0, s, x, (u16*)out, n4, d / 4);
    return hipGetLastError();
}

// out[r, :] = [x[r, :] | x[r, :]] (16-bit rows, d % 8 == 0): the materialised [A | A] operand of the launches too small for the
// kernel that re-reads A itself (GemmArgs::a_wrap)
namespace {
__global__ __launch_bounds__(256) void dup_cols16_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, size_t n8, int d8) {
    leaf_fp16_sat_mode();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const size_t r = i / d8;
        const int c = (int)(i - r * d8);
        const uint4 v = x[i];
        uint4* o = out + r * (size_t)(2 * d8) + c;
        o[0] = v; o[d8] = v;
    }
}
}  // namespace
hipError_t leaf_launch_dup_cols16(const void* x, void* out, int rows, int d, hipStream_t s) {
    if (d % 8 || rows < 1) return hipErrorInvalidValue;
    const size_t n8 = (size_t)rows * (d / 8);
    const int grid = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    hipLaunchKernelGGL(dup_cols16_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)x, (uint4*)out, n8, d / 8);
    return hipGetLastError();
}

hipError_t leaf_launch_split16_rows_lo8(const void* x16, const void* lo8, void* out, int rows, int d, int dtype, hipStream_t s) {
    if (d % 4 || rows < 1) return hipErrorInvalidValue;
    const size_t n4 = (size_t)rows * (d / 4);
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (dtype == LEAF_F16) hipLaunchKernelGGL((split16_rows_lo8_kernel<F16>), dim3(grid), dim3(256), 0, s, (const u16*)x16, (const unsigned char*)lo8, (u16*)out, n4, d / 4);
    else hipLaunchKernelGGL((split16_rows_lo8_kernel<BF16>), dim3(grid), dim3(256), 0, s, (const u16*)x16, (const unsigned char*)lo8, (u16*)out, n4, d / 4);
    return hipGetLastError();
}

hipError_t leaf_launch_split_pack(const float* W, const float* g, void* out, float* srow, int N, int K, int triple, int dtype, hipStream_t s) {
    if (N < 1 || K < 1) return hipErrorInvalidValue;
    if (dtype == LEAF_F16) hipLaunchKernelGGL((split_pack_kernel<F16>), dim3(N), dim3(256), 0, s, W, g, (u16*)out, srow, K, triple);
    else hipLaunchKernelGGL((split_pack_kernel<BF16>), dim3(N), dim3(256), 0, s, W, g, (u16*)out, srow, K, triple);
    return hipGetLastError();
}

hipError_t leaf_launch_cast(const float* src, void* dst, size_t n, int dtype, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    size_t n4 = n / 4;
    int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    if (grid == 0) return hipSuccess;
    if (dtype == LEAF_F16)
        hipLaunchKernelGGL((cast_kernel<F16>), dim3(grid), dim3(256), 0, s, src, (u16*)dst, n4);
    else
        hipLaunchKernelGGL((cast_kernel<BF16>), dim3(grid), dim3(256), 0, s, src, (u16*)dst, n4);
    return hipGetLastError();
}
