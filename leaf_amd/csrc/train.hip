// Training-step kernels (TextFARE loss + backward + AdamW).  These run on B sequences once per outer
// step (about 3/(2*rho*k+4) of the step's FLOPs -- SURVEY.md 8a row a8), so they are written for clarity
// and coalesced access, with the heavy contractions (data/weight gradients) routed through the MFMA
// GEMM in gemm.hip.  Gradient-side 16-bit tensors are either fp16 with a per-step power-of-two loss scale S (default:
// 11 significand bits, the reference's fp16-autocast + GradScaler regime; conversions saturate, never inf) or bf16
// unscaled (S = 1).  gscale = {S, 1/S} lives on the device; parameter gradients are accumulated un-scaled in fp32.
//
// Reference: utils_AT.py:317-337 (loss, backward), train_AT_text_only.py:326-341 (AdamW groups).
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float load_as_f32(const void* p, int kind, size_t i) {
    if (kind == 0) return BF16::to_f32(((const __bf16*)p)[i]);
    if (kind == 1) return F16::to_f32(((const _Float16*)p)[i]);
    return ((const float*)p)[i];
}
__device__ __forceinline__ void store16(void* p, int kind, size_t i, float v) {
    if (kind == 0) ((__bf16*)p)[i] = BF16::from_f32(v);
    else ((_Float16*)p)[i] = F16::from_f32(v);
}
__device__ __forceinline__ uint2 pack4k(int kind, float a, float b, float c, float d) {
    return kind == 0 ? pack4<BF16>(a, b, c, d) : pack4<F16>(a, b, c, d);
}
__device__ __forceinline__ void unpack4k(int kind, uint2 u, float (&o)[4]) {
    if (kind == 0) unpack4<BF16>(u, o); else unpack4<F16>(u, o);
}

// All 4 GEMM weights of all layers in ONE launch: w16_bwd[l][m] = transpose(params[l][m]) in the gradient path's 16-bit
// type.  The fp32 weights of a layer are contiguous in the order qkv [3d,d], out [d,d], fc [4d,d], proj [d,4d] (12 d^2
// floats) and the 16-bit pack uses the same offsets; grid = (32 x 32 tiles of one layer = 12 d^2 / 1024, layers).
__global__ __launch_bounds__(256) void pack_transpose_kernel(const float* __restrict__ src, void* __restrict__ dst,
                                                             int dkind, int d) {
    leaf_fp16_sat_mode();
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int dt = d >> 5;                         // tiles along a d-wide side
    int b = blockIdx.x;
    int rows, cols;
    size_t moff;
    if (b < 3 * dt * dt) { rows = 3 * d; cols = d; moff = 0; }
    else if (b < 4 * dt * dt) { b -= 3 * dt * dt; rows = d; cols = d; moff = (size_t)3 * d * d; }
    else if (b < 8 * dt * dt) { b -= 4 * dt * dt; rows = 4 * d; cols = d; moff = (size_t)4 * d * d; }
    else { b -= 8 * dt * dt; rows = d; cols = 4 * d; moff = (size_t)8 * d * d; }
    const int tc = cols >> 5;
    const int c0 = (b % tc) * 32, r0 = (b / tc) * 32;
    const size_t base = (size_t)blockIdx.y * 12 * d * d + moff;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = src[base + (size_t)(r0 + ty + 8 * i) * cols + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) store16(dst, dkind, base + (size_t)(c0 + ty + 8 * i) * rows + r0 + tx, tile[tx][ty + 8 * i]);
}

__global__ __launch_bounds__(256) void cast16_kernel(const void* __restrict__ src, int kind, void* __restrict__ dst,
                                                     int dkind, size_t n) {
    leaf_fp16_sat_mode();
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) store16(dst, dkind, i, load_as_f32(src, kind, i));
}

// loss = mean_b sum_j (a-f)^2 ; dout = 2 (f-a)/B * scale ; gscale = {S, 1/S}.  Two launches: one block per caption writes
// its dout row and {row sum, row max|dout|} partials, one block reduces the partials in a fixed order.
__global__ __launch_bounds__(256) void fare_rows_kernel(const float* __restrict__ feat, const float* __restrict__ anchor,
                                                        int B, int D, float scale, float* __restrict__ dout,
                                                        float* __restrict__ partial, const float* __restrict__ norms) {
    leaf_fp16_sat_mode();
    __shared__ float red[12];
    const int tid = threadIdx.x, b = blockIdx.x;
    const float k = 2.0f / (float)B * scale;
    constexpr int PER = 8;                      // D <= 2048
    float g[PER], f[PER];
    float s = 0.f, dot = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int j = tid + 256 * u;
        g[u] = 0.f; f[u] = 0.f;
        if (j < D) {
            const size_t i = (size_t)b * D + j;
            f[u] = feat[i];
            const float df = f[u] - anchor[i];
            s = fmaf(df, df, s);
            g[u] = k * df;
            dot = fmaf(f[u], g[u], dot);
        }
    }
    s = wave_sum(s);
    dot = wave_sum(dot);
    if ((tid & 63) == 0) { red[tid >> 6] = s; red[8 + (tid >> 6)] = dot; }
    __syncthreads();
    // --normalize_fare (utils_AT.py:319): feat = f / ||f||; d f = (d feat - feat (feat . d feat)) / ||f||
    const float dotb = (red[8] + red[9]) + (red[10] + red[11]);
    const float inv_n = norms ? 1.0f / fmaxf(norms[b], 1e-12f) : 1.0f;
    float amax = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int j = tid + 256 * u;
        if (j < D) {
            const float go = norms ? (g[u] - f[u] * dotb) * inv_n : g[u];
            amax = fmaxf(amax, fabsf(go));
            if (dout) dout[(size_t)b * D + j] = go;
        }
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = amax;
    __syncthreads();
    if (tid == 0) {
        partial[2 * b] = (red[0] + red[1]) + (red[2] + red[3]);
        partial[2 * b + 1] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    }
}

// in place: x[b] /= max(||x[b]||, 1e-12) and norms[b] = ||x[b]||   (F.normalize of the training features, one wave per row)
__global__ __launch_bounds__(256) void normalize_rows_kernel(float* __restrict__ x, float* __restrict__ norms, int M, int D) {
    leaf_fp16_sat_mode();
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s = fmaf(xr[c], xr[c], s);
    const float n = sqrtf(wave_sum(s));
    const float inv = 1.0f / fmaxf(n, 1e-12f);
    for (int c = lane; c < D; c += 64) xr[c] *= inv;
    if (lane == 0) norms[row] = n;
}

__global__ __launch_bounds__(256) void fare_reduce_kernel(const float* __restrict__ partial, int B, float* __restrict__ loss,
                                                          float* __restrict__ gscale, int use_scaling,
                                                          const float* __restrict__ scaler) {
    leaf_fp16_sat_mode();
    __shared__ float red[8];
    const int tid = threadIdx.x;
    float s = 0.f, amax = 0.f;
    for (int b = tid; b < B; b += 256) { s += partial[2 * b]; amax = fmaxf(amax, partial[2 * b + 1]); }
    s = wave_sum(s);
    amax = wave_max(amax);
    if ((tid & 63) == 0) { red[tid >> 6] = s; red[4 + (tid >> 6)] = amax; }
    __syncthreads();
    if (tid == 0) {
        if (loss) *loss = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
        float S = 1.f;
        const float a = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        if (use_scaling && a > 0.f && a < 3.0e38f) {
            int e;
            frexpf(a, &e);                      // a = m * 2^e, m in [0.5, 1)
            e = 4 - e;                          // bring max|dout| into [8, 16)
            // persistent back-off (GradScaler semantics, leaf_hip.h "gradient scaler"): scaler[LEAF_SC_BACKOFF] <= 0 halvings, raised
            // by every step whose 16-bit gradients saturated, lowered again after a run of clean steps
            if (scaler) e += (int)scaler[LEAF_SC_BACKOFF];
            e = e > 40 ? 40 : (e < -40 ? -40 : e);
            S = ldexpf(1.f, e);
        }
        gscale[0] = S;
        gscale[1] = 1.f / S;
    }
}

// dproj[k][j] += sum_b pooled[b][k] * dout[b][j]      grid = d blocks
__global__ __launch_bounds__(256) void proj_wgrad_kernel(const float* __restrict__ pooled, const float* __restrict__ dout,
                                                         float* __restrict__ dproj, int B, int d, int D) {
    leaf_fp16_sat_mode();
    // column k of `pooled` goes through LDS once, the dout loads of eight captions are in flight together (the first form issued one
    // dependent pair of loads per caption: 107 us for 0.15 GFLOP); same single accumulator per (k, j), captions in ascending order
    __shared__ float pk[256];
    const int k = blockIdx.x;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                 // j = threadIdx.x + 256 u  (D <= 1024 per pass of the outer loop)
    for (int j0 = 0; j0 < D; j0 += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = 0.f;
        for (int b0 = 0; b0 < B; b0 += 256) {
            __syncthreads();
            if (b0 + (int)threadIdx.x < B) pk[threadIdx.x] = pooled[(size_t)(b0 + threadIdx.x) * d + k];
            __syncthreads();
            const int nb = B - b0 < 256 ? B - b0 : 256;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + threadIdx.x + 256 * u;
                if (j >= D) continue;
                const float* dj = dout + (size_t)b0 * D + j;
                float a = acc[u];
                int bb = 0;
                for (; bb + 8 <= nb; bb += 8) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = dj[(size_t)(bb + q) * D];
#pragma unroll
                    for (int q = 0; q < 8; ++q) a = fmaf(pk[bb + q], v[q], a);
                }
                for (; bb < nb; ++bb) a = fmaf(pk[bb], dj[(size_t)bb * D], a);
                acc[u] = a;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + threadIdx.x + 256 * u;
            if (j < D) dproj[(size_t)k * D + j] += acc[u];
        }
    }
}

// ---- backward of projection + ln_final on the pooled (EOT) rows, three small launches (the first version did all of it
// in one block per sequence: 128 blocks x 768 serial wave dot products = 0.85 ms)
// dpooled[b][k] = sum_j dout[b][j] * proj[k][j];   grid (n_seq, d / 32): a wave owns 8 k with 8 independent sums
__global__ __launch_bounds__(256) void dpooled_kernel(const float* __restrict__ dout, const float* __restrict__ proj,
                                                      float* __restrict__ dpooled, int d, int D) {
    leaf_fp16_sat_mode();
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k0 = blockIdx.y * 32 + wid * 8;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int j = lane; j < D; j += 64) {
        const float dj = dout[(size_t)b * D + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = fmaf(dj, proj[(size_t)(k0 + u) * D + j], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float v = wave_sum(acc[u]);
        if (lane == 0) dpooled[(size_t)b * d + k0 + u] = v;
    }
}

constexpr int MAXCH = 8;

// LN_final backward of one pooled row per wave: dx row (carries the loss scale) and the row's {mu, rstd}
__global__ __launch_bounds__(256) void pool_ln_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ x,
                                                          const int32_t* __restrict__ eot_idx, const float* __restrict__ g,
                                                          float eps, float* __restrict__ dx, float* __restrict__ stats,
                                                          const float* __restrict__ gscale, int n_seq, RowMap map, int d) {
    leaf_fp16_sat_mode();
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_seq) return;
    const int nq = d >> 2;
    const size_t row = (size_t)seq_row(map, map.s0 + b) + eot_idx[b];
    const float* xr = x + row * d;
    float4 v[MAXCH], dyv[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nq) { v[i] = *(const float4*)(xr + 4 * c); s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nq) {
            v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nq) {
            const float4 dy = *(const float4*)(dpooled + (size_t)b * d + 4 * c), gg = *(const float4*)(g + 4 * c);
            v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;                 // x_hat
            dyv[i] = float4{dy.x * gg.x, dy.y * gg.y, dy.z * gg.z, dy.w * gg.w};           // d x_hat
            m1 += (dyv[i].x + dyv[i].y) + (dyv[i].z + dyv[i].w);
            m2 += (dyv[i].x * v[i].x + dyv[i].y * v[i].y) + (dyv[i].z * v[i].z + dyv[i].w * v[i].w);
        }
    }
    const float mean1 = wave_sum(m1) / (float)d, mean2 = wave_sum(m2) / (float)d;
    const float sc = rstd * gscale[0];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nq)
            *(float4*)(dx + row * d + 4 * c) =
                float4{sc * (dyv[i].x - mean1 - v[i].x * mean2), sc * (dyv[i].y - mean1 - v[i].y * mean2),
                       sc * (dyv[i].z - mean1 - v[i].z * mean2), sc * (dyv[i].w - mean1 - v[i].w * mean2)};
    }
    if (lane == 0) { stats[2 * b] = mu; stats[2 * b + 1] = rstd; }
}

// dg[c] += sum_b dpooled[b][c] * x_hat[b][c], db[c] += sum_b dpooled[b][c]: one thread per column, fixed order
__global__ __launch_bounds__(256) void pool_ln_wgrad_kernel(const float* __restrict__ dpooled, const float* __restrict__ x,
                                                            const int32_t* __restrict__ eot_idx,
                                                            const float* __restrict__ stats, float* __restrict__ dg,
                                                            float* __restrict__ db, int n_seq, RowMap map, int d) {
    leaf_fp16_sat_mode();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    float sg = 0.f, sb = 0.f;
    // the loads of eight captions are issued together (their addresses hang on per-caption table look-ups: one caption at a time was a
    // chain of dependent round trips, 42 us for 128 captions); the sums still run over the captions in ascending order
    int b = 0;
    for (; b + 8 <= n_seq; b += 8) {
        float dy[8], xv[8], mu[8], rs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const size_t row = (size_t)seq_row(map, map.s0 + b + q) + eot_idx[b + q];
            dy[q] = dpooled[(size_t)(b + q) * d + c];
            xv[q] = x[row * d + c];
            mu[q] = stats[2 * (b + q)];
            rs[q] = stats[2 * (b + q) + 1];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { sg = fmaf(dy[q], (xv[q] - mu[q]) * rs[q], sg); sb += dy[q]; }
    }
    for (; b < n_seq; ++b) {
        const size_t row = (size_t)seq_row(map, map.s0 + b) + eot_idx[b];
        const float dy = dpooled[(size_t)b * d + c];
        sg = fmaf(dy, (x[row * d + c] - stats[2 * b]) * stats[2 * b + 1], sg);
        sb += dy;
    }
    dg[c] += sg;
    db[c] += sb;
}

// LayerNorm backward.  One wave per row, SIXTEEN waves per workgroup (rows blockIdx * NW + wave, grid-stride): at the 3,200
// rows of a training batch every wave owns one row, so x / dy / dx of all rows are in flight at once instead of four rows in
// sequence per wave.  Parameter gradients: every wave accumulates (dy xhat | dy) of its rows in a private [2][d] fp32 image in
// LDS, the workgroup sums the NW images in wave order and writes ONE partial row part[blockIdx][2][d]; a single launch at the
// end of the backward (ln_param_reduce_kernel) adds the partials of every LayerNorm of the tower to dg / db.  No atomics:
// the first version's 310 k same-address global atomics per launch cost 33 of its 47 us, and the result is now deterministic.
// NCH = float4 chunks per lane = ceil(d / 256) (3 for ViT-L, 4 ViT-H, 5 bigG), NW = waves per workgroup: 16 while the row fits
// 128 VGPRs (NCH <= 4), else 8.
template <int NCH, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ g, float eps, float* __restrict__ dx,
                                                         void* __restrict__ dx16, int gkind, float* __restrict__ part,
                                                         int rows, int d) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nq = d >> 2;
    float4* mine = (float4*)smem + (size_t)wid * 2 * nq;   // this wave's [2][d] image: dg | db terms
    float4 gg[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        gg[i] = c < nq ? *(const float4*)(g + 4 * c) : float4{0.f, 0.f, 0.f, 0.f};
        if (part && c < nq) { mine[c] = float4{0.f, 0.f, 0.f, 0.f}; mine[nq + c] = float4{0.f, 0.f, 0.f, 0.f}; }
    }
    for (int row = blockIdx.x * NW + wid; row < rows; row += gridDim.x * NW) {
        const float* xr = x + (size_t)row * d;
        const float* dyr = dy + (size_t)row * d;
        float* dxr = dx + (size_t)row * d;
        float4 xv[NCH], dv[NCH], ov[NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {     // all three streams of the row requested up front
            const int c = lane + 64 * i;
            if (c < nq) {
                xv[i] = *(const float4*)(xr + 4 * c); dv[i] = *(const float4*)(dyr + 4 * c); ov[i] = *(const float4*)(dxr + 4 * c);
                s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
            }
        }
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nq) {
                xv[i].x -= mu; xv[i].y -= mu; xv[i].z -= mu; xv[i].w -= mu;
                q += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nq) {
                xv[i].x *= rstd; xv[i].y *= rstd; xv[i].z *= rstd; xv[i].w *= rstd;   // xhat
                if (part) {   // the image belongs to this wave alone: plain read-modify-write
                    float4 pg = mine[c], pb = mine[nq + c];
                    pg.x = fmaf(dv[i].x, xv[i].x, pg.x); pg.y = fmaf(dv[i].y, xv[i].y, pg.y);
                    pg.z = fmaf(dv[i].z, xv[i].z, pg.z); pg.w = fmaf(dv[i].w, xv[i].w, pg.w);
                    pb.x += dv[i].x; pb.y += dv[i].y; pb.z += dv[i].z; pb.w += dv[i].w;
                    mine[c] = pg; mine[nq + c] = pb;
                }
                dv[i].x *= gg[i].x; dv[i].y *= gg[i].y; dv[i].z *= gg[i].z; dv[i].w *= gg[i].w;  // dxhat
                m1 += (dv[i].x + dv[i].y) + (dv[i].z + dv[i].w);
                m2 += (dv[i].x * xv[i].x + dv[i].y * xv[i].y) + (dv[i].z * xv[i].z + dv[i].w * xv[i].w);
            }
        }
        m1 = wave_sum(m1) / (float)d;
        m2 = wave_sum(m2) / (float)d;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nq) {
                float4 o = ov[i];
                o.x += rstd * (dv[i].x - m1 - xv[i].x * m2); o.y += rstd * (dv[i].y - m1 - xv[i].y * m2);
                o.z += rstd * (dv[i].z - m1 - xv[i].z * m2); o.w += rstd * (dv[i].w - m1 - xv[i].w * m2);
                *(float4*)(dxr + 4 * c) = o;
                if (dx16) *(uint2*)((u16*)dx16 + (size_t)row * d + 4 * c) = pack4k(gkind, o.x, o.y, o.z, o.w);
            }
        }
    }
    if (!part) return;   // input-gradient-only backward (embedding-space PGD): no parameter gradients wanted
    __syncthreads();
    const float4* img = (const float4*)smem;
    float4* dst = (float4*)(part + (size_t)blockIdx.x * 2 * d);
    for (int c = threadIdx.x; c < 2 * nq; c += NW * 64) {
        float4 a = img[c];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float4 t = img[(size_t)w * 2 * nq + c];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        dst[c] = a;
    }
}

// dg[i][c] += inv_s * sum_wg part[i][wg][0][c] (db likewise) for n LayerNorms in one launch: grid (n, 2d / 64), a workgroup
// owns 64 columns, its four waves sum a quarter of the partial rows each (ascending), combined in wave order: deterministic
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(LnReduceArgs a) {
    leaf_fp16_sat_mode();
    __shared__ float red[4][64];
    const int i = blockIdx.x, cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cg;              // column of the [2][d] partial row
    const int per = (a.grid + 3) / 4;
    const int p0 = rg * per, p1 = p0 + per < a.grid ? p0 + per : a.grid;
    const float* src = a.part + (size_t)i * a.grid * 2 * a.d + c;
    float sum = 0.f;
    if (c < 2 * a.d)
        for (int p = p0; p < p1; ++p) sum += src[(size_t)p * 2 * a.d];
    red[rg][cg] = sum;
    __syncthreads();
    if (rg == 0 && c < 2 * a.d) {
        const float t = ((red[0][cg] + red[1][cg]) + red[2][cg]) + red[3][cg];
        float* dst = c < a.d ? a.dg[i] + c : a.db[i] + (c - a.d);
        *dst += t * a.inv_s[0];
    }
}

// dpos[p][:] += sum_n dx[n*ctx+p][:]   (grid = ctx) ; dtok via atomics (grid-stride over rows)
__global__ __launch_bounds__(256) void pos_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ gscale,
                                                      float* __restrict__ dpos, int n_seq, RowMap map, int d) {
    leaf_fp16_sat_mode();
    const int p = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;     // grid (ctx, ceil(d / 256)): one column per thread
    if (c < d) {
        float s = 0.f;
        // eight sequences' loads in flight (each address hangs on two table look-ups); summed in ascending order as before
        int n = 0;
        for (; n + 8 <= n_seq; n += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = p < seq_len(map, map.s0 + n + q) ? dx[((size_t)seq_row(map, map.s0 + n + q) + p) * d + c] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; n < n_seq; ++n)
            if (p < seq_len(map, map.s0 + n)) s += dx[((size_t)seq_row(map, map.s0 + n) + p) * d + c];
        dpos[(size_t)p * d + c] += s * gscale[1];
    }
}
__global__ __launch_bounds__(256) void tok_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ gscale,
                                                      const int32_t* __restrict__ tokens, float* __restrict__ dtok, int rows,
                                                      int n_seq, RowMap map, int d, int vocab) {
    leaf_fp16_sat_mode();
    const int row = blockIdx.x;
    const int sq = seq_of_row(map, row, n_seq);
    int tok = tokens[(size_t)sq * map.ctx + (row - seq_row(map, sq))];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    const float inv = gscale[1];
    for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dtok + (size_t)tok * d + c, dx[(size_t)row * d + c] * inv);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n4,
                                                    size_t n_decay, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float sqrt_bc2, float gscale_host,
                                                    const float* __restrict__ clip_coef) {
    leaf_fp16_sat_mode();
    // clip_coef: optional device scalar from clip_coef_kernel (--grad-clip-norm / the non-finite guard): gradients are
    // multiplied by it; a NEGATIVE value means the gradient norm was inf / NaN and the whole step is skipped (what
    // torch.cuda.amp.GradScaler.step does, train_AT_text_only.py:347, utils_AT.py:339-362)
    if (clip_coef && clip_coef[0] < 0.f) return;
    const float gscale = clip_coef ? gscale_host * clip_coef[0] : gscale_host;
    // with the guard the step counter of the bias corrections lives on the device (clip_coef_kernel): it counts APPLIED steps,
    // a skipped step does not advance it (torch's per-parameter `step` under GradScaler)
    if (clip_coef) { bc1 = clip_coef[LEAF_SC_BC1]; sqrt_bc2 = clip_coef[LEAF_SC_SQRT_BC2]; }
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) {
        float4 pp = ((float4*)p)[i], gg = ((const float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
        float* P = (float*)&pp; float* G = (float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float decay = (4 * i + e) < n_decay ? wd : 0.f;
            const float gr = G[e] * gscale;
            P[e] *= 1.0f - lr * decay;
            M[e] = b1 * M[e] + (1.0f - b1) * gr;
            V[e] = b2 * V[e] + (1.0f - b2) * gr * gr;
            const float denom = sqrtf(V[e]) / sqrt_bc2 + eps;
            P[e] -= (lr / bc1) * (M[e] / denom);
        }
        ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
    }
}

// ---- --grad-clip-norm (utils_AT.py:348-357: torch.nn.utils.clip_grad_norm_(parameters, c, 2.0) before the step)
// partial[b] = sum of squares of this block's grid-stride share of g
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n4, float* __restrict__ partial) {
    leaf_fp16_sat_mode();
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = ((const float4*)g)[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[LEAF_SC_COEF] = clip coefficient min(1, c / (norm + 1e-6)) or -1 (skip), out[LEAF_SC_NORM] = grad_scale * sqrt(sum partial)
// (fixed order), plus the gradient-scaler bookkeeping of leaf_hip.h ("gradient scaler"): one thread, once per optimizer step.
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int nb, float grad_scale,
                                                        float max_norm, float* __restrict__ out, int step_host, float beta1,
                                                        float beta2) {
    leaf_fp16_sat_mode();
    __shared__ double red[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) s += (double)partial[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = grad_scale * (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const float coef = max_norm / (norm + 1e-6f);
        const bool finite = norm == norm && norm < 3.0e38f;
        out[LEAF_SC_NORM] = norm;
        if (!finite) {
            // GradScaler.step / .update (train_AT_text_only.py:347, utils_AT.py:339-362): found inf -> skip the step, halve the scale
            out[LEAF_SC_COEF] = -1.f;                                  // adamw_kernel returns at once
            out[LEAF_SC_SKIPPED] += 1.f;
            if (out[LEAF_SC_SAT_FLAG] != 0.f) out[LEAF_SC_SAT_STEPS] += 1.f;
            out[LEAF_SC_BACKOFF] = fmaxf(out[LEAF_SC_BACKOFF] - 1.f, -30.f);
            out[LEAF_SC_GOOD] = 0.f;
        } else {
            out[LEAF_SC_COEF] = coef < 1.f ? coef : 1.f;
            const float interval = out[LEAF_SC_INTERVAL] > 0.f ? out[LEAF_SC_INTERVAL] : 2000.f;   // GradScaler's growth_interval
            out[LEAF_SC_GOOD] += 1.f;
            if (out[LEAF_SC_GOOD] >= interval) { out[LEAF_SC_BACKOFF] = fminf(out[LEAF_SC_BACKOFF] + 1.f, 0.f); out[LEAF_SC_GOOD] = 0.f; }
            const double applied = (double)step_host - (double)out[LEAF_SC_SKIPPED];      // steps really taken, this one included
            out[LEAF_SC_APPLIED] = (float)applied;
            out[LEAF_SC_BC1] = (float)(1.0 - pow((double)beta1, applied));
            out[LEAF_SC_SQRT_BC2] = sqrtf((float)(1.0 - pow((double)beta2, applied)));
        }
        out[LEAF_SC_SAT_FLAG] = 0.f;
    }
}

// In-place form for gradient accumulation (utils_AT.py:348-357 with --accum-freq > 1: clip_grad_norm_ runs after EVERY micro-batch's
// backward, on the running sum): coef[0] = pre * min(1, c / (pre * ||g|| + 1e-6)), 1 (gradients untouched) when the norm is not
// finite -- the optimizer step's guard then skips the step as it would have without this call.
__global__ __launch_bounds__(256) void clip_inplace_coef_kernel(const float* __restrict__ partial, int nb, float pre, float max_norm,
                                                                float* __restrict__ coef) {
    leaf_fp16_sat_mode();
    __shared__ double red[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) s += (double)partial[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = pre * (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const float c = max_norm / (norm + 1e-6f);
        const bool finite = norm == norm && norm < 3.0e38f;
        coef[0] = finite ? pre * (c < 1.f ? c : 1.f) : 1.f;
        coef[1] = norm;
    }
}
__global__ __launch_bounds__(256) void scale_inplace_kernel(float* __restrict__ g, size_t n4, const float* __restrict__ coef) {
    leaf_fp16_sat_mode();
    const float c = coef[0];
    if (c == 1.f) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ((float4*)g)[i];
        ((float4*)g)[i] = float4{v.x * c, v.y * c, v.z * c, v.w * c};
    }
}

// Saturation check of the 16-bit gradient tensors of one transformer block (fp16 gradient path): a stored value of magnitude
// >= 65504 (0x7BFF: the conversions saturate there; 0x7C00.. = inf / NaN) means the loss scale was too large for this step.
// The reference's GradScaler finds that as an inf in the unscaled gradients; here the kernel raises scaler[LEAF_SC_SAT_FLAG] and
// POISONS gradient element 0 with NaN -- the first element of the token-embedding gradient (token id 0 is the BPE token '!', not
// padding: what matters is that every later writer of that element only adds to it, and NaN + x stays NaN) -- so
// that the non-finite guard of the optimizer step skips the step on THIS rank and, through the gradient all-reduce, on every rank.
struct SatArgs { const uint16_t* buf[5]; unsigned long long n8[5]; };   // element counts in units of 8 (16-byte chunks)
__global__ __launch_bounds__(256) void sat_check16_kernel(SatArgs a, float* __restrict__ scaler, float* __restrict__ poison) {
    leaf_fp16_sat_mode();
    unsigned hit = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const uint4* p = (const uint4*)a.buf[b];
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < a.n8[b]; i += (unsigned long long)gridDim.x * 256) {
            const uint4 v = p[i];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned m = w[e] & 0x7FFF7FFFu;
                hit |= ((m & 0xFFFFu) >= 0x7BFFu) | ((m >> 16) >= 0x7BFFu);
            }
        }
    }
    if (__any(hit != 0) && (threadIdx.x & 63) == 0) {
        scaler[LEAF_SC_SAT_FLAG] = 1.f;
        *poison = __builtin_nanf("");
    }
}

}  // namespace

namespace {
// ---------------------------------------------------------------- optional embedding-space PGD mode
// dst = src * scale[0]  (un-scaling of the loss-scaled gradient stream: d loss / d delta)
__global__ __launch_bounds__(256) void scale_copy_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                         float* __restrict__ dst, size_t n4) {
    leaf_fp16_sat_mode();
    const float sc = scale[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ((const float4*)src)[i];
        ((float4*)dst)[i] = float4{v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    }
}

// The fused grad-sign-project-clip step on the perturbation that stays resident in HBM across the k inner iterations:
//   linf: delta <- clamp(delta + alpha * sign(g), -eps, eps)                (utils_attacks.py:693-694)
//   l2  : delta <- renorm_2(delta + alpha * g / max(||g||_2, 1e-12), eps)   (src/robust_vlm/train/utils.py:96-114),
// norms taken per SEQUENCE over its kept rows x width (positions after EOT carry delta = 0 and a zero gradient).
// One block per sequence; the l2 form makes two passes over the sequence's rows with block reductions in between.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float linf_update(float x, float g, float alpha, float eps) {
    const float sgn = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
    return fminf(fmaxf(x + alpha * sgn, -eps), eps);
}

__global__ __launch_bounds__(256) void pgd_step_kernel(float* __restrict__ delta, const float* __restrict__ grad, RowMap map,
                                                       int d, float alpha, float eps, int norm_l2) {
    leaf_fp16_sat_mode();
    __shared__ float red[4];
    const int sq = map.s0 + blockIdx.x;
    const size_t base = (size_t)seq_row(map, sq) * d;
    const size_t n4 = (size_t)seq_len(map, sq) * d / 4;
    float4* dl = (float4*)(delta + base);
    const float4* g = (const float4*)(grad + base);
    if (!norm_l2) {
        for (size_t i = threadIdx.x; i < n4; i += 256) {
            const float4 a = dl[i], b = g[i];
            dl[i] = float4{linf_update(a.x, b.x, alpha, eps), linf_update(a.y, b.y, alpha, eps),
                           linf_update(a.z, b.z, alpha, eps), linf_update(a.w, b.w, alpha, eps)};
        }
        return;
    }
    float s = 0.f;
    for (size_t i = threadIdx.x; i < n4; i += 256) { const float4 b = g[i]; s += (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w); }
    const float gn = fmaxf(sqrtf(block_sum(s, red)), 1e-12f);
    const float step = alpha / gn;
    float q = 0.f;
    for (size_t i = threadIdx.x; i < n4; i += 256) {
        float4 a = dl[i];
        const float4 b = g[i];
        a = float4{fmaf(step, b.x, a.x), fmaf(step, b.y, a.y), fmaf(step, b.z, a.z), fmaf(step, b.w, a.w)};
        dl[i] = a;
        q += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
    const float dn = sqrtf(block_sum(q, red));
    if (dn > eps) {
        const float f = eps / (dn + 1e-7f);
        for (size_t i = threadIdx.x; i < n4; i += 256) { float4 a = dl[i]; dl[i] = float4{a.x * f, a.y * f, a.z * f, a.w * f}; }
    }
}
}  // namespace

hipError_t leaf_launch_scale_copy(const float* src, const float* scale_dev, float* dst, size_t n, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4, nb = (n4 + 255) / 256;
    hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, s, src, scale_dev, dst, n4);
    return hipGetLastError();
}

hipError_t leaf_launch_pgd_step(float* delta, const float* grad, int n_seq, RowMap map, int d, float alpha, float eps,
                                int norm_l2, hipStream_t s) {
    if (d % 4 || n_seq < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pgd_step_kernel, dim3(n_seq), dim3(256), 0, s, delta, grad, map, d, alpha, eps, norm_l2);
    return hipGetLastError();
}

hipError_t leaf_launch_pack_transpose(const float* src, void* dst, int dst_kind, int d, int layers, hipStream_t s) {
    if (d % 32 || layers < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_transpose_kernel, dim3(12 * (d / 32) * (d / 32), layers), dim3(256), 0, s, src, dst, dst_kind, d);
    return hipGetLastError();
}

hipError_t leaf_launch_cast16(const void* src, int src_kind, void* dst, int dst_kind, size_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(cast16_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, s, src, src_kind, dst,
                       dst_kind, n);
    return hipGetLastError();
}

hipError_t leaf_launch_normalize_rows(float* x, float* norms, int M, int D, hipStream_t s) {
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, norms, M, D);
    return hipGetLastError();
}

hipError_t leaf_launch_fare_loss(const float* feat, const float* anchor, int B, int D, float scale, float* loss,
                                 float* dout, float* gscale, int use_scaling, hipStream_t s, const float* norms,
                                 const float* scaler) {
    if (D > 2048) return hipErrorInvalidValue;
    float* partial = gscale + 64;   // [B][2] right behind the {S, 1/S} slot (carve_bwd reserves it)
    hipLaunchKernelGGL(fare_rows_kernel, dim3(B), dim3(256), 0, s, feat, anchor, B, D, scale, dout, partial, norms);
    hipLaunchKernelGGL(fare_reduce_kernel, dim3(1), dim3(256), 0, s, partial, B, loss, gscale, use_scaling, scaler);
    return hipGetLastError();
}

hipError_t leaf_launch_pool_project_bwd(const float* dout, const float* pooled, const float* x, const int32_t* eot_idx,
                                        const float* g, const float* b, float eps, const float* proj, float* dx,
                                        float* dproj, float* dg, float* db, const float* gscale, int n_seq, RowMap map,
                                        int d, int D, float* scratch /* [n_seq, d + 2] fp32 */, hipStream_t s) {
    (void)b;
    if (d % 32 || d > 256 * MAXCH) return hipErrorInvalidValue;
    float* dpooled = scratch;
    float* stats = scratch + (size_t)n_seq * d;
    if (dproj) hipLaunchKernelGGL(proj_wgrad_kernel, dim3(d), dim3(256), 0, s, pooled, dout, dproj, n_seq, d, D);
    hipLaunchKernelGGL(dpooled_kernel, dim3(n_seq, d / 32), dim3(256), 0, s, dout, proj, dpooled, d, D);
    hipLaunchKernelGGL(pool_ln_bwd_kernel, dim3((n_seq + 3) / 4), dim3(256), 0, s, dpooled, x, eot_idx, g, eps, dx, stats,
                       gscale, n_seq, map, d);
    if (dg) hipLaunchKernelGGL(pool_ln_wgrad_kernel, dim3((d + 255) / 256), dim3(256), 0, s, dpooled, x, eot_idx, stats, dg, db,
                               n_seq, map, d);
    return hipGetLastError();
}

static int ln_bwd_nw(int d) { return (d / 4 + 63) / 64 <= 4 ? 16 : 8; }
int leaf_ln_bwd_grid(int rows, int d) {
    const int nw = ln_bwd_nw(d);
    const int grid = (rows + nw - 1) / nw;
    return grid > 256 ? 256 : grid < 1 ? 1 : grid;
}

hipError_t leaf_launch_layernorm_bwd(const float* dy, const float* x, const float* g, float eps, float* dx_inout,
                                     void* dx16, int gkind, float* part, int rows, int d, hipStream_t s) {
    if (d % 4 || d > 256 * MAXCH) return hipErrorInvalidValue;
    const int nch = (d / 4 + 63) / 64;
    const int grid = leaf_ln_bwd_grid(rows, d);
#define LEAF_LN_BWD(N, NW)                                                                                             \
    {                                                                                                                  \
        const size_t lds = part ? (size_t)NW * 2 * d * sizeof(float) : 0;                                              \
        static bool attr = false;                                                                                      \
        if (!attr) {                                                                                                   \
            (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<N, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                      NW * 2 * 256 * N * (int)sizeof(float));                                          \
            attr = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((ln_bwd_kernel<N, NW>), dim3(grid), dim3(NW * 64), lds, s, dy, x, g, eps, dx_inout, dx16, gkind, \
                           part, rows, d);                                                                             \
    }
    switch (nch) {
        case 1: LEAF_LN_BWD(1, 16) break;
        case 2: LEAF_LN_BWD(2, 16) break;
        case 3: LEAF_LN_BWD(3, 16) break;
        case 4: LEAF_LN_BWD(4, 16) break;
        case 5: LEAF_LN_BWD(5, 8) break;
        default: LEAF_LN_BWD(MAXCH, 8) break;
    }
#undef LEAF_LN_BWD
    return hipGetLastError();
}

hipError_t leaf_launch_ln_param_reduce(const LnReduceArgs& a, hipStream_t s) {
    if (a.n < 1 || a.n > LN_REDUCE_MAX || a.grid < 1 || a.d % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(a.n, (2 * a.d + 63) / 64), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t leaf_launch_attention_bwd(const void* qkv, int qkv_dtype, const void* dout16, void* dqkv16, int gkind,
                                     int n_seq, RowMap map, int heads, int d, hipStream_t s) {
    // the MFMA kernel of attention_bwd.hip (ctx <= 96, no cached prefix: what the training pass runs).  The round-1 fp32 VALU
    // kernel that used to stand behind LEAF_ATTN_BWD=0 is gone (round 4: nothing shipped ever dispatched it)
    if (map.ctx > 96 || map.prefix) return hipErrorInvalidValue;
    return leaf_launch_attention_bwd_mfma(qkv, qkv_dtype, dout16, dqkv16, gkind, n_seq, map, heads, d, s);
}
hipError_t leaf_launch_sat_check16(const void* const* bufs, const size_t* numel, int n, float* scaler, float* poison, hipStream_t s) {
    if (n < 1 || n > 5 || !scaler || !poison) return hipErrorInvalidValue;
    SatArgs a{};
    unsigned long long tot = 0;
    for (int i = 0; i < n; ++i) {
        if (numel[i] % 8 || ((uintptr_t)bufs[i] & 15)) return hipErrorInvalidValue;
        a.buf[i] = (const uint16_t*)bufs[i]; a.n8[i] = numel[i] / 8; tot += a.n8[i];
    }
    const unsigned long long nb = (tot + 255) / 256 / 4 + 1;
    hipLaunchKernelGGL(sat_check16_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, s, a, scaler, poison);
    return hipGetLastError();
}

hipError_t leaf_launch_embed_bwd(const float* dx, const float* gscale, const int32_t* tokens, float* dtok, float* dpos,
                                 int rows, int n_seq, RowMap map, int d, int vocab, hipStream_t s) {
    hipLaunchKernelGGL(pos_bwd_kernel, dim3(map.ctx, (d + 255) / 256), dim3(256), 0, s, dx, gscale, dpos, n_seq, map, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tok_bwd_kernel, dim3(rows), dim3(256), 0, s, dx, gscale, tokens, dtok, rows, n_seq, map, d, vocab);
    return hipGetLastError();
}

hipError_t leaf_launch_clip_inplace(float* g, size_t n, float pre_scale, float max_norm, float* ws, hipStream_t s) {
    if (n % 4 || !ws) return hipErrorInvalidValue;
    const size_t n4 = n / 4, nb = (n4 + 255) / 256;
    const int pb = (int)(nb < 2048 ? nb : 2048);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(pb), dim3(256), 0, s, g, n4, ws + 2);
    hipLaunchKernelGGL(clip_inplace_coef_kernel, dim3(1), dim3(256), 0, s, ws + 2, pb, pre_scale, max_norm, ws);
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, s, g, n4, ws);
    return hipGetLastError();
}

hipError_t leaf_launch_adamw(float* p, const float* g, float* m, float* v, size_t n, size_t n_decay, float lr,
                             float beta1, float beta2, float eps, float wd, int step, float grad_scale, hipStream_t s,
                             float max_norm, float* clip_ws) {
    if (n % 4 || step < 1) return hipErrorInvalidValue;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
    size_t n4 = n / 4, nb = (n4 + 255) / 256;
    const float* coef = nullptr;
    if (max_norm > 0.f) {   // clip_ws: [LEAF_SC_WORDS + 2048] floats: the scaler state of leaf_hip.h, then the partial sums
        if (!clip_ws) return hipErrorInvalidValue;
        const int pb = (int)(nb < 2048 ? nb : 2048);
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3(pb), dim3(256), 0, s, g, n4, clip_ws + LEAF_SC_WORDS);
        hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, s, clip_ws + LEAF_SC_WORDS, pb, grad_scale, max_norm, clip_ws,
                           step, beta1, beta2);
        coef = clip_ws;
    }
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, s, p, g, m, v, n4, n_decay,
                       lr, beta1, beta2, eps, wd, bc1, sqrtf(bc2), grad_scale, coef);
    return hipGetLastError();
}
