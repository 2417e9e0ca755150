// Final LayerNorm + text projection of the POOLED rows (one row per sequence), exact fp32 on the matrix cores.
//
// Reference: x = ln_final(x); x = text_global_pool(x, text); x = x @ text_projection  (src/open_clip/model.py:276-282,
// transformer.py:653-665,727).  Only the pooled (EOT) row of each sequence reaches the output, so LayerNorm runs on
// those rows alone and the projection is an [M, d] x [d, D] fp32 GEMM with M = sequences (6400 per search stage).
// It runs on v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulate: no operand rounding, so the feature error budget
// of the 16-bit layer GEMMs is not touched) instead of the VALU dot products of pool_project_kernel, which re-read the
// whole projection matrix from L2 once per 8 sequences.
//
//   ln_rows_f32_kernel    one wave per row: two-pass mean / variance in registers, y = (x - mu) * rstd * g + b
//   proj_f32_kernel       tile 64 (rows) x 128 (cols), 4 waves of 32 x 64, k-step 16, register-staged double buffer
//   l2norm_rows_kernel    F.normalize(dim=-1) for the 'sim' / 'dissim' objectives
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAXCH = 8;   // float4 chunks per lane -> d <= 2048

__global__ __launch_bounds__(256) void ln_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ b, float eps, float* __restrict__ y,
                                                          int M, int d) {
    leaf_fp16_sat_mode();
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nq = d >> 2;
    const float* xi = x + (size_t)row * d;
    float4 v[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nq) { v[i] = *(const float4*)(xi + 4 * c); s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nq) {
            const float a0 = v[i].x - mu, a1 = v[i].y - mu, a2 = v[i].z - mu, a3 = v[i].w - mu;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nq) {
            const float4 gg = *(const float4*)(g + 4 * c), bb = *(const float4*)(b + 4 * c);
            *(float4*)(y + (size_t)row * d + 4 * c) =
                float4{(v[i].x - mu) * rstd * gg.x + bb.x, (v[i].y - mu) * rstd * gg.y + bb.y,
                       (v[i].z - mu) * rstd * gg.z + bb.z, (v[i].w - mu) * rstd * gg.w + bb.w};
        }
    }
}

constexpr int BM = 64, BN = 128, BK = 16;
constexpr int A_LD = BK + 1;      // floats; +1: the 16 rows a fragment read touches land on 16 different banks
constexpr int B_LD = BN + 16;     // floats; 144 % 32 = 16: the two k rows a 32-lane half reads use disjoint banks
constexpr int A_TILE = BM * A_LD, B_TILE = BK * B_LD;
typedef float f32x4v __attribute__((ext_vector_type(4)));

// C[M,N] = A[M,K] B[K,N], all fp32 row-major; N % 128 == 0, K % 16 == 0, rows >= M masked
__global__ __launch_bounds__(256) void proj_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
    leaf_fp16_sat_mode();
    __shared__ float As[2][A_TILE];
    __shared__ float Bs[2][B_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    // staging: A tile 64 x 16 floats = 256 float4 (one per thread); B tile 16 x 128 floats = 512 float4 (two per thread)
    const int ar = tid >> 2, ac = (tid & 3) * 4;
    const int br = tid >> 5, bc = (tid & 31) * 4;
    const bool arow_ok = m0 + ar < M;
    const float* ap = A + (size_t)(arow_ok ? m0 + ar : 0) * K + ac;
    const float* bp = B + (size_t)br * N + n0 + bc;
    f32x4v a4, b4a, b4b;
    const f32x4v z4 = f32x4v{0.f, 0.f, 0.f, 0.f};
#define LOAD_T(k0)                                                                       \
    a4 = arow_ok ? *(const f32x4v*)(ap + (k0)) : z4;                                     \
    b4a = *(const f32x4v*)(bp + (size_t)(k0) * N);                                       \
    b4b = *(const f32x4v*)(bp + (size_t)((k0) + 8) * N);
#define STORE_T(buf)                                                                     \
    As[buf][ar * A_LD + ac] = a4[0]; As[buf][ar * A_LD + ac + 1] = a4[1];                \
    As[buf][ar * A_LD + ac + 2] = a4[2]; As[buf][ar * A_LD + ac + 3] = a4[3];            \
    *(f32x4v*)(&Bs[buf][br * B_LD + bc]) = b4a;                                          \
    *(f32x4v*)(&Bs[buf][(br + 8) * B_LD + bc]) = b4b;
    const int wm = wid >> 1, wn = wid & 1;           // wave tile 32 x 64
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK;
    LOAD_T(0)
    STORE_T(0)
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) { LOAD_T((ks + 1) * BK) }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            // A fragment: lane holds A[m = r16][k = kk + g]; B fragment: B[k = kk + g][n = r16]
            float af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[buf][(wm * 32 + 16 * i + r16) * A_LD + kk + g];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = Bs[buf][(kk + g) * B_LD + wn * 64 + 16 * j + r16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) { STORE_T(buf ^ 1) }
        __syncthreads();
    }
#undef LOAD_T
#undef STORE_T
    // C layout: lane holds rows 4 g + e, column r16 of each 16 x 16 tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * 32 + 16 * i + 4 * g + e;
            if (m < M) {
                float* cp = C + (size_t)m * N + n0 + wn * 64 + r16;
#pragma unroll
                for (int j = 0; j < 4; ++j) cp[16 * j] = acc[i][j][e];
            }
        }
}

__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, int M, int D) {
    leaf_fp16_sat_mode();
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);   // F.normalize eps
    for (int c = lane; c < D; c += 64) xr[c] *= inv;
}

}  // namespace

bool leaf_project_rows_ok(int d, int D) { return D % BN == 0 && d % BK == 0 && d % 4 == 0 && d <= 4 * 64 * MAXCH; }

// out[M, D] = [normalize](LN(xg[M, d]; g, b) @ proj[d, D]); xn = [M, d] fp32 scratch
hipError_t leaf_launch_project_rows(const float* xg, const float* g, const float* b, float eps, const float* proj,
                                    float* xn, float* out, int M, int d, int D, int normalize, hipStream_t s) {
    if (!leaf_project_rows_ok(d, D) || M < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_rows_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, s, xg, g, b, eps, xn, M, d);
    hipLaunchKernelGGL(proj_f32_kernel, dim3(((M + BM - 1) / BM) * (D / BN)), dim3(256), 0, s, xn, proj, out, M, D, d);
    if (normalize) hipLaunchKernelGGL(l2norm_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, out, M, D);
    return hipGetLastError();
}
