// Weight / bias gradients of the linear layers of one transformer block, ONE grouped launch.
//
//     dW[n][k] += alpha * sum_r dY[r][n] * X[r][k]          db[n] += alpha * sum_r dY[r][n]
//
// (autograd of nn.Linear / in_proj under the TextFARE backward, utils_AT.py:321-337; alpha = 1 / loss scale).  Both
// operands are row-major with the REDUCTION index r (packed token rows) as their row, i.e. a "TN" product.  Instead of
// materialising dY^T and X^T (two transposes per weight, the first implementation) the 32-row operand slabs are staged
// row-major in LDS and read as MFMA fragments with the gfx950 transposing read ds_read_b64_tr_b16.
//
// Tile 128 (n) x 128 (k), 4 waves of 64 x 64, k-step = 32 rows, two slabs in LDS + three in flight in registers, one barrier per step.
// The problems of a block (c_proj, c_fc, out_proj, in_proj) share one launch: blockIdx -> (problem, tile) through a
// small table, so the 432 tiles of a ViT-L block fill the chip where the largest single weight has 144.
// Column sums (bias gradients) are accumulated by the blocks of the first k-tile column while they stage dY and reduced
// through LDS in a fixed order: the whole result is deterministic (no atomics).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TM = 128, TN = 128, KS = 32;
constexpr int LD = 136;                         // LDS row stride (elements): 272 B
constexpr int SLAB = KS * LD * 2;               // 8,704 B
constexpr int LDS_BYTES = 4 * SLAB;             // dY and X slabs, double-buffered: 34,816 B

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's uint4 struct under ?: lands in scratch
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ s16x8 tr8(const char* img, int row0, int col0, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const char* a = img + (row0 + q) * (LD * 2) + (col0 + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * LD * 2));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <class XT, class GT>
__device__ __forceinline__ u32x4 cvt8(u32x4 v) {
    if constexpr (__is_same(XT, GT)) {
        return v;
    } else {
        typename XT::vec8 a = __builtin_bit_cast(typename XT::vec8, v);
        typename GT::vec8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = GT::from_f32(XT::to_f32(a[j]));
        return __builtin_bit_cast(u32x4, b);
    }
}

template <class XT, class GT>
__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // XCD-aware order: blocks that share an XCD (blockIdx % 8) take a contiguous range of tiles, so the tiles that re-read one
    // dY column panel / X column panel hit that XCD's L2 instead of every XCD fetching its own copy
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < a.nprob && bid >= a.p[i].tile0) pi = i;
    const WgradProb& P = a.p[pi];
    const int local = bid - P.tile0;
    const int tn = local / P.tiles_k, tk = local % P.tiles_k;
    const int n0 = tn * TM, k0 = tk * TN;
    const int rows = a.rows;
    const bool do_bias = tk == 0 && P.db != nullptr;

    // staging: thread -> (row = tid >> 4 [+16], 8-column chunk = tid & 15) of each 32 x 128 slab
    const int srow = tid >> 4, sch = tid & 15;
    const u16* yp = P.dY + (size_t)srow * P.ldy + n0 + sch * 8;
    const u16* xp = P.X + (size_t)srow * P.ldx + k0 + sch * 8;
    const size_t ystep = (size_t)16 * P.ldy, xstep = (size_t)16 * P.ldx;
    const int lds_off = srow * (LD * 2) + sch * 16;
    // three slabs in flight in registers (sets 0-2, slab j lives in set j % 3) on top of the two in LDS: with one slab of
    // look-ahead every 32-row step waited a full memory round trip for 16 MFMAs per wave (1.7 k cycles per step: 107 us per
    // ViT-L block at 3,200 rows)
    u32x4 y0[3], y1[3], x0[3], x1[3];
    const u32x4 z4 = u32x4{0u, 0u, 0u, 0u};
#define LOAD_SLAB(S, ks)                                                                                    \
    {                                                                                                       \
        const int r_ = (ks) * KS + srow;                                                                    \
        const size_t oy_ = (size_t)(ks) * KS * P.ldy, ox_ = (size_t)(ks) * KS * P.ldx;                      \
        y0[S] = r_ < rows ? *(const u32x4*)(yp + oy_) : z4;                                                 \
        x0[S] = r_ < rows ? cvt8<XT, GT>(*(const u32x4*)(xp + ox_)) : z4;                                   \
        y1[S] = r_ + 16 < rows ? *(const u32x4*)(yp + oy_ + ystep) : z4;                                    \
        x1[S] = r_ + 16 < rows ? cvt8<XT, GT>(*(const u32x4*)(xp + ox_ + xstep)) : z4;                      \
    }
#define STORE_SLAB(S, buf)                                                                                  \
    {                                                                                                       \
        char* yb_ = smem + (buf) * 2 * SLAB;                                                                \
        *(u32x4*)(yb_ + lds_off) = y0[S];                                                                   \
        *(u32x4*)(yb_ + lds_off + 16 * LD * 2) = y1[S];                                                     \
        *(u32x4*)(yb_ + SLAB + lds_off) = x0[S];                                                            \
        *(u32x4*)(yb_ + SLAB + lds_off + 16 * LD * 2) = x1[S];                                              \
    }
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
#define ADD_COLS(v)                                                                                         \
    {                                                                                                       \
        const typename GT::vec8 e_ = __builtin_bit_cast(typename GT::vec8, v);                              \
        cs[0] += GT::to_f32(e_[0]); cs[1] += GT::to_f32(e_[1]); cs[2] += GT::to_f32(e_[2]);                 \
        cs[3] += GT::to_f32(e_[3]); cs[4] += GT::to_f32(e_[4]); cs[5] += GT::to_f32(e_[5]);                 \
        cs[6] += GT::to_f32(e_[6]); cs[7] += GT::to_f32(e_[7]);                                             \
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wm = wid >> 1, wn = wid & 1;
    const int g = lane >> 4;
    const int nk = (rows + KS - 1) / KS;

    LOAD_SLAB(0, 0)
    LOAD_SLAB(1, 1)       // (slabs past the last row load nothing: zero registers)
    LOAD_SLAB(2, 2)
    if (do_bias) { ADD_COLS(y0[0]) ADD_COLS(y1[0]) }
    STORE_SLAB(0, 0)
    __syncthreads();
    // one 32-row step: request slab ks + 3 into the set slab ks came from, multiply slab ks out of LDS, move slab ks + 1 from
    // its registers into the other LDS buffer
#define STEP(S_CUR, S_NEXT, ks)                                                                             \
    {                                                                                                       \
        const int buf = (ks) & 1;                                                                           \
        LOAD_SLAB(S_CUR, (ks) + 3)                                                                          \
        const char* yb = smem + buf * 2 * SLAB;                                                             \
        const char* xb = yb + SLAB;                                                                         \
        typename GT::vec8 bf[4];                                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
            bf[j] = __builtin_bit_cast(typename GT::vec8, tr8(xb, 8 * g, wn * 64 + 16 * j, lane));          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
            const typename GT::vec8 af = __builtin_bit_cast(typename GT::vec8, tr8(yb, 8 * g, wm * 64 + 16 * i, lane)); \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = GT::mfma(af, bf[j], acc[i][j]);       \
        }                                                                                                   \
        if ((ks) + 1 < nk) {                                                                                \
            if (do_bias) { ADD_COLS(y0[S_NEXT]) ADD_COLS(y1[S_NEXT]) }                                      \
            STORE_SLAB(S_NEXT, buf ^ 1)                                                                     \
        }                                                                                                   \
        __syncthreads();                                                                                    \
    }
    for (int ks = 0; ks < nk; ks += 3) {
        STEP(0, 1, ks)
        if (ks + 1 < nk) STEP(1, 2, ks + 1)
        if (ks + 2 < nk) STEP(2, 0, ks + 2)
    }
#undef STEP
#undef LOAD_SLAB
#undef STORE_SLAB
#undef ADD_COLS
    const float alpha = a.alpha ? *a.alpha : 1.0f;
    // ---- dW tile: lane holds weight rows 16 i + 4 g + e, column 16 j + (lane & 15)
    {
        float* wp = P.dW + (size_t)(n0 + wm * 64 + 4 * g) * P.Kw + k0 + wn * 64 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* rp = wp + (size_t)(16 * i + e) * P.Kw;
#pragma unroll
                for (int j = 0; j < 4; ++j) rp[16 * j] = fmaf(alpha, acc[i][j][e], rp[16 * j]);
            }
    }
    // ---- bias: 16 row-group partials per column, summed in a fixed order
    if (do_bias) {
        float* red = (float*)smem;   // [16][128]; the slabs are dead (last loop iteration ended with a barrier)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[srow * 128 + sch * 8 + j] = cs[j];
        __syncthreads();
        if (tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * 128 + tid];
            P.db[n0 + tid] = fmaf(alpha, s, P.db[n0 + tid]);
        }
    }
}

}  // namespace

// eligibility of one problem for the grouped TN kernel
bool leaf_wgrad_tn_ok(int Nw, int Kw, int ldy, int ldx) { return Nw % TM == 0 && Kw % TN == 0 && ldy % 8 == 0 && ldx % 8 == 0; }

hipError_t leaf_launch_wgrad_group(WgradArgs a, int x_dtype, int g_dtype, hipStream_t s) {
    if (a.nprob < 1 || a.nprob > 4 || a.rows < 1) return hipErrorInvalidValue;
    int tiles = 0;
    for (int i = 0; i < a.nprob; ++i) {
        WgradProb& p = a.p[i];
        if (!leaf_wgrad_tn_ok(p.Nw, p.Kw, p.ldy, p.ldx) || !p.dY || !p.X || !p.dW) return hipErrorInvalidValue;
        p.tile0 = tiles;
        p.tiles_k = p.Kw / TN;
        tiles += (p.Nw / TM) * p.tiles_k;
    }
    const bool xf = x_dtype == LEAF_F16, gf = g_dtype == LEAF_F16;
#define LEAF_WG(XT, GT) hipLaunchKernelGGL((wgrad_tn_kernel<XT, GT>), dim3(tiles), dim3(256), LDS_BYTES, s, a)
    if (xf && gf) LEAF_WG(F16, F16);
    else if (xf) LEAF_WG(F16, BF16);
    else if (gf) LEAF_WG(BF16, F16);
    else LEAF_WG(BF16, BF16);
#undef LEAF_WG
    return hipGetLastError();
}
