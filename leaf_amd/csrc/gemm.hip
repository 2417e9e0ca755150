// NT GEMM on MFMA for the text tower:  C[M,N] (+)= A[M,K] * B[N,K]^T, 16-bit operands, fp32 accumulate.
// This file: the first (register-staged) kernel, the two-stage 256^2 kernel and the DISPATCH over all GEMM kernels
// (leaf_launch_gemm at the end: gemm256h.hip for >= 128 tiles of 256^2, gemm64.hip for small launches, these otherwise).
//
// Both operands are K-contiguous (nn.Linear weight layout [N,K]; activations [rows,K]), which is the
// natural fragment shape of v_mfma_f32_16x16x32_{f16,bf16}: lane l holds 8 consecutive k of row l&15.
// Tile 128x128x64, 256 threads = 4 waves (2x2, 64x64 each), LDS double buffer (2 x 32 KiB), register-
// staged prefetch of tile t+1 issued before the MFMAs of tile t and written to LDS after them
// (one barrier per K-tile).  LDS rows are 128 B with a 16-B-chunk XOR swizzle (chunk ^ (row&7)) so both
// the ds_write_b128 staging stores and the ds_read_b128 fragment loads are bank-conflict free.
// The MFMA is issued as D = Wfrag * Xfrag^T so the accumulator holds 4 CONSECUTIVE n of one row m:
// epilogues then move 8 B (16-bit) or 16 B (fp32) per lane.  Grid is 1-D with an XCD-aware remap so
// the tiles that share an A panel run on one XCD's L2.
//
// Requirements checked on the host: N % 128 == 0, K % 64 == 0, row strides % 8 == 0; M is arbitrary
// (load rows are clamped, stores masked).
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// MI = 16-row MFMA tiles per wave in M: MI = 4 is the 128 x 128 tile described above (waves 2 x 2 of 64 x 64); MI = 2 is a
// 64 x 128 tile (waves 2 x 2 of 32 x 64, 48 KiB of LDS, up to 3 workgroups per CU) for launches whose 128-row tiling
// would leave CUs idle (the B-caption passes: 3,200 rows x 768 columns = 150 tiles of 128^2 on 256 CUs).
template <class TT, int EPI, int MI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs p) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BMt = 32 * MI;                 // tile rows
    constexpr int ATILE = BMt * BK * 2;          // A tile bytes; the B tile stays TILE_BYTES
    constexpr int BUF = ATILE + TILE_BYTES;      // one stage
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = p.N / BN;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (logical / tiles_n) * BMt, n0 = (logical % tiles_n) * BN;

    // ---- staging assignment: thread -> (row r0 + 32 i, 16-B chunk c)
    const int c = tid & 7, r0 = tid >> 3;
    const u16* __restrict__ A = (const u16*)p.A;
    const u16* __restrict__ B = (const u16*)p.B;
    // rows r0 + 32 i share (row & 7) -> one swizzled LDS offset + i * 4096
    const int soff = lds_off(r0, c);
    auto arow = [&](int i) { int r = m0 + r0 + 32 * i; return r < p.M ? r : p.M - 1; };
    const u16* a_ptr0 = A + (size_t)arow(0) * p.lda + 8 * c;
    const u16* a_ptr1 = A + (size_t)arow(1) * p.lda + 8 * c;
    const u16* a_ptr2 = A + (size_t)arow(MI > 2 ? 2 : 0) * p.lda + 8 * c;
    const u16* a_ptr3 = A + (size_t)arow(MI > 2 ? 3 : 0) * p.lda + 8 * c;
    const u16* b_ptr0 = B + (size_t)(n0 + r0) * p.ldb + 8 * c;
    const u16* b_ptr1 = b_ptr0 + (size_t)32 * p.ldb;
    const u16* b_ptr2 = b_ptr0 + (size_t)64 * p.ldb;
    const u16* b_ptr3 = b_ptr0 + (size_t)96 * p.ldb;
    // staging registers as first-class vector values (arrays of HIP's uint4 struct end up in scratch)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
#define G_LOAD(k0)                                                                                         \
    sa0 = *(const u32x4*)(a_ptr0 + (k0)); sb0 = *(const u32x4*)(b_ptr0 + (k0));                            \
    sa1 = *(const u32x4*)(a_ptr1 + (k0)); sb1 = *(const u32x4*)(b_ptr1 + (k0));                            \
    if constexpr (MI > 2) { sa2 = *(const u32x4*)(a_ptr2 + (k0)); }                                        \
    sb2 = *(const u32x4*)(b_ptr2 + (k0));                                                                  \
    if constexpr (MI > 2) { sa3 = *(const u32x4*)(a_ptr3 + (k0)); }                                        \
    sb3 = *(const u32x4*)(b_ptr3 + (k0));
#define S_STORE(buf)                                                                                       \
    {                                                                                                      \
        char* base_ = smem + (buf) * BUF + soff;                                                           \
        *(u32x4*)(base_) = sa0;          *(u32x4*)(base_ + ATILE) = sb0;                                   \
        *(u32x4*)(base_ + 4096) = sa1;   *(u32x4*)(base_ + ATILE + 4096) = sb1;                            \
        if constexpr (MI > 2) { *(u32x4*)(base_ + 8192) = sa2; }                                           \
        *(u32x4*)(base_ + ATILE + 8192) = sb2;                                                             \
        if constexpr (MI > 2) { *(u32x4*)(base_ + 12288) = sa3; }                                          \
        *(u32x4*)(base_ + ATILE + 12288) = sb3;                                                            \
    }

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fkc = lane >> 4;
    // fragment byte offsets inside a tile (constant over the K loop)
    int xo[2][MI], wo[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MI; ++i) xo[ks][i] = lds_off(wm * (16 * MI) + i * 16 + frow, ks * 4 + fkc);
#pragma unroll
        for (int i = 0; i < 4; ++i) wo[ks][i] = lds_off(wn * 64 + i * 16 + frow, ks * 4 + fkc);
    }
#define COMPUTE(buf)                                                                              \
    {                                                                                             \
        const char* sA_ = smem + (buf) * BUF;                                                     \
        const char* sB_ = sA_ + ATILE;                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                        \
            typename TT::vec8 xa[MI], wb[4];                                                      \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                        \
                xa[i] = *(const typename TT::vec8*)(sA_ + xo[ks][i]);                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
                wb[i] = *(const typename TT::vec8*)(sB_ + wo[ks][i]);                             \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
                    acc[i][j] = TT::mfma(wb[j], xa[i], acc[i][j]);                                \
        }                                                                                         \
    }

    const int nt = p.K / BK;
    G_LOAD(0)
    S_STORE(0)
    __syncthreads();
    // steady state: prefetch tile t+1 into registers, MFMA on tile t, then park the prefetch in the other buffer
    for (int t = 0; t < nt - 1; ++t) {
        const int cur = t & 1;
        G_LOAD((t + 1) * BK)
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch issue ABOVE the MFMAs (hipcc sinks it to its use)
        COMPUTE(cur)
        __builtin_amdgcn_sched_barrier(0);
        S_STORE(cur ^ 1)
        __syncthreads();
    }
    COMPUTE((nt - 1) & 1)
#undef G_LOAD
#undef S_STORE
#undef COMPUTE

    // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + 4*(lane>>4)
    {
        const int nbase = n0 + wn * 64 + 4 * (lane >> 4);
        float4 bias4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = p.bias ? *(const float4*)(p.bias + nbase + 16 * j) : float4{0.f, 0.f, 0.f, 0.f};
        int mrow[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) mrow[i] = m0 + wm * (16 * MI) + i * 16 + frow;
        epilogue_block<TT, EPI, MI, 4>(p, mrow, nbase, bias4, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves (2 x 4, 128 x 64 each), operands streamed HBM/L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4, no staging registers), two 64-KiB stages, ONE raw s_barrier per K-tile:
//     wait own DMA of tile t | barrier | issue DMA of tile t+1 into the other stage | MFMAs on tile t
// so the next tile's transfer has the whole 64-MFMA phase to land.  The LDS image is lane-linear per DMA
// piece (8 rows x 128 B), the XOR swizzle is applied to the per-lane SOURCE chunk and again on the read.
// Halves L2->LDS bytes per FLOP vs the 128^2 kernel (1/128 B/FLOP) and LDS bytes per MFMA (0.375 KiB).
constexpr int BM2 = 256, BN2 = 256;
constexpr int TILE2 = BM2 * BK * 2;   // 32 KiB per operand per stage
constexpr int STAGE2 = 2 * TILE2;     // 64 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// Diagnostic build only (-DLEAF_GEMM_STAMPS): wave 0 stores s_memtime at five points of every block into p.stamps
// (a buffer nothing else reads).  The shipped library is built without it.
#ifdef LEAF_GEMM_STAMPS
#define STAMP(i)                                                                                          \
    if (p.stamps && threadIdx.x == 0) {                                                                   \
        unsigned long long t_;                                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
        ((unsigned long long*)p.stamps)[(size_t)blockIdx.x * 8 + (i)] = t_;                               \
    }
#else
#define STAMP(i)
#endif

template <class TT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(GemmArgs p) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int tiles_n = p.N / BN2;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (logical / tiles_n) * BM2, n0 = (logical % tiles_n) * BN2;

    // ---- DMA sources: wave w moves pieces w*4 + j (8 rows x 128 B each) of the A and of the B tile
    const int prow = lane >> 3;                       // row inside the piece
    const int schunk = (lane & 7) ^ prow;             // source-side swizzle (piece base is a multiple of 8 rows)
    const u16* __restrict__ A = (const u16*)p.A;
    const u16* __restrict__ B = (const u16*)p.B;
    auto arow = [&](int j) { int r = m0 + (wid * 4 + j) * 8 + prow; return r < p.M ? r : p.M - 1; };
    const u16* a0 = A + (size_t)arow(0) * p.lda + schunk * 8;
    const u16* a1 = A + (size_t)arow(1) * p.lda + schunk * 8;
    const u16* a2 = A + (size_t)arow(2) * p.lda + schunk * 8;
    const u16* a3 = A + (size_t)arow(3) * p.lda + schunk * 8;
    const u16* b0 = B + (size_t)(n0 + wid * 32 + prow) * p.ldb + schunk * 8;
    const size_t bstep = (size_t)8 * p.ldb;
    const int piece = wid * 4096;                     // byte offset of this wave's first piece inside a tile
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
#define ISSUE_TILE(stage, k0)                                                        \
    {                                                                                \
        char* sa_ = smem + (stage) * STAGE2 + piece;                                  \
        char* sb_ = sa_ + TILE2;                                                     \
        DMA16(a0 + (k0), sa_);          DMA16(a1 + (k0), sa_ + 1024);                \
        DMA16(a2 + (k0), sa_ + 2048);   DMA16(a3 + (k0), sa_ + 3072);                \
        DMA16(b0 + (k0), sb_);          DMA16(b0 + bstep + (k0), sb_ + 1024);        \
        DMA16(b0 + 2 * bstep + (k0), sb_ + 2048); DMA16(b0 + 3 * bstep + (k0), sb_ + 3072); \
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fkc = lane >> 4;
    // fragment offsets: rows frow + 16 i keep (row & 7) = frow & 7 -> one swizzled base per k-step + i * 2048
    const int fo0 = lds_off(frow, fkc), fo1 = lds_off(frow, 4 + fkc);
    const int xbase = wm * 128 * 128, wbase = TILE2 + wn * 64 * 128;
#define COMPUTE2(stage)                                                                        \
    {                                                                                          \
        const char* st_ = smem + (stage) * STAGE2;                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                     \
            const int fo_ = ks ? fo1 : fo0;                                                    \
            typename TT::vec8 xa[8], wb[4];                                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                      \
                wb[j] = *(const typename TT::vec8*)(st_ + wbase + fo_ + j * 2048);             \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                      \
                xa[i] = *(const typename TT::vec8*)(st_ + xbase + fo_ + i * 2048);             \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                  \
                    acc[i][j] = TT::mfma(wb[j], xa[i], acc[i][j]);                             \
        }                                                                                      \
    }

    const int nt = p.K / BK;
    STAMP(0)
    ISSUE_TILE(0, 0)
    for (int t = 0; t < nt - 1; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0) { STAMP(1) }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ISSUE_TILE((t + 1) & 1, (t + 1) * BK)
        COMPUTE2(t & 1)
    }
    STAMP(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    COMPUTE2((nt - 1) & 1)
#undef DMA16
#undef ISSUE_TILE
#undef COMPUTE2

    STAMP(3)
    {
        const int nbase = n0 + wn * 64 + 4 * (lane >> 4);
        float4 bias4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = p.bias ? *(const float4*)(p.bias + nbase + 16 * j) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two batches of 4 x 4 fragments: 16 loads in flight per lane
            int mrow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) mrow[i] = m0 + wm * 128 + (4 * h + i) * 16 + frow;
            f32x4 (&sub)[4][4] = *reinterpret_cast<f32x4 (*)[4][4]>(&acc[4 * h]);
            epilogue_block<TT, EPI, 4, 4>(p, mrow, nbase, bias4, sub);
        }
    }
    STAMP(4)
}

template <class TT>
hipError_t launch_t(const GemmArgs& p, int epi, hipStream_t s) {
    static int use256 = -1;
    if (use256 < 0) { const char* e = getenv("LEAF_GEMM256"); use256 = (e && e[0] == '0') ? 0 : 1; }
    const bool big = use256 && p.N % BN2 == 0 && (long)((p.M + BM2 - 1) / BM2) * (p.N / BN2) >= 128;
    // 128-row tiles unless they would leave CUs idle (fewer tiles than the chip has CUs): then 64-row tiles.
    // LEAF_GEMM_BM64=0 disables the 64-row variant (A/B runs).
    static int bm64 = -1;
    if (bm64 < 0) { const char* e = getenv("LEAF_GEMM_BM64"); bm64 = (e && e[0] == '0') ? 0 : 1; }
    const long t128 = (long)((p.M + BM - 1) / BM) * (p.N / BN);
    const bool half_m = !big && bm64 && t128 < 256 && p.M > 64;
    const int grid = big ? ((p.M + BM2 - 1) / BM2) * (p.N / BN2)
                         : (half_m ? ((p.M + 63) / 64) * (p.N / BN) : (int)t128);
    const size_t lds = big ? 2 * STAGE2 : (half_m ? 2 * (64 * BK * 2 + TILE_BYTES) : 4 * TILE_BYTES);
#define LEAF_GEMM_CASE(E)                                                                    \
    case E: {                                                                                \
        static bool attr_done = false;                                                       \
        if (!attr_done) {                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<TT, E, 4>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES); \
            (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<TT, E, 2>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES); \
            (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<TT, E>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE2);     \
            attr_done = true;                                                                \
        }                                                                                    \
        if (big) hipLaunchKernelGGL((gemm_nt256_kernel<TT, E>), dim3(grid), dim3(512), lds, s, p); \
        else if (half_m) hipLaunchKernelGGL((gemm_nt_kernel<TT, E, 2>), dim3(grid), dim3(256), lds, s, p); \
        else hipLaunchKernelGGL((gemm_nt_kernel<TT, E, 4>), dim3(grid), dim3(256), lds, s, p); \
        break;                                                                               \
    }
    switch (epi) {
        LEAF_GEMM_CASE(EPI_STORE_T)
        LEAF_GEMM_CASE(EPI_ACT_T)
        LEAF_GEMM_CASE(EPI_RESID_F32)
        LEAF_GEMM_CASE(EPI_STORE_F32)
        LEAF_GEMM_CASE(EPI_ACTGRAD_T)
        LEAF_GEMM_CASE(EPI_LNFOLD_T)
        LEAF_GEMM_CASE(EPI_LNFOLD_ACT_T)
        LEAF_GEMM_CASE(EPI_RESID_LN)
        LEAF_GEMM_CASE(EPI_RESID_LN8)
        default: return hipErrorInvalidValue;
    }
#undef LEAF_GEMM_CASE
    return hipGetLastError();
}

}  // namespace

// small launches: the 64 x 128 LDS-DMA ring kernel (gemm64.hip) unless LEAF_GEMM64=0
static bool use_gemm64(const GemmArgs& p) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("LEAF_GEMM64"); on = (e && e[0] == '0') ? 0 : 1; }
    return on && leaf_gemm64_eligible(p);
}

// which kernel leaf_launch_gemm will pick: 4 = half-stage ring (gemm256h.hip; every launch with >= 128 tiles of 256^2 it can
// take), 1 = gemm_nt256_kernel (two-stage 256^2: the act'(pre) epilogue of the backward), 6 = 64 x 128 ring (gemm64.hip, small
// launches), 0 = gemm_nt_kernel (register-staged; K < 192 and the tiny test config)
int leaf_gemm_family(const GemmArgs& p, int epi) {
    static int use256h = -1;   // LEAF_GEMM256H=0: A/B and test switch (the two-stage kernel then takes the big launches)
    if (use256h < 0) { const char* e = getenv("LEAF_GEMM256H"); use256h = (e && e[0] == '0') ? 0 : 1; }
#ifdef LEAF_VARIANTS         // diagnostic builds only (make variants): the round-3 two-workgroups-per-CU experiment, never in libleaf_hip.so
    static int use_pp = -1;    // LEAF_GEMM_PP=1: the two-workgroups-per-CU 128 x 256 kernel (variants/gemm128pp.hip) takes the big launches
    if (use_pp < 0) { const char* e = getenv("LEAF_GEMM_PP"); use_pp = (e && (e[0] == '1' || e[0] == '2')) ? 1 : 0; }
    // LEAF_GEMM_PP=2: only the HBM-bound residual shape (out_proj: N = K = d), where the ping-pong kernel measured 5-9 % faster alone
    static int pp_mode = -1;
    if (pp_mode < 0) { const char* e = getenv("LEAF_GEMM_PP"); pp_mode = e ? atoi(e) : 0; }
    const bool pp_shape = pp_mode != 2 || (p.N == p.K && (epi == EPI_RESID_LN || epi == EPI_RESID_F32));
    if (use_pp && pp_shape && p.M > 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && leaf_gemm128pp_eligible(p, epi)) return 7;
#endif
    if (use256h && p.M > 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && leaf_gemm256h_eligible(p, epi)) return 4;
    static int use256 = -1;
    if (use256 < 0) { const char* e = getenv("LEAF_GEMM256"); use256 = (e && e[0] == '0') ? 0 : 1; }
    if (use256 && p.N % 256 == 0 && (long)((p.M + 255) / 256) * (p.N / 256) >= 128) return 1;
    return use_gemm64(p) ? 6 : 0;
}

static void* g_stamps = nullptr;
void leaf_gemm_set_stamps(void* p) { g_stamps = p; }
void* leaf_gemm_get_stamps() { return g_stamps; }

hipError_t leaf_launch_gemm(const GemmArgs& p_in, int dtype, int epi, hipStream_t s) {
    GemmArgs p = p_in;
    p.stamps = g_stamps;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % BN || p.K % BK || p.lda % 8 || p.ldb % 8 || p.ldc % 4)
        return hipErrorInvalidValue;
    const int fam = leaf_gemm_family(p, epi);
#ifdef LEAF_VARIANTS
    if (fam == 7) return leaf_launch_gemm128pp(p, dtype, epi, s);
    static int use_w4 = -1;    // LEAF_GEMM_W4=1: the four-wave 128 x 128-per-wave experiment (variants/gemm256w4.hip) where it is eligible
    if (use_w4 < 0) { const char* e = getenv("LEAF_GEMM_W4"); use_w4 = (e && e[0] == '1') ? 1 : 0; }
    if (use_w4 && fam == 4 && leaf_gemm256w4_eligible(p, epi)) return leaf_launch_gemm256w4(p, dtype, epi, s);
#endif
    if (p.a_wrap && fam != 4) return hipErrorInvalidValue;      // only the half-stage ring kernel re-reads A (the caller duplicates it otherwise)
    if (fam == 4) return leaf_launch_gemm256h(p, dtype, epi, s);
    if (fam == 6) return leaf_launch_gemm64(p, dtype, epi, s);
    return dtype == LEAF_F16 ? launch_t<F16>(p, epi, s) : launch_t<BF16>(p, epi, s);
}
