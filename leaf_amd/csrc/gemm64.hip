// NT GEMM for SMALL launches (the B-caption passes: anchor, clean-caption K/V cache, training forward, data gradients;
// M = 3,200 rows at B = 128): (32 MI) x 128 output tile, MI = 2 | 3 | 4, K streamed in 64-deep stages through a 3-slot
// LDS-DMA ring.  These launches are bound by the L2 -> LDS fill rate of the CUs that hold the most tiles, so the host
// picks the MI that minimises  max-tiles-per-CU x stage bytes  (3,234 x 768: 96-row tiles = 204 tiles, one per CU).
//
// The register-staged kernel of gemm.hip pays one exposed global round trip per 64-k tile (prefetch distance 1) and a
// 3,200 x 768 x 3,072 product took ~40 us on it whatever the tile count.  Here the loads are LDS-DMA pieces of
// 8 rows x 128 B (whole lines, source-side XOR swizzle, as in gemm256h.hip) requested TWO stages ahead, a stage is
// 24 KiB at MI = 2 (A 64 rows, B 128 rows; three stages = 72 KiB, two workgroups per CU), and one barrier per stage.
//     wait own DMAs of stage t | barrier | request stage t+2 into the slot stage t-1 just left | MFMAs on stage t
// 4 waves (2 x 2) of (16 MI) x 64: acc[MI][4], the k order and the epilogue arithmetic of every other GEMM kernel (bit-identical
// rows).  Requirements (host-checked): N % 128 == 0, K % 64 == 0, K >= 192, row strides % 8 == 0; M arbitrary.
#include "gemm_epilogue.h"

namespace {

constexpr int BN = 128, BK = 64;
constexpr int B_BYTES = BN * BK * 2;      // 16 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ int lds_off_h(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N) : "memory");
}

// NS = ring slots: NS - 1 stages are in flight ahead of the MFMAs (3 slots -> two workgroups per CU; 5-6 slots -> one
// workgroup per CU with a DMA latency of 4-5 stage times hidden, the better choice when a CU holds about one tile)
template <class TT, int EPI, int MI, int NS>
__global__ __launch_bounds__(256, 2) void gemm_nt64_ring_kernel(GemmArgs p) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 32 * MI;
    constexpr int A_BYTES = BM * BK * 2;      // 8 / 12 / 16 KiB
    constexpr int STAGE = A_BYTES + B_BYTES;  // 24 / 28 / 32 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = p.N / BN;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;

    // ---- DMA sources: a piece is 8 rows x 128 B; wave w moves A pieces MI w .. MI w + MI - 1 and B pieces 4w .. 4w+3
    const int prow = lane >> 3;
    const int schunk = (lane & 7) ^ prow;
    const char* __restrict__ A = (const char*)p.A;
    const char* __restrict__ B = (const char*)p.B;
    auto arow = [&](int j) { int r = m0 + wid * (8 * MI) + 8 * j + prow; return r < p.M ? r : p.M - 1; };
    const unsigned a0 = (unsigned)arow(0) * (unsigned)p.lda * 2u + schunk * 16;
    const unsigned a1 = (unsigned)arow(1) * (unsigned)p.lda * 2u + schunk * 16;
    const unsigned a2 = (unsigned)arow(MI > 2 ? 2 : 0) * (unsigned)p.lda * 2u + schunk * 16;
    const unsigned a3 = (unsigned)arow(MI > 3 ? 3 : 0) * (unsigned)p.lda * 2u + schunk * 16;
    const unsigned b0 = (unsigned)(n0 + wid * 32 + prow) * (unsigned)p.ldb * 2u + schunk * 16;
    const unsigned bstep = 16u * (unsigned)p.ldb;   // 8 rows, bytes
    const int apiece = wid * (1024 * MI), bpiece = A_BYTES + wid * 4096;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
    // the MI + 4 pieces of K stage `kt` into ring slot `slot`, in four groups (G = 0..3: two B pieces | A pieces 0-1 | the other two
    // B pieces | A pieces 2-3 where the tile has them)
#define ISSUE_GROUP(slot, kt, G)                                                                             \
    {                                                                                                        \
        char* st_ = smem + (slot) * STAGE;                                                                   \
        const size_t ko_ = (size_t)(kt) * (BK * 2);                                                          \
        if ((G) == 0) { DMA16(B + ko_ + b0, st_ + bpiece); DMA16(B + ko_ + bstep + b0, st_ + bpiece + 1024); } \
        if ((G) == 1) { DMA16(A + ko_ + a0, st_ + apiece); DMA16(A + ko_ + a1, st_ + apiece + 1024); }       \
        if ((G) == 2) { DMA16(B + ko_ + 2 * (size_t)bstep + b0, st_ + bpiece + 2048); DMA16(B + ko_ + 3 * (size_t)bstep + b0, st_ + bpiece + 3072); } \
        if ((G) == 3) {                                                                                      \
            if constexpr (MI > 2) DMA16(A + ko_ + a2, st_ + apiece + 2048);                                  \
            if constexpr (MI > 3) DMA16(A + ko_ + a3, st_ + apiece + 3072);                                  \
        }                                                                                                    \
    }
#define ISSUE_STAGE(slot, kt) { ISSUE_GROUP(slot, kt, 0) ISSUE_GROUP(slot, kt, 1) ISSUE_GROUP(slot, kt, 2) ISSUE_GROUP(slot, kt, 3) }

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fkc = lane >> 4;
    int xo[2][MI], wo[2][4];   // fragment byte offsets inside a stage, [k-step][tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MI; ++i) xo[ks][i] = lds_off_h(wm * (16 * MI) + i * 16 + frow, ks * 4 + fkc);
#pragma unroll
        for (int j = 0; j < 4; ++j) wo[ks][j] = A_BYTES + lds_off_h(wn * 64 + j * 16 + frow, ks * 4 + fkc);
    }
    // One 64-deep stage = two 32-deep k-steps.  The fragments of k-step 1 are read while the MFMAs of k-step 0 run, and the
    // DMA pieces of the stage requested this turn (IS0..IS3: MI + 4 pieces in four groups) are issued between the MFMA rows:
    // with one workgroup per CU (MI = 3 | 4) there is one wave per SIMD and every LDS read / DMA issue in front of the MFMAs is
    // exposed (tools/wgrad_stamps.py measured the same structure in the weight-gradient kernel: nothing overlaps inside one
    // wave unless it is interleaved by hand).  Same k order, same MFMA per accumulator as before: bit-identical results.
#define RD_STEP(xa_, wb_, ks_)                                                                    \
    {                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
            wb_[j] = *(const typename TT::vec8*)(st_ + wo[ks_][j]);                               \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                            \
            xa_[i] = *(const typename TT::vec8*)(st_ + xo[ks_][i]);                               \
    }
#define MROW64(xa_, wb_, i_) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i_][j] = TT::mfma(wb_[j], xa_[i_], acc[i_][j]);
#define SB64 __builtin_amdgcn_sched_barrier(0);
#define COMPUTE(slot, IS0, IS1, IS2, IS3)                                                         \
    {                                                                                             \
        const char* st_ = smem + (slot) * STAGE;                                                  \
        typename TT::vec8 xa0[MI], wb0[4], xa1[MI], wb1[4];                                       \
        RD_STEP(xa0, wb0, 0)                                                                      \
        SB64 IS0 SB64                                                                             \
        RD_STEP(xa1, wb1, 1)                                                                      \
        SB64 MROW64(xa0, wb0, 0) SB64 IS1                                                         \
        SB64 MROW64(xa0, wb0, 1) SB64 IS2                                                         \
        if constexpr (MI > 2) { SB64 MROW64(xa0, wb0, 2) }                                        \
        if constexpr (MI > 3) { SB64 MROW64(xa0, wb0, 3) }                                        \
        SB64 IS3 SB64                                                                             \
        MROW64(xa1, wb1, 0) MROW64(xa1, wb1, 1)                                                   \
        if constexpr (MI > 2) { MROW64(xa1, wb1, 2) }                                             \
        if constexpr (MI > 3) { MROW64(xa1, wb1, 3) }                                             \
    }
#define NOP64
    // publish: own DMAs of the stage retired (counted: MI + 4 per stage, `ahead` younger stages may stay in flight),
    // own LDS reads retired (their slot is recycled by the request issued right after the barrier), then the barrier
#define SYNC_AHEAD(ahead)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    wait_vm_lgkm0<(ahead) * (MI + 4)>();                                                          \
    __builtin_amdgcn_s_barrier();                                                                 \
    asm volatile("" ::: "memory");

    const int nt = p.K / BK;   // >= NS
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) ISSUE_STAGE(u, u)
    int slot = 0, nslot = NS - 1;   // slot of stage t, slot that stage t + NS - 1 goes to
    for (int t = 0; t < nt - (NS - 1); ++t) {
        SYNC_AHEAD(NS - 2)
        COMPUTE(slot, ISSUE_GROUP(nslot, t + NS - 1, 0), ISSUE_GROUP(nslot, t + NS - 1, 1), ISSUE_GROUP(nslot, t + NS - 1, 2),
                ISSUE_GROUP(nslot, t + NS - 1, 3))
        slot = slot == NS - 1 ? 0 : slot + 1;
        nslot = nslot == NS - 1 ? 0 : nslot + 1;
    }
    // tail: the last NS - 1 stages, nothing left to request
#define TAIL(k)                                                                                   \
    if constexpr (NS - 2 - (k) >= 0) {                                                            \
        SYNC_AHEAD(NS - 2 - (k))                                                                  \
        COMPUTE(slot, NOP64, NOP64, NOP64, NOP64)                                                 \
        slot = slot == NS - 1 ? 0 : slot + 1;                                                     \
    }
    TAIL(0) TAIL(1) TAIL(2) TAIL(3) TAIL(4)
#undef TAIL
#undef DMA16
#undef ISSUE_STAGE
#undef ISSUE_GROUP
#undef COMPUTE
#undef RD_STEP
#undef MROW64
#undef SB64
#undef NOP64
#undef SYNC_AHEAD

    // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4)
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

template <class TT, int MI, int NS>
hipError_t launch64(const GemmArgs& p, int epi, hipStream_t s) {
    constexpr int BM = 32 * MI;
    constexpr int RING = NS * (BM * BK * 2 + B_BYTES);
    if (p.K < NS * BK) return hipErrorInvalidValue;
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
#define LEAF_CASE(E)                                                                                         \
    case E: {                                                                                                \
        static bool attr_done = false;                                                                       \
        if (!attr_done) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt64_ring_kernel<TT, E, MI, NS>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, RING);                     \
            attr_done = true;                                                                                \
        }                                                                                                    \
        hipLaunchKernelGGL((gemm_nt64_ring_kernel<TT, E, MI, NS>), dim3(grid), dim3(256), RING, s, p);           \
        break;                                                                                               \
    }
    switch (epi) {
        LEAF_CASE(EPI_STORE_T)
        LEAF_CASE(EPI_ACT_T)
        LEAF_CASE(EPI_RESID_F32)
        LEAF_CASE(EPI_STORE_F32)
        LEAF_CASE(EPI_ACTGRAD_T)
        LEAF_CASE(EPI_LNFOLD_T)
        LEAF_CASE(EPI_LNFOLD_ACT_T)
        LEAF_CASE(EPI_RESID_LN)
        LEAF_CASE(EPI_RESID_LN8)
        default: return hipErrorInvalidValue;
    }
#undef LEAF_CASE
    return hipGetLastError();
}

// tile height: minimise (most tiles on one CU) x (bytes per stage); LEAF_GEMM64_MI = 2 | 3 | 4 forces one (A/B runs)
int pick_mi(const GemmArgs& p) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("LEAF_GEMM64_MI"); forced = e ? atoi(e) : 0; }
    if (forced >= 2 && forced <= 4) return forced;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        if (ncu < 1) ncu = 256;
    }
    int best = 2;
    long best_cost = -1;
    for (int mi = 2; mi <= 4; ++mi) {
        const long tiles = (long)((p.M + 32 * mi - 1) / (32 * mi)) * (p.N / BN);
        const long cost = ((tiles + ncu - 1) / ncu) * (32 * mi + BN);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mi; }
    }
    return best;
}

}  // namespace

bool leaf_gemm64_eligible(const GemmArgs& p) {
    // 32-bit byte offsets from the operand bases: both operands must span < 4 GiB
    const bool fits32 = (unsigned long long)p.M * p.lda * 2ull < (1ull << 32) && (unsigned long long)p.N * p.ldb * 2ull < (1ull << 32);
    return p.M > 0 && p.N % BN == 0 && p.K % BK == 0 && p.K >= 3 * BK && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 4 == 0 && fits32;
}

hipError_t leaf_launch_gemm64(const GemmArgs& p, int dtype, int epi, hipStream_t s) {
    const bool f = dtype == LEAF_F16;
    // LEAF_GEMM64_DEEP=1: 5-6 slot ring (one workgroup per CU, 4-5 stages of DMA latency hidden).  Measured no faster
    // than the 3-slot ring (3,234 x 768 x 3,072: 30 us either way, and the same for 64 / 96 / 128-row tiles): these
    // launches move ~270 MB from L2 to LDS in 30 us = 9 TB/s, the chip's L2 -> LDS rate, so the default stays 3 slots.
    static int deep_on = -1;
    if (deep_on < 0) { const char* e = getenv("LEAF_GEMM64_DEEP"); deep_on = (e && e[0] == '1') ? 1 : 0; }
    const int mi = pick_mi(p);
    const bool deep = deep_on && p.K >= 6 * BK;
#define GO(MI_, NS_) return f ? launch64<F16, MI_, NS_>(p, epi, s) : launch64<BF16, MI_, NS_>(p, epi, s)
    if (deep) {
        if (mi == 2) GO(2, 6);      // 6 x 24 KiB = 144 KiB
        if (mi == 3) GO(3, 5);      // 5 x 28 KiB = 140 KiB
        GO(4, 5);                   // 5 x 32 KiB = 160 KiB
    }
    if (mi == 2) GO(2, 3);
    if (mi == 3) GO(3, 3);
    GO(4, 3);
#undef GO
}
