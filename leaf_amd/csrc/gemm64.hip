// NT GEMM for SMALL launches (the B-caption passes: anchor, clean-caption K/V cache, training forward, data gradients;
// M = 3,200 rows at B = 128): 64 x 128 output tile, K streamed in 64-deep stages through a 3-slot LDS-DMA ring.
//
// The register-staged kernel of gemm.hip pays one exposed global round trip per 64-k tile (prefetch distance 1) and a
// 3,200 x 768 x 3,072 product took ~40 us on it whatever the tile count.  Here the loads are LDS-DMA pieces of
// 8 rows x 128 B (whole lines, source-side XOR swizzle, as in gemm256h.hip) requested TWO stages ahead, a stage is
// 24 KiB (A 64 rows, B 128 rows), three stages = 72 KiB so two workgroups share a CU, and one barrier per stage.
//     wait own DMAs of stage t | barrier | request stage t+2 into the slot stage t-1 just left | MFMAs on stage t
// 4 waves (2 x 2) of 32 x 64: acc[2][4], the k order and the epilogue arithmetic of every other GEMM kernel (bit-identical
// rows).  Requirements (host-checked): N % 128 == 0, K % 64 == 0, K >= 192, row strides % 8 == 0; M arbitrary.
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 64, BN = 128, BK = 64, NS = 3;
constexpr int A_BYTES = BM * BK * 2;      // 8 KiB
constexpr int B_BYTES = BN * BK * 2;      // 16 KiB
constexpr int STAGE = A_BYTES + B_BYTES;  // 24 KiB
constexpr int RING = NS * STAGE;          // 72 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ int lds_off_h(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <class TT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt64_ring_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = p.N / BN;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;

    // ---- DMA sources: a piece is 8 rows x 128 B; wave w moves A pieces 2w, 2w+1 and B pieces 4w .. 4w+3
    const int prow = lane >> 3;
    const int schunk = (lane & 7) ^ prow;
    const char* __restrict__ A = (const char*)p.A;
    const char* __restrict__ B = (const char*)p.B;
    auto arow = [&](int j) { int r = m0 + wid * 16 + 8 * j + prow; return r < p.M ? r : p.M - 1; };
    const unsigned a0 = (unsigned)arow(0) * (unsigned)p.lda * 2u + schunk * 16;
    const unsigned a1 = (unsigned)arow(1) * (unsigned)p.lda * 2u + schunk * 16;
    const unsigned b0 = (unsigned)(n0 + wid * 32 + prow) * (unsigned)p.ldb * 2u + schunk * 16;
    const unsigned bstep = 16u * (unsigned)p.ldb;   // 8 rows, bytes
    const int apiece = wid * 2048, bpiece = A_BYTES + wid * 4096;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
    // all six pieces of K stage `kt` into ring slot `slot`
#define ISSUE_STAGE(slot, kt)                                                                                \
    {                                                                                                        \
        char* st_ = smem + (slot) * STAGE;                                                                   \
        const size_t ko_ = (size_t)(kt) * (BK * 2);                                                          \
        DMA16(A + ko_ + a0, st_ + apiece);                                                                   \
        DMA16(A + ko_ + a1, st_ + apiece + 1024);                                                            \
        DMA16(B + ko_ + b0, st_ + bpiece);                                                                   \
        DMA16(B + ko_ + bstep + b0, st_ + bpiece + 1024);                                                    \
        DMA16(B + ko_ + 2 * (size_t)bstep + b0, st_ + bpiece + 2048);                                        \
        DMA16(B + ko_ + 3 * (size_t)bstep + b0, st_ + bpiece + 3072);                                        \
    }

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fkc = lane >> 4;
    int xo[2][2], wo[2][4];   // fragment byte offsets inside a stage, [k-step][tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xo[ks][i] = lds_off_h(wm * 32 + i * 16 + frow, ks * 4 + fkc);
#pragma unroll
        for (int j = 0; j < 4; ++j) wo[ks][j] = A_BYTES + lds_off_h(wn * 64 + j * 16 + frow, ks * 4 + fkc);
    }
#define COMPUTE(slot)                                                                             \
    {                                                                                             \
        const char* st_ = smem + (slot) * STAGE;                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                        \
            typename TT::vec8 xa[2], wb[4];                                                       \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
                xa[i] = *(const typename TT::vec8*)(st_ + xo[ks][i]);                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                         \
                wb[j] = *(const typename TT::vec8*)(st_ + wo[ks][j]);                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
                    acc[i][j] = TT::mfma(wb[j], xa[i], acc[i][j]);                                \
        }                                                                                         \
    }
    // publish: own DMAs of the stage retired (counted: 6 per stage in flight behind it), own LDS reads retired (their
    // slot is recycled by the request issued right after the barrier), then the workgroup barrier
#define SYNC(cnt)                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    asm volatile("s_waitcnt vmcnt(" #cnt ") lgkmcnt(0)" ::: "memory");                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    asm volatile("" ::: "memory");

    const int nt = p.K / BK;   // >= 3
    ISSUE_STAGE(0, 0)
    ISSUE_STAGE(1, 1)
    int slot = 0, nslot = 2;   // slot of stage t, slot that stage t + 2 goes to
    for (int t = 0; t < nt - 2; ++t) {
        SYNC(6)
        ISSUE_STAGE(nslot, t + 2)
        __builtin_amdgcn_sched_barrier(0);
        COMPUTE(slot)
        slot = slot == NS - 1 ? 0 : slot + 1;
        nslot = nslot == NS - 1 ? 0 : nslot + 1;
    }
    SYNC(6)
    COMPUTE(slot)
    slot = slot == NS - 1 ? 0 : slot + 1;
    SYNC(0)
    COMPUTE(slot)
#undef DMA16
#undef ISSUE_STAGE
#undef COMPUTE
#undef SYNC

    // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4)
    {
        const int nbase = n0 + wn * 64 + 4 * (lane >> 4);
        float4 bias4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = p.bias ? *(const float4*)(p.bias + nbase + 16 * j) : float4{0.f, 0.f, 0.f, 0.f};
        int mrow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) mrow[i] = m0 + wm * 32 + i * 16 + frow;
        epilogue_block<TT, EPI, 2, 4>(p, mrow, nbase, bias4, acc);
    }
}

template <class TT>
hipError_t launch64(const GemmArgs& p, int epi, hipStream_t s) {
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
#define LEAF_CASE(E)                                                                                         \
    case E: {                                                                                                \
        static bool attr_done = false;                                                                       \
        if (!attr_done) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt64_ring_kernel<TT, E>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, RING);                     \
            attr_done = true;                                                                                \
        }                                                                                                    \
        hipLaunchKernelGGL((gemm_nt64_ring_kernel<TT, E>), dim3(grid), dim3(256), RING, s, p);               \
        break;                                                                                               \
    }
    switch (epi) {
        LEAF_CASE(EPI_STORE_T)
        LEAF_CASE(EPI_ACT_T)
        LEAF_CASE(EPI_RESID_F32)
        LEAF_CASE(EPI_STORE_F32)
        LEAF_CASE(EPI_ACTGRAD_T)
        default: return hipErrorInvalidValue;
    }
#undef LEAF_CASE
    return hipGetLastError();
}

}  // namespace

bool leaf_gemm64_eligible(const GemmArgs& p) {
    return p.M > 0 && p.N % BN == 0 && p.K % BK == 0 && p.K >= NS * BK && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 4 == 0;
}

hipError_t leaf_launch_gemm64(const GemmArgs& p, int dtype, int epi, hipStream_t s) {
    return dtype == LEAF_F16 ? launch64<F16>(p, epi, s) : launch64<BF16>(p, epi, s);
}
