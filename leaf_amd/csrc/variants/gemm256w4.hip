// EXPERIMENT (round 5): the 256 x 256 x 64 NT GEMM as FOUR waves of 128 x 128 -- the wave shape of the vendor's
// hand-written hipBLASLt kernel (Custom_Cijk_..._MT256x256x64_MI16x16x1: 256 threads, 256 accumulator registers per lane, one wave
// per SIMD; tools/vendor_counters.sh measured it at a higher MFMA-busy x clock than gemm256h.hip's eight waves of 128 x 64) -- on
// THIS repo's operand path: the 5-slot half-stage LDS-DMA ring, whole-line pieces, source-side XOR swizzle.
//
// Per 32-deep k-step a wave issues 64 MFMAs against 16 fragment reads (4 : 1, the eight-wave kernel 2.67 : 1) and nothing of a
// partner wave competes for its SIMD.  Two fragment register sets: while set X multiplies, set Y (the next k-step) is read, so
// every ds_read has a whole k-step (64 MFMAs = 1,024 cycles) to land.  One barrier per K tile, placed between its two k-steps:
// behind it K tile T + 1 is visible (its first fragments are read during T's second k-step) and T's two slots are free.
//
// Accumulation order per element = ascending k in 32-steps with the same MFMA into one fp32 accumulator: the SAME BITS as every
// other GEMM kernel of the repo (tests/test_gpu_variants.py).  Diagnostic builds only (make variants; LEAF_GEMM_W4=1).
#include <type_traits>

#include "../common.h"
#include "../kernels.h"
#include "../lnfold.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NSLOT = 5;
constexpr int HALF = BM * BK * 2;       // 32 KiB: one operand panel of one K tile
constexpr int RING = NSLOT * HALF;      // 160 KiB
constexpr int WSLICE = 16384;           // epilogue staging per wave and pass

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ int lds_off_w4(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <class TT, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt256_w4_kernel(GemmArgs p) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    // tile order of gemm256h.hip: N tiles in groups of p.ngroup, inside a group M-major / N-minor, XCD-contiguous logical ids
    int m0, n0;
    {
        const int logical = xcd_remap(blockIdx.x, ntiles);
        int G = (p.ngroup > 0 && p.ngroup < tiles_n) ? p.ngroup : tiles_n;
        int g = logical / (tiles_m * G);
        const int ng = (tiles_n + G - 1) / G;
        if (g > ng - 1) g = ng - 1;
        const int rem = logical - g * tiles_m * G;
        const int gsz = g == ng - 1 ? tiles_n - g * G : G;
        m0 = (rem / gsz) * BM;
        n0 = (g * G + rem % gsz) * BN;
    }
    // ---- DMA sources: wave w moves pieces 8w .. 8w+7 (8 rows x 128 B each) of whichever panel a half-stage carries
    const char* __restrict__ A = (const char*)p.A;
    const char* __restrict__ B = (const char*)p.B;
    const int prow = lane >> 3;
    const int schunk = (lane & 7) ^ prow;
    unsigned aoff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int r = m0 + wid * 64 + 8 * q + prow;
        r = r < p.M ? r : p.M - 1;
        aoff[q] = (unsigned)r * (unsigned)p.lda * 2u + schunk * 16;
    }
    const unsigned b0 = (unsigned)(n0 + wid * 64 + prow) * (unsigned)p.ldb * 2u + schunk * 16;
    const unsigned bstep = 16u * (unsigned)p.ldb;   // 8 rows, bytes
    const int piece = wid * 8192;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
#define ISSUE_A(so, kt, q) DMA16(A + (size_t)((kt) * (BK * 2)) + aoff[q], smem + (so) + piece + (q) * 1024)
#define ISSUE_B(so, kt, q) DMA16(B + (size_t)((kt) * (BK * 2) + (q) * bstep) + b0, smem + (so) + piece + (q) * 1024)
#define SB __builtin_amdgcn_sched_barrier(0);

    typedef typename TT::vec8 frag_t;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag_t xa[8], xb[8], ya[8], yb[8];

    const int frow = lane & 15, fkc = lane >> 4;
    const int fo0 = lds_off_w4(frow, fkc), fo1 = lds_off_w4(frow, 4 + fkc);
    const int abase = wm * 128 * 128, bbase = wn * 128 * 128;
#define LD(ptr) (*(const frag_t*)(ptr))

    // One k-step: the 64 MFMAs of fragment set (CA, CB) in A-row-major order; in their shadows the 16 fragment reads of the NEXT
    // k-step into (NA, NB) from panel offsets (nsa, nsb) + nfo when RD, and the 8 DMA pieces IS(q) of one half-stage when DM.
#define KSTEP(CA, CB, NA, NB, RD, nsa, nsb, nfo, DM, IS)                                                    \
    {                                                                                                       \
        const char* pa_ = smem + (nsa) + abase + (nfo);                                                     \
        const char* pb_ = smem + (nsb) + bbase + (nfo);                                                     \
        _Pragma("unroll") for (int m_ = 0; m_ < 64; ++m_) {                                                  \
            const int i_ = m_ >> 3, j_ = m_ & 7;                                                            \
            SB acc[i_][j_] = TT::mfma(CB[j_], CA[i_], acc[i_][j_]);                                         \
            if (RD && (m_ & 3) == 1) {                                                                      \
                const int r_ = m_ >> 2;                                                                     \
                SB if (r_ < 8) NB[r_ & 7] = LD(pb_ + (r_ & 7) * 2048); else NA[r_ & 7] = LD(pa_ + (r_ & 7) * 2048); \
            }                                                                                               \
            if (DM && (m_ & 7) == 3) { SB IS(m_ >> 3) }                                                     \
        }                                                                                                   \
        SB                                                                                                  \
    }
#define SYNC(cnt)                                                                                           \
    SB asm volatile("s_waitcnt vmcnt(" #cnt ") lgkmcnt(0)" ::: "memory");                                   \
    __builtin_amdgcn_s_barrier();                                                                           \
    asm volatile("" ::: "memory"); SB

    const int nt = p.K / BK;   // >= 4 (host-checked)
    // half-stage u (u = 2T: A panel of K tile T, u = 2T + 1: B panel) lives in ring slot u % 5
    int sa = 0, sb = HALF;                 // slots of (A_T, B_T)
    int ia = 4 * HALF;                     // slot of A_{T+2} (requested during T's first k-step); B_{T+2} goes to A_T's slot
#define ADV2(x) { x += 2 * HALF; if (x >= RING) x -= RING; }
#pragma unroll
    for (int q = 0; q < 8; ++q) { ISSUE_A(0, 0, q); }
#pragma unroll
    for (int q = 0; q < 8; ++q) { ISSUE_B(HALF, 0, q); }
#pragma unroll
    for (int q = 0; q < 8; ++q) { ISSUE_A(2 * HALF, 1, q); }
#pragma unroll
    for (int q = 0; q < 8; ++q) { ISSUE_B(3 * HALF, 1, q); }
    SYNC(16)
#pragma unroll
    for (int r = 0; r < 8; ++r) { xb[r] = LD(smem + sb + bbase + fo0 + r * 2048); xa[r] = LD(smem + sa + abase + fo0 + r * 2048); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define IS_A2(q) ISSUE_A(ia, T + 2, q);
#define IS_B2(q) ISSUE_B(sa, T + 2, q);
#define IS_NONE(q)
    int T = 0;
    for (; T < nt - 2; ++T) {
        int na = sa, nb = sb;
        ADV2(na) ADV2(nb)
        KSTEP(xa, xb, ya, yb, true, sa, sb, fo1, true, IS_A2)       // k-step 0 of T; reads k-step 1 of T; requests A_{T+2}
        SYNC(8)                                                      // A_{T+1}, B_{T+1} landed (A_{T+2} may be in flight); T's slots are read out
        KSTEP(ya, yb, xa, xb, true, na, nb, fo0, true, IS_B2)       // k-step 1 of T; reads k-step 0 of T + 1; requests B_{T+2} into A_T's slot
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sa = na; sb = nb; ADV2(ia)
    }
    // ---- tile nt - 2: nothing left to request
    {
        int na = sa, nb = sb;
        ADV2(na) ADV2(nb)
        KSTEP(xa, xb, ya, yb, true, sa, sb, fo1, false, IS_NONE)
        SYNC(0)
        KSTEP(ya, yb, xa, xb, true, na, nb, fo0, false, IS_NONE)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sa = na; sb = nb;
    }
    // ---- tile nt - 1
    KSTEP(xa, xb, ya, yb, true, sa, sb, fo1, false, IS_NONE)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KSTEP(ya, yb, xa, xb, false, sa, sb, fo0, false, IS_NONE)

    // ---------------- epilogue: 16-bit outputs, two passes of 64 rows x 128 columns through this wave's 16-KiB LDS slice
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    char* sl = smem + wid * WSLICE;
    const int fq = lane >> 4, efrow = lane & 15;
    const int nb_ = n0 + wn * 128, mb_ = m0 + wm * 128;
    float4 bias4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias4[j] = p.bias ? *(const float4*)(p.bias + nb_ + 16 * j + 4 * fq) : float4{0.f, 0.f, 0.f, 0.f};
    static_assert(EPI == EPI_STORE_T, "experiment: only the bias + 16-bit store epilogue is built");
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * pass + ii;
            const int row = 16 * ii + efrow;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x2 v0 = f32x2{acc[i][j][0], acc[i][j][1]} + f32x2{bias4[j].x, bias4[j].y};
                const f32x2 v1 = f32x2{acc[i][j][2], acc[i][j][3]} + f32x2{bias4[j].z, bias4[j].w};
                const int c = 2 * j + (fq >> 1);
                *(uint2*)(sl + row * 256 + ((c ^ (row & 15)) << 4) + (fq & 1) * 8) = uint2{TT::pack2(v0), TT::pack2(v1)};
            }
        }
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = 4 * it + (lane >> 4), pc = lane & 15;
            const u32x4_t v = *(const u32x4_t*)(sl + row * 256 + (pc << 4));
            const int m = mb_ + 64 * pass + row;
            if (m < p.M) __builtin_nontemporal_store(v, (u32x4_t*)((u16*)p.C + (size_t)m * p.ldc + nb_ + ((pc ^ (row & 15)) << 3)));
        }
    }
#undef DMA16
#undef ISSUE_A
#undef ISSUE_B
#undef SB
#undef LD
#undef KSTEP
#undef SYNC
#undef ADV2
#undef IS_A2
#undef IS_B2
#undef IS_NONE
}

template <class TT>
hipError_t launch_w4(const GemmArgs& p_in, int epi, hipStream_t s) {
    GemmArgs p = p_in;
    p.ngroup = leaf_gemm256h_pick_ngroup(p);
    const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    if (epi != EPI_STORE_T) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_nt256_w4_kernel<TT, EPI_STORE_T>, hipFuncAttributeMaxDynamicSharedMemorySize, RING);
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_nt256_w4_kernel<TT, EPI_STORE_T>), dim3(ntiles), dim3(256), RING, s, p);
    return hipGetLastError();
}

}  // namespace

bool leaf_gemm256w4_eligible(const GemmArgs& p, int epi) {
    const long tiles = (long)((p.M + BM - 1) / BM) * (p.N / BN);
    const bool fits32 = (unsigned long long)p.M * p.lda * 2ull < (1ull << 32) && (unsigned long long)p.N * p.ldb * 2ull < (1ull << 32);
    return epi == EPI_STORE_T && p.N % BN == 0 && tiles >= 128 && p.K % BK == 0 && p.K >= 4 * BK && p.ldc % 8 == 0 && fits32;
}

hipError_t leaf_launch_gemm256w4(const GemmArgs& p, int dtype, int epi, hipStream_t s) {
    return dtype == LEAF_F16 ? launch_w4<F16>(p, epi, s) : launch_w4<BF16>(p, epi, s);
}
