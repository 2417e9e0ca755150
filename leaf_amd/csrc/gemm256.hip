// NT GEMM, 256x256 output tile, 4-stage LDS-DMA ring -- the second-generation kernel (LEAF_GEMM_V=2); the default for
// the text tower's big GEMMs is now the half-stage ring of gemm256h.hip, which grew out of this one.
//
//   C[M,N] (+)= A[M,K] * B[N,K]^T      16-bit operands (fp16 / bf16), fp32 accumulate, fused epilogues
//
// Why this shape (measured on MI355X, profiles/, tools/gemm_stamps.py): at K = 768 a 256^2 tile is only 12 K-tiles
// deep, so (a) one K-tile of look-ahead left every iteration waiting ~3k cycles for its LDS-DMA against 2k cycles
// of MFMA, and (b) a row-per-lane store epilogue cost 16-33k cycles per tile (store-issue bound, ~7 B/clk/CU)
// against ~25k cycles of MFMA.  Hence:
//   * K is streamed in 32-deep stages (32 KiB: A 256x32 + B 256x32) through a 4-slot ring; three stages
//     (96 KiB per CU) are in flight ahead of the one being multiplied, retired with COUNTED s_waitcnt vmcnt(8/4/0)
//     and one raw s_barrier per stage (hipcc's __syncthreads would drain the DMA queue);
//   * operands go HBM/L2 -> LDS by global_load_lds_dwordx4 (no staging VGPRs); a DMA piece is 16 rows x 64 B,
//     LDS-linear, with the bank swizzle chunk ^= 3*((row>>3)&1) applied to the per-lane SOURCE address and again on
//     the ds_read_b128 fragment reads (conflict-free for the 16x16x32 MFMA operand pattern);
//   * 8 waves as 2(M) x 4(N), 128x64 per wave = 8x4 MFMA 16x16x32 tiles (128 accumulator VGPRs), issued as
//     D = Wfrag * Xfrag^T so a lane owns 4 consecutive n of one row m;
//   * the epilogue re-shapes each wave's sub-tile through its private 16 KiB slice of the (now idle) ring and
//     writes/reads global memory in whole 128-B (16-bit) or 256-B (fp32) row segments, 16 B per lane.
// Host-side requirements: N % 256 == 0, K % 32 == 0, K >= 128, lda/ldb % 8 == 0, ldc % 8 == 0; M arbitrary.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

#ifndef LEAF_NSTAGE
#define LEAF_NSTAGE 4
#endif
constexpr int BM = 256, BN = 256, BKS = 32, NSTAGE = LEAF_NSTAGE;   // ring slots (4: 128 KiB, 5: all 160 KiB of LDS)
constexpr int PART = BM * BKS * 2;      // 16 KiB: one operand of one stage
constexpr int STAGE = 2 * PART;         // 32 KiB
constexpr int RING = NSTAGE * STAGE;    // 128 KiB
constexpr int SLICE = RING / 8;         // 16 KiB per wave for the epilogue

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

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
__global__ __launch_bounds__(512, 2) void gemm_nt256_ring_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int tiles_n = p.N / BN;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
#ifdef LEAF_NGROUP
    // column-grouped order: all M-tiles for a group of LEAF_NGROUP N-tiles, then the next group (B group stays in L2)
    const int tiles_m = (p.M + BM - 1) / BM;
    const int per_group = tiles_m * LEAF_NGROUP;
    const int grp = logical / per_group, rem = logical - grp * per_group;
    const int gw = (tiles_n - grp * LEAF_NGROUP) < LEAF_NGROUP ? (tiles_n - grp * LEAF_NGROUP) : LEAF_NGROUP;
    const int m0 = (rem / gw) * BM, n0 = (grp * LEAF_NGROUP + rem % gw) * BN;
#else
    const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
#endif

    // ---- DMA sources: wave w owns pieces 2w, 2w+1 (16 rows x 64 B each) of the A part and of the B part
    const int prow = lane >> 2;
    const int schunk = (lane & 3) ^ (((prow >> 3) & 1) * 3);
    const u16* __restrict__ A = (const u16*)p.A;
    const u16* __restrict__ B = (const u16*)p.B;
    auto arow = [&](int j) { int r = m0 + wid * 32 + 16 * j + prow; return r < p.M ? r : p.M - 1; };
    const u16* a0 = A + (size_t)arow(0) * p.lda + schunk * 8;
    const u16* a1 = A + (size_t)arow(1) * p.lda + schunk * 8;
    const u16* b0 = B + (size_t)(n0 + wid * 32 + prow) * p.ldb + schunk * 8;
    const u16* b1 = b0 + (size_t)16 * p.ldb;
    const int piece = wid * 2048;
#ifndef LEAF_A_AUX
#define LEAF_A_AUX 0
#endif
#ifndef LEAF_B_AUX
#define LEAF_B_AUX 0
#endif
#define DMA16(src, dst, aux) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, aux)
#define ISSUE_STAGE(slot, k0)                                                              \
    {                                                                                      \
        char* sa_ = smem + (slot) * STAGE + piece;                                         \
        DMA16(a0 + (k0), sa_, LEAF_A_AUX);        DMA16(a1 + (k0), sa_ + 1024, LEAF_A_AUX);               \
        DMA16(b0 + (k0), sa_ + PART, LEAF_B_AUX); DMA16(b1 + (k0), sa_ + PART + 1024, LEAF_B_AUX);        \
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fkc = lane >> 4;
    const int fo = frow * 64 + ((fkc ^ (((frow >> 3) & 1) * 3)) << 4);
    const int xoff = wm * 8192 + fo, woff = PART + wn * 4096 + fo;
    typedef typename TT::vec8 frag_t;
    // Software pipeline over stages: the 12 fragment reads of stage t are issued right after the barrier that
    // publishes it and are covered by the second half of stage t-1's MFMAs (rows 64..127 of the wave tile), which
    // still run from registers; then the first half of stage t follows.  Two named fragment sets (F, G) alternate.
    frag_t Fx0, Fx1, Fx2, Fx3, Fx4, Fx5, Fx6, Fx7, Fw0, Fw1, Fw2, Fw3;
    frag_t Gx0, Gx1, Gx2, Gx3, Gx4, Gx5, Gx6, Gx7, Gw0, Gw1, Gw2, Gw3;
#define LD(ptr) (*(const frag_t*)(ptr))
#define READ_FRAGS(P, slot)                                                                                  \
    {                                                                                                        \
        const char* st_ = smem + (slot) * STAGE;                                                             \
        P##w0 = LD(st_ + woff); P##w1 = LD(st_ + woff + 1024); P##w2 = LD(st_ + woff + 2048); P##w3 = LD(st_ + woff + 3072); \
        P##x0 = LD(st_ + xoff); P##x1 = LD(st_ + xoff + 1024); P##x2 = LD(st_ + xoff + 2048); P##x3 = LD(st_ + xoff + 3072); \
        P##x4 = LD(st_ + xoff + 4096); P##x5 = LD(st_ + xoff + 5120); P##x6 = LD(st_ + xoff + 6144); P##x7 = LD(st_ + xoff + 7168); \
    }
#define MROW(P, i, xi)                                                                                       \
    acc[i][0] = TT::mfma(P##w0, xi, acc[i][0]); acc[i][1] = TT::mfma(P##w1, xi, acc[i][1]);                   \
    acc[i][2] = TT::mfma(P##w2, xi, acc[i][2]); acc[i][3] = TT::mfma(P##w3, xi, acc[i][3]);
#define MFMA_H1(P) MROW(P, 0, P##x0) MROW(P, 1, P##x1) MROW(P, 2, P##x2) MROW(P, 3, P##x3)
#define MFMA_H2(P) MROW(P, 4, P##x4) MROW(P, 5, P##x5) MROW(P, 6, P##x6) MROW(P, 7, P##x7)
    // publish stage: own DMAs of it retired (counted), own outstanding LDS reads retired (their slot is recycled by the
    // DMA issued right after the barrier), then the workgroup barrier
#define SYNC_STAGE(cnt)                                                                                      \
    __builtin_amdgcn_sched_barrier(0);   /* MFMAs are register-only: "memory" alone does not pin them */      \
    asm volatile("s_waitcnt vmcnt(" #cnt ") lgkmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("" ::: "memory");
    // One pipeline step.  The four DMA issues of the stage three ahead are spread between the MFMA row-groups:
    // issuing them back to back right after the barrier makes all 8 waves queue on the CU's address unit at once and
    // the MFMAs behind them wait (an LDS-DMA issue costs 60-185 cycles in a busy phase, MI355X_MICROARCH.md).
#define SB __builtin_amdgcn_sched_barrier(0);
#define SLOT(x) (NSTAGE == 4 ? ((x) & 3) : ((x) % NSTAGE))
#define STEP(PREV, CUR, tt, cnt, issue)                                                                      \
    SYNC_STAGE(cnt)                                                                                          \
    READ_FRAGS(CUR, SLOT(tt))                                                                                \
    SB                                                                                                       \
    {                                                                                                        \
        char* sa_ = smem + SLOT((tt) + NSTAGE - 1) * STAGE + piece;                                          \
        const int k0_ = ((tt) + NSTAGE - 1) * BKS;                                                                    \
        MROW(PREV, 4, PREV##x4) SB                                                                           \
        if (issue) DMA16(a0 + k0_, sa_, LEAF_A_AUX);                                                         \
        SB MROW(PREV, 5, PREV##x5) SB                                                                        \
        if (issue) DMA16(a1 + k0_, sa_ + 1024, LEAF_A_AUX);                                                  \
        SB MROW(PREV, 6, PREV##x6) SB                                                                        \
        if (issue) DMA16(b0 + k0_, sa_ + PART, LEAF_B_AUX);                                                  \
        SB MROW(PREV, 7, PREV##x7) SB                                                                        \
        if (issue) DMA16(b1 + k0_, sa_ + PART + 1024, LEAF_B_AUX);                                           \
        SB                                                                                                   \
    }                                                                                                        \
    MFMA_H1(CUR)

    const int nt = p.K / BKS;   // even, >= 8 (host-checked)
    STAMP(0)
    ISSUE_STAGE(0, 0)
    ISSUE_STAGE(1, BKS)
    ISSUE_STAGE(2, 2 * BKS)
#if LEAF_NSTAGE == 5
    ISSUE_STAGE(3, 3 * BKS)
    SYNC_STAGE(12)
    STAMP(1)
    ISSUE_STAGE(4, 4 * BKS)
#else
    SYNC_STAGE(8)
    STAMP(1)
    ISSUE_STAGE(3, 3 * BKS)
#endif
    READ_FRAGS(F, 0)
    __builtin_amdgcn_sched_barrier(0);
    MFMA_H1(F)
    int t = 1;
#if LEAF_NSTAGE == 5
    for (; t <= nt - 7; t += 2) {        // steps t, t+1 <= nt-6: four stages in flight, a new one issued each step
        STEP(F, G, t, 12, true)
        STEP(G, F, t + 1, 12, true)
    }
    STEP(F, G, t, 12, true)              // t = nt-5 (last issue)
    STAMP(2)
    STEP(G, F, t + 1, 12, false)         // nt-4
    STEP(F, G, t + 2, 8, false)          // nt-3
    STEP(G, F, t + 3, 4, false)          // nt-2
    STEP(F, G, t + 4, 0, false)          // nt-1
    MFMA_H2(G)
#else
    for (; t <= nt - 5; t += 2) {        // steps t, t+1 <= nt-4: counted wait 8, a new stage issued each step
        STEP(F, G, t, 8, true)
        STEP(G, F, t + 1, 8, true)
    }
    STAMP(2)
    STEP(F, G, t, 8, false)              // t = nt-3
    STEP(G, F, t + 1, 4, false)          // nt-2
    STEP(F, G, t + 2, 0, false)          // nt-1
    MFMA_H2(G)
#endif
    STAMP(3)
#undef DMA16
#undef ISSUE_STAGE
#undef READ_FRAGS
#undef MROW
#undef MFMA_H1
#undef MFMA_H2
#undef SYNC_STAGE
#undef STEP
#undef SB
#undef SLOT
#undef LD

    // ---------------- epilogue through this wave's private LDS slice (ring is idle after one more barrier)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (p.alpha) {   // gradient un-scaling (weight gradients of the fp16 loss-scaled backward)
        const float al = *p.alpha;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= al;
    }
    char* sl = smem + wid * SLICE;
    const int fq = lane >> 4;
    const int nb = n0 + wn * 64;          // first column of this wave's sub-tile
    const int mb = m0 + wm * 128;         // first row
    float4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        bias4[j] = p.bias ? *(const float4*)(p.bias + nb + 16 * j + 4 * fq) : float4{0.f, 0.f, 0.f, 0.f};

    if constexpr (EPI == EPI_STORE_T || EPI == EPI_ACT_T) {
        // two passes of 64 rows x 64 cols of 16-bit: LDS rows of 128 B, 16-B chunks XOR-swizzled by (row & 7)
        // ACTC: std::integral_constant<int, -1 | ACT_GELU | ACT_QUICKGELU> - the activation is fixed at compile time inside
        // the element loops (a run-time id there costs one branch per element and serialises the transcendental chains)
        auto stage16 = [&](int pass, auto ACTC) {
            constexpr int ACT = decltype(ACTC)::value;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = 4 * pass + ii;
                const int row = 16 * ii + frow;
                float v[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j][0] = acc[i][j][0] + bias4[j].x; v[j][1] = acc[i][j][1] + bias4[j].y;
                    v[j][2] = acc[i][j][2] + bias4[j].z; v[j][3] = acc[i][j][3] + bias4[j].w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = act_fwd_t<ACT>(v[j][e]);     // 16 independent chains
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 2 * j + (fq >> 1);
                    *(uint2*)(sl + row * 128 + ((c ^ (row & 7)) << 4) + (fq & 1) * 8) = pack4<TT>(v[j][0], v[j][1], v[j][2], v[j][3]);
                }
            }
        };
        typedef std::integral_constant<int, -1> NoAct;
        typedef std::integral_constant<int, ACT_GELU> Gelu;
        typedef std::integral_constant<int, ACT_QUICKGELU> QuickGelu;
        auto flush16 = [&](int pass, u16* dst) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 8 * it + (lane >> 3), pc = lane & 7;
                const uint4 v = *(const uint4*)(sl + row * 128 + (pc << 4));
                const int m = mb + 64 * pass + row;
                if (m < p.M) *(uint4*)(dst + (size_t)m * p.ldc + nb + ((pc ^ (row & 7)) << 3)) = v;
            }
        };
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (EPI == EPI_ACT_T && p.aux) {   // training forward: pre-activation stash first
                stage16(pass, NoAct());
                flush16(pass, (u16*)p.aux);
            }
            if (EPI != EPI_ACT_T) stage16(pass, NoAct());
            else if (p.act == ACT_QUICKGELU) stage16(pass, QuickGelu());
            else stage16(pass, Gelu());
            flush16(pass, (u16*)p.C);
        }
    } else {
        // fp32 outputs: four passes of 32 rows x 64 cols: LDS rows of 256 B, 16-B chunks XOR-swizzled by (row & 15)
        const float beta = (EPI == EPI_RESID_F32) ? 1.f : p.beta;
        const float* rsrc = (EPI == EPI_RESID_F32 && p.aux) ? (const float*)p.aux : (const float*)p.C;   // out-of-place residual
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            float4 res[8];
            if (beta != 0.f) {   // fetch the residual rows of this pass first: 8 coalesced 16-B loads in flight
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = 4 * it + (lane >> 4), pc = lane & 15;
                    const int m = mb + 32 * pass + row;
                    res[it] = m < p.M ? *(const float4*)(rsrc + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2))
                                      : float4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * pass + ii;
                const int row = 16 * ii + frow;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * j + fq;
                    *(float4*)(sl + row * 256 + ((c ^ (row & 15)) << 4)) =
                        float4{acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z,
                               acc[i][j][3] + bias4[j].w};
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 4 * it + (lane >> 4), pc = lane & 15;
                float4 v = *(const float4*)(sl + row * 256 + (pc << 4));
                const int m = mb + 32 * pass + row;
                if (beta != 0.f) {
                    v.x = __builtin_fmaf(res[it].x, beta, v.x); v.y = __builtin_fmaf(res[it].y, beta, v.y);
                    v.z = __builtin_fmaf(res[it].z, beta, v.z); v.w = __builtin_fmaf(res[it].w, beta, v.w);
                }
                if (m < p.M) *(float4*)((float*)p.C + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2)) = v;
            }
        }
    }
    STAMP(4)
}

template <class TT>
hipError_t launch256(const GemmArgs& p, int epi, hipStream_t s) {
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
#define LEAF_CASE(E)                                                                                         \
    case E: {                                                                                                \
        static bool attr_done = false;                                                                       \
        if (!attr_done) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt256_ring_kernel<TT, E>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, RING);                     \
            attr_done = true;                                                                                \
        }                                                                                                    \
        hipLaunchKernelGGL((gemm_nt256_ring_kernel<TT, E>), dim3(grid), dim3(512), RING, s, p);              \
        break;                                                                                               \
    }
    switch (epi) {
        LEAF_CASE(EPI_STORE_T)
        LEAF_CASE(EPI_ACT_T)
        LEAF_CASE(EPI_RESID_F32)
        LEAF_CASE(EPI_STORE_F32)
        default: return hipErrorInvalidValue;
    }
#undef LEAF_CASE
    return hipGetLastError();
}

}  // namespace

bool leaf_gemm256_eligible(const GemmArgs& p, int epi) {
    // at least half a wave of 256^2 tiles over the 256 CUs, else the 128^2 kernel fills the chip better (weight gradients)
    const long tiles = (long)((p.M + BM - 1) / BM) * (p.N / BN);
    return p.N % BN == 0 && tiles >= 128 && p.K % (2 * BKS) == 0 && p.K >= 8 * BKS && p.ldc % 8 == 0 && epi != EPI_ACTGRAD_T;
}

hipError_t leaf_launch_gemm256(const GemmArgs& p, int dtype, int epi, hipStream_t s) {
    return dtype == LEAF_F16 ? launch256<F16>(p, epi, s) : launch256<BF16>(p, epi, s);
}
