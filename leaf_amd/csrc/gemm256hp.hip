// PERSISTENT form of the half-stage ring GEMM (gemm256h.hip): one workgroup per CU walks its tiles and the 5-slot ring
// simply keeps turning across tile seams.  After a tile's K loop its last two slots become the epilogue's staging area
// (8 KiB per wave) while the other three already receive the next tile's first half-stages (A0, B0, A1), requested
// BEFORE the epilogue: the 5-8 k-tick "first DMA wait" of every non-persistent tile (12-15 % of a K=768 tile,
// profiles/r01_gemm_stamps.txt) overlaps the epilogue instead.  The first barrier of a follow-on tile waits vmcnt(0)
// (the epilogue's stores share the counter with the DMAs); everything else is gemm256h.hip's schedule verbatim.
// Tile order: XCD x owns a contiguous range of the M-major / N-minor tile list, its workgroups take consecutive tiles
// round-robin (tiles in flight on one XCD share A panels through that XCD's L2).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NSLOT = 5;
constexpr int HALF = BM * BK * 2;       // 32 KiB: one operand panel of one K tile
constexpr int RING = NSLOT * HALF;      // 160 KiB

// In-kernel stamps (diagnostic builds only: -DLEAF_GEMM_STAMPS): s_memtime at phase boundaries, one 8-slot record per
// workgroup in a caller-supplied buffer that nothing else reads.
#ifdef LEAF_GEMM_STAMPS
#define STAMP(i)                                                                                          \
    if (p.stamps && tid == 0) {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        ((unsigned long long*)p.stamps)[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
#else
#define STAMP(i)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ int lds_off_h(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <class TT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt256_halfp_kernel(GemmArgs p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int tiles_n = p.N / BN;
    // ---- this workgroup's tiles: first, stride, count
    int t_first, t_stride, t_count;
    if ((int)gridDim.x == ntiles) {
        t_first = xcd_remap(blockIdx.x, ntiles); t_stride = 0; t_count = 1;
    } else {   // gridDim.x is a multiple of 8: gridDim.x / 8 workgroups per XCD
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int t0 = x * q + (x < r ? x : r), cnt = q + (x < r ? 1 : 0);
        t_first = t0 + slot; t_stride = per;
        t_count = slot < cnt ? (cnt - slot + per - 1) / per : 0;
    }
    if (t_count == 0) return;
    int m0 = 0, n0 = 0;

    // ---- DMA sources: wave w moves pieces 4w..4w+3 (8 rows x 128 B) of whichever panel a half-stage carries
    const int prow = lane >> 3;
    const int schunk = (lane & 7) ^ prow;
    const char* __restrict__ A = (const char*)p.A;
    const char* __restrict__ B = (const char*)p.B;
    // 32-bit byte offsets from the (uniform) operand bases: saddr + voffset addressing; recomputed per tile
    unsigned a0, a1, a2, a3, b0;
    const unsigned bstep = 16u * (unsigned)p.ldb;   // 8 rows, bytes
#define SET_TILE(t)                                                                                          \
    {                                                                                                        \
        {                                                                                                    \
            const int G_ = (p.ngroup > 0 && p.ngroup < tiles_n) ? p.ngroup : tiles_n;                        \
            const int tm_ = (p.M + BM - 1) / BM, ng_ = (tiles_n + G_ - 1) / G_;                              \
            int g_ = (t) / (tm_ * G_);                                                                       \
            if (g_ > ng_ - 1) g_ = ng_ - 1;                                                                  \
            const int rem_ = (t) - g_ * tm_ * G_;                                                            \
            const int gsz_ = g_ == ng_ - 1 ? tiles_n - g_ * G_ : G_;                                         \
            m0 = (rem_ / gsz_) * BM; n0 = (g_ * G_ + rem_ % gsz_) * BN;                                      \
        }                                                                                                    \
        const int r_ = m0 + wid * 32 + prow;                                                                 \
        const int c0_ = r_ < p.M ? r_ : p.M - 1, c1_ = r_ + 8 < p.M ? r_ + 8 : p.M - 1;                      \
        const int c2_ = r_ + 16 < p.M ? r_ + 16 : p.M - 1, c3_ = r_ + 24 < p.M ? r_ + 24 : p.M - 1;          \
        a0 = (unsigned)c0_ * (unsigned)p.lda * 2u + schunk * 16; a1 = (unsigned)c1_ * (unsigned)p.lda * 2u + schunk * 16; \
        a2 = (unsigned)c2_ * (unsigned)p.lda * 2u + schunk * 16; a3 = (unsigned)c3_ * (unsigned)p.lda * 2u + schunk * 16; \
        b0 = (unsigned)(n0 + wid * 32 + prow) * (unsigned)p.ldb * 2u + schunk * 16;                          \
    }
    const int piece = wid * 4096;
#ifdef LEAF_DIAG_NODMA   // diagnostic only: no operand traffic at all (results are garbage) - isolates the MFMA + LDS-read loop
#define DMA16(src, dst) asm volatile("" ::"v"(src), "v"(dst))
#else
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
#endif
    // piece q (0..3) of half-stage u: u even = A panel of K tile u/2, u odd = B panel
    // (so = byte offset of the ring slot that half-stage lands in, kt = its K tile)
#define ISSUE_A(so, kt, q) DMA16(A + (size_t)((kt) * (BK * 2)) + ((q) == 0 ? a0 : (q) == 1 ? a1 : (q) == 2 ? a2 : a3), smem + (so) + piece + (q) * 1024)
#define ISSUE_B(so, kt, q) DMA16(B + (size_t)((kt) * (BK * 2) + (q) * bstep) + b0, smem + (so) + piece + (q) * 1024)
#define ISSUE_HALF_A(so, kt) ISSUE_A(so, kt, 0); ISSUE_A(so, kt, 1); ISSUE_A(so, kt, 2); ISSUE_A(so, kt, 3);
#define ISSUE_HALF_B(so, kt) ISSUE_B(so, kt, 0); ISSUE_B(so, kt, 1); ISSUE_B(so, kt, 2); ISSUE_B(so, kt, 3);

    f32x4 acc[8][4];

    const int frow = lane & 15, fkc = lane >> 4;
    const int fo0 = lds_off_h(frow, fkc), fo1 = lds_off_h(frow, 4 + fkc);   // k-step 0 / 1 inside a 64-deep tile
    const int xbase = wm * 128 * 128, wbase = wn * 64 * 128;
    typedef typename TT::vec8 frag_t;
#ifdef LEAF_DIAG_NOREAD
    frag_t dummy_frag;
    asm volatile("" : "=v"(dummy_frag));
#endif
    frag_t Fx0, Fx1, Fx2, Fx3, Fx4, Fx5, Fx6, Fx7, Fw0, Fw1, Fw2, Fw3;
    frag_t Gx0, Gx1, Gx2, Gx3, Gx4, Gx5, Gx6, Gx7, Gw0, Gw1, Gw2, Gw3;
#ifdef LEAF_DIAG_NOREAD   // diagnostic only: no fragment reads (stale registers feed the MFMAs)
#define LD(ptr) (dummy_frag)
#else
#define LD(ptr) (*(const frag_t*)(ptr))
#endif
    // fragments of one 32-deep k-step: A panel at ring offset sa, B panel at sb
#define READ_FRAGS(P, sa, sb, fo)                                                                            \
    {                                                                                                        \
        const char* sa_ = smem + (sa) + xbase + (fo);                                                        \
        const char* sb_ = smem + (sb) + wbase + (fo);                                                        \
        P##w0 = LD(sb_); P##w1 = LD(sb_ + 2048); P##w2 = LD(sb_ + 4096); P##w3 = LD(sb_ + 6144);             \
        P##x0 = LD(sa_); P##x1 = LD(sa_ + 2048); P##x2 = LD(sa_ + 4096); P##x3 = LD(sa_ + 6144);             \
        P##x4 = LD(sa_ + 8192); P##x5 = LD(sa_ + 10240); P##x6 = LD(sa_ + 12288); P##x7 = LD(sa_ + 14336);   \
    }
#define MROW(P, i, xi)                                                                                       \
    acc[i][0] = TT::mfma(P##w0, xi, acc[i][0]); acc[i][1] = TT::mfma(P##w1, xi, acc[i][1]);                   \
    acc[i][2] = TT::mfma(P##w2, xi, acc[i][2]); acc[i][3] = TT::mfma(P##w3, xi, acc[i][3]);
#define MFMA_H1(P) MROW(P, 0, P##x0) MROW(P, 1, P##x1) MROW(P, 2, P##x2) MROW(P, 3, P##x3)
#define MFMA_H2(P) MROW(P, 4, P##x4) MROW(P, 5, P##x5) MROW(P, 6, P##x6) MROW(P, 7, P##x7)
#define SB __builtin_amdgcn_sched_barrier(0);
#ifdef LEAF_DIAG_NOBAR    // diagnostic only: no per-tile workgroup barrier
#define DIAG_BARRIER
#else
#define DIAG_BARRIER __builtin_amdgcn_s_barrier();
#endif
#define SYNC_TILE(cnt)                                                                                       \
    SB                                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #cnt ") lgkmcnt(0)" ::: "memory");                                       \
    DIAG_BARRIER                                                                                             \
    asm volatile("" ::: "memory");
    // one 32-deep k-step, software-pipelined at instruction granularity: the 12 fragment reads of this k-step (CUR) are
    // issued ONE AT A TIME in the shadow of individual MFMAs - first under the second half (rows 4-7) of the previous
    // k-step's MFMAs, the last four (rows 4-7 of CUR, not needed before the next k-step) under CUR's own first half -
    // instead of twelve back-to-back ds_read_b128 that stall the wave's in-order MFMA issue (diagnostic builds: the
    // bunched reads cost 10 % of a tile).  The four DMA pieces of a half-stage are spread in between (IS = issue macro).
#define MF(P, i, j) acc[i][j] = TT::mfma(P##w##j, P##x##i, acc[i][j]);
#define RDW(P, n) P##w##n = LD(sb_ + (n) * 2048);
#define RDX(P, n) P##x##n = LD(sa_ + (n) * 2048);
#ifdef LEAF_GEMM_BUNCHED_READS   // the previous schedule, kept for A/B builds
#define KSTEP(PREV, CUR, sa, sb, fo, IS0, IS1, IS2, IS3)                                                     \
    READ_FRAGS(CUR, sa, sb, fo)                                                                              \
    SB MROW(PREV, 4, PREV##x4) SB IS0                                                                        \
    SB MROW(PREV, 5, PREV##x5) SB IS1                                                                        \
    SB MROW(PREV, 6, PREV##x6) SB IS2                                                                        \
    SB MROW(PREV, 7, PREV##x7) SB IS3                                                                        \
    SB MFMA_H1(CUR)
#else
#define KSTEP(PREV, CUR, sa, sb, fo, IS0, IS1, IS2, IS3)                                                     \
    {                                                                                                        \
        const char* sa_ = smem + (sa) + xbase + (fo);                                                        \
        const char* sb_ = smem + (sb) + wbase + (fo);                                                        \
        SB MF(PREV, 4, 0) SB RDW(CUR, 0) SB MF(PREV, 4, 1) SB RDW(CUR, 1) SB MF(PREV, 4, 2) SB RDW(CUR, 2)    \
        SB MF(PREV, 4, 3) SB RDW(CUR, 3) SB IS0                                                              \
        SB MF(PREV, 5, 0) SB RDX(CUR, 0) SB MF(PREV, 5, 1) MF(PREV, 5, 2) SB RDX(CUR, 1) SB MF(PREV, 5, 3)    \
        SB IS1                                                                                               \
        SB MF(PREV, 6, 0) SB RDX(CUR, 2) SB MF(PREV, 6, 1) MF(PREV, 6, 2) SB RDX(CUR, 3) SB MF(PREV, 6, 3)    \
        SB IS2                                                                                               \
        SB MROW(PREV, 7, PREV##x7) SB IS3                                                                    \
        SB MF(CUR, 0, 0) SB RDX(CUR, 4) SB MF(CUR, 0, 1) MF(CUR, 0, 2) SB RDX(CUR, 5) SB MF(CUR, 0, 3)        \
        SB MF(CUR, 1, 0) SB RDX(CUR, 6) SB MF(CUR, 1, 1) MF(CUR, 1, 2) SB RDX(CUR, 7) SB MF(CUR, 1, 3)        \
        SB MROW(CUR, 2, CUR##x2) MROW(CUR, 3, CUR##x3) SB                                                    \
    }
#endif
#define NOP_

    const int nt = p.K / BK;   // K tiles, >= 4 (host-checked)
    // half-stage u lives in ring slot u % 5; the offsets below are uniform and advance by two slots per K tile, and
    // keep advancing from one output tile to the next
#define ADV(x) { x += 2 * HALF; if (x >= RING) x -= RING; }
#define WRAP(x) ((x) >= RING ? (x) - RING : (x))
    int sa = 0, sb = HALF;               // slots of (A, B) of the K tile being multiplied
    SET_TILE(t_first)
    ISSUE_HALF_A(sa, 0) ISSUE_HALF_B(sb, 0) ISSUE_HALF_A(2 * HALF, 1)
    for (int it = 0; it < t_count; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int i0 = WRAP(sa + 3 * HALF), i1 = WRAP(sa + 4 * HALF);   // slots of the half-stages requested during K tile T: 2T+3 (B), 2T+4 (A)
    // ---- K tile 0: a follow-on tile's first barrier also waits for the previous epilogue's stores (same counter)
    if (it == 0) { SYNC_TILE(4) } else { SYNC_TILE(0) }
    READ_FRAGS(G, sa, sb, fo0)
    SB ISSUE_B(i0, 1, 0); ISSUE_B(i0, 1, 1); SB
    MROW(G, 0, Gx0) MROW(G, 1, Gx1) SB ISSUE_B(i0, 1, 2); ISSUE_B(i0, 1, 3); SB MROW(G, 2, Gx2) MROW(G, 3, Gx3)
    KSTEP(G, F, sa, sb, fo1, ISSUE_A(i1, 2, 0);, ISSUE_A(i1, 2, 1);, ISSUE_A(i1, 2, 2);, ISSUE_A(i1, 2, 3);)
    ADV(sa) ADV(sb) ADV(i0) ADV(i1)
    // ---- K tiles 1 .. nt-3: request B of tile T+1 and A of tile T+2
    int T = 1;
    for (; T <= nt - 3; ++T) {
        SYNC_TILE(4)
        KSTEP(F, G, sa, sb, fo0, ISSUE_B(i0, T + 1, 0);, ISSUE_B(i0, T + 1, 1);, ISSUE_B(i0, T + 1, 2);, ISSUE_B(i0, T + 1, 3);)
        KSTEP(G, F, sa, sb, fo1, ISSUE_A(i1, T + 2, 0);, ISSUE_A(i1, T + 2, 1);, ISSUE_A(i1, T + 2, 2);, ISSUE_A(i1, T + 2, 3);)
        ADV(sa) ADV(sb) ADV(i0) ADV(i1)
    }
    // ---- K tile nt-2: only the B panel of the last tile is left to request
    SYNC_TILE(4)
    KSTEP(F, G, sa, sb, fo0, ISSUE_B(i0, T + 1, 0);, ISSUE_B(i0, T + 1, 1);, ISSUE_B(i0, T + 1, 2);, ISSUE_B(i0, T + 1, 3);)
    KSTEP(G, F, sa, sb, fo1, NOP_, NOP_, NOP_, NOP_)
    ADV(sa) ADV(sb)
    // ---- K tile nt-1
    SYNC_TILE(0)
    KSTEP(F, G, sa, sb, fo0, NOP_, NOP_, NOP_, NOP_)
    KSTEP(G, F, sa, sb, fo1, NOP_, NOP_, NOP_, NOP_)
    SB MFMA_H2(F) SB
    // ---------------- seam: the last K tile's two slots become the staging area, the other three take the next tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nb = n0 + wn * 64;          // first column of this wave's sub-tile (of the tile just multiplied)
    const int mb = m0 + wm * 128;         // first row
    char* sl = smem + (wid < 4 ? sa : sb) + (wid & 3) * 8192;
    ADV(sa) ADV(sb)
    if (it + 1 < t_count) {
        SET_TILE(t_first + (it + 1) * t_stride)
        ISSUE_HALF_A(sa, 0) ISSUE_HALF_B(sb, 0) ISSUE_HALF_A(WRAP(sa + 2 * HALF), 1)
    }
    if (p.alpha) {   // gradient un-scaling (weight gradients of the fp16 loss-scaled backward)
        const float al = *p.alpha;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= al;
    }
    const int fq = lane >> 4;
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
                if (m < p.M) {
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)(dst + (size_t)m * p.ldc + nb + ((pc ^ (row & 7)) << 3)));
                }
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
                if (m < p.M) {
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, (f32x4_t*)((float*)p.C + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2)));
                }
            }
        }
    }
    }   // tiles of this workgroup
#undef ADV
#undef DMA16
#undef ISSUE_A
#undef ISSUE_B
#undef ISSUE_HALF_A
#undef ISSUE_HALF_B
#undef READ_FRAGS
#undef MROW
#undef MFMA_H1
#undef MFMA_H2
#undef SYNC_TILE
#undef KSTEP
#undef MF
#undef RDW
#undef RDX
#undef NOP_
#undef SB
#undef LD

#undef WRAP
#undef SET_TILE
}

template <class TT>
hipError_t launch256hp(const GemmArgs& p_in, int epi, hipStream_t s) {
    GemmArgs p = p_in;
    p.ngroup = leaf_gemm256h_pick_ngroup(p);
    const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        ncu = prop.multiProcessorCount & ~7;   // multiple of 8 (XCDs)
        if (ncu < 8) ncu = 8;
    }
    const int grid = ntiles <= ncu ? ntiles : ncu;
#define LEAF_CASE(E)                                                                                         \
    case E: {                                                                                                \
        static bool attr_done = false;                                                                       \
        if (!attr_done) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt256_halfp_kernel<TT, E>,                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, RING);                     \
            attr_done = true;                                                                                \
        }                                                                                                    \
        hipLaunchKernelGGL((gemm_nt256_halfp_kernel<TT, E>), dim3(grid), dim3(512), RING, s, p, ntiles);     \
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

hipError_t leaf_launch_gemm256hp(const GemmArgs& p, int dtype, int epi, hipStream_t s) {
    return dtype == LEAF_F16 ? launch256hp<F16>(p, epi, s) : launch256hp<BF16>(p, epi, s);
}
