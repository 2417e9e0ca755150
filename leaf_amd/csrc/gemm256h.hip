// NT GEMM, 256x256 output tile, FULL-LINE LDS-DMA pieces in a 5-slot half-stage ring: the GEMM of every launch with >= 128 tiles.
//
// 8 waves (2 x 4, 128 x 64 each), software-pipelined fragment reads, DMA issue spread between MFMA row groups, LDS-staged
// epilogue.  K is streamed as 64-deep HALF-stages: one half-stage = ONE operand panel of 256 rows x 64 k (32 KiB, rows of 128 B), so a
// DMA piece is 8 rows x 128 B = whole 128-B lines (tools/dma_probe*.hip: 45 B/clk/CU L2-hit fill against 25 B/clk/CU
// for the 16 x 64 B pieces of the 32-deep stages).  Five half-slots = all 160 KiB of LDS: while tile t (A_t, B_t) is
// multiplied, A_{t+1}, B_{t+1} and A_{t+2} are in flight; ONE barrier per 64 k.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "lnfold.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NSLOT = 5;
constexpr int HALF = BM * BK * 2;       // 32 KiB: one operand panel of one K tile
constexpr int RING = NSLOT * HALF;      // 160 KiB
constexpr int SLICE = 16384;            // epilogue staging per wave (inside the idle ring)

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

// PERSIST (the 16-bit-output epilogues: QKV, c_fc): one workgroup per CU walks tiles v = blockIdx.x, + gridDim.x, ... in
// the same logical order a plain launch is dispatched in, and requests the NEXT tile's first A / B half-stages before it
// runs the epilogue of the current one, so that the 12-15 % first-DMA wait of a K = 768 tile (all 256 workgroups asking
// L2 for their first 96 KiB at once) disappears under the epilogue.  The epilogue stages through 8 KiB per wave in ring
// slots 3-4, slots 0-1 take the prefetch, slot 2 the LN-folding row table; the next tile's barrier protects all three.
template <class TT, int EPI, bool PERSIST>
__global__ __launch_bounds__(512, 2) void gemm_nt256_half_kernel(GemmArgs p) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    // Tile order: N tiles in groups of p.ngroup; inside a group M-major / N-minor.  An XCD (a contiguous range of logical
    // ids) then sweeps many M panels against ONE group's B panels, which stay in its 4-MiB L2 instead of being re-fetched
    // for every round of 32 tiles (the host picks the group size, launch256h).
    auto tile_coords = [&](int v, int& m0_, int& n0_) {
        const int logical = xcd_remap(v, ntiles);
        // ngroup < 0: M-super-panel order, the mirror image -- M tiles in groups of P = -ngroup; inside a group N-major / M-minor.
        // An XCD then sweeps ALL N tiles against P A panels (P x 256 rows x K, L2-resident after the first touch) before it moves
        // on: the activations leave HBM once, the weight panels are re-fetched per group from the Infinity Cache (which they
        // always fit).  Same arithmetic with the roles of the two tile dimensions swapped (branch-free: uniform selects).
        const bool mp = p.ngroup < 0;
        const int D = mp ? tiles_m : tiles_n, O = mp ? tiles_n : tiles_m;    // grouped dimension / the other one
        int G = mp ? -p.ngroup : p.ngroup;
        G = (G > 0 && G < D) ? G : D;
        int g = logical / (O * G);
        const int ng = (D + G - 1) / G;
        if (g > ng - 1) g = ng - 1;
        const int rem = logical - g * O * G;
        const int gsz = g == ng - 1 ? D - g * G : G;
        const int oth = rem / gsz, din = g * G + rem % gsz;
        m0_ = (mp ? din : oth) * BM;
        n0_ = (mp ? oth : din) * BN;
    };
    int v = blockIdx.x;
    int m0, n0;
    tile_coords(v, m0, n0);

    // ---- DMA sources: wave w moves pieces 4w..4w+3 (8 rows x 128 B) of whichever panel a half-stage carries
    const char* __restrict__ A = (const char*)p.A;
    const char* __restrict__ B = (const char*)p.B;
    // 32-bit byte offsets from the (uniform) operand bases: saddr + voffset addressing, 5 VGPRs instead of 10
    unsigned a0, a1, a2, a3, b0;
    // (ln = the lane id; the persistent epilogue passes its opaque copy so that nothing of this is hoisted out of the tile loop
    // and kept -- spilled -- across the K loop)
    auto set_sources = [&](int m0_, int n0_, int ln) {
        const int prow = ln >> 3;
        const int schunk = (ln & 7) ^ prow;
        auto arow = [&](int j) { int r = m0_ + wid * 32 + 8 * j + prow; return r < p.M ? r : p.M - 1; };
        a0 = (unsigned)arow(0) * (unsigned)p.lda * 2u + schunk * 16;
        a1 = (unsigned)arow(1) * (unsigned)p.lda * 2u + schunk * 16;
        a2 = (unsigned)arow(2) * (unsigned)p.lda * 2u + schunk * 16;
        a3 = (unsigned)arow(3) * (unsigned)p.lda * 2u + schunk * 16;
        b0 = (unsigned)(n0_ + wid * 32 + prow) * (unsigned)p.ldb * 2u + schunk * 16;
    };
    set_sources(m0, n0, lane);
    const unsigned bstep = 16u * (unsigned)p.ldb;   // 8 rows, bytes
    const int piece = wid * 4096;
#ifdef LEAF_DIAG_NODMA   // diagnostic only: no operand traffic at all (results are garbage) - isolates the MFMA + LDS-read loop
#define DMA16(src, dst) asm volatile("" ::"v"(src), "v"(dst))
#else
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
#endif
    // piece q (0..3) of half-stage u: u even = A panel of K tile u/2, u odd = B panel
    // (so = byte offset of the ring slot that half-stage lands in, kt = its K tile)
    // (A's k offset wraps after awrap K tiles -- GemmArgs::a_wrap, [A | A] operands; awrap >= nt otherwise: two scalar instructions per tile)
    const int awrap = p.a_wrap > 0 ? p.a_wrap : 0x7fffffff;
#define AKT(kt) ((kt) >= awrap ? (kt) - awrap : (kt))
#define ISSUE_A(so, kt, q) DMA16(A + (size_t)(AKT(kt) * (BK * 2)) + ((q) == 0 ? a0 : (q) == 1 ? a1 : (q) == 2 ? a2 : a3), smem + (so) + piece + (q) * 1024)
#define ISSUE_B(so, kt, q) DMA16(B + (size_t)((kt) * (BK * 2) + (q) * bstep) + b0, smem + (so) + piece + (q) * 1024)
#define ISSUE_HALF_A(so, kt) ISSUE_A(so, kt, 0); ISSUE_A(so, kt, 1); ISSUE_A(so, kt, 2); ISSUE_A(so, kt, 3);
#define ISSUE_HALF_B(so, kt) ISSUE_B(so, kt, 0); ISSUE_B(so, kt, 1); ISSUE_B(so, kt, 2); ISSUE_B(so, kt, 3);

    constexpr bool FOLD = (EPI == EPI_LNFOLD_T || EPI == EPI_LNFOLD_ACT_T);
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
#define MFMA_H2(P) MROW(P, 4, XSEL_4(P)) MROW(P, 5, XSEL_5(P)) MROW(P, 6, XSEL_6(P)) MROW(P, 7, XSEL_7(P))
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
#ifdef LEAF_DIAG_READ23   // diagnostic only (wrong results): activation-fragment rows 4-7 reuse the registers of rows 0-3, i.e. 8 instead
                          // of 12 ds_read_b128 per k-step = the LDS read bytes per MFMA of a 128 x 128 per-wave tile (256 B), on random data
#define XSEL_0(P) P##x0
#define XSEL_1(P) P##x1
#define XSEL_2(P) P##x2
#define XSEL_3(P) P##x3
#define XSEL_4(P) P##x0
#define XSEL_5(P) P##x1
#define XSEL_6(P) P##x2
#define XSEL_7(P) P##x3
#define RDX(P, n) if ((n) < 4) { P##x##n = LD(sa_ + (n) * 2048); }
#else
#define XSEL_0(P) P##x0
#define XSEL_1(P) P##x1
#define XSEL_2(P) P##x2
#define XSEL_3(P) P##x3
#define XSEL_4(P) P##x4
#define XSEL_5(P) P##x5
#define XSEL_6(P) P##x6
#define XSEL_7(P) P##x7
#define RDX(P, n) P##x##n = LD(sa_ + (n) * 2048);
#endif
#define MF(P, i, j) acc[i][j] = TT::mfma(P##w##j, XSEL_##i(P), acc[i][j]);
#define RDW(P, n) P##w##n = LD(sb_ + (n) * 2048);
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
        SB MROW(PREV, 7, XSEL_7(PREV)) SB IS3                                                              \
        SB MF(CUR, 0, 0) SB RDX(CUR, 4) SB MF(CUR, 0, 1) MF(CUR, 0, 2) SB RDX(CUR, 5) SB MF(CUR, 0, 3)        \
        SB MF(CUR, 1, 0) SB RDX(CUR, 6) SB MF(CUR, 1, 1) MF(CUR, 1, 2) SB RDX(CUR, 7) SB MF(CUR, 1, 3)        \
        SB MROW(CUR, 2, XSEL_2(CUR)) MROW(CUR, 3, XSEL_3(CUR)) SB                                                    \
    }
#endif
#define NOP_

    const int nt = p.K / BK;   // K tiles, >= 4 (host-checked)
    // half-stage u lives in ring slot u % 5; the offsets below are uniform and advance by two slots per K tile
#define ADV(x) { x += 2 * HALF; if (x >= RING) x -= RING; }
    bool first = true;        // PERSIST: first output tile of this workgroup (nothing prefetched yet)
    bool counted = false;     // PERSIST: the previous epilogue issued exactly NSTORE stores behind the prefetch (full tile)
  for (;;) {                  // output tiles of this workgroup (one pass unless PERSIST)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sa = 0, sb = HALF;               // slots of (A, B) of the tile being multiplied: half-stages 2T, 2T+1
    int i0 = 3 * HALF, i1 = 4 * HALF;    // slots of the half-stages requested during tile T: 2T+3 (B), 2T+4 (A)
    STAMP(0)
    // LN folding: thread t < 256 fetches (mean, rstd) of A row m0 + t now (2 registers held across the K loop), so the epilogue
    // waits for no memory
    float2 my_rowstat = float2{0.f, 0.f};
    if constexpr (!PERSIST) {
        if constexpr (FOLD) {
            if (tid < BM && m0 + tid < p.M) my_rowstat = p.rowstat[m0 + tid];
        }
        ISSUE_HALF_A(0, 0) ISSUE_HALF_B(HALF, 0) ISSUE_HALF_A(2 * HALF, 1)
        // ---- tile 0
        SYNC_TILE(4)
    } else {
        // A0 / B0 of this tile: requested just now (first tile) or before the previous tile's epilogue, whose NSTORE output
        // stores were issued after them (vmcnt completes in order: "all but the youngest NSTORE" = the prefetch has landed)
        constexpr int NSTORE = 16;
        if (first) { ISSUE_HALF_A(0, 0) ISSUE_HALF_B(HALF, 0) }
        if (!first && counted) { SYNC_TILE(16) } else { SYNC_TILE(0) }
        static_assert(NSTORE == 16, "the counted wait above");
        // behind the barrier: every wave has left the previous epilogue, slot 2 (row table) and slots 3-4 (staging) are free
        ISSUE_HALF_A(2 * HALF, 1)
        if constexpr (FOLD) {
            if (tid < BM && m0 + tid < p.M) my_rowstat = p.rowstat[m0 + tid];
        }
    }
    STAMP(1)
    READ_FRAGS(G, sa, sb, fo0)
    SB ISSUE_B(i0, 1, 0); ISSUE_B(i0, 1, 1); SB
    MROW(G, 0, Gx0) MROW(G, 1, Gx1) SB ISSUE_B(i0, 1, 2); ISSUE_B(i0, 1, 3); SB MROW(G, 2, Gx2) MROW(G, 3, Gx3)
    KSTEP(G, F, sa, sb, fo1, ISSUE_A(i1, 2, 0);, ISSUE_A(i1, 2, 1);, ISSUE_A(i1, 2, 2);, ISSUE_A(i1, 2, 3);)
    ADV(sa) ADV(sb) ADV(i0) ADV(i1)
    // ---- tiles 1 .. nt-3: request B of tile T+1 and A of tile T+2
    int T = 1;
    for (; T <= nt - 3; ++T) {
        SYNC_TILE(4)
        KSTEP(F, G, sa, sb, fo0, ISSUE_B(i0, T + 1, 0);, ISSUE_B(i0, T + 1, 1);, ISSUE_B(i0, T + 1, 2);, ISSUE_B(i0, T + 1, 3);)
        KSTEP(G, F, sa, sb, fo1, ISSUE_A(i1, T + 2, 0);, ISSUE_A(i1, T + 2, 1);, ISSUE_A(i1, T + 2, 2);, ISSUE_A(i1, T + 2, 3);)
        ADV(sa) ADV(sb) ADV(i0) ADV(i1)
    }
    // ---- tile nt-2: only the B panel of the last tile is left to request
    STAMP(2)
    SYNC_TILE(4)
    KSTEP(F, G, sa, sb, fo0, ISSUE_B(i0, T + 1, 0);, ISSUE_B(i0, T + 1, 1);, ISSUE_B(i0, T + 1, 2);, ISSUE_B(i0, T + 1, 3);)
    KSTEP(G, F, sa, sb, fo1, NOP_, NOP_, NOP_, NOP_)
    ADV(sa) ADV(sb)
    // ---- tile nt-1
    SYNC_TILE(0)
    KSTEP(F, G, sa, sb, fo0, NOP_, NOP_, NOP_, NOP_)
    KSTEP(G, F, sa, sb, fo1, NOP_, NOP_, NOP_, NOP_)
    SB MFMA_H2(F) SB
    STAMP(3)
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
    // staging slices / row table: above the whole ring's first 128 KiB normally; PERSIST keeps slots 0-1 for the prefetch
    constexpr int TABLE_OFF = PERSIST ? 2 * HALF : 8 * SLICE;
    if constexpr (FOLD) {
        // (mean, rstd) of this tile's 256 A rows (loaded before the K loop) -> a 2-KiB table
        if (tid < BM) *(float2*)(smem + TABLE_OFF + tid * 8) = my_rowstat;
        __syncthreads();
    }
    char* sl = PERSIST ? smem + 3 * HALF + wid * 8192 : smem + wid * SLICE;
    // epilogue addressing derives from an OPAQUE copy of the lane id: in the persistent form the compiler would otherwise hoist
    // these per-lane offsets out of the tile loop and keep them in VGPRs across the K loop (spills)
    int elane = lane;
    if constexpr (PERSIST) asm volatile("" : "+v"(elane));
    const int fq = elane >> 4, efrow = elane & 15;
    const int nb = n0 + wn * 64;          // first column of this wave's sub-tile
    const int mb = m0 + wm * 128;         // first row
    float4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        bias4[j] = p.bias ? *(const float4*)(p.bias + nb + 16 * j + 4 * fq) : float4{0.f, 0.f, 0.f, 0.f};

    float4 s4[4];   // LN folding: row sums of the gamma-scaled weights for this lane's columns
    if constexpr (FOLD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s4[j] = *(const float4*)(p.ln_s + nb + 16 * j + 4 * fq);
    }

    // (mean, rstd) of this lane's 8 rows, out of the table (before the prefetch below: see the note on LDS accesses there)
    float2 rs_all[8];
    if constexpr (FOLD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rs_all[i] = *(const float2*)(smem + TABLE_OFF + (wm * 128 + 16 * i + efrow) * 8);
    }

    // PERSIST: request the next tile's first half-stages now; the epilogue below then only stores.  While those LDS-DMAs are
    // in flight every compiler-visible LDS load / store would be preceded by s_waitcnt vmcnt(0) (the compiler cannot prove
    // that the staging slices and the DMA's slots are disjoint), so the persistent epilogue stages through inline-asm
    // ds_write / ds_read with its own lgkmcnt waits.  The bias / s loads above
    // are retired first (an ordinary load pending beside LDS-DMAs makes the compiler wait for vmcnt(0) at its first use).
    int v_next = v, m0_next = m0, n0_next = n0;
    bool has_next = false;
    if constexpr (PERSIST) {
        v_next = v + (int)gridDim.x;
        has_next = v_next < ntiles;
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0) (bias / s / row-table loads); on BOTH paths (else the join waits)
        if (has_next) {
            tile_coords(v_next, m0_next, n0_next);
            set_sources(m0_next, n0_next, elane);
            ISSUE_HALF_A(0, 0) ISSUE_HALF_B(HALF, 0)
        }
    }

    if constexpr (EPI == EPI_STORE_T || EPI == EPI_ACT_T || EPI == EPI_ACTGRAD_T || FOLD) {
        // two passes of 64 rows x 64 cols of 16-bit: LDS rows of 128 B, 16-B chunks XOR-swizzled by (row & 7)
        // ACTC: std::integral_constant<int, -1 | ACT_GELU | ACT_QUICKGELU> - the activation is fixed at compile time inside
        // the element loops (a run-time id there costs one branch per element and serialises the transcendental chains)
        auto stage16 = [&](int pass, auto ACTC) {
            constexpr int ACT = decltype(ACTC)::value;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = 4 * pass + ii;
                const int row = 16 * ii + efrow;
                // two adjacent columns per operation: packed-f32 VALU ops (v_pk_fma / v_pk_mul / v_pk_add_f32), the same
                // roundings per element as the scalar forms of the register-direct epilogues (gemm_epilogue.h)
                f32x2 v[4][2];
                if constexpr (FOLD) {
                    const float2 rs = rs_all[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j][0] = lnfold_apply2(f32x2{acc[i][j][0], acc[i][j][1]}, rs.x, rs.y, f32x2{s4[j].x, s4[j].y}, f32x2{bias4[j].x, bias4[j].y});
                        v[j][1] = lnfold_apply2(f32x2{acc[i][j][2], acc[i][j][3]}, rs.x, rs.y, f32x2{s4[j].z, s4[j].w}, f32x2{bias4[j].z, bias4[j].w});
                    }
                } else if constexpr (EPI == EPI_ACTGRAD_T) {
                    // backward of the MLP activation: C16 = acc * act'(pre), pre = the forward's stashed pre-activations (aux,
                    // 16-bit of the forward dtype, same shape as C); rows past M are clamped for the load and never stored
                    int mr = mb + 64 * pass + row;
                    mr = mr < p.M ? mr : p.M - 1;
                    uint2 u[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = *(const uint2*)((const u16*)p.aux + (size_t)mr * p.ldc + nb + 16 * j + 4 * fq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float pre[4];
                        if (p.aux_f16) unpack4<F16>(u[j], pre); else unpack4<BF16>(u[j], pre);
                        v[j][0] = f32x2{acc[i][j][0] * act_bwd(pre[0], p.act), acc[i][j][1] * act_bwd(pre[1], p.act)};
                        v[j][1] = f32x2{acc[i][j][2] * act_bwd(pre[2], p.act), acc[i][j][3] * act_bwd(pre[3], p.act)};
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j][0] = f32x2{acc[i][j][0], acc[i][j][1]} + f32x2{bias4[j].x, bias4[j].y};
                        v[j][1] = f32x2{acc[i][j][2], acc[i][j][3]} + f32x2{bias4[j].z, bias4[j].w};
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j][0] = act_fwd2<ACT>(v[j][0]);     // 8 independent pairs
                    v[j][1] = act_fwd2<ACT>(v[j][1]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 2 * j + (fq >> 1);
                    const uint2 pk = uint2{TT::pack2(v[j][0]), TT::pack2(v[j][1])};
                    char* dst = sl + row * 128 + ((c ^ (row & 7)) << 4) + (fq & 1) * 8;
                    if constexpr (PERSIST) {
                        // the next tile's LDS-DMAs are in flight: a compiler-visible LDS store here makes hipcc wait vmcnt(0)
                        // for them (possible write-after-write on LDS; slots 0-1 and the slices are disjoint) -> opaque store
                        typedef __attribute__((address_space(3))) char lds_char_t;
                        const unsigned off = (unsigned)(uintptr_t)(lds_char_t*)dst;
                        const unsigned long long pk64 = __builtin_bit_cast(unsigned long long, pk);
                        asm volatile("ds_write_b64 %0, %1" ::"v"(off), "v"(pk64) : "memory");
                    } else {
                        *(uint2*)dst = pk;
                    }
                }
            }
            if constexpr (PERSIST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the flush below reads what was just staged
        };
        typedef std::integral_constant<int, -1> NoAct;
        typedef std::integral_constant<int, ACT_GELU> Gelu;
        typedef std::integral_constant<int, ACT_QUICKGELU> QuickGelu;
        auto flush16 = [&](int pass, u16* dst) {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t fv[8];
            if constexpr (PERSIST) {
                // all eight reads of the pass in flight at once (rows 8 it + lane / 8: 1 KiB apart), each store then waits for
                // its own read only (LDS returns a wave's reads in order: lgkmcnt(7 - it))
                typedef __attribute__((address_space(3))) char lds_char_t;
                const unsigned off = (unsigned)(uintptr_t)(lds_char_t*)(sl + (elane >> 3) * 128 + ((elane & 7) << 4));
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                             "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
                             "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168"
                             : "=&v"(fv[0]), "=&v"(fv[1]), "=&v"(fv[2]), "=&v"(fv[3]), "=&v"(fv[4]), "=&v"(fv[5]), "=&v"(fv[6]), "=&v"(fv[7])
                             : "v"(off) : "memory");
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 8 * it + (elane >> 3), pc = elane & 7;
                if constexpr (PERSIST) {
                    switch (it) {   // the wait is tied to the value it releases ("+v"): its store cannot move above it
                        case 0: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fv[0]) :: "memory"); break;
                        case 1: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fv[1]) :: "memory"); break;
                        case 2: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fv[2]) :: "memory"); break;
                        case 3: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fv[3]) :: "memory"); break;
                        case 4: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fv[4]) :: "memory"); break;
                        case 5: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fv[5]) :: "memory"); break;
                        case 6: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fv[6]) :: "memory"); break;
                        default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fv[7]) :: "memory"); break;
                    }
                } else {
                    fv[it] = *(const u32x4_t*)(sl + row * 128 + (pc << 4));
                }
                const int m = mb + 64 * pass + row;
#ifdef LEAF_DIAG_NOQKVSTORE   // diagnostic only (garbage results): the QKV GEMM computes and stages its tile but never stores it -- with
                              // LEAF_DIAG_ATTN_L2 an upper bound on what fusing QKV GEMM -> attention could save
                if constexpr (EPI == EPI_LNFOLD_T) { asm volatile("" ::"v"(fv[it]), "v"(m)); } else
#endif
                if (m < p.M)
                    __builtin_nontemporal_store(fv[it], (u32x4_t*)(dst + (size_t)m * p.ldc + nb + ((pc ^ (row & 7)) << 3)));
            }
        };
        constexpr bool ACTIVE = (EPI == EPI_ACT_T || EPI == EPI_LNFOLD_ACT_T);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (EPI == EPI_ACT_T && p.aux) {   // training forward: pre-activation stash first
                stage16(pass, NoAct());
                flush16(pass, (u16*)p.aux);
            }
            if (!ACTIVE) stage16(pass, NoAct());
            else if (p.act == ACT_QUICKGELU) stage16(pass, QuickGelu());
            else stage16(pass, Gelu());
            flush16(pass, (u16*)p.C);
        }
    } else if constexpr (EPI == EPI_RESID_LN8) {
        // the residual stream in 16 + 8 bits (common.h resid_lo4): the passes of the fp32 epilogue below, but the residual rows come
        // from -- and the finished rows go back to -- their 16-bit copy (8 B per lane and row) and remainder bytes (4 B): 8 d bytes
        // per row cross HBM in the out-projection instead of 12 d, and no fp32 row is written at all.
        // Addresses: 32-bit byte offsets from the two (uniform) bases -- lane part (its first row, its column for each of the four
        // values of row & 15 >> 2) once per tile, the row steps of the passes are uniform adds; the launcher checks both spans < 4 GiB.
        const unsigned lr = (unsigned)elane >> 4, pc = (unsigned)elane & 15;
        unsigned o16[4], o8[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned col = (unsigned)nb + ((pc ^ (4 * k + lr)) << 2);
            o16[k] = (lr * (unsigned)p.ldx16 + col) * 2u;
            o8[k] = lr * (unsigned)p.ldc + col;
        }
        // (uniform row bases: scalar arithmetic; the lane offsets above never change)
        const char* __restrict__ x16b = (const char*)p.x16 + (size_t)mb * p.ldx16 * 2;
        const char* __restrict__ lo8b = (const char*)p.C + (size_t)mb * p.ldc;
        const bool full = mb + 128 <= p.M;     // wave-uniform: every row of this wave's 128 exists (all but the last M tile)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            uint2 rh[8];
            unsigned rl[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int step = 32 * pass + 4 * it;
                const bool ok = full || mb + step + (int)lr < p.M;
                rh[it] = ok ? *(const uint2*)(x16b + (size_t)step * p.ldx16 * 2 + o16[it & 3]) : uint2{0u, 0u};
                rl[it] = ok ? *(const unsigned*)(lo8b + (size_t)step * p.ldc + o8[it & 3]) : 0u;
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * pass + ii;
                const int row = 16 * ii + efrow;
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
                const int row = 4 * it + (int)lr;
                float4 v = *(const float4*)(sl + row * 256 + (pc << 4));
                const int step = 32 * pass + 4 * it;
                const float4 r = resid_decode4<TT>(rh[it], rl[it]);
                v.x = __builtin_fmaf(r.x, 1.f, v.x); v.y = __builtin_fmaf(r.y, 1.f, v.y);
                v.z = __builtin_fmaf(r.z, 1.f, v.z); v.w = __builtin_fmaf(r.w, 1.f, v.w);
                const float gs = row16_sum(lnfold_sum4(v.x, v.y, v.z, v.w));
                const float gq = row16_sum(lnfold_dev4(v.x, v.y, v.z, v.w, gs * (1.0f / 64.0f)));
                if (full || mb + step + (int)lr < p.M) {
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    const uint2 hi = pack4<TT>(v.x, v.y, v.z, v.w);
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x2_t, hi), (u32x2_t*)(const_cast<char*>(x16b) + (size_t)step * p.ldx16 * 2 + o16[it & 3]));
                    __builtin_nontemporal_store(resid_lo4<TT>(v.x, v.y, v.z, v.w, hi), (unsigned*)(const_cast<char*>(lo8b) + (size_t)step * p.ldc + o8[it & 3]));
                    if (pc == 0) p.stat_out[(size_t)(nb >> 6) * p.stat_ld + (mb + step + (int)lr)] = float2{gs, gq};
                }
            }
        }
    } else {
        // fp32 outputs: four passes of 32 rows x 64 cols: LDS rows of 256 B, 16-B chunks XOR-swizzled by (row & 15)
        constexpr bool RESID = (EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN);
        const float beta = RESID ? 1.f : p.beta;
        const float* rsrc = (RESID && p.aux) ? (const float*)p.aux : (const float*)p.C;   // out-of-place residual
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            float4 res[8];
            if (beta != 0.f) {   // fetch the residual rows of this pass first: 8 coalesced 16-B loads in flight
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = 4 * it + (elane >> 4), pc = elane & 15;
                    const int m = mb + 32 * pass + row;
                    res[it] = m < p.M ? *(const float4*)(rsrc + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2))
                                      : float4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * pass + ii;
                const int row = 16 * ii + efrow;
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
                const int row = 4 * it + (elane >> 4), pc = elane & 15;
                float4 v = *(const float4*)(sl + row * 256 + (pc << 4));
                const int m = mb + 32 * pass + row;
                if (beta != 0.f) {
                    v.x = __builtin_fmaf(res[it].x, beta, v.x); v.y = __builtin_fmaf(res[it].y, beta, v.y);
                    v.z = __builtin_fmaf(res[it].z, beta, v.z); v.w = __builtin_fmaf(res[it].w, beta, v.w);
                }
#ifdef LEAF_DIAG_NOF32STORE   // diagnostic only (garbage results): the residual GEMM with statistics never stores its fp32 rows -- the byte
                              // volume of a residual stream kept as the 16-bit copy + an 8-bit extension (8 d instead of 12 d per row)
                if (m < p.M && EPI != EPI_RESID_LN) {
#else
                if (m < p.M) {
#endif
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, (f32x4_t*)((float*)p.C + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2)));
                }
                if constexpr (EPI == EPI_RESID_LN) {
                    // LN folding: 16-bit copy of the finished row segment + (sum, M2) of its 64 columns (the 16 lanes of the row)
                    const float gs = row16_sum(lnfold_sum4(v.x, v.y, v.z, v.w));
                    const float gq = row16_sum(lnfold_dev4(v.x, v.y, v.z, v.w, gs * (1.0f / 64.0f)));
                    if (m < p.M) {
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#ifdef LEAF_X16_REGULAR   // A/B build: the 16-bit copy (re-read by the next GEMM) through the cache hierarchy instead of around it
                        *(u32x2_t*)((u16*)p.x16 + (size_t)m * p.ldx16 + nb + ((pc ^ (row & 15)) << 2)) =
                            __builtin_bit_cast(u32x2_t, pack4<TT>(v.x, v.y, v.z, v.w));
#else
                        __builtin_nontemporal_store(__builtin_bit_cast(u32x2_t, pack4<TT>(v.x, v.y, v.z, v.w)),
                                                    (u32x2_t*)((u16*)p.x16 + (size_t)m * p.ldx16 + nb + ((pc ^ (row & 15)) << 2)));
#endif
                        if (pc == 0) p.stat_out[(size_t)(nb >> 6) * p.stat_ld + m] = float2{gs, gq};
                    }
                }
            }
        }
    }
    STAMP(4)
    if constexpr (!PERSIST) break;
    if (!has_next) break;
    // exactly NSTORE stores behind the prefetch: a full tile without the optional pre-activation stash
    counted = (m0 + BM <= p.M) && !(EPI == EPI_ACT_T && p.aux);
#ifdef LEAF_DIAG_NOQKVSTORE
    if constexpr (EPI == EPI_LNFOLD_T) counted = false;
#endif
    first = false;
    v = v_next; m0 = m0_next; n0 = n0_next;
  }
#undef ADV
#undef DMA16
#undef ISSUE_A
#undef AKT
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
#undef XSEL_0
#undef XSEL_1
#undef XSEL_2
#undef XSEL_3
#undef XSEL_4
#undef XSEL_5
#undef XSEL_6
#undef XSEL_7
#undef NOP_
#undef SB
#undef LD
}

// N tiles per group: minimise the bytes that miss L2.  A group's B panels (G x 256 rows x K) stay resident in an XCD's L2
// when they are small (<= ~2 MiB); the A operand is then streamed once per group.  With everything in one group (the
// M-major order) B is re-fetched for every round of 32 concurrent tiles per XCD once it no longer fits beside A.
// LEAF_GEMM_NGROUP = n forces a size (0 = never group).
static int pick_ngroup(const GemmArgs& p) {
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("LEAF_GEMM_NGROUP"); forced = e ? atoi(e) : -1; }
    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    // LEAF_GEMM_MPANEL = P: M-super-panel order with P M tiles per group for every launch with more than one N tile and more
    // than P M tiles (returned as -P); 0 / unset = the N-group order below
    static int mpanel = -2;
    if (mpanel == -2) { const char* e = getenv("LEAF_GEMM_MPANEL"); mpanel = e ? atoi(e) : 0; }
    if (mpanel > 0 && tiles_n > 1 && tiles_m > mpanel) return -mpanel;
    if (forced >= 0) return forced < tiles_n ? forced : 0;
    const double a_bytes = (double)p.M * p.K * 2, b_tile = (double)BN * p.K * 2;
    const double rounds = (double)tiles_m * tiles_n / 256.0;    // rounds of 32 tiles per XCD
    int best = 0;
    double best_cost = -1;
    for (int G = 1; G <= tiles_n; ++G) {
        const int ng = (tiles_n + G - 1) / G;
        const bool resident = G * b_tile <= 2.0 * 1024 * 1024;
        const double b_cost = resident ? b_tile * tiles_n * 8 : b_tile * tiles_n * 8 * (rounds > 1 ? rounds : 1);
        const double cost = a_bytes * ng + b_cost;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = G; }
    }
    return best >= tiles_n ? 0 : best;
}

template <class TT>
hipError_t launch256h(const GemmArgs& p_in, int epi, hipStream_t s) {
    GemmArgs p = p_in;
    p.ngroup = pick_ngroup(p);
    const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    // persistent form for the 16-bit-output epilogues (QKV, c_fc): one workgroup per CU; LEAF_GEMM_PERSIST=0 disables (A/B)
    static int persist_on = -1, ncu = 0;
    if (persist_on < 0) {
        const char* e = getenv("LEAF_GEMM_PERSIST");
        persist_on = (e && e[0] == '0') ? 0 : 1;
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        ncu = ncu >= 8 ? ncu / 8 * 8 : 8;      // a multiple of the 8 XCDs keeps every workgroup's tiles on one XCD
    }
#define LEAF_LAUNCH(E, PS, GRID)                                                                             \
    {                                                                                                        \
        static bool attr_done = false;                                                                       \
        if (!attr_done) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt256_half_kernel<TT, E, PS>,                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, RING);                     \
            attr_done = true;                                                                                \
        }                                                                                                    \
        hipLaunchKernelGGL((gemm_nt256_half_kernel<TT, E, PS>), dim3(GRID), dim3(512), RING, s, p);          \
    }
#define LEAF_CASE(E)  case E: LEAF_LAUNCH(E, false, ntiles) break;
#define LEAF_CASE_P(E)                                                                                       \
    case E:                                                                                                  \
        if (persist_on && ntiles > ncu) LEAF_LAUNCH(E, true, ncu)                                            \
        else LEAF_LAUNCH(E, false, ntiles)                                                                   \
        break;
    switch (epi) {
        LEAF_CASE_P(EPI_STORE_T)
        LEAF_CASE_P(EPI_ACT_T)
        LEAF_CASE(EPI_RESID_F32)
        LEAF_CASE(EPI_STORE_F32)
        LEAF_CASE(EPI_ACTGRAD_T)
        LEAF_CASE_P(EPI_LNFOLD_T)
        LEAF_CASE_P(EPI_LNFOLD_ACT_T)
        LEAF_CASE(EPI_RESID_LN)
        LEAF_CASE(EPI_RESID_LN8)
        default: return hipErrorInvalidValue;
    }
#undef LEAF_CASE
#undef LEAF_CASE_P
#undef LEAF_LAUNCH
    return hipGetLastError();
}

}  // namespace

int leaf_gemm256h_pick_ngroup(const GemmArgs& p) { return pick_ngroup(p); }

// fewest 256^2 tiles for which this kernel is dispatched (tuned default below; leaf_debug_gemm_min_tiles / LEAF_GEMM256H_MIN_TILES)
static int g_min_tiles = -1;
void leaf_gemm256h_set_min_tiles(int n) { g_min_tiles = n; }

bool leaf_gemm256h_eligible(const GemmArgs& p, int epi) {
    if (g_min_tiles < 0) { const char* e = getenv("LEAF_GEMM256H_MIN_TILES"); g_min_tiles = e ? atoi(e) : 128; }
    const long tiles = (long)((p.M + BM - 1) / BM) * (p.N / BN);
    // the DMA sources are 32-bit byte offsets from the operand bases (saddr + voffset): both operands must span < 4 GiB
    const bool fits32 = (unsigned long long)p.M * p.lda * 2ull < (1ull << 32) && (unsigned long long)p.N * p.ldb * 2ull < (1ull << 32);
    // EPI_RESID_LN8 addresses its 16-bit rows and remainder bytes by 32-bit byte offsets too
    const bool fits8 = epi != EPI_RESID_LN8 || ((unsigned long long)p.M * p.ldx16 * 2ull < (1ull << 32) && (unsigned long long)p.M * p.ldc < (1ull << 32));
    // a_wrap: A is [M, 64 a_wrap], read K / (64 a_wrap) times (at most twice: one wrap per tile walk)
    const bool wrap_ok = p.a_wrap == 0 || (p.a_wrap > 0 && p.K == 2 * BK * p.a_wrap && p.lda >= BK * p.a_wrap);
    return p.N % BN == 0 && tiles >= g_min_tiles && p.K % BK == 0 && p.K >= 4 * BK && p.ldc % 8 == 0 && fits32 && fits8 && wrap_ok;
}

hipError_t leaf_launch_gemm256h(const GemmArgs& p, int dtype, int epi, hipStream_t s) {
    return dtype == LEAF_F16 ? launch256h<F16>(p, epi, s) : launch256h<BF16>(p, epi, s);
}
