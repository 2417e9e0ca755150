// QKV projection -> causal attention in ONE launch: the q|k|v rows of the scoring passes never reach HBM.
//
// Reference op: nn.MultiheadAttention = in_proj -> per-head softmax(q k^T / sqrt(64) + causal mask) v (-> out_proj, a separate
// GEMM here), src/open_clip/transformer.py:225,239-252, with the LayerNorm in front of it folded into the in_proj GEMM (lnfold.h).
//
// Round 3 ran this as two kernels around a [rows, 3d] 16-bit buffer: the LN-folded QKV GEMM wrote 6d bytes per row (541 MB per
// launch of a ViT-L scoring stage) that attn_fwd_kernel read straight back.  A diagnostic build that never stores that buffer
// and lets the attention read an L2-resident window bounded the round trip at 3.5-4.0 of 41.4 ms per step
// (profiles/r04_run1_tileorder_fusebound.txt).  Here a workgroup computes, for ONE head h and a 256-row M tile made of WHOLE
// sequences (packed rows, RowMap), the tile's [q_h | k_h | v_h] = 192 columns with the half-stage LDS-DMA ring of gemm256h.hip
// (same K order, same MFMA, same LN-fold epilogue arithmetic: a row has the same bits as from the stand-alone GEMM), stages
// them as fp16 in the now idle ring (Q, K, V: 32 KiB each), pulls the cached prefix K/V of the tile's <= 3 clean captions into
// LDS by LDS-DMA (once per tile instead of once per candidate), and then its eight waves run attn_fwd_kernel's MFMA body per
// sequence with every operand out of LDS.  Only the attention output (2d bytes per row) is written.
//
// LDS map (160 KiB): [0, 96 K) ring slots 0-2 = Q | K | V staging, 256 rows x 128 B each (after the K loop); [96 K, 156 K) up to
// three caption images (K rows then V rows, caprows x 128 B each; caprows = the launch's longest sequence rounded up to 16, <= 80);
// [156 K, 160 K) row statistics, sequence table, a zero line.
//
// What was tried and not kept: 128-row tiles with two 4-wave workgroups per CU (operand traffic of a
// 128 x 192 tile exceeds what a CU takes in from L2; slower), attention work dealt per (sequence, query tile) through an LDS-atomic
// list (the stage is instruction bound, not idle; slower), a persistent form (no LDS left to prefetch the next tile into while the
// attention stage holds Q | K | V and the caption images).
#include "common.h"
#include "kernels.h"
#include "lnfold.h"

namespace {

constexpr int BM = 256, BK = 64, NSLOT = 5, HD = 64;
constexpr int HALF = BM * BK * 2;            // 32 KiB: one ring slot (an A panel; a B panel uses 24 KiB of it)
constexpr int RING = NSLOT * HALF;           // 160 KiB
constexpr int NCAP = LEAF_QKVATTN_NCAP, CAPROWS = LEAF_QKVATTN_CAPROWS;
constexpr int MAXT = CAPROWS / 16;           // 16-row key / query tiles per sequence
constexpr int CAP_OFF = 3 * HALF;
constexpr int MISC_OFF = CAP_OFF + NCAP * 2 * CAPROWS * 128;   // images: p.ncap x (K rows, then V rows) of p.caprows rows each
constexpr int STAT_OFF = MISC_OFF;           // (2 KiB unused: the row statistics go straight from global memory to registers)
constexpr int SEQ_OFF = STAT_OFF + 2048;     // u32[256]: row | len << 9 | prefix << 16 | slot << 23 per sequence of the tile
constexpr int EOT_OFF = SEQ_OFF + 1024;      // u8[256]: pooled position per sequence (last-layer mode)
constexpr int ZERO_OFF = EOT_OFF + 256;      // 128 zero bytes: V rows beyond a sequence's end
static_assert(ZERO_OFF + 128 <= RING, "LDS map");
static_assert(MAXT * 16 == CAPROWS && CAPROWS <= 96, "caption image rows");

// In-kernel stamps (diagnostic builds only: -DLEAF_GEMM_STAMPS): s_memtime at phase boundaries, one 8-slot record per workgroup
#ifdef LEAF_GEMM_STAMPS
#ifdef LEAF_QA_PHASES
constexpr int STAMP_SLOTS = 16;
#else
constexpr int STAMP_SLOTS = 8;
#endif
#define STAMP(i)                                                                                          \
    if (p.stamps && tid == 0) {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        ((unsigned long long*)p.stamps)[(size_t)blockIdx.x * STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
#else
#define STAMP(i)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((address_space(3))) char lds_char_t;

__device__ __forceinline__ int lds_off_h(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int vswz(int row) { return 2 * ((row >> 1) & 3); }     // attention.hip's V image swizzle

// x[l] (op) x[l ^ 16] (op) x[l ^ 32] (op) x[l ^ 48] for the four 16-lane rows of a wave on the gfx950 row / half swaps instead of
// __shfl_xor's ds_bpermute (an LDS round trip each, four per query tile, on the critical path of a wave that has one partner to
// hide behind).  v_permlane16_swap_b32 a, b: odd rows of a <-> even rows of b;  v_permlane32_swap_b32 a, b: lanes 32-63 of a <->
// lanes 0-31 of b.  With a = b = x beforehand every lane then holds (its pair's low member, its pair's high member): the same
// two operands as x and __shfl_xor(x, 16 | 32), in an order that max and the (commutative) IEEE add do not see -- same bits.
// (Inline asm with the hazard pad inside: the builtin form folds max(r[0], r[1]) of a self-swap away.)
__device__ __forceinline__ void swap_rows16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_half32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows4_max(float m) {
    float a = m, b = m;
    swap_rows16(a, b);
    // (v_max_f32 in asm: behind the asm swaps the compiler would canonicalise both inputs of an fmaxf first, two more instructions
    // per step; the scores are never NaN)
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    a = m; b = m;
    swap_half32(a, b);
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ float rows4_sum(float x) {
    float a = x, b = x;
    swap_rows16(a, b);
    x = a + b;
    a = x; b = x;
    swap_half32(a, b);
    return a + b;
}

template <class TT>
__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(QkvAttnArgs p) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    // work item = (M tile, head); the heads of a tile are consecutive logical ids, an XCD owns a contiguous range: the tile's A
    // panel is fetched from beyond L2 once and shared by its 12-20 heads
    int tile, h;
    if (p.hsplit <= 1) {
        const int logical = xcd_remap(blockIdx.x, p.n_tiles * p.heads);
        tile = logical / p.heads;
        h = logical - tile * p.heads;
    } else {
        // head-split order: 8 / hsplit contiguous M ranges, each owned by hsplit XCDs that run heads / hsplit heads of every tile
        // of the range -- an XCD's live weight panels shrink to heads / hsplit (ViT-L, 2: 1.8 MB of its 4-MiB L2) at the price
        // of the A panel being fetched by hsplit XCDs
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int R = 8 / p.hsplit, r = x / p.hsplit, hg = x - r * p.hsplit, HG = p.heads / p.hsplit;
        const int q = p.n_tiles / R, rem = p.n_tiles - q * R;
        const int base = r * q + (r < rem ? r : rem), cnt = q + (r < rem ? 1 : 0);
        const int t = slot / HG;
        if (t >= cnt) return;
        tile = base + t;
        h = hg * HG + (slot - t * HG);
    }
    // the weight panel of K tile 0 needs nothing but the head: on its way before the first scalar load has come back
    // (wave w moves pieces 3w..3w+2 of a B panel; B panel row r (0..191) = weight row (r / 64) * d + h * 64 + r % 64 of [q; k; v])
    unsigned b0, b1, b2;
    {
        const int prow = lane >> 3;
        const int schunk = (lane & 7) ^ prow;
        auto brow = [&](int q) { const int pc = 3 * wid + q; return (pc >> 3) * p.d + h * HD + (pc & 7) * 8 + prow; };
        b0 = (unsigned)brow(0) * (unsigned)p.ldb * 2u + schunk * 16;
        b1 = (unsigned)brow(1) * (unsigned)p.ldb * 2u + schunk * 16;
        b2 = (unsigned)brow(2) * (unsigned)p.ldb * 2u + schunk * 16;
        const char* Bw = (const char*)p.B;
        char* dst = smem + HALF + wid * 3072;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(Bw + b0), (lds_void_t*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(Bw + b1), (lds_void_t*)(dst + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(Bw + b2), (lds_void_t*)(dst + 2048), 16, 0, 0);
    }
    // the tile's sequences (indices inside the launch) and its first row (launch-local), cut on the host: ONE round trip of scalar
    // loads before the first operand DMA can go out (looking the row up in cu[] behind the sequence index was a second one)
    const int2 tb = ((const int2*)p.tile_seq)[tile], te = ((const int2*)p.tile_seq)[tile + 1];
    const int s_b = tb.x, s_e = te.x, r0 = tb.y;
    const int nseq = s_e - s_b;
    const int d = p.d;
    const int cap_img = p.caprows * 128;         // one K (or V) image of a caption

    // ---- per-sequence facts and row statistics: requested now, consumed in the epilogue (registers across the K loop)
    // (raw loaded values only: any arithmetic on them here would make the compiler wait for these loads -- and for the weight DMA
    // above -- before the activation DMAs below can go out)
    int my_c0 = 0, my_c1 = 0, my_pf = 0, my_eot = 0;
    if (tid < nseq) {
        const int sg = p.map.s0 + s_b + tid;
        if (p.map.cu) { my_c0 = p.map.cu[sg]; my_c1 = p.map.cu[sg + 1]; }
        else { my_c0 = sg * p.map.ctx; my_c1 = my_c0 + p.map.ctx; }
        if (p.map.prefix) my_pf = p.map.prefix[sg];
        if (p.eot_pos) my_eot = p.eot_pos[s_b + tid];
    }
    // caption images: rows of the tile's captions inside the cache (uniform: scalar loads)
    int cap_row0[NCAP], cap_rows[NCAP];
    {
        const int first = p.map.s0 + s_b > p.map.group_off ? p.map.s0 + s_b : p.map.group_off;
        const int last = p.map.s0 + s_e - 1;
#pragma unroll
        for (int c = 0; c < NCAP; ++c) {
            cap_row0[c] = 0; cap_rows[c] = 0;
            if (c < p.ncap && p.map.prefix && p.kv_base && last >= p.map.group_off) {
                const int cap = (first - p.map.group_off) / p.map.group + c;
                if (cap <= (last - p.map.group_off) / p.map.group) {
                    cap_row0[c] = p.map.base_cu[cap];
                    cap_rows[c] = p.map.base_cu[cap + 1];      // (raw: the row count is formed in the epilogue)
                }
            }
        }
    }

    // ---- DMA sources (gemm256h.hip): wave w moves pieces 4w..4w+3 of an A panel, 3w..3w+2 of a B panel (8 rows x 128 B each)
    const char* __restrict__ A = (const char*)p.A;
    const char* __restrict__ B = (const char*)p.B;
    unsigned a0, a1, a2, a3;
    {
        const int prow = lane >> 3;
        const int schunk = (lane & 7) ^ prow;
        auto arow = [&](int j) { int r = r0 + wid * 32 + 8 * j + prow; return r < p.M ? r : p.M - 1; };
        a0 = (unsigned)arow(0) * (unsigned)p.lda * 2u + schunk * 16;
        a1 = (unsigned)arow(1) * (unsigned)p.lda * 2u + schunk * 16;
        a2 = (unsigned)arow(2) * (unsigned)p.lda * 2u + schunk * 16;
        a3 = (unsigned)arow(3) * (unsigned)p.lda * 2u + schunk * 16;
    }
    const int apiece = wid * 4096, bpiece = wid * 3072;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
#define ISSUE_A(so, kt, q) DMA16(A + (size_t)((kt) * (BK * 2)) + ((q) == 0 ? a0 : (q) == 1 ? a1 : (q) == 2 ? a2 : a3), smem + (so) + apiece + (q) * 1024)
#define ISSUE_B(so, kt, q) DMA16(B + (size_t)((kt) * (BK * 2)) + ((q) == 0 ? b0 : (q) == 1 ? b1 : b2), smem + (so) + bpiece + (q) * 1024)
#define ISSUE_HALF_A(so, kt) ISSUE_A(so, kt, 0); ISSUE_A(so, kt, 1); ISSUE_A(so, kt, 2); ISSUE_A(so, kt, 3);
#define ISSUE_HALF_B(so, kt) ISSUE_B(so, kt, 0); ISSUE_B(so, kt, 1); ISSUE_B(so, kt, 2);

    f32x4 acc[8][3];
    const int frow = lane & 15, fkc = lane >> 4;
    const int fo0 = lds_off_h(frow, fkc), fo1 = lds_off_h(frow, 4 + fkc);
    const int xbase = wm * 128 * 128, wbase = wn * 48 * 128;
    typedef typename TT::vec8 frag_t;
    frag_t Fx0, Fx1, Fx2, Fx3, Fx4, Fx5, Fx6, Fx7, Fw0, Fw1, Fw2;
    frag_t Gx0, Gx1, Gx2, Gx3, Gx4, Gx5, Gx6, Gx7, Gw0, Gw1, Gw2;
#define LD(ptr) (*(const frag_t*)(ptr))
#define READ_FRAGS(P, sa, sb, fo)                                                                            \
    {                                                                                                        \
        const char* sa_ = smem + (sa) + xbase + (fo);                                                        \
        const char* sb_ = smem + (sb) + wbase + (fo);                                                        \
        P##w0 = LD(sb_); P##w1 = LD(sb_ + 2048); P##w2 = LD(sb_ + 4096);                                     \
        P##x0 = LD(sa_); P##x1 = LD(sa_ + 2048); P##x2 = LD(sa_ + 4096); P##x3 = LD(sa_ + 6144);             \
        P##x4 = LD(sa_ + 8192); P##x5 = LD(sa_ + 10240); P##x6 = LD(sa_ + 12288); P##x7 = LD(sa_ + 14336);   \
    }
#define MF(P, i, j) acc[i][j] = TT::mfma(P##w##j, P##x##i, acc[i][j]);
#define MROW(P, i) MF(P, i, 0) MF(P, i, 1) MF(P, i, 2)
#define RDX(P, n) P##x##n = LD(sa_ + (n) * 2048);
#define RDW(P, n) P##w##n = LD(sb_ + (n) * 2048);
#define SB __builtin_amdgcn_sched_barrier(0);
#define SYNC_TILE(cnt)                                                                                       \
    SB                                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #cnt ") lgkmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("" ::: "memory");
    // one 32-deep k-step (24 MFMAs, 11 fragment reads), software-pipelined as in gemm256h.hip: CUR's reads are issued one at a
    // time in the shadow of PREV's rows 4-7 and CUR's own rows 0-1, the DMA pieces of a half-stage between the MFMA rows
#define KSTEP(PREV, CUR, sa, sb, fo, IS0, IS1, IS2, IS3)                                                     \
    {                                                                                                        \
        const char* sa_ = smem + (sa) + xbase + (fo);                                                        \
        const char* sb_ = smem + (sb) + wbase + (fo);                                                        \
        SB MF(PREV, 4, 0) SB RDW(CUR, 0) SB MF(PREV, 4, 1) SB RDW(CUR, 1) SB MF(PREV, 4, 2) SB RDW(CUR, 2) SB IS0 \
        SB MF(PREV, 5, 0) SB RDX(CUR, 0) SB MF(PREV, 5, 1) SB RDX(CUR, 1) SB MF(PREV, 5, 2) SB IS1           \
        SB MF(PREV, 6, 0) SB RDX(CUR, 2) SB MF(PREV, 6, 1) SB RDX(CUR, 3) SB MF(PREV, 6, 2) SB IS2           \
        SB MROW(PREV, 7) SB IS3                                                                              \
        SB MF(CUR, 0, 0) SB RDX(CUR, 4) SB MF(CUR, 0, 1) SB RDX(CUR, 5) SB MF(CUR, 0, 2)                     \
        SB MF(CUR, 1, 0) SB RDX(CUR, 6) SB MF(CUR, 1, 1) SB RDX(CUR, 7) SB MF(CUR, 1, 2)                     \
        SB MROW(CUR, 2) MROW(CUR, 3) SB                                                                      \
    }
#define NOP_
#define ADV(x) { x += 2 * HALF; if (x >= RING) x -= RING; }
    const int nt = p.K / BK;   // >= 4 (host-checked)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sa = 0, sb = HALF;
    int i0 = 3 * HALF, i1 = 4 * HALF;
    STAMP(0)
    ISSUE_HALF_A(0, 0) ISSUE_HALF_A(2 * HALF, 1)      // (B of K tile 0: requested at the top of the kernel)
    SYNC_TILE(4)
    STAMP(1)
    READ_FRAGS(G, sa, sb, fo0)
    SB ISSUE_B(i0, 1, 0); ISSUE_B(i0, 1, 1); SB
    MROW(G, 0) MROW(G, 1) SB ISSUE_B(i0, 1, 2); SB MROW(G, 2) MROW(G, 3)
    KSTEP(G, F, sa, sb, fo1, ISSUE_A(i1, 2, 0);, ISSUE_A(i1, 2, 1);, ISSUE_A(i1, 2, 2);, ISSUE_A(i1, 2, 3);)
    ADV(sa) ADV(sb) ADV(i0) ADV(i1)
    int T = 1;
    for (; T <= nt - 3; ++T) {
        SYNC_TILE(4)
        KSTEP(F, G, sa, sb, fo0, ISSUE_B(i0, T + 1, 0);, ISSUE_B(i0, T + 1, 1);, ISSUE_B(i0, T + 1, 2);, NOP_)
        KSTEP(G, F, sa, sb, fo1, ISSUE_A(i1, T + 2, 0);, ISSUE_A(i1, T + 2, 1);, ISSUE_A(i1, T + 2, 2);, ISSUE_A(i1, T + 2, 3);)
        ADV(sa) ADV(sb) ADV(i0) ADV(i1)
    }
    SYNC_TILE(4)
    KSTEP(F, G, sa, sb, fo0, ISSUE_B(i0, T + 1, 0);, ISSUE_B(i0, T + 1, 1);, ISSUE_B(i0, T + 1, 2);, NOP_)
    // LN-fold operands of this lane's columns, requested behind the LAST operand DMA (from here on every wait is vmcnt(0), so
    // these loads cannot shift a counted wait) with a K tile and a half of MFMAs to land under: tile column cb + 4 fq + e,
    // cb = wn * 48 + 16 j, belongs to part cb / 64 (q, k, v)
    const int fq = lane >> 4, efrow = lane & 15;
    float4 bias4[3], s4[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int cb = wn * 48 + 16 * j;
        const int gn = (cb >> 6) * d + h * HD + (cb & 63) + 4 * fq;
        bias4[j] = *(const float4*)(p.bias + gn);
        s4[j] = *(const float4*)(p.ln_s + gn);
    }
    // ... and the (mean, rstd) of this lane's eight rows (fragment row 16 i + lane % 16 of the wave's 128)
    float2 rs_all[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = r0 + wm * 128 + 16 * i + efrow;
        rs_all[i] = p.rowstat[r < p.M ? r : p.M - 1];
    }
    KSTEP(G, F, sa, sb, fo1, NOP_, NOP_, NOP_, NOP_)
    ADV(sa) ADV(sb)
    SYNC_TILE(0)
    KSTEP(F, G, sa, sb, fo0, NOP_, NOP_, NOP_, NOP_)
    KSTEP(G, F, sa, sb, fo1, NOP_, NOP_, NOP_, NOP_)
    SB MROW(F, 4) MROW(F, 5) MROW(F, 6) MROW(F, 7) SB
    STAMP(2)

    // ================================================================ epilogue: the ring is idle after this barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tid < nseq) {
        const int sg = p.map.s0 + s_b + tid;
        const int row = my_c0 - p.map.row0 - r0, len = my_c1 - my_c0, pfx = my_pf;
        int slot = 0;
        if (pfx > 0) {
            const int first = p.map.s0 + s_b > p.map.group_off ? p.map.s0 + s_b : p.map.group_off;
            slot = (sg - p.map.group_off) / p.map.group - (first - p.map.group_off) / p.map.group;
            slot = slot < 0 ? 0 : (slot > p.ncap - 1 ? p.ncap - 1 : slot);
        }
        const unsigned my_seq = (unsigned)row | ((unsigned)len << 9) | ((unsigned)pfx << 16) | ((unsigned)slot << 23);
        *(unsigned*)(smem + SEQ_OFF + tid * 4) = my_seq;
        *(unsigned char*)(smem + EOT_OFF + tid) = (unsigned char)my_eot;
    }
    if (tid < 32) *(unsigned*)(smem + ZERO_OFF + tid * 4) = 0u;
    // (the tables and the zero line are read in the attention stage only, behind the barrier that ends the staging)
    // everything loaded so far is retired before the caption DMAs go out: with LDS-DMAs in flight the compiler would drain
    // vmcnt(0) in front of the first use of any ordinary load, and in front of any LDS access it can see (the staging below is asm)
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
    // ---- cached prefix K / V of the tile's captions -> LDS images (waves 0..2 NCAP - 1: wave = 2 slot + (0: K, 1: V))
    if (wid < 2 * p.ncap) {
        const int slot = wid >> 1, isv = wid & 1;
        int rows = 0, row0 = 0;
#pragma unroll
        for (int c = 0; c < NCAP; ++c) if (c == slot) { rows = cap_rows[c] - cap_row0[c]; row0 = cap_row0[c]; }
        rows = rows < p.caprows ? rows : p.caprows;
        const int vr = lane >> 3, vc = lane & 7;
        const u16* src0 = (const u16*)p.kv_base + (size_t)row0 * p.kv_ld + (1 + isv) * d + h * HD;
        char* img = smem + CAP_OFF + (2 * slot + isv) * cap_img;
        for (int pc = 0; 8 * pc < rows; ++pc) {
            const int r = 8 * pc + vr;
            const int sw = isv ? vswz(r) : (r & 7);
            if (r < rows) DMA16(src0 + (size_t)r * p.kv_ld + ((vc ^ sw) << 3), img + pc * 1024);
        }
    }
    // ---- this wave's 128 x 48 accumulators -> LN-fold -> 16 bit -> staging (inline-asm stores: see above)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + 16 * i + efrow;
        const float2 rs = rs_all[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int cb = wn * 48 + 16 * j, part = cb >> 6;
            const f32x2 v0 = lnfold_apply2(f32x2{acc[i][j][0], acc[i][j][1]}, rs.x, rs.y, f32x2{s4[j].x, s4[j].y}, f32x2{bias4[j].x, bias4[j].y});
            const f32x2 v1 = lnfold_apply2(f32x2{acc[i][j][2], acc[i][j][3]}, rs.x, rs.y, f32x2{s4[j].z, s4[j].w}, f32x2{bias4[j].z, bias4[j].w});
            const uint2 pk = uint2{TT::pack2(v0), TT::pack2(v1)};
            const int chunk = ((cb & 63) >> 3) + (fq >> 1);
            const int sw = part == 2 ? vswz(row) : (row & 7);
            char* dst = smem + part * HALF + row * 128 + ((chunk ^ sw) << 4) + (fq & 1) * 8;
            const unsigned off = (unsigned)(uintptr_t)(lds_char_t*)dst;
            const unsigned long long pk64 = __builtin_bit_cast(unsigned long long, pk);
#ifdef LEAF_DIAG_QA_NOSTAGE   // diagnostic only (garbage results): the q|k|v staging stores are not issued -- attributes the launch's LDS bank conflicts
            asm volatile("" ::"v"(off), "v"(pk64));
#else
            asm volatile("ds_write_b64 %0, %1" ::"v"(off), "v"(pk64) : "memory");
#endif
        }
    }
    STAMP(3)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP(4)

    // ================================================================ attention: one wave per sequence, attn_fwd_kernel's body
    const int r16 = lane & 15, g = lane >> 4;
    // diagnostic build (make qa_phases: -DLEAF_GEMM_STAMPS -DLEAF_QA_PHASES, caller's buffer 16 slots per workgroup): where wave 0's
    // attention time goes, summed over its sequences, into slots 8..13 -- 0 table + K fragments, 1 Q + S + mask + max, 2 exp2 + sum +
    // rcp, 3 P + V^T + PV, 4 pack + store issue, 5 loop overhead.  Every boundary waits for the LDS and for the stamp itself
    // (~150 ticks, counted in the phase that follows): shares, not times.
#if defined(LEAF_GEMM_STAMPS) && defined(LEAF_QA_PHASES)
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long ph_prev = __builtin_amdgcn_s_memtime();
#define PH(i)                                                                             \
    {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                       \
        ph[i] += t_ - ph_prev;                                                            \
        ph_prev = t_;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                \
    }
#else
#define PH(i)
#endif
#ifdef LEAF_DIAG_QA_NOATTN    // diagnostic only (garbage results): no attention stage at all
    for (int si = wid; si < 0; si += 8) {
#else
    for (int si = wid; si < nseq; si += 8) {
#endif
        const unsigned sq = *(const unsigned*)(smem + SEQ_OFF + si * 4);
        const int row_s = sq & 511, len = (sq >> 9) & 127, pfx = (sq >> 16) & 127, slot = (sq >> 23) & 3;
        const int ctx = pfx + len;
        const int eot = p.eot_pos ? (int)*(const unsigned char*)(smem + EOT_OFF + si) : -1;
        const int ntl = (ctx + 15) >> 4;
        const int qt0 = eot >= 0 ? eot >> 4 : pfx >> 4;
        const int qt1 = eot >= 0 ? eot >> 4 : ntl - 1;
        // position pos of this sequence -> its K (or Q) row image, chunk-swizzled by the row's index inside that image
        // (byte offsets and selects, no pointers from two bases: the compiler turned those into exec-masked branches)
        const int capK_off = CAP_OFF + 2 * slot * cap_img, capV_off = capK_off + cap_img;
        auto krow = [&](int pos, int chunk) -> const char* {
            const bool inp = pos < pfx;
            const int r = inp ? pos : row_s + pos - pfx;
            return smem + (inp ? capK_off : HALF) + r * 128 + ((chunk ^ (r & 7)) << 4);
        };
        typename TT::vec8 kf[MAXT][2];
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt < ntl) {
                int row = kt * 16 + r16; row = row < ctx ? row : ctx - 1;
                kf[kt][0] = *(const typename TT::vec8*)krow(row, g);
                kf[kt][1] = *(const typename TT::vec8*)krow(row, g + 4);
            }
        }
        PH(0)
#pragma unroll
        for (int qt = 0; qt < MAXT; ++qt) {
            if (qt < ntl && qt >= qt0 && qt <= qt1) {
                const int qidx = qt * 16 + r16;
                int qv = qidx < ctx ? qidx : ctx - 1;
                qv = qv < pfx ? pfx : qv;
                const int qr = row_s + qv - pfx;
                const typename TT::vec8 qf0 = *(const typename TT::vec8*)(smem + qr * 128 + ((g ^ (qr & 7)) << 4));
                const typename TT::vec8 qf1 = *(const typename TT::vec8*)(smem + qr * 128 + (((g + 4) ^ (qr & 7)) << 4));
                f32x4 sc[MAXT];
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt <= qt; ++kt) {
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                    a = TT::mfma(kf[kt][0], qf0, a);
                    a = TT::mfma(kf[kt][1], qf1, a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float s = a[e] * 0.18033688011112042f;   // 1/sqrt(64) * log2(e)
                        if (kt == qt) s = (kt * 16 + 4 * g + e) > qv ? -INFINITY : s;
                        a[e] = s;
                        m = __builtin_fmaxf(m, s);
                    }
                    sc[kt] = a;
                }
                m = rows4_max(m);
                PH(1)
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pe = __builtin_amdgcn_exp2f(sc[kt][e] - m);
                        sc[kt][e] = pe;
                        sum += pe;
                    }
                sum = rows4_sum(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);   // (attention.hip: the same v_rcp_f32, not the IEEE division sequence)
                PH(2)
                f32x4 o[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < MAXT; ++kt) {
                    if (kt <= qt) {
                        const s16x4 pf = __builtin_bit_cast(
                            s16x4, pack4_bounded<TT>(sc[kt][0] * inv, sc[kt][1] * inv, sc[kt][2] * inv, sc[kt][3] * inv));
                        // V^T fragment (attention.hip load_vt_frag): lane 4 q + pp of a 16-lane group supplies key kt 16 + 4 g + q,
                        // dims dim0 + 4 pp .. + 3; keys beyond the sequence read the zero line (P is zero there; 0 x anything must stay +0)
                        const int q4 = r16 >> 2, pp = r16 & 3;
                        const int pos = kt * 16 + 4 * g + q4;
                        const bool past = pos >= ctx, inp = pos < pfx;
                        const int vr = past ? 0 : (inp ? pos : row_s + pos - pfx);                  // row inside its image (0: the zero line)
                        const int vbase = (past ? ZERO_OFF : (inp ? capV_off : 2 * HALF) + vr * 128) + (pp & 1) * 8;
                        const int vsw = past ? 0 : vswz(vr);                                         // (any chunk of the zero line is zero)
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                            const int chunk = (2 * dt + (pp >> 1)) ^ vsw;
                            const s16x4 vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + vbase + (chunk << 4)));
                            o[dt] = TT::mfma16(vf, pf, o[dt]);
                        }
                    }
                }
                PH(3)
                if (eot >= 0 ? qidx == eot : (qidx < ctx && qidx >= pfx)) {
                    // (32-bit element offset: rows * d * 2 B < 4 GiB is checked by the launcher)
                    u16* op = (u16*)p.out + (unsigned)((eot >= 0 ? s_b + si : r0 + row_s + qidx - pfx) * d + h * HD + 4 * g);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        *(uint2*)(op + dt * 16) = pack4_bounded<TT>(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
                }
                PH(4)
            }
        }
        PH(5)
    }
#if defined(LEAF_GEMM_STAMPS) && defined(LEAF_QA_PHASES)
    if (p.stamps && tid == 0)
        for (int i = 0; i < 6; ++i) ((unsigned long long*)p.stamps)[(size_t)blockIdx.x * 16 + 8 + i] = ph[i];
#endif
#undef PH
#ifdef LEAF_GEMM_STAMPS
    __syncthreads();      // the stamp below then is the end of the SLOWEST wave's sequences
#endif
    STAMP(5)
#undef ADV
#undef DMA16
#undef ISSUE_A
#undef ISSUE_B
#undef ISSUE_HALF_A
#undef ISSUE_HALF_B
#undef READ_FRAGS
#undef MROW
#undef SYNC_TILE
#undef KSTEP
#undef MF
#undef RDW
#undef RDX
#undef NOP_
#undef SB
#undef LD
}

}  // namespace

// Rows per caption image for sequences of <= max_len positions, and how many captions' images the LDS then holds
static int caprows_for(int max_len) { const int m = max_len > 0 ? max_len : CAPROWS; return (m + 15) / 16 * 16; }
int leaf_qkv_attn_tile_rows() { return BM; }
int leaf_qkv_attn_ncap(int max_len) {
    const int cr = caprows_for(max_len);
    if (cr > CAPROWS) return 0;
    const int n = (MISC_OFF - CAP_OFF) / (2 * cr * 128);
    return n > NCAP ? NCAP : n;
}
void leaf_qkv_attn_lds_plan(int max_len, int* ncap, int* caprows) {
    *ncap = leaf_qkv_attn_ncap(max_len);
    *caprows = caprows_for(max_len);
}

// Greedy cut of the launch's sequences into M tiles of whole sequences: <= tile_rows rows, and (prefix mode) the prefixed sequences
// of a tile belong to <= ncap consecutive captions (sequence s >= group_off belongs to caption (s - group_off) / group).
// lens[i] = rows sequence s0 + i computes; out = (first sequence, first row) pairs, both launch-relative: out[2 t], out[2 t + 1] for
// tile t and (n, total rows) behind the last tile (2 (n_tiles + 1) ints).
int leaf_qkv_attn_plan(const int32_t* lens, int ctx, int s0, int n, int prefixed, int group, int group_off, int tile_rows, int ncap,
                       int32_t* out) {
    int nt = 0, rows = 0, first_cap = -1, total = 0;
    out[0] = 0; out[1] = 0;
    for (int i = 0; i < n; ++i) {
        const int L = lens ? lens[i] : ctx;
        const int sg = s0 + i;
        const int cap = (prefixed && sg >= group_off && group > 0) ? (sg - group_off) / group : -1;
        const bool cut = i > out[2 * nt] && (rows + L > tile_rows || (cap >= 0 && first_cap >= 0 && cap - first_cap >= ncap));
        if (cut) { ++nt; out[2 * nt] = i; out[2 * nt + 1] = total; rows = 0; first_cap = -1; }
        rows += L;
        total += L;
        if (cap >= 0 && first_cap < 0) first_cap = cap;
    }
    ++nt;
    out[2 * nt] = n; out[2 * nt + 1] = total;
    return nt;
}

bool leaf_qkv_attn_eligible(int d, int heads, int ctx, int K, int max_len) {
    return d == heads * HD && d % 64 == 0 && K % BK == 0 && K >= 4 * BK && ctx <= CAPROWS && leaf_qkv_attn_ncap(max_len > 0 ? max_len : ctx) >= 1 &&
           (unsigned long long)3 * d * K * 2ull < (1ull << 32);
}

hipError_t leaf_launch_qkv_attn(const QkvAttnArgs& a_in, int dtype, hipStream_t s) {
    QkvAttnArgs a = a_in;
    a.stamps = leaf_gemm_get_stamps();
    if (a.n_tiles < 1 || a.heads < 1 || a.ncap < 1 || a.ncap > NCAP || a.caprows < 16 || a.caprows % 16 || a.caprows > CAPROWS ||
        2 * a.ncap * a.caprows * 128 > MISC_OFF - CAP_OFF || (unsigned long long)a.M * a.lda * 2ull >= (1ull << 32))
        return hipErrorInvalidValue;
    // head groups per M range: the fewest (1, 2, 4) that bring an XCD's live weight panels (heads / hsplit x 192 rows x K) under half
    // of its 4-MiB L2 -- ViT-L 2 (1.8 MB; beyond-L2 fetches 911 -> 503 MB per launch, L2 hit rate 77 -> 86 %, launch -1.5 %:
    // profiles/r04_fused_attn_hsplit.txt), ViT-H 4 (1.6 MB), bigG 4 (2.5 MB).  LEAF_QKVATTN_HSPLIT=1|2|4 overrides.
    static const int hsplit_env = [] { const char* e = getenv("LEAF_QKVATTN_HSPLIT"); return e ? atoi(e) : 0; }();
    a.hsplit = 1;
    if (hsplit_env == 1 || ((hsplit_env == 2 || hsplit_env == 4) && a.heads % hsplit_env == 0)) {
        a.hsplit = hsplit_env;
    } else {
        for (int hs = 1; hs <= 4; hs *= 2) {
            if (a.heads % hs) break;
            a.hsplit = hs;
            if ((size_t)(a.heads / hs) * 192 * a.K * 2 <= (2u << 20)) break;
        }
    }
    const int R = 8 / a.hsplit;
    const dim3 grid(a.hsplit > 1 ? 8 * ((a.n_tiles + R - 1) / R) * (a.heads / a.hsplit) : a.n_tiles * a.heads), blk(512);
#define LEAF_QA(TT)                                                                                                    \
    {                                                                                                                  \
        static bool attr = false;                                                                                      \
        if (!attr) { (void)hipFuncSetAttribute((const void*)qkv_attn_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, RING); attr = true; } \
        hipLaunchKernelGGL((qkv_attn_kernel<TT>), grid, blk, RING, s, a);                                              \
    }
    if (dtype == LEAF_F16) LEAF_QA(F16) else LEAF_QA(BF16)
#undef LEAF_QA
    return hipGetLastError();
}
