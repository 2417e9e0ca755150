// Weight / bias gradients of the linear layers of one transformer block, ONE grouped launch.
//
//     dW[n][k] += alpha * sum_r dY[r][n] * X[r][k]          db[n] += alpha * sum_r dY[r][n]
//
// (autograd of nn.Linear / in_proj under the TextFARE backward, utils_AT.py:321-337; alpha = 1 / loss scale).  Both
// operands are row-major with the REDUCTION index r (packed token rows) as their row, i.e. a "TN" product.  Instead of
// materialising dY^T and X^T (two transposes per weight, the first implementation) the 32-row operand slabs are staged
// row-major in LDS and read as MFMA fragments with the gfx950 transposing read ds_read_b64_tr_b16.
//
// Tile 128 (n) x 128 (k), 4 waves of 64 x 64, k-step = 32 rows, one barrier per step.
// Two kernels: wgrad_tn_dma_kernel (both operands of one 16-bit type, the default: slabs arrive by LDS-DMA into an
// XOR-swizzled image, software-pipelined over four LDS buffers) and wgrad_tn_kernel (X converted on the way: register-staged,
// double-buffered; LEAF_GRAD_DTYPE=bf16 beside an fp16 forward, or a bf16 forward beside fp16 gradients).  Measured per ViT-L block at 3,200 rows inside the step: 107 us (first version: one slab of look-ahead through
// registers) -> 70 us; tools/wgrad_stamps.py for where a step's cycles go.
// The problems of a block (c_proj, c_fc, out_proj, in_proj) share one launch: blockIdx -> (problem, tile) through a
// small table, so the 432 tiles of a ViT-L block fill the chip where the largest single weight has 144.
// Column sums (bias gradients) are accumulated by the blocks of the first k-tile column while they stage dY and reduced
// through LDS in a fixed order: the whole result is deterministic (no atomics).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int TM = 128, TN = 128, KS = 32;
constexpr int LD = 136;                         // LDS row stride (elements): 272 B
constexpr int SLAB = KS * LD * 2;               // 8,704 B
constexpr int LDS_BYTES = 4 * SLAB;             // register-staged kernel: dY and X slabs, double-buffered: 34,816 B

#ifdef LEAF_GEMM_STAMPS   // diagnostic builds: s_memtime at phase boundaries (tools/wgrad_stamps.py)
#define WSTAMP(i)                                                                                         \
    if (a.stamps && threadIdx.x == 0) {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        ((unsigned long long*)a.stamps)[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
#else
#define WSTAMP(i)
#endif

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

// Tile order inside one problem: consecutive tiles (they land on one XCD, xcd_remap) walk down BH <= 8 rows of the tile grid
// before they advance one column, so the ~54 tiles an XCD holds at a time form a near-square block: they share BH dY panels
// and 54 / BH X panels instead of 2-3 and 18-24 -- about half as many distinct operand panels to pull through that XCD's L2.
__device__ __forceinline__ void tile_of(int local, int tiles_n, int tiles_k, int& tn, int& tk) {
    int bh = tiles_n < 8 ? tiles_n : 8;
    while (tiles_n % bh) --bh;
    const int blk = local / (bh * tiles_k), r = local - blk * (bh * tiles_k);
    tn = blk * bh + r % bh;
    tk = r / bh;
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

// Register-staged form (operand types differ: X is converted on its way to LDS).  Compiler-managed loads, one slab of
// look-ahead, two LDS buffers -- the first version of this kernel.  (A deeper pipeline with inline-asm loads and counted
// waits was tried here and withdrawn: nothing ties an asynchronous load's destination registers to the later wait, so under
// register pressure the compiler may move or re-use them while the load is in flight -- the bf16-forward / fp16-gradient
// instantiation faulted.  The LDS-DMA form below has no such registers.)
template <class XT, class GT>
__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradArgs a) {
    leaf_fp16_sat_mode();
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
    int tn, tk;
    tile_of(local, P.Nw / TM, P.tiles_k, tn, tk);
    const int n0 = tn * TM, k0 = tk * TN;
    const int rows = a.rows;
    const bool do_bias = tk == 0 && P.db != nullptr;

    // staging: thread -> (row = tid >> 4 [+16], 8-column chunk = tid & 15) of each 32 x 128 slab
    const int srow = tid >> 4, sch = tid & 15;
    const u16* yp = P.dY + (size_t)srow * P.ldy + n0 + sch * 8;
    const u16* xp = P.X + (size_t)srow * P.ldx + k0 + sch * 8;
    const size_t ystep = (size_t)16 * P.ldy, xstep = (size_t)16 * P.ldx;
    const int lds_off = srow * (LD * 2) + sch * 16;
    u32x4 y0, y1, x0, x1;
    const u32x4 z4 = u32x4{0u, 0u, 0u, 0u};
#define LOAD_SLAB(ks)                                                                                       \
    {                                                                                                       \
        const int r_ = (ks) * KS + srow;                                                                    \
        const size_t oy_ = (size_t)(ks) * KS * P.ldy, ox_ = (size_t)(ks) * KS * P.ldx;                      \
        y0 = r_ < rows ? *(const u32x4*)(yp + oy_) : z4;                                                    \
        x0 = r_ < rows ? cvt8<XT, GT>(*(const u32x4*)(xp + ox_)) : z4;                                      \
        y1 = r_ + 16 < rows ? *(const u32x4*)(yp + oy_ + ystep) : z4;                                       \
        x1 = r_ + 16 < rows ? cvt8<XT, GT>(*(const u32x4*)(xp + ox_ + xstep)) : z4;                         \
    }
#define STORE_SLAB(buf)                                                                                     \
    {                                                                                                       \
        char* yb_ = smem + (buf) * 2 * SLAB;                                                                \
        *(u32x4*)(yb_ + lds_off) = y0;                                                                      \
        *(u32x4*)(yb_ + lds_off + 16 * LD * 2) = y1;                                                        \
        *(u32x4*)(yb_ + SLAB + lds_off) = x0;                                                               \
        *(u32x4*)(yb_ + SLAB + lds_off + 16 * LD * 2) = x1;                                                 \
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

    LOAD_SLAB(0)
    if (do_bias) { ADD_COLS(y0) ADD_COLS(y1) }
    STORE_SLAB(0)
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) {
            LOAD_SLAB(ks + 1)
        }
        const char* yb = smem + buf * 2 * SLAB;
        const char* xb = yb + SLAB;
        typename GT::vec8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = __builtin_bit_cast(typename GT::vec8, tr8(xb, 8 * g, wn * 64 + 16 * j, lane));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const typename GT::vec8 af = __builtin_bit_cast(typename GT::vec8, tr8(yb, 8 * g, wm * 64 + 16 * i, lane));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = GT::mfma(af, bf[j], acc[i][j]);
        }
        if (ks + 1 < nk) {
            if (do_bias) { ADD_COLS(y0) ADD_COLS(y1) }
            STORE_SLAB(buf ^ 1)
        }
        __syncthreads();
    }
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


// ---------------------------------------------------------------------------------------------------------------------------
// LDS-DMA form (operands of one 16-bit type: the default fp16 / fp16 gradient path).  tools/wgrad_stamps.py: the register-staged
// kernel above spends 1,190 cycles per 32-row step for 256 cycles of MFMA per wave, however deep its loads are pipelined --
// a CU takes in only ~12 B/clk through global_load -> VGPR (MI355X_MICROARCH.md: "prologue HBM burst ~11 B/cyc/CU"), and a
// workgroup needs 16 KB per step.  global_load_lds moves whole 1-KiB pieces L2 -> LDS at ~45 B/clk/CU (tools/dma_probe2) and
// takes the VGPR -> LDS store path (13 cycles per ds_write_b128) out of the step as well.
//
// Slab image in LDS: 32 rows of 256 B (128 columns), no padding; the 16-byte chunk C of row r sits at position C ^ swz(r)
// (applied on the SOURCE side of the DMA), swz(r) = 2 ((r & 3) | ((r >> 3) & 1) << 2): the eight rows {0-3, 8-11} (+4, +16)
// that one half-wave cycle of ds_read_b64_tr_b16 touches then map their 32-byte column pairs to eight different 32-byte
// bank groups -- conflict-free transposing reads without row padding.
// Pipeline (four buffers of dY | X slabs = 64 KiB, two workgroups per CU): slab j is requested at step j - 4 (after that step's
// barrier, into the buffer whose fragments were read two barriers ago), awaited with a counted vmcnt before the barrier of
// step j - 1, its fragments are read during step j - 1 and multiplied during step j.  ONE raw s_barrier per step.
constexpr int DSLAB = KS * 256;            // 8 KiB
constexpr int DBUF = 2 * DSLAB;            // dY slab | X slab
constexpr int DNBUF = 4;
constexpr int DLDS_BYTES = DNBUF * DBUF;   // 64 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ int swz(int row) { return 2 * ((row & 3) | (((row >> 3) & 1) << 2)); }

template <class GT>
__global__ __launch_bounds__(256, 2) void wgrad_tn_dma_kernel(WgradArgs a) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < a.nprob && bid >= a.p[i].tile0) pi = i;
    const WgradProb& P = a.p[pi];
    const int local = bid - P.tile0;
    int tn, tk;
    tile_of(local, P.Nw / TM, P.tiles_k, tn, tk);
    const int n0 = tn * TM, k0 = tk * TN;
    const int rows = a.rows;
    const bool do_bias = tk == 0 && P.db != nullptr;
    const int nk = (rows + KS - 1) / KS;
    const int wm = wid >> 1, wn = wid & 1, g = lane >> 4;

    // ---- DMA sources: wave w moves rows 8 w .. 8 w + 7 of both slabs, as two 1-KiB pieces (4 rows x 16 chunks) each
    const int prow = 8 * wid + (lane >> 4);                 // slab row of this lane in piece 0 (piece 1: + 4)
    const int pc0 = ((lane & 15) ^ swz(prow)) * 8, pc1 = ((lane & 15) ^ swz(prow + 4)) * 8;   // source column of the lane's chunk
    const int pdst = wid * 8 * 256;                         // byte offset of the wave's first piece inside a slab
    // The DMA is issued as inline asm (M0 = LDS byte address of the 1-KiB piece): with the builtin, hipcc waits vmcnt(0) before
    // every later compiler-visible LDS read (it cannot prove that the DMA's destination and the read do not overlap), which
    // drains the look-ahead every step.  Ordering is explicit instead: counted vmcnt + s_barrier below.
    typedef __attribute__((address_space(3))) char lds_char_t;
#define DMA16(src, dst)                                                                                     \
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off"                                      \
                 :: "v"((const void*)(src)), "s"((unsigned)(uintptr_t)(lds_char_t*)(dst)) : "memory")
#define ISSUE_PIECE(j, Q)   /* piece Q (dY rows 0-3, dY rows 4-7, X rows 0-3, X rows 4-7 of this wave's 8 rows) of slab j into  \
                               buffer j % DNBUF; rows past the end are clamped here and zeroed in AWAIT_SLAB */          \
    {                                                                                                       \
        char* b_ = smem + ((j) % DNBUF) * DBUF + pdst + ((Q) & 1) * 1024 + ((Q) >> 1) * DSLAB;              \
        const int r_ = (j) * KS + prow + ((Q) & 1) * 4;                                                     \
        const size_t q_ = r_ < rows ? r_ : rows - 1;                                                        \
        if ((Q) < 2) { DMA16(P.dY + q_ * P.ldy + n0 + (((Q) & 1) ? pc1 : pc0), b_); }                       \
        else { DMA16(P.X + q_ * P.ldx + k0 + (((Q) & 1) ? pc1 : pc0), b_); }                                \
    }
#define ISSUE_SLAB(j) ISSUE_PIECE(j, 0) ISSUE_PIECE(j, 1) ISSUE_PIECE(j, 2) ISSUE_PIECE(j, 3)
    // this wave's pieces of slab j have landed once only the 8 younger pieces (two slabs) are outstanding; the rows of the last
    // slab that lie past the end are then zeroed by the wave that fetched them (the reduction must not see them)
#define AWAIT_SLAB(j)                                                                                       \
    {                                                                                                       \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                    \
        if ((j) * KS + KS > rows) {                                                                         \
            char* b_ = smem + ((j) % DNBUF) * DBUF + pdst + lane * 16;                                      \
            const u32x4 z_ = u32x4{0u, 0u, 0u, 0u};                                                         \
            if ((j) * KS + prow >= rows) { *(u32x4*)b_ = z_; *(u32x4*)(b_ + DSLAB) = z_; }                  \
            if ((j) * KS + prow + 4 >= rows) { *(u32x4*)(b_ + 1024) = z_; *(u32x4*)(b_ + DSLAB + 1024) = z_; } \
        }                                                                                                   \
    }
    // ---- fragment addresses (loop-invariant lane offsets; the buffer offset is a compile-time constant per unrolled step)
    const int fi = lane & 15, fq = fi >> 2, fp = fi & 3;
    const int frow = 8 * g + fq;                            // lo rows; hi = + 4 (same swizzle)
    int offb[4], offa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        offb[j] = frow * 256 + (((wn * 8 + 2 * j + (fp >> 1)) ^ swz(frow)) << 4) + (fp & 1) * 8;
        offa[j] = frow * 256 + (((wm * 8 + 2 * j + (fp >> 1)) ^ swz(frow)) << 4) + (fp & 1) * 8;
    }
    typename GT::vec8 af[2][4], bf[2][4];
    auto tr8s = [&](const char* p_) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p_ + 4 * 256));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(typename GT::vec8, r);
    };
    // bias column sums (workgroups of the first k-tile column): thread -> (row tid >> 4 [+16], chunk tid & 15) of the dY slab
    const int srow = tid >> 4, sch = tid & 15;
    const int cs_off0 = srow * 256 + ((sch ^ swz(srow)) << 4), cs_off1 = (srow + 16) * 256 + ((sch ^ swz(srow + 16)) << 4);
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
    // fragments of slab j in four quarters (two operand fragments each), so that a step can issue them between its MFMA rows
#define READ_Q(F, j, Q)                                                                                     \
    {                                                                                                       \
        const char* yb_ = smem + ((j) % DNBUF) * DBUF;                                                      \
        if ((Q) == 0) { bf[F][0] = tr8s(yb_ + DSLAB + offb[0]); bf[F][1] = tr8s(yb_ + DSLAB + offb[1]); }   \
        if ((Q) == 1) { bf[F][2] = tr8s(yb_ + DSLAB + offb[2]); bf[F][3] = tr8s(yb_ + DSLAB + offb[3]); }   \
        if ((Q) == 2) { af[F][0] = tr8s(yb_ + offa[0]); af[F][1] = tr8s(yb_ + offa[1]); }                   \
        if ((Q) == 3) {                                                                                     \
            af[F][2] = tr8s(yb_ + offa[2]); af[F][3] = tr8s(yb_ + offa[3]);                                 \
            if (do_bias) {                                                                                  \
                const typename GT::vec8 e0_ = *(const typename GT::vec8*)(yb_ + cs_off0);                   \
                const typename GT::vec8 e1_ = *(const typename GT::vec8*)(yb_ + cs_off1);                   \
                _Pragma("unroll") for (int t = 0; t < 8; ++t) cs[t] += GT::to_f32(e0_[t]);                  \
                _Pragma("unroll") for (int t = 0; t < 8; ++t) cs[t] += GT::to_f32(e1_[t]);                  \
            }                                                                                               \
        }                                                                                                   \
    }
#define READ_FRAGS(F, j) READ_Q(F, j, 0) READ_Q(F, j, 1) READ_Q(F, j, 2) READ_Q(F, j, 3)
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    WSTAMP(0)
    ISSUE_SLAB(0) ISSUE_SLAB(1) ISSUE_SLAB(2)
    AWAIT_SLAB(0)
    ISSUE_SLAB(3)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    READ_FRAGS(0, 0)
    WSTAMP(1)
    // One step.  The MFMA rows of slab ks (fragments read a step ago) alternate with the quarter reads of slab ks + 1; the
    // sched_barriers pin that order (hoisted in front of the MFMAs, the reads would make the compiler's in-order lgkmcnt
    // wait for them before the first MFMA).  Slabs past the end are all zeros (AWAIT_SLAB), so the trip count is rounded up to
    // a multiple of four and the body stays branch-free.
#define MROW(F, i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = GT::mfma(af[F][i], bf[F][j], acc[i][j]);
#define SB __builtin_amdgcn_sched_barrier(0);
#define STEP(U, ks)                                                                                         \
    {                                                                                                       \
        AWAIT_SLAB((ks) + 1)                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* this wave's reads of slab ks (issued a step ago) are done */ \
        __builtin_amdgcn_s_barrier();                                                                       \
        asm volatile("" ::: "memory");                                                                      \
        /* slab ks + 4 goes into the buffer slab ks was read from; its four pieces are issued between the MFMA rows */ \
        SB MROW(U & 1, 0) SB ISSUE_PIECE((ks) + 4, 0) READ_Q((U + 1) & 1, (ks) + 1, 0)                      \
        SB MROW(U & 1, 1) SB ISSUE_PIECE((ks) + 4, 1) READ_Q((U + 1) & 1, (ks) + 1, 1)                      \
        SB MROW(U & 1, 2) SB ISSUE_PIECE((ks) + 4, 2) READ_Q((U + 1) & 1, (ks) + 1, 2)                      \
        SB MROW(U & 1, 3) SB ISSUE_PIECE((ks) + 4, 3) READ_Q((U + 1) & 1, (ks) + 1, 3) SB                   \
    }
    for (int ks = 0; ks < nk; ks += 4) { STEP(0, ks) STEP(1, ks + 1) STEP(2, ks + 2) STEP(3, ks + 3) }
#undef MROW
#undef SB
#undef READ_Q
#undef STEP
#undef READ_FRAGS
#undef AWAIT_SLAB
#undef ISSUE_SLAB
#undef ISSUE_PIECE
#undef DMA16
    WSTAMP(2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the look-ahead requests past the last slab
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float alpha = a.alpha ? *a.alpha : 1.0f;
    // ---- dW tile: lane holds weight rows 16 i + 4 g + e, column 16 j + (lane & 15).  All 64 old values are requested before
    // the first is used (one memory round trip instead of sixteen)
    {
        float* wp = P.dW + (size_t)(n0 + wm * 64 + 4 * g) * P.Kw + k0 + wn * 64 + (lane & 15);
        float old_[4][4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) old_[i][e][j] = wp[(size_t)(16 * i + e) * P.Kw + 16 * j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) wp[(size_t)(16 * i + e) * P.Kw + 16 * j] = fmaf(alpha, acc[i][j][e], old_[i][e][j]);
    }
    // ---- bias: 16 row-group partials per column, summed in a fixed order
    if (do_bias) {
        float* red = (float*)smem;   // [16][128]
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
    WSTAMP(3)
}

}  // namespace

// eligibility of one problem for the grouped TN kernel
bool leaf_wgrad_tn_ok(int Nw, int Kw, int ldy, int ldx) { return Nw % TM == 0 && Kw % TN == 0 && ldy % 8 == 0 && ldx % 8 == 0; }

hipError_t leaf_launch_wgrad_group(WgradArgs a, int x_dtype, int g_dtype, hipStream_t s) {
    if (a.nprob < 1 || a.nprob > 4 || a.rows < 1) return hipErrorInvalidValue;
    a.stamps = leaf_gemm_get_stamps();
    int tiles = 0;
    for (int i = 0; i < a.nprob; ++i) {
        WgradProb& p = a.p[i];
        if (!leaf_wgrad_tn_ok(p.Nw, p.Kw, p.ldy, p.ldx) || !p.dY || !p.X || !p.dW) return hipErrorInvalidValue;
        p.tile0 = tiles;
        p.tiles_k = p.Kw / TN;
        tiles += (p.Nw / TM) * p.tiles_k;
    }
    const bool xf = x_dtype == LEAF_F16, gf = g_dtype == LEAF_F16;
    // LEAF_WGRAD_DMA=0: the register-staged kernel also for equal operand types (A/B)
    static int use_dma = -1;
    if (use_dma < 0) { const char* e = getenv("LEAF_WGRAD_DMA"); use_dma = (e && e[0] == '0') ? 0 : 1; }
#define LEAF_WG(XT, GT) hipLaunchKernelGGL((wgrad_tn_kernel<XT, GT>), dim3(tiles), dim3(256), LDS_BYTES, s, a)
#define LEAF_WG_DMA(GT)                                                                                      \
    {                                                                                                        \
        static bool attr = false;                                                                            \
        if (!attr) {                                                                                         \
            (void)hipFuncSetAttribute((const void*)wgrad_tn_dma_kernel<GT>, hipFuncAttributeMaxDynamicSharedMemorySize, DLDS_BYTES); \
            attr = true;                                                                                     \
        }                                                                                                    \
        hipLaunchKernelGGL((wgrad_tn_dma_kernel<GT>), dim3(tiles), dim3(256), DLDS_BYTES, s, a);            \
    }
    if (xf == gf && use_dma) { if (gf) LEAF_WG_DMA(F16) else LEAF_WG_DMA(BF16) }
    else if (xf && gf) LEAF_WG(F16, F16);
    else if (xf) LEAF_WG(F16, BF16);
    else if (gf) LEAF_WG(BF16, F16);
    else LEAF_WG(BF16, BF16);
#undef LEAF_WG
#undef LEAF_WG_DMA
    return hipGetLastError();
}
