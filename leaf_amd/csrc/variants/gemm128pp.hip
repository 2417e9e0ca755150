// NT GEMM "ping-pong": 128 x 256 output tile per 4-WAVE workgroup, TWO workgroups per CU (80 KiB of LDS, <= 256 VGPRs each).
//
// Why (round 3; profiles/r03_baseline_mfma_util.json): in the one-workgroup-per-CU 256 x 256 kernel (gemm256h.hip) the MFMA pipes
// are busy 48-50 % of the cycles on the K = 768 shapes -- nothing overlaps a tile's epilogue (19-25 % of a tile, 50 % for out_proj),
// its first-DMA wait or its barriers.  Here the two co-resident workgroups of a CU run half a tile out of phase: while one
// stores its tile, the other one's K loop has all four matrix pipes (one wave per SIMD each).
//
// What makes two rings fit into 160 KiB: a K tile (64 deep) of this shape is THREE 16-KiB units -- the A panel (128 rows x
// 128 B) and the two halves of the B panel (B_lo: columns 0-127, B_hi: 128-255) -- and the K tile is multiplied in two PHASES:
// phase 0 = A x B_lo (output columns 0-127), phase 1 = A x B_hi, one barrier in front of each.  B_lo is dead after phase 0,
// so the ring needs 2 A slots + 3 B slots = 5 units = 80 KiB for a look-ahead of one whole K tile on every unit:
//
//   barrier X_t : A_t, B_lo_t landed                | issue A_{t+1} -> A slot (t+1) & 1, B_lo_{t+1} -> B slot (2t+2) % 3
//   phase 0     : 16 fragment reads, 32 MFMAs       |   (freed by: A_{t-1}, B_hi_{t-1}, both dead since X_t)
//   barrier Y_t : B_hi_t landed, B_lo_t dead        | issue B_hi_{t+1} -> B slot (2t+3) % 3 = the slot of B_lo_t
//   phase 1     : 8 fragment reads (A fragments are kept in registers), 32 MFMAs
//
// DMA pieces are whole 128-B lines (8 rows x 128 B per wave instruction, source-side XOR swizzle, gemm256h.hip's scheme); a
// wave moves 12 pieces per K tile.  Per wave: 64 x 64 outputs per phase = acc[2][4][4] (128 registers), 24 ds_read_b128 per
// 64 MFMAs (384 B per MFMA, as the 256 x 256 kernel).  L2 -> LDS fill: 48 KiB per 2 x 128 x 256 x 64 FLOP = 48 B / clk / CU at
// full MFMA rate (the 256 x 256 tile: 32).
//
// Same bits per row as every other GEMM kernel: k ascends in 32-steps through ONE fp32 accumulator per element with the
// same MFMA, and the epilogues are the 256 x 256 kernel's, pass for pass (tests: test_gemm_rows_do_not_depend_on_the_kernel,
// test_lnfold_rows_do_not_depend_on_the_kernel).
#include <type_traits>

#include "../common.h"
#include "../kernels.h"
#include "../lnfold.h"

namespace {

constexpr int BM = 128, BN = 256, BK = 64;
constexpr int UNIT = 128 * BK * 2;      // 16 KiB: 128 rows x 128 B
constexpr int RING = 5 * UNIT;          // 80 KiB: A slots at 0, UNIT; B slots at 2..4 * UNIT
constexpr int SLICE = 16384;            // epilogue staging per wave (inside the idle ring)
constexpr int TABLE_OFF = 4 * SLICE;    // (mean, rstd) of the tile's 128 rows: 1 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ int lds_off_h(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <class TT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt128pp_kernel(GemmArgs p) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    // tile order: N tiles in groups of p.ngroup, inside a group M-major / N-minor over an XCD-contiguous logical id (gemm256h.hip)
    int m0, n0;
    {
        const int logical = xcd_remap(blockIdx.x, ntiles);
        const int G = (p.ngroup > 0 && p.ngroup < tiles_n) ? p.ngroup : tiles_n;
        int g = logical / (tiles_m * G);
        const int ng = (tiles_n + G - 1) / G;
        if (g > ng - 1) g = ng - 1;
        const int rem = logical - g * tiles_m * G;
        const int gsz = g == ng - 1 ? tiles_n - g * G : G;
        m0 = (rem / gsz) * BM;
        n0 = (g * G + rem % gsz) * BN;
    }

    // ---- DMA sources: wave w moves pieces 4w..4w+3 (8 rows x 128 B each) of whichever unit is requested
    const char* __restrict__ A = (const char*)p.A;
    const char* __restrict__ B = (const char*)p.B;
    unsigned a0, a1, a2, a3, b0;
    {
        const int prow = lane >> 3;
        const int schunk = (lane & 7) ^ prow;
        auto arow = [&](int j) { int r = m0 + wid * 32 + 8 * j + prow; return r < p.M ? r : p.M - 1; };
        a0 = (unsigned)arow(0) * (unsigned)p.lda * 2u + schunk * 16;
        a1 = (unsigned)arow(1) * (unsigned)p.lda * 2u + schunk * 16;
        a2 = (unsigned)arow(2) * (unsigned)p.lda * 2u + schunk * 16;
        a3 = (unsigned)arow(3) * (unsigned)p.lda * 2u + schunk * 16;
        b0 = (unsigned)(n0 + wid * 32 + prow) * (unsigned)p.ldb * 2u + schunk * 16;
    }
    const unsigned bstep = 16u * (unsigned)p.ldb;     // 8 rows, bytes
    const unsigned bhalf = 256u * (unsigned)p.ldb;    // 128 rows (B_lo -> B_hi), bytes
    const int piece = wid * 4096;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
#define ISSUE_A(so, kt, q) DMA16(A + (size_t)((kt) * (BK * 2)) + ((q) == 0 ? a0 : (q) == 1 ? a1 : (q) == 2 ? a2 : a3), smem + (so) + piece + (q) * 1024)
#define ISSUE_B(so, kt, q, hi) DMA16(B + (size_t)((kt) * (BK * 2) + (q) * bstep + (hi) * bhalf) + b0, smem + (so) + piece + (q) * 1024)
#define ISSUE_UNIT_A(so, kt) { ISSUE_A(so, kt, 0); ISSUE_A(so, kt, 1); ISSUE_A(so, kt, 2); ISSUE_A(so, kt, 3); }
#define ISSUE_UNIT_B(so, kt, hi) { ISSUE_B(so, kt, 0, hi); ISSUE_B(so, kt, 1, hi); ISSUE_B(so, kt, 2, hi); ISSUE_B(so, kt, 3, hi); }
#define SB __builtin_amdgcn_sched_barrier(0);
#define SYNC(cnt)                                                                                            \
    SB                                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #cnt ") lgkmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("" ::: "memory");                                                                           \
    SB

    constexpr bool FOLD = (EPI == EPI_LNFOLD_T || EPI == EPI_LNFOLD_ACT_T);
    // LN folding: thread t < 128 fetches (mean, rstd) of A row m0 + t now, BEFORE the first DMA (so that the hand-counted
    // vmcnt waits below see it as the oldest operation); the epilogue then waits for no memory
    float2 my_rowstat = float2{0.f, 0.f};
    if constexpr (FOLD) {
        if (tid < BM && m0 + tid < p.M) my_rowstat = p.rowstat[m0 + tid];
    }
    // first round of the launch: the second workgroup of each CU (dispatch order: blocks b and b + 256 share a CU while the
    // chip fills; speed only) starts half a tile late, and its successors inherit the phase
    if (p.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }

    f32x4 acc[2][4][4];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ph][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fkc = lane >> 4;
    const int fo0 = lds_off_h(frow, fkc), fo1 = lds_off_h(frow, 4 + fkc);   // k-step 0 / 1 inside a 64-deep tile
    const int xbase = wm * 64 * 128, wbase = wn * 64 * 128;
    typedef typename TT::vec8 frag_t;
    frag_t X0[4], X1[4], W0[4], W1[4];
#define LD(ptr) (*(const frag_t*)(ptr))
#define READ_X(so)                                                                                           \
    {                                                                                                        \
        const char* s_ = smem + (so) + xbase;                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { X0[i] = LD(s_ + fo0 + i * 2048); }                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { X1[i] = LD(s_ + fo1 + i * 2048); }                   \
    }
#define READ_W(so)                                                                                           \
    {                                                                                                        \
        const char* s_ = smem + (so) + wbase;                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { W0[j] = LD(s_ + fo0 + j * 2048); }                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { W1[j] = LD(s_ + fo1 + j * 2048); }                   \
    }
#define MFMA_PHASE(ph)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[ph][i][j] = TT::mfma(W0[j], X0[i], acc[ph][i][j]); \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[ph][i][j] = TT::mfma(W1[j], X1[i], acc[ph][i][j]); \
    }

    const int nt = p.K / BK;   // K tiles, >= 2 (host-checked)
    // ring state of K tile t: A_t in A slot t & 1; B_lo_t = B unit 2t in B slot (2t) % 3, B_hi_t in B slot (2t + 1) % 3
    int sa = 0;                       // byte offset of A_t's slot
    int jlo = 0;                      // (2t) % 3
    auto bslot = [](int j) { return (2 + j) * UNIT; };
    auto inc = [](int j, int d) { j += d; return j >= 3 ? j - 3 : j; };
    ISSUE_UNIT_A(0, 0) ISSUE_UNIT_B(bslot(0), 0, 0) ISSUE_UNIT_B(bslot(1), 0, 1)
    int t = 0;
    for (; t < nt - 1; ++t) {
        const int jhi = inc(jlo, 1), jn = inc(jlo, 2);        // B_hi_t; B_lo_{t+1} (the slot B_hi_{t-1} has left)
        SYNC(4)                                                // X_t: all but B_hi_t landed
        ISSUE_UNIT_A(sa ^ UNIT, t + 1) ISSUE_UNIT_B(bslot(jn), t + 1, 0)
        SB
        READ_X(sa) READ_W(bslot(jlo))
        MFMA_PHASE(0)
        SYNC(8)                                                // Y_t: B_hi_t landed (A_{t+1}, B_lo_{t+1} may be in flight); B_lo_t dead
        ISSUE_UNIT_B(bslot(jlo), t + 1, 1)                     // B_hi_{t+1} = B unit 2t + 3 -> slot (2t + 3) % 3 = (2t) % 3
        SB
        READ_W(bslot(jhi))
        MFMA_PHASE(1)
        sa ^= UNIT;
        jlo = jn;
    }
    {   // last K tile: nothing left to request
        const int jhi = inc(jlo, 1);
        SYNC(4)
        READ_X(sa) READ_W(bslot(jlo))
        MFMA_PHASE(0)
        SYNC(0)
        READ_W(bslot(jhi))
        MFMA_PHASE(1)
    }

    // ---------------- epilogue through this wave's private LDS slice (the ring is idle after one more barrier; every fragment
    // read of this wave has returned before it arrives there: s_barrier itself waits for no counter)
    SB
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    SB
    if (p.alpha) {   // gradient un-scaling (weight gradients of the fp16 loss-scaled backward)
        const float al = *p.alpha;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ph][i][j] *= al;
    }
    if constexpr (FOLD) {
        if (tid < BM) *(float2*)(smem + TABLE_OFF + tid * 8) = my_rowstat;
        __syncthreads();
    }
    char* sl = smem + wid * SLICE;
    const int fq = lane >> 4, efrow = lane & 15;
    const int mb = m0 + wm * 64;                       // first row of this wave's sub-tile (both phases)
    float2 rs_all[4];
    if constexpr (FOLD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rs_all[i] = *(const float2*)(smem + TABLE_OFF + (wm * 64 + 16 * i + efrow) * 8);
    }

    if constexpr (EPI == EPI_STORE_T || EPI == EPI_ACT_T || EPI == EPI_ACTGRAD_T || FOLD) {
        // per phase ONE pass of 64 rows x 64 cols of 16-bit: LDS rows of 128 B, 16-B chunks XOR-swizzled by (row & 7)
        auto stage16 = [&](int ph, int nb, const float4 (&bias4)[4], const float4 (&s4)[4], auto ACTC) {
            constexpr int ACT = decltype(ACTC)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * i + efrow;
                f32x2 v[4][2];
                if constexpr (FOLD) {
                    const float2 rs = rs_all[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j][0] = lnfold_apply2(f32x2{acc[ph][i][j][0], acc[ph][i][j][1]}, rs.x, rs.y, f32x2{s4[j].x, s4[j].y}, f32x2{bias4[j].x, bias4[j].y});
                        v[j][1] = lnfold_apply2(f32x2{acc[ph][i][j][2], acc[ph][i][j][3]}, rs.x, rs.y, f32x2{s4[j].z, s4[j].w}, f32x2{bias4[j].z, bias4[j].w});
                    }
                } else if constexpr (EPI == EPI_ACTGRAD_T) {
                    int mr = mb + row;
                    mr = mr < p.M ? mr : p.M - 1;
                    uint2 u[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = *(const uint2*)((const u16*)p.aux + (size_t)mr * p.ldc + nb + 16 * j + 4 * fq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float pre[4];
                        if (p.aux_f16) unpack4<F16>(u[j], pre); else unpack4<BF16>(u[j], pre);
                        v[j][0] = f32x2{acc[ph][i][j][0] * act_bwd(pre[0], p.act), acc[ph][i][j][1] * act_bwd(pre[1], p.act)};
                        v[j][1] = f32x2{acc[ph][i][j][2] * act_bwd(pre[2], p.act), acc[ph][i][j][3] * act_bwd(pre[3], p.act)};
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j][0] = f32x2{acc[ph][i][j][0], acc[ph][i][j][1]} + f32x2{bias4[j].x, bias4[j].y};
                        v[j][1] = f32x2{acc[ph][i][j][2], acc[ph][i][j][3]} + f32x2{bias4[j].z, bias4[j].w};
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j][0] = act_fwd2<ACT>(v[j][0]);
                    v[j][1] = act_fwd2<ACT>(v[j][1]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 2 * j + (fq >> 1);
                    const uint2 pk = uint2{TT::pack2(v[j][0]), TT::pack2(v[j][1])};
                    *(uint2*)(sl + row * 128 + ((c ^ (row & 7)) << 4) + (fq & 1) * 8) = pk;
                }
            }
        };
        typedef std::integral_constant<int, -1> NoAct;
        typedef std::integral_constant<int, ACT_GELU> Gelu;
        typedef std::integral_constant<int, ACT_QUICKGELU> QuickGelu;
        auto flush16 = [&](int nb, u16* dst) {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 8 * it + (lane >> 3), pc = lane & 7;
                const u32x4_t fv = *(const u32x4_t*)(sl + row * 128 + (pc << 4));
                const int m = mb + row;
                if (m < p.M)
                    __builtin_nontemporal_store(fv, (u32x4_t*)(dst + (size_t)m * p.ldc + nb + ((pc ^ (row & 7)) << 3)));
            }
        };
        constexpr bool ACTIVE = (EPI == EPI_ACT_T || EPI == EPI_LNFOLD_ACT_T);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int nb = n0 + 128 * ph + wn * 64;      // first column of this wave's sub-tile in this phase
            float4 bias4[4], s4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bias4[j] = p.bias ? *(const float4*)(p.bias + nb + 16 * j + 4 * fq) : float4{0.f, 0.f, 0.f, 0.f};
                s4[j] = FOLD ? *(const float4*)(p.ln_s + nb + 16 * j + 4 * fq) : float4{0.f, 0.f, 0.f, 0.f};
            }
            if (EPI == EPI_ACT_T && p.aux) {   // training forward: pre-activation stash first
                stage16(ph, nb, bias4, s4, NoAct());
                flush16(nb, (u16*)p.aux);
            }
            if (!ACTIVE) stage16(ph, nb, bias4, s4, NoAct());
            else if (p.act == ACT_QUICKGELU) stage16(ph, nb, bias4, s4, QuickGelu());
            else stage16(ph, nb, bias4, s4, Gelu());
            flush16(nb, (u16*)p.C);
        }
    } else {
        // fp32 outputs: per phase two passes of 32 rows x 64 cols: LDS rows of 256 B, 16-B chunks XOR-swizzled by (row & 15)
        constexpr bool RESID = (EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN);
        const float beta = RESID ? 1.f : p.beta;
        const float* rsrc = (RESID && p.aux) ? (const float*)p.aux : (const float*)p.C;   // out-of-place residual
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int nb = n0 + 128 * ph + wn * 64;
            float4 bias4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bias4[j] = p.bias ? *(const float4*)(p.bias + nb + 16 * j + 4 * fq) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 res[8];
                if (beta != 0.f) {   // the residual rows of this pass first: 8 coalesced 16-B loads in flight
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = 4 * it + (lane >> 4), pc = lane & 15;
                        const int m = mb + 32 * q + row;
                        res[it] = m < p.M ? *(const float4*)(rsrc + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2))
                                          : float4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * q + ii;
                    const int row = 16 * ii + efrow;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = 4 * j + fq;
                        *(float4*)(sl + row * 256 + ((c ^ (row & 15)) << 4)) =
                            float4{acc[ph][i][j][0] + bias4[j].x, acc[ph][i][j][1] + bias4[j].y, acc[ph][i][j][2] + bias4[j].z,
                                   acc[ph][i][j][3] + bias4[j].w};
                    }
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = 4 * it + (lane >> 4), pc = lane & 15;
                    float4 v = *(const float4*)(sl + row * 256 + (pc << 4));
                    const int m = mb + 32 * q + row;
                    if (beta != 0.f) {
                        v.x = __builtin_fmaf(res[it].x, beta, v.x); v.y = __builtin_fmaf(res[it].y, beta, v.y);
                        v.z = __builtin_fmaf(res[it].z, beta, v.z); v.w = __builtin_fmaf(res[it].w, beta, v.w);
                    }
                    if (m < p.M) {
                        typedef float f32x4_t __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, (f32x4_t*)((float*)p.C + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2)));
                    }
                    if constexpr (EPI == EPI_RESID_LN) {
                        // LN folding: 16-bit copy of the finished row segment + (sum, M2) of its 64 columns (the 16 lanes of the row)
                        const float gs = row16_sum(lnfold_sum4(v.x, v.y, v.z, v.w));
                        const float gq = row16_sum(lnfold_dev4(v.x, v.y, v.z, v.w, gs * (1.0f / 64.0f)));
                        if (m < p.M) {
                            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                            __builtin_nontemporal_store(__builtin_bit_cast(u32x2_t, pack4<TT>(v.x, v.y, v.z, v.w)),
                                                        (u32x2_t*)((u16*)p.x16 + (size_t)m * p.ldx16 + nb + ((pc ^ (row & 15)) << 2)));
                            if (pc == 0) p.stat_out[(size_t)(nb >> 6) * p.stat_ld + m] = float2{gs, gq};
                        }
                    }
                }
            }
        }
    }
#undef DMA16
#undef ISSUE_A
#undef ISSUE_B
#undef ISSUE_UNIT_A
#undef ISSUE_UNIT_B
#undef SB
#undef SYNC
#undef LD
#undef READ_X
#undef READ_W
#undef MFMA_PHASE
}

// N tiles per group (see gemm256h.hip pick_ngroup): a group's B panels stay in an XCD's 4-MiB L2 while the XCD sweeps M panels
// against them; here 64 tiles run concurrently per XCD.
static int pick_ngroup(const GemmArgs& p) {
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("LEAF_GEMM_NGROUP"); forced = e ? atoi(e) : -1; }
    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    if (forced >= 0) return forced < tiles_n ? forced : 0;
    const double a_bytes = (double)p.M * p.K * 2, b_tile = (double)BN * p.K * 2;
    const double rounds = (double)tiles_m * tiles_n / 512.0;
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
hipError_t launch128pp(const GemmArgs& p_in, int epi, hipStream_t s) {
    GemmArgs p = p_in;
    p.ngroup = pick_ngroup(p);
    static int stagger = -2;   // LEAF_GEMM_PP_STAGGER: units of 2,048 cycles; default = half a tile's K loop
    if (stagger == -2) { const char* e = getenv("LEAF_GEMM_PP_STAGGER"); stagger = e ? atoi(e) : -1; }
    p.stagger = stagger >= 0 ? stagger : (p.K / BK) / 2 + 1;
    const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
#define LEAF_CASE(E)                                                                                         \
    case E: {                                                                                                \
        static bool attr_done = false;                                                                       \
        if (!attr_done) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt128pp_kernel<TT, E>,                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, RING);                     \
            attr_done = true;                                                                                \
        }                                                                                                    \
        hipLaunchKernelGGL((gemm_nt128pp_kernel<TT, E>), dim3(ntiles), dim3(256), RING, s, p);               \
    } break;
    switch (epi) {
        LEAF_CASE(EPI_STORE_T)
        LEAF_CASE(EPI_ACT_T)
        LEAF_CASE(EPI_RESID_F32)
        LEAF_CASE(EPI_STORE_F32)
        LEAF_CASE(EPI_ACTGRAD_T)
        LEAF_CASE(EPI_LNFOLD_T)
        LEAF_CASE(EPI_LNFOLD_ACT_T)
        LEAF_CASE(EPI_RESID_LN)
        default: return hipErrorInvalidValue;
    }
#undef LEAF_CASE
    return hipGetLastError();
}

}  // namespace

// fewest 128 x 256 tiles for which this kernel is dispatched (LEAF_GEMM_PP_MIN_TILES; default = two per CU)
static int g_min_tiles = -1;
void leaf_gemm128pp_set_min_tiles(int n) { g_min_tiles = n; }

bool leaf_gemm128pp_eligible(const GemmArgs& p, int epi) {
    if (g_min_tiles < 0) { const char* e = getenv("LEAF_GEMM_PP_MIN_TILES"); g_min_tiles = e ? atoi(e) : 512; }
    const long tiles = (long)((p.M + BM - 1) / BM) * (p.N / BN);
    // the DMA sources are 32-bit byte offsets from the (uniform) operand bases: both operands must span < 4 GiB
    const bool fits32 = (unsigned long long)p.M * p.lda * 2ull < (1ull << 32) && (unsigned long long)p.N * p.ldb * 2ull < (1ull << 32);
    return epi < EPI_RESID_LN8 && p.N % BN == 0 && tiles >= g_min_tiles && p.K % BK == 0 && p.K >= 2 * BK && p.ldc % 8 == 0 && fits32;
}

hipError_t leaf_launch_gemm128pp(const GemmArgs& p, int dtype, int epi, hipStream_t s) {
    return dtype == LEAF_F16 ? launch128pp<F16>(p, epi, s) : launch128pp<BF16>(p, epi, s);
}
