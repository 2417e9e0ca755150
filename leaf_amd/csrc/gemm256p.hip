// NT GEMM, 256x256 tile, PERSISTENT workgroups: the 4-slot LDS-DMA ring of gemm256.hip keeps running across tile
// seams.  One workgroup per CU walks its tiles; the three stages in flight ahead of the MFMAs simply roll over into
// the next tile, so a tile's first stages land while the previous tile's epilogue runs (tools/gemm_stamps.py measured
// 10-14 % of every non-persistent tile waiting for its first stage, plus the launch tail).  The epilogue re-shapes
// each wave's sub-tile through a private 4 KiB slice of the 32 KiB of LDS the ring leaves free (160 KiB total).
// Everything else (stage layout, swizzle, software-pipelined fragment reads, spread DMA issue, wave tiling) is the
// ring kernel's; see gemm256.hip for the rationale and measurements.
//
// Tile order: XCD x (workgroups with blockIdx % 8 == x, an observed-not-promised placement used for speed only) owns
// a contiguous range of the M-major/N-minor tile list and its 32 workgroups take consecutive tiles round-robin, so
// the tiles in flight on one XCD share A panels through that XCD's L2.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 256, BN = 256, BKS = 32, NSTAGE = 4;
constexpr int PART = BM * BKS * 2;      // 16 KiB
constexpr int STAGE = 2 * PART;         // 32 KiB
constexpr int RING = NSTAGE * STAGE;    // 128 KiB
constexpr int SLICE = 4096;             // epilogue staging per wave
constexpr int LDS_TOTAL = RING + 8 * SLICE;   // 160 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

template <class TT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt256_persist_kernel(GemmArgs p, int ntiles) {
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
    const int nt = p.K / BKS;                 // stages per tile (even, >= 6)
    const int total = t_count * nt;           // stages this workgroup consumes

    // ---- DMA cursor: stage `issued` (global over the workgroup's tiles) is the next one to request
    const int prow = lane >> 2;
    const int schunk = (lane & 3) ^ (((prow >> 3) & 1) * 3);
    const u16* __restrict__ A = (const u16*)p.A;
    const u16* __restrict__ B = (const u16*)p.B;
    // per-lane 32-bit element offsets inside a tile's panels + wave-uniform 64-bit panel bases (SGPRs): 3 VGPRs of
    // addressing instead of four 64-bit pointers
    const char* abase;
    const char* bbase;
    unsigned aoff0, aoff1;
    const unsigned boff = ((unsigned)(wid * 32 + prow) * (unsigned)p.ldb + schunk * 8) * 2u;
    const unsigned bstep = 16u * (unsigned)p.ldb * 2u;
    int issued = 0, issue_k = 0, issue_tile = 0;
    auto set_issue_tile = [&](int j) {
        const int tile = t_first + j * t_stride;
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        abase = (const char*)(A + (size_t)m0 * p.lda);
        bbase = (const char*)(B + (size_t)n0 * p.ldb);
        int r0 = wid * 32 + prow, r1 = r0 + 16;
        const int last = p.M - 1 - m0;                     // rows past M re-read the last valid row (never stored)
        r0 = r0 < last ? r0 : last; r1 = r1 < last ? r1 : last;
        aoff0 = ((unsigned)r0 * (unsigned)p.lda + schunk * 8) * 2u;
        aoff1 = ((unsigned)r1 * (unsigned)p.lda + schunk * 8) * 2u;
    };
    set_issue_tile(0);
    const int piece = wid * 2048;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
    // one of the four 1-KiB pieces of the next stage (q = 0..3); the cursor advances after the fourth
#define ISSUE_PIECE(q)                                                                                       \
    if (issued < total) {                                                                                    \
        char* sa_ = smem + (issued & 3) * STAGE + piece;                                                     \
        if ((q) == 0) DMA16(abase + issue_k + aoff0, sa_);                                                   \
        if ((q) == 1) DMA16(abase + issue_k + aoff1, sa_ + 1024);                                            \
        if ((q) == 2) DMA16(bbase + issue_k + boff, sa_ + PART);                                             \
        if ((q) == 3) {                                                                                      \
            DMA16(bbase + issue_k + boff + bstep, sa_ + PART + 1024);                                        \
            ++issued; issue_k += 2 * BKS;                                                                    \
            if (issue_k == 2 * p.K) { issue_k = 0; ++issue_tile; if (issue_tile < t_count) set_issue_tile(issue_tile); } \
        }                                                                                                    \
    }
#define ISSUE_STAGE() ISSUE_PIECE(0) ISSUE_PIECE(1) ISSUE_PIECE(2) ISSUE_PIECE(3)

    const int frow = lane & 15, fkc = lane >> 4;
    const int fo = frow * 64 + ((fkc ^ (((frow >> 3) & 1) * 3)) << 4);
    const int xoff = wm * 8192 + fo, woff = PART + wn * 4096 + fo;
    typedef typename TT::vec8 frag_t;
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
#define SB __builtin_amdgcn_sched_barrier(0);
    // retire stage g: of the stages issued after it, min(2, total - 1 - g) may stay in flight (4 DMAs each).  Memory
    // operations of a previous tile's epilogue are younger than those stages' DMAs only when they were issued later, so
    // the counted wait can only be stricter than needed, never looser.
#define SYNC_STAGE(g)                                                                                        \
    SB                                                                                                       \
    {                                                                                                        \
        const int ahead_ = total - 1 - (g);                                                                  \
        if (ahead_ >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                         \
        else if (ahead_ == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                    \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                     \
    }                                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("" ::: "memory");
    // steady-state step: fragments of stage g, second half of the previous stage's MFMAs (covers the LDS latency) with
    // the next DMA stage's four pieces spread in between, then the first half of stage g
#define STEP(PREV, CUR, g)                                                                                   \
    SYNC_STAGE(g)                                                                                            \
    READ_FRAGS(CUR, (g) & 3)                                                                                 \
    SB MROW(PREV, 4, PREV##x4) SB ISSUE_PIECE(0)                                                             \
    SB MROW(PREV, 5, PREV##x5) SB ISSUE_PIECE(1)                                                             \
    SB MROW(PREV, 6, PREV##x6) SB ISSUE_PIECE(2)                                                             \
    SB MROW(PREV, 7, PREV##x7) SB ISSUE_PIECE(3)                                                             \
    SB MROW(CUR, 0, CUR##x0) MROW(CUR, 1, CUR##x1) MROW(CUR, 2, CUR##x2) MROW(CUR, 3, CUR##x3)

    ISSUE_STAGE()
    ISSUE_STAGE()
    ISSUE_STAGE()
    int g = 0;   // stage being consumed (global)
    char* sl = smem + RING + wid * SLICE;

    for (int ti = 0; ti < t_count; ++ti) {
        const int tile = t_first + ti * t_stride;
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // first stage of the tile: nothing of this tile to overlap the fragment reads with
        SYNC_STAGE(g)
        READ_FRAGS(F, g & 3)
        SB ISSUE_PIECE(0) ISSUE_PIECE(1) SB
        MROW(F, 0, Fx0) MROW(F, 1, Fx1) SB ISSUE_PIECE(2) ISSUE_PIECE(3) SB MROW(F, 2, Fx2) MROW(F, 3, Fx3)
        ++g;
        for (int t = 1; t < nt - 1; t += 2) {
            STEP(F, G, g)
            ++g;
            STEP(G, F, g)
            ++g;
        }
        STEP(F, G, g)      // t = nt - 1
        ++g;
        SB MROW(G, 4, Gx4) MROW(G, 5, Gx5) MROW(G, 6, Gx6) MROW(G, 7, Gx7) SB

        // ---------------- epilogue (the ring keeps filling for the next tile meanwhile)
        // lane id re-materialised through an empty asm: keeps the epilogue's per-lane address arithmetic from being
        // hoisted out of the tile loop, where it would sit in registers across the MFMA loop (spills at 256 VGPRs)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int efrow = ln & 15, efq = ln >> 4;
        if (p.alpha) {
            const float al = *p.alpha;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] *= al;
        }
        const int nb = n0 + wn * 64, mb = m0 + wm * 128;
        float4 bias4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bias4[j] = p.bias ? *(const float4*)(p.bias + nb + 16 * j + 4 * efq) : float4{0.f, 0.f, 0.f, 0.f};

        if constexpr (EPI == EPI_STORE_T || EPI == EPI_ACT_T) {
            // four passes of 32 rows x 64 cols of 16-bit: LDS rows of 128 B, 16-B chunks XOR-swizzled by (row & 7)
            auto stage16 = [&](int pass, bool activated) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * pass + ii;
                    const int row = 16 * ii + efrow;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v[4] = {acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z,
                                      acc[i][j][3] + bias4[j].w};
                        if (activated) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], p.act);
                        }
                        const int c = 2 * j + (efq >> 1);
                        *(uint2*)(sl + row * 128 + ((c ^ (row & 7)) << 4) + (efq & 1) * 8) = pack4<TT>(v[0], v[1], v[2], v[3]);
                    }
                }
            };
            auto flush16 = [&](int pass, u16* dst) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = 8 * it + (ln >> 3), pc = ln & 7;
                    const uint4 v = *(const uint4*)(sl + row * 128 + (pc << 4));
                    const int m = mb + 32 * pass + row;
                    if (m < p.M) *(uint4*)(dst + (size_t)m * p.ldc + nb + ((pc ^ (row & 7)) << 3)) = v;
                }
            };
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                if (EPI == EPI_ACT_T && p.aux) {
                    stage16(pass, false);
                    flush16(pass, (u16*)p.aux);
                }
                stage16(pass, EPI == EPI_ACT_T);
                flush16(pass, (u16*)p.C);
            }
        } else {
            // fp32 outputs: eight passes of 16 rows x 64 cols: LDS rows of 256 B, chunks XOR-swizzled by (row & 15)
            const float beta = (EPI == EPI_RESID_F32) ? 1.f : p.beta;
            const float* rsrc = (EPI == EPI_RESID_F32 && p.aux) ? (const float*)p.aux : (const float*)p.C;   // out-of-place residual
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                float4 res[4];
                if (beta != 0.f) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = 4 * it + (ln >> 4), pc = ln & 15;
                        const int m = mb + 16 * pass + row;
                        res[it] = m < p.M ? *(const float4*)(rsrc + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2))
                                          : float4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                {
                    const int i = pass, row = efrow;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = 4 * j + efq;
                        *(float4*)(sl + row * 256 + ((c ^ (row & 15)) << 4)) =
                            float4{acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z,
                                   acc[i][j][3] + bias4[j].w};
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = 4 * it + (ln >> 4), pc = ln & 15;
                    float4 v = *(const float4*)(sl + row * 256 + (pc << 4));
                    const int m = mb + 16 * pass + row;
                    if (beta != 0.f) {
                        v.x = __builtin_fmaf(res[it].x, beta, v.x); v.y = __builtin_fmaf(res[it].y, beta, v.y);
                    v.z = __builtin_fmaf(res[it].z, beta, v.z); v.w = __builtin_fmaf(res[it].w, beta, v.w);
                    }
                    if (m < p.M) *(float4*)((float*)p.C + (size_t)m * p.ldc + nb + ((pc ^ (row & 15)) << 2)) = v;
                }
            }
        }
    }
}

template <class TT>
hipError_t launch256p(const GemmArgs& p, int epi, hipStream_t s) {
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
            (void)hipFuncSetAttribute((const void*)gemm_nt256_persist_kernel<TT, E>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);                \
            attr_done = true;                                                                                \
        }                                                                                                    \
        hipLaunchKernelGGL((gemm_nt256_persist_kernel<TT, E>), dim3(grid), dim3(512), LDS_TOTAL, s, p, ntiles); \
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

hipError_t leaf_launch_gemm256p(const GemmArgs& p, int dtype, int epi, hipStream_t s) {
    return dtype == LEAF_F16 ? launch256p<F16>(p, epi, s) : launch256p<BF16>(p, epi, s);
}
