// Causal multi-head self-attention for context <= 96 (CLIP text: 77), head_dim 64, forward only.
//
// Reference: nn.MultiheadAttention(batch_first) with an additive causal mask
// (src/open_clip/transformer.py:225,239-252,758-764): softmax(q k^T / sqrt(64) + mask) v per head.
//
// One WAVE owns one (sequence, head).  Everything stays on chip:
//   * K fragments (<= 6 key tiles x 2 k-steps) are loaded straight from HBM into registers in MFMA
//     fragment shape (16 B per lane); Q fragments likewise per 16-row query tile.
//   * V goes row-major into LDS once, by LDS-DMA (round 2: global_load_lds into an unpadded, XOR-swizzled image; a third of
//     the kernel's bytes no longer pass through the VGPR ingest path: -6.5 % per launch), and is consumed COLUMN-wise as
//     the MFMA A operand through the gfx950 transposing read ds_read_b64_tr_b16 (no software transpose).
//   * scores are computed TRANSPOSED (S^T = K Q^T, MFMA 16x16x32): a lane then holds 4 consecutive
//     keys of ONE query per tile, so the row softmax is in-lane plus two wave shuffles (xor 16, 32).
//   * O^T = V^T P^T runs on the k = 16 MFMA (16x16x16): its B operand wants, per lane, 4 consecutive keys of one
//     query - exactly what the S^T accumulators hold - so P goes from registers straight into the MFMA (no LDS
//     round trip); the accumulator then holds 4 consecutive head dims of one query -> 8-byte stores.
// Causality skips key tiles above the diagonal; rows >= ctx are clamped on load and never stored.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int HD = 64;          // head dim (all CLIP text towers)
constexpr int MAXT_ALL = 6;     // 16-row tiles -> ctx <= 96; kernels are instantiated for 2..6 tiles (register count
                                // follows the tile count: 140 VGPRs at 6 tiles, 3 waves per SIMD; fewer tiles, more resident waves)
// per-wave LDS: the V image of `vrows` rows (a multiple of 16 covering the longest sequence of the launch, <= 96), rows of
// 128 B (one head's 64 dims) WITHOUT padding: 12,288 B at 96 rows, 6,144 B at 48 - sizing it by the launch's longest sequence
// keeps more waves resident per CU.  The rows arrive by LDS-DMA (global_load_lds: no trip through VGPRs, whose ingest rate
// of ~12 B/clk per CU the K / Q fragment loads already use), eight rows per 1-KiB piece; the 16-byte chunk c of row r sits at
// position c ^ vswz(r), so that the eight rows one half-wave cycle of ds_read_b64_tr_b16 touches hit eight different 32-byte
// bank groups (even rows: groups 0-3, odd rows: 4-7, selected by (r >> 1) & 3).
constexpr int V_ROW = 128;      // bytes per V row in LDS
__host__ __device__ constexpr int wave_lds_bytes(int vrows) { return vrows * V_ROW; }
__device__ __forceinline__ int vswz(int row) { return 2 * ((row >> 1) & 3); }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;


template <bool USE_TR>
__device__ __forceinline__ s16x4 load_vt_frag(const char* vlds, int key0, int dim0, int lane) {
    // A-operand fragment (k = 16) of V^T: lane holds V[key0 + 4*(lane>>4) + j][dim0 + (lane&15)], j = 0..3
    const int g = lane >> 4, i = lane & 15;
    if constexpr (USE_TR) {
        // ds_read_b64_tr_b16: per 16-lane group a 4x16 block; lane 4q+p supplies row q, cols 4p..4p+3,
        // lane i receives column i of the 4 rows.
        const int q = i >> 2, p = i & 3;
        const int row = key0 + 4 * g + q;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(vlds + row * V_ROW + ((((dim0 >> 3) + (p >> 1)) ^ vswz(row)) << 4) + (p & 1) * 8));
    } else {
        s16x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = key0 + 4 * g + j, col = dim0 + i;
            r[j] = *(const short*)(vlds + row * V_ROW + (((col >> 3) ^ vswz(row)) << 4) + (col & 7) * 2);
        }
        return r;
    }
}

template <class TT, bool USE_TR, int MAXT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const u16* __restrict__ qkv, const u16* __restrict__ kv_base,
                                                       u16* __restrict__ out, int n_items, RowMap map, int heads, int d,
                                                       const int32_t* __restrict__ eot_pos, int vrows) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wid;
    if (item >= n_items) return;   // whole wave exits together (item is wave-uniform)
    char* vlds = smem + wid * wave_lds_bytes(vrows);
    const int n = item / heads, h = item % heads;
    const int ld = 3 * d;
    const int sg = map.s0 + n;
    const int row_s = seq_row(map, sg);
    const int pfx = seq_prefix(map, sg);            // positions < pfx: K/V from the clean caption's cache
    int ctx = pfx + seq_len(map, sg);               // total length of this sequence
    ctx = ctx < vrows ? ctx : vrows;                // never past the LDS image (the host sizes vrows from the true maximum)
    const u16* own = qkv + (size_t)row_s * ld + h * HD;
    const u16* cached = pfx ? kv_base + (size_t)map.base_cu[(sg - map.group_off) / map.group] * ld + h * HD : own;
#ifdef LEAF_DIAG_ATTN_L2   // diagnostic only (garbage results): every q / k / v row comes out of the first 512 rows of the launch's qkv buffer
                           // (2.4 MB: resident in every XCD's L2) -- attention as if its operands never left the chip
    auto rowptr = [&](int pos) { return qkv + (size_t)((row_s + pos) & 511) * ld + h * HD; };
    (void)cached;
#else
    auto rowptr = [&](int pos) { return pos < pfx ? cached + (size_t)pos * ld : own + (size_t)(pos - pfx) * ld; };
#endif
    const int r16 = lane & 15, g = lane >> 4;
    const int nt = (ctx + 15) >> 4;
    // query tiles below the prefix are not needed; with eot_pos (last layer: only the pooled row is consumed downstream)
    // just the tile holding that position, whose row is written to out[sequence]
    const int eot = eot_pos ? eot_pos[n] : -1;
    const int qt0 = eot >= 0 ? eot >> 4 : pfx >> 4;
    const int qt1 = eot >= 0 ? eot >> 4 : nt - 1;

    // ---- K fragments -> registers (straight from global memory; routing them through the LDS image by DMA as well measured
    // the same: 79.0 against 79.4 us per launch)
    typename TT::vec8 kf[MAXT][2];
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
        if (kt < nt) {
            int row = kt * 16 + r16; row = row < ctx ? row : ctx - 1;
            const u16* kp = rowptr(row) + d + g * 8;
#ifdef LEAF_DIAG_NOPREFIX   // diagnostic only (wrong results): the cached prefix's K / V are never fetched -- an upper bound on what
                            // sharing a caption's prefix across its candidates in LDS could save
            if (row < pfx) { kf[kt][0] = typename TT::vec8{}; kf[kt][1] = typename TT::vec8{}; } else
#endif
            {
            kf[kt][0] = __builtin_bit_cast(typename TT::vec8, *(const uint4*)(kp));
            kf[kt][1] = __builtin_bit_cast(typename TT::vec8, *(const uint4*)(kp + 32));
            }
        }
    }
    // ---- V rows -> LDS by LDS-DMA: piece pc = rows 8 pc .. 8 pc + 7, lane -> (row 8 pc + lane / 8, position lane % 8) fetches the
    // source chunk (lane % 8) ^ vswz(row).  Rows ctx .. nt*16-1 of the last tile are zeroed FIRST (plain stores, no DMA in flight
    // yet; P is zero there, but 0 x stale NaN would not be); lanes of rows >= ctx are masked off in the DMA.  The wait for the DMA
    // (explicit vmcnt(0)) sits in front of the first transposing read of V, i.e. after Q K^T and the softmax of a query tile.
    // Rows beyond the sequence's own last 16-row tile are never read (key tiles kt <= qt < nt).
    {
        const int vr = lane >> 3, vc = lane & 7;
        for (int key = ctx + vr; key < nt * 16; key += 8) *(uint4*)(vlds + key * V_ROW + vc * 16) = uint4{0u, 0u, 0u, 0u};
        for (int pc = 0; pc < 2 * nt; ++pc) {
            const int key = 8 * pc + vr;
#ifdef LEAF_DIAG_NOPREFIX
            if (key < ctx && key >= pfx)
#else
            if (key < ctx)
#endif
                __builtin_amdgcn_global_load_lds((glb_void_t*)(rowptr(key) + 2 * d + ((vc ^ vswz(key)) << 3)),
                                                 (lds_void_t*)(vlds + pc * 8 * V_ROW), 16, 0, 0);
        }
    }
#pragma unroll
    for (int qt = 0; qt < MAXT; ++qt) {
        if (qt < nt && qt >= qt0 && qt <= qt1) {
            const int qidx = qt * 16 + r16;
            int qv = qidx < ctx ? qidx : ctx - 1;
            qv = qv < pfx ? pfx : qv;
            const u16* qp = rowptr(qv) + g * 8;
            typename TT::vec8 qf0 = __builtin_bit_cast(typename TT::vec8, *(const uint4*)(qp));
            typename TT::vec8 qf1 = __builtin_bit_cast(typename TT::vec8, *(const uint4*)(qp + 32));
            // S^T tiles for key tiles 0..qt
            f32x4 sc[MAXT];
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt <= qt; ++kt) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                a = TT::mfma(kf[kt][0], qf0, a);
                a = TT::mfma(kf[kt][1], qf1, a);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = a[e] * 0.18033688011112042f;   // 1/sqrt(64) * log2(e): the softmax below runs in base 2
                    // causal mask: key tiles below the diagonal tile hold only keys < every query of this tile (a clamped
                    // query row qv >= ctx - 1 >= those keys as well), so only the diagonal tile needs the compare
                    if (kt == qt) s = (kt * 16 + 4 * g + e) > qv ? -INFINITY : s;
                    a[e] = s;
                    m = __builtin_fmaxf(m, s);
                }
                sc[kt] = a;
            }
            m = __builtin_fmaxf(m, __shfl_xor(m, 16, 64));
            m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float p = __builtin_amdgcn_exp2f(sc[kt][e] - m);
                    sc[kt][e] = p;
                    sum += p;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            // v_rcp_f32 (1 ulp) instead of the 11-instruction IEEE division: P is rounded to 16 bits right below; qkv_attn.hip's
            // attention stage must use the same instruction (its outputs are bit-identical to this kernel's: tests/test_gpu_fused_attn.py)
            const float inv = __builtin_amdgcn_rcpf(sum);
            // O^T[dim][q] = sum_key V^T[dim][key] P^T[key][q], 16 keys per MFMA, P from the S^T accumulators
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the V image has landed: explicit, hipcc does NOT order LDS reads behind the DMA's LDS writes by itself (no vmcnt
            // wait in the ISA when K went the same way)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int kt = 0; kt < MAXT; ++kt) {
                if (kt <= qt) {
                    const s16x4 pf = __builtin_bit_cast(
                        s16x4, pack4_bounded<TT>(sc[kt][0] * inv, sc[kt][1] * inv, sc[kt][2] * inv, sc[kt][3] * inv));
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        o[dt] = TT::mfma16(load_vt_frag<USE_TR>(vlds, kt * 16, dt * 16, lane), pf, o[dt]);
                }
            }
            if (eot >= 0 ? qidx == eot : (qidx < ctx && qidx >= pfx)) {
                u16* op = out + (eot >= 0 ? (size_t)n : (size_t)row_s + qidx - pfx) * d + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *(uint2*)(op + dt * 16) = pack4_bounded<TT>(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
            }
        }
    }
}

}  // namespace

hipError_t leaf_launch_attention_fwd(const void* qkv, const void* kv_base, void* out, int n_seq, RowMap map, int heads,
                                     int d, int dtype, hipStream_t s, const int32_t* eot_pos, int max_len) {
    if (d != heads * HD || map.ctx > 16 * MAXT_ALL || map.ctx < 1) return hipErrorInvalidValue;
    if (max_len <= 0 || max_len > map.ctx) max_len = map.ctx;
    const int vrows = (max_len + 15) / 16 * 16;
    static int use_tr = -1;
    if (use_tr < 0) {
        const char* e = getenv("LEAF_ATTN_TR");
        use_tr = (e && e[0] == '0') ? 0 : 1;
    }
    const int items = n_seq * heads;
    const dim3 grid((items + 3) / 4), blk(256);
    const size_t lds = 4 * (size_t)wave_lds_bytes(vrows);
    const size_t lds_max = 4 * (size_t)wave_lds_bytes(96);
#define LEAF_ATTN(TT, TR, MT)                                                                               \
    do {                                                                                                    \
        static bool attr = false;                                                                           \
        if (!attr) {                                                                                        \
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<TT, TR, MT>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);            \
            attr = true;                                                                                    \
        }                                                                                                   \
        hipLaunchKernelGGL((attn_fwd_kernel<TT, TR, MT>), grid, blk, lds, s, (const u16*)qkv, (const u16*)kv_base,  \
                           (u16*)out, items, map, heads, d, eot_pos, vrows);                                \
    } while (0)
#define LEAF_ATTN_T(TT, TR)                                                                                 \
    switch (vrows / 16) {                                                                                   \
        case 1: case 2: LEAF_ATTN(TT, TR, 2); break;                                                        \
        case 3: LEAF_ATTN(TT, TR, 3); break;                                                                \
        case 4: LEAF_ATTN(TT, TR, 4); break;                                                                \
        case 5: LEAF_ATTN(TT, TR, 5); break;                                                                \
        default: LEAF_ATTN(TT, TR, 6); break;                                                               \
    }
    if (dtype == LEAF_F16) { if (use_tr) { LEAF_ATTN_T(F16, true); } else { LEAF_ATTN_T(F16, false); } }
    else                   { if (use_tr) { LEAF_ATTN_T(BF16, true); } else { LEAF_ATTN_T(BF16, false); } }
#undef LEAF_ATTN_T
#undef LEAF_ATTN
    return hipGetLastError();
}
