// Causal multi-head self-attention BACKWARD for context <= 96, head_dim 64 (training rows; no prefix mode).
//
// Reference: autograd of nn.MultiheadAttention with the additive causal mask (src/open_clip/transformer.py:239-252),
// reached from the TextFARE backward (utils_AT.py:321-337).  Given q, k, v (stashed forward qkv) and dO:
//     P = softmax(q k^T / 8 + mask)      dP = dO v^T      dS = P o (dP - rowsum(P o dP)) / 8
//     dq = dS k        dk = dS^T q        dv = P^T dO
//
// One WAVE owns one (sequence, head), like the forward kernel (attention.hip); all five products run on MFMA:
//   * S^T = K Q^T and dP^T = V dO^T (16x16x32, operands straight from HBM in fragment shape): a lane holds 4 consecutive
//     keys of ONE query, so the softmax statistics and rowsum(P o dP) are in-lane plus two wave shuffles;
//   * P and dS of the current 16-query tile go through two 3 KiB LDS tiles [query][key];
//   * dQ^T = K^T dS^T (16x16x32): K^T fragments by the transposing LDS read ds_read_b64_tr_b16 of row-major K;
//   * dK^T += Q^T dS and dV^T += dO^T P (16x16x16, k = the tile's 16 queries): both operands by transposing reads;
//     their accumulators (4 dim-tiles x <= 6 key-tiles x 2) stay in registers across the query-tile loop.
// Types: S uses the FORWARD operand type FT (so P is the forward's P); every gradient product uses the gradient type GT
// (q, k, v are converted once when FT != GT).  dS carries the loss scale of dO.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int HD = 64;
constexpr int MAXT = 6;                      // 16-row tiles -> ctx <= 96
constexpr int X_LD = 72;                     // LDS row stride of the Q / K / dO images (elements): 144 B
constexpr int P_LD = 104;                    // LDS row stride of the P / dS tiles (elements): 208 B
constexpr int X_BYTES = 16 * MAXT * X_LD * 2;
constexpr int P_BYTES = 16 * P_LD * 2;
constexpr int WAVE_LDS = 3 * X_BYTES + 2 * P_BYTES;   // 48,128 B

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// transposing read of a 4-row x 16-column block of a row-major 16-bit LDS image: lane i of each 16-lane group gets
// {X[row0][col0+i], ..., X[row0+3][col0+i]}  (row0 may differ per group)
__device__ __forceinline__ s16x4 tr4(const char* img, int ld_bytes, int row0, int col0, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (row0 + q) * ld_bytes + (col0 + 4 * p) * 2));
}
__device__ __forceinline__ s16x8 tr8(const char* img, int ld_bytes, int row0, int col0, int lane) {
    const s16x4 lo = tr4(img, ld_bytes, row0, col0, lane), hi = tr4(img, ld_bytes, row0 + 4, col0, lane);
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <class FT, class GT>
__device__ __forceinline__ uint4 cvt8(uint4 v) {
    if constexpr (__is_same(FT, GT)) {
        return v;
    } else {
        typename FT::vec8 a = __builtin_bit_cast(typename FT::vec8, v);
        typename GT::vec8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = GT::from_f32(FT::to_f32(a[j]));
        return __builtin_bit_cast(uint4, b);
    }
}

template <class FT, class GT>
__global__ __launch_bounds__(64) void attn_bwd_mfma_kernel(const u16* __restrict__ qkv, const u16* __restrict__ dO,
                                                           u16* __restrict__ dqkv, RowMap map, int heads, int d) {
    leaf_fp16_sat_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const int ld = 3 * d;
    const int sg = map.s0 + n;
    const int row_s = seq_row(map, sg);
    const int L = seq_len(map, sg);
    const int nt = (L + 15) >> 4;
    const int r16 = lane & 15, g = lane >> 4;
    char* qlds = smem;
    char* klds = qlds + X_BYTES;
    char* olds = klds + X_BYTES;
    char* plds = olds + X_BYTES;
    char* slds = plds + P_BYTES;
    const u16* qp = qkv + (size_t)row_s * ld + h * HD;
    const u16* dop = dO + (size_t)row_s * d + h * HD;
    u16* outp = dqkv + (size_t)row_s * ld + h * HD;

    // ---- Q, K (as GT) and dO rows -> LDS, rows >= L zero
    for (int idx = lane; idx < 16 * MAXT * 8; idx += 64) {
        const int row = idx >> 3, ch = idx & 7;
        uint4 q8 = uint4{0u, 0u, 0u, 0u}, k8 = q8, o8 = q8;
        if (row < L) {
            q8 = cvt8<FT, GT>(*(const uint4*)(qp + (size_t)row * ld + ch * 8));
            k8 = cvt8<FT, GT>(*(const uint4*)(qp + (size_t)row * ld + d + ch * 8));
            o8 = *(const uint4*)(dop + (size_t)row * d + ch * 8);
        }
        *(uint4*)(qlds + row * (X_LD * 2) + ch * 16) = q8;
        *(uint4*)(klds + row * (X_LD * 2) + ch * 16) = k8;
        *(uint4*)(olds + row * (X_LD * 2) + ch * 16) = o8;
    }
    // ---- K (FT, for S) and V (GT, for dP) row fragments -> registers
    typename FT::vec8 kf[MAXT][2];
    typename GT::vec8 vf[MAXT][2];
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
        if (kt < nt) {
            int row = kt * 16 + r16; row = row < L ? row : L - 1;
            const u16* kp = qp + (size_t)row * ld + d + g * 8;
            kf[kt][0] = __builtin_bit_cast(typename FT::vec8, *(const uint4*)(kp));
            kf[kt][1] = __builtin_bit_cast(typename FT::vec8, *(const uint4*)(kp + 32));
            vf[kt][0] = __builtin_bit_cast(typename GT::vec8, cvt8<FT, GT>(*(const uint4*)(kp + d)));
            vf[kt][1] = __builtin_bit_cast(typename GT::vec8, cvt8<FT, GT>(*(const uint4*)(kp + d + 32)));
        }
    }
    f32x4 dk[MAXT][4], dv[MAXT][4];
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { dk[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll
    for (int qt = 0; qt < MAXT; ++qt) {
        if (qt < nt) {
            const int qidx = qt * 16 + r16;
            const bool valid = qidx < L;
            const int qv = valid ? qidx : L - 1;
            const u16* qrow = qp + (size_t)qv * ld + g * 8;
            const u16* orow = dop + (size_t)qv * d + g * 8;
            const typename FT::vec8 qf0 = __builtin_bit_cast(typename FT::vec8, *(const uint4*)(qrow));
            const typename FT::vec8 qf1 = __builtin_bit_cast(typename FT::vec8, *(const uint4*)(qrow + 32));
            const typename GT::vec8 of0 = __builtin_bit_cast(typename GT::vec8, *(const uint4*)(orow));
            const typename GT::vec8 of1 = __builtin_bit_cast(typename GT::vec8, *(const uint4*)(orow + 32));
            f32x4 sc[MAXT], dp[MAXT];
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt <= qt; ++kt) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
                a = FT::mfma(kf[kt][0], qf0, a);
                a = FT::mfma(kf[kt][1], qf1, a);
                b = GT::mfma(vf[kt][0], of0, b);
                b = GT::mfma(vf[kt][1], of1, b);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kidx = kt * 16 + 4 * g + e;
                    float s = a[e] * 0.125f;
                    s = kidx > qv ? -INFINITY : s;
                    a[e] = s;
                    m = __builtin_fmaxf(m, s);
                }
                sc[kt] = a;
                dp[kt] = b;
            }
            m = __builtin_fmaxf(m, __shfl_xor(m, 16, 64));
            m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __expf(sc[kt][e] - m);
                    sc[kt][e] = p;
                    sum += p;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = valid ? 1.0f / sum : 0.f;      // padded query rows contribute nothing to dK / dV
            float rd = 0.f;
#pragma unroll
            for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sc[kt][e] *= inv;
                    rd = fmaf(sc[kt][e], dp[kt][e], rd);
                }
            rd += __shfl_xor(rd, 16, 64);
            rd += __shfl_xor(rd, 32, 64);
            // P, dS -> LDS [16 q][keys]; dS zero-filled up to the 32-key k-step boundary
            const int nks = (qt + 2) >> 1;
#pragma unroll
            for (int kt = 0; kt < MAXT; ++kt) {
                if (kt < 2 * nks) {
                    uint2 pk = uint2{0u, 0u}, sk = pk;
                    if (kt <= qt) {
                        pk = pack4<GT>(sc[kt][0], sc[kt][1], sc[kt][2], sc[kt][3]);
                        sk = pack4<GT>(sc[kt][0] * (dp[kt][0] - rd) * 0.125f, sc[kt][1] * (dp[kt][1] - rd) * 0.125f,
                                       sc[kt][2] * (dp[kt][2] - rd) * 0.125f, sc[kt][3] * (dp[kt][3] - rd) * 0.125f);
                    }
                    *(uint2*)(plds + r16 * (P_LD * 2) + (kt * 16 + 4 * g) * 2) = pk;
                    *(uint2*)(slds + r16 * (P_LD * 2) + (kt * 16 + 4 * g) * 2) = sk;
                }
            }
            // dQ^T[dim][q] = sum_key K^T[dim][key] dS^T[key][q]
            f32x4 dq[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) dq[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                if (ks < nks) {
                    const typename GT::vec8 sf = *(const typename GT::vec8*)(slds + r16 * (P_LD * 2) + (ks * 32 + 8 * g) * 2);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const typename GT::vec8 ktf =
                            __builtin_bit_cast(typename GT::vec8, tr8(klds, X_LD * 2, ks * 32 + 8 * g, ct * 16, lane));
                        dq[ct] = GT::mfma(ktf, sf, dq[ct]);
                    }
                }
            }
            if (valid) {
                u16* op = outp + (size_t)qidx * ld + 4 * g;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) *(uint2*)(op + ct * 16) = pack4<GT>(dq[ct][0], dq[ct][1], dq[ct][2], dq[ct][3]);
            }
            // dK^T[dim][key] += sum_q Q^T[dim][q] dS[q][key];   dV^T[dim][key] += sum_q dO^T[dim][q] P[q][key]
            s16x4 qa[4], oa[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                qa[ct] = tr4(qlds, X_LD * 2, qt * 16 + 4 * g, ct * 16, lane);
                oa[ct] = tr4(olds, X_LD * 2, qt * 16 + 4 * g, ct * 16, lane);
            }
#pragma unroll
            for (int kt = 0; kt <= qt; ++kt) {
                const s16x4 sb = tr4(slds, P_LD * 2, 4 * g, kt * 16, lane);
                const s16x4 pb = tr4(plds, P_LD * 2, 4 * g, kt * 16, lane);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    dk[kt][ct] = GT::mfma16(qa[ct], sb, dk[kt][ct]);
                    dv[kt][ct] = GT::mfma16(oa[ct], pb, dv[kt][ct]);
                }
            }
        }
    }
    // ---- dK, dV rows
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
        const int key = kt * 16 + r16;
        if (kt < nt && key < L) {
            u16* op = outp + (size_t)key * ld + 4 * g;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                *(uint2*)(op + d + ct * 16) = pack4<GT>(dk[kt][ct][0], dk[kt][ct][1], dk[kt][ct][2], dk[kt][ct][3]);
                *(uint2*)(op + 2 * d + ct * 16) = pack4<GT>(dv[kt][ct][0], dv[kt][ct][1], dv[kt][ct][2], dv[kt][ct][3]);
            }
        }
    }
}

}  // namespace

hipError_t leaf_launch_attention_bwd_mfma(const void* qkv, int qkv_dtype, const void* dout16, void* dqkv16, int gkind,
                                          int n_seq, RowMap map, int heads, int d, hipStream_t s) {
    if (d != heads * HD || map.ctx > 16 * MAXT || map.ctx < 1 || map.prefix) return hipErrorInvalidValue;
    const dim3 grid(n_seq * heads), blk(64);
#define LEAF_ATTN_BWD(FT, GT)                                                                               \
    do {                                                                                                    \
        static bool attr = false;                                                                           \
        if (!attr) {                                                                                        \
            (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<FT, GT>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, WAVE_LDS);                \
            attr = true;                                                                                    \
        }                                                                                                   \
        hipLaunchKernelGGL((attn_bwd_mfma_kernel<FT, GT>), grid, blk, WAVE_LDS, s, (const u16*)qkv,         \
                           (const u16*)dout16, (u16*)dqkv16, map, heads, d);                                \
    } while (0)
    const bool ff = qkv_dtype == LEAF_F16, gf = gkind == LEAF_F16;
    if (ff && gf) LEAF_ATTN_BWD(F16, F16);
    else if (ff) LEAF_ATTN_BWD(F16, BF16);
    else if (gf) LEAF_ATTN_BWD(BF16, F16);
    else LEAF_ATTN_BWD(BF16, BF16);
#undef LEAF_ATTN_BWD
    return hipGetLastError();
}
