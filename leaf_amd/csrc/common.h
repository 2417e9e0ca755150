// Shared device helpers for the LEAF text-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16;

// 16-bit MFMA operand traits.  Both run v_mfma_f32_16x16x32_* at the same rate; F16 carries
// 11 significand bits (forward / scoring path), BF16 carries fp32's exponent (gradient path).
// MODE.FP16_OVFL = 1 for this wave (hwreg(HW_REG_MODE, 23, 1)): fp16 conversions saturate at +-65504 instead of overflowing to inf
// (tools/fp16_ovfl_probe.hip: 65520.f, 1e6f, 3e38f -> 7bff; +-inf stay inf).  First statement of EVERY kernel: the wave's MODE
// register starts from the kernel descriptor's default (0) and F16::pack2 relies on it.
__device__ __forceinline__ void leaf_fp16_sat_mode() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }

struct F16 {
    using elem = _Float16;
    using vec8 = f16x8;
    using vec4 = f16x4;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    // k = 16 form (operands as raw 16-bit lanes): lane holds A[m = l & 15][k = 4 (l >> 4) + j], B[k = 4 (l >> 4) + j][n = l & 15]
    static __device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ elem from_f32(float x) {
        x = __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f);  // saturate, never inf
        return (_Float16)x;                                           // v_cvt_f16_f32, RTN-even
    }
    static __device__ __forceinline__ float to_f32(elem x) { return (float)x; }
    // two elements: v_cvt_pk_f16_f32 (RTN-even); the saturation at +-65504 is the hardware's -- every kernel sets MODE.FP16_OVFL at
    // entry (leaf_fp16_sat_mode above: an overflowed fp16 result becomes +-MAX_FP16 instead of +-inf, a true inf stays inf) -- so the
    // pair costs one instruction instead of three (v_pk_min_f16 / v_pk_max_f16 behind it: 15 % of the c_fc epilogue's vector
    // instructions, 27 % of the fused launch's q|k|v staging).  Same value as from_f32 for every finite input.
    static __device__ __forceinline__ unsigned pack2(f32x2 v) {
        f16x2 h = {(_Float16)v.x, (_Float16)v.y};
        return __builtin_bit_cast(unsigned, h);
    }
};
struct BF16 {
    using elem = __bf16;
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ elem from_f32(float x) { return (__bf16)x; }  // RTN-even, NaN kept
    static __device__ __forceinline__ float to_f32(elem x) { return (float)x; }
    static __device__ __forceinline__ unsigned pack2(f32x2 v) {
        bf16x2 h = {(__bf16)v.x, (__bf16)v.y};
        return __builtin_bit_cast(unsigned, h);
    }
};

template <class TT>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    return uint2{TT::pack2(f32x2{a, b}), TT::pack2(f32x2{c, d})};
}
// pack4 without the fp16 saturation clamp, for values whose magnitude is bounded by construction (softmax
// probabilities; attention outputs = convex combinations of 16-bit V values): one v_med3 less per element
template <class TT>
__device__ __forceinline__ uint2 pack4_bounded(float a, float b, float c, float d) {
    typename TT::vec4 v;
    v[0] = (typename TT::elem)a; v[1] = (typename TT::elem)b; v[2] = (typename TT::elem)c; v[3] = (typename TT::elem)d;
    return __builtin_bit_cast(uint2, v);
}
template <class TT>
__device__ __forceinline__ void unpack4(uint2 u, float (&o)[4]) {
    typename TT::vec4 v = __builtin_bit_cast(typename TT::vec4, u);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = TT::to_f32(v[i]);
}

// ---- Residual stream of the forward-only passes in 16 + 8 bits (api.hip "compact residual").  A residual value x is kept as
// hi = the 16-bit copy the next GEMM multiplies anyway (x16) and lo8 = its remainder x - hi (exact in fp32, at most half an ulp of
// hi) as an fp8 e4m3 number in a BLOCK-SCALED format: the four values a lane holds (4 consecutive columns) share one power-of-two
// scale -- the exponent of the largest |hi| among them, at least fp16's smallest normal -- and CDNA4's MX conversions do the rest:
// v_cvt_scalef32_pk_fp8_f32 rounds two values to e4m3 (nearest-even) after dividing by the scale operand's power of two,
// v_cvt_scalef32_pk_f32_fp8 multiplies it back (tools/scalef32_probe.hip).  With the remainder pre-multiplied by 2^(MANT + 7) the
// largest one lands on 64, inside e4m3's top binade (steps of 4): every value of the chunk comes back to 2^-(MANT + 6) of the
// chunk's LARGEST magnitude -- 2^-16 for fp16, against fp32's 2^-24 and the 2^-11 of every GEMM operand -- from 3 bytes instead
// of the 6 of an fp32 row beside its 16-bit copy, for ~7 vector instructions per element (encode + decode).
template <class TT> struct ResidLo;
template <> struct ResidLo<F16> { static constexpr float up = 131072.f, down = 1.f / 131072.f, floor = 6.103515625e-05f, xmax = 65520.f; };     // 2^17, 2^-14
template <> struct ResidLo<BF16> { static constexpr float up = 16384.f, down = 1.f / 16384.f, floor = 1.1754943508222875e-38f, xmax = 3.0e38f; };  // 2^14, 2^-126
// the chunk's scale operand: only its exponent field is used by the conversions
__device__ __forceinline__ float resid_scale4(const float (&h)[4], float floor_) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(h[0]), __builtin_fabsf(h[1])), floor_),
                           __builtin_fmaxf(__builtin_fabsf(h[2]), __builtin_fabsf(h[3])));
}
// four residual values + their packed 16-bit copy -> the four remainder bytes (one dword)
template <class TT>
__device__ __forceinline__ unsigned resid_lo4(float a, float b, float c, float d, uint2 hi) {
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    float h[4];
    unpack4<TT>(hi, h);
    const float sc = resid_scale4(h, ResidLo<TT>::floor);
    // (a value beyond the 16-bit type's largest saturates hi; its remainder is cut at half an ulp of that largest: never NaN)
    constexpr float xm = ResidLo<TT>::xmax;
    a = __builtin_amdgcn_fmed3f(a, -xm, xm); b = __builtin_amdgcn_fmed3f(b, -xm, xm);
    c = __builtin_amdgcn_fmed3f(c, -xm, xm); d = __builtin_amdgcn_fmed3f(d, -xm, xm);
    const f32x2 r01 = (f32x2{a, b} - f32x2{h[0], h[1]}) * ResidLo<TT>::up;
    const f32x2 r23 = (f32x2{c, d} - f32x2{h[2], h[3]}) * ResidLo<TT>::up;
    s16x2_t w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, r01.x, r01.y, sc, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, r23.x, r23.y, sc, true);
    return __builtin_bit_cast(unsigned, w);
}
template <class TT>
__device__ __forceinline__ float4 resid_decode4(uint2 hi, unsigned lo) {
    float h[4];
    unpack4<TT>(hi, h);
    const float sc = resid_scale4(h, ResidLo<TT>::floor);
    const f32x2 r01 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, sc, false);
    const f32x2 r23 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(lo, sc, true);
    const f32x2 dn = f32x2{ResidLo<TT>::down, ResidLo<TT>::down};
    const f32x2 x01 = __builtin_elementwise_fma(r01, dn, f32x2{h[0], h[1]});
    const f32x2 x23 = __builtin_elementwise_fma(r23, dn, f32x2{h[2], h[3]});
    return float4{x01.x, x01.y, x23.x, x23.y};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// x + (x of another lane of the same 16-lane row), as ONE VALU instruction (v_add_f32 with a DPP source): quad_perm
// [1,0,3,2] / [2,3,0,1] exchange inside a quad, row_half_mirror reverses each 8-lane half, row_mirror the 16-lane row.
// Applied in that order to a value they form an all-reduce over 16 lanes whose pairing is the xor-butterfly 1, 2, 4, 8
// (after the first two steps every lane of a quad holds the quad sum, so "mirror" pairs quad q with q ^ 1, half h with h ^ 1).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float x) {
    x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);   // row_half_mirror
    x = dpp_add<0x140>(x);   // row_mirror
    return x;
}

enum { ACT_GELU = 0, ACT_QUICKGELU = 1 };

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the 16-bit rounding of the activation) on the hardware
// rcp / exp2: ~12 VALU ops instead of libm erff's ~40 in the c_fc GEMM epilogue (which costs as much as the K = d MFMAs)
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return __builtin_copysignf(1.0f - p * t * e, x);
}

__device__ __forceinline__ float act_fwd(float x, int act) {
    // x * sigmoid(1.702 x) with the hardware exp2 / rcp (1 ulp each): 5 VALU ops instead of a ~20-op IEEE divide
    if (act == ACT_QUICKGELU) return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
    return 0.5f * x * (1.f + fast_erf(x * 0.70710678118654752f));
}
// Two elements at once on the packed-f32 VALU ops (v_pk_mul / v_pk_add / v_pk_fma_f32: two lanes per issue slot; the
// transcendentals stay scalar).  Per element this is the SAME sequence of roundings as act_fwd above (the build uses
// -ffp-contract=off), so a value does not depend on which form a kernel uses.
__device__ __forceinline__ f32x2 fast_erf2(f32x2 x) {
    const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, ax, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2 p = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(t, p, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(t, p, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(t, p, f32x2{0.254829592f, 0.254829592f});
    const f32x2 q = (ax * -1.4426950408889634f) * ax;
    const f32x2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    const f32x2 r = 1.0f - (p * t) * e;
    return f32x2{__builtin_copysignf(r.x, x.x), __builtin_copysignf(r.y, x.y)};
}
template <int ACT>
__device__ __forceinline__ f32x2 act_fwd2(f32x2 x) {
    if constexpr (ACT < 0) return x;
    else if constexpr (ACT == ACT_QUICKGELU) {
        const f32x2 z = x * -2.4554669595930157f;
        const f32x2 u = f32x2{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + 1.f;
        return x * f32x2{__builtin_amdgcn_rcpf(u.x), __builtin_amdgcn_rcpf(u.y)};
    } else {
        return (x * 0.5f) * (fast_erf2(x * 0.70710678118654752f) + 1.f);
    }
}

// compile-time selected form for the GEMM epilogues (ACT < 0: identity): with the activation id a run-time value inside
// the per-element loop the compiler emitted one branch per ELEMENT and a serial mul-exp-add-rcp-mul chain per lane
template <int ACT>
__device__ __forceinline__ float act_fwd_t(float x) {
    if constexpr (ACT < 0) return x;
    else return act_fwd(x, ACT);
}

// derivative of the activation (backward of the MLP: the act'(pre) epilogue of the dgrad GEMM).  Round 5: on the hardware exp2 / rcp
// (1 ulp each) and fast_erf instead of an IEEE division, __expf and libm's erff -- ~12 VALU operations per element instead of
// ~40 (the epilogue of a 256 x 256 tile: 128 elements per lane, two waves per SIMD); the result is rounded to 16 bits right after
__device__ __forceinline__ float act_bwd(float x, int act) {
    if (act == ACT_QUICKGELU) {
        const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));   // sigmoid(1.702 x)
        return s * (1.f + 1.702f * x * (1.f - s));
    }
    const float cdf = 0.5f * (1.f + fast_erf(x * 0.70710678118654752f));
    const float pdf = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x) * 0.3989422804014327f;       // exp(-x^2 / 2) / sqrt(2 pi)
    return cdf + x * pdf;
}

// XCD-aware bijective remap of a 1-D grid: blocks that share an XCD (bid % 8) get a contiguous
// range of logical ids, so neighbouring tiles (same A panel) hit one XCD's L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    int q = nb >> 3, r = nb & 7, x = bid & 7, slot = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
}

// ---- packed (EOT-trimmed) row layout -------------------------------------------------------------------------
// Sequences keep only their first len = eot + 1 rows; cu[s] is the global packed row of sequence s's first row
// (cu == nullptr means dense: cu[s] = s * ctx).  A launch covers sequences [s0, s0 + n) whose rows start at row0.
//
// Prefix reuse: a search candidate shares every position before its first changed token with its
// clean caption, and under the causal mask those rows are identical at every layer.  In prefix mode a sequence
// computes only positions [prefix[s], prefix[s] + len) ("suffix rows", that is what cu counts); attention reads the
// K/V of positions < prefix[s] from the clean caption's per-layer qkv cache (caption index s / group, rows at
// base_cu[.] inside the cache).
struct RowMap {
    const int32_t* cu;
    int s0, row0, ctx;
    const int32_t* prefix;   // device, per sequence, or null
    const int32_t* base_cu;  // device, packed row offsets of the clean captions inside the kv cache
    int group;               // candidates per clean caption
    int group_off;           // sequences [0, group_off) of the pass are the clean captions themselves (fused K/V + scoring pass):
                             // candidate s belongs to caption (s - group_off) / group
};
__device__ __forceinline__ int seq_row(const RowMap& m, int s) { return (m.cu ? m.cu[s] : s * m.ctx) - m.row0; }
__device__ __forceinline__ int seq_len(const RowMap& m, int s) { return m.cu ? m.cu[s + 1] - m.cu[s] : m.ctx; }
__device__ __forceinline__ int seq_prefix(const RowMap& m, int s) { return m.prefix ? m.prefix[s] : 0; }
// sequence owning local row r (n sequences in this launch)
__device__ __forceinline__ int seq_of_row(const RowMap& m, int r, int n) {
    const int R = m.row0 + r;
    if (!m.cu) return R / m.ctx;
    int lo = m.s0, hi = m.s0 + n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (m.cu[mid] <= R) lo = mid; else hi = mid - 1;
    }
    return lo;
}
