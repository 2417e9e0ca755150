// LayerNorm folded into the GEMMs that surround it (forward-only passes: anchor, K/V cache, candidate scoring).
//
//   y[m,n] = sum_k LN(x)[m,k] W[n,k] + bias[n],   LN(x)[m,k] = (x[m,k] - mean_m) rstd_m g[k] + b[k]
//          = rstd_m * ( sum_k x[m,k] (g[k] W[n,k])  -  mean_m * sum_k g[k] W[n,k] )  +  ( sum_k b[k] W[n,k] + bias[n] )
//          = rstd_m * ( acc[m,n] - mean_m * s[n] ) + c[n]
//
// with acc from the MFMA on the RAW residual row rounded to 16 bits and the weights W' = 16-bit(g W) (s[n] = sum_k W'[n,k] of
// the ROUNDED weights, so the mean term cancels exactly what the accumulator holds).  The residual GEMM that produces x
// (out_proj / c_proj) writes that 16-bit copy and the row statistics from its epilogue, so the separate LayerNorm kernel --
// 4d bytes read + 2d written per row, two launches per block -- disappears.  Same rounding budget as rounding LN(x): measured
// with the oracle's operand-rounding emulation on ViT-L 9.2e-4 against 8.8e-4 rel-L2 of the embedding.
//
// Statistics are kept per 64-column group as (sum, M2 = sum (x - group mean)^2) and merged with Chan's formula (equal
// counts), which does not cancel when |mean| >> std, by a tiny kernel (ln_finalize_kernel, elementwise.hip) that turns the
// [group][row] partials into one (mean, rstd) pair per row; the consuming GEMM loads that pair at kernel START (2 registers)
// so its epilogue waits for nothing.  Every producer reduces a group in the SAME tree (4-element chunks
// (x0+x1)+(x2+x3), then a butterfly over the chunk index bits 0,1,2,3) and every consumer merges the groups in ascending
// order with the same expressions, so a row's statistics -- and hence its GEMM output -- do not depend on which kernel or
// tile computed them (the property prefix reuse rests on; the build uses -ffp-contract=off, fused operations are explicit).
#pragma once
#include "common.h"

// (mean, rstd) of row m from its ngroups <= 32 partial (sum, M2) pairs stat[g * ld + m]; all loads are issued before the
// first use (one memory round trip, not one per group)
constexpr int LNFOLD_MAXG = 32;
__device__ __forceinline__ float2 lnfold_row_stat(const float2* __restrict__ stat, int ld, int m, int ngroups, float eps) {
    float2 p[LNFOLD_MAXG];
#pragma unroll
    for (int g = 0; g < LNFOLD_MAXG; ++g) p[g] = g < ngroups ? stat[(size_t)g * ld + m] : float2{0.f, 0.f};
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < LNFOLD_MAXG; ++g) tot += p[g].x;           // + 0 for absent groups: exact
    const float inv_d = 1.0f / (float)(64 * ngroups);
    const float mean = tot * inv_d;
    float m2 = 0.f;
#pragma unroll
    for (int g = 0; g < LNFOLD_MAXG; ++g) {
        const float dm = p[g].x * (1.0f / 64.0f) - mean;
        m2 += g < ngroups ? p[g].y + 64.0f * (dm * dm) : 0.f;
    }
    return float2{mean, 1.0f / sqrtf(m2 * inv_d + eps)};
}

// the consumer's epilogue arithmetic for one element
__device__ __forceinline__ float lnfold_apply(float acc, float mean, float rstd, float s, float c) {
    return __builtin_fmaf(rstd, __builtin_fmaf(-mean, s, acc), c);
}

// ... and for two adjacent columns of one row on v_pk_fma_f32 (the same two roundings per element)
__device__ __forceinline__ f32x2 lnfold_apply2(f32x2 acc, float mean, float rstd, f32x2 s, f32x2 c) {
    return __builtin_elementwise_fma(f32x2{rstd, rstd}, __builtin_elementwise_fma(f32x2{-mean, -mean}, s, acc), c);
}

// chunk-level pieces of the producer's reduction: a lane holds 4 consecutive columns
__device__ __forceinline__ float lnfold_sum4(float a, float b, float c, float d) { return (a + b) + (c + d); }
__device__ __forceinline__ float lnfold_dev4(float a, float b, float c, float d, float mean) {
    const float a0 = a - mean, a1 = b - mean, a2 = c - mean, a3 = d - mean;
    return (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
}
