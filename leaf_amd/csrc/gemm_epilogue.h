// Register-direct GEMM epilogues shared by the small-launch kernels (gemm.hip, gemm64.hip): every kernel applies the
// same arithmetic in the same association, so a row's result does not depend on which kernel computed it.
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

// Epilogue for one accumulator row-group: NJ fragments (columns n0j + 16 j) of NI rows.  Bias is loaded once per
// column group; read-modify-write operands (residual / pre-activation) are fetched for the whole batch BEFORE any
// arithmetic so the loads overlap instead of serialising one L2 round trip per fragment.
template <class TT, int EPI, int NI, int NJ>
__device__ __forceinline__ void epilogue_block(const GemmArgs& p, const int (&m)[NI], int nbase, const float4 (&bias)[NJ],
                                               f32x4 (&acc)[NI][NJ]) {
    if (p.alpha) {   // gradient un-scaling (weight gradients of the fp16 loss-scaled backward)
        const float al = *p.alpha;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] *= al;
    }
    if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) {
        float4 r[NI][NJ];
        const bool rd = (EPI == EPI_RESID_F32) || p.beta != 0.f;
        // residual source: p.aux when given (out-of-place: the training forward keeps both x and x + f(x)), else C itself
        const float* rsrc = (EPI == EPI_RESID_F32 && p.aux) ? (const float*)p.aux : (const float*)p.C;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                r[i][j] = (rd && m[i] < p.M) ? *(const float4*)(rsrc + (size_t)m[i] * p.ldc + nbase + 16 * j)
                                             : float4{0.f, 0.f, 0.f, 0.f};
        const float beta = (EPI == EPI_RESID_F32) ? 1.f : p.beta;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (m[i] >= p.M) continue;
                // (acc + bias) first, then ONE fused multiply-add with the residual: the same association in every GEMM
                // kernel, so a row's result does not depend on which kernel the tile-count dispatch picked
                float4 o;
                o.x = __builtin_fmaf(r[i][j].x, beta, acc[i][j][0] + bias[j].x);
                o.y = __builtin_fmaf(r[i][j].y, beta, acc[i][j][1] + bias[j].y);
                o.z = __builtin_fmaf(r[i][j].z, beta, acc[i][j][2] + bias[j].z);
                o.w = __builtin_fmaf(r[i][j].w, beta, acc[i][j][3] + bias[j].w);
                *(float4*)((float*)p.C + (size_t)m[i] * p.ldc + nbase + 16 * j) = o;
            }
    } else if constexpr (EPI == EPI_ACTGRAD_T) {
        uint2 u[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                u[i][j] = m[i] < p.M ? *(const uint2*)((const u16*)p.aux + (size_t)m[i] * p.ldc + nbase + 16 * j) : uint2{0u, 0u};
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (m[i] >= p.M) continue;
                float pre[4];
                if (p.aux_f16) unpack4<F16>(u[i][j], pre); else unpack4<BF16>(u[i][j], pre);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * act_bwd(pre[e], p.act);
                *(uint2*)((u16*)p.C + (size_t)m[i] * p.ldc + nbase + 16 * j) = pack4<TT>(v[0], v[1], v[2], v[3]);
            }
    } else {
        // the activation id is fixed at compile time inside the element loops (act_fwd_t, common.h)
        auto body = [&](auto ACTC) {
            constexpr int ACT = decltype(ACTC)::value;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (m[i] >= p.M) continue;
                    float v[4] = {acc[i][j][0] + bias[j].x, acc[i][j][1] + bias[j].y, acc[i][j][2] + bias[j].z,
                                  acc[i][j][3] + bias[j].w};
                    const size_t o = (size_t)m[i] * p.ldc + nbase + 16 * j;
                    if constexpr (EPI == EPI_ACT_T) {
                        if (p.aux) *(uint2*)((u16*)p.aux + o) = pack4<TT>(v[0], v[1], v[2], v[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fwd_t<ACT>(v[e]);
                    }
                    *(uint2*)((u16*)p.C + o) = pack4<TT>(v[0], v[1], v[2], v[3]);
                }
        };
        if constexpr (EPI != EPI_ACT_T) body(std::integral_constant<int, -1>());
        else if (p.act == ACT_QUICKGELU) body(std::integral_constant<int, ACT_QUICKGELU>());
        else body(std::integral_constant<int, ACT_GELU>());
    }
}

}  // namespace
