// Register-direct GEMM epilogues shared by the small-launch kernels (gemm.hip, gemm64.hip): every kernel applies the
// same arithmetic in the same association, so a row's result does not depend on which kernel computed it.
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "lnfold.h"

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
    if constexpr (EPI == EPI_RESID_LN8) {
        // the residual rows come from (and go back to) their 16-bit copy + remainder byte; everything else as EPI_RESID_LN below
        static_assert(EPI != EPI_RESID_LN8 || NJ == 4, "a wave owns whole 64-column groups");
        uint2 rh[NI][NJ];
        unsigned rl[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool ok = m[i] < p.M;
                rh[i][j] = ok ? *(const uint2*)((const u16*)p.x16 + (size_t)m[i] * p.ldx16 + nbase + 16 * j) : uint2{0u, 0u};
                rl[i][j] = ok ? *(const unsigned*)((const unsigned char*)p.C + (size_t)m[i] * p.ldc + nbase + 16 * j) : 0u;
            }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float4 o[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 r = resid_decode4<TT>(rh[i][j], rl[i][j]);
                o[j].x = __builtin_fmaf(r.x, 1.f, acc[i][j][0] + bias[j].x);
                o[j].y = __builtin_fmaf(r.y, 1.f, acc[i][j][1] + bias[j].y);
                o[j].z = __builtin_fmaf(r.z, 1.f, acc[i][j][2] + bias[j].z);
                o[j].w = __builtin_fmaf(r.w, 1.f, acc[i][j][3] + bias[j].w);
            }
            float t[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                t[j] = lnfold_sum4(o[j].x, o[j].y, o[j].z, o[j].w);
                t[j] += __shfl_xor(t[j], 16, 64);
                t[j] += __shfl_xor(t[j], 32, 64);
            }
            const float gs = (t[0] + t[1]) + (t[2] + t[3]);
            const float gm = gs * (1.0f / 64.0f);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                t[j] = lnfold_dev4(o[j].x, o[j].y, o[j].z, o[j].w, gm);
                t[j] += __shfl_xor(t[j], 16, 64);
                t[j] += __shfl_xor(t[j], 32, 64);
            }
            const float gq = (t[0] + t[1]) + (t[2] + t[3]);
            if (m[i] < p.M) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const uint2 hi = pack4<TT>(o[j].x, o[j].y, o[j].z, o[j].w);
                    *(uint2*)((u16*)p.x16 + (size_t)m[i] * p.ldx16 + nbase + 16 * j) = hi;
                    *(unsigned*)((unsigned char*)p.C + (size_t)m[i] * p.ldc + nbase + 16 * j) = resid_lo4<TT>(o[j].x, o[j].y, o[j].z, o[j].w, hi);
                }
                if ((threadIdx.x & 63) < 16) p.stat_out[(size_t)((nbase & ~63) >> 6) * p.stat_ld + m[i]] = float2{gs, gq};
            }
        }
    } else if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32 || EPI == EPI_RESID_LN) {
        constexpr bool RESID = (EPI == EPI_RESID_F32 || EPI == EPI_RESID_LN);
        float4 r[NI][NJ];
        const bool rd = RESID || p.beta != 0.f;
        // residual source: p.aux when given (out-of-place: the training forward keeps both x and x + f(x)), else C itself
        const float* rsrc = (RESID && p.aux) ? (const float*)p.aux : (const float*)p.C;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                r[i][j] = (rd && m[i] < p.M) ? *(const float4*)(rsrc + (size_t)m[i] * p.ldc + nbase + 16 * j)
                                             : float4{0.f, 0.f, 0.f, 0.f};
        const float beta = RESID ? 1.f : p.beta;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float4 o[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // (acc + bias) first, then ONE fused multiply-add with the residual: the same association in every GEMM
                // kernel, so a row's result does not depend on which kernel the tile-count dispatch picked
                o[j].x = __builtin_fmaf(r[i][j].x, beta, acc[i][j][0] + bias[j].x);
                o[j].y = __builtin_fmaf(r[i][j].y, beta, acc[i][j][1] + bias[j].y);
                o[j].z = __builtin_fmaf(r[i][j].z, beta, acc[i][j][2] + bias[j].z);
                o[j].w = __builtin_fmaf(r[i][j].w, beta, acc[i][j][3] + bias[j].w);
                if (m[i] < p.M) *(float4*)((float*)p.C + (size_t)m[i] * p.ldc + nbase + 16 * j) = o[j];
            }
            if constexpr (EPI == EPI_RESID_LN) {
                // LN folding (lnfold.h): 16-bit copy + (sum, M2) of the row's 64-column group.  This lane holds chunk 4 j + fq
                // of the group (fq = lane >> 4): butterfly over the chunk bits 0, 1 (lanes ^ 16, ^ 32), then 2, 3 (registers) --
                // the same tree as the 16-lane row reduction of gemm256h.hip
                static_assert(EPI != EPI_RESID_LN || NJ == 4, "a wave owns whole 64-column groups");
                float t[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    t[j] = lnfold_sum4(o[j].x, o[j].y, o[j].z, o[j].w);
                    t[j] += __shfl_xor(t[j], 16, 64);
                    t[j] += __shfl_xor(t[j], 32, 64);
                }
                const float gs = (t[0] + t[1]) + (t[2] + t[3]);
                const float gm = gs * (1.0f / 64.0f);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    t[j] = lnfold_dev4(o[j].x, o[j].y, o[j].z, o[j].w, gm);
                    t[j] += __shfl_xor(t[j], 16, 64);
                    t[j] += __shfl_xor(t[j], 32, 64);
                }
                const float gq = (t[0] + t[1]) + (t[2] + t[3]);
                if (m[i] < p.M) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        *(uint2*)((u16*)p.x16 + (size_t)m[i] * p.ldx16 + nbase + 16 * j) = pack4<TT>(o[j].x, o[j].y, o[j].z, o[j].w);
                    if ((threadIdx.x & 63) < 16) p.stat_out[(size_t)((nbase & ~63) >> 6) * p.stat_ld + m[i]] = float2{gs, gq};
                }
            }
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
        constexpr bool FOLD = (EPI == EPI_LNFOLD_T || EPI == EPI_LNFOLD_ACT_T);
        constexpr bool ACTIVE = (EPI == EPI_ACT_T || EPI == EPI_LNFOLD_ACT_T);
        // LN folding: (mean, rstd) of this lane's rows, s[n] for its columns
        float2 rs[NI];
        float4 s4[NJ];
        if constexpr (FOLD) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                rs[i] = m[i] < p.M ? p.rowstat[m[i]] : float2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) s4[j] = *(const float4*)(p.ln_s + nbase + 16 * j);
        }
        // the activation id is fixed at compile time inside the element loops (act_fwd_t, common.h)
        auto body = [&](auto ACTC) {
            constexpr int ACT = decltype(ACTC)::value;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (m[i] >= p.M) continue;
                    float v[4];
                    if constexpr (FOLD) {
                        v[0] = lnfold_apply(acc[i][j][0], rs[i].x, rs[i].y, s4[j].x, bias[j].x);
                        v[1] = lnfold_apply(acc[i][j][1], rs[i].x, rs[i].y, s4[j].y, bias[j].y);
                        v[2] = lnfold_apply(acc[i][j][2], rs[i].x, rs[i].y, s4[j].z, bias[j].z);
                        v[3] = lnfold_apply(acc[i][j][3], rs[i].x, rs[i].y, s4[j].w, bias[j].w);
                    } else {
                        v[0] = acc[i][j][0] + bias[j].x; v[1] = acc[i][j][1] + bias[j].y;
                        v[2] = acc[i][j][2] + bias[j].z; v[3] = acc[i][j][3] + bias[j].w;
                    }
                    const size_t o = (size_t)m[i] * p.ldc + nbase + 16 * j;
                    if constexpr (ACTIVE) {
                        if (EPI == EPI_ACT_T && p.aux) *(uint2*)((u16*)p.aux + o) = pack4<TT>(v[0], v[1], v[2], v[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fwd_t<ACT>(v[e]);
                    }
                    *(uint2*)((u16*)p.C + o) = pack4<TT>(v[0], v[1], v[2], v[3]);
                }
        };
        if constexpr (!ACTIVE) body(std::integral_constant<int, -1>());
        else if (p.act == ACT_QUICKGELU) body(std::integral_constant<int, ACT_QUICKGELU>());
        else body(std::integral_constant<int, ACT_GELU>());
    }
}

}  // namespace
