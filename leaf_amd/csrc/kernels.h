// Internal launcher interface between the C-ABI layer (api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/leaf_hip.h"
#include "common.h"

enum { LEAF_BF16 = 0, LEAF_F16 = 1 };

enum {
    EPI_STORE_T = 0,    // C16 = acc + bias
    EPI_ACT_T = 1,      // C16 = act(acc + bias); aux16 (optional) = acc + bias (pre-activation stash)
    EPI_RESID_F32 = 2,  // C32 += acc + bias        (residual stream, in place)
    EPI_STORE_F32 = 3,  // C32 = beta*C32 + acc + bias
    EPI_ACTGRAD_T = 4,  // C16 = acc * act'(aux16)
    // ---- LayerNorm folded into the GEMMs of the forward-only passes (lnfold.h): the residual GEMMs also emit the 16-bit
    // copy of the new residual row and its per-64-column statistics; the consuming GEMM multiplies that RAW 16-bit row with
    // gamma-scaled weights and applies mean / rstd in its epilogue.  No LayerNorm kernel, no LayerNorm output in HBM.
    EPI_LNFOLD_T = 5,     // C16 = rstd[m] * (acc - mean[m] * ln_s[n]) + bias[n]   ((mean, rstd) = rowstat[m]; bias = LN-beta . W^T + linear bias)
    EPI_LNFOLD_ACT_T = 6, // C16 = act(the same)
    EPI_RESID_LN = 7,     // C32 += acc + bias;  x16[m,n] = 16-bit(C32);  stat_out[n / 64][m] = (sum, M2) of that row's 64 columns
    // the same on the 16 + 8-bit residual stream (common.h resid_lo4): x = decode(x16, C8) + acc + bias; x16 = 16-bit(x), C8 = its
    // remainder byte, both IN PLACE (C = the [M, ldc] byte matrix of remainders); statistics of x as for EPI_RESID_LN
    EPI_RESID_LN8 = 8,
    EPI_COUNT = 9
};
// profiler / bench key of a GEMM launch: kernel family, operand type, epilogue id
#define LEAF_PROF_KEY(family, dtype, epi) ((family) * 32 + (dtype) * 16 + (epi))
#define LEAF_PROF_BIG 1024     // added to the key of launches of >= 16,384 rows (leaf_prof_end_shapes)


struct GemmArgs {
    const void* A;  // [M,K] 16-bit, row stride lda (elements)
    const void* B;  // [N,K] 16-bit, row stride ldb
    void* C;        // [M,N] 16-bit or fp32, row stride ldc
    const float* bias;  // [N] or null
    void* aux;          // see EPI_*
    int M, N, K, lda, ldb, ldc;
    int act;      // ACT_GELU / ACT_QUICKGELU
    int aux_f16;  // EPI_ACTGRAD_T: aux is fp16 (1) or bf16 (0)
    float beta;
    void* stamps; // diagnostic builds only (-DLEAF_GEMM_STAMPS)
    const float* alpha;  // optional DEVICE scalar multiplying the accumulator (gradient un-scaling), or null
    int ngroup;   // tile order of the 256^2 kernels: N tiles per group (0 = all: M-major / N-minor over the whole matrix)
    // LayerNorm folding (EPI_LNFOLD_* read rowstat / ln_s, EPI_RESID_LN writes x16 / stat_out); the partial statistics are
    // laid out [64-column group][row] with row stride stat_ld, one (sum, M2) pair per (group, row)
    const float* ln_s;        // [N] row sums of the gamma-scaled 16-bit weights
    const float2* rowstat;    // [M] (mean, rstd) of the A rows (leaf_launch_ln_finalize of the producer's partials)
    float2* stat_out;         // [N / 64][stat_ld]
    void* x16;                // [M, N] 16-bit copy of the fp32 output, row stride ldx16
    int stat_ld, ldx16;
    float ln_eps;
    int stagger;  // gemm128pp.hip: start delay (x 2,048 cycles) of the second workgroup of each CU in the first round
    // A operand re-read along K (gemm256h.hip only; leaf_launch_gemm refuses it elsewhere): after a_wrap K tiles of 64 the A panel's k
    // offset restarts at 0, i.e. the kernel multiplies [A | A | ...] [M, K] against B [N, K] with A stored ONCE as [M, 64 a_wrap].
    // 0 = off.  The out-projection of a split block: [A | A] x [W_hi | W_lo]^T in one launch (api.hip).
    int a_wrap;
};

hipError_t leaf_launch_gemm(const GemmArgs& p, int dtype, int epi, hipStream_t s);
void leaf_gemm_set_stamps(void* p);
int leaf_gemm_family(const GemmArgs& p, int epi);   // 0 = gemm_nt_kernel, 1 = gemm_nt256_kernel (gemm.hip), 4 = half-stage ring, 6 = gemm64, 7 = ping-pong
// 256 x 256 tile, 64-deep half-stage LDS-DMA ring with full-line pieces (gemm256h.hip): the kernel of every launch with >= 128 tiles
bool leaf_gemm256h_eligible(const GemmArgs& p, int epi);
void leaf_gemm256h_set_min_tiles(int n);
int leaf_gemm256h_pick_ngroup(const GemmArgs& p);   // N tiles per L2-sized group (0 = one group)
hipError_t leaf_launch_gemm256h(const GemmArgs& p, int dtype, int epi, hipStream_t s);
// 128 x 256 tiles, 4 waves, two workgroups per CU out of phase (variants/gemm128pp.hip: diagnostic builds only, -DLEAF_VARIANTS)
bool leaf_gemm128pp_eligible(const GemmArgs& p, int epi);
void leaf_gemm128pp_set_min_tiles(int n);
hipError_t leaf_launch_gemm128pp(const GemmArgs& p, int dtype, int epi, hipStream_t s);
// round-5 experiment: 256 x 256 tile as FOUR waves of 128 x 128 (variants/gemm256w4.hip: diagnostic builds only, LEAF_GEMM_W4=1)
bool leaf_gemm256w4_eligible(const GemmArgs& p, int epi);
hipError_t leaf_launch_gemm256w4(const GemmArgs& p, int dtype, int epi, hipStream_t s);
// 64 x 128 tiles on a 3-slot LDS-DMA ring for small launches (gemm64.hip)
bool leaf_gemm64_eligible(const GemmArgs& p);
hipError_t leaf_launch_gemm64(const GemmArgs& p, int dtype, int epi, hipStream_t s);

// ---- QKV projection + causal attention in one launch (qkv_attn.hip): the scoring passes' q|k|v rows never reach HBM
#define LEAF_QKVATTN_NCAP 3        // clean captions whose cached prefix K/V one M tile may need (LDS images)
#define LEAF_QKVATTN_CAPROWS 80    // rows per caption image = longest sequence the fused kernel takes (CLIP: 77)
struct QkvAttnArgs {
    const void* A;            // [M, K] 16-bit copy of the residual stream (row stride lda)
    const void* B;            // [3 d, K] gamma-scaled in_proj weights, standard [q; k; v] row order (row stride ldb)
    const float* bias;        // [3 d] c = W . ln_beta + in_proj_bias
    const float* ln_s;        // [3 d] row sums of B
    const float2* rowstat;    // [M] (mean, rstd) of the A rows
    void* out;                // attention output [M, d] 16-bit (eot_pos: [n_seq, d], one row per sequence)
    const void* kv_base;      // this layer's cached q|k|v rows of the clean captions (row stride kv_ld), or null
    const int32_t* eot_pos;   // last-layer mode: pooled position per sequence of the launch, or null
    const int32_t* tile_seq;  // [n_tiles + 1][2] (first sequence, first row), launch-relative, of every M tile (leaf_qkv_attn_plan)
    RowMap map;
    int M, K, lda, ldb, heads, d, n_tiles, n_seq, kv_ld;
    int hsplit;               // head groups an M range is split into across XCDs (1 = an XCD runs all heads of its tiles)
    int ncap, caprows;        // caption images in LDS: how many, rows each (leaf_qkv_attn_lds_plan; the tile plan was cut for ncap)
    void* stamps;             // diagnostic builds only (-DLEAF_GEMM_STAMPS)
};
int leaf_qkv_attn_tile_rows();                       // rows of an M tile of the kernel form in use
int leaf_qkv_attn_ncap(int max_len);                 // captions per tile for sequences of <= max_len positions (0: not taken)
void leaf_qkv_attn_lds_plan(int max_len, int* ncap, int* caprows);
int leaf_qkv_attn_plan(const int32_t* lens, int ctx, int s0, int n, int prefixed, int group, int group_off, int tile_rows, int ncap,
                       int32_t* out);
bool leaf_qkv_attn_eligible(int d, int heads, int ctx, int K, int max_len);
hipError_t leaf_launch_qkv_attn(const QkvAttnArgs& a, int dtype, hipStream_t s);

// ---- forward elementwise / reduction kernels (elementwise.hip)
// x[r,:] = tok_emb[tokens[r],:] + pos_emb[r % ctx,:]   and   xn = LN(x) (16-bit)
hipError_t leaf_launch_embed_ln(const int32_t* tokens, const float* tok_emb, const float* pos_emb, const float* g,
                                const float* b, float eps, float* x, void* xn, int rows, int n_seq, RowMap map, int d,
                                int vocab, int dtype, hipStream_t s, const float* delta = nullptr /* [rows,d] additive embedding perturbation */);
hipError_t leaf_launch_layernorm(const float* x, const float* g, const float* b, float eps, void* xn, int rows, int d,
                                 int dtype, hipStream_t s);
// LN-folded forward (lnfold.h): x = tok_emb[token] + pos_emb[pos] (+ delta), its 16-bit copy and per-64-column (sum, M2);
// lo8: x is the [rows, d] byte matrix of remainders of the 16 + 8-bit residual stream (common.h resid_lo4), no fp32 row is written
hipError_t leaf_launch_embed_fold(const int32_t* tokens, const float* tok_emb, const float* pos_emb, float* x, void* x16,
                                  float2* stat, int stat_ld, int rows, int n_seq, RowMap map, int d, int vocab, int dtype,
                                  hipStream_t s, const float* delta = nullptr, bool lo8 = false, void* split3 = nullptr);
// rowstat[m] = (mean, rstd) of row m from the [ngroups][ld] (sum, M2) partials (Chan merge, lnfold.h)
hipError_t leaf_launch_ln_finalize(const float2* stat, int ld, int rows, int ngroups, float eps, float2* rowstat, hipStream_t s);
// gamma-scaled 16-bit QKV / c_fc weights W'[n,:] = 16-bit(g * W[n,:]) of ALL layers and their vectors s[n] = sum W'[n,:],
// c[n] = b . W[n,:] + bias[n] in one launch; layer l's operands sit l * {w,v,wp,aux}_stride elements behind layer 0's
hipError_t leaf_launch_fold_pack(const float* qkv_w, const float* fc_w, size_t w_stride, const float* ln1_w, const float* ln1_b,
                                 const float* qkv_b, const float* ln2_w, const float* ln2_b, const float* fc_b, size_t v_stride,
                                 void* qkv_p, void* fc_p, size_t wp_stride, float* aux, size_t aux_stride, int d, int layers,
                                 int dtype, hipStream_t s);
// out[n,:] = LN_final(x[n*ctx + eot(n),:]) @ P  (fp32 math), optional L2 normalisation; pooled (optional) keeps
// the normalised EOT row and eot_idx the pooled position (training stash).
// project.hip: out[M,D] = [normalize](LN(xg[M,d]) @ proj[d,D]) on the fp32 matrix-core path; xn = [M,d] fp32 scratch
bool leaf_project_rows_ok(int d, int D);
hipError_t leaf_launch_project_rows(const float* xg, const float* g, const float* b, float eps, const float* proj,
                                    float* xn, float* out, int M, int d, int D, int normalize, hipStream_t s);
hipError_t leaf_launch_pool_project(const float* x, const int32_t* tokens, const float* g, const float* b, float eps,
                                    const float* proj, float* out, float* pooled, int32_t* eot_idx, int n_seq, RowMap map,
                                    int d, int D, int normalize, hipStream_t s, int rows_are_pooled = 0);
// loss[b,r] per objective, arg-max over rho (first maximum wins), best feature gather
hipError_t leaf_launch_score(const float* feat, const float* anchor, int B, int rho, int D, int objective,
                             int32_t* best_idx, float* best_feat, float* loss, hipStream_t s);
// device-to-device byte copy in one launch
hipError_t leaf_launch_copy_bytes(const void* src, void* dst, size_t bytes, hipStream_t s);
// optional higher-precision blocks (hi + lo operand splits over a 3x longer K): out[rows, 3 d] = [hi | lo | hi] of x[rows, d];
// weight rows W'[n,:] = (g .* W)[n,:] as [hi | hi | lo] + s[n] = sum (hi + lo) (triple), or the lo halves alone
hipError_t leaf_launch_split16_rows(const float* x, void* out, int rows, int d, int dtype, hipStream_t s);
hipError_t leaf_launch_dup_cols16(const void* x, void* out, int rows, int d, hipStream_t s);
hipError_t leaf_launch_split16_rows_lo8(const void* x16, const void* lo8, void* out, int rows, int d, int dtype, hipStream_t s);
hipError_t leaf_launch_split_pack(const float* W, const float* g, void* out, float* srow, int N, int K, int triple, int dtype, hipStream_t s);
// fp32 -> 16-bit straight copy (weight packing)
hipError_t leaf_launch_cast(const float* src, void* dst, size_t n, int dtype, hipStream_t s);

// ---- attention (attention.hip): qkv [rows, 3d] 16-bit (q | k | v, heads inside each), out [rows, d]
// kv_base: this layer's qkv rows of the clean captions (prefix mode, map.prefix != null) or null
// eot_pos (device, per sequence of the launch, absolute position) != null: compute only that query row of each sequence
// and write it to out[sequence] (last-layer trimming)
hipError_t leaf_launch_attention_fwd(const void* qkv, const void* kv_base, void* out, int n_seq, RowMap map, int heads,
                                     int d, int dtype, hipStream_t s, const int32_t* eot_pos = nullptr, int max_len = 0);
// eot_pos[n] = first index of the maximum token id among the kept positions (torch argmax pooling)
hipError_t leaf_launch_eot_positions(const int32_t* tokens, int32_t* eot_pos, int n_seq, RowMap map, hipStream_t s);
// out[n,:] = x[row of position eot_pos[n] of sequence n,:]   (fp32)
hipError_t leaf_launch_gather_rows(const float* x, const int32_t* eot_pos, float* out, int n_seq, RowMap map, int d,
                                   hipStream_t s);
// ... and as they are: o16[n,:], o8[n,:] = both halves of that row
hipError_t leaf_launch_gather_rows_pair(const void* x16, const void* lo8, const int32_t* eot_pos, void* o16, void* o8, int n_seq, RowMap map,
                                        int d, hipStream_t s);
// the 16 + 8-bit residual format (common.h resid_lo4), element-wise: fp32 -> (x16, lo8) and back (n % 4 == 0)
hipError_t leaf_launch_resid_pack(const float* x, void* x16, void* lo8, size_t n, int dtype, hipStream_t s);
hipError_t leaf_launch_resid_unpack(const void* x16, const void* lo8, float* x, size_t n, int dtype, hipStream_t s);
// the same out of the 16 + 8-bit residual stream: out[n,:] = decode(x16, lo8) of that row
hipError_t leaf_launch_gather_rows_lo8(const void* x16, const void* lo8, const int32_t* eot_pos, float* out, int n_seq, RowMap map, int d,
                                       int dtype, hipStream_t s);

// ---- training-only kernels (train.hip)
hipError_t leaf_launch_cast16(const void* src, int src_kind, void* dst, int dst_kind, size_t n, hipStream_t s);
// loss = mean_b sum_j (anchor-feat)^2 ; dout = 2 (feat-anchor) / B * scale.  gscale[0] = S, gscale[1] = 1/S: the
// power-of-two loss scale of the fp16 gradient path (S = 1 when use_scaling == 0), chosen so that max|dout| * S ~ 16.
hipError_t leaf_launch_fare_loss(const float* feat, const float* anchor, int B, int D, float scale, float* loss,
                                 float* dout, float* gscale, int use_scaling, hipStream_t s,
                                 const float* norms = nullptr /* --normalize_fare: ||f|| per caption, feat is f / ||f|| */,
                                 const float* scaler = nullptr /* gradient-scaler state (leaf_hip.h): S *= 2^scaler[LEAF_SC_BACKOFF] */);
// fp16 gradient path: raise scaler[LEAF_SC_SAT_FLAG] and poison *poison (gradient element 0) with NaN when any of the n <= 5
// 16-bit tensors holds a saturated (|x| >= 65504) or non-finite value; numel multiples of 8, pointers 16-byte aligned
hipError_t leaf_launch_sat_check16(const void* const* bufs, const size_t* numel, int n, float* scaler, float* poison, hipStream_t s);
hipError_t leaf_launch_normalize_rows(float* x, float* norms, int M, int D, hipStream_t s);
// projection + pooling + final-LN backward: writes dx (fp32 [rows,d], zero except EOT rows), accumulates
// dproj [d,D], dg/db of ln_final.
hipError_t leaf_launch_pool_project_bwd(const float* dout, const float* pooled, const float* x, const int32_t* eot_idx,
                                        const float* g, const float* b, float eps, const float* proj, float* dx,
                                        float* dproj, float* dg, float* db, const float* gscale, int n_seq, RowMap map,
                                        int d, int D, float* scratch /* [n_seq, d + 2] fp32 */, hipStream_t s);
// dx_out = dx_in + LNbwd(dy, x, g); dx16 = 16-bit(dx_out) of kind gkind (optional).  Parameter gradients: with `part` the launch
// writes leaf_ln_bwd_grid(rows, d) per-workgroup partial sums part[wg][2][d] = (sum_r dy xhat | sum_r dy) over the
// workgroup's rows (fixed order, no atomics); leaf_launch_ln_param_reduce adds them to dg / db for many LayerNorms at once.
int leaf_ln_bwd_grid(int rows, int d);
hipError_t leaf_launch_layernorm_bwd(const float* dy, const float* x, const float* g, float eps, float* dx_inout,
                                     void* dx16, int gkind, float* part, int rows, int d, hipStream_t s);
constexpr int LN_REDUCE_MAX = 64;
struct LnReduceArgs {
    const float* part;            // [n][grid][2][d]
    float* dg[LN_REDUCE_MAX];     // dg[i][c] += inv_s * sum_wg part[i][wg][0][c]
    float* db[LN_REDUCE_MAX];     // db[i][c] += inv_s * sum_wg part[i][wg][1][c]
    const float* inv_s;           // device scalar (1 / loss scale)
    int n, grid, d;
};
hipError_t leaf_launch_ln_param_reduce(const LnReduceArgs& a, hipStream_t s);
// attention backward: q,k,v from qkv (fwd dtype), dO [rows,d] -> dqkv [rows,3d], both 16-bit of kind gkind
// grouped TN weight/bias gradient launch (wgrad.hip): up to 4 problems dW[Nw,Kw] += alpha dY^T X, db[Nw] += alpha colsum(dY)
struct WgradProb {
    const uint16_t* dY;   // [rows, Nw] 16-bit gradient type, row stride ldy
    const uint16_t* X;    // [rows, Kw] 16-bit x_dtype, row stride ldx
    float* dW;            // [Nw, Kw] fp32, accumulated into
    float* db;            // [Nw] fp32 or null
    int Nw, Kw, ldy, ldx;
    int tile0, tiles_k;   // filled by the launcher
};
struct WgradArgs {
    WgradProb p[4];
    int nprob, rows;
    const float* alpha;   // device scalar or null (1.0)
    void* stamps;         // diagnostic builds only (-DLEAF_GEMM_STAMPS): 8 s_memtime slots per workgroup, or null
};
void* leaf_gemm_get_stamps();
bool leaf_wgrad_tn_ok(int Nw, int Kw, int ldy, int ldx);
hipError_t leaf_launch_wgrad_group(WgradArgs a, int x_dtype, int g_dtype, hipStream_t s);
// transposed 16-bit copies of all layers' GEMM weights in one launch (train.hip)
hipError_t leaf_launch_pack_transpose(const float* src, void* dst, int dst_kind, int d, int layers, hipStream_t s);
// optional embedding-space PGD mode, train.hip
hipError_t leaf_launch_scale_copy(const float* src, const float* scale_dev, float* dst, size_t n, hipStream_t s);
hipError_t leaf_launch_pgd_step(float* delta, const float* grad, int n_seq, RowMap map, int d, float alpha, float eps,
                                int norm_l2, hipStream_t s);
hipError_t leaf_launch_attention_bwd_mfma(const void* qkv, int qkv_dtype, const void* dout16, void* dqkv16, int gkind,
                                          int n_seq, RowMap map, int heads, int d, hipStream_t s);   // attention_bwd.hip
hipError_t leaf_launch_attention_bwd(const void* qkv, int qkv_dtype, const void* dout16, void* dqkv16, int gkind,
                                     int n_seq, RowMap map, int heads, int d, hipStream_t s);
// dtok[tokens[r],:] += dx[r,:] ; dpos[r % ctx,:] += dx[r,:]
hipError_t leaf_launch_embed_bwd(const float* dx, const float* gscale, const int32_t* tokens, float* dtok, float* dpos,
                                 int rows, int n_seq, RowMap map, int d, int vocab, hipStream_t s);
hipError_t leaf_launch_clip_inplace(float* g, size_t n, float pre_scale, float max_norm, float* ws, hipStream_t s);
hipError_t leaf_launch_adamw(float* p, const float* g, float* m, float* v, size_t n, size_t n_decay, float lr,
                             float beta1, float beta2, float eps, float wd, int step, float grad_scale, hipStream_t s,
                             float max_norm = 0.f /* > 0: clip_grad_norm_ first */, float* clip_ws = nullptr /* [LEAF_SC_WORDS + 2048] fp32 */);
