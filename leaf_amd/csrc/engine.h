// Host-side handle shared by api.hip (inference / scoring) and api_train.hip (training step).
#pragma once
#include <string>
#include <vector>

#include "../../include/leaf_hip.h"
#include "kernels.h"

struct TensorInfo {
    std::string name;
    size_t offset;
    int64_t rows, cols;  // cols == 0 -> 1-D
    size_t numel() const { return (size_t)rows * (cols ? cols : 1); }
};

struct LayerOff {  // offsets (floats) into the flat parameter buffer
    size_t ln1_w, ln1_b, qkv_w, qkv_b, out_w, out_b, ln2_w, ln2_b, fc_w, fc_b, proj_w, proj_b;
};

struct leaf_text {
    leaf_text_cfg cfg;
    int fwd_dtype;
    int chunk;  // sequences per pass
    int last_trim;   // last transformer block: attention output / out-proj / MLP only for the pooled (EOT) row
    int grad_dtype;  // 16-bit type of the gradient path: LEAF_F16 (loss-scaled, default) or LEAF_BF16
    int normalize_fare = 0;   // --normalize_fare: the training forward returns F.normalize(features), the backward follows
    float* scaler = nullptr;  // gradient-scaler state (device, leaf_hip.h LEAF_SC_*) or null: no saturation check, no back-off
    // Two-stream chunk pipeline of the forward-only passes (api.hip forward_all): sequence chunks alternate between the
    // caller's stream and a side stream owned by the handle, so one chunk's HBM-bound kernels (LN, attention, the
    // residual epilogues) and grid tails overlap the other chunk's MFMA-bound K loops.  Measured: kernels of the two
    // streams do run concurrently, but the search pass is no faster (57.4 vs 56.0 ms), so the default is streams = 1.
    int streams;
    // QKV GEMM -> attention as one launch in the big forward-only passes (qkv_attn.hip); option 'fuse_attn' / LEAF_FUSE_ATTN=0
    // restores the two kernels around the [rows, 3d] buffer.  The M-tile plan of a pass (first sequence of every tile) is cut on
    // the host and travels through a small ring of pinned buffers (an event per slot guards its re-use).
    int fuse_attn = 1;
    static constexpr int PLAN_RING = 8;
    int32_t* plan_host[PLAN_RING] = {};
    hipEvent_t plan_ev[PLAN_RING] = {};
    size_t plan_cap = 0;      // ints per slot
    int plan_next = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<TensorInfo> tensors;
    std::vector<LayerOff> layer;
    size_t tok_emb, pos_emb, text_proj, lnf_w, lnf_b;
    size_t n_params, n_decay;
    // element offsets into the 16-bit weight pack, per layer: qkv, out, fc, proj (12 d^2 per layer)
    size_t w16_layer_elems() const { return (size_t)12 * cfg.width * cfg.width; }
    size_t w16_qkv(int l) const { return l * w16_layer_elems(); }
    size_t w16_out(int l) const { return w16_qkv(l) + (size_t)3 * cfg.width * cfg.width; }
    size_t w16_fc(int l) const { return w16_out(l) + (size_t)cfg.width * cfg.width; }
    size_t w16_proj(int l) const { return w16_fc(l) + (size_t)4 * cfg.width * cfg.width; }
    // ---- LN folding (lnfold.h), forward-only passes.  The 16-bit forward pack continues behind the standard copies with the
    // gamma-scaled QKV and c_fc weights of every layer (7 d^2 per layer), followed (256-B aligned) by the fp32 vectors
    // s_qkv[3d], c_qkv[3d], s_fc[4d], c_fc[4d] per layer.
    int ln_fold = 1;          // option 'ln_fold' / LEAF_LN_FOLD=0: separate LayerNorm kernels instead
    // Residual stream of the LN-folded forward-only passes in 16 + 8 bits (common.h resid_lo4): the 16-bit copy the next GEMM
    // multiplies anyway + a remainder byte per element (block-scaled e4m3, common.h resid_lo4) instead of an fp32 row beside that copy -- the out-projection moves 8 d bytes
    // per row instead of 12 d, c_proj 14 d instead of 18 d.  Option 'compact_resid' / LEAF_COMPACT_RESID=0: fp32 rows (also what the
    // split blocks of leaf_text_split_pack and widths the fp32 matrix-core projection does not take run on).
    int compact_resid = 1;
    size_t w16_std_elems() const { return w16_layer_elems() * cfg.layers; }
    size_t w16_fold_qkv(int l) const { return w16_std_elems() + (size_t)l * 7 * cfg.width * cfg.width; }
    size_t w16_fold_fc(int l) const { return w16_fold_qkv(l) + (size_t)3 * cfg.width * cfg.width; }
    size_t w16_aux_byte_off() const { return ((w16_std_elems() + (size_t)cfg.layers * 7 * cfg.width * cfg.width) * 2 + 255) / 256 * 256; }
    size_t w16_total_bytes() const { return w16_aux_byte_off() + (size_t)cfg.layers * 14 * cfg.width * 4; }
    // fp32 aux vectors of layer l inside a forward pack at `w16`
    const float* fold_aux(const void* w16, int l) const { return (const float*)((const char*)w16 + w16_aux_byte_off()) + (size_t)l * 14 * cfg.width; }
    const float* fold_s_qkv(const void* w16, int l) const { return fold_aux(w16, l); }
    const float* fold_c_qkv(const void* w16, int l) const { return fold_aux(w16, l) + 3 * cfg.width; }
    const float* fold_s_fc(const void* w16, int l) const { return fold_aux(w16, l) + 6 * cfg.width; }
    const float* fold_c_fc(const void* w16, int l) const { return fold_aux(w16, l) + 10 * cfg.width; }
    // ---- optional higher-precision leading blocks (leaf_text_split_pack): hi + lo operand splits for the four GEMMs of blocks
    // [0, split_blocks) of the forward-only passes.  Caller-owned buffer: per block 27 d^2 16-bit elements -- QKV' [3d][3d] and
    // c_fc' [4d][3d] as [hi | hi | lo] of the gamma-scaled weights, out_proj as [hi | lo] [d][2d] (ONE launch against [A | A]:
    // GemmArgs::a_wrap), the lo half of c_proj [d][4d] (a second launch) -- then (256-B aligned) per block the fp32 row sums
    // s_qkv[3d], s_fc[4d] of hi + lo.
    int split_blocks = 0;
    const void* split_buf = nullptr;
    // WHICH GEMMs of those blocks run on splits: one mask per block (leaf_text_split_pack_masks), bit 0 = QKV (both operands, three
    // passes), bit 1 = out_proj (weights, two passes), bit 2 = c_fc (both operands), bit 3 = c_proj (weights); leaf_text_split_pack =
    // 15 for every block.  profiles/r06_precision_budget_sites.txt and the census price the choices.
    int split_mask[64] = {};
    bool split_on(int l) const { return split_buf && l < split_blocks && l < 64; }
    bool split_qkv(int l) const { return split_on(l) && (split_mask[l] & 1); }
    bool split_out(int l) const { return split_on(l) && (split_mask[l] & 2); }
    bool split_fc(int l) const { return split_on(l) && (split_mask[l] & 4); }
    bool split_proj(int l) const { return split_on(l) && (split_mask[l] & 8); }
    size_t split_block_elems() const { return (size_t)27 * cfg.width * cfg.width; }
    size_t split_aux_byte_off(int n) const { return ((size_t)n * split_block_elems() * 2 + 255) / 256 * 256; }
    size_t split_bytes(int n) const { return split_aux_byte_off(n) + (size_t)n * 7 * cfg.width * 4; }
    const uint16_t* split_qkv3(int l) const { return (const uint16_t*)split_buf + (size_t)l * split_block_elems(); }
    const uint16_t* split_fc3(int l) const { return split_qkv3(l) + (size_t)9 * cfg.width * cfg.width; }
    const uint16_t* split_out2(int l) const { return split_fc3(l) + (size_t)12 * cfg.width * cfg.width; }
    const uint16_t* split_proj_lo(int l) const { return split_out2(l) + (size_t)2 * cfg.width * cfg.width; }
    const float* split_s_qkv(int l) const { return (const float*)((const char*)split_buf + split_aux_byte_off(split_blocks)) + (size_t)l * 7 * cfg.width; }
    const float* split_s_fc(int l) const { return split_s_qkv(l) + 3 * cfg.width; }
};

size_t leaf_train_ws_bytes(const leaf_text* h, int n_seq);  // api_train.hip
// GEMM launch shared by both API files; when the profiler is armed (leaf_prof_begin) each launch is bracketed
// by HIP events on its own stream and accounted under its epilogue id.
// LN folding operands of leaf_gemm (EPI_LNFOLD_* consume rowstat / ln_s, EPI_RESID_LN produces x16 / stat_out)
struct GemmLn {
    const float* ln_s = nullptr;
    const float2* rowstat = nullptr;
    float2* stat_out = nullptr;
    void* x16 = nullptr;
    int stat_ld = 0, ldx16 = 0;
    float eps = 1e-5f;
};
int leaf_gemm(int dtype, int epi, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
              void* aux, int M, int N, int K, int act, hipStream_t s, float beta = 0.f, int aux_f16 = 0,
              const float* alpha = nullptr, const GemmLn* ln = nullptr, int a_wrap = 0);
int leaf_qkv_attn(const QkvAttnArgs& a, int dtype, hipStream_t s);   // leaf_launch_qkv_attn + profiler accounting
void leaf_set_error(const char* fmt, ...);
int leaf_check(hipError_t e, const char* what);

#define LEAF_TRY(expr)                                   \
    do {                                                 \
        int _rc = leaf_check((expr), #expr);             \
        if (_rc) return _rc;                             \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over a caller-provided workspace
struct Carver {
    char* base;
    size_t off, cap;
    Carver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes) {}
    void* take(size_t bytes) {
        size_t o = align_up(off, 256);
        off = o + bytes;
        return base ? base + o : nullptr;
    }
    bool ok() const { return off <= cap; }
};
