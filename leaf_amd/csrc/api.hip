// C ABI (include/leaf_hip.h): handle + flat parameter layout + weight packing + encode_text +
// score_candidates.  Host orchestration only -- every device operation is a launch on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "engine.h"

static thread_local char g_err[512] = "";

void leaf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int leaf_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    leaf_set_error("%s: %s", what, hipGetErrorString(e));
    return 1;
}

extern "C" const char* leaf_last_error(void) { return g_err; }
extern "C" int leaf_version(void) { return 1; }

extern "C" int leaf_text_create(const leaf_text_cfg* cfg, int fwd_dtype, leaf_text_t* out) {
    if (!cfg || !out) { leaf_set_error("null argument"); return 1; }
    const int d = cfg->width;
    if (cfg->layers < 1 || d < 128 || d % 128 || cfg->heads * 64 != d || cfg->embed_dim < 4 || cfg->embed_dim % 4 ||
        cfg->embed_dim > 2048 || d > 2048 || cfg->context_length < 1 || cfg->context_length > 96 ||
        cfg->vocab_size < 2 || (cfg->activation != 0 && cfg->activation != 1) ||
        (fwd_dtype != LEAF_DTYPE_BF16 && fwd_dtype != LEAF_DTYPE_FP16)) {
        leaf_set_error("unsupported text config (need width %% 128 == 0, head_dim 64, ctx <= 96, embed_dim %% 4 == 0)");
        return 1;
    }
    leaf_text* h = new leaf_text();
    h->cfg = *cfg;
    h->fwd_dtype = fwd_dtype;
    h->chunk = 4096;
    h->streams = 1;   // 2 measured no faster on MI355X: the option stays for A/B runs
    { const char* e = getenv("LEAF_LAST_TRIM"); h->last_trim = (e && e[0] == '0') ? 0 : 1; }
    { const char* e = getenv("LEAF_LN_FOLD"); h->ln_fold = (e && e[0] == '0') ? 0 : 1; }
    { const char* e = getenv("LEAF_FUSE_ATTN"); h->fuse_attn = (e && e[0] == '0') ? 0 : 1; }
    { const char* e = getenv("LEAF_COMPACT_RESID"); h->compact_resid = (e && e[0] == '0') ? 0 : 1; }
    {   // gradient path: fp16 + per-step power-of-two loss scale unless LEAF_GRAD_DTYPE=bf16
        const char* e = getenv("LEAF_GRAD_DTYPE");
        h->grad_dtype = (e && (e[0] == 'b' || e[0] == 'B')) ? LEAF_DTYPE_BF16 : LEAF_DTYPE_FP16;
    }
    const int L = cfg->layers, D = cfg->embed_dim;
    size_t off = 0;
    auto add = [&](const std::string& name, int64_t rows, int64_t cols) {
        TensorInfo t{name, off, rows, cols};
        off += t.numel();
        h->tensors.push_back(t);
        return t.offset;
    };
    h->layer.resize(L);
    // ---- weight-decay group (train_AT_text_only.py:323-331)
    h->tok_emb = add("token_embedding.weight", cfg->vocab_size, d);
    h->pos_emb = add("positional_embedding", cfg->context_length, d);
    h->text_proj = add("text_projection", d, D);
    for (int l = 0; l < L; ++l) {
        std::string p = "transformer.resblocks." + std::to_string(l) + ".";
        h->layer[l].qkv_w = add(p + "attn.in_proj_weight", 3 * d, d);
        h->layer[l].out_w = add(p + "attn.out_proj.weight", d, d);
        h->layer[l].fc_w = add(p + "mlp.c_fc.weight", 4 * d, d);
        h->layer[l].proj_w = add(p + "mlp.c_proj.weight", d, 4 * d);
    }
    h->n_decay = off;
    // ---- no-decay group
    for (int l = 0; l < L; ++l) {
        std::string p = "transformer.resblocks." + std::to_string(l) + ".";
        h->layer[l].ln1_w = add(p + "ln_1.weight", d, 0);
        h->layer[l].ln1_b = add(p + "ln_1.bias", d, 0);
        h->layer[l].qkv_b = add(p + "attn.in_proj_bias", 3 * d, 0);
        h->layer[l].out_b = add(p + "attn.out_proj.bias", d, 0);
        h->layer[l].ln2_w = add(p + "ln_2.weight", d, 0);
        h->layer[l].ln2_b = add(p + "ln_2.bias", d, 0);
        h->layer[l].fc_b = add(p + "mlp.c_fc.bias", 4 * d, 0);
        h->layer[l].proj_b = add(p + "mlp.c_proj.bias", d, 0);
    }
    h->lnf_w = add("ln_final.weight", d, 0);
    h->lnf_b = add("ln_final.bias", d, 0);
    h->n_params = off;
    *out = h;
    return 0;
}

extern "C" void leaf_text_destroy(leaf_text_t h) {
    if (!h) return;
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (int i = 0; i < leaf_text::PLAN_RING; ++i) {
        if (h->plan_ev[i]) { (void)hipEventSynchronize(h->plan_ev[i]); (void)hipEventDestroy(h->plan_ev[i]); }
        if (h->plan_host[i]) (void)hipHostFree(h->plan_host[i]);
    }
    delete h;
}

extern "C" int leaf_text_set_chunk(leaf_text_t h, int seqs) {
    if (!h || seqs < 1) { leaf_set_error("bad chunk"); return 1; }
    h->chunk = seqs;
    return 0;
}

extern "C" int leaf_text_set_option(leaf_text_t h, const char* name, int value) {
    if (!h || !name) { leaf_set_error("null argument"); return 1; }
    if (!strcmp(name, "chunk")) return leaf_text_set_chunk(h, value);
    if (!strcmp(name, "last_layer_trim")) { h->last_trim = value ? 1 : 0; return 0; }
    if (!strcmp(name, "streams")) { h->streams = value >= 2 ? 2 : 1; return 0; }
    if (!strcmp(name, "normalize_fare")) { h->normalize_fare = value ? 1 : 0; return 0; }
    if (!strcmp(name, "ln_fold")) {
        const int on = (value && h->cfg.width % 64 == 0) ? 1 : 0;
        if (!on && h->split_blocks) { leaf_set_error("ln_fold = 0 with split blocks active: the split GEMMs are LN-folded (leaf_text_split_pack(.., 0, ..) first)"); return 1; }
        h->ln_fold = on;
        return 0;
    }
    if (!strcmp(name, "fuse_attn")) { h->fuse_attn = value ? 1 : 0; return 0; }
    if (!strcmp(name, "compact_resid")) { h->compact_resid = value ? 1 : 0; return 0; }
    leaf_set_error("unknown option '%s'", name);
    return 1;
}

extern "C" int leaf_text_get_option(leaf_text_t h, const char* name) {
    if (!h || !name) return -1;
    if (!strcmp(name, "chunk")) return h->chunk;
    if (!strcmp(name, "last_layer_trim")) return h->last_trim;
    if (!strcmp(name, "streams")) return h->streams;
    if (!strcmp(name, "normalize_fare")) return h->normalize_fare;
    if (!strcmp(name, "ln_fold")) return h->ln_fold;
    if (!strcmp(name, "fuse_attn")) return h->fuse_attn;
    if (!strcmp(name, "compact_resid")) return h->compact_resid;
    if (!strcmp(name, "split_blocks")) return h->split_blocks;
    return -1;
}

extern "C" size_t leaf_text_param_count(leaf_text_t h) { return h->n_params; }
extern "C" size_t leaf_text_decay_count(leaf_text_t h) { return h->n_decay; }
extern "C" int leaf_text_num_tensors(leaf_text_t h) { return (int)h->tensors.size(); }
extern "C" int leaf_text_param_info(leaf_text_t h, int index, char* name, size_t name_len, size_t* offset,
                                    int64_t* rows, int64_t* cols) {
    if (!h || index < 0 || index >= (int)h->tensors.size()) { leaf_set_error("tensor index out of range"); return 1; }
    const TensorInfo& t = h->tensors[index];
    if (name && name_len) { strncpy(name, t.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (offset) *offset = t.offset;
    if (rows) *rows = t.rows;
    if (cols) *cols = t.cols;
    return 0;
}

extern "C" size_t leaf_text_w16_bytes(leaf_text_t h) { return h->w16_total_bytes(); }

extern "C" size_t leaf_text_split_bytes(leaf_text_t h, int blocks) { return (h && blocks > 0) ? h->split_bytes(blocks) : 0; }

extern "C" int leaf_text_split_pack_masks(leaf_text_t h, const float* params, const int32_t* masks, int blocks, void* buf, leaf_stream_t s_) {
    if (!h) { leaf_set_error("null handle"); return 1; }
    if (blocks == 0) { h->split_blocks = 0; h->split_buf = nullptr; return 0; }
    if (blocks < 0 || blocks > h->cfg.layers - 1 || blocks > 64) { leaf_set_error("split blocks %d out of range 0..%d", blocks, h->cfg.layers - 1); return 1; }
    if (!params || !buf || !masks) { leaf_set_error("leaf_text_split_pack: params / masks / buf is null"); return 1; }
    if (!h->ln_fold) { leaf_set_error("leaf_text_split_pack: the split GEMMs exist in the LN-folded forward only (option ln_fold is 0)"); return 1; }
    for (int l = 0; l < blocks; ++l)
        if (masks[l] < 0 || masks[l] > 15) { leaf_set_error("split mask %d of block %d: bits 0..3 (QKV, out_proj, c_fc, c_proj)", masks[l], l); return 1; }
    hipStream_t s = (hipStream_t)s_;
    const int d = h->cfg.width, dt = h->fwd_dtype;
    h->split_blocks = blocks;
    h->split_buf = buf;
    for (int l = 0; l < 64; ++l) h->split_mask[l] = l < blocks ? masks[l] : 0;
    for (int l = 0; l < blocks; ++l) {      // only what the masks multiply
        const LayerOff& o = h->layer[l];
        if (h->split_qkv(l)) LEAF_TRY(leaf_launch_split_pack(params + o.qkv_w, params + o.ln1_w, (void*)h->split_qkv3(l), (float*)h->split_s_qkv(l), 3 * d, d, 1, dt, s));
        if (h->split_fc(l)) LEAF_TRY(leaf_launch_split_pack(params + o.fc_w, params + o.ln2_w, (void*)h->split_fc3(l), (float*)h->split_s_fc(l), 4 * d, d, 1, dt, s));
        if (h->split_out(l)) LEAF_TRY(leaf_launch_split_pack(params + o.out_w, nullptr, (void*)h->split_out2(l), nullptr, d, d, 2, dt, s));
        if (h->split_proj(l)) LEAF_TRY(leaf_launch_split_pack(params + o.proj_w, nullptr, (void*)h->split_proj_lo(l), nullptr, d, 4 * d, 0, dt, s));
    }
    return 0;
}

extern "C" int leaf_text_split_pack(leaf_text_t h, const float* params, int blocks, void* buf, leaf_stream_t s) {
    int32_t masks[64];
    for (int l = 0; l < 64; ++l) masks[l] = 15;
    return leaf_text_split_pack_masks(h, params, masks, blocks, buf, s);
}

extern "C" int leaf_text_pack_weights(leaf_text_t h, const float* params, void* w16_fwd, void* w16_bwd,
                                      leaf_stream_t s_) {
    hipStream_t s = (hipStream_t)s_;
    const int d = h->cfg.width, L = h->cfg.layers;
    // The fp32 GEMM weights of all layers are one contiguous run (qkv, out, fc, proj per layer = 12 d^2 floats, the
    // layout of leaf_text_create) and the 16-bit packs use the same offsets: one cast launch for the forward copy, one
    // grouped transpose launch for the data-gradient copy (was 48 + 48 launches).
    const float* w32 = params + h->layer[0].qkv_w;
    for (int l = 0; l < L; ++l)
        if (h->layer[l].qkv_w != h->layer[0].qkv_w + (size_t)l * h->w16_layer_elems() ||
            h->layer[l].proj_w != h->layer[l].qkv_w + (size_t)8 * d * d) {
            leaf_set_error("unexpected parameter layout");
            return 1;
        }
    if (w16_fwd) {
        LEAF_TRY(leaf_launch_cast(w32, w16_fwd, h->w16_layer_elems() * L, h->fwd_dtype, s));
        // LN folding (lnfold.h): gamma-scaled QKV / c_fc weights + their s[n], c[n] vectors, all layers in one launch
        const LayerOff& o = h->layer[0];
        const size_t vstride = L > 1 ? h->layer[1].ln1_w - o.ln1_w : 0;
        for (int l = 1; l < L; ++l)
            if (h->layer[l].ln1_w != o.ln1_w + l * vstride || h->layer[l].fc_b != o.fc_b + l * vstride) { leaf_set_error("unexpected parameter layout"); return 1; }
        LEAF_TRY(leaf_launch_fold_pack(params + o.qkv_w, params + o.fc_w, h->w16_layer_elems(), params + o.ln1_w, params + o.ln1_b,
                                       params + o.qkv_b, params + o.ln2_w, params + o.ln2_b, params + o.fc_b, vstride,
                                       (uint16_t*)w16_fwd + h->w16_fold_qkv(0), (uint16_t*)w16_fwd + h->w16_fold_fc(0),
                                       (size_t)7 * d * d, const_cast<float*>(h->fold_aux(w16_fwd, 0)), (size_t)14 * d, d, L,
                                       h->fwd_dtype, s));
    }
    if (w16_bwd) LEAF_TRY(leaf_launch_pack_transpose(w32, w16_bwd, h->grad_dtype, d, L, s));
    return 0;
}

// ------------------------------------------------------------------ forward
// ------------------------------------------------------------------ per-launch GEMM profiler (bench.py roofline)
namespace {
struct ProfRec { hipEvent_t a, b; int key, M, N, K; double flops, bytes; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}  // namespace

int leaf_gemm(int dtype, int epi, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
              void* aux, int M, int N, int K, int act, hipStream_t s, float beta, int aux_f16, const float* alpha,
              const GemmLn* ln, int a_wrap) {
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.act = act; g.aux_f16 = aux_f16; g.beta = beta; g.stamps = nullptr; g.alpha = alpha; g.ngroup = 0;
    g.ln_s = nullptr; g.rowstat = nullptr; g.stat_out = nullptr; g.x16 = nullptr; g.stat_ld = 0; g.ldx16 = 0; g.ln_eps = 0.f; g.stagger = 0; g.a_wrap = a_wrap;
    if (epi == EPI_LNFOLD_T || epi == EPI_LNFOLD_ACT_T || epi == EPI_RESID_LN || epi == EPI_RESID_LN8) {
        const bool fold = epi != EPI_RESID_LN && epi != EPI_RESID_LN8;
        if (!ln || (fold && (!ln->ln_s || !ln->rowstat || !bias)) || (!fold && (!ln->x16 || !ln->stat_out || N % 64 || ln->stat_ld < M))) {
            leaf_set_error("gemm: LN-folding operands missing or misshapen (epilogue %d)", epi);
            return 1;
        }
        g.ln_s = ln->ln_s; g.rowstat = ln->rowstat; g.stat_out = ln->stat_out; g.x16 = ln->x16;
        g.stat_ld = ln->stat_ld; g.ldx16 = ln->ldx16; g.ln_eps = ln->eps;
    }
    if (!g_prof_on) return leaf_check(leaf_launch_gemm(g, dtype, epi, s), "gemm");
    ProfRec r;
    r.key = LEAF_PROF_KEY(leaf_gemm_family(g, epi), dtype, epi);
    r.M = M; r.N = N; r.K = K;
    r.flops = 2.0 * (double)M * (double)N * (double)K;
    {   // algorithmic bytes of the launch: both operands once + the output (+ the fp32 read of a residual/accumulate)
        const bool resid = epi == EPI_RESID_F32 || epi == EPI_RESID_LN;
        const double out_b = (resid || epi == EPI_STORE_F32) ? 4.0 : 2.0;
        const double rmw = (resid || (epi == EPI_STORE_F32 && beta != 0.f)) ? 4.0 : 0.0;
        r.bytes = 2.0 * ((double)M * K + (double)N * K) + (double)M * N * (out_b + rmw) + (epi == EPI_ACTGRAD_T ? 2.0 * M * N : 0.0) +
                  (epi == EPI_RESID_LN ? 2.0 * M * N + 8.0 * M * (N / 64) : 0.0) +                    // 16-bit copy + statistics
                  ((epi == EPI_LNFOLD_T || epi == EPI_LNFOLD_ACT_T) ? 8.0 * M : 0.0);                  // (mean, rstd) per row
    }
    // 16 + 8-bit residual stream: 3 bytes read + 3 written per output element, statistics
    if (epi == EPI_RESID_LN8) r.bytes = 2.0 * ((double)M * K + (double)N * K) + 6.0 * (double)M * N + 8.0 * M * (N / 64);
    LEAF_TRY(hipEventCreate(&r.a));
    LEAF_TRY(hipEventCreate(&r.b));
    LEAF_TRY(hipEventRecord(r.a, s));
    int rc = leaf_check(leaf_launch_gemm(g, dtype, epi, s), "gemm");
    LEAF_TRY(hipEventRecord(r.b, s));
    g_prof.push_back(r);
    return rc;
}

// the fused QKV + attention launch, accounted like a GEMM launch (family 8, epilogue id of the LN-folded QKV GEMM): FLOPs = the
// projection's 2 M 3d K (the attention's own products are not counted), bytes = A + weights + attention output
int leaf_qkv_attn(const QkvAttnArgs& a, int dtype, hipStream_t s) {
    if (!g_prof_on) return leaf_check(leaf_launch_qkv_attn(a, dtype, s), "qkv_attn");
    ProfRec r;
    r.key = LEAF_PROF_KEY(8, dtype, EPI_LNFOLD_T);
    r.M = a.M; r.N = 3 * a.d; r.K = a.K;
    r.flops = 2.0 * (double)a.M * (3.0 * a.d) * (double)a.K;
    r.bytes = 2.0 * ((double)a.M * a.K + 3.0 * a.d * a.K) + 2.0 * (double)(a.eot_pos ? a.n_seq : a.M) * a.d + 8.0 * a.M;
    LEAF_TRY(hipEventCreate(&r.a));
    LEAF_TRY(hipEventCreate(&r.b));
    LEAF_TRY(hipEventRecord(r.a, s));
    int rc = leaf_check(leaf_launch_qkv_attn(a, dtype, s), "qkv_attn");
    LEAF_TRY(hipEventRecord(r.b, s));
    g_prof.push_back(r);
    return rc;
}

extern "C" int leaf_prof_begin(void) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = true;
    return 0;
}

// Suspends (1) / resumes (0) the recording without dropping what has been recorded: a caller that times MANY steps takes the per-launch
// events on a sample of them only (each event pair costs the queue a completion barrier: round 4, 1.2 ms of a 50-ms step)
extern "C" int leaf_prof_pause(int paused) {
    g_prof_on = !paused;
    return 0;
}

// Stops recording, waits for the recorded events and sums per key (= LEAF_PROF_KEY: kernel_family*32 + dtype*16 + epilogue id, < 512):
// ms[key], flops[key], bytes[key] (algorithmic operand + output bytes, may be null), count[key].
extern "C" int leaf_prof_end(double* ms, double* flops, double* bytes, int64_t* count, int n_keys) {
    g_prof_on = false;
    for (int i = 0; i < n_keys; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; if (bytes) bytes[i] = 0; }
    for (auto& r : g_prof) {
        LEAF_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        LEAF_TRY(hipEventElapsedTime(&t, r.a, r.b));
        if (r.key >= 0 && r.key < n_keys) { ms[r.key] += t; flops[r.key] += r.flops; count[r.key] += 1; if (bytes) bytes[r.key] += r.bytes; }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return 0;
}

// Same as leaf_prof_end but grouped by (key, N, K) -- one line per GEMM SHAPE of each kernel, so that e.g. the two
// residual GEMMs (out_proj: N = K = d, HBM-bound; c_proj: K = 4d, MFMA-bound) are reported separately.  info[i] =
// {key (+ LEAF_PROF_BIG = 1024 for launches of >= 16,384 rows), N, K, launches}; rows[i] = sum of M over the launches.  Returns the number of groups in *n_out (<= max_groups).
extern "C" int leaf_prof_end_shapes(double* ms, double* flops, double* bytes, int64_t* rows, int32_t* info, int max_groups,
                                    int* n_out) {
    g_prof_on = false;
    int n = 0;
    for (auto& r : g_prof) {
        LEAF_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        LEAF_TRY(hipEventElapsedTime(&t, r.a, r.b));
        // launches of >= 16,384 rows (the scoring passes) and the small B-caption launches of the same kernel and shape are
        // reported apart: group key = key + LEAF_PROF_BIG for the big ones
        const int gkey = r.key + (r.M >= 16384 ? LEAF_PROF_BIG : 0);
        int i = 0;
        for (; i < n; ++i)
            if (info[4 * i] == gkey && info[4 * i + 1] == r.N && info[4 * i + 2] == r.K) break;
        if (i == n) {
            if (n == max_groups) continue;
            info[4 * n] = gkey; info[4 * n + 1] = r.N; info[4 * n + 2] = r.K; info[4 * n + 3] = 0;
            ms[n] = 0; flops[n] = 0; bytes[n] = 0; rows[n] = 0;
            ++n;
        }
        ms[i] += t; flops[i] += r.flops; bytes[i] += r.bytes; rows[i] += r.M; info[4 * i + 3] += 1;
    }
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    if (n_out) *n_out = n;
    return 0;
}

namespace {

struct FwdBuf {
    float* x;      // [rows,d] fp32 residual stream
    uint16_t* a;   // [rows,d]   LN output / attention output (aliased)
    uint16_t* qkv; // [rows,3d]
    uint16_t* hh;  // [rows,4d]
    int32_t* eot;  // [seqs] pooled position per sequence (last-layer trimming)
    uint16_t* x16; // [rows,d]   LN folding: 16-bit copy of the residual stream (A operand of the QKV / c_fc GEMMs)
    float2* stat;  // [d/64][rows] LN folding: (sum, M2) per row and 64-column group
    float2* rowstat; // [rows]     LN folding: (mean, rstd) per row
    int32_t* tile_seq; // [seqs + 1][2] fused QKV + attention: (first sequence, first row) of every M tile (a sequence has >= 1 row)
};

size_t fwd_chunk_bytes(const leaf_text* h, int cs) {
    const size_t rows = (size_t)cs * h->cfg.context_length, d = h->cfg.width;
    Carver c(nullptr, 0);
    c.take(rows * d * 4); c.take(rows * d * 2); c.take(rows * 3 * d * 2); c.take(rows * 4 * d * 2); c.take(rows * 4);
    c.take(rows * d * 2); c.take(rows * (d / 64) * 8); c.take(rows * 8); c.take((rows + 2) * 8);
    return align_up(c.off, 256);
}

FwdBuf carve_fwd(const leaf_text* h, Carver& c, int cs) {
    const size_t rows = (size_t)cs * h->cfg.context_length, d = h->cfg.width;
    FwdBuf b;
    b.x = (float*)c.take(rows * d * 4);
    b.a = (uint16_t*)c.take(rows * d * 2);
    b.qkv = (uint16_t*)c.take(rows * 3 * d * 2);
    b.hh = (uint16_t*)c.take(rows * 4 * d * 2);
    b.eot = (int32_t*)c.take(rows * 4);   // one per sequence; a sequence has >= 1 row
    b.x16 = (uint16_t*)c.take(rows * d * 2);
    b.stat = (float2*)c.take(rows * (d / 64) * 8);
    b.rowstat = (float2*)c.take(rows * 8);
    b.tile_seq = (int32_t*)c.take((rows + 2) * 8);
    return b;
}

// run the layer stack on `cs` sequences (rows = their packed row count); features -> out [cs, D]
// kv_write: per-layer qkv of this (single) chunk is produced INTO the cache (base captions, stride kv_stride elements per
// layer).  kv_read: prefix mode, attention reads the prefix K/V of layer l from kv_read + l * kv_stride.
struct KvPlan {
    uint16_t* kv_write = nullptr;
    const uint16_t* kv_read = nullptr;
    size_t kv_stride = 0;
    // fused pass (leaf_score_candidates_prefix_fused): the clean captions are the FIRST kv_self_rows rows of this chunk; the
    // candidates' attention reads their K/V from the chunk's own qkv buffer, and every layer's rows are copied to kv_copy
    uint16_t* kv_copy = nullptr;
    size_t kv_self_rows = 0;
    // fused QKV + attention launches (qkv_attn.hip): number of M tiles cut for this chunk (b.tile_seq holds them); 0 = two kernels
    int attn_tiles = 0;
    int attn_max_len = 0;     // longest sequence (prefix + computed rows) the plan was cut for: sizes the caption images
};

int forward_chunk(const leaf_text* h, const float* P, const uint16_t* W, const int32_t* tokens, int cs, int rows,
                  RowMap map, float* out, int normalize, const FwdBuf& b_in, hipStream_t s, const KvPlan& kv = KvPlan(),
                  int max_len = 0 /* longest sequence (prefix + computed rows); 0 = context_length */) {
    FwdBuf b = b_in;
    const leaf_text_cfg& c = h->cfg;
    const int d = c.width, dt = h->fwd_dtype;
    const LayerOff& o0 = h->layer[0];
    // LN folding (lnfold.h): no LayerNorm kernel between the GEMMs.  The embedding kernel and the residual GEMMs (out_proj,
    // c_proj) emit the 16-bit copy of the residual row + its per-group statistics; the QKV / c_fc GEMMs multiply that raw
    // row with the gamma-scaled weights and apply mean / rstd in their epilogue.
    const bool fold = h->ln_fold != 0;
    GemmLn ln;                       // statistics / 16-bit copy of the FULL row set of this chunk
    ln.rowstat = b.rowstat; ln.stat_out = b.stat; ln.stat_ld = rows; ln.x16 = b.x16; ln.ldx16 = d; ln.eps = c.ln_eps;
    // x = residual stream; qkv_gemm / fc_gemm: LN + linear of `m` rows whose (folded) statistics live in `g`
    // optional higher-precision leading blocks (leaf_text_split_pack): hi + lo splits of both operands over a 3x longer K.  The
    // [rows, 3d] split copy of the fp32 residual rows lives in buffers that are dead at that point: the hidden buffer in front of
    // the QKV GEMM, the chunk's own q|k|v scratch in front of c_fc (these blocks run the two-kernel attention path).
    const bool splits = fold && h->split_buf && h->split_blocks > 0;
    auto sq = [&](int l) { return splits && h->split_qkv(l); };
    auto sf = [&](int l) { return splits && h->split_fc(l); };
    // 16 + 8-bit residual stream (engine.h compact_resid): b.x holds the [rows, d] remainder bytes, b.x16 is the other half
    const bool lo8 = fold && h->compact_resid && leaf_project_rows_ok(d, c.embed_dim);
    // [hi | lo | hi] copy of the current residual rows (fp32 rows, or the 16 + 8-bit stream) for a three-pass GEMM; block 0's QKV
    // operand comes out of the embedding kernel (the exact fp32 sum is in its registers)
    auto split_rows = [&](void* dst, int m) -> int {
        return lo8 ? leaf_check(leaf_launch_split16_rows_lo8(b.x16, b.x, dst, m, d, dt, s), "split16_rows_lo8")
                   : leaf_check(leaf_launch_split16_rows(b.x, dst, m, d, dt, s), "split16_rows");
    };
    auto qkv_gemm = [&](int l, int m, const GemmLn& g, const void* xn) -> int {
        const LayerOff& o = h->layer[l];
        if (!fold) return leaf_gemm(dt, EPI_STORE_T, xn, d, W + h->w16_qkv(l), d, b.qkv, 3 * d, P + o.qkv_b, nullptr, m, 3 * d, d, 0, s);
        GemmLn q = g; q.ln_s = h->fold_s_qkv(W, l);
        if (sq(l)) {     // b.hh holds [hi | lo | hi] of these rows (built by the caller)
            q.ln_s = h->split_s_qkv(l);
            return leaf_gemm(dt, EPI_LNFOLD_T, b.hh, 3 * d, h->split_qkv3(l), 3 * d, b.qkv, 3 * d, h->fold_c_qkv(W, l), nullptr, m, 3 * d,
                             3 * d, 0, s, 0.f, 0, nullptr, &q);
        }
        return leaf_gemm(dt, EPI_LNFOLD_T, g.x16, d, W + h->w16_fold_qkv(l), d, b.qkv, 3 * d, h->fold_c_qkv(W, l), nullptr, m, 3 * d, d,
                         0, s, 0.f, 0, nullptr, &q);
    };
    auto fc_gemm = [&](int l, int m, const GemmLn& g, const void* xn, void* hid) -> int {
        const LayerOff& o = h->layer[l];
        if (!fold) return leaf_gemm(dt, EPI_ACT_T, xn, d, W + h->w16_fc(l), d, hid, 4 * d, P + o.fc_b, nullptr, m, 4 * d, d, c.activation, s);
        GemmLn q = g; q.ln_s = h->fold_s_fc(W, l);
        if (sf(l)) {     // scratch: the chunk's own q|k|v buffer (dead behind the attention)
            if (split_rows(b_in.qkv, m)) return 1;
            q.ln_s = h->split_s_fc(l);
            return leaf_gemm(dt, EPI_LNFOLD_ACT_T, b_in.qkv, 3 * d, h->split_fc3(l), 3 * d, hid, 4 * d, h->fold_c_fc(W, l), nullptr, m, 4 * d,
                             3 * d, c.activation, s, 0.f, 0, nullptr, &q);
        }
        return leaf_gemm(dt, EPI_LNFOLD_ACT_T, g.x16, d, W + h->w16_fold_fc(l), d, hid, 4 * d, h->fold_c_fc(W, l), nullptr, m, 4 * d, d,
                         c.activation, s, 0.f, 0, nullptr, &q);
    };
    // residual GEMM x += A W^T + bias; with folding (and a LayerNorm following) it also emits x16 / statistics
    // ... and the tiny finalize launch turns the [group][row] partials into (mean, rstd) per row for the consuming GEMM.
    // w_lo (split blocks): the A operand is a stored 16-bit tensor (exact as it is), so only the weights are split: x += A W_hi^T +
    // bias, then x += A W_lo^T in a second launch, which is the one whose x16 / statistics of the finished rows are read
    auto resid_gemm = [&](const void* A, int K, size_t w_off, const float* bias, float* x, int m, const GemmLn* g,
                          const uint16_t* w_lo = nullptr) -> int {
        const uint16_t* Wl = W + w_off;
        const bool l8 = lo8 && x == b.x;
        if (w_lo) {
            if (l8) { if (leaf_gemm(dt, EPI_RESID_LN8, A, K, Wl, K, x, d, bias, nullptr, m, d, K, 0, s, 0.f, 0, nullptr, &ln)) return 1; }
            else if (leaf_gemm(dt, EPI_RESID_F32, A, K, Wl, K, x, d, bias, nullptr, m, d, K, 0, s)) return 1;
            Wl = w_lo; bias = nullptr;
        }
        if (l8) {
            // (the last block's c_proj has no LayerNorm behind it: its statistics go to the chunk's buffers all the same, unread)
            if (leaf_gemm(dt, EPI_RESID_LN8, A, K, Wl, K, x, d, bias, nullptr, m, d, K, 0, s, 0.f, 0, nullptr, &ln)) return 1;
            return g ? leaf_check(leaf_launch_ln_finalize(ln.stat_out, ln.stat_ld, m, d / 64, ln.eps, const_cast<float2*>(ln.rowstat), s), "ln_finalize") : 0;
        }
        if (!(fold && g)) return leaf_gemm(dt, EPI_RESID_F32, A, K, Wl, K, x, d, bias, nullptr, m, d, K, 0, s);
        if (leaf_gemm(dt, EPI_RESID_LN, A, K, Wl, K, x, d, bias, nullptr, m, d, K, 0, s, 0.f, 0, nullptr, g)) return 1;
        return leaf_check(leaf_launch_ln_finalize(g->stat_out, g->stat_ld, m, d / 64, g->eps, const_cast<float2*>(g->rowstat), s), "ln_finalize");
    };
    // out-projection of a split block: x += [A | A] [W_hi | W_lo]^T + bias in ONE launch over K = 2d (the residual stream is
    // read-modify-written once).  The half-stage ring kernel re-reads A itself (GemmArgs::a_wrap); launches too small for it multiply
    // a duplicated copy of A (scratch: the chunk's own q|k|v buffer, dead behind the attention) -- the same operands in the same k
    // order, so a row's bits do not depend on which of the two ran.
    auto out_gemm = [&](int l, const void* A, float* x, int m, const GemmLn* g) -> int {
        const LayerOff& o = h->layer[l];
        if (!(splits && h->split_out(l))) return resid_gemm(A, d, h->w16_out(l), P + o.out_b, x, m, g);
        const bool l8 = lo8 && x == b.x;
        const int epi = l8 ? EPI_RESID_LN8 : (fold && g) ? EPI_RESID_LN : EPI_RESID_F32;
        const GemmLn* gl = l8 ? &ln : g;
        GemmArgs probe{};
        probe.M = m; probe.N = d; probe.K = 2 * d; probe.lda = d; probe.ldb = 2 * d; probe.ldc = d; probe.ldx16 = gl ? gl->ldx16 : 0; probe.a_wrap = d / 64;
        const bool ring = d % 64 == 0 && leaf_gemm_family(probe, epi) == 4;
        const void* A2 = A;
        if (!ring) {
            LEAF_TRY(leaf_launch_dup_cols16(A, b_in.qkv, m, d, s));
            A2 = b_in.qkv;
        }
        if (leaf_gemm(dt, epi, A2, ring ? d : 2 * d, h->split_out2(l), 2 * d, x, d, P + o.out_b, nullptr, m, d, 2 * d, 0, s, 0.f, 0, nullptr,
                      epi == EPI_RESID_F32 ? nullptr : gl, ring ? d / 64 : 0)) return 1;
        if (epi == EPI_RESID_F32 || (l8 && !g)) return 0;
        return leaf_check(leaf_launch_ln_finalize(gl->stat_out, gl->stat_ld, m, d / 64, gl->eps, const_cast<float2*>(gl->rowstat), s), "ln_finalize");
    };
    if (fold) {
        LEAF_TRY(leaf_launch_embed_fold(tokens, P + h->tok_emb, P + h->pos_emb, b.x, b.x16, b.stat, rows, rows, cs, map, d,
                                        c.vocab_size, dt, s, nullptr, lo8, sq(0) ? b.hh : nullptr));
        LEAF_TRY(leaf_launch_ln_finalize(b.stat, rows, rows, d / 64, c.ln_eps, b.rowstat, s));
    }
    else
        LEAF_TRY(leaf_launch_embed_ln(tokens, P + h->tok_emb, P + h->pos_emb, P + o0.ln1_w, P + o0.ln1_b, c.ln_eps, b.x,
                                      b.a, rows, cs, map, d, c.vocab_size, dt, s));
    for (int l = 0; l < c.layers; ++l) {
        const LayerOff& o = h->layer[l];
        const bool last = l == c.layers - 1;
        if (l > 0 && !fold) LEAF_TRY(leaf_launch_layernorm(b.x, P + o.ln1_w, P + o.ln1_b, c.ln_eps, b.a, rows, d, dt, s));
        if (kv.kv_write) b.qkv = kv.kv_write + (size_t)l * kv.kv_stride;
        const void* kvl = kv.kv_read ? kv.kv_read + (size_t)l * kv.kv_stride : nullptr;
        // QKV GEMM -> attention in one launch (qkv_attn.hip): q|k|v stay on chip, b.a receives the attention output.  In the fused
        // first stage the captions' q|k|v rows -- which their candidates' attention reads as cached prefix and the second stage
        // reads again -- come from a small GEMM of their own straight into the cache (the same bits as from any other kernel).
        // a split QKV block multiplies [hi | lo | hi] rows (b.hh) with the [hi | hi | lo] weights over K = 3d: through the same fused
        // launch when it takes that K, else as QKV GEMM + attention kernel
        const bool q3 = sq(l);
        const bool fused_attn = kv.attn_tiles > 0 && (!q3 || leaf_qkv_attn_eligible(d, c.heads, c.context_length, 3 * d, kv.attn_max_len));
        if (q3 && l > 0 && split_rows(b.hh, rows)) return 1;      // (block 0: written by the embedding kernel)
        const void* qA = q3 ? (const void*)b.hh : ln.x16;
        const uint16_t* qB = q3 ? h->split_qkv3(l) : W + h->w16_fold_qkv(l);
        const float* qS = q3 ? h->split_s_qkv(l) : h->fold_s_qkv(W, l);
        const int qK = q3 ? 3 * d : d;
        if (fused_attn) {
            if (kv.kv_self_rows) {
                // only the K and V thirds: nothing ever reads a caption's q rows from the cache (its own attention runs inside the
                // fused launch below, from LDS) -- weight rows [d, 3d) into cache columns [d, 3d) of the [rows, 3d] layout
                uint16_t* dst = kv.kv_copy + (size_t)l * kv.kv_stride;
                GemmLn q = ln; q.ln_s = qS + d;
                if (leaf_gemm(dt, EPI_LNFOLD_T, qA, qK, qB + (size_t)d * qK, qK, dst + d, 3 * d, h->fold_c_qkv(W, l) + d,
                              nullptr, (int)kv.kv_self_rows, 2 * d, qK, 0, s, 0.f, 0, nullptr, &q)) return 1;
                kvl = dst;
            }
            const bool trim = last && h->last_trim && out;
            if (trim) LEAF_TRY(leaf_launch_eot_positions(tokens, b.eot, cs, map, s));
            if (last && !out) break;
            QkvAttnArgs qa;
            qa.A = qA; qa.B = qB; qa.bias = h->fold_c_qkv(W, l); qa.ln_s = qS;
            qa.rowstat = ln.rowstat; qa.out = b.a; qa.kv_base = kvl; qa.eot_pos = trim ? b.eot : nullptr; qa.tile_seq = b.tile_seq;
            qa.map = map; qa.M = rows; qa.K = qK; qa.lda = qK; qa.ldb = qK; qa.heads = c.heads; qa.d = d; qa.n_tiles = kv.attn_tiles;
            qa.n_seq = cs; qa.kv_ld = 3 * d; qa.stamps = nullptr;
            leaf_qkv_attn_lds_plan(kv.attn_max_len, &qa.ncap, &qa.caprows);
            if (leaf_qkv_attn(qa, dt, s)) return 1;
        } else
        if (qkv_gemm(l, rows, ln, b.a)) return 1;
        if (!fused_attn && kv.kv_self_rows) {
            kvl = b.qkv;
#ifdef LEAF_COPY_MEMCPY   // A/B build: the runtime's copy (three dispatches per 15-MB copy)
            LEAF_TRY(hipMemcpyAsync(kv.kv_copy + (size_t)l * kv.kv_stride, b.qkv, (size_t)kv.kv_self_rows * 3 * d * 2, hipMemcpyDeviceToDevice, s));
#else
            LEAF_TRY(leaf_launch_copy_bytes(b.qkv, kv.kv_copy + (size_t)l * kv.kv_stride, (size_t)kv.kv_self_rows * 3 * d * 2, s));
#endif
        }
        if (last && !out) break;          // K/V-only pass (clean captions for the cache): nothing consumes the rest
        if (last && h->last_trim && !kv.kv_write) {
            // Last block: only the pooled row (first maximum token id = EOT) of each sequence reaches the output, and every later op is
            // row-wise, so attention output, out-projection, LN2 and the MLP run on ONE row per sequence (bit-identical).
            // Scratch: after attention and the gather, the qkv and fc buffers (carved back to back, 14*d*rows bytes) are
            // dead; the gathered residual rows xg (fp32 [cs,d]) and the MLP hidden rows hb (16-bit [cs,4d]) need
            // 12*d*cs <= 14*d*rows bytes.  With folding the chunk's x16 / statistics buffers (dead after the QKV GEMM above)
            // take the pooled rows' 16-bit copy and statistics (row stride cs).
            if (!fused_attn) {
                LEAF_TRY(leaf_launch_eot_positions(tokens, b.eot, cs, map, s));
                LEAF_TRY(leaf_launch_attention_fwd(b.qkv, kvl, b.a, cs, map, c.heads, d, dt, s, b.eot, max_len));
            }
            if (lo8) {
                // 16 + 8-bit stream: the pooled rows stay in that format through the block (the same EPI_RESID_LN8 arithmetic as the
                // untrimmed pass: bit-identical), scratch = [x16 rows | remainder rows | hidden rows], 11 d cs bytes; the fp32 rows the
                // final LayerNorm + projection read are decoded into the hidden rows' space once c_proj has consumed them
                uint16_t* g16 = (uint16_t*)b.qkv;
                unsigned char* g8 = (unsigned char*)b.qkv + align_up((size_t)cs * d * 2, 256);
                uint16_t* hb = (uint16_t*)(g8 + align_up((size_t)cs * d, 256));
                LEAF_TRY(leaf_launch_gather_rows_pair(b.x16, b.x, b.eot, g16, g8, cs, map, d, s));
                GemmLn lg = ln;
                lg.stat_ld = cs; lg.x16 = g16;
                if (leaf_gemm(dt, EPI_RESID_LN8, b.a, d, W + h->w16_out(l), d, g8, d, P + o.out_b, nullptr, cs, d, d, 0, s, 0.f, 0, nullptr, &lg)) return 1;
                LEAF_TRY(leaf_launch_ln_finalize(lg.stat_out, lg.stat_ld, cs, d / 64, lg.eps, const_cast<float2*>(lg.rowstat), s));
                if (fc_gemm(l, cs, lg, b.a, hb)) return 1;
                if (leaf_gemm(dt, EPI_RESID_LN8, hb, 4 * d, W + h->w16_proj(l), 4 * d, g8, d, P + o.proj_b, nullptr, cs, d, 4 * d, 0, s, 0.f, 0, nullptr, &lg)) return 1;
                float* xg = (float*)hb;
                LEAF_TRY(leaf_launch_resid_unpack(g16, g8, xg, (size_t)cs * d, dt, s));
                LEAF_TRY(leaf_launch_project_rows(xg, P + h->lnf_w, P + h->lnf_b, c.ln_eps, P + h->text_proj, xg + (size_t)cs * d, out,
                                                  cs, d, c.embed_dim, normalize, s));
                return 0;
            }
            float* xg = (float*)b.qkv;
            uint16_t* hb = (uint16_t*)((char*)b.qkv + align_up((size_t)cs * d * 4, 256));
            LEAF_TRY(leaf_launch_gather_rows(b.x, b.eot, xg, cs, map, d, s));
            GemmLn lg = ln;
            lg.stat_ld = cs;
            if (resid_gemm(b.a, d, h->w16_out(l), P + o.out_b, xg, cs, &lg)) return 1;
            if (!fold) LEAF_TRY(leaf_launch_layernorm(xg, P + o.ln2_w, P + o.ln2_b, c.ln_eps, b.a, cs, d, dt, s));
            if (fc_gemm(l, cs, lg, b.a, hb)) return 1;
            if (resid_gemm(hb, 4 * d, h->w16_proj(l), P + o.proj_b, xg, cs, nullptr)) return 1;
            if (leaf_project_rows_ok(d, c.embed_dim)) {   // hb is dead after the c_proj GEMM: its space takes LN(xg)
                LEAF_TRY(leaf_launch_project_rows(xg, P + h->lnf_w, P + h->lnf_b, c.ln_eps, P + h->text_proj, (float*)hb, out,
                                                  cs, d, c.embed_dim, normalize, s));
                return 0;
            }
            LEAF_TRY(leaf_launch_pool_project(xg, tokens, P + h->lnf_w, P + h->lnf_b, c.ln_eps, P + h->text_proj, out, nullptr,
                                              nullptr, cs, map, d, c.embed_dim, normalize, s, /*rows_are_pooled=*/1));
            return 0;
        }
        if (!fused_attn) LEAF_TRY(leaf_launch_attention_fwd(b.qkv, kvl, b.a, cs, map, c.heads, d, dt, s, nullptr, max_len));
        if (out_gemm(l, b.a, b.x, rows, &ln)) return 1;
        if (!fold) LEAF_TRY(leaf_launch_layernorm(b.x, P + o.ln2_w, P + o.ln2_b, c.ln_eps, b.a, rows, d, dt, s));
        if (fc_gemm(l, rows, ln, b.a, b.hh)) return 1;
        if (resid_gemm(b.hh, 4 * d, h->w16_proj(l), P + o.proj_b, b.x, rows, last ? nullptr : &ln,   // ln_final runs on the pooled rows
                       (splits && h->split_proj(l)) ? h->split_proj_lo(l) : nullptr)) return 1;
    }
    if (out && leaf_project_rows_ok(d, c.embed_dim)) {
        // same op sequence as the trimmed path (bit-identical features): gather the pooled rows, LN, fp32 projection.
        // qkv / fc scratch is dead here.
        // scratch in the chunk's OWN qkv / fc buffers (b_in): with kv.kv_write, b.qkv points into the caller's K/V cache
        float* xg = (float*)b_in.qkv;
        float* xn = (float*)((char*)b_in.qkv + align_up((size_t)cs * d * 4, 256));
        LEAF_TRY(leaf_launch_eot_positions(tokens, b.eot, cs, map, s));
        if (lo8) LEAF_TRY(leaf_launch_gather_rows_lo8(b.x16, b.x, b.eot, xg, cs, map, d, dt, s));
        else LEAF_TRY(leaf_launch_gather_rows(b.x, b.eot, xg, cs, map, d, s));
        LEAF_TRY(leaf_launch_project_rows(xg, P + h->lnf_w, P + h->lnf_b, c.ln_eps, P + h->text_proj, xn, out, cs, d,
                                          c.embed_dim, normalize, s));
        return 0;
    }
    if (out)
        LEAF_TRY(leaf_launch_pool_project(b.x, tokens, P + h->lnf_w, P + h->lnf_b, c.ln_eps, P + h->text_proj, out, nullptr,
                                          nullptr, cs, map, d, c.embed_dim, normalize, s));
    return 0;
}

// Sequences are processed in chunks bounded by a ROW budget (chunk * ctx rows): with EOT-trimmed (packed) rows a
// chunk holds more sequences, so the GEMMs keep their M.  lens == nullptr -> dense (every sequence ctx rows).
struct PrefixPlan {   // prefix reuse: see RowMap in common.h
    const int32_t* prefix_dev = nullptr;
    const int32_t* base_cu_dev = nullptr;
    const uint16_t* kv = nullptr;
    size_t kv_stride = 0;
    int group = 1;
    int max_len = 0;   // upper bound of prefix + suffix length over the candidates (0 = context_length)
    int group_off = 0;          // fused pass: the first group_off sequences are the clean captions
    uint16_t* kv_copy = nullptr;
    size_t kv_self_rows = 0;
};

// rows below which a pass is not split across the two streams (the halves would not fill the chip anyway)
constexpr size_t SPLIT_MIN_ROWS = 16384;

int ensure_side_stream(leaf_text* h) {
    if (h->side) return 0;
    LEAF_TRY(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    LEAF_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    LEAF_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    return 0;
}

int forward_all(leaf_text* h, const float* P, const void* W, const int32_t* tokens, const int32_t* lens,
                const int32_t* cu_dev, int n_seq, float* out, int normalize, Carver& c, hipStream_t s,
                const PrefixPlan& pp = PrefixPlan()) {
    const int ctx = h->cfg.context_length;
    if ((lens == nullptr) != (cu_dev == nullptr)) { leaf_set_error("seq_lens (host) and cu_rows (device) go together"); return 1; }
    const size_t budget = (size_t)(n_seq < h->chunk ? n_seq : h->chunk) * ctx;
    size_t total = 0;
    int max_len = 0;
    for (int i = 0; i < n_seq; ++i) {
        const int L = lens ? lens[i] : ctx;
        if (L < 1 || L > ctx) { leaf_set_error("seq_lens[%d] = %d out of range 1..%d", i, L, ctx); return 1; }
        total += L;
        max_len = L > max_len ? L : max_len;
    }
    if (pp.prefix_dev) {   // computed rows are suffixes: the bound comes from the caller
        if (pp.max_len > 0 && pp.max_len < max_len) { leaf_set_error("max_len %d below a suffix length %d", pp.max_len, max_len); return 1; }
        max_len = pp.max_len > 0 && pp.max_len <= ctx ? pp.max_len : ctx;
    }
    const int nsets = (h->streams >= 2 && total >= SPLIT_MIN_ROWS && n_seq >= 2 && !pp.kv_self_rows) ? 2 : 1;
    size_t nchunks = (total + budget - 1) / budget;
    if (pp.kv_self_rows && nchunks > 1) { leaf_set_error("fused pass: the captions and their candidates must fit one chunk (%zu rows > %zu)", total, budget); return 2; }
    if (nsets == 2) nchunks = (nchunks + 1) / 2 * 2;                 // an even number of near-equal chunks
    const size_t target = (total + nchunks - 1) / nchunks;           // <= budget
    FwdBuf bufs[2];
    bufs[0] = carve_fwd(h, c, (int)(budget / ctx));
    if (nsets == 2) bufs[1] = carve_fwd(h, c, (int)(budget / ctx));
    if (!c.ok()) { leaf_set_error("workspace too small: need %zu bytes, have %zu", c.off, c.cap); return 1; }
    if (nsets == 2) {
        if (ensure_side_stream(h)) return 1;
        LEAF_TRY(hipEventRecord(h->ev_fork, s));
        LEAF_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    }
    int s0 = 0, ci = 0;
    size_t row0 = 0;
    while (s0 < n_seq) {
        int s1 = s0;
        size_t rows = 0;
        while (s1 < n_seq) {
            const int L = lens ? lens[s1] : ctx;
            if (rows + L > target && s1 > s0) break;
            if (rows + L > budget) break;
            rows += L;
            ++s1;
        }
        RowMap map{cu_dev, s0, (int)row0, ctx, pp.prefix_dev, pp.base_cu_dev, pp.group, pp.group_off};
        KvPlan kv;
        kv.kv_read = pp.kv;
        kv.kv_stride = pp.kv_stride;
        kv.kv_copy = pp.kv_copy;
        kv.kv_self_rows = pp.kv_self_rows;
        if (s1 == s0) { leaf_set_error("a sequence does not fit the row budget"); return 1; }
        const int set = nsets == 2 ? (ci & 1) : 0;
        // fused QKV + attention for the big passes: cut the chunk's sequences into M tiles of whole sequences on the host and send
        // the cut points behind the work already queued (a pinned ring slot; its event keeps a slot from being rewritten early)
        if (h->fuse_attn && h->ln_fold && leaf_qkv_attn_eligible(h->cfg.width, h->cfg.heads, ctx, h->cfg.width, max_len) &&
            rows / 256 * (size_t)h->cfg.heads >= 256) {      // at least a chip's worth of (256-row tile, head) items
            const size_t need = 2 * ((size_t)(s1 - s0) + 2);
            if (h->plan_cap < need) {
                for (int i = 0; i < leaf_text::PLAN_RING; ++i) {
                    if (h->plan_ev[i]) LEAF_TRY(hipEventSynchronize(h->plan_ev[i]));
                    if (h->plan_host[i]) { LEAF_TRY(hipHostFree(h->plan_host[i])); h->plan_host[i] = nullptr; }
                }
                h->plan_cap = need * 2;
            }
            const int k = h->plan_next;
            h->plan_next = (k + 1) % leaf_text::PLAN_RING;
            if (!h->plan_host[k]) LEAF_TRY(hipHostMalloc((void**)&h->plan_host[k], h->plan_cap * sizeof(int32_t), hipHostMallocDefault));
            if (!h->plan_ev[k]) LEAF_TRY(hipEventCreateWithFlags(&h->plan_ev[k], hipEventDisableTiming));
            else LEAF_TRY(hipEventSynchronize(h->plan_ev[k]));
            kv.attn_tiles = leaf_qkv_attn_plan(lens ? lens + s0 : nullptr, ctx, s0, s1 - s0, pp.prefix_dev != nullptr, pp.group, pp.group_off,
                                               leaf_qkv_attn_tile_rows(), leaf_qkv_attn_ncap(max_len), h->plan_host[k]);
            kv.attn_max_len = max_len;
            hipStream_t cs_ = set ? h->side : s;
            LEAF_TRY(hipMemcpyAsync(bufs[set].tile_seq, h->plan_host[k], (size_t)(kv.attn_tiles + 1) * 2 * sizeof(int32_t), hipMemcpyHostToDevice, cs_));
            LEAF_TRY(hipEventRecord(h->plan_ev[k], cs_));
        }
        if (forward_chunk(h, P, (const uint16_t*)W, tokens, s1 - s0, (int)rows, map, out + (size_t)s0 * h->cfg.embed_dim,
                          normalize, bufs[set], set ? h->side : s, kv, max_len))
            return 1;
        s0 = s1;
        row0 += rows;
        ++ci;
    }
    if (nsets == 2) {
        LEAF_TRY(hipEventRecord(h->ev_join, h->side));
        LEAF_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
    }
    return 0;
}

}  // namespace

extern "C" size_t leaf_text_workspace_bytes(leaf_text_t h, int n_seq, int mode) {
    if (!h || n_seq < 1) return 0;
    const int cs = n_seq < h->chunk ? n_seq : h->chunk;
    size_t b = (h->streams >= 2 ? 2 : 1) * (fwd_chunk_bytes(h, cs) + 256) + 256;   // one buffer set per stream
    if (mode == 1) b += align_up((size_t)n_seq * h->cfg.embed_dim * 4, 256) + 256;
    if (mode == 2) return leaf_train_ws_bytes(h, n_seq);
    return b;
}

extern "C" int leaf_text_forward(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                                 const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, float* out, int normalize,
                                 void* ws, size_t ws_bytes, leaf_stream_t s) {
    if (!h || !params || !w16_fwd || !tokens || !out || !ws || n_seq < 1) { leaf_set_error("null/invalid argument"); return 1; }
    Carver c(ws, ws_bytes);
    return forward_all(h, params, w16_fwd, tokens, seq_lens, cu_rows, n_seq, out, normalize, c, (hipStream_t)s);
}

extern "C" int leaf_score_candidates(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                                     const int32_t* seq_lens, const int32_t* cu_rows, const float* anchor, int B,
                                     int rho, int objective, int32_t* best_idx,
                                     float* best_feat, float* loss, void* ws, size_t ws_bytes, leaf_stream_t s) {
    if (!h || !params || !w16_fwd || !tokens || !anchor || !best_idx || !ws || B < 1 || rho < 1) {
        leaf_set_error("null/invalid argument");
        return 1;
    }
    if (objective < 0 || objective > 3) { leaf_set_error("unknown objective %d", objective); return 1; }
    Carver c(ws, ws_bytes);
    const int n_seq = B * rho;
    float* feat = (float*)c.take((size_t)n_seq * h->cfg.embed_dim * 4);
    const int normalize = (objective == LEAF_OBJ_SIM || objective == LEAF_OBJ_DISSIM);
    if (forward_all(h, params, w16_fwd, tokens, seq_lens, cu_rows, n_seq, feat, normalize, c, (hipStream_t)s)) return 1;
    LEAF_TRY(leaf_launch_score(feat, anchor, B, rho, h->cfg.embed_dim, objective, best_idx, best_feat, loss,
                               (hipStream_t)s));
    return 0;
}

// ------------------------------------------------------------------ prefix reuse
extern "C" size_t leaf_text_kv_bytes(leaf_text_t h, int n_seq) {
    if (!h || n_seq < 1) return 0;
    return (size_t)h->cfg.layers * n_seq * h->cfg.context_length * 3 * h->cfg.width * 2;
}

extern "C" int leaf_text_forward_kv(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                                    const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, float* out,
                                    int normalize, void* kv, size_t kv_bytes, void* ws, size_t ws_bytes, leaf_stream_t s) {
    if (!h || !params || !w16_fwd || !tokens || !kv || !ws || n_seq < 1) { leaf_set_error("null/invalid argument"); return 1; }
    if ((seq_lens == nullptr) != (cu_rows == nullptr)) { leaf_set_error("seq_lens (host) and cu_rows (device) go together"); return 1; }
    const int ctx = h->cfg.context_length;
    size_t rows = 0;
    int max_len = 0;
    for (int i = 0; i < n_seq; ++i) {
        const int L = seq_lens ? seq_lens[i] : ctx;
        rows += L;
        max_len = L > max_len ? L : max_len;
    }
    const size_t stride = rows * 3 * h->cfg.width;
    if (stride * 2 * h->cfg.layers > kv_bytes) { leaf_set_error("kv cache too small"); return 1; }
    if (rows > (size_t)h->chunk * ctx) { leaf_set_error("forward_kv needs the captions to fit one chunk (%zu rows)", rows); return 1; }
    Carver c(ws, ws_bytes);
    FwdBuf b = carve_fwd(h, c, (int)((rows + ctx - 1) / ctx));
    if (!c.ok()) { leaf_set_error("workspace too small: need %zu bytes, have %zu", c.off, c.cap); return 1; }
    KvPlan kvp;
    kvp.kv_write = (uint16_t*)kv;
    kvp.kv_stride = stride;
    RowMap map{cu_rows, 0, 0, ctx, nullptr, nullptr, 1};
    return forward_chunk(h, params, (const uint16_t*)w16_fwd, tokens, n_seq, (int)rows, map, out, normalize, b,
                         (hipStream_t)s, kvp, max_len);
}

extern "C" int leaf_score_candidates_prefix(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                                            const int32_t* suffix_lens, const int32_t* cu_suffix, const int32_t* prefix,
                                            const int32_t* base_cu, const void* kv, size_t base_rows, int max_len,
                                            const float* anchor, int B, int rho, int objective, int32_t* best_idx, float* best_feat, float* loss,
                                            void* ws, size_t ws_bytes, leaf_stream_t s) {
    if (!h || !params || !w16_fwd || !tokens || !suffix_lens || !cu_suffix || !prefix || !base_cu || !kv || !anchor ||
        !best_idx || !ws || B < 1 || rho < 1) {
        leaf_set_error("null/invalid argument");
        return 1;
    }
    if (objective < 0 || objective > 3) { leaf_set_error("unknown objective %d", objective); return 1; }
    Carver c(ws, ws_bytes);
    const int n_seq = B * rho;
    float* feat = (float*)c.take((size_t)n_seq * h->cfg.embed_dim * 4);
    const int normalize = (objective == LEAF_OBJ_SIM || objective == LEAF_OBJ_DISSIM);
    PrefixPlan pp;
    pp.prefix_dev = prefix;
    pp.base_cu_dev = base_cu;
    pp.max_len = max_len;
    pp.kv = (const uint16_t*)kv;
    pp.kv_stride = base_rows * 3 * h->cfg.width;
    pp.group = rho;
    if (forward_all(h, params, w16_fwd, tokens, suffix_lens, cu_suffix, n_seq, feat, normalize, c, (hipStream_t)s, pp)) return 1;
    LEAF_TRY(leaf_launch_score(feat, anchor, B, rho, h->cfg.embed_dim, objective, best_idx, best_feat, loss, (hipStream_t)s));
    return 0;
}

// The clean-caption K/V pass and the FIRST stage's scoring in one pass: sequences 0 .. B-1 are the captions (full rows, prefix 0),
// B .. B + B rho - 1 their candidates; seq_lens (host) / cu_rows / prefix (device) cover all of them.  Returns 2 (and an error
// text) when the rows do not fit one chunk: the caller then runs leaf_text_forward_kv + leaf_score_candidates_prefix.
extern "C" int leaf_score_candidates_prefix_fused(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                                                  const int32_t* seq_lens, const int32_t* cu_rows, const int32_t* prefix,
                                                  int max_len, const float* anchor, int B, int rho, int objective,
                                                  int32_t* best_idx, float* best_feat, float* loss, void* kv, size_t kv_bytes,
                                                  void* ws, size_t ws_bytes, leaf_stream_t s) {
    if (!h || !params || !w16_fwd || !tokens || !seq_lens || !cu_rows || !prefix || !anchor || !best_idx || !kv || !ws || B < 1 || rho < 1) {
        leaf_set_error("null/invalid argument");
        return 1;
    }
    if (objective < 0 || objective > 3) { leaf_set_error("unknown objective %d", objective); return 1; }
    size_t base_rows = 0;
    for (int i = 0; i < B; ++i) base_rows += seq_lens[i];
    const size_t stride = base_rows * 3 * h->cfg.width;
    if (stride * 2 * h->cfg.layers > kv_bytes) { leaf_set_error("kv cache too small"); return 1; }
    Carver c(ws, ws_bytes);
    const int n_seq = B + B * rho;
    float* feat = (float*)c.take((size_t)n_seq * h->cfg.embed_dim * 4);
    const int normalize = (objective == LEAF_OBJ_SIM || objective == LEAF_OBJ_DISSIM);
    PrefixPlan pp;
    pp.prefix_dev = prefix;
    pp.base_cu_dev = cu_rows;          // the captions are the first sequences: their row offsets are the first B entries
    pp.max_len = max_len;
    pp.kv = nullptr;
    pp.kv_stride = stride;
    pp.group = rho;
    pp.group_off = B;
    pp.kv_copy = (uint16_t*)kv;
    pp.kv_self_rows = base_rows;
    const int rc = forward_all(h, params, w16_fwd, tokens, seq_lens, cu_rows, n_seq, feat, normalize, c, (hipStream_t)s, pp);
    if (rc) return rc;
    LEAF_TRY(leaf_launch_score(feat + (size_t)B * h->cfg.embed_dim, anchor, B, rho, h->cfg.embed_dim, objective, best_idx, best_feat, loss,
                               (hipStream_t)s));
    return 0;
}

// ------------------------------------------------------------------ single-kernel hooks for the parity tests
extern "C" int leaf_op_gemm(int dtype, int epi, const void* A, const void* B, void* C, const float* bias, void* aux,
                            int M, int N, int K, int act, float beta, int aux_f16, leaf_stream_t s) {
    return leaf_gemm(dtype, epi, A, K, B, K, C, N, bias, aux, M, N, K, act, (hipStream_t)s, beta, aux_f16);
}
extern "C" int leaf_op_gemm_ld(int dtype, int epi, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                               const float* bias, void* aux, int M, int N, int K, int act, float beta, int aux_f16,
                               leaf_stream_t s) {
    return leaf_gemm(dtype, epi, A, lda, B, ldb, C, ldc, bias, aux, M, N, K, act, (hipStream_t)s, beta, aux_f16);
}
// C32 += [A | A] B^T + bias over K = 2 Ka with A stored once as [M, Ka] (GemmArgs::a_wrap: the half-stage ring kernel re-reads the A
// panel; fails for launches that kernel does not take) -- the out-projection of a split block; the parity tests hold it to the
// launch on a materialised [A | A]
extern "C" int leaf_op_gemm_awrap(int dtype, const void* A, const void* B, float* C, const float* bias, int M, int N, int Ka, leaf_stream_t s) {
    if (Ka % 64) { leaf_set_error("gemm_awrap: Ka %% 64"); return 1; }
    return leaf_gemm(dtype, EPI_RESID_F32, A, Ka, B, 2 * Ka, C, N, bias, nullptr, M, N, 2 * Ka, 0, (hipStream_t)s, 0.f, 0, nullptr, nullptr, Ka / 64);
}
// LN folding (lnfold.h): the producing residual GEMM (C32 += A B^T + bias, x16 = 16-bit(C32), stat[N/64][M] = (sum, M2)) ...
extern "C" int leaf_op_gemm_resid_ln(int dtype, const void* A, const void* B, float* C, const float* bias, void* x16, void* stat,
                                     int M, int N, int K, leaf_stream_t s) {
    GemmLn ln;
    ln.x16 = x16; ln.ldx16 = N; ln.stat_out = (float2*)stat; ln.stat_ld = M;
    return leaf_gemm(dtype, EPI_RESID_LN, A, K, B, K, C, N, bias, nullptr, M, N, K, 0, (hipStream_t)s, 0.f, 0, nullptr, &ln);
}
// the same on the 16 + 8-bit residual stream: (x16, lo8) <- encode(decode(x16, lo8) + A B^T + bias) in place, statistics as above
extern "C" int leaf_op_gemm_resid_ln8(int dtype, const void* A, const void* B, void* lo8, const float* bias, void* x16, void* stat,
                                      int M, int N, int K, leaf_stream_t s) {
    GemmLn ln;
    ln.x16 = x16; ln.ldx16 = N; ln.stat_out = (float2*)stat; ln.stat_ld = M;
    return leaf_gemm(dtype, EPI_RESID_LN8, A, K, B, K, lo8, N, bias, nullptr, M, N, K, 0, (hipStream_t)s, 0.f, 0, nullptr, &ln);
}
extern "C" int leaf_op_resid_pack(int dtype, const float* x, void* x16, void* lo8, size_t n, leaf_stream_t s) {
    return leaf_check(leaf_launch_resid_pack(x, x16, lo8, n, dtype, (hipStream_t)s), "resid_pack");
}
extern "C" int leaf_op_resid_unpack(int dtype, const void* x16, const void* lo8, float* x, size_t n, leaf_stream_t s) {
    return leaf_check(leaf_launch_resid_unpack(x16, lo8, x, n, dtype, (hipStream_t)s), "resid_unpack");
}
// ... the merge of the partials into rowstat[M] = (mean, rstd), and the consuming GEMM: C16 = [act](rstd[m] (A Bp^T - mean[m] s[n]) + c[n]); act < 0: none
extern "C" int leaf_op_ln_finalize(const void* stat, int ld, int rows, int ngroups, float eps, void* rowstat, leaf_stream_t s) {
    return leaf_check(leaf_launch_ln_finalize((const float2*)stat, ld, rows, ngroups, eps, (float2*)rowstat, (hipStream_t)s), "ln_finalize");
}
extern "C" int leaf_op_gemm_lnfold(int dtype, int act, const void* A, const void* Bp, void* C16, const float* c_vec,
                                   const float* s_vec, const void* rowstat, int M, int N, int K, leaf_stream_t s) {
    GemmLn ln;
    ln.ln_s = s_vec; ln.rowstat = (const float2*)rowstat;
    return leaf_gemm(dtype, act < 0 ? EPI_LNFOLD_T : EPI_LNFOLD_ACT_T, A, K, Bp, K, C16, N, c_vec, nullptr, M, N, K, act < 0 ? 0 : act,
                     (hipStream_t)s, 0.f, 0, nullptr, &ln);
}
// The fused QKV + attention launch alone (tools/qkv_attn_bench.py): lens (host) = rows per sequence, cu / prefix / base_cu device
// arrays as in leaf_score_candidates_prefix (prefix / base_cu / kv may be null: no cached prefix), tile_seq = device scratch
// [n_seq + 1].  Synchronous plan upload: a measuring / test hook, not a step of the pipeline.
extern "C" int leaf_op_qkv_attn(int dtype, const void* x16, const void* Wp, const float* c_vec, const float* s_vec, const void* rowstat,
                                void* out, const void* kv, const int32_t* lens, const int32_t* cu, const int32_t* prefix,
                                const int32_t* base_cu, const int32_t* eot_pos, int32_t* tile_seq, int n_seq, int rows, int group,
                                int ctx, int heads, int width, leaf_stream_t s) {
    if (!leaf_qkv_attn_eligible(width, heads, ctx, width, 0)) { leaf_set_error("qkv_attn: unsupported shape"); return 1; }
    std::vector<int32_t> plan(2 * ((size_t)n_seq + 2));
    QkvAttnArgs qa;
    leaf_qkv_attn_lds_plan(ctx, &qa.ncap, &qa.caprows);
    qa.n_tiles = leaf_qkv_attn_plan(lens, ctx, 0, n_seq, prefix != nullptr, group, 0, leaf_qkv_attn_tile_rows(), qa.ncap, plan.data());
    LEAF_TRY(hipMemcpy(tile_seq, plan.data(), (size_t)(qa.n_tiles + 1) * 2 * sizeof(int32_t), hipMemcpyHostToDevice));
    qa.A = x16; qa.B = Wp; qa.bias = c_vec; qa.ln_s = s_vec; qa.rowstat = (const float2*)rowstat; qa.out = out; qa.kv_base = kv;
    qa.eot_pos = eot_pos; qa.tile_seq = tile_seq;
    qa.map = RowMap{cu, 0, 0, ctx, prefix, base_cu, group > 0 ? group : 1, 0};
    qa.M = rows; qa.K = width; qa.lda = width; qa.ldb = width; qa.heads = heads; qa.d = width; qa.n_seq = n_seq; qa.kv_ld = 3 * width;
    qa.stamps = nullptr;
    return leaf_check(leaf_launch_qkv_attn(qa, dtype, (hipStream_t)s), "qkv_attn");
}
// ---- diagnostic exports (include/leaf_hip_diag.h): only in the builds of `make variants` / `make stamps` ... under tools/diag/
#ifdef LEAF_VARIANTS
extern "C" int leaf_debug_qkv_attn_plan(const int32_t* lens, int ctx, int s0, int n, int prefixed, int group, int group_off, int tile_rows,
                                        int ncap, int32_t* out) {
    return leaf_qkv_attn_plan(lens, ctx, s0, n, prefixed, group, group_off, tile_rows > 0 ? tile_rows : leaf_qkv_attn_tile_rows(),
                              ncap > 0 ? ncap : leaf_qkv_attn_ncap(ctx), out);
}
extern "C" int leaf_debug_gemm_stamps(void* buf) { leaf_gemm_set_stamps(buf); return 0; }
extern "C" int leaf_debug_gemm_min_tiles(int n) { leaf_gemm256h_set_min_tiles(n); return 0; }
#endif
extern "C" int leaf_op_attention_fwd(const void* qkv, void* out, int n_seq, int ctx, int heads, int width, int dtype,
                                     leaf_stream_t s) {
    return leaf_check(leaf_launch_attention_fwd(qkv, nullptr, out, n_seq, RowMap{nullptr, 0, 0, ctx, nullptr, nullptr, 1}, heads, width, dtype,
                                                (hipStream_t)s), "attention_fwd");
}
extern "C" int leaf_op_layernorm(const float* x, const float* g, const float* b, float eps, void* out16, int rows,
                                 int width, int dtype, leaf_stream_t s) {
    return leaf_check(leaf_launch_layernorm(x, g, b, eps, out16, rows, width, dtype, (hipStream_t)s), "layernorm");
}
extern "C" int leaf_op_attention_bwd(const void* qkv, int qkv_dtype, const void* dout_bf16, void* dqkv_bf16, int n_seq,
                                     int ctx, int heads, int width, leaf_stream_t s) {
    return leaf_check(leaf_launch_attention_bwd(qkv, qkv_dtype, dout_bf16, dqkv_bf16, LEAF_BF16, n_seq,
                                                RowMap{nullptr, 0, 0, ctx, nullptr, nullptr, 1}, heads, width, (hipStream_t)s), "attention_bwd");
}
extern "C" int leaf_op_attention_bwd_t(const void* qkv, int qkv_dtype, const void* dout16, void* dqkv16, int g_dtype,
                                       int n_seq, int ctx, int heads, int width, leaf_stream_t s) {
    return leaf_check(leaf_launch_attention_bwd(qkv, qkv_dtype, dout16, dqkv16, g_dtype, n_seq,
                                                RowMap{nullptr, 0, 0, ctx, nullptr, nullptr, 1}, heads, width, (hipStream_t)s), "attention_bwd");
}
extern "C" int leaf_op_wgrad(const void* dY, const void* X, float* dW, float* db, int rows, int Nw, int Kw, int x_dtype,
                             int g_dtype, const float* alpha_dev, leaf_stream_t s) {
    if (!leaf_wgrad_tn_ok(Nw, Kw, Nw, Kw)) { leaf_set_error("wgrad needs Nw, Kw multiples of 128"); return 1; }
    WgradArgs wa{};
    wa.p[0] = WgradProb{(const uint16_t*)dY, (const uint16_t*)X, dW, db, Nw, Kw, Nw, Kw, 0, 0};
    wa.nprob = 1; wa.rows = rows; wa.alpha = alpha_dev;
    return leaf_check(leaf_launch_wgrad_group(wa, x_dtype, g_dtype, (hipStream_t)s), "wgrad_group");
}
extern "C" size_t leaf_op_layernorm_bwd_ws_bytes(int rows, int d) {
    return (size_t)leaf_ln_bwd_grid(rows, d) * 2 * d * sizeof(float);
}
extern "C" int leaf_op_layernorm_bwd(const float* dy, const float* x, const float* g, float eps, float* dx_inout, void* dx16,
                                     int g_dtype, const float* gscale, float* dg, float* db, int rows, int d, void* ws,
                                     size_t ws_bytes, leaf_stream_t s) {
    if (!dy || !x || !g || !dx_inout || !gscale || rows < 1 || (dg == nullptr) != (db == nullptr)) { leaf_set_error("layernorm_bwd: null/invalid argument"); return 1; }
    if (dg && (!ws || ws_bytes < leaf_op_layernorm_bwd_ws_bytes(rows, d))) { leaf_set_error("layernorm_bwd: workspace too small (need %zu bytes)", leaf_op_layernorm_bwd_ws_bytes(rows, d)); return 1; }
    LEAF_TRY(leaf_launch_layernorm_bwd(dy, x, g, eps, dx_inout, dx16, g_dtype == LEAF_F16 ? 1 : 0, dg ? (float*)ws : nullptr, rows, d, (hipStream_t)s));
    if (!dg) return 0;
    LnReduceArgs ra{};
    ra.part = (const float*)ws; ra.n = 1; ra.grid = leaf_ln_bwd_grid(rows, d); ra.d = d; ra.inv_s = gscale + 1;
    ra.dg[0] = dg; ra.db[0] = db;
    return leaf_check(leaf_launch_ln_param_reduce(ra, (hipStream_t)s), "ln_param_reduce");
}
extern "C" int leaf_op_project_rows(const float* xg, const float* g, const float* b, float eps, const float* proj,
                                    float* xn_scratch, float* out, int M, int d, int D, int normalize, leaf_stream_t s) {
    if (!leaf_project_rows_ok(d, D)) { leaf_set_error("project_rows needs D %% 128 == 0 and d %% 16 == 0"); return 1; }
    return leaf_check(leaf_launch_project_rows(xg, g, b, eps, proj, xn_scratch, out, M, d, D, normalize, (hipStream_t)s), "project_rows");
}
