// C ABI, training half: forward with activation stash, TextFARE loss + backward into the flat gradient
// buffer, fused AdamW.  Host orchestration only; see include/leaf_hip.h for the contract.
#include "engine.h"

namespace {

struct Stash {
    float* xin;      // [(L+1)][rows,d]  layer inputs; xin[L] = final residual stream
    float* x1;       // [L][rows,d]      after the attention residual
    uint16_t* xn1;   // [L][rows,d]
    uint16_t* qkv;   // [L][rows,3d]
    uint16_t* ao;    // [L][rows,d]
    uint16_t* xn2;   // [L][rows,d]
    uint16_t* pre;   // [L][rows,4d]
    uint16_t* hh;    // [L][rows,4d]
    float* pooled;   // [n,d]
    int32_t* eot;    // [n]
    float* norms;    // [n]  ||features|| of the last training forward (--normalize_fare)
};

size_t total_rows(const leaf_text* h, const int32_t* lens, int n_seq) {
    if (!lens) return (size_t)n_seq * h->cfg.context_length;
    size_t r = 0;
    for (int i = 0; i < n_seq; ++i) r += lens[i];
    return r;
}

Stash carve_stash(const leaf_text* h, Carver& c, int n_seq, size_t rows) {
    const size_t d = h->cfg.width, L = h->cfg.layers;
    Stash s;
    s.xin = (float*)c.take((L + 1) * rows * d * 4);
    s.x1 = (float*)c.take(L * rows * d * 4);
    s.xn1 = (uint16_t*)c.take(L * rows * d * 2);
    s.qkv = (uint16_t*)c.take(L * rows * 3 * d * 2);
    s.ao = (uint16_t*)c.take(L * rows * d * 2);
    s.xn2 = (uint16_t*)c.take(L * rows * d * 2);
    s.pre = (uint16_t*)c.take(L * rows * 4 * d * 2);
    s.hh = (uint16_t*)c.take(L * rows * 4 * d * 2);
    s.pooled = (float*)c.take((size_t)n_seq * d * 4);
    s.eot = (int32_t*)c.take((size_t)n_seq * 4);
    s.norms = (float*)c.take((size_t)n_seq * 4);
    return s;
}

struct BwdBuf {
    float* dx;        // [rows,d]
    uint16_t* dx16;   // [rows,d] 16-bit gradient at the block output (dY of c_proj)
    uint16_t* dx16b;  // [rows,d] 16-bit gradient after the MLP branch (dY of out_proj)
    float* dxn;       // [rows,d]
    uint16_t* big16;  // [rows,4d] bf16
    uint16_t* dqkv;   // [rows,3d] bf16
    uint16_t* do16;   // [rows,d] bf16
    float* dout;      // [n,D]
    float* gscale;    // {S, 1/S}
    float* lnpart;    // [2 layers][leaf_ln_bwd_grid][2][d] per-workgroup partial sums of the LayerNorm parameter gradients
};

BwdBuf carve_bwd(const leaf_text* h, Carver& c, int n_seq, size_t rows) {
    const size_t d = h->cfg.width;
    BwdBuf b;
    b.dx = (float*)c.take(rows * d * 4);
    b.dx16 = (uint16_t*)c.take(rows * d * 2);
    b.dx16b = (uint16_t*)c.take(rows * d * 2);
    b.dxn = (float*)c.take(rows * d * 4 + (size_t)n_seq * 8);   // + {mu, rstd} per sequence (pool backward scratch)
    b.big16 = (uint16_t*)c.take(rows * 4 * d * 2);
    b.dqkv = (uint16_t*)c.take(rows * 3 * d * 2);
    b.do16 = (uint16_t*)c.take(rows * d * 2);
    b.dout = (float*)c.take((size_t)n_seq * h->cfg.embed_dim * 4);
    b.gscale = (float*)c.take(256 + (size_t)n_seq * 8);   // {S, 1/S} + per-caption loss partials
    b.lnpart = (float*)c.take((size_t)2 * h->cfg.layers * leaf_ln_bwd_grid((int)rows, (int)d) * 2 * d * 4);
    return b;
}

}  // namespace

size_t leaf_train_ws_bytes(const leaf_text* h, int n_seq) {
    Carver c(nullptr, 0);
    carve_bwd(h, c, n_seq, (size_t)n_seq * h->cfg.context_length);
    return align_up(c.off, 256) + 256;
}

extern "C" size_t leaf_text_stash_bytes(leaf_text_t h, int n_seq) {
    if (!h || n_seq < 1) return 0;
    Carver c(nullptr, 0);
    carve_stash(h, c, n_seq, (size_t)n_seq * h->cfg.context_length);
    return align_up(c.off, 256) + 256;
}

extern "C" int leaf_text_forward_train_delta(leaf_text_t h, const float* P, const void* w16_fwd, const int32_t* tokens,
                                             const int32_t* seq_lens, const int32_t* cu_rows, int n_seq,
                                             const float* delta, float* out, void* stash, size_t stash_bytes, void* ws,
                                             size_t ws_bytes, leaf_stream_t s_) {
    if (!h || !P || !w16_fwd || !tokens || !out || !stash || n_seq < 1) { leaf_set_error("null/invalid argument"); return 1; }
    hipStream_t s = (hipStream_t)s_;
    if ((seq_lens == nullptr) != (cu_rows == nullptr)) { leaf_set_error("seq_lens (host) and cu_rows (device) go together"); return 1; }
    const leaf_text_cfg& cf = h->cfg;
    const int rows = (int)total_rows(h, seq_lens, n_seq);
    Carver c(stash, stash_bytes);
    Stash st = carve_stash(h, c, n_seq, rows);
    if (!c.ok()) { leaf_set_error("stash too small: need %zu bytes, have %zu", c.off, c.cap); return 1; }
    const RowMap map{cu_rows, 0, 0, cf.context_length, nullptr, nullptr, 1};
    int max_len = 0;
    for (int i = 0; i < n_seq; ++i) { const int Ls = seq_lens ? seq_lens[i] : cf.context_length; max_len = Ls > max_len ? Ls : max_len; }
    const int d = cf.width, dt = h->fwd_dtype, L = cf.layers;
    const size_t rd = (size_t)rows * d;
    const uint16_t* W = (const uint16_t*)w16_fwd;
    LEAF_TRY(leaf_launch_embed_ln(tokens, P + h->tok_emb, P + h->pos_emb, P + h->layer[0].ln1_w, P + h->layer[0].ln1_b,
                                  cf.ln_eps, st.xin, st.xn1, rows, n_seq, map, d, cf.vocab_size, dt, s, delta));
    for (int l = 0; l < L; ++l) {
        const LayerOff& o = h->layer[l];
        float* xin = st.xin + l * rd; float* x1 = st.x1 + l * rd; float* xout = st.xin + (l + 1) * rd;
        uint16_t* xn1 = st.xn1 + l * rd; uint16_t* qkv = st.qkv + 3 * l * rd; uint16_t* ao = st.ao + l * rd;
        uint16_t* xn2 = st.xn2 + l * rd; uint16_t* pre = st.pre + 4 * l * rd; uint16_t* hh = st.hh + 4 * l * rd;
        if (l > 0) LEAF_TRY(leaf_launch_layernorm(xin, P + o.ln1_w, P + o.ln1_b, cf.ln_eps, xn1, rows, d, dt, s));
        if (leaf_gemm(dt, EPI_STORE_T, xn1, d, W + h->w16_qkv(l), d, qkv, 3 * d, P + o.qkv_b, nullptr, rows, 3 * d, d, 0, s)) return 1;
        LEAF_TRY(leaf_launch_attention_fwd(qkv, nullptr, ao, n_seq, map, cf.heads, d, dt, s, nullptr, max_len));
        // out-of-place residual adds (aux = residual source): the stash keeps xin, x1 and xout
        if (leaf_gemm(dt, EPI_RESID_F32, ao, d, W + h->w16_out(l), d, x1, d, P + o.out_b, xin, rows, d, d, 0, s)) return 1;
        LEAF_TRY(leaf_launch_layernorm(x1, P + o.ln2_w, P + o.ln2_b, cf.ln_eps, xn2, rows, d, dt, s));
        if (leaf_gemm(dt, EPI_ACT_T, xn2, d, W + h->w16_fc(l), d, hh, 4 * d, P + o.fc_b, pre, rows, 4 * d, d, cf.activation, s)) return 1;
        if (leaf_gemm(dt, EPI_RESID_F32, hh, 4 * d, W + h->w16_proj(l), 4 * d, xout, d, P + o.proj_b, x1, rows, d, 4 * d, 0, s)) return 1;
    }
    if (leaf_project_rows_ok(d, cf.embed_dim) && ws && ws_bytes >= (size_t)n_seq * d * 4) {
        // pooled rows -> ws, ln_final -> st.pooled (the stash the backward reads), fp32 matrix-core projection -> out
        float* xg = (float*)ws;
        LEAF_TRY(leaf_launch_eot_positions(tokens, st.eot, n_seq, map, s));
        LEAF_TRY(leaf_launch_gather_rows(st.xin + (size_t)L * rd, st.eot, xg, n_seq, map, d, s));
        LEAF_TRY(leaf_launch_project_rows(xg, P + h->lnf_w, P + h->lnf_b, cf.ln_eps, P + h->text_proj, st.pooled, out, n_seq, d,
                                          cf.embed_dim, 0, s));
    } else {
        LEAF_TRY(leaf_launch_pool_project(st.xin + (size_t)L * rd, tokens, P + h->lnf_w, P + h->lnf_b, cf.ln_eps,
                                          P + h->text_proj, out, st.pooled, st.eot, n_seq, map, d, cf.embed_dim, 0, s));
    }
    if (h->normalize_fare) LEAF_TRY(leaf_launch_normalize_rows(out, st.norms, n_seq, cf.embed_dim, s));   // utils_AT.py:319
    return 0;
}

extern "C" int leaf_text_forward_train(leaf_text_t h, const float* P, const void* w16_fwd, const int32_t* tokens,
                                       const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, float* out,
                                       void* stash, size_t stash_bytes, void* ws, size_t ws_bytes, leaf_stream_t s_) {
    return leaf_text_forward_train_delta(h, P, w16_fwd, tokens, seq_lens, cu_rows, n_seq, nullptr, out, stash, stash_bytes,
                                         ws, ws_bytes, s_);
}

// Shared body of the parameter-gradient backward (G != null) and of the input-gradient-only backward of the optional
// embedding-space PGD mode (G == null, d_embed != null: no weight / bias / LayerNorm / embedding-table gradients).
static int backward_impl(leaf_text_t h, const float* P, const void* w16_bwd, const int32_t* tokens,
                         const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, const float* feat,
                         const float* anchor, float accum_scale, const void* stash, float* G, float* d_embed,
                         float* loss_out, void* ws, size_t ws_bytes, leaf_stream_t s_, void* const* layer_events = nullptr) {
    if (!h || !P || !w16_bwd || !tokens || !feat || !anchor || !stash || (!G && !d_embed) || !ws || n_seq < 1) {
        leaf_set_error("null/invalid argument");
        return 1;
    }
    hipStream_t s = (hipStream_t)s_;
    if ((seq_lens == nullptr) != (cu_rows == nullptr)) { leaf_set_error("seq_lens (host) and cu_rows (device) go together"); return 1; }
    const leaf_text_cfg& cf = h->cfg;
    const int rows = (int)total_rows(h, seq_lens, n_seq);
    Carver cs((void*)stash, (size_t)-1);
    Stash st = carve_stash(h, cs, n_seq, rows);
    Carver cw(ws, ws_bytes);
    BwdBuf b = carve_bwd(h, cw, n_seq, rows);
    if (!cw.ok()) { leaf_set_error("workspace too small: need %zu bytes, have %zu", cw.off, cw.cap); return 1; }
    const RowMap map{cu_rows, 0, 0, cf.context_length, nullptr, nullptr, 1};
    const int d = cf.width, L = cf.layers, D = cf.embed_dim;
    const size_t rd = (size_t)rows * d;
    const int fk = h->fwd_dtype == LEAF_DTYPE_FP16 ? 1 : 0;  // source kind of stashed activations
    const int gk = h->grad_dtype == LEAF_DTYPE_FP16 ? 1 : 0;  // 16-bit kind of the gradient path (kind == LEAF_* dtype id)
    const float* inv_s = b.gscale + 1;
    const uint16_t* WT = (const uint16_t*)w16_bwd;

    LEAF_TRY(leaf_launch_fare_loss(feat, anchor, n_seq, D, accum_scale, loss_out, b.dout, b.gscale, gk == 1, s,
                                   h->normalize_fare ? st.norms : nullptr, h->scaler));
    const bool sat_check = h->scaler && G && gk == 1;     // fp16 gradient path with a scaler attached (leaf_hip.h "gradient scaler")
    LEAF_TRY(hipMemsetAsync(b.dx, 0, rd * 4, s));
    LEAF_TRY(leaf_launch_pool_project_bwd(b.dout, st.pooled, st.xin + (size_t)L * rd, st.eot, P + h->lnf_w, P + h->lnf_b,
                                          cf.ln_eps, P + h->text_proj, b.dx, G ? G + h->text_proj : nullptr,
                                          G ? G + h->lnf_w : nullptr, G ? G + h->lnf_b : nullptr, b.gscale, n_seq, map, d, D,
                                          b.dxn /* free until the first dgrad */, s));
    LEAF_TRY(leaf_launch_cast16(b.dx, 2, b.dx16, gk, rd, s));

    const int ln_grid = leaf_ln_bwd_grid(rows, d);
    const size_t ln_stride = (size_t)ln_grid * 2 * d;      // floats per LayerNorm in b.lnpart
    for (int l = L - 1; l >= 0; --l) {
        const LayerOff& o = h->layer[l];
        const float* xin = st.xin + l * rd; const float* x1 = st.x1 + l * rd;
        const uint16_t* xn1 = st.xn1 + l * rd; const uint16_t* qkv = st.qkv + 3 * l * rd; const uint16_t* ao = st.ao + l * rd;
        const uint16_t* xn2 = st.xn2 + l * rd; const uint16_t* pre = st.pre + 4 * l * rd; const uint16_t* hh = st.hh + 4 * l * rd;
        // ---- MLP
        if (leaf_gemm(gk, EPI_ACTGRAD_T, b.dx16, d, WT + h->w16_proj(l), d, b.big16, 4 * d, nullptr, (void*)pre, rows,
                 4 * d, d, cf.activation, s, 0.f, fk)) return 1;
        if (leaf_gemm(gk, EPI_STORE_F32, b.big16, 4 * d, WT + h->w16_fc(l), 4 * d, b.dxn, d, nullptr, nullptr, rows, d,
                 4 * d, 0, s)) return 1;
        LEAF_TRY(leaf_launch_layernorm_bwd(b.dxn, x1, P + o.ln2_w, cf.ln_eps, b.dx, b.dx16b, gk,
                                           G ? b.lnpart + (size_t)(2 * l + 1) * ln_stride : nullptr, rows, d, s));
        // ---- attention
        if (leaf_gemm(gk, EPI_STORE_T, b.dx16b, d, WT + h->w16_out(l), d, b.do16, d, nullptr, nullptr, rows, d, d, 0, s)) return 1;
        LEAF_TRY(leaf_launch_attention_bwd(qkv, h->fwd_dtype, b.do16, b.dqkv, gk, n_seq, map, cf.heads, d, s));
        if (G) {
            // all four weight + bias gradients of the block in one launch (wgrad.hip); must run before ln1's backward
            // overwrites dx16, the dY of c_proj
            WgradArgs wa{};
            wa.p[0] = WgradProb{b.dx16, hh, G + o.proj_w, G + o.proj_b, d, 4 * d, d, 4 * d, 0, 0};
            wa.p[1] = WgradProb{b.big16, xn2, G + o.fc_w, G + o.fc_b, 4 * d, d, 4 * d, d, 0, 0};
            wa.p[2] = WgradProb{b.dx16b, ao, G + o.out_w, G + o.out_b, d, d, d, d, 0, 0};
            wa.p[3] = WgradProb{b.dqkv, xn1, G + o.qkv_w, G + o.qkv_b, 3 * d, d, 3 * d, d, 0, 0};
            wa.nprob = 4; wa.rows = rows; wa.alpha = inv_s;
            LEAF_TRY(leaf_launch_wgrad_group(wa, fk, gk, s));
        }
        if (sat_check) {
            // the five 16-bit gradient tensors of this block, before ln1's backward overwrites dx16
            const void* bufs[5] = {b.dx16, b.big16, b.dx16b, b.do16, b.dqkv};
            const size_t numel[5] = {rd, 4 * rd, rd, rd, 3 * rd};
            // poison target: the first element of the token-embedding gradient.  Everything that still touches it after this
            // launch only ADDS to it (the embedding scatter's atomics: NaN stays NaN), and it travels in the LAST bucket of the
            // data-parallel reduction (step.bucket_plan: offset 0, behind the last block's event), so no rank can ship a clean
            // element 0 before another rank's poison exists (tests/test_dp_gloo.py pins both facts)
            LEAF_TRY(leaf_launch_sat_check16(bufs, numel, 5, h->scaler, G + h->tok_emb, s));
        }
        if (leaf_gemm(gk, EPI_STORE_F32, b.dqkv, 3 * d, WT + h->w16_qkv(l), 3 * d, b.dxn, d, nullptr, nullptr, rows, d,
                 3 * d, 0, s)) return 1;
        LEAF_TRY(leaf_launch_layernorm_bwd(b.dxn, xin, P + o.ln1_w, cf.ln_eps, b.dx, b.dx16, gk,
                                           G ? b.lnpart + (size_t)(2 * l) * ln_stride : nullptr, rows, d, s));
        // every gradient of block l is final here: the caller may start reducing this block's bucket (step.py)
        if (layer_events && layer_events[l]) LEAF_TRY(hipEventRecord((hipEvent_t)layer_events[l], s));
    }
    if (G) {
        // LayerNorm parameter gradients of all blocks: one reduction of the per-workgroup partials (LN_REDUCE_MAX per launch)
        for (int i0 = 0; i0 < 2 * L; i0 += LN_REDUCE_MAX) {
            LnReduceArgs ra{};
            ra.n = 2 * L - i0 < LN_REDUCE_MAX ? 2 * L - i0 : LN_REDUCE_MAX;
            ra.part = b.lnpart + (size_t)i0 * ln_stride; ra.grid = ln_grid; ra.d = d; ra.inv_s = inv_s;
            for (int i = 0; i < ra.n; ++i) {
                const LayerOff& o = h->layer[(i0 + i) >> 1];
                const bool second = (i0 + i) & 1;
                ra.dg[i] = G + (second ? o.ln2_w : o.ln1_w);
                ra.db[i] = G + (second ? o.ln2_b : o.ln1_b);
            }
            LEAF_TRY(leaf_launch_ln_param_reduce(ra, s));
        }
    }
    if (G) LEAF_TRY(leaf_launch_embed_bwd(b.dx, b.gscale, tokens, G + h->tok_emb, G + h->pos_emb, rows, n_seq, map, d, cf.vocab_size, s));
    if (d_embed) LEAF_TRY(leaf_launch_scale_copy(b.dx, inv_s, d_embed, rd, s));   // d loss / d (token embedding), un-scaled
    if (layer_events && layer_events[L]) LEAF_TRY(hipEventRecord((hipEvent_t)layer_events[L], s));
    return 0;
}

extern "C" int leaf_text_set_grad_scaler(leaf_text_t h, float* state) {
    if (!h) { leaf_set_error("null handle"); return 1; }
    if (state && h->tok_emb != 0) { leaf_set_error("gradient scaler: the poison element must be offset 0 of the flat buffer"); return 1; }
    h->scaler = state;
    return 0;
}

extern "C" int leaf_textfare_backward(leaf_text_t h, const float* P, const void* w16_bwd, const int32_t* tokens,
                                      const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, const float* feat,
                                      const float* anchor, float accum_scale,
                                      const void* stash, float* G, float* loss_out, void* ws, size_t ws_bytes,
                                      leaf_stream_t s_) {
    if (!G) { leaf_set_error("null/invalid argument"); return 1; }
    return backward_impl(h, P, w16_bwd, tokens, seq_lens, cu_rows, n_seq, feat, anchor, accum_scale, stash, G, nullptr,
                         loss_out, ws, ws_bytes, s_);
}

// The same backward with completion events for gradient-bucket overlap: layer_events[l], l = L-1 .. 0, is recorded
// on the stream as soon as every gradient of transformer block l is final in `grads`, layer_events[L] after the last kernel
// (embedding tables; ln_final / text_projection gradients are written first).  NULL entries are skipped.
extern "C" int leaf_textfare_backward_events(leaf_text_t h, const float* P, const void* w16_bwd, const int32_t* tokens,
                                             const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, const float* feat,
                                             const float* anchor, float accum_scale, const void* stash, float* G,
                                             float* loss_out, void* ws, size_t ws_bytes, leaf_stream_t s_,
                                             void* const* layer_events) {
    if (!G) { leaf_set_error("null/invalid argument"); return 1; }
    return backward_impl(h, P, w16_bwd, tokens, seq_lens, cu_rows, n_seq, feat, anchor, accum_scale, stash, G, nullptr,
                         loss_out, ws, ws_bytes, s_, layer_events);
}

extern "C" int leaf_textfare_input_grad(leaf_text_t h, const float* P, const void* w16_bwd, const int32_t* tokens,
                                        const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, const float* feat,
                                        const float* anchor, const void* stash, float* d_embed, float* loss_out, void* ws,
                                        size_t ws_bytes, leaf_stream_t s_) {
    if (!d_embed) { leaf_set_error("null/invalid argument"); return 1; }
    return backward_impl(h, P, w16_bwd, tokens, seq_lens, cu_rows, n_seq, feat, anchor, 1.0f, stash, nullptr, d_embed,
                         loss_out, ws, ws_bytes, s_);
}

extern "C" int leaf_pgd_step(float* delta, const float* grad, const int32_t* cu_rows, int n_seq, int ctx, int width,
                             float alpha, float eps, int norm, leaf_stream_t s_) {
    if (!delta || !grad || n_seq < 1 || ctx < 1 || (norm != 0 && norm != 2) || alpha < 0.f || eps < 0.f) {
        leaf_set_error("null/invalid argument (norm: 0 = linf, 2 = l2)");
        return 1;
    }
    const RowMap map{cu_rows, 0, 0, ctx, nullptr, nullptr, 1};
    return leaf_check(leaf_launch_pgd_step(delta, grad, n_seq, map, width, alpha, eps, norm == 2, (hipStream_t)s_), "pgd_step");
}

extern "C" int leaf_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                               size_t n_decay, float lr, float beta1, float beta2, float eps, float wd, int step,
                               float grad_scale, leaf_stream_t s) {
    if (!params || !grads || !exp_avg || !exp_avg_sq) { leaf_set_error("null argument"); return 1; }
    return leaf_check(leaf_launch_adamw(params, grads, exp_avg, exp_avg_sq, n, n_decay, lr, beta1, beta2, eps, wd, step,
                                        grad_scale, (hipStream_t)s), "adamw");
}

extern "C" int leaf_clip_grads_inplace(float* grads, size_t n, float pre_scale, float max_norm, float* ws, leaf_stream_t s) {
    if (!grads || !ws || !(max_norm > 0.f) || !(pre_scale > 0.f)) { leaf_set_error("null/invalid argument"); return 1; }
    return leaf_check(leaf_launch_clip_inplace(grads, n, pre_scale, max_norm, ws, (hipStream_t)s), "clip_grads_inplace");
}

extern "C" int leaf_adamw_step_clip(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                                    size_t n_decay, float lr, float beta1, float beta2, float eps, float wd, int step,
                                    float grad_scale, float max_norm, float* clip_ws, leaf_stream_t s) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !clip_ws || !(max_norm > 0.f)) { leaf_set_error("null/invalid argument"); return 1; }
    return leaf_check(leaf_launch_adamw(params, grads, exp_avg, exp_avg_sq, n, n_decay, lr, beta1, beta2, eps, wd, step,
                                        grad_scale, (hipStream_t)s, max_norm, clip_ws), "adamw_clip");
}
