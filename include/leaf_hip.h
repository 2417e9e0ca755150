/* libleaf_hip.so -- C ABI of the MI355X-native LEAF text hot path.
 *
 * The reference (LIONS-EPFL/LEAF) has no FFI: its seams are Python call signatures.
 * Each entry point below names the reference call site it replaces.  Conventions:
 *   - plain C symbols, `int` status (0 = ok), message via leaf_last_error(); no exceptions cross the ABI;
 *   - the library allocates NO caller-visible device memory: the caller (PyTorch) owns parameters,
 *     gradients, optimizer state, activations and workspace and passes raw device pointers + a hipStream_t;
 *   - every call is asynchronous on the given stream; a handle is host-only state, not thread-safe (it owns one small ring of
 *     PINNED HOST buffers for the tile plans of the fused QKV + attention launches, freed by leaf_text_destroy);
 *   - tokens are int32 [n_seq, ctx] row-major, pad id 0, EOT = row maximum (open_clip tokenizer.py:256-263).
 *
 * Parameter buffer: ONE flat fp32 array (leaf_text_param_count floats).  Tensors that receive weight
 * decay (train_AT_text_only.py:323-331: ndim >= 2 and no "bn"/"ln"/"bias"/"logit_scale" in the name) come
 * first (leaf_text_decay_count floats), the rest after; leaf_text_param_info enumerates
 * (open_clip state_dict key, offset, shape).  Gradients and AdamW moments use the same layout, so data
 * parallelism is a single flat all-reduce.
 */
#ifndef LEAF_HIP_H
#define LEAF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct leaf_text_cfg {
    int32_t layers, width, heads, embed_dim; /* src/open_clip/model_configs/ViT-*.json text_cfg + embed_dim */
    int32_t context_length;                  /* 77 */
    int32_t vocab_size;                      /* 49408 */
    int32_t activation;                      /* 0 = nn.GELU (erf), 1 = QuickGELU (transformer.py:33-36) */
    float ln_eps;                            /* 1e-5 */
} leaf_text_cfg;

typedef struct leaf_text* leaf_text_t;
typedef void* leaf_stream_t; /* hipStream_t */

enum { LEAF_DTYPE_BF16 = 0, LEAF_DTYPE_FP16 = 1 };                               /* MFMA operand type */
enum { LEAF_OBJ_L2 = 0, LEAF_OBJ_NEGL2 = 1, LEAF_OBJ_DISSIM = 2, LEAF_OBJ_SIM = 3 }; /* utils_attacks.py:332-346 */

const char* leaf_last_error(void);
int leaf_version(void);

/* handle: replaces open_clip.factory.create_model's text tower construction (model.py:173-217) */
int leaf_text_create(const leaf_text_cfg* cfg, int fwd_dtype, leaf_text_t* out);
void leaf_text_destroy(leaf_text_t h);
int leaf_text_set_chunk(leaf_text_t h, int seqs_per_chunk); /* sequences processed per pass through the layers */
/* options: "chunk" (as above), "last_layer_trim" (0/1, default 1: the last block's attention output, out-projection and
 * MLP are computed for the pooled EOT row only -- exact, every op after attention is row-wise), "compact_resid" (0/1, default 1:
 * the residual stream of the LN-folded forward-only passes as the 16-bit copy + a remainder byte per element instead of an fp32
 * row beside that copy -- see leaf_op_gemm_resid_ln8 below; 0, or LEAF_COMPACT_RESID=0, keeps fp32 rows), and the A/B switches "streams", "ln_fold", "fuse_attn", "normalize_fare" */
int leaf_text_set_option(leaf_text_t h, const char* name, int value);
/* current value of an option of leaf_text_set_option (+ "split_blocks"); -1 = unknown name */
int leaf_text_get_option(leaf_text_t h, const char* name);

/* flat parameter layout */
size_t leaf_text_param_count(leaf_text_t h);
size_t leaf_text_decay_count(leaf_text_t h);
int leaf_text_num_tensors(leaf_text_t h);
int leaf_text_param_info(leaf_text_t h, int index, char* name, size_t name_len, size_t* offset, int64_t* rows,
                         int64_t* cols); /* cols = 0 for 1-D tensors */

/* 16-bit operand copies of the GEMM weights.  w16_fwd: [N,K] as stored, forward dtype.  w16_bwd (may be
 * NULL): transposed [K,N] copies for the data-gradient GEMMs in the gradient path's 16-bit type (fp16 with a
 * per-step power-of-two loss scale by default, bf16 unscaled with LEAF_GRAD_DTYPE=bf16).  Call after every optimizer step. */
size_t leaf_text_w16_bytes(leaf_text_t h);
int leaf_text_pack_weights(leaf_text_t h, const float* params, void* w16_fwd, void* w16_bwd, leaf_stream_t s);

/* Split blocks: higher-precision GEMMs in the leading transformer blocks of the forward-only passes (encode_text, score_candidates*,
 * forward_kv; the training forward is untouched -- the reference trains under fp16 autocast, utils_AT.py:317-319).  A split GEMM
 * multiplies hi + lo 16-bit splits (x_hi W_hi + x_lo W_hi + x_hi W_lo in the fp32 accumulator, over a longer K through the unchanged
 * GEMM kernels -- the fused QKV + attention launch included; stored q|k|v / attention / hidden rows stay 16-bit).  The embedding's
 * error is made early (profiles/r06_precision_budget_sites.txt).  leaf_text_split_pack: all four GEMMs of the
 * first `blocks` blocks (QKV / c_fc both operands, out_proj / c_proj weights).  leaf_text_split_pack_masks: one mask per block,
 * bit 0 = QKV, bit 1 = out_proj, bit 2 = c_fc, bit 3 = c_proj -- leaf_amd's default arithmetic is such a mask list
 * (leaf_amd/model.py PRECISION_MODES; profiles/r06_row_error_census*.txt).  Works on both residual-stream formats
 * ("compact_resid").  `buf` (DEVICE, leaf_text_split_bytes(h, blocks) bytes, caller-owned, must outlive the calls that use it)
 * receives the split weight copies; call again after every optimizer step (as leaf_text_pack_weights).  blocks = 0 (buf ignored)
 * switches splits off.  0 <= blocks <= min(layers - 1, 64); needs option ln_fold = 1. */
size_t leaf_text_split_bytes(leaf_text_t h, int blocks);
int leaf_text_split_pack(leaf_text_t h, const float* params, int blocks, void* buf, leaf_stream_t s);
int leaf_text_split_pack_masks(leaf_text_t h, const float* params, const int32_t* masks, int blocks, void* buf, leaf_stream_t s);

/* workspace sizes (bytes): mode 0 = forward, 1 = score_candidates, 2 = train backward */
size_t leaf_text_workspace_bytes(leaf_text_t h, int n_seq, int mode);
size_t leaf_text_stash_bytes(leaf_text_t h, int n_seq);

/* Exact work skipping ("EOT trimming"): the causal mask makes a sequence's pooled EOT state independent of every
 * position after EOT, so only the first len = eot_pos + 1 rows of each sequence need computing.  Entry points that
 * run the layer stack take an OPTIONAL pair: seq_lens (HOST int32 [n_seq], 1..ctx) and cu_rows (DEVICE int32
 * [n_seq+1], exclusive prefix sum of seq_lens).  Both NULL = dense (ctx rows per sequence).  Results are
 * bit-identical either way (tests/test_gpu_forward.py::test_packed_rows_are_bit_exact). */

/* CLIP.encode_text (src/open_clip/model.py:269-284): tokens [n_seq,ctx] -> out fp32 [n_seq,embed_dim] */
int leaf_text_forward(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                      const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, float* out, int normalize, void* ws,
                      size_t ws_bytes, leaf_stream_t s);

/* The same function in fp32-grade arithmetic ("precise" mode, leaf_amd/csrc/precise.hip): fp32 stored intermediates, the fp32
 * MASTER weights (no 16-bit pack is read) and three MFMA passes of on-the-fly fp16 hi / lo splits per GEMM, fp32 LayerNorm /
 * softmax / activation.  For embeddings that leave the engine (export, eval_textfare.py:119-141) and, optionally, the frozen
 * model's anchor pass (utils_AT.py:296); B-caption sized work, about 8x the time per row of leaf_text_forward.  Same row-plan
 * arguments (seq_lens host / cu_rows device, both NULL = dense). */
size_t leaf_text_precise_workspace_bytes(leaf_text_t h, int n_seq);
int leaf_text_forward_precise(leaf_text_t h, const float* params, const int32_t* tokens, const int32_t* seq_lens,
                              const int32_t* cu_rows, int n_seq, float* out, int normalize, void* ws, size_t ws_bytes,
                              leaf_stream_t s);

/* one search stage of attack_text_leaf (utils_attacks.py:330-348 / 368-386,393): forward of B*rho candidates,
 * loss per objective against anchor [B,embed_dim], first-index arg-max over rho, gather of the winning rows.
 * loss (fp32 [B,rho]) and best_feat (fp32 [B,embed_dim]) may be NULL. */
int leaf_score_candidates(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                          const int32_t* seq_lens, const int32_t* cu_rows, const float* anchor, int B, int rho,
                          int objective, int32_t* best_idx, float* best_feat, float* loss, void* ws, size_t ws_bytes,
                          leaf_stream_t s);

/* Exact work skipping, part 2 ("prefix reuse"): a search candidate equals its clean caption at every
 * position before its first changed token, and under the causal mask those rows are identical at every layer.
 * leaf_text_forward_kv runs the clean captions once and keeps each layer's q|k|v rows in `kv` (leaf_text_kv_bytes);
 * leaf_score_candidates_prefix then computes, for candidate i of caption i / rho, only positions
 * [prefix[i], prefix[i] + suffix_lens[i]) and reads the K/V of earlier positions from the cache.  cu_suffix = exclusive
 * prefix sum of suffix_lens (device), base_cu = packed row offsets of the captions inside the cache (device, [B+1]),
 * base_rows = total cached rows per layer, max_len = an upper bound of prefix[i] + suffix_lens[i] over the candidates
 * (sizes the attention kernel's LDS; 0 = context_length).  Results are bit-identical to leaf_score_candidates. */
size_t leaf_text_kv_bytes(leaf_text_t h, int n_seq);
int leaf_text_forward_kv(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                         const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, float* out /* may be NULL */,
                         int normalize, void* kv, size_t kv_bytes, void* ws, size_t ws_bytes, leaf_stream_t s);
int leaf_score_candidates_prefix(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                                 const int32_t* suffix_lens, const int32_t* cu_suffix, const int32_t* prefix,
                                 const int32_t* base_cu, const void* kv, size_t base_rows, int max_len,
                                 const float* anchor, int B, int rho, int objective, int32_t* best_idx, float* best_feat, float* loss, void* ws,
                                 size_t ws_bytes, leaf_stream_t s);

/* leaf_text_forward_kv and the FIRST stage's leaf_score_candidates_prefix in one pass (the clean-caption pass is a chain of ~80
 * small launches; here its 3,200 rows ride in the candidates' launches): tokens [(B + B*rho), ctx] with the B captions FIRST,
 * seq_lens (host) = rows to compute per sequence (captions: their length; candidates: suffix length), cu_rows = exclusive
 * prefix sum (device, [B + B*rho + 1]), prefix (device; 0 for the captions).  `kv` receives the captions' per-layer q|k|v rows
 * exactly as leaf_text_forward_kv writes them (later stages use leaf_score_candidates_prefix with it).  Bit-identical
 * results.  Returns 2 when the rows do not fit one chunk (use the two separate calls then). */
int leaf_score_candidates_prefix_fused(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                                       const int32_t* seq_lens, const int32_t* cu_rows, const int32_t* prefix, int max_len,
                                       const float* anchor, int B, int rho, int objective, int32_t* best_idx, float* best_feat,
                                       float* loss, void* kv, size_t kv_bytes, void* ws, size_t ws_bytes, leaf_stream_t s);

/* training forward (utils_AT.py:317-319) keeping activations in `stash` for the backward pass */
int leaf_text_forward_train(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                            const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, float* out, void* stash,
                            size_t stash_bytes, void* ws, size_t ws_bytes, leaf_stream_t s);

/* OPTIONAL embedding-space PGD mode (no reference path runs it on text, the pieces are the
 * reference's): forward of tokens whose embeddings carry an additive perturbation delta (fp32 [rows, width], packed like
 * the activations, resident in HBM across the inner iterations; NULL = none) - the embedding-input forward of
 * src/pez/open_clip_pez/model.py:210-228 fused into the embedding gather + first LayerNorm. */
int leaf_text_forward_train_delta(leaf_text_t h, const float* params, const void* w16_fwd, const int32_t* tokens,
                                  const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, const float* delta,
                                  float* out, void* stash, size_t stash_bytes, void* ws, size_t ws_bytes, leaf_stream_t s);
/* TextFARE loss of (anchor, feat) and ONLY its gradient with respect to the token embeddings / delta (fp32
 * [rows, width], un-scaled): the backward of the continuous attack loop (utils_attacks.py:683-692) without any
 * parameter gradient.  loss_out: one device float (mean over captions of the squared distance). */
int leaf_textfare_input_grad(leaf_text_t h, const float* params, const void* w16_bwd, const int32_t* tokens,
                             const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, const float* feat,
                             const float* anchor, const void* stash, float* d_embed, float* loss_out, void* ws,
                             size_t ws_bytes, leaf_stream_t s);
/* fused grad-sign / normalise - step - project on the resident perturbation, per sequence over its kept rows:
 * norm 0 (linf): delta <- clamp(delta + alpha sign(grad), -eps, eps)       (utils_attacks.py:693-694)
 * norm 2 (l2)  : delta <- renorm_2(delta + alpha grad / ||grad||_2, eps)   (src/robust_vlm/train/utils.py:96-114)
 * cu_rows: device prefix sums of the kept rows ([n_seq + 1]) or NULL for dense ctx rows per sequence. */
int leaf_pgd_step(float* delta, const float* grad, const int32_t* cu_rows, int n_seq, int ctx, int width, float alpha,
                  float eps, int norm, leaf_stream_t s);

/* TextFARE loss + backward (utils_AT.py:321-337): loss = mean_b sum_j (anchor - feat)^2; back-propagates
 * loss * accum_scale (= 1/accum_freq) and ACCUMULATES (+=) into grads (flat fp32, parameter layout).
 * loss_out: one device float (unscaled loss). */
int leaf_textfare_backward(leaf_text_t h, const float* params, const void* w16_bwd, const int32_t* tokens,
                           const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, const float* feat,
                           const float* anchor, float accum_scale, const void* stash, float* grads, float* loss_out,
                           void* ws, size_t ws_bytes, leaf_stream_t s);

/* the same backward, recording completion events for the overlap of the data-parallel gradient reduction with the backward
 * (SURVEY.md 8e): layer_events = L + 1 hipEvent_t (NULL entries skipped); [l] is recorded on `s` once every gradient of
 * transformer block l is final in `grads` (blocks finish in the order L-1 .. 0), [L] after the last kernel. */
int leaf_textfare_backward_events(leaf_text_t h, const float* params, const void* w16_bwd, const int32_t* tokens,
                                  const int32_t* seq_lens, const int32_t* cu_rows, int n_seq, const float* feat,
                                  const float* anchor, float accum_scale, const void* stash, float* grads, float* loss_out,
                                  void* ws, size_t ws_bytes, leaf_stream_t s, void* const* layer_events);

/* torch.optim.AdamW step over the flat buffers (train_AT_text_only.py:326-341): decoupled weight decay `wd`
 * on the first n_decay elements, 0 on the rest; step counts from 1; grads are multiplied by grad_scale first
 * (1/world_size after a sum all-reduce). */
int leaf_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, size_t n_decay,
                    float lr, float beta1, float beta2, float eps, float wd, int step, float grad_scale,
                    leaf_stream_t s);
/* GRADIENT SCALER (the reference trains under fp16 autocast with torch.cuda.amp.GradScaler: train_AT_text_only.py:347,
 * utils_AT.py:79-83,339-362).  The fp16 gradient path multiplies the loss gradient by a power of two S chosen on the device
 * each step (max|d loss / d feat| * S in [8, 16)) and its 16-bit conversions SATURATE at +-65504 instead of producing inf, so
 * an overflow would clip silently.  With a scaler state attached (leaf_text_set_grad_scaler) the backward checks every 16-bit
 * gradient tensor of every block for saturated / non-finite values; a hit poisons gradient element 0 with NaN (the first element
 * of the token-embedding gradient: every later writer only adds to it, and it is reduced with the last data-parallel bucket), so the guarded optimizer step below -- after the data-parallel all-reduce,
 * hence on every rank alike -- skips the step, and halves S for the following steps (state[LEAF_SC_BACKOFF] -= 1); after
 * state[LEAF_SC_INTERVAL] (default 2000, GradScaler's growth_interval) applied steps in a row the back-off is taken back by
 * one halving.  The state is the head of the clip_ws buffer of leaf_adamw_step_clip: device fp32 [LEAF_SC_WORDS + 2048],
 * zeroed once by the caller. */
enum {
    LEAF_SC_COEF = 0,      /* clip coefficient applied by the last guarded step, -1 = that step was skipped */
    LEAF_SC_NORM = 1,      /* total gradient norm seen by the last guarded step */
    LEAF_SC_SKIPPED = 2,   /* optimizer steps skipped so far (non-finite norm: overflow, saturation, NaN) */
    LEAF_SC_BACKOFF = 3,   /* <= 0: log2 of the persistent factor on the per-step loss scale */
    LEAF_SC_GOOD = 4,      /* applied steps since the scale last changed */
    LEAF_SC_SAT_FLAG = 5,  /* a 16-bit gradient tensor saturated during the current optimizer step */
    LEAF_SC_SAT_STEPS = 6, /* skipped steps that had the saturation flag up */
    LEAF_SC_INTERVAL = 7,  /* growth interval in applied steps (0 = 2000) */
    LEAF_SC_APPLIED = 8,   /* AdamW steps really taken = the `step` of the bias corrections (a skipped step does not count) */
    LEAF_SC_BC1 = 9, LEAF_SC_SQRT_BC2 = 10, /* bias corrections of the last applied step */
    LEAF_SC_WORDS = 16
};
/* attach (or detach with NULL) the scaler state used by leaf_textfare_backward* of this handle */
int leaf_text_set_grad_scaler(leaf_text_t h, float* state);

/* the same step preceded by torch.nn.utils.clip_grad_norm_(parameters, max_norm, 2.0) (--grad-clip-norm,
 * utils_AT.py:348-357): total norm = grad_scale * ||grads||_2, gradients are multiplied by min(1, max_norm / (norm + 1e-6))
 * inside the AdamW kernel (grads themselves are left untouched).  clip_ws: fp32 device scratch [LEAF_SC_WORDS + 2048], zeroed
 * once by the caller (layout above).  NON-FINITE GUARD: when the norm is inf / NaN the whole step is skipped (parameters and
 * moments untouched, GradScaler.step semantics), clip_ws[LEAF_SC_COEF] = -1 and clip_ws[LEAF_SC_SKIPPED] counts the skipped
 * steps.  `step` counts ATTEMPTED steps from 1; the bias corrections use step - skipped (clip_ws[LEAF_SC_APPLIED]).
 * max_norm = +inf gives the guard without clipping. */
int leaf_adamw_step_clip(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, size_t n_decay,
                         float lr, float beta1, float beta2, float eps, float wd, int step, float grad_scale,
                         float max_norm, float* clip_ws, leaf_stream_t s);

/* torch.nn.utils.clip_grad_norm_ IN PLACE, for --grad-clip-norm together with --accum-freq > 1 (utils_AT.py:348-357 clips the
 * running gradient sum after every micro-batch's backward, not only before the step): grads <- grads * pre_scale *
 * min(1, max_norm / (pre_scale * ||grads||_2 + 1e-6)).  pre_scale folds a pending factor into the same pass (data parallel: the
 * replicated running sum is divided by the world size before the next micro-batch's local gradients are added and summed over
 * ranks); max_norm = +inf only applies pre_scale.  A non-finite norm leaves the gradients untouched (the step's guard skips).
 * ws: fp32 device scratch [2 + 2048]; ws[0] = the factor applied, ws[1] = the norm. */
int leaf_clip_grads_inplace(float* grads, size_t n, float pre_scale, float max_norm, float* ws, leaf_stream_t s);

/* ---- native host side of the search: CLIP BPE + single-edit mutation, multithreaded ----
 * leaf_tok_create takes the DECOMPRESSED text of bpe_simple_vocab_16e6.txt (src/open_clip/tokenizer.py:139-150).
 * Fast path = printable ASCII without '&'; other inputs are flagged in `fallback` (1 byte per text / per candidate)
 * and must be tokenised by the caller's reference-equivalent tokenizer.  lens[i] = EOT position + 1. */
typedef struct leaf_tok* leaf_tok_t;
int leaf_tok_create(const char* merges_text, size_t len, leaf_tok_t* out);
void leaf_tok_destroy(leaf_tok_t tk);
/* SimpleTokenizer.__call__ (tokenizer.py:226-265): texts -> int32 [n, ctx] */
int leaf_tok_encode_batch(leaf_tok_t tk, const char* const* texts, const int32_t* text_len, int n, int ctx,
                          int32_t* tokens, int32_t* lens, uint8_t* fallback, int n_threads);
/* generate_sentence(S, z, u, V, alternative=-1) (utils_attacks.py:169-213) for B sentences x rho (z, c = V[u]) pairs,
 * then tokenisation of every candidate: -> int32 [B*rho, ctx] */
int leaf_tok_mutate_encode(leaf_tok_t tk, const char* const* sentences, const int32_t* sent_len, int B, const int32_t* z,
                           const int32_t* c, int rho, int ctx, int32_t* tokens, int32_t* lens, uint8_t* fallback,
                           int n_threads);

/* --constrain (utils_attacks.py:110-143,321-325,360-364): valid[b * rho + r] = 1 iff candidate generate_sentence(S_b, z, c,
 * alternative = -1) holds STRICTLY FEWER distinct dictionary words than S_b.  The word set is built once (leaf_dict_create:
 * '\n'-separated words); only the whitespace-delimited window around the edit is re-tokenised.  tokenizer_kind 0 = the regex
 * tokenizer [A-Za-z0-9]+|[^\sA-Za-z0-9] (exact for all ASCII), 1 = nltk.word_tokenize: NLTKWordTokenizer's substitution pipeline
 * (nltk/tokenize/destructive.py, third-party: requirements.txt:14) restated window-locally, exact for ASCII text whose tokens
 * cannot depend on the Punkt sentence model -- a sentence / candidate in which a lone '.' ends a whitespace-delimited chunk
 * before the end of the text gets fallback = 1 from THIS entry point and is decided by the caller with the real nltk
 * (leaf_tok_constrain_ranges / leaf_tok_constrain_punkt below decide those natively too). */
typedef struct leaf_dict* leaf_dict_t;
int leaf_dict_create(const char* words, size_t len, leaf_dict_t* out);
void leaf_dict_destroy(leaf_dict_t d);
int64_t leaf_dict_size(leaf_dict_t d);
int leaf_tok_constrain(leaf_dict_t d, int tokenizer_kind, const char* const* sentences, const int32_t* sent_len, int B,
                       const int32_t* z, const int32_t* c, int rho, uint8_t* valid, uint8_t* fallback, int n_threads);
/* the same with the caller's SENTENCE SEGMENTATION (tokenizer_kind 1 only): nltk.word_tokenize tokenises sentence by sentence, and
 * where a sentence ends is decided by nltk's trained Punkt model, which is not restated.  ranges / ranges_off: for sentence b the
 * spans [ranges[2 i], ranges[2 i + 1]) , i in [ranges_off[b], ranges_off[b + 1]), as PunktSentenceTokenizer.span_tokenize returns them
 * for the lower-cased caption (none = handled as by leaf_tok_constrain).  A candidate is decided natively when its edit window lies
 * inside one span, holds no '.', '?' or '!' before or after the edit and does not directly follow a token that ends in one (Punkt
 * looks at the token carrying the period and the one after it); the others get fallback = 1. */
int leaf_tok_constrain_ranges(leaf_dict_t d, int tokenizer_kind, const char* const* sentences, const int32_t* sent_len, int B,
                              const int32_t* z, const int32_t* c, int rho, const int32_t* ranges, const int32_t* ranges_off,
                              uint8_t* valid, uint8_t* fallback, int n_threads);
/* nltk's Punkt sentence splitter restated (nltk/tokenize/punkt.py PunktSentenceTokenizer.span_tokenize, realign_boundaries = True;
 * third-party, requirements.txt:14): the algorithm is fixed, the trained model only fills four tables.  The tables as '\n'-separated
 * UTF-8 lines: abbreviation types; collocations "first\tsecond"; frequent sentence starters; orthographic contexts "type\tflags"
 * (PunktParameters.abbrev_types / collocations / sent_starters / ortho_context).  ASCII text without control characters only:
 * leaf_punkt_spans returns 2 for anything else.  spans: (start, end) pairs, at most cap_pairs (1 = overflow / bad arguments). */
typedef struct leaf_punkt* leaf_punkt_t;
int leaf_punkt_create(const char* abbrev, size_t abbrev_len, const char* colloc, size_t colloc_len, const char* starters,
                      size_t starters_len, const char* ortho, size_t ortho_len, leaf_punkt_t* out);
void leaf_punkt_destroy(leaf_punkt_t p);
/* strict = 1 (the default): a text in which one whitespace-delimited chunk holds two or more candidate break positions ("what?! yes",
 * "wow!!! nice") is declined (2): nltk 3.6.6 rewrote the scan for period contexts and the generations are only provably equal
 * elsewhere.  strict = 0 decides every text as nltk 3.6.5 does (the generation the golden vectors were made with). */
int leaf_punkt_set_strict(leaf_punkt_t p, int strict);
int leaf_punkt_spans(leaf_punkt_t p, const char* text, int len, int32_t* spans, int cap_pairs, int32_t* n_pairs);
/* leaf_tok_constrain for nltk.word_tokenize (tokenizer_kind 1) with the Punkt tables: the sentence spans of captions AND candidates
 * are computed natively, so only non-ASCII text / control characters get fallback = 1. */
int leaf_tok_constrain_punkt(leaf_dict_t d, leaf_punkt_t punkt, const char* const* sentences, const int32_t* sent_len, int B,
                             const int32_t* z, const int32_t* c, int rho, uint8_t* valid, uint8_t* fallback, int n_threads);
/* number of distinct dictionary words of ONE text (utils_attacks.py:135,139), for the strings leaf_tok_constrain* declines: with the
 * sentence spans of that text from the caller's Punkt (n_ranges pairs; 0 = the text must not depend on sentence boundaries) the count is
 * native too -- one Punkt call instead of a whole nltk.word_tokenize.  Returns 0; 2 = declined (non-ASCII, or spans missing). */
int leaf_tok_count_words(leaf_dict_t d, int tokenizer_kind, const char* text, int len, const int32_t* ranges, int n_ranges,
                         int32_t* count);
/* test / debug hook: the word tokens of `text` under tokenizer_kind (0 = the regex stand-in of leaf_amd/attacks.py, 1 =
 * nltk.word_tokenize restated: nltk/tokenize/destructive.py's substitution pipeline), '\n'-joined into out[0, cap).
 * Returns 0; 2 when the text is declined (kind 1: non-ASCII, or a lone '.' ends a chunk inside the text, where the result would
 * depend on nltk's trained Punkt sentence splitter); 1 on bad arguments. */
int leaf_tok_word_tokens(int tokenizer_kind, const char* text, int len, char* out, int cap, int* out_len);
/* dup_of[b * rho + r] = first r' <= r with tokens[b, r', :] == tokens[b, r, :] (SURVEY 8f-2 dedupe: src/open_clip/tokenizer.py:83-85,139
 * lower-cases and collapses whitespace, so distinct edits can tokenise identically; first index wins as in torch.argmax,
 * utils_attacks.py:348,386).  tokens int32 [B, rho, ctx] on the host. */
int leaf_tok_duplicate_map(const int32_t* tokens, int B, int rho, int ctx, int32_t* dup_of, int n_threads);

/* ---- per-launch GEMM timing (bench.py roofline): between begin/end every GEMM launch is bracketed by HIP events on
 * its stream; end() sums duration / algorithmic FLOPs / algorithmic bytes / launches per key = kernel_family*16 + operand_dtype*8 + epilogue id
 * (family 0 = gemm_nt_kernel, 1 = gemm_nt256_kernel, 4 = gemm_nt256_half_kernel, 6 = gemm_nt64_ring_kernel; keys < 128). */
int leaf_prof_begin(void);
int leaf_prof_pause(int paused);   /* suspend / resume the recording (per-launch events on a sample of the timed steps only) */
int leaf_prof_end(double* ms, double* flops, double* bytes /* algorithmic, may be NULL */, int64_t* count, int n_keys);
/* same records grouped by (key, N, K): info[4 i] = {key, N, K, launches}, rows[i] = sum of M; *n_out groups written */
int leaf_prof_end_shapes(double* ms, double* flops, double* bytes, int64_t* rows, int32_t* info, int max_groups, int* n_out);

/* ---- single-kernel hooks (used by the parity tests to check each HIP kernel against the oracle) ---- */
/* C[M,N] = epilogue(A[M,K] * B[N,K]^T): epi 0 store16(+bias), 1 act16(+bias, aux = pre-activation), 2 fp32 += ,
 * 3 fp32 = beta*C + acc, 4 store16(acc * act'(aux)).  A, B 16-bit of `dtype`, contiguous. */
int leaf_op_gemm(int dtype, int epi, const void* A, const void* B, void* C, const float* bias, void* aux, int M, int N,
                 int K, int act, float beta, int aux_f16, leaf_stream_t s);
/* same with explicit row strides (elements) */
int leaf_op_gemm_ld(int dtype, int epi, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                    void* aux, int M, int N, int K, int act, float beta, int aux_f16, leaf_stream_t s);
/* C32 += [A | A] B^T + bias: A [M, Ka] stored once and re-read along K = 2 Ka by the half-stage ring kernel (B [N, 2 Ka] = e.g. the
 * [hi | lo] split of a weight: the out-projection of a split block); an error for launches that kernel does not take */
int leaf_op_gemm_awrap(int dtype, const void* A, const void* B, float* C, const float* bias, int M, int N, int Ka, leaf_stream_t s);
/* LayerNorm folded into the GEMMs of the forward-only passes (leaf_amd/csrc/lnfold.h; replaces the LayerNorm launches between
 * the GEMMs of transformer.py:254-265).  Producer: C32[M,N] += A B^T + bias, x16 = 16-bit copy of the result, stat = float2
 * [N/64][M] (sum, M2 = sum (x - group mean)^2) of each row's 64-column groups.  Finalize: rowstat[m] = (mean, rstd) merged from
 * stat [ngroups][ld] (Chan).  Consumer: C16 = [act](rstd[m] (A Bp^T - mean[m] s[n]) + c[n]), Bp = 16-bit(gamma * W), s[n] =
 * sum_k Bp[n,k], c[n] = beta . W[n,:] + bias[n]; act: -1 none, 0 GELU, 1 QuickGELU. */
int leaf_op_gemm_resid_ln(int dtype, const void* A, const void* B, float* C, const float* bias, void* x16, void* stat, int M, int N,
                          int K, leaf_stream_t s);
/* The residual stream of the LN-folded forward-only passes in 16 + 8 bits (option 'compact_resid', default on): a residual value x is
 * held as x16 = 16-bit(x) -- the copy the next GEMM multiplies anyway -- and lo8 = the remainder x - x16 as an e4m3 byte, block-scaled (four consecutive values share
 * the power of two of their largest |x16|; v_cvt_scalef32_pk_fp8_f32 / _f32_fp8), i.e. to 2^-16 of that largest value (fp16) in 3 bytes instead of an fp32 row beside the 16-bit copy (6 bytes).  leaf_op_gemm_resid_ln8: the producer above
 * on that format, (x16, lo8) [M,N] updated IN PLACE, stat as above.  leaf_op_resid_pack / _unpack: the format itself, element-wise
 * (n % 4 == 0).  The reference's residual stream is fp32 (transformer.py:254-265); the difference is far inside the 16-bit operand
 * rounding of every GEMM (tests/test_gpu_kernels.py, test_compact_residual_*). */
int leaf_op_gemm_resid_ln8(int dtype, const void* A, const void* B, void* lo8, const float* bias, void* x16, void* stat, int M, int N,
                           int K, leaf_stream_t s);
int leaf_op_resid_pack(int dtype, const float* x, void* x16, void* lo8, size_t n, leaf_stream_t s);
int leaf_op_resid_unpack(int dtype, const void* x16, const void* lo8, float* x, size_t n, leaf_stream_t s);
int leaf_op_ln_finalize(const void* stat, int ld, int rows, int ngroups, float eps, void* rowstat, leaf_stream_t s);
int leaf_op_gemm_lnfold(int dtype, int act, const void* A, const void* Bp, void* C16, const float* c_vec, const float* s_vec,
                        const void* rowstat, int M, int N, int K, leaf_stream_t s);
/* LN-folded QKV projection + causal attention in ONE launch (leaf_amd/csrc/qkv_attn.hip; nn.MultiheadAttention's in_proj ->
 * softmax(q k^T / 8 + causal mask) v, src/open_clip/transformer.py:225,239-252): out[rows, width] (eot_pos: [n_seq, width]) from the
 * 16-bit residual copy x16 [rows, width], Wp = 16-bit(gamma * in_proj_weight) [3 width, width] with its c / s vectors and the rows'
 * (mean, rstd).  lens (HOST) = rows per sequence, cu (device) their exclusive prefix sum; prefix / base_cu / kv (device, or all
 * NULL) = cached-prefix mode as in leaf_score_candidates_prefix with candidates grouped `group` per caption; tile_seq = device
 * scratch [2 (n_seq + 1)].  Kernel hook for tools/qkv_attn_bench.py; the scoring passes call the same launch internally. */
int leaf_op_qkv_attn(int dtype, const void* x16, const void* Wp, const float* c_vec, const float* s_vec, const void* rowstat,
                     void* out, const void* kv, const int32_t* lens, const int32_t* cu, const int32_t* prefix,
                     const int32_t* base_cu, const int32_t* eot_pos, int32_t* tile_seq, int n_seq, int rows, int group,
                     int ctx, int heads, int width, leaf_stream_t s);
int leaf_op_attention_fwd(const void* qkv, void* out, int n_seq, int ctx, int heads, int width, int dtype,
                          leaf_stream_t s);
int leaf_op_layernorm(const float* x, const float* g, const float* b, float eps, void* out16, int rows, int width,
                      int dtype, leaf_stream_t s);
int leaf_op_attention_bwd(const void* qkv, int qkv_dtype, const void* dout_bf16, void* dqkv_bf16, int n_seq, int ctx,
                          int heads, int width, leaf_stream_t s);
/* attention backward with an explicit gradient dtype (LEAF_DTYPE_*) for dout / dqkv */
int leaf_op_attention_bwd_t(const void* qkv, int qkv_dtype, const void* dout16, void* dqkv16, int g_dtype, int n_seq,
                            int ctx, int heads, int width, leaf_stream_t s);
/* one problem of the grouped weight-gradient kernel: dW[Nw,Kw] += alpha dY^T X, db[Nw] += alpha colsum(dY)
 * (dY [rows,Nw] in g_dtype, X [rows,Kw] in x_dtype, alpha a device scalar or NULL = 1; Nw, Kw multiples of 128) */
int leaf_op_wgrad(const void* dY, const void* X, float* dW, float* db, int rows, int Nw, int Kw, int x_dtype,
                  int g_dtype, const float* alpha_dev, leaf_stream_t s);
/* LayerNorm backward of `rows` rows (autograd of F.layer_norm, src/open_clip/transformer.py:15-30): dx_inout[rows,d] +=
 * d LN / d x, dx16 (optional) = its 16-bit copy (g_dtype), dg[d] += sum_r dy * xhat / S, db[d] += sum_r dy / S with
 * gscale = {S, 1/S} on the device (the loss scale of the fp16 gradient path); dg = db = NULL: input gradient only.
 * ws: leaf_op_layernorm_bwd_ws_bytes(rows, d) bytes of device scratch for the per-workgroup partial sums (needed with dg) */
size_t leaf_op_layernorm_bwd_ws_bytes(int rows, int d);
int leaf_op_layernorm_bwd(const float* dy, const float* x, const float* g, float eps, float* dx_inout, void* dx16,
                          int g_dtype, const float* gscale, float* dg, float* db, int rows, int d, void* ws, size_t ws_bytes,
                          leaf_stream_t s);
/* out[M,D] = [normalize](LayerNorm(xg[M,d]; g, b) @ proj[d,D]) in fp32 on the matrix cores; xn_scratch = fp32 [M,d] */
int leaf_op_project_rows(const float* xg, const float* g, const float* b, float eps, const float* proj,
                         float* xn_scratch, float* out, int M, int d, int D, int normalize, leaf_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
