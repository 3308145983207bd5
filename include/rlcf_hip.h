/* rlcf_hip.h — C ABI of librlcf_hip.so: the MI355X (gfx950) implementation of the
 * RLCF per-sample test-time-adaptation hot path.
 *
 * The reference (mzhaoshuai/RLCF, TPT/) is pure Python on torch ops and has no FFI;
 * this header is the boundary a maintainer would bind from Python (ctypes, see
 * INTEGRATION.md).  Every entry point cites the reference call site it replaces
 * (paths relative to the reference tree).
 *
 * Conventions
 *   - all `const float*` / `float*` / `int32_t*` arguments are DEVICE pointers into
 *     caller-owned (torch) memory, contiguous row-major, unless marked HOST;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, negative rlcf_status on error (rlcf_last_error() has text);
 *   - an engine is not thread-safe; every scratch buffer of the engine calls belongs to the engine (one engine = one device, one
 *     stream at a time; two engines never share memory).  Workspaces whose size depends on the call (views x tokens, class bank)
 *     are sized by the FIRST call that needs them and only ever grow: a steady stream of equal-sized samples allocates nothing
 *     after its first call.  The growth path is hipMalloc (+ hipFree of the smaller buffer: both synchronise the device) and a
 *     fill enqueued on the CALLER's stream, so it is ordered against the kernels that follow on any kind of stream — blocking,
 *     non-blocking or the NULL stream (the whole GPU suite runs in both modes: tests/conftest.py, RLCF_TEST_STREAM).  The stateless
 *     op-level calls that need scratch (rlcf_gemm_nt in split-f16 mode, rlcf_reward_loss*) take it from the stream-ordered
 *     allocator of `stream` (hipMallocAsync / hipFreeAsync);
 *   - an engine owns one side stream: rlcf_tta_sample (one test image per call) forks the reward models' tower pass onto it behind
 *     an event recorded on `stream` and joins it back with a second event before the loss kernel, so the call stays ordered on
 *     `stream` as a whole (work enqueued on `stream` afterwards sees all its results).  No host thread, no host synchronisation.
 */
#ifndef RLCF_HIP_H
#define RLCF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* rlcf_stream;            /* hipStream_t */
typedef struct rlcf_engine rlcf_engine;

enum rlcf_status { RLCF_OK = 0, RLCF_ERR_ARG = -1, RLCF_ERR_HIP = -2, RLCF_ERR_STATE = -3, RLCF_ERR_NOMEM = -4 };
enum rlcf_precision {
    RLCF_PREC_F32 = 0,   /* f32 storage, f32-input MFMA (exact f32 FMA chains, 157 TF pipe)                       */
    RLCF_PREC_F16 = 1,   /* PERFORMANCE mode, NOT parity-grade: the forward tower pipeline in plain f16 (one MFMA per product, f32
                          * accumulate; LayerNorm / softmax / losses / AdamW in f32) = the arithmetic of the reference's fp16-autocast
                          * GPU path (TPT/tpt_cls_rl.py:52).  Logits deviate from the f32 reference by ~1e-2 at logit scale 100
                          * (measured per run by bench.py and tests: max |dlogit|, top-1 agreement); engine calls only        */
    RLCF_PREC_F16X3 = 2  /* split-f16: each f32 operand = hi+lo f16, 3 f16 MFMAs per product, f32-grade results (the default: meets
                          * the 1e-3 logit contract)                                                                         */
};
enum rlcf_epilogue {     /* GEMM epilogues (TPT/clip/model.py:166-168,177-181,190-191) */
    RLCF_EPI_NONE = 0,
    RLCF_EPI_QUICKGELU = 1,      /* y = v*sigmoid(1.702 v)                                   */
    RLCF_EPI_QUICKGELU_BWD = 2,  /* y = v * d/df[f*sigmoid(1.702 f)] with f = aux            */
    RLCF_EPI_RELU = 3            /* y = max(v (+ residual), 0): conv+bn(+identity)+relu of ModifiedResNet, model.py:24-56 */
};
enum rlcf_text_mode {
    RLCF_TEXT_DENSE = 0,   /* reference graph: every class runs all context_length positions */
    RLCF_TEXT_PACKED = 1,  /* rows after EOT dropped (exact under the causal mask)          */
    RLCF_TEXT_SHARED = 2   /* PACKED + the class-independent [SOT|ctx] rows computed once    */
};
enum rlcf_which { RLCF_STUDENT = 0, RLCF_REWARD = 1 /* reward slot m (0-based) is RLCF_REWARD + m */ };

const char* rlcf_last_error(void);
int rlcf_version(void);

/* CLIP geometry: constructor arguments of the reference `CLIP` class (TPT/clip/model.py:244-257).  The reference passes
 * `vision_layers` as an int for a VisionTransformer and as a 4-tuple of Bottleneck counts for a ModifiedResNet (:262-270): here
 * vision_stages[0] > 0 selects the ModifiedResNet (inference only: reward models, frozen student image encoder of the prompt
 * path) and vision_layers / vision_patch_size are then ignored. */
typedef struct {
    int embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size;
    int context_length, vocab_size, text_width, text_heads, text_layers;
    int vision_stages[4];
} rlcf_clip_cfg;

/* One attention sequence over a packed token matrix: queries are rows
 * [q_start, q_start+q_len); keys are rows [pre_start, pre_start+pre_len) followed by the
 * query rows themselves.  Causal: query i sees all prefix keys and own keys <= i. */
typedef struct { int q_start, q_len, pre_start, pre_len; } rlcf_seq;

/* ------------------------------------------------------------------ op level ----
 * Stateless kernels, exposed for parity tests and for autograd glue. */

/* C[M,N] = epi(alpha * A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N]).  Replaces the
 * nn.Linear / in_proj / out_proj / `@ proj` call sites TPT/clip/model.py:175-191,235-238,
 * custom_clip.py:71,332-333.  K % 16 == 0.  bias/residual/aux may be NULL. */
int rlcf_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias,
                 const float* residual, int ldr, const float* aux, int ldaux,
                 float* C, int ldc, int M, int N, int K, float alpha, int epilogue, int precision,
                 rlcf_stream stream);

/* Split-f16 operand pairs (RLCF_PREC_F16X3): x = hi + lo with hi = f16(x), lo = f16(x-hi) (operands should be O(1):
 * below |x| = 2^-3 the absolute error floor is 2^-25; the engine pre-scales weights and gradients by exact powers of two);
 * n % 8 == 0.  rlcf_gemm_f16x3 is rlcf_gemm_nt on pre-split operands (3 f16 MFMAs per product,
 * f32 accumulate); output f32 (C) and/or a split pair (Chi, Clo).  K % 32 == 0. */
int rlcf_split_f16x2(const float* x, void* hi, void* lo, int64_t n, rlcf_stream stream);
int rlcf_gemm_f16x3(const void* Ahi, const void* Alo, int lda, const void* Whi, const void* Wlo, int ldw, const float* bias,
                    const float* residual, int ldr, const float* aux, int ldaux, float* C, int ldc, void* Chi, void* Clo,
                    int ldch, int M, int N, int K, float alpha, int epilogue, rlcf_stream stream);

/* Single-pass f16 GEMM (RLCF_PREC_F16, the labelled performance mode — NOT parity-grade): the arithmetic of the reference's own GPU runs,
 * every nn.Linear of TPT/clip/model.py:171-192 under torch.cuda.amp.autocast() (TPT/tpt_cls_rl.py:52,182,261): f16 operands, one f16 MFMA
 * per product, f32 accumulate.  A [M,K] f16 (lda halves), W [N,K] f16 (ldw halves), bias f32 [N] or NULL; outputs: C f32 [M,N] (+ residual
 * f32 [M,N], may alias C) and / or C16 f16 [M,N] (epilogue RLCF_EPI_NONE | RLCF_EPI_QUICKGELU applied before the f16 rounding).
 * K % 64 == 0, N % 4 == 0, lda / ldw % 8 == 0.  Grids of >= 256 tiles of 256x128 run the dedicated 256x256 eight-phase kernel
 * (rlcf_amd/csrc/gemm_f16.hip), smaller ones the 128x128 / 256x128 kernels. */
int rlcf_gemm_f16(const void* A, int lda, const void* W, int ldw, const float* bias, const float* residual, int ldr, float* C, int ldc,
                  void* C16, int ldch, int M, int N, int K, float alpha, int epilogue, rlcf_stream stream);

/* The same product with the preceding LayerNorm FOLDED in (RLCF_PREC_F16 image towers; rlcf_amd/csrc/gemm_f16.hip, MODE 1 / 2).  Under
 * autocast the reference's residual stream is an fp16 tensor — its LayerNorm computes in fp32 and casts back to the input type, and
 * x = x + attention(ln_1(x)) adds fp16 tensors (TPT/clip/model.py:157-163,187-192 under TPT/tpt_cls_rl.py:52) — so
 *   mode 1 (ln_1 -> in_proj, ln_2 -> c_fc):  out16[M,N] = epi(rstd_r (alpha x16.Wg^T - mu_r s) + bias),  x16 = the f16 residual rows, Wg = the
 *          f16 copy of W diag(gamma), s [N] = its row sums (times alpha), bias = W beta + b, ln_mr [M][2] = (mean, rstd) of x16's rows;
 *   mode 2 (out_proj, c_proj + residual add): out16[M,N] (the residual stream, IN PLACE) = f16(out16 + alpha A.W^T + bias), and the partial
 *          (sum, sum of squares) of every updated row over each 64-column slice goes to ln_part [(N/64)][M][2];
 * rlcf_ln_stats_final adds the N/64 partials of a row in order -> ln_mr (eps 1e-5, biased variance: nn.LayerNorm);  rlcf_resid16_init
 * rounds an f32 stream to f16 rows and leaves their (mean, rstd).  N % 256 == 0, K % 128 == 0, K >= 256. */
int rlcf_gemm_f16_ln(const void* A, int lda, const void* W, int ldw, const float* bias, void* out16, int ldo, int M, int N, int K, float alpha,
                     int epilogue, int mode, const float* ln_mr, const float* ln_s, float* ln_part, rlcf_stream stream);
int rlcf_ln_stats_final(const float* ln_part, int parts, int rows, int width, float* ln_mr, rlcf_stream stream);
int rlcf_resid16_init(const float* x, void* x16, float* ln_mr, int rows, int width, rlcf_stream stream);

/* 3x3 convolution, stride 1, padding 1, NHWC, as an IMPLICIT GEMM on the f16 matrix cores with split-f16 operands (no patch matrix: the
 * 256x256 GEMM kernel's DMA reads every tap's K tile straight from the activation's operand pairs): the `conv2` of a Bottleneck
 * (TPT/clip/model.py:20,44) with its BatchNorm folded, as the engine's ResNet towers run it.  x [n,H,W,Cin], w [Cout,3,3,Cin] ((ky,kx,c)
 * order), bias [Cout] or NULL, residual [n,H,W,Cout] or NULL (added before the ReLU), epilogue RLCF_EPI_NONE | RLCF_EPI_RELU -> y
 * [n,H,W,Cout].  Cin % 32 == 0, Cout % 4 == 0, ceil(n H W / 256) * ceil(Cout / 256) >= 192 (smaller grids: the engine keeps the
 * patch-matrix form); operands should be O(1) (no pre-scale is applied here). */
int rlcf_conv3x3_nhwc_f16x3(const float* x, const float* w, const float* bias, const float* residual, float* y, int n, int H, int W,
                            int Cin, int Cout, int epilogue, rlcf_stream stream);

/* Row LayerNorm, fp32, eps 1e-5, biased variance (TPT/clip/model.py:157-163). */
int rlcf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       int rows, int width, rlcf_stream stream);
/* dx (and, if dgamma/dbeta non-NULL, ACCUMULATED parameter grads) of the above. */
int rlcf_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx,
                       float* dgamma, float* dbeta, int rows, int width, rlcf_stream stream);

/* Multi-head attention core of nn.MultiheadAttention (TPT/clip/model.py:175,185-187):
 * qkv[T,3W] packed [q|k|v], head_dim 64, scale 1/8, softmax over keys, out[T,W].
 * seqs: DEVICE array of n_seq descriptors; max_q_len bounds q_len; lse[T,H] optional. */
int rlcf_attention_fwd(const float* qkv, const rlcf_seq* seqs, int n_seq, int max_q_len,
                       int width, int causal, float* out, float* lse, int precision, rlcf_stream stream);
/* The same forward on PRODUCER-EMITTED operands (attention_pair.hip; what the engine's image towers run since round 3): qkv_pairs is
 * the in_proj output [T,3W] already written as interleaved f16 pairs (RLCF_PREC_F16X3: per row, every block of 32 columns = its 32 hi
 * halves followed by its 32 lo halves, row = 12W bytes — rlcf_split_pairs / the GEMM's pair epilogue) or as a plain f16 matrix
 * (RLCF_PREC_F16).  K / V stages move by LDS-DMA, V^T comes out of ds_read_b64_tr_b16: no f32 re-read, no in-kernel split.
 * Non-causal sequences (optional prefix).  out (f32 [T,W]) and / or out_pairs ([T,W] in the operand layout of the precision);
 * lse[T,H] optional.  TPT/clip/model.py:175,185-187. */
int rlcf_attention_fwd_pairs(const void* qkv_pairs, const rlcf_seq* seqs, int n_seq, int max_q_len, int width, float* out,
                             void* out_pairs, float* lse, int precision, rlcf_stream stream);
/* measurement switch of the call above (tools/attn_ab.py): variant = the build mask of attention_pair.hip (1 = the shipped arithmetic,
 * 0 = eager softmax rescale; 8 / 32 / 40 / 64 = ablation builds of the 8-wave launch that give WRONG numbers by design; 16 = s_memtime
 * trace).  The first argument is reserved (0). */
int rlcf_attention_debug(int reserved, int variant);
/* Few-row products of the one-image path (M <= 256: the sparse text forward / backward over the ~239 rows of the sampled classes,
 * TPT/clip/custom_clip.py:53-73 under TPT/tpt_cls_rl.py:60-75): C[M,N] = epi(alpha A W^T + bias) (+ residual) with A f32 [M,K]
 * (split into f16 pairs inside the kernel) and W as interleaved operand pairs [N,2K] (rlcf_split_pairs); K % 32 == 0, N % 4 == 0.
 * amax_in: device max|A| (the power-of-two operand scale is derived from it) or NULL; local_amax != 0 with amax_in == NULL: every
 * workgroup scales its rows by their own max, found in the kernel (gradients nobody measured); both 0 / NULL: scale 1. */
int rlcf_gemm_skinny(const float* A, int lda, const void* W_pairs, const float* bias, const float* residual, int ldr, const float* aux,
                     int ldaux, float* C, int ldc, int M, int N, int K, float alpha, int epilogue, const float* amax_in, int local_amax,
                     rlcf_stream stream);
/* x[n] f32 -> the operand layout above: interleaved (hi | lo) pairs per 32-element block (RLCF_PREC_F16X3, 4n bytes) or plain f16
 * (RLCF_PREC_F16, 2n bytes); n % 32 == 0 and rows that are multiples of 32 columns keep their row structure. */
int rlcf_split_pairs(const float* x, void* pairs, int64_t n, int precision, rlcf_stream stream);
/* Backward (dX only): dqkv[T,3W] from dout[T,W]; dqkv must be zero-filled by the caller
 * (keys accumulate across query blocks / sequences).  max_keys (prefix + queries) <= 320. */
int rlcf_attention_bwd(const float* qkv, const float* dout, const rlcf_seq* seqs, int n_seq, int max_keys,
                       int width, int causal, float* dqkv, rlcf_stream stream);
/* The same backward on the f32 matrix cores, any sequence length: needs the forward's output out[T,W] and log-sum-exp
 * lse[T,H] (rlcf_attention_fwd with precision F32).  Used by the LayerNorm-tuning backward through the image tower. */
int rlcf_attention_bwd_flash(const float* qkv, const float* out, const float* lse, const float* dout, const rlcf_seq* seqs,
                             int n_seq, int max_q_len, int width, int causal, float* dqkv, rlcf_stream stream);

/* Per-row entropy H = -sum softmax*log_softmax and the int(N*top) lowest-entropy rows in
 * ascending order: select_confident_samples, TPT/tpt_cls_rl.py:32-35.  idx[n_sel]. */
/* the same with the arithmetic chosen: RLCF_PREC_F32 = rlcf_attention_bwd_flash; RLCF_PREC_F16X3 = the split-f16 kernel the engine's
 * backward passes use in that mode (attention_bwd_x3.hip: three f16 MFMAs per product, dout lifted by a power of two found on the
 * device).  The op-level call reads the sequence descriptors back once to find the extent of dout (engine calls do not). */
int rlcf_attention_bwd_flash_prec(const float* qkv, const float* out, const float* lse, const float* dout, const rlcf_seq* seqs, int n_seq,
                                  int max_q_len, int width, int causal, float* dqkv, int precision, rlcf_stream stream);
/* (ABI version 12) Stand-alone conveniences of the harness: avg_entropy (TPT/tpt_cls_rl.py:38-44: entropy of the views' mean softmax; inside a
 * tuning step the regulariser and its gradient come from rlcf_reward_loss) and accuracy for topk = (1, 5) (TPT/utils/tools.py:84-98):
 * out[0] / out[1] = percentage of rows whose target is in the top 1 / top 5; target int64 [B]; top5_scratch int32 [B, 5]. */
int rlcf_avg_entropy(const float* logits, int n, int C, float* out, rlcf_stream stream);
int rlcf_accuracy(const float* logits, const int64_t* target, int B, int C, int32_t* top5_scratch, float* out, rlcf_stream stream);
int rlcf_entropy_select(const float* logits, int n, int C, int n_sel, float* entropy,
                        int32_t* idx, rlcf_stream stream);

/* Flags of rlcf_reward_loss (TPT/params.py:55-59,65-66). */
#define RLCF_MAX_REWARDS 4
enum { RLCF_F_REWARD_PROCESS = 1, RLCF_F_AMPLIFY = 2, RLCF_F_PROCESS_BATCH = 4, RLCF_F_MIN_ENTROPY = 8,
       RLCF_F_NO_SELECTION = 16 /* image-encoder tuning calls: rows taken in order, no confidence selection (set by rlcf_tta_retrieval_image) */ };
/* The loss section of test_time_tuning (TPT/tpt_cls_rl.py:63-74) with
 * CLIPRewards.CLIPScore / rewards_post_process (TPT/clip_reward.py:111-128,152-165):
 * rows = logits[sel[i]] (sel NULL: rows = logits[i]); top-K classes per row; CLIPScore
 * against class_feat[C,Dr] and reward_img[n_sel,Dr]; baseline; loss = mean(r*CE)
 * (+ w*avg_entropy); dlogits[n_sel,C] = dloss/drows (dense).  All outputs optional but dlogits.  1 <= K <= 32.
 * The retrieval policy of the reference (retrieval/clip_ret_policy.py:76-137: tune_image / tune_text) is the same arithmetic over a
 * bank of 5 000 images or 25 000 captions with K = 12 / 20 (retrieval/scripts/tta_coco_ret.sh:19-20): rows = the query's logits
 * over the bank, class_feat = the reward model's bank features, reward_img = its query features. */
int rlcf_reward_loss(const float* logits, int ld_logits, const int32_t* sel, int n_sel, int C, int K,
                     const float* class_feat, const float* reward_img, int Dr,
                     float clipscore_weight, int flags, float min_entropy_w,
                     int32_t* topk_idx, float* clip_score, float* rewards, float* loss,
                     float* dlogits, rlcf_stream stream);

/* The same with an ENSEMBLE of reward models (CLIPRewardsMultiple.CLIPScore, TPT/clip_reward.py:228-257): per model m the
 * clamped score max(w*<class_feats[m][idx], reward_imgs[m][i]>, 0); final score = sum_m mix[m]*score_m (weighted_scores) or
 * (sum_m score_m)/n_models (mean != 0).  class_feats / reward_imgs / Dr / mix are HOST arrays of n_models (<= RLCF_MAX_REWARDS)
 * entries; the pointers in them are device pointers. */
int rlcf_reward_loss_ensemble(const float* logits, int ld_logits, const int32_t* sel, int n_sel, int C, int K, int n_models,
                              const float* const* class_feats, const float* const* reward_imgs, const int* Dr, const float* mix,
                              int mean, float clipscore_weight, int flags, float min_entropy_w,
                              int32_t* topk_idx, float* clip_score, float* rewards, float* loss, float* dlogits,
                              rlcf_stream stream);

/* torch.optim.AdamW single step (amsgrad off), TPT/tpt_cls_rl.py:78,120. step is 1-based. */
int rlcf_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, int step,
                    float lr, float beta1, float beta2, float eps, float weight_decay,
                    rlcf_stream stream);

/* ------------------------------------------------------------------ views ------
 * Device-side replacement of the reference's per-sample CPU view pipeline (TPT/data/datautils.py:76-128 AugMixAugmenter with an
 * empty aug_list; transforms of TPT/tpt_cls_rl.py:132-150): from ONE decoded uint8 RGB image (HWC, device) write
 * views[1 + n_crops, 3, res, res] float32: view 0 = Resize(res, bicubic) + CenterCrop(res); view 1+i = resized_crop(crops[i],
 * bilinear) with optional horizontal flip; all followed by ToTensor and Normalize(mean, std).  Bit-exact with Pillow's 8-bit
 * resampler.  crops, mean3, std3 are HOST arrays; the random boxes / flips are drawn by the caller (RandomResizedCrop.get_params,
 * RandomHorizontalFlip).  scratch: device, >= rlcf_make_views_scratch_bytes(H, n_crops, res). */
typedef struct { int top, left, h, w, flip; } rlcf_crop;
size_t rlcf_make_views_scratch_bytes(int H, int n_crops, int res);
int rlcf_make_views(const uint8_t* image, int H, int W, const rlcf_crop* crops, int n_crops, int res, const float* mean3,
                    const float* std3, float* views, void* scratch, size_t scratch_bytes, rlcf_stream stream);

/* The same with the AugMix op chains the reference switches on for the fine-grained sets (AugMixAugmenter(augmix=True),
 * TPT/tpt_cls_rl.py:149-150, scripts/rlcf-prompt-fine.sh): `augmix` of TPT/data/datautils.py:94-110 with aug_list =
 * TPT/data/augmix_ops.py:144-147.  View 1+v becomes  m[v] * pre(x) + (1 - m[v]) * sum_i w[v][i] * pre(chain_{v,i}(x))  where x is the
 * 8-bit resized crop, pre = ToTensor + Normalize, and chain_{v,i} applies ops[(v*3 + i)*3 + 0..2] in turn (op = -1: none).
 * op ids follow the reference's list: 0 autocontrast, 1 equalize, 2 posterize (ip = bits kept), 3 rotate, 4 solarize (ip =
 * threshold), 5 shear_x, 6 shear_y, 7 translate_x, 8 translate_y; ops 3 and 5-8 carry the six coefficients Image.transform(AFFINE)
 * receives in c (for rotate: the matrix Image.rotate builds).  Bit-exact with Pillow (look-up tables; bilinear affine resampling
 * in double precision, truncated).  ops, w [n_crops,3], m [n_crops] are HOST arrays drawn by the caller with the reference's
 * numpy calls; they are copied (through pinned staging) before the call returns, which does not wait for the device. */
typedef struct { int op; int ip; double c[6]; } rlcf_augmix_op;
size_t rlcf_make_views_augmix_scratch_bytes(int H, int n_crops, int res);
int rlcf_make_views_augmix(const uint8_t* image, int H, int W, const rlcf_crop* crops, int n_crops, int res, const float* mean3,
                           const float* std3, const rlcf_augmix_op* ops, const float* w, const float* m, float* views, void* scratch,
                           size_t scratch_bytes, rlcf_stream stream);

/* The `hard_aug` pre-augmentation (get_preaugment(hard_aug=True), TPT/data/datautils.py:77-87; --hard_aug 1 of TPT/tune_cls_tpt.py:115 /
 * tune_cls_kd.py:122): between the resized crop (RandomResizedCrop(224, scale=(0.2, 1))) and the flip, view 1+v goes through
 * RandomApply([ColorJitter(0.4, 0.4, 0.2, 0.1)], 0.5), RandomGrayscale(0.2) and RandomApply([GaussianBlur(3, (0.1, 2))], 0.1).
 * hard[v] carries the draws of view 1+v: order = ColorJitter.get_params' randperm(4) (0 brightness, 1 contrast, 2 saturation, 3 hue;
 * order[0] = -1: ColorJitter skipped), b / c / s = the three factors, hue = uint8(hue_factor * 255) (0..255, added to H modulo 256),
 * gray = RandomGrayscale's coin, blur / k = GaussianBlur applied / its float32 3x3 kernel, row-major (torchvision
 * _get_gaussian_kernel2d).  Bit-exact with Pillow's ImageEnhance / HSV / L conversions; the blur is nine float32 fused multiply-adds
 * in row-major order, rounded half to even (what torch's CPU conv2d computes).  ops / w / m: the AugMix plan as in
 * rlcf_make_views_augmix, or all NULL (no AugMix).  hard, ops, w, m are HOST arrays, copied before the call returns. */
typedef struct { int order[4]; float b, c, s; int hue; int gray; int blur; float k[9]; } rlcf_hard_aug;
size_t rlcf_make_views_hard_scratch_bytes(int H, int n_crops, int res);
int rlcf_make_views_hard(const uint8_t* image, int H, int W, const rlcf_crop* crops, int n_crops, int res, const float* mean3,
                         const float* std3, const rlcf_hard_aug* hard, const rlcf_augmix_op* ops, const float* w, const float* m,
                         float* views, void* scratch, size_t scratch_bytes, rlcf_stream stream);

/* ------------------------------------------------------------------ engine ------
 * Owns device copies of the weights (plus derived layouts) and all workspace. */
rlcf_engine* rlcf_engine_create(const rlcf_clip_cfg* student, const rlcf_clip_cfg* reward /*NULL: none*/,
                                int max_views, int max_classes, int precision);
/* Engine with n_rewards (0..RLCF_MAX_REWARDS) reward models: get_reward_model(args.multiple_reward_models=1) builds
 * CLIPRewardsMultiple over a list of CLIP archs (TPT/clip_reward.py:29-34,180-225).  Slot m is addressed as which = RLCF_REWARD+m. */
rlcf_engine* rlcf_engine_create_ensemble(const rlcf_clip_cfg* student, const rlcf_clip_cfg* rewards, int n_rewards,
                                         int max_views, int max_classes, int precision);
/* Ensemble scoring rule: mix[n] = the reference's round(w/sum(w),2) weights (clip_reward.py:206,250-253) or, with mean != 0,
 * torch.mean over models (:255).  n must equal the number of reward slots.  Default: slot 0 alone, weight 1. */
int rlcf_engine_set_reward_mix(rlcf_engine*, const float* mix, int n, int mean);
void rlcf_engine_destroy(rlcf_engine*);
/* Copy one OpenAI-layout state-dict tensor (TPT/clip/model.py:399-436) into the engine.  The call takes no stream: dev_ptr must be complete
 * on the device when it is made (it is read on the NULL stream); the copy is complete when the call returns. */
int rlcf_engine_load_weight(rlcf_engine*, int which, const char* key, const float* dev_ptr, int64_t numel);
/* After all weights: checks completeness, builds transposed / low-precision copies. */
int rlcf_engine_finalize(rlcf_engine*, rlcf_stream stream);

/* PromptLearner.__init__/reset_classnames (TPT/clip/custom_clip.py:77-196) +
 * BaseRewards.set_class_features (TPT/clip_reward.py:55-57): tokens HOST int32 [C,context_length]
 * as produced by clip.tokenize; builds the packed text layout, gathers token embeddings and
 * caches the reward model's class features.  ctx_init DEVICE [n_ctx,W] becomes ctx_init_state. */
int rlcf_engine_set_class_bank(rlcf_engine*, const int32_t* tokens_host, int C, int n_ctx,
                               const float* ctx_init, int text_mode, rlcf_stream stream);

/* The same for a PromptLearner whose class tokens do not sit at the END of the prompt (class_token_position 'front' / 'middle', or
 * '[CLS]' inside ctx_init: custom_clip.py:92-97,239-284): ctx_pos HOST [C, n_ctx] = position of learnable vector k in prompt c
 * ('front': 1 + name_len + k; 'middle': 1 + k for k < half, 1 + half + name_len + (k - half) after), student_tokens HOST
 * [C, context_length] = the token ids in that re-arranged order (values at the learnable positions are ignored); tokens = the prompts
 * as tokenised ("<prefix> <class>."): the reward models' bank and the EOT positions.  Rows common to all classes ('middle': SOS + the
 * first half of the context) are still computed once in RLCF_TEXT_SHARED mode. */
int rlcf_engine_set_class_bank_ex(rlcf_engine*, const int32_t* tokens_host, int C, int n_ctx, const float* ctx_init, int text_mode,
                                  const int32_t* student_tokens_host, const int32_t* ctx_pos_host, rlcf_stream stream);

/* encode_image + L2 normalise (TPT/clip/model.py:223-240,340-341; custom_clip.py:327-330;
 * clip_reward.py:130-137).  images [n,3,R,R]; feats [n,D]. */
int rlcf_encode_image(rlcf_engine*, int which, const float* images, int n, float* feats, rlcf_stream stream);
/* same for images [n,3,in_res,in_res] at another resolution than the model's: bicubic, align_corners=True resample first
 * (nn.functional.interpolate at clip_reward.py:133-134). */
int rlcf_encode_image_resized(rlcf_engine*, int which, const float* images, int n, int in_res, float* feats, rlcf_stream stream);
/* ClipTestTimeTuning.get_text_features (custom_clip.py:315-323): txt [C,D] normalised. */
int rlcf_text_features(rlcf_engine*, const float* ctx, float* txt, rlcf_stream stream);
/* reward class bank of slot `which` (>= RLCF_REWARD) cached by set_class_bank: copies [C,Dr] out. */
int rlcf_reward_class_features(rlcf_engine*, int which, float* out, rlcf_stream stream);
/* logits = exp(logit_scale) * img @ txt^T (custom_clip.py:332-333). */
int rlcf_logits(rlcf_engine*, const float* img, int n, const float* txt, int C, float* logits, rlcf_stream stream);
/* d loss / d ctx given dlogits[n,C] w.r.t. logits = scale*img@txt(ctx)^T — what autograd does for
 * loss.backward() at TPT/tpt_cls_rl.py:77.  Dense over classes (any dlogits). */
int rlcf_text_backward_dense(rlcf_engine*, const float* ctx, const float* img, int n, const float* dlogits,
                             float* dctx, rlcf_stream stream);

typedef struct {                 /* flags read on the path, TPT/params.py:13-98 */
    float selection_p; int tta_steps; int sample_k;
    float lr, weight_decay, beta1, beta2, eps;
    int flags;                   /* RLCF_F_* */
    float clipscore_weight, min_entropy_w;
    int sparse_backward;         /* 1: back-propagate only the n_sel*K touched classes when exact */
    int skip_final;              /* 1: tuning steps only (test_time_tuning); the caller runs the final inference */
    const float* ctx_in;         /* DEVICE [n_ctx,W] starting prompt, NULL = ctx_init_state (model.reset()) */
    int n_sel;                   /* int(N * selection_p) as the caller's Python computes it (a double product, tpt_cls_rl.py:34);
                                    0 = derive it here from the float selection_p (can differ by one at exact boundaries) */
} rlcf_tta_args;

typedef struct {                 /* every pointer optional (NULL = not wanted); DEVICE */
    float* logits;               /* [N,C] first-step student logits          */
    float* entropy;              /* [N]                                      */
    int32_t* selected_idx;       /* [n_sel]                                  */
    int32_t* topk_idx;           /* [n_sel,K] first step                     */
    float* clip_score;           /* [n_sel*K]                                */
    float* rewards;              /* [n_sel*K]                                */
    float* loss;                 /* [1]                                      */
    float* dlogits;              /* [n_sel,C]                                */
    float* ctx_grad;             /* [n_ctx,W] first step                     */
    float* ctx_after;            /* [n_ctx,W]                                */
    float* reward_image_features;/* [n_sel,Dr]                               */
    float* final_logits;         /* [C]                                      */
    int32_t* top5;               /* [5]                                      */
    float* ln_grad;              /* [(4L+4)*Wv] first-step gradient of the visual LayerNorm parameters (rlcf_tta_sample_ln) */
    float* ln_after;             /* [(4L+4)*Wv] adapted LayerNorm parameters                                                 */
    float* vis_grad;             /* [visual_param_count] first-step gradient of the other visual parameters (rlcf_tta_sample_visual) */
    float* vis_after;            /* [visual_param_count] adapted values                                                      */
    int32_t* step_skipped;       /* [tta_steps] 1 = the gradient of that step held an inf / NaN and the optimizer step was skipped
                                    (GradScaler.step semantics, TPT/tpt_cls_rl.py:76-79); single-sample calls only                */
} rlcf_tta_out;

/* One iteration of the harness loop TPT/tpt_cls_rl.py:251-262: reset ctx and optimizer state,
 * test_time_tuning (:47-79), final one-view inference on views[0], top-5.  views [N,3,R,R]. */
int rlcf_tta_sample(rlcf_engine*, const float* views, int N, const rlcf_tta_args* args,
                    const rlcf_tta_out* out, rlcf_stream stream);
/* LayerNorm-tuning variant (TPT/tune_cls_rl.py with CLIPCLS_TTA(only_visual=True, only_norm=True),
 * custom_clip.py:364-497): the class text features are cached, the IMAGE encoder runs with grad and the tunable set is
 * every visual LayerNorm weight/bias, laid out [ln_pre.w, ln_pre.b, (ln_1.w, ln_1.b, ln_2.w, ln_2.b) x layers, ln_post.w,
 * ln_post.b].  Per call: reset LN state + optimizer, S tuning steps (backward through the n_sel selected views only:
 * the other views get zero gradient), final clean-view inference (unless skip_final).  The engine's live LayerNorms are
 * restored to the pristine state on return; ctx_in / sparse_backward are ignored. */
int rlcf_tta_sample_ln(rlcf_engine*, const float* views, int N, const rlcf_tta_args* args, const rlcf_tta_out* out,
                       rlcf_stream stream);
int rlcf_engine_ln_param_count(rlcf_engine*);
/* (ABI version 11) How many GEMM weights of model `which` sit on the fp16 grid — their split-f16 lo halves are identically zero, as for every
 * Conv / Linear / MultiheadAttention / projection weight of a released CLIP checkpoint (the archives store them as fp16; TPT/clip/model.py:375-436
 * copies them into float32 parameters) — and how many do not (*others, may be NULL).  For such a weight the a_hi . w_lo MFMA pass of the 256x256
 * split-f16 kernel adds exact zeros and is dropped at launch: the same bits in two passes instead of three (RLCF_X3_WLO0=0 at finalize: off). */
int rlcf_engine_f16_grid_weights(rlcf_engine*, int which, int* others);
/* (ABI version 14) RLCF_PREC_F16 only.  on = 1: the image towers keep the residual stream as f16 rows and fold every LayerNorm into the
 * product that follows it (the reference's own autocast arithmetic: its LayerNorm casts back to the fp16 input type,
 * TPT/clip/model.py:157-163, and x + attention(...) adds fp16 tensors, :187-192, under TPT/tpt_cls_rl.py:52) — 7 % faster, and on the
 * 32-sample reference stream one top-1 of 32 differs from the reference's float32 run.  Default off (f32 residual stream, 32 / 32):
 * the environment variable RLCF_F16_LNFOLD=1 when the engine is created turns it on as well. */
int rlcf_engine_set_f16_lnfold(rlcf_engine*, int on);
/* on = 0: one-image calls keep every launch on the caller's stream instead of moving the reward models' pass to the engine's own side
 * stream (engine_tta_sample; what RLCF_NO_OVERLAP=1 does process-wide).  For engines that serve samples IN FLIGHT side by side from
 * several host threads (rlcf_amd.tpt_cls_rl.test_time_adapt_eval(in_flight=K), one engine per lane): the overlap comes from the other
 * lanes, and a side stream that lands on the hardware queue of another lane's stream would serialise the two.  Default on.
 * Replaces nothing in the reference (its loop runs one sample at a time, TPT/tpt_cls_rl.py:233-262). */
int rlcf_engine_set_side_stream(rlcf_engine*, int on);
/* (ABI version 14) Samples IN FLIGHT from ONE host thread.  The reference feeds its loop one test image per call
 * (TPT/tpt_cls_rl.py:233-262); a sample's step is a 64-view tower pass that fills the chip followed by a long tail of few-row launches
 * that does not.  A lanes object holds K engines over the same checkpoints and class bank (each with its own weights copies, scratch
 * and state: rlcf_engine_create x K) and one non-blocking stream per engine; rlcf_lanes_submit enqueues `count` test images (normally
 * 1) as ONE rlcf_tta_batch / rlcf_tta_batch_ln (norm_layers != 0) call on the next lane, round robin, behind an event recorded on
 * `producer` (the stream the views were produced on), and returns that lane's index (>= 0) or a negative rlcf_status.  Nothing waits
 * on the host: the caller's thread enqueues lane after lane and the device runs them side by side — sample i + 1's tower pass under
 * sample i's tail.  Per-sample arithmetic is exactly the one-image call's (bit-identical results).  rlcf_lanes_join makes `consumer`
 * wait for everything submitted so far (events; no host wait); views / outputs of a sample must stay alive until its lane has run
 * (rlcf_lanes_stream exposes lane k's hipStream_t, e.g. for torch's record_stream).  Creating a lanes object turns the engines' side
 * stream off (rlcf_engine_set_side_stream(e, 0)); destroying it waits for the lanes and restores the setting; the engines stay the
 * caller's.  rlcf_tta_lanes = create + `count` submits of one image each + join + destroy, for callers that hold a whole batch.
 * rlcf_top5_hits: out[0] += #(target[b] == top5[b, 0]), out[1] += #(target[b] in top5[b, :]) — the loop's hit counters
 * (accuracy() + AverageMeter, TPT/tpt_cls_rl.py:265-268) from the engine's own top-5 output, accumulated on the device. */
typedef struct rlcf_lanes rlcf_lanes;
rlcf_lanes* rlcf_lanes_create(rlcf_engine* const* engines, int n);
/* the same on the CALLER's streams (one per lane, e.g. torch.cuda.Stream objects, whose allocator then understands record_stream on them):
 * used, synchronised at destroy time, never destroyed */
rlcf_lanes* rlcf_lanes_create_on(rlcf_engine* const* engines, int n, const rlcf_stream* streams);
void rlcf_lanes_destroy(rlcf_lanes*);
int rlcf_lanes_count(const rlcf_lanes*);
rlcf_stream rlcf_lanes_stream(const rlcf_lanes*, int k);
int rlcf_lanes_submit(rlcf_lanes*, const float* views, int count, int N, const rlcf_tta_args* args, float* final_logits, int32_t* top5,
                      int norm_layers, rlcf_stream producer);
int rlcf_lanes_join(rlcf_lanes*, rlcf_stream consumer);
int rlcf_tta_lanes(rlcf_engine* const* engines, int lanes, const float* views, int count, int N, const rlcf_tta_args* args,
                   float* final_logits, int32_t* top5, int norm_layers, rlcf_stream stream);
int rlcf_top5_hits(const int32_t* top5, const int64_t* target, int B, float* out, rlcf_stream stream);
/* A ModifiedResNet student (arch RN50 .. RN50x64) has BatchNorms where the ViT has LayerNorms: CLIPCLS_TTA(only_norm=True) tunes the
 * weight / bias of every BatchNorm2d whose name contains 'bn' (custom_clip.py:481-485; downsample.1 stays frozen) and
 * rlcf_tta_sample_ln / rlcf_tta_batch_ln / rlcf_engine_{ln_param_count,get_ln_params,set_ln_params,momentum_update} serve them
 * unchanged: layout [bn.weight, bn.bias] per layer in named_parameters order (visual.bn1..3, then layerS.B.bn1..3).  The tuning
 * passes AND the final clean-view inference run the BatchNorms on batch statistics (the reference's CLIPCLS_TTA.train() keeps the
 * norm layers in train mode whatever the mode, custom_clip.py:487-497).
 * prior_strength (TPT/params.py:91, tune_cls_rl.py:35-44,73-76): < 0 (the default) = nn.BatchNorm2d train mode (running statistics
 * updated with momentum 0.1); >= 0 = `_modified_bn_forward`, prior = s / (s + 1) blends running and (unbiased) batch statistics. */
int rlcf_engine_set_bn_prior_strength(rlcf_engine*, int prior_strength);
/* running statistics [running_mean, running_var] per BatchNorm in execution order (visual.bn1..3, then per Bottleneck bn1, bn2, bn3,
 * downsample.1): as the last rlcf_tta_sample_ln left them (what its final inference saw), or the checkpoint's (pristine). */
/* L2-normalised image features [n, embed_dim] of the ResNet student with its BatchNorms in the form the tuning passes use (batch
 * statistics over these n images / the prior blend) and the LIVE tunable parameters (rlcf_engine_set_ln_params): what model(image)
 * computes after tune_cls_rl.py:218's model.eval(), which leaves CLIPCLS_TTA's norm layers in train mode (custom_clip.py:487-497).
 * In train mode it updates the running statistics like any other pass. */
int rlcf_engine_encode_image_bn(rlcf_engine*, const float* images, int n, float* out, rlcf_stream stream);
/* The same with the BatchNorm form chosen by the caller — form < 0: the tuning form above; form = 0: EVAL form on the running statistics
 * as the last tuning pass left them.  Version 7: every-parameter tuning of a ModifiedResNet student — CLIPCLS_TTA(only_norm=False) with
 * `--arch RN50`, the parser defaults of TPT/tune_cls_rl.py (TPT/params.py:23,73; tune_cls_rl.py:67-71; custom_clip.py:477-479) — is
 * served by rlcf_tta_sample_visual and the rlcf_engine_*visual* calls: the flat vector then holds, in named_parameters order, every
 * visual tensor whose name does not contain 'bn' (convolution weights as stored [cout, cin, k, k], downsample.1.weight / .bias, the
 * attention pool's positional_embedding, k / q / v / c projections); the 'bn' tensors stay in the norm-layer vector.  There
 * CLIPCLS_TTA.train(mode) is plain nn.Module.train(mode) (custom_clip.py:487-497): the tuning passes run the BatchNorms in train mode,
 * model(image) after model.eval() in eval form (form = 0). */
int rlcf_engine_encode_image_bn_form(rlcf_engine*, const float* images, int n, int form, float* out, rlcf_stream stream);
int rlcf_engine_bn_stats_count(rlcf_engine*);
int rlcf_engine_get_bn_stats(rlcf_engine*, float* out, int pristine, rlcf_stream stream);
/* copy the student's visual LayerNorm parameters out of / into the engine (layout above); `pristine` selects the
 * reset state (initial_state_dict, custom_clip.py:395-399) instead of the live one. */
int rlcf_engine_get_ln_params(rlcf_engine*, float* out, int pristine, rlcf_stream stream);
int rlcf_engine_set_ln_params(rlcf_engine*, const float* in, rlcf_stream stream);
/* CLIPCLS_TTA.momentum_update_model (custom_clip.py:460-475) for the tunable LayerNorm set: momentum_state = m*momentum_state +
 * (1-m)*current (the tuned parameters of the sample just processed, device [ln_param_count]); with apply != 0 (the caller's
 * update_counter reached update_freq) the reset state becomes (1-w)*checkpoint + w*momentum_state.  Makes test samples
 * order-dependent: single replica only. */
int rlcf_engine_momentum_update(rlcf_engine*, const float* current, double momentum, double update_w, int apply, rlcf_stream stream);
/* The state part of CLIPCLS_TTA.reset_classnames_and_state (custom_clip.py:449-454): visual.load_state_dict(clip_state_dict), then
 * initial_state_dict and momentum_state_dict re-initialised from it — the reset state and the EMA of the tunable LayerNorms (and of the
 * flat visual vector, once rlcf_tta_sample_visual has been used) return to the checkpoint.  Call when the harness moves to the next
 * dataset, so that the EMA of one set does not leak into the next. */
int rlcf_engine_reset_visual_state(rlcf_engine*, rlcf_stream stream);

/* Full image-encoder tuning: CLIPCLS_TTA(only_visual=True, only_norm=False) of TPT/clip/custom_clip.py:364-497, whose
 * parameters() is then clip_model.visual.parameters() (:477-479) — the `--tune_norm 0` default (TPT/params.py:73) that
 * scripts/rlcf-tune.sh runs.  Same per-call contract as rlcf_tta_sample_ln (reset, S steps through the n_sel selected views,
 * clean-view inference, pristine state restored on return) with EVERY visual parameter tuned: the LayerNorm tensors as in
 * rlcf_tta_sample_ln (out->ln_grad / ln_after), all others in one flat vector (out->vis_grad / vis_after) laid out
 * [class_embedding, positional_embedding, proj, conv1.weight, then per block attn.in_proj_weight, attn.in_proj_bias,
 * attn.out_proj.weight, attn.out_proj.bias, mlp.c_fc.weight, mlp.c_fc.bias, mlp.c_proj.weight, mlp.c_proj.bias], each tensor
 * starting at a multiple of 64 floats (rlcf_engine_visual_param_layout).  VisionTransformer students only. */
int rlcf_tta_sample_visual(rlcf_engine*, const float* views, int N, const rlcf_tta_args* args, const rlcf_tta_out* out,
                           rlcf_stream stream);
/* Image -> text retrieval with test-time adaptation of the image encoder: `tune_image` of retrieval/clip_ret_policy.py:76-103 followed
 * by the evaluation step of its loop (:171-176, 178-181: logits of the tuned model, reset).  The caption bank is a class bank without
 * learnable rows (rlcf_engine_set_class_bank with n_ctx = 0: the student's caption features = CLIPRet_TTA.set_text_features, the
 * reward model's = CLIPRewards.set_many_text_features); per call: reward features of the n query images (set_image_features), then
 * tta_steps of  logits_per_image -> top-K captions -> CLIPScore(text_index) -> baseline -> mean(r * CE) -> backward through the
 * image encoder -> AdamW over clip_model.visual.parameters(), and out->final_logits = logits_per_image[0] of the tuned encoder.
 * It is rlcf_tta_sample_visual with every image selected (the classification path picks int(N * selection_p) views): same kernels,
 * same outputs (out->topk_idx [n,K], clip_score, rewards, loss, ln_* / vis_* vectors).  K = sample_k <= 32 (scripts: 20).
 * The text -> image direction (tune_text, :106-137) is rlcf_tta_retrieval_text below. */
int rlcf_tta_retrieval_image(rlcf_engine*, const float* images, int n, const rlcf_tta_args* args, const rlcf_tta_out* out,
                             rlcf_stream stream);
/* Text -> image retrieval with test-time adaptation of the TEXT encoder: `tune_text` of retrieval/clip_ret_policy.py:106-137 with
 * CLIPRet_TTA(only_visual=False) (retrieval/custom_models.py:139-147: parameters() = every parameter whose name has no 'visual' —
 * token_embedding.weight, positional_embedding, the text transformer, ln_final, text_projection, logit_scale), followed by the
 * evaluation step of its loop (:193-196: logits_per_text of the tuned model, reset).
 * rlcf_engine_set_image_bank: the bank of n images — their L2-normalised features under the student [n, D]
 * (CLIPRet_TTA.set_image_features, custom_models.py:91-95) and under every reward model, blocks [n, Dr_m] one after another
 * (CLIPRewards.set_image_features, retrieval/clip_reward.py:130-137); device pointers, copied.  It replaces the class / caption bank
 * (the prompt, LayerNorm and image-encoder calls then refuse until rlcf_engine_set_class_bank is called again).
 * rlcf_tta_retrieval_text: tokens_host = HOST [context_length], ONE query caption (bs = 1 in the reference).  Per step:
 * logits_per_text [1, n] = exp(logit_scale) * text_features @ bank^T -> top-K images -> CLIPScore(images_index) of the reward models
 * -> baseline -> mean(r * CE) -> backward through the text encoder -> AdamW (inf / NaN gradients skip the step).
 * Outputs: out->logits [n], topk_idx [K], clip_score [K], rewards [K], loss, dlogits [n], reward_image_features [Dr_0] (= the
 * reward model's features of the QUERY TEXT) of the first step; out->vis_grad / vis_after = the flat text parameter vector
 * (rlcf_engine_text_param_layout: token_embedding.weight, positional_embedding, text_projection, per block in_proj_weight,
 * in_proj_bias, out_proj.weight, out_proj.bias, c_fc.weight, c_fc.bias, c_proj.weight, c_proj.bias, then logit_scale; slots
 * 64-float aligned), out->ln_grad / ln_after = [ln_final.weight | ln_final.bias | per block ln_1.weight ln_1.bias ln_2.weight
 * ln_2.bias]; out->final_logits [n]; out->step_skipped [tta_steps].  The engine is left in its pristine state. */
int rlcf_engine_set_image_bank(rlcf_engine*, const float* student_feats, const float* reward_feats, int n, rlcf_stream stream);
int rlcf_tta_retrieval_text(rlcf_engine*, const int32_t* tokens_host, const rlcf_tta_args* args, const rlcf_tta_out* out,
                            rlcf_stream stream);
/* floats in the flat text vector (padding included; 0 + error on failure); *ln_count (optional) = floats of the LayerNorm vector */
int64_t rlcf_engine_text_param_count(rlcf_engine*, int* ln_count, rlcf_stream stream);
/* offsets / element counts of its 3 + 8*layers + 1 tensors in the order above; returns the number of tensors (or a negative error) */
int rlcf_engine_text_param_layout(rlcf_engine*, int64_t* offsets, int64_t* numels, int max_entries, rlcf_stream stream);
/* copy the flat vector and / or the LayerNorm vector out (either pointer may be NULL): which = 0 live, 1 reset state */
int rlcf_engine_get_text_params(rlcf_engine*, float* flat, float* ln, int which, rlcf_stream stream);
/* CLIPRet_TTA.momentum_update_model (retrieval/custom_models.py:128-143) for the text side: cur_* = out->vis_after / ln_after of the
 * caption just processed; mom = m * mom + (1 - m) * cur; apply != 0: reset state = (1 - update_w) * checkpoint + update_w * mom */
int rlcf_engine_momentum_update_text(rlcf_engine*, const float* cur_flat, const float* cur_ln, double momentum, double update_w, int apply,
                                     rlcf_stream stream);
/* floats in the flat vector (padding included); 0 + error for a ModifiedResNet student */
int64_t rlcf_engine_visual_param_count(rlcf_engine*, rlcf_stream stream);
/* offsets / element counts of its 4 + 8*layers tensors in the order above; returns the number of tensors (or a negative error) */
int rlcf_engine_visual_param_layout(rlcf_engine*, int64_t* offsets, int64_t* numels, int max_entries, rlcf_stream stream);
/* copy the flat vector out: which = 0 live, 1 reset state (initial_state_dict), 2 checkpoint (clip_state_dict), 3 momentum state */
int rlcf_engine_get_visual_params(rlcf_engine*, float* out, int which, rlcf_stream stream);
/* load the live flat vector (and refresh the transposed / split copies the GEMMs read): inference with adapted weights */
int rlcf_engine_set_visual_params(rlcf_engine*, const float* in, rlcf_stream stream);
/* CLIPCLS_TTA.momentum_update_model (custom_clip.py:460-475) for the flat vector; `current` = out->vis_after of the sample just
 * processed.  Call next to rlcf_engine_momentum_update (which handles the LayerNorm set) with the same arguments. */
int rlcf_engine_momentum_update_visual(rlcf_engine*, const float* current, double momentum, double update_w, int apply, rlcf_stream stream);

/* Same for `count` consecutive samples (views [count,N,3,R,R]); top5 [count,5], final_logits
 * [count,C] (optional).  One host call per batch of test images. */
int rlcf_tta_batch(rlcf_engine*, const float* views, int count, int N, const rlcf_tta_args* args,
                   float* final_logits, int32_t* top5, rlcf_stream stream);

/* bookkeeping for bench/roofline: FLOPs actually executed by the last rlcf_tta_sample. */
/* The LayerNorm-tuning step (rlcf_tta_sample_ln) for `count` consecutive samples, B = max_views / N of them per tower pass
 * (every sample starts from the same reset state; LayerNorm parameters, their gradients, the AdamW state and the clean-view
 * inference stay per sample: row groups of the token matrix read their own (gamma, beta) sets).  Not combinable with rlcf_engine_momentum_update between the samples of one call. */
int rlcf_tta_batch_ln(rlcf_engine*, const float* views, int count, int N, const rlcf_tta_args* args, float* final_logits, int32_t* top5,
                      rlcf_stream stream);
double rlcf_engine_last_flops(rlcf_engine*);
int rlcf_engine_text_rows(rlcf_engine*);   /* rows of the packed text layout */

/* Optional per-launch timing (HIP events on the launch stream) of the GEMM kernels: enable, run one pass, read
 * {launches, total ms, total algorithmic FLOPs} of one kernel kind: 3 = gemm_nt_f16x3_v3_kernel (the dominant kernel),
 * 2 = gemm_nt_f16x3_v2_kernel, 1 = gemm_nt_f16x3_kernel, 0 = the f32-MFMA kernels, -1 = all. */
int rlcf_profile_gemm(int enable);
int rlcf_profile_read(int kind, int* launches, double* total_ms, double* total_flops);
/* the same records one by one (launch order): kind as above (10 = the fused attention forward of the split-f16 pipeline; 11 = the LayerNorm forward that writes
 * operand pairs, an HBM-bound kernel: its `flops` value is its ALGORITHMIC BYTES, dims3 = {rows, width, 0}; round 3: 12 = the
 * attention backward of the image tower (flops = 10 * pairs * width), 13 = the LayerNorm backward of a transformer block, HBM-bound
 * like 11: `flops` = bytes of x, dy and the residual gradient in + dx out),
 * HIP-event duration in ms, algorithmic FLOPs, dims3 = {M, N, K} of a GEMM / {rows, width, longest sequence} of an attention launch */
int rlcf_profile_count(void);
int rlcf_profile_entry(int i, int* kind, double* ms, double* flops, int* dims3);

#ifdef __cplusplus
}
#endif
#endif
