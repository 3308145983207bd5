// Internal launchers (device pointers, async on the stream).  One per fused kernel.
#pragma once
#include "common.h"

// LayerNorm fwd: f32 output (launch_layernorm_fwd) or a split-f16 pair for the next GEMM (launch_layernorm_fwd_split).
int launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int width, hipStream_t st,
                         int group_rows = 0, int group_stride = 0);
int launch_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* dres, float* dx,
                         float* dgamma, float* dbeta, int rows, int width, hipStream_t st, int group_rows = 0, int group_stride = 0,
                         int gamma_stride = 0, float* part_ws = nullptr, size_t part_ws_floats = 0);
// part_ws (launch_layernorm_bwd, launch_colsum, launch_vit_assemble_bwd): caller-owned scratch; with it the parameter / column sums are
// parked per wave (row block) and added in a fixed order (bit-reproducible), without it they meet by atomicAdd
#define RLCF_PARTS_WS_FLOATS ((size_t)4224 * 2 * 1024)      // 4096 (+ group round-up) walks x (gamma, beta) x width <= 1024
// images [n,3,R,R] -> patches [n*G*G, Kp] in (c,i,j) order, zero padded to Kp (f32 and/or a split-f16 pair)
int launch_im2col(const float* images, float* out, void* out_hi, void* out_lo, int n, int R, int ps, int Kp, hipStream_t st, int il = 0);
int launch_layernorm_fwd_split(const float* x, const float* gamma, const float* beta, float* y, void* yh, void* yl, int rows, int width,
                               hipStream_t st, int group_rows = 0, int group_stride = 0, int il = 0);
int launch_layernorm_add_fwd(float* x, const void* d16, const float* gamma, const float* beta, void* yh, int rows, int width, hipStream_t st,
                             int group_rows = 0, int group_stride = 0);            // x += d16; yh = LayerNorm(x) as plain f16 (RLCF_PREC_F16)
int launch_add_f16(float* x, const void* d16, int64_t n, hipStream_t st);         // x += d16
// x[n,1+G*G,W] = ln_pre([cls | patch_out] + pos)
int launch_vit_assemble(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                        const float* beta, float* x, int n, int tokens, int width, hipStream_t st, int group_imgs = 0, int group_stride = 0);
// X[r] = E[r] + (ctx_row[r] >= 0 ? ctx[ctx_row[r]] : 0)
int launch_text_assemble(const float* E, const int32_t* row_src, const int32_t* ctx_row, const float* ctx, float* X, int rows,
                         int width, int rep_rows, int ctx_stride, hipStream_t st);
// out[i] = in[idx[i]] rows (idx device, may be null = identity)
// LayerNorm folded into the single-pass f16 products (rowops.hip, end of file)
int launch_resid16_init(const float* x, void* x16, float* mr, int rows, int width, hipStream_t st);
int launch_ln_stats_final(const float* part, int P, int rows, int width, float* mr, hipStream_t st);
int launch_rows_h2f(const void* in16, int ld_in, const int32_t* idx, float* out, int ld_out, int rows, int width, hipStream_t st);
int launch_ln_fold_w(const float* W, const float* gamma, const float* beta, const float* b, float* Wg, float* bprime, int N, int K, hipStream_t st);
int launch_rowsum_f16(const void* Wf16, float inv_scale, float* s, int N, int K, hipStream_t st);
int launch_gather_rows(const float* in, int ld_in, const int32_t* idx, float* out, int ld_out, int rows, int width, hipStream_t st);
int launch_l2norm_rows(const float* in, float* out, float* inv_norm, int rows, int width, hipStream_t st);
// backward of t = u/|u|: du = (dt - t <t,dt>) * inv_norm
int launch_l2norm_bwd(const float* t, const float* dt, const float* inv_norm, float* du, int rows, int width, hipStream_t st);
// dst[rows[i]] = src[i] (scatter rows; rows device)
int launch_scatter_rows(const float* src, const int32_t* rows_idx, float* dst, int n, int width, hipStream_t st);
// dctx[j] = sum_s dX[ctx_rows[s*n_ctx + j]]   (n_copies sequences carry the ctx rows)
int launch_ctx_grad(const float* dX, const int32_t* ctx_rows, int n_copies, int n_ctx, int width, float* dctx, hipStream_t st);
// dtxt[c,:] = scale * sum_i dlogits[i,c] * img[i,:]
int launch_dtxt_dense(const float* dlogits, const float* img, int n, int C, int D, float scale, float* dtxt, hipStream_t st);
int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t st);   // out[cols,rows]
// full image-encoder tuning (weight gradients): out[cols, ld_out] = in[rows, cols]^T zero padded; out[c] += column sums; ln_pre input
int launch_transpose_pad(const float* in, int ld_in, float* out, int rows, int cols, int ld_out, hipStream_t st);
int launch_colsum(const float* in, int ld, int rows, int cols, float* out, hipStream_t st, float* part_ws = nullptr, size_t part_ws_floats = 0);
int launch_vit_preln(const float* patch_out, const float* cls, const float* pos, float* pre, int n, int tokens, int width, hipStream_t st);

int launch_attention_fwd_f32(const float* qkv, const rlcf_seq* seqs, int n_seq, int max_q_len, int width,
                             int causal, float* out, float* lse, hipStream_t st, void* out_hi = nullptr, void* out_lo = nullptr);
// pre_ws (optional, n_seq * max_pre * 2 * width floats): the contributions to the dK / dV rows of a SHARED prefix are parked per
// sequence and added in sequence order (bit-reproducible); without it they go through atomicAdd
int launch_attention_bwd(const float* qkv, const float* dout, const rlcf_seq* seqs, int n_seq, int max_keys, int width,
                         int causal, float* dqkv, hipStream_t st, float* pre_ws = nullptr, size_t pre_ws_floats = 0, int max_pre = 0);

int launch_entropy_select(const float* logits, int n, int C, int n_sel, float* entropy, int32_t* idx, hipStream_t st);
int launch_iota(int32_t* p, int n, hipStream_t st);
int launch_reward_loss(const float* logits, int ld_logits, const int32_t* sel, int n_sel, int C, int K,
                       const float* class_feat, const float* reward_img, int Dr, float clipscore_weight,
                       int flags, float min_entropy_w, int32_t* topk_idx, float* clip_score, float* rewards,
                       float* loss, float* dlogits, float* stats, hipStream_t st);
// skip: optional device flags, one per group of per_group consecutive parameters: a flagged group is left untouched (GradScaler's
// inf / NaN step skip); launch_grad_nonfinite fills such flags from the gradients
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1,
                 float b2, float eps, float wd, hipStream_t st, const int32_t* skip = nullptr, int64_t per_group = 0);
int launch_grad_nonfinite(const float* g, int64_t per_group, int groups, int32_t* flag, hipStream_t st, bool accumulate = false);   // accumulate: OR into flags already set (a second gradient buffer of the same optimizer)
int launch_avg_entropy(const float* logits, int n, int C, float* out, hipStream_t st);
int launch_accuracy(const float* logits, const int64_t* target, int B, int C, int32_t* top5_scratch, float* out, hipStream_t st);
int launch_top5_hits(const int32_t* top5, const int64_t* target, int B, float* out, hipStream_t st);
int launch_top5(const float* logits, int C, int32_t* top5, hipStream_t st);
int launch_quickgelu(const float* f, float* g, int64_t n, hipStream_t st);
int launch_build_sparse_layout(const int32_t* cls, int groups, int n_e, const int32_t* class_start, const int32_t* class_len,
                               const int32_t* class_eot_off, int lmax, int pre_rows, rlcf_seq* seqs, int32_t* eot_rows,
                               int32_t* row_src, hipStream_t st);
int launch_dtxt_sparse(const float* dlogits, const int32_t* cls, const float* img, int n_e, int K, int C, int D, float scale,
                       float* dtxt, hipStream_t st);
int launch_gemm_f16x3(const void* Ahi, const void* Alo, int lda, const void* Whi, const void* Wlo, int ldw, const float* bias,
                      const float* residual, int ldr, const float* aux, int ldaux, float* C, int ldc, void* Chi, void* Clo, int ldch,
                      int M, int N, int K, float alpha, int epilogue, hipStream_t st, const float* alpha_dev = nullptr,
                      unsigned int* amax_out = nullptr, int c_il = 0, float* splitk_ws = nullptr, size_t splitk_ws_bytes = 0,
                      int single = 0 /* plain f16 operands, one MFMA per product (RLCF_PREC_F16): see GemmX3Args */,
                      const float* out_scale_dev = nullptr /* device scalar multiplied into the split output (GemmX3Args) */,
                      unsigned* sk_epoch = nullptr /* host launch counter of this workspace: enables the stream-K tail of the 256x256 kernel */);
// gemm_f16.hip: the dedicated single-pass f16 kernel (256x256 tile, eight phases per two K tiles) and its applicability test
bool gemm_f16_p8_ok(const void* C, const void* Chi, const float* residual, const float* aux, int epilogue, const float* alpha_dev,
                    unsigned int* amax_out, const float* out_scale_dev, int N, int K, int lda, int ldw, int ldc, int ldr, int ldch);
int launch_gemm_f16_pp_ln(const void* A, int lda, const void* W, int ldw, const float* bias, void* out16, int ldo, int M, int N, int K, float alpha,
                          int epilogue, int mode, const float* ln_mr, const float* ln_s, float* ln_part, hipStream_t st);
void gemm_f16x3_next_packed_w(const void* wpk);          // plain f16 [N, K] copy of the next launch's weight hi halves (GemmX3Args::Wpk)
void gemm_f16x3_next_col_scale(const float* cs);          // per-output-column factor of the next launch_gemm_f16x3 / _conv3x3 call of this thread (GemmX3Args::col_scale)
int launch_f16_grid_check(const float* w, int64_t n, float scale, int* flag, hipStream_t st);       // flag[0] |= 1 unless every w * scale is an f16 number
// the output scale of the NEXT pair-emitting launch_gemm_f16x3 / _conv3x3 call of this thread is derived inside that launch from an upper
// bound of |C| (device scalars max|input| / max|identity|, host constants gain / bmax) and published in out2 = (s, 1 / s): gemm_x3.h
void gemm_f16x3_next_bound(const float* amax_in, const float* amax_res, float gain, float bmax, float* out2);
int launch_gemm_f16_p8(const void* A, int lda, const void* W, int ldw, const float* bias, const float* residual, int ldr, float* C, int ldc,
                       void* Cf16, int ldch, int M, int N, int K, float alpha, int epilogue, int tile_group, hipStream_t st);
// M <= 256 rows of an f32 activation against a pre-split (interleaved-pair) weight, A split in the kernel (gemm_f16x3.hip)
bool gemm_skinny_x3_ok(int M, int N, int K, int lda, int ldc);
int launch_gemm_skinny_x3(const float* A, int lda, const void* Wpairs, const float* bias, const float* residual, int ldr, const float* aux,
                          int ldaux, float* C, int ldc, int M, int N, int K, float alpha, int epilogue, const float* amax_in,
                          unsigned int* amax_out, float* ws, size_t ws_bytes, float* inv_scale_scratch, hipStream_t st, int local_amax = 0);
#define X3_SPLITK_WS_BYTES ((size_t)4 * 128 * 128 * 128 * sizeof(float))   // 4 slices x (<= 128 tiles of 128x128): the largest split-K launch
// engine GEMM scratch: split-K partial tiles (above) or the stream-K slabs (256 workgroups x 256 KB), + 8 KB of stream-K flag words at its end
#define X3_WS_BYTES ((size_t)256 * 262144 + 8192)
#define X3_SK_FLAG_BYTES_RESERVED 8192      // (the flag words at the end of the workspace: never handed out as split-K space)
int launch_dyn_scale(const float* x, int64_t n, float* scratch3, hipStream_t st);     // scratch3 = {max|x|, s, 1/s}, s = 2^k
// implicit 3x3 convolution (stride 1, pad 1) on operand pairs of the NHWC activation; zpage: >= 1 KB of zeros (gemm_f16x3.hip)
bool gemm_f16x3_conv3x3_ok(int M, int N, int Cin);
int launch_gemm_f16x3_conv3x3(const void* act_pairs, int n, int H, int W, int Cin, const void* Wpairs, int Cout, const float* bias,
                              const float* residual, int ldr, float* C, int ldc, float alpha, int epilogue, const float* alpha_dev,
                              unsigned int* amax_out, const void* zpage, hipStream_t st, void* Cpairs = nullptr,
                              const float* out_scale_dev = nullptr, int wlo0 = 0 /* the weight's lo halves are zero: two MFMA passes */);
int launch_dyn_scale_from(const float* amax_dev, float* scale2, hipStream_t st);            // scale2 = {s, 1/s} from a known max|x|
int launch_split_f16x2_dev(const float* x, void* hi, void* lo, int64_t n, const float* scale_dev, hipStream_t st, int il = 0);
int launch_split_f16x2_dyn(const float* x, void* hi, void* lo, int64_t n, float* scratch3, hipStream_t st, int il = 0);
int launch_split_f16x2(const float* x, void* hi, void* lo, int64_t n, hipStream_t st, float scale = 1.0f, int il = 0, int gelu = 0 /* operand = QuickGELU(x) */);
int launch_absmax(const float* x, int64_t n, float* out_dev, hipStream_t st);
// the same by scanning: dctx[b, j] = sum of dX rows r of group b whose source row (row_src[r], or r itself) carries learnable vector j
int launch_ctx_grad_scan(const float* dX, const int32_t* row_src, const int32_t* ctx_row, int groups, int group_rows, int n_ctx, int width,
                         float* dctx, hipStream_t st);
int launch_ctx_grad_grouped(const float* dX, const int32_t* ctx_rows, int n_copies, int n_ctx, int width, int groups, int group_rows,
                            float* dctx, hipStream_t st);
int launch_replicate_layout(const rlcf_seq* seqs, int n_seq, const int32_t* eot_rows, int C, int T, int B, rlcf_seq* seqs_rep,
                            int32_t* eot_rep, hipStream_t st);
int launch_broadcast_rows(const float* in, float* out, int n, int B, hipStream_t st);
int launch_entropy_select_batched(const float* logits, int B, int n, int C, int n_sel, float* entropy, int32_t* idx_global, hipStream_t st);
struct RewardBank {                  // reward models of one CLIPScore evaluation (CLIPRewards: n = 1; CLIPRewardsMultiple: n <= 4)
    int n;
    const float* class_feat[RLCF_MAX_REWARDS];   // [C, Dr[m]]
    const float* reward_img[RLCF_MAX_REWARDS];   // [rows, Dr[m]]
    int Dr[RLCF_MAX_REWARDS];
    float mix[RLCF_MAX_REWARDS];                 // score = (sum_m mix[m] * max(w*dot_m, 0)) / post_div
    float post_div;
};
// stats: caller-owned scratch of reward_loss_stats_floats(groups * n_sel) floats (engine: sized at set_class_bank)
size_t reward_loss_stats_floats(int rows);
int launch_topk_rows(const float* logits, int ld_logits, int rows, int C, int K, int32_t* topk_idx, float* stats, hipStream_t st);
int launch_reward_loss_bank(const float* logits, int ld_logits, const int32_t* sel, int groups, int n_sel, int C, int K,
                            const RewardBank& bank, float clipscore_weight, int flags, float min_entropy_w, int32_t* topk_idx,
                            float* clip_score, float* rewards, float* loss, float* dlogits, float* stats, hipStream_t st);
int launch_reward_loss_grouped(const float* logits, int ld_logits, const int32_t* sel, int groups, int n_sel, int C, int K,
                               const float* class_feat, const float* reward_img, int Dr, float clipscore_weight, int flags,
                               float min_entropy_w, int32_t* topk_idx, float* clip_score, float* rewards, float* loss,
                               float* dlogits, float* stats, hipStream_t st);
int launch_final_logits_batched(const float* img, int img_row_stride, const float* txt, int B, int C, int D, float scale, float* out,
                                hipStream_t st);
int launch_group_logits(const float* img, int rows_per_group, const float* txt, int B, int C, int D, float scale, float* out, hipStream_t st);
int launch_top5_batched(const float* logits, int B, int C, int32_t* top5, hipStream_t st);
int launch_attention_fwd_x3(const float* qkv, const rlcf_seq* seqs, int n_seq, int max_q_len, int width, int causal, float* out,
                            void* out_hi, void* out_lo, hipStream_t st, int il = 0, float* lse = nullptr, int single = 0 /* plain f16, one MFMA per product */,
                            const int32_t* row_seq_start = nullptr /* packed short sequences: see attention_x3.hip */);
// attention_pair.hip: the same forward on producer-emitted operands — qkv2 = the in_proj output as interleaved f16 pairs (single: plain
// f16), K / V staged by LDS-DMA, V^T through ds_read_b64_tr_b16; non-causal sequences (optional prefix) only
int launch_attention_fwd_pair(const void* qkv2, const rlcf_seq* seqs, int n_seq, int max_q_len, int width, float* out, void* out_pairs,
                              hipStream_t st, float* lse = nullptr, int single = 0);
void attention_pair_debug(int oneshot, int var);      // measurement switches of attention_pair.hip
int launch_attention_bwd_mfma(const float* qkv, const float* out, const float* lse, const float* dout, const rlcf_seq* seqs, int n_seq,
                              int max_q_len, int width, int causal, float* dqkv, hipStream_t st, float* park = nullptr);
// adds the parked per-query-block contributions to dK / dV in block order (attention_bwd_x3.hip; also behind the f32-MFMA backward)
int launch_attention_bwd_park_reduce(const float* park, const rlcf_seq* seqs, int n_seq, int park_rows, int n_qb, int width, float* dqkv, hipStream_t st);
// split-f16 form of launch_attention_bwd_mfma (attention_bwd_x3.hip); amax_dout: device scalar, max |dout| (launch_absmax)
// two-kernel form for sequences without a shared prefix / causal mask (attention_bwd_x3b.hip): dqkv written completely, single writers
int launch_attention_bwd_x3_split(const float* qkv, const float* out, const float* lse, const float* dout, const float* amax_dout, const rlcf_seq* seqs,
                                  int n_seq, int max_q_len, int width, float* dqkv, hipStream_t st);
int launch_attention_bwd_x3(const float* qkv, const float* out, const float* lse, const float* dout, const float* amax_dout, const rlcf_seq* seqs,
                            int n_seq, int max_q_len, int width, int causal, float* dqkv, hipStream_t st, float* park = nullptr);
int launch_attention_bwd_long(const float* qkv, const float* dout, const rlcf_seq* seqs, int n_seq, int max_q_len, int max_keys,
                              int width, int causal, float* dqkv, hipStream_t st);
int launch_vit_assemble_bwd(const float* patch_out, const float* cls, const float* pos, const float* dy, float* dgamma, float* dbeta, int n,
                            int tokens, int width, hipStream_t st, int group_imgs = 0, int group_stride = 0, float* part_ws = nullptr,
                            size_t part_ws_floats = 0);
int launch_dimg(const float* dlogits, const float* txt, int n, int C, int D, float scale, float* dimg, hipStream_t st);
int launch_bicubic(const float* in, float* out, int planes, int Ri, int Ro, hipStream_t st);

// views.hip
size_t views_scratch_bytes(int H, int n_views, int res);
int launch_make_views(const uint8_t* image, int H, int W, const rlcf_crop* crops_host, int n_crops, int res, const float* mean3,
                      const float* std3, float* views, void* scratch, size_t scratch_bytes, hipStream_t st, uint8_t* u8_out = nullptr);
size_t views_augmix_scratch_bytes(int H, int n_views, int res);
int launch_make_views_augmix(const uint8_t* image, int H, int W, const rlcf_crop* crops_host, int n_crops, int res, const float* mean3,
                             const float* std3, const rlcf_augmix_op* ops_host, const float* w_host, const float* m_host, float* views,
                             void* scratch, size_t scratch_bytes, hipStream_t st);
size_t views_hard_scratch_bytes(int H, int n_views, int res);
int launch_make_views_hard(const uint8_t* image, int H, int W, const rlcf_crop* crops_host, int n_crops, int res, const float* mean3,
                           const float* std3, const rlcf_hard_aug* hard_host, const rlcf_augmix_op* ops_host, const float* w_host,
                           const float* m_host, float* views, void* scratch, size_t scratch_bytes, hipStream_t st);
int launch_momentum_update(float* mom, const float* cur, const float* clip, float* init, int64_t n, double momentum, double update_w, int apply,
                           hipStream_t st);
