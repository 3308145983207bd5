// extern "C" surface of librlcf_hip.so (include/rlcf_hip.h): argument validation, error text,
// engine construction.  No torch types cross this boundary.
#include "engine.h"
#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <cstdlib>
#include <new>

static thread_local char g_err[1024] = "";
void rlcf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#include <mutex>
#include <map>
int rlcf_func_lds(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;          // (device, kernel) -> largest size granted so far
    int dev = 0;
    RLCF_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    size_t& cur = done[{dev, fn}];
    if (bytes > cur) {
        RLCF_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        cur = bytes;
    }
    return RLCF_OK;
}

extern "C" {

const char* rlcf_last_error(void) { return g_err; }
int rlcf_version(void) { return 14; }   // 14: samples in flight from one host thread (rlcf_lanes_*, rlcf_tta_lanes), rlcf_top5_hits; 13: rlcf_engine_set_side_stream (lane engines of samples in flight keep to the caller's stream); 12: rlcf_avg_entropy, rlcf_accuracy (the harness mirror's conveniences as kernels); 11: rlcf_engine_f16_grid_weights (two-pass products for weights on the fp16 grid); 10: rlcf_gemm_f16_ln / rlcf_ln_stats_final / rlcf_resid16_init (LayerNorm folded into the single-pass f16 products); 9: rlcf_gemm_f16 (single-pass f16 GEMM of the performance mode); 8: rlcf_make_views_hard (the hard_aug pre-augmentation); 7: every-parameter tuning of a ModifiedResNet student (rlcf_tta_sample_visual, rlcf_engine_encode_image_bn_form); 6: pair-operand attention, BatchNorm tuning of a ResNet student (rlcf_engine_*bn*), profile kinds 12 / 13; 5: text -> image retrieval; 4: rlcf_tta_args.n_sel, rlcf_tta_out.step_skipped, rlcf_engine_reset_visual_state, engine-owned scratch
//    // 2: rlcf_clip_cfg.vision_stages, reward slots, views, LN batch; 3: rlcf_tta_out.vis_*, rlcf_tta_sample_visual

// ------------------------------------------------------------------ op level
int rlcf_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* residual, int ldr,
                 const float* aux, int ldaux, float* C, int ldc, int M, int N, int K, float alpha, int epilogue,
                 int precision, rlcf_stream stream) {
    RLCF_ARG_CHECK(A && W && C && (precision == RLCF_PREC_F32 || precision == RLCF_PREC_F16X3));
    RLCF_ARG_CHECK(epilogue >= RLCF_EPI_NONE && epilogue <= RLCF_EPI_RELU);
    RLCF_ARG_CHECK(epilogue != RLCF_EPI_QUICKGELU_BWD || aux);
    if (precision == RLCF_PREC_F16X3) {          // op-level convenience: both operands split into stream-ordered scratch
        RLCF_ARG_CHECK(lda == K && ldw == K && M > 0 && N > 0 && K > 0);
        hipStream_t st = (hipStream_t)stream;
        const size_t ab = (size_t)M * K * 2, wb = (size_t)N * K * 2;
        char* buf = nullptr;
        RLCF_HIP_CHECK(hipMallocAsync((void**)&buf, 2 * ab + 2 * wb + 1024, st));
        void *ah = buf, *al = buf + ab, *wh = buf + 2 * ab, *wl = buf + 2 * ab + wb;
        int rc = launch_split_f16x2(A, ah, al, (int64_t)M * K, st);
        if (!rc) rc = launch_split_f16x2(W, wh, wl, (int64_t)N * K, st);
        if (!rc) rc = launch_gemm_f16x3(ah, al, K, wh, wl, K, bias, residual, ldr, aux, ldaux, C, ldc, nullptr, nullptr, 0, M, N, K, alpha,
                                        epilogue, st);
        (void)hipFreeAsync(buf, st);
        return rc;
    }
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.residual = residual; g.ldr = ldr; g.aux = aux; g.ldaux = ldaux;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.epilogue = epilogue;
    return launch_gemm_f32(g, (hipStream_t)stream);
}
int rlcf_split_f16x2(const float* x, void* hi, void* lo, int64_t n, rlcf_stream stream) {
    RLCF_ARG_CHECK(x && hi && lo);
    return launch_split_f16x2(x, hi, lo, n, (hipStream_t)stream);
}
int rlcf_gemm_f16(const void* A, int lda, const void* W, int ldw, const float* bias, const float* residual, int ldr, float* C, int ldc,
                  void* C16, int ldch, int M, int N, int K, float alpha, int epilogue, rlcf_stream stream) {
    RLCF_ARG_CHECK(A && W && (C || C16) && M > 0 && N > 0 && K > 0 && K % 64 == 0);
    RLCF_ARG_CHECK(epilogue == RLCF_EPI_NONE || epilogue == RLCF_EPI_QUICKGELU);
    // (plain f16 rows: the "lo" pointers of the pair interface are the second 32 halves of every 64-half block)
    return launch_gemm_f16x3(A, (const _Float16*)A + 32, lda, W, (const _Float16*)W + 32, ldw, bias, residual, ldr, nullptr, 0, C, ldc, C16, nullptr,
                             ldch, M, N, K, alpha, epilogue, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, 0, 1);
}

int rlcf_gemm_f16_ln(const void* A, int lda, const void* W, int ldw, const float* bias, void* out16, int ldo, int M, int N, int K, float alpha,
                     int epilogue, int mode, const float* ln_mr, const float* ln_s, float* ln_part, rlcf_stream stream) {
    return launch_gemm_f16_pp_ln(A, lda, W, ldw, bias, out16, ldo, M, N, K, alpha, epilogue, mode, ln_mr, ln_s, ln_part, (hipStream_t)stream);
}
int rlcf_ln_stats_final(const float* ln_part, int parts, int rows, int width, float* ln_mr, rlcf_stream stream) {
    return launch_ln_stats_final(ln_part, parts, rows, width, ln_mr, (hipStream_t)stream);
}
int rlcf_resid16_init(const float* x, void* x16, float* ln_mr, int rows, int width, rlcf_stream stream) {
    return launch_resid16_init(x, x16, ln_mr, rows, width, (hipStream_t)stream);
}
int rlcf_gemm_f16x3(const void* Ahi, const void* Alo, int lda, const void* Whi, const void* Wlo, int ldw, const float* bias,
                    const float* residual, int ldr, const float* aux, int ldaux, float* C, int ldc, void* Chi, void* Clo, int ldch,
                    int M, int N, int K, float alpha, int epilogue, rlcf_stream stream) {
    RLCF_ARG_CHECK(epilogue >= RLCF_EPI_NONE && epilogue <= RLCF_EPI_RELU && (epilogue != RLCF_EPI_QUICKGELU_BWD || aux));
    hipStream_t st = (hipStream_t)stream;
    // stateless call: the stream-K scratch of the 256x256 kernel (grids of >= 64 tiles that do not fill their last round) comes from
    // the stream-ordered allocator; the engine path owns its own (engine.h: gemm_ws)
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    char* ws = nullptr;
    unsigned epoch = 0;
    static int sk_on = -1;                                   // (off by default: gemm_f16x3.hip says why)
    if (sk_on < 0) { const char* ev = getenv("RLCF_X3_SK"); sk_on = ev ? atoi(ev) : 0; }
    if (sk_on && tiles >= 64 && tiles % 256 != 0 && (const char*)Alo == (const char*)Ahi + 64) RLCF_HIP_CHECK(hipMallocAsync((void**)&ws, X3_WS_BYTES, st));
    const int rc = launch_gemm_f16x3(Ahi, Alo, lda, Whi, Wlo, ldw, bias, residual, ldr, aux, ldaux, C, ldc, Chi, Clo, ldch, M, N, K, alpha,
                                     epilogue, st, nullptr, nullptr, 0, (float*)ws,
                                     ws ? X3_WS_BYTES : 0, 0, nullptr, ws ? &epoch : nullptr);
    if (ws) (void)hipFreeAsync(ws, st);
    return rc;
}
int rlcf_conv3x3_nhwc_f16x3(const float* x, const float* w, const float* bias, const float* residual, float* y, int n, int H, int W, int Cin,
                            int Cout, int epilogue, rlcf_stream stream) {
    RLCF_ARG_CHECK(x && w && y && n > 0 && H > 0 && W > 0 && (epilogue == RLCF_EPI_NONE || epilogue == RLCF_EPI_RELU));
    const int M = n * H * W, K = 9 * Cin;
    if (!gemm_f16x3_conv3x3_ok(M, Cout, Cin)) {
        rlcf_set_error("rlcf_conv3x3_nhwc_f16x3: needs Cin %% 32 == 0, Cout %% 4 == 0 and >= 192 tiles of 256x256 (n*H*W = %d, Cout = %d)", M, Cout);
        return RLCF_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t ab = (size_t)M * Cin * 4, wb = (size_t)Cout * K * 4;            // operand pairs: 4 B per element
    char* buf = nullptr;
    RLCF_HIP_CHECK(hipMallocAsync((void**)&buf, ab + wb + 4096, st));
    char *ap = buf, *wp = buf + ab, *zp = buf + ab + wb;
    RLCF_HIP_CHECK(hipMemsetAsync(zp, 0, 4096, st));
    int rc = launch_split_f16x2(x, ap, ap + 64, (int64_t)M * Cin, st, 1.0f, 1);
    if (!rc) rc = launch_split_f16x2(w, wp, wp + 64, (int64_t)Cout * K, st, 1.0f, 1);
    if (!rc) rc = launch_gemm_f16x3_conv3x3(ap, n, H, W, Cin, wp, Cout, bias, residual, Cout, y, Cout, 1.0f, epilogue, nullptr, nullptr, zp, st);
    (void)hipFreeAsync(buf, st);
    return rc;
}
int rlcf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int width, rlcf_stream stream) {
    RLCF_ARG_CHECK(x && gamma && beta && y);
    return launch_layernorm_fwd(x, gamma, beta, y, rows, width, (hipStream_t)stream);
}
int rlcf_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int rows,
                       int width, rlcf_stream stream) {
    RLCF_ARG_CHECK(x && gamma && dy && dx);
    return launch_layernorm_bwd(x, gamma, dy, nullptr, dx, dgamma, dbeta, rows, width, (hipStream_t)stream);
}
int rlcf_attention_fwd(const float* qkv, const rlcf_seq* seqs, int n_seq, int max_q_len, int width, int causal, float* out,
                       float* lse, int precision, rlcf_stream stream) {
    RLCF_ARG_CHECK(qkv && seqs && out && (precision == RLCF_PREC_F32 || precision == RLCF_PREC_F16X3 || precision == RLCF_PREC_F16));
    if (precision != RLCF_PREC_F32) {
        RLCF_ARG_CHECK(!lse);
        return launch_attention_fwd_x3(qkv, seqs, n_seq, max_q_len, width, causal, out, nullptr, nullptr, (hipStream_t)stream, 0, nullptr,
                                       precision == RLCF_PREC_F16);
    }
    return launch_attention_fwd_f32(qkv, seqs, n_seq, max_q_len, width, causal, out, lse, (hipStream_t)stream);
}
int rlcf_attention_fwd_pairs(const void* qkv_pairs, const rlcf_seq* seqs, int n_seq, int max_q_len, int width, float* out, void* out_pairs,
                             float* lse, int precision, rlcf_stream stream) {
    RLCF_ARG_CHECK(qkv_pairs && seqs && (out || out_pairs) && (precision == RLCF_PREC_F16X3 || precision == RLCF_PREC_F16));
    return launch_attention_fwd_pair(qkv_pairs, seqs, n_seq, max_q_len, width, out, out_pairs, (hipStream_t)stream, lse,
                                     precision == RLCF_PREC_F16);
}
int rlcf_attention_debug(int oneshot, int variant) { attention_pair_debug(oneshot, variant); return RLCF_OK; }
int rlcf_gemm_skinny(const float* A, int lda, const void* W_pairs, const float* bias, const float* residual, int ldr, const float* aux,
                     int ldaux, float* C, int ldc, int M, int N, int K, float alpha, int epilogue, const float* amax_in, int local_amax,
                     rlcf_stream stream) {
    RLCF_ARG_CHECK(A && W_pairs && C && gemm_skinny_x3_ok(M, N, K, lda, ldc) && ldr % 4 == 0 && ldaux % 4 == 0);
    // stateless call: the K-slice scratch (and the inverse-scale word behind it) comes from the stream-ordered allocator of the CURRENT
    // device, per call — two callers on different streams / threads / GPUs never share it (the engine passes its own workspace)
    hipStream_t st = (hipStream_t)stream;
    // sized by the launcher's own K-slice rule (min(K / 256, 256 / tiles) slices of [M, N], gemm_f16x3.hip) up to the engine's workspace
    // size, so that this call and the engine's pick the same number of slices and sum in the same order (bit-identical results)
    const int tiles = ((N + 31) / 32) * ((M + 127) / 128);
    const size_t slices = (size_t)std::max(1, std::min(K / 256, std::max(1, 256 / tiles)));
    const size_t ws_bytes = std::min<size_t>((size_t)X3_SPLITK_WS_BYTES, std::max<size_t>(slices * M * N * sizeof(float), 4096));
    float* ws = nullptr;
    RLCF_HIP_CHECK(hipMallocAsync((void**)&ws, ws_bytes + 256, st));
    const int rc = launch_gemm_skinny_x3(A, lda, W_pairs, bias, residual, ldr, aux, ldaux, C, ldc, M, N, K, alpha, epilogue, amax_in, nullptr, ws,
                                         ws_bytes, (float*)((char*)ws + ws_bytes), st, local_amax);
    const hipError_t fe = hipFreeAsync(ws, st);
    if (fe != hipSuccess && rc == RLCF_OK) { rlcf_set_error("rlcf_gemm_skinny: hipFreeAsync: %s", hipGetErrorString(fe)); return RLCF_ERR_HIP; }
    return rc;
}
int rlcf_split_pairs(const float* x, void* pairs, int64_t n, int precision, rlcf_stream stream) {
    RLCF_ARG_CHECK(x && pairs && n > 0 && n % 32 == 0 && (precision == RLCF_PREC_F16X3 || precision == RLCF_PREC_F16));
    if (precision == RLCF_PREC_F16) return launch_split_f16x2(x, pairs, nullptr, n, (hipStream_t)stream, 1.0f, 0);
    return launch_split_f16x2(x, pairs, (char*)pairs + 64, n, (hipStream_t)stream, 1.0f, 1);
}
int rlcf_attention_bwd(const float* qkv, const float* dout, const rlcf_seq* seqs, int n_seq, int max_keys, int width, int causal,
                       float* dqkv, rlcf_stream stream) {
    RLCF_ARG_CHECK(qkv && dout && seqs && dqkv);
    if (max_keys > 96) return launch_attention_bwd_long(qkv, dout, seqs, n_seq, max_keys, max_keys, width, causal, dqkv, (hipStream_t)stream);
    return launch_attention_bwd(qkv, dout, seqs, n_seq, max_keys, width, causal, dqkv, (hipStream_t)stream);
}
int rlcf_attention_bwd_flash(const float* qkv, const float* out, const float* lse, const float* dout, const rlcf_seq* seqs, int n_seq,
                            int max_q_len, int width, int causal, float* dqkv, rlcf_stream stream) {
    RLCF_ARG_CHECK(qkv && out && lse && dout && seqs && dqkv);
    return launch_attention_bwd_mfma(qkv, out, lse, dout, seqs, n_seq, max_q_len, width, causal, dqkv, (hipStream_t)stream);
}
int rlcf_attention_bwd_flash_prec(const float* qkv, const float* out, const float* lse, const float* dout, const rlcf_seq* seqs, int n_seq,
                                  int max_q_len, int width, int causal, float* dqkv, int precision, rlcf_stream stream) {
    RLCF_ARG_CHECK(qkv && out && lse && dout && seqs && dqkv && (precision == RLCF_PREC_F32 || precision == RLCF_PREC_F16X3));
    if (precision == RLCF_PREC_F32)
        return launch_attention_bwd_mfma(qkv, out, lse, dout, seqs, n_seq, max_q_len, width, causal, dqkv, (hipStream_t)stream);
    hipStream_t st = (hipStream_t)stream;
    int rows = 0;                               // rows of dout: one past the last query row any sequence names (HOST copy of the descriptors)
    std::vector<rlcf_seq> hs(n_seq);
    RLCF_HIP_CHECK(hipMemcpyAsync(hs.data(), seqs, (size_t)n_seq * sizeof(rlcf_seq), hipMemcpyDeviceToHost, st));
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    bool plain = !causal;                       // no shared prefix anywhere, no mask: the two-kernel form (attention_bwd_x3b.hip), as the engine picks it
    for (const rlcf_seq& q : hs) { rows = std::max(rows, q.q_start + q.q_len); plain = plain && q.pre_len == 0 && q.q_len <= max_q_len; }
    static int bwd_old = -1;
    if (bwd_old < 0) { const char* ev = getenv("RLCF_ATTN_BWD_OLD"); bwd_old = ev ? atoi(ev) : 0; }
    float* amax = nullptr;
    RLCF_HIP_CHECK(hipMallocAsync((void**)&amax, sizeof(float), st));
    int rc = launch_absmax(dout, (int64_t)rows * width, amax, st);
    if (rc == RLCF_OK && plain && !bwd_old) rc = launch_attention_bwd_x3_split(qkv, out, lse, dout, amax, seqs, n_seq, max_q_len, width, dqkv, st);
    else if (rc == RLCF_OK) rc = launch_attention_bwd_x3(qkv, out, lse, dout, amax, seqs, n_seq, max_q_len, width, causal, dqkv, st);
    (void)hipFreeAsync(amax, st);
    return rc;
}
int rlcf_entropy_select(const float* logits, int n, int C, int n_sel, float* entropy, int32_t* idx, rlcf_stream stream) {
    RLCF_ARG_CHECK(logits && entropy && (idx || n_sel == 0));
    return launch_entropy_select(logits, n, C, n_sel, entropy, idx, (hipStream_t)stream);
}
int rlcf_reward_loss(const float* logits, int ld_logits, const int32_t* sel, int n_sel, int C, int K, const float* class_feat,
                     const float* reward_img, int Dr, float clipscore_weight, int flags, float min_entropy_w, int32_t* topk_idx,
                     float* clip_score, float* rewards, float* loss, float* dlogits, rlcf_stream stream) {
    RLCF_ARG_CHECK(logits && class_feat && reward_img && n_sel > 0);
    float* stats = nullptr;          // stateless call: scratch from the stream-ordered allocator (the engine path owns its own)
    RLCF_HIP_CHECK(hipMallocAsync((void**)&stats, reward_loss_stats_floats(n_sel) * sizeof(float), (hipStream_t)stream));
    const int rc = launch_reward_loss(logits, ld_logits, sel, n_sel, C, K, class_feat, reward_img, Dr, clipscore_weight, flags, min_entropy_w,
                                      topk_idx, clip_score, rewards, loss, dlogits, stats, (hipStream_t)stream);
    (void)hipFreeAsync(stats, (hipStream_t)stream);
    return rc;
}
int rlcf_reward_loss_ensemble(const float* logits, int ld_logits, const int32_t* sel, int n_sel, int C, int K, int n_models,
                              const float* const* class_feats, const float* const* reward_imgs, const int* Dr, const float* mix, int mean,
                              float clipscore_weight, int flags, float min_entropy_w, int32_t* topk_idx, float* clip_score, float* rewards,
                              float* loss, float* dlogits, rlcf_stream stream) {
    RLCF_ARG_CHECK(logits && class_feats && reward_imgs && Dr && (mean || mix) && n_models >= 1 && n_models <= RLCF_MAX_REWARDS);
    RewardBank b{};
    b.n = n_models;
    for (int m = 0; m < n_models; ++m) {
        b.class_feat[m] = class_feats[m]; b.reward_img[m] = reward_imgs[m]; b.Dr[m] = Dr[m]; b.mix[m] = mean ? 1.f : mix[m];
    }
    b.post_div = mean ? (float)n_models : 1.f;
    RLCF_ARG_CHECK(n_sel > 0);
    float* stats = nullptr;
    RLCF_HIP_CHECK(hipMallocAsync((void**)&stats, reward_loss_stats_floats(n_sel) * sizeof(float), (hipStream_t)stream));
    const int rc = launch_reward_loss_bank(logits, ld_logits, sel, 1, n_sel, C, K, b, clipscore_weight, flags, min_entropy_w, topk_idx,
                                           clip_score, rewards, loss, dlogits, stats, (hipStream_t)stream);
    (void)hipFreeAsync(stats, (hipStream_t)stream);
    return rc;
}
int rlcf_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float beta1, float beta2,
                    float eps, float weight_decay, rlcf_stream stream) {
    RLCF_ARG_CHECK(p && g && m && v);
    return launch_adamw(p, g, m, v, n, step, lr, beta1, beta2, eps, weight_decay, (hipStream_t)stream);
}

size_t rlcf_make_views_scratch_bytes(int H, int n_crops, int res) { return views_scratch_bytes(H, 1 + n_crops, res); }
int rlcf_make_views(const uint8_t* image, int H, int W, const rlcf_crop* crops, int n_crops, int res, const float* mean3, const float* std3,
                    float* views, void* scratch, size_t scratch_bytes, rlcf_stream stream) {
    return launch_make_views(image, H, W, crops, n_crops, res, mean3, std3, views, scratch, scratch_bytes, (hipStream_t)stream);
}
size_t rlcf_make_views_augmix_scratch_bytes(int H, int n_crops, int res) { return views_augmix_scratch_bytes(H, 1 + n_crops, res); }
int rlcf_make_views_augmix(const uint8_t* image, int H, int W, const rlcf_crop* crops, int n_crops, int res, const float* mean3,
                           const float* std3, const rlcf_augmix_op* ops, const float* w, const float* m, float* views, void* scratch,
                           size_t scratch_bytes, rlcf_stream stream) {
    return launch_make_views_augmix(image, H, W, crops, n_crops, res, mean3, std3, ops, w, m, views, scratch, scratch_bytes, (hipStream_t)stream);
}
size_t rlcf_make_views_hard_scratch_bytes(int H, int n_crops, int res) { return views_hard_scratch_bytes(H, 1 + n_crops, res); }
int rlcf_make_views_hard(const uint8_t* image, int H, int W, const rlcf_crop* crops, int n_crops, int res, const float* mean3,
                         const float* std3, const rlcf_hard_aug* hard, const rlcf_augmix_op* ops, const float* w, const float* m,
                         float* views, void* scratch, size_t scratch_bytes, rlcf_stream stream) {
    return launch_make_views_hard(image, H, W, crops, n_crops, res, mean3, std3, hard, ops, w, m, views, scratch, scratch_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------ engine
static bool cfg_ok(const rlcf_clip_cfg* c) {
    if (!c) return false;
    const bool text_ok = c->embed_dim > 0 && c->text_width % HEAD_DIM == 0 && c->text_width <= 1024 && c->text_layers > 0 &&
                         c->context_length > 3 && c->vocab_size > 2 && c->embed_dim % 4 == 0 && c->text_heads * HEAD_DIM == c->text_width;
    if (is_resnet(*c))        // ModifiedResNet: four stages, width/2 stem channels, /32 attention-pool map (model.py:102-127)
        return text_ok && c->vision_stages[1] > 0 && c->vision_stages[2] > 0 && c->vision_stages[3] > 0 && c->vision_width % 16 == 0 &&
               c->vision_width > 0 && c->image_resolution > 0 && c->image_resolution % 32 == 0;
    return text_ok && c->vision_width % HEAD_DIM == 0 && c->vision_width <= 1024 && c->vision_patch_size > 0 &&
           c->image_resolution % c->vision_patch_size == 0 && c->vision_layers > 0;
}

static bool which_ok(const rlcf_engine* e, int which) { return e && which >= 0 && which <= RLCF_MAX_REWARDS && e->model[which].present; }

rlcf_engine* rlcf_engine_create(const rlcf_clip_cfg* student, const rlcf_clip_cfg* reward, int max_views, int max_classes, int precision) {
    return rlcf_engine_create_ensemble(student, reward, reward ? 1 : 0, max_views, max_classes, precision);
}
rlcf_engine* rlcf_engine_create_ensemble(const rlcf_clip_cfg* student, const rlcf_clip_cfg* rewards, int n_rewards, int max_views,
                                         int max_classes, int precision) {
    bool cfgs_ok = cfg_ok(student) && n_rewards >= 0 && n_rewards <= RLCF_MAX_REWARDS && (n_rewards == 0 || rewards);
    for (int m = 0; cfgs_ok && m < n_rewards; ++m) cfgs_ok = cfg_ok(&rewards[m]);
    if (!cfgs_ok || max_views <= 0 || max_classes <= 0) {
        rlcf_set_error("rlcf_engine_create: bad geometry / sizes");
        return nullptr;
    }
    if (precision != RLCF_PREC_F32 && precision != RLCF_PREC_F16X3 && precision != RLCF_PREC_F16) {
        rlcf_set_error("rlcf_engine_create: precision %d not built", precision);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { rlcf_set_error("no HIP device"); return nullptr; }
    rlcf_engine* e = new (std::nothrow) rlcf_engine();
    if (!e) return nullptr;
    e->precision = precision; e->max_views = max_views; e->max_classes = max_classes;
    { const char* ev = getenv("RLCF_F16_LNFOLD"); e->f16_lnfold = ev ? (atoi(ev) != 0) : 0; }
    e->model[0].cfg = *student; e->model[0].present = true;
    e->n_rewards = n_rewards;
    for (int m = 0; m < n_rewards; ++m) { e->model[1 + m].cfg = rewards[m]; e->model[1 + m].present = true; }
    int Tmax = 0, Wmax = 0, Pmax = 0, Kpmax = 0, Dmax = 0;
    std::vector<rlcf_seq> seqs((size_t)(1 + RLCF_MAX_REWARDS) * max_views), seqs_cls(seqs.size());
    std::vector<int32_t> cls_idx(seqs.size(), 0);
    for (int w = 0; w <= RLCF_MAX_REWARDS; ++w) {
        if (!e->model[w].present) continue;
        const rlcf_clip_cfg& c = e->model[w].cfg;
        Dmax = std::max(Dmax, c.embed_dim);
        if (is_resnet(c)) continue;                     // its workspace is sized per image chunk (resnet.hip)
        const int g = c.image_resolution / c.vision_patch_size, tok = g * g + 1;
        Tmax = std::max(Tmax, max_views * tok); Wmax = std::max(Wmax, c.vision_width); Pmax = std::max(Pmax, max_views * g * g);
        Kpmax = std::max(Kpmax, (3 * c.vision_patch_size * c.vision_patch_size + 63) / 64 * 64); Dmax = std::max(Dmax, c.embed_dim);
        for (int i = 0; i < max_views; ++i) {
            seqs[(size_t)w * max_views + i] = rlcf_seq{i * tok, tok, 0, 0};
            seqs_cls[(size_t)w * max_views + i] = rlcf_seq{i * tok, 1, i * tok + 1, tok - 1};     // query = class token, keys = all tok rows
            cls_idx[(size_t)w * max_views + i] = i * tok;
        }
    }
    bool ok = true;
    {
        Tower& t = e->vt;
        const size_t n = (size_t)Tmax * Wmax * sizeof(float);
        ok = ok && t.x.ensure(n) == 0 && t.h.ensure(n) == 0 && t.qkv.ensure(3 * n) == 0 && t.a.ensure(n) == 0 && t.f.ensure(4 * n) == 0;
        t.T = Tmax; t.width = Wmax;
    }
    ok = ok && e->patches.ensure((size_t)Pmax * Kpmax * sizeof(float)) == 0 && e->patch_out.ensure((size_t)Pmax * Wmax * sizeof(float)) == 0;
    ok = ok && e->cls_rows.ensure((size_t)max_views * Wmax * sizeof(float)) == 0 && e->cls_ln.ensure((size_t)max_views * Wmax * sizeof(float)) == 0;
    ok = ok && e->feat_raw.ensure((size_t)max_views * Dmax * sizeof(float)) == 0;
    ok = ok && e->vit_seqs.ensure(seqs.size() * sizeof(rlcf_seq)) == 0 && e->vit_seqs_cls.ensure(seqs.size() * sizeof(rlcf_seq)) == 0 &&
         e->vit_cls_idx.ensure(cls_idx.size() * sizeof(int32_t)) == 0;
    if (precision == RLCF_PREC_F16X3 || precision == RLCF_PREC_F16) {
        e->a_split_elems = std::max((size_t)Tmax * Wmax * 4, (size_t)Pmax * Kpmax);
        ok = ok && e->a_hi.ensure(e->a_split_elems * 4) == 0 && e->gemm_ws.ensure(X3_WS_BYTES) == 0 &&
             e->gemm_ws2.ensure(X3_WS_BYTES) == 0;
    }
    ok = ok && hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) == hipSuccess;
    if (ok) ok = hipMemcpy(e->vit_seqs.p, seqs.data(), seqs.size() * sizeof(rlcf_seq), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(e->vit_seqs_cls.p, seqs_cls.data(), seqs_cls.size() * sizeof(rlcf_seq), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(e->vit_cls_idx.p, cls_idx.data(), cls_idx.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { rlcf_engine_destroy(e); return nullptr; }
    return e;
}

void rlcf_engine_destroy(rlcf_engine* e) {
    if (!e) return;
    // wait for THIS engine's work only: the streams its calls were enqueued on and its own side stream — not hipDeviceSynchronize, which
    // would also wait for every other lane's queue.  (The hipFree of each buffer below still synchronises the device by HIP's own rules:
    // engines are created and destroyed at session boundaries, not per sample.)
    for (hipStream_t s : e->used_streams) (void)hipStreamSynchronize(s);
    if (e->side) (void)hipStreamSynchronize(e->side);
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    for (hipEvent_t& ev : e->ev_part) if (ev) (void)hipEventDestroy(ev);
    delete e;            // every DevBuf (towers, layouts, weights, scratch of every path) frees itself: ~DevBuf
}

int rlcf_engine_load_weight(rlcf_engine* e, int which, const char* key, const float* dev_ptr, int64_t numel) {
    RLCF_ARG_CHECK(e && which_ok(e, which) && key && dev_ptr && numel > 0);
    if (!e->model[which].present) { rlcf_set_error("model %d not configured", which); return RLCF_ERR_STATE; }
    DevBuf& d = e->model[which].raw[key];
    if (d.bytes != (size_t)numel * sizeof(float)) { d.release(); int rc = d.ensure((size_t)numel * sizeof(float)); if (rc) return rc; }
    RLCF_HIP_CHECK(hipMemcpy(d.p, dev_ptr, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice));
    RLCF_HIP_CHECK(hipStreamSynchronize(nullptr));     // a device-to-device hipMemcpy runs on the NULL stream and does not wait for it: the copy is
                                                       // complete before finalize / a call on a non-blocking stream reads the weight
    e->model[which].finalized = false;
    return RLCF_OK;
}
int rlcf_engine_finalize(rlcf_engine* e, rlcf_stream stream) {
    RLCF_ARG_CHECK(e);
    for (int w = 0; w <= RLCF_MAX_REWARDS; ++w)
        if (e->model[w].present) { int rc = engine_finalize(e, w, engine_stream(e, stream)); if (rc) return rc; }
    return RLCF_OK;
}
int rlcf_engine_set_class_bank(rlcf_engine* e, const int32_t* tokens_host, int C, int n_ctx, const float* ctx_init, int text_mode,
                               rlcf_stream stream) {
    RLCF_ARG_CHECK(e && tokens_host && (ctx_init || n_ctx == 0));
    return engine_set_class_bank(e, tokens_host, C, n_ctx, ctx_init, text_mode, engine_stream(e, stream));
}
int rlcf_engine_set_class_bank_ex(rlcf_engine* e, const int32_t* tokens_host, int C, int n_ctx, const float* ctx_init, int text_mode,
                                  const int32_t* student_tokens_host, const int32_t* ctx_pos_host, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && tokens_host && ctx_init && student_tokens_host && ctx_pos_host && n_ctx > 0);
    for (int i = 0; i < C * n_ctx; ++i) RLCF_ARG_CHECK(ctx_pos_host[i] >= 1 && ctx_pos_host[i] < e->model[RLCF_STUDENT].cfg.context_length);
    return engine_set_class_bank(e, tokens_host, C, n_ctx, ctx_init, text_mode, engine_stream(e, stream), student_tokens_host, ctx_pos_host);
}
int rlcf_encode_image(rlcf_engine* e, int which, const float* images, int n, float* feats, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && which_ok(e, which) && images && feats);
    return engine_encode_image(e, which, images, n, feats, engine_stream(e, stream));
}
int rlcf_encode_image_resized(rlcf_engine* e, int which, const float* images, int n, int in_res, float* feats, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && which_ok(e, which) && images && feats && in_res > 0);
    return engine_encode_image(e, which, images, n, feats, engine_stream(e, stream), in_res);
}
int rlcf_text_features(rlcf_engine* e, const float* ctx, float* txt, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && txt && (ctx || e->n_ctx == 0));
    return engine_text_features(e, RLCF_STUDENT, ctx, txt, engine_stream(e, stream));
}
int rlcf_reward_class_features(rlcf_engine* e, int which, float* out, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && out && which >= RLCF_REWARD && which_ok(e, which));
    if (e->C <= 0) { rlcf_set_error("reward class bank not set"); return RLCF_ERR_STATE; }
    RLCF_HIP_CHECK(hipMemcpyAsync(out, e->reward_cls[which - RLCF_REWARD].p, (size_t)e->C * e->model[which].cfg.embed_dim * sizeof(float),
                                  hipMemcpyDeviceToDevice, engine_stream(e, stream)));
    return RLCF_OK;
}
int rlcf_engine_set_reward_mix(rlcf_engine* e, const float* mix, int n, int mean) {
    RLCF_ARG_CHECK(e && n == e->n_rewards && (mean || mix));
    e->reward_mean = mean ? 1 : 0;
    for (int m = 0; m < n; ++m) e->reward_mix[m] = mix ? mix[m] : 1.f;
    return RLCF_OK;
}
int rlcf_logits(rlcf_engine* e, const float* img, int n, const float* txt, int C, float* logits, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && img && txt && logits && n > 0 && C > 0);
    return engine_logits(e, img, n, txt, C, logits, engine_stream(e, stream));
}
int rlcf_text_backward_dense(rlcf_engine* e, const float* ctx, const float* img, int n, const float* dlogits, float* dctx,
                             rlcf_stream stream) {
    RLCF_ARG_CHECK(e && ctx && img && dlogits && dctx && n > 0);
    return engine_text_backward_dense(e, ctx, img, n, dlogits, dctx, engine_stream(e, stream));
}
int rlcf_tta_sample(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* args, const rlcf_tta_out* out, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && views && args);
    return engine_tta_sample(e, views, N, args, out, engine_stream(e, stream));
}
int rlcf_tta_sample_ln(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* args, const rlcf_tta_out* out, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && views && args);
    return engine_tta_sample_ln(e, views, N, args, out, engine_stream(e, stream));
}
int rlcf_tta_sample_visual(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* args, const rlcf_tta_out* out, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && views && args);
    return engine_tta_sample_visual(e, views, N, args, out, engine_stream(e, stream));
}
int rlcf_tta_retrieval_image(rlcf_engine* e, const float* images, int n, const rlcf_tta_args* args, const rlcf_tta_out* out, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && images && args && n > 0);
    rlcf_tta_args a = *args;
    a.selection_p = 1.0f; a.n_sel = n; a.flags |= RLCF_F_NO_SELECTION;     // every query image carries reward and gradient, in loader order
    return engine_tta_sample_visual(e, images, n, &a, out, engine_stream(e, stream));
}
int rlcf_engine_set_image_bank(rlcf_engine* e, const float* student_feats, const float* reward_feats, int n, rlcf_stream stream) {
    RLCF_ARG_CHECK(e);
    return engine_set_image_bank(e, student_feats, reward_feats, n, engine_stream(e, stream));
}
int rlcf_tta_retrieval_text(rlcf_engine* e, const int32_t* tokens_host, const rlcf_tta_args* args, const rlcf_tta_out* out, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && tokens_host && args);
    return engine_tta_retrieval_text(e, tokens_host, args, out, engine_stream(e, stream));
}
int64_t rlcf_engine_text_param_count(rlcf_engine* e, int* ln_count, rlcf_stream stream) {
    if (!e || engine_text_enable(e, engine_stream(e, stream)) != RLCF_OK) return 0;
    if (ln_count) *ln_count = e->tln_count;
    return (int64_t)e->tw_count;
}
int rlcf_engine_text_param_layout(rlcf_engine* e, int64_t* offsets, int64_t* numels, int max_entries, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && offsets && numels);
    int rc = engine_text_enable(e, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    RLCF_ARG_CHECK(max_entries >= (int)e->tw_slots.size());
    for (size_t i = 0; i < e->tw_slots.size(); ++i) { offsets[i] = (int64_t)e->tw_slots[i].off; numels[i] = (int64_t)e->tw_slots[i].numel; }
    return (int)e->tw_slots.size();
}
int rlcf_engine_get_text_params(rlcf_engine* e, float* flat, float* ln, int which, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && (flat || ln) && which >= 0 && which <= 1);
    int rc = engine_text_enable(e, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    if (flat) RLCF_HIP_CHECK(hipMemcpyAsync(flat, (which ? e->tw_init : e->tw).p, e->tw_count * sizeof(float), hipMemcpyDeviceToDevice, engine_stream(e, stream)));
    if (ln) RLCF_HIP_CHECK(hipMemcpyAsync(ln, (which ? e->tln_init : e->tln).p, (size_t)e->tln_count * sizeof(float), hipMemcpyDeviceToDevice, engine_stream(e, stream)));
    return RLCF_OK;
}
int rlcf_engine_momentum_update_text(rlcf_engine* e, const float* cur_flat, const float* cur_ln, double momentum, double update_w, int apply,
                                     rlcf_stream stream) {
    RLCF_ARG_CHECK(e && cur_flat && cur_ln && momentum >= 0.0 && momentum <= 1.0);
    hipStream_t st = engine_stream(e, stream);
    int rc = engine_text_enable(e, st);
    if (rc != RLCF_OK) return rc;
    rc = launch_momentum_update(e->tw_mom.as<float>(), cur_flat, e->tw_clip.as<float>(), e->tw_init.as<float>(), (int64_t)e->tw_count, momentum,
                                update_w, apply, st);
    if (rc != RLCF_OK) return rc;
    rc = launch_momentum_update(e->tln_mom.as<float>(), cur_ln, e->tln_clip.as<float>(), e->tln_init.as<float>(), (int64_t)e->tln_count, momentum,
                                update_w, apply, st);
    if (rc != RLCF_OK) return rc;
    return apply ? engine_text_reset(e, st, true) : RLCF_OK;      // reset_initial() loads the new initial_state_dict: live copy + derived forms follow
}
int64_t rlcf_engine_visual_param_count(rlcf_engine* e, rlcf_stream stream) {
    if (!e || engine_visual_enable(e, engine_stream(e, stream)) != RLCF_OK) return 0;
    return (int64_t)e->vw_count;
}
int rlcf_engine_visual_param_layout(rlcf_engine* e, int64_t* offsets, int64_t* numels, int max_entries, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && offsets && numels);
    int rc = engine_visual_enable(e, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    RLCF_ARG_CHECK(max_entries >= (int)e->vw_slots.size());
    for (size_t i = 0; i < e->vw_slots.size(); ++i) { offsets[i] = (int64_t)e->vw_slots[i].off; numels[i] = (int64_t)e->vw_slots[i].numel; }
    return (int)e->vw_slots.size();
}
int rlcf_engine_get_visual_params(rlcf_engine* e, float* out, int which, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && out && which >= 0 && which <= 3);
    int rc = engine_visual_enable(e, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    const DevBuf* src[4] = {&e->vw, &e->vw_init, &e->vw_clip, &e->vw_mom};
    RLCF_HIP_CHECK(hipMemcpyAsync(out, src[which]->p, e->vw_count * sizeof(float), hipMemcpyDeviceToDevice, engine_stream(e, stream)));
    return RLCF_OK;
}
int rlcf_engine_set_visual_params(rlcf_engine* e, const float* in, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && in);
    int rc = engine_visual_enable(e, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    RLCF_HIP_CHECK(hipMemcpyAsync(e->vw.p, in, e->vw_count * sizeof(float), hipMemcpyDeviceToDevice, engine_stream(e, stream)));
    e->vw_dirty = true;                  // (a later tuning call starts with its own reset)
    return engine_visual_refresh(e, engine_stream(e, stream));
}
int rlcf_engine_momentum_update_visual(rlcf_engine* e, const float* current, double momentum, double update_w, int apply, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && current && momentum >= 0.0 && momentum <= 1.0);
    int rc = engine_visual_enable(e, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    rc = launch_momentum_update(e->vw_mom.as<float>(), current, e->vw_clip.as<float>(), e->vw_init.as<float>(), (int64_t)e->vw_count, momentum,
                                update_w, apply, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    if (apply) {         // model.reset() loads the new initial_state_dict (custom_clip.py:456-458): the live copy and its derived forms follow
        RLCF_HIP_CHECK(hipMemcpyAsync(e->vw.p, e->vw_init.p, e->vw_count * sizeof(float), hipMemcpyDeviceToDevice, engine_stream(e, stream)));
        e->vw_dirty = false;
        e->vw_init_is_ckpt = false;      // (an averaged reset state is off the fp16 grid: three MFMA passes from here on)
        return engine_visual_refresh(e, engine_stream(e, stream));
    }
    return RLCF_OK;
}
// Tunable norm-layer floats / BatchNorm statistics of a ModifiedResNet student, from the geometry alone (the layout engine_bn_enable
// builds: stem bn1..3, then bn1..3 of every Bottleneck tuned; downsample.1 frozen but with statistics) — the getters have no side effects
static void resnet_norm_counts(const rlcf_clip_cfg& c, int* tuned, int* stats) {
    const int w = c.vision_width;
    int p = 2 * (w / 2 + w / 2 + w), s = p, inpl = w;
    for (int st = 0; st < 4; ++st) {
        const int planes = w << st;
        for (int b = 0; b < c.vision_stages[st]; ++b) {
            p += 2 * 6 * planes; s += 2 * 6 * planes;
            const int stride = (b == 0 && st > 0) ? 2 : 1;
            if (stride > 1 || inpl != planes * 4) s += 2 * 4 * planes;        // model.py:30-36 (downsample)
            inpl = planes * 4;
        }
    }
    *tuned = p; *stats = s;
}
int rlcf_avg_entropy(const float* logits, int n, int C, float* out, rlcf_stream stream) { return launch_avg_entropy(logits, n, C, out, (hipStream_t)stream); }
int rlcf_accuracy(const float* logits, const int64_t* target, int B, int C, int32_t* top5_scratch, float* out, rlcf_stream stream) {
    return launch_accuracy(logits, target, B, C, top5_scratch, out, (hipStream_t)stream);
}
int rlcf_engine_f16_grid_weights(rlcf_engine* e, int which, int* others) {
    RLCF_ARG_CHECK(e && (which == RLCF_STUDENT || which == RLCF_REWARD || (which >= 0 && which < (int)(sizeof(e->model) / sizeof(e->model[0])))));
    int on = 0, off = 0;
    for (auto& kv : e->model[which].split_of) (kv.second.lo_zero ? on : off)++;
    if (others) *others = off;
    return on;
}
int rlcf_engine_ln_param_count(rlcf_engine* e) {
    if (!e) return 0;
    const ClipModel& s = e->model[RLCF_STUDENT];
    if (s.finalized && is_resnet(s.cfg)) {
        if (prec_single(e)) return 0;                  // BatchNorm tuning is refused in the single-pass f16 mode (engine_bn_enable says why)
        int p = 0, st = 0;
        resnet_norm_counts(s.cfg, &p, &st);
        return p;
    }
    return e->ln_count;
}
// the calls that READ or WRITE the tunable vector build a ResNet student's train form first, on the caller's stream, and hand its
// error code (and text) back
static int norm_ready(rlcf_engine* e, hipStream_t st) {
    const ClipModel& s = e->model[RLCF_STUDENT];
    if (s.finalized && is_resnet(s.cfg) && !s.rn.bn_enabled) return engine_bn_enable(e, st);
    if (e->ln_count <= 0) { rlcf_set_error("the student has no tunable norm layers (not finalized?)"); return RLCF_ERR_STATE; }
    return RLCF_OK;
}
int rlcf_engine_set_f16_lnfold(rlcf_engine* e, int on) {
    RLCF_ARG_CHECK(e);
    e->f16_lnfold = on ? 1 : 0;
    return RLCF_OK;
}
int rlcf_engine_set_side_stream(rlcf_engine* e, int on) {
    RLCF_ARG_CHECK(e);
    e->no_side = !on;
    return RLCF_OK;
}
int rlcf_engine_set_bn_prior_strength(rlcf_engine* e, int prior_strength) {
    RLCF_ARG_CHECK(e);
    e->bn_prior_strength = prior_strength < 0 ? -1 : prior_strength;
    return RLCF_OK;
}
int rlcf_engine_encode_image_bn_form(rlcf_engine* e, const float* images, int n, int form, float* out, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && images && out && n > 0 && n <= e->max_views && form <= 0);      // feat_raw / the GEMM scratch are sized for max_views
    ClipModel& s = e->model[RLCF_STUDENT];
    if (!s.finalized || !is_resnet(s.cfg)) { rlcf_set_error("rlcf_engine_encode_image_bn needs a finalized ModifiedResNet student"); return RLCF_ERR_STATE; }
    const int rc = engine_bn_enable(e, engine_stream(e, stream));
    return rc != RLCF_OK ? rc : rn_forward_train(e, s, images, n, out, engine_stream(e, stream), form < 0 ? -1 : 0);
}
int rlcf_engine_encode_image_bn(rlcf_engine* e, const float* images, int n, float* out, rlcf_stream stream) {
    return rlcf_engine_encode_image_bn_form(e, images, n, -1, out, stream);
}
int rlcf_engine_bn_stats_count(rlcf_engine* e) {
    if (!e) return 0;
    const ClipModel& s = e->model[RLCF_STUDENT];
    if (!s.finalized || !is_resnet(s.cfg) || prec_single(e)) return 0;
    int p = 0, st = 0;
    resnet_norm_counts(s.cfg, &p, &st);
    return st;
}
int rlcf_engine_get_bn_stats(rlcf_engine* e, float* out, int pristine, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && out);
    const int n = rlcf_engine_bn_stats_count(e);
    if (n <= 0) { rlcf_set_error("the student has no BatchNorm statistics (VisionTransformer, RLCF_PREC_F16, or not finalized)"); return RLCF_ERR_STATE; }
    { const int rc = norm_ready(e, engine_stream(e, stream)); if (rc != RLCF_OK) return rc; }
    RLCF_HIP_CHECK(hipMemcpyAsync(out, pristine ? e->bn_stats_init.p : e->bn_stats.p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice,
                                  engine_stream(e, stream)));
    return RLCF_OK;
}
int rlcf_engine_get_ln_params(rlcf_engine* e, float* out, int pristine, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && out);
    { const int rc = norm_ready(e, engine_stream(e, stream)); if (rc != RLCF_OK) return rc; }
    RLCF_HIP_CHECK(hipMemcpyAsync(out, pristine ? e->ln_init.p : e->ln_params.p, (size_t)e->ln_count * sizeof(float), hipMemcpyDeviceToDevice,
                                  engine_stream(e, stream)));
    return RLCF_OK;
}
int rlcf_engine_set_ln_params(rlcf_engine* e, const float* in, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && in);
    { const int rc = norm_ready(e, engine_stream(e, stream)); if (rc != RLCF_OK) return rc; }
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, in, (size_t)e->ln_count * sizeof(float), hipMemcpyDeviceToDevice, engine_stream(e, stream)));
    e->lnfold_stale = true;          // (RLCF_PREC_F16: weights with the finalize-time gamma / beta folded in are no longer valid)
    return RLCF_OK;
}
int rlcf_engine_momentum_update(rlcf_engine* e, const float* current, double momentum, double update_w, int apply, rlcf_stream stream) {
    RLCF_ARG_CHECK(e && current && momentum >= 0.0 && momentum <= 1.0);
    int rc = norm_ready(e, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    rc = launch_momentum_update(e->ln_mom.as<float>(), current, e->ln_clip.as<float>(), e->ln_init.as<float>(), e->ln_count, momentum,
                                    update_w, apply, engine_stream(e, stream));
    if (rc != RLCF_OK) return rc;
    if (apply) {         // model.reset() loads the new initial_state_dict (custom_clip.py:456-458): the live copy follows
        RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, (size_t)e->ln_count * sizeof(float), hipMemcpyDeviceToDevice,
                                      engine_stream(e, stream)));
        e->lnfold_stale = true;
    }
    return RLCF_OK;
}
int rlcf_engine_reset_visual_state(rlcf_engine* e, rlcf_stream stream) {
    RLCF_ARG_CHECK(e);
    hipStream_t st = engine_stream(e, stream);
    { const int rc = norm_ready(e, st); if (rc != RLCF_OK) return rc; }
    const size_t nb = (size_t)e->ln_count * sizeof(float);
    for (DevBuf* d : {&e->ln_params, &e->ln_init, &e->ln_mom}) RLCF_HIP_CHECK(hipMemcpyAsync(d->p, e->ln_clip.p, nb, hipMemcpyDeviceToDevice, st));
    if (e->vw_count) {
        const size_t vb = e->vw_count * sizeof(float);
        for (DevBuf* d : {&e->vw, &e->vw_init, &e->vw_mom}) RLCF_HIP_CHECK(hipMemcpyAsync(d->p, e->vw_clip.p, vb, hipMemcpyDeviceToDevice, st));
        e->vw_dirty = false;
        e->vw_init_is_ckpt = true;
        e->lnfold_stale = false;
        return engine_visual_refresh(e, st, true);
    }
    e->lnfold_stale = false;         // the live LayerNorm parameters are the checkpoint's again: the gamma / beta folded at finalize match them
    return RLCF_OK;
}
int rlcf_tta_batch(rlcf_engine* e, const float* views, int count, int N, const rlcf_tta_args* args, float* final_logits, int32_t* top5,
                   rlcf_stream stream) {
    RLCF_ARG_CHECK(e && views && args && count > 0 && top5);
    return engine_tta_batch(e, views, count, N, args, final_logits, top5, engine_stream(e, stream));
}
int rlcf_tta_batch_ln(rlcf_engine* e, const float* views, int count, int N, const rlcf_tta_args* args, float* final_logits, int32_t* top5,
                      rlcf_stream stream) {
    RLCF_ARG_CHECK(e && views && args && count > 0 && top5);
    return engine_tta_batch_ln(e, views, count, N, args, final_logits, top5, engine_stream(e, stream));
}
// ------------------------------------------------------------------ samples in flight (ABI 14)
// K engines over the same checkpoints, one non-blocking stream each; sample i runs on lane i mod K as exactly the one-image call
// (engine_tta_batch with one image), enqueued from the CALLER's thread: the hand-offs are events, there is no host thread and no queue.
struct rlcf_lanes {
    std::vector<rlcf_engine*> eng;
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> done;        // recorded on lane k after its last submitted sample
    std::vector<bool> side_was_off;
    hipEvent_t ready = nullptr;          // "the producer's work so far" (views, labels), re-recorded per submit
    int next = 0;
    bool own_streams = true;             // false: the caller's streams (rlcf_lanes_create_on): never destroyed here
};
void rlcf_lanes_destroy(rlcf_lanes* l) {
    if (!l) return;
    for (size_t k = 0; k < l->st.size(); ++k)
        if (l->st[k]) {
            (void)hipStreamSynchronize(l->st[k]);
            if (l->own_streams) {            // the engine of this lane must not keep (and later wait for) a handle that is about to die
                if (k < l->eng.size()) {
                    auto& us = l->eng[k]->used_streams;
                    us.erase(std::remove(us.begin(), us.end(), l->st[k]), us.end());
                }
                (void)hipStreamDestroy(l->st[k]);
            }
        }
    for (hipEvent_t ev : l->done) if (ev) (void)hipEventDestroy(ev);
    if (l->ready) (void)hipEventDestroy(l->ready);
    for (size_t k = 0; k < l->eng.size() && k < l->side_was_off.size(); ++k) l->eng[k]->no_side = l->side_was_off[k];
    delete l;
}
static rlcf_lanes* lanes_create(rlcf_engine* const* engines, int n, const rlcf_stream* streams);
rlcf_lanes* rlcf_lanes_create(rlcf_engine* const* engines, int n) { return lanes_create(engines, n, nullptr); }
rlcf_lanes* rlcf_lanes_create_on(rlcf_engine* const* engines, int n, const rlcf_stream* streams) {
    if (!streams) { rlcf_set_error("rlcf_lanes_create_on: streams"); return nullptr; }
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < k; ++j)
            if (streams[j] == streams[k]) { rlcf_set_error("rlcf_lanes_create_on: one stream per lane"); return nullptr; }
    return lanes_create(engines, n, streams);
}
static rlcf_lanes* lanes_create(rlcf_engine* const* engines, int n, const rlcf_stream* streams) {
    if (!engines || n < 1 || n > 16) { rlcf_set_error("rlcf_lanes_create: 1..16 engines"); return nullptr; }
    for (int k = 0; k < n; ++k)
        for (int j = 0; j <= k; ++j)
            if (!engines[k] || (j < k && engines[j] == engines[k])) { rlcf_set_error("rlcf_lanes_create: engines must be distinct (an engine serves one call at a time)"); return nullptr; }
    rlcf_lanes* l = new rlcf_lanes();
    l->own_streams = streams == nullptr;
    bool ok = hipEventCreateWithFlags(&l->ready, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < n && ok; ++k) {
        hipStream_t s = streams ? (hipStream_t)streams[k] : nullptr; hipEvent_t ev = nullptr;
        ok = (streams || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        l->st.push_back(s); l->done.push_back(ev);
    }
    if (!ok) { rlcf_set_error("rlcf_lanes_create: stream / event creation failed"); rlcf_lanes_destroy(l); return nullptr; }
    for (int k = 0; k < n; ++k) {
        l->eng.push_back(engines[k]);
        l->side_was_off.push_back(engines[k]->no_side);
        engines[k]->no_side = true;      // the overlap comes from the other lanes: a lane's one-image call keeps to the lane's stream
    }
    return l;
}
int rlcf_lanes_count(const rlcf_lanes* l) { return l ? (int)l->eng.size() : 0; }
rlcf_stream rlcf_lanes_stream(const rlcf_lanes* l, int k) { return (l && k >= 0 && k < (int)l->st.size()) ? (rlcf_stream)l->st[k] : nullptr; }
int rlcf_lanes_submit(rlcf_lanes* l, const float* views, int count, int N, const rlcf_tta_args* args, float* final_logits, int32_t* top5,
                      int norm_layers, rlcf_stream producer) {
    RLCF_ARG_CHECK(l && views && args && count > 0 && top5);
    const int k = l->next;
    RLCF_HIP_CHECK(hipEventRecord(l->ready, (hipStream_t)producer));
    RLCF_HIP_CHECK(hipStreamWaitEvent(l->st[k], l->ready, 0));
    rlcf_engine* e = l->eng[k];
    const int rc = norm_layers ? engine_tta_batch_ln(e, views, count, N, args, final_logits, top5, engine_stream(e, (rlcf_stream)l->st[k]))
                               : engine_tta_batch(e, views, count, N, args, final_logits, top5, engine_stream(e, (rlcf_stream)l->st[k]));
    RLCF_HIP_CHECK(hipEventRecord(l->done[k], l->st[k]));       // (also after an error: a join must not miss what was enqueued)
    l->next = (k + 1) % (int)l->eng.size();
    return rc < 0 ? rc : k;
}
int rlcf_lanes_join(rlcf_lanes* l, rlcf_stream consumer) {
    RLCF_ARG_CHECK(l);
    for (size_t k = 0; k < l->st.size(); ++k) RLCF_HIP_CHECK(hipStreamWaitEvent((hipStream_t)consumer, l->done[k], 0));
    return RLCF_OK;
}
int rlcf_tta_lanes(rlcf_engine* const* engines, int lanes, const float* views, int count, int N, const rlcf_tta_args* args, float* final_logits,
                   int32_t* top5, int norm_layers, rlcf_stream stream) {
    RLCF_ARG_CHECK(engines && lanes >= 1 && views && args && count > 0 && top5);
    rlcf_lanes* l = rlcf_lanes_create(engines, lanes);
    if (!l) return RLCF_ERR_HIP;
    const rlcf_clip_cfg& c = engines[0]->model[RLCF_STUDENT].cfg;
    const size_t per = (size_t)N * 3 * c.image_resolution * c.image_resolution;
    int rc = RLCF_OK;
    for (int i = 0; i < count && rc >= 0; ++i)
        rc = rlcf_lanes_submit(l, views + (size_t)i * per, 1, N, args, final_logits ? final_logits + (size_t)i * engines[0]->C : nullptr, top5 + (size_t)i * 5,
                               norm_layers, stream);
    const int rj = rlcf_lanes_join(l, stream);
    rlcf_lanes_destroy(l);               // (waits for the lanes: this convenience form is synchronous at its end; a loop keeps a lanes object)
    return rc < 0 ? rc : rj;
}
int rlcf_top5_hits(const int32_t* top5, const int64_t* target, int B, float* out, rlcf_stream stream) { return launch_top5_hits(top5, target, B, out, (hipStream_t)stream); }

double rlcf_engine_last_flops(rlcf_engine* e) { return e ? e->last_flops : 0.0; }
int rlcf_engine_text_rows(rlcf_engine* e) { return e ? e->lay[0].T : 0; }

// per-launch timing of the GEMM kernel (bench.py roofline leg)
int rlcf_profile_gemm(int enable) {
    g_prof.enabled = enable != 0;
    g_prof.n = 0;
    return RLCF_OK;
}
int rlcf_profile_read(int kind, int* launches, double* total_ms, double* total_flops) {
    RLCF_ARG_CHECK(launches && total_ms && total_flops);
    double ms = 0.0, fl = 0.0;
    int cnt = 0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (kind >= 0 && g_prof.kind[i] != kind) continue;
        ++cnt;
        RLCF_HIP_CHECK(hipEventSynchronize(g_prof.ev[2 * i + 1]));
        float t = 0.f;
        RLCF_HIP_CHECK(hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        ms += t; fl += g_prof.flops[i];
    }
    *launches = cnt; *total_ms = ms; *total_flops = fl;
    return RLCF_OK;
}

int rlcf_profile_count(void) { return g_prof.n; }
int rlcf_profile_entry(int i, int* kind, double* ms, double* flops, int* dims3) {
    RLCF_ARG_CHECK(i >= 0 && i < g_prof.n && kind && ms && flops && dims3);
    RLCF_HIP_CHECK(hipEventSynchronize(g_prof.ev[2 * i + 1]));
    float t = 0.f;
    RLCF_HIP_CHECK(hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    *kind = g_prof.kind[i]; *ms = t; *flops = g_prof.flops[i];
    for (int d = 0; d < 3; ++d) dims3[d] = g_prof.dims[3 * i + d];
    return RLCF_OK;
}

}  // extern "C"
