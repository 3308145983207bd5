// Host-side runtime of the RLCF hot path: owns weights, layouts and workspace, sequences the
// HIP kernels of one CLIP tower pass / one TTA sample on a stream.  No allocation per sample.
#pragma once
#include <map>
#include <unordered_map>
#include <string>
#include <vector>
#include "kernels.h"

struct DevBuf {                      // owning, move-only: an engine that is deleted gives back every buffer it ever grew
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
    ~DevBuf() { release(); }
    int ensure(size_t n);            // (re)allocate if smaller; contents undefined after growth
    void release();
    template <typename T> T* as() const { return (T*)p; }
};

struct BlockW {                      // one ResidualAttentionBlock (TPT/clip/model.py:171-192)
    const float *ln1_w, *ln1_b, *in_w, *in_b, *out_w, *out_b, *ln2_w, *ln2_b, *fc_w, *fc_b, *proj_w, *proj_b;
    const float *in_wT = nullptr, *out_wT = nullptr, *fc_wT = nullptr, *proj_wT = nullptr;   // for dX = dY.W
};
struct TowerW {
    int layers = 0, width = 0;
    std::vector<BlockW> blk;
};

struct TextLayout {                  // packed token matrix of a class bank
    int T = 0, C = 0, n_seq = 0, max_q_len = 0, max_keys = 0, pre_rows = 0, lmax = 0, n_copies = 0, n_ctx = 0;
    bool ctx_general = false;        // learnable rows at class-dependent positions (class_token_position 'front' / 'middle'): the ctx
                                     // gradient is gathered by scanning ctx_row instead of through the fixed row lists
    DevBuf seqs, eot_rows, ctx_row, E, class_start, class_len, class_eot_off, ctx_rows_list;
    // packed form of the class sequences for the no-grad text passes (shared-prefix layouts): runs of consecutive rows holding several
    // whole sequences, at most 32 - pre_rows rows each, + per row the first row of its sequence (attention_x3.hip, `rss`)
    DevBuf pk_seqs, pk_rss;
    int n_pk = 0;
    DevBuf row_token, row_pos;       // per row: token id (-1 = learnable row) and position index — E is rebuilt from them when the
                                     // token / positional embeddings are tuned (text-encoder tuning)
    long tokens_total = 0;           // sum of rows that carry real tokens (FLOP accounting)
    long attn_pairs = 0;             // visible (query,key) pairs over all sequences
    double mean_len = 0;             // mean class_len
};

struct SavedLayer { float *x, *qkv, *a, *x1, *f, *lse; };      // lse [T, heads]: log-sum-exp of the attention rows (MFMA backward)

struct Tower {                       // workspace of one transformer pass over T rows
    int T = 0, width = 0;
    DevBuf x, h, qkv, a, f;          // f32 activations
    DevBuf h2, a2, f2;               // interleaved split-f16 operand pairs written by the producers (F16X3 mode): LN out, attention out, fc out
    DevBuf x16, lnmr, lnpart;        // RLCF_PREC_F16 image towers with folded LayerNorms: f16 residual stream, (mean, rstd) per row, partial row statistics
    int x3_T = 0, x3_W = 0;
    DevBuf saved;                    // per-layer saved activations for backward
    std::vector<SavedLayer> sv;
    int saved_T = 0, saved_layers = 0;
};

struct ConvW {                       // conv with folded BatchNorm
    const float* w = nullptr; const float* b = nullptr; int cin = 0, cout = 0, k = 1, Kp = 0;
    // the same convolution with the BatchNorm scale kept OUT of the weight: wg = W permuted / padded like w but unscaled, cs[co] = gamma /
    // sqrt(var + eps) applied per output column in the GEMM epilogue (GemmX3Args::col_scale).  Set only when wg sits on the fp16 grid (a
    // released checkpoint's convolution weights do): its products then run two MFMA passes; used by the pair-emitting path
    const float* wg = nullptr; const float* cs = nullptr;
    float gain = 0.f, bmax = 0.f;    // max_row sum_k |w[row, k]| and max |b|: |conv(x)| <= gain max|x| + bmax, the bound the scale of a
};                                   // pair-emitting epilogue is chosen from BEFORE the launch (resnet.hip)
struct BottleW { ConvW c1, c2, c3, down; bool has_down = false; int stride = 1; };                          // model.py:10-55
// train-form view of one Conv2d(bias=False) + BatchNorm2d pair of a ModifiedResNet STUDENT whose norm layers are tuned
// (CLIPCLS_TTA(only_norm=True) on a ResNet, TPT/clip/custom_clip.py:481-497; BatchNorm forward of TPT/tune_cls_rl.py:35-44,73-76)
struct BnUnit {
    ConvW raw;                       // the UNFOLDED weight in the GEMM layout [cout, Kp] ((ky,kx,ci) order, zero padded); raw.b = nullptr
    const float* wT = nullptr;       // operand of dX = dZ . W: [cin, cout] (1x1) or [cin, KpT] with the taps flipped (3x3); nullptr: no dX (stem conv1)
    int KpT = 0;
    int pofs = -1;                   // offset of (gamma | beta) in the tunable vector (e->ln_params); -1: frozen (downsample.1)
    int sofs = 0;                    // offset of (running_mean | running_var) in the statistics vector (e->bn_stats)
    const float *gamma0 = nullptr, *beta0 = nullptr;     // checkpoint tensors (what a frozen unit reads); every-parameter tuning: the live
                                                         // downsample.1 weight / bias inside e->vw
    // every-parameter tuning (engine_rn_visual_enable): the live convolution weight as stored ([cout, cin, k, k], inside e->vw; the
    // GEMM-layout copies raw.w / wT and their split pairs are rebuilt from it after every optimizer step: rn_visual_refresh) and
    // where the unit's gradients go in e->vw_grad; the input the last train-form forward fed it (weight gradient = dZ^T . patches)
    const float* w_live = nullptr;
    long vofs_w = -1, vofs_g = -1;                       // offsets of the weight / of downsample.1's (weight | bias) in the flat vector
    float* wT_buf = nullptr;                             // (wT, writable)
    const float* in_ptr = nullptr; int in_H = 0, in_W = 0, in_stride = 1; bool in_nchw = false;
};
struct ResNetW {                     // ModifiedResNet (TPT/clip/model.py:94-154), inference form
    bool present = false;
    // train form (engine_bn_enable): stem conv1..3, then per Bottleneck conv1, conv2, conv3 (, downsample) in execution order
    bool bn_enabled = false;
    std::vector<BnUnit> units;
    std::vector<int> block_unit;     // index of a block's first unit
    const float *q_wT = nullptr, *kv_wT = nullptr, *c_wT = nullptr;      // transposes for the attention pool's backward
    // every-parameter tuning of the student (CLIPCLS_TTA(only_norm=False) on a ModifiedResNet, the parser defaults of tune_cls_rl.py):
    // offsets of the attention pool's tensors in the flat vector e->vw, in named_parameters order (positional_embedding, k_proj.weight,
    // k_proj.bias, q_proj.weight, q_proj.bias, v_proj.weight, v_proj.bias, c_proj.weight, c_proj.bias); the live k|v concatenations
    bool full_enabled = false;
    long vofs_pool[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
    float *kv_w_buf = nullptr, *kv_b_buf = nullptr, *q_wT_buf = nullptr, *kv_wT_buf = nullptr, *c_wT_buf = nullptr;
    int n_stats = 0;                 // floats of the statistics vector
    ConvW stem[3];
    std::vector<BottleW> blocks;
    const float *pos = nullptr, *q_w = nullptr, *q_b = nullptr, *kv_w = nullptr, *kv_b = nullptr, *c_w = nullptr, *c_b = nullptr;
    int E = 0, heads = 0, out_hw = 0;
    size_t act_per_img = 0, col_per_img = 0;     // floats of workspace per image
};

struct ClipModel {
    rlcf_clip_cfg cfg{};
    bool present = false, finalized = false;
    std::map<std::string, DevBuf> raw;     // state-dict tensors as loaded
    std::vector<DevBuf> derived;           // transposed / bf16 copies
    TowerW vis, txt;
    ResNetW rn;                            // image tower when cfg.vision_stages[0] > 0
    const float *conv_w = nullptr;         // [Wv, Kp] zero padded
    const float *conv_raw = nullptr;       // visual.conv1.weight as stored [Wv, 3*ps*ps] (== conv_w when Kp == 3*ps*ps)
    const float *cls = nullptr, *vpos = nullptr, *lnpre_w = nullptr, *lnpre_b = nullptr, *lnpost_w = nullptr, *lnpost_b = nullptr;
    const float *vprojT = nullptr;         // [D, Wv]
    const float *tok_emb = nullptr, *tpos = nullptr, *lnf_w = nullptr, *lnf_b = nullptr;
    const float *tproj = nullptr;          // [Wt, D] as stored
    const float *tprojT = nullptr;         // [D, Wt]
    const float* vproj = nullptr;          // [Wv, D] as stored (backward operand)
    float logit_scale_exp = 1.f;
    int Kp = 0, tokens = 0;
    struct SplitW { void *hi, *lo; float inv_scale; bool lo_zero = false; void* hi_only = nullptr; bool lo_zero_ckpt = false; /* lo_zero of the checkpoint's values (hi_only is their copy) */ /* lo_zero: the hi halves alone as a plain f16 matrix */ };   // W * 2^s = hi + lo; inv_scale = 2^-s; lo_zero: W 2^s sits on the f16 grid (two MFMA passes suffice)
    std::unordered_map<const float*, SplitW> split_of;                    // f32 weight -> split-f16 copy (F16X3 / F16 modes)
    std::unordered_map<const float*, SplitW> f16_of;                      // f32 weight -> plain f16 copy (.hi; RLCF_PREC_F16 mode only)
    // RLCF_PREC_F16 image towers: in_proj / c_fc weight with the preceding LayerNorm folded in (engine.hip make_lnfold): w16 = f16 copy of
    // W diag(gamma) 2^s, inv_scale = 2^-s, s = row sums of that copy times 2^-s, bprime = W beta + b
    struct LnFold { void* w16; float inv_scale; const float* s; const float* bprime; };
    std::unordered_map<const float*, LnFold> lnfold_of;
};

// full image-encoder tuning (CLIPCLS_TTA only_norm=False): one entry per non-LayerNorm visual tensor of the flat buffer, and the
// derived copies (padded / transposed / split-f16) that have to follow the live weights after an optimizer step or a reset
struct VwSlot { size_t off, numel; };
struct VwRefresh { int kind; const float* src; float* dst; size_t rows, cols; void *hi, *lo; float scale; int il; ClipModel::SplitW* sw = nullptr; };
enum { VW_PAD = 0, VW_TRANSPOSE = 1, VW_SPLIT = 2 };

struct rlcf_engine {
    int precision = RLCF_PREC_F32, max_views = 0, max_classes = 0;
    ClipModel model[1 + RLCF_MAX_REWARDS];      // [0] student, [1..] reward slots (CLIPRewards: 1, CLIPRewardsMultiple: up to 4)
    int n_rewards = 0, reward_mean = 0;
    float reward_mix[RLCF_MAX_REWARDS] = {1.f, 0.f, 0.f, 0.f};
    // ViT workspace (shared by student and reward passes)
    Tower vt;
    DevBuf patches, patch_out, vit_seqs /*[1 + rewards][max_views]*/, cls_rows, cls_ln, feat_raw, resized;
    DevBuf vit_seqs_cls, vit_cls_idx;   // per model: one-query descriptors of the class token (keys: the whole sequence) and its row ids
    DevBuf cls_a2, cls_h2, cls_f2;      // operand pairs of the class-token-only last block [views, W] / [views, 4W]
    // text
    int text_mode = RLCF_TEXT_SHARED, n_ctx = 0, C = 0;
    TextLayout lay[1 + RLCF_MAX_REWARDS];   // [student], [reward slots]
    Tower tt;                        // text workspace (max of both layouts)
    DevBuf eot_x, eot_ln, u, inv_norm, txt, txt0;    // [C,*]; txt0 = text features at ctx_init
    DevBuf ctx_init, ctx, adam_m, adam_v, ctx_grad;  // [n_ctx, Wt]
    DevBuf reward_cls[RLCF_MAX_REWARDS];   // [C, Dr] per reward slot
    // sparse backward layout (n_e entries)
    int sp_max_e = 0, sp_T = 0;
    DevBuf sp_seqs, sp_eot_rows, sp_row_src, sp_ctx_rows_list, sp_dtxt, sp_txt, sp_inv_norm, sp_eot_x, sp_eot_ln, sp_u, sp_du, sp_dxe;
    Tower st;                        // sparse pass workspace (with saved activations)
    DevBuf dX, dA, dH, dF, dQKV;     // backward scratch (sized for the largest backward pass)
    DevBuf attn_pre_ws;              // per-sequence dK / dV contributions to shared prefix rows (ordered reduction, text attention backward)
    int bwd_T = 0;
    // TTA step scratch
    DevBuf img_feat, sel_feat, logits, sel_logits, entropy, sel_idx, rimg[RLCF_MAX_REWARDS], views_sel, topk_idx, clip_score, rewards, loss, dlogits,
        dtxt_dense, final_logits, top5;
    // sample-batched step (rlcf_tta_batch): B test images share every tower pass
    DevBuf b_pk_rep, b_rss_rep;      // the packed sequence runs replicated per test sample
    DevBuf b_seqs_rep, b_eot_rep, b_ctx, b_m, b_v, b_grad, b_txt, b_eot_x, b_eot_ln, b_u, b_inv, b_logits;
    int b_cap = 0, sp_groups = 0;
    // LayerNorm-tuning path (CLIPCLS_TTA only_norm): all visual LN parameters of the student in one tunable buffer
    const float* lng_base = nullptr;  // when set: per-view LayerNorm sets [views, ln_count] override the student's tunable LayerNorms
    int lng_views = 1;               // consecutive views that share one set (1: per view; n_sel: per test sample)
    DevBuf b_ln, b_ln_m, b_ln_v, b_ln_grad;   // per-sample LayerNorm sets of the batched LN-tuning path [B, ln_count]
    DevBuf ln_clip, ln_mom;          // pristine checkpoint values / momentum state of the tunable LayerNorms (momentum_update)
    DevBuf ln_params, ln_init, ln_grad, ln_m, ln_v, vit_inv_norm, cls_row_idx, dfeat, dcls, txt0T, ln_feat;
    int ln_count = 0;                // (4*layers + 4) * Wv
    size_t bwd_elems = 0;
    // full image-encoder tuning: class_embedding, positional_embedding, proj, conv1.weight, then per block in_proj_weight, in_proj_bias,
    // out_proj.weight, out_proj.bias, c_fc.weight, c_fc.bias, c_proj.weight, c_proj.bias (each slot 64-float aligned) in one flat buffer
    DevBuf vw, vw_init, vw_grad, vw_m, vw_v, vw_clip, vw_mom;
    size_t vw_count = 0;
    bool no_side = false;            // rlcf_engine_set_side_stream(e, 0): one-image calls keep to the caller's stream
    bool vw_dirty = false;           // live weights differ from the reset state
    bool vw_init_is_ckpt = true;     // the reset state (vw_init) still holds the checkpoint's values (no EMA has been applied to it)
    std::vector<VwSlot> vw_slots;
    std::vector<VwRefresh> vw_refresh;
    // text-encoder tuning (retrieval text -> image, CLIPRet_TTA only_visual=False: every non-visual parameter, custom_models.py:139-147):
    // token_embedding.weight, positional_embedding, text_projection, per block the 8 Linear tensors (order of vw), logit_scale — one
    // flat buffer; the text LayerNorms [ln_final.w | ln_final.b | per block ln_1.w ln_1.b ln_2.w ln_2.b] in a second one
    DevBuf tw, tw_init, tw_grad, tw_m, tw_v, tln, tln_init, tln_grad, tln_m, tln_v;
    DevBuf tw_clip, tw_mom, tln_clip, tln_mom;       // pristine checkpoint values / momentum state (CLIPRet_TTA.momentum_update_model)
    size_t tw_count = 0;
    int tln_count = 0;
    bool tw_dirty = false;
    std::vector<VwSlot> tw_slots;
    std::vector<VwRefresh> tw_refresh;
    TextLayout qlay[1 + RLCF_MAX_REWARDS];   // the query caption: [student], [reward slots]
    bool image_bank = false;         // the bank of e->C entries holds IMAGE features (rlcf_engine_set_image_bank), not texts
    DevBuf q_feat, q_dfeat, q_ls;    // query text features / their gradient [D]; {d logit_scale} scratch
    DevBuf wg_yt, wg_xt, w_hi;       // weight-gradient GEMM operands: dY^T, X^T (token dimension padded) and the split copy of X^T
    DevBuf rn_pairs[3] /*block input / conv1 / conv2 outputs as operand pairs*/, rn_scale, rn_buf[5], rn_col, rn_tok, rn_q, rn_kv, rn_att, rn_amax /*max|activation| per buffer, written by GEMM epilogues*/;   // ModifiedResNet workspace (one chunk of images)
    DevBuf bwd_amax;                 // max|dF| handed from one backward GEMM's epilogue to the next one's operand scale
    DevBuf dyn;                      // {max|A|, s, 1/s} of a dynamically scaled split (ResNet activations)
    DevBuf a_hi;                     // interleaved split copy of the current GEMM A operand (F16X3 mode)
    size_t a_split_elems = 0;
    DevBuf gemm_ws;                  // split-K workspace of the small-grid split-f16 GEMM (X3_SPLITK_WS_BYTES, sized at create)
    // one-image calls: the reward models' tower pass of the selected views runs on a second stream next to the student's sparse text
    // forward (both leave most of the 256 CUs idle at one image's sizes); fork / join by events, its own split-K workspace
    DevBuf gemm_ws2;
    unsigned ws_epoch[2] = {0, 0};   // stream-K launch counters of gemm_ws / gemm_ws2 (gemm_f16x3.hip: flags carry the launch's epoch)
    // ... its own copies of the image-tower scratch (engine_encode_image on the side stream: the reward models' pass of one part of a
    // sample batch runs beside the student tower of the next part, tta_batch_pipelined) ...
    // norm-layer tuning of a ModifiedResNet student: running statistics of every BatchNorm2d (reset per sample, updated by train-mode
    // passes), `--prior_strength` (< 0: torch's train-mode BatchNorm), activations saved by the train-form forward
    DevBuf rn_wg_tmp;                // GEMM-layout weight gradient of one 3x3 convolution / the k|v projection (every-parameter tuning)
    DevBuf bn_stats, bn_stats_init, bn_scratch, bn_saved, bn_grad_a, bn_grad_b, bn_grad_c, bn_dlog, bn_amax;
    int f16_lnfold = 0;              // RLCF_PREC_F16 image towers: f16 residual stream with the LayerNorms folded into the products (opt-in: RLCF_F16_LNFOLD=1 at create / rlcf_engine_set_f16_lnfold)
    bool lnfold_stale = false;       // the student's LayerNorm parameters were written after finalize (rlcf_engine_set_ln_params, an applied EMA): the
                                     // gamma / beta folded into the RLCF_PREC_F16 image-tower weights no longer match them -> the unfolded pipeline runs
    DevBuf zpage;                    // 4 KB of zeros: what the implicit 3x3 convolution reads outside the image
    DevBuf parts_ws, attn_park;      // scratch of the bit-reproducible reductions: parameter-gradient partial sums; dK / dV per query block
    int bn_prior_strength = -1;
    std::vector<float*> bn_z, bn_y;  // per unit: pre-BatchNorm GEMM output and the unit's output, inside bn_saved
    std::vector<float*> bn_ms;       // per unit: (mean | rstd) used by the pass, inside bn_scratch
    int bn_saved_n = 0;
    float *bn_q = nullptr, *bn_kv = nullptr;     // attention pool: projected query [n, E] and keys|values [n*T, 2E] of the saved pass
    float *bn_tok = nullptr, *bn_att = nullptr;  // ... its tokens [n*T, E] and attended output [n, E] (weight gradients of every-parameter tuning)
    struct ImgSide { DevBuf patch_out, patches, cls_rows, cls_ln, feat_raw, cls_a2, cls_h2, cls_f2, resized; Tower vt; } side_img;
    hipEvent_t ev_part[8] = {};      // "student tower of part k done" (created on first use)
    DevBuf a_hi2;                    // ... and its own A-operand split buffer: the main stream's text passes re-split into a_hi (M > 512:
    size_t a_split2_elems = 0;       // dense text mode, large sample_k / selection_p) while the side stream's reward towers read theirs
    int ws_sel = 0;                  // which workspace the GEMM launchers hand out (1 while the side stream's launches are enqueued)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // the caller's streams this engine has enqueued work on (api.hip: engine_stream; a handful at most — a lane uses one): what
    // rlcf_engine_destroy waits for, instead of the whole device
    std::vector<hipStream_t> used_streams;
    DevBuf rl_stats;                 // per-row scratch of the reward / loss kernels
    DevBuf step_skip;                // int32 per test sample: gradient held an inf / NaN -> optimizer step skipped (GradScaler semantics)
    // packed class-sequence runs of the text pass being enqueued (set by text_forward around transformer_forward)
    const rlcf_seq* pk_cur = nullptr; int n_pk_cur = 0; const int32_t* rss_cur = nullptr;
    double last_flops = 0.0;
};

struct GemmProfile {                 // optional per-launch timing of the dominant (GEMM) kernel
    bool enabled = false;
    int n = 0;
    std::vector<hipEvent_t> ev;      // 2 per launch
    std::vector<double> flops;
    std::vector<int> kind;           // 0 = f32 MFMA kernels, 1 = f16x3 128x128, 2 = f16x3 256x128, 3 = f16x3 256x256 (the dominant kernel),
                                     // 10 = fused attention forward (split-f16 pipeline), 11 = LayerNorm forward with pair output (HBM-bound:
                                     // `flops` holds its algorithmic bytes)
    std::vector<int> dims;           // 3 per launch: GEMM M, N, K; attention: rows, width, longest sequence
};
extern int g_last_x3_variant;
extern GemmProfile g_prof;

static inline hipStream_t engine_stream(rlcf_engine* e, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (e) {
        bool seen = false;
        for (hipStream_t s : e->used_streams) seen = seen || s == st;
        if (!seen && e->used_streams.size() < 64) e->used_streams.push_back(st);
    }
    return st;
}
static inline bool is_resnet(const rlcf_clip_cfg& c) { return c.vision_stages[0] > 0; }
// RLCF_PREC_F16 is the split-f16 engine with ONE change: the forward tower pipeline (LayerNorm -> GEMM -> attention -> GEMM ... of
// transformer_forward, the patch embedding) carries plain f16 operands and spends one MFMA per product — the arithmetic of the
// reference's fp16-autocast GPU path (tpt_cls_rl.py:52).  Everything else (backward, small GEMMs, losses) stays f32-grade.
static inline bool prec_x3(const rlcf_engine* e) { return e->precision == RLCF_PREC_F16X3 || e->precision == RLCF_PREC_F16; }
static inline bool prec_single(const rlcf_engine* e) { return e->precision == RLCF_PREC_F16; }
// resnet.hip
int resnet_finalize(rlcf_engine* e, ClipModel& m, hipStream_t st);
int resnet_encode(rlcf_engine* e, ClipModel& m, const float* images, int n, float* feats, hipStream_t st);
// norm-layer tuning of a ResNet student (resnet.hip): build the train-form weights / flat buffers once; one tuning sample
int engine_gemm_conv3x3(rlcf_engine* e, const float* in, const float* scale2_dev, const float* W, const float* bias, const float* res, int ldr,
                        float* C, int ldc, int n, int H, int Wd, int cin, int cout, int epi, hipStream_t st, float* amax_out,
                        const void* in_pairs = nullptr, void* Cpairs = nullptr, const float* out_scale_dev = nullptr);
int engine_gemm_pairs(rlcf_engine* e, const void* Apairs, int K, const float* alpha_dev, const float* W, const float* bias, const float* res, int ldr,
                      float* C, int ldc, void* Cpairs, const float* out_scale_dev, int M, int N, int epi, hipStream_t st, float* amax_out);
int engine_split_operand(rlcf_engine* e, const float* in, int64_t n, const float* amax_in, void** pairs, const float** scale2, hipStream_t st);
int engine_bn_enable(rlcf_engine* e, hipStream_t st);
int rn_forward_train(rlcf_engine* e, ClipModel& m, const float* images, int n, float* feats, hipStream_t st, int mode_override = -1);
int rn_backward_bn(rlcf_engine* e, ClipModel& m, int n, const float* feats, float* dfeat, float* bn_grad, hipStream_t st, float* vgrad = nullptr);
int engine_tta_sample_bn(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st, bool full = false);
int engine_rn_visual_enable(rlcf_engine* e, hipStream_t st);
int rn_visual_refresh(rlcf_engine* e, hipStream_t st, bool at_checkpoint = false);
// dW[N, K] = dY[T, N]^T X[T, K] (+ db[N] += column sums of dY) through the NT GEMM of the engine's precision (engine.hip)
int engine_wgrad(rlcf_engine* e, const float* dY, int ldy, int N, const float* X, int ldx, int K, int T, float* dW, float* db, hipStream_t st);
// engine.hip services used by resnet.hip
int engine_gemm(rlcf_engine* e, const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldr, float* C,
                int ldc, int M, int N, int K, int epi, hipStream_t st, const float* amax_in = nullptr, float* amax_out = nullptr);
int engine_gemm_presplit(rlcf_engine* e, const float* W, const float* bias, const float* res, int ldr, float* C, int ldc, int M, int N,
                         int K, int epi, const float* alpha_dev, hipStream_t st, float* amax_out = nullptr);
bool engine_has_split(const rlcf_engine* e, const float* W);
int engine_make_split(rlcf_engine* e, ClipModel& m, const float* w, size_t numel, hipStream_t st);

// engine internals used by api.hip
int engine_finalize(rlcf_engine* e, int which, hipStream_t st);
int engine_set_class_bank(rlcf_engine* e, const int32_t* tokens, int C, int n_ctx, const float* ctx_init, int text_mode, hipStream_t st,
                          const int32_t* student_tokens = nullptr, const int32_t* ctx_pos = nullptr);
int engine_encode_image(rlcf_engine* e, int which, const float* images, int n, float* feats, hipStream_t st, int in_res = 0);
int engine_text_features(rlcf_engine* e, int which, const float* ctx, float* txt, hipStream_t st);
int engine_logits(rlcf_engine* e, const float* img, int n, const float* txt, int C, float* logits, hipStream_t st);
int engine_text_backward_dense(rlcf_engine* e, const float* ctx, const float* img, int n, const float* dlogits, float* dctx, hipStream_t st);
int engine_tta_sample(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st);
int engine_tta_sample_ln(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st);
int engine_tta_sample_visual(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st);
int engine_visual_enable(rlcf_engine* e, hipStream_t st);
int engine_text_enable(rlcf_engine* e, hipStream_t st);
int engine_text_reset(rlcf_engine* e, hipStream_t st, bool force);
int engine_set_image_bank(rlcf_engine* e, const float* student_feats, const float* reward_feats, int n, hipStream_t st);
int engine_tta_retrieval_text(rlcf_engine* e, const int32_t* tokens, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st);
int engine_visual_refresh(rlcf_engine* e, hipStream_t st, bool at_checkpoint = false);   // at_checkpoint: the live weights ARE the checkpoint's (reset of a pristine initial state)
int engine_tta_batch_ln(rlcf_engine* e, const float* views, int count, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                        hipStream_t st);
int engine_tta_batch(rlcf_engine* e, const float* views, int count, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                     hipStream_t st);
