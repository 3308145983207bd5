// The small, latency-bound kernels of the RLCF step: per-view entropy + confidence selection
// (TPT/tpt_cls_rl.py:32-35), top-K sampling + CLIPScore + baseline + reward-weighted CE and its
// gradient (TPT/tpt_cls_rl.py:63-74, TPT/clip_reward.py:111-128,152-165), AdamW (TPT/tpt_cls_rl.py:78,120),
// top-5 of the final logits (TPT/utils/tools.py:84-98).  Wave-shuffle reductions, f32 throughout.
#include "kernels.h"

#define TTA_THREADS 256
#define MAX_K 32                       // retrieval scripts sample 12 (text->image) and 20 (image->text) candidates

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < TTA_THREADS / 64; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
    for (int i = 1; i < TTA_THREADS / 64; ++i) s = fmaxf(s, red[i]);
    return s;
}
// Total order used by every selection here: NaN ranks ABOVE +inf (torch.topk / torch.argsort treat NaN as the largest value), so a
// non-finite logits row still yields in-range, distinct indices (the GradScaler guard then skips the step: launch_grad_nonfinite).
__device__ __forceinline__ bool ord_gt(float a, float b) { return a > b || (a != a && b == b); }
__device__ __forceinline__ bool ord_eq(float a, float b) { return a == b || (a != a && b != b); }
// max(x, 0) that lets a NaN through, like torch.maximum(similarity, zeros) (clip_reward.py:126)
__device__ __forceinline__ float relu_nan(float x) { return x > 0.f ? x : (x == x ? 0.f : x); }
// argmax over (value, index) with lowest index winning ties
__device__ __forceinline__ void block_argmax(float& v, int& i, float* redv, int* redi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o);
        int oi = __shfl_xor(i, o);
        if (ord_gt(ov, v) || (ord_eq(ov, v) && oi < i)) { v = ov; i = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = v; redi[threadIdx.x >> 6] = i; }
    __syncthreads();
    v = redv[0]; i = redi[0];
    for (int w = 1; w < TTA_THREADS / 64; ++w)
        if (ord_gt(redv[w], v) || (ord_eq(redv[w], v) && redi[w] < i)) { v = redv[w]; i = redi[w]; }
}

// ---------------------------------------------------------------- entropy per row
__global__ __launch_bounds__(TTA_THREADS) void row_entropy_kernel(const float* __restrict__ logits, int C, float* __restrict__ entropy) {
    __shared__ float red[TTA_THREADS / 64];
    const float* x = logits + (size_t)blockIdx.x * C;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += TTA_THREADS) mx = fmaxf(mx, x[c]);
    mx = block_max(mx, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += TTA_THREADS) s += expf(x[c] - mx);
    s = block_sum(s, red);
    const float lse = mx + logf(s);
    float hsum = 0.f;
    for (int c = threadIdx.x; c < C; c += TTA_THREADS) { float lp = x[c] - lse; hsum += expf(lp) * lp; }
    hsum = block_sum(hsum, red);
    if (threadIdx.x == 0) entropy[blockIdx.x] = -hsum;
}
// rank-select the n_sel lowest entropies, ascending (argsort semantics; ties by index)
__global__ void select_lowest_kernel(const float* __restrict__ entropy, int n, int n_sel, int32_t* __restrict__ idx) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float e = entropy[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float f = entropy[j];
            rank += (ord_gt(e, f) || (ord_eq(f, e) && j < i)) ? 1 : 0;          // NaN entropies sort last, as in torch.argsort
        }
        if (rank < n_sel) idx[rank] = i;
    }
}
// B samples of n views each: idx[b*n_sel + r] = b*n + (view with the r-th lowest entropy of sample b)  (global row ids)
__global__ void select_lowest_batched_kernel(const float* __restrict__ entropy, int n, int n_sel, int32_t* __restrict__ idx) {
    const float* ent = entropy + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float e = ent[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float f = ent[j];
            rank += (ord_gt(e, f) || (ord_eq(f, e) && j < i)) ? 1 : 0;          // NaN entropies sort last, as in torch.argsort
        }
        if (rank < n_sel) idx[blockIdx.x * n_sel + rank] = blockIdx.x * n + i;
    }
}
__global__ void iota_kernel(int32_t* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
int launch_iota(int32_t* p, int n, hipStream_t st) {          // p[i] = i: "every row selected, in order" (retrieval: no view selection)
    RLCF_ARG_CHECK(p && n > 0);
    iota_kernel<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(p, n);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_entropy_select_batched(const float* logits, int B, int n, int C, int n_sel, float* entropy, int32_t* idx_global, hipStream_t st) {
    RLCF_ARG_CHECK(B > 0 && n > 0 && C > 0 && n_sel > 0 && n_sel <= n);
    row_entropy_kernel<<<dim3(B * n), dim3(TTA_THREADS), 0, st>>>(logits, C, entropy);
    RLCF_LAUNCH_CHECK();
    select_lowest_batched_kernel<<<dim3(B), dim3(256), 0, st>>>(entropy, n, n_sel, idx_global);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_entropy_select(const float* logits, int n, int C, int n_sel, float* entropy, int32_t* idx, hipStream_t st) {
    RLCF_ARG_CHECK(n > 0 && C > 0 && n_sel >= 0 && n_sel <= n);
    row_entropy_kernel<<<dim3(n), dim3(TTA_THREADS), 0, st>>>(logits, C, entropy);
    RLCF_LAUNCH_CHECK();
    if (n_sel > 0) {
        select_lowest_kernel<<<dim3(1), dim3(256), 0, st>>>(entropy, n, n_sel, idx);
        RLCF_LAUNCH_CHECK();
    }
    return RLCF_OK;
}

// ---------------------------------------------------------------- reward loss, stage A (one block per selected row)
// top-K classes, log-sum-exp, CE per sampled class, CLIPScore per sampled class.
// stats[i] = {lse, ce[K], score[K]} packed as 1 + 2*MAX_K floats.
#define STAT_LD (1 + 2 * MAX_K)
__global__ __launch_bounds__(TTA_THREADS) void reward_stage_a_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ sel,
                                                                     int C, int K, RewardBank bank, float weight,
                                                                     int32_t* __restrict__ topk_idx, float* __restrict__ stats) {
    __shared__ float red[TTA_THREADS / 64];
    __shared__ float redv[TTA_THREADS / 64];
    __shared__ int redi[TTA_THREADS / 64];
    __shared__ int chosen[MAX_K];
    const int i = blockIdx.x;
    const float* x = logits + (size_t)(sel ? sel[i] : i) * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += TTA_THREADS) mx = fmaxf(mx, x[c]);
    mx = block_max(mx, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += TTA_THREADS) s += expf(x[c] - mx);
    s = block_sum(s, red);
    const float lse = mx + logf(s);
    for (int k = 0; k < K; ++k) {                      // torch.topk order: descending value
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int c = threadIdx.x; c < C; c += TTA_THREADS) {
            bool used = false;
            for (int p = 0; p < k; ++p) used |= (chosen[p] == c);
            const float v = x[c];
            if (!used && (ord_gt(v, bv) || (ord_eq(v, bv) && c < bi))) { bv = v; bi = c; }
        }
        block_argmax(bv, bi, redv, redi);
        if (threadIdx.x == 0) {
            chosen[k] = bi;
            topk_idx[i * K + k] = bi;
            stats[i * STAT_LD + 1 + k] = lse - bv;     // cross entropy of class bi
        }
        __syncthreads();
    }
    for (int k = 0; k < K; ++k) {                      // CLIPScore: w * <class_feat[idx], img_i>, clamped at 0, mixed over the reward models
        float score = 0.f;
        for (int m = 0; m < bank.n; ++m) {
            const int Dr = bank.Dr[m];
            const float* t = bank.class_feat[m] + (size_t)chosen[k] * Dr;
            const float* im = bank.reward_img[m] + (size_t)i * Dr;
            float d = 0.f;
            for (int c = threadIdx.x; c < Dr; c += TTA_THREADS) d += t[c] * im[c];
            d = block_sum(d, red);
            score += bank.mix[m] * relu_nan(weight * d);
        }
        if (threadIdx.x == 0) stats[i * STAT_LD + 1 + MAX_K + k] = bank.post_div == 1.f ? score : score / bank.post_div;
    }
    if (threadIdx.x == 0) stats[i * STAT_LD] = lse;
}

// stage B (one block per selected row): rewards for every (row, k), loss, dense dlogits of the row.
__global__ __launch_bounds__(TTA_THREADS) void reward_stage_b_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ sel,
                                                                     int n_sel, int C, int K, int flags, float min_entropy_w,
                                                                     const int32_t* __restrict__ topk_idx, const float* __restrict__ stats,
                                                                     float* __restrict__ clip_score, float* __restrict__ rewards,
                                                                     float* __restrict__ loss, float* __restrict__ dlogits) {
    // rows are grouped per test sample: n_sel consecutive rows form one group (gridDim.x = groups * n_sel)
    extern __shared__ float avg[];                      // [C] log of the view-averaged probability (min-entropy only)
    __shared__ float red[TTA_THREADS / 64];
    __shared__ float r_row[MAX_K];
    __shared__ float r_all_sum;
    const int grp0 = (blockIdx.x / n_sel) * n_sel;      // first row of this row's group
    const int i = blockIdx.x;
    const bool first = (i == grp0);
    stats += (size_t)grp0 * STAT_LD;                    // group-relative views of the per-row tables
    if (clip_score) clip_score += (size_t)grp0 * K;
    if (rewards) rewards += (size_t)grp0 * K;
    if (loss) loss += blockIdx.x / n_sel;
    const int il = i - grp0;                            // row within the group
    const int total = n_sel * K;
    // ---- rewards_post_process (clip_reward.py:152-165); every block recomputes the tiny table
    float batch_mean = 0.f, batch_std = 1.f;
    const bool process = (flags & RLCF_F_REWARD_PROCESS) != 0;
    const bool amplify = (flags & RLCF_F_AMPLIFY) != 0;
    const bool batch = (flags & RLCF_F_PROCESS_BATCH) != 0;
    const int grp = batch ? total : K;                  // size of the last dimension the baseline runs over
    if (batch) {
        float s = 0.f;
        for (int e = 0; e < total; ++e) s += stats[(e / K) * STAT_LD + 1 + MAX_K + e % K];
        batch_mean = s / total;
        float q = 0.f;
        for (int e = 0; e < total; ++e) { float d = stats[(e / K) * STAT_LD + 1 + MAX_K + e % K] - batch_mean; q += d * d; }
        batch_std = sqrtf(q / (total - 1)) + 1e-5f;
    }
    auto reward_of = [&](int row, int k) -> float {
        const float sc = stats[row * STAT_LD + 1 + MAX_K + k];
        if (!(process && grp > 1)) return sc;
        float mean = batch_mean, sd = batch_std;
        if (!batch) {
            float s = 0.f;
            for (int kk = 0; kk < K; ++kk) s += stats[row * STAT_LD + 1 + MAX_K + kk];
            mean = s / K;
            if (amplify) {
                float q = 0.f;
                for (int kk = 0; kk < K; ++kk) { float d = stats[row * STAT_LD + 1 + MAX_K + kk] - mean; q += d * d; }
                sd = sqrtf(q / (K - 1)) + 1e-5f;
            }
        }
        return amplify ? (sc - mean) / sd : (sc - mean);
    };
    if (threadIdx.x < K) r_row[threadIdx.x] = reward_of(il, threadIdx.x);
    __syncthreads();
    float rsum = 0.f;
    for (int k = 0; k < K; ++k) rsum += r_row[k];
    if (first && threadIdx.x == 0) {
        float l = 0.f;
        for (int e = 0; e < total; ++e) {
            const float r = reward_of(e / K, e % K);
            if (rewards) rewards[e] = r;
            if (clip_score) clip_score[e] = stats[(e / K) * STAT_LD + 1 + MAX_K + e % K];
            l += r * stats[(e / K) * STAT_LD + 1 + e % K];
        }
        r_all_sum = l / total;
    }
    // ---- optional min-entropy regulariser (tpt_cls_rl.py:38-44,73-74)
    const float* x = logits + (size_t)(sel ? sel[i] : i) * ld;
    const float lse = stats[il * STAT_LD];
    float pa_dot = 0.f, hreg = 0.f;
    const bool minent = (flags & RLCF_F_MIN_ENTROPY) != 0;
    if (minent) {
        for (int c = threadIdx.x; c < C; c += TTA_THREADS) {
            float mx = -INFINITY;
            for (int j = 0; j < n_sel; ++j) mx = fmaxf(mx, logits[(size_t)(sel ? sel[grp0 + j] : grp0 + j) * ld + c] - stats[j * STAT_LD]);
            float s = 0.f;
            for (int j = 0; j < n_sel; ++j) s += expf(logits[(size_t)(sel ? sel[grp0 + j] : grp0 + j) * ld + c] - stats[j * STAT_LD] - mx);
            float a = mx + logf(s) - logf((float)n_sel);
            a = fmaxf(a, -3.4028234663852886e38f);
            avg[c] = a;
            pa_dot += expf(x[c] - lse) * a;
            hreg += a * expf(a);
        }
        pa_dot = block_sum(pa_dot, red);
        hreg = block_sum(hreg, red);
    }
    __syncthreads();
    if (first && threadIdx.x == 0 && loss) loss[0] = r_all_sum + (minent ? -min_entropy_w * hreg : 0.f);
    // ---- dlogits row: d/dx of mean_{i,k} r_ik * CE(x_i, idx_ik)  (+ w * d avg_entropy)
    const float inv_total = 1.0f / total;
    for (int c = threadIdx.x; c < C; c += TTA_THREADS) {
        const float p = expf(x[c] - lse);
        float g = p * rsum;
        for (int k = 0; k < K; ++k)
            if (topk_idx[i * K + k] == c) g -= r_row[k];
        g *= inv_total;
        if (minent) g += min_entropy_w * (p / n_sel) * (pa_dot - avg[c]);
        dlogits[(size_t)i * C + c] = g;
    }
}

size_t reward_loss_stats_floats(int rows) { return (size_t)(rows > 0 ? rows : 1) * STAT_LD; }
// groups test samples of n_sel rows each (rows = groups*n_sel); loss[groups]; everything else row-major over all rows
int launch_reward_loss_bank(const float* logits, int ld_logits, const int32_t* sel, int groups, int n_sel, int C, int K,
                            const RewardBank& bank, float clipscore_weight, int flags,
                            float min_entropy_w, int32_t* topk_idx, float* clip_score, float* rewards, float* loss,
                            float* dlogits, float* stats, hipStream_t st) {
    RLCF_ARG_CHECK(groups > 0 && n_sel > 0 && C > 0 && K > 0 && K <= MAX_K && K <= C && dlogits && topk_idx && stats);
    RLCF_ARG_CHECK(bank.n >= 1 && bank.n <= RLCF_MAX_REWARDS);
    for (int m = 0; m < bank.n; ++m) RLCF_ARG_CHECK(bank.class_feat[m] && bank.reward_img[m] && bank.Dr[m] > 0);
    const int rows = groups * n_sel;
    reward_stage_a_kernel<<<dim3(rows), dim3(TTA_THREADS), 0, st>>>(logits, ld_logits, sel, C, K, bank, clipscore_weight, topk_idx, stats);
    RLCF_LAUNCH_CHECK();
    const size_t sh = (flags & RLCF_F_MIN_ENTROPY) ? (size_t)C * sizeof(float) : 0;
    if (sh > 160 * 1024 - 256) { rlcf_set_error("reward_loss: min-entropy regulariser over %d classes does not fit the 160 KB LDS", C); return RLCF_ERR_ARG; }
    if (sh > 48 * 1024) {                                // large banks (retrieval: 5k-25k captions)
        int rc_ = rlcf_func_lds((const void*)reward_stage_b_kernel, sh);
        if (rc_ != RLCF_OK) return rc_;
    }
    reward_stage_b_kernel<<<dim3(rows), dim3(TTA_THREADS), sh, st>>>(logits, ld_logits, sel, n_sel, C, K, flags, min_entropy_w,
                                                                     topk_idx, stats, clip_score, rewards, loss, dlogits);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// top-K classes of every row (torch.topk order) and their cross entropies only: stage A without a reward bank.  The scores it leaves
// in `stats` are zeros; launch_reward_loss_bank on the same rows later recomputes the identical indices and fills the scores in.
int launch_topk_rows(const float* logits, int ld_logits, int rows, int C, int K, int32_t* topk_idx, float* stats, hipStream_t st) {
    RLCF_ARG_CHECK(rows > 0 && C > 0 && K > 0 && K <= MAX_K && K <= C && topk_idx && stats);
    RewardBank none{};
    none.post_div = 1.f;
    reward_stage_a_kernel<<<dim3(rows), dim3(TTA_THREADS), 0, st>>>(logits, ld_logits, nullptr, C, K, none, 0.f, topk_idx, stats);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_reward_loss_grouped(const float* logits, int ld_logits, const int32_t* sel, int groups, int n_sel, int C, int K,
                               const float* class_feat, const float* reward_img, int Dr, float clipscore_weight, int flags,
                               float min_entropy_w, int32_t* topk_idx, float* clip_score, float* rewards, float* loss,
                               float* dlogits, float* stats, hipStream_t st) {
    RewardBank bank{};
    bank.n = 1; bank.class_feat[0] = class_feat; bank.reward_img[0] = reward_img; bank.Dr[0] = Dr; bank.mix[0] = 1.f; bank.post_div = 1.f;
    return launch_reward_loss_bank(logits, ld_logits, sel, groups, n_sel, C, K, bank, clipscore_weight, flags, min_entropy_w, topk_idx,
                                   clip_score, rewards, loss, dlogits, stats, st);
}
int launch_reward_loss(const float* logits, int ld_logits, const int32_t* sel, int n_sel, int C, int K,
                       const float* class_feat, const float* reward_img, int Dr, float clipscore_weight, int flags,
                       float min_entropy_w, int32_t* topk_idx, float* clip_score, float* rewards, float* loss,
                       float* dlogits, float* stats, hipStream_t st) {
    return launch_reward_loss_grouped(logits, ld_logits, sel, 1, n_sel, C, K, class_feat, reward_img, Dr, clipscore_weight, flags,
                                      min_entropy_w, topk_idx, clip_score, rewards, loss, dlogits, stats, st);
}

// ---------------------------------------------------------------- AdamW
// GradScaler semantics of the reference (torch.cuda.amp.GradScaler.step at TPT/tpt_cls_rl.py:76-79): the optimizer step of a
// sample whose gradient holds an inf / NaN is SKIPPED (parameters and moments untouched).  nonfinite_kernel raises flag[group]
// for every parameter group (one group per test sample in the batched calls) with such a gradient; adamw_kernel leaves the
// elements of a flagged group alone.  Device-side: no host synchronisation, the step count handed to the kernel stays the host's.
__global__ void nonfinite_kernel(const float* __restrict__ g, int64_t per_group, int32_t* __restrict__ flag) {
    const float* gg = g + (int64_t)blockIdx.y * per_group;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_group; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = gg[i];
        bad |= !(fabsf(v) <= 3.4028234663852886e38f);             // inf or NaN
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag + blockIdx.y, 1);
}
int launch_grad_nonfinite(const float* g, int64_t per_group, int groups, int32_t* flag, hipStream_t st, bool accumulate) {
    RLCF_ARG_CHECK(g && flag && per_group > 0 && groups > 0 && groups <= 65535);
    if (!accumulate) RLCF_HIP_CHECK(hipMemsetAsync(flag, 0, (size_t)groups * sizeof(int32_t), st));
    int bx = (int)((per_group + 255) / 256);
    if (bx > 256) bx = 256;
    nonfinite_kernel<<<dim3(bx, groups), dim3(256), 0, st>>>(g, per_group, flag);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2,
                             const int32_t* __restrict__ skip, int64_t per_group) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (skip && skip[i / per_group]) continue;
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2,
                 float eps, float wd, hipStream_t st, const int32_t* skip, int64_t per_group) {
    RLCF_ARG_CHECK(n > 0 && step >= 1 && (!skip || per_group > 0));
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    adamw_kernel<<<dim3(blocks), dim3(256), 0, st>>>(p, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1, (float)sqrt(bc2), skip,
                                                     per_group > 0 ? per_group : n);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- cross-sample momentum update (CLIPCLS_TTA.momentum_update_model)
// mom = m*mom + (1-m)*cur; if apply: init = (1-w)*clip + w*mom   (custom_clip.py:460-475; float32, products rounded separately
// as torch's elementwise ops do)
__global__ void momentum_update_kernel(float* __restrict__ mom, const float* __restrict__ cur, const float* __restrict__ clip,
                                       float* __restrict__ init, int64_t n, float m, float one_minus_m, float w, float one_minus_w, int apply) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = __fadd_rn(__fmul_rn(m, mom[i]), __fmul_rn(one_minus_m, cur[i]));
        mom[i] = v;
        if (apply) init[i] = __fadd_rn(__fmul_rn(one_minus_w, clip[i]), __fmul_rn(w, v));
    }
}
int launch_momentum_update(float* mom, const float* cur, const float* clip, float* init, int64_t n, double momentum, double update_w, int apply,
                           hipStream_t st) {
    RLCF_ARG_CHECK(mom && cur && clip && init && n > 0);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    momentum_update_kernel<<<dim3(blocks), dim3(256), 0, st>>>(mom, cur, clip, init, n, (float)momentum, (float)(1.0 - momentum), (float)update_w,
                                                               (float)(1.0 - update_w), apply);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- top-5 of one logits row
__global__ __launch_bounds__(TTA_THREADS) void top5_kernel(const float* __restrict__ x, int C, int32_t* __restrict__ top5) {
    x += (size_t)blockIdx.x * C;
    top5 += blockIdx.x * 5;
    __shared__ float redv[TTA_THREADS / 64];
    __shared__ int redi[TTA_THREADS / 64];
    __shared__ int chosen[5];
    const int n = C < 5 ? C : 5;
    for (int k = 0; k < n; ++k) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int c = threadIdx.x; c < C; c += TTA_THREADS) {
            bool used = false;
            for (int p = 0; p < k; ++p) used |= (chosen[p] == c);
            const float v = x[c];
            if (!used && (ord_gt(v, bv) || (ord_eq(v, bv) && c < bi))) { bv = v; bi = c; }
        }
        block_argmax(bv, bi, redv, redi);
        if (threadIdx.x == 0) { chosen[k] = bi; top5[k] = bi; }
        __syncthreads();
    }
}
// per-sample final logits: out[b, c] = scale * <img[b * img_stride_rows], txt[b, c]>   (one wave per class)
__global__ __launch_bounds__(256) void final_logits_batched_kernel(const float* __restrict__ img, int img_row_stride, const float* __restrict__ txt,
                                                                   int C, int D, float scale, float* __restrict__ out) {
    const int b = blockIdx.y, c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    const float* im = img + (size_t)b * img_row_stride * D;
    const float* t = txt + ((size_t)b * C + c) * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += im[d] * t[d];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)b * C + c] = scale * s;
}
int launch_final_logits_batched(const float* img, int img_row_stride, const float* txt, int B, int C, int D, float scale, float* out,
                                hipStream_t st) {
    final_logits_batched_kernel<<<dim3((C + 3) / 4, B), dim3(256), 0, st>>>(img, img_row_stride, txt, C, D, scale, out);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// per-sample logits of a few rows: out[b*rows + i, c] = scale * <img[b*rows + i], txt[b, c]>   (one wave per class)
__global__ __launch_bounds__(256) void group_logits_kernel(const float* __restrict__ img, int rows, const float* __restrict__ txt, int C, int D,
                                                           float scale, float* __restrict__ out) {
    const int gr = blockIdx.y, b = gr / rows, c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    const float* im = img + (size_t)gr * D;
    const float* t = txt + ((size_t)b * C + c) * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += im[d] * t[d];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)gr * C + c] = scale * s;
}
int launch_group_logits(const float* img, int rows_per_group, const float* txt, int B, int C, int D, float scale, float* out, hipStream_t st) {
    RLCF_ARG_CHECK(B > 0 && rows_per_group > 0 && B * rows_per_group <= 65535);
    group_logits_kernel<<<dim3((C + 3) / 4, B * rows_per_group), dim3(256), 0, st>>>(img, rows_per_group, txt, C, D, scale, out);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_top5_batched(const float* logits, int B, int C, int32_t* top5, hipStream_t st) {
    top5_kernel<<<dim3(B), dim3(TTA_THREADS), 0, st>>>(logits, C, top5);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_top5(const float* logits, int C, int32_t* top5, hipStream_t st) {
    top5_kernel<<<dim3(1), dim3(TTA_THREADS), 0, st>>>(logits, C, top5);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}


// ---- stand-alone conveniences of the harness mirror (round 5: they were torch expressions) ----------------------------------------
// avg_entropy (TPT/tpt_cls_rl.py:38-44): H of the mean over the n views of softmax(logits): logp = x - lse(x) per row,
// avg_c = lse_n(logp[n, c]) - log n (clamped at the smallest float, as torch.clamp(min = finfo.min)), out = -sum_c avg_c exp(avg_c).
// One workgroup; row log-sum-exps first (LDS, n <= 4096), then the classes.
__global__ __launch_bounds__(TTA_THREADS) void avg_entropy_kernel(const float* __restrict__ x, int n, int C, float* __restrict__ out) {
    extern __shared__ float lse[];                    // [n]
    __shared__ float red[TTA_THREADS / 64];
    // a NaN or +inf logit makes the reference's result NaN (x - logsumexp(x) is NaN there and torch.clamp / logsumexp carry it on);
    // fmaxf would drop it silently, so such inputs are counted and answered with NaN
    float bad = 0.f;
    for (int r = 0; r < n; ++r) {
        const float* row = x + (size_t)r * C;
        float m = -INFINITY;
        for (int c = threadIdx.x; c < C; c += TTA_THREADS) { const float v = row[c]; bad += (v != v || v == INFINITY) ? 1.f : 0.f; m = fmaxf(m, v); }
        m = block_max(m, red);
        float sum = 0.f;
        for (int c = threadIdx.x; c < C; c += TTA_THREADS) sum += expf(row[c] - m);
        sum = block_sum(sum, red);
        if (threadIdx.x == 0) lse[r] = m + logf(sum);
        __syncthreads();
    }
    float acc = 0.f;
    const float logn = logf((float)n);
    for (int c = threadIdx.x; c < C; c += TTA_THREADS) {
        float m = -INFINITY;
        for (int r = 0; r < n; ++r) m = fmaxf(m, x[(size_t)r * C + c] - lse[r]);
        float sum = 0.f;
        for (int r = 0; r < n; ++r) sum += expf(x[(size_t)r * C + c] - lse[r] - m);
        float a = m + logf(sum) - logn;
        a = fmaxf(a, -3.402823466e+38f);          // (torch.clamp(min = finfo(float32).min))
        acc -= a * expf(a);
    }
    acc = block_sum(acc, red);
    bad = block_sum(bad, red);
    if (threadIdx.x == 0) out[0] = bad > 0.f ? NAN : acc;
}
int launch_avg_entropy(const float* logits, int n, int C, float* out, hipStream_t st) {
    RLCF_ARG_CHECK(logits && out && n > 0 && n <= 4096 && C > 0);
    avg_entropy_kernel<<<dim3(1), dim3(TTA_THREADS), (size_t)n * sizeof(float), st>>>(logits, n, C, out);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// accuracy (TPT/utils/tools.py:84-98) for topk = (1, 5): out[0] / out[1] = 100 / B * #(target in the top 1 / top 5 of its row), from the
// rows' top-5 indices (top5_kernel: ties go to the lower index)
__global__ void accuracy_kernel(const int32_t* __restrict__ top5, const int64_t* __restrict__ target, int B, int C, float* __restrict__ out) {
    __shared__ int h1, h5;
    if (threadIdx.x == 0) { h1 = 0; h5 = 0; }
    __syncthreads();
    const int nk = C < 5 ? C : 5;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int t = (int)target[b];
        int a1 = top5[b * 5] == t, a5 = 0;
        for (int k = 0; k < nk; ++k) a5 |= (top5[b * 5 + k] == t);
        if (a1) atomicAdd(&h1, 1);
        if (a5) atomicAdd(&h5, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = 100.0f * h1 / B; out[1] = 100.0f * h5 / B; }
}
int launch_accuracy(const float* logits, const int64_t* target, int B, int C, int32_t* top5_scratch, float* out, hipStream_t st) {
    RLCF_ARG_CHECK(logits && target && top5_scratch && out && B > 0 && C > 0);
    top5_kernel<<<dim3(B), dim3(TTA_THREADS), 0, st>>>(logits, C, top5_scratch);
    RLCF_LAUNCH_CHECK();
    accuracy_kernel<<<dim3(1), dim3(256), 0, st>>>(top5_scratch, target, B, C, out);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// Hit counts of a harness loop from the engine's own top-5 indices (rlcf_tta_batch / rlcf_lanes_submit outputs): out[0] += #(target ==
// top5[b, 0]), out[1] += #(target in top5[b, :]) — what the reference accumulates per image through accuracy() + AverageMeter
// (TPT/tpt_cls_rl.py:265-268), as counts.  Accumulating: the loop calls it once per pass / per group of samples on one float[2].
__global__ void top5_hits_kernel(const int32_t* __restrict__ top5, const int64_t* __restrict__ target, int B, float* __restrict__ out) {
    __shared__ int h1, h5;
    if (threadIdx.x == 0) { h1 = 0; h5 = 0; }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int t = (int)target[b];
        int a5 = 0;
        for (int k = 0; k < 5; ++k) a5 |= (top5[b * 5 + k] == t);
        if (top5[b * 5] == t) atomicAdd(&h1, 1);
        if (a5) atomicAdd(&h5, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) { out[0] += (float)h1; out[1] += (float)h5; }
}
int launch_top5_hits(const int32_t* top5, const int64_t* target, int B, float* out, hipStream_t st) {
    RLCF_ARG_CHECK(top5 && target && out && B > 0);
    top5_hits_kernel<<<dim3(1), dim3(256), 0, st>>>(top5, target, B, out);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
