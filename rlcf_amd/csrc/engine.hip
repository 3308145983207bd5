// Host runtime of the RLCF hot path (see engine.h).  Every function cites the reference lines
// whose behaviour it reproduces; the arithmetic itself lives in the kernels it sequences.
#include "engine.h"
#include <algorithm>
#include <cmath>
#include <cstring>

#define TRY(x) do { int rc_ = (x); if (rc_ != RLCF_OK) return rc_; } while (0)

// int(N * selection_p) of select_confident_samples (tpt_cls_rl.py:34) is a DOUBLE product in Python; the float field of the
// argument block can round the other way (N=10, p=0.7 -> 6), so the host passes its own value in a->n_sel (0: derive it here).
static inline int n_selected(const rlcf_tta_args* a, int N) {
    return a->n_sel > 0 ? a->n_sel : (int)((double)N * (double)a->selection_p);
}

int DevBuf::ensure(size_t n) {
    if (n <= bytes && p) return RLCF_OK;
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
    if (n == 0) return RLCF_OK;
    hipError_t err = hipMalloc(&p, n);
    if (err != hipSuccess) { rlcf_set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(err)); p = nullptr; return RLCF_ERR_NOMEM; }
    bytes = n;
    return RLCF_OK;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }

// ------------------------------------------------------------------ profiling of GEMM launches
GemmProfile g_prof;
static int prof_begin(hipStream_t st, double flops, int M = 0, int N = 0, int K = 0) {
    if (!g_prof.enabled) return -1;
    if ((int)g_prof.ev.size() < 2 * (g_prof.n + 1)) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        g_prof.ev.push_back(a); g_prof.ev.push_back(b);
    }
    g_prof.flops.resize(g_prof.n + 1);
    g_prof.kind.resize(g_prof.n + 1);
    g_prof.dims.resize(3 * (g_prof.n + 1));
    g_prof.dims[3 * g_prof.n] = M; g_prof.dims[3 * g_prof.n + 1] = N; g_prof.dims[3 * g_prof.n + 2] = K;
    g_prof.flops[g_prof.n] = flops;
    g_prof.kind[g_prof.n] = 0;
    (void)hipEventRecord(g_prof.ev[2 * g_prof.n], st);
    return g_prof.n;
}
static void prof_end(int slot, hipStream_t st, int kind = 0) {
    if (slot < 0) return;
    g_prof.kind[slot] = kind;
    (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
    g_prof.n = slot + 1;
}

// Split-f16 operands of the engine are INTERLEAVED pairs: per row, each block of 32 K-columns is stored as 32 hi halves followed by
// its 32 lo halves (row stride 2K halves), so that one 128-B LDS-DMA segment brings both parts of a k-tile (a 64-B segment per
// part reaches 25.8 B/clk/CU, a 128-B one 45.5: profiles/r1_gemm_sq_counters.txt).  The lo part of a pair therefore starts 64 B after hi.
static inline float* ws_ptr(rlcf_engine* e) { return (e->ws_sel ? e->gemm_ws2 : e->gemm_ws).as<float>(); }
static inline size_t ws_bytes(rlcf_engine* e) { return (e->ws_sel ? e->gemm_ws2 : e->gemm_ws).bytes; }
static inline unsigned* ws_epoch(rlcf_engine* e) { return &e->ws_epoch[e->ws_sel ? 1 : 0]; }
// image-tower scratch / A-operand split buffer of the stream whose launches are being enqueued (see ws_sel)
#define IMG_BUF(e, name) ((e)->ws_sel ? (e)->side_img.name : (e)->name)
static inline void* a_ptr(rlcf_engine* e) { return e->ws_sel ? e->a_hi2.p : e->a_hi.p; }
// scratch of the bit-reproducible parameter-gradient reductions (rowops.hip: per-wave partial sums added in a fixed order)
#define PARTS_WS(e) (e)->parts_ws.as<float>(), ((e)->parts_ws.p ? RLCF_PARTS_WS_FLOATS : (size_t)0)
static inline size_t a_cap(const rlcf_engine* e) { return e->ws_sel ? e->a_split2_elems : e->a_split_elems; }
static inline void* lo_of(void* hi) { return (char*)hi + 64; }
static inline const void* lo_of(const void* hi) { return (const char*)hi + 64; }

// C = epi(alpha A W^T + b) (+res): dispatch on the engine precision
// a_scale: exact power of two applied to A before it is split into f16 pairs (undone in alpha); forward activations use 1.
// dyn_scale: the power of two is found on the device from max|A| (operands without a known range: ResNet activations, and the
// gradients of the backward passes, 1e-8..1e-1 depending on checkpoint, loss scale and depth, which must be lifted out of f16's
// subnormal range without overflowing it).  amax_in: max|A| already known (written by the producing GEMM's epilogue).
static int gemm(rlcf_engine* e, const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldr,
                const float* aux, int ldaux, float* C, int ldc, int M, int N, int K, float alpha, int epi, hipStream_t st,
                float a_scale = 1.0f, bool dyn_scale = false, const float* amax_in = nullptr, unsigned int* amax_out = nullptr) {
    GemmArgs g{};
    g.amax_out = amax_out;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.residual = res; g.ldr = ldr; g.aux = aux; g.ldaux = ldaux;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.epilogue = epi;
    e->last_flops += 2.0 * M * N * K;
    static int x3_min_m = -1;                             // RLCF_X3_MIN_M: smallest M that takes the split-f16 kernels (benchmarks)
    if (x3_min_m < 0) { const char* ev = getenv("RLCF_X3_MIN_M"); x3_min_m = ev ? atoi(ev) : 512; }
    if (prec_x3(e) && M > x3_min_m && K % 32 == 0 && lda == K && ldw == K) {
        // split-f16 path: W was split at finalize; A is split here (producers will emit pairs directly)
        const ClipModel::SplitW* sp = nullptr;
        for (auto& m : e->model) { auto it = m.split_of.find(W); if (it != m.split_of.end()) { sp = &it->second; break; } }
        if (sp && (size_t)M * K <= a_cap(e)) {
            const float* alpha_dev = nullptr;
            if (dyn_scale) {       // operand range unknown (un-normalised ResNet activations): power-of-two scale found on the device
                TRY(e->dyn.ensure(3 * sizeof(float)));
                if (amax_in) {             // max|A| was produced by the GEMM that wrote A
                    TRY(launch_dyn_scale_from(amax_in, e->dyn.as<float>() + 1, st));
                    TRY(launch_split_f16x2_dev(A, a_ptr(e), lo_of(a_ptr(e)), (int64_t)M * K, e->dyn.as<float>() + 1, st, 1));
                } else {
                    TRY(launch_split_f16x2_dyn(A, a_ptr(e), lo_of(a_ptr(e)), (int64_t)M * K, e->dyn.as<float>(), st, 1));
                }
                alpha_dev = e->dyn.as<float>() + 2;
            } else {
                TRY(launch_split_f16x2(A, a_ptr(e), lo_of(a_ptr(e)), (int64_t)M * K, st, a_scale, 1));
            }
            const int slot = prof_begin(st, 2.0 * M * N * K, M, N, K);
            if (sp->lo_zero && sp->hi_only) gemm_f16x3_next_packed_w(sp->hi_only);
            int rc = launch_gemm_f16x3(a_ptr(e), lo_of(a_ptr(e)), 2 * K, sp->hi, sp->lo, 2 * K, bias, res, ldr, aux, ldaux, C, ldc, nullptr,
                                       nullptr, 0, M, N, K, alpha * sp->inv_scale / a_scale, epi, st, alpha_dev, amax_out, 0,
                                       ws_ptr(e), ws_bytes(e), sp->lo_zero ? 2 : 0, nullptr, ws_epoch(e));
            prof_end(slot, st, g_last_x3_variant);
            return rc;
        }
    }
    // small M (the one-image path's sparse text passes, ~239 rows): the skinny split-f16 kernel splits A itself, scaled by 1 (forward
    // activations: known range), by the max|A| the producer left behind, or — gradients nobody measured — per wave by the max of its
    // own 32 rows, found in the kernel.  RLCF_SKINNY=0 switches it off (A/B measurements)
    static int skinny = -1;
    if (skinny < 0) { const char* ev = getenv("RLCF_SKINNY"); skinny = ev ? atoi(ev) : 1; }
    if (skinny && prec_x3(e) && !prec_single(e) && M > 32 && a_scale == 1.0f && lda % 4 == 0 && ldw == K && C &&
        gemm_skinny_x3_ok(M, N, K, lda, ldc) && ldr % 4 == 0 && ldaux % 4 == 0) {
        const ClipModel::SplitW* sp = nullptr;
        for (auto& m : e->model) { auto it = m.split_of.find(W); if (it != m.split_of.end()) { sp = &it->second; break; } }
        if (sp && sp->lo == lo_of(sp->hi)) {
            TRY(e->dyn.ensure(3 * sizeof(float)));
            const int slot = prof_begin(st, 2.0 * M * N * K, M, N, K);
            int rc = launch_gemm_skinny_x3(A, lda, sp->hi, bias, res, ldr, aux, ldaux, C, ldc, M, N, K, alpha * sp->inv_scale, epi,
                                           dyn_scale ? amax_in : nullptr, amax_out, ws_ptr(e), ws_bytes(e) - X3_SK_FLAG_BYTES_RESERVED,
                                           e->dyn.as<float>() + 2, st, dyn_scale && !amax_in);
            prof_end(slot, st, g_last_x3_variant);
            return rc;
        }
    }
    const int slot = prof_begin(st, 2.0 * M * N * K, M, N, K);
    int rc = launch_gemm_f32(g, st);
    prof_end(slot, st);
    return rc;
}

int engine_gemm(rlcf_engine* e, const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldr, float* C,
                int ldc, int M, int N, int K, int epi, hipStream_t st, const float* amax_in, float* amax_out) {
    return gemm(e, A, lda, W, ldw, bias, res, ldr, nullptr, 0, C, ldc, M, N, K, 1.f, epi, st, 1.0f, true, amax_in, (unsigned int*)amax_out);
}

// A operand already split into e->a_hi / e->a_lo by the caller (scaled by the device scalar whose inverse is *alpha_dev)
int engine_gemm_presplit(rlcf_engine* e, const float* W, const float* bias, const float* res, int ldr, float* C, int ldc, int M, int N,
                         int K, int epi, const float* alpha_dev, hipStream_t st, float* amax_out) {
    const ClipModel::SplitW* sp = nullptr;
    for (auto& m : e->model) { auto it = m.split_of.find(W); if (it != m.split_of.end()) { sp = &it->second; break; } }
    if (!sp) { rlcf_set_error("engine_gemm_presplit: weight has no split copy"); return RLCF_ERR_STATE; }
    e->last_flops += 2.0 * M * N * K;
    const int slot = prof_begin(st, 2.0 * M * N * K, M, N, K);
    if (sp->lo_zero && sp->hi_only) gemm_f16x3_next_packed_w(sp->hi_only);
    int rc = launch_gemm_f16x3(a_ptr(e), lo_of(a_ptr(e)), 2 * K, sp->hi, sp->lo, 2 * K, bias, res, ldr, nullptr, 0, C, ldc, nullptr, nullptr, 0, M, N, K,
                               sp->inv_scale, epi, st, alpha_dev, (unsigned int*)amax_out, 0, ws_ptr(e), ws_bytes(e), sp->lo_zero ? 2 : 0, nullptr, ws_epoch(e));
    prof_end(slot, st, g_last_x3_variant);
    return rc;
}
// implicit 3x3 convolution: the activation [n*H*W, cin] was split into e->a_hi / a_lo by the caller (interleaved pairs, scaled by
// the device scalar whose inverse is *alpha_dev); W = the folded / raw convolution weight [cout, 9*cin] with a split copy
// C = epi(A W^T + b) (+res) with A already an interleaved operand pair matrix [M, K] scaled by 1 / *alpha_dev; output f32 and / or pairs
// (Cpairs [M, N], scaled by *out_scale_dev)
int engine_gemm_pairs(rlcf_engine* e, const void* Apairs, int K, const float* alpha_dev, const float* W, const float* bias, const float* res, int ldr,
                      float* C, int ldc, void* Cpairs, const float* out_scale_dev, int M, int N, int epi, hipStream_t st, float* amax_out) {
    const ClipModel::SplitW* sp = nullptr;
    for (auto& m : e->model) { auto it = m.split_of.find(W); if (it != m.split_of.end()) { sp = &it->second; break; } }
    if (!sp) { rlcf_set_error("engine_gemm_pairs: weight has no split copy"); return RLCF_ERR_STATE; }
    e->last_flops += 2.0 * M * N * K;
    const int slot = prof_begin(st, 2.0 * M * N * K, M, N, K);
    if (sp->lo_zero && sp->hi_only) gemm_f16x3_next_packed_w(sp->hi_only);
    int rc = launch_gemm_f16x3(Apairs, lo_of(Apairs), 2 * K, sp->hi, sp->lo, 2 * K, bias, res, ldr, nullptr, 0, C, ldc, Cpairs, Cpairs ? lo_of(Cpairs) : nullptr,
                               2 * N, M, N, K, sp->inv_scale, epi, st, alpha_dev, (unsigned int*)amax_out, 1, ws_ptr(e), ws_bytes(e), sp->lo_zero ? 2 : 0, out_scale_dev, ws_epoch(e));
    prof_end(slot, st, g_last_x3_variant);
    return rc;
}
// f32 matrix -> operand pairs in the engine's A scratch, scaled by the power of two found from max|in| (amax_in if known); *scale2 = {s, 1/s}
int engine_split_operand(rlcf_engine* e, const float* in, int64_t n, const float* amax_in, void** pairs, const float** scale2, hipStream_t st) {
    if ((size_t)n > a_cap(e)) { rlcf_set_error("engine_split_operand: operand scratch too small"); return RLCF_ERR_STATE; }
    TRY(e->dyn.ensure(3 * sizeof(float)));
    if (amax_in) TRY(launch_dyn_scale_from(amax_in, e->dyn.as<float>() + 1, st));
    else TRY(launch_dyn_scale(in, n, e->dyn.as<float>(), st));
    TRY(launch_split_f16x2_dev(in, a_ptr(e), lo_of(a_ptr(e)), n, e->dyn.as<float>() + 1, st, 1));
    *pairs = a_ptr(e); *scale2 = e->dyn.as<float>() + 1;
    return RLCF_OK;
}
int engine_gemm_conv3x3(rlcf_engine* e, const float* in, const float* scale2_dev, const float* W, const float* bias, const float* res, int ldr,
                        float* C, int ldc, int n, int H, int Wd, int cin, int cout, int epi, hipStream_t st, float* amax_out,
                        const void* in_pairs, void* Cpairs, const float* out_scale_dev) {
    const ClipModel::SplitW* sp = nullptr;
    for (auto& m : e->model) { auto it = m.split_of.find(W); if (it != m.split_of.end()) { sp = &it->second; break; } }
    if (!sp) { rlcf_set_error("engine_gemm_conv3x3: weight has no split copy"); return RLCF_ERR_STATE; }
    if (!e->zpage.p) { TRY(e->zpage.ensure(4096)); RLCF_HIP_CHECK(hipMemsetAsync(e->zpage.p, 0, 4096, st)); }
    const int M = n * H * Wd;
    if (!in_pairs) {
        if ((size_t)M * cin > a_cap(e)) { rlcf_set_error("engine_gemm_conv3x3: operand scratch too small"); return RLCF_ERR_STATE; }
        TRY(launch_split_f16x2_dev(in, a_ptr(e), lo_of(a_ptr(e)), (int64_t)M * cin, scale2_dev, st, 1));      // scale2_dev = {s, 1/s}
        in_pairs = a_ptr(e);
    }
    const float* alpha_dev = scale2_dev + 1;
    e->last_flops += 2.0 * M * cout * 9.0 * cin;
    const int slot = prof_begin(st, 2.0 * M * cout * 9.0 * cin, M, cout, 9 * cin);
    if (sp->lo_zero && sp->hi_only) gemm_f16x3_next_packed_w(sp->hi_only);
    int rc = launch_gemm_f16x3_conv3x3(in_pairs, n, H, Wd, cin, sp->hi, cout, bias, res, ldr, C, ldc, sp->inv_scale, epi, alpha_dev,
                                       (unsigned int*)amax_out, e->zpage.p, st, Cpairs, out_scale_dev, sp->lo_zero ? 1 : 0);
    prof_end(slot, st, g_last_x3_variant);
    return rc;
}
bool engine_has_split(const rlcf_engine* e, const float* W) {
    for (auto& m : e->model) if (m.split_of.find(W) != m.split_of.end()) return true;
    return false;
}

// pre-split A operand (written by the producing kernel): C f32 and/or a split pair
static const ClipModel::SplitW* split_of(rlcf_engine* e, const float* W) {
    for (auto& m : e->model) { auto it = m.split_of.find(W); if (it != m.split_of.end()) return &it->second; }
    return nullptr;
}
// A (and the optional split output) are interleaved pairs; lda / ldch are given in logical columns.
// RLCF_PREC_F16: A, the weight copy and the optional output are plain f16 matrices instead (one MFMA per product).
static int gemm_pre(rlcf_engine* e, const void* A2, int lda, const float* W, const float* bias, const float* res, int ldr,
                    float* C, int ldc, void* C2, int ldch, int M, int N, int K, int epi, hipStream_t st) {
    e->last_flops += 2.0 * M * N * K;
    if (prec_single(e)) {
        const ClipModel::SplitW* fw = nullptr;
        for (auto& m : e->model) { auto it = m.f16_of.find(W); if (it != m.f16_of.end()) { fw = &it->second; break; } }
        if (!fw || K % 64) { rlcf_set_error("gemm_pre: weight has no plain f16 copy (K = %d)", K); return RLCF_ERR_STATE; }
        const int slot = prof_begin(st, 2.0 * M * N * K, M, N, K);
        int rc = launch_gemm_f16x3(A2, lo_of(A2), lda, fw->hi, lo_of(fw->hi), K, bias, res, ldr, nullptr, 0, C, ldc, C2, nullptr, ldch, M, N, K,
                                   fw->inv_scale, epi, st, nullptr, nullptr, 0, ws_ptr(e), ws_bytes(e), 1, nullptr, ws_epoch(e));
        prof_end(slot, st, g_last_x3_variant);
        return rc;
    }
    const ClipModel::SplitW* sp = split_of(e, W);
    if (!sp) { rlcf_set_error("gemm_pre: weight has no split copy"); return RLCF_ERR_STATE; }
    const int slot = prof_begin(st, 2.0 * M * N * K, M, N, K);
    if (sp->lo_zero && sp->hi_only) gemm_f16x3_next_packed_w(sp->hi_only);
    int rc = launch_gemm_f16x3(A2, lo_of(A2), 2 * lda, sp->hi, sp->lo, 2 * K, bias, res, ldr, nullptr, 0, C, ldc, C2, C2 ? lo_of(C2) : nullptr,
                               2 * ldch, M, N, K, sp->inv_scale, epi, st, nullptr, nullptr, 1, ws_ptr(e), ws_bytes(e), sp->lo_zero ? 2 : 0, nullptr, ws_epoch(e));
    prof_end(slot, st, g_last_x3_variant);
    return rc;
}
static int x3_ensure(Tower& t, int T, int W) {
    if (T <= t.x3_T && W <= t.x3_W) return RLCF_OK;
    T = std::max(T, t.x3_T); W = std::max(W, t.x3_W);
    const size_t n = (size_t)T * W * 4;                     // one interleaved (hi, lo) pair per element
    TRY(t.h2.ensure(n)); TRY(t.a2.ensure(n)); TRY(t.f2.ensure(4 * n));
    t.x3_T = T; t.x3_W = W;
    return RLCF_OK;
}

// ------------------------------------------------------------------ weights
static const float* rawp(ClipModel& m, const std::string& k, size_t numel) {
    auto it = m.raw.find(k);
    if (it == m.raw.end()) { rlcf_set_error("missing weight '%s'", k.c_str()); return nullptr; }
    if (it->second.bytes != numel * sizeof(float)) {
        rlcf_set_error("weight '%s': expected %zu elements, got %zu", k.c_str(), numel, it->second.bytes / sizeof(float));
        return nullptr;
    }
    return it->second.as<float>();
}
static int make_split(rlcf_engine* e, ClipModel& m, const float* w, size_t numel, hipStream_t st) {
    if (!prec_x3(e) || !w) return RLCF_OK;
    DevBuf hi;                                             // both parts in one allocation
    TRY(hi.ensure(numel * 4));
    const int il = numel % 32 == 0;                        // K % 32 == 0 (every weight the split-f16 GEMM accepts): interleaved pairs
    void* lo = il ? lo_of(hi.p) : (void*)((char*)hi.p + numel * 2);
    // exact power-of-two pre-scale that lifts the tensor to max|w| in [2^9, 2^10): lo parts of all but negligible
    // elements are then normal f16 numbers (full 22-bit operand), far from f16 overflow
    DevBuf amax;                                           // (finalize-time scratch, released below)
    TRY(amax.ensure(sizeof(float)));
    TRY(launch_absmax(w, (int64_t)numel, amax.as<float>(), st));
    float mx = 0.f;
    RLCF_HIP_CHECK(hipMemcpyAsync(&mx, amax.p, sizeof(float), hipMemcpyDeviceToHost, st));
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    amax.release();
    int sh = 0;
    if (mx > 0.f && std::isfinite(mx)) sh = std::max(-8, std::min(12, 9 - (int)std::floor(std::log2(mx))));
    const float scale = std::ldexp(1.0f, sh);
    TRY(launch_split_f16x2(w, hi.p, lo, (int64_t)numel, st, scale, il));
    // Is the lo half identically zero?  It is for every GEMM weight of a released CLIP checkpoint (stored as fp16 in the archives the
    // reference loads, TPT/clip/clip.py:120-141 / model.py:399-436), and then the a_hi . w_lo pass of every product with this weight adds
    // exact zeros: the 256x256 kernel drops it (gemm_f16x3.hip, WLO0).  RLCF_X3_WLO0=0 keeps three passes (A/B; read at finalize).
    bool lo_zero = false;
    {
        const char* ev = getenv("RLCF_X3_WLO0");
        if (il && !(ev && atoi(ev) == 0)) {
            DevBuf flag;
            TRY(flag.ensure(sizeof(int)));
            RLCF_HIP_CHECK(hipMemsetAsync(flag.p, 0, sizeof(int), st));
            TRY(launch_f16_grid_check(w, (int64_t)numel, scale, (int*)flag.p, st));
            int bad = 1;
            RLCF_HIP_CHECK(hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
            RLCF_HIP_CHECK(hipStreamSynchronize(st));
            flag.release();
            lo_zero = bad == 0;
        }
    }
    ClipModel::SplitW sw{hi.p, lo, 1.0f / scale};
    sw.lo_zero = lo_zero; sw.lo_zero_ckpt = lo_zero;
    if (lo_zero) {                                         // the hi halves alone, row-major: what the packed-W form of the 256x256 kernel stages
        DevBuf ho;
        TRY(ho.ensure(numel * 2 + 256));
        TRY(launch_split_f16x2(w, ho.p, nullptr, (int64_t)numel, st, scale, 0));
        sw.hi_only = ho.p;
        m.derived.push_back(std::move(ho));
    }
    m.split_of[w] = sw;
    m.derived.push_back(std::move(hi));
    if (prec_single(e) && numel % 64 == 0) {                 // plain f16 copy for the single-pass forward pipeline (same pre-scale)
        DevBuf f;
        TRY(f.ensure(numel * 2 + 64));
        TRY(launch_split_f16x2(w, f.p, nullptr, (int64_t)numel, st, scale, 0));
        m.f16_of[w] = ClipModel::SplitW{f.p, nullptr, 1.0f / scale};
        m.derived.push_back(std::move(f));
    }
    return RLCF_OK;
}
int engine_make_split(rlcf_engine* e, ClipModel& m, const float* w, size_t numel, hipStream_t st) { return make_split(e, m, w, numel, st); }
// RLCF_PREC_F16, image towers: the f16 copy of W diag(gamma) (own power-of-two pre-scale, as make_split), its row sums and W beta + b —
// the operands of the LayerNorm-folded products (gemm_f16.hip MODE 1; rowops.hip, end of file)
static int make_lnfold(rlcf_engine* e, ClipModel& m, const float* W, const float* gamma, const float* beta, const float* b, int N, int K, hipStream_t st) {
    if (!prec_single(e) || (size_t)N * K % 64) return RLCF_OK;
    DevBuf wg, amax, w16, sb;
    TRY(wg.ensure((size_t)N * K * sizeof(float)));
    TRY(sb.ensure((size_t)2 * N * sizeof(float)));                 // [s | bprime]
    TRY(launch_ln_fold_w(W, gamma, beta, b, wg.as<float>(), sb.as<float>() + N, N, K, st));
    TRY(amax.ensure(sizeof(float)));
    TRY(launch_absmax(wg.as<float>(), (int64_t)N * K, amax.as<float>(), st));
    float mx = 0.f;
    RLCF_HIP_CHECK(hipMemcpyAsync(&mx, amax.p, sizeof(float), hipMemcpyDeviceToHost, st));
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    int sh = 0;
    if (mx > 0.f && std::isfinite(mx)) sh = std::max(-8, std::min(12, 9 - (int)std::floor(std::log2(mx))));
    const float scale = std::ldexp(1.0f, sh);
    TRY(w16.ensure((size_t)N * K * 2 + 64));
    TRY(launch_split_f16x2(wg.as<float>(), w16.p, nullptr, (int64_t)N * K, st, scale, 0));
    TRY(launch_rowsum_f16(w16.p, 1.0f / scale, sb.as<float>(), N, K, st));
    RLCF_HIP_CHECK(hipStreamSynchronize(st));                        // (wg / amax are released here)
    m.lnfold_of[W] = ClipModel::LnFold{w16.p, 1.0f / scale, sb.as<float>(), sb.as<float>() + N};
    m.derived.push_back(std::move(w16));
    m.derived.push_back(std::move(sb));
    wg.release(); amax.release();
    return RLCF_OK;
}
static const float* make_transposed(ClipModel& m, const float* w, int rows, int cols, hipStream_t st) {
    m.derived.emplace_back();
    DevBuf& d = m.derived.back();
    if (d.ensure((size_t)rows * cols * sizeof(float)) != RLCF_OK) return nullptr;
    if (launch_transpose(w, d.as<float>(), rows, cols, st) != RLCF_OK) return nullptr;
    return d.as<float>();
}
#define NEED(ptr) do { if (!(ptr)) return RLCF_ERR_STATE; } while (0)

static int resolve_tower(ClipModel& m, TowerW& t, const std::string& prefix, int layers, int width, bool need_T, hipStream_t st) {
    t.layers = layers; t.width = width;
    t.blk.resize(layers);
    const size_t W = width;
    for (int i = 0; i < layers; ++i) {
        const std::string p = prefix + ".resblocks." + std::to_string(i) + ".";
        BlockW& b = t.blk[i];
        NEED(b.ln1_w = rawp(m, p + "ln_1.weight", W));  NEED(b.ln1_b = rawp(m, p + "ln_1.bias", W));
        NEED(b.in_w = rawp(m, p + "attn.in_proj_weight", 3 * W * W));  NEED(b.in_b = rawp(m, p + "attn.in_proj_bias", 3 * W));
        NEED(b.out_w = rawp(m, p + "attn.out_proj.weight", W * W));  NEED(b.out_b = rawp(m, p + "attn.out_proj.bias", W));
        NEED(b.ln2_w = rawp(m, p + "ln_2.weight", W));  NEED(b.ln2_b = rawp(m, p + "ln_2.bias", W));
        NEED(b.fc_w = rawp(m, p + "mlp.c_fc.weight", 4 * W * W));  NEED(b.fc_b = rawp(m, p + "mlp.c_fc.bias", 4 * W));
        NEED(b.proj_w = rawp(m, p + "mlp.c_proj.weight", 4 * W * W));  NEED(b.proj_b = rawp(m, p + "mlp.c_proj.bias", W));
        if (need_T) {
            NEED(b.in_wT = make_transposed(m, b.in_w, 3 * width, width, st));
            NEED(b.out_wT = make_transposed(m, b.out_w, width, width, st));
            NEED(b.fc_wT = make_transposed(m, b.fc_w, 4 * width, width, st));
            NEED(b.proj_wT = make_transposed(m, b.proj_w, width, 4 * width, st));
        }
    }
    return RLCF_OK;
}

// pad conv1.weight [Wv, 3*ps*ps] to [Wv, Kp]
__global__ void pad_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int k, int kp) {
    const long total = (long)rows * kp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / kp), c = (int)(i % kp);
        out[i] = c < k ? in[(size_t)r * k + c] : 0.f;
    }
}

int engine_finalize(rlcf_engine* e, int which, hipStream_t st) {
    ClipModel& m = e->model[which];
    if (!m.present) { rlcf_set_error("model %d not configured", which); return RLCF_ERR_STATE; }
    const rlcf_clip_cfg& c = m.cfg;
    for (auto& d : m.derived) d.release();
    m.derived.clear();
    m.derived.reserve(32 * (c.vision_layers + c.text_layers) + 16);
    m.vis.blk.clear(); m.vis.layers = 0;
    m.split_of.clear();
    m.f16_of.clear();
    m.lnfold_of.clear();
    if (which == RLCF_STUDENT) e->lnfold_stale = false;
    const int Wt = c.text_width, D = c.embed_dim;
    const bool rn = is_resnet(c);
    if (which == RLCF_STUDENT) {          // the flat tunable buffer pointed into the previous weights
        e->vw_count = 0; e->vw_dirty = false; e->vw_slots.clear(); e->vw_refresh.clear();
        e->tw_count = 0; e->tw_dirty = false; e->tw_slots.clear(); e->tw_refresh.clear(); e->tln_count = 0;
    }
    if (rn) {
        if (which == RLCF_STUDENT) e->ln_count = 0;       // no LayerNorm to tune: the LN path refuses a ResNet student
        TRY(resnet_finalize(e, m, st));
    } else {
    const int Wv = c.vision_width, ps = c.vision_patch_size;
    const int K = 3 * ps * ps;
    m.Kp = (K + 63) / 64 * 64;
    m.tokens = (c.image_resolution / ps) * (c.image_resolution / ps) + 1;
    const float* conv = rawp(m, "visual.conv1.weight", (size_t)Wv * K);
    NEED(conv);
    m.conv_raw = conv;
    if (m.Kp == K) m.conv_w = conv;
    else {
        m.derived.emplace_back();
        TRY(m.derived.back().ensure((size_t)Wv * m.Kp * sizeof(float)));
        pad_rows_kernel<<<dim3(1024), dim3(256), 0, st>>>(conv, m.derived.back().as<float>(), Wv, K, m.Kp);
        RLCF_LAUNCH_CHECK();
        m.conv_w = m.derived.back().as<float>();
    }
    NEED(m.cls = rawp(m, "visual.class_embedding", Wv));
    NEED(m.vpos = rawp(m, "visual.positional_embedding", (size_t)m.tokens * Wv));
    NEED(m.lnpre_w = rawp(m, "visual.ln_pre.weight", Wv));   NEED(m.lnpre_b = rawp(m, "visual.ln_pre.bias", Wv));
    NEED(m.lnpost_w = rawp(m, "visual.ln_post.weight", Wv)); NEED(m.lnpost_b = rawp(m, "visual.ln_post.bias", Wv));
    const float* vproj = rawp(m, "visual.proj", (size_t)Wv * D);
    NEED(vproj);
    m.vproj = vproj;
    NEED(m.vprojT = make_transposed(m, vproj, Wv, D, st));
    TRY(resolve_tower(m, m.vis, "visual.transformer", c.vision_layers, Wv, which == RLCF_STUDENT, st));   // W^T: LN-tuning backward
    if (which == RLCF_STUDENT) {
        // every visual LayerNorm parameter in one tunable buffer (CLIPCLS_TTA.parameters() with only_norm,
        // custom_clip.py:477-485, in named_parameters order); the towers read LN weights from it
        const int L = c.vision_layers;
        e->ln_count = (4 * L + 4) * Wv;
        const size_t nb = (size_t)e->ln_count * sizeof(float);
        TRY(e->ln_params.ensure(nb)); TRY(e->ln_init.ensure(nb)); TRY(e->ln_grad.ensure(nb)); TRY(e->ln_m.ensure(nb)); TRY(e->ln_v.ensure(nb));
        float* P = e->ln_params.as<float>();
        std::vector<const float**> slots = {&m.lnpre_w, &m.lnpre_b};
        for (BlockW& b : m.vis.blk) { slots.push_back(&b.ln1_w); slots.push_back(&b.ln1_b); slots.push_back(&b.ln2_w); slots.push_back(&b.ln2_b); }
        slots.push_back(&m.lnpost_w); slots.push_back(&m.lnpost_b);
        for (size_t i = 0; i < slots.size(); ++i) {
            RLCF_HIP_CHECK(hipMemcpyAsync(P + i * Wv, *slots[i], Wv * sizeof(float), hipMemcpyDeviceToDevice, st));
            *slots[i] = P + i * Wv;
        }
        RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_init.p, P, nb, hipMemcpyDeviceToDevice, st));
        TRY(e->ln_clip.ensure(nb)); TRY(e->ln_mom.ensure(nb));       // clip_state_dict / momentum_state_dict (custom_clip.py:395-399)
        RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_clip.p, P, nb, hipMemcpyDeviceToDevice, st));
        RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_mom.p, P, nb, hipMemcpyDeviceToDevice, st));
        std::vector<int32_t> idx(e->max_views);
        for (int i = 0; i < e->max_views; ++i) idx[i] = i * m.tokens;
        TRY(e->cls_row_idx.ensure(idx.size() * sizeof(int32_t)));
        RLCF_HIP_CHECK(hipMemcpyAsync(e->cls_row_idx.p, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        RLCF_HIP_CHECK(hipStreamSynchronize(st));
        TRY(e->vit_inv_norm.ensure(e->max_views * sizeof(float)));
    }
    }
    NEED(m.tok_emb = rawp(m, "token_embedding.weight", (size_t)c.vocab_size * Wt));
    NEED(m.tpos = rawp(m, "positional_embedding", (size_t)c.context_length * Wt));
    NEED(m.lnf_w = rawp(m, "ln_final.weight", Wt));  NEED(m.lnf_b = rawp(m, "ln_final.bias", Wt));
    NEED(m.tproj = rawp(m, "text_projection", (size_t)Wt * D));
    NEED(m.tprojT = make_transposed(m, m.tproj, Wt, D, st));
    TRY(resolve_tower(m, m.txt, "transformer", c.text_layers, Wt, which == RLCF_STUDENT, st));
    const float* ls = rawp(m, "logit_scale", 1);
    NEED(ls);
    float lsh = 0.f;
    RLCF_HIP_CHECK(hipMemcpyAsync(&lsh, ls, sizeof(float), hipMemcpyDeviceToHost, st));
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    m.logit_scale_exp = expf(lsh);
    // split-f16 copies of every forward GEMM weight (F16X3 mode)
    if (!rn) {
        TRY(make_split(e, m, m.conv_w, (size_t)c.vision_width * m.Kp, st));
        TRY(make_split(e, m, m.vprojT, (size_t)c.vision_width * D, st));
    }
    TRY(make_split(e, m, m.tprojT, (size_t)Wt * D, st));
    for (TowerW* t : {&m.vis, &m.txt})
        for (BlockW& b : t->blk) {
            const size_t W2 = (size_t)t->width * t->width;
            TRY(make_split(e, m, b.in_w, 3 * W2, st)); TRY(make_split(e, m, b.out_w, W2, st));
            TRY(make_split(e, m, b.fc_w, 4 * W2, st)); TRY(make_split(e, m, b.proj_w, 4 * W2, st));
            if (b.in_wT) {                               // backward (dX = dY.W) operands of the student text tower
                TRY(make_split(e, m, b.in_wT, 3 * W2, st)); TRY(make_split(e, m, b.out_wT, W2, st));
                TRY(make_split(e, m, b.fc_wT, 4 * W2, st)); TRY(make_split(e, m, b.proj_wT, 4 * W2, st));
            }
        }
    if (!rn && prec_single(e) && c.vision_width % 256 == 0)
        for (BlockW& b : m.vis.blk) {                    // LayerNorm-folded in_proj / c_fc of the image tower (RLCF_PREC_F16)
            const int Wv = c.vision_width;
            TRY(make_lnfold(e, m, b.in_w, b.ln1_w, b.ln1_b, b.in_b, 3 * Wv, Wv, st));
            TRY(make_lnfold(e, m, b.fc_w, b.ln2_w, b.ln2_b, b.fc_b, 4 * Wv, Wv, st));
        }
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    m.finalized = true;
    return RLCF_OK;
}

// ------------------------------------------------------------------ workspaces
// (Growth path only: the workspaces are sized by the first call that needs them.  The fill is enqueued on the CALLER's stream — the
// stream every kernel that touches the buffer is enqueued on — so it is ordered against them whatever kind of stream that is; a fill on
// the NULL stream is not ordered against a non-blocking stream at all, which is how a fresh lane engine once raced its own workspace.)
static int tower_ensure(Tower& t, int T, int width, hipStream_t st) {
    if (T <= t.T && width <= t.width) return RLCF_OK;
    T = std::max(T, t.T); width = std::max(width, t.width);
    const size_t n = (size_t)T * width * sizeof(float);
    TRY(t.x.ensure(n)); TRY(t.h.ensure(n)); TRY(t.qkv.ensure(3 * n)); TRY(t.a.ensure(n)); TRY(t.f.ensure(4 * n));
    RLCF_HIP_CHECK(hipMemsetAsync(t.a.p, 0, n, st));
    t.T = T; t.width = width;
    return RLCF_OK;
}
static int tower_ensure_saved(Tower& t, int T, int width, int layers, hipStream_t st) {
    if (T <= t.saved_T && layers <= t.saved_layers && (int)t.sv.size() == layers) return RLCF_OK;
    const size_t per = (size_t)T * width;               // floats
    const size_t per_lse = ((size_t)T * (width / HEAD_DIM) + 63) / 64 * 64;     // keeps the following layers 256-B aligned
    const size_t per_layer = per * (1 + 3 + 1 + 1 + 4) + per_lse;
    TRY(t.saved.ensure(per_layer * layers * sizeof(float)));
    RLCF_HIP_CHECK(hipMemsetAsync(t.saved.p, 0, per_layer * layers * sizeof(float), st));
    t.sv.resize(layers);
    float* p = t.saved.as<float>();
    for (int l = 0; l < layers; ++l) {
        t.sv[l].x = p; p += per;
        t.sv[l].qkv = p; p += 3 * per;
        t.sv[l].a = p; p += per;
        t.sv[l].x1 = p; p += per;
        t.sv[l].f = p; p += 4 * per;
        t.sv[l].lse = p; p += per_lse;
    }
    t.saved_T = T; t.saved_layers = layers;
    return RLCF_OK;
}
static int bwd_ensure(rlcf_engine* e, int T, int width) {
    if ((size_t)T * width <= e->bwd_elems) return RLCF_OK;
    e->bwd_elems = (size_t)T * width;
    const size_t n = (size_t)T * width * sizeof(float);
    TRY(e->dX.ensure(n)); TRY(e->dA.ensure(n)); TRY(e->dH.ensure(n)); TRY(e->dF.ensure(4 * n)); TRY(e->dQKV.ensure(3 * n));
    e->bwd_T = T;
    return RLCF_OK;
}

// ------------------------------------------------------------------ full image-encoder tuning
// CLIPCLS_TTA(only_norm=False): parameters() = clip_model.visual.parameters() (TPT/clip/custom_clip.py:477-479), what
// scripts/rlcf-tune.sh runs (`--tune_norm` defaults to 0, params.py:73).  The LayerNorm tensors stay in e->ln_params; every other
// visual tensor moves into ONE flat buffer (AdamW = one launch, reset = one copy) and the towers read the weights from it.
int engine_visual_enable(rlcf_engine* e, hipStream_t st) {
    if (e->vw_count) return RLCF_OK;
    ClipModel& m = e->model[RLCF_STUDENT];
    if (!m.finalized) { rlcf_set_error("student model not finalized"); return RLCF_ERR_STATE; }
    if (is_resnet(m.cfg)) return engine_rn_visual_enable(e, st);          // (resnet.hip: convolutions, BatchNorms, attention pool)
    if (prec_single(e)) { rlcf_set_error("encoder tuning runs in RLCF_PREC_F32 / RLCF_PREC_F16X3 (RLCF_PREC_F16 is the prompt path's performance mode)"); return RLCF_ERR_STATE; }
    const rlcf_clip_cfg& c = m.cfg;
    const size_t Wv = c.vision_width, D = c.embed_dim, K = (size_t)3 * c.vision_patch_size * c.vision_patch_size, W2 = Wv * Wv;
    struct Item { const float** slot; size_t numel; };
    std::vector<Item> items = {{&m.cls, Wv}, {&m.vpos, (size_t)m.tokens * Wv}, {&m.vproj, Wv * D}, {&m.conv_raw, Wv * K}};
    for (BlockW& b : m.vis.blk) {
        items.push_back({&b.in_w, 3 * W2}); items.push_back({&b.in_b, 3 * Wv}); items.push_back({&b.out_w, W2}); items.push_back({&b.out_b, Wv});
        items.push_back({&b.fc_w, 4 * W2}); items.push_back({&b.fc_b, 4 * Wv}); items.push_back({&b.proj_w, 4 * W2}); items.push_back({&b.proj_b, Wv});
    }
    size_t total = 0;
    e->vw_slots.clear();
    for (const Item& it : items) { e->vw_slots.push_back(VwSlot{total, it.numel}); total += (it.numel + 63) / 64 * 64; }
    const size_t nb = total * sizeof(float);
    for (DevBuf* d : {&e->vw, &e->vw_init, &e->vw_grad, &e->vw_m, &e->vw_v, &e->vw_clip, &e->vw_mom}) TRY(d->ensure(nb));
    RLCF_HIP_CHECK(hipMemsetAsync(e->vw.p, 0, nb, st));
    for (size_t i = 0; i < items.size(); ++i) {
        const float* old = *items[i].slot;
        float* dst = e->vw.as<float>() + e->vw_slots[i].off;
        RLCF_HIP_CHECK(hipMemcpyAsync(dst, old, items[i].numel * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (items[i].slot == &m.conv_raw && m.conv_w == old) m.conv_w = dst;       // Kp == K: the GEMM reads conv1.weight as stored
        auto sp = m.split_of.find(old);
        if (sp != m.split_of.end()) { const ClipModel::SplitW s = sp->second; m.split_of.erase(sp); m.split_of[dst] = s; }
        *items[i].slot = dst;
    }
    for (DevBuf* d : {&e->vw_init, &e->vw_clip, &e->vw_mom}) RLCF_HIP_CHECK(hipMemcpyAsync(d->p, e->vw.p, nb, hipMemcpyDeviceToDevice, st));
    // derived copies that must follow the live weights (the pre-scale of a split copy stays the one of the checkpoint: a tuning
    // step moves a weight by ~lr, far inside the 2^6 headroom of the scaled f16 range)
    e->vw_refresh.clear();
    auto add_split = [&](const float* w, size_t numel) {
        auto it = m.split_of.find(w);
        // a TUNED weight leaves the fp16 grid at its first optimizer step (three passes from then on) and is back on it after every reset
        // to the checkpoint's values (refresh_derived: lo_zero follows; hi_only stays the checkpoint's copy)
        if (it != m.split_of.end())
            e->vw_refresh.push_back(VwRefresh{VW_SPLIT, w, nullptr, numel, 0, it->second.hi, it->second.lo, 1.0f / it->second.inv_scale,
                                              it->second.lo == lo_of(it->second.hi), &it->second});
    };
    auto add_T = [&](const float* w, const float* wT, size_t rows, size_t cols) {
        if (!wT) return;
        e->vw_refresh.push_back(VwRefresh{VW_TRANSPOSE, w, (float*)wT, rows, cols, nullptr, nullptr, 1.f, 0});
        add_split(wT, rows * cols);
    };
    if (m.conv_w != m.conv_raw) e->vw_refresh.push_back(VwRefresh{VW_PAD, m.conv_raw, (float*)m.conv_w, Wv, K, nullptr, nullptr, 1.f, 0});
    add_split(m.conv_w, Wv * m.Kp);
    add_T(m.vproj, m.vprojT, Wv, D);
    for (BlockW& b : m.vis.blk) {
        add_split(b.in_w, 3 * W2); add_split(b.out_w, W2); add_split(b.fc_w, 4 * W2); add_split(b.proj_w, 4 * W2);
        add_T(b.in_w, b.in_wT, 3 * Wv, Wv); add_T(b.out_w, b.out_wT, Wv, Wv); add_T(b.fc_w, b.fc_wT, 4 * Wv, Wv); add_T(b.proj_w, b.proj_wT, Wv, 4 * Wv);
    }
    e->vw_count = total;
    e->vw_dirty = false;
    e->vw_init_is_ckpt = true;
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    return RLCF_OK;
}

static int refresh_derived(const std::vector<VwRefresh>& list, int Kp, hipStream_t st, bool at_checkpoint = false) {
    for (const VwRefresh& r : list) {
        if (r.kind == VW_PAD) {
            pad_rows_kernel<<<dim3(1024), dim3(256), 0, st>>>(r.src, r.dst, (int)r.rows, (int)r.cols, Kp);
            RLCF_LAUNCH_CHECK();
        } else if (r.kind == VW_TRANSPOSE) {
            TRY(launch_transpose(r.src, r.dst, (int)r.rows, (int)r.cols, st));
        } else {
            TRY(launch_split_f16x2(r.src, r.hi, r.lo, (int64_t)r.rows, st, r.scale, r.il));
            if (r.sw) r.sw->lo_zero = at_checkpoint && r.sw->lo_zero_ckpt;      // (host-side launch choice: in stream order with the split above)
        }
    }
    return RLCF_OK;
}
int engine_visual_refresh(rlcf_engine* e, hipStream_t st, bool at_checkpoint) {
    if (is_resnet(e->model[RLCF_STUDENT].cfg)) return rn_visual_refresh(e, st, at_checkpoint);
    return refresh_derived(e->vw_refresh, e->model[RLCF_STUDENT].Kp, st, at_checkpoint);
}

// Linear weight gradient dW[N,K] = dY[T,N]^T X[T,K] (+ db[N] += column sums of dY): both operands are transposed to K-major
// [*, Tp] (token dimension zero padded to the GEMM's K granule) and go through the NT GEMM of the engine's precision; in
// split-f16 mode dY^T is scaled by a power of two found on the device (gradients sit far below f16's normal range).
static int wgrad(rlcf_engine* e, const float* dY, int ldy, int N, const float* X, int ldx, int K, int T, float* dW, float* db, hipStream_t st) {
    const int Tp = (T + 31) / 32 * 32;
    TRY(e->wg_yt.ensure((size_t)N * Tp * sizeof(float))); TRY(e->wg_xt.ensure((size_t)K * Tp * sizeof(float)));
    float *yt = e->wg_yt.as<float>(), *xt = e->wg_xt.as<float>();
    TRY(launch_transpose_pad(dY, ldy, yt, T, N, Tp, st));
    TRY(launch_transpose_pad(X, ldx, xt, T, K, Tp, st));
    e->last_flops += 2.0 * N * K * T;
    int rc;
    // split-f16 kernels for wide outputs — and for ANY output when the token dimension is long (the convolutions of a ResNet student see
    // n*H*W = 10^5..10^6 rows: there the split-f16 launcher cuts the K loop into slices, gemm_f16x3.hip; the f32 kernel would walk it in
    // a handful of workgroups)
    if (prec_x3(e) && (N >= 256 || Tp >= 8192) && K % 4 == 0) {
        if ((size_t)N * Tp > e->a_split_elems) { TRY(e->a_hi.ensure((size_t)N * Tp * 4)); e->a_split_elems = (size_t)N * Tp; }
        TRY(e->w_hi.ensure((size_t)K * Tp * 4));
        TRY(e->dyn.ensure(3 * sizeof(float)));
        TRY(launch_split_f16x2_dyn(yt, e->a_hi.p, lo_of(e->a_hi.p), (int64_t)N * Tp, e->dyn.as<float>(), st, 1));
        TRY(launch_split_f16x2(xt, e->w_hi.p, lo_of(e->w_hi.p), (int64_t)K * Tp, st, 1.0f, 1));
        const int slot = prof_begin(st, 2.0 * N * K * Tp, N, K, Tp);
        rc = launch_gemm_f16x3(e->a_hi.p, lo_of(e->a_hi.p), 2 * Tp, e->w_hi.p, lo_of(e->w_hi.p), 2 * Tp, nullptr, nullptr, 0, nullptr, 0, dW, K,
                               nullptr, nullptr, 0, N, K, Tp, 1.f, RLCF_EPI_NONE, st, e->dyn.as<float>() + 2, nullptr, 0, ws_ptr(e),
                               ws_bytes(e), 0, nullptr, ws_epoch(e));
        prof_end(slot, st, g_last_x3_variant);
    } else {
        GemmArgs g{};
        g.A = yt; g.lda = Tp; g.W = xt; g.ldw = Tp; g.C = dW; g.ldc = K; g.M = N; g.N = K; g.K = Tp; g.alpha = 1.f; g.epilogue = RLCF_EPI_NONE;
        const int slot = prof_begin(st, 2.0 * N * K * Tp, N, K, Tp);
        rc = launch_gemm_f32(g, st);
        prof_end(slot, st);
    }
    TRY(rc);
    if (db) TRY(launch_colsum(dY, ldy, T, N, db, st, PARTS_WS(e)));
    return RLCF_OK;
}

int engine_wgrad(rlcf_engine* e, const float* dY, int ldy, int N, const float* X, int ldx, int K, int T, float* dW, float* db, hipStream_t st) {
    if (db) TRY(e->parts_ws.ensure(RLCF_PARTS_WS_FLOATS * sizeof(float)));      // (bias column sums in a fixed order: bit-reproducible)
    return wgrad(e, dY, ldy, N, X, ldx, K, T, dW, db, st);
}

// ------------------------------------------------------------------ transformer passes
// Transformer.forward, TPT/clip/model.py:195-203 with ResidualAttentionBlock :189-192.
// x0: [T,W] input (ws.x, or sv[0].x when saving).  Result always lands in ws.x.
// Per-view LayerNorm sets (batched LN-tuning inference): a tunable LayerNorm pointer is redirected into e->lng_base and the
// kernels pick the set of the row's view; any other LayerNorm (text tower, reward models) is left alone.
struct LnRef { const float* p; int group_rows, group_stride; };
static inline LnRef ln_ref(const rlcf_engine* e, const float* p, int view_rows) {      // view_rows: rows one view has in this matrix
    const float* lo = e->ln_params.as<float>();
    if (e->lng_base && lo && p >= lo && p < lo + e->ln_count) return LnRef{e->lng_base + (p - lo), view_rows * e->lng_views, e->ln_count};
    return LnRef{p, 0, 0};
}
#define LN_FWD(xp, wp, bp, yp, rows, W)                                                                                    \
    do { const LnRef gw_ = ln_ref(e, (wp), ln_view_rows), gb_ = ln_ref(e, (bp), ln_view_rows);                              \
         TRY(launch_layernorm_fwd((xp), gw_.p, gb_.p, (yp), (rows), (W), st, gw_.group_rows, gw_.group_stride)); } while (0)
#define LN_FWD_SPLIT(xp, wp, bp, hh, hl, rows, W)                                                                           \
    do { const LnRef gw_ = ln_ref(e, (wp), ln_view_rows), gb_ = ln_ref(e, (bp), ln_view_rows);                              \
         const bool sg_ = prec_single(e);                 /* RLCF_PREC_F16: plain f16 rows, no lo part */                    \
         /* profile record of kind 11: an HBM-bound kernel, the `flops` field carries its ALGORITHMIC BYTES (f32 row in, pair row out) */ \
         const int ps_ = prof_begin(st, (double)(rows) * (W) * (sg_ ? 6.0 : 8.0), (rows), (W), 0);                             \
         int rc_ln_ = launch_layernorm_fwd_split((xp), gw_.p, gb_.p, nullptr, (hh), sg_ ? nullptr : (hl), (rows), (W), st, gw_.group_rows, \
                                                 gw_.group_stride, sg_ ? 0 : 1);                                              \
         prof_end(ps_, st, 11);                                                                                                \
         TRY(rc_ln_); } while (0)

// RLCF_PREC_F16: x += d (the f16 output of the preceding out_proj / c_proj product, in dh), then LayerNorm(x) -> plain f16 into dh itself
#define LN_ADD_FWD(xp, wp, bp, dh, rows, W)                                                                                 \
    do { const LnRef gw_ = ln_ref(e, (wp), ln_view_rows), gb_ = ln_ref(e, (bp), ln_view_rows);                              \
         const int ps_ = prof_begin(st, (double)(rows) * (W) * 12.0, (rows), (W), 0);       /* kind 11: f32 row in + out, f16 row in + out */ \
         int rc_ln_ = launch_layernorm_add_fwd((xp), (dh), gw_.p, gb_.p, (dh), (rows), (W), st, gw_.group_rows, gw_.group_stride);          \
         prof_end(ps_, st, 11);                                                                                                \
         TRY(rc_ln_); } while (0)

// cls_seqs / cls_idx / cls_out (image towers, split-f16 pipeline): only row `cls_idx[s]` of every sequence is consumed after the
// last block (ln_post(x[:, 0]) @ proj, model.py:235-238), and out_proj, the MLP and the residual adds act row by row — so the LAST
// block runs its attention for that one query per sequence (keys: the whole sequence; cls_seqs = {q_start = cls row, q_len = 1,
// prefix = the other rows}) and everything after it on the n_seq gathered rows only; the result lands compact in cls_out [n_seq, W]
// (ws.x then holds the input of the last block).  Exact: no other row of the last block's output is ever read.
static int transformer_forward(rlcf_engine* e, const TowerW& w, Tower& ws, const rlcf_seq* seqs, int n_seq, int max_q_len,
                               long attn_pairs, int causal, int T, bool save, hipStream_t st, const rlcf_seq* cls_seqs = nullptr,
                               const int32_t* cls_idx = nullptr, float* cls_out = nullptr, int row0 = 0 /* first row of this call in the tower buffers (chunked image passes) */) {
    const int W = w.width, L = w.layers;
    const int ln_view_rows = max_q_len;          // (per-view LayerNorm sets only exist for the image tower: one sequence per view)
    if (prec_x3(e) && !save && T > 512 && W % 32 == 0 && (!prec_single(e) || W % 64 == 0)) {
        // split-f16 pipeline: LN, attention and the QuickGELU epilogue emit (hi, lo) f16 pairs for the next GEMM
        TRY(x3_ensure(ws, row0 + T, W));
        // this call works on rows [row0, row0 + T) of the tower buffers (row0 != 0: a chunk of views of a larger pass); sequence descriptors
        // and class-token indices hold ABSOLUTE rows, so the attention kernel and the row gathers take the base pointers
        const size_t es = prec_single(e) ? 2 : 4;                           // bytes per element of an operand matrix (plain f16 / hi|lo pair)
        float* const xb = ws.x.as<float>();
        float* x = xb + (size_t)row0 * W;
        void* const h2 = (char*)ws.h2.p + (size_t)row0 * W * es;
        void* const a2 = (char*)ws.a2.p + (size_t)row0 * W * es;
        void* const qkv = (char*)ws.qkv.p + (size_t)row0 * 3 * W * es;
        void* const f2 = (char*)ws.f2.p + (size_t)row0 * 4 * W * es;
        // RLCF_PREC_F16: out_proj / c_proj write their product d as f16 (into the LayerNorm buffer, dead by then) and the residual add rides
        // in the LayerNorm kernel that follows (x += d; h = LN(x), in place over d) — the linear output rounded to f16 before the add is
        // the reference's own autocast arithmetic (TPT/clip/model.py:187-192 under tpt_cls_rl.py:52); the GEMM's epilogue then moves 128 KB
        // per 256x256 tile instead of 512 KB and all four products of a block run on the persistent f16 kernel (gemm_f16.hip).
        // RLCF_F16_RESADD=0: the f32 residual epilogues (A/B measurements)
        static int f16res_env = -1;
        if (f16res_env < 0) { const char* ev = getenv("RLCF_F16_RESADD"); f16res_env = ev ? atoi(ev) : 1; }
        const bool f16res = prec_single(e) && f16res_env && W % 4 == 0;
        // RLCF_PREC_F16 image towers: LayerNorm FOLDED into the products (round 5).  The residual stream is kept as f16 rows x16 (the
        // reference's own autocast arithmetic: its LayerNorm casts back to the fp16 input type and x + attention(...) adds fp16 tensors,
        // TPT/clip/model.py:157-163,187-192 under tpt_cls_rl.py:52); in_proj / c_fc read x16 ITSELF against the gamma-folded weight and finish
        // the normalisation per row in the epilogue (MODE 1); out_proj / c_proj add into x16 in place and leave partial row statistics
        // (MODE 2), which one small kernel turns into (mean, rstd).  No LayerNorm launch, no normalised copy of the stream.
        // Round 6: OFF by default — an opt-in (RLCF_F16_LNFOLD=1 when the engine is created, or rlcf_engine_set_f16_lnfold): on the 32-sample
        // reference stream the f32 residual stream below keeps the reference's top-1 on 32 of 32 samples, the f16 stream on 31
        // (profiles/r6_notes.md; tests/test_gpu_round2.py::test_f16_single_pass_mode_b16_stream reports both against the reference's
        // float32 run AND its own fp16-autocast run).
        const ClipModel::LnFold* fold_in0 = nullptr;
        if (f16res && e->f16_lnfold && !causal && !e->lng_base && W % 256 == 0 && cls_out && cls_seqs && cls_idx && !(e->lnfold_stale && &w == &e->model[RLCF_STUDENT].vis))
            for (auto& mm : e->model) { auto it = mm.lnfold_of.find(w.blk[0].in_w); if (it != mm.lnfold_of.end()) fold_in0 = &it->second; }
        if (fold_in0) {
            auto fold_of = [&](const float* Wp) -> const ClipModel::LnFold* {
                for (auto& mm : e->model) { auto it = mm.lnfold_of.find(Wp); if (it != mm.lnfold_of.end()) return &it->second; }
                return nullptr;
            };
            auto f16_of = [&](const float* Wp) -> const ClipModel::SplitW* {
                for (auto& mm : e->model) { auto it = mm.f16_of.find(Wp); if (it != mm.f16_of.end()) return &it->second; }
                return nullptr;
            };
            const int P = (W / 256) * 4, Tall = row0 + T;
            TRY(ws.x16.ensure((size_t)Tall * W * 2)); TRY(ws.lnmr.ensure((size_t)Tall * 2 * sizeof(float)));
            TRY(ws.lnpart.ensure((size_t)P * T * 2 * sizeof(float)));
            _Float16* const x16b = (_Float16*)ws.x16.p;
            _Float16* const x16 = x16b + (size_t)row0 * W;
            float* const mr = ws.lnmr.as<float>() + (size_t)row0 * 2;
            float* const part = ws.lnpart.as<float>();
            {
                const int ps_ = prof_begin(st, (double)T * W * 6.0, T, W, 0);       // kind 11 (HBM-bound row kernel): f32 row in, f16 row out
                const int rc_init = launch_resid16_init(x, x16, mr, T, W, st);
                prof_end(ps_, st, 11);
                TRY(rc_init);
            }
            // one folded / in-place product with its profile record (the GEMM table of bench.py keys on M, N, K)
            auto prod = [&](const void* A, int lda, const void* Wf, float inv_scale, const float* bias, void* out, int ldo, int N, int K, int epi,
                            int mode, const float* s_vec) -> int {
                e->last_flops += 2.0 * T * N * K;
                const int slot = prof_begin(st, 2.0 * T * N * K, T, N, K);
                int rc = launch_gemm_f16_pp_ln(A, lda, Wf, K, bias, out, ldo, T, N, K, inv_scale, epi, mode, mr, s_vec, part, st);
                prof_end(slot, st, 3);          // (tag of the 256x256 kernels: bench.py keys its GEMM table on it)
                return rc;
            };
            for (int l = 0; l < L; ++l) {
                const BlockW& b = w.blk[l];
                const ClipModel::LnFold *fi = fold_of(b.in_w), *ff = fold_of(b.fc_w);
                const ClipModel::SplitW *wo = f16_of(b.out_w), *wp = f16_of(b.proj_w);
                if (!fi || !ff || !wo || !wp) { rlcf_set_error("transformer_forward: block %d has no folded / f16 weights", l); return RLCF_ERR_STATE; }
                TRY(prod(x16, W, fi->w16, fi->inv_scale, fi->bprime, qkv, 3 * W, 3 * W, W, RLCF_EPI_NONE, 1, fi->s));
                if (l == L - 1) {
                    // last block, class-token rows only (see the comment above transformer_forward): the small path of the unfolded pipeline
                    const size_t nw = (size_t)n_seq * W * sizeof(float);
                    TRY(IMG_BUF(e, cls_a2).ensure(nw)); TRY(IMG_BUF(e, cls_h2).ensure(nw)); TRY(IMG_BUF(e, cls_f2).ensure(4 * nw));
                    TRY(launch_attention_fwd_pair(ws.qkv.p, cls_seqs, n_seq, 1, W, nullptr, ws.a2.p, st, nullptr, 1));
                    e->last_flops += 4.0 * (double)n_seq * max_q_len * W;
                    TRY(launch_gather_rows((const float*)ws.a2.p, W / 2, cls_idx, IMG_BUF(e, cls_a2).as<float>(), W / 2, n_seq, W / 2, st));
                    TRY(launch_rows_h2f(x16b, W, cls_idx, cls_out, W, n_seq, W, st));
                    TRY(gemm_pre(e, IMG_BUF(e, cls_a2).p, W, b.out_w, b.out_b, cls_out, W, cls_out, W, nullptr, 0, n_seq, W, W, RLCF_EPI_NONE, st));
                    {
                        const int ln_view_rows = 1;
                        LN_FWD_SPLIT(cls_out, b.ln2_w, b.ln2_b, IMG_BUF(e, cls_h2).p, lo_of(IMG_BUF(e, cls_h2).p), n_seq, W);
                    }
                    TRY(gemm_pre(e, IMG_BUF(e, cls_h2).p, W, b.fc_w, b.fc_b, nullptr, 0, nullptr, 0, IMG_BUF(e, cls_f2).p, 4 * W, n_seq, 4 * W, W, RLCF_EPI_QUICKGELU, st));
                    TRY(gemm_pre(e, IMG_BUF(e, cls_f2).p, 4 * W, b.proj_w, b.proj_b, cls_out, W, cls_out, W, nullptr, 0, n_seq, W, 4 * W, RLCF_EPI_NONE, st));
                    return RLCF_OK;
                }
                {
                    const int slot = prof_begin(st, 4.0 * attn_pairs * W, T, W, max_q_len);          // kind 10: fused attention forward
                    const int arc = launch_attention_fwd_pair(ws.qkv.p, seqs, n_seq, max_q_len, W, nullptr, ws.a2.p, st, nullptr, 1);
                    prof_end(slot, st, 10);
                    TRY(arc);
                }
                e->last_flops += 4.0 * attn_pairs * W;
                TRY(prod(a2, W, wo->hi, wo->inv_scale, b.out_b, x16, W, W, W, RLCF_EPI_NONE, 2, nullptr));
                TRY(launch_ln_stats_final(part, P, T, W, mr, st));
                TRY(prod(x16, W, ff->w16, ff->inv_scale, ff->bprime, f2, 4 * W, 4 * W, W, RLCF_EPI_QUICKGELU, 1, ff->s));
                TRY(prod(f2, 4 * W, wp->hi, wp->inv_scale, b.proj_b, x16, W, W, 4 * W, RLCF_EPI_NONE, 2, nullptr));
                TRY(launch_ln_stats_final(part, P, T, W, mr, st));
            }
            return RLCF_OK;                                  // (not reached: the fold path is only taken with the class-token shortcut)
        }
        bool have_d = false;                                  // ws.h2 holds a product still to be added to x
        for (int l = 0; l < L; ++l) {
            const BlockW& b = w.blk[l];
            if (have_d) { LN_ADD_FWD(x, b.ln1_w, b.ln1_b, h2, T, W); have_d = false; }
            else LN_FWD_SPLIT(x, b.ln1_w, b.ln1_b, h2, lo_of(h2), T, W);
            // image towers (non-causal): in_proj writes Q / K / V as the f16 operand pairs the attention kernel DMAs into LDS
            // (attention_pair.hip; a pair row is as long as an f32 row, so the same buffer serves); RLCF_ATTN_OLD=1 keeps the f32 hand-over
            static int attn_old = -1;
            if (attn_old < 0) { const char* ev = getenv("RLCF_ATTN_OLD"); attn_old = ev ? atoi(ev) : 0; }
            const bool pair_attn = !causal && !attn_old;
            if (pair_attn) TRY(gemm_pre(e, h2, W, b.in_w, b.in_b, nullptr, 0, nullptr, 0, qkv, 3 * W, T, 3 * W, W, RLCF_EPI_NONE, st));
            else TRY(gemm_pre(e, h2, W, b.in_w, b.in_b, nullptr, 0, (float*)qkv, 3 * W, nullptr, 0, T, 3 * W, W, RLCF_EPI_NONE, st));
            if (l == L - 1 && cls_out && cls_seqs && cls_idx && !causal) {
                // last block, class-token rows only (see above).  Pair rows are W * 4 bytes like f32 rows: gather_rows moves both.
                const size_t nw = (size_t)n_seq * W * sizeof(float);
                TRY(IMG_BUF(e, cls_a2).ensure(nw)); TRY(IMG_BUF(e, cls_h2).ensure(nw)); TRY(IMG_BUF(e, cls_f2).ensure(4 * nw));
                {
                    const bool sg = prec_single(e);
                    if (pair_attn) TRY(launch_attention_fwd_pair(ws.qkv.p, cls_seqs, n_seq, 1, W, nullptr, ws.a2.p, st, nullptr, sg ? 1 : 0));
                    else TRY(launch_attention_fwd_x3(ws.qkv.as<float>(), cls_seqs, n_seq, 1, W, 0, nullptr, ws.a2.p, sg ? nullptr : lo_of(ws.a2.p), st,
                                                     sg ? 0 : 1, nullptr, sg ? 1 : 0));
                }
                e->last_flops += 4.0 * (double)n_seq * max_q_len * W;
                {
                    const int rw = prec_single(e) ? W / 2 : W;           // row length of the operand matrix in floats (plain f16: W halves)
                    TRY(launch_gather_rows((const float*)ws.a2.p, rw, cls_idx, IMG_BUF(e, cls_a2).as<float>(), rw, n_seq, rw, st));
                }
                TRY(launch_gather_rows(xb, W, cls_idx, cls_out, W, n_seq, W, st));
                TRY(gemm_pre(e, IMG_BUF(e, cls_a2).p, W, b.out_w, b.out_b, cls_out, W, cls_out, W, nullptr, 0, n_seq, W, W, RLCF_EPI_NONE, st));
                {   // LayerNorm sets per view (batched LN-tuning inference): one row per view here
                    const int ln_view_rows = 1;
                    LN_FWD_SPLIT(cls_out, b.ln2_w, b.ln2_b, IMG_BUF(e, cls_h2).p, lo_of(IMG_BUF(e, cls_h2).p), n_seq, W);
                }
                TRY(gemm_pre(e, IMG_BUF(e, cls_h2).p, W, b.fc_w, b.fc_b, nullptr, 0, nullptr, 0, IMG_BUF(e, cls_f2).p, 4 * W, n_seq, 4 * W, W, RLCF_EPI_QUICKGELU, st));
                TRY(gemm_pre(e, IMG_BUF(e, cls_f2).p, 4 * W, b.proj_w, b.proj_b, cls_out, W, cls_out, W, nullptr, 0, n_seq, W, 4 * W, RLCF_EPI_NONE, st));
                return RLCF_OK;
            }
            {
                const int slot = prof_begin(st, 4.0 * attn_pairs * W, T, W, max_q_len);          // kind 10: fused attention forward
                const bool sg = prec_single(e);
                static int nopack = -1;                       // RLCF_TEXT_NOPACK=1: one MFMA tile per class sequence (benchmarks)
                if (nopack < 0) { const char* ev = getenv("RLCF_TEXT_NOPACK"); nopack = ev ? atoi(ev) : 0; }
                const bool packed = causal && e->pk_cur && e->n_pk_cur > 0 && !nopack;      // several class prompts per tile (attention_x3.hip)
                const int arc = pair_attn ? launch_attention_fwd_pair(ws.qkv.p, seqs, n_seq, max_q_len, W, nullptr, ws.a2.p, st, nullptr, sg ? 1 : 0)
                                          : launch_attention_fwd_x3(ws.qkv.as<float>(), packed ? e->pk_cur : seqs, packed ? e->n_pk_cur : n_seq,
                                                                    packed ? 32 : max_q_len, W, causal, nullptr, ws.a2.p, sg ? nullptr : lo_of(ws.a2.p), st,
                                                                    sg ? 0 : 1, nullptr, sg ? 1 : 0, packed ? e->rss_cur : nullptr);
                prof_end(slot, st, 10);
                TRY(arc);
            }
            e->last_flops += 4.0 * attn_pairs * W;
            if (f16res) {
                TRY(gemm_pre(e, a2, W, b.out_w, b.out_b, nullptr, 0, nullptr, 0, h2, W, T, W, W, RLCF_EPI_NONE, st));
                LN_ADD_FWD(x, b.ln2_w, b.ln2_b, h2, T, W);
            } else {
                TRY(gemm_pre(e, a2, W, b.out_w, b.out_b, x, W, x, W, nullptr, 0, T, W, W, RLCF_EPI_NONE, st));
                LN_FWD_SPLIT(x, b.ln2_w, b.ln2_b, h2, lo_of(h2), T, W);
            }
            TRY(gemm_pre(e, h2, W, b.fc_w, b.fc_b, nullptr, 0, nullptr, 0, f2, 4 * W, T, 4 * W, W, RLCF_EPI_QUICKGELU, st));
            if (f16res) {
                TRY(gemm_pre(e, f2, 4 * W, b.proj_w, b.proj_b, nullptr, 0, nullptr, 0, h2, W, T, W, 4 * W, RLCF_EPI_NONE, st));
                have_d = true;
            } else TRY(gemm_pre(e, f2, 4 * W, b.proj_w, b.proj_b, x, W, x, W, nullptr, 0, T, W, 4 * W, RLCF_EPI_NONE, st));
        }
        if (have_d) TRY(launch_add_f16(x, h2, (int64_t)T * W, st));                 // (no class-token shortcut: every row's last product)
        if (cls_out && cls_idx) TRY(launch_gather_rows(xb, W, cls_idx, cls_out, W, n_seq, W, st));
        return RLCF_OK;
    }
    static int save_pairs = -1;                               // RLCF_SAVE_NOPAIRS=1: the saved forward back on f32 hand-overs (A/B)
    if (save_pairs < 0) { const char* ev = getenv("RLCF_SAVE_NOPAIRS"); save_pairs = ev && atoi(ev) ? 0 : 1; }
    if (save_pairs && save && prec_x3(e) && !prec_single(e) && T > 512 && W % 32 == 0) {
        // the saved forward of the tuning paths with the no-grad pipeline's hand-overs: LayerNorm and the attention kernel write the
        // next GEMM's operand pairs themselves (the attention output also as f32: the backward reads it), and c_proj's operand comes
        // from ONE pass over the saved pre-activation (QuickGELU + split) — the same values as the f32 hand-overs, four stand-alone
        // passes per layer fewer (configs[2]: 0.47 ms/image)
        TRY(x3_ensure(ws, T, W));
        for (int l = 0; l < L; ++l) {
            const BlockW& b = w.blk[l];
            float* xin = ws.sv[l].x;
            float* x1 = ws.sv[l].x1;
            float* xout = l + 1 < L ? ws.sv[l + 1].x : ws.x.as<float>();
            LN_FWD_SPLIT(xin, b.ln1_w, b.ln1_b, ws.h2.p, lo_of(ws.h2.p), T, W);
            TRY(gemm_pre(e, ws.h2.p, W, b.in_w, b.in_b, nullptr, 0, ws.sv[l].qkv, 3 * W, nullptr, 0, T, 3 * W, W, RLCF_EPI_NONE, st));
            TRY(launch_attention_fwd_x3(ws.sv[l].qkv, seqs, n_seq, max_q_len, W, causal, ws.sv[l].a, ws.a2.p, lo_of(ws.a2.p), st, 1, ws.sv[l].lse));
            e->last_flops += 4.0 * attn_pairs * W;
            TRY(gemm_pre(e, ws.a2.p, W, b.out_w, b.out_b, xin, W, x1, W, nullptr, 0, T, W, W, RLCF_EPI_NONE, st));
            LN_FWD_SPLIT(x1, b.ln2_w, b.ln2_b, ws.h2.p, lo_of(ws.h2.p), T, W);
            TRY(gemm_pre(e, ws.h2.p, W, b.fc_w, b.fc_b, nullptr, 0, ws.sv[l].f, 4 * W, nullptr, 0, T, 4 * W, W, RLCF_EPI_NONE, st));
            TRY(launch_split_f16x2(ws.sv[l].f, ws.f2.p, lo_of(ws.f2.p), (int64_t)T * 4 * W, st, 1.0f, 1, 1));
            TRY(gemm_pre(e, ws.f2.p, 4 * W, b.proj_w, b.proj_b, x1, W, xout, W, nullptr, 0, T, W, 4 * W, RLCF_EPI_NONE, st));
        }
        if (cls_out && cls_idx) TRY(launch_gather_rows(ws.x.as<float>(), W, cls_idx, cls_out, W, n_seq, W, st));
        return RLCF_OK;
    }
    for (int l = 0; l < L; ++l) {
        const BlockW& b = w.blk[l];
        float* xin = save ? ws.sv[l].x : ws.x.as<float>();
        float* x1 = save ? ws.sv[l].x1 : ws.x.as<float>();
        float* xout = (save && l + 1 < L) ? ws.sv[l + 1].x : ws.x.as<float>();
        float* qkv = save ? ws.sv[l].qkv : ws.qkv.as<float>();
        float* a = save ? ws.sv[l].a : ws.a.as<float>();
        float* h = ws.h.as<float>();
        float* f = ws.f.as<float>();
        LN_FWD(xin, b.ln1_w, b.ln1_b, h, T, W);
        TRY(gemm(e, h, W, b.in_w, W, b.in_b, nullptr, 0, nullptr, 0, qkv, 3 * W, T, 3 * W, W, 1.f, RLCF_EPI_NONE, st));
        if (prec_x3(e))      // (f32 output + log-sum-exp for the backward; the split-f16 kernel is ~3x the f32-MFMA one)
            TRY(launch_attention_fwd_x3(qkv, seqs, n_seq, max_q_len, W, causal, a, nullptr, nullptr, st, 0, save ? ws.sv[l].lse : nullptr));
        else
            TRY(launch_attention_fwd_f32(qkv, seqs, n_seq, max_q_len, W, causal, a, save ? ws.sv[l].lse : nullptr, st));
        e->last_flops += 4.0 * attn_pairs * W;
        TRY(gemm(e, a, W, b.out_w, W, b.out_b, xin, W, nullptr, 0, x1, W, T, W, W, 1.f, RLCF_EPI_NONE, st));
        LN_FWD(x1, b.ln2_w, b.ln2_b, h, T, W);
        if (save) {
            TRY(gemm(e, h, W, b.fc_w, W, b.fc_b, nullptr, 0, nullptr, 0, ws.sv[l].f, 4 * W, T, 4 * W, W, 1.f, RLCF_EPI_NONE, st));
            TRY(launch_quickgelu(ws.sv[l].f, f, (int64_t)T * 4 * W, st));
        } else {
            TRY(gemm(e, h, W, b.fc_w, W, b.fc_b, nullptr, 0, nullptr, 0, f, 4 * W, T, 4 * W, W, 1.f, RLCF_EPI_QUICKGELU, st));
        }
        TRY(gemm(e, f, 4 * W, b.proj_w, 4 * W, b.proj_b, x1, W, nullptr, 0, xout, W, T, W, 4 * W, 1.f, RLCF_EPI_NONE, st));
    }
    if (cls_out && cls_idx) TRY(launch_gather_rows(ws.x.as<float>(), W, cls_idx, cls_out, W, n_seq, W, st));
    return RLCF_OK;
}

// dX-only backward of the above (all weights frozen: TPT/tpt_cls_rl.py:103-105); dX in/out in e->dX.
// wgrad (full image-encoder tuning): flat gradient buffer of the engine's non-LayerNorm visual parameters (e->vw_slots layout);
// the Linear weight / bias gradients of every block are then formed next to the dX chain, from the same saved activations.
static int wgrad(rlcf_engine* e, const float* dY, int ldy, int N, const float* X, int ldx, int K, int T, float* dW, float* db, hipStream_t st);
static int transformer_backward(rlcf_engine* e, const TowerW& w, Tower& ws, const rlcf_seq* seqs, int n_seq, int max_keys,
                                long attn_pairs, int causal, int T, hipStream_t st, float* ln_grad = nullptr, int max_q_len = 0,
                                int group_rows = 0, int group_stride = 0, float* wgrad_base = nullptr, const VwSlot* wslots = nullptr) {
    const int W = w.width, L = w.layers;
    if (wgrad_base && !wslots) wslots = e->vw_slots.data() + 4;      // image encoder: class_embedding, positional_embedding, proj, conv1 come first
    float *dX = e->dX.as<float>(), *dA = e->dA.as<float>(), *dH = e->dH.as<float>(), *dF = e->dF.as<float>(), *dQKV = e->dQKV.as<float>();
    if (ln_grad || wgrad_base) TRY(e->parts_ws.ensure(RLCF_PARTS_WS_FLOATS * sizeof(float)));
    for (int l = L - 1; l >= 0; --l) {
        const BlockW& b = w.blk[l];
        const SavedLayer& s = ws.sv[l];
        // slots of block l: in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, c_fc.weight, c_fc.bias, c_proj.weight, c_proj.bias
        float* G[8] = {};
        if (wgrad_base) for (int i = 0; i < 8; ++i) G[i] = wgrad_base + wslots[8 * l + i].off;
        if (wgrad_base) {                  // c_proj: y = QuickGELU(f) Wp^T + b, dY = dX (gradient at the block output)
            TRY(launch_quickgelu(s.f, ws.f.as<float>(), (int64_t)T * 4 * W, st));
            TRY(wgrad(e, dX, W, W, ws.f.as<float>(), 4 * W, 4 * W, T, G[6], G[7], st));
        }
        TRY(e->bwd_amax.ensure(sizeof(float)));
        RLCF_HIP_CHECK(hipMemsetAsync(e->bwd_amax.p, 0, sizeof(float), st));
        TRY(gemm(e, dX, W, b.proj_wT, W, nullptr, nullptr, 0, s.f, 4 * W, dF, 4 * W, T, 4 * W, W, 1.f, RLCF_EPI_QUICKGELU_BWD, st, 1.0f, true,
                 nullptr, (unsigned int*)e->bwd_amax.p));                 // max|dF| comes out of the epilogue ...
        if (wgrad_base) {                  // c_fc: pre-activation = LN2(x1) Wfc^T + b
            TRY(launch_layernorm_fwd(s.x1, b.ln2_w, b.ln2_b, ws.h.as<float>(), T, W, st));
            TRY(wgrad(e, dF, 4 * W, 4 * W, ws.h.as<float>(), W, W, T, G[4], G[5], st));
        }
        TRY(gemm(e, dF, 4 * W, b.fc_wT, 4 * W, nullptr, nullptr, 0, nullptr, 0, dH, W, T, W, 4 * W, 1.f, RLCF_EPI_NONE, st, 1.0f, true,
                 e->bwd_amax.as<float>()));                               // ... and scales the next operand without another pass
        float* g1 = ln_grad ? ln_grad + (size_t)(2 + 4 * l) * W : nullptr;        // [ln_1.w | ln_1.b | ln_2.w | ln_2.b] of layer l
        const LnRef g2w = ln_ref(e, b.ln2_w, 1), g1w = ln_ref(e, b.ln1_w, 1);       // per-sample LayerNorm sets (batched LN tuning, step > 0)
        {   // profile record of kind 13: LayerNorm backward, HBM-bound — `flops` carries its ALGORITHMIC BYTES (x, dy, residual gradient in, dx out)
            const int ls = prof_begin(st, (double)T * W * 16.0, T, W, 0);
            const int lrc = launch_layernorm_bwd(s.x1, g2w.p, dH, dX, dX, g1 ? g1 + 2 * W : nullptr, g1 ? g1 + 3 * W : nullptr, T, W, st, group_rows,
                                                 group_stride, group_rows > 0 ? g2w.group_stride : 0, PARTS_WS(e));
            prof_end(ls, st, 13);
            TRY(lrc);
        }
        if (wgrad_base) TRY(wgrad(e, dX, W, W, s.a, W, W, T, G[2], G[3], st));      // out_proj: x1 = x + a Wo^T + b, dY = d x1
        TRY(gemm(e, dX, W, b.out_wT, W, nullptr, nullptr, 0, nullptr, 0, dA, W, T, W, W, 1.f, RLCF_EPI_NONE, st, 1.0f, true));
        static int bwd_f32 = -1;                                // RLCF_ATTN_BWD_F32=1: the f32-MFMA backward also in split-f16 mode (benchmarks)
        if (bwd_f32 < 0) { const char* ev = getenv("RLCF_ATTN_BWD_F32"); bwd_f32 = ev ? atoi(ev) : 0; }
        // profile record of kind 12: the attention backward (flops = 10 * pairs * W; dims = rows, width, longest sequence)
        const int pslot = prof_begin(st, 10.0 * attn_pairs * W, T, W, max_q_len > 0 ? max_q_len : max_keys);
        // image towers (no shared prefix, max_q_len given): dK / dV parked per query block and added in block order (bit-reproducible,
        // and dQKV needs no zero fill); beyond 16 GB of parking space (or RLCF_ATTN_BWD_ATOMIC=1) they meet by atomicAdd
        static int bwd_atomic = -1;
        if (bwd_atomic < 0) { const char* ev = getenv("RLCF_ATTN_BWD_ATOMIC"); bwd_atomic = ev ? atoi(ev) : 0; }
        float* park = nullptr;
        static int bwd_old0 = -1;
        if (bwd_old0 < 0) { const char* ev = getenv("RLCF_ATTN_BWD_OLD"); bwd_old0 = ev ? atoi(ev) : 0; }
        // (the f32-MFMA backward — RLCF_PREC_F32, or RLCF_ATTN_BWD_F32=1 — parks too since round 5: its float atomics were the last source
        // of run-to-run differences in that mode)
        const bool f32_form = max_keys > 96 && (!prec_x3(e) || bwd_f32);
        if (((bwd_old0 && prec_x3(e) && !bwd_f32) || f32_form) && max_keys > 96 && !causal && max_q_len > 0 && max_q_len == max_keys && !bwd_atomic) {
            const size_t need = (size_t)n_seq * ((max_q_len + 31) / 32) * max_q_len * 2 * W * sizeof(float);
            if (need <= ((size_t)16 << 30)) {
                // no room for the parking space (another engine holds the memory): clear the error and meet by atomicAdd instead
                if (e->attn_park.ensure(need) == RLCF_OK) park = e->attn_park.as<float>();
                else (void)hipGetLastError();
            }
        }
        // ... and since round 4 the two-kernel form (attention_bwd_x3b.hip): dQ per 64 queries, dK / dV per 64 keys with the accumulators
        // kept in registers across all query blocks — single writers, nothing parked.  RLCF_ATTN_BWD_OLD=1 switches back (A/B)
        static int bwd_old = -1;
        if (bwd_old < 0) { const char* ev = getenv("RLCF_ATTN_BWD_OLD"); bwd_old = ev ? atoi(ev) : 0; }
        const bool split_form = max_keys > 96 && prec_x3(e) && !bwd_f32 && !causal && max_q_len > 0 && max_q_len == max_keys && !bwd_atomic && !bwd_old;
        if (split_form) {
            TRY(launch_absmax(dA, (int64_t)T * W, e->bwd_amax.as<float>(), st));
            TRY(launch_attention_bwd_x3_split(s.qkv, s.a, s.lse, dA, e->bwd_amax.as<float>(), seqs, n_seq, max_q_len, W, dQKV, st));
        } else {
        if (!park) RLCF_HIP_CHECK(hipMemsetAsync(dQKV, 0, (size_t)T * 3 * W * sizeof(float), st));
        if (max_keys > 96 && prec_x3(e) && !bwd_f32) {
            TRY(launch_absmax(dA, (int64_t)T * W, e->bwd_amax.as<float>(), st));       // range of dO for the f16 pairs
            TRY(launch_attention_bwd_x3(s.qkv, s.a, s.lse, dA, e->bwd_amax.as<float>(), seqs, n_seq, max_q_len > 0 ? max_q_len : max_keys, W, causal,
                                        dQKV, st, park));
        } else if (max_keys > 96) TRY(launch_attention_bwd_mfma(s.qkv, s.a, s.lse, dA, seqs, n_seq, max_q_len > 0 ? max_q_len : max_keys, W, causal, dQKV, st, park));
        else {
            // shared-prefix layouts: the prefix rows' dK / dV are summed in sequence order through a per-sequence workspace (reproducible
            // bit for bit); sized on first use like the other lazily grown scratch, atomics beyond 512 MB (dense backward of a huge bank)
            const size_t need = (size_t)n_seq * max_keys * 2 * W;
            float* pws = nullptr;
            if (need * sizeof(float) <= ((size_t)512 << 20)) { TRY(e->attn_pre_ws.ensure(need * sizeof(float))); pws = e->attn_pre_ws.as<float>(); }
            TRY(launch_attention_bwd(s.qkv, dA, seqs, n_seq, max_keys, W, causal, dQKV, st, pws, pws ? need : 0, max_keys));
        }
        }
        prof_end(pslot, st, 12);
        e->last_flops += 10.0 * attn_pairs * W;
        if (wgrad_base) {                  // in_proj: qkv = LN1(x) Win^T + b
            TRY(launch_layernorm_fwd(s.x, b.ln1_w, b.ln1_b, ws.h.as<float>(), T, W, st));
            TRY(wgrad(e, dQKV, 3 * W, 3 * W, ws.h.as<float>(), W, W, T, G[0], G[1], st));
        }
        TRY(gemm(e, dQKV, 3 * W, b.in_wT, 3 * W, nullptr, nullptr, 0, nullptr, 0, dH, W, T, W, 3 * W, 1.f, RLCF_EPI_NONE, st, 1.0f, true));
        {
            const int ls = prof_begin(st, (double)T * W * 16.0, T, W, 0);
            const int lrc = launch_layernorm_bwd(s.x, g1w.p, dH, dX, dX, g1, g1 ? g1 + W : nullptr, T, W, st, group_rows, group_stride,
                                                 group_rows > 0 ? g1w.group_stride : 0, PARTS_WS(e));
            prof_end(ls, st, 13);
            TRY(lrc);
        }
    }
    return RLCF_OK;
}

// ------------------------------------------------------------------ image tower
// VisionTransformer.forward (TPT/clip/model.py:223-240) + L2 normalise (custom_clip.py:330,
// clip_reward.py:136).  Runs under no_grad in the reference (custom_clip.py:325-327).
int engine_encode_image(rlcf_engine* e, int which, const float* images, int n, float* feats, hipStream_t st, int in_res) {
    ClipModel& m = e->model[which];
    if (!m.finalized) { rlcf_set_error("model %d not finalized", which); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(n > 0 && n <= e->max_views);
    const rlcf_clip_cfg& c = m.cfg;
    if (in_res > 0 && in_res != c.image_resolution) {
        // the model wants another input size than the views have: bicubic, align_corners=True (clip_reward.py:133-134)
        TRY(IMG_BUF(e, resized).ensure((size_t)e->max_views * 3 * c.image_resolution * c.image_resolution * sizeof(float)));
        TRY(launch_bicubic(images, IMG_BUF(e, resized).as<float>(), n * 3, in_res, c.image_resolution, st));
        images = IMG_BUF(e, resized).as<float>();
    }
    if (is_resnet(c)) return resnet_encode(e, m, images, n, feats, st);
    const int Wv = c.vision_width, tok = m.tokens, G2 = tok - 1, T = n * tok, D = c.embed_dim;
    if (e->ws_sel) {               // side-stream scratch grows on demand (first call of a size only)
        auto& sb = e->side_img;
        TRY(sb.patch_out.ensure((size_t)n * G2 * Wv * sizeof(float))); TRY(sb.patches.ensure((size_t)n * G2 * m.Kp * sizeof(float)));
        TRY(sb.cls_rows.ensure((size_t)n * Wv * sizeof(float))); TRY(sb.cls_ln.ensure((size_t)n * Wv * sizeof(float)));
        TRY(sb.feat_raw.ensure((size_t)n * D * sizeof(float)));
        TRY(tower_ensure(sb.vt, T, Wv, st));
    }
    if (prec_x3(e) && n * G2 > 512 && (size_t)n * G2 * m.Kp <= a_cap(e)) {
        const bool sg = prec_single(e) && m.Kp % 64 == 0 && m.f16_of.count(m.conv_w);
        TRY(launch_im2col(images, nullptr, a_ptr(e), sg ? nullptr : lo_of(a_ptr(e)), n, c.image_resolution, c.vision_patch_size, m.Kp, st, sg ? 0 : 1));
        TRY(gemm_pre(e, a_ptr(e), m.Kp, m.conv_w, nullptr, nullptr, 0, IMG_BUF(e, patch_out).as<float>(), Wv, nullptr, 0, n * G2, Wv,
                     m.Kp, RLCF_EPI_NONE, st));
    } else {
        TRY(launch_im2col(images, IMG_BUF(e, patches).as<float>(), nullptr, nullptr, n, c.image_resolution, c.vision_patch_size, m.Kp, st));
        TRY(gemm(e, IMG_BUF(e, patches).as<float>(), m.Kp, m.conv_w, m.Kp, nullptr, nullptr, 0, nullptr, 0, IMG_BUF(e, patch_out).as<float>(), Wv,
                 n * G2, Wv, m.Kp, 1.f, RLCF_EPI_NONE, st));
    }
    {
        const LnRef gw = ln_ref(e, m.lnpre_w, 1), gb = ln_ref(e, m.lnpre_b, 1);
        TRY(launch_vit_assemble(IMG_BUF(e, patch_out).as<float>(), m.cls, m.vpos, gw.p, gb.p, IMG_BUF(e, vt).x.as<float>(), n, tok, Wv, st, gw.group_rows, gw.group_stride));
    }
    // class-token rows come out compact: the last block is evaluated for them only (transformer_forward)
    // RLCF_PREC_F16, large passes: the blocks run over CHUNKS of views (all 12 blocks on one chunk, then the next): a view's rows never
    // meet another view's, and a chunk's hand-over tensors (Q/K/V, the MLP's hidden rows: 58 + 77 MB per ViT-B/16 test image in f16) are
    // then still in the 256-MB Infinity Cache when the next kernel reads them instead of coming back from HBM.  RLCF_F16_CHUNK_VIEWS=n
    // sets the chunk (0 = one chunk)
    static int chunk_views = -1;
    if (chunk_views < 0) { const char* ev = getenv("RLCF_F16_CHUNK_VIEWS"); chunk_views = ev ? atoi(ev) : 0; }
    const int cv = (prec_single(e) && prec_x3(e) && chunk_views > 0 && T > 512) ? chunk_views : n;
    if (cv < n) TRY(x3_ensure(IMG_BUF(e, vt), T, Wv));                    // (the buffers must not move between chunks)
    for (int s0 = 0; s0 < n; s0 += cv) {
        const int ns = std::min(cv, n - s0);
        const size_t so = (size_t)which * e->max_views + s0;
        TRY(transformer_forward(e, m.vis, IMG_BUF(e, vt), e->vit_seqs.as<rlcf_seq>() + so, ns, tok, (long)ns * tok * tok, 0, ns * tok,
                                false, st, e->vit_seqs_cls.as<rlcf_seq>() + so, e->vit_cls_idx.as<int32_t>() + so,
                                IMG_BUF(e, cls_rows).as<float>() + (size_t)s0 * Wv, s0 * tok));
    }
    {
        const LnRef gw = ln_ref(e, m.lnpost_w, 1), gb = ln_ref(e, m.lnpost_b, 1);      // one class-token row per view
        TRY(launch_layernorm_fwd(IMG_BUF(e, cls_rows).as<float>(), gw.p, gb.p, IMG_BUF(e, cls_ln).as<float>(), n, Wv, st, gw.group_rows, gw.group_stride));
    }
    TRY(gemm(e, IMG_BUF(e, cls_ln).as<float>(), Wv, m.vprojT, Wv, nullptr, nullptr, 0, nullptr, 0, IMG_BUF(e, feat_raw).as<float>(), D, n, D, Wv, 1.f,
             RLCF_EPI_NONE, st));
    TRY(launch_l2norm_rows(IMG_BUF(e, feat_raw).as<float>(), feats, nullptr, n, D, st));
    return RLCF_OK;
}

// ------------------------------------------------------------------ text layouts
__global__ void build_E_kernel(const float* __restrict__ tok_emb, const float* __restrict__ pos, const int32_t* __restrict__ row_token,
                               const int32_t* __restrict__ row_pos, float* __restrict__ E, int rows, int width) {
    const int r = blockIdx.x;
    const int tkn = row_token[r], ps = row_pos[r];
    for (int c = threadIdx.x; c < width; c += blockDim.x)
        E[(size_t)r * width + c] = pos[(size_t)ps * width + c] + (tkn >= 0 ? tok_emb[(size_t)tkn * width + c] : 0.f);
}

template <typename T>
static int upload(DevBuf& d, const std::vector<T>& v, hipStream_t st) {
    TRY(d.ensure(std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) RLCF_HIP_CHECK(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    RLCF_HIP_CHECK(hipStreamSynchronize(st));     // v is a host temporary
    return RLCF_OK;
}

// Builds the packed layout of a class bank.  tokens: HOST [C, L] as clip.tokenize produces
// (TPT/clip/clip.py:197-233); EOT = argmax id (custom_clip.py:71).  n_ctx > 0: rows 1..n_ctx of
// every prompt are the learnable context (custom_clip.py:198-238).
// ctx_pos (student only, HOST [C, n_ctx]): position of learnable vector k in prompt c when the class tokens are not at the END of the
// prompt (PromptLearner.forward with class_token_position 'front' / 'middle', custom_clip.py:239-284); `tokens` then holds the token ids
// in that re-arranged order (anything at the learnable positions).  NULL: learnable vector k sits at position 1 + k.
static int build_layout(rlcf_engine* e, ClipModel& m, TextLayout& L, const int32_t* tokens, int C, int n_ctx, bool has_ctx, int mode,
                        hipStream_t st, const int32_t* ctx_pos = nullptr) {
    const int CL = m.cfg.context_length, Wt = m.cfg.text_width;
    std::vector<int> eot(C);
    for (int c = 0; c < C; ++c) {
        int best = 0;
        for (int j = 1; j < CL; ++j) if (tokens[(size_t)c * CL + j] > tokens[(size_t)c * CL + best]) best = j;
        eot[c] = best;
    }
    // which learnable vector (or -1) sits at position j of prompt c
    auto ctx_at = [&](int c, int j) -> int {
        if (!has_ctx) return -1;
        if (!ctx_pos) return (j >= 1 && j <= n_ctx) ? j - 1 : -1;
        for (int k = 0; k < n_ctx; ++k) if (ctx_pos[(size_t)c * n_ctx + k] == j) return k;
        return -1;
    };
    const bool general = has_ctx && ctx_pos != nullptr;
    int pre = 0;
    if (mode == RLCF_TEXT_SHARED && !general) {
        pre = 1 + n_ctx;
        bool ok = true;
        for (int c = 0; c < C && ok; ++c) {
            if (eot[c] < pre) ok = false;
            for (int j = 0; j < pre && ok; ++j) {
                const bool is_ctx = has_ctx && j >= 1;
                if (!is_ctx && tokens[(size_t)c * CL + j] != tokens[j]) ok = false;
            }
        }
        if (!ok) { pre = 0; mode = RLCF_TEXT_PACKED; }
    } else if (mode == RLCF_TEXT_SHARED) {
        // longest run of leading positions that is the same row for every class ('front': SOS only; 'middle': SOS + the first half of ctx)
        int min_eot = CL;
        for (int c = 0; c < C; ++c) min_eot = std::min(min_eot, eot[c]);
        while (pre < min_eot) {
            const int k0 = ctx_at(0, pre);
            bool same = true;
            for (int c = 1; c < C && same; ++c) same = ctx_at(c, pre) == k0 && (k0 >= 0 || tokens[(size_t)c * CL + pre] == tokens[pre]);
            if (!same) break;
            ++pre;
        }
        if (pre == 0) mode = RLCF_TEXT_PACKED;
    }
    std::vector<int32_t> row_token, row_pos, ctx_row, class_start(C), class_len(C), class_eot_off(C), eot_rows(C), ctx_list;
    std::vector<rlcf_seq> seqs;
    auto push_row = [&](int token, int pos_idx, int cr) { row_token.push_back(token); row_pos.push_back(pos_idx); ctx_row.push_back(cr); };
    for (int j = 0; j < pre; ++j) {
        const int k = ctx_at(0, j);
        push_row(k >= 0 ? -1 : tokens[j], j, k);
    }
    long pairs = 0;
    int lmax = 0;
    L.tokens_total = pre;
    for (int c = 0; c < C; ++c) {
        const int first = pre, last = (mode == RLCF_TEXT_DENSE) ? CL - 1 : eot[c];
        class_start[c] = (int)row_token.size();
        class_len[c] = last - first + 1;
        class_eot_off[c] = eot[c] - first;
        eot_rows[c] = class_start[c] + class_eot_off[c];
        for (int j = first; j <= last; ++j) {
            const int k = ctx_at(c, j);
            push_row(k >= 0 ? -1 : tokens[(size_t)c * CL + j], j, k);
        }
        seqs.push_back(rlcf_seq{class_start[c], class_len[c], 0, pre});
        lmax = std::max(lmax, class_len[c]);
        for (int i = 0; i < class_len[c]; ++i) pairs += pre + i + 1;
        L.tokens_total += class_len[c];
        if (has_ctx && pre == 0 && !general) for (int j = 0; j < n_ctx; ++j) ctx_list.push_back(class_start[c] + 1 + j);
    }
    if (pre > 0) {
        seqs.push_back(rlcf_seq{0, pre, 0, 0});
        for (int i = 0; i < pre; ++i) pairs += i + 1;
        if (has_ctx && !general) for (int j = 0; j < n_ctx; ++j) ctx_list.push_back(1 + j);
    }
    // packed runs (shared-prefix layouts only): consecutive class sequences, whole, while the run stays within 32 - pre rows
    {
        std::vector<rlcf_seq> pk;
        std::vector<int32_t> rss(row_token.size(), 0);
        if (pre > 0 && pre < 24) {
            const int cap = 32 - pre;
            int c = 0;
            while (c < C) {
                const int r0 = class_start[c];
                int rows = 0, c1 = c;
                while (c1 < C && class_len[c1] <= cap && rows + class_len[c1] <= cap && class_start[c1] == r0 + rows) { rows += class_len[c1]; ++c1; }
                if (c1 == c) { pk.clear(); break; }                       // a sequence longer than the cap: no packing for this bank
                for (int k = c; k < c1; ++k) for (int i = 0; i < class_len[k]; ++i) rss[class_start[k] + i] = class_start[k];
                pk.push_back(rlcf_seq{r0, rows, 0, pre});
                c = c1;
            }
            if (!pk.empty()) pk.push_back(rlcf_seq{0, pre, 0, 0});       // the prefix itself (rss = 0: one sequence starting at row 0)
        }
        L.n_pk = (int)pk.size();
        TRY(upload(L.pk_seqs, pk, st)); TRY(upload(L.pk_rss, rss, st));
    }
    L.ctx_general = general;
    L.T = (int)row_token.size(); L.C = C; L.n_seq = (int)seqs.size(); L.pre_rows = pre; L.lmax = lmax;
    L.max_q_len = std::max(lmax, pre); L.max_keys = pre + lmax; L.n_ctx = has_ctx ? n_ctx : 0;
    L.n_copies = has_ctx ? (pre > 0 ? 1 : C) : 0;
    L.attn_pairs = pairs;
    TRY(upload(L.seqs, seqs, st)); TRY(upload(L.eot_rows, eot_rows, st)); TRY(upload(L.ctx_row, ctx_row, st));
    TRY(upload(L.class_start, class_start, st)); TRY(upload(L.class_len, class_len, st));
    TRY(upload(L.class_eot_off, class_eot_off, st)); TRY(upload(L.ctx_rows_list, ctx_list, st));
    TRY(upload(L.row_token, row_token, st)); TRY(upload(L.row_pos, row_pos, st));
    TRY(L.E.ensure((size_t)L.T * Wt * sizeof(float)));
    build_E_kernel<<<dim3(L.T), dim3(128), 0, st>>>(m.tok_emb, m.tpos, L.row_token.as<int32_t>(), L.row_pos.as<int32_t>(), L.E.as<float>(), L.T, Wt);
    RLCF_LAUNCH_CHECK();
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    double mean_len = 0;
    for (int c = 0; c < C; ++c) mean_len += class_len[c];
    L.mean_len = mean_len / C;
    return RLCF_OK;
}

// Text tower over a layout: TextEncoder.forward (custom_clip.py:62-73) / CLIP.encode_text
// (model.py:343-356), then L2 normalise (custom_clip.py:320).  row_src != null: sparse re-pack.
struct TextPassIO {
    const rlcf_seq* seqs; int n_seq, max_q_len, T, n_cls; long attn_pairs;
    const int32_t* eot_rows; const int32_t* row_src;
    float *eot_x, *eot_ln, *u, *inv_norm, *txt;
    int rep_rows = 0, ctx_stride = 0;      // replicated layout: one replica (and one prompt) per test sample
    const int32_t* ctx_row_tab = nullptr;  // set when the layout has its learnable rows at class-dependent positions (TextLayout::ctx_general)
    const rlcf_seq* pk_seqs = nullptr; int n_pk = 0; const int32_t* pk_rss = nullptr;      // packed runs of the same sequences (no-grad passes)
};
static int text_forward(rlcf_engine* e, ClipModel& m, const TextLayout& L, Tower& ws, const float* ctx, const TextPassIO& io, bool save,
                        hipStream_t st) {
    const int Wt = m.cfg.text_width, D = m.cfg.embed_dim;
    float* x0 = save ? ws.sv[0].x : ws.x.as<float>();
    TRY(launch_text_assemble(L.E.as<float>(), io.row_src, L.ctx_row.as<int32_t>(), ctx, x0, io.T, Wt, io.rep_rows, io.ctx_stride, st));
    if (!save) { e->pk_cur = io.pk_seqs; e->n_pk_cur = io.n_pk; e->rss_cur = io.pk_rss; }
    const int rc_tf = transformer_forward(e, m.txt, ws, io.seqs, io.n_seq, io.max_q_len, io.attn_pairs, 1, io.T, save, st);
    e->pk_cur = nullptr; e->n_pk_cur = 0; e->rss_cur = nullptr;
    TRY(rc_tf);
    TRY(launch_gather_rows(ws.x.as<float>(), Wt, io.eot_rows, io.eot_x, Wt, io.n_cls, Wt, st));
    TRY(launch_layernorm_fwd(io.eot_x, m.lnf_w, m.lnf_b, io.eot_ln, io.n_cls, Wt, st));
    TRY(gemm(e, io.eot_ln, Wt, m.tprojT, Wt, nullptr, nullptr, 0, nullptr, 0, io.u, D, io.n_cls, D, Wt, 1.f, RLCF_EPI_NONE, st));
    TRY(launch_l2norm_rows(io.u, io.txt, io.inv_norm, io.n_cls, D, st));
    return RLCF_OK;
}
// Backward of text_forward w.r.t. ctx, given dtxt [n_cls, D] (autograd's work at tpt_cls_rl.py:77).
static int text_backward(rlcf_engine* e, ClipModel& m, Tower& ws, const TextPassIO& io, int max_keys, const float* dtxt, float* du,
                         float* dxe, const int32_t* ctx_rows_list, int n_copies, int n_ctx, float* dctx, hipStream_t st) {
    const int Wt = m.cfg.text_width, D = m.cfg.embed_dim;
    TRY(launch_l2norm_bwd(io.txt, dtxt, io.inv_norm, du, io.n_cls, D, st));
    TRY(gemm(e, du, D, m.tproj, D, nullptr, nullptr, 0, nullptr, 0, dxe, Wt, io.n_cls, Wt, D, 1.f, RLCF_EPI_NONE, st));
    TRY(launch_layernorm_bwd(io.eot_x, m.lnf_w, dxe, nullptr, dxe, nullptr, nullptr, io.n_cls, Wt, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->dX.p, 0, (size_t)io.T * Wt * sizeof(float), st));
    TRY(launch_scatter_rows(dxe, io.eot_rows, e->dX.as<float>(), io.n_cls, Wt, st));
    TRY(transformer_backward(e, m.txt, ws, io.seqs, io.n_seq, max_keys, io.attn_pairs, 1, io.T, st));
    if (io.ctx_row_tab) TRY(launch_ctx_grad_scan(e->dX.as<float>(), io.row_src, io.ctx_row_tab, 1, io.T, n_ctx, Wt, dctx, st));
    else TRY(launch_ctx_grad(e->dX.as<float>(), ctx_rows_list, n_copies, n_ctx, Wt, dctx, st));
    return RLCF_OK;
}

static TextPassIO full_io(rlcf_engine* e, const TextLayout& L) {
    TextPassIO io{};
    io.seqs = L.seqs.as<rlcf_seq>(); io.n_seq = L.n_seq; io.max_q_len = L.max_q_len; io.T = L.T; io.n_cls = L.C;
    io.attn_pairs = L.attn_pairs; io.eot_rows = L.eot_rows.as<int32_t>(); io.row_src = nullptr;
    io.eot_x = e->eot_x.as<float>(); io.eot_ln = e->eot_ln.as<float>(); io.u = e->u.as<float>();
    io.inv_norm = e->inv_norm.as<float>(); io.txt = e->txt.as<float>();
    io.ctx_row_tab = L.ctx_general ? L.ctx_row.as<int32_t>() : nullptr;
    if (L.n_pk > 0) { io.pk_seqs = L.pk_seqs.as<rlcf_seq>(); io.n_pk = L.n_pk; io.pk_rss = L.pk_rss.as<int32_t>(); }
    return io;
}

// per-step scratch of the TTA calls for a bank of C entries (class prompts, captions or images)
static int tta_scratch_ensure(rlcf_engine* e, int C) {
    ClipModel& s = e->model[RLCF_STUDENT];
    const int D = s.cfg.embed_dim;
    const int N = e->max_views;
    TRY(e->img_feat.ensure((size_t)N * D * sizeof(float))); TRY(e->logits.ensure((size_t)N * C * sizeof(float)));
    TRY(e->entropy.ensure(N * sizeof(float))); TRY(e->sel_idx.ensure(N * sizeof(int32_t)));
    TRY(e->topk_idx.ensure((size_t)N * 32 * sizeof(int32_t))); TRY(e->clip_score.ensure((size_t)N * 32 * sizeof(float)));
    TRY(e->rewards.ensure((size_t)N * 32 * sizeof(float))); TRY(e->loss.ensure(sizeof(float)));
    TRY(e->dlogits.ensure((size_t)N * C * sizeof(float))); TRY(e->final_logits.ensure((size_t)C * sizeof(float)));
    TRY(e->top5.ensure(5 * sizeof(int32_t))); TRY(e->sel_feat.ensure((size_t)N * D * sizeof(float)));
    TRY(e->sel_logits.ensure((size_t)N * C * sizeof(float)));
    TRY(e->rl_stats.ensure(reward_loss_stats_floats(N) * sizeof(float)));          // scratch of the loss kernel: at most N selected rows
    TRY(e->step_skip.ensure((size_t)std::max(N, 64) * sizeof(int32_t)));             // non-finite-gradient flags, one per test sample of a pass
    if (e->n_rewards > 0) {
        const int R = s.cfg.image_resolution;                // selected views are kept at the student's resolution
        TRY(e->views_sel.ensure((size_t)N * 3 * R * R * sizeof(float)));
    }
    return RLCF_OK;
}

int engine_set_class_bank(rlcf_engine* e, const int32_t* tokens, int C, int n_ctx, const float* ctx_init, int text_mode, hipStream_t st,
                          const int32_t* student_tokens, const int32_t* ctx_pos) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (!s.finalized) { rlcf_set_error("student not finalized"); return RLCF_ERR_STATE; }
    // n_ctx == 0: a bank of plain texts without learnable rows (the caption bank of the retrieval task, the raw class prompts of
    // CLIPCLS_TTA): the image-encoder tuning calls use it; the prompt-tuning calls refuse it
    RLCF_ARG_CHECK(C > 0 && C <= e->max_classes && n_ctx >= 0 && n_ctx + 3 <= s.cfg.context_length && (n_ctx == 0 || ctx_init));
    RLCF_ARG_CHECK(text_mode >= RLCF_TEXT_DENSE && text_mode <= RLCF_TEXT_SHARED);
    e->text_mode = text_mode; e->n_ctx = n_ctx; e->C = C; e->image_bank = false;
    const int Wt = s.cfg.text_width, D = s.cfg.embed_dim;
    RLCF_ARG_CHECK((ctx_pos == nullptr) == (student_tokens == nullptr) && (!ctx_pos || n_ctx > 0));
    TRY(build_layout(e, s, e->lay[0], ctx_pos ? student_tokens : tokens, C, n_ctx, true, text_mode, st, ctx_pos));
    int Tmax = e->lay[0].T, Wmax = Wt, Dmax = D;
    for (int m = 0; m < e->n_rewards; ++m) {
        ClipModel& r = e->model[RLCF_REWARD + m];
        if (!r.finalized) { rlcf_set_error("reward model %d not finalized", m); return RLCF_ERR_STATE; }
        RLCF_ARG_CHECK(r.cfg.context_length == s.cfg.context_length);
        TRY(build_layout(e, r, e->lay[1 + m], tokens, C, n_ctx, false, text_mode, st));
        Tmax = std::max(Tmax, e->lay[1 + m].T); Wmax = std::max(Wmax, r.cfg.text_width); Dmax = std::max(Dmax, r.cfg.embed_dim);
    }
    TRY(tower_ensure(e->tt, Tmax, Wmax, st));
    if (prec_x3(e) && (size_t)Tmax * Wmax * 4 > e->a_split_elems) {
        e->a_split_elems = (size_t)Tmax * Wmax * 4;
        TRY(e->a_hi.ensure(e->a_split_elems * 4));
    }
    const size_t cw = (size_t)C * Wmax * sizeof(float), cd = (size_t)C * Dmax * sizeof(float);
    TRY(e->eot_x.ensure(cw)); TRY(e->eot_ln.ensure(cw)); TRY(e->u.ensure(cd)); TRY(e->inv_norm.ensure(C * sizeof(float)));
    TRY(e->txt.ensure(cd)); TRY(e->dtxt_dense.ensure(cd));
    const size_t cb = (size_t)n_ctx * Wt * sizeof(float);
    if (cb) {
        TRY(e->ctx_init.ensure(cb)); TRY(e->ctx.ensure(cb)); TRY(e->adam_m.ensure(cb)); TRY(e->adam_v.ensure(cb)); TRY(e->ctx_grad.ensure(cb));
        RLCF_HIP_CHECK(hipMemcpyAsync(e->ctx_init.p, ctx_init, cb, hipMemcpyDeviceToDevice, st));
    }
    TRY(tta_scratch_ensure(e, C));
    const int N = e->max_views;
    for (int m = 0; m < e->n_rewards; ++m) {
        ClipModel& r = e->model[RLCF_REWARD + m];
        const int Dr = r.cfg.embed_dim;
        TRY(e->rimg[m].ensure((size_t)N * Dr * sizeof(float)));
        TRY(e->reward_cls[m].ensure((size_t)C * Dr * sizeof(float)));
        // BaseRewards.set_class_features (clip_reward.py:55-57,139-150,272-289): once per class bank, per reward model
        TextPassIO io = full_io(e, e->lay[1 + m]);
        io.txt = e->reward_cls[m].as<float>();
        TRY(text_forward(e, r, e->lay[1 + m], e->tt, nullptr, io, false, st));
    }
    // Text features of the pristine prompt: every test sample starts from ctx_init (model.reset(),
    // tpt_cls_rl.py:251-253), so the first-step text forward is sample-independent -> computed once here.
    TRY(e->txt0.ensure(cd));
    {
        TextPassIO io0 = full_io(e, e->lay[0]);
        io0.txt = e->txt0.as<float>();
        TRY(text_forward(e, s, e->lay[0], e->tt, e->ctx_init.as<float>(), io0, false, st));
    }
    TRY(e->txt0T.ensure(cd));
    TRY(launch_transpose(e->txt0.as<float>(), e->txt0T.as<float>(), C, D, st));
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    e->sp_max_e = 0; e->sp_groups = 0; e->b_cap = 0;   // sparse / batch layouts are (re)built lazily
    return RLCF_OK;
}

int engine_text_features(rlcf_engine* e, int which, const float* ctx, float* txt, hipStream_t st) {
    ClipModel& m = e->model[which];
    if (e->C <= 0 || e->image_bank) { rlcf_set_error("class bank not set"); return RLCF_ERR_STATE; }
    TextPassIO io = full_io(e, e->lay[which]);
    if (txt) io.txt = txt;
    return text_forward(e, m, e->lay[which], e->tt, ctx, io, false, st);
}

int engine_logits(rlcf_engine* e, const float* img, int n, const float* txt, int C, float* logits, hipStream_t st) {
    ClipModel& m = e->model[RLCF_STUDENT];
    const int D = m.cfg.embed_dim;
    return gemm(e, img, D, txt, D, nullptr, nullptr, 0, nullptr, 0, logits, C, n, C, D, m.logit_scale_exp, RLCF_EPI_NONE, st);
}

// Dense backward: forward with saved activations over the full layout, then backward.
int engine_text_backward_dense(rlcf_engine* e, const float* ctx, const float* img, int n, const float* dlogits, float* dctx, hipStream_t st) {
    ClipModel& m = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    if (e->C <= 0 || e->image_bank) { rlcf_set_error("class bank not set"); return RLCF_ERR_STATE; }
    const int Wt = m.cfg.text_width, D = m.cfg.embed_dim;
    TRY(tower_ensure_saved(e->tt, L.T, Wt, m.cfg.text_layers, st));
    TRY(bwd_ensure(e, L.T, Wt));
    TRY(e->sp_du.ensure((size_t)L.C * D * sizeof(float)));
    TRY(e->sp_dxe.ensure((size_t)L.C * Wt * sizeof(float)));
    TextPassIO io = full_io(e, L);
    TRY(text_forward(e, m, L, e->tt, ctx, io, true, st));
    TRY(launch_dtxt_dense(dlogits, img, n, L.C, D, m.logit_scale_exp, e->dtxt_dense.as<float>(), st));
    TRY(text_backward(e, m, e->tt, io, L.max_keys, e->dtxt_dense.as<float>(), e->sp_du.as<float>(), e->sp_dxe.as<float>(),
                      L.ctx_rows_list.as<int32_t>(), L.n_copies, L.n_ctx, dctx, st));
    return RLCF_OK;
}

// The per-path parts of this translation unit (they share the static helpers above: one TU, three files by path)
#include "engine_prompt.inl"        // sparse class passes, engine_tta_sample, engine_tta_batch
#include "engine_tuning.inl"        // LayerNorm / BatchNorm / every-parameter tuning of the image encoder
#include "engine_retrieval.inl"     // text-encoder tuning (retrieval, text -> image)
