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

// Sparse backward layout for n_e = n_sel*K sampled (view, class) pairs.
static int sparse_ensure(rlcf_engine* e, int n_e_per_group, hipStream_t st, int groups = 1) {
    if (n_e_per_group <= e->sp_max_e && groups <= e->sp_groups) return RLCF_OK;
    ClipModel& m = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    const int Wt = m.cfg.text_width, D = m.cfg.embed_dim;
    groups = std::max(groups, e->sp_groups);
    n_e_per_group = std::max(n_e_per_group, e->sp_max_e);
    const int T = groups * (L.pre_rows + n_e_per_group * L.lmax);
    const int n_e = groups * n_e_per_group;
    TRY(e->sp_seqs.ensure((size_t)(n_e + groups) * sizeof(rlcf_seq))); TRY(e->sp_eot_rows.ensure(n_e * sizeof(int32_t)));
    TRY(e->sp_row_src.ensure((size_t)T * sizeof(int32_t)));
    std::vector<int32_t> list;
    if (L.pre_rows > 0) for (int j = 0; j < L.n_ctx; ++j) list.push_back(1 + j);
    else for (int k = 0; k < n_e_per_group; ++k) for (int j = 0; j < L.n_ctx; ++j) list.push_back(k * L.lmax + 1 + j);
    TRY(upload(e->sp_ctx_rows_list, list, st));
    TRY(e->sp_dtxt.ensure((size_t)n_e * D * sizeof(float))); TRY(e->sp_txt.ensure((size_t)n_e * D * sizeof(float)));
    TRY(e->sp_inv_norm.ensure(n_e * sizeof(float))); TRY(e->sp_eot_x.ensure((size_t)n_e * Wt * sizeof(float)));
    TRY(e->sp_eot_ln.ensure((size_t)n_e * Wt * sizeof(float))); TRY(e->sp_u.ensure((size_t)n_e * D * sizeof(float)));
    TRY(e->sp_du.ensure((size_t)std::max(n_e, L.C) * D * sizeof(float))); TRY(e->sp_dxe.ensure((size_t)std::max(n_e, L.C) * Wt * sizeof(float)));
    TRY(tower_ensure(e->st, T, Wt, st));
    TRY(tower_ensure_saved(e->st, T, Wt, m.cfg.text_layers, st));
    TRY(bwd_ensure(e, T, Wt));
    e->sp_max_e = n_e_per_group; e->sp_T = T; e->sp_groups = groups;
    return RLCF_OK;
}

// the two halves of the sparse pass: the forward over the sampled (view, class) pairs needs only their class indices (top-K of the
// student's own logits) — the one-image call runs it next to the reward models' tower pass —, the backward needs the rewards
static TextPassIO sparse_io(rlcf_engine* e, int n_e) {
    const TextLayout& L = e->lay[0];
    TextPassIO io{};
    io.seqs = e->sp_seqs.as<rlcf_seq>(); io.n_seq = n_e + (L.pre_rows > 0 ? 1 : 0); io.max_q_len = L.max_q_len; io.T = L.pre_rows + n_e * L.lmax;
    io.n_cls = n_e;
    io.attn_pairs = (long)(n_e * (L.mean_len * (L.pre_rows + (L.mean_len + 1) * 0.5)));
    io.eot_rows = e->sp_eot_rows.as<int32_t>(); io.row_src = e->sp_row_src.as<int32_t>();
    io.eot_x = e->sp_eot_x.as<float>(); io.eot_ln = e->sp_eot_ln.as<float>(); io.u = e->sp_u.as<float>();
    io.inv_norm = e->sp_inv_norm.as<float>(); io.txt = e->sp_txt.as<float>();
    io.ctx_row_tab = L.ctx_general ? L.ctx_row.as<int32_t>() : nullptr;
    return io;
}
static int sparse_forward(rlcf_engine* e, const float* ctx, const int32_t* cls, int n_e, hipStream_t st) {
    ClipModel& m = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    TRY(launch_build_sparse_layout(cls, 1, n_e, L.class_start.as<int32_t>(), L.class_len.as<int32_t>(), L.class_eot_off.as<int32_t>(),
                                   L.lmax, L.pre_rows, e->sp_seqs.as<rlcf_seq>(), e->sp_eot_rows.as<int32_t>(),
                                   e->sp_row_src.as<int32_t>(), st));
    return text_forward(e, m, L, e->st, ctx, sparse_io(e, n_e), true, st);
}
static int sparse_backward_only(rlcf_engine* e, const float* sel_feat, const int32_t* cls, int n_e, int K, const float* dlogits, float* dctx,
                                hipStream_t st) {
    ClipModel& m = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    const int D = m.cfg.embed_dim;
    TRY(launch_dtxt_sparse(dlogits, cls, sel_feat, n_e, K, L.C, D, m.logit_scale_exp, e->sp_dtxt.as<float>(), st));
    return text_backward(e, m, e->st, sparse_io(e, n_e), L.max_keys, e->sp_dtxt.as<float>(), e->sp_du.as<float>(), e->sp_dxe.as<float>(),
                         e->sp_ctx_rows_list.as<int32_t>(), L.pre_rows > 0 ? 1 : n_e, L.n_ctx, dctx, st);
}
static int sparse_backward(rlcf_engine* e, const float* ctx, const float* sel_feat, const int32_t* cls, int n_e, int K,
                           const float* dlogits, float* dctx, hipStream_t st) {
    TRY(sparse_forward(e, ctx, cls, n_e, st));
    return sparse_backward_only(e, sel_feat, cls, n_e, K, dlogits, dctx, st);
}

// ------------------------------------------------------------------ one test sample
// set_image_features of every reward model on the selected views (clip_reward.py:59-61,130-137,259-270); the optional
// output is the per-model blocks [rows, Dr_m] one after another.
static int reward_encode(rlcf_engine* e, int rows, int in_res, float* out_concat, hipStream_t st) {
    for (int m = 0; m < e->n_rewards; ++m) {
        const int Dr = e->model[RLCF_REWARD + m].cfg.embed_dim;
        TRY(engine_encode_image(e, RLCF_REWARD + m, e->views_sel.as<float>(), rows, e->rimg[m].as<float>(), st, in_res));
        if (out_concat) {
            RLCF_HIP_CHECK(hipMemcpyAsync(out_concat, e->rimg[m].p, (size_t)rows * Dr * sizeof(float), hipMemcpyDeviceToDevice, st));
            out_concat += (size_t)rows * Dr;
        }
    }
    return RLCF_OK;
}
static RewardBank reward_bank(const rlcf_engine* e) {
    RewardBank b{};
    b.n = e->n_rewards;
    for (int m = 0; m < e->n_rewards; ++m) {
        b.class_feat[m] = e->reward_cls[m].as<float>(); b.reward_img[m] = e->rimg[m].as<float>();
        b.Dr[m] = e->model[RLCF_REWARD + m].cfg.embed_dim;
        b.mix[m] = e->reward_mean ? 1.f : e->reward_mix[m];
    }
    b.post_div = e->reward_mean ? (float)e->n_rewards : 1.f;
    return b;
}

// Harness body TPT/tpt_cls_rl.py:251-262 around test_time_tuning (:47-79).
#define COPY_OUT(dst, src, bytes) do { if (dst) RLCF_HIP_CHECK(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToDevice, st)); } while (0)
int engine_tta_sample(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || e->image_bank || e->n_rewards <= 0) { rlcf_set_error("class bank / reward model not set"); return RLCF_ERR_STATE; }
    if (e->n_ctx <= 0) { rlcf_set_error("prompt tuning needs a class bank with learnable context rows (n_ctx > 0)"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(N > 0 && N <= e->max_views && a && a->tta_steps >= 0 && a->sample_k > 0 && a->sample_k <= 32);
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim, Wt = s.cfg.text_width, n_ctx = e->n_ctx;
    const int n_sel = n_selected(a, N);                       // int() truncation, tpt_cls_rl.py:34
    RLCF_ARG_CHECK(K <= C);
    if (a->tta_steps > 0 && n_sel <= 0) { rlcf_set_error("int(N*selection_p) == 0 views selected (N=%d, p=%g)", N, a->selection_p); return RLCF_ERR_ARG; }
    const size_t cb = (size_t)n_ctx * Wt * sizeof(float);
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const rlcf_tta_out none{};
    if (!out) out = &none;
    const bool sparse_ok = a->sparse_backward && (a->flags & RLCF_F_REWARD_PROCESS) && !(a->flags & RLCF_F_PROCESS_BATCH) &&
                           !(a->flags & RLCF_F_MIN_ENTROPY) && K > 1;
    const int n_e = n_sel * K;
    if (a->tta_steps > 0) {
        if (sparse_ok) TRY(sparse_ensure(e, n_e, st));
        else {
            TRY(tower_ensure_saved(e->tt, e->lay[0].T, Wt, s.cfg.text_layers, st));
            TRY(bwd_ensure(e, e->lay[0].T, Wt));
            TRY(e->sp_du.ensure((size_t)C * D * sizeof(float))); TRY(e->sp_dxe.ensure((size_t)C * Wt * sizeof(float)));
        }
    }
    e->last_flops = 0.0;
    float* ctx = e->ctx.as<float>();
    // model.reset() + optimizer.load_state_dict(optim_state): custom_clip.py:161-164, tpt_cls_rl.py:251-255
    RLCF_HIP_CHECK(hipMemcpyAsync(ctx, a->ctx_in ? (const void*)a->ctx_in : e->ctx_init.p, cb, hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->adam_m.p, 0, cb, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->adam_v.p, 0, cb, st));
    // student image features of all N views: computed once (the image tower is frozen, custom_clip.py:325-327)
    TRY(engine_encode_image(e, RLCF_STUDENT, views, N, e->img_feat.as<float>(), st));
    TextPassIO io = full_io(e, e->lay[0]);
    static int no_overlap = -1;                              // RLCF_NO_OVERLAP=1: everything on the caller's stream (benchmarks)
    if (no_overlap < 0) { const char* ev = getenv("RLCF_NO_OVERLAP"); no_overlap = ev ? atoi(ev) : 0; }
    bool vit_rewards = true;
    for (int m = 0; m < e->n_rewards; ++m) vit_rewards = vit_rewards && !is_resnet(e->model[RLCF_REWARD + m].cfg);
    const bool overlap = sparse_ok && vit_rewards && !no_overlap && !e->no_side && !g_prof.enabled && e->side && prec_x3(e) && !prec_single(e);
    bool fwd_done = false;
    for (int j = 0; j < a->tta_steps; ++j) {
        // step 0 runs on ctx == ctx_init: its text features are the cached txt0 (the dense-backward
        // path still needs this pass for its saved activations)
        const bool cached = (j == 0 && sparse_ok && !a->ctx_in);
        if (!cached) TRY(text_forward(e, s, e->lay[0], e->tt, ctx, io, !sparse_ok, st));
        const float* txt_j = cached ? e->txt0.as<float>() : e->txt.as<float>();
        const float* rows_logits;
        if (j == 0) {   // tpt_cls_rl.py:57-59
            TRY(engine_logits(e, e->img_feat.as<float>(), N, txt_j, C, e->logits.as<float>(), st));
            TRY(launch_entropy_select(e->logits.as<float>(), N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
            TRY(launch_gather_rows(e->img_feat.as<float>(), D, e->sel_idx.as<int32_t>(), e->sel_feat.as<float>(), D, n_sel, D, st));
            TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, n_sel, (int)img_elems, st));
            // the reward models' pass over the selected views depends on nothing the student does from here to the loss, and at one
            // image's sizes neither it nor the sparse text forward fills the chip: second stream, joined before the loss kernel
            if (overlap) {
                // the side stream's own A-operand buffer: the patch matrix of the selected views or a <= 512-row token matrix against a
                // W x 4W weight, whichever reward model needs more (the main stream keeps a_hi for the text passes it runs meanwhile)
                size_t need2 = 0;
                for (int m = 0; m < e->n_rewards; ++m) {
                    const ClipModel& rm = e->model[RLCF_REWARD + m];
                    need2 = std::max(need2, (size_t)n_sel * rm.tokens * std::max(rm.Kp, 4 * rm.cfg.vision_width));
                }
                if (need2 > e->a_split2_elems) { TRY(e->a_hi2.ensure(need2 * 4)); e->a_split2_elems = need2; }
                RLCF_HIP_CHECK(hipEventRecord(e->ev_fork, st));
                RLCF_HIP_CHECK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
                e->ws_sel = 1;
                const int rc_side = reward_encode(e, n_sel, s.cfg.image_resolution, out->reward_image_features, e->side);
                e->ws_sel = 0;
                const hipError_t er = hipEventRecord(e->ev_join, e->side);      // (recorded even after an error: the main stream must not run ahead)
                if (rc_side != RLCF_OK || er != hipSuccess) {
                    (void)hipStreamWaitEvent(st, e->ev_join, 0);
                    if (er != hipSuccess) { (void)hipStreamSynchronize(e->side); rlcf_set_error("hipEventRecord(ev_join): %s", hipGetErrorString(er)); return RLCF_ERR_HIP; }
                    return rc_side;
                }
            } else {
                TRY(reward_encode(e, n_sel, s.cfg.image_resolution, out->reward_image_features, st));
            }
            TRY(launch_gather_rows(e->logits.as<float>(), C, e->sel_idx.as<int32_t>(), e->sel_logits.as<float>(), C, n_sel, C, st));
            rows_logits = e->sel_logits.as<float>();
            if (overlap) {
                // whatever happens on the main stream, it joins the side stream before this call returns: the side stream writes
                // e->vt, e->rimg and the caller's reward_image_features
                int rc_main = launch_topk_rows(rows_logits, C, n_sel, C, K, e->topk_idx.as<int32_t>(), e->rl_stats.as<float>(), st);
                if (rc_main == RLCF_OK) rc_main = sparse_forward(e, ctx, e->topk_idx.as<int32_t>(), n_e, st);
                fwd_done = true;
                const hipError_t ej = hipStreamWaitEvent(st, e->ev_join, 0);
                if (rc_main != RLCF_OK) { if (ej != hipSuccess) (void)hipStreamSynchronize(e->side); return rc_main; }
                if (ej != hipSuccess) { (void)hipStreamSynchronize(e->side); rlcf_set_error("hipStreamWaitEvent(ev_join): %s", hipGetErrorString(ej)); return RLCF_ERR_HIP; }
            }
            COPY_OUT(out->logits, e->logits.p, (size_t)N * C * sizeof(float));
            COPY_OUT(out->entropy, e->entropy.p, N * sizeof(float));
            COPY_OUT(out->selected_idx, e->sel_idx.p, n_sel * sizeof(int32_t));
        } else {        // tpt_cls_rl.py:55 — selected views only; their image features are unchanged
            TRY(engine_logits(e, e->sel_feat.as<float>(), n_sel, txt_j, C, e->sel_logits.as<float>(), st));
            rows_logits = e->sel_logits.as<float>();
        }
        TRY(launch_reward_loss_bank(rows_logits, C, nullptr, 1, n_sel, C, K, reward_bank(e),
                               a->clipscore_weight, a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), e->clip_score.as<float>(),
                               e->rewards.as<float>(), e->loss.as<float>(), e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        if (sparse_ok && fwd_done) {
            TRY(sparse_backward_only(e, e->sel_feat.as<float>(), e->topk_idx.as<int32_t>(), n_e, K, e->dlogits.as<float>(), e->ctx_grad.as<float>(), st));
            fwd_done = false;
        } else if (sparse_ok) {
            TRY(sparse_backward(e, ctx, e->sel_feat.as<float>(), e->topk_idx.as<int32_t>(), n_e, K, e->dlogits.as<float>(),
                                e->ctx_grad.as<float>(), st));
        } else {
            TRY(launch_dtxt_dense(e->dlogits.as<float>(), e->sel_feat.as<float>(), n_sel, C, D, s.logit_scale_exp, e->dtxt_dense.as<float>(), st));
            TRY(text_backward(e, s, e->tt, io, e->lay[0].max_keys, e->dtxt_dense.as<float>(), e->sp_du.as<float>(), e->sp_dxe.as<float>(),
                              e->lay[0].ctx_rows_list.as<int32_t>(), e->lay[0].n_copies, n_ctx, e->ctx_grad.as<float>(), st));
        }
        if (j == 0) {
            COPY_OUT(out->topk_idx, e->topk_idx.p, (size_t)n_e * sizeof(int32_t));
            COPY_OUT(out->clip_score, e->clip_score.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->rewards, e->rewards.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->loss, e->loss.p, sizeof(float));
            COPY_OUT(out->dlogits, e->dlogits.p, (size_t)n_sel * C * sizeof(float));
            COPY_OUT(out->ctx_grad, e->ctx_grad.p, cb);
        }
        // scaler.step(optimizer) (tpt_cls_rl.py:78): a gradient with an inf / NaN skips the update (the same inputs give the same
        // gradient at the following steps, so the host-side step number j + 1 never meets an applied step after a skipped one)
        TRY(launch_grad_nonfinite(e->ctx_grad.as<float>(), (int64_t)n_ctx * Wt, 1, e->step_skip.as<int32_t>(), st));
        TRY(launch_adamw(ctx, e->ctx_grad.as<float>(), e->adam_m.as<float>(), e->adam_v.as<float>(), (int64_t)n_ctx * Wt, j + 1, a->lr,
                         a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)n_ctx * Wt));
        if (out->step_skipped) COPY_OUT(out->step_skipped + j, e->step_skip.p, sizeof(int32_t));
    }
    // final inference on the clean view (views[0]) with the adapted prompt, tpt_cls_rl.py:260-262;
    // its image feature is row 0 of img_feat (frozen image tower: identical to re-encoding it).
    COPY_OUT(out->ctx_after, ctx, cb);
    if (a->skip_final) return RLCF_OK;
    TRY(text_forward(e, s, e->lay[0], e->tt, ctx, io, false, st));
    TRY(engine_logits(e, e->img_feat.as<float>(), 1, e->txt.as<float>(), C, e->final_logits.as<float>(), st));
    TRY(launch_top5(e->final_logits.as<float>(), C, e->top5.as<int32_t>(), st));
    COPY_OUT(out->final_logits, e->final_logits.p, (size_t)C * sizeof(float));
    COPY_OUT(out->top5, e->top5.p, 5 * sizeof(int32_t));
    return RLCF_OK;
}

// ------------------------------------------------------------------ B test samples per pass
// Same arithmetic per sample as engine_tta_sample (default RLCF configuration: one tuning step, sparse class backward),
// but every tower pass runs once for the whole batch: B*N views through the student image tower, B*n_sel views through
// the reward tower, B*n_sel*K class prompts through the sparse forward/backward (each sample with its own copy of the
// prompt prefix), B adapted prompts through one replicated final text pass.  Samples stay independent (no cross-sample
// arithmetic); larger M per GEMM is what fills 256 CUs.
static int batch_ensure(rlcf_engine* e, int B, hipStream_t st) {
    if (B <= e->b_cap) return RLCF_OK;
    ClipModel& s = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    const int Wt = s.cfg.text_width, D = s.cfg.embed_dim, C = L.C;
    TRY(e->b_seqs_rep.ensure((size_t)B * L.n_seq * sizeof(rlcf_seq))); TRY(e->b_eot_rep.ensure((size_t)B * C * sizeof(int32_t)));
    TRY(launch_replicate_layout(L.seqs.as<rlcf_seq>(), L.n_seq, L.eot_rows.as<int32_t>(), C, L.T, B, e->b_seqs_rep.as<rlcf_seq>(),
                                e->b_eot_rep.as<int32_t>(), st));
    if (L.n_pk > 0) {        // packed runs: the descriptors shift like the sequences, the per-row sequence starts like row ids
        TRY(e->b_pk_rep.ensure((size_t)B * L.n_pk * sizeof(rlcf_seq))); TRY(e->b_rss_rep.ensure((size_t)B * L.T * sizeof(int32_t)));
        TRY(launch_replicate_layout(L.pk_seqs.as<rlcf_seq>(), L.n_pk, L.pk_rss.as<int32_t>(), L.T, L.T, B, e->b_pk_rep.as<rlcf_seq>(),
                                    e->b_rss_rep.as<int32_t>(), st));
    }
    const size_t cb = (size_t)B * e->n_ctx * Wt * sizeof(float);
    TRY(e->b_ctx.ensure(cb)); TRY(e->b_m.ensure(cb)); TRY(e->b_v.ensure(cb)); TRY(e->b_grad.ensure(cb));
    TRY(e->b_txt.ensure((size_t)B * C * D * sizeof(float))); TRY(e->b_u.ensure((size_t)B * C * D * sizeof(float)));
    TRY(e->b_eot_x.ensure((size_t)B * C * Wt * sizeof(float))); TRY(e->b_eot_ln.ensure((size_t)B * C * Wt * sizeof(float)));
    TRY(e->b_inv.ensure((size_t)B * C * sizeof(float))); TRY(e->b_logits.ensure((size_t)B * C * sizeof(float)));
    TRY(tower_ensure(e->tt, B * L.T, Wt, st));
    if (prec_x3(e) && (size_t)B * L.T * Wt * 4 > e->a_split_elems) {
        e->a_split_elems = (size_t)B * L.T * Wt * 4;
        TRY(e->a_hi.ensure(e->a_split_elems * 4));
    }
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    e->b_cap = B;
    return RLCF_OK;
}

// img_feat: the student image features [B*N, D] of these views — nullptr: computed here (the whole pass on one stream); else they
// were produced by the caller (tta_batch_pipelined: the tower of this part ran on the other stream) and only the rest of the pass runs
static int tta_batch_fused(rlcf_engine* e, const float* views, int B, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                           hipStream_t st, const float* img_feat = nullptr) {
    ClipModel& s = e->model[RLCF_STUDENT];
    const TextLayout& L = e->lay[0];
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim, Wt = s.cfg.text_width, n_ctx = e->n_ctx;
    const int n_sel = n_selected(a, N), n_e = n_sel * K, BN = B * N, BS = B * n_sel;
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    if (!img_feat) {
        TRY(batch_ensure(e, B, st));
        TRY(sparse_ensure(e, n_e, st, B));
        e->last_flops = 0.0;
        // 1. student image features of all B*N views
        TRY(engine_encode_image(e, RLCF_STUDENT, views, BN, e->img_feat.as<float>(), st));
        img_feat = e->img_feat.as<float>();
    }
    // first-step logits against the cached pristine-prompt text features
    TRY(engine_logits(e, img_feat, BN, e->txt0.as<float>(), C, e->logits.as<float>(), st));
    // 2. per-sample confidence selection (global row ids), gathers, reward features of the selected views
    TRY(launch_entropy_select_batched(e->logits.as<float>(), B, N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
    TRY(launch_gather_rows(img_feat, D, e->sel_idx.as<int32_t>(), e->sel_feat.as<float>(), D, BS, D, st));
    TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, BS, (int)img_elems, st));
    TRY(reward_encode(e, BS, s.cfg.image_resolution, nullptr, st));
    TRY(launch_gather_rows(e->logits.as<float>(), C, e->sel_idx.as<int32_t>(), e->sel_logits.as<float>(), C, BS, C, st));
    const int gT = L.pre_rows + n_e * L.lmax, T = B * gT, nE = B * n_e;
    const int64_t np = (int64_t)n_ctx * Wt;
    // reset state of every sample: ctx = ctx_init, Adam moments zero (custom_clip.py:161-164, tpt_cls_rl.py:251-255)
    TRY(launch_broadcast_rows(e->ctx_init.as<float>(), e->b_ctx.as<float>(), (int)np, B, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->b_m.p, 0, (size_t)B * np * sizeof(float), st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->b_v.p, 0, (size_t)B * np * sizeof(float), st));
    TextPassIO fo{};                 // full class bank, one replica (and one prompt) per sample
    fo.seqs = e->b_seqs_rep.as<rlcf_seq>(); fo.n_seq = B * L.n_seq; fo.max_q_len = L.max_q_len; fo.T = B * L.T; fo.n_cls = B * C;
    fo.attn_pairs = (long)B * L.attn_pairs; fo.eot_rows = e->b_eot_rep.as<int32_t>(); fo.row_src = nullptr;
    fo.eot_x = e->b_eot_x.as<float>(); fo.eot_ln = e->b_eot_ln.as<float>(); fo.u = e->b_u.as<float>(); fo.inv_norm = e->b_inv.as<float>();
    fo.txt = e->b_txt.as<float>(); fo.rep_rows = L.T; fo.ctx_stride = (int)np;
    if (L.n_pk > 0) { fo.pk_seqs = e->b_pk_rep.as<rlcf_seq>(); fo.n_pk = B * L.n_pk; fo.pk_rss = e->b_rss_rep.as<int32_t>(); }
    for (int j = 0; j < a->tta_steps; ++j) {
        if (j > 0) {
            // tpt_cls_rl.py:55: logits of the selected views under each sample's current prompt
            TRY(text_forward(e, s, L, e->tt, e->b_ctx.as<float>(), fo, false, st));
            TRY(launch_group_logits(e->sel_feat.as<float>(), n_sel, e->b_txt.as<float>(), B, C, D, s.logit_scale_exp, e->sel_logits.as<float>(), st));
            e->last_flops += 2.0 * BS * C * D;
        }
        // 3. top-K sampling, CLIP reward, baseline, reward-weighted CE and dlogits, grouped per sample
        TRY(launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, B, n_sel, C, K, reward_bank(e),
                                    a->clipscore_weight, a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), nullptr, nullptr, nullptr,
                                    e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        // 4. sparse backward of all B*n_e sampled (view, class) pairs; each sample owns a copy of the prompt prefix
        TRY(launch_build_sparse_layout(e->topk_idx.as<int32_t>(), B, n_e, L.class_start.as<int32_t>(), L.class_len.as<int32_t>(),
                                       L.class_eot_off.as<int32_t>(), L.lmax, L.pre_rows, e->sp_seqs.as<rlcf_seq>(), e->sp_eot_rows.as<int32_t>(),
                                       e->sp_row_src.as<int32_t>(), st));
        TextPassIO io{};
        io.seqs = e->sp_seqs.as<rlcf_seq>(); io.n_seq = B * (n_e + (L.pre_rows > 0 ? 1 : 0)); io.max_q_len = L.max_q_len; io.T = T; io.n_cls = nE;
        io.attn_pairs = (long)(nE * (L.mean_len * (L.pre_rows + (L.mean_len + 1) * 0.5)));
        io.eot_rows = e->sp_eot_rows.as<int32_t>(); io.row_src = e->sp_row_src.as<int32_t>();
        io.eot_x = e->sp_eot_x.as<float>(); io.eot_ln = e->sp_eot_ln.as<float>(); io.u = e->sp_u.as<float>();
        io.inv_norm = e->sp_inv_norm.as<float>(); io.txt = e->sp_txt.as<float>();
        io.rep_rows = gT; io.ctx_stride = (int)np;                  // group b of gT rows reads prompt b
        TRY(text_forward(e, s, L, e->st, e->b_ctx.as<float>(), io, true, st));
        TRY(launch_dtxt_sparse(e->dlogits.as<float>(), e->topk_idx.as<int32_t>(), e->sel_feat.as<float>(), nE, K, C, D, s.logit_scale_exp,
                               e->sp_dtxt.as<float>(), st));
        {   // text_backward with the per-sample (grouped) ctx-gradient reduction
            TRY(launch_l2norm_bwd(io.txt, e->sp_dtxt.as<float>(), io.inv_norm, e->sp_du.as<float>(), nE, D, st));
            TRY(gemm(e, e->sp_du.as<float>(), D, s.tproj, D, nullptr, nullptr, 0, nullptr, 0, e->sp_dxe.as<float>(), Wt, nE, Wt, D, 1.f, RLCF_EPI_NONE, st));
            TRY(launch_layernorm_bwd(io.eot_x, s.lnf_w, e->sp_dxe.as<float>(), nullptr, e->sp_dxe.as<float>(), nullptr, nullptr, nE, Wt, st));
            RLCF_HIP_CHECK(hipMemsetAsync(e->dX.p, 0, (size_t)T * Wt * sizeof(float), st));
            TRY(launch_scatter_rows(e->sp_dxe.as<float>(), io.eot_rows, e->dX.as<float>(), nE, Wt, st));
            TRY(transformer_backward(e, s.txt, e->st, io.seqs, io.n_seq, L.max_keys, io.attn_pairs, 1, T, st));
            if (L.ctx_general)
                TRY(launch_ctx_grad_scan(e->dX.as<float>(), io.row_src, L.ctx_row.as<int32_t>(), B, gT, n_ctx, Wt, e->b_grad.as<float>(), st));
            else
                TRY(launch_ctx_grad_grouped(e->dX.as<float>(), e->sp_ctx_rows_list.as<int32_t>(), L.pre_rows > 0 ? 1 : n_e, n_ctx, Wt, B, gT,
                                            e->b_grad.as<float>(), st));
        }
        // 5. AdamW step j+1 of every sample (tpt_cls_rl.py:76-79)
        TRY(launch_grad_nonfinite(e->b_grad.as<float>(), np, B, e->step_skip.as<int32_t>(), st));
        TRY(launch_adamw(e->b_ctx.as<float>(), e->b_grad.as<float>(), e->b_m.as<float>(), e->b_v.as<float>(), B * np, j + 1, a->lr, a->beta1,
                         a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), np));
    }
    // 6. final clean-view inference: B adapted prompts through one replicated text pass
    TRY(text_forward(e, s, L, e->tt, e->b_ctx.as<float>(), fo, false, st));
    float* fl = final_logits ? final_logits : e->b_logits.as<float>();
    TRY(launch_final_logits_batched(img_feat, N, e->b_txt.as<float>(), B, C, D, s.logit_scale_exp, fl, st));
    e->last_flops += 2.0 * B * C * D;
    TRY(launch_top5_batched(fl, B, C, top5, st));
    return RLCF_OK;
}

// One pass of B test images in `parts` parts on TWO streams: the student image tower of part k+1 (chip-filling GEMMs) runs on the
// caller's stream while everything behind the tower of part k — reward models' pass over the selected views, loss, sparse text
// forward / backward, AdamW, the replicated final text pass: small launch-bound kernels, ~20 % of a pass — runs on the side stream
// with the side stream's own scratch (ws_sel: A-operand buffer, split-K workspace, image-tower scratch).  Samples are independent, so
// the per-sample results are those of the one-stream pass up to the round-off of the GEMM forms the part sizes select
// (test_batch_pipeline_equals_single_stream).  MEASURED SLOWER than the one-stream pass on BASELINE configs[1] (115.7 images/s
// against 114.1 / 109.3 in 2 / 4 parts: a workgroup of the small kernels blocks a CU for the 139-KB GEMM workgroups of the tower just as
// it does alone, so the two-stream form hides nothing) — built only when RLCF_BATCH_PARTS=n asks for it.
static int tta_batch_pipelined(rlcf_engine* e, const float* views, int B, int N, int parts, const rlcf_tta_args* a, float* final_logits,
                               int32_t* top5, hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    const int D = s.cfg.embed_dim, n_sel = n_selected(a, N), n_e = n_sel * a->sample_k;
    const size_t per = (size_t)N * 3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const int Bp = (B + parts - 1) / parts;
    TRY(batch_ensure(e, Bp, st));
    TRY(sparse_ensure(e, n_e, st, Bp));
    {   // side-stream operand buffer: the text passes of a part and the reward models' patch matrices
        size_t need2 = (size_t)Bp * e->lay[0].T * s.cfg.text_width * 4;
        for (int m = 0; m < e->n_rewards; ++m) {
            const ClipModel& rm = e->model[RLCF_REWARD + m];
            need2 = std::max(need2, (size_t)Bp * n_sel * rm.tokens * std::max(rm.Kp, 4 * rm.cfg.vision_width));
        }
        if (need2 > e->a_split2_elems) { TRY(e->a_hi2.ensure(need2 * 4)); e->a_split2_elems = need2; }
    }
    for (int k = 0; k < parts && k < 8; ++k)
        if (!e->ev_part[k]) RLCF_HIP_CHECK(hipEventCreateWithFlags(&e->ev_part[k], hipEventDisableTiming));
    e->last_flops = 0.0;
    RLCF_HIP_CHECK(hipEventRecord(e->ev_fork, st));
    RLCF_HIP_CHECK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
    int rc = RLCF_OK;
    for (int k = 0, b0 = 0; k < parts && b0 < B && rc == RLCF_OK; ++k, b0 += Bp) {
        const int Bk = std::min(Bp, B - b0);
        float* feat_k = e->img_feat.as<float>() + (size_t)b0 * N * D;
        rc = engine_encode_image(e, RLCF_STUDENT, views + (size_t)b0 * per, Bk * N, feat_k, st);
        if (rc != RLCF_OK) break;
        if (hipEventRecord(e->ev_part[k], st) != hipSuccess || hipStreamWaitEvent(e->side, e->ev_part[k], 0) != hipSuccess) { rc = RLCF_ERR_HIP; break; }
        e->ws_sel = 1;
        rc = tta_batch_fused(e, views + (size_t)b0 * per, Bk, N, a, final_logits ? final_logits + (size_t)b0 * e->C : nullptr, top5 + (size_t)b0 * 5,
                             e->side, feat_k);
        e->ws_sel = 0;
    }
    // the caller's stream joins the side stream whatever happened
    const hipError_t e1 = hipEventRecord(e->ev_join, e->side);
    const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, e->ev_join, 0) : e1;
    if (e2 != hipSuccess) { (void)hipStreamSynchronize(e->side); if (rc == RLCF_OK) { rlcf_set_error("tta_batch_pipelined: join: %s", hipGetErrorString(e2)); rc = RLCF_ERR_HIP; } }
    return rc;
}

int engine_tta_batch(rlcf_engine* e, const float* views, int count, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                     hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || e->image_bank || e->n_rewards <= 0) { rlcf_set_error("class bank / reward model not set"); return RLCF_ERR_STATE; }
    if (e->n_ctx <= 0) { rlcf_set_error("prompt tuning needs a class bank with learnable context rows (n_ctx > 0)"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(N > 0 && N <= e->max_views && a->sample_k > 0 && a->sample_k <= 32 && a->sample_k <= e->C);
    const size_t per = (size_t)N * 3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const int n_sel = n_selected(a, N);
    const bool sparse_ok = a->sparse_backward && (a->flags & RLCF_F_REWARD_PROCESS) && !(a->flags & RLCF_F_PROCESS_BATCH) &&
                           !(a->flags & RLCF_F_MIN_ENTROPY) && a->sample_k > 1;
    const int Bmax = e->max_views / N;
    const bool fused = Bmax >= 2 && a->tta_steps >= 1 && sparse_ok && !a->ctx_in && !a->skip_final && n_sel > 0;
    double flops = 0.0;
    int i = 0;
    while (i < count) {
        const int B = fused ? std::min(Bmax, count - i) : 1;
        if (fused && B >= 2) {
            // parts of a pass on two streams (tta_batch_pipelined): needs the side stream, ViT towers everywhere (the ModifiedResNet
            // pass keeps its scratch in the engine), no per-launch profile (its event pairs serialise), one tuning step
            static int parts_env = -1;
            if (parts_env < 0) { const char* ev = getenv("RLCF_BATCH_PARTS"); parts_env = ev ? atoi(ev) : 0; }
            bool vit_all = !is_resnet(s.cfg);
            for (int m = 0; m < e->n_rewards; ++m) vit_all = vit_all && !is_resnet(e->model[RLCF_REWARD + m].cfg);
            int parts = parts_env > 0 ? parts_env : 1;          // (measured slower than one stream on BASELINE configs[1]: off unless asked for)
            parts = std::min(std::min(parts, 8), B / 2);
            if (parts >= 2 && vit_all && e->side && !g_prof.enabled && prec_x3(e)) {
                TRY(tta_batch_pipelined(e, views + (size_t)i * per, B, N, parts, a, final_logits ? final_logits + (size_t)i * e->C : nullptr,
                                        top5 + (size_t)i * 5, st));
            } else
            TRY(tta_batch_fused(e, views + (size_t)i * per, B, N, a, final_logits ? final_logits + (size_t)i * e->C : nullptr, top5 + (size_t)i * 5, st));
        } else {
            rlcf_tta_out o{};
            o.top5 = top5 + (size_t)i * 5;
            o.final_logits = final_logits ? final_logits + (size_t)i * e->C : nullptr;
            TRY(engine_tta_sample(e, views + (size_t)i * per, N, a, &o, st));
        }
        flops += e->last_flops;
        i += B;
    }
    e->last_flops = flops / count;
    return RLCF_OK;
}

// ------------------------------------------------------------------ LayerNorm-tuning step (BASELINE configs[2])
// Image tower forward of n views WITH saved activations (CLIPCLS_TTA.forward, custom_clip.py:423-432).
static int vit_forward_saved(rlcf_engine* e, ClipModel& m, const float* images, int n, float* feats, hipStream_t st) {
    const rlcf_clip_cfg& c = m.cfg;
    const int Wv = c.vision_width, tok = m.tokens, G2 = tok - 1, T = n * tok, D = c.embed_dim;
    TRY(tower_ensure_saved(e->vt, T, Wv, c.vision_layers, st));
    TRY(launch_im2col(images, e->patches.as<float>(), nullptr, nullptr, n, c.image_resolution, c.vision_patch_size, m.Kp, st));
    TRY(gemm(e, e->patches.as<float>(), m.Kp, m.conv_w, m.Kp, nullptr, nullptr, 0, nullptr, 0, e->patch_out.as<float>(), Wv, n * G2, Wv, m.Kp,
             1.f, RLCF_EPI_NONE, st));
    {
        const LnRef gw = ln_ref(e, m.lnpre_w, 1), gb = ln_ref(e, m.lnpre_b, 1);
        TRY(launch_vit_assemble(e->patch_out.as<float>(), m.cls, m.vpos, gw.p, gb.p, e->vt.sv[0].x, n, tok, Wv, st, gw.group_rows, gw.group_stride));
    }
    TRY(transformer_forward(e, m.vis, e->vt, e->vit_seqs.as<rlcf_seq>(), n, tok, (long)n * tok * tok, 0, T, true, st));
    TRY(launch_gather_rows(e->vt.x.as<float>(), tok * Wv, nullptr, e->cls_rows.as<float>(), Wv, n, Wv, st));
    {
        const LnRef gw = ln_ref(e, m.lnpost_w, 1), gb = ln_ref(e, m.lnpost_b, 1);
        TRY(launch_layernorm_fwd(e->cls_rows.as<float>(), gw.p, gb.p, e->cls_ln.as<float>(), n, Wv, st, gw.group_rows, gw.group_stride));
    }
    TRY(gemm(e, e->cls_ln.as<float>(), Wv, m.vprojT, Wv, nullptr, nullptr, 0, nullptr, 0, e->feat_raw.as<float>(), D, n, D, Wv, 1.f,
             RLCF_EPI_NONE, st));
    TRY(launch_l2norm_rows(e->feat_raw.as<float>(), feats, e->vit_inv_norm.as<float>(), n, D, st));
    return RLCF_OK;
}
// d loss / d (visual LN parameters) given dlogits [n, C] of the n views whose activations vit_forward_saved holds.
// groups > 1: the n views belong to `groups` test samples (n / groups consecutive views each) and ln_grad is [groups, ln_count]
// wgrad_base (single sample only): also the gradient of every other visual parameter, into the flat e->vw_slots layout
static int vit_backward_ln(rlcf_engine* e, ClipModel& m, const float* feats, int n, const float* dlogits, float* ln_grad, hipStream_t st,
                           int groups = 1, float* wgrad_base = nullptr) {
    const rlcf_clip_cfg& c = m.cfg;
    const int Wv = c.vision_width, tok = m.tokens, T = n * tok, D = c.embed_dim, C = e->C, L = c.vision_layers;
    const int per = n / groups, gs = groups > 1 ? e->ln_count : 0;
    TRY(bwd_ensure(e, T, Wv));
    TRY(e->dfeat.ensure((size_t)e->max_views * D * sizeof(float))); TRY(e->dcls.ensure((size_t)e->max_views * Wv * sizeof(float)));
    RLCF_HIP_CHECK(hipMemsetAsync(ln_grad, 0, (size_t)groups * e->ln_count * sizeof(float), st));
    // d feat = scale * dlogits @ class_features  (logits = scale * feat @ class_features^T, custom_clip.py:429-430)
    TRY(launch_dimg(dlogits, e->txt0.as<float>(), n, C, D, m.logit_scale_exp, e->dfeat.as<float>(), st));
    TRY(launch_l2norm_bwd(feats, e->dfeat.as<float>(), e->vit_inv_norm.as<float>(), e->dfeat.as<float>(), n, D, st));
    if (wgrad_base)                        // visual.proj [Wv, D]: feat = ln_post(cls) @ proj  (model.py:237-238)
        TRY(wgrad(e, e->cls_ln.as<float>(), Wv, Wv, e->dfeat.as<float>(), D, D, n, wgrad_base + e->vw_slots[2].off, nullptr, st));
    TRY(gemm(e, e->dfeat.as<float>(), D, m.vproj, D, nullptr, nullptr, 0, nullptr, 0, e->dcls.as<float>(), Wv, n, Wv, D, 1.f, RLCF_EPI_NONE, st));
    float* gpost = ln_grad + (size_t)(2 + 4 * L) * Wv;
    const LnRef gpw = ln_ref(e, m.lnpost_w, 1);
    TRY(e->parts_ws.ensure(RLCF_PARTS_WS_FLOATS * sizeof(float)));
    TRY(launch_layernorm_bwd(e->cls_rows.as<float>(), gpw.p, e->dcls.as<float>(), nullptr, e->dcls.as<float>(), gpost, gpost + Wv, n, Wv, st,
                             groups > 1 ? per : 0, gs, groups > 1 ? gpw.group_stride : 0, PARTS_WS(e)));
    RLCF_HIP_CHECK(hipMemsetAsync(e->dX.p, 0, (size_t)T * Wv * sizeof(float), st));
    TRY(launch_scatter_rows(e->dcls.as<float>(), e->cls_row_idx.as<int32_t>(), e->dX.as<float>(), n, Wv, st));
    TRY(transformer_backward(e, m.vis, e->vt, e->vit_seqs.as<rlcf_seq>(), n, tok, (long)n * tok * tok, 0, T, st, ln_grad, tok,
                             groups > 1 ? per * tok : 0, gs, wgrad_base));
    if (!wgrad_base) {
        TRY(launch_vit_assemble_bwd(e->patch_out.as<float>(), m.cls, m.vpos, e->dX.as<float>(), ln_grad, ln_grad + Wv, n, tok, Wv, st,
                                    groups > 1 ? per : 0, gs, PARTS_WS(e)));
        return RLCF_OK;
    }
    // through ln_pre into the embedding (model.py:224-229): pre = [class_embedding | conv1(patches)] + positional_embedding
    float *pre = e->dH.as<float>(), *dpre = e->dA.as<float>(), *dpatch = e->dF.as<float>();      // backward scratch, free by now
    const int G2 = tok - 1, K = 3 * m.cfg.vision_patch_size * m.cfg.vision_patch_size;
    TRY(launch_vit_preln(e->patch_out.as<float>(), m.cls, m.vpos, pre, n, tok, Wv, st));
    TRY(launch_layernorm_bwd(pre, m.lnpre_w, e->dX.as<float>(), nullptr, dpre, ln_grad, ln_grad + Wv, T, Wv, st, 0, 0, 0, PARTS_WS(e)));
    float* gpos = wgrad_base + e->vw_slots[1].off;
    TRY(launch_colsum(dpre, tok * Wv, n, tok * Wv, gpos, st, PARTS_WS(e)));                                     // d positional_embedding = sum over views
    RLCF_HIP_CHECK(hipMemcpyAsync(wgrad_base + e->vw_slots[0].off, gpos, Wv * sizeof(float), hipMemcpyDeviceToDevice, st));   // d class_embedding = its row 0
    for (int v = 0; v < n; ++v)
        RLCF_HIP_CHECK(hipMemcpyAsync(dpatch + (size_t)v * G2 * Wv, dpre + ((size_t)v * tok + 1) * Wv, (size_t)G2 * Wv * sizeof(float),
                                      hipMemcpyDeviceToDevice, st));
    // conv1.weight [Wv, 3*ps*ps] (stride == kernel convolution = patches @ W^T, model.py:224): the patch matrix of vit_forward_saved is still there
    TRY(wgrad(e, dpatch, Wv, Wv, e->patches.as<float>(), m.Kp, K, n * G2, wgrad_base + e->vw_slots[3].off, nullptr, st));
    return RLCF_OK;
}

// LayerNorm tuning of B test images per tower pass (one AdamW step): every sample starts from the same reset state, so the
// selection forward, the reward pass and the saved forward of the selected views run on all B samples at once; the backward
// keeps the LayerNorm gradients per sample (grouped reductions), AdamW updates B parameter sets, and the clean-view inference
// runs once per sample with its own adapted LayerNorms (tune_cls_rl.py:206-227).
static int tta_batch_ln_fused(rlcf_engine* e, const float* views, int B, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                              hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim;
    const int n_sel = n_selected(a, N), BN = B * N, BS = B * n_sel;
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const size_t nb = (size_t)e->ln_count * sizeof(float), np = (size_t)e->ln_count;
    TRY(e->ln_feat.ensure((size_t)e->max_views * D * sizeof(float)));
    TRY(e->b_ln.ensure(B * nb)); TRY(e->b_ln_m.ensure(B * nb)); TRY(e->b_ln_v.ensure(B * nb)); TRY(e->b_ln_grad.ensure(B * nb));
    TRY(e->b_logits.ensure((size_t)B * C * sizeof(float)));
    e->last_flops = 0.0;
    const float* cls_feat = e->txt0.as<float>();
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    // 1. selection on all B*N views (pristine LayerNorms), 2. reward features of the selected views
    TRY(engine_encode_image(e, RLCF_STUDENT, views, BN, e->img_feat.as<float>(), st));
    TRY(engine_logits(e, e->img_feat.as<float>(), BN, cls_feat, C, e->logits.as<float>(), st));
    TRY(launch_entropy_select_batched(e->logits.as<float>(), B, N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
    TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, BS, (int)img_elems, st));
    TRY(reward_encode(e, BS, s.cfg.image_resolution, nullptr, st));
    // reset state of every sample (custom_clip.py:456-458 + optimizer.load_state_dict)
    TRY(launch_broadcast_rows(e->ln_init.as<float>(), e->b_ln.as<float>(), (int)np, B, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->b_ln_m.p, 0, B * nb, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->b_ln_v.p, 0, B * nb, st));
    for (int j = 0; j < a->tta_steps; ++j) {
        // 3. forward with saved activations on the selected views (each sample under its own LayerNorms), loss per sample,
        // 4. backward with per-sample LayerNorm gradients, 5. AdamW step j+1 of every sample
        e->lng_base = e->b_ln.as<float>(); e->lng_views = n_sel;
        int rc = vit_forward_saved(e, s, e->views_sel.as<float>(), BS, e->ln_feat.as<float>(), st);
        if (rc == RLCF_OK) rc = engine_logits(e, e->ln_feat.as<float>(), BS, cls_feat, C, e->sel_logits.as<float>(), st);
        if (rc == RLCF_OK) rc = launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, B, n_sel, C, K, reward_bank(e), a->clipscore_weight,
                                                        a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), nullptr, nullptr, nullptr,
                                                        e->dlogits.as<float>(), e->rl_stats.as<float>(), st);
        if (rc == RLCF_OK) rc = vit_backward_ln(e, s, e->ln_feat.as<float>(), BS, e->dlogits.as<float>(), e->b_ln_grad.as<float>(), st, B);
        e->lng_base = nullptr; e->lng_views = 1;
        TRY(rc);
        TRY(launch_grad_nonfinite(e->b_ln_grad.as<float>(), (int64_t)np, B, e->step_skip.as<int32_t>(), st));
        TRY(launch_adamw(e->b_ln.as<float>(), e->b_ln_grad.as<float>(), e->b_ln_m.as<float>(), e->b_ln_v.as<float>(), (int64_t)B * np, j + 1,
                         a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)np));
    }
    // 6. clean-view inference of the B samples in one pass: view b reads LayerNorm set b (tune_cls_rl.py:219-221)
    float* fl = final_logits ? final_logits : e->b_logits.as<float>();
    for (int b = 0; b < B; ++b)
        RLCF_HIP_CHECK(hipMemcpyAsync(e->views_sel.as<float>() + (size_t)b * img_elems, views + (size_t)b * N * img_elems,
                                      img_elems * sizeof(float), hipMemcpyDeviceToDevice, st));
    e->lng_base = e->b_ln.as<float>(); e->lng_views = 1;
    int rc = engine_encode_image(e, RLCF_STUDENT, e->views_sel.as<float>(), B, e->sel_feat.as<float>(), st);
    e->lng_base = nullptr;
    TRY(rc);
    TRY(engine_logits(e, e->sel_feat.as<float>(), B, cls_feat, C, fl, st));
    TRY(launch_top5_batched(fl, B, C, top5, st));
    return RLCF_OK;
}

int engine_tta_batch_ln(rlcf_engine* e, const float* views, int count, int N, const rlcf_tta_args* a, float* final_logits, int32_t* top5,
                        hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || e->image_bank || e->n_rewards <= 0) { rlcf_set_error("class bank / reward model not set"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(N > 0 && N <= e->max_views && a->sample_k > 0 && a->sample_k <= 32 && a->sample_k <= e->C);
    const bool rn = is_resnet(s.cfg);         // BatchNorm tuning: the batch statistics couple one sample's views, samples run one by one
    RLCF_ARG_CHECK(rn || s.tokens <= 320);
    const size_t per = (size_t)N * 3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const int n_sel = n_selected(a, N), Bmax = e->max_views / N;
    const bool fused = !rn && Bmax >= 2 && a->tta_steps >= 1 && !a->skip_final && n_sel > 0;
    double flops = 0.0;
    int i = 0;
    while (i < count) {
        const int B = fused ? std::min(Bmax, count - i) : 1;
        if (fused && B >= 2) {
            TRY(tta_batch_ln_fused(e, views + (size_t)i * per, B, N, a, final_logits ? final_logits + (size_t)i * e->C : nullptr, top5 + (size_t)i * 5, st));
        } else {
            rlcf_tta_out o{};
            o.top5 = top5 + (size_t)i * 5;
            o.final_logits = final_logits ? final_logits + (size_t)i * e->C : nullptr;
            TRY(engine_tta_sample_ln(e, views + (size_t)i * per, N, a, &o, st));
        }
        flops += e->last_flops;
        i += B;
    }
    e->last_flops = flops / count;
    return RLCF_OK;
}

// ------------------------------------------------------------------ BatchNorm tuning of a ModifiedResNet student
// One iteration of tune_cls_rl.py:183-256 with CLIPCLS_TTA(arch=RN*, only_norm=True): the tuned tensors are the weight / bias of every
// BatchNorm2d whose name contains 'bn' (custom_clip.py:481-485: the downsample BatchNorms stay frozen).  The tuning passes run the
// BatchNorms on BATCH statistics (nn.BatchNorm2d in train mode, running statistics updated, or `_modified_bn_forward` under
// --prior_strength >= 0, tune_cls_rl.py:35-44), step 0 over all N views (the selection reads its logits; the statistics couple the
// views, so the backward covers all N with zero logit gradients outside the selection), later steps over the selected views.
// CLIPCLS_TTA.train() (custom_clip.py:487-497) puts the norm layers in train mode whatever `mode` is, so the final clean-view
// inference ALSO normalises with batch statistics (of that one image): reproduced.  The running statistics the sample leaves behind
// stay in e->bn_stats (rlcf_engine_get_bn_stats) until the next sample resets them.
static int rn_visual_reset(rlcf_engine* e, hipStream_t st) {   // visual.load_state_dict(initial_state_dict) for the flat buffer of a ResNet student
    if (!e->vw_count || !e->vw_dirty) return RLCF_OK;
    RLCF_HIP_CHECK(hipMemcpyAsync(e->vw.p, e->vw_init.p, e->vw_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    e->vw_dirty = false;
    return rn_visual_refresh(e, st, e->vw_init_is_ckpt);
}
// full: every visual parameter is tuned (CLIPCLS_TTA(only_norm=False) on a ModifiedResNet — the parser defaults of tune_cls_rl.py,
// TPT/params.py:23,73): convolution / downsample.1 / attention-pool gradients into e->vw_grad, a second AdamW launch, the derived
// weight forms rebuilt after every step, and the final clean-view inference with the BatchNorms in EVAL form on the running statistics
// the tuning passes left behind (model.eval() is plain nn.Module.eval() when only_norm is off, custom_clip.py:487-497)
int engine_tta_sample_bn(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st, bool full) {
    ClipModel& s = e->model[RLCF_STUDENT];
    TRY(engine_bn_enable(e, st));
    if (full) TRY(engine_rn_visual_enable(e, st));
    if (!full && s.rn.full_enabled && e->vw_dirty) TRY(rn_visual_reset(e, st));
    const size_t vb = full ? e->vw_count * sizeof(float) : 0;
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim;
    const int n_sel = n_selected(a, N), n_e = n_sel * K;
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const size_t nb = (size_t)e->ln_count * sizeof(float);
    const rlcf_tta_out none{};
    if (!out) out = &none;
    TRY(e->ln_feat.ensure((size_t)e->max_views * D * sizeof(float)));
    TRY(e->dfeat.ensure((size_t)e->max_views * D * sizeof(float)));
    TRY(e->bn_dlog.ensure((size_t)N * C * sizeof(float)));
    e->last_flops = 0.0;
    // model.reset(): visual.load_state_dict(initial_state_dict) restores parameters AND buffers (running statistics)
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->bn_stats.p, e->bn_stats_init.p, (size_t)s.rn.n_stats * sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->ln_m.p, 0, nb, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->ln_v.p, 0, nb, st));
    if (full) {
        TRY(rn_visual_reset(e, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->vw_m.p, 0, vb, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->vw_v.p, 0, vb, st));
    }
    const float* cls_feat = e->txt0.as<float>();
    for (int j = 0; j < a->tta_steps; ++j) {
        const int n = j == 0 ? N : n_sel;
        TRY(rn_forward_train(e, s, j == 0 ? views : e->views_sel.as<float>(), n, e->ln_feat.as<float>(), st));
        const float* dlog = e->dlogits.as<float>();
        if (j == 0) {
            TRY(engine_logits(e, e->ln_feat.as<float>(), N, cls_feat, C, e->logits.as<float>(), st));
            TRY(launch_entropy_select(e->logits.as<float>(), N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
            TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, n_sel, (int)img_elems, st));
            TRY(reward_encode(e, n_sel, s.cfg.image_resolution, out->reward_image_features, st));
            TRY(launch_gather_rows(e->logits.as<float>(), C, e->sel_idx.as<int32_t>(), e->sel_logits.as<float>(), C, n_sel, C, st));
            COPY_OUT(out->logits, e->logits.p, (size_t)N * C * sizeof(float));
            COPY_OUT(out->entropy, e->entropy.p, N * sizeof(float));
            COPY_OUT(out->selected_idx, e->sel_idx.p, n_sel * sizeof(int32_t));
        } else {
            TRY(engine_logits(e, e->ln_feat.as<float>(), n_sel, cls_feat, C, e->sel_logits.as<float>(), st));
        }
        TRY(launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, 1, n_sel, C, K, reward_bank(e),
                               a->clipscore_weight, a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), e->clip_score.as<float>(),
                               e->rewards.as<float>(), e->loss.as<float>(), e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        if (j == 0) {
            RLCF_HIP_CHECK(hipMemsetAsync(e->bn_dlog.p, 0, (size_t)N * C * sizeof(float), st));
            TRY(launch_scatter_rows(e->dlogits.as<float>(), e->sel_idx.as<int32_t>(), e->bn_dlog.as<float>(), n_sel, C, st));
            dlog = e->bn_dlog.as<float>();
        }
        // d feat = scale * dlogits @ class_features (custom_clip.py:429-430), then the tower's backward down to the stem's first BatchNorm
        TRY(launch_dimg(dlog, cls_feat, n, C, D, s.logit_scale_exp, e->dfeat.as<float>(), st));
        if (full) RLCF_HIP_CHECK(hipMemsetAsync(e->vw_grad.p, 0, vb, st));
        TRY(rn_backward_bn(e, s, n, e->ln_feat.as<float>(), e->dfeat.as<float>(), e->ln_grad.as<float>(), st, full ? e->vw_grad.as<float>() : nullptr));
        if (j == 0) {
            if (full) COPY_OUT(out->vis_grad, e->vw_grad.p, vb);
            COPY_OUT(out->topk_idx, e->topk_idx.p, (size_t)n_e * sizeof(int32_t));
            COPY_OUT(out->clip_score, e->clip_score.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->rewards, e->rewards.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->loss, e->loss.p, sizeof(float));
            COPY_OUT(out->dlogits, e->dlogits.p, (size_t)n_sel * C * sizeof(float));
            COPY_OUT(out->ln_grad, e->ln_grad.p, nb);
        }
        TRY(launch_grad_nonfinite(e->ln_grad.as<float>(), e->ln_count, 1, e->step_skip.as<int32_t>(), st));
        if (full) TRY(launch_grad_nonfinite(e->vw_grad.as<float>(), (int64_t)e->vw_count, 1, e->step_skip.as<int32_t>(), st, true));
        if (out->step_skipped) COPY_OUT(out->step_skipped + j, e->step_skip.p, sizeof(int32_t));
        TRY(launch_adamw(e->ln_params.as<float>(), e->ln_grad.as<float>(), e->ln_m.as<float>(), e->ln_v.as<float>(), e->ln_count, j + 1,
                         a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), e->ln_count));
        if (full) {
            TRY(launch_adamw(e->vw.as<float>(), e->vw_grad.as<float>(), e->vw_m.as<float>(), e->vw_v.as<float>(), (int64_t)e->vw_count, j + 1,
                             a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)e->vw_count));
            e->vw_dirty = true;
            TRY(rn_visual_refresh(e, st));
        }
    }
    COPY_OUT(out->ln_after, e->ln_params.p, nb);
    if (full) COPY_OUT(out->vis_after, e->vw.p, vb);
    if (!a->skip_final) {
        // norm-layer tuning: the BatchNorms stay in train form (see the header comment); every-parameter tuning: eval form
        TRY(rn_forward_train(e, s, views, 1, e->img_feat.as<float>(), st, full ? 0 : -1));
        TRY(engine_logits(e, e->img_feat.as<float>(), 1, cls_feat, C, e->final_logits.as<float>(), st));
        TRY(launch_top5(e->final_logits.as<float>(), C, e->top5.as<int32_t>(), st));
        COPY_OUT(out->final_logits, e->final_logits.p, (size_t)C * sizeof(float));
        COPY_OUT(out->top5, e->top5.p, 5 * sizeof(int32_t));
    }
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    if (full) TRY(rn_visual_reset(e, st));
    return RLCF_OK;
}

static int tta_sample_backbone(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st, bool full);
int engine_tta_sample_ln(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st) {
    return tta_sample_backbone(e, views, N, a, out, st, false);
}
// every visual parameter tuned (CLIPCLS_TTA only_norm=False, custom_clip.py:477-479)
int engine_tta_sample_visual(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st) {
    TRY(engine_visual_enable(e, st));
    return tta_sample_backbone(e, views, N, a, out, st, true);
}
static int visual_reset(rlcf_engine* e, hipStream_t st) {      // visual.load_state_dict(initial_state_dict) for the flat buffer
    if (!e->vw_count || !e->vw_dirty) return RLCF_OK;
    RLCF_HIP_CHECK(hipMemcpyAsync(e->vw.p, e->vw_init.p, e->vw_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    e->vw_dirty = false;
    return engine_visual_refresh(e, st, e->vw_init_is_ckpt);
}
static int tta_sample_backbone(rlcf_engine* e, const float* views, int N, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st, bool full) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || e->image_bank || e->n_rewards <= 0) { rlcf_set_error("class bank / reward model not set"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(N > 0 && N <= e->max_views && a && a->tta_steps >= 0 && a->sample_k > 0 && a->sample_k <= 32 && a->sample_k <= e->C);
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim;
    const int n_sel = n_selected(a, N), n_e = n_sel * K;
    if (a->tta_steps > 0 && n_sel <= 0) { rlcf_set_error("int(N*selection_p) == 0 views selected (N=%d, p=%g)", N, a->selection_p); return RLCF_ERR_ARG; }
    if (is_resnet(s.cfg)) {
        return engine_tta_sample_bn(e, views, N, a, out, st, full);
    }
    RLCF_ARG_CHECK(s.tokens <= 320);
    const size_t img_elems = (size_t)3 * s.cfg.image_resolution * s.cfg.image_resolution;
    const size_t nb = (size_t)e->ln_count * sizeof(float);
    const rlcf_tta_out none{};
    if (!out) out = &none;
    TRY(e->ln_feat.ensure((size_t)e->max_views * D * sizeof(float)));
    e->last_flops = 0.0;
    // model.reset() (visual.load_state_dict(initial_state_dict), custom_clip.py:456-458) + optimizer state reset
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->ln_m.p, 0, nb, st));
    RLCF_HIP_CHECK(hipMemsetAsync(e->ln_v.p, 0, nb, st));
    const size_t vb = full ? e->vw_count * sizeof(float) : 0;
    if (full) {
        TRY(visual_reset(e, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->vw_m.p, 0, vb, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->vw_v.p, 0, vb, st));
    }
    const float* cls_feat = e->txt0.as<float>();           // cached class text features (custom_clip.py:405-409)
    for (int j = 0; j < a->tta_steps; ++j) {
        if (j == 0) {
            // all N views decide the selection; only the selected ones carry gradient (rows outside idx get zero grad)
            TRY(engine_encode_image(e, RLCF_STUDENT, views, N, e->img_feat.as<float>(), st));
            TRY(engine_logits(e, e->img_feat.as<float>(), N, cls_feat, C, e->logits.as<float>(), st));
            TRY(launch_entropy_select(e->logits.as<float>(), N, C, n_sel, e->entropy.as<float>(), e->sel_idx.as<int32_t>(), st));
            if (a->flags & RLCF_F_NO_SELECTION) TRY(launch_iota(e->sel_idx.as<int32_t>(), n_sel, st));     // retrieval: every row, in order
            TRY(launch_gather_rows(views, (int)img_elems, e->sel_idx.as<int32_t>(), e->views_sel.as<float>(), (int)img_elems, n_sel, (int)img_elems, st));
            TRY(reward_encode(e, n_sel, s.cfg.image_resolution, out->reward_image_features, st));
            COPY_OUT(out->logits, e->logits.p, (size_t)N * C * sizeof(float));
            COPY_OUT(out->entropy, e->entropy.p, N * sizeof(float));
            COPY_OUT(out->selected_idx, e->sel_idx.p, n_sel * sizeof(int32_t));
        }
        TRY(vit_forward_saved(e, s, e->views_sel.as<float>(), n_sel, e->ln_feat.as<float>(), st));
        TRY(engine_logits(e, e->ln_feat.as<float>(), n_sel, cls_feat, C, e->sel_logits.as<float>(), st));
        TRY(launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, 1, n_sel, C, K, reward_bank(e),
                               a->clipscore_weight, a->flags, a->min_entropy_w, e->topk_idx.as<int32_t>(), e->clip_score.as<float>(),
                               e->rewards.as<float>(), e->loss.as<float>(), e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        if (full) RLCF_HIP_CHECK(hipMemsetAsync(e->vw_grad.p, 0, vb, st));
        TRY(vit_backward_ln(e, s, e->ln_feat.as<float>(), n_sel, e->dlogits.as<float>(), e->ln_grad.as<float>(), st, 1,
                            full ? e->vw_grad.as<float>() : nullptr));
        if (j == 0) {
            if (full) COPY_OUT(out->vis_grad, e->vw_grad.p, vb);
            COPY_OUT(out->topk_idx, e->topk_idx.p, (size_t)n_e * sizeof(int32_t));
            COPY_OUT(out->clip_score, e->clip_score.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->rewards, e->rewards.p, (size_t)n_e * sizeof(float));
            COPY_OUT(out->loss, e->loss.p, sizeof(float));
            COPY_OUT(out->dlogits, e->dlogits.p, (size_t)n_sel * C * sizeof(float));
            COPY_OUT(out->ln_grad, e->ln_grad.p, nb);
        }
        // one optimizer over both buffers: an inf / NaN anywhere skips the whole step (GradScaler.step, tpt_cls_rl.py:78)
        TRY(launch_grad_nonfinite(e->ln_grad.as<float>(), e->ln_count, 1, e->step_skip.as<int32_t>(), st));
        if (full) TRY(launch_grad_nonfinite(e->vw_grad.as<float>(), (int64_t)e->vw_count, 1, e->step_skip.as<int32_t>(), st, true));
        if (out->step_skipped) COPY_OUT(out->step_skipped + j, e->step_skip.p, sizeof(int32_t));
        TRY(launch_adamw(e->ln_params.as<float>(), e->ln_grad.as<float>(), e->ln_m.as<float>(), e->ln_v.as<float>(), e->ln_count, j + 1,
                         a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), e->ln_count));
        if (full) {
            TRY(launch_adamw(e->vw.as<float>(), e->vw_grad.as<float>(), e->vw_m.as<float>(), e->vw_v.as<float>(), (int64_t)e->vw_count, j + 1,
                             a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)e->vw_count));
            e->vw_dirty = true;
            TRY(engine_visual_refresh(e, st));
        }
    }
    COPY_OUT(out->ln_after, e->ln_params.p, nb);
    if (full) COPY_OUT(out->vis_after, e->vw.p, vb);
    if (!a->skip_final) {
        // final clean-view inference with the adapted LayerNorms (tune_cls_rl.py:219-221)
        TRY(engine_encode_image(e, RLCF_STUDENT, views, 1, e->img_feat.as<float>(), st));
        TRY(engine_logits(e, e->img_feat.as<float>(), 1, cls_feat, C, e->final_logits.as<float>(), st));
        TRY(launch_top5(e->final_logits.as<float>(), C, e->top5.as<int32_t>(), st));
        COPY_OUT(out->final_logits, e->final_logits.p, (size_t)C * sizeof(float));
        COPY_OUT(out->top5, e->top5.p, 5 * sizeof(int32_t));
    }
    // leave the engine in its pristine state for the prompt path (which assumes frozen, pristine weights)
    RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.p, e->ln_init.p, nb, hipMemcpyDeviceToDevice, st));
    if (full) TRY(visual_reset(e, st));
    return RLCF_OK;
}

// ------------------------------------------------------------------ text-encoder tuning (retrieval, text -> image)
// retrieval/clip_ret_policy.py:106-137 (tune_text) with CLIPRet_TTA(only_visual=False): parameters() = every parameter of the CLIP
// model whose name does not contain 'visual' (custom_models.py:139-147) — token_embedding.weight, positional_embedding, the text
// transformer, ln_final, text_projection and logit_scale — tuned on ONE query caption against a fixed bank of image features.
// Same buffer scheme as the image encoder (engine_visual_enable): the LayerNorm tensors in one small buffer, everything else in a
// flat one, the tower reads the live weights from them, derived copies (transposes, split-f16 pairs, the query's embedding rows)
// follow after every optimizer step and after the reset.
int engine_text_enable(rlcf_engine* e, hipStream_t st) {
    if (e->tw_count) return RLCF_OK;
    ClipModel& m = e->model[RLCF_STUDENT];
    if (!m.finalized) { rlcf_set_error("student model not finalized"); return RLCF_ERR_STATE; }
    if (prec_single(e)) { rlcf_set_error("encoder tuning runs in RLCF_PREC_F32 / RLCF_PREC_F16X3 (RLCF_PREC_F16 is the prompt path's performance mode)"); return RLCF_ERR_STATE; }
    const rlcf_clip_cfg& c = m.cfg;
    const size_t Wt = c.text_width, D = c.embed_dim, W2 = Wt * Wt;
    struct Item { const float** slot; size_t numel; };
    std::vector<Item> items = {{&m.tok_emb, (size_t)c.vocab_size * Wt}, {&m.tpos, (size_t)c.context_length * Wt}, {&m.tproj, Wt * D}};
    for (BlockW& b : m.txt.blk) {
        items.push_back({&b.in_w, 3 * W2}); items.push_back({&b.in_b, 3 * Wt}); items.push_back({&b.out_w, W2}); items.push_back({&b.out_b, Wt});
        items.push_back({&b.fc_w, 4 * W2}); items.push_back({&b.fc_b, 4 * Wt}); items.push_back({&b.proj_w, 4 * W2}); items.push_back({&b.proj_b, Wt});
    }
    size_t total = 0;
    e->tw_slots.clear();
    for (const Item& it : items) { e->tw_slots.push_back(VwSlot{total, it.numel}); total += (it.numel + 63) / 64 * 64; }
    e->tw_slots.push_back(VwSlot{total, 1});              // logit_scale (the parameter, not its exponential)
    total += 64;
    const size_t nb = total * sizeof(float);
    for (DevBuf* d : {&e->tw, &e->tw_init, &e->tw_grad, &e->tw_m, &e->tw_v}) TRY(d->ensure(nb));
    RLCF_HIP_CHECK(hipMemsetAsync(e->tw.p, 0, nb, st));
    for (size_t i = 0; i < items.size(); ++i) {
        const float* old = *items[i].slot;
        float* dst = e->tw.as<float>() + e->tw_slots[i].off;
        RLCF_HIP_CHECK(hipMemcpyAsync(dst, old, items[i].numel * sizeof(float), hipMemcpyDeviceToDevice, st));
        auto sp = m.split_of.find(old);
        if (sp != m.split_of.end()) { const ClipModel::SplitW s = sp->second; m.split_of.erase(sp); m.split_of[dst] = s; }
        *items[i].slot = dst;
    }
    const float* ls = rawp(m, "logit_scale", 1);
    NEED(ls);
    RLCF_HIP_CHECK(hipMemcpyAsync(e->tw.as<float>() + e->tw_slots.back().off, ls, sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->tw_init.p, e->tw.p, nb, hipMemcpyDeviceToDevice, st));
    for (DevBuf* d : {&e->tw_clip, &e->tw_mom}) { TRY(d->ensure(nb)); RLCF_HIP_CHECK(hipMemcpyAsync(d->p, e->tw.p, nb, hipMemcpyDeviceToDevice, st)); }
    // LayerNorms: [ln_final.weight | ln_final.bias | per block ln_1.weight ln_1.bias ln_2.weight ln_2.bias] (transformer_backward's layout)
    const int L = c.text_layers;
    e->tln_count = (int)((4 * L + 2) * Wt);
    const size_t lb = (size_t)e->tln_count * sizeof(float);
    for (DevBuf* d : {&e->tln, &e->tln_init, &e->tln_grad, &e->tln_m, &e->tln_v}) TRY(d->ensure(lb));
    {
        float* P = e->tln.as<float>();
        std::vector<const float**> slots = {&m.lnf_w, &m.lnf_b};
        for (BlockW& b : m.txt.blk) { slots.push_back(&b.ln1_w); slots.push_back(&b.ln1_b); slots.push_back(&b.ln2_w); slots.push_back(&b.ln2_b); }
        for (size_t i = 0; i < slots.size(); ++i) {
            RLCF_HIP_CHECK(hipMemcpyAsync(P + i * Wt, *slots[i], Wt * sizeof(float), hipMemcpyDeviceToDevice, st));
            *slots[i] = P + i * Wt;
        }
        RLCF_HIP_CHECK(hipMemcpyAsync(e->tln_init.p, P, lb, hipMemcpyDeviceToDevice, st));
        for (DevBuf* d : {&e->tln_clip, &e->tln_mom}) { TRY(d->ensure(lb)); RLCF_HIP_CHECK(hipMemcpyAsync(d->p, P, lb, hipMemcpyDeviceToDevice, st)); }
    }
    e->tw_refresh.clear();
    auto add_split = [&](const float* w, size_t numel) {
        auto it = m.split_of.find(w);
        if (it != m.split_of.end()) it->second.lo_zero = false;          // (a TUNED weight leaves the fp16 grid at its first step: three passes)
        if (it != m.split_of.end())
            e->tw_refresh.push_back(VwRefresh{VW_SPLIT, w, nullptr, numel, 0, it->second.hi, it->second.lo, 1.0f / it->second.inv_scale,
                                              it->second.lo == lo_of(it->second.hi)});
    };
    auto add_T = [&](const float* w, const float* wT, size_t rows, size_t cols) {
        if (!wT) return;
        e->tw_refresh.push_back(VwRefresh{VW_TRANSPOSE, w, (float*)wT, rows, cols, nullptr, nullptr, 1.f, 0});
        add_split(wT, rows * cols);
    };
    add_T(m.tproj, m.tprojT, Wt, D);
    for (BlockW& b : m.txt.blk) {
        add_split(b.in_w, 3 * W2); add_split(b.out_w, W2); add_split(b.fc_w, 4 * W2); add_split(b.proj_w, 4 * W2);
        add_T(b.in_w, b.in_wT, 3 * Wt, Wt); add_T(b.out_w, b.out_wT, Wt, Wt); add_T(b.fc_w, b.fc_wT, 4 * Wt, Wt); add_T(b.proj_w, b.proj_wT, Wt, 4 * Wt);
    }
    e->tw_count = total;
    e->tw_dirty = false;
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    return RLCF_OK;
}

// x[i] *= exp(*log_scale)
__global__ void scale_by_exp_kernel(float* __restrict__ x, const float* __restrict__ log_scale, int n) {
    const float s = expf(*log_scale);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] *= s;
}
// out[0] = sum_i a[i] * b[i]  (one block, fixed reduction order)
__global__ void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += a[i] * b[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}
// embedding gradients of one packed text: x[r] = token_embedding[token[r]] + positional_embedding[pos[r]] (model.py:344-346), so
// d positional_embedding[pos[r]] += dX[r], d token_embedding[token[r]] += dX[r].  One thread per column walks the rows in order:
// repeated tokens accumulate in a fixed order, no atomics.
__global__ void embed_grad_kernel(const float* __restrict__ dX, const int32_t* __restrict__ row_token, const int32_t* __restrict__ row_pos,
                                  float* __restrict__ g_tok, float* __restrict__ g_pos, int rows, int width) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    for (int r = 0; r < rows; ++r) {
        const float g = dX[(size_t)r * width + c];
        g_pos[(size_t)row_pos[r] * width + c] += g;
        const int t = row_token[r];
        if (t >= 0) g_tok[(size_t)t * width + c] += g;
    }
}

// the bank of the text -> image direction: n images, their features under the student (the fixed side of logits_per_text,
// CLIPRet_TTA.set_image_features, custom_models.py:91-95) and under every reward model (CLIPRewards.set_image_features,
// retrieval/clip_reward.py:130-137), blocks [n, Dr_m] one after another.  Device pointers, copied.
int engine_set_image_bank(rlcf_engine* e, const float* student_feats, const float* reward_feats, int n, hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (!s.finalized || e->n_rewards <= 0) { rlcf_set_error("student / reward model not set"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(student_feats && reward_feats && n > 0 && n <= e->max_classes);
    const int D = s.cfg.embed_dim;
    e->C = n; e->image_bank = true; e->n_ctx = 0;
    e->sp_max_e = 0; e->sp_groups = 0; e->b_cap = 0;
    TRY(tta_scratch_ensure(e, n));
    TRY(e->txt0.ensure((size_t)n * D * sizeof(float)));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->txt0.p, student_feats, (size_t)n * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    for (int m = 0; m < e->n_rewards; ++m) {
        ClipModel& r = e->model[RLCF_REWARD + m];
        if (!r.finalized) { rlcf_set_error("reward model %d not finalized", m); return RLCF_ERR_STATE; }
        const int Dr = r.cfg.embed_dim;
        TRY(e->rimg[m].ensure((size_t)e->max_views * Dr * sizeof(float)));
        TRY(e->reward_cls[m].ensure((size_t)n * Dr * sizeof(float)));
        RLCF_HIP_CHECK(hipMemcpyAsync(e->reward_cls[m].p, reward_feats, (size_t)n * Dr * sizeof(float), hipMemcpyDeviceToDevice, st));
        reward_feats += (size_t)n * Dr;
    }
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    return RLCF_OK;
}

int engine_text_reset(rlcf_engine* e, hipStream_t st, bool force) {      // clip_model.load_state_dict(initial_state_dict) for the text side
    if (!e->tw_count || (!e->tw_dirty && !force)) return RLCF_OK;
    RLCF_HIP_CHECK(hipMemcpyAsync(e->tw.p, e->tw_init.p, e->tw_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->tln.p, e->tln_init.p, (size_t)e->tln_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    e->tw_dirty = false;
    return refresh_derived(e->tw_refresh, 0, st);
}
static int rebuild_E(ClipModel& m, TextLayout& L, hipStream_t st) {
    build_E_kernel<<<dim3(L.T), dim3(128), 0, st>>>(m.tok_emb, m.tpos, L.row_token.as<int32_t>(), L.row_pos.as<int32_t>(), L.E.as<float>(), L.T,
                                                     m.cfg.text_width);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// tokens: HOST [context_length], the query caption.  Per step: logits_per_text [1, n] of the query against the image bank, top-K
// images, CLIPScore(images_index=...) of the reward model, REINFORCE loss, backward through the whole text encoder, AdamW over the
// two parameter buffers.  Then logits_per_text of the tuned encoder (clip_ret_policy.py:193-195) and the reset (:196).
// Outputs (rlcf_tta_out): logits [n], topk_idx, clip_score, rewards, loss, dlogits [n] of the first step; vis_grad / vis_after = the flat
// text buffer (rlcf_engine_text_param_layout), ln_grad / ln_after = the text LayerNorm buffer; final_logits [n]; step_skipped.
int engine_tta_retrieval_text(rlcf_engine* e, const int32_t* tokens, const rlcf_tta_args* a, const rlcf_tta_out* out, hipStream_t st) {
    ClipModel& s = e->model[RLCF_STUDENT];
    if (e->C <= 0 || !e->image_bank || e->n_rewards <= 0) { rlcf_set_error("image bank / reward model not set"); return RLCF_ERR_STATE; }
    RLCF_ARG_CHECK(tokens && a && a->tta_steps >= 0 && a->sample_k > 0 && a->sample_k <= 32 && a->sample_k <= e->C);
    TRY(engine_text_enable(e, st));
    const int C = e->C, K = a->sample_k, D = s.cfg.embed_dim, Wt = s.cfg.text_width, L = s.cfg.text_layers;
    const rlcf_tta_out none{};
    if (!out) out = &none;
    e->last_flops = 0.0;
    // model.reset_initial() + a fresh optimizer state (clip_ret_policy.py:186-190)
    TRY(engine_text_reset(e, st, false));
    const size_t wb = e->tw_count * sizeof(float), lb = (size_t)e->tln_count * sizeof(float);
    for (DevBuf* d : {&e->tw_m, &e->tw_v}) RLCF_HIP_CHECK(hipMemsetAsync(d->p, 0, wb, st));
    for (DevBuf* d : {&e->tln_m, &e->tln_v}) RLCF_HIP_CHECK(hipMemsetAsync(d->p, 0, lb, st));
    // layouts of the query under the student and under every reward model (one packed sequence each)
    TextLayout& Q = e->qlay[0];
    TRY(build_layout(e, s, Q, tokens, 1, 0, false, RLCF_TEXT_PACKED, st));
    int Wmax = Wt, Dmax = D, Tmax = Q.T;
    for (int m = 0; m < e->n_rewards; ++m) {
        ClipModel& r = e->model[RLCF_REWARD + m];
        RLCF_ARG_CHECK(r.cfg.context_length == s.cfg.context_length);
        TRY(build_layout(e, r, e->qlay[1 + m], tokens, 1, 0, false, RLCF_TEXT_PACKED, st));
        Wmax = std::max(Wmax, r.cfg.text_width); Dmax = std::max(Dmax, r.cfg.embed_dim); Tmax = std::max(Tmax, e->qlay[1 + m].T);
    }
    TRY(tower_ensure(e->tt, Tmax, Wmax, st));
    TRY(tower_ensure(e->st, Q.T, Wt, st));
    TRY(tower_ensure_saved(e->st, Q.T, Wt, L, st));
    TRY(bwd_ensure(e, Q.T, Wt));
    if (prec_x3(e) && (size_t)Tmax * Wmax * 4 > e->a_split_elems) {
        e->a_split_elems = (size_t)Tmax * Wmax * 4;
        TRY(e->a_hi.ensure(e->a_split_elems * 4));
    }
    TRY(e->eot_x.ensure((size_t)Wmax * sizeof(float))); TRY(e->eot_ln.ensure((size_t)Wmax * sizeof(float)));
    TRY(e->u.ensure((size_t)Dmax * sizeof(float))); TRY(e->inv_norm.ensure(sizeof(float))); TRY(e->txt.ensure((size_t)Dmax * sizeof(float)));
    TRY(e->q_feat.ensure((size_t)D * sizeof(float))); TRY(e->q_dfeat.ensure((size_t)D * sizeof(float))); TRY(e->q_ls.ensure(64 * sizeof(float)));
    TRY(e->sp_du.ensure((size_t)D * sizeof(float))); TRY(e->sp_dxe.ensure((size_t)Wt * sizeof(float)));
    auto query_io = [&](const TextLayout& Lq, float* txt) {
        TextPassIO io{};
        io.seqs = Lq.seqs.as<rlcf_seq>(); io.n_seq = Lq.n_seq; io.max_q_len = Lq.max_q_len; io.T = Lq.T; io.n_cls = 1;
        io.attn_pairs = Lq.attn_pairs; io.eot_rows = Lq.eot_rows.as<int32_t>(); io.row_src = nullptr;
        io.eot_x = e->eot_x.as<float>(); io.eot_ln = e->eot_ln.as<float>(); io.u = e->u.as<float>(); io.inv_norm = e->inv_norm.as<float>();
        io.txt = txt;
        return io;
    };
    // reward_model.set_text_features(captions=text) (clip_ret_policy.py:117): the query under every reward model, frozen
    for (int m = 0; m < e->n_rewards; ++m)
        TRY(text_forward(e, e->model[RLCF_REWARD + m], e->qlay[1 + m], e->tt, nullptr, query_io(e->qlay[1 + m], e->rimg[m].as<float>()), false, st));
    const TextPassIO io = query_io(Q, e->q_feat.as<float>());
    float* const ls = e->tw.as<float>() + e->tw_slots.back().off;
    float* const G = e->tw_grad.as<float>();
    float* const GL = e->tln_grad.as<float>();
    auto logits_per_text = [&](float* dst) -> int {
        TRY(gemm(e, e->q_feat.as<float>(), D, e->txt0.as<float>(), D, nullptr, nullptr, 0, nullptr, 0, dst, C, 1, C, D, 1.f, RLCF_EPI_NONE, st));
        scale_by_exp_kernel<<<dim3(std::min((C + 255) / 256, 1024)), dim3(256), 0, st>>>(dst, ls, C);
        RLCF_LAUNCH_CHECK();
        return RLCF_OK;
    };
    for (int j = 0; j < a->tta_steps; ++j) {
        TRY(text_forward(e, s, Q, e->st, nullptr, io, true, st));
        TRY(logits_per_text(e->sel_logits.as<float>()));
        TRY(launch_reward_loss_bank(e->sel_logits.as<float>(), C, nullptr, 1, 1, C, K, reward_bank(e), a->clipscore_weight, a->flags,
                                    a->min_entropy_w, e->topk_idx.as<int32_t>(), e->clip_score.as<float>(), e->rewards.as<float>(),
                                    e->loss.as<float>(), e->dlogits.as<float>(), e->rl_stats.as<float>(), st));
        RLCF_HIP_CHECK(hipMemsetAsync(G, 0, wb, st));
        RLCF_HIP_CHECK(hipMemsetAsync(GL, 0, lb, st));
        // logits = exp(logit_scale) * <feat, bank>:  d logit_scale = <dlogits, logits>,  d feat = exp(logit_scale) * dlogits @ bank
        dot_kernel<<<dim3(1), dim3(256), 0, st>>>(e->dlogits.as<float>(), e->sel_logits.as<float>(), C, G + e->tw_slots.back().off);
        RLCF_LAUNCH_CHECK();
        TRY(launch_dimg(e->dlogits.as<float>(), e->txt0.as<float>(), 1, C, D, 1.0f, e->q_dfeat.as<float>(), st));
        scale_by_exp_kernel<<<dim3(1), dim3(256), 0, st>>>(e->q_dfeat.as<float>(), ls, D);
        RLCF_LAUNCH_CHECK();
        // text_features = normalize(ln_final(x[eot]) @ text_projection) (model.py:351-356, custom_models.py:82-83)
        float *du = e->sp_du.as<float>(), *dxe = e->sp_dxe.as<float>();
        TRY(launch_l2norm_bwd(io.txt, e->q_dfeat.as<float>(), io.inv_norm, du, 1, D, st));
        TRY(wgrad(e, io.eot_ln, Wt, Wt, du, D, D, 1, G + e->tw_slots[2].off, nullptr, st));
        TRY(gemm(e, du, D, s.tproj, D, nullptr, nullptr, 0, nullptr, 0, dxe, Wt, 1, Wt, D, 1.f, RLCF_EPI_NONE, st));
        TRY(launch_layernorm_bwd(io.eot_x, s.lnf_w, dxe, nullptr, dxe, GL, GL + Wt, 1, Wt, st));
        RLCF_HIP_CHECK(hipMemsetAsync(e->dX.p, 0, (size_t)io.T * Wt * sizeof(float), st));
        TRY(launch_scatter_rows(dxe, io.eot_rows, e->dX.as<float>(), 1, Wt, st));
        TRY(transformer_backward(e, s.txt, e->st, io.seqs, io.n_seq, Q.max_keys, io.attn_pairs, 1, io.T, st, GL, 0, 0, 0, G, e->tw_slots.data() + 3));
        embed_grad_kernel<<<dim3((Wt + 127) / 128), dim3(128), 0, st>>>(e->dX.as<float>(), Q.row_token.as<int32_t>(), Q.row_pos.as<int32_t>(),
                                                                       G + e->tw_slots[0].off, G + e->tw_slots[1].off, io.T, Wt);
        RLCF_LAUNCH_CHECK();
        if (j == 0) {
            COPY_OUT(out->logits, e->sel_logits.p, (size_t)C * sizeof(float));
            COPY_OUT(out->topk_idx, e->topk_idx.p, (size_t)K * sizeof(int32_t));
            COPY_OUT(out->clip_score, e->clip_score.p, (size_t)K * sizeof(float));
            COPY_OUT(out->rewards, e->rewards.p, (size_t)K * sizeof(float));
            COPY_OUT(out->loss, e->loss.p, sizeof(float));
            COPY_OUT(out->dlogits, e->dlogits.p, (size_t)C * sizeof(float));
            COPY_OUT(out->vis_grad, G, wb);
            COPY_OUT(out->ln_grad, GL, lb);
            COPY_OUT(out->reward_image_features, e->rimg[0].p, (size_t)e->model[RLCF_REWARD].cfg.embed_dim * sizeof(float));
        }
        TRY(launch_grad_nonfinite(GL, e->tln_count, 1, e->step_skip.as<int32_t>(), st));
        TRY(launch_grad_nonfinite(G, (int64_t)e->tw_count, 1, e->step_skip.as<int32_t>(), st, true));
        if (out->step_skipped) COPY_OUT(out->step_skipped + j, e->step_skip.p, sizeof(int32_t));
        TRY(launch_adamw(e->tln.as<float>(), GL, e->tln_m.as<float>(), e->tln_v.as<float>(), e->tln_count, j + 1, a->lr, a->beta1, a->beta2,
                         a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), e->tln_count));
        TRY(launch_adamw(e->tw.as<float>(), G, e->tw_m.as<float>(), e->tw_v.as<float>(), (int64_t)e->tw_count, j + 1, a->lr, a->beta1, a->beta2,
                         a->eps, a->weight_decay, st, e->step_skip.as<int32_t>(), (int64_t)e->tw_count));
        e->tw_dirty = true;
        TRY(refresh_derived(e->tw_refresh, 0, st));
        TRY(rebuild_E(s, Q, st));
    }
    COPY_OUT(out->vis_after, e->tw.p, wb);
    COPY_OUT(out->ln_after, e->tln.p, lb);
    if (!a->skip_final) {
        TRY(text_forward(e, s, Q, e->st, nullptr, io, false, st));
        TRY(logits_per_text(e->final_logits.as<float>()));
        COPY_OUT(out->final_logits, e->final_logits.p, (size_t)C * sizeof(float));
    }
    TRY(engine_text_reset(e, st, false));
    return RLCF_OK;
}
