// Device-side generation of the N views of one test image: what the reference does on CPU workers with
// torchvision/PIL per sample (TPT/data/datautils.py:76-128 AugMixAugmenter with an empty aug_list; transforms of
// TPT/tpt_cls_rl.py:132-150) and then ships over PCIe as 38.5 MB of float views (:236-248).  Here ONE decoded uint8 image goes up
// and the views are produced in HBM:
//   view 0      Resize(res, bicubic) + CenterCrop(res)
//   view 1..n   resized_crop(top, left, h, w -> res x res, bilinear) [+ horizontal flip]
//   all         ToTensor (/255) + Normalize(mean, std), float32 NCHW
// The resampler is Pillow's 8-bit two-pass separable one (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc with
// PRECISION_BITS = 22, horizontal pass rounded to 8 bits, then vertical pass), reproduced bit for bit: double-precision
// coefficients without FMA contraction, int32 accumulation, the same rounding and clamping.
#include "kernels.h"
#include <cmath>

#include <cstring>
#include <mutex>

// Host -> device hand-over of the small parameter arrays (crop boxes, AugMix / hard_aug plans): a copy from PAGEABLE host memory is
// stream-ordered but blocks the calling thread until every earlier launch of that stream has finished — one full device drain per test
// image in a loop that makes its views on the fly.  The arrays go through a ring of pinned slots instead (host memcpy, then a truly
// asynchronous copy; an event per slot guards its reuse), so the call returns at once and the caller's arrays are free again.
#define VW_SLOTS 32
#define VW_SLOT_BYTES (128 * 1024)
#define VW_MAX_DEVICES 16
static int views_upload(void* dst_dev, const void* src_host, size_t bytes, hipStream_t st) {
    // one ring (pinned slots + their events) PER DEVICE: an event belongs to the device it was created on, and recording it on a stream of
    // another GPU fails.  A ring whose set-up failed half way is torn down and that device keeps the blocking form.
    struct Ring { char* mem = nullptr; hipEvent_t ev[VW_SLOTS]; int next = 0; bool failed = false; };
    static std::mutex mu;
    static Ring rings[VW_MAX_DEVICES];
    if (bytes == 0) return RLCF_OK;
    int dev = 0;
    RLCF_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    Ring* r = (dev >= 0 && dev < VW_MAX_DEVICES) ? &rings[dev] : nullptr;
    if (r && !r->mem && !r->failed && bytes <= VW_SLOT_BYTES) {
        int made = 0;
        if (hipHostMalloc((void**)&r->mem, (size_t)VW_SLOTS * VW_SLOT_BYTES, hipHostMallocDefault) != hipSuccess) r->mem = nullptr;
        else for (; made < VW_SLOTS; ++made) if (hipEventCreateWithFlags(&r->ev[made], hipEventDisableTiming) != hipSuccess) break;
        if (!r->mem || made < VW_SLOTS) {
            for (int i = 0; i < made; ++i) (void)hipEventDestroy(r->ev[i]);
            if (r->mem) (void)hipHostFree(r->mem);
            r->mem = nullptr; r->failed = true;
            (void)hipGetLastError();
        }
    }
    if (!r || !r->mem || bytes > VW_SLOT_BYTES) {         // (too large, or no pinned ring on this device: the blocking form)
        RLCF_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        RLCF_HIP_CHECK(hipStreamSynchronize(st));
        return RLCF_OK;
    }
    const int slot = r->next;
    r->next = (r->next + 1) % VW_SLOTS;
    RLCF_HIP_CHECK(hipEventSynchronize(r->ev[slot]));     // (its previous copy has been consumed; a fresh event is complete)
    char* p = r->mem + (size_t)slot * VW_SLOT_BYTES;
    memcpy(p, src_host, bytes);
    RLCF_HIP_CHECK(hipMemcpyAsync(dst_dev, p, bytes, hipMemcpyHostToDevice, st));
    RLCF_HIP_CHECK(hipEventRecord(r->ev[slot], st));
    return RLCF_OK;
}

#define VW_KMAX 64                 // taps per output sample: ceil(support)*2+1 <= 64  (bilinear: downscale <= 31x, bicubic <= 15.5x)
#define VW_PREC 22

struct ViewGeom { int top, left, h, w, flip, out_w, out_h, off_x, off_y, bicubic; };

__device__ __forceinline__ ViewGeom view_geom(const rlcf_crop* crops, int v, int H, int W, int res, int nw, int nh, int off_x, int off_y) {
    ViewGeom g;
    if (v == 0) { g.top = 0; g.left = 0; g.h = H; g.w = W; g.flip = 0; g.out_w = nw; g.out_h = nh; g.off_x = off_x; g.off_y = off_y; g.bicubic = 1; }
    else {
        const rlcf_crop c = crops[v - 1];
        g.top = c.top; g.left = c.left; g.h = c.h; g.w = c.w; g.flip = c.flip; g.out_w = res; g.out_h = res; g.off_x = 0; g.off_y = 0; g.bicubic = 0;
    }
    return g;
}

__device__ double vw_filter(int bicubic, double x) {
#pragma clang fp contract(off)
    x = fabs(x);
    if (!bicubic) return x < 1.0 ? 1.0 - x : 0.0;
    const double a = -0.5;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// coefficient tables: tab[((v*2 + axis)*res + i)] = {xmin, count, k[VW_KMAX]}; one thread per (view, axis, output sample of the window)
__global__ void views_coeffs_kernel(const rlcf_crop* __restrict__ crops, int n_views, int H, int W, int res, int nw, int nh, int off_x, int off_y,
                                    int32_t* __restrict__ tab) {
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_views * 2 * res) return;
    const int i = idx % res, axis = (idx / res) & 1, v = idx / (2 * res);
    const ViewGeom g = view_geom(crops, v, H, W, res, nw, nh, off_x, off_y);
    const int in_size = axis == 0 ? g.w : g.h, out_size = axis == 0 ? g.out_w : g.out_h, xx = i + (axis == 0 ? g.off_x : g.off_y);
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = (g.bicubic ? 2.0 : 1.0) * filterscale;
    const double center = (xx + 0.5) * scale, ss = 1.0 / filterscale;
    int lo = (int)(center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5);
    if (hi > in_size) hi = in_size;
    int n = hi - lo;
    if (n > VW_KMAX) n = VW_KMAX;                         // the host refuses geometries that need more
    int32_t* t = tab + (size_t)idx * (VW_KMAX + 2);
    double ww = 0.0;
    for (int x = 0; x < n; ++x) ww += vw_filter(g.bicubic, (x + lo - center + 0.5) * ss);
    for (int x = 0; x < n; ++x) {
        double w = vw_filter(g.bicubic, (x + lo - center + 0.5) * ss);
        if (ww != 0.0) w /= ww;
        t[2 + x] = w < 0 ? (int)(-0.5 + w * (double)(1 << VW_PREC)) : (int)(0.5 + w * (double)(1 << VW_PREC));
    }
    t[0] = lo; t[1] = n;
}

__device__ __forceinline__ int clip8(int v) { v >>= VW_PREC; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass: tmp[v][y][x][c] for the rows of the view's box and the res output columns of its window
__global__ void views_horizontal_kernel(const uint8_t* __restrict__ img, const rlcf_crop* __restrict__ crops, int H, int W, int res, int nw, int nh,
                                        int off_x, int off_y, const int32_t* __restrict__ tab, uint8_t* __restrict__ tmp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, v = blockIdx.z;
    const ViewGeom g = view_geom(crops, v, H, W, res, nw, nh, off_x, off_y);
    if (x >= res || y >= g.h) return;
    const int32_t* t = tab + ((size_t)(v * 2 + 0) * res + x) * (VW_KMAX + 2);
    const int lo = t[0], n = t[1];
    const uint8_t* row = img + ((size_t)(g.top + y) * W + g.left + lo) * 3;
    int s0 = 1 << (VW_PREC - 1), s1 = s0, s2 = s0;
    for (int k = 0; k < n; ++k) {
        const int c = t[2 + k];
        s0 += row[3 * k] * c; s1 += row[3 * k + 1] * c; s2 += row[3 * k + 2] * c;
    }
    uint8_t* o = tmp + (((size_t)v * H + y) * res + x) * 3;
    o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}

// vertical pass + flip + ToTensor + Normalize
__global__ void views_vertical_kernel(const rlcf_crop* __restrict__ crops, int H, int W, int res, int nw, int nh, int off_x, int off_y,
                                      const int32_t* __restrict__ tab, const uint8_t* __restrict__ tmp, float m0, float m1, float m2, float d0,
                                      float d1, float d2, float* __restrict__ out, uint8_t* __restrict__ u8_out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, v = blockIdx.z;
    if (x >= res) return;
    const ViewGeom g = view_geom(crops, v, H, W, res, nw, nh, off_x, off_y);
    const int32_t* t = tab + ((size_t)(v * 2 + 1) * res + y) * (VW_KMAX + 2);
    const int lo = t[0], n = t[1];
    const uint8_t* col = tmp + (((size_t)v * H + lo) * res + x) * 3;
    int s0 = 1 << (VW_PREC - 1), s1 = s0, s2 = s0;
    for (int k = 0; k < n; ++k) {
        const int c = t[2 + k];
        const uint8_t* p = col + (size_t)k * res * 3;
        s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
    }
    const int ox = g.flip ? res - 1 - x : x;
    float* o = out + (size_t)v * 3 * res * res + (size_t)y * res + ox;
    const size_t plane = (size_t)res * res;
    o[0] = ((float)clip8(s0) / 255.0f - m0) / d0;
    o[plane] = ((float)clip8(s1) / 255.0f - m1) / d1;
    o[2 * plane] = ((float)clip8(s2) / 255.0f - m2) / d2;
    if (u8_out) {                       // the 8-bit view itself (HWC), input of the AugMix op chains
        uint8_t* u = u8_out + ((size_t)v * plane + (size_t)y * res + ox) * 3;
        u[0] = (uint8_t)clip8(s0); u[1] = (uint8_t)clip8(s1); u[2] = (uint8_t)clip8(s2);
    }
}

static void view0_geometry(int H, int W, int res, int* nw, int* nh, int* off_x, int* off_y) {
    // torchvision _compute_resized_output_size (shorter side -> res, longer side int(res*long/short)) and center_crop offsets
    const int sh = W <= H ? W : H, lg = W <= H ? H : W;
    const int nl = (int)((double)res * (double)lg / (double)sh);
    *nw = W <= H ? res : nl;
    *nh = W <= H ? nl : res;
    *off_y = (int)lrint((*nh - res) / 2.0);        // Python round(): half to even, as lrint in the default rounding mode
    *off_x = (int)lrint((*nw - res) / 2.0);
}

size_t views_scratch_bytes(int H, int n_views, int res) {
    return (size_t)n_views * 2 * res * (VW_KMAX + 2) * sizeof(int32_t) + (((size_t)n_views * H * res * 3 + 15) & ~(size_t)15) + (size_t)n_views * sizeof(rlcf_crop);
}

int launch_make_views(const uint8_t* image, int H, int W, const rlcf_crop* crops_host, int n_crops, int res, const float* mean3,
                      const float* std3, float* views, void* scratch, size_t scratch_bytes, hipStream_t st, uint8_t* u8_out) {
    RLCF_ARG_CHECK(image && views && scratch && H > 0 && W > 0 && H <= 65535 && res > 0 && n_crops >= 0 && n_crops < 65535 &&
                   (n_crops == 0 || crops_host) && mean3 && std3);
    const int n_views = 1 + n_crops;
    RLCF_ARG_CHECK(scratch_bytes >= views_scratch_bytes(H, n_views, res));
    int nw, nh, off_x, off_y;
    view0_geometry(H, W, res, &nw, &nh, &off_x, &off_y);
    // tap budget: view 0 shrinks by min(H,W)/res with the bicubic (support 2) filter, a crop by at most max(H,W)/res (bilinear)
    const double s0 = (double)(W <= H ? W : H) / res, s1 = (double)(W <= H ? H : W) / res;
    const int k0 = (int)ceil(2.0 * (s0 < 1.0 ? 1.0 : s0)) * 2 + 1, k1 = (int)ceil(s1 < 1.0 ? 1.0 : s1) * 2 + 1;
    if (k0 > VW_KMAX || k1 > VW_KMAX) {
        rlcf_set_error("make_views: a %dx%d image needs %d / %d resampling taps (> %d): downscale it first", H, W, k0, k1, VW_KMAX);
        return RLCF_ERR_ARG;
    }
    for (int i = 0; i < n_crops; ++i) {
        const rlcf_crop& c = crops_host[i];
        if (c.h <= 0 || c.w <= 0 || c.top < 0 || c.left < 0 || c.top + c.h > H || c.left + c.w > W) {
            rlcf_set_error("make_views: crop %d (top %d, left %d, h %d, w %d) leaves the %dx%d image", i, c.top, c.left, c.h, c.w, H, W);
            return RLCF_ERR_ARG;
        }
    }
    int32_t* tab = (int32_t*)scratch;
    uint8_t* tmp = (uint8_t*)scratch + (size_t)n_views * 2 * res * (VW_KMAX + 2) * sizeof(int32_t);
    rlcf_crop* crops = (rlcf_crop*)(tmp + (((size_t)n_views * H * res * 3 + 15) & ~(size_t)15));
    if (n_crops) { int rcu = views_upload(crops, crops_host, (size_t)n_crops * sizeof(rlcf_crop), st); if (rcu != RLCF_OK) return rcu; }
    const int nt = n_views * 2 * res;
    views_coeffs_kernel<<<dim3((nt + 127) / 128), dim3(128), 0, st>>>(crops, n_views, H, W, res, nw, nh, off_x, off_y, tab);
    RLCF_LAUNCH_CHECK();
    views_horizontal_kernel<<<dim3((res + 63) / 64, H, n_views), dim3(64), 0, st>>>(image, crops, H, W, res, nw, nh, off_x, off_y, tab, tmp);
    RLCF_LAUNCH_CHECK();
    views_vertical_kernel<<<dim3((res + 63) / 64, res, n_views), dim3(64), 0, st>>>(crops, H, W, res, nw, nh, off_x, off_y, tab, tmp, mean3[0],
                                                                                  mean3[1], mean3[2], std3[0], std3[1], std3[2], views, u8_out);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// AugMix op chains (TPT/data/datautils.py:94-110 `augmix` with aug_list = TPT/data/augmix_ops.py:144-147, switched on for the
// fine-grained sets, tpt_cls_rl.py:149-150): every augmented view is  m * pre(x) + (1 - m) * sum_i w_i * pre(chain_i(x)),  three
// chains of 1-3 Pillow ops on the 8-bit view.  The ops are reproduced bit for bit: ImageOps.autocontrast / equalize / posterize /
// solarize are per-band 256-entry look-up tables (built from the band histograms with Pillow's arithmetic), Image.rotate /
// Image.transform(AFFINE, BILINEAR) are libImaging's affine_transform + bilinear_filter32RGB (double arithmetic without FMA
// contraction, truncation to 8 bits, 0 outside the source).  Random draws (op, level, sign, Dirichlet / Beta weights) stay on the
// host, made with the numpy calls the reference makes, in the same order.
#define AUG_NONE (-1)
enum { AUG_AUTOCONTRAST = 0, AUG_EQUALIZE = 1, AUG_POSTERIZE = 2, AUG_ROTATE = 3, AUG_SOLARIZE = 4, AUG_SHEAR_X = 5, AUG_SHEAR_Y = 6,
       AUG_TRANSLATE_X = 7, AUG_TRANSLATE_Y = 8 };
__device__ __forceinline__ bool aug_is_lut(int op) { return op == AUG_AUTOCONTRAST || op == AUG_EQUALIZE || op == AUG_POSTERIZE || op == AUG_SOLARIZE; }

// source image of chain c in round r: round 0 reads the view itself, later rounds the other ping-pong buffer
__device__ __forceinline__ const uint8_t* aug_src(const uint8_t* xo, const uint8_t* b0, const uint8_t* b1, int round, int chain, size_t img) {
    return round == 0 ? xo + (size_t)(1 + chain / 3) * img : (round == 1 ? b0 : b1) + (size_t)chain * img;
}

// one block per chain: band histograms of the source (LDS atomics) -> the op's 3 x 256 look-up table
__global__ __launch_bounds__(256) void augmix_lut_kernel(const uint8_t* __restrict__ xo, const uint8_t* __restrict__ b0,
                                                         const uint8_t* __restrict__ b1, const rlcf_augmix_op* __restrict__ ops, int round,
                                                         int res, uint8_t* __restrict__ luts) {
#pragma clang fp contract(off)
    const int chain = blockIdx.x;
    const rlcf_augmix_op op = ops[chain * 3 + round];
    if (!aug_is_lut(op.op)) return;
    uint8_t* lut = luts + (size_t)chain * 768;
    if (op.op == AUG_POSTERIZE || op.op == AUG_SOLARIZE) {
        for (int i = threadIdx.x; i < 768; i += 256) {
            const int v = i & 255;
            lut[i] = op.op == AUG_POSTERIZE ? (uint8_t)(v & ~((1 << (8 - op.ip)) - 1)) : (uint8_t)(v < op.ip ? v : 255 - v);
        }
        return;
    }
    __shared__ int hist[768];
    for (int i = threadIdx.x; i < 768; i += 256) hist[i] = 0;
    __syncthreads();
    const size_t img = (size_t)res * res * 3;
    const uint8_t* src = aug_src(xo, b0, b1, round, chain, img);
    for (size_t i = threadIdx.x; i < img; i += 256) atomicAdd(&hist[(int)(i % 3) * 256 + src[i]], 1);
    __syncthreads();
    if (threadIdx.x < 3) {
        const int* h = hist + threadIdx.x * 256;
        uint8_t* l = lut + threadIdx.x * 256;
        if (op.op == AUG_AUTOCONTRAST) {            // ImageOps.autocontrast(cutoff=0)
            int lo = 0, hi = 255;
            while (lo < 256 && !h[lo]) ++lo;
            while (hi >= 0 && !h[hi]) --hi;
            if (hi <= lo) { for (int i = 0; i < 256; ++i) l[i] = (uint8_t)i; }
            else {
                const double scale = 255.0 / (double)(hi - lo), offset = -(double)lo * scale;
                for (int i = 0; i < 256; ++i) {
                    const double t = (double)i * scale;
                    int v = (int)(t + offset);
                    l[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
            }
        } else {                                      // ImageOps.equalize
            int nz = 0, last = 0, total = 0;
            for (int i = 0; i < 256; ++i) if (h[i]) { ++nz; last = h[i]; total += h[i]; }
            const int step = nz <= 1 ? 0 : (total - last) / 255;
            if (!step) { for (int i = 0; i < 256; ++i) l[i] = (uint8_t)i; }
            else {
                int n = step / 2;
                for (int i = 0; i < 256; ++i) { const int q = n / step; l[i] = (uint8_t)(q > 255 ? 255 : q); n += h[i]; }
            }
        }
    }
}

// one thread per (chain, pixel): look-up table, affine bilinear resample, or copy (chains shorter than the round count)
__global__ __launch_bounds__(256) void augmix_apply_kernel(const uint8_t* __restrict__ xo, const uint8_t* __restrict__ b0,
                                                           const uint8_t* __restrict__ b1, const rlcf_augmix_op* __restrict__ ops, int round,
                                                           int res, const uint8_t* __restrict__ luts, uint8_t* __restrict__ dst_all) {
#pragma clang fp contract(off)
    const int chain = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= res * res) return;
    const rlcf_augmix_op op = ops[chain * 3 + round];
    const size_t img = (size_t)res * res * 3;
    const uint8_t* src = aug_src(xo, b0, b1, round, chain, img);
    uint8_t* dst = dst_all + (size_t)chain * img + (size_t)p * 3;
    if (aug_is_lut(op.op)) {
        const uint8_t* lut = luts + (size_t)chain * 768;
        dst[0] = lut[src[(size_t)p * 3]]; dst[1] = lut[256 + src[(size_t)p * 3 + 1]]; dst[2] = lut[512 + src[(size_t)p * 3 + 2]];
        return;
    }
    if (op.op == AUG_NONE) { dst[0] = src[(size_t)p * 3]; dst[1] = src[(size_t)p * 3 + 1]; dst[2] = src[(size_t)p * 3 + 2]; return; }
    // Geometry.c: affine_transform at the pixel centre, bilinear_filter32RGB
    const int x = p % res, y = p / res;
    const double xc = (double)x + 0.5, yc = (double)y + 0.5;
    double xin = op.c[0] * xc + op.c[1] * yc + op.c[2];
    double yin = op.c[3] * xc + op.c[4] * yc + op.c[5];
    if (xin < 0.0 || xin >= (double)res || yin < 0.0 || yin >= (double)res) { dst[0] = 0; dst[1] = 0; dst[2] = 0; return; }
    xin -= 0.5; yin -= 0.5;
    const int xf = (int)floor(xin), yf = (int)floor(yin);
    const double dx = xin - (double)xf, dy = yin - (double)yf;
    const int x0 = min(max(xf, 0), res - 1), x1 = min(max(xf + 1, 0), res - 1), y0 = min(max(yf, 0), res - 1);
    const bool has2 = yf + 1 >= 0 && yf + 1 < res;
    const uint8_t* r0 = src + (size_t)y0 * res * 3;
    const uint8_t* r1 = src + (size_t)(has2 ? yf + 1 : y0) * res * 3;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const double a = (double)r0[x0 * 3 + b], bb = (double)r0[x1 * 3 + b];
        const double v1 = a + (bb - a) * dx;
        double v2 = v1;
        if (has2) { const double c = (double)r1[x0 * 3 + b], d = (double)r1[x1 * 3 + b]; v2 = c + (d - c) * dx; }
        dst[b] = (uint8_t)(v1 + (v2 - v1) * dy);
    }
}

// views[1 + v] = m * pre(x) + (1 - m) * ((0 + w0 pre(c0)) + w1 pre(c1)) + w2 pre(c2)),  float32, no FMA contraction
__global__ __launch_bounds__(256) void augmix_mix_kernel(const uint8_t* __restrict__ xo, const uint8_t* __restrict__ fin, const float* __restrict__ w,
                                                         const float* __restrict__ m, int res, float m0, float m1, float m2, float d0, float d1,
                                                         float d2, float* __restrict__ views) {
#pragma clang fp contract(off)
    const int v = blockIdx.y;                          // augmented view index (0-based; view 1 + v of the output)
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= res * res) return;
    const size_t plane = (size_t)res * res, img = plane * 3;
    const float mean[3] = {m0, m1, m2}, sd[3] = {d0, d1, d2};
    const float mm = m[v], om = 1.0f - mm;
    const uint8_t* x = xo + (size_t)(1 + v) * img + (size_t)p * 3;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float xp = ((float)x[b] / 255.0f - mean[b]) / sd[b];
        float mix = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float pi = ((float)fin[(size_t)(v * 3 + i) * img + (size_t)p * 3 + b] / 255.0f - mean[b]) / sd[b];
            const float t = w[v * 3 + i] * pi;
            mix = mix + t;
        }
        const float a = mm * xp, c = om * mix;
        views[(size_t)(1 + v) * img + (size_t)b * plane + p] = a + c;
    }
}

size_t views_augmix_scratch_bytes(int H, int n_views, int res) {
    const size_t img = (size_t)res * res * 3, chains = (size_t)(n_views - 1) * 3;
    size_t n = (views_scratch_bytes(H, n_views, res) + 255) & ~(size_t)255;
    n += ((size_t)n_views * img + 255) & ~(size_t)255;                     // the 8-bit views
    n += 2 * ((chains * img + 255) & ~(size_t)255);                        // ping-pong chain images
    n += (chains * 768 + 255) & ~(size_t)255;                              // look-up tables
    n += (chains * 3 * sizeof(rlcf_augmix_op) + 255) & ~(size_t)255;       // the plan
    n += ((size_t)(n_views - 1) * 4 * sizeof(float) + 255) & ~(size_t)255; // w, m
    return n;
}

// the AugMix rounds + the mix on 8-bit views already in `xo` ([n_views] images, view 0 unused); p = scratch after the views
static int augmix_stage(const uint8_t* xo, int n_crops, int res, const float* mean3, const float* std3, const rlcf_augmix_op* ops_host,
                        const float* w_host, const float* m_host, float* views, char* p, hipStream_t st) {
    const int chains = n_crops * 3;
    for (int i = 0; i < chains * 3; ++i) {
        const rlcf_augmix_op& o = ops_host[i];
        if (o.op < AUG_NONE || o.op > AUG_TRANSLATE_Y || (o.op == AUG_POSTERIZE && (o.ip < 0 || o.ip > 8))) {
            rlcf_set_error("make_views_augmix: op %d of chain %d is not an AugMix op (op %d, parameter %d)", i % 3, i / 3, o.op, o.ip);
            return RLCF_ERR_ARG;
        }
    }
    const size_t img = (size_t)res * res * 3;
    uint8_t* b0 = (uint8_t*)p; p += ((size_t)chains * img + 255) & ~(size_t)255;
    uint8_t* b1 = (uint8_t*)p; p += ((size_t)chains * img + 255) & ~(size_t)255;
    uint8_t* luts = (uint8_t*)p; p += ((size_t)chains * 768 + 255) & ~(size_t)255;
    rlcf_augmix_op* ops = (rlcf_augmix_op*)p; p += ((size_t)chains * 3 * sizeof(rlcf_augmix_op) + 255) & ~(size_t)255;
    float* wm = (float*)p;
    { int rcu = views_upload(ops, ops_host, (size_t)chains * 3 * sizeof(rlcf_augmix_op), st); if (rcu != RLCF_OK) return rcu; }
    { int rcu = views_upload(wm, w_host, (size_t)n_crops * 3 * sizeof(float), st); if (rcu != RLCF_OK) return rcu; }
    { int rcu = views_upload(wm + n_crops * 3, m_host, (size_t)n_crops * sizeof(float), st); if (rcu != RLCF_OK) return rcu; }
    const dim3 grid((res * res + 255) / 256, chains);
    for (int round = 0; round < 3; ++round) {                      // round r applies op r of every chain: b0 <- x, b1 <- b0, b0 <- b1
        augmix_lut_kernel<<<dim3(chains), dim3(256), 0, st>>>(xo, b0, b1, ops, round, res, luts);
        RLCF_LAUNCH_CHECK();
        augmix_apply_kernel<<<grid, dim3(256), 0, st>>>(xo, b0, b1, ops, round, res, luts, round == 1 ? b1 : b0);
        RLCF_LAUNCH_CHECK();
    }
    augmix_mix_kernel<<<dim3((res * res + 255) / 256, n_crops), dim3(256), 0, st>>>(xo, b0, wm, wm + n_crops * 3, res, mean3[0], mean3[1], mean3[2],
                                                                                   std3[0], std3[1], std3[2], views);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

int launch_make_views_augmix(const uint8_t* image, int H, int W, const rlcf_crop* crops_host, int n_crops, int res, const float* mean3,
                             const float* std3, const rlcf_augmix_op* ops_host, const float* w_host, const float* m_host, float* views,
                             void* scratch, size_t scratch_bytes, hipStream_t st) {
    RLCF_ARG_CHECK(n_crops > 0 && ops_host && w_host && m_host && scratch && res > 0);
    const int n_views = 1 + n_crops;
    RLCF_ARG_CHECK(scratch_bytes >= views_augmix_scratch_bytes(H, n_views, res));
    const size_t img = (size_t)res * res * 3;
    char* p = (char*)scratch + ((views_scratch_bytes(H, n_views, res) + 255) & ~(size_t)255);
    uint8_t* xo = (uint8_t*)p; p += ((size_t)n_views * img + 255) & ~(size_t)255;
    int rc = launch_make_views(image, H, W, crops_host, n_crops, res, mean3, std3, views, scratch, views_scratch_bytes(H, n_views, res), st, xo);
    if (rc != RLCF_OK) return rc;
    rc = augmix_stage(xo, n_crops, res, mean3, std3, ops_host, w_host, m_host, views, p, st);
    if (rc != RLCF_OK) return rc;
    return RLCF_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// hard_aug (TPT/data/datautils.py:77-87 get_preaugment(hard_aug=True); --hard_aug 1 of tune_cls_tpt.py:115 / tune_cls_kd.py:122): between
// the resized crop and the flip the recipe applies, each with its own coin, ColorJitter(0.4, 0.4, 0.2, 0.1), RandomGrayscale and
// GaussianBlur(3).  On PIL images torchvision routes the first two to Pillow and the blur to its tensor kernel; reproduced here on the
// 8-bit views bit for bit (the flip, already applied by the resampling kernel, commutes with all of it):
//   brightness / contrast / saturation = Image.blend(degenerate, image, factor), libImaging/Blend.c: (UINT8)(in1 + alpha * (in2 - in1))
//     in C float arithmetic, clipped when alpha leaves [0, 1]; degenerate = black / the solid grey int(mean(L) + 0.5) / the pixel's L,
//     L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
//   hue = RGB -> HSV (Convert.c rgb2hsv_row) -> h += shift (mod 256) -> RGB (hsv2rgb), float / double mix as in the C source;
//   grayscale = L in all three bands;  blur = nine fused multiply-adds of the float32 3x3 kernel in row-major order over the
//     reflect-padded view, round half to even (torchvision F_t.gaussian_blur on the uint8 tensor).
// One workgroup per view: the contrast step needs the mean luminance of the WHOLE view as it is when ColorJitter reaches that step,
// so the workgroup first sums L over the view with the preceding steps applied, then rewrites the view in place.
__device__ __forceinline__ int hard_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
__device__ __forceinline__ int hard_blend(int in1, int in2, float alpha) {
#pragma clang fp contract(off)
    const float t = (float)in1 + alpha * (float)(in2 - in1);
    if (alpha >= 0.f && alpha <= 1.0f) return (int)t;
    return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}
__device__ void hard_hue(int& r, int& g, int& b, int shift) {
#pragma clang fp contract(off)
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
        const float cr = (float)(maxc - minc);
        const float s = cr / (float)maxc;
        const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
        float h;
        if (r == maxc) h = bc - gc;
        else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
        else h = (float)(4.0 + (double)gc - (double)rc);
        h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
        uh = min(max((int)((double)h * 255.0), 0), 255);
        us = min(max((int)((double)s * 255.0), 0), 255);
    }
    uh = (uh + shift) & 255;
    if (us == 0) { r = g = b = uv; return; }
    const double h6 = (double)(float)uh * 6.0 / 255.0;
    const int i = (int)floor(h6);
    const float f = (float)(h6 - (double)i);
    const float fs = (float)((double)(float)us / 255.0);
    const double vd = (double)uv;
    const int p = min(max((int)floor(vd * (1.0 - (double)fs) + 0.5), 0), 255);
    const int q = min(max((int)floor(vd * (1.0 - (double)(fs * f)) + 0.5), 0), 255);
    const int t = min(max((int)floor(vd * (1.0 - (double)(fs * (1.0f - f))) + 0.5), 0), 255);
    switch (i % 6) {
        case 0: r = uv; g = t; b = p; break;
        case 1: r = q; g = uv; b = p; break;
        case 2: r = p; g = uv; b = t; break;
        case 3: r = p; g = q; b = uv; break;
        case 4: r = t; g = p; b = uv; break;
        default: r = uv; g = p; b = q; break;
    }
}
__device__ __forceinline__ void hard_step(int fn, const rlcf_hard_aug& pl, int mean, int& r, int& g, int& b) {
    if (fn == 0) { r = hard_blend(0, r, pl.b); g = hard_blend(0, g, pl.b); b = hard_blend(0, b, pl.b); }
    else if (fn == 1) { r = hard_blend(mean, r, pl.c); g = hard_blend(mean, g, pl.c); b = hard_blend(mean, b, pl.c); }
    else if (fn == 2) { const int l = hard_l(r, g, b); r = hard_blend(l, r, pl.s); g = hard_blend(l, g, pl.s); b = hard_blend(l, b, pl.s); }
    else hard_hue(r, g, b, pl.hue);
}
__global__ __launch_bounds__(1024) void hard_pixel_kernel(uint8_t* __restrict__ xo, const rlcf_hard_aug* __restrict__ plans, int res) {
    const int v = blockIdx.x, npix = res * res;
    const rlcf_hard_aug pl = plans[v];
    uint8_t* x = xo + (size_t)(1 + v) * npix * 3;
    const bool jitter = pl.order[0] >= 0;
    if (!jitter && !pl.gray) return;
    __shared__ unsigned long long part[16];
    __shared__ int mean_s;
    int mean = 0;
    if (jitter) {
        int pc = 0;
        while (pc < 4 && pl.order[pc] != 1) ++pc;                         // the steps before the contrast step
        unsigned long long sum = 0;
        for (int p = threadIdx.x; p < npix; p += 1024) {
            int r = x[p * 3], g = x[p * 3 + 1], b = x[p * 3 + 2];
            for (int k = 0; k < pc; ++k) hard_step(pl.order[k], pl, 0, r, g, b);
            sum += (unsigned)hard_l(r, g, b);
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long tot = 0;
            for (int i = 0; i < 16; ++i) tot += part[i];
            mean_s = (int)((double)tot / (double)npix + 0.5);             // int(ImageStat.Stat(L).mean[0] + 0.5)
        }
        __syncthreads();
        mean = mean_s;
    }
    for (int p = threadIdx.x; p < npix; p += 1024) {
        int r = x[p * 3], g = x[p * 3 + 1], b = x[p * 3 + 2];
        if (jitter)
            for (int k = 0; k < 4; ++k) hard_step(pl.order[k], pl, mean, r, g, b);
        if (pl.gray) { const int l = hard_l(r, g, b); r = g = b = l; }
        x[p * 3] = (uint8_t)r; x[p * 3 + 1] = (uint8_t)g; x[p * 3 + 2] = (uint8_t)b;
    }
}
// xh[1 + v] = blur(xo[1 + v]) (or a copy); with `views`: also the normalised float view (no AugMix stage follows)
__global__ __launch_bounds__(256) void hard_blur_kernel(const uint8_t* __restrict__ xo, const rlcf_hard_aug* __restrict__ plans, int res,
                                                        uint8_t* __restrict__ xh, float m0, float m1, float m2, float d0, float d1, float d2,
                                                        float* __restrict__ views) {
    const int v = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= res * res) return;
    const size_t plane = (size_t)res * res;
    const uint8_t* x = xo + (size_t)(1 + v) * plane * 3;
    int out[3];
    if (plans[v].blur) {
        const int y = p / res, xx = p - y * res;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int yy = y + i - 1;
            yy = yy < 0 ? -yy : (yy >= res ? 2 * res - 2 - yy : yy);     // reflect (no edge repeat)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int xj = xx + j - 1;
                xj = xj < 0 ? -xj : (xj >= res ? 2 * res - 2 - xj : xj);
                const float k = plans[v].k[i * 3 + j];
                const uint8_t* s = x + ((size_t)yy * res + xj) * 3;
                acc[0] = __fmaf_rn(k, (float)s[0], acc[0]); acc[1] = __fmaf_rn(k, (float)s[1], acc[1]); acc[2] = __fmaf_rn(k, (float)s[2], acc[2]);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = min(max((int)rintf(acc[c]), 0), 255);
    } else { out[0] = x[(size_t)p * 3]; out[1] = x[(size_t)p * 3 + 1]; out[2] = x[(size_t)p * 3 + 2]; }
    uint8_t* o = xh + ((size_t)(1 + v) * plane + p) * 3;
    o[0] = (uint8_t)out[0]; o[1] = (uint8_t)out[1]; o[2] = (uint8_t)out[2];
    if (views) {
        float* f = views + (size_t)(1 + v) * 3 * plane + p;
        f[0] = ((float)out[0] / 255.0f - m0) / d0; f[plane] = ((float)out[1] / 255.0f - m1) / d1; f[2 * plane] = ((float)out[2] / 255.0f - m2) / d2;
    }
}

size_t views_hard_scratch_bytes(int H, int n_views, int res) {
    const size_t img = (size_t)res * res * 3;
    return ((views_augmix_scratch_bytes(H, n_views, res) + 255) & ~(size_t)255) + (((size_t)n_views * img + 255) & ~(size_t)255) +
           (((size_t)(n_views - 1) * sizeof(rlcf_hard_aug) + 255) & ~(size_t)255);
}

int launch_make_views_hard(const uint8_t* image, int H, int W, const rlcf_crop* crops_host, int n_crops, int res, const float* mean3,
                           const float* std3, const rlcf_hard_aug* hard_host, const rlcf_augmix_op* ops_host, const float* w_host,
                           const float* m_host, float* views, void* scratch, size_t scratch_bytes, hipStream_t st) {
    RLCF_ARG_CHECK(n_crops > 0 && hard_host && scratch && res > 1 && (!ops_host || (w_host && m_host)));
    const int n_views = 1 + n_crops;
    RLCF_ARG_CHECK(scratch_bytes >= views_hard_scratch_bytes(H, n_views, res));
    for (int i = 0; i < n_crops; ++i) {
        const rlcf_hard_aug& h = hard_host[i];
        bool ok = h.hue >= 0 && h.hue <= 255;
        if (h.order[0] >= 0) {
            int seen = 0;
            for (int k = 0; k < 4; ++k) if (h.order[k] >= 0 && h.order[k] < 4) seen |= 1 << h.order[k];
            ok = ok && seen == 15;
        }
        if (!ok) { rlcf_set_error("make_views_hard: plan %d is not a ColorJitter draw (order must be a permutation of 0..3 or start with -1; hue shift 0..255)", i); return RLCF_ERR_ARG; }
    }
    const size_t img = (size_t)res * res * 3;
    char* p = (char*)scratch + ((views_scratch_bytes(H, n_views, res) + 255) & ~(size_t)255);
    uint8_t* xo = (uint8_t*)p; p += ((size_t)n_views * img + 255) & ~(size_t)255;
    char* aug = p;
    p = (char*)scratch + ((views_augmix_scratch_bytes(H, n_views, res) + 255) & ~(size_t)255);
    uint8_t* xh = (uint8_t*)p; p += ((size_t)n_views * img + 255) & ~(size_t)255;
    rlcf_hard_aug* plans = (rlcf_hard_aug*)p;
    int rc = launch_make_views(image, H, W, crops_host, n_crops, res, mean3, std3, views, scratch, views_scratch_bytes(H, n_views, res), st, xo);
    if (rc != RLCF_OK) return rc;
    { int rcu = views_upload(plans, hard_host, (size_t)n_crops * sizeof(rlcf_hard_aug), st); if (rcu != RLCF_OK) return rcu; }
    hard_pixel_kernel<<<dim3(n_crops), dim3(1024), 0, st>>>(xo, plans, res);
    RLCF_LAUNCH_CHECK();
    hard_blur_kernel<<<dim3((res * res + 255) / 256, n_crops), dim3(256), 0, st>>>(xo, plans, res, xh, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                                                                                  std3[2], ops_host ? nullptr : views);
    RLCF_LAUNCH_CHECK();
    if (ops_host) {
        rc = augmix_stage(xh, n_crops, res, mean3, std3, ops_host, w_host, m_host, views, aug, st);
        if (rc != RLCF_OK) return rc;
    }
    return RLCF_OK;
}
