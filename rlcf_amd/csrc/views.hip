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
                                      float d1, float d2, float* __restrict__ out) {
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
                      const float* std3, float* views, void* scratch, size_t scratch_bytes, hipStream_t st) {
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
    if (n_crops) RLCF_HIP_CHECK(hipMemcpyAsync(crops, crops_host, (size_t)n_crops * sizeof(rlcf_crop), hipMemcpyHostToDevice, st));
    const int nt = n_views * 2 * res;
    views_coeffs_kernel<<<dim3((nt + 127) / 128), dim3(128), 0, st>>>(crops, n_views, H, W, res, nw, nh, off_x, off_y, tab);
    RLCF_LAUNCH_CHECK();
    views_horizontal_kernel<<<dim3((res + 63) / 64, H, n_views), dim3(64), 0, st>>>(image, crops, H, W, res, nw, nh, off_x, off_y, tab, tmp);
    RLCF_LAUNCH_CHECK();
    views_vertical_kernel<<<dim3((res + 63) / 64, res, n_views), dim3(64), 0, st>>>(crops, H, W, res, nw, nh, off_x, off_y, tab, tmp, mean3[0],
                                                                                  mean3[1], mean3[2], std3[0], std3[1], std3[2], views);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
