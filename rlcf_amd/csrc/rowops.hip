// HBM-bound row kernels of the CLIP towers: LayerNorm fwd/bwd, patch gather, embedding
// assembly, L2 normalisation, row gathers.  One wave (64 lanes) per row, shuffle reductions,
// float4 accesses where the width allows.
#include "kernels.h"
#include <algorithm>

#define ROWS_PER_BLOCK 4
// split-f16 pair outputs may be interleaved per 32-column block: column c of a row of `width` sits at row*2*width + (c/32)*64 + c%32
// (hi) and 32 halves later (lo, which is what the caller passes as the lo pointer)
__device__ __forceinline__ size_t pair_off(int row, int c, int width, int il) {
    return il ? (size_t)row * 2 * width + (((c >> 5) << 6) | (c & 31)) : (size_t)row * width + c;
}
#define MAX_PER_LANE 16   // width <= 1024

// ---------------------------------------------------------------- LayerNorm fwd
// TPT/clip/model.py:157-163 (fp32, eps 1e-5, biased variance about the mean)
// ADD (RLCF_PREC_F16 pipeline): the residual add of the block rides here — x <- x + d, d = the f16 output of the preceding out_proj / c_proj
// product (TPT/clip/model.py:187-192: x = x + attention(ln_1(x)); x = x + mlp(ln_2(x))), read as f16, added in f32, x written back; the
// normalised row may overwrite d in place (yh == add16: a row is read completely before any of it is written)
template <bool ADD>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            _Float16* yh,
                                                            _Float16* __restrict__ yl, int rows, int width, int group_rows, int group_stride,
                                                            int ilf, const _Float16* add16 = nullptr, float* xout = nullptr) {
    const int il = ilf & 1;                     // bit 1 of ilf: whole-line stores of the interleaved pair rows (below)
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    if (group_rows > 0) {                // every group of rows (one test sample) has its own (gamma, beta)
        const size_t go = (size_t)(row / group_rows) * group_stride;
        gamma += go; beta += go;
    }
    const float* xr = x + (size_t)row * width;
    float v[MAX_PER_LANE];
    float s = 0.f;
    const bool vec = (width & 255) == 0;
    const int n4 = width >> 8;                  // float4 groups per lane when vec
    if (vec) {
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE / 4; ++j)
            if (j < n4) {
                float4 t = *(const float4*)(xr + j * 256 + lane * 4);
                if constexpr (ADD) {
                    const h16x4 d4 = *(const h16x4*)(add16 + (size_t)row * width + j * 256 + lane * 4);
                    t.x += (float)d4[0]; t.y += (float)d4[1]; t.z += (float)d4[2]; t.w += (float)d4[3];
                    *(float4*)(xout + (size_t)row * width + j * 256 + lane * 4) = t;
                }
                v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
                s += t.x + t.y + t.z + t.w;
            }
    } else {
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            v[j] = c < width ? xr[c] : 0.f;
            if constexpr (ADD) {
                if (c < width) { v[j] += (float)add16[(size_t)row * width + c]; xout[(size_t)row * width + c] = v[j]; }
            }
            s += v[j];
        }
    }
    const float mu = wave_sum(s) / width;
    float q = 0.f;
    if (vec) {
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j)
            if ((j >> 2) < n4) { float d = v[j] - mu; q += d * d; }
    } else {
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) { float d = v[j] - mu; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / width + LN_EPS);
    if (vec) {
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE / 4; ++j)
            if (j < n4) {
                const int c = j * 256 + lane * 4;
                float4 gg = *(const float4*)(gamma + c), bb = *(const float4*)(beta + c), o;
                o.x = (v[4 * j] - mu) * rstd * gg.x + bb.x;
                o.y = (v[4 * j + 1] - mu) * rstd * gg.y + bb.y;
                o.z = (v[4 * j + 2] - mu) * rstd * gg.z + bb.z;
                o.w = (v[4 * j + 3] - mu) * rstd * gg.w + bb.w;
                if (y) *(float4*)(y + (size_t)row * width + c) = o;
                if (yh) {                               // split-f16 pair for the consuming GEMM (gemm_f16x3.hip)
                    h16x4 hh, ll;
                    const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hh[q] = (_Float16)ov[q]; ll[q] = (_Float16)(ov[q] - (float)hh[q]); }
                    if ((ilf & 2) && yl == yh + 32) {
                        // interleaved pair rows: a 128-byte line = [hi of 32 columns | lo of the same 32 columns].  Written as they fall (all hi
                        // halves, then all lo halves) every line is requested twice, half a line each time; the lanes of a line pair swap
                        // with lane ^ 8 what the partner's line needs, and each instruction writes whole lines (round 6: a CU's stores are
                        // paced by line requests, profiles/r6_notes.md section 1b).  Same bits at the same addresses.
                        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                        const u32x2_ hv = __builtin_bit_cast(u32x2_, hh), lv = __builtin_bit_cast(u32x2_, ll);
                        const bool up = (lane & 8) != 0;                          // this lane's columns sit in the odd 32-column block
                        const u32x2_ send = up ? hv : lv;
                        u32x2_ recv;
                        recv[0] = (unsigned)__shfl_xor((int)send[0], 8); recv[1] = (unsigned)__shfl_xor((int)send[1], 8);
                        const u32x2_ even_line = up ? recv : hv, odd_line = up ? lv : recv;
                        _Float16* lp = yh + (size_t)row * 2 * width + (size_t)(c >> 6) * 128 + (lane & 15) * 4;      // the even block's line of this 64-column pair
                        if (ilf & 4) { __builtin_nontemporal_store(even_line, (u32x2_*)lp); __builtin_nontemporal_store(odd_line, (u32x2_*)(lp + 64)); }     // (RLCF_LN_LINEST=2: measurement)
                        else { *(u32x2_*)lp = even_line; *(u32x2_*)(lp + 64) = odd_line; }
                    } else {
                        *(h16x4*)(yh + pair_off(row, c, width, il)) = hh;
                        if (yl) *(h16x4*)(yl + pair_off(row, c, width, il)) = ll;       // (yl null: plain f16 output, RLCF_PREC_F16)
                    }
                }
            }
    } else {
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) {
                float o = (v[j] - mu) * rstd * gamma[c] + beta[c];
                if (y) y[(size_t)row * width + c] = o;
                if (yh) {
                    const _Float16 hh = (_Float16)o;
                    yh[pair_off(row, c, width, il)] = hh;
                    if (yl) yl[pair_off(row, c, width, il)] = (_Float16)(o - (float)hh);
                }
            }
        }
    }
}

int launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int width, hipStream_t st, int group_rows,
                         int group_stride) {
    return launch_layernorm_fwd_split(x, gamma, beta, y, nullptr, nullptr, rows, width, st, group_rows, group_stride, 0);
}
int launch_layernorm_fwd_split(const float* x, const float* gamma, const float* beta, float* y, void* yh, void* yl, int rows, int width,
                               hipStream_t st, int group_rows, int group_stride, int il) {
    RLCF_ARG_CHECK(rows > 0 && width > 0 && width <= 64 * MAX_PER_LANE && group_rows >= 0);
    static int linest = -1;                     // RLCF_LN_LINEST=0: hi halves and lo halves of the interleaved pair rows as two half-line stores (A/B)
    if (linest < 0) { const char* e = getenv("RLCF_LN_LINEST"); linest = e ? atoi(e) : 1; }
    layernorm_fwd_kernel<false><<<dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st>>>(x, gamma, beta, y, (_Float16*)yh,
                                                                                                          (_Float16*)yl, rows, width, group_rows,
                                                                                                          group_rows > 0 ? group_stride : 0,
                                                                                                          (il ? 1 : 0) | (il && linest ? 2 : 0) | (il && linest == 2 ? 4 : 0));
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// x <- x + d16 (f32), yh <- LayerNorm(x) as plain f16 (may be d16 itself): the residual add of the single-pass f16 pipeline
int launch_layernorm_add_fwd(float* x, const void* d16, const float* gamma, const float* beta, void* yh, int rows, int width, hipStream_t st,
                             int group_rows, int group_stride) {
    RLCF_ARG_CHECK(rows > 0 && width > 0 && width <= 64 * MAX_PER_LANE && group_rows >= 0 && x && d16 && yh && width % 4 == 0);
    layernorm_fwd_kernel<true><<<dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st>>>(
        x, gamma, beta, nullptr, (_Float16*)yh, nullptr, rows, width, group_rows, group_rows > 0 ? group_stride : 0, 0, (const _Float16*)d16, x);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// x <- x + d16 alone (after the last block of a tower whose every row is consumed)
__global__ __launch_bounds__(256) void add_f16_kernel(float* __restrict__ x, const _Float16* __restrict__ d, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 t = *(const float4*)(x + i * 4);
        const h16x4 d4 = *(const h16x4*)(d + i * 4);
        t.x += (float)d4[0]; t.y += (float)d4[1]; t.z += (float)d4[2]; t.w += (float)d4[3];
        *(float4*)(x + i * 4) = t;
    }
}
int launch_add_f16(float* x, const void* d16, int64_t n, hipStream_t st) {
    RLCF_ARG_CHECK(n > 0 && n % 4 == 0);
    add_f16_kernel<<<dim3((unsigned)std::min<int64_t>((n / 4 + 255) / 256, 8192)), dim3(256), 0, st>>>(x, (const _Float16*)d16, n / 4);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- LayerNorm bwd
// dx = rstd * (g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)); dgamma += dy*xhat; dbeta += dy
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, const float* __restrict__ dres,
                                                            float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows, int width, int group_rows, int group_stride, int gamma_stride) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    if (group_rows > 0) {                // parameter gradients (and, with gamma_stride, the parameters) kept per group of rows
        const size_t go = (size_t)(row / group_rows) * group_stride;
        if (dgamma) dgamma += go;
        if (dbeta) dbeta += go;
        gamma += (size_t)(row / group_rows) * gamma_stride;
    }
    const float* xr = x + (size_t)row * width;
    const float* dr = dy + (size_t)row * width;
    float v[MAX_PER_LANE], d[MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        v[j] = c < width ? xr[c] : 0.f;
        d[j] = c < width ? dr[c] : 0.f;
        s += v[j];
    }
    const float mu = wave_sum(s) / width;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) { float t = v[j] - mu; q += t * t; }
    }
    const float rstd = rsqrtf(wave_sum(q) / width + LN_EPS);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) {
            float xh = (v[j] - mu) * rstd;
            float gd = d[j] * gamma[c];
            if (dgamma) atomicAdd(dgamma + c, d[j] * xh);
            if (dbeta) atomicAdd(dbeta + c, d[j]);
            v[j] = xh; d[j] = gd;
            s1 += gd; s2 += gd * xh;
        }
    }
    s1 = wave_sum(s1) / width;
    s2 = wave_sum(s2) / width;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) {
            float o = rstd * (d[j] - s1 - v[j] * s2);
            if (dres) o += dres[(size_t)row * width + c];
            dx[(size_t)row * width + c] = o;
        }
    }
}

// Same as layernorm_bwd_kernel but every wave walks rpw (<= LNB_ROWS) consecutive rows and keeps its dgamma/dbeta partial sums
// in registers: one atomicAdd per column per wave instead of one per column per row (LayerNorm-tuning backward).  rpw is chosen
// by the launcher so that a small token matrix (one test image: ~1200 rows) still spreads over enough waves.
#define LNB_ROWS 16
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ dy, const float* __restrict__ dres,
                                                                   float* __restrict__ dx, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int rows, int width, int group_rows,
                                                                   int group_stride, int gamma_stride, int rpw) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * rpw;
    float ag[MAX_PER_LANE], ab[MAX_PER_LANE];
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
    int grp = (group_rows > 0 && row0 < rows) ? row0 / group_rows : 0;
    for (int rr = 0; rr < rpw; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        if (group_rows > 0 && row / group_rows != grp) {          // the walk crosses into the next sample: flush the partial sums
#pragma unroll
            for (int j = 0; j < MAX_PER_LANE; ++j) {
                int c = j * 64 + lane;
                if (c < width) {
                    atomicAdd(dgamma + (size_t)grp * group_stride + c, ag[j]); atomicAdd(dbeta + (size_t)grp * group_stride + c, ab[j]);
                    ag[j] = 0.f; ab[j] = 0.f;
                }
            }
            grp = row / group_rows;
        }
        const float* xr = x + (size_t)row * width;
        const float* dr = dy + (size_t)row * width;
        float v[MAX_PER_LANE], d[MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            v[j] = c < width ? xr[c] : 0.f;
            d[j] = c < width ? dr[c] : 0.f;
            s += v[j];
        }
        const float mu = wave_sum(s) / width;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) { float t = v[j] - mu; q += t * t; }
        }
        const float rstd = rsqrtf(wave_sum(q) / width + LN_EPS);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) {
                float xh = (v[j] - mu) * rstd;
                float gd = d[j] * gamma[(group_rows > 0 ? (size_t)(row / group_rows) * gamma_stride : 0) + c];
                ag[j] += d[j] * xh; ab[j] += d[j];
                v[j] = xh; d[j] = gd;
                s1 += gd; s2 += gd * xh;
            }
        }
        s1 = wave_sum(s1) / width;
        s2 = wave_sum(s2) / width;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) {
                float o = rstd * (d[j] - s1 - v[j] * s2);
                if (dres) o += dres[(size_t)row * width + c];
                dx[(size_t)row * width + c] = o;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width && row0 < rows) {
            atomicAdd(dgamma + (size_t)grp * group_stride + c, ag[j]); atomicAdd(dbeta + (size_t)grp * group_stride + c, ab[j]);
        }
    }
}

// ---- bit-reproducible parameter gradients: per-wave partial sums, parked, then added in a fixed order ------------------------------
// The atomicAdd forms above make dgamma / dbeta depend on the arrival order of ~10^4 partial sums.  Here wave w of group g walks rows
// [g * group_rows + k * rpw, ... + rpw) (k = w % wpg: a walk never leaves its group) and parks its two column sums in part[w][2][width];
// colparts_reduce_kernel then adds the wpg partials of a group in order k = 0, 1, ... (16 interleaved sub-sums met in a fixed order).
__global__ __launch_bounds__(256) void layernorm_bwd_parts_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ dy, const float* __restrict__ dres,
                                                                  float* __restrict__ dx, float* __restrict__ part, int rows, int width,
                                                                  int group_rows, int gamma_stride, int rpw, int wpg, int n_groups) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int g = w / wpg, k = w - g * wpg;
    if (g >= n_groups) return;
    const int gend = min(rows, (g + 1) * group_rows), row0 = g * group_rows + k * rpw, row1 = min(gend, row0 + rpw);
    const float* gm = gamma + (size_t)g * gamma_stride;
    float ag[MAX_PER_LANE], ab[MAX_PER_LANE];
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
    for (int row = row0; row < row1; ++row) {
        const float* xr = x + (size_t)row * width;
        const float* dr = dy + (size_t)row * width;
        float v[MAX_PER_LANE], d[MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            v[j] = c < width ? xr[c] : 0.f;
            d[j] = c < width ? dr[c] : 0.f;
            s += v[j];
        }
        const float mu = wave_sum(s) / width;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) { float t = v[j] - mu; q += t * t; }
        }
        const float rstd = rsqrtf(wave_sum(q) / width + LN_EPS);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) {
                float xh = (v[j] - mu) * rstd;
                float gd = d[j] * gm[c];
                ag[j] += d[j] * xh; ab[j] += d[j];
                v[j] = xh; d[j] = gd;
                s1 += gd; s2 += gd * xh;
            }
        }
        s1 = wave_sum(s1) / width;
        s2 = wave_sum(s2) / width;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) {
                float o = rstd * (d[j] - s1 - v[j] * s2);
                if (dres) o += dres[(size_t)row * width + c];
                dx[(size_t)row * width + c] = o;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) { part[((size_t)w * 2) * width + c] = ag[j]; part[((size_t)w * 2 + 1) * width + c] = ab[j]; }
    }
}
// The same for widths that are multiples of 256 (every CLIP tower: 512 / 768 / 1024 / 1280): 16-byte accesses (lane l owns columns
// 256 g + 4 l .. + 3 of group g) and the next row's x / dy fetched while this row's four wave reductions run — the scalar-access form
// above streams at 1.7 TB/s (r3_config2_kernel_stats.txt), bound by the dependent reduction chain of one row at a time.
template <int NG>
__global__ __launch_bounds__(256) void layernorm_bwd_parts4_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ dy, const float* __restrict__ dres,
                                                                   float* __restrict__ dx, float* __restrict__ part, int rows, int group_rows,
                                                                   int gamma_stride, int rpw, int wpg, int n_groups) {
    constexpr int width = NG * 256;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int g = w / wpg, k = w - g * wpg;
    if (g >= n_groups) return;
    const int gend = min(rows, (g + 1) * group_rows), row0 = g * group_rows + k * rpw, row1 = min(gend, row0 + rpw);
    float4 gm[NG], ag[NG], ab[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        gm[j] = *(const float4*)(gamma + (size_t)g * gamma_stride + j * 256 + lane * 4);
        ag[j] = make_float4(0.f, 0.f, 0.f, 0.f); ab[j] = ag[j];
    }
    float4 v[NG], d[NG], vn[NG], dn[NG];
    if (row0 < row1) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            vn[j] = *(const float4*)(x + (size_t)row0 * width + j * 256 + lane * 4);
            dn[j] = *(const float4*)(dy + (size_t)row0 * width + j * 256 + lane * 4);
        }
    }
    for (int row = row0; row < row1; ++row) {
#pragma unroll
        for (int j = 0; j < NG; ++j) { v[j] = vn[j]; d[j] = dn[j]; }
        if (row + 1 < row1) {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                vn[j] = *(const float4*)(x + (size_t)(row + 1) * width + j * 256 + lane * 4);
                dn[j] = *(const float4*)(dy + (size_t)(row + 1) * width + j * 256 + lane * 4);
            }
        }
        float4 rz[NG];
        if (dres) {
#pragma unroll
            for (int j = 0; j < NG; ++j) rz[j] = *(const float4*)(dres + (size_t)row * width + j * 256 + lane * 4);
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NG; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float mu = wave_sum(s) / width;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rstd = rsqrtf(wave_sum(q) / width + LN_EPS);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            v[j].x *= rstd; v[j].y *= rstd; v[j].z *= rstd; v[j].w *= rstd;                 // x hat
            ag[j].x += d[j].x * v[j].x; ag[j].y += d[j].y * v[j].y; ag[j].z += d[j].z * v[j].z; ag[j].w += d[j].w * v[j].w;
            ab[j].x += d[j].x; ab[j].y += d[j].y; ab[j].z += d[j].z; ab[j].w += d[j].w;
            d[j].x *= gm[j].x; d[j].y *= gm[j].y; d[j].z *= gm[j].z; d[j].w *= gm[j].w;     // dy gamma
            s1 += (d[j].x + d[j].y) + (d[j].z + d[j].w);
            s2 += (d[j].x * v[j].x + d[j].y * v[j].y) + (d[j].z * v[j].z + d[j].w * v[j].w);
        }
        s1 = wave_sum(s1) / width;
        s2 = wave_sum(s2) / width;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            float4 o = make_float4(rstd * (d[j].x - s1 - v[j].x * s2), rstd * (d[j].y - s1 - v[j].y * s2), rstd * (d[j].z - s1 - v[j].z * s2),
                                   rstd * (d[j].w - s1 - v[j].w * s2));
            if (dres) { o.x += rz[j].x; o.y += rz[j].y; o.z += rz[j].z; o.w += rz[j].w; }
            *(float4*)(dx + (size_t)row * width + j * 256 + lane * 4) = o;
        }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        *(float4*)(part + ((size_t)w * 2) * width + j * 256 + lane * 4) = ag[j];
        *(float4*)(part + ((size_t)w * 2 + 1) * width + j * 256 + lane * 4) = ab[j];
    }
}
// out_a[g * stride + c] += sum_k part[g * wpg + k][0][c], out_b likewise from [1]: block = 64 columns x 16 sub-sums (k = q, q + 16, ...)
__global__ __launch_bounds__(1024) void colparts_reduce_kernel(const float* __restrict__ part, int wpg, int width, float* __restrict__ out_a,
                                                              float* __restrict__ out_b, int stride) {
    __shared__ float r1[16][64], r2[16][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + l, g = blockIdx.y;
    float a = 0.f, b = 0.f;
    if (c < width)
        for (int k = q; k < wpg; k += 16) {
            const size_t w = (size_t)g * wpg + k;
            a += part[(w * 2) * width + c]; b += part[(w * 2 + 1) * width + c];
        }
    r1[q][l] = a; r2[q][l] = b;
    __syncthreads();
    if (q != 0 || c >= width) return;
    a = 0.f; b = 0.f;
    for (int k = 0; k < 16; ++k) { a += r1[k][l]; b += r2[k][l]; }
    out_a[(size_t)g * stride + c] += a;
    if (out_b) out_b[(size_t)g * stride + c] += b;
}
// walk plan of the partial-sum kernels: ~4096 waves over all groups, at least 2 rows per wave
static void parts_plan(int rows, int group_rows, int& rpw, int& wpg, int& n_groups) {
    n_groups = (rows + group_rows - 1) / group_rows;
    const int wpg_target = std::max(1, 4096 / n_groups);
    rpw = std::max(2, (group_rows + wpg_target - 1) / wpg_target);
    wpg = (group_rows + rpw - 1) / rpw;
}

int launch_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* dres, float* dx, float* dgamma,
                         float* dbeta, int rows, int width, hipStream_t st, int group_rows, int group_stride, int gamma_stride,
                         float* part_ws, size_t part_ws_floats) {
    RLCF_ARG_CHECK(rows > 0 && width > 0 && width <= 64 * MAX_PER_LANE && group_rows >= 0);
    if (group_rows == 0) { group_stride = 0; gamma_stride = 0; }
    if (dgamma && dbeta && part_ws) {                         // bit-reproducible form (the engine's backward passes)
        int rpw, wpg, ng;
        parts_plan(rows, group_rows > 0 ? group_rows : rows, rpw, wpg, ng);
        const size_t waves = (size_t)wpg * ng;
        if (waves * 2 * width <= part_ws_floats) {
            const dim3 grid((unsigned)((waves + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
            const int gr = group_rows > 0 ? group_rows : rows;
            const bool al16 = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)(dres ? dres : x)) & 15) == 0 && gamma_stride % 4 == 0;
#define LNB4(NG_) layernorm_bwd_parts4_kernel<NG_><<<grid, dim3(256), 0, st>>>(x, gamma, dy, dres, dx, part_ws, rows, gr, gamma_stride, rpw, wpg, ng)
            if (al16 && width == 512) LNB4(2);
            else if (al16 && width == 768) LNB4(3);
            else if (al16 && width == 1024) LNB4(4);
            else if (al16 && width == 1280) LNB4(5);
            else layernorm_bwd_parts_kernel<<<grid, dim3(256), 0, st>>>(x, gamma, dy, dres, dx, part_ws, rows, width, gr, gamma_stride, rpw, wpg, ng);
#undef LNB4
            RLCF_LAUNCH_CHECK();
            colparts_reduce_kernel<<<dim3((width + 63) / 64, ng), dim3(1024), 0, st>>>(part_ws, wpg, width, dgamma, dbeta, group_stride);
            RLCF_LAUNCH_CHECK();
            return RLCF_OK;
        }
    }
    if (dgamma && dbeta && rows >= 256) {
        const int rpw = rows >= 16384 ? LNB_ROWS : rows >= 4096 ? 4 : 2;          // >= 512 waves whenever the matrix allows it
        const int per_block = ROWS_PER_BLOCK * rpw;
        layernorm_bwd_params_kernel<<<dim3((rows + per_block - 1) / per_block), dim3(256), 0, st>>>(x, gamma, dy, dres, dx, dgamma, dbeta, rows, width, group_rows, group_stride, gamma_stride, rpw);
        RLCF_LAUNCH_CHECK();
        return RLCF_OK;
    }
    layernorm_bwd_kernel<<<dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st>>>(x, gamma, dy, dres, dx, dgamma, dbeta, rows, width, group_rows, group_stride, gamma_stride);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- patch gather (im2col)
// stride==kernel convolution of TPT/clip/model.py:211,224 written as gather + GEMM.
__global__ void im2col_kernel(const float* __restrict__ img, float* __restrict__ out, _Float16* __restrict__ oh, _Float16* __restrict__ ol,
                              int n, int R, int ps, int Kp, int il) {
    const int G = R / ps;
    const int K = 3 * ps * ps;
    const int kq = Kp >> 2;                               // float4 groups per patch row
    const long total = (long)n * G * G * kq;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(idx % kq) * 4;
        const long p = idx / kq;
        const int gx = (int)(p % G), gy = (int)((p / G) % G), b = (int)(p / ((long)G * G));
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k4 + e;
            if (k < K) {
                const int c = k / (ps * ps), i = (k / ps) % ps, j = k % ps;
                o[e] = img[(((size_t)b * 3 + c) * R + gy * ps + i) * R + gx * ps + j];
            } else o[e] = 0.f;
        }
        if (out) *(float4*)(out + (size_t)p * Kp + k4) = make_float4(o[0], o[1], o[2], o[3]);
        if (oh) {
            h16x4 hh, ll;
#pragma unroll
            for (int q = 0; q < 4; ++q) { hh[q] = (_Float16)o[q]; ll[q] = (_Float16)(o[q] - (float)hh[q]); }
            *(h16x4*)(oh + pair_off((int)p, k4, Kp, il)) = hh;
            if (ol) *(h16x4*)(ol + pair_off((int)p, k4, Kp, il)) = ll;
        }
    }
}
int launch_im2col(const float* images, float* out, void* out_hi, void* out_lo, int n, int R, int ps, int Kp, hipStream_t st, int il) {
    RLCF_ARG_CHECK(n > 0 && R % ps == 0 && Kp % 4 == 0 && Kp >= 3 * ps * ps);
    const long total = (long)n * (R / ps) * (R / ps) * (Kp / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    im2col_kernel<<<dim3(blocks), dim3(256), 0, st>>>(images, out, (_Float16*)out_hi, (_Float16*)out_lo, n, R, ps, Kp, il);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- ViT embed assemble + ln_pre
// TPT/clip/model.py:227-229: x = ln_pre(cat([cls, patches]) + pos)
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ x,
                                                           int n, int tokens, int width, int group_imgs, int group_stride) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= n * tokens) return;
    const int lane = threadIdx.x & 63;
    const int b = row / tokens, tok = row % tokens;
    if (group_imgs > 0) { gamma += (size_t)(b / group_imgs) * group_stride; beta += (size_t)(b / group_imgs) * group_stride; }
    const float* src = tok == 0 ? cls : patch_out + ((size_t)b * (tokens - 1) + tok - 1) * width;
    const float* pr = pos + (size_t)tok * width;
    float v[MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        v[j] = c < width ? src[c] + pr[c] : 0.f;
        s += v[j];
    }
    const float mu = wave_sum(s) / width;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) { float d = v[j] - mu; q += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(q) / width + LN_EPS);
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) x[(size_t)row * width + c] = (v[j] - mu) * rstd * gamma[c] + beta[c];
    }
}
int launch_vit_assemble(const float* patch_out, const float* cls, const float* pos, const float* gamma, const float* beta,
                        float* x, int n, int tokens, int width, hipStream_t st, int group_imgs, int group_stride) {
    RLCF_ARG_CHECK(width <= 64 * MAX_PER_LANE);
    const int rows = n * tokens;
    vit_assemble_kernel<<<dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st>>>(patch_out, cls, pos, gamma, beta, x, n, tokens, width, group_imgs, group_stride);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- text assemble
// PromptLearner.forward + positional add (TPT/clip/custom_clip.py:198-238,63): E already holds
// token embedding (non-ctx rows) + positional embedding; ctx rows add the learnable vectors.
// row_src (optional): X row r is built from row row_src[r] of E / ctx_row (-1: zero row) — the
// sparse-backward layout re-packs a few class prompts this way.
// rep_rows > 0: rows come in groups of rep_rows, one per test sample, and group b reads its own context block
// ctx + b * ctx_stride4 (one prompt per sample).  Without row_src the layout itself is replicated (row r is layout row
// r % rep_rows); a row_src table is indexed by the global row (every group has its own sampled classes).
__global__ void text_assemble_kernel(const float* __restrict__ E, const int32_t* __restrict__ row_src, const int32_t* __restrict__ ctx_row,
                                     const float* __restrict__ ctx, float* __restrict__ X, int rows, int w4, int rep_rows, int ctx_stride4) {
    const long total = (long)rows * w4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / w4), c = (int)(idx % w4);
        const int rep = rep_rows > 0 ? r / rep_rows : 0;
        const int rl = rep_rows > 0 ? r - rep * rep_rows : r;
        const int sr = row_src ? row_src[r] : rl;
        if (sr < 0) { ((float4*)X)[idx] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
        float4 e = ((const float4*)E)[(size_t)sr * w4 + c];
        const int cr = ctx_row[sr];
        if (cr >= 0) {
            float4 t = ((const float4*)ctx)[(size_t)rep * ctx_stride4 + (size_t)cr * w4 + c];
            e.x += t.x; e.y += t.y; e.z += t.z; e.w += t.w;
        }
        ((float4*)X)[idx] = e;
    }
}
int launch_text_assemble(const float* E, const int32_t* row_src, const int32_t* ctx_row, const float* ctx, float* X, int rows,
                         int width, int rep_rows, int ctx_stride, hipStream_t st) {
    RLCF_ARG_CHECK(width % 4 == 0 && rows > 0);
    const long total = (long)rows * (width / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    text_assemble_kernel<<<dim3(blocks), dim3(256), 0, st>>>(E, row_src, ctx_row, ctx, X, rows, width / 4, rep_rows, ctx_stride / 4);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- row gather / scatter
// grid (rows, column chunks): a row of a few hundred floats is one block, the 150 528-float rows of the selected views' images (6 rows:
// one block per row took 250 us) are cut into 4096-float chunks copied with 16-byte accesses
__global__ void gather_rows_kernel(const float* __restrict__ in, int ld_in, const int32_t* __restrict__ idx, float* __restrict__ out,
                                   int ld_out, int rows, int width, int vec4) {
    const int r = blockIdx.x;
    const int src = idx ? idx[r] : r;
    const int c0 = blockIdx.y * 4096, c1 = min(width, c0 + 4096);
    const float* ip = in + (size_t)src * ld_in;
    float* op = out + (size_t)r * ld_out;
    if (vec4) {
        for (int c = c0 + threadIdx.x * 4; c < c1; c += blockDim.x * 4) *(float4*)(op + c) = *(const float4*)(ip + c);
    } else {
        for (int c = c0 + threadIdx.x; c < c1; c += blockDim.x) op[c] = ip[c];
    }
}
int launch_gather_rows(const float* in, int ld_in, const int32_t* idx, float* out, int ld_out, int rows, int width, hipStream_t st) {
    RLCF_ARG_CHECK(rows > 0 && width > 0);
    const int vec4 = width % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0;
    gather_rows_kernel<<<dim3(rows, (width + 4095) / 4096), dim3(256), 0, st>>>(in, ld_in, idx, out, ld_out, rows, width, vec4);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows_idx, float* __restrict__ dst, int width) {
    const int r = blockIdx.x;
    const int d = rows_idx[r];
    for (int c = threadIdx.x; c < width; c += blockDim.x) dst[(size_t)d * width + c] = src[(size_t)r * width + c];
}
int launch_scatter_rows(const float* src, const int32_t* rows_idx, float* dst, int n, int width, hipStream_t st) {
    RLCF_ARG_CHECK(n > 0);
    scatter_rows_kernel<<<dim3(n), dim3(256), 0, st>>>(src, rows_idx, dst, width);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- L2 normalise (custom_clip.py:320,330)
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ inv_norm,
                                                     int rows, int width) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < width; c += 64) { float v = in[(size_t)row * width + c]; s += v * v; }
    const float nrm = sqrtf(wave_sum(s));
    for (int c = lane; c < width; c += 64) out[(size_t)row * width + c] = in[(size_t)row * width + c] / nrm;
    if (inv_norm && lane == 0) inv_norm[row] = 1.0f / nrm;
}
int launch_l2norm_rows(const float* in, float* out, float* inv_norm, int rows, int width, hipStream_t st) {
    RLCF_ARG_CHECK(rows > 0);
    l2norm_kernel<<<dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st>>>(in, out, inv_norm, rows, width);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ t, const float* __restrict__ dt,
                                                         const float* __restrict__ inv_norm, float* __restrict__ du, int rows, int width) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < width; c += 64) s += t[(size_t)row * width + c] * dt[(size_t)row * width + c];
    s = wave_sum(s);
    const float inv = inv_norm[row];
    for (int c = lane; c < width; c += 64)
        du[(size_t)row * width + c] = (dt[(size_t)row * width + c] - t[(size_t)row * width + c] * s) * inv;
}
int launch_l2norm_bwd(const float* t, const float* dt, const float* inv_norm, float* du, int rows, int width, hipStream_t st) {
    RLCF_ARG_CHECK(rows > 0);
    l2norm_bwd_kernel<<<dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st>>>(t, dt, inv_norm, du, rows, width);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- ctx gradient (custom_clip.py:198-238 backward)
__global__ void ctx_grad_kernel(const float* __restrict__ dX, const int32_t* __restrict__ ctx_rows, int n_copies, int n_ctx,
                                int width, int group_rows, float* __restrict__ dctx) {
    const int j = blockIdx.y, b = blockIdx.z;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    float s = 0.f;
    for (int k = 0; k < n_copies; ++k) s += dX[((size_t)b * group_rows + ctx_rows[k * n_ctx + j]) * width + c];
    dctx[((size_t)b * n_ctx + j) * width + c] = s;
}
int launch_ctx_grad(const float* dX, const int32_t* ctx_rows, int n_copies, int n_ctx, int width, float* dctx, hipStream_t st) {
    return launch_ctx_grad_grouped(dX, ctx_rows, n_copies, n_ctx, width, 1, 0, dctx, st);
}
// dctx[b] from group b's rows: ctx_rows are group-relative, group b starts at row b*group_rows
int launch_ctx_grad_grouped(const float* dX, const int32_t* ctx_rows, int n_copies, int n_ctx, int width, int groups, int group_rows,
                            float* dctx, hipStream_t st) {
    RLCF_ARG_CHECK(n_copies > 0 && n_ctx > 0 && groups > 0);
    ctx_grad_kernel<<<dim3((width + 63) / 64, n_ctx, groups), dim3(64), 0, st>>>(dX, ctx_rows, n_copies, n_ctx, width, group_rows, dctx);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// learnable rows at class-dependent positions ('front' / 'middle' class-token position, custom_clip.py:239-284): no fixed row lists —
// every row of the group is looked up in ctx_row (through row_src for re-packed layouts); fixed summation order, deterministic
__global__ void ctx_grad_scan_kernel(const float* __restrict__ dX, const int32_t* __restrict__ row_src, const int32_t* __restrict__ ctx_row,
                                     int group_rows, int n_ctx, int width, float* __restrict__ dctx) {
    const int j = blockIdx.y, b = blockIdx.z;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    const size_t base = (size_t)b * group_rows;
    float s = 0.f;
    for (int r = 0; r < group_rows; ++r) {
        const int src = row_src ? row_src[base + r] : r;
        if (src >= 0 && ctx_row[src] == j) s += dX[(base + r) * width + c];
    }
    dctx[((size_t)b * n_ctx + j) * width + c] = s;
}
int launch_ctx_grad_scan(const float* dX, const int32_t* row_src, const int32_t* ctx_row, int groups, int group_rows, int n_ctx, int width,
                         float* dctx, hipStream_t st) {
    RLCF_ARG_CHECK(dX && ctx_row && dctx && groups > 0 && group_rows > 0 && n_ctx > 0);
    ctx_grad_scan_kernel<<<dim3((width + 63) / 64, n_ctx, groups), dim3(64), 0, st>>>(dX, row_src, ctx_row, group_rows, n_ctx, width, dctx);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- d txt from d logits (dense)
__global__ void dtxt_dense_kernel(const float* __restrict__ dlogits, const float* __restrict__ img, int n, int C, int D, float scale,
                                  float* __restrict__ dtxt) {
    const int c = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += dlogits[(size_t)i * C + c] * img[(size_t)i * D + d];
        dtxt[(size_t)c * D + d] = scale * s;
    }
}
int launch_dtxt_dense(const float* dlogits, const float* img, int n, int C, int D, float scale, float* dtxt, hipStream_t st) {
    RLCF_ARG_CHECK(n > 0 && C > 0);
    dtxt_dense_kernel<<<dim3(C), dim3(256), 0, st>>>(dlogits, img, n, C, D, scale, dtxt);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- conversions
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        int r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        int c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[threadIdx.x][j];
    }
}
int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t st) {
    transpose_kernel<<<dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, st>>>(in, out, rows, cols);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// out[c, 0..ld_out) = in[0..rows, c] followed by zeros (ld_out >= rows): the K-major operands of a weight-gradient GEMM
// dW = dY^T X, with the token dimension padded to the GEMM's K granule
__global__ void transpose_pad_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int rows, int cols, int ld_out) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        int c = bx + j, r = by + threadIdx.x;
        if (r < ld_out && c < cols) out[(size_t)c * ld_out + r] = tile[threadIdx.x][j];
    }
}
int launch_transpose_pad(const float* in, int ld_in, float* out, int rows, int cols, int ld_out, hipStream_t st) {
    RLCF_ARG_CHECK(in && out && rows > 0 && cols > 0 && ld_out >= rows && ld_in >= cols);
    transpose_pad_kernel<<<dim3((cols + 31) / 32, (ld_out + 31) / 32), dim3(32, 8), 0, st>>>(in, ld_in, out, rows, cols, ld_out);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// out[c] += sum_r in[r, c]  (bias gradients; out pre-zeroed by the caller): 64 columns x 4 row lanes per block, COLSUM_ROWS rows per block
#define COLSUM_ROWS 64
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int ld, int rows, int cols, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int r0 = blockIdx.y * COLSUM_ROWS, r1 = min(rows, r0 + COLSUM_ROWS);
    float s = 0.f;
    if (c < cols) for (int r = r0 + q; r < r1; r += 4) s += in[(size_t)r * ld + c];
    part[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < cols) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}
// ... bit-reproducible: the row blocks park their sums in part[by][cols], colparts_reduce_kernel adds them in order
__global__ __launch_bounds__(256) void colsum_parts_kernel(const float* __restrict__ in, int ld, int rows, int cols, int rows_per_block,
                                                           float* __restrict__ part) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (c < cols) for (int r = r0 + q; r < r1; r += 4) s += in[(size_t)r * ld + c];
    sm[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < cols) {
        const int l = threadIdx.x;
        part[((size_t)blockIdx.y * 2) * cols + c] = (sm[0][l] + sm[1][l]) + (sm[2][l] + sm[3][l]);
    }
}
int launch_colsum(const float* in, int ld, int rows, int cols, float* out, hipStream_t st, float* part_ws, size_t part_ws_floats) {
    RLCF_ARG_CHECK(in && out && rows > 0 && cols > 0 && ld >= cols);
    if (part_ws) {
        const int rpb = std::max(COLSUM_ROWS, (rows + 1023) / 1024), nb = (rows + rpb - 1) / rpb;
        if ((size_t)nb * 2 * cols <= part_ws_floats) {
            colsum_parts_kernel<<<dim3((cols + 63) / 64, nb), dim3(256), 0, st>>>(in, ld, rows, cols, rpb, part_ws);
            RLCF_LAUNCH_CHECK();
            colparts_reduce_kernel<<<dim3((cols + 63) / 64, 1), dim3(1024), 0, st>>>(part_ws, nb, cols, out, nullptr, 0);
            RLCF_LAUNCH_CHECK();
            return RLCF_OK;
        }
    }
    colsum_kernel<<<dim3((cols + 63) / 64, (rows + COLSUM_ROWS - 1) / COLSUM_ROWS), dim3(256), 0, st>>>(in, ld, rows, cols, out);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// input of ln_pre: pre[b, t, :] = (t == 0 ? class_embedding : patch_out[b, t-1, :]) + positional_embedding[t, :]  (model.py:225-229)
__global__ void vit_preln_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls, const float* __restrict__ pos,
                                 float* __restrict__ pre, long total, int tokens, int width) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % width);
        const long row = i / width;
        const int tok = (int)(row % tokens);
        const long b = row / tokens;
        const float v = tok == 0 ? cls[c] : patch_out[(b * (tokens - 1) + tok - 1) * width + c];
        pre[i] = v + pos[(size_t)tok * width + c];
    }
}
int launch_vit_preln(const float* patch_out, const float* cls, const float* pos, float* pre, int n, int tokens, int width, hipStream_t st) {
    RLCF_ARG_CHECK(patch_out && cls && pos && pre && n > 0);
    const long total = (long)n * tokens * width;
    vit_preln_kernel<<<dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st>>>(patch_out, cls, pos, pre, total, tokens, width);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- QuickGELU elementwise (model.py:166-168)
__global__ void quickgelu_kernel(const float* __restrict__ f, float* __restrict__ g, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = ((const float4*)f)[i];
        v.x = quick_gelu(v.x); v.y = quick_gelu(v.y); v.z = quick_gelu(v.z); v.w = quick_gelu(v.w);
        ((float4*)g)[i] = v;
    }
}
int launch_quickgelu(const float* f, float* g, int64_t n, hipStream_t st) {
    RLCF_ARG_CHECK(n % 4 == 0);
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    quickgelu_kernel<<<dim3(blocks), dim3(256), 0, st>>>(f, g, n / 4);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- sparse-backward layout
// entry e re-packs class cls[e] (device) at rows [pre_rows + e*lmax, +class_len): builds the
// sequence descriptors, EOT rows and the row_src indirection consumed by text_assemble.
__global__ void build_sparse_layout_kernel(const int32_t* __restrict__ cls, int n_e, const int32_t* __restrict__ class_start,
                                           const int32_t* __restrict__ class_len, const int32_t* __restrict__ class_eot_off,
                                           int lmax, int pre_rows, rlcf_seq* __restrict__ seqs, int32_t* __restrict__ eot_rows,
                                           int32_t* __restrict__ row_src) {
    // blockIdx.y = test sample (group): each group owns pre_rows + n_e*lmax rows and its own copy of the prefix
    const int e = blockIdx.x, b = blockIdx.y;
    const int gbase = b * (pre_rows + n_e * lmax);
    const int nseq_g = n_e + (pre_rows > 0 ? 1 : 0);
    if (e == n_e) {                                   // the group's prefix rows (and their own sequence)
        for (int r = threadIdx.x; r < pre_rows; r += blockDim.x) row_src[gbase + r] = r;
        if (threadIdx.x == 0 && pre_rows > 0) { rlcf_seq s = {gbase, pre_rows, 0, 0}; seqs[b * nseq_g + n_e] = s; }
        return;
    }
    const int c = cls[b * n_e + e];
    const int base = gbase + pre_rows + e * lmax, len = class_len[c], st = class_start[c];
    for (int j = threadIdx.x; j < lmax; j += blockDim.x) row_src[base + j] = j < len ? st + j : -1;
    if (threadIdx.x == 0) {
        rlcf_seq s = {base, len, gbase, pre_rows};
        seqs[b * nseq_g + e] = s;
        eot_rows[b * n_e + e] = base + class_eot_off[c];
    }
}
int launch_build_sparse_layout(const int32_t* cls, int groups, int n_e, const int32_t* class_start, const int32_t* class_len,
                               const int32_t* class_eot_off, int lmax, int pre_rows, rlcf_seq* seqs, int32_t* eot_rows,
                               int32_t* row_src, hipStream_t st) {
    RLCF_ARG_CHECK(n_e > 0 && lmax > 0 && groups > 0);
    build_sparse_layout_kernel<<<dim3(n_e + 1, groups), dim3(64), 0, st>>>(cls, n_e, class_start, class_len, class_eot_off, lmax, pre_rows,
                                                                  seqs, eot_rows, row_src);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// dtxt[e,:] = scale * dlogits[e / K, cls[e]] * img[e / K,:]   (one entry per sampled (view, class) pair)
__global__ void dtxt_sparse_kernel(const float* __restrict__ dlogits, const int32_t* __restrict__ cls, const float* __restrict__ img,
                                   int K, int C, int D, float scale, float* __restrict__ dtxt) {
    const int e = blockIdx.x, i = e / K;
    const float g = scale * dlogits[(size_t)i * C + cls[e]];
    // duplicates of one class within a row share one dlogits entry: split it evenly
    int dup = 0;
    for (int k = 0; k < K; ++k) dup += (cls[i * K + k] == cls[e]) ? 1 : 0;
    for (int d = threadIdx.x; d < D; d += blockDim.x) dtxt[(size_t)e * D + d] = g / dup * img[(size_t)i * D + d];
}
int launch_dtxt_sparse(const float* dlogits, const int32_t* cls, const float* img, int n_e, int K, int C, int D, float scale,
                       float* dtxt, hipStream_t st) {
    RLCF_ARG_CHECK(n_e > 0 && K > 0);
    dtxt_sparse_kernel<<<dim3(n_e), dim3(256), 0, st>>>(dlogits, cls, img, K, C, D, scale, dtxt);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- replicated text layout (one replica per test sample)
__global__ void replicate_layout_kernel(const rlcf_seq* __restrict__ seqs, int n_seq, const int32_t* __restrict__ eot_rows, int C, int T,
                                        int B, rlcf_seq* __restrict__ seqs_rep, int32_t* __restrict__ eot_rep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * n_seq) {
        const int b = i / n_seq;
        rlcf_seq s = seqs[i - b * n_seq];
        s.q_start += b * T; s.pre_start += b * T;
        seqs_rep[i] = s;
    }
    if (i < B * C) { const int b = i / C; eot_rep[i] = eot_rows[i - b * C] + b * T; }
}
int launch_replicate_layout(const rlcf_seq* seqs, int n_seq, const int32_t* eot_rows, int C, int T, int B, rlcf_seq* seqs_rep,
                            int32_t* eot_rep, hipStream_t st) {
    const int n = B * (n_seq > C ? n_seq : C);
    replicate_layout_kernel<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(seqs, n_seq, eot_rows, C, T, B, seqs_rep, eot_rep);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// out[b, :] = in[:] for b < B
__global__ void broadcast_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int B) {
    const long total = (long)n * B;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) out[i] = in[i % n];
}
int launch_broadcast_rows(const float* in, float* out, int n, int B, hipStream_t st) {
    const long total = (long)n * B;
    broadcast_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(in, out, n, B);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- ln_pre parameter gradients
// backward of vit_assemble w.r.t. (gamma, beta) of ln_pre: dgamma += dy * xhat, dbeta += dy, xhat recomputed.
__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                               const float* __restrict__ pos, const float* __restrict__ dy,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, int n, int tokens,
                                                               int width, int group_imgs, int group_stride) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= n * tokens) return;
    const int lane = threadIdx.x & 63;
    const int b = row / tokens, tok = row % tokens;
    if (group_imgs > 0) { dgamma += (size_t)(b / group_imgs) * group_stride; dbeta += (size_t)(b / group_imgs) * group_stride; }
    const float* src = tok == 0 ? cls : patch_out + ((size_t)b * (tokens - 1) + tok - 1) * width;
    const float* pr = pos + (size_t)tok * width;
    float v[MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        v[j] = c < width ? src[c] + pr[c] : 0.f;
        s += v[j];
    }
    const float mu = wave_sum(s) / width;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) { float d = v[j] - mu; q += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(q) / width + LN_EPS);
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) {
            const float g = dy[(size_t)row * width + c];
            atomicAdd(dgamma + c, g * (v[j] - mu) * rstd);
            atomicAdd(dbeta + c, g);
        }
    }
}
// ... bit-reproducible form (see layernorm_bwd_parts_kernel): waves walk rpw rows inside their group and park their column sums
__global__ __launch_bounds__(256) void vit_assemble_bwd_parts_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                                     const float* __restrict__ pos, const float* __restrict__ dy,
                                                                     float* __restrict__ part, int rows, int tokens, int width,
                                                                     int group_rows, int rpw, int wpg, int n_groups) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int g = w / wpg, k = w - g * wpg;
    if (g >= n_groups) return;
    const int gend = min(rows, (g + 1) * group_rows), row0 = g * group_rows + k * rpw, row1 = min(gend, row0 + rpw);
    float ag[MAX_PER_LANE], ab[MAX_PER_LANE];
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
    for (int row = row0; row < row1; ++row) {
        const int b = row / tokens, tok = row % tokens;
        const float* src = tok == 0 ? cls : patch_out + ((size_t)b * (tokens - 1) + tok - 1) * width;
        const float* pr = pos + (size_t)tok * width;
        float v[MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            v[j] = c < width ? src[c] + pr[c] : 0.f;
            s += v[j];
        }
        const float mu = wave_sum(s) / width;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) { float d = v[j] - mu; q += d * d; }
        }
        const float rstd = rsqrtf(wave_sum(q) / width + LN_EPS);
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            int c = j * 64 + lane;
            if (c < width) {
                const float gg = dy[(size_t)row * width + c];
                ag[j] += gg * (v[j] - mu) * rstd; ab[j] += gg;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        int c = j * 64 + lane;
        if (c < width) { part[((size_t)w * 2) * width + c] = ag[j]; part[((size_t)w * 2 + 1) * width + c] = ab[j]; }
    }
}
int launch_vit_assemble_bwd(const float* patch_out, const float* cls, const float* pos, const float* dy, float* dgamma, float* dbeta, int n,
                            int tokens, int width, hipStream_t st, int group_imgs, int group_stride, float* part_ws, size_t part_ws_floats) {
    RLCF_ARG_CHECK(width <= 64 * MAX_PER_LANE && n > 0);
    const int rows = n * tokens;
    if (part_ws) {
        int rpw, wpg, ng;
        const int grows = group_imgs > 0 ? group_imgs * tokens : rows;
        parts_plan(rows, grows, rpw, wpg, ng);
        const size_t waves = (size_t)wpg * ng;
        if (waves * 2 * width <= part_ws_floats) {
            vit_assemble_bwd_parts_kernel<<<dim3((unsigned)((waves + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, st>>>(
                patch_out, cls, pos, dy, part_ws, rows, tokens, width, grows, rpw, wpg, ng);
            RLCF_LAUNCH_CHECK();
            colparts_reduce_kernel<<<dim3((width + 63) / 64, ng), dim3(1024), 0, st>>>(part_ws, wpg, width, dgamma, dbeta, group_imgs > 0 ? group_stride : 0);
            RLCF_LAUNCH_CHECK();
            return RLCF_OK;
        }
    }
    vit_assemble_bwd_kernel<<<dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st>>>(patch_out, cls, pos, dy, dgamma, dbeta, n,
                                                                                                   tokens, width, group_imgs, group_stride);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// d img[i,:] = scale * sum_c dlogits[i,c] * txt[c,:]    (backward of logits = scale * img @ txt^T w.r.t. img)
// block = (view i, 64-wide d chunk); its 4 waves take the classes c = q, q + 4, ... and meet in LDS in a fixed order: bit-reproducible
// (the 16-way atomicAdd version this replaces made every backward pass depend on the arrival order of its partial sums)
__global__ __launch_bounds__(256) void dimg_kernel(const float* __restrict__ dlogits, const float* __restrict__ txt, int C, int D, float scale,
                                                   float* __restrict__ dimg) {
    __shared__ float part[4][64];
    const int i = blockIdx.x, l = threadIdx.x & 63, q = threadIdx.x >> 6, d = blockIdx.y * 64 + l;
    float s = 0.f;
    if (d < D)
        for (int c = q; c < C; c += 4) s += dlogits[(size_t)i * C + c] * txt[(size_t)c * D + d];
    part[q][l] = s;
    __syncthreads();
    if (q == 0 && d < D) dimg[(size_t)i * D + d] = scale * ((part[0][l] + part[1][l]) + (part[2][l] + part[3][l]));
}
int launch_dimg(const float* dlogits, const float* txt, int n, int C, int D, float scale, float* dimg, hipStream_t st) {
    RLCF_ARG_CHECK(n > 0 && C > 0 && D > 0);
    dimg_kernel<<<dim3(n, (D + 63) / 64), dim3(256), 0, st>>>(dlogits, txt, C, D, scale, dimg);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- max |x| (weight pre-scaling of the split-f16 GEMMs)
// one atomicMax per block: 4096 atomics on one address cost ~48 us whatever n is (measured), 256 cost ~3 us
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, unsigned int* __restrict__ out) {
    __shared__ float part[4];
    float m = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));   // non-negative floats order like their bit patterns
}
int launch_absmax(const float* x, int64_t n, float* out_dev, hipStream_t st) {
    RLCF_ARG_CHECK(((uintptr_t)x & 15) == 0);
    RLCF_HIP_CHECK(hipMemsetAsync(out_dev, 0, sizeof(float), st));
    int blocks = (int)((n + 4095) / 4096);                 // >= 16 elements per thread
    if (blocks > 512) blocks = 512;
    absmax_kernel<<<dim3(blocks), dim3(256), 0, st>>>(x, n, (unsigned int*)out_dev);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---------------------------------------------------------------- bicubic resample, align_corners=True
// nn.functional.interpolate(images, size=res, mode='bicubic', align_corners=True) of TPT/clip_reward.py:133-134,264-266:
// cubic convolution kernel with A = -0.75, source index = dst * (in-1)/(out-1), taps clamped at the border.
__device__ __forceinline__ void cubic_w(float t, float* w) {
    const float A = -0.75f;
    w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = ((A * (2.f - t) - 5.f * A) * (2.f - t) + 8.f * A) * (2.f - t) - 4.f * A;
}
__global__ void bicubic_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int Ri, int Ro) {
    const long total = (long)planes * Ro * Ro;
    const float sc = Ro > 1 ? (float)(Ri - 1) / (float)(Ro - 1) : 0.f;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Ro), oy = (int)((idx / Ro) % Ro);
        const long pl = idx / ((long)Ro * Ro);
        const float sx = ox * sc, sy = oy * sc;
        const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
        float wx[4], wy[4];
        cubic_w(sx - x0, wx);
        cubic_w(sy - y0, wy);
        const float* p = in + pl * Ri * Ri;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(y0 - 1 + i, 0), Ri - 1);
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) r += wx[j] * p[(size_t)yy * Ri + min(max(x0 - 1 + j, 0), Ri - 1)];
            acc += wy[i] * r;
        }
        out[idx] = acc;
    }
}
int launch_bicubic(const float* in, float* out, int planes, int Ri, int Ro, hipStream_t st) {
    RLCF_ARG_CHECK(planes > 0 && Ri > 0 && Ro > 0);
    const long total = (long)planes * Ro * Ro;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    bicubic_kernel<<<dim3(blocks), dim3(256), 0, st>>>(in, out, planes, Ri, Ro);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ====================================================================================================================================
// LayerNorm folded into the single-pass f16 products of an image tower (RLCF_PREC_F16; gemm_f16.hip MODE 1 / 2, engine.hip).
// Under torch.cuda.amp.autocast the reference's residual stream is an fp16 tensor (TPT/clip/model.py:157-163: LayerNorm computes in
// fp32 and casts back to the input's type; :187-192: x = x + attention(ln_1(x)) adds two fp16 tensors), so the stream is kept as f16
// rows x16 here; LN(x16) W^T + b = rstd (x16 (W gamma)^T - mu rowsum(W gamma)) + (W beta + b): the consumer product reads x16 itself
// against the gamma-folded weight, the producer product (out_proj / c_proj) adds into x16 and leaves the row statistics behind.
// ====================================================================================================================================
// x (f32, the tower input after ln_pre) -> x16, and (mean, rstd) of the STORED f16 rows; one wave per row
__global__ __launch_bounds__(256) void resid16_init_kernel(const float* __restrict__ x, _Float16* __restrict__ x16, float* __restrict__ mr,
                                                           int rows, int width) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f, q = 0.f;
    for (int c = lane * 4; c < width; c += 256) {
        const float4 t = *(const float4*)(x + (size_t)row * width + c);
        const h16x4 o = {(_Float16)t.x, (_Float16)t.y, (_Float16)t.z, (_Float16)t.w};
        *(h16x4*)(x16 + (size_t)row * width + c) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float v = (float)o[e]; s += v; q += v * v; }
    }
    s = wave_sum(s); q = wave_sum(q);
    if (lane == 0) {
        const float mu = s / width, var = fmaxf(q / width - mu * mu, 0.f);
        *(float2*)(mr + (size_t)row * 2) = make_float2(mu, rsqrtf(var + LN_EPS));
    }
}
int launch_resid16_init(const float* x, void* x16, float* mr, int rows, int width, hipStream_t st) {
    RLCF_ARG_CHECK(rows > 0 && width > 0 && width % 4 == 0 && x && x16 && mr);
    resid16_init_kernel<<<dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st>>>(x, (_Float16*)x16, mr, rows, width);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// the P partial (sum, sum of squares) of every row (written by the MODE 2 epilogues, part-major) -> (mean, rstd); added in part order
__global__ __launch_bounds__(256) void ln_stats_final_kernel(const float* __restrict__ part, int P, int rows, int width, float* __restrict__ mr) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float s = 0.f, q = 0.f;
    for (int p = 0; p < P; ++p) { const float2 t = *(const float2*)(part + ((size_t)p * rows + row) * 2); s += t.x; q += t.y; }
    const float mu = s / width, var = fmaxf(q / width - mu * mu, 0.f);
    *(float2*)(mr + (size_t)row * 2) = make_float2(mu, rsqrtf(var + LN_EPS));
}
int launch_ln_stats_final(const float* part, int P, int rows, int width, float* mr, hipStream_t st) {
    RLCF_ARG_CHECK(rows > 0 && P > 0 && part && mr);
    ln_stats_final_kernel<<<dim3((rows + 255) / 256), dim3(256), 0, st>>>(part, P, rows, width, mr);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// rows idx[r] of an f16 matrix -> f32 rows (idx == nullptr: every row in order)
__global__ __launch_bounds__(256) void rows_h2f_kernel(const _Float16* __restrict__ in, int ld_in, const int32_t* __restrict__ idx,
                                                       float* __restrict__ out, int ld_out, int width) {
    const int r = blockIdx.x;
    const size_t src = idx ? (size_t)idx[r] : (size_t)r;
    for (int c = threadIdx.x * 4; c < width; c += 1024) {
        const h16x4 t = *(const h16x4*)(in + src * ld_in + c);
        *(float4*)(out + (size_t)r * ld_out + c) = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    }
}
int launch_rows_h2f(const void* in16, int ld_in, const int32_t* idx, float* out, int ld_out, int rows, int width, hipStream_t st) {
    RLCF_ARG_CHECK(rows > 0 && width > 0 && width % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && in16 && out);
    rows_h2f_kernel<<<dim3(rows), dim3(256), 0, st>>>((const _Float16*)in16, ld_in, idx, out, ld_out, width);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// finalize-time preparation of a folded product: Wg[c, k] = W[c, k] gamma[k];  bprime[c] = b[c] + sum_k W[c, k] beta[k];
// s[c] = inv_scale * sum_k Wf16[c, k] over the ROUNDED f16 copy (what the MFMAs add up for a row of ones).  One workgroup per output row.
__global__ __launch_bounds__(256) void ln_fold_w_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ b, float* __restrict__ Wg, float* __restrict__ bprime, int K) {
    const int c = blockIdx.x;
    __shared__ double red[256];
    double acc = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = W[(size_t)c * K + k];
        Wg[(size_t)c * K + k] = w * gamma[k];
        acc += (double)w * (double)beta[k];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) bprime[c] = (float)((double)(b ? b[c] : 0.f) + red[0]);
}
__global__ __launch_bounds__(256) void rowsum_f16_kernel(const _Float16* __restrict__ Wf, float inv_scale, float* __restrict__ s, int K) {
    const int c = blockIdx.x;
    __shared__ double red[256];
    double acc = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) acc += (double)(float)Wf[(size_t)c * K + k];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) s[c] = (float)(red[0] * (double)inv_scale);
}
int launch_ln_fold_w(const float* W, const float* gamma, const float* beta, const float* b, float* Wg, float* bprime, int N, int K, hipStream_t st) {
    RLCF_ARG_CHECK(N > 0 && K > 0 && W && gamma && beta && Wg && bprime);
    ln_fold_w_kernel<<<dim3(N), dim3(256), 0, st>>>(W, gamma, beta, b, Wg, bprime, K);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_rowsum_f16(const void* Wf16, float inv_scale, float* s, int N, int K, hipStream_t st) {
    RLCF_ARG_CHECK(N > 0 && K > 0 && Wf16 && s);
    rowsum_f16_kernel<<<dim3(N), dim3(256), 0, st>>>((const _Float16*)Wf16, inv_scale, s, K);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
