// ModifiedResNet image tower of CLIP (reference TPT/clip/model.py:10-154), inference only: the reward models RN50x64 & co
// (clip_reward.py:21-34) and the frozen image encoder of a ResNet student in the prompt path (custom_clip.py:325-327 runs it
// under no_grad).  MI355X layout: activations are NHWC matrices [n*H*W, C] so that every 1x1 convolution IS a GEMM on the
// resident activation, 3x3 convolutions are a (ky,kx,c)-ordered patch gather + the same GEMM, eval-mode BatchNorm is folded
// into the GEMM weights/bias at finalize and ReLU / identity-add run in the GEMM epilogue (RLCF_EPI_RELU).
#include "engine.h"
#include <algorithm>
#include <cmath>

#define TRY(x) do { int rc_ = (x); if (rc_ != RLCF_OK) return rc_; } while (0)
#define NEED(ptr) do { if (!(ptr)) return RLCF_ERR_STATE; } while (0)

// ------------------------------------------------------------------ kernels
// Conv2d(bias=False) + BatchNorm2d(eval) -> one affine map: W'[co,(ky,kx,ci)] = W[co,ci,ky,kx]*s[co], b'[co] = beta - mean*s,
// s = gamma / sqrt(var + 1e-5)  (model.py:18-31 in eval mode).  Rows are zero padded to Kp.
__global__ void conv_fold_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ mean, const float* __restrict__ var, float* __restrict__ wout,
                                 float* __restrict__ bout, int Cin, int kk, int Kp) {
    const int co = blockIdx.x;
    const float s = gamma[co] / sqrtf(var[co] + 1e-5f);
    for (int i = threadIdx.x; i < Kp; i += blockDim.x) {
        float v = 0.f;
        if (i < kk * Cin) {
            const int tap = i / Cin, ci = i - tap * Cin;
            v = w[((size_t)co * Cin + ci) * kk + tap] * s;
        }
        wout[(size_t)co * Kp + i] = v;
    }
    if (threadIdx.x == 0) bout[co] = beta[co] - mean[co] * s;
}

// 3x3 patch gather, padding 1: col[(img*Ho + oy)*Wo + ox, (ky*3+kx)*Cin + c] = in(img, oy*stride+ky-1, ox*stride+kx-1, c).
// The input is addressed through element strides, so it reads NCHW images (stem conv1) and NHWC activations alike.
__global__ void im2col3x3_kernel(const float* __restrict__ in, float* __restrict__ col, long total, int Cin, int H, int W, int Ho, int Wo,
                                 int stride, int Kp, long sN, long sC, long sH, long sW) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / Kp;
        const int k = (int)(i - row * Kp);
        float v = 0.f;
        if (k < 9 * Cin) {
            const int tap = k / Cin, c = k - tap * Cin, ky = tap / 3, kx = tap - ky * 3;
            const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho);
            const long img = row / ((long)Wo * Ho);
            const int y = oy * stride + ky - 1, x = ox * stride + kx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) v = in[img * sN + c * sC + y * sH + x * sW];
        }
        col[i] = v;
    }
}

// The same gather for NHWC activations with Cin % 8 == 0, written directly as the split-f16 operand pair of the GEMM:
// 8 channels of one tap per thread (two 16-B loads, two 16-B stores), scaled by the device scalar scale[0] (a power of two).
__global__ void im2col3x3_split_kernel(const float* __restrict__ in, _Float16* __restrict__ hi, _Float16* __restrict__ lo, long total8,
                                       int Cin, int H, int W, int Ho, int Wo, int stride, int Kp, const float* __restrict__ scale) {
    const float sc = scale[0];
    const int K8 = Kp / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        const long row = i / K8;
        const int k = (int)(i - row * K8) * 8;
        h16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) { vh[e] = (_Float16)0.f; vl[e] = (_Float16)0.f; }
        if (k < 9 * Cin) {
            const int tap = k / Cin, c = k - tap * Cin, ky = tap / 3, kx = tap - ky * 3;
            const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho);
            const long img = row / ((long)Wo * Ho);
            const int y = oy * stride + ky - 1, x = ox * stride + kx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const float* p = in + ((img * H + y) * W + x) * Cin + c;
                const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
                const float v[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 hh = (_Float16)v[e];
                    vh[e] = hh;
                    vl[e] = (_Float16)(v[e] - (float)hh);
                }
            }
        }
        const long dst = ((i >> 2) << 3) + (i & 3);          // interleaved pair layout (Kp % 32 == 0): lo = hi + 32 halves
        ((h16x8*)hi)[dst] = vh;
        ((h16x8*)lo)[dst] = vl;
    }
}

// AvgPool2d(2) on an NHWC activation (model.py:25,37,117): out[img, y, x, c] = mean of the 2x2 window
__global__ void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, long total, int C, int Ho, int Wo) {
    const int W = 2 * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int x = (int)(p % Wo), y = (int)((p / Wo) % Ho);
        const long img = p / ((long)Wo * Ho);
        const float* b = in + ((img * 2 * Ho + 2 * y) * W + 2 * x) * C + c;
        out[i] = (((b[0] + b[C]) + b[(long)W * C]) + b[(long)W * C + C]) * 0.25f;
    }
}

// AttentionPool2d tokens (model.py:69-71): tok[img,0] = mean over positions + pos[0]; tok[img,1+p] = x[img,p] + pos[1+p]
__global__ void attnpool_tokens_kernel(const float* __restrict__ x, const float* __restrict__ pos, float* __restrict__ tok, int HW, int E) {
    const int img = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= E) return;
    const float* xi = x + (size_t)img * HW * E + c;
    float* ti = tok + (size_t)img * (HW + 1) * E + c;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) {
        const float v = xi[(size_t)p * E];
        s += v;
        ti[(size_t)(p + 1) * E] = v + pos[(size_t)(p + 1) * E + c];
    }
    ti[0] = s / (float)HW + pos[c];
}

// The single-query attention of AttentionPool2d (model.py:72-90): per (image, head) softmax(q.k^T / 8) v over the T tokens.
// q [n, E] (bias added, unscaled), kv [n*T, 2E] = (k | v), out [n, E]; head_dim 64.  One block of 4 waves per (head, image).
__global__ __launch_bounds__(256) void attnpool_attend_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                             float* __restrict__ out, int T, int E) {
    extern __shared__ float sc[];                 // [T] scores, then probabilities; [4*64] partial outputs
    float* part = sc + T;
    __shared__ float red[4];
    const int head = blockIdx.x, img = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float qd = q[(size_t)img * E + head * 64 + lane] * 0.125f;
    const float* kbase = kv + (size_t)img * T * 2 * E + head * 64 + lane;
    for (int t = wave; t < T; t += 4) {
        const float d = wave_sum(qd * kbase[(size_t)t * 2 * E]);
        if (lane == 0) sc[t] = d;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256) mx = fmaxf(mx, sc[t]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { const float p = expf(sc[t] - mx); sc[t] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.f / ((red[0] + red[1]) + (red[2] + red[3]));
    float acc = 0.f;
    const float* vbase = kbase + E;
    for (int t = wave; t < T; t += 4) acc += sc[t] * vbase[(size_t)t * 2 * E];
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0) out[(size_t)img * E + head * 64 + lane] = ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) * inv;
}

// cat of two device vectors/matrices (k_proj | v_proj rows)
static int concat2(ClipModel& m, const float* a, const float* b, size_t na, size_t nb, const float** out, hipStream_t st) {
    m.derived.emplace_back();
    TRY(m.derived.back().ensure((na + nb) * sizeof(float)));
    float* d = m.derived.back().as<float>();
    RLCF_HIP_CHECK(hipMemcpyAsync(d, a, na * sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(d + na, b, nb * sizeof(float), hipMemcpyDeviceToDevice, st));
    *out = d;
    return RLCF_OK;
}

static const float* raw_of(ClipModel& m, const std::string& k, size_t numel) {
    auto it = m.raw.find(k);
    if (it == m.raw.end()) { rlcf_set_error("missing weight '%s'", k.c_str()); return nullptr; }
    if (it->second.bytes != numel * sizeof(float)) {
        rlcf_set_error("weight '%s': expected %zu elements, got %zu", k.c_str(), numel, it->second.bytes / sizeof(float));
        return nullptr;
    }
    return it->second.as<float>();
}

// max_row sum_k |w[row, k]| -> out[0], max |b| -> out[1] (bit patterns of non-negative floats, atomicMax)
__global__ __launch_bounds__(256) void conv_gain_kernel(const float* __restrict__ w, const float* __restrict__ b, int Kp, unsigned int* __restrict__ out) {
    __shared__ float part[4];
    const int row = blockIdx.x;
    float a = 0.f;
    for (int i = threadIdx.x; i < Kp; i += 256) a += fabsf(w[(size_t)row * Kp + i]);
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(out, __float_as_uint((part[0] + part[1]) + (part[2] + part[3])));
        atomicMax(out + 1, __float_as_uint(fabsf(b[row])));
    }
}
// s = 2^k with (gain max|in| + bmax + max|res|) s in [2^14, 2^15): |C| cannot exceed that bound, so the pairs the epilogue writes as
// C s stay inside f16's range (max 65504) WITHOUT knowing max|C|, which only exists once the launch has finished.  The bound is
// 10-30x above the real maximum for these layers: the pairs then sit around 2^10, where the split-f16 scheme wants them.
__global__ void conv_bound_scale_kernel(const float* __restrict__ amax_in, const float* __restrict__ amax_res, float gain, float bmax,
                                        float* __restrict__ out2) {
    const float B = 1.02f * (gain * amax_in[0] + bmax + (amax_res ? amax_res[0] : 0.f));
    int sh = 0;
    if (B > 0.f && B < INFINITY) sh = 14 - (int)floorf(log2f(B));
    sh = sh < -40 ? -40 : (sh > 40 ? 40 : sh);
    out2[0] = ldexpf(1.0f, sh);
    out2[1] = ldexpf(1.0f, -sh);
}
__global__ void conv_permute_kernel(const float* __restrict__ w, float* __restrict__ wout, int Cin, int kk, int Kp);        // (below)
// s[co] = gamma / sqrt(var + 1e-5): the factor conv_fold_kernel multiplies into the weight, kept apart (ConvW::cs)
__global__ void conv_bn_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ var, float* __restrict__ cs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cs[i] = gamma[i] / sqrtf(var[i] + 1e-5f);
}
static int fold(rlcf_engine* e, ClipModel& m, const std::string& conv, const std::string& bn, int cout, int cin, int k, ConvW& out,
                hipStream_t st) {
    const int kk = k * k;
    const float* w = raw_of(m, conv + ".weight", (size_t)cout * cin * kk);
    const float* g = raw_of(m, bn + ".weight", cout);
    const float* b = raw_of(m, bn + ".bias", cout);
    const float* mu = raw_of(m, bn + ".running_mean", cout);
    const float* var = raw_of(m, bn + ".running_var", cout);
    NEED(w); NEED(g); NEED(b); NEED(mu); NEED(var);
    out.cin = cin; out.cout = cout; out.k = k;
    out.Kp = k == 1 ? cin : (kk * cin + 31) / 32 * 32;
    if (out.Kp % 16) { rlcf_set_error("ModifiedResNet: channel count %d of %s is not a multiple of 16", cin, conv.c_str()); return RLCF_ERR_ARG; }
    m.derived.emplace_back();
    TRY(m.derived.back().ensure((size_t)cout * out.Kp * sizeof(float)));
    float* wp = m.derived.back().as<float>();          // (take the pointers now: the vector may grow)
    m.derived.emplace_back();
    TRY(m.derived.back().ensure((size_t)cout * sizeof(float)));
    float* bp = m.derived.back().as<float>();
    conv_fold_kernel<<<dim3(cout), dim3(256), 0, st>>>(w, g, b, mu, var, wp, bp, cin, kk, out.Kp);
    RLCF_LAUNCH_CHECK();
    out.w = wp; out.b = bp;
    {   // gain / bias bound of the folded convolution (host constants of the pair-emitting epilogues' scale choice)
        TRY(e->dyn.ensure(3 * sizeof(float)));
        RLCF_HIP_CHECK(hipMemsetAsync(e->dyn.p, 0, 2 * sizeof(float), st));
        conv_gain_kernel<<<dim3(cout), dim3(256), 0, st>>>(wp, bp, out.Kp, (unsigned int*)e->dyn.p);
        RLCF_LAUNCH_CHECK();
        float gb[2];
        RLCF_HIP_CHECK(hipMemcpyAsync(gb, e->dyn.p, sizeof(gb), hipMemcpyDeviceToHost, st));
        RLCF_HIP_CHECK(hipStreamSynchronize(st));
        out.gain = gb[0]; out.bmax = gb[1];
    }
    TRY(engine_make_split(e, m, out.w, (size_t)cout * out.Kp, st));
    // The unfolded form, kept only if the raw weight sits on the fp16 grid (then its products need two MFMA passes, and
    // folding the BatchNorm scale into it would take it off the grid).  RLCF_X3_WLO0=0 (read by engine_make_split) and RLCF_CONV_GRID=0: off
    out.wg = nullptr; out.cs = nullptr;
    const char* ev = getenv("RLCF_CONV_GRID");
    if (!(ev && atoi(ev) == 0) && out.Kp % 32 == 0) {
        DevBuf wgb, csb;
        TRY(wgb.ensure((size_t)cout * out.Kp * sizeof(float)));
        conv_permute_kernel<<<dim3(cout), dim3(256), 0, st>>>(w, wgb.as<float>(), cin, kk, out.Kp);
        RLCF_LAUNCH_CHECK();
        TRY(engine_make_split(e, m, wgb.as<float>(), (size_t)cout * out.Kp, st));
        auto it = m.split_of.find(wgb.as<float>());
        if (it != m.split_of.end() && it->second.lo_zero) {
            TRY(csb.ensure((size_t)cout * sizeof(float)));
            conv_bn_scale_kernel<<<dim3((cout + 255) / 256), dim3(256), 0, st>>>(g, var, csb.as<float>(), cout);
            RLCF_LAUNCH_CHECK();
            out.wg = wgb.as<float>(); out.cs = csb.as<float>();
            m.derived.push_back(std::move(wgb));
            m.derived.push_back(std::move(csb));
        } else {
            // (not on the grid: drop the copy again — its split pair stays in m.derived until the next finalize, unused)
            if (it != m.split_of.end()) m.split_of.erase(it);
            RLCF_HIP_CHECK(hipStreamSynchronize(st));
            wgb.release();
        }
    }
    return RLCF_OK;
}

// build_model's ResNet branch (model.py:408-412) fixed the geometry; here: fold, permute and (F16X3) split every weight
int resnet_finalize(rlcf_engine* e, ClipModel& m, hipStream_t st) {
    const rlcf_clip_cfg& c = m.cfg;
    ResNetW& r = m.rn;
    const int w = c.vision_width;
    r.blocks.clear();
    r.bn_enabled = false; r.units.clear(); r.block_unit.clear();
    TRY(fold(e, m, "visual.conv1", "visual.bn1", w / 2, 3, 3, r.stem[0], st));
    TRY(fold(e, m, "visual.conv2", "visual.bn2", w / 2, w / 2, 3, r.stem[1], st));
    TRY(fold(e, m, "visual.conv3", "visual.bn3", w, w / 2, 3, r.stem[2], st));
    int inpl = w;
    for (int s = 0; s < 4; ++s) {
        const int planes = w << s;
        for (int b = 0; b < c.vision_stages[s]; ++b) {
            const std::string p = "visual.layer" + std::to_string(s + 1) + "." + std::to_string(b) + ".";
            BottleW bw;
            bw.stride = (b == 0 && s > 0) ? 2 : 1;
            TRY(fold(e, m, p + "conv1", p + "bn1", planes, inpl, 1, bw.c1, st));
            TRY(fold(e, m, p + "conv2", p + "bn2", planes, planes, 3, bw.c2, st));
            TRY(fold(e, m, p + "conv3", p + "bn3", planes * 4, planes, 1, bw.c3, st));
            bw.has_down = bw.stride > 1 || inpl != planes * 4;
            if (bw.has_down) TRY(fold(e, m, p + "downsample.0", p + "downsample.1", planes * 4, inpl, 1, bw.down, st));
            r.blocks.push_back(bw);
            inpl = planes * 4;
        }
    }
    const int E = w * 32, D = c.embed_dim;
    r.E = E; r.heads = E / 64; r.out_hw = c.image_resolution / 32;
    const int T = r.out_hw * r.out_hw + 1;
    NEED(r.pos = raw_of(m, "visual.attnpool.positional_embedding", (size_t)T * E));
    const float *kw, *vw, *kb, *vb;
    NEED(r.q_w = raw_of(m, "visual.attnpool.q_proj.weight", (size_t)E * E)); NEED(r.q_b = raw_of(m, "visual.attnpool.q_proj.bias", E));
    NEED(kw = raw_of(m, "visual.attnpool.k_proj.weight", (size_t)E * E));    NEED(kb = raw_of(m, "visual.attnpool.k_proj.bias", E));
    NEED(vw = raw_of(m, "visual.attnpool.v_proj.weight", (size_t)E * E));    NEED(vb = raw_of(m, "visual.attnpool.v_proj.bias", E));
    NEED(r.c_w = raw_of(m, "visual.attnpool.c_proj.weight", (size_t)D * E)); NEED(r.c_b = raw_of(m, "visual.attnpool.c_proj.bias", D));
    TRY(concat2(m, kw, vw, (size_t)E * E, (size_t)E * E, &r.kv_w, st));
    TRY(concat2(m, kb, vb, E, E, &r.kv_b, st));
    TRY(engine_make_split(e, m, r.kv_w, (size_t)2 * E * E, st));
    // per-image workspace: the widest activation is [R/2 * R/2, width] (== [R/4 * R/4, 4*width]); the largest 3x3 patch matrix
    // is found by walking the convolutions (stem conv2 and layer2.0.conv2 tie at 9/8 * R^2 * width before padding)
    const size_t R = c.image_resolution;
    r.act_per_img = R * R / 4 * w;
    r.col_per_img = std::max(R * R / 4 * (size_t)r.stem[1].Kp, R * R / 4 * (size_t)r.stem[0].Kp);
    size_t H = R / 4;
    for (const BottleW& b : r.blocks) {
        r.col_per_img = std::max(r.col_per_img, H * H * (size_t)b.c2.Kp);
        H /= b.stride;
    }
    r.present = true;
    return RLCF_OK;
}

static int rn_ensure(rlcf_engine* e, const ResNetW& r, int chunk, int T) {
    for (DevBuf& b : e->rn_buf) TRY(b.ensure((size_t)chunk * r.act_per_img * sizeof(float)));
    TRY(e->rn_col.ensure((size_t)chunk * r.col_per_img * sizeof(float)));
    TRY(e->rn_tok.ensure((size_t)chunk * T * r.E * sizeof(float)));
    TRY(e->rn_kv.ensure((size_t)chunk * T * 2 * r.E * sizeof(float)));
    TRY(e->rn_q.ensure((size_t)chunk * r.E * sizeof(float)));
    TRY(e->rn_att.ensure((size_t)chunk * r.E * sizeof(float)));
    if (prec_x3(e)) {
        const size_t need = std::max((size_t)chunk * r.col_per_img, (size_t)chunk * r.act_per_img);
        if (need > e->a_split_elems) {
            TRY(e->a_hi.ensure(need * 4));
            e->a_split_elems = need;
        }
    }
    return RLCF_OK;
}

// kernels that end in one atomicMax per block: few, fat blocks (thousands of atomics on one address cost more than the pass itself)
static inline dim3 grid_amax(long total) { return dim3((unsigned)std::min<long>((total + 255) / 256, 2048)); }
static inline dim3 grid_for(long total) { return dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)); }

// conv (+folded bn) (+identity) (+relu) on an NHWC activation; 3x3 goes through the patch matrix
// in_amax / out_amax: device scalars holding max|in| (if known) and receiving max|out| (GEMM epilogue), so that the split-f16
// operand scale of the next convolution needs no extra pass over the activation
static int conv(rlcf_engine* e, const ConvW& cw, const float* in, const float* in_amax, int n, int H, int W, int stride, bool nchw,
                const float* res, int epi, float* out, float* out_amax, hipStream_t st) {
    const int Ho = H / stride, Wo = W / stride;
    const long M = (long)n * Ho * Wo;
    const float* A = in;
    static int no_implicit = -1;                          // RLCF_CONV_IM2COL=1: 3x3 convolutions back on the patch matrix (A/B measurements)
    if (no_implicit < 0) { const char* ev = getenv("RLCF_CONV_IM2COL"); no_implicit = ev ? atoi(ev) : 0; }
    if (cw.k == 3 && !nchw && stride == 1 && prec_x3(e) && !no_implicit && cw.Kp == 9 * cw.cin && gemm_f16x3_conv3x3_ok((int)M, cw.cout, cw.cin) &&
        (size_t)M * cw.cin <= e->a_split_elems && engine_has_split(e, cw.w) && !prec_single(e)) {
        // implicit GEMM: the activation is split ONCE into operand pairs (M x cin, not the 9x larger patch matrix) and the 256x256
        // kernel's DMA gathers every tap's K tiles from it directly
        TRY(e->dyn.ensure(3 * sizeof(float)));
        if (in_amax) TRY(launch_dyn_scale_from(in_amax, e->dyn.as<float>() + 1, st));
        else TRY(launch_dyn_scale(in, (int64_t)M * cw.cin, e->dyn.as<float>(), st));
        // (weight on the fp16 grid: the unfolded copy — two MFMA passes — with the BatchNorm scale as the epilogue's column factor)
        if (cw.wg && engine_has_split(e, cw.wg)) gemm_f16x3_next_col_scale(cw.cs);
        return engine_gemm_conv3x3(e, in, e->dyn.as<float>() + 1, cw.wg && engine_has_split(e, cw.wg) ? cw.wg : cw.w, cw.b, res, cw.cout, out, cw.cout, n, H, W,
                                   cw.cin, cw.cout, epi, st, out_amax);
    }
    if (cw.k == 3 && !nchw && prec_x3(e) && cw.cin % 8 == 0 && M > 512 && (size_t)M * cw.Kp <= e->a_split_elems &&
        engine_has_split(e, cw.w)) {
        // split-f16 mode: the patch matrix is written once, already as the (hi, lo) operand pair; its power-of-two scale comes
        // from max|activation| (the patch matrix holds the same values), found on the 9x smaller activation
        TRY(e->dyn.ensure(3 * sizeof(float)));
        if (in_amax) TRY(launch_dyn_scale_from(in_amax, e->dyn.as<float>() + 1, st));
        else TRY(launch_dyn_scale(in, (int64_t)n * H * W * cw.cin, e->dyn.as<float>(), st));
        const long total8 = M * (cw.Kp / 8);
        im2col3x3_split_kernel<<<grid_for(total8), dim3(256), 0, st>>>(in, (_Float16*)e->a_hi.p, (_Float16*)e->a_hi.p + 32, total8, cw.cin, H, W,
                                                                      Ho, Wo, stride, cw.Kp, e->dyn.as<float>() + 1);
        RLCF_LAUNCH_CHECK();
        if (cw.wg && engine_has_split(e, cw.wg)) gemm_f16x3_next_col_scale(cw.cs);
        return engine_gemm_presplit(e, cw.wg && engine_has_split(e, cw.wg) ? cw.wg : cw.w, cw.b, res, cw.cout, out, cw.cout, (int)M, cw.cout, cw.Kp, epi,
                                    e->dyn.as<float>() + 2, st, out_amax);
    }
    if (cw.k == 3) {
        const long total = M * cw.Kp;
        const long sN = (long)cw.cin * H * W, sC = nchw ? (long)H * W : 1, sH = nchw ? W : (long)W * cw.cin, sW = nchw ? 1 : cw.cin;
        im2col3x3_kernel<<<grid_for(total), dim3(256), 0, st>>>(in, e->rn_col.as<float>(), total, cw.cin, H, W, Ho, Wo, stride, cw.Kp, sN, sC,
                                                                sH, sW);
        RLCF_LAUNCH_CHECK();
        A = e->rn_col.as<float>();
    }
    return engine_gemm(e, A, cw.Kp, cw.w, cw.Kp, cw.b, res, cw.cout, out, cw.cout, (int)M, cw.cout, cw.Kp, epi, st,
                       cw.k == 1 ? in_amax : nullptr, out_amax);
}

// ---- pair-emitting form (inference tower, split-f16 mode): a convolution reads its input as operand pairs (written by the producing
// GEMM's epilogue, or split here from f32) and can write its output as pairs for the next one, scaled by a power of two chosen from an
// upper bound of |output| (conv_bound_scale_kernel): no stand-alone split pass, and no f32 copy of tensors only convolutions read.
struct ConvIn { const float* f32 = nullptr; const float* amax = nullptr; const void* pairs = nullptr; const float* scale2 = nullptr; };
struct ConvOut { float* f32 = nullptr; void* pairs = nullptr; float* scale2 = nullptr; float* amax = nullptr; };
static bool conv_pairs_ok(rlcf_engine* e, const ConvW& cw, long M) {
    if (!prec_x3(e) || prec_single(e) || M <= 512 || cw.cin % 32 || cw.cout % 32 || !engine_has_split(e, cw.w)) return false;
    if ((size_t)M * cw.cin > e->a_split_elems) return false;
    return cw.k == 1 || (cw.Kp == 9 * cw.cin && gemm_f16x3_conv3x3_ok((int)M, cw.cout, cw.cin));
}
// stride 1, NHWC; res / res_amax: identity branch and max|identity| (for the bound)
static int conv_pairs(rlcf_engine* e, const ConvW& cw, ConvIn in, int n, int H, int W, const float* res, const float* res_amax, int epi,
                      ConvOut out, hipStream_t st) {
    const long M = (long)n * H * W;
    if (!in.pairs) {
        void* p = nullptr;
        TRY(engine_split_operand(e, in.f32, M * cw.cin, in.amax, &p, &in.scale2, st));
        in.pairs = p;
    }
    if (out.pairs) {
        if (!in.amax) { rlcf_set_error("conv_pairs: the bound of a pair output needs max|input|"); return RLCF_ERR_STATE; }
        // RLCF_CONV_BOUND_KERNEL=0: derive the scale INSIDE the GEMM launch instead of this one-thread launch (GemmX3Args::bnd_*; same values,
        // 136 launches per test image fewer at RN50x64) — built in round 5 and measured SLOWER, twice, A/B/A/B on one box: configs[4]
        // 64.4 -> 69.2 ms/image (profiles/r5_notes.md); the stand-alone launch stays the default
        static int bound_kernel = -1;
        if (bound_kernel < 0) { const char* ev = getenv("RLCF_CONV_BOUND_KERNEL"); bound_kernel = ev ? atoi(ev) : 1; }
        if (bound_kernel) {
            conv_bound_scale_kernel<<<dim3(1), dim3(1), 0, st>>>(in.amax, res_amax, cw.gain, cw.bmax, out.scale2);
            RLCF_LAUNCH_CHECK();
        } else gemm_f16x3_next_bound(in.amax, res_amax, cw.gain, cw.bmax, out.scale2);     // (derived inside the GEMM launch below)
    }
    // out.amax is a fresh slot of the chunk's zeroed max|.| arena (resnet_encode): nothing to clear here
    // weight on the fp16 grid: the unfolded copy (two MFMA passes) with the BatchNorm scale as the epilogue's column factor
    const float* Wuse = cw.wg ? cw.wg : cw.w;
    if (cw.wg) gemm_f16x3_next_col_scale(cw.cs);
    if (cw.k == 1)
        return engine_gemm_pairs(e, in.pairs, cw.cin, in.scale2 + 1, Wuse, cw.b, res, cw.cout, out.f32, cw.cout, out.pairs, out.scale2, (int)M,
                                 cw.cout, epi, st, out.amax);
    return engine_gemm_conv3x3(e, nullptr, in.scale2, Wuse, cw.b, res, cw.cout, out.f32, cw.cout, n, H, W, cw.cin, cw.cout, epi, st, out.amax,
                               in.pairs, out.pairs, out.scale2);
}

static int avgpool2(const float* in, float* out, int n, int Ho, int Wo, int C, hipStream_t st) {
    const long total = (long)n * Ho * Wo * C;
    avgpool2_kernel<<<grid_for(total), dim3(256), 0, st>>>(in, out, total, C, Ho, Wo);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ModifiedResNet.forward (model.py:138-154) + L2 normalisation of the features (custom_clip.py:330, clip_reward.py:136)
int resnet_encode(rlcf_engine* e, ClipModel& m, const float* images, int n_total, float* feats, hipStream_t st) {
    const rlcf_clip_cfg& c = m.cfg;
    const ResNetW& r = m.rn;
    const int R = c.image_resolution, w = c.vision_width, E = r.E, D = c.embed_dim, HW = r.out_hw * r.out_hw, T = HW + 1;
    const size_t budget = (size_t)1 << 30;                                     // floats in the patch matrix of one chunk (4 GB)
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_total, budget / r.col_per_img));
    TRY(rn_ensure(e, r, chunk, T));
    float *X = e->rn_buf[0].as<float>(), *A = e->rn_buf[1].as<float>(), *B = e->rn_buf[2].as<float>(), *Cb = e->rn_buf[3].as<float>(),
          *Dd = e->rn_buf[4].as<float>();
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = std::min(chunk, n_total - i0);
        const float* img = images + (size_t)i0 * 3 * R * R;
        int H = R / 2;
        // max|.| of the activation buffers (X, A, B, Dd), written (atomicMax) by whichever GEMM writes the buffer; an average-pooled copy
        // inherits the scalar of its source (a valid, slightly conservative bound).  Every write of a buffer gets a FRESH scalar of an arena
        // that is cleared ONCE per chunk (round 4 cleared one scalar per convolution: 195 four-byte fills per test image at RN50x64,
        // 0.96 ms of the 65 ms step — profiles/r4_config5_kernel_stats.txt)
        const int am_slots = 8 + 6 * (int)r.blocks.size();
        TRY(e->rn_amax.ensure((size_t)am_slots * sizeof(float)));
        float* am = e->rn_amax.as<float>();
        RLCF_HIP_CHECK(hipMemsetAsync(am, 0, (size_t)am_slots * sizeof(float), st));
        int am_next = 0;
        float *amX = nullptr, *amA = nullptr, *amB = nullptr, *amD = nullptr;
#define ZERO(p) do { if (am_next >= am_slots) { rlcf_set_error("resnet_encode: max|.| arena exhausted"); return RLCF_ERR_STATE; } (p) = am + am_next++; } while (0)
        // stem (model.py:139-145): conv 3x3 s2 -> conv 3x3 -> conv 3x3 -> avgpool 2
        ZERO(amA);
        TRY(conv(e, r.stem[0], img, nullptr, n, R, R, 2, true, nullptr, RLCF_EPI_RELU, A, amA, st));
        ZERO(amB);
        TRY(conv(e, r.stem[1], A, amA, n, H, H, 1, false, nullptr, RLCF_EPI_RELU, B, amB, st));
        ZERO(amA);
        TRY(conv(e, r.stem[2], B, amB, n, H, H, 1, false, nullptr, RLCF_EPI_RELU, A, amA, st));
        H /= 2;
        TRY(avgpool2(A, X, n, H, H, w, st));
        amX = amA;                                                             // (the pooled copy's bound is its source's)
        static int no_fuse = -1;                                               // RLCF_RN_NOFUSE=1: every block on the f32-activation path (A/B)
        if (no_fuse < 0) { const char* ev = getenv("RLCF_RN_NOFUSE"); no_fuse = ev ? atoi(ev) : 0; }
        bool x_pairs = false;                                                  // Xp holds X as operand pairs (scale sX)
        for (DevBuf& pb : e->rn_pairs) TRY(pb.ensure((size_t)chunk * r.act_per_img * 4));
        TRY(e->rn_scale.ensure(8 * sizeof(float)));
        void *Xp = e->rn_pairs[0].p, *Ap = e->rn_pairs[1].p, *Bp = e->rn_pairs[2].p;
        float *sX = e->rn_scale.as<float>(), *sA = sX + 2, *sB = sX + 4;
        for (const BottleW& b : r.blocks) {                                    // Bottleneck.forward, model.py:42-55
            const int planes = b.c1.cout, Ho = H / b.stride;
            const long Mi = (long)n * H * H;
            if (!no_fuse && b.stride == 1 && conv_pairs_ok(e, b.c1, Mi) && conv_pairs_ok(e, b.c2, Mi) && conv_pairs_ok(e, b.c3, Mi) &&
                (!b.has_down || conv_pairs_ok(e, b.down, Mi))) {
                // conv1 -> pairs -> conv2 (implicit 3x3) -> pairs -> conv3 (+ identity, ReLU) -> X as f32 (the next identity) and as pairs
                ConvIn in1;
                if (x_pairs) { in1.pairs = Xp; in1.scale2 = sX; in1.amax = amX; } else { in1.f32 = X; in1.amax = amX; }
                ZERO(amA);
                TRY(conv_pairs(e, b.c1, in1, n, H, H, nullptr, nullptr, RLCF_EPI_RELU, ConvOut{nullptr, Ap, sA, amA}, st));
                if (!in1.pairs) { in1.f32 = X; }                                    // (conv_pairs split into the shared scratch: not reusable)
                ZERO(amB);
                TRY(conv_pairs(e, b.c2, ConvIn{nullptr, amA, Ap, sA}, n, H, H, nullptr, nullptr, RLCF_EPI_RELU, ConvOut{nullptr, Bp, sB, amB}, st));
                const float* idn = X;
                const float* idn_amax = amX;
                if (b.has_down) {
                    ZERO(amD);
                    TRY(conv_pairs(e, b.down, in1, n, H, H, nullptr, nullptr, RLCF_EPI_NONE, ConvOut{Dd, nullptr, nullptr, amD}, st));
                    idn = Dd; idn_amax = amD;
                }
                // (the bound kernel inside reads max|identity| of the OLD X; the new X — updated in place — gets its own scalar)
                float* amXn = nullptr;
                ZERO(amXn);
                TRY(conv_pairs(e, b.c3, ConvIn{nullptr, amB, Bp, sB}, n, H, H, idn, idn_amax, RLCF_EPI_RELU, ConvOut{X, Xp, sX, amXn}, st));
                amX = amXn;
                x_pairs = true;
                continue;
            }
            x_pairs = false;
            ZERO(amA);
            TRY(conv(e, b.c1, X, amX, n, H, H, 1, false, nullptr, RLCF_EPI_RELU, A, amA, st));
            ZERO(amB);
            TRY(conv(e, b.c2, A, amA, n, H, H, 1, false, nullptr, RLCF_EPI_RELU, B, amB, st));
            const float* t2 = B;
            if (b.stride > 1) { TRY(avgpool2(B, A, n, Ho, Ho, planes, st)); t2 = A; }      // max|A| <= max|B|: keep using amB
            const float* idn = X;
            if (b.has_down) {
                const float* xp = X;
                if (b.stride > 1) { TRY(avgpool2(X, Cb, n, Ho, Ho, b.down.cin, st)); xp = Cb; }
                ZERO(amD);
                TRY(conv(e, b.down, xp, amX, n, Ho, Ho, 1, false, nullptr, RLCF_EPI_NONE, Dd, amD, st));
                idn = Dd;
            }
            // in place when idn == X (elementwise): the GEMM reads X only as the residual operand; the new X gets its own scalar
            ZERO(amX);
            TRY(conv(e, b.c3, t2, amB, n, Ho, Ho, 1, false, idn, RLCF_EPI_RELU, X, amX, st));
            H = Ho;
        }
#undef ZERO
        // attention pool (model.py:68-91)
        attnpool_tokens_kernel<<<dim3((E + 255) / 256, n), dim3(256), 0, st>>>(X, r.pos, e->rn_tok.as<float>(), HW, E);
        RLCF_LAUNCH_CHECK();
        TRY(engine_gemm(e, e->rn_tok.as<float>(), T * E, r.q_w, E, r.q_b, nullptr, 0, e->rn_q.as<float>(), E, n, E, E, RLCF_EPI_NONE, st));
        TRY(engine_gemm(e, e->rn_tok.as<float>(), E, r.kv_w, E, r.kv_b, nullptr, 0, e->rn_kv.as<float>(), 2 * E, n * T, 2 * E, E,
                        RLCF_EPI_NONE, st));
        attnpool_attend_kernel<<<dim3(r.heads, n), dim3(256), (T + 256) * sizeof(float), st>>>(e->rn_q.as<float>(), e->rn_kv.as<float>(),
                                                                                               e->rn_att.as<float>(), T, E);
        RLCF_LAUNCH_CHECK();
        TRY(engine_gemm(e, e->rn_att.as<float>(), E, r.c_w, E, r.c_b, nullptr, 0, e->feat_raw.as<float>(), D, n, D, E, RLCF_EPI_NONE, st));
        TRY(launch_l2norm_rows(e->feat_raw.as<float>(), feats + (size_t)i0 * D, nullptr, n, D, st));
    }
    return RLCF_OK;
}

// ====================================================================================================================================
// Norm-layer tuning of a ModifiedResNet STUDENT: CLIPCLS_TTA(arch = RN*, only_norm = True) of TPT/tune_cls_rl.py — parameters() are
// the BatchNorm2d weights / biases whose name contains 'bn' (custom_clip.py:481-485; `downsample.1` is missed by that test and stays
// frozen), and CLIPCLS_TTA.train (custom_clip.py:487-497) keeps EVERY BatchNorm2d in training mode, also under model.eval().  So every
// pass of the student — the N views of the first step, the selected views of later steps and the final clean-view inference — runs
// the TRAIN-FORM tower below: convolution as a GEMM on the UNFOLDED weights, per-channel batch statistics over (n, H, W), then
//   `--prior_strength` < 0 (parser default): y = gamma (z - mu_B) / sqrt(var_B + eps) + beta, running statistics updated in place
//                                            (momentum 0.1, unbiased variance), gradient THROUGH the statistics (views coupled);
//   `--prior_strength` s >= 0 (tune_cls_rl.py:35-44,73-76): statistics = prior * running + (1 - prior) * batch (unbiased variance),
//                                            prior = s / (s + 1), as constants of the backward; running statistics untouched.
// Only BatchNorm parameters receive gradients (weights are frozen): the backward is the dX chain through the convolutions
// (dX = dZ . W as a GEMM on the transposed / tap-flipped weights), average pools, ReLUs, the attention pool, and per BatchNorm the
// two column sums d beta = sum g, d gamma = sum g x_hat (fixed-order two-stage reductions: bit-reproducible).
// ====================================================================================================================================
#define BN_EPS 1e-5f
#define BN_ROWS 256                  // rows per partial sum of the column reductions

// W[co, ci, ky, kx] -> [co, (ky,kx,ci)] zero padded to Kp (conv_fold_kernel without the BatchNorm scale)
__global__ void conv_permute_kernel(const float* __restrict__ w, float* __restrict__ wout, int Cin, int kk, int Kp) {
    const int co = blockIdx.x;
    for (int i = threadIdx.x; i < Kp; i += blockDim.x) {
        float v = 0.f;
        if (i < kk * Cin) { const int tap = i / Cin, ci = i - tap * Cin; v = w[((size_t)co * Cin + ci) * kk + tap]; }
        wout[(size_t)co * Kp + i] = v;
    }
}
// operand of the 3x3 convolution's dX: wT[ci, (ky,kx,co)] = W[co, ci, 2-ky, 2-kx], zero padded to KpT (a stride-1, pad-1 correlation's
// input gradient is the correlation of dZ with the flipped, channel-transposed kernel)
__global__ void conv_flip_kernel(const float* __restrict__ w, float* __restrict__ wT, int Cin, int Cout, int KpT) {
    const int ci = blockIdx.x;
    for (int i = threadIdx.x; i < KpT; i += blockDim.x) {
        float v = 0.f;
        if (i < 9 * Cout) {
            const int tap = i / Cout, co = i - tap * Cout, ky = tap / 3, kx = tap - ky * 3;
            v = w[((size_t)co * Cin + ci) * 9 + (2 - ky) * 3 + (2 - kx)];
        }
        wT[(size_t)ci * KpT + i] = v;
    }
}
// stage 1 of a column reduction over M rows: part[chunk, 0, c] = sum z, part[chunk, 1, c] = sum z^2 over the chunk's BN_ROWS rows
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, float* __restrict__ part, long M, int C) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    __shared__ float s1[4][64], s2[4][64];
    const long r0 = (long)blockIdx.y * BN_ROWS, r1 = r0 + BN_ROWS < M ? r0 + BN_ROWS : M;
    float a = 0.f, b = 0.f;
    if (c < C)
        for (long r = r0 + q; r < r1; r += 4) { const float v = z[r * C + c]; a += v; b += v * v; }
    s1[q][threadIdx.x & 63] = a; s2[q][threadIdx.x & 63] = b;
    __syncthreads();
    if (q == 0 && c < C) {
        const int l = threadIdx.x;
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = (s1[0][l] + s1[1][l]) + (s1[2][l] + s1[3][l]);
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = (s2[0][l] + s2[1][l]) + (s2[2][l] + s2[3][l]);
    }
}
// stage 2: batch mean / variance (double; block = 64 channels x 16 chunk groups meeting in LDS in a fixed order), the statistics the
// pass normalises with, running-statistics update.
//   mode 0: eval (running statistics)   1: train (batch statistics, running <- 0.9 running + 0.1 batch, unbiased variance)
//   2: prior blend (prior * running + (1 - prior) * batch with the UNBIASED batch variance; running untouched)
// ms[c] = mean used, ms[C + c] = 1 / sqrt(var used + eps)
__global__ __launch_bounds__(1024) void bn_stats_final_kernel(const float* __restrict__ part, int chunks, long M, int C, float* __restrict__ rmean,
                                                              float* __restrict__ rvar, float* __restrict__ ms, int mode, float prior) {
    __shared__ double r1[16][64], r2[16][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + l;
    double s = 0.0, s2 = 0.0;
    if (mode != 0 && c < C)
        for (int k = q; k < chunks; k += 16) { s += part[((size_t)k * 2) * C + c]; s2 += part[((size_t)k * 2 + 1) * C + c]; }
    r1[q][l] = s; r2[q][l] = s2;
    __syncthreads();
    if (q != 0 || c >= C) return;
    float mu_u = rmean[c], var_u = rvar[c];
    if (mode != 0) {
        s = 0.0; s2 = 0.0;
        for (int k = 0; k < 16; ++k) { s += r1[k][l]; s2 += r2[k][l]; }
        const double mu = s / (double)M;
        double var = s2 / (double)M - mu * mu;
        if (var < 0.0) var = 0.0;
        const double varu = M > 1 ? var * (double)M / (double)(M - 1) : var;
        if (mode == 1) {
            mu_u = (float)mu; var_u = (float)var;
            rmean[c] = 0.9f * rmean[c] + 0.1f * (float)mu;
            rvar[c] = 0.9f * rvar[c] + 0.1f * (float)varu;
        } else {
            mu_u = prior * rmean[c] + (1.f - prior) * (float)mu;
            var_u = prior * rvar[c] + (1.f - prior) * (float)varu;
        }
    }
    ms[c] = mu_u;
    ms[C + c] = 1.0f / sqrtf(var_u + BN_EPS);
}
// block-wide max of non-negative floats -> one atomicMax (their bit patterns order like the values)
__device__ __forceinline__ void block_amax(float m, unsigned int* __restrict__ out) {
    __shared__ float red[16];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < (blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
        atomicMax(out, __float_as_uint(m));
    }
}
// y = gamma (z - mean) rstd + beta (+ identity) (ReLU); 4 channels per thread
// amax (optional): receives max|y| (the split-f16 operand scale of the convolution that reads y needs no pass of its own)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ ms, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ idn, float* __restrict__ y,
                                                       long total4, int C, int relu, unsigned int* __restrict__ amax) {
    float am = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % C);
        const float4 v = ((const float4*)z)[i];
        const float4 mu = *(const float4*)(ms + c), rs = *(const float4*)(ms + C + c), g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
        float o[4] = {(v.x - mu.x) * rs.x * g.x + b.x, (v.y - mu.y) * rs.y * g.y + b.y, (v.z - mu.z) * rs.z * g.z + b.z, (v.w - mu.w) * rs.w * g.w + b.w};
        if (idn) { const float4 r = ((const float4*)idn)[i]; o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w; }
        if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
        ((float4*)y)[i] = make_float4(o[0], o[1], o[2], o[3]);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    if (amax) block_amax(am, amax);
}
// backward stage 1: g = dy (* [y > 0] when the unit ends in a ReLU); part[chunk, 0, c] = sum g, part[chunk, 1, c] = sum g x_hat
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                                             const float* __restrict__ ms, float* __restrict__ part, long M, int C, int relu) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    __shared__ float s1[4][64], s2[4][64];
    const long r0 = (long)blockIdx.y * BN_ROWS, r1 = r0 + BN_ROWS < M ? r0 + BN_ROWS : M;
    float a = 0.f, b = 0.f;
    if (c < C) {
        const float mu = ms[c], rs = ms[C + c];
        for (long r = r0 + q; r < r1; r += 4) {
            float g = dy[r * C + c];
            if (relu && !(y[r * C + c] > 0.f)) g = 0.f;
            a += g; b += g * ((z[r * C + c] - mu) * rs);
        }
    }
    s1[q][threadIdx.x & 63] = a; s2[q][threadIdx.x & 63] = b;
    __syncthreads();
    if (q == 0 && c < C) {
        const int l = threadIdx.x;
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = (s1[0][l] + s1[1][l]) + (s1[2][l] + s1[3][l]);
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = (s2[0][l] + s2[1][l]) + (s2[2][l] + s2[3][l]);
    }
}
// backward stage 2: sums[c] = d beta, sums[C + c] = d gamma (block = 64 channels x 16 chunk groups, fixed order); grads written when
// the unit is tuned
__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const float* __restrict__ part, int chunks, int C, float* __restrict__ sums,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float r1[16][64], r2[16][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + l;
    float s = 0.f, s2 = 0.f;
    if (c < C)
        for (int k = q; k < chunks; k += 16) { s += part[((size_t)k * 2) * C + c]; s2 += part[((size_t)k * 2 + 1) * C + c]; }
    r1[q][l] = s; r2[q][l] = s2;
    __syncthreads();
    if (q != 0 || c >= C) return;
    s = 0.f; s2 = 0.f;
    for (int k = 0; k < 16; ++k) { s += r1[k][l]; s2 += r2[k][l]; }
    sums[c] = s; sums[C + c] = s2;
    if (dgamma) { dgamma[c] = s2; dbeta[c] = s; }
}
// backward stage 3: dz = gamma rstd (g - [train mode: (sum g + x_hat sum g x_hat) / M]); gout (optional) = g, the gradient that also
// flows into the identity branch of a block's last unit
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                    const float* __restrict__ ms, const float* __restrict__ gamma, const float* __restrict__ sums,
                                    float* __restrict__ dz, float* __restrict__ gout, long total, long M, int C, int relu, int through_stats,
                                    unsigned int* __restrict__ amax) {
    const float invM = 1.0f / (float)M;
    float am = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float g = dy[i];
        if (relu && !(y[i] > 0.f)) g = 0.f;
        if (gout) gout[i] = g;
        const float rs = ms[C + c];
        float t = g;
        if (through_stats) t -= (sums[c] + (z[i] - ms[c]) * rs * sums[C + c]) * invM;
        const float o = gamma[c] * rs * t;
        dz[i] = o;
        am = fmaxf(am, fabsf(o));
    }
    if (amax) block_amax(am, amax);
}
// AvgPool2d(2) backward on NHWC: din[img, 2y+a, 2x+b, c] = dout[img, y, x, c] / 4
__global__ void avgpool2_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, long total_in, int C, int Ho, int Wo) {
    const int W = 2 * Wo, H = 2 * Ho;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_in; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int x = (int)(p % W), y = (int)((p / W) % H);
        const long img = p / ((long)W * H);
        din[i] = 0.25f * dout[((img * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c];
    }
}
__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] += b[i];
}
// attention pool, forward with the probabilities kept: as attnpool_attend_kernel, plus prob[img, head, t]
__global__ __launch_bounds__(256) void attnpool_attend_save_kernel(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ out,
                                                                  float* __restrict__ prob, int T, int E) {
    extern __shared__ float sc[];
    float* part = sc + T;
    __shared__ float red[4];
    const int head = blockIdx.x, img = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float qd = q[(size_t)img * E + head * 64 + lane] * 0.125f;
    const float* kbase = kv + (size_t)img * T * 2 * E + head * 64 + lane;
    for (int t = wave; t < T; t += 4) { const float d = wave_sum(qd * kbase[(size_t)t * 2 * E]); if (lane == 0) sc[t] = d; }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256) mx = fmaxf(mx, sc[t]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { const float p = expf(sc[t] - mx); sc[t] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int t = threadIdx.x; t < T; t += 256) prob[((size_t)img * gridDim.x + head) * T + t] = sc[t] * inv;
    float acc = 0.f;
    const float* vbase = kbase + E;
    for (int t = wave; t < T; t += 4) acc += sc[t] * vbase[(size_t)t * 2 * E];
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0) out[(size_t)img * E + head * 64 + lane] = ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) * inv;
}
// ... and its backward: datt [n, E] -> dq [n, E] (gradient of the projected, unscaled query), dkv [n*T, 2E]
//   dv_t = p_t datt;  dp_t = <datt, v_t>;  ds = p (dp - sum p dp);  dq = sum_t ds_t k_t / 8;  dk_t = ds_t q / 8
__global__ __launch_bounds__(256) void attnpool_attend_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ prob,
                                                                 const float* __restrict__ datt, float* __restrict__ dq, float* __restrict__ dkv,
                                                                 int T, int E) {
    extern __shared__ float sc[];                 // [T] dp, then ds; [4*64] partial dq
    float* part = sc + T;
    __shared__ float red[4];
    const int head = blockIdx.x, img = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* pr = prob + ((size_t)img * gridDim.x + head) * T;
    const float da = datt[(size_t)img * E + head * 64 + lane];
    const float qd = q[(size_t)img * E + head * 64 + lane];
    const float* kbase = kv + (size_t)img * T * 2 * E + head * 64 + lane;
    float* dkbase = dkv + (size_t)img * T * 2 * E + head * 64 + lane;
    for (int t = wave; t < T; t += 4) {
        const float d = wave_sum(da * kbase[(size_t)t * 2 * E + E]);
        if (lane == 0) sc[t] = d;
        dkbase[(size_t)t * 2 * E + E] = pr[t] * da;                  // dv
    }
    __syncthreads();
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) s += pr[t] * sc[t];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) sc[t] = pr[t] * (sc[t] - tot);
    __syncthreads();
    float acc = 0.f;
    for (int t = wave; t < T; t += 4) {
        const float ds = sc[t];
        acc += ds * kbase[(size_t)t * 2 * E];
        dkbase[(size_t)t * 2 * E] = ds * qd * 0.125f;              // dk
    }
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0) dq[(size_t)img * E + head * 64 + lane] = ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) * 0.125f;
}
// tokens backward: dx[img, p, c] = dtok[img, 1 + p, c] + dtok[img, 0, c] / HW  (the mean token reads every position)
__global__ void attnpool_tokens_bwd_kernel(const float* __restrict__ dtok, float* __restrict__ dx, int HW, int E, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % E);
        const long p = (i / E) % HW, img = i / ((long)E * HW);
        dx[i] = dtok[((size_t)img * (HW + 1) + 1 + p) * E + c] + dtok[(size_t)img * (HW + 1) * E + c] / (float)HW;
    }
}
// dtok[img, 0, :] += dq0[img, :]
__global__ void attnpool_add_row0_kernel(float* __restrict__ dtok, const float* __restrict__ dq0, int T, int E, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        dtok[(size_t)(i / E) * T * E + (i % E)] += dq0[i];
}

static const float* derived_buf(ClipModel& m, size_t floats, float** out) {
    m.derived.emplace_back();
    if (m.derived.back().ensure(floats * sizeof(float)) != RLCF_OK) return nullptr;
    *out = m.derived.back().as<float>();
    return *out;
}
static const float* transposed_of(ClipModel& m, const float* w, int rows, int cols, hipStream_t st) {
    float* d = nullptr;
    if (!derived_buf(m, (size_t)rows * cols, &d)) return nullptr;
    if (launch_transpose(w, d, rows, cols, st) != RLCF_OK) return nullptr;
    return d;
}

// build the train-form weights, the flat tunable vector (e->ln_params & co: the ABI of the LayerNorm path serves the BatchNorm
// parameters of a ResNet student unchanged) and the statistics vector; once per finalize
int engine_bn_enable(rlcf_engine* e, hipStream_t st) {
    ClipModel& m = e->model[RLCF_STUDENT];
    if (!m.finalized || !is_resnet(m.cfg)) { rlcf_set_error("BatchNorm tuning needs a finalized ModifiedResNet student"); return RLCF_ERR_STATE; }
    ResNetW& r = m.rn;
    if (r.bn_enabled) return RLCF_OK;
    if (prec_single(e)) { rlcf_set_error("BatchNorm tuning runs in RLCF_PREC_F32 / RLCF_PREC_F16X3 (RLCF_PREC_F16 is the prompt path's performance mode)"); return RLCF_ERR_STATE; }
    m.derived.reserve(m.derived.size() + 8 * (r.blocks.size() * 4 + 3) + 16);
    const rlcf_clip_cfg& c = m.cfg;
    const int w = c.vision_width;
    r.units.clear(); r.block_unit.clear();
    int pofs = 0, sofs = 0;
    auto add = [&](const std::string& conv, const std::string& bn, int cout, int cin, int k, bool tuned, bool need_dx) -> int {
        BnUnit u;
        const int kk = k * k;
        const float* wr = raw_of(m, conv + ".weight", (size_t)cout * cin * kk);
        u.gamma0 = raw_of(m, bn + ".weight", cout); u.beta0 = raw_of(m, bn + ".bias", cout);
        NEED(wr); NEED(u.gamma0); NEED(u.beta0);
        u.raw.cin = cin; u.raw.cout = cout; u.raw.k = k;
        u.raw.Kp = k == 1 ? cin : (kk * cin + 31) / 32 * 32;
        float* wp = nullptr;
        NEED(derived_buf(m, (size_t)cout * u.raw.Kp, &wp));
        conv_permute_kernel<<<dim3(cout), dim3(256), 0, st>>>(wr, wp, cin, kk, u.raw.Kp);
        RLCF_LAUNCH_CHECK();
        u.raw.w = wp; u.raw.b = nullptr;
        u.w_live = wr;
        TRY(engine_make_split(e, m, u.raw.w, (size_t)cout * u.raw.Kp, st));
        if (need_dx) {
            if (k == 1) {
                u.KpT = cout;
                NEED(u.wT = transposed_of(m, wr, cout, cin, st));                 // [cout, cin] -> [cin, cout]
            } else {
                u.KpT = (9 * cout + 31) / 32 * 32;
                float* wt = nullptr;
                NEED(derived_buf(m, (size_t)cin * u.KpT, &wt));
                conv_flip_kernel<<<dim3(cin), dim3(256), 0, st>>>(wr, wt, cin, cout, u.KpT);
                RLCF_LAUNCH_CHECK();
                u.wT = wt;
            }
            u.wT_buf = (float*)u.wT;
            if (u.KpT % 32 == 0) TRY(engine_make_split(e, m, u.wT, (size_t)cin * u.KpT, st));
        }
        if (tuned) { u.pofs = pofs; pofs += 2 * cout; }
        u.sofs = sofs; sofs += 2 * cout;
        r.units.push_back(u);
        return RLCF_OK;
    };
    TRY(add("visual.conv1", "visual.bn1", w / 2, 3, 3, true, false));
    TRY(add("visual.conv2", "visual.bn2", w / 2, w / 2, 3, true, true));
    TRY(add("visual.conv3", "visual.bn3", w, w / 2, 3, true, true));
    int inpl = w;
    for (int s = 0; s < 4; ++s) {
        const int planes = w << s;
        for (int b = 0; b < c.vision_stages[s]; ++b) {
            const std::string p = "visual.layer" + std::to_string(s + 1) + "." + std::to_string(b) + ".";
            r.block_unit.push_back((int)r.units.size());
            TRY(add(p + "conv1", p + "bn1", planes, inpl, 1, true, true));
            TRY(add(p + "conv2", p + "bn2", planes, planes, 3, true, true));
            TRY(add(p + "conv3", p + "bn3", planes * 4, planes, 1, true, true));
            if (r.blocks[r.block_unit.size() - 1].has_down) TRY(add(p + "downsample.0", p + "downsample.1", planes * 4, inpl, 1, false, true));
            inpl = planes * 4;
        }
    }
    const int E = r.E, D = c.embed_dim;
    NEED(r.q_wT = transposed_of(m, r.q_w, E, E, st));
    NEED(r.kv_wT = transposed_of(m, r.kv_w, 2 * E, E, st));
    NEED(r.c_wT = transposed_of(m, r.c_w, D, E, st));
    r.q_wT_buf = (float*)r.q_wT; r.kv_wT_buf = (float*)r.kv_wT; r.c_wT_buf = (float*)r.c_wT;
    TRY(engine_make_split(e, m, r.q_wT, (size_t)E * E, st));
    TRY(engine_make_split(e, m, r.kv_wT, (size_t)2 * E * E, st));
    TRY(engine_make_split(e, m, r.c_wT, (size_t)D * E, st));
    // flat vectors: tunable (gamma | beta per tuned unit, named_parameters order = execution order) and statistics (mean | var per unit)
    e->ln_count = pofs;
    r.n_stats = sofs;
    const size_t nb = (size_t)pofs * sizeof(float), sb = (size_t)sofs * sizeof(float);
    for (DevBuf* d : {&e->ln_params, &e->ln_init, &e->ln_grad, &e->ln_m, &e->ln_v, &e->ln_clip, &e->ln_mom}) TRY(d->ensure(nb));
    TRY(e->bn_stats.ensure(sb)); TRY(e->bn_stats_init.ensure(sb));
    size_t ui = 0;
    auto stat_names = [&](size_t i) -> std::string {
        if (i < 3) return "visual.bn" + std::to_string(i + 1);
        size_t k = 3;
        for (size_t bi = 0; bi < r.blocks.size(); ++bi) {
            const size_t nu = r.blocks[bi].has_down ? 4 : 3;
            if (i < k + nu) {
                int s = 0, b = (int)bi;
                while (b >= c.vision_stages[s]) { b -= c.vision_stages[s]; ++s; }
                const std::string p = "visual.layer" + std::to_string(s + 1) + "." + std::to_string(b) + ".";
                const size_t j = i - k;
                return j < 3 ? p + "bn" + std::to_string(j + 1) : p + "downsample.1";
            }
            k += nu;
        }
        return "";
    };
    for (; ui < r.units.size(); ++ui) {
        const BnUnit& u = r.units[ui];
        const int co = u.raw.cout;
        if (u.pofs >= 0) {
            RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.as<float>() + u.pofs, u.gamma0, co * sizeof(float), hipMemcpyDeviceToDevice, st));
            RLCF_HIP_CHECK(hipMemcpyAsync(e->ln_params.as<float>() + u.pofs + co, u.beta0, co * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        const std::string bn = stat_names(ui);
        const float *rm = raw_of(m, bn + ".running_mean", co), *rv = raw_of(m, bn + ".running_var", co);
        NEED(rm); NEED(rv);
        RLCF_HIP_CHECK(hipMemcpyAsync(e->bn_stats.as<float>() + u.sofs, rm, co * sizeof(float), hipMemcpyDeviceToDevice, st));
        RLCF_HIP_CHECK(hipMemcpyAsync(e->bn_stats.as<float>() + u.sofs + co, rv, co * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (DevBuf* d : {&e->ln_init, &e->ln_clip, &e->ln_mom}) RLCF_HIP_CHECK(hipMemcpyAsync(d->p, e->ln_params.p, nb, hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(e->bn_stats_init.p, e->bn_stats.p, sb, hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    r.bn_enabled = true;
    return RLCF_OK;
}

// ---- train-form pass -----------------------------------------------------------------------------------------------------------
struct BnPass {
    rlcf_engine* e; ClipModel* m; hipStream_t st; int n; int mode; float prior; bool save;
    float* vgrad = nullptr;          // every-parameter tuning: e->vw_grad (convolution / downsample.1 / attention-pool gradients)
};
static inline const float* bn_gamma(const rlcf_engine* e, const BnUnit& u) { return u.pofs >= 0 ? e->ln_params.as<float>() + u.pofs : u.gamma0; }
static inline const float* bn_beta(const rlcf_engine* e, const BnUnit& u) { return u.pofs >= 0 ? e->ln_params.as<float>() + u.pofs + u.raw.cout : u.beta0; }

// conv (unfolded) -> batch statistics -> normalise (+ identity) (+ ReLU).  z / y: where the GEMM output and the unit's output go
// in_unit: the unit whose output (or an average pool of it: same bound) `in` is, -1 = unknown range (the images)
static int bn_unit_fwd(const BnPass& P, int ui, const float* in, int in_unit, int H, int W, int stride, bool nchw, const float* idn, bool relu,
                       float* z, float* y) {
    rlcf_engine* e = P.e;
    BnUnit& u = P.m->rn.units[ui];
    u.in_ptr = in; u.in_H = H; u.in_W = W; u.in_stride = stride; u.in_nchw = nchw;       // (weight gradients of every-parameter tuning)
    const int Ho = H / stride, Wo = W / stride, C = u.raw.cout;
    const long M = (long)P.n * Ho * Wo;
    TRY(conv(e, u.raw, in, in_unit >= 0 ? e->bn_amax.as<float>() + in_unit : nullptr, P.n, H, W, stride, nchw, nullptr, RLCF_EPI_NONE, z, nullptr, P.st));
    const int chunks = (int)((M + BN_ROWS - 1) / BN_ROWS);
    TRY(e->bn_grad_c.ensure((size_t)chunks * 2 * C * sizeof(float)));
    float* ms = e->bn_ms[ui];
    float* st_ = e->bn_stats.as<float>() + u.sofs;
    if (P.mode != 0) {
        bn_stats_partial_kernel<<<dim3((C + 63) / 64, chunks), dim3(256), 0, P.st>>>(z, e->bn_grad_c.as<float>(), M, C);
        RLCF_LAUNCH_CHECK();
    }
    bn_stats_final_kernel<<<dim3((C + 63) / 64), dim3(1024), 0, P.st>>>(e->bn_grad_c.as<float>(), chunks, M, C, st_, st_ + C, ms, P.mode, P.prior);
    RLCF_LAUNCH_CHECK();
    const long total4 = M * C / 4;
    bn_apply_kernel<<<grid_amax(total4), dim3(256), 0, P.st>>>(z, ms, bn_gamma(e, u), bn_beta(e, u), idn, y, total4, C, relu ? 1 : 0,
                                                              e->bn_amax.as<unsigned int>() + ui);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// activation sizes of unit outputs for n images (floats), in unit order; also the spatial size every unit's OUTPUT has
static void bn_plan(const ClipModel& m, std::vector<size_t>& out_elems, std::vector<int>& out_hw) {
    const ResNetW& r = m.rn;
    const int R = m.cfg.image_resolution;
    out_elems.clear(); out_hw.clear();
    int H = R / 2;
    for (int i = 0; i < 3; ++i) { out_elems.push_back((size_t)H * H * r.units[i].raw.cout); out_hw.push_back(H); }
    H /= 2;
    for (size_t b = 0; b < r.blocks.size(); ++b) {
        const BottleW& bw = r.blocks[b];
        const int u0 = r.block_unit[b], Ho = H / bw.stride;
        out_elems.push_back((size_t)H * H * r.units[u0].raw.cout); out_hw.push_back(H);
        out_elems.push_back((size_t)H * H * r.units[u0 + 1].raw.cout); out_hw.push_back(H);
        out_elems.push_back((size_t)Ho * Ho * r.units[u0 + 2].raw.cout); out_hw.push_back(Ho);
        if (bw.has_down) { out_elems.push_back((size_t)Ho * Ho * r.units[u0 + 3].raw.cout); out_hw.push_back(Ho); }
        H = Ho;
    }
}

// ModifiedResNet.forward in train form over n images (ALL of them in one pass: the batch statistics couple them) + L2 normalisation.
// z / y of every unit, the pooled intermediates and the attention pool's tensors are kept for rn_backward_bn (every unit writes its
// own slots: no in-place reuse to reason about; RN50 at 64 views of 224^2: 5 GB, RN50x64 at 32 views of 448^2: ~55 GB of the 288).
int rn_forward_train(rlcf_engine* e, ClipModel& m, const float* images, int n, float* feats, hipStream_t st, int mode_override) {
    const bool save = true;
    const rlcf_clip_cfg& c = m.cfg;
    ResNetW& r = m.rn;
    const int R = c.image_resolution, w = c.vision_width, E = r.E, D = c.embed_dim, HW = r.out_hw * r.out_hw, T = HW + 1;
    const int nu = (int)r.units.size();
    TRY(rn_ensure(e, r, n, T));
    // statistics scratch (mean | rstd per unit) and, when saving, z / y of every unit
    std::vector<size_t> oe; std::vector<int> ohw;
    bn_plan(m, oe, ohw);
    size_t ms_total = 0, act_total = 0;
    for (int i = 0; i < nu; ++i) { ms_total += 2 * (size_t)r.units[i].raw.cout; act_total += oe[i]; }
    TRY(e->bn_scratch.ensure(ms_total * sizeof(float)));
    e->bn_ms.resize(nu); e->bn_z.resize(nu); e->bn_y.resize(nu);
    { float* p = e->bn_scratch.as<float>(); for (int i = 0; i < nu; ++i) { e->bn_ms[i] = p; p += 2 * (size_t)r.units[i].raw.cout; } }
    {
        // + pooled copies: stem pool, per strided block the pooled conv2 output and the pooled block input
        size_t extra = (size_t)(R / 4) * (R / 4) * w + (size_t)T * 3 * E + 2 * E;    // (+ the attention pool's k|v, q, tokens and output)
        { int H = R / 4; for (const BottleW& b : r.blocks) { if (b.stride > 1) extra += (size_t)(H / 2) * (H / 2) * (b.c2.cout + b.down.cin); H /= b.stride; } }
        TRY(e->bn_saved.ensure(((size_t)n * (2 * act_total + extra)) * sizeof(float)));
        float* p = e->bn_saved.as<float>();
        for (int i = 0; i < nu; ++i) { e->bn_z[i] = p; p += (size_t)n * oe[i]; e->bn_y[i] = p; p += (size_t)n * oe[i]; }
        e->bn_saved_n = n;
    }
    // mode_override 0: every BatchNorm in EVAL form on the running statistics (the final inference of every-parameter tuning: there
    // CLIPCLS_TTA.train(mode) is plain nn.Module.train(mode), custom_clip.py:487-497, and the harness calls model.eval() first)
    const int mode = mode_override >= 0 ? mode_override : (e->bn_prior_strength >= 0 ? 2 : 1);
    const float prior = e->bn_prior_strength >= 0 ? (float)e->bn_prior_strength / (float)(e->bn_prior_strength + 1) : 0.f;
    BnPass P{e, &m, st, n, mode, prior, save};
    float* pooled = e->bn_saved.as<float>() + (size_t)n * 2 * act_total;
    int H = R / 2;
    auto Z = [&](int ui) { return e->bn_z[ui]; };
    auto Y = [&](int ui) { return e->bn_y[ui]; };
    TRY(e->bn_amax.ensure((size_t)2 * nu * sizeof(float)));
    RLCF_HIP_CHECK(hipMemsetAsync(e->bn_amax.p, 0, (size_t)nu * sizeof(float), st));
    TRY(bn_unit_fwd(P, 0, images, -1, R, R, 2, true, nullptr, true, Z(0), Y(0)));
    TRY(bn_unit_fwd(P, 1, Y(0), 0, H, H, 1, false, nullptr, true, Z(1), Y(1)));
    TRY(bn_unit_fwd(P, 2, Y(1), 1, H, H, 1, false, nullptr, true, Z(2), Y(2)));
    H /= 2;
    float* xin = pooled;
    pooled += (size_t)n * H * H * w;
    TRY(avgpool2(Y(2), xin, n, H, H, w, st));
    const float* x = xin;                                          // block input
    int xu = 2;                                                    // ... and the unit it came from
    for (size_t bi = 0; bi < r.blocks.size(); ++bi) {
        const BottleW& b = r.blocks[bi];
        const int u0 = r.block_unit[bi], planes = b.c1.cout, Ho = H / b.stride;
        TRY(bn_unit_fwd(P, u0, x, xu, H, H, 1, false, nullptr, true, Z(u0), Y(u0)));
        TRY(bn_unit_fwd(P, u0 + 1, Y(u0), u0, H, H, 1, false, nullptr, true, Z(u0 + 1), Y(u0 + 1)));
        const float* t2 = Y(u0 + 1);
        const float* xp = x;
        if (b.stride > 1) {
            float* t2p = pooled;
            pooled += (size_t)n * Ho * Ho * planes;
            TRY(avgpool2(t2, t2p, n, Ho, Ho, planes, st));
            t2 = t2p;
            float* xpp = pooled;
            pooled += (size_t)n * Ho * Ho * b.down.cin;
            TRY(avgpool2(x, xpp, n, Ho, Ho, b.down.cin, st));
            xp = xpp;
        }
        const float* idn = x;
        if (b.has_down) {
            TRY(bn_unit_fwd(P, u0 + 3, xp, xu, Ho, Ho, 1, false, nullptr, false, Z(u0 + 3), Y(u0 + 3)));
            idn = Y(u0 + 3);
        }
        TRY(bn_unit_fwd(P, u0 + 2, t2, u0 + 1, Ho, Ho, 1, false, idn, true, Z(u0 + 2), Y(u0 + 2)));
        x = Y(u0 + 2);
        xu = u0 + 2;
        H = Ho;
    }
    // attention pool (model.py:68-91), probabilities kept for the backward
    // tokens / q / k|v / probabilities / attended output stay with the saved pass (a ModifiedResNet reward model's encode, between this
    // pass and its backward, reuses e->rn_tok / rn_q / rn_kv / rn_att)
    e->bn_kv = pooled; pooled += (size_t)n * T * 2 * E;
    e->bn_q = pooled; pooled += (size_t)n * E;
    e->bn_tok = pooled; pooled += (size_t)n * T * E;
    e->bn_att = pooled;
    attnpool_tokens_kernel<<<dim3((E + 255) / 256, n), dim3(256), 0, st>>>(x, r.pos, e->bn_tok, HW, E);
    RLCF_LAUNCH_CHECK();
    TRY(engine_gemm(e, e->bn_tok, T * E, r.q_w, E, r.q_b, nullptr, 0, e->bn_q, E, n, E, E, RLCF_EPI_NONE, st));
    TRY(engine_gemm(e, e->bn_tok, E, r.kv_w, E, r.kv_b, nullptr, 0, e->bn_kv, 2 * E, n * T, 2 * E, E, RLCF_EPI_NONE, st));
    TRY(e->bn_grad_b.ensure((size_t)n * r.heads * T * sizeof(float)));
    attnpool_attend_save_kernel<<<dim3(r.heads, n), dim3(256), (T + 256) * sizeof(float), st>>>(e->bn_q, e->bn_kv, e->bn_att,
                                                                                                e->bn_grad_b.as<float>(), T, E);
    RLCF_LAUNCH_CHECK();
    TRY(engine_gemm(e, e->bn_att, E, r.c_w, E, r.c_b, nullptr, 0, e->feat_raw.as<float>(), D, n, D, E, RLCF_EPI_NONE, st));
    TRY(e->vit_inv_norm.ensure((size_t)std::max(n, e->max_views) * sizeof(float)));
    TRY(launch_l2norm_rows(e->feat_raw.as<float>(), feats, e->vit_inv_norm.as<float>(), n, D, st));
    return RLCF_OK;
}

// one BatchNorm's backward: dy = gradient at the unit's output (before its ReLU mask), -> dz (gradient at the GEMM output), the
// parameter gradients (tuned units) and, optionally, gout = the masked gradient (what also flows into a block's identity branch)
static int bn_unit_bwd(const BnPass& P, int ui, const float* dy, long M, bool relu, float* dz, float* gout, float* bn_grad) {
    rlcf_engine* e = P.e;
    const BnUnit& u = P.m->rn.units[ui];
    const int C = u.raw.cout;
    const int chunks = (int)((M + BN_ROWS - 1) / BN_ROWS);
    TRY(e->bn_grad_c.ensure((size_t)chunks * 2 * C * sizeof(float)));
    TRY(e->bn_grad_a.ensure((size_t)2 * C * sizeof(float)));
    const float *z = e->bn_z[ui], *y = e->bn_y[ui], *ms = e->bn_ms[ui];
    bn_bwd_partial_kernel<<<dim3((C + 63) / 64, chunks), dim3(256), 0, P.st>>>(dy, y, z, ms, e->bn_grad_c.as<float>(), M, C, relu ? 1 : 0);
    RLCF_LAUNCH_CHECK();
    float* dg = (bn_grad && u.pofs >= 0) ? bn_grad + u.pofs : nullptr;
    if (u.pofs < 0 && P.vgrad && u.vofs_g >= 0) dg = P.vgrad + u.vofs_g;        // downsample.1 (every-parameter tuning): weight | bias in e->vw_grad
    bn_bwd_final_kernel<<<dim3((C + 63) / 64), dim3(1024), 0, P.st>>>(e->bn_grad_c.as<float>(), chunks, C, e->bn_grad_a.as<float>(), dg, dg ? dg + C : nullptr);
    RLCF_LAUNCH_CHECK();
    if (dz) {
        const long total = M * C;
        bn_bwd_apply_kernel<<<grid_amax(total), dim3(256), 0, P.st>>>(dy, y, z, ms, bn_gamma(e, u), e->bn_grad_a.as<float>(), dz, gout, total, M, C,
                                                                     relu ? 1 : 0, P.mode == 1 ? 1 : 0,
                                                                     e->bn_amax.as<unsigned int>() + P.m->rn.units.size() + ui);
        RLCF_LAUNCH_CHECK();
    }
    return RLCF_OK;
}
// dX of a convolution unit: dz [M, cout] -> dx [M, cin] (stride-1 units only); res (optional) is added.  A 1x1 convolution's dX is
// the 1x1 convolution with the transposed weight, a 3x3 (pad 1) one's the 3x3 convolution with the flipped, channel-transposed kernel:
// both go through conv(), the operand scale from max|dz| (bn_bwd_apply_kernel left it in the unit's slot)
static int bn_conv_dx(const BnPass& P, int ui, const float* dz, int H, int W, const float* res, float* dx) {
    rlcf_engine* e = P.e;
    const BnUnit& u = P.m->rn.units[ui];
    ConvW t{};
    t.w = u.wT; t.b = nullptr; t.cin = u.raw.cout; t.cout = u.raw.cin; t.k = u.raw.k; t.Kp = u.KpT;
    if (t.k == 3) {
        const size_t total = (size_t)P.n * H * W * t.Kp;
        TRY(e->rn_col.ensure(total * sizeof(float)));
        if (prec_x3(e) && total > e->a_split_elems) { TRY(e->a_hi.ensure(total * 4)); e->a_split_elems = total; }
    }
    return conv(e, t, dz, e->bn_amax.as<float>() + P.m->rn.units.size() + ui, P.n, H, W, 1, false, res, RLCF_EPI_NONE, dx, nullptr, P.st);
}
static int avgpool2_bwd(const float* dout, float* din, int n, int Ho, int Wo, int C, hipStream_t st) {
    const long total = (long)n * 4 * Ho * Wo * C;
    avgpool2_bwd_kernel<<<grid_for(total), dim3(256), 0, st>>>(dout, din, total, C, Ho, Wo);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ---- every-parameter tuning of a ModifiedResNet student: weight gradients ------------------------------------------------------------
// GEMM-layout gradient [cout, Kp] ((ky, kx, ci) order) -> the state-dict layout [cout, cin, k, k]
__global__ void conv_unpermute_kernel(const float* __restrict__ g, float* __restrict__ out, int Cin, int kk, int Kp) {
    const int co = blockIdx.x;
    for (int i = threadIdx.x; i < kk * Cin; i += blockDim.x) {
        const int tap = i / Cin, ci = i - tap * Cin;
        out[((size_t)co * Cin + ci) * kk + tap] = g[(size_t)co * Kp + i];
    }
}
// d positional_embedding[t, c] = sum over the images of dtok[img, t, c] (fixed order)
__global__ void attnpool_dpos_kernel(const float* __restrict__ dtok, float* __restrict__ dpos, int n, long TE) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < TE; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < n; ++b) s += dtok[(size_t)b * TE + i];
        dpos[i] = s;
    }
}
// d conv.weight of unit ui = dZ^T . patches(input the forward fed it): dz [M, cout] at the unit's OUTPUT resolution
static int rn_conv_wgrad(const BnPass& P, int ui, const float* dz, long M) {
    rlcf_engine* e = P.e;
    const BnUnit& u = P.m->rn.units[ui];
    if (!P.vgrad || u.vofs_w < 0) return RLCF_OK;
    const int cout = u.raw.cout, cin = u.raw.cin, Kp = u.raw.Kp;
    float* dst = P.vgrad + u.vofs_w;
    if (u.raw.k == 1) return engine_wgrad(e, dz, cout, cout, u.in_ptr, cin, cin, (int)M, dst, nullptr, P.st);     // [cout, cin] as stored
    const int H = u.in_H, W = u.in_W, Ho = H / u.in_stride, Wo = W / u.in_stride;
    const long total = M * Kp;
    TRY(e->rn_col.ensure((size_t)total * sizeof(float)));
    const long sN = (long)cin * H * W, sC = u.in_nchw ? (long)H * W : 1, sH = u.in_nchw ? W : (long)W * cin, sW = u.in_nchw ? 1 : cin;
    im2col3x3_kernel<<<grid_for(total), dim3(256), 0, P.st>>>(u.in_ptr, e->rn_col.as<float>(), total, cin, H, W, Ho, Wo, u.in_stride, Kp, sN, sC, sH, sW);
    RLCF_LAUNCH_CHECK();
    TRY(e->rn_wg_tmp.ensure((size_t)cout * Kp * sizeof(float)));
    float* tmp = e->rn_wg_tmp.as<float>();
    TRY(engine_wgrad(e, dz, cout, cout, e->rn_col.as<float>(), Kp, Kp, (int)M, tmp, nullptr, P.st));
    conv_unpermute_kernel<<<dim3(cout), dim3(256), 0, P.st>>>(tmp, dst, cin, 9, Kp);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// d loss / d (BatchNorm weights, biases) of the pass rn_forward_train just ran over n images, from dfeat [n, D] = d loss / d (the
// L2-normalised features).  bn_grad: [e->ln_count] in the tunable vector's layout.
int rn_backward_bn(rlcf_engine* e, ClipModel& m, int n, const float* feats, float* dfeat, float* bn_grad, hipStream_t st, float* vgrad) {
    const rlcf_clip_cfg& c = m.cfg;
    ResNetW& r = m.rn;
    const int R = c.image_resolution, E = r.E, D = c.embed_dim, HW = r.out_hw * r.out_hw, T = HW + 1;
    if (e->bn_saved_n != n) { rlcf_set_error("rn_backward_bn: the saved pass holds %d images, not %d", e->bn_saved_n, n); return RLCF_ERR_STATE; }
    const int mode = e->bn_prior_strength >= 0 ? 2 : 1;
    BnPass P{e, &m, st, n, mode, 0.f, true};
    P.vgrad = vgrad;
    RLCF_HIP_CHECK(hipMemsetAsync(e->bn_amax.as<float>() + r.units.size(), 0, r.units.size() * sizeof(float), st));
    float *G0 = e->rn_buf[0].as<float>(), *G1 = e->rn_buf[1].as<float>(), *G2 = e->rn_buf[2].as<float>(), *G3 = e->rn_buf[3].as<float>(),
          *G4 = e->rn_buf[4].as<float>();
    // features -> attention pool
    TRY(launch_l2norm_bwd(feats, dfeat, e->vit_inv_norm.as<float>(), dfeat, n, D, st));
    if (vgrad) TRY(engine_wgrad(e, dfeat, D, D, e->bn_att, E, E, n, vgrad + r.vofs_pool[7], vgrad + r.vofs_pool[8], st));      // c_proj.weight / .bias
    TRY(engine_gemm(e, dfeat, D, r.c_wT, D, nullptr, nullptr, 0, G0, E, n, E, D, RLCF_EPI_NONE, st));                 // d att [n, E]
    float *dq = G1, *dkv = G2;
    attnpool_attend_bwd_kernel<<<dim3(r.heads, n), dim3(256), (T + 256) * sizeof(float), st>>>(e->bn_q, e->bn_kv, e->bn_grad_b.as<float>(), G0, dq,
                                                                                               dkv, T, E);
    RLCF_LAUNCH_CHECK();
    float* dtok = G3;
    if (vgrad) {
        // q_proj (its input: token 0 of every image, row stride T*E); k_proj | v_proj as ONE product on the concatenated gradient, then
        // the halves go to their own slots
        TRY(engine_wgrad(e, dq, E, E, e->bn_tok, T * E, E, n, vgrad + r.vofs_pool[3], vgrad + r.vofs_pool[4], st));
        TRY(e->rn_wg_tmp.ensure(((size_t)2 * E * E + 2 * E) * sizeof(float)));
        float *dwkv = e->rn_wg_tmp.as<float>(), *dbkv = dwkv + (size_t)2 * E * E;
        RLCF_HIP_CHECK(hipMemsetAsync(dbkv, 0, 2 * E * sizeof(float), st));
        TRY(engine_wgrad(e, dkv, 2 * E, 2 * E, e->bn_tok, E, E, n * T, dwkv, dbkv, st));
        RLCF_HIP_CHECK(hipMemcpyAsync(vgrad + r.vofs_pool[1], dwkv, (size_t)E * E * sizeof(float), hipMemcpyDeviceToDevice, st));
        RLCF_HIP_CHECK(hipMemcpyAsync(vgrad + r.vofs_pool[5], dwkv + (size_t)E * E, (size_t)E * E * sizeof(float), hipMemcpyDeviceToDevice, st));
        RLCF_HIP_CHECK(hipMemcpyAsync(vgrad + r.vofs_pool[2], dbkv, E * sizeof(float), hipMemcpyDeviceToDevice, st));
        RLCF_HIP_CHECK(hipMemcpyAsync(vgrad + r.vofs_pool[6], dbkv + E, E * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    TRY(engine_gemm(e, dkv, 2 * E, r.kv_wT, 2 * E, nullptr, nullptr, 0, dtok, E, n * T, E, 2 * E, RLCF_EPI_NONE, st));
    TRY(engine_gemm(e, dq, E, r.q_wT, E, nullptr, nullptr, 0, G0, E, n, E, E, RLCF_EPI_NONE, st));                    // d tok0 through q_proj
    { const long tot = (long)n * E; attnpool_add_row0_kernel<<<grid_for(tot), dim3(256), 0, st>>>(dtok, G0, T, E, tot); RLCF_LAUNCH_CHECK(); }
    if (vgrad) { const long TE = (long)T * E; attnpool_dpos_kernel<<<grid_for(TE), dim3(256), 0, st>>>(dtok, vgrad + r.vofs_pool[0], n, TE); RLCF_LAUNCH_CHECK(); }
    float* dOut = G4;                                                  // gradient at the last block's output [n*HW, E]
    { const long tot = (long)n * HW * E; attnpool_tokens_bwd_kernel<<<grid_for(tot), dim3(256), 0, st>>>(dtok, dOut, HW, E, tot); RLCF_LAUNCH_CHECK(); }
    // spatial size of every block's input
    std::vector<int> Hin(r.blocks.size());
    { int H = R / 4; for (size_t b = 0; b < r.blocks.size(); ++b) { Hin[b] = H; H /= r.blocks[b].stride; } }
    // buffers rotate: dOut lives in G4 at entry of every block iteration
    for (int bi = (int)r.blocks.size() - 1; bi >= 0; --bi) {
        const BottleW& b = r.blocks[bi];
        const int u0 = r.block_unit[bi], H = Hin[bi], Ho = H / b.stride, planes = b.c1.cout, inpl = b.c1.cin;
        const long Mo = (long)n * Ho * Ho, Mi = (long)n * H * H;
        // conv3 + bn3 (+ identity) + ReLU: dz3 -> G0, masked gradient g -> G1 (identity branch)
        TRY(bn_unit_bwd(P, u0 + 2, dOut, Mo, true, G0, G1, bn_grad));
        TRY(rn_conv_wgrad(P, u0 + 2, G0, Mo));
        TRY(bn_conv_dx(P, u0 + 2, G0, Ho, Ho, nullptr, G2));                                   // d t2 [Mo, planes]
        const float* dB = G2;
        if (b.stride > 1) { TRY(avgpool2_bwd(G2, G0, n, Ho, Ho, planes, st)); dB = G0; }       // -> [Mi, planes]
        // conv2 + bn2 + ReLU
        float* dz2 = dB == G0 ? G2 : G0;
        TRY(bn_unit_bwd(P, u0 + 1, dB, Mi, true, dz2, nullptr, bn_grad));
        TRY(rn_conv_wgrad(P, u0 + 1, dz2, Mi));
        float* dA = dz2 == G0 ? G2 : G0;
        TRY(bn_conv_dx(P, u0 + 1, dz2, H, H, nullptr, dA));                                   // [Mi, planes]
        // conv1 + bn1 + ReLU
        float* dz1 = dA == G0 ? G2 : G0;
        TRY(bn_unit_bwd(P, u0, dA, Mi, true, dz1, nullptr, bn_grad));
        TRY(rn_conv_wgrad(P, u0, dz1, Mi));
        // identity branch into G3 [Mi, inpl], then the main branch adds onto it (GEMM residual) -> new dOut in G4
        const float* idg = G1;                                      // g [Mo, 4 planes]
        if (b.has_down) {
            float* dzd = dz1 == G0 ? G2 : G0;                       // (the buffer dz1 does not use)
            TRY(bn_unit_bwd(P, u0 + 3, G1, Mo, false, dzd, nullptr, nullptr));
            TRY(rn_conv_wgrad(P, u0 + 3, dzd, Mo));
            if (b.stride > 1) {
                TRY(bn_conv_dx(P, u0 + 3, dzd, Ho, Ho, nullptr, G1));                         // [Mo, inpl] (G1's g is consumed)
                TRY(avgpool2_bwd(G1, G3, n, Ho, Ho, inpl, st));                               // [Mi, inpl]
            } else TRY(bn_conv_dx(P, u0 + 3, dzd, Ho, Ho, nullptr, G3));
            idg = G3;
        }
        TRY(bn_conv_dx(P, u0, dz1, H, H, idg, G4));                                           // d x = main + identity
        dOut = G4;
    }
    // stem: avgpool <- conv3/bn3/relu <- conv2/bn2/relu <- conv1/bn1/relu (no dX beyond conv1's BatchNorm)
    const int w = c.vision_width, Hs = R / 2;
    const long Ms = (long)n * Hs * Hs;
    TRY(avgpool2_bwd(dOut, G0, n, R / 4, R / 4, w, st));
    TRY(bn_unit_bwd(P, 2, G0, Ms, true, G1, nullptr, bn_grad));
    TRY(rn_conv_wgrad(P, 2, G1, Ms));
    TRY(bn_conv_dx(P, 2, G1, Hs, Hs, nullptr, G2));
    TRY(bn_unit_bwd(P, 1, G2, Ms, true, G0, nullptr, bn_grad));
    TRY(rn_conv_wgrad(P, 1, G0, Ms));
    TRY(bn_conv_dx(P, 1, G0, Hs, Hs, nullptr, G1));
    // the stem's first convolution: only its BatchNorm gradient is needed for norm-layer tuning; its weight gradient needs dz as well
    TRY(bn_unit_bwd(P, 0, G1, Ms, true, vgrad ? G2 : nullptr, nullptr, bn_grad));
    if (vgrad) TRY(rn_conv_wgrad(P, 0, G2, Ms));
    return RLCF_OK;
}


// ---- every-parameter tuning of a ModifiedResNet student -------------------------------------------------------------------------------
// CLIPCLS_TTA(only_norm=False) with `--arch RN50` — the PARSER DEFAULTS of tune_cls_rl.py (TPT/params.py:23,73; tune_cls_rl.py:67-71;
// parameters() = clip_model.visual.parameters(), custom_clip.py:477-479).  The BatchNorm tensors whose name contains 'bn' stay in
// e->ln_params (the norm-layer path's vector); every other visual tensor — the convolution weights as stored [cout, cin, k, k],
// downsample.1's weight / bias, the attention pool — moves into the flat buffer e->vw in named_parameters order, so AdamW is two
// launches and reset() two copies, as for a VisionTransformer student (engine_visual_enable).  After every optimizer step / reset the
// derived forms the passes read (GEMM-layout weights, flipped / transposed dX operands, k|v concatenations, split-f16 pairs) are
// rebuilt from the live tensors: rn_visual_refresh.
static int resplit(rlcf_engine* e, ClipModel& m, const float* w, size_t numel, hipStream_t st, bool at_checkpoint) {
    auto it = m.split_of.find(w);
    if (it == m.split_of.end()) return RLCF_OK;
    // a weight that is re-split is a TUNED weight: off the fp16 grid after its first step, back on it when a reset has put the checkpoint's
    // values back (hi_only is still their copy)
    it->second.lo_zero = at_checkpoint && it->second.lo_zero_ckpt;
    const ClipModel::SplitW& sp = it->second;
    return launch_split_f16x2(w, sp.hi, sp.lo, (int64_t)numel, st, 1.0f / sp.inv_scale, sp.lo == (void*)((char*)sp.hi + 64) ? 1 : 0);
}
int rn_visual_refresh(rlcf_engine* e, hipStream_t st, bool at_checkpoint) {
    ClipModel& m = e->model[RLCF_STUDENT];
    ResNetW& r = m.rn;
    if (!r.full_enabled) return RLCF_OK;
    for (BnUnit& u : r.units) {
        const int cout = u.raw.cout, cin = u.raw.cin, kk = u.raw.k * u.raw.k;
        conv_permute_kernel<<<dim3(cout), dim3(256), 0, st>>>(u.w_live, (float*)u.raw.w, cin, kk, u.raw.Kp);
        RLCF_LAUNCH_CHECK();
        TRY(resplit(e, m, u.raw.w, (size_t)cout * u.raw.Kp, st, at_checkpoint));
        if (u.wT_buf) {
            if (u.raw.k == 1) TRY(launch_transpose(u.w_live, u.wT_buf, cout, cin, st));
            else { conv_flip_kernel<<<dim3(cin), dim3(256), 0, st>>>(u.w_live, u.wT_buf, cin, cout, u.KpT); RLCF_LAUNCH_CHECK(); }
            TRY(resplit(e, m, u.wT, (size_t)cin * u.KpT, st, at_checkpoint));
        }
    }
    const size_t E = r.E, D = m.cfg.embed_dim;
    const float* vw = e->vw.as<float>();
    RLCF_HIP_CHECK(hipMemcpyAsync(r.kv_w_buf, vw + r.vofs_pool[1], E * E * sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(r.kv_w_buf + E * E, vw + r.vofs_pool[5], E * E * sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(r.kv_b_buf, vw + r.vofs_pool[2], E * sizeof(float), hipMemcpyDeviceToDevice, st));
    RLCF_HIP_CHECK(hipMemcpyAsync(r.kv_b_buf + E, vw + r.vofs_pool[6], E * sizeof(float), hipMemcpyDeviceToDevice, st));
    TRY(resplit(e, m, r.kv_w, 2 * E * E, st, at_checkpoint));
    TRY(resplit(e, m, r.q_w, E * E, st, at_checkpoint));
    TRY(resplit(e, m, r.c_w, D * E, st, at_checkpoint));
    TRY(launch_transpose(r.q_w, r.q_wT_buf, (int)E, (int)E, st));
    TRY(launch_transpose(r.kv_w, r.kv_wT_buf, (int)(2 * E), (int)E, st));
    TRY(launch_transpose(r.c_w, r.c_wT_buf, (int)D, (int)E, st));
    TRY(resplit(e, m, r.q_wT, E * E, st, at_checkpoint));
    TRY(resplit(e, m, r.kv_wT, 2 * E * E, st, at_checkpoint));
    TRY(resplit(e, m, r.c_wT, D * E, st, at_checkpoint));
    return RLCF_OK;
}

int engine_rn_visual_enable(rlcf_engine* e, hipStream_t st) {
    ClipModel& m = e->model[RLCF_STUDENT];
    if (!m.finalized || !is_resnet(m.cfg)) { rlcf_set_error("every-parameter tuning: the student is not a finalized ModifiedResNet"); return RLCF_ERR_STATE; }
    ResNetW& r = m.rn;
    if (r.full_enabled) return RLCF_OK;
    TRY(engine_bn_enable(e, st));
    const size_t E = r.E, D = m.cfg.embed_dim, T = (size_t)r.out_hw * r.out_hw + 1;
    // slots in named_parameters order of clip_model.visual, the 'bn' tensors (e->ln_params) left out
    size_t total = 0;
    e->vw_slots.clear();
    auto slot = [&](size_t numel) -> long { const size_t off = total; e->vw_slots.push_back(VwSlot{off, numel}); total += (numel + 63) / 64 * 64; return (long)off; };
    // (downsample.1's weight | bias: the BatchNorm backward writes d gamma at vofs_g and d beta right behind it, at vofs_g + cout, so
    // the two slots are adjacent WITHOUT padding between them)
    total = 0; e->vw_slots.clear();
    for (BnUnit& u : r.units) {
        u.vofs_w = slot((size_t)u.raw.cout * u.raw.cin * u.raw.k * u.raw.k);
        if (u.pofs < 0) {
            const size_t off = total;
            e->vw_slots.push_back(VwSlot{off, (size_t)u.raw.cout});
            e->vw_slots.push_back(VwSlot{off + u.raw.cout, (size_t)u.raw.cout});
            total += ((size_t)2 * u.raw.cout + 63) / 64 * 64;
            u.vofs_g = (long)off;
        }
    }
    const size_t pool_numel[9] = {T * E, E * E, E, E * E, E, E * E, E, D * E, D};
    for (int i = 0; i < 9; ++i) r.vofs_pool[i] = slot(pool_numel[i]);
    const size_t nb = total * sizeof(float);
    for (DevBuf* d : {&e->vw, &e->vw_init, &e->vw_grad, &e->vw_m, &e->vw_v, &e->vw_clip, &e->vw_mom}) TRY(d->ensure(nb));
    RLCF_HIP_CHECK(hipMemsetAsync(e->vw.p, 0, nb, st));
    float* vw = e->vw.as<float>();
    auto take = [&](const float* src, long off, size_t numel) -> const float* {
        (void)hipMemcpyAsync(vw + off, src, numel * sizeof(float), hipMemcpyDeviceToDevice, st);
        return vw + off;
    };
    for (BnUnit& u : r.units) {
        u.w_live = take(u.w_live, u.vofs_w, (size_t)u.raw.cout * u.raw.cin * u.raw.k * u.raw.k);
        if (u.pofs < 0) { u.gamma0 = take(u.gamma0, u.vofs_g, u.raw.cout); u.beta0 = take(u.beta0, u.vofs_g + u.raw.cout, u.raw.cout); }
    }
    // attention pool: positional embedding, q / c projections are read where they live; k | v keep their concatenated copies
    const float *kw = raw_of(m, "visual.attnpool.k_proj.weight", E * E), *kb = raw_of(m, "visual.attnpool.k_proj.bias", E);
    const float *vwt = raw_of(m, "visual.attnpool.v_proj.weight", E * E), *vb = raw_of(m, "visual.attnpool.v_proj.bias", E);
    NEED(kw); NEED(kb); NEED(vwt); NEED(vb);
    auto move_split = [&](const float* old, const float* now) {        // the split copy follows the tensor to its new address
        auto sp = m.split_of.find(old);
        if (sp != m.split_of.end()) { const ClipModel::SplitW s_ = sp->second; m.split_of.erase(sp); m.split_of[now] = s_; }
    };
    { const float* n_ = take(r.pos, r.vofs_pool[0], T * E); r.pos = n_; }
    take(kw, r.vofs_pool[1], E * E); take(kb, r.vofs_pool[2], E);
    { const float* n_ = take(r.q_w, r.vofs_pool[3], E * E); move_split(r.q_w, n_); r.q_w = n_; }
    { const float* n_ = take(r.q_b, r.vofs_pool[4], E); r.q_b = n_; }
    take(vwt, r.vofs_pool[5], E * E); take(vb, r.vofs_pool[6], E);
    { const float* n_ = take(r.c_w, r.vofs_pool[7], D * E); move_split(r.c_w, n_); r.c_w = n_; }
    { const float* n_ = take(r.c_b, r.vofs_pool[8], D); r.c_b = n_; }
    r.kv_w_buf = (float*)r.kv_w; r.kv_b_buf = (float*)r.kv_b;
    for (DevBuf* d : {&e->vw_init, &e->vw_clip, &e->vw_mom}) RLCF_HIP_CHECK(hipMemcpyAsync(d->p, e->vw.p, nb, hipMemcpyDeviceToDevice, st));
    e->vw_count = total;
    e->vw_dirty = false;
    r.full_enabled = true;
    e->vw_init_is_ckpt = true;
    RLCF_HIP_CHECK(hipStreamSynchronize(st));
    return RLCF_OK;
}
