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
    return engine_make_split(e, m, out.w, (size_t)cout * out.Kp, st);
}

// build_model's ResNet branch (model.py:408-412) fixed the geometry; here: fold, permute and (F16X3) split every weight
int resnet_finalize(rlcf_engine* e, ClipModel& m, hipStream_t st) {
    const rlcf_clip_cfg& c = m.cfg;
    ResNetW& r = m.rn;
    const int w = c.vision_width;
    r.blocks.clear();
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

static inline dim3 grid_for(long total) { return dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)); }

// conv (+folded bn) (+identity) (+relu) on an NHWC activation; 3x3 goes through the patch matrix
// in_amax / out_amax: device scalars holding max|in| (if known) and receiving max|out| (GEMM epilogue), so that the split-f16
// operand scale of the next convolution needs no extra pass over the activation
static int conv(rlcf_engine* e, const ConvW& cw, const float* in, const float* in_amax, int n, int H, int W, int stride, bool nchw,
                const float* res, int epi, float* out, float* out_amax, hipStream_t st) {
    const int Ho = H / stride, Wo = W / stride;
    const long M = (long)n * Ho * Wo;
    const float* A = in;
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
        return engine_gemm_presplit(e, cw.w, cw.b, res, cw.cout, out, cw.cout, (int)M, cw.cout, cw.Kp, epi, e->dyn.as<float>() + 2, st, out_amax);
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
        // max|.| of the five activation buffers (X, A, B, Cb, Dd), refreshed by whichever GEMM writes the buffer; an
        // average-pooled copy inherits the scalar of its source (a valid, slightly conservative bound)
        TRY(e->rn_amax.ensure(8 * sizeof(float)));
        float* am = e->rn_amax.as<float>();
        float *amX = am, *amA = am + 1, *amB = am + 2, *amD = am + 4;
#define ZERO(p) RLCF_HIP_CHECK(hipMemsetAsync((p), 0, sizeof(float), st))
        // stem (model.py:139-145): conv 3x3 s2 -> conv 3x3 -> conv 3x3 -> avgpool 2
        ZERO(amA);
        TRY(conv(e, r.stem[0], img, nullptr, n, R, R, 2, true, nullptr, RLCF_EPI_RELU, A, amA, st));
        ZERO(amB);
        TRY(conv(e, r.stem[1], A, amA, n, H, H, 1, false, nullptr, RLCF_EPI_RELU, B, amB, st));
        ZERO(amA);
        TRY(conv(e, r.stem[2], B, amB, n, H, H, 1, false, nullptr, RLCF_EPI_RELU, A, amA, st));
        H /= 2;
        TRY(avgpool2(A, X, n, H, H, w, st));
        RLCF_HIP_CHECK(hipMemcpyAsync(amX, amA, sizeof(float), hipMemcpyDeviceToDevice, st));
        for (const BottleW& b : r.blocks) {                                    // Bottleneck.forward, model.py:42-55
            const int planes = b.c1.cout, Ho = H / b.stride;
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
            // in place when idn == X (elementwise); amX is re-zeroed first: the GEMM reads X only as the residual operand
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
