// Attention backward for the image towers (no shared prefix, no causal mask) as TWO kernels, each with a single writer per output row:
//   attention_bwd_dq_kernel   one workgroup per (sequence, head, 64 queries)  walks the key chunks  -> dQ
//   attention_bwd_dkv_kernel  one workgroup per (sequence, head, 64 keys)     walks the query blocks -> dK, dV
// Same arithmetic as attention_bwd_x3.hip (split-f16 operands, three v_mfma_f32_32x32x16_f16 per product, P and dS rebuilt from the
// forward's log-sum-exp and output; that file's header has the derivation of the two operand orientations) — what changes is who
// accumulates.  There, a workgroup = one 32-query block: its role-B waves produce that block's CONTRIBUTION to every key's dK / dV,
// parked in memory ([sequence][query block][key][K | V], 1.2 MB per (sequence, head) at 257 tokens) and summed by a second launch
// (attention_bwd_park_reduce_kernel): 0.71 + part of 1.74 ms per image at BASELINE configs[2].  Here the dK / dV accumulators of a
// 64-key chunk stay in registers across ALL query blocks and leave once; nothing is parked, nothing is zero-filled, no atomics:
// bit-reproducible by construction.  K / V chunks are converted 5 x 5 instead of 9 x 5 times per (sequence, head) at 257 tokens.
#include "kernels.h"

#define BY_KLD 72      // halves per row of the [row][d] tiles (64 + 8 pad = 144 B: conflict-free ds_read_b128 fragments)
#define BY_TLD 40      // halves per row of the transposed [d][slot] tiles (32 + 8 pad = 80 B)

__device__ __forceinline__ void by_split8(const float* v, h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}
// slot of row q (0..31) of a 32x32 accumulator in its register order (attention_bwd_x3.hip: bx_slot)
__device__ __forceinline__ int by_slot(int q) { return ((q >> 4) << 4) | (((q >> 2) & 1) << 3) | (((q >> 3) & 1) << 2) | (q & 3); }

#define BY_MMA3(acc, ah, al, bh, bl)                                          \
    do {                                                                      \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);   \
    } while (0)

__device__ __forceinline__ void by_gscale(const unsigned int* amax_dout, float& gscale, float& inv_gscale) {
    // power of two lifting max|dO| into [2^3, 2^4): dP = dO V^T and dS = P o (dP - D) pass through f16 pairs too and need the headroom
    const float am = __uint_as_float(*amax_dout);
    int sh = 0;
    if (am > 0.f && am < INFINITY) sh = max(-40, min(60, 3 - (int)floorf(log2f(am))));
    gscale = ldexpf(1.0f, sh); inv_gscale = ldexpf(1.0f, -sh);
}

// ---- dQ: waves (kt, qb): query block qb of the workgroup's 64 queries against key tile kt of every 64-key chunk ------------------------
__global__ __launch_bounds__(256, 2) void attention_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                  const float* __restrict__ lse, const float* __restrict__ dout,
                                                                  const unsigned int* __restrict__ amax_dout, const rlcf_seq* __restrict__ seqs,
                                                                  int width, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char by_smem[];
    const rlcf_seq sq = seqs[blockIdx.y];
    const int head = blockIdx.z, t = threadIdx.x, lane = t & 63, wave = t >> 6, l32 = lane & 31, h = lane >> 5;
    const int kt = wave & 1, qb = wave >> 1, q0 = blockIdx.x * 64 + qb * 32;
    if ((int)blockIdx.x * 64 >= sq.q_len) return;
    const int nk = sq.q_len, ld = 3 * width, H = width / HEAD_DIM;
    _Float16* K_h = (_Float16*)by_smem;
    _Float16* K_l = K_h + 64 * BY_KLD;
    _Float16* V_h = K_l + 64 * BY_KLD;
    _Float16* V_l = V_h + 64 * BY_KLD;
    _Float16* KT_h = V_l + 64 * BY_KLD;                     // [2 tiles][64 d][40]
    _Float16* KT_l = KT_h + 2 * 64 * BY_TLD;
    float gscale, inv_gscale;
    by_gscale(amax_dout, gscale, inv_gscale);
    // operand registers of this wave's queries: lane (query l32, half h) holds d = ks*16 + h*8 + [0,8) of Q and of dO * gscale
    const int qi = min(q0 + l32, sq.q_len - 1);
    const bool q_ok = q0 + l32 < sq.q_len;
    h16x8 qh[4], ql[4], gh[4], gl[4];
    float lse_q = 0.f, D_q = 0.f;
    {
        const float* qp = qkv + (size_t)(sq.q_start + qi) * ld + head * HEAD_DIM + h * 8;
        const float* gp = dout + (size_t)(sq.q_start + qi) * width + head * HEAD_DIM + h * 8;
        const float* op = out + (size_t)(sq.q_start + qi) * width + head * HEAD_DIM + h * 8;
        float Dp = 0.f;
        const float z = q_ok ? 1.f : 0.f, gz = gscale * z;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 a = *(const float4*)(qp + ks * 16), b = *(const float4*)(qp + ks * 16 + 4);
            const float4 c = *(const float4*)(gp + ks * 16), d = *(const float4*)(gp + ks * 16 + 4);
            const float4 o0 = *(const float4*)(op + ks * 16), o1 = *(const float4*)(op + ks * 16 + 4);
            const float qv[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z};
            const float gv[8] = {c.x * gz, c.y * gz, c.z * gz, c.w * gz, d.x * gz, d.y * gz, d.z * gz, d.w * gz};
            by_split8(qv, qh[ks], ql[ks]);
            by_split8(gv, gh[ks], gl[ks]);
            Dp += c.x * o0.x + c.y * o0.y + c.z * o0.z + c.w * o0.w + d.x * o1.x + d.y * o1.y + d.z * o1.z + d.w * o1.w;
        }
        Dp += __shfl_xor(Dp, 32);
        D_q = q_ok ? Dp * gscale : 0.f;
        lse_q = q_ok ? lse[(size_t)(sq.q_start + qi) * H + head] : 0.f;
    }
    f32x16 dq0, dq1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
    constexpr float LOG2E = 1.44269504088896341f, SC = 0.125f * LOG2E;
    const bool wave_on = q0 < sq.q_len;                     // (the second query block of the last group may be empty)
    for (int kc = 0; kc < nk; kc += 64) {
        __syncthreads();
        {   // stage the chunk: thread -> (key t / 4, d part (t % 4) * 16): K, V as [key][d] pairs and K^T as two [d][slot] tiles
            const int j = t >> 2, d0 = (t & 3) * 16, kap = kc + j;
            const bool ok = kap < nk;
            const float* p = qkv + (size_t)(sq.q_start + min(kap, nk - 1)) * ld + head * HEAD_DIM + d0 + width;
            const float z = ok ? 1.f : 0.f;
            const int tile = j >> 5, sl = by_slot(j & 31);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const float4 a = *(const float4*)(p + part * 8), b = *(const float4*)(p + part * 8 + 4);
                const float4 c = *(const float4*)(p + width + part * 8), d = *(const float4*)(p + width + part * 8 + 4);
                const float kv[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z};
                const float vv[8] = {c.x * z, c.y * z, c.z * z, c.w * z, d.x * z, d.y * z, d.z * z, d.w * z};
                h16x8 kh8, kl8, vh8, vl8;
                by_split8(kv, kh8, kl8);
                by_split8(vv, vh8, vl8);
                *(h16x8*)(K_h + j * BY_KLD + d0 + part * 8) = kh8; *(h16x8*)(K_l + j * BY_KLD + d0 + part * 8) = kl8;
                *(h16x8*)(V_h + j * BY_KLD + d0 + part * 8) = vh8; *(h16x8*)(V_l + j * BY_KLD + d0 + part * 8) = vl8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    KT_h[(tile * 64 + d0 + part * 8 + e) * BY_TLD + sl] = kh8[e];
                    KT_l[(tile * 64 + d0 + part * 8 + e) * BY_TLD + sl] = kl8[e];
                }
            }
        }
        __syncthreads();
        const int k0 = kc + 32 * kt;
        if (!wave_on || k0 >= nk) continue;                 // (wave-uniform; the barriers are at the loop head)
        const _Float16* kh_ = K_h + (32 * kt + l32) * BY_KLD + h * 8;
        const _Float16* kl_ = K_l + (32 * kt + l32) * BY_KLD + h * 8;
        const _Float16* vh_ = V_h + (32 * kt + l32) * BY_KLD + h * 8;
        const _Float16* vl_ = V_l + (32 * kt + l32) * BY_KLD + h * 8;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                    // rows = keys, lanes = queries
            const h16x8 kh8 = *(const h16x8*)(kh_ + ks * 16), kl8 = *(const h16x8*)(kl_ + ks * 16);
            const h16x8 vh8 = *(const h16x8*)(vh_ + ks * 16), vl8 = *(const h16x8*)(vl_ + ks * 16);
            BY_MMA3(s, kh8, kl8, qh[ks], ql[ks]);
            BY_MMA3(dp, vh8, vl8, gh[ks], gl[ks]);
        }
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + mfma32_row(r, h);
            const bool ok = key < nk && q_ok;
            const float p = ok ? __builtin_amdgcn_exp2f(s[r] * SC - lse_q * LOG2E) : 0.f;
            sv[r] = p * (dp[r] - D_q);
        }
        h16x8 sh[2], sl_[2];
        by_split8(sv, sh[0], sl_[0]); by_split8(sv + 8, sh[1], sl_[1]);
        const _Float16* kth = KT_h + kt * 64 * BY_TLD;
        const _Float16* ktl = KT_l + kt * 64 * BY_TLD;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int o0 = l32 * BY_TLD + tt * 16 + h * 8, o1 = (32 + l32) * BY_TLD + tt * 16 + h * 8;
            const h16x8 k0h = *(const h16x8*)(kth + o0), k0l = *(const h16x8*)(ktl + o0);
            const h16x8 k1h = *(const h16x8*)(kth + o1), k1l = *(const h16x8*)(ktl + o1);
            BY_MMA3(dq0, sh[tt], sl_[tt], k0h, k0l);
            BY_MMA3(dq1, sh[tt], sl_[tt], k1h, k1l);
        }
    }
    // dQ of a query block: the two key-tile waves' partials summed through LDS (the chunk area is free), times 1/8 / gscale
    __syncthreads();
    float* red = (float*)by_smem;                           // [qb][kt][32][64] floats = 32 KB
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[((qb * 2 + kt) * 32 + mfma32_row(r, h)) * 64 + l32] = dq0[r];
        red[((qb * 2 + kt) * 32 + mfma32_row(r, h)) * 64 + 32 + l32] = dq1[r];
    }
    __syncthreads();
    const float fq = 0.125f * inv_gscale;
    for (int idx = t; idx < 64 * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63, b = i >> 5, il = i & 31, qrow = blockIdx.x * 64 + i;
        if (qrow < sq.q_len)
            dqkv[(size_t)(sq.q_start + qrow) * ld + head * HEAD_DIM + d] = (red[((b * 2) * 32 + il) * 64 + d] + red[((b * 2 + 1) * 32 + il) * 64 + d]) * fq;
    }
}

// ---- dK, dV: waves (kt, qpar): key tile kt of the workgroup's 64 keys against the query blocks 2 i + qpar, i = 0, 1, ... ------------------
__global__ __launch_bounds__(256, 2) void attention_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                   const float* __restrict__ lse, const float* __restrict__ dout,
                                                                   const unsigned int* __restrict__ amax_dout, const rlcf_seq* __restrict__ seqs,
                                                                   int width, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char by_smem[];
    const rlcf_seq sq = seqs[blockIdx.y];
    const int head = blockIdx.z, t = threadIdx.x, lane = t & 63, wave = t >> 6, l32 = lane & 31, h = lane >> 5;
    const int kt = wave & 1, qpar = wave >> 1, kbase = blockIdx.x * 64, k0 = kbase + 32 * kt;
    if (kbase >= sq.q_len) return;
    const int nk = sq.q_len, ld = 3 * width, H = width / HEAD_DIM;
    // LDS: Q, dO of 64 queries as [query][d] pairs (A operands of S / dP), their transposes per query block [d][slot] (B operands of
    // dK / dV), lse and D of the 64 queries
    _Float16* Q_h = (_Float16*)by_smem;
    _Float16* Q_l = Q_h + 64 * BY_KLD;
    _Float16* G_h = Q_l + 64 * BY_KLD;
    _Float16* G_l = G_h + 64 * BY_KLD;
    _Float16* QT_h = G_l + 64 * BY_KLD;                     // [2 query blocks][64 d][40]
    _Float16* QT_l = QT_h + 2 * 64 * BY_TLD;
    _Float16* GT_h = QT_l + 2 * 64 * BY_TLD;
    _Float16* GT_l = GT_h + 2 * 64 * BY_TLD;
    float* Ls = (float*)(GT_l + 2 * 64 * BY_TLD);           // [64]
    float* Ds = Ls + 64;                                    // [64] D * gscale
    float gscale, inv_gscale;
    by_gscale(amax_dout, gscale, inv_gscale);
    // operand registers of this wave's keys: lane (key l32, half h) holds d = ks*16 + h*8 + [0,8) of K and V (B operands: keys = lanes)
    const int ki = min(k0 + l32, nk - 1);
    const bool k_ok = k0 + l32 < nk;
    h16x8 kh[4], kl[4], vh[4], vl[4];
    {
        const float* kp = qkv + (size_t)(sq.q_start + ki) * ld + head * HEAD_DIM + h * 8 + width;
        const float z = k_ok ? 1.f : 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 a = *(const float4*)(kp + ks * 16), b = *(const float4*)(kp + ks * 16 + 4);
            const float4 c = *(const float4*)(kp + width + ks * 16), d = *(const float4*)(kp + width + ks * 16 + 4);
            const float kv[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z};
            const float vv[8] = {c.x * z, c.y * z, c.z * z, c.w * z, d.x * z, d.y * z, d.z * z, d.w * z};
            by_split8(kv, kh[ks], kl[ks]);
            by_split8(vv, vh[ks], vl[ks]);
        }
    }
    f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
    constexpr float LOG2E = 1.44269504088896341f, SC = 0.125f * LOG2E;
    const bool wave_on = k0 < nk;
    for (int qc = 0; qc < sq.q_len; qc += 64) {
        __syncthreads();
        {   // stage 64 queries: thread -> (query t / 4, d part (t % 4) * 16): Q, dO * gscale as [query][d] pairs + transposed tiles; D, lse
            const int j = t >> 2, d0 = (t & 3) * 16, qa = qc + j;
            const bool ok = qa < sq.q_len;
            const size_t row = (size_t)(sq.q_start + min(qa, sq.q_len - 1));
            const float* qp = qkv + row * ld + head * HEAD_DIM + d0;
            const float* gp = dout + row * width + head * HEAD_DIM + d0;
            const float* op = out + row * width + head * HEAD_DIM + d0;
            const float z = ok ? 1.f : 0.f, gz = gscale * z;
            const int blk = j >> 5, sl = by_slot(j & 31);
            float Dp = 0.f;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const float4 a = *(const float4*)(qp + part * 8), b = *(const float4*)(qp + part * 8 + 4);
                const float4 c = *(const float4*)(gp + part * 8), d = *(const float4*)(gp + part * 8 + 4);
                const float4 o0 = *(const float4*)(op + part * 8), o1 = *(const float4*)(op + part * 8 + 4);
                const float qv[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z};
                const float gv[8] = {c.x * gz, c.y * gz, c.z * gz, c.w * gz, d.x * gz, d.y * gz, d.z * gz, d.w * gz};
                Dp += c.x * o0.x + c.y * o0.y + c.z * o0.z + c.w * o0.w + d.x * o1.x + d.y * o1.y + d.z * o1.z + d.w * o1.w;
                h16x8 xh, xl, yh, yl;
                by_split8(qv, xh, xl);
                by_split8(gv, yh, yl);
                *(h16x8*)(Q_h + j * BY_KLD + d0 + part * 8) = xh; *(h16x8*)(Q_l + j * BY_KLD + d0 + part * 8) = xl;
                *(h16x8*)(G_h + j * BY_KLD + d0 + part * 8) = yh; *(h16x8*)(G_l + j * BY_KLD + d0 + part * 8) = yl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = (blk * 64 + d0 + part * 8 + e) * BY_TLD + sl;
                    QT_h[o] = xh[e]; QT_l[o] = xl[e]; GT_h[o] = yh[e]; GT_l[o] = yl[e];
                }
            }
            // D of the query = sum over its 64 d of dO o O: the four threads of a query (consecutive lanes) hold 16 d each
            Dp += __shfl_xor(Dp, 1);
            Dp += __shfl_xor(Dp, 2);
            if ((t & 3) == 0) { Ds[j] = ok ? Dp * gscale : 0.f; Ls[j] = ok ? lse[row * H + head] : 0.f; }
        }
        __syncthreads();
        const int q0 = qc + 32 * qpar;                      // this wave's query block of the 64
        if (!wave_on || q0 >= sq.q_len) continue;           // (wave-uniform; the barriers are at the loop head)
        const _Float16* qh_ = Q_h + (32 * qpar + l32) * BY_KLD + h * 8;
        const _Float16* ql_ = Q_l + (32 * qpar + l32) * BY_KLD + h * 8;
        const _Float16* gh_ = G_h + (32 * qpar + l32) * BY_KLD + h * 8;
        const _Float16* gl_ = G_l + (32 * qpar + l32) * BY_KLD + h * 8;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                    // rows = queries, lanes = keys
            const h16x8 qh8 = *(const h16x8*)(qh_ + ks * 16), ql8 = *(const h16x8*)(ql_ + ks * 16);
            const h16x8 gh8 = *(const h16x8*)(gh_ + ks * 16), gl8 = *(const h16x8*)(gl_ + ks * 16);
            BY_MMA3(s, qh8, ql8, kh[ks], kl[ks]);
            BY_MMA3(dp, gh8, gl8, vh[ks], vl[ks]);
        }
        float pv[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma32_row(r, h);
            const bool ok = k_ok && q0 + row < sq.q_len;
            const float p = ok ? __builtin_amdgcn_exp2f(s[r] * SC - Ls[32 * qpar + row] * LOG2E) : 0.f;
            pv[r] = p * 64.0f;
            sv[r] = p * (dp[r] - Ds[32 * qpar + row]);
        }
        h16x8 ph[2], pl[2], sh[2], sl_[2];
        by_split8(pv, ph[0], pl[0]); by_split8(pv + 8, ph[1], pl[1]);
        by_split8(sv, sh[0], sl_[0]); by_split8(sv + 8, sh[1], sl_[1]);
        const _Float16 *gth = GT_h + qpar * 64 * BY_TLD, *gtl = GT_l + qpar * 64 * BY_TLD;
        const _Float16 *qth = QT_h + qpar * 64 * BY_TLD, *qtl = QT_l + qpar * 64 * BY_TLD;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int o0 = l32 * BY_TLD + tt * 16 + h * 8, o1 = (32 + l32) * BY_TLD + tt * 16 + h * 8;
            const h16x8 g0h = *(const h16x8*)(gth + o0), g0l = *(const h16x8*)(gtl + o0);
            const h16x8 g1h = *(const h16x8*)(gth + o1), g1l = *(const h16x8*)(gtl + o1);
            const h16x8 q0h = *(const h16x8*)(qth + o0), q0l = *(const h16x8*)(qtl + o0);
            const h16x8 q1h = *(const h16x8*)(qth + o1), q1l = *(const h16x8*)(qtl + o1);
            BY_MMA3(dv0, ph[tt], pl[tt], g0h, g0l);
            BY_MMA3(dv1, ph[tt], pl[tt], g1h, g1l);
            BY_MMA3(dk0, sh[tt], sl_[tt], q0h, q0l);
            BY_MMA3(dk1, sh[tt], sl_[tt], q1h, q1l);
        }
    }
    // dK / dV of a key tile: the two query-parity waves' partials summed through LDS (fixed order: even blocks + odd blocks)
    __syncthreads();
    float* red = (float*)by_smem;                           // [qpar][kt][32 keys][K 64 | V 64] floats = 64 KB
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* b_ = red + (((qpar * 2 + kt) * 32 + mfma32_row(r, h)) * 128);
        b_[l32] = dk0[r]; b_[32 + l32] = dk1[r]; b_[64 + l32] = dv0[r]; b_[96 + l32] = dv1[r];
    }
    __syncthreads();
    const float fv = inv_gscale * 0.015625f, fk = inv_gscale * 0.125f;
    for (int idx = t; idx < 64 * 128; idx += 256) {
        const int i = idx >> 7, c = idx & 127, tile = i >> 5, il = i & 31, key = kbase + i;
        if (key < nk) {
            const float v = red[((0 * 2 + tile) * 32 + il) * 128 + c] + red[((1 * 2 + tile) * 32 + il) * 128 + c];
            float* dst = dqkv + (size_t)(sq.q_start + key) * ld + head * HEAD_DIM + (c < 64 ? width + c : 2 * width + (c - 64));
            *dst = v * (c < 64 ? fk : fv);
        }
    }
}

// sequences without a shared prefix, no causal mask (the image towers): dqkv is written completely, by single writers
int launch_attention_bwd_x3_split(const float* qkv, const float* out, const float* lse, const float* dout, const float* amax_dout, const rlcf_seq* seqs,
                                  int n_seq, int max_q_len, int width, float* dqkv, hipStream_t st) {
    RLCF_ARG_CHECK(n_seq > 0 && width % HEAD_DIM == 0 && max_q_len > 0 && qkv && out && lse && dout && dqkv && amax_dout && n_seq <= 65535);
    const size_t lds_dq = std::max((size_t)(4 * 64 * BY_KLD + 2 * 2 * 64 * BY_TLD) * sizeof(_Float16), (size_t)4 * 32 * 64 * sizeof(float));
    const size_t lds_dkv = std::max((size_t)(4 * 64 * BY_KLD + 4 * 2 * 64 * BY_TLD) * sizeof(_Float16) + 128 * sizeof(float), (size_t)4 * 32 * 128 * sizeof(float));
    { int rc_ = rlcf_func_lds((const void*)attention_bwd_dq_kernel, lds_dq); if (rc_ != RLCF_OK) return rc_; }
    { int rc_ = rlcf_func_lds((const void*)attention_bwd_dkv_kernel, lds_dkv); if (rc_ != RLCF_OK) return rc_; }
    const dim3 grid((max_q_len + 63) / 64, n_seq, width / HEAD_DIM);
    attention_bwd_dkv_kernel<<<grid, dim3(256), lds_dkv, st>>>(qkv, out, lse, dout, (const unsigned int*)amax_dout, seqs, width, dqkv);
    RLCF_LAUNCH_CHECK();
    attention_bwd_dq_kernel<<<grid, dim3(256), lds_dq, st>>>(qkv, out, lse, dout, (const unsigned int*)amax_dout, seqs, width, dqkv);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
