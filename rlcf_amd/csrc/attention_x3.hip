// Attention forward on the f16 matrix cores with f32-grade accuracy (split-f16, see gemm_f16x3.hip):
// every operand of both contractions is carried as hi + lo (two f16 numbers) and each product costs three
// v_mfma_f32_32x32x16_f16 into ONE f32 accumulator (hi*hi + hi*lo + lo*hi; P is carried times 2^6 so that its lo
// parts stay in f16's normal range).  Same dataflow as attention_f32.hip (one wave = 32 queries, both
// contractions transposed so a lane owns one query, online softmax lane-local), but
//   * NW waves (= NW query blocks of one (sequence, head)) share each converted K/V chunk in LDS;
//   * K is stored [key][d] (rows padded to 144 B), V is stored TRANSPOSED [d][key-slot] (rows padded
//     to 80 B) in the key order the S^T accumulator already has, so P feeds the second MFMA from
//     registers and every operand fetch is one conflict-free ds_read_b128.
// Replaces nn.MultiheadAttention's core (TPT/clip/model.py:175,185-187) in RLCF_PREC_F16X3 mode.
#include "kernels.h"

#define AX_KLD 72      // halves per K row  (64 + 8 pad  = 144 B)
#define AX_VLD 40      // halves per Vt row (32 + 8 pad  =  80 B)

__device__ __forceinline__ void split8(const float* v, h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}

// SINGLE: one MFMA per product on the hi parts only — plain f16 attention (RLCF_PREC_F16, the arithmetic of the reference's fp16
// autocast); the lo tiles are neither written nor read, the output is a plain f16 matrix (ol null).
template <int NW, bool SINGLE>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void attention_fwd_x3_kernel(const float* __restrict__ qkv, const rlcf_seq* __restrict__ seqs,
                                                                    int width, int causal, float* __restrict__ out,
                                                                    _Float16* __restrict__ oh, _Float16* __restrict__ ol, int il, int qb0,
                                                                    float* __restrict__ lse, const int32_t* __restrict__ rss) {
    // rss (one-wave blocks only): PACKED short sequences — the descriptor names a run of up to 32 - pre_len consecutive rows that holds
    // several whole sequences sharing one prefix; rss[row] = first row of the sequence `row` belongs to, and a query sees the prefix
    // keys plus the keys of its own sequence up to itself.  One MFMA tile then serves ~7 class prompts instead of one.
    // qb0: first 32-query block this launch covers (a 257-token ViT-L/14 sequence = one 8-wave block for queries 0..255 plus a
    // one-wave launch for the last query, instead of a second 8-wave block that would re-stage every K/V chunk for one row)
    const rlcf_seq sq = seqs[blockIdx.y];
    const int head = blockIdx.z;
    if ((qb0 + blockIdx.x * NW) * 32 >= sq.q_len) return;
    // two images of the converted K / V^T chunk: chunk c+1 is fetched (registers) and written while chunk c is consumed
    constexpr int NB = NW > 1 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) _Float16 Kh[NB][32 * AX_KLD], Kl[NB][32 * AX_KLD], Vh[NB][64 * AX_VLD], Vl[NB][64 * AX_VLD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l32 = lane & 31, h = lane >> 5;
    const int ld = 3 * width;
    const int qb = qb0 + blockIdx.x * NW + wave;
    const bool active = qb * 32 < sq.q_len;                      // waves past the end only help loading
    const int qi = min(qb * 32 + l32, sq.q_len - 1);
    const int nkeys = sq.pre_len + sq.q_len;
    const int qpos = sq.pre_len + qi;
    const int last_q = min((qb0 + blockIdx.x * NW) * 32 + NW * 32, sq.q_len);      // one past the last query of this workgroup
    const int kend = causal ? min(nkeys, sq.pre_len + last_q) : nkeys;
    const int my_kend = causal ? min(nkeys, sq.pre_len + qb * 32 + 32) : nkeys;

    // Q fragments: lane (q, h) owns d = ks*16 + h*8 + [0,8) for ks = 0..3; the 1/8 scale is applied to the scores (exact)
    h16x8 qh[4], ql[4];
    {
        const float* qp = qkv + (size_t)(sq.q_start + qi) * ld + head * HEAD_DIM + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 a = *(const float4*)(qp + ks * 16), b = *(const float4*)(qp + ks * 16 + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            split8(v, qh[ks], ql[ks]);
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -INFINITY, lsum = 0.f;
    constexpr float SC = 0.125f * 1.44269504088896341f;           // softmax(s/8) through v_exp_f32 (2^x)

    // chunk staging: thread -> (key, PER consecutive d); the key's V slot in the transposed tile is the position its
    // score occupies in the S^T accumulator: key = (e&3) + 8*(2t + (e>>2)) + 4h  <->  slot = 16t + 8h + e
    constexpr int PER = 32 / NW, PARTS = 64 / PER, NV = PER / 4;
    const int lkey = t / PARTS, d0 = (t % PARTS) * PER;
    const int slot = ((lkey >> 4) << 4) | (((lkey >> 2) & 1) << 3) | (((lkey >> 3) & 1) << 2) | (lkey & 3);
    float4 kq[NV], vq[NV];
#define AX_FETCH(kc_)                                                                                                    \
    {                                                                                                                    \
        const int kap = (kc_) + lkey;                                                                                    \
        if (kap < nkeys) {                                                                                               \
            const int row = kap < sq.pre_len ? sq.pre_start + kap : sq.q_start + kap - sq.pre_len;                       \
            const float* p = qkv + (size_t)row * ld + head * HEAD_DIM + d0;                                              \
            _Pragma("unroll") for (int j = 0; j < NV; ++j) {                                                             \
                kq[j] = *(const float4*)(p + width + 4 * j);                                                             \
                vq[j] = *(const float4*)(p + 2 * width + 4 * j);                                                         \
            }                                                                                                            \
        } else {                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < NV; ++j) { kq[j] = make_float4(0.f, 0.f, 0.f, 0.f); vq[j] = kq[j]; }   \
        }                                                                                                                \
    }
#define AX_STORE(buf_)                                                                                                   \
    {                                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < NV; ++j) {                                                                 \
            const float kk[4] = {kq[j].x, kq[j].y, kq[j].z, kq[j].w}, vv[4] = {vq[j].x, vq[j].y, vq[j].z, vq[j].w};      \
            h16x4 a, b;                                                                                                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
                a[e] = (_Float16)kk[e]; b[e] = (_Float16)(kk[e] - (float)a[e]);                                          \
                const _Float16 hh = (_Float16)vv[e];                                                                     \
                Vh[buf_][(d0 + 4 * j + e) * AX_VLD + slot] = hh;                                                         \
                if constexpr (!SINGLE) Vl[buf_][(d0 + 4 * j + e) * AX_VLD + slot] = (_Float16)(vv[e] - (float)hh);       \
            }                                                                                                            \
            *(h16x4*)(Kh[buf_] + lkey * AX_KLD + d0 + 4 * j) = a;                                                        \
            if constexpr (!SINGLE) *(h16x4*)(Kl[buf_] + lkey * AX_KLD + d0 + 4 * j) = b;                                 \
        }                                                                                                                \
    }
    // one-wave blocks (text sequences: a single chunk of <= 32 keys) gain nothing from the prefetch and pay for its registers
    constexpr bool PIPE = NW > 1;
    // staging pipeline: chunk c is consumed from one LDS image while chunk c+1 (fetched during the PREVIOUS iteration) is converted and
    // written to the other image right after the barrier, and the fetch of chunk c+2 is issued at once into the same registers — the
    // loads have a whole iteration (compute + barrier skew) to land instead of one compute phase (measured -6...-8 % on the kernel)
    if (PIPE) {
        AX_FETCH(0)
        AX_STORE(0)
        if (32 < kend) AX_FETCH(32)
        __syncthreads();
    }
    int buf = 0;
    for (int kc = 0; kc < kend; kc += 32, buf ^= PIPE ? 1 : 0) {
        const bool has_next = PIPE && kc + 32 < kend;
        if (!PIPE) {
            AX_FETCH(kc)
            AX_STORE(0)
            __syncthreads();
        }
        if (has_next) {
            AX_STORE((buf ^ 1) & (NB - 1))
            if (kc + 64 < kend) AX_FETCH(kc + 64)
        }
        if (active && kc < my_kend) {
            const _Float16 *kh_ = Kh[buf], *kl_ = Kl[buf], *vh_ = Vh[buf], *vl_ = Vl[buf];
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kh = *(const h16x8*)(kh_ + l32 * AX_KLD + ks * 16 + h * 8);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
                if constexpr (!SINGLE) {
                    const h16x8 kl = *(const h16x8*)(kl_ + l32 * AX_KLD + ks * 16 + h * 8);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
                }
            }
            float cm = -INFINITY;
            if (rss) {
                const int first = rss[sq.q_start + qi] - sq.q_start + sq.pre_len;       // key index of the query's own sequence start
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kc + mfma32_row(r, h);
                    if (key >= nkeys || key > qpos || (key >= sq.pre_len && key < first)) s[r] = -INFINITY;
                }
            } else if (kc + 32 > nkeys || (causal && kc + 31 > sq.pre_len + qb * 32)) {       // chunk holds masked keys
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kc + mfma32_row(r, h);
                    if (key >= nkeys || (causal && key > qpos)) s[r] = -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) cm = fmaxf(cm, s[r]);
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            const float mn = fmaxf(m, cm);
            const float alpha = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m - mn) * SC);
            const float mb = mn * SC - 6.0f;                     // P is carried times 2^6: folded into the exponent (lsum carries it too)
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] * SC - mb); ps += s[r]; }
            lsum = lsum * alpha + ps;
            if (alpha != 1.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float pv[8];
#pragma unroll
                for (int e2 = 0; e2 < 8; ++e2) pv[e2] = s[8 * tt + e2];
                h16x8 ph, pl;
                split8(pv, ph, pl);
                const h16x8 v0h = *(const h16x8*)(vh_ + l32 * AX_VLD + tt * 16 + h * 8);
                const h16x8 v1h = *(const h16x8*)(vh_ + (32 + l32) * AX_VLD + tt * 16 + h * 8);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, ph, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, ph, o1, 0, 0, 0);
                if constexpr (!SINGLE) {
                    const h16x8 v0l = *(const h16x8*)(vl_ + l32 * AX_VLD + tt * 16 + h * 8);
                    const h16x8 v1l = *(const h16x8*)(vl_ + (32 + l32) * AX_VLD + tt * 16 + h * 8);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, pl, o0, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0l, ph, o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, pl, o1, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1l, ph, o1, 0, 0, 0);
                }
            }
            m = mn;
        }
        __syncthreads();
    }
    if (!active) return;
    const float ltot = lsum + __shfl_xor(lsum, 32);
    if (qb * 32 + l32 < sq.q_len) {
        const float inv = 1.0f / ltot;                            // numerator and denominator both carry the 2^6 of P
        // log-sum-exp of the row's scores s/8 (saved for the flash-style backward): m is the running max of the raw scores
        if (lse && h == 0) lse[(size_t)(sq.q_start + qi) * (width / HEAD_DIM) + head] = m * 0.125f + logf(ltot * 0.015625f);
        const size_t obase = (size_t)(sq.q_start + qi) * width + head * HEAD_DIM;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * h;
            float v0[4], v1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v0[q] = o0[4 * g + q] * inv;
                v1[q] = o1[4 * g + q] * inv;
            }
            if (out) {
                *(float4*)(out + obase + d) = make_float4(v0[0], v0[1], v0[2], v0[3]);
                *(float4*)(out + obase + 32 + d) = make_float4(v1[0], v1[1], v1[2], v1[3]);
            }
            if (oh) {
                h16x4 h0, l0, h1, l1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h0[q] = (_Float16)v0[q]; l0[q] = (_Float16)(v0[q] - (float)h0[q]);
                    h1[q] = (_Float16)v1[q]; l1[q] = (_Float16)(v1[q] - (float)h1[q]);
                }
                // interleaved pair layout (il): the head's two 32-column blocks sit at row*2W + head*128 (+64), lo 32 halves after hi
                const size_t p0 = il ? (size_t)(sq.q_start + qi) * 2 * width + head * 128 + d : obase + d;
                const size_t p1 = il ? p0 + 64 : p0 + 32;
                *(h16x4*)(oh + p0) = h0; *(h16x4*)(oh + p1) = h1;
                if (ol) { *(h16x4*)(ol + p0) = l0; *(h16x4*)(ol + p1) = l1; }
            }
        }
    }
}

template <int NW>
static void attn_launch(bool single, dim3 grid, hipStream_t st, const float* qkv, const rlcf_seq* seqs, int width, int causal, float* out,
                        _Float16* oh, _Float16* ol, int il, int qb0, float* lse, const int32_t* rss = nullptr) {
    if (single) attention_fwd_x3_kernel<NW, true><<<grid, dim3(64 * NW), 0, st>>>(qkv, seqs, width, causal, out, oh, ol, il, qb0, lse, rss);
    else attention_fwd_x3_kernel<NW, false><<<grid, dim3(64 * NW), 0, st>>>(qkv, seqs, width, causal, out, oh, ol, il, qb0, lse, rss);
}
int launch_attention_fwd_x3(const float* qkv, const rlcf_seq* seqs, int n_seq, int max_q_len, int width, int causal, float* out,
                            void* out_hi, void* out_lo, hipStream_t st, int il, float* lse, int single, const int32_t* row_seq_start) {
    RLCF_ARG_CHECK(!row_seq_start || (max_q_len <= 32 && causal));
    RLCF_ARG_CHECK(n_seq > 0 && max_q_len > 0 && width % HEAD_DIM == 0 && (out || (out_hi && (out_lo || single))));
    RLCF_ARG_CHECK(n_seq <= 65535 * 16);
    if (max_q_len > 128) {         // ViT sequences (197 / 257 tokens): 8 query blocks share every converted K/V chunk
        const int full = max_q_len / 256, tail = max_q_len - full * 256;
        const bool split_tail = full >= 1 && tail > 0 && tail <= 32;        // 257 tokens: the odd query goes to a one-wave launch
        dim3 grid(split_tail ? full : (max_q_len + 255) / 256, n_seq, width / HEAD_DIM);
        RLCF_ARG_CHECK(grid.y <= 65535);
        attn_launch<8>(single, grid, st, qkv, seqs, width, causal, out, (_Float16*)out_hi, (_Float16*)out_lo, il, 0, lse);
        if (split_tail) {
            RLCF_LAUNCH_CHECK();
            attn_launch<1>(single, dim3(1, n_seq, width / HEAD_DIM), st, qkv, seqs, width, causal, out, (_Float16*)out_hi, (_Float16*)out_lo, il,
                           full * 8, lse);
        }
    } else if (max_q_len > 32) {
        dim3 grid((max_q_len + 127) / 128, n_seq, width / HEAD_DIM);
        RLCF_ARG_CHECK(grid.y <= 65535);
        attn_launch<4>(single, grid, st, qkv, seqs, width, causal, out, (_Float16*)out_hi, (_Float16*)out_lo, il, 0, lse);
    } else {
        dim3 grid(1, n_seq, width / HEAD_DIM);
        RLCF_ARG_CHECK(grid.y <= 65535);
        attn_launch<1>(single, grid, st, qkv, seqs, width, causal, out, (_Float16*)out_hi, (_Float16*)out_lo, il, 0, lse, row_seq_start);
    }
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
