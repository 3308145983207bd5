// Multi-head attention core (nn.MultiheadAttention as used by TPT/clip/model.py:175,185-187),
// parity mode: f32 storage, f32-input MFMA, online softmax in f32 with accurate expf.
//
// Forward: one wave = one 32-query block of one (sequence, head); keys streamed in chunks of 32
// through LDS.  Both contractions are computed TRANSPOSED so that every lane owns ONE query:
//     S^T[key][q] = sum_d K[key][d] Q[q][d]        (A = K from LDS, B = Q held in 32 VGPRs)
//     O^T[d][q]  += sum_key V[key][d] P[q][key]    (A = V from LDS, B = P = the S^T accumulator)
// In the 32x32 MFMA C layout a lane holds column (lane&31) = its query and 16 rows (keys / d),
// so row max / row sum / rescale are lane-local plus one exchange between the two half-waves,
// and P feeds the second MFMA straight from the accumulator registers (no LDS round trip).
#include "kernels.h"

__global__ __launch_bounds__(64) void attention_fwd_f32_kernel(const float* __restrict__ qkv, const rlcf_seq* __restrict__ seqs,
                                                               int width, int causal, float* __restrict__ out,
                                                               float* __restrict__ lse, _Float16* __restrict__ oh,
                                                               _Float16* __restrict__ ol) {
    const rlcf_seq sq = seqs[blockIdx.y];
    const int qb = blockIdx.x, head = blockIdx.z;
    if (qb * 32 >= sq.q_len) return;
    __shared__ float Ks[32][65];
    __shared__ float Vs[32][64];
    const int lane = threadIdx.x, l32 = lane & 31, h = lane >> 5;
    const int ld = 3 * width, H = width / HEAD_DIM;
    const int qi = min(qb * 32 + l32, sq.q_len - 1);            // clamp: duplicate last query, never stored
    const int nkeys = sq.pre_len + sq.q_len;
    const int qpos = sq.pre_len + qi;                           // last key this query may see when causal
    const int kend = causal ? min(nkeys, sq.pre_len + qb * 32 + 32) : nkeys;

    float q[32];
    {
        const float* qp = qkv + (size_t)(sq.q_start + qi) * ld + head * HEAD_DIM + h * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 t = *(const float4*)(qp + j * 4);
            q[4 * j] = t.x * 0.125f; q[4 * j + 1] = t.y * 0.125f; q[4 * j + 2] = t.z * 0.125f; q[4 * j + 3] = t.w * 0.125f;
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -INFINITY, lsum = 0.f;

    for (int kc = 0; kc < kend; kc += 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + i * 64, key = idx >> 4, c4 = (idx & 15) * 4;
            const int kap = kc + key;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (kap < nkeys) {
                const int row = kap < sq.pre_len ? sq.pre_start + kap : sq.q_start + kap - sq.pre_len;
                const float* p = qkv + (size_t)row * ld + head * HEAD_DIM + c4;
                kv = *(const float4*)(p + width);
                vv = *(const float4*)(p + 2 * width);
            }
            Ks[key][c4] = kv.x; Ks[key][c4 + 1] = kv.y; Ks[key][c4 + 2] = kv.z; Ks[key][c4 + 3] = kv.w;
            *(float4*)&Vs[key][c4] = vv;
        }
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 32; ++st) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l32][h * 32 + st], q[st], s, 0, 0, 0);
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kc + mfma32_row(r, h);
            if (key >= nkeys || (causal && key > qpos)) s[r] = -INFINITY;
            cm = fmaxf(cm, s[r]);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);
        const float alpha = (m == -INFINITY) ? 0.f : expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - mn); ps += s[r]; }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int kk = mfma32_row(st, h);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kk][l32], s[st], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kk][32 + l32], s[st], o1, 0, 0, 0);
        }
        m = mn;
        __syncthreads();
    }
    const float ltot = lsum + __shfl_xor(lsum, 32);
    if (qb * 32 + l32 < sq.q_len) {
        const float inv = 1.0f / ltot;
        const size_t obase = (size_t)(sq.q_start + qi) * width + head * HEAD_DIM;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * h;
            const float v0[4] = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
            const float v1[4] = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
            if (out) {
                *(float4*)(out + obase + d) = make_float4(v0[0], v0[1], v0[2], v0[3]);
                *(float4*)(out + obase + 32 + d) = make_float4(v1[0], v1[1], v1[2], v1[3]);
            }
            if (oh) {                                   // split-f16 pair for the out-proj GEMM
                h16x4 h0, l0, h1, l1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h0[q] = (_Float16)v0[q]; l0[q] = (_Float16)(v0[q] - (float)h0[q]);
                    h1[q] = (_Float16)v1[q]; l1[q] = (_Float16)(v1[q] - (float)h1[q]);
                }
                *(h16x4*)(oh + obase + d) = h0; *(h16x4*)(ol + obase + d) = l0;
                *(h16x4*)(oh + obase + 32 + d) = h1; *(h16x4*)(ol + obase + 32 + d) = l1;
            }
        }
        if (lse && h == 0) lse[(size_t)(sq.q_start + qi) * H + head] = m + logf(ltot);
    }
}

int launch_attention_fwd_f32(const float* qkv, const rlcf_seq* seqs, int n_seq, int max_q_len, int width, int causal,
                             float* out, float* lse, hipStream_t st, void* out_hi, void* out_lo) {
    RLCF_ARG_CHECK(n_seq > 0 && max_q_len > 0 && width % HEAD_DIM == 0);
    dim3 grid((max_q_len + 31) / 32, n_seq, width / HEAD_DIM);
    RLCF_ARG_CHECK(grid.y <= 65535 && grid.z <= 65535);
    attention_fwd_f32_kernel<<<grid, dim3(64), 0, st>>>(qkv, seqs, width, causal, out, lse, (_Float16*)out_hi, (_Float16*)out_lo);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ------------------------------------------------------------------------------------------
// Backward (dX only) for short sequences (text tower: <= 96 keys): one 256-thread workgroup per
// (sequence, head) with Q,K,V,dO and the probability matrix resident in LDS.  The default
// RLCF configuration back-propagates only n_sel*K (= 18) class prompts,
// so this kernel is latency- not throughput-critical and stays on the f32 VALU.
//   P = softmax(QK^T/8 + mask); dV = P^T dO; dP = dO V^T; dS = P*(dP - rowsum(P*dP));
//   dQ = dS K / 8; dK = dS^T Q / 8.   dK/dV use atomicAdd: prefix keys are shared by sequences.
#define ABWD_MAXK 96
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                            const rlcf_seq* __restrict__ seqs, int width, int causal,
                                                            float* __restrict__ dqkv, float* __restrict__ pre_ws, int max_pre) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const rlcf_seq sq = seqs[blockIdx.x];
    const int head = blockIdx.y;
    const int nq = sq.q_len, nk = sq.pre_len + sq.q_len;
    if (nq <= 0) return;
    const int ld = 3 * width, t = threadIdx.x;
    const int LDQ = 65, LDP = nk + 1;
    float* Qs = sm;                       // [nq][65]  (scaled by 1/8)
    float* Ks = Qs + nq * LDQ;            // [nk][65]
    float* Vs = Ks + nk * LDQ;            // [nk][65]
    float* Gs = Vs + nk * LDQ;            // [nq][65]  dO
    float* Ps = Gs + nq * LDQ;            // [nq][nk+1]
    for (int idx = t; idx < nq * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63;
        Qs[i * LDQ + d] = qkv[(size_t)(sq.q_start + i) * ld + head * HEAD_DIM + d] * 0.125f;
        Gs[i * LDQ + d] = dout[(size_t)(sq.q_start + i) * width + head * HEAD_DIM + d];
    }
    for (int idx = t; idx < nk * 64; idx += 256) {
        const int j = idx >> 6, d = idx & 63;
        const int row = j < sq.pre_len ? sq.pre_start + j : sq.q_start + j - sq.pre_len;
        Ks[j * LDQ + d] = qkv[(size_t)row * ld + width + head * HEAD_DIM + d];
        Vs[j * LDQ + d] = qkv[(size_t)row * ld + 2 * width + head * HEAD_DIM + d];
    }
    __syncthreads();
    for (int idx = t; idx < nq * nk; idx += 256) {
        const int i = idx / nk, j = idx % nk;
        float s = -INFINITY;
        if (!causal || j <= sq.pre_len + i) {
            s = 0.f;
#pragma unroll 16
            for (int d = 0; d < 64; ++d) s += Qs[i * LDQ + d] * Ks[j * LDQ + d];
        }
        Ps[i * LDP + j] = s;
    }
    __syncthreads();
    for (int i = t; i < nq; i += 256) {
        float mx = -INFINITY;
        for (int j = 0; j < nk; ++j) mx = fmaxf(mx, Ps[i * LDP + j]);
        float sum = 0.f;
        for (int j = 0; j < nk; ++j) { float e = expf(Ps[i * LDP + j] - mx); Ps[i * LDP + j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < nk; ++j) Ps[i * LDP + j] *= inv;
    }
    __syncthreads();
    // dV[j][d] = sum_i P[i][j] dO[i][d]
    for (int idx = t; idx < nk * 64; idx += 256) {
        const int j = idx >> 6, d = idx & 63;
        float s = 0.f;
        for (int i = 0; i < nq; ++i) s += Ps[i * LDP + j] * Gs[i * LDQ + d];
        // own rows: this workgroup is their only writer.  Prefix rows are shared by sequences: with a workspace every sequence parks
        // its contribution in its own slot and attention_bwd_prefix_reduce_kernel adds them in sequence order (deterministic)
        if (j < sq.pre_len && pre_ws) pre_ws[((size_t)blockIdx.x * max_pre + j) * 2 * width + width + head * HEAD_DIM + d] = s;
        else {
            const int row = j < sq.pre_len ? sq.pre_start + j : sq.q_start + j - sq.pre_len;
            atomicAdd(dqkv + (size_t)row * ld + 2 * width + head * HEAD_DIM + d, s);
        }
    }
    __syncthreads();
    // dS = P * (dP - D), D_i = sum_j P_ij dP_ij, dP_ij = dO_i . V_j
    for (int i = t; i < nq; i += 256) {
        float D = 0.f;
        for (int j = 0; j < nk; ++j) {
            float dp = 0.f;
            const float p = Ps[i * LDP + j];
            if (p != 0.f) {
#pragma unroll 16
                for (int d = 0; d < 64; ++d) dp += Gs[i * LDQ + d] * Vs[j * LDQ + d];
            }
            D += p * dp;
        }
        // stash D in the padding column
        Ps[i * LDP + nk] = D;
    }
    __syncthreads();
    for (int idx = t; idx < nq * nk; idx += 256) {
        const int i = idx / nk, j = idx % nk;
        const float p = Ps[i * LDP + j];
        float dp = 0.f;
        if (p != 0.f) {
#pragma unroll 16
            for (int d = 0; d < 64; ++d) dp += Gs[i * LDQ + d] * Vs[j * LDQ + d];
        }
        Ps[i * LDP + j] = p * (dp - Ps[i * LDP + nk]);
    }
    __syncthreads();
    // dQ[i][d] = sum_j dS[i][j] K[j][d] / 8 ;  dK[j][d] = sum_i dS[i][j] Qs[i][d]  (Qs already /8)
    for (int idx = t; idx < nq * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63;
        float s = 0.f;
        for (int j = 0; j < nk; ++j) s += Ps[i * LDP + j] * Ks[j * LDQ + d];
        dqkv[(size_t)(sq.q_start + i) * ld + head * HEAD_DIM + d] = s * 0.125f;
    }
    for (int idx = t; idx < nk * 64; idx += 256) {
        const int j = idx >> 6, d = idx & 63;
        float s = 0.f;
        for (int i = 0; i < nq; ++i) s += Ps[i * LDP + j] * Qs[i * LDQ + d];
        if (j < sq.pre_len && pre_ws) pre_ws[((size_t)blockIdx.x * max_pre + j) * 2 * width + head * HEAD_DIM + d] = s;
        else {
            const int row = j < sq.pre_len ? sq.pre_start + j : sq.q_start + j - sq.pre_len;
            atomicAdd(dqkv + (size_t)row * ld + width + head * HEAD_DIM + d, s);
        }
    }
}
// dK / dV of the shared prefix rows: one workgroup per RUN of consecutive sequences with the same prefix (a class bank's prompts, or
// one test sample's group of them) adds the parked contributions in sequence order on top of what the prefix sequence itself wrote
__global__ __launch_bounds__(256) void attention_bwd_prefix_reduce_kernel(const rlcf_seq* __restrict__ seqs, int n_seq, int max_pre, int width,
                                                                          const float* __restrict__ pre_ws, float* __restrict__ dqkv) {
    const int s0 = blockIdx.x, j = blockIdx.y;              // (sequence, prefix row)
    const rlcf_seq a = seqs[s0];
    if (j >= a.pre_len) return;
    if (s0 > 0) { const rlcf_seq b = seqs[s0 - 1]; if (b.pre_len == a.pre_len && b.pre_start == a.pre_start) return; }     // not the head of its run
    int s1 = s0 + 1;
    while (s1 < n_seq && seqs[s1].pre_len == a.pre_len && seqs[s1].pre_start == a.pre_start) ++s1;
    const int ld = 3 * width, per = 2 * width;
    for (int c = threadIdx.x; c < per; c += 256) {
        float* dst = dqkv + (size_t)(a.pre_start + j) * ld + width + c;
        const float* src = pre_ws + ((size_t)s0 * max_pre + j) * per + c;
        const size_t step = (size_t)max_pre * per;
        float acc = *dst;
        int s = s0;
        for (; s + 8 <= s1; s += 8) {                        // eight loads in flight, added in sequence order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s - s0 + u) * step];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; s < s1; ++s) acc += src[(size_t)(s - s0) * step];
        *dst = acc;
    }
}

int launch_attention_bwd(const float* qkv, const float* dout, const rlcf_seq* seqs, int n_seq, int max_keys, int width,
                         int causal, float* dqkv, hipStream_t st, float* pre_ws, size_t pre_ws_floats, int max_pre) {
    RLCF_ARG_CHECK(n_seq > 0 && width % HEAD_DIM == 0 && max_keys > 0 && max_keys <= ABWD_MAXK);
    const size_t bytes = (size_t)(4 * max_keys * 65 + max_keys * (max_keys + 1)) * sizeof(float);
    {
        const size_t cap = (size_t)(4 * ABWD_MAXK * 65 + ABWD_MAXK * (ABWD_MAXK + 1)) * sizeof(float);
        int rc_ = rlcf_func_lds((const void*)attention_bwd_kernel, cap);
        if (rc_ != RLCF_OK) return rc_;
    }
    if (max_pre <= 0 || !pre_ws || (size_t)n_seq * max_pre * 2 * width > pre_ws_floats) { pre_ws = nullptr; max_pre = 0; }     // (atomics: order-dependent sums)
    attention_bwd_kernel<<<dim3(n_seq, width / HEAD_DIM), dim3(256), bytes, st>>>(qkv, dout, seqs, width, causal, dqkv, pre_ws, max_pre);
    RLCF_LAUNCH_CHECK();
    if (pre_ws) {
        attention_bwd_prefix_reduce_kernel<<<dim3(n_seq, max_pre), dim3(256), 0, st>>>(seqs, n_seq, max_pre, width, pre_ws, dqkv);
        RLCF_LAUNCH_CHECK();
    }
    return RLCF_OK;
}

// ------------------------------------------------------------------------------------------
// Backward (dX only) for long sequences (image tower: 197 / 257 keys, LayerNorm-tuning path of
// TPT/tune_cls_rl.py): one 256-thread workgroup per (sequence, head, 32-query block); K/V streamed in
// 64-key chunks, the block's probability and dP rows resident in LDS.  dQ is stored, dK/dV are
// accumulated with atomicAdd (several query blocks and, with a shared prefix, several sequences hit
// the same key rows).  f32 VALU: the backward touches only the n_sel selected views.
#define ABL_MAXK 320
__global__ __launch_bounds__(256) void attention_bwd_long_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                 const rlcf_seq* __restrict__ seqs, int width, int causal,
                                                                 float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const rlcf_seq sq = seqs[blockIdx.y];
    const int head = blockIdx.z, q0 = blockIdx.x * 32;
    if (q0 >= sq.q_len) return;
    const int nk = sq.pre_len + sq.q_len, ld = 3 * width, t = threadIdx.x;
    const int LDP = ABL_MAXK + 1;
    float* Qs = sm;                  // [32][65]
    float* Gs = Qs + 32 * 65;        // [32][65]
    float* Ps = Gs + 32 * 65;        // [32][LDP]
    float* Ds = Ps + 32 * LDP;       // [32][LDP]  dP
    float* Kc = Ds + 32 * LDP;       // [64][65]
    float* Dr = Kc + 64 * 65;        // [32] row terms D_i
    for (int idx = t; idx < 32 * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63, qi = q0 + i;
        const bool ok = qi < sq.q_len;
        Qs[i * 65 + d] = ok ? qkv[(size_t)(sq.q_start + qi) * ld + head * HEAD_DIM + d] * 0.125f : 0.f;
        Gs[i * 65 + d] = ok ? dout[(size_t)(sq.q_start + qi) * width + head * HEAD_DIM + d] : 0.f;
    }
    auto load_chunk = [&](int kc, int which /*1 = K, 2 = V*/) {
        for (int idx = t; idx < 64 * 64; idx += 256) {
            const int j = idx >> 6, d = idx & 63, kap = kc + j;
            float v = 0.f;
            if (kap < nk) {
                const int row = kap < sq.pre_len ? sq.pre_start + kap : sq.q_start + kap - sq.pre_len;
                v = qkv[(size_t)row * ld + which * width + head * HEAD_DIM + d];
            }
            Kc[j * 65 + d] = v;
        }
    };
    // pass 1: scores
    for (int kc = 0; kc < nk; kc += 64) {
        __syncthreads();
        load_chunk(kc, 1);
        __syncthreads();
        for (int idx = t; idx < 32 * 64; idx += 256) {
            const int i = idx >> 6, j = idx & 63, kap = kc + j;
            float s = -INFINITY;
            if (kap < nk && (!causal || kap <= sq.pre_len + q0 + i)) {
                s = 0.f;
#pragma unroll 16
                for (int d = 0; d < 64; ++d) s += Qs[i * 65 + d] * Kc[j * 65 + d];
            }
            if (kap < ABL_MAXK) Ps[i * LDP + kap] = s;
        }
    }
    __syncthreads();
    if (t < 32) {
        float mx = -INFINITY;
        for (int j = 0; j < nk; ++j) mx = fmaxf(mx, Ps[t * LDP + j]);
        float sum = 0.f;
        for (int j = 0; j < nk; ++j) { const float e = expf(Ps[t * LDP + j] - mx); Ps[t * LDP + j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < nk; ++j) Ps[t * LDP + j] *= inv;
    }
    // pass 2: dP = dO V^T, dV += P^T dO
    for (int kc = 0; kc < nk; kc += 64) {
        __syncthreads();
        load_chunk(kc, 2);
        __syncthreads();
        for (int idx = t; idx < 32 * 64; idx += 256) {
            const int i = idx >> 6, j = idx & 63, kap = kc + j;
            if (kap >= nk) continue;
            float dp = 0.f;
#pragma unroll 16
            for (int d = 0; d < 64; ++d) dp += Gs[i * 65 + d] * Kc[j * 65 + d];
            Ds[i * LDP + kap] = dp;
        }
        for (int idx = t; idx < 64 * 64; idx += 256) {
            const int j = idx >> 6, d = idx & 63, kap = kc + j;
            if (kap >= nk) continue;
            float s = 0.f;
            for (int i = 0; i < 32; ++i) s += Ps[i * LDP + kap] * Gs[i * 65 + d];
            const int row = kap < sq.pre_len ? sq.pre_start + kap : sq.q_start + kap - sq.pre_len;
            atomicAdd(dqkv + (size_t)row * ld + 2 * width + head * HEAD_DIM + d, s);
        }
    }
    __syncthreads();
    if (t < 32) {
        float D = 0.f;
        for (int j = 0; j < nk; ++j) D += Ps[t * LDP + j] * Ds[t * LDP + j];
        Dr[t] = D;
    }
    __syncthreads();
    for (int idx = t; idx < 32 * nk; idx += 256) {
        const int i = idx / nk, j = idx - i * nk;
        Ps[i * LDP + j] = Ps[i * LDP + j] * (Ds[i * LDP + j] - Dr[i]);      // dS
    }
    // pass 3: dQ = dS K / 8, dK += dS^T Q / 8  (Qs already carries the 1/8)
    const int qi_t = t >> 3, d0 = (t & 7) * 8;
    float dq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] = 0.f;
    for (int kc = 0; kc < nk; kc += 64) {
        __syncthreads();
        load_chunk(kc, 1);
        __syncthreads();
        const int jn = min(64, nk - kc);
        for (int j = 0; j < jn; ++j) {
            const float ds = Ps[qi_t * LDP + kc + j];
#pragma unroll
            for (int e = 0; e < 8; ++e) dq[e] += ds * Kc[j * 65 + d0 + e];
        }
        for (int idx = t; idx < 64 * 64; idx += 256) {
            const int j = idx >> 6, d = idx & 63, kap = kc + j;
            if (kap >= nk) continue;
            float s = 0.f;
            for (int i = 0; i < 32; ++i) s += Ps[i * LDP + kap] * Qs[i * 65 + d];
            const int row = kap < sq.pre_len ? sq.pre_start + kap : sq.q_start + kap - sq.pre_len;
            atomicAdd(dqkv + (size_t)row * ld + width + head * HEAD_DIM + d, s);
        }
    }
    if (q0 + qi_t < sq.q_len) {
        float* p = dqkv + (size_t)(sq.q_start + q0 + qi_t) * ld + head * HEAD_DIM + d0;
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = dq[e] * 0.125f;
    }
}

int launch_attention_bwd_long(const float* qkv, const float* dout, const rlcf_seq* seqs, int n_seq, int max_q_len, int max_keys,
                              int width, int causal, float* dqkv, hipStream_t st) {
    RLCF_ARG_CHECK(n_seq > 0 && width % HEAD_DIM == 0 && max_keys > 0 && max_keys <= ABL_MAXK && max_q_len > 0);
    const size_t bytes = (size_t)(2 * 32 * 65 + 2 * 32 * (ABL_MAXK + 1) + 64 * 65 + 32) * sizeof(float);
    { int rc_ = rlcf_func_lds((const void*)attention_bwd_long_kernel, bytes); if (rc_ != RLCF_OK) return rc_; }
    dim3 grid((max_q_len + 31) / 32, n_seq, width / HEAD_DIM);
    RLCF_ARG_CHECK(grid.y <= 65535);
    attention_bwd_long_kernel<<<grid, dim3(256), bytes, st>>>(qkv, dout, seqs, width, causal, dqkv);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// ------------------------------------------------------------------------------------------
// Backward (dX only) for long sequences on the f32 matrix cores (v_mfma_f32_32x32x2_f32), flash-attention style: with the
// forward's log-sum-exp and output saved, one workgroup (4 waves) per (sequence, head, 32-query block) walks the keys in
// chunks of 128 — wave w owns the 32-key tile w of the chunk:
//   S = Q K^T, dP = dO V^T (two MFMA chains); P = exp(S - lse), dS = P o (dP - D) with D = rowsum(dO o O), in registers;
//   P and dS go through LDS once to become MFMA operands: dV += P^T dO, dK += dS^T Q (atomicAdd: other query blocks and,
//   with a shared prefix, other sequences hit the same key rows), dQ += dS K (registers, reduced over the 4 waves at the end).
// Same contract as attention_bwd_long_kernel (which it replaces when lse / out are available): dqkv pre-zeroed, <= 320 keys not
// required here (any length).
#define ABM_LD 65
#define ABM_PLD 129
__global__ __launch_bounds__(256) void attention_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                 const float* __restrict__ lse, const float* __restrict__ dout,
                                                                 const rlcf_seq* __restrict__ seqs, int width, int causal,
                                                                 float* __restrict__ dqkv, float* __restrict__ park, int park_rows) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const rlcf_seq sq = seqs[blockIdx.y];
    const int head = blockIdx.z, q0 = blockIdx.x * 32;
    if (q0 >= sq.q_len) return;
    const int nk = sq.pre_len + sq.q_len, ld = 3 * width, H = width / HEAD_DIM, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, l32 = lane & 31, h = lane >> 5;
    float* Qs = sm;                          // [32][65]  Q / 8
    float* Gs = Qs + 32 * ABM_LD;            // [32][65]  dO
    float* Kc = Gs + 32 * ABM_LD;            // [128][65]
    float* Vc = Kc + 128 * ABM_LD;           // [128][65]
    float* Pt = Vc + 128 * ABM_LD;           // [32][129] P
    float* St = Pt + 32 * ABM_PLD;           // [32][129] dS
    float* Ls = St + 32 * ABM_PLD;           // [32] lse
    float* Dr = Ls + 32;                     // [32] D
    for (int idx = t; idx < 32 * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63, qi = q0 + i;
        const bool ok = qi < sq.q_len;
        Qs[i * ABM_LD + d] = ok ? qkv[(size_t)(sq.q_start + qi) * ld + head * HEAD_DIM + d] * 0.125f : 0.f;
        Gs[i * ABM_LD + d] = ok ? dout[(size_t)(sq.q_start + qi) * width + head * HEAD_DIM + d] : 0.f;
    }
    if (t < 32) {
        const int qi = q0 + t;
        float D = 0.f, l = 0.f;
        if (qi < sq.q_len) {
            const float* o = out + (size_t)(sq.q_start + qi) * width + head * HEAD_DIM;
            const float* g = dout + (size_t)(sq.q_start + qi) * width + head * HEAD_DIM;
            for (int d = 0; d < 64; ++d) D += o[d] * g[d];
            l = lse[(size_t)(sq.q_start + qi) * H + head];
        }
        Dr[t] = D; Ls[t] = l;
    }
    f32x16 dq0, dq1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
    for (int kc = 0; kc < nk; kc += 128) {
        __syncthreads();
        for (int idx = t; idx < 128 * 64; idx += 256) {
            const int j = idx >> 6, d = idx & 63, kap = kc + j;
            float kv = 0.f, vv = 0.f;
            if (kap < nk) {
                const int row = kap < sq.pre_len ? sq.pre_start + kap : sq.q_start + kap - sq.pre_len;
                const float* p = qkv + (size_t)row * ld + head * HEAD_DIM + d;
                kv = p[width]; vv = p[2 * width];
            }
            Kc[j * ABM_LD + d] = kv; Vc[j * ABM_LD + d] = vv;
        }
        __syncthreads();
        const int kt = kc + 32 * wave;                    // this wave's 32 keys
        const bool live = kt < nk;                        // (wave-uniform)
        if (live) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
            const float* kq = Kc + (32 * wave + l32) * ABM_LD + h;
            const float* vq = Vc + (32 * wave + l32) * ABM_LD + h;
            const float* qq = Qs + l32 * ABM_LD + h;
            const float* gq = Gs + l32 * ABM_LD + h;
#pragma unroll 8
            for (int st = 0; st < 32; ++st) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qq[2 * st], kq[2 * st], sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(gq[2 * st], vq[2 * st], pacc, 0, 0, 0);
            }
            const int key = kt + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma32_row(r, h);
                const bool ok = key < nk && q0 + row < sq.q_len && (!causal || key <= sq.pre_len + q0 + row);
                const float p = ok ? expf(sacc[r] - Ls[row]) : 0.f;
                Pt[row * ABM_PLD + 32 * wave + l32] = p;
                St[row * ABM_PLD + 32 * wave + l32] = p * (pacc[r] - Dr[row]);
            }
        }
        __syncthreads();
        if (live) {
            // dV / dK tiles [32 keys x 32 d] x 2 (contraction over the 32 queries), dQ partial [32 q x 32 d] x 2 (over these 32 keys)
            f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
            const float* pa = Pt + h * ABM_PLD + 32 * wave + l32;          // A[m = key][k = q]
            const float* sa = St + h * ABM_PLD + 32 * wave + l32;
            const float* gb = Gs + h * ABM_LD + l32;                       // B[k = q][n = d]
            const float* qb = Qs + h * ABM_LD + l32;
#pragma unroll 8
            for (int st = 0; st < 16; ++st) {
                const float pv = pa[2 * st * ABM_PLD], sv = sa[2 * st * ABM_PLD];
                dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, gb[2 * st * ABM_LD], dv0, 0, 0, 0);
                dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, gb[2 * st * ABM_LD + 32], dv1, 0, 0, 0);
                dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, qb[2 * st * ABM_LD], dk0, 0, 0, 0);
                dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, qb[2 * st * ABM_LD + 32], dk1, 0, 0, 0);
            }
            const float* sq_a = St + l32 * ABM_PLD + 32 * wave + h;        // A[m = q][k = key]
            const float* kb = Kc + (32 * wave + h) * ABM_LD + l32;         // B[k = key][n = d]
#pragma unroll 8
            for (int st = 0; st < 16; ++st) {
                const float a = sq_a[2 * st];
                dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, kb[2 * st * ABM_LD], dq0, 0, 0, 0);
                dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, kb[2 * st * ABM_LD + 32], dq1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kap = kt + mfma32_row(r, h);
                if (kap < nk) {
                    if (park) {        // (no shared prefix) this query block's contribution to key kap, parked [seq][q block][key][K | V] and added
                                       // in block order by attention_bwd_park_reduce_kernel: bit-reproducible, no zero fill of the K / V parts
                        float* base = park + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * park_rows + kap) * (2 * width) + head * HEAD_DIM + l32;
                        base[width] = dv0[r]; base[width + 32] = dv1[r];
                        base[0] = dk0[r];     base[32] = dk1[r];
                    } else {
                        const int row = kap < sq.pre_len ? sq.pre_start + kap : sq.q_start + kap - sq.pre_len;
                        float* base = dqkv + (size_t)row * ld + head * HEAD_DIM + l32;
                        atomicAdd(base + 2 * width, dv0[r]); atomicAdd(base + 2 * width + 32, dv1[r]);
                        atomicAdd(base + width, dk0[r]);     atomicAdd(base + width + 32, dk1[r]);
                    }
                }
            }
        }
    }
    // dQ: sum of the four waves' partials (through the K chunk buffer), scaled by 1/8
    __syncthreads();
    float* red = Kc + wave * (32 * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[mfma32_row(r, h) * 64 + l32] = dq0[r];
        red[mfma32_row(r, h) * 64 + 32 + l32] = dq1[r];
    }
    __syncthreads();
    for (int idx = t; idx < 32 * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63;
        if (q0 + i < sq.q_len) {
            const float v = (Kc[idx] + Kc[2048 + idx]) + (Kc[4096 + idx] + Kc[6144 + idx]);
            dqkv[(size_t)(sq.q_start + q0 + i) * ld + head * HEAD_DIM + d] = v * 0.125f;
        }
    }
}

// park (optional; sequences WITHOUT a shared prefix and without a causal mask only — the image towers): n_seq * ceil(max_q_len / 32) * max_q_len
// * 2 * width floats; with it dK / dV are bit-reproducible run to run (as launch_attention_bwd_x3) and dqkv needs no zero fill
int launch_attention_bwd_mfma(const float* qkv, const float* out, const float* lse, const float* dout, const rlcf_seq* seqs, int n_seq,
                              int max_q_len, int width, int causal, float* dqkv, hipStream_t st, float* park) {
    RLCF_ARG_CHECK(n_seq > 0 && width % HEAD_DIM == 0 && max_q_len > 0 && qkv && out && lse && dout && dqkv);
    const size_t bytes = (size_t)(2 * 32 * ABM_LD + 2 * 128 * ABM_LD + 2 * 32 * ABM_PLD + 64) * sizeof(float);
    { int rc_ = rlcf_func_lds((const void*)attention_bwd_mfma_kernel, bytes); if (rc_ != RLCF_OK) return rc_; }
    dim3 grid((max_q_len + 31) / 32, n_seq, width / HEAD_DIM);
    RLCF_ARG_CHECK(grid.y <= 65535);
    attention_bwd_mfma_kernel<<<grid, dim3(256), bytes, st>>>(qkv, out, lse, dout, seqs, width, causal, dqkv, park, max_q_len);
    RLCF_LAUNCH_CHECK();
    if (park) return launch_attention_bwd_park_reduce(park, seqs, n_seq, max_q_len, (int)grid.x, width, dqkv, st);
    return RLCF_OK;
}
