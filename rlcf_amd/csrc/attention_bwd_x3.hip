// Attention backward (dX only) for long sequences on the f16 matrix cores with f32-grade accuracy (split-f16: every operand is
// hi + lo, three v_mfma_f32_32x32x16_f16 per product — see gemm_f16x3.hip / attention_x3.hip), flash-attention style: the forward's
// log-sum-exp and output are saved, nothing of size [queries, keys] is stored.  Replaces attention_bwd_mfma_kernel (the same algorithm
// on v_mfma_f32_32x32x2_f32) when the engine runs in RLCF_PREC_F16X3; the backward of nn.MultiheadAttention (TPT/clip/model.py:175,
// 185-187) under the LayerNorm / image-encoder tuning paths (TPT/tune_cls_rl.py:206-227).
//
// One workgroup (4 waves) per (sequence, head, 32-query block) walks the keys in chunks of 64 = two 32-key tiles.  Each tile is
// worked on by TWO waves in different roles, because the five products need the score tile in both orientations and an MFMA
// accumulator can only be re-used as an operand along its register dimension:
//   role B (waves 0, 1):  S = Q K^T and dP = dO V^T with queries in the accumulator ROWS (registers) and keys in the lanes ->
//                         P = exp(S - lse), dS = P o (dP - D) are A operands (row = key lane, contraction over the queries) of
//                         dV += P^T dO  and  dK += dS^T Q        (B operands: the transposed dO / Q tiles of the block, in LDS);
//   role A (waves 2, 3):  the same two products with the operands swapped (keys in the rows, queries in the lanes: lse and D are
//                         per-lane scalars) -> dS is the A operand (row = query lane, contraction over the keys) of
//                         dQ += dS K                              (B operand: the transposed K tile of the chunk, in LDS).
// Operand registers of Q and dO are the same in both roles (lane (row, half) holds 8 consecutive d).  dV / dK leave through
// atomicAdd (other query blocks — and, with a shared prefix, other sequences — hit the same key rows), coalesced: rows = keys,
// lanes = d.  dQ is summed over the two role-A waves at the end.
// Ranges: Q, K, V enter at full scale (the 1/8 of the scores sits in the exponent and in the dK / dQ output factors); dO is multiplied by a power of two found on the device from max|dO| (gradients sit far
// below f16's normal range) that is divided out of the three results; P is carried times 2^6 as in the forward.
#include "kernels.h"

#define BX_KLD 72      // halves per row of the [key][d] tiles (64 + 8 pad = 144 B: conflict-free ds_read_b128 fragments)
#define BX_TLD 40      // halves per row of the transposed [d][slot] tiles (32 + 8 pad = 80 B)
#define BX_CHUNK 64

typedef uint32_t bx_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bx_split8(const float* v, h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}
// position of row q (0..31) of a 32x32 accumulator in its register order: register 8*tt + e of lane-half h holds row
// (e & 3) + 8 * (2 * tt + (e >> 2)) + 4 * h, i.e. slot 16 * tt + 8 * h + e
__device__ __forceinline__ int bx_slot(int q) { return ((q >> 4) << 4) | (((q >> 2) & 1) << 3) | (((q >> 3) & 1) << 2) | (q & 3); }

#define BX_MMA3(acc, ah, al, bh, bl)                                          \
    do {                                                                      \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);   \
    } while (0)

__global__ __launch_bounds__(256, 2) void attention_bwd_x3_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                  const float* __restrict__ lse, const float* __restrict__ dout,
                                                                  const unsigned int* __restrict__ amax_dout,
                                                                  const rlcf_seq* __restrict__ seqs, int width, int causal,
                                                                  float* __restrict__ dqkv, float* __restrict__ park, int park_rows) {
    extern __shared__ __attribute__((aligned(16))) char bx_smem[];
    const rlcf_seq sq = seqs[blockIdx.y];
    const int head = blockIdx.z, q0 = blockIdx.x * 32;
    if (q0 >= sq.q_len) return;
    const int nk = sq.pre_len + sq.q_len, ld = 3 * width, H = width / HEAD_DIM, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, l32 = lane & 31, h = lane >> 5;
    const int kt = wave & 1, role_a = wave >> 1;
    // LDS: transposed Q / dO of the block (hi, lo each), then the chunk: K, V [64][72] (hi, lo), K^T two tiles [64][40] (hi, lo)
    _Float16* QT_h = (_Float16*)bx_smem;
    _Float16* QT_l = QT_h + 64 * BX_TLD;
    _Float16* GT_h = QT_l + 64 * BX_TLD;
    _Float16* GT_l = GT_h + 64 * BX_TLD;
    _Float16* K_h = GT_l + 64 * BX_TLD;
    _Float16* K_l = K_h + BX_CHUNK * BX_KLD;
    _Float16* V_h = K_l + BX_CHUNK * BX_KLD;
    _Float16* V_l = V_h + BX_CHUNK * BX_KLD;
    _Float16* KT_h = V_l + BX_CHUNK * BX_KLD;              // [2 tiles][64 d][40]
    _Float16* KT_l = KT_h + 2 * 64 * BX_TLD;
    float* Ls = (float*)(KT_l + 2 * 64 * BX_TLD);           // [32] lse of the block's queries
    float* Ds = Ls + 32;                                    // [32] D * gscale

    // power of two lifting max|dO| into [2^3, 2^4): dP = dO V^T and dS = P o (dP - D) pass through f16 pairs too and need the headroom
    // (|dP| <= 64 * max|dO| * max|v|)
    float gscale, inv_gscale;
    {
        const float am = __uint_as_float(*amax_dout);
        int sh = 0;
        if (am > 0.f && am < INFINITY) sh = max(-40, min(60, 3 - (int)floorf(log2f(am))));
        gscale = ldexpf(1.0f, sh); inv_gscale = ldexpf(1.0f, -sh);
    }
    // operand registers of the block's queries: lane (query l32, half h) holds d = ks*16 + h*8 + [0,8) of Q/8 and of dO*gscale
    const int qi = min(q0 + l32, sq.q_len - 1);
    const bool q_ok = q0 + l32 < sq.q_len;
    h16x8 qh[4], ql[4], gh[4], gl[4];
    float lse_q = 0.f, D_q = 0.f;
    {
        const float* qp = qkv + (size_t)(sq.q_start + qi) * ld + head * HEAD_DIM + h * 8;
        const float* gp = dout + (size_t)(sq.q_start + qi) * width + head * HEAD_DIM + h * 8;
        const float* op = out + (size_t)(sq.q_start + qi) * width + head * HEAD_DIM + h * 8;
        float Dp = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 a = *(const float4*)(qp + ks * 16), b = *(const float4*)(qp + ks * 16 + 4);
            const float4 c = *(const float4*)(gp + ks * 16), d = *(const float4*)(gp + ks * 16 + 4);
            const float4 o0 = *(const float4*)(op + ks * 16), o1 = *(const float4*)(op + ks * 16 + 4);
            const float z = q_ok ? 1.f : 0.f;
            const float qv[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z};
            const float gz = gscale * z;
            const float gv[8] = {c.x * gz, c.y * gz, c.z * gz, c.w * gz, d.x * gz, d.y * gz, d.z * gz, d.w * gz};
            bx_split8(qv, qh[ks], ql[ks]);
            bx_split8(gv, gh[ks], gl[ks]);
            Dp += c.x * o0.x + c.y * o0.y + c.z * o0.z + c.w * o0.w + d.x * o1.x + d.y * o1.y + d.z * o1.z + d.w * o1.w;
        }
        Dp += __shfl_xor(Dp, 32);
        D_q = q_ok ? Dp * gscale : 0.f;
        lse_q = q_ok ? lse[(size_t)(sq.q_start + qi) * H + head] : 0.f;
        if (wave == 0 && h == 0) { Ls[l32] = lse_q; Ds[l32] = D_q; }
    }
    // transposed tiles of the block: thread -> (query t / 8, d part (t % 8) * 8)
    {
        const int q = t >> 3, d0 = (t & 7) * 8, qq = min(q0 + q, sq.q_len - 1), sl = bx_slot(q);
        const float z = q0 + q < sq.q_len ? 1.f : 0.f;
        const float* qp = qkv + (size_t)(sq.q_start + qq) * ld + head * HEAD_DIM + d0;
        const float* gp = dout + (size_t)(sq.q_start + qq) * width + head * HEAD_DIM + d0;
        const float4 a = *(const float4*)qp, b = *(const float4*)(qp + 4), c = *(const float4*)gp, d = *(const float4*)(gp + 4);
        const float qz = z, gz = gscale * z;
        const float qv[8] = {a.x * qz, a.y * qz, a.z * qz, a.w * qz, b.x * qz, b.y * qz, b.z * qz, b.w * qz};
        const float gv[8] = {c.x * gz, c.y * gz, c.z * gz, c.w * gz, d.x * gz, d.y * gz, d.z * gz, d.w * gz};
        h16x8 xh, xl, yh, yl;
        bx_split8(qv, xh, xl);
        bx_split8(gv, yh, yl);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            QT_h[(d0 + e) * BX_TLD + sl] = xh[e]; QT_l[(d0 + e) * BX_TLD + sl] = xl[e];
            GT_h[(d0 + e) * BX_TLD + sl] = yh[e]; GT_l[(d0 + e) * BX_TLD + sl] = yl[e];
        }
    }
    __syncthreads();
    f32x16 dq0, dq1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
    constexpr float LOG2E = 1.44269504088896341f, SC = 0.125f * LOG2E;       // scores = Q K^T / 8: Q stays at full scale (its lo parts stay normal f16)
    const int kend = causal ? min(nk, sq.pre_len + min(q0 + 32, sq.q_len)) : nk;     // keys past the block's last query see nothing

    // The two roles run their own copy of the chunk loop (same barriers in both: every wave of the workgroup passes the same number
    // of them) so that each keeps only its own accumulators in registers.
    if (!role_a) {
    for (int kc = 0; kc < kend; kc += BX_CHUNK) {
        __syncthreads();
        // stage the chunk: thread -> (key t / 4, d part (t % 4) * 16)
        {
            const int j = t >> 2, d0 = (t & 3) * 16, kap = kc + j;
            const bool ok = kap < nk;
            const int kcl = min(kap, nk - 1);
            const int row = kcl < sq.pre_len ? sq.pre_start + kcl : sq.q_start + kcl - sq.pre_len;
            const float* p = qkv + (size_t)row * ld + head * HEAD_DIM + d0 + width;
            const float z = ok ? 1.f : 0.f;
            const int tile = j >> 5, sl = bx_slot(j & 31);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const float4 a = *(const float4*)(p + part * 8), b = *(const float4*)(p + part * 8 + 4);
                const float4 c = *(const float4*)(p + width + part * 8), d = *(const float4*)(p + width + part * 8 + 4);
                const float kv[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z};
                const float vv[8] = {c.x * z, c.y * z, c.z * z, c.w * z, d.x * z, d.y * z, d.z * z, d.w * z};
                h16x8 kh8, kl8, vh8, vl8;
                bx_split8(kv, kh8, kl8);
                bx_split8(vv, vh8, vl8);
                *(h16x8*)(K_h + j * BX_KLD + d0 + part * 8) = kh8; *(h16x8*)(K_l + j * BX_KLD + d0 + part * 8) = kl8;
                *(h16x8*)(V_h + j * BX_KLD + d0 + part * 8) = vh8; *(h16x8*)(V_l + j * BX_KLD + d0 + part * 8) = vl8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    KT_h[(tile * 64 + d0 + part * 8 + e) * BX_TLD + sl] = kh8[e];
                    KT_l[(tile * 64 + d0 + part * 8 + e) * BX_TLD + sl] = kl8[e];
                }
            }
        }
        __syncthreads();
        const int k0 = kc + 32 * kt;                        // first key of this wave's tile
        if (k0 >= kend) continue;                           // (wave-uniform; the barriers are at the loop head)
        const _Float16* kh_ = K_h + (32 * kt + l32) * BX_KLD + h * 8;
        const _Float16* kl_ = K_l + (32 * kt + l32) * BX_KLD + h * 8;
        const _Float16* vh_ = V_h + (32 * kt + l32) * BX_KLD + h * 8;
        const _Float16* vl_ = V_l + (32 * kt + l32) * BX_KLD + h * 8;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            // ---- role B: rows = queries, lanes = keys
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kh8 = *(const h16x8*)(kh_ + ks * 16), kl8 = *(const h16x8*)(kl_ + ks * 16);
                const h16x8 vh8 = *(const h16x8*)(vh_ + ks * 16), vl8 = *(const h16x8*)(vl_ + ks * 16);
                BX_MMA3(s, qh[ks], ql[ks], kh8, kl8);
                BX_MMA3(dp, gh[ks], gl[ks], vh8, vl8);
            }
            const int key = k0 + l32;
            float pv[16], sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma32_row(r, h);
                const bool ok = key < nk && q0 + row < sq.q_len && (!causal || key <= sq.pre_len + q0 + row);
                const float p = ok ? __builtin_amdgcn_exp2f(s[r] * SC - Ls[row] * LOG2E) : 0.f;   // (lse / D of the accumulator's query rows: LDS)
                pv[r] = p * 64.0f;
                sv[r] = p * (dp[r] - Ds[row]);
            }
            h16x8 ph[2], pl[2], sh[2], sl_[2];
            bx_split8(pv, ph[0], pl[0]); bx_split8(pv + 8, ph[1], pl[1]);
            bx_split8(sv, sh[0], sl_[0]); bx_split8(sv + 8, sh[1], sl_[1]);
            f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int o0 = l32 * BX_TLD + tt * 16 + h * 8, o1 = (32 + l32) * BX_TLD + tt * 16 + h * 8;
                const h16x8 g0h = *(const h16x8*)(GT_h + o0), g0l = *(const h16x8*)(GT_l + o0);
                const h16x8 g1h = *(const h16x8*)(GT_h + o1), g1l = *(const h16x8*)(GT_l + o1);
                const h16x8 q0h = *(const h16x8*)(QT_h + o0), q0l = *(const h16x8*)(QT_l + o0);
                const h16x8 q1h = *(const h16x8*)(QT_h + o1), q1l = *(const h16x8*)(QT_l + o1);
                BX_MMA3(dv0, ph[tt], pl[tt], g0h, g0l);
                BX_MMA3(dv1, ph[tt], pl[tt], g1h, g1l);
                BX_MMA3(dk0, sh[tt], sl_[tt], q0h, q0l);
                BX_MMA3(dk1, sh[tt], sl_[tt], q1h, q1l);
            }
            const float fv = inv_gscale * 0.015625f, fk = inv_gscale * 0.125f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kap = k0 + mfma32_row(r, h);
                if (kap < nk) {
                    if (park) {        // (no shared prefix) this query block's contribution to key kap, parked: [seq][q block][key][K | V]
                        float* base = park + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * park_rows + kap) * (2 * width) + head * HEAD_DIM + l32;
                        base[width] = dv0[r] * fv; base[width + 32] = dv1[r] * fv;
                        base[0] = dk0[r] * fk;     base[32] = dk1[r] * fk;
                    } else {
                        const int row = kap < sq.pre_len ? sq.pre_start + kap : sq.q_start + kap - sq.pre_len;
                        float* base = dqkv + (size_t)row * ld + head * HEAD_DIM + l32;
                        atomicAdd(base + 2 * width, dv0[r] * fv); atomicAdd(base + 2 * width + 32, dv1[r] * fv);
                        atomicAdd(base + width, dk0[r] * fk);     atomicAdd(base + width + 32, dk1[r] * fk);
                    }
                }
            }
    }
    } else {
    for (int kc = 0; kc < kend; kc += BX_CHUNK) {
        __syncthreads();
        // stage the chunk: thread -> (key t / 4, d part (t % 4) * 16)
        {
            const int j = t >> 2, d0 = (t & 3) * 16, kap = kc + j;
            const bool ok = kap < nk;
            const int kcl = min(kap, nk - 1);
            const int row = kcl < sq.pre_len ? sq.pre_start + kcl : sq.q_start + kcl - sq.pre_len;
            const float* p = qkv + (size_t)row * ld + head * HEAD_DIM + d0 + width;
            const float z = ok ? 1.f : 0.f;
            const int tile = j >> 5, sl = bx_slot(j & 31);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const float4 a = *(const float4*)(p + part * 8), b = *(const float4*)(p + part * 8 + 4);
                const float4 c = *(const float4*)(p + width + part * 8), d = *(const float4*)(p + width + part * 8 + 4);
                const float kv[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z};
                const float vv[8] = {c.x * z, c.y * z, c.z * z, c.w * z, d.x * z, d.y * z, d.z * z, d.w * z};
                h16x8 kh8, kl8, vh8, vl8;
                bx_split8(kv, kh8, kl8);
                bx_split8(vv, vh8, vl8);
                *(h16x8*)(K_h + j * BX_KLD + d0 + part * 8) = kh8; *(h16x8*)(K_l + j * BX_KLD + d0 + part * 8) = kl8;
                *(h16x8*)(V_h + j * BX_KLD + d0 + part * 8) = vh8; *(h16x8*)(V_l + j * BX_KLD + d0 + part * 8) = vl8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    KT_h[(tile * 64 + d0 + part * 8 + e) * BX_TLD + sl] = kh8[e];
                    KT_l[(tile * 64 + d0 + part * 8 + e) * BX_TLD + sl] = kl8[e];
                }
            }
        }
        __syncthreads();
        const int k0 = kc + 32 * kt;                        // first key of this wave's tile
        if (k0 >= kend) continue;                           // (wave-uniform; the barriers are at the loop head)
        const _Float16* kh_ = K_h + (32 * kt + l32) * BX_KLD + h * 8;
        const _Float16* kl_ = K_l + (32 * kt + l32) * BX_KLD + h * 8;
        const _Float16* vh_ = V_h + (32 * kt + l32) * BX_KLD + h * 8;
        const _Float16* vl_ = V_l + (32 * kt + l32) * BX_KLD + h * 8;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            // ---- role A: rows = keys, lanes = queries
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kh8 = *(const h16x8*)(kh_ + ks * 16), kl8 = *(const h16x8*)(kl_ + ks * 16);
                const h16x8 vh8 = *(const h16x8*)(vh_ + ks * 16), vl8 = *(const h16x8*)(vl_ + ks * 16);
                BX_MMA3(s, kh8, kl8, qh[ks], ql[ks]);
                BX_MMA3(dp, vh8, vl8, gh[ks], gl[ks]);
            }
            float sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + mfma32_row(r, h);
                const bool ok = key < nk && q_ok && (!causal || key <= sq.pre_len + q0 + l32);
                const float p = ok ? __builtin_amdgcn_exp2f(s[r] * SC - lse_q * LOG2E) : 0.f;
                sv[r] = p * (dp[r] - D_q);
            }
            h16x8 sh[2], sl_[2];
            bx_split8(sv, sh[0], sl_[0]); bx_split8(sv + 8, sh[1], sl_[1]);
            const _Float16* kth = KT_h + kt * 64 * BX_TLD;
            const _Float16* ktl = KT_l + kt * 64 * BX_TLD;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int o0 = l32 * BX_TLD + tt * 16 + h * 8, o1 = (32 + l32) * BX_TLD + tt * 16 + h * 8;
                const h16x8 k0h = *(const h16x8*)(kth + o0), k0l = *(const h16x8*)(ktl + o0);
                const h16x8 k1h = *(const h16x8*)(kth + o1), k1l = *(const h16x8*)(ktl + o1);
                BX_MMA3(dq0, sh[tt], sl_[tt], k0h, k0l);
                BX_MMA3(dq1, sh[tt], sl_[tt], k1h, k1l);
            }
    }
    }
    // dQ: the two role-A waves' partials summed through LDS (the chunk area is free now), times 1/8 (scores = Q K^T / 8) / gscale
    __syncthreads();
    float* red = (float*)K_h;                               // [2][32][64] floats = 16 KB
    if (role_a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            red[(kt * 32 + mfma32_row(r, h)) * 64 + l32] = dq0[r];
            red[(kt * 32 + mfma32_row(r, h)) * 64 + 32 + l32] = dq1[r];
        }
    }
    __syncthreads();
    const float fq = 0.125f * inv_gscale;
    for (int idx = t; idx < 32 * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63;
        if (q0 + i < sq.q_len) dqkv[(size_t)(sq.q_start + q0 + i) * ld + head * HEAD_DIM + d] = (red[idx] + red[2048 + idx]) * fq;
    }
}

// dK / dV of sequences without a shared prefix: the contributions of a sequence's query blocks, parked by the kernel, added in block order
__global__ __launch_bounds__(256) void attention_bwd_park_reduce_kernel(const float* __restrict__ park, const rlcf_seq* __restrict__ seqs,
                                                                        int park_rows, int n_qb, int width, float* __restrict__ dqkv) {
    const rlcf_seq sq = seqs[blockIdx.y];
    const int key = blockIdx.x;
    if (key >= sq.q_len) return;
    const int nqb = (sq.q_len + 31) / 32, w4 = 2 * width / 4;
    float4* dst = (float4*)(dqkv + (size_t)(sq.q_start + key) * 3 * width + width);
    for (int c = threadIdx.x; c < w4; c += 256) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < nqb; ++b) {
            const float4 v = ((const float4*)(park + (((size_t)blockIdx.y * n_qb + b) * park_rows + key) * (2 * width)))[c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        dst[c] = a;
    }
}

// amax_dout: device scalar (bit pattern of a non-negative float), max |dout| over the whole matrix (launch_absmax)
// park (optional; sequences WITHOUT a shared prefix only: the image towers): n_seq * ceil(max_q_len / 32) * max_q_len * 2 * width floats;
// with it dK / dV are bit-reproducible (parked per query block, added in block order) and dqkv needs no zero fill
int launch_attention_bwd_x3(const float* qkv, const float* out, const float* lse, const float* dout, const float* amax_dout, const rlcf_seq* seqs,
                            int n_seq, int max_q_len, int width, int causal, float* dqkv, hipStream_t st, float* park) {
    RLCF_ARG_CHECK(n_seq > 0 && width % HEAD_DIM == 0 && max_q_len > 0 && qkv && out && lse && dout && dqkv && amax_dout);
    const size_t bytes = (size_t)(4 * 64 * BX_TLD + 4 * BX_CHUNK * BX_KLD + 2 * 2 * 64 * BX_TLD) * sizeof(_Float16) + 64 * sizeof(float);
    { int rc_ = rlcf_func_lds((const void*)attention_bwd_x3_kernel, bytes); if (rc_ != RLCF_OK) return rc_; }
    dim3 grid((max_q_len + 31) / 32, n_seq, width / HEAD_DIM);
    RLCF_ARG_CHECK(grid.y <= 65535);
    attention_bwd_x3_kernel<<<grid, dim3(256), bytes, st>>>(qkv, out, lse, dout, (const unsigned int*)amax_dout, seqs, width, causal, dqkv, park,
                                                            max_q_len);
    RLCF_LAUNCH_CHECK();
    if (park) return launch_attention_bwd_park_reduce(park, seqs, n_seq, max_q_len, (int)grid.x, width, dqkv, st);
    return RLCF_OK;
}
int launch_attention_bwd_park_reduce(const float* park, const rlcf_seq* seqs, int n_seq, int park_rows, int n_qb, int width, float* dqkv, hipStream_t st) {
    attention_bwd_park_reduce_kernel<<<dim3(park_rows, n_seq), dim3(256), 0, st>>>(park, seqs, park_rows, n_qb, width, dqkv);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
