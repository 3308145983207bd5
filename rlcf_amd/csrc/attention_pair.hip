// Attention forward of the image towers on PRODUCER-EMITTED operands (split-f16, see gemm_f16x3.hip / attention_x3.hip).
//
// attention_x3.hip reads Q / K / V as f32 from the in_proj output, splits every K / V chunk into f16 hi / lo with VALU code and
// writes it to LDS (V transposed, 2-byte scattered stores) once per workgroup and chunk: the round-2 ablation priced that staging
// at 56 % of the kernel.  Here the in_proj GEMM's epilogue has already written Q, K and V as interleaved f16 pairs
// (row of a head = 256 B: [hi d0-31 | lo d0-31 | hi d32-63 | lo d32-63]; RLCF_PREC_F16: a plain f16 row of 128 B), so
//   * K / V chunks go global -> LDS by DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging registers, no VALU, no
//     ds_write) into a 2-stage ring of 64-key stages: ONE barrier per 64 keys, the DMA of chunk c+1 in flight under chunk c;
//   * the LDS image of a stage is [quad][key][64 B] for K and for V, quad = one 64-B quarter of the key's row (hi / lo of d 0-31,
//     hi / lo of d 32-63): a DMA piece moves 16 keys x 64 B.  Four consecutive keys of a quad fill one 256-B bank row, so the
//     transposing V read (4 keys x 64 B per 32 lanes) is conflict free as stored, and the K operand's ds_read_b128 (16 keys, one
//     16-B slot) is conflict free with the slot XOR-ed by (key>>2)&3 on the DMA SOURCE address (the DMA writes LDS lane-linearly).
//     Every operand address is then ONE lane register + an immediate (quad, key block, k-step): no per-read address arithmetic
//     (SQ_LDS_BANK_CONFLICT = 0);
//   * the V^T operand of O^T = V^T.P^T comes out of ROW-major V with ds_read_b64_tr_b16 (gfx950): a 16-lane group reads a
//     [4 keys][16 d] block and every lane receives one d column of it — two such reads are exactly the 8 keys (e&3) + 8*(e>>2) (+ 4h
//     + 16tt) a lane's slice of the S^T accumulator holds, so P still feeds the second MFMA straight from registers.
// Same arithmetic as attention_fwd_x3_kernel (three v_mfma_f32_32x32x16_f16 per product into one f32 accumulator, P carried times
// 2^6, online softmax lane-local through v_exp_f32): results agree to the last bits of the f32 accumulation order.
//
// One workgroup per (sequence, head, group of NW query blocks); every wave issues its share of the DMA pieces and computes one block
// of 32 queries.  Measured alternatives that LOST (profiles/r3_attention_experiments.txt): persistent workgroups with a dedicated
// producer wave (barrier hand-shake, or LDS flags without any barrier), a persistent loop of this kernel with the chunk stream
// running across items, and a 4-stage ring of 32-key stages — the hardware's fresh-workgroup dispatch balances the CUs better than any
// static item walk, and the kernel sits between its two rooflines (HBM: Q, K, V pairs in and O pairs out = 3.1 GB per ViT-B/16 layer
// at 1 280 sequences = 0.55 ms; matrix pipe: 0.27 ms), not on a latency that more prefetch depth would hide.
// Non-causal sequences with an optional prefix (the class-token-only last block: one query, prefix = the other rows).
// Replaces nn.MultiheadAttention's core (TPT/clip/model.py:175,185-187) for VisionTransformer towers in RLCF_PREC_F16X3 / _F16.
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// ds_read_b64_tr_b16 as inline asm: through the builtin hipcc cannot tell the read from the LDS-DMA writes still in flight for the NEXT
// stage and drains them (s_waitcnt vmcnt(0)) in front of every first read of a k-step, which serialises the ring.  An asm load is
// invisible to hipcc's counters: AP_TR_WAIT* names every destination "+v", so nothing that consumes them can be scheduled above the
// wait; extra outstanding LDS operations only make hipcc's own (in-order) lgkmcnt waits stricter, never weaker.
#define AP_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define AP_TR_WAIT8(a, b, c, d, e, f, g, h_)                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h_))
#define AP_TR_WAIT4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
__device__ __forceinline__ h16x8 ap_cat(u32x2 x, u32x2 y) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {x[0], x[1], y[0], y[1]};
    return __builtin_bit_cast(h16x8, v);
}
__device__ __forceinline__ void ap_split8(const float* v, h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}
template <int N> __device__ __forceinline__ void ap_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else static_assert(N == 0, "add the count to ap_wait_vm");
}

// VAR: bit 0 = lazy rescale (the shipped arithmetic; RLCF_ATTN_VAR=0 keeps the eager form).  Measurement builds of the 8-wave launch
// (rlcf_attention_debug / RLCF_ATTN_VAR, wrong numbers by design): 8 = no MFMAs, 32 = no DMA, 64 = no per-block arithmetic,
// 16 = s_memtime stamps of every wave (tools/attn_trace.py).
#define AP_MFMA(acc, a, b)                                                                                                \
    {                                                                                                                     \
        if constexpr (VAR & 8) asm volatile("" : "+v"(acc) : "v"(a), "v"(b));                                             \
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);                                            \
    }

// per-lane state of one block of 32 queries
struct ApAcc {
    f32x16 o0, o1;          // O^T accumulators: d 0-31 / 32-63 in the rows, the lane's query in the column
    float m, lsum;          // reference of the exponent (running maximum of the raw scores), row sum of P * 2^6
};
__device__ __forceinline__ void ap_acc_init(ApAcc& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a.o0[r] = 0.f; a.o1[r] = 0.f; }
    a.m = -INFINITY; a.lsum = 0.f;
}

// Q fragments of one query row (B operand of S^T = K.Q^T): lane (q, h) owns d = ks*16 + h*8 + [0,8) for ks = 0..3; the 1/8 scale
// is applied to the scores (exact).  quad of (d block ks>>1, part): (ks>>1) * (NQ/2) + part; inside the quad the 16-B slot (ks&1)*2 + h
template <bool SINGLE>
__device__ __forceinline__ void ap_load_q(const _Float16* __restrict__ qp /* row + head offset + h*8 */, h16x8 (&qh)[4], h16x8 (&ql)[4]) {
    constexpr int NQ = SINGLE ? 2 : 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qh[ks] = *(const h16x8*)(qp + (ks >> 1) * (NQ / 2) * 32 + (ks & 1) * 16);
        if constexpr (!SINGLE) ql[ks] = *(const h16x8*)(qp + ((ks >> 1) * 2 + 1) * 32 + (ks & 1) * 16);
        else ql[ks] = qh[ks];
    }
}

// one block of 32 keys (rows kc .. kc+31 of the sequence's key list) against the wave's 32 queries.
//   ke / ko: the lane's K row in LDS for even / odd k-steps; va: LDS byte address (integer) of the lane's transposing V read;
//   QB: bytes of one quad of the stage
template <bool SINGLE, int VAR, int QB>
__device__ __forceinline__ void ap_sub(ApAcc& a, const h16x8 (&qh)[4], const h16x8 (&ql)[4], const char* ke, const char* ko, unsigned va,
                                       int kc, int nkeys, int h, unsigned long long* trc = nullptr) {
#define AP_SUBSTAMP(i) if constexpr (VAR & 16) { if (trc) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) trc[i] = tm_; } }
    AP_SUBSTAMP(24)
    constexpr int NQ = SINGLE ? 2 : 4;
    constexpr float SC = 0.125f * 1.44269504088896341f;           // softmax(s/8) through v_exp_f32 (2^x)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const char* kp = ((ks & 1) ? ko : ke) + (ks >> 1) * (NQ / 2) * QB;
        const h16x8 kh = *(const h16x8*)kp;
        AP_MFMA(s, kh, qh[ks])
        if constexpr (!SINGLE) {
            const h16x8 kl = *(const h16x8*)(kp + QB);
            AP_MFMA(s, kh, ql[ks])
            AP_MFMA(s, kl, qh[ks])
        }
    }
    if (kc + 32 > nkeys) {                              // block holds rows past the last key
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (kc + mfma32_row(r, h) >= nkeys) s[r] = -INFINITY;
    }
    float cm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) cm = fmaxf(cm, s[r]);
    cm = fmaxf(cm, __shfl_xor(cm, 32));
    AP_SUBSTAMP(25)
    float mn, mb;
    if constexpr (VAR & 1) {
        // lazy rescale: m is the REFERENCE of the exponent, not the running maximum — it follows the maximum only when a row
        // outgrows it by more than 2^6 (P is carried times 2^6: a row's largest P then stays below 2^12, far inside f16)
        constexpr float THR = 6.0f;
        if (__any((cm - a.m) * SC > THR)) {              // (first block: m = -inf)
            mn = fmaxf(a.m, cm);
            const float alpha = (a.m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((a.m - mn) * SC);
            a.lsum *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a.o0[r] *= alpha; a.o1[r] *= alpha; }
        } else mn = a.m;
        mb = mn * SC - 6.0f;
    } else {
        mn = fmaxf(a.m, cm);
        const float alpha = (a.m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((a.m - mn) * SC);
        mb = mn * SC - 6.0f;                            // P is carried times 2^6: folded into the exponent (lsum carries it too)
        a.lsum *= alpha;
        if (alpha != 1.f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { a.o0[r] *= alpha; a.o1[r] *= alpha; }
        }
    }
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] * SC - mb);
        ps += s[r];
    }
    a.lsum += ps;
    AP_SUBSTAMP(26)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        // V^T fragments: quad (db, part) at + (db * (NQ/2) + part) * QB, second key quartet (r = 1) at + 8 keys * 64 B
        const unsigned vt = va + tt * 16 * 64;
        // hi parts first; the lo parts are fetched into the same registers once the four MFMAs that read the hi parts have been issued
        // (their latency hides under those MFMAs; 8 instead of 16 fragment registers live)
        u32x2 a0, a1, b0, b1;
        AP_TR(a0, vt, 0); AP_TR(a1, vt, 512);
        AP_TR(b0, vt, (NQ / 2) * QB); AP_TR(b1, vt, (NQ / 2) * QB + 512);
        float pv[8];
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) pv[e2] = s[8 * tt + e2];
        h16x8 ph, pl;
        ap_split8(pv, ph, pl);
        AP_SUBSTAMP(27 + 2 * tt)
        AP_TR_WAIT4(a0, a1, b0, b1);
        AP_SUBSTAMP(28 + 2 * tt)
        const h16x8 v0h = ap_cat(a0, a1), v1h = ap_cat(b0, b1);
        AP_MFMA(a.o0, v0h, ph)
        AP_MFMA(a.o1, v1h, ph)
        if constexpr (!SINGLE) {
            AP_MFMA(a.o0, v0h, pl)
            AP_MFMA(a.o1, v1h, pl)
            u32x2 c0, c1, d0, d1;
            AP_TR(c0, vt, QB); AP_TR(c1, vt, QB + 512);
            AP_TR(d0, vt, 3 * QB); AP_TR(d1, vt, 3 * QB + 512);
            AP_TR_WAIT4(c0, c1, d0, d1);
            const h16x8 v0l = ap_cat(c0, c1), v1l = ap_cat(d0, d1);
            AP_MFMA(a.o0, v0l, ph)
            AP_MFMA(a.o1, v1l, ph)
        }
    }
    a.m = mn;
    if constexpr (VAR & 16) asm volatile("" :: "v"(a.o0), "v"(a.o1));        // (stamp after the PV MFMAs have been ISSUED, not completed)
    AP_SUBSTAMP(31)
#undef AP_SUBSTAMP
}

// Four normalised output values -> their 8-byte piece of hi halves and (pair rows) the piece of lo halves.  The roundings are PINNED: the
// product o * inv is made opaque before anything consumes it and both conversions are v_cvt_pk_f16_f32.  Left to itself the compiler
// contracts o * inv - hi into one FMA or rounds a product straight to f16 (v_fma_mixlo_f16) in one of the two store forms below and not in
// the other — an ulp of lo (or, rarely, of a plain-f16 value) between two builds of the same arithmetic.  vout: the f32 values.
typedef unsigned ap_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ap_h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ap_pk_f16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool LO>
__device__ __forceinline__ void ap_piece(float o_0, float o_1, float o_2, float o_3, float inv, h16x4& hi, h16x4& lo, float* vout) {
    float v[4] = {o_0 * inv, o_1 * inv, o_2 * inv, o_3 * inv};
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    ap_u32x2 hp = {ap_pk_f16(v[0], v[1]), ap_pk_f16(v[2], v[3])};
    hi = __builtin_bit_cast(h16x4, hp);
    if constexpr (LO) {
        float d[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = v[q] - (float)hi[q];           // exact: hi is within half an f16 ulp of v
        ap_u32x2 lp = {ap_pk_f16(d[0], d[1]), ap_pk_f16(d[2], d[3])};
        lo = __builtin_bit_cast(h16x4, lp);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) vout[q] = v[q];
}

// normalise and store one query row (lane (q, h): d = 8g + 4h + [0,4) of both 32-column blocks for g = 0..3); a.lsum already summed
// over the two half-waves
__device__ __forceinline__ void ap_store(const ApAcc& a, size_t row, int head, int width, int h, float* __restrict__ out, _Float16* __restrict__ oh,
                                         int il, float* __restrict__ lse) {
    const float ltot = a.lsum;
    const float inv = 1.0f / ltot;                               // numerator and denominator both carry the 2^6 of P
    // log-sum-exp of the row's scores s/8 (saved for the flash-style backward): m is the reference of the exponent
    if (lse && h == 0) lse[row * (width / HEAD_DIM) + head] = a.m * 0.125f + logf(ltot * 0.015625f);
    const size_t obase = row * width + head * HEAD_DIM;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d = 8 * g + 4 * h;
        float v0[4], v1[4];
        h16x4 h0, l0, h1, l1;
        ap_piece<true>(a.o0[4 * g], a.o0[4 * g + 1], a.o0[4 * g + 2], a.o0[4 * g + 3], inv, h0, l0, v0);
        ap_piece<true>(a.o1[4 * g], a.o1[4 * g + 1], a.o1[4 * g + 2], a.o1[4 * g + 3], inv, h1, l1, v1);
        if (out) {
            *(float4*)(out + obase + d) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            *(float4*)(out + obase + 32 + d) = make_float4(v1[0], v1[1], v1[2], v1[3]);
        }
        if (oh) {
            // interleaved pair layout (il): the head's two 32-column blocks sit at row*2W + head*128 (+64), lo 32 halves after hi
            const size_t p0 = il ? row * 2 * width + head * 128 + d : obase + d;
            const size_t p1 = il ? p0 + 64 : p0 + 32;
            *(h16x4*)(oh + p0) = h0; *(h16x4*)(oh + p1) = h1;
            if (il) { *(h16x4*)(oh + p0 + 32) = l0; *(h16x4*)(oh + p1 + 32) = l1; }
        }
    }
}

// Round 6: the same rows as WHOLE 128-byte lines.  ap_store writes the accumulator layout as it falls: one instruction = 32 rows x 16 bytes, every
// line of the output requested 8 times (pair rows: 16 instructions x 32 line requests for 64 lines).  A CU's stores are paced by line
// requests (profiles/r6_notes.md section 1b), so the block goes through LDS instead: 16 rows at a time into the wave's slab (RB bytes per
// row: 256 = the head's interleaved pair row [hi b0 | lo b0 | hi b1 | lo b1], 128 = plain f16 [b0 | b1]; 16-byte slots XOR-ed with the row
// so that neither side of the transpose piles up on a bank), read back as 16 bytes per lane, one instruction = 1 KB of whole lines.
// Same bits at the same addresses, half as many store instructions, an eighth of the line requests.
template <int RB>
__device__ __forceinline__ void ap_store_lines(const ApAcc& a, char* slab, int lane, int nvalid, _Float16* __restrict__ orow0, size_t row_halves, bool nt) {
    constexpr int SLOTS = RB / 16, RPI = 1024 / RB;            // 16-byte slots per row; rows per store instruction
    const int l32 = lane & 31, h = lane >> 5, rl = l32 & 15;
    const float inv = 1.0f / a.lsum;
    constexpr int NP = RB / 64;                                 // 8-byte pieces per lane and g: hi b0, lo b0, hi b1, lo b1 (pair rows) or b0, b1
    h16x4 pc[4][NP];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float vv[4];
        h16x4 lo_;
        if constexpr (RB == 256) {
            ap_piece<true>(a.o0[4 * g], a.o0[4 * g + 1], a.o0[4 * g + 2], a.o0[4 * g + 3], inv, pc[g][0], pc[g][1], vv);
            ap_piece<true>(a.o1[4 * g], a.o1[4 * g + 1], a.o1[4 * g + 2], a.o1[4 * g + 3], inv, pc[g][2], pc[g][3], vv);
        } else {
            ap_piece<false>(a.o0[4 * g], a.o0[4 * g + 1], a.o0[4 * g + 2], a.o0[4 * g + 3], inv, pc[g][0], lo_, vv);
            ap_piece<false>(a.o1[4 * g], a.o1[4 * g + 1], a.o1[4 * g + 2], a.o1[4 * g + 3], inv, pc[g][1], lo_, vv);
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if ((l32 >> 4) == p) {
            char* wr = slab + rl * RB + 8 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < NP; ++i) *(h16x4*)(wr + (((4 * i + g) ^ (rl & 7)) * 16)) = pc[g][i];
        }
        __builtin_amdgcn_wave_barrier();                          // (one wave: its LDS operations execute in order; this keeps the compiler from reordering them)
#pragma unroll
        for (int k = 0; k < 16 / RPI; ++k) {
            const int r = k * RPI + lane / SLOTS, sl = lane % SLOTS;
            // (read with the element type it was written with: under type-based alias analysis a load of `unsigned` may be moved above
            //  stores of `_Float16` — the first build of this function did exactly that and stored the slab's previous contents)
            const h16x8 v = *(const h16x8*)(slab + r * RB + ((sl ^ (r & 7)) * 16));
            if (16 * p + r < nvalid) {
                if (nt) __builtin_nontemporal_store(v, (h16x8*)(orow0 + (size_t)(16 * p + r) * row_halves + sl * 8));      // (RLCF_ATTN_LINEST=2: measurement)
                else *(h16x8*)(orow0 + (size_t)(16 * p + r) * row_halves + sl * 8) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// one-shot kernel: NW waves = NW blocks of 32 queries of one (sequence, head); SK keys per ring stage; every wave stages and computes
template <int NW, int SK, bool SINGLE, int VAR = 0>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : (NW == 4 ? 2 : 1)) void attention_fwd_pair_kernel(
    const _Float16* __restrict__ qkv2, const rlcf_seq* __restrict__ seqs, int width, float* __restrict__ out, _Float16* __restrict__ oh,
    int ilf, int qb0, float* __restrict__ lse) {
    constexpr int NQ = SINGLE ? 2 : 4;                // 64-B quads of a key's K (or V) row of one head: (d block) x (hi / lo)
    constexpr int QB = SK * 64;                       // bytes of one quad of a stage: [key][64 B]
    constexpr int REGION = NQ * QB;                   // K (or V) part of a stage
    constexpr int STAGE = 2 * REGION;
    constexpr int PPR = REGION / 1024;                // DMA pieces per region (a piece = 16 keys x 64 B of one quad)
    constexpr int PPQ = QB / 1024;                    // pieces per quad
    constexpr int PPW = 2 * PPR / NW;                 // pieces per wave and stage
    static_assert((2 * PPR) % NW == 0 && PPW >= 1, "pieces must divide over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][STAGE]
    const rlcf_seq sq = seqs[blockIdx.y];
    const int head = blockIdx.z;
    if ((qb0 + blockIdx.x * NW) * 32 >= sq.q_len) return;
    const int t = threadIdx.x, lane = t & 63, l32 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ld = SINGLE ? 3 * width : 6 * width;    // halves per row of the operand matrix
    const int hoff = SINGLE ? head * 64 : head * 128;
    const int koff = (SINGLE ? width : 2 * width) + hoff, voff = (SINGLE ? 2 * width : 4 * width) + hoff;
    const int qb = qb0 + blockIdx.x * NW + wave;
    const bool active = qb * 32 < sq.q_len;           // waves past the end only help loading
    const int qi = min(qb * 32 + l32, sq.q_len - 1);
    const int nkeys = sq.pre_len + sq.q_len;

    h16x8 qh[4], ql[4];
    ap_load_q<SINGLE>(qkv2 + (size_t)(sq.q_start + qi) * ld + hoff + h * 8, qh, ql);
    // piece P of a stage (wave-uniform): region P / PPR (0 = K, 1 = V), quad (P % PPR) / PPQ, keys ((P % PPQ) * 16 + [0,16));
    // lane j lands at piece + 16 j: key j>>2, 16-B slot j&3 of the quad (K: the slot it SOURCES is XOR-ed with (key>>2)&3)
#define AP_ISSUE(kc_, stage_)                                                                                             \
    {                                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < PPW; ++p) {                                                                \
            const int P = wave * PPW + p, region = P / PPR, pr = P - region * PPR, quad = pr / PPQ;                        \
            const int kl = (pr - quad * PPQ) * 16 + (lane >> 2);                                                          \
            const int ss = region ? (lane & 3) : ((lane & 3) ^ ((kl >> 2) & 3));                                          \
            const int key = min((kc_) + kl, nkeys - 1);                                                                   \
            const int row = key < sq.pre_len ? sq.pre_start + key : sq.q_start + key - sq.pre_len;                        \
            const _Float16* src = qkv2 + (size_t)row * ld + (region ? voff : koff) + quad * 32 + ss * 8;                  \
            if constexpr (!(VAR & 32)) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (stage_) * STAGE + pr * 1024 + region * REGION), 16, 0, 0); \
        }                                                                                                                 \
    }
    // VAR & 16: s_memtime stamps per wave into the buffer passed in `lse` (24 x u64 per wave; tools/attn_trace.py)
    unsigned long long* trc = nullptr;
    if constexpr (VAR & 16) trc = (unsigned long long*)lse + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * NW + wave) * 32;
#define AP_STAMP(i)                                                                                                       \
    if constexpr (VAR & 16) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0) trc[i] = tm_; }
    const int nchunk = (nkeys + SK - 1) / SK;
    unsigned long long t_start = 0;
    if constexpr (VAR & 16) t_start = __builtin_amdgcn_s_memtime();
    AP_ISSUE(0, 0)
    if (nchunk > 1) AP_ISSUE(SK, 1)

    ApAcc acc;
    ap_acc_init(acc);
    // lane constants of the operand reads: K slot (ks&1)*2 + h XOR-ed with (key>>2)&3 -> one register for even, one for odd k-steps
    const int kx = (l32 >> 2) & 3;
    const int kb_e = l32 * 64 + ((h ^ kx) * 16), kb_o = l32 * 64 + (((2 + h) ^ kx) * 16);
    // transposing V read: lane (group g = lane>>4: dh = g&1, h = g>>1; i = lane&15) supplies the address of 4 halves of key
    // 4h + (i>>2) (+ 8r + 16tt) at columns dh*16 + (i&3)*4 and receives column dh*16 + i of keys 4h + 8r + [0,4)
    const int i16 = lane & 15, dh = (lane >> 4) & 1;
    const int vb_l = (4 * h + (i16 >> 2)) * 64 + dh * 32 + (i16 & 3) * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);        // LDS byte address of the ring

    for (int c = 0; c < nchunk; ++c) {
        if (c == 0 && nchunk > 1) ap_wait_vm<PPW>();              // chunk 1 may stay in flight
        else ap_wait_vm<0>();
        if (c == 0) { if constexpr (VAR & 16) { if (lane == 0) trc[0] = t_start; } AP_STAMP(1) }
        __builtin_amdgcn_s_barrier();                              // chunk c has landed for every wave; nobody reads chunk c-1 any more
        __builtin_amdgcn_sched_barrier(0);
        if (c < 4) AP_STAMP(2 + 4 * c)
        if (c >= 1 && c + 1 < nchunk) AP_ISSUE((c + 1) * SK, (c + 1) & 1)      // into the stage chunk c-1 left
        if (c < 4) AP_STAMP(3 + 4 * c)
        const char* sk_ = smem + (c & 1) * STAGE;
        const unsigned vl_ = lds0 + (c & 1) * STAGE + REGION + vb_l;
        if (active) {
#pragma unroll
            for (int sub = 0; sub < SK / 32; ++sub) {
                const int kc = c * SK + sub * 32;
                if (kc >= nkeys) break;
                if constexpr (!(VAR & 64)) ap_sub<SINGLE, VAR, QB>(acc, qh, ql, sk_ + kb_e + sub * 32 * 64, sk_ + kb_o + sub * 32 * 64, vl_ + sub * 32 * 64, kc, nkeys, h,
                                                                   (c == 1 && sub == 1) ? trc : nullptr);
                if (c < 4 && sub < 2) AP_STAMP(4 + 4 * c + sub)
            }
        }
    }
    AP_STAMP(18)
#undef AP_ISSUE
    if (active) {
        acc.lsum += __shfl_xor(acc.lsum, 32);
        const int il = ilf & 1;
        // whole-line stores (ilf & 2) go through the stage that chunk nchunk-2 left: every wave is past the last chunk's barrier (nobody reads
        // it any more), nothing was issued into it after that chunk — each wave owns STAGE / NW bytes of it, no further barrier
        constexpr int RB = SINGLE ? 128 : 256;                      // bytes of a head's output row: plain f16 / interleaved pair
        if constexpr (STAGE / NW >= 16 * RB && !(VAR & 16)) {
            if ((ilf & 2) && !out && oh && il == (SINGLE ? 0 : 1)) {
                const int nvalid = min(32, sq.q_len - qb * 32);
                if (lse && h == 0 && l32 < nvalid)
                    lse[(size_t)(sq.q_start + qi) * (width / HEAD_DIM) + head] = acc.m * 0.125f + logf(acc.lsum * 0.015625f);
                char* slab = smem + (nchunk & 1) * STAGE + wave * (STAGE / NW);
                const size_t row0 = (size_t)sq.q_start + (size_t)qb * 32;
                if constexpr (!SINGLE) ap_store_lines<256>(acc, slab, lane, nvalid, oh + row0 * 2 * width + head * 128, (size_t)2 * width, (ilf & 4) != 0);
                else ap_store_lines<128>(acc, slab, lane, nvalid, oh + row0 * width + head * HEAD_DIM, (size_t)width, (ilf & 4) != 0);
                return;
            }
        }
        if (qb * 32 + l32 < sq.q_len)
            ap_store(acc, (size_t)(sq.q_start + qi), head, width, h, out, oh, il, (VAR & 16) ? nullptr : lse);
    }
    AP_STAMP(19)
    if constexpr (VAR & 16) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); AP_STAMP(20) }
#undef AP_STAMP
}

// ------------------------------------------------------------------------------------------------------------------------------
static int g_ap_var = -1;                                  // measurement switch: RLCF_ATTN_VAR, or rlcf_attention_debug
static int ap_var() {
    if (g_ap_var < 0) { const char* e = getenv("RLCF_ATTN_VAR"); g_ap_var = e ? atoi(e) : 1; }
    return g_ap_var;
}
void attention_pair_debug(int /*reserved*/, int var) { g_ap_var = var; }
template <int NW, int SK>
static int ap_launch(bool single, dim3 grid, hipStream_t st, const _Float16* qkv2, const rlcf_seq* seqs, int width, float* out, _Float16* oh,
                     int il, int qb0, float* lse) {
    const size_t lds_pair = (size_t)2 * 2 * 4 * SK * 64, lds_single = (size_t)2 * 2 * 2 * SK * 64;      // [2 stages][K | V][quads][SK][64 B]
    const int var = ap_var();
#define AP_GO(S, V, LDS)                                                                                                  \
    {                                                                                                                     \
        int rc = rlcf_func_lds((const void*)attention_fwd_pair_kernel<NW, SK, S, V>, LDS);                                \
        if (rc != RLCF_OK) return rc;                                                                                     \
        attention_fwd_pair_kernel<NW, SK, S, V><<<grid, dim3(64 * NW), LDS, st>>>(qkv2, seqs, width, out, oh, il, qb0, lse); \
        RLCF_LAUNCH_CHECK();                                                                                              \
        return RLCF_OK;                                                                                                   \
    }
    if (single) {
#ifdef RLCF_ATTN_ABLATION                                    // (measurement builds: the single-pass form's ablations, as below for the pair form)
        if constexpr (NW == 8 && SK == 64) {
            if (var == 8) AP_GO(true, 8, lds_single)
            if (var == 32) AP_GO(true, 32, lds_single)
            if (var == 40) AP_GO(true, 40, lds_single)
            if (var == 64) AP_GO(true, 64, lds_single)
        }
#endif
        if (var & 1) AP_GO(true, 1, lds_single)
        AP_GO(true, 0, lds_single)
    }
#ifdef RLCF_ATTN_ABLATION                                    // measurement builds only (make ABLATION=1): never in the shipped library
    if constexpr (NW == 8) {                                 // ablation / trace builds of the 8-wave launch (timing only: wrong numbers by design)
        if (var == 8) AP_GO(false, 8, lds_pair)
        if (var == 32) AP_GO(false, 32, lds_pair)
        if (var == 40) AP_GO(false, 40, lds_pair)
        if (var == 64) AP_GO(false, 64, lds_pair)
        if (var == 16) {                                      // s_memtime stamps of every wave of the launch, dumped to RLCF_ATTN_TRACE_FILE
            static unsigned long long* tb = nullptr;
            static size_t tb_n = 0;
            const size_t n = (size_t)grid.y * grid.z * NW * 32;
            if (n > tb_n) { if (tb) (void)hipFree(tb); RLCF_HIP_CHECK(hipMalloc((void**)&tb, n * 8)); tb_n = n; }
            RLCF_HIP_CHECK(hipMemsetAsync(tb, 0, n * 8, st));
            int rc = rlcf_func_lds((const void*)attention_fwd_pair_kernel<NW, SK, false, 16>, lds_pair);
            if (rc != RLCF_OK) return rc;
            attention_fwd_pair_kernel<NW, SK, false, 16><<<grid, dim3(64 * NW), lds_pair, st>>>(qkv2, seqs, width, out, oh, il, qb0, (float*)tb);
            RLCF_LAUNCH_CHECK();
            RLCF_HIP_CHECK(hipStreamSynchronize(st));
            if (const char* f = getenv("RLCF_ATTN_TRACE_FILE")) {
                std::vector<unsigned long long> hb(n);
                RLCF_HIP_CHECK(hipMemcpy(hb.data(), tb, n * 8, hipMemcpyDeviceToHost));
                if (FILE* fp = fopen(f, "wb")) { fwrite(hb.data(), 8, n, fp); fclose(fp); }
            }
            return RLCF_OK;
        }
    }
#else
    if (var & ~1) { rlcf_set_error("attention variant %d is an ablation build (make ABLATION=1); only 0 / 1 are shipped", var); return RLCF_ERR_ARG; }
#endif
    if (var & 1) AP_GO(false, 1, lds_pair)
    AP_GO(false, 0, lds_pair)
#undef AP_GO
}

// qkv2: the in_proj output as interleaved f16 pairs [T, 3W] (row = 12W bytes; single: plain f16 [T, 3W]).  out (f32 [T, W]) and / or
// out_pairs (interleaved pairs [T, W]; single: plain f16) receive the attention output; lse [T, H] optional.
int launch_attention_fwd_pair(const void* qkv2, const rlcf_seq* seqs, int n_seq, int max_q_len, int width, float* out, void* out_pairs,
                              hipStream_t st, float* lse, int single) {
    RLCF_ARG_CHECK(qkv2 && seqs && n_seq > 0 && max_q_len > 0 && width % HEAD_DIM == 0 && (out || out_pairs));
    RLCF_ARG_CHECK(n_seq <= 65535);
    const _Float16* q2 = (const _Float16*)qkv2;
    _Float16* oh = (_Float16*)out_pairs;
    const char* lse_ = getenv("RLCF_ATTN_LINEST");      // =0: the output rows as 16-byte pieces per lane, as the accumulators hold them (A/B; read per launch)
    const int linest = lse_ ? atoi(lse_) : 1;
    const int il = (single ? 0 : 1) | (linest ? 2 : 0) | (linest == 2 ? 4 : 0), H = width / HEAD_DIM;       // bit 0: interleaved pair rows; bit 1: whole-line stores
    if (max_q_len > 128) {          // ViT sequences (197 / 257 / 577 tokens): 8 query blocks share every K / V stage
        const int full = max_q_len / 256, tail = max_q_len - full * 256;
        const bool split_tail = full >= 1 && tail > 0 && tail <= 32;        // 257 tokens: the odd query goes to a one-wave launch
        dim3 grid(split_tail ? full : (max_q_len + 255) / 256, n_seq, H);
        // RLCF_ATTN_SK=128 (measurement, read per launch; single-pass f16 operands only — the pair form would need 128 KB of LDS per
        // workgroup): stages of 128 keys, ONE barrier per 128 keys instead of per 64.  Round 6, order-controlled in-process A/B at 1 280
        // sequences: 399.5 - 402.9 us against 384.3 - 385.8 for the 64-key stages — 3.7 % SLOWER (a first A/B that ran it right behind a
        // streaming-only ablation build had shown it 4 % faster: a clock carry-over, not the kernel).  profiles/r6_attention_breakdown.txt.
        const char* ske = getenv("RLCF_ATTN_SK");
        const bool sk128 = ske && atoi(ske) == 128;
        // RLCF_ATTN_NW=4 (measurement, single-pass form): workgroups of FOUR waves = 128 queries — a 197-token sequence is two workgroups
        // that each stage all of K / V (twice the K / V reads, from L2 at best), four independent workgroups per CU instead of two
        const char* nwe = getenv("RLCF_ATTN_NW");
        if (single && nwe && atoi(nwe) == 4) {
            dim3 g4((max_q_len + 127) / 128, n_seq, H);
            return sk128 ? ap_launch<4, 128>(single, g4, st, q2, seqs, width, out, oh, il, 0, lse) : ap_launch<4, 64>(single, g4, st, q2, seqs, width, out, oh, il, 0, lse);
        }
        int rc = (single && sk128) ? ap_launch<8, 128>(single, grid, st, q2, seqs, width, out, oh, il, 0, lse)
                                   : ap_launch<8, 64>(single, grid, st, q2, seqs, width, out, oh, il, 0, lse);
        if (rc != RLCF_OK) return rc;
        if (split_tail) return ap_launch<1, 32>(single, dim3(1, n_seq, H), st, q2, seqs, width, out, oh, il, full * 8, lse);
        return RLCF_OK;
    }
    if (max_q_len > 32) return ap_launch<4, 64>(single, dim3((max_q_len + 127) / 128, n_seq, H), st, q2, seqs, width, out, oh, il, 0, lse);
    return ap_launch<1, 32>(single, dim3(1, n_seq, H), st, q2, seqs, width, out, oh, il, 0, lse);
}
