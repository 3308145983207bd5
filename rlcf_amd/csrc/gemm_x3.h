// Shared by the GEMM translation units (gemm_f16x3.hip: split-f16 kernels; gemm_f16.hip: the single-pass f16 kernel):
// the argument block and the compile-time epilogues of the DMA-ring kernels.
#pragma once
#include "kernels.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct GemmX3Args {
    const _Float16 *Ahi, *Alo; int lda;
    const _Float16 *Whi, *Wlo; int ldw;
    const float* bias;
    const float* residual; int ldr;
    const float* aux; int ldaux;
    float* C; int ldc;                 // f32 output (may be null)
    _Float16 *Chi, *Clo; int ldch;     // split output (may be null)
    int M, N, K;
    float alpha; int epilogue;
    unsigned int* amax_out;            // optional: atomicMax of |C| (float bits), see common.h
    int c_il;                          // split output interleaved: column c of a row sits at (c/32)*64 + c%32 (hi) / +32 (lo, = Clo)
    const float* alpha_dev;            // optional device scalar multiplied into alpha (undoes a data-dependent operand pre-scale)
    int kstep;                         // halves between consecutive K tiles in A/W rows: 32 (separate hi/lo arrays) or 64 (interleaved)
    int ksplit;                        // 128x128 DMA-ring kernel only: blockIdx.y walks K tiles [y*per, (y+1)*per); raw partial tiles go
    float* ws;                         // to ws[ksplit][M][N] and gemm_x3_splitk_reduce_kernel applies alpha / bias / epilogue
    int no_fast_epi;                   // RLCF_X3_NOFASTEPI=1: 256x256 kernel keeps the generic per-row epilogue (A/B measurements)
    // implicit 3x3 convolution (stride 1, pad 1; 256x256 kernel only): A is the NHWC activation as operand pairs [n*H*W, 2*conv_C]
    // (lda = 2*conv_C), K = 9*conv_C in (ky, kx, c) order; row r's K tile of tap (ky, kx) is pixel r + (ky-1)*W + (kx-1)'s channel block,
    // or 128 B of zeros (zpage) outside the image.  conv_C = 0: plain GEMM
    int conv_C, conv_H, conv_W;
    const _Float16* zpage;
    int tile_group;                    // 256x256 interleaved kernel: M tiles per scheduling group (0 = 8)
    const float* out_scale_dev;        // optional device scalar: the split output carries C * out_scale_dev[0] (a power of two chosen from
                                       // an upper bound of |C| before the launch: the consumer undoes it through ITS alpha_dev)
    // stream-K tail of the 256x256 interleaved kernel (launch_gemm_f16x3): tiles [0, sk_first) of the linear order are whole-tile
    // workgroups of the plain launch; the K steps of the remaining tiles are shared by the 2 * sk_blocks workgroups of the SK launch
    int sk_first, sk_blocks;
    unsigned sk_epoch;                 // value a "partial tile published" flag carries in THIS launch (flags are never reset)
    unsigned* sk_flags;                // [X3_SK_MAX_BLOCKS] + [1] time-out marker
    float* sk_ws;                      // [sk_blocks][256 x 256] raw partial accumulators, register layout
    // Output scale of a pair-emitting convolution chosen INSIDE the launch (round 5; before: conv_bound_scale_kernel, a one-thread launch
    // in front of every such GEMM — 136 launches per test image at RN50x64): bnd_in / bnd_res = device scalars max|input| and
    // max|identity| (complete before this launch), bnd_gain / bnd_bmax = host constants of the folded convolution; every thread derives
    // the same power of two s with (gain max|in| + bmax + max|res|) s in [2^14, 2^15) and thread 0 of workgroup 0 publishes (s, 1 / s) in
    // bnd_out2 for the CONSUMER of the pairs (its alpha_dev).  bnd_in == nullptr: out_scale_dev as before.
    const float *bnd_in, *bnd_res; float bnd_gain, bnd_bmax; float* bnd_out2;
    // LayerNorm folded into the single-pass f16 products of an image tower (gemm_f16.hip, gemm_nt_f16_pp_kernel<EPI, MODE>):
    //   MODE 1 (in_proj, c_fc): A is the f16 RESIDUAL STREAM itself and W carries the LayerNorm's gamma; the epilogue finishes the
    //          normalisation per row: rstd_r (alpha acc - mu_r s_c) + b'_c with ln_mr = [M][2] (mu, rstd), ln_s = [N] row sums of the folded
    //          f16 weight (what the MFMAs summed for a row of ones), bias = W beta + b
    //   MODE 2 (out_proj, c_proj): Chi IS the residual stream: x[r, c] = f16(x[r, c] + alpha acc + b_c) in place, and the row's
    //          (sum, sum of squares) over this wave's 64 columns goes to ln_part[(n0 / 256) * 4 + wn][r] for the next LayerNorm's statistics
    const float* ln_mr; const float* ln_s; float* ln_part;
    // 256x256 split-f16 kernel, measurement (RLCF_X3_STAGGER=P): the workgroups of the FIRST tile round start c/P of a tile late (c = index
    // inside the XCD mod P), so that the CUs are out of step for the whole launch and their store bursts do not meet (profiles/r5_notes.md)
    int stagger;
    // optional per-output-column factor applied to alpha * acc before the bias (generic epilogues only): the BatchNorm scale of a ResNet
    // convolution whose weight is kept UNFOLDED on the fp16 grid so that its products run two MFMA passes (resnet.hip)
    const float* col_scale;
    // 256x256 split-f16 kernel, WLO0 == 2: the weight's hi halves alone as a plain f16 matrix [N, K] (ldwpk halves per row).  One 128-byte
    // row block then holds the hi halves of TWO K tiles, so W pieces are staged every other K tile only (6 instead of 8 LDS-DMA
    // instructions per wave and K tile) and no zero lo half travels
    const _Float16* Wpk; int ldwpk;
};
#define X3_SK_MAX_BLOCKS 1024
#define X3_SK_FLAG_BYTES 8192                                   // flags + time-out word, at the end of the workspace
#define X3_SK_SLAB_BYTES 262144
// SINGLE (template flag of the kernels): plain f16 operands, ONE MFMA per product (RLCF_PREC_F16 — the arithmetic of the reference's
// own fp16-autocast GPU path, tpt_cls_rl.py:52; NOT f32-grade).  A plain f16 row of K halves has exactly the memory layout of an
// interleaved pair row of K/2 logical columns (every 128-B block = 32 "hi" + 32 "lo" halves), so the kernels run unchanged with
// g.K = K/2 and the MFMA triple (hi*hi, hi*lo, lo*hi) replaced by the two products of the block's own halves: "hi" x "hi" (K columns
// 0..31 of the block) and "lo" x "lo" (columns 32..63).  Split output (Chi) is then a plain f16 matrix; Clo may be null.

__device__ __forceinline__ int x3_ocol(const GemmX3Args& g, int col) { return g.c_il ? (((col >> 5) << 6) | (col & 31)) : col; }
__device__ __forceinline__ float x3_alpha(const GemmX3Args& g) { return g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha; }
// the scale the split output carries (1 when none): a device scalar chosen before the launch, or the bound-derived power of two (see bnd_in).
// Evaluated where the epilogue uses it (NOT at kernel entry: a value live across the K loop costs the 256-register kernels a spill).
__device__ __forceinline__ int x3_bound_shift(const GemmX3Args& g) {
    const float B = 1.02f * (g.bnd_gain * g.bnd_in[0] + g.bnd_bmax + (g.bnd_res ? g.bnd_res[0] : 0.f));
    int sh = 0;
    if (B > 0.f && B < INFINITY) sh = 14 - (int)floorf(log2f(B));
    return sh < -40 ? -40 : (sh > 40 ? 40 : sh);
}
__device__ __forceinline__ float x3_out_scale(const GemmX3Args& g) {
    if (!g.bnd_in) return g.out_scale_dev ? g.out_scale_dev[0] : 1.0f;
    return ldexpf(1.0f, x3_bound_shift(g));
}
// end of a kernel: thread 0 of workgroup 0 publishes (s, 1 / s) for the consumer of the pairs
__device__ __forceinline__ void x3_publish_scale(const GemmX3Args& g) {
    if (g.bnd_in && g.bnd_out2 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        const int sh = x3_bound_shift(g);
        g.bnd_out2[0] = ldexpf(1.0f, sh);
        g.bnd_out2[1] = ldexpf(1.0f, -sh);
    }
}
// Epilogue of the DMA-ring kernels for the shapes of the forward towers, specialised at compile time (no per-row branches on the
// epilogue kind) and written so that NOTHING waits inside the row loop: on gfx9 stores count in vmcnt like loads, so a residual load
// issued after a store cannot be awaited without draining that store — the generic loops (load, wait, store, per row) pay one
// HBM round trip per row, ~12 us per 256x256 tile.  Here the 16 residual rows of a 64-row slab are fetched BEFORE the accumulators
// are parked and combined in registers (no store yet); then the 16 rows stream out back to back.  Each wave parks and re-reads only
// its own LDS slice, so one barrier (the ring is no longer read) is all the synchronisation there is.
// One call = the 64x64 slab of one wave: a0..a3 = accumulator tiles (i, j) = (0,0) (0,1) (1,0) (1,1); row0 / col0 = its origin.
// LEAN (the experimental 4-wave kernel, which has no register to spare): the forward towers' form only -- host alpha, no ReLU, no
// max|C|, no output scale
template <int EPI, bool RES, bool F32OUT, bool PAIR, bool LEAN = false, int NIT = 16>
__device__ __forceinline__ void x3_epilogue_slab(const GemmX3Args& g, const f32x16& a0, const f32x16& a1, const f32x16& a2, const f32x16& a3,
                                                 float* park, int row0, int col0, int lane, float& am) {
    constexpr int ELD = 68;
    const int l32 = lane & 31, h = lane >> 5;
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int col = col0 + c4;
    const bool colok = col < g.N;
    const int colc = colok ? col : 0;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) bv = *(const float4*)(g.bias + colc);
    const float al = LEAN ? g.alpha : x3_alpha(g);         // (alpha_dev: the device-side undo of a data-dependent operand scale)
    // alpha times the optional per-column factor (GemmX3Args::col_scale: the BatchNorm scale of an unfolded ResNet convolution; exactly alpha when none)
    float4 alc = make_float4(al, al, al, al);
    if (!LEAN && g.col_scale) { const float4 cs4 = *(const float4*)(g.col_scale + colc); alc = make_float4(al * cs4.x, al * cs4.y, al * cs4.z, al * cs4.w); }
    const bool relu = !LEAN && g.epilogue == RLCF_EPI_RELU;     // ResNet convolutions: ReLU after the identity add (wave-uniform, one select per value)
    const float os = (!LEAN && PAIR) ? x3_out_scale(g) : 1.0f;
    const bool want_amax = !LEAN && g.amax_out != nullptr;
    const bool nt = !LEAN && (g.no_fast_epi & 2);             // non-temporal output stores (RLCF_X3_NT=0: default cache policy, A/B)
    const int ocol = g.c_il ? (((colc >> 5) << 6) | (colc & 31)) : colc;
    const int rbase = row0 + rsub;
    float4 rr[16];
    if constexpr (RES) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) rr[it] = *(const float4*)(g.residual + (size_t)min(rbase + it * 4, g.M - 1) * g.ldr + colc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pr = mfma32_row(r, h) * ELD + l32;
        park[pr] = a0[r]; park[pr + 32] = a1[r];
        if constexpr (NIT == 16) { park[pr + 32 * ELD] = a2[r]; park[pr + 32 * ELD + 32] = a3[r]; }        // (NIT = 8: a 32-row half slab, a0 / a1 only)
    }
    // Round 6: LINE-COMPLETE pair stores (kinds 3 / 4: in_proj -> Q / K / V pairs, c_fc + QuickGELU -> pairs; interleaved layout, 64 whole columns).
    // A lane holds 4 columns of a row; its hi halves (8 B) and lo halves (8 B) used to go out as two instructions that each wrote HALF of
    // every 128-byte line they touched ([hi c0-31 | lo c0-31] per 32-column block): every line requested twice, 256 line requests per 64x64
    // slab.  The single-pass f16 kernel showed (profiles/r6_notes.md section 1b) that a CU's store burst is paced by LINE REQUESTS, not bytes.
    // Now the 16 lanes of a row swap with lane ^ 8 what the partner's line needs (the lanes of block 0 send their lo halves and receive block
    // 1's hi halves): instruction A writes the whole line of block 0, instruction B the whole line of block 1 — each line requested once.
    // Same values, same bits, same number of store instructions.  RLCF_X3_LINEST=0: off.
    const bool line_ok = PAIR && !RES && !F32OUT && !LEAN && g.c_il && g.Clo == g.Chi + 32 && !(g.no_fast_epi & 4) && col0 + 64 <= g.N && (g.ldch & 3) == 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const float4 a4 = *(const float4*)(park + (it * 4 + rsub) * ELD + c4);
        float v[4] = {alc.x * a4.x + bv.x, alc.y * a4.y + bv.y, alc.z * a4.z + bv.z, alc.w * a4.w + bv.w};
        if constexpr (EPI == RLCF_EPI_QUICKGELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = quick_gelu_fast(v[q]);
        }
        if constexpr (RES) {
            rr[it].x += v[0]; rr[it].y += v[1]; rr[it].z += v[2]; rr[it].w += v[3];
            if (relu) { rr[it].x = fmaxf(rr[it].x, 0.f); rr[it].y = fmaxf(rr[it].y, 0.f); rr[it].z = fmaxf(rr[it].z, 0.f); rr[it].w = fmaxf(rr[it].w, 0.f); }
            asm volatile("" : "+v"(rr[it].x), "+v"(rr[it].y), "+v"(rr[it].z), "+v"(rr[it].w));     // materialise here: the sums must not sink
        } else {                                                                                     // into the store loop (IR sinking re-fuses the phases)
            const int row = rbase + it * 4;
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            // (the lane exchange of the line-complete stores sits OUTSIDE every per-lane condition: a cross-lane operation under a branch the
            //  compiler cannot prove uniform stops it from unrolling this loop, and the un-unrolled loop indexes rr[] dynamically = scratch)
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
            h16x4 hh_, ll_;
            u32x2_ ld0 = {0u, 0u}, ld1 = {0u, 0u};
            if constexpr (PAIR && !F32OUT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float vs = v[q] * os; hh_[q] = (_Float16)vs; ll_[q] = (_Float16)(vs - (float)hh_[q]); }
                const u32x2_ hv = __builtin_bit_cast(u32x2_, hh_), lv = __builtin_bit_cast(u32x2_, ll_);
                const bool up = (lane & 8) != 0;                 // this lane's columns belong to block 1 of the slab (columns 32-63)
                const u32x2_ send = up ? hv : lv;
                // the swap with lane ^ 8 goes through the slab this wave has just read (the lane's own four floats of this pass are consumed:
                // it overwrites two of them and reads its partner's two; LDS operations of one wave execute in order) — plain LDS accesses,
                // NOT a cross-lane intrinsic: a convergent operation in this loop keeps the compiler from unrolling it, and the loop that
                // is not unrolled indexes its register arrays dynamically (= scratch)
                float2* slot = (float2*)(park + (it * 4 + rsub) * ELD + c4);
                *slot = __builtin_bit_cast(float2, send);
                const u32x2_ recv = __builtin_bit_cast(u32x2_, *(const float2*)(park + (it * 4 + rsub) * ELD + (c4 ^ 32)));
                ld0 = up ? recv : hv; ld1 = up ? lv : recv;      // line of block 0 = [hi | lo] of columns 0-31; line of block 1
            }
            if (colok && row < g.M) {
                if (want_amax) am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if constexpr (F32OUT) { const f32x4 o4_ = {v[0], v[1], v[2], v[3]}; if (nt) __builtin_nontemporal_store(o4_, (f32x4*)(g.C + (size_t)row * g.ldc + col)); else *(f32x4*)(g.C + (size_t)row * g.ldc + col) = o4_; }
                if constexpr (PAIR) {
                    h16x4 hh, ll;
                    if constexpr (!F32OUT) { hh = hh_; ll = ll_; }        // (split once, above)
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { const float vs = v[q] * os; hh[q] = (_Float16)vs; ll[q] = (_Float16)(vs - (float)hh[q]); }
                    }
                    if (line_ok) {
                        _Float16* op = g.Chi + (size_t)row * g.ldch + 2 * (size_t)col0 + (lane & 15) * 4;
                        if (nt) { __builtin_nontemporal_store(ld0, (u32x2_*)op); __builtin_nontemporal_store(ld1, (u32x2_*)(op + 64)); }
                        else { *(u32x2_*)op = ld0; *(u32x2_*)(op + 64) = ld1; }
                    } else
                    if (nt) { __builtin_nontemporal_store(hh, (h16x4*)(g.Chi + (size_t)row * g.ldch + ocol)); if (g.Clo) __builtin_nontemporal_store(ll, (h16x4*)(g.Clo + (size_t)row * g.ldch + ocol)); }
                    else {
                    *(h16x4*)(g.Chi + (size_t)row * g.ldch + ocol) = hh;
                    if (g.Clo) *(h16x4*)(g.Clo + (size_t)row * g.ldch + ocol) = ll;
                    }
                }
            }
        }
    }
    if constexpr (RES) {
        asm volatile("" ::: "memory");                     // no store moves above this point, no load below it
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = rbase + it * 4;
            if (colok && row < g.M) {
                if (want_amax) am = fmaxf(am, fmaxf(fmaxf(fabsf(rr[it].x), fabsf(rr[it].y)), fmaxf(fabsf(rr[it].z), fabsf(rr[it].w))));
                if constexpr (F32OUT) { const f32x4 o4_ = {rr[it].x, rr[it].y, rr[it].z, rr[it].w}; if (nt) __builtin_nontemporal_store(o4_, (f32x4*)(g.C + (size_t)row * g.ldc + col)); else *(f32x4*)(g.C + (size_t)row * g.ldc + col) = o4_; }
                if constexpr (PAIR) {
                    const float v[4] = {rr[it].x * os, rr[it].y * os, rr[it].z * os, rr[it].w * os};
                    h16x4 hh, ll;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hh[q] = (_Float16)v[q]; ll[q] = (_Float16)(v[q] - (float)hh[q]); }
                    *(h16x4*)(g.Chi + (size_t)row * g.ldch + ocol) = hh;
                    if (g.Clo) *(h16x4*)(g.Clo + (size_t)row * g.ldch + ocol) = ll;
                }
            }
        }
        asm volatile("" ::: "memory");
    }
}
// which specialisation (wave-uniform): 0 = none (generic epilogue), 1 = f32 out, 2 = f32 out + residual, 3 = QuickGELU -> operand pair,
// 4 = operand pair only (in_proj of the image towers: Q / K / V go to the attention kernel as f16 pairs, attention_pair.hip)
__device__ __forceinline__ int x3_epilogue_kind(const GemmX3Args& g) {
    if (g.aux || (g.no_fast_epi & 1) || g.ksplit > 1) return 0;
    const bool f32o = g.C != nullptr, pair = g.Chi != nullptr, res = g.residual != nullptr;
    // kinds 1 / 2 also carry the ResNet convolutions' epilogue: device-side alpha, ReLU after the identity add, max|C| for the next scale
    const bool lin = g.epilogue == RLCF_EPI_NONE || g.epilogue == RLCF_EPI_RELU;
    if (lin && f32o && !pair) return res ? 2 : 1;
    if (lin && !f32o && pair && !res) return 4;            // (in_proj -> Q / K / V pairs; ResNet conv1 / conv2 -> pairs of the next convolution)
    if (lin && f32o && pair && res) return 5;              // ResNet conv3: block output as f32 (the next identity) AND as pairs (the next conv1)
    if (g.amax_out || g.alpha_dev || g.out_scale_dev || g.bnd_in || g.col_scale) return 0;
    if (g.epilogue == RLCF_EPI_QUICKGELU && !f32o && pair && !res) return 3;
    if (g.epilogue == RLCF_EPI_NONE && !f32o && pair && !res) return 4;
    return 0;
}
#define X3_EPILOGUE_SLAB(kind, ...)                                                                                      \
    {                                                                                                                    \
        if ((kind) == 1) x3_epilogue_slab<RLCF_EPI_NONE, false, true, false>(__VA_ARGS__);       /* in_proj (QKV), conv1 */      \
        else if ((kind) == 2) x3_epilogue_slab<RLCF_EPI_NONE, true, true, false>(__VA_ARGS__);   /* out_proj / c_proj + residual */ \
        else if ((kind) == 3) x3_epilogue_slab<RLCF_EPI_QUICKGELU, false, false, true>(__VA_ARGS__);   /* c_fc + QuickGELU -> pair */ \
        else if ((kind) == 5) x3_epilogue_slab<RLCF_EPI_NONE, true, true, true>(__VA_ARGS__);    /* conv3 + identity -> f32 and pairs */ \
        else x3_epilogue_slab<RLCF_EPI_NONE, false, false, true>(__VA_ARGS__);                   /* in_proj -> Q / K / V pairs */    \
    }

#define X3_EPILOGUE_HALFSLAB(kind, ...)                                                                                  \
    {                                                                                                                    \
        if ((kind) == 1) x3_epilogue_slab<RLCF_EPI_NONE, false, true, false, false, 8>(__VA_ARGS__);                     \
        else if ((kind) == 2) x3_epilogue_slab<RLCF_EPI_NONE, true, true, false, false, 8>(__VA_ARGS__);                 \
        else if ((kind) == 3) x3_epilogue_slab<RLCF_EPI_QUICKGELU, false, false, true, false, 8>(__VA_ARGS__);           \
        else if ((kind) == 5) x3_epilogue_slab<RLCF_EPI_NONE, true, true, true, false, 8>(__VA_ARGS__);                  \
        else x3_epilogue_slab<RLCF_EPI_NONE, false, false, true, false, 8>(__VA_ARGS__);                                 \
    }

