// Split-f16 GEMM: f32-grade results on the f16 matrix cores of gfx950.
//
// Every f32 operand x is carried as two f16 numbers  x = hi + lo,  hi = f16(x),  lo = f16(x - hi)
// (22 significant bits while lo is a normal f16, i.e. |x| >= 2^-3; below that the ABSOLUTE error is <= 2^-25 — weights
// are therefore pre-multiplied by an exact power of two at split time, undone in alpha), and a product a*b is evaluated
// as  ahi*bhi + ahi*blo + alo*bhi  with THREE v_mfma_f32_32x32x16_f16 into ONE f32 accumulator (each f16 x f16 product is
// exact in f32); the dropped lo*lo term is 2^-22 relative.  Measured on the ViT-B/16 image tower the
// feature error vs f64 is 6.9e-8 (plain f32: 6.1e-8, plain f16: 5.3e-5, bf16: 4.7e-4), i.e. the
// 1e-3 logit tolerance of the RLCF parity contract holds with the margin of the f32 path while
// the contraction runs on the 2.5 PF f16 MFMA pipe instead of the 157 TF f32 one.
//
// C[M,N] = epi(alpha * A.W^T + bias) (+ residual);  A = (Ahi,Alo) [M,K], W = (Whi,Wlo) [N,K].
// Tiling: 128x128 block tile, BK = 32, 4 waves (2x2) each 64x64 = 2x2 MFMA tiles x {main, corr}
// accumulators; operands register-staged into double-buffered LDS with rows padded to 80 B so
// that every ds_read_b128 operand fetch is bank-conflict free (16-lane groups hit 16 distinct
// 16-B slots); one barrier per K tile.
#include "kernels.h"
#include <cstdlib>
#include <algorithm>

#include "gemm_x3.h"
#define X3_BM 128
#define X3_BN 128
#define X3_BK 32
#define X3_LD 40                        // halves per LDS row (32 + 8 pad = 80 B)
#define X3_TILE (X3_BM * X3_LD)         // halves per operand tile

template <bool SINGLE>
__global__ __launch_bounds__(256, 2) void gemm_nt_f16x3_kernel(GemmX3Args g) {
    float am = 0.f;                 // max|C| of this thread's outputs (amax_out)
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];   // [2][4][128][40]
    // XCD-aware tile mapping: consecutive block ids run on different XCDs (id % 8); give each XCD
    // a contiguous range of tiles so that neighbours sharing an A row-panel share one L2.
    const int tiles_n = (g.N + X3_BN - 1) / X3_BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * X3_BM, n0 = tile_n * X3_BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: each operand tile = 128 rows x 64 B = 512 16-B chunks, 2 per thread (rows r0 and r0+64)
    const int r0 = t >> 2, c8 = (t & 3) * 8;
    const size_t ra0 = (size_t)min(m0 + r0, g.M - 1) * g.lda + c8, ra1 = (size_t)min(m0 + r0 + 64, g.M - 1) * g.lda + c8;
    const size_t rw0 = (size_t)min(n0 + r0, g.N - 1) * g.ldw + c8, rw1 = (size_t)min(n0 + r0 + 64, g.N - 1) * g.ldw + c8;
    const _Float16 *pah0 = g.Ahi + ra0, *pah1 = g.Ahi + ra1, *pal0 = g.Alo + ra0, *pal1 = g.Alo + ra1;
    const _Float16 *pwh0 = g.Whi + rw0, *pwh1 = g.Whi + rw1, *pwl0 = g.Wlo + rw0, *pwl1 = g.Wlo + rw1;
    const int d0 = r0 * X3_LD + c8, d1 = (r0 + 64) * X3_LD + c8;
    uint4 sah0, sah1, sal0, sal1, swh0, swh1, swl0, swl1;
#define X3_GLOAD(k0)                                                                              \
    sah0 = *(const uint4*)(pah0 + (k0)); sah1 = *(const uint4*)(pah1 + (k0));                     \
    sal0 = *(const uint4*)(pal0 + (k0)); sal1 = *(const uint4*)(pal1 + (k0));                     \
    swh0 = *(const uint4*)(pwh0 + (k0)); swh1 = *(const uint4*)(pwh1 + (k0));                     \
    swl0 = *(const uint4*)(pwl0 + (k0)); swl1 = *(const uint4*)(pwl1 + (k0));
#define X3_LSTORE(buf)                                                                            \
    {                                                                                             \
        _Float16* b_ = lds + (buf) * 4 * X3_TILE;                                                 \
        *(uint4*)(b_ + 0 * X3_TILE + d0) = sah0; *(uint4*)(b_ + 0 * X3_TILE + d1) = sah1;         \
        *(uint4*)(b_ + 1 * X3_TILE + d0) = sal0; *(uint4*)(b_ + 1 * X3_TILE + d1) = sal1;         \
        *(uint4*)(b_ + 2 * X3_TILE + d0) = swh0; *(uint4*)(b_ + 2 * X3_TILE + d1) = swh1;         \
        *(uint4*)(b_ + 3 * X3_TILE + d0) = swl0; *(uint4*)(b_ + 3 * X3_TILE + d1) = swl1;         \
    }

    const int nk = g.K / X3_BK;
    X3_GLOAD(0)
    X3_LSTORE(0)
    __syncthreads();
    const int arow = (wm * 64 + l32) * X3_LD + h * 8;
    const int brow = (wn * 64 + l32) * X3_LD + h * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { X3_GLOAD((kt + 1) * g.kstep) }
        const _Float16* base = lds + cur * 4 * X3_TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *(const h16x8*)(base + 0 * X3_TILE + arow + i * 32 * X3_LD + ks * 16);
                al[i] = *(const h16x8*)(base + 1 * X3_TILE + arow + i * 32 * X3_LD + ks * 16);
                bh[i] = *(const h16x8*)(base + 2 * X3_TILE + brow + i * 32 * X3_LD + ks * 16);
                bl[i] = *(const h16x8*)(base + 3 * X3_TILE + brow + i * 32 * X3_LD + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    if constexpr (SINGLE) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bl[j], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (kt + 1 < nk) X3_LSTORE(cur ^ 1)
        __syncthreads();
    }

    if (g.N % 4 == 0 && g.ldc % 4 == 0 && g.ldr % 4 == 0 && g.ldaux % 4 == 0 && g.ldch % 4 == 0) {
        // epilogue through LDS (see v2): 16-byte row-wise accesses instead of 64 scalar ones per lane
        constexpr int ELD = 68;
        float* park = (float*)lds + wave * (64 * ELD);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    park[(i * 32 + mfma32_row(r, h)) * ELD + j * 32 + l32] = acc[i][j][r];
        __syncthreads();
        const int c4 = (lane & 15) * 4, rsub = lane >> 4;
        const int col = n0 + wn * 64 + c4;
        if (col < g.N) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.bias) bv = *(const float4*)(g.bias + col);
            const float4 cs4 = g.col_scale ? *(const float4*)(g.col_scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);      // (exactly 1 when none)
            const float osd_ = g.Chi ? x3_out_scale(g) : 1.0f;   // scale of the split output, read BEFORE the first store of the loop (exactly 1 when none: v * 1 == v)
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int rl = it * 4 + rsub, row = m0 + wm * 64 + rl;
                if (row >= g.M) continue;
                const float4 a4 = *(const float4*)(park + rl * ELD + c4);
                const float al = x3_alpha(g);
                float v[4] = {al * cs4.x * a4.x + bv.x, al * cs4.y * a4.y + bv.y, al * cs4.z * a4.z + bv.z, al * cs4.w * a4.w + bv.w};
                if (g.epilogue == RLCF_EPI_QUICKGELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = quick_gelu_fast(v[q]);
                } else if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) {
                    const float4 x4 = *(const float4*)(g.aux + (size_t)row * g.ldaux + col);
                    v[0] *= quick_gelu_grad_fast(x4.x); v[1] *= quick_gelu_grad_fast(x4.y); v[2] *= quick_gelu_grad_fast(x4.z); v[3] *= quick_gelu_grad_fast(x4.w);
                }
                if (g.residual) {
                    const float4 r4 = *(const float4*)(g.residual + (size_t)row * g.ldr + col);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                if (g.epilogue == RLCF_EPI_RELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (g.amax_out) am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if (g.C) *(float4*)(g.C + (size_t)row * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                if (g.Chi) {
                    h16x4 hh, ll;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float vs_ = v[q] * osd_; hh[q] = (_Float16)vs_; ll[q] = (_Float16)(vs_ - (float)hh[q]); }
                    *(h16x4*)(g.Chi + (size_t)row * g.ldch + x3_ocol(g, col)) = hh;
                    if (g.Clo) *(h16x4*)(g.Clo + (size_t)row * g.ldch + x3_ocol(g, col)) = ll;
                }
            }
        }
        amax_commit(g.amax_out, am); x3_publish_scale(g);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l32;
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + mfma32_row(r, h);
                if (row >= g.M) continue;
                float v = x3_alpha(g) * acc[i][j][r] + bv;
                if (g.epilogue == RLCF_EPI_QUICKGELU) v = quick_gelu_fast(v);
                else if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) v *= quick_gelu_grad_fast(g.aux[(size_t)row * g.ldaux + col]);
                if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
                if (g.epilogue == RLCF_EPI_RELU) v = fmaxf(v, 0.f);
                if (g.amax_out) am = fmaxf(am, fabsf(v));
                if (g.C) g.C[(size_t)row * g.ldc + col] = v;
                if (g.Chi) {
                    const _Float16 hi = (_Float16)v;
                    g.Chi[(size_t)row * g.ldch + x3_ocol(g, col)] = hi;
                    if (g.Clo) g.Clo[(size_t)row * g.ldch + x3_ocol(g, col)] = (_Float16)(v - (float)hi);
                }
            }
        }
    amax_commit(g.amax_out, am); x3_publish_scale(g);
}

// ------------------------------------------------------------------------------------------
// v2: 256x128 block tile, 8 waves (4x2) of 64x64, BK = 32, operands DMA'd straight into LDS
// (global_load_lds_dwordx4, no staging registers, no ds_write) through a 3-stage ring with a
// counted s_waitcnt vmcnt so that two K tiles stay in flight across the single barrier per tile.
// glds writes LDS lane-linearly (wave-uniform base + lane*16 B), so the bank swizzle is applied
// to the per-lane SOURCE address and undone on the ds_read side: 16-B chunk c of row r lives at
// chunk slot c ^ ((r>>2)&3); a 16-lane ds_read_b128 group then touches 16 distinct slots.
// The same kernel at a 128x128 tile with 4 waves (NWM = 2, "v2s") serves grids too small for the big tiles (one test image's
// token matrix: 60-240 blocks): same DMA ring instead of the register staging of the 128x128 kernel above, whose K tile costs
// ~1.3 us of exposed global-load latency when one block runs per CU.
#define V2_BM 256
#define V2_BN 128
#define V2_GROUP_M 4
#define V2_STAGE 49152                  // bytes (NWM = 4): Ahi 16K | Alo 16K | Whi 8K | Wlo 8K

// RT = 32-row accumulator tiles per wave: 2 (64 x 64 per wave) or 1 — <4, ., 1> is the 128x128 tile on EIGHT waves of 32 x 64: the grids
// the 128x128 tile serves put one workgroup on a CU, and with four waves that is one wave per SIMD, whose LDS reads and barriers nothing
// hides (~1.0 us per K tile against 0.67 us per 128x128-equivalent on the 8-wave 256x128 kernel, measured on [4095, 1536, 512])
template <int NWM, bool SINGLE, int RT = 2, bool WLO0 = false>         // wave rows: 4 -> 256x128 tile, 8 waves; 2 -> 128x128 tile, 4 waves
__global__ __launch_bounds__(128 * NWM, NWM == 4 ? 2 : 1) void gemm_nt_f16x3_v2_kernel(GemmX3Args g) {
    constexpr int BM = 32 * RT * NWM, A_BYTES = BM * 64, W_BYTES = V2_BN * 64;
    constexpr int NP = 2 * RT + 2 * (4 / NWM);             // DMA pieces per wave and stage
    constexpr int STAGE = 2 * A_BYTES + 2 * W_BYTES, ALO = A_BYTES, WHI = 2 * A_BYTES, WLO = WHI + W_BYTES;
    constexpr int WPW = 4 / NWM;            // W-tile DMA instructions per wave and array (512 chunks over 2*NWM waves)
    float am = 0.f;                 // max|C| of this thread's outputs (amax_out)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [3][V2_STAGE]
    const int tiles_n = (g.N + V2_BN - 1) / V2_BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped order inside an XCD's range: V2_GROUP_M tile-rows are walked column by column, so the ~32 blocks an XCD
    // runs concurrently form a 4 x 8 patch that shares 4 A panels and 8 W panels in its L2
    const int tiles_m = (g.M + BM - 1) / BM;
    const int per_group = V2_GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * V2_GROUP_M;
    const int gsize = min(tiles_m - first_m, V2_GROUP_M), in_g = bid - grp * per_group;
    const int m0 = (first_m + in_g % gsize) * BM, n0 = (in_g / gsize) * V2_BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;

    f32x16 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA sources: A tile = BM x 4 chunks (RT wave-instructions per wave per array), W tile = 512 (WPW each)
    const int qa0 = (wave * RT) * 64 + lane, qa1 = qa0 + 64, qw = (wave * WPW) * 64 + lane, qw1 = qw + 64;
    const int ra0 = qa0 >> 2, ra1 = qa1 >> 2, rw = qw >> 2, rw1 = qw1 >> 2;
    const size_t sa0 = (size_t)min(m0 + ra0, g.M - 1) * g.lda + (((qa0 & 3) ^ ((ra0 >> 2) & 3)) * 8);
    const size_t sa1 = (size_t)min(m0 + ra1, g.M - 1) * g.lda + (((qa1 & 3) ^ ((ra1 >> 2) & 3)) * 8);
    const size_t sw = (size_t)min(n0 + rw, g.N - 1) * g.ldw + (((qw & 3) ^ ((rw >> 2) & 3)) * 8);
    const size_t sw1 = (size_t)min(n0 + rw1, g.N - 1) * g.ldw + (((qw1 & 3) ^ ((rw1 >> 2) & 3)) * 8);       // (WPW == 2 only)
    const int da0 = (wave * RT) * 1024, da1 = da0 + 1024, dw = (wave * WPW) * 1024, dw1 = dw + 1024;     // wave-uniform LDS byte offsets
    // one DMA piece (1 KiB per wave-instruction); the six pieces of a stage are issued BETWEEN the MFMA groups of the
    // current tile so that their ~100-cycle issue cost hides under matrix work instead of delaying it
#define V2_PIECE(idx, kk, sb_)                                                                                           \
    {                                                                                                                    \
        if ((idx) == 0) __builtin_amdgcn_global_load_lds((gptr_t)(g.Ahi + sa0 + (kk)), (lptr_t)((sb_) + da0), 16, 0, 0);            \
        if ((idx) == 1 && RT == 2) __builtin_amdgcn_global_load_lds((gptr_t)(g.Ahi + sa1 + (kk)), (lptr_t)((sb_) + da1), 16, 0, 0); \
        if ((idx) == 2) __builtin_amdgcn_global_load_lds((gptr_t)(g.Alo + sa0 + (kk)), (lptr_t)((sb_) + ALO + da0), 16, 0, 0);      \
        if ((idx) == 3 && RT == 2) __builtin_amdgcn_global_load_lds((gptr_t)(g.Alo + sa1 + (kk)), (lptr_t)((sb_) + ALO + da1), 16, 0, 0); \
        if ((idx) == 4) __builtin_amdgcn_global_load_lds((gptr_t)(g.Whi + sw + (kk)), (lptr_t)((sb_) + WHI + dw), 16, 0, 0);        \
        if ((idx) == 5) __builtin_amdgcn_global_load_lds((gptr_t)(g.Wlo + sw + (kk)), (lptr_t)((sb_) + WLO + dw), 16, 0, 0);        \
        if ((idx) == 6 && WPW == 2) __builtin_amdgcn_global_load_lds((gptr_t)(g.Whi + sw1 + (kk)), (lptr_t)((sb_) + WHI + dw1), 16, 0, 0); \
        if ((idx) == 7 && WPW == 2) __builtin_amdgcn_global_load_lds((gptr_t)(g.Wlo + sw1 + (kk)), (lptr_t)((sb_) + WLO + dw1), 16, 0, 0); \
    }
#define V2_ISSUE(k0, stage)                                                                                              \
    {                                                                                                                    \
        char* sbi_ = smem + (stage) * STAGE;                                                                             \
        V2_PIECE(0, k0, sbi_) V2_PIECE(1, k0, sbi_) V2_PIECE(2, k0, sbi_) V2_PIECE(3, k0, sbi_) V2_PIECE(4, k0, sbi_) V2_PIECE(5, k0, sbi_) \
        V2_PIECE(6, k0, sbi_) V2_PIECE(7, k0, sbi_)                                                                      \
    }
#define V2_LDFRAG(ks, AH, AL, BH, BL)                                                                                    \
    {                                                                                                                    \
        const int co_ = (((ks) * 2 + h) ^ swz) * 16;                                                                     \
        _Pragma("unroll") for (int i = 0; i < RT; ++i) {                                                                 \
            AH[i] = *(const h16x8*)(sb + aoff + i * 2048 + co_);                                                         \
            AL[i] = *(const h16x8*)(sb + ALO + aoff + i * 2048 + co_);                                                   \
        }                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
            BH[i] = *(const h16x8*)(sb + WHI + boff + i * 2048 + co_);                                                   \
            BL[i] = *(const h16x8*)(sb + WLO + boff + i * 2048 + co_);                                                   \
        }                                                                                                                \
    }
// WLO0 (constexpr in scope; round 5): the weight's lo halves are all ZERO — the weight sits on the f16 grid, as every Linear / conv / projection
// weight of a released CLIP checkpoint does (the archives store them as fp16; TPT/clip/model.py:399-436 copies them into float32
// parameters) — so the product a_hi . w_lo adds exact zeros and its MFMA is dropped: two passes instead of three, the same bits
// (up to the sign of an exact zero).  The engine finds out per weight at finalize (engine.hip make_split).
#define V2_MMA3(i, j, AH, AL, BH, BL)                                                                                    \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[j], acc[i][j], 0, 0, 0);                                \
    if constexpr (SINGLE) {                                                                                              \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BL[j], acc[i][j], 0, 0, 0);                            \
    } else {                                                                                                             \
        if constexpr (!WLO0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[j], acc[i][j], 0, 0, 0);       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[j], acc[i][j], 0, 0, 0);                            \
    }
// the same for the tile pair (i, 0), (i, 1), pass by pass: no MFMA directly follows its predecessor on the same accumulator (each
// accumulator still takes its three products in the same order: bit-identical results; measured -0.9 % on the layer's four products)
#define V2_MMA3_PAIR(i, AH, AL, BH, BL)                                                                                  \
    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[0], acc[i][0], 0, 0, 0);                                \
    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[1], acc[i][1], 0, 0, 0);                                \
    if constexpr (SINGLE) {                                                                                              \
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BL[0], acc[i][0], 0, 0, 0);                            \
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BL[1], acc[i][1], 0, 0, 0);                            \
    } else {                                                                                                             \
        if constexpr (!WLO0) {                                                                                           \
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[0], acc[i][0], 0, 0, 0);                        \
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[1], acc[i][1], 0, 0, 0);                        \
        }                                                                                                                \
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[0], acc[i][0], 0, 0, 0);                            \
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[1], acc[i][1], 0, 0, 0);                            \
    }
#define V2_FENCE __builtin_amdgcn_sched_barrier(0);

    // split-K (a few tiles, long K loop): this block walks K tiles [kt0, kt0 + nk) and leaves its raw partial tile in g.ws
    int nk = g.K / X3_BK, kt0 = 0;
    if (g.ksplit > 1) {
        const int per = (nk + g.ksplit - 1) / g.ksplit;
        kt0 = blockIdx.y * per;
        nk = min(nk, kt0 + per) - kt0;               // (the launcher makes every slice non-empty)
    }
    const int kb = kt0 * g.kstep;
    V2_ISSUE(kb, 0)
    if (nk > 1) V2_ISSUE(kb + g.kstep, 1)
    const int swz = (l32 >> 2) & 3;
    const int aoff = (wm * (32 * RT) + l32) * 64, boff = (wn * 64 + l32) * 64;      // row byte offsets
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");      // the NP pieces of the newest stage may stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool pf = kt + 2 < nk;
        const int kn = kb + (kt + 2) * g.kstep;
        char* sn = smem + (cur >= 1 ? cur - 1 : 2) * STAGE;                // stage (cur + 2) % 3
        const char* sb = smem + cur * STAGE;
        h16x8 ah0[RT], al0[RT], bh0[2], bl0[2], ah1[RT], al1[RT], bh1[2], bl1[2];
        V2_LDFRAG(0, ah0, al0, bh0, bl0)
        V2_FENCE
        V2_MMA3(0, 0, ah0, al0, bh0, bl0) V2_FENCE
        if (pf) V2_PIECE(0, kn, sn)
        V2_LDFRAG(1, ah1, al1, bh1, bl1)
        V2_FENCE
        V2_MMA3(0, 1, ah0, al0, bh0, bl0) V2_FENCE
        if constexpr (RT == 2) {
            if (pf) V2_PIECE(1, kn, sn)
            V2_FENCE
            V2_MMA3(1, 0, ah0, al0, bh0, bl0) V2_FENCE
            if (pf) V2_PIECE(2, kn, sn)
            V2_FENCE
            V2_MMA3(1, 1, ah0, al0, bh0, bl0) V2_FENCE
            if (pf) V2_PIECE(3, kn, sn)
            V2_FENCE
            V2_MMA3(0, 0, ah1, al1, bh1, bl1) V2_FENCE
            if (pf) { V2_PIECE(4, kn, sn) V2_PIECE(6, kn, sn) }
            V2_FENCE
            V2_MMA3(0, 1, ah1, al1, bh1, bl1) V2_FENCE
            if (pf) { V2_PIECE(5, kn, sn) V2_PIECE(7, kn, sn) }
            V2_FENCE
            V2_MMA3(1, 0, ah1, al1, bh1, bl1)
            V2_MMA3(1, 1, ah1, al1, bh1, bl1)
        } else {                            // one row tile per wave: four MFMA groups, the four pieces (A hi, A lo, W hi, W lo) between them
            if (pf) V2_PIECE(2, kn, sn)
            V2_FENCE
            V2_MMA3(0, 0, ah1, al1, bh1, bl1) V2_FENCE
            if (pf) { V2_PIECE(4, kn, sn) V2_PIECE(6, kn, sn) }
            V2_FENCE
            V2_MMA3(0, 1, ah1, al1, bh1, bl1)
            if (pf) { V2_PIECE(5, kn, sn) V2_PIECE(7, kn, sn) }
        }
        cur = cur == 2 ? 0 : cur + 1;
    }

    // epilogue through LDS: each wave parks its 64x64 f32 tile in its own 17 KB slice of the (now idle) stage ring and
    // re-reads it row-wise, so bias / residual / stores are 16-byte accesses covering whole 256-B row segments
    __syncthreads();
    if (const int ek = x3_epilogue_kind(g)) {
        if constexpr (RT == 2) {
            X3_EPILOGUE_SLAB(ek, g, acc[0][0], acc[0][1], acc[1][0], acc[1][1], (float*)smem + wave * (64 * 68), m0 + wm * 64, n0 + wn * 64, lane, am)
        } else {
            X3_EPILOGUE_HALFSLAB(ek, g, acc[0][0], acc[0][1], acc[0][0], acc[0][1], (float*)smem + wave * (32 * 68), m0 + wm * 32, n0 + wn * 64, lane, am)
        }
        amax_commit(g.amax_out, am); x3_publish_scale(g);
        return;
    }
    {
        constexpr int ELD = 68;                                            // floats per parked row (64 + 4 pad)
        float* park = (float*)smem + wave * (32 * RT * ELD);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    park[(i * 32 + mfma32_row(r, h)) * ELD + j * 32 + l32] = acc[i][j][r];
        __syncthreads();
        const int c4 = (lane & 15) * 4, rsub = lane >> 4;
        const int col = n0 + wn * 64 + c4;
        if (g.ksplit > 1) {                                                 // raw partial sums; the reduce kernel finishes the job
            float* wsz = g.ws + (size_t)blockIdx.y * g.M * g.N;
            if (col < g.N)
#pragma unroll 4
                for (int it = 0; it < 8 * RT; ++it) {
                    const int rl = it * 4 + rsub, row = m0 + wm * (32 * RT) + rl;
                    if (row < g.M) *(float4*)(wsz + (size_t)row * g.N + col) = *(const float4*)(park + rl * ELD + c4);
                }
            return;
        }
        if (col < g.N) {                                                    // N % 4 == 0 is checked by the launcher
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.bias) bv = *(const float4*)(g.bias + col);
            const float4 cs4 = g.col_scale ? *(const float4*)(g.col_scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);      // (exactly 1 when none)
            const float osd_ = g.Chi ? x3_out_scale(g) : 1.0f;   // scale of the split output, read BEFORE the first store of the loop (exactly 1 when none: v * 1 == v)
#pragma unroll 4
            for (int it = 0; it < 8 * RT; ++it) {
                const int rl = it * 4 + rsub, row = m0 + wm * (32 * RT) + rl;
                if (row >= g.M) continue;
                const float4 a4 = *(const float4*)(park + rl * ELD + c4);
                const float al = x3_alpha(g);
                float v[4] = {al * cs4.x * a4.x + bv.x, al * cs4.y * a4.y + bv.y, al * cs4.z * a4.z + bv.z, al * cs4.w * a4.w + bv.w};
                if (g.epilogue == RLCF_EPI_QUICKGELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = quick_gelu_fast(v[q]);
                } else if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) {
                    const float4 x4 = *(const float4*)(g.aux + (size_t)row * g.ldaux + col);
                    v[0] *= quick_gelu_grad_fast(x4.x); v[1] *= quick_gelu_grad_fast(x4.y); v[2] *= quick_gelu_grad_fast(x4.z); v[3] *= quick_gelu_grad_fast(x4.w);
                }
                if (g.residual) {
                    const float4 r4 = *(const float4*)(g.residual + (size_t)row * g.ldr + col);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                if (g.epilogue == RLCF_EPI_RELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (g.amax_out) am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if (g.C) *(float4*)(g.C + (size_t)row * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                if (g.Chi) {
                    h16x4 hh, ll;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float vs_ = v[q] * osd_; hh[q] = (_Float16)vs_; ll[q] = (_Float16)(vs_ - (float)hh[q]); }
                    *(h16x4*)(g.Chi + (size_t)row * g.ldch + x3_ocol(g, col)) = hh;
                    if (g.Clo) *(h16x4*)(g.Clo + (size_t)row * g.ldch + x3_ocol(g, col)) = ll;
                }
            }
        }
    }
    amax_commit(g.amax_out, am); x3_publish_scale(g);
}

// ------------------------------------------------------------------------------------------
// v3: 256x256 block tile, 8 waves (2x4) of 128x64 (8 MFMA tiles = 128 accumulator registers), BK = 32, DMA into a
// 2-stage LDS ring (64 KB per stage: 512 operand rows x 64 B x {hi, lo}); the two row groups of waves run the K loop half an
// iteration apart (ping-pong, see the loop).  Two thirds of v2's DMA bytes and three
// quarters of its LDS operand reads per MFMA — the resources the round-1 ablations identified as the limiter.
#define V3_BM 256
#define V3_BN 256
#define V3_STAGE 65536                  // bytes: Ahi 16K | Alo 16K | Whi 16K | Wlo 16K
#define V3_ALO 16384
#define V3_WHI 32768
#define V3_WLO 49152
__global__ __launch_bounds__(512, 2) void gemm_nt_f16x3_v3_kernel(GemmX3Args g) {
    constexpr bool SINGLE = false;      // (separate hi / lo arrays: the plain C-ABI call, split-f16 only)
    constexpr bool WLO0 = false;
    float am = 0.f;                 // max|C| of this thread's outputs (amax_out)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][V3_STAGE] (+ epilogue parking)
    const int tiles_n = (g.N + V3_BN - 1) / V3_BN, tiles_m = (g.M + V3_BM - 1) / V3_BM;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = 8 * tiles_n, grp = bid / per_group, first_m = grp * 8;
    const int gsize = min(tiles_m - first_m, 8), in_g = bid - grp * per_group;
    const int m0 = (first_m + in_g % gsize) * V3_BM, n0 = (in_g / gsize) * V3_BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3, l32 = lane & 31, h = lane >> 5;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA: every operand tile = 256 rows x 4 chunks = 1024 chunks = 2 pieces per wave; 8 pieces per wave per stage
    size_t sa[2], sw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (wave * 2 + j) * 64 + lane, r = q >> 2, c = ((q & 3) ^ ((r >> 2) & 3)) * 8;
        sa[j] = (size_t)min(m0 + r, g.M - 1) * g.lda + c;
        sw[j] = (size_t)min(n0 + r, g.N - 1) * g.ldw + c;
    }
#define V3_PIECE(idx, kk, sb_)                                                                                                         \
    {                                                                                                                                  \
        if ((idx) < 2) __builtin_amdgcn_global_load_lds((gptr_t)(g.Ahi + sa[(idx) & 1] + (kk)), (lptr_t)((sb_) + (wave * 2 + ((idx) & 1)) * 1024), 16, 0, 0);                   \
        else if ((idx) < 4) __builtin_amdgcn_global_load_lds((gptr_t)(g.Alo + sa[(idx) & 1] + (kk)), (lptr_t)((sb_) + V3_ALO + (wave * 2 + ((idx) & 1)) * 1024), 16, 0, 0);     \
        else if ((idx) < 6) __builtin_amdgcn_global_load_lds((gptr_t)(g.Whi + sw[(idx) & 1] + (kk)), (lptr_t)((sb_) + V3_WHI + (wave * 2 + ((idx) & 1)) * 1024), 16, 0, 0);     \
        else __builtin_amdgcn_global_load_lds((gptr_t)(g.Wlo + sw[(idx) & 1] + (kk)), (lptr_t)((sb_) + V3_WLO + (wave * 2 + ((idx) & 1)) * 1024), 16, 0, 0);                    \
    }
#define V3_LDA(ks, AH, AL)                                                                                               \
    {                                                                                                                    \
        const int co_ = (((ks) * 2 + h) ^ swz) * 16;                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                  \
            AH[i] = *(const h16x8*)(sb + aoff + i * 2048 + co_);                                                         \
            AL[i] = *(const h16x8*)(sb + V3_ALO + aoff + i * 2048 + co_);                                                \
        }                                                                                                                \
    }
#define V3_LDB(ks, BH, BL)                                                                                               \
    {                                                                                                                    \
        const int co_ = (((ks) * 2 + h) ^ swz) * 16;                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
            BH[j] = *(const h16x8*)(sb + V3_WHI + boff + j * 2048 + co_);                                                \
            BL[j] = *(const h16x8*)(sb + V3_WLO + boff + j * 2048 + co_);                                                \
        }                                                                                                                \
    }
    const int nk = g.K / X3_BK;
    {
        char* s0 = smem;
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) V3_PIECE(pi, 0, s0)
    }
    const int swz = (l32 >> 2) & 3;
    const int aoff = (wm * 128 + l32) * 64, boff = (wn * 64 + l32) * 64;
    // Ping-pong: the two waves of a SIMD belong to the row groups wm = 0 / 1, which run the same K loop half an iteration apart.
    // Two barriers per K tile (g = 2kt: tile kt has landed; g = 2kt+1); in every interval one group is in its pure-MFMA half
    // (second k-substep, fragments already in registers) while the other waits for its first fragments, so the matrix pipe always
    // has work.  Both groups issue their DMA pieces of tile kt+1 in the interval [2kt, 2kt+1].
    h16x8 ah0[4], al0[4], bh0[2], bl0[2], ah1[4], al1[4], bh1[2], bl1[2];
    if (wm == 0) {
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool pf = kt + 1 < nk;
            const int kn = (kt + 1) * g.kstep;
            char* sn = smem + ((kt + 1) & 1) * V3_STAGE;
            const char* sb = smem + (kt & 1) * V3_STAGE;
            V3_LDA(0, ah0, al0) V3_LDB(0, bh0, bl0)
            V2_FENCE
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                V2_MMA3_PAIR(i, ah0, al0, bh0, bl0) V2_FENCE
                if (pf) { V3_PIECE(i * 2, kn, sn) V3_PIECE(i * 2 + 1, kn, sn) }
                if (i == 0) { V3_LDA(1, ah1, al1) V3_LDB(1, bh1, bl1) }
                V2_FENCE
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { V2_MMA3_PAIR(i, ah1, al1, bh1, bl1) V2_FENCE }
        }
        __builtin_amdgcn_s_barrier();                              // pairs with the other group's last half step
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool pf = kt + 1 < nk;
            const int kn = (kt + 1) * g.kstep;
            char* sn = smem + ((kt + 1) & 1) * V3_STAGE;
            const char* sb = smem + (kt & 1) * V3_STAGE;
            if (kt > 0) {                                          // second k-substep of tile kt-1 + the DMA of tile kt+1
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    V2_MMA3_PAIR(i, ah1, al1, bh1, bl1) V2_FENCE
                    if (pf) { V3_PIECE(i * 2, kn, sn) V3_PIECE(i * 2 + 1, kn, sn) }
                    V2_FENCE
                }
            } else if (pf) {
#pragma unroll
                for (int pi = 0; pi < 8; ++pi) V3_PIECE(pi, kn, sn)
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            V3_LDA(0, ah0, al0) V3_LDB(0, bh0, bl0)
            V2_FENCE
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                V2_MMA3_PAIR(i, ah0, al0, bh0, bl0) V2_FENCE
                if (i == 0) { V3_LDA(1, ah1, al1) V3_LDB(1, bh1, bl1) }
                V2_FENCE
            }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { V2_MMA3_PAIR(i, ah1, al1, bh1, bl1) V2_FENCE }
    }
    // epilogue through LDS, two passes of 64 rows per wave (8 waves x 64 x 68 floats = 139 KB)
    constexpr int ELD = 68;
    float* park = (float*)smem + wave * (64 * ELD);
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int col = n0 + wn * 64 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && col < g.N) bv = *(const float4*)(g.bias + col);
    const float4 cs4 = (g.col_scale && col < g.N) ? *(const float4*)(g.col_scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);      // (exactly 1 when none)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) park[(i * 32 + mfma32_row(r, h)) * ELD + j * 32 + l32] = acc[half * 2 + i][j][r];
        __syncthreads();
        if (col < g.N) {
            const float osd_ = g.Chi ? x3_out_scale(g) : 1.0f;   // scale of the split output, read BEFORE the first store of the loop (exactly 1 when none: v * 1 == v)
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int rl = it * 4 + rsub, row = m0 + wm * 128 + half * 64 + rl;
                if (row >= g.M) continue;
                const float4 a4 = *(const float4*)(park + rl * ELD + c4);
                const float al = x3_alpha(g);
                float v[4] = {al * cs4.x * a4.x + bv.x, al * cs4.y * a4.y + bv.y, al * cs4.z * a4.z + bv.z, al * cs4.w * a4.w + bv.w};
                if (g.epilogue == RLCF_EPI_QUICKGELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = quick_gelu_fast(v[q]);
                } else if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) {
                    const float4 x4 = *(const float4*)(g.aux + (size_t)row * g.ldaux + col);
                    v[0] *= quick_gelu_grad_fast(x4.x); v[1] *= quick_gelu_grad_fast(x4.y); v[2] *= quick_gelu_grad_fast(x4.z); v[3] *= quick_gelu_grad_fast(x4.w);
                }
                if (g.residual) {
                    const float4 r4 = *(const float4*)(g.residual + (size_t)row * g.ldr + col);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                if (g.epilogue == RLCF_EPI_RELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (g.amax_out) am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if (g.C) *(float4*)(g.C + (size_t)row * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                if (g.Chi) {
                    h16x4 hh, ll;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float vs_ = v[q] * osd_; hh[q] = (_Float16)vs_; ll[q] = (_Float16)(vs_ - (float)hh[q]); }
                    *(h16x4*)(g.Chi + (size_t)row * g.ldch + x3_ocol(g, col)) = hh;
                    if (g.Clo) *(h16x4*)(g.Clo + (size_t)row * g.ldch + x3_ocol(g, col)) = ll;
                }
            }
        }
    }
    amax_commit(g.amax_out, am); x3_publish_scale(g);
}

// cache-policy bits of the operand DMA (one-off builds: -DV3_AUX_A=.. / -DV3_AUX_W=..; bit 0 sc0, bit 1 nt, bit 4 sc1)
#ifndef V3_AUX_A
#define V3_AUX_A 0
#endif
#ifndef V3_AUX_W
#define V3_AUX_W 0
#endif
// Stream-K tail (template flag SK; launch_gemm_f16x3).  With one 139-KB workgroup per CU a launch of T tiles costs ceil(T / 256) tile
// rounds, however few tiles the last round holds (out_proj at 20 images per pass: 2955 tiles = 11.54 rounds -> 12; one image: 150
// tiles on 256 CUs; the reward tower of a pass: 279 tiles).  The first sk_first = 256 * floor(T / 256) tiles stay whole-tile workgroups
// (the plain instantiation, launched first); the K steps of the remaining R tiles are shared out evenly by a second launch:
//   * the R tiles are cut into 8 chunks of whole tiles, chunk x served by the workgroups s with s % 8 == x (workgroups are dealt
//     round-robin to the XCDs, so a chunk's workgroups share an L2 and no dependency crosses chunks);
//   * inside a chunk the unit list (tile, k) is cut into q = sk_blocks / 8 contiguous ranges of <= nk units; a range touches at
//     most two tiles and is worked on from its END: its FIRST piece is its part of the last tile it touches (workgroup s of the
//     launch), its SECOND piece — if it straddles a tile boundary — the tail of the tile before (a workgroup of the second half of
//     the grid, handed out shortest-first-piece first so that the two pieces of a range add up to the same time on whichever CU takes
//     them: the CUs come free in the order of their first pieces);
//   * the piece that ENDS a tile (k1 == nk) owns the tile: every other part of that tile is the first piece of a LOWER-numbered range
//     of the chunk — a workgroup with a lower index (dispatched earlier, it never waits itself) — so the owner never waits for a
//     workgroup that is not yet resident, and normally finds the parts already published;
//   * a non-owner writes its raw accumulators to its slab with write-through (sc1) 16-byte stores in register layout (coalesced, no
//     LDS pass), every wave drains vmcnt, one lane publishes the launch's epoch in the range's flag word (relaxed, agent scope); the
//     owner polls that word relaxed, takes ONE agent-scope acquire, adds the slabs in increasing-k order with sc1 loads (fixed order:
//     bit-reproducible) and runs the epilogue (cdna_hip_programming.md, Guideline 16 / R1).
// MT = 32-row accumulator tiles per wave group: 4 = the 256-row tile; 3 = a 192-row tile for launches whose 256-row tiles fill
// little more than half of one round of workgroups (one image's token matrix against a W x W / W x 4W weight: 150 tiles on 256 CUs
// -> 198 tiles of 3/4 the work each)
template <bool SINGLE, bool CONV = false, bool SK = false, int MT = 4, int WLO0 = 0>       // WLO0: 0 three passes, 1 two passes (w_lo == 0), 2 two passes + packed hi-only W
__global__ __launch_bounds__(512, 2) void gemm_nt_f16x3_v3i_kernel(GemmX3Args g) {
    static_assert(MT == 4 || (MT == 3 && !SK && !CONV), "192-row tiles: plain products only");
    constexpr int BM = MT * 64;
    float am = 0.f;                 // max|C| of this thread's outputs (amax_out)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][V3_STAGE] (+ epilogue parking)
    const int tiles_n = (g.N + V3_BN - 1) / V3_BN, tiles_m = (g.M + BM - 1) / BM;
    int bid, k0 = 0, k1 = g.K / X3_BK;
    int sk_chunk = 0, sk_idx = 0, sk_q = 1, sk_units = 1, sk_tile_c = 0;
    if constexpr (!SK) {
        const int nwg = gridDim.x;
        bid = blockIdx.x;
        if (g.stagger > 1 && blockIdx.x < 256) {                        // (see GemmX3Args::stagger)
            const int c = (blockIdx.x >> 3) % g.stagger;
            const int cyc = c * (k1 * 5200 + 12000) / g.stagger;        // ~ cycles per tile: ~5 200 per K tile of 32 + epilogue
            for (int q2 = cyc >> 13; q2 > 0; --q2) __builtin_amdgcn_s_sleep(127);       // (127 x 64 cycles ~ 2^13)
        }
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    } else {
        const int nk_full = g.K / X3_BK;
        int s_ = blockIdx.x;
        const bool second = s_ >= g.sk_blocks;                         // the second piece of some range of this chunk
        if (second) s_ -= g.sk_blocks;
        const int R = tiles_m * tiles_n - g.sk_first;
        sk_chunk = s_ & 7; sk_idx = s_ >> 3; sk_q = g.sk_blocks >> 3;
        const int c0 = sk_chunk * R / 8, c1 = (sk_chunk + 1) * R / 8;       // (32-bit throughout: R < 256, <= 32 tiles x nk units per chunk)
        sk_units = (c1 - c0) * nk_full;
        if (sk_units <= 0) return;
        if (second) {
            // rank this chunk's two-piece ranges by (length of the first piece, index); this workgroup takes the one of rank sk_idx
            int* pick = (int*)smem;
            if (threadIdx.x == 0) *pick = -1;
            __syncthreads();
            if ((int)threadIdx.x < sk_q) {
                auto key = [&](int i) -> int {
                    const int a_ = i * sk_units / sk_q, b_ = (i + 1) * sk_units / sk_q;
                    if (b_ <= a_ || a_ / nk_full == (b_ - 1) / nk_full) return 0x7fffffff;               // empty, or one piece only
                    return (b_ - ((b_ - 1) / nk_full) * nk_full) * 64 + i;
                };
                const int mine = key(threadIdx.x);
                int rank = 0;
                for (int i = 0; i < sk_q; ++i) rank += key(i) < mine ? 1 : 0;
                if (mine != 0x7fffffff && rank == sk_idx) *pick = threadIdx.x;
            }
            __syncthreads();
            sk_idx = __builtin_amdgcn_readfirstlane(*pick);                // (wave-uniform: everything derived from it stays scalar)
            __syncthreads();
            if (sk_idx < 0) return;                                         // (fewer two-piece ranges than workgroups)
        }
        const int ua = sk_idx * sk_units / sk_q, ub = (sk_idx + 1) * sk_units / sk_q;
        if (ub <= ua) return;                                               // (more workgroups than units in this chunk)
        const int tb = (ub - 1) / nk_full, ta = ua / nk_full;
        if (second && ta == tb) return;
        if (!second) { sk_tile_c = tb; k0 = max(ua, tb * nk_full) - tb * nk_full; k1 = ub - tb * nk_full; }
        else { sk_tile_c = ta; k0 = ua - ta * nk_full; k1 = nk_full; }
        bid = g.sk_first + c0 + sk_tile_c;
    }
    // M tiles per scheduling group: 8 (16 / 32: -1.5 / -8 %, measured); RLCF_X3_GROUP pins it (measurements)
    const int G = g.tile_group % 100 > 0 ? g.tile_group % 100 : 8;
    const int per_group = G * tiles_n, grp = bid / per_group, first_m = grp * G;
    const int gsize = min(tiles_m - first_m, G), in_g = bid - grp * per_group;
    // order inside a group: M-fastest (neighbours share a W tile), or — problems 3-4 tiles wide, e.g. c_proj 768 x 3072 — N-fastest
    // (neighbours share the A panel: 414 -> 421 TF; on the 9 / 12-tile-wide products it costs 1-2 %).  tile_group >= 100 forces it
    const bool nfast = g.tile_group >= 100 || (g.tile_group == 0 && tiles_n <= 4);
    const int m0 = (nfast ? first_m + in_g / tiles_n : first_m + in_g % gsize) * BM;
    const int n0 = (nfast ? in_g % tiles_n : in_g / gsize) * V3_BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3, l32 = lane & 31, h = lane >> 5;

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA: every operand tile = 256 rows x 4 chunks = 1024 chunks = 2 pieces per wave; 8 pieces per wave per stage
    // Interleaved operands: row r of A (and W) holds, per K tile, one 128-byte block [32 hi | 32 lo]; a DMA instruction moves 8 rows
    // x 128 B (whole cache lines: the 64-B row segments of the separate-array layout cost the address/tag path twice the lines per
    // byte: 25.8 against 45.5 B/clk/CU in the delivery micro-benchmark).  LDS row = 128 B; 16-B chunk c (0-3 hi, 4-7 lo) of row r sits
    // in slot c ^ ((r>>1)&7), so a 16-lane ds_read_b128 group (16 consecutive rows, one chunk) covers all 64 banks.
    size_t sa[4], sw[4];
    size_t swp[4] = {0, 0, 0, 0};                               // (WLO0 == 2) the same rows of the packed hi-only copy
    unsigned vm[4] = {0x1ffu, 0x1ffu, 0x1ffu, 0x1ffu};      // CONV: bit t = tap t of this lane's row lies inside the image
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (wave * 4 + j) * 64 + lane, r = q >> 3, c = ((q & 7) ^ ((r >> 1) & 7)) * 8;
        // (MT = 3: the A tile is 192 rows = 3 pieces per wave, piece j of wave w = rows 8 (3 w + j) ..)
        const int qa = (wave * MT + j) * 64 + lane, ra = MT == 4 ? r : qa >> 3, ca = MT == 4 ? c : ((qa & 7) ^ ((ra >> 1) & 7)) * 8;
        const int row = min(m0 + ra, g.M - 1);
        sa[j] = (size_t)row * g.lda + ca + (SK ? (size_t)k0 * g.kstep : (size_t)0);        // (stream-K pieces start at K step k0)
        sw[j] = (size_t)min(n0 + r, g.N - 1) * g.ldw + c + (SK ? (size_t)k0 * g.kstep : (size_t)0);
        if constexpr (WLO0 == 2) swp[j] = (size_t)min(n0 + r, g.N - 1) * g.ldwpk + c;
        if constexpr (CONV) {
            const int ox = row % g.conv_W, oy = (row / g.conv_W) % g.conv_H;
            unsigned m = 0;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int iy = oy + tp / 3 - 1, ix = ox + tp % 3 - 1;
                if (iy >= 0 && iy < g.conv_H && ix >= 0 && ix < g.conv_W) m |= 1u << tp;
            }
            vm[j] = m;
        }
    }
    const _Float16* zlane = CONV ? g.zpage + lane * 8 : nullptr;       // 16 B of zeros per lane
    // CONV: element offset of K tile kt in the activation rows = ((ky-1) W + (kx-1)) lda + 64 (kt % (C/32)), and its tap
    const int cblocks = CONV ? g.conv_C / 32 : 1;
    auto conv_off = [&](int kt, int& tap) -> long {
        tap = kt / cblocks;
        const int cb = kt - tap * cblocks, ky = tap / 3, kx = tap - ky * 3;
        return ((long)(ky - 1) * g.conv_W + (kx - 1)) * (long)g.lda + (long)cb * 64;
    };
#undef V3_PIECE
#define V3_PIECE(idx, kk, sb_)                                                                                                         \
    {                                                                                                                                  \
        if ((idx) < 4) {                                                                                                               \
            if (MT < 4 && ((idx) & 3) >= MT) { /* a 192-row A tile has 3 pieces per wave */ }                                        \
            else if constexpr (CONV) {                                                                                                      \
                const _Float16* pa_ = ((vm[(idx) & 3] >> ctap_) & 1u) ? g.Ahi + (long)sa[(idx) & 3] + coff_ : zlane;                     \
                __builtin_amdgcn_global_load_lds((gptr_t)pa_, (lptr_t)((sb_) + (wave * MT + ((idx) & 3)) * 1024), 16, 0, V3_AUX_A);            \
            } else __builtin_amdgcn_global_load_lds((gptr_t)(g.Ahi + sa[(idx) & 3] + (kk)), (lptr_t)((sb_) + (wave * MT + ((idx) & 3)) * 1024), 16, 0, V3_AUX_A);           \
        } else if constexpr (WLO0 == 2) {                                                                                                \
            /* packed W: the pair of K tiles (kt+1, kt+2) when kt+1 is even, into the W buffer of that pair's parity */                     \
            if (wnext_) __builtin_amdgcn_global_load_lds((gptr_t)(g.Wpk + swp[(idx) & 3] + kkw_), (lptr_t)(smem + wb_ * V3_STAGE + 32768 + (wave * 4 + ((idx) & 3)) * 1024), 16, 0, V3_AUX_W); \
        } else __builtin_amdgcn_global_load_lds((gptr_t)(g.Whi + sw[(idx) & 3] + (kk)), (lptr_t)((sb_) + 32768 + (wave * 4 + ((idx) & 3)) * 1024), 16, 0, V3_AUX_W);              \
    }
#undef V3_LDA
#undef V3_LDB
#define V3_LDA(ks, AH, AL)                                                                                               \
    {                                                                                                                    \
        const int ch_ = (((ks) * 2 + h) ^ swz) * 16, cl_ = ((4 + (ks) * 2 + h) ^ swz) * 16;                              \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                 \
            AH[i] = *(const h16x8*)(sb + aoff + i * 4096 + ch_);                                                         \
            AL[i] = *(const h16x8*)(sb + aoff + i * 4096 + cl_);                                                         \
        }                                                                                                                \
    }
#define V3_LDB(ks, BH, BL)                                                                                               \
    {                                                                                                                    \
        if constexpr (WLO0 == 2) {                                                                                       \
            /* K tile kt = half (kt & 1) of its pair's 128-byte rows (chunks 0-3 / 4-7), in the W buffer of the pair's parity */ \
            const int ch_ = ((((kt) & 1) * 4 + (ks) * 2 + h) ^ swz) * 16;                                                  \
            const char* sw_ = smem + (((kt) >> 1) & 1) * V3_STAGE + 32768 + boff;                                          \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) { BH[j] = *(const h16x8*)(sw_ + j * 4096 + ch_); BL[j] = BH[j]; }  \
        } else {                                                                                                         \
        const int ch_ = (((ks) * 2 + h) ^ swz) * 16, cl_ = ((4 + (ks) * 2 + h) ^ swz) * 16;                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
            BH[j] = *(const h16x8*)(sb + 32768 + boff + j * 4096 + ch_);                                                 \
            BL[j] = *(const h16x8*)(sb + 32768 + boff + j * 4096 + cl_);                                                 \
        }                                                                                                                \
        }                                                                                                                \
    }
    const int nk = SK ? k1 - k0 : g.K / X3_BK;
    int ctap_ = 0;
    long coff_ = 0;
    if constexpr (CONV) coff_ = conv_off(0, ctap_);
    {
        char* s0 = smem;
        const bool wnext_ = true; const int kkw_ = 0, wb_ = 0;      // (WLO0 == 2: the first pair of K tiles)
        (void)wnext_; (void)kkw_; (void)wb_;
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) V3_PIECE(pi, 0, s0)
    }
    const int swz = (l32 >> 1) & 7;
    const int aoff = (wm * (MT * 32) + l32) * 128, boff = (wn * 64 + l32) * 128;
    // Ping-pong: the two waves of a SIMD belong to the row groups wm = 0 / 1, which run the same K loop half an iteration apart.
    // Two barriers per K tile (g = 2kt: tile kt has landed; g = 2kt+1); in every interval one group is in its pure-MFMA half
    // (second k-substep, fragments already in registers) while the other waits for its first fragments, so the matrix pipe always
    // has work.  Both groups issue their DMA pieces of tile kt+1 in the interval [2kt, 2kt+1].
    h16x8 ah0[MT], al0[MT], bh0[2], bl0[2], ah1[MT], al1[MT], bh1[2], bl1[2];
    if (wm == 0) {
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool pf = kt + 1 < nk;
            const int kn = (kt + 1) * g.kstep;
            const bool wnext_ = ((kt + 1) & 1) == 0; const int kkw_ = ((kt + 1) >> 1) * 64, wb_ = ((kt + 1) >> 1) & 1;      // (WLO0 == 2)
            (void)wnext_; (void)kkw_; (void)wb_;
            if constexpr (CONV) { if (pf) coff_ = conv_off(kt + 1, ctap_); }
            char* sn = smem + ((kt + 1) & 1) * V3_STAGE;
            const char* sb = smem + (kt & 1) * V3_STAGE;
            V3_LDA(0, ah0, al0) V3_LDB(0, bh0, bl0)
            V2_FENCE
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                V2_MMA3_PAIR(i, ah0, al0, bh0, bl0) V2_FENCE
                if (pf) { V3_PIECE(i * 2, kn, sn) V3_PIECE(i * 2 + 1, kn, sn) }
                if (MT == 3 && i == 2 && pf) { V3_PIECE(6, kn, sn) V3_PIECE(7, kn, sn) }
                if (i == 0) { V3_LDA(1, ah1, al1) V3_LDB(1, bh1, bl1) }
                V2_FENCE
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) { V2_MMA3_PAIR(i, ah1, al1, bh1, bl1) V2_FENCE }
        }
        __builtin_amdgcn_s_barrier();                              // pairs with the other group's last half step
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool pf = kt + 1 < nk;
            const int kn = (kt + 1) * g.kstep;
            const bool wnext_ = ((kt + 1) & 1) == 0; const int kkw_ = ((kt + 1) >> 1) * 64, wb_ = ((kt + 1) >> 1) & 1;      // (WLO0 == 2)
            (void)wnext_; (void)kkw_; (void)wb_;
            if constexpr (CONV) { if (pf) coff_ = conv_off(kt + 1, ctap_); }
            char* sn = smem + ((kt + 1) & 1) * V3_STAGE;
            const char* sb = smem + (kt & 1) * V3_STAGE;
            if (kt > 0) {                                          // second k-substep of tile kt-1 + the DMA of tile kt+1
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    V2_MMA3_PAIR(i, ah1, al1, bh1, bl1) V2_FENCE
                    if (pf) { V3_PIECE(i * 2, kn, sn) V3_PIECE(i * 2 + 1, kn, sn) }
                    if (MT == 3 && i == 2 && pf) { V3_PIECE(6, kn, sn) V3_PIECE(7, kn, sn) }
                    V2_FENCE
                }
            } else if (pf) {
#pragma unroll
                for (int pi = 0; pi < 8; ++pi) V3_PIECE(pi, kn, sn)
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            V3_LDA(0, ah0, al0) V3_LDB(0, bh0, bl0)
            V2_FENCE
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                V2_MMA3_PAIR(i, ah0, al0, bh0, bl0) V2_FENCE
                if (i == 0) { V3_LDA(1, ah1, al1) V3_LDB(1, bh1, bl1) }
                V2_FENCE
            }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) { V2_MMA3_PAIR(i, ah1, al1, bh1, bl1) V2_FENCE }
    }
    if constexpr (SK) {
        const int nk_full = g.K / X3_BK;
        if (k1 < nk_full) {
            // ---- a part that does not end its tile: publish the raw accumulators (register layout: 32 x 16 B per thread, coalesced)
            const int gid = sk_chunk + 8 * sk_idx;                      // this range's slab / flag
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((char*)g.sk_ws + (size_t)gid * X3_SK_SLAB_BYTES, 0, X3_SK_SLAB_BYTES, 0x00020000);
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const f32x16& a_ = acc[q >> 3][(q >> 2) & 1];
                const f32x4 v_ = {a_[(q & 3) * 4], a_[(q & 3) * 4 + 1], a_[(q & 3) * 4 + 2], a_[(q & 3) * 4 + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v_), rs, (q * 512 + t) * 16, 0, 16 /* sc1: write-through */);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // EVERY storing wave drains its stores ...
            __syncthreads();
            if (t == 0) __hip_atomic_store(g.sk_flags + gid, g.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then ONE lane publishes
            return;
        }
        // ---- the part that ends the tile: add the earlier parts (ranges first .. sk_idx - 1 of this chunk) in increasing-k order
        const int first = ((sk_tile_c * nk_full + 1) * sk_q - 1) / sk_units;                      // range holding the tile's K step 0
        for (int pi = first; pi < sk_idx; ++pi) {
            const int gp = sk_chunk + 8 * pi;
            if (t == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(g.sk_flags + gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.sk_epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 22)) { __hip_atomic_store(g.sk_flags + X3_SK_MAX_BLOCKS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((char*)g.sk_ws + (size_t)gp * X3_SK_SLAB_BYTES, 0, X3_SK_SLAB_BYTES, 0x00020000);
#pragma unroll
            for (int q8 = 0; q8 < 4; ++q8) {
                u32x4 v_[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v_[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((q8 * 8 + u) * 512 + t) * 16, 0, 16 /* sc1 */);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q8 * 8 + u;
                    const f32x4 f_ = __builtin_bit_cast(f32x4, v_[u]);
                    f32x16& a_ = acc[q >> 3][(q >> 2) & 1];
                    a_[(q & 3) * 4] += f_[0]; a_[(q & 3) * 4 + 1] += f_[1]; a_[(q & 3) * 4 + 2] += f_[2]; a_[(q & 3) * 4 + 3] += f_[3];
                }
            }
        }
    }
    if (const int ek = x3_epilogue_kind(g)) {
        float* parkf = (float*)smem + wave * (64 * 68);
        __syncthreads();
        // (written out, not a `#pragma unroll` loop over the halves: with every epilogue kind inlined the loop body exceeds the pragma-unroll
        //  size limit once the pair kinds carry the line-complete stores — the loop then stays a loop and acc[half * 2] becomes scratch)
        X3_EPILOGUE_SLAB(ek, g, acc[0][0], acc[0][1], acc[1][0], acc[1][1], parkf, m0 + wm * (MT * 32), n0 + wn * 64, lane, am)
        if constexpr (MT == 4)
            X3_EPILOGUE_SLAB(ek, g, acc[2][0], acc[2][1], acc[3][0], acc[3][1], parkf, m0 + wm * (MT * 32) + 64, n0 + wn * 64, lane, am)
        if constexpr (MT == 3)           // the third 32-row tile: half a slab
            X3_EPILOGUE_HALFSLAB(ek, g, acc[2][0], acc[2][1], acc[2][0], acc[2][1], parkf, m0 + wm * 96 + 64, n0 + wn * 64, lane, am)
        amax_commit(g.amax_out, am); x3_publish_scale(g);
        return;
    }
    // epilogue through LDS, two passes of 64 rows per wave (8 waves x 64 x 68 floats = 139 KB)
    constexpr int ELD = 68;
    float* park = (float*)smem + wave * (64 * ELD);
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int col = n0 + wn * 64 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && col < g.N) bv = *(const float4*)(g.bias + col);
    const float4 cs4 = (g.col_scale && col < g.N) ? *(const float4*)(g.col_scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);      // (exactly 1 when none)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (half * 2 + i < MT) park[(i * 32 + mfma32_row(r, h)) * ELD + j * 32 + l32] = acc[half * 2 + i < MT ? half * 2 + i : 0][j][r];
        __syncthreads();
        if (col < g.N) {
            const float osd_ = g.Chi ? x3_out_scale(g) : 1.0f;   // scale of the split output, read BEFORE the first store of the loop (exactly 1 when none: v * 1 == v)
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int rl = it * 4 + rsub, row = m0 + wm * (MT * 32) + half * 64 + rl;
                if (row >= g.M || half * 64 + rl >= MT * 32) continue;
                const float4 a4 = *(const float4*)(park + rl * ELD + c4);
                const float al = x3_alpha(g);
                float v[4] = {al * cs4.x * a4.x + bv.x, al * cs4.y * a4.y + bv.y, al * cs4.z * a4.z + bv.z, al * cs4.w * a4.w + bv.w};
                if (g.epilogue == RLCF_EPI_QUICKGELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = quick_gelu_fast(v[q]);
                } else if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) {
                    const float4 x4 = *(const float4*)(g.aux + (size_t)row * g.ldaux + col);
                    v[0] *= quick_gelu_grad_fast(x4.x); v[1] *= quick_gelu_grad_fast(x4.y); v[2] *= quick_gelu_grad_fast(x4.z); v[3] *= quick_gelu_grad_fast(x4.w);
                }
                if (g.residual) {
                    const float4 r4 = *(const float4*)(g.residual + (size_t)row * g.ldr + col);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                if (g.epilogue == RLCF_EPI_RELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (g.amax_out) am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if (g.C) *(float4*)(g.C + (size_t)row * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                if (g.Chi) {
                    h16x4 hh, ll;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float vs_ = v[q] * osd_; hh[q] = (_Float16)vs_; ll[q] = (_Float16)(vs_ - (float)hh[q]); }
                    *(h16x4*)(g.Chi + (size_t)row * g.ldch + x3_ocol(g, col)) = hh;
                    if (g.Clo) *(h16x4*)(g.Clo + (size_t)row * g.ldch + x3_ocol(g, col)) = ll;
                }
            }
        }
    }
    amax_commit(g.amax_out, am); x3_publish_scale(g);
}

// second pass of the split-K form: C = epi(alpha * sum_z ws[z] + bias) (+ residual), the epilogue of the kernels above, one thread
// per 4 columns of a row; the slices are added in a fixed order (deterministic)
__global__ __launch_bounds__(256) void gemm_x3_splitk_reduce_kernel(GemmX3Args g) {
    float am = 0.f;
    const int n4 = g.N >> 2;
    const long total = (long)g.M * n4;
    const float osd_ = g.Chi ? x3_out_scale(g) : 1.0f;   // scale of the split output, read BEFORE the first store of the loop (exactly 1 when none: v * 1 == v)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / n4), col = (int)(i % n4) * 4;
        // bias / aux / residual first, then the slices four at a time (independent loads in flight together; the adds stay in slice order)
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), x4 = bv, r4 = bv;
        if (g.bias) bv = *(const float4*)(g.bias + col);
        const float4 cs4 = g.col_scale ? *(const float4*)(g.col_scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
        if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) x4 = *(const float4*)(g.aux + (size_t)row * g.ldaux + col);
        if (g.residual) r4 = *(const float4*)(g.residual + (size_t)row * g.ldr + col);
        float4 a4 = *(const float4*)(g.ws + (size_t)row * g.N + col);
        for (int z = 1; z < g.ksplit; z += 4) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *(const float4*)(g.ws + ((size_t)min(z + u, g.ksplit - 1) * g.M + row) * g.N + col);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (z + u < g.ksplit) { a4.x += t[u].x; a4.y += t[u].y; a4.z += t[u].z; a4.w += t[u].w; }
        }
        const float al = x3_alpha(g);
        float v[4] = {al * cs4.x * a4.x + bv.x, al * cs4.y * a4.y + bv.y, al * cs4.z * a4.z + bv.z, al * cs4.w * a4.w + bv.w};
        if (g.epilogue == RLCF_EPI_QUICKGELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = quick_gelu_fast(v[q]);
        } else if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) {
            v[0] *= quick_gelu_grad_fast(x4.x); v[1] *= quick_gelu_grad_fast(x4.y); v[2] *= quick_gelu_grad_fast(x4.z); v[3] *= quick_gelu_grad_fast(x4.w);
        }
        if (g.residual) { v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
        if (g.epilogue == RLCF_EPI_RELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        if (g.amax_out) am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        if (g.C) *(float4*)(g.C + (size_t)row * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
        if (g.Chi) {
            h16x4 hh, ll;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float vs_ = v[q] * osd_; hh[q] = (_Float16)vs_; ll[q] = (_Float16)(vs_ - (float)hh[q]); }
            *(h16x4*)(g.Chi + (size_t)row * g.ldch + x3_ocol(g, col)) = hh;
            if (g.Clo) *(h16x4*)(g.Clo + (size_t)row * g.ldch + x3_ocol(g, col)) = ll;
        }
    }
    amax_commit(g.amax_out, am); x3_publish_scale(g);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// 256x256 tile on FOUR waves (one per SIMD, 492 registers each: 256 accumulators in AGPRs), wave tile 128x128 = 4x4 MFMA tiles.
// EXPERIMENTAL, off by default (RLCF_X3_V4=1; 2 / 3 = the no-DMA / no-MFMA ablations): per K tile a wave reads 32 fragments for 96
// MFMAs where the 8-wave kernel reads 24 for 48 (a third less LDS -> register traffic per flop) and there is ONE barrier per K tile.
// With a single wave per SIMD nothing else covers its latencies, so the wave pipelines itself: one memory instruction behind each
// MFMA, the fragments of a k-substep read during the MFMAs of the previous one, the barrier in the MIDDLE of the second substep
// (everyone has read the tile's slots by then) and the first fragments of tile kt+1 read behind the remaining MFMAs of tile kt.
// Measured (profiles/r3_gemm_experiments.txt): the main loop runs at the SAME 2.45 us per K tile as the 8-wave kernel, the four-slab
// epilogue of a wave costs the K = 768 shapes 4-9 %, and under sustained load both kernels sit on the 1400 W socket power limit
// (8-wave: 1506 MHz, 386 TF; this one: 1575 MHz, 381 TF; without its DMA 1775 MHz, 458 TF): what bounds the product is energy per
// flop, not a latency the schedule leaves exposed.  Interleaved operands (kstep 64), compile-time epilogues (x3_epilogue_kind != 0).
template <int ABL>          // ABL (measurements only): 1 = no DMA after the prologue, 2 = no MFMAs
__global__ __launch_bounds__(256, 1) void gemm_nt_f16x3_v4_kernel(GemmX3Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][V3_STAGE]
    const int tiles_n = (g.N + V3_BN - 1) / V3_BN, tiles_m = (g.M + V3_BM - 1) / V3_BM;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = 8 * tiles_n, grp = bid / per_group, first_m = grp * 8;
    const int gsize = min(tiles_m - first_m, 8), in_g = bid - grp * per_group;
    const int m0 = (first_m + in_g % gsize) * V3_BM, n0 = (in_g / gsize) * V3_BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA: a stage = 256 A rows + 256 W rows x 128 B = 64 pieces of 8 rows; wave w moves A pieces 8w..8w+7 and W pieces 8w..8w+7
    size_t sa[8], sw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = (wave * 8 + j) * 64 + lane, r = q >> 3, c = ((q & 7) ^ ((r >> 1) & 7)) * 8;
        sa[j] = (size_t)min(m0 + r, g.M - 1) * g.lda + c;
        sw[j] = (size_t)min(n0 + r, g.N - 1) * g.ldw + c;
    }
    // LDS: A ring of THREE 32-KB slots at [0, 96 KB), W ring of two at [96 KB, 160 KB).  The A tiles are unique to the workgroup and
    // come from HBM / the MALL, the W tiles are hot in every L2: A pieces go out two K tiles ahead, W pieces one -- a K tile's worth of
    // operands (64 KB) in flight per CU does not cover the loaded memory latency at this request rate (64 KB per ~1.8 us).
#define V4_WBASE 98304
#define V4_PA(j, kk, slot) __builtin_amdgcn_global_load_lds((gptr_t)(g.Ahi + sa[j] + (kk)), (lptr_t)(smem + (slot) * 32768 + (wave * 8 + (j)) * 1024), 16, 0, 0);
#define V4_PW(j, kk, slot) __builtin_amdgcn_global_load_lds((gptr_t)(g.Whi + sw[j] + (kk)), (lptr_t)(smem + V4_WBASE + (slot) * 32768 + (wave * 8 + (j)) * 1024), 16, 0, 0);
    const int swz = (l32 >> 1) & 7;
    const int aoff = (wm * 128 + l32) * 128, boff = V4_WBASE + (wn * 128 + l32) * 128;
#define V4_M(n, i, AH, AL, BH, BL)                                                                                       \
    if (ABL != 2) acc[i][(n) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16((n) < 8 ? AH[i] : AL[i], ((n) >> 2) == 1 ? BL[(n) & 3] : BH[(n) & 3], acc[i][(n) & 3], 0, 0, 0);
    // read q (0..3) of row tile i of a k-substep: A hi, A lo, W hi, W lo (sa_ / sw_: byte offsets of the A / W slots)
#define V4_RD(ks, i, q, AH, AL, BH, BL, sa_, sw_)                                                                       \
    {                                                                                                                    \
        const int ch_ = (((ks) * 2 + h) ^ swz) * 16, cl_ = ((4 + (ks) * 2 + h) ^ swz) * 16;                              \
        if ((q) == 0) AH[i] = *(const h16x8*)(smem + (sa_) + aoff + (i) * 4096 + ch_);                                   \
        else if ((q) == 1) AL[i] = *(const h16x8*)(smem + (sa_) + aoff + (i) * 4096 + cl_);                              \
        else if ((q) == 2) BH[i] = *(const h16x8*)(smem + (sw_) + boff + (i) * 4096 + ch_);                              \
        else BL[i] = *(const h16x8*)(smem + (sw_) + boff + (i) * 4096 + cl_);                                            \
    }
    const int nk = g.K / X3_BK;
    h16x8 ah0[4], al0[4], bh0[4], bl0[4], ah1[4], al1[4], bh1[4], bl1[4];
    // prologue: A(0), W(0), then A(1), W(1)
#pragma unroll
    for (int j = 0; j < 8; ++j) V4_PA(j, 0, 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) V4_PW(j, 0, 0)
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) V4_PA(j, g.kstep, 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) V4_PW(j, g.kstep, 1)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) V4_RD(0, i, q, ah0, al0, bh0, bl0, 0, 0)
    int as_cur = 0;                                       // A slot of tile kt (kt % 3)
    for (int kt = 0; kt < nk; ++kt) {
        const int as_nxt = as_cur == 2 ? 0 : as_cur + 1, as_nn = as_nxt == 2 ? 0 : as_nxt + 1;    // slots of tiles kt+1, kt+2
        const int ao = as_cur * 32768, an = as_nxt * 32768, wo = (kt & 1) * 32768, wn_ = ((kt + 1) & 1) * 32768;
        const bool pf1 = kt + 1 < nk, pf2 = kt + 2 < nk && ABL != 1;
        const int k2 = (kt + 2) * g.kstep;
        // substep 0: MFMAs on F0; behind them the fragments of substep 1 and the A pieces of tile kt+2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                V4_M(n, i, ah0, al0, bh0, bl0) V2_FENCE
                if (n < 4) V4_RD(1, i, n, ah1, al1, bh1, bl1, ao, wo)
                if (pf2 && i < 2 && n >= 4 && (n & 1) == 0) V4_PA(i * 4 + ((n - 4) >> 1), k2, as_nn)
                V2_FENCE
            }
        }
        // substep 1, first half
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int n = 0; n < 12; ++n) V4_M(n, i, ah1, al1, bh1, bl1)
            V2_FENCE
        }
        // every wave has read the slots of tile kt; A(kt+1) and W(kt+1) have landed (the 8 pieces of A(kt+2) may still be in flight)
        if (pf2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // second half: behind the MFMAs the first fragments of tile kt+1 and the W pieces of tile kt+2 (into the W slot of tile kt)
#pragma unroll
        for (int i = 2; i < 4; ++i) {
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                V4_M(n, i, ah1, al1, bh1, bl1) V2_FENCE
                if (pf1 && n < 8) V4_RD(0, (i - 2) * 2 + (n >> 2), n & 3, ah0, al0, bh0, bl0, an, wn_)
                if (pf2 && n >= 8) V4_PW((i - 2) * 4 + (n - 8), k2, kt & 1)
                V2_FENCE
            }
        }
        as_cur = as_nxt;
    }
    const int ek = x3_epilogue_kind(g);
    float* parkf = (float*)smem + wave * (64 * 68);
    float am = 0.f;
    __syncthreads();
    // (kinds 1-4 in their LEAN form: with the ResNet additions of the epilogue -- device alpha, ReLU, max|C|, output scale -- the 492
    // registers no longer suffice and the accumulators spill; the launcher sends those products to the 8-wave kernel)
#define V4_SLAB(...)                                                                                                     \
    {                                                                                                                    \
        if (ek == 1) x3_epilogue_slab<RLCF_EPI_NONE, false, true, false, true>(__VA_ARGS__);                             \
        else if (ek == 2) x3_epilogue_slab<RLCF_EPI_NONE, true, true, false, true>(__VA_ARGS__);                         \
        else if (ek == 3) x3_epilogue_slab<RLCF_EPI_QUICKGELU, false, false, true, true>(__VA_ARGS__);                   \
        else x3_epilogue_slab<RLCF_EPI_NONE, false, false, true, true>(__VA_ARGS__);                                     \
    }
#pragma unroll
    for (int hr = 0; hr < 2; ++hr)
#pragma unroll
        for (int hc = 0; hc < 2; ++hc)
            V4_SLAB(g, acc[hr * 2][hc * 2], acc[hr * 2][hc * 2 + 1], acc[hr * 2 + 1][hc * 2], acc[hr * 2 + 1][hc * 2 + 1], parkf,
                    m0 + wm * 128 + hr * 64, n0 + wn * 128 + hc * 64, lane, am)
    amax_commit(g.amax_out, am); x3_publish_scale(g);
}

extern int g_last_x3_variant;
// ------------------------------------------------------------------------------------------------------------------------------------
// Skinny split-f16 GEMM: M <= 256 rows of an f32 activation against a pre-split weight (the one-image path: the sparse text forward /
// backward works on ~239 rows, 96 such products per test image; they ran on the f32-MFMA split-K kernel at 12-23 TF).
//   * A stays f32 in memory and is split into (hi, lo) IN the kernel, by the wave that owns the rows — no stand-alone split launch,
//     no f16 copy of A; its power-of-two scale is 1 (forward activations) or comes from a device scalar max|A| the producing kernel
//     left behind (amax_in: the scale 2^(9 - floor(log2 max)) is formed here, its inverse folded into alpha);
//   * W = the interleaved pair rows the big kernels read ([N, 2K] halves, 32 hi | 32 lo per K block);
//   * grid (N / 32, 2 [, K slices]): a workgroup = 4 waves x one 32x32 output tile each (rows [128 y + 32 w, +32), columns
//     [32 x, +32)), K walked in steps of 16, three MFMAs per step; fragments go global -> registers (A rows are L2-resident, a W
//     fragment is 2 x 16 B per lane);
//   * K slices (K >= 1024): raw partial tiles to ws[slice][M][N], gemm_x3_splitk_reduce_kernel applies alpha / bias / epilogue /
//     residual in a fixed order; otherwise the epilogue runs here.
struct SkinnyArgs {
    const float* A; int lda;
    const float* amax_in;              // optional: device max|A| -> power-of-two operand scale
    int local_amax;                    // no max|A| known: every workgroup scales its 128 rows by their own max (found in the kernel)
    float* inv_scale_out;              // where 1 / scale goes when K slices hand the epilogue to the reduce kernel (alpha_dev of it)
    GemmX3Args g;                      // W pairs (Whi, ldw), bias, residual, aux, C, M, N, K, alpha, epilogue, amax_out, ksplit, ws
};
__global__ __launch_bounds__(256) void gemm_skinny_x3_kernel(SkinnyArgs s) {
    const GemmX3Args& g = s.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, mb = blockIdx.y * 128, m0 = mb + wave * 32;
    // LDS: one K chunk (64 columns) of the workgroup's 128 A rows as f16 hi | lo (row = 128 B hi + 128 B lo, + 16 B: a 16-lane ds_read_b128
    // group of consecutive rows covers all 64 banks) and of its 32 W rows as they lie in memory (2 blocks of [32 hi | 32 lo])
    constexpr int RS = 272;
    __shared__ __attribute__((aligned(16))) char lds[(128 + 32) * RS];
    __shared__ float red[4];
    char* la = lds;
    char* lw = lds + 128 * RS;
    float scale = 1.f, inv = 1.f;
    if (s.amax_in) {
        const float mx = s.amax_in[0];
        int sh = 0;
        if (mx > 0.f && mx < INFINITY) sh = 9 - (int)floorf(log2f(mx));
        sh = sh < -40 ? -40 : (sh > 40 ? 40 : sh);
        scale = ldexpf(1.0f, sh); inv = ldexpf(1.0f, -sh);
        if (s.inv_scale_out && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) s.inv_scale_out[0] = inv;
    }
    // cooperative, coalesced loads: thread (r16 = tid >> 4, p = tid & 15) takes 16 bytes p of the 256-byte K chunk of rows r16 + 16 j
    const int r16 = tid >> 4, p16 = tid & 15;
    const int nc = (g.K + 63) / 64;                       // K chunks of 64 columns (K % 32 == 0: the last one may be half)
    int c_lo = 0, c_hi = nc;
    if (g.ksplit > 1) { const int per = (nc + g.ksplit - 1) / g.ksplit; c_lo = blockIdx.z * per; c_hi = min(nc, c_lo + per); }
    const float* arow[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) arow[j] = s.A + (size_t)min(mb + r16 + 16 * j, g.M - 1) * s.lda;
    const _Float16* wrow[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) wrow[q] = g.Whi + (size_t)min(n0 + r16 + 16 * q, g.N - 1) * g.ldw;
    // every load is UNCONDITIONAL on a clamped address and masked afterwards: a conditional load compiles to a branch followed by
    // s_waitcnt vmcnt(0) — one exposed round trip per load, which is what the first forms of this kernel spent their 20-30 us on
    if (s.local_amax) {
        // operand range unknown and no producer left a max|A| behind (the gradients entering the sparse text backward): the workgroup
        // scales its 128 rows by their own max over the WHOLE K range (every K slice of these rows finds the same scale)
        float mx = 0.f;
        for (int c = 0; c < nc; c += 4) {                   // 32 independent loads in flight per thread (A is L2-resident: ~1 us per group)
            float4 v[4][8];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int ko = min(min(c + d, nc - 1) * 64 + p16 * 4, g.K - 4);       // (a repeated piece does not change the max)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[d][j] = *(const float4*)(arow[j] + ko);
            }
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[d][j].x), fabsf(v[d][j].y)), fmaxf(fabsf(v[d][j].z), fabsf(v[d][j].w))));
        }
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        int sh = 0;
        if (mx > 0.f && mx < INFINITY) sh = 9 - (int)floorf(log2f(mx));
        sh = sh < -40 ? -40 : (sh > 40 ? 40 : sh);
        scale = ldexpf(1.0f, sh); inv = ldexpf(1.0f, -sh);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // The direct form of this kernel (every lane fetching its own row's 16 bytes per K step) ran 25-30 us per launch whatever the
    // batching of its loads: 32 cache lines per load instruction, four instructions per K step and wave — the texture addresser, not
    // latency, was the limit.  Here a chunk costs 320 line requests per workgroup instead of 2048.
    // ... and the loop is a chain of memory latencies (a few dozen workgroups in the whole launch, W straight from HBM): chunks are
    // fetched SK_D ahead into registers, so K = 512 exposes two round trips instead of eight
    constexpr int SK_D = 4;
    float4 ar[SK_D][8];
    h16x8 wr[SK_D][2];
#define SK_LOAD(c_, d_)                                                                                                  \
    {                                                                                                                    \
        const int ka_ = min((c_) * 64 + p16 * 4, g.K - 4), kw_ = min((c_) * 128 + p16 * 8, 2 * g.K - 8);                 \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) ar[d_][j] = *(const float4*)(arow[j] + ka_);                      \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) wr[d_][q] = *(const h16x8*)(wrow[q] + kw_);                       \
    }
#pragma unroll
    for (int d = 0; d < SK_D; ++d)
        SK_LOAD(min(c_lo + d, nc - 1), d)
    const char* fa = la + (wave * 32 + l32) * RS + h * 16;
    const char* fw = lw + l32 * RS + h * 16;
    for (int c0 = c_lo; c0 < c_hi; c0 += SK_D) {
#pragma unroll
        for (int d = 0; d < SK_D; ++d) {
            const int c = c0 + d;
            if (c >= c_hi) break;
            __syncthreads();                                // the previous chunk's fragments have been read
            const bool kok = c * 64 + p16 * 4 < g.K;        // (K % 64 == 32: the upper half of the last chunk is zero)
            const bool wok = c * 64 + (p16 >> 3) * 32 < g.K;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool rok = kok && mb + r16 + 16 * j < g.M;
                const float v[4] = {ar[d][j].x * scale, ar[d][j].y * scale, ar[d][j].z * scale, ar[d][j].w * scale};
                h16x4 hh, ll;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = rok ? v[e] : 0.f;
                    hh[e] = (_Float16)x;
                    ll[e] = (_Float16)(x - (float)hh[e]);
                }
                char* dst = la + (r16 + 16 * j) * RS + p16 * 8;
                *(h16x4*)dst = hh;
                *(h16x4*)(dst + 128) = ll;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4 w_ = __builtin_bit_cast(u32x4, wr[d][q]);
                const unsigned m_ = wok ? 0xffffffffu : 0u;
                w_[0] &= m_; w_[1] &= m_; w_[2] &= m_; w_[3] &= m_;
                *(u32x4*)(lw + (r16 + 16 * q) * RS + p16 * 16) = w_;
            }
            __syncthreads();
            SK_LOAD(min(c + SK_D, nc - 1), d)                // refill this slot (clamped: a spare load at the end): in flight for the next SK_D - 1 chunks
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const h16x8 ah = *(const h16x8*)(fa + u * 32), al = *(const h16x8*)(fa + 128 + u * 32);
                const h16x8 wh = *(const h16x8*)(fw + (u >> 1) * 128 + (u & 1) * 32), wl = *(const h16x8*)(fw + (u >> 1) * 128 + 64 + (u & 1) * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc, 0, 0, 0);
            }
        }
    }
#undef SK_LOAD
    if (m0 >= g.M) return;
    const int col = n0 + l32;
    if (col >= g.N) return;
    if (g.ksplit > 1) {                                  // raw partial tile; alpha / bias / epilogue in the reduce pass
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = m0 + mfma32_row(r, h);
            if (rr < g.M) g.ws[((size_t)blockIdx.z * g.M + rr) * g.N + col] = s.local_amax ? acc[r] * inv : acc[r];     // (a power of two: exact)
        }
        return;
    }
    const float al_ = g.alpha * inv;
    const float bv = g.bias ? g.bias[col] : 0.f;
    float am = 0.f;
    // aux / residual of all 16 rows are fetched first, unconditionally (rows clamped): a load inside the row loop's `if` would be one
    // exposed round trip per row
    float auxv[16], resv[16];
    const bool has_aux = g.epilogue == RLCF_EPI_QUICKGELU_BWD, has_res = g.residual != nullptr;
    if (has_aux) {
#pragma unroll
        for (int r = 0; r < 16; ++r) auxv[r] = g.aux[(size_t)min(m0 + mfma32_row(r, h), g.M - 1) * g.ldaux + col];
    }
    if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) resv[r] = g.residual[(size_t)min(m0 + mfma32_row(r, h), g.M - 1) * g.ldr + col];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = m0 + mfma32_row(r, h);
        float v = al_ * acc[r] + bv;
        if (g.epilogue == RLCF_EPI_QUICKGELU) v = quick_gelu_fast(v);
        else if (has_aux) v *= quick_gelu_grad_fast(auxv[r]);
        if (has_res) v += resv[r];
        if (g.epilogue == RLCF_EPI_RELU) v = fmaxf(v, 0.f);
        if (rr < g.M) {
            am = fmaxf(am, fabsf(v));
            g.C[(size_t)rr * g.ldc + col] = v;
        }
    }
    amax_commit(g.amax_out, am); x3_publish_scale(g);
}
// A [M, K] f32 (lda), W pairs [N, 2K] interleaved; C f32 [M, N].  ws / ws_bytes: split-K scratch (K >= 1024); inv_scale_scratch: one
// device float (needed with amax_in AND K slices).  Returns RLCF_ERR_ARG for shapes it does not serve (the caller falls back).
bool gemm_skinny_x3_ok(int M, int N, int K, int lda, int ldc) { return M > 0 && M <= 256 && N % 4 == 0 && K % 32 == 0 && lda % 4 == 0 && ldc % 4 == 0; }
int launch_gemm_skinny_x3(const float* A, int lda, const void* Wpairs, const float* bias, const float* residual, int ldr, const float* aux,
                          int ldaux, float* C, int ldc, int M, int N, int K, float alpha, int epilogue, const float* amax_in,
                          unsigned int* amax_out, float* ws, size_t ws_bytes, float* inv_scale_scratch, hipStream_t st, int local_amax) {
    RLCF_ARG_CHECK(A && Wpairs && C && gemm_skinny_x3_ok(M, N, K, lda, ldc) && ((uintptr_t)A & 15) == 0);
    RLCF_ARG_CHECK(epilogue != RLCF_EPI_QUICKGELU_BWD || aux);
    SkinnyArgs s{};
    s.A = A; s.lda = lda; s.amax_in = amax_in; s.local_amax = (local_amax && !amax_in) ? 1 : 0;
    GemmX3Args& g = s.g;
    g.Whi = (const _Float16*)Wpairs; g.Wlo = g.Whi + 32; g.ldw = 2 * K; g.bias = bias; g.residual = residual; g.ldr = ldr; g.aux = aux; g.ldaux = ldaux;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.epilogue = epilogue; g.amax_out = amax_out; g.kstep = 64;
    const int tiles = ((N + 31) / 32) * ((M + 127) / 128);
    int ksplit = 1;
    static int sk_kmin = -1, sk_per = -1;
    if (sk_kmin < 0) { const char* e = getenv("RLCF_SKINNY_KMIN"); sk_kmin = e ? atoi(e) : 1024; }
    if (sk_per < 0) { const char* e = getenv("RLCF_SKINNY_KPER"); sk_per = e ? atoi(e) : 256; }
    if (K >= sk_kmin && tiles < 192 && ws && N % 4 == 0) {
        ksplit = std::min(K / sk_per, std::max(1, 256 / tiles));
        while (ksplit > 1 && (size_t)ksplit * M * N * sizeof(float) > ws_bytes) --ksplit;
        if (amax_in && !inv_scale_scratch) ksplit = 1;
    }
    g.ksplit = ksplit; g.ws = ws;
    if (ksplit > 1 && amax_in) { s.inv_scale_out = inv_scale_scratch; g.alpha_dev = inv_scale_scratch; }
    gemm_skinny_x3_kernel<<<dim3((N + 31) / 32, (M + 127) / 128, ksplit), dim3(256), 0, st>>>(s);
    RLCF_LAUNCH_CHECK();
    if (ksplit > 1) {
        const long groups = (long)M * (N / 4);
        gemm_x3_splitk_reduce_kernel<<<dim3((unsigned)std::min<long>((groups + 255) / 256, 2048)), dim3(256), 0, st>>>(g);
        RLCF_LAUNCH_CHECK();
    }
    g_last_x3_variant = 4;
    return RLCF_OK;
}

// bound-derived output scale of the NEXT pair-emitting launch of this thread (GemmX3Args::bnd_*; resnet.hip's conv_pairs): consumed by
// launch_gemm_f16x3 / launch_gemm_f16x3_conv3x3
struct X3Bound { const float* in; const float* res; float gain, bmax; float* out2; };
static thread_local const _Float16* g_next_wpk = nullptr;      // packed hi-only copy of the NEXT launch's weight (GemmX3Args::Wpk), row length K halves
void gemm_f16x3_next_packed_w(const void* wpk) { g_next_wpk = (const _Float16*)wpk; }
static thread_local const float* g_next_col_scale = nullptr;      // per-output-column factor of the NEXT launch (GemmX3Args::col_scale)
void gemm_f16x3_next_col_scale(const float* cs) { g_next_col_scale = cs; }
static thread_local X3Bound g_next_bound = {nullptr, nullptr, 0.f, 0.f, nullptr};
void gemm_f16x3_next_bound(const float* amax_in, const float* amax_res, float gain, float bmax, float* out2) {
    g_next_bound = X3Bound{amax_in, amax_res, gain, bmax, out2};
}
// what the caller announced for THIS launch: taken (and cleared) on entry of the launcher, before any argument check can return, so that an
// announcement can never outlive the call it was made for
struct X3Next { X3Bound b; const float* col_scale; const _Float16* wpk; };
static inline X3Next x3_take_next() {
    const X3Next n{g_next_bound, g_next_col_scale, g_next_wpk};
    g_next_bound = X3Bound{nullptr, nullptr, 0.f, 0.f, nullptr};
    g_next_col_scale = nullptr;
    g_next_wpk = nullptr;
    return n;
}
static inline void x3_apply_next(GemmX3Args& g, const X3Next& n) {
    g.bnd_in = n.b.in; g.bnd_res = n.b.res; g.bnd_gain = n.b.gain; g.bnd_bmax = n.b.bmax; g.bnd_out2 = n.b.out2;
    g.col_scale = n.col_scale;
}
int g_last_x3_variant = 0;          // 1 = 128x128 register-staged kernel, 2 = 256x128 DMA-ring kernel (profiling tag)
// splitk_ws / splitk_ws_bytes: caller-owned scratch for the split-K form of the small-grid kernel (the engine sizes it once at
// create); without it those shapes run unsplit.  Nothing is allocated here.
int launch_gemm_f16x3(const void* Ahi, const void* Alo, int lda, const void* Whi, const void* Wlo, int ldw, const float* bias,
                      const float* residual, int ldr, const float* aux, int ldaux, float* C, int ldc, void* Chi, void* Clo, int ldch,
                      int M, int N, int K, float alpha, int epilogue, hipStream_t st, const float* alpha_dev, unsigned int* amax_out, int c_il,
                      float* splitk_ws, size_t splitk_ws_bytes, int single, const float* out_scale_dev, unsigned* sk_epoch) {
    const X3Next next = x3_take_next();
    RLCF_ARG_CHECK(M > 0 && N > 0 && K > 0 && K % X3_BK == 0 && lda % 8 == 0 && ldw % 8 == 0);
    const bool wlo0 = single == 2;           // single: 0 = split-f16 (three passes), 1 = plain f16 operands (RLCF_PREC_F16), 2 = split-f16 with an all-zero W lo part (two passes)
    if (wlo0) single = 0;
    // packed hi-only W (announced by the caller, plain f16 [N, K]): K tiles come in pairs.  RLCF_X3_WPK=0: interleaved W rows (A/B measurements)
    static int wpk_on = -1;
    if (wpk_on < 0) { const char* e = getenv("RLCF_X3_WPK"); wpk_on = e ? atoi(e) : 1; }
    const bool wpk_ok = wlo0 && next.wpk && wpk_on && K % 64 == 0 && (size_t)N * K < ((size_t)1 << 31);
    // stream-K scratch (caller-owned, X3_WS_BYTES): slabs from the start of the workspace, flag words in its last X3_SK_FLAG_BYTES;
    // *sk_epoch = the caller's launch counter for THIS workspace (0: flags not yet zeroed)
    const bool sk_avail = sk_epoch && splitk_ws && splitk_ws_bytes >= (size_t)X3_SK_FLAG_BYTES + 8 * (size_t)X3_SK_SLAB_BYTES;
    unsigned* sk_flags = sk_avail ? (unsigned*)((char*)splitk_ws + splitk_ws_bytes - X3_SK_FLAG_BYTES) : nullptr;
    if (sk_avail) splitk_ws_bytes -= X3_SK_FLAG_BYTES;
    RLCF_ARG_CHECK(Ahi && Alo && Whi && Wlo && (C || (Chi && (Clo || single))));
    if (single) {       // plain f16 operands: a row of K halves == an interleaved pair row of K/2 logical columns (see GemmX3Args)
        RLCF_ARG_CHECK(K % 64 == 0 && Alo == (const void*)((const _Float16*)Ahi + 32) && Wlo == (const void*)((const _Float16*)Whi + 32));
        K /= 2;
    }
    GemmX3Args g{};
    g.Ahi = (const _Float16*)Ahi; g.Alo = (const _Float16*)Alo; g.lda = lda; g.Whi = (const _Float16*)Whi; g.Wlo = (const _Float16*)Wlo;
    g.ldw = ldw; g.bias = bias; g.residual = residual; g.ldr = ldr; g.aux = aux; g.ldaux = ldaux; g.C = C; g.ldc = ldc;
    g.Chi = (_Float16*)Chi; g.Clo = (_Float16*)Clo; g.ldch = ldch; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.epilogue = epilogue;
    g.alpha_dev = alpha_dev; g.amax_out = amax_out; g.c_il = c_il; g.out_scale_dev = out_scale_dev;
    x3_apply_next(g, next);
    g.kstep = (Alo == (const void*)((const _Float16*)Ahi + 32)) ? 64 : X3_BK;     // interleaved [hi32|lo32] blocks
    RLCF_ARG_CHECK((g.kstep == 64) == (Wlo == (const void*)((const _Float16*)Whi + 32)));   // both operands in the same layout
    const size_t sh = (size_t)2 * 4 * X3_TILE * sizeof(_Float16);
#define X3_LDS(fn, bytes) do { int rc_ = rlcf_func_lds((const void*)(fn), (bytes)); if (rc_ != RLCF_OK) return rc_; } while (0)
    const int blocks2 = ((M + V2_BM - 1) / V2_BM) * ((N + V2_BN - 1) / V2_BN);
    static int force = -1;                                   // RLCF_X3_KERNEL=1|2 pins a variant (benchmarks)
    if (force < 0) { const char* e = getenv("RLCF_X3_KERNEL"); force = e ? atoi(e) : 0; }
    static int nofast = -1;
    if (nofast < 0) { const char* e = getenv("RLCF_X3_NOFASTEPI"); nofast = e ? atoi(e) : 0; }
    static int x3nt = -1;
    if (x3nt < 0) { const char* e = getenv("RLCF_X3_NT"); x3nt = e ? atoi(e) : 1; }      // (on since round 5: +0.5-0.8 % on the layer's four products, bit-identical results; fabric traffic unchanged)
    static int linest = -1;                                  // RLCF_X3_LINEST=0: the pair epilogues keep the 8-byte hi / lo stores (A/B)
    if (linest < 0) { const char* e = getenv("RLCF_X3_LINEST"); linest = e ? atoi(e) : 1; }
    g.no_fast_epi = (nofast ? 1 : 0) | (x3nt ? 2 : 0) | (linest ? 0 : 4);
    static int tgroup = -1;
    if (tgroup < 0) { const char* e = getenv("RLCF_X3_GROUP"); tgroup = e ? atoi(e) : 0; }
    g.tile_group = tgroup;
    static int stagger = -1;
    if (stagger < 0) { const char* e = getenv("RLCF_X3_STAGGER"); stagger = e ? atoi(e) : 0; }
    g.stagger = stagger;
    const bool v2_ok = N % 4 == 0 && ldc % 4 == 0 && ldr % 4 == 0 && ldaux % 4 == 0 && ldch % 4 == 0;
    const int blocks3 = ((M + V3_BM - 1) / V3_BM) * ((N + V3_BN - 1) / V3_BN);
    // tile choice: both big kernels run one block per CU, so a launch costs ceil(tiles/256) block rounds; a 256x128 round takes
    // ~0.62 of a 256x256 round (half the work at ~20 % lower efficiency).  Measured at M = 12608 (one test image): N = 768 runs
    // 61 us as one 59 %-full 256x256 round against 70 us as two 256x128 rounds; K = 3072: 188 against 233 us.
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    // Stream-K tail (see the kernel): the tiles of the last, partly filled round are shared out over the CUs by K steps.  Its cost in
    // tile rounds = (K steps of the longest range + ~5 steps' worth of second launch and slab hand-over) / (nk + ~3 for a whole tile's
    // epilogue).  MEASURED SLOWER THAN THE PLAIN LAUNCH on every shape tried (profiles/r4_gemm_tile_overhead.txt: one image's token
    // matrix, 150 tiles on 256 CUs: 63.5 -> 100.5 us; out_proj at 20 images per pass, 11.54 rounds: 898 -> 943 us) — the partial tiles
    // are 256 KB each, a round's worth of them is 30-64 MB written through to memory and read back, ~25 us of the chip's bandwidth
    // against the ~35 us the balanced tail saves — so it is OFF by default; RLCF_X3_SK=1 switches it on (correct, bit-reproducible)
    static int sk_mode = -1;
    if (sk_mode < 0) { const char* e = getenv("RLCF_X3_SK"); sk_mode = e ? atoi(e) : 0; }
    int sk_first = 0, sk_blocks = 0;
    double cost3 = (double)((blocks3 + ncu - 1) / ncu);
    {
        const int nkt3 = K / X3_BK, R = blocks3 % ncu, min_steps = 6;
        if (sk_mode && sk_avail && !single && g.kstep == 64 && v2_ok && ncu % 8 == 0 && ncu <= 256 && R > 0 && blocks3 >= 64 && nkt3 >= 2 * min_steps) {
            int S = std::min(ncu, 8 * (int)((long)R * nkt3 / (8 * min_steps)));
            S = std::min(S, (int)(std::min<size_t>(X3_SK_MAX_BLOCKS, splitk_ws_bytes / X3_SK_SLAB_BYTES) / 8 * 8));
            if (S >= 8 * ((R + 7) / 8)) {                    // every range <= nk units: at most two tiles each
                const int per = (((R + 7) / 8) * nkt3 + S / 8 - 1) / (S / 8);       // the longest range (the chunk with the most tiles)
                const double tail = (per + 5.0) / (nkt3 + 3.0);
                if (tail < 0.9) { sk_first = blocks3 - R; sk_blocks = S; cost3 = (double)(blocks3 / ncu) + tail; }
            }
        }
    }
    // (0.62 since round 4, re-measured with W from HBM: [12608, 3072, 768] 5 rounds of 256x128 = 181.6 us against 3 rounds of 256x256 =
    // 175.0 us; [81900, 512, 512] 151.2 against 138.3 us; [81900, 512, 2048] 482 against 452 us — tools/gemm_mid_bench.py)
    const double cost2 = 0.62 * (double)((blocks2 + ncu - 1) / ncu);
    // 192 x 256 tiles (MT = 3): a round costs ~0.78 of a 256 x 256 round (3/4 of the products and of the epilogue, the same W tile
    // traffic); taken where the rounds it saves outweigh that — one image's token matrix against a W x W / W x 4W weight: 150 tiles =
    // one 59 %-full round -> 198 tiles = one 77 %-full round of 3/4 the length.  Only in the big-tile regime (>= 256 tiles of 256 x 128):
    // below it the 128 x 128 split-K kernel is the alternative and measures the same or better (M = 6272 convolutions of RN50x64's layer4).
    // RLCF_X3_MT3=0 switches it off, =2 forces it
    static int mt3 = -1;
    if (mt3 < 0) { const char* e = getenv("RLCF_X3_MT3"); mt3 = e ? atoi(e) : 1; }
    const int blocks3h = ((M + 191) / 192) * ((N + V3_BN - 1) / V3_BN);
    const double cost3h = 0.78 * (double)((blocks3h + ncu - 1) / ncu);
    const bool pick3h = v2_ok && !single && g.kstep == 64 && sk_blocks == 0 && force == 0 && blocks3h >= 128 && blocks2 >= 256 && !(C && Chi && residual) &&      // (conv3's f32 + pairs + identity epilogue: slower there)
                       
                        (mt3 == 2 || (mt3 == 1 && cost3h < 0.97 * std::min(cost3, blocks2 >= 256 ? cost2 : cost3)));
    if (pick3h) {
        const size_t sh3 = (size_t)8 * 64 * 68 * sizeof(float) > (size_t)2 * V3_STAGE ? (size_t)8 * 64 * 68 * sizeof(float) : (size_t)2 * V3_STAGE;
        if (wlo0 && wpk_ok) {
            g.Wpk = next.wpk; g.ldwpk = K;
            X3_LDS((gemm_nt_f16x3_v3i_kernel<false, false, false, 3, 2>), sh3);
            gemm_nt_f16x3_v3i_kernel<false, false, false, 3, 2><<<dim3(blocks3h), dim3(512), sh3, st>>>(g);
        } else if (wlo0) {
            X3_LDS((gemm_nt_f16x3_v3i_kernel<false, false, false, 3, 1>), sh3);
            gemm_nt_f16x3_v3i_kernel<false, false, false, 3, 1><<<dim3(blocks3h), dim3(512), sh3, st>>>(g);
        } else {
        X3_LDS((gemm_nt_f16x3_v3i_kernel<false, false, false, 3>), sh3);
        gemm_nt_f16x3_v3i_kernel<false, false, false, 3><<<dim3(blocks3h), dim3(512), sh3, st>>>(g);
        }
        g_last_x3_variant = 5;
        RLCF_LAUNCH_CHECK();
        return RLCF_OK;
    }
    // RLCF_PREC_F16 at the big-tile regime: the dedicated single-pass kernel (gemm_f16.hip) instead of the K/2 alias of the pair kernels
    if (single && force == 0 && v2_ok && blocks2 >= 256 && cost3 <= cost2 &&
        gemm_f16_p8_ok(C, Chi, residual, aux, epilogue, alpha_dev, amax_out, out_scale_dev, N, 2 * K, lda, ldw, ldc, ldr, ldch)) {
        const int rc = launch_gemm_f16_p8(Ahi, lda, Whi, ldw, bias, residual, ldr, C, ldc, Chi, ldch, M, N, 2 * K, alpha, epilogue, tgroup, st);
        g_last_x3_variant = 3;
        return rc;
    }
    const bool pick3 = (blocks2 >= 256 || sk_blocks > 0) && cost3 <= cost2;
    // smallest grid of 256x128 tiles that takes the 8-wave kernel: 160 (two waves per SIMD on 160+ CUs beat one wave per SIMD on twice as
    // many 128x128 workgroups: [4095, 1536, 512] 30.8 -> 27.3 us; below ~100 tiles the small tile wins).  RLCF_X3_V2MIN=256: the old rule
    static int v2min = -1;
    if (v2min < 0) { const char* e = getenv("RLCF_X3_V2MIN"); v2min = e ? atoi(e) : 160; }
    // RLCF_X3_V4=1: the 4-wave form of the 256x256 tile where it applies (interleaved pairs, compile-time epilogues)
    static int v4 = -1;
    if (v4 < 0) { const char* e = getenv("RLCF_X3_V4"); v4 = e ? atoi(e) : 0; }
    {
        const bool f32o = C != nullptr, pair = Chi != nullptr, res = residual != nullptr;
        const bool fast = !amax_out && !alpha_dev && !aux && !nofast &&
                          ((epilogue == RLCF_EPI_NONE && f32o && !pair) || (epilogue == RLCF_EPI_QUICKGELU && !f32o && pair && !res) ||
                           (epilogue == RLCF_EPI_NONE && !f32o && pair && !res));
        if (v4 && v2_ok && !single && g.kstep == 64 && fast && (force == 3 || (force == 0 && pick3))) {
            const size_t sh4 = (size_t)5 * 32768;
            X3_LDS(gemm_nt_f16x3_v4_kernel<0>, sh4);
            X3_LDS(gemm_nt_f16x3_v4_kernel<1>, sh4);
            X3_LDS(gemm_nt_f16x3_v4_kernel<2>, sh4);
            if (v4 == 2) gemm_nt_f16x3_v4_kernel<1><<<dim3(blocks3), dim3(256), sh4, st>>>(g);
            else if (v4 == 3) gemm_nt_f16x3_v4_kernel<2><<<dim3(blocks3), dim3(256), sh4, st>>>(g);
            else gemm_nt_f16x3_v4_kernel<0><<<dim3(blocks3), dim3(256), sh4, st>>>(g);
            g_last_x3_variant = 3;
            RLCF_LAUNCH_CHECK();
            return RLCF_OK;
        }
    }
    if (v2_ok && (force == 3 || (force == 0 && pick3))) {
        const size_t sh3 = (size_t)8 * 64 * 68 * sizeof(float) > (size_t)2 * V3_STAGE ? (size_t)8 * 64 * 68 * sizeof(float) : (size_t)2 * V3_STAGE;
        if (g.kstep == 64 && single) {
            X3_LDS(gemm_nt_f16x3_v3i_kernel<true>, sh3);
            gemm_nt_f16x3_v3i_kernel<true><<<dim3(blocks3), dim3(512), sh3, st>>>(g);
        } else if (g.kstep == 64 && sk_blocks > 0) {
            if (*sk_epoch == 0) RLCF_HIP_CHECK(hipMemsetAsync(sk_flags, 0, X3_SK_FLAG_BYTES, st));      // first use of this workspace
            if (++*sk_epoch == 0) { RLCF_HIP_CHECK(hipMemsetAsync(sk_flags, 0, X3_SK_FLAG_BYTES, st)); *sk_epoch = 1; }     // (wrapped)
            g.sk_first = sk_first; g.sk_blocks = sk_blocks; g.sk_epoch = *sk_epoch; g.sk_flags = sk_flags; g.sk_ws = splitk_ws;
            X3_LDS(gemm_nt_f16x3_v3i_kernel<false>, sh3);
            X3_LDS((gemm_nt_f16x3_v3i_kernel<false, false, true>), sh3);
            // whole-tile workgroups for the full rounds, then the stream-K launch for the rest (first pieces, then second pieces)
            if (sk_first > 0) gemm_nt_f16x3_v3i_kernel<false><<<dim3(sk_first), dim3(512), sh3, st>>>(g);
            gemm_nt_f16x3_v3i_kernel<false, false, true><<<dim3(2 * sk_blocks), dim3(512), sh3, st>>>(g);
        } else if (g.kstep == 64 && wlo0 && wpk_ok) {
            g.Wpk = next.wpk; g.ldwpk = K;
            X3_LDS((gemm_nt_f16x3_v3i_kernel<false, false, false, 4, 2>), sh3);
            gemm_nt_f16x3_v3i_kernel<false, false, false, 4, 2><<<dim3(blocks3), dim3(512), sh3, st>>>(g);
        } else if (g.kstep == 64 && wlo0) {
            X3_LDS((gemm_nt_f16x3_v3i_kernel<false, false, false, 4, 1>), sh3);
            gemm_nt_f16x3_v3i_kernel<false, false, false, 4, 1><<<dim3(blocks3), dim3(512), sh3, st>>>(g);
        } else if (g.kstep == 64) {
            X3_LDS(gemm_nt_f16x3_v3i_kernel<false>, sh3);
            gemm_nt_f16x3_v3i_kernel<false><<<dim3(blocks3), dim3(512), sh3, st>>>(g);
        } else {
            X3_LDS(gemm_nt_f16x3_v3_kernel, sh3);
            gemm_nt_f16x3_v3_kernel<<<dim3(blocks3), dim3(512), sh3, st>>>(g);
        }
        g_last_x3_variant = 3;
        RLCF_LAUNCH_CHECK();
        return RLCF_OK;
    }
    if (v2_ok && (force == 2 || (force == 0 && blocks2 >= v2min))) {
        const size_t sh2 = (size_t)3 * V2_STAGE;
        if (single) {
            X3_LDS((gemm_nt_f16x3_v2_kernel<4, true>), sh2);
            gemm_nt_f16x3_v2_kernel<4, true><<<dim3(blocks2), dim3(512), sh2, st>>>(g);
        } else {
            if (wlo0) {
                X3_LDS((gemm_nt_f16x3_v2_kernel<4, false, 2, true>), sh2);
                gemm_nt_f16x3_v2_kernel<4, false, 2, true><<<dim3(blocks2), dim3(512), sh2, st>>>(g);
            } else {
            X3_LDS((gemm_nt_f16x3_v2_kernel<4, false>), sh2);
            gemm_nt_f16x3_v2_kernel<4, false><<<dim3(blocks2), dim3(512), sh2, st>>>(g);
            }
        }
        g_last_x3_variant = 2;
        RLCF_LAUNCH_CHECK();
        return RLCF_OK;
    }
    static int nov2s = -1;                                   // RLCF_X3_NOV2S=1: small grids back on the register-staged kernel
    if (nov2s < 0) { const char* e = getenv("RLCF_X3_NOV2S"); nov2s = e ? atoi(e) : 0; }
    if (v2_ok && !nov2s && (force == 4 || force == 0)) {
        const int blocks2s = ((M + 127) / 128) * ((N + V2_BN - 1) / V2_BN);
        const size_t sh2s = (size_t)3 * (2 * 128 * 64 + 2 * V2_BN * 64) > (size_t)4 * 64 * 68 * sizeof(float)
                                ? (size_t)3 * (2 * 128 * 64 + 2 * V2_BN * 64) : (size_t)4 * 64 * 68 * sizeof(float);
        // Grids of 65-128 tiles of 128x128 leave half of the chip idle, and a workgroup's K tile is bound by its own MFMA issue (~0.55 us,
        // whatever the ring depth or the wave count): they run on 64x128 tiles (<2, ., 1>: four waves of 32 x 64, twice the workgroups —
        // [1182, 768, 768] 20.0 -> 14.2 us, [4095, 512, 2048] 46.1 -> 35.0, [771, 1024, 4096] 84 -> 62); larger grids on the eight-wave
        // form of the 128x128 tile (<4, ., 1>: +2-3 %).  Same products in the same order: bit-identical (tools/gemm_mid_bench.py prints a
        // checksum).  RLCF_V2S8=0: the four-wave 128x128 kernel everywhere (measurements)
        static int v2s8 = -1;
        if (v2s8 < 0) { const char* e = getenv("RLCF_V2S8"); v2s8 = e ? atoi(e) : 1; }
        // (grids of <= 64 tiles keep the 128x128 tile with K slices: [1182, 768, 3072] 27 us so against 37 us on sliced half tiles)
        const bool half_m = v2s8 && !single && blocks2s > 64 && blocks2s <= 128;
        const bool eight = v2s8 && !single && !half_m;
        const int blocks_h = ((M + 63) / 64) * ((N + V2_BN - 1) / V2_BN);
        const int nblk = half_m ? blocks_h : blocks2s;           // workgroups per K slice
        if (single) X3_LDS((gemm_nt_f16x3_v2_kernel<2, true>), sh2s);
        else if (eight && wlo0) X3_LDS((gemm_nt_f16x3_v2_kernel<4, false, 1, true>), sh2s);
        else if (eight) X3_LDS((gemm_nt_f16x3_v2_kernel<4, false, 1>), sh2s);
        else if (half_m && wlo0) X3_LDS((gemm_nt_f16x3_v2_kernel<2, false, 1, true>), sh2s);
        else if (half_m) X3_LDS((gemm_nt_f16x3_v2_kernel<2, false, 1>), sh2s);
        else if (wlo0) X3_LDS((gemm_nt_f16x3_v2_kernel<2, false, 2, true>), sh2s);
        else X3_LDS((gemm_nt_f16x3_v2_kernel<2, false>), sh2s);
        // few tiles and a long K loop (one image's token matrix against a W x 4W / W x 3W weight): split the K loop over blockIdx.y
        // and finish in a reduce + epilogue pass (RLCF_X3_NOSPLITK=1 switches it off)
        static int nosplit = -1;
        if (nosplit < 0) { const char* e = getenv("RLCF_X3_NOSPLITK"); nosplit = e ? atoi(e) : 0; }
        const int nkt = K / X3_BK;
        int ksplit = 1;
        if (!nosplit && nblk <= 128 && nkt >= 48) ksplit = nkt >= 96 ? 4 : 3;
        // ... and the 24-tile K loops of grids below a quarter of the chip (one image's reward tower: [1182, 768, 768] = 60 workgroups)
        // in three slices (bench.py --batch 1, with the 256x128 threshold below: 75.2-75.7 -> 76.6 images/s).  RLCF_X3_SPLIT24=0: off
        static int split24 = -1;
        if (split24 < 0) { const char* e = getenv("RLCF_X3_SPLIT24"); split24 = e ? atoi(e) : 2; }
        if (split24 && !nosplit && ksplit == 1 && nblk <= 64 && nkt >= 24) ksplit = split24 + 1;
        // a very long K loop over a small output (the weight gradient of a convolution: K = n*H*W = 10^5..10^6 rows): as many slices as
        // it takes to put ~384 workgroups on the chip, each at least 64 K tiles long, within the workspace
        if (!nosplit && nkt >= 1024 && nblk < 384 && splitk_ws) {
            const long by_ws = (long)(splitk_ws_bytes / ((size_t)M * N * sizeof(float)));
            const long want = std::min<long>(std::min<long>((384 + nblk - 1) / nblk, nkt / 64), std::min<long>(by_ws, 64));      // (<= 64 slices of >= 64 K tiles: no slice is empty)
            if (want > ksplit) ksplit = (int)want;
        }
        if (ksplit > 1 && (!splitk_ws || (size_t)ksplit * M * N * sizeof(float) > splitk_ws_bytes)) ksplit = 1;
        g.ksplit = ksplit; g.ws = splitk_ws;
        if (single) gemm_nt_f16x3_v2_kernel<2, true><<<dim3(blocks2s, ksplit), dim3(256), sh2s, st>>>(g);
        else if (eight && wlo0) gemm_nt_f16x3_v2_kernel<4, false, 1, true><<<dim3(blocks2s, ksplit), dim3(512), sh2s, st>>>(g);
        else if (eight) gemm_nt_f16x3_v2_kernel<4, false, 1><<<dim3(blocks2s, ksplit), dim3(512), sh2s, st>>>(g);
        else if (half_m && wlo0) gemm_nt_f16x3_v2_kernel<2, false, 1, true><<<dim3(blocks_h, ksplit), dim3(256), sh2s, st>>>(g);
        else if (half_m) gemm_nt_f16x3_v2_kernel<2, false, 1><<<dim3(blocks_h, ksplit), dim3(256), sh2s, st>>>(g);
        else if (wlo0) gemm_nt_f16x3_v2_kernel<2, false, 2, true><<<dim3(blocks2s, ksplit), dim3(256), sh2s, st>>>(g);
        else gemm_nt_f16x3_v2_kernel<2, false><<<dim3(blocks2s, ksplit), dim3(256), sh2s, st>>>(g);
        g_last_x3_variant = 1;
        RLCF_LAUNCH_CHECK();
        if (ksplit > 1) {
            const long groups = (long)M * (N / 4);
            gemm_x3_splitk_reduce_kernel<<<dim3((unsigned)std::min<long>((groups + 255) / 256, 2048)), dim3(256), 0, st>>>(g);
            RLCF_LAUNCH_CHECK();
        }
        return RLCF_OK;
    }
    const int blocks = ((M + X3_BM - 1) / X3_BM) * ((N + X3_BN - 1) / X3_BN);
    if (single) {
        X3_LDS(gemm_nt_f16x3_kernel<true>, sh);
        gemm_nt_f16x3_kernel<true><<<dim3(blocks), dim3(256), sh, st>>>(g);
    } else {
        X3_LDS(gemm_nt_f16x3_kernel<false>, sh);
        gemm_nt_f16x3_kernel<false><<<dim3(blocks), dim3(256), sh, st>>>(g);
    }
    g_last_x3_variant = 1;
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// Implicit 3x3 convolution (stride 1, pad 1) on the 256x256 kernel: act = the NHWC activation [n*H*W, Cin] as interleaved operand
// pairs (row stride 2*Cin halves), W = the convolution weight in the GEMM layout [Cout, 9*Cin] ((ky, kx, c) order) as interleaved pairs.
// No patch matrix exists: a workgroup's DMA reads the K tile of tap (ky, kx) straight from pixel (y + ky - 1, x + kx - 1) of the
// activation, or from `zpage` (>= 1 KB of zeros) outside the image.  Same products in the same order as the patch-matrix form.
bool gemm_f16x3_conv3x3_ok(int M, int N, int Cin) {
    const long tiles = (long)((M + V3_BM - 1) / V3_BM) * ((N + V3_BN - 1) / V3_BN);
    return Cin % 32 == 0 && N % 4 == 0 && tiles >= 192;
}
int launch_gemm_f16x3_conv3x3(const void* act_pairs, int n, int H, int W, int Cin, const void* Wpairs, int Cout, const float* bias,
                              const float* residual, int ldr, float* C, int ldc, float alpha, int epilogue, const float* alpha_dev,
                              unsigned int* amax_out, const void* zpage, hipStream_t st, void* Cpairs, const float* out_scale_dev, int wlo0) {
    const X3Next next = x3_take_next();
    const int M = n * H * W, K = 9 * Cin;
    RLCF_ARG_CHECK(act_pairs && Wpairs && (C || Cpairs) && zpage && gemm_f16x3_conv3x3_ok(M, Cout, Cin) && ldc % 4 == 0 && ldr % 4 == 0);
    GemmX3Args g{};
    g.Ahi = (const _Float16*)act_pairs; g.Alo = g.Ahi + 32; g.lda = 2 * Cin;
    g.Whi = (const _Float16*)Wpairs; g.Wlo = g.Whi + 32; g.ldw = 2 * K;
    g.bias = bias; g.residual = residual; g.ldr = ldr; g.C = C; g.ldc = ldc; g.M = M; g.N = Cout; g.K = K; g.alpha = alpha;
    g.epilogue = epilogue; g.alpha_dev = alpha_dev; g.amax_out = amax_out; g.kstep = 64;
    g.conv_C = Cin; g.conv_H = H; g.conv_W = W; g.zpage = (const _Float16*)zpage;
    x3_apply_next(g, next);
    if (Cpairs) { RLCF_ARG_CHECK(Cout % 32 == 0); g.Chi = (_Float16*)Cpairs; g.Clo = g.Chi + 32; g.ldch = 2 * Cout; g.c_il = 1; g.out_scale_dev = out_scale_dev; }
    static int nofast = -1;
    if (nofast < 0) { const char* e = getenv("RLCF_X3_NOFASTEPI"); nofast = e ? atoi(e) : 0; }
    g.no_fast_epi = nofast;
    const int blocks3 = ((M + V3_BM - 1) / V3_BM) * ((Cout + V3_BN - 1) / V3_BN);
    const size_t sh3 = (size_t)8 * 64 * 68 * sizeof(float) > (size_t)2 * V3_STAGE ? (size_t)8 * 64 * 68 * sizeof(float) : (size_t)2 * V3_STAGE;
    static int wpk_on = -1;
    if (wpk_on < 0) { const char* e = getenv("RLCF_X3_WPK"); wpk_on = e ? atoi(e) : 1; }
    if (wlo0 && next.wpk && wpk_on && K % 64 == 0 && (size_t)Cout * K < ((size_t)1 << 31)) {       // hi-only W rows, two K tiles per row block
        g.Wpk = next.wpk; g.ldwpk = K;
        { int rc_ = rlcf_func_lds((const void*)(gemm_nt_f16x3_v3i_kernel<false, true, false, 4, 2>), sh3); if (rc_ != RLCF_OK) return rc_; }
        gemm_nt_f16x3_v3i_kernel<false, true, false, 4, 2><<<dim3(blocks3), dim3(512), sh3, st>>>(g);
    } else if (wlo0) {
        { int rc_ = rlcf_func_lds((const void*)(gemm_nt_f16x3_v3i_kernel<false, true, false, 4, 1>), sh3); if (rc_ != RLCF_OK) return rc_; }
        gemm_nt_f16x3_v3i_kernel<false, true, false, 4, 1><<<dim3(blocks3), dim3(512), sh3, st>>>(g);
    } else {
    { int rc_ = rlcf_func_lds((const void*)(gemm_nt_f16x3_v3i_kernel<false, true>), sh3); if (rc_ != RLCF_OK) return rc_; }
    gemm_nt_f16x3_v3i_kernel<false, true><<<dim3(blocks3), dim3(512), sh3, st>>>(g);
    }
    g_last_x3_variant = 3;
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

// does every element of w * scale sit on the f16 grid (its lo half is zero)?  flag[0] |= 1 otherwise
__global__ void f16_grid_check_kernel(const float* __restrict__ w, int64_t n, float scale, int* __restrict__ flag) {
    if (*(volatile int*)flag) return;                      // (another workgroup has found an off-grid element already)
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n && !bad; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = w[i] * scale;
        if ((float)(_Float16)v != v) bad = 1;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0 && *(volatile int*)flag == 0) atomicOr(flag, 1);
}
int launch_f16_grid_check(const float* w, int64_t n, float scale, int* flag, hipStream_t st) {
    RLCF_ARG_CHECK(w && flag && n > 0);
    f16_grid_check_kernel<<<dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, st>>>(w, n, scale, flag);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
// x -> (hi, lo): hi = f16(x), lo = f16((x - hi) * 2^11).  8 elements per thread.
// il: interleaved output — 8-element group i (32-element block i/4, position i%4) lands at 8*((i/4)*8 + i%4) (hi) and 32 halves later (lo)
__device__ __forceinline__ int64_t split_dst(int64_t i, int il) { return il ? ((i >> 2) << 3) + (i & 3) : i; }
// GELU: the operand is QuickGELU(x) (model.py:166-168, the exact form of quickgelu_kernel): the saved-forward pass of the tuning paths
// keeps the pre-activation for the backward and feeds c_proj from this one pass instead of a QuickGELU pass plus a split pass
template <bool GELU>
__global__ void split_f16x2_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int64_t n8, float scale, int il) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = ((const float4*)x)[2 * i], b = ((const float4*)x)[2 * i + 1];
        if constexpr (GELU) {
            a.x = quick_gelu(a.x); a.y = quick_gelu(a.y); a.z = quick_gelu(a.z); a.w = quick_gelu(a.w);
            b.x = quick_gelu(b.x); b.y = quick_gelu(b.y); b.z = quick_gelu(b.z); b.w = quick_gelu(b.w);
        }
        const float v[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale, b.x * scale, b.y * scale, b.z * scale, b.w * scale};
        h16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 hh = (_Float16)v[e];
            vh[e] = hh;
            vl[e] = (_Float16)(v[e] - (float)hh);
        }
        ((h16x8*)hi)[split_dst(i, il)] = vh;
        if (lo) ((h16x8*)lo)[split_dst(i, il)] = vl;      // (lo null: plain f16 copy, RLCF_PREC_F16)
    }
}
// data-dependent pre-scale for operands without a known range (ResNet activations): s = 2^k lifting max|x| into [2^9, 2^10);
// out[0] = s (read by the split kernel), out[1] = 1/s (folded into the GEMM's alpha)
__global__ void dyn_scale_kernel(const float* __restrict__ amax, float* __restrict__ out) {
    const float mx = amax[0];
    int sh = 0;
    if (mx > 0.f && mx < INFINITY) sh = 9 - (int)floorf(log2f(mx));
    sh = sh < -40 ? -40 : (sh > 40 ? 40 : sh);
    out[0] = ldexpf(1.0f, sh);
    out[1] = ldexpf(1.0f, -sh);
}
__global__ void split_f16x2_dyn_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int64_t n8,
                                       const float* __restrict__ scale_dev, int il) {
    const float scale = scale_dev[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = ((const float4*)x)[2 * i], b = ((const float4*)x)[2 * i + 1];
        const float v[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale, b.x * scale, b.y * scale, b.z * scale, b.w * scale};
        h16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 hh = (_Float16)v[e];
            vh[e] = hh;
            vl[e] = (_Float16)(v[e] - (float)hh);
        }
        ((h16x8*)hi)[split_dst(i, il)] = vh;
        if (lo) ((h16x8*)lo)[split_dst(i, il)] = vl;      // (lo null: plain f16 copy, RLCF_PREC_F16)
    }
}
// scratch: 3 floats on the device {max|x|, s, 1/s}
int launch_dyn_scale(const float* x, int64_t n, float* scratch3, hipStream_t st) {
    RLCF_ARG_CHECK(n > 0 && scratch3);
    int rc = launch_absmax(x, n, scratch3, st);
    if (rc != RLCF_OK) return rc;
    dyn_scale_kernel<<<dim3(1), dim3(1), 0, st>>>(scratch3, scratch3 + 1);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_dyn_scale_from(const float* amax_dev, float* scale2, hipStream_t st) {
    dyn_scale_kernel<<<dim3(1), dim3(1), 0, st>>>(amax_dev, scale2);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_split_f16x2_dev(const float* x, void* hi, void* lo, int64_t n, const float* scale_dev, hipStream_t st, int il) {
    RLCF_ARG_CHECK(n > 0 && n % 8 == 0 && scale_dev);
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    split_f16x2_dyn_kernel<<<dim3(blocks), dim3(256), 0, st>>>(x, (_Float16*)hi, (_Float16*)lo, n / 8, scale_dev, il);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_split_f16x2_dyn(const float* x, void* hi, void* lo, int64_t n, float* scratch3, hipStream_t st, int il) {
    RLCF_ARG_CHECK(n > 0 && n % 8 == 0 && scratch3);
    int rc = launch_dyn_scale(x, n, scratch3, st);
    if (rc != RLCF_OK) return rc;
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    split_f16x2_dyn_kernel<<<dim3(blocks), dim3(256), 0, st>>>(x, (_Float16*)hi, (_Float16*)lo, n / 8, scratch3 + 1, il);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
int launch_split_f16x2(const float* x, void* hi, void* lo, int64_t n, hipStream_t st, float scale, int il, int gelu) {
    RLCF_ARG_CHECK(n > 0 && n % 8 == 0);
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (gelu) split_f16x2_kernel<true><<<dim3(blocks), dim3(256), 0, st>>>(x, (_Float16*)hi, (_Float16*)lo, n / 8, scale, il);
    else split_f16x2_kernel<false><<<dim3(blocks), dim3(256), 0, st>>>(x, (_Float16*)hi, (_Float16*)lo, n / 8, scale, il);
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
