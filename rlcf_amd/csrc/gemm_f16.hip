// Single-pass f16 GEMM for the performance mode (RLCF_PREC_F16): plain f16 operands, ONE v_mfma_f32_32x32x16_f16 per product, f32
// accumulate — the arithmetic of the reference's own fp16-autocast GPU path (TPT/tpt_cls_rl.py:52; every nn.Linear of
// TPT/clip/model.py:171-192 under torch.cuda.amp.autocast).  Not parity-grade; the split-f16 kernels of gemm_f16x3.hip are.
//
// C[M,N] = epi(alpha * A.W^T + bias) (+ residual);  A [M,K] f16 row-major (lda halves), W [N,K] f16 row-major (ldw halves), K % 64 == 0.
//
// Structure (its own kernel, not the K/2 alias of the pair kernels): 256x256 block tile, BK = 64, eight waves as 2 (M) x 4 (N), each
// wave a contiguous 128x64 output block = 4x2 accumulator tiles of 32x32 (128 accumulator registers).  A K tile is FOUR half tiles of
// 128 rows x 128 B (16 KB): A_lo / A_hi = the first / second 64 rows of every wave's 128, B_lo / B_hi = the first / second 32 columns
// of every wave's 64 (the DMA's per-lane source address makes any row set free).  A K tile is worked through in four PHASES, one
// accumulator quadrant (64 rows x 32 columns x K = 64: 8 MFMAs) each:
//      P1 (A_lo, B_lo)   P2 (A_lo, B_hi)   P3 (A_hi, B_hi)   P4 (A_hi, B_lo)
// so that each phase reads ONE new operand part from LDS (P1: both) and each half-tile buffer is free again right after the phase that
// reads it last: A_lo after P1, B_hi after P2 (P3 re-uses P2's B_hi fragments from registers), A_hi after P3,
// B_lo after P4.  Every phase therefore re-fills the buffer the previous phase freed with a half tile of K tile t+2 (P1: B_lo of t+1)
// — 2 DMA instructions (global_load_lds_dwordx4, 1 KB each) per wave and phase, 1.75 K tiles of prefetch distance in TWO K tiles of
// LDS (128 KB), one counted s_waitcnt vmcnt(6) per K tile (never 0 inside the loop).
// A phase = [LDS reads of its operand part + its 2 DMA issues | s_barrier | 8 MFMAs | s_barrier]; the two wave groups (waves 0-3 /
// 4-7 = the two waves of every SIMD) run ONE barrier apart, so one wave of each SIMD is always in its pure-MFMA section while the
// other issues its reads and DMA (cdna_hip_programming.md, "256^2 8-phase template").
// Ordering rules kept (same source): a buffer is read one phase after the vmcnt that retires it (each wave waits BEFORE its barrier,
// the readers are past a later barrier); a buffer is re-staged one phase after its last read and every wave retires its reads
// (lgkmcnt(0)) BEFORE the barrier that ends its read section, so the other group's reads are complete too.
#include "gemm_x3.h"
#include <cstdlib>
#include <algorithm>

#define P8_HALF 16384                   // bytes of a half tile: 128 rows x 128 B
#define P8_PAR 65536                    // bytes of a K tile: A_lo | A_hi | B_lo | B_hi
#define P8_B0 32768

template <bool PRIO>
__global__ __launch_bounds__(512, 2) void gemm_nt_f16_p8_kernel(GemmX3Args g) {
    float am = 0.f;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][P8_PAR] (+ epilogue parking: 8 x 64 x 68 floats)
    const int tiles_n = (g.N + 255) / 256, tiles_m = (g.M + 255) / 256;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // every XCD takes a contiguous range of tiles, walked in groups of G tile rows: M-fastest (neighbours share a W tile), N-fastest for
    // problems 3-4 tiles wide (neighbours share the A panel) — the order of gemm_nt_f16x3_v3i_kernel
    const int G = g.tile_group % 100 > 0 ? g.tile_group % 100 : 8;
    const int per_group = G * tiles_n, grp = bid / per_group, first_m = grp * G;
    const int gsize = min(tiles_m - first_m, G), in_g = bid - grp * per_group;
    const bool nfast = g.tile_group >= 100 || (g.tile_group == 0 && tiles_n <= 4);
    const int m0 = (nfast ? first_m + in_g / tiles_n : first_m + in_g % gsize) * 256;
    const int n0 = (nfast ? in_g % tiles_n : in_g / gsize) * 256;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3, l32 = lane & 31, h = lane >> 5;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA: a half tile = 16 pieces of 8 rows x 128 B; wave w moves pieces 2w, 2w+1 = local rows 16w .. 16w+15.  LDS rows are 128 B; the
    // 16-B chunk c of local row r sits in slot c ^ ((r >> 1) & 7) (applied to the SOURCE address: the DMA writes lane-linearly), so the
    // 16 lanes of a ds_read_b128 group (16 consecutive rows, one chunk) cover all 64 banks.
    // source = scalar base of the tile's first row (+ the K tile) + a 32-bit per-lane byte offset inside the tile's rows
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const char* abase = (const char*)(g.Ahi + (size_t)m0 * g.lda);
    const char* wbase = (const char*)(g.Whi + (size_t)n0 * g.ldw);
    unsigned sa[2][2], sw[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lr = wave * 16 + j * 8 + (lane >> 3), c = ((lane & 7) ^ ((lr >> 1) & 7)) * 16;
            sa[hf][j] = (unsigned)(min((lr >> 6) * 128 + hf * 64 + (lr & 63), g.M - 1 - m0) * g.lda) * 2u + c;
            sw[hf][j] = (unsigned)(min((lr >> 5) * 64 + hf * 32 + (lr & 31), g.N - 1 - n0) * g.ldw) * 2u + c;
        }
#define P8_STAGE_A(hf, kt)                                                                                                          \
    {                                                                                                                               \
        char* d_ = smem + ((kt) & 1) * P8_PAR + (hf) * P8_HALF + wave_s * 2048;                                                      \
        const char* s_ = abase + (size_t)(kt) * 128;                                                                                \
        __builtin_amdgcn_global_load_lds((gptr_t)(s_ + sa[hf][0]), (lptr_t)d_, 16, 0, 0);                                            \
        __builtin_amdgcn_global_load_lds((gptr_t)(s_ + sa[hf][1]), (lptr_t)(d_ + 1024), 16, 0, 0);                                   \
    }
#define P8_STAGE_B(hf, kt)                                                                                                          \
    {                                                                                                                               \
        char* d_ = smem + ((kt) & 1) * P8_PAR + P8_B0 + (hf) * P8_HALF + wave_s * 2048;                                              \
        const char* s_ = wbase + (size_t)(kt) * 128;                                                                                \
        __builtin_amdgcn_global_load_lds((gptr_t)(s_ + sw[hf][0]), (lptr_t)d_, 16, 0, 0);                                            \
        __builtin_amdgcn_global_load_lds((gptr_t)(s_ + sw[hf][1]), (lptr_t)(d_ + 1024), 16, 0, 0);                                   \
    }
    const int nk = g.K / 64;
    // prologue: K tile 0 complete, then the three half tiles of K tile 1 the loop does not stage itself (B_lo of t+1 is staged by P1 of t)
    P8_STAGE_A(0, 0) P8_STAGE_B(0, 0) P8_STAGE_B(1, 0) P8_STAGE_A(1, 0)
    if (nk > 1) {
        P8_STAGE_A(0, 1) P8_STAGE_B(1, 1) P8_STAGE_A(1, 1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    const int swz = (l32 >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + h) ^ swz) * 16;
    const int aoff = (wm * 64 + l32) * 128, boff = P8_B0 + (wn * 32 + l32) * 128;
    h16x8 a[2][4], b[4];
#define P8_LDA(hf, par)                                                                                                             \
    _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2)                                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                            \
            a[i2][ks] = *(const h16x8*)(smem + (par) * P8_PAR + (hf) * P8_HALF + aoff + i2 * 4096 + coff[ks]);
#define P8_LDB(hf, par)                                                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) b[ks] = *(const h16x8*)(smem + (par) * P8_PAR + (hf) * P8_HALF + boff + coff[ks]);
#define P8_MMA(ib, j)                                                                                                               \
    {                                                                                                                               \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                            \
            _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2)                                                                        \
                acc[(ib) * 2 + i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i2][ks], b[ks], acc[(ib) * 2 + i2][j], 0, 0, 0);    \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                                    \
    }
// end of a phase's read section: this wave's LDS reads are complete BEFORE it arrives (the buffer may be re-staged by the other group
// right after this barrier)
#define P8_MID                                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
#define P8_END                                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
#define P8_KTILE(kt, par)                                                                                                           \
    {                                                                                                                               \
        P8_LDB(0, par) P8_LDA(0, par)                                                                                               \
        if ((kt) + 1 < nk) P8_STAGE_B(0, (kt) + 1)                                                                                  \
        P8_MID P8_MMA(0, 0) P8_END                                                                                                  \
        P8_LDB(1, par)                                                                                                              \
        if ((kt) + 2 < nk) P8_STAGE_A(0, (kt) + 2)                                                                                  \
        P8_MID P8_MMA(0, 1) P8_END                                                                                                  \
        P8_LDA(1, par)                                                                                                              \
        if ((kt) + 2 < nk) P8_STAGE_B(1, (kt) + 2)                                                                                  \
        P8_MID P8_MMA(1, 1) P8_END                                                                                                  \
        P8_LDB(0, par)                                                                                                              \
        if ((kt) + 2 < nk) {                                                                                                        \
            P8_STAGE_A(1, (kt) + 2)                                                                                                 \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                                        \
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                     \
        P8_MID P8_MMA(1, 0) P8_END                                                                                                  \
    }
    if (wm == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }       // the second group runs one barrier behind
    for (int kt = 0; kt < nk; kt += 2) {
        P8_KTILE(kt, 0)
        if (kt + 1 < nk) P8_KTILE(kt + 1, 1)
    }
    if (wm == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    // epilogue: the compile-time forms of the forward towers (gemm_x3.h), each wave through its own 17-KB LDS slice
    const int ek = x3_epilogue_kind(g);
    float* parkf = (float*)smem + wave * (64 * 68);
    __syncthreads();
    X3_EPILOGUE_SLAB(ek, g, acc[0][0], acc[0][1], acc[1][0], acc[1][1], parkf, m0 + wm * 128, n0 + wn * 64, lane, am)          // (the two halves written out: see gemm_f16x3.hip)
    X3_EPILOGUE_SLAB(ek, g, acc[2][0], acc[2][1], acc[3][0], acc[3][1], parkf, m0 + wm * 128 + 64, n0 + wn * 64, lane, am)
    amax_commit(g.amax_out, am); x3_publish_scale(g);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The PERSISTENT form for f16 outputs: the four products of a block in RLCF_PREC_F16.  With K = 768 a 256x256 tile is only 12 K tiles
// long: launched one workgroup per tile, a third of every tile's time went to the first loads (nothing to compute on), the LDS-parked
// epilogue and the write burst at the end of every tile round (one workgroup per tile, gemm_nt_f16_p8_kernel above: 4.21 ms for the
// layer's four products against 3.82 here, profiles/r5_notes.md section 1).  Here one workgroup per CU walks its XCD's tile range and
//   * the DMA ring never drains: the last two K tiles of a tile stage the first 1.75 K tiles of the NEXT tile (buffer_load ... lds
//     through a per-tile buffer descriptor: scalar base / soffset, one constant 32-bit lane offset per piece, rows beyond M / N read as
//     zeros by the descriptor's range check — no clamps, no per-piece address arithmetic), and B_lo of its K tile 1 goes out at the
//     START of the epilogue, ahead of the stores (PP_KTILE_FIRST);
//   * the epilogue uses no LDS and no barrier, and no transpose either: the MFMAs run with the operands SWAPPED (W fragment first), so an
//     accumulator tile is C^T — a lane owns a ROW, and with the two tiles of a wave interleaved at 4-column granularity a register quad
//     pair is 8 consecutive columns: alpha / bias / QuickGELU, f16 rounding, one 16-byte store;
//   * MODE 1 / 2 fold the LayerNorms of an image tower into the products (GemmX3Args::ln_*; engine.hip, transformer_forward):
//     MODE 1 normalises per row in the epilogue (A = the f16 residual stream itself, W carries gamma), MODE 2 adds into the residual
//     stream in place and leaves per-row partial (sum, sum of squares) for the next LayerNorm's statistics;
//   * the two wave groups line up for the epilogue (one extra barrier each per tile) so that both run it at the same time.
// What was measured and dropped on the way (start-time cohorts, deferred stores, an early touch of the residual tile): r5_notes.md.
//   * DEFER (round 6, MODE 0): half of a tile's output leaves under the NEXT tile's K loop instead of in the epilogue's burst.  The
//     epilogue converts the whole tile but stores only the upper 64 rows of every wave's 128 (8 of its 16 store instructions); the
//     lower 64 rows wait as 32 packed registers and go out ONE instruction per K tile, K tiles 2 .. 9 of the next tile, each placed
//     right in front of that K tile's counted wait, which is raised by one (vmcnt(7)): the store is YOUNGER than the six prefetch pieces it
//     is issued behind, so no load of the ring ever waits for it, and it has a whole K tile to retire before the next wait covers it.
//     Round 5 had tried the registers but issued all 8 stores under K tiles 0 and 1 — directly behind the burst of the other half,
//     where every wait of those K tiles then sat behind a congested store; spread over the K loop the chip sees them at 1/12 of the
//     burst rate.  MEASURED SLOWER (2-3 % on the layer): the K loop is bound by vector-memory issue, a store costs it what a DMA piece costs.
//     Off by default, RLCF_F16_PP_DEFER=1 turns it on (A/B, tests).  profiles/r6_notes.md section 1.
//   * TS (round 6, MODE 0): FULL-LINE stores.  The transpose-free epilogue above writes 32 rows x 32 B per store instruction: a lane owns
//     one row, so the 64 lanes of an instruction touch 32 different 128-byte lines — and the round-6 trace shows the epilogue paced at
//     ~80 ticks per store instruction and CU (12.5 B per tick) WHATEVER the other CUs do (start-time cohorts of 1/4 or 1/8 of the CUs:
//     the same 6 400 / 10 200 ticks): it is bound by line requests per CU, not by bytes and not by the chip-wide burst.  With TS a wave
//     moves each 32-row block of its tile through a private 4-KB LDS slab (ds_write_b128 lane = row, ds_read_b128 lane = (row of 8,
//     16-byte chunk), XOR-swizzled) and stores 8 rows x 128 B per instruction: a quarter of the line requests.  LDS: the 32 KB behind
//     the ring, 4 KB per wave; the wave's bias slice sits in the first KB of its own slab (read into registers before the slab is re-used).
// QuickGELU of two values with the non-transcendental steps as PACKED f32 instructions (v_pk_mul_f32 / v_pk_add_f32: two floats per
// lane and instruction at the full rate) — the same operations in the same order as quick_gelu_fast (bit-identical), 9 instead of 11
// instructions per pair: the epilogue of c_fc is bound by exactly this arithmetic (round-6 trace: ~1 900 ticks per 32-row block with or
// without full-line stores).  hipcc keeps `r * constant` and `1 + t` as two scalar VOP2 instructions each (literal operands).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (c2 / one2 reach this function as OPAQUE register pairs — the caller hides their values behind an empty asm — so that the vector
//  operations below become v_pk_mul_f32 / v_pk_add_f32; the instructions themselves are the compiler's, which also places the
//  wait states a transcendental result needs before a packed instruction may read it: hand-written asm here read stale registers.)
__device__ __forceinline__ f32x2 quick_gelu2_fast(f32x2 r, f32x2 c2, f32x2 one2) {
    f32x2 z = r * c2;
    z[0] = __builtin_amdgcn_exp2f(z[0]); z[1] = __builtin_amdgcn_exp2f(z[1]);
    f32x2 d = z + one2;
    d[0] = __builtin_amdgcn_rcpf(d[0]); d[1] = __builtin_amdgcn_rcpf(d[1]);
    return r * d;
}
template <int EPI, int MODE = 0, int DEFER = 0, int TRACE = 0, int TS = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_f16_pp_kernel(GemmX3Args g) {
#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass only needs the stub: the buffer-descriptor type below is a device-only type)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][P8_PAR] + 8 x 4 KB: one slab per wave (bias slice, TS transposes)
    const int tiles_n = (g.N + 255) / 256, tiles_m = (g.M + 255) / 256, ntiles = tiles_m * tiles_n;
    // XCD x owns a contiguous range of the tile order; its G/8 workgroups walk it G/8 tiles at a time
    const int wpx = gridDim.x >> 3, xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
    const int tq = ntiles / 8, tr = ntiles % 8;
    const int xstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, xend = xstart + tq + (xcd < tr ? 1 : 0);
    int lin = xstart + widx;
    if (lin >= xend) return;
    const int G = g.tile_group % 100 > 0 ? g.tile_group % 100 : 8;
    const bool nfast = g.tile_group >= 100 || (g.tile_group == 0 && tiles_n <= 4);
    auto tile_origin = [&](int bid, int& m0_, int& n0_) {
        const int per_group = G * tiles_n, grp = bid / per_group, first_m = grp * G;
        const int gsize = min(tiles_m - first_m, G), in_g = bid - grp * per_group;
        m0_ = (nfast ? first_m + in_g / tiles_n : first_m + in_g % gsize) * 256;
        n0_ = (nfast ? in_g % tiles_n : in_g / gsize) * 256;
    };
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 2, wn = wave & 3, l32 = lane & 31, h = lane >> 5;

    // DMA lane offsets (bytes from the tile's first row; the same for every tile and K tile)
    unsigned va[2][2], vw[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lr = wave * 16 + j * 8 + (lane >> 3), c = ((lane & 7) ^ ((lr >> 1) & 7)) * 16;
            va[hf][j] = (unsigned)(((lr >> 6) * 128 + hf * 64 + (lr & 63)) * g.lda) * 2u + c;
            // B half tiles: the two 32-column accumulator tiles of a wave interleave at 4-column granularity (tile j holds columns 8 q + 4 j
            // + {0..3} of the wave's 64), so that in the transposed accumulator (operands swapped) a lane owns 8 CONSECUTIVE columns of a row: 16-byte stores
            vw[hf][j] = (unsigned)(((lr >> 5) * 64 + ((lr & 31) >> 2) * 8 + hf * 4 + (lr & 3)) * g.ldw) * 2u + c;
        }
    auto rsrc_a = [&](int m0_) {
        const size_t bytes = (size_t)(g.M - m0_) * g.lda * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(g.Ahi + (size_t)m0_ * g.lda), 0, (int)(unsigned)(bytes > 0xfffff000u ? 0xfffff000u : bytes), 0x00020000);
    };
    auto rsrc_w = [&](int n0_) {
        const size_t bytes = (size_t)(g.N - n0_) * g.ldw * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(g.Whi + (size_t)n0_ * g.ldw), 0, (int)(unsigned)(bytes > 0xfffff000u ? 0xfffff000u : bytes), 0x00020000);
    };
    // The tile's 256 bias values travel through LDS (round 6): one LDS-DMA piece (64 lanes x 16 B; columns beyond N read as zeros by the
    // descriptor's range check) issued by EVERY wave (identical bytes: the vmcnt bookkeeping stays wave-uniform) at the head of K tile
    // nk - 2, i.e. older than that K tile's own pieces and covered by its counted wait; the epilogue reads them with ds_read_b128.  Before,
    // the epilogue's global loads of the bias made the compiler wait vmcnt(0) in front of the first output value — for the next tile's
    // whole prefetch including the B_lo pieces issued a few instructions earlier.
    const bool has_bias = g.bias != nullptr && (((uintptr_t)g.bias) & 3) == 0;
    auto rsrc_bias = [&](int n0_) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(has_bias ? g.bias + n0_ : (const float*)g.Whi), 0, has_bias ? (int)((g.N - n0_) * 4) : 0, 0x00020000);
    };
    const unsigned vbias = (unsigned)lane * 16u;
#define PP_STAGE_BIAS(rs) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + 2 * P8_PAR + wave_s * 4096), 16, vbias, 0, 0, 0);     // (every wave its own copy, in its own slab)
#define PP_STAGE_A(hf, par, rs, kt)                                                                                                 \
    {                                                                                                                               \
        char* d_ = smem + (par) * P8_PAR + (hf) * P8_HALF + wave_s * 2048;                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)d_, 16, va[hf][0], (kt) * 128, 0, 0);                                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(d_ + 1024), 16, va[hf][1], (kt) * 128, 0, 0);                           \
    }
#define PP_STAGE_B(hf, par, rs, kt)                                                                                                 \
    {                                                                                                                               \
        char* d_ = smem + (par) * P8_PAR + P8_B0 + (hf) * P8_HALF + wave_s * 2048;                                                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)d_, 16, vw[hf][0], (kt) * 128, 0, 0);                                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(d_ + 1024), 16, vw[hf][1], (kt) * 128, 0, 0);                           \
    }
    const int nk = g.K / 64;                     // even (launcher)
    // De-synchronised cohorts: all workgroups start together and every tile takes the same time, so the epilogues of all 256 CUs — 32 MB
    // of stores — would hit the fabric in the same ~4 us of every tile round, and the next tile's first loads queue behind them (s_memtime
    // trace: the K loop after a store burst runs 20 % slower per K tile than in the middle of a long tile).  Cohort c of P (by the
    // workgroup's index inside its XCD) therefore starts c / P of a tile late: the bursts of the cohorts interleave with the others' K loops.
    if (g.sk_blocks > 1 || g.sk_blocks < -1) {
        const int P = g.sk_blocks > 0 ? g.sk_blocks : -g.sk_blocks;
        const int c = g.sk_blocks > 0 ? widx % P : xcd % P;                 // (negative: whole XCDs as cohorts — their workgroups stay in step)
        const int cyc = c * (nk * 2800 + 6000) / P;              // ~ cycles per tile: 8 intervals of ~350 per K tile + epilogue
        for (int q = cyc >> 10; q > 0; --q) __builtin_amdgcn_s_sleep(16);            // (s_sleep counts 64-cycle units)
    }
    int m0, n0;
    tile_origin(lin, m0, n0);
    auto ra = rsrc_a(m0), rw = rsrc_w(n0);
    PP_STAGE_A(0, 0, ra, 0) PP_STAGE_B(0, 0, rw, 0) PP_STAGE_B(1, 0, rw, 0) PP_STAGE_A(1, 0, ra, 0)
    PP_STAGE_A(0, 1, ra, 1) PP_STAGE_B(1, 1, rw, 1) PP_STAGE_A(1, 1, ra, 1) PP_STAGE_B(0, 1, rw, 1)      // (B_lo of K tile 1 last: see PP_KTILE_FIRST)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    const int swz = (l32 >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + h) ^ swz) * 16;
    const int aoff = (wm * 64 + l32) * 128, boff = P8_B0 + (wn * 32 + l32) * 128;
    h16x8 a[2][4], b[4];
    f32x16 acc[4][2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
// first K tile of a tile: the first product of every accumulator takes C = 0 (no zero fill, and the accumulators are not live across the
// tile boundary: the epilogue's outputs use their registers)
#define PP_MMA0(ib, j)                                                                                                              \
    {                                                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                              \
        _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2)                                                                            \
            acc[(ib) * 2 + i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[0], a[i2][0], zero16, 0, 0, 0);                         \
        _Pragma("unroll") for (int ks = 1; ks < 4; ++ks)                                                                            \
            _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2)                                                                        \
                acc[(ib) * 2 + i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks], a[i2][ks], acc[(ib) * 2 + i2][j], 0, 0, 0);    \
        __builtin_amdgcn_s_setprio(0);                                                                                              \
    }
#define PP_MMA(ib, j)                                                                                                               \
    {                                                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                              \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                            \
            _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2)                                                                        \
                acc[(ib) * 2 + i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks], a[i2][ks], acc[(ib) * 2 + i2][j], 0, 0, 0);    \
        __builtin_amdgcn_s_setprio(0);                                                                                              \
    }
// one K tile of the steady state: stages B_lo of K tile kt+1 and the other three half tiles of kt+2, all of THIS tile.  MMA_ = PP_MMA, or
// PP_MMA0 for K tile 0; H_(n) = a hook in the read section of phase n (unused since the deferred-store experiment: r5_notes.md section 1)
#define PP_KTILE_X(kt, par, MMA_, H_)                                                                                               \
    {                                                                                                                               \
        P8_LDB(0, par) P8_LDA(0, par)                                                                                               \
        PP_STAGE_B(0, (par) ^ 1, rw, (kt) + 1)                                                                                      \
        H_(0)                                                                                                                       \
        P8_MID MMA_(0, 0) P8_END                                                                                                    \
        P8_LDB(1, par)                                                                                                              \
        PP_STAGE_A(0, par, ra, (kt) + 2)                                                                                            \
        H_(1)                                                                                                                       \
        P8_MID MMA_(0, 1) P8_END                                                                                                    \
        P8_LDA(1, par)                                                                                                              \
        PP_STAGE_B(1, par, rw, (kt) + 2)                                                                                            \
        H_(2)                                                                                                                       \
        P8_MID MMA_(1, 1) P8_END                                                                                                    \
        P8_LDB(0, par)                                                                                                              \
        PP_STAGE_A(1, par, ra, (kt) + 2)                                                                                            \
        PP_WAIT_##H_                                                                                                                \
        H_(3)                                                                                                                       \
        P8_MID MMA_(1, 0) P8_END                                                                                                    \
    }
// a steady-state K tile that also sends pending store IDX (0..7) of the PREVIOUS tile: row block 2 + (IDX >> 2), column group IDX & 3
#define PP_KTILE_ST(kt, par, IDX)                                                                                                   \
    {                                                                                                                               \
        P8_LDB(0, par) P8_LDA(0, par)                                                                                               \
        PP_STAGE_B(0, (par) ^ 1, rw, (kt) + 1)                                                                                      \
        P8_MID PP_MMA(0, 0) P8_END                                                                                                  \
        P8_LDB(1, par)                                                                                                              \
        PP_STAGE_A(0, par, ra, (kt) + 2)                                                                                            \
        P8_MID PP_MMA(0, 1) P8_END                                                                                                  \
        P8_LDA(1, par)                                                                                                              \
        PP_STAGE_B(1, par, rw, (kt) + 2)                                                                                            \
        P8_MID PP_MMA(1, 1) P8_END                                                                                                  \
        P8_LDB(0, par)                                                                                                              \
        PP_STAGE_A(1, par, ra, (kt) + 2)                                                                                            \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pend[IDX]), rs_pend, vout,                                 \
                                               ((2 + ((IDX) >> 2)) * 32 * g.ldch + ((IDX) & 3) * 16) * 2, 0);                        \
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                                                            \
        P8_MID PP_MMA(1, 0) P8_END                                                                                                  \
    }
#define PP_NOHOOK(n)
#define PP_WAIT_PP_NOHOOK asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#define PP_ST(v, p) { if (g.sk_epoch) __builtin_nontemporal_store(v, (h16x8*)(p)); else *(h16x8*)(p) = v; }      // (g.sk_epoch: RLCF_F16_PP_NT)
// K tile 0 of a tile (round 5).  Stores count in vmcnt like loads and retire in order, and the epilogue's 16 stores per wave take
// ~6 us to retire when every CU of the chip bursts its 128-KB tile at the same moment (without stores in_proj runs 818 instead of
// 1031 us, c_fc 1023 instead of 1363: profiles/r5_notes.md) — whatever load the NEXT wait covers that was issued AFTER them waits for
// them.  The first form staged B_lo of K tile 1 in phase 1 of K tile 0, i.e. behind the stores, and needed it one K tile later.  Now B_lo
// of K tile 1 is staged at the START of the epilogue (its buffer — parity 1 — was last read in phase 4 of the tile's last K tile, and both
// wave groups have lined up behind that phase), so the loads K tile 1 needs are all OLDER than the stores: K tile 0 stages only its
// three half tiles of K tile 2 and its wait leaves them AND the stores outstanding (vmcnt(6 + stores)); the stores have until the end
// of K tile 1, two K tiles instead of three quarters of one, to retire.  (Measured and dropped on the way: keeping half of a tile's
// output in registers and storing it one instruction per phase under K tiles 0 and 1 — slower, 1051 -> 1078 us on in_proj: every one
// of those K tiles' waits then sits behind a store.)
// nst (wave-uniform): 0 = no store of this wave may be outstanding (first tile of the workgroup; a tile at the matrix edge, whose
// epilogue drains), 1 = the 16 stores of a whole tile, 2 = the 20 of a MODE 2 tile (+ 4 partial-statistics stores)
#define PP_KTILE_FIRST(MMA_)                                                                                                        \
    {                                                                                                                               \
        P8_LDB(0, 0) P8_LDA(0, 0)                                                                                                   \
        P8_MID MMA_(0, 0) P8_END                                                                                                    \
        P8_LDB(1, 0)                                                                                                                \
        PP_STAGE_A(0, 0, ra, 2)                                                                                                     \
        P8_MID MMA_(0, 1) P8_END                                                                                                    \
        P8_LDA(1, 0)                                                                                                                \
        PP_STAGE_B(1, 0, rw, 2)                                                                                                     \
        P8_MID MMA_(1, 1) P8_END                                                                                                    \
        P8_LDB(0, 0)                                                                                                                \
        PP_STAGE_A(1, 0, ra, 2)                                                                                                     \
        if (nst == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                              \
        else if (nst == 1) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");                                                        \
        else if (nst == 3) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");                                                        \
        else asm volatile("s_waitcnt vmcnt(26)" ::: "memory");                                                                      \
        P8_MID MMA_(1, 0) P8_END                                                                                                    \
    }
#define PP_KTILE(kt, par) PP_KTILE_X(kt, par, PP_MMA, PP_NOHOOK)
// the last two K tiles of a tile: K tile nk-2 stages B_lo of nk-1 (this tile) and A_lo / B_hi / A_hi of the NEXT tile's K tile 0;
// K tile nk-1 stages the next tile's B_lo of K tile 0 and A_lo / B_hi / A_hi of its K tile 1 (nothing when this is the last tile)
#define PP_KTILE_TAIL(par, first)                                                                                                   \
    {                                                                                                                               \
        if (first) PP_STAGE_BIAS(rb)                                                                                                \
        P8_LDB(0, par) P8_LDA(0, par)                                                                                               \
        if (first) PP_STAGE_B(0, (par) ^ 1, rw, nk - 1)                                                                             \
        else if (have_next) PP_STAGE_B(0, (par) ^ 1, rwn, 0)                                                                        \
        P8_MID PP_MMA(0, 0) P8_END                                                                                                  \
        P8_LDB(1, par)                                                                                                              \
        if (have_next) PP_STAGE_A(0, par, ran, (first) ? 0 : 1)                                                                     \
        P8_MID PP_MMA(0, 1) P8_END                                                                                                  \
        P8_LDA(1, par)                                                                                                              \
        if (have_next) PP_STAGE_B(1, par, rwn, (first) ? 0 : 1)                                                                     \
        P8_MID PP_MMA(1, 1) P8_END                                                                                                  \
        P8_LDB(0, par)                                                                                                              \
        if (have_next) {                                                                                                            \
            PP_STAGE_A(1, par, ran, (first) ? 0 : 1)                                                                                \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                                        \
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                     \
        P8_MID PP_MMA(1, 0) P8_END                                                                                                  \
    }
    const float al = g.alpha;
    if (wm == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }       // the second group runs one barrier behind
    // (measurement: g.ws != null -> s_memtime stamps of waves 0 / 4 for the first 64 tiles of every workgroup: K loop start, K loop end,
    //  groups lined up, epilogue issued)
    unsigned long long* trace = TRACE ? (unsigned long long*)g.ws : nullptr;
    int tile_it = 0;
    int nst = 0;                                 // what the previous epilogue of this wave left outstanding (see PP_KTILE_FIRST)
    static_assert(MODE == 0 || MODE == 1 || MODE == 2, "epilogue mode");
    static_assert(DEFER == 0 || MODE == 0, "deferred stores: MODE 0 only");
    // DEFER: the previous tile's lower 64 rows of this wave (row blocks 2 / 3 x 4 column groups), packed, and their two row pointers
    h16x8 pend[DEFER ? 8 : 1];
    // whole tiles leave through BUFFER stores: a per-tile descriptor in scalar registers (base = the tile's first element, 256 rows of
    // range), ONE per-lane byte offset that is the same for every tile (row l32 of the wave's block, 8 columns at 8 h) and the row block /
    // column group as the scalar offset — no 64-bit address arithmetic per store, and a deferred store needs no address registers at all
    const unsigned vout = (unsigned)(((wm * 128 + l32) * g.ldch + wn * 64 + 8 * h) * 2);
    auto rsrc_out = [&](int m0_, int n0_) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(g.Chi + (size_t)m0_ * g.ldch + n0_), 0, (int)(unsigned)((size_t)256 * g.ldch * 2), 0x00020000);
    };
    auto rs_pend = rsrc_out(0, 0);
    bool have_pend = false;                      // (wave-uniform: set by the epilogue of a whole tile that has a successor)
    const bool can_defer = DEFER && nk >= 12 && g.ksplit == 0 && !g.sk_epoch && (size_t)256 * g.ldch * 2 < 0xfffff000u;
// (TRACE is its own instantiation: the stamp code costs the production kernel registers it needs for the deferred stores)
#define PP_STAMP(k) if constexpr (TRACE != 0) { if (trace && (wave & 3) == 0 && lane == 0 && tile_it < 64) trace[(((size_t)blockIdx.x * 64 + tile_it) * 2 + wm) * 16 + (k)] = __builtin_amdgcn_s_memtime(); }
    for (;;) {
        const int nlin = lin + wpx;
        const bool have_next = nlin < xend;
        PP_STAMP(0)
        int m0n = 0, n0n = 0;
        if (have_next) tile_origin(g.ksplit == 8 ? lin : nlin, m0n, n0n);          // (ksplit 8: measurement — every tile of a workgroup is its first)
        auto ran = rsrc_a(m0n), rwn = rsrc_w(n0n);
        auto rb = rsrc_bias(n0);
        PP_KTILE_FIRST(PP_MMA0)
        PP_KTILE(1, 1)
        PP_STAMP(4)
        int kt0 = 2;
        if constexpr (DEFER != 0) {
            if (have_pend) {                     // (nk >= 12: K tiles 2 .. 9 exist below the tail)
                PP_KTILE_ST(2, 0, 0) PP_KTILE_ST(3, 1, 1)
                PP_STAMP(5)
                PP_KTILE_ST(4, 0, 2) PP_KTILE_ST(5, 1, 3)
                PP_STAMP(6)
                PP_KTILE_ST(6, 0, 4) PP_KTILE_ST(7, 1, 5)
                PP_STAMP(7)
                PP_KTILE_ST(8, 0, 6) PP_KTILE_ST(9, 1, 7)
                PP_STAMP(8)
                kt0 = 10;
                have_pend = false;
            }
        }
        for (int kt = kt0; kt + 2 < nk; kt += 2) {
            PP_KTILE(kt, 0)
            PP_KTILE(kt + 1, 1)
            if (kt < 20) { PP_STAMP(4 + (kt >> 1)) }
        }
        PP_KTILE_TAIL(0, true)
        PP_KTILE_TAIL(1, false)
        // ---- epilogue of this tile, both groups at the same time
        PP_STAMP(1)
        if (wm == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        PP_STAMP(2)
        // B_lo of the next tile's K tile 1, BEFORE this tile's stores (see PP_KTILE_FIRST)
        if (have_next) PP_STAGE_B(0, 1, rwn, 1)
        PP_STAMP(10)
        if (g.ksplit != 2) {                                                // (ksplit 2: measurement without any epilogue)
            // The MFMAs ran with the operands swapped (W fragment first): an accumulator tile is C^T, so lane l32 owns ROW l32 of the 32-row
            // tile i and register r the tile column c = 8 (r >> 2) + 4 h + (r & 3), i.e. (B half tiles interleaved at 4-column granularity,
            // see vw) column 16 (r >> 2) + 8 h + 4 j + (r & 3) of the wave's 64: the two tiles j give a lane 8 CONSECUTIVE columns per register
            // quad -> one 16-byte store, with no transpose at all (the round-5 first form moved every value through two DPP exchanges and
            // two selects to get there: 6.5 VALU operations per output value with the matrix pipe idle, now 2.5).  A store instruction
            // writes 32 rows x 32 B; the four register quads of a row follow each other and complete its 128-byte line in the L2.
            const int ocol = n0 + wn * 64 + 8 * h;
            const int orow = m0 + wm * 128 + l32;
            const bool inside = m0 + 256 <= g.M && n0 + 256 <= g.N && (g.ldch & 7) == 0;       // (whole tile, 16-byte aligned rows: no masks)
            _Float16* obase = g.Chi + (size_t)orow * g.ldch + ocol;
            const bool defer_now = DEFER && can_defer && inside && have_next;
            const bool bufst = inside && (!g.sk_epoch || TS != 0) && (size_t)256 * g.ldch * 2 < 0xfffff000u;      // (RLCF_F16_PP_NT: the whole-line path carries its own nt form)
            auto rs_out = rsrc_out(m0, n0);
            if constexpr (DEFER != 0) {
                if (defer_now) rs_pend = rs_out;
            }
            float bj[4][8];
            if (has_bias) {              // from LDS (PP_STAGE_BIAS): landed behind the counted wait of K tile nk - 2; zeros beyond N
                // (inline asm: a C++ read of LDS here gets a compiler-inserted s_waitcnt vmcnt(0) in front of it — the pass orders LDS reads
                //  behind every LDS-DMA in flight, i.e. behind the next tile's prefetch.  This wave's own copy of the slice IS complete: it
                //  was issued ahead of K tile nk - 2's pieces, which that K tile's counted wait covered two K tiles ago.)
                //  The LDS address is formed INSIDE the statement: as a loop-invariant C++ value the compiler keeps it in a register across the K
                //  loop, runs out of registers in the DEFER build and spills it — and a scratch reload is a vector-memory load with its own vmcnt(0).
                const unsigned bias_base = (unsigned)(uintptr_t)(lptr_t)(smem + 2 * P8_PAR + wave_s * 4096), bias_col = (unsigned)(wn * 64 + 8 * h);
                unsigned bl;
                f32x4 t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("v_lshl_add_u32 %8, %9, 2, %10\n\t"
                             "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:80\n\t"
                             "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:144\n\tds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:208\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(bl)
                             : "v"(bias_col), "s"(bias_base) : "memory");
                const f32x4 tq[8] = {t0, t1, t2, t3, t4, t5, t6, t7};
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bj[gq][e] = tq[gq * 2 + (e >> 2)][e & 3];
            } else {                     // (no bias; a float pointer is always dword-aligned, so has_bias == (bias != null): no global load here —
                                         //  one in EITHER arm would make the compiler wait vmcnt(0) at the join)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bj[gq][e] = 0.f;
            }
            if constexpr (MODE == 2) {
                // residual rows: every load is issued before the first store (stores count in vmcnt: a load behind one would wait for it)
                h16x8 xr[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16* rp = g.Chi + (size_t)min(orow + i * 32, g.M - 1) * g.ldch + ocol;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) xr[i][gq] = *(const h16x8*)(rp + gq * 16);
                }
                const int part = (n0 >> 8) * 4 + wn;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = orow + i * 32;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        h16x8 o;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                o[j * 4 + q] = (_Float16)((float)xr[i][gq][j * 4 + q] + (al * acc[i][j][gq * 4 + q] + bj[gq][j * 4 + q]));
                        // the statistics describe the STORED f16 row (what the next product reads): v_dot2_f32_f16 adds two products of f16
                        // values (exact in f32) per instruction — sum of squares with the pair itself, sum with (1, 1)
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            const h16x2 v2 = {o[2 * e2], o[2 * e2 + 1]};
                            const h16x2 one2 = {(_Float16)1.0f, (_Float16)1.0f};
                            s1 = __builtin_amdgcn_fdot2(v2, one2, s1, false);
                            s2 = __builtin_amdgcn_fdot2(v2, v2, s2, false);
                        }
                        if (row < g.M) *(h16x8*)(g.Chi + (size_t)row * g.ldch + ocol + gq * 16) = o;
                    }
                    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);        // the other half-wave holds the row's other 32 columns
                    if (h == 0 && row < g.M) *(float2*)(g.ln_part + ((size_t)part * g.M + row) * 2) = make_float2(s1, s2);
                }
            } else {
            float sj[4][8];
            float rA[4], rB[4];
            if constexpr (MODE == 1) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 s0 = *(const float4*)(g.ln_s + ocol + gq * 16), s1 = *(const float4*)(g.ln_s + ocol + gq * 16 + 4);
                    sj[gq][0] = s0.x; sj[gq][1] = s0.y; sj[gq][2] = s0.z; sj[gq][3] = s0.w;
                    sj[gq][4] = s1.x; sj[gq][5] = s1.y; sj[gq][6] = s1.z; sj[gq][7] = s1.w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2 mr = *(const float2*)(g.ln_mr + (size_t)min(orow + i * 32, g.M - 1) * 2);
                    rA[i] = al * mr.y; rB[i] = -mr.y * mr.x;          // rstd (alpha acc - mu s) + b' = (alpha rstd) acc + ((-rstd mu) s + b')
                }
            }
            // TS: lane constants of the wave's private slab.  write: row l32, 16-byte chunk (2 gq + h) XOR (l32 & 7); read: row 8 k + (lane >> 3),
            // chunk (lane & 7) XOR (lane >> 3); store: the same row and chunk of the output tile
            const unsigned ts_slab = (unsigned)(uintptr_t)(lptr_t)(smem + 2 * P8_PAR + wave_s * 4096);
            const unsigned ts_w0 = ts_slab + (unsigned)l32 * 128u + (unsigned)((h ^ (l32 & 7)) << 4);      // (chunk 2 gq + h: XOR with gq << 5 below)
            const unsigned ts_r0 = ts_slab + (unsigned)(lane >> 3) * 128u + (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
            const unsigned ts_v0 = (unsigned)(((wm * 128 + (lane >> 3)) * g.ldch + wn * 64 + (lane & 7) * 8) * 2);
            const bool ts_now = TS && MODE != 2 && bufst;
            f32x2 gelu_c2 = {-1.702f * 1.44269504088896341f, -1.702f * 1.44269504088896341f}, gelu_one2 = {1.0f, 1.0f};
            asm volatile("" : "+v"(gelu_c2), "+v"(gelu_one2));          // (opaque: see quick_gelu2_fast)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h16x8 oq[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    h16x8 o;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            f32x2 r;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                if constexpr (MODE == 1) r[e] = rA[i] * acc[i][j][gq * 4 + q + e] + (rB[i] * sj[gq][j * 4 + q + e] + bj[gq][j * 4 + q + e]);
                                else r[e] = al * acc[i][j][gq * 4 + q + e] + bj[gq][j * 4 + q + e];
                            }
                            if constexpr (EPI == RLCF_EPI_QUICKGELU) r = quick_gelu2_fast(r, gelu_c2, gelu_one2);
                            o[j * 4 + q] = (_Float16)r[0];
                            o[j * 4 + q + 1] = (_Float16)r[1];
                        }
                    oq[gq] = o;
                    if constexpr (TS != 0) { if (ts_now) continue; }
                    _Float16* op = obase + (size_t)(i * 32) * g.ldch + gq * 16;
                    if (g.ksplit == 1 && o[0] != (_Float16)123.0f) continue;            // (measurement: no stores)
                    if constexpr (DEFER != 0) {
                        if (i >= 2 && defer_now) { pend[(i - 2) * 4 + gq] = o; continue; }
                    }
                    if (bufst) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, vout, (i * 32 * g.ldch + gq * 16) * 2, 0);
                    } else if (inside) {
                        PP_ST(o, op)
                    } else if (orow + i * 32 < g.M) {
                        const int c0 = ocol + gq * 16;
                        if (c0 + 8 <= g.N && (g.ldch & 7) == 0) *(h16x8*)op = o;
                        else {
                            if (c0 + 4 <= g.N) *(h16x4*)op = h16x4{o[0], o[1], o[2], o[3]};
                            if (c0 + 8 <= g.N) *(h16x4*)(op + 4) = h16x4{o[4], o[5], o[6], o[7]};
                        }
                    }
                    if (gq == 0 && i == 0) { PP_STAMP(11) }       // (trace: first store issued; then after 4 / 8 / 12 of the wave's 16)
                    if (gq == 3 && i == 0) { PP_STAMP(12) }
                    if (gq == 3 && i == 1) { PP_STAMP(13) }
                    if (gq == 3 && i == 2) { PP_STAMP(14) }
                }
                if constexpr (TS != 0) {
                    if (ts_now) {
                        // the 32-row block through the slab (inline asm: a C++ access of LDS gets a compiler vmcnt(0) for the LDS-DMA in flight);
                        // the slab is this wave's alone: no barrier, its own lgkmcnt orders write -> read -> next block's write
                        u32x4 t0, t1, t2, t3;
                        asm volatile("ds_write_b128 %4, %8\n\tds_write_b128 %5, %9\n\tds_write_b128 %6, %10\n\tds_write_b128 %7, %11\n\t"
                                     "s_waitcnt lgkmcnt(0)\n\t"
                                     "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:1024\n\tds_read_b128 %2, %12 offset:2048\n\tds_read_b128 %3, %12 offset:3072\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                                     : "v"(ts_w0), "v"(ts_w0 ^ 32u), "v"(ts_w0 ^ 64u), "v"(ts_w0 ^ 96u),
                                       "v"(__builtin_bit_cast(u32x4, oq[0])), "v"(__builtin_bit_cast(u32x4, oq[1])), "v"(__builtin_bit_cast(u32x4, oq[2])),
                                       "v"(__builtin_bit_cast(u32x4, oq[3])), "v"(ts_r0)
                                     : "memory");
                        const int so = (i * 32) * g.ldch * 2;
                        if (g.sk_epoch) {                                  // RLCF_F16_PP_NT: the cache-policy operand is an immediate (2 = nt)
                            __builtin_amdgcn_raw_buffer_store_b128(t0, rs_out, ts_v0, so, 2);
                            __builtin_amdgcn_raw_buffer_store_b128(t1, rs_out, ts_v0, so + 8 * g.ldch * 2, 2);
                            __builtin_amdgcn_raw_buffer_store_b128(t2, rs_out, ts_v0, so + 16 * g.ldch * 2, 2);
                            __builtin_amdgcn_raw_buffer_store_b128(t3, rs_out, ts_v0, so + 24 * g.ldch * 2, 2);
                        } else {
                        __builtin_amdgcn_raw_buffer_store_b128(t0, rs_out, ts_v0, so, 0);
                        if (i == 0) { PP_STAMP(11) }
                        __builtin_amdgcn_raw_buffer_store_b128(t1, rs_out, ts_v0, so + 8 * g.ldch * 2, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(t2, rs_out, ts_v0, so + 16 * g.ldch * 2, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(t3, rs_out, ts_v0, so + 24 * g.ldch * 2, 0);
                        }
                        if (i == 0) { PP_STAMP(12) }
                        if (i == 1) { PP_STAMP(13) }
                        if (i == 2) { PP_STAMP(14) }
                    }
                }
            }
            }
        }
        PP_STAMP(3)
        ++tile_it;
        if (!have_next) break;
        {
            // what this wave's epilogue left in flight behind the early B_lo: the 16 (MODE 2: 20) stores of a whole tile, or nothing —
            // a tile at the matrix edge issues a data-dependent number of masked stores, so it drains instead
            const bool whole = g.ksplit == 0 && m0 + 256 <= g.M && n0 + 256 <= g.N && (g.ldch & 7) == 0;
            if (g.ksplit == 0 && !whole) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            nst = whole ? (MODE == 2 ? 2 : 1) : 0;
            if constexpr (DEFER != 0) {
                if (whole && can_defer) { nst = 3; have_pend = true; }            // (have_next holds here: 8 stores went out, 8 wait in pend)
            }
        }
        if (wm == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }   // back to one barrier behind
        lin = nlin; m0 = m0n; n0 = n0n; ra = ran; rw = rwn;
    }
#endif
}

// which launches take this kernel: the compile-time epilogues (f32 out; f32 out + residual; QuickGELU -> f16; f16 only) — the four
// products of a ViT layer in the single-pass forward pipeline.  RLCF_F16_P8=0 switches it off (the K/2 alias kernels: A/B measurements)
bool gemm_f16_p8_ok(const void* C, const void* Chi, const float* residual, const float* aux, int epilogue, const float* alpha_dev,
                    unsigned int* amax_out, const float* out_scale_dev, int N, int K, int lda, int ldw, int ldc, int ldr, int ldch) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("RLCF_F16_P8"); on = e ? atoi(e) : 1; }
    if (!on || aux || alpha_dev || amax_out || out_scale_dev) return false;
    if (K % 64 || N % 4 || lda % 8 || ldw % 8 || ldc % 4 || ldr % 4 || ldch % 4) return false;
    const bool f32o = C != nullptr, f16o = Chi != nullptr, res = residual != nullptr;
    if (epilogue == RLCF_EPI_NONE && f32o && !f16o) return true;                 // kinds 1 / 2
    if (epilogue == RLCF_EPI_NONE && !f32o && f16o && !res) return true;         // kind 4
    if (epilogue == RLCF_EPI_QUICKGELU && !f32o && f16o && !res) return true;    // kind 3
    return false;
}

// RLCF_F16_PP=0: the one-workgroup-per-tile kernel for f16 outputs too (A/B measurements)
static int f16_pp_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("RLCF_F16_PP"); on = e ? atoi(e) : 1; }
    return on;
}

// RLCF_F16_PP_NT=1: non-temporal output stores.  The first (transposing) epilogue wrote whole 128-byte lines per instruction and gained ~2 %
// from them; the transpose-free epilogue writes 32 rows x 32 B per instruction, and a non-temporal store of a PART of a line is a partial
// write at the memory (tools/probes/store_rate.hip: 32x32-B pieces, 256 CUs: 5.6 TB/s with the default policy, 0.97 TB/s non-temporal)
// Round 6: the whole-line epilogue (TS) is back, and with it the gain: non-temporal WHOLE-line stores take 2 % off in_proj and 1.5 - 5 % off c_fc
// at op level, +0.4 - 0.55 % on the f16 mode's step (profiles/r6_f16_gemm_nt_whole_lines_ab.txt) — so the default is "auto": non-temporal
// exactly where the tile leaves as whole lines, the default policy everywhere else.  =0: never; =1: everywhere (the round-5 measurement form).
static int pp_nt_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("RLCF_F16_PP_NT"); on = e ? atoi(e) : 2; }
    return on;
}
static unsigned pp_nt_for(bool whole_lines) { const int m = pp_nt_enabled(); return (m == 1 || (m == 2 && whole_lines)) ? 1u : 0u; }
// LayerNorm-folded products of the single-pass f16 image towers (GemmX3Args::ln_*): always the persistent kernel.
//   mode 1: out[M, N] (f16) = epi(rstd_r (alpha A.W'^T - mu_r s) + bias'),  A = the f16 residual stream, ln_mr [M][2], ln_s [N]
//   mode 2: x[M, N] (f16, in place) += alpha A.W^T + bias;  ln_part [(N / 256) * 4][M][2] receives the partial row statistics
int launch_gemm_f16_pp_ln(const void* A, int lda, const void* W, int ldw, const float* bias, void* out16, int ldo, int M, int N, int K, float alpha,
                          int epilogue, int mode, const float* ln_mr, const float* ln_s, float* ln_part, hipStream_t st) {
    RLCF_ARG_CHECK(M > 0 && A && W && out16 && (mode == 1 || mode == 2) && K % 128 == 0 && K >= 256 && N % 256 == 0 && ldo % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0);
    RLCF_ARG_CHECK(mode == 1 ? (ln_mr && ln_s && (epilogue == RLCF_EPI_NONE || epilogue == RLCF_EPI_QUICKGELU)) : (ln_part && epilogue == RLCF_EPI_NONE));
    RLCF_ARG_CHECK((size_t)256 * lda * 2 < 0x7fffffffu && (size_t)256 * ldw * 2 < 0x7fffffffu && (((uintptr_t)ln_s) & 15) == 0 && (((uintptr_t)bias) & 15) == 0);
    GemmX3Args g{};
    g.Ahi = (const _Float16*)A; g.lda = lda; g.Whi = (const _Float16*)W; g.ldw = ldw; g.bias = bias; g.Chi = (_Float16*)out16; g.ldch = ldo;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.epilogue = epilogue; g.kstep = 64; g.ksplit = 0; g.tile_group = 0; g.sk_blocks = 1;
    g.ln_mr = ln_mr; g.ln_s = ln_s; g.ln_part = ln_part;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const int blocks = ((M + 255) / 256) * (N / 256);
    const int grid = std::min((ncu / 8) * 8, ((blocks + 7) / 8) * 8);
    const size_t shp = (size_t)2 * P8_PAR + 8 * 4096;      // + a 4-KB slab per wave (bias slice; TS: the transposing epilogue)
    const char* tse = getenv("RLCF_F16_PP_TSTORE");          // (mode 1: full-line stores through the per-wave LDS slab, as the plain products)
    const bool tstore = (tse ? atoi(tse) : 1) != 0;
    g.sk_epoch = pp_nt_for(tstore && mode == 1);
#define PP_LN_GO(E, MD, S)                                                                                                          \
    {                                                                                                                               \
        int rc = rlcf_func_lds((const void*)gemm_nt_f16_pp_kernel<E, MD, 0, 0, S>, shp);                                             \
        if (rc != RLCF_OK) return rc;                                                                                               \
        gemm_nt_f16_pp_kernel<E, MD, 0, 0, S><<<dim3(grid), dim3(512), shp, st>>>(g);                                                \
    }
    if (mode == 2) PP_LN_GO(RLCF_EPI_NONE, 2, 0)
    else if (epilogue == RLCF_EPI_QUICKGELU) { if (tstore) PP_LN_GO(RLCF_EPI_QUICKGELU, 1, 1) else PP_LN_GO(RLCF_EPI_QUICKGELU, 1, 0) }
    else { if (tstore) PP_LN_GO(RLCF_EPI_NONE, 1, 1) else PP_LN_GO(RLCF_EPI_NONE, 1, 0) }
#undef PP_LN_GO
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}

int launch_gemm_f16_p8(const void* A, int lda, const void* W, int ldw, const float* bias, const float* residual, int ldr, float* C, int ldc,
                       void* Cf16, int ldch, int M, int N, int K, float alpha, int epilogue, int tile_group, hipStream_t st) {
    RLCF_ARG_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0 && A && W && (C || Cf16));
    GemmX3Args g{};
    g.Ahi = (const _Float16*)A; g.Alo = nullptr; g.lda = lda; g.Whi = (const _Float16*)W; g.Wlo = nullptr; g.ldw = ldw;
    g.bias = bias; g.residual = residual; g.ldr = ldr; g.C = C; g.ldc = ldc; g.Chi = (_Float16*)Cf16; g.Clo = nullptr; g.ldch = ldch;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.epilogue = epilogue; g.kstep = 64; g.ksplit = 1; g.tile_group = tile_group;
    static int abl = -1;                                     // RLCF_F16_PP_ABL: measurement ablations of the persistent kernel (wrong results)
    if (abl < 0) { const char* e = getenv("RLCF_F16_PP_ABL"); abl = e ? atoi(e) : 0; }
    const int blocks = ((M + 255) / 256) * ((N + 255) / 256);
    if (f16_pp_enabled() && !C && !residual && Cf16 && K % 128 == 0 && K >= 256 && ldch % 4 == 0 && (size_t)256 * lda * 2 < 0x7fffffffu && (size_t)256 * ldw * 2 < 0x7fffffffu) {
        // persistent form: one workgroup per CU (a multiple of 8: the XCDs take equal shares of the workgroups)
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            hipDeviceProp_t pr;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
            if (ncu <= 0) ncu = 256;
        }
        const int grid = std::min((ncu / 8) * 8, ((blocks + 7) / 8) * 8);
        const size_t shp = (size_t)2 * P8_PAR + 8 * 4096;      // + a 4-KB slab per wave (bias slice; TS: the transposing epilogue)
        g.ksplit = abl;
        static int desync = -1;                              // RLCF_F16_PP_DESYNC=P: start-time cohorts (1 = all together)
        if (desync < 0) { const char* e = getenv("RLCF_F16_PP_DESYNC"); desync = e ? atoi(e) : 1; }
        g.sk_blocks = blocks > grid ? desync : 1;            // (one tile per workgroup: nothing to interleave)
        // RLCF_F16_PP_TRACE=1 (measurement): stamp buffer, dumped after every launch (synchronises: not for timing runs)
        static int trace_on = -1;
        static unsigned long long* trace_buf = nullptr;
        if (trace_on < 0) { const char* e = getenv("RLCF_F16_PP_TRACE"); trace_on = e ? atoi(e) : 0; }
        const size_t trace_n = (size_t)256 * 64 * 2 * 16;
        if (trace_on) {
            if (!trace_buf) RLCF_HIP_CHECK(hipMalloc((void**)&trace_buf, trace_n * 8));
            RLCF_HIP_CHECK(hipMemsetAsync(trace_buf, 0, trace_n * 8, st));
            g.ws = (float*)trace_buf;
        }
        // RLCF_F16_PP_DEFER=1 (read per launch: A/B inside one process): half of every tile's stores ride under the next tile's K loop.
        // OFF by default: bit-identical, but 2-3 % SLOWER on the layer's four products (in_proj +-0, out_proj -9 %, c_fc -3.5 %) — the K
        // loop has no spare vector-memory issue capacity, a deferred store costs it what a DMA piece costs (profiles/r6_notes.md section 1)
        const char* de = getenv("RLCF_F16_PP_DEFER");
        const bool defer = (de ? atoi(de) : 0) != 0 && abl == 0 && blocks > grid && K >= 768;
        // RLCF_F16_PP_TSTORE (read per launch): 1 = full-line stores through the per-wave LDS slab (TS), 0 = 32 rows x 32 B per instruction
        const char* tse = getenv("RLCF_F16_PP_TSTORE");
        const bool tstore = (tse ? atoi(tse) : 1) != 0 && !defer && abl == 0;
        g.sk_epoch = pp_nt_for(tstore);
#define PP_GO(E, D, T, S)                                                                                                           \
    {                                                                                                                               \
        int rc = rlcf_func_lds((const void*)gemm_nt_f16_pp_kernel<E, 0, D, T, S>, shp);                                              \
        if (rc != RLCF_OK) return rc;                                                                                               \
        gemm_nt_f16_pp_kernel<E, 0, D, T, S><<<dim3(grid), dim3(512), shp, st>>>(g);                                                 \
    }
#define PP_GO_T(E, D) { if (tstore) { if (trace_on) PP_GO(E, 0, 1, 1) else PP_GO(E, 0, 0, 1) } else if (trace_on) PP_GO(E, D, 1, 0) else PP_GO(E, D, 0, 0) }
        if (epilogue == RLCF_EPI_QUICKGELU) { if (defer) PP_GO_T(RLCF_EPI_QUICKGELU, 1) else PP_GO_T(RLCF_EPI_QUICKGELU, 0) }
        else { if (defer) PP_GO_T(RLCF_EPI_NONE, 1) else PP_GO_T(RLCF_EPI_NONE, 0) }
#undef PP_GO_T
#undef PP_GO
        RLCF_LAUNCH_CHECK();
        if (trace_on) {
            static unsigned long long* host = nullptr;
            if (!host) host = (unsigned long long*)malloc(trace_n * 8);
            RLCF_HIP_CHECK(hipStreamSynchronize(st));
            RLCF_HIP_CHECK(hipMemcpy(host, trace_buf, trace_n * 8, hipMemcpyDeviceToHost));
            // per group: mean cycles of K loop / line-up wait / epilogue / tile-to-tile, over the tiles 1.. of a few workgroups
            for (int wgi : {0, 1, 8, 100, 255}) {
                if (wgi >= grid) continue;
                for (int grp = 0; grp < 2; ++grp) {
                    double kl = 0, lu = 0, ep = 0, tt = 0, pairs[10] = {0}; int n = 0;
                    for (int ti = 1; ti < 64; ++ti) {
                        const unsigned long long* r = host + (((size_t)wgi * 64 + ti) * 2 + grp) * 16;
                        const unsigned long long* pr = r - 32;
                        if (!r[3] || !pr[0]) break;
                        kl += (double)(r[1] - r[0]); lu += (double)(r[2] - r[1]); ep += (double)(r[3] - r[2]); tt += (double)(r[0] - pr[0]); ++n;
                        unsigned long long prev = r[0];
                        for (int q = 0; q < 10 && 2 * q + 2 < K / 64; ++q) { pairs[q] += (double)(r[4 + q] - prev); prev = r[4 + q]; }
                    }
                    if (n) {
                        fprintf(stderr, "[pp trace]   K-tile pairs:");
                        for (int q = 0; q < 10 && 2 * q + 2 < K / 64; ++q) fprintf(stderr, " %.0f", pairs[q] / n);
                        fprintf(stderr, "\n");
                    }
                    if (n) fprintf(stderr, "[pp trace] M=%d N=%d K=%d wg %3d group %d: tiles %2d  K loop %.0f  line-up %.0f  epilogue %.0f  tile-to-tile %.0f (s_memtime ticks)\n",
                                   M, N, K, wgi, grp, n, kl / n, lu / n, ep / n, tt / n);
                    if (n && K / 64 <= 12) {     // inside the epilogue (slots 10 .. 14 are free at K = 768): ticks since the groups lined up
                        double e[5] = {0, 0, 0, 0, 0}; int m = 0;
                        for (int ti = 1; ti < 64; ++ti) {
                            const unsigned long long* r = host + (((size_t)wgi * 64 + ti) * 2 + grp) * 16;
                            if (!r[3] || !r[14]) break;
                            for (int q = 0; q < 5; ++q) e[q] += (double)(r[10 + q] - r[2]);
                            ++m;
                        }
                        if (m) fprintf(stderr, "[pp trace]   epilogue, since line-up: next-tile B_lo staged %.0f | first store issued %.0f | 4 stores %.0f | 8 stores %.0f | 12 stores %.0f | 16 stores %.0f\n",
                                       e[0] / m, e[1] / m, e[2] / m, e[3] / m, e[4] / m, ep / n);
                    }
                }
            }
        }
        return RLCF_OK;
    }
    const size_t sh = (size_t)8 * 64 * 68 * sizeof(float);               // 139 264 B: the parking space of the epilogue > 2 K tiles (131 072 B)
    static int prio = -1;
    if (prio < 0) { const char* e = getenv("RLCF_F16_P8_PRIO"); prio = e ? atoi(e) : 1; }
    if (prio) {
        int rc = rlcf_func_lds((const void*)gemm_nt_f16_p8_kernel<true>, sh);
        if (rc != RLCF_OK) return rc;
        gemm_nt_f16_p8_kernel<true><<<dim3(blocks), dim3(512), sh, st>>>(g);
    } else {
        int rc = rlcf_func_lds((const void*)gemm_nt_f16_p8_kernel<false>, sh);
        if (rc != RLCF_OK) return rc;
        gemm_nt_f16_p8_kernel<false><<<dim3(blocks), dim3(512), sh, st>>>(g);
    }
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
