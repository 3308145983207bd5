// C = epi(alpha * A . W^T + bias) (+ residual), f32 storage, f32-input MFMA
// (v_mfma_f32_32x32x2_f32: bit-exact f32 FMA chains at the f32 vector rate, 157 TF peak).
// This is the parity-mode GEMM behind every Linear of the CLIP towers
// (reference call sites: TPT/clip/model.py:175-191,235-238; custom_clip.py:71,332-333).
//
// Tiling: BMxBN block tile, 4 waves as 2x2, each wave (BM/2)x(BN/2) in 32x32 MFMA tiles,
// BK = 16, double-buffered LDS held k-major so that an MFMA operand read (32 consecutive
// rows of one k) is one conflict-free ds_read_b32 per half-wave.
#include "common.h"

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(GemmArgs g) {
    constexpr int BK = 16;
    constexpr int TM = BM / 64, TN = BN / 64;           // 32x32 tiles per wave
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int NA = BM * 4 / 256, NB = BN * 4 / 256;  // float4 loads per thread per tile
    __shared__ float lds[2 * BK * LDA + 2 * BK * LDB];
    float* As = lds;
    float* Bs = lds + 2 * BK * LDA;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, h = lane >> 5;
    const float* __restrict__ A = (const float*)g.A;
    const float* __restrict__ W = (const float*)g.W;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* pa[NA];
    const float* pb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int idx = t + i * 256, row = idx >> 2, kv = idx & 3;
        int gr = min(m0 + row, g.M - 1);
        pa[i] = A + (size_t)gr * g.lda + kv * 4;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int idx = t + i * 256, row = idx >> 2, kv = idx & 3;
        int gr = min(n0 + row, g.N - 1);
        pb[i] = W + (size_t)gr * g.ldw + kv * 4;
    }
    float4 ra[NA], rb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *(const float4*)(pa[i] + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *(const float4*)(pb[i] + k0);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int idx = t + i * 256, row = idx >> 2, kv = idx & 3;
            float* d = As + buf * BK * LDA + (kv * 4) * LDA + row;
            d[0] = ra[i].x; d[LDA] = ra[i].y; d[2 * LDA] = ra[i].z; d[3 * LDA] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int idx = t + i * 256, row = idx >> 2, kv = idx & 3;
            float* d = Bs + buf * BK * LDB + (kv * 4) * LDB + row;
            d[0] = rb[i].x; d[LDB] = rb[i].y; d[2 * LDB] = rb[i].z; d[3 * LDB] = rb[i].w;
        }
    };

    const int nk = g.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const float* as = As + cur * BK * LDA + wm * (BM / 2) + l32;
        const float* bs = Bs + cur * BK * LDB + wn * (BN / 2) + l32;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = s * 2 + h;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[kk * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[kk * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    float* __restrict__ C = (float*)g.C;
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + l32;
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + mfma32_row(r, h);
                if (row >= g.M) continue;
                float v = g.alpha * acc[i][j][r] + bv;
                if (g.epilogue == RLCF_EPI_QUICKGELU) v = quick_gelu(v);
                else if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) v *= quick_gelu_grad(g.aux[(size_t)row * g.ldaux + col]);
                if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
                if (g.epilogue == RLCF_EPI_RELU) v = fmaxf(v, 0.f);
                if (g.amax_out) am = fmaxf(am, fabsf(v));
                C[(size_t)row * g.ldc + col] = v;
            }
        }
    amax_commit(g.amax_out, am);
}

// Small-M variant (sparse backward, reward views): one 32x32 output tile per 4-wave block, the
// waves split K between them (64-wide chunks, operands straight from global memory into the
// MFMA registers: lane (row, half) owns 32 consecutive k of its row) and the four partial tiles
// are summed through LDS in a fixed order (deterministic).  Latency of a K=2048 GEMM: ~7 us
// instead of ~110 us for the 64x64-per-wave tiling, and M=239 fills 128+ CUs instead of 32.
// NWV waves split K; the operand loads of a wave's next chunk are in flight while the MFMAs of the current one run (the kernel is
// latency-bound: 16 float4 loads per lane and chunk, then 32 MFMAs).  Eight waves for K >= 512 (one test image's sparse text passes
// are ~100 such launches: 20 -> ~8 us each).
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void gemm_nt_f32_splitk_kernel(GemmArgs g) {
    __shared__ float part[NWV][32 * 33];
    const int tiles_n = (g.N + 31) / 32;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l32 = lane & 31, h = lane >> 5;
    const float* __restrict__ ap = (const float*)g.A + (size_t)min(m0 + l32, g.M - 1) * g.lda + h * 32;
    const float* __restrict__ wp = (const float*)g.W + (size_t)min(n0 + l32, g.N - 1) * g.ldw + h * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = g.K / 64;
    float4 a[8], b[8], a2[8], b2[8];
    if (wave < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = *(const float4*)(ap + wave * 64 + j * 4); b[j] = *(const float4*)(wp + wave * 64 + j * 4); }
    }
    for (int c = wave; c < nchunk; c += NWV) {
        const bool more = c + NWV < nchunk;
        if (more) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a2[j] = *(const float4*)(ap + (c + NWV) * 64 + j * 4); b2[j] = *(const float4*)(wp + (c + NWV) * 64 + j * 4); }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b[j].w, acc, 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = a2[j]; b[j] = b2[j]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][mfma32_row(r, h) * 33 + l32] = acc[r];
    __syncthreads();
    float* __restrict__ C = (float*)g.C;
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < 1024 / (64 * NWV); ++e) {
        const int idx = t + e * 64 * NWV, rr = idx >> 5, cc = idx & 31;
        const int row = m0 + rr, col = n0 + cc;
        if (row >= g.M || col >= g.N) continue;
        float v = part[0][rr * 33 + cc];
#pragma unroll
        for (int w = 1; w < NWV; ++w) v += part[w][rr * 33 + cc];
        v = g.alpha * v + (g.bias ? g.bias[col] : 0.f);
        if (g.epilogue == RLCF_EPI_QUICKGELU) v = quick_gelu(v);
        else if (g.epilogue == RLCF_EPI_QUICKGELU_BWD) v *= quick_gelu_grad(g.aux[(size_t)row * g.ldaux + col]);
        if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
        if (g.epilogue == RLCF_EPI_RELU) v = fmaxf(v, 0.f);
        if (g.amax_out) am = fmaxf(am, fabsf(v));
        C[(size_t)row * g.ldc + col] = v;
    }
    amax_commit(g.amax_out, am);
}

int launch_gemm_f32(const GemmArgs& g, hipStream_t st) {
    RLCF_ARG_CHECK(g.M > 0 && g.N > 0 && g.K > 0 && g.K % 16 == 0);
    RLCF_ARG_CHECK(g.lda % 4 == 0 && g.ldw % 4 == 0);
    RLCF_ARG_CHECK(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.W & 15) == 0);
    const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    static int small_mode = -1;              // RLCF_F32_SMALL=1: small-M products on the LDS-tiled 64x64 kernel (A/B measurements)
    if (small_mode < 0) { const char* e = getenv("RLCF_F32_SMALL"); small_mode = e ? atoi(e) : 0; }
    if (g.M <= 512 && g.K % 64 == 0 && !(small_mode == 1 && g.M > 64)) {
        const long nb = (long)((g.M + 31) / 32) * ((g.N + 31) / 32);
        if (g.K >= 512) gemm_nt_f32_splitk_kernel<8><<<dim3((unsigned)nb), dim3(512), 0, st>>>(g);
        else gemm_nt_f32_splitk_kernel<4><<<dim3((unsigned)nb), dim3(256), 0, st>>>(g);
    } else if (big >= 192) {
        gemm_nt_f32_kernel<128, 128><<<dim3((unsigned)big), dim3(256), 0, st>>>(g);
    } else {
        const long small = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
        gemm_nt_f32_kernel<64, 64><<<dim3((unsigned)small), dim3(256), 0, st>>>(g);
    }
    RLCF_LAUNCH_CHECK();
    return RLCF_OK;
}
