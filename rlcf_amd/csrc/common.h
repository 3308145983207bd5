// Shared helpers for the gfx950 kernels of librlcf_hip.so (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/rlcf_hip.h"

void rlcf_set_error(const char* fmt, ...);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel): the attribute belongs to the kernel ON A DEVICE, so a
// process-wide "already set" flag would leave the second GPU of a process at the 64 KB default.  Returns an rlcf_status.
int rlcf_func_lds(const void* fn, size_t bytes);

#define RLCF_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            rlcf_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return RLCF_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)
#define RLCF_LAUNCH_CHECK() RLCF_HIP_CHECK(hipGetLastError())
#define RLCF_ARG_CHECK(cond)                                                              \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            rlcf_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond);         \
            return RLCF_ERR_ARG;                                                          \
        }                                                                                 \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

#define LN_EPS 1e-5f
#define HEAD_DIM 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// row of a 32x32 MFMA accumulator register r held by a lane in half h (cdna guide §3)
__device__ __forceinline__ int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float quick_gelu(float v) { return v / (1.0f + expf(-1.702f * v)); }
// the same through v_exp_f32 / v_rcp_f32 (1 ulp each) instead of expf + an IEEE division (~25 VALU operations per element): the
// epilogue of the split-f16 c_fc product evaluates it 128 times per thread and tile (relative error ~3e-7, inside the 22-bit operands)
__device__ __forceinline__ float quick_gelu_fast(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896341f * v));
}
__device__ __forceinline__ float quick_gelu_grad_fast(float f) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896341f * f));
    return s * (1.0f + 1.702f * f * (1.0f - s));
}
__device__ __forceinline__ float quick_gelu_grad(float f) {
    float s = 1.0f / (1.0f + expf(-1.702f * f));
    return s * (1.0f + 1.702f * f * (1.0f - s));
}

// ---- internal launchers shared between translation units (all async on `st`) ----
// max|v| of a GEMM's output, folded into the epilogue (non-negative floats order like their bit patterns)
__device__ __forceinline__ void amax_commit(unsigned int* out, float m) {
    if (!out) return;
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

struct GemmArgs {
    const void* A; int lda;        // [M,K] f32
    const void* W; int ldw;        // [N,K]
    const float* bias;             // [N] or null
    const float* residual; int ldr;
    const float* aux; int ldaux;
    void* C; int ldc;              // [M,N] f32
    int M, N, K;
    float alpha; int epilogue;
    unsigned int* amax_out;        // optional: atomicMax of |C| (as float bits) — the next GEMM's split-f16 operand scale
};
int launch_gemm_f32(const GemmArgs& g, hipStream_t st);
