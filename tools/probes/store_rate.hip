// Per-CU global store throughput on gfx950 for the epilogue patterns of the persistent f16 GEMM (tools/probes, measurement only).
// 256 workgroups x 512 threads; every wave issues `per_wave` 16-byte-per-lane stores per "tile", `reps` tiles; patterns:
//   0: 8 rows x 128 B per instruction, row stride ld (the GEMM's f16 output tile, first form)      1: 1 KB contiguous per instruction
//   2: 32 rows x 32 B per instruction (the transpose-free epilogue: lane = row; four instructions complete a row's 128-B line)
// policy: 0 default, 1 nontemporal, 2 sc1 (write-through), 3 sc0 sc1
// build: hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip ; run: ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int PAT, int POL>
__global__ __launch_bounds__(512) void k(char* out, size_t ld, int reps, size_t tile_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    for (int r = 0; r < reps; ++r) {
        char* base = out + ((size_t)blockIdx.x * reps + r) * tile_stride;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            char* p;
            if (PAT == 0) p = base + (size_t)(wm * 128 + s * 8 + (lane >> 3)) * ld + wn * 128 + (lane & 7) * 16;
            else if (PAT == 2) p = base + (size_t)(wm * 128 + (s >> 2) * 32 + (lane & 31)) * ld + wn * 128 + (s & 3) * 32 + (lane >> 5) * 16;
            else p = base + (size_t)(wave * 16 + s) * 1024 + lane * 16;
            if (POL == 0) *(u32x4*)p = v;
            else if (POL == 1) __builtin_nontemporal_store(v, (u32x4*)p);
            else {
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
            }
            v.x += 1;
        }
    }
}
int main() {
    const int reps = 64;
    const size_t ld = 4608, tile_stride = (size_t)256 * ld;         // pattern 0: a 256-row band of a [*, 2304] f16 matrix per tile
    char* buf; size_t bytes = (size_t)256 * reps * tile_stride;
    if (hipMalloc((void**)&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256;
    auto run = [&](const char* name, void (*fn)(char*, size_t, int, size_t), size_t ts) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            fn<<<grid, 512>>>(buf, ld, reps, ts);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double gb = (double)grid * reps * 8 * 16 * 1024 / 1e9;
            if (it == 2) printf("%-40s %8.1f us  %7.1f GB/s  %6.1f B/clk/CU @2.4GHz  (%.2f us per 128 KB tile)\n", name, ms * 1e3, gb / ms * 1e3, gb / ms * 1e3 / grid / 2.4, ms * 1e3 / reps);
        }
    };
    for (int gsz : {8, 32, 64, 128}) { grid = gsz; printf("grid %d: ", gsz); run("rows 8x128B, default", k<0, 0>, tile_stride); }
    grid = 256;
    run("rows 8x128B, default", k<0, 0>, tile_stride);
    run("rows 8x128B, nontemporal", k<0, 1>, tile_stride);
    run("rows 8x128B, sc1", k<0, 2>, tile_stride);
    run("rows 32x32B, default", k<2, 0>, tile_stride);
    run("rows 32x32B, nontemporal", k<2, 1>, tile_stride);
    for (int gsz : {8, 64}) { grid = gsz; printf("grid %d: ", gsz); run("rows 32x32B, nontemporal", k<2, 1>, tile_stride); }
    grid = 256;
    run("contiguous 1KB, default", k<1, 0>, 131072);
    run("contiguous 1KB, nontemporal", k<1, 1>, 131072);
    run("contiguous 1KB, sc1", k<1, 2>, 131072);
    return 0;
}
