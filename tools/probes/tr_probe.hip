#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
// every lane supplies the address of 4 halves: lds[addr_tab[lane]]; output: what each lane got
__global__ void k(const int* addr_tab, float* out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)(float)i;     // value = index (exact up to 2048)
    __syncthreads();
    const int l = threadIdx.x;
    fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(lds + addr_tab[l]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main() {
    std::vector<int> tab(64);
    // experiment: lane l supplies address 4*l + 256*(something) so every lane's chunk is identifiable
    for (int l = 0; l < 64; ++l) tab[l] = 4 * l + 8 * (l / 16) * 64;   // group g at base g*512+..., lanes 4 halves each
    for (int l = 0; l < 64; ++l) tab[l] = 4 * l;
    int* d_tab; float* d_out;
    hipMalloc(&d_tab, 256); hipMalloc(&d_out, 1024);
    hipMemcpy(d_tab, tab.data(), 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_tab, d_out);
    std::vector<float> o(256);
    hipMemcpy(o.data(), d_out, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4.0f %4.0f %4.0f %4.0f\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    return 0;
}
