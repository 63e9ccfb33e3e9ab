// Semantics of ds_read_b64_tr_b8 on gfx950: which (row, col) bytes does each lane receive when lane 2q+p of a 16-lane group
// supplies &M[r0 + q][8p] (8 rows x 16 byte-columns per group)?   hipcc --offload-arch=gfx950 -o tools/probe_tr8 tools/probe_tr8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
__global__ void k(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t m[64 * 80];       // M[row][col], row stride 80 bytes; value encodes row (hi) / col (lo) mod 16
    for (int i = threadIdx.x; i < 64 * 80; i += 64) m[i] = (uint8_t)(((i / 80) & 15) << 4 | ((i % 80) & 15));
    __syncthreads();
    const int lane = threadIdx.x;
    const int grp = lane >> 4, li = lane & 15, q = li >> 1, p = li & 1;
    const uint8_t* addr = &m[(8 * grp + q) * 80 + 8 * p];
    i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)addr);
    for (int e = 0; e < 8; ++e) out[lane * 8 + e] = (uint8_t)(((unsigned)v[e >> 2] >> (8 * (e & 3))) & 0xff);
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 8 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 34; ++l) { printf("lane %2d:", l); for (int e = 0; e < 8; ++e) printf(" (r%d,c%d)", h[l*8+e] >> 4, h[l*8+e] & 15); printf("\n"); }
    return 0;
}
