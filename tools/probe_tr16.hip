#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t m[64 * 64];     // M[row][col], row stride 64 elements (128 B)
    for (int i = threadIdx.x; i < 64 * 64; i += 64) m[i] = (uint16_t)i;   // value = row*64 + col
    __syncthreads();
    const int lane = threadIdx.x;
    const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // every 16-lane group reads the 4x16 block at rows 4*grp.., cols 0..15: lane 4q+p supplies &M[r0+q][4p]
    const uint16_t* addr = &m[(4 * grp + q) * 64 + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)v[e];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l*4+e] / 64, h[l*4+e] % 64); printf("\n"); }
    return 0;
}
