// Semantics of the gfx950 fp4 pieces a block-scaled fp4 cross term would use (round 3 study; hipcc --offload-arch=gfx950 -O2 -o tools/fp4_probe tools/fp4_probe.hip):
//   1. v_cvt_scalef32_pk_fp4_f16: is the result fp4(x / scale) or fp4(x * scale)?  which nibble gets the first value?  rounding?
//   2. v_cvt_scalef32_pk_f16_fp8: f16(fp8 * scale) or / scale?
//   3. v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands: which nibble of which register is K index i of a lane; how the E8M0 scale bytes apply.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void cvt_kernel(unsigned* o4, float* o16, const float* vals, int n, float scale) {
    const int i = threadIdx.x;
    if (i >= n) return;
    h2 v = {(_Float16)vals[i], (_Float16)(-vals[i] * 0.5f)};
    o4[i * 2 + 0] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0xAAAAAAAAu, v, scale, 0);
    o4[i * 2 + 1] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0xAAAAAAAAu, v, scale, 2);
    const unsigned b = 0x38u | (0x40u << 8) | (0xB8u << 16) | (0x30u << 24);      // e4m3: 1.0, 2.0, -1.0, 0.5
    const h2 lo = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(b, scale, false), hi = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(b, scale, true);
    o16[i * 4 + 0] = (float)lo.x; o16[i * 4 + 1] = (float)lo.y; o16[i * 4 + 2] = (float)hi.x; o16[i * 4 + 3] = (float)hi.y;
}

// one MFMA: lane (row = l & 31, half = l >> 5) holds K indices 32 half .. 32 half + 31 as 32 nibbles in a[0..3]
__global__ void mfma_kernel(float* out, int which_k, int sa, int sb) {
    const int l = threadIdx.x, row = l & 31, half = l >> 5;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    // A[row][k] = 1.0 (code 2) only at k == which_k; B[k][col] = code (col % 8) at every k (value table 0, .5, 1, 1.5, 2, 3, 4, 6)
    if (which_k / 32 == half) { const int kk = which_k % 32; a[kk / 8] = 0x2 << (4 * (kk % 8)); }
    const unsigned code = row % 8;
    const unsigned rep = code * 0x11111111u;
    b[0] = rep; b[1] = rep; b[2] = rep; b[3] = rep;
    if (which_k < 0) { a[0] = a[1] = a[2] = a[3] = 0x22222222; }                // all ones: dot over K = 64 * value
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, sa + (l & 1 ? 0 : 0), 0, sb);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = acc[r];
}
// scale per lane: lanes of row r use scale byte sa_even / sa_odd by row parity -> does the scale follow the lane's row?
__global__ void mfma_scale_kernel(float* out, int sa_even, int sa_odd, int sb_half0, int sb_half1) {
    const int l = threadIdx.x, row = l & 31, half = l >> 5;
    i32x8 a = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0}, b = a;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, (row & 1) ? sa_odd : sa_even, 0, half ? sb_half1 : sb_half0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = acc[r];
}

int main() {
    const float vals[] = {0.f, 0.2f, 0.25f, 0.3f, 0.5f, 0.74f, 0.75f, 0.76f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.5f, 3.f, 3.5f, 4.f, 5.f, 6.f, 7.f, 8.f, 100.f};
    const int n = sizeof(vals) / sizeof(float);
    float* dv; unsigned* d4; float* d16;
    hipMalloc(&dv, sizeof(vals)); hipMalloc(&d4, n * 8); hipMalloc(&d16, n * 16);
    hipMemcpy(dv, vals, sizeof(vals), hipMemcpyHostToDevice);
    for (float scale : {1.f, 2.f, 0.5f}) {
        cvt_kernel<<<1, 64>>>(d4, d16, dv, n, scale);
        unsigned h4[64]; float h16[128];
        hipMemcpy(h4, d4, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h16, d16, n * 16, hipMemcpyDeviceToHost);
        printf("scale %g: fp8 bytes (1, 2, -1, .5) -> f16 %g %g %g %g\n", scale, h16[0], h16[1], h16[2], h16[3]);
        for (int i = 0; i < n; ++i) printf("  x = %6.2f, -x/2 = %6.2f -> sel0 %08x sel2 %08x\n", vals[i], -vals[i] * 0.5f, h4[2 * i], h4[2 * i + 1]);
    }
    float* dout; hipMalloc(&dout, 64 * 16 * 4);
    float hout[64 * 16];
    auto show = [&](const char* what) {
        hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
        printf("%s: lane 0 regs:", what); for (int r = 0; r < 16; ++r) printf(" %g", hout[r]);
        printf("\n   lanes 0..9 reg 0:"); for (int l = 0; l < 10; ++l) printf(" %g", hout[l * 16]);
        printf("   lane 32 reg 0: %g\n", hout[32 * 16]);
    };
    mfma_kernel<<<1, 64>>>(dout, -1, 127, 127); show("A = all 1.0, B[k][col] = table[col % 8], scales 127/127 (expect 64 * table[col % 8] in column col)");
    for (int k : {0, 1, 7, 8, 31, 32, 33, 63}) { mfma_kernel<<<1, 64>>>(dout, k, 127, 127); char buf[64]; snprintf(buf, 64, "A = 1.0 at k = %d only", k); show(buf); }
    mfma_kernel<<<1, 64>>>(dout, -1, 128, 127); show("scale_a 128 (x2?)");
    mfma_kernel<<<1, 64>>>(dout, -1, 127, 125); show("scale_b 125 (/4?)");
    mfma_scale_kernel<<<1, 64>>>(dout, 127, 129, 127, 127); show("scale_a by row parity (even 127, odd 129), ones x ones = 64 -> rows: reg r of lane = row 8*(r/4) + 4*half + r%4");
    mfma_scale_kernel<<<1, 64>>>(dout, 127, 127, 127, 130); show("scale_b by K half (half0 127, half1 130): 32 * 1 + 32 * 8 = 288 if the scale follows the lane's K half");
    hipDeviceSynchronize();
    return 0;
}
