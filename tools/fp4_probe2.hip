// which nibble position of the B operand meets nibble position p of the A operand in v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float* out, int p, int mode, int ahalf) {
    const int l = threadIdx.x, half = l >> 5;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == ahalf) a[p / 8] = 0x2 << (4 * (p % 8));                   // A = 1.0 at nibble position p of lane half `ahalf`
    for (int q = 0; q < 32; ++q) {                                          // B: nibble position q of half h holds code f(q, h)
        const int code = mode == 0 ? (q % 8) : mode == 1 ? (q / 4) : (half ? 3 : 5);
        b[q / 8] |= code << (4 * (q % 8));
    }
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 127, 0, 127);
    if (l == 0) out[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4); float h;
    const float tab[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6};
    for (int ahalf = 0; ahalf < 2; ++ahalf)
        for (int p = 0; p < 32; ++p) {
            float v[3];
            for (int mode = 0; mode < 3; ++mode) { k<<<1, 64>>>(d, p, mode, ahalf); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); v[mode] = h; }
            int c0 = -1, c1 = -1; for (int i = 0; i < 8; ++i) { if (tab[i] == v[0]) c0 = i; if (tab[i] == v[1]) c1 = i; }
            // position q with q % 8 == c0 and q / 4 == c1
            int q = -1; for (int t = 0; t < 32; ++t) if (t % 8 == c0 && t / 4 == c1) q = t;
            printf("A half %d nibble %2d  meets B nibble %2d of half %s (values %g %g %g)\n", ahalf, p, q, v[2] == 1.5f ? "1" : v[2] == 3.f ? "0" : "?", v[0], v[1], v[2]);
        }
    return 0;
}
