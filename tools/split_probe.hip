// Does MODE.FP16_OVFL make v_cvt_f16_f32 / the fp8 conversions saturate on gfx950, and does v_cvt_scalef32_pk_fp8_f32 divide by its scale?
// Compares the clamped reference encoding of wsu_split4_f16f8 with a shorter instruction sequence on edge values.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(2))) short i16x2;
__device__ float clamp8(float v) { return __builtin_amdgcn_fmed3f(v, -448.f, 448.f); }
__global__ void k(const float* x, uint32_t* out, int n, int ovfl) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float v0 = x[2 * i], v1 = x[2 * i + 1];
    // reference
    const float c0 = __builtin_amdgcn_fmed3f(v0, -65504.f, 65504.f), c1 = __builtin_amdgcn_fmed3f(v1, -65504.f, 65504.f);
    const _Float16 h0 = (_Float16)c0, h1 = (_Float16)c1;
    const uint32_t hi_ref = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
    const int lo_ref = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8((c0 - (float)h0) * 4096.f), clamp8((c1 - (float)h1) * 4096.f), 0, false);
    const int x_ref = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(c0 * 0.25f), clamp8(c1 * 0.25f), 0, false);
    // short sequence
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);        // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
    const _Float16 g0 = (_Float16)v0, g1 = (_Float16)v1;
    const uint32_t hi_new = (uint32_t)__builtin_bit_cast(uint16_t, g0) | ((uint32_t)__builtin_bit_cast(uint16_t, g1) << 16);
    i16x2 z = {0, 0};
    const i16x2 lo_new = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, v0 - (float)g0, v1 - (float)g1, 0x1p-12f, false);
    const i16x2 x_new = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, v0, v1, 4.0f, false);
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 0);
    out[6 * i + 0] = hi_ref; out[6 * i + 1] = (uint32_t)lo_ref & 0xffff; out[6 * i + 2] = (uint32_t)x_ref & 0xffff;
    out[6 * i + 3] = hi_new; out[6 * i + 4] = (uint16_t)lo_new[0]; out[6 * i + 5] = (uint16_t)x_new[0];
}
int main() {
    std::vector<float> x = {0.f, 1.f, 1e-8f, 1e-3f, 0.0123f, 0.7f, 3.14159f, 100.3f, 447.f, 448.f, 449.f, 460.f, 1000.7f, 1791.f, 1793.f, 5000.f,
                            65503.f, 65504.f, 65520.f, 70000.f, 1e6f, 3e38f, -1.f, -0.3f, -449.f, -70000.f, 6.1e-5f, 5e-5f, 2.5f, 1.0009765f, 255.9f, 0.06f};
    uint32_t st = 99;
    for (int i = 0; i < 96; ++i) { st = st * 1664525u + 1013904223u; x.push_back(((int)(st >> 8) % 2000001 - 1000000) * 1e-3f * ((i & 3) == 0 ? 0.001f : 1.f)); }
    const int n = (int)x.size() / 2;
    float* dx; uint32_t* dout; hipMalloc(&dx, x.size() * 4); hipMalloc(&dout, n * 24);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, n, ovfl);
        std::vector<uint32_t> o(n * 6);
        if (hipMemcpy(o.data(), dout, n * 24, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            const bool ok = o[6 * i] == o[6 * i + 3] && o[6 * i + 1] == o[6 * i + 4] && o[6 * i + 2] == o[6 * i + 5];
            if (!ok) { ++bad; printf("FP16_OVFL=%d  x=(%g, %g): hi %08x vs %08x  lo %04x vs %04x  x8 %04x vs %04x\n", ovfl, x[2 * i], x[2 * i + 1],
                              o[6 * i], o[6 * i + 3], o[6 * i + 1], o[6 * i + 4], o[6 * i + 2], o[6 * i + 5]); }
        }
        printf("FP16_OVFL=%d: %d of %d pairs differ\n", ovfl, bad, n);
    }
    return 0;
}
