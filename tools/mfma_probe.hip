// Sustained MFMA issue rate under the board's power cap, per operand format (gfx950).  The conv3x3 kernel's run time follows its MFMA
// count (profiles/r01/conv3x3_ablation.md), so what a format costs is its *sustained* rate, not the data-sheet one.  Every wave
// keeps 4 independent accumulator chains busy from registers only (no LDS / memory in the loop); operands are random finite values.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip && tools/mfma_probe [seconds-per-case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// KIND 0: bf16 32x32x16; 1: block-scaled fp8(e4m3) 32x32x64; 2: two bf16 + one fp8 per step (hi*hi in bf16, both cross terms in one fp8);
// 3: bf16 16x16x32; 4: block-scaled fp6(e2m3) 32x32x64; 5: block-scaled fp8 16x16x128; 6 / 7: the 9 f16 : 5 fp8 instruction mix of
// conv3x3_pl in the 32x32 / 16x16 shapes (round 3: is a change of MFMA shape worth a rewrite of the matrix section?)
template <int KIND>
__global__ __launch_bounds__(512) void probe(const uint32_t* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    i32x8 a8, b8;
    #pragma unroll
    for (int i = 0; i < 8; ++i) { a8[i] = src[(lane * 8 + i) & 4095]; b8[i] = src[(2048 + lane * 8 + i) & 4095]; }
    bf16x8 ah, bh;
    {
        union { uint32_t u[4]; bf16x8 v; } ua, ub;
        #pragma unroll
        for (int i = 0; i < 4; ++i) { ua.u[i] = src[4096 + ((lane * 4 + i) & 1023)]; ub.u[i] = src[5120 + ((lane * 4 + i) & 1023)]; }
        ah = ua.v; bh = ub.v;
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int it = 0; it < 2 * iters; ++it) {
        // every other pass uses -A: the accumulators oscillate instead of saturating where the addend falls below one ulp
        #pragma unroll
        for (int i = 0; i < 8; ++i) a8[i] ^= (KIND == 4 ? 0x20820820 : 0x80808080);
        { union { uint32_t u[4]; bf16x8 v; } t; t.v = ah;
          _Pragma("unroll") for (int i = 0; i < 4; ++i) t.u[i] ^= 0x80008000u;
          ah = t.v; }
        if constexpr (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c3, 0, 0, 0);
        } else if constexpr (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else if constexpr (KIND == 2) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c3, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, c3, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c2, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c3, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
        } else if constexpr (KIND == 3) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d3, 0, 0, 0);
        } else if constexpr (KIND == 5) {                                  // block-scaled fp8 in the 16x16 shape
            d0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            d1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            d2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            d3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else if constexpr (KIND == 6) {                                  // the f16f8 mix of conv3x3_pl per 18 taps of one 32x32 tile: 18 f16 + 10 fp8 (32x32 shapes)
            _Pragma("unroll") for (int t = 0; t < 9; ++t) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, c1, 0, 0, 0);
            }
            _Pragma("unroll") for (int t = 0; t < 5; ++t) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
            }
        } else if constexpr (KIND == 7) {                                  // the same work in the 16x16 shapes: 36 x 16x16x32 (K = 2 taps) + 20 x 16x16x128 per 2 x 4 tiles
            _Pragma("unroll") for (int t = 0; t < 9; ++t) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, d3, 0, 0, 0);
            }
            _Pragma("unroll") for (int t = 0; t < 5; ++t) {
                d0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d0, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
                d1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d1, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
                d2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d2, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
                d3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d3, 0, 0, 0, 0x7f767f76, 0, 0x7f767f76);
            }
        } else {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c2, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c3, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    float s = 0.f;
    #pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    #pragma unroll
    for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;   // keep the chains alive, never true in practice
}

struct Case { const char* name; void (*fn)(const uint32_t*, float*, int); double flop_per_iter_per_wave; double bf16_passes_per_iter; };

int main(int argc, char** argv) {
    double secs = argc > 1 ? atof(argv[1]) : 2.0;
    std::vector<uint32_t> h(6144);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    // bytes 0..4095 words: fp8 e4m3 / fp6 payloads with exponent fields kept off the NaN pattern (0x7f / 0xff): clear bit 6 of each byte
    for (int i = 0; i < 4096; ++i) { uint32_t w = (rnd() << 8) ^ rnd(); h[i] = w & 0xbfbfbfbfu; }
    // bf16 pairs in roughly [-2, 2): sign random, exponent 0x3f or 0x3e.., mantissa random
    for (int i = 4096; i < 6144; ++i) {
        uint32_t r = rnd();
        uint16_t lo = (uint16_t)(((r & 1) << 15) | (0x3f00 - ((r >> 1) & 3) * 0x80) | ((r >> 3) & 0x7f));
        uint16_t hi = (uint16_t)((((r >> 10) & 1) << 15) | (0x3f00 - ((r >> 11) & 3) * 0x80) | ((r >> 13) & 0x7f));
        h[i] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
    uint32_t* src; float* out;
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int blocks = 512, threads = 512;                     // 2 workgroups of 8 waves per CU, the conv3x3 kernel's occupancy
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    Case cases[] = {
        {"bf16 32x32x16", probe<0>, 8 * 32768.0, 8},
        {"fp8(e4m3) scaled 32x32x64", probe<1>, 8 * 131072.0, 16},
        {"mix 8 bf16 + 4 fp8 (x3-equivalent of 12 bf16)", probe<2>, 16 * 32768.0 + 8 * 131072.0, 32},
        {"bf16 16x16x32", probe<3>, 8 * 16384.0, 4},
        {"fp6(e2m3) scaled 32x32x64", probe<4>, 8 * 131072.0, 8},
        {"fp8(e4m3) scaled 16x16x128", probe<5>, 8 * 65536.0, 8},
        // the conv3x3_pl matrix mix (9 f16 : 5 fp8 instructions per chunk and tile), 38 units of 32 cycles per iteration in both shapes
        {"conv3x3_pl mix, 32x32 shapes (18 f16 + 10 fp8)", probe<6>, 2 * (18 * 32768.0 + 10 * 131072.0), 2 * 38},
        {"conv3x3_pl mix, 16x16 shapes (36 f16 + 20 fp8)", probe<7>, 2 * (36 * 16384.0 + 20 * 65536.0), 2 * 38},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& c : cases) {
        int iters = 20000;
        hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(threads), 0, 0, src, out, iters); CK(hipDeviceSynchronize());
        // calibrate one launch to ~50 ms, then repeat launches for `secs` so the clock settles under the power cap
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(threads), 0, 0, src, out, iters); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        iters = (int)(iters * 50.0 / ms); if (iters < 1000) iters = 1000;
        int reps = (int)(secs * 1000.0 / 50.0); if (reps < 4) reps = 4;
        double first = 0, last = 0, total = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(threads), 0, 0, src, out, iters); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (r == 0) first = ms; last = ms; if (r >= reps / 2) total += ms;
        }
        double avg = total / (reps - reps / 2);
        double waves = (double)blocks * threads / 64;
        double tflops = c.flop_per_iter_per_wave * iters * waves / (avg * 1e-3) / 1e12;
        // 1024 SIMDs; a bf16 32x32x16 occupies its SIMD for 32 cycles; the 4th Case field counts an iteration in those units
        double simd_busy_cycles = c.bf16_passes_per_iter * 32.0 /*cycles per 32x32x16 bf16*/ * iters * (waves / 1024.0);
        double ghz = simd_busy_cycles / (avg * 1e-3) / 1e9;
        printf("%-48s first %.2f ms  settled %.2f ms  last %.2f ms  %.0f TFLOP/s  (clock if MFMA-pipe-bound: %.2f GHz)\n", c.name, first, avg, last, tflops, ghz);
        fflush(stdout);
    }
    return 0;
}
