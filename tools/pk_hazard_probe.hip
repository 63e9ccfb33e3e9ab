// Hardware probe for the packed-f32 read-after-write pattern of tools/pk_hazard.py:
//
//     ds_read_b64 w0 ; ds_read_b64 w1 ; s_waitcnt lgkmcnt(1)
//     v_pk_fma_f32 acc, x0, w0, 0      op_sel_hi:[1,0,0]
//     <BETWEEN>
//     v_pk_fma_f32 acc, x1, w1, acc    op_sel_hi:[1,0,1]
//
// BETWEEN = 0: `s_waitcnt lgkmcnt(0)` (the 8 sites of the SLP build of conv3x3_pl_kernel<1 / 4>), 1: `s_nop 0` + the wait (what the
// patched build runs), 2: nothing at all with both reads waited for up front (never emitted by hipcc: it pads these), 3: the wait placed
// BEFORE the first pk_fma and `s_nop 0` between.  Each lane checks its result bitwise against fmaf(x1, w1, fmaf(x0, w0, 0)); a fraction of
// the waves runs MFMAs instead (SIMD partners as in the real kernel).  Prints mismatches per variant and per (register half, lane quarter).
// Build: hipcc --offload-arch=gfx950 -O2 tools/pk_hazard_probe.hip -o tools/pk_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int BETWEEN>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ xin, unsigned* __restrict__ bad, int iters) {
    __shared__ __attribute__((aligned(16))) float s_w[2048];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 2048; i += 512) s_w[i] = xin[(blockIdx.x * 2048 + i) & 0xFFFF] + 0.25f;
    __syncthreads();
    if (wv >= 4) {                                           // SIMD partners: keep the matrix pipe and the LDS busy
        f32x16 acc = {};
        f16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.01f * (lane + k)); b[k] = (_Float16)(0.02f * (lane - k)); }
        for (int it = 0; it < iters * 2; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            a[it & 7] = (_Float16)s_w[(it * 64 + lane) & 2047];
        }
        if (acc[0] == 12345.678f) bad[63] = 1;              // keep it alive
        return;
    }
    unsigned nbad[2] = {0, 0};
    for (int it = 0; it < iters; ++it) {
        const int o = ((it * 37 + wv * 11) & 127) * 16;     // 8-byte aligned, moves around the LDS
        const float x0f[2] = {xin[(it * 64 + lane) & 0xFFFF], xin[(it * 64 + lane + 7777) & 0xFFFF]};
        const float x1f[2] = {xin[(it * 64 + lane + 333) & 0xFFFF], xin[(it * 64 + lane + 4444) & 0xFFFF]};
        const unsigned long long x0 = (unsigned long long)__builtin_bit_cast(unsigned, x0f[0]) | ((unsigned long long)__builtin_bit_cast(unsigned, x0f[1]) << 32);
        const unsigned long long x1 = (unsigned long long)__builtin_bit_cast(unsigned, x1f[0]) | ((unsigned long long)__builtin_bit_cast(unsigned, x1f[1]) << 32);
        const unsigned addr = (unsigned)(size_t)(s_w) + o + 0;   // wave-uniform address: every lane reads the same weights (as the head does)
        unsigned long long w0, w1, acc;                      // 64-bit integers as asm operands = VGPR pairs (lo dword = element 0)
        if constexpr (BETWEEN == 0)
            asm volatile("ds_read_b64 %1, %3 offset:32\n\tds_read_b64 %2, %3\n\ts_waitcnt lgkmcnt(1)\n\t"
                         "v_pk_fma_f32 %0, %4, %1, 0 op_sel_hi:[1,0,0]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_pk_fma_f32 %0, %5, %2, %0 op_sel_hi:[1,0,1]\n\ts_nop 1"
                         : "=&v"(acc), "=&v"(w0), "=&v"(w1) : "v"(addr), "v"(x0), "v"(x1) : "memory");
        else if constexpr (BETWEEN == 1)
            asm volatile("ds_read_b64 %1, %3 offset:32\n\tds_read_b64 %2, %3\n\ts_waitcnt lgkmcnt(1)\n\t"
                         "v_pk_fma_f32 %0, %4, %1, 0 op_sel_hi:[1,0,0]\n\t"
                         "s_nop 0\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "v_pk_fma_f32 %0, %5, %2, %0 op_sel_hi:[1,0,1]\n\ts_nop 1"
                         : "=&v"(acc), "=&v"(w0), "=&v"(w1) : "v"(addr), "v"(x0), "v"(x1) : "memory");
        else if constexpr (BETWEEN == 2)
            asm volatile("ds_read_b64 %1, %3 offset:32\n\tds_read_b64 %2, %3\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "v_pk_fma_f32 %0, %4, %1, 0 op_sel_hi:[1,0,0]\n\t"
                         "v_pk_fma_f32 %0, %5, %2, %0 op_sel_hi:[1,0,1]\n\ts_nop 1"
                         : "=&v"(acc), "=&v"(w0), "=&v"(w1) : "v"(addr), "v"(x0), "v"(x1) : "memory");
        else
            asm volatile("ds_read_b64 %1, %3 offset:32\n\tds_read_b64 %2, %3\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "v_pk_fma_f32 %0, %4, %1, 0 op_sel_hi:[1,0,0]\n\t"
                         "s_nop 0\n\t"
                         "v_pk_fma_f32 %0, %5, %2, %0 op_sel_hi:[1,0,1]\n\ts_nop 1"
                         : "=&v"(acc), "=&v"(w0), "=&v"(w1) : "v"(addr), "v"(x0), "v"(x1) : "memory");
        const float wa = s_w[(o + 32) / 4], wb = s_w[o / 4];
        const float e0 = __builtin_fmaf(x1f[0], wb, __builtin_fmaf(x0f[0], wa, 0.f));
        const float e1 = __builtin_fmaf(x1f[1], wb, __builtin_fmaf(x0f[1], wa, 0.f));
        nbad[0] += (unsigned)acc != __builtin_bit_cast(unsigned, e0);
        nbad[1] += (unsigned)(acc >> 32) != __builtin_bit_cast(unsigned, e1);
    }
    if (nbad[0]) atomicAdd(&bad[0 * 4 + (lane >> 4)], nbad[0]);
    if (nbad[1]) atomicAdd(&bad[1 * 4 + (lane >> 4)], nbad[1]);
}

// The form that turned out to be the failing one (profiles/r03/pk_hazard.md): a packed add whose src0 has op_sel_hi = 0 -- hipcc's hazard
// rule does not see it as a forwarding producer and leaves the consumer adjacent -- followed by a plain VALU reading the LOW dword:
//     v_pk_add_f32 v[200:201], bias, acc op_sel_hi:[0,1]      ; (acc.q0 + b, acc.q1 + b)
//     <BETWEEN: nothing | s_nop 0>
//     v_max_f32 r0, 0, v200 ; v_max_f32 r1, 0, v201
// The operands of the add come straight out of an MFMA accumulator chain, as in the epilogue of conv3x3_pl_kernel.
template <int BETWEEN>
__global__ __launch_bounds__(512) void probe_add(const float* __restrict__ xin, unsigned* __restrict__ bad, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned nbad[2] = {0, 0};
    f32x16 acc = {};
    f16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.01f * (lane + k)); b[k] = (_Float16)(0.02f * (lane - k)); }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        a[it & 7] = (_Float16)xin[(it * 64 + lane) & 0xFFFF];
        const float q0 = acc[it & 15] * 1e-3f + xin[(it * 64 + lane + 99) & 0xFFFF], q1 = acc[(it + 5) & 15] * 1e-3f - xin[(it * 64 + lane + 777) & 0xFFFF];
        const float bias = xin[(it * 3 + 17) & 0xFFFF];
        const unsigned long long accp = (unsigned long long)__builtin_bit_cast(unsigned, q0) | ((unsigned long long)__builtin_bit_cast(unsigned, q1) << 32);
        const unsigned long long bp = (unsigned long long)__builtin_bit_cast(unsigned, bias);
        float r0, r1;
        if constexpr (BETWEEN == 0)
            asm volatile("v_pk_add_f32 v[200:201], %2, %3 op_sel_hi:[0,1]\n\tv_max_f32 %0, 0, v200\n\tv_max_f32 %1, 0, v201\n\ts_nop 1"
                         : "=&v"(r0), "=&v"(r1) : "v"(bp), "v"(accp) : "v200", "v201");
        else
            asm volatile("v_pk_add_f32 v[200:201], %2, %3 op_sel_hi:[0,1]\n\ts_nop 0\n\tv_max_f32 %0, 0, v200\n\tv_max_f32 %1, 0, v201\n\ts_nop 1"
                         : "=&v"(r0), "=&v"(r1) : "v"(bp), "v"(accp) : "v200", "v201");
        const float e0 = fmaxf(q0 + bias, 0.f), e1 = fmaxf(q1 + bias, 0.f);
        nbad[0] += __builtin_bit_cast(unsigned, r0) != __builtin_bit_cast(unsigned, e0);
        nbad[1] += __builtin_bit_cast(unsigned, r1) != __builtin_bit_cast(unsigned, e1);
    }
    if (nbad[0]) atomicAdd(&bad[0 * 4 + (lane >> 4)], nbad[0]);
    if (nbad[1]) atomicAdd(&bad[1 * 4 + (lane >> 4)], nbad[1]);
    if (acc[0] == 12345.678f) bad[63] = 1;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000, launches = argc > 2 ? atoi(argv[2]) : 20;
    std::vector<float> h(65536);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 22); }
    float* x; unsigned* bad;
    hipMalloc(&x, h.size() * 4); hipMalloc(&bad, 64 * 4);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"pk_fma ; s_waitcnt lgkmcnt(0) ; pk_fma   (the 8 sites)", "pk_fma ; s_nop 0 ; s_waitcnt ; pk_fma    (patched)",
                            "pk_fma ; pk_fma adjacent                 (never emitted)", "pk_fma ; s_nop 0 ; pk_fma                (compiler's padding)"};
    for (int v = 0; v < 4; ++v) {
        hipMemset(bad, 0, 64 * 4);
        for (int l = 0; l < launches; ++l) {
            if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, x, bad, iters);
            if (v == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, x, bad, iters);
            if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, x, bad, iters);
            if (v == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, x, bad, iters);
        }
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned hb[8];
        hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
        const double total = (double)launches * 256 * 256 * iters;
        printf("variant %d  %s : wrong lo-half by lane quarter [%u %u %u %u]  hi-half [%u %u %u %u]  of %.3g lane-results each\n", v, names[v],
               hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], hb[7], total / 4);
    }
    const char* names2[2] = {"pk_add op_sel_hi:[0,1] ; v_max adjacent   (the failing sites)", "pk_add op_sel_hi:[0,1] ; s_nop 0 ; v_max  (patched)"};
    for (int v = 0; v < 2; ++v) {
        hipMemset(bad, 0, 64 * 4);
        for (int l = 0; l < launches; ++l) {
            if (v == 0) hipLaunchKernelGGL(probe_add<0>, dim3(256), dim3(512), 0, 0, x, bad, iters);
            if (v == 1) hipLaunchKernelGGL(probe_add<1>, dim3(256), dim3(512), 0, 0, x, bad, iters);
        }
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned hb[8];
        hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
        printf("variant %d  %s : wrong lo-half by lane quarter [%u %u %u %u]  hi-half [%u %u %u %u]  of %.3g lane-results each\n", 4 + v, names2[v],
               hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], hb[7], (double)launches * 256 * 512 * iters / 4);
    }
    return 0;
}
