// Microbenchmark: can an MFMA-only wave and a memory-streaming wave on the SAME SIMD overlap on MI355X?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a register-only bf16 MFMA loop, waves 4-7 (their SIMD
// partners) stream-copy global memory (optionally through LDS).  mode 1 = MFMA only, 2 = copy only, 3 = both.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap.hip -o gpurun_out/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int VIA_LDS>
__global__ __launch_bounds__(512, 2) void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t vec_per_block,
                                            int mfma_iters, int mode, float* sink, int split, int duty, int valu, int prio, int ldsfrag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, grp = tid >> 8, gt = tid & 255;
    // split = 1: even workgroups (CUs) only do the MFMA part with 2x the iterations, odd ones only the copy with 2x the bytes
    if (split) {
        if ((blockIdx.x & 1) != grp) return;
        mfma_iters *= 2;
    }
    if (grp == 0) {
        if (!(mode & 1)) return;
        if (prio == 2) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        u32x4 ra = {0x3f803f80u + tid, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u}, rb = {0x3f803f80u, 0x3f003f00u + tid, 0x3f803f80u, 0x3f803f80u};
        const bf16x8 a = __builtin_bit_cast(bf16x8, ra), b = __builtin_bit_cast(bf16x8, rb);
        if (ldsfrag) {
            // conv-like matrix phase: per 12 MFMAs the wave re-reads 8 x 16 B fragments per lane from LDS (conflict free)
            u32x4* l = reinterpret_cast<u32x4*>(smem) + 1024;           // beyond the copy group's 16 KB
            for (int i = gt; i < 2048; i += 256) l[i] = ra;
            __builtin_amdgcn_s_barrier();
            for (int it = 0; it < mfma_iters / 3; ++it) {
                u32x4 f[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = l[((it + q) & 7) * 256 + gt];
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[(i & 1) + 2 * (t & 1)]),
                                                                           __builtin_bit_cast(bf16x8, f[4 + (i >> 1) + 2 * (t >> 1)]), acc[i], 0, 0, 0);
            }
        } else
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            if (duty) __builtin_amdgcn_s_sleep(2);          // ~128 idle cycles per 128 MFMA cycles
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 12345.678f) sink[tid] = s;
    } else {
        if (ldsfrag && (mode & 1)) __builtin_amdgcn_s_barrier();
        if (!(mode & 2)) return;
        if (prio == 1) __builtin_amdgcn_s_setprio(3);
        const size_t bidx = split ? (blockIdx.x >> 1) : blockIdx.x;
        if (split) vec_per_block *= 2;
        const u32x4* s = src + bidx * vec_per_block;
        u32x4* d = dst + bidx * vec_per_block;
        u32x4* l = reinterpret_cast<u32x4*>(smem);
        for (size_t i = gt; i < vec_per_block; i += 256 * 4) {
            u32x4 v0 = s[i], v1 = s[i + 256], v2 = s[i + 512], v3 = s[i + 768];
            for (int q = 0; q < valu; ++q) {       // VALU-heavy partner: `valu` x 16 dependent-free integer ops per 64 B
                v0 += v1 * 3u; v1 += v2 * 5u; v2 += v3 * 7u; v3 += v0 * 9u;
            }
            if (VIA_LDS) {
                l[gt] = v0; l[gt + 256] = v1; l[gt + 512] = v2; l[gt + 768] = v3;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                v0 = l[(gt + 64) & 255]; v1 = l[256 + ((gt + 64) & 255)]; v2 = l[512 + ((gt + 64) & 255)]; v3 = l[768 + ((gt + 64) & 255)];
            }
            d[i] = v0; d[i + 256] = v1; d[i + 512] = v2; d[i + 768] = v3;
        }
    }
}

int main(int argc, char** argv) {
    const int ncu = 256;
    const size_t vec_per_block = (size_t)(argc > 1 ? atoi(argv[1]) : 16384) * 1024 / 16;      // KB per block
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const size_t bytes = vec_per_block * 16 * ncu;
    u32x4 *src, *dst; float* sink;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 4096);
    hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"thin partner, dense MFMA", "even/odd CUs", "thin partner, MFMA 50% duty", "VALU x8 partner", "VALU x8 partner, partner prio 3",
                           "VALU x8 partner, MFMA prio 3", "VALU x32 partner", "VALU x32 partner, partner prio 3", "MFMA reads LDS frags, thin partner", "MFMA reads LDS frags, partner via LDS"};
    for (int cfg = 0; cfg < 10; ++cfg)
        for (int mode = 1; mode <= 3; ++mode) {
            const int via = cfg == 9, split = cfg == 1, duty = cfg == 2, ldsfrag = cfg >= 8;
            const int valu = cfg >= 8 ? 0 : cfg >= 6 ? 32 : (cfg >= 3 ? 8 : 0), prio = (cfg == 4 || cfg == 7) ? 1 : (cfg == 5 ? 2 : 0);
            if (mode == 1 && cfg >= 3 && cfg != 5 && cfg != 8) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (via) hipLaunchKernelGGL(k<1>, dim3(ncu), dim3(512), 65536, 0, src, dst, vec_per_block, iters, mode, sink, split, duty, valu, prio, ldsfrag);
                else     hipLaunchKernelGGL(k<0>, dim3(ncu), dim3(512), 65536, 0, src, dst, vec_per_block, iters, mode, sink, split, duty, valu, prio, ldsfrag);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double tf = 4.0 * iters * 2.0 * 32 * 32 * 16 * 4 * ncu / (best * 1e-3) / 1e12;
            const double tbs = 2.0 * bytes / (best * 1e-3) / 1e12;
            printf("%-34s mode=%d (%s): %.3f ms   %s%.0f TFLOP/s %s%.2f TB/s (r+w)\n", names[cfg], mode,
                   mode == 1 ? "mfma only" : mode == 2 ? "copy only" : "both", best, (mode & 1) ? "" : "(", (mode & 1) ? tf : 0.0,
                   (mode & 2) ? "" : "(", (mode & 2) ? tbs : 0.0);
        }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
