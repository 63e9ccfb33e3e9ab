// Checks the operand / scale layout assumed for v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3) on gfx950 against a host loop:
//   lane (l31 = lane & 31, hh = lane >> 5) supplies A[row l31][k] for k = 16*hh + j (operand registers 0-3, byte j) and k = 32 + 16*hh + j
//   (registers 4-7), B[k][col l31] likewise: scale block 0 (k < 32) is registers 0-3 of ALL lanes, block 1 registers 4-7 of all lanes.
//   Byte 0 (op_sel 0) of the scale register of lane l31 scales block 0 of row / column l31, that of lane 32 + l31 block 1 (E8M0, 2^(e-127)).
//   (Measured with one-hot operands and single-lane scale changes; a [32 contiguous k per lane] layout fails this check.)
//   C/D map = the 32x32 one: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * hh for accumulator register r.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_scale_layout_probe tools/mfma_scale_layout_probe.hip && tools/mfma_scale_layout_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(const uint8_t* A, const uint8_t* Bt, const uint8_t* sA, const uint8_t* sB, float* C, const float* Cin) {
    const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) {
        const int k0 = (i >> 2) * 32 + 16 * hh + 4 * (i & 3);
        a[i] = *reinterpret_cast<const int*>(A + l31 * 64 + k0);
        b[i] = *reinterpret_cast<const int*>(Bt + l31 * 64 + k0);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = Cin[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + l31];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, (int)sA[l31 * 2 + hh] | 0x55aa3300, 0, (int)sB[l31 * 2 + hh] | 0x11ee7700);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + l31] = c[r];
}

static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}

int main() {
    uint32_t st = 7u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 9; };
    uint8_t *dA, *dB, *dsA, *dsB; float *dC, *dCin;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsA, 64); hipMalloc(&dsB, 64); hipMalloc(&dC, 4096); hipMalloc(&dCin, 4096);
    int bad = 0;
    for (int cas = 0; cas < 5; ++cas) {
        // 0: unit scales, normal values only   1: unit scales, subnormals too   2: random scales   3: random scales + accumulator input ~ the dot
        // 4: small dot (scales 2^-12) onto an accumulator of O(100): the use in the conv kernels
        std::vector<uint8_t> A(2048), Bt(2048), sA(64), sB(64); std::vector<float> Cin(1024, 0.f), C(1024);
        for (auto* v : {&A, &Bt}) for (auto& x : *v) { x = rnd() & 0xff; if ((x & 0x7f) == 0x7f) x ^= 1; if (cas == 0 && (x & 0x78) == 0) x |= 0x20; }
        for (auto& v : sA) v = cas < 2 ? 127 : (cas == 4 ? 115 : 120 + rnd() % 12);
        for (auto& v : sB) v = cas < 2 ? 127 : (cas == 4 ? 121 : 115 + rnd() % 20);
        if (cas == 3) for (auto& v : Cin) v = ((int)(rnd() % 20001) - 10000) * 3.7f;
        if (cas == 4) for (auto& v : Cin) v = (100.f + (rnd() % 10000) * 0.01f) * ((rnd() & 1) ? 1.f : -1.f) + 1e-3f * (rnd() % 1000);
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsA, sA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsB, sB.data(), 64, hipMemcpyHostToDevice);
        hipMemcpy(dCin, Cin.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsA, dsB, dC, dCin);
        if (hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
        double worst = 0, worst_ulp = 0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double ref = Cin[r * 32 + c], mag = fabs(ref), dotmag = 0;
                for (int kk = 0; kk < 64; ++kk) {
                    const double t = (double)e4m3(A[r * 64 + kk]) * e4m3(Bt[c * 64 + kk]) * ldexp(1.0, sA[r * 2 + kk / 32] - 127) * ldexp(1.0, sB[c * 2 + kk / 32] - 127);
                    ref += t; mag += fabs(t); dotmag += fabs(t);
                }
                const double err = fabs(ref - C[r * 32 + c]);
                if (err / (mag + 1e-30) > worst) worst = err / (mag + 1e-30);
                // budget: one ulp of the accumulator plus 2^-12 of the dot's own terms
                const double ulp = ldexp(1.0, ilogb(fabs((double)Cin[r * 32 + c]) + 1e-300) - 23) + dotmag * ldexp(1.0, -12);
                if (err / ulp > worst_ulp) worst_ulp = err / ulp;
            }
        printf("case %d: worst |gpu - host| / sum|terms| = %.3g, worst error / (ulp(acc_in) + 2^-12 sum|dot terms|) = %.3g\n", cas, worst, worst_ulp);
        if (worst_ulp > 2.0) bad = 1;
    }
    printf("layout %s\n", bad ? "WRONG" : "CONFIRMED");
    return bad;
}
