// K7p: the pieces of the PLANAR training path (mode F16F8P activations, gradients in the same three-plane layout with the gradient's
// residual scaling) that are not the matrix kernels themselves -- autograd of src/unet/model/unet.py:137-189 (loop pattern
// src/detector/train.py:55-95; oracle: oracle/unet_ref.py under torch.autograd):
//   * reflect-padding adjoint of the 3x3 data gradient: gather of the gradient's border rows / columns into strips, and the fold of the
//     strips' 1x3 convolutions (computed by conv3x3_pl_kernel<GRAD> with the ring weight sets) back onto rows 1, H-2 / columns 1, W-2
//   * (further down) max-pool backward, head backward, first-layer weight gradient, bias sums
// Planar layout (include/wsu.h): [n][C/16][3 planes][H][W][16 B], planes = f16 ch 0-7 | f16 ch 8-15 | e4m3 residuals ch 0-15; activations
// scale the residual by 2^12, gradients by 2^14 (wsu_device.h).  Everything is deterministic (fixed-order reductions, no float atomics).
#include "wsu_device.h"

namespace {

__device__ __forceinline__ size_t pl_off(int n, int nch, int ch, int plane, size_t hw, size_t pix) {
    return ((((size_t)n * nch + ch) * 3 + plane) * hw + pix) * 16;
}

// 3 stored granules of one pixel -> 16 fp32 values (f16 part + e4m3 residual * lo_mul)
__device__ __forceinline__ void pl_decode16(const u32x4& h0, const u32x4& h1, const u32x4& lo, float lo_mul, float (&v)[16]) {
    const f16x8 a = __builtin_bit_cast(f16x8, h0), b = __builtin_bit_cast(f16x8, h1);
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = (float)a[e]; v[8 + e] = (float)b[e]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int wd = (int)lo[k];
        v[4 * k + 0] += __builtin_amdgcn_cvt_f32_fp8(wd, 0) * lo_mul;
        v[4 * k + 1] += __builtin_amdgcn_cvt_f32_fp8(wd, 1) * lo_mul;
        v[4 * k + 2] += __builtin_amdgcn_cvt_f32_fp8(wd, 2) * lo_mul;
        v[4 * k + 3] += __builtin_amdgcn_cvt_f32_fp8(wd, 3) * lo_mul;
    }
}
__device__ __forceinline__ void pl_encode16(const float (&v)[16], float div_lo, u32x4& h0, u32x4& h1, u32x4& lo) {
    uint32_t h[8], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wsu_split4_f16r8(mk_f4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]), div_lo, h[2 * k], h[2 * k + 1], l[k]);
    h0 = mk_u4(h[0], h[1], h[2], h[3]); h1 = mk_u4(h[4], h[5], h[6], h[7]); lo = mk_u4(l[0], l[1], l[2], l[3]);
}
// 8 mask bits (stored f16 > 0) of one granule
__device__ __forceinline__ unsigned pl_pos_bits(const u32x4& g) {
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int wd = (int)g[e];
        bits |= ((short)(wd & 0xFFFF) > 0 ? 1u : 0u) << (2 * e);
        bits |= (wd >= 0x10000 ? 1u : 0u) << (2 * e + 1);
    }
    return bits;
}

// ---- ring strips: [4 sets: top, bottom, left, right][nch][3][n rows][L + 2][16 B]; data at columns 1..W (1..H), zeros elsewhere ----------
__global__ __launch_bounds__(256) void ring_gather_pl_kernel(const char* __restrict__ g, char* __restrict__ strips,
                                                             int n, int h, int w, int nch, int L) {
    const size_t hw = (size_t)h * w;
    const long long total = (long long)4 * nch * 3 * n * (L + 2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long t = i;
        const int j = (int)(t % (L + 2)); t /= (L + 2);
        const int r = (int)(t % n); t /= n;
        const int plane = (int)(t % 3); t /= 3;
        const int ch = (int)(t % nch); const int set = (int)(t / nch);
        u32x4 v = mk_u4(0, 0, 0, 0);
        const int len = set < 2 ? w : h;
        if (j >= 1 && j <= len) {
            const int yy = set == 0 ? 0 : set == 1 ? h - 1 : j - 1;
            const int xx = set == 2 ? 0 : set == 3 ? w - 1 : j - 1;
            v = *reinterpret_cast<const u32x4*>(g + pl_off(r, nch, ch, plane, hw, (size_t)yy * w + xx));
        }
        *reinterpret_cast<u32x4*>(strips + (size_t)i * 16) = v;
    }
}

// so: the strips' conv outputs [4][nci][3][n][L + 2][16 B] (gradient encoding).  One thread per (image, ring pixel, 16-channel chunk).
__global__ __launch_bounds__(256) void ring_fold_pl_kernel(const char* __restrict__ so, char* __restrict__ dx1, char* __restrict__ dx2,
                                                           const char* __restrict__ mask1, const char* __restrict__ mask2,
                                                           int n, int h, int w, int nci, int nco1, int L) {
    const int nrows = (h - 2 != 1) ? 2 : 1, ncols = (w - 2 != 1) ? 2 : 1;
    const int rlo = min(1, h - 2), rhi = max(1, h - 2);
    const int per_img = nrows * w + ncols * (h - nrows);
    const size_t hw = (size_t)h * w, shw = (size_t)n * (L + 2);
    const long long total = (long long)n * per_img * nci;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long t = i;
        int b = (int)(t % per_img); t /= per_img;
        const int ch = (int)(t % nci); const int img = (int)(t / nci);
        int y, x;
        if (b < nrows * w) { y = (b / w == 0) ? 1 : h - 2; x = b % w; }
        else {
            b -= nrows * w;
            int row = b / ncols;
            if (row >= rlo) ++row;
            if (nrows == 2 && row >= rhi) ++row;
            y = row; x = (b % ncols == 0) ? 1 : w - 2;
        }
        const bool d1 = ch < nco1;
        char* dx = d1 ? dx1 : dx2;
        const char* mk = d1 ? mask1 : mask2;
        const int ncd = d1 ? nco1 : nci - nco1, chd = d1 ? ch : ch - nco1;
        const size_t pix = (size_t)y * w + x;
        float v[16];
        pl_decode16(*reinterpret_cast<const u32x4*>(dx + pl_off(img, ncd, chd, 0, hw, pix)), *reinterpret_cast<const u32x4*>(dx + pl_off(img, ncd, chd, 1, hw, pix)),
                    *reinterpret_cast<const u32x4*>(dx + pl_off(img, ncd, chd, 2, hw, pix)), WSU_F8_GLO_DIV, v);
        float add[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) add[e] = 0.f;
        auto take = [&](int set, int col) {                                  // strips image `set`, row `img`, column `col`
            const size_t p = (size_t)img * (L + 2) + col;
            float s[16];
            pl_decode16(*reinterpret_cast<const u32x4*>(so + pl_off(set, nci, ch, 0, shw, p)), *reinterpret_cast<const u32x4*>(so + pl_off(set, nci, ch, 1, shw, p)),
                        *reinterpret_cast<const u32x4*>(so + pl_off(set, nci, ch, 2, shw, p)), WSU_F8_GLO_DIV, s);
#pragma unroll
            for (int e = 0; e < 16; ++e) add[e] += s[e];
        };
        // fixed order: top (centre, left corner, right corner), bottom (same), left, right
        if (y == 1)     { take(0, x + 1); if (x == 1) take(0, 0); if (x == w - 2) take(0, w + 1); }
        if (y == h - 2) { take(1, x + 1); if (x == 1) take(1, 0); if (x == w - 2) take(1, w + 1); }
        if (x == 1) take(2, y + 1);
        if (x == w - 2) take(3, y + 1);
        unsigned m0 = 0xFFu, m1 = 0xFFu;
        if (mk) {
            m0 = pl_pos_bits(*reinterpret_cast<const u32x4*>(mk + pl_off(img, ncd, chd, 0, hw, pix)));
            m1 = pl_pos_bits(*reinterpret_cast<const u32x4*>(mk + pl_off(img, ncd, chd, 1, hw, pix)));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] += ((m0 >> e) & 1u) ? add[e] : 0.f;
            v[8 + e] += ((m1 >> e) & 1u) ? add[8 + e] : 0.f;
        }
        u32x4 h0, h1, lo;
        pl_encode16(v, WSU_F8_GLO_DIV, h0, h1, lo);
        *reinterpret_cast<u32x4*>(dx + pl_off(img, ncd, chd, 0, hw, pix)) = h0;
        *reinterpret_cast<u32x4*>(dx + pl_off(img, ncd, chd, 1, hw, pix)) = h1;
        *reinterpret_cast<u32x4*>(dx + pl_off(img, ncd, chd, 2, hw, pix)) = lo;
    }
}

inline unsigned grid_for(long long total) { return (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

}  // namespace

// internal (conv3x3_pl.hip): the two thin kernels around the strips' conv
extern "C" int wsu_ring_gather_pl(const void* g, void* strips, int n, int h, int w, int c, int L, void* stream) {
    const long long total = (long long)4 * (c / 16) * 3 * n * (L + 2);
    hipLaunchKernelGGL(ring_gather_pl_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), (const char*)g, (char*)strips, n, h, w, c / 16, L);
    return wsu_check_launch("ring_gather_pl_kernel");
}
extern "C" int wsu_ring_fold_pl(const void* strips_out, void* dx1, void* dx2, const void* mask1, const void* mask2,
                                int n, int h, int w, int cin, int csplit, int L, void* stream) {
    const int nrows = (h - 2 != 1) ? 2 : 1, ncols = (w - 2 != 1) ? 2 : 1;
    const long long total = (long long)n * (nrows * w + ncols * (h - nrows)) * (cin / 16);
    hipLaunchKernelGGL(ring_fold_pl_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), (const char*)strips_out, (char*)dx1, (char*)dx2,
                       (const char*)mask1, (const char*)mask2, n, h, w, cin / 16, csplit / 16, L);
    return wsu_check_launch("ring_fold_pl_kernel");
}
