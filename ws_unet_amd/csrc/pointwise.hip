// Bandwidth-bound kernels of the forward path: first layer (few input planes), 1x1 head + sigmoid,
// stand-alone max-pool, UniformDropout, u8 -> unit float, WS residual statistics.
#include "wsu_device.h"
// No fused multiply-adds in this file: the residual statistics follow numpy's float32 operation sequence (src/unet/evaluate.py:125-132).
// (Until round 3 the SLP vectorizer happened to pack these products into v_pk_mul_f32 / v_pk_add_f32, which cannot fuse; built without it
// (Makefile) hipcc's default -ffp-contract=fast would fuse them.)
#pragma clang fp contract(off)

namespace {

// ---------------------------------------------------------------------------------------------------
// First layer (e11, src/unet/model/unet.py:82,141): 3x3 reflect conv from `cin` NCHW fp32 planes to
// `cout` NHWC channels.  HBM-write bound (1 plane in, 64 channels out): one lane produces 8 consecutive
// output channels of one pixel, so 8 lanes write one pixel's 64 channels as contiguous 16/32-byte pieces.
// ---------------------------------------------------------------------------------------------------
constexpr int FIRST_PASSES = 8;

template <int MODE>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, char* __restrict__ y,
                                                            int n, int h, int wd, int cin, int cout, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);              // [ci*9 + tap][cout]
    const int tid = threadIdx.x;
    for (int i = tid; i < cin * 9 * cout; i += 256) {
        const int co = i % cout, k = i / cout;              // k = ci*9 + tap
        wl[i] = w[(size_t)co * cin * 9 + k];                // OIHW: ((co*cin + ci)*3 + u)*3 + v
    }
    __syncthreads();
    const int ngrp = cout >> 3, ppb = 256 / ngrp;
    const int grp = tid % ngrp, pl = tid / ngrp;
    const long long npix = (long long)n * h * wd;
    float b8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[e] = bias ? bias[grp * 8 + e] : 0.f;
    for (int pass = 0; pass < FIRST_PASSES; ++pass) {
        const long long p = ((long long)blockIdx.x * FIRST_PASSES + pass) * ppb + pl;
        if (p >= npix) break;
        const int xx = (int)(p % wd); const long long t = p / wd;
        const int yy = (int)(t % h); const int nn = (int)(t / h);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = b8[e];
        for (int ci = 0; ci < cin; ++ci) {
            const float* xp = x + ((size_t)nn * cin + ci) * h * wd;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int sy = wsu_reflect(yy + tap / 3 - 1, h), sx = wsu_reflect(xx + tap % 3 - 1, wd);
                const float xv = xp[(size_t)sy * wd + sx];
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + (ci * 9 + tap) * cout + grp * 8);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + (ci * 9 + tap) * cout + grp * 8 + 4);
                acc[0] = fmaf(xv, w0.x, acc[0]); acc[1] = fmaf(xv, w0.y, acc[1]);
                acc[2] = fmaf(xv, w0.z, acc[2]); acc[3] = fmaf(xv, w0.w, acc[3]);
                acc[4] = fmaf(xv, w1.x, acc[4]); acc[5] = fmaf(xv, w1.y, acc[5]);
                acc[6] = fmaf(xv, w1.z, acc[6]); acc[7] = fmaf(xv, w1.w, acc[7]);
            }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
        }
        if constexpr (MODE == WSU_MODE_BF16) {
            *reinterpret_cast<u32x4*>(y + ((size_t)p * cout + grp * 8) * 2) =
                mk_u4(wsu_pack_bf16x2(acc[0], acc[1]), wsu_pack_bf16x2(acc[2], acc[3]),
                           wsu_pack_bf16x2(acc[4], acc[5]), wsu_pack_bf16x2(acc[6], acc[7]));
        } else {
            f32x4* o = reinterpret_cast<f32x4*>(y + ((size_t)p * cout + grp * 8) * 4);
            o[0] = mk_f4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = mk_f4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Head (outconv 1x1 + sigmoid, unet.py:135,189): per pixel a C-wide dot product per output channel.
// One lane per 16-byte granule of the pixel's channel vector, butterfly reduction over the lanes of a
// pixel (16 lanes fp32 / 8 lanes bf16 at C = 64), NCHW fp32 output.
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void conv1x1_sigmoid_kernel(const char* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              float* __restrict__ logit, int n, int hw, int c, int cout) {
    constexpr int EPG = (MODE == WSU_MODE_BF16) ? 8 : 4;     // elements per 16-byte granule
    const int gpp = c / EPG;                                 // lanes per pixel (power of two, <= 64)
    const long long npix = (long long)n * hw;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long p = gid / gpp;
    const int g = (int)(gid % gpp);
    const bool valid = p < npix;
    float xv[EPG];
    if (valid) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(x + ((size_t)p * c + g * EPG) * (16 / EPG));
        if constexpr (MODE == WSU_MODE_BF16) {
            const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = wsu_bf16_to_f32((u[e >> 1] >> ((e & 1) * 16)) & 0xFFFF);
        } else {
            const f32x4 f = __builtin_bit_cast(f32x4, raw);
            xv[0] = f.x; xv[1] = f.y; xv[2] = f.z; xv[3] = f.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPG; ++e) xv[e] = 0.f;
    }
    for (int co = 0; co < cout; ++co) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPG; ++e) s = fmaf(xv[e], w[(size_t)co * c + g * EPG + e], s);
        for (int off = gpp >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (valid && g == 0) {
            const float z = s + (bias ? bias[co] : 0.f);
            const long long nn = p / hw, r = p % hw;
            const size_t o = ((size_t)nn * cout + co) * hw + r;
            if (logit) logit[o] = z;
            out[o] = 1.f / (1.f + expf(-z));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Stand-alone 2x2/2 max-pool (nn.MaxPool2d, unet.py:86,93) with first-max-wins argmax.
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void maxpool_kernel(const char* __restrict__ x, char* __restrict__ y, uint8_t* __restrict__ pidx,
                                                      int n, int h, int w, int c) {
    constexpr int EPG = (MODE == WSU_MODE_BF16) ? 8 : 4;
    constexpr int ESZ = 16 / EPG;
    const int hp = h >> 1, wp = w >> 1, gpp = c / EPG;
    const long long total = (long long)n * hp * wp * gpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % gpp); long long t = i / gpp;
        const int px = (int)(t % wp); t /= wp;
        const int py = (int)(t % hp); const int nn = (int)(t / hp);
        const char* base = x + ((((size_t)nn * h + 2 * py) * w + 2 * px) * c + g * EPG) * ESZ;
        const u32x4 q[4] = {*reinterpret_cast<const u32x4*>(base), *reinterpret_cast<const u32x4*>(base + (size_t)c * ESZ),
                            *reinterpret_cast<const u32x4*>(base + (size_t)w * c * ESZ),
                            *reinterpret_cast<const u32x4*>(base + (size_t)(w + 1) * c * ESZ)};
        float best[EPG]; uint32_t bi[EPG]; uint32_t raw[EPG];
#pragma unroll
        for (int e = 0; e < EPG; ++e) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t* u = reinterpret_cast<const uint32_t*>(&q[k]);
                float v; uint32_t rv;
                if constexpr (MODE == WSU_MODE_BF16) { rv = (u[e >> 1] >> ((e & 1) * 16)) & 0xFFFF; v = wsu_bf16_to_f32(rv); }
                else { rv = u[e]; v = __builtin_bit_cast(float, rv); }
                if (k == 0 || v > best[e] || v != v) { best[e] = v; bi[e] = k; raw[e] = rv; }
            }
        }
        const size_t eo = (((size_t)nn * hp + py) * wp + px) * c + g * EPG;
        if constexpr (MODE == WSU_MODE_BF16) {
            *reinterpret_cast<u32x4*>(y + eo * 2) = mk_u4(raw[0] | raw[1] << 16, raw[2] | raw[3] << 16, raw[4] | raw[5] << 16, raw[6] | raw[7] << 16);
            if (pidx) *reinterpret_cast<u32x2*>(pidx + eo) = mk_u2(bi[0] | bi[1] << 8 | bi[2] << 16 | bi[3] << 24, bi[4] | bi[5] << 8 | bi[6] << 16 | bi[7] << 24);
        } else {
            *reinterpret_cast<u32x4*>(y + eo * 4) = mk_u4(raw[0], raw[1], raw[2], raw[3]);
            if (pidx) *reinterpret_cast<uint32_t*>(pidx + eo) = bi[0] | bi[1] << 8 | bi[2] << 16 | bi[3] << 24;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// UniformDropout (unet.py:32-42), out of place:  y = x*mask + KB(x)*(1-mask) on one plane, copy others.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void uniform_dropout_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              const float* __restrict__ mask, float* __restrict__ mask_out,
                                                              int n, int c, int h, int w, int channel, float keep_prob, uint64_t seed) {
    const long long total = (long long)n * c * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w); long long t = i / w;
        const int yy = (int)(t % h); t /= h;
        const int ch = (int)(t % c); const int nn = (int)(t / c);
        const float v = x[i];
        if (ch != channel) { y[i] = v; continue; }
        const long long mi = ((long long)nn * h + yy) * w + xx;          // mask is (N,1,H,W)
        float m;
        if (mask) m = mask[mi];
        else {
            const uint64_t r = mix64(mix64((uint64_t)mi * 0x9E3779B97F4A7C15ull + seed) + 0x9E3779B97F4A7C15ull);
            m = ((double)(r >> 32) * (1.0 / 4294967296.0) < (double)keep_prob) ? 1.f : 0.f;
        }
        if (mask_out) mask_out[mi] = m;
        const float* p = x + ((size_t)nn * c + ch) * h * w;
        const int ym = wsu_reflect(yy - 1, h), yp = wsu_reflect(yy + 1, h), xm = wsu_reflect(xx - 1, w), xp = wsu_reflect(xx + 1, w);
        // KB kernel [-1 2 -1; 2 0 2; -1 2 -1] / 4  (unet.py:23-27); products are exact, row-major accumulation
        float kb = -0.25f * p[(size_t)ym * w + xm];
        kb += 0.5f * p[(size_t)ym * w + xx];
        kb += -0.25f * p[(size_t)ym * w + xp];
        kb += 0.5f * p[(size_t)yy * w + xm];
        kb += 0.5f * p[(size_t)yy * w + xp];
        kb += -0.25f * p[(size_t)yp * w + xm];
        kb += 0.5f * p[(size_t)yp * w + xx];
        kb += -0.25f * p[(size_t)yp * w + xp];
        y[i] = v * m + kb * (1.f - m);
    }
}

__global__ __launch_bounds__(256) void u8_to_unit_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        y[i] = (float)x[i] / 255.0f;          // IEEE division, as numpy's x / 255. in float32 (evaluate.py:45)
}

// ---------------------------------------------------------------------------------------------------
// WS residual statistics (src/unet/evaluate.py:125-132).  One workgroup per image, fp64 accumulation in
// a fixed order (strided per-thread sums, then a binary LDS tree) -> bitwise reproducible.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ws_stats_kernel(const uint8_t* __restrict__ xu8, const float* __restrict__ y01,
                                                        float* __restrict__ beta_hat, float* __restrict__ l1, int h, int w) {
    __shared__ double sb[1024];
    __shared__ double sl[1024];
    const int nn = blockIdx.x, tid = threadIdx.x;
    const int ih = h - 2, iw = w - 2;
    const long long cnt = (long long)ih * iw;
    double ab = 0.0, al = 0.0;
    for (long long i = tid; i < cnt; i += 1024) {
        const int r = (int)(i / iw) + 1, c = (int)(i % iw) + 1;
        const size_t o = ((size_t)nn * h + r) * w + c;
        const uint8_t u = xu8[o];
        const float xf = (float)u;
        const float xbar = (float)(uint8_t)(u ^ 1);
        const float xhat = y01[o] * 255.0f;                  // evaluate.py:51
        const float d = xf - xhat;                           // float32 like numpy
        ab += (double)((xf - xbar) * d);
        al += (double)fabsf(d);
    }
    sb[tid] = ab; sl[tid] = al;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) { sb[tid] += sb[tid + s]; sl[tid] += sl[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        beta_hat[nn] = (float)(sb[0] / (double)cnt);
        l1[nn] = (float)(sl[0] / (double)cnt);
    }
}

// Epoch meter of the training loop, src/_defs/metrics.py:122-142 `WSMeter.update`: per-image beta_hat on the [1:-1,1:-1] interior from
// the float inputs (xi = x*255 in float32, x_bar = int(round(xi)) ^ 1, the product in float64 like numpy's float32 - int64 promotion).
__global__ __launch_bounds__(1024) void ws_meter_kernel(const float* __restrict__ x01, const float* __restrict__ y01,
                                                        double* __restrict__ beta_hat, int h, int w) {
    __shared__ double sb[1024];
    const int nn = blockIdx.x, tid = threadIdx.x;
    const int ih = h - 2, iw = w - 2;
    const long long cnt = (long long)ih * iw;
    double ab = 0.0;
    for (long long i = tid; i < cnt; i += 1024) {
        const int r = (int)(i / iw) + 1, c = (int)(i % iw) + 1;
        const size_t o = ((size_t)nn * h + r) * w + c;
        const float xi = __fmul_rn(x01[o], 255.0f), xh = __fmul_rn(y01[o], 255.0f);
        const long long xbar = (long long)rintf(xi) ^ 1LL;                       // np.round = half-to-even = rintf
        ab += ((double)xi - (double)xbar) * (double)__fsub_rn(xi, xh) / (double)cnt;
    }
    sb[tid] = ab;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) sb[tid] += sb[tid + s];
        __syncthreads();
    }
    if (tid == 0) beta_hat[nn] = sb[0];
}

}  // namespace

extern "C" {

int wsu_conv3x3_first_fwd(const float* x_nchw, const float* w_oihw, const float* bias, void* y,
                          int n, int h, int w, int cin, int cout, int mode, int relu, void* stream) {
    WSU_REQUIRE(mode >= 0 && mode <= 2, "conv3x3_first: bad mode %d", mode);
    WSU_REQUIRE(x_nchw && w_oihw && y, "conv3x3_first: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_first: bad shape n=%d h=%d w=%d (reflect pad 1 needs h,w >= 2)", n, h, w);
    WSU_REQUIRE(cin >= 1 && cin <= 8, "conv3x3_first: cin=%d outside 1..8", cin);
    WSU_REQUIRE(cout >= 8 && cout % 8 == 0 && 256 % (cout / 8) == 0, "conv3x3_first: cout=%d unsupported", cout);
    const int ppb = 256 / (cout / 8) * FIRST_PASSES;
    const long long npix = (long long)n * h * w;
    const long long nblk = (npix + ppb - 1) / ppb;
    WSU_REQUIRE(nblk < 0x7FFFFFFFLL, "conv3x3_first: grid too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)cin * 9 * cout * sizeof(float);
    if (mode == WSU_MODE_BF16)
        hipLaunchKernelGGL(conv3x3_first_kernel<WSU_MODE_BF16>, dim3((unsigned)nblk), dim3(256), lds, s, x_nchw, w_oihw, bias, (char*)y, n, h, w, cin, cout, relu);
    else
        hipLaunchKernelGGL(conv3x3_first_kernel<WSU_MODE_F32>, dim3((unsigned)nblk), dim3(256), lds, s, x_nchw, w_oihw, bias, (char*)y, n, h, w, cin, cout, relu);
    return wsu_check_launch("conv3x3_first_kernel");
}

int wsu_conv1x1_sigmoid_fwd(const void* x, const float* w, const float* bias, float* out, float* logit,
                            int n, int h, int w_, int c, int cout, int mode, void* stream) {
    WSU_REQUIRE(mode >= 0 && mode <= 2, "conv1x1_sigmoid: bad mode %d", mode);
    WSU_REQUIRE(x && w && out, "conv1x1_sigmoid: null pointer");
    WSU_REQUIRE(n > 0 && h > 0 && w_ > 0 && cout > 0, "conv1x1_sigmoid: bad shape");
    const int epg = mode == WSU_MODE_BF16 ? 8 : 4;
    const int gpp = c / epg;
    WSU_REQUIRE(c % epg == 0 && gpp >= 1 && gpp <= 64 && (gpp & (gpp - 1)) == 0, "conv1x1_sigmoid: c=%d must give a power-of-two lane group <= 64", c);
    const long long threads = (long long)n * h * w_ * gpp;
    const long long nblk = (threads + 255) / 256;
    WSU_REQUIRE(nblk < 0x7FFFFFFFLL, "conv1x1_sigmoid: grid too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_BF16)
        hipLaunchKernelGGL(conv1x1_sigmoid_kernel<WSU_MODE_BF16>, dim3((unsigned)nblk), dim3(256), 0, s, (const char*)x, w, bias, out, logit, n, h * w_, c, cout);
    else
        hipLaunchKernelGGL(conv1x1_sigmoid_kernel<WSU_MODE_F32>, dim3((unsigned)nblk), dim3(256), 0, s, (const char*)x, w, bias, out, logit, n, h * w_, c, cout);
    return wsu_check_launch("conv1x1_sigmoid_kernel");
}

int wsu_maxpool2x2_fwd(const void* x, void* y, uint8_t* pool_idx, int n, int h, int w, int c, int mode, void* stream) {
    WSU_REQUIRE(mode >= 0 && mode <= 2, "maxpool2x2: bad mode %d", mode);
    WSU_REQUIRE(x && y, "maxpool2x2: null pointer");
    const int epg = mode == WSU_MODE_BF16 ? 8 : 4;
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2 && c > 0 && c % epg == 0, "maxpool2x2: bad shape n=%d h=%d w=%d c=%d", n, h, w, c);
    const long long total = (long long)n * (h / 2) * (w / 2) * (c / epg);
    const unsigned nblk = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_BF16) hipLaunchKernelGGL(maxpool_kernel<WSU_MODE_BF16>, dim3(nblk), dim3(256), 0, s, (const char*)x, (char*)y, pool_idx, n, h, w, c);
    else hipLaunchKernelGGL(maxpool_kernel<WSU_MODE_F32>, dim3(nblk), dim3(256), 0, s, (const char*)x, (char*)y, pool_idx, n, h, w, c);
    return wsu_check_launch("maxpool_kernel");
}

int wsu_uniform_dropout_fwd(const float* x, float* y, const float* mask, float* mask_out,
                            int n, int c, int h, int w, int channel, float keep_prob, uint64_t seed, void* stream) {
    WSU_REQUIRE(x && y && x != y, "uniform_dropout: null or aliased pointers (kernel is out of place)");
    WSU_REQUIRE(n > 0 && c > 0 && h >= 2 && w >= 2 && channel >= 0 && channel < c, "uniform_dropout: bad shape");
    const long long total = (long long)n * c * h * w;
    const unsigned nblk = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(uniform_dropout_kernel, dim3(nblk), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, mask, mask_out, n, c, h, w, channel, keep_prob, seed);
    return wsu_check_launch("uniform_dropout_kernel");
}

int wsu_u8_to_unit_f32(const uint8_t* x, float* y, size_t count, void* stream) {
    WSU_REQUIRE(x && y, "u8_to_unit_f32: null pointer");
    if (count == 0) return WSU_OK;
    const unsigned nblk = (unsigned)((count + 255) / 256 < 16384 ? (count + 255) / 256 : 16384);
    hipLaunchKernelGGL(u8_to_unit_kernel, dim3(nblk), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, count);
    return wsu_check_launch("u8_to_unit_kernel");
}

int wsu_ws_residual_stats(const uint8_t* x_u8, const float* y01, float* beta_hat, float* l1, int n, int h, int w, void* stream) {
    WSU_REQUIRE(x_u8 && y01 && beta_hat && l1, "ws_residual_stats: null pointer");
    WSU_REQUIRE(n > 0 && h >= 3 && w >= 3, "ws_residual_stats: bad shape n=%d h=%d w=%d", n, h, w);
    hipLaunchKernelGGL(ws_stats_kernel, dim3(n), dim3(1024), 0, static_cast<hipStream_t>(stream), x_u8, y01, beta_hat, l1, h, w);
    return wsu_check_launch("ws_stats_kernel");
}

int wsu_ws_meter_beta(const float* x01, const float* y01, double* beta_hat, int n, int h, int w, void* stream) {
    WSU_REQUIRE(x01 && y01 && beta_hat, "ws_meter_beta: null pointer");
    WSU_REQUIRE(n > 0 && h >= 3 && w >= 3, "ws_meter_beta: bad shape n=%d h=%d w=%d", n, h, w);
    hipLaunchKernelGGL(ws_meter_kernel, dim3(n), dim3(1024), 0, static_cast<hipStream_t>(stream), x01, y01, beta_hat, h, w);
    return wsu_check_launch("ws_meter_kernel");
}

}  // extern "C"
