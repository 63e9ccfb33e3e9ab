// K11: the weighted-stego (WS) payload estimate of the caller of the predictor, src/ws/estimate.py:55-136 `attack`:
//
//   x      = cover/stego plane (uint8 values),   x_bar = x ^ 1,    interior = [1:-1, 1:-1]
//   x_hat  = pixel_estimator(x)                                       (:89)  UNet output*255, a host array, or a 3x3 filter
//   mu     = convolve(x,   mean_estimator, 'valid');  mu2 = convolve(x*x, mean_estimator, 'valid')          (:93-94)
//   var    = mu2 - mu^2;   weights = 1/(5+var) (weighted=1) | 5+var (weighted=-1) | 1 (weighted=0)          (:95-110)
//   weights /= sum(weights)
//   beta   = clip( sum(weights * (x - x_bar) * (x - x_hat)), 0, None )                                       (:118-121)
//   if correct_bias:  beta -= beta * sum(weights * (x - x_bar) * pixel_estimator(x_bar - x))                 (:126-128)
//
// Everything per pixel is float32 like the numpy original (separately rounded mul / sub, no contraction); the three sums
// are fp64 in a fixed order (per-thread strided, LDS tree, 64 partial blocks per image, second tree), so a result is
// bitwise reproducible.  With the reference's default mean estimator (AVG, 8 taps of 1/8) mu, mu2 and var are exact in
// float32 for uint8 pixels, whatever the order of the nine products.
// HBM-bound: 1 B (pixel, neighbours hit L1/L2) + 4 B (prediction) [+ 4 B bias prediction] per pixel.
#include "wsu_device.h"
// No fused multiply-adds in this file: the WS estimator follows numpy's float32 operation sequence (src/ws/estimate.py:90-121).
// (Until round 3 the SLP vectorizer happened to pack these products into v_pk_mul_f32 / v_pk_add_f32, which cannot fuse; built without it
// (Makefile) hipcc's default -ffp-contract=fast would fuse them.)
#pragma clang fp contract(off)

namespace {

constexpr int WSA_PARTS = 64;                           // partial blocks per image
struct Taps { float k[9]; };                            // K[a][b] of the (3,3,1) numpy kernel, a = row tap, b = column tap

// true convolution, 'valid': out(r,c) = sum_{a,b} K[a][b] * v(r+1-a, c+1-b); v[i][j] holds the 3x3 neighbourhood, i,j = 0..2
__device__ __forceinline__ float conv9(const Taps& t, const float v[3][3]) {
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc = __fadd_rn(acc, __fmul_rn(t.k[a * 3 + b], v[2 - a][2 - b]));
    return acc;
}

__global__ __launch_bounds__(256) void ws_attack_partial_kernel(
    const uint8_t* __restrict__ xu8, const float* __restrict__ xhat, const float* __restrict__ xbias,
    Taps mean_taps, Taps pixel_taps, int use_pixel_filter, int hat_full, float hat_scale, int weighted, int correct_bias,
    double* __restrict__ partial, int h, int w) {
    __shared__ double red[3][256];
    const int nn = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const uint8_t* img = xu8 + (size_t)nn * h * w;
    const int ih = h - 2, iw = w - 2;
    const size_t hat_base = hat_full ? (size_t)nn * h * w : (size_t)nn * ih * iw;
    double sw = 0.0, sb = 0.0, sc = 0.0;
    for (int r = 1 + part; r <= h - 2; r += WSA_PARTS) {
        for (int c = 1 + tid; c <= w - 2; c += 256) {
            float v[3][3], v2[3][3];
            uint8_t u[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    u[i][j] = img[(size_t)(r - 1 + i) * w + (c - 1 + j)];
                    v[i][j] = (float)u[i][j];
                    v2[i][j] = __fmul_rn(v[i][j], v[i][j]);
                }
            float wgt = 1.0f;
            if (weighted != 0) {
                const float mu = conv9(mean_taps, v);
                const float mu2 = conv9(mean_taps, v2);
                const float var = __fsub_rn(mu2, __fmul_rn(mu, mu));
                const float t = __fadd_rn(5.0f, var);
                wgt = weighted > 0 ? __fdiv_rn(1.0f, t) : t;
            }
            const float x = v[1][1];
            const float s = __fsub_rn(x, (float)(uint8_t)(u[1][1] ^ 1));             // x - x_bar = +-1
            float hat, bias = 0.f;
            if (use_pixel_filter) {
                float q[3][3], qb[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        q[i][j] = __fdiv_rn(v[i][j], 255.0f);                           // filters/evaluate.py:136-141
                        qb[i][j] = __fdiv_rn(__fsub_rn((float)(uint8_t)(u[i][j] ^ 1), v[i][j]), 255.0f);
                    }
                hat = __fmul_rn(conv9(pixel_taps, q), 255.0f);
                if (correct_bias) bias = __fmul_rn(conv9(pixel_taps, qb), 255.0f);
            } else {
                const size_t o = hat_full ? hat_base + (size_t)r * w + c : hat_base + (size_t)(r - 1) * iw + (c - 1);
                hat = __fmul_rn(xhat[o], hat_scale);
                if (correct_bias) bias = __fmul_rn(xbias[o], hat_scale);
            }
            const float ws = __fmul_rn(wgt, s);
            sw += (double)wgt;
            sb += (double)__fmul_rn(ws, __fsub_rn(x, hat));
            sc += (double)__fmul_rn(ws, bias);
        }
    }
    red[0][tid] = sw; red[1][tid] = sb; red[2][tid] = sc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; red[2][tid] += red[2][tid + st];
        }
        __syncthreads();
    }
    if (tid < 3) partial[((size_t)nn * WSA_PARTS + part) * 3 + tid] = red[tid][0];
}

__global__ __launch_bounds__(64) void ws_attack_finish_kernel(const double* __restrict__ partial, float* __restrict__ beta_hat,
                                                              double* __restrict__ sums, int correct_bias) {
    __shared__ double red[3][WSA_PARTS];
    const int nn = blockIdx.x, tid = threadIdx.x;
    for (int k = 0; k < 3; ++k) red[k][tid] = partial[((size_t)nn * WSA_PARTS + tid) * 3 + k];
    __syncthreads();
    for (int st = WSA_PARTS / 2; st > 0; st >>= 1) {
        if (tid < st) for (int k = 0; k < 3; ++k) red[k][tid] += red[k][tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        const double sw = red[0][0], sb = red[1][0], sc = red[2][0];
        double beta = sb / sw;
        beta = beta > 0.0 ? beta : 0.0;                                               // np.clip(beta_hat, 0, None)
        if (correct_bias) beta -= beta * (sc / sw);
        beta_hat[nn] = (float)beta;
        if (sums) { sums[nn * 3 + 0] = sw; sums[nn * 3 + 1] = sb; sums[nn * 3 + 2] = sc; }
    }
}

__global__ __launch_bounds__(256) void lsb_delta_unit_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const uint8_t u = x[i];
        y[i] = __fdiv_rn(__fsub_rn((float)(uint8_t)(u ^ 1), (float)u), 255.0f);        // (x_bar - x) / 255.
    }
}

// filters/evaluate.py:136-141 `infere_single`: convolve(x / 255., K, 'valid') * 255. on one fp32 plane
__global__ __launch_bounds__(256) void filter3x3_valid_kernel(const float* __restrict__ x, Taps taps, float* __restrict__ y,
                                                              int n, int h, int w) {
    const int ih = h - 2, iw = w - 2;
    const long long total = (long long)n * ih * iw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % iw), r = (int)((i / iw) % ih), nn = (int)(i / ((long long)iw * ih));
        const float* p = x + ((size_t)nn * h + r) * w + c;
        float q[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) q[a][b] = __fdiv_rn(p[(size_t)a * w + b], 255.0f);
        y[i] = __fmul_rn(conv9(taps, q), 255.0f);
    }
}

}  // namespace

extern "C" {

size_t wsu_ws_attack_workspace_bytes(int n) { return (size_t)n * WSA_PARTS * 3 * sizeof(double); }

int wsu_ws_attack(const uint8_t* x_u8, const float* x_hat, const float* x_bias, const float* pixel_filter, const float* mean_filter,
                  int hat_full, float hat_scale, int weighted, int correct_bias, float* beta_hat, double* sums,
                  void* workspace, size_t workspace_bytes, int n, int h, int w, void* stream) {
    WSU_REQUIRE(x_u8 && beta_hat && workspace, "ws_attack: null pointer");
    WSU_REQUIRE((x_hat != nullptr) != (pixel_filter != nullptr), "ws_attack: give exactly one of x_hat / pixel_filter");
    WSU_REQUIRE(weighted >= -1 && weighted <= 1, "ws_attack: weighted=%d outside {-1,0,1}", weighted);
    WSU_REQUIRE(weighted == 0 || mean_filter, "ws_attack: weighted estimate needs mean_filter");
    WSU_REQUIRE(!correct_bias || pixel_filter || x_bias, "ws_attack: correct_bias needs x_bias = pixel_estimator(x_bar - x)");
    WSU_REQUIRE(n > 0 && n <= 65535 && h >= 3 && w >= 3, "ws_attack: bad shape n=%d h=%d w=%d", n, h, w);
    WSU_REQUIRE(workspace_bytes >= wsu_ws_attack_workspace_bytes(n), "ws_attack: workspace too small");
    Taps mt{}, pt{};
    for (int i = 0; i < 9; ++i) { mt.k[i] = mean_filter ? mean_filter[i] : 0.f; pt.k[i] = pixel_filter ? pixel_filter[i] : 0.f; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(workspace);
    hipLaunchKernelGGL(ws_attack_partial_kernel, dim3(WSA_PARTS, n), dim3(256), 0, s, x_u8, x_hat, x_bias, mt, pt,
                       pixel_filter ? 1 : 0, hat_full, hat_scale, weighted, correct_bias, partial, h, w);
    int rc = wsu_check_launch("ws_attack_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(ws_attack_finish_kernel, dim3(n), dim3(WSA_PARTS), 0, s, partial, beta_hat, sums, correct_bias);
    return wsu_check_launch("ws_attack_finish_kernel");
}

int wsu_filter3x3_valid_f32(const float* x, const float* filter, float* y, int n, int h, int w, void* stream) {
    WSU_REQUIRE(x && filter && y, "filter3x3_valid: null pointer");
    WSU_REQUIRE(n > 0 && h >= 3 && w >= 3, "filter3x3_valid: bad shape n=%d h=%d w=%d", n, h, w);
    Taps t{};
    for (int i = 0; i < 9; ++i) t.k[i] = filter[i];
    const long long total = (long long)n * (h - 2) * (w - 2);
    const long long nblk = (total + 255) / 256;
    hipLaunchKernelGGL(filter3x3_valid_kernel, dim3((unsigned)(nblk < 65536 ? nblk : 65536)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, t, y, n, h, w);
    return wsu_check_launch("filter3x3_valid_kernel");
}

int wsu_lsb_delta_unit_f32(const uint8_t* x, float* y, size_t count, void* stream) {
    WSU_REQUIRE(x && y, "lsb_delta_unit: null pointer");
    if (count == 0) return 0;
    const size_t nblk = (count + 255) / 256;
    hipLaunchKernelGGL(lsb_delta_unit_kernel, dim3((unsigned)(nblk < 65536 ? nblk : 65536)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, count);
    return wsu_check_launch("lsb_delta_unit_kernel");
}

}  // extern "C"
