// K8 loss (+ its gradient) and K9 optimiser of the reconstructed UNet train step.
//
// Loss: src/_defs/losses.py -- L1Loss :28-36, WSLoss :45-90, L1WSLoss :93-121 (sum of the two, loss_lambda unused):
//   L1  = mean |covers - out|            (use_l1 = 1)      L2 = mean (covers - out)^2   (use_l1 = 2, L2Loss :39-42)
//   WS  = mean_n | relu(beta_hat_n) - alpha_n / 2 |,
//         beta_hat_n = sum_p (1/(C*H*W)) * (in255 - flip(in255)) * (in255 - out255),  in255 = in*255, out255 = out*255,
//         flip(v) = float(int(rint(v)) ^ 1)                       (integer LSB flip, bit-exact)
// Optimiser: torch.optim.AdamW(params, lr) as in src/detector/train.py:228 (betas .9/.999, eps 1e-8, wd 1e-2).
// Reductions are per-image fp64 trees in a fixed order (deterministic); the scalar loss is fp32.
#include "wsu_device.h"
// No fused multiply-adds in this file: the loss follows the reference's float32 operation sequence (out * 255 rounded, then subtracted: src/_defs/losses.py:55-63) -- a fused multiply-add
// rounds once and moves the WS term by ~1e-4 relative through its cancellation.
// (Until round 3 the SLP vectorizer happened to pack these products into v_pk_mul_f32 / v_pk_add_f32, which cannot fuse; built without it
// (Makefile) hipcc's default -ffp-contract=fast would fuse them.)
#pragma clang fp contract(off)
#include <cmath>

namespace {

// one workgroup per image: s1 = sum |cov - out|, beta = sum w * s * (in255 - out255)
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ out, const float* __restrict__ cov,
                                                           const float* __restrict__ inp, double* __restrict__ s1,
                                                           double* __restrict__ beta, long long per_img, int squared) {
    __shared__ double ra[1024];
    __shared__ double rb[1024];
    const int n = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)n * per_img;
    const float wgt = 1.0f / (float)per_img;                       // torch: ones / (numel / N), float32
    double a = 0.0, b = 0.0;
    for (long long i = tid; i < per_img; i += 1024) {
        const float o = out[base + i];
        const float d = cov[base + i] - o;
        a += (double)(squared ? d * d : fabsf(d));                  // L2Loss (losses.py:39-42) reuses the L1 slot
        const float in255 = inp[base + i] * 255.0f;
        const float bar = (float)(((int)rintf(in255)) ^ 1);
        b += (double)(wgt * (in255 - bar) * (in255 - o * 255.0f));  // float32 products like the reference
    }
    ra[tid] = a; rb[tid] = b;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) { ra[tid] += ra[tid + s]; rb[tid] += rb[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { s1[n] = ra[0]; beta[n] = rb[0]; }
}

// scalar loss + per-image coefficient of the WS gradient
__global__ void loss_finish_kernel(const double* __restrict__ s1, const double* __restrict__ beta, const float* __restrict__ alphas,
                                   float* __restrict__ loss, float* __restrict__ loss_parts, float* __restrict__ coef,
                                   float* __restrict__ beta_hat, int n, long long per_img, int use_l1, int use_ws) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double l1 = 0.0, ws = 0.0;
    for (int i = 0; i < n; ++i) {
        l1 += s1[i];
        const double bh = beta[i] > 0.0 ? beta[i] : 0.0;
        const double e = bh - (double)alphas[i] / 2.0;
        ws += fabs(e);
        const double sg = e > 0.0 ? 1.0 : (e < 0.0 ? -1.0 : 0.0);
        coef[i] = (float)(beta[i] > 0.0 ? sg / n : 0.0);            // d|e|/dbeta_hat * relu'(beta) / N
        if (beta_hat) beta_hat[i] = (float)bh;
    }
    l1 /= (double)n * (double)per_img;
    ws /= (double)n;
    if (loss_parts) { loss_parts[0] = (float)l1; loss_parts[1] = (float)ws; }
    *loss = (float)((use_l1 ? l1 : 0.0) + (use_ws ? ws : 0.0));
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ out, const float* __restrict__ cov,
                                                        const float* __restrict__ inp, const float* __restrict__ coef,
                                                        float* __restrict__ dout, int n, long long per_img, int use_l1, int use_ws) {
    const long long total = (long long)n * per_img;
    const float inv = 1.0f / (float)total;
    const float wgt = 1.0f / (float)per_img;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float g = 0.f;
        if (use_l1) {
            const float d = cov[i] - out[i];
            if (use_l1 == 2) g += -2.f * d * inv;                    // d(cov - out)^2/dout
            else g += (d > 0.f ? -inv : (d < 0.f ? inv : 0.f));     // d|cov - out|/dout = -sign(cov - out)
        }
        if (use_ws) {
            const float in255 = inp[i] * 255.0f;
            const float bar = (float)(((int)rintf(in255)) ^ 1);
            g += coef[i / per_img] * (-255.0f * wgt) * (in255 - bar);
        }
        dout[i] = g;
    }
}

struct AdamTensor { float* p; const float* g; float* m; float* v; long long n; long long first_block; };

__global__ __launch_bounds__(256) void adamw_kernel(const AdamTensor* __restrict__ tab, int ntensors,
                                                    float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float grad_scale, const int* __restrict__ skip_flag) {
    if (skip_flag && skip_flag[0]) return;                          // a non-finite gradient bucket: leave parameters and moments alone
    // locate this block's tensor (tables are tiny: <= a few dozen entries)
    int t = 0;
    while (t + 1 < ntensors && (long long)blockIdx.x >= tab[t + 1].first_block) ++t;
    const AdamTensor e = tab[t];
    const long long i0 = ((long long)blockIdx.x - e.first_block) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = i0 + threadIdx.x + k * 256;
        if (i >= e.n) break;
        const float g = e.g[i] * grad_scale;
        float p = e.p[i];
        p *= 1.0f - lr * wd;                                        // decoupled weight decay
        const float m = b1 * e.m[i] + (1.0f - b1) * g;
        const float v = b2 * e.v[i] + (1.0f - b2) * g * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p -= (lr / bc1) * (m / denom);
        e.p[i] = p; e.m[i] = m; e.v[i] = v;
    }
}

// ---- gradient scale of the f16f8x backward chain, finite guard, multi-tensor scaling ---------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = fabsf(x[i]);
        m = (v > m || v != v) ? v : m;                              // NaN propagates (its bit pattern wins the integer max below)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(m, o, 64);
        m = (w > m || w != w) ? w : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));     // non-negative floats order like their bits
}

// scale = 2^floor(2 - log2(max |x|)): max |x| * scale in (2, 4] -- 2^14 of headroom below f16's largest value for gradients that grow on the
// way down the network, while values 2^-27 of that maximum still keep an absolute error below theirs (autograd.py)
__global__ void pow2_scale_kernel(const unsigned* __restrict__ bits, float* __restrict__ scale2) {
    float m = __builtin_bit_cast(float, bits[0]);
    if (!(m >= 1e-30f)) m = 1e-30f;                                  // zero gradient (or NaN: the finite guard deals with that)
    if (m > 1e30f) m = 1e30f;
    const float s = exp2f(floorf(2.0f - log2f(m)));
    scale2[0] = s; scale2[1] = 1.0f / s;
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, const float* __restrict__ factor) {
    const float f = factor[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = x[i] * f;
}

struct ScaleTensor { float* p; long long n; long long first_block; };

__global__ __launch_bounds__(256) void scale_multi_kernel(const ScaleTensor* __restrict__ tab, int ntensors, const float* __restrict__ factor) {
    int t = 0;
    while (t + 1 < ntensors && (long long)blockIdx.x >= tab[t + 1].first_block) ++t;
    const ScaleTensor e = tab[t];
    const float f = factor[0];
    const long long i0 = ((long long)blockIdx.x - e.first_block) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = i0 + threadIdx.x + k * 256;
        if (i < e.n) e.p[i] *= f;
    }
}

__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ g, long long n, int* __restrict__ flag) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.4028234e38f);                       // inf or NaN
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ void nonfinite_count_kernel(int* __restrict__ flag) { if (flag[0]) flag[1] += 1; }

}  // namespace

extern "C" {

// Power-of-two scale of the f16f8x backward chain (model/autograd.py), computed on the device: scale2[0] = 2^floor(2 - log2 max|x|),
// scale2[1] = its reciprocal.  workspace: 4 bytes.  Replaces the abs / max / log2 / floor / exp2 chain of ATen launches.
int wsu_pow2_grad_scale(const float* x, long long n, float* scale2, void* workspace, void* stream) {
    WSU_REQUIRE(x && scale2 && workspace && n > 0, "pow2_grad_scale: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(workspace, 0, 4, s) != hipSuccess) { wsu_set_error("pow2_grad_scale: memset failed"); return WSU_ERR_HIP; }
    const unsigned nblk = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(absmax_kernel, dim3(nblk), dim3(256), 0, s, x, n, static_cast<unsigned*>(workspace));
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, s, static_cast<const unsigned*>(workspace), scale2);
    return wsu_check_launch("pow2_scale_kernel");
}

// y = x * factor[0] (factor on the device; y may alias x).
int wsu_scale_f32(const float* x, float* y, long long n, const float* factor, void* stream) {
    WSU_REQUIRE(x && y && factor && n > 0, "scale_f32: bad arguments");
    const unsigned nblk = (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(scale_kernel, dim3(nblk), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, factor);
    return wsu_check_launch("scale_kernel");
}

// In-place scaling of many tensors by factor[0] in ONE launch.  table: DEVICE array of records {float* p; int64 n; int64 first_block}
// (first_block = prefix sum of ceil(n / 1024)), total_blocks = sum ceil(n / 1024).
int wsu_scale_multi_tensor(const void* table, int ntensors, long long total_blocks, const float* factor, void* stream) {
    WSU_REQUIRE(table && factor && ntensors > 0 && total_blocks > 0 && total_blocks < 0x7FFFFFFFLL, "scale_multi_tensor: bad arguments");
    hipLaunchKernelGGL(scale_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const ScaleTensor*>(table), ntensors, factor);
    return wsu_check_launch("scale_multi_kernel");
}

// Finite guard of the gradient bucket: flag[0] = 1 if g holds an inf / NaN else 0; flag[1] counts the flagged calls.  The flag is what
// wsu_adamw_multi_tensor(skip_flag) reads -- no host synchronisation anywhere.
int wsu_nonfinite_flag(const float* g, long long n, int* flag, void* stream) {
    WSU_REQUIRE(g && flag && n > 0, "nonfinite_flag: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(flag, 0, 4, s) != hipSuccess) { wsu_set_error("nonfinite_flag: memset failed"); return WSU_ERR_HIP; }
    const unsigned nblk = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(nonfinite_kernel, dim3(nblk), dim3(256), 0, s, g, n, flag);
    hipLaunchKernelGGL(nonfinite_count_kernel, dim3(1), dim3(1), 0, s, flag);
    return wsu_check_launch("nonfinite_kernel");
}

size_t wsu_l1ws_loss_workspace_bytes(int n) { return (size_t)n * (2 * sizeof(double) + sizeof(float)); }

// out, covers, inputs: (N, C, H, W) fp32 (any contiguous layout, same for all three); alphas: (N) fp32.
// loss: 1 float; loss_parts: optional 2 floats (l1, ws); dout: same shape as out (dLoss/dout); beta_hat: optional (N).
int wsu_l1ws_loss_fwd_bwd(const float* out, const float* covers, const float* inputs, const float* alphas,
                          float* loss, float* loss_parts, float* dout, float* beta_hat, void* workspace, size_t workspace_bytes,
                          int n, long long per_image, int use_l1, int use_ws, void* stream) {
    WSU_REQUIRE(out && covers && inputs && alphas && loss && dout && workspace, "l1ws_loss: null pointer");
    WSU_REQUIRE(n > 0 && per_image > 0 && (use_l1 || use_ws), "l1ws_loss: bad arguments");
    WSU_REQUIRE(workspace_bytes >= wsu_l1ws_loss_workspace_bytes(n), "l1ws_loss: workspace too small");
    double* s1 = static_cast<double*>(workspace);
    double* beta = s1 + n;
    float* coef = reinterpret_cast<float*>(beta + n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(n), dim3(1024), 0, s, out, covers, inputs, s1, beta, per_image, use_l1 == 2 ? 1 : 0);
    int rc = wsu_check_launch("loss_reduce_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, s1, beta, alphas, loss, loss_parts, coef, beta_hat, n, per_image, use_l1, use_ws);
    rc = wsu_check_launch("loss_finish_kernel");
    if (rc) return rc;
    const long long total = (long long)n * per_image;
    const unsigned nblk = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(nblk), dim3(256), 0, s, out, covers, inputs, coef, dout, n, per_image, use_l1, use_ws);
    return wsu_check_launch("loss_grad_kernel");
}

// Multi-tensor AdamW.  `table` is a DEVICE array of ntensors records {p, g, m, v, n, first_block} (6 x 8 bytes each,
// first_block = prefix sum of ceil(n/1024)), built once by the host side; total_blocks = sum ceil(n/1024).
// step is the 1-based update count; grad_scale multiplies every gradient (1/world_size after a sum all-reduce); skip_flag (optional
// device int, wsu_nonfinite_flag): when non-zero the launch changes nothing (a poisoned bucket must not reach the moments).
int wsu_adamw_multi_tensor(const void* table, int ntensors, long long total_blocks,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                           const int* skip_flag, void* stream) {
    WSU_REQUIRE(table && ntensors > 0 && total_blocks > 0 && total_blocks < 0x7FFFFFFFLL && step >= 1, "adamw: bad arguments");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));         // host double like torch's Python floats
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const AdamTensor*>(table), ntensors, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale, skip_flag);
    return wsu_check_launch("adamw_kernel");
}

}  // extern "C"
