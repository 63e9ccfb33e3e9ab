// K7, the non-GEMM-heavy pieces of the backward pass (fp32 storage throughout):
//   * reflect-padding adjoint ("border fold") that completes the 3x3 data gradient,
//   * 2x2 max-pool backward (first-max-wins routing) fused with the ReLU mask,
//   * 1x1 head + sigmoid backward,
//   * 2x2 stride-2 transposed-conv data gradient (MFMA, same staging as the forward kernels),
//   * first-layer (few input planes) weight / bias gradient.
// Reference semantics: autograd through src/unet/model/unet.py:137-189 (oracle: oracle/unet_ref.py under
// torch.autograd).  Everything is deterministic: fixed-order two-stage reductions, no float atomics.
#include "wsu_device.h"

extern "C" int wsu_conv3x3_launch_ex(const void* x1, const void* x2, const void* w_packed, const float* bias,
                                     void* y, void* y2, int csplit, void* y_pool, uint8_t* pool_idx,
                                     const void* relu_mask, const void* relu_mask2,
                                     int n, int h, int w, int c1, int c2, int cout,
                                     int mode, int relu, int pad_zero, void* stream);

extern "C" int wsu_conv3x3_pack_dgrad_swapped(const float* w_oihw, void* w_packed, int cin, int cout, int mode, void* stream);

namespace {

// ---------------------------------------------------------------------------------------------------
// Reflect adjoint.  y = conv(reflect_pad(x)) means dx = fold(dxpad), dxpad = full correlation of the zero-extended g with the
// flipped weights.  The interior part (dxpad restricted to the image) is the zero-padded conv computed by conv3x3_kernel with
// transposed weights; what the padded border RING (rows -1 and H, columns -1 and W) folds back is added here:
//   dx[y, x] += sum over (yp, xp) in preimage(y, x) \ {(y, x)} of dxpad[yp, xp],
//   preimage rows of y = {y} U {-1 if y == 1} U {H if y == H-2}, same for columns  (only rows 1, H-2 and columns 1, W-2 receive).
// The ring values are themselves zero-padded convolutions of thin strips of g, so they go through the same MFMA kernel:
//   top / bottom:  2 x (W+2) images [0 ; g row 0] and [g row H-1 ; 0]          -> output row 0 / 1 = dxpad[-1, -1..W] / dxpad[H, ..]
//   left / right:  the columns g[:, 0] and g[:, W-1] laid out as ROWS of 2 x (H+2) images, multiplied with the tap-swapped
//                  weights                                                      -> dxpad[-1..H, -1] / dxpad[.., W]
// (2 launches of ~3 % of the main conv each instead of a scalar kernel), then one thread per destination adds its 1-3 ring values
// in a fixed order and applies the ReLU mask of the producing layer.
// ---------------------------------------------------------------------------------------------------
// strips[b][r][j][c], b in [0, 2N): images 0..N-1 = top (left), N..2N-1 = bottom (right); L = W (H) is the strip length
__global__ __launch_bounds__(256) void ring_gather_kernel(const float* __restrict__ g, float* __restrict__ tb, float* __restrict__ lr,
                                                          int n, int h, int w, int c) {
    const int c4 = c >> 2;
    const long long ntb = (long long)2 * n * 2 * (w + 2) * c4, nlr = (long long)2 * n * 2 * (h + 2) * c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ntb + nlr; i += (long long)gridDim.x * 256) {
        const bool is_tb = i < ntb;
        long long t = is_tb ? i : i - ntb;
        const int len = is_tb ? w : h;
        const int q = (int)(t % c4); t /= c4;
        const int j = (int)(t % (len + 2)); t /= (len + 2);
        const int r = (int)(t & 1); const int bimg = (int)(t >> 1);
        const int img = bimg < n ? bimg : bimg - n;
        const bool second = bimg >= n;                                   // bottom / right
        f32x4 v = mk_f4(0.f, 0.f, 0.f, 0.f);
        if (j >= 1 && j <= len && r == (second ? 0 : 1)) {
            const int yy = is_tb ? (second ? h - 1 : 0) : j - 1;
            const int xx = is_tb ? j - 1 : (second ? w - 1 : 0);
            v = *reinterpret_cast<const f32x4*>(g + ((size_t)(img * h + yy) * w + xx) * c + 4 * q);
        }
        float* dst = is_tb ? tb : lr;
        *reinterpret_cast<f32x4*>(dst + (((size_t)bimg * 2 + r) * (len + 2) + j) * c + 4 * q) = v;
    }
}

__global__ __launch_bounds__(256) void ring_fold_kernel(const float* __restrict__ tb, const float* __restrict__ lr,
                                                        float* __restrict__ dx1, float* __restrict__ dx2,
                                                        const float* __restrict__ mask1, const float* __restrict__ mask2,
                                                        int n, int h, int w, int cin, int csplit) {
    // destination list per image: row 1 (w px), row h-2 (w px, if h-2 != 1), then columns 1 and w-2 for the other rows
    const int nrows = (h - 2 != 1) ? 2 : 1;
    const int ncols = (w - 2 != 1) ? 2 : 1;
    const int per_img = nrows * w + ncols * (h - nrows);
    const int c4 = cin >> 2;
    const long long total = (long long)n * per_img * c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % c4); long long t = i / c4;
        int b = (int)(t % per_img); const int img = (int)(t / per_img);
        int y, x;
        if (b < nrows * w) { y = (b / w == 0) ? 1 : h - 2; x = b % w; }
        else {
            b -= nrows * w;
            const int k = b / ncols, which = b % ncols;
            int row = k;
            if (row >= 1) ++row;                              // skip row 1
            if (nrows == 2 && row >= h - 2) ++row;            // skip row h-2
            y = row; x = (which == 0) ? 1 : w - 2;
        }
        int ys[3], xs[3]; int ny = 0, nx = 0;
        ys[ny++] = y; if (y == 1) ys[ny++] = -1; if (y == h - 2) ys[ny++] = h;
        xs[nx++] = x; if (x == 1) xs[nx++] = -1; if (x == w - 2) xs[nx++] = w;
        const int ci = 4 * q;
        f32x4 acc = mk_f4(0.f, 0.f, 0.f, 0.f);
        for (int iy = 0; iy < ny; ++iy)
            for (int ix = 0; ix < nx; ++ix) {
                if (iy == 0 && ix == 0) continue;             // (y, x) itself is the interior part
                const int yp = ys[iy], xp = xs[ix];
                const float* src;
                if (yp == -1)      src = tb + (((size_t)img * 2 + 0) * (w + 2) + xp + 1) * cin;
                else if (yp == h)  src = tb + (((size_t)(n + img) * 2 + 1) * (w + 2) + xp + 1) * cin;
                else if (xp == -1) src = lr + (((size_t)img * 2 + 0) * (h + 2) + yp + 1) * cin;
                else               src = lr + (((size_t)(n + img) * 2 + 1) * (h + 2) + yp + 1) * cin;
                acc = acc + *reinterpret_cast<const f32x4*>(src + ci);
            }
        float* dst; const float* mk; int c, cc;
        if (ci < csplit) { dst = dx1; mk = mask1; c = csplit; cc = ci; }
        else             { dst = dx2; mk = mask2; c = cin - csplit; cc = ci - csplit; }
        const size_t o = ((size_t)(img * h + y) * w + x) * c + cc;
        if (mk) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(mk + o);
            if (!(m.x > 0.f)) acc.x = 0.f; if (!(m.y > 0.f)) acc.y = 0.f;
            if (!(m.z > 0.f)) acc.z = 0.f; if (!(m.w > 0.f)) acc.w = 0.f;
        }
        *reinterpret_cast<f32x4*>(dst + o) = *reinterpret_cast<const f32x4*>(dst + o) + acc;
    }
}

// ---------------------------------------------------------------------------------------------------
// max-pool backward, gather form: every full-resolution element adds its pooled gradient iff it is the
// recorded argmax; optional ReLU mask of the pooled activation (xp > 0  <=>  the argmax element > 0).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(float* __restrict__ gfull, const float* __restrict__ dyp,
                                                          const uint8_t* __restrict__ idx, const float* __restrict__ xp_mask,
                                                          int n, int h, int w, int c, int accumulate) {
    const int gpp = c >> 2, hp = h >> 1, wp = w >> 1;
    const long long total = (long long)n * h * w * gpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int gq = (int)(i % gpp); long long t = i / gpp;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h); const int nn = (int)(t / h);
        f32x4* gp = reinterpret_cast<f32x4*>(gfull + ((size_t)(nn * h + y) * w + x) * c + gq * 4);
        f32x4 out = accumulate ? *gp : mk_f4(0.f, 0.f, 0.f, 0.f);
        const int py = y >> 1, px = x >> 1;
        if (py < hp && px < wp) {
            const size_t po = ((size_t)(nn * hp + py) * wp + px) * c + gq * 4;
            const uint32_t k4 = *reinterpret_cast<const uint32_t*>(idx + po);
            const f32x4 d = *reinterpret_cast<const f32x4*>(dyp + po);
            const uint32_t me = (uint32_t)((y & 1) * 2 + (x & 1));
            f32x4 m = mk_f4(1.f, 1.f, 1.f, 1.f);
            if (xp_mask) m = *reinterpret_cast<const f32x4*>(xp_mask + po);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (((k4 >> (8 * e)) & 0xFF) == me && m[e] > 0.f) out[e] += d[e];
        }
        *gp = out;
    }
}

// ---------------------------------------------------------------------------------------------------
// Head backward: z = w.x + b, out = sigmoid(z).  dz = dout * out * (1 - out);
//   gx[p, c] = (x[p, c] > 0) * sum_co dz[co] * w[co, c]      (x is the post-ReLU input, so this is the
//                                                             pre-activation gradient of the layer below)
//   dw[co, c] = sum_p dz[co] * x[p, c],   db[co] = sum_p dz[co]
// ---------------------------------------------------------------------------------------------------
constexpr int HEAD_MAXCO = 4;
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ out, const float* __restrict__ dout,
                                                       float* __restrict__ gx, float* __restrict__ part,
                                                       int n, int hw, int c, int cout, int apply_relu_mask) {
    const int gpp = c >> 2;                                   // lanes per pixel
    const int ppb = 256 / gpp;                                // pixels per block pass
    const int gq = threadIdx.x % gpp, pl = threadIdx.x / gpp;
    const long long npix = (long long)n * hw;
    f32x4 wv[HEAD_MAXCO];
#pragma unroll
    for (int co = 0; co < HEAD_MAXCO; ++co)
        wv[co] = co < cout ? *reinterpret_cast<const f32x4*>(w + (size_t)co * c + gq * 4) : mk_f4(0.f, 0.f, 0.f, 0.f);
    f32x4 aw[HEAD_MAXCO]; float ab[HEAD_MAXCO];
#pragma unroll
    for (int co = 0; co < HEAD_MAXCO; ++co) { aw[co] = mk_f4(0.f, 0.f, 0.f, 0.f); ab[co] = 0.f; }
    for (long long p = (long long)blockIdx.x * ppb + pl; p < npix; p += (long long)gridDim.x * ppb) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)p * c + gq * 4);
        const long long nn = p / hw, r = p % hw;
        f32x4 gv = mk_f4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int co = 0; co < HEAD_MAXCO; ++co) {
            if (co < cout) {
                const size_t o = ((size_t)nn * cout + co) * hw + r;
                const float ov = out[o];
                const float dz = dout[o] * ov * (1.f - ov);
                gv += wv[co] * dz;
                aw[co] += xv * dz;
                ab[co] += dz;
            }
        }
        if (apply_relu_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!(xv[e] > 0.f)) gv[e] = 0.f;
        }
        *reinterpret_cast<f32x4*>(gx + (size_t)p * c + gq * 4) = gv;
    }
    // block reduction over the ppb pixel lanes that share a granule (fixed order), then one partial row per block
    __shared__ float red[256 * 5];
    for (int co = 0; co < cout; ++co) {
        __syncthreads();
        red[threadIdx.x * 5 + 0] = aw[co][0]; red[threadIdx.x * 5 + 1] = aw[co][1];
        red[threadIdx.x * 5 + 2] = aw[co][2]; red[threadIdx.x * 5 + 3] = aw[co][3];
        red[threadIdx.x * 5 + 4] = ab[co];
        __syncthreads();
        if (pl == 0) {
            float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < ppb; ++k)
#pragma unroll
                for (int e = 0; e < 5; ++e) s[e] += red[(k * gpp + gq) * 5 + e];
            float* dst = part + ((size_t)blockIdx.x * cout + co) * (c + 1);
            dst[gq * 4 + 0] = s[0]; dst[gq * 4 + 1] = s[1]; dst[gq * 4 + 2] = s[2]; dst[gq * 4 + 3] = s[3];
            if (gq == 0) dst[c] = s[4];                       // every granule lane saw every dz of its pixels: use lane 0's
        }
    }
}
__global__ void head_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                       int nblocks, int c, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cout * (c + 1)) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * cout * (c + 1) + i];
    const int co = i / (c + 1), k = i % (c + 1);
    if (k < c) dw[(size_t)co * c + k] = s; else db[co] = s;
}

// ---------------------------------------------------------------------------------------------------
// Transposed-conv data gradient:  dx[n,i,j,ci] = sum_{a,b,co} dy[n,2i+a,2j+b,co] * w[ci,co,a,b]
// GEMM  M = 64 ci, N = 4 x 32 input-resolution pixels, K = 4 sub-positions x Cout.  The dy tile (8 x 64
// output pixels) is staged de-interleaved into 4 sub-position images so each MFMA B-operand read is a
// contiguous 512 bytes.  One wave per tile row, 64 ci x 32 px per wave.  Epilogue applies the ReLU mask of
// the layer below and stores NHWC 16-byte pieces.
// ---------------------------------------------------------------------------------------------------
constexpr int CTB_TW = 32, CTB_TH = 4, CTB_NPIX = CTB_TW * CTB_TH;
constexpr int CTB_PLANE = CTB_NPIX * 16 + 32;                       // one (sub, granule) plane of 128 pixels
constexpr int CTB_LDS_IN = 4 * WSU_GRAN * CTB_PLANE;                // 33280
constexpr int CTB_LDS_W = 4 * WSU_GRAN * WSU_COB * 16;              // 16384
struct CtbArgs {
    const char* dy; const char* wp; char* dx; const char* mask;
    int n, h, w, cin, cout, tiles_x, tiles_y, ncb, nch;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void convt2x2_bwd_data_kernel(const CtbArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CK = 16;                                          // fp32 storage modes only
    constexpr int STRIDE = WSU_COB * 4 + 16;
    const int tid = threadIdx.x;
    const unsigned lid = wsu_xcd_remap(blockIdx.x, gridDim.x);
    const int cb = lid % a.ncb;
    int tile = lid / a.ncb;
    const int tx = tile % a.tiles_x; tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int n = tile / a.tiles_y;
    const int y0 = ty * CTB_TH, x0 = tx * CTB_TW;
    const int oh = 2 * a.h, ow = 2 * a.w;
    const int wv = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    for (int c = 0; c < a.nch; ++c) {
        __syncthreads();
        // dy tile: 8 x 64 output pixels x 64 B; item = (output pixel, granule or half)
        constexpr int NITEM = (MODE == WSU_MODE_BF16X3) ? 8 * 64 * 2 : 8 * 64 * 4;
        for (int i = tid; i < NITEM; i += 256) {
            const int op = (MODE == WSU_MODE_BF16X3) ? (i >> 1) : (i >> 2);
            const int sub = (MODE == WSU_MODE_BF16X3) ? (i & 1) : (i & 3);
            const int orow = op >> 6, ocol = op & 63;
            const int oy = min(2 * y0 + orow, oh - 1), ox = min(2 * x0 + ocol, ow - 1);
            const int s = (orow & 1) * 2 + (ocol & 1);
            const int p = (orow >> 1) * CTB_TW + (ocol >> 1);
            const char* src = a.dy + (((size_t)(n * oh + oy) * ow + ox) * a.cout + c * CK) * 4;
            if constexpr (MODE == WSU_MODE_BF16X3) {
                const u32x4* gsrc = reinterpret_cast<const u32x4*>(src + sub * 32);
                u32x4 hi, lo;
                wsu_split8(__builtin_bit_cast(f32x4, gsrc[0]), __builtin_bit_cast(f32x4, gsrc[1]), hi, lo);
                *reinterpret_cast<u32x4*>(smem + (s * WSU_GRAN + sub) * CTB_PLANE + p * 16) = hi;
                *reinterpret_cast<u32x4*>(smem + (s * WSU_GRAN + 2 + sub) * CTB_PLANE + p * 16) = lo;
            } else {
                *reinterpret_cast<u32x4*>(smem + (s * WSU_GRAN + sub) * CTB_PLANE + p * 16) = *reinterpret_cast<const u32x4*>(src + sub * 16);
            }
        }
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp + ((size_t)cb * a.nch + c) * CTB_LDS_W);
        u32x4* wdst = reinterpret_cast<u32x4*>(smem + CTB_LDS_IN);
        for (int i = tid; i < CTB_LDS_W / 16; i += 256) wdst[i] = wsrc[i];
        __syncthreads();
        const char* ldsA = smem + CTB_LDS_IN + l31 * 16;
        const char* ldsB = smem + (wv * CTB_TW + l31) * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (MODE == WSU_MODE_BF16X3) {
                u32x4 ahi[2], alo[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ahi[m] = *reinterpret_cast<const u32x4*>(ldsA + ((s * 4 + hh) * 64 + m * 32) * 16);
                    alo[m] = *reinterpret_cast<const u32x4*>(ldsA + ((s * 4 + 2 + hh) * 64 + m * 32) * 16);
                }
                const u32x4 bhi = *reinterpret_cast<const u32x4*>(ldsB + (s * 4 + hh) * CTB_PLANE);
                const u32x4 blo = *reinterpret_cast<const u32x4*>(ldsB + (s * 4 + 2 + hh) * CTB_PLANE);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    wsu_mfma_step<MODE>(alo[m], bhi, acc[m]);
                    wsu_mfma_step<MODE>(ahi[m], blo, acc[m]);
                    wsu_mfma_step<MODE>(ahi[m], bhi, acc[m]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int g = 2 * ks + hh;
                    const u32x4 bv = *reinterpret_cast<const u32x4*>(ldsB + (s * 4 + g) * CTB_PLANE);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const u32x4 av = *reinterpret_cast<const u32x4*>(ldsA + ((s * 4 + g) * 64 + m * 32) * 16);
                        wsu_mfma_step<MODE>(av, bv, acc[m]);
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int ci = m * 32 + 8 * g4 + 4 * hh;
            *reinterpret_cast<f32x4*>(smem + (wv * CTB_TW + l31) * STRIDE + ci * 4) =
                mk_f4(acc[m][4 * g4 + 0], acc[m][4 * g4 + 1], acc[m][4 * g4 + 2], acc[m][4 * g4 + 3]);
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CTB_NPIX * 16 / 256; ++k) {
        const int i = tid + k * 256;
        const int px = i >> 4, v = i & 15;
        const int r = px / CTB_TW, cc = px % CTB_TW;
        if (y0 + r < a.h && x0 + cc < a.w) {
            f32x4 val = *reinterpret_cast<const f32x4*>(smem + px * STRIDE + v * 16);
            const size_t off = (((size_t)(n * a.h + y0 + r) * a.w + x0 + cc) * a.cin + cb * WSU_COB) * 4 + v * 16;
            if (a.mask) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(a.mask + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(mk[e] > 0.f)) val[e] = 0.f;
            }
            *reinterpret_cast<f32x4*>(a.dx + off) = val;
        }
    }
}

// (Cin, Cout, 2, 2) fp32 -> [ci block][chunk over co][sub][granule][ci 64][16 B]
template <int MODE>
__global__ void pack_convt_dgrad_kernel(const float* __restrict__ w, char* __restrict__ dst, int cin, int cout) {
    constexpr int CK = 16;
    constexpr int EPG = (MODE == WSU_MODE_F32) ? 4 : 8;
    const int nch = cout / CK;
    const long long total = (long long)(cin / WSU_COB) * nch * 4 * WSU_GRAN * WSU_COB * EPG;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int e = t % EPG; t /= EPG;
        const int cil = t % WSU_COB; t /= WSU_COB;
        const int g = t % WSU_GRAN; t /= WSU_GRAN;
        const int sub = t % 4; t /= 4;
        const int c = t % nch; t /= nch;
        const int cb = (int)t;
        int co, part = 0;
        if (MODE == WSU_MODE_F32) co = c * CK + 4 * g + e;
        else { co = c * CK + 8 * (g & 1) + e; part = g >> 1; }
        const float val = w[((size_t)(cb * WSU_COB + cil) * cout + co) * 4 + sub];
        if (MODE == WSU_MODE_F32) reinterpret_cast<float*>(dst)[d] = val;
        else {
            const __bf16 hv = (__bf16)(part ? wsu_bf16_lo_residual(val) : val);
            reinterpret_cast<uint16_t*>(dst)[d] = __builtin_bit_cast(uint16_t, hv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// First-layer weight / bias gradient:  dW[co, ci, tap] = sum_p g[p, co] * xpad_ci[p + tap],  db[co] = sum_p g[p, co]
// x: NCHW fp32 planes, g: NHWC (N,H,W,cout).  64 output channels per block column, pixels chunked over
// blocks, two-stage fixed-order reduction.
// ---------------------------------------------------------------------------------------------------
// Pixels are walked in tiles of 256: the g tile [256 px][64 co] (coalesced 16-byte loads) and the 3x3 input windows [256 px][12] go
// through LDS, then thread (cg = 4 output channels, ps = pixel slice) accumulates its 9 x 4 products over the pixels ps, ps+16, ...
// (a wave reads 4 consecutive 256-byte rows per step: conflict free).  One block column per 64 output channels.
constexpr int FW_TILE = 256;
constexpr int FW_LDS = FW_TILE * 64 * 4 + FW_TILE * 12 * 4;       // 77824 B

__global__ __launch_bounds__(256, 2) void first_wgrad_partial_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                     float* __restrict__ part, int n, int h, int w, int cin, int cout, int chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gl = reinterpret_cast<float*>(smem);                    // [256][64]
    float* xl = gl + FW_TILE * 64;                                 // [256][12]
    const int tid = threadIdx.x, cg = tid & 15, ps = tid >> 4;
    const int ch0 = blockIdx.y * 64;
    const long long npix = (long long)n * h * w;
    const long long p0 = (long long)blockIdx.x * chunk, p1 = min(p0 + chunk, npix);
    float* dst = part + (size_t)blockIdx.x * (cin * 9 + 1) * cout;
    for (int ci = 0; ci < cin; ++ci) {                             // the bias sum rides on the ci == 0 pass
        f32x4 acc[9], accb = mk_f4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = mk_f4(0.f, 0.f, 0.f, 0.f);
        for (long long t0 = p0; t0 < p1; t0 += FW_TILE) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < FW_TILE * 16 / 256; ++k) {             // 16 granules of 16 B per pixel row
                const int i = tid + k * 256, pl = i >> 4, q = i & 15;
                f32x4 v = mk_f4(0.f, 0.f, 0.f, 0.f);
                if (t0 + pl < p1) v = *reinterpret_cast<const f32x4*>(g + (size_t)(t0 + pl) * cout + ch0 + 4 * q);
                *reinterpret_cast<f32x4*>(gl + pl * 64 + 4 * q) = v;
            }
            {
                const long long p = t0 + tid;                          // one pixel's window per thread
                float win[12];
#pragma unroll
                for (int t = 0; t < 12; ++t) win[t] = 0.f;
                if (p < p1) {
                    const int xx = (int)(p % w); const long long t2 = p / w;
                    const int yy = (int)(t2 % h); const int nn = (int)(t2 / h);
                    const float* xp = x + ((size_t)nn * cin + ci) * h * w;
#pragma unroll
                    for (int t = 0; t < 9; ++t) win[t] = xp[(size_t)wsu_reflect(yy + t / 3 - 1, h) * w + wsu_reflect(xx + t % 3 - 1, w)];
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) *reinterpret_cast<f32x4*>(xl + tid * 12 + 4 * t) = mk_f4(win[4 * t], win[4 * t + 1], win[4 * t + 2], win[4 * t + 3]);
            }
            __syncthreads();
#pragma unroll 4
            for (int i = 0; i < FW_TILE / 16; ++i) {
                const int pl = ps + 16 * i;
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gl + pl * 64 + 4 * cg);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(xl + pl * 12), w1 = *reinterpret_cast<const f32x4*>(xl + pl * 12 + 4);
                const float w8 = xl[pl * 12 + 8];
                accb = accb + gv;
                acc[0] = acc[0] + gv * w0.x; acc[1] = acc[1] + gv * w0.y; acc[2] = acc[2] + gv * w0.z; acc[3] = acc[3] + gv * w0.w;
                acc[4] = acc[4] + gv * w1.x; acc[5] = acc[5] + gv * w1.y; acc[6] = acc[6] + gv * w1.z; acc[7] = acc[7] + gv * w1.w;
                acc[8] = acc[8] + gv * w8;
            }
        }
        // reduce the 16 pixel slices in a fixed order through LDS: red[slice][10 rows][64 co]
        __syncthreads();
        float* red = gl;                                            // 16 * 10 * 64 floats = 40 KB, inside the g tile
#pragma unroll
        for (int t = 0; t < 10; ++t)
            *reinterpret_cast<f32x4*>(red + (ps * 10 + t) * 64 + 4 * cg) = t < 9 ? acc[t < 9 ? t : 0] : accb;
        __syncthreads();
        for (int i = tid; i < 10 * 64; i += 256) {
            const int t = i >> 6, co = i & 63;
            if (t == 9 && ci != 0) continue;
            float sum = 0.f;
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) sum += red[(sl * 10 + t) * 64 + co];
            const int row = t < 9 ? ci * 9 + t : cin * 9;
            dst[(size_t)row * cout + ch0 + co] = sum;
        }
    }
}
__global__ void first_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                          int nchunks, int cin, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = cin * 9 + 1;
    if (i >= rows * cout) return;
    float s = 0.f;
    for (int k = 0; k < nchunks; ++k) s += part[(size_t)k * rows * cout + i];
    const int r = i / cout, co = i % cout;
    if (r < cin * 9) dw[(size_t)co * cin * 9 + r] = s; else if (db) db[co] = s;
}

// ---------------------------------------------------------------------------------------------------
// First-layer data gradient (input saliency, src/saliency.py:159-174): dx[n,ci,y,x] = sum over the padded positions that
// reflect onto (y,x) of sum_{u,v,co} W[co,ci,u,v] * g[n, yp-u+1, xp-v+1, co]  (g zero outside the image).
// One thread per input element; weights transposed to [ci][tap][co] in LDS so the 64-wide dot reads are contiguous.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void first_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                          float* __restrict__ dx, int n, int h, int wd, int cin, int cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);                 // [ci][tap][co]
    for (int i = threadIdx.x; i < cin * 9 * cout; i += blockDim.x) {
        const int co = i % cout, tap = (i / cout) % 9, ci = i / (9 * cout);
        wl[i] = w[((size_t)co * cin + ci) * 9 + tap];
    }
    __syncthreads();
    const long long total = (long long)n * cin * h * wd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % wd); long long t = i / wd;
        const int y = (int)(t % h); t /= h;
        const int ci = (int)(t % cin); const int img = (int)(t / cin);
        int ys[3], xs[3]; int ny = 0, nx = 0;
        ys[ny++] = y; if (y == 1) ys[ny++] = -1; if (y == h - 2) ys[ny++] = h;
        xs[nx++] = x; if (x == 1) xs[nx++] = -1; if (x == wd - 2) xs[nx++] = wd;
        float acc = 0.f;
        for (int iy = 0; iy < ny; ++iy)
            for (int ix = 0; ix < nx; ++ix)
                for (int u = 0; u < 3; ++u) {
                    const int sy = ys[iy] - u + 1;
                    if (sy < 0 || sy >= h) continue;
                    for (int v = 0; v < 3; ++v) {
                        const int sx = xs[ix] - v + 1;
                        if (sx < 0 || sx >= wd) continue;
                        const f32x4* gp = reinterpret_cast<const f32x4*>(g + ((size_t)(img * h + sy) * wd + sx) * cout);
                        const f32x4* wp = reinterpret_cast<const f32x4*>(wl + (ci * 9 + u * 3 + v) * cout);
                        float s = 0.f;
                        for (int c4 = 0; c4 < cout / 4; ++c4) {
                            const f32x4 a = gp[c4], b = wp[c4];
                            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
                        }
                        acc += s;
                    }
                }
        dx[i] = acc;
    }
}

}  // namespace

extern "C" {

// dx (N, cin, H, W) NCHW fp32 of the first layer from its pre-activation gradient g (N,H,W,cout).
int wsu_conv3x3_first_bwd_data(const float* g, const float* w_oihw, float* dx_nchw, int n, int h, int w, int cin, int cout, void* stream) {
    WSU_REQUIRE(g && w_oihw && dx_nchw, "conv3x3_first_bwd_data: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2 && cin >= 1 && cin <= 8 && cout % 4 == 0 && cout > 0, "conv3x3_first_bwd_data: bad shape");
    const long long total = (long long)n * cin * h * w;
    const unsigned nblk = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(first_dgrad_kernel, dim3(nblk), dim3(256), (size_t)cin * 9 * cout * sizeof(float), static_cast<hipStream_t>(stream),
                       g, w_oihw, dx_nchw, n, h, w, cin, cout);
    return wsu_check_launch("first_dgrad_kernel");
}

// dx (= pre-activation gradient of the producing layer when relu_mask is given) of the 3x3 reflect conv.
//   g: (N,H,W,Cout) fp32;  w_packed_dgrad: wsu_conv3x3_pack_dgrad output;  w_oihw: the plain weight (re-packed with swapped taps
//   for the left / right strips);  workspace: wsu_conv3x3_bwd_data_workspace_bytes().
//   dx1: (N,H,W,csplit), dx2: (N,H,W,cin-csplit) or NULL (csplit == cin);  relu_mask*: tensors of the dx shapes or NULL.
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t wsu_conv3x3_bwd_data_workspace_bytes(int n, int h, int w, int cin, int cout, int mode) {
    if (n <= 0 || h < 2 || w < 2 || cin <= 0 || cout <= 0) return 0;
    const size_t strips = (size_t)2 * n * 2 * ((size_t)(w + 2) + (size_t)(h + 2));
    return align256(wsu_conv3x3_packed_bytes(cout, cin, mode)) + align256(strips * cout * sizeof(float)) + align256(strips * cin * sizeof(float));
}

int wsu_conv3x3_bwd_data(const void* g, const void* w_packed_dgrad, const float* w_oihw, void* workspace, size_t workspace_bytes,
                         void* dx1, void* dx2, int csplit, const void* relu_mask1, const void* relu_mask2,
                         int n, int h, int w, int cin, int cout, int mode, void* stream) {
    WSU_REQUIRE(mode == WSU_MODE_F32 || mode == WSU_MODE_BF16X3 || mode == WSU_MODE_F16F8X, "conv3x3_bwd_data: fp32-storage modes only (got %d)", mode);
    WSU_REQUIRE(w_oihw && workspace, "conv3x3_bwd_data: null weight / workspace");
    WSU_REQUIRE(cin % WSU_COB == 0 && csplit % 4 == 0 && cout % 4 == 0, "conv3x3_bwd_data: cin=%d must be a multiple of %d", cin, WSU_COB);
    WSU_REQUIRE(workspace_bytes >= wsu_conv3x3_bwd_data_workspace_bytes(n, h, w, cin, cout, mode), "conv3x3_bwd_data: workspace too small");
    int rc = wsu_conv3x3_launch_ex(g, nullptr, w_packed_dgrad, nullptr, dx1, dx2, csplit, nullptr, nullptr,
                                   relu_mask1, relu_mask2, n, h, w, cout, 0, cin, mode, 0, 1, stream);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* wsp = static_cast<char*>(workspace);
    void* wp_swapped = wsp;
    float* in_tb = reinterpret_cast<float*>(wsp + align256(wsu_conv3x3_packed_bytes(cout, cin, mode)));
    float* in_lr = in_tb + (size_t)2 * n * 2 * (w + 2) * cout;
    float* out_tb = reinterpret_cast<float*>(reinterpret_cast<char*>(in_tb) + align256((size_t)2 * n * 2 * ((size_t)(w + 2) + (h + 2)) * cout * sizeof(float)));
    float* out_lr = out_tb + (size_t)2 * n * 2 * (w + 2) * cin;
    rc = wsu_conv3x3_pack_dgrad_swapped(w_oihw, wp_swapped, cin, cout, mode, stream);
    if (rc) return rc;
    {
        const long long items = (long long)2 * n * 2 * ((long long)(w + 2) + (h + 2)) * (cout / 4);
        const unsigned nblk = (unsigned)((items + 255) / 256 < 65536 ? (items + 255) / 256 : 65536);
        hipLaunchKernelGGL(ring_gather_kernel, dim3(nblk), dim3(256), 0, s, (const float*)g, in_tb, in_lr, n, h, w, cout);
        rc = wsu_check_launch("ring_gather_kernel");
        if (rc) return rc;
    }
    rc = wsu_conv3x3_launch_ex(in_tb, nullptr, w_packed_dgrad, nullptr, out_tb, nullptr, cin, nullptr, nullptr, nullptr, nullptr,
                               2 * n, 2, w + 2, cout, 0, cin, mode, 0, 1, stream);
    if (rc) return rc;
    rc = wsu_conv3x3_launch_ex(in_lr, nullptr, wp_swapped, nullptr, out_lr, nullptr, cin, nullptr, nullptr, nullptr, nullptr,
                               2 * n, 2, h + 2, cout, 0, cin, mode, 0, 1, stream);
    if (rc) return rc;
    const int nrows = (h - 2 != 1) ? 2 : 1, ncols = (w - 2 != 1) ? 2 : 1;
    const long long items = (long long)n * (nrows * w + ncols * (h - nrows)) * (cin / 4);
    const unsigned nblk = (unsigned)((items + 255) / 256 < 65536 ? (items + 255) / 256 : 65536);
    hipLaunchKernelGGL(ring_fold_kernel, dim3(nblk), dim3(256), 0, s, out_tb, out_lr, (float*)dx1, (float*)dx2,
                       (const float*)relu_mask1, (const float*)relu_mask2, n, h, w, cin, csplit);
    return wsu_check_launch("ring_fold_kernel");
}

int wsu_maxpool2x2_bwd(float* g_full, const float* dy_pool, const uint8_t* pool_idx, const float* xp_relu_mask,
                       int n, int h, int w, int c, int accumulate, void* stream) {
    WSU_REQUIRE(g_full && dy_pool && pool_idx, "maxpool2x2_bwd: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2 && c > 0 && c % 4 == 0, "maxpool2x2_bwd: bad shape");
    const long long total = (long long)n * h * w * (c / 4);
    const unsigned nblk = (unsigned)((total + 255) / 256 < 131072 ? (total + 255) / 256 : 131072);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblk), dim3(256), 0, static_cast<hipStream_t>(stream), g_full, dy_pool, pool_idx, xp_relu_mask, n, h, w, c, accumulate);
    return wsu_check_launch("maxpool_bwd_kernel");
}

size_t wsu_head_bwd_workspace_bytes(int c, int cout) { return (size_t)1024 * cout * (c + 1) * sizeof(float); }

int wsu_conv1x1_sigmoid_bwd(const float* x, const float* w, const float* out, const float* dout,
                            float* gx, float* dw, float* db, float* workspace, size_t workspace_bytes,
                            int n, int h, int w_, int c, int cout, int apply_relu_mask, void* stream) {
    WSU_REQUIRE(x && w && out && dout && gx && dw && db && workspace, "conv1x1_sigmoid_bwd: null pointer");
    const int gpp = c / 4;
    WSU_REQUIRE(c % 4 == 0 && gpp >= 1 && gpp <= 64 && (gpp & (gpp - 1)) == 0, "conv1x1_sigmoid_bwd: c=%d unsupported", c);
    WSU_REQUIRE(cout >= 1 && cout <= HEAD_MAXCO, "conv1x1_sigmoid_bwd: cout=%d outside 1..%d", cout, HEAD_MAXCO);
    const long long npix = (long long)n * h * w_;
    const int ppb = 256 / gpp;
    int nblk = (int)((npix + ppb - 1) / ppb < 1024 ? (npix + ppb - 1) / ppb : 1024);
    WSU_REQUIRE((size_t)nblk * cout * (c + 1) * sizeof(float) <= workspace_bytes, "conv1x1_sigmoid_bwd: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(head_bwd_kernel, dim3(nblk), dim3(256), 0, s, x, w, out, dout, gx, workspace, n, h * w_, c, cout, apply_relu_mask);
    int rc = wsu_check_launch("head_bwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(head_bwd_reduce_kernel, dim3((cout * (c + 1) + 63) / 64), dim3(64), 0, s, workspace, dw, db, nblk, c, cout);
    return wsu_check_launch("head_bwd_reduce_kernel");
}

size_t wsu_convt2x2_packed_dgrad_bytes(int cin, int cout, int mode) {
    if (cin <= 0 || cout <= 0 || (mode != WSU_MODE_F32 && mode != WSU_MODE_BF16X3)) return 0;
    return (size_t)cin * cout * 4 * 4;
}

int wsu_convt2x2_pack_dgrad(const float* w_iohw, void* w_packed, int cin, int cout, int mode, void* stream) {
    WSU_REQUIRE(w_iohw && w_packed, "convt2x2_pack_dgrad: null pointer");
    WSU_REQUIRE(mode == WSU_MODE_F32 || mode == WSU_MODE_BF16X3, "convt2x2_pack_dgrad: fp32-storage modes only");
    WSU_REQUIRE(cin > 0 && cin % WSU_COB == 0 && cout > 0 && cout % 16 == 0, "convt2x2_pack_dgrad: bad channels cin=%d cout=%d", cin, cout);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_F32) hipLaunchKernelGGL(pack_convt_dgrad_kernel<WSU_MODE_F32>, dim3(512), dim3(256), 0, s, w_iohw, (char*)w_packed, cin, cout);
    else hipLaunchKernelGGL(pack_convt_dgrad_kernel<WSU_MODE_BF16X3>, dim3(512), dim3(256), 0, s, w_iohw, (char*)w_packed, cin, cout);
    return wsu_check_launch("pack_convt_dgrad_kernel");
}

int wsu_convt2x2_bwd_data(const void* dy, const void* w_packed_dgrad, void* dx, const void* relu_mask,
                          int n, int h, int w, int cin, int cout, int mode, void* stream) {
    WSU_REQUIRE(dy && w_packed_dgrad && dx, "convt2x2_bwd_data: null pointer");
    WSU_REQUIRE(mode == WSU_MODE_F32 || mode == WSU_MODE_BF16X3, "convt2x2_bwd_data: fp32-storage modes only");
    WSU_REQUIRE(n > 0 && h > 0 && w > 0 && cin % WSU_COB == 0 && cout % 16 == 0, "convt2x2_bwd_data: bad shape");
    CtbArgs a;
    a.dy = (const char*)dy; a.wp = (const char*)w_packed_dgrad; a.dx = (char*)dx; a.mask = (const char*)relu_mask;
    a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
    a.tiles_x = (w + CTB_TW - 1) / CTB_TW; a.tiles_y = (h + CTB_TH - 1) / CTB_TH; a.ncb = cin / WSU_COB; a.nch = cout / 16;
    const long long nblk = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nblk > 0 && nblk < 0x7FFFFFFFLL, "convt2x2_bwd_data: grid out of range");
    constexpr int EPI = CTB_NPIX * (WSU_COB * 4 + 16);
    constexpr int LDS = (CTB_LDS_IN + CTB_LDS_W) > EPI ? (CTB_LDS_IN + CTB_LDS_W) : EPI;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_F32) hipLaunchKernelGGL(convt2x2_bwd_data_kernel<WSU_MODE_F32>, dim3((unsigned)nblk), dim3(256), LDS, s, a);
    else hipLaunchKernelGGL(convt2x2_bwd_data_kernel<WSU_MODE_BF16X3>, dim3((unsigned)nblk), dim3(256), LDS, s, a);
    return wsu_check_launch("convt2x2_bwd_data_kernel");
}

static int first_bwd_chunk(long long npix) {                  // ~1024 chunks whatever the batch, at least 2048 pixels each
    const long long c = (npix + 1023) / 1024;
    return (int)(c < 2048 ? 2048 : c);
}

size_t wsu_first_bwd_workspace_bytes(int n, int h, int w, int cin, int cout) {
    const long long npix = (long long)n * h * w;
    const int chunk = first_bwd_chunk(npix);
    const long long nchunks = (npix + chunk - 1) / chunk;
    return (size_t)nchunks * (cin * 9 + 1) * cout * sizeof(float);
}

int wsu_conv3x3_first_bwd_weight(const float* g, const float* x_nchw, float* dw, float* db,
                                 float* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout, void* stream) {
    WSU_REQUIRE(g && x_nchw && dw && workspace, "conv3x3_first_bwd_weight: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2 && cin >= 1 && cin <= 8 && cout % 64 == 0, "conv3x3_first_bwd_weight: bad shape");
    const long long npix = (long long)n * h * w;
    const int chunk = first_bwd_chunk(npix);
    const int nchunks = (int)((npix + chunk - 1) / chunk);
    WSU_REQUIRE(wsu_first_bwd_workspace_bytes(n, h, w, cin, cout) <= workspace_bytes, "conv3x3_first_bwd_weight: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&first_wgrad_partial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FW_LDS);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(first_wgrad): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        attr_done = true;
    }
    hipLaunchKernelGGL(first_wgrad_partial_kernel, dim3(nchunks, cout / 64), dim3(256), FW_LDS, s, g, x_nchw, workspace, n, h, w, cin, cout, chunk);
    int rc = wsu_check_launch("first_wgrad_partial_kernel");
    if (rc) return rc;
    const int tot = (cin * 9 + 1) * cout;
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, workspace, dw, db, nchunks, cin, cout);
    return wsu_check_launch("first_wgrad_reduce_kernel");
}

}  // extern "C"
