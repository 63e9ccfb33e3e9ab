// K3: 2x2 stride-2 transposed convolution + bias as four independent GEMMs on the matrix cores.
//
// Replaces nn.ConvTranspose2d(k=2, stride=2) of the reference (src/unet/model/unet.py:74,125,130; used at
// :177,183):   y[n, 2i+a, 2j+b, co] = bias[co] + sum_ci x[n, i, j, ci] * w[ci, co, a, b]
// Stride == kernel, so outputs never overlap: each of the 4 sub-positions (a,b) is a plain GEMM
// [64 co] x [Cin] x [pixels]; one wave per sub-position, 64 co x 64 input pixels (2 rows x 32) per wave.
// Same operand staging as conv3x3.hip (granule-planar LDS, register double buffering).  The epilogue
// interleaves the 4 sub-position tiles into a [4 x 64 output pixels][64 co] LDS tile so that the NHWC
// stores are whole 16-byte pieces of contiguous output rows.
#include "wsu_device.h"
#include <cstdlib>

namespace {

constexpr int TW = 32, TH = 2;                          // input-pixel tile
constexpr int NPIX = TW * TH;                            // 64
constexpr int PLANE_IN = NPIX * 16 + 32;                 // 1056 B
constexpr int LDS_IN = WSU_GRAN * PLANE_IN;              // 4224
constexpr int LDS_W = 4 * WSU_GRAN * WSU_COB * 16;       // 16384
constexpr int LDS_MAIN = LDS_IN + LDS_W;
constexpr int NT = 256;
constexpr int W_VEC = LDS_W / 16 / NT;                   // 4

struct CtArgs {
    const char* x; const char* wp; const float* bias; char* y;
    int n, h, w, cin, cout, tiles_x, tiles_y, ncb, nch;
    int out_split;                   // API mode WSU_MODE_BF16X3S: write the output already split (the input side is the template flag PS)
};


template <int MODE, bool PS = false>
__device__ __forceinline__ void ct_load(const CtArgs& a, int cb, int c, int tid, bool has_item, const char* xsrc,
                                        u32x4 (&st_in)[2], u32x4 (&st_w)[W_VEC]) {
    constexpr int ESZ = (MODE == WSU_MODE_BF16) ? 2 : 4;
    constexpr int CK = (MODE == WSU_MODE_BF16) ? 32 : 16;
    st_in[0] = mk_u4(0, 0, 0, 0); st_in[1] = st_in[0];
    if (has_item) {
        const u32x4* g = reinterpret_cast<const u32x4*>(xsrc + (size_t)c * ((MODE == WSU_MODE_F16F8 && PS) ? 48 : CK * ESZ));
        st_in[0] = g[0];
        if constexpr ((MODE == WSU_MODE_BF16X3 || MODE == WSU_MODE_F16F8) && !PS) st_in[1] = g[1];
    }
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp + ((size_t)cb * a.nch + c) * LDS_W);
    WSU_STATIC_FOR(W_VEC, k, st_w[k] = wsrc[tid + k * NT];);
}

template <int MODE, bool PS = false>
__device__ __forceinline__ void ct_commit(char* smem, int tid, bool has_item, int ldsoff,
                                          const u32x4 (&st_in)[2], const u32x4 (&st_w)[W_VEC]) {
    if (has_item) {
        if constexpr (MODE == WSU_MODE_F16F8 && !PS) {             // fp32 storage (API mode F16F8X): split 8 channels into planes half, 2, 3
            const int half = ldsoff >= PLANE_IN ? 1 : 0, pixoff = ldsoff - half * PLANE_IN;
            uint32_t h0, h1, h2, h3, l0, l1, x0, x1;
            wsu_split4_f16f8(__builtin_bit_cast(f32x4, st_in[0]), WSU_F8_XLO_DIV, WSU_F8_X_DIV, h0, h1, l0, x0);
            wsu_split4_f16f8(__builtin_bit_cast(f32x4, st_in[1]), WSU_F8_XLO_DIV, WSU_F8_X_DIV, h2, h3, l1, x1);
            *reinterpret_cast<u32x4*>(smem + ldsoff) = mk_u4(h0, h1, h2, h3);
            *reinterpret_cast<u32x2*>(smem + 2 * PLANE_IN + pixoff + half * 8) = mk_u2(l0, l1);
            *reinterpret_cast<u32x2*>(smem + 3 * PLANE_IN + pixoff + half * 8) = mk_u2(x0, x1);
        } else if constexpr (MODE == WSU_MODE_BF16X3 && !PS) {
            u32x4 hi, lo;
            wsu_split8(__builtin_bit_cast(f32x4, st_in[0]), __builtin_bit_cast(f32x4, st_in[1]), hi, lo);
            *reinterpret_cast<u32x4*>(smem + ldsoff) = hi;
            *reinterpret_cast<u32x4*>(smem + ldsoff + 2 * PLANE_IN) = lo;
        } else {
            *reinterpret_cast<u32x4*>(smem + ldsoff) = st_in[0];
            if constexpr (MODE == WSU_MODE_F16F8 && PS) {         // an f16 piece: its 8 e4m3 copies fill half of the pixel's slot in plane 3
                if (ldsoff < 2 * PLANE_IN) {
                    const int half = ldsoff >= PLANE_IN ? 1 : 0;
                    *reinterpret_cast<u32x2*>(smem + 3 * PLANE_IN + (ldsoff - half * PLANE_IN) + half * 8) = wsu_f16x8_to_fp8(st_in[0]);
                }
            }
        }
    }
    u32x4* wdst = reinterpret_cast<u32x4*>(smem + LDS_IN);
    WSU_STATIC_FOR(W_VEC, k, wdst[tid + k * NT] = st_w[k];);
}

template <int MODE, bool PS = false>
__global__ __launch_bounds__(NT, 2) void convt2x2_kernel(const CtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ESZ = (MODE == WSU_MODE_BF16) ? 2 : 4;
    constexpr int STRIDE = WSU_COB * ESZ + 16;
    constexpr int VPP = WSU_COB * ESZ / 16;

    const int tid = threadIdx.x;
    const unsigned lid = wsu_xcd_remap(blockIdx.x, gridDim.x);
    const int cb = lid % a.ncb;
    int tile = lid / a.ncb;
    const int tx = tile % a.tiles_x; tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int n = tile / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // staging plan: F32/BF16: 256 items (pixel, granule); BF16X3: 128 items (pixel, half) of 32 B
    constexpr bool SPLIT_HERE = (MODE == WSU_MODE_BF16X3 || MODE == WSU_MODE_F16F8) && !PS;   // PS: pre-split input, 4 granule items per pixel like F32 / BF16
    constexpr bool STORED48 = MODE == WSU_MODE_F16F8 && PS;            // 3 stored pieces of 16 B per pixel and chunk (plane 3 is derived)
    constexpr int NITEMS = SPLIT_HERE ? NPIX * 2 : (STORED48 ? NPIX * 3 : NPIX * 4);
    const int pix = SPLIT_HERE ? (tid >> 1) : (STORED48 ? tid / 3 : (tid >> 2));
    const int sub = SPLIT_HERE ? (tid & 1) : (STORED48 ? tid - 3 * pix : (tid & 3));
    const int pr = (pix / TW) % TH, pc = pix % TW;
    const int yy = min(y0 + pr, a.h - 1), xx = min(x0 + pc, a.w - 1);   // clamp: out-of-image lanes are never stored
    const size_t pidx = (size_t)(n * a.h + yy) * a.w + xx;
    const bool has_item = tid < NITEMS;
    const int ldsoff = sub * PLANE_IN + (pix % NPIX) * 16;

    u32x4 st_in[2]; u32x4 st_w[W_VEC];
    const char* xsrc = a.x + pidx * a.cin * (STORED48 ? 3 : ESZ) + sub * (SPLIT_HERE ? 32 : 16);

    const int wv = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;   // wave = sub-position a*2+b
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
    const char* ldsA = smem + LDS_IN + ((wv * WSU_GRAN) * 64 + l31) * 16;      // + (g*64 + m*32)*16
    const char* ldsB = smem + l31 * 16;                                         // + g*PLANE_IN + q*32*16

    u32x4 f8a0[2], f8a1[2], f8b0[2], f8b1[2];                 // F16F8: fp8 operands of this lane half's chunk, held across two chunks
#pragma unroll
    for (int m = 0; m < 2; ++m) { f8a0[m] = mk_u4(0, 0, 0, 0); f8a1[m] = f8a0[m]; f8b0[m] = f8a0[m]; f8b1[m] = f8a0[m]; }
    ct_load<MODE, PS>(a, cb, 0, tid, has_item, xsrc, st_in, st_w);
    for (int c = 0; c < a.nch; ++c) {
        __syncthreads();
        ct_commit<MODE, PS>(smem, tid, has_item, ldsoff, st_in, st_w);
        __syncthreads();
        if (c + 1 < a.nch) ct_load<MODE, PS>(a, cb, c + 1, tid, has_item, xsrc, st_in, st_w);
        if constexpr (MODE == WSU_MODE_F16F8) {
            // f16(w) * f16(x) per 16-channel chunk; the two cross terms of TWO chunks share one block-scaled fp8 instruction (a scale block
            // is 32 k: lanes 0-31 carry the even chunk's 16 channels, lanes 32-63 the odd chunk's).  Each lane half reads its operands
            // while its chunk is in LDS and holds them in registers; the instruction is issued on odd chunks (cin % 32 == 0, checked).
            if ((c & 1) == hh) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f8a0[m] = *reinterpret_cast<const u32x4*>(ldsA + (2 * 64 + m * 32) * 16);
                    f8a1[m] = *reinterpret_cast<const u32x4*>(ldsA + (3 * 64 + m * 32) * 16);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f8b0[q] = *reinterpret_cast<const u32x4*>(ldsB + 2 * PLANE_IN + q * TW * 16);
                    f8b1[q] = *reinterpret_cast<const u32x4*>(ldsB + 3 * PLANE_IN + q * TW * 16);
                }
            }
            u32x4 ah[2], bh[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) ah[m] = *reinterpret_cast<const u32x4*>(ldsA + (hh * 64 + m * 32) * 16);
#pragma unroll
            for (int q = 0; q < 2; ++q) bh[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE_IN + q * TW * 16);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) wsu_mfma_f16(ah[m], bh[q], acc[m][q]);
            if (c & 1) {
                const int sc_a = hh ? WSU_F8_SCALE_WLO : WSU_F8_SCALE_W, sc_b = hh ? WSU_F8_SCALE_X : WSU_F8_SCALE_XLO;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q) wsu_mfma_f8x2(f8a0[m], f8a1[m], f8b0[q], f8b1[q], sc_a, sc_b, acc[m][q]);
            }
        } else if constexpr (MODE == WSU_MODE_BF16X3) {
            u32x4 ahi[2], alo[2], bhi[2], blo[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ahi[m] = *reinterpret_cast<const u32x4*>(ldsA + (hh * 64 + m * 32) * 16);
                alo[m] = *reinterpret_cast<const u32x4*>(ldsA + ((2 + hh) * 64 + m * 32) * 16);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bhi[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE_IN + q * TW * 16);
                blo[q] = *reinterpret_cast<const u32x4*>(ldsB + (2 + hh) * PLANE_IN + q * TW * 16);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    wsu_mfma_step<MODE>(alo[m], bhi[q], acc[m][q]);
                    wsu_mfma_step<MODE>(ahi[m], blo[q], acc[m][q]);
                    wsu_mfma_step<MODE>(ahi[m], bhi[q], acc[m][q]);
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int g = 2 * ks + hh;
                u32x4 av[2], bv[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) av[m] = *reinterpret_cast<const u32x4*>(ldsA + (g * 64 + m * 32) * 16);
#pragma unroll
                for (int q = 0; q < 2; ++q) bv[q] = *reinterpret_cast<const u32x4*>(ldsB + g * PLANE_IN + q * TW * 16);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(av[m], bv[q], acc[m][q]);
            }
        }
    }

    // ---- epilogue: [4 output rows x 64 output cols][64 co] tile in LDS --------------------------------
    __syncthreads();
    const int sa = wv >> 1, sb = wv & 1;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = m * 32 + 8 * g4 + 4 * hh;
            f32x4 b4 = mk_f4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + cb * WSU_COB + co);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float v0 = acc[m][q][4 * g4 + 0] + b4.x, v1 = acc[m][q][4 * g4 + 1] + b4.y;
                const float v2 = acc[m][q][4 * g4 + 2] + b4.z, v3 = acc[m][q][4 * g4 + 3] + b4.w;
                const int opx = (2 * q + sa) * (2 * TW) + 2 * l31 + sb;
                if constexpr (ESZ == 4) *reinterpret_cast<f32x4*>(smem + opx * STRIDE + co * 4) = mk_f4(v0, v1, v2, v3);
                else *reinterpret_cast<u32x2*>(smem + opx * STRIDE + co * 2) = mk_u2(wsu_pack_bf16x2(v0, v1), wsu_pack_bf16x2(v2, v3));
            }
        }
    }
    __syncthreads();
    const int oh = 2 * a.h, ow = 2 * a.w;
    if constexpr (MODE == WSU_MODE_F16F8 && PS) {
        // encode the tile in place (a 16-channel chunk keeps its 64-byte slot, 48 bytes used), then copy out with 12 consecutive lanes per
        // output pixel so that every store instruction writes whole 192-byte pixel rows (see store_f16f8 in conv3x3.hip)
        for (int i = tid; i < 4 * NPIX * 4; i += NT) {
            u32x4* row = reinterpret_cast<u32x4*>(smem + (i >> 2) * STRIDE + (i & 3) * 64);
            u32x4 hi0, hi1, lo8;
            wsu_split16_f16f8(__builtin_bit_cast(f32x4, row[0]), __builtin_bit_cast(f32x4, row[1]), __builtin_bit_cast(f32x4, row[2]),
                              __builtin_bit_cast(f32x4, row[3]), hi0, hi1, lo8);
            row[0] = hi0; row[1] = hi1; row[2] = lo8;
        }
        __syncthreads();
        for (int i = tid; i < 4 * NPIX * 12; i += NT) {
            const int opx = i / 12, piece = i - 12 * opx;
            const int r = opx / (2 * TW), c = opx % (2 * TW);
            const int oy = 2 * y0 + r, ox = 2 * x0 + c;
            if (oy < oh && ox < ow)
                *reinterpret_cast<u32x4*>(a.y + (((size_t)(n * oh + oy) * ow + ox) * a.cout + cb * WSU_COB) * 3 + piece * 16) =
                    *reinterpret_cast<const u32x4*>(smem + opx * STRIDE + (piece / 3) * 64 + (piece % 3) * 16);
        }
        return;
    }
    if (MODE == WSU_MODE_BF16X3 && a.out_split) {
        for (int i = tid; i < 4 * NPIX * 8; i += NT) {            // one item = 8 channels of one output pixel: hi piece + lo piece
            const int opx = i >> 3, g8 = i & 7;
            const int r = opx / (2 * TW), c = opx % (2 * TW);
            const int oy = 2 * y0 + r, ox = 2 * x0 + c;
            if (oy < oh && ox < ow) {
                const float* row = reinterpret_cast<const float*>(smem + opx * STRIDE) + 8 * g8;
                u32x4 hi, lo;
                wsu_split8(*reinterpret_cast<const f32x4*>(row), *reinterpret_cast<const f32x4*>(row + 4), hi, lo);
                char* dst = a.y + (((size_t)(n * oh + oy) * ow + ox) * a.cout + cb * WSU_COB) * 4 + (g8 >> 1) * 64 + (g8 & 1) * 16;
                *reinterpret_cast<u32x4*>(dst) = hi;
                *reinterpret_cast<u32x4*>(dst + 32) = lo;
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4 * NPIX * VPP / NT; ++k) {
        const int i = tid + k * NT;
        const int opx = i / VPP, v = i % VPP;
        const int r = opx / (2 * TW), c = opx % (2 * TW);
        const int oy = 2 * y0 + r, ox = 2 * x0 + c;
        if (oy < oh && ox < ow) {
            const u32x4 val = *reinterpret_cast<const u32x4*>(smem + opx * STRIDE + v * 16);
            *reinterpret_cast<u32x4*>(a.y + (((size_t)(n * oh + oy) * ow + ox) * a.cout + cb * WSU_COB) * ESZ + v * 16) = val;
        }
    }
}

// ---- mode F16F8, 4 x 32 input pixels per workgroup ------------------------------------------------------------------------------------
// The 2 x 32 tile above stages a 16 KB weight slice per 16-channel chunk for 64 input pixels: 2.6 weight bytes through L2 -> LDS per
// output byte, and two barriers per 8 matrix instructions of a wave (measured 2.6 / 1.8 TB/s of HBM traffic on upconv4 / upconv3).
// This variant doubles the pixels per workgroup (wave = one sub-position x 64 co x 128 px, 128 accumulator registers) and writes the
// 8 x 64 output pixels in two passes through the same 69.6 KB epilogue tile.
namespace f8t {
constexpr int TH4 = 4, NPIX4 = TW * TH4;                     // 128 input pixels
constexpr int PLANE4 = NPIX4 * 16 + 32;                       // 2080 B
constexpr int LDS_IN4 = WSU_GRAN * PLANE4;                    // 8320
constexpr int LDS_MAIN4 = LDS_IN4 + LDS_W;
constexpr int STRIDE = WSU_COB * 4 + 16;
constexpr int EPI4 = 4 * NPIX * STRIDE;                       // one pass = 4 output rows x 64 columns
}

__global__ __launch_bounds__(NT, 2) void convt2x2_f16f8_kernel(const CtArgs a) {
    using namespace f8t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const unsigned lid = wsu_xcd_remap(blockIdx.x, gridDim.x);
    const int cb = lid % a.ncb;
    int tile = lid / a.ncb;
    const int tx = tile % a.tiles_x; tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int n = tile / a.tiles_y;
    const int y0 = ty * TH4, x0 = tx * TW;

    // staging plan: 384 items (pixel, 16-byte piece of the 48 stored bytes of a chunk), two rounds of 256 threads
    const char* xsrc[2]; int ldsoff[2]; bool has[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + j * NT;
        const int pix = i / 3, sub = i - 3 * pix;
        has[j] = i < NPIX4 * 3;
        const int pr = (pix / TW) % TH4, pc = pix % TW;
        const int yy = min(y0 + pr, a.h - 1), xx = min(x0 + pc, a.w - 1);        // clamp: out-of-image lanes are never stored
        xsrc[j] = a.x + ((size_t)(n * a.h + yy) * a.w + xx) * a.cin * 3 + sub * 16;
        ldsoff[j] = sub * PLANE4 + (pix % NPIX4) * 16;
    }
    u32x4 st_in[2], st_w[W_VEC];
    auto load = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            st_in[j] = mk_u4(0, 0, 0, 0);
            if (has[j]) st_in[j] = *reinterpret_cast<const u32x4*>(xsrc[j] + (size_t)c * 48);
        }
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp + ((size_t)cb * a.nch + c) * LDS_W);
        WSU_STATIC_FOR(W_VEC, k, st_w[k] = wsrc[tid + k * NT];);
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (has[j]) {
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = st_in[j];
                if (ldsoff[j] < 2 * PLANE4) {                         // an f16 piece: its 8 e4m3 copies fill half of the pixel's slot in plane 3
                    const int half = ldsoff[j] >= PLANE4 ? 1 : 0;
                    *reinterpret_cast<u32x2*>(smem + 3 * PLANE4 + (ldsoff[j] - half * PLANE4) + half * 8) = wsu_f16x8_to_fp8(st_in[j]);
                }
            }
        u32x4* wdst = reinterpret_cast<u32x4*>(smem + LDS_IN4);
        WSU_STATIC_FOR(W_VEC, k, wdst[tid + k * NT] = st_w[k];);
    };

    const int wv = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;   // wave = sub-position a*2+b
    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
    const char* ldsA = smem + LDS_IN4 + ((wv * WSU_GRAN) * 64 + l31) * 16;      // + (g*64 + m*32)*16
    const char* ldsB = smem + l31 * 16;                                          // + g*PLANE4 + q*32*16
    const u32x4 z4 = mk_u4(0, 0, 0, 0);
    u32x4 f8a0[2] = {z4, z4}, f8a1[2] = {z4, z4}, f8b0[4] = {z4, z4, z4, z4}, f8b1[4] = {z4, z4, z4, z4};
    const int sc_a = hh ? WSU_F8_SCALE_WLO : WSU_F8_SCALE_W, sc_b = hh ? WSU_F8_SCALE_X : WSU_F8_SCALE_XLO;

    load(0);
    for (int c = 0; c < a.nch; ++c) {
        __syncthreads();
        commit();
        __syncthreads();
        if (c + 1 < a.nch) load(c + 1);
        if ((c & 1) == hh) {                                  // this lane half's chunk of the pair: fetch its fp8 operands (see convt2x2_kernel)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f8a0[m] = *reinterpret_cast<const u32x4*>(ldsA + (2 * 64 + m * 32) * 16);
                f8a1[m] = *reinterpret_cast<const u32x4*>(ldsA + (3 * 64 + m * 32) * 16);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f8b0[q] = *reinterpret_cast<const u32x4*>(ldsB + 2 * PLANE4 + q * TW * 16);
                f8b1[q] = *reinterpret_cast<const u32x4*>(ldsB + 3 * PLANE4 + q * TW * 16);
            }
        }
        u32x4 ah[2], bh[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) ah[m] = *reinterpret_cast<const u32x4*>(ldsA + (hh * 64 + m * 32) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) bh[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE4 + q * TW * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) wsu_mfma_f16(ah[m], bh[q], acc[m][q]);
        if (c & 1) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) wsu_mfma_f8x2(f8a0[m], f8a1[m], f8b0[q], f8b1[q], sc_a, sc_b, acc[m][q]);
        }
    }

    // ---- epilogue, two passes: input rows (2p, 2p+1) -> 4 output rows x 64 output columns through the [pixel][64 co] fp32 tile ----
    const int sa = wv >> 1, sb = wv & 1;
    const int oh = 2 * a.h, ow = 2 * a.w;
    WSU_STATIC_FOR(2, p, {
        __syncthreads();
_Pragma("unroll")
        for (int m = 0; m < 2; ++m)
_Pragma("unroll")
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co = m * 32 + 8 * g4 + 4 * hh;
                f32x4 b4 = mk_f4(0.f, 0.f, 0.f, 0.f);
                if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + cb * WSU_COB + co);
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) {
                    const f32x16& v = acc[m][2 * p + q];
                    const int opx = (2 * q + sa) * (2 * TW) + 2 * l31 + sb;
                    *reinterpret_cast<f32x4*>(smem + opx * STRIDE + co * 4) =
                        mk_f4(v[4 * g4 + 0] + b4.x, v[4 * g4 + 1] + b4.y, v[4 * g4 + 2] + b4.z, v[4 * g4 + 3] + b4.w);
                }
            }
        __syncthreads();
        for (int i = tid; i < 4 * NPIX * 4; i += NT) {            // encode in place: a 16-channel chunk keeps its 64-byte slot (48 used)
            u32x4* row = reinterpret_cast<u32x4*>(smem + (i >> 2) * STRIDE + (i & 3) * 64);
            u32x4 hi0, hi1, lo8;
            wsu_split16_f16f8(__builtin_bit_cast(f32x4, row[0]), __builtin_bit_cast(f32x4, row[1]), __builtin_bit_cast(f32x4, row[2]),
                              __builtin_bit_cast(f32x4, row[3]), hi0, hi1, lo8);
            row[0] = hi0; row[1] = hi1; row[2] = lo8;
        }
        __syncthreads();
        for (int i = tid; i < 4 * NPIX * 12; i += NT) {           // 12 consecutive lanes per output pixel: whole 192-byte rows per store
            const int opx = i / 12, piece = i - 12 * opx;
            const int r = opx / (2 * TW), c = opx % (2 * TW);
            const int oy = 2 * y0 + 4 * p + r, ox = 2 * x0 + c;
            if (oy < oh && ox < ow)
                *reinterpret_cast<u32x4*>(a.y + (((size_t)(n * oh + oy) * ow + ox) * a.cout + cb * WSU_COB) * 3 + piece * 16) =
                    *reinterpret_cast<const u32x4*>(smem + opx * STRIDE + (piece / 3) * 64 + (piece % 3) * 16);
        }
    });
}

int launch_ct_f16f8(const CtArgs& a_in, hipStream_t s) {
    CtArgs a = a_in;
    a.tiles_y = (a.h + f8t::TH4 - 1) / f8t::TH4;
    const int lds = f8t::EPI4 > f8t::LDS_MAIN4 ? f8t::EPI4 : f8t::LDS_MAIN4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2x2_f16f8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(convt2x2_f16f8): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        attr_done = true;
    }
    const long long nblk = (long long)a.n * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk <= 0 || nblk > 0x7FFFFFFFLL) { wsu_set_error("convt2x2: grid of %lld workgroups out of range", nblk); return WSU_ERR_ARG; }
    hipLaunchKernelGGL(convt2x2_f16f8_kernel, dim3((unsigned)nblk), dim3(NT), lds, s, a);
    return wsu_check_launch("convt2x2_f16f8_kernel");
}

template <int MODE, bool PS = false>
int launch_ct(const CtArgs& a, hipStream_t s) {
    constexpr int ESZ = (MODE == WSU_MODE_BF16) ? 2 : 4;
    constexpr int EPI = 4 * NPIX * (WSU_COB * ESZ + 16);
    const int lds = EPI > LDS_MAIN ? EPI : LDS_MAIN;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2x2_kernel<MODE, PS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(convt2x2): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        attr_done = true;
    }
    const long long nblk = (long long)a.n * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk <= 0 || nblk > 0x7FFFFFFFLL) { wsu_set_error("convt2x2: grid of %lld workgroups out of range", nblk); return WSU_ERR_ARG; }
    hipLaunchKernelGGL((convt2x2_kernel<MODE, PS>), dim3((unsigned)nblk), dim3(NT), lds, s, a);
    return wsu_check_launch("convt2x2_kernel");
}

// (Cin, Cout, 2, 2) fp32 -> [cob][chunk][sub = a*2+b][granule][co 64][16 B]
template <int MODE>
__global__ void pack_convt_kernel(const float* __restrict__ w, char* __restrict__ dst, int cin, int cout) {
    constexpr int CK = (MODE == WSU_MODE_BF16) ? 32 : 16;
    constexpr int EPG = (MODE == WSU_MODE_F32) ? 4 : 8;
    const int nch = cin / CK;
    const long long total = (long long)(cout / WSU_COB) * nch * 4 * WSU_GRAN * WSU_COB * EPG;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int e = t % EPG; t /= EPG;
        const int co = t % WSU_COB; t /= WSU_COB;
        const int g = t % WSU_GRAN; t /= WSU_GRAN;
        const int sub = t % 4; t /= 4;
        const int c = t % nch; t /= nch;
        const int cb = (int)t;
        int ci, part = 0;
        if (MODE == WSU_MODE_F32) ci = c * CK + 4 * g + e;
        else if (MODE == WSU_MODE_BF16) ci = c * CK + 8 * g + e;
        else { ci = c * CK + 8 * (g & 1) + e; part = g >> 1; }
        const float val = w[((size_t)ci * cout + cb * WSU_COB + co) * 4 + sub];
        if (MODE == WSU_MODE_F32) reinterpret_cast<float*>(dst)[d] = val;
        else {
            const __bf16 hv = (__bf16)(part ? wsu_bf16_lo_residual(val) : val);
            reinterpret_cast<uint16_t*>(dst)[d] = __builtin_bit_cast(uint16_t, hv);
        }
    }
}

// F16F8: one thread per (cob, chunk, sub, co) row of 16 input channels -> [f16 0-7][f16 8-15][e4m3(w * 2^6)][e4m3((w - f16 w) * 2^18)]
__global__ void pack_convt_f16f8_kernel(const float* __restrict__ w, char* __restrict__ dst, int cin, int cout) {
    const int nch = cin / 16;
    const long long total = (long long)(cout / WSU_COB) * nch * 4 * WSU_COB;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int co = t % WSU_COB; t /= WSU_COB;
        const int sub = t % 4; t /= 4;
        const int c = t % nch; t /= nch;
        const int cb = (int)t;
        f32x4 q[4];
#pragma unroll
        for (int e = 0; e < 16; ++e) q[e >> 2][e & 3] = w[((size_t)(c * 16 + e) * cout + cb * WSU_COB + co) * 4 + sub];
        uint32_t h[8], l[4], x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wsu_split4_f16f8(q[k], WSU_F8_WLO_DIV, WSU_F8_W_DIV, h[2 * k], h[2 * k + 1], l[k], x[k]);
        char* base = dst + (((size_t)cb * nch + c) * 4 + sub) * (WSU_GRAN * WSU_COB * 16) + co * 16;
        *reinterpret_cast<u32x4*>(base) = mk_u4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<u32x4*>(base + WSU_COB * 16) = mk_u4(h[4], h[5], h[6], h[7]);
        *reinterpret_cast<u32x4*>(base + 2 * WSU_COB * 16) = mk_u4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<u32x4*>(base + 3 * WSU_COB * 16) = mk_u4(l[0], l[1], l[2], l[3]);
    }
}

}  // namespace

extern "C" {

size_t wsu_convt2x2_packed_bytes(int cin, int cout, int mode) {
    if (cin <= 0 || cout <= 0 || mode < 0 || (mode > 2 && mode != WSU_MODE_F16F8 && mode != WSU_MODE_F16F8X)) return 0;
    return (size_t)cin * cout * 4 * (mode == WSU_MODE_BF16 ? 2 : 4);
}

int wsu_convt2x2_pack(const float* w_iohw, void* w_packed, int cin, int cout, int mode, void* stream) {
    WSU_REQUIRE(w_iohw && w_packed, "convt2x2_pack: null pointer");
    if (mode == WSU_MODE_F16F8X) mode = WSU_MODE_F16F8;
    WSU_REQUIRE((mode >= 0 && mode <= 2) || mode == WSU_MODE_F16F8, "convt2x2_pack: bad mode %d", mode);
    WSU_REQUIRE(cin > 0 && cin % wsu_chunk_channels(mode) == 0, "convt2x2_pack: cin=%d not a multiple of %d", cin, wsu_chunk_channels(mode));
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0, "convt2x2_pack: cout=%d not a multiple of %d", cout, WSU_COB);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_F16F8) hipLaunchKernelGGL(pack_convt_f16f8_kernel, dim3(512), dim3(256), 0, s, w_iohw, (char*)w_packed, cin, cout);
    else if (mode == WSU_MODE_F32) hipLaunchKernelGGL(pack_convt_kernel<WSU_MODE_F32>, dim3(512), dim3(256), 0, s, w_iohw, (char*)w_packed, cin, cout);
    else if (mode == WSU_MODE_BF16X3) hipLaunchKernelGGL(pack_convt_kernel<WSU_MODE_BF16X3>, dim3(512), dim3(256), 0, s, w_iohw, (char*)w_packed, cin, cout);
    else hipLaunchKernelGGL(pack_convt_kernel<WSU_MODE_BF16>, dim3(512), dim3(256), 0, s, w_iohw, (char*)w_packed, cin, cout);
    return wsu_check_launch("pack_convt_kernel");
}

int wsu_convt2x2_fwd(const void* x, const void* w_packed, const float* bias, void* y,
                     int n, int h, int w, int cin, int cout, int mode, void* stream) {
    WSU_REQUIRE(mode >= 0 && mode <= 5, "convt2x2: bad mode %d", mode);
    const bool presplit = mode == WSU_MODE_BF16X3S;             // input and output stored already split (see wsu.h)
    if (presplit) mode = WSU_MODE_BF16X3;
    const bool f16f8x = mode == WSU_MODE_F16F8X;                // F16F8 arithmetic on fp32 tensors
    if (f16f8x) mode = WSU_MODE_F16F8;
    WSU_REQUIRE(x && w_packed && y, "convt2x2: null pointer");
    WSU_REQUIRE(n > 0 && h > 0 && w > 0, "convt2x2: bad shape n=%d h=%d w=%d", n, h, w);
    WSU_REQUIRE(cin > 0 && cin % wsu_chunk_channels(mode) == 0, "convt2x2: cin=%d not a multiple of %d", cin, wsu_chunk_channels(mode));
    WSU_REQUIRE(mode != WSU_MODE_F16F8 || cin % 32 == 0, "convt2x2: mode F16F8 pairs 16-channel chunks, cin=%d must be a multiple of 32", cin);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0, "convt2x2: cout=%d not a multiple of %d", cout, WSU_COB);
    WSU_REQUIRE((long long)n * h * w * 4 < 0x7FFFFFFFLL, "convt2x2: output pixel count overflows int32");
    CtArgs a;
    a.x = (const char*)x; a.wp = (const char*)w_packed; a.bias = bias; a.y = (char*)y;
    a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ncb = cout / WSU_COB;
    a.nch = cin / wsu_chunk_channels(mode);
    a.out_split = presplit;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_F32) return launch_ct<WSU_MODE_F32>(a, s);
    if (presplit) return launch_ct<WSU_MODE_BF16X3, true>(a, s);
    if (f16f8x) return launch_ct<WSU_MODE_F16F8, false>(a, s);
    if (mode == WSU_MODE_F16F8) {
        static int small = -1;                                    // WSU_CONVT_TILE=2: the 2 x 32 tile of the generic kernel (A/B runs)
        if (small < 0) { const char* e = getenv("WSU_CONVT_TILE"); small = (e && e[0] == '2') ? 1 : 0; }
        return small ? launch_ct<WSU_MODE_F16F8, true>(a, s) : launch_ct_f16f8(a, s);
    }
    if (mode == WSU_MODE_BF16X3) return launch_ct<WSU_MODE_BF16X3>(a, s);
    return launch_ct<WSU_MODE_BF16>(a, s);
}

}  // extern "C"
