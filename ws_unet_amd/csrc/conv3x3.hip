// K1 (+K2 pool epilogue, +K4 fused cat): 3x3 convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces nn.Conv2d(k=3, padding=1, padding_mode='reflect') + F.relu, torch.cat([up, skip]) in front of
// d*1 and nn.MaxPool2d(2,2) behind e*2 of the reference (src/unet/model/unet.py:73,82-132,141-186).
//
// GEMM view per workgroup:   D[co, p] = sum_{tap, ci} Wt[co, (tap, ci)] * X[(tap, ci), p]
//   M = 64 output channels (A operand = packed weights), N = 8x32 output pixels (B operand = activations),
//   K = 9 taps x Cin, walked in chunks of 64 bytes of channels per pixel (see wsu_device.h).
// The M/N orientation puts 4 consecutive output channels of one pixel in 4 consecutive accumulator
// registers (v_mfma 32x32 C/D map: col = lane&31 -> pixel, row = (r&3)+8(r>>2)+4(lane>>5) -> channel),
// so the epilogue packs 16-byte NHWC pieces without any cross-lane traffic.
//
// LDS (one workgroup = 4 waves, 2 workgroups per CU):
//   input tile   [4 granule planes][10 x 34 pixels][16 B]   halo resolved (reflect / zero) while staging
//   weight tile  [9 taps][4 granule planes][64 co][16 B]    straight copy of the packed weights
//   both granule-planar, so every fragment read is a contiguous 512-byte ds_read_b128 (conflict free).
// Staging is register-double-buffered: chunk c+1's global loads are in flight while chunk c is multiplied.
// The epilogue re-uses the LDS as a [pixel][channel] tile: bias + ReLU (+ReLU-mask for the data-gradient
// pass), coalesced 16-byte NHWC stores, and the fused 2x2 max-pool with first-max-wins argmax.
#include "wsu_device.h"

namespace {

constexpr int TW = 32, TH = 8;
constexpr int IW = TW + 2, IH = TH + 2;
constexpr int NPIX_IN = IW * IH;                         // 340
constexpr int PLANE_IN = NPIX_IN * 16 + 96;              // 5536 B: planes 8 dwords apart mod 32 banks
constexpr int LDS_IN = WSU_GRAN * PLANE_IN;              // 22144
constexpr int LDS_W = 9 * WSU_GRAN * WSU_COB * 16;       // 36864
constexpr int LDS_MAIN = LDS_IN + LDS_W;                 // 59008
constexpr int NT = 256;
constexpr int W_VEC = LDS_W / 16 / NT;                   // 9 x 16 B per thread
constexpr int IN_VEC = 6;                                // ceil(1360 / 256) x 16 B per thread

struct ConvArgs {
    const char* x1; const char* x2; const char* wp; const float* bias;
    char* y; char* y2; char* ypool; uint8_t* pidx; const char* relu_mask; const char* relu_mask2;
    int n, h, w, c1, c2, cout, csplit;
    int tiles_x, tiles_y, ncb, nch1, nch;
    int relu, pad_zero;
};

template <int MODE> struct Epi {
    static constexpr int ESZ = (MODE == WSU_MODE_BF16) ? 2 : 4;
    static constexpr int STRIDE = WSU_COB * ESZ + 16;    // bytes per pixel in the epilogue tile
    static constexpr int BYTES = TH * TW * STRIDE;
    static constexpr int VPP = WSU_COB * ESZ / 16;       // 16-byte pieces per pixel
};


// Global -> registers for chunk c (input tile items + this workgroup's packed-weight slice).
template <int MODE>
__device__ __forceinline__ void stage_load(const ConvArgs& a, int cb, int c, int tid, const int (&pixidx)[IN_VEC],
                                           u32x4 (&st_in)[IN_VEC], u32x4 (&st_w)[W_VEC]) {
    constexpr int ESZ = Epi<MODE>::ESZ;
    constexpr int CK = (MODE == WSU_MODE_BF16) ? 32 : 16;
    constexpr int NLOOP = (MODE == WSU_MODE_BF16X3) ? 3 : IN_VEC;
    const char* src; int csrc, ch0;
    if (c < a.nch1) { src = a.x1; csrc = a.c1; ch0 = c * CK; }
    else            { src = a.x2; csrc = a.c2; ch0 = (c - a.nch1) * CK; }
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        const int p = pixidx[j];
        if constexpr (MODE == WSU_MODE_BF16X3) {
            const int sub = (tid + j * NT) & 1;
            u32x4 v0 = mk_u4(0, 0, 0, 0), v1 = v0;
            if (p >= 0) {
                const u32x4* g = reinterpret_cast<const u32x4*>(src + ((size_t)p * csrc + ch0) * 4 + sub * 32);
                v0 = g[0]; v1 = g[1];
            }
            st_in[2 * j] = v0; st_in[2 * j + 1] = v1;
        } else {
            const int sub = (tid + j * NT) & 3;
            u32x4 v = mk_u4(0, 0, 0, 0);
            if (p >= 0) v = *reinterpret_cast<const u32x4*>(src + ((size_t)p * csrc + ch0) * ESZ + sub * 16);
            st_in[j] = v;
        }
    }
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp + ((size_t)cb * a.nch + c) * LDS_W);
#pragma unroll
    for (int k = 0; k < W_VEC; ++k) st_w[k] = wsrc[tid + k * NT];
}

// Registers -> LDS (granule-planar input tile, linear weight tile); BF16X3 splits fp32 into bf16 hi/lo here.
template <int MODE>
__device__ __forceinline__ void stage_commit(char* smem, int tid, const int (&pixidx)[IN_VEC], const int (&ldsoff)[IN_VEC],
                                             const u32x4 (&st_in)[IN_VEC], const u32x4 (&st_w)[W_VEC]) {
    constexpr int NLOOP = (MODE == WSU_MODE_BF16X3) ? 3 : IN_VEC;
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        if (pixidx[j] != -2) {
            if constexpr (MODE == WSU_MODE_BF16X3) {
                u32x4 hi, lo;
                wsu_split8(__builtin_bit_cast(f32x4, st_in[2 * j]), __builtin_bit_cast(f32x4, st_in[2 * j + 1]), hi, lo);
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = hi;
                *reinterpret_cast<u32x4*>(smem + ldsoff[j] + 2 * PLANE_IN) = lo;
            } else {
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = st_in[j];
            }
        }
    }
    u32x4* wdst = reinterpret_cast<u32x4*>(smem + LDS_IN);
#pragma unroll
    for (int k = 0; k < W_VEC; ++k) wdst[tid + k * NT] = st_w[k];
}

template <int MODE>
__global__ __launch_bounds__(NT, 2) void conv3x3_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ESZ = Epi<MODE>::ESZ;
    const int tid = threadIdx.x;
    const unsigned lid = wsu_xcd_remap(blockIdx.x, gridDim.x);
    const int cb = lid % a.ncb;
    int tile = lid / a.ncb;
    const int tx = tile % a.tiles_x; tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int n = tile / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- per-thread staging plan (identical for every chunk) ---------------------------------------
    int pixidx[IN_VEC];      // linear pixel index (n*H + y)*W + x of the source, -1 = zero, -2 = no item
    int ldsoff[IN_VEC];
    constexpr int NITEMS = (MODE == WSU_MODE_BF16X3) ? NPIX_IN * 2 : NPIX_IN * 4;
    constexpr int NLOOP = (MODE == WSU_MODE_BF16X3) ? 3 : IN_VEC;
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        const int i = tid + j * NT;
        const int pix = (MODE == WSU_MODE_BF16X3) ? (i >> 1) : (i >> 2);
        const int sub = (MODE == WSU_MODE_BF16X3) ? (i & 1) : (i & 3);
        const int r = pix / IW, c = pix - r * IW;
        int yy = y0 - 1 + r, xx = x0 - 1 + c;
        int p;
        if (a.pad_zero) {
            p = (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) ? (n * a.h + yy) * a.w + xx : -1;
        } else {
            yy = wsu_reflect(yy, a.h); xx = wsu_reflect(xx, a.w);
            p = (n * a.h + yy) * a.w + xx;
        }
        pixidx[j] = (i < NITEMS) ? p : -2;
        ldsoff[j] = sub * PLANE_IN + pix * 16;          // BF16X3: hi plane `sub`, lo plane `2 + sub`
    }

    u32x4 st_in[IN_VEC];
    u32x4 st_w[W_VEC];
    // ---- main loop ------------------------------------------------------------------------------------
    const int wv = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    const char* ldsA = smem + LDS_IN + l31 * 16;                     // + ((tap*4+g)*64 + mt*32)*16
    const char* ldsB = smem + ((2 * wv) * IW + l31) * 16;            // + g*PLANE_IN + ((nt+dy)*IW + dx)*16

    stage_load<MODE>(a, cb, 0, tid, pixidx, st_in, st_w);
    for (int c = 0; c < a.nch; ++c) {
        __syncthreads();
        stage_commit<MODE>(smem, tid, pixidx, ldsoff, st_in, st_w);
        __syncthreads();
        if (c + 1 < a.nch) stage_load<MODE>(a, cb, c + 1, tid, pixidx, st_in, st_w);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            if constexpr (MODE == WSU_MODE_BF16X3) {
                u32x4 ahi[2], alo[2], bhi[2], blo[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ahi[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4 + hh) * 64 + m * 32) * 16);
                    alo[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4 + 2 + hh) * 64 + m * 32) * 16);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    bhi[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE_IN + ((q + dy) * IW + dx) * 16);
                    blo[q] = *reinterpret_cast<const u32x4*>(ldsB + (2 + hh) * PLANE_IN + ((q + dy) * IW + dx) * 16);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        wsu_mfma_step<MODE>(alo[m], bhi[q], acc[m][q]);     // small terms first
                        wsu_mfma_step<MODE>(ahi[m], blo[q], acc[m][q]);
                        wsu_mfma_step<MODE>(ahi[m], bhi[q], acc[m][q]);
                    }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int g = 2 * ks + hh;
                    u32x4 av[2], bv[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        av[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4) * 64 + m * 32) * 16 + g * (64 * 16));
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        bv[q] = *reinterpret_cast<const u32x4*>(ldsB + g * PLANE_IN + ((q + dy) * IW + dx) * 16);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(av[m], bv[q], acc[m][q]);
                }
            }
        }
    }

    // ---- epilogue: accumulators -> [pixel][channel] LDS tile ---------------------------------------
    __syncthreads();
    constexpr int STRIDE = Epi<MODE>::STRIDE;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = m * 32 + 8 * g4 + 4 * hh;                  // 4 consecutive channels co..co+3
            f32x4 b4 = mk_f4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + cb * WSU_COB + co);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v0 = acc[m][q][4 * g4 + 0] + b4.x, v1 = acc[m][q][4 * g4 + 1] + b4.y;
                float v2 = acc[m][q][4 * g4 + 2] + b4.z, v3 = acc[m][q][4 * g4 + 3] + b4.w;
                if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                const int px = (2 * wv + q) * TW + l31;
                if constexpr (ESZ == 4) {
                    *reinterpret_cast<f32x4*>(smem + px * STRIDE + co * 4) = mk_f4(v0, v1, v2, v3);
                } else {
                    *reinterpret_cast<u32x2*>(smem + px * STRIDE + co * 2) = mk_u2(wsu_pack_bf16x2(v0, v1), wsu_pack_bf16x2(v2, v3));
                }
            }
        }
    }
    __syncthreads();

    // ---- coalesced NHWC store (16 B per thread), optional ReLU mask of the data-gradient pass --------
    constexpr int VPP = Epi<MODE>::VPP;
    const int cglob = cb * WSU_COB;                                   // first output channel of this block
    char* ydst = a.y; int ych = a.csplit, ycoff = cglob;
    const char* msk = a.relu_mask;
    if (cglob >= a.csplit) { ydst = a.y2; ych = a.cout - a.csplit; ycoff = cglob - a.csplit; msk = a.relu_mask2; }
#pragma unroll
    for (int k = 0; k < TH * TW * VPP / NT; ++k) {
        const int i = tid + k * NT;
        const int px = i / VPP, v = i % VPP;
        const int r = px / TW, c = px % TW;
        if (y0 + r < a.h && x0 + c < a.w) {
            u32x4 val = *reinterpret_cast<const u32x4*>(smem + px * STRIDE + v * 16);
            const size_t off = (((size_t)(n * a.h + y0 + r) * a.w + x0 + c) * ych + ycoff) * ESZ + v * 16;
            if (msk) {
                const u32x4 mk = *reinterpret_cast<const u32x4*>(msk + off);
                if constexpr (ESZ == 4) {
                    const f32x4 mf = __builtin_bit_cast(f32x4, mk);
                    if (!(mf.x > 0.f)) val.x = 0; if (!(mf.y > 0.f)) val.y = 0;
                    if (!(mf.z > 0.f)) val.z = 0; if (!(mf.w > 0.f)) val.w = 0;
                } else {
                    const uint32_t mm[4] = {mk.x, mk.y, mk.z, mk.w};
                    uint32_t vv[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!(wsu_bf16_to_f32(mm[e] & 0xFFFF) > 0.f)) vv[e] &= 0xFFFF0000u;
                        if (!(wsu_bf16_to_f32(mm[e] >> 16) > 0.f)) vv[e] &= 0x0000FFFFu;
                    }
                    val = mk_u4(vv[0], vv[1], vv[2], vv[3]);
                }
            }
            *reinterpret_cast<u32x4*>(ydst + off) = val;
        }
    }

    // ---- fused 2x2/2 max-pool with first-max-wins argmax ------------------------------------------------
    if (a.ypool) {
        const int hp = a.h >> 1, wp2 = a.w >> 1;
#pragma unroll
        for (int k = 0; k < (TH / 2) * (TW / 2) * VPP / NT; ++k) {
            const int i = tid + k * NT;
            const int pp = i / VPP, v = i % VPP;
            const int pr = pp / (TW / 2), pc = pp % (TW / 2);
            const int gy = (y0 >> 1) + pr, gx = (x0 >> 1) + pc;
            if (gy < hp && gx < wp2) {
                const char* base = smem + ((2 * pr) * TW + 2 * pc) * STRIDE + v * 16;
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(base);
                const u32x4 w1 = *reinterpret_cast<const u32x4*>(base + STRIDE);
                const u32x4 w2 = *reinterpret_cast<const u32x4*>(base + TW * STRIDE);
                const u32x4 w3 = *reinterpret_cast<const u32x4*>(base + (TW + 1) * STRIDE);
                const size_t eoff = ((size_t)(n * hp + gy) * wp2 + gx) * a.cout + cglob;   // element offset
                if constexpr (ESZ == 4) {
                    const float* f0 = reinterpret_cast<const float*>(&w0); const float* f1 = reinterpret_cast<const float*>(&w1);
                    const float* f2 = reinterpret_cast<const float*>(&w2); const float* f3 = reinterpret_cast<const float*>(&w3);
                    float o[4]; uint32_t idx = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float best = f0[e]; uint32_t bi = 0;
                        if (f1[e] > best || f1[e] != f1[e]) { best = f1[e]; bi = 1; }
                        if (f2[e] > best || f2[e] != f2[e]) { best = f2[e]; bi = 2; }
                        if (f3[e] > best || f3[e] != f3[e]) { best = f3[e]; bi = 3; }
                        o[e] = best; idx |= bi << (8 * e);
                    }
                    *reinterpret_cast<f32x4*>(a.ypool + (eoff + v * 4) * 4) = mk_f4(o[0], o[1], o[2], o[3]);
                    if (a.pidx) *reinterpret_cast<uint32_t*>(a.pidx + eoff + v * 4) = idx;
                } else {
                    const uint32_t u0[4] = {w0.x, w0.y, w0.z, w0.w}, u1[4] = {w1.x, w1.y, w1.z, w1.w};
                    const uint32_t u2[4] = {w2.x, w2.y, w2.z, w2.w}, u3[4] = {w3.x, w3.y, w3.z, w3.w};
                    uint32_t o[4]; uint32_t idx[2] = {0, 0};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int sh = (e & 1) * 16;
                        const uint16_t h0 = (u0[e >> 1] >> sh) & 0xFFFF, h1 = (u1[e >> 1] >> sh) & 0xFFFF;
                        const uint16_t h2 = (u2[e >> 1] >> sh) & 0xFFFF, h3 = (u3[e >> 1] >> sh) & 0xFFFF;
                        float best = wsu_bf16_to_f32(h0); uint16_t bb = h0; uint32_t bi = 0; float t;
                        t = wsu_bf16_to_f32(h1); if (t > best || t != t) { best = t; bb = h1; bi = 1; }
                        t = wsu_bf16_to_f32(h2); if (t > best || t != t) { best = t; bb = h2; bi = 2; }
                        t = wsu_bf16_to_f32(h3); if (t > best || t != t) { best = t; bb = h3; bi = 3; }
                        if (e & 1) o[e >> 1] |= (uint32_t)bb << 16; else o[e >> 1] = bb;
                        idx[e >> 2] |= bi << (8 * (e & 3));
                    }
                    *reinterpret_cast<u32x4*>(a.ypool + (eoff + v * 8) * 2) = mk_u4(o[0], o[1], o[2], o[3]);
                    if (a.pidx) *reinterpret_cast<u32x2*>(a.pidx + eoff + v * 8) = mk_u2(idx[0], idx[1]);
                }
            }
        }
    }
}

template <int MODE>
int launch_conv(const ConvArgs& a, hipStream_t s) {
    const int lds = Epi<MODE>::BYTES > LDS_MAIN ? Epi<MODE>::BYTES : LDS_MAIN;
    static bool attr_done = false;     // benign race: idempotent
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        attr_done = true;
    }
    const long long nblk = (long long)a.n * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk <= 0 || nblk > 0x7FFFFFFFLL) { wsu_set_error("conv3x3: grid of %lld workgroups out of range", nblk); return WSU_ERR_ARG; }
    hipLaunchKernelGGL(conv3x3_kernel<MODE>, dim3((unsigned)nblk), dim3(NT), lds, s, a);
    return wsu_check_launch("conv3x3_kernel");
}

// ---- weight packing: OIHW fp32 -> [cob][chunk][tap][granule][co 64][16 B] -----------------------------
// transpose_flip: pack Wd[ci][co][u'][v'] = W[co][ci][2-u'][2-v'] (data-gradient weights: roles of Cin/Cout swap).
template <int MODE>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, char* __restrict__ dst, int cin, int cout, int transpose_flip) {
    // logical problem after the optional role swap: K = kin input channels, M = mout output channels
    const int kin = transpose_flip ? cout : cin, mout = transpose_flip ? cin : cout;
    constexpr int CK = (MODE == WSU_MODE_BF16) ? 32 : 16;
    constexpr int EPG = (MODE == WSU_MODE_F32) ? 4 : 8;               // elements per granule
    const int nch = kin / CK;
    const long long total = (long long)(mout / WSU_COB) * nch * 9 * WSU_GRAN * WSU_COB * EPG;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int e = t % EPG; t /= EPG;
        const int co = t % WSU_COB; t /= WSU_COB;
        const int g = t % WSU_GRAN; t /= WSU_GRAN;
        const int tap = t % 9; t /= 9;
        const int c = t % nch; t /= nch;
        const int cb = (int)t;
        int ci, part = 0;
        if (MODE == WSU_MODE_F32) ci = c * CK + 4 * g + e;
        else if (MODE == WSU_MODE_BF16) ci = c * CK + 8 * g + e;
        else { ci = c * CK + 8 * (g & 1) + e; part = g >> 1; }
        const int m = cb * WSU_COB + co;
        const int u = tap / 3, v = tap % 3;
        float val;
        if (transpose_flip) val = w[(((size_t)ci * cin + m) * 3 + (2 - u)) * 3 + (2 - v)];   // W[co=ci_d][ci=m]
        else                val = w[(((size_t)m * cin + ci) * 3 + u) * 3 + v];
        if (MODE == WSU_MODE_F32) {
            reinterpret_cast<float*>(dst)[d] = val;
        } else {
            const float x = part ? wsu_bf16_lo_residual(val) : val;
            const __bf16 hv = (__bf16)x;
            reinterpret_cast<uint16_t*>(dst)[d] = __builtin_bit_cast(uint16_t, hv);
        }
    }
}

int pack_impl(const float* w, void* dst, int cin, int cout, int mode, int tf, void* stream) {
    const int kin = tf ? cout : cin, mout = tf ? cin : cout;
    WSU_REQUIRE(w && dst, "conv3x3_pack: null pointer");
    WSU_REQUIRE(mode >= 0 && mode <= 2, "conv3x3_pack: bad mode %d", mode);
    WSU_REQUIRE(kin > 0 && kin % wsu_chunk_channels(mode) == 0, "conv3x3_pack: reduction channels %d not a multiple of %d", kin, wsu_chunk_channels(mode));
    WSU_REQUIRE(mout > 0 && mout % WSU_COB == 0, "conv3x3_pack: output channels %d not a multiple of %d", mout, WSU_COB);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = 1024;
    if (mode == WSU_MODE_F32) hipLaunchKernelGGL(pack_conv3x3_kernel<WSU_MODE_F32>, dim3(blocks), dim3(256), 0, s, w, (char*)dst, cin, cout, tf);
    else if (mode == WSU_MODE_BF16X3) hipLaunchKernelGGL(pack_conv3x3_kernel<WSU_MODE_BF16X3>, dim3(blocks), dim3(256), 0, s, w, (char*)dst, cin, cout, tf);
    else hipLaunchKernelGGL(pack_conv3x3_kernel<WSU_MODE_BF16>, dim3(blocks), dim3(256), 0, s, w, (char*)dst, cin, cout, tf);
    return wsu_check_launch("pack_conv3x3_kernel");
}

}  // namespace

extern "C" {

size_t wsu_conv3x3_packed_bytes(int cin, int cout, int mode) {
    if (cin <= 0 || cout <= 0 || mode < 0 || mode > 2) return 0;
    const size_t per_elem = mode == WSU_MODE_BF16 ? 2 : 4;    // BF16X3 stores hi + lo bf16 = 4 bytes
    return (size_t)cin * cout * 9 * per_elem;
}

int wsu_conv3x3_pack(const float* w_oihw, void* w_packed, int cin, int cout, int mode, void* stream) {
    return pack_impl(w_oihw, w_packed, cin, cout, mode, 0, stream);
}

int wsu_conv3x3_pack_dgrad(const float* w_oihw, void* w_packed, int cin, int cout, int mode, void* stream) {
    return pack_impl(w_oihw, w_packed, cin, cout, mode, 1, stream);
}

// Extended launcher shared by the forward op and the data-gradient op (wsu_conv3x3_bwd_data in conv3x3_bwd.hip).
int wsu_conv3x3_launch_ex(const void* x1, const void* x2, const void* w_packed, const float* bias,
                          void* y, void* y2, int csplit, void* y_pool, uint8_t* pool_idx,
                          const void* relu_mask, const void* relu_mask2,
                          int n, int h, int w, int c1, int c2, int cout,
                          int mode, int relu, int pad_zero, void* stream) {
    WSU_REQUIRE(mode >= 0 && mode <= 2, "conv3x3: bad mode %d", mode);
    const int ck = wsu_chunk_channels(mode);
    WSU_REQUIRE(x1 && w_packed && y, "conv3x3: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3: bad shape n=%d h=%d w=%d (reflect pad 1 needs h,w >= 2)", n, h, w);
    WSU_REQUIRE(c1 > 0 && c1 % ck == 0, "conv3x3: c1=%d must be a positive multiple of %d in mode %d", c1, ck, mode);
    WSU_REQUIRE(c2 >= 0 && c2 % ck == 0 && (c2 == 0) == (x2 == nullptr), "conv3x3: c2=%d inconsistent with x2 / not a multiple of %d", c2, ck);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0, "conv3x3: cout=%d must be a multiple of %d", cout, WSU_COB);
    WSU_REQUIRE(csplit > 0 && csplit <= cout && csplit % WSU_COB == 0 && (csplit == cout) == (y2 == nullptr),
                "conv3x3: bad output split %d of %d", csplit, cout);
    WSU_REQUIRE((long long)n * h * w < 0x7FFFFFFFLL, "conv3x3: n*h*w overflows int32");
    WSU_REQUIRE(!(pool_idx && !y_pool), "conv3x3: pool_idx without y_pool");
    ConvArgs a;
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.wp = (const char*)w_packed; a.bias = bias;
    a.y = (char*)y; a.y2 = (char*)y2; a.ypool = (char*)y_pool; a.pidx = pool_idx;
    a.relu_mask = (const char*)relu_mask; a.relu_mask2 = (const char*)relu_mask2;
    a.n = n; a.h = h; a.w = w; a.c1 = c1; a.c2 = c2; a.cout = cout; a.csplit = csplit;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ncb = cout / WSU_COB;
    a.nch1 = c1 / ck; a.nch = (c1 + c2) / ck;
    a.relu = relu; a.pad_zero = pad_zero;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_F32) return launch_conv<WSU_MODE_F32>(a, s);
    if (mode == WSU_MODE_BF16X3) return launch_conv<WSU_MODE_BF16X3>(a, s);
    return launch_conv<WSU_MODE_BF16>(a, s);
}

int wsu_conv3x3_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias,
                    void* y, void* y_pool, uint8_t* pool_idx,
                    int n, int h, int w, int c1, int c2, int cout,
                    int mode, int relu, int pad_zero, void* stream) {
    return wsu_conv3x3_launch_ex(x1, x2, w_packed, bias, y, nullptr, cout, y_pool, pool_idx, nullptr, nullptr,
                                 n, h, w, c1, c2, cout, mode, relu, pad_zero, stream);
}

}  // extern "C"
