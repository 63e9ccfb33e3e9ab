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
// LDS (one workgroup = 8 waves for bf16x3 / bf16, 4 waves for f32; 2 workgroups per CU):
//   input tile   [4 granule planes][10 x 34 pixels][16 B]   halo resolved (reflect / zero) while staging
//   weight tile  [9 taps][4 granule planes][64 co][16 B]    straight copy of the packed weights
//   both granule-planar, so every fragment read is a contiguous 512-byte ds_read_b128 (conflict free).
// MFMA tile: v_mfma_f32_32x32x16_bf16 (v_mfma_f32_32x32x2_f32 for mode f32); template flag S16 selects
// v_mfma_f32_16x16x32_bf16 (lane = k-group x 16 rows; C/D: 4 consecutive channels of one pixel per lane), the default for
// bf16 storage (profiles/r01/conv3x3_ablation.md, second pass).
// Staging is register-double-buffered: chunk c+1's global loads are in flight while chunk c is multiplied.
// The epilogue re-uses the LDS as a [pixel][channel] tile: bias + ReLU (+ReLU-mask for the data-gradient
// pass), coalesced 16-byte NHWC stores, and the fused 2x2 max-pool with first-max-wins argmax.
#include "wsu_device.h"
#include <cstdlib>

namespace {

constexpr int TW = 32, TH = 8;
constexpr int IW = TW + 2, IH = TH + 2;
constexpr int NPIX_IN = IW * IH;                         // 340
constexpr int PLANE_IN = NPIX_IN * 16 + 96;              // 5536 B: planes 8 dwords apart mod 32 banks
constexpr int LDS_IN = WSU_GRAN * PLANE_IN;              // 22144
constexpr int LDS_W = 9 * WSU_GRAN * WSU_COB * 16;       // 36864
[[maybe_unused]] constexpr int LDS_MAIN = LDS_IN + LDS_W;                 // 59008
constexpr int W_ITEMS = LDS_W / 16;                      // 2304 x 16 B
[[maybe_unused]] constexpr int IN_ITEMS = NPIX_IN * WSU_GRAN;             // 1360 x 16 B (680 x 32 B for BF16X3)
// Workgroup shapes on the same 8x32-pixel x 64-channel tile:
//   NW = 4: 256 threads, wave tile 64 co x 64 px (4 MFMA tiles), 2 waves/SIMD
//   NW = 8: 512 threads, wave tile 32 co x 64 px (2 MFMA tiles), <= 128 VGPRs -> 4 waves/SIMD
//   NW = 16: 1024 threads on a 16x32-pixel tile, wave tile 32 co x 64 px, one workgroup per CU: the weight tile is shared by 512
//            pixels and the halo shrinks from 1.33 to 1.20 (-35 % staged bytes per output)
template <int NW> struct Shape {
    static constexpr int NT = NW * 64;
    static constexpr int MT = NW == 4 ? 2 : 1;           // 32-channel MFMA row tiles per wave
    static constexpr int TH = NW == 16 ? 16 : 8;         // output rows per workgroup
    static constexpr int IH = TH + 2;
    static constexpr int NPIX_IN = IW * IH;              // 340 / 612
    static constexpr int PLANE_IN = NPIX_IN * 16 + 96;   // planes 8 dwords apart mod 32 banks
    static constexpr int LDS_IN = WSU_GRAN * PLANE_IN;
    static constexpr int LDS_MAIN = LDS_IN + LDS_W;
    static constexpr int IN_ITEMS = NPIX_IN * WSU_GRAN;
    static constexpr int W_VEC = (W_ITEMS + NT - 1) / NT;
    static constexpr int IN_VEC = (IN_ITEMS + NT - 1) / NT;
    static constexpr int IN_VEC3 = (NPIX_IN * 2 + NT - 1) / NT;   // BF16X3 items of 32 B
    static constexpr int IN_VEC48 = (NPIX_IN * 3 + NT - 1) / NT;  // F16F8: a stored chunk is 3 x 16 B (the 4th granule plane is derived)
    static constexpr int ST_IN = IN_VEC > 2 * IN_VEC3 ? IN_VEC : 2 * IN_VEC3;
};

// Diagnostic in-kernel stamps (WSU_CONV_ABLATE bit 512; never in production): s_memtime per phase of the first 2048
// workgroups, read back with wsu_debug_read_stamps().  Values go to a buffer nothing else reads.
#define WSU_NSTAMP 32
__device__ unsigned long long g_stamps[2048 * WSU_NSTAMP];
#define WSU_STAMP(k) do { if ((a.ablate & 512) && blockIdx.x < 2048 && threadIdx.x == 0) \
        g_stamps[blockIdx.x * WSU_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

struct ConvArgs {
    const char* x1; const char* x2; const char* wp; const float* bias;
    char* y; char* y2; char* ypool; uint8_t* pidx; const char* relu_mask; const char* relu_mask2;
    int n, h, w, c1, c2, cout, csplit;
    int tiles_x, tiles_y, ncb, nch1, nch;
    int relu, pad_zero;
    int ablate;        // timing-only experiment mask (WSU_CONV_ABLATE), 0 in production
    // fused 1x1 head + sigmoid (outconv, unet.py:189) on this layer's 64 output channels; y may then be null
    const float* head_w; const float* head_b; float* head_out; float* head_logit; int head_cout;    // fused first layer (e11, unet.py:141): the 64 input channels of THIS conv are computed while staging from a 1-plane image
    // (img: (N,1,H,W) fp32, w1: (64,1,3,3), b1: (64) or null); x1 is then null
    const float* img; const float* w1; const float* b1;
    // API mode WSU_MODE_BF16X3S: activations stored already split (per pixel and 16-channel chunk: hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15,
    // 4 x 16 B = the fp32 chunk size).  The input side is the template flag PS (staging becomes a plain copy), the output side this flag.
    int out_split;
};

template <int MODE> struct Epi {
    static constexpr int ESZ = (MODE == WSU_MODE_BF16) ? 2 : 4;
    static constexpr int STRIDE = WSU_COB * ESZ + 16;    // bytes per pixel in the epilogue tile
    static constexpr int BYTES = TH * TW * STRIDE;       // 8-row tile; the 16-row tile needs twice that
    static constexpr int VPP = WSU_COB * ESZ / 16;       // 16-byte pieces per pixel
};


// Global -> registers for chunk c (input tile items + this workgroup's packed-weight slice).
template <int MODE, int NW, bool F1 = false, bool PS = false>
__device__ __forceinline__ void stage_load(const ConvArgs& a, int cb, int c, int tid, const int (&pixidx)[Shape<NW>::IN_VEC],
                                           u32x4 (&st_in)[Shape<NW>::ST_IN], u32x4 (&st_w)[Shape<NW>::W_VEC]) {
    constexpr int NT = Shape<NW>::NT, W_VEC = Shape<NW>::W_VEC;
    constexpr int ESZ = Epi<MODE>::ESZ;
    constexpr int CK = (MODE == WSU_MODE_BF16) ? 32 : 16;
    constexpr bool SPLIT_HERE = (MODE == WSU_MODE_BF16X3 || MODE == WSU_MODE_F16F8) && !PS;     // fp32 in HBM, split while committing (2 items of 32 B per pixel)
    constexpr int NLOOP = SPLIT_HERE ? Shape<NW>::IN_VEC3 : (MODE == WSU_MODE_F16F8 ? Shape<NW>::IN_VEC48 : Shape<NW>::IN_VEC);
    const char* src; int csrc, ch0;
    if (c < a.nch1) { src = a.x1; csrc = a.c1; ch0 = c * CK; }
    else            { src = a.x2; csrc = a.c2; ch0 = (c - a.nch1) * CK; }
    if constexpr (!F1)                                          // fused first layer: the activations are computed in the commit
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        const int p = pixidx[j];
        if constexpr (SPLIT_HERE) {
            const int sub = (tid + j * NT) & 1;
            u32x4 v0 = mk_u4(0, 0, 0, 0), v1 = v0;
            if (p >= 0) {
                const u32x4* g = reinterpret_cast<const u32x4*>(src + ((size_t)p * csrc + ch0) * 4 + sub * 32);
                v0 = g[0]; v1 = g[1];
            }
            st_in[2 * j] = v0; st_in[2 * j + 1] = v1;
        } else if constexpr (MODE == WSU_MODE_F16F8) {
            const int sub = (tid + j * NT) % 3;                  // 48 stored bytes per pixel and chunk: f16 0-7 | f16 8-15 | residuals
            u32x4 v = mk_u4(0, 0, 0, 0);
            if (p >= 0) v = *reinterpret_cast<const u32x4*>(src + ((size_t)p * csrc + ch0) * 3 + sub * 16);
            st_in[j] = v;
        } else {
            const int sub = (tid + j * NT) & 3;
            u32x4 v = mk_u4(0, 0, 0, 0);
            if (p >= 0) v = *reinterpret_cast<const u32x4*>(src + ((size_t)p * csrc + ch0) * ESZ + sub * 16);
            st_in[j] = v;
        }
    }
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp + ((size_t)cb * a.nch + c) * LDS_W);
#pragma unroll
    for (int k = 0; k < W_VEC; ++k)
        if (W_ITEMS % NT == 0 || tid + k * NT < W_ITEMS) st_w[k] = wsrc[tid + k * NT];
}

// ---- fused first layer: LDS extras behind the main region -------------------------------------------------------
//   P  [input-tile pixel][12]  the 3x3 image neighbourhood (reflect) of the e11 output pixel that this tile position maps to
//   W1T [9 taps][64 channels]  e11 taps, tap-major for the packed FMAs of e11_oct (the region keeps its 64 x 48 bytes), B1 [64]
constexpr int F1_W1_OFF(int npix) { return npix * 48; }
constexpr int F1_B1_OFF(int npix) { return npix * 48 + 64 * 48; }
constexpr int F1_BYTES(int npix) { return npix * 48 + 64 * 48 + 256; }

// relu(b + sum_t x_t * w_t) for channels ch..ch+7 (ch..ch+3), taps in the order of conv3x3_first_kernel (bitwise the same values).
// Two channels per v_pk_fma_f32: the weights sit tap-major in LDS (W1T [9 taps][64 ch]) so that a channel pair is one aligned register
// pair, and the pixel value is broadcast by op_sel -- 36 + 8 VALU instructions per 8 channels instead of 80 (the fused first layer is
// bound by this computation, not by its matrix work).
__device__ __forceinline__ void e11_oct(const float* P12, const float* W1T, const float* B1, int ch, f32x4& r0, f32x4& r1) {
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(P12), p1 = *reinterpret_cast<const f32x4*>(P12 + 4), p2 = *reinterpret_cast<const f32x4*>(P12 + 8);
    const float p[9] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x};
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(B1 + ch), b1 = *reinterpret_cast<const f32x4*>(B1 + ch + 4);
    f32x2 a0 = {b0.x, b0.y}, a1 = {b0.z, b0.w}, a2 = {b1.x, b1.y}, a3 = {b1.z, b1.w};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(W1T + t * 64 + ch), w1 = *reinterpret_cast<const f32x4*>(W1T + t * 64 + ch + 4);
        const f32x2 pp = {p[t], p[t]};
        a0 = __builtin_elementwise_fma(pp, (f32x2){w0.x, w0.y}, a0);
        a1 = __builtin_elementwise_fma(pp, (f32x2){w0.z, w0.w}, a1);
        a2 = __builtin_elementwise_fma(pp, (f32x2){w1.x, w1.y}, a2);
        a3 = __builtin_elementwise_fma(pp, (f32x2){w1.z, w1.w}, a3);
    }
    r0 = mk_f4(fmaxf(a0.x, 0.f), fmaxf(a0.y, 0.f), fmaxf(a1.x, 0.f), fmaxf(a1.y, 0.f));
    r1 = mk_f4(fmaxf(a2.x, 0.f), fmaxf(a2.y, 0.f), fmaxf(a3.x, 0.f), fmaxf(a3.y, 0.f));
}
__device__ __forceinline__ f32x4 e11_quad(const float* P12, const float* W1T, const float* B1, int ch) {
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(P12), p1 = *reinterpret_cast<const f32x4*>(P12 + 4), p2 = *reinterpret_cast<const f32x4*>(P12 + 8);
    const float p[9] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x};
    const f32x4 b = *reinterpret_cast<const f32x4*>(B1 + ch);
    f32x2 a0 = {b.x, b.y}, a1 = {b.z, b.w};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W1T + t * 64 + ch);
        const f32x2 pp = {p[t], p[t]};
        a0 = __builtin_elementwise_fma(pp, (f32x2){w.x, w.y}, a0);
        a1 = __builtin_elementwise_fma(pp, (f32x2){w.z, w.w}, a1);
    }
    return mk_f4(fmaxf(a0.x, 0.f), fmaxf(a0.y, 0.f), fmaxf(a1.x, 0.f), fmaxf(a1.y, 0.f));
}

// Chunk c of the virtual 64-channel input, computed into the granule-planar tile (same LDS contents as staging e11's output)
template <int MODE, int NW>
__device__ __forceinline__ void stage_commit_first(char* smem, int tid, int c, const int (&pixidx)[Shape<NW>::IN_VEC], const int (&ldsoff)[Shape<NW>::IN_VEC],
                                                   const u32x4 (&st_w)[Shape<NW>::W_VEC]) {
    constexpr int NT = Shape<NW>::NT, W_VEC = Shape<NW>::W_VEC;
    constexpr int PLANE_IN = Shape<NW>::PLANE_IN, LDS_IN = Shape<NW>::LDS_IN, NPIX_IN = Shape<NW>::NPIX_IN;
    constexpr int NLOOP = (MODE == WSU_MODE_BF16X3 || MODE == WSU_MODE_F16F8) ? Shape<NW>::IN_VEC3 : Shape<NW>::IN_VEC;
    const float* P = reinterpret_cast<const float*>(smem + Shape<NW>::LDS_MAIN);
    const float* W1 = reinterpret_cast<const float*>(smem + Shape<NW>::LDS_MAIN + F1_W1_OFF(NPIX_IN));
    const float* B1 = reinterpret_cast<const float*>(smem + Shape<NW>::LDS_MAIN + F1_B1_OFF(NPIX_IN));
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        if (pixidx[j] != -2) {
            const int i = tid + j * NT;
            if constexpr (MODE == WSU_MODE_BF16X3) {
                const int pix = i >> 1, ch0 = c * 16 + (i & 1) * 8;
                u32x4 hi, lo; f32x4 q0, q1;
                e11_oct(P + pix * 12, W1, B1, ch0, q0, q1);
                wsu_split8(q0, q1, hi, lo);
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = hi;
                *reinterpret_cast<u32x4*>(smem + ldsoff[j] + 2 * PLANE_IN) = lo;
            } else if constexpr (MODE == WSU_MODE_F16F8) {
                const int pix = i >> 1, half = i & 1, ch0 = c * 16 + half * 8;
                uint32_t h0, h1, h2, h3, l0, l1, x0, x1; f32x4 q0, q1;
                e11_oct(P + pix * 12, W1, B1, ch0, q0, q1);
                wsu_split4_f16f8(q0, WSU_F8_XLO_DIV, WSU_F8_X_DIV, h0, h1, l0, x0);
                wsu_split4_f16f8(q1, WSU_F8_XLO_DIV, WSU_F8_X_DIV, h2, h3, l1, x1);
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = mk_u4(h0, h1, h2, h3);                            // plane `half`
                *reinterpret_cast<u32x2*>(smem + 2 * PLANE_IN + pix * 16 + half * 8) = mk_u2(l0, l1);
                *reinterpret_cast<u32x2*>(smem + 3 * PLANE_IN + pix * 16 + half * 8) = mk_u2(x0, x1);
            } else if constexpr (MODE == WSU_MODE_F32) {
                const int pix = i >> 2, ch0 = c * 16 + (i & 3) * 4;
                *reinterpret_cast<f32x4*>(smem + ldsoff[j]) = e11_quad(P + pix * 12, W1, B1, ch0);
            } else {
                const int pix = i >> 2, ch0 = c * 32 + (i & 3) * 8;
                f32x4 q0, q1;
                e11_oct(P + pix * 12, W1, B1, ch0, q0, q1);
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = mk_u4(wsu_pack_bf16x2(q0.x, q0.y), wsu_pack_bf16x2(q0.z, q0.w),
                                                                   wsu_pack_bf16x2(q1.x, q1.y), wsu_pack_bf16x2(q1.z, q1.w));
            }
        }
    }
    u32x4* wdst = reinterpret_cast<u32x4*>(smem + LDS_IN);
#pragma unroll
    for (int k = 0; k < W_VEC; ++k)
        if (W_ITEMS % NT == 0 || tid + k * NT < W_ITEMS) wdst[tid + k * NT] = st_w[k];
}

// Registers -> LDS (granule-planar input tile, linear weight tile); BF16X3 splits fp32 into bf16 hi/lo here.
template <int MODE, int NW, bool PS = false>
__device__ __forceinline__ void stage_commit(char* smem, int tid, const int (&pixidx)[Shape<NW>::IN_VEC], const int (&ldsoff)[Shape<NW>::IN_VEC],
                                             const u32x4 (&st_in)[Shape<NW>::ST_IN], const u32x4 (&st_w)[Shape<NW>::W_VEC]) {
    constexpr int NT = Shape<NW>::NT, W_VEC = Shape<NW>::W_VEC;
    constexpr int PLANE_IN = Shape<NW>::PLANE_IN, LDS_IN = Shape<NW>::LDS_IN;
    constexpr bool SPLIT_HERE = (MODE == WSU_MODE_BF16X3 || MODE == WSU_MODE_F16F8) && !PS;
    constexpr int NLOOP = SPLIT_HERE ? Shape<NW>::IN_VEC3 : (MODE == WSU_MODE_F16F8 ? Shape<NW>::IN_VEC48 : Shape<NW>::IN_VEC);
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        if (pixidx[j] != -2) {
            if constexpr (SPLIT_HERE && MODE == WSU_MODE_F16F8) {
                // fp32 storage (API mode F16F8X): 8 channels of one pixel -> f16 piece, 8 residuals, 8 e4m3 copies (planes half, 2, 3)
                const int half = ldsoff[j] >= PLANE_IN ? 1 : 0, pixoff = ldsoff[j] - half * PLANE_IN;
                uint32_t h0, h1, h2, h3, l0, l1, x0, x1;
                wsu_split4_f16f8(__builtin_bit_cast(f32x4, st_in[2 * j]), WSU_F8_XLO_DIV, WSU_F8_X_DIV, h0, h1, l0, x0);
                wsu_split4_f16f8(__builtin_bit_cast(f32x4, st_in[2 * j + 1]), WSU_F8_XLO_DIV, WSU_F8_X_DIV, h2, h3, l1, x1);
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = mk_u4(h0, h1, h2, h3);
                *reinterpret_cast<u32x2*>(smem + 2 * PLANE_IN + pixoff + half * 8) = mk_u2(l0, l1);
                *reinterpret_cast<u32x2*>(smem + 3 * PLANE_IN + pixoff + half * 8) = mk_u2(x0, x1);
            } else if constexpr (SPLIT_HERE) {
                u32x4 hi, lo;
                wsu_split8(__builtin_bit_cast(f32x4, st_in[2 * j]), __builtin_bit_cast(f32x4, st_in[2 * j + 1]), hi, lo);
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = hi;
                *reinterpret_cast<u32x4*>(smem + ldsoff[j] + 2 * PLANE_IN) = lo;
            } else if constexpr (MODE == WSU_MODE_F16F8) {
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = st_in[j];
                if (ldsoff[j] < 2 * PLANE_IN) {                  // an f16 piece: its 8 e4m3 copies go to half of the pixel's slot in plane 3
                    const int half = ldsoff[j] >= PLANE_IN ? 1 : 0;
                    *reinterpret_cast<u32x2*>(smem + 3 * PLANE_IN + (ldsoff[j] - half * PLANE_IN) + half * 8) = wsu_f16x8_to_fp8(st_in[j]);
                }
            } else {
                *reinterpret_cast<u32x4*>(smem + ldsoff[j]) = st_in[j];
            }
        }
    }
    u32x4* wdst = reinterpret_cast<u32x4*>(smem + LDS_IN);
#pragma unroll
    for (int k = 0; k < W_VEC; ++k)
        if (W_ITEMS % NT == 0 || tid + k * NT < W_ITEMS) wdst[tid + k * NT] = st_w[k];
}

// F16F8 stores from the fp32 [pixel][channel] LDS tile.  The 3 x 16 B of a (pixel, 16-channel chunk) are produced by one thread, so
// storing them directly would put 16 B into every 64 B per instruction (half-empty write requests: measured +11 % on the store-heavy
// first layer).  Instead the tile is encoded IN PLACE (an encoded chunk fits in its fp32 source) and then copied out with 12
// consecutive lanes per pixel: every store instruction writes whole 192-byte pixel rows.  The fused 2x2 max-pool reads the fp32
// tile first and stores its (4x smaller) output directly.
template <int NT, int TH, int STRIDE>
__device__ __forceinline__ void store_f16f8(const ConvArgs& a, char* smem, int tid, int n, int y0, int x0, int cglob, char* ydst, int ych, int ycoff) {
    // fused 2x2 max-pool: one (pooled pixel, chunk) item per thread, encoded into registers now, staged and stored after the main tile
    static_assert((TH / 2) * (TW / 2) * 4 <= NT, "one pooled item per thread");
    u32x4 p_hi0, p_hi1, p_lo8;
    const bool pool_item = a.ypool && tid < (TH / 2) * (TW / 2) * 4;
    if (pool_item) {
        const int pp = tid >> 2, j = tid & 3;
        const int pr = pp / (TW / 2), pc = pp % (TW / 2);
        const f32x4* b0 = reinterpret_cast<const f32x4*>(smem + ((2 * pr) * TW + 2 * pc) * STRIDE + j * 64);
        f32x4 m0 = b0[0], m1 = b0[1], m2 = b0[2], m3 = b0[3];
#pragma unroll
        for (int wdx = 1; wdx < 4; ++wdx) {                            // window order of the fp32 path (first max wins; NaN propagates)
            const f32x4* bq = reinterpret_cast<const f32x4*>(smem + ((2 * pr + (wdx >> 1)) * TW + 2 * pc + (wdx & 1)) * STRIDE + j * 64);
            const f32x4 q0 = bq[0], q1 = bq[1], q2 = bq[2], q3 = bq[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (q0[e] > m0[e] || q0[e] != q0[e]) m0[e] = q0[e];
                if (q1[e] > m1[e] || q1[e] != q1[e]) m1[e] = q1[e];
                if (q2[e] > m2[e] || q2[e] != q2[e]) m2[e] = q2[e];
                if (q3[e] > m3[e] || q3[e] != q3[e]) m3[e] = q3[e];
            }
        }
        wsu_split16_f16f8(m0, m1, m2, m3, p_hi0, p_hi1, p_lo8);
    }
    // piece p (0..11) of a pixel's 192 stored bytes = chunk p / 3, 16-byte piece p % 3; in the LDS tile a chunk keeps its 64-byte slot
    if (ydst) {
        __syncthreads();                                               // pool / head readers of the fp32 tile are done
        for (int i = tid; i < TH * TW * 4; i += NT) {
            u32x4* row = reinterpret_cast<u32x4*>(smem + (i >> 2) * STRIDE + (i & 3) * 64);
            u32x4 hi0, hi1, lo8;
            wsu_split16_f16f8(__builtin_bit_cast(f32x4, row[0]), __builtin_bit_cast(f32x4, row[1]), __builtin_bit_cast(f32x4, row[2]),
                              __builtin_bit_cast(f32x4, row[3]), hi0, hi1, lo8);
            row[0] = hi0; row[1] = hi1; row[2] = lo8;
        }
        __syncthreads();
        for (int i = tid; i < TH * TW * 12; i += NT) {
            const int px = i / 12, piece = i - 12 * px;
            const int r = px / TW, c = px % TW;
            if (y0 + r < a.h && x0 + c < a.w)
                *reinterpret_cast<u32x4*>(ydst + (((size_t)(n * a.h + y0 + r) * a.w + x0 + c) * ych + ycoff) * 3 + piece * 16) =
                    *reinterpret_cast<const u32x4*>(smem + px * STRIDE + (piece / 3) * 64 + (piece % 3) * 16);
        }
    }
    if (a.ypool) {
        __syncthreads();                                               // the tile has been read out (or was never needed): reuse its first rows
        if (pool_item) {
            u32x4* row = reinterpret_cast<u32x4*>(smem + (tid >> 2) * STRIDE + (tid & 3) * 64);
            row[0] = p_hi0; row[1] = p_hi1; row[2] = p_lo8;
        }
        __syncthreads();
        const int hp = a.h >> 1, wp2 = a.w >> 1;
        for (int i = tid; i < (TH / 2) * (TW / 2) * 12; i += NT) {
            const int pp = i / 12, piece = i - 12 * pp;
            const int gy = (y0 >> 1) + pp / (TW / 2), gx = (x0 >> 1) + pp % (TW / 2);
            if (gy < hp && gx < wp2)
                *reinterpret_cast<u32x4*>(a.ypool + (((size_t)(n * hp + gy) * wp2 + gx) * a.cout + cglob) * 3 + piece * 16) =
                    *reinterpret_cast<const u32x4*>(smem + pp * STRIDE + (piece / 3) * 64 + (piece % 3) * 16);
        }
    }
    WSU_STAMP(27);
    if ((a.ablate & 512) && blockIdx.x < 2048 && threadIdx.x == 0) g_stamps[blockIdx.x * WSU_NSTAMP + 31] = __builtin_amdgcn_s_memrealtime();
}

// Everything behind the [pixel][channel] LDS tile (TH x 32 pixels, Epi<MODE>::STRIDE bytes per pixel): fused 1x1 head, coalesced
// NHWC stores with the optional ReLU mask, fused 2x2 max-pool with first-max-wins argmax.  Shared by the direct and the Winograd kernel.
template <int MODE, int NT, int TH>
__device__ __forceinline__ void tile_epilogue(const ConvArgs& a, char* smem, int tid, int n, int y0, int x0, int cb) {
    constexpr int ESZ = Epi<MODE>::ESZ;
    constexpr int STRIDE = Epi<MODE>::STRIDE;
    // ---- coalesced NHWC store (16 B per thread), optional ReLU mask of the data-gradient pass --------
    constexpr int VPP = Epi<MODE>::VPP;
    const int cglob = cb * WSU_COB;                                   // first output channel of this block
    char* ydst = a.y; int ych = a.csplit, ycoff = cglob;
    const char* msk = a.relu_mask;
    if (cglob >= a.csplit) { ydst = a.y2; ych = a.cout - a.csplit; ycoff = cglob - a.csplit; msk = a.relu_mask2; }
    if (a.head_w && tid < TH * TW) {
        // fused head: one thread per pixel, 64-wide dot per output plane from the LDS tile, sigmoid, NCHW fp32 store
        const int r = tid / TW, c = tid % TW;
        if (y0 + r < a.h && x0 + c < a.w) {
            float z[4];
#pragma unroll
            for (int co = 0; co < 4; ++co) z[co] = (co < a.head_cout && a.head_b) ? a.head_b[co] : 0.f;
#pragma unroll
            for (int v = 0; v < VPP; ++v) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(smem + tid * STRIDE + v * 16);
                float xv[16 / ESZ];
                if constexpr (ESZ == 4) { const f32x4 f = __builtin_bit_cast(f32x4, raw); xv[0] = f.x; xv[1] = f.y; xv[2] = f.z; xv[3] = f.w; }
                else {
                    const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[e] = wsu_bf16_to_f32((u[e >> 1] >> ((e & 1) * 16)) & 0xFFFF);
                }
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    if (co < a.head_cout)
#pragma unroll
                        for (int e = 0; e < 16 / ESZ; ++e) z[co] = fmaf(xv[e], a.head_w[co * WSU_COB + v * (16 / ESZ) + e], z[co]);
            }
            const size_t hw = (size_t)a.h * a.w, pix = (size_t)(y0 + r) * a.w + x0 + c;
#pragma unroll
            for (int co = 0; co < 4; ++co)
                if (co < a.head_cout) {
                    const size_t o = ((size_t)n * a.head_cout + co) * hw + pix;
                    if (a.head_logit) a.head_logit[o] = z[co];
                    a.head_out[o] = 1.f / (1.f + expf(-z[co]));
                }
        }
    }
    if constexpr (MODE == WSU_MODE_F16F8) {
        if (a.out_split) { store_f16f8<NT, TH, STRIDE>(a, smem, tid, n, y0, x0, cglob, ydst, ych, ycoff); return; }   // else fp32 stores (F16F8X)
    }
    if constexpr (MODE == WSU_MODE_BF16X3) {
        if (a.out_split) {
            // ---- pre-split stores (mode BF16X3S): per pixel and 16-channel chunk  hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15; one item = 8 channels
            if (ydst)
                for (int i = tid; i < TH * TW * 8; i += NT) {
                    const int px = i >> 3, g8 = i & 7;
                    const int r = px / TW, c = px % TW;
                    if (y0 + r < a.h && x0 + c < a.w) {
                        const float* row = reinterpret_cast<const float*>(smem + px * STRIDE) + 8 * g8;
                        u32x4 hi, lo;
                        wsu_split8(*reinterpret_cast<const f32x4*>(row), *reinterpret_cast<const f32x4*>(row + 4), hi, lo);
                        char* dst = ydst + (((size_t)(n * a.h + y0 + r) * a.w + x0 + c) * ych + ycoff) * 4 + (g8 >> 1) * 64 + (g8 & 1) * 16;
                        *reinterpret_cast<u32x4*>(dst) = hi;
                        *reinterpret_cast<u32x4*>(dst + 32) = lo;
                    }
                }
            if (a.ypool) {
                const int hp = a.h >> 1, wp2 = a.w >> 1;
                for (int i = tid; i < (TH / 2) * (TW / 2) * 8; i += NT) {
                    const int pp = i >> 3, g8 = i & 7;
                    const int pr = pp / (TW / 2), pc = pp % (TW / 2);
                    const int gy = (y0 >> 1) + pr, gx = (x0 >> 1) + pc;
                    if (gy < hp && gx < wp2) {
                        const float* b0 = reinterpret_cast<const float*>(smem + ((2 * pr) * TW + 2 * pc) * STRIDE) + 8 * g8;
                        f32x4 m0 = *reinterpret_cast<const f32x4*>(b0), m1 = *reinterpret_cast<const f32x4*>(b0 + 4);
#pragma unroll
                        for (int wdx = 1; wdx < 4; ++wdx) {                // window order of the fp32 path (first max wins; NaN propagates)
                            const float* bq = reinterpret_cast<const float*>(smem + ((2 * pr + (wdx >> 1)) * TW + 2 * pc + (wdx & 1)) * STRIDE) + 8 * g8;
                            const f32x4 q0 = *reinterpret_cast<const f32x4*>(bq), q1 = *reinterpret_cast<const f32x4*>(bq + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (q0[e] > m0[e] || q0[e] != q0[e]) m0[e] = q0[e];
                                if (q1[e] > m1[e] || q1[e] != q1[e]) m1[e] = q1[e];
                            }
                        }
                        u32x4 hi, lo;
                        wsu_split8(m0, m1, hi, lo);
                        char* dst = a.ypool + (((size_t)(n * hp + gy) * wp2 + gx) * a.cout + cglob) * 4 + (g8 >> 1) * 64 + (g8 & 1) * 16;
                        *reinterpret_cast<u32x4*>(dst) = hi;
                        *reinterpret_cast<u32x4*>(dst + 32) = lo;
                    }
                }
            }
            return;
        }
    }
    if (ydst)
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

    WSU_STAMP(27);
    if ((a.ablate & 512) && blockIdx.x < 2048 && threadIdx.x == 0) g_stamps[blockIdx.x * WSU_NSTAMP + 31] = __builtin_amdgcn_s_memrealtime();
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

template <int MODE, int NW, bool S16 = false, bool F1 = false, bool PS = false>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 4 : 2) void conv3x3_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ESZ = Epi<MODE>::ESZ;
    constexpr int NT = Shape<NW>::NT, MT = Shape<NW>::MT, IN_VEC = Shape<NW>::IN_VEC;
    constexpr int TH = Shape<NW>::TH, NPIX_IN = Shape<NW>::NPIX_IN, PLANE_IN = Shape<NW>::PLANE_IN, LDS_IN = Shape<NW>::LDS_IN;
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
    constexpr bool SPLIT_HERE = (MODE == WSU_MODE_BF16X3 || MODE == WSU_MODE_F16F8) && !PS;     // F1 computes its input and always splits here
    static_assert(MODE != WSU_MODE_F16F8 || (!(PS && F1) && (NW == 8 || NW == 4 || NW == 16) && !S16), "F16F8: stored-split, fp32 or self-computed input");
    constexpr bool STORED48 = MODE == WSU_MODE_F16F8 && PS;         // 3 stored pieces of 16 B per pixel and chunk
    constexpr int NITEMS = SPLIT_HERE ? NPIX_IN * 2 : (STORED48 ? NPIX_IN * 3 : NPIX_IN * 4);
    constexpr int NLOOP = SPLIT_HERE ? Shape<NW>::IN_VEC3 : (STORED48 ? Shape<NW>::IN_VEC48 : IN_VEC);
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        const int i = tid + j * NT;
        const int pix = SPLIT_HERE ? (i >> 1) : (STORED48 ? i / 3 : (i >> 2));
        const int sub = SPLIT_HERE ? (i & 1) : (STORED48 ? i - 3 * pix : (i & 3));
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

    u32x4 st_in[Shape<NW>::ST_IN];
    u32x4 st_w[Shape<NW>::W_VEC];
    // ---- main loop ------------------------------------------------------------------------------------
    const int wv = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int rowpair = NW == 4 ? wv : (wv >> 1);                    // output rows 2*rowpair, 2*rowpair + 1
    const int mbase = NW == 4 ? 0 : (wv & 1) * 32;                   // first output channel of this wave
    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    const char* ldsA = smem + LDS_IN + (mbase + l31) * 16;           // + ((tap*4+g)*64 + mt*32)*16
    const char* ldsB = smem + ((2 * rowpair) * IW + l31) * 16;       // + g*PLANE_IN + ((nt+dy)*IW + dx)*16
    // 16x16x32 variant (S16): lane = (k-group kg = lane>>4 -> granule plane, row/column l15 = lane&15); per wave MT*2 channel
    // tiles x 4 pixel tiles (output row q = pt>>1, column half pt&1) of 16x16, 4 accumulator registers each.
    const int l15 = lane & 15, kg = lane >> 4;
    f32x4 acc16[MT * 2][4];
#pragma unroll
    for (int m = 0; m < MT * 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc16[m][q] = mk_f4(0.f, 0.f, 0.f, 0.f);
    const char* ldsA16 = smem + LDS_IN + (mbase + l15) * 16;         // + ((tap*4+plane)*64 + ct*16)*16
    const char* ldsB16 = smem + ((2 * rowpair) * IW + l15) * 16;     // + plane*PLANE_IN + ((q+dy)*IW + dx + half*16)*16

    if constexpr (F1) {
        // image neighbourhoods of the tile's 340 input positions: position -> e11 output pixel (reflect) -> its 3x3 window (reflect)
        float* P = reinterpret_cast<float*>(smem + Shape<NW>::LDS_MAIN);
        float* W1 = reinterpret_cast<float*>(smem + Shape<NW>::LDS_MAIN + F1_W1_OFF(NPIX_IN));
        float* B1 = reinterpret_cast<float*>(smem + Shape<NW>::LDS_MAIN + F1_B1_OFF(NPIX_IN));
        const float* img = a.img + (size_t)n * a.h * a.w;
        for (int i = tid; i < NPIX_IN * 12; i += NT) {
            const int pix = i / 12, t = i - pix * 12;
            float v = 0.f;
            if (t < 9) {
                const int r = pix / IW, c = pix - r * IW;
                const int yy = wsu_reflect(y0 - 1 + r, a.h), xx = wsu_reflect(x0 - 1 + c, a.w);
                v = img[(size_t)wsu_reflect(yy + t / 3 - 1, a.h) * a.w + wsu_reflect(xx + t % 3 - 1, a.w)];
            }
            P[i] = v;
        }
        for (int i = tid; i < 9 * 64; i += NT) { const int t = i >> 6, ch = i & 63; W1[i] = a.w1[ch * 9 + t]; }      // tap-major (e11_oct)
        if (tid < 64) B1[tid] = a.b1 ? a.b1[tid] : 0.f;
    }
    WSU_STAMP(0);
    if ((a.ablate & 512) && blockIdx.x < 2048 && threadIdx.x == 0) g_stamps[blockIdx.x * WSU_NSTAMP + 30] = __builtin_amdgcn_s_memrealtime();
    stage_load<MODE, NW, F1, PS>(a, cb, 0, tid, pixidx, st_in, st_w);
    WSU_STAMP(1);
    for (int c = 0; c < a.nch; ++c) {
        __syncthreads();
        if (c < 6) WSU_STAMP(2 + 4 * c);
        if constexpr (F1) stage_commit_first<MODE, NW>(smem, tid, c, pixidx, ldsoff, st_w);
        else if (!(a.ablate & 2) || c == 0) stage_commit<MODE, NW, PS>(smem, tid, pixidx, ldsoff, st_in, st_w);
        __syncthreads();
        if (c < 6) WSU_STAMP(3 + 4 * c);
        if (c + 1 < a.nch && !(a.ablate & 1)) stage_load<MODE, NW, F1, PS>(a, cb, c + 1, tid, pixidx, st_in, st_w);
        if (c < 6) WSU_STAMP(4 + 4 * c);
        if (a.ablate & 4) continue;                                     // no LDS fragment reads, no MFMA
        if constexpr (S16 && MODE == WSU_MODE_BF16) {
            // one v_mfma_f32_16x16x32_bf16 covers a whole 32-channel chunk of one tap: k-group kg = granule plane kg
            WSU_STATIC_FOR(9, tap, {
                constexpr int dy = tap / 3, dx = tap % 3;
                u32x4 av[MT * 2], bv[4];
                WSU_STATIC_FOR(MT * 2, m, { av[m] = *reinterpret_cast<const u32x4*>(ldsA16 + ((tap * 4) * 64 + m * 16) * 16 + kg * (64 * 16)); });
                WSU_STATIC_FOR(4, pt, { bv[pt] = *reinterpret_cast<const u32x4*>(ldsB16 + kg * PLANE_IN + (((pt >> 1) + dy) * IW + dx + (pt & 1) * 16) * 16); });
                WSU_STATIC_FOR(MT * 2, m, { WSU_STATIC_FOR(4, pt, {
                    acc16[m][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[m]), __builtin_bit_cast(bf16x8, bv[pt]), acc16[m][pt], 0, 0, 0);
                }); });
            });
        } else if constexpr (S16 && MODE == WSU_MODE_BF16X3) {
            // K = 32 per instruction = two 16-channel operand halves:
            //   type 1 (per tap):       A = [w_hi | w_lo] (planes 0..3 in k-group order), B = [x_hi | x_hi]   -> w_hi*x_hi + w_lo*x_hi
            //   type 2 (per tap pair):  A = [w_hi(t) | w_hi(t+1)],                        B = [x_lo(t) | x_lo(t+1)]
            // tap 8 has no partner: its type-2 instruction carries zeros in the upper half (1 of 27 instructions half empty).
            const int khalf = kg >> 1, kp = kg & 1;
            auto type1 = [&](auto tap_c) __attribute__((always_inline)) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int dy = tap / 3, dx = tap % 3;
                u32x4 av[MT * 2], bv[4];
                WSU_STATIC_FOR(MT * 2, m, { av[m] = *reinterpret_cast<const u32x4*>(ldsA16 + ((tap * 4) * 64 + m * 16) * 16 + kg * (64 * 16)); });
                WSU_STATIC_FOR(4, pt, { bv[pt] = *reinterpret_cast<const u32x4*>(ldsB16 + kp * PLANE_IN + (((pt >> 1) + dy) * IW + dx + (pt & 1) * 16) * 16); });
                WSU_STATIC_FOR(MT * 2, m, { WSU_STATIC_FOR(4, pt, {
                    acc16[m][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[m]), __builtin_bit_cast(bf16x8, bv[pt]), acc16[m][pt], 0, 0, 0);
                }); });
            };
            auto type2 = [&](auto tap_c) __attribute__((always_inline)) {
                constexpr int t0 = decltype(tap_c)::value;
                constexpr int t1 = t0 + 1 < 9 ? t0 + 1 : t0;
                constexpr bool single = t0 + 1 >= 9;
                const int aoff = (khalf ? t1 : t0) * 4 * 64 * 16 + kp * (64 * 16);
                const int boff = (2 + kp) * PLANE_IN + (khalf ? ((t1 / 3) * IW + t1 % 3) : ((t0 / 3) * IW + t0 % 3)) * 16;
                u32x4 av[MT * 2], bv[4];
                WSU_STATIC_FOR(MT * 2, m, {
                    av[m] = *reinterpret_cast<const u32x4*>(ldsA16 + aoff + m * 16 * 16);
                    if (single && khalf) av[m] = mk_u4(0, 0, 0, 0);
                });
                WSU_STATIC_FOR(4, pt, {
                    bv[pt] = *reinterpret_cast<const u32x4*>(ldsB16 + boff + ((pt >> 1) * IW + (pt & 1) * 16) * 16);
                    if (single && khalf) bv[pt] = mk_u4(0, 0, 0, 0);
                });
                WSU_STATIC_FOR(MT * 2, m, { WSU_STATIC_FOR(4, pt, {
                    acc16[m][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[m]), __builtin_bit_cast(bf16x8, bv[pt]), acc16[m][pt], 0, 0, 0);
                }); });
            };
            WSU_STATIC_FOR(5, tp, {
                type2(std::integral_constant<int, 2 * tp>{});
                type1(std::integral_constant<int, 2 * tp>{});
                if constexpr (2 * tp + 1 < 9) type1(std::integral_constant<int, 2 * tp + 1>{});
            });
        } else if constexpr (MODE == WSU_MODE_F16F8) {
            // per tap: f16(w) * f16(x) on v_mfma_f32_32x32x16_f16; per PAIR of taps: both cross terms of both taps in one block-scaled
            // fp8 instruction (block 0 = planes 2 of A and B, block 1 = planes 3; inside a block lanes 0-31 carry the 16 channels of the
            // pair's first tap and lanes 32-63 those of its second).  Tap 8 has no partner: its upper lanes pass zeros.
            const int sc_a = hh ? WSU_F8_SCALE_WLO : WSU_F8_SCALE_W, sc_b = hh ? WSU_F8_SCALE_X : WSU_F8_SCALE_XLO;
            auto cross = [&](auto tp_c) __attribute__((always_inline)) {
                constexpr int tp = decltype(tp_c)::value;
#if WSU_PROBE == 1                                              // timing probe: 3 instead of 5 fp8 instructions (one cross term per product)
                if constexpr (tp == 1 || tp == 3) return;
#elif WSU_PROBE == 3                                            // timing probe: no cross terms at all (plain f16)
                return;
#endif
                constexpr int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 2 * tp;
                constexpr bool single = 2 * tp + 1 >= 9;
                const int aoff = ((hh ? t1 : t0) * 4 + 2) * 64 * 16;
                const int boff = 2 * PLANE_IN + (hh ? ((t1 / 3) * IW + t1 % 3) : ((t0 / 3) * IW + t0 % 3)) * 16;
                u32x4 a0[MT], a1[MT], b0[2], b1[2];
_Pragma("unroll")
                for (int m = 0; m < MT; ++m) {
                    a0[m] = *reinterpret_cast<const u32x4*>(ldsA + aoff + m * 32 * 16);
                    a1[m] = *reinterpret_cast<const u32x4*>(ldsA + aoff + 64 * 16 + m * 32 * 16);
                }
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) {
                    b0[q] = *reinterpret_cast<const u32x4*>(ldsB + boff + q * IW * 16);
                    b1[q] = *reinterpret_cast<const u32x4*>(ldsB + boff + PLANE_IN + q * IW * 16);
                }
                if (single && hh) {
                    const u32x4 z = mk_u4(0, 0, 0, 0);
_Pragma("unroll")
                    for (int m = 0; m < MT; ++m) { a0[m] = z; a1[m] = z; }
                    b0[0] = z; b0[1] = z; b1[0] = z; b1[1] = z;
                }
_Pragma("unroll")
                for (int m = 0; m < MT; ++m)
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) wsu_mfma_f8x2(a0[m], a1[m], b0[q], b1[q], sc_a, sc_b, acc[m][q]);
            };
            auto main_term = [&](auto tap_c) __attribute__((always_inline)) {
                constexpr int tap = decltype(tap_c)::value, dy = tap / 3, dx = tap % 3;
                u32x4 ah[MT], bh[2];
_Pragma("unroll")
                for (int m = 0; m < MT; ++m) ah[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4 + hh) * 64 + m * 32) * 16);
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) bh[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE_IN + ((q + dy) * IW + dx) * 16);
_Pragma("unroll")
                for (int m = 0; m < MT; ++m)
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) wsu_mfma_f16(ah[m], bh[q], acc[m][q]);
            };
            // scheduling fences keep the operand reads of later taps behind these instructions (128-VGPR budget: without them the
            // compiler hoists the reads and spills staging registers)
            WSU_STATIC_FOR(5, tp, {
                cross(std::integral_constant<int, tp>{});
                __builtin_amdgcn_sched_barrier(0);
                main_term(std::integral_constant<int, 2 * tp>{});
                if constexpr (2 * tp + 1 < 9) main_term(std::integral_constant<int, 2 * tp + 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        } else
        WSU_STATIC_FOR(9, tap, {
            constexpr int dy = tap / 3, dx = tap % 3;
            if ((a.ablate & 16) && tap >= 6) return;                   // timing probe: 2/3 of the matrix work (results wrong)
            if constexpr (MODE == WSU_MODE_BF16X3) {
                u32x4 ahi[MT], alo[MT], bhi[2], blo[2];
_Pragma("unroll")
                for (int m = 0; m < MT; ++m) {
                    ahi[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4 + hh) * 64 + m * 32) * 16);
                    alo[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4 + 2 + hh) * 64 + m * 32) * 16);
                }
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) {
                    bhi[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE_IN + ((q + dy) * IW + dx) * 16);
                    blo[q] = *reinterpret_cast<const u32x4*>(ldsB + (2 + hh) * PLANE_IN + ((q + dy) * IW + dx) * 16);
                }
                // term-major: consecutive MFMAs go to different accumulators (same per-accumulator order, so bit-identical)
_Pragma("unroll")
                for (int m = 0; m < MT; ++m)
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(alo[m], bhi[q], acc[m][q]);     // small terms first
_Pragma("unroll")
                for (int m = 0; m < MT; ++m)
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(ahi[m], blo[q], acc[m][q]);
_Pragma("unroll")
                for (int m = 0; m < MT; ++m)
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(ahi[m], bhi[q], acc[m][q]);
            } else {
_Pragma("unroll")
                for (int ks = 0; ks < 2; ++ks) {
                    const int g = 2 * ks + hh;
                    u32x4 av[MT], bv[2];
_Pragma("unroll")
                    for (int m = 0; m < MT; ++m)
                        av[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4) * 64 + m * 32) * 16 + g * (64 * 16));
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q)
                        bv[q] = *reinterpret_cast<const u32x4*>(ldsB + g * PLANE_IN + ((q + dy) * IW + dx) * 16);
_Pragma("unroll")
                    for (int m = 0; m < MT; ++m)
_Pragma("unroll")
                        for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(av[m], bv[q], acc[m][q]);
                }
            }
        });
        if (c < 6) WSU_STAMP(5 + 4 * c);
    }

    // ---- epilogue: accumulators -> [pixel][channel] LDS tile ---------------------------------------
    __syncthreads();
    WSU_STAMP(26);
    constexpr int STRIDE = Epi<MODE>::STRIDE;
    if constexpr (S16 && MODE != WSU_MODE_F32) {
        // 16x16 C/D map: column = lane&15 -> pixel, row = 4*(lane>>4) + r -> 4 consecutive channels per lane
#pragma unroll
        for (int m = 0; m < MT * 2; ++m) {
            const int co = mbase + m * 16 + 4 * kg;
            f32x4 b4 = mk_f4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + cb * WSU_COB + co);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                float v0 = acc16[m][pt][0] + b4.x, v1 = acc16[m][pt][1] + b4.y, v2 = acc16[m][pt][2] + b4.z, v3 = acc16[m][pt][3] + b4.w;
                if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                const int px = (2 * rowpair + (pt >> 1)) * TW + (pt & 1) * 16 + l15;
                if constexpr (ESZ == 4) {
                    *reinterpret_cast<f32x4*>(smem + px * STRIDE + co * 4) = mk_f4(v0, v1, v2, v3);
                } else {
                    *reinterpret_cast<u32x2*>(smem + px * STRIDE + co * 2) = mk_u2(wsu_pack_bf16x2(v0, v1), wsu_pack_bf16x2(v2, v3));
                }
            }
        }
    } else
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = mbase + m * 32 + 8 * g4 + 4 * hh;          // 4 consecutive channels co..co+3
            f32x4 b4 = mk_f4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + cb * WSU_COB + co);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v0 = acc[m][q][4 * g4 + 0] + b4.x, v1 = acc[m][q][4 * g4 + 1] + b4.y;
                float v2 = acc[m][q][4 * g4 + 2] + b4.z, v3 = acc[m][q][4 * g4 + 3] + b4.w;
                if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                const int px = (2 * rowpair + q) * TW + l31;
                if constexpr (ESZ == 4) {
                    *reinterpret_cast<f32x4*>(smem + px * STRIDE + co * 4) = mk_f4(v0, v1, v2, v3);
                } else {
                    *reinterpret_cast<u32x2*>(smem + px * STRIDE + co * 2) = mk_u2(wsu_pack_bf16x2(v0, v1), wsu_pack_bf16x2(v2, v3));
                }
            }
        }
    }
    __syncthreads();

    tile_epilogue<MODE, NT, TH>(a, smem, tid, n, y0, x0, cb);
}


// =====================================================================================================
// Winograd F(2,3) along x (mode bf16x3, forward only; wsu_conv3x3_wino_fwd).  The run time of the direct kernel is affine in its
// MFMA count (profiles/r01/conv3x3_ablation.md, second pass), so this kernel trades MFMAs for a few VALU adds:
//   y[2t], y[2t+1] = A^T [ (G w) .* (B^T d) ],  d = x[2t-1 .. 2t+2],  w = the three column taps of one kernel row
//   B^T d = (d0-d2, d1+d2, d2-d1, d1-d3)   G w = (w0, (w0+w1+w2)/2, (w0-w1+w2)/2, w2)   A^T m = (m0+m1+m2, m1-m2-m3)
// -> 3 rows x 4 positions = 12 "taps", each a GEMM over HALF the columns: 36 instead of 54 MFMAs per wave and 16-channel chunk.
// Workgroup = 12 x 32 output pixels x 64 channels, 12 waves (3 per SIMD, 168 registers: 16 waves at 128 registers spilled), wave =
// 32 channels x (2 rows x 16 column pairs) with one accumulator tile per position (4 x 16 registers).
// LDS: transformed input  V[granule plane 4][position 4][14 rows][16 column pairs][16 B]   57 344 B  (bf16 hi / lo of B^T d)
//      transformed weights U[12 taps][4 planes][64 co][16 B]                               49 152 B  (packed by pack_wino)
// Staging: waves 0-6 own one (row, column pair, 8-channel half) each: their own two pixels (+ the tile's edge pixels) from global
// memory, the other two from the neighbour lanes (ds_bpermute), B^T d in fp32, split, 8 x 16 B to LDS; waves 7-11 copy the weight
// tile.  Loads of the next chunk are issued one per MFMA step; the epilogue applies A^T on the accumulators and then is the shared
// tile epilogue (bias, ReLU, stores, pool, head).
// Status (profiles/r01/conv3x3_ablation.md): parity-green and as fast as the direct kernel (0.96-1.02x) -- the MFMA saving (its
// matrix + epilogue part is 1.2x faster) is spent on the input transform, which one workgroup per CU cannot hide behind another
// workgroup's matrix phase.  Not the product path.
// =====================================================================================================
constexpr int WN_TAPS = 12, WN_TC = 16;
constexpr int WN_LDS_W = WN_TAPS * WSU_GRAN * WSU_COB * 16;          // 49152
constexpr int WN_TH = 12, WN_IH = WN_TH + 2, WN_NT = 768;           // 12 waves = 3 per SIMD -> 168 registers each (16 waves spilled at 128)
constexpr int WN_PLANE = 4 * WN_IH * WN_TC * 16;                     // 18432 B per granule plane
constexpr int WN_LDS_V = WSU_GRAN * WN_PLANE;                        // 73728
constexpr int WN_VTHREADS = WN_IH * WN_TC * 2;                       // 448 = waves 0..6
constexpr int WN_WTHREADS = WN_NT - WN_VTHREADS;                     // 320
constexpr int WN_WITEMS = WN_LDS_W / 16;                             // 3072
constexpr int WN_WVEC = (WN_WITEMS + WN_WTHREADS - 1) / WN_WTHREADS; // 10
constexpr int WN_ST = WN_WVEC > 8 ? WN_WVEC : 8;

struct WinoPlan { int rowbase; int xbase; int ldsoff; };             // rowbase < 0: the whole source row is zero padding

// Staging registers of a V thread (row, column pair tc, 8-channel half): st[0..1] = own pixel d1, st[2..3] = own pixel d2,
// st[4..5] = the halo pixel that no neighbour lane owns (d0 for tc == 0, d3 for tc == 15); d0 / d3 of the inner column pairs come
// from the neighbour lanes at commit time (ds_bpermute), so every input pixel is fetched from global memory once per workgroup.
template <int K>
__device__ __forceinline__ void wino_load_slot(const ConvArgs& a, int cb, int c, int tid, const WinoPlan& pl, u32x4 (&st)[WN_ST]) {
    if (tid < WN_VTHREADS) {
        if constexpr (K < 6) {
            const char* src; int csrc, ch0;
            if (c < a.nch1) { src = a.x1; csrc = a.c1; ch0 = c * 16; }
            else            { src = a.x2; csrc = a.c2; ch0 = (c - a.nch1) * 16; }
            constexpr int half = K & 1;
            const int tc = (tid >> 1) & 15;
            const int i = K < 2 ? 1 : (K < 4 ? 2 : (tc == 0 ? 0 : 3));
            int xx = pl.xbase + i;
            bool ok = pl.rowbase >= 0 && (K < 4 || tc == 0 || tc == 15);
            if (a.pad_zero) ok = ok && xx >= 0 && xx < a.w; else xx = wsu_reflect(xx, a.w);
            u32x4 v = mk_u4(0, 0, 0, 0);
            if (ok) v = reinterpret_cast<const u32x4*>(src + ((size_t)(pl.rowbase + xx) * csrc + ch0) * 4 + (tid & 1) * 32)[half];
            st[K] = v;
        }
    } else {
        if constexpr (K < WN_WVEC) {
            const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp + ((size_t)cb * a.nch + c) * WN_LDS_W);
            const int t = tid - WN_VTHREADS;
            if (t + K * WN_WTHREADS < WN_WITEMS) st[K] = wsrc[t + K * WN_WTHREADS];
        }
    }
}

__device__ __forceinline__ void wino_load(const ConvArgs& a, int cb, int c, int tid, const WinoPlan& pl, u32x4 (&st)[WN_ST]) {
    WSU_STATIC_FOR(WN_ST, k, { wino_load_slot<k>(a, cb, c, tid, pl, st); });
}

__device__ __forceinline__ void wino_commit(char* smem, int tid, const WinoPlan& pl, const u32x4 (&st)[WN_ST]) {
    if (tid < WN_VTHREADS) {
        f32x4 d[4][2];
        d[1][0] = __builtin_bit_cast(f32x4, st[0]); d[1][1] = __builtin_bit_cast(f32x4, st[1]);
        d[2][0] = __builtin_bit_cast(f32x4, st[2]); d[2][1] = __builtin_bit_cast(f32x4, st[3]);
        {
            // d0 = pixel 2tc-1 = the d2 of column pair tc-1 (two lanes down), d3 = pixel 2tc+2 = the d1 of column pair tc+1
            const int lane = tid & 63, tc = (tid >> 1) & 15;
            const int from_lo = ((lane - 2) & 63) * 4, from_hi = ((lane + 2) & 63) * 4;
            // (inline asm: hipcc folded the per-element __builtin_amdgcn_ds_bpermute calls of one vector into a single exchange)
            auto exch = [](int addr, const f32x4& v) __attribute__((always_inline)) {
                f32x4 r;
                asm volatile("ds_bpermute_b32 %0, %4, %5\n\tds_bpermute_b32 %1, %4, %6\n\tds_bpermute_b32 %2, %4, %7\n\tds_bpermute_b32 %3, %4, %8\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w) : "v"(addr), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                return r;
            };
            WSU_STATIC_FOR(2, k, {                          // compile-time k: `st[4 + k]` in a plain unrolled loop pins st[] in scratch
                const f32x4 halo = __builtin_bit_cast(f32x4, st[4 + k]);
                const f32x4 n0 = exch(from_lo, d[2][k]);
                d[0][k] = tc == 0 ? halo : n0;
                const f32x4 n3 = exch(from_hi, d[1][k]);
                d[3][k] = tc == 15 ? halo : n3;
            });
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f32x4 v0, v1;
            if (p == 0)      { v0 = d[0][0] - d[2][0]; v1 = d[0][1] - d[2][1]; }
            else if (p == 1) { v0 = d[1][0] + d[2][0]; v1 = d[1][1] + d[2][1]; }
            else if (p == 2) { v0 = d[2][0] - d[1][0]; v1 = d[2][1] - d[1][1]; }
            else             { v0 = d[1][0] - d[3][0]; v1 = d[1][1] - d[3][1]; }
            u32x4 hi, lo;
            wsu_split8(v0, v1, hi, lo);
            char* dst = smem + pl.ldsoff + p * (WN_IH * WN_TC * 16);
            *reinterpret_cast<u32x4*>(dst) = hi;
            *reinterpret_cast<u32x4*>(dst + 2 * WN_PLANE) = lo;
        }
    } else {
        u32x4* wdst = reinterpret_cast<u32x4*>(smem + WN_LDS_V);
        const int t = tid - WN_VTHREADS;
        WSU_STATIC_FOR(WN_WVEC, k, { if (t + k * WN_WTHREADS < WN_WITEMS) wdst[t + k * WN_WTHREADS] = st[k]; });
    }
}

__global__ __launch_bounds__(WN_NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv3x3_wino_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MODE = WSU_MODE_BF16X3;
    const int tid = threadIdx.x;
    const unsigned lid = wsu_xcd_remap(blockIdx.x, gridDim.x);
    const int cb = lid % a.ncb;
    int tile = lid / a.ncb;
    const int tx = tile % a.tiles_x; tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int n = tile / a.tiles_y;
    const int y0 = ty * WN_TH, x0 = tx * TW;

    WinoPlan pl;
    {
        const int h = tid & 1, tc = (tid >> 1) & 15, row = tid >> 5;            // staging threads: tid < 448 -> row 0..13
        int yy = y0 - 1 + row;
        bool ok = true;
        if (a.pad_zero) ok = yy >= 0 && yy < a.h; else yy = wsu_reflect(yy, a.h);
        pl.rowbase = ok ? (n * a.h + yy) * a.w : -1;
        pl.xbase = x0 - 1 + 2 * tc;
        pl.ldsoff = h * WN_PLANE + (row * WN_TC + tc) * 16;                      // + position * (18 * 16 * 16); lo planes at + 2 * WN_PLANE
    }

    const int wv = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int rowpair = wv >> 1, mbase = (wv & 1) * 32;
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    const char* ldsA = smem + WN_LDS_V + (mbase + l31) * 16;                     // + ((tap*4 + g)*64)*16
    const char* ldsB = smem + ((2 * rowpair) * WN_TC + l31) * 16;                // + g*WN_PLANE + (p*18 + dy)*16*16   (l31 = row-in-pair*16 + tc)

    u32x4 st[WN_ST];
    wino_load(a, cb, 0, tid, pl, st);
    for (int c = 0; c < a.nch; ++c) {
        __syncthreads();
        if (!(a.ablate & 2) || c == 0) wino_commit(smem, tid, pl, st);
        __syncthreads();
        const bool prefetch = c + 1 < a.nch && !(a.ablate & 1);
        if (a.ablate & 4) { if (prefetch) wino_load(a, cb, c + 1, tid, pl, st); continue; }
        WSU_STATIC_FOR(3, dy, {
            WSU_STATIC_FOR(4, p, {
                constexpr int tap = dy * 4 + p;
                // next chunk's loads, one slot per MFMA step; only vector-memory instructions are pinned between the fences
                __builtin_amdgcn_sched_barrier(0x38F);
                if (prefetch) wino_load_slot<tap>(a, cb, c + 1, tid, pl, st);
                __builtin_amdgcn_sched_barrier(0x38F);
                const u32x4 ahi = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4 + hh) * 64) * 16);
                const u32x4 alo = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4 + 2 + hh) * 64) * 16);
                const u32x4 bhi = *reinterpret_cast<const u32x4*>(ldsB + hh * WN_PLANE + (p * WN_IH + dy) * (WN_TC * 16));
                const u32x4 blo = *reinterpret_cast<const u32x4*>(ldsB + (2 + hh) * WN_PLANE + (p * WN_IH + dy) * (WN_TC * 16));
                wsu_mfma_step<MODE>(alo, bhi, acc[p]);
                wsu_mfma_step<MODE>(ahi, blo, acc[p]);
                wsu_mfma_step<MODE>(ahi, bhi, acc[p]);
            });
        });
    }

    // ---- output transform A^T on the accumulators -> [pixel][channel] tile ------------------------------
    __syncthreads();
    constexpr int STRIDE = Epi<MODE>::STRIDE;
    const int prow = 2 * rowpair + (l31 >> 4), pcol = 2 * (l31 & 15);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int co = mbase + 8 * g4 + 4 * hh;
        f32x4 b4 = mk_f4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + cb * WSU_COB + co);
        f32x4 e, o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float m0 = acc[0][4 * g4 + k], m1 = acc[1][4 * g4 + k], m2 = acc[2][4 * g4 + k], m3 = acc[3][4 * g4 + k];
            float ye = (m0 + m1) + m2 + b4[k], yo = (m1 - m2) - m3 + b4[k];
            if (a.relu) { ye = fmaxf(ye, 0.f); yo = fmaxf(yo, 0.f); }
            e[k] = ye; o[k] = yo;
        }
        *reinterpret_cast<f32x4*>(smem + (prow * TW + pcol) * STRIDE + co * 4) = e;
        *reinterpret_cast<f32x4*>(smem + (prow * TW + pcol + 1) * STRIDE + co * 4) = o;
    }
    __syncthreads();
    tile_epilogue<MODE, WN_NT, WN_TH>(a, smem, tid, n, y0, x0, cb);
}

// OIHW fp32 -> [cob][chunk of 16 ci][tap = row*4 + position][plane: hi ch 0-7, hi ch 8-15, lo 0-7, lo 8-15][co 64][8 x bf16]
__global__ void pack_wino_kernel(const float* __restrict__ w, uint16_t* __restrict__ dst, int cin, int cout) {
    const int nch = cin / 16;
    const long long total = (long long)(cout / WSU_COB) * nch * WN_TAPS * WSU_GRAN * WSU_COB * 8;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int e = t % 8; t /= 8;
        const int co = t % WSU_COB; t /= WSU_COB;
        const int g = t % WSU_GRAN; t /= WSU_GRAN;
        const int tap = t % WN_TAPS; t /= WN_TAPS;
        const int c = t % nch; t /= nch;
        const int cb = (int)t;
        const int ci = c * 16 + 8 * (g & 1) + e, part = g >> 1;
        const int u = tap >> 2, pos = tap & 3;
        const float* wr = w + (((size_t)(cb * WSU_COB + co) * cin + ci) * 3 + u) * 3;
        const float w0 = wr[0], w1 = wr[1], w2 = wr[2];
        float val;
        if (pos == 0) val = w0;
        else if (pos == 1) val = 0.5f * ((w0 + w2) + w1);
        else if (pos == 2) val = 0.5f * ((w0 + w2) - w1);
        else val = w2;
        const float x = part ? wsu_bf16_lo_residual(val) : val;
        dst[d] = __builtin_bit_cast(uint16_t, (__bf16)x);
    }
}

// =====================================================================================================
// Ping-pong kernel (experimental, WSU_CONV_IMPL=pp).  Why it was written: with two independent workgroups per CU the v1 kernel's matrix phase
// (MFMA over a staged chunk) and memory phase (global -> LDS staging, epilogue stores) of co-resident
// workgroups run in lock-step, so its time is the SUM of the two (measured by ablation, DESIGN.md section 5:
// e12 bf16x3 1966 us = 950 us MFMA-only + 1016 us memory-only).  Here ONE persistent 512-thread workgroup per
// CU holds two 4-wave groups that are in anti-phase BY CONSTRUCTION: in every barrier interval one group runs
// the MFMAs of its chunk while the other group (its waves sit on the same four SIMDs) commits the next chunk
// to LDS, issues the prefetch after that and stores a finished tile -- matrix beside memory on every SIMD.
//   * each group owns an 8x32-pixel tile (64 co) -> per wave 64 co x 64 px as in v1;
//   * the two groups work on tile pairs with the SAME output-channel block, so the weight tile is staged
//     once (by group 0, double-buffered) and read by both: weight traffic (62 % of v1's staged bytes) halves;
//   * the epilogue goes straight from the accumulators to NHWC (4 consecutive channels = 16 B per lane), the
//     2x2 max-pool is taken in registers (rows = the wave's two MFMA column tiles, columns = lane pairs);
//     no LDS round trip, no extra barriers;
//   * work is walked persistently with an XCD-contiguous order, chunk 0 of the next tile is prefetched during
//     the last chunk of the current one.
// Interval i:  group g is in its memory phase when (i - g) is even, in its matrix phase when odd.
// =====================================================================================================
constexpr int PP_LDS = 2 * LDS_IN + 2 * LDS_W;           // 118016 B

struct PPItem { int n, y0, x0, cb, valid; };

__device__ __forceinline__ PPItem pp_item(const ConvArgs& a, int pair, int grp) {
    PPItem it;
    const int t2 = pair / a.ncb;
    it.cb = pair - t2 * a.ncb;
    int tile = 2 * t2 + grp;
    const int ntiles = a.n * a.tiles_x * a.tiles_y;
    it.valid = tile < ntiles;
    tile = min(tile, ntiles - 1);
    const int tx = tile % a.tiles_x; tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    it.n = tile / a.tiles_y; it.y0 = ty * TH; it.x0 = tx * TW;
    return it;
}

template <int MODE>
__device__ __forceinline__ void pp_plan(const ConvArgs& a, const PPItem& it, int gt, int (&pixidx)[Shape<4>::IN_VEC]) {
    constexpr int NT = 256;
    constexpr int NITEMS = (MODE == WSU_MODE_BF16X3) ? NPIX_IN * 2 : NPIX_IN * 4;
    constexpr int NLOOP = (MODE == WSU_MODE_BF16X3) ? Shape<4>::IN_VEC3 : Shape<4>::IN_VEC;
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        const int i = gt + j * NT;
        const int pix = (MODE == WSU_MODE_BF16X3) ? (i >> 1) : (i >> 2);
        const int r = pix / IW, c = pix - r * IW;
        int yy = it.y0 - 1 + r, xx = it.x0 - 1 + c;
        int p;
        if (a.pad_zero) {
            p = (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) ? (it.n * a.h + yy) * a.w + xx : -1;
        } else {
            yy = wsu_reflect(yy, a.h); xx = wsu_reflect(xx, a.w);
            p = (it.n * a.h + yy) * a.w + xx;
        }
        if (!it.valid) p = -1;
        pixidx[j] = (i < NITEMS) ? p : -2;
    }
}

// global -> registers: this group's input-tile items of chunk c; group 0 also takes the shared weight slice
template <int MODE>
__device__ __forceinline__ void pp_load(const ConvArgs& a, int c, int gt, const int (&pixidx)[Shape<4>::IN_VEC],
                                        u32x4 (&st_in)[Shape<4>::ST_IN]) {
    constexpr int ESZ = Epi<MODE>::ESZ;
    constexpr int CK = (MODE == WSU_MODE_BF16) ? 32 : 16;
    constexpr int NT = 256;
    constexpr int NLOOP = (MODE == WSU_MODE_BF16X3) ? Shape<4>::IN_VEC3 : Shape<4>::IN_VEC;
    const char* src; int csrc, ch0;
    if (c < a.nch1) { src = a.x1; csrc = a.c1; ch0 = c * CK; }
    else            { src = a.x2; csrc = a.c2; ch0 = (c - a.nch1) * CK; }
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        const int p = pixidx[j];
        if constexpr (MODE == WSU_MODE_BF16X3) {
            const int sub = (gt + j * NT) & 1;
            u32x4 v0 = mk_u4(0, 0, 0, 0), v1 = v0;
            if (p >= 0) {
                const u32x4* g = reinterpret_cast<const u32x4*>(src + ((size_t)p * csrc + ch0) * 4 + sub * 32);
                v0 = g[0]; v1 = g[1];
            }
            st_in[2 * j] = v0; st_in[2 * j + 1] = v1;
        } else {
            const int sub = (gt + j * NT) & 3;
            u32x4 v = mk_u4(0, 0, 0, 0);
            if (p >= 0) v = *reinterpret_cast<const u32x4*>(src + ((size_t)p * csrc + ch0) * ESZ + sub * 16);
            st_in[j] = v;
        }
    }
}

// The shared weight slice of (cb, chunk c) goes global -> LDS by LDS-DMA (global_load_lds_dwordx4): a linear 36 KB
// copy, 9 wave-instructions per wave of group 0, no staging registers and no ds_write.  LDS destination = wave-uniform
// base (M0) + lane * 16, global source per lane.
typedef __attribute__((address_space(3))) void wsu_lds_void;
typedef __attribute__((address_space(1))) const void wsu_glb_void;
__device__ __forceinline__ void pp_dma_weights(const ConvArgs& a, int cb, int c, char* w_lds, int gt, int wv4) {
    const char* wsrc = a.wp + ((size_t)cb * a.nch + c) * LDS_W + (size_t)gt * 16;
    char* ldst = w_lds + wv4 * 64 * 16;
#pragma unroll
    for (int k = 0; k < W_ITEMS / 256; ++k)
        __builtin_amdgcn_global_load_lds((wsu_glb_void*)(wsrc + (size_t)k * 256 * 16), (wsu_lds_void*)(ldst + k * 256 * 16), 16, 0, 0);
}

template <int MODE>
__device__ __forceinline__ void pp_commit(char* in_lds, int gt, const int (&pixidx)[Shape<4>::IN_VEC],
                                          const u32x4 (&st_in)[Shape<4>::ST_IN]) {
    constexpr int NT = 256;
    constexpr int NLOOP = (MODE == WSU_MODE_BF16X3) ? Shape<4>::IN_VEC3 : Shape<4>::IN_VEC;
#pragma unroll
    for (int j = 0; j < NLOOP; ++j) {
        if (pixidx[j] != -2) {
            const int i = gt + j * NT;
            if constexpr (MODE == WSU_MODE_BF16X3) {
                const int off = (i & 1) * PLANE_IN + (i >> 1) * 16;
                u32x4 hi, lo;
                wsu_split8(__builtin_bit_cast(f32x4, st_in[2 * j]), __builtin_bit_cast(f32x4, st_in[2 * j + 1]), hi, lo);
                *reinterpret_cast<u32x4*>(in_lds + off) = hi;
                *reinterpret_cast<u32x4*>(in_lds + off + 2 * PLANE_IN) = lo;
            } else {
                *reinterpret_cast<u32x4*>(in_lds + (i & 3) * PLANE_IN + (i >> 2) * 16) = st_in[j];
            }
        }
    }
}

// Fragment set of one tap: 8 x 16 B per lane.
//   BF16X3: {ahi[0], ahi[1], alo[0], alo[1], bhi[0], bhi[1], blo[0], blo[1]}
//   others: {a[ks0][0], a[ks0][1], b[ks0][0], b[ks0][1], a[ks1][0], a[ks1][1], b[ks1][0], b[ks1][1]}
template <int MODE, int TAP>
__device__ __forceinline__ void pp_load_frags(u32x4 (&f)[8], const char* ldsA, const char* ldsB, int hh) {
    constexpr int dy = TAP / 3, dx = TAP % 3;
    if constexpr (MODE == WSU_MODE_BF16X3) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f[m] = *reinterpret_cast<const u32x4*>(ldsA + ((TAP * 4 + hh) * 64 + m * 32) * 16);
            f[2 + m] = *reinterpret_cast<const u32x4*>(ldsA + ((TAP * 4 + 2 + hh) * 64 + m * 32) * 16);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f[4 + q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE_IN + ((q + dy) * IW + dx) * 16);
            f[6 + q] = *reinterpret_cast<const u32x4*>(ldsB + (2 + hh) * PLANE_IN + ((q + dy) * IW + dx) * 16);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int g = 2 * ks + hh;
#pragma unroll
            for (int m = 0; m < 2; ++m) f[4 * ks + m] = *reinterpret_cast<const u32x4*>(ldsA + ((TAP * 4 + g) * 64 + m * 32) * 16);
#pragma unroll
            for (int q = 0; q < 2; ++q) f[4 * ks + 2 + q] = *reinterpret_cast<const u32x4*>(ldsB + g * PLANE_IN + ((q + dy) * IW + dx) * 16);
        }
    }
}

template <int MODE>
__device__ __forceinline__ void pp_mfma_frags(const u32x4 (&f)[8], f32x16 (&acc)[2][2]) {
    if constexpr (MODE == WSU_MODE_BF16X3) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(f[2 + m], f[4 + q], acc[m][q]);      // lo * hi   (small terms first)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(f[m], f[6 + q], acc[m][q]);          // hi * lo
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(f[m], f[4 + q], acc[m][q]);          // hi * hi
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) wsu_mfma_step<MODE>(f[4 * ks + m], f[4 * ks + 2 + q], acc[m][q]);
    }
}

// One chunk of the matrix phase.  Only ONE wave per SIMD issues MFMAs during an interval, so nothing else hides its
// LDS latency: the fragments are software-pipelined by a whole tap (two register sets) and the next tap's eight
// ds_read_b128 are interleaved one by one under the current tap's MFMAs (sched_group_barrier pins the interleave).
template <int MODE>
__device__ __forceinline__ void pp_compute(const char* in_lds, const char* w_lds, int wv4, int l31, int hh, f32x16 (&acc)[2][2]) {
    const char* ldsA = w_lds + l31 * 16;
    const char* ldsB = in_lds + ((2 * wv4) * IW + l31) * 16;
    constexpr int NMFMA = MODE == WSU_MODE_BF16X3 ? 12 : (MODE == WSU_MODE_F32 ? 32 : 8);
    u32x4 fa[8], fb[8];
    pp_load_frags<MODE, 0>(fa, ldsA, ldsB, hh);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                              // prologue reads form their own group
    WSU_STATIC_FOR(9, tap,
        if constexpr (tap + 1 < 9) {
            if constexpr (tap % 2 == 0) pp_load_frags<MODE, (tap + 1 < 9 ? tap + 1 : 0)>(fb, ldsA, ldsB, hh);
            else pp_load_frags<MODE, (tap + 1 < 9 ? tap + 1 : 0)>(fa, ldsA, ldsB, hh);
        }
        if constexpr (tap % 2 == 0) pp_mfma_frags<MODE>(fa, acc); else pp_mfma_frags<MODE>(fb, acc);
        if constexpr (tap + 1 < 9) {
            _Pragma("unroll") for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA / 8, 0);          // MFMA
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - 8 * (NMFMA / 8), 0);
        }
    );
}

// accumulators -> NHWC (+bias, ReLU, ReLU mask of the data-gradient pass), fused 2x2 max-pool in registers
template <int MODE>
__device__ __forceinline__ void pp_epilogue(const ConvArgs& a, const PPItem& it, int wv4, int l31, int hh, const f32x16 (&acc)[2][2]) {
    constexpr int ESZ = Epi<MODE>::ESZ;
    const int cglob = it.cb * WSU_COB;
    char* ydst = a.y; int ych = a.csplit, ycoff = cglob;
    const char* msk = a.relu_mask;
    if (cglob >= a.csplit) { ydst = a.y2; ych = a.cout - a.csplit; ycoff = cglob - a.csplit; msk = a.relu_mask2; }
    const int col = it.x0 + l31;
    const int hp = a.h >> 1, wp2 = a.w >> 1;
    const int gy = (it.y0 >> 1) + wv4, gx = (it.x0 >> 1) + (l31 >> 1);
    const bool pool_lane = a.ypool && !(l31 & 1) && gy < hp && gx < wp2;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = m * 32 + 8 * g4 + 4 * hh;
            f32x4 b4 = mk_f4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + cglob + co);
            f32x4 v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                v[q] = mk_f4(acc[m][q][4 * g4 + 0] + b4.x, acc[m][q][4 * g4 + 1] + b4.y,
                             acc[m][q][4 * g4 + 2] + b4.z, acc[m][q][4 * g4 + 3] + b4.w);
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] = fmaxf(v[q][e], 0.f);
                }
                if constexpr (ESZ == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] = (float)(__bf16)v[q][e];   // compare / pool on the stored values
                }
                const int row = it.y0 + 2 * wv4 + q;
                if (row < a.h && col < a.w) {
                    const size_t eoff = ((size_t)(it.n * a.h + row) * a.w + col) * ych + ycoff + co;
                    if constexpr (ESZ == 4) {
                        f32x4 val = v[q];
                        if (msk) {
                            const f32x4 mk = *reinterpret_cast<const f32x4*>(msk + eoff * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (!(mk[e] > 0.f)) val[e] = 0.f;
                        }
                        *reinterpret_cast<f32x4*>(ydst + eoff * 4) = val;
                    } else {
                        *reinterpret_cast<u32x2*>(ydst + eoff * 2) = mk_u2(wsu_pack_bf16x2(v[q][0], v[q][1]), wsu_pack_bf16x2(v[q][2], v[q][3]));
                    }
                }
            }
            if (a.ypool) {                                          // wave-uniform: every lane takes part in the shuffles
                f32x4 best; uint32_t idx = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n0 = __shfl_xor(v[0][e], 1, 64), n1 = __shfl_xor(v[1][e], 1, 64);
                    float bv = v[0][e]; uint32_t bi = 0;              // window order (0,0) (0,1) (1,0) (1,1), first max wins
                    if (n0 > bv || n0 != n0) { bv = n0; bi = 1; }
                    if (v[1][e] > bv || v[1][e] != v[1][e]) { bv = v[1][e]; bi = 2; }
                    if (n1 > bv || n1 != n1) { bv = n1; bi = 3; }
                    best[e] = bv; idx |= bi << (8 * e);
                }
                if (pool_lane) {
                    const size_t eoff = ((size_t)(it.n * hp + gy) * wp2 + gx) * a.cout + cglob + co;
                    if constexpr (ESZ == 4) *reinterpret_cast<f32x4*>(a.ypool + eoff * 4) = best;
                    else *reinterpret_cast<u32x2*>(a.ypool + eoff * 2) = mk_u2(wsu_pack_bf16x2(best[0], best[1]), wsu_pack_bf16x2(best[2], best[3]));
                    if (a.pidx) *reinterpret_cast<uint32_t*>(a.pidx + eoff) = idx;
                }
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);          // wave-uniform: waves 0-3 / 4-7
    const int gt = tid & 255, wv4 = (tid >> 6) & 3, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const bool with_w = grp == 0;
    char* in_lds = smem + grp * LDS_IN;
    char* w_base = smem + 2 * LDS_IN;

    const int G = gridDim.x;
    const int lw = (int)wsu_xcd_remap(blockIdx.x, G);
    const int ntiles = a.n * a.tiles_x * a.tiles_y;
    const int npairs = ((ntiles + 1) >> 1) * a.ncb;
    const int K = npairs > lw ? (npairs - lw + G - 1) / G : 0;         // tile pairs walked by this workgroup
    const int J = K * a.nch;                                            // chunk steps per group

    int pixidx[Shape<4>::IN_VEC];
    u32x4 st_in[Shape<4>::ST_IN];
    f32x16 acc[2][2];
    PPItem cur = pp_item(a, lw, grp);                                   // item being multiplied / stored
    if (J > 0) {
        if (with_w) pp_dma_weights(a, cur.cb, 0, w_base, gt, wv4);      // W(0) -> Wbuf[0]
        pp_plan<MODE>(a, cur, gt, pixidx);
        pp_load<MODE>(a, 0, gt, pixidx, st_in);
    }
    const int lag = (a.ablate & 256) ? 0 : grp;          // experiment: 256 = both groups in phase
    for (int i = 0; i <= 2 * J + 1; ++i) {
        const int ph = i - lag;
        if (ph >= 0) {
            const int j = ph >> 1;
            if (!(ph & 1)) {
                // ---- memory phase: commit chunk j, store the tile finished in the previous interval, prefetch chunk j+1.
                // Everything this wave issued earlier (input prefetch two intervals ago, weight DMA one interval ago, old
                // stores) has had at least a full matrix phase to land: draining here is ~free and makes the DMA'd
                // weights visible at the barrier that ends this interval.  The prefetch issued BELOW stays in flight.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int c = j % a.nch;
                if (j < J) pp_commit<MODE>(in_lds, gt, pixidx, st_in);
                if (j > 0 && c == 0 && cur.valid && !(a.ablate & 8)) pp_epilogue<MODE>(a, cur, wv4, l31, hh, acc);
                if (j + 1 < J && !(a.ablate & 1)) {
                    const int cn = c + 1 == a.nch ? 0 : c + 1;
                    if (cn == 0) {
                        const PPItem nxt = pp_item(a, lw + ((j + 1) / a.nch) * G, grp);
                        pp_plan<MODE>(a, nxt, gt, pixidx);
                    }
                    pp_load<MODE>(a, cn, gt, pixidx, st_in);
                }
            } else if (j < J) {
                // ---- matrix phase
                const int c = j % a.nch;
                if (with_w && j + 1 < J && !(a.ablate & 1)) {             // W(j+1) -> the buffer nobody reads any more
                    const int pair = lw + ((j + 1) / a.nch) * G;
                    pp_dma_weights(a, pair % a.ncb, (j + 1) % a.nch, w_base + ((j + 1) & 1) * LDS_W, gt, wv4);
                }
                if (c == 0) {
                    cur = pp_item(a, lw + (j / a.nch) * G, grp);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
                }
                // (s_setprio on either phase was measured to make no difference: profiles/r01/conv3x3_ablation.md)
                if (cur.valid && !(a.ablate & 4)) pp_compute<MODE>(in_lds, w_base + (j & 1) * LDS_W, wv4, l31, hh, acc);
            }
        }
        // Interval barrier.  NOT __syncthreads(): its release fence is `s_waitcnt vmcnt(0)`, which would make every
        // interval wait for the prefetch it has just issued (and for the epilogue stores).  Only LDS traffic has to be
        // ordered here: drain the LDS queue, then a bare s_barrier; global loads are waited for where their registers
        // are consumed (the commit two intervals later), stores never.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

template <int MODE>
int launch_conv_pp(const ConvArgs& a, hipStream_t s) {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            wsu_set_error("conv3x3: cannot query the device"); return WSU_ERR_HIP;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp_kernel<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3_pp): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        ncu = prop.multiProcessorCount;
    }
    const long long ntiles = (long long)a.n * a.tiles_x * a.tiles_y;
    const long long npairs = ((ntiles + 1) / 2) * a.ncb;
    if (npairs <= 0 || ntiles > 0x3FFFFFFFLL) { wsu_set_error("conv3x3: %lld tiles out of range", ntiles); return WSU_ERR_ARG;}
    const int grid = (int)(npairs < ncu ? npairs : ncu);
    hipLaunchKernelGGL(conv3x3_pp_kernel<MODE>, dim3(grid), dim3(512), PP_LDS, s, a);
    return wsu_check_launch("conv3x3_pp_kernel");
}

template <int MODE, int NW, bool S16 = false, bool F1 = false, bool PS = false>
int launch_conv_nw(const ConvArgs& a_in, hipStream_t s) {
    constexpr int EPI_BYTES = Shape<NW>::TH * TW * Epi<MODE>::STRIDE;
    constexpr int MAIN_BYTES = Shape<NW>::LDS_MAIN + (F1 ? F1_BYTES(Shape<NW>::NPIX_IN) : 0);
    const int lds = EPI_BYTES > MAIN_BYTES ? EPI_BYTES : MAIN_BYTES;
    ConvArgs a = a_in;
    a.tiles_y = (a.h + Shape<NW>::TH - 1) / Shape<NW>::TH;
    static bool attr_done = false;     // benign race: idempotent
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<MODE, NW, S16, F1, PS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        attr_done = true;
    }
    const long long nblk = (long long)a.n * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk <= 0 || nblk > 0x7FFFFFFFLL) { wsu_set_error("conv3x3: grid of %lld workgroups out of range", nblk); return WSU_ERR_ARG; }
    hipLaunchKernelGGL((conv3x3_kernel<MODE, NW, S16, F1, PS>), dim3((unsigned)nblk), dim3(NW * 64), lds, s, a);
    return wsu_check_launch("conv3x3_kernel");
}

// WSU_CONV_WAVES=4|8 overrides the workgroup shape (tuning / A-B runs); default chosen per mode by measurement.
template <int MODE>
int launch_conv(const ConvArgs& a, hipStream_t s, bool in_split = false) {
    if constexpr (MODE == WSU_MODE_BF16X3) {
        if (in_split) return launch_conv_nw<MODE, 8, false, false, true>(a, s);      // pre-split input: the measured default shape only
    }
    if constexpr (MODE == WSU_MODE_F16F8) {
        if (a.img) return launch_conv_nw<MODE, 8, false, true, false>(a, s);
        if (!in_split) return launch_conv_nw<MODE, 8, false, false, false>(a, s);      // API mode F16F8X: fp32 storage on both sides
        static int nw = -1;                 // WSU_CONV_WAVES=4: 64 co x 64 px per wave (fewer LDS fragment reads per MFMA); 16: 16x32-pixel tile
        if (nw < 0) { const char* e = getenv("WSU_CONV_WAVES"); nw = e ? atoi(e) : 8; }
        if (nw == 16) return launch_conv_nw<MODE, 16, false, false, true>(a, s);
        return nw == 4 ? launch_conv_nw<MODE, 4, false, false, true>(a, s) : launch_conv_nw<MODE, 8, false, false, true>(a, s);
    } else {
    // Default = the per-tile kernel (v1): measured faster (bench conv3x3 15.8 ms vs 17.9 ms per batch-32 forward in bf16x3).
    // WSU_CONV_IMPL=pp selects the ping-pong kernel (kept for the next tuning round; profiles/r01/conv3x3_ablation.md).
    if (a.img) {                                                // fused first layer: the measured default shape of each mode
        if constexpr (MODE == WSU_MODE_F32) return launch_conv_nw<MODE, 4, false, true>(a, s);
        else if constexpr (MODE == WSU_MODE_BF16) return launch_conv_nw<MODE, 8, true, true>(a, s);
        else return launch_conv_nw<MODE, 8, false, true>(a, s);
    }
    static int impl = -1;
    if (impl < 0) { const char* e = getenv("WSU_CONV_IMPL"); impl = (e && e[0] == 'p' && e[1] == 'p') ? 0 : 1; }
    const bool mask_bf16 = MODE == WSU_MODE_BF16 && (a.relu_mask || a.relu_mask2);
    if (impl == 0 && !mask_bf16 && !a.head_w) return launch_conv_pp<MODE>(a, s);
    static int nw = 0;
    if (nw == 0) {
        const char* e = getenv("WSU_CONV_WAVES");
        nw = (e && atoi(e) == 4) ? 4 : ((e && atoi(e) == 8) ? 8 : ((e && atoi(e) == 16) ? 16 : (MODE == WSU_MODE_F32 ? 4 : 8)));
    }
    if (nw == 16) return launch_conv_nw<MODE, 16>(a, s);
    if constexpr (MODE != WSU_MODE_F32) {
        // MFMA tile shape: v_mfma_f32_16x16x32_bf16 is the default for bf16 storage (measured +3-5 % over 32x32x16 at 8 waves,
        // a higher sustained clock at equal cycles); for bf16x3 the two shapes time the same and 32x32x16 stays.
        // WSU_CONV_MFMA=16|32 overrides.
        static int s16 = -1;
        if (s16 < 0) { const char* e = getenv("WSU_CONV_MFMA"); s16 = e ? (atoi(e) == 16) : (MODE == WSU_MODE_BF16 ? 1 : 0); }
        if (s16) return nw == 4 ? launch_conv_nw<MODE, 4, true>(a, s) : launch_conv_nw<MODE, 8, true>(a, s);
    }
    return nw == 4 ? launch_conv_nw<MODE, 4>(a, s) : launch_conv_nw<MODE, 8>(a, s);
    }
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
        if (transpose_flip == 2) val = w[(((size_t)ci * cin + m) * 3 + (2 - v)) * 3 + (2 - u)];   // as below with the two tap axes swapped
        else if (transpose_flip) val = w[(((size_t)ci * cin + m) * 3 + (2 - u)) * 3 + (2 - v)];   // W[co=ci_d][ci=m]
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

// F16F8: one thread per (cob, chunk, tap, co) row of 16 channels -> [f16 0-7][f16 8-15][e4m3(w * 2^6)][e4m3((w - f16 w) * 2^18)]
// transpose_flip as in pack_conv3x3_kernel (data-gradient weights: the roles of Cin / Cout swap, taps flipped; 2 = tap axes swapped too).
__global__ void pack_conv3x3_f16f8_kernel(const float* __restrict__ w, char* __restrict__ dst, int cin, int cout, int transpose_flip) {
    const int kin = transpose_flip ? cout : cin, mout = transpose_flip ? cin : cout;
    const int nch = kin / 16;
    const long long total = (long long)(mout / WSU_COB) * nch * 9 * WSU_COB;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int co = t % WSU_COB; t /= WSU_COB;
        const int tap = t % 9; t /= 9;
        const int c = t % nch; t /= nch;
        const int cb = (int)t;
        f32x4 q[4];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = c * 16 + e, m = cb * WSU_COB + co, u = tap / 3, v = tap % 3;
            float val;
            if (transpose_flip >= 3) {
                // border-ring sets of the reflect adjoint (train_pl.hip): a 1x3 filter in the middle tap row, applied along a strip of the
                // gradient's first / last row (3: top, 4: bottom) or first / last column (5: left, 6: right)
                const int ky = transpose_flip == 3 ? 0 : transpose_flip == 4 ? 2 : 2 - v;
                const int kx = transpose_flip == 5 ? 0 : transpose_flip == 6 ? 2 : 2 - v;
                val = u == 1 ? w[(((size_t)ci * cin + m) * 3 + ky) * 3 + kx] : 0.f;
            }
            else if (transpose_flip == 2) val = w[(((size_t)ci * cin + m) * 3 + (2 - v)) * 3 + (2 - u)];
            else if (transpose_flip) val = w[(((size_t)ci * cin + m) * 3 + (2 - u)) * 3 + (2 - v)];
            else                     val = w[(((size_t)m * cin + ci) * 3 + u) * 3 + v];
            q[e >> 2][e & 3] = val;
        }
        uint32_t h[8], l[4], x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wsu_split4_f16f8(q[k], WSU_F8_WLO_DIV, WSU_F8_W_DIV, h[2 * k], h[2 * k + 1], l[k], x[k]);
        char* base = dst + (((size_t)cb * nch + c) * 9 + tap) * (WSU_GRAN * WSU_COB * 16) + co * 16;
        *reinterpret_cast<u32x4*>(base) = mk_u4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<u32x4*>(base + WSU_COB * 16) = mk_u4(h[4], h[5], h[6], h[7]);
        *reinterpret_cast<u32x4*>(base + 2 * WSU_COB * 16) = mk_u4(x[0], x[1], x[2], x[3]);        // plane 2: e4m3(w), meets the residuals of x
        *reinterpret_cast<u32x4*>(base + 3 * WSU_COB * 16) = mk_u4(l[0], l[1], l[2], l[3]);        // plane 3: residuals of w, meet e4m3(x)
    }
}

int pack_impl(const float* w, void* dst, int cin, int cout, int mode, int tf, void* stream) {
    if (mode == WSU_MODE_F16F8X) mode = WSU_MODE_F16F8;
    const int kin = tf ? cout : cin, mout = tf ? cin : cout;
    WSU_REQUIRE(w && dst, "conv3x3_pack: null pointer");
    WSU_REQUIRE((mode >= 0 && mode <= 2) || mode == WSU_MODE_F16F8, "conv3x3_pack: bad mode %d", mode);
    WSU_REQUIRE(kin > 0 && kin % wsu_chunk_channels(mode) == 0, "conv3x3_pack: reduction channels %d not a multiple of %d", kin, wsu_chunk_channels(mode));
    WSU_REQUIRE(mout > 0 && mout % WSU_COB == 0, "conv3x3_pack: output channels %d not a multiple of %d", mout, WSU_COB);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = 1024;
    if (mode == WSU_MODE_F16F8) hipLaunchKernelGGL(pack_conv3x3_f16f8_kernel, dim3(blocks), dim3(256), 0, s, w, (char*)dst, cin, cout, tf);
    else if (mode == WSU_MODE_F32) hipLaunchKernelGGL(pack_conv3x3_kernel<WSU_MODE_F32>, dim3(blocks), dim3(256), 0, s, w, (char*)dst, cin, cout, tf);
    else if (mode == WSU_MODE_BF16X3) hipLaunchKernelGGL(pack_conv3x3_kernel<WSU_MODE_BF16X3>, dim3(blocks), dim3(256), 0, s, w, (char*)dst, cin, cout, tf);
    else hipLaunchKernelGGL(pack_conv3x3_kernel<WSU_MODE_BF16>, dim3(blocks), dim3(256), 0, s, w, (char*)dst, cin, cout, tf);
    return wsu_check_launch("pack_conv3x3_kernel");
}

}  // namespace

extern "C" {

// diagnostic only (not part of include/wsu.h): copy the in-kernel stamps of the last conv3x3 v1 launch to the host
int wsu_debug_read_stamps(unsigned long long* host_dst, int nblocks) {
    if (nblocks > 2048) nblocks = 2048;
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), (size_t)nblocks * WSU_NSTAMP * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}

size_t wsu_conv3x3_packed_bytes(int cin, int cout, int mode) {
    if (cin <= 0 || cout <= 0 || mode < 0 || (mode > 2 && mode != WSU_MODE_F16F8 && mode != WSU_MODE_F16F8X)) return 0;
    const size_t per_elem = mode == WSU_MODE_BF16 ? 2 : 4;    // BF16X3 stores hi + lo bf16, F16F8 f16 + two e4m3 = 4 bytes
    return (size_t)cin * cout * 9 * per_elem;
}

int wsu_conv3x3_pack(const float* w_oihw, void* w_packed, int cin, int cout, int mode, void* stream) {
    return pack_impl(w_oihw, w_packed, cin, cout, mode, 0, stream);
}

// internal (backward.hip): data-gradient weights with the row / column tap axes swapped, for the transposed left / right border strips
int wsu_conv3x3_pack_dgrad_swapped(const float* w_oihw, void* w_packed, int cin, int cout, int mode, void* stream) {
    return pack_impl(w_oihw, w_packed, cin, cout, mode, 2, stream);
}

int wsu_conv3x3_pack_dgrad(const float* w_oihw, void* w_packed, int cin, int cout, int mode, void* stream) {
    return pack_impl(w_oihw, w_packed, cin, cout, mode, 1, stream);
}

// The four border-ring weight sets (top, bottom, left, right) of the planar data gradient, 4 x wsu_conv3x3_packed_bytes(cin, cout, F16F8)
// bytes: wsu_conv3x3_pl_bwd_data runs them over strips of the gradient's border rows / columns (reflect-padding adjoint).
int wsu_conv3x3_pack_ring(const float* w_oihw, void* w_packed, int cin, int cout, void* stream) {
    const size_t set = (size_t)cin * cout * 9 * 4;
    for (int k = 0; k < 4; ++k) {
        int rc = pack_impl(w_oihw, (char*)w_packed + k * set, cin, cout, WSU_MODE_F16F8, 3 + k, stream);
        if (rc) return rc;
    }
    return 0;
}

// Extended launcher shared by the forward op and the data-gradient op (wsu_conv3x3_bwd_data in conv3x3_bwd.hip).
static int conv3x3_launch_full(const void* x1, const void* x2, const void* w_packed, const float* bias,
                          void* y, void* y2, int csplit, void* y_pool, uint8_t* pool_idx,
                          const void* relu_mask, const void* relu_mask2,
                          const float* head_w, const float* head_b, float* head_out, float* head_logit, int head_cout,
                          int n, int h, int w, int c1, int c2, int cout,
                          int mode, int relu, int pad_zero, void* stream,
                          const float* first_img = nullptr, const float* first_w = nullptr, const float* first_b = nullptr) {
    WSU_REQUIRE(mode >= 0 && mode <= 5, "conv3x3: bad mode %d", mode);
    const bool presplit = mode == WSU_MODE_BF16X3S;             // split-bf16 arithmetic on activations stored already split
    if (presplit) mode = WSU_MODE_BF16X3;
    const bool f16f8x = mode == WSU_MODE_F16F8X;                // F16F8 arithmetic on fp32 storage (training forward: pool_idx allowed)
    if (f16f8x) mode = WSU_MODE_F16F8;
    WSU_REQUIRE(!f16f8x || (!first_img && !head_w), "conv3x3: mode F16F8X runs on fp32 tensors (no fused first layer / head)");
    WSU_REQUIRE(!(presplit || (mode == WSU_MODE_F16F8 && !f16f8x)) || (!pool_idx && !relu_mask && !relu_mask2 && !y2 && !pad_zero),
                "conv3x3: modes BF16X3S / F16F8 are forward inference formats (no pool_idx, ReLU masks, split outputs or zero padding)");
    const int ck = wsu_chunk_channels(mode);
    WSU_REQUIRE((x1 || first_img) && w_packed && (y || head_w), "conv3x3: null pointer");
    WSU_REQUIRE(!first_img || (first_w && !x1 && !x2 && c1 == 64 && c2 == 0 && !pad_zero && !relu_mask),
                "conv3x3: the fused first layer feeds exactly 64 channels into a reflect-padded forward conv");
    WSU_REQUIRE(!head_w || (head_out && cout == WSU_COB && head_cout >= 1 && head_cout <= 4 && !y2),
                "conv3x3: fused head needs cout == %d, 1..4 head planes and an output pointer", WSU_COB);
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
    a.head_w = head_w; a.head_b = head_b; a.head_out = head_out; a.head_logit = head_logit; a.head_cout = head_cout;
    a.img = first_img; a.w1 = first_w; a.b1 = first_b;
    a.out_split = presplit || (mode == WSU_MODE_F16F8 && !f16f8x);
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("WSU_CONV_ABLATE"); ablate = e ? atoi(e) : 0; }
    a.ablate = ablate;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_F32) return launch_conv<WSU_MODE_F32>(a, s);
    if (mode == WSU_MODE_BF16X3) return launch_conv<WSU_MODE_BF16X3>(a, s, presplit && !first_img);
    if (mode == WSU_MODE_F16F8) return launch_conv<WSU_MODE_F16F8>(a, s, !f16f8x);
    return launch_conv<WSU_MODE_BF16>(a, s);
}

// e11 + e12 (+pool) in one launch for single-plane inputs: the first layer's 64 channels (unet.py:141) are computed while staging and
// never reach HBM.  img: (N,1,H,W) fp32, w1: (64,1,3,3), b1: (64) or NULL; the rest as wsu_conv3x3_fwd with cin = 64.
int wsu_conv3x3_fused_first_fwd(const float* img, const float* w1, const float* b1, const void* w_packed, const float* bias,
                                void* y, void* y_pool, uint8_t* pool_idx, int n, int h, int w, int cout, int mode, int relu, void* stream) {
    WSU_REQUIRE(img && w1, "conv3x3_fused_first: null pointer");
    return conv3x3_launch_full(nullptr, nullptr, w_packed, bias, y, nullptr, cout, y_pool, pool_idx, nullptr, nullptr,
                               nullptr, nullptr, nullptr, nullptr, 0, n, h, w, 64, 0, cout, mode, relu, 0, stream, img, w1, b1);
}

size_t wsu_conv3x3_wino_packed_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0 || cin % 16 || cout % WSU_COB) return 0;
    return (size_t)(cout / WSU_COB) * (cin / 16) * WN_LDS_W;
}

int wsu_conv3x3_wino_pack(const float* w_oihw, void* w_packed, int cin, int cout, void* stream) {
    WSU_REQUIRE(w_oihw && w_packed, "conv3x3_wino_pack: null pointer");
    WSU_REQUIRE(cin > 0 && cin % 16 == 0 && cout > 0 && cout % WSU_COB == 0, "conv3x3_wino_pack: cin=%d must be a multiple of 16, cout=%d of %d", cin, cout, WSU_COB);
    hipLaunchKernelGGL(pack_wino_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw, (uint16_t*)w_packed, cin, cout);
    return wsu_check_launch("pack_wino_kernel");
}

// Forward 3x3 conv in mode bf16x3 through the Winograd F(2,3) kernel: same arguments and fusions as wsu_conv3x3_fwd /
// wsu_conv3x3_head_fwd (head_w == NULL: no head), weights from wsu_conv3x3_wino_pack, fp32 NHWC activations.
int wsu_conv3x3_wino_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias, void* y,
                         void* y_pool, uint8_t* pool_idx,
                         const float* head_w, const float* head_b, float* head_out, float* head_logit, int head_cout,
                         int n, int h, int w, int c1, int c2, int cout, int relu, void* stream) {
    WSU_REQUIRE(x1 && w_packed && (y || head_w), "conv3x3_wino: null pointer");
    WSU_REQUIRE(!head_w || (head_out && cout == WSU_COB && head_cout >= 1 && head_cout <= 4), "conv3x3_wino: fused head needs cout == %d and 1..4 head planes", WSU_COB);
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_wino: bad shape n=%d h=%d w=%d (reflect pad 1 needs h,w >= 2)", n, h, w);
    WSU_REQUIRE(c1 > 0 && c1 % 16 == 0 && c2 >= 0 && c2 % 16 == 0 && (c2 == 0) == (x2 == nullptr), "conv3x3_wino: c1=%d c2=%d must be multiples of 16", c1, c2);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0, "conv3x3_wino: cout=%d must be a multiple of %d", cout, WSU_COB);
    WSU_REQUIRE((long long)n * h * w < 0x7FFFFFFFLL, "conv3x3_wino: n*h*w overflows int32");
    WSU_REQUIRE(!(pool_idx && !y_pool), "conv3x3_wino: pool_idx without y_pool");
    ConvArgs a{};
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.wp = (const char*)w_packed; a.bias = bias;
    a.y = (char*)y; a.y2 = nullptr; a.ypool = (char*)y_pool; a.pidx = pool_idx;
    a.n = n; a.h = h; a.w = w; a.c1 = c1; a.c2 = c2; a.cout = cout; a.csplit = cout;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + WN_TH - 1) / WN_TH; a.ncb = cout / WSU_COB;
    a.nch1 = c1 / 16; a.nch = (c1 + c2) / 16;
    a.relu = relu; a.pad_zero = 0;
    static int ablate = -1;                                            // timing-only experiment mask, 0 in production
    if (ablate < 0) { const char* e = getenv("WSU_CONV_ABLATE"); ablate = e ? atoi(e) : 0; }
    a.ablate = ablate;
    a.head_w = head_w; a.head_b = head_b; a.head_out = head_out; a.head_logit = head_logit; a.head_cout = head_cout;
    constexpr int EPI_BYTES = WN_TH * TW * Epi<WSU_MODE_BF16X3>::STRIDE;
    constexpr int LDS = EPI_BYTES > WN_LDS_V + WN_LDS_W ? EPI_BYTES : WN_LDS_V + WN_LDS_W;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3_wino): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        attr_done = true;
    }
    const long long nblk = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nblk > 0 && nblk <= 0x7FFFFFFFLL, "conv3x3_wino: grid of %lld workgroups out of range", nblk);
    hipLaunchKernelGGL(conv3x3_wino_kernel, dim3((unsigned)nblk), dim3(WN_NT), LDS, static_cast<hipStream_t>(stream), a);
    return wsu_check_launch("conv3x3_wino_kernel");
}

int wsu_conv3x3_launch_ex(const void* x1, const void* x2, const void* w_packed, const float* bias,
                          void* y, void* y2, int csplit, void* y_pool, uint8_t* pool_idx,
                          const void* relu_mask, const void* relu_mask2,
                          int n, int h, int w, int c1, int c2, int cout,
                          int mode, int relu, int pad_zero, void* stream) {
    return conv3x3_launch_full(x1, x2, w_packed, bias, y, y2, csplit, y_pool, pool_idx, relu_mask, relu_mask2,
                               nullptr, nullptr, nullptr, nullptr, 0, n, h, w, c1, c2, cout, mode, relu, pad_zero, stream);
}

// K1 + K5 fused: out = sigmoid(head_w . relu(conv3x3(cat[x1,x2]) + bias) + head_b), NCHW fp32; the 64-channel conv output is
// stored only if y != NULL.  Replaces d42 + outconv + sigmoid (unet.py:186,189) in one launch.
int wsu_conv3x3_head_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias, void* y,
                         const float* head_w, const float* head_b, float* out, float* logit,
                         int n, int h, int w, int c1, int c2, int cout, int head_cout, int mode, void* stream) {
    WSU_REQUIRE(head_w && out, "conv3x3_head: null pointer");
    return conv3x3_launch_full(x1, x2, w_packed, bias, y, nullptr, cout, nullptr, nullptr, nullptr, nullptr,
                               head_w, head_b, out, logit, head_cout, n, h, w, c1, c2, cout, mode, 1, 0, stream);
}

int wsu_conv3x3_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias,
                    void* y, void* y_pool, uint8_t* pool_idx,
                    int n, int h, int w, int c1, int c2, int cout,
                    int mode, int relu, int pad_zero, void* stream) {
    return wsu_conv3x3_launch_ex(x1, x2, w_packed, bias, y, nullptr, cout, y_pool, pool_idx, nullptr, nullptr,
                                 n, h, w, c1, c2, cout, mode, relu, pad_zero, stream);
}

}  // extern "C"
