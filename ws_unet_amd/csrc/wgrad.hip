// K7 (weight / bias gradients) of the 3x3 reflect convolution and of the 2x2 transposed convolution,
// exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32), deterministic.
//
// autograd of nn.Conv2d(k3, reflect) / nn.ConvTranspose2d(k2, s2) in the reference's train step
// (src/unet/model/unet.py:82-132; loop pattern src/detector/train.py:55-95):
//   conv :  dW[co, ci, u, v] = sum_{n,y,x} g[n,y,x,co] * xpad[n, y+u-1, x+v-1, ci]      db[co] = sum g[.., co]
//   convT:  dW[ci, co, a, b] = sum_{n,i,j} x[n,i,j,ci] * dy[n, 2i+a, 2j+b, co]          db[co] = sum dy[.., co]
// Both are GEMMs with the PIXEL index as the reduction dimension:  D_t[m, n] = sum_p U[p, m] * V_t[p, n]
//   conv :  U = g  (m = co),  V_t = x shifted by tap t (n = ci; two sources for the fused concat), 9 taps
//   convT:  U = x  (m = ci),  V_t = dy gathered at sub-position t (n = co),                          4 taps
// With NHWC fp32 tiles in LDS as [pixel][64 channels], lane (i = lane&31, k = lane>>5) of the 32x32x2 MFMA reads
// U[pixel 2s+k][i] / V[pixel' 2s+k][i]: 32 consecutive floats per half-wave, conflict free, no transpose.
// One workgroup = 64 m x 64 n x all taps, 4 waves as 2x2 of 32x32 tiles, NTAPS accumulator tiles per wave.
// Split-K over pixel tiles: workgroup (split, mb, nb) walks its share of the tiles and writes one partial
// slab; wgrad_reduce sums the slabs in a fixed order into the OIHW / IOHW gradient (bitwise reproducible --
// no float atomics).
#include "wsu_device.h"
#include <cstdlib>

namespace {

constexpr int NT = 256;
constexpr int TW = 32;

struct WgArgs {
    const float* u;                 // (N, Hu, Wu, cu)  unshifted operand
    const float* v1; const float* v2;   // gathered operand(s): (N, Hv, Wv, cv1) [, (N, Hv, Wv, cv2)]
    float* part;                    // [nsplit][nmb][nnb][NTAPS][64][64]
    float* bpart;                   // [nsplit][nmb][64] column sums of U (written by nb == 0 workgroups) or null
    int n, hu, wu, cu, cv1, cv2;
    int tiles_x, tiles_y, ntiles, nsplit, nmb, nnb, tiles_per_split;
};

// KIND 0: conv3x3 (TH = 2, V tile (TH+2) x (TW+2), reflect);  KIND 1: convT2x2 (TH = 1, V tile 2TH x 2TW)
template <int KIND> struct Geo {
    static constexpr int NTAPS = KIND == 0 ? 9 : 4;
    static constexpr int TH = KIND == 0 ? 2 : 1;
    static constexpr int VH = KIND == 0 ? TH + 2 : 2 * TH;
    static constexpr int VW = KIND == 0 ? TW + 2 : 2 * TW;
    static constexpr int U_BYTES = TH * TW * 256;
    static constexpr int V_BYTES = VH * VW * 256;
    static constexpr int LDS = U_BYTES + V_BYTES;
};

template <int KIND>
__global__ __launch_bounds__(NT, 2) void wgrad_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<KIND>;
    constexpr int NTAPS = G::NTAPS, TH = G::TH, VH = G::VH, VW = G::VW;
    float* ul = reinterpret_cast<float*>(smem);                    // [TH*TW][64]
    float* vl = reinterpret_cast<float*>(smem + G::U_BYTES);       // [VH*VW][64]

    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int nb = b % a.nnb; b /= a.nnb;
    const int mb = b % a.nmb;
    const int split = b / a.nmb;
    const int hv = KIND == 0 ? a.hu : 2 * a.hu, wv = KIND == 0 ? a.wu : 2 * a.wu;
    const float* vsrc; int cv, vch0;
    if (nb * 64 < a.cv1) { vsrc = a.v1; cv = a.cv1; vch0 = nb * 64; }
    else                 { vsrc = a.v2; cv = a.cv2; vch0 = nb * 64 - a.cv1; }

    const int wv_ = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wm = wv_ >> 1, wn = wv_ & 1;
    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;                                               // column sum of U for channel tid&63, pixels == tid>>6 mod 4

    const int t0 = split * a.tiles_per_split;
    const int t1 = min(t0 + a.tiles_per_split, a.ntiles);
    for (int tile = t0; tile < t1; ++tile) {
        int tt = tile;
        const int tx = tt % a.tiles_x; tt /= a.tiles_x;
        const int ty = tt % a.tiles_y;
        const int n = tt / a.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        __syncthreads();                                            // previous tile fully consumed
        // ---- stage U (zero outside the image) and V (reflect / clamp: finite values, weighted by U == 0 outside)
        for (int i = tid; i < TH * TW * 16; i += NT) {
            const int p = i >> 4, gq = i & 15;
            const int r = p / TW, c = p % TW;
            u32x4 val = mk_u4(0, 0, 0, 0);
            if (y0 + r < a.hu && x0 + c < a.wu)
                val = *reinterpret_cast<const u32x4*>(a.u + ((size_t)(n * a.hu + y0 + r) * a.wu + x0 + c) * a.cu + mb * 64 + gq * 4);
            *reinterpret_cast<u32x4*>(ul + p * 64 + gq * 4) = val;
        }
        for (int i = tid; i < VH * VW * 16; i += NT) {
            const int p = i >> 4, gq = i & 15;
            const int r = p / VW, c = p % VW;
            int yy, xx;
            if (KIND == 0) { yy = wsu_reflect(y0 - 1 + r, hv); xx = wsu_reflect(x0 - 1 + c, wv); }
            else           { yy = min(2 * y0 + r, hv - 1);     xx = min(2 * x0 + c, wv - 1); }
            *reinterpret_cast<u32x4*>(vl + p * 64 + gq * 4) =
                *reinterpret_cast<const u32x4*>(vsrc + ((size_t)(n * hv + yy) * wv + xx) * cv + vch0 + gq * 4);
        }
        __syncthreads();
        if (a.bpart && nb == 0) {
            const int ch = tid & 63;
            for (int p = tid >> 6; p < TH * TW; p += 4) bsum += ul[p * 64 + ch];
        }
        // ---- MFMA over pixel pairs
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            for (int c2 = 0; c2 < TW / 2; ++c2) {
                const int c = 2 * c2 + hh;
                const float av = ul[(r * TW + c) * 64 + wm * 32 + l31];
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    int vp;
                    if (KIND == 0) vp = (r + t / 3) * VW + c + t % 3;
                    else           vp = (2 * r + (t >> 1)) * VW + 2 * c + (t & 1);
                    const float bv = vl[vp * 64 + wn * 32 + l31];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // ---- partial slab: part[((split*nmb + mb)*nnb + nb)*NTAPS + t][m][n]
    float* dst = a.part + ((size_t)((split * a.nmb + mb) * a.nnb + nb) * NTAPS) * 4096;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            dst[(size_t)t * 4096 + m * 64 + wn * 32 + l31] = acc[t][r];
        }
    if (a.bpart && nb == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64) a.bpart[(size_t)(split * a.nmb + mb) * 64 + tid] = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
    }
}

// ---------------------------------------------------------------------------------------------------
// Split-bf16 variant (train_mode 'bf16x3'): same GEMM, operands split hi + lo bf16 while staging, 3 bf16 MFMAs per
// product (lo*hi, hi*lo, hi*hi), fp32 accumulate -- ~3x the fp32-MFMA rate at ~2^-17 relative error.
// The reduction index is the PIXEL, so each lane of v_mfma_f32_32x32x16_bf16 needs 8 consecutive pixels of ONE channel.
// The LDS tiles stay row-major [pixel][64 ch] (coalesced 16-byte staging writes); the transposition is done by the
// hardware: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of the 4 rows whose addresses the group's lanes
// supplied (lane 4q+p -> row q, columns 4p..4p+3; semantics probed on the device: tools/probe_tr16.hip).  Two such reads
// give the 8 k-values of a fragment.  Rows may be any pixels, so the stride-2 gather of the transposed conv is free.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int X3_ROW = 144;                       // bytes per pixel row: 64 bf16 + 16 pad (16-byte aligned rows)

constexpr int F8_ROW = 96;                        // bytes per pixel row of an e4m3 image: 64 + 32 pad (conflict-free transposing reads)

// THX: tile rows (default: Geo's; the planar transposed-conv kernel with f16 products takes 2 -- its LDS images are small enough)
template <int KIND, int THX = Geo<KIND>::TH> struct GeoX3 {
    static_assert(KIND == 1 || THX == Geo<KIND>::TH, "only the transposed conv's tile height is a parameter");
    static constexpr int NTAPS = Geo<KIND>::NTAPS, TH = THX, VH = KIND == 0 ? TH + 2 : 2 * TH, VW = Geo<KIND>::VW;
    static constexpr int U_PIX = TH * TW, V_PIX = VH * VW;
    static constexpr int U_BYTES = U_PIX * X3_ROW, V_BYTES = V_PIX * X3_ROW;      // per hi / lo image
    static constexpr int LDS = 2 * U_BYTES + 2 * V_BYTES;
    // F8 variant (mode F16F8X): one f16 image + two e4m3 images (copy, residual) per operand
    static constexpr int U8_BYTES = U_PIX * F8_ROW, V8_BYTES = V_PIX * F8_ROW;
    static constexpr int LDS_F8 = U_BYTES + 2 * U8_BYTES + V_BYTES + 2 * V8_BYTES;
};

// F8 variant: 16 k-values (pixels row0.., row step `rstep`) of the byte column `col + lane&15 (+16 for odd groups)` of an e4m3 image:
// ds_read_b64_tr_b8 hands lane i of a 16-lane group column i of the 8 rows whose addresses the group's lanes supplied (lane 2q+p -> row
// q, columns 8p..8p+7; tools/probe_tr8.hip); two reads = one 16-byte scale-block half of v_mfma_scale_f32_32x32x64_f8f6f4.
typedef __attribute__((ext_vector_type(2))) int i32x2_t;
typedef __attribute__((address_space(3))) i32x2_t lds_i32x2;
__device__ __forceinline__ u32x4 f8_frag(const char* img, int row0, int rstep, int colbyte) {
    const int lane = threadIdx.x & 63, li = lane & 15, q = li >> 1, pp = li & 1;
    const char* a0 = img + (row0 + q * rstep) * F8_ROW + colbyte + pp * 8;
    const i32x2_t r0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)a0);
    const i32x2_t r1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(a0 + 8 * rstep * F8_ROW));
    return mk_u4((uint32_t)r0[0], (uint32_t)r0[1], (uint32_t)r1[0], (uint32_t)r1[1]);
}
// 8 fp32 channels of one pixel -> f16 row piece (16 B) + e4m3 copy and residual row pieces (8 B each).  `grad`: the operand is the
// (power-of-two scaled, see model/autograd.py) gradient: copy = e4m3(v * 4), residual = e4m3((v - f16 v) * 2^14); else the activation
// encoding of wsu_device.h (copy = e4m3(v / 4), residual * 2^12).
__device__ __forceinline__ void f8_stage(char* hi, char* c8, char* l8, int p, int cg, const f32x4& s0, const f32x4& s1, bool grad) {
    const float dlo = grad ? 0x1p-14f : WSU_F8_XLO_DIV, dx = grad ? 0.25f : WSU_F8_X_DIV;
    uint32_t h0, h1, h2, h3, l0, l1, x0, x1;
    wsu_split4_f16f8(s0, dlo, dx, h0, h1, l0, x0);
    wsu_split4_f16f8(s1, dlo, dx, h2, h3, l1, x1);
    *reinterpret_cast<u32x4*>(hi + p * X3_ROW + cg * 16) = mk_u4(h0, h1, h2, h3);
    *reinterpret_cast<u32x2*>(c8 + p * F8_ROW + cg * 8) = mk_u2(x0, x1);
    *reinterpret_cast<u32x2*>(l8 + p * F8_ROW + cg * 8) = mk_u2(l0, l1);
}

// 8 k-values (pixels row0.., row step `rstep` pixels) of channel column `col0 + lane&15 (+16 for odd groups)`
__device__ __forceinline__ u32x4 x3_frag(const char* img, int row0, int rstep, int colbyte) {
    const int lane = threadIdx.x & 63, li = lane & 15, q = li >> 2, pp = li & 3;
    const char* a0 = img + (row0 + q * rstep) * X3_ROW + colbyte + pp * 8;
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * rstep * X3_ROW));
    const u32x2 lo = __builtin_bit_cast(u32x2, r0), hi = __builtin_bit_cast(u32x2, r1);
    return mk_u4(lo.x, lo.y, hi.x, hi.y);
}

// F8 = true (mode F16F8X): hi*hi on v_mfma_f32_32x32x16_f16 per 16 pixels, both cross terms of 32 pixels in one block-scaled fp8 MFMA
// (4 instead of 6 matrix units per 32 pixels and tap); the gradient operand (U for the conv, V for the transposed conv) arrives scaled.
template <int KIND, bool F8 = false>
__global__ __launch_bounds__(NT, 2) void wgrad_x3_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = GeoX3<KIND>;
    constexpr int NTAPS = G::NTAPS, TH = G::TH, VH = G::VH, VW = G::VW;
    char* u_hi = smem;
    char* u_lo = smem + G::U_BYTES;                        // F8: e4m3 copy image, then (u_l8) the residual image
    char* u_l8 = u_lo + G::U8_BYTES;
    char* v_hi = F8 ? smem + G::U_BYTES + 2 * G::U8_BYTES : smem + 2 * G::U_BYTES;
    char* v_lo = v_hi + G::V_BYTES;
    char* v_l8 = v_lo + G::V8_BYTES;
    constexpr bool UGRAD = KIND == 0;                      // which operand is the gradient

    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int nb = b % a.nnb; b /= a.nnb;
    const int mb = b % a.nmb;
    const int split = b / a.nmb;
    const int hv = KIND == 0 ? a.hu : 2 * a.hu, wv = KIND == 0 ? a.wu : 2 * a.wu;
    const float* vsrc; int cv, vch0;
    if (nb * 64 < a.cv1) { vsrc = a.v1; cv = a.cv1; vch0 = nb * 64; }
    else                 { vsrc = a.v2; cv = a.cv2; vch0 = nb * 64 - a.cv1; }

    const int wv_ = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wm = wv_ >> 1, wn = wv_ & 1;
    const int colA = (wm * 32 + ((lane >> 4) & 1) * 16) * 2;         // byte offset of this 16-lane group's channel block
    const int colB = (wn * 32 + ((lane >> 4) & 1) * 16) * 2;
    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // bias gradient (exact fp32 column sums of U = g) rides on the staging loop of the workgroups with nb == 0: a thread always
    // stages the same 8-channel group (NT is a multiple of 8), so it keeps 8 running sums and the 32 threads of a group meet in LDS.
    const bool bias_on = KIND == 0 && a.bpart != nullptr && nb == 0;
    f32x4 bs0 = mk_f4(0.f, 0.f, 0.f, 0.f), bs1 = bs0;

    const int t0 = split * a.tiles_per_split;
    const int t1 = min(t0 + a.tiles_per_split, a.ntiles);
    // KIND 0 walks its tiles DOWN the image columns (ty fastest): vertically adjacent tiles share VH - TH = 2 input rows, which
    // stay in LDS (rolling row slots, `rot`), and the TH new rows + the next U tile are fetched into registers while the current
    // tile is multiplied.  Only the first tile of a split / of a column stages everything directly.
    constexpr bool ROLL = KIND == 0;
    constexpr int U_IT = (G::U_PIX * 8 + NT - 1) / NT;                 // U items of 8 channels per thread (exact: 2)
    constexpr int VN_ROW0 = ROLL ? VH - TH : 0;                        // first window row that is new per step (KIND 1: all rows)
    constexpr int VN_PIX = (VH - VN_ROW0) * VW;
    constexpr int VN_IT = (VN_PIX * 8 + NT - 1) / NT;                  // 3 (KIND 0), 4 (KIND 1)
    static_assert((G::U_PIX * 8) % NT == 0, "U tile must divide evenly over the threads");
    f32x4 pu[U_IT][2], pv[VN_IT][2];
    bool have_pref = false;
    int rot = 0;

    auto decode = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
        int tt = tile;
        if (ROLL) { const int ty = tt % a.tiles_y; tt /= a.tiles_y; const int tx = tt % a.tiles_x; n = tt / a.tiles_x; y0 = ty * TH; x0 = tx * TW; }
        else      { const int tx = tt % a.tiles_x; tt /= a.tiles_x; const int ty = tt % a.tiles_y; n = tt / a.tiles_y; y0 = ty * TH; x0 = tx * TW; }
    };
    auto u_src = [&](int n, int y0, int x0, int i, bool& ok) __attribute__((always_inline)) {
        const int p = i >> 3, cg = i & 7, r = p / TW, c = p % TW;
        ok = y0 + r < a.hu && x0 + c < a.wu;
        return reinterpret_cast<const f32x4*>(a.u + ((size_t)(n * a.hu + y0 + r) * a.wu + x0 + c) * a.cu + mb * 64 + cg * 8);
    };
    auto v_src = [&](int n, int y0, int x0, int p, int cg) __attribute__((always_inline)) {
        const int r = p / VW, c = p % VW;
        int yy, xx;
        if (KIND == 0) { yy = wsu_reflect(y0 - 1 + r, hv); xx = wsu_reflect(x0 - 1 + c, wv); }
        else           { yy = min(2 * y0 + r, hv - 1);     xx = min(2 * x0 + c, wv - 1); }
        return reinterpret_cast<const f32x4*>(vsrc + ((size_t)(n * hv + yy) * wv + xx) * cv + vch0 + cg * 8);
    };

    for (int tile = t0; tile < t1; ++tile) {
        int n, y0, x0;
        decode(tile, n, y0, x0);
        __syncthreads();
        if (!have_pref) {
            // ---- direct staging: 8 fp32 channels per item -> bf16 hi row piece + bf16 lo row piece (16 B each)
            rot = 0;
            for (int i = tid; i < G::U_PIX * 8; i += NT) {
                const int p = i >> 3, cg = i & 7;
                bool ok;
                const f32x4* src = u_src(n, y0, x0, i, ok);
                u32x4 hi = mk_u4(0, 0, 0, 0), lo = hi;
                f32x4 s0 = mk_f4(0.f, 0.f, 0.f, 0.f), s1 = s0;
                if (ok) {
                    s0 = src[0]; s1 = src[1];
                    if (bias_on) { bs0 = bs0 + s0; bs1 = bs1 + s1; }
                    if constexpr (!F8) wsu_split8(s0, s1, hi, lo);
                }
                if constexpr (F8) f8_stage(u_hi, u_lo, u_l8, p, cg, s0, s1, UGRAD);
                else {
                    *reinterpret_cast<u32x4*>(u_hi + p * X3_ROW + cg * 16) = hi;
                    *reinterpret_cast<u32x4*>(u_lo + p * X3_ROW + cg * 16) = lo;
                }
            }
            for (int i = tid; i < G::V_PIX * 8; i += NT) {
                const int p = i >> 3, cg = i & 7;
                const f32x4* src = v_src(n, y0, x0, p, cg);
                if constexpr (F8) f8_stage(v_hi, v_lo, v_l8, p, cg, src[0], src[1], !UGRAD);
                else {
                    u32x4 hi, lo;
                    wsu_split8(src[0], src[1], hi, lo);
                    *reinterpret_cast<u32x4*>(v_hi + p * X3_ROW + cg * 16) = hi;
                    *reinterpret_cast<u32x4*>(v_lo + p * X3_ROW + cg * 16) = lo;
                }
            }
        } else {
            // ---- commit the prefetched registers: the whole U tile and the TH new V rows (window rows VH-TH .. VH-1)
            if (ROLL) rot = (rot + TH) & (VH - 1);
#pragma unroll
            for (int k = 0; k < U_IT; ++k) {
                const int i = tid + k * NT, p = i >> 3, cg = i & 7;
                if (bias_on) { bs0 = bs0 + pu[k][0]; bs1 = bs1 + pu[k][1]; }
                if constexpr (F8) f8_stage(u_hi, u_lo, u_l8, p, cg, pu[k][0], pu[k][1], UGRAD);
                else {
                    u32x4 hi, lo;
                    wsu_split8(pu[k][0], pu[k][1], hi, lo);
                    *reinterpret_cast<u32x4*>(u_hi + p * X3_ROW + cg * 16) = hi;
                    *reinterpret_cast<u32x4*>(u_lo + p * X3_ROW + cg * 16) = lo;
                }
            }
#pragma unroll
            for (int k = 0; k < VN_IT; ++k) {
                const int i = tid + k * NT;
                if (i < VN_PIX * 8) {
                    const int pn = i >> 3, cg = i & 7, rn = pn / VW, c = pn % VW;
                    const int slot = (VN_ROW0 + rn + rot) & (VH - 1);
                    if constexpr (F8) f8_stage(v_hi, v_lo, v_l8, slot * VW + c, cg, pv[k][0], pv[k][1], !UGRAD);
                    else {
                        u32x4 hi, lo;
                        wsu_split8(pv[k][0], pv[k][1], hi, lo);
                        *reinterpret_cast<u32x4*>(v_hi + (slot * VW + c) * X3_ROW + cg * 16) = hi;
                        *reinterpret_cast<u32x4*>(v_lo + (slot * VW + c) * X3_ROW + cg * 16) = lo;
                    }
                }
            }
        }
        __syncthreads();
        {
            // ---- prefetch for the next tile (KIND 0: only if it continues this column: same image, same tx, ty + 1)
            have_pref = tile + 1 < t1 && (!ROLL || (tile + 1) % a.tiles_y != 0);
            if (have_pref) {
                int n2, y2, x2;
                decode(tile + 1, n2, y2, x2);
#pragma unroll
                for (int k = 0; k < U_IT; ++k) {
                    bool ok;
                    const f32x4* src = u_src(n2, y2, x2, tid + k * NT, ok);
                    pu[k][0] = mk_f4(0.f, 0.f, 0.f, 0.f); pu[k][1] = pu[k][0];
                    if (ok) { pu[k][0] = src[0]; pu[k][1] = src[1]; }
                }
#pragma unroll
                for (int k = 0; k < VN_IT; ++k) {
                    const int i = tid + k * NT;
                    if (i < VN_PIX * 8) {
                        const f32x4* src = v_src(n2, y2, x2, VN_ROW0 * VW + (i >> 3), i & 7);
                        pv[k][0] = src[0]; pv[k][1] = src[1];
                    }
                }
            }
        }
        // ---- MFMA: k-steps of 16 pixels along a tile row (only the tap loop is unrolled: 144 accumulator + 40 prefetch registers
        //      leave no room for the fragment working set of several k-steps at once)
        if constexpr (F8) {
            // per tile row: two f16 k-steps of 16 pixels + ONE fp8 step of 32 pixels (scale block 0: e4m3(U) x residual(V), block 1:
            // residual(U) x e4m3(V); lanes 0-31 carry pixels 0-15 of the row, lanes 32-63 pixels 16-31)
            const int colA8 = wm * 32 + ((lane >> 4) & 1) * 16, colB8 = wn * 32 + ((lane >> 4) & 1) * 16;
            constexpr int SU8 = UGRAD ? 125 : WSU_F8_SCALE_X, SUL = UGRAD ? 113 : WSU_F8_SCALE_XLO;     // e4m3(g * 4), (g - f16 g) * 2^14
            constexpr int SV8 = UGRAD ? WSU_F8_SCALE_X : 125, SVL = UGRAD ? WSU_F8_SCALE_XLO : 113;
            const int sc_a = hh ? SUL : SU8, sc_b = hh ? SV8 : SVL;
#pragma unroll 1
            for (int r = 0; r < TH; ++r) {
                const u32x4 a8 = f8_frag(u_lo, r * TW + 16 * hh, 1, colA8), al8 = f8_frag(u_l8, r * TW + 16 * hh, 1, colA8);
                const u32x4 ah0 = x3_frag(u_hi, r * TW + 8 * hh, 1, colA), ah1 = x3_frag(u_hi, r * TW + 16 + 8 * hh, 1, colA);
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    int vrow, vstep;
                    if (KIND == 0) { vrow = ((r + t / 3 + rot) & (VH - 1)) * VW + t % 3; vstep = 1; }
                    else           { vrow = (2 * r + (t >> 1)) * VW + (t & 1); vstep = 2; }
                    const u32x4 bl8 = f8_frag(v_l8, vrow + vstep * 16 * hh, vstep, colB8), b8 = f8_frag(v_lo, vrow + vstep * 16 * hh, vstep, colB8);
                    wsu_mfma_f8x2(a8, al8, bl8, b8, sc_a, sc_b, acc[t]);
                    const u32x4 bh0 = x3_frag(v_hi, vrow + vstep * 8 * hh, vstep, colB), bh1 = x3_frag(v_hi, vrow + vstep * (16 + 8 * hh), vstep, colB);
                    wsu_mfma_f16(ah0, bh0, acc[t]);
                    wsu_mfma_f16(ah1, bh1, acc[t]);
                }
            }
        } else
#pragma unroll 1
        for (int r = 0; r < TH; ++r) {
#pragma unroll 1
            for (int c0 = 0; c0 < TW; c0 += 16) {
                const int up = r * TW + c0 + 8 * hh;                                    // first U pixel of this lane half
                const u32x4 ahi = x3_frag(u_hi, up, 1, colA), alo = x3_frag(u_lo, up, 1, colA);
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    int vp, vstep;
                    if (KIND == 0) { vp = ((r + t / 3 + rot) & (VH - 1)) * VW + c0 + 8 * hh + t % 3; vstep = 1; }
                    else           { vp = (2 * r + (t >> 1)) * VW + 2 * (c0 + 8 * hh) + (t & 1); vstep = 2; }
                    const u32x4 bhi = x3_frag(v_hi, vp, vstep, colB), blo = x3_frag(v_lo, vp, vstep, colB);
                    wsu_mfma_step<WSU_MODE_BF16X3>(alo, bhi, acc[t]);
                    wsu_mfma_step<WSU_MODE_BF16X3>(ahi, blo, acc[t]);
                    wsu_mfma_step<WSU_MODE_BF16X3>(ahi, bhi, acc[t]);
                }
            }
        }
    }
    float* dst = a.part + ((size_t)((split * a.nmb + mb) * a.nnb + nb) * NTAPS) * 4096;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            dst[(size_t)t * 4096 + m * 64 + wn * 32 + l31] = acc[t][r];
        }
    if (bias_on) {                                                     // uniform per workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);                   // [32 threads of a group][64 channels]
        const int cg = tid & 7, k = tid >> 3;
        *reinterpret_cast<f32x4*>(red + k * 64 + cg * 8) = bs0;
        *reinterpret_cast<f32x4*>(red + k * 64 + cg * 8 + 4) = bs1;
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
            for (int j = 0; j < NT / 8; ++j) sum += red[j * 64 + tid];
            if (KIND == 0) a.bpart[(size_t)split * (a.nmb * 64) + mb * 64 + tid] = sum;
            else           a.bpart[(size_t)split * (a.nnb * 64) + nb * 64 + tid] = sum;
        }
    }
}

// ---- PLANAR operands (mode F16F8P training path): the same workgroup shape, LDS images, matrix section and partial slabs as
// wgrad_x3_kernel<KIND, true>, but U / V arrive in the planar three-plane layout ([n][C/16][3][H][W][16 B]: f16 ch 0-7 | f16 ch 8-15 | e4m3
// residuals; gradients with the 2^14 residual scaling): a staging item = (pixel, 8-channel group) is one stored 16-byte f16 granule + 8 bytes
// of the residual granule -- no fp32 loads, no split arithmetic (the e4m3 copies are 3 instructions per pair from the f16 granule), and
// consecutive lanes fetch consecutive pixels of one plane (contiguous 512-byte runs).  The bias gradient (KIND 0) is the sum of the DECODED
// gradient values, accumulated per thread and 8-channel group while staging (fixed order -> deterministic).
struct WgPlArgs {
    const char* u; const char* v1; const char* v2;
    float* part; float* bpart;
    int n, hu, wu, cu, cv1, cv2;
    int tiles_x, tiles_y, ntiles, nsplit, nmb, nnb, tiles_per_split;
    int honly;                      // 1: f16 products only (wsu.h WSU_PRODUCTS_F16; ring kernel variant HONLY)
    int ablate;                     // timing-only experiments (WSU_WGRAD_ABLATE; results wrong when != 0): 1 = no matrix section, 2 = staging of the first tile only, 4 = no derived copies (the ring kernel's variant HONLY = products F16 is the product form of "no cross terms")
};

// HONLY (products = WSU_PRODUCTS_F16): the residual halves are not loaded, no e4m3 images are written or multiplied; bias sums of the f16 parts.
// THX (transposed conv only): input rows per tile -- one tile of TH = 1 is 8 MFMAs per wave between two workgroup barriers and a register
// prefetch of 20 KB; with f16 products the LDS images of two rows still fit twice per CU.
template <int KIND, bool HONLY = false, int THX = Geo<KIND>::TH>
__global__ __launch_bounds__(NT, 2) void wgrad_pl_kernel(const WgPlArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = GeoX3<KIND, THX>;
    constexpr int NTAPS = G::NTAPS, TH = G::TH, VH = G::VH, VW = G::VW;
    char* u_hi = smem;
    char* u_lo = smem + G::U_BYTES;                        // e4m3 copy image, then (u_l8) the residual image
    char* u_l8 = u_lo + G::U8_BYTES;
    char* v_hi = smem + G::U_BYTES + (HONLY ? 0 : 2 * G::U8_BYTES);   // HONLY: the e4m3 images do not exist (LDS = U_BYTES + V_BYTES)
    char* v_lo = v_hi + G::V_BYTES;
    char* v_l8 = v_lo + G::V8_BYTES;
    constexpr bool UGRAD = KIND == 0;                      // which operand is the gradient

    const int tid = threadIdx.x;
    int b = (int)wsu_xcd_remap(blockIdx.x, gridDim.x);      // the (mb, nb) workgroups of a split share their U / V tiles: same XCD, one L2
    const int nb = b % a.nnb; b /= a.nnb;
    const int mb = b % a.nmb;
    const int split = b / a.nmb;
    const int hv = KIND == 0 ? a.hu : 2 * a.hu, wv = KIND == 0 ? a.wu : 2 * a.wu;
    const char* vsrc; int cv, vch0;
    if (nb * 64 < a.cv1) { vsrc = a.v1; cv = a.cv1; vch0 = nb * 64; }
    else                 { vsrc = a.v2; cv = a.cv2; vch0 = nb * 64 - a.cv1; }
    const size_t hwu = (size_t)a.hu * a.wu, hwv = (size_t)hv * wv;

    const int wv_ = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wm = wv_ >> 1, wn = wv_ & 1;
    const int colA = (wm * 32 + ((lane >> 4) & 1) * 16) * 2;
    const int colB = (wn * 32 + ((lane >> 4) & 1) * 16) * 2;
    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    constexpr bool ROLL = KIND == 0;
    // staging items: thread = (8-channel group cg = tid >> 5, pixel lane tid & 31); its k-th item is pixel (tid & 31) + 32 k of the (sub)tile --
    // 32 consecutive lanes fetch 32 consecutive pixels of one plane, and a thread keeps ONE channel group (8 bias sums)
    constexpr int U_IT = G::U_PIX / 32;                                // 2 (KIND 0), 1 (KIND 1)
    constexpr int VN_ROW0 = ROLL ? VH - TH : 0;
    constexpr int VN_PIX = (VH - VN_ROW0) * VW;                        // 68 (KIND 0), 128 (KIND 1)
    constexpr int VN_IT = (VN_PIX + 31) / 32;                          // 3 (KIND 0), 4 (KIND 1)
    constexpr int V_IT = (G::V_PIX + 31) / 32;                         // 5 (KIND 0), 4 (KIND 1)
    static_assert(NT == 256 && G::U_PIX % 32 == 0, "8 channel groups x 32 pixel lanes");
    const int scg = tid >> 5, spx = tid & 31;
    // bias gradient = per-channel sum of the gradient operand: U for the conv (workgroups with nb == 0), V for the transposed conv (mb == 0)
    const bool bias_on = a.bpart != nullptr && (KIND == 0 ? nb == 0 : mb == 0);
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;

    struct Item { u32x4 h; u32x2 r; };
    Item pu[U_IT], pv[VN_IT];
    bool have_pref = false;
    int rot = 0;

    const int t0 = split * a.tiles_per_split;
    const int t1 = min(t0 + a.tiles_per_split, a.ntiles);
    auto decode = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
        int tt = tile;
        if (ROLL) { const int ty = tt % a.tiles_y; tt /= a.tiles_y; const int tx = tt % a.tiles_x; n = tt / a.tiles_x; y0 = ty * TH; x0 = tx * TW; }
        else      { const int tx = tt % a.tiles_x; tt /= a.tiles_x; const int ty = tt % a.tiles_y; n = tt / a.tiles_y; y0 = ty * TH; x0 = tx * TW; }
    };
    auto u_load = [&](int n, int y0, int x0, int p) __attribute__((always_inline)) {
        const int cg = scg, r = p / TW, c = p % TW;
        Item it; it.h = mk_u4(0, 0, 0, 0); it.r = mk_u2(0, 0);
        if (y0 + r < a.hu && x0 + c < a.wu) {
            const char* base = a.u + (((size_t)n * (a.cu >> 4) + mb * 4 + (cg >> 1)) * 3) * hwu * 16 + ((size_t)(y0 + r) * a.wu + x0 + c) * 16;
            it.h = *reinterpret_cast<const u32x4*>(base + (cg & 1) * hwu * 16);
            if constexpr (!HONLY) it.r = *reinterpret_cast<const u32x2*>(base + 2 * hwu * 16 + (cg & 1) * 8);
        }
        return it;
    };
    auto v_load = [&](int n, int y0, int x0, int p, int cg) __attribute__((always_inline)) {
        const int r = p / VW, c = p % VW;
        int yy, xx;
        if (KIND == 0) { yy = wsu_reflect(y0 - 1 + r, hv); xx = wsu_reflect(x0 - 1 + c, wv); }
        else           { yy = min(2 * y0 + r, hv - 1);     xx = min(2 * x0 + c, wv - 1); }
        const char* base = vsrc + (((size_t)n * (cv >> 4) + (vch0 >> 4) + (cg >> 1)) * 3) * hwv * 16 + ((size_t)yy * wv + xx) * 16;
        Item it;
        it.h = *reinterpret_cast<const u32x4*>(base + (cg & 1) * hwv * 16);
        if constexpr (HONLY) it.r = mk_u2(0, 0); else it.r = *reinterpret_cast<const u32x2*>(base + 2 * hwv * 16 + (cg & 1) * 8);
        return it;
    };
    auto v_valid = [&](int y0, int x0, int p) __attribute__((always_inline)) {    // KIND 1: the window pixel lies inside the image (clamped copies do not count)
        return 2 * y0 + p / VW < hv && 2 * x0 + p % VW < wv;
    };
    auto put = [&](char* hi, char* c8, char* l8, int p, int cg, const Item& it, bool grad) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4*>(hi + p * X3_ROW + cg * 16) = it.h;
        if constexpr (!HONLY) {
            *reinterpret_cast<u32x2*>(c8 + p * F8_ROW + cg * 8) = grad ? wsu_f16x8_to_fp8_grad(it.h) : wsu_f16x8_to_fp8(it.h);
            *reinterpret_cast<u32x2*>(l8 + p * F8_ROW + cg * 8) = it.r;
        }
    };
    auto bias_add = [&](float (&s)[8], const Item& it) __attribute__((always_inline)) {
        const f16x8 hv8 = __builtin_bit_cast(f16x8, it.h);
        if constexpr (HONLY) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)hv8[e];
            return;
        }
        const int r0 = (int)it.r.x, r1 = (int)it.r.y;
        s[0] += (float)hv8[0] + __builtin_amdgcn_cvt_f32_fp8(r0, 0) * WSU_F8_GLO_DIV;
        s[1] += (float)hv8[1] + __builtin_amdgcn_cvt_f32_fp8(r0, 1) * WSU_F8_GLO_DIV;
        s[2] += (float)hv8[2] + __builtin_amdgcn_cvt_f32_fp8(r0, 2) * WSU_F8_GLO_DIV;
        s[3] += (float)hv8[3] + __builtin_amdgcn_cvt_f32_fp8(r0, 3) * WSU_F8_GLO_DIV;
        s[4] += (float)hv8[4] + __builtin_amdgcn_cvt_f32_fp8(r1, 0) * WSU_F8_GLO_DIV;
        s[5] += (float)hv8[5] + __builtin_amdgcn_cvt_f32_fp8(r1, 1) * WSU_F8_GLO_DIV;
        s[6] += (float)hv8[6] + __builtin_amdgcn_cvt_f32_fp8(r1, 2) * WSU_F8_GLO_DIV;
        s[7] += (float)hv8[7] + __builtin_amdgcn_cvt_f32_fp8(r1, 3) * WSU_F8_GLO_DIV;
    };

    for (int tile = t0; tile < t1; ++tile) {
        int n, y0, x0;
        decode(tile, n, y0, x0);
        __syncthreads();
        if ((a.ablate & 2) && tile > t0) {
            // timing experiment: no staging after the first tile
        } else
        if (!have_pref) {
            rot = 0;
#pragma unroll
            for (int k = 0; k < U_IT; ++k) {
                const int p = spx + 32 * k;
                const Item it = u_load(n, y0, x0, p);
                if (KIND == 0 && bias_on) bias_add(bs, it);
                put(u_hi, u_lo, u_l8, p, scg, it, UGRAD);
            }
#pragma unroll
            for (int k = 0; k < V_IT; ++k) {
                const int p = spx + 32 * k;
                if (p < G::V_PIX) {
                    const Item it = v_load(n, y0, x0, p, scg);
                    if (KIND == 1 && bias_on && v_valid(y0, x0, p)) bias_add(bs, it);
                    put(v_hi, v_lo, v_l8, p, scg, it, !UGRAD);
                }
            }
        } else {
            if (ROLL) rot = (rot + TH) & (VH - 1);
#pragma unroll
            for (int k = 0; k < U_IT; ++k) {
                if (KIND == 0 && bias_on) bias_add(bs, pu[k]);
                put(u_hi, u_lo, u_l8, spx + 32 * k, scg, pu[k], UGRAD);
            }
#pragma unroll
            for (int k = 0; k < VN_IT; ++k) {
                const int pn = spx + 32 * k;
                if (pn < VN_PIX) {
                    const int rn = pn / VW, c = pn % VW;
                    const int slot = ROLL ? ((VN_ROW0 + rn + rot) & (VH - 1)) : rn;   // (the rolling window's VH is a power of two; the transposed conv's need not be)
                    if (KIND == 1 && bias_on && v_valid(y0, x0, pn)) bias_add(bs, pv[k]);
                    put(v_hi, v_lo, v_l8, slot * VW + c, scg, pv[k], !UGRAD);
                }
            }
        }
        __syncthreads();
        {
            have_pref = tile + 1 < t1 && (!ROLL || (tile + 1) % a.tiles_y != 0) && !(a.ablate & 2);
            if (have_pref) {
                int n2, y2, x2;
                decode(tile + 1, n2, y2, x2);
#pragma unroll
                for (int k = 0; k < U_IT; ++k) pu[k] = u_load(n2, y2, x2, spx + 32 * k);
#pragma unroll
                for (int k = 0; k < VN_IT; ++k) {
                    const int pn = spx + 32 * k;
                    if (pn < VN_PIX) pv[k] = v_load(n2, y2, x2, VN_ROW0 * VW + pn, scg);
                }
            }
        }
        if (!(a.ablate & 1)) {
            const int colA8 = wm * 32 + ((lane >> 4) & 1) * 16, colB8 = wn * 32 + ((lane >> 4) & 1) * 16;
            constexpr int SU8 = UGRAD ? WSU_F8_SCALE_G : WSU_F8_SCALE_X, SUL = UGRAD ? WSU_F8_SCALE_GLO : WSU_F8_SCALE_XLO;
            constexpr int SV8 = UGRAD ? WSU_F8_SCALE_X : WSU_F8_SCALE_G, SVL = UGRAD ? WSU_F8_SCALE_XLO : WSU_F8_SCALE_GLO;
            const int sc_a = hh ? SUL : SU8, sc_b = hh ? SV8 : SVL;
#pragma unroll 1
            for (int r = 0; r < TH; ++r) {
                u32x4 a8 = mk_u4(0, 0, 0, 0), al8 = a8;
                if constexpr (!HONLY) { a8 = f8_frag(u_lo, r * TW + 16 * hh, 1, colA8); al8 = f8_frag(u_l8, r * TW + 16 * hh, 1, colA8); }
                const u32x4 ah0 = x3_frag(u_hi, r * TW + 8 * hh, 1, colA), ah1 = x3_frag(u_hi, r * TW + 16 + 8 * hh, 1, colA);
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    int vrow, vstep;
                    if (KIND == 0) { vrow = ((r + t / 3 + rot) & (VH - 1)) * VW + t % 3; vstep = 1; }
                    else           { vrow = (2 * r + (t >> 1)) * VW + (t & 1); vstep = 2; }
                    if constexpr (!HONLY) {
                        const u32x4 bl8 = f8_frag(v_l8, vrow + vstep * 16 * hh, vstep, colB8), b8 = f8_frag(v_lo, vrow + vstep * 16 * hh, vstep, colB8);
                        wsu_mfma_f8x2(a8, al8, bl8, b8, sc_a, sc_b, acc[t]);
                    }
                    const u32x4 bh0 = x3_frag(v_hi, vrow + vstep * 8 * hh, vstep, colB), bh1 = x3_frag(v_hi, vrow + vstep * (16 + 8 * hh), vstep, colB);
                    wsu_mfma_f16(ah0, bh0, acc[t]);
                    wsu_mfma_f16(ah1, bh1, acc[t]);
                }
            }
        }
    }
    float* dst = a.part + ((size_t)((split * a.nmb + mb) * a.nnb + nb) * NTAPS) * 4096;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            dst[(size_t)t * 4096 + m * 64 + wn * 32 + l31] = acc[t][r];
        }
    if (bias_on) {                                                     // uniform per workgroup
        float* red = reinterpret_cast<float*>(smem);                   // [256 threads][8]: the 32 threads of a channel group meet here
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bs[e];
        __syncthreads();
        if (tid < 64) {
            const int cg = tid >> 3, e = tid & 7;
            float sum = 0.f;
            for (int j = 0; j < 32; ++j) sum += red[(cg * 32 + j) * 8 + e];
            if (KIND == 0) a.bpart[(size_t)split * (a.nmb * 64) + mb * 64 + tid] = sum;
            else           a.bpart[(size_t)split * (a.nnb * 64) + nb * 64 + tid] = sum;
        }
    }
}

// ---- conv 3x3 weight gradient with loader waves and LDS-DMA (the production kernel of train_mode 'f16f8p') ------------------------------------
// Ablation of wgrad_pl_kernel<0> (tools/time_wgrad.py): its time = matrix section + 70 % of the staging's HBM time -- one tile
// of register prefetch (1.35 us) is shorter than the loaded HBM latency, and a second register set does not fit beside 144 accumulators.
//   * workgroup = one (split, 64 co, 64 ci) slab like wgrad_pl_kernel, but 1024 threads, one per CU: twelve MATRIX waves = (32 co x 32 ci block,
//     tap row ky) with 3 accumulator tiles (kx) -- 48 instead of 144 registers, so 16 waves of 128 registers fit -- and four LOADER waves;
//   * steps walk DOWN an image column: a step brings one U tile (2 x 32 px) and the TWO new rows of the 4-row V window by LDS-DMA (32 pieces of
//     64 lanes x 16 B, 8 per loader wave: a loader waits with `s_waitcnt vmcnt(8)` -- the next step's pieces stay in flight) into the padded
//     [pixel][144 B] f16 / [pixel][96 B] residual images of wgrad_pl_kernel (lanes that would hit a pad unit idle, out-of-image U pixels fetch
//     zeros); U tiles live in three slots, V row pairs in four, so the DMA of step k+2 is issued behind barrier k; a column (or the workgroup's
//     range) starts with a prologue step that only contributes V rows;
//   * the e4m3-copy images are derived by the loader lane that fetched the f16 unit (its own 16 bytes: no cross-wave dependency); bias sums are
//     taken from the U images one step later (everything landed), 8 fixed channels per lane, reduced in lane order at the end;
//   * one raw s_barrier per step.
// (Two versions on the way, both slower than the register-staged kernel: converting the copies from the f16 fragments in the matrix waves --
// 18 conversions per V element, VALU-bound; producer waves staging through two register sets -- spills at the 128-register cap put vmcnt(0) waits
// behind every load group.)
namespace wgr {
constexpr int NMAT = 12, NLOAD = 4, NTD = (NMAT + NLOAD) * 64;
constexpr int U_PIX = 64, VW = 34, V_PIX = 2 * VW;                  // a V slot = one pair of window rows
// LDS images are split by channel half (the 32 channels of one matrix-wave block): [half][pixel][64 B] f16, [half][pixel][32 B] e4m3.  A
// half-wave's transposing read then covers 4 (8) consecutive rows x 64 (32) B = 256 contiguous bytes: every bank once, no padding -- the
// 144-byte pitch of wgrad_pl_kernel puts rows r and r+2 on 8 common banks (2 passes per ds_read_b64_tr_b16).
constexpr int HROW = 64, BROW = 32;
constexpr int U_HH = U_PIX * HROW, U_BH = U_PIX * BROW;              // 4096, 2048 per half
constexpr int V_HH = V_PIX * HROW, V_BH = V_PIX * BROW;              // 4352, 2176
constexpr int U_HI = 0, U_C8 = 2 * U_HH, U_L8 = U_C8 + 2 * U_BH, U_SLOT = U_L8 + 2 * U_BH;     // 16384
constexpr int V_HI = 0, V_C8 = 2 * V_HH, V_L8 = V_C8 + 2 * V_BH, V_SLOT = V_L8 + 2 * V_BH;     // 17408
constexpr int DEPTH = 3;                                             // the DMA of step k + DEPTH is issued behind barrier k
constexpr int NU = DEPTH + 1, NV = DEPTH + 2;
constexpr int V_BASE = NU * U_SLOT;
constexpr int LDS_TOTAL = V_BASE + NV * V_SLOT;                      // 152576
// DMA pieces of a step (64 lanes x 16 B): per half-image ceil(units / 64)
constexpr int PH_UH = U_HH / 1024, PH_UL = U_BH / 1024;              // 4, 2 per half
constexpr int PH_VH = (V_HH + 1023) / 1024, PH_VL = (V_BH + 1023) / 1024;   // 5, 3
constexpr int P_UH = 2 * PH_UH, P_UL = 2 * PH_UL, P_VH = 2 * PH_VH, P_VL = 2 * PH_VL;          // 8, 4, 10, 6
constexpr int NPIECE = P_UH + P_UL + P_VH + P_VL;                    // 28
constexpr int PER = NPIECE / NLOAD;                                  // 7
static_assert(NPIECE % NLOAD == 0 && U_HH % 1024 == 0 && U_BH % 1024 == 0, "piece bookkeeping");
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
}



typedef __attribute__((address_space(3))) void lds_void_w;
typedef __attribute__((address_space(1))) const void glb_void_w;

struct WgrWalk { int t; bool pro; int ty, tx, n; };
__device__ __forceinline__ WgrWalk wgr_walk_at(const WgPlArgs& a, int tile) {
    WgrWalk w; w.t = tile; w.pro = true;
    int tt = tile;
    w.ty = tt % a.tiles_y; tt /= a.tiles_y;
    w.tx = tt % a.tiles_x; w.n = tt / a.tiles_x;
    return w;
}
__device__ __forceinline__ void wgr_advance(const WgPlArgs& a, int t1, WgrWalk& w) {
    if (w.pro) { w.pro = false; return; }
    ++w.t;
    if (++w.ty == a.tiles_y) { w.ty = 0; if (++w.tx == a.tiles_x) { w.tx = 0; ++w.n; } }
    w.pro = w.t < t1 && w.ty == 0;
}

// piece p (0..27) of a step: image kind (0: U f16, 1: U residual, 2: V f16, 3: V residual), channel half, piece inside the half-image
__host__ __device__ constexpr int wgr_kind(int p) { return p < wgr::P_UH ? 0 : p < wgr::P_UH + wgr::P_UL ? 1 : p < wgr::P_UH + wgr::P_UL + wgr::P_VH ? 2 : 3; }
__host__ __device__ constexpr int wgr_pidx(int p) { return p < wgr::P_UH ? p : p < wgr::P_UH + wgr::P_UL ? p - wgr::P_UH : p < wgr::P_UH + wgr::P_UL + wgr::P_VH ? p - wgr::P_UH - wgr::P_UL : p - wgr::P_UH - wgr::P_UL - wgr::P_VH; }
__host__ __device__ constexpr int wgr_pph(int kd) { return kd == 0 ? wgr::PH_UH : kd == 1 ? wgr::PH_UL : kd == 2 ? wgr::PH_VH : wgr::PH_VL; }       // pieces per half-image
__host__ __device__ constexpr int wgr_half_bytes(int kd) { return kd == 0 ? wgr::U_HH : kd == 1 ? wgr::U_BH : kd == 2 ? wgr::V_HH : wgr::V_BH; }
__host__ __device__ constexpr int wgr_img_off(int kd) { return kd == 0 ? wgr::U_HI : kd == 1 ? wgr::U_L8 : kd == 2 ? wgr::V_HI : wgr::V_L8; }  // inside its slot
// LDS byte offset (inside the slot) of piece p
__host__ __device__ constexpr int wgr_piece_off(int p) {
    return wgr_img_off(wgr_kind(p)) + (wgr_pidx(p) / wgr_pph(wgr_kind(p))) * wgr_half_bytes(wgr_kind(p)) + (wgr_pidx(p) % wgr_pph(wgr_kind(p))) * 1024;
}

// fragment reads on the half-images: 8 (16) k-values = pixels row0.. of the 32 channels of half-image `img`
__device__ __forceinline__ u32x4 wgr_frag16(const char* img, int row0) {
    const int lane = threadIdx.x & 63, li = lane & 15, q = li >> 2, pp = li & 3;
    const char* a0 = img + (row0 + q) * wgr::HROW + ((lane >> 4) & 1) * 32 + pp * 8;
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * wgr::HROW));
    const u32x2 lo = __builtin_bit_cast(u32x2, r0), hi = __builtin_bit_cast(u32x2, r1);
    return mk_u4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ u32x4 wgr_frag8(const char* img, int row0) {
    const int lane = threadIdx.x & 63, li = lane & 15, q = li >> 1, pp = li & 1;
    const char* a0 = img + (row0 + q) * wgr::BROW + ((lane >> 4) & 1) * 16 + pp * 8;
    const i32x2_t r0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)a0);
    const i32x2_t r1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(a0 + 8 * wgr::BROW));
    return mk_u4((uint32_t)r0[0], (uint32_t)r0[1], (uint32_t)r1[0], (uint32_t)r1[1]);
}

typedef __attribute__((address_space(3))) char lds_char_w;
constexpr unsigned WGR_OOB = 0xFFFFFFF0u;                             // beyond every descriptor below: the hardware range check returns zeros

// Round 3: the loader's instruction stream.  Round 2 built a 64-bit source address per piece and step (tile origin, reflected column, row
// select, inside-the-image test and a select between the address and a block of zeros): ~260 vector instructions per step and wave, issued
// on the SIMD its three matrix waves need.  Now a piece is one `buffer_load_dwordx4 ... offen lds`:
//   * U pieces: descriptor = the (image, 64-channel block)'s 12 planes, scalar offset = the tile's origin, per-lane offset = the unit's
//     (chunk, plane, row, column) inside the tile -- computed ONCE per kernel; lanes beyond the tile's pixels carry an out-of-range offset
//     and read zeros.  Only tiles that cross the image's right / bottom edge take a per-step select (wave-uniform branch).
//   * V pieces: per-lane offset = unit (chunk, plane) + reflected column, recomputed when the walk enters a new tile column; the two rows of
//     the pair differ by a scalar (+- one image row): one multiply-add per piece and step.
// HONLY (products = WSU_PRODUCTS_F16): the residual pieces (kinds 1, 3) are not fetched, no e4m3 copies are derived; the LDS layout stays.
template <int LW, bool HONLY>
__device__ __forceinline__ void wgr_loader(const WgPlArgs& a, char* smem, int lane, int split, int mb, int nb, int t0, int t1) {
    using namespace wgr;
    constexpr auto fetched = [](int p) constexpr { return !HONLY || (wgr_kind(p) & 1) == 0; };
    constexpr int NDMA = [&]() constexpr { int c = 0; for (int k = 0; k < PER; ++k) c += fetched(LW + NLOAD * k) ? 1 : 0; return c; }();   // DMA instructions per step
    const int L = LW * 64 + lane;
    const char* vsrc; int cv, vch0;
    if (nb * 64 < a.cv1) { vsrc = a.v1; cv = a.cv1; vch0 = nb * 64; }
    else                 { vsrc = a.v2; cv = a.cv2; vch0 = nb * 64 - a.cv1; }
    const unsigned hw16 = (unsigned)(a.hu * a.wu) * 16u;
    const bool bias_on = a.bpart != nullptr && nb == 0;
    if (t0 >= t1) {                                                     // empty split: no barriers on either side; the partials it owns are zeros
        if (bias_on && L < 64) a.bpart[(size_t)split * (a.nmb * 64) + mb * 64 + L] = 0.f;
        return;
    }
    lds_char_w* smem3 = (lds_char_w*)smem;
    // per piece k (p = LW + 4 k), fixed for the kernel: the lane's unit -- alive or beyond the slot's pixels --, its (row, column) inside the
    // tile / row pair, its offset 16 * ((chunk * 3 + plane) * H * W [+ r * W + c for U]) and where its e4m3 copy goes
    unsigned uoff[PER];                                                 // U pieces: complete; V pieces: the (chunk, plane) part
    int rc[PER], c8off[PER];
    unsigned alive = 0;
    WSU_STATIC_FOR(PER, k, {
        constexpr int p = LW + NLOAD * k, kd = wgr_kind(p), upr = (kd & 1) ? 2 : 4, npx = kd < 2 ? U_PIX : V_PIX, roww = kd < 2 ? TW : VW;
        constexpr int half = wgr_pidx(p) / wgr_pph(kd), pih = wgr_pidx(p) % wgr_pph(kd);
        const int sidx = pih * 64 + lane;
        const int pxl = sidx / upr, u = sidx - pxl * upr;              // pixel, 16-byte unit inside the half-row
        if (pxl < npx && fetched(p)) alive |= 1u << k;
        const int g = half * upr + u;                                   // unit of the whole 64-channel row
        const int chunk = (kd & 1) ? g : g >> 1, plane = (kd & 1) ? 2 : (g & 1);        // f16: unit = 2 chunk + plane; residual: unit = chunk
        const int r = pxl / roww, c = pxl - r * roww;
        rc[k] = r * 256 + c;
        uoff[k] = (unsigned)(chunk * 3 + plane) * hw16 + (kd < 2 ? (unsigned)(r * a.wu + c) * 16u : 0u);
        if (!(pxl < npx)) { uoff[k] = WGR_OOB; rc[k] = 0; }             // a dead lane stays out of range whatever is added per step (its row index is 0)
        c8off[k] = half * (kd < 2 ? U_BH : V_BH) + pxl * BROW + u * 8; // the 8 copies of an f16 unit inside the copy image
    });
    unsigned vxoff[PER];                                                // V pieces: uoff + 16 * reflected column of the current tile column
    int vx_tx = -1;
    auto issue = [&](const WgrWalk& w, int k_step) __attribute__((always_inline)) {
        const int y0 = w.ty * 2, x0 = w.tx * TW;
        const char* ubase = a.u + ((size_t)w.n * (a.cu >> 4) + mb * 4) * 3 * hw16;                 // wave-uniform
        const char* vbase = vsrc + ((size_t)w.n * (cv >> 4) + (vch0 >> 4)) * 3 * hw16;
        const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ubase), 0, (int)(12u * hw16), 0x00020000);
        const auto rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, (int)(12u * hw16), 0x00020000);
        const int u_org = (y0 * a.wu + x0) * 16;                        // scalar offset of the tile's first pixel
        const int vy = w.pro ? y0 - 1 : y0 + 1;                         // first of the two padded rows this step brings
        // the two rows of the pair are one image row apart, in either order (reflection at the top / bottom edge): the scalar offset carries the
        // smaller one, the lanes of the other row add the (positive) distance -- a per-lane offset must never go below zero (it would wrap
        // around 2^32 and the range check would turn the access into zeros)
        const int vrA = wsu_reflect(vy, a.hu) * a.wu * 16, vrB = wsu_reflect(vy + 1, a.hu) * a.wu * 16;
        const int vrow0 = min(vrA, vrB), dvrow = abs(vrB - vrA), rsel = vrB > vrA ? 1 : 0;                 // lanes with row index == rsel add dvrow
        const int ulim_y = min(a.hu - y0, 2), ulim_x = min(a.wu - x0, TW);
        const bool u_full = ulim_y == 2 && ulim_x == TW;               // wave-uniform: every U pixel of the tile is inside the image
        if (vx_tx != w.tx) {                                            // a new tile column: the reflected columns of the V units
            vx_tx = w.tx;
            WSU_STATIC_FOR(PER, k, {
                constexpr int p = LW + NLOAD * k, kd = wgr_kind(p);
                if constexpr (kd >= 2) {
                    const int xx = wsu_reflect(x0 - 1 + (rc[k] & 255), a.wu);
                    vxoff[k] = (alive & (1u << k)) ? uoff[k] + (unsigned)xx * 16u : WGR_OOB;
                } else {
                    vxoff[k] = 0;
                }
            });
        }
        lds_char_w* us = smem3 + (k_step % NU) * U_SLOT;
        lds_char_w* vs = smem3 + V_BASE + (k_step % NV) * V_SLOT;
        WSU_STATIC_FOR(PER, k, {
            constexpr int p = LW + NLOAD * k, kd = wgr_kind(p);
            // every piece has live lanes (static layout): a wave issues exactly NDMA instructions per step -- the vmcnt arithmetic below
            if constexpr (!fetched(p)) {
            } else if constexpr (kd < 2) {                                     // (a prologue step fetches its tile's U too -- unused, it keeps the piece count constant)
                unsigned vo = uoff[k];
                if (!u_full) {
                    const bool inside = (rc[k] >> 8) < ulim_y && (rc[k] & 255) < ulim_x;
                    vo = inside ? vo : WGR_OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lds_void_w*)(us + wgr_piece_off(p)), 16, vo, u_org, 0, 0);
            } else {
                const unsigned v2 = vxoff[k] + ((rc[k] >> 8) == rsel ? (unsigned)dvrow : 0u);
                // the last piece of a V half-image is only partly alive (68 pixels): its dead lanes must not write -- their LDS bytes belong
                // to the next half-image
                constexpr int upr = (kd & 1) ? 2 : 4, pih = wgr_pidx(p) % wgr_pph(kd), live = V_PIX * upr - pih * 64;
                if constexpr (live < 64) {
                    if (lane < live) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_void_w*)(vs + wgr_piece_off(p)), 16, v2, vrow0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_void_w*)(vs + wgr_piece_off(p)), 16, v2, vrow0, 0, 0);
                }
            }
        });
    };
    // e4m3 copies of the f16 units this lane fetched for step k_step (its own 16 bytes, landed: vmcnt)
    auto derive = [&](int k_step) __attribute__((always_inline)) {
        char* us = smem + (k_step % NU) * U_SLOT;
        char* vs = smem + V_BASE + (k_step % NV) * V_SLOT;
        WSU_STATIC_FOR(PER, k, {
            constexpr int p = LW + NLOAD * k, kd = wgr_kind(p);
            if constexpr (!HONLY && (kd == 0 || kd == 2)) {
                if (alive & (1u << k)) {
                    char* img = kd == 0 ? us : vs;
                    const u32x4 hgr = *reinterpret_cast<const u32x4*>(img + wgr_piece_off(p) + lane * 16);
                    *reinterpret_cast<u32x2*>(img + (kd == 0 ? U_C8 : V_C8) + c8off[k]) = kd == 0 ? wsu_f16x8_to_fp8_grad(hgr) : wsu_f16x8_to_fp8(hgr);
                }
            }
        });
    };
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;
    auto bias = [&](int k_step) __attribute__((always_inline)) {        // lane = (8-channel group L >> 5, pixels (L & 31) + 32 k) of the landed U tile
        const char* us = smem + (k_step % NU) * U_SLOT;
        const int cg = L >> 5;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = (L & 31) + 32 * k;
            const u32x4 hq = *reinterpret_cast<const u32x4*>(us + U_HI + (cg >> 2) * U_HH + p * HROW + (cg & 3) * 16);
            if constexpr (HONLY) {                                          // the residual image was not fetched: sums of the f16 parts
                bs[0] = wsu_add_f16_lo(bs[0], hq.x); bs[1] = wsu_add_f16_hi(bs[1], hq.x); bs[2] = wsu_add_f16_lo(bs[2], hq.y); bs[3] = wsu_add_f16_hi(bs[3], hq.y);
                bs[4] = wsu_add_f16_lo(bs[4], hq.z); bs[5] = wsu_add_f16_hi(bs[5], hq.z); bs[6] = wsu_add_f16_lo(bs[6], hq.w); bs[7] = wsu_add_f16_hi(bs[7], hq.w);
                continue;
            }
            const u32x2 rr = *reinterpret_cast<const u32x2*>(us + U_L8 + (cg >> 2) * U_BH + p * BROW + (cg & 3) * 8);
            const int r0 = (int)rr.x, r1 = (int)rr.y;
            // sum += f16 part (v_fma_mix_f32 reads the half directly) + residual * 2^-14: three instructions per value
            bs[0] = fmaf(__builtin_amdgcn_cvt_f32_fp8(r0, 0), WSU_F8_GLO_DIV, wsu_add_f16_lo(bs[0], hq.x)); bs[1] = fmaf(__builtin_amdgcn_cvt_f32_fp8(r0, 1), WSU_F8_GLO_DIV, wsu_add_f16_hi(bs[1], hq.x));
            bs[2] = fmaf(__builtin_amdgcn_cvt_f32_fp8(r0, 2), WSU_F8_GLO_DIV, wsu_add_f16_lo(bs[2], hq.y)); bs[3] = fmaf(__builtin_amdgcn_cvt_f32_fp8(r0, 3), WSU_F8_GLO_DIV, wsu_add_f16_hi(bs[3], hq.y));
            bs[4] = fmaf(__builtin_amdgcn_cvt_f32_fp8(r1, 0), WSU_F8_GLO_DIV, wsu_add_f16_lo(bs[4], hq.z)); bs[5] = fmaf(__builtin_amdgcn_cvt_f32_fp8(r1, 1), WSU_F8_GLO_DIV, wsu_add_f16_hi(bs[5], hq.z));
            bs[6] = fmaf(__builtin_amdgcn_cvt_f32_fp8(r1, 2), WSU_F8_GLO_DIV, wsu_add_f16_lo(bs[6], hq.w)); bs[7] = fmaf(__builtin_amdgcn_cvt_f32_fp8(r1, 3), WSU_F8_GLO_DIV, wsu_add_f16_hi(bs[7], hq.w));
        }
    };
    // Every existing step k gets exactly one barrier k on both sides, plus two closing barriers: S + 2 barriers per wave (S >= 2).
    // DMA of step k + DEPTH goes out behind barrier k; `ahead` = steps issued beyond the one being waited for (PER pieces each stay in flight).
    auto wait_all_but = [&](int ahead) __attribute__((always_inline)) {
        if (ahead >= 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NDMA) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(DEPTH == 3 && 2 * NDMA < 64, "the vmcnt immediates above");
    WgrWalk wi = wgr_walk_at(a, t0);                                    // the walk position of the next DMA issue
    WgrWalk wb = wi;                                                    // the step whose barrier comes next (for the bias pass)
    int issued = 0;                                                     // steps issued so far
    for (; issued < DEPTH && wi.t < t1; ++issued) { issue(wi, issued); wgr_advance(a, t1, wi); }        // steps 0 .. DEPTH-1
    wait_all_but(issued - 1);                                           // step 0 landed (this wave's pieces)
    derive(0);
    for (int k = 0; ; ++k) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's derived copies of step k are written
        __builtin_amdgcn_s_barrier();                                   // barrier k: the matrix waves start step k; all of step k is visible
        asm volatile("" ::: "memory");
        if (wi.t < t1) { if (!(a.ablate & 2)) issue(wi, issued); wgr_advance(a, t1, wi); ++issued; }     // step k + DEPTH: its slots held steps k-1 (U) / k-2 (V)
        if (bias_on && !wb.pro) bias(k);
        wgr_advance(a, t1, wb);
        if (!(wb.t < t1)) break;                                        // no step k+1
        wait_all_but((a.ablate & 2) ? 0 : issued - (k + 2));             // step k+1 landed
        if (!(a.ablate & 4)) derive(k + 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // closing barrier 1: the matrix waves have left the last step
    asm volatile("" ::: "memory");
    if (bias_on) {                                                      // per-lane partials -> LDS (the stages are free now)
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[L * 8 + e] = bs[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // closing barrier 2 (all 16 waves): the partials are visible
    asm volatile("" ::: "memory");
    if (bias_on && L < 64) {                                            // one sum per channel over the 32 lanes of its group, in lane order
        const float* red = reinterpret_cast<const float*>(smem);
        const int cg = L >> 3, e = L & 7;
        float sum = 0.f;
        for (int j = 0; j < 32; ++j) sum += red[(cg * 32 + j) * 8 + e];
        a.bpart[(size_t)split * (a.nmb * 64) + mb * 64 + L] = sum;
    }
}


template <bool HONLY>
__global__ __launch_bounds__(wgr::NTD) void wgrad_ring_kernel(const WgPlArgs a) {
    using namespace wgr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    // the (mb, nb) workgroups of one split read the same U / V tiles at the same pace: consecutive logical ids share an XCD (one L2)
    int b = (int)wsu_xcd_remap(blockIdx.x, gridDim.x);
    const int nb = b % a.nnb; b /= a.nnb;
    const int mb = b % a.nmb;
    const int split = b / a.nmb;
    const int t0 = split * a.tiles_per_split;
    const int t1 = min(t0 + a.tiles_per_split, a.ntiles);
    // step sequence (the same state machine on both sides, WgrWalk): tile t0's prologue, t0, t0+1, ..., a prologue before every tile that starts a
    // column; tiles are numbered down the image columns (ty fastest, then tx, then the image)
    if (wv >= NMAT) {
        // ================= loader waves (wgr_loader<LW>: the piece kinds of a wave are compile-time) =======================================
        switch (wv - NMAT) {
            case 0: wgr_loader<0, HONLY>(a, smem, lane, split, mb, nb, t0, t1); break;
            case 1: wgr_loader<1, HONLY>(a, smem, lane, split, mb, nb, t0, t1); break;
            case 2: wgr_loader<2, HONLY>(a, smem, lane, split, mb, nb, t0, t1); break;
            default: wgr_loader<3, HONLY>(a, smem, lane, split, mb, nb, t0, t1); break;
        }
        return;
    }

    // ================= matrix waves ========================================================================================================
    const int blk = wv & 3, ky = wv >> 2, wm = blk >> 1, wn = blk & 1;
    const int sc_a = hh ? WSU_F8_SCALE_GLO : WSU_F8_SCALE_G, sc_b = hh ? WSU_F8_SCALE_X : WSU_F8_SCALE_XLO;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool empty = t0 >= t1;
    WgrWalk w = wgr_walk_at(a, t0);
    for (int k = 0; w.t < t1; ++k) {
        __builtin_amdgcn_s_barrier();                                   // step k is in LDS
        asm volatile("" ::: "memory");
        if (!w.pro && !(a.ablate & 1)) {
            const char* us = smem + (k % NU) * U_SLOT;
            const char* u_hi = us + U_HI + wm * U_HH; const char* u_c8 = us + U_C8 + wm * U_BH; const char* u_l8 = us + U_L8 + wm * U_BH;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const u32x4 ah0 = wgr_frag16(u_hi, r * TW + 8 * hh), ah1 = wgr_frag16(u_hi, r * TW + 16 + 8 * hh);
                u32x4 a8 = mk_u4(0, 0, 0, 0), al8 = a8;
                if constexpr (!HONLY) { a8 = wgr_frag8(u_c8, r * TW + 16 * hh); al8 = wgr_frag8(u_l8, r * TW + 16 * hh); }
                const int i = r + ky;                                   // window row 0..3: rows 0, 1 came with step k-1, rows 2, 3 with step k
                const char* vs = smem + V_BASE + ((i < 2 ? k + NV - 1 : k) % NV) * V_SLOT;
                const char* v_hi = vs + V_HI + wn * V_HH; const char* v_c8 = vs + V_C8 + wn * V_BH; const char* v_l8 = vs + V_L8 + wn * V_BH;
                const int vrow0 = (i & 1) * VW;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int vrow = vrow0 + kx;
                    if constexpr (!HONLY) {
                        const u32x4 bl8 = wgr_frag8(v_l8, vrow + 16 * hh), b8 = wgr_frag8(v_c8, vrow + 16 * hh);
                        wsu_mfma_f8x2(a8, al8, bl8, b8, sc_a, sc_b, acc[kx]);
                    }
                    const u32x4 bh0 = wgr_frag16(v_hi, vrow + 8 * hh), bh1 = wgr_frag16(v_hi, vrow + 16 + 8 * hh);
                    wsu_mfma_f16(ah0, bh0, acc[kx]);
                    wsu_mfma_f16(ah1, bh1, acc[kx]);
                }
            }
        }
        wgr_advance(a, t1, w);
    }
    if (!empty) {
        __builtin_amdgcn_s_barrier();                                   // the two closing barriers (the loaders publish their bias partials between them)
        __builtin_amdgcn_s_barrier();
    }
    float* dst = a.part + ((size_t)((split * a.nmb + mb) * a.nnb + nb) * 9) * 4096;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            dst[(size_t)(ky * 3 + kx) * 4096 + m * 64 + wn * 32 + l31] = acc[kx][r];
        }
}

// dW (conv: OIHW [M = co][Ntot = ci][3][3]; convT: IOHW [M = ci][Ntot = co][2][2]) = sum over splits, fixed order
template <int KIND>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart,
                                                           float* __restrict__ dw, float* __restrict__ db,
                                                           int nsplit, int nmb, int nnb, int nbias) {
    constexpr int NTAPS = Geo<KIND>::NTAPS;
    const int mtot = nmb * 64, ntot = nnb * 64;
    const long long total = (long long)mtot * ntot * NTAPS;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // thread index ordered like the partial slabs (n fastest) for coalesced reads
        long long t = i;
        const int nl = t % 64; t /= 64;
        const int ml = t % 64; t /= 64;
        const int tap = t % NTAPS; t /= NTAPS;
        const int nb = t % nnb; const int mb = (int)(t / nnb);
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp)
            s += part[((size_t)((sp * nmb + mb) * nnb + nb) * NTAPS + tap) * 4096 + ml * 64 + nl];
        const int m = mb * 64 + ml, n = nb * 64 + nl;
        dw[((size_t)m * ntot + n) * NTAPS + tap] = s;
    }
    if (db && bpart) {                                                 // nbias = channels of the gradient operand (M for the conv, N for the transposed conv)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nbias; i += gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) s += bpart[(size_t)sp * nbias + i];
            db[i] = s;
        }
    }
}

// plain per-channel sum over pixels (bias gradient of the transposed conv: db[co] = sum dy[..., co])
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long long npix, int c, int chunk) {
    // grid = (nchunks, c/64); block: 16 channel quads (16-byte loads) x 16 pixel lanes, 4 pixels in flight per lane
    const int q = threadIdx.x & 15, ps = threadIdx.x >> 4;
    const int ch0 = blockIdx.y * 64 + 4 * q;
    const long long p0 = (long long)blockIdx.x * chunk, p1 = min(p0 + chunk, npix);
    f32x4 s0 = mk_f4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    long long p = p0 + ps;
    for (; p + 48 < p1; p += 64) {
        s0 = s0 + *reinterpret_cast<const f32x4*>(x + p * c + ch0);
        s1 = s1 + *reinterpret_cast<const f32x4*>(x + (p + 16) * c + ch0);
        s2 = s2 + *reinterpret_cast<const f32x4*>(x + (p + 32) * c + ch0);
        s3 = s3 + *reinterpret_cast<const f32x4*>(x + (p + 48) * c + ch0);
    }
    for (; p < p1; p += 16) s0 = s0 + *reinterpret_cast<const f32x4*>(x + p * c + ch0);
    __shared__ float red[16][64];
    *reinterpret_cast<f32x4*>(&red[ps][4 * q]) = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        part[(size_t)blockIdx.x * c + blockIdx.y * 64 + threadIdx.x] = t;
    }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nchunks, int c) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float s = 0.f;
    for (int k = 0; k < nchunks; ++k) s += part[(size_t)k * c + ch];
    out[ch] = s;
}

template <int KIND>
int run_wgrad(WgArgs a, float* dw, float* db, float* workspace, size_t workspace_bytes, hipStream_t s, bool x3 = false, bool f8 = false) {
    using G = Geo<KIND>;
    a.tiles_x = (a.wu + TW - 1) / TW; a.tiles_y = (a.hu + G::TH - 1) / G::TH;
    a.ntiles = a.n * a.tiles_x * a.tiles_y;
    a.nmb = a.cu / 64; a.nnb = (a.cv1 + a.cv2) / 64;
    // enough workgroups to fill 256 CUs x 2, bounded by the tile count and the workspace
    int nsplit = (512 + a.nmb * a.nnb - 1) / (a.nmb * a.nnb);
    nsplit = max(1, min(nsplit, a.ntiles));
    const size_t slab = (size_t)a.nmb * a.nnb * G::NTAPS * 4096 * sizeof(float);
    const size_t bslab = (size_t)a.nmb * 64 * sizeof(float);
    while (nsplit > 1 && nsplit * (slab + bslab) > workspace_bytes) --nsplit;
    if (nsplit * (slab + bslab) > workspace_bytes) {
        wsu_set_error("wgrad: workspace of %zu bytes too small (need >= %zu)", workspace_bytes, slab + bslab);
        return WSU_ERR_ARG;
    }
    a.nsplit = nsplit;
    a.tiles_per_split = (a.ntiles + nsplit - 1) / nsplit;
    a.part = workspace;
    a.bpart = (db && (!x3 || KIND == 0)) ? workspace + (size_t)nsplit * slab / sizeof(float) : nullptr;
    if (x3) {
        static bool attr_x3 = false;
        if (!attr_x3) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x3_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, GeoX3<KIND>::LDS);
            if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(wgrad_x3): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
            attr_x3 = true;
        }
        if (f8) {
            static bool attr_f8 = false;
            if (!attr_f8) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x3_kernel<KIND, true>), hipFuncAttributeMaxDynamicSharedMemorySize, GeoX3<KIND>::LDS_F8);
                if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(wgrad_f8): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
                attr_f8 = true;
            }
            hipLaunchKernelGGL((wgrad_x3_kernel<KIND, true>), dim3(nsplit * a.nmb * a.nnb), dim3(NT), GeoX3<KIND>::LDS_F8, s, a);
        } else
        hipLaunchKernelGGL(wgrad_x3_kernel<KIND>, dim3(nsplit * a.nmb * a.nnb), dim3(NT), GeoX3<KIND>::LDS, s, a);
        int rc = wsu_check_launch("wgrad_x3_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(wgrad_reduce_kernel<KIND>, dim3(512), dim3(256), 0, s, a.part, (const float*)a.bpart, dw, a.bpart ? db : (float*)nullptr, nsplit, a.nmb, a.nnb, a.nmb * 64);
        return wsu_check_launch("wgrad_reduce_kernel");
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(wgrad): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        attr_done = true;
    }
    hipLaunchKernelGGL(wgrad_kernel<KIND>, dim3(nsplit * a.nmb * a.nnb), dim3(NT), G::LDS, s, a);
    int rc = wsu_check_launch("wgrad_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel<KIND>, dim3(512), dim3(256), 0, s, a.part, a.bpart, dw, db, nsplit, a.nmb, a.nnb, a.nmb * 64);
    return wsu_check_launch("wgrad_reduce_kernel");
}

constexpr int CT_TH_F16 = 3;                                           // tile rows of wgrad_pl_kernel<1, HONLY>

template <int KIND>
int run_wgrad_pl(WgPlArgs a, float* dw, float* db, float* workspace, size_t workspace_bytes, hipStream_t s) {
    using G = Geo<KIND>;
    const int th = (KIND == 1 && a.honly) ? CT_TH_F16 : G::TH;
    a.tiles_x = (a.wu + TW - 1) / TW; a.tiles_y = (a.hu + th - 1) / th;
    a.ntiles = a.n * a.tiles_x * a.tiles_y;
    a.nmb = a.cu / 64; a.nnb = (a.cv1 + a.cv2) / 64;
    int nsplit = (512 + a.nmb * a.nnb - 1) / (a.nmb * a.nnb);
    nsplit = max(1, min(nsplit, a.ntiles));
    const size_t slab = (size_t)a.nmb * a.nnb * G::NTAPS * 4096 * sizeof(float);
    const int nbias = (KIND == 0 ? a.nmb : a.nnb) * 64;                // channels of the gradient operand
    const size_t bslab = (size_t)nbias * sizeof(float);
    while (nsplit > 1 && nsplit * (slab + bslab) > workspace_bytes) --nsplit;
    if (nsplit * (slab + bslab) > workspace_bytes) {
        wsu_set_error("wgrad_pl: workspace of %zu bytes too small (need >= %zu)", workspace_bytes, slab + bslab);
        return WSU_ERR_ARG;
    }
    a.nsplit = nsplit;
    a.tiles_per_split = (a.ntiles + nsplit - 1) / nsplit;
    a.part = workspace;
    a.bpart = db ? workspace + (size_t)nsplit * slab / sizeof(float) : nullptr;
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("WSU_WGRAD_ABLATE"); ablate = e ? atoi(e) : 0; }
    a.ablate = ablate;
    static int impl = -1;                                              // WSU_WGRAD_IMPL=reg: the register-staged kernel (A/B runs)
    if (impl < 0) { const char* e = getenv("WSU_WGRAD_IMPL"); impl = (e && e[0] == 'r') ? 0 : 1; }
    if (KIND == 0 && impl == 1) {
        // one persistent-style workgroup per CU: the splits cover the tiles, fewer and longer than the register-staged kernel's
        static int ncu = 0;
        if (ncu == 0) {
            int dev = 0; hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { wsu_set_error("wgrad_ring: cannot query the device"); return WSU_ERR_HIP; }
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, wgr::LDS_TOTAL);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, wgr::LDS_TOTAL);
            if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(wgrad_ring): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
            ncu = prop.multiProcessorCount;
        }
        int ns = (ncu + a.nmb * a.nnb - 1) / (a.nmb * a.nnb);
        ns = max(1, min(ns, min(nsplit, a.ntiles)));
        a.nsplit = ns;
        a.tiles_per_split = (a.ntiles + ns - 1) / ns;
        a.bpart = db ? workspace + (size_t)ns * slab / sizeof(float) : nullptr;
        if (a.honly) hipLaunchKernelGGL(wgrad_ring_kernel<true>, dim3(ns * a.nmb * a.nnb), dim3(wgr::NTD), wgr::LDS_TOTAL, s, a);
        else hipLaunchKernelGGL(wgrad_ring_kernel<false>, dim3(ns * a.nmb * a.nnb), dim3(wgr::NTD), wgr::LDS_TOTAL, s, a);
        int rc = wsu_check_launch("wgrad_ring_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(wgrad_reduce_kernel<KIND>, dim3(512), dim3(256), 0, s, a.part, (const float*)a.bpart, dw, a.bpart ? db : (float*)nullptr, ns, a.nmb, a.nnb, nbias);
        return wsu_check_launch("wgrad_reduce_kernel");
    }
    constexpr int THP = KIND == 1 ? CT_TH_F16 : G::TH;                 // the f16-products instantiation: tile rows, LDS (no e4m3 images)
    using GP = GeoX3<KIND, THP>;
    constexpr int LDS_H = GP::U_BYTES + GP::V_BYTES;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pl_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, GeoX3<KIND>::LDS_F8);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pl_kernel<KIND, true, THP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_H);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(wgrad_pl): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        attr = true;
    }
    if (a.honly) hipLaunchKernelGGL((wgrad_pl_kernel<KIND, true, THP>), dim3(nsplit * a.nmb * a.nnb), dim3(NT), LDS_H, s, a);
    else hipLaunchKernelGGL(wgrad_pl_kernel<KIND>, dim3(nsplit * a.nmb * a.nnb), dim3(NT), GeoX3<KIND>::LDS_F8, s, a);
    int rc = wsu_check_launch("wgrad_pl_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel<KIND>, dim3(512), dim3(256), 0, s, a.part, (const float*)a.bpart, dw, a.bpart ? db : (float*)nullptr, nsplit, a.nmb, a.nnb, nbias);
    return wsu_check_launch("wgrad_reduce_kernel");
}

}  // namespace

extern "C" {

// K7p weight / bias gradients on PLANAR operands (layout and gradient encodings: wsu.h).  Conv: g (cout channels, gradient), x1 / x2 (the
// layer's saved planar input(s)), dw (cout, c1 + c2, 3, 3), db (cout) or NULL.  Transposed conv: x (cin, at h x w), dy (cout, gradient, at
// 2h x 2w), dw (cin, cout, 2, 2), db (cout) or NULL (sum of dy, taken while staging).  Channel counts multiples of 64.  Workspace:
// wsu_wgrad_workspace_bytes.  Deterministic.
int wsu_conv3x3_pl_bwd_weight(const void* g, const void* x1, const void* x2, float* dw, float* db, float* workspace, size_t workspace_bytes,
                              int n, int h, int w, int c1, int c2, int cout, int products, void* stream) {
    WSU_REQUIRE(g && x1 && dw && workspace, "conv3x3_pl_bwd_weight: null pointer");
    WSU_REQUIRE(products == WSU_PRODUCTS_F16F8 || products == WSU_PRODUCTS_F16, "conv3x3_pl_bwd_weight: products must be WSU_PRODUCTS_F16F8 or WSU_PRODUCTS_F16");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_pl_bwd_weight: bad shape");
    WSU_REQUIRE(c1 > 0 && c1 % 64 == 0 && c2 >= 0 && c2 % 64 == 0 && (c2 == 0) == (x2 == nullptr), "conv3x3_pl_bwd_weight: c1=%d c2=%d must be multiples of 64", c1, c2);
    WSU_REQUIRE(cout > 0 && cout % 64 == 0, "conv3x3_pl_bwd_weight: cout=%d must be a multiple of 64", cout);
    WSU_REQUIRE((long long)h * w * 192 < 0xFFFFFFF0LL, "conv3x3_pl_bwd_weight: h*w too large (the 12 planes of a 64-channel block must stay below 4 GiB)");
    WgPlArgs a{};
    a.u = (const char*)g; a.v1 = (const char*)x1; a.v2 = (const char*)x2; a.n = n; a.hu = h; a.wu = w; a.cu = cout; a.cv1 = c1; a.cv2 = c2;
    a.honly = products == WSU_PRODUCTS_F16 ? 1 : 0;
    return run_wgrad_pl<0>(a, dw, db, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int wsu_convt2x2_pl_bwd_weight(const void* x, const void* dy, float* dw, float* db, float* workspace, size_t workspace_bytes,
                               int n, int h, int w, int cin, int cout, int products, void* stream) {
    WSU_REQUIRE(x && dy && dw && workspace, "convt2x2_pl_bwd_weight: null pointer");
    WSU_REQUIRE(products == WSU_PRODUCTS_F16F8 || products == WSU_PRODUCTS_F16, "convt2x2_pl_bwd_weight: products must be WSU_PRODUCTS_F16F8 or WSU_PRODUCTS_F16");
    WSU_REQUIRE(n > 0 && h > 0 && w > 0, "convt2x2_pl_bwd_weight: bad shape");
    WSU_REQUIRE(cin > 0 && cin % 64 == 0 && cout > 0 && cout % 64 == 0, "convt2x2_pl_bwd_weight: cin=%d cout=%d must be multiples of 64", cin, cout);
    WgPlArgs a{};
    a.u = (const char*)x; a.v1 = (const char*)dy; a.v2 = nullptr; a.n = n; a.hu = h; a.wu = w; a.cu = cin; a.cv1 = cout; a.cv2 = 0;
    a.honly = products == WSU_PRODUCTS_F16 ? 1 : 0;
    return run_wgrad_pl<1>(a, dw, db, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

size_t wsu_wgrad_workspace_bytes(int cm, int cn, int ntaps) {
    // room for up to 512 partial slabs' worth of workgroups: ceil(512 / blocks) splits, each (cm/64)(cn/64) slabs
    if (cm <= 0 || cn <= 0 || ntaps <= 0) return 0;
    const size_t nmb = cm / 64, nnb = cn / 64;
    size_t nsplit = (512 + nmb * nnb - 1) / (nmb * nnb);
    return nsplit * (nmb * nnb * ntaps * 4096 + (nmb > nnb ? nmb : nnb) * 64) * sizeof(float);   // slabs + bias partials of either operand
}

static int colsum_channels(const float* x, float* out, float* workspace, size_t workspace_bytes, long long npix, int c, hipStream_t s) {
    // ~1024 pixel chunks whatever the batch: the final kernel walks them serially per channel
    long long chunk_ll = (npix + 1023) / 1024;
    const int chunk = (int)(chunk_ll < 4096 ? 4096 : chunk_ll);
    const int nchunks = (int)((npix + chunk - 1) / chunk);
    WSU_REQUIRE((size_t)nchunks * c * sizeof(float) <= workspace_bytes, "bias reduction: workspace too small");
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nchunks, c / 64), dim3(256), 0, s, x, workspace, npix, c, chunk);
    int rc = wsu_check_launch("colsum_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((c + 63) / 64), dim3(64), 0, s, workspace, out, nchunks, c);
    return wsu_check_launch("colsum_final_kernel");
}

int wsu_conv3x3_bwd_weight(const float* g, const float* x1, const float* x2, float* dw, float* db,
                           float* workspace, size_t workspace_bytes,
                           int n, int h, int w, int c1, int c2, int cout, int mode, void* stream) {
    WSU_REQUIRE(mode == WSU_MODE_F32 || mode == WSU_MODE_BF16X3 || mode == WSU_MODE_F16F8X, "conv3x3_bwd_weight: mode must be f32, bf16x3 or f16f8x");
    WSU_REQUIRE(g && x1 && dw && workspace, "conv3x3_bwd_weight: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_bwd_weight: bad shape");
    WSU_REQUIRE(c1 > 0 && c1 % 64 == 0 && c2 >= 0 && c2 % 64 == 0 && (c2 == 0) == (x2 == nullptr), "conv3x3_bwd_weight: c1=%d c2=%d must be multiples of 64", c1, c2);
    WSU_REQUIRE(cout > 0 && cout % 64 == 0, "conv3x3_bwd_weight: cout=%d must be a multiple of 64", cout);
    WgArgs a{};
    a.u = g; a.v1 = x1; a.v2 = x2; a.n = n; a.hu = h; a.wu = w; a.cu = cout; a.cv1 = c1; a.cv2 = c2;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == WSU_MODE_F32) return run_wgrad<0>(a, dw, db, workspace, workspace_bytes, s);
    return run_wgrad<0>(a, dw, db, workspace, workspace_bytes, s, true, mode == WSU_MODE_F16F8X);      // exact fp32 bias gradient from the staging loop
}

int wsu_convt2x2_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                            float* workspace, size_t workspace_bytes,
                            int n, int h, int w, int cin, int cout, int mode, void* stream) {
    WSU_REQUIRE(mode == WSU_MODE_F32 || mode == WSU_MODE_BF16X3 || mode == WSU_MODE_F16F8X, "convt2x2_bwd_weight: mode must be f32, bf16x3 or f16f8x");
    WSU_REQUIRE(x && dy && dw && workspace, "convt2x2_bwd_weight: null pointer");
    WSU_REQUIRE(n > 0 && h > 0 && w > 0, "convt2x2_bwd_weight: bad shape");
    WSU_REQUIRE(cin > 0 && cin % 64 == 0 && cout > 0 && cout % 64 == 0, "convt2x2_bwd_weight: cin=%d cout=%d must be multiples of 64", cin, cout);
    WgArgs a{};
    a.u = x; a.v1 = dy; a.v2 = nullptr; a.n = n; a.hu = h; a.wu = w; a.cu = cin; a.cv1 = cout; a.cv2 = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = run_wgrad<1>(a, dw, nullptr, workspace, workspace_bytes, s, mode != WSU_MODE_F32, mode == WSU_MODE_F16F8X);
    if (rc || !db) return rc;
    return colsum_channels(dy, db, workspace, workspace_bytes, (long long)n * h * w * 4, cout, s);   // db[co] = sum of dy
}

}  // extern "C"
