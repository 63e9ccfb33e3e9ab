// Shared device-side helpers for the gfx950 kernels of libwsu.
#pragma once
#ifndef WSU_PROBE
#define WSU_PROBE 0             // timing-only build variants of the f16f8 matrix section (make probes; results are wrong when != 0)
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/wsu.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// Native vector types for all register staging: HIP's uint4/float4 are structs whose copies lower to
// memcpy between address spaces and pin the staging arrays in scratch.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__host__ __device__ __forceinline__ u32x4 mk_u4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { u32x4 v = {a, b, c, d}; return v; }
__host__ __device__ __forceinline__ u32x2 mk_u2(uint32_t a, uint32_t b) { u32x2 v = {a, b}; return v; }
__host__ __device__ __forceinline__ f32x4 mk_f4(float a, float b, float c, float d) { f32x4 v = {a, b, c, d}; return v; }

// ---- host-side error plumbing (thread-local message, never throws) -------------------------------
void wsu_set_error(const char* fmt, ...);
int wsu_check_launch(const char* what);
#define WSU_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            wsu_set_error(__VA_ARGS__);   \
            return WSU_ERR_ARG;           \
        }                                 \
    } while (0)

// ---- tiling constants shared by packers and kernels ----------------------------------------------
// A "chunk" is 64 bytes of channel data per pixel, staged in LDS as 4 granule planes of 16 bytes:
//   F32    : 16 channels, granule g = channels 4g..4g+3 (fp32)
//   BF16   : 32 channels, granule g = channels 8g..8g+7 (bf16)
//   BF16X3 : 16 channels, granules 0,1 = bf16 hi of channels 0..7 / 8..15, granules 2,3 = bf16 lo
//   F16F8  : 16 channels, granules 0,1 = f16 of channels 0..7 / 8..15, granule 2 = e4m3 residuals, granule 3 = e4m3 copies (below).
//            In HBM a chunk is 48 bytes (granules 0-2): granule 3 is derived from granules 0,1 while staging.
#define WSU_GRAN 4
#define WSU_COB 64            // output channels per workgroup
__host__ __device__ inline int wsu_chunk_channels(int mode) { return mode == WSU_MODE_BF16 ? 32 : 16; }

// PyTorch 'reflect' for pad 1 followed by a clamp (the clamp only matters for out-of-image tile lanes
// whose results are never stored): -1 -> 1, n -> n-2.
__device__ __forceinline__ int wsu_reflect(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ uint32_t wsu_pack_bf16x2(float a, float b) {
    __bf16 x = (__bf16)a, y = (__bf16)b;     // v_cvt_pk_bf16_f32, round-to-nearest-even
    return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}
__device__ __forceinline__ float wsu_bf16_lo_residual(float a) { return a - (float)(__bf16)a; }

// 8 fp32 -> 8 bf16 (hi) and 8 bf16 (lo = bf16_rne(x - hi)), each as a 16-byte granule.  hi is rounded to nearest by integer
// arithmetic on the fp32 word (add 0x8000, keep the upper half: ties go away from zero instead of to even, nothing else differs
// from v_cvt_pk_bf16_f32) and two heads are packed by one v_perm: ~4 VALU instructions per value instead of the ~9 the
// convert / widen / canonicalise sequence of `(float)(__bf16)x` compiled to (profiles/r01/conv3x3_ablation.md, instruction mix).
// Inf / NaN inputs: the add can carry into the exponent of the largest finite values (-> inf), like any round-to-nearest does.
__device__ __forceinline__ void wsu_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const uint32_t ra = __builtin_bit_cast(uint32_t, a) + 0x8000u, rb = __builtin_bit_cast(uint32_t, b) + 0x8000u;
    hi = __builtin_amdgcn_perm(rb, ra, 0x07060302u);                      // (ra >> 16) | (rb & 0xFFFF0000)
    lo = wsu_pack_bf16x2(a - __builtin_bit_cast(float, ra & 0xFFFF0000u), b - __builtin_bit_cast(float, rb & 0xFFFF0000u));
}
__device__ __forceinline__ void wsu_split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    wsu_split2(a.x, a.y, h0, l0); wsu_split2(a.z, a.w, h1, l1);
    wsu_split2(b.x, b.y, h2, l2); wsu_split2(b.z, b.w, h3, l3);
    hi = mk_u4(h0, h1, h2, h3); lo = mk_u4(l0, l1, l2, l3);
}

// ---- mode F16F8 (include/wsu.h): x = f16(x) + residual; the residual and an e4m3 copy of x feed the block-scaled fp8 MFMA ------
// Fixed power-of-two scales (E8M0 bytes handed to v_mfma_scale_*: 127 + log2 of the factor that undoes the storage scaling):
//   activations  x_lo8 = e4m3((x - f16 x) * 2^12) -> 115      x8 = e4m3(x * 2^-2)  -> 129
//   weights      w8    = e4m3(w * 2^6)            -> 121      w_lo8 = e4m3((w - f16 w) * 2^18) -> 109
// A residual is at most 2^-12 of its value, so the four encodings are in e4m3's normal range (2^-6 .. 448, 4 significant bits) for
// |x| in ~[2^-6, 448] and |w| in ~[2^-12, 7]; smaller values keep an absolute error below 2^-22 (x) / 2^-28 (w), larger ones saturate
// their e4m3 encodings (the product then has plain f16 accuracy).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
// divisors handed to v_cvt_scalef32_pk_fp8_f32 (it converts src / scale; tools/split_probe.hip)
#define WSU_F8_XLO_DIV 0x1p-12f
#define WSU_F8_X_DIV 4.f
#define WSU_F8_W_DIV 0x1p-6f
#define WSU_F8_WLO_DIV 0x1p-18f
#define WSU_F8_SCALE_XLO 115
#define WSU_F8_SCALE_X 129
#define WSU_F8_SCALE_W 121
#define WSU_F8_SCALE_WLO 109
// gradients (pre-scaled by a power of two so that max |dL/dout| lies in (2, 4], model/autograd.py):
//   g_lo8 = e4m3((g - f16 g) * 2^14) -> 113      g8 = e4m3(g * 4) -> 125
#define WSU_F8_GLO_DIV 0x1p-14f
#define WSU_F8_G_DIV 0.25f
#define WSU_F8_SCALE_GLO 113
#define WSU_F8_SCALE_G 125
#define WSU_F8_RANGE 448.f             // |x| beyond this: the e4m3 residual (x - f16 x) * 2^12 saturates -> plain f16 accuracy for that value
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;
// v - float(f16 half of `hpair`) in ONE instruction: v_fma_mix_f32 reads the f16 half directly (float(h) * -1 + v, exact: float(h) is exact and
// the subtraction rounds once like v_sub_f32 -- bitwise the two-instruction form v_cvt_f32_f16 + v_sub_f32 that hipcc emits for the C
// expression; it does not form the mixed instruction itself).  One VALU instruction less per stored value in every encoding epilogue.
__device__ __forceinline__ float wsu_sub_f16_lo(float v, uint32_t hpair) {
    float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(v)); return r;
}
__device__ __forceinline__ float wsu_sub_f16_hi(float v, uint32_t hpair) {
    float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(v)); return r;
}
// a * b + float(f16 half of `hpair`) in one instruction (decoding a stored value: residual * 2^-k + f16 part; bitwise fmaf(a, b, float(h)))
__device__ __forceinline__ float wsu_fma_f16_lo(float a, float b, uint32_t hpair) {
    float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(hpair)); return r;
}
__device__ __forceinline__ float wsu_fma_f16_hi(float a, float b, uint32_t hpair) {
    float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(hpair)); return r;
}
// 8 stored f16 values (one 16-byte granule) -> 8 bits "value > 0" (bit e = channel e of the granule): the ReLU-mask byte of the planar
// training path (include/wsu.h, relu_mask planes).  f16 > 0 <=> its 16 bits > 0 as a signed integer (-0, NaN with the sign bit: not positive).
__device__ __forceinline__ unsigned wsu_f16x8_pos_bits(const u32x4& h) {
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int wd = (int)h[e];
        bits |= ((short)(wd & 0xFFFF) > 0 ? 1u : 0u) << (2 * e);
        bits |= (wd >= 0x10000 ? 1u : 0u) << (2 * e + 1);
    }
    return bits;
}
// geometry of a ReLU-mask plane: rows padded to 32 pixels, row count to 16 (a tile of the persistent conv is 16 x 32: its mask rows are whole,
// aligned 32-byte runs that the data gradient's loaders bring in by LDS-DMA)
__host__ __device__ inline int wsu_mask_wp(int w) { return (w + 31) & ~31; }
__host__ __device__ inline int wsu_mask_hp(int h) { return (h + 15) & ~15; }
// acc + float(f16 half) in one instruction (running sums over stored values)
__device__ __forceinline__ float wsu_add_f16_lo(float acc, uint32_t hpair) {
    float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(acc)); return r;
}
__device__ __forceinline__ float wsu_add_f16_hi(float acc, uint32_t hpair) {
    float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(acc)); return r;
}
// 4 values -> 2 dwords of f16 (round to nearest even), 1 dword of residuals, 1 dword of e4m3 copies: 5.5 VALU instructions per value
// (v_cvt_pk_f16_f32, widen + subtract, two v_med3 -- the fp8 conversions overflow to NaN instead of saturating, also with
// MODE.FP16_OVFL set (tools/split_probe.hip) -- and the scaling conversions).  |v| > 65504 overflows the f16 part like any f16 pipeline.
__device__ __forceinline__ void wsu_split4_f16f8(const f32x4& v, float div_lo, float div_x, uint32_t& h01, uint32_t& h23, uint32_t& lo, uint32_t& x8) {
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
    h01 = __builtin_bit_cast(uint32_t, ha); h23 = __builtin_bit_cast(uint32_t, hb);
    const float lim_lo = 448.f * div_lo, lim_x = 448.f * div_x;
    const float r0 = __builtin_amdgcn_fmed3f(wsu_sub_f16_lo(v[0], h01), -lim_lo, lim_lo), r1 = __builtin_amdgcn_fmed3f(wsu_sub_f16_hi(v[1], h01), -lim_lo, lim_lo);
    const float r2 = __builtin_amdgcn_fmed3f(wsu_sub_f16_lo(v[2], h23), -lim_lo, lim_lo), r3 = __builtin_amdgcn_fmed3f(wsu_sub_f16_hi(v[3], h23), -lim_lo, lim_lo);
    i16x2 l = {0, 0}, x = {0, 0};
    l = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l, r0, r1, div_lo, false);
    l = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l, r2, r3, div_lo, true);
    x = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(x, __builtin_amdgcn_fmed3f(v[0], -lim_x, lim_x), __builtin_amdgcn_fmed3f(v[1], -lim_x, lim_x), div_x, false);
    x = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(x, __builtin_amdgcn_fmed3f(v[2], -lim_x, lim_x), __builtin_amdgcn_fmed3f(v[3], -lim_x, lim_x), div_x, true);
    lo = __builtin_bit_cast(uint32_t, l); x8 = __builtin_bit_cast(uint32_t, x);
}
// the same without the e4m3 copy (planar storage keeps f16 + residual; the copy is derived from the f16 part by the consumer's loader waves)
__device__ __forceinline__ void wsu_split4_f16r8(const f32x4& v, float div_lo, uint32_t& h01, uint32_t& h23, uint32_t& lo) {
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
    h01 = __builtin_bit_cast(uint32_t, ha); h23 = __builtin_bit_cast(uint32_t, hb);
    const float lim_lo = 448.f * div_lo;
    const float r0 = __builtin_amdgcn_fmed3f(wsu_sub_f16_lo(v[0], h01), -lim_lo, lim_lo), r1 = __builtin_amdgcn_fmed3f(wsu_sub_f16_hi(v[1], h01), -lim_lo, lim_lo);
    const float r2 = __builtin_amdgcn_fmed3f(wsu_sub_f16_lo(v[2], h23), -lim_lo, lim_lo), r3 = __builtin_amdgcn_fmed3f(wsu_sub_f16_hi(v[3], h23), -lim_lo, lim_lo);
    i16x2 l = {0, 0};
    l = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l, r0, r1, div_lo, false);
    l = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l, r2, r3, div_lo, true);
    lo = __builtin_bit_cast(uint32_t, l);
}
// 16 channels of one pixel -> the 3 x 16 B of a stored F16F8 chunk (the e4m3 copy is derived from the f16 part while staging)
__device__ __forceinline__ void wsu_split16_f16f8(const f32x4& q0, const f32x4& q1, const f32x4& q2, const f32x4& q3,
                                                  u32x4& hi0, u32x4& hi1, u32x4& lo8) {
    uint32_t a0, a1, a2, a3, a4, a5, a6, a7, l0, l1, l2, l3, x0, x1, x2, x3;
    wsu_split4_f16f8(q0, WSU_F8_XLO_DIV, WSU_F8_X_DIV, a0, a1, l0, x0); wsu_split4_f16f8(q1, WSU_F8_XLO_DIV, WSU_F8_X_DIV, a2, a3, l1, x1);
    wsu_split4_f16f8(q2, WSU_F8_XLO_DIV, WSU_F8_X_DIV, a4, a5, l2, x2); wsu_split4_f16f8(q3, WSU_F8_XLO_DIV, WSU_F8_X_DIV, a6, a7, l3, x3);
    hi0 = mk_u4(a0, a1, a2, a3); hi1 = mk_u4(a4, a5, a6, a7); lo8 = mk_u4(l0, l1, l2, l3);
}
// 8 stored f16 values -> their 8 e4m3 copies e4m3(x / 4) (LDS plane 3 of a chunk): clamp to the e4m3 range on the packed f16 pipe
// (the conversion overflows to NaN), then v_cvt_scalef32_pk_fp8_f16: 3 instructions per pair of values.
__device__ __forceinline__ u32x2 wsu_f16x8_to_fp8(const u32x4& h) {
    const _Float16 m = (_Float16)1792.f;
    const f16x8 lim = {m, m, m, m, m, m, m, m};
    const f16x8 c = __builtin_elementwise_max(__builtin_elementwise_min(__builtin_bit_cast(f16x8, h), lim), -lim);
    i16x2 lo = {0, 0}, hi = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, __builtin_shufflevector(c, c, 0, 1), WSU_F8_X_DIV, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, __builtin_shufflevector(c, c, 2, 3), WSU_F8_X_DIV, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, __builtin_shufflevector(c, c, 4, 5), WSU_F8_X_DIV, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, __builtin_shufflevector(c, c, 6, 7), WSU_F8_X_DIV, true);
    return mk_u2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
}
// ---- block-scaled fp4 (e2m1) cross terms (round 3; since round 4 on planar Q storage, below; semantics probed on the device: tools/fp4_probe.hip) ----
// A block = the 16 channels of one (pixel, chunk) resp. one (output channel, tap, chunk); its E8M0 scale is 2^E, the smallest power of two
// with (largest |f16 part|) / 2^E <= 6 = fp4's largest value (wsu_q4_block_exp: the largest element lands in [2, 3) or [4, 6]); both halves of a block
// share it -- the copy c = f16 part and the residual pre-scaled by 2^11 (|residual| <= 2^-11 |value|).  v_cvt_scalef32_pk_fp4_f16 / _f32
// return fp4(x / scale), round to nearest even, saturating, low nibble first; the MFMA multiplies a lane's 32 nibbles by 2^(scale byte - 127).
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
// block exponent E of a block whose largest |f16| has the bit pattern b: exponent field - 15 - 1 (largest / 2^E in [2, 4)), and one binade finer when
// that element's mantissa is <= 1.5 (largest / 2^E in [4, 6], 6 = fp4's largest value): the smallest power-of-two scale that does not saturate --
// every other element of the block gets a grid twice as fine in ~half of the blocks (MAE of the whole net 2.54e-5 -> 2.08e-5, profiles/r03/f16f4p.md)
__device__ __forceinline__ int wsu_q4_block_exp(uint32_t b) {
    const int ef = (int)(b >> 10);
    return ef - 16 - ((ef > 0 && (b & 0x3FFu) <= 0x200u) ? 1 : 0);
}
__device__ __forceinline__ float wsu_pow2f(int e) { return __builtin_bit_cast(float, (uint32_t)(e + 127) << 23); }
// largest |f16| bit pattern of the 16 values of two granules
__device__ __forceinline__ uint32_t wsu_f16x16_max_abs_bits(const u32x4& h0, const u32x4& h1) {
    u16x2_t m = __builtin_bit_cast(u16x2_t, h0.x & 0x7FFF7FFFu);
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, h0.y & 0x7FFF7FFFu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, h0.z & 0x7FFF7FFFu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, h0.w & 0x7FFF7FFFu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, h1.x & 0x7FFF7FFFu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, h1.y & 0x7FFF7FFFu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, h1.z & 0x7FFF7FFFu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, h1.w & 0x7FFF7FFFu));
    return m.x > m.y ? m.x : m.y;
}
// 8 f16 values (one granule) -> 8 fp4 nibbles (one dword), value / scale.  Inline assembly: with the builtin, hipcc (ROCm 7.2) converted the
// FIRST dword of the granule four times (`v_cvt_scalef32_pk_fp4_f16 v14, v2, v5` with every op_sel) -- the element index of the bit-cast source was lost.
__device__ __forceinline__ uint32_t wsu_f16x8_to_fp4(const u32x4& h, float scale) {
    uint32_t d = 0;
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2" : "+v"(d) : "v"(h.x), "v"(scale));
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2 op_sel:[0,0,1,0]" : "+v"(d) : "v"(h.y), "v"(scale));
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2 op_sel:[0,0,0,1]" : "+v"(d) : "v"(h.z), "v"(scale));
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2 op_sel:[0,0,1,1]" : "+v"(d) : "v"(h.w), "v"(scale));
    return d;
}
// ---- planar Q storage ("F16F4P" tensors, round 4; layout: include/wsu.h): the PRODUCER emits what the fp4 conv multiplies -----------------------
// Per (image, 16-channel chunk): planes f16 ch 0-7 | f16 ch 8-15 | Q (the 16-byte granule above) as [H][W][16 B], then the E8M0 scale bytes as
// 16 x 32-pixel tile blocks [ceil(H/16)][ceil(W/32)][16][32] (a conv tile's 512 scale bytes are one contiguous run; its halo is gathered per lane).
__host__ __device__ inline size_t wsu_q_sblocks(int h, int w) { return (size_t)((h + 15) >> 4) * (size_t)((w + 31) >> 5); }
__host__ __device__ inline size_t wsu_q_chunk_bytes(int h, int w) { return (size_t)48 * h * w + 512 * wsu_q_sblocks(h, w); }
// byte offset of pixel (y, x)'s scale byte inside the scale plane of a chunk
__device__ __forceinline__ unsigned wsu_q_soff(int y, int x, int tiles_x) {
    return (unsigned)(((y >> 4) * tiles_x + (x >> 5)) * 512 + (y & 15) * 32 + (x & 31));
}
__device__ __forceinline__ void wsu_swap32(uint32_t& upper_of, uint32_t& lower_of) {     // lanes 32-63 of `upper_of` <-> lanes 0-31 of `lower_of`
    const auto r = __builtin_amdgcn_permlane32_swap(upper_of, lower_of, false, false);
    upper_of = r[0]; lower_of = r[1];
}
// Accumulator-layout producer (the MFMA epilogues): this lane holds X = channels 4 hh + 0..3 and Y = channels 8 + 4 hh + 0..3 of one pixel's
// 16-channel chunk (hh = lane >> 5), lane ^ 32 the other eight.  Out: g = the f16 granule this lane stores to plane hh (lanes 0-31: ch 0-7,
// lanes 32-63: ch 8-15, as the e4m3 format's epilogues arrange it); dhi / dres = this lane's eight fp4 nibbles of the f16 parts / of the
// residuals (x - f16 x) * 2^11, both as [X: 16 bits | Y: 16 bits] and both divided by the block scale 2^E; sb = E + 127 (equal in both lanes).
// The residual nibbles come from the exact fp32 residual (the loaders of round 3 re-rounded the stored e4m3 residual).
__device__ __forceinline__ void wsu_q4_pre(const f32x4& X, const f32x4& Y, u32x4& g, uint32_t& dhi, uint32_t& dres, uint32_t& sb) {
    const f32x2 xa = {X[0], X[1]}, xb = {X[2], X[3]}, ya = {Y[0], Y[1]}, yb = {Y[2], Y[3]};
    uint32_t xh0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(xa, f16x2)), xh1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(xb, f16x2));
    uint32_t yh0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ya, f16x2)), yh1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(yb, f16x2));
    const float r0 = wsu_sub_f16_lo(X[0], xh0), r1 = wsu_sub_f16_hi(X[1], xh0), r2 = wsu_sub_f16_lo(X[2], xh1), r3 = wsu_sub_f16_hi(X[3], xh1);
    const float r4 = wsu_sub_f16_lo(Y[0], yh0), r5 = wsu_sub_f16_hi(Y[1], yh0), r6 = wsu_sub_f16_lo(Y[2], yh1), r7 = wsu_sub_f16_hi(Y[3], yh1);
    // largest |f16| of the block: eight values here, eight in the partner lane
    u16x2_t m = __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, xh0 & 0x7FFF7FFFu), __builtin_bit_cast(u16x2_t, xh1 & 0x7FFF7FFFu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, yh0 & 0x7FFF7FFFu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, yh1 & 0x7FFF7FFFu));
    uint32_t mw = __builtin_bit_cast(uint32_t, m), mp = mw;
    wsu_swap32(mw, mp);                                                   // one of (mw, mp) is now the partner's, the other this lane's
    m = __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, mw), __builtin_bit_cast(u16x2_t, mp));
    const int e = wsu_q4_block_exp(m.x > m.y ? m.x : m.y);
    const float sc = wsu_pow2f(e), scr = wsu_pow2f(e - 11);
    sb = (uint32_t)(e + 127);
    uint32_t dh = 0, dr = 0;
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2" : "+v"(dh) : "v"(xh0), "v"(sc));
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2 op_sel:[0,0,1,0]" : "+v"(dh) : "v"(xh1), "v"(sc));
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2 op_sel:[0,0,0,1]" : "+v"(dh) : "v"(yh0), "v"(sc));
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2 op_sel:[0,0,1,1]" : "+v"(dh) : "v"(yh1), "v"(sc));
    dr = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dr, r0, r1, scr, 0);
    dr = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dr, r2, r3, scr, 1);
    dr = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dr, r4, r5, scr, 2);
    dr = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dr, r6, r7, scr, 3);
    dhi = dh; dres = dr;
    wsu_swap32(xh0, yh0); wsu_swap32(xh1, yh1);                            // lanes 0-31: f16 ch 0-7, lanes 32-63: f16 ch 8-15
    g = mk_u4(xh0, xh1, yh0, yh1);
}
// Q granules of TWO pixels from their nibble words: lanes 0-31 return pixel 0's granule, lanes 32-63 pixel 1's (one exchange serves both: a
// lower lane needs its partner's words of pixel 0, an upper lane its partner's words of pixel 1).
__device__ __forceinline__ u32x4 wsu_q4_pair(uint32_t dhi0, uint32_t dres0, uint32_t dhi1, uint32_t dres1) {
    wsu_swap32(dhi0, dhi1); wsu_swap32(dres0, dres1);                      // every lane: word 0 = channels 0-3 | 8-11, word 1 = channels 4-7 | 12-15 of ITS pixel
    return mk_u4(__builtin_amdgcn_perm(dhi1, dhi0, 0x05040100u), __builtin_amdgcn_perm(dhi1, dhi0, 0x07060302u),
                 __builtin_amdgcn_perm(dres1, dres0, 0x05040100u), __builtin_amdgcn_perm(dres1, dres0, 0x07060302u));
}
// ... of ONE pixel: valid in lanes 0-31
__device__ __forceinline__ u32x4 wsu_q4_single(uint32_t dhi, uint32_t dres) {
    uint32_t ph = dhi, pr = dres;
    wsu_swap32(dhi, ph); wsu_swap32(dres, pr);                             // lanes 0-31: ph / pr = the partner's words
    return mk_u4(__builtin_amdgcn_perm(ph, dhi, 0x05040100u), __builtin_amdgcn_perm(ph, dhi, 0x07060302u),
                 __builtin_amdgcn_perm(pr, dres, 0x05040100u), __builtin_amdgcn_perm(pr, dres, 0x07060302u));
}
// Pixel-per-thread producer (first layer): all 16 channels of a (pixel, chunk) in one thread -> the three granules and the scale byte
__device__ __forceinline__ void wsu_q4_encode16(const f32x4 (&v)[4], u32x4& h0, u32x4& h1, u32x4& q, uint32_t& sb) {
    uint32_t h[8]; float r[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 a = {v[k][0], v[k][1]}, b = {v[k][2], v[k][3]};
        h[2 * k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, f16x2)); h[2 * k + 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, f16x2));
        r[4 * k] = wsu_sub_f16_lo(v[k][0], h[2 * k]); r[4 * k + 1] = wsu_sub_f16_hi(v[k][1], h[2 * k]);
        r[4 * k + 2] = wsu_sub_f16_lo(v[k][2], h[2 * k + 1]); r[4 * k + 3] = wsu_sub_f16_hi(v[k][3], h[2 * k + 1]);
    }
    h0 = mk_u4(h[0], h[1], h[2], h[3]); h1 = mk_u4(h[4], h[5], h[6], h[7]);
    const int e = wsu_q4_block_exp(wsu_f16x16_max_abs_bits(h0, h1));
    const float sc = wsu_pow2f(e), scr = wsu_pow2f(e - 11);
    sb = (uint32_t)(e + 127);
    uint32_t q2 = 0, q3 = 0;
    q2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q2, r[0], r[1], scr, 0); q2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q2, r[2], r[3], scr, 1);
    q2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q2, r[4], r[5], scr, 2); q2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q2, r[6], r[7], scr, 3);
    q3 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q3, r[8], r[9], scr, 0); q3 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q3, r[10], r[11], scr, 1);
    q3 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q3, r[12], r[13], scr, 2); q3 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q3, r[14], r[15], scr, 3);
    q = mk_u4(wsu_f16x8_to_fp4(h0, sc), wsu_f16x8_to_fp4(h1, sc), q2, q3);
}
// fp4 cross-term MFMA: one granule per operand, per-lane E8M0 scale bytes
#ifndef WSU_PROBE16
#define WSU_PROBE16 0           // timing-only build (make qprobe16; results are wrong): every 32x32 matrix instruction of conv3x3_q.hip as TWO 16x16 instructions
#endif                          // of the same cycles and products on the same operand registers -- what the 16x16x32 / 16x16x128 shapes would buy at no padding
__device__ __forceinline__ void wsu_mfma_q4(const u32x4& a, const u32x4& b, int scale_a, int scale_b, f32x16& acc) {
    const i32x8 av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0}, bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
#if WSU_PROBE16
    f32x4 c0 = {acc[0], acc[1], acc[2], acc[3]}, c1 = {acc[8], acc[9], acc[10], acc[11]};
    c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c0, 4, 4, 0, scale_a, 0, scale_b);
    c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c1, 4, 4, 0, scale_a, 0, scale_b);
    acc[0] = c0[0]; acc[1] = c0[1]; acc[2] = c0[2]; acc[3] = c0[3]; acc[8] = c1[0]; acc[9] = c1[1]; acc[10] = c1[2]; acc[11] = c1[3];
#else
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 4, 4, 0, scale_a, 0, scale_b);
#endif
}

// the same for a stored GRADIENT granule: e4m3(g * 4)
__device__ __forceinline__ u32x2 wsu_f16x8_to_fp8_grad(const u32x4& h) {
    const _Float16 m = (_Float16)112.f;
    const f16x8 lim = {m, m, m, m, m, m, m, m};
    const f16x8 c = __builtin_elementwise_max(__builtin_elementwise_min(__builtin_bit_cast(f16x8, h), lim), -lim);
    i16x2 lo = {0, 0}, hi = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, __builtin_shufflevector(c, c, 0, 1), WSU_F8_G_DIV, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, __builtin_shufflevector(c, c, 2, 3), WSU_F8_G_DIV, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, __builtin_shufflevector(c, c, 4, 5), WSU_F8_G_DIV, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, __builtin_shufflevector(c, c, 6, 7), WSU_F8_G_DIV, true);
    return mk_u2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
}
// hi*hi on the f16 pipe; (a8, b8) = {block 0: e4m3(w) x residual(x), block 1: residual(w) x e4m3(x)} on the block-scaled fp8 pipe.
// Operand layout of the scaled instruction (tools/mfma_scale_layout_probe.hip): registers 0-3 of every lane are scale block 0, registers
// 4-7 block 1; inside a block lanes 0-31 hold k = 0..15 and lanes 32-63 k = 16..31; block b is scaled by byte 0 of the scale registers of
// lanes 32b .. 32b+31 -> a lane passes (hh ? block-1 scale : block-0 scale).
__device__ __forceinline__ void wsu_mfma_f16(const u32x4& a, const u32x4& b, f32x16& acc) {
#if WSU_PROBE16
    f32x4 c0 = {acc[4], acc[5], acc[6], acc[7]}, c1 = {acc[12], acc[13], acc[14], acc[15]};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c1, 0, 0, 0);
    acc[4] = c0[0]; acc[5] = c0[1]; acc[6] = c0[2]; acc[7] = c0[3]; acc[12] = c1[0]; acc[13] = c1[1]; acc[14] = c1[2]; acc[15] = c1[3];
#else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ void wsu_mfma_f8x2(const u32x4& a_blk0, const u32x4& a_blk1, const u32x4& b_blk0, const u32x4& b_blk1,
                                              int scale_a, int scale_b, f32x16& acc) {
    i32x8 a = {(int)a_blk0.x, (int)a_blk0.y, (int)a_blk0.z, (int)a_blk0.w, (int)a_blk1.x, (int)a_blk1.y, (int)a_blk1.z, (int)a_blk1.w};
    i32x8 b = {(int)b_blk0.x, (int)b_blk0.y, (int)b_blk0.z, (int)b_blk0.w, (int)b_blk1.x, (int)b_blk1.y, (int)b_blk1.z, (int)b_blk1.w};
#if WSU_PROBE == 2                                                  // timing probe: the same registers read as fp4 operands (4x the bf16 rate)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, scale_a, 0, scale_b);
#elif WSU_PROBE == 4                                                // timing probe: fp6 (e2m3) operands
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, 0, scale_a, 0, scale_b);
#else
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, scale_a, 0, scale_b);
#endif
}

__device__ __forceinline__ float wsu_bf16_to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

// One MFMA "k-group" on two 16-byte operand granules per lane, for every mode.
//   F32   : a/b hold 4 fp32 each -> 4x v_mfma_f32_32x32x2_f32 (k = {4g_lo+e, 4g_hi+e})
//   BF16  : a/b hold 8 bf16 each -> 1x v_mfma_f32_32x32x16_bf16
template <int MODE>
__device__ __forceinline__ void wsu_mfma_step(const u32x4& a, const u32x4& b, f32x16& acc) {
    if constexpr (MODE == WSU_MODE_F32) {
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
}

// Compile-time loop: the body receives std::integral_constant indices, so register arrays are indexed by
// front-end constants and never demoted to scratch (hipcc keeps `arr[k]` of a `#pragma unroll` loop on the
// stack when the loop is unrolled after SROA).
template <int N> struct wsu_static_for_t {
    template <class F> __device__ __forceinline__ static void run(F&& f) {
        wsu_static_for_t<N - 1>::run(f);
        f(std::integral_constant<int, N - 1>{});
    }
};
template <> struct wsu_static_for_t<0> { template <class F> __device__ __forceinline__ static void run(F&&) {} };
#define WSU_STATIC_FOR(N, VAR, ...) wsu_static_for_t<N>::run([&](auto VAR##_c) __attribute__((always_inline)) { constexpr int VAR = decltype(VAR##_c)::value; __VA_ARGS__ })

// XCD-aware remap of the linear workgroup id: consecutive logical ids (neighbouring tiles, which share
// halo rows and weights) land on the same XCD / L2.  Bijective for any grid size (guide section 5, T1).
__device__ __forceinline__ unsigned wsu_xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, k = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
