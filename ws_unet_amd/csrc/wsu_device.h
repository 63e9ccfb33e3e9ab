// Shared device-side helpers for the gfx950 kernels of libwsu.
#pragma once
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
