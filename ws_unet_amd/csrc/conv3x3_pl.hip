// K1p: the 3x3 convolution of the inference path in mode F16F8P -- the f16f8 arithmetic of conv3x3.hip on PLANAR activations, fed by
// LDS-DMA and software-pipelined across chunks AND tiles by one persistent workgroup per CU.
//
// Why a second kernel (profiles/r02/conv3x3_units_probe.md): in conv3x3_kernel the time is affine in the matrix work, t = a + b * units, and
// the part `a` that does not overlap the matrix pipe (register staging: global loads at 48 used bytes per 192..768-byte pixel stride, the
// ds_write commit with its fp8 derivation, two barriers per chunk, the LDS round trip of the epilogue, an exposed prologue per tile) is
// 45-50 % of every layer.  Here none of that work exists:
//   * activations are stored planar, [n][C/16 chunks][3 planes][H][W][16 B] (planes: f16 ch 0-7 | f16 ch 8-15 | e4m3 residuals ch 0-15 =
//     three of the four LDS planes of a chunk; the fourth, the e4m3 copies e4m3(x / 4), is derived from the f16 planes by the loader that
//     fetched them), so an input-tile row is one contiguous 544-byte run per plane and
//     `global_load_lds_dwordx4` moves it into the granule-planar LDS image with no registers, no VALU and no ds_write;
//   * two LDS stages (input tile 18 x 34 pixels x 64 B + the 36 KB weight slice = 76 KB each): the DMA of chunk c+1 -- or of the NEXT
//     tile's first chunk -- is in flight while chunk c is multiplied; ONE raw s_barrier per chunk;
//   * workgroup = 8 waves x (64 co x 2 rows x 32 px) on a 16 x 32-pixel tile, one workgroup per CU, <= 256 VGPRs (4 accumulator tiles);
//   * the epilogue goes from the accumulators straight to planar global memory: bias + ReLU, the f16 / e4m3 encodings, then two
//     v_permlane32_swap per 8 channels put whole 16-byte granules into single lanes so that every store instruction writes two
//     contiguous 512-byte runs -- no LDS, no barrier; the 2x2 max-pool is taken in registers (rows = the wave's two rows, columns =
//     lane pairs), the 1x1 head + sigmoid is a per-lane dot product plus one swap.
// Replaces nn.Conv2d(k=3, reflect) + F.relu (+ torch.cat, nn.MaxPool2d, outconv + sigmoid) of src/unet/model/unet.py:141-189 like
// conv3x3.hip; same packed weights (wsu_conv3x3_pack, mode F16F8).
#include "wsu_device.h"
#include <cstdlib>

namespace {

constexpr int TW = 32, TH = 16, IW = TW + 2, IH = TH + 2;
constexpr int NPIX = IW * IH;                             // 612 input-tile pixels
constexpr int PLANE = NPIX * 16;                          // 9792 B per granule plane
constexpr int LDS_IN = WSU_GRAN * PLANE;                  // 39168
constexpr int LDS_W = 9 * WSU_GRAN * WSU_COB * 16;        // 36864
constexpr int STAGE = LDS_IN + LDS_W;                     // 76032
constexpr int LDS_EXTRA = 2 * STAGE;                      // bias [1024] | head_w [4][64] | head_b [4]
constexpr int LDS_F1 = LDS_EXTRA + 1024 * 4 + 4 * 64 * 4 + 16;   // fused first layer: w1 tap-major [9][64] | b1 [64]
constexpr int LDS_TOTAL = LDS_F1 + 9 * 64 * 4 + 64 * 4;
// f16 products only (kernel variant HONLY): a stage holds 2 input planes + the 2 f16 weight planes of every tap = half the bytes, so FOUR stages
// fit where two did and the DMA runs three steps ahead -- a step of 9 instead of 19 matrix units (1.2 us) no longer covers the DMA's latency
constexpr int LDS_IN_H = 2 * PLANE;                       // 19584
constexpr int STAGE_H = LDS_IN_H + 9 * 2 * WSU_COB * 16;  // 38016
constexpr int NSTAGE_H = 4;
static_assert(NSTAGE_H * STAGE_H <= 2 * STAGE, "the HONLY stages live in the two full stages' LDS");
constexpr int NWAVE = 8, NLOAD = 4, NT = (NWAVE + NLOAD) * 64;      // 8 matrix waves + 4 loader waves (one per SIMD)
constexpr int IN_SEG = (NPIX + 63) / 64;                  // 10 wave-instructions per plane (the last one 36 lanes wide)
constexpr int HBM_PLANES = 3;                             // stored planes per chunk: f16 ch 0-7 | f16 ch 8-15 | e4m3 residuals; LDS plane 3 is derived
constexpr int IN_SLOTS = HBM_PLANES * IN_SEG;             // 30
// The DMA of a step is issued by FOUR dedicated loader waves (waves 8-11, one per SIMD: 16-17 pieces each), the matrix waves never touch the
// vector-memory pipe inside the loop.  Measured on the way (profiles/r02/conv3x3_pl_stamps.md): a piece costs its issuing wave ~150 cycles
// in the queue, so (v1) all eight waves issuing their pieces after the barrier idled the matrix pipe for 2-3 k cycles per step, (v2/v4)
// threading the pieces through the matrix section stalled the in-order waves just as long, (v3) giving them to one wave per SIMD let its
// partner run alone (68 % pipe time), (v5) two loader waves were the critical path.  A wave that only loads costs 168 instead of 256
// registers per matrix wave -- nothing else.  Round 3 made the loaders' instruction stream lean (below, "loader side").
constexpr int IN_PER_WAVE = (IN_SLOTS + NLOAD - 1) / NLOAD;   // 8 (slots 30, 31 do not exist)
constexpr int W_SLOTS = LDS_W / 1024;                     // 36
constexpr int W_PER_WAVE = W_SLOTS / NLOAD;               // 9
static_assert(W_SLOTS % NLOAD == 0, "weight DMA slots divide over the loader waves");

struct PlArgs {
    const char* x1; const char* x2; const char* wp; const float* bias;
    char* y; char* ypool;
    const float* head_w; const float* head_b; float* head_out; float* head_logit; int head_cout;
    int n, h, w, c1, c2, cout;
    int tiles_x, tiles_y, ncb, nch1, nch;
    int relu;
    int ntiles;                                           // n * tiles_y * tiles_x * ncb
    const float* img; const float* w1; const float* b1;   // fused first layer (kernel variant F1): the 64 input channels are computed by the loaders
    int xres;                                             // 0: the inputs' residual plane (plane 2) is not used (kernel variant XRES = false)
    unsigned* range_flag;                                 // optional: bit 0 is set when a stored activation exceeds the encodable range (|x| > 448)
    int ablate;                                           // timing-only experiments (WSU_PL_ABLATE, bits; results wrong when != 0): 1 = no DMA after step 0,
                                                          // 2 = no epilogue (accumulators dropped), 8 = the loaders do not derive LDS plane 3
    // data-gradient variant (GRAD): zero padding, gradient encodings, no bias / ReLU; output chunks < nco1 go to y, the others to y2 (fused
    // concat: one gradient per source); mask / mask2 (optional, planar activations shaped like y / y2): the ReLU mask (x > 0) of the layer
    // that produced this conv's input, applied to the result; images n >= k * imgs_per_wset take weight set k (ring strips)
    char* y2; int nco1;
    const char* mask; const char* mask2;
    int imgs_per_wset; size_t wset_bytes;
    // 1-bit ReLU masks (round 3; layout: wsu.h "relu_mask planes", [n][C/8][hp][wp] bytes).  Forward (plain variant): relu_mask_out (optional)
    // receives one byte per stored (pixel, 8-channel granule).  GRAD: mbits / mbits2 (optional) replace `mask` / `mask2` -- the loaders bring
    // the tile's 8 granule planes x 512 bytes in by LDS-DMA instead of re-reading 2 bytes per element of the producing layer's f16 planes.
    unsigned char* relu_mask_out; const unsigned char* mbits; const unsigned char* mbits2;
    int msplit;                                           // 1: work items are half-blocks of 32 output channels (kernel variant MSPLIT); ncb = 2 * cout / 64
    int honly;                                            // GRAD, 1: f16 products only (kernel variant HONLY; wsu.h "products" of the backward entry points)
};

struct Tile { int n, y0, x0, cb, mh; };                   // mh: the 32-channel half of block cb this item computes (kernel variant MSPLIT), else 0

// Diagnostic stamps (only in the -DWSU_PL_STAMPS build): per workgroup the accumulated shader cycles of each phase of the chunk loop and
// the s_memrealtime span, read back with wsu_debug_read_pl_stamps().  Values go to a buffer nothing else reads.
__device__ unsigned long long g_pl_stamps[256 * 8];

__device__ __forceinline__ Tile tile_of(const PlArgs& a, int t) {
    Tile r;
    r.cb = t % a.ncb; t /= a.ncb;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    r.n = t / a.tiles_y; r.y0 = ty * TH; r.x0 = tx * TW;
    r.mh = 0;
    if (a.msplit) { r.mh = r.cb & 1; r.cb >>= 1; }         // a.ncb counts half-blocks then
    return r;
}

// ---- loader side: LDS-DMA through buffer descriptors ------------------------------------------------------------------------------
// Round 3: the loaders' instruction stream.  Round 2's loader computed, per piece and step, its plane / segment from a run-time wave index
// (divisions by 10), a 64-bit per-lane source address, a select between it and the zero block and a lane-alive test: ~540 vector and ~600
// scalar instructions per step and wave for 17 DMA instructions (SGPRs spilled into VGPR lanes on the way), issued on the SIMD its two
// matrix waves need for their MFMAs (profiles/r03/conv3x3_pl_loader.md).  Now:
//   * the loader body is instantiated per loader wave (LW = 0..3): slot -> (plane, segment) is compile-time;
//   * a piece = `buffer_load_dwordx4 ... offen lds` with a wave-uniform descriptor of the (image, chunk)'s 3 stored planes and a per-lane
//     32-bit byte offset that is computed ONCE PER TILE (8 registers); out-of-image lanes of the GRAD variant carry an offset beyond the
//     descriptor's size -- the hardware range check makes them read zeros (the zero padding), no select, no zero block;
//   * weight pieces: one descriptor for the packed weights, offset = chunk base (scalar) + lane * 16.
// Per step and wave that leaves ~3 instructions per piece, two of them scalar.
typedef __attribute__((address_space(3))) void lds_void;

template <int LW> struct LoaderGeo {
    static constexpr int slot(int k) { return LW + NLOAD * k; }
    static constexpr bool exists(int k) { return slot(k) < IN_SLOTS; }
    static constexpr int plane(int k) { return slot(k) / IN_SEG; }
    static constexpr int seg(int k) { return slot(k) % IN_SEG; }
};
constexpr unsigned OOB = 0xFFFFFFF0u;                              // beyond any descriptor (num_records = 48 * h * w < 2^32 - 16): reads as zeros, a store is dropped

// per-lane byte offsets (inside one (image, chunk)'s 3 planes) of this wave's input slots for tile t
template <int LW, bool GRAD>
__device__ __forceinline__ void plan_tile(const PlArgs& a, const Tile& t, int lane, unsigned (&voff)[IN_PER_WAVE]) {
    const unsigned hw = (unsigned)(a.h * a.w);
    WSU_STATIC_FOR(IN_PER_WAVE, k, {
        if constexpr (LoaderGeo<LW>::exists(k)) {
            constexpr int plane = LoaderGeo<LW>::plane(k), seg = LoaderGeo<LW>::seg(k);
            const int idx = seg * 64 + lane;
            const int r = idx / IW, c = idx - r * IW;
            if constexpr (GRAD) {                                      // zero padding: pixels outside the image read zeros
                const int yy = t.y0 - 1 + r, xx = t.x0 - 1 + c;
                const bool inside = yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
                voff[k] = inside ? (plane * hw + (unsigned)(yy * a.w + xx)) * 16u : OOB;
            } else {
                const int yy = wsu_reflect(t.y0 - 1 + r, a.h), xx = wsu_reflect(t.x0 - 1 + c, a.w);
                voff[k] = (plane * hw + (unsigned)(yy * a.w + xx)) * 16u;
            }
        } else {
            voff[k] = 0;
        }
    });
}

// the DMA of one chunk step into stage `st`: this wave's 7-8 input pieces, then its 9 weight pieces
typedef __attribute__((address_space(3))) char lds_char;
template <int LW, bool XRES, bool WEIGHTS_ONLY, bool HONLY = false>
__device__ __forceinline__ void issue_dma(const PlArgs& a, int tn, int tcb, int c, lds_char* st, int lane, const unsigned (&voff)[IN_PER_WAVE], int lw_rt = LW) {
    const unsigned plane4 = (unsigned)(a.h * a.w) * 16u * HBM_PLANES;          // bytes of one chunk of one image (3 stored planes)
    const char* in_src = c < a.nch1 ? a.x1 + ((size_t)tn * a.nch1 + c) * plane4
                                    : a.x2 + ((size_t)tn * (a.nch - a.nch1) + (c - a.nch1)) * plane4;
    const char* w_src = a.wp;
    if (a.imgs_per_wset > 0) w_src += (size_t)(tn / a.imgs_per_wset) * a.wset_bytes;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_src), 0, (int)plane4, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w_src), 0, 0x7FFFFFF0, 0x00020000);
    const int w_base = (tcb * a.nch + c) * LDS_W;                                // scalar
    const unsigned lane16 = (unsigned)lane * 16u;
    // input pieces first, the 9 weight pieces last: the loader derives LDS plane 3 from its landed input pieces (`s_waitcnt vmcnt(9)`)
    // while its weight pieces are still in flight
    if constexpr (!WEIGHTS_ONLY) {
        WSU_STATIC_FOR(IN_PER_WAVE, k, {
            if constexpr (LoaderGeo<LW>::exists(k) && (XRES || LoaderGeo<LW>::plane(k) != 2)) {
                constexpr int plane = LoaderGeo<LW>::plane(k), seg = LoaderGeo<LW>::seg(k);
                lds_void* dst = (lds_void*)(st + plane * PLANE + seg * 1024);
                if constexpr (seg == IN_SEG - 1) {                      // the last segment of a plane is 36 lanes wide
                    if (lane < NPIX - (IN_SEG - 1) * 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, dst, 16, voff[k], 0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, dst, 16, voff[k], 0, 0, 0);
                }
            }
        });
    }
    // weight slot = tap * 4 + granule plane (f16 ci 0-7 | f16 ci 8-15 | e4m3 copies | e4m3 residuals), so loader wave LW carries granule
    // plane LW of every tap: with f16 products only (HONLY) waves 2 and 3 have no weight pieces
    if constexpr (HONLY) {
        if constexpr (LW < 2) {                                                  // LDS: [tap][f16 plane LW][64 co][16 B] behind the two input planes
            WSU_STATIC_FOR(W_PER_WAVE, k, {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(st + LDS_IN_H + (k * 2 + LW) * 1024), 16, lane16, w_base + (LW + NLOAD * k) * 1024, 0, 0);
            });
        }
    } else {
        WSU_STATIC_FOR(W_PER_WAVE, k, {
            const int wslot = (WEIGHTS_ONLY ? lw_rt : LW) + NLOAD * k;           // compile-time unless the fused-first-layer loader (one copy, run-time wave index)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(st + LDS_IN + wslot * 1024), 16, lane16, w_base + wslot * 1024, 0, 0);
        });
    }
}

// value of lane ^ 1 by a DPP quad permutation (a VALU modifier: no LDS crossbar round trip like ds_bpermute)
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}

__device__ __forceinline__ void swap32(uint32_t& upper_of, uint32_t& lower_of) {
    // lanes 32-63 of `upper_of` <-> lanes 0-31 of `lower_of`
    const auto r = __builtin_amdgcn_permlane32_swap(upper_of, lower_of, false, false);
    upper_of = r[0]; lower_of = r[1];
}

// Stamps are compiled in only with -DWSU_PL_STAMPS (make probes -> libwsu_plstamp.so): in the product build STAMP() is empty.
#ifdef WSU_PL_STAMPS
#define STAMP(var) var = __builtin_amdgcn_s_memtime()
#else
#define STAMP(var) do {} while (0)
#endif

// ================= loader wave LW: the whole DMA of step j+1 right after the barrier that opens step j ====================================
template <int LW, bool XRES, bool F1, bool GRAD, bool HONLY = false>
__device__ __forceinline__ void pl_loader(const PlArgs& a, char* smem, int lane, int lw, int G, int J, int lw_rt = LW) {
    // F1 (the loaders compute e11): ONE copy of this body with a run-time wave index -- its 27 image values + 16 accumulators per lane, inlined
    // four times beside the matrix waves' code, made the register allocator spill ~160 registers; the input-slot geometry is unused there
    const int lw8 = F1 ? lw_rt : LW;
    float* s_w1 = reinterpret_cast<float*>(smem + LDS_F1);
    float* s_b1 = s_w1 + 9 * 64;
    unsigned char* s_mask = reinterpret_cast<unsigned char*>(smem + LDS_EXTRA);   // GRAD: [4 output chunks][2 f16 planes][512 px] bytes of 8 mask bits (the bias slot)
    [[maybe_unused]] unsigned long long t_wait = 0, t_bar = 0, t_dma = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0, t0 = 0, rt0 = 0;     // stamps build only
    STAMP(t0);
#ifdef WSU_PL_STAMPS
    rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    unsigned voff[IN_PER_WAVE];
    Tile t = tile_of(a, lw);
    // ---- fused first layer (F1): the loaders COMPUTE the input planes of a step instead of fetching them -- relu(b1 + w1 * 3x3 window of
    // the image), fp32 FMAs in the tap order of first_pl_kernel, the same encodings, written where the DMA would have put them: the
    // result is bitwise that of first_pl + this kernel, xe11 never exists in HBM.  Lane = 3 of the tile's 612 positions (fixed per
    // tile, their 27 image values stay in registers over the 4 chunks); only the 36 weight pieces of a step still come by DMA.
    constexpr int F1_PX = (NPIX + NLOAD * 64 - 1) / (NLOAD * 64);           // 3
    float pimg[F1 ? F1_PX : 1][9];
    float f1_max = 0.f;                                                     // range flag of the computed (never stored) xe11 values
    auto f1_window = [&](const Tile& tt) __attribute__((always_inline)) {
        if constexpr (F1) {
            const float* img = a.img + (size_t)tt.n * a.h * a.w;
#pragma unroll
            for (int k = 0; k < F1_PX; ++k) {
                const int idx = min(lw8 * 64 + lane + NLOAD * 64 * k, NPIX - 1);
                const int r = idx / IW, cc = idx - r * IW;
                const int yy = wsu_reflect(tt.y0 - 1 + r, a.h), xx = wsu_reflect(tt.x0 - 1 + cc, a.w);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
                    pimg[k][tp] = img[(size_t)wsu_reflect(yy + tp / 3 - 1, a.h) * a.w + wsu_reflect(xx + tp % 3 - 1, a.w)];
            }
        }
    };
    auto f1_chunk = [&](int c, char* st) __attribute__((always_inline)) {
        if constexpr (F1) {
#pragma unroll
            for (int k = 0; k < F1_PX; ++k) {
                const int idx = lw8 * 64 + lane + NLOAD * 64 * k;
                f32x4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const f32x4*>(s_b1 + c * 16 + 4 * g);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(s_w1 + tp * 64 + c * 16 + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = fmaf(pimg[k][tp], w4[e], v[g][e]);
                    }
                uint32_t h[8], lo[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[g][e] = fmaxf(v[g][e], 0.f); f1_max = fmaxf(f1_max, v[g][e]); }
                    wsu_split4_f16r8(v[g], WSU_F8_XLO_DIV, h[2 * g], h[2 * g + 1], lo[g]);
                }
                // the e4m3 copies from the f16 values, exactly as derive_x8 makes them for fetched planes
                const u32x2 xa = wsu_f16x8_to_fp8(mk_u4(h[0], h[1], h[2], h[3])), xb = wsu_f16x8_to_fp8(mk_u4(h[4], h[5], h[6], h[7]));
                if (idx < NPIX) {
                    char* d = st + idx * 16;
                    *reinterpret_cast<u32x4*>(d) = mk_u4(h[0], h[1], h[2], h[3]);
                    *reinterpret_cast<u32x4*>(d + PLANE) = mk_u4(h[4], h[5], h[6], h[7]);
                    *reinterpret_cast<u32x4*>(d + 2 * PLANE) = mk_u4(lo[0], lo[1], lo[2], lo[3]);
                    *reinterpret_cast<u32x4*>(d + 3 * PLANE) = mk_u4(xa.x, xa.y, xb.x, xb.y);
                }
            }
        }
    };
    // LDS plane 3 (the e4m3 copies e4m3(x / 4) of the second cross term) is not stored in HBM: each loader derives it from the f16 granules
    // IT fetched (same lane, after its own vmcnt wait -- no cross-wave dependency), 3 instructions per pair of values.
    auto derive_x8 = [&](char* st) __attribute__((always_inline)) {
        if constexpr (!F1) {
            WSU_STATIC_FOR(IN_PER_WAVE, k, {
                if constexpr (LoaderGeo<LW>::exists(k) && LoaderGeo<LW>::plane(k) < 2) {
                    constexpr int plane = LoaderGeo<LW>::plane(k), seg = LoaderGeo<LW>::seg(k);
                    const int idx = seg * 64 + lane;
                    if (seg < IN_SEG - 1 || idx < NPIX) {
                        const u32x4 hgr = *reinterpret_cast<const u32x4*>(st + plane * PLANE + idx * 16);
                        *reinterpret_cast<u32x2*>(st + 3 * PLANE + idx * 16 + plane * 8) = GRAD ? wsu_f16x8_to_fp8_grad(hgr) : wsu_f16x8_to_fp8(hgr);
                    }
                }
            });
        }
    };
    // GRAD: the ReLU mask of the tile = sign test of the producing layer's f16 planes at the tile's 512 pixels x 64 output channels:
    // 16 granules per loader lane, fetched during the tile's first step and committed as one byte each before its second barrier
    // (the epilogue reads them after the last one; needs nch >= 2).
    constexpr int MK = GRAD ? 16 : 1;
    u32x4 mreg[MK];
    bool mask_pending = false;
    auto mask_issue = [&](const Tile& tt) __attribute__((always_inline)) {
        if constexpr (GRAD) {
            const bool d1 = tt.cb * 4 < a.nco1;
            const unsigned char* mb = d1 ? a.mbits : a.mbits2;
            if (mb != nullptr) {
                // 1-bit masks: the tile's 8 granule planes x (16 rows x 32 bytes) go straight into s_mask by LDS-DMA, 256 bytes (8 rows) per
                // instruction, 4 instructions per loader wave (granule planes 2 LW, 2 LW + 1); rows and row pitch are padded to the tile
                // grid (wsu_mask_hp / _wp), so every access is aligned and inside the plane
                const int ncm = d1 ? a.nco1 : (a.cout >> 4) - a.nco1, oc0 = d1 ? tt.cb * 4 : tt.cb * 4 - a.nco1;
                const int hp = wsu_mask_hp(a.h), wp = wsu_mask_wp(a.w);
                const auto rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(mb), 0, 0x7FFFFFF0, 0x00020000);
                const unsigned mvoff = (unsigned)((lane >> 3) * wp + (lane & 7) * 4);
                lds_char* sm3 = (lds_char*)s_mask;
                WSU_STATIC_FOR(4, i, {
                    constexpr int gp = 2 * LW + (i >> 1), half = i & 1;
                    const int soff = (((tt.n * ncm * 2 + oc0 * 2 + gp) * hp) + tt.y0 + half * 8) * wp + tt.x0;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_void*)(sm3 + gp * 512 + half * 256), 4, mvoff, soff, 0, 0);
                });
                return;
            }
            const char* mk = d1 ? a.mask : a.mask2;
            if (mk == nullptr) return;
            const int ncm = d1 ? a.nco1 : (a.cout >> 4) - a.nco1, oc0 = d1 ? tt.cb * 4 : tt.cb * 4 - a.nco1;
            const size_t hw = (size_t)a.h * a.w;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int combo = k >> 1, pix = (k & 1) * 256 + lw8 * 64 + lane;          // combo = output chunk * 2 + plane
                const int yy = min(tt.y0 + (pix >> 5), a.h - 1), xx = min(tt.x0 + (pix & 31), a.w - 1);
                mreg[k] = *reinterpret_cast<const u32x4*>(mk + ((((size_t)tt.n * ncm + oc0 + (combo >> 1)) * HBM_PLANES + (combo & 1)) * hw + (size_t)yy * a.w + xx) * 16);
            }
            mask_pending = true;
        }
    };
    auto mask_commit = [&]() __attribute__((always_inline)) {
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                unsigned bits = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {                                             // f16 > 0 <=> its 16 bits > 0 as a signed integer
                    const int wd = (int)mreg[k][e];
                    bits |= ((short)(wd & 0xFFFF) > 0 ? 1u : 0u) << (2 * e);
                    bits |= (wd >= 0x10000 ? 1u : 0u) << (2 * e + 1);
                }
                s_mask[(k >> 1) * 512 + (k & 1) * 256 + lw8 * 64 + lane] = (unsigned char)bits;
            }
        }
    };
    lds_char* smem3 = (lds_char*)smem;                                    // LDS address space from here on: no generic-pointer null checks per piece
    if constexpr (HONLY) {
        // ---- four stages, DMA three steps ahead.  Issue order of this wave: P_0 P_1 P_2 | barrier 0 | M_T0 P_3 | barrier 1 | P_4 | ... where P_s = the
        // PER pieces of step s and M_T = the 4 mask pieces of the tile whose first step just opened (issued BEFORE that barrier's P, so they are
        // older than every piece issued later).  `s_waitcnt vmcnt(n)` = "all but my n youngest operations have landed": before barrier j the
        // wave needs P_j -- younger than it are the steps issued behind it and, during a tile's first three steps, its M -- and, before the
        // barrier of a tile's LAST step, its M (the epilogue reads them): younger than M are only the steps from (first step + 3) on.
        static_assert(!F1 && GRAD && !XRES, "HONLY is the data gradient's variant");
        constexpr int NST = NSTAGE_H, AHEAD = NST - 1, PER = 5 + (LW < 2 ? W_PER_WAVE : 0);
        static_assert((AHEAD - 1) * PER + 4 < 64 && IN_SLOTS == 30, "vmcnt immediates / 5 input pieces of planes 0, 1 per wave");
        auto wait_vm = [&](int steps, bool plus_masks) __attribute__((always_inline)) {
            switch (steps * 2 + (plus_masks ? 1 : 0)) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER + 4) : "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER + 4) : "memory"); break;
            }
        };
        int ci = 0, kti = 0;                                               // chunk / tile counter of the next ISSUE (tile t, offsets voff)
        auto issue_step = [&](int js) __attribute__((always_inline)) {
            issue_dma<LW, false, false, true>(a, t.n, t.cb, ci, smem3 + (js % NST) * STAGE_H, lane, voff, LW);
            if (++ci == a.nch && js + 1 < J) { ci = 0; ++kti; t = tile_of(a, lw + kti * G); plan_tile<LW, GRAD>(a, t, lane, voff); }
        };
        if (J > 0) plan_tile<LW, GRAD>(a, t, lane, voff);
        for (int k = 0; k < AHEAD && k < J; ++k) issue_step(k);
        Tile tb = tile_of(a, lw);                                          // tile / chunk of the step whose barrier comes next
        int cb = 0, ktb = 0;
        bool masks_dma = false;                                            // this tile's masks came by DMA (4 pieces in this wave's vmcnt order)
        for (int j = 0; j < J; ++j) {
            const int ya = min(J - 1 - j, AHEAD - 1);                      // steps issued behind step j so far
            int steps = ya; bool plus = masks_dma && cb <= AHEAD - 1;
            if (masks_dma && cb + 1 == a.nch) { steps = max(0, min(ya, cb + ya - AHEAD + 1)); plus = false; }
            wait_vm(steps, plus);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (cb == 0) {
                mask_issue(tb);
                masks_dma = !mask_pending && ((tb.cb * 4 < a.nco1 ? a.mbits : a.mbits2) != nullptr);
                if (mask_pending) {                                        // masks from the activations' f16 planes (no relu_mask planes given): registers,
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drained on the spot -- the slow path; committed before the tile's second barrier
                    mask_commit(); mask_pending = false;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            if (j + AHEAD < J) issue_step(j + AHEAD);                      // its stage held step j - 1: every matrix wave is past it
            if (++cb == a.nch) { cb = 0; ++ktb; masks_dma = false; if (j + 1 < J) tb = tile_of(a, lw + ktb * G); }   // (the finished tile's M landed before its last barrier)
        }
        return;
    }
    if (J > 0) {
        if constexpr (F1) f1_window(t); else plan_tile<LW, GRAD>(a, t, lane, voff);
        issue_dma<LW, XRES, F1, HONLY>(a, t.n, t.cb, 0, smem3, lane, voff, lw8);
        f1_chunk(0, smem);
    }
    int c = 0, kt = 0;
    for (int j = 0; j < J; ++j) {
        STAMP(s0);
        static_assert(W_PER_WAVE == 9, "the vmcnt immediate below");
        if constexpr (!HONLY) {                                           // (f16 products only: LDS plane 3 is not used, nothing to derive)
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");                  // this wave's INPUT pieces of step j have landed (everything older than its
        if (!(a.ablate & 8)) derive_x8(smem + (j & 1) * STAGE);           // 9 youngest operations: the weight pieces, or mask loads issued after them)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // ... its weight pieces (and the mask granules) too
        if (mask_pending) { mask_commit(); mask_pending = false; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // ... and its derived / computed planes are written
        STAMP(s1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(s2);
        const bool first_chunk = c == 0;
        const Tile tj = t;
        if (j + 1 < J && !(a.ablate & 1)) {
            if (++c == a.nch) {
                c = 0; ++kt;
                t = tile_of(a, lw + kt * G);
                if constexpr (F1) f1_window(t); else plan_tile<LW, GRAD>(a, t, lane, voff);
            }
            issue_dma<LW, XRES, F1, HONLY>(a, t.n, t.cb, c, smem3 + ((j + 1) & 1) * STAGE, lane, voff, lw8);
            f1_chunk(c, smem + ((j + 1) & 1) * STAGE);
        }
        if (GRAD && first_chunk) mask_issue(tj);
        STAMP(s3);
        t_wait += s1 - s0; t_bar += s2 - s1; t_dma += s3 - s2;
    }
    if constexpr (F1) {
        if (a.range_flag && __builtin_amdgcn_ballot_w64(!(f1_max <= WSU_F8_RANGE)) != 0 && lane == 0) atomicOr(a.range_flag, 1u);
    }
#ifdef WSU_PL_STAMPS
    if (lane == 0 && LW == 0 && blockIdx.x < 32) {
        unsigned long long* d = g_pl_stamps + (blockIdx.x * 2 + 1) * 8;
        d[0] = __builtin_amdgcn_s_memtime() - t0; d[1] = __builtin_amdgcn_s_memrealtime() - rt0;
        d[2] = t_wait; d[3] = t_bar; d[4] = t_dma; d[5] = 0; d[6] = 0; d[7] = (unsigned long long)J;
    }
#endif
}


template <bool ON> __device__ __forceinline__ int opaque_if(int v) { if constexpr (ON) asm volatile("" : "+v"(v)); return v; }

// HEAD / POOL are compile-time: the kernel sits at the 168-register step (three waves per SIMD), and the head's partial sums or the pool's
// exchange registers would otherwise be carried -- and spilled -- by the variants that do not use them.
// XRES = false: the activations' residual plane is neither loaded nor multiplied (one cross term per product, the weights' residual:
// 9 f16 + 3 fp8 instructions = 15 instead of 19 matrix units per chunk, 30 instead of 40 input DMA pieces) -- `x_residual = 0`, see wsu.h.
// GRAD: the data gradient of the conv (K7p, autograd of unet.py:141-189): the same pipeline over the pre-activation gradient with the
// transposed / flipped weights (wsu_conv3x3_pack_dgrad) -- zero padding, the gradient's e4m3 scalings, no bias / ReLU, the ReLU mask of
// the producing layer applied from a bit image in LDS: the loader waves bring the layer's 1-bit relu_mask planes in by LDS-DMA (round 3) or,
// without them, build it from that layer's stored f16 planes.  The reflect adjoint's border ring is added by the caller (train_pl.hip).
// MSPLIT (round 3, small grids): a work item is HALF a tile's output channels (32 of the 64: m-half = item & 1 -- each matrix wave keeps 2
// instead of 4 accumulator tiles), so a layer with fewer tiles than CUs (e31 / e32 of unet_2 at batch 1: 128 tiles) occupies twice as many
// CUs with half the matrix work per step each; the input tile and the whole 64-channel weight slice are fetched as before.
// HONLY (round 3, a training arithmetic of the data gradient: wsu.h "products"): f16 products only -- the 9 f16 instructions of a chunk, no
// cross terms; the residual plane of the gradient, the e4m3 weight planes and the derived plane are neither fetched nor built (needs XRES = false).
// (Round 3's Q4 variant -- block-scaled fp4 cross terms with the fp4 operands derived by the loader waves -- moved to csrc/conv3x3_q.hip in round 4, where
// the producers store those operands: the default inference mode no longer runs this kernel; training forwards, the data gradient and 'f16f8p' do.)
template <int HC, bool POOL, bool XRES = true, bool F1 = false, bool GRAD = false, bool MSPLIT = false, bool HONLY = false>   // HC = head planes compiled in: 0 (no head), 1 (the reference's single output plane) or 4 (1..4)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 8)))
void conv3x3_pl_kernel(const PlArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int l31_k = l31, hh_k = hh;
    const int G = gridDim.x;
    const int lw = (int)wsu_xcd_remap(blockIdx.x, G);
    const int K = a.ntiles > lw ? (a.ntiles - lw + G - 1) / G : 0;           // tiles walked by this workgroup
    const int J = K * a.nch;                                                  // chunk steps

    float* s_bias = reinterpret_cast<float*>(smem + LDS_EXTRA);
    float* s_hw = s_bias + 1024;
    float* s_hb = s_hw + 4 * 64;
    if constexpr (!GRAD) { for (int i = tid; i < a.cout; i += NT) s_bias[i] = a.bias ? a.bias[i] : 0.f; }
    unsigned char* s_mask = reinterpret_cast<unsigned char*>(smem + LDS_EXTRA);   // GRAD: [4 output chunks][2 f16 planes][512 px] bytes of 8 mask bits (the bias slot)
    constexpr bool HEAD = HC > 0;
    if constexpr (HEAD) {
        for (int i = tid; i < a.head_cout * 64; i += NT) s_hw[i] = a.head_w[i];
        if (tid < 4) s_hb[tid] = (a.head_b && tid < a.head_cout) ? a.head_b[tid] : 0.f;
    }
    float* s_w1 = reinterpret_cast<float*>(smem + LDS_F1);
    float* s_b1 = s_w1 + 9 * 64;
    if constexpr (F1) {
        for (int i = tid; i < 9 * 64; i += NT) { const int tp = i >> 6, ch = i & 63; s_w1[i] = a.w1[ch * 9 + tp]; }     // tap-major
        if (tid < 64) s_b1[tid] = a.b1 ? a.b1[tid] : 0.f;
    }
    // (these plain loads have retired -- their values went into the LDS stores -- before the first vmcnt wait below; the barrier of
    // step 0 publishes them.  The fused first layer reads w1 / b1 BEFORE that barrier -- the loaders compute step 0's planes from them --
    // so that variant takes one barrier of its own here: without it the first tile of a workgroup was intermittently computed from
    // table entries another wave had not written yet.)
    if constexpr (F1) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    [[maybe_unused]] unsigned long long t_wait = 0, t_bar = 0, t_dma = 0, t_mma = 0, t_epi = 0, t0 = 0, rt0 = 0;               // stamps build only
    [[maybe_unused]] unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    STAMP(t0);
#ifdef WSU_PL_STAMPS
    rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    if (wv >= NWAVE) {
        // ================= loader waves (pl_loader<LW, ...>: the slot geometry of a wave is compile-time) ===========================
        if constexpr (F1) {
            pl_loader<0, XRES, F1, GRAD, HONLY>(a, smem, lane, lw, G, J, wv - NWAVE);
        } else {
            switch (wv - NWAVE) {
                case 0: pl_loader<0, XRES, F1, GRAD, HONLY>(a, smem, lane, lw, G, J); break;
                case 1: pl_loader<1, XRES, F1, GRAD, HONLY>(a, smem, lane, lw, G, J); break;
                case 2: pl_loader<2, XRES, F1, GRAD, HONLY>(a, smem, lane, lw, G, J); break;
                default: pl_loader<3, XRES, F1, GRAD, HONLY>(a, smem, lane, lw, G, J); break;
            }
        }
        return;
    }

    // ================= matrix waves ===================================================================================================
#ifdef WSU_PL_MATRIX_PRIO
    __builtin_amdgcn_s_setprio(WSU_PL_MATRIX_PRIO);                           // experiment: matrix waves above the loader wave of their SIMD
#endif
    Tile cur = tile_of(a, lw);
    constexpr int MH = MSPLIT ? 1 : 2;                                        // accumulator tiles along the output channels
    f32x16 acc[2][2];                                                         // [MH][2] used (declared with the template-dependent bound, hipcc (ROCm 7.2)
                                                                              // silently emits no host stub for ANY instantiation of the kernel)
    const int sc_a = hh ? WSU_F8_SCALE_WLO : WSU_F8_SCALE_W;
    const int sc_b = GRAD ? (hh ? WSU_F8_SCALE_G : WSU_F8_SCALE_GLO) : (hh ? WSU_F8_SCALE_X : WSU_F8_SCALE_XLO);
    int c = 0, kt = 0, j = 0;                                                 // chunk inside the tile, tile counter, chunk step
    // per-step state (set by begin_step; the matrix section's lambdas below capture it by reference)
    char* st = smem;
    const char* ldsA = smem;                                                  // + ((tap*4 + g)*64 + m*32)*16   (HONLY: tap*2 + g)
    const char* ldsB = smem;                                                  // + g*PLANE + ((q+dy)*IW + dx)*16
    const int hh_q = hh;
#if WSU_PROBE == 5
    u32x4 sa0[2], sa1[2], sb0[2], sb1[2], sah[2], sbh[2];
#endif
    auto begin_step = [&]() __attribute__((always_inline)) {
        // ---- step j: its DMA (issued by the loaders one step ago) has had a whole matrix section to land -----------------------
        STAMP(s0);
        STAMP(s1);
        __builtin_amdgcn_s_barrier();                                         // the loaders' pieces landed; everyone left the other stage
        asm volatile("" ::: "memory");
        STAMP(s2);
        st = HONLY ? smem + (j % NSTAGE_H) * STAGE_H : smem + (j & 1) * STAGE;
        STAMP(s3);
        ldsA = st + (HONLY ? LDS_IN_H : LDS_IN) + (cur.mh * 32 + l31) * 16;
        ldsB = st + ((2 * wv) * IW + l31) * 16;
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
    };
    // ---- matrix section: identical arithmetic (and accumulation order) to conv3x3_kernel<F16F8> -----------------------------
    {
        auto cross = [&](auto tp_c) __attribute__((always_inline)) {
            constexpr int tp = decltype(tp_c)::value;
            constexpr int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 2 * tp;
            constexpr bool single = 2 * tp + 1 >= 9;
            const int aoff = ((hh_q ? t1 : t0) * 4 + 2) * 64 * 16;
            const int boff = 2 * PLANE + (hh_q ? ((t1 / 3) * IW + t1 % 3) : ((t0 / 3) * IW + t0 % 3)) * 16;
            u32x4 a0[2], a1[2], b0[2], b1[2];
#if WSU_PROBE == 5                                                      // timing probe 5 (make probes): LDS fragments are read for the first group of a step only
            if (tp == 0) {
#else
            {
#endif
_Pragma("unroll")
            for (int m = 0; m < MH; ++m) {
                a0[m] = *reinterpret_cast<const u32x4*>(ldsA + aoff + m * 32 * 16);
                a1[m] = *reinterpret_cast<const u32x4*>(ldsA + aoff + 64 * 16 + m * 32 * 16);
            }
_Pragma("unroll")
            for (int q = 0; q < 2; ++q) {
                b0[q] = *reinterpret_cast<const u32x4*>(ldsB + boff + q * IW * 16);
                b1[q] = *reinterpret_cast<const u32x4*>(ldsB + boff + PLANE + q * IW * 16);
            }
#if WSU_PROBE == 5
            for (int m = 0; m < 2; ++m) { sa0[m] = a0[m]; sa1[m] = a1[m]; sb0[m] = b0[m]; sb1[m] = b1[m]; }
            } else { for (int m = 0; m < 2; ++m) { a0[m] = sa0[m]; a1[m] = sa1[m]; b0[m] = sb0[m]; b1[m] = sb1[m]; } }
#else
            }
#endif
            if (single && hh) {
                const u32x4 z = mk_u4(0, 0, 0, 0);
_Pragma("unroll")
                for (int m = 0; m < MH; ++m) { a0[m] = z; a1[m] = z; }
                b0[0] = z; b0[1] = z; b1[0] = z; b1[1] = z;
            }
_Pragma("unroll")
            for (int m = 0; m < MH; ++m)
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) wsu_mfma_f8x2(a0[m], a1[m], b0[q], b1[q], sc_a, sc_b, acc[m][q]);
        };
        auto main_term = [&](auto tap_c, auto ms_c) __attribute__((always_inline)) {
            constexpr int tap = decltype(tap_c)::value, dy = tap / 3, dx = tap % 3;
            constexpr int ms = decltype(ms_c)::value, ML = ms < 0 ? 0 : ms, MU = ms < 0 ? MH : ms + 1;
            u32x4 ah[2], bh[2];
#if WSU_PROBE == 5
            if (tap == 0) {
#else
            {
#endif
_Pragma("unroll")
            for (int m = ML; m < MU; ++m) ah[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * (HONLY ? 2 : 4) + hh) * 64 + m * 32) * 16);
_Pragma("unroll")
            for (int q = 0; q < 2; ++q) bh[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE + ((q + dy) * IW + dx) * 16);
#if WSU_PROBE == 5
            for (int m = 0; m < 2; ++m) { sah[m] = ah[m]; sbh[m] = bh[m]; }
            } else { for (int m = 0; m < 2; ++m) { ah[m] = sah[m]; bh[m] = sbh[m]; } }
#else
            }
#endif
_Pragma("unroll")
            for (int m = ML; m < MU; ++m)
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) wsu_mfma_f16(ah[m], bh[q], acc[m][q]);
        };
        constexpr std::integral_constant<int, -1> all_m{};
        // Measured neutral on this section (gpurun_out/ab_prio.log, time_pl*.log): raising the priority of waves 4-7 for its second half so
        // that SIMD partners reach the barrier together; fetching fragments one unit ahead of their matrix instructions behind scheduling
        // fences (two ahead needs 190 registers).
        static_assert(!HONLY || !XRES, "HONLY reads neither residual plane");
        auto units_all = [&]() __attribute__((always_inline)) {
        if constexpr (HONLY) {
            WSU_STATIC_FOR(9, tap, { main_term(std::integral_constant<int, tap>{}, all_m); });
        } else if constexpr (XRES) {
            WSU_STATIC_FOR(5, tp, {
#if WSU_PROBE != 3                                                      // timing probe 3 (make probes): no cross terms at all = plain f16, 9 units
                cross(std::integral_constant<int, tp>{});
#endif
                main_term(std::integral_constant<int, 2 * tp>{}, all_m);
                if constexpr (2 * tp + 1 < 9) main_term(std::integral_constant<int, 2 * tp + 1>{}, all_m);
            });
        } else {
            // one cross term: residual(w) x e4m3(x), FOUR taps per fp8 instruction -- scale block b (registers 4b..4b+3), lane half hh carry
            // the 16 channels of tap 4g + 2b + hh; both blocks use the scales (WLO, X); slots of taps > 8 pass zeros
            auto cross1 = [&](auto g_c) __attribute__((always_inline)) {
                constexpr int g = decltype(g_c)::value;
                u32x4 ab[2][2], bb[2][2];                               // [block][m] / [block][q]
                WSU_STATIC_FOR(2, b, {
                    constexpr int ta = 4 * g + 2 * b, tb = ta + 1;      // tap of lanes 0-31 / 32-63
                    constexpr int tac = ta < 9 ? ta : 8, tbc = tb < 9 ? tb : 8;
                    const int aoff = ((hh ? tbc : tac) * 4 + 3) * 64 * 16;
                    const int boff = 3 * PLANE + (hh ? ((tbc / 3) * IW + tbc % 3) : ((tac / 3) * IW + tac % 3)) * 16;
                    const bool dead = hh ? tb > 8 : ta > 8;
                    const u32x4 z = mk_u4(0, 0, 0, 0);
_Pragma("unroll")
                    for (int m = 0; m < MH; ++m) { ab[b][m] = *reinterpret_cast<const u32x4*>(ldsA + aoff + m * 32 * 16); if (dead) ab[b][m] = z; }
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) { bb[b][q] = *reinterpret_cast<const u32x4*>(ldsB + boff + q * IW * 16); if (dead) bb[b][q] = z; }
                });
_Pragma("unroll")
                for (int m = 0; m < MH; ++m)
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) wsu_mfma_f8x2(ab[0][m], ab[1][m], bb[0][q], bb[1][q], WSU_F8_SCALE_WLO, WSU_F8_SCALE_X, acc[m][q]);
            };
            WSU_STATIC_FOR(3, g, {
                cross1(std::integral_constant<int, g>{});
                WSU_STATIC_FOR(4, i, { if constexpr (4 * g + i < 9) main_term(std::integral_constant<int, 4 * g + i>{}, all_m); });
            });
        }
        };
        // ---- epilogue of the tile: accumulators -> planar global memory ---------------------------------------------------------
        auto finish_tile = [&]() __attribute__((always_inline)) {
            // (HC = 4, the four-plane head: opaque per-tile copies of the lane coordinates -- hoisted out of the tile loop, what the epilogue derives
            // from them pushed this variant two registers past its 168-register step; the other variants are untouched)
            const int l31 = opaque_if<HC == 4>(l31_k), hh = opaque_if<HC == 4>(hh_k);
            const int col = cur.x0 + l31;
            const size_t hw = (size_t)a.h * a.w;
            const int nco = a.cout >> 4;                                       // output chunks
            float hz[2][HC > 0 ? HC : 1];
            float vmax = 0.f;                                                  // largest stored activation of this lane (range flag)
            const float relu_floor = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int o = 0; o < (HC > 0 ? HC : 1); ++o) hz[q][o] = 0.f;
            auto epi_m = [&](auto m_c) __attribute__((always_inline)) {   // one accumulator tile along the output channels: 32 channels x this wave's 2 x 32 pixels
                constexpr int m = decltype(m_c)::value;
                auto piece = [&](auto cp_c) __attribute__((always_inline)) {   // 16 output channels = accumulator groups g4 = 2cp, 2cp+1
                    constexpr int cp = decltype(cp_c)::value;
                    const int oc = cur.cb * 4 + (m + cur.mh) * 2 + cp;
                    const int co0 = oc * 16 + 4 * hh;                          // this lane: channels co0..co0+3 (X) and co0+8..co0+11 (Y)
                    f32x4 vx[2], vy[2];
                    if constexpr (GRAD) {
                        const bool d1 = oc < a.nco1;                           // wave-uniform
                        const bool masked = d1 ? (a.mask != nullptr || a.mbits != nullptr) : (a.mask2 != nullptr || a.mbits2 != nullptr);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            unsigned mx = 0xFu, my = 0xFu;
                            if (masked) {
                                const int pix = (2 * wv + q) * 32 + l31;
                                mx = s_mask[((m * 2 + cp) * 2 + 0) * 512 + pix] >> (4 * hh);
                                my = s_mask[((m * 2 + cp) * 2 + 1) * 512 + pix] >> (4 * hh);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                vx[q][e] = ((mx >> e) & 1u) ? acc[m][q][8 * cp + e] : 0.f;
                                vy[q][e] = ((my >> e) & 1u) ? acc[m][q][8 * cp + 4 + e] : 0.f;
                            }
                        }
                    } else {
                    const f32x4 bx = *reinterpret_cast<const f32x4*>(s_bias + co0), by = *reinterpret_cast<const f32x4*>(s_bias + co0 + 8);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // ReLU as one max against a wave-uniform floor (0 or -inf): no select per value
                            const float x = fmaxf(acc[m][q][8 * cp + e] + bx[e], relu_floor), y = fmaxf(acc[m][q][8 * cp + 4 + e] + by[e], relu_floor);
                            vx[q][e] = x; vy[q][e] = y;
                            vmax = fmaxf(vmax, fmaxf(fabsf(x), fabsf(y)));
                        }
                    }
                    }
                    if constexpr (HEAD) {
                        const int lc = (m * 2 + cp) * 16 + 4 * hh;             // channel inside the 64-wide block
#pragma unroll
                        for (int o = 0; o < HC; ++o)
                            if (o < a.head_cout)
#pragma unroll
                                for (int q = 0; q < 2; ++q)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        hz[q][o] = fmaf(vx[q][e], s_hw[o * 64 + lc + e], fmaf(vy[q][e], s_hw[o * 64 + lc + 8 + e], hz[q][o]));
                    }
                    // addresses = wave-uniform 64-bit base (image, output chunk) + 32-bit lane offset (pixel, plane): the stores take the
                    // SGPR-base form and the epilogue carries no 64-bit address registers (it sits at the 168-register step)
                    auto store_px = [&](const f32x4& X, const f32x4& Y, char* base, uint32_t off, uint32_t plane_bytes, bool ok, unsigned char* mdst = nullptr, bool have = true) __attribute__((always_inline)) {
                        uint32_t xh0, xh1, xlo, yh0, yh1, ylo;
                        wsu_split4_f16r8(X, GRAD ? WSU_F8_GLO_DIV : WSU_F8_XLO_DIV, xh0, xh1, xlo);
                        wsu_split4_f16r8(Y, GRAD ? WSU_F8_GLO_DIV : WSU_F8_XLO_DIV, yh0, yh1, ylo);
                        swap32(xh0, yh0); swap32(xh1, yh1);                     // lanes 0-31: f16 ch 0-7, lanes 32-63: f16 ch 8-15
                        uint32_t xlp = xlo, ylp = ylo;
                        swap32(xlo, xlp); swap32(ylo, ylp);                     // lanes 0-31: xlp / ylp = the partner lane's residuals (ch 4-7 / 12-15)
                        if (ok) {
                            // (one 32-bit lane offset per store: written `base + off + hh * plane_bytes` the loop-invariant 64-bit `hh * plane_bytes` stayed live
                            // across the tile loop, was spilled, and every re-load drained the epilogue's stores with an s_waitcnt vmcnt(0))
                            uint32_t hoff = hh ? plane_bytes : 0u;
                            asm volatile("" : "+v"(hoff));                     // (kept out of the 64-bit address arithmetic)
                            *reinterpret_cast<u32x4*>(base + (uint32_t)(off + hoff)) = mk_u4(xh0, xh1, yh0, yh1);
                            if constexpr (!HONLY) {                                   // (HONLY = products F16: gradient tensors carry no residual plane)
                                if (!hh) *reinterpret_cast<u32x4*>(base + (uint32_t)(off + 2u * plane_bytes)) = mk_u4(xlo, xlp, ylo, ylp);
                            }
                            if constexpr (!GRAD && !HEAD && !POOL) {            // the training forward's ReLU-mask byte of this lane's granule
                                if (mdst) *mdst = (unsigned char)wsu_f16x8_pos_bits(mk_u4(xh0, xh1, yh0, yh1));
                            }
                        }
                    };
                    if constexpr (GRAD) {
                        const bool d1 = oc < a.nco1;
                        char* base = d1 ? a.y + (((size_t)cur.n * a.nco1 + oc) * HBM_PLANES) * hw * 16
                                        : a.y2 + (((size_t)cur.n * (nco - a.nco1) + (oc - a.nco1)) * HBM_PLANES) * hw * 16;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int row = cur.y0 + 2 * wv + q;
                            store_px(vx[q], vy[q], base, (uint32_t)(row * a.w + col) * 16u, (uint32_t)hw * 16u, row < a.h && col < a.w);
                        }
                    } else
                    if (a.y) {
                        WSU_STATIC_FOR(2, q, {
                            const int row = cur.y0 + 2 * wv + q;
                            char* base = a.y + (((size_t)cur.n * nco + oc) * HBM_PLANES) * hw * 16;
                            unsigned char* mdst = nullptr;
                            if constexpr (!GRAD && !HEAD && !POOL) {
                                if (a.relu_mask_out)
                                    mdst = a.relu_mask_out + (((size_t)cur.n * (nco * 2) + oc * 2 + hh) * wsu_mask_hp(a.h) + row) * wsu_mask_wp(a.w) + col;
                            }
                            store_px(vx[q], vy[q], base, (uint32_t)(row * a.w + col) * 16u, (uint32_t)hw * 16u, row < a.h && col < a.w, mdst, a.y != nullptr);
                        });
                    }
                    if constexpr (POOL) {                                      // every lane takes part in the exchanges
                        f32x4 px, py;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // 2x2 window = this lane's two rows x the lane pair (l, l ^ 1): one max down the column, one across the pair (the
                            // neighbour arrives as a DPP operand).  Round 2 spent ~14 instructions per pooled value on a NaN-propagating
                            // compare / select chain; values reaching this point went through v_max(x, floor), which never returns NaN
                            // for floor = 0 (ReLU) -- with relu = 0 a NaN in one window element is dropped like IEEE maxNum drops it.
                            const float cx = fmaxf(vx[0][e], vx[1][e]), cy = fmaxf(vy[0][e], vy[1][e]);
                            px[e] = fmaxf(cx, dpp_xor1(cx)); py[e] = fmaxf(cy, dpp_xor1(cy));
                        }
                        const int hp = a.h >> 1, wp2 = a.w >> 1;
                        const int gy = (cur.y0 >> 1) + wv, gx = (cur.x0 >> 1) + (l31 >> 1);
                        char* base = a.ypool + (((size_t)cur.n * nco + oc) * HBM_PLANES) * hp * wp2 * 16;
                        store_px(px, py, base, (uint32_t)(gy * wp2 + gx) * 16u, (uint32_t)(hp * wp2) * 16u, !(l31 & 1) && gy < hp && gx < wp2);
                    }
                };
                piece(std::integral_constant<int, 0>{});
                piece(std::integral_constant<int, 1>{});
            };
            WSU_STATIC_FOR(MH, m, { (void)m; epi_m(m_c); });
            if constexpr (HEAD) {
                // the other 32 channels of this pixel sit in the partner lane (lane ^ 32)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = cur.y0 + 2 * wv + q;
#pragma unroll
                    for (int o = 0; o < HC; ++o)
                        if (o < a.head_cout) {
                            uint32_t mine = __builtin_bit_cast(uint32_t, hz[q][o]), other = mine;
                            swap32(mine, other);                            // lanes 0-31: other = partner's sum; lanes 32-63: mine = partner's
                            const float z = __builtin_bit_cast(float, mine) + __builtin_bit_cast(float, other) + s_hb[o];
                            if (!hh && row < a.h && col < a.w) {
                                const size_t plane = ((size_t)cur.n * a.head_cout + o) * hw;      // wave-uniform
                                const uint32_t off = (uint32_t)(row * a.w + col);
                                if (a.head_logit) (a.head_logit + plane)[off] = z;
                                (a.head_out + plane)[off] = 1.f / (1.f + expf(-z));
                            }
                        }
                }
            }
            // beyond +-448 the e4m3 residual saturates (plain f16 accuracy), beyond +-65504 the f16 part overflows: tell the caller once
            if (!GRAD && a.range_flag && (a.y || POOL) && __builtin_amdgcn_ballot_w64(!(vmax <= WSU_F8_RANGE)) != 0 && lane == 0)
                atomicOr(a.range_flag, 1u);
            ++kt;
            c = 0;
            if (j + 1 < J) cur = tile_of(a, lw + kt * G);
#ifdef WSU_PL_STAMPS
            t_epi += __builtin_amdgcn_s_memtime() - s4;
#endif
        };
        {
            for (; j < J; ++j) {
                begin_step();
                if (c == 0) zero_acc();
                units_all();
                STAMP(s4);
                t_wait += s1 - s0; t_bar += s2 - s1; t_dma += s3 - s2; t_mma += s4 - s3;
                if (c + 1 == a.nch && (a.ablate & 2)) {                        // timing only: the tile's results are dropped (kept alive for the compiler)
#pragma unroll
                    for (int m = 0; m < MH; ++m)
#pragma unroll
                        for (int q = 0; q < 2; ++q) asm volatile("" :: "v"(acc[m][q]));
                    ++kt; c = 0;
                    if (j + 1 < J) cur = tile_of(a, lw + kt * G);
                } else if (c + 1 == a.nch) {
                    finish_tile();
                } else {
                    ++c;
                }
            }
        }
    }
#ifdef WSU_PL_STAMPS
    if (lane == 0 && blockIdx.x >= 64 && blockIdx.x < 128) g_pl_stamps[(blockIdx.x * 2) * 8 + wv] = t_bar;          // per-wave barrier waits / section times
    if (lane == 0 && blockIdx.x >= 32 && blockIdx.x < 64) g_pl_stamps[(blockIdx.x * 2) * 8 + wv] = t_mma;
    if (tid == 64 && blockIdx.x < 32) {                                        // matrix wave 1's view
        unsigned long long* d = g_pl_stamps + (blockIdx.x * 2) * 8;
        d[0] = __builtin_amdgcn_s_memtime() - t0; d[1] = __builtin_amdgcn_s_memrealtime() - rt0;
        d[2] = t_wait; d[3] = t_bar; d[4] = t_dma; d[5] = t_mma; d[6] = t_epi; d[7] = (unsigned long long)J;
    }
#endif
}

// one place that knows the instantiations: attributes once, then the variant the arguments select
int pl_launch(PlArgs a, bool first, hipStream_t s, bool grad = false) {
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("WSU_PL_ABLATE"); ablate = e ? atoi(e) : 0; }
    a.ablate = ablate;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            wsu_set_error("conv3x3_pl: cannot query the device"); return WSU_ERR_HIP;
        }
        const void* fns[9] = {reinterpret_cast<const void*>(&conv3x3_pl_kernel<0, false, false, false, true, false, true>), reinterpret_cast<const void*>(&conv3x3_pl_kernel<0, false, true, false, true>),reinterpret_cast<const void*>(&conv3x3_pl_kernel<0, false>), reinterpret_cast<const void*>(&conv3x3_pl_kernel<0, true>),
                              reinterpret_cast<const void*>(&conv3x3_pl_kernel<1, false>), reinterpret_cast<const void*>(&conv3x3_pl_kernel<4, false>),
                              reinterpret_cast<const void*>(&conv3x3_pl_kernel<0, false, false>),
                              reinterpret_cast<const void*>(&conv3x3_pl_kernel<0, false, true, true>), reinterpret_cast<const void*>(&conv3x3_pl_kernel<0, true, true, true>)};
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pl_kernel<0, false, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e0 != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3_pl): %s", hipGetErrorString(e0)); return WSU_ERR_HIP; }
        for (const void* fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
            if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3_pl): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        }
        ncu = prop.multiProcessorCount;
    }
    // small grids: the plain variant splits every tile's 64 output channels over two work items when that still fits the CUs
    static int msplit_on = -1;
    if (msplit_on < 0) { const char* e = getenv("WSU_PL_MSPLIT"); msplit_on = e ? atoi(e) : 1; }
    a.msplit = 0;
    if (msplit_on && !grad && !first && !a.head_w && !a.ypool && a.xres && 2 * (long long)a.ntiles <= ncu) {
        a.msplit = 1; a.ncb *= 2; a.ntiles *= 2;
        hipLaunchKernelGGL((conv3x3_pl_kernel<0, false, true, false, false, true>), dim3(a.ntiles), dim3(NT), LDS_TOTAL, s, a);
        return wsu_check_launch("conv3x3_pl_kernel");
    }
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    const dim3 g(grid), b(NT);
    if (grad && a.honly) hipLaunchKernelGGL((conv3x3_pl_kernel<0, false, false, false, true, false, true>), g, b, LDS_TOTAL, s, a);
    else if (grad) hipLaunchKernelGGL((conv3x3_pl_kernel<0, false, true, false, true>), g, b, LDS_TOTAL, s, a);
    else if (first) {
        if (a.ypool) hipLaunchKernelGGL((conv3x3_pl_kernel<0, true, true, true>), g, b, LDS_TOTAL, s, a);
        else hipLaunchKernelGGL((conv3x3_pl_kernel<0, false, true, true>), g, b, LDS_TOTAL, s, a);
    } else if (a.head_w && a.head_cout == 1) hipLaunchKernelGGL((conv3x3_pl_kernel<1, false>), g, b, LDS_TOTAL, s, a);
    else if (a.head_w) hipLaunchKernelGGL((conv3x3_pl_kernel<4, false>), g, b, LDS_TOTAL, s, a);
    else if (a.ypool) hipLaunchKernelGGL((conv3x3_pl_kernel<0, true>), g, b, LDS_TOTAL, s, a);
    else if (!a.xres) hipLaunchKernelGGL((conv3x3_pl_kernel<0, false, false>), g, b, LDS_TOTAL, s, a);
    else hipLaunchKernelGGL((conv3x3_pl_kernel<0, false>), g, b, LDS_TOTAL, s, a);
    return wsu_check_launch("conv3x3_pl_kernel");
}

}  // namespace

extern "C" {

// diagnostic only (rows alternate: matrix wave 1, loader wave 8 of workgroups 0..127) (not part of include/wsu.h): phase stamps of the last conv3x3_pl launch with WSU_PL_STAMP=1
int wsu_debug_read_pl_stamps(unsigned long long* host_dst, int nblocks) {
    if (nblocks > 256) nblocks = 256;
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_pl_stamps), (size_t)nblocks * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}

size_t wsu_relu_mask_bytes(int n, int c, int h, int w) {
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)n * (size_t)((c + 7) / 8) * (size_t)wsu_mask_hp(h) * (size_t)wsu_mask_wp(w);
}

// Forward 3x3 reflect conv + bias + ReLU on planar F16F8P activations (layout: wsu.h).  x1 (c1 channels) and optional x2 (c2, fused
// concat), packed weights of wsu_conv3x3_pack(mode F16F8); range_flag (optional device word): bit 0 is OR-ed in when a stored value leaves
// the format's full-accuracy range; outputs, each optional: y (cout channels, planar), y_pool (2x2 max-pooled,
// planar), head (1x1 conv + sigmoid on the 64 output channels: out / logit NCHW fp32; needs cout == 64).  c1, c2 multiples of 16,
// cout of 64.  Asynchronous on `stream`; allocates nothing.
int wsu_conv3x3_pl_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias, void* y, void* y_pool,
                       const float* head_w, const float* head_b, float* head_out, float* head_logit, int head_cout,
                       int n, int h, int w, int c1, int c2, int cout, int relu, int x_residual, unsigned* range_flag,
                       unsigned char* relu_mask_out, void* stream) {
    WSU_REQUIRE(x1 && w_packed && (y || y_pool || head_w), "conv3x3_pl: null pointer");
    WSU_REQUIRE(!relu_mask_out || (y && !y_pool && !head_w && x_residual), "conv3x3_pl: relu_mask_out needs the plain variant (y only)");
    WSU_REQUIRE(!relu_mask_out || (long long)n * (cout / 8) * wsu_mask_hp(h) * wsu_mask_wp(w) < 0x7FFFFFF0LL, "conv3x3_pl: mask plane too large");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_pl: bad shape n=%d h=%d w=%d (reflect pad 1 needs h,w >= 2)", n, h, w);
    WSU_REQUIRE(c1 > 0 && c1 % 16 == 0 && c2 >= 0 && c2 % 16 == 0 && (c2 == 0) == (x2 == nullptr), "conv3x3_pl: c1=%d c2=%d must be multiples of 16", c1, c2);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0 && cout <= 1024, "conv3x3_pl: cout=%d must be a multiple of %d (<= 1024)", cout, WSU_COB);
    WSU_REQUIRE(!head_w || (head_out && cout == WSU_COB && head_cout >= 1 && head_cout <= 4), "conv3x3_pl: fused head needs cout == %d and 1..4 head planes", WSU_COB);
    WSU_REQUIRE(!y_pool || (h % 2 == 0 && w % 2 == 0), "conv3x3_pl: fused pool needs even h, w");
    WSU_REQUIRE(!(y_pool && head_w), "conv3x3_pl: the fused pool and the fused head exclude each other");
    WSU_REQUIRE(x_residual || (!y_pool && !head_w), "conv3x3_pl: x_residual = 0 is built for the plain variant (no fused pool / head)");
    WSU_REQUIRE((long long)h * w * 48 < 0xFFFFFFF0LL, "conv3x3_pl: h*w too large (a plane triple must stay below 4 GiB)");
    PlArgs a;
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.wp = (const char*)w_packed; a.bias = bias;
    a.y = (char*)y; a.ypool = (char*)y_pool;
    a.head_w = head_w; a.head_b = head_b; a.head_out = head_out; a.head_logit = head_logit; a.head_cout = head_cout;
    a.range_flag = range_flag; a.xres = x_residual ? 1 : 0;
    WSU_REQUIRE(x_residual == 0 || x_residual == 1, "conv3x3_pl: x_residual must be 0 (one e4m3 cross term) or 1 (both); the block-scaled fp4 cross terms "
                "(x_residual = 2 of round 3) moved to wsu_conv3x3_q_fwd on planar Q tensors");
    a.n = n; a.h = h; a.w = w; a.c1 = c1; a.c2 = c2; a.cout = cout;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ncb = cout / WSU_COB;
    a.nch1 = c1 / 16; a.nch = (c1 + c2) / 16; a.relu = relu;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "conv3x3_pl: %lld tiles out of range", nt);
    a.ntiles = (int)nt;
    a.img = nullptr; a.w1 = nullptr; a.b1 = nullptr;
    a.y2 = nullptr; a.nco1 = cout / 16; a.mask = nullptr; a.mask2 = nullptr; a.imgs_per_wset = 0; a.wset_bytes = 0;
    a.relu_mask_out = relu_mask_out; a.mbits = nullptr; a.mbits2 = nullptr; a.honly = 0;
    return pl_launch(a, false, static_cast<hipStream_t>(stream));
}

// e11 + e12 (+pool) of the planar path in one launch (unet.py:141-144, single-plane inputs): img (N,1,H,W) fp32, w1 (64,1,3,3), b1 (64) or NULL;
// the loader waves compute the 64 input channels of the 3x3 conv into the LDS stages (kernel variant F1), xe11 never reaches HBM.  Bitwise the
// result of wsu_conv3x3_first_pl_fwd followed by wsu_conv3x3_pl_fwd.  w_packed / bias: the second conv (cin = 64); y / y_pool as there.
int wsu_conv3x3_pl_fused_first_fwd(const float* img, const float* w1, const float* b1, const void* w_packed, const float* bias,
                                   void* y, void* y_pool, int n, int h, int w, int cout, int relu, unsigned* range_flag, void* stream) {
    WSU_REQUIRE(img && w1 && w_packed && (y || y_pool), "conv3x3_pl_fused_first: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_pl_fused_first: bad shape n=%d h=%d w=%d (reflect pad 1 needs h,w >= 2)", n, h, w);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0 && cout <= 1024, "conv3x3_pl_fused_first: cout=%d must be a multiple of %d (<= 1024)", cout, WSU_COB);
    WSU_REQUIRE(!y_pool || (h % 2 == 0 && w % 2 == 0), "conv3x3_pl_fused_first: fused pool needs even h, w");
    WSU_REQUIRE((long long)h * w * 48 < 0xFFFFFFF0LL, "conv3x3_pl_fused_first: h*w too large (a plane triple must stay below 4 GiB)");
    PlArgs a;
    a.x1 = nullptr; a.x2 = nullptr; a.wp = (const char*)w_packed; a.bias = bias;
    a.y = (char*)y; a.ypool = (char*)y_pool;
    a.head_w = nullptr; a.head_b = nullptr; a.head_out = nullptr; a.head_logit = nullptr; a.head_cout = 0;
    a.range_flag = range_flag; a.xres = 1;
    a.img = img; a.w1 = w1; a.b1 = b1;
    a.n = n; a.h = h; a.w = w; a.c1 = 64; a.c2 = 0; a.cout = cout;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ncb = cout / WSU_COB;
    a.nch1 = 4; a.nch = 4; a.relu = relu;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "conv3x3_pl_fused_first: %lld tiles out of range", nt);
    a.ntiles = (int)nt;
    a.y2 = nullptr; a.nco1 = cout / 16; a.mask = nullptr; a.mask2 = nullptr; a.imgs_per_wset = 0; a.wset_bytes = 0;
    a.relu_mask_out = nullptr; a.mbits = nullptr; a.mbits2 = nullptr; a.honly = 0;
    return pl_launch(a, true, static_cast<hipStream_t>(stream));
}

// K7p: data gradient of the 3x3 reflect conv on planar tensors (autograd of unet.py:141-189).  g: the pre-activation gradient, cout channels,
// planar with the gradient encodings (power-of-two scaled, model/autograd.py); w_packed_dgrad from wsu_conv3x3_pack_dgrad(mode F16F8);
// dx1 gets input channels [0, csplit), dx2 (optional, fused concat) the rest; mask1 / mask2 (optional): planar ACTIVATIONS shaped like dx1 /
// dx2 whose sign is the ReLU mask of the layer that produced that input (relu'(0) = 0).  pad_zero = 0: reflect padding -- the border ring of
// the adjoint runs as one more launch over strips of g's border rows / columns with the weight sets of wsu_conv3x3_pack_ring (w_packed_ring)
// in `workspace` (wsu_conv3x3_pl_bwd_data_workspace_bytes); pad_zero = 1: zero padding (ring arguments unused).
size_t wsu_conv3x3_pl_bwd_data_workspace_bytes(int n, int h, int w, int cin, int cout) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
    const size_t L = (size_t)(h > w ? h : w);
    return (size_t)4 * n * (L + 2) * 3 * (size_t)(cin + cout);
}

int wsu_ring_gather_pl(const void* g, void* strips, int n, int h, int w, int c, int L, void* stream);
int wsu_ring_fold_pl(const void* strips_out, void* dx1, void* dx2, const void* mask1, const void* mask2,
                     int n, int h, int w, int cin, int csplit, int L, int gres, void* stream);

int wsu_conv3x3_pl_bwd_data(const void* g, const void* w_packed_dgrad, const void* w_packed_ring, void* workspace, size_t workspace_bytes,
                            void* dx1, void* dx2, int csplit, const void* mask1, const void* mask2,
                            const unsigned char* mask1_bits, const unsigned char* mask2_bits,
                            int n, int h, int w, int cin, int cout, int pad_zero, int products, void* stream) {
    WSU_REQUIRE(g && w_packed_dgrad && dx1, "conv3x3_pl_bwd_data: null pointer");
    WSU_REQUIRE(products == WSU_PRODUCTS_F16F8 || products == WSU_PRODUCTS_F16, "conv3x3_pl_bwd_data: products must be WSU_PRODUCTS_F16F8 or WSU_PRODUCTS_F16");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_pl_bwd_data: bad shape n=%d h=%d w=%d", n, h, w);
    WSU_REQUIRE(cout >= 32 && cout % 16 == 0, "conv3x3_pl_bwd_data: cout=%d must be a multiple of 16 (>= 32)", cout);
    WSU_REQUIRE(cin > 0 && cin % WSU_COB == 0 && cin <= 1024, "conv3x3_pl_bwd_data: cin=%d must be a multiple of %d (<= 1024)", cin, WSU_COB);
    WSU_REQUIRE(csplit > 0 && csplit <= cin && csplit % WSU_COB == 0 && (csplit < cin) == (dx2 != nullptr), "conv3x3_pl_bwd_data: csplit=%d (cin=%d) must be a multiple of %d, dx2 given iff csplit < cin", csplit, cin, WSU_COB);
    WSU_REQUIRE(!mask2 || dx2, "conv3x3_pl_bwd_data: mask2 without dx2");
    WSU_REQUIRE((!mask1_bits || mask1) && (!mask2_bits || mask2), "conv3x3_pl_bwd_data: the 1-bit masks come WITH the activations they were taken from (the border fold reads those)");
    WSU_REQUIRE(!(mask1_bits || mask2_bits) || (long long)n * (cin / 8) * wsu_mask_hp(h) * wsu_mask_wp(w) < 0x7FFFFFF0LL, "conv3x3_pl_bwd_data: mask plane too large");
    WSU_REQUIRE((long long)h * w * 48 < 0xFFFFFFF0LL && (long long)n * ((h > w ? h : w) + 2) * 48 < 0xFFFFFFF0LL, "conv3x3_pl_bwd_data: h*w too large (a plane triple must stay below 4 GiB)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    PlArgs a;
    a.x1 = (const char*)g; a.x2 = nullptr; a.wp = (const char*)w_packed_dgrad; a.bias = nullptr;
    a.y = (char*)dx1; a.ypool = nullptr; a.y2 = (char*)dx2; a.nco1 = csplit / 16; a.mask = (const char*)mask1; a.mask2 = (const char*)mask2;
    a.head_w = nullptr; a.head_b = nullptr; a.head_out = nullptr; a.head_logit = nullptr; a.head_cout = 0;
    a.range_flag = nullptr; a.xres = 1; a.img = nullptr; a.w1 = nullptr; a.b1 = nullptr;
    a.imgs_per_wset = 0; a.wset_bytes = 0;
    a.relu_mask_out = nullptr; a.mbits = mask1_bits; a.mbits2 = mask2_bits; a.honly = products == WSU_PRODUCTS_F16 ? 1 : 0;
    a.n = n; a.h = h; a.w = w; a.c1 = cout; a.c2 = 0; a.cout = cin;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ncb = cin / WSU_COB;
    a.nch1 = cout / 16; a.nch = cout / 16; a.relu = 0;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "conv3x3_pl_bwd_data: %lld tiles out of range", nt);
    a.ntiles = (int)nt;
    int rc = pl_launch(a, false, s, true);
    if (rc || pad_zero) return rc;
    // ---- reflect adjoint: what the padded border ring folds back (train_pl.hip) ----
    WSU_REQUIRE(w_packed_ring && workspace, "conv3x3_pl_bwd_data: reflect padding needs the ring weights and a workspace");
    WSU_REQUIRE(workspace_bytes >= wsu_conv3x3_pl_bwd_data_workspace_bytes(n, h, w, cin, cout), "conv3x3_pl_bwd_data: workspace of %zu bytes too small", workspace_bytes);
    const int L = h > w ? h : w;
    char* strips_in = (char*)workspace;
    char* strips_out = strips_in + (size_t)4 * n * (L + 2) * 3 * cout;
    rc = wsu_ring_gather_pl(g, strips_in, n, h, w, cout, L, stream);
    if (rc) return rc;
    PlArgs r = a;
    r.x1 = strips_in; r.wp = (const char*)w_packed_ring; r.y = strips_out; r.y2 = nullptr; r.nco1 = cin / 16; r.mask = nullptr; r.mask2 = nullptr;
    r.mbits = nullptr; r.mbits2 = nullptr;
    r.imgs_per_wset = 1; r.wset_bytes = (size_t)cin * cout * 9 * 4;
    r.n = 4; r.h = n; r.w = L + 2;
    r.tiles_x = (r.w + TW - 1) / TW; r.tiles_y = (r.h + TH - 1) / TH;
    r.ntiles = 4 * r.tiles_x * r.tiles_y * r.ncb;
    rc = pl_launch(r, false, s, true);
    if (rc) return rc;
    return wsu_ring_fold_pl(strips_out, dx1, dx2, mask1, mask2, n, h, w, cin, csplit, L, a.honly ? 0 : 1, stream);
}

}  // extern "C"
