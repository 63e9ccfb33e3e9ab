// K1q (round 4): the 3x3 convolution of the DEFAULT inference mode 'f16f4p' on planar Q tensors -- the f16 products on the f16 matrix pipe, both
// residual cross terms of a tap pair as ONE block-scaled fp4 (e2m1) instruction (the arithmetic of round 3's Q4 variant of conv3x3_pl.hip) -- with
// the operands stored the way they are multiplied:
//   * a planar Q tensor ("F16F4P" storage, include/wsu.h) holds per (pixel, 16-channel chunk) the two f16 granules, the 16-byte Q granule
//     (fp4 of the f16 parts | fp4 of the residuals * 2^11, both divided by the block's power-of-two scale) and the E8M0 scale byte -- written by
//     the PRODUCING epilogue from its fp32 values (wsu_device.h: wsu_q4_pre / wsu_q4_pair).  Round 3 stored an e4m3 residual plane and every
//     consumer's loader waves re-derived the Q granule and the scale from it (conversion VALU on the SIMDs of the matrix waves, residual
//     granules through registers, a three-slot ring to give the conversion a step of time);
//   * so the four loader waves are PURE DMA again: per step 40 input pieces (three 16-byte planes by `buffer_load_dwordx4 ... lds`, the scale
//     bytes by `buffer_load_ubyte ... lds`, one dword slot per pixel) + 28 weight pieces, no vector ALU, no registers beyond the per-tile offsets;
//   * RQ = 2 (default): 8 matrix waves x (64 co x 2 rows x 32 px, 4 accumulator tiles), two per SIMD, as in conv3x3_pl.hip.  RQ = 4
//     (WSU_Q_ROWS=4, VERDICT r03 next #3a): ONE matrix wave per SIMD -- 4 waves x (64 co x 4 rows x 32 px, 8 accumulator tiles = 128 of the 256
//     registers a wave may use at two waves per SIMD): no partner arbitration on the matrix pipe, 6 instead of 4 fragment reads per 8 instead of
//     4 matrix instructions, the fragment reads of unit u + 1 issued before the matrix instructions of unit u.  Measured 4 % SLOWER on one box
//     (6 % before the explicit fragment pipeline; profiles/r04/ab_forward_organisations.md) -- kept as an experiment switch, same bits.
// Same persistent structure as conv3x3_pl.hip: one workgroup per CU walks 16 x 32-pixel x 64-co tiles; inputs travel two steps ahead into a ring
// of three slots, weights one step ahead into two; ONE s_barrier per chunk step; epilogue straight from the accumulators (bias, ReLU, fused 2x2
// max-pool, fused 1x1 head + sigmoid), the last step of a tile split by accumulator tile so that its second half overlaps the first half's stores.
// Replaces nn.Conv2d(k=3, reflect) + F.relu (+ torch.cat, nn.MaxPool2d, outconv + sigmoid) of src/unet/model/unet.py:141-189.
// Weights: wsu_conv3x3_pack_f4 (below).
#include "wsu_device.h"
#include <cstdlib>

#ifndef WSU_Q_EPO
#define WSU_Q_EPO 1             // 1 = the last step of a tile is split by accumulator tile and its second half shares a basic block with the first half of the epilogue
#endif
#ifndef WSU_Q_PIPE2
#define WSU_Q_PIPE2 0           // experiment: 1 = the eight-wave organisation (RQ = 2) also issues the fragment reads of unit u + 1 before the matrix instructions of unit u
#endif
#ifndef WSU_Q_PIPE_DEPTH
#define WSU_Q_PIPE_DEPTH 1      // units of fragment reads in flight in front of the matrix instructions that use them (explicit pipeline)
#endif
#ifndef WSU_Q_PROBE_NOX8
#define WSU_Q_PROBE_NOX8 0      // timing-only probe (results wrong; make qnox8): 1 = the half-empty fp4 instruction of the unpaired ninth tap is skipped
#endif                          // (profiles/r04/ab_forward_organisations.md, third A/B: the time follows the ENERGY of the useful products)
#ifndef WSU_Q_EPO_FENCE
#define WSU_Q_EPO_FENCE 1       // 1 = one scheduling region per hook of that block (RQ = 4: without the fences the scheduler hoists the fragment reads of all five tap
#endif                          //     pairs above the epilogue and spills 60-70 registers)

namespace {

constexpr int TW = 32, TH = 16, IW = TW + 2, IH = TH + 2;
constexpr int NPIX = IW * IH;                             // 612 input-tile pixels
constexpr int PLANE = NPIX * 16;                          // 9792 B per granule plane
constexpr int IN_SEG = (NPIX + 63) / 64;                  // 10 wave-instructions per plane (the last one 36 lanes wide)
constexpr int S_LDS = IN_SEG * 256;                       // scale bytes: one DWORD slot per pixel (LDS-DMA writes lane * 4), 2560 B
constexpr int IN_SLOT = 3 * PLANE + S_LDS;                // f16 ch 0-7 | f16 ch 8-15 | Q | S = 31936
constexpr int W_GRAN = 9 * 3 * WSU_COB * 16;              // 27648
constexpr int W_SLOT = W_GRAN + 1024;                     // + [9][64] scale bytes, padded to a DMA piece: the packed slice of a (block, chunk)
constexpr int NIN = 3, NWS = 2;
constexpr int W_BASE = NIN * IN_SLOT;                     // 95808
constexpr int LDS_EXTRA = W_BASE + NWS * W_SLOT;          // 153152: bias [1024] | head_w [4][64] | head_b [4]
constexpr int LDS_TOTAL = LDS_EXTRA + 1024 * 4 + 4 * 64 * 4 + 16;     // 158288
static_assert(LDS_TOTAL <= 160 * 1024 && W_SLOT % 1024 == 0, "LDS budget");
constexpr int NLOAD = 4;
constexpr int W_PIECES = W_SLOT / 1024;                   // 28
constexpr unsigned OOB = 0xFFFFFFF0u;                     // beyond any descriptor: a store is dropped

struct QArgs {
    const char* x1; const char* x2; const char* wp; const float* bias;
    char* y; char* ypool;
    const float* head_w; const float* head_b; float* head_out; float* head_logit; int head_cout;
    int n, h, w, c1, c2, cout;
    int tiles_x, tiles_y, ncb, nch1, nch;
    int relu;
    int ntiles;                                           // n * tiles_y * tiles_x * ncb
    unsigned* range_flag;
    int msplit;                                           // 1: work items are half-blocks of 32 output channels (kernel variant MSPLIT); ncb = 2 * cout / 64
    int ablate;                                           // timing-only experiments (WSU_PL_ABLATE bits; results wrong when != 0): 1 = no DMA after step 0, 2 = no epilogue
    const float* img; const float* w1; const float* b1;   // fused first layer (kernel variant F1): the 64 input channels are computed by the loader waves; w1 = tap-major [9][64]
};

struct Tile { int n, y0, x0, cb, mh; };
__device__ __forceinline__ Tile tile_of(const QArgs& a, int t) {
    Tile r;
    r.cb = t % a.ncb; t /= a.ncb;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    r.n = t / a.tiles_y; r.y0 = ty * TH; r.x0 = tx * TW;
    r.mh = 0;
    if (a.msplit) { r.mh = r.cb & 1; r.cb >>= 1; }
    return r;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ float dpp_xor1(float v) {       // value of lane ^ 1 by a DPP quad permutation
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}

// ================= loader wave LW: pure DMA =================================================================================================
// Input pieces by 64-pixel segment of the 612-pixel tile: wave LW fetches segments LW, LW + 4, LW + 8 (< 10) -- four pieces each (planes 0, 1, Q:
// 16 bytes per lane, S: one byte per lane into a dword slot); waves 0 / 1 (three segments = 12 pieces) take 5 weight pieces, waves 2 / 3 (8) take
// 9: 17 DMA instructions per wave and step.  Issue order of a wave: W(0) IN(0) IN(1) | barrier 0 | W(1) IN(2) | barrier 1 | W(2) IN(3) | ...;
// `s_waitcnt vmcnt(n)` = all but the n youngest have landed, so before barrier j + 1 the wave waits for everything but IN(j + 2).
template <int LW>
__device__ __forceinline__ void q_loader(const QArgs& a, char* smem, int lane, int lw, int G, int J) {
    constexpr int NSEG = LW < 2 ? 3 : 2;
    constexpr int NIN_OPS = 4 * NSEG;
    constexpr int W0 = LW < 2 ? LW * 5 : 10 + (LW - 2) * 9, NWP = LW < 2 ? 5 : 9;
    static_assert(2 * 5 + 2 * 9 == W_PIECES, "weight pieces over the loader waves");
    lds_char* smem3 = (lds_char*)smem;
    const unsigned hw16 = (unsigned)(a.h * a.w) * 16u;
    const unsigned chunk_bytes = (unsigned)wsu_q_chunk_bytes(a.h, a.w);
    unsigned voff[NSEG], soff[NSEG];
    auto plan = [&](const Tile& t) __attribute__((always_inline)) {
        WSU_STATIC_FOR(NSEG, k, {
            const int idx = min((LW + NLOAD * k) * 64 + lane, NPIX - 1);
            const int r = idx / IW, c = idx - r * IW;
            const int yy = wsu_reflect(t.y0 - 1 + r, a.h), xx = wsu_reflect(t.x0 - 1 + c, a.w);
            voff[k] = (unsigned)(yy * a.w + xx) * 16u;
            soff[k] = 3u * hw16 + wsu_q_soff(yy, xx, a.tiles_x);
        });
    };
    Tile ti = tile_of(a, lw); int ci = 0, kti = 0;                    // cursor of the input issue
    int cbw = ti.cb, cw = 0, ktw = 0;                                 // cursor of the weight issue (block, chunk)
    auto issue_in = [&](int s) __attribute__((always_inline)) {
        const char* in_src = ci < a.nch1 ? a.x1 + ((size_t)ti.n * a.nch1 + ci) * chunk_bytes : a.x2 + ((size_t)ti.n * (a.nch - a.nch1) + (ci - a.nch1)) * chunk_bytes;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_src), 0, (int)chunk_bytes, 0x00020000);
        lds_char* slot = smem3 + (s % NIN) * IN_SLOT;
        WSU_STATIC_FOR(NSEG, k, {
            constexpr int seg = LW + NLOAD * k;
            const bool live = seg < IN_SEG - 1 || lane < NPIX - (IN_SEG - 1) * 64;
            if (live) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(slot + seg * 1024), 16, voff[k], 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(slot + PLANE + seg * 1024), 16, voff[k], (int)hw16, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(slot + 2 * PLANE + seg * 1024), 16, voff[k], (int)(2u * hw16), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(slot + 3 * PLANE + seg * 256), 1, soff[k], 0, 0, 0);
            }
        });
        if (++ci == a.nch && s + 1 < J) { ci = 0; ++kti; ti = tile_of(a, lw + kti * G); plan(ti); }
    };
    auto issue_w = [&](int s) __attribute__((always_inline)) {
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wp), 0, 0x7FFFFFF0, 0x00020000);
        const int w_base = (cbw * a.nch + cw) * W_SLOT;
        lds_char* slot = smem3 + W_BASE + (s % NWS) * W_SLOT;
        WSU_STATIC_FOR(NWP, k, {
            constexpr int piece = W0 + k;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(slot + piece * 1024), 16, (unsigned)lane * 16u, w_base + piece * 1024, 0, 0);
        });
        if (++cw == a.nch && s + 1 < J) { cw = 0; ++ktw; cbw = tile_of(a, lw + ktw * G).cb; }
    };
    if (J <= 0) return;
    plan(ti);
    issue_w(0);
    issue_in(0);
    if (J > 1) issue_in(1);
    if (J > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIN_OPS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j = 0; ; ++j) {
        __builtin_amdgcn_s_barrier();                                 // barrier j: step j is complete in LDS; every matrix wave has left step j - 1
        asm volatile("" ::: "memory");
        if (j + 1 >= J) break;
        if (!(a.ablate & 1)) {
            issue_w(j + 1);                                           // its slot held step j - 1
            const bool more = j + 2 < J;
            if (more) issue_in(j + 2);                                // its slot held step j - 1
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIN_OPS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

// ================= loader waves of the kernel variant F1 (e11 + e12 in one launch, unet.py:141-144) ==========================================
// The loaders COMPUTE the input slot of a step instead of fetching it: relu(b1 + w1 * 3x3 window of the image) for the step's 16 channels at the
// tile's 612 halo pixels -- fp32 fused multiply-adds in the tap order of first_pl_kernel, the same wsu_q4_encode16, written where the DMA would
// have put the three granule planes and the scale slot: the operands of the matrix waves are bitwise those of wsu_conv3x3_first_pl_fwd ->
// wsu_conv3x3_q_fwd, and xe11 (1.4 GB written and read back at batch 32) never exists.  Lane = up to 3 of the tile's pixels, their 27 image
// values stay in registers over the tile's 4 chunks; only the weight pieces still come by LDS-DMA.  ONE copy of the body with a run-time wave
// index (four inlined copies beside the matrix waves' code cost the e4m3 kernel of round 2 ~160 spilled registers).
__device__ __forceinline__ void q_loader_f1(const QArgs& a, char* smem, int lane, int lw, int G, int J, int lw8) {
    if (J <= 0) return;
    if (a.ablate & 16) __builtin_amdgcn_s_setprio(3);                 // experiment (WSU_Q_F1_PRIO=1): the computing loaders' instructions ahead of the matrix waves'
    lds_char* smem3 = (lds_char*)smem;
    constexpr int F1_PX = 3;                                          // segments lw8, lw8 + 4, lw8 + 8 (< 10)
    float pimg[F1_PX][9];
    float f1_max = 0.f;                                               // range flag of the computed (never stored) xe11 values
    Tile ti = tile_of(a, lw); int ci = 0, kti = 0;                    // cursor of the computed inputs
    int cbw = ti.cb, cw = 0, ktw = 0;                                 // cursor of the weight issue (block, chunk)
    auto window = [&](const Tile& tt) __attribute__((always_inline)) {
        const float* img = a.img + (size_t)tt.n * a.h * a.w;
#pragma unroll
        for (int k = 0; k < F1_PX; ++k) {
            const int idx = min((lw8 + NLOAD * k) * 64 + lane, NPIX - 1);
            const int r = idx / IW, cc = idx - r * IW;
            const int yy = wsu_reflect(tt.y0 - 1 + r, a.h), xx = wsu_reflect(tt.x0 - 1 + cc, a.w);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
                pimg[k][tp] = img[(size_t)wsu_reflect(yy + tp / 3 - 1, a.h) * a.w + wsu_reflect(xx + tp % 3 - 1, a.w)];
        }
    };
    // e11's weights and bias are wave-uniform: they come through scalar loads (constant address space: `s_load_dwordx16` per tap) straight into
    // the packed multiply-adds' scalar operand -- no LDS table, no vector loads; two channels per instruction (v_pk_fma_f32 = two independent
    // fused multiply-adds, bitwise the fmaf chain of first_pl_kernel).  Packed-f32 read-after-write hazard (Makefile): an accumulator pair is
    // re-read 24 instructions later by its own next tap, and an `s_nop 0` separates the last update from the first read of the encoding.
    typedef __attribute__((address_space(4))) const f32x2 cst_f32x2;
    auto compute_in = [&](int s) __attribute__((always_inline)) {
        char* slot = smem + (s % NIN) * IN_SLOT;
        cst_f32x2* wt = (cst_f32x2*)(a.w1 + ci * 16);                   // tap-major table [9][64]: + tap * 32 pairs
        cst_f32x2* bt = (cst_f32x2*)(a.b1 + ci * 16);
        f32x2 acc[F1_PX][8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x2 b = bt[g];
#pragma unroll
            for (int k = 0; k < F1_PX; ++k) acc[k][g] = b;
        }
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            f32x2 wv[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) wv[g] = wt[tp * 32 + g];
#pragma unroll
            for (int k = 0; k < F1_PX; ++k) {
                // (packed operands are 64-bit register pairs: the window value rides in one half of a pair and op_sel / op_sel_hi pick that half for both products)
                const f32x2 pw = {pimg[k][tp & ~1], pimg[k][(tp | 1) < 9 ? (tp | 1) : tp]};
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (tp & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[k][g]) : "v"(pw), "s"(wv[g]));
                    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[k][g]) : "v"(pw), "s"(wv[g]));
                }
            }
        }
        asm volatile("s_nop 0" ::: "memory");
#pragma unroll
        for (int k = 0; k < F1_PX; ++k) {
            const int idx = (lw8 + NLOAD * k) * 64 + lane;
            f32x4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) v[g] = mk_f4(acc[k][2 * g][0], acc[k][2 * g][1], acc[k][2 * g + 1][0], acc[k][2 * g + 1][1]);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[g][e] = fmaxf(v[g][e], 0.f); f1_max = fmaxf(f1_max, v[g][e]); }
            u32x4 h0, h1, qg; uint32_t sb;
            wsu_q4_encode16(v, h0, h1, qg, sb);
            if (lw8 + NLOAD * k < IN_SEG && idx < NPIX) {               // (waves 2, 3 have two segments: their third is computed on a clamped pixel and dropped)
                char* d = slot + idx * 16;
                *reinterpret_cast<u32x4*>(d) = h0;
                *reinterpret_cast<u32x4*>(d + PLANE) = h1;
                *reinterpret_cast<u32x4*>(d + 2 * PLANE) = qg;
                *reinterpret_cast<uint32_t*>(slot + 3 * PLANE + idx * 4) = sb;       // the pixel's dword slot: byte 0 = its scale byte
            }
        }
        if (++ci == a.nch && s + 1 < J) { ci = 0; ++kti; ti = tile_of(a, lw + kti * G); window(ti); }
    };
    const int w0 = lw8 < 2 ? lw8 * 5 : 10 + (lw8 - 2) * 9, nwp = lw8 < 2 ? 5 : 9;      // the weight pieces of q_loader<LW>
    auto issue_w = [&](int s) __attribute__((always_inline)) {
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wp), 0, 0x7FFFFFF0, 0x00020000);
        const int w_base = (cbw * a.nch + cw) * W_SLOT;
        lds_char* slot = smem3 + W_BASE + (s % NWS) * W_SLOT;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (k < nwp)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(slot + (w0 + k) * 1024), 16, (unsigned)lane * 16u, w_base + (w0 + k) * 1024, 0, 0);
        if (++cw == a.nch && s + 1 < J) { cw = 0; ++ktw; cbw = tile_of(a, lw + ktw * G).cb; }
    };
    window(ti);
    issue_w(0);
    compute_in(0);
    if (J > 1) compute_in(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int j = 0; ; ++j) {
        __builtin_amdgcn_s_barrier();                                 // barrier j: step j is complete in LDS; every matrix wave has left step j - 1
        asm volatile("" ::: "memory");
        if (j + 1 >= J) break;
        issue_w(j + 1);                                               // its slot held step j - 1
        if (j + 2 < J) compute_in(j + 2);                             // its slot held step j - 1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (a.range_flag && __builtin_amdgcn_ballot_w64(!(f1_max <= WSU_F8_RANGE)) != 0 && lane == 0) atomicOr(a.range_flag, 1u);
}

// RQ: rows of the 16-row tile per matrix wave (4: four matrix waves, one per SIMD; 2: eight).  HC = head planes compiled in: 0, 1 or 4 (1..4).
// FQ: the stored outputs (y, y_pool) are planar Q tensors (else the e4m3-residual format of conv3x3_pl.hip: what the transposed convs read).
// MSPLIT (small grids): a work item is HALF a tile's output channels (m-half = item & 1).
// F1: the input channels (64 = e11's outputs) are computed from the image by the loader waves (q_loader_f1) instead of fetched.
template <int RQ, int HC, bool POOL, bool FQ, bool MSPLIT, bool F1 = false>
__global__ __launch_bounds__((16 / RQ + NLOAD) * 64) __attribute__((amdgpu_waves_per_eu(RQ == 4 ? 2 : 3, RQ == 4 ? 2 : 3)))
void conv3x3_q_kernel(const QArgs a) {
    constexpr int NWAVE = 16 / RQ, NT = (NWAVE + NLOAD) * 64;
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
    for (int i = tid; i < a.cout; i += NT) s_bias[i] = a.bias ? a.bias[i] : 0.f;
    constexpr bool HEAD = HC > 0;
    if constexpr (HEAD) {
        for (int i = tid; i < a.head_cout * 64; i += NT) s_hw[i] = a.head_w[i];
        if (tid < 4) s_hb[tid] = (a.head_b && tid < a.head_cout) ? a.head_b[tid] : 0.f;
    }
    // (these plain loads have retired -- their values went into the LDS stores -- before the first vmcnt wait of a loader; the barrier of step 0
    // publishes them)

    if (wv >= NWAVE) {
        if constexpr (F1) { q_loader_f1(a, smem, lane, lw, G, J, wv - NWAVE); return; }
        switch (wv - NWAVE) {
            case 0: q_loader<0>(a, smem, lane, lw, G, J); break;
            case 1: q_loader<1>(a, smem, lane, lw, G, J); break;
            case 2: q_loader<2>(a, smem, lane, lw, G, J); break;
            default: q_loader<3>(a, smem, lane, lw, G, J); break;
        }
        return;
    }

    // ================= matrix waves ===================================================================================================
    Tile cur = tile_of(a, lw);
    constexpr int MH = MSPLIT ? 1 : 2;                                        // accumulator tiles along the output channels
    constexpr bool PIPE = RQ == 4 || WSU_Q_PIPE2;                             // explicit software pipeline of the fragment reads (below)
    constexpr bool EPO = WSU_Q_EPO && !MSPLIT && HC == 0 && RQ != 4;
    f32x16 acc[2][4];                                                         // [MH][RQ] used (fixed bounds: a template-dependent bound made hipcc (ROCm 7.2) drop the host stubs)
    int kt = 0, j = 0;
    unsigned q_in_off = 0, q_w_off = 0;                                       // this step's input / weight slot
    const char* ldsA = smem;                                                  // + ((tap * 3 + g) * 64 + m * 32) * 16
    const char* ldsB = smem;                                                  // + g * PLANE + ((q + dy) * IW + dx) * 16
    int hh_q = hh;                                                            // an opaque copy of the lane half per step: keeps the tap-pair offsets of the cross terms from being hoisted out of the tile loop (and spilled)
    auto begin_step = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();                                         // the loaders' pieces landed; everyone left the slots that are refilled next
        asm volatile("" ::: "memory");
        q_in_off = (unsigned)(j % NIN) * IN_SLOT; q_w_off = W_BASE + (unsigned)(j % NWS) * W_SLOT;
        ldsA = smem + q_w_off + (cur.mh * 32 + l31) * 16;
        ldsB = smem + q_in_off + ((RQ * wv) * IW + l31) * 16;
        hh_q = hh;
        asm volatile("" : "+v"(hh_q));
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int q = 0; q < RQ; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
    };
    auto main_term = [&](auto tap_c, auto ms_c) __attribute__((always_inline)) {
        constexpr int tap = decltype(tap_c)::value, dy = tap / 3, dx = tap % 3;
        constexpr int ms = decltype(ms_c)::value, ML = ms < 0 ? 0 : ms, MU = ms < 0 ? MH : ms + 1;
        u32x4 ah[2], bh[4];
_Pragma("unroll")
        for (int m = ML; m < MU; ++m) ah[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 3 + hh) * 64 + m * 32) * 16);
_Pragma("unroll")
        for (int q = 0; q < RQ; ++q) bh[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE + ((q + dy) * IW + dx) * 16);
_Pragma("unroll")
        for (int m = ML; m < MU; ++m)
_Pragma("unroll")
            for (int q = 0; q < RQ; ++q) wsu_mfma_f16(ah[m], bh[q], acc[m][q]);
    };
    // both cross terms of a tap pair in one fp4 instruction -- lane half hh carries tap 2 tp + hh: weight granule plane 2 and the pixel's Q granule,
    // each with its E8M0 scale byte (per (tap, co) / per pixel)
    auto cross_q4 = [&](auto tp_c, auto ms_c) __attribute__((always_inline)) {
        constexpr int tp = decltype(tp_c)::value;
        constexpr int ms = decltype(ms_c)::value, ML = ms < 0 ? 0 : ms, MU = ms < 0 ? MH : ms + 1;
        constexpr int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 2 * tp;
        constexpr bool single = 2 * tp + 1 >= 9;
        const int tap = hh_q ? t1 : t0;
        const int pixoff = (tap / 3) * IW + tap % 3;
        u32x4 a4[2], b4[4]; int sa[2], sb[4];
        typedef __attribute__((address_space(3))) const unsigned char lds_cuchar;
        typedef __attribute__((address_space(3))) const int lds_cint;
        typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
        lds_char* L = (lds_char*)smem;
        const unsigned wbase = q_w_off + (unsigned)(cur.mh * 32 + l31) * 16u + (unsigned)((tap * 3 + 2) * 64) * 16u;
        const unsigned sabase = q_w_off + W_GRAN + (unsigned)(tap * 64 + cur.mh * 32 + l31);
        const unsigned pix = (unsigned)((RQ * wv) * IW + l31 + pixoff);
_Pragma("unroll")
        for (int m = ML; m < MU; ++m) {
            a4[m] = *(lds_cu32x4*)(L + wbase + m * 32 * 16);
            sa[m] = *(lds_cuchar*)(L + sabase + m * 32);
        }
_Pragma("unroll")
        for (int q = 0; q < RQ; ++q) {
            b4[q] = *(lds_cu32x4*)(L + q_in_off + 2 * PLANE + (pix + q * IW) * 16u);
            sb[q] = *(lds_cint*)(L + q_in_off + 3 * PLANE + (pix + q * IW) * 4u);      // the pixel's dword slot: byte 0 = its scale byte
        }
        if (single && hh_q) {
            const u32x4 z = mk_u4(0, 0, 0, 0);
_Pragma("unroll")
            for (int m = ML; m < MU; ++m) a4[m] = z;
_Pragma("unroll")
            for (int q = 0; q < RQ; ++q) b4[q] = z;
        }
_Pragma("unroll")
        for (int m = ML; m < MU; ++m)
_Pragma("unroll")
            for (int q = 0; q < RQ; ++q) wsu_mfma_q4(a4[m], b4[q], sa[m], sb[q], acc[m][q]);
    };
    auto units_range = [&](auto ms_c, auto lo_c, auto hi_c) __attribute__((always_inline)) {     // tap pairs [lo, hi)
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
        if constexpr (EPO) asm volatile("" : "+v"(hh_q));                     // (per call: the paths of a step must not share -- and hoist -- their lane offsets)
        WSU_STATIC_FOR(hi - lo, i, {
            constexpr int tp = lo + i;
            if constexpr (!(WSU_Q_PROBE_NOX8 && tp == 4)) cross_q4(std::integral_constant<int, tp>{}, ms_c);
            main_term(std::integral_constant<int, 2 * tp>{}, ms_c);
            if constexpr (2 * tp + 1 < 9) main_term(std::integral_constant<int, 2 * tp + 1>{}, ms_c);
        });
    };
    constexpr std::integral_constant<int, -1> all_m{};
    // ---- RQ = 4, one matrix wave per SIMD: nobody fills the matrix pipe while this wave waits for its fragments, and hipcc issues a unit's LDS reads
    // right in front of its matrix instructions (ISA of the first build: `10 x ds_read, s_waitcnt, mfma` fourteen times per step = fourteen exposed
    // LDS latencies; 6 % SLOWER than two waves per SIMD).  So the step is written as a software pipeline: the fragments of unit u + 1 are requested
    // before the matrix instructions of unit u, into the other of two fragment sets, one scheduling region per unit (the compiler's
    // `s_waitcnt lgkmcnt(n)` then leaves exactly the younger unit's reads in flight).  Units of a step, in the accumulation order of the other
    // variants: u = 3 tp + k: k = 0 the fp4 cross terms of tap pair tp, k = 1 / 2 the f16 products of taps 2 tp / 2 tp + 1 (14 units).
    struct Frag { u32x4 a[2]; u32x4 b[4]; int sa[2]; int sb[4]; };
    auto load_unit = [&](auto u_c, Frag& f) __attribute__((always_inline)) {
        constexpr int u = decltype(u_c)::value, tp = u / 3, k = u % 3;
        typedef __attribute__((address_space(3))) const unsigned char lds_cuchar;
        typedef __attribute__((address_space(3))) const int lds_cint;
        typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
        lds_char* L = (lds_char*)smem;
        if constexpr (k == 0) {
            constexpr int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 2 * tp;
            const int tap = hh_q ? t1 : t0;
            const int pixoff = (tap / 3) * IW + tap % 3;
            const unsigned wbase = q_w_off + (unsigned)(cur.mh * 32 + l31) * 16u + (unsigned)((tap * 3 + 2) * 64) * 16u;
            const unsigned sabase = q_w_off + W_GRAN + (unsigned)(tap * 64 + cur.mh * 32 + l31);
            const unsigned pix = (unsigned)((RQ * wv) * IW + l31 + pixoff);
_Pragma("unroll")
            for (int m = 0; m < MH; ++m) {
                f.a[m] = *(lds_cu32x4*)(L + wbase + m * 32 * 16);
                f.sa[m] = *(lds_cuchar*)(L + sabase + m * 32);
            }
_Pragma("unroll")
            for (int q = 0; q < RQ; ++q) {
                f.b[q] = *(lds_cu32x4*)(L + q_in_off + 2 * PLANE + (pix + q * IW) * 16u);
                f.sb[q] = *(lds_cint*)(L + q_in_off + 3 * PLANE + (pix + q * IW) * 4u);
            }
        } else {
            constexpr int tap = 2 * tp + k - 1, dy = tap / 3, dx = tap % 3;
            const unsigned abase = q_w_off + (unsigned)((cur.mh * 32 + l31) * 16) + (unsigned)(hh_q * 64 * 16);          // 32-bit LDS offsets (no 64-bit pointer arithmetic per read)
            const unsigned bbase = q_in_off + (unsigned)(((RQ * wv) * IW + l31) * 16) + (unsigned)(hh_q * PLANE);
_Pragma("unroll")
            for (int m = 0; m < MH; ++m) f.a[m] = *(lds_cu32x4*)(L + abase + ((tap * 3) * 64 + m * 32) * 16);
_Pragma("unroll")
            for (int q = 0; q < RQ; ++q) f.b[q] = *(lds_cu32x4*)(L + bbase + ((q + dy) * IW + dx) * 16);
        }
    };
    auto mma_unit = [&](auto u_c, Frag& f) __attribute__((always_inline)) {
        constexpr int u = decltype(u_c)::value, tp = u / 3, k = u % 3;
        if constexpr (k == 0) {
            if (2 * tp + 1 >= 9 && hh_q) {                                    // the ninth tap has no partner: lanes 32-63 multiply zeros
                const u32x4 z = mk_u4(0, 0, 0, 0);
_Pragma("unroll")
                for (int m = 0; m < MH; ++m) f.a[m] = z;
_Pragma("unroll")
                for (int q = 0; q < RQ; ++q) f.b[q] = z;
            }
_Pragma("unroll")
            for (int m = 0; m < MH; ++m)
_Pragma("unroll")
                for (int q = 0; q < RQ; ++q) wsu_mfma_q4(f.a[m], f.b[q], f.sa[m], f.sb[q], acc[m][q]);
        } else {
_Pragma("unroll")
            for (int m = 0; m < MH; ++m)
_Pragma("unroll")
                for (int q = 0; q < RQ; ++q) wsu_mfma_f16(f.a[m], f.b[q], acc[m][q]);
        }
    };
    auto units_pipelined = [&]() __attribute__((always_inline)) {
        constexpr int NU = 14, D = WSU_Q_PIPE_DEPTH, NS = D + 1;              // D units of reads in flight, NS fragment sets
        Frag fr[NS];
        WSU_STATIC_FOR(D, u, { load_unit(u_c, fr[u % NS]); });
        WSU_STATIC_FOR(NU, u, {
            if constexpr (u + D < NU) load_unit(std::integral_constant<int, u + D>{}, fr[(u + D) % NS]);
            __builtin_amdgcn_sched_barrier(0);                                // (the reads FIRST: left to itself the scheduler sinks them behind most of the unit's matrix instructions)
            mma_unit(u_c, fr[u % NS]);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto units_all = [&](auto ms_c) __attribute__((always_inline)) {
        if constexpr (PIPE && decltype(ms_c)::value < 0) { units_pipelined(); return; }      // (the split last step of EPO keeps the unpipelined units of its half)
        units_range(ms_c, std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
    };

    // ---- epilogue of the tile: accumulators -> planar global memory ---------------------------------------------------------------------
    auto finish_tile = [&]() __attribute__((always_inline)) {
        // opaque per-tile copies of the lane coordinates: everything the epilogue derives from them (column, plane offsets, store predicates) is
        // recomputed per tile -- hoisted out of the tile loop those values were spilled at the kernel entry and every re-load drained the stores
        int l31 = l31_k, hh = hh_k;
        asm volatile("" : "+v"(l31), "+v"(hh));
        const int col = cur.x0 + l31;
        const unsigned hw16 = (unsigned)(a.h * a.w) * 16u;
        const int nco = a.cout >> 4;                                           // output chunks
        float hz[4][HC > 0 ? HC : 1];
        float vmax = 0.f;                                                      // largest stored activation of this lane (range flag)
        const float relu_floor = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int q = 0; q < RQ; ++q)
#pragma unroll
            for (int o = 0; o < (HC > 0 ? HC : 1); ++o) hz[q][o] = 0.f;
        // stores: wave-uniform descriptor of the (image, output chunk), 32-bit lane offsets; a lane that must not store passes an offset beyond the
        // descriptor's extent, which the hardware drops -- no branch (a branch would end the basic block the matrix instructions are scheduled in)
        // format Q: rows in pairs -- lanes 0-31 assemble and store the Q granule and scale byte of the pair's first row, lanes 32-63 the second's
        auto store_pair_q = [&](const f32x4& X0, const f32x4& Y0, const f32x4& X1, const f32x4& Y1, char* base, unsigned chunk_bytes, unsigned plane_bytes,
                                unsigned off0, unsigned off1, unsigned soff0, unsigned soff1, bool ok0, bool ok1, bool have) __attribute__((always_inline)) {
            u32x4 g0, g1; uint32_t dh0, dr0, sb0, dh1, dr1, sb1;
            wsu_q4_pre(X0, Y0, g0, dh0, dr0, sb0);
            wsu_q4_pre(X1, Y1, g1, dh1, dr1, sb1);
            const u32x4 qg = wsu_q4_pair(dh0, dr0, dh1, dr1);
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, have ? (int)chunk_bytes : 0, 0x00020000);
            const unsigned hp = hh ? plane_bytes : 0u;
            __builtin_amdgcn_raw_buffer_store_b128(g0, rs, (int)(ok0 ? off0 + hp : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(g1, rs, (int)(ok1 ? off1 + hp : OOB), 0, 0);
            const bool okm = hh ? ok1 : ok0;
            __builtin_amdgcn_raw_buffer_store_b128(qg, rs, (int)(okm ? (hh ? off1 : off0) + 2u * plane_bytes : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(hh ? sb1 : sb0), rs, (int)(okm ? 3u * plane_bytes + (hh ? soff1 : soff0) : OOB), 0, 0);
        };
        // ... one row (the pooled row of RQ = 2): lanes 0-31 store the Q granule and the scale byte
        auto store_one_q = [&](const f32x4& X, const f32x4& Y, char* base, unsigned chunk_bytes, unsigned plane_bytes, unsigned off, unsigned soff, bool ok, bool have) __attribute__((always_inline)) {
            u32x4 g; uint32_t dh, dr, sb;
            wsu_q4_pre(X, Y, g, dh, dr, sb);
            const u32x4 qg = wsu_q4_single(dh, dr);
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, have ? (int)chunk_bytes : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(g, rs, (int)(ok ? off + (hh ? plane_bytes : 0u) : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(qg, rs, (int)((ok && !hh) ? off + 2u * plane_bytes : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sb, rs, (int)((ok && !hh) ? 3u * plane_bytes + soff : OOB), 0, 0);
        };
        // format A (the e4m3-residual format of conv3x3_pl.hip: f16 | f16 | e4m3((x - f16 x) * 2^12))
        auto store_one_a = [&](const f32x4& X, const f32x4& Y, char* base, unsigned plane_bytes, unsigned off, bool ok, bool have) __attribute__((always_inline)) {
            uint32_t xh0, xh1, xlo, yh0, yh1, ylo;
            wsu_split4_f16r8(X, WSU_F8_XLO_DIV, xh0, xh1, xlo);
            wsu_split4_f16r8(Y, WSU_F8_XLO_DIV, yh0, yh1, ylo);
            wsu_swap32(xh0, yh0); wsu_swap32(xh1, yh1);                        // lanes 0-31: f16 ch 0-7, lanes 32-63: f16 ch 8-15
            uint32_t xlp = xlo, ylp = ylo;
            wsu_swap32(xlo, xlp); wsu_swap32(ylo, ylp);                        // lanes 0-31: xlp / ylp = the partner lane's residuals (ch 4-7 / 12-15)
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, have ? (int)(3u * plane_bytes) : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(mk_u4(xh0, xh1, yh0, yh1), rs, (int)(ok ? off + (hh ? plane_bytes : 0u) : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(mk_u4(xlo, xlp, ylo, ylp), rs, (int)((ok && !hh) ? off + 2u * plane_bytes : OOB), 0, 0);
        };
        auto epi_m = [&](auto m_c, auto&& hook) __attribute__((always_inline)) {   // one accumulator tile along the output channels: 32 channels x this wave's RQ x 32 pixels
            constexpr int m = decltype(m_c)::value;
            constexpr int NH = RQ / 2;                                         // hooks per piece: one before each row pair
            auto piece = [&](auto cp_c) __attribute__((always_inline)) {       // 16 output channels = accumulator groups g4 = 2cp, 2cp+1
                constexpr int cp = decltype(cp_c)::value;
                const int oc = cur.cb * 4 + (m + cur.mh) * 2 + cp;
                const int co0 = oc * 16 + 4 * hh;                              // this lane: channels co0..co0+3 (X) and co0+8..co0+11 (Y)
                const bool have_y = a.y != nullptr;
                f32x4 px[2], py[2];                                            // POOL: the pooled rows of this piece
                // row pair by row pair (bias, ReLU, encoding and stores of two rows at a time: the values of at most two rows are live beside the
                // accumulators -- with all RQ rows of a piece at once the one-wave-per-SIMD variants spilled 60-70 registers in this block)
                WSU_STATIC_FOR(RQ / 2, rp, {
                    hook(std::integral_constant<int, NH * cp + rp>{});
                    f32x4 vx[2], vy[2];
                    const f32x4 bx = *reinterpret_cast<const f32x4*>(s_bias + co0), by = *reinterpret_cast<const f32x4*>(s_bias + co0 + 8);
_Pragma("unroll")
                    for (int u = 0; u < 2; ++u) {
_Pragma("unroll")
                        for (int e = 0; e < 4; ++e) {
                            // ReLU as one max against a wave-uniform floor (0 or -inf): no select per value
                            const float x = fmaxf(acc[m][2 * rp + u][8 * cp + e] + bx[e], relu_floor), y = fmaxf(acc[m][2 * rp + u][8 * cp + 4 + e] + by[e], relu_floor);
                            vx[u][e] = x; vy[u][e] = y;
                            vmax = fmaxf(vmax, fmaxf(fabsf(x), fabsf(y)));
                        }
                    }
                    if constexpr (HEAD) {
                        const int lc = (m * 2 + cp) * 16 + 4 * hh;             // channel inside the 64-wide block
_Pragma("unroll")
                        for (int o = 0; o < HC; ++o)
                            if (o < a.head_cout)
_Pragma("unroll")
                                for (int u = 0; u < 2; ++u)
_Pragma("unroll")
                                    for (int e = 0; e < 4; ++e)
                                        hz[2 * rp + u][o] = fmaf(vx[u][e], s_hw[o * 64 + lc + e], fmaf(vy[u][e], s_hw[o * 64 + lc + 8 + e], hz[2 * rp + u][o]));
                    }
                    const int r0 = RQ * wv + 2 * rp, row0 = cur.y0 + r0;
                    if (!HEAD || have_y) {
                        if constexpr (FQ) {
                            const unsigned cb_ = (unsigned)wsu_q_chunk_bytes(a.h, a.w);
                            char* base = a.y + ((size_t)cur.n * nco + oc) * cb_;
                            const unsigned sblk = (unsigned)(((cur.y0 >> 4) * a.tiles_x + (cur.x0 >> 5)) * 512);
                            store_pair_q(vx[0], vy[0], vx[1], vy[1], base, cb_, hw16,
                                         (unsigned)(row0 * a.w + col) * 16u, (unsigned)((row0 + 1) * a.w + col) * 16u,
                                         sblk + (unsigned)(r0 * 32 + l31), sblk + (unsigned)((r0 + 1) * 32 + l31),
                                         row0 < a.h && col < a.w, row0 + 1 < a.h && col < a.w, have_y);
                        } else {
                            char* base = a.y + (((size_t)cur.n * nco + oc) * 3) * hw16;
                            WSU_STATIC_FOR(2, u, {
                                store_one_a(vx[u], vy[u], base, hw16, (unsigned)((row0 + u) * a.w + col) * 16u, row0 + u < a.h && col < a.w, have_y);
                            });
                        }
                    }
                    if constexpr (POOL) {                                      // every lane takes part in the exchanges
                        // 2x2 window = the row pair x the lane pair (l, l ^ 1): one max down the column, one across the pair (the neighbour arrives as a
                        // DPP operand); values reaching this point went through v_max(x, floor), which never returns NaN for floor = 0 (ReLU)
_Pragma("unroll")
                        for (int e = 0; e < 4; ++e) {
                            const float cx = fmaxf(vx[0][e], vx[1][e]), cy = fmaxf(vy[0][e], vy[1][e]);
                            px[rp][e] = fmaxf(cx, dpp_xor1(cx)); py[rp][e] = fmaxf(cy, dpp_xor1(cy));
                        }
                    }
                });
                if constexpr (POOL) {
                    const int hp = a.h >> 1, wp2 = a.w >> 1;
                    const int gy0 = (cur.y0 >> 1) + (RQ / 2) * wv, gx = (cur.x0 >> 1) + (l31 >> 1);
                    const bool okx = !(l31 & 1) && gx < wp2;
                    const unsigned php16 = (unsigned)(hp * wp2) * 16u;
                    if constexpr (FQ) {
                        const unsigned cbp = (unsigned)wsu_q_chunk_bytes(hp, wp2);
                        const int ptx = (wp2 + 31) >> 5;
                        char* base = a.ypool + ((size_t)cur.n * nco + oc) * cbp;
                        if constexpr (RQ == 4) {
                            store_pair_q(px[0], py[0], px[1], py[1], base, cbp, php16, (unsigned)(gy0 * wp2 + gx) * 16u, (unsigned)((gy0 + 1) * wp2 + gx) * 16u,
                                         wsu_q_soff(gy0, gx, ptx), wsu_q_soff(gy0 + 1, gx, ptx), okx && gy0 < hp, okx && gy0 + 1 < hp, true);
                        } else {
                            store_one_q(px[0], py[0], base, cbp, php16, (unsigned)(gy0 * wp2 + gx) * 16u, wsu_q_soff(gy0, gx, ptx), okx && gy0 < hp, true);
                        }
                    } else {
                        char* base = a.ypool + (((size_t)cur.n * nco + oc) * 3) * php16;
#pragma unroll
                        for (int rp = 0; rp < RQ / 2; ++rp)
                            store_one_a(px[rp], py[rp], base, php16, (unsigned)((gy0 + rp) * wp2 + gx) * 16u, okx && gy0 + rp < hp, true);
                    }
                }
            };
            piece(std::integral_constant<int, 0>{});
            piece(std::integral_constant<int, 1>{});
        };
        auto nothing = [](auto) __attribute__((always_inline)) {};
        if constexpr (EPO) {
            // the last step's units of m = 1, spread over the hooks of m = 0's epilogue: hook h of NHT carries tap pairs [5 h / NHT, 5 (h + 1) / NHT)
            constexpr int NHT = 2 * (RQ / 2);
            epi_m(std::integral_constant<int, 0>{}, [&](auto h_c) __attribute__((always_inline)) {
                constexpr int h = decltype(h_c)::value;
                constexpr int lo = 5 * h / NHT, hi = 5 * (h + 1) / NHT;
                if constexpr (WSU_Q_EPO_FENCE && RQ == 4) __builtin_amdgcn_sched_barrier(0);
                if constexpr (hi > lo) units_range(std::integral_constant<int, 1>{}, std::integral_constant<int, lo>{}, std::integral_constant<int, hi>{});
            });
            if constexpr (WSU_Q_EPO_FENCE && RQ == 4) __builtin_amdgcn_sched_barrier(0);
            epi_m(std::integral_constant<int, 1>{}, nothing);
        } else {
            WSU_STATIC_FOR(MH, m, { (void)m; epi_m(m_c, nothing); });
        }
        if constexpr (HEAD) {
            const size_t hw = (size_t)a.h * a.w;
            // the other 32 channels of this pixel sit in the partner lane (lane ^ 32)
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const int row = cur.y0 + RQ * wv + q;
#pragma unroll
                for (int o = 0; o < HC; ++o)
                    if (o < a.head_cout) {
                        uint32_t mine = __builtin_bit_cast(uint32_t, hz[q][o]), other = mine;
                        wsu_swap32(mine, other);                            // lanes 0-31: other = partner's sum; lanes 32-63: mine = partner's
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
        // beyond +-448 the e4m3 residual of format A saturates (plain f16 accuracy), beyond +-65504 the f16 part overflows: tell the caller once
        if (a.range_flag && (a.y || POOL) && __builtin_amdgcn_ballot_w64(!(vmax <= WSU_F8_RANGE)) != 0 && lane == 0)
            atomicOr(a.range_flag, 1u);
        ++kt;
        if (j + 1 < J) cur = tile_of(a, lw + kt * G);
    };
    // one loop nest per tile, the last step spelled out after the inner loop
    for (int t = 0; t < K; ++t) {
        zero_acc();
        for (int cc = 1; cc < a.nch; ++cc) { begin_step(); units_all(all_m); ++j; }
        begin_step();
        if constexpr (EPO) units_all(std::integral_constant<int, 0>{});       // the last step: m = 0 first; m = 1 follows inside the epilogue
        else units_all(all_m);
        if (a.ablate & 2) {                                                   // timing only: the tile's results are dropped (kept alive for the compiler)
            if constexpr (EPO) units_all(std::integral_constant<int, 1>{});
#pragma unroll
            for (int m = 0; m < MH; ++m)
#pragma unroll
                for (int q = 0; q < RQ; ++q) asm volatile("" :: "v"(acc[m][q]));
            ++kt;
            if (j + 1 < J) cur = tile_of(a, lw + kt * G);
        } else {
            finish_tile();
        }
        ++j;
    }
}

// one thread per (block, chunk, tap, co): 16 weights -> two f16 granules, the fp4 granule and the scale byte (layout: wsu_conv3x3_pack_f4 below)
__global__ void pack_conv3x3_f4_kernel(const float* __restrict__ w, char* __restrict__ dst, int cin, int cout) {
    const int nch = cin / 16;
    const long long total = (long long)(cout / WSU_COB) * nch * 9 * WSU_COB;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int co = (int)(t % WSU_COB); t /= WSU_COB;
        const int tap = (int)(t % 9); t /= 9;
        const int c = (int)(t % nch); const int cb = (int)(t / nch);
        float v[16], r[16];
        uint32_t h[8];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = w[(((size_t)(cb * WSU_COB + co) * cin + c * 16 + e) * 3 + tap / 3) * 3 + tap % 3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 a = (_Float16)v[2 * e], b = (_Float16)v[2 * e + 1];              // round to nearest even, as every f16 part of the library
            h[e] = (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
            r[2 * e] = (v[2 * e] - (float)a) * 2048.f; r[2 * e + 1] = (v[2 * e + 1] - (float)b) * 2048.f;
        }
        const u32x4 h0 = mk_u4(h[0], h[1], h[2], h[3]), h1 = mk_u4(h[4], h[5], h[6], h[7]);
        const int E = wsu_q4_block_exp(wsu_f16x16_max_abs_bits(h0, h1));
        const float sc = wsu_pow2f(E);
        uint32_t q[4] = {0, 0, 0, 0};
        WSU_STATIC_FOR(8, e, { q[e >> 2] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q[e >> 2], r[2 * e], r[2 * e + 1], sc, e & 3); });   // nibbles 0-15: residuals, meet fp4(x)
        q[2] = wsu_f16x8_to_fp4(h0, sc); q[3] = wsu_f16x8_to_fp4(h1, sc);                                                            // nibbles 16-31: copies, meet the residuals of x
        char* slice = dst + ((size_t)cb * nch + c) * W_SLOT;
        char* base = slice + (size_t)(tap * 3) * (WSU_COB * 16) + co * 16;
        *reinterpret_cast<u32x4*>(base) = h0;
        *reinterpret_cast<u32x4*>(base + WSU_COB * 16) = h1;
        *reinterpret_cast<u32x4*>(base + 2 * WSU_COB * 16) = mk_u4(q[0], q[1], q[2], q[3]);
        slice[W_GRAN + tap * 64 + co] = (char)(E + 127 - 11);
        if (tap == 0 && co < 7) *reinterpret_cast<u32x4*>(slice + W_GRAN + 576 + co * 64) = mk_u4(0, 0, 0, 0), *reinterpret_cast<u32x4*>(slice + W_GRAN + 576 + co * 64 + 16) = mk_u4(0, 0, 0, 0),
            *reinterpret_cast<u32x4*>(slice + W_GRAN + 576 + co * 64 + 32) = mk_u4(0, 0, 0, 0), *reinterpret_cast<u32x4*>(slice + W_GRAN + 576 + co * 64 + 48) = mk_u4(0, 0, 0, 0);   // the 448 pad bytes
    }
}

#define WSU_Q_INST(RQ) \
    template __global__ void conv3x3_q_kernel<RQ, 0, false, false, false>(const QArgs); \
    template __global__ void conv3x3_q_kernel<RQ, 0, false, true, false>(const QArgs);  \
    template __global__ void conv3x3_q_kernel<RQ, 0, true, false, false>(const QArgs);  \
    template __global__ void conv3x3_q_kernel<RQ, 0, true, true, false>(const QArgs);   \
    template __global__ void conv3x3_q_kernel<RQ, 1, false, false, false>(const QArgs); \
    template __global__ void conv3x3_q_kernel<RQ, 4, false, false, false>(const QArgs); \
    template __global__ void conv3x3_q_kernel<RQ, 0, false, false, true>(const QArgs);  \
    template __global__ void conv3x3_q_kernel<RQ, 0, false, true, true>(const QArgs);
WSU_Q_INST(4)
WSU_Q_INST(2)
template __global__ void conv3x3_q_kernel<2, 0, true, true, false, true>(const QArgs);

template <int RQ>
int q_launch_rq(QArgs a, int yq, hipStream_t s, int ncu, bool msplit_on) {
    constexpr int NT = (16 / RQ + NLOAD) * 64;
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[8] = {reinterpret_cast<const void*>(&conv3x3_q_kernel<RQ, 0, false, false, false>), reinterpret_cast<const void*>(&conv3x3_q_kernel<RQ, 0, false, true, false>),
                              reinterpret_cast<const void*>(&conv3x3_q_kernel<RQ, 0, true, false, false>), reinterpret_cast<const void*>(&conv3x3_q_kernel<RQ, 0, true, true, false>),
                              reinterpret_cast<const void*>(&conv3x3_q_kernel<RQ, 1, false, false, false>), reinterpret_cast<const void*>(&conv3x3_q_kernel<RQ, 4, false, false, false>),
                              reinterpret_cast<const void*>(&conv3x3_q_kernel<RQ, 0, false, false, true>), reinterpret_cast<const void*>(&conv3x3_q_kernel<RQ, 0, false, true, true>)};
        for (const void* fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
            if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3_q): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        }
        attr_done = true;
    }
    a.msplit = 0;
    if (msplit_on && !a.head_w && !a.ypool && 2 * (long long)a.ntiles <= ncu) {       // small grids: half-block work items
        a.msplit = 1; a.ncb *= 2; a.ntiles *= 2;
        if (yq) hipLaunchKernelGGL((conv3x3_q_kernel<RQ, 0, false, true, true>), dim3(a.ntiles), dim3(NT), LDS_TOTAL, s, a);
        else hipLaunchKernelGGL((conv3x3_q_kernel<RQ, 0, false, false, true>), dim3(a.ntiles), dim3(NT), LDS_TOTAL, s, a);
        return wsu_check_launch("conv3x3_q_kernel<msplit>");
    }
    const dim3 g(a.ntiles < ncu ? a.ntiles : ncu), b(NT);
    if (a.head_w && a.head_cout == 1) hipLaunchKernelGGL((conv3x3_q_kernel<RQ, 1, false, false, false>), g, b, LDS_TOTAL, s, a);
    else if (a.head_w) hipLaunchKernelGGL((conv3x3_q_kernel<RQ, 4, false, false, false>), g, b, LDS_TOTAL, s, a);
    else if (a.ypool && yq) hipLaunchKernelGGL((conv3x3_q_kernel<RQ, 0, true, true, false>), g, b, LDS_TOTAL, s, a);
    else if (a.ypool) hipLaunchKernelGGL((conv3x3_q_kernel<RQ, 0, true, false, false>), g, b, LDS_TOTAL, s, a);
    else if (yq) hipLaunchKernelGGL((conv3x3_q_kernel<RQ, 0, false, true, false>), g, b, LDS_TOTAL, s, a);
    else hipLaunchKernelGGL((conv3x3_q_kernel<RQ, 0, false, false, false>), g, b, LDS_TOTAL, s, a);
    return wsu_check_launch("conv3x3_q_kernel");
}

}  // namespace

extern "C" {

// Weights of the fp4-cross-term conv: per (64-channel output block, 16-channel input chunk) one 28 KB slice =
// [tap 9][plane 3][64 co][16 B] with planes f16 ci 0-7 | f16 ci 8-15 | fp4(residual * 2^11 / 2^E) ci 0-15, fp4(f16 part / 2^E) ci 0-15 (nibble i =
// channel i), then [tap 9][64 co] scale bytes E + 127 - 11 (the 2^-11 of the residual's pre-scaling rides in the weight's scale), zero padded to 1 KB.
size_t wsu_conv3x3_packed_f4_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0 || cin % 16 || cout % WSU_COB) return 0;
    return (size_t)(cout / WSU_COB) * (cin / 16) * W_SLOT;
}
int wsu_conv3x3_pack_f4(const float* w_oihw, void* w_packed, int cin, int cout, void* stream) {
    WSU_REQUIRE(w_oihw && w_packed, "conv3x3_pack_f4: null pointer");
    WSU_REQUIRE(cin > 0 && cin % 16 == 0 && cout > 0 && cout % WSU_COB == 0, "conv3x3_pack_f4: cin=%d must be a multiple of 16, cout=%d of %d", cin, cout, WSU_COB);
    hipLaunchKernelGGL(pack_conv3x3_f4_kernel, dim3(512), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw, (char*)w_packed, cin, cout);
    return wsu_check_launch("pack_conv3x3_f4_kernel");
}

// e11 + e12 (+ pool) of the default inference mode in ONE launch for single-plane inputs (unet.py:141-144; kernel variant F1 above): img (N,1,H,W)
// fp32; w1_taps (9, 64) fp32 = e11's weights tap-major (w1.reshape(64, 9).T: the loaders fetch a tap's 16 channels with one scalar load), b1 (64);
// w_packed_f4 / bias: the second conv (cin = 64, wsu_conv3x3_pack_f4); y and y_pool: planar Q tensors, cout
// channels at (h, w) and (h/2, w/2).  Bitwise the result of wsu_conv3x3_first_pl_fwd(y_format Q) followed by wsu_conv3x3_q_fwd; xe11 never reaches
// HBM.  h, w even; cout a multiple of 64.  range_flag as in wsu_conv3x3_q_fwd (it also covers the computed xe11 values).
int wsu_conv3x3_q_fused_first_fwd(const float* img, const float* w1_taps, const float* b1, const void* w_packed_f4, const float* bias, void* y, void* y_pool,
                                  int n, int h, int w, int cout, int relu, unsigned* range_flag, void* stream) {
    const float* w1 = w1_taps;
    WSU_REQUIRE(img && w1 && b1 && w_packed_f4 && y && y_pool, "conv3x3_q_fused_first: null pointer");
    WSU_REQUIRE(((uintptr_t)w1 & 63) == 0 && ((uintptr_t)b1 & 63) == 0, "conv3x3_q_fused_first: w1_taps and b1 must be 64-byte aligned (scalar 16-dword loads)");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "conv3x3_q_fused_first: bad shape n=%d h=%d w=%d (even h, w: the pooled output)", n, h, w);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0 && cout <= 1024, "conv3x3_q_fused_first: cout=%d must be a multiple of %d (<= 1024)", cout, WSU_COB);
    WSU_REQUIRE((long long)h * w * 50 < 0xFFFFFFF0LL, "conv3x3_q_fused_first: h*w too large (a chunk must stay below 4 GiB)");
    QArgs a;
    a.x1 = nullptr; a.x2 = nullptr; a.wp = (const char*)w_packed_f4; a.bias = bias;
    a.y = (char*)y; a.ypool = (char*)y_pool;
    a.head_w = nullptr; a.head_b = nullptr; a.head_out = nullptr; a.head_logit = nullptr; a.head_cout = 0;
    a.range_flag = range_flag;
    a.n = n; a.h = h; a.w = w; a.c1 = 64; a.c2 = 0; a.cout = cout;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ncb = cout / WSU_COB;
    static int prio = -1;
    if (prio < 0) { const char* e = getenv("WSU_Q_F1_PRIO"); prio = (e && atoi(e)) ? 1 : 0; }
    a.nch1 = 4; a.nch = 4; a.relu = relu; a.msplit = 0; a.ablate = prio ? 16 : 0;
    a.img = img; a.w1 = w1; a.b1 = b1;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "conv3x3_q_fused_first: %lld tiles out of range", nt);
    a.ntiles = (int)nt;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            wsu_set_error("conv3x3_q_fused_first: cannot query the device"); return WSU_ERR_HIP;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_q_kernel<2, 0, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3_q<F1>): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        ncu = prop.multiProcessorCount;
    }
    hipLaunchKernelGGL((conv3x3_q_kernel<2, 0, true, true, false, true>), dim3(a.ntiles < ncu ? a.ntiles : ncu), dim3((8 + NLOAD) * 64), LDS_TOTAL,
                       static_cast<hipStream_t>(stream), a);
    return wsu_check_launch("conv3x3_q_kernel<F1>");
}

// Bytes of a planar Q tensor (n images, c channels -- a multiple of 16 -- at h x w): n * c/16 chunks of 48 h w + 512 ceil(h/16) ceil(w/32) bytes.
size_t wsu_planar_q_bytes(int n, int c, int h, int w) {
    if (n <= 0 || c <= 0 || c % 16 || h <= 0 || w <= 0) return 0;
    return (size_t)n * (size_t)(c / 16) * wsu_q_chunk_bytes(h, w);
}

// Forward 3x3 reflect conv + bias + ReLU in the fp4-cross-term arithmetic on planar Q activations (layout: wsu.h).  x1 (c1 channels) and optional
// x2 (c2, fused concat): planar Q tensors; packed weights of wsu_conv3x3_pack_f4; outputs, each optional: y (cout channels) and y_pool (2x2
// max-pooled) -- planar Q tensors when y_format = WSU_PLANAR_Q, else the e4m3-residual planar format (what wsu_convt2x2_pl_fwd reads) --, head
// (1x1 conv + sigmoid on the 64 output channels: out / logit NCHW fp32; needs cout == 64; a y beside the head is written in the e4m3 format).
// range_flag as in wsu_conv3x3_pl_fwd.  c1, c2 multiples of 16, cout of 64.  Asynchronous on `stream`; allocates nothing.
int wsu_conv3x3_q_fwd(const void* x1, const void* x2, const void* w_packed_f4, const float* bias, void* y, void* y_pool,
                      const float* head_w, const float* head_b, float* head_out, float* head_logit, int head_cout,
                      int n, int h, int w, int c1, int c2, int cout, int relu, int y_format, unsigned* range_flag, void* stream) {
    WSU_REQUIRE(x1 && w_packed_f4 && (y || y_pool || head_w), "conv3x3_q: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_q: bad shape n=%d h=%d w=%d (reflect pad 1 needs h,w >= 2)", n, h, w);
    WSU_REQUIRE(c1 > 0 && c1 % 16 == 0 && c2 >= 0 && c2 % 16 == 0 && (c2 == 0) == (x2 == nullptr), "conv3x3_q: c1=%d c2=%d must be multiples of 16", c1, c2);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0 && cout <= 1024, "conv3x3_q: cout=%d must be a multiple of %d (<= 1024)", cout, WSU_COB);
    WSU_REQUIRE(!head_w || (head_out && cout == WSU_COB && head_cout >= 1 && head_cout <= 4), "conv3x3_q: fused head needs cout == %d and 1..4 head planes", WSU_COB);
    WSU_REQUIRE(!y_pool || (h % 2 == 0 && w % 2 == 0), "conv3x3_q: fused pool needs even h, w");
    WSU_REQUIRE(!(y_pool && head_w), "conv3x3_q: the fused pool and the fused head exclude each other");
    WSU_REQUIRE(y_format == WSU_PLANAR_A || y_format == WSU_PLANAR_Q, "conv3x3_q: y_format must be WSU_PLANAR_A or WSU_PLANAR_Q");
    WSU_REQUIRE(!(head_w && y && y_format != WSU_PLANAR_A), "conv3x3_q: a y beside the fused head is written in format WSU_PLANAR_A");
    WSU_REQUIRE((long long)h * w * 50 < 0xFFFFFFF0LL, "conv3x3_q: h*w too large (a chunk must stay below 4 GiB)");
    QArgs a;
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.wp = (const char*)w_packed_f4; a.bias = bias;
    a.y = (char*)y; a.ypool = (char*)y_pool;
    a.head_w = head_w; a.head_b = head_b; a.head_out = head_out; a.head_logit = head_logit; a.head_cout = head_cout;
    a.range_flag = range_flag;
    a.n = n; a.h = h; a.w = w; a.c1 = c1; a.c2 = c2; a.cout = cout;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ncb = cout / WSU_COB;
    a.nch1 = c1 / 16; a.nch = (c1 + c2) / 16; a.relu = relu;
    a.img = nullptr; a.w1 = nullptr; a.b1 = nullptr;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "conv3x3_q: %lld tiles out of range", nt);
    a.ntiles = (int)nt;
    static int ncu = 0, rq = 0, msplit_on = 1, ablate = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            wsu_set_error("conv3x3_q: cannot query the device"); return WSU_ERR_HIP;
        }
        const char* e = getenv("WSU_Q_ROWS"); rq = (e && atoi(e) == 4) ? 4 : 2;       // experiment switch: 4 = four matrix waves (one per SIMD) x four rows
        e = getenv("WSU_PL_MSPLIT"); msplit_on = e ? atoi(e) : 1;
        e = getenv("WSU_PL_ABLATE"); ablate = e ? atoi(e) : 0;
        ncu = prop.multiProcessorCount;
    }
    a.ablate = ablate;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return rq != 4 ? q_launch_rq<2>(a, y_format == WSU_PLANAR_Q, s, ncu, msplit_on != 0) : q_launch_rq<4>(a, y_format == WSU_PLANAR_Q, s, ncu, msplit_on != 0);
}

}  // extern "C"
