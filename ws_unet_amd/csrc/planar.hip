// The remaining kernels of the planar (F16F8P) inference path, next to the 3x3 conv of conv3x3_pl.hip:
//   K3p  convt2x2_pl_kernel   nn.ConvTranspose2d(k=2, s=2) + bias (src/unet/model/unet.py:125,130,177,183) on planar activations, with the
//                              machinery of conv3x3_pl: one persistent workgroup per CU, loader waves feeding two LDS stages by LDS-DMA,
//                              matrix waves that only multiply, results from the accumulators straight to planar global memory
//   K0p  first_pl_kernel       the first layer e11 (unet.py:82,141): 1..8 input planes (NCHW fp32, the model input) -> planar 16 k channels
// Storage format and arithmetic: include/wsu.h (F16F8P), wsu_device.h.
#include "wsu_device.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__device__ __forceinline__ void swap32(uint32_t& upper_of, uint32_t& lower_of) {
    const auto r = __builtin_amdgcn_permlane32_swap(upper_of, lower_of, false, false);   // lanes 32-63 of the first <-> lanes 0-31 of the second
    upper_of = r[0]; lower_of = r[1];
}

// X, Y = 4 + 4 values of one pixel (channels c0 + 4 hh .. and c0 + 8 + 4 hh ..; hh = lane >> 5): encode, gather whole 16-byte granules into
// single lanes (conv3x3_pl.hip) and store the chunk's three planes (f16 | f16 | residuals): every instruction writes contiguous runs of 32 lanes x 16 B.
__device__ __forceinline__ void store_chunk_px(const f32x4& X, const f32x4& Y, char* dst, size_t plane_bytes, int hh, bool ok, float div_lo = WSU_F8_XLO_DIV, bool with_res = true) {
    uint32_t xh0, xh1, xlo, yh0, yh1, ylo;
    wsu_split4_f16r8(X, div_lo, xh0, xh1, xlo);
    wsu_split4_f16r8(Y, div_lo, yh0, yh1, ylo);
    swap32(xh0, yh0); swap32(xh1, yh1);
    uint32_t xlp = xlo, ylp = ylo;
    swap32(xlo, xlp); swap32(ylo, ylp);                                  // lanes 0-31 collect all 16 residuals of the chunk
    if (ok) {
        *reinterpret_cast<u32x4*>(dst + hh * plane_bytes) = mk_u4(xh0, xh1, yh0, yh1);
        if (with_res && !hh) *reinterpret_cast<u32x4*>(dst + 2 * plane_bytes) = mk_u4(xlo, xlp, ylo, ylp);
    }
}

// =====================================================================================================================================
// K3p.  y[n, 2i+a, 2j+b, co] = bias[co] + sum_ci x[n, i, j, ci] * w[ci, co, a, b]: four 1-tap GEMMs that share their B operand.
// Tile = 4 x 32 INPUT pixels x 64 co x 4 sub-positions.  Matrix wave w: output-row parity a = w & 1, input rows 2 (w>>1 & 1) + {0, 1}, output
// channels 32 (w >> 2) + 0..31, BOTH column parities b: its lane l31 holds the output pixels 2 (x0 + l31) and 2 (x0 + l31) + 1, i.e. 32
// contiguous bytes per plane -- the accumulators go straight to planar global memory (two 16-byte stores side by side).
// A step = 2 chunks of 16 input channels (the fp8 instruction pairs two chunks: a sub-position has one tap); stage = 2 x (input 4 planes x
// 128 px x 16 B + weights [4 sub-positions][4 planes][64 co][16 B]) = 48 KB, two stages; 48 DMA pieces per step over 4 loader waves.
// =====================================================================================================================================
namespace ct {
constexpr int TW = 32, TH = 4, NPIX = TW * TH;            // 128 input pixels
constexpr int PLANE = NPIX * 16;                          // 2048
constexpr int IN1 = WSU_GRAN * PLANE;                     // 8192 per chunk
constexpr int W1 = 4 * WSU_GRAN * WSU_COB * 16;           // 16384 per chunk (layout of wsu_convt2x2_pack, mode F16F8)
constexpr int CHUNK = IN1 + W1;                           // 24576
constexpr int STAGE = 2 * CHUNK;                          // 49152
constexpr int NSTAGE = 3;                                 // the DMA of step j+2 goes out behind barrier j (a step of this kernel is ~1.5 us of
                                                          // matrix work: one step of distance left the DMA's latency exposed -- 4.3 k cycles per step)
constexpr int LDS_EXTRA = NSTAGE * STAGE;                 // bias [1024]
constexpr int LDS_TOTAL = LDS_EXTRA + 1024 * 4;
constexpr int NWAVE = 8, NLOAD = 4, NT = (NWAVE + NLOAD) * 64;
constexpr int HBM_PLANES = 3;                             // stored planes per chunk (LDS plane 3 is derived by the loaders)
constexpr int IN_SLOTS = 2 * HBM_PLANES * 2;              // 2 chunks x 3 planes x 2 segments of 64 pixels = 12
constexpr int W_SLOTS = 2 * W1 / 1024;                    // 32
constexpr int IN_PER = IN_SLOTS / NLOAD, W_PER = W_SLOTS / NLOAD;   // 3, 8
}

struct CtpArgs {
    const char* x; const char* wp; const float* bias; char* y; unsigned* range_flag;
    int n, h, w, cin, cout;
    int tiles_x, tiles_y, ncb, nst;                       // nst = steps per tile = cin / 32
    int ntiles;
    int yq;                                               // 1: y is a planar Q tensor (the 3x3 convs of mode 'f16f4p' read it), 0: the e4m3-residual format
};

struct CtTile { int n, y0, x0, cb; };
__device__ __forceinline__ CtTile ct_tile_of(const CtpArgs& a, int t) {
    CtTile r;
    r.cb = t % a.ncb; t /= a.ncb;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    r.n = t / a.tiles_y; r.y0 = ty * ct::TH; r.x0 = tx * ct::TW;
    return r;
}

typedef __attribute__((address_space(3))) char lds_char;

// Loader wave LW of the transposed conv (round 3, same treatment as conv3x3_pl.hip: the slot -> (chunk, plane, segment) geometry is
// compile-time, a piece is one `buffer_load_dwordx4 ... offen lds` with a wave-uniform descriptor + scalar offset and a per-lane pixel
// offset computed once per tile; round 2 recomputed a clamped 64-bit address per piece and step).
template <int LW>
__device__ __forceinline__ void ct_issue_dma(const CtpArgs& a, const CtTile& t, int step, lds_char* st, int lane, const unsigned (&pixoff)[2]) {
    using namespace ct;
    const unsigned hw16 = (unsigned)(a.h * a.w) * 16u;
    const int nch = a.cin >> 4;
    const char* in_base = a.x + ((size_t)t.n * nch + 2 * step) * HBM_PLANES * hw16;        // the step's two chunks: 6 planes
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_base), 0, (int)(6u * hw16), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wp), 0, 0x7FFFFFF0, 0x00020000);
    const int w_base = (t.cb * nch + 2 * step) * W1;
    const unsigned lane16 = (unsigned)lane * 16u;
    WSU_STATIC_FOR(IN_PER, k, {
        constexpr int slot = LW + NLOAD * k, ck = slot / 6, plane = (slot % 6) >> 1, seg = slot & 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(st + ck * CHUNK + plane * PLANE + seg * 1024), 16, pixoff[seg],
                                                 (int)((ck * HBM_PLANES + plane) * hw16), 0, 0);
    });
    WSU_STATIC_FOR(W_PER, k, {
        constexpr int slot = LW + NLOAD * k, ck = slot >> 4, piece = slot & 15;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(st + ck * CHUNK + IN1 + piece * 1024), 16, lane16, w_base + ck * W1 + piece * 1024, 0, 0);
    });
}

template <int LW>
__device__ __forceinline__ void ct_loader(const CtpArgs& a, char* smem, int lane, int lw, int G, int J) {
    using namespace ct;
    static_assert(IN_PER + W_PER == 11, "the vmcnt immediate below");
    // ---- every wave issues IN_PER + W_PER = 11 pieces per step (all lanes live), so `vmcnt(11)` = "my pieces of step j have landed, those of
    //      step j+1 are still in flight" ---------------------------------------------------------------------------------------------------
    lds_char* smem3 = (lds_char*)smem;
    CtTile t = ct_tile_of(a, lw);                                       // tile / step of the NEXT issue
    int c = 0, kt = 0;
    unsigned pixoff[2];
    auto plan = [&]() __attribute__((always_inline)) {                   // lane -> pixel of the tile (clamped: out-of-image lanes are never stored)
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const int pix = seg * 64 + lane;
            const int yy = min(t.y0 + pix / TW, a.h - 1), xx = min(t.x0 + pix % TW, a.w - 1);
            pixoff[seg] = (unsigned)(yy * a.w + xx) * 16u;
        }
    };
    plan();
    auto issue_next = [&](int j_issue) __attribute__((always_inline)) {
        ct_issue_dma<LW>(a, t, c, smem3 + (j_issue % NSTAGE) * STAGE, lane, pixoff);
        if (++c == a.nst) { c = 0; ++kt; t = ct_tile_of(a, lw + kt * G); plan(); }
    };
    if (J > 0) issue_next(0);
    if (J > 1) issue_next(1);
    for (int j = 0; j < J; ++j) {
        if (j + 1 < J) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        else           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {   // LDS plane 3 = e4m3 copies of the f16 granules this wave fetched (same lanes: no cross-wave dependency)
            char* st = smem + (j % NSTAGE) * STAGE;
            WSU_STATIC_FOR(IN_PER, k, {
                constexpr int slot = LW + NLOAD * k, ck = slot / 6, plane = (slot % 6) >> 1, seg = slot & 1;
                if constexpr (plane < 2) {
                    const int pix = seg * 64 + lane;
                    const u32x4 hgr = *reinterpret_cast<const u32x4*>(st + ck * CHUNK + plane * PLANE + pix * 16);
                    *reinterpret_cast<u32x2*>(st + ck * CHUNK + 3 * PLANE + pix * 16 + plane * 8) = wsu_f16x8_to_fp8(hgr);
                }
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + 2 < J) issue_next(j + 2);                               // its stage held step j-1: every matrix wave is past it
    }
}

template <bool YQ>                                                           // YQ: y is a planar Q tensor (compile-time: with both epilogues in one kernel the register allocator spilled 33 registers -- 0.45 -> 0.68 ms per launch)
__global__ __launch_bounds__(ct::NT) void convt2x2_pl_kernel(const CtpArgs a) {
    using namespace ct;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int G = gridDim.x;
    const int lw = (int)wsu_xcd_remap(blockIdx.x, G);
    const int K = a.ntiles > lw ? (a.ntiles - lw + G - 1) / G : 0;
    const int J = K * a.nst;
    float* s_bias = reinterpret_cast<float*>(smem + LDS_EXTRA);
    for (int i = tid; i < a.cout; i += NT) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    if (wv >= NWAVE) {
        switch (wv - NWAVE) {
            case 0: ct_loader<0>(a, smem, lane, lw, G, J); break;
            case 1: ct_loader<1>(a, smem, lane, lw, G, J); break;
            case 2: ct_loader<2>(a, smem, lane, lw, G, J); break;
            default: ct_loader<3>(a, smem, lane, lw, G, J); break;
        }
        return;
    }

    // ---- matrix waves --------------------------------------------------------------------------------------------------------------
    const int pa = wv & 1, half = (wv >> 1) & 1, mh = wv >> 2;
    const int l31_k = l31, hh_k = hh;
    CtTile cur = ct_tile_of(a, lw);
    f32x16 acc[2][2];                                                   // [b][q]
    const int sc_a = hh ? WSU_F8_SCALE_WLO : WSU_F8_SCALE_W, sc_b = hh ? WSU_F8_SCALE_X : WSU_F8_SCALE_XLO;
    int c = 0, kt = 0;
    for (int j = 0; j < J; ++j) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + (j % NSTAGE) * STAGE;
        if (c == 0) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][q][r] = 0.f;
        }
        // A: st + ck*CHUNK + IN1 + (((2 pa + b) * 4 + g) * 64 + mh*32 + l31) * 16;  B: st + ck*CHUNK + g*PLANE + ((2 half + q) * 32 + l31) * 16
        const char* ldsA = st + IN1 + ((2 * pa * 4) * 64 + mh * 32 + l31) * 16;
        const char* ldsB = st + ((2 * half) * TW + l31) * 16;
        u32x4 a0[2], a1[2], b0[2], b1[2];                              // fp8 operands: this lane half's chunk of the pair
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            a0[b] = *reinterpret_cast<const u32x4*>(ldsA + hh * CHUNK + ((b * 4 + 2) * 64) * 16);
            a1[b] = *reinterpret_cast<const u32x4*>(ldsA + hh * CHUNK + ((b * 4 + 3) * 64) * 16);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            b0[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * CHUNK + 2 * PLANE + q * TW * 16);
            b1[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * CHUNK + 3 * PLANE + q * TW * 16);
        }
#pragma unroll
        for (int ck = 0; ck < 2; ++ck) {
            u32x4 ah[2], bh[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) ah[b] = *reinterpret_cast<const u32x4*>(ldsA + ck * CHUNK + ((b * 4 + hh) * 64) * 16);
#pragma unroll
            for (int q = 0; q < 2; ++q) bh[q] = *reinterpret_cast<const u32x4*>(ldsB + ck * CHUNK + hh * PLANE + q * TW * 16);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) wsu_mfma_f16(ah[b], bh[q], acc[b][q]);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) wsu_mfma_f8x2(a0[b], a1[b], b0[q], b1[q], sc_a, sc_b, acc[b][q]);

        if (c + 1 == a.nst) {
            // ---- epilogue: this lane's pixels 2 (x0 + l31) + b of output rows 2 (y0 + 2 half + q) + pa, channels cb*64 + mh*32 + ... --------
            const int oh = 2 * a.h, ow = 2 * a.w;
            const size_t ohw = (size_t)oh * ow;
            const int nco = a.cout >> 4;
            // (opaque per-tile copies of the lane coordinates: what the epilogue derives from them is recomputed per tile instead of being hoisted out of
            // the step loop and spilled -- the Q variant sits at the kernel's 168-register step)
            int l31 = l31_k, hh = hh_k;
            if constexpr (YQ) asm volatile("" : "+v"(l31), "+v"(hh));
            const int icol = cur.x0 + l31;
            float vmax = 0.f;
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                const int oc = cur.cb * 4 + mh * 2 + cp;
                const int co0 = oc * 16 + 4 * hh;
                const f32x4 bx = *reinterpret_cast<const f32x4*>(s_bias + co0), by = *reinterpret_cast<const f32x4*>(s_bias + co0 + 8);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int irow = cur.y0 + 2 * half + q;
                    const bool ok = irow < a.h && icol < a.w;
                    auto value = [&](int b_, f32x4& X, f32x4& Y) __attribute__((always_inline)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            X[e] = acc[b_][q][8 * cp + e] + bx[e]; Y[e] = acc[b_][q][8 * cp + 4 + e] + by[e];
                            vmax = fmaxf(vmax, fmaxf(fabsf(X[e]), fabsf(Y[e])));
                        }
                    };
                    if constexpr (YQ) {                              // the lane's two output pixels are a pair of the Q format's epilogue (wsu_device.h)
                        const int orow = 2 * irow + pa, ocol = 2 * icol;
                        // stores through a wave-uniform descriptor of the (image, output chunk) + 32-bit lane offsets: no 64-bit address registers (the
                        // kernel sits at its 168-register step); a lane that must not store passes an offset beyond the descriptor, which the hardware drops
                        const unsigned cbytes = (unsigned)wsu_q_chunk_bytes(oh, ow), pb = (unsigned)ohw * 16u;
                        const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.y + ((size_t)cur.n * nco + oc) * cbytes, 0, (int)cbytes, 0x00020000);
                        const unsigned off = (unsigned)(orow * ow + ocol) * 16u;
                        const unsigned OOB_ = 0xFFFFFFF0u;
                        const int otx = (ow + 31) >> 5;
                        f32x4 X, Y; u32x4 g; uint32_t dh0, dr0, sb0, dh1, dr1, sb1;
                        value(0, X, Y);
                        wsu_q4_pre(X, Y, g, dh0, dr0, sb0);
                        __builtin_amdgcn_raw_buffer_store_b128(g, rs, (int)(ok ? off + (hh ? pb : 0u) : OOB_), 0, 0);
                        value(1, X, Y);
                        wsu_q4_pre(X, Y, g, dh1, dr1, sb1);
                        __builtin_amdgcn_raw_buffer_store_b128(g, rs, (int)(ok ? off + 16u + (hh ? pb : 0u) : OOB_), 0, 0);
                        const u32x4 qg = wsu_q4_pair(dh0, dr0, dh1, dr1);
                        __builtin_amdgcn_raw_buffer_store_b128(qg, rs, (int)(ok ? 2u * pb + off + (hh ? 16u : 0u) : OOB_), 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(hh ? sb1 : sb0), rs, (int)(ok ? 3u * pb + wsu_q_soff(orow, ocol + hh, otx) : OOB_), 0, 0);
                        __builtin_amdgcn_sched_barrier(0);           // one (16 channels, row) group at a time: interleaved, the four groups of a tile spilled
                    } else {
                        char* dst = a.y + ((((size_t)cur.n * nco + oc) * HBM_PLANES) * ohw + (size_t)(2 * irow + pa) * ow + 2 * icol) * 16;
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            f32x4 X, Y;
                            value(b, X, Y);
                            store_chunk_px(X, Y, dst + b * 16, ohw * 16, hh, ok);
                        }
                    }
                }
            }
            if (a.range_flag && __builtin_amdgcn_ballot_w64(!(vmax <= WSU_F8_RANGE)) != 0 && lane == 0) atomicOr(a.range_flag, 1u);
            ++kt; c = 0;
            if (j + 1 < J) cur = ct_tile_of(a, lw + kt * G);
        } else {
            ++c;
        }
    }
}

// =====================================================================================================================================
// K7p.  Data gradient of the transposed conv (autograd of unet.py:177,183):  dx[n, i, j, ci] = sum_{a,b,co} dy[n, 2i+a, 2j+b, co] * w[ci, co, a, b],
// times the ReLU mask of the layer below.  GEMM M = 64 ci, N = 4 x 32 input-resolution pixels, K = 4 sub-positions x Cout; persistent
// workgroup per CU: 8 matrix waves (32 ci x one tile row each) + 8 loader waves (the kernel is DMA-issue / HBM bound: 40 pieces per chunk).
// The dy tile (8 x 64 output pixels) arrives row by row -- LDS image [plane][row parity a][tile row r][64 output px][16 B] -- and the B-operand
// read of sub-position (a, b) takes every other granule of a row (ctb_loader explains why the de-interleaving moved from the DMA to the read).  Weights [sub][plane][64 ci][16 B] per chunk from
// wsu_convt2x2_pl_pack_dgrad.  Gradient encodings in and out (wsu_device.h).
// =====================================================================================================================================
namespace ctb {
constexpr int TW = 32, TH = 4, NPIX = TW * TH;
constexpr int IN1 = 4 * 4 * NPIX * 16;                    // 32768: [plane 4][sub 4][128 px][16 B]
constexpr int W1 = 4 * 4 * WSU_COB * 16;                  // 16384: [sub 4][plane 4][64 ci][16 B]
constexpr int STAGE = IN1 + W1;                           // 49152
constexpr int NSTAGE = 3;                                 // DMA two steps ahead (see ct::NSTAGE)
constexpr int LDS_TOTAL = NSTAGE * STAGE;
constexpr int NWAVE = 8, NLOAD = 8, NT = (NWAVE + NLOAD) * 64;
constexpr int IN_SLOTS = 3 * 4 * 2, W_SLOTS = W1 / 1024;  // 24 + 16 pieces per chunk
constexpr int PER = (IN_SLOTS + W_SLOTS) / NLOAD;         // 5
static_assert((IN_SLOTS + W_SLOTS) % NLOAD == 0, "DMA pieces divide over the loader waves");
}

struct CtbPlArgs {
    const char* dy; const char* wp; char* dx; const char* mask;
    int n, h, w, cin, cout;
    int tiles_x, tiles_y, ncb, nch, ntiles;
};

__device__ __forceinline__ CtTile ctb_tile_of(const CtbPlArgs& a, int t) {
    CtTile r;
    r.cb = t % a.ncb; t /= a.ncb;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    r.n = t / a.tiles_y; r.y0 = ty * ctb::TH; r.x0 = tx * ctb::TW;
    return r;
}

// Loader wave LW (0..7) of the transposed conv's data gradient: its input pieces are the planes of ONE output row of the tile -- row parity
// a = (LW >> 2) & 1, tile row r = LW & 3 -- 64 CONSECUTIVE output pixels per piece (both column parities; so a lane has one pixel offset per
// tile), then its weight pieces.  Descriptors + scalar plane offsets as in ct_issue_dma.
// (Round 2 / early round 3 de-interleaved while fetching -- a piece = every other pixel of a row, the LDS image [plane][sub][128 px] -- so every
// DMA instruction used 16 of each 32 bytes it touched and the L2 -> CU path carried the rows twice: 3.2 - 3.4 TB/s of HBM traffic whatever the
// pipeline depth.  Now the LDS image is [plane][a][r][64 output px] and the B-operand reads take the stride of 2 granules instead.)
// HONLY (products = WSU_PRODUCTS_F16, round 3): only the f16 planes of dy and of the weights travel -- 16 + 8 KB instead of 32 + 16 KB per step,
// so SIX stages fit and the DMA runs five steps ahead (80 KB of dy in flight per CU instead of 48: the kernel is HBM-latency bound).
template <bool HONLY> struct CtbGeo {
    static constexpr int NPL = HONLY ? 2 : 3;                          // dy planes fetched
    static constexpr int IN1 = (HONLY ? 2 : 4) * 4 * ctb::NPIX * 16;   // [plane][a 2][r 4][64 output px][16 B]
    static constexpr int W1L = HONLY ? ctb::W1 / 2 : ctb::W1;          // HONLY: [sub 4][plane 2][64 ci][16 B]
    static constexpr int STAGE = IN1 + W1L;
    static constexpr int NSTAGE = HONLY ? 6 : 3;
    static constexpr int LDS_TOTAL = NSTAGE * STAGE;
    static constexpr int PER = HONLY ? 3 : 5;                          // DMA instructions per loader wave and step
};
static_assert(CtbGeo<false>::STAGE == ctb::STAGE && CtbGeo<false>::LDS_TOTAL == ctb::LDS_TOTAL && CtbGeo<true>::LDS_TOTAL <= 160 * 1024 - 1024, "LDS budget");

template <int LW, bool HONLY>
__device__ __forceinline__ void ctb_loader(const CtbPlArgs& a, char* smem, int lane, int lw, int G, int J) {
    using namespace ctb;
    using GEO = CtbGeo<HONLY>;
    static_assert(PER == 5 && IN_SLOTS == 24 && NLOAD == 8, "the slot arithmetic and the vmcnt immediates below");   // every piece has live lanes: GEO::PER DMA instructions per wave and step
    constexpr int pa = (LW >> 2) & 1, pr = LW & 3;
    constexpr int AHEAD = GEO::NSTAGE - 1;                              // the DMA of step j + AHEAD goes out behind barrier j
    lds_char* smem3 = (lds_char*)smem;
    const int oh = 2 * a.h, ow = 2 * a.w;
    const unsigned ohw16 = (unsigned)(oh * ow) * 16u;
    CtTile t = ctb_tile_of(a, lw);                                      // tile / chunk of the NEXT issue
    int c = 0, kt = 0;
    unsigned pixoff = 0;
    auto plan = [&]() __attribute__((always_inline)) {
        const int oy = min(2 * (t.y0 + pr) + pa, oh - 1), ox = min(2 * t.x0 + lane, ow - 1);
        pixoff = (unsigned)(oy * ow + ox) * 16u;
    };
    plan();
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wp), 0, 0x7FFFFFF0, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_next = [&](int j_issue) __attribute__((always_inline)) {
        lds_char* st = smem3 + (j_issue % GEO::NSTAGE) * GEO::STAGE;
        const char* in_base = a.dy + ((size_t)t.n * a.nch + c) * 3 * ohw16;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_base), 0, (int)(3u * ohw16), 0x00020000);
        const int w_base = (t.cb * a.nch + c) * W1;
        WSU_STATIC_FOR(GEO::NPL, plane, {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(st + (((plane * 2 + pa) * 4 + pr) * 64) * 16), 16, pixoff, (int)(plane * ohw16), 0, 0);
        });
        if constexpr (HONLY) {                                          // packed piece (sub, plane) = sub * 4 + plane, planes 0 / 1 only: wave LW takes (LW >> 1, LW & 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(st + GEO::IN1 + LW * 1024), 16, lane16, w_base + ((LW >> 1) * 4 + (LW & 1)) * 1024, 0, 0);
        } else {
            WSU_STATIC_FOR(2, k, {
                constexpr int piece = LW + NLOAD * k;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(st + GEO::IN1 + piece * 1024), 16, lane16, w_base + piece * 1024, 0, 0);
            });
        }
        if (++c == a.nch) { c = 0; ++kt; t = ctb_tile_of(a, lw + kt * G); plan(); }
    };
    for (int k = 0; k < AHEAD && k < J; ++k) issue_next(k);
    for (int j = 0; j < J; ++j) {
        // my pieces of step j landed; those of the steps issued behind it (at most AHEAD - 1 of them) stay in flight
        const int ahead = min(J - 1 - j, AHEAD - 1);
        if constexpr (HONLY) {
            static_assert(GEO::PER == 3 && AHEAD == 5, "vmcnt immediates");
            if (ahead >= 4)      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (ahead == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            static_assert(HONLY || (GEO::PER == 5 && AHEAD == 2), "vmcnt immediates");
            if (ahead >= 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (!HONLY) {   // LDS plane 3 = e4m3(g * 4) of the f16 granules this wave fetched
            char* st = smem + (j % GEO::NSTAGE) * GEO::STAGE;
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) {
                const u32x4 hgr = *reinterpret_cast<const u32x4*>(st + ((((plane * 2 + pa) * 4 + pr) * 64) + lane) * 16);
                *reinterpret_cast<u32x2*>(st + ((((3 * 2 + pa) * 4 + pr) * 64) + lane) * 16 + plane * 8) = wsu_f16x8_to_fp8_grad(hgr);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + AHEAD < J) issue_next(j + AHEAD);                       // its stage held step j-1: every matrix wave is past it
    }
}

template <bool HONLY>
__global__ __launch_bounds__(ctb::NT) void convt2x2_bwd_pl_kernel(const CtbPlArgs a) {
    using namespace ctb;
    using GEO = CtbGeo<HONLY>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int G = gridDim.x;
    const int lw = (int)wsu_xcd_remap(blockIdx.x, G);
    const int K = a.ntiles > lw ? (a.ntiles - lw + G - 1) / G : 0;
    const int J = K * a.nch;

    if (wv >= NWAVE) {
        switch (wv - NWAVE) {
            case 0: ctb_loader<0, HONLY>(a, smem, lane, lw, G, J); break;
            case 1: ctb_loader<1, HONLY>(a, smem, lane, lw, G, J); break;
            case 2: ctb_loader<2, HONLY>(a, smem, lane, lw, G, J); break;
            case 3: ctb_loader<3, HONLY>(a, smem, lane, lw, G, J); break;
            case 4: ctb_loader<4, HONLY>(a, smem, lane, lw, G, J); break;
            case 5: ctb_loader<5, HONLY>(a, smem, lane, lw, G, J); break;
            case 6: ctb_loader<6, HONLY>(a, smem, lane, lw, G, J); break;
            default: ctb_loader<7, HONLY>(a, smem, lane, lw, G, J); break;
        }
        return;
    }

    const int row = wv & 3, mh = wv >> 2;
    CtTile cur = ctb_tile_of(a, lw);
    f32x16 acc;
    const int sc_a = hh ? WSU_F8_SCALE_WLO : WSU_F8_SCALE_W, sc_b = hh ? WSU_F8_SCALE_G : WSU_F8_SCALE_GLO;
    const size_t hw = (size_t)a.h * a.w;
    const int nci = a.cin >> 4;
    u32x2 mk[2][2];                                                      // [chunk-in-wave cp][plane]: this lane's 4 + 4 mask values
    int c = 0, kt = 0;
    for (int j = 0; j < J; ++j) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + (j % GEO::NSTAGE) * GEO::STAGE;
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        const int irow = cur.y0 + row, icol = cur.x0 + l31;
        if (c == 0 && a.mask) {                                          // the tile's mask values travel during ALL its matrix sections (fetched in the
                                                                         // last one -- 4 to 6 MFMAs -- the epilogue sat out an HBM latency per tile)
            const int yy = min(irow, a.h - 1), xx = min(icol, a.w - 1);
#pragma unroll
            for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    mk[cp][pl] = *reinterpret_cast<const u32x2*>(a.mask + ((((size_t)cur.n * nci + cur.cb * 4 + mh * 2 + cp) * 3 + pl) * hw + (size_t)yy * a.w + xx) * 16 + hh * 8);
        }
        const char* ldsA = st + GEO::IN1 + (mh * 32 + l31) * 16;       // + ((sub * 4 + plane) * 64) * 16   (HONLY: (sub * 2 + plane))
        const char* ldsB = st + (row * 64 + 2 * l31) * 16;              // + (((plane * 2 + a) * 4) * 64 + b) * 16, sub = 2 a + b
        auto boff = [](int plane, int sub) constexpr { return (((plane * 2 + (sub >> 1)) * 4) * 64 + (sub & 1)) * 16; };
        if constexpr (HONLY) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const u32x4 ah = *reinterpret_cast<const u32x4*>(ldsA + ((s2 * 2 + hh) * 64) * 16);
                const u32x4 bh = *reinterpret_cast<const u32x4*>(ldsB + boff(hh, s2));
                wsu_mfma_f16(ah, bh, acc);
            }
        } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int s = 2 * t + hh;
            const u32x4 a0 = *reinterpret_cast<const u32x4*>(ldsA + ((s * 4 + 2) * 64) * 16), a1 = *reinterpret_cast<const u32x4*>(ldsA + ((s * 4 + 3) * 64) * 16);
            const u32x4 b0 = *reinterpret_cast<const u32x4*>(ldsB + boff(2, s)), b1 = *reinterpret_cast<const u32x4*>(ldsB + boff(3, s));
            wsu_mfma_f8x2(a0, a1, b0, b1, sc_a, sc_b, acc);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s2 = 2 * t + u;
                const u32x4 ah = *reinterpret_cast<const u32x4*>(ldsA + ((s2 * 4 + hh) * 64) * 16);
                const u32x4 bh = *reinterpret_cast<const u32x4*>(ldsB + boff(hh, s2));
                wsu_mfma_f16(ah, bh, acc);
            }
        }
        }
        if (c + 1 == a.nch) {
            const bool ok = irow < a.h && icol < a.w;
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                const int oc = cur.cb * 4 + mh * 2 + cp;
                f32x4 X, Y;
#pragma unroll
                for (int e = 0; e < 4; ++e) { X[e] = acc[8 * cp + e]; Y[e] = acc[8 * cp + 4 + e]; }
                if (a.mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                            // f16 > 0 <=> its bits > 0 as a signed 16-bit integer
                        const uint32_t wx = e < 2 ? mk[cp][0].x : mk[cp][0].y, wy = e < 2 ? mk[cp][1].x : mk[cp][1].y;
                        const short hx = (short)((e & 1) ? wx >> 16 : wx & 0xFFFF), hy = (short)((e & 1) ? wy >> 16 : wy & 0xFFFF);
                        if (!(hx > 0)) X[e] = 0.f;
                        if (!(hy > 0)) Y[e] = 0.f;
                    }
                }
                char* dst = a.dx + ((((size_t)cur.n * nci + oc) * 3) * hw + (size_t)min(irow, a.h - 1) * a.w + min(icol, a.w - 1)) * 16;
                store_chunk_px(X, Y, dst, hw * 16, hh, ok, WSU_F8_GLO_DIV, !HONLY);      // (products F16: gradient tensors carry no residual plane)
            }
            ++kt; c = 0;
            if (j + 1 < J) cur = ctb_tile_of(a, lw + kt * G);
        } else {
            ++c;
        }
    }
}

// [cb = ci / 64][chunk = co / 16][sub = 2a + b][plane][64 ci][16 B]: planes f16 co 0-7 | f16 co 8-15 | e4m3(w * 2^6) | e4m3((w - f16 w) * 2^18)
__global__ void pack_convt_dgrad_pl_kernel(const float* __restrict__ w, char* __restrict__ dst, int cin, int cout) {
    const int nch = cout / 16;
    const long long total = (long long)(cin / WSU_COB) * nch * 4 * WSU_COB;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int m = t % WSU_COB; t /= WSU_COB;
        const int sub = t % 4; t /= 4;
        const int c = t % nch; const int cb = (int)(t / nch);
        f32x4 q[4];
#pragma unroll
        for (int e = 0; e < 16; ++e) q[e >> 2][e & 3] = w[(((size_t)(cb * WSU_COB + m) * cout + c * 16 + e) * 2 + (sub >> 1)) * 2 + (sub & 1)];
        uint32_t h[8], l[4], x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wsu_split4_f16f8(q[k], WSU_F8_WLO_DIV, WSU_F8_W_DIV, h[2 * k], h[2 * k + 1], l[k], x[k]);
        char* base = dst + (((size_t)cb * nch + c) * 4 + sub) * (4 * WSU_COB * 16) + m * 16;
        *reinterpret_cast<u32x4*>(base) = mk_u4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<u32x4*>(base + WSU_COB * 16) = mk_u4(h[4], h[5], h[6], h[7]);
        *reinterpret_cast<u32x4*>(base + 2 * WSU_COB * 16) = mk_u4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<u32x4*>(base + 3 * WSU_COB * 16) = mk_u4(l[0], l[1], l[2], l[3]);
    }
}

// =====================================================================================================================================
// K0p.  First layer: y = relu(conv3x3_reflect(x) + b), x (N, cin <= 8, H, W) fp32 NCHW, y planar with cout = 16 k channels.  One thread per
// pixel (consecutive lanes = consecutive pixels of a row: every store instruction writes 64 x 16 contiguous bytes), 16 output channels at a
// time from tap-major weights in LDS; fp32 FMAs in the tap order of conv3x3_first_kernel (pointwise.hip), so the values before encoding are
// bitwise those of the NHWC first-layer kernel.  HBM-write bound: 4 bytes per output element.
// =====================================================================================================================================
struct FirstPlArgs { const float* x; const float* w; const float* b; char* y; unsigned* range_flag; int n, h, w_, cin, cout, relu; unsigned char* relu_mask_out; int yq; };

template <bool YQ, int CINMAX>                                              // YQ: y is a planar Q tensor (compile-time, like convt2x2_pl_kernel); CINMAX: 1 (the
__global__ __launch_bounds__(256) void first_pl_kernel(const FirstPlArgs a) {   // published runs' single plane: 9 window registers instead of 72) or 8
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);                         // [ci][tap][cout]
    float* bl = wl + a.cin * 9 * a.cout;
    for (int i = threadIdx.x; i < a.cin * 9 * a.cout; i += blockDim.x) {
        const int co = i % a.cout, tap = (i / a.cout) % 9, ci = i / (9 * a.cout);
        wl[i] = a.w[((size_t)co * a.cin + ci) * 9 + tap];
    }
    for (int i = threadIdx.x; i < a.cout; i += blockDim.x) bl[i] = a.b ? a.b[i] : 0.f;
    __syncthreads();
    const size_t hw = (size_t)a.h * a.w_;
    const long long total = (long long)a.n * hw;
    const int nco = a.cout >> 4;
    float vmax = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.w_); long long t = i / a.w_;
        const int y = (int)(t % a.h); const int img = (int)(t / a.h);
        float p[CINMAX][9];
#pragma unroll
        for (int ci = 0; ci < CINMAX; ++ci)
            if (ci < a.cin) {
                const float* src = a.x + ((size_t)img * a.cin + ci) * hw;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
                    p[ci][tp] = src[(size_t)wsu_reflect(y + tp / 3 - 1, a.h) * a.w_ + wsu_reflect(x + tp % 3 - 1, a.w_)];
            }
        for (int oc = 0; oc < nco; ++oc) {
            // 16 channels = eight accumulator PAIRS updated by v_pk_fma_f32 (two fused multiply-adds per instruction: bitwise the fmaf chain, half the
            // issue slots; VERDICT r03 next #8: -fno-slp-vectorize took the compiler's packed forms away and cost this kernel 5-8 %).  Written by
            // hand because of the packed-f32 read-after-write hazard (profiles/r03/pk_hazard.md): a pair is re-read eight instructions later by its
            // own next tap, and an explicit `s_nop 0` separates the last update from the epilogue's first read (tests/test_isa_lint.py scans for
            // adjacent producer / consumer pairs).
            f32x2 acc2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc2[k] = *reinterpret_cast<const f32x2*>(bl + oc * 16 + 2 * k);
#pragma unroll
            for (int ci = 0; ci < CINMAX; ++ci)
                if (ci < a.cin)
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const float* wr = wl + (ci * 9 + tp) * a.cout + oc * 16;
                        // (packed operands are 64-bit register pairs: the window value rides in one half of a pair and op_sel / op_sel_hi pick that half for both products)
                        const f32x2 pw = {p[ci][tp & ~1], p[ci][(tp | 1) < 9 ? (tp | 1) : tp]};
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * g);
                            const f32x2 wlo = {w4[0], w4[1]}, whi = {w4[2], w4[3]};
                            if (tp & 1) {
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc2[2 * g]) : "v"(pw), "v"(wlo));
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc2[2 * g + 1]) : "v"(pw), "v"(whi));
                            } else {
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc2[2 * g]) : "v"(pw), "v"(wlo));
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc2[2 * g + 1]) : "v"(pw), "v"(whi));
                            }
                        }
                    }
            asm volatile("s_nop 0" ::: "memory");
            f32x4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) v[g] = mk_f4(acc2[2 * g][0], acc2[2 * g][1], acc2[2 * g + 1][0], acc2[2 * g + 1][1]);
            uint32_t h[8], lo[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) vmax = fmaxf(vmax, fabsf(v[g][e]));
            }
            if constexpr (YQ) {                                          // planar Q output: f16 | f16 | Q | scale byte (wsu_device.h)
                u32x4 h0, h1, qg; uint32_t sb;
                wsu_q4_encode16(v, h0, h1, qg, sb);
                char* chunk = a.y + ((size_t)img * nco + oc) * wsu_q_chunk_bytes(a.h, a.w_);
                char* dq = chunk + ((size_t)y * a.w_ + x) * 16;
                *reinterpret_cast<u32x4*>(dq) = h0;
                *reinterpret_cast<u32x4*>(dq + hw * 16) = h1;
                *reinterpret_cast<u32x4*>(dq + 2 * hw * 16) = qg;
                *reinterpret_cast<unsigned char*>(chunk + 3 * hw * 16 + wsu_q_soff(y, x, (a.w_ + 31) >> 5)) = (unsigned char)sb;
                continue;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) wsu_split4_f16r8(v[g], WSU_F8_XLO_DIV, h[2 * g], h[2 * g + 1], lo[g]);
            char* dst = a.y + ((((size_t)img * nco + oc) * 3) * hw + (size_t)y * a.w_ + x) * 16;
            *reinterpret_cast<u32x4*>(dst) = mk_u4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<u32x4*>(dst + hw * 16) = mk_u4(h[4], h[5], h[6], h[7]);
            *reinterpret_cast<u32x4*>(dst + 2 * hw * 16) = mk_u4(lo[0], lo[1], lo[2], lo[3]);
            if (a.relu_mask_out) {                                       // the training forward's 1-bit ReLU mask: one byte per (pixel, 8 channels)
                unsigned char* m = a.relu_mask_out + (((size_t)img * (nco * 2) + oc * 2) * wsu_mask_hp(a.h) + y) * wsu_mask_wp(a.w_) + x;
                m[0] = (unsigned char)wsu_f16x8_pos_bits(mk_u4(h[0], h[1], h[2], h[3]));
                m[(size_t)wsu_mask_hp(a.h) * wsu_mask_wp(a.w_)] = (unsigned char)wsu_f16x8_pos_bits(mk_u4(h[4], h[5], h[6], h[7]));
            }
        }
    }
    if (a.range_flag && !(vmax <= WSU_F8_RANGE)) atomicOr(a.range_flag, 1u);      // rare: at most one atomic per lane
}

}  // namespace

extern "C" {

// K3p: transposed 2x2 stride-2 conv + bias on planar F16F8P activations.  x: cin channels at (h, w); y: cout channels at (2h, 2w); weights from
// wsu_convt2x2_pack(mode F16F8).  cin a multiple of 32, cout of 64.
int wsu_convt2x2_pl_fwd(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin, int cout,
                        int y_format, unsigned* range_flag, void* stream) {
    WSU_REQUIRE(y_format == WSU_PLANAR_A || y_format == WSU_PLANAR_Q, "convt2x2_pl: y_format must be WSU_PLANAR_A or WSU_PLANAR_Q");
    WSU_REQUIRE(x && w_packed && y, "convt2x2_pl: null pointer");
    WSU_REQUIRE(n > 0 && h > 0 && w > 0, "convt2x2_pl: bad shape n=%d h=%d w=%d", n, h, w);
    WSU_REQUIRE(cin > 0 && cin % 32 == 0, "convt2x2_pl: cin=%d must be a multiple of 32", cin);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0 && cout <= 1024, "convt2x2_pl: cout=%d must be a multiple of %d (<= 1024)", cout, WSU_COB);
    WSU_REQUIRE((long long)h * w * 200 < 0xFFFFFFF0LL, "convt2x2_pl: h*w too large (two input chunks and one output chunk must stay below 4 GiB)");
    CtpArgs a;
    a.x = (const char*)x; a.wp = (const char*)w_packed; a.bias = bias; a.y = (char*)y; a.range_flag = range_flag;
    a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout; a.yq = y_format == WSU_PLANAR_Q ? 1 : 0;
    a.tiles_x = (w + ct::TW - 1) / ct::TW; a.tiles_y = (h + ct::TH - 1) / ct::TH; a.ncb = cout / WSU_COB; a.nst = cin / 32;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "convt2x2_pl: %lld tiles out of range", nt);
    a.ntiles = (int)nt;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            wsu_set_error("convt2x2_pl: cannot query the device"); return WSU_ERR_HIP;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2x2_pl_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ct::LDS_TOTAL);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2x2_pl_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ct::LDS_TOTAL);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(convt2x2_pl): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        ncu = prop.multiProcessorCount;
    }
    const int grid = (int)(nt < ncu ? nt : ncu);
    if (a.yq) hipLaunchKernelGGL(convt2x2_pl_kernel<true>, dim3(grid), dim3(ct::NT), ct::LDS_TOTAL, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(convt2x2_pl_kernel<false>, dim3(grid), dim3(ct::NT), ct::LDS_TOTAL, static_cast<hipStream_t>(stream), a);
    return wsu_check_launch("convt2x2_pl_kernel");
}

// K7p: data gradient of the transposed conv on planar tensors.  dy: cout channels at (2h, 2w), planar gradient; dx: cin channels at (h, w),
// planar gradient; mask (optional): the planar ACTIVATION the transposed conv consumed (its sign = ReLU mask of the layer below).  Weights from
// wsu_convt2x2_pl_pack_dgrad (cin * cout * 16 bytes).  cin a multiple of 64, cout of 16.
int wsu_convt2x2_pl_pack_dgrad(const float* w_iohw, void* w_packed, int cin, int cout, void* stream) {
    WSU_REQUIRE(w_iohw && w_packed, "convt2x2_pl_pack_dgrad: null pointer");
    WSU_REQUIRE(cin > 0 && cin % WSU_COB == 0 && cout > 0 && cout % 16 == 0, "convt2x2_pl_pack_dgrad: bad channels cin=%d cout=%d", cin, cout);
    hipLaunchKernelGGL(pack_convt_dgrad_pl_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), w_iohw, (char*)w_packed, cin, cout);
    return wsu_check_launch("pack_convt_dgrad_pl_kernel");
}

int wsu_convt2x2_pl_bwd_data(const void* dy, const void* w_packed_dgrad, void* dx, const void* mask,
                             int n, int h, int w, int cin, int cout, int products, void* stream) {
    WSU_REQUIRE(dy && w_packed_dgrad && dx, "convt2x2_pl_bwd_data: null pointer");
    WSU_REQUIRE(products == WSU_PRODUCTS_F16F8 || products == WSU_PRODUCTS_F16, "convt2x2_pl_bwd_data: products must be WSU_PRODUCTS_F16F8 or WSU_PRODUCTS_F16");
    WSU_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cin % WSU_COB == 0 && cout > 0 && cout % 16 == 0, "convt2x2_pl_bwd_data: bad shape (cin %% 64, cout %% 16)");
    WSU_REQUIRE((long long)h * w * 192 < 0xFFFFFFF0LL, "convt2x2_pl_bwd_data: h*w too large (a plane triple of dy must stay below 4 GiB)");
    CtbPlArgs a;
    a.dy = (const char*)dy; a.wp = (const char*)w_packed_dgrad; a.dx = (char*)dx; a.mask = (const char*)mask;
    a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
    a.tiles_x = (w + ctb::TW - 1) / ctb::TW; a.tiles_y = (h + ctb::TH - 1) / ctb::TH; a.ncb = cin / WSU_COB; a.nch = cout / 16;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "convt2x2_pl_bwd_data: %lld tiles out of range", nt);
    a.ntiles = (int)nt;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            wsu_set_error("convt2x2_pl_bwd_data: cannot query the device"); return WSU_ERR_HIP;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2x2_bwd_pl_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CtbGeo<false>::LDS_TOTAL);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2x2_bwd_pl_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CtbGeo<true>::LDS_TOTAL);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(convt2x2_bwd_pl): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        ncu = prop.multiProcessorCount;
    }
    const int grid = (int)(nt < ncu ? nt : ncu);
    if (products == WSU_PRODUCTS_F16) hipLaunchKernelGGL(convt2x2_bwd_pl_kernel<true>, dim3(grid), dim3(ctb::NT), CtbGeo<true>::LDS_TOTAL, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(convt2x2_bwd_pl_kernel<false>, dim3(grid), dim3(ctb::NT), CtbGeo<false>::LDS_TOTAL, static_cast<hipStream_t>(stream), a);
    return wsu_check_launch("convt2x2_bwd_pl_kernel");
}

// K0p: first layer into planar storage.  x_nchw: (N, cin, H, W) fp32, cin 1..8; w_oihw: (cout, cin, 3, 3); cout a multiple of 16 (<= 128).
int wsu_conv3x3_first_pl_fwd(const float* x_nchw, const float* w_oihw, const float* bias, void* y, int n, int h, int w, int cin, int cout,
                             int relu, int y_format, unsigned* range_flag, unsigned char* relu_mask_out, void* stream) {
    WSU_REQUIRE(y_format == WSU_PLANAR_A || y_format == WSU_PLANAR_Q, "conv3x3_first_pl: y_format must be WSU_PLANAR_A or WSU_PLANAR_Q");
    WSU_REQUIRE(!(relu_mask_out && y_format == WSU_PLANAR_Q), "conv3x3_first_pl: relu_mask_out belongs to the training forward (format WSU_PLANAR_A)");
    WSU_REQUIRE((long long)h * w * 50 < 0xFFFFFFF0LL, "conv3x3_first_pl: h*w too large");
    WSU_REQUIRE(x_nchw && w_oihw && y, "conv3x3_first_pl: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2 && cin >= 1 && cin <= 8 && cout > 0 && cout % 16 == 0 && cout <= 128, "conv3x3_first_pl: bad shape");
    WSU_REQUIRE(!relu_mask_out || (long long)n * (cout / 8) * wsu_mask_hp(h) * wsu_mask_wp(w) < 0x7FFFFFF0LL, "conv3x3_first_pl: mask plane too large");
    FirstPlArgs a{x_nchw, w_oihw, bias, (char*)y, range_flag, n, h, w, cin, cout, relu, relu_mask_out, y_format == WSU_PLANAR_Q ? 1 : 0};
    const long long total = (long long)n * h * w;
    const unsigned nblk = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const size_t lds = ((size_t)cin * 9 * cout + cout) * sizeof(float);
    hipStream_t s_ = static_cast<hipStream_t>(stream);
    if (cin == 1) {
        if (a.yq) hipLaunchKernelGGL((first_pl_kernel<true, 1>), dim3(nblk), dim3(256), lds, s_, a);
        else hipLaunchKernelGGL((first_pl_kernel<false, 1>), dim3(nblk), dim3(256), lds, s_, a);
    } else {
        if (a.yq) hipLaunchKernelGGL((first_pl_kernel<true, 8>), dim3(nblk), dim3(256), lds, s_, a);
        else hipLaunchKernelGGL((first_pl_kernel<false, 8>), dim3(nblk), dim3(256), lds, s_, a);
    }
    return wsu_check_launch("first_pl_kernel");
}

}  // extern "C"
