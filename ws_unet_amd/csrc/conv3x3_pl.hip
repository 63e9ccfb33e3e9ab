// K1p: the 3x3 convolution of the inference path in mode F16F8P -- the f16f8 arithmetic of conv3x3.hip on PLANAR activations, fed by
// LDS-DMA and software-pipelined across chunks AND tiles by one persistent workgroup per CU.
//
// Why a second kernel (profiles/r02/conv3x3_units_probe.md): in conv3x3_kernel the time is affine in the matrix work, t = a + b * units, and
// the part `a` that does not overlap the matrix pipe (register staging: global loads at 48 used bytes per 192..768-byte pixel stride, the
// ds_write commit with its fp8 derivation, two barriers per chunk, the LDS round trip of the epilogue, an exposed prologue per tile) is
// 45-50 % of every layer.  Here none of that work exists:
//   * activations are stored planar, [n][C/16 chunks][4 planes][H][W][16 B] (planes: f16 ch 0-7 | f16 ch 8-15 | e4m3 residuals ch 0-15 |
//     e4m3 copies ch 0-15 = exactly the four LDS planes of a chunk), so an input-tile row is one contiguous 544-byte run per plane and
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

namespace {

constexpr int TW = 32, TH = 16, IW = TW + 2, IH = TH + 2;
constexpr int NPIX = IW * IH;                             // 612 input-tile pixels
constexpr int PLANE = NPIX * 16;                          // 9792 B per granule plane
constexpr int LDS_IN = WSU_GRAN * PLANE;                  // 39168
constexpr int LDS_W = 9 * WSU_GRAN * WSU_COB * 16;        // 36864
constexpr int STAGE = LDS_IN + LDS_W;                     // 76032
constexpr int LDS_EXTRA = 2 * STAGE;                      // bias [1024] | head_w [4][64] | head_b [4]
constexpr int LDS_TOTAL = LDS_EXTRA + 1024 * 4 + 4 * 64 * 4 + 16;
constexpr int NT = 512, NWAVE = 8;
constexpr int IN_SEG = (NPIX + 63) / 64;                  // 10 wave-instructions per plane (the last one 36 lanes wide)
constexpr int IN_SLOTS = WSU_GRAN * IN_SEG;               // 40
constexpr int IN_PER_WAVE = IN_SLOTS / NWAVE;             // 5
constexpr int W_SLOTS = LDS_W / 1024;                     // 36
constexpr int W_PER_WAVE = (W_SLOTS + NWAVE - 1) / NWAVE; // 5 (waves 4..7: 4)
static_assert(IN_SLOTS % NWAVE == 0, "input DMA slots divide over the waves");

struct PlArgs {
    const char* x1; const char* x2; const char* wp; const float* bias;
    char* y; char* ypool;
    const float* head_w; const float* head_b; float* head_out; float* head_logit; int head_cout;
    int n, h, w, c1, c2, cout;
    int tiles_x, tiles_y, ncb, nch1, nch;
    int relu;
    int ntiles;                                           // n * tiles_y * tiles_x * ncb
};

struct Tile { int n, y0, x0, cb; };

__device__ __forceinline__ Tile tile_of(const PlArgs& a, int t) {
    Tile r;
    r.cb = t % a.ncb; t /= a.ncb;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    r.n = t / a.tiles_y; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// source offsets (in 16-byte units, inside one chunk's 4 planes of one image) of this lane's 5 input DMA slots; -1 = lane idle
__device__ __forceinline__ void plan_tile(const PlArgs& a, const Tile& t, int wv, int lane, int (&goff)[IN_PER_WAVE]) {
    const int hw = a.h * a.w;
#pragma unroll
    for (int k = 0; k < IN_PER_WAVE; ++k) {
        const int slot = wv + NWAVE * k;
        const int plane = slot / IN_SEG, seg = slot - plane * IN_SEG;
        const int idx = seg * 64 + lane;
        const int r = idx / IW, c = idx - r * IW;
        const int yy = wsu_reflect(t.y0 - 1 + r, a.h), xx = wsu_reflect(t.x0 - 1 + c, a.w);
        goff[k] = idx < NPIX ? plane * hw + yy * a.w + xx : -1;
    }
}

// LDS-DMA of chunk c of tile t into stage `st`: 40 input pieces + 36 weight pieces of 1 KiB, 9-10 per wave, nothing waits here
__device__ __forceinline__ void issue_dma(const PlArgs& a, int tn, int tcb, int c, char* st, int wv, int lane, const int (&goff)[IN_PER_WAVE]) {
    const size_t plane4 = (size_t)a.h * a.w * 64;                       // bytes of one chunk of one image (4 planes)
    const char* src = c < a.nch1 ? a.x1 + ((size_t)tn * a.nch1 + c) * plane4
                                 : a.x2 + ((size_t)tn * (a.nch - a.nch1) + (c - a.nch1)) * plane4;
#pragma unroll
    for (int k = 0; k < IN_PER_WAVE; ++k) {
        const int slot = wv + NWAVE * k;
        const int plane = slot / IN_SEG, seg = slot - plane * IN_SEG;
        if (goff[k] >= 0)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + (size_t)goff[k] * 16), (lds_void*)(st + plane * PLANE + seg * 1024), 16, 0, 0);
    }
    const char* wsrc = a.wp + ((size_t)tcb * a.nch + c) * LDS_W + lane * 16;
#pragma unroll
    for (int k = 0; k < W_PER_WAVE; ++k) {
        const int slot = wv + NWAVE * k;
        if (slot < W_SLOTS)
            __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + slot * 1024), (lds_void*)(st + LDS_IN + slot * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ void swap32(uint32_t& upper_of, uint32_t& lower_of) {
    // lanes 32-63 of `upper_of` <-> lanes 0-31 of `lower_of`
    const auto r = __builtin_amdgcn_permlane32_swap(upper_of, lower_of, false, false);
    upper_of = r[0]; lower_of = r[1];
}

__global__ __launch_bounds__(NT, 2) void conv3x3_pl_kernel(const PlArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int G = gridDim.x;
    const int lw = (int)wsu_xcd_remap(blockIdx.x, G);
    const int K = a.ntiles > lw ? (a.ntiles - lw + G - 1) / G : 0;           // tiles walked by this workgroup
    const int J = K * a.nch;                                                  // chunk steps

    float* s_bias = reinterpret_cast<float*>(smem + LDS_EXTRA);
    float* s_hw = s_bias + 1024;
    float* s_hb = s_hw + 4 * 64;
    for (int i = tid; i < a.cout; i += NT) s_bias[i] = a.bias ? a.bias[i] : 0.f;
    if (a.head_w) {
        for (int i = tid; i < a.head_cout * 64; i += NT) s_hw[i] = a.head_w[i];
        if (tid < 4) s_hb[tid] = (a.head_b && tid < a.head_cout) ? a.head_b[tid] : 0.f;
    }
    // the plain loads above must have retired before the first counted / zero vmcnt wait below means anything: they have, the values
    // were consumed by the LDS stores; the barrier of step 0 publishes them

    int goff[IN_PER_WAVE];
    Tile cur = tile_of(a, lw), nxt = cur;
    if (J > 0) {
        plan_tile(a, cur, wv, lane, goff);
        issue_dma(a, cur.n, cur.cb, 0, smem, wv, lane, goff);
    }

    f32x16 acc[2][2];
    const int sc_a = hh ? WSU_F8_SCALE_WLO : WSU_F8_SCALE_W, sc_b = hh ? WSU_F8_SCALE_X : WSU_F8_SCALE_XLO;
    int c = 0, kt = 0;                                                        // chunk inside the tile, tile counter
    for (int j = 0; j < J; ++j) {
        // ---- step j: its DMA (issued one step ago) has had a whole matrix section to land -------------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                         // everyone's pieces landed; everyone left the other stage
        asm volatile("" ::: "memory");
        char* st = smem + (j & 1) * STAGE;
        if (j + 1 < J) {
            const int cn = c + 1 == a.nch ? 0 : c + 1;
            if (cn == 0) {
                nxt = tile_of(a, lw + (kt + 1) * G);
                plan_tile(a, nxt, wv, lane, goff);
            }
            issue_dma(a, cn == 0 ? nxt.n : cur.n, cn == 0 ? nxt.cb : cur.cb, cn, smem + ((j + 1) & 1) * STAGE, wv, lane, goff);
        }
        if (c == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
        }
        // ---- matrix section: identical arithmetic (and accumulation order) to conv3x3_kernel<F16F8> -----------------------------
        const char* ldsA = st + LDS_IN + l31 * 16;                            // + ((tap*4 + g)*64 + m*32)*16
        const char* ldsB = st + ((2 * wv) * IW + l31) * 16;                   // + g*PLANE + ((q+dy)*IW + dx)*16
        auto cross = [&](auto tp_c) __attribute__((always_inline)) {
            constexpr int tp = decltype(tp_c)::value;
            constexpr int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 2 * tp;
            constexpr bool single = 2 * tp + 1 >= 9;
            const int aoff = ((hh ? t1 : t0) * 4 + 2) * 64 * 16;
            const int boff = 2 * PLANE + (hh ? ((t1 / 3) * IW + t1 % 3) : ((t0 / 3) * IW + t0 % 3)) * 16;
            u32x4 a0[2], a1[2], b0[2], b1[2];
_Pragma("unroll")
            for (int m = 0; m < 2; ++m) {
                a0[m] = *reinterpret_cast<const u32x4*>(ldsA + aoff + m * 32 * 16);
                a1[m] = *reinterpret_cast<const u32x4*>(ldsA + aoff + 64 * 16 + m * 32 * 16);
            }
_Pragma("unroll")
            for (int q = 0; q < 2; ++q) {
                b0[q] = *reinterpret_cast<const u32x4*>(ldsB + boff + q * IW * 16);
                b1[q] = *reinterpret_cast<const u32x4*>(ldsB + boff + PLANE + q * IW * 16);
            }
            if (single && hh) {
                const u32x4 z = mk_u4(0, 0, 0, 0);
_Pragma("unroll")
                for (int m = 0; m < 2; ++m) { a0[m] = z; a1[m] = z; }
                b0[0] = z; b0[1] = z; b1[0] = z; b1[1] = z;
            }
_Pragma("unroll")
            for (int m = 0; m < 2; ++m)
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) wsu_mfma_f8x2(a0[m], a1[m], b0[q], b1[q], sc_a, sc_b, acc[m][q]);
        };
        auto main_term = [&](auto tap_c) __attribute__((always_inline)) {
            constexpr int tap = decltype(tap_c)::value, dy = tap / 3, dx = tap % 3;
            u32x4 ah[2], bh[2];
_Pragma("unroll")
            for (int m = 0; m < 2; ++m) ah[m] = *reinterpret_cast<const u32x4*>(ldsA + ((tap * 4 + hh) * 64 + m * 32) * 16);
_Pragma("unroll")
            for (int q = 0; q < 2; ++q) bh[q] = *reinterpret_cast<const u32x4*>(ldsB + hh * PLANE + ((q + dy) * IW + dx) * 16);
_Pragma("unroll")
            for (int m = 0; m < 2; ++m)
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) wsu_mfma_f16(ah[m], bh[q], acc[m][q]);
        };
        WSU_STATIC_FOR(5, tp, {
            cross(std::integral_constant<int, tp>{});
            main_term(std::integral_constant<int, 2 * tp>{});
            if constexpr (2 * tp + 1 < 9) main_term(std::integral_constant<int, 2 * tp + 1>{});
        });

        // ---- epilogue of the tile: accumulators -> planar global memory ---------------------------------------------------------
        if (c + 1 == a.nch) {
            const int col = cur.x0 + l31;
            const size_t hw = (size_t)a.h * a.w;
            const int nco = a.cout >> 4;                                       // output chunks
            float hz[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int o = 0; o < 4; ++o) hz[q][o] = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {                               // 16 output channels = accumulator groups g4 = 2cp, 2cp+1
                    const int oc = cur.cb * 4 + m * 2 + cp;
                    const int co0 = oc * 16 + 4 * hh;                          // this lane: channels co0..co0+3 (X) and co0+8..co0+11 (Y)
                    const f32x4 bx = *reinterpret_cast<const f32x4*>(s_bias + co0), by = *reinterpret_cast<const f32x4*>(s_bias + co0 + 8);
                    f32x4 vx[2], vy[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = acc[m][q][8 * cp + e] + bx[e], y = acc[m][q][8 * cp + 4 + e] + by[e];
                            if (a.relu) { x = fmaxf(x, 0.f); y = fmaxf(y, 0.f); }
                            vx[q][e] = x; vy[q][e] = y;
                        }
                    }
                    if (a.head_w) {
                        const int lc = (m * 2 + cp) * 16 + 4 * hh;             // channel inside the 64-wide block
#pragma unroll
                        for (int o = 0; o < 4; ++o)
                            if (o < a.head_cout)
#pragma unroll
                                for (int q = 0; q < 2; ++q)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        hz[q][o] = fmaf(vx[q][e], s_hw[o * 64 + lc + e], fmaf(vy[q][e], s_hw[o * 64 + lc + 8 + e], hz[q][o]));
                    }
                    auto store_px = [&](const f32x4& X, const f32x4& Y, char* dst, size_t plane_bytes, bool ok) __attribute__((always_inline)) {
                        uint32_t xh0, xh1, xlo, xx8, yh0, yh1, ylo, yx8;
                        wsu_split4_f16f8(X, WSU_F8_XLO_DIV, WSU_F8_X_DIV, xh0, xh1, xlo, xx8);
                        wsu_split4_f16f8(Y, WSU_F8_XLO_DIV, WSU_F8_X_DIV, yh0, yh1, ylo, yx8);
                        swap32(xh0, yh0); swap32(xh1, yh1);                     // lanes 0-31: f16 ch 0-7, lanes 32-63: f16 ch 8-15
                        swap32(xlo, xx8); swap32(ylo, yx8);                     // lanes 0-31: residuals ch 0-15, lanes 32-63: e4m3 copies
                        if (ok) {
                            *reinterpret_cast<u32x4*>(dst + hh * plane_bytes) = mk_u4(xh0, xh1, yh0, yh1);
                            *reinterpret_cast<u32x4*>(dst + (2 + hh) * plane_bytes) = mk_u4(xlo, xx8, ylo, yx8);
                        }
                    };
                    if (a.y) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int row = cur.y0 + 2 * wv + q;
                            char* dst = a.y + ((((size_t)cur.n * nco + oc) * 4) * hw + (size_t)row * a.w + col) * 16;
                            store_px(vx[q], vy[q], dst, hw * 16, row < a.h && col < a.w);
                        }
                    }
                    if (a.ypool) {                                             // wave-uniform: every lane takes part in the exchanges
                        f32x4 px, py;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {                           // window order (0,0) (0,1) (1,0) (1,1); max is order-free, NaN propagates
                            const float x1 = __shfl_xor(vx[0][e], 1, 64), x3 = __shfl_xor(vx[1][e], 1, 64);
                            const float y1 = __shfl_xor(vy[0][e], 1, 64), y3 = __shfl_xor(vy[1][e], 1, 64);
                            float bxv = vx[0][e], byv = vy[0][e];
                            if (x1 > bxv || x1 != x1) bxv = x1;
                            if (vx[1][e] > bxv || vx[1][e] != vx[1][e]) bxv = vx[1][e];
                            if (x3 > bxv || x3 != x3) bxv = x3;
                            if (y1 > byv || y1 != y1) byv = y1;
                            if (vy[1][e] > byv || vy[1][e] != vy[1][e]) byv = vy[1][e];
                            if (y3 > byv || y3 != y3) byv = y3;
                            px[e] = bxv; py[e] = byv;
                        }
                        const int hp = a.h >> 1, wp2 = a.w >> 1;
                        const int gy = (cur.y0 >> 1) + wv, gx = (cur.x0 >> 1) + (l31 >> 1);
                        char* dst = a.ypool + ((((size_t)cur.n * nco + oc) * 4) * hp * wp2 + (size_t)gy * wp2 + gx) * 16;
                        store_px(px, py, dst, (size_t)hp * wp2 * 16, !(l31 & 1) && gy < hp && gx < wp2);
                    }
                }
            }
            if (a.head_w) {
                // the other 32 channels of this pixel sit in the partner lane (lane ^ 32)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = cur.y0 + 2 * wv + q;
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o < a.head_cout) {
                            const float z = hz[q][o] + __shfl_xor(hz[q][o], 32, 64) + s_hb[o];
                            if (!hh && row < a.h && col < a.w) {
                                const size_t off = ((size_t)cur.n * a.head_cout + o) * hw + (size_t)row * a.w + col;
                                if (a.head_logit) a.head_logit[off] = z;
                                a.head_out[off] = 1.f / (1.f + expf(-z));
                            }
                        }
                }
            }
            cur = nxt;
            ++kt;
            c = 0;
        } else {
            ++c;
        }
    }
}

}  // namespace

extern "C" {

// Forward 3x3 reflect conv + bias + ReLU on planar F16F8P activations (layout: wsu.h).  x1 (c1 channels) and optional x2 (c2, fused
// concat), packed weights of wsu_conv3x3_pack(mode F16F8); outputs, each optional: y (cout channels, planar), y_pool (2x2 max-pooled,
// planar), head (1x1 conv + sigmoid on the 64 output channels: out / logit NCHW fp32; needs cout == 64).  c1, c2 multiples of 16,
// cout of 64.  Asynchronous on `stream`; allocates nothing.
int wsu_conv3x3_pl_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias, void* y, void* y_pool,
                       const float* head_w, const float* head_b, float* head_out, float* head_logit, int head_cout,
                       int n, int h, int w, int c1, int c2, int cout, int relu, void* stream) {
    WSU_REQUIRE(x1 && w_packed && (y || y_pool || head_w), "conv3x3_pl: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2, "conv3x3_pl: bad shape n=%d h=%d w=%d (reflect pad 1 needs h,w >= 2)", n, h, w);
    WSU_REQUIRE(c1 > 0 && c1 % 16 == 0 && c2 >= 0 && c2 % 16 == 0 && (c2 == 0) == (x2 == nullptr), "conv3x3_pl: c1=%d c2=%d must be multiples of 16", c1, c2);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0 && cout <= 1024, "conv3x3_pl: cout=%d must be a multiple of %d (<= 1024)", cout, WSU_COB);
    WSU_REQUIRE(!head_w || (head_out && cout == WSU_COB && head_cout >= 1 && head_cout <= 4), "conv3x3_pl: fused head needs cout == %d and 1..4 head planes", WSU_COB);
    WSU_REQUIRE(!y_pool || (h % 2 == 0 && w % 2 == 0), "conv3x3_pl: fused pool needs even h, w");
    WSU_REQUIRE((long long)h * w * 4 < 0x7FFFFFFFLL, "conv3x3_pl: h*w too large");
    PlArgs a;
    a.x1 = (const char*)x1; a.x2 = (const char*)x2; a.wp = (const char*)w_packed; a.bias = bias;
    a.y = (char*)y; a.ypool = (char*)y_pool;
    a.head_w = head_w; a.head_b = head_b; a.head_out = head_out; a.head_logit = head_logit; a.head_cout = head_cout;
    a.n = n; a.h = h; a.w = w; a.c1 = c1; a.c2 = c2; a.cout = cout;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ncb = cout / WSU_COB;
    a.nch1 = c1 / 16; a.nch = (c1 + c2) / 16; a.relu = relu;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "conv3x3_pl: %lld tiles out of range", nt);
    a.ntiles = (int)nt;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            wsu_set_error("conv3x3_pl: cannot query the device"); return WSU_ERR_HIP;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3_pl): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        ncu = prop.multiProcessorCount;
    }
    const int grid = (int)(nt < ncu ? nt : ncu);
    hipLaunchKernelGGL(conv3x3_pl_kernel, dim3(grid), dim3(NT), LDS_TOTAL, static_cast<hipStream_t>(stream), a);
    return wsu_check_launch("conv3x3_pl_kernel");
}

}  // extern "C"
