// K1u (round 4): the first conv of a decoder block TOGETHER with the transposed conv in front of it -- xd = relu(conv3x3(cat[convT2x2(x_low), skip]))
// (src/unet/model/unet.py:171-173, 177-179, 183-185) in ONE launch, in the arithmetic of the default inference mode 'f16f4p' (conv3x3_q.hip).
// There is no non-linearity between nn.ConvTranspose2d(k2, s2) and the 3x3 conv, so their composition is linear in x_low:
//   * an output pixel (2i + py, 2j + px) of PARITY CLASS (py, px) sees, through its 3x3 window on the upsampled tensor, exactly 2 x 2 pixels of
//     x_low -- rows i - 1 + py + dy, columns j - 1 + px + dx (dy, dx in {0, 1}) -- and the weights of that 2x2-tap conv are the products
//     Wc[py,px][dy,dx][co][c] = sum_{ci} sum_{(ky,kx) -> (dy,dx)} w3[co][ci][ky][kx] * wT[c][ci][sy][sx]  (pack kernel below, fp32);
//   * the reflect padding of the 3x3 conv at the high resolution is CLAMP padding of x_low (xu[-1] = xu[1] = the sub-position 1 of low row 0);
//   * the transposed conv's bias passes through all nine taps everywhere (reflect padding has no missing tap): one combined bias per co.
// So the upsampled half costs 4 taps x C_low = 512 multiply-adds per output and channel pair instead of 9 x C_low/2 = 576, the transposed conv
// (its launch, its 2.56 B/element output written and read back) disappears, and `xu` never exists.  The skip half is the ordinary 3x3 conv.
//
// Structure: the persistent workgroup of conv3x3_q.hip (one per CU, 16 x 32-pixel x 64-co tiles, 4 pure-DMA loader waves + 8 matrix waves, one
// s_barrier per step), with two differences forced by the parity classes:
//   * a matrix wave owns ONE class: wave = (class, half of the tile's 8 low rows), its two 32-pixel matrix tiles are 2 low rows x 16 low columns
//     of that class (so all 32 pixels of a matrix instruction share the combined weights).  The skip half's halo tile is therefore staged as four
//     class planes of 9 x 17 pixels -- the loaders' DMA gathers every other pixel of a row (per-lane source address, contiguous LDS
//     destination) -- and lanes 16-31 of a matrix tile hold their row's columns rotated by one, which makes the 16-byte fragment reads of two
//     rows 272 B (528 B in the low tile) apart conflict free (lane groups of ds_read_b128, MI355X guide, LDS);
//   * steps differ in size (a skip chunk: 9 taps, 28 KB of weights, 31 KB of input; half a low chunk: 2 taps x 4 classes, 25 KB of weights; a
//     low chunk's input tile: 17 KB, shared by its two half-steps), so LDS is ONE ring of five 31.5 KB slots that inputs and weights are
//     allocated from in step order; the loaders run as far ahead as the ring allows (2-3 allocations beyond the next step) and count
//     their outstanding DMA instructions per allocation (`s_waitcnt vmcnt(n)`).
// Accuracy: Wc is formed in fp32 and then split like any weight (f16 + fp4 residual terms); `xu` is never rounded to storage -- the fused
// result is closer to the exact composition than the two-kernel path.  Replaces ops.convt2x2_pl + ops.conv3x3_q in UNet._forward_planar.
#include "wsu_device.h"
#include <cstdlib>

namespace {

constexpr int TW = 32, TH = 16;
// skip half: the 18 x 34 halo tile as class planes [row parity][column parity][9][17]
constexpr int CW = 17, CH = 9, CPIX = CW * CH;            // 153
constexpr int NPIX_S = 4 * CPIX;                          // 612
constexpr int PLANE_S = NPIX_S * 16;                      // 9792
constexpr int SEG_S = (NPIX_S + 63) / 64;                 // 10
constexpr int IN_S = 3 * PLANE_S + SEG_S * 256;           // 31936: f16 ch 0-7 | f16 ch 8-15 | Q | one dword slot per pixel for the scale byte
// low half: (8 + 2) x (16 + 2) low-resolution pixels at a row pitch of 33 (528 B = 16 mod 256, like the class planes' 272 B)
constexpr int LP = 33, LH = TH / 2 + 2, LW_ = TW / 2 + 2;
constexpr int NPIX_L = LP * LH;                           // 330
constexpr int PLANE_L = NPIX_L * 16;                      // 5280
constexpr int SEG_L = (NPIX_L + 63) / 64;                 // 6
constexpr int IN_L = 3 * PLANE_L + SEG_L * 256;           // 17376
constexpr int W_GRAN_S = 9 * 3 * WSU_COB * 16;            // 27648
constexpr int W_S = W_GRAN_S + 1024;                      // the (block, chunk) slice of wsu_conv3x3_pack_f4
constexpr int W_UNITS_L = 8;                              // (class, dx) units of half a low chunk
constexpr int W_GRAN_L = W_UNITS_L * 3 * WSU_COB * 16;    // 24576
constexpr int W_L = W_GRAN_L + 1024;                      // + [8][64] scale bytes, padded to a DMA piece
constexpr int PIECES_S = W_S / 1024, PIECES_L = W_L / 1024;   // 28, 25
constexpr int NSLOT = 5, SLOT = 32256;
constexpr int LDS_BIAS = NSLOT * SLOT;                    // 161280
constexpr int MAX_COUT = 512;
constexpr int LDS_TOTAL = LDS_BIAS + MAX_COUT * 4;        // 163328
static_assert(SLOT >= IN_S && SLOT >= IN_L && SLOT >= W_L && SLOT >= W_S && SLOT % 16 == 0 && LDS_TOTAL <= 160 * 1024, "LDS budget");
constexpr int NLOAD = 4, NWAVE = 8, NT = (NWAVE + NLOAD) * 64;
constexpr unsigned OOB = 0xFFFFFFF0u;

struct UArgs {
    const char* xl; const char* xs; const char* wps; const char* wpl; const float* bias;
    char* y;
    int n, h, w, hl, wl, cl, c2, cout;
    int tiles_x, tiles_y, ltiles_x, ncb, nchS, nchL;
    int relu, ntiles;
    unsigned* range_flag;
    int ablate;                                           // timing-only experiments (WSU_QU_ABLATE bits; results wrong when != 0): 1 = no DMA after the first allocations, 2 = no epilogue; 4 = loader waves at priority 3 (results right), 8 = low inputs from one cached tile, 16 = half of the low weights' pieces
};

struct Tile { int n, y0, x0, cb; };
__device__ __forceinline__ Tile tile_of(const UArgs& a, int t) {
    Tile r;
    r.cb = t % a.ncb; t /= a.ncb;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    r.n = t / a.tiles_y; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

// `s_waitcnt vmcnt(n)` for a run-time n (the counter is 6 bits wide)
__device__ __forceinline__ void wait_vm(int n) {
#define WSU_VM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        WSU_VM_CASE(0) WSU_VM_CASE(1) WSU_VM_CASE(2) WSU_VM_CASE(3) WSU_VM_CASE(4) WSU_VM_CASE(5) WSU_VM_CASE(6) WSU_VM_CASE(7) WSU_VM_CASE(8) WSU_VM_CASE(9)
        WSU_VM_CASE(10) WSU_VM_CASE(11) WSU_VM_CASE(12) WSU_VM_CASE(13) WSU_VM_CASE(14) WSU_VM_CASE(15) WSU_VM_CASE(16) WSU_VM_CASE(17) WSU_VM_CASE(18) WSU_VM_CASE(19)
        WSU_VM_CASE(20) WSU_VM_CASE(21) WSU_VM_CASE(22) WSU_VM_CASE(23) WSU_VM_CASE(24) WSU_VM_CASE(25) WSU_VM_CASE(26) WSU_VM_CASE(27) WSU_VM_CASE(28) WSU_VM_CASE(29)
        WSU_VM_CASE(30) WSU_VM_CASE(31) WSU_VM_CASE(32) WSU_VM_CASE(33) WSU_VM_CASE(34) WSU_VM_CASE(35) WSU_VM_CASE(36) WSU_VM_CASE(37) WSU_VM_CASE(38) WSU_VM_CASE(39)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;       // (never: three allocations of a wave are at most 33 instructions)
    }
#undef WSU_VM_CASE
}

// Steps of a tile: nchS skip chunks (kind S), then per low chunk the half-steps L0 (dy = 0) and L1 (dy = 1).  Allocations, in ring order:
// S: input, weights;  L0: weights (dy 0), input;  L1: weights (dy 1).  Everything of a step dies with the step except a low chunk's input (dies
// with L1): lifetimes end in allocation order, so "freed" is a prefix of the ring.
__device__ __forceinline__ int step_allocs(const UArgs& a, int s) { return (s >= a.nchS && ((s - a.nchS) & 1)) ? 1 : 2; }
__device__ __forceinline__ bool step_is_l0(const UArgs& a, int s) { return s >= a.nchS && !((s - a.nchS) & 1); }

// ================= loader wave LW: pure DMA, walks the allocation sequence ahead of the matrix waves ==========================================
template <int LW>
__device__ __forceinline__ void u_loader(const UArgs& a, char* smem, int lane, int lw, int G, int K) {
    constexpr int NSS = LW < 2 ? 3 : 2;                   // skip-input segments LW, LW + 4, LW + 8 (< 10)
    constexpr int NSL = LW < 2 ? 2 : 1;                   // low-input segments LW, LW + 4 (< 6)
    constexpr int OPS_INS = 4 * NSS, OPS_INL = 4 * NSL;
    constexpr int WS0 = LW < 2 ? LW * 5 : 10 + (LW - 2) * 9, NWS = LW < 2 ? 5 : 9;
    constexpr int WL0 = LW < 2 ? LW * 5 : (LW == 2 ? 10 : 17), NWL = LW < 2 ? 5 : (LW == 2 ? 7 : 8);
    static_assert(2 * 5 + 2 * 9 == PIECES_S && 2 * 5 + 7 + 8 == PIECES_L, "weight pieces over the loader waves");
    lds_char* smem3 = (lds_char*)smem;
    const unsigned hw16 = (unsigned)(a.h * a.w) * 16u, hwl16 = (unsigned)(a.hl * a.wl) * 16u;
    const unsigned cbytes_s = (unsigned)wsu_q_chunk_bytes(a.h, a.w), cbytes_l = (unsigned)wsu_q_chunk_bytes(a.hl, a.wl);
    const int T = a.nchS + 2 * a.nchL, A = 2 * a.nchS + 3 * a.nchL;
    const int J = K * T, total = K * A;
    if (J <= 0) return;
    unsigned voS[NSS], soS[NSS], voL[NSL], soL[NSL];
    auto plan = [&](const Tile& t) __attribute__((always_inline)) {
        WSU_STATIC_FOR(NSS, k, {
            const int idx = min((LW + NLOAD * k) * 64 + lane, NPIX_S - 1);
            const int p = idx / CPIX, rem = idx - p * CPIX;
            const int lr = rem / CW, lc = rem - lr * CW;
            int yy = wsu_reflect(t.y0 - 1 + 2 * lr + (p >> 1), a.h), xx = wsu_reflect(t.x0 - 1 + 2 * lc + (p & 1), a.w);
            if (a.ablate & 32) { yy = wsu_reflect(t.y0 - 1 + idx / 34, a.h); xx = wsu_reflect(t.x0 - 1 + idx % 34, a.w); }      // (32: the natural row-major tile -- no gather; results wrong)
            voS[k] = (unsigned)(yy * a.w + xx) * 16u;
            soS[k] = 3u * hw16 + wsu_q_soff(yy, xx, a.tiles_x);
        });
        WSU_STATIC_FOR(NSL, k, {
            const int idx = min((LW + NLOAD * k) * 64 + lane, NPIX_L - 1);
            const int r = idx / LP, c = min(idx - r * LP, LW_ - 1);                 // (columns 18..32 of a row are padding: any valid pixel)
            const int yy = min(max((t.y0 >> 1) - 1 + r, 0), a.hl - 1), xx = min(max((t.x0 >> 1) - 1 + c, 0), a.wl - 1);
            voL[k] = (unsigned)(yy * a.wl + xx) * 16u;
            soL[k] = 3u * hwl16 + wsu_q_soff(yy, xx, a.ltiles_x);
        });
    };
    // ---- the allocation cursor
    int a_kt = 0, a_c = 0, a_r = 0; bool a_low = false;
    Tile at = tile_of(a, lw);
    int issued = 0, slot_off = 0, n1 = 0, n2 = 0, n3 = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
        lds_char* slot = smem3 + slot_off;
        int nops;
        const bool is_in = a_low ? a_r == 1 : a_r == 0;
        const int skip_kind = a.ablate >> 8;                                 // 0x100 IN_S, 0x200 W_S, 0x400 IN_L, 0x800 W_L: that kind is not fetched (timing only)
        if (skip_kind & (is_in ? (a_low ? 4 : 1) : (a_low ? 8 : 2))) {
            nops = 0;
        } else if (is_in && !a_low) {
            const char* src = a.xs + ((size_t)at.n * a.nchS + a_c) * cbytes_s;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)cbytes_s, 0x00020000);
            WSU_STATIC_FOR(NSS, k, {
                constexpr int seg = LW + NLOAD * k;
                if (seg < SEG_S - 1 || lane < NPIX_S - (SEG_S - 1) * 64) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + seg * 1024), 16, voS[k], 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + PLANE_S + seg * 1024), 16, voS[k], (int)hw16, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + 2 * PLANE_S + seg * 1024), 16, voS[k], (int)(2u * hw16), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + 3 * PLANE_S + seg * 256), 1, soS[k], 0, 0, 0);
                }
            });
            nops = OPS_INS;
        } else if (is_in) {
            const char* src = (a.ablate & 8) ? a.xl : a.xl + ((size_t)at.n * a.nchL + a_c) * cbytes_l;      // (8: every low input from image 0, chunk 0 -- cache hits)
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)cbytes_l, 0x00020000);
            WSU_STATIC_FOR(NSL, k, {
                constexpr int seg = LW + NLOAD * k;
                if (seg < SEG_L - 1 || lane < NPIX_L - (SEG_L - 1) * 64) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + seg * 1024), 16, voL[k], 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + PLANE_L + seg * 1024), 16, voL[k], (int)hwl16, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + 2 * PLANE_L + seg * 1024), 16, voL[k], (int)(2u * hwl16), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + 3 * PLANE_L + seg * 256), 1, soL[k], 0, 0, 0);
                }
            });
            nops = OPS_INL;
        } else if (!a_low) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wps), 0, 0x7FFFFFF0, 0x00020000);
            const int base = (at.cb * a.nchS + a_c) * W_S;
            WSU_STATIC_FOR(NWS, k, {
                constexpr int piece = WS0 + k;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + piece * 1024), 16, (unsigned)lane * 16u, base + piece * 1024, 0, 0);
            });
            nops = NWS;
        } else {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wpl), 0, 0x7FFFFFF0, 0x00020000);
            const int base = ((at.cb * a.nchL + a_c) * 2 + (a_r == 2 ? 1 : 0)) * W_L;
            if (a.ablate & 16) {                                          // (16: half of the low weights' pieces)
                WSU_STATIC_FOR((NWL + 1) / 2, k, {
                    constexpr int piece = WL0 + 2 * k;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + piece * 1024), 16, (unsigned)lane * 16u, base + piece * 1024, 0, 0);
                });
                nops = (NWL + 1) / 2;
            } else {
                WSU_STATIC_FOR(NWL, k, {
                    constexpr int piece = WL0 + k;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(slot + piece * 1024), 16, (unsigned)lane * 16u, base + piece * 1024, 0, 0);
                });
                nops = NWL;
            }
        }
        n3 = n2; n2 = n1; n1 = nops;
        ++issued;
        slot_off += SLOT; if (slot_off == NSLOT * SLOT) slot_off = 0;
        // advance the cursor
        ++a_r;
        if (!a_low) {
            if (a_r == 2) { a_r = 0; if (++a_c == a.nchS) { a_c = 0; a_low = true; } }
        } else if (a_r == 3) { a_r = 0; ++a_c; }
        if (a_low && a_c == a.nchL) {                                     // (nchL == 0 never happens: the entry point requires a low half)
            a_low = false; a_c = 0; a_r = 0;
            if (++a_kt < K) { at = tile_of(a, lw + a_kt * G); plan(at); }
        }
    };
    auto wait_but = [&](int x) __attribute__((always_inline)) {          // everything has landed except the x youngest allocations
        wait_vm(x <= 0 ? 0 : x == 1 ? n1 : x == 2 ? n1 + n2 : n1 + n2 + n3);
    };
    plan(at);
    {
        const int lim = min(NSLOT, total);
        while (issued < lim) issue_next();
        wait_but(issued - step_allocs(a, 0));
    }
    int s = 0, cb = 0;                                                    // step inside the tile; allocations begun at steps < j
    for (int j = 0; ; ++j) {
        __builtin_amdgcn_s_barrier();                                     // barrier j: step j is complete in LDS; every matrix wave has left step j - 1
        asm volatile("" ::: "memory");
        if (j + 1 >= J) break;
        const int freed = cb - ((s > 0 && step_is_l0(a, s - 1)) ? 1 : 0);
        const int lim = min(freed + NSLOT, total);
        const int s1 = s + 1 == T ? 0 : s + 1;
        if (!(a.ablate & 1)) {
            while (issued < lim) issue_next();
            const int need = cb + step_allocs(a, s) + step_allocs(a, s1);     // allocations of the steps <= j + 1
            wait_but(issued - need);
        }
        cb += step_allocs(a, s);
        s = s1;
    }
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3)))
void conv3x3_qu_kernel(const UArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int G = gridDim.x;
    const int lw = (int)wsu_xcd_remap(blockIdx.x, G);
    const int K = a.ntiles > lw ? (a.ntiles - lw + G - 1) / G : 0;          // tiles walked by this workgroup
    float* s_bias = reinterpret_cast<float*>(smem + LDS_BIAS);
    for (int i = tid; i < a.cout; i += NT) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    if (wv >= NWAVE) {
        if (a.ablate & 4) __builtin_amdgcn_s_setprio(3);                       // experiment: the loaders' DMA instructions issue ahead of the matrix waves' streams
        switch (wv - NWAVE) {
            case 0: u_loader<0>(a, smem, lane, lw, G, K); break;
            case 1: u_loader<1>(a, smem, lane, lw, G, K); break;
            case 2: u_loader<2>(a, smem, lane, lw, G, K); break;
            default: u_loader<3>(a, smem, lane, lw, G, K); break;
        }
        return;
    }

    // ================= matrix waves ===================================================================================================
    const int cls = wv & 3, py = cls >> 1, px = cls & 1, rh = wv >> 2;      // wave-uniform
    const int l31 = lane & 31, hh = lane >> 5;
    const int lrow = l31 >> 4, lcol = lrow ? ((l31 + 15) & 15) : l31;       // lanes 16-31: the next low row, columns rotated by one (bank-conflict-free fragment reads)
    const unsigned laneS = (unsigned)(((4 * rh + lrow) * CW + lcol) * 16), laneL = (unsigned)(((4 * rh + lrow) * LP + lcol) * 16);
    constexpr unsigned QS = 2 * CW * 16, QL = 2 * LP * 16;                   // second matrix tile: two low rows further
    Tile cur = tile_of(a, lw);
    f32x16 acc[2][2];                                                       // [32-channel half][matrix tile]
    unsigned ring = 0;                                                       // byte offset of the next allocation's slot
    unsigned in_off = 0, w_off = 0;
    auto next_slot = [&]() __attribute__((always_inline)) { const unsigned r = ring; ring += SLOT; if (ring == NSLOT * SLOT) ring = 0; return r; };
    int hh_q = hh;
    typedef __attribute__((address_space(3))) const unsigned char lds_cuchar;
    typedef __attribute__((address_space(3))) const int lds_cint;
    typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
    lds_char* L = (lds_char*)smem;
    auto sync_step = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        hh_q = hh;
        asm volatile("" : "+v"(hh_q));
    };
    // byte offset of tap (ky, kx)'s source pixel relative to the lane's own class pixel, in the class planes of the skip tile
    auto tap_s = [&](int ky, int kx) __attribute__((always_inline)) {
        return (unsigned)(((((py + ky) & 1) * 2 + ((px + kx) & 1)) * CPIX + ((py + ky) >> 1) * CW + ((px + kx) >> 1)) * 16);
    };
    // ---- a skip chunk: the nine taps of the ordinary conv, operands of conv3x3_q.hip (f16 products per tap, both cross terms of a tap pair as one fp4 instruction)
    auto skip_units = [&]() __attribute__((always_inline)) {
        WSU_STATIC_FOR(5, tp, {
            constexpr int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 2 * tp;
            {
                const unsigned tapo = hh_q ? tap_s(t1 / 3, t1 % 3) : tap_s(t0 / 3, t0 % 3);
                const int tap = hh_q ? t1 : t0;
                u32x4 a4[2], b4[2]; int sa[2], sb[2];
                const unsigned wb = w_off + (unsigned)((tap * 3 + 2) * 64 + l31) * 16u, sab = w_off + W_GRAN_S + (unsigned)(tap * 64 + l31);
                const unsigned pb = tapo + laneS;
_Pragma("unroll")
                for (int m = 0; m < 2; ++m) { a4[m] = *(lds_cu32x4*)(L + wb + m * 512); sa[m] = *(lds_cuchar*)(L + sab + m * 32); }
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) {
                    b4[q] = *(lds_cu32x4*)(L + in_off + 2 * PLANE_S + pb + q * QS);
                    sb[q] = *(lds_cint*)(L + in_off + 3 * PLANE_S + ((pb + q * QS) >> 2));
                }
                if (2 * tp + 1 >= 9 && hh_q) {                            // the ninth tap has no partner: lanes 32-63 multiply zeros
                    const u32x4 z = mk_u4(0, 0, 0, 0);
                    a4[0] = z; a4[1] = z; b4[0] = z; b4[1] = z;
                }
_Pragma("unroll")
                for (int m = 0; m < 2; ++m)
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) wsu_mfma_q4(a4[m], b4[q], sa[m], sb[q], acc[m][q]);
            }
            WSU_STATIC_FOR((2 * tp + 1 < 9 ? 2 : 1), k, {
                constexpr int tap = 2 * tp + k;
                u32x4 ah[2], bh[2];
                const unsigned wb = w_off + (unsigned)((tap * 3) * 64 + l31) * 16u + (unsigned)hh * 1024u;
                const unsigned pb = in_off + (unsigned)hh * PLANE_S + tap_s(tap / 3, tap % 3) + laneS;
_Pragma("unroll")
                for (int m = 0; m < 2; ++m) ah[m] = *(lds_cu32x4*)(L + wb + m * 512);
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) bh[q] = *(lds_cu32x4*)(L + pb + q * QS);
_Pragma("unroll")
                for (int m = 0; m < 2; ++m)
_Pragma("unroll")
                    for (int q = 0; q < 2; ++q) wsu_mfma_f16(ah[m], bh[q], acc[m][q]);
            });
        });
    };
    // ---- half a low chunk: taps (dy, 0) and (dy, 1) of this wave's class -- units cls * 2 + dx of the slice [8][3 planes][64 co][16 B] + [8][64] scale bytes
    auto low_units = [&](int dy) __attribute__((always_inline)) {
        const unsigned rowo = (unsigned)(((py + dy) * LP + px) * 16);
        {
            const int u = cls * 2 + hh_q;                                   // lanes 32-63: dx = 1
            u32x4 a4[2], b4[2]; int sa[2], sb[2];
            const unsigned wb = w_off + (unsigned)((u * 3 + 2) * 64 + l31) * 16u, sab = w_off + W_GRAN_L + (unsigned)(u * 64 + l31);
            const unsigned pb = rowo + (unsigned)hh_q * 16u + laneL;
_Pragma("unroll")
            for (int m = 0; m < 2; ++m) { a4[m] = *(lds_cu32x4*)(L + wb + m * 512); sa[m] = *(lds_cuchar*)(L + sab + m * 32); }
_Pragma("unroll")
            for (int q = 0; q < 2; ++q) {
                b4[q] = *(lds_cu32x4*)(L + in_off + 2 * PLANE_L + pb + q * QL);
                sb[q] = *(lds_cint*)(L + in_off + 3 * PLANE_L + ((pb + q * QL) >> 2));
            }
_Pragma("unroll")
            for (int m = 0; m < 2; ++m)
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) wsu_mfma_q4(a4[m], b4[q], sa[m], sb[q], acc[m][q]);
        }
        WSU_STATIC_FOR(2, dx, {
            u32x4 ah[2], bh[2];
            const unsigned wb = w_off + (unsigned)(((cls * 2 + dx) * 3) * 64 + l31) * 16u + (unsigned)hh * 1024u;
            const unsigned pb = in_off + (unsigned)hh * PLANE_L + rowo + dx * 16u + laneL;
_Pragma("unroll")
            for (int m = 0; m < 2; ++m) ah[m] = *(lds_cu32x4*)(L + wb + m * 512);
_Pragma("unroll")
            for (int q = 0; q < 2; ++q) bh[q] = *(lds_cu32x4*)(L + pb + q * QL);
_Pragma("unroll")
            for (int m = 0; m < 2; ++m)
_Pragma("unroll")
                for (int q = 0; q < 2; ++q) wsu_mfma_f16(ah[m], bh[q], acc[m][q]);
        });
    };

    // ---- epilogue of the tile: bias, ReLU, planar Q encoding (the producing epilogue of conv3x3_q.hip), straight from the accumulators.  This
    // lane's pixels: matrix tile q -> (y0 + 2 (4 rh + 2 q + lrow) + py, x0 + 2 lcol + px)
    auto finish_tile = [&]() __attribute__((always_inline)) {
        int l31o = l31, hho = hh, lro = lrow, lco = lcol;
        asm volatile("" : "+v"(l31o), "+v"(hho), "+v"(lro), "+v"(lco));   // per-tile copies (see conv3x3_q.hip: hoisted lane values were spilled)
        const unsigned hw16 = (unsigned)(a.h * a.w) * 16u;
        const unsigned cbytes = (unsigned)wsu_q_chunk_bytes(a.h, a.w);
        const int nco = a.cout >> 4;
        const int X = cur.x0 + 2 * lco + px;
        const int Y0 = cur.y0 + 2 * (4 * rh + lro) + py, Y1 = Y0 + 4;
        const bool ok0 = Y0 < a.h && X < a.w, ok1 = Y1 < a.h && X < a.w;
        const unsigned off0 = (unsigned)(Y0 * a.w + X) * 16u, off1 = (unsigned)(Y1 * a.w + X) * 16u;
        const unsigned so0 = wsu_q_soff(Y0, X, a.tiles_x), so1 = wsu_q_soff(Y1, X, a.tiles_x);
        const float relu_floor = a.relu ? 0.f : -__builtin_inff();
        float vmax = 0.f;
        WSU_STATIC_FOR(2, m, {
            WSU_STATIC_FOR(2, cp, {
                const int oc = cur.cb * 4 + m * 2 + cp;
                const int co0 = oc * 16 + 4 * hho;
                const f32x4 bx = *reinterpret_cast<const f32x4*>(s_bias + co0), by = *reinterpret_cast<const f32x4*>(s_bias + co0 + 8);
                f32x4 vx[2], vy[2];
_Pragma("unroll")
                for (int q = 0; q < 2; ++q)
_Pragma("unroll")
                    for (int e = 0; e < 4; ++e) {
                        const float x = fmaxf(acc[m][q][8 * cp + e] + bx[e], relu_floor), y = fmaxf(acc[m][q][8 * cp + 4 + e] + by[e], relu_floor);
                        vx[q][e] = x; vy[q][e] = y;
                        vmax = fmaxf(vmax, fmaxf(fabsf(x), fabsf(y)));
                    }
                u32x4 g0, g1; uint32_t dh0, dr0, sb0, dh1, dr1, sb1;
                wsu_q4_pre(vx[0], vy[0], g0, dh0, dr0, sb0);
                wsu_q4_pre(vx[1], vy[1], g1, dh1, dr1, sb1);
                const u32x4 qg = wsu_q4_pair(dh0, dr0, dh1, dr1);           // lanes 0-31: matrix tile 0's granule, lanes 32-63: tile 1's
                char* base = a.y + ((size_t)cur.n * nco + oc) * cbytes;
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)cbytes, 0x00020000);
                const unsigned hp = hho ? hw16 : 0u;
                __builtin_amdgcn_raw_buffer_store_b128(g0, rs, (int)(ok0 ? off0 + hp : OOB), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(g1, rs, (int)(ok1 ? off1 + hp : OOB), 0, 0);
                const bool okm = hho ? ok1 : ok0;
                __builtin_amdgcn_raw_buffer_store_b128(qg, rs, (int)(okm ? (hho ? off1 : off0) + 2u * hw16 : OOB), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(hho ? sb1 : sb0), rs, (int)(okm ? 3u * hw16 + (hho ? so1 : so0) : OOB), 0, 0);
            });
        });
        if (a.range_flag && __builtin_amdgcn_ballot_w64(!(vmax <= WSU_F8_RANGE)) != 0 && lane == 0) atomicOr(a.range_flag, 1u);
    };

    for (int t = 0; t < K; ++t) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
        for (int c = 0; c < a.nchS; ++c) {
            sync_step();
            in_off = next_slot(); w_off = next_slot();
            skip_units();
        }
        for (int c = 0; c < a.nchL; ++c) {
            sync_step();
            w_off = next_slot(); in_off = next_slot();
            low_units(0);
            sync_step();
            w_off = next_slot();
            low_units(1);
        }
        if (a.ablate & 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) asm volatile("" :: "v"(acc[m][q]));
        } else {
            finish_tile();
        }
        if (t + 1 < K) cur = tile_of(a, lw + (t + 1) * G);
    }
}

// ---- packing of the low half: one thread per (block, low chunk, dy, class, dx, co) forms its 16 combined weights in fp32 (ci outer, then ky, kx,
// fused multiply-adds) and encodes them like wsu_conv3x3_pack_f4 encodes a weight block.  wc_dense (optional): the fp32 values,
// [cout][cl][py][px][dy][dx] -- what the tests emulate the arithmetic on.
__global__ void pack_up_low_kernel(const float* __restrict__ w3, const float* __restrict__ wt, char* __restrict__ dst, float* __restrict__ wc_dense,
                                   int cup, int c2, int cl, int cout) {
    const int nchL = cl / 16, ctot = cup + c2;
    const long long total = (long long)(cout / WSU_COB) * nchL * 2 * W_UNITS_L * WSU_COB;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        long long t = d;
        const int co = (int)(t % WSU_COB); t /= WSU_COB;
        const int u = (int)(t % W_UNITS_L); t /= W_UNITS_L;
        const int dy = (int)(t & 1); t >>= 1;
        const int c = (int)(t % nchL); const int cb = (int)(t / nchL);
        const int cls = u >> 1, dx = u & 1, py = cls >> 1, px = cls & 1;
        // taps of the 3x3 window that fall on low row i - 1 + py + dy: py = 0: dy 0 <- ky 0; dy 1 <- ky 1, 2.  py = 1: dy 0 <- ky 0, 1; dy 1 <- ky 2
        const int ky0 = py == 0 ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), ky1 = py == 0 ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
        const int kx0 = px == 0 ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), kx1 = px == 0 ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
        const int cog = cb * WSU_COB + co;
        float v[16], r[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.f;
        for (int ci = 0; ci < cup; ++ci)
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) {
                    const float w = w3[(((size_t)cog * ctot + ci) * 3 + ky) * 3 + kx];
                    const int sy = (py + ky + 1) & 1, sx = (px + kx + 1) & 1;       // sub-position of the upsampled pixel under this tap
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = fmaf(w, wt[(((size_t)(c * 16 + e) * cup + ci) * 2 + sy) * 2 + sx], v[e]);
                }
        if (wc_dense)
#pragma unroll
            for (int e = 0; e < 16; ++e) wc_dense[((((size_t)cog * cl + c * 16 + e) * 2 + py) * 2 + px) * 4 + dy * 2 + dx] = v[e];
        uint32_t h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 x = (_Float16)v[2 * e], y = (_Float16)v[2 * e + 1];
            h[e] = (uint32_t)__builtin_bit_cast(unsigned short, x) | ((uint32_t)__builtin_bit_cast(unsigned short, y) << 16);
            r[2 * e] = (v[2 * e] - (float)x) * 2048.f; r[2 * e + 1] = (v[2 * e + 1] - (float)y) * 2048.f;
        }
        const u32x4 h0 = mk_u4(h[0], h[1], h[2], h[3]), h1 = mk_u4(h[4], h[5], h[6], h[7]);
        const int E = wsu_q4_block_exp(wsu_f16x16_max_abs_bits(h0, h1));
        const float sc = wsu_pow2f(E);
        uint32_t q[4] = {0, 0, 0, 0};
        WSU_STATIC_FOR(8, e, { q[e >> 2] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q[e >> 2], r[2 * e], r[2 * e + 1], sc, e & 3); });
        q[2] = wsu_f16x8_to_fp4(h0, sc); q[3] = wsu_f16x8_to_fp4(h1, sc);
        char* slice = dst + (((size_t)cb * nchL + c) * 2 + dy) * W_L;
        char* base = slice + (size_t)(u * 3) * (WSU_COB * 16) + co * 16;
        *reinterpret_cast<u32x4*>(base) = h0;
        *reinterpret_cast<u32x4*>(base + WSU_COB * 16) = h1;
        *reinterpret_cast<u32x4*>(base + 2 * WSU_COB * 16) = mk_u4(q[0], q[1], q[2], q[3]);
        slice[W_GRAN_L + u * 64 + co] = (char)(E + 127 - 11);
        if (u == 0 && co < 32) *reinterpret_cast<u32x4*>(slice + W_GRAN_L + 512 + co * 16) = mk_u4(0, 0, 0, 0);      // the 512 pad bytes
    }
}

// combined bias: b3[co] + sum_{ci < cup} bT[ci] * sum_{taps} w3[co][ci][tap] (reflect padding: every output pixel has all nine taps)
__global__ void up_bias_kernel(const float* __restrict__ w3, const float* __restrict__ bt, const float* __restrict__ b3, float* __restrict__ out, int cup, int c2, int cout) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= cout) return;
    float s = b3 ? b3[co] : 0.f;
    if (bt)
        for (int ci = 0; ci < cup; ++ci) {
            float t = 0.f;
            for (int k = 0; k < 9; ++k) t += w3[((size_t)co * (cup + c2) + ci) * 9 + k];
            s = fmaf(bt[ci], t, s);
        }
    out[co] = s;
}

}  // namespace

extern "C" {

// Bytes of the packed low half: per (64-co block, 16-channel chunk of x_low, dy) one 25 KB slice [class 4][dx 2][plane 3][64 co][16 B] + [8][64] scale bytes.
size_t wsu_conv3x3_up_packed_bytes(int cl, int cout) {
    if (cl <= 0 || cout <= 0 || cl % 16 || cout % WSU_COB) return 0;
    return (size_t)(cout / WSU_COB) * (cl / 16) * 2 * W_L;
}

// Packs the upsampled half of a decoder block's first conv.  w3: (cout, cup + c2, 3, 3) OIHW fp32, the conv's weights (input channels
// [0, cup) = the transposed conv's output, as torch.cat([xu, skip]) orders them, unet.py:172,178,184); wt: (cl, cup, 2, 2) fp32, the
// nn.ConvTranspose2d weights; bt (cup) / b3 (cout): their biases (optional).  Out: w_low_packed (wsu_conv3x3_up_packed_bytes), bias_out (cout
// floats: the combined bias) and, optional, wc_dense (cout * cl * 16 floats [cout][cl][py][px][dy][dx]: the combined weights in fp32).
// The skip half is wsu_conv3x3_pack_f4 of w3[:, cup:].
int wsu_conv3x3_up_pack(const float* w3_oihw, const float* wt, const float* bt, const float* b3, void* w_low_packed, float* bias_out, float* wc_dense,
                        int cl, int cup, int c2, int cout, void* stream) {
    WSU_REQUIRE(w3_oihw && wt && w_low_packed && bias_out, "conv3x3_up_pack: null pointer");
    WSU_REQUIRE(cl > 0 && cl % 16 == 0 && cup > 0 && c2 >= 0 && cout > 0 && cout % WSU_COB == 0,
                "conv3x3_up_pack: cl=%d must be a multiple of 16, cout=%d of %d, cup=%d > 0", cl, cout, WSU_COB, cup);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pack_up_low_kernel, dim3(1024), dim3(128), 0, s, w3_oihw, wt, (char*)w_low_packed, wc_dense, cup, c2, cl, cout);
    int rc = wsu_check_launch("pack_up_low_kernel");
    if (rc != WSU_OK) return rc;
    hipLaunchKernelGGL(up_bias_kernel, dim3((cout + 63) / 64), dim3(64), 0, s, w3_oihw, bt, b3, bias_out, cup, c2, cout);
    return wsu_check_launch("up_bias_kernel");
}

// Forward of relu(conv3x3_reflect(cat[convT2x2_s2(x_low), x_skip])) in one launch (K1u above).  x_low: planar Q tensor, cl channels at (h/2) x
// (w/2); x_skip: planar Q tensor, c2 channels at h x w; w_skip_packed: wsu_conv3x3_pack_f4(w3[:, cup:], c2, cout); w_low_packed / bias: from
// wsu_conv3x3_up_pack; y: planar Q tensor, cout channels at h x w.  h, w even; cl, c2 multiples of 16 (> 0), cout of 64 (<= 512).
// range_flag as in wsu_conv3x3_q_fwd.  Asynchronous on `stream`; allocates nothing.
int wsu_conv3x3_up_q_fwd(const void* x_low, const void* x_skip, const void* w_skip_packed, const void* w_low_packed, const float* bias, void* y,
                         int n, int h, int w, int cl, int c2, int cout, int relu, unsigned* range_flag, void* stream) {
    WSU_REQUIRE(x_low && x_skip && w_skip_packed && w_low_packed && y, "conv3x3_up_q: null pointer");
    WSU_REQUIRE(n > 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "conv3x3_up_q: bad shape n=%d h=%d w=%d (the output of a stride-2 transposed conv is even)", n, h, w);
    WSU_REQUIRE(cl > 0 && cl % 16 == 0 && c2 > 0 && c2 % 16 == 0, "conv3x3_up_q: cl=%d c2=%d must be positive multiples of 16", cl, c2);
    WSU_REQUIRE(cout > 0 && cout % WSU_COB == 0 && cout <= MAX_COUT, "conv3x3_up_q: cout=%d must be a multiple of %d (<= %d)", cout, WSU_COB, MAX_COUT);
    WSU_REQUIRE((long long)h * w * 50 < 0xFFFFFFF0LL, "conv3x3_up_q: h*w too large (a chunk must stay below 4 GiB)");
    UArgs a;
    a.xl = (const char*)x_low; a.xs = (const char*)x_skip; a.wps = (const char*)w_skip_packed; a.wpl = (const char*)w_low_packed; a.bias = bias;
    a.y = (char*)y;
    a.n = n; a.h = h; a.w = w; a.hl = h / 2; a.wl = w / 2; a.cl = cl; a.c2 = c2; a.cout = cout;
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH; a.ltiles_x = (a.wl + 31) / 32; a.ncb = cout / WSU_COB;
    a.nchS = c2 / 16; a.nchL = cl / 16; a.relu = relu; a.range_flag = range_flag;
    const long long nt = (long long)n * a.tiles_x * a.tiles_y * a.ncb;
    WSU_REQUIRE(nt > 0 && nt < 0x3FFFFFFFLL, "conv3x3_up_q: %lld tiles out of range", nt);
    WSU_REQUIRE((long long)a.ncb * a.nchL * 2 * W_L < 0x7FFFFFF0LL && (long long)a.ncb * a.nchS * W_S < 0x7FFFFFF0LL, "conv3x3_up_q: packed weights beyond 2 GiB");
    a.ntiles = (int)nt;
    static int ncu = 0, ablate = 0;
    if (ncu == 0) {
        const char* ev = getenv("WSU_QU_ABLATE"); ablate = ev ? atoi(ev) : 0;
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            wsu_set_error("conv3x3_up_q: cannot query the device"); return WSU_ERR_HIP;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_qu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) { wsu_set_error("hipFuncSetAttribute(conv3x3_qu): %s", hipGetErrorString(e)); return WSU_ERR_HIP; }
        ncu = prop.multiProcessorCount;
    }
    a.ablate = ablate;
    hipLaunchKernelGGL(conv3x3_qu_kernel, dim3(a.ntiles < ncu ? a.ntiles : ncu), dim3(NT), LDS_TOTAL, static_cast<hipStream_t>(stream), a);
    return wsu_check_launch("conv3x3_qu_kernel");
}

}  // extern "C"
