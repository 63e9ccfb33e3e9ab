/* wsu.h -- C ABI of the MI355X-native UNet pixel-predictor hot path (libwsu.so).
 *
 * The reference (uibk-uncover/ws-unet) is pure Python on PyTorch and has NO FFI /
 * plugin interface; the compute this library replaces is what `UNet.forward`
 * (src/unet/model/unet.py:137-189) dispatches to ATen.  Each entry point below
 * names the reference call site it stands in for.  The Python host side
 * (ws_unet_amd/model/unet.py) binds these with ctypes and keeps the reference's
 * own module / evaluate API on top.
 *
 * Conventions
 *  - plain C types only; every buffer pointer is a DEVICE pointer owned by the
 *    caller (PyTorch-ROCm caching allocator in our host code); the library
 *    allocates nothing and keeps no global mutable state (re-entrant; one host
 *    thread per GPU may call concurrently);
 *  - `stream` is a hipStream_t passed as void*; all work is asynchronous on it;
 *  - returns 0 on success, a negative code on bad arguments / HIP error, message
 *    via wsu_last_error() (thread-local);  never throws;
 *  - activations are NHWC (N*H*W*C contiguous).  With C == 1 this coincides with
 *    the reference's NCHW, so model input (N,1,H,W) and output (N,1,H,W) need no
 *    re-layout;
 *  - `mode` selects storage + arithmetic:
 *      WSU_MODE_F32    = 0  fp32 storage, exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *      WSU_MODE_BF16X3 = 1  fp32 storage, split-bf16 (hi*hi + hi*lo + lo*hi) on
 *                           v_mfma_f32_32x32x16_bf16, fp32 accumulate (~2^-17 rel. error)
 *      WSU_MODE_BF16   = 2  bf16 storage, bf16 MFMA, fp32 accumulate
 *      WSU_MODE_BF16X3S = 3 the arithmetic of BF16X3 on activations stored ALREADY SPLIT by their producer: per pixel and 16-channel
 *                           chunk  [bf16 hi of ch 0-7][hi 8-15][lo 0-7][lo 8-15]  (4 x 16 B = the fp32 chunk size, so tensors keep
 *                           their fp32 allocation).  Staging becomes a plain copy; results are bitwise those of BF16X3.  Forward
 *                           entry points only (wsu_conv3x3_fwd / _head_fwd / _fused_first_fwd, wsu_convt2x2_fwd); weights are packed
 *                           as for BF16X3; no pool_idx, no zero padding.
 *      WSU_MODE_F16F8 = 4   two-level split on the f16 and block-scaled fp8 matrix pipes (forward inference format like BF16X3S):
 *                           w*x ~ f16(w)*f16(x) + e4m3(w)*e4m3(x - f16(x)) + e4m3(w - f16(w))*e4m3(x), fp32 accumulate.  The first
 *                           product is exact (v_mfma_f32_32x32x16_f16), the two cross terms share one
 *                           v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 rate; its two 32-element scale blocks carry 2^-12-sized
 *                           residuals at full e4m3 precision): ~2^-15 relative error per product at 2/3 of BF16X3's matrix cycles.
 *                           Activations take 3 bytes per element: per pixel and 16-channel chunk [f16 ch 0-7][f16 ch 8-15]
 *                           [e4m3((x - f16 x) * 2^12) ch 0-15] = 48 bytes, pixel stride C * 3 bytes (the e4m3 copy e4m3(x / 4) that
 *                           the second cross term needs is derived from the f16 part while staging); weights packed by the pack
 *                           entry points with this mode (4 bytes per weight).  Values beyond +-448 lose the residual term (plain f16 accuracy); beyond +-65504 the f16 part overflows
 *                           like any f16 pipeline (use BF16X3S for such networks).
 *      WSU_MODE_F16F8X = 5  the arithmetic of F16F8 on fp32 tensors (operands encoded while staging, fp32 results): the matrix kernels of
 *                           the training path -- wsu_conv3x3_fwd (with pool and pool_idx), wsu_convt2x2_fwd, wsu_conv3x3_bwd_data,
 *                           wsu_conv3x3_bwd_weight, wsu_convt2x2_bwd_weight; weights packed as for F16F8.  Gradient operands must be
 *                           brought into f16's range by the caller (a power-of-two scale, undone on the results).
 */
#ifndef WSU_H
#define WSU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSU_VERSION 100

enum { WSU_MODE_F32 = 0, WSU_MODE_BF16X3 = 1, WSU_MODE_BF16 = 2, WSU_MODE_BF16X3S = 3, WSU_MODE_F16F8 = 4, WSU_MODE_F16F8X = 5 };

/* `products` of the planar BACKWARD matrix kernels (wsu_conv3x3_pl_bwd_data / _bwd_weight, wsu_convt2x2_pl_bwd_data / _bwd_weight) -- which terms of
 * (f16 a + residual a)(f16 b + residual b) are multiplied:
 *      WSU_PRODUCTS_F16F8 = 0  f16 a * f16 b on the f16 pipe + both residual cross terms on the block-scaled fp8 pipe (the forward's arithmetic,
 *                              ~2^-16 per product; 19 matrix units per 9 of plain f16 in the data gradient, 4 per 2 in the weight gradient);
 *      WSU_PRODUCTS_F16   = 1  f16 a * f16 b only (2^-11 per operand, unbiased rounding, fp32 accumulation): the residual planes are neither
 *                              fetched nor multiplied.  A weight gradient sums 10^6..10^7 such products per element and a data gradient
 *                              9 * cout of them, so the rounding noise averages out to a relative L2 of <= 2e-4 per weight gradient and
 *                              ~1e-4 per data-gradient layer (tests/test_gpu_planar_train.py) -- below the 1e-3 that ReLU-mask flips from
 *                              the FORWARD's own rounding put between any two arithmetics on these networks, and 8x finer than bf16
 *                              autocast.  The forward pass and the activations' format do not change; gradient tensors then hold f16
 *                              values (their residual plane is not used, see the K7p notes below). */
enum { WSU_PRODUCTS_F16F8 = 0, WSU_PRODUCTS_F16 = 1 };

enum {
    WSU_OK = 0,
    WSU_ERR_ARG = -1,      /* bad argument (shape, mode, null pointer) */
    WSU_ERR_HIP = -2,      /* a HIP call / launch failed */
    WSU_ERR_UNSUPPORTED = -3
};

int wsu_version(void);
const char* wsu_last_error(void);

/* Bytes per activation element for a mode (4, 4, 2, 4, 3, 4). */
int wsu_act_elem_size(int mode);

/* ---- weight packing (done once per weight update; replaces nothing in the reference:
 *      it re-lays nn.Conv2d's OIHW fp32 weight, unet.py:82-132, for the MFMA A operand) */
size_t wsu_conv3x3_packed_bytes(int cin, int cout, int mode);
int wsu_conv3x3_pack(const float* w_oihw, void* w_packed, int cin, int cout, int mode, void* stream);
/* transposed: packs w^T flipped (for the data-gradient pass), from the same OIHW source */
int wsu_conv3x3_pack_dgrad(const float* w_oihw, void* w_packed, int cin, int cout, int mode, void* stream);
size_t wsu_convt2x2_packed_bytes(int cin, int cout, int mode);
/* w is nn.ConvTranspose2d's (Cin, Cout, 2, 2) fp32 weight, unet.py:125,130 */
int wsu_convt2x2_pack(const float* w_iohw, void* w_packed, int cin, int cout, int mode, void* stream);

/* ---- K1 (+K2, K4): y = [relu](conv3x3_reflect(cat[x1, x2]) + bias), optional fused 2x2 max-pool.
 *      Replaces nn.Conv2d(k3, pad 1, reflect) + F.relu (unet.py:141-186), torch.cat (unet.py:178,184)
 *      and nn.MaxPool2d (unet.py:144,149).
 *      x1: (N,H,W,C1), x2: (N,H,W,C2) or NULL (C2 == 0); channels of x1 come first (upsampled, then skip).
 *      C1, C2 multiples of 16 (fp32 modes) / 32 (bf16); Cout multiple of 64; H, W >= 2.
 *      y: (N,H,W,Cout).  y_pool: (N,H/2,W/2,Cout) or NULL.  pool_idx: uint8 (N,H/2,W/2,Cout) argmax in
 *      row-major window order with first-max-wins ties, or NULL.  pad_zero != 0 selects zero instead of
 *      reflect padding (used by the data-gradient pass). */
int wsu_conv3x3_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias,
                    void* y, void* y_pool, uint8_t* pool_idx,
                    int n, int h, int w, int c1, int c2, int cout,
                    int mode, int relu, int pad_zero, void* stream);

/* ---- K1 + K5 fused (last layer): out[n,co,y,x] = sigmoid(head_b[co] + sum_c head_w[co,c] * relu(conv3x3(cat[x1,x2]) + bias)[n,y,x,c]).
 *      Replaces d42 + outconv + F.sigmoid (unet.py:186,189) in one launch; cout must be 64, head_cout 1..4.
 *      out / logit (optional): NCHW fp32.  y (optional): also store the 64-channel conv output (NHWC). */
int wsu_conv3x3_head_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias, void* y,
                         const float* head_w, const float* head_b, float* out, float* logit,
                         int n, int h, int w, int c1, int c2, int cout, int head_cout, int mode, void* stream);

/* ---- K1f: first layer fused into the conv behind it (e11 -> e12 [-> pool], unet.py:141-144) for single-plane inputs: the 64
 *      channels of e11 are computed from the image while e12 stages its input tile and never reach HBM.  Bitwise the same result
 *      as wsu_conv3x3_first_fwd followed by wsu_conv3x3_fwd.  img: (N,1,H,W) fp32; w1: (64,1,3,3) OIHW; b1: (64) or NULL;
 *      w_packed / bias: the second conv (cin = 64) as for wsu_conv3x3_fwd; y: (N,H,W,cout); y_pool / pool_idx optional. */
int wsu_conv3x3_fused_first_fwd(const float* img, const float* w1, const float* b1, const void* w_packed, const float* bias,
                                void* y, void* y_pool, uint8_t* pool_idx, int n, int h, int w, int cout, int mode, int relu, void* stream);

/* ---- K1w: the same forward conv (mode bf16x3 only: fp32 NHWC activations) through a Winograd F(2,3) kernel along x:
 *      1.5x fewer MFMAs, results equal to the direct kernel up to fp32 re-association.  Weights from wsu_conv3x3_wino_pack
 *      ([cob][chunk][12 taps][4 planes][64 co][16 B], row taps pre-multiplied by G).  Same fusions: x2 = second concat source,
 *      y_pool / pool_idx = fused 2x2 max-pool (+argmax), head_* = fused 1x1 head + sigmoid (y may then be NULL). */
size_t wsu_conv3x3_wino_packed_bytes(int cin, int cout);
int wsu_conv3x3_wino_pack(const float* w_oihw, void* w_packed, int cin, int cout, void* stream);
int wsu_conv3x3_wino_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias, void* y,
                         void* y_pool, uint8_t* pool_idx,
                         const float* head_w, const float* head_b, float* head_out, float* head_logit, int head_cout,
                         int n, int h, int w, int c1, int c2, int cout, int relu, void* stream);

/* ---- K1p: the same forward conv in the f16f8 arithmetic on PLANAR activations ("F16F8P" storage, forward inference only): a tensor of C
 *      channels is [n][C/16 chunks][3 planes][H][W][16 B] with, per pixel and 16-channel chunk, plane 0 = f16 ch 0-7, plane 1 = f16 ch
 *      8-15, plane 2 = e4m3((x - f16 x) * 2^12) ch 0-15 (3 bytes per element).  The planes are three of the four LDS planes of the matrix
 *      kernel (the fourth, e4m3(x / 4), is derived from the f16 planes by the loader wave that fetched them), so staging is a pure LDS-DMA
 *      (global_load_lds) and a persistent workgroup per CU pipelines it across chunks and tiles (csrc/conv3x3_pl.hip).  Weights from wsu_conv3x3_pack(mode F16F8).  Outputs, each optional: y (planar),
 *      y_pool (2x2 max-pooled, planar), head (1x1 conv + sigmoid on cout == 64 channels; out / logit NCHW fp32).  range_flag (optional
 *      device word, all three planar entry points): bit 0 is OR-ed in when a stored activation exceeds +-448, where the e4m3 residual
 *      saturates and that value keeps only f16 accuracy (NaN / Inf set it too) -- the caller's signal to switch to BF16X3S.
 *      x_residual = 0 (plain variant only): the inputs' residual plane is neither loaded nor multiplied -- w*x ~ f16(w)*f16(x) + e4m3(w - f16 w)*e4m3(x),
 *      15 instead of 19 matrix units per chunk; the activation rounding of THIS layer's inputs then costs ~2.5e-5 MAE on the network output
 *      (profiles/r02/conv3x3_units_probe.md), so a caller spends it on at most a few layers.  Replaces the same
 *      reference lines as wsu_conv3x3_fwd / wsu_conv3x3_head_fwd (unet.py:141-189). */
/*      x_residual = 2 (round 3): both cross terms in ONE block-scaled fp4 (e2m1) MFMA per tap pair -- w*x ~ f16(w)*f16(x) + 2^(Ew+Ex-11) [fp4(rw 2^11/2^Ew)
 *      fp4(f16 x/2^Ex) + fp4(f16 w/2^Ew) fp4(rx 2^11/2^Ex)], rw / rx the residuals, 2^Ew / 2^Ex E8M0 block scales per (output channel, tap, 16 input
 *      channels) / per (pixel, 16 channels) = the smallest power of two that does not saturate fp4 (the block's largest f16 part maps into [2, 3) or [4, 6]): 14 instead of 19 matrix units per chunk.  The
 *      activations' stored format does not change (the loader waves derive the fp4 granule and scale byte of a pixel from its three stored granules);
 *      the weights come from wsu_conv3x3_pack_f4 (wsu_conv3x3_packed_f4_bytes).  MAE of the unet_2 output ~2.1e-5 instead of 4e-6. */
size_t wsu_conv3x3_packed_f4_bytes(int cin, int cout);
int wsu_conv3x3_pack_f4(const float* w_oihw, void* w_packed, int cin, int cout, void* stream);
int wsu_conv3x3_pl_fwd(const void* x1, const void* x2, const void* w_packed, const float* bias, void* y, void* y_pool,
                       const float* head_w, const float* head_b, float* head_out, float* head_logit, int head_cout,
                       int n, int h, int w, int c1, int c2, int cout, int relu, int x_residual, unsigned* range_flag,
                       unsigned char* relu_mask_out, void* stream);
/*      relu_mask planes (round 3; training): relu_mask_out (optional; plain variant: y only) receives the 1-bit ReLU mask of y --
 *      [n][cout/8][hp][wp] bytes, hp = 16 * ceil(h / 16), wp = 32 * ceil(w / 32) (wsu_relu_mask_bytes), byte (pixel, 8-channel granule),
 *      bit e = (stored f16 value of channel 8 g + e > 0).  The data gradient of the NEXT conv reads it (mask1_bits / mask2_bits of
 *      wsu_conv3x3_pl_bwd_data) instead of the 2 bytes per element of y's f16 planes.  Rows beyond h / columns beyond w are not written. */
size_t wsu_relu_mask_bytes(int n, int c, int h, int w);

/* ---- K1q (round 4): the forward conv of the DEFAULT inference mode 'f16f4p' on planar Q tensors ("F16F4P" storage; csrc/conv3x3_q.hip).
 *      The arithmetic is x_residual = 2's: w*x ~ f16(w)*f16(x) + both residual cross terms as ONE block-scaled fp4 (e2m1) MFMA per tap pair.  What changed is
 *      WHO makes the fp4 operands: a planar Q tensor stores, per image and 16-channel chunk,
 *          plane 0 = f16 ch 0-7, plane 1 = f16 ch 8-15, plane 2 = Q   as [H][W][16 B] each, then the scale plane S,
 *      Q = per pixel 32 fp4 nibbles: nibble i = fp4(f16 part of channel i / 2^E), nibble 16 + i = fp4((x - f16 x) * 2^11 / 2^E) (low nibble first, round to
 *      nearest even, saturating), 2^E = the smallest power of two with (largest |f16 part| of the 16 channels) / 2^E <= 6; S = one byte E + 127 per pixel in
 *      16 x 32-pixel tile blocks [ceil(H/16)][ceil(W/32)][16][32] (a conv tile's scale bytes are one 512-byte run; rows / columns beyond the image are
 *      padding).  A chunk is 48 H W + 512 ceil(H/16) ceil(W/32) bytes (3.06 B per element; wsu_planar_q_bytes).  The PRODUCING epilogue writes all of it
 *      from its fp32 values (residual nibbles from the exact residual; round 3's loader waves re-rounded a stored e4m3 residual), so the consumer's loader
 *      waves only issue LDS-DMA.  y_format (this entry point, wsu_convt2x2_pl_fwd, wsu_conv3x3_first_pl_fwd): WSU_PLANAR_Q, or WSU_PLANAR_A = the
 *      e4m3-residual F16F8P format above (what the transposed conv and every training kernel read).  Weights: wsu_conv3x3_pack_f4.
 *      Replaces the same reference lines as wsu_conv3x3_pl_fwd (unet.py:141-189). */
#define WSU_PLANAR_A 0
#define WSU_PLANAR_Q 1
size_t wsu_planar_q_bytes(int n, int c, int h, int w);
int wsu_conv3x3_q_fwd(const void* x1, const void* x2, const void* w_packed_f4, const float* bias, void* y, void* y_pool,
                      const float* head_w, const float* head_b, float* head_out, float* head_logit, int head_cout,
                      int n, int h, int w, int c1, int c2, int cout, int relu, int y_format, unsigned* range_flag, void* stream);

/* ---- K0p + K1q fused (round 4): e11 -> e12 -> pool of the DEFAULT mode in one launch for single-plane inputs (unet.py:141-144; kernel variant F1 of
 *      csrc/conv3x3_q.hip).  The loader waves of the persistent kernel compute e11's 64 channels from the image straight into the LDS input slots -- fp32
 *      fused multiply-adds in the tap order of wsu_conv3x3_first_pl_fwd, the same planar-Q encoding -- instead of fetching them: bitwise the result of
 *      wsu_conv3x3_first_pl_fwd(y_format Q) followed by wsu_conv3x3_q_fwd, and xe11 never reaches HBM.  img (N,1,H,W) fp32; w1_taps (9, 64) fp32 = e11's
 *      weights TAP-MAJOR (w1.reshape(64, 9).T -- a tap's 16 channels are one scalar 16-dword load of the loader waves), b1 (64); both 64-byte aligned;
 *      w_packed_f4 / bias: the second conv (cin = 64; wsu_conv3x3_pack_f4); y, y_pool: planar Q tensors (both required).  h, w even; cout % 64 == 0. */
int wsu_conv3x3_q_fused_first_fwd(const float* img, const float* w1_taps, const float* b1, const void* w_packed_f4, const float* bias, void* y, void* y_pool,
                                  int n, int h, int w, int cout, int relu, unsigned* range_flag, void* stream);

/* ---- K1u (round 4): a decoder block's transposed conv + concat + first 3x3 conv in ONE launch, default inference mode (csrc/conv3x3_qu.hip):
 *          y = relu(conv3x3_reflect(cat[conv_transpose2x2_s2(x_low), x_skip]))        (unet.py:171-173, 177-179, 183-185)
 *      There is no non-linearity between the two convs, so the upsampled half is a 2 x 2-tap conv on x_low with weights combined per PARITY CLASS of
 *      the output pixel (row parity py, column parity px): Wc[py,px][dy,dx][co][c] = sum_ci sum_{(ky,kx) -> (dy,dx)} w3[co][ci][ky][kx] wT[c][ci][sy][sx],
 *      reading x_low rows i - 1 + py + dy, columns j - 1 + px + dx of output pixel (2 i + py, 2 j + px), CLAMPED at the border (= the reflect
 *      padding of the upsampled tensor); the transposed conv's bias folds into one combined bias per output channel.  512 instead of 576
 *      multiply-adds per output and channel pair, no transposed-conv launch, and the upsampled tensor never exists.  Arithmetic and operand
 *      formats as K1q (f16 products + block-scaled fp4 cross terms, planar Q tensors in and out).
 *      wsu_conv3x3_up_pack: w3 (cout, cup + c2, 3, 3) OIHW fp32 (input channels [0, cup) = the upsampled tensor, as torch.cat([xu, skip]) orders
 *      them), wt (cl, cup, 2, 2) fp32 (nn.ConvTranspose2d), bt (cup) / b3 (cout) biases or NULL -> w_low_packed (wsu_conv3x3_up_packed_bytes: per
 *      (64-co block, 16-channel chunk of x_low, dy) a 25 KB slice [class 4][dx 2][plane 3][64 co][16 B] + [8][64] scale bytes), bias_out (cout
 *      floats), and optionally wc_dense (cout * cl * 16 floats, [cout][cl][py][px][dy][dx]).  The skip half is wsu_conv3x3_pack_f4 of w3[:, cup:].
 *      wsu_conv3x3_up_q_fwd: x_low planar Q (cl channels at h/2 x w/2), x_skip planar Q (c2 channels at h x w) -> y planar Q (cout at h x w).
 *      h, w even; cl, c2 positive multiples of 16; cout a multiple of 64, <= 512.  Asynchronous; allocates nothing. */
size_t wsu_conv3x3_up_packed_bytes(int cl, int cout);
int wsu_conv3x3_up_pack(const float* w3_oihw, const float* wt, const float* bt, const float* b3, void* w_low_packed, float* bias_out, float* wc_dense,
                        int cl, int cup, int c2, int cout, void* stream);
int wsu_conv3x3_up_q_fwd(const void* x_low, const void* x_skip, const void* w_skip_packed, const void* w_low_packed, const float* bias, void* y,
                         int n, int h, int w, int cl, int c2, int cout, int relu, unsigned* range_flag, void* stream);

/* ---- K1p + K0p fused: e11 -> e12 (-> pool) of the planar path in one launch for single-plane inputs (unet.py:141-144).  The loader waves of
 *      the persistent kernel compute e11's 64 channels from the image straight into the LDS stages (instead of fetching them); bitwise the
 *      result of wsu_conv3x3_first_pl_fwd followed by wsu_conv3x3_pl_fwd, and xe11 never reaches HBM.  img (N,1,H,W) fp32, w1 (64,1,3,3),
 *      b1 (64) or NULL; w_packed / bias: the second conv (cin = 64); y / y_pool (each optional, not both NULL) planar. */
int wsu_conv3x3_pl_fused_first_fwd(const float* img, const float* w1, const float* b1, const void* w_packed, const float* bias,
                                   void* y, void* y_pool, int n, int h, int w, int cout, int relu, unsigned* range_flag, void* stream);

/* ---- K7p: data gradient of the 3x3 reflect conv on PLANAR tensors (csrc/conv3x3_pl.hip kernel variant GRAD + csrc/train_pl.hip; autograd of
 *      unet.py:141-189).  Gradients use the F16F8P layout with the gradient's residual scaling: planes f16 | f16 | e4m3((g - f16 g) * 2^14),
 *      values pre-scaled by a power of two (wsu_pow2_grad_scale).  g: pre-activation gradient (cout channels); w_packed_dgrad from
 *      wsu_conv3x3_pack_dgrad(mode F16F8); dx1 = input channels [0, csplit), dx2 (NULL iff csplit == cin) the rest (fused concat); mask1 /
 *      mask2 (optional): planar ACTIVATIONS shaped like dx1 / dx2 -- the input of this conv as the producing layer stored it -- whose sign is
 *      that layer's ReLU mask (relu'(0) = 0).  pad_zero = 0: reflect padding; the border ring of the adjoint is one more launch of the same
 *      kernel over strips of g's border rows / columns with the four 1x3 weight sets of wsu_conv3x3_pack_ring (4 x
 *      wsu_conv3x3_packed_bytes(cin, cout, F16F8) bytes) in `workspace` (>= wsu_conv3x3_pl_bwd_data_workspace_bytes).  pad_zero = 1: the
 *      adjoint of the zero-padded conv (ring arguments unused).  cin, csplit multiples of 64, cout of 16. */
size_t wsu_conv3x3_pl_bwd_data_workspace_bytes(int n, int h, int w, int cin, int cout);
int wsu_conv3x3_pack_ring(const float* w_oihw, void* w_packed, int cin, int cout, void* stream);
int wsu_conv3x3_pl_bwd_data(const void* g, const void* w_packed_dgrad, const void* w_packed_ring, void* workspace, size_t workspace_bytes,
                            void* dx1, void* dx2, int csplit, const void* mask1, const void* mask2,
                            const unsigned char* mask1_bits, const unsigned char* mask2_bits,
                            int n, int h, int w, int cin, int cout, int pad_zero, int products, void* stream);
/*      mask1_bits / mask2_bits (optional, each only together with its mask1 / mask2): the relu_mask planes of those activations; the
 *      persistent kernel then brings a tile's mask in by LDS-DMA (4 KB) instead of re-reading the activations' f16 planes (64 KB); the
 *      border fold still reads mask1 / mask2.  products: WSU_PRODUCTS_F16F8 or WSU_PRODUCTS_F16 (above). */

/* ---- K7p: weight / bias gradients on PLANAR operands (csrc/wgrad.hip wgrad_pl_kernel: the split-K MFMA GEMM over pixels of
 *      wsu_conv3x3_bwd_weight with planar staging -- one stored f16 granule + half a residual granule per (pixel, 8 channels), no split arithmetic).
 *      conv: g (cout channels, planar gradient), x1 / x2 (the layer's saved planar inputs, c1 / c2 channels), dw (cout, c1 + c2, 3, 3), db (cout) or
 *      NULL (sum of the decoded gradient values, fixed order).  Transposed conv: x (cin channels at h x w), dy (cout channels, planar gradient at
 *      2h x 2w), dw (cin, cout, 2, 2), db (cout) or NULL.  All channel counts multiples of 64; workspace >= wsu_wgrad_workspace_bytes.  Deterministic.
 *      products: WSU_PRODUCTS_F16F8 or WSU_PRODUCTS_F16 (above; db is then the sum of the gradient's f16 parts). */
int wsu_conv3x3_pl_bwd_weight(const void* g, const void* x1, const void* x2, float* dw, float* db, float* workspace, size_t workspace_bytes,
                              int n, int h, int w, int c1, int c2, int cout, int products, void* stream);
int wsu_convt2x2_pl_bwd_weight(const void* x, const void* dy, float* dw, float* db, float* workspace, size_t workspace_bytes,
                               int n, int h, int w, int cin, int cout, int products, void* stream);

/* ---- K7p: the other backward kernels of the planar training path (csrc/planar.hip, csrc/train_pl.hip; autograd of unet.py:137-189).
 *      wsu_convt2x2_pl_bwd_data: dx[n,i,j,ci] = sum dy[n,2i+a,2j+b,co] w[ci,co,a,b] times the ReLU mask of the layer below (mask: the planar
 *        ACTIVATION the transposed conv consumed, optional); dy (cout channels at 2h x 2w) and dx (cin at h x w) planar gradients; weights from
 *        wsu_convt2x2_pl_pack_dgrad (cin * cout * 16 bytes); cin a multiple of 64, cout of 16; products as above (WSU_PRODUCTS_F16: only the
 *        f16 planes of dy and of the weights travel, six LDS stages instead of three).
 *      wsu_maxpool2x2_pl_bwd: g = (skip_g + routing of dy_pool onto the first maximum of each 2x2 window) * (act > 0); act = the stored activation
 *        the pool consumed (the argmax is recomputed from it); skip_g optional; g may alias skip_g.
 *      wsu_conv1x1_sigmoid_pl_bwd: head backward (unet.py:186-188): x planar activation (c in {16..128}), w (cout <= 4, c), out / dout (N, cout, H, W)
 *        fp32 -> g (planar gradient w.r.t. the pre-activation of the layer that produced x), dw (cout, c), db (cout).
 *      wsu_colsum_pl: per-channel sums of a planar gradient (bias gradient of the transposed conv).
 *      wsu_conv3x3_first_pl_bwd_weight: first layer, single input plane: dw (c, 1, 3, 3), db (c) from the planar gradient g and img (N, 1, H, W).
 *      All reductions are two-stage with a fixed order (deterministic).
 *      products (every K7p entry point): with WSU_PRODUCTS_F16 a GRADIENT tensor carries no residual plane -- its producers (the data
 *        gradients, the pool and head backward) write the two f16 planes only and its consumers read only those (plane 2 of such a tensor
 *        is never touched: 2 instead of 3 bytes of traffic per gradient element); a gradient produced with one setting must be consumed
 *        with the same.  Activations are always read whole (the pool's argmax compares the stored values). */
int wsu_convt2x2_pl_pack_dgrad(const float* w_iohw, void* w_packed, int cin, int cout, void* stream);
int wsu_convt2x2_pl_bwd_data(const void* dy, const void* w_packed_dgrad, void* dx, const void* mask,
                             int n, int h, int w, int cin, int cout, int products, void* stream);
int wsu_maxpool2x2_pl_bwd(const void* skip_g, const void* dy_pool, const void* act, void* g, int n, int h, int w, int c, int products, void* stream);
size_t wsu_head_pl_bwd_workspace_bytes(int c, int cout);
int wsu_conv1x1_sigmoid_pl_bwd(const void* x, const float* w, const float* out, const float* dout, void* g, float* dw, float* db,
                               float* workspace, size_t workspace_bytes, int n, int h, int wd, int c, int cout, int products, void* stream);
size_t wsu_chansum_pl_workspace_bytes(int c);
int wsu_colsum_pl(const void* g, float* db, float* workspace, size_t workspace_bytes, int n, int h, int w, int c, int products, void* stream);
int wsu_conv3x3_first_pl_bwd_weight(const void* g, const float* img, float* dw, float* db, float* workspace, size_t workspace_bytes,
                                    int n, int h, int w, int c, int products, void* stream);
/*      wsu_conv3x3_first_pl_bwd_data (round 4): the INPUT gradient of the planar training path (saliency, src/saliency.py:159-174): g (planar gradient,
 *        c channels), w_oihw (c, cin <= 8, 3, 3) -> dx (N, cin, H, W) fp32 in g's power-of-two scale (the reflect adjoint included). */
int wsu_conv3x3_first_pl_bwd_data(const void* g, const float* w_oihw, float* dx_nchw, int n, int h, int w, int cin, int c, int products, void* stream);

/* ---- K3p / K0p: the other two kernels of the planar (F16F8P) inference path (csrc/planar.hip).
 *      wsu_convt2x2_pl_fwd: nn.ConvTranspose2d(k2, s2) + bias (unet.py:125,130,177,183), x: cin channels at (h, w) planar -> y: cout channels at
 *      (2h, 2w) planar; weights from wsu_convt2x2_pack(mode F16F8); cin a multiple of 32, cout of 64.
 *      wsu_conv3x3_first_pl_fwd: the first layer e11 (unet.py:82,141), x_nchw (N, cin <= 8, H, W) fp32 -> y: cout (multiple of 16) channels planar.
 *      y_format (round 4): WSU_PLANAR_A (the format above) or WSU_PLANAR_Q (K1q; the inputs of the transposed conv stay WSU_PLANAR_A). */
int wsu_convt2x2_pl_fwd(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin, int cout,
                        int y_format, unsigned* range_flag, void* stream);
int wsu_conv3x3_first_pl_fwd(const float* x_nchw, const float* w_oihw, const float* bias, void* y, int n, int h, int w, int cin, int cout,
                             int relu, int y_format, unsigned* range_flag, unsigned char* relu_mask_out, void* stream);

/* ---- first layer: conv3x3 reflect on a few input planes given as NCHW fp32 (the model input).
 *      Replaces e11 (unet.py:82,141).  cin <= 8, cout multiple of 8.  w is plain OIHW fp32. */
int wsu_conv3x3_first_fwd(const float* x_nchw, const float* w_oihw, const float* bias, void* y,
                          int n, int h, int w, int cin, int cout, int mode, int relu, void* stream);

/* ---- K2 standalone 2x2/2 max-pool (used when not fused). */
int wsu_maxpool2x2_fwd(const void* x, void* y, uint8_t* pool_idx, int n, int h, int w, int c, int mode, void* stream);

/* ---- K3: y[n,2i+a,2j+b,co] = bias[co] + sum_ci x[n,i,j,ci] * w[ci,co,a,b]   (unet.py:177,183) */
int wsu_convt2x2_fwd(const void* x, const void* w_packed, const float* bias, void* y,
                     int n, int h, int w, int cin, int cout, int mode, void* stream);

/* ---- K5: out[n,co,y,x] = sigmoid(b[co] + sum_c x[n,y,x,c] * w[co,c])   (unet.py:189).
 *      w: (cout, c) fp32 (outconv.weight squeezed), out: NCHW fp32; logit: optional NCHW fp32 pre-sigmoid. */
int wsu_conv1x1_sigmoid_fwd(const void* x, const float* w, const float* bias, float* out, float* logit,
                            int n, int h, int w_, int c, int cout, int mode, void* stream);

/* ---- K6: UniformDropout (unet.py:32-42) on plane `channel` of an NCHW fp32 tensor, out of place:
 *      y = x*mask + KB(x)*(1-mask).  mask: (N,1,H,W) fp32 keep-mask, or NULL to draw it from a
 *      counter-based hash of (seed, element index) with keep probability keep_prob. */
int wsu_uniform_dropout_fwd(const float* x, float* y, const float* mask, float* mask_out,
                            int n, int c, int h, int w, int channel,
                            float keep_prob, uint64_t seed, void* stream);

/* ---- K10: per-image WS residual statistics (src/unet/evaluate.py:125-132).
 *      x_u8: (N,H,W) cover/stego pixels; y01: (N,H,W) fp32 network output in [0,1].
 *      Over the interior [1:-1,1:-1]:  xhat = y*255;  beta_hat = mean((x - (x^1)) * (x - xhat));
 *      l1 = mean|x - xhat|.  Deterministic (fixed-order fp64 tree). */
int wsu_ws_residual_stats(const uint8_t* x_u8, const float* y01, float* beta_hat, float* l1,
                          int n, int h, int w, void* stream);

/* ---- K11: weighted-stego payload estimate of the predictor's caller, src/ws/estimate.py:55-136 `attack`, batched.
 *      x_u8: (N,H,W) DEVICE pixels of the analysed plane.  The pixel prediction is either
 *        x_hat (DEVICE fp32): hat_full=1 -> (N,H,W), interior read at [r][c] (a network output; hat_scale = 255),
 *                             hat_full=0 -> (N,H-2,W-2) (what `pixel_estimator(x)` returns; hat_scale = 1), or
 *        pixel_filter (HOST, 9 floats K[a][b] of the reference's (3,3,1) kernel array, filters/evaluate.py:30-50,136-141):
 *                             x_hat = convolve(x/255, K, 'valid')*255 evaluated inside the kernel.
 *      mean_filter (HOST, 9 floats, same layout; NAMED_FILTERS['AVG'] by default, estimate.py:60,93-95) feeds the local
 *      variance for weighted = 1 (1/(5+var)) or -1 (5+var); weighted = 0 -> uniform weights (:97-110).
 *      correct_bias (:126-128) needs x_bias = pixel_estimator(x_bar - x) in x_hat's layout (not with pixel_filter).
 *      beta_hat: (N) fp32, clipped at 0 (:121); sums: optional (N,3) fp64 {sum w, sum w*s*(x-x_hat), sum w*s*x_bias}. */
size_t wsu_ws_attack_workspace_bytes(int n);
int wsu_ws_attack(const uint8_t* x_u8, const float* x_hat, const float* x_bias, const float* pixel_filter, const float* mean_filter,
                  int hat_full, float hat_scale, int weighted, int correct_bias, float* beta_hat, double* sums,
                  void* workspace, size_t workspace_bytes, int n, int h, int w, void* stream);

/* Linear pixel predictor on its own (filters/evaluate.py:136-141): y (N,H-2,W-2) = convolve(x/255., K, 'valid')*255.
 * x: DEVICE (N,H,W) fp32; filter: HOST 9 floats K[a][b]. */
int wsu_filter3x3_valid_f32(const float* x, const float* filter, float* y, int n, int h, int w, void* stream);

/* (x ^ 1 - x) / 255. as fp32: the network input of the bias term `pixel_estimator(x_bar - x)` (estimate.py:127, evaluate.py:45) */
int wsu_lsb_delta_unit_f32(const uint8_t* x, float* y, size_t count, void* stream);

/* Epoch meter of the training loop (src/_defs/metrics.py:122-142 WSMeter.update): per-image fp64 beta_hat over the interior from the
 * FLOAT inputs x01 and outputs y01, both (N,H,W) single-plane fp32 in [0,1]; the caller clips and averages |beta - alpha/2|. */
int wsu_ws_meter_beta(const float* x01, const float* y01, double* beta_hat, int n, int h, int w, void* stream);

/* ---- u8 -> [0,1] fp32, numpy float32 division semantics of evaluate.py:45 (x / 255.) */
int wsu_u8_to_unit_f32(const uint8_t* x, float* y, size_t count, void* stream);

/* ======================= backward / train step (K7, K8, K9) =======================
 * The reference publishes no UNet training script; these entry points are the autograd of
 * UNet.forward (unet.py:137-189) + the loss classes (src/_defs/losses.py:28-121) + torch.optim.AdamW as used
 * by the detector loop (src/detector/train.py:55-95,228).  Gradients are exact-fp32 / split-bf16 on fp32 storage;
 * every reduction is two-stage in a fixed order (bitwise reproducible, no float atomics).
 * "g" always denotes a PRE-activation gradient (dLoss/d(conv output before ReLU)). */

/* dx of conv3x3-reflect: zero-padded transposed conv on the matrix cores + reflect-adjoint border fold.
 * g: (N,H,W,Cout) fp32.  dx1: (N,H,W,csplit), dx2: (N,H,W,cin-csplit) or NULL (the two inputs of a fused concat).
 * relu_mask1/2 (optional, shapes of dx1/dx2): saved post-ReLU activations; where they are <= 0 the gradient is zeroed,
 * which makes dx the pre-activation gradient of the producing layers.  mode: WSU_MODE_F32 or WSU_MODE_BF16X3. */
size_t wsu_conv3x3_bwd_data_workspace_bytes(int n, int h, int w, int cin, int cout, int mode);
int wsu_conv3x3_bwd_data(const void* g, const void* w_packed_dgrad, const float* w_oihw, void* workspace, size_t workspace_bytes,
                         void* dx1, void* dx2, int csplit, const void* relu_mask1, const void* relu_mask2,
                         int n, int h, int w, int cin, int cout, int mode, void* stream);

/* dW (OIHW, Cout x (c1+c2) x 3 x 3) and db (Cout, optional) of conv3x3-reflect on cat[x1, x2]: exact fp32 MFMA (mode F32) or
 * split-bf16 MFMA fed by transposing LDS reads (mode BF16X3); db is always an exact fp32 sum. */
size_t wsu_wgrad_workspace_bytes(int cm, int cn, int ntaps);
int wsu_conv3x3_bwd_weight(const float* g, const float* x1, const float* x2, float* dw, float* db,
                           float* workspace, size_t workspace_bytes,
                           int n, int h, int w, int c1, int c2, int cout, int mode /* F32 exact | BF16X3 */, void* stream);

/* first layer: dW (Cout x cin x 3 x 3), db from g (N,H,W,Cout) and the NCHW fp32 input planes. */
size_t wsu_first_bwd_workspace_bytes(int n, int h, int w, int cin, int cout);
int wsu_conv3x3_first_bwd_weight(const float* g, const float* x_nchw, float* dw, float* db,
                                 float* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout, void* stream);

/* first layer: dx (N,cin,H,W) NCHW fp32 -- the input saliency gradient (src/saliency.py:159-174). */
int wsu_conv3x3_first_bwd_data(const float* g, const float* w_oihw, float* dx_nchw, int n, int h, int w, int cin, int cout, void* stream);

/* transposed conv: dW (Cin x Cout x 2 x 2), db (Cout, optional) from x (N,h,w,Cin) and dy (N,2h,2w,Cout) ... */
int wsu_convt2x2_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                            float* workspace, size_t workspace_bytes,
                            int n, int h, int w, int cin, int cout, int mode, void* stream);
/* ... and dx (N,h,w,Cin), optionally masked by the saved post-ReLU activation of the producing layer. */
size_t wsu_convt2x2_packed_dgrad_bytes(int cin, int cout, int mode);
int wsu_convt2x2_pack_dgrad(const float* w_iohw, void* w_packed, int cin, int cout, int mode, void* stream);
int wsu_convt2x2_bwd_data(const void* dy, const void* w_packed_dgrad, void* dx, const void* relu_mask,
                          int n, int h, int w, int cin, int cout, int mode, void* stream);

/* 2x2 max-pool backward into the full-resolution gradient buffer (accumulate != 0: add to what the skip path
 * already stored there).  pool_idx from the forward; xp_relu_mask: pooled activation (gradient only where > 0). */
int wsu_maxpool2x2_bwd(float* g_full, const float* dy_pool, const uint8_t* pool_idx, const float* xp_relu_mask,
                       int n, int h, int w, int c, int accumulate, void* stream);

/* head: gx (N,H,W,C) = relu_mask(x) * W^T (dout * out * (1-out)); dw (cout x C), db (cout); cout <= 4. */
size_t wsu_head_bwd_workspace_bytes(int c, int cout);
int wsu_conv1x1_sigmoid_bwd(const float* x, const float* w, const float* out, const float* dout,
                            float* gx, float* dw, float* db, float* workspace, size_t workspace_bytes,
                            int n, int h, int w_, int c, int cout, int apply_relu_mask, void* stream);

/* K8: L1 + WS loss and dLoss/dout in one call (losses.py:33-36,47-89,99-116).
 * use_l1: 0 = off, 1 = mean |covers - out| (L1Loss), 2 = mean (covers - out)^2 (L2Loss, losses.py:39-42). */
size_t wsu_l1ws_loss_workspace_bytes(int n);
int wsu_l1ws_loss_fwd_bwd(const float* out, const float* covers, const float* inputs, const float* alphas,
                          float* loss, float* loss_parts, float* dout, float* beta_hat, void* workspace, size_t workspace_bytes,
                          int n, long long per_image, int use_l1, int use_ws, void* stream);

/* K9: multi-tensor AdamW (decoupled weight decay), one launch for all parameters.
 * table: DEVICE array of ntensors records {float* p; const float* g; float* m; float* v; int64 n; int64 first_block}.
 * skip_flag: optional device int; non-zero = the launch changes nothing. */
int wsu_adamw_multi_tensor(const void* table, int ntensors, long long total_blocks,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                           const int* skip_flag, void* stream);

/* Helpers of the train step that keep its scalar arithmetic on the device and inside libwsu (reference: there is none -- PyTorch's
 * GradScaler plays this role for fp16 training; pattern src/detector/train.py:88-95):
 *   wsu_pow2_grad_scale     scale2 = {2^floor(2 - log2 max|x|), its reciprocal}: the power-of-two scale of the f16f8x backward chain
 *   wsu_scale_f32           y = x * factor[0]
 *   wsu_scale_multi_tensor  in-place, many tensors, one launch; table records {float* p; int64 n; int64 first_block}
 *   wsu_nonfinite_flag      flag[0] = bucket holds inf / NaN, flag[1] += flag[0]; feed flag to wsu_adamw_multi_tensor(skip_flag) */
int wsu_pow2_grad_scale(const float* x, long long n, float* scale2, void* workspace, void* stream);
int wsu_scale_f32(const float* x, float* y, long long n, const float* factor, void* stream);
int wsu_scale_multi_tensor(const void* table, int ntensors, long long total_blocks, const float* factor, void* stream);
int wsu_nonfinite_flag(const float* g, long long n, int* flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WSU_H */
