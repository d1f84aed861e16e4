/*
 * diinn_hip.h -- C ABI of the MI355X (gfx950) DIINN implicit-decoder library
 * (libdiinn_hip.so).
 *
 * This is the drop-in boundary for ONE path of the reference
 * (robotic-vision-lab/Dual-Interactive-Implicit-Neural-Network):
 *     ImplicitDecoder.forward(x, size, bsize)   mode=3, init_q=False
 *     src/models/components/diinn.py:163-173   (helpers :94-110, :132-139, :149-160)
 * The reference is pure Python/ATen and has no FFI of its own; the entry points
 * below are what a ctypes binding inside that forward() would call
 * (INTEGRATION.md shows the stub).  Plain pointers and sizes only: no torch
 * types, no exceptions, no hidden allocation, no global mutable state.
 *
 * Conventions
 *   - every function returns a DIINN_* status (0 = ok), except the size queries;
 *   - "host" pointers are ordinary CPU memory, "dev" pointers are HIP device
 *     memory on the device that is current for the calling thread;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *     Launch functions only enqueue work: they never synchronise, allocate or
 *     copy, so they can be captured into a hipGraph;
 *   - device buffers are borrowed for the lifetime of the enqueued work only;
 *   - all arithmetic is fp32 ("f32"); indices are int32.
 */
#ifndef DIINN_HIP_H
#define DIINN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIINN_ABI_VERSION 9
/* What each ABI number added: INTEGRATION.md, "ABI history".  v9 (this header): ONE trunk entry point (diinn_rdn_forward_ex; the four
 * one-algorithm entry points are wrappers kept for this number only), the F(4x4) split hand-off fails loudly (sticky status word,
 * diinn_conv_wino4_ws_status), the training step's own kernels for the hoisted conv's weight gradient and its Winograd forward
 * (diinn_backward_cell_sum_ex, diinn_unfold_tiled, diinn_sum_parts, diinn_precompute_P_wpu), the small-map kernels (diinn_conv_t16, diinn_conv1x1_t16). */

/* status codes */
#define DIINN_OK                 0
#define DIINN_ERR_INVALID_ARG    1   /* null pointer, non-positive size, bad range   */
#define DIINN_ERR_UNSUPPORTED    2   /* decoder variant the HIP path does not cover  */
#define DIINN_ERR_HIP            3   /* a HIP runtime call failed (see diinn_last_hip_error) */
#define DIINN_ERR_TOO_LARGE      4   /* an extent overflows the kernels' 32-bit indexing */

/* decoder geometry fixed by the reference's defaults (diinn.py:40): in_channels=64,
 * hidden_dims=[256]*4, 3x3 unfold -> 576, RGB head. */
#define DIINN_IN_CHANNELS 64
#define DIINN_HIDDEN      256
#define DIINN_LAYERS      4
#define DIINN_OUT         3
#define DIINN_P_CHANNELS  (DIINN_LAYERS * DIINN_HIDDEN)   /* 1024 floats per LR cell */

/* sine evaluation used by the synthesis branch (reference: torch.sin, diinn.py:25-26).  The inference kernels keep
 * the synthesis branch in REVOLUTIONS (weights, biases and the Q0 table of sections 10-12 are pre-multiplied by
 * fp32(1/(2 pi))), so the accumulator already holds the argument v_sin_f32 takes; the training forward
 * (diinn_decode_train_fwd) saves sine arguments in radians and uses the radian forms. */
#define DIINN_SIN_ACCURATE 0   /* Cody-Waite reduction + polynomial, <= ~3 ulp (inference: after multiplying back
                                  to radians) */
#define DIINN_SIN_HW       1   /* inference: the bare v_sin_f32, 1 VALU op, valid for |arg| <= 512 pi; training
                                  forward: v_sin_f32(fract(x / 2 pi)), abs error ~ |x| * 6e-8 */
#define DIINN_SIN_HW_REDUCED 2 /* the default.  inference: v_sin_f32(r - rint(r)) on r in revolutions, an exact
                                  reduction valid for any magnitude, 3 VALU ops, measured max error 3e-8 on the fixtures;
                                  training forward: two-term Cody-Waite reduction in revolutions + v_sin_f32, 5 ops,
                                  <= 4e-7 for |x| <= 1e4 */

/* arithmetic of the per-pixel layers 1..3 (everything else is always fp32) */
#define DIINN_COMPUTE_F32  0   /* v_mfma_f32_32x32x2_f32: the reference's precision, parity <= 1e-4           */
#define DIINN_COMPUTE_BF16 1   /* v_mfma_f32_32x32x16_bf16: bf16 weights/activations, fp32 accumulate;
                                  parity restated to 2e-3 * max|ref| (BASELINE config 5)                     */

#define DIINN_COMPUTE_BF16_FULL 3  /* DIINN_COMPUTE_BF16 plus the hoisted 3x3 conv (P) on bf16 operands too (features and
                                   weights rounded to bf16, fp32 accumulate): ~1.6e-3 relative at default-init
                                   weights; the fastest path, BASELINE config 5 */
#define DIINN_COMPUTE_BF16X3 4     /* split bf16 (r03, optional): the per-pixel layers 1..3 on the bf16 MFMA with every operand
                                  carried as hi + lo bf16 parts -- hi*hi + hi*lo + lo*hi, fp32 accumulation (packed
                                  section 14); on maps of >= 32,768 cells the hoisted 3x3 conv P as well (section 15;
                                  smaller maps: the fp32 Winograd form); seeds, biases, sine, layer 0 and the head stay fp32.  As
                                  accurate as DIINN_COMPUTE_F32 on every fixture at default-init weights (5e-8) and
                                  inside the 1e-4 x max(1,|ref|) tolerance on the x3 stress set (2e-4 of |out| 6.8) */
#define DIINN_COMPUTE_F32_QONLY 2 /* decoder modes 1 and 2 (diinn.py:116-131): fp32, synthesis GEMM only; the
                                  workspace slots 1..3 hold the per-cell modulation k_i (>= 0) produced by
                                  diinn_cell_chain from P (diinn_decode_ex runs P, the chain and the decode) */

int         diinn_abi_version(void);
const char* diinn_status_string(int status);
/* hipError_t of the most recent failing HIP call made by this thread (0 if none). */
int         diinn_last_hip_error(void);

/* ---- weights: reference state_dict -> packed device image ------------------
 * Replaces: the 18 parameter tensors ImplicitDecoder.__init__ registers
 * (diinn.py:73-80,92; names/shapes in SURVEY.md App. A.1).
 * Inputs are HOST pointers to contiguous fp32 tensors in the reference's own
 * layouts:
 *   K0w [256,576]        K0b [256]          (K.0.0.weight / .bias)
 *   Kw[i] [256,832]      Kb[i] [256]   i=0..2 -> K.1..K.3 (first 256 input
 *                                        channels multiply q, last 576 the unfolded features, diinn.py:136)
 *   Q0w [256,3]          Q0b [256]          (Q.0.0: inputs rel_h, rel_w, ratio)
 *   Qw[i] [256,256]      Qb[i] [256]   i=0..2 -> Q.1..Q.3
 *   Lw [3,256]           Lb [3]             (last_layer)
 * Output: `packed` HOST buffer of diinn_packed_weight_floats() floats, to be
 * copied verbatim to the device by the caller.  Pure host function. */
size_t diinn_packed_weight_floats(void);
/* Sections of the packed image (offset and size in floats): 0 WL stacked per-pixel layers, 1 WP the
 * hoisted 3x3 conv, 2 bK, 3 Q0 (+bQ0), 4 bQ1..3, 5 L, 6 bL, 7 WLB (bf16 copy of WL), 8 WLT (WL transposed,
 * read by the backward pass), 9 WPB (bf16 copy of WP), 10 BQR (bQ1..3 / (2 pi): the bf16 kernels evaluate the
 * sine on revolutions, and the synthesis rows inside section 7 are pre-multiplied by 1/(2 pi) to match), 11 Q0R
 * (section 3 / (2 pi)), 12 WLR (section 0 with its synthesis pieces / (2 pi): the fp32 inference kernels evaluate the
 * sine on revolutions as well), 13 WPU (section 1 in Winograd F(2x2,3x3) form, U = G Wx G^T: what the fp32 inference
 * entry points -- everything but diinn_precompute_P -- read, at every map size), 14 WLX (the per-pixel
 * layers as hi + lo bf16 parts, hi = bf16(w), lo = bf16(w - hi), for DIINN_COMPUTE_BF16X3), 15 WPX (the hoisted 3x3 conv
 * in the same form), 16 WL16 (the per-pixel layers as A operands of v_mfma_f32_16x16x4_f32 for the 16-pixel latency kernel,
 * synthesis rows in revolutions).  Every section but 7 and
 * 9..16 is a pure permutation (plus zero padding) of the reference tensors, so a training loop can re-pack on the
 * device with one gather; sections 7 and 9..16 hold derived values, read by the inference kernels only (the
 * training forward and LIIF read sections 0, 1 and 4; diinn_precompute_P reads section 1 whatever the map size).
 *
 * VALIDITY WORD.  The pad word behind bL (float index 3 of section 6) holds the bit pattern DIINN_PACKED_MAGIC in an
 * image written by diinn_pack_weights, i.e. one whose derived sections are filled.  An image assembled some other
 * way (the device gather of a training step leaves the word and the derived sections zero) is only good for the entry
 * points that read the permutation sections (diinn_precompute_P, diinn_decode_train_fwd, diinn_backward_data,
 * diinn_liif_decode).  The launch functions cannot look into device memory without synchronising, so the check is
 * made by the kernels: every inference entry point that reads a derived section (diinn_precompute_P_ex / _win,
 * diinn_decode*, in every compute mode) writes NaN into ALL of its outputs when the word is missing -- a loud
 * wrong answer instead of a silent one (tests/test_gpu_parity.py::test_image_without_derived_sections_is_refused). */
#define DIINN_PACKED_MAGIC 0x44493038u   /* "DI08": changes whenever the packed layout does (sections, sizes) */
/* A TRAINING image (the permutation sections gathered on the device, derived sections absent) may carry this word instead once
 * its section 13 (WPU, the hoisted conv in Winograd form) has been filled on the device as well: diinn_precompute_P_wpu accepts
 * it; every other entry point that reads a derived section still answers NaN. */
#define DIINN_PACKED_MAGIC_WPU 0x44495750u   /* "DIWP" */
int    diinn_packed_section(int section, size_t* offset_floats, size_t* size_floats);
int    diinn_pack_weights(const float* K0w, const float* K0b,
                          const float* const Kw[3], const float* const Kb[3],
                          const float* Q0w, const float* Q0b,
                          const float* const Qw[3], const float* const Qb[3],
                          const float* Lw, const float* Lb,
                          float* packed);

/* ---- coordinates / nearest-exact indices (host, bit-exact) ------------------
 * Replaces: ImplicitDecoder._make_pos_encoding (diinn.py:94-110) and the index
 * map of F.interpolate(mode='nearest-exact') (diinn.py:106,168) for ONE axis.
 * idx[n_out] = source LR index of every HR sample, rel[n_out] = (up - in[idx]) * n_in
 * in the reference's fp32 operation order.  `small_output` != 0 selects ATen's
 * small-output CPU kernel rounding, which the reference hits when Hu + Wu <= 128
 * (see diinn_uses_small_output_kernel).  Either output pointer may be NULL. */
int diinn_make_axis_tables(int n_in, int n_out, int small_output, int32_t* idx, float* rel);
int diinn_uses_small_output_kernel(int Hu, int Wu);
/* Same tables evaluated by the device code the decode kernel uses (for tests).
 * idx_dev/rel_dev: DEVICE buffers of n_out elements. */
int diinn_make_axis_tables_device(void* stream, int n_in, int n_out, int small_output,
                                  int32_t* idx_dev, float* rel_dev);

/* The device sine of `sin_mode`, elementwise over n floats (tests: accuracy of the synthesis
 * branch's activation, reference torch.sin, diinn.py:25-26). */
int diinn_eval_sin_device(void* stream, int sin_mode, const float* x_dev, float* y_dev, int n);

/* ---- workspace ---------------------------------------------------------------
 * The per-cell modulation image P[B,H,W,1024] fp32 is the
 * only workspace.  rows [r0,r1) of the LR map are needed to decode HR rows
 * [y0,y1): diinn_lr_rows_for_band reports them (no halo: the 3x3 halo is read
 * from `feat` directly). */
size_t diinn_workspace_bytes(int B, int H, int W);
int    diinn_lr_rows_for_band(int H, int Hu, int Wu, int y0, int y1, int* r0, int* r1);

/* ---- kernels -------------------------------------------------------------------
 * diinn_precompute_P  (replaces: F.unfold(x,3,padding=1) + the feature half of
 *   K[0..3], diinn.py:168,133,136): P[b,y,x,i*256+ch] = Wx_i . unfold(feat)[b,:,y,x] + bK_i
 *   for LR rows [r0,r1).  feat_dev [B,64,H,W] contiguous NCHW; P_dev [B,H,W,1024].
 * diinn_decode_band   (replaces: nearest-exact replication + step() mode 3 +
 *   batched_step, diinn.py:168,132-139,149-160): writes out_dev[b, :, y0:y1, :]
 *   of a contiguous [B,3,Hu,Wu] tensor.  Needs P rows for the band.
 * diinn_decode        = precompute_P on the rows the band needs, then decode_band. */
int diinn_precompute_P(void* stream, const float* feat_dev, const float* packed_dev,
                       float* P_dev, int B, int H, int W, int r0, int r1);
/* the same with an explicit arithmetic: DIINN_COMPUTE_BF16_FULL runs the conv on bf16 operands (section 9 of the
 * packed image); every other mode is the fp32 conv above. */
int diinn_precompute_P_ex(void* stream, const float* feat_dev, const float* packed_dev,
                          float* P_dev, int B, int H, int W, int r0, int r1, int compute);
/* The fp32 Winograd form of P for an image whose validity word is DIINN_PACKED_MAGIC or DIINN_PACKED_MAGIC_WPU (the training
 * forward: 0.21 against the direct kernel's 0.48 ms at B = 16, 48 x 48); any other word: NaN into every P value. */
int diinn_precompute_P_wpu(void* stream, const float* feat_dev, const float* packed_dev,
                           float* P_dev, int B, int H, int W, int r0, int r1);
int diinn_decode_band(void* stream, const float* P_dev, const float* packed_dev,
                      float* out_dev, int B, int H, int W, int Hu, int Wu,
                      int y0, int y1, int sin_mode);
int diinn_decode(void* stream, const float* feat_dev, const float* packed_dev,
                 float* workspace_dev, float* out_dev,
                 int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode);

/* Same with an explicit arithmetic for the per-pixel layers (`compute` = DIINN_COMPUTE_*). */
int diinn_decode_band_ex(void* stream, const float* P_dev, const float* packed_dev,
                         float* out_dev, int B, int H, int W, int Hu, int Wu,
                         int y0, int y1, int sin_mode, int compute);
int diinn_decode_ex(void* stream, const float* feat_dev, const float* packed_dev,
                    float* workspace_dev, float* out_dev,
                    int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode, int compute);

/* ---- row windows (multi-GPU row bands) --------------------
 * Under the row-band split a rank holds only the slice of each buffer its band touches.  A WINDOW is a
 * contiguous range of rows of a full-size tensor stored as a tensor of its own:
 *   features  feat_win_dev [B,64,feat_rows,W]   = LR rows [feat_row0, feat_row0+feat_rows) of [B,64,H,W]
 *   workspace P_win_dev    [B,p_rows,W,1024]    = LR rows [p_row0, p_row0+p_rows)        of [B,H,W,1024]
 *   output    out_win_dev  [B,3,out_rows,Wu]    = HR rows [out_row0, out_row0+out_rows)  of [B,3,Hu,Wu]
 * Row numbers (r0,r1,y0,y1) stay those of the full tensors, so results are bit-identical to the
 * full-buffer entry points.  Requirements (else DIINN_ERR_INVALID_ARG): the feature window holds rows
 * [max(r0-1,0), min(r1+1,H)) (the band's cells plus the 3x3 halo; rows outside the map are the unfold's
 * zero padding, diinn.py:168), the P window holds [r0,r1), the output window holds [y0,y1).
 * diinn_window_rows reports all three for an HR band: what a rank must receive and allocate.
 * Replaces (per band) the same reference lines as diinn_precompute_P_ex / diinn_decode_band_ex /
 * diinn_decode_ex; the reference itself has no multi-GPU inference (benchmarks.py:13 devices=1). */
int diinn_window_rows(int H, int Hu, int Wu, int y0, int y1,
                      int* feat_row0, int* feat_rows, int* p_row0, int* p_rows);
int diinn_precompute_P_win(void* stream, const float* feat_win_dev, int feat_row0, int feat_rows,
                           const float* packed_dev, float* P_win_dev, int p_row0, int p_rows,
                           int B, int H, int W, int r0, int r1, int compute);
int diinn_decode_band_win(void* stream, const float* P_win_dev, int p_row0, int p_rows, const float* packed_dev,
                          float* out_win_dev, int out_row0, int out_rows,
                          int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode, int compute);
int diinn_decode_win(void* stream, const float* feat_win_dev, int feat_row0, int feat_rows,
                     const float* packed_dev, float* P_win_dev, int p_row0, int p_rows,
                     float* out_win_dev, int out_row0, int out_rows,
                     int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode, int compute);

/* ---- tiles: column range and output strides (SURVEY.md section 8 row b2's sketch `diinn_decode_tile(..., y0,y1,x0,x1, out,
 * out_strides, ...)`; reference analogue: batched_step's column strips, diinn.py:149-160) -------------------------------
 * diinn_decode_tile_win decodes HR rows [y0,y1) x columns [x0,x1) of the [B,3,Hu,Wu] image from the P window (as
 * diinn_decode_band_win: P_win_dev holds LR rows [p_row0, p_row0+p_rows), full width) and writes them THROUGH STRIDES:
 * pixel (b, c, y, x) goes to
 *     out_tile_dev[b * out_batch_stride + c * out_plane_stride + (y - y0) * out_row_stride + (x - x0)]      (strides in floats)
 * i.e. out_tile_dev points at the tile's first pixel inside whatever the caller owns -- a crop buffer of its own, or a
 * view into a larger canvas (nothing outside the tile's pixels is written).  Requirements (else DIINN_ERR_INVALID_ARG):
 * out_row_stride >= x1 - x0, out_plane_stride >= (y1 - y0 - 1) * out_row_stride + (x1 - x0), out_batch_stride >=
 * 2 * out_plane_stride + that (rows, planes and batch items do not interleave), out_batch_stride <= 2^40.
 * Every compute mode.  A tile is BIT-IDENTICAL to the same pixels of a whole-image decode: blocks are anchored at (x0, y0),
 * no pixel's arithmetic depends on its place in a block, and kernel variants that differ in rounding are chosen from the
 * full image's geometry (H, W, Hu, Wu), never from the tile.  diinn_decode_band_ex / _win are this function with
 * x0 = 0, x1 = Wu and the strides of a contiguous [B,3,rows,Wu] window.  Enqueues one kernel; no allocation, no sync. */
int diinn_decode_tile_win(void* stream, const float* P_win_dev, int p_row0, int p_rows, const float* packed_dev,
                          float* out_tile_dev, long long out_row_stride, long long out_plane_stride,
                          long long out_batch_stride, int B, int H, int W, int Hu, int Wu,
                          int y0, int y1, int x0, int x1, int sin_mode, int compute);

/* Modes 1 and 2 only: the modulation chain per LR cell, k_0 = relu(P_0), k_i = relu(K_i^k k_{i-1} + P_i)
 * (diinn.py:118-121,126-129), for LR rows [r0,r1); k_i overwrites the P_i slot (i = 1..3) of P_dev. */
int diinn_cell_chain(void* stream, float* P_dev, const float* packed_dev, int B, int H, int W, int r0, int r1);

/* ---- training ------------------------------------
 * TILED PLANES.  Every per-pixel training buffer is a group of C channel rows over npix = B*Hu*Wu
 * pixels (pixel index (b*Hu + y)*Wu + x), stored as [ceil(npix/32) tiles][C rows][32 pixels] fp32:
 * element (c, pix) at ((pix >> 5) * C + c) * 32 + (pix & 31).  The padding of the last tile is never
 * read as data and never written.  diinn_training_plane_floats(npix, C) = ceil(npix/32) * C * 32
 * (or -1 for npix > DIINN_TRAIN_MAX_PIXELS). */
#define DIINN_TRAIN_MAX_PIXELS 1073741824LL
long long diinn_training_plane_floats(long long npix, int rows);

/* Training forward.  Replaces: ImplicitDecoder.step(), mode 3, run under autograd -- what forward()
 * does when bsize is None (diinn.py:170-171, called from SRLitModule.training_step,
 * sr_module.py:127-129).  Same network as diinn_decode_band (whole image, fp32); in addition
 * `acts_dev` = 4 tiled groups of 512 rows, one per layer i = 0..3, receives the rectified modulation
 * k_i = relu(.) in rows 0..255 and the sine argument s_i in rows 256..511 (q_i = k_i * sin(s_i)),
 * channels in the reference's order. */
int diinn_decode_train_fwd(void* stream, const float* P_dev, const float* packed_dev, float* out_dev,
                           float* acts_dev, int B, int H, int W, int Hu, int Wu, int sin_mode);

/* Backward pass of the per-pixel layers.  Replaces: autograd's backward through step()
 * (diinn.py:132-139), the part that runs per HR pixel.
 * Inputs: gout_planes_dev = d loss / d out as PLAIN planes [3][npix]; acts_dev from
 * diinn_decode_train_fwd; the packed image (section 8, the transposed layers, is read).
 * Outputs (tiled): G_dev, 4 groups of 512 rows: rows 0..255 of group i = g_a,i = d loss / d (modulation
 * pre-activation of layer i), rows 256..511 = g_s,i = d loss / d (sine argument of layer i);
 * Q_dev, 4 groups of 256 rows: q_i = k_i * sin(s_i).
 * From these every parameter gradient is a GEMM over the pixel axis (the two functions below):
 *   d[Wq_i ; Qw_i] = G_i q_{i-1}^T,  dL = g_out q_3^T,  (dbK_0 ; dQ0, dbQ0) = G_0 (syn, 1)^T,
 *   dP_i[cell] = sum of g_a,i (diinn_backward_cell_sum).
 * Enqueues 3 kernels on `stream` (the head's gates are computed inside layer 3's kernel since round 6); no allocation, no
 * synchronisation. */
int diinn_backward_data(void* stream, const float* gout_planes_dev, const float* acts_dev,
                        const float* packed_dev, float* G_dev, float* Q_dev, long long npix);

/* Weight-gradient GEMM over the pixel axis: C[M x Nc] = A . B^T where A = rows [a_row0, a_row0+M) of a tiled
 * group with a_rows rows per tile and B = rows [b_row0, b_row0+Nc) of a tiled group with b_rows rows;
 * M % 128 == 0, Nc % 256 == 0.  Split-K without atomics: part_dev is [ksplit][M][ldc] and slice ks
 * holds the product over its share of the plane tiles; the caller adds the slices.  ldc = Nc, or
 * Nc + 1 with `with_rowsum`, whose extra column receives the row sums of A (the bias gradients).
 * Replaces the weight-gradient GEMMs autograd runs for the 1x1 convolutions of step() (diinn.py:132-139). */
int diinn_plane_gemm_nt(void* stream, const float* A_dev, int a_rows, int a_row0, const float* B_dev, int b_rows,
                        int b_row0, float* part_dev, int M, int Nc, long long npix, int ksplit, int with_rowsum);

/* Skinny product over the pixel axis: C[M x 4] = A . S^T, A = rows [0, M) of a tiled group with a_rows rows,
 * S a tiled 4-row group (its last-tile padding must be zero).  part_dev [splits][M][4], added by the
 * caller.  Used for layer 0 (S = rel_h, rel_w, ratio, 1: diinn.py:165-167) and the head (S = g_out rows). */
int diinn_plane_rowdot(void* stream, const float* A_dev, int a_rows, const float* S_dev, float* part_dev,
                       int M, long long npix, int splits);
/* The add of the split partials the two functions above leave: part_dev [groups][nparts][n] -> out_dev [groups][n], slices added in
 * order (deterministic); n % 4 == 0, 16-byte aligned buffers. */
int diinn_sum_parts(void* stream, const float* part_dev, float* out_dev, int groups, int nparts, long long n);

/* dP[b][256 i + ch][cy][cx] = sum of g_a,i (G_dev, from diinn_backward_data) over the HR pixels whose
 * nearest LR cell is (cy, cx): the adjoint of the nearest-exact replication of diinn.py:168, i.e. the
 * gradient at the hoisted 3x3 convolution's output, NCHW [B][1024][H][W].  seg_h_dev [H+1] / seg_w_dev
 * [W+1] (int32, device) give the first HR row / column of every LR row / column (last entry Hu / Wu);
 * the index tables are monotone, so a cell's pixels form a rectangle.  Deterministic (no atomics). */
int diinn_backward_cell_sum(void* stream, const float* G_dev, const int32_t* seg_h_dev, const int32_t* seg_w_dev,
                            float* dP_dev, int B, int H, int W, int Hu, int Wu);
/* _ex: additionally (dP_tiled_dev non-NULL) the same sums as a tiled plane group over the CELL axis, [ceil(B H W / 32)][1024][32]:
 * the A operand of the hoisted conv's weight-gradient GEMM, dWx^T [576(+64) x 1024] = unfold . dP^T (diinn_plane_gemm_nt), whose
 * B operand diinn_unfold_tiled writes: the reference's F.unfold(feat, 3, padding=1) (diinn.py:168; row = c * 9 + ky * 3 + kx) as a
 * tiled group [ceil(B H W / 32)][rows][32], rows >= 576, the rows past 576 zero.  With these two the decoder's training step has
 * no library convolution left (sr_module.py:127-137). */
int diinn_backward_cell_sum_ex(void* stream, const float* G_dev, const int32_t* seg_h_dev, const int32_t* seg_w_dev,
                               float* dP_dev, float* dP_tiled_dev, int B, int H, int W, int Hu, int Wu);
int diinn_unfold_tiled(void* stream, const float* feat_dev, float* out_tiled_dev, int rows, int B, int H, int W);

/* ---- LIIF comparison decoder ------------------------
 * Replaces: LIIF.query_rgb + batched_predict + reshape_pred (liif.py:59-127,129-146), constructor
 * defaults (local_ensemble, feat_unfold, cell_decode): for every HR pixel the 580->256->256->256->256->3
 * ReLU MLP on the unfolded features of the 4 shifted nearest cells, blended by opposite areas.
 * The weights travel in the DIINN packed image (diinn_pack_weights) with this mapping:
 *   K0w/K0b <- imnet.layers.0 weight[:, :576] / bias;  Q0w [256,3] <- weight[:, 576:579]; Q0b <- weight[:, 579];
 *   Qw[i]/Qb[i] <- imnet.layers.{2,4,6} weight / bias;  Lw/Lb <- imnet.layers.8;  Kw[i], Kb[i] <- zeros.
 * workspace_dev: diinn_workspace_bytes(B,H,W) bytes.  Enqueues 2 kernels; no allocation, no sync.
 * diinn_liif_make_axis_tables: host restatement of the per-axis nearest index / relative coordinate for
 * ensemble shift v = -1 or +1 (grid_sample nearest, align_corners=False, as ATen's CPU kernel rounds). */
int diinn_liif_decode(void* stream, const float* feat_dev, const float* packed_dev, float* workspace_dev,
                      float* out_dev, int B, int H, int W, int Hu, int Wu);
int diinn_liif_make_axis_tables(int n_in, int n_out, int v, int32_t* idx, float* rel, float* rel_cell);

/* ---- MetaSR comparison decoder ----------------------
 * Replaces: MetaSR.query_rgb + batched_predict + reshape_pred (metasr.py:70-104,106-123): per HR pixel the
 * meta-network imnet = Linear(3,256)+ReLU+Linear(256,1728) predicts a [576 x 3] filter from
 * (rel_h, rel_w, r_rev), applied to the unfolded 3x3 features of the pixel's nearest (corner-anchored) cell.
 * Own packed image: diinn_metasr_pack_weights(W1 [256,3], b1 [256], W2 [1728,256], b2 [1728]) ->
 * diinn_metasr_packed_floats() floats (host).  workspace_dev: diinn_metasr_workspace_bytes(B,H,W) bytes
 * (the per-cell unfolded features).  Enqueues 2 kernels; no allocation, no synchronisation.
 * diinn_metasr_make_axis_tables: host restatement of the per-axis index / relative coordinate. */
size_t diinn_metasr_packed_floats(void);
int    diinn_metasr_pack_weights(const float* W1, const float* b1, const float* W2, const float* b2, float* packed);
size_t diinn_metasr_workspace_bytes(int B, int H, int W);
int    diinn_metasr_decode(void* stream, const float* feat_dev, const float* packed_dev, float* workspace_dev,
                           float* out_dev, int B, int H, int W, int Hu, int Wu);
int    diinn_metasr_make_axis_tables(int n_in, int n_out, int32_t* idx, float* rel, float* r_rev);

/* ---- RDN encoder trunk ------------------------------------
 * Replaces: RDN.forward after SFENet1 (rdn.py:95-105), config 'B' (16 RDBs x 8 dense 3x3 convs, growth 64,
 * 1x1 local/global fusion), inference.  Split-K convolution kernel (csrc/diinn_encoder.hip): faster than the
 * library convolution from 48x48 (3x: launch/latency-bound there) to 512x512 (1.09x) feature maps.
 * diinn_conv_ksplit: one 64-output convolution (taps = 9: 3x3 zero-padded, 1: 1x1) over Cin % 64 == 0 input
 *   channel planes at in_dev + b*in_batch_stride + c*H*W; epilogue = + bias, optional ReLU, optional
 *   + residual, written to one or two destinations (element strides in floats).  packed_w_dev holds the
 *   weight W[64][Cin][taps] as [half 2][wave 8][tap][group Cin/64][lane 64][4]:
 *   value = W[32 half + (lane&31)][wave*Cin/8 + 8 group + 2 e + (lane>>5)][tap]   (MFMA A-operand order per K-slice).
 * diinn_rdn_forward_ex: the whole trunk as 147 launches.  sfe1_dev [B,64,H,W] = SFENet1(x) (diinn_sfe1_forward, or computed by
 *   the caller); packed_dev = the 147 packed weights in execution order (SFENet2; per block: 8 dense convs, LFF; GFF.0, GFF.1;
 *   always read: the 1x1 layers and small maps); biases_dev = their 147 x 64 biases; out_dev [B,64,H,W].
 *   `algo` caps the kernel family of the 3x3 layers, the images that family reads must be given (else DIINN_ERR_INVALID_ARG),
 *   images of other families are ignored and may be NULL:
 *     DIINN_RDN_ALGO_DIRECT  direct sum everywhere (packed_dev only): the split-K kernel, and on small maps with fewer of its units than
 *                            compute units the strip kernel on the same image (diinn_conv_t16_applies);
 *     DIINN_RDN_ALGO_WINO    Winograd F(2x2,3x3) (packed_wino_dev) from B*H*W >= 8192 on, the split-K kernel below;
 *     DIINN_RDN_ALGO_WINO4   Winograd F(4x4,3x3) (packed_wino4_dev) where that kernel needs fewer rounds of workgroups than
 *                            F(2x2) (diinn_rdn_wino4_applies; from about 35,000 pixels on, depending on how its 2 * ceil(tiles
 *                            / 32) work items per image fill the last round), else as WINO; packed_wino_dev may be NULL for a map
 *                            with diinn_rdn_wino4_applies(B, H, W) == 1 (the F(2x2) image is not read then);
 *     DIINN_RDN_ALGO_AUTO    the fastest fp32 form the given images allow (= WINO4, WINO or DIRECT by what is non-NULL);
 *     DIINN_RDN_ALGO_X3      optional arithmetic: 3x3 and local-fusion layers in split bf16 (packed_x3_dev; packed_wino_dev too)
 *                            from B*H*W >= 32768 on, as WINO below.
 *   planes_dev: diinn_rdn_planes_floats(algo, B, H, W) floats, 16-byte aligned, any content (2,240 floats per pixel; X3: 2,816).
 *   w4ws_dev: the F(4x4) kernel's split area, diinn_conv_wino4_workspace_floats() floats (34.6 MB), or NULL (no layer is then
 *   split over its input channels: same results up to the reassociation documented at diinn_conv_wino4_ws, partly filled rounds
 *   cost whole ones).  Its first 4 KiB are control words the OWNER zeroes once at allocation; a forward re-zeroes only the
 *   arrival counters (DIINN_WINO4_COUNTER_BYTES), never the sticky status word (diinn_conv_wino4_ws): keep ONE such area per
 *   (device, stream) across forwards and a hand-off that ever gave up stays visible to diinn_conv_wino4_ws_status. */
#define DIINN_RDN_ALGO_AUTO   0
#define DIINN_RDN_ALGO_DIRECT 1
#define DIINN_RDN_ALGO_WINO   2
#define DIINN_RDN_ALGO_WINO4  3
#define DIINN_RDN_ALGO_X3     4
size_t diinn_rdn_planes_floats(int algo, int B, int H, int W);
int    diinn_rdn_forward_ex(void* stream, int algo, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                            const float* packed_wino4_dev, const float* packed_x3_dev, const float* biases_dev,
                            float* planes_dev, float* w4ws_dev, float* out_dev, int B, int H, int W);
int    diinn_conv_ksplit(void* stream, const float* in_dev, long long in_batch_stride, int Cin, int taps,
                        const float* packed_w_dev, const float* bias_dev,
                        const float* res_dev, long long res_batch_stride,
                        float* out0_dev, long long out0_batch_stride, float* out1_dev, long long out1_batch_stride,
                        int relu, int B, int H, int W);
size_t diinn_rdn_packed_floats(void);

/* Winograd F(2x2, 3x3) for the trunk's 3x3 convolutions (csrc/diinn_winograd.hip): 2.25x fewer MFMAs than the direct
 * sum, fp32, results equal up to reassociation (~1e-6 relative).
 * diinn_conv_wino: one 3x3 zero-padded 64-output convolution over Cin % 8 == 0 input planes (addressing, epilogue
 *   and error behaviour as diinn_conv_ksplit with one destination).  packed_u_dev holds U = G W G^T (G of F(2x2,3x3))
 *   as [row i 4][chunk Cin/8][col j 4][half 2][lane 64][4]:
 *   value = s_j U[32 half + (lane&31)][8 chunk + 2 e + (lane>>5)][i][j], s_2 = -1, else 1   (16 * 64 * Cin floats;
 *   bias_dev 16-byte aligned).
 * diinn_rdn_wino_packed_floats: floats of the 130 transformed 3x3 weights of the trunk, in execution order
 *   (diinn_rdn_forward_ex's packed_wino_dev). */
/* diinn_sfe1_forward: SFENet1 (rdn.py:96): 3x3 zero-padded convolution n_colors = Cin (1..4) -> 64 on the image itself,
 *   x_dev [B,Cin,H,W], w_dev [64,Cin,3,3] and bias_dev [64] in the reference's own layout, out_dev [B,64,H,W] = the
 *   sfe1_dev argument of diinn_rdn_forward_ex: with it the whole encoder runs through this library. */
int    diinn_sfe1_forward(void* stream, const float* x_dev, int Cin, const float* w_dev, const float* bias_dev,
                          float* out_dev, int B, int H, int W);
int    diinn_conv_wino(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                      const float* packed_u_dev, const float* bias_dev,
                      const float* res_dev, long long res_batch_stride,
                      float* out_dev, long long out_batch_stride, int relu, int B, int H, int W);
size_t diinn_rdn_wino_packed_floats(void);

/* Winograd F(4x4, 3x3) for the same layers on larger maps (csrc/diinn_winograd4.hip): 36 multiplies per (input, output)
 * pair and 4x4 output block, 1.78x fewer MFMAs than F(2x2, 3x3); fp32, transformed weights computed in float64 and
 * rounded once.  Accuracy: ~1e-5 of max|out| per layer on unit-variance inputs (worst measured 2e-5; F(2x2) and the direct
 * fp32 sum: 4e-7; tests/test_encoder_trunk.py bounds it at 4e-5) and 3e-6 of max|feat| through the whole trunk against
 * float64 (F(2x2): 3e-7; the direct fp32 sum: 4e-7; tools/enc_wino43_error.py).
 * diinn_conv_wino4: as diinn_conv_wino.  packed_u_dev holds U = G W G^T (6x6 per pair, G of F(4x4,3x3): Lavin & Gray,
 *   points 0, +-1, +-2, inf) as [wave 12][half 2][chunk Cin/8][q 3][lane 64][4]:
 *   value = U[32 half + (lane&31)][8 chunk + 2 e + (lane>>5)][i][j] with 6 i + j = 3 wave + q     (36 * 64 * Cin floats).
 * diinn_conv_wino4_ws: the same with a workspace (diinn_conv_wino4_workspace_floats() floats, 16-byte aligned).  Its first
 *   4 KiB are control words: the OWNER zeroes them once when it allocates the workspace; words [0, 512)
 *   (DIINN_WINO4_COUNTER_BYTES: arrival tickets and ready counts) are zero on entry and zero again when a launch has finished,
 *   and may be re-zeroed before a launch; word 1023 is the STICKY STATUS (below) and must not be re-zeroed by launch paths.  A launch of I work items (2 per 32
 *   Winograd tiles) on N compute units runs floor(I / N) * N of them whole and, where the cost model says it pays
 *   (DIINN_ENC_WINO4_SPLIT: 0 never, 2 always), cuts the input-channel chunks of the I mod N items left into N runs of equal time:
 *   a workgroup that computed part of an item's channels leaves its partial outputs in the workspace and the last one to
 *   arrive adds the parts in a fixed order.  Results are deterministic and depend on (shape, N) only; they differ from
 *   diinn_conv_wino4's by a reassociation in the transformed domain (up to ~2e-5 of max|out|: the size of either one's
 *   distance to the exact convolution).  The last arriver of an item waits -- bounded -- for the other parts' stores, which wait
 *   for nobody.  If such a wait ever gives up (a broken device or hand-off; never observed), the failure is LOUD: every output
 *   of that item is NaN, the status word is set, and every split item of every later launch on this workspace is NaN too until
 *   the host has cleared the status -- the library's rule (cf. the packed image's validity word): a NaN, never a plausible
 *   wrong picture.  Launches that share a workspace must be ordered (same stream).  diinn_conv_wino4 = diinn_conv_wino4_ws
 *   without a workspace (never splits).
 * diinn_conv_wino4_ws_status: *status = 1 if a wait has given up on this workspace since it was last cleared (else 0);
 *   clear != 0 zeroes the 4 KiB of control words behind the read.  NOT a launch function: it copies 4 bytes to the host and
 *   SYNCHRONISES `stream` (call it after a forward where a host check is wanted: tests, bench.py, RDN.handoff_status()).
 * diinn_conv_wino4_plan: what diinn_conv_wino4_ws would do on the current device: info[0] work items, [1] items run whole,
 *   [2] split workgroups, [3] units per split workgroup (chunks of 8 input channels + 4 overhead units per block entered).
 * diinn_rdn_wino4_packed_floats: floats of the 130 such weights of the trunk, in execution order (diinn_rdn_forward_ex's
 *   packed_wino4_dev).  diinn_rdn_wino4_applies: 1 if the trunk takes diinn_conv_wino4 for this map on the current device (one
 *   F(4x4) round = 1.40 F(2x2) rounds; DIINN_ENC_WINO4_MIN = n: from n pixels on), else the F(4x4) image is not read. */
int    diinn_conv_wino4(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                        const float* packed_u_dev, const float* bias_dev,
                        const float* res_dev, long long res_batch_stride,
                        float* out_dev, long long out_batch_stride, int relu, int B, int H, int W);
size_t diinn_conv_wino4_workspace_floats(void);
int    diinn_conv_wino4_ws(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                           const float* packed_u_dev, const float* bias_dev,
                           const float* res_dev, long long res_batch_stride,
                           float* out_dev, long long out_batch_stride, int relu, int B, int H, int W,
                           float* ws_dev, size_t ws_floats);
int    diinn_conv_wino4_plan(int Cin, int B, int H, int W, int with_workspace, int info[4]);
int    diinn_conv_wino4_ws_status(void* stream, float* ws_dev, int clear, int* status);
#define DIINN_WINO4_MAX_SPLIT_WGS 256   /* split workgroups of a launch (one per compute unit of an MI355X) */
#define DIINN_WINO4_COUNTER_BYTES 2048  /* the workspace's arrival counters (what a forward may re-zero); the status word is word 1023 */
size_t diinn_rdn_wino4_packed_floats(void);
int    diinn_rdn_wino4_applies(int B, int H, int W);

/* Split-bf16 arithmetic for the trunk's 3x3 convolutions (csrc/diinn_conv_x3.hip; optional, large maps): the direct sum
 * on v_mfma_f32_32x32x16_bf16 with every operand as hi + lo bf16 parts (hi = bf16(v), lo = bf16(v - hi)), a product as
 * w_lo.x_hi + w_hi.x_lo + w_hi.x_hi, fp32 accumulation.  Per layer ~4e-6 of max|out| against float64 (fp32 Winograd:
 * ~5e-7); through the whole trunk 7e-6 of max|feat| and 1e-7 in the decoded image.
 * diinn_conv3x3_x3: one 3x3 zero-padded 64-output convolution over Cin % 16 == 0 input planes (addressing and epilogue as
 *   diinn_conv_wino).  packed_x3_dev: [group Cin/16][tap 9][M-tile 2][hi, lo][lane 64][8 bf16] with
 *   value = part(W[32 mt + (lane&31)][16 group + 8 (lane>>5) + j][tap / 3][tap % 3])   (9 * 64 * Cin floats).
 * diinn_rdn_x3_packed_floats: floats of the 130 such weights of the trunk in execution order, followed by the 16 local-fusion
 *   1x1 weights (64 x 576 each) in the same format with one tap: [group 36][M-tile 2][hi, lo][lane 64][8 bf16].
 * DIINN_RDN_ALGO_X3 (diinn_rdn_forward_ex): inside the trunk these layers exchange their activations already split, through the
 *   planes' last buffer ([B][72 groups of 8 channels][hi, lo][H][W] x 16 B), from B*H*W >= 32768 pixels on (DIINN_ENC_X3_MIN). */
int    diinn_conv3x3_x3(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                        const float* packed_x3_dev, const float* bias_dev,
                        const float* res_dev, long long res_batch_stride,
                        float* out_dev, long long out_batch_stride, int relu, int B, int H, int W);
size_t diinn_rdn_x3_packed_floats(void);

/* Small maps (csrc/diinn_conv_t16.hip; the reference's own timing protocol, runtime_test.py:13: a 48 x 48 crop): the same
 * 3x3 layers as a direct fp32 sum (v_mfma_f32_16x16x4_f32) over strips 16 pixels wide and output quarters of 16 -- a
 * workgroup owns one quarter and 1 .. 3 consecutive rows of a strip and streams its quarter's weights once -- so that all
 * compute units work where the split-K kernel's (32 pixels x 32 outputs) units number fewer than the compute units.
 * diinn_conv_t16: addressing, epilogue and packed_w_dev as diinn_conv_ksplit with taps = 9 (it reads the SAME image) and one
 *   destination; W % 4 == 0, Cin % 64 == 0, and a map whose strips' rows can be dealt at most 3 to a workgroup
 *   (DIINN_ERR_UNSUPPORTED otherwise).  diinn_conv_t16_applies: 1 if the trunk gives this map's 3x3 layers to this kernel on
 *   the current device (a map below the Winograd kernels', with fewer split-K units than compute units; DIINN_ENC_NO_T16 = 1:
 *   never).  Results differ from diinn_conv_ksplit's by the order of the fp32 sum only. */
int    diinn_conv_t16(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                      const float* packed_w_dev, const float* bias_dev,
                      const float* res_dev, long long res_batch_stride,
                      float* out_dev, long long out_batch_stride, int relu, int B, int H, int W);
int    diinn_conv_t16_applies(int B, int H, int W);
/* diinn_conv_t16_plan: the partition a launch on this map would use on the current device: info[0] workgroups per output quarter
 *   (0: the kernel does not take this map), [1] workgroups per strip, [2] rows of each (the strip's last one: what is left),
 *   [3] strips = B * ceil(W / 16).  Workgroup s of a quarter owns strip s / info[1], rows (s % info[1]) * info[2] onwards. */
int    diinn_conv_t16_plan(int B, int H, int W, int info[4]);
/* diinn_conv1x1_t16: the local-fusion layers (1x1; rdn.py:34) on the same maps and units: addressing, epilogue (bias, ReLU,
 *   residual, two destinations) and packed_w_dev as diinn_conv_ksplit with taps = 1; W % 4 == 0, Cin % 64 == 0, Cin <= 640
 *   (every load of a workgroup is in flight at once: five groups of 16 channels per wave), else DIINN_ERR_UNSUPPORTED.  The
 *   trunk gives it the 16 fusion layers of a map with diinn_conv_t16_applies == 1 (the 1024-channel global fusion keeps
 *   diinn_conv_ksplit). */
int    diinn_conv1x1_t16(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                         const float* packed_w_dev, const float* bias_dev,
                         const float* res_dev, long long res_batch_stride,
                         float* out0_dev, long long out0_batch_stride, float* out1_dev, long long out1_batch_stride,
                         int relu, int B, int H, int W);

/* DEPRECATED, kept for this ABI number only: the one-algorithm trunk entry points of ABI <= 8 = diinn_rdn_forward_ex with algo
 * DIRECT / WINO / WINO4 / X3 and ONE workspace laid out [F(4x4) split area][planes] (diinn_rdn_workspace_floats = the split
 * area + 2,240 floats per pixel; diinn_rdn_x3_workspace_floats: + 576 per pixel), any content: as before v9 these entry points zero
 * the area's 4 KiB of control words themselves at every forward (so they keep no sticky status: a hand-off that gives up is NaN in
 * that forward's output only). */
size_t diinn_rdn_workspace_floats(int B, int H, int W);
size_t diinn_rdn_x3_workspace_floats(int B, int H, int W);
int    diinn_rdn_forward(void* stream, const float* sfe1_dev, const float* packed_dev, const float* biases_dev,
                         float* workspace_dev, float* out_dev, int B, int H, int W);
int    diinn_rdn_forward_wino(void* stream, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                              const float* biases_dev, float* workspace_dev, float* out_dev, int B, int H, int W);
int    diinn_rdn_forward_wino4(void* stream, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                               const float* packed_wino4_dev, const float* biases_dev, float* workspace_dev, float* out_dev,
                               int B, int H, int W);
int    diinn_rdn_forward_x3(void* stream, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                            const float* packed_x3_dev, const float* biases_dev, float* workspace_dev, float* out_dev,
                            int B, int H, int W);

/* Launch geometry of decode_kernel (the throughput kernel: workgroups of 16 x 8 pixels) for HR rows [y0, y1). */
int diinn_decode_launch_info(int B, int Hu, int Wu, int y0, int y1,
                             int* grid_x, int* grid_y, int* grid_z, int* block);
/* Which decode kernel diinn_decode_tile_win (and its wrappers) launch for HR rows [y0,y1) x columns [x0,x1) in `compute` on the
 * current device -- the launch function's own choice (cost model, CU count and DIINN_F32_KERNEL included), for benchmarks
 * that label a time or a roofline fraction with a kernel: info[0] = kernel, info[1..3] = its grid (x, y, z). */
#define DIINN_DECODE_KERNEL_THROUGHPUT 1   /* decode_kernel: 32 pixels per wave, activations register-resident           */
#define DIINN_DECODE_KERNEL_LATENCY16  3   /* decode_coop16_kernel: 16-pixel tiles, four workgroups per CU, bit-equal      */
#define DIINN_DECODE_KERNEL_OTHER      0   /* a bf16 / split-bf16 kernel (chosen from the full image: DIINN_BF16_KERNEL ...) */
int diinn_decode_kernel_info(int B, int Hu, int Wu, int y0, int y1, int x0, int x1, int compute, int info[4]);
/* Which form of the hoisted 3x3 convolution diinn_precompute_P_ex / _win / diinn_decode* run for LR rows [r0,r1) of a
 * [B,64,H,W] map in `compute` (the diagnostic overrides below included): what a benchmark should label its P time with. */
#define DIINN_P_ALGO_DIRECT      0   /* implicit-im2col GEMM, fp32 (1,179,648 FLOP per cell)                 */
#define DIINN_P_ALGO_WINOGRAD    1   /* Winograd F(2x2,3x3), fp32: 2.25x fewer MFMAs                          */
#define DIINN_P_ALGO_DIRECT_BF16 2   /* implicit-im2col GEMM on bf16 operands (DIINN_COMPUTE_BF16_FULL)       */
#define DIINN_P_ALGO_DIRECT_BF16X3 3 /* direct sum in split-bf16 arithmetic (DIINN_COMPUTE_BF16X3, maps >= 32,768 cells) */
int diinn_p_launch_info(int B, int H, int W, int r0, int r1, int compute, int* algorithm);

/* ---- diagnostic overrides (tests, A/B timing; never needed in production) -------------------------------------
 * The launch functions pick kernel variants by launch size; each choice can be forced.  The knobs are named like the
 * environment variables that seed them (csrc/diinn_knobs.h lists values and defaults):
 *   thresholds a deployment may tune -- DIINN_P_X3_MIN, DIINN_P_WINO_MIN, DIINN_ENC_WINO_MIN, DIINN_ENC_WINO4_MIN,
 *     DIINN_ENC_X3_MIN, DIINN_ENC_WINO4_SPLIT;
 *   TEST / A-B ONLY (force a kernel variant that the launch cost models would not take, or inject a fault; results stay
 *     within the documented equivalences) -- DIINN_F32_KERNEL, DIINN_BF16_KERNEL, DIINN_X3_KERNEL, DIINN_PBF16_KERNEL,
 *     DIINN_P_KERNEL, DIINN_ENC_X3_ROWS, DIINN_ENC_S1_MIN_BLOCKS, DIINN_ENC_NO_STREAM1X1, DIINN_ENC_LAT_MAX_TILES,
 *     DIINN_ENC_WINO_HALF_MAX, DIINN_ENC_WINO_PERSIST, DIINN_ENC_NO_T16 (1: small maps keep the split-K 3x3 kernel), DIINN_TRAIN_SPLIT_HEAD, DIINN_DEBUG_NCU (the compute-unit count cost models, split plans and persistent grids assume),
 *     DIINN_ENC_WINO4_FAULT (1: the split hand-off's give-up path on demand: NaN outputs + sticky status).  The environment is read ONCE, at the first launch or the first call of either function; afterwards
 * only diinn_debug_set changes a knob (process-wide, atomic stores: safe against concurrent launches, which see
 * either the old or the new value).  Unknown name -> DIINN_ERR_INVALID_ARG.  This is the library's only mutable
 * global state. */
int diinn_debug_set(const char* name, long long value);
int diinn_debug_get(const char* name, long long* value);
/* Diagnosis only (tools/clock_trace.py): enqueue ONE wave that takes n samples of the shader clock beside whatever else runs on
 * the device -- samples_dev[3 i] = s_memtime ticks (shader cycles) that passed while s_memrealtime (100 MHz) advanced by
 * samples_dev[3 i + 1] >= realtime_ticks, [3 i + 2] = the sample's start on the realtime counter.  clock = [3i] / [3i+1] x 100 MHz. */
int diinn_debug_clock_probe(void* stream, unsigned long long* samples_dev, int n, unsigned realtime_ticks);

#ifdef __cplusplus
}
#endif
#endif /* DIINN_HIP_H */
