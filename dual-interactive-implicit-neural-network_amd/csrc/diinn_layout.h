// diinn_layout.h -- packed-weight image layout and the coordinate formulas shared
// by the host packer (diinn_host.cpp) and the gfx950 kernels (diinn_*.hip via diinn_device.h).
//
// The decode kernel keeps the 256-channel activation q of 32 HR pixels in the
// registers of one wave, in the accumulator layout of v_mfma_f32_32x32x2_f32:
//   lane l = 32*h + j   holds pixel j, and for M-tile m (32 channels) register r
//   holds channel  32*m + (r&3) + 8*(r>>2) + 4*h                      [chan_of()]
// so an accumulator register is directly the B operand (k = h) of the next
// layer's MFMA.  All weight images are pre-permuted on the host to that order.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DIINN_HD __host__ __device__ __forceinline__
#else
#define DIINN_HD inline
#endif

namespace diinn {

constexpr int C_IN   = 64;    // encoder feature channels   (diinn.py:40 in_channels)
constexpr int HID    = 256;   // hidden width               (diinn.py:40 hidden_dims)
constexpr int NLAYER = 4;
constexpr int UNF    = C_IN * 9;          // 576 unfolded channels (diinn.py:168)
constexpr int PCH    = NLAYER * HID;      // 1024 floats of P per LR cell
constexpr int NOUT   = 3;

// ---- packed image sections (offsets in floats) --------------------------------
// WL: stacked per-pixel layers i=1..3.  [layer 3][m 8][kg 32][part 2][lane 64][e 4]
//     part 0 = q-half of K[i] (modulation), part 1 = Q[i] (synthesis);
//     value = W_part[ out = 32m + (lane&31) ][ in = chan_of(kk = 4kg+e, lane>>5) ]
constexpr int    WL_KG      = HID / 2 / 4;                 // 32 groups of 4 k-steps
constexpr size_t WL_PIECE   = 64 * 4;                      // floats per (lane,e) piece = 1 KiB
constexpr size_t WL_LAYER   = (size_t)8 * WL_KG * 2 * WL_PIECE;
constexpr size_t OFF_WL     = 0;
constexpr size_t SZ_WL      = 3 * WL_LAYER;                // 393,216 floats = 1.5 MiB
// WP: the hoisted 3x3 conv 64 -> 1024.  [mp 16][kg 72][t 2][lane 64][e 4], M-tile mo = 2mp+t
//     (step order of the P kernel: two M-tiles advance together through K)
//     k-step kk = 4kg+e: tap = kk/32 (= ky*3+kx), channel c = 2*(kk%32) + (lane>>5);
//     value = Wx[ o = 32mo + (lane&31) ][ c ][ ky ][ kx ],  o = i*256 + ch
constexpr int    WP_KSTEPS  = UNF / 2;                     // 288
constexpr int    WP_KG      = WP_KSTEPS / 4;               // 72
constexpr size_t OFF_WP     = OFF_WL + SZ_WL;
constexpr size_t SZ_WP      = (size_t)32 * WP_KG * WL_PIECE;   // 589,824 floats = 2.25 MiB
// small tables, natural channel order
constexpr size_t OFF_BK     = OFF_WP + SZ_WP;              // bK[4][256]
constexpr size_t OFF_Q0     = OFF_BK + 4 * HID;            // Q0h[256] Q0w[256] Q0r[256] bQ0[256]
constexpr size_t OFF_BQ     = OFF_Q0 + 4 * HID;            // bQ[1..3][256]
constexpr size_t OFF_L      = OFF_BQ + 3 * HID;            // L[3][256]
constexpr size_t OFF_BL     = OFF_L + 3 * HID;             // bL[3] + pad
// WLB: the per-pixel layers again, as bf16 A operands of v_mfma_f32_32x32x16_bf16 (optional
//     reduced-precision path, BASELINE config 5).  [layer 3][m 8][ks 16][part 2][lane 64][j 8] bf16;
//     value = bf16(W_part[ out = 32m + (lane&31) ][ in = chan_of_bf16(ks, lane>>5, j) ]), 16 B per lane;
//     part 1 (synthesis rows) is bf16(W * fp32(1/(2 pi))): see BQR below.
constexpr size_t WLB_PIECE  = 64 * 4;                      // floats (= 64 lanes x 8 bf16) per piece = 1 KiB
constexpr size_t WLB_LAYER  = (size_t)8 * 16 * 2 * WLB_PIECE;
constexpr size_t OFF_WLB    = OFF_BL + 4;
constexpr size_t SZ_WLB     = 3 * WLB_LAYER;               // 196,608 floats = 768 KiB
// WLT: the per-pixel layers transposed, for the backward pass (training): g_q[i-1] = Wq_i^T g_a + Qw_i^T g_s.
//     Same shape as WL, [layer 3][m 8][kg 32][part 2][lane 64][e 4], with the roles of the two channel
//     indices swapped: value = W_part[ out = chan_of(4kg+e, lane>>5) ][ in = 32m + (lane&31) ].
constexpr size_t OFF_WLT    = OFF_WLB + SZ_WLB;
constexpr size_t SZ_WLT     = SZ_WL;
// WPB: the hoisted 3x3 conv as bf16 A operands of v_mfma_f32_32x32x16_bf16 (DIINN_COMPUTE_BF16_FULL).
//     [mp 16][ks 36][t 2][lane 64][j 8] bf16, M-tile mo = 2mp+t; k-step ks = 4*tap + cg covers channels
//     16cg .. 16cg+15 of tap = ky*3+kx;  value = bf16(Wx[ o = 32mo + (lane&31) ][ c = 16cg + 8(lane>>5) + j ][ky][kx]).
constexpr int    WPB_KS     = 36;
constexpr size_t OFF_WPB    = OFF_WLT + SZ_WLT;
constexpr size_t SZ_WPB     = (size_t)16 * WPB_KS * 2 * WL_PIECE;    // 294,912 floats
// BQR: the synthesis biases of layers 1..3 in REVOLUTIONS, bQ_i / (2 pi), for the bf16 decode kernels: there the
//     synthesis rows of WLB are stored pre-multiplied by 1/(2 pi) as well, so the accumulator already holds the
//     argument v_sin_f32 wants (revolutions) and the epilogue needs no range-reduction arithmetic.
constexpr size_t OFF_BQR    = OFF_WPB + SZ_WPB;
// Q0R: the Q0 table (Q0h, Q0w, Q0r, bQ0; [4][256]) in revolutions as well, for the cooperative bf16 kernel
constexpr size_t OFF_Q0R    = OFF_BQR + 3 * HID;
// WLR: WL with the synthesis rows (part 1) multiplied by fp32(1/(2 pi)): the fp32 inference kernels (decode_kernel
//     without SAVE, decode_coop16_kernel) keep the synthesis branch in revolutions too, which turns the epilogue's sine
//     into v_fract + v_sin (on gfx950 every VALU instruction costs fp32-MFMA issue time).  WL itself stays a pure
//     permutation of the reference tensors: the training forward saves sine arguments in radians, LIIF reads the
//     synthesis slots as plain MLP weights, and the device re-pack of a training step is a gather.
constexpr size_t OFF_WLR    = OFF_Q0R + 4 * HID;
// WPU: the hoisted 3x3 conv in Winograd F(2x2, 3x3) form, U = G Wx G^T (4 x 4 per output / input channel pair; G folds
//     the halves), for precompute_P_wino_kernel.  [mt 32][row i 4][sg 8][col j 4][lane 64][e 4]; M-tile mt = outputs
//     32 mt .. 32 mt + 31, k-step 4 sg + e = input channels 8 sg + 2 e + (lane>>5);
//     value = s_j U[ o = 32 mt + (lane&31) ][ c ][ i ][ j ],  s_2 = -1 (the kernel's input transform yields column 2
//     negated), else 1.  Derived (float64, rounded once): an inference-only section like WLR.
constexpr size_t OFF_WPU    = OFF_WLR + SZ_WL;
constexpr size_t SZ_WPU     = (size_t)32 * 4 * 8 * 4 * WL_PIECE;   // 1,048,576 floats = 4 MiB
// WLX: the per-pixel layers for the split-bf16 arithmetic (DIINN_COMPUTE_BF16X3): every weight w (synthesis rows:
//     w / (2 pi)) is carried as hi + lo with hi = bf16(w) (the value section WLB holds) and lo = bf16(w - hi); the kernel
//     adds hi*hi + hi*lo + lo*hi on the bf16 MFMA.  [layer 3][m 8][ks 16][piece 4][lane 64][j 8] bf16 with the pieces of
//     a k-step -- k_hi, q_hi, k_lo, q_lo -- contiguous (4 KiB: one scalar offset per k-step, the pieces by immediate
//     offsets); lane / j as in WLB.  Derived, inference only.
constexpr size_t WLX_KSTEP  = 4 * WLB_PIECE;               // floats per k-step
constexpr size_t WLX_LAYER  = (size_t)8 * 16 * WLX_KSTEP;
constexpr size_t OFF_WLX    = OFF_WPU + SZ_WPU;
constexpr size_t SZ_WLX     = 3 * WLX_LAYER;               // 393,216 floats = 1.5 MiB
// WPX: the hoisted 3x3 conv for the split-bf16 arithmetic (DIINN_COMPUTE_BF16X3): [og 16][group 4][tap 9][M-tile 2][hi, lo]
//     [lane 64][j 8] bf16: output channel 64 og + 32 mt + (lane&31) of the 1024, input channel 16 group + 8 (lane>>5) + j,
//     hi = bf16(w), lo = bf16(w - hi).  Derived, inference only.
constexpr size_t OFF_WPX    = OFF_WLX + SZ_WLX;
constexpr size_t SZ_WPX     = (size_t)16 * 4 * 9 * 2 * 2 * WLB_PIECE;   // 589,824 floats = 2.25 MiB
// WL16: the per-pixel layers as A operands of v_mfma_f32_16x16x4_f32 (decode_coop16_kernel: the fp32 latency form for the
//     smallest launches, a 16-pixel tile per workgroup).  [layer 3][wave 4][i 64][half 2][lane 64][T 4]: wave w owns output
//     channels 64 w .. 64 w + 63 of both branches as four 16-row M-tiles T; half 0 = modulation rows (K.i[:, :256]), half 1
//     = synthesis rows (Q.i, in revolutions like WLR); k-step i covers the four activations at POSITIONS 4 i + g, g = lane >> 4,
//     where position p is the p-th term of decode_kernel's accumulation order: channel chan_of(p >> 1, p & 1);
//     value = W_half[ out = 64 w + 16 T + (lane & 15) ][ in = chan_of((4 i + g) >> 1, g & 1) ].  Derived, inference only.
constexpr size_t WL16_KSTEP = 2 * WL_PIECE;                // floats per (wave, k-step): the two 1 KiB pieces
constexpr size_t WL16_WAVE  = 64 * WL16_KSTEP;
constexpr size_t WL16_LAYER = 4 * WL16_WAVE;
constexpr size_t OFF_WL16   = OFF_WPX + SZ_WPX;
constexpr size_t SZ_WL16    = 3 * WL16_LAYER;              // 393,216 floats = 1.5 MiB
constexpr size_t PACKED_FLOATS = OFF_WL16 + SZ_WL16;       // 4,691,204

// channel held by activation register (m, r) of lane-half h
DIINN_HD int chan_of(int kk /* = 16*m + r */, int h) {
    const int m = kk >> 4, r = kk & 15;
    return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// bf16 path: k-step ks = 2*m' + s covers 16 channels of activation tile m'; MFMA k index 8*h + j
// is accumulator register 8s + j of lane-half h (cdna_hip_programming.md, "An accumulator tile as
// the next MFMA's operand"), i.e. channel 32m' + 16s + 8(j>>2) + 4h + (j&3).
DIINN_HD int chan_of_bf16(int ks, int h, int j) {
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
}

// ---- coordinates (reference: diinn.py:94-110, ATen nearest-exact)
struct Axis {
    float c0_in, c1_in;     // fp32(-1 + 1/n_in), fp32(2/n_in)      (python doubles -> fp32)
    float c0_out, c1_out;   // same for n_out
    float n_in_f;           // fp32(n_in)
    float scale;            // fp32(n_in) / fp32(n_out)
    int   n_in;
    int   small_output;     // ATen small-output kernel rounding (Hu + Wu <= 128)
};

inline Axis make_axis(int n_in, int n_out, int small_output) {
    Axis a;
    a.c0_in  = (float)(-1.0 + 1.0 / (double)n_in);
    a.c1_in  = (float)(2.0 / (double)n_in);
    a.c0_out = (float)(-1.0 + 1.0 / (double)n_out);
    a.c1_out = (float)(2.0 / (double)n_out);
    a.n_in_f = (float)n_in;
    a.scale  = (float)n_in / (float)n_out;
    a.n_in = n_in;
    a.small_output = small_output;
    return a;
}

// Index of the LR sample and relative coordinate of HR sample j along one axis.
// Every operation is an individually rounded fp32 (or fp64) op in the reference's
// order; translation units including this header are built with -ffp-contract=off.
DIINN_HD void axis_eval(const Axis& a, int j, int& idx, float& rel) {
    const float jf = (float)j;
    int id;
    if (a.small_output) {
        const double p = ((double)j + 0.5) * (double)a.scale;
        id = (int)__builtin_floorf((float)p);
    } else {
        float r = __builtin_fmaf(a.scale, jf + 0.5f, -0.5f);   // single rounding (ATen builds with FMA)
        r = r < 0.0f ? 0.0f : r;
        id = (int)__builtin_floorf((float)((double)r + 0.5));
    }
    id = id < a.n_in - 1 ? id : a.n_in - 1;
    const float g_out = a.c1_out * jf + a.c0_out;               // mul, then add (not contracted)
    const float g_in  = a.c1_in * (float)id + a.c0_in;
    idx = id;
    rel = (g_out - g_in) * a.n_in_f;
}

// ---- LIIF comparison decoder (reference liif.py:59-127): nearest cell of the coordinate shifted by
// v/n_in + 1e-6 (v = -1, +1), through grid_sample(mode='nearest', align_corners=False) as ATen's
// vectorised CPU kernel evaluates it: idx = nearbyint((c + 1) * (n_in / 2) - 0.5), half to even.
struct LiifAxis {
    Axis  base;             // centres of both grids (make_coord == the DIINN centres)
    float shift[2];         // fp32(v / n_in + 1e-6) for v = -1, +1   (liif.py:91-92)
    float lo, hi;           // fp32(-1 + 1e-6), fp32(1 - 1e-6)         (liif.py:93)
    float half_n;           // fp32(n_in) / 2
    float rel_cell;         // fp32(2 / n_out) * fp32(n_in)            (liif.py:55-56,108-110)
};

inline LiifAxis make_liif_axis(int n_in, int n_out) {
    LiifAxis a;
    a.base = make_axis(n_in, n_out, 0);
    a.shift[0] = (float)(-1.0 * (2.0 / (double)n_in / 2.0) + 1e-6);
    a.shift[1] = (float)(1.0 * (2.0 / (double)n_in / 2.0) + 1e-6);
    a.lo = (float)(-1.0 + 1e-6);
    a.hi = (float)(1.0 - 1e-6);
    a.half_n = (float)n_in / 2.0f;
    a.rel_cell = (float)(2.0 / (double)n_out) * (float)n_in;
    return a;
}

// vi = 0 (v = -1) or 1 (v = +1)
DIINN_HD void liif_axis_eval(const LiifAxis& a, int j, int vi, int& idx, float& rel) {
    const float c = a.base.c1_out * (float)j + a.base.c0_out;
    float cs = c + a.shift[vi];
    cs = cs < a.lo ? a.lo : (cs > a.hi ? a.hi : cs);
    const float x = (cs + 1.0f) * a.half_n - 0.5f;
    int id = (int)__builtin_rintf(x);
    id = id < 0 ? 0 : (id > a.base.n_in - 1 ? a.base.n_in - 1 : id);       // in range by construction; guards the gather
    const float q = a.base.c1_in * (float)id + a.base.c0_in;
    idx = id;
    rel = (c - q) * a.base.n_in_f;
}

// ---- MetaSR comparison decoder (reference metasr.py:70-104)
// Packed image of its own: W2 [o 3][mm 18][kg 32][lane 64][e 4] = imnet.layers.2.weight[n = 3k + o][chan_of(4kg+e, lane>>5)],
// k = 32mm + (lane&31): output channels regrouped by RGB component so that an M-tile feeds one
// component and its rows are 32 consecutive unfolded-feature indices k; then the first layer as a
// [4][256] table (columns rel_h, rel_w, r_rev, bias) and the second bias as [3][576].
constexpr int    MS_K       = UNF;                             // 576 predicted weights per RGB component
constexpr int    MS_MM      = MS_K / 32;                       // 18 M-tiles per component
constexpr size_t MS_OFF_W2  = 0;
constexpr size_t MS_SZ_W2   = (size_t)3 * MS_MM * WL_KG * WL_PIECE;   // 442,368 floats
constexpr size_t MS_OFF_Q0  = MS_OFF_W2 + MS_SZ_W2;
constexpr size_t MS_OFF_B2  = MS_OFF_Q0 + 4 * HID;
constexpr size_t MS_PACKED_FLOATS = MS_OFF_B2 + 3 * MS_K;      // 445,120

struct MetaAxis {
    Axis  base;
    float half_cell;        // fp32(2 / n_out) / 2                     (metasr.py:64-66,80-82)
    float eps, lo, hi;      // fp32(1e-6), fp32(-1 + 1e-6), fp32(1 - 1e-6)   (metasr.py:83)
    float half_n;           // fp32(n_in) / 2  (grid_sample unnormalise)
    float corner;           // fp32((2 / n_in) / 2): feat_coord is shifted to the cell corner (metasr.py:75-76)
    float rel_scale;        // fp32(n_in / 2)                          (metasr.py:94-95)
    float r_rev;            // fp32(2 / n_out) * fp32(n_in / 2)        (metasr.py:97; rows axis)
};

inline MetaAxis make_meta_axis(int n_in, int n_out) {
    MetaAxis a;
    a.base = make_axis(n_in, n_out, 0);
    a.half_cell = (float)(2.0 / (double)n_out) / 2.0f;
    a.eps = (float)1e-6;
    a.lo = (float)(-1.0 + 1e-6);
    a.hi = (float)(1.0 - 1e-6);
    a.half_n = (float)n_in / 2.0f;
    a.corner = (float)((2.0 / (double)n_in) / 2.0);
    a.rel_scale = (float)((double)n_in / 2.0);
    a.r_rev = (float)(2.0 / (double)n_out) * a.rel_scale;
    return a;
}

DIINN_HD void meta_axis_eval(const MetaAxis& a, int j, int& idx, float& rel) {
    const float c = a.base.c1_out * (float)j + a.base.c0_out;
    const float c_ = c - a.half_cell;
    float cq = c_ + a.eps;
    cq = cq < a.lo ? a.lo : (cq > a.hi ? a.hi : cq);
    const float x = (cq + 1.0f) * a.half_n - 0.5f;
    int id = (int)__builtin_rintf(x);
    id = id < 0 ? 0 : (id > a.base.n_in - 1 ? a.base.n_in - 1 : id);
    const float q = (a.base.c1_in * (float)id + a.base.c0_in) - a.corner;
    idx = id;
    rel = (c_ - q) * a.rel_scale;
}

}  // namespace diinn
