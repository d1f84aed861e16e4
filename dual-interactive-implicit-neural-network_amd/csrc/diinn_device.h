// diinn_device.h -- what the gfx950 translation units of libdiinn_hip.so share: vector types, the
// sine variants, the weight-stream load, tile geometry, the tiled-plane helpers of the training
// path, and the small host-side helpers of the C ABI launch functions.
//
// Translation units (all built with hipcc --offload-arch=gfx950 -O3 -ffp-contract=off; build.py: HIP_SOURCES + HOST_SOURCES):
//   diinn_decode.hip           decode_kernel (+ modes 1/2 chain, training forward), decode_coop16_kernel (latency form),
//                              and the decode entry points of the C ABI
//   diinn_precompute.hip       precompute_P_kernel (direct fp32), precompute_P_bf16_kernel / _bf16_wide_kernel, launch_P
//   diinn_precompute_wino.hip  precompute_P_wino_kernel (the fp32 hoisted conv in Winograd F(2x2,3x3) form: inference)
//   diinn_precompute_x3.hip    precompute_P_x3_kernel (the hoisted conv in split-bf16 arithmetic: DIINN_COMPUTE_BF16X3, large maps)
//   diinn_bf16.hip             decode_bf16_kernel, decode_bf16x2_kernel, decode_bf16_coop_kernel (4 waves),
//                              decode_bf16_coop8_kernel (8 waves), decode_bf16_coop8p_kernel (8 waves, persistent)
//   diinn_bf16x3.hip           decode_bf16x3h_kernel (persistent, hi weight pieces through LDS), decode_bf16x3_kernel: split bf16
//   diinn_training.hip         backward pass: bwd_head / bwd_layer, plane_gemm, plane_rowdot, cell_sum
//   diinn_baselines.hip        LIIF and MetaSR comparison decoders
//   diinn_encoder.hip          RDN trunk: conv_ksplit kernels (small maps), conv1x1_stream_kernel, sfe1_conv_kernel
//   diinn_winograd.hip         RDN trunk: conv_wino_kernel / conv_wino_half_kernel (3x3 layers, Winograd F(2x2,3x3))
//   diinn_conv_x3.hip          RDN trunk, optional split-bf16 arithmetic: conv3x3_x3m_kernel / conv3x3_x3_kernel (3x3 layers), conv1x1_x3_kernel (fusion)
//   diinn_misc.hip             device sine / axis-table test hooks, error state
//   diinn_host.cpp (host only) weight packing, coordinate tables, size queries, the knob table (diinn_knobs.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diinn_hip.h"
#include "diinn_layout.h"
#include "diinn_knobs.h"

using namespace diinn;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int N> struct IC { static constexpr int value = N; };   // compile-time index for generic lambdas

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------
// sine (reference: torch.sin via SineAct, diinn.py:21-26)
// ---------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float dsin(float x);

// Cody-Waite reduction to [-pi/2, pi/2] by multiples of pi, odd minimax polynomial.
// Absolute error <= ~2.5e-7 for |x| <= 1e4 (arguments here are O(1)..O(100)).
template <>
__device__ __forceinline__ float dsin<DIINN_SIN_ACCURATE>(float x) {
    const float k = __builtin_rintf(x * 0.31830988618379067154f);
    float r = __builtin_fmaf(k, -3.14159274101257324219f, x);      // pi, fp32 head
    r = __builtin_fmaf(k, 8.74227765734758577309e-08f, r);          // -(pi - head)
    const float s = r * r;
    // flip sign for odd k: (-1)^k
    const int ki = (int)k;
    r = __builtin_bit_cast(float, __builtin_bit_cast(int, r) ^ (ki << 31));
    float u = 2.6083159809786593541503e-06f;
    u = __builtin_fmaf(u, s, -0.0001981069071916863322258f);
    u = __builtin_fmaf(u, s, 0.00833307858556509017944336f);
    u = __builtin_fmaf(u, s, -0.166666597127914428710938f);
    return __builtin_fmaf(s, u * r, r);
}

// v_sin_f32 takes revolutions; fract keeps it inside the instruction's valid domain.
// The product x/(2 pi) is rounded to fp32 before the reduction, so the absolute error grows
// like |x| * 6e-8: fine for O(1)..O(100) arguments, the fastest form (3 VALU ops).
template <>
__device__ __forceinline__ float dsin<DIINN_SIN_HW>(float x) {
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189533577f));
}

// Reduction done in revolutions, Cody-Waite style: t = x * c with c = 1/(2 pi) split into an fp32
// head and tail; r = fma(x, c_hi, -rint(t)) is the exact head product minus an integer, the tail
// adds the rest.  Then v_sin_f32 on r in [-0.5, 0.5]: 5 VALU ops, error independent of |x|
// (measured max abs error ~4e-7 for |x| <= 1e4).
template <>
__device__ __forceinline__ float dsin<DIINN_SIN_HW_REDUCED>(float x) {
    constexpr float C_HI = 0.15915494309189533577f;                  // fp32(1/(2 pi)) = 0.159154936671257019...
    constexpr float C_LO = 6.4206383650924e-09f;                     // 1/(2 pi) - C_HI
    const float k = __builtin_rintf(x * C_HI);
    float r = __builtin_fmaf(x, C_HI, -k);
    r = __builtin_fmaf(x, C_LO, r);
    return __builtin_amdgcn_sinf(r);
}

// The same three modes for an argument already in REVOLUTIONS (the bf16 kernels: synthesis weights and biases
// are stored pre-divided by 2 pi, packed sections 7 / 10): HW is the bare instruction (valid for |x| <= 256
// revolutions), HW_REDUCED adds the exact reduction x - rint(x) that makes it valid for any magnitude, ACCURATE converts back to
// radians and runs the polynomial.
template <int MODE>
__device__ __forceinline__ float dsin_rev(float x);
template <>
__device__ __forceinline__ float dsin_rev<DIINN_SIN_HW>(float x) { return __builtin_amdgcn_sinf(x); }
template <>
__device__ __forceinline__ float dsin_rev<DIINN_SIN_HW_REDUCED>(float x) {
    // x - rint(x) is exact and keeps the precision of small arguments (fract maps a small negative x to 1 - |x|,
    // whose ulp is 6e-8: 7e-8 output error against 3e-8, measured on the fixtures)
    return __builtin_amdgcn_sinf(x - __builtin_rintf(x));
}
template <>
__device__ __forceinline__ float dsin_rev<DIINN_SIN_ACCURATE>(float x) {
    return dsin<DIINN_SIN_ACCURATE>(x * 6.28318530717958647692f);
}

// ---------------------------------------------------------------------------------
// decode kernel
// ---------------------------------------------------------------------------------
struct DecodeParams {
    const float* P;        // [B,H,W,1024]
    const float* Wt;       // packed image
    float* out;            // [B,3,Hu,Wu]
    int B, H, W, Hu, Wu, y0, y1;
    // row windows (include/diinn_hip.h "row windows"): P holds LR rows [Prow0, Prow0+Prows), out holds HR rows
    // [Orow0, Orow0+Orows); the full-buffer entry points pass (0, H) and (0, Hu)
    int Prow0, Prows, Orow0, Orows;
    // column range and output strides (include/diinn_hip.h "tiles"): the launch covers HR columns [x0, x1) -- blocks are
    // anchored at x0, which changes no pixel's arithmetic -- and pixel (b, c, y, x) is written to
    // out[b * o_bs + c * o_ps + (y - Orow0) * o_rs + (x - Ocol0)]; the row-band entry points pass (0, Wu), Ocol0 = 0 and
    // the strides of a contiguous [B,3,Orows,Wu] tensor
    int x0, x1, Ocol0;
    long long o_bs, o_ps, o_rs;
    int seed_cols;         // decode_bf16_coop_kernel: LR columns of a block's footprint (row length of its LDS seed slab)
    int xcd_runs;          // decode_bf16_coop8_kernel: walk the blocks XCD by XCD (set when neighbouring blocks share P rows)
    int pg[6];             // decode_bf16_coop8p_kernel: block grid (x, y, z), super-tile grid (x, y), super-tiles per XCD
    float ratio;           // fp32(H*W / (Hu*Wu))   (diinn.py:166)
    Axis ah, aw;
    float* acts;           // training forward only (SAVE): saved activations, tiled planes [4 layers][ntiles][512][32]
    long long npix;        // SAVE: B*Hu*Wu
#ifdef DIINN_STAMPS
    unsigned long long* stamps;   // diagnostic build only: 8 x u64 per wave (never in the shipped library)
#endif
};

#ifdef DIINN_STAMPS
// In-kernel stamps (cdna_hip_programming.md section 7): one asm statement, fenced, values go to a
// buffer nothing else reads.  STAMP(i) records s_memtime; slot 7 records s_memrealtime (100 MHz).
#define STAMP(i)                                                                              \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (p.stamps && lane == 0) p.stamps[stamp_base + (i)] = t_;                           \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// colour plane 0 of HR pixel (b, y, x) in the caller's output; the other two planes are p.o_ps floats apart
__device__ __forceinline__ float* out_px(const DecodeParams& p, int b, int y, int x) {
    return p.out + (long long)b * p.o_bs + (long long)(y - p.Orow0) * p.o_rs + (x - p.Ocol0);
}

constexpr int TILE_W = 8, TILE_H = 4;           // one wave: 8x4 HR pixels
#ifndef WSTREAM_AUX
#define WSTREAM_AUX 0                           // cache-policy bits of the weight-stream loads (sc0=1, nt=2, sc1=16)
#endif
#ifndef P_PREFETCH
#define P_PREFETCH 4
#endif
#ifndef DECODE_PREFETCH
#define DECODE_PREFETCH 4                       // weight ring depth, in steps of 8 MFMAs
#endif

// Validity word of the packed image (include/diinn_hip.h "VALIDITY WORD"): 0 when the derived sections are filled, the
// quiet-NaN bit pattern otherwise.  OR-ing it into the bits of a float leaves the float alone or turns it into a NaN;
// the inference kernels that read a derived section fold it into a bias they add to every output, outside their
// inner loops.  Integer operations because they are the cheapest way to force a NaN, not because float NaN handling is
// unreliable: the translation units are built WITHOUT -fno-honor-nans (relu0 below relies on NaN propagation; build.py
// says why the flag must not come back).
__device__ __forceinline__ unsigned derived_nan_mask(const float* __restrict__ Wt) {
    return __builtin_bit_cast(unsigned, Wt[OFF_BL + 3]) == DIINN_PACKED_MAGIC ? 0u : 0x7fc00000u;
}
__device__ __forceinline__ float or_bits(float v, unsigned m) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) | m);
}
__device__ __forceinline__ f32x4 or_bits(f32x4 v, unsigned m) {
    return f32x4{or_bits(v[0], m), or_bits(v[1], m), or_bits(v[2], m), or_bits(v[3], m)};
}

// relu as ONE instruction that PROPAGATES NaN, like the reference's torch.relu (diinn.py:133-138: a NaN feature or
// weight makes the output NaN there, and so it does here): llvm.maximum -> v_maximum3_f32 x, 0, 0, gfx950's
// IEEE-754-2019 maximum.  (v_max_f32 returns the other operand for a NaN: relu would turn a NaN modulation value into
// k = 0 and the pixel into a finite colour.  Rounds 1-2 used v_max_f32 -- first as inline asm, which gets no
// MFMA -> VALU wait states from hipcc and read accumulators too early once a kernel evaluated its epilogue right behind
// the MFMAs, then as fmaxf under -fno-honor-nans.)  The translation units are built WITHOUT -fno-honor-nans now.
__device__ __forceinline__ float relu0(float x) { return __builtin_elementwise_maximum(x, 0.0f); }

// Weight-stream loads go through a buffer descriptor: address = SGPR descriptor base + SGPR byte
// offset (scalar unit) + one constant per-lane VGPR offset, so the stream costs no VALU address
// arithmetic.  That matters here: on gfx950 the fp32 MFMA shares its issue/datapath with the VALU
// (tools/ubench/mfma_rate.hip: every VALU op between MFMAs costs ~3.2 cycles of MFMA time).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld_piece(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, WSTREAM_AUX));
}
constexpr int PIECE_BYTES = (int)(WL_PIECE * sizeof(float));
// L2 warm-up touch: one dword per lane from (descriptor base + voff), deposited by LDS-DMA into a dead 256-byte LDS area of the wave.
// There is NO destination register, so a late arrival has nothing to overwrite.  (Rounds 2-5 issued these loads from inline asm into
// registers: the compiler takes an asm output for written AT the statement and is free to copy it out and reuse the register while
// the load is still in flight.  It did, once an unrelated edit of round 6 changed precompute_P_wino_kernel's allocation -- the fourth
// load's destination became the patch loads' offset register: P came out wrong and different from run to run on maps that warm up.
// cdna_hip_programming.md 5.7: what hipcc does not do for an asm statement.)
__device__ __forceinline__ void l2_touch(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, float* sink_wave) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)sink_wave, 4, (int)voff, 0, 0, 0);
}
constexpr int WG_TILES_X = 2, WG_TILES_Y = 2;   // 4 waves -> 16x8 HR pixels per workgroup

// KPART = false (decoder modes 1 and 2, diinn.py:116-131): the modulation chain k_i depends on the LR
// cell only, so the caller leaves k_i[cell] (already rectified) in the workspace slot of P_i and the
// per-pixel layers run the synthesis GEMM alone: half the MFMAs, half the weight stream.
//
// SAVE = true (training forward, reference step() under autograd: diinn.py:132-139 called with
// bsize=None from sr_module.py:127-129): the same network, and every layer's rectified modulation
// k_i and sine argument s_i are written to p.acts for the backward pass.  One wave then owns 32
// consecutive pixels of the flattened (b, y, x) index -- one PLANE TILE.
//
// Training planes are stored tiled: a group of C channel rows over npix pixels is
// [ceil(npix/32) tiles][C rows][32 pixels], element (c, pix) at ((pix >> 5) * C + c) * 32 + (pix & 31).
// Everything a wave touches for its 32 pixels is one contiguous block (64 KiB for C = 512), every
// row segment is a full 128-byte line, and the weight-gradient GEMM over the pixel axis reads
// contiguous [rows][32] panels (with plain [C][npix] planes each of its loads touched 32 rows
// megabytes apart: 1.6x slower, measured).
constexpr int PLANE_TILE = 32;
constexpr int ACT_ROWS = 2 * HID;                               // rows 0..255: k_i (or g_a,i); 256..511: s_i (or g_s,i)
constexpr unsigned PLANE_ROW_BYTES = PLANE_TILE * sizeof(float);   // 128
// (aux 2 = nt: the training planes -- 4.8 GB forward, 1.8 GB per backward layer at B = 16, 192 x 192 -- are read back by LATER kernels
// only after all of them are written and do not fit any cache: streaming stores leave the L2 to the weight stream.  Round 6, one box,
// A/B: training forward 4.09 -> 3.98 ms, backward chain 4.92 -> 4.85, step 14.65 -> 14.55; write-through (sc1) instead: +0.25 ms.)
#ifndef ST_ACT_AUX
#define ST_ACT_AUX 2
#endif
__device__ __forceinline__ void st_act(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)voff, (int)soff, ST_ACT_AUX);
}
// Workgroups are dealt round-robin over the 8 XCDs in launch order (x fastest): blocks b and b + 8 share an L2.  This
// gives every XCD a contiguous run of the launch's blocks instead, so that neighbouring blocks -- which at non-integer
// scales read the same P rows -- meet in ONE L2.  A bijection of the 3-D grid onto itself (the last < 8 blocks keep
// their place).
struct BlockXYZ { int x, y, z; };
__device__ __forceinline__ BlockXYZ xcd_run_block() {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z, full = total & ~7u;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned t = lin < full ? (lin & 7u) * (full >> 3) + (lin >> 3) : lin;
    const unsigned z = t / (gx * gy), r = t - z * gx * gy, y = r / gx;
    return BlockXYZ{__builtin_amdgcn_readfirstlane((int)(r - y * gx)), __builtin_amdgcn_readfirstlane((int)y), __builtin_amdgcn_readfirstlane((int)z)};
}

// 128-bit buffer store with a scalar offset register.  Measured on gfx950 (tools/: the LFF layer's second destination
// came out wrong in 16 lanes, first dword, some launches): a VALU instruction that writes one of the store's data
// registers in the slot right behind the store overwrites what the store's first quarter-wave still has to read.
// hipcc inserts a wait state for this hazard only when the store has NO offset register (the documented condition),
// so the data registers are pinned across an s_nop here.
__device__ __forceinline__ void st_b128(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)voff, (int)soff, 0);
    asm volatile("s_nop 1" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
}
__device__ __forceinline__ float ld_act(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}
// descriptor of one tile (rows x 32 floats) of a tiled plane group
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float* group, long long tile, int rows) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(group + (size_t)tile * rows * PLANE_TILE), 0,
                                             rows * (int)PLANE_ROW_BYTES, 0x00020000);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------
// host side of the launch functions
// ---------------------------------------------------------------------------------
__attribute__((visibility("hidden"))) extern thread_local int g_last_hip_error;       // diinn_misc.hip
#ifdef DIINN_STAMPS
extern unsigned long long* g_stamps;            // diinn_misc.hip
#endif

__attribute__((visibility("hidden"))) int device_cus();                                // diinn_misc.hip
// diinn_winograd4.hip: the F(4x4,3x3) kernel's time for `total` work items, in rounds of whole items on the compute units it plans for (w4_cus)
__attribute__((visibility("hidden"))) double w4_rounds(long long total, bool have_ws);

static inline int hip_status(hipError_t e) {
    if (e == hipSuccess) return DIINN_OK;
    g_last_hip_error = (int)e;
    return DIINN_ERR_HIP;
}

// the arithmetic modes of the C ABI (include/diinn_hip.h)
static inline bool compute_ok(int compute) {
    return compute == DIINN_COMPUTE_F32 || compute == DIINN_COMPUTE_BF16 || compute == DIINN_COMPUTE_F32_QONLY ||
           compute == DIINN_COMPUTE_BF16_FULL || compute == DIINN_COMPUTE_BF16X3;
}

// x / d as a multiplication, for the tile arithmetic in kernel preambles (an integer division there is ~40 dependent instructions,
// ~0.1 us in front of a launch's first load): m = 2^32 / d + 1 is exact for 0 <= x < 65536, 1 < d < 65536; m == 0: divide.
__device__ __forceinline__ int div_m(int x, int d, unsigned m) { return m ? (int)(((unsigned long long)(unsigned)x * m) >> 32) : x / d; }
static inline unsigned magic_m(long long d, long long x_max) { return (d > 1 && d < 65536 && x_max < 65536) ? (unsigned)((1ull << 32) / (unsigned long long)d + 1) : 0u; }
static inline int check_dims(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return DIINN_ERR_INVALID_ARG;
    if ((double)B * H * W * PCH >= 9.0e18 || B > 65535 || H > 65535) return DIINN_ERR_TOO_LARGE;
    return DIINN_OK;
}

// the window [row0, row0+rows) must lie inside [0, full) and contain [need0, need1)
static inline int check_window(int row0, int rows, int full, int need0, int need1) {
    if (row0 < 0 || rows <= 0 || row0 > full - rows) return DIINN_ERR_INVALID_ARG;
    if (need0 < row0 || need1 > row0 + rows) return DIINN_ERR_INVALID_ARG;
    return DIINN_OK;
}

static inline int check_npix(long long npix) {
    if (npix <= 0) return DIINN_ERR_INVALID_ARG;
    if (npix > DIINN_TRAIN_MAX_PIXELS) return DIINN_ERR_TOO_LARGE;
    return DIINN_OK;
}

// diinn_precompute.hip: the hoisted conv for mp_total M-tile pairs (16 = all 1024 channels), fp32 or bf16 operands
struct RowWin { int row0, rows; };             // rows [row0, row0+rows) of a full-size tensor, stored on their own
__attribute__((visibility("hidden")))
int launch_P(void* stream, const float* feat_dev, const float* packed_dev, float* P_dev,
             int B, int H, int W, int r0, int r1, int mp_total, int arith = 0,
             const RowWin* feat_win = nullptr, const RowWin* p_win = nullptr, bool derived_ok = false);
// `arith` of launch_P: 0 fp32, 1 bf16 operands (DIINN_COMPUTE_BF16_FULL), 2 split bf16 (DIINN_COMPUTE_BF16X3 -- and, round 6,
// DIINN_COMPUTE_BF16: that mode's contract is "P at fp32 accuracy", which the split-bf16 P kernel holds (it is bound by the
// fp32 path's own 1e-4 tolerance) at 0.67x the fp32 Winograd kernel's time; below DIINN_P_X3_MIN cells both fall to Winograd)
static inline int p_arith(int compute) {
    return compute == DIINN_COMPUTE_BF16_FULL ? 1 : (compute == DIINN_COMPUTE_BF16X3 || compute == DIINN_COMPUTE_BF16) ? 2 : 0;
}
// diinn_precompute_x3.hip: the hoisted conv of all 1024 channels in split-bf16 arithmetic (needs section WPX)
__attribute__((visibility("hidden")))
int launch_P_x3(void* stream, const float* feat_dev, const float* packed_dev, float* P_dev,
                int B, int H, int W, int r0, int r1, RowWin fw, RowWin pw);
// diinn_precompute_wino.hip: the fp32 hoisted conv of all 1024 channels as Winograd F(2x2,3x3) (needs section WPU)
__attribute__((visibility("hidden")))
int launch_P_wino(void* stream, const float* feat_dev, const float* packed_dev, float* P_dev,
                  int B, int H, int W, int r0, int r1, RowWin fw, RowWin pw, bool wpu_only_ok = false);
// diinn_bf16.hip: the bf16 decode of HR rows [p.y0, p.y1) (grid of the one-tile kernel: gx, gy, gz)
__attribute__((visibility("hidden")))
int launch_decode_bf16(void* stream, const DecodeParams& p, int gx, int gy, int gz, int sin_mode);
// diinn_conv_x3.hip: a split-bf16 3x3 layer reading planes or the trunk's split-format buffer (and optionally extending it);
// a block's 1x1 fusion layer from that buffer (diinn_rdn_forward_x3)
extern "C" __attribute__((visibility("hidden")))
int diinn_conv3x3_x3_split(void* stream, const float* in_dev, long long in_bs, float* xs_dev, long long xs_bs16, int xs_out_g8,
                           int Cin, const float* wx_dev, const float* bias_dev, float* out_dev, long long out_bs, int relu,
                           int B, int H, int W);
extern "C" __attribute__((visibility("hidden")))
int diinn_conv1x1_x3_split(void* stream, float* xs_dev, long long xs_bs16, int Cin, const float* wx_dev, const float* bias_dev,
                           const float* res_dev, long long res_bs, float* o0_dev, long long o0_bs, float* o1_dev, long long o1_bs,
                           int B, int H, int W);
// diinn_bf16.hip: the split-bf16 decode (DIINN_COMPUTE_BF16X3) of HR rows [p.y0, p.y1)
__attribute__((visibility("hidden")))
int launch_decode_bf16x3(void* stream, const DecodeParams& p, int gx, int gy, int gz, int sin_mode);
