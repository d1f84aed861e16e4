// diinn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the DIINN implicit decoder
// and the launch half of the C ABI (include/diinn_hip.h).
//
// Reference path replaced: ImplicitDecoder.forward, mode 3
//   /root/reference/src/models/components/diinn.py:163-173 (+ :94-110, :132-139, :149-160)
//
// Kernels (DESIGN.md has the derivation, the roofline and the measurements of each):
//   precompute_P_kernel : per LR cell, P_i = Wx_i . unfold3x3(feat) + bK_i, i=0..3
//                         (implicit-im2col GEMM 576 -> 1024 on v_mfma_f32_32x32x2_f32,
//                         feature halo tile in LDS)
//   decode_kernel       : per HR pixel, the dual-branch MLP.  One wave owns 32 pixels
//                         and keeps their 256-channel activation in registers for the
//                         whole network: the accumulator layout of one layer IS the
//                         B-operand layout of the next (diinn_layout.h), so activations
//                         never touch LDS or HBM.  Weights stream from the packed image.
//   decode_bf16_kernel  : the same with bf16 operands in layers 1..3 (optional path).
//   axis_tables_kernel, sin_kernel : the device coordinate / sine code, exposed for tests.
// Compile-time hooks that never ship enabled: ABL_* (timing ablations, wrong results),
// DIINN_STAMPS (s_memtime stamps for tools/stamp_report.py).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (explicit fmaf where wanted)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diinn_hip.h"
#include "diinn_layout.h"

using namespace diinn;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// ---------------------------------------------------------------------------------
// decode kernel
// ---------------------------------------------------------------------------------
struct DecodeParams {
    const float* P;        // [B,H,W,1024]
    const float* Wt;       // packed image
    float* out;            // [B,3,Hu,Wu]
    int B, H, W, Hu, Wu, y0, y1;
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

constexpr int TILE_W = 8, TILE_H = 4;           // one wave: 8x4 HR pixels
// timing-ablation hooks (wrong results when defined; never in the shipped build)
#ifdef ABL_WSTREAM
#define ABL_STEP(x) ((x) & 3)
#else
#define ABL_STEP(x) (x)
#endif
#ifdef ABL_NOSIN
#define ABL_SIN(x) (x)
#else
#define ABL_SIN(x) dsin<SIN_MODE>(x)
#endif
#ifndef DECODE_RUN_LAYERS
#define DECODE_RUN_LAYERS 3                     // < 3 only in timing-ablation builds (wrong results)
#endif
#ifndef WSTREAM_AUX
#define WSTREAM_AUX 0                           // cache-policy bits of the weight-stream loads (sc0=1, nt=2, sc1=16)
#endif
#ifndef P_PREFETCH
#define P_PREFETCH 4
#endif
#ifndef DECODE_PREFETCH
#define DECODE_PREFETCH 4                       // weight ring depth, in steps of 8 MFMAs
#endif

// relu without the canonicalising v_max that fmaxf(x, 0) emits for an MFMA result
__device__ __forceinline__ float relu0(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));     // one VALU op (fmaxf adds a canonicalising v_max)
    return r;
}

// Weight-stream loads go through a buffer descriptor: address = SGPR descriptor base + SGPR byte
// offset (scalar unit) + one constant per-lane VGPR offset, so the stream costs no VALU address
// arithmetic.  That matters here: on gfx950 the fp32 MFMA shares its issue/datapath with the VALU
// (tools/ubench/mfma_rate.hip: every VALU op between MFMAs costs ~3.2 cycles of MFMA time).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld_piece(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, WSTREAM_AUX));
}
constexpr int PIECE_BYTES = (int)(WL_PIECE * sizeof(float));
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
__device__ __forceinline__ void st_act(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)voff, (int)soff, 0);
}
// descriptor of one tile (rows x 32 floats) of a tiled plane group
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float* group, long long tile, int rows) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(group + (size_t)tile * rows * PLANE_TILE), 0,
                                             rows * (int)PLANE_ROW_BYTES, 0x00020000);
}

template <int SIN_MODE, bool KPART = true, bool SAVE = false>
__global__ __launch_bounds__(256, 1) void decode_kernel(const DecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform (scalar offsets)
    const int h = lane >> 5, j = lane & 31;

    int x, y, b;
    const long long ptile = (long long)blockIdx.x * 4 + wave;   // SAVE: plane tile of this wave
    if constexpr (SAVE) {                                        // 32 consecutive flattened pixels
        const long long pix = ptile * PLANE_TILE + j;
        const long long pc = pix < p.npix ? pix : p.npix - 1;    // lanes past the end compute on the last pixel
        const int hw = p.Hu * p.Wu;
        b = (int)(pc / hw);
        const int rem = (int)(pc - (long long)b * hw);
        y = rem / p.Wu;
        x = pix < p.npix ? rem - y * p.Wu : p.Wu;                // ... and are marked invalid below
    } else {
        x = blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
        y = p.y0 + blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
        b = blockIdx.z;
    }
    const bool valid = (x < p.Wu) && (y < p.y1);
    // whole wave outside the band/image: nothing to do (wave-uniform, no barriers in this kernel)
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.y1 ? y : p.y1 - 1;

    int iy, ix;
    float relh, relw;
    axis_eval(p.ah, yc, iy, relh);
    axis_eval(p.aw, xc, ix, relw);

#ifdef DIINN_STAMPS
    const size_t stamp_base = ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
    if (p.stamps && lane == 0) {
        unsigned long long rt;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
        p.stamps[stamp_base + 7] = rt;
    }
#endif
    STAMP(0);
    const float* __restrict__ Wt = p.Wt;
    const float* __restrict__ Pc = p.P + (((size_t)b * p.H + iy) * p.W + ix) * PCH + 4 * h;

    // saved-activation planes (SAVE): one buffer descriptor per layer covering this wave's plane tile
    // (512 rows x 32 pixels); a lane's offset is its pixel inside the row of channel 4h, the channel
    // row is a compile-time scalar offset.  Lanes past the end carry an offset outside the
    // descriptor's range: the store is dropped.
    const long long act_tiles = (p.npix + PLANE_TILE - 1) / PLANE_TILE;
    const unsigned act_voff = (SAVE && valid) ? 4u * j + 4u * h * PLANE_ROW_BYTES : 0xFFFFFFF0u;
    auto act_rsrc = [&](int layer) {
        return tile_rsrc(p.acts + (size_t)layer * act_tiles * ACT_ROWS * PLANE_TILE, ptile, ACT_ROWS);
    };

    // ---- layer 0: q0 = relu(P_0[cell]) * sin(Q0 . (rel_h, rel_w, ratio) + bQ0)   (diinn.py:133-134)
    float q[128];
    {
        const float* __restrict__ Q0 = Wt + OFF_Q0 + 4 * h;
        const __amdgpu_buffer_rsrc_t ar0 = act_rsrc(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 pv = *(const f32x4*)(Pc + c0);
                const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                const f32x4 wr = *(const f32x4*)(Q0 + 2 * HID + c0);
                const f32x4 bq = *(const f32x4*)(Q0 + 3 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = __builtin_fmaf(wr[e], p.ratio, bq[e]);
                    a = __builtin_fmaf(ww[e], relw, a);
                    a = __builtin_fmaf(wh[e], relh, a);
                    const float kv = relu0(pv[e]);
                    q[16 * m + 4 * g + e] = kv * dsin<SIN_MODE>(a);
                    if constexpr (SAVE) {
                        st_act(ar0, act_voff, (unsigned)(c0 + e) * PLANE_ROW_BYTES, kv);
                        st_act(ar0, act_voff, (unsigned)(HID + c0 + e) * PLANE_ROW_BYTES, a);
                    }
                }
            }
        }
    }

    STAMP(1);
    // ---- layers 1..3: [k;s] = [Wq_i;Qw_i] . q + [P_i[cell]; bQ_i];  q = relu(k) * sin(s)   (diinn.py:135-137)
    // Software pipeline, spelled out in program order (the loops below are fully unrolled):
    //   * weight pieces are fetched PF steps (8 MFMAs = 512 cycles each) ahead into a register ring
    //     that is carried across tiles and layers (the packed image is contiguous in step order);
    //   * the accumulator seeds of tile m+1 (P_i[cell], bQ_i) are fetched during tile m;
    //   * the VALU epilogue (relu * sin) of tile m-1 is spread over the MFMA stream of tile m.
    constexpr int PF = DECODE_PREFETCH;
    static_assert(WL_KG % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);   // reads past the end return 0
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WL * sizeof(float));                      // byte offset; advances one layer per iteration
    f32x4 rk[PF], rq[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        if constexpr (KPART) rk[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        rq[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }
    f32x4 sk[4], sq[4];                                          // seeds of the next tile
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sk[g] = *(const f32x4*)(Pc + HID + 8 * g);
        sq[g] = *(const f32x4*)(Wt + OFF_BQ + 4 * h + 8 * g);
    }
#pragma unroll 1
    for (int layer = 0; layer < DECODE_RUN_LAYERS; ++layer) {
        const int nl = layer < 2 ? layer + 1 : 2;                // seeds of the next layer's tile 0 (clamped)
        const float* __restrict__ Pl = Pc + (layer + 1) * HID;
        const float* __restrict__ Bq = Wt + OFF_BQ + layer * HID + 4 * h;
        const float* __restrict__ Pn = Pc + (nl + 1) * HID;
        const float* __restrict__ Bn = Wt + OFF_BQ + nl * HID + 4 * h;
        float qn[128];
        f32x16 pk, ps;                                           // finished accumulators of the previous tile
        const __amdgpu_buffer_rsrc_t arl = act_rsrc(layer + 1);
        (void)arl;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x16 ak, as;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ak[4 * g + e] = sk[g][e];
                    as[4 * g + e] = sq[g][e];
                }
            }
#pragma unroll
            for (int kg = 0; kg < WL_KG; ++kg) {
                const int s = m * WL_KG + kg;
                const f32x4 wq = rq[s % PF];
                if constexpr (KPART) {
                    const f32x4 wk = rk[s % PF];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ak = MFMA32(wk[e], q[4 * kg + e], ak);
                        as = MFMA32(wq[e], q[4 * kg + e], as);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) as = MFMA32(wq[e], q[4 * kg + e], as);
                }
                // refill the ring slot just consumed with the piece PF steps ahead
                if constexpr (KPART) rk[s % PF] = ld_piece(wrs, lane_off, wp + (2 * ABL_STEP(s + PF) + 0) * PIECE_BYTES);
                rq[s % PF] = ld_piece(wrs, lane_off, wp + (2 * ABL_STEP(s + PF) + 1) * PIECE_BYTES);
                if (kg == 4) {                                    // seeds for the next tile
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        sk[g] = *(const f32x4*)((m < 7 ? Pl + 32 * (m + 1) : Pn) + 8 * g);
                        sq[g] = *(const f32x4*)((m < 7 ? Bq + 32 * (m + 1) : Bn) + 8 * g);
                    }
                }
                if (m > 0 && (kg & 1) == 0) {                     // one epilogue element of tile m-1 every 16 MFMAs
                    const int r = kg >> 1;
                    const float kv = relu0(pk[r]);
                    qn[16 * (m - 1) + r] = kv * ABL_SIN(ps[r]);
                    if constexpr (SAVE) {                        // register r of tile m-1 = channel 32(m-1) + (r&3) + 8(r>>2) + 4h
                        const unsigned so = (unsigned)(32 * (m - 1) + (r & 3) + 8 * (r >> 2)) * PLANE_ROW_BYTES;
                        st_act(arl, act_voff, so, kv);
                        st_act(arl, act_voff, so + HID * PLANE_ROW_BYTES, ps[r]);
                    }
                }
            }
            pk = ak;
            ps = as;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float kv = relu0(pk[r]);
            qn[16 * 7 + r] = kv * dsin<SIN_MODE>(ps[r]);
            if constexpr (SAVE) {
                const unsigned so = (unsigned)(32 * 7 + (r & 3) + 8 * (r >> 2)) * PLANE_ROW_BYTES;
                st_act(arl, act_voff, so, kv);
                st_act(arl, act_voff, so + HID * PLANE_ROW_BYTES, ps[r]);
            }
        }
#pragma unroll
        for (int i = 0; i < 128; ++i) q[i] = qn[i];
        wp += (int)(WL_LAYER * sizeof(float));
#ifdef DIINN_STAMPS
        if (layer == 0) STAMP(2); else if (layer == 1) STAMP(3); else STAMP(4);
#endif
    }

    // ---- head: out = L . q3 + bL   (diinn.py:138)
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    {
        const float* __restrict__ L = Wt + OFF_L + 4 * h;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 l0 = *(const f32x4*)(L + 0 * HID + c0);
                const f32x4 l1 = *(const f32x4*)(L + 1 * HID + c0);
                const f32x4 l2 = *(const f32x4*)(L + 2 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = q[16 * m + 4 * g + e];
                    o0 = __builtin_fmaf(l0[e], v, o0);
                    o1 = __builtin_fmaf(l1[e], v, o1);
                    o2 = __builtin_fmaf(l2[e], v, o2);
                }
            }
        }
    }
    o0 += __shfl_xor(o0, 32);
    o1 += __shfl_xor(o1, 32);
    o2 += __shfl_xor(o2, 32);
    if (valid && h == 0) {
        const size_t plane = (size_t)p.Hu * p.Wu;
        float* o = p.out + (size_t)b * 3 * plane + (size_t)y * p.Wu + x;
        o[0] = o0 + Wt[OFF_BL + 0];
        o[plane] = o1 + Wt[OFF_BL + 1];
        o[2 * plane] = o2 + Wt[OFF_BL + 2];
    }
    STAMP(5);
}



// ---------------------------------------------------------------------------------
// cell_chain_kernel (decoder modes 1 and 2, diinn.py:116-131): the modulation chain depends on the
// LR cell only: k_0 = relu(P_0), k_i = relu(K_i^k k_{i-1} + P_i), i = 1..3.  Same register-resident
// scheme as decode_kernel with LR cells in place of HR pixels and the modulation half of the
// stacked weights alone (part 0 of the WL image holds K.i[:, :256] in A-operand order already).
// k_i overwrites the P_i slot of the workspace; decode_kernel<SIN, KPART=false> then reads it as
// the multiplier of the synthesis branch.
// ---------------------------------------------------------------------------------
struct ChainParams {
    float* P;            // [B,H,W,1024], slots 1..3 updated in place for rows [r0,r1)
    const float* Wt;
    int B, H, W, r0, r1;
};

__global__ __launch_bounds__(256, 1) void cell_chain_kernel(const ChainParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int x = blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int y = p.r0 + blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.W) && (y < p.r1);
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.W ? x : p.W - 1;
    const int yc = y < p.r1 ? y : p.r1 - 1;
    float* __restrict__ Pc = p.P + (((size_t)b * p.H + yc) * p.W + xc) * PCH + 4 * h;

    float k[128];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const f32x4 pv = *(const f32x4*)(Pc + 8 * i);             // channels 32m + 8g + 4h .., i = 4m + g
#pragma unroll
        for (int e = 0; e < 4; ++e) k[4 * i + e] = relu0(pv[e]);
    }

    constexpr int PF = DECODE_PREFETCH;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WL * sizeof(float));
    f32x4 rk[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) rk[d] = ld_piece(wrs, lane_off, wp + (2 * d) * PIECE_BYTES);
#pragma unroll 1
    for (int layer = 0; layer < 3; ++layer) {
        float* __restrict__ Pl = Pc + (layer + 1) * HID;
        float kn[128];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x16 ak;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 pv = *(const f32x4*)(Pl + 32 * m + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) ak[4 * g + e] = pv[e];
            }
#pragma unroll
            for (int kg = 0; kg < WL_KG; ++kg) {
                const int s = m * WL_KG + kg;
                const f32x4 wk = rk[s % PF];
#pragma unroll
                for (int e = 0; e < 4; ++e) ak = MFMA32(wk[e], k[4 * kg + e], ak);
                rk[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF)) * PIECE_BYTES);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = relu0(ak[4 * g + e]);
                    kn[16 * m + 4 * g + e] = v[e];
                }
                if (valid) *(f32x4*)(Pl + 32 * m + 8 * g) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 128; ++i) k[i] = kn[i];
        wp += (int)(WL_LAYER * sizeof(float));
    }
}

// ---------------------------------------------------------------------------------
// liif_kernel -- the LIIF comparison decoder (reference liif.py:59-127) on the
// same machinery: per HR pixel, for each of the 4 ensemble shifts, the 580 -> 256 -> 256 -> 256 -> 256 -> 3
// ReLU MLP on [unfolded features of the shifted nearest cell ; rel_coord ; rel_cell], blended by the
// diagonally opposite areas.  The first layer is hoisted like DIINN's: its 576 feature columns are a
// 3x3 conv per LR cell (precompute_P_kernel, first 256 channels of P, bias folded in), the 4 coordinate
// columns are 4 FMAs per channel here.  Layers 2..4 are the register-resident MFMA chain of
// decode_kernel with a ReLU epilogue; the weights travel in the same packed image (pack_liif in
// decoder.py maps imnet.layers.{2,4,6} to the synthesis slots of WL, the 4 coordinate columns to the
// Q0 table, the head to L).
// ---------------------------------------------------------------------------------
struct LiifParams {
    const float* P;        // [B,H,W,1024], channels 0..255 = first-layer pre-activation of the cell
    const float* Wt;
    float* out;            // [B,3,Hu,Wu]
    int B, H, W, Hu, Wu;
    LiifAxis ah, aw;
};

__global__ __launch_bounds__(256, 1) void liif_kernel(const LiifParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int x = blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int y = blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.Wu) && (y < p.Hu);
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.Hu ? y : p.Hu - 1;

    int iy[2], ix[2];
    float rh[2], rw[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        liif_axis_eval(p.ah, yc, v, iy[v], rh[v]);
        liif_axis_eval(p.aw, xc, v, ix[v], rw[v]);
    }
    // areas in the reference's member order (vx outer, vy inner), liif.py:117-118
    float area[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) area[v] = __builtin_fabsf(rh[v >> 1] * rw[v & 1]) + 1e-9f;
    const float tot = ((area[0] + area[1]) + area[2]) + area[3];

    const float* __restrict__ Wt = p.Wt;
    constexpr int PF = DECODE_PREFETCH;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;

#pragma unroll 1
    for (int v = 0; v < 4; ++v) {
        const int vh = v >> 1, vw = v & 1;
        const float relh = rh[vh], relw = rw[vw];
        const float* __restrict__ Pc = p.P + (((size_t)b * p.H + iy[vh]) * p.W + ix[vw]) * PCH + 4 * h;
        // ---- layer 1: relu(P[cell] + W1[:, 576:580] . (rel_h, rel_w, cell_h, cell_w))
        float q[128];
        {
            const float* __restrict__ Q0 = Wt + OFF_Q0 + 4 * h;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * m + 8 * g;
                    const f32x4 pv = *(const f32x4*)(Pc + c0);
                    const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                    const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                    const f32x4 wch = *(const f32x4*)(Q0 + 2 * HID + c0);
                    const f32x4 wcw = *(const f32x4*)(Q0 + 3 * HID + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = __builtin_fmaf(wh[e], relh, pv[e]);
                        a = __builtin_fmaf(ww[e], relw, a);
                        a = __builtin_fmaf(wch[e], p.ah.rel_cell, a);
                        a = __builtin_fmaf(wcw[e], p.aw.rel_cell, a);
                        q[16 * m + 4 * g + e] = relu0(a);
                    }
                }
            }
        }
        // ---- layers 2..4: q = relu(W q + b), weights in the synthesis slots (part 1) of WL
        int wp = (int)(OFF_WL * sizeof(float));
        f32x4 rq[PF];
#pragma unroll
        for (int d = 0; d < PF; ++d) rq[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
        f32x4 sq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) sq[g] = *(const f32x4*)(Wt + OFF_BQ + 4 * h + 8 * g);
#pragma unroll 1
        for (int layer = 0; layer < 3; ++layer) {
            const int nl = layer < 2 ? layer + 1 : 2;
            const float* __restrict__ Bq = Wt + OFF_BQ + layer * HID + 4 * h;
            const float* __restrict__ Bn = Wt + OFF_BQ + nl * HID + 4 * h;
            float qn[128];
            f32x16 ps;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                f32x16 as;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) as[4 * g + e] = sq[g][e];
#pragma unroll
                for (int kg = 0; kg < WL_KG; ++kg) {
                    const int s = m * WL_KG + kg;
                    const f32x4 wq = rq[s % PF];
#pragma unroll
                    for (int e = 0; e < 4; ++e) as = MFMA32(wq[e], q[4 * kg + e], as);
                    rq[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 1) * PIECE_BYTES);
                    if (kg == 4) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            sq[g] = *(const f32x4*)((m < 7 ? Bq + 32 * (m + 1) : Bn) + 8 * g);
                    }
                    if (m > 0 && (kg & 1) == 0) {
                        const int r = kg >> 1;
                        qn[16 * (m - 1) + r] = relu0(ps[r]);
                    }
                }
                ps = as;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) qn[16 * 7 + r] = relu0(ps[r]);
#pragma unroll
            for (int i = 0; i < 128; ++i) q[i] = qn[i];
            wp += (int)(WL_LAYER * sizeof(float));
        }
        // ---- head and the ensemble weight: member v is weighted by the area of the opposite member
        float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
        {
            const float* __restrict__ L = Wt + OFF_L + 4 * h;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * m + 8 * g;
                    const f32x4 l0 = *(const f32x4*)(L + 0 * HID + c0);
                    const f32x4 l1 = *(const f32x4*)(L + 1 * HID + c0);
                    const f32x4 l2 = *(const f32x4*)(L + 2 * HID + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float qv = q[16 * m + 4 * g + e];
                        m0 = __builtin_fmaf(l0[e], qv, m0);
                        m1 = __builtin_fmaf(l1[e], qv, m1);
                        m2 = __builtin_fmaf(l2[e], qv, m2);
                    }
                }
            }
        }
        m0 += __shfl_xor(m0, 32);
        m1 += __shfl_xor(m1, 32);
        m2 += __shfl_xor(m2, 32);
        float aw = area[0];                                      // area[3 - v] without dynamic register indexing
        aw = v == 0 ? area[3] : aw;
        aw = v == 1 ? area[2] : aw;
        aw = v == 2 ? area[1] : aw;
        const float wgt = aw / tot;
        o0 = __builtin_fmaf(m0 + Wt[OFF_BL + 0], wgt, o0);
        o1 = __builtin_fmaf(m1 + Wt[OFF_BL + 1], wgt, o1);
        o2 = __builtin_fmaf(m2 + Wt[OFF_BL + 2], wgt, o2);
    }
    if (valid && h == 0) {
        const size_t plane = (size_t)p.Hu * p.Wu;
        float* o = p.out + (size_t)b * 3 * plane + (size_t)y * p.Wu + x;
        o[0] = o0;
        o[plane] = o1;
        o[2 * plane] = o2;
    }
}

// ---------------------------------------------------------------------------------
// MetaSR comparison decoder (reference metasr.py:70-104): per HR pixel the
// meta-network 3 -> 256 -> 1728 predicts a [576 x 3] filter from (rel_h, rel_w, r_rev) and applies it to
// the unfolded 3x3 features of the pixel's cell.
//   unfold_cells_kernel : U[cell][k = c*9 + ky*3 + kx] = feat[c][cy+ky-1][cx+kx-1] (zero outside): the rows
//                         the final contraction gathers, contiguous per cell (2,304 B).
//   metasr_kernel       : one wave per 32 pixels; hidden = relu(W1 . inp + b1) in registers (128), then the
//                         1728 x 256 second layer as 54 MFMA tiles whose rows are regrouped by RGB
//                         component (diinn_layout.h): the epilogue of a tile is 16 FMAs against the
//                         tile's 32 feature values, so the 1728 predicted weights never leave registers.
// ---------------------------------------------------------------------------------
struct UnfoldParams {
    const float* feat;     // [B,64,H,W]
    float* U;              // [B,H,W,576]
    int B, H, W;
};

__global__ __launch_bounds__(192) void unfold_cells_kernel(const UnfoldParams p) {
    const int cx = blockIdx.x, cy = blockIdx.y, b = blockIdx.z;
    float* __restrict__ dst = p.U + (((size_t)b * p.H + cy) * p.W + cx) * MS_K;
    for (int k = threadIdx.x; k < MS_K; k += 192) {
        const int c = k / 9, t = k - 9 * c;
        const int yy = cy + t / 3 - 1, xx = cx + t % 3 - 1;
        float v = 0.0f;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = p.feat[(((size_t)b * C_IN + c) * p.H + yy) * p.W + xx];
        dst[k] = v;
    }
}

struct MetaParams {
    const float* U;        // [B,H,W,576]
    const float* Wt;       // MetaSR packed image
    float* out;            // [B,3,Hu,Wu]
    int B, H, W, Hu, Wu;
    MetaAxis ah, aw;
};

__global__ __launch_bounds__(256, 1) void metasr_kernel(const MetaParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int x = blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int y = blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.Wu) && (y < p.Hu);
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.Hu ? y : p.Hu - 1;
    int iy, ix;
    float relh, relw;
    meta_axis_eval(p.ah, yc, iy, relh);
    meta_axis_eval(p.aw, xc, ix, relw);
    const float* __restrict__ Wt = p.Wt;
    const float* __restrict__ Uc = p.U + (((size_t)b * p.H + iy) * p.W + ix) * MS_K + 4 * h;

    // ---- hidden = relu(W1 . (rel_h, rel_w, r_rev) + b1)   (imnet.layers.0, metasr.py:98-101)
    float q[128];
    {
        const float* __restrict__ Q0 = Wt + MS_OFF_Q0 + 4 * h;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                const f32x4 wr = *(const f32x4*)(Q0 + 2 * HID + c0);
                const f32x4 bq = *(const f32x4*)(Q0 + 3 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = __builtin_fmaf(wr[e], p.ah.r_rev, bq[e]);
                    a = __builtin_fmaf(ww[e], relw, a);
                    a = __builtin_fmaf(wh[e], relh, a);
                    q[16 * m + 4 * g + e] = relu0(a);
                }
            }
        }
    }

    // ---- second layer (1728 x 256) and the contraction with the cell's 576 unfolded features
    constexpr int PF = DECODE_PREFETCH;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(MS_PACKED_FLOATS * sizeof(float)), 0x00020000);   // reads past the end return 0
    const int lane_off = lane * 16;
    int wp = (int)(MS_OFF_W2 * sizeof(float));
    f32x4 rq[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) rq[d] = ld_piece(wrs, lane_off, wp + d * PIECE_BYTES);
    f32x4 sq[4], uv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) sq[g] = *(const f32x4*)(Wt + MS_OFF_B2 + 4 * h + 8 * g);
    const size_t plane = (size_t)p.Hu * p.Wu;
    float* __restrict__ o = p.out + (size_t)b * 3 * plane + (size_t)y * p.Wu + x;

#pragma unroll 1
    for (int comp = 0; comp < 3; ++comp) {
        const int nc = comp < 2 ? comp + 1 : 2;                       // seeds of the next component's first tile (clamped)
        const float* __restrict__ B2 = Wt + MS_OFF_B2 + comp * MS_K + 4 * h;
        const float* __restrict__ Bn = Wt + MS_OFF_B2 + nc * MS_K + 4 * h;
        float acc = 0.0f;
        f32x16 ps;
        f32x4 pu[4];
#pragma unroll
        for (int mm = 0; mm < MS_MM; ++mm) {
            f32x16 as;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) as[4 * g + e] = sq[g][e];
#pragma unroll
            for (int kg = 0; kg < WL_KG; ++kg) {
                const int s = mm * WL_KG + kg;
                const f32x4 wq = rq[s % PF];
#pragma unroll
                for (int e = 0; e < 4; ++e) as = MFMA32(wq[e], q[4 * kg + e], as);
                rq[s % PF] = ld_piece(wrs, lane_off, wp + (s + PF) * PIECE_BYTES);
                if (kg == 4) {                                        // next tile's bias seeds, this tile's feature values
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        sq[g] = *(const f32x4*)((mm < MS_MM - 1 ? B2 + 32 * (mm + 1) : Bn) + 8 * g);
                        uv[g] = *(const f32x4*)(Uc + 32 * mm + 8 * g);
                    }
                }
                if (mm > 0 && (kg & 1) == 0) {                        // one epilogue element of tile mm-1 every 8 MFMAs
                    const int r = kg >> 1;
                    acc = __builtin_fmaf(ps[r], pu[r >> 2][r & 3], acc);
                }
            }
            ps = as;
#pragma unroll
            for (int g = 0; g < 4; ++g) pu[g] = uv[g];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = __builtin_fmaf(ps[r], pu[r >> 2][r & 3], acc);
        acc += __shfl_xor(acc, 32);
        if (valid && h == 0) o[comp * plane] = acc;
        wp += (int)(MS_MM * WL_KG * PIECE_BYTES);
    }
}

// ---------------------------------------------------------------------------------
// backward pass of the per-pixel layers (training; reference: autograd through step(), diinn.py:132-139)
//
// With q_i = k_i * sin(s_i) and the planes k_i, s_i saved by decode_kernel<SAVE>:
//     g_a,i = g_q,i * sin(s_i) * [k_i > 0]      (gradient at the modulation pre-activation)
//     g_s,i = g_q,i * k_i * cos(s_i)            (gradient at the sine argument)
//     g_q,i-1 = Wq_i^T g_a,i + Qw_i^T g_s,i     (stacked [256 x 512] GEMM per pixel)
// bwd_head_kernel  : g_q,3 = L^T g_out, gates of layer 3 (elementwise, HBM-bound).
// bwd_layer_kernel : one launch per layer i = 3, 2, 1.  A wave owns one plane tile (32 pixels), loads
//                    their 512 gate gradients G_i = (g_a,i ; g_s,i) into registers as the MFMA B operand
//                    (the rows are read in accumulator order, so no shuffle is needed), streams the
//                    transposed weights (WLT section) exactly like the forward kernel streams WL, and
//                    its epilogue applies the gates of layer i-1 and writes G_{i-1} and q_{i-1}.
// plane_gemm_kernel / plane_rowdot_kernel : the parameter gradients, GEMMs over the pixel axis of the
//                    planes written here (dW_i = G_i q_{i-1}^T ...).
// All planes are tiled (see PLANE_TILE above): acts, G [4][ntiles][512][32]; Q [4][ntiles][256][32].
// ---------------------------------------------------------------------------------
struct BwdParams {
    const float* Wt;         // packed image
    const float* acts;       // k_i (rows 0..255), s_i (rows 256..511)
    const float* gout;       // [3][npix] plain planes: d loss / d out
    float* G;                // g_a,i (rows 0..255), g_s,i (rows 256..511)
    float* Q;                // q_i
    long long npix, ntiles;
    int layer;               // bwd_layer_kernel: consumes G_layer, produces G_{layer-1}, Q_{layer-1}
};

__device__ __forceinline__ void dsincos(float x, float& sn, float& cs) {
    constexpr float C_HI = 0.15915494309189533577f;
    constexpr float C_LO = 6.4206383650924e-09f;
    const float k = __builtin_rintf(x * C_HI);
    float r = __builtin_fmaf(x, C_HI, -k);
    r = __builtin_fmaf(x, C_LO, r);
    sn = __builtin_amdgcn_sinf(r);
    cs = __builtin_amdgcn_cosf(r);
}

__device__ __forceinline__ float ld_act(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}

__global__ __launch_bounds__(256) void bwd_head_kernel(const BwdParams p) {
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= p.npix) return;
    const size_t np = (size_t)p.npix;
    const float g0 = p.gout[pix], g1 = p.gout[np + pix], g2 = p.gout[2 * np + pix];
    const float* __restrict__ L = p.Wt + OFF_L;
    const size_t tile = (size_t)(pix >> 5), lane = (size_t)(pix & 31);
    const size_t a0 = ((size_t)3 * p.ntiles + tile) * ACT_ROWS * PLANE_TILE + lane;   // layer 3 tile, row 0
    const size_t q0 = ((size_t)3 * p.ntiles + tile) * HID * PLANE_TILE + lane;
    const int c0 = blockIdx.y * 16;
#pragma unroll 4
    for (int c = c0; c < c0 + 16; ++c) {
        float g = L[c] * g0;
        g = __builtin_fmaf(L[HID + c], g1, g);
        g = __builtin_fmaf(L[2 * HID + c], g2, g);
        const float kv = p.acts[a0 + (size_t)c * PLANE_TILE];
        const float sv = p.acts[a0 + (size_t)(HID + c) * PLANE_TILE];
        float sn, cs;
        dsincos(sv, sn, cs);
        p.G[a0 + (size_t)c * PLANE_TILE] = kv > 0.0f ? g * sn : 0.0f;
        p.G[a0 + (size_t)(HID + c) * PLANE_TILE] = g * kv * cs;
        p.Q[q0 + (size_t)c * PLANE_TILE] = kv * sn;
    }
}

__global__ __launch_bounds__(256, 1) void bwd_layer_kernel(const BwdParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= p.ntiles) return;                                // wave-uniform
    const bool valid = tile * PLANE_TILE + j < p.npix;

    const int li = p.layer;                                     // 1..3
    const size_t agroup = (size_t)p.ntiles * ACT_ROWS * PLANE_TILE;     // floats per layer of acts / G
    const size_t qgroup = (size_t)p.ntiles * HID * PLANE_TILE;
    // lanes past the end: offset outside the descriptor, loads return 0 and stores are dropped
    const unsigned voff = valid ? 4u * j + 4u * h * PLANE_ROW_BYTES : 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t inG = tile_rsrc(p.G + (size_t)li * agroup, tile, ACT_ROWS);
    const __amdgpu_buffer_rsrc_t act = tile_rsrc(p.acts + (size_t)(li - 1) * agroup, tile, ACT_ROWS);
    const __amdgpu_buffer_rsrc_t outG = tile_rsrc(p.G + (size_t)(li - 1) * agroup, tile, ACT_ROWS);
    const __amdgpu_buffer_rsrc_t outQ = tile_rsrc(p.Q + (size_t)(li - 1) * qgroup, tile, HID);

    // B operand: register kk = 16m + r of lane-half h holds channel chan_of(kk, h) of this lane's pixel.
    // Only the first BLD k-groups are fetched up front; the rest stream in BLD groups ahead of the
    // MFMAs of the first output tile (which walks all 32 k-groups), so the 64 KiB a wave reads
    // hide behind its own arithmetic instead of in front of it.
    constexpr int BLD = 8;
    float ga[128], gs[128];
    auto load_group = [&](int kg) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = 4 * kg + e;
            const unsigned so = (unsigned)(32 * (kk >> 4) + (kk & 3) + 8 * ((kk & 15) >> 2)) * PLANE_ROW_BYTES;
            ga[kk] = ld_act(inG, voff, so);
            gs[kk] = ld_act(inG, voff, so + HID * PLANE_ROW_BYTES);
        }
    };
#pragma unroll
    for (int kg = 0; kg < BLD; ++kg) load_group(kg);

    constexpr int PF = DECODE_PREFETCH;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);   // reads past the end return 0
    const int lane_off = lane * 16;
    const int wp = (int)((OFF_WLT + (size_t)(li - 1) * WL_LAYER) * sizeof(float));
    f32x4 rk[PF], rq[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        rk[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        rq[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }

    // gates of layer li-1 for one finished element: g = d loss / d q_{li-1}[channel, pixel]
    auto gate_store = [&](int mt, int r, float g, float kv, float sv) {
        const unsigned so = (unsigned)(32 * mt + (r & 3) + 8 * (r >> 2)) * PLANE_ROW_BYTES;
        float sn, cs;
        dsincos(sv, sn, cs);
        st_act(outG, voff, so, kv > 0.0f ? g * sn : 0.0f);
        st_act(outG, voff, so + HID * PLANE_ROW_BYTES, g * kv * cs);
        st_act(outQ, voff, so, kv * sn);
    };

    f32x16 pg;                                                   // finished g_q tile (sum of the two accumulators)
    float kt[16], st[16];                                        // saved k, s of the tile being finished
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 ak, as;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[r] = 0.0f; as[r] = 0.0f; }
#pragma unroll
        for (int kg = 0; kg < WL_KG; ++kg) {
            const int s = m * WL_KG + kg;
            const f32x4 wk = rk[s % PF];
            const f32x4 wq = rq[s % PF];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ak = MFMA32(wk[e], ga[4 * kg + e], ak);
                as = MFMA32(wq[e], gs[4 * kg + e], as);
            }
            rk[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 0) * PIECE_BYTES);
            rq[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 1) * PIECE_BYTES);
            if (m == 0 && kg + BLD < WL_KG) load_group(kg + BLD); // rest of the B operand, BLD groups ahead
            if (m > 0 && kg == 0) {                               // saved planes of tile m-1, used from kg = 8 on
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned so = (unsigned)(32 * (m - 1) + (r & 3) + 8 * (r >> 2)) * PLANE_ROW_BYTES;
                    kt[r] = ld_act(act, voff, so);
                    st[r] = ld_act(act, voff, so + HID * PLANE_ROW_BYTES);
                }
            }
            if (m > 0 && kg >= 8 && kg < 24) {                    // one epilogue element of tile m-1 every 8 MFMAs
                const int r = kg - 8;
                gate_store(m - 1, r, pg[r], kt[r], st[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) pg[r] = ak[r] + as[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned so = (unsigned)(32 * 7 + (r & 3) + 8 * (r >> 2)) * PLANE_ROW_BYTES;
        kt[r] = ld_act(act, voff, so);
        st[r] = ld_act(act, voff, so + HID * PLANE_ROW_BYTES);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) gate_store(7, r, pg[r], kt[r], st[r]);
}

// ---------------------------------------------------------------------------------
// cell_sum_kernel (training backward): dP_i[b, ch, cy, cx] = sum of g_a,i over the HR pixels whose
// nearest LR cell is (cy, cx) -- the adjoint of the nearest-exact replication (diinn.py:168).  The
// index tables are monotone, so a cell's pixels are the rectangle [seg_h[cy], seg_h[cy+1]) x
// [seg_w[cx], seg_w[cx+1]).  One thread per output element, cx fastest: neighbouring lanes read
// neighbouring column segments of the same HR rows.  Fixed summation order (no atomics).  Output is
// NCHW [B][1024][H][W], channel = 256 i + ch: the layout the 3x3 convolution's weight/input
// gradients are taken in.  HBM-bound (reads the g_a rows of G once).
// ---------------------------------------------------------------------------------
struct CellSumParams {
    const float* G;          // tiled [4][ntiles][512][32]; rows 0..255 (g_a) are summed
    float* dP;               // [B][1024][H][W]
    const int* seg_h;        // [H+1] first HR row of every LR row (seg_h[H] = Hu)
    const int* seg_w;        // [W+1]
    int B, H, W, Hu, Wu;
    long long ntiles;
};

__global__ __launch_bounds__(256) void cell_sum_kernel(const CellSumParams p) {
    const int cx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int hb = (p.H + 3) / 4;
    const int b = blockIdx.y / hb;
    const int cy = (blockIdx.y - b * hb) * 4 + (threadIdx.x >> 6);
    if (cx >= p.W || cy >= p.H) return;
    const int plane = blockIdx.z;                                    // 256 i + ch
    const float* __restrict__ src = p.G + ((size_t)(plane >> 8) * p.ntiles * ACT_ROWS + (plane & 255)) * PLANE_TILE;
    const int y0 = p.seg_h[cy], y1 = p.seg_h[cy + 1];
    const int x0 = p.seg_w[cx], x1 = p.seg_w[cx + 1];
    float acc = 0.0f;
    for (int y = y0; y < y1; ++y) {
        const long long rowpix = ((long long)b * p.Hu + y) * p.Wu;
        float r = 0.0f;
        for (int x = x0; x < x1; ++x) {
            const long long pix = rowpix + x;
            r += src[(size_t)(pix >> 5) * (ACT_ROWS * PLANE_TILE) + (size_t)(pix & 31)];
        }
        acc += r;
    }
    p.dP[(((size_t)b * PCH + plane) * p.H + cy) * p.W + cx] = acc;
}

// ---------------------------------------------------------------------------------
// plane_gemm_kernel (training backward, weight gradients): C[M x Nc] = A[M x npix] . B[Nc x npix]^T, A and B
// being rows [a_row0, a_row0+M) / [b_row0, b_row0+Nc) of tiled plane groups, i.e. a GEMM whose reduction
// axis is the pixel axis.  Split-K: workgroup (block, ks) reduces the plane tiles of chunk ks for a
// 128 x 256 output block and writes its partial product to part[ks]; the caller adds the ksplit
// partials (fixed order, no atomics).  4 waves = 2 (M) x 2 (N), wave tile 64 x 128 = 2 x 4 MFMA tiles
// (128 accumulator registers).  Operand fragments go global -> registers directly: with tiled planes a
// 32-row x 32-pixel MFMA panel is one contiguous 4 KiB block; lane (row = l&31, half = l>>5) reads
// 16 bytes of its row per load, four loads cover the row's whole 128-byte line.  The MFMA k-pair
// (pixel e, pixel 4+e) is the same for A and B, and the sum over pixels does not care about the order.
// Optional extra column Nc: row sums of A (bias gradients).
// ---------------------------------------------------------------------------------
struct PlaneGemmParams {
    const float* A;          // tiled group, a_rows rows per tile; rows [a_row0, a_row0 + M) are used
    const float* Bm;         // tiled group, b_rows rows per tile; rows [b_row0, b_row0 + Nc)
    float* part;             // [ksplit][M][ldc]
    long long npix;
    int a_rows, a_row0, b_rows, b_row0;
    int M, Nc, ldc, tiles_per_split, with_rowsum;
};

template <int NB>
__global__ __launch_bounds__(256, 1) void plane_gemm_kernel(const PlaneGemmParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    constexpr int WGN = 2 * NB * 32;                          // output columns per workgroup
    const int nblk = p.Nc / WGN;
    const int m0 = (blockIdx.x / nblk) * 128 + (wave & 1) * 64;
    const int n0 = (blockIdx.x % nblk) * WGN + (wave >> 1) * (NB * 32);
    const int ks = blockIdx.y;
    const long long ntiles = (p.npix + PLANE_TILE - 1) / PLANE_TILE;
    const long long t0 = (long long)ks * p.tiles_per_split;
    long long t1 = t0 + p.tiles_per_split;
    if (t1 > ntiles) t1 = ntiles;
    const int nt = t1 > t0 ? (int)(t1 - t0) : 0;                 // tiles this workgroup reduces

    // descriptors start at this split's first tile (offsets inside a split stay far below 4 GiB)
    const unsigned a_pitch = (unsigned)p.a_rows * PLANE_ROW_BYTES, b_pitch = (unsigned)p.b_rows * PLANE_ROW_BYTES;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + ((size_t)t0 * p.a_rows + p.a_row0 + m0) * PLANE_TILE), 0, (int)(nt * a_pitch), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.Bm + ((size_t)t0 * p.b_rows + p.b_row0 + n0) * PLANE_TILE), 0, (int)(nt * b_pitch), 0x00020000);
    const unsigned voff = (unsigned)j * PLANE_ROW_BYTES + 16u * h;
    constexpr unsigned MFMA_ROWS = 32u * PLANE_ROW_BYTES;        // one 32-row MFMA panel: 4 KiB

    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float rs[2] = {0.0f, 0.0f};

    f32x4 fa[2][2][4], fb[2][NB][4];                              // [buffer][panel][t]: 4 pixels each
    auto load = [&](int buf, int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
                fa[buf][a][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    ra, (int)voff, (int)(t * a_pitch + a * MFMA_ROWS + 32u * q), 0));
#pragma unroll
            for (int b = 0; b < NB; ++b)
                fb[buf][b][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rb, (int)voff, (int)(t * b_pitch + b * MFMA_ROWS + 32u * q), 0));
        }
    };
    auto compute = [&](int buf, int t) {
        const long long pix0 = (t0 + t) * PLANE_TILE;
        if (pix0 + PLANE_TILE > p.npix) {                        // ragged last tile: its padding was never written
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = pix0 + 8 * q + 4 * h + e < p.npix;
#pragma unroll
                    for (int a = 0; a < 2; ++a) fa[buf][a][q][e] = in ? fa[buf][a][q][e] : 0.0f;
#pragma unroll
                    for (int b = 0; b < NB; ++b) fb[buf][b][q][e] = in ? fb[buf][b][q][e] : 0.0f;
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[a][b] = MFMA32(fa[buf][a][q][e], fb[buf][b][q][e], acc[a][b]);
                    rs[a] += fa[buf][a][q][e];
                }
            }
    };

    if (nt > 0) {
        load(0, 0);
        for (int t = 0; t < nt; t += 2) {
            if (t + 1 < nt) load(1, t + 1);
            compute(0, t);
            if (t + 1 < nt) {
                if (t + 2 < nt) load(0, t + 2);
                compute(1, t + 1);
            }
        }
    }

    float* __restrict__ dst = p.part + (size_t)ks * p.M * p.ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                dst[(size_t)row * p.ldc + n0 + 32 * b + j] = acc[a][b][r];
            }
    if (p.with_rowsum && n0 == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float v = rs[a] + __shfl_xor(rs[a], 32);
            if (h == 0) dst[(size_t)(m0 + 32 * a + j) * p.ldc + p.Nc] = v;
        }
    }
}

// ---------------------------------------------------------------------------------
// plane_rowdot_kernel (training backward, the two skinny products): C[M x 4] = A[M x npix] . S[4 x npix]^T
// with A rows of a tiled group (M = 256 or 512) and S a tiled 4-row group.  Used for
//   layer 0: (g_a,0 ; g_s,0) . (rel_h, rel_w, ratio, 1)^T  -> dbK_0, dQ0, dbQ0          (diinn.py:133-134,165-167)
//   head   : q_3 . (g_out0, g_out1, g_out2, 0)^T            -> d last_layer.weight      (diinn.py:138)
// HBM-bound (reads A once); split over the tiles like plane_gemm_kernel, partials added by the caller.
// ---------------------------------------------------------------------------------
struct RowDotParams {
    const float* A;          // tiled, a_rows per tile, rows [0, M)
    const float* S;          // tiled [ntiles][4][32]
    float* part;             // [splits][M][4]
    long long npix;
    int a_rows, M, tiles_per_split;
};

__global__ __launch_bounds__(256) void plane_rowdot_kernel(const RowDotParams p) {
    const long long ntiles = (p.npix + PLANE_TILE - 1) / PLANE_TILE;
    const long long t0 = (long long)blockIdx.x * p.tiles_per_split;
    long long t1 = t0 + p.tiles_per_split;
    if (t1 > ntiles) t1 = ntiles;
    for (int row = threadIdx.x; row < p.M; row += 256) {
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
        for (long long t = t0; t < t1; ++t) {
            const f32x4* __restrict__ a = (const f32x4*)(p.A + ((size_t)t * p.a_rows + row) * PLANE_TILE);
            const f32x4* __restrict__ s = (const f32x4*)(p.S + (size_t)t * 4 * PLANE_TILE);
            const int left = (int)(p.npix - t * PLANE_TILE < PLANE_TILE ? p.npix - t * PLANE_TILE : PLANE_TILE);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 av = a[q];
                if (left < PLANE_TILE) {                          // ragged last tile: padding was never written
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = 4 * q + e < left ? av[e] : 0.0f;
                }
                const f32x4 s0 = s[q], s1 = s[8 + q], s2 = s[16 + q], s3 = s[24 + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    c0 = __builtin_fmaf(av[e], s0[e], c0);
                    c1 = __builtin_fmaf(av[e], s1[e], c1);
                    c2 = __builtin_fmaf(av[e], s2[e], c2);
                    c3 = __builtin_fmaf(av[e], s3[e], c3);
                }
            }
        }
        float* __restrict__ dst = p.part + ((size_t)blockIdx.x * p.M + row) * 4;
        dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = c3;
    }
}

// ---------------------------------------------------------------------------------
// decode kernel, bf16 operands (optional path, BASELINE config 5; tolerance restated in
// DESIGN.md): same structure as decode_kernel -- one wave owns 32 pixels and keeps their
// activation in registers -- but layers 1..3 run on v_mfma_f32_32x32x16_bf16: weights are bf16
// (WLB section), the activation is packed to bf16 straight from the epilogue (accumulator registers
// 8s..8s+7 of a tile ARE the B fragment of k-step 2m+s, see chan_of_bf16), accumulation, seeds
// (P, biases), sine, layer 0 and the RGB head stay fp32.  A first version: the weight stream
// (1 KiB per MFMA per wave, 8x the fp32 path's bytes per cycle) is L1-bandwidth-bound here.
// ---------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#ifndef DECODE_BF16_PREFETCH
#define DECODE_BF16_PREFETCH 8                  // ring depth in k-steps (2 pieces, 2 MFMAs each)
#endif

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_bf16_kernel(const DecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int x = blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int y = p.y0 + blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.Wu) && (y < p.y1);
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.y1 ? y : p.y1 - 1;
    int iy, ix;
    float relh, relw;
    axis_eval(p.ah, yc, iy, relh);
    axis_eval(p.aw, xc, ix, relw);
    const float* __restrict__ Wt = p.Wt;
    const float* __restrict__ Pc = p.P + (((size_t)b * p.H + iy) * p.W + ix) * PCH + 4 * h;

    // ---- layer 0 (fp32), packed to bf16 fragments: register r = 4g+e of tile m -> qb[2m + (r>>3)][r&7]
    bf16x8 qb[16];
    {
        const float* __restrict__ Q0 = Wt + OFF_Q0 + 4 * h;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 pv = *(const f32x4*)(Pc + c0);
                const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                const f32x4 wr = *(const f32x4*)(Q0 + 2 * HID + c0);
                const f32x4 bq = *(const f32x4*)(Q0 + 3 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = __builtin_fmaf(wr[e], p.ratio, bq[e]);
                    a = __builtin_fmaf(ww[e], relw, a);
                    a = __builtin_fmaf(wh[e], relh, a);
                    qb[2 * m + (g >> 1)][4 * (g & 1) + e] = (__bf16)(relu0(pv[e]) * dsin<SIN_MODE>(a));
                }
            }
        }
    }

    constexpr int PF = DECODE_BF16_PREFETCH;
    static_assert(16 % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WLB * sizeof(float));
    f32x4 rk[PF], rq[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        rk[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        rq[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }
    f32x4 sk[4], sq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sk[g] = *(const f32x4*)(Pc + HID + 8 * g);
        sq[g] = *(const f32x4*)(Wt + OFF_BQ + 4 * h + 8 * g);
    }
    float q3[128];                                               // fp32 copy of the last activation for the head
#pragma unroll 1
    for (int layer = 0; layer < 3; ++layer) {
        const int nl = layer < 2 ? layer + 1 : 2;
        const float* __restrict__ Pl = Pc + (layer + 1) * HID;
        const float* __restrict__ Bq = Wt + OFF_BQ + layer * HID + 4 * h;
        const float* __restrict__ Pn = Pc + (nl + 1) * HID;
        const float* __restrict__ Bn = Wt + OFF_BQ + nl * HID + 4 * h;
        bf16x8 qn[16];
        f32x16 pk, ps;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x16 ak, as;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ak[4 * g + e] = sk[g][e];
                    as[4 * g + e] = sq[g][e];
                }
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int s = m * 16 + ks;
                ak = MFMA_BF16(__builtin_bit_cast(bf16x8, rk[s % PF]), qb[ks], ak);
                as = MFMA_BF16(__builtin_bit_cast(bf16x8, rq[s % PF]), qb[ks], as);
                rk[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 0) * PIECE_BYTES);
                rq[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 1) * PIECE_BYTES);
                if (ks == 2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        sk[g] = *(const f32x4*)((m < 7 ? Pl + 32 * (m + 1) : Pn) + 8 * g);
                        sq[g] = *(const f32x4*)((m < 7 ? Bq + 32 * (m + 1) : Bn) + 8 * g);
                    }
                }
                if (m > 0) {                                      // one epilogue element of tile m-1 per k-step
                    const float v = relu0(pk[ks]) * dsin<SIN_MODE>(ps[ks]);
                    qn[2 * (m - 1) + (ks >> 3)][ks & 7] = (__bf16)v;
                    q3[16 * (m - 1) + ks] = v;
                }
            }
            pk = ak;
            ps = as;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = relu0(pk[r]) * dsin<SIN_MODE>(ps[r]);
            qn[14 + (r >> 3)][r & 7] = (__bf16)v;
            q3[16 * 7 + r] = v;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) qb[i] = qn[i];
        wp += (int)(WLB_LAYER * sizeof(float));
    }

    // ---- head (fp32) on the unrounded last activation
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    {
        const float* __restrict__ L = Wt + OFF_L + 4 * h;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 l0 = *(const f32x4*)(L + 0 * HID + c0);
                const f32x4 l1 = *(const f32x4*)(L + 1 * HID + c0);
                const f32x4 l2 = *(const f32x4*)(L + 2 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = q3[16 * m + 4 * g + e];
                    o0 = __builtin_fmaf(l0[e], v, o0);
                    o1 = __builtin_fmaf(l1[e], v, o1);
                    o2 = __builtin_fmaf(l2[e], v, o2);
                }
            }
        }
    }
    o0 += __shfl_xor(o0, 32);
    o1 += __shfl_xor(o1, 32);
    o2 += __shfl_xor(o2, 32);
    if (valid && h == 0) {
        const size_t plane = (size_t)p.Hu * p.Wu;
        float* o = p.out + (size_t)b * 3 * plane + (size_t)y * p.Wu + x;
        o[0] = o0 + Wt[OFF_BL + 0];
        o[plane] = o1 + Wt[OFF_BL + 1];
        o[2 * plane] = o2 + Wt[OFF_BL + 2];
    }
}

// ---------------------------------------------------------------------------------
// decode_bf16x2_kernel: the bf16 decode with TWO pixel tiles (2 x 32 pixels) per wave.  The single-tile
// kernel above is bound by its weight stream (1 KiB per 32-cycle MFMA per wave through a 64 B/clk L1);
// here every weight fragment feeds two MFMAs, which halves the bytes per MFMA.  To fit two tiles in the
// register file the next layer's activation is parked in a wave-private LDS slab (32 KiB per wave: each
// lane writes and later re-reads only its own 16-byte fragments, so no barrier is involved) and the RGB
// head is accumulated inside the last layer's epilogue instead of from an fp32 copy of the activation.
// A workgroup covers 16 x 16 HR pixels: wave w owns the 8x4 tile of decode_kernel's mapping and the one
// 8 rows below it.
// ---------------------------------------------------------------------------------
#ifndef DECODE_BF16X2_PREFETCH
#define DECODE_BF16X2_PREFETCH 4
#endif

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_bf16x2_kernel(const DecodeParams p) {
    __shared__ __attribute__((aligned(16))) bf16x8 park[4][2][16][64];     // [wave][tile][fragment][lane] = 128 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int x = blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int yb = p.y0 + blockIdx.y * (2 * TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    int y[2];
    bool valid[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        y[t] = yb + t * (TILE_H * WG_TILES_Y);
        valid[t] = (x < p.Wu) && (y[t] < p.y1);
    }
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid[0] || valid[1]) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    int ix;
    float relw;
    axis_eval(p.aw, xc, ix, relw);
    const float* __restrict__ Wt = p.Wt;
    const float* __restrict__ Pc[2];
    float relh[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int yc = y[t] < p.y1 ? y[t] : p.y1 - 1;
        int iy;
        axis_eval(p.ah, yc, iy, relh[t]);
        Pc[t] = p.P + (((size_t)b * p.H + iy) * p.W + ix) * PCH + 4 * h;
    }

    // ---- layer 0 (fp32), packed to bf16 fragments: register r = 4g+e of tile m -> qb[2m + (r>>3)][r&7]
    bf16x8 qb[2][16];
    {
        const float* __restrict__ Q0 = Wt + OFF_Q0 + 4 * h;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                const f32x4 wr = *(const f32x4*)(Q0 + 2 * HID + c0);
                const f32x4 bq = *(const f32x4*)(Q0 + 3 * HID + c0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 pv = *(const f32x4*)(Pc[t] + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = __builtin_fmaf(wr[e], p.ratio, bq[e]);
                        a = __builtin_fmaf(ww[e], relw, a);
                        a = __builtin_fmaf(wh[e], relh[t], a);
                        qb[t][2 * m + (g >> 1)][4 * (g & 1) + e] = (__bf16)(relu0(pv[e]) * dsin<SIN_MODE>(a));
                    }
                }
            }
        }
    }

    constexpr int PF = DECODE_BF16X2_PREFETCH;
    static_assert(16 % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WLB * sizeof(float));
    f32x4 rk[PF], rq[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        rk[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        rq[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }
    f32x4 sk[2][4], sq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sk[0][g] = *(const f32x4*)(Pc[0] + HID + 8 * g);
        sk[1][g] = *(const f32x4*)(Pc[1] + HID + 8 * g);
        sq[g] = *(const f32x4*)(Wt + OFF_BQ + 4 * h + 8 * g);
    }
    float o[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    bf16x8 (*mine)[16][64] = park[wave];

    // the layer loop is fully unrolled (a bf16 layer is 512 MFMAs): LAST is a compile-time constant per copy and
    // fuses the RGB head (diinn.py:138) into the epilogue, on the unrounded activation
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        const bool LAST = layer == 2;
        const int nl = layer < 2 ? layer + 1 : 2;
        const float* __restrict__ Bq = Wt + OFF_BQ + layer * HID + 4 * h;
        const float* __restrict__ Bn = Wt + OFF_BQ + nl * HID + 4 * h;
        const float* __restrict__ L = Wt + OFF_L + 4 * h;
        f32x16 pk[2], ps[2];
        bf16x8 frag[2];
        f32x4 l0[4], l1[4], l2[4];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x16 ak[2], as[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ak[t][4 * g + e] = sk[t][g][e];
                        as[t][4 * g + e] = sq[g][e];
                    }
            if (LAST && m > 0) {                                  // head rows of the tile being finished
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    l0[g] = *(const f32x4*)(L + 0 * HID + 32 * (m - 1) + 8 * g);
                    l1[g] = *(const f32x4*)(L + 1 * HID + 32 * (m - 1) + 8 * g);
                    l2[g] = *(const f32x4*)(L + 2 * HID + 32 * (m - 1) + 8 * g);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int s = m * 16 + ks;
                const bf16x8 wk = __builtin_bit_cast(bf16x8, rk[s % PF]);
                const bf16x8 wq = __builtin_bit_cast(bf16x8, rq[s % PF]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ak[t] = MFMA_BF16(wk, qb[t][ks], ak[t]);
                    as[t] = MFMA_BF16(wq, qb[t][ks], as[t]);
                }
                rk[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 0) * PIECE_BYTES);
                rq[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 1) * PIECE_BYTES);
                if (ks == 2) {
                    const int ln = layer + 1;                     // P slot of this layer; next layer's for the last tile
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            sk[t][g] = *(const f32x4*)(Pc[t] + (m < 7 ? ln * HID + 32 * (m + 1) : (nl + 1) * HID) + 8 * g);
                        sq[g] = *(const f32x4*)((m < 7 ? Bq + 32 * (m + 1) : Bn) + 8 * g);
                    }
                }
                if (m > 0) {                                      // one epilogue element of tile m-1 per k-step, both pixel tiles
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float v = relu0(pk[t][ks]) * dsin<SIN_MODE>(ps[t][ks]);
                        if (LAST) {
                            o[t][0] = __builtin_fmaf(l0[ks >> 2][ks & 3], v, o[t][0]);
                            o[t][1] = __builtin_fmaf(l1[ks >> 2][ks & 3], v, o[t][1]);
                            o[t][2] = __builtin_fmaf(l2[ks >> 2][ks & 3], v, o[t][2]);
                        } else {
                            frag[t][ks & 7] = (__bf16)v;
                            if ((ks & 7) == 7) mine[t][2 * (m - 1) + (ks >> 3)][lane] = frag[t];
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pk[t] = ak[t];
                ps[t] = as[t];
            }
        }
        if (LAST) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                l0[g] = *(const f32x4*)(L + 0 * HID + 32 * 7 + 8 * g);
                l1[g] = *(const f32x4*)(L + 1 * HID + 32 * 7 + 8 * g);
                l2[g] = *(const f32x4*)(L + 2 * HID + 32 * 7 + 8 * g);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = relu0(pk[t][r]) * dsin<SIN_MODE>(ps[t][r]);
                if (LAST) {
                    o[t][0] = __builtin_fmaf(l0[r >> 2][r & 3], v, o[t][0]);
                    o[t][1] = __builtin_fmaf(l1[r >> 2][r & 3], v, o[t][1]);
                    o[t][2] = __builtin_fmaf(l2[r >> 2][r & 3], v, o[t][2]);
                } else {
                    frag[t][r & 7] = (__bf16)v;
                    if ((r & 7) == 7) mine[t][14 + (r >> 3)][lane] = frag[t];
                }
            }
        }
        if (!LAST) {                                              // the parked activation becomes the next layer's B operand
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) qb[t][i] = mine[t][i][lane];
        }
        wp += (int)(WLB_LAYER * sizeof(float));
    }

#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float o0 = o[t][0], o1 = o[t][1], o2 = o[t][2];
        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        if (valid[t] && h == 0) {
            const size_t plane = (size_t)p.Hu * p.Wu;
            float* op = p.out + (size_t)b * 3 * plane + (size_t)y[t] * p.Wu + x;
            op[0] = o0 + Wt[OFF_BL + 0];
            op[plane] = o1 + Wt[OFF_BL + 1];
            op[2 * plane] = o2 + Wt[OFF_BL + 2];
        }
    }
}

// ---------------------------------------------------------------------------------
// P kernel: P[b,y,x, i*256+ch] = sum_{c,ky,kx} Wx_i[ch,c,ky,kx] * feat[b,c,y+ky-1,x+kx-1] + bK_i[ch]
// (zero padding; diinn.py:168 unfold + the feature columns of K[i], diinn.py:133,136)
// An implicit-im2col GEMM [cells x 576] . [576 x 1024] on v_mfma_f32_32x32x2_f32.
// Workgroup = 4 waves = a 4-row x 32-column block of LR cells; its 6 x 34 x 64 feature halo
// tile is staged once in LDS (zero padded), and each wave (one row of 32 cells) streams the
// whole packed WP image past it two M-tiles at a time: A operands from the packed image
// (identical for the 4 waves -> one L1 fill), B operands by ds_read_b32 at an immediate
// offset per (tap, channel).  Few registers -> 2 workgroups per CU hide each other's waits.
// ---------------------------------------------------------------------------------
struct PParams {
    const float* feat;   // [B,64,H,W]
    const float* Wt;
    float* P;            // [B,H,W,1024]
    int B, H, W, r0, r1;
    int msplit;          // the M-tile pairs are divided over `msplit` workgroups (blockIdx.z = b*msplit + part)
    int mp_total;        // M-tile pairs (64 channels each) to compute: 16 = all 1024 channels; LIIF needs the first 4
};

constexpr int PT_ROWS = 4, PT_COLS = 32;                 // cells per workgroup: 4 x 32
constexpr int PT_LR = PT_ROWS + 2, PT_LC = PT_COLS + 2;  // with the 3x3 halo: 6 x 34
constexpr int PT_CH = PT_LR * PT_LC;                     // 204 floats per channel
constexpr int PT_LDS_FLOATS = C_IN * PT_CH;              // 13,056 floats = 52,224 B

__global__ __launch_bounds__(256, 2) void precompute_P_kernel(const PParams p) {
    __shared__ __attribute__((aligned(16))) float tile[PT_LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int b = blockIdx.z / p.msplit;
    const int part = blockIdx.z - b * p.msplit;
    const int mp_count = p.mp_total / p.msplit, mp_begin = part * mp_count;
    const int x0 = blockIdx.x * PT_COLS;
    const int y0 = p.r0 + blockIdx.y * PT_ROWS;

    // stage feat[b, :, y0-1 .. y0+4, x0-1 .. x0+32] (zeros outside the map)
    const float* __restrict__ fb = p.feat + (size_t)b * C_IN * p.H * p.W;
    static_assert(PT_LDS_FLOATS % 256 == 0, "staging loop has a fixed trip count");
    // fixed trip count, unrolled in batches so that many loads are in flight (a rolled loop would pay
    // one memory latency per element)
#pragma unroll 17
    for (int it = 0; it < PT_LDS_FLOATS / 256; ++it) {
        const int idx = it * 256 + threadIdx.x;
        const int c = idx / PT_CH;
        const int rem = idx - c * PT_CH;
        const int ly = rem / PT_LC, lx = rem - ly * PT_LC;
        const int yy = y0 + ly - 1, xx = x0 + lx - 1;
        const bool ok = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
        // unconditional load from a clamped address, then select: a load under `ok ? .. : 0` compiles
        // to a branch and a vmcnt(0) per element (51 serial memory round trips per workgroup)
        const int yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
        const int xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
        const float v = fb[((size_t)c * p.H + yc) * p.W + xc];
        tile[idx] = ok ? v : 0.0f;
    }
    __syncthreads();

    const int x = x0 + j, y = y0 + wave;
    const bool store = (x < p.W) && (y < p.r1);
    // a wave whose row is past the band still runs (cheap at the band edge) -- no barrier follows,
    // so it may simply leave.
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(store) == 0ull))) return;

    // B operand of k-step kk = 32*t + cp (tap t = ky*3+kx, channel 2*cp + h):
    //   tile[(2cp + h) * PT_CH + (wave + ky) * PT_LC + j + kx]
    const int tb_off = h * PT_CH + wave * PT_LC + j;

    constexpr int PF = P_PREFETCH;
    static_assert(WP_KG % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);   // reads past the end return 0
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WP * sizeof(float)) + mp_begin * (WP_KG * 2 * PIECE_BYTES);   // advances one M-tile pair per iteration
    const float* __restrict__ Bk = p.Wt + OFF_BK + 4 * h;
    float* __restrict__ Pout = p.P + (((size_t)b * p.H + (y < p.H ? y : p.H - 1)) * p.W + (x < p.W ? x : p.W - 1)) * PCH + 4 * h;
    f32x4 r0v[PF], r1v[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        r0v[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        r1v[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }
#pragma unroll 1
    for (int mp = mp_begin; mp < mp_begin + mp_count; ++mp) {
        // the B operands do not depend on mp: hide the base from LICM, or all 288 LDS reads are
        // hoisted out of this loop and live (spilled) across it
        int off = tb_off;
        asm volatile("" : "+v"(off));
        const float* tbm = tile + off;          // still an LDS (ds_read) address
        f32x16 a0, a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 s0 = *(const f32x4*)(Bk + 64 * mp + 8 * g);
            const f32x4 s1 = *(const f32x4*)(Bk + 64 * mp + 32 + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[4 * g + e] = s0[e];
                a1[4 * g + e] = s1[e];
            }
        }
#pragma unroll
        for (int kg = 0; kg < WP_KG; ++kg) {
            const f32x4 u0 = r0v[kg % PF], u1 = r1v[kg % PF];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = 4 * kg + e;
                const int t = kk >> 5, cp = kk & 31;
                const float bv = tbm[(2 * cp) * PT_CH + (t / 3) * PT_LC + (t % 3)];
                a0 = MFMA32(u0[e], bv, a0);
                a1 = MFMA32(u1[e], bv, a1);
            }
            r0v[kg % PF] = ld_piece(wrs, lane_off, wp + (2 * (kg + PF) + 0) * PIECE_BYTES);
            r1v[kg % PF] = ld_piece(wrs, lane_off, wp + (2 * (kg + PF) + 1) * PIECE_BYTES);
        }
        if (store) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v0, v1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = a0[4 * g + e];
                    v1[e] = a1[4 * g + e];
                }
                *(f32x4*)(Pout + 64 * mp + 8 * g) = v0;
                *(f32x4*)(Pout + 64 * mp + 32 + 8 * g) = v1;
            }
        }
        wp += WP_KG * 2 * PIECE_BYTES;
    }
}

// ---------------------------------------------------------------------------------
// precompute_P_bf16_kernel (DIINN_COMPUTE_BF16_FULL): the hoisted 3x3 conv on v_mfma_f32_32x32x16_bf16.
// Same tiling as precompute_P_kernel (4 x 32 cells per workgroup, a wave per cell row, two M-tiles
// advancing together), with the feature halo tile converted to bf16 while it is staged and laid out
// channel-innermost in LDS: [6 x 34 pixels][64 channels + 8 pad] -> the B fragment of a k-step (16
// channels of one tap, 8 per lane-half) is one ds_read_b128, and the 144-byte pixel pitch spreads the
// 32 lanes of a row over all banks.  Accumulation, bias seeds and the stored P stay fp32.
// Bound: the bf16 weight stream through L1 (1 KiB per MFMA per wave) and the 4 KiB/cell store of P.
// ---------------------------------------------------------------------------------
constexpr int PB_PITCH = C_IN + 8;                        // bf16 elements per staged pixel (144 bytes)
constexpr int PB_LDS = PT_CH * PB_PITCH;                  // 14,688 bf16 = 29,376 B
constexpr int PB_ITEMS = PT_CH * (C_IN / 2);              // staged as channel pairs: 6,528 32-bit items

__global__ __launch_bounds__(256, 2) void precompute_P_bf16_kernel(const PParams p) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[PB_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int b = blockIdx.z / p.msplit;
    const int part = blockIdx.z - b * p.msplit;
    const int mp_count = p.mp_total / p.msplit, mp_begin = part * mp_count;
    const int x0 = blockIdx.x * PT_COLS;
    const int y0 = p.r0 + blockIdx.y * PT_ROWS;

    const float* __restrict__ fb = p.feat + (size_t)b * C_IN * p.H * p.W;
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll 13
    for (int it = 0; it < (PB_ITEMS + 255) / 256; ++it) {
        const int idx = it * 256 + threadIdx.x;                   // (channel pair, pixel), pixel fastest
        const int cp = idx / PT_CH;
        const int pix = idx - cp * PT_CH;
        const int ly = pix / PT_LC, lx = pix - ly * PT_LC;
        const int yy = y0 + ly - 1, xx = x0 + lx - 1;
        const bool ok = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W) && (idx < PB_ITEMS);
        const int yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
        const int xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
        const int cc = cp < C_IN / 2 ? cp : C_IN / 2 - 1;         // last iteration runs past the item count
        const float v0 = fb[((size_t)(2 * cc) * p.H + yc) * p.W + xc];
        const float v1 = fb[((size_t)(2 * cc + 1) * p.H + yc) * p.W + xc];
        bf16x2 pk;
        pk[0] = (__bf16)(ok ? v0 : 0.0f);
        pk[1] = (__bf16)(ok ? v1 : 0.0f);
        if (idx < PB_ITEMS) *(bf16x2*)(tile + pix * PB_PITCH + 2 * cp) = pk;
    }
    __syncthreads();

    const int x = x0 + j, y = y0 + wave;
    const bool store = (x < p.W) && (y < p.r1);
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(store) == 0ull))) return;

    // B fragment of k-step ks = 4*tap + cg: tile[((wave + ky) * 34 + j + kx) * 72 + 16cg + 8h .. +7]
    const int tb_off = (wave * PT_LC + j) * PB_PITCH + 8 * h;

    constexpr int PF = P_PREFETCH;
    static_assert(WPB_KS % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);   // reads past the end return 0
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WPB * sizeof(float)) + mp_begin * (WPB_KS * 2 * PIECE_BYTES);
    const float* __restrict__ Bk = p.Wt + OFF_BK + 4 * h;
    float* __restrict__ Pout = p.P + (((size_t)b * p.H + (y < p.H ? y : p.H - 1)) * p.W + (x < p.W ? x : p.W - 1)) * PCH + 4 * h;
    f32x4 r0v[PF], r1v[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        r0v[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        r1v[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }
#pragma unroll 1
    for (int mp = mp_begin; mp < mp_begin + mp_count; ++mp) {
        int off = tb_off;                                         // hide the base from LICM (see precompute_P_kernel)
        asm volatile("" : "+v"(off));
        const __bf16* tbm = tile + off;
        f32x16 a0, a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 s0 = *(const f32x4*)(Bk + 64 * mp + 8 * g);
            const f32x4 s1 = *(const f32x4*)(Bk + 64 * mp + 32 + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[4 * g + e] = s0[e];
                a1[4 * g + e] = s1[e];
            }
        }
#pragma unroll
        for (int ks = 0; ks < WPB_KS; ++ks) {
            const int t = ks >> 2, cg = ks & 3;
            const bf16x8 bv = *(const bf16x8*)(tbm + ((t / 3) * PT_LC + (t % 3)) * PB_PITCH + 16 * cg);
            a0 = MFMA_BF16(__builtin_bit_cast(bf16x8, r0v[ks % PF]), bv, a0);
            a1 = MFMA_BF16(__builtin_bit_cast(bf16x8, r1v[ks % PF]), bv, a1);
            r0v[ks % PF] = ld_piece(wrs, lane_off, wp + (2 * (ks + PF) + 0) * PIECE_BYTES);
            r1v[ks % PF] = ld_piece(wrs, lane_off, wp + (2 * (ks + PF) + 1) * PIECE_BYTES);
        }
        if (store) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v0, v1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = a0[4 * g + e];
                    v1[e] = a1[4 * g + e];
                }
                *(f32x4*)(Pout + 64 * mp + 8 * g) = v0;
                *(f32x4*)(Pout + 64 * mp + 32 + 8 * g) = v1;
            }
        }
        wp += WPB_KS * 2 * PIECE_BYTES;
    }
}

// ---------------------------------------------------------------------------------
// sine kernel (tests): the device sine of each mode, elementwise
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ void sin_kernel(const float* x, float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = dsin<MODE>(x[i]);
}

// ---------------------------------------------------------------------------------
// tables kernel (tests): the device evaluation of axis_eval
// ---------------------------------------------------------------------------------
__global__ void axis_tables_kernel(Axis a, int n_out, int32_t* idx, float* rel) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    int id;
    float r;
    axis_eval(a, j, id, r);
    if (idx) idx[j] = id;
    if (rel) rel[j] = r;
}

// ---------------------------------------------------------------------------------
// C ABI: launch functions
// ---------------------------------------------------------------------------------
static thread_local int g_last_hip_error = 0;
#ifdef DIINN_STAMPS
static unsigned long long* g_stamps = nullptr;
extern "C" int diinn_debug_set_stamp_buffer(void* dev_ptr) { g_stamps = (unsigned long long*)dev_ptr; return 0; }
#endif

static int hip_status(hipError_t e) {
    if (e == hipSuccess) return DIINN_OK;
    g_last_hip_error = (int)e;
    return DIINN_ERR_HIP;
}

extern "C" {

int diinn_last_hip_error(void) { return g_last_hip_error; }

int diinn_make_axis_tables_device(void* stream, int n_in, int n_out, int small_output,
                                  int32_t* idx_dev, float* rel_dev) {
    if (n_in <= 0 || n_out <= 0) return DIINN_ERR_INVALID_ARG;
    const Axis a = make_axis(n_in, n_out, small_output ? 1 : 0);
    hipLaunchKernelGGL(axis_tables_kernel, dim3((n_out + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, a, n_out, idx_dev, rel_dev);
    return hip_status(hipGetLastError());
}

int diinn_eval_sin_device(void* stream, int sin_mode, const float* x_dev, float* y_dev, int n) {
    if (!x_dev || !y_dev || n <= 0) return DIINN_ERR_INVALID_ARG;
    const dim3 grid((n + 255) / 256), blk(256);
    switch (sin_mode) {
        case DIINN_SIN_ACCURATE: hipLaunchKernelGGL(sin_kernel<DIINN_SIN_ACCURATE>, grid, blk, 0, (hipStream_t)stream, x_dev, y_dev, n); break;
        case DIINN_SIN_HW: hipLaunchKernelGGL(sin_kernel<DIINN_SIN_HW>, grid, blk, 0, (hipStream_t)stream, x_dev, y_dev, n); break;
        case DIINN_SIN_HW_REDUCED: hipLaunchKernelGGL(sin_kernel<DIINN_SIN_HW_REDUCED>, grid, blk, 0, (hipStream_t)stream, x_dev, y_dev, n); break;
        default: return DIINN_ERR_UNSUPPORTED;
    }
    return hip_status(hipGetLastError());
}

static int check_dims(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return DIINN_ERR_INVALID_ARG;
    if ((double)B * H * W * PCH >= 9.0e18 || B > 65535 || H > 65535) return DIINN_ERR_TOO_LARGE;
    return DIINN_OK;
}

static int launch_P(void* stream, const float* feat_dev, const float* packed_dev, float* P_dev,
                    int B, int H, int W, int r0, int r1, int mp_total, bool bf16 = false);

int diinn_precompute_P(void* stream, const float* feat_dev, const float* packed_dev,
                       float* P_dev, int B, int H, int W, int r0, int r1) {
    return launch_P(stream, feat_dev, packed_dev, P_dev, B, H, W, r0, r1, 16);
}

int diinn_precompute_P_ex(void* stream, const float* feat_dev, const float* packed_dev,
                          float* P_dev, int B, int H, int W, int r0, int r1, int compute) {
    if (compute != DIINN_COMPUTE_F32 && compute != DIINN_COMPUTE_BF16 && compute != DIINN_COMPUTE_F32_QONLY &&
        compute != DIINN_COMPUTE_BF16_FULL)
        return DIINN_ERR_UNSUPPORTED;
    return launch_P(stream, feat_dev, packed_dev, P_dev, B, H, W, r0, r1, 16, compute == DIINN_COMPUTE_BF16_FULL);
}

static int launch_P(void* stream, const float* feat_dev, const float* packed_dev, float* P_dev,
                    int B, int H, int W, int r0, int r1, int mp_total, bool bf16) {
    if (!feat_dev || !packed_dev || !P_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (r0 < 0 || r1 > H || r0 >= r1) return DIINN_ERR_INVALID_ARG;
    // Small maps: split the 1024 output channels over up to 16 workgroups per cell block so the
    // launch still fills the chip (2 workgroups/CU resident -> aim for >= 2 rounds of 512).
    const long long blocks = (long long)((W + PT_COLS - 1) / PT_COLS) * ((r1 - r0 + PT_ROWS - 1) / PT_ROWS) * B;
    int msplit = 1;
    while (msplit < mp_total && blocks * msplit < 1024) msplit *= 2;
    if ((long long)B * msplit > 65535) return DIINN_ERR_TOO_LARGE;
    PParams p{feat_dev, packed_dev, P_dev, B, H, W, r0, r1, msplit, mp_total};
    const dim3 grid((W + PT_COLS - 1) / PT_COLS, (r1 - r0 + PT_ROWS - 1) / PT_ROWS, B * msplit);
    if (bf16)
        hipLaunchKernelGGL(precompute_P_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(precompute_P_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_cell_chain(void* stream, float* P_dev, const float* packed_dev, int B, int H, int W, int r0, int r1) {
    if (!P_dev || !packed_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (r0 < 0 || r1 > H || r0 >= r1) return DIINN_ERR_INVALID_ARG;
    ChainParams p{P_dev, packed_dev, B, H, W, r0, r1};
    const dim3 grid((W + TILE_W * WG_TILES_X - 1) / (TILE_W * WG_TILES_X),
                    (r1 - r0 + TILE_H * WG_TILES_Y - 1) / (TILE_H * WG_TILES_Y), B);
    if (grid.y > 65535) return DIINN_ERR_TOO_LARGE;
    hipLaunchKernelGGL(cell_chain_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_decode_launch_info(int B, int Hu, int Wu, int y0, int y1,
                             int* grid_x, int* grid_y, int* grid_z, int* block) {
    if (B <= 0 || Hu <= 0 || Wu <= 0 || y0 < 0 || y1 > Hu || y0 >= y1) return DIINN_ERR_INVALID_ARG;
    if (grid_x) *grid_x = (Wu + TILE_W * WG_TILES_X - 1) / (TILE_W * WG_TILES_X);
    if (grid_y) *grid_y = (y1 - y0 + TILE_H * WG_TILES_Y - 1) / (TILE_H * WG_TILES_Y);
    if (grid_z) *grid_z = B;
    if (block) *block = 256;
    return DIINN_OK;
}

int diinn_decode_band(void* stream, const float* P_dev, const float* packed_dev,
                      float* out_dev, int B, int H, int W, int Hu, int Wu,
                      int y0, int y1, int sin_mode) {
    return diinn_decode_band_ex(stream, P_dev, packed_dev, out_dev, B, H, W, Hu, Wu, y0, y1, sin_mode,
                                DIINN_COMPUTE_F32);
}

int diinn_decode_band_ex(void* stream, const float* P_dev, const float* packed_dev,
                         float* out_dev, int B, int H, int W, int Hu, int Wu,
                         int y0, int y1, int sin_mode, int compute) {
    if (compute != DIINN_COMPUTE_F32 && compute != DIINN_COMPUTE_BF16 && compute != DIINN_COMPUTE_F32_QONLY &&
        compute != DIINN_COMPUTE_BF16_FULL)
        return DIINN_ERR_UNSUPPORTED;
    if (!P_dev || !packed_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Hu <= 0 || Wu <= 0 || y0 < 0 || y1 > Hu || y0 >= y1) return DIINN_ERR_INVALID_ARG;
    if ((double)Hu * Wu >= 2.0e9) return DIINN_ERR_TOO_LARGE;
    if (sin_mode < DIINN_SIN_ACCURATE || sin_mode > DIINN_SIN_HW_REDUCED) return DIINN_ERR_UNSUPPORTED;
    int gx, gy, gz, blk;
    diinn_decode_launch_info(B, Hu, Wu, y0, y1, &gx, &gy, &gz, &blk);
    if (gy > 65535 || gz > 65535) return DIINN_ERR_TOO_LARGE;   // HIP grid.y / grid.z limits
    DecodeParams p;
    p.P = P_dev; p.Wt = packed_dev; p.out = out_dev;
    p.B = B; p.H = H; p.W = W; p.Hu = Hu; p.Wu = Wu; p.y0 = y0; p.y1 = y1;
    p.ratio = (float)(((double)H * (double)W) / ((double)Hu * (double)Wu));
    p.acts = nullptr; p.npix = 0;
#ifdef DIINN_STAMPS
    p.stamps = g_stamps;
#endif
    const int small = diinn_uses_small_output_kernel(Hu, Wu);
    p.ah = make_axis(H, Hu, small);
    p.aw = make_axis(W, Wu, small);
    const dim3 grid(gx, gy, gz);
    if (compute == DIINN_COMPUTE_BF16 || compute == DIINN_COMPUTE_BF16_FULL) {
        // two pixel tiles per wave (half the weight bytes per MFMA) once the launch still fills the chip;
        // small images keep one tile per wave (twice the workgroups)
        const dim3 grid2(gx, (y1 - y0 + 2 * TILE_H * WG_TILES_Y - 1) / (2 * TILE_H * WG_TILES_Y), gz);   // 16 x 16 pixels per workgroup
        const bool two_tiles = (long long)grid2.x * grid2.y * grid2.z >= 512;
        if (!two_tiles) {
            if (sin_mode == DIINN_SIN_HW)
                hipLaunchKernelGGL(decode_bf16_kernel<DIINN_SIN_HW>, grid, dim3(blk), 0, (hipStream_t)stream, p);
            else if (sin_mode == DIINN_SIN_HW_REDUCED)
                hipLaunchKernelGGL(decode_bf16_kernel<DIINN_SIN_HW_REDUCED>, grid, dim3(blk), 0, (hipStream_t)stream, p);
            else
                hipLaunchKernelGGL(decode_bf16_kernel<DIINN_SIN_ACCURATE>, grid, dim3(blk), 0, (hipStream_t)stream, p);
        } else {
            if (sin_mode == DIINN_SIN_HW)
                hipLaunchKernelGGL(decode_bf16x2_kernel<DIINN_SIN_HW>, grid2, dim3(blk), 0, (hipStream_t)stream, p);
            else if (sin_mode == DIINN_SIN_HW_REDUCED)
                hipLaunchKernelGGL(decode_bf16x2_kernel<DIINN_SIN_HW_REDUCED>, grid2, dim3(blk), 0, (hipStream_t)stream, p);
            else
                hipLaunchKernelGGL(decode_bf16x2_kernel<DIINN_SIN_ACCURATE>, grid2, dim3(blk), 0, (hipStream_t)stream, p);
        }
        return hip_status(hipGetLastError());
    }
    if (compute == DIINN_COMPUTE_F32_QONLY) {
        if (sin_mode == DIINN_SIN_HW)
            hipLaunchKernelGGL((decode_kernel<DIINN_SIN_HW, false>), grid, dim3(blk), 0, (hipStream_t)stream, p);
        else if (sin_mode == DIINN_SIN_HW_REDUCED)
            hipLaunchKernelGGL((decode_kernel<DIINN_SIN_HW_REDUCED, false>), grid, dim3(blk), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((decode_kernel<DIINN_SIN_ACCURATE, false>), grid, dim3(blk), 0, (hipStream_t)stream, p);
        return hip_status(hipGetLastError());
    }
    if (sin_mode == DIINN_SIN_HW)
        hipLaunchKernelGGL(decode_kernel<DIINN_SIN_HW>, grid, dim3(blk), 0, (hipStream_t)stream, p);
    else if (sin_mode == DIINN_SIN_HW_REDUCED)
        hipLaunchKernelGGL(decode_kernel<DIINN_SIN_HW_REDUCED>, grid, dim3(blk), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(decode_kernel<DIINN_SIN_ACCURATE>, grid, dim3(blk), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

long long diinn_training_plane_floats(long long npix, int rows) {
    if (npix <= 0 || rows <= 0 || npix > DIINN_TRAIN_MAX_PIXELS) return -1;
    return (npix + PLANE_TILE - 1) / PLANE_TILE * rows * PLANE_TILE;
}

static int check_npix(long long npix) {
    if (npix <= 0) return DIINN_ERR_INVALID_ARG;
    if (npix > DIINN_TRAIN_MAX_PIXELS) return DIINN_ERR_TOO_LARGE;
    return DIINN_OK;
}

int diinn_decode_train_fwd(void* stream, const float* P_dev, const float* packed_dev, float* out_dev,
                           float* acts_dev, int B, int H, int W, int Hu, int Wu, int sin_mode) {
    if (!P_dev || !packed_dev || !out_dev || !acts_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Hu <= 0 || Wu <= 0) return DIINN_ERR_INVALID_ARG;
    if (sin_mode < DIINN_SIN_ACCURATE || sin_mode > DIINN_SIN_HW_REDUCED) return DIINN_ERR_UNSUPPORTED;
    if ((double)Hu * Wu >= 2.0e9) return DIINN_ERR_TOO_LARGE;
    const long long npix = (long long)B * Hu * Wu;
    st = check_npix(npix);
    if (st) return st;
    const dim3 grid((unsigned)((npix + 4 * PLANE_TILE - 1) / (4 * PLANE_TILE)));
    DecodeParams p;
    p.P = P_dev; p.Wt = packed_dev; p.out = out_dev;
    p.B = B; p.H = H; p.W = W; p.Hu = Hu; p.Wu = Wu; p.y0 = 0; p.y1 = Hu;
    p.ratio = (float)(((double)H * (double)W) / ((double)Hu * (double)Wu));
    p.acts = acts_dev; p.npix = npix;
#ifdef DIINN_STAMPS
    p.stamps = nullptr;
#endif
    const int small = diinn_uses_small_output_kernel(Hu, Wu);
    p.ah = make_axis(H, Hu, small);
    p.aw = make_axis(W, Wu, small);
    if (sin_mode == DIINN_SIN_HW)
        hipLaunchKernelGGL((decode_kernel<DIINN_SIN_HW, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (sin_mode == DIINN_SIN_HW_REDUCED)
        hipLaunchKernelGGL((decode_kernel<DIINN_SIN_HW_REDUCED, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((decode_kernel<DIINN_SIN_ACCURATE, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_backward_data(void* stream, const float* gout_planes_dev, const float* acts_dev,
                        const float* packed_dev, float* G_dev, float* Q_dev, long long npix) {
    if (!gout_planes_dev || !acts_dev || !packed_dev || !G_dev || !Q_dev) return DIINN_ERR_INVALID_ARG;
    const int stp = check_npix(npix);
    if (stp) return stp;
    BwdParams p;
    p.Wt = packed_dev; p.acts = acts_dev; p.gout = gout_planes_dev; p.G = G_dev; p.Q = Q_dev;
    p.npix = npix; p.ntiles = (npix + PLANE_TILE - 1) / PLANE_TILE; p.layer = 0;
    hipLaunchKernelGGL(bwd_head_kernel, dim3((unsigned)((npix + 255) / 256), HID / 16), dim3(256), 0,
                       (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_status(e);
    const unsigned blocks = (unsigned)((p.ntiles + 3) / 4);
    for (int layer = 3; layer >= 1; --layer) {
        p.layer = layer;
        hipLaunchKernelGGL(bwd_layer_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_status(e);
    }
    return DIINN_OK;
}

int diinn_plane_gemm_nt(void* stream, const float* A_dev, int a_rows, int a_row0, const float* B_dev, int b_rows,
                        int b_row0, float* part_dev, int M, int Nc, long long npix, int ksplit, int with_rowsum) {
    if (!A_dev || !B_dev || !part_dev || M <= 0 || Nc <= 0 || ksplit <= 0 || a_row0 < 0 || b_row0 < 0 ||
        a_row0 + M > a_rows || b_row0 + Nc > b_rows)
        return DIINN_ERR_INVALID_ARG;
    const int stp = check_npix(npix);
    if (stp) return stp;
    if (M % 128 || Nc % 128) return DIINN_ERR_UNSUPPORTED;
    if (ksplit > 65535) return DIINN_ERR_TOO_LARGE;
    PlaneGemmParams p;
    p.A = A_dev; p.Bm = B_dev; p.part = part_dev; p.npix = npix;
    p.a_rows = a_rows; p.a_row0 = a_row0; p.b_rows = b_rows; p.b_row0 = b_row0;
    p.M = M; p.Nc = Nc;
    p.ldc = Nc + (with_rowsum ? 1 : 0);
    p.with_rowsum = with_rowsum ? 1 : 0;
    const long long ntiles = (npix + PLANE_TILE - 1) / PLANE_TILE;
    const long long per = (ntiles + ksplit - 1) / ksplit;
    // offsets inside one split are 32-bit: tiles_per_split * rows * 128 bytes must stay below 2 GiB
    if (per * (long long)(a_rows > b_rows ? a_rows : b_rows) * PLANE_ROW_BYTES >= 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    p.tiles_per_split = (int)per;
#ifndef PLANE_GEMM_NB
#define PLANE_GEMM_NB 4
#endif
    if (PLANE_GEMM_NB == 4 && Nc % 256 == 0)
        hipLaunchKernelGGL(plane_gemm_kernel<4>, dim3((M / 128) * (Nc / 256), ksplit), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(plane_gemm_kernel<2>, dim3((M / 128) * (Nc / 128), ksplit), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_plane_rowdot(void* stream, const float* A_dev, int a_rows, const float* S_dev, float* part_dev,
                       int M, long long npix, int splits) {
    if (!A_dev || !S_dev || !part_dev || M <= 0 || M > a_rows || splits <= 0) return DIINN_ERR_INVALID_ARG;
    const int stp = check_npix(npix);
    if (stp) return stp;
    RowDotParams p;
    p.A = A_dev; p.S = S_dev; p.part = part_dev; p.npix = npix; p.a_rows = a_rows; p.M = M;
    const long long ntiles = (npix + PLANE_TILE - 1) / PLANE_TILE;
    p.tiles_per_split = (int)((ntiles + splits - 1) / splits);
    hipLaunchKernelGGL(plane_rowdot_kernel, dim3(splits), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_backward_cell_sum(void* stream, const float* G_dev, const int32_t* seg_h_dev, const int32_t* seg_w_dev,
                            float* dP_dev, int B, int H, int W, int Hu, int Wu) {
    if (!G_dev || !seg_h_dev || !seg_w_dev || !dP_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Hu <= 0 || Wu <= 0) return DIINN_ERR_INVALID_ARG;
    const long long npix = (long long)B * Hu * Wu;
    st = check_npix(npix);
    if (st) return st;
    if ((long long)((H + 3) / 4) * B > 65535) return DIINN_ERR_TOO_LARGE;
    CellSumParams p{G_dev, dP_dev, seg_h_dev, seg_w_dev, B, H, W, Hu, Wu, (npix + PLANE_TILE - 1) / PLANE_TILE};
    hipLaunchKernelGGL(cell_sum_kernel, dim3((W + 63) / 64, ((H + 3) / 4) * B, PCH), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_metasr_decode(void* stream, const float* feat_dev, const float* packed_dev, float* workspace_dev,
                        float* out_dev, int B, int H, int W, int Hu, int Wu) {
    if (!feat_dev || !packed_dev || !workspace_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Hu <= 0 || Wu <= 0) return DIINN_ERR_INVALID_ARG;
    if ((double)Hu * Wu >= 2.0e9) return DIINN_ERR_TOO_LARGE;
    UnfoldParams u{feat_dev, workspace_dev, B, H, W};
    hipLaunchKernelGGL(unfold_cells_kernel, dim3(W, H, B), dim3(192), 0, (hipStream_t)stream, u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_status(e);
    int gx, gy, gz, blk;
    diinn_decode_launch_info(B, Hu, Wu, 0, Hu, &gx, &gy, &gz, &blk);
    if (gy > 65535 || gz > 65535) return DIINN_ERR_TOO_LARGE;
    MetaParams p;
    p.U = workspace_dev; p.Wt = packed_dev; p.out = out_dev;
    p.B = B; p.H = H; p.W = W; p.Hu = Hu; p.Wu = Wu;
    p.ah = make_meta_axis(H, Hu);
    p.aw = make_meta_axis(W, Wu);
    hipLaunchKernelGGL(metasr_kernel, dim3(gx, gy, gz), dim3(blk), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_liif_decode(void* stream, const float* feat_dev, const float* packed_dev, float* workspace_dev,
                      float* out_dev, int B, int H, int W, int Hu, int Wu) {
    if (!out_dev || !workspace_dev) return DIINN_ERR_INVALID_ARG;
    int st = launch_P(stream, feat_dev, packed_dev, workspace_dev, B, H, W, 0, H, 4);   // first 256 channels of P
    if (st) return st;
    if (Hu <= 0 || Wu <= 0) return DIINN_ERR_INVALID_ARG;
    if ((double)Hu * Wu >= 2.0e9) return DIINN_ERR_TOO_LARGE;
    int gx, gy, gz, blk;
    diinn_decode_launch_info(B, Hu, Wu, 0, Hu, &gx, &gy, &gz, &blk);
    if (gy > 65535 || gz > 65535) return DIINN_ERR_TOO_LARGE;
    LiifParams p;
    p.P = workspace_dev; p.Wt = packed_dev; p.out = out_dev;
    p.B = B; p.H = H; p.W = W; p.Hu = Hu; p.Wu = Wu;
    p.ah = make_liif_axis(H, Hu);
    p.aw = make_liif_axis(W, Wu);
    hipLaunchKernelGGL(liif_kernel, dim3(gx, gy, gz), dim3(blk), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_decode(void* stream, const float* feat_dev, const float* packed_dev,
                 float* workspace_dev, float* out_dev,
                 int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode) {
    return diinn_decode_ex(stream, feat_dev, packed_dev, workspace_dev, out_dev, B, H, W, Hu, Wu, y0, y1,
                           sin_mode, DIINN_COMPUTE_F32);
}

int diinn_decode_ex(void* stream, const float* feat_dev, const float* packed_dev,
                    float* workspace_dev, float* out_dev,
                    int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode, int compute) {
    if (!workspace_dev) return DIINN_ERR_INVALID_ARG;

    int r0, r1;
    int st = diinn_lr_rows_for_band(H, Hu, Wu, y0, y1, &r0, &r1);
    if (st) return st;
    st = diinn_precompute_P_ex(stream, feat_dev, packed_dev, workspace_dev, B, H, W, r0, r1, compute);
    if (st) return st;
    if (compute == DIINN_COMPUTE_F32_QONLY) {                   // modes 1 and 2: per-cell modulation chain
        st = diinn_cell_chain(stream, workspace_dev, packed_dev, B, H, W, r0, r1);
        if (st) return st;
    }
    return diinn_decode_band_ex(stream, workspace_dev, packed_dev, out_dev, B, H, W, Hu, Wu, y0, y1, sin_mode,
                                compute);
}

}  // extern "C"
