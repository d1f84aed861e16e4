// diinn_decode.hip -- gfx950 (MI355X, CDNA4): the fp32 decode kernel of the DIINN implicit decoder and the
// decode entry points of the C ABI (include/diinn_hip.h).  Sibling translation units: see diinn_device.h.
//
// Reference path replaced: ImplicitDecoder.forward, mode 3
//   /root/reference/src/models/components/diinn.py:163-173 (+ :94-110, :132-139, :149-160)
//
// Kernels (DESIGN.md has the derivation, the roofline and the measurements of each):
//   precompute_P_kernel : per LR cell, P_i = Wx_i . unfold3x3(feat) + bK_i, i=0..3
//                         (implicit-im2col GEMM 576 -> 1024 on v_mfma_f32_32x32x2_f32,
//                         feature halo tile in LDS); precompute_P_bf16_kernel: bf16 operands
//   decode_kernel       : per HR pixel, the dual-branch MLP.  One wave owns 32 pixels
//                         and keeps their 256-channel activation in registers for the
//                         whole network: the accumulator layout of one layer IS the
//                         B-operand layout of the next (diinn_layout.h), so activations
//                         never touch LDS or HBM.  Weights stream from the packed image.
//                         <KPART=false>: decoder modes 1/2 (with cell_chain_kernel);
//                         <SAVE>: training forward, also writes k_i, s_i as tiled planes
//   decode_bf16_kernel, decode_bf16x2_kernel : bf16 operands in layers 1..3 (optional paths)
//   bwd_layer_kernel (+ bwd_head_kernel), plane_gemm_lds_kernel, plane_rowdot_kernel, cell_sum_kernel :
//                         backward pass of the decoder (training)
//   liif_kernel; unfold_cells_kernel + metasr_kernel : the LIIF and MetaSR comparison decoders
//   axis_tables_kernel, sin_kernel : the device coordinate / sine code, exposed for tests.
// One compile-time hook that never ships enabled: DIINN_STAMPS (s_memtime stamps for tools/stamp_report.py).  The
// timing-ablation hooks of rounds 1-4 (ABL_*: wrong results by construction) were removed in round 5; the builds behind
// profiles/r0[1-4]_*ablation*.txt are reproducible from the commits of those rounds (git show 84fe2c6:<path>).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (explicit fmaf where wanted)
#include "diinn_device.h"
#include <stdlib.h>

struct TagCoopF { static constexpr bool value = false; };
struct TagCoopT { static constexpr bool value = true; };

template <int SIN_MODE, bool KPART = true, bool SAVE = false>
__global__ __launch_bounds__(256, 1) void decode_kernel(const DecodeParams p) {
    // The small tables of layer 0 and of the head go through LDS: a vector-memory instruction blocks its wave for
    // ~60 cycles (stamps), and a wave reading them straight from the packed image issued 128 + 96
    // of those per tile.  Rows: Q0h, Q0w, t = fma(Q0r, ratio, bQ0) (the pixel-independent part of the sine
    // argument, the same first fma the per-pixel chain used to start with: results are bit-identical), L0, L1, L2.
    __shared__ __attribute__((aligned(16))) float tab[6 * HID + 4];      // + the head bias bL
    // Inference (no SAVE) keeps the synthesis branch in REVOLUTIONS: weights, biases and the Q0 table come from the
    // sections divided by 2 pi (WLR, BQR, Q0R), so every sine is v_fract + v_sin instead of a 5-op reduction (each VALU
    // instruction costs ~3 cycles of fp32-MFMA issue).  The training forward saves sine arguments in radians and
    // stays on the plain sections.
    constexpr size_t S_WL = SAVE ? OFF_WL : OFF_WLR, S_BQ = SAVE ? OFF_BQ : OFF_BQR, S_Q0 = SAVE ? OFF_Q0 : OFF_Q0R;
    auto sine = [](float v) {
        if constexpr (SAVE) return dsin<SIN_MODE>(v); else return dsin_rev<SIN_MODE>(v);
    };
    {
        const int i = threadIdx.x & 63, part = threadIdx.x >> 6;
        const float* __restrict__ Q0s = p.Wt + S_Q0 + 4 * i;
        if (part == 0) {
            *(f32x4*)(tab + 0 * HID + 4 * i) = *(const f32x4*)(Q0s + 0 * HID);
            *(f32x4*)(tab + 1 * HID + 4 * i) = *(const f32x4*)(Q0s + 1 * HID);
        } else if (part == 1) {
            const f32x4 wr = *(const f32x4*)(Q0s + 2 * HID), bq = *(const f32x4*)(Q0s + 3 * HID);
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(wr[e], p.ratio, bq[e]);
            *(f32x4*)(tab + 2 * HID + 4 * i) = t;
        } else if (part == 2) {
            *(f32x4*)(tab + 3 * HID + 4 * i) = *(const f32x4*)(p.Wt + OFF_L + 0 * HID + 4 * i);
            *(f32x4*)(tab + 4 * HID + 4 * i) = *(const f32x4*)(p.Wt + OFF_L + 1 * HID + 4 * i);
        } else {
            *(f32x4*)(tab + 5 * HID + 4 * i) = *(const f32x4*)(p.Wt + OFF_L + 2 * HID + 4 * i);
            if (i == 0) {                                        // bL + the image's validity word: inference reads derived
                const f32x4 bl = *(const f32x4*)(p.Wt + OFF_BL); // sections, so an image without them decodes to NaN
                // (the word is read on its own: hipcc 7.2 compiled `bit_cast<unsigned>(bl[3]) == MAGIC` on the scalarised
                // vector load into a compare of element 0)
                const unsigned nanm = SAVE ? 0u : derived_nan_mask(p.Wt);
                *(f32x4*)(tab + 6 * HID) = or_bits(bl, nanm);
            }
        }
    }
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
        x = p.x0 + blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
        y = p.y0 + blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
        b = blockIdx.z;
    }
    const bool valid = (x < p.x1) && (y < p.y1);
    __syncthreads();                                             // the tables are in LDS
    // whole wave outside the band/image: nothing to do (wave-uniform; no barrier follows)
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
    const float* __restrict__ Pc = p.P + (((size_t)b * p.Prows + (iy - p.Prow0)) * p.W + ix) * PCH + 4 * h;

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
        const float* __restrict__ Q0 = tab + 4 * h;
        const __amdgpu_buffer_rsrc_t ar0 = act_rsrc(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 pv = *(const f32x4*)(Pc + c0);
                const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                const f32x4 tq = *(const f32x4*)(Q0 + 2 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = tq[e];
                    a = __builtin_fmaf(ww[e], relw, a);
                    a = __builtin_fmaf(wh[e], relh, a);
                    const float kv = relu0(pv[e]);
                    q[16 * m + 4 * g + e] = kv * sine(a);
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
    int wp = (int)(S_WL * sizeof(float));                        // byte offset; advances one layer per iteration
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
        sq[g] = *(const f32x4*)(Wt + S_BQ + 4 * h + 8 * g);
    }
#pragma unroll 1
    for (int layer = 0; layer < 3; ++layer) {
        const int nl = layer < 2 ? layer + 1 : 2;                // seeds of the next layer's tile 0 (clamped)
        const float* __restrict__ Pl = Pc + (layer + 1) * HID;
        const float* __restrict__ Bq = Wt + S_BQ + layer * HID + 4 * h;
        const float* __restrict__ Pn = Pc + (nl + 1) * HID;
        const float* __restrict__ Bn = Wt + S_BQ + nl * HID + 4 * h;
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
                if constexpr (KPART) rk[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 0) * PIECE_BYTES);
                rq[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 1) * PIECE_BYTES);
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
                    qn[16 * (m - 1) + r] = kv * sine(ps[r]);
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
            qn[16 * 7 + r] = kv * sine(ps[r]);
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
        const float* __restrict__ L = tab + 3 * HID + 4 * h;
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
        const long long plane = p.o_ps;
        float* o = out_px(p, b, y, x);
        o[0] = o0 + tab[6 * HID + 0];
        o[plane] = o1 + tab[6 * HID + 1];
        o[2 * plane] = o2 + tab[6 * HID + 2];
    }
    STAMP(5);
}



// ---------------------------------------------------------------------------------
// decode_coop16_kernel: the fp32 latency form for small and partly filled launches (BASELINE config 1: 96 x 96).
// decode_kernel gives one wave a 32-pixel tile and 6,144 dependent MFMAs (0.165 ms at any size).  (Round 2's form -- four
// waves sharing ONE 32-pixel tile, a chain of 3 x 512 MFMAs of 64 clocks, 41 us -- never won against this one and was
// deleted in round 5: DESIGN_HISTORY.md.)  Here a workgroup owns a
// 4 x 4 = 16-pixel tile on v_mfma_f32_16x16x4_f32 (32 clocks): wave w owns output channels 64 w .. 64 w + 63 of both
// branches as four 16-row M-tiles each (8 accumulators of 4 registers, 512 MFMAs per layer, 20 us per tile), twice
// the workgroups at half the chain, four workgroups per CU (38 KiB of LDS, < 128 registers).  Measured at c1 (r04,
// profiles/r04_c1_latency.txt): decode 0.099 -> 0.076-0.079 ms, step 0.122 -> 0.100-0.102 ms.  What bounds it now is the
// makespan of 576 tiles on 256 CUs (three per busiest CU x ~23 us); capping the workgroups per CU at 3 or 2, weights kept
// L1-resident, or no weight requests at all (timing ablations) move it by < 4 %.
//   * A operands: packed section 16 (WL16), [i][half][lane][T]: a k-step's two 1 KiB pieces feed its 8 MFMAs;
//   * B operand: the activation in LDS in POSITION order -- position p = the p-th term of decode_kernel's accumulation
//     chain, channel chan_of(p >> 1, p & 1) -- as [p >> 4][16 (p & 3) + pixel][(p >> 2) & 3]: lane (g, n) reads the
//     four k-steps 4 ib .. 4 ib + 3 of its k = g with one ds_read_b128;
//   * per output channel the arithmetic is decode_kernel's: same seed, the same k-ordered chain (the 16x16x4 MFMA adds
//     its four products in k order like the 32x32x2 one adds its two: tests hold the two kernels bit-equal), same
//     epilogue, and wave 0 runs decode_kernel's head order on the last activation.
// ---------------------------------------------------------------------------------
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
constexpr int T16_W = 4, T16_H = 4;

template <int SIN_MODE>
__global__ __launch_bounds__(256, 4) void decode_coop16_kernel(const DecodeParams p) {
    __shared__ __attribute__((aligned(16))) f32x4 qs[2][16][64];           // 2 x 16 KiB: activation in position order
    __shared__ __attribute__((aligned(16))) float tab[6 * HID + 4];        // Q0h, Q0w, fma(Q0r, ratio, bQ0), L0..L2, bL
    {
        const int i = threadIdx.x & 63, part = threadIdx.x >> 6;
        const float* __restrict__ Q0s = p.Wt + OFF_Q0R + 4 * i;
        if (part == 0) {
            *(f32x4*)(tab + 0 * HID + 4 * i) = *(const f32x4*)(Q0s + 0 * HID);
            *(f32x4*)(tab + 1 * HID + 4 * i) = *(const f32x4*)(Q0s + 1 * HID);
        } else if (part == 1) {
            const f32x4 wr = *(const f32x4*)(Q0s + 2 * HID), bq = *(const f32x4*)(Q0s + 3 * HID);
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(wr[e], p.ratio, bq[e]);
            *(f32x4*)(tab + 2 * HID + 4 * i) = t;
        } else if (part == 2) {
            *(f32x4*)(tab + 3 * HID + 4 * i) = *(const f32x4*)(p.Wt + OFF_L + 0 * HID + 4 * i);
            *(f32x4*)(tab + 4 * HID + 4 * i) = *(const f32x4*)(p.Wt + OFF_L + 1 * HID + 4 * i);
        } else {
            *(f32x4*)(tab + 5 * HID + 4 * i) = *(const f32x4*)(p.Wt + OFF_L + 2 * HID + 4 * i);
            if (i == 0) *(f32x4*)(tab + 6 * HID) = or_bits(*(const f32x4*)(p.Wt + OFF_BL), derived_nan_mask(p.Wt));
        }
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, n = lane & 15;
    const int x = p.x0 + blockIdx.x * T16_W + (n & (T16_W - 1));
    const int y = p.y0 + blockIdx.y * T16_H + (n / T16_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.x1) && (y < p.y1);
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.y1 ? y : p.y1 - 1;
    int iy, ix;
    float relh, relw;
    axis_eval(p.ah, yc, iy, relh);
    axis_eval(p.aw, xc, ix, relw);
    const float* __restrict__ Wt = p.Wt;
    // this lane's channels: 64 wave + 16 T + 4 g + j (T, j = 0..3) = accumulator register j of M-tile T
    const float* __restrict__ Pc = p.P + (((size_t)b * p.Prows + (iy - p.Prow0)) * p.W + ix) * PCH + 64 * wave + 4 * g;
    // where channel 64 wave + 16 T + 4 g + j goes in a position-ordered image (floats): base + 256 T + {0, 128, 1, 129}[j]
    // (position 32 m + 8 a + 2 j + h with m = 2 wave + (T >> 1), a = 2 (T & 1) + (g >> 1), h = g & 1)
    float* const qf = reinterpret_cast<float*>(&qs[0][0][0]);
    const int wbase = (4 * wave * 64 + 16 * (g & 1) + n) * 4 + 2 * (g >> 1);
    auto put = [&](const int img, const int T, const f32x4 v) {
        float* d = qf + img * (16 * 64 * 4) + wbase + 256 * T;
        d[0] = v[0]; d[128] = v[1]; d[1] = v[2]; d[129] = v[3];
    };
    __syncthreads();

    // ---- layer 0 (diinn.py:133-134), decode_kernel's operation order
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const int c0 = 64 * wave + 16 * T + 4 * g;
        const f32x4 pv = *(const f32x4*)(Pc + 16 * T);
        const f32x4 wh = *(const f32x4*)(tab + 0 * HID + c0);
        const f32x4 ww = *(const f32x4*)(tab + 1 * HID + c0);
        const f32x4 tq = *(const f32x4*)(tab + 2 * HID + c0);
        f32x4 q0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = tq[e];
            a = __builtin_fmaf(ww[e], relw, a);
            a = __builtin_fmaf(wh[e], relh, a);
            q0[e] = relu0(pv[e]) * dsin_rev<SIN_MODE>(a);
        }
        put(0, T, q0);
    }

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WL16 * sizeof(float)) + wave * (int)(WL16_WAVE * sizeof(float));
    auto ld_w = [&](const int i, const int half) {
        return ld_piece(wrs, lane_off, wp + (2 * i + half) * PIECE_BYTES);
    };
    __syncthreads();

    // The weight ring (RING - 1 k-steps of two pieces in flight; 8 MFMAs = 256 clocks per k-step) and the accumulator seeds
    // run on ACROSS the layers: the last steps of a layer request the next layer's first pieces, and its seeds (P_{i+1} of
    // the pixel's cell, bQ) are requested in the middle of the layer before -- nothing a layer needs is fetched at its start.
    constexpr int RING = 4;
    f32x4 rw[RING][2];
#pragma unroll
    for (int d = 0; d < RING - 1; ++d) {
        rw[d][0] = ld_w(d, 0);
        rw[d][1] = ld_w(d, 1);
    }
    f32x4 seed[8];
    auto ld_seeds = [&](const int layer) {
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            seed[T] = *(const f32x4*)(Pc + (layer + 1) * HID + 16 * T);
            seed[4 + T] = *(const f32x4*)(Wt + OFF_BQR + layer * HID + 64 * wave + 16 * T + 4 * g);
        }
    };
    ld_seeds(0);
    auto layer_body = [&](auto cur_tag, auto last_tag, const int layer) {
        constexpr int CUR = decltype(cur_tag)::value ? 1 : 0;
        constexpr bool LAST = decltype(last_tag)::value;
        f32x4 acc[8];                                            // [T]: modulation, [4 + T]: synthesis
#pragma unroll
        for (int T = 0; T < 8; ++T) acc[T] = seed[T];
        f32x4 bq = qs[CUR][0][lane];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float bv = bq[i & 3];
            if ((i & 3) == 3 && i + 1 < 64) bq = qs[CUR][(i + 1) >> 2][lane];
            if (i + RING - 1 < 64) {
                rw[(i + RING - 1) % RING][0] = ld_w(i + RING - 1, 0);
                rw[(i + RING - 1) % RING][1] = ld_w(i + RING - 1, 1);
            } else if (!LAST) {                                  // the next layer's first k-steps (64 % RING == 0: same slots)
                rw[(i + RING - 1) % RING][0] = ld_w(i + RING - 1 + (int)(WL16_LAYER / WL16_KSTEP) - 64, 0);
                rw[(i + RING - 1) % RING][1] = ld_w(i + RING - 1 + (int)(WL16_LAYER / WL16_KSTEP) - 64, 1);
            }
            if (i == 32 && !LAST) ld_seeds(layer + 1);
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                acc[T] = MFMA16(rw[i % RING][0][T], bv, acc[T]);
                acc[4 + T] = MFMA16(rw[i % RING][1][T], bv, acc[4 + T]);
            }
            // a k-step's two requests stay RING - 1 steps (768 clocks of MFMAs) ahead of their use: left alone, hipcc sinks
            // every load to 4 MFMAs in front of its first use and the wave waits for the L2 at each step
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: q = relu(k) * sin(s) for this wave's 64 channels -> the other image
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            f32x4 qn;
#pragma unroll
            for (int e = 0; e < 4; ++e) qn[e] = relu0(acc[T][e]) * dsin_rev<SIN_MODE>(acc[4 + T][e]);
            put(1 - CUR, T, qn);
        }
        __syncthreads();
    };
    static_assert(64 % RING == 0 && WL16_LAYER == 4 * WL16_WAVE && WL16_WAVE == 64 * WL16_KSTEP, "ring runs on across layers");
    layer_body(TagCoopF{}, TagCoopF{}, 0);                       // image 0 -> 1
    wp += (int)(WL16_LAYER * sizeof(float));
    layer_body(TagCoopT{}, TagCoopF{}, 1);                       // 1 -> 0
    wp += (int)(WL16_LAYER * sizeof(float));
    layer_body(TagCoopF{}, TagCoopT{}, 2);                       // 0 -> 1

    // ---- head (diinn.py:138): decode_kernel's order -- per lane half h the channels 32 m + 8 gq + 4 h + e in (m, gq, e)
    // order, then the two halves added -- on the last activation (image 1), by lanes (h, n) of wave 0
    if (wave == 0 && lane < 32) {
        const int h = lane >> 4;
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
        const float* __restrict__ L = tab + 3 * HID + 4 * h;
        const float* __restrict__ q1 = qf + 16 * 64 * 4 + (16 * h + n) * 4;     // position 32 m + 8 gq + 2 e + h
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c0 = 32 * m + 8 * gq;
                const f32x4 l0 = *(const f32x4*)(L + 0 * HID + c0);
                const f32x4 l1 = *(const f32x4*)(L + 1 * HID + c0);
                const f32x4 l2 = *(const f32x4*)(L + 2 * HID + c0);
                // positions 32 m + 8 gq + 2 e + h: image row 2 m + (gq >> 1), lane group 2 (e & 1) + h, element 2 (gq & 1) + (e >> 1)
                const float* __restrict__ r = q1 + (2 * m + (gq >> 1)) * 256 + 2 * (gq & 1);
                const float v[4] = {r[0], r[128], r[1], r[129]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0 = __builtin_fmaf(l0[e], v[e], o0);
                    o1 = __builtin_fmaf(l1[e], v[e], o1);
                    o2 = __builtin_fmaf(l2[e], v[e], o2);
                }
            }
        }
        o0 += __shfl_xor(o0, 16);
        o1 += __shfl_xor(o1, 16);
        o2 += __shfl_xor(o2, 16);
        if (valid && h == 0) {
            const long long plane = p.o_ps;
            float* op = out_px(p, b, y, x);
            op[0] = o0 + tab[6 * HID + 0];
            op[plane] = o1 + tab[6 * HID + 1];
            op[2 * plane] = o2 + tab[6 * HID + 2];
        }
    }
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
    float* P;            // [B,Prows,W,1024] = LR rows [Prow0, Prow0+Prows); slots 1..3 updated in place for rows [r0,r1)
    const float* Wt;
    int B, H, W, r0, r1, Prow0, Prows;
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
    float* __restrict__ Pc = p.P + (((size_t)b * p.Prows + (yc - p.Prow0)) * p.W + xc) * PCH + 4 * h;

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


static int cell_chain_impl(void* stream, float* P_dev, const float* packed_dev, int B, int H, int W, int r0, int r1,
                           RowWin pw) {
    if (!P_dev || !packed_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (r0 < 0 || r1 > H || r0 >= r1) return DIINN_ERR_INVALID_ARG;
    st = check_window(pw.row0, pw.rows, H, r0, r1);
    if (st) return st;
    ChainParams p{P_dev, packed_dev, B, H, W, r0, r1, pw.row0, pw.rows};
    const dim3 grid((W + TILE_W * WG_TILES_X - 1) / (TILE_W * WG_TILES_X),
                    (r1 - r0 + TILE_H * WG_TILES_Y - 1) / (TILE_H * WG_TILES_Y), B);
    if (grid.y > 65535) return DIINN_ERR_TOO_LARGE;
    hipLaunchKernelGGL(cell_chain_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

// Where the pixels of a launch go: `out` points at pixel (b = 0, c = 0, y = row0, x = col0) of the caller's buffer, the
// strides are in floats.  The row-band entry points describe a contiguous [B,3,rows,Wu] window this way (contig()).
struct OutView { int row0, rows, col0; long long bs, ps, rs; };
static inline OutView contig(RowWin ow, int Wu) {
    return OutView{ow.row0, ow.rows, 0, 3LL * ow.rows * Wu, (long long)ow.rows * Wu, (long long)Wu};
}

// HR rows [y0,y1) x columns [x0,x1) of the image, every arithmetic mode.  Blocks are anchored at (x0, y0); no pixel's
// arithmetic depends on its place in a block, and every kernel-variant choice that is not bit-neutral is made from the
// FULL image, so a tile is bit-identical to the same pixels of a whole-image decode.
// Which fp32 kernel a tile launch takes, and its grid (one function for the launch and for diinn_decode_kernel_info).
// decode_kernel runs whole ROUNDS of one workgroup of 16 x 8 pixels per compute unit, 181 us each whatever the fill; the
// 16-pixel latency kernel is work-conserving (four workgroups per CU) at 0.0896 us per tile on 256 CUs + ~25 us -- 1.3 %
// behind on exact rounds (c2: 5.87 vs 5.79 ms), far ahead on a partly filled last round (48 -> 192: 0.234 vs 0.363 ms;
// r04, tools/f32_kernel_choice.py, profiles/r04_f32_kernel_choice.txt).  The launch takes the cheaper by that model, with
// the device's own CU count in it.  DIINN_F32_KERNEL = 1 / 3 forces the throughput / the 16-pixel latency kernel (tests,
// A-B timing).
struct F32Choice { int kernel; dim3 grid; };
static F32Choice f32_kernel_choice(int B, int y0, int y1, int x0, int x1) {
    const int gx = (x1 - x0 + TILE_W * WG_TILES_X - 1) / (TILE_W * WG_TILES_X);
    const int gy = (y1 - y0 + TILE_H * WG_TILES_Y - 1) / (TILE_H * WG_TILES_Y);
    const int force = (int)knob(diinn_knobs().f32_kernel);
    const int ncu = device_cus();
    const dim3 grid16((x1 - x0 + T16_W - 1) / T16_W, (y1 - y0 + T16_H - 1) / T16_H, B);
    const double t16 = (double)grid16.x * grid16.y * grid16.z, rounds = (double)(((long long)gx * gy * B + ncu - 1) / ncu);
    const bool cheaper16 = 0.0896 * t16 * (256.0 / ncu) + 25.0 < 181.1 * rounds + 3.0;
    if (grid16.y <= 65535 && (force ? force == 3 : cheaper16)) return F32Choice{DIINN_DECODE_KERNEL_LATENCY16, grid16};
    return F32Choice{DIINN_DECODE_KERNEL_THROUGHPUT, dim3(gx, gy, B)};
}

static int decode_tile_impl(void* stream, const float* P_dev, const float* packed_dev,
                            float* out_dev, int B, int H, int W, int Hu, int Wu,
                            int y0, int y1, int x0, int x1, int sin_mode, int compute, RowWin pw, OutView ov) {
    if (!compute_ok(compute))
        return DIINN_ERR_UNSUPPORTED;
    if (!P_dev || !packed_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Hu <= 0 || Wu <= 0 || y0 < 0 || y1 > Hu || y0 >= y1 || x0 < 0 || x1 > Wu || x0 >= x1) return DIINN_ERR_INVALID_ARG;
    if ((double)Hu * Wu >= 2.0e9) return DIINN_ERR_TOO_LARGE;
    {
        int r0, r1;
        st = diinn_lr_rows_for_band(H, Hu, Wu, y0, y1, &r0, &r1);
        if (st) return st;
        st = check_window(pw.row0, pw.rows, H, r0, r1);
        if (st) return st;
        st = check_window(ov.row0, ov.rows, Hu, y0, y1);
        if (st) return st;
    }
    // strides: a row holds the tile's columns, a plane its rows, a batch item its three planes (no overlap); < 2^40 floats
    if (ov.col0 < 0 || ov.col0 > x0 || ov.rs < (long long)(x1 - ov.col0) || ov.ps < ov.rs * (y1 - ov.row0 - 1) + (x1 - ov.col0) ||
        ov.bs < 2 * ov.ps + ov.rs * (y1 - ov.row0 - 1) + (x1 - ov.col0) || ov.bs > (1LL << 40))
        return DIINN_ERR_INVALID_ARG;
    if (sin_mode < DIINN_SIN_ACCURATE || sin_mode > DIINN_SIN_HW_REDUCED) return DIINN_ERR_UNSUPPORTED;
    const int gx = (x1 - x0 + TILE_W * WG_TILES_X - 1) / (TILE_W * WG_TILES_X);
    const int gy = (y1 - y0 + TILE_H * WG_TILES_Y - 1) / (TILE_H * WG_TILES_Y);
    const int gz = B, blk = 256;
    if (gy > 65535 || gz > 65535) return DIINN_ERR_TOO_LARGE;   // HIP grid.y / grid.z limits
    DecodeParams p;
    p.P = P_dev; p.Wt = packed_dev; p.out = out_dev;
    p.B = B; p.H = H; p.W = W; p.Hu = Hu; p.Wu = Wu; p.y0 = y0; p.y1 = y1;
    p.ratio = (float)(((double)H * (double)W) / ((double)Hu * (double)Wu));
    p.Prow0 = pw.row0; p.Prows = pw.rows; p.Orow0 = ov.row0; p.Orows = ov.rows;
    p.x0 = x0; p.x1 = x1; p.Ocol0 = ov.col0; p.o_bs = ov.bs; p.o_ps = ov.ps; p.o_rs = ov.rs;
    p.acts = nullptr; p.npix = 0; p.seed_cols = 0; p.xcd_runs = 0;
    for (int i = 0; i < 6; ++i) p.pg[i] = 0;
#ifdef DIINN_STAMPS
    p.stamps = g_stamps;
#endif
    const int small = diinn_uses_small_output_kernel(Hu, Wu);
    p.ah = make_axis(H, Hu, small);
    p.aw = make_axis(W, Wu, small);
    const dim3 grid(gx, gy, gz);
    if (compute == DIINN_COMPUTE_BF16 || compute == DIINN_COMPUTE_BF16_FULL)
        return launch_decode_bf16(stream, p, gx, gy, gz, sin_mode);
    if (compute == DIINN_COMPUTE_BF16X3)
        return launch_decode_bf16x3(stream, p, gx, gy, gz, sin_mode);
    if (compute == DIINN_COMPUTE_F32_QONLY) {
        if (sin_mode == DIINN_SIN_HW)
            hipLaunchKernelGGL((decode_kernel<DIINN_SIN_HW, false>), grid, dim3(blk), 0, (hipStream_t)stream, p);
        else if (sin_mode == DIINN_SIN_HW_REDUCED)
            hipLaunchKernelGGL((decode_kernel<DIINN_SIN_HW_REDUCED, false>), grid, dim3(blk), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((decode_kernel<DIINN_SIN_ACCURATE, false>), grid, dim3(blk), 0, (hipStream_t)stream, p);
        return hip_status(hipGetLastError());
    }
    // Which of the two fp32 kernels (bit-equal: tests/test_gpu_parity.py::test_latency_kernel_is_bit_identical_...)?
    const F32Choice ch = f32_kernel_choice(B, y0, y1, x0, x1);
    if (ch.kernel == DIINN_DECODE_KERNEL_LATENCY16) {
        if (sin_mode == DIINN_SIN_HW)
            hipLaunchKernelGGL(decode_coop16_kernel<DIINN_SIN_HW>, ch.grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (sin_mode == DIINN_SIN_HW_REDUCED)
            hipLaunchKernelGGL(decode_coop16_kernel<DIINN_SIN_HW_REDUCED>, ch.grid, dim3(256), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(decode_coop16_kernel<DIINN_SIN_ACCURATE>, ch.grid, dim3(256), 0, (hipStream_t)stream, p);
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

static int decode_band_impl(void* stream, const float* P_dev, const float* packed_dev,
                            float* out_dev, int B, int H, int W, int Hu, int Wu,
                            int y0, int y1, int sin_mode, int compute, RowWin pw, RowWin ow) {
    if (Wu <= 0) return DIINN_ERR_INVALID_ARG;
    return decode_tile_impl(stream, P_dev, packed_dev, out_dev, B, H, W, Hu, Wu, y0, y1, 0, Wu, sin_mode, compute, pw,
                            contig(ow, Wu));
}

static int decode_impl(void* stream, const float* feat_dev, const float* packed_dev,
                       float* workspace_dev, float* out_dev,
                       int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode, int compute,
                       RowWin fw, RowWin pw, RowWin ow) {
    if (!workspace_dev) return DIINN_ERR_INVALID_ARG;

    int r0, r1;
    int st = diinn_lr_rows_for_band(H, Hu, Wu, y0, y1, &r0, &r1);
    if (st) return st;
    if (!compute_ok(compute))
        return DIINN_ERR_UNSUPPORTED;
    st = launch_P(stream, feat_dev, packed_dev, workspace_dev, B, H, W, r0, r1, 16, p_arith(compute),
                  &fw, &pw, true);
    if (st) return st;
    if (compute == DIINN_COMPUTE_F32_QONLY) {                   // modes 1 and 2: per-cell modulation chain
        st = cell_chain_impl(stream, workspace_dev, packed_dev, B, H, W, r0, r1, pw);
        if (st) return st;
    }
    return decode_band_impl(stream, workspace_dev, packed_dev, out_dev, B, H, W, Hu, Wu, y0, y1, sin_mode,
                            compute, pw, ow);
}

extern "C" {

int diinn_cell_chain(void* stream, float* P_dev, const float* packed_dev, int B, int H, int W, int r0, int r1) {
    return cell_chain_impl(stream, P_dev, packed_dev, B, H, W, r0, r1, RowWin{0, H});
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

int diinn_decode_kernel_info(int B, int Hu, int Wu, int y0, int y1, int x0, int x1, int compute, int info[4]) {
    if (!info || B <= 0 || Hu <= 0 || Wu <= 0 || y0 < 0 || y1 > Hu || y0 >= y1 || x0 < 0 || x1 > Wu || x0 >= x1) return DIINN_ERR_INVALID_ARG;
    if (!compute_ok(compute)) return DIINN_ERR_UNSUPPORTED;
    if (compute == DIINN_COMPUTE_F32) {
        const F32Choice ch = f32_kernel_choice(B, y0, y1, x0, x1);
        info[0] = ch.kernel; info[1] = (int)ch.grid.x; info[2] = (int)ch.grid.y; info[3] = (int)ch.grid.z;
        return DIINN_OK;
    }
    info[0] = compute == DIINN_COMPUTE_F32_QONLY ? DIINN_DECODE_KERNEL_THROUGHPUT : DIINN_DECODE_KERNEL_OTHER;
    info[1] = (x1 - x0 + TILE_W * WG_TILES_X - 1) / (TILE_W * WG_TILES_X);
    info[2] = (y1 - y0 + TILE_H * WG_TILES_Y - 1) / (TILE_H * WG_TILES_Y);
    info[3] = B;
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
    return decode_band_impl(stream, P_dev, packed_dev, out_dev, B, H, W, Hu, Wu, y0, y1, sin_mode, compute,
                            RowWin{0, H}, RowWin{0, Hu});
}

int diinn_decode_band_win(void* stream, const float* P_win_dev, int p_row0, int p_rows, const float* packed_dev,
                          float* out_win_dev, int out_row0, int out_rows,
                          int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode, int compute) {
    return decode_band_impl(stream, P_win_dev, packed_dev, out_win_dev, B, H, W, Hu, Wu, y0, y1, sin_mode, compute,
                            RowWin{p_row0, p_rows}, RowWin{out_row0, out_rows});
}

int diinn_decode_tile_win(void* stream, const float* P_win_dev, int p_row0, int p_rows, const float* packed_dev,
                          float* out_tile_dev, long long out_row_stride, long long out_plane_stride,
                          long long out_batch_stride, int B, int H, int W, int Hu, int Wu,
                          int y0, int y1, int x0, int x1, int sin_mode, int compute) {
    // the view starts at the tile's own first pixel: row0 = y0, col0 = x0, rows = what the tile covers
    return decode_tile_impl(stream, P_win_dev, packed_dev, out_tile_dev, B, H, W, Hu, Wu, y0, y1, x0, x1, sin_mode, compute,
                            RowWin{p_row0, p_rows},
                            OutView{y0, y1 > y0 ? y1 - y0 : 1, x0, out_batch_stride, out_plane_stride, out_row_stride});
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
    p.Prow0 = 0; p.Prows = H; p.Orow0 = 0; p.Orows = Hu;
    p.x0 = 0; p.x1 = Wu; p.Ocol0 = 0; p.o_rs = Wu; p.o_ps = (long long)Hu * Wu; p.o_bs = 3 * p.o_ps;
    p.acts = acts_dev; p.npix = npix; p.seed_cols = 0; p.xcd_runs = 0;
    for (int i = 0; i < 6; ++i) p.pg[i] = 0;
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

int diinn_decode(void* stream, const float* feat_dev, const float* packed_dev,
                 float* workspace_dev, float* out_dev,
                 int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode) {
    return diinn_decode_ex(stream, feat_dev, packed_dev, workspace_dev, out_dev, B, H, W, Hu, Wu, y0, y1,
                           sin_mode, DIINN_COMPUTE_F32);
}

int diinn_decode_ex(void* stream, const float* feat_dev, const float* packed_dev,
                    float* workspace_dev, float* out_dev,
                    int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode, int compute) {
    return decode_impl(stream, feat_dev, packed_dev, workspace_dev, out_dev, B, H, W, Hu, Wu, y0, y1, sin_mode, compute,
                       RowWin{0, H}, RowWin{0, H}, RowWin{0, Hu});
}

int diinn_decode_win(void* stream, const float* feat_win_dev, int feat_row0, int feat_rows,
                     const float* packed_dev, float* P_win_dev, int p_row0, int p_rows,
                     float* out_win_dev, int out_row0, int out_rows,
                     int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode, int compute) {
    return decode_impl(stream, feat_win_dev, packed_dev, P_win_dev, out_win_dev, B, H, W, Hu, Wu, y0, y1, sin_mode,
                       compute, RowWin{feat_row0, feat_rows}, RowWin{p_row0, p_rows}, RowWin{out_row0, out_rows});
}

}  // extern "C"
