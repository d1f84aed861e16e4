// diinn_bf16.hip -- the optional bf16-operand decode kernels (DIINN_COMPUTE_BF16 / DIINN_COMPUTE_BF16_FULL)
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
#include "diinn_device.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------
// decode kernel, bf16 operands (optional path, BASELINE config 5; tolerance restated in
// DESIGN.md): same structure as decode_kernel -- one wave owns 32 pixels and keeps their
// activation in registers -- but layers 1..3 run on v_mfma_f32_32x32x16_bf16: weights are bf16
// (WLB section), the activation is packed to bf16 straight from the epilogue (accumulator registers
// 8s..8s+7 of a tile ARE the B fragment of k-step 2m+s, see chan_of_bf16), accumulation, seeds
// (P, biases), sine, layer 0 and the RGB head stay fp32.  A first version: the weight stream
// (1 KiB per MFMA per wave, 8x the fp32 path's bytes per cycle) is L1-bandwidth-bound here.
// ---------------------------------------------------------------------------------
#ifndef DECODE_BF16_PREFETCH
#define DECODE_BF16_PREFETCH 8                  // ring depth in k-steps (2 pieces, 2 MFMAs each)
#endif

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_bf16_kernel(const DecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int x = p.x0 + blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int y = p.y0 + blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.x1) && (y < p.y1);
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.y1 ? y : p.y1 - 1;
    int iy, ix;
    float relh, relw;
    axis_eval(p.ah, yc, iy, relh);
    axis_eval(p.aw, xc, ix, relw);
    const float* __restrict__ Wt = p.Wt;
    const float* __restrict__ Pc = p.P + (((size_t)b * p.Prows + (iy - p.Prow0)) * p.W + ix) * PCH + 4 * h;

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
        sq[g] = *(const f32x4*)(Wt + OFF_BQR + 4 * h + 8 * g);
    }
    float q3[128];                                               // fp32 copy of the last activation for the head
#pragma unroll 1
    for (int layer = 0; layer < 3; ++layer) {
        const int nl = layer < 2 ? layer + 1 : 2;
        const float* __restrict__ Pl = Pc + (layer + 1) * HID;
        const float* __restrict__ Bq = Wt + OFF_BQR + layer * HID + 4 * h;
        const float* __restrict__ Pn = Pc + (nl + 1) * HID;
        const float* __restrict__ Bn = Wt + OFF_BQR + nl * HID + 4 * h;
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
                    const float v = relu0(pk[ks]) * dsin_rev<SIN_MODE>(ps[ks]);
                    qn[2 * (m - 1) + (ks >> 3)][ks & 7] = (__bf16)v;
                    q3[16 * (m - 1) + ks] = v;
                }
            }
            pk = ak;
            ps = as;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = relu0(pk[r]) * dsin_rev<SIN_MODE>(ps[r]);
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
        const long long plane = p.o_ps;
        float* o = out_px(p, b, y, x);
        const unsigned nanm = derived_nan_mask(Wt);
        o[0] = o0 + or_bits(Wt[OFF_BL + 0], nanm);
        o[plane] = o1 + or_bits(Wt[OFF_BL + 1], nanm);
        o[2 * plane] = o2 + or_bits(Wt[OFF_BL + 2], nanm);
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
    const int x = p.x0 + blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int yb = p.y0 + blockIdx.y * (2 * TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    int y[2];
    bool valid[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        y[t] = yb + t * (TILE_H * WG_TILES_Y);
        valid[t] = (x < p.x1) && (y[t] < p.y1);
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
        Pc[t] = p.P + (((size_t)b * p.Prows + (iy - p.Prow0)) * p.W + ix) * PCH + 4 * h;
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
        sq[g] = *(const f32x4*)(Wt + OFF_BQR + 4 * h + 8 * g);
    }
    float o[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    bf16x8 (*mine)[16][64] = park[wave];

    // the layer loop is fully unrolled (a bf16 layer is 512 MFMAs): LAST is a compile-time constant per copy and
    // fuses the RGB head (diinn.py:138) into the epilogue, on the unrounded activation
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        const bool LAST = layer == 2;
        const int nl = layer < 2 ? layer + 1 : 2;
        const float* __restrict__ Bq = Wt + OFF_BQR + layer * HID + 4 * h;
        const float* __restrict__ Bn = Wt + OFF_BQR + nl * HID + 4 * h;
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
                        float v = relu0(pk[t][ks]) * dsin_rev<SIN_MODE>(ps[t][ks]);
                        // opaque use: the element stays behind its k-step.  Without it the scheduler gathers the
                        // epilogue of a tile into one clump and spills ~360 registers (r02; tests/test_kernel_resources.py)
                        asm volatile("" : "+v"(v));
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
                const float v = relu0(pk[t][r]) * dsin_rev<SIN_MODE>(ps[t][r]);
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
            const long long plane = p.o_ps;
            float* op = out_px(p, b, y[t], x);
            const unsigned nanm = derived_nan_mask(Wt);
            op[0] = o0 + or_bits(Wt[OFF_BL + 0], nanm);
            op[plane] = o1 + or_bits(Wt[OFF_BL + 1], nanm);
            op[2 * plane] = o2 + or_bits(Wt[OFF_BL + 2], nanm);
        }
    }
}

// ---------------------------------------------------------------------------------
// The COOPERATIVE bf16 decode (round 2): the waves of a workgroup work on one block of 4 pixel tiles (16 x 8 HR pixels)
// instead of each owning pixels of its own.  What follows describes the first, four-wave form of it -- the structure the
// eight-wave kernels below (decode_bf16_coop8_kernel, decode_bf16_coop8p_kernel) inherit; the four-wave kernel itself
// (decode_bf16_coop_kernel: ~390 registers, one wave per SIMD, 13.1 k cycles per layer against the 8.2 k of its MFMAs) had
// no dispatch path left and was deleted in round 5 (DESIGN_HISTORY.md section 4).  Wave w owned 64 of the 256
// channels of every layer (M-tiles 2w and 2w+1 of the modulation rows and of the synthesis rows, so
// relu(k) * sin(s) stays register-local) for ALL pixels of the block:
//   * weights: a wave reads only its own quarter of a layer (64 KiB) and keeps the 32 fragments of the current
//     M-tile in registers, where each feeds 4 MFMAs (one per pixel tile): a quarter of the single-tile kernel's
//     vector-memory instructions and L1 bytes per MFMA, with no sharing protocol between the waves;
//   * activations: the layer input of the whole block lives in LDS as ready-made B fragments
//     [tile][k-step][lane] x 16 B (written and read lane-linearly: conflict-free), one ds_read_b128 per two
//     MFMAs; input and output images are double-buffered (2 x 64 KiB) and the epilogue stores straight into the
//     other buffer;
//   * accumulator seeds: the slice P_{layer+1} of the block's LR cells (<= 24 cells, the launch checks the scale)
//     and the synthesis biases are staged in LDS once per layer by whole-row loads (1 KiB per cell per
//     instruction) instead of four scattered 16-byte loads per pixel tile and M-tile;
//   * the RGB head is accumulated in the last layer's epilogue (fp32, unrounded activation) and the four
//     waves' partial sums meet in LDS.
// At one wave per SIMD a wave can issue about 8 instructions per 32-cycle bf16 MFMA, the MFMA included
// (tools + PMC: DESIGN.md section 3.4), so the instruction stream between two MFMAs is placed by hand
// (sched_barrier): LDS read | MFMA | half an epilogue element | MFMA | the other half + refill loads.
// Same products and the same k-order of accumulation as decode_bf16_kernel.
// ---------------------------------------------------------------------------------
constexpr int CO_TILES = 4;                                   // 2 x 2 tiles of 8 x 4 pixels = 16 x 8 HR pixels
constexpr int CO_MT_BYTES = 16 * 2 * PIECE_BYTES;             // one M-tile of the WLB image: [ks 16][part 2] pieces
constexpr int CO_SEED_CELLS = 24;                             // LR cells of a block's footprint the seed slab holds
constexpr int CO_SEED_PITCH = HID + 4;                        // floats per staged cell row (+16 B: rotates the banks)
constexpr int CO_BRING = 4;                                   // LDS read-ahead of the B fragments, in k-steps
#define CO_SB() __builtin_amdgcn_sched_barrier(0)

struct CoopTagFalse { static constexpr bool value = false; };
struct CoopTagTrue { static constexpr bool value = true; };

// the sine of the synthesis branch on revolutions, split so that its two halves can sit in different MFMA gaps
template <int MODE> __device__ __forceinline__ float co_sin_prep(float x) { return x; }
template <> __device__ __forceinline__ float co_sin_prep<DIINN_SIN_HW_REDUCED>(float x) { return __builtin_amdgcn_fractf(x); }   // bf16: the fract's 6e-8 is noise
template <int MODE> __device__ __forceinline__ float co_sin_fin(float y) { return __builtin_amdgcn_sinf(y); }
template <> __device__ __forceinline__ float co_sin_fin<DIINN_SIN_ACCURATE>(float y) { return dsin_rev<DIINN_SIN_ACCURATE>(y); }

#ifdef DIINN_STAMPS
#define CO_STAMP(i)                                                                           \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (p.stamps && lane == 0) p.stamps[co_stamp_base + (i)] = t_;                        \
    } while (0)
#else
#define CO_STAMP(i) do {} while (0)
#endif

// ---------------------------------------------------------------------------------
// decode_bf16_coop8_kernel: the cooperative kernel with EIGHT waves per workgroup, two per SIMD.
// The four-wave form needed ~390 registers per wave, so each SIMD hosted one wave, and whenever that wave was
// blocked in a vector-memory instruction (~60 cycles each), evaluates its epilogue or runs the fp32 layer-0 prologue,
// the matrix pipe idles: a layer takes 13.1 k cycles against the 8.2 k of its MFMAs, the prologue 15 k.  Here wave w
// owns ONE M-tile (channels 32w .. 32w+31 of both branches) for the same 16 x 8 pixel block: 32 weight fragments
// (128 registers) stay put for the whole layer, a single accumulator pair serves the four pixel tiles one after the
// other, the epilogue of a tile runs right after its MFMAs -- and the other wave of the SIMD (w +- 4, which walks the
// tiles in the order 2,3,0,1 so that the two are never in the same phase) has the matrix pipe meanwhile.  Everything
// else -- B fragments in LDS, double-buffered images, staged seeds / biases / tables, revolutions -- is the coop
// kernel's.  ~230 registers, 512 threads.
// ---------------------------------------------------------------------------------
template <int SIN_MODE>
__global__ __launch_bounds__(512, 2) void decode_bf16_coop8_kernel(const DecodeParams p) {
    constexpr int TILES = CO_TILES;
    __shared__ __attribute__((aligned(16))) bf16x8 qa[2][TILES][16][64];       // 128 KiB: B fragments, double-buffered
    __shared__ __attribute__((aligned(16))) float seed[CO_SEED_CELLS * CO_SEED_PITCH];   // P slice of the block's cells
    __shared__ __attribute__((aligned(16))) float bias[3 * HID];               // bQ1..3 in revolutions
    __shared__ __attribute__((aligned(16))) float ltab[3 * HID];               // head rows L0..L2
    float (*red)[TILES * 32][3] = reinterpret_cast<float (*)[TILES * 32][3]>(seed);   // partial RGB per (wave, lane half)
    static_assert(sizeof(seed) >= 16 * TILES * 32 * 3 * sizeof(float), "red aliases the seed slab");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7 = this wave's M-tile
    const int grp = wave >> 2;                                    // waves w and w + 4 share a SIMD
    const int h = lane >> 5, j = lane & 31;
    // At non-integer scales neighbouring blocks stage the same P rows: give every XCD (= L2) a contiguous run of
    // blocks then (c5: -1.5 %).  Where blocks and cells line up there is nothing to share, and the plain round-robin
    // order is the faster one (c2: -3.6 % on this kernel, and the P kernel behind it finds its weights still in L2).
    const BlockXYZ blk = p.xcd_runs ? xcd_run_block() : BlockXYZ{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int b = blk.z;
    const float* __restrict__ Wt = p.Wt;
    const int ncx = p.seed_cols;
    const int ncx_inv = 65536 / ncx + 1;
#ifdef DIINN_STAMPS
    const size_t co_stamp_base = ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * 16;
#endif
    CO_STAMP(0);

    const int x0 = p.x0 + blk.x * (2 * TILE_W) + (j & (TILE_W - 1));
    const int yb = p.y0 + blk.y * (2 * TILE_H) + (j / TILE_W);
    int ixs[2], iys[2];
    float relws[2], relhs[2];
#pragma unroll
    for (int tx = 0; tx < 2; ++tx) {
        const int x = x0 + tx * TILE_W;
        axis_eval(p.aw, x < p.Wu ? x : p.Wu - 1, ixs[tx], relws[tx]);
    }
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
        const int y = yb + ty * TILE_H;
        axis_eval(p.ah, y < p.y1 ? y : p.y1 - 1, iys[ty], relhs[ty]);
    }
    const int ix0 = __builtin_amdgcn_readfirstlane(ixs[0]), iy0 = __builtin_amdgcn_readfirstlane(iys[0]);
    // this wave visits the tiles in the order tile(ti) = ti ^ (2 grp); per visit: byte offset of the pixel's staged P
    // row (+ this lane-half's 16 bytes) and of the tile's B fragments
    int srow[TILES], qoff[TILES];
#pragma unroll
    for (int ti = 0; ti < TILES; ++ti) {
        const int ta = ti, tb2 = ti ^ 2;                          // static candidates, selected by the wave's group
        const int sa = (((iys[ta >> 1] - iy0) * ncx + (ixs[ta & 1] - ix0)) * CO_SEED_PITCH + 4 * h) * (int)sizeof(float);
        const int sb = (((iys[tb2 >> 1] - iy0) * ncx + (ixs[tb2 & 1] - ix0)) * CO_SEED_PITCH + 4 * h) * (int)sizeof(float);
        srow[ti] = grp ? sb : sa;
        qoff[ti] = (grp ? tb2 : ta) * 16 * 64;                    // in bf16x8 elements
    }
    // cells this wave stages (slab rows wave, wave + 8, wave + 16)
    size_t scell[CO_SEED_CELLS / 8];
#pragma unroll
    for (int i = 0; i < CO_SEED_CELLS / 8; ++i) {
        const int c = wave + 8 * i;
        const int cq = (c * ncx_inv) >> 16;
        int cy = iy0 + cq, cx = ix0 + (c - cq * ncx);
        const int ylast = p.Prow0 + p.Prows - 1;
        cy = cy < ylast ? cy : ylast;
        cx = cx < p.W - 1 ? cx : p.W - 1;
        scell[i] = ((size_t)(b * p.Prows + (cy - p.Prow0)) * p.W + cx) * PCH + 4 * lane;
    }
    f32x4 st[CO_SEED_CELLS / 8];
    auto stage_load = [&](const int slice) {
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i) st[i] = *(const f32x4*)(p.P + scell[i] + slice * HID);
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i)
            *(f32x4*)(seed + (wave + 8 * i) * CO_SEED_PITCH + 4 * lane) = st[i];
    };

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    auto ld_w = [&](const int mt, const int pc) {
        return ld_piece(wrs, lane_off + (pc & 3) * PIECE_BYTES, mt + (pc >> 2) * 4 * PIECE_BYTES);
    };
    // this wave's M-tile of the current layer: [k-step] x {modulation, synthesis}, refilled with the next layer's
    // fragment as soon as the last pixel tile has consumed it
    f32x4 Ak[16], As[16];
    int wp = (int)(OFF_WLB * sizeof(float)) + wave * CO_MT_BYTES;

    // ---- prologue: tables and the P_0 / P_1 slices into LDS (the tables of layer 0 live in the second image)
    float* const q0tab = reinterpret_cast<float*>(&qa[1][0][0][0]);          // [3][256]: Q0h, Q0w, fma(Q0r, ratio, bQ0)
    float* const seed0 = q0tab + 4 * HID;
    {
        f32x4 s0[CO_SEED_CELLS / 8];
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i) s0[i] = *(const f32x4*)(p.P + scell[i]);
        stage_load(1);
        const int ti = threadIdx.x;
        f32x4 tq = {}, bq0 = {};
        if (ti < 192) tq = *(const f32x4*)(Wt + OFF_Q0R + 4 * ti);          // rows Q0h, Q0w, Q0r (64 threads each)
        if (ti >= 128 && ti < 192) bq0 = *(const f32x4*)(Wt + OFF_Q0R + HID + 4 * ti);
        // the first M-tile's 32 weight fragments (256 KiB per workgroup: ~4 k cycles of L1 time) are requested AFTER
        // the few rows layer 0 waits for, and arrive while layer 0 is computed
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Ak[ks] = ld_w(wp, 2 * ks + 0);
            As[ks] = ld_w(wp, 2 * ks + 1);
        }
        asm volatile("" ::: "memory");
        if (ti < 192) {
            if (ti >= 128) {                                                 // row 2 becomes t = fma(Q0r, ratio, bQ0)
#pragma unroll
                for (int e = 0; e < 4; ++e) tq[e] = __builtin_fmaf(tq[e], p.ratio, bq0[e]);
            }
            *(f32x4*)(q0tab + 4 * ti) = tq;
        } else if (ti < 384) {
            *(f32x4*)(bias + 4 * (ti - 192)) = *(const f32x4*)(Wt + OFF_BQR + 4 * (ti - 192));
        } else if (ti < 384 + 128) {
            const int k = ti - 384;                                          // 128 threads: 192 pieces of L, two rounds
            *(f32x4*)(ltab + 4 * k) = *(const f32x4*)(Wt + OFF_L + 4 * k);
            if (k < 64) *(f32x4*)(ltab + 4 * (k + 128)) = *(const f32x4*)(Wt + OFF_L + 4 * (k + 128));
        }
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i)
            *(f32x4*)(seed0 + (wave + 8 * i) * CO_SEED_PITCH + 4 * lane) = s0[i];
        stage_store();
    }
    const unsigned nanm = derived_nan_mask(Wt);
    const float bl0 = or_bits(Wt[OFF_BL + 0], nanm), bl1 = or_bits(Wt[OFF_BL + 1], nanm), bl2 = or_bits(Wt[OFF_BL + 2], nanm);
    CO_STAMP(14);
    __syncthreads();
    CO_STAMP(15);

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    // ---- layer 0 (fp32): wave w evaluates tile w & 3, channels 128 grp .. 128 grp + 127 (k-steps 8 grp .. 8 grp + 7)
    {
        const float* __restrict__ Q0 = q0tab + 4 * h;
        const int t = wave & 3;
        const int ix = (t & 1) ? ixs[1] : ixs[0], iy = (t >> 1) ? iys[1] : iys[0];
        const float relw = (t & 1) ? relws[1] : relws[0], relh = (t >> 1) ? relhs[1] : relhs[0];
        const float* __restrict__ Pc = seed0 + ((iy - iy0) * ncx + (ix - ix0)) * CO_SEED_PITCH + 4 * h;
        const int g0 = 16 * grp;                                 // first group of 4 channels (c0 = 8 (g0 + i))
        f32x4 cpv, cwh, cww, ctq, npv, nwh, nww, ntq;
        auto fetch = [&](const int i, f32x4& pv, f32x4& wh, f32x4& ww, f32x4& tq) {
            const int c0 = 8 * (g0 + i);
            pv = *(const f32x4*)(Pc + c0);
            wh = *(const f32x4*)(Q0 + 0 * HID + c0);
            ww = *(const f32x4*)(Q0 + 1 * HID + c0);
            tq = *(const f32x4*)(Q0 + 2 * HID + c0);
        };
        fetch(0, cpv, cwh, cww, ctq);
        u32x4 fragw;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i + 1 < 16) fetch(i + 1, npv, nwh, nww, ntq);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = __builtin_fmaf(cww[e], relw, ctq[e]);
                a = __builtin_fmaf(cwh[e], relh, a);
                v[e] = relu0(cpv[e]) * co_sin_fin<SIN_MODE>(co_sin_prep<SIN_MODE>(a));   // as the epilogues: v_fract + v_sin on revolutions
            }
            const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
            fragw[2 * (i & 1) + 0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
            fragw[2 * (i & 1) + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
            if (i & 1) qa[0][t][8 * grp + (i >> 1)][lane] = __builtin_bit_cast(bf16x8, fragw);
            cpv = npv; cwh = nwh; cww = nww; ctq = ntq;
            asm volatile("" ::: "memory");
        }
    }
    float o[TILES][3];
    CO_STAMP(1);
    __syncthreads();
    CO_STAMP(2);

    auto layer_body = [&](auto last_tag, auto cur_tag, const int layer) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int CUR = decltype(cur_tag)::value ? 1 : 0;
        constexpr int NXT = 1 - CUR;
        const bf16x8* __restrict__ qin = &qa[CUR][0][0][lane];
        const float* __restrict__ bl = bias + layer * HID + 32 * wave + 4 * h;
        const float* __restrict__ hl = ltab + 32 * wave + 4 * h;
        if (LAST) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) o[t][0] = o[t][1] = o[t][2] = 0.0f;
        }
        bf16x8 bq[CO_BRING];
#pragma unroll
        for (int i = 0; i < CO_BRING; ++i) bq[i] = qin[qoff[0] + i * 64];
#pragma unroll
        for (int ti = 0; ti < TILES; ++ti) {
            f32x16 ak, as;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const f32x4 sk = *(const f32x4*)((const char*)seed + srow[ti] + (32 * wave + 8 * gg) * (int)sizeof(float));
                const f32x4 sq = *(const f32x4*)(bl + 8 * gg);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ak[4 * gg + e] = sk[e];
                    as[4 * gg + e] = sq[e];
                }
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const bf16x8 bv = bq[ks % CO_BRING];
                ak = MFMA_BF16(__builtin_bit_cast(bf16x8, Ak[ks]), bv, ak);
                as = MFMA_BF16(__builtin_bit_cast(bf16x8, As[ks]), bv, as);
                if (ks + CO_BRING < 16) bq[ks % CO_BRING] = qin[qoff[ti] + (ks + CO_BRING) * 64];
                else if (ti + 1 < TILES) bq[ks % CO_BRING] = qin[qoff[ti + 1] + (ks + CO_BRING - 16) * 64];
                if (ti == TILES - 1 && !LAST) {                   // last use of this fragment: fetch the next layer's
                    const int nwp = wp + (int)(WLB_LAYER * sizeof(float));
                    Ak[ks] = ld_w(nwp, 2 * ks + 0);
                    As[ks] = ld_w(nwp, 2 * ks + 1);
                }
                if (!LAST && ti == 1 && ks == 8) stage_load(layer + 2);   // next layer's P slice, into registers
                if ((ks & 3) == 3) asm volatile("" ::: "memory");
            }
            // epilogue of this tile (the other wave of the SIMD has the matrix pipe): q = relu(k) * sin(s)
            u32x4 fragw;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    v[i] = relu0(ak[r + i]) * co_sin_fin<SIN_MODE>(co_sin_prep<SIN_MODE>(as[r + i]));
                if (LAST) {                                       // head rows of this element's channel, from the LDS table
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int c = 8 * ((r + i) >> 2) + ((r + i) & 3);
                        o[ti][0] = __builtin_fmaf(hl[0 * HID + c], v[i], o[ti][0]);
                        o[ti][1] = __builtin_fmaf(hl[1 * HID + c], v[i], o[ti][1]);
                        o[ti][2] = __builtin_fmaf(hl[2 * HID + c], v[i], o[ti][2]);
                    }
                } else {
                    fragw[(r >> 1) & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                    if ((r & 7) == 6)
                        *(bf16x8*)(&qa[NXT][0][0][lane] + qoff[ti] + (2 * wave + (r >> 3)) * 64) = __builtin_bit_cast(bf16x8, fragw);
                }
            }
            if (LAST) asm volatile("" : "+v"(o[ti][0]), "+v"(o[ti][1]), "+v"(o[ti][2]));
        }
        CO_STAMP(3 + 3 * (LAST ? 2 : CUR));
        __syncthreads();                                          // layer output complete; input and seed slab are free
        CO_STAMP(4 + 3 * (LAST ? 2 : CUR));
        if (!LAST) {
            stage_store();
            __syncthreads();
        }
        CO_STAMP(5 + 3 * (LAST ? 2 : CUR));
    };

    layer_body(CoopTagFalse{}, CoopTagFalse{}, 0);
    wp += (int)(WLB_LAYER * sizeof(float));
    layer_body(CoopTagFalse{}, CoopTagTrue{}, 1);
    wp += (int)(WLB_LAYER * sizeof(float));
    layer_body(CoopTagTrue{}, CoopTagFalse{}, 2);

    // ---- head: the 16 partial sums of a pixel (8 waves x 2 lane halves) meet in LDS (diinn.py:138).
    // o[ti] belongs to tile ti ^ (2 grp)
#pragma unroll
    for (int ti = 0; ti < TILES; ++ti) {
        const int t = ti ^ (2 * grp);
#pragma unroll
        for (int k = 0; k < 3; ++k) red[2 * wave + h][t * 32 + j][k] = o[ti][k];
    }
    __syncthreads();
    if (threadIdx.x < TILES * 32) {
        const int t = threadIdx.x >> 5, jj = threadIdx.x & 31;
        const int x = p.x0 + blk.x * (2 * TILE_W) + (t & 1) * TILE_W + (jj & (TILE_W - 1));
        const int y = p.y0 + blk.y * (2 * TILE_H) + (t >> 1) * TILE_H + (jj / TILE_W);
        float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int w16 = 0; w16 < 16; ++w16)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[k] += red[w16][threadIdx.x][k];
        if (x < p.x1 && y < p.y1) {
            const long long plane = p.o_ps;
            float* op = out_px(p, b, y, x);
            op[0] = acc[0] + bl0;
            op[plane] = acc[1] + bl1;
            op[2 * plane] = acc[2] + bl2;
        }
    }
    CO_STAMP(12);
}

// ---------------------------------------------------------------------------------
// decode_bf16_coop8p_kernel: decode_bf16_coop8_kernel with PERSISTENT workgroups (r03).  Stamps of the one-block-per-
// workgroup form (profiles/r02_bf16_coop8_stamps.txt) put a quarter of a wave's life into the prologue, and more than
// half of that into waiting for loads: the first 256 KiB of weights alone are 4 k cycles of L1 time, requested by a
// workgroup that has nothing else to do yet.  Here 256 workgroups (8 XCDs x 32) each walk a list of blocks, and
// everything the NEXT block's start needs is requested inside the LAST layer of the current one, where it hides
// behind MFMAs: the first layer's weight fragments (refilled as the last pixel tile consumes the old ones, exactly
// like the refill between layers), the P_0 / P_1 rows of the next block's cells, the Q0 table.  The second
// activation image, free during the last layer, receives the next block's layer-0 tables meanwhile.  bias / head
// tables are staged once per workgroup.
// Block order: the 32 workgroups of an XCD (= one L2) work on one SUPER-TILE of 8 x 4 neighbouring blocks at a
// time and walk a contiguous run of super-tiles.  At non-integer scales neighbouring blocks stage the same P rows
// (c5, x3.3: a block's footprint is ~5 x 3.4 cells, each cell is wanted by up to 4 blocks); with the plain dispatch
// order those blocks ran in 8 different L2s (measured 6.28 GB of HBM traffic against 3.9 GB algorithmic).
// Arithmetic, k-order and results are decode_bf16_coop8_kernel's, bit for bit.
// ---------------------------------------------------------------------------------
constexpr int CO_ST_X = 8, CO_ST_Y = 4;                      // blocks per super-tile (32 = workgroups per XCD)
constexpr int CO_PGRID = 8 * CO_ST_X * CO_ST_Y;              // persistent workgroups of a launch: 8 XCDs x 32 CUs.  The workgroups wait
                                                             // for nobody, so the grid is right on any device (a partitioned one
                                                             // queues them); the super-tile walk is TUNED for this geometry

template <int SIN_MODE>
__global__ __launch_bounds__(512, 2) void decode_bf16_coop8p_kernel(const DecodeParams p) {
    constexpr int TILES = CO_TILES;
    __shared__ __attribute__((aligned(16))) bf16x8 qa[2][TILES][16][64];       // 128 KiB: B fragments, double-buffered
    __shared__ __attribute__((aligned(16))) float seed[CO_SEED_CELLS * CO_SEED_PITCH];   // P slice of the block's cells
    __shared__ __attribute__((aligned(16))) float bias[3 * HID];               // bQ1..3 in revolutions
    __shared__ __attribute__((aligned(16))) float ltab[3 * HID];               // head rows L0..L2
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7 = this wave's M-tile
    const int grp = wave >> 2;                                    // waves w and w + 4 share a SIMD
    const float* __restrict__ Wt = p.Wt;
    const int ncx = p.seed_cols;
    const int ncx_inv = 65536 / ncx + 1;

    // ---- this workgroup's blocks: position (dx, dy) inside every super-tile of its XCD's run
    const int gx = p.pg[0], gy = p.pg[1], stx = p.pg[3], sty = p.pg[4], per = p.pg[5];
    const int nst = stx * sty * p.pg[2];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int dx = slot & (CO_ST_X - 1), dy = slot / CO_ST_X;
    int stile = xcd * per;
    const int st_end = stile + per < nst ? stile + per : nst;
    struct Blk { int x, y, z; };
    auto locate = [&](const int st_, Blk& bk) {                   // block (dx, dy) of super-tile st_; false: outside the grid
        const int z = st_ / (stx * sty), r = st_ - z * stx * sty;
        const int sy = r / stx, sx = r - sy * stx;
        bk.x = sx * CO_ST_X + dx; bk.y = sy * CO_ST_Y + dy; bk.z = z;
        return bk.x < gx && bk.y < gy;
    };
    auto next_block = [&](int& st_, Blk& bk) {                    // first super-tile >= st_ that holds a block for us
        while (st_ < st_end && !locate(st_, bk)) ++st_;
        return st_ < st_end;
    };
    Blk blk;
    if (!next_block(stile, blk)) return;                          // uniform over the workgroup; no barrier yet

    // ---- per-block coordinates
    int ixs[2], iys[2];
    float relws[2], relhs[2];
    int ix0, iy0;
    int srow[TILES], qoff[TILES];
#pragma unroll
    for (int ti = 0; ti < TILES; ++ti) qoff[ti] = (grp ? (ti ^ 2) : ti) * 16 * 64;   // tile order of this wave's group
    auto coords = [&](const Blk& bk) {
        int ln = lane;                                            // opaque: see stage_load
        asm volatile("" : "+v"(ln));
        const int j = ln & 31, h = ln >> 5;
        const int x0 = p.x0 + bk.x * (2 * TILE_W) + (j & (TILE_W - 1));
        const int yb = p.y0 + bk.y * (2 * TILE_H) + (j / TILE_W);
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
            const int x = x0 + tx * TILE_W;
            axis_eval(p.aw, x < p.Wu ? x : p.Wu - 1, ixs[tx], relws[tx]);
        }
#pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
            const int y = yb + ty * TILE_H;
            axis_eval(p.ah, y < p.y1 ? y : p.y1 - 1, iys[ty], relhs[ty]);
        }
        ix0 = __builtin_amdgcn_readfirstlane(ixs[0]);
        iy0 = __builtin_amdgcn_readfirstlane(iys[0]);
#pragma unroll
        for (int ti = 0; ti < TILES; ++ti) {
            const int ta = ti, tb2 = ti ^ 2;
            const int sa = (((iys[ta >> 1] - iy0) * ncx + (ixs[ta & 1] - ix0)) * CO_SEED_PITCH + 4 * h) * (int)sizeof(float);
            const int sb = (((iys[tb2 >> 1] - iy0) * ncx + (ixs[tb2 & 1] - ix0)) * CO_SEED_PITCH + 4 * h) * (int)sizeof(float);
            srow[ti] = grp ? sb : sa;
        }
    };
    // first LR cell of a block (wave-uniform): the lane-0 pixel's indices
    auto first_cell = [&](const Blk& bk, int& cx0, int& cy0) {
        const int x = p.x0 + bk.x * (2 * TILE_W), y = p.y0 + bk.y * (2 * TILE_H);
        int a, b2;
        float rel;
        axis_eval(p.aw, x < p.Wu ? x : p.Wu - 1, a, rel);
        axis_eval(p.ah, y < p.y1 ? y : p.y1 - 1, b2, rel);
        cx0 = __builtin_amdgcn_readfirstlane(a);
        cy0 = __builtin_amdgcn_readfirstlane(b2);
    };
    // Staging of the block's P rows (slab cell c = 0..23 of the footprint whose first cell is (cx0, cy0) of batch item z):
    //  * P_0, read by layer 0 across all channels: whole 1 KiB rows, cells wave, wave + 8, wave + 16 per wave
    //    (scell: wave-uniform element offsets, the lane adds its 16 bytes);
    //  * P_1..3, the accumulator seeds of layers 1..3: wave w only ever reads ITS 32 channels of every cell, so it stages
    //    exactly that column itself -- cell (lane >> 3) + 8 i, 16-byte chunk lane & 7: one 128-byte line per cell -- and
    //    writes it once its own last read of the old column is behind it.  No other wave touches the column: no
    //    barrier for the seeds, one barrier per layer instead of two (ccell: per-lane cell numbers in the P window).
    size_t scell[CO_SEED_CELLS / 8];
    auto cells_of = [&](const int cx0, const int cy0, const int z) {
        const int ylast = p.Prow0 + p.Prows - 1;
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i) {
            const int c = wave + 8 * i;
            const int cq = (c * ncx_inv) >> 16;
            int cy = cy0 + cq, cx = cx0 + (c - cq * ncx);
            cy = cy < ylast ? cy : ylast;
            cx = cx < p.W - 1 ? cx : p.W - 1;
            scell[i] = ((size_t)(z * p.Prows + (cy - p.Prow0)) * p.W + cx) * PCH;
        }
    };
    f32x4 st[CO_SEED_CELLS / 8];
    // (the lane number enters through an opaque copy and the per-lane cell arithmetic -- a dozen operations per cell --
    // is redone at every staging: held in registers across the block loop it would not fit)
    auto stage_load = [&](const int slice, const int cx0, const int cy0, const int z) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int ccol = 32 * wave + 4 * (ln & 7);               // this lane's 4 channels inside the wave's column
        const int ylast = p.Prow0 + p.Prows - 1;
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i) {
            const int c2 = (ln >> 3) + 8 * i;
            const int cq2 = (c2 * ncx_inv) >> 16;
            int cy2 = cy0 + cq2, cx2 = cx0 + (c2 - cq2 * ncx);
            cy2 = cy2 < ylast ? cy2 : ylast;
            cx2 = cx2 < p.W - 1 ? cx2 : p.W - 1;
            const size_t cell = (size_t)(z * p.Prows + (cy2 - p.Prow0)) * p.W + cx2;
            st[i] = *(const f32x4*)(p.P + cell * PCH + slice * HID + ccol);
        }
    };
    auto stage_store = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        float* dst = seed + (ln >> 3) * CO_SEED_PITCH + 32 * wave + 4 * (ln & 7);
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i) *(f32x4*)(dst + 8 * i * CO_SEED_PITCH) = st[i];
    };

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    auto ld_w = [&](const int mt, const int pc) {
        return ld_piece(wrs, lane_off + (pc & 3) * PIECE_BYTES, mt + (pc >> 2) * 4 * PIECE_BYTES);
    };
    f32x4 Ak[16], As[16];
    const int wp0 = (int)(OFF_WLB * sizeof(float)) + wave * CO_MT_BYTES;   // this wave's M-tile of the first stacked layer
    int wp = wp0;

    // the tables of layer 0 live in the second activation image: free until layer 1's epilogue starts writing it, and
    // free again during the whole last layer (whose results go to the head, not to an image)
    float* const q0tab = reinterpret_cast<float*>(&qa[1][0][0][0]);          // [3][256]: Q0h, Q0w, fma(Q0r, ratio, bQ0)
    float* const seed0 = q0tab + 3 * HID;                                     // P_0 rows of the block's cells
    // partial RGB sums of the head per (wave, pixel), written tile by tile during the last layer (12 registers fewer
    // than carrying all four tiles' sums to the end)
    float (*red)[TILES * 32][3] = reinterpret_cast<float (*)[TILES * 32][3]>(seed0 + CO_SEED_CELLS * CO_SEED_PITCH);
    static_assert((3 * HID + CO_SEED_CELLS * CO_SEED_PITCH + 8 * TILES * 32 * 3) * sizeof(float) <= sizeof(qa) / 2,
                  "layer-0 tables and the head sums fit in qa[1]");
    const int tix = threadIdx.x;
    // the Q0 rows of thread tix (192 threads x 16 B), row 2 folded with the ratio: identical for every block
    auto q0_load = [&](const int tx, f32x4& tq, f32x4& bq0) {
        tq = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        bq0 = tq;
        if (tx < 192) tq = *(const f32x4*)(Wt + OFF_Q0R + 4 * tx);
        if (tx >= 128 && tx < 192) bq0 = *(const f32x4*)(Wt + OFF_Q0R + HID + 4 * tx);
    };
    auto q0_store = [&](const int tx, f32x4 tq, const f32x4 bq0) {
        if (tx < 192) {
            if (tx >= 128) {
#pragma unroll
                for (int e = 0; e < 4; ++e) tq[e] = __builtin_fmaf(tq[e], p.ratio, bq0[e]);
            }
            *(f32x4*)(q0tab + 4 * tx) = tq;
        }
    };

    // ---- prologue of the FIRST block (later blocks are prepared inside the previous block's last layer)
    {
        int cx0, cy0;
        first_cell(blk, cx0, cy0);
        cells_of(cx0, cy0, blk.z);
        f32x4 s0[CO_SEED_CELLS / 8];
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i) s0[i] = *(const f32x4*)(p.P + scell[i] + 4 * lane);
        stage_load(1, cx0, cy0, blk.z);
        f32x4 tq, bq0;
        q0_load(tix, tq, bq0);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Ak[ks] = ld_w(wp, 2 * ks + 0);
            As[ks] = ld_w(wp, 2 * ks + 1);
        }
        asm volatile("" ::: "memory");
        q0_store(tix, tq, bq0);
        if (tix >= 192 && tix < 384) {
            *(f32x4*)(bias + 4 * (tix - 192)) = *(const f32x4*)(Wt + OFF_BQR + 4 * (tix - 192));
        } else if (tix >= 384) {
            const int k = tix - 384;                                         // 128 threads: 192 pieces of L, two rounds
            *(f32x4*)(ltab + 4 * k) = *(const f32x4*)(Wt + OFF_L + 4 * k);
            if (k < 64) *(f32x4*)(ltab + 4 * (k + 128)) = *(const f32x4*)(Wt + OFF_L + 4 * (k + 128));
        }
#pragma unroll
        for (int i = 0; i < CO_SEED_CELLS / 8; ++i)
            *(f32x4*)(seed0 + (wave + 8 * i) * CO_SEED_PITCH + 4 * lane) = s0[i];
        stage_store();
    }

    __syncthreads();                                              // the first block's tables and P_0 / P_1 slices are in LDS
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bool has_next = false;
    Blk nblk = blk;
    int ncx0 = 0, ncy0 = 0;                                       // first cell of the next block

    auto layer_body = [&](auto last_tag, auto cur_tag, const int layer) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int CUR = decltype(cur_tag)::value ? 1 : 0;
        constexpr int NXT = 1 - CUR;
        int ln = lane, tx = tix;                                  // opaque copies: see stage_load
        asm volatile("" : "+v"(ln), "+v"(tx));
        const int h = ln >> 5, j = ln & 31;
        const bf16x8* __restrict__ qin = &qa[CUR][0][0][ln];
        const float* __restrict__ bl = bias + layer * HID + 32 * wave + 4 * h;
        const float* __restrict__ hl = ltab + 32 * wave + 4 * h;
        // LAST: what the next block's start reads is requested here and parked at the end of the tile: tile 0: P_0 rows
        // 0, 1 and tile 1: P_0 row 2 into the free second image; tile 2: the Q0 table, likewise; tile 3: this wave's
        // column of the P_1 slice, into the seed slab (its last read of the old column is the tile's own seed), and,
        // fragment by fragment, the first layer's weights
        static_assert(CO_SEED_CELLS / 8 == 3, "prefetch schedule of the last layer");
        f32x4 na, nb;
        auto nrow = [&](const int r) -> const float* { return p.P + scell[r] + 4 * ln; };
        auto nslot = [&](const int r) -> float* { return seed0 + (wave + 8 * r) * CO_SEED_PITCH + 4 * ln; };
        bf16x8 bq[CO_BRING];
#pragma unroll
        for (int i = 0; i < CO_BRING; ++i) bq[i] = qin[qoff[0] + i * 64];
#pragma unroll
        for (int ti = 0; ti < TILES; ++ti) {
            f32x16 ak, as;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const f32x4 sk = *(const f32x4*)((const char*)seed + srow[ti] + (32 * wave + 8 * gg) * (int)sizeof(float));
                const f32x4 sq = *(const f32x4*)(bl + 8 * gg);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ak[4 * gg + e] = sk[e];
                    as[4 * gg + e] = sq[e];
                }
            }
            if (ti == TILES - 1) {                                // that was this wave's last read of its seed column
                asm volatile("" ::: "memory");
                if (!LAST) stage_store();                         // the next layer's column (requested during tile 1)
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const bf16x8 bv = bq[ks % CO_BRING];
                ak = MFMA_BF16(__builtin_bit_cast(bf16x8, Ak[ks]), bv, ak);
                as = MFMA_BF16(__builtin_bit_cast(bf16x8, As[ks]), bv, as);
                if (ks + CO_BRING < 16) bq[ks % CO_BRING] = qin[qoff[ti] + (ks + CO_BRING) * 64];
                else if (ti + 1 < TILES) bq[ks % CO_BRING] = qin[qoff[ti + 1] + (ks + CO_BRING - 16) * 64];
                if (ti == TILES - 1) {                            // last use of this fragment: fetch the next layer's --
                    const int nwp = LAST ? wp0 : wp + (int)(WLB_LAYER * sizeof(float));   // or the next block's first layer
                    if (!LAST || has_next) {
                        Ak[ks] = ld_w(nwp, 2 * ks + 0);
                        As[ks] = ld_w(nwp, 2 * ks + 1);
                    }
                }
                if (!LAST && ti == 1 && ks == 8) stage_load(layer + 2, ix0, iy0, blk.z);   // next layer's seed column, into registers
                if (LAST && ks == 0 && has_next) {
                    if (ti == 0) {
                        na = *(const f32x4*)nrow(0);
                        nb = *(const f32x4*)nrow(1);
                    } else if (ti == 1) {
                        na = *(const f32x4*)nrow(2);
                    } else if (ti == 2) {
                        q0_load(tx, na, nb);
                    } else {
                        stage_load(1, ncx0, ncy0, nblk.z);        // the next block's P_1 column
                    }
                }
                if ((ks & 3) == 3) asm volatile("" ::: "memory");
            }
            // epilogue of this tile (the other wave of the SIMD has the matrix pipe): q = relu(k) * sin(s)
            u32x4 fragw;
            float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    v[i] = relu0(ak[r + i]) * co_sin_fin<SIN_MODE>(co_sin_prep<SIN_MODE>(as[r + i]));
                if (LAST) {                                       // head rows of this element's channel, from the LDS table
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int c = 8 * ((r + i) >> 2) + ((r + i) & 3);
                        o0 = __builtin_fmaf(hl[0 * HID + c], v[i], o0);
                        o1 = __builtin_fmaf(hl[1 * HID + c], v[i], o1);
                        o2 = __builtin_fmaf(hl[2 * HID + c], v[i], o2);
                    }
                } else {
                    fragw[(r >> 1) & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                    if ((r & 7) == 6)
                        *(bf16x8*)(&qa[NXT][0][0][ln] + qoff[ti] + (2 * wave + (r >> 3)) * 64) = __builtin_bit_cast(bf16x8, fragw);
                }
            }
            if (LAST) {                                           // this wave's 32 channels of the head, both lane halves
                asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2));
                o0 += __shfl_xor(o0, 32);
                o1 += __shfl_xor(o1, 32);
                o2 += __shfl_xor(o2, 32);
                if (h == 0) {
                    float* r3 = red[wave][(ti ^ (2 * grp)) * 32 + j];    // visit ti of this wave's group is tile ti ^ (2 grp)
                    r3[0] = o0; r3[1] = o1; r3[2] = o2;
                }
            }
            if (LAST && has_next) {                               // ... parked
                if (ti == 0) {
                    *(f32x4*)nslot(0) = na;
                    *(f32x4*)nslot(1) = nb;
                } else if (ti == 1) {
                    *(f32x4*)nslot(2) = na;
                } else if (ti == 2) {
                    q0_store(tx, na, nb);
                } else {
                    stage_store();
                }
            }
        }
        __syncthreads();                                          // layer output complete, its input image is free
    };

    for (;;) {
        coords(blk);
        // (no barrier here: the tables and P_0 / P_1 slices of this block were parked inside the previous block's last layer,
        // whose closing barrier made them visible -- the first block's prologue is followed by a barrier of its own; nothing
        // writes the second image again before the barrier behind layer 0)
        // ---- layer 0 (fp32): wave w evaluates tile w & 3, channels 128 grp .. 128 grp + 127 (k-steps 8 grp .. 8 grp + 7)
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int h = ln >> 5;
            const float* __restrict__ Q0 = q0tab + 4 * h;
            const int t = wave & 3;
            const int ix = (t & 1) ? ixs[1] : ixs[0], iy = (t >> 1) ? iys[1] : iys[0];
            const float relw = (t & 1) ? relws[1] : relws[0], relh = (t >> 1) ? relhs[1] : relhs[0];
            const float* __restrict__ Pc = seed0 + ((iy - iy0) * ncx + (ix - ix0)) * CO_SEED_PITCH + 4 * h;
            const int g0 = 16 * grp;
            f32x4 cpv, cwh, cww, ctq, npv, nwh, nww, ntq2;
            auto fetch = [&](const int i, f32x4& pv, f32x4& wh, f32x4& ww, f32x4& tq) {
                const int c0 = 8 * (g0 + i);
                pv = *(const f32x4*)(Pc + c0);
                wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                tq = *(const f32x4*)(Q0 + 2 * HID + c0);
            };
            fetch(0, cpv, cwh, cww, ctq);
            u32x4 fragw;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + 1 < 16) fetch(i + 1, npv, nwh, nww, ntq2);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = __builtin_fmaf(cww[e], relw, ctq[e]);
                    a = __builtin_fmaf(cwh[e], relh, a);
                    v[e] = relu0(cpv[e]) * co_sin_fin<SIN_MODE>(co_sin_prep<SIN_MODE>(a));   // as the epilogues: v_fract + v_sin on revolutions
                }
                const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
                fragw[2 * (i & 1) + 0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
                fragw[2 * (i & 1) + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
                if (i & 1) qa[0][t][8 * grp + (i >> 1)][ln] = __builtin_bit_cast(bf16x8, fragw);
                cpv = npv; cwh = nwh; cww = nww; ctq = ntq2;
                asm volatile("" ::: "memory");
            }
        }
        // the block after this one (wave-uniform scalar work, overlaps the barrier)
        int nst_i = stile + 1;
        has_next = next_block(nst_i, nblk);
        __syncthreads();

        wp = wp0;
        layer_body(CoopTagFalse{}, CoopTagFalse{}, 0);
        wp += (int)(WLB_LAYER * sizeof(float));
        layer_body(CoopTagFalse{}, CoopTagTrue{}, 1);
        wp += (int)(WLB_LAYER * sizeof(float));
        if (has_next) {                                           // P rows of the next block's cells: read by the last layer's prefetch
            first_cell(nblk, ncx0, ncy0);
            cells_of(ncx0, ncy0, nblk.z);
        }
        layer_body(CoopTagTrue{}, CoopTagFalse{}, 2);

        // ---- head: the 8 partial sums of a pixel (one per wave) meet here (diinn.py:138); the last layer's closing
        // barrier made them visible
        int tid = threadIdx.x;                                    // opaque: keeps the pixel arithmetic inside the loop
        asm volatile("" : "+v"(tid));
        if (tid < TILES * 32) {
            const int t = tid >> 5, jj = tid & 31;
            const int x = p.x0 + blk.x * (2 * TILE_W) + (t & 1) * TILE_W + (jj & (TILE_W - 1));
            const int y = p.y0 + blk.y * (2 * TILE_H) + (t >> 1) * TILE_H + (jj / TILE_W);
            float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8)
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k] += red[w8][tid][k];
            if (x < p.x1 && y < p.y1) {
                const unsigned nanm = derived_nan_mask(Wt);      // (read per block: three scalar loads instead of registers held across the loop)
                const float bl0 = or_bits(Wt[OFF_BL + 0], nanm), bl1 = or_bits(Wt[OFF_BL + 1], nanm), bl2 = or_bits(Wt[OFF_BL + 2], nanm);
                const long long plane = p.o_ps;
                float* op = out_px(p, blk.z, y, x);
                op[0] = acc[0] + bl0;
                op[plane] = acc[1] + bl1;
                op[2 * plane] = acc[2] + bl2;
            }
        }
        // (r05: the head spread over six waves -- one (pixel, colour) per thread -- changed nothing: 7.69 vs 7.76 ms at c5)
        if (!has_next) break;
        blk = nblk;
        stile = nst_i;
    }
}

// cells of the LR footprint of the widest 16 x 8 pixel block: (columns, rows).  The maximum over EVERY start column / row
// of the FULL image, not over the blocks of the launch at hand (they are anchored at its x0 / y0), so that the kernel
// choice below is the same for a row band, for a column tile and for the whole image.
static int axis_footprint(const Axis& a, int n_out, int span) {
    int m = 1, lo, hi;
    float rel;
    for (int v = 0; v < n_out; ++v) {
        const int vl = v + span - 1 < n_out ? v + span - 1 : n_out - 1;
        axis_eval(a, v, lo, rel);
        axis_eval(a, vl, hi, rel);
        m = hi - lo + 1 > m ? hi - lo + 1 : m;
    }
    return m;
}

static void coop_footprint(const DecodeParams& p, int& ncx, int& ncy) {
    struct Key { int n_in, n_out, small, val; };
    static thread_local Key kx = {-1, -1, -1, 1}, ky = {-1, -1, -1, 1};
    if (kx.n_in != p.W || kx.n_out != p.Wu || kx.small != p.aw.small_output)
        kx = Key{p.W, p.Wu, p.aw.small_output, axis_footprint(p.aw, p.Wu, 2 * TILE_W)};
    if (ky.n_in != p.H || ky.n_out != p.Hu || ky.small != p.ah.small_output)
        ky = Key{p.H, p.Hu, p.ah.small_output, axis_footprint(p.ah, p.Hu, 2 * TILE_H)};
    ncx = kx.val;
    ncy = ky.val;
}

int launch_decode_bf16(void* stream, const DecodeParams& p, int gx, int gy, int gz, int sin_mode) {
    const int y0 = p.y0, y1 = p.y1, blk = 256;
    const dim3 grid(gx, gy, gz);

        // two pixel tiles per wave (half the weight bytes per MFMA) once the launch still fills the chip;
        // small images keep one tile per wave (twice the workgroups)
        const dim3 grid2(gx, (y1 - y0 + 2 * TILE_H * WG_TILES_Y - 1) / (2 * TILE_H * WG_TILES_Y), gz);   // 16 x 16 pixels per workgroup
        // the choice between the variants is made from the FULL image (Hu rows), never from the band [y0,y1): the
        // variants agree only to rounding (tests: 5e-4 x scale), and a row band of a multi-GPU split has to stitch
        // bit-exactly into the same rows of an unsharded decode (sharded.py)
        const long long full_gy = (p.Hu + TILE_H * WG_TILES_Y - 1) / (TILE_H * WG_TILES_Y);
        const long long full_gy2 = (p.Hu + 2 * TILE_H * WG_TILES_Y - 1) / (2 * TILE_H * WG_TILES_Y);
        const long long full_gx = (p.Wu + TILE_W * WG_TILES_X - 1) / (TILE_W * WG_TILES_X);   // ... nor from a column tile [x0,x1)
        const bool two_tiles = full_gx * full_gy2 * gz >= 512;
        // diagnostic override (tests / A-B timing): DIINN_BF16_KERNEL = 1 one tile per wave, 2 two tiles per wave,
        // 8 cooperative, one block per workgroup, 9 cooperative persistent; unset = pick by launch size and scale
        const int force = (int)knob(diinn_knobs().bf16_kernel);
        // the cooperative kernel stages the P rows of a block's LR footprint in LDS: needs the footprint to fit
        // (scales from about x3 up), and >= 2 rounds of workgroups to be worth its prologue
        DecodeParams pc = p;
        int ncx, ncy;
        coop_footprint(p, ncx, ncy);
        pc.seed_cols = ncx;
        // a 16 x 8 block covers whole cells iff 16 / scale_x and 8 / scale_y are integers
        pc.xcd_runs = !((16LL * p.W) % p.Wu == 0 && (8LL * p.H) % p.Hu == 0);
        const bool coop_ok = ncx * ncy <= CO_SEED_CELLS;
        const bool coop = coop_ok && (force ? (force == 8 || force == 9) : full_gx * full_gy * gz >= 1024);
        // (the polynomial sine needs a few registers more than the 256 a wave may have at two per SIMD once the block
        // loop's state is added: DIINN_SIN_ACCURATE stays on the one-block-per-workgroup form)
        if (coop && force != 8 && sin_mode != DIINN_SIN_ACCURATE) {   // 8 waves, persistent workgroups over super-tiles of 8 x 4 blocks
            pc.pg[0] = gx; pc.pg[1] = gy; pc.pg[2] = gz;
            pc.pg[3] = (gx + CO_ST_X - 1) / CO_ST_X; pc.pg[4] = (gy + CO_ST_Y - 1) / CO_ST_Y;
            const long long nst = (long long)pc.pg[3] * pc.pg[4] * gz;
            if (nst > 0x7fffffffLL) return DIINN_ERR_TOO_LARGE;
            pc.pg[5] = (int)((nst + 7) / 8);
            const dim3 gridp(CO_PGRID);
            if (sin_mode == DIINN_SIN_HW)
                hipLaunchKernelGGL(decode_bf16_coop8p_kernel<DIINN_SIN_HW>, gridp, dim3(512), 0, (hipStream_t)stream, pc);
            else
                hipLaunchKernelGGL(decode_bf16_coop8p_kernel<DIINN_SIN_HW_REDUCED>, gridp, dim3(512), 0, (hipStream_t)stream, pc);
        } else if (coop) {                                        // one block per workgroup (DIINN_BF16_KERNEL=8)
            if (sin_mode == DIINN_SIN_HW)
                hipLaunchKernelGGL(decode_bf16_coop8_kernel<DIINN_SIN_HW>, grid, dim3(512), 0, (hipStream_t)stream, pc);
            else if (sin_mode == DIINN_SIN_HW_REDUCED)
                hipLaunchKernelGGL(decode_bf16_coop8_kernel<DIINN_SIN_HW_REDUCED>, grid, dim3(512), 0, (hipStream_t)stream, pc);
            else
                hipLaunchKernelGGL(decode_bf16_coop8_kernel<DIINN_SIN_ACCURATE>, grid, dim3(512), 0, (hipStream_t)stream, pc);
        } else if (!((force == 1 || force == 2) ? force == 2 : two_tiles)) {
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
