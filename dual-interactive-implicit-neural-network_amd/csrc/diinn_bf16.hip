// diinn_bf16.hip -- the optional bf16-operand decode kernels (DIINN_COMPUTE_BF16 / DIINN_COMPUTE_BF16_FULL)
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
#include "diinn_device.h"

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
        const size_t plane = (size_t)p.Orows * p.Wu;
        float* o = p.out + (size_t)b * 3 * plane + (size_t)(y - p.Orow0) * p.Wu + x;
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
            const size_t plane = (size_t)p.Orows * p.Wu;
            float* op = p.out + (size_t)b * 3 * plane + (size_t)(y[t] - p.Orow0) * p.Wu + x;
            op[0] = o0 + Wt[OFF_BL + 0];
            op[plane] = o1 + Wt[OFF_BL + 1];
            op[2 * plane] = o2 + Wt[OFF_BL + 2];
        }
    }
}

int launch_decode_bf16(void* stream, const DecodeParams& p, int gx, int gy, int gz, int sin_mode) {
    const int y0 = p.y0, y1 = p.y1, blk = 256;
    const dim3 grid(gx, gy, gz);

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
