// diinn_baselines.hip -- the LIIF and MetaSR comparison decoders
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
#include "diinn_device.h"

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
                        // opaque use: keeps each rectification where it is written, between the MFMAs.  Left free, the
                        // scheduler regroups the 128 of them and the layer needs 954 registers more than exist (r02).
                        asm volatile("" : "+v"(qn[16 * (m - 1) + r]));
                    }
                }
                ps = as;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                qn[16 * 7 + r] = relu0(ps[r]);
                asm volatile("" : "+v"(qn[16 * 7 + r]));
            }
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

extern "C" {

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

}  // extern "C"
