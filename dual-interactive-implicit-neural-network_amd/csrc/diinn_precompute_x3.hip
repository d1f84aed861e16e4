// diinn_precompute_x3.hip -- the hoisted 3x3 convolution P = Wx . unfold3x3(feat) + bK in split-bf16 arithmetic
// (DIINN_COMPUTE_BF16X3 on maps of >= 32,768 cells; part of libdiinn_hip.so, shared definitions in diinn_device.h)
//
// The decoder's per-pixel layers of that mode carry every operand as hi + lo bf16 parts (diinn_bf16x3.hip); with them at
// 2 ms the fp32 Winograd form of this convolution is 13-18 % of a step.  Here the same arithmetic -- w_lo.x_hi + w_hi.x_lo
// + w_hi.x_hi on v_mfma_f32_32x32x16_bf16, fp32 accumulation -- evaluates the direct sum, on the machinery of the encoder's
// conv3x3_x3m_kernel (diinn_conv_x3.hip): a workgroup of eight waves owns 32 x 8 cells; the halo patch of all 64 input
// channels is gathered, split and staged in LDS ONCE (B-fragment order, 85 KiB), then the 1024 output channels are worked
// in 16 groups of 64 -- waves 0..3 the first 32 of a group, 4..7 the other 32, each its two cell rows -- with the weights
// (packed section 15, WPX) streamed through a register ring.  A group's 64 x 256 results leave at once: transposed through a
// wave-private LDS slab so that eight lanes write a cell's 128 contiguous bytes of its P row, while the next group's MFMAs
// run -- no store burst at the end of the launch.
// Row windows as in the other P kernels (include/diinn_hip.h "row windows"): `feat` holds LR rows [Frow0, Frow0 + Frows),
// `P` rows [Prow0, Prow0 + Prows); rows outside the map read as zeros (the convolution's padding).
#include "diinn_device.h"

constexpr int PX_TX = 32, PX_TY = 8;                  // cells of a workgroup
constexpr int PX_PW = PX_TX + 2, PX_PH = PX_TY + 2;   // its halo patch
constexpr int PX_TASKS = 4 * 2 * PX_PH * PX_PW;       // staging tasks: (group of 16 channels, k-half, patch cell) -> 8 channels
constexpr int PX_ITERS = (PX_TASKS + 511) / 512;
constexpr int PX_RING = 6;                            // weight steps ((group, tap): hi + lo piece of the wave's M-tile) in flight
constexpr int PX_TR_PITCH = 36;                       // floats per cell in the transposition slab (32 channels + 4: rotates the banks)

struct PX3Params {
    const float* feat;   // [B,64,Frows,W]
    const float* Wt;     // packed image (sections WPX, BK)
    float* P;            // [B,Prows,W,1024]
    int B, H, W, r0, r1;
    int Frow0, Frows, Prow0, Prows;
    int stream_stores;
};

__global__ __launch_bounds__(512, 1) void precompute_P_x3_kernel(const PX3Params p) {
    __shared__ __attribute__((aligned(16))) bf16x8 stage[4][2][2][PX_PH][PX_PW];   // [group][hi, lo][k-half][row][col] = 85 KiB
    __shared__ __attribute__((aligned(16))) float tr[8][32 * PX_TR_PITCH];          // per wave: [cell][channel] = 36 KiB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wave >> 2, rw = wave & 3;
    const int h = lane >> 5, px = lane & 31;
    const int x0 = blockIdx.x * PX_TX, y0 = p.r0 + blockIdx.y * PX_TY, b = blockIdx.z;
    const float* __restrict__ Wt = p.Wt;

    // ---- the patch of all 64 channels: gathered from the planes, split into hi / lo, staged once
    {
        const size_t fplane = (size_t)p.Frows * p.W;
        const float* __restrict__ fb = p.feat + (size_t)b * C_IN * fplane;
        bf16x8* __restrict__ st = &stage[0][0][0][0][0];
#pragma unroll
        for (int i = 0; i < PX_ITERS; ++i) {
            const int task = tid + 512 * i;
            if (task >= PX_TASKS) break;
            const int g = task / (2 * PX_PH * PX_PW), r1_ = task - g * (2 * PX_PH * PX_PW);
            const int kh = r1_ / (PX_PH * PX_PW), rem = r1_ - kh * (PX_PH * PX_PW);
            const int row = rem / PX_PW, col = rem - row * PX_PW;
            const int y = y0 + row - 1, x = x0 + col - 1;
            const bool inside = y >= 0 && y < p.H && x >= 0 && x < p.W && y >= p.Frow0 && y < p.Frow0 + p.Frows;
            bf16x8 vh, vl;
            const float* __restrict__ src = fb + (size_t)(16 * g + 8 * kh) * fplane + (inside ? (size_t)(y - p.Frow0) * p.W + x : 0);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float v = inside ? src[(size_t)c * fplane] : 0.0f;
                const __bf16 a = (__bf16)v;
                vh[c] = a;
                vl[c] = (__bf16)(v - (float)a);
            }
            const int cell = (kh * PX_PH + row) * PX_PW + col;
            st[(size_t)(g * 2 + 0) * (2 * PX_PH * PX_PW) + cell] = vh;
            st[(size_t)(g * 2 + 1) * (2 * PX_PH * PX_PW) + cell] = vl;
        }
    }

    // ---- the weight stream of this wave: step s = (og * 4 + g) * 9 + tap, pieces [mt][hi, lo] of 1 KiB each
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(Wt + OFF_WPX), 0, (int)(SZ_WPX * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16 + mt * 2 * PIECE_BYTES;
    f32x4 Wr[PX_RING][2];
    auto load_step = [&](const int slot, const int s) {
        Wr[slot][0] = ld_piece(wrs, lane_off, s * 4 * PIECE_BYTES);
        Wr[slot][1] = ld_piece(wrs, lane_off + PIECE_BYTES, s * 4 * PIECE_BYTES);
    };
#pragma unroll
    for (int d = 0; d < PX_RING; ++d) load_step(d, d);
    __syncthreads();                                             // the patch is in LDS; no barrier after this one

    float* __restrict__ slab = &tr[wave][0];
    const size_t prow = (size_t)p.W * PCH;
    float* __restrict__ Pb = p.P + (size_t)b * p.Prows * prow;
    const unsigned nanm = derived_nan_mask(Wt);                  // an image without its derived sections answers NaN
    constexpr int STEPS_OG = 4 * 9;
    static_assert(STEPS_OG % PX_RING == 0, "ring index must be static");
#pragma unroll 1
    for (int og = 0; og < 16; ++og) {
        f32x16 acc[2];
        {
            const float* __restrict__ bk = Wt + OFF_BK + 64 * og + 32 * mt + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = or_bits(*(const f32x4*)(bk + 8 * q), nanm);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[0][4 * q + i] = acc[1][4 * q + i] = bv[i];
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x8* __restrict__ sh = &stage[g][0][h][2 * rw][px];
            const bf16x8* __restrict__ sl = &stage[g][1][h][2 * rw][px];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int s_local = g * 9 + tap;                 // static: the ring slot
                const int ky = tap / 3, kx = tap - 3 * ky;
                const bf16x8 xh0 = sh[(0 + ky) * PX_PW + kx], xl0 = sl[(0 + ky) * PX_PW + kx];
                const bf16x8 xh1 = sh[(1 + ky) * PX_PW + kx], xl1 = sl[(1 + ky) * PX_PW + kx];
                const bf16x8 wh = __builtin_bit_cast(bf16x8, Wr[s_local % PX_RING][0]);
                const bf16x8 wl = __builtin_bit_cast(bf16x8, Wr[s_local % PX_RING][1]);
                acc[0] = MFMA_BF16(wl, xh0, acc[0]);
                acc[1] = MFMA_BF16(wl, xh1, acc[1]);
                acc[0] = MFMA_BF16(wh, xl0, acc[0]);
                acc[1] = MFMA_BF16(wh, xl1, acc[1]);
                acc[0] = MFMA_BF16(wh, xh0, acc[0]);
                acc[1] = MFMA_BF16(wh, xh1, acc[1]);
                const int s_next = og * STEPS_OG + s_local + PX_RING;   // past the end: answered with zeros, never used
                load_step(s_local % PX_RING, s_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- the group's 32 channels x 2 rows x 32 cells of this wave: transposed through the slab, a cell's 128 bytes by 8 lanes
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0 + 2 * rw + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[r][4 * q + i];
                *(f32x4*)(slab + px * PX_TR_PITCH + 8 * q + 4 * h) = v;
            }
            if (y < p.r1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cell = (lane >> 3) + 8 * i, chunk = lane & 7;
                    const f32x4 v = *(const f32x4*)(slab + cell * PX_TR_PITCH + 4 * chunk);
                    const int x = x0 + cell;
                    if (x < p.W) {
                        float* dst = Pb + (size_t)(y - p.Prow0) * prow + (size_t)x * PCH + 64 * og + 32 * mt + 4 * chunk;
                        if (p.stream_stores) __builtin_nontemporal_store(v, (f32x4*)dst);
                        else *(f32x4*)dst = v;
                    }
                }
            }
        }
    }
}

int launch_P_x3(void* stream, const float* feat_dev, const float* packed_dev, float* P_dev,
                int B, int H, int W, int r0, int r1, RowWin fw, RowWin pw) {
    PX3Params p{feat_dev, packed_dev, P_dev, B, H, W, r0, r1, fw.row0, fw.rows, pw.row0, pw.rows, 0};
    p.stream_stores = (double)B * (r1 - r0) * W * PCH * 4.0 >= 128.0 * 1024 * 1024;
    if ((long long)C_IN * fw.rows * W * 4 > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    if (B > 65535) return DIINN_ERR_TOO_LARGE;
    const dim3 grid((unsigned)((W + PX_TX - 1) / PX_TX), (unsigned)((r1 - r0 + PX_TY - 1) / PX_TY), (unsigned)B);
    hipLaunchKernelGGL(precompute_P_x3_kernel, grid, dim3(512), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}
