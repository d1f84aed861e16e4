// diinn_precompute_wino.hip -- the hoisted 3x3 convolution P = Wx . unfold3x3(feat) + bK as Winograd F(2x2, 3x3)
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
//
// Same product as precompute_P_kernel (diinn.py:168 unfold + the feature columns of K[i], diinn.py:133,136), with
// 16 multiplies per (input, output) pair and 2x2 block of LR cells instead of 36 (csrc/diinn_winograd.hip has the
// algebra): 2.25x fewer MFMAs for the 64 -> 1024 convolution that is ~9 % of a decode step.  fp32 throughout; the
// transformed weight U = G Wx G^T is section WPU of the packed image (float64 on the host, rounded once).
//
// A workgroup owns 8 x 4 Winograd tiles (16 x 8 cells) = one 32-wide MFMA N-tile and ALL 1024 outputs.  Wave i of 4
// (one per SIMD, the whole register file) owns row i of the transformed tile:
//   * prologue: the transformed input V[col j][k-step] of its row for all 64 channels -- 64 patch-row loads, 128
//     registers -- is computed ONCE and stays in registers;
//   * main loop over the 32 M-tiles of 32 outputs: 4 columns x 32 k-steps = 128 MFMAs with A pieces streamed from
//     the packed image through a register ring, B operands straight from the V registers: no LDS, no VALU;
//   * the output transform of M-tile m runs inside the MFMA stream of M-tile m + 1 (two accumulator sets): column half
//     in registers, row half through a double-buffered LDS exchange (one barrier per M-tile) that also re-arranges
//     the results so that every store instruction writes whole 128-byte lines of P (a cell's 32 channels of this
//     M-tile; 8 lanes per cell), + bK, streaming stores for big P.
// The tile grid is anchored at even absolute rows / columns whatever band is computed, so a band's P is bit-identical
// to the same rows of a full launch.  Patch rows a band's feature window does not hold are only ever combined into
// output rows outside the band (never stored); they read as zeros.
#include "diinn_device.h"
#include <stdlib.h>

struct PWinoParams {
    const float* feat;   // [B,64,Frows,W] = LR rows [Frow0, Frow0+Frows) of the [B,64,H,W] map
    const float* Wt;     // packed image (sections WPU, BK)
    float* P;            // [B,Prows,W,1024] = LR rows [Prow0, Prow0+Prows)
    int B, H, W, r0, r1;
    int Frow0, Frows, Prow0, Prows;
    int ty_first;        // first Winograd tile row of the band: r0 >> 1
    int bx_n, by_n;      // blocks of 8 x 4 tiles
    int msplit;          // the 32 M-tiles are divided over `msplit` workgroups per block (small maps: fills the chip)
    int stream_stores;
    int warm_l2;         // touch the Winograd weight section first (maps whose P image has flushed it out of the L2s)
    int wpu_only_ok;     // the validity word may also be DIINN_PACKED_MAGIC_WPU (a training image: permutation sections + WPU only)
};

constexpr int PWN_TX = 8, PWN_TY = 4;
constexpr int PWN_PITCH = 36;                                    // floats per (row, q, tile) in the exchange: 32 channels + 4 (rotates banks)
constexpr int PWN_ZS = 4 * 2 * 32 * PWN_PITCH;                   // one exchange buffer: [row i][q][tile][36] = 36,864 B
constexpr int PWN_RING = 4;                                      // A ring: k-groups (4 pieces each) in flight + the one in use
constexpr int PWN_MT_BYTES = 4 * 8 * 4 * PIECE_BYTES;            // bytes of WPU per M-tile (all four rows)

#define PWN_SB() __builtin_amdgcn_sched_barrier(0)

template <bool EDGE>
__device__ __forceinline__ void precompute_P_wino_body(const PWinoParams& p, float* __restrict__ zs, int b, int tx0, int ty0, int mt0, int mtn) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = row i of the transformed tile
    const int h = lane >> 5, m = lane & 31;
    const int tx = tx0 + (m & (PWN_TX - 1)), ty = ty0 + m / PWN_TX;
    const unsigned plane_b = (unsigned)((size_t)p.Frows * p.W * sizeof(float));      // one channel of the feature window
    const int lane_off = lane * 16;
    constexpr unsigned OUTSIDE = 0x80000000u;

    // ---- prologue: V[j][s] = column j of row i of B^T d B for channel pair s (channel 2s + h), all 32 pairs
    f32x2 V01[32], V23[32];                                      // columns 0 1 | -2 3
    {
        const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
        const int rb = wave == 0 ? 2 : (wave == 2 ? 1 : (wave == 1 ? 2 : 3));
        const float sgn = wave == 1 ? 1.0f : -1.0f;
        const bool tile_in = 2 * tx < p.W && 2 * ty < p.H;
        const bool left = EDGE && tx == 0;
        const bool ok2 = 2 * tx + 1 < p.W, ok3 = 2 * tx + 2 < p.W;
        const int xc = left ? 0 : 2 * tx - 1;
        const int ya = 2 * ty - 1 + ra, yb = 2 * ty - 1 + rb;
        const int f1 = p.Frow0 + p.Frows;                        // rows of the map the window holds: [Frow0, f1)
        const unsigned offa = (tile_in && ya >= p.Frow0 && ya < f1) ? (unsigned)h * plane_b + (unsigned)((ya - p.Frow0) * p.W + xc) * 4u : OUTSIDE;
        const unsigned offb = (tile_in && yb >= p.Frow0 && yb < f1) ? (unsigned)h * plane_b + (unsigned)((yb - p.Frow0) * p.W + xc) * 4u : OUTSIDE;
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.feat + (size_t)b * C_IN * p.Frows * p.W), 0, (int)((unsigned)C_IN * plane_b), 0x00020000);
        f32x2 pm = {-1.0f, 1.0f};
        asm volatile("" : "+v"(pm));
        const f32x2 sgn2 = {sgn, sgn};
#pragma unroll
        for (int s0 = 0; s0 < 32; s0 += 8) {                     // 16 loads in flight at a time
            f32x4 ra4[8], rb4[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ra4[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frs, (int)offa, (int)((unsigned)(2 * (s0 + u)) * plane_b), 0));
                rb4[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frs, (int)offb, (int)((unsigned)(2 * (s0 + u)) * plane_b), 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f32x2 t01 = __builtin_elementwise_fma(sgn2, f32x2{rb4[u][0], rb4[u][1]}, f32x2{ra4[u][0], ra4[u][1]});
                f32x2 t23 = __builtin_elementwise_fma(sgn2, f32x2{rb4[u][2], rb4[u][3]}, f32x2{ra4[u][2], ra4[u][3]});
                if constexpr (EDGE) {
                    const float d0 = left ? 0.0f : t01[0];
                    const float d1 = left ? t01[0] : t01[1];
                    float d2 = left ? t01[1] : t23[0];
                    float d3 = left ? t23[0] : t23[1];
                    d2 = ok2 ? d2 : 0.0f;
                    d3 = ok3 ? d3 : 0.0f;
                    t01 = f32x2{d0, d1};
                    t23 = f32x2{d2, d3};
                }
                V01[s0 + u] = __builtin_elementwise_fma(f32x2{t23[0], t23[0]}, pm, t01);
                V23[s0 + u] = f32x2{t01[1], t01[1]} - t23;
            }
        }
    }

    // ---- output side: wave = output row pr of the 2x2 block and tile half th; store group k, lane L: cell n = 8k + (L >> 3)
    // of the wave's 32 cells (tile 16 th + (n >> 1), column q = n & 1), chunk c = L & 7 (channels 4c .. 4c+3 of the M-tile)
    const int pr = wave & 1, th = wave >> 1;
    const float s2 = pr == 0 ? 1.0f : -1.0f;                    // row 0: Z0 + Z1 + Z2, row 1: Z1 - Z2 - Z3
    const int chunk = lane & 7;
    unsigned zoff[4];                                            // LDS float offset of (row pr, q, tile, chunk) per store group
    float* pdst[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = 8 * k + (lane >> 3);
        const int tm = 16 * th + (n >> 1), q = n & 1;
        zoff[k] = (unsigned)(((pr * 2 + q) * 32 + tm) * PWN_PITCH + 4 * chunk);
        const int oy = 2 * (ty0 + tm / PWN_TX) + pr, ox = 2 * (tx0 + (tm & (PWN_TX - 1))) + q;
        const bool in = oy >= p.r0 && oy < p.r1 && ox < p.W;
        pdst[k] = in ? p.P + (((size_t)b * p.Prows + (oy - p.Prow0)) * p.W + ox) * PCH + 4 * chunk : nullptr;
    }
    const float* __restrict__ bk = p.Wt + OFF_BK + 4 * chunk;
    unsigned nanm = derived_nan_mask(p.Wt);                    // section WPU missing -> NaN into every P value
    if (p.wpu_only_ok && __builtin_bit_cast(unsigned, p.Wt[OFF_BL + 3]) == DIINN_PACKED_MAGIC_WPU) nanm = 0u;

    // ---- main loop
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.Wt + OFF_WPU + (size_t)wave * 8 * 4 * WL_PIECE), 0, (int)(SZ_WPU * sizeof(float)), 0x00020000);
    f32x4 A[PWN_RING][4];                                        // [k-group in the ring][col j]
    f32x16 acc[2][4];                                            // [M-tile parity][col j]
    // piece (mt, sg, j) of this wave: byte offset mt * PWN_MT_BYTES + (sg * 4 + j) * PIECE_BYTES; the stream is linear
    // in g = 8 mt + sg apart from the jump between M-tiles
    auto request = [&](auto SLOT_, auto J_, int mt, int sg) {
        constexpr int SLOT = decltype(SLOT_)::value, J = decltype(J_)::value;
        A[SLOT][J] = ld_piece(wrs, lane_off, mt * PWN_MT_BYTES + (sg * 4 + J) * PIECE_BYTES);
    };
    static_assert(PWN_RING == 4, "the prologue requests k-groups 0..2 of the first M-tile");
    request(IC<0>{}, IC<0>{}, mt0, 0); request(IC<0>{}, IC<1>{}, mt0, 0); request(IC<0>{}, IC<2>{}, mt0, 0); request(IC<0>{}, IC<3>{}, mt0, 0);
    request(IC<1>{}, IC<0>{}, mt0, 1); request(IC<1>{}, IC<1>{}, mt0, 1); request(IC<1>{}, IC<2>{}, mt0, 1); request(IC<1>{}, IC<3>{}, mt0, 1);
    request(IC<2>{}, IC<0>{}, mt0, 2); request(IC<2>{}, IC<1>{}, mt0, 2); request(IC<2>{}, IC<2>{}, mt0, 2); request(IC<2>{}, IC<3>{}, mt0, 2);
    PWN_SB();

    f32x4 zc[2][4];                                              // column-transformed previous M-tile: [q][group g of 4 registers]
    f32x4 bias4 = {};
    // finish steps of the previous M-tile (parity PP, index mprev), spread over the k-groups of the current one
    auto col_transform = [&](auto PP_, auto G_) {                // one group of 4 accumulator registers
        constexpr int PP = decltype(PP_)::value, g = decltype(G_)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float m0 = acc[PP][0][r], m1 = acc[PP][1][r], m2 = acc[PP][2][r], m3 = acc[PP][3][r];
            zc[0][g][e] = (m0 + m1) + m2;
            zc[1][g][e] = (m1 - m2) - m3;
        }
    };
    auto exchange_write = [&](int buf) {
        float* __restrict__ zb = zs + buf * PWN_ZS;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(zb + (((wave * 2 + q) * 32 + m) * PWN_PITCH + 4 * (2 * g + h))) = zc[q][g];
    };
    auto store_group = [&](int k, int buf, int mprev) {
        const float* __restrict__ zb = zs + buf * PWN_ZS + zoff[k];
        const f32x4 z0 = *reinterpret_cast<const f32x4*>(zb);
        const f32x4 z1 = *reinterpret_cast<const f32x4*>(zb + 2 * 32 * PWN_PITCH);
        const f32x4 z2 = *reinterpret_cast<const f32x4*>(zb + 4 * 32 * PWN_PITCH);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf(s2, z2[e], __builtin_fmaf(s2, z1[e], z0[e])) + bias4[e];
        if (pdst[k]) {
            f32x4* dst = reinterpret_cast<f32x4*>(pdst[k] + 32 * mprev);
            if (p.stream_stores) __builtin_nontemporal_store(y, dst);
            else *dst = y;
        }
    };
    // one M-tile: 8 k-groups x (4 k-steps x 4 columns) MFMAs into acc[PAR]; PREV: finish M-tile mt - 1 meanwhile
    auto mtile = [&](auto PAR_, int mt, bool prev) {
        constexpr int PAR = decltype(PAR_)::value;
        const f32x16 zero = {};
        auto kgroup = [&](auto SG_) {
            constexpr int SG = decltype(SG_)::value;
            constexpr int SLOT = SG % PWN_RING, SLOTL = (SG + PWN_RING - 1) % PWN_RING;
            // the k-group requested now: RING - 1 ahead, possibly in the next M-tile (past the last M-tile the range
            // check of the descriptor answers)
            const int sgl = (SG + PWN_RING - 1) & 7;
            const int mtl = mt + ((SG + PWN_RING - 1) >> 3);
            auto kstep = [&](auto E_) {
                constexpr int e = decltype(E_)::value;
                constexpr int s = 4 * SG + e;
                constexpr bool FIRST = SG == 0 && e == 0;        // the accumulators start from an inline zero
                acc[PAR][0] = MFMA32(A[SLOT][0][e], V01[s][0], FIRST ? zero : acc[PAR][0]);
                PWN_SB();
                acc[PAR][1] = MFMA32(A[SLOT][1][e], V01[s][1], FIRST ? zero : acc[PAR][1]);
                PWN_SB();
                acc[PAR][2] = MFMA32(A[SLOT][2][e], V23[s][0], FIRST ? zero : acc[PAR][2]);
                PWN_SB();
                acc[PAR][3] = MFMA32(A[SLOT][3][e], V23[s][1], FIRST ? zero : acc[PAR][3]);
                request(IC<SLOTL>{}, IC<e>{}, mtl, sgl);
                if constexpr (SG == 0) {                         // the previous M-tile's column transform, a quarter per k-step
                    if (prev) col_transform(IC<1 - PAR>{}, IC<e>{});
                }
                PWN_SB();
            };
            kstep(IC<0>{}); kstep(IC<1>{}); kstep(IC<2>{}); kstep(IC<3>{});
        };
        kgroup(IC<0>{});
        kgroup(IC<1>{});
        if (prev) {
            exchange_write((mt - 1) & 1);
            bias4 = or_bits(*reinterpret_cast<const f32x4*>(bk + 32 * (mt - 1)), nanm);
            __syncthreads();
        }
        PWN_SB();
        kgroup(IC<2>{});
        if (prev) store_group(0, (mt - 1) & 1, mt - 1);
        PWN_SB();
        kgroup(IC<3>{});
        if (prev) store_group(1, (mt - 1) & 1, mt - 1);
        PWN_SB();
        kgroup(IC<4>{});
        if (prev) store_group(2, (mt - 1) & 1, mt - 1);
        PWN_SB();
        kgroup(IC<5>{});
        if (prev) store_group(3, (mt - 1) & 1, mt - 1);
        PWN_SB();
        kgroup(IC<6>{});
        kgroup(IC<7>{});
    };
    // this workgroup's M-tiles [mt0, mt0 + mtn), any count >= 1 (r04: small maps split the 32 M-tiles over any number of
    // workgroups up to 16, not only powers of two): the accumulator sets alternate from set 0 at mt0
    int mt = mt0;
#pragma unroll 1
    for (; mt + 1 < mt0 + mtn; mt += 2) {
        mtile(IC<0>{}, mt, mt > mt0);
        mtile(IC<1>{}, mt + 1, true);
    }
    const bool odd = mt < mt0 + mtn;                             // workgroup-uniform
    if (odd) mtile(IC<0>{}, mt, mt > mt0);
    const int mtl = mt0 + mtn - 1;
    // the last M-tile (set 1, or set 0 after an odd count)
    if (odd) {
        col_transform(IC<0>{}, IC<0>{}); col_transform(IC<0>{}, IC<1>{}); col_transform(IC<0>{}, IC<2>{}); col_transform(IC<0>{}, IC<3>{});
    } else {
        col_transform(IC<1>{}, IC<0>{}); col_transform(IC<1>{}, IC<1>{}); col_transform(IC<1>{}, IC<2>{}); col_transform(IC<1>{}, IC<3>{});
    }
    exchange_write(mtl & 1);
    bias4 = or_bits(*reinterpret_cast<const f32x4*>(bk + 32 * mtl), nanm);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) store_group(k, mtl & 1, mtl);
}

__global__ __launch_bounds__(256, 1) void precompute_P_wino_kernel(const PWinoParams p) {
    __shared__ __attribute__((aligned(16))) float zs[2 * PWN_ZS];
    const int per_xcd = gridDim.x >> 3;                          // every XCD a contiguous run of blocks
    int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int per_b = p.bx_n * p.by_n;
    if (t >= p.B * per_b * p.msplit) return;
    // L2 warm-up (r03; the direct kernel has had it since r02).  Inside a decode step this kernel starts right after
    // decode_kernel has streamed hundreds of MB of P through every L2, so the 4 MiB Winograd weight section is gone;
    // the workgroups of an XCD then walk it in lock-step and the first round advances at HBM-latency pace (a block takes
    // 167 us in the two rounds of c2 against 142 us in the eight of a 512 x 512 map).  Every workgroup of the first
    // round touches its share of the section's 128-byte lines first; the values are dead.  In the c2 step: 0.325 ->
    // 0.293 ms (same box, A/B); c5: neutral; on maps whose whole working set stays in the L2s (c1: +1.6 us on 26) it is
    // pure overhead, so the host switches it on from 8,192 cells (a P image of 32 MiB = the eight L2s) on.
    // (round 6: the touches are LDS-DMA loads into a dead LDS area -- l2_touch, diinn_device.h, says why not registers)
    __shared__ float warm_sink[4][64];
    {
        const unsigned slot = blockIdx.x >> 3;                   // this workgroup's index inside its XCD
        const unsigned first = (unsigned)per_xcd < 32u ? (unsigned)per_xcd : 32u;   // workgroups of an XCD's first round (one per CU)
        constexpr unsigned LINES = (unsigned)(SZ_WPU * sizeof(float) / 128);
        const unsigned share = (LINES + first - 1) / first;      // <= 1024 once 32 workgroups share the section
        if (p.warm_l2 && slot < first) {
            const __amdgpu_buffer_rsrc_t wpu = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wt + OFF_WPU), 0, (int)(SZ_WPU * sizeof(float)), 0x00020000);
            float* sink = warm_sink[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned k = threadIdx.x + 256 * i;
                unsigned line = slot * share + k;
                line = (k < share && line < LINES) ? line : 0;
                l2_touch(wpu, line * 128u, sink);
            }
        }
    }
    const int part = __builtin_amdgcn_readfirstlane(t % p.msplit);   // the parts of a block are neighbours: they share its patch rows
    t /= p.msplit;
    const int mt0 = part * 32 / p.msplit, mtn = (part + 1) * 32 / p.msplit - mt0;   // parts of floor or ceil(32 / msplit) M-tiles
    const int b = __builtin_amdgcn_readfirstlane(t / per_b);
    t -= b * per_b;
    const int by = __builtin_amdgcn_readfirstlane(t / p.bx_n), bx = t - by * p.bx_n;
    const int tx0 = bx * PWN_TX, ty0 = p.ty_first + by * PWN_TY;
    const bool edge = tx0 == 0 || 2 * (tx0 + PWN_TX - 1) + 2 >= p.W;
    if (edge) precompute_P_wino_body<true>(p, zs, b, tx0, ty0, mt0, mtn);
    else      precompute_P_wino_body<false>(p, zs, b, tx0, ty0, mt0, mtn);
}

// Winograd form of launch_P for the fp32 hoisted convolution of all 1024 channels (diinn_precompute.hip decides when)
int launch_P_wino(void* stream, const float* feat_dev, const float* packed_dev, float* P_dev,
                  int B, int H, int W, int r0, int r1, RowWin fw, RowWin pw, bool wpu_only_ok) {
    PWinoParams p;
    p.wpu_only_ok = wpu_only_ok ? 1 : 0;
    p.feat = feat_dev; p.Wt = packed_dev; p.P = P_dev;
    p.B = B; p.H = H; p.W = W; p.r0 = r0; p.r1 = r1;
    p.Frow0 = fw.row0; p.Frows = fw.rows; p.Prow0 = pw.row0; p.Prows = pw.rows;
    p.ty_first = r0 >> 1;
    const int ty_n = ((r1 - 1) >> 1) - p.ty_first + 1, tx_n = (W + 1) / 2;
    p.bx_n = (tx_n + PWN_TX - 1) / PWN_TX;
    p.by_n = (ty_n + PWN_TY - 1) / PWN_TY;
    long long blocks = (long long)p.bx_n * p.by_n * B;
    // The 32 M-tiles of a block may be divided over up to 16 workgroups (one per CU at a time): pick the split with the
    // fewest rounds x (prologue + M-tiles per workgroup).  It fills the chip on small maps and evens out the last round
    // on odd ones; results do not depend on it.
    {
        const double prologue_us = 4.0, mtile_us = 3.4;          // measured: V + pipeline fill; 128 MFMAs
        double best = 1e30;
        p.msplit = 1;
        for (int ms = 1; ms <= 16; ++ms) {                       // (r04: any split -- c1's 18 blocks take 11 parts of <= 3 M-tiles
            const double rounds = (double)((blocks * ms + 255) / 256);   //  in one round where powers of two offered 8 parts of 4)
            const double cost = rounds * (prologue_us + ((32 + ms - 1) / ms) * mtile_us);
            if (cost < best * 0.97) { best = cost; p.msplit = ms; }   // prefer the coarser split unless clearly worse
        }
    }
    blocks *= p.msplit;
    if (blocks > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    if ((long long)C_IN * fw.rows * W * 4 > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;   // the feature window is addressed with 32-bit byte offsets
    p.stream_stores = (double)B * (r1 - r0) * W * PCH * 4.0 >= 128.0 * 1024 * 1024;
    p.warm_l2 = (long long)B * (r1 - r0) * W >= 8192;
    hipLaunchKernelGGL(precompute_P_wino_kernel, dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}
