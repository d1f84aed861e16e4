// diinn_conv_t16.hip -- the 3x3 convolutions of the RDN trunk on SMALL maps (gfx950; round 6).
// (part of libdiinn_hip.so; shared definitions in diinn_device.h)
//
// Reference: src/models/components/rdn.py:9-35,90-105 on the reference's own timing protocol (runtime_test.py:13,31-33,59-62: a
// 1 x 3 x 48 x 48 crop).  What bounds a small map's layer (profiles/r05_small_map_analysis.txt, r06_small_map_s16.txt): the
// split-K kernel gives a 48 x 48 map 144 workgroups of (32 pixels, 32 outputs, all of Cin) = 9.4 MFLOP each at Cin 512, a layer
// costs one such workgroup's MFMA time on its CU, and 112 of the 256 CUs idle.  Finer units only help if a CU does not re-fetch
// the weights per unit (a CU takes in 46-70 GB/s from its L2).
//
// This kernel: the map is cut into STRIPS 16 pixels wide (one N-tile of v_mfma_f32_16x16x4_f32 per row) and the outputs into
// quarters of 16; a workgroup owns ONE output quarter and 1 .. 3 consecutive rows of one strip -- up to 64 workgroups per quarter
// on a 256-CU part, as many per strip as the busiest one's rows ask for (t16_plan; 48 x 48: 3 strips x 16 workgroups of 3 rows =
// 7.1 MFLOP on the busiest CU instead of 9.4) -- and streams its quarter's weights ONCE, each group of 16 input channels used for all of
// the workgroup's rows, whose halo rows it stages once (3 rows: 5 halo rows for 3 tiles).  8 waves split the reduction (wave w
// takes the groups w, w + 8, ...).  Direct fp32 sum: the reference's arithmetic up to the order of the sum.
//   * a step = one group: 16 channels x (rows + 2) halo rows x 24 columns (x0 - 4 .. x0 + 19, whole 16-byte pieces: W % 4 == 0)
//     by wave-private LDS-DMA into one of two stages (channel pitch 32 (rows + 2) - 16 floats = 16 banks mod 32, so the two
//     channels a 32-lane group of a ds_read_b32 touches fall on disjoint banks; the padding pieces and everything outside the
//     map are out-of-range lanes, which deposit zero), and the group's 9 KiB of weights into registers; 36 MFMAs per row.
//   * a step waits for ITS stage and weights with vmcnt(0) -- nothing younger is in flight then -- and computes, the next
//     step's loads going out between its MFMAs (MI355X_MICROARCH.md, Two waves per SIMD, item 7: LDS-DMA data is ordered for
//     the issuing wave's own ds_reads by its vmcnt).
//   * the weights are read from the SPLIT-K kernel's image (diinn_conv_ksplit: modules.pack_conv_ksplit): no second direct-sum
//     image, no change to diinn_rdn_forward_ex; the lane mapping is at the loads below.
//   * the 8 partial sums meet ONCE in LDS at the end (over the stages, behind a barrier); bias, ReLU / residual, 64-byte runs.
// Measured (profiles/r06_small_map_t16.txt): per trunk 48 x 48 1.95 -> 1.57 ms (with the 1x1 sibling below), 32 x 32 1.87 -> 0.95,
// 24 x 24 1.92 -> 0.87; per layer 5.7 + 1.9 us per 64 input channels at 48 x 48 (split-K: 6.1 + 2.33).  The workgroup's share of the map
// arrives as multipliers from the host (t16_plan): six integer divisions in the preamble cost every launch ~0.4 us.
#include "diinn_device.h"

constexpr int T16_WAVES = 8, T16_MAXR = 3;
constexpr int T16_GROUP = 16;                                  // input channels per step / weight run
constexpr int T16_STAGE = T16_GROUP * (32 * (T16_MAXR + 2) - 16);   // 2,304 floats = 9 KiB: the largest stage (3 rows)
constexpr int T16_LDS_FLOATS = T16_WAVES * 2 * T16_STAGE;      // 147,456 bytes
static_assert(T16_LDS_FLOATS * 4 <= 160 * 1024 && T16_WAVES * T16_MAXR * 4 * 64 <= T16_LDS_FLOATS, "LDS budget");
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#ifndef T16_W_AUX
#define T16_W_AUX 0          // cache policy of the weight loads / the LDS-DMA pieces (2 = nt; A/B builds: profiles/r06_small_map_t16.txt)
#endif
#ifndef T16_DMA_AUX
#define T16_DMA_AUX 0
#endif
__device__ __forceinline__ f32x4 t16_ld_w(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, T16_W_AUX));
}

struct ConvT16Params {
    const float* in;         // input channel planes: in + b*in_bs + c*H*W
    const float* w;          // the split-K kernel's image (diinn_conv_ksplit, 9 taps)
    const float* bias;       // [64]
    const float* res;        // optional residual [B,64,H,W] (batch stride res_bs), added after bias (no ReLU with it)
    float* out;              // out + b*out_bs + co*H*W
    float* out1;             // 1x1 kernel only: optional second destination (batch stride out1_bs)
    long long in_bs, out_bs, out1_bs, res_bs;
    int Cin, B, H, W, relu;
    // the partition, worked out once on the host (t16_plan): a strip has `per_strip` workgroups of `rows` rows (the last one what is
    // left); divisions by per_strip, tiles_x and Cin / 64 as multiplications (x / d == (x * m) >> 32 for x, d < 65536, m = 2^32 / d + 1)
    int per_strip, rows, tiles_x;
    unsigned m_per_strip, m_tiles_x, inv_runs;
#ifdef T16_STAMPS
    unsigned long long* stamps;   // tools/ubench/t16_bench.hip -DT16_STAMPS: s_memrealtime (100 MHz) per (workgroup, wave, point 0..7)
#endif
};
#ifdef T16_STAMPS
#define T16_STAMP(i)                                                                                             \
    do {                                                                                                         \
        unsigned long long t_;                                                                                   \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
        if (lane == 0) p.stamps[((size_t)blockIdx.x * T16_WAVES + wave) * 8 + (i)] = t_;                         \
    } while (0)
#else
#define T16_STAMP(i) do {} while (0)
#endif

// workgroup (quarter, slot) of the launch: strip slot / per_strip (image b, columns x0 .. x0 + 15), rows y0 .. y0 + rows - 1
struct T16Share { int b, x0, y0, rows; };
__device__ __forceinline__ T16Share t16_share(const ConvT16Params& p, int slot) {
    // (m = 2^32 / d + 1 does not fit 32 bits for d = 1)
    const int strip = p.per_strip == 1 ? slot : (int)(((unsigned long long)(unsigned)slot * p.m_per_strip) >> 32), j = slot - strip * p.per_strip;
    const int b = p.tiles_x == 1 ? strip : (int)(((unsigned long long)(unsigned)strip * p.m_tiles_x) >> 32);
    const int y0 = j * p.rows, left = p.H - y0;
    return T16Share{b, 16 * (strip - b * p.tiles_x), y0, left < p.rows ? left : p.rows};
}

template <int NR>
__device__ __forceinline__ void conv_t16_body(const ConvT16Params& p, float* __restrict__ lds, int quarter, int b, int y0, int x0) {
    constexpr int R = NR + 2, CP = 32 * R - 16, PPC = CP / 4, NDMA = 2 * R - 1, STAGE = T16_GROUP * CP;
    static_assert(NDMA * 64 == T16_GROUP * PPC && STAGE <= T16_STAGE, "stage geometry");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_b = (unsigned)(plane * sizeof(float));
    const int G = p.Cin / T16_GROUP;
    const int my = wave < G ? (G - wave + T16_WAVES - 1) >> 3 : 0;        // this wave's groups: wave, wave + 8, ...
    constexpr unsigned OUTSIDE = 0x80000000u;

    // the epilogue's thread: output co of column x; its bias is asked for now, not behind the last barrier
    const int ev = (int)threadIdx.x & 255, er = ev >> 6, el = ev & 63;
    const int co = 16 * quarter + 4 * (el >> 4) + er;
    const float bias = p.bias[co];
    T16_STAMP(0);                                                // body entered
#ifdef T16_STAMPS
    {
        unsigned hw_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
        if (lane == 0) p.stamps[((size_t)blockIdx.x * T16_WAVES + wave) * 8 + 7] = hw_;     // which SIMD / CU the wave sits on
    }
#endif
    f32x4 acc[NR];                                               // row ti: D[out 4 (lane >> 4) + r][column lane & 15]
#pragma unroll
    for (int ti = 0; ti < NR; ++ti) acc[ti] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    if (my > 0) {
        float* __restrict__ ring = lds + wave * 2 * T16_STAGE;
        unsigned voff[NDMA];
        const float* in_b = p.in + (size_t)b * p.in_bs;
        auto dma_rsrc = [&](int gi) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)(in_b + (size_t)T16_GROUP * (wave + T16_WAVES * gi) * plane), 0,
                                                     (int)((unsigned)T16_GROUP * plane_b), 0x00020000);
        };
        auto dma_piece = [&](auto PAR_, const __amdgpu_buffer_rsrc_t irs, int i) {
            constexpr int PAR = decltype(PAR_)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(ring + PAR * T16_STAGE + i * 256), 16, (int)voff[i], 0, 0, T16_DMA_AUX);
        };
        // the weights come from the split-K kernel's image ([half 2][wave 8][tap 9][run Cin/64][h 2][i 32][4]: the 16 bytes at
        // (half, wave w, tap, run g, h, i) hold W[32 half + i][8 (w Cin/64 + g) + 2 e + h][tap], e = 0 .. 3), so the trunk keeps ONE
        // direct-sum image: lane (k = lane >> 4, m = lane & 15) takes output 16 quarter + m and, of the group's 16 channels,
        // the run of 8 number k >> 1 at h = k & 1 -- k-step e multiplies the group's channels 8 (k >> 1) + 2 e + (k & 1)
        const int runs = p.Cin / 64;
        const unsigned inv_runs = p.inv_runs;                     // r / runs = (r * inv_runs) >> 16 for r < 128
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.Cin * (64 * 9 * 4), 0x00020000);
        const int wlane = (quarter >> 1) * (8 * 9 * runs * 1024) + ((lane >> 4) & 1) * 512 + (16 * (quarter & 1) + (lane & 15)) * 16;
        f32x4 wr[2][9];
        auto woff = [&](int gi) {
            const unsigned r8 = 2u * (wave + T16_WAVES * gi) + (lane >> 5);
            const unsigned w8 = (r8 * inv_runs) >> 16, g = r8 - w8 * runs;
            return wlane + (int)((w8 * 9 * runs + g) * 1024);
        };
        // B operand of (row ti, k-step e, tap (ky, kx)): the lane's channel of the k-step, halo row ti + ky, column (lane & 15) + kx - 1
        const float* __restrict__ bbase = ring + (8 * (lane >> 5) + ((lane >> 4) & 1)) * CP + (lane & 15) + 3;
        {
            const int off = woff(0);                             // the first weights are on their way while the stage's addresses are worked out
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wr[0][tap] = t16_ld_w(wrs, off, tap * runs * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        // piece i * 64 + lane of a stage: channel ch, halo row hr, piece pc of the row -- or one of the channel's padding pieces
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int pi = i * 64 + lane;
            const int ch = pi / PPC, rem = pi - ch * PPC;
            const int hr = rem / 6, pc = rem - hr * 6;
            const int y = y0 - 1 + hr, x = x0 - 4 + 4 * pc;
            voff[i] = (rem < 6 * R && y >= 0 && y < p.H && x >= 0 && x < p.W) ? (unsigned)ch * plane_b + (unsigned)(y * p.W + x) * 4u : OUTSIDE;
        }
        {
            const __amdgpu_buffer_rsrc_t irs = dma_rsrc(0);
#pragma unroll
            for (int i = 0; i < NDMA; ++i) dma_piece(IC<0>{}, irs, i);
        }
        auto do_group = [&](auto PAR_, int gi) {
            constexpr int PAR = decltype(PAR_)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this group's stage has landed (and its weights): nothing younger is in flight
            if (gi == 0) T16_STAMP(1);                           // the first group's loads have landed
            // The next group's loads (after the last group its own again, read by nobody) go out BETWEEN the MFMAs, one or two per
            // (k-step, halo row): a VMEM instruction holds its wave's issue for 60 - 100 cycles, and the two waves of a SIMD reach a
            // step's start together (spread over the first 3 / 5 / 7 tenths of the step instead of all of it: the same time).
            // They fill the stage group gi - 1 was read from (its reads fed MFMAs already issued).
            const int gn = gi + 1 < my ? gi + 1 : gi;
            const int off = woff(gn);
            const __amdgpu_buffer_rsrc_t irs = dma_rsrc(gn);
            const float* __restrict__ bs = bbase + PAR * T16_STAGE;
            constexpr int NL = 9 + NDMA, NJ = 4 * R;
            float bq[2][3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) bq[0][kx] = bs[kx];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {                        // j = (k-step s, halo row hr)
                const int s = j / R, hr = j % R;
                if (j + 1 < NJ) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) bq[(j + 1) & 1][kx] = bs[((j + 1) / R) * 2 * CP + ((j + 1) % R) * 24 + kx];
                }
#pragma unroll
                for (int l = j * NL / NJ; l < (j + 1) * NL / NJ; ++l) {
                    if (l < 9) wr[PAR ^ 1][l] = t16_ld_w(wrs, off, l * runs * 1024);
                    else dma_piece(IC<PAR ^ 1>{}, irs, l - 9);
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ti = 0; ti < NR; ++ti)
                        if (hr - ti >= 0 && hr - ti < 3) acc[ti] = MFMA16(wr[PAR][3 * (hr - ti) + kx][s], bq[j & 1][kx], acc[ti]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int gi = 0;
        for (; gi + 1 < my; gi += 2) {
            do_group(IC<0>{}, gi);
            do_group(IC<1>{}, gi + 1);
        }
        if (gi < my) do_group(IC<0>{}, gi);
        T16_STAMP(2);                                            // the last MFMA has been issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the surplus loads have landed: nothing arrives in this LDS later
        T16_STAMP(3);
    }
    // ---- the 8 partial sums of every row meet in LDS, over the stages once every wave has left them
    __syncthreads();
    T16_STAMP(4);                                                // every wave of the workgroup is here
#pragma unroll
    for (int ti = 0; ti < NR; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[((wave * T16_MAXR + ti) * 4 + r) * 64 + lane] = acc[ti][r];
    __syncthreads();
    const int r = er, l = el;
    const int x = x0 + (l & 15);
    for (int ti = (int)threadIdx.x >> 8; ti < NR; ti += 2) {
        float s = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < T16_WAVES; ++w8) s += lds[((w8 * T16_MAXR + ti) * 4 + r) * 64 + l];
        s += bias;
        if (p.relu) s = relu0(s);
        if (x < p.W) {
            const size_t o = (size_t)co * plane + (size_t)(y0 + ti) * p.W + x;
            if (p.res) s += p.res[(size_t)b * p.res_bs + o];
            p.out[(size_t)b * p.out_bs + o] = s;
        }
    }
    T16_STAMP(5);                                                // stores issued
#ifdef T16_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T16_STAMP(6);                                                // stores acknowledged
#endif
}

__global__ __launch_bounds__(512, 2) void conv_t16_kernel(const ConvT16Params p) {
    __shared__ __attribute__((aligned(16))) float lds[T16_LDS_FLOATS];
    const int quarter = (int)blockIdx.x & 3, slot = (int)blockIdx.x >> 2;
    const T16Share sh = t16_share(p, slot);
    if (sh.rows == 1) conv_t16_body<1>(p, lds, quarter, sh.b, sh.y0, sh.x0);
    else if (sh.rows == 2) conv_t16_body<2>(p, lds, quarter, sh.b, sh.y0, sh.x0);
    else if (sh.rows == 3) conv_t16_body<3>(p, lds, quarter, sh.b, sh.y0, sh.x0);  // (more than 3: refused at launch)
}

// ---- the local-fusion layers (1x1, Cin = 576; rdn.py:34) on the same maps: the same units, 4 MFMAs per row and group -- a
// latency problem, not an arithmetic one (170 MFLOP per layer at 48 x 48).  EVERY load of the workgroup goes out at once: a wave's
// <= 5 groups of 16 channels have a stage each (16 channels x rows x 16 columns by LDS-DMA, channel pitch 16 / 48 / 48 floats =
// 16 banks mod 32) and a 1 KiB weight run each in registers; bias and residual are asked for before that; one wait, <= 60 MFMAs,
// the reduction through LDS, two destinations.  The split-K kernel's 1x1 form stages a chunk, multiplies, stages the next.
constexpr int T16_1X1_MAXG = 5;                                // groups per wave: Cin <= 640
static_assert(T16_1X1_MAXG * T16_GROUP * 48 <= 2 * T16_STAGE, "the 1x1 stages fit the wave's share of the LDS");

template <int NR>
__device__ __forceinline__ void conv1x1_t16_body(const ConvT16Params& p, float* __restrict__ lds, int quarter, int b, int y0, int x0) {
    constexpr int CP = NR == 1 ? 16 : 48, PPC = CP / 4, NDMA = T16_GROUP * PPC / 64, STAGE = T16_GROUP * CP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_b = (unsigned)(plane * sizeof(float));
    const int G = p.Cin / T16_GROUP;
    const int my = wave < G ? (G - wave + T16_WAVES - 1) >> 3 : 0;        // this wave's groups: wave, wave + 8, ... (<= T16_1X1_MAXG)
    constexpr unsigned OUTSIDE = 0x80000000u;
    // the epilogue's thread: output co of column x, rows y0 + rsel (+ 2): bias and residual are on their way before anything else
    const int ev = (int)threadIdx.x & 255, er = ev >> 6, el = ev & 63, rsel = (int)threadIdx.x >> 8;
    const int co = 16 * quarter + 4 * (el >> 4) + er;
    const int x = x0 + (el & 15);
    const float bias = p.bias[co];
    float resv[2] = {0.0f, 0.0f};
    if (p.res && x < p.W) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (rsel + 2 * u < NR) resv[u] = p.res[(size_t)b * p.res_bs + (size_t)co * plane + (size_t)(y0 + rsel + 2 * u) * p.W + x];
    }
    f32x4 acc[NR];
#pragma unroll
    for (int ti = 0; ti < NR; ++ti) acc[ti] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (my > 0) {
        float* __restrict__ ring = lds + wave * 2 * T16_STAGE;
        const int runs = p.Cin / 64;
        const unsigned inv_runs = p.inv_runs;                     // r / runs = (r * inv_runs) >> 16 for r < 128
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.Cin * (64 * 4), 0x00020000);
        // the split-K kernel's 1x1 image: [half 2][wave 8][run Cin/64][h 2][i 32][4] (conv_t16_body's lane mapping, one tap)
        const int wlane = (quarter >> 1) * (8 * runs * 1024) + ((lane >> 4) & 1) * 512 + (16 * (quarter & 1) + (lane & 15)) * 16;
        f32x4 wr[T16_1X1_MAXG];
#pragma unroll
        for (int gi = 0; gi < T16_1X1_MAXG; ++gi)
            if (gi < my) {
                const unsigned r8 = 2u * (wave + T16_WAVES * gi) + (lane >> 5);
                const unsigned w8 = (r8 * inv_runs) >> 16, g = r8 - w8 * runs;
                wr[gi] = t16_ld_w(wrs, wlane + (int)((w8 * runs + g) * 1024), 0);
            }
        // piece i * 64 + lane of a stage: channel ch, row, piece pc of the row's 16 columns -- or padding
        unsigned voff[NDMA];
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int pi = i * 64 + lane;
            const int ch = pi / PPC, rem = pi - ch * PPC;
            const int row = rem >> 2, xx = x0 + 4 * (rem & 3);
            voff[i] = (row < NR && xx < p.W) ? (unsigned)ch * plane_b + (unsigned)((y0 + row) * p.W + xx) * 4u : OUTSIDE;
        }
        const float* in_b = p.in + (size_t)b * p.in_bs;
#pragma unroll
        for (int gi = 0; gi < T16_1X1_MAXG; ++gi)
            if (gi < my) {
                const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(in_b + (size_t)T16_GROUP * (wave + T16_WAVES * gi) * plane), 0, (int)((unsigned)T16_GROUP * plane_b), 0x00020000);
#pragma unroll
                for (int i = 0; i < NDMA; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(ring + gi * STAGE + i * 256), 16, (int)voff[i], 0, 0, T16_DMA_AUX);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every stage has landed
        const float* __restrict__ bbase = ring + (8 * (lane >> 5) + ((lane >> 4) & 1)) * CP + (lane & 15);
#pragma unroll
        for (int gi = 0; gi < T16_1X1_MAXG; ++gi)
            if (gi < my) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ti = 0; ti < NR; ++ti) acc[ti] = MFMA16(wr[gi][e], bbase[gi * STAGE + e * 2 * CP + ti * 16], acc[ti]);
            }
    }
    __syncthreads();                                             // the sums meet over the stages once every wave has left them
#pragma unroll
    for (int ti = 0; ti < NR; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[((wave * T16_MAXR + ti) * 4 + r) * 64 + lane] = acc[ti][r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ti = rsel + 2 * u;
        if (ti < NR) {
            float s = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < T16_WAVES; ++w8) s += lds[((w8 * T16_MAXR + ti) * 4 + er) * 64 + el];
            s += bias;
            if (p.relu) s = relu0(s);
            s += resv[u];
            if (x < p.W) {
                const size_t o = (size_t)co * plane + (size_t)(y0 + ti) * p.W + x;
                p.out[(size_t)b * p.out_bs + o] = s;
                if (p.out1) p.out1[(size_t)b * p.out1_bs + o] = s;
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void conv1x1_t16_kernel(const ConvT16Params p) {
    __shared__ __attribute__((aligned(16))) float lds[T16_LDS_FLOATS];
    const int quarter = (int)blockIdx.x & 3, slot = (int)blockIdx.x >> 2;
    const T16Share sh = t16_share(p, slot);
    if (sh.rows == 1) conv1x1_t16_body<1>(p, lds, quarter, sh.b, sh.y0, sh.x0);
    else if (sh.rows == 2) conv1x1_t16_body<2>(p, lds, quarter, sh.b, sh.y0, sh.x0);
    else if (sh.rows == 3) conv1x1_t16_body<3>(p, lds, quarter, sh.b, sh.y0, sh.x0);
}

#ifdef T16_STAMPS
static unsigned long long* g_t16_stamps = nullptr;               // set by tools/ubench/t16_bench.hip before each launch
#endif
// the launch's partition for a map of `strips` strips of H rows (workgroups per output quarter; 0: not a map for this kernel)
static int t16_plan(long long strips, int H, int W, int Cin, ConvT16Params* p) {
    const long long per_quarter = device_cus() / 4 > 0 ? device_cus() / 4 : 1;
    long long per_strip = per_quarter / strips;                  // workgroups a strip can have
    if (per_strip < 1 || strips > 65535) return 0;               // a strip would be left without a workgroup
    if (per_strip > H) per_strip = H;
    const long long rows = (H + per_strip - 1) / per_strip;      // the busiest workgroup's rows: the layer's time
    if (rows > T16_MAXR) return 0;
    // no more workgroups than that load asks for (48 x 48: 3 strips x 16 workgroups of exactly 3 rows = 192 of 256 CUs, not 3 x 21 of 2
    // or 3 rows): the busiest workgroup is the layer either way (measured: 1.651 / 1.649 against 1.655 / 1.633 ms per trunk), the idle CUs draw less
    per_strip = (H + rows - 1) / rows;
    if (p) {
        p->per_strip = (int)per_strip; p->rows = (int)rows; p->tiles_x = (W + 15) / 16;
        p->m_per_strip = (unsigned)((1ull << 32) / (unsigned long long)per_strip + 1);
        p->m_tiles_x = (unsigned)((1ull << 32) / (unsigned long long)p->tiles_x + 1);
        p->inv_runs = Cin >= 64 ? (65536u + (unsigned)(Cin / 64) - 1) / (unsigned)(Cin / 64) : 0;
    }
    return (int)(strips * per_strip);
}

extern "C" {

int diinn_conv_t16_applies(int B, int H, int W) {
    // the small-map kernel's maps: whole 16-byte pieces per row, at most T16_MAXR rows per workgroup, and fewer split-K units
    // (tiles of 8 x 4 pixels x 2 output halves) than compute units -- where that kernel leaves CUs idle
    if (B <= 0 || H <= 0 || W <= 0 || (W & 3) || knob(diinn_knobs().enc_no_t16) != 0) return 0;
    if ((long long)B * H * W >= knob(diinn_knobs().enc_wino_min)) return 0;
    if (t16_plan((long long)B * ((W + 15) / 16), H, W, 0, nullptr) == 0) return 0;
    const long long ks_units = 2LL * B * ((W + 7) / 8) * ((H + 3) / 4);
    return ks_units < device_cus();
}

int diinn_conv_t16_plan(int B, int H, int W, int info[4]) {
    if (!info || check_dims(B, H, W)) return DIINN_ERR_INVALID_ARG;
    ConvT16Params p;
    const int slots = t16_plan((long long)B * ((W + 15) / 16), H, W, 64, &p);
    info[0] = slots; info[1] = slots ? p.per_strip : 0; info[2] = slots ? p.rows : 0; info[3] = B * ((W + 15) / 16);
    return DIINN_OK;
}

int diinn_conv_t16(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                   const float* packed_w_dev, const float* bias_dev, const float* res_dev, long long res_batch_stride,
                   float* out_dev, long long out_batch_stride, int relu, int B, int H, int W) {
    if (!in_dev || !packed_w_dev || !bias_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin <= 0 || Cin % 64 || (W & 3)) return DIINN_ERR_UNSUPPORTED;
    if ((((size_t)in_dev) & 15) || (((size_t)packed_w_dev) & 15) || (in_batch_stride & 3)) return DIINN_ERR_INVALID_ARG;
    ConvT16Params p;
    const int slots = t16_plan((long long)B * ((W + 15) / 16), H, W, Cin, &p);
    if (slots == 0) return DIINN_ERR_UNSUPPORTED;                // a map for the other kernels
    if ((long long)H * W * 4 * T16_GROUP > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    p.in = in_dev; p.w = packed_w_dev; p.bias = bias_dev; p.res = res_dev; p.out = out_dev;
    p.in_bs = in_batch_stride; p.out_bs = out_batch_stride; p.res_bs = res_batch_stride;
    p.out1 = nullptr; p.out1_bs = 0;
    p.Cin = Cin; p.B = B; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
#ifdef T16_STAMPS
    p.stamps = g_t16_stamps;
#endif
    hipLaunchKernelGGL(conv_t16_kernel, dim3((unsigned)(4 * slots)), dim3(512), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_conv1x1_t16(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                      const float* packed_w_dev, const float* bias_dev, const float* res_dev, long long res_batch_stride,
                      float* out0_dev, long long out0_batch_stride, float* out1_dev, long long out1_batch_stride,
                      int relu, int B, int H, int W) {
    if (!in_dev || !packed_w_dev || !bias_dev || !out0_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin <= 0 || Cin % 64 || Cin > T16_1X1_MAXG * T16_WAVES * T16_GROUP || (W & 3)) return DIINN_ERR_UNSUPPORTED;
    if ((((size_t)in_dev) & 15) || (((size_t)packed_w_dev) & 15) || (in_batch_stride & 3)) return DIINN_ERR_INVALID_ARG;
    ConvT16Params p;
    const int slots = t16_plan((long long)B * ((W + 15) / 16), H, W, Cin, &p);
    if (slots == 0) return DIINN_ERR_UNSUPPORTED;                // a map for the other kernels
    if ((long long)H * W * 4 * T16_GROUP > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    p.in = in_dev; p.w = packed_w_dev; p.bias = bias_dev; p.res = res_dev; p.out = out0_dev; p.out1 = out1_dev;
    p.in_bs = in_batch_stride; p.out_bs = out0_batch_stride; p.out1_bs = out1_batch_stride; p.res_bs = res_batch_stride;
    p.Cin = Cin; p.B = B; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
#ifdef T16_STAMPS
    p.stamps = nullptr;
#endif
    hipLaunchKernelGGL(conv1x1_t16_kernel, dim3((unsigned)(4 * slots)), dim3(512), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

}  // extern "C"
