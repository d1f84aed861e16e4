// diinn_conv_x3.hip -- the RDN trunk's 3x3 convolutions in split-bf16 arithmetic on the bf16 MFMA (optional, large maps)
// (part of libdiinn_hip.so; shared definitions in diinn_device.h)
//
// Reference: src/models/components/rdn.py:9-35,90-105 -- the same 130 layers diinn_winograd.hip runs as fp32 Winograd
// (3x3, stride 1, zero padding 1, 64 outputs, 64..512 inputs).  Here the DIRECT sum is evaluated on
// v_mfma_f32_32x32x16_bf16 with every operand carried as hi + lo bf16 parts (hi = bf16(v), lo = bf16(v - hi)) and a
// product as  w_lo.x_hi + w_hi.x_lo + w_hi.x_hi  with fp32 accumulation: 2.25x the
// multiplies of F(2x2, 3x3), three MFMAs per product, at 16x the fp32 MFMA rate = 0.42 of the matrix-core time.
// Per layer the error against float64 is ~4e-6 of max|out| (fp32 Winograd: ~5e-7); through the whole trunk it does
// not pile up (7e-6 of max|feat|, 1e-7 in the decoded image against the 1e-4 bound: tools/enc_x3_error.py).
//
// Work split: a workgroup owns 32 x 8 output pixels and all 64 outputs; wave w owns pixel rows 2w, 2w+1 (two 32-pixel
// N-tiles x two 32-output M-tiles = 4 accumulators).  Per group of 16 input channels the 10 x 34 halo patch is read
// from the feature planes, split into hi / lo and staged in LDS in B-fragment order -- [part][k-half][row][col] x 16 B:
// the fragment of tap (ky, kx) for pixel (r, x) is the 16 bytes at (r + ky, x + kx), lane-linear, conflict-free --
// double-buffered: the planes of group g+1 are requested before group g's MFMAs and written behind them, one barrier
// per group.  Per (group, tap): 4 ds_read_b128 and 4 weight pieces of 1 KiB (hi / lo x two M-tiles, the same for
// every workgroup: L1 / L2 hits) feed 12 MFMAs.
#include "diinn_device.h"

constexpr int CX_TX = 32;                             // output pixels of a workgroup: 32 x (4 waves x CX_ROWS rows)
constexpr int CX_PW = CX_TX + 2;                      // its halo patch: CX_PW x (4 CX_ROWS + 2)

struct ConvX3Params {
    const float* in;         // input channel planes: in + b*in_bs + c*H*W
    const float* wx;         // packed weight: [group Cin/16][tap 9][M-tile 2][hi, lo][lane 64][8 bf16]
    const float* bias;       // [64]
    const float* res;        // optional residual [B,64,H,W] (batch stride res_bs), added after bias (no ReLU with it)
    float* out;              // out + b*out_bs + co*H*W
    long long in_bs, out_bs, res_bs;
    int Cin, B, H, W, relu;
    // the SPLIT activation format (optional): [b][group of 8 channels][hi, lo][H][W] x 16 bytes -- 8 bf16 per pixel, the
    // same 4 bytes per element as fp32, already in B-fragment order: staging a pixel is two 16-byte copies, no
    // conversion.  xs_in != null: the layer reads ALL its inputs from there (`in` is not touched); xs_out != null: the
    // outputs are written there as well (channel groups xs_out_g8 .. + 7) for the 3x3 layers that follow in the block
    const u32x4* xs_in;
    u32x4* xs_out;
    long long xs_bs;         // batch stride of both, in 16-byte units
    int xs_out_g8;
};

// CX_ROWS = pixel rows per wave.  2: every weight piece feeds 3 MFMAs on average (L1: 43 B/clk per CU), 256 workgroups
// on a 256x256 map; 1: twice the workgroups, 1.5 MFMAs per piece (maps of a few blocks only).
// WAHEAD: a whole group's 36 weight pieces live in registers (144) and each tap's are replaced by the next group's right
// behind their MFMAs -- for launches of about one workgroup per CU, where no other workgroup covers a wait: the
// vector-memory counter is in order, so a weight piece requested behind the patch loads is usable only once the patch
// (HBM latency) has arrived; one tap ahead that stalls every group, a group ahead it never does.
template <int CX_ROWS, bool SPLIT_IN, bool WAHEAD>
__global__ __launch_bounds__(256, WAHEAD ? 1 : 2) void conv3x3_x3_kernel(const ConvX3Params p) {
    constexpr int CX_TY = 4 * CX_ROWS, CX_PH = CX_TY + 2;
    constexpr int CX_TASKS = CX_PH * CX_PW * 2;           // staging tasks per group: (pixel, k-half) -> 8 channels
    constexpr int CX_ITERS = (CX_TASKS + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16x8 stage[2][2][2][CX_PH][CX_PW];   // [buffer][hi, lo][k-half][row][col]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, px = lane & 31;
    const int x0 = blockIdx.x * CX_TX, y0 = blockIdx.y * CX_TY, b = blockIdx.z;
    const size_t plane = (size_t)p.H * p.W;
    const float* __restrict__ in_b = p.in + (size_t)b * p.in_bs;
    const int ngroups = p.Cin / 16;

    // ---- staging tasks of this thread: task = k-half * (PH * PW) + row * PW + col
    int t_off[CX_ITERS];          // element offset inside a plane, or -1 outside the map (zero padding) / no task
    int t_lds[CX_ITERS];          // index into stage[buf][part] (in bf16x8 units), or -1
    int t_half[CX_ITERS];
#pragma unroll
    for (int i = 0; i < CX_ITERS; ++i) {
        const int task = tid + 256 * i;
        const bool has = task < CX_TASKS;
        const int kh = task / (CX_PH * CX_PW), rem = task - kh * (CX_PH * CX_PW);
        const int row = rem / CX_PW, col = rem - row * CX_PW;
        const int y = y0 + row - 1, x = x0 + col - 1;
        const bool inside = has && y >= 0 && y < p.H && x >= 0 && x < p.W;
        t_off[i] = inside ? y * p.W + x : -1;
        t_lds[i] = has ? (kh * CX_PH + row) * CX_PW + col : -1;
        t_half[i] = kh;
    }
    float sv[SPLIT_IN ? 1 : CX_ITERS][8];
    u32x4 xh_[SPLIT_IN ? CX_ITERS : 1], xl_[SPLIT_IN ? CX_ITERS : 1];
    const u32x4* __restrict__ xs_b = SPLIT_IN ? p.xs_in + (size_t)b * p.xs_bs : nullptr;
    auto stage_load = [&](const int g) {
        if constexpr (SPLIT_IN) {
#pragma unroll
            for (int i = 0; i < CX_ITERS; ++i) {
                const u32x4* __restrict__ src = xs_b + (size_t)(2 * (2 * g + t_half[i])) * plane;     // group 2g + k-half, hi plane
                const u32x4 z = {0u, 0u, 0u, 0u};
                xh_[i] = t_off[i] >= 0 ? src[t_off[i]] : z;
                xl_[i] = t_off[i] >= 0 ? src[plane + t_off[i]] : z;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CX_ITERS; ++i) {
                const float* __restrict__ src = in_b + (size_t)(16 * g + 8 * t_half[i]) * plane;
#pragma unroll
                for (int c = 0; c < 8; ++c) sv[i][c] = t_off[i] >= 0 ? src[(size_t)c * plane + t_off[i]] : 0.0f;
            }
        }
    };
    auto stage_store = [&](const int buf) {
        bf16x8* __restrict__ hi = &stage[buf][0][0][0][0];
        bf16x8* __restrict__ lo = &stage[buf][1][0][0][0];
#pragma unroll
        for (int i = 0; i < CX_ITERS; ++i) {
            if (t_lds[i] < 0) continue;
            if constexpr (SPLIT_IN) {
                hi[t_lds[i]] = __builtin_bit_cast(bf16x8, xh_[i]);
                lo[t_lds[i]] = __builtin_bit_cast(bf16x8, xl_[i]);
            } else {
                bf16x8 vh, vl;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const __bf16 a = (__bf16)sv[i][c];
                    vh[c] = a;
                    vl[c] = (__bf16)(sv[i][c] - (float)a);
                }
                hi[t_lds[i]] = vh;
                lo[t_lds[i]] = vl;
            }
        }
    };

    // ---- weights: 4 pieces per (group, tap): [M-tile][hi, lo], one tap ahead in a three-slot ring (tap t sits in slot
    // t % 3, and 9 % 3 == 0 keeps that true across groups).  Two or three workgroups share a CU: while one waits for its
    // patch (the vector-memory counter is in order: the first weight piece behind the patch loads waits for them too)
    // the others run their MFMAs.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wx, 0, (int)((size_t)ngroups * 9 * 4 * PIECE_BYTES), 0x00020000);
    const int lane_off = lane * 16;
    f32x4 A[WAHEAD ? 1 : 3][4];
    f32x4 Wg[WAHEAD ? 9 : 1][4];
    auto load_w = [&](const int slot, const int gt) {            // gt = group * 9 + tap
#pragma unroll
        for (int i = 0; i < 4; ++i) A[slot][i] = ld_piece(wrs, lane_off + i * PIECE_BYTES, gt * 4 * PIECE_BYTES);
    };
    auto load_tap = [&](auto tap_tag, const int g) {
        constexpr int T = decltype(tap_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) Wg[WAHEAD ? T : 0][i] = ld_piece(wrs, lane_off + i * PIECE_BYTES, (g * 9 + T) * 4 * PIECE_BYTES);
    };

    f32x16 acc[CX_ROWS][2];       // [pixel row of the wave][M-tile]
#pragma unroll
    for (int r = 0; r < CX_ROWS; ++r)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][mt][e] = 0.0f;

    if constexpr (WAHEAD) {
        // every tap's weights live in their own registers and are replaced by the NEXT group's right after their MFMAs:
        // those loads queue up behind the next group's patch loads, but nothing needs them for eight taps
        load_tap(IC<0>{}, 0); load_tap(IC<1>{}, 0); load_tap(IC<2>{}, 0); load_tap(IC<3>{}, 0); load_tap(IC<4>{}, 0);
        load_tap(IC<5>{}, 0); load_tap(IC<6>{}, 0); load_tap(IC<7>{}, 0); load_tap(IC<8>{}, 0);
        stage_load(0);
        stage_store(0);
        __syncthreads();
        for (int g = 0; g < ngroups; ++g) {
            const int buf = g & 1;
            const bool more = g + 1 < ngroups;
            if (more) stage_load(g + 1);
            const bf16x8* __restrict__ sh = &stage[buf][0][h][CX_ROWS * wave][px];
            const bf16x8* __restrict__ sl = &stage[buf][1][h][CX_ROWS * wave][px];
            // the B fragments of tap t+1 are read from LDS while tap t's MFMAs run (a register set each way)
            bf16x8 Bf[2][CX_ROWS][2];
            auto read_b = [&](auto tap_tag) {
                constexpr int tap = decltype(tap_tag)::value;
                constexpr int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int r = 0; r < CX_ROWS; ++r) {
                    Bf[tap & 1][r][0] = sh[(r + ky) * CX_PW + kx];
                    Bf[tap & 1][r][1] = sl[(r + ky) * CX_PW + kx];
                }
            };
            read_b(IC<0>{});
            auto tap_step = [&](auto tap_tag) {
                constexpr int tap = decltype(tap_tag)::value;
                if constexpr (tap < 8) read_b(IC<tap + 1>{});
                const bf16x8 wh0 = __builtin_bit_cast(bf16x8, Wg[WAHEAD ? tap : 0][0]), wl0 = __builtin_bit_cast(bf16x8, Wg[WAHEAD ? tap : 0][1]);
                const bf16x8 wh1 = __builtin_bit_cast(bf16x8, Wg[WAHEAD ? tap : 0][2]), wl1 = __builtin_bit_cast(bf16x8, Wg[WAHEAD ? tap : 0][3]);
#pragma unroll
                for (int r = 0; r < CX_ROWS; ++r) {
                    const bf16x8 xh = Bf[tap & 1][r][0], xl = Bf[tap & 1][r][1];
                    acc[r][0] = MFMA_BF16(wl0, xh, acc[r][0]);
                    acc[r][1] = MFMA_BF16(wl1, xh, acc[r][1]);
                    acc[r][0] = MFMA_BF16(wh0, xl, acc[r][0]);
                    acc[r][1] = MFMA_BF16(wh1, xl, acc[r][1]);
                    acc[r][0] = MFMA_BF16(wh0, xh, acc[r][0]);
                    acc[r][1] = MFMA_BF16(wh1, xh, acc[r][1]);
                }
                load_tap(tap_tag, more ? g + 1 : g);
                __builtin_amdgcn_sched_barrier(0);
            };
            tap_step(IC<0>{}); tap_step(IC<1>{}); tap_step(IC<2>{}); tap_step(IC<3>{}); tap_step(IC<4>{});
            tap_step(IC<5>{}); tap_step(IC<6>{}); tap_step(IC<7>{}); tap_step(IC<8>{});
            if (more) stage_store(buf ^ 1);
            __syncthreads();
        }
    } else {
        load_w(0, 0);
        stage_load(0);
        stage_store(0);
        __syncthreads();
        const int total = ngroups * 9;
        for (int g = 0; g < ngroups; ++g) {
            const int buf = g & 1;
            load_w(1, g * 9 + 1);
            if (g + 1 < ngroups) stage_load(g + 1);              // in flight behind this group's MFMAs
            const bf16x8* __restrict__ sh = &stage[buf][0][h][CX_ROWS * wave][px];
            const bf16x8* __restrict__ sl = &stage[buf][1][h][CX_ROWS * wave][px];
            bf16x8 Bf[2][CX_ROWS][2];                            // B fragments, one tap ahead
#pragma unroll
            for (int r = 0; r < CX_ROWS; ++r) {
                Bf[0][r][0] = sh[r * CX_PW];
                Bf[0][r][1] = sl[r * CX_PW];
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int gt = g * 9 + tap;
                if (tap < 8) {
                    const int ky = (tap + 1) / 3, kx = (tap + 1) - 3 * ky;
#pragma unroll
                    for (int r = 0; r < CX_ROWS; ++r) {
                        Bf[(tap + 1) & 1][r][0] = sh[(r + ky) * CX_PW + kx];
                        Bf[(tap + 1) & 1][r][1] = sl[(r + ky) * CX_PW + kx];
                    }
                }
                if (tap > 0) load_w((tap + 1) % 3, gt + 1 < total ? gt + 1 : gt);
                const bf16x8 wh0 = __builtin_bit_cast(bf16x8, A[tap % 3][0]), wl0 = __builtin_bit_cast(bf16x8, A[tap % 3][1]);
                const bf16x8 wh1 = __builtin_bit_cast(bf16x8, A[tap % 3][2]), wl1 = __builtin_bit_cast(bf16x8, A[tap % 3][3]);
#pragma unroll
                for (int r = 0; r < CX_ROWS; ++r) {
                    const bf16x8 xh = Bf[tap & 1][r][0], xl = Bf[tap & 1][r][1];
                    acc[r][0] = MFMA_BF16(wl0, xh, acc[r][0]);
                    acc[r][1] = MFMA_BF16(wl1, xh, acc[r][1]);
                    acc[r][0] = MFMA_BF16(wh0, xl, acc[r][0]);
                    acc[r][1] = MFMA_BF16(wh1, xl, acc[r][1]);
                    acc[r][0] = MFMA_BF16(wh0, xh, acc[r][0]);
                    acc[r][1] = MFMA_BF16(wh1, xh, acc[r][1]);
                }
                __builtin_amdgcn_sched_barrier(0);               // (384x384 trunk: 20.9 ms with the fence, 24.8 without)
            }
            if (g + 1 < ngroups) stage_store(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- bias, ReLU, residual, store: accumulator register e of M-tile mt is output channel 32 mt + 8 (e >> 2) + (e & 3) + 4 h
    const int x = x0 + px;
#pragma unroll
    for (int r = 0; r < CX_ROWS; ++r) {
        const int y = y0 + CX_ROWS * wave + r;
        if (y >= p.H || x >= p.W) continue;
        const size_t pix = (size_t)y * p.W + x;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = 32 * mt + 8 * (e >> 2) + (e & 3) + 4 * h;
                float v = acc[r][mt][e] + p.bias[co];
                if (p.relu) v = relu0(v);
                if (p.res) v += p.res[(size_t)b * p.res_bs + (size_t)co * plane + pix];
                if (p.out) __builtin_nontemporal_store(v, &p.out[(size_t)b * p.out_bs + (size_t)co * plane + pix]);
                acc[r][mt][e] = v;
            }
        if (p.xs_out) {
            // registers 4q .. 4q+3 of M-tile mt are channels 32 mt + 8 q + 4 h + (0..3): this lane's half (8 bytes hi, 8 bytes
            // lo) of the pixel's vector in channel group xs_out_g8 + 4 mt + q
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
            u32x2_* __restrict__ dst = reinterpret_cast<u32x2_*>(p.xs_out + (size_t)b * p.xs_bs);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 vh, vl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = acc[r][mt][4 * q + i];
                        const __bf16 a = (__bf16)v;
                        vh[i] = a;
                        vl[i] = (__bf16)(v - (float)a);
                    }
                    const size_t g8 = (size_t)(p.xs_out_g8 + 4 * mt + q);
                    dst[((2 * g8 + 0) * plane + pix) * 2 + h] = __builtin_bit_cast(u32x2_, vh);
                    dst[((2 * g8 + 1) * plane + pix) * 2 + h] = __builtin_bit_cast(u32x2_, vl);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv3x3_x3m_kernel: the dense-block form for launches of about one workgroup per CU (256x256: 256 blocks of 32 x 8
// pixels).  With four waves such a workgroup leaves every SIMD a single wave, whose instruction stream is serial: LDS
// reads, weight loads and the staging all add to the MFMAs' time (3.0 us per group of 16 channels against 1.65 us of MFMAs).
// Here the SAME block is worked by EIGHT waves -- waves 0..3 the first 32 outputs, waves 4..7 the other 32, each its two
// pixel rows -- so every SIMD holds two waves that cover each other's non-MFMA instructions.  Same MFMAs per SIMD, the B
// fragments are read twice (LDS: 48 B/clk per CU), a wave keeps its own M-tile's weights of a whole group in registers (72)
// and replaces each tap's behind its MFMAs (as WAHEAD above).  Split-format input only.
__global__ __launch_bounds__(512, 1) void conv3x3_x3m_kernel(const ConvX3Params p) {
    constexpr int CX_ROWS = 2, CX_TY = 8, CX_PH = CX_TY + 2;
    constexpr int CX_TASKS = CX_PH * CX_PW * 2;
    constexpr int CX_ITERS = (CX_TASKS + 511) / 512;
    __shared__ __attribute__((aligned(16))) bf16x8 stage[2][2][2][CX_PH][CX_PW];   // [buffer][hi, lo][k-half][row][col]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wave >> 2, rw = wave & 3;
    const int h = lane >> 5, px = lane & 31;
    const int x0 = blockIdx.x * CX_TX, y0 = blockIdx.y * CX_TY, b = blockIdx.z;
    const size_t plane = (size_t)p.H * p.W;
    const int ngroups = p.Cin / 16;
    int t_off[CX_ITERS], t_lds[CX_ITERS], t_half[CX_ITERS];
#pragma unroll
    for (int i = 0; i < CX_ITERS; ++i) {
        const int task = tid + 512 * i;
        const bool has = task < CX_TASKS;
        const int kh = task / (CX_PH * CX_PW), rem = task - kh * (CX_PH * CX_PW);
        const int row = rem / CX_PW, col = rem - row * CX_PW;
        const int y = y0 + row - 1, x = x0 + col - 1;
        const bool inside = has && y >= 0 && y < p.H && x >= 0 && x < p.W;
        t_off[i] = inside ? y * p.W + x : -1;
        t_lds[i] = has ? (kh * CX_PH + row) * CX_PW + col : -1;
        t_half[i] = kh;
    }
    u32x4 xh_[CX_ITERS], xl_[CX_ITERS];
    const u32x4* __restrict__ xs_b = p.xs_in + (size_t)b * p.xs_bs;
    auto stage_load = [&](const int g) {
#pragma unroll
        for (int i = 0; i < CX_ITERS; ++i) {
            const u32x4* __restrict__ src = xs_b + (size_t)(2 * (2 * g + t_half[i])) * plane;
            const u32x4 z = {0u, 0u, 0u, 0u};
            xh_[i] = t_off[i] >= 0 ? src[t_off[i]] : z;
            xl_[i] = t_off[i] >= 0 ? src[plane + t_off[i]] : z;
        }
    };
    auto stage_store = [&](const int buf) {
        bf16x8* __restrict__ hi = &stage[buf][0][0][0][0];
        bf16x8* __restrict__ lo = &stage[buf][1][0][0][0];
#pragma unroll
        for (int i = 0; i < CX_ITERS; ++i) {
            if (t_lds[i] < 0) continue;
            hi[t_lds[i]] = __builtin_bit_cast(bf16x8, xh_[i]);
            lo[t_lds[i]] = __builtin_bit_cast(bf16x8, xl_[i]);
        }
    };
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wx, 0, (int)((size_t)ngroups * 9 * 4 * PIECE_BYTES), 0x00020000);
    const int lane_off = lane * 16 + mt * 2 * PIECE_BYTES;       // this wave's M-tile: pieces 2 mt (hi), 2 mt + 1 (lo)
    f32x4 Wg[9][2];
    auto load_tap = [&](auto tap_tag, const int g) {
        constexpr int T = decltype(tap_tag)::value;
        Wg[T][0] = ld_piece(wrs, lane_off, (g * 9 + T) * 4 * PIECE_BYTES);
        Wg[T][1] = ld_piece(wrs, lane_off + PIECE_BYTES, (g * 9 + T) * 4 * PIECE_BYTES);
    };
    // the accumulators start from the bias (requested with the first loads of the kernel, not at its end)
    f32x16 acc[CX_ROWS];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float bv = p.bias[32 * mt + 8 * (e >> 2) + (e & 3) + 4 * h];
#pragma unroll
        for (int r = 0; r < CX_ROWS; ++r) acc[r][e] = bv;
    }

    load_tap(IC<0>{}, 0); load_tap(IC<1>{}, 0); load_tap(IC<2>{}, 0); load_tap(IC<3>{}, 0); load_tap(IC<4>{}, 0);
    load_tap(IC<5>{}, 0); load_tap(IC<6>{}, 0); load_tap(IC<7>{}, 0); load_tap(IC<8>{}, 0);
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        const int buf = g & 1;
        const bool more = g + 1 < ngroups;
        if (more) stage_load(g + 1);
        const bf16x8* __restrict__ sh = &stage[buf][0][h][CX_ROWS * rw][px];
        const bf16x8* __restrict__ sl = &stage[buf][1][h][CX_ROWS * rw][px];
        bf16x8 Bf[2][CX_ROWS][2];
        auto read_b = [&](auto tap_tag) {
            constexpr int tap = decltype(tap_tag)::value;
            constexpr int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int r = 0; r < CX_ROWS; ++r) {
                Bf[tap & 1][r][0] = sh[(r + ky) * CX_PW + kx];
                Bf[tap & 1][r][1] = sl[(r + ky) * CX_PW + kx];
            }
        };
        read_b(IC<0>{});
        auto tap_step = [&](auto tap_tag) {
            constexpr int tap = decltype(tap_tag)::value;
            if constexpr (tap < 8) read_b(IC<tap + 1>{});
            const bf16x8 wh = __builtin_bit_cast(bf16x8, Wg[tap][0]), wl = __builtin_bit_cast(bf16x8, Wg[tap][1]);
            acc[0] = MFMA_BF16(wl, Bf[tap & 1][0][0], acc[0]);
            acc[1] = MFMA_BF16(wl, Bf[tap & 1][1][0], acc[1]);
            acc[0] = MFMA_BF16(wh, Bf[tap & 1][0][1], acc[0]);
            acc[1] = MFMA_BF16(wh, Bf[tap & 1][1][1], acc[1]);
            acc[0] = MFMA_BF16(wh, Bf[tap & 1][0][0], acc[0]);
            acc[1] = MFMA_BF16(wh, Bf[tap & 1][1][0], acc[1]);
            load_tap(tap_tag, more ? g + 1 : g);
            __builtin_amdgcn_sched_barrier(0);
        };
        tap_step(IC<0>{}); tap_step(IC<1>{}); tap_step(IC<2>{}); tap_step(IC<3>{}); tap_step(IC<4>{});
        tap_step(IC<5>{}); tap_step(IC<6>{}); tap_step(IC<7>{}); tap_step(IC<8>{});
        if (more) stage_store(buf ^ 1);
        __syncthreads();
    }

    const int x = x0 + px;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int r = 0; r < CX_ROWS; ++r) {
        const int y = y0 + CX_ROWS * rw + r;
        if (y >= p.H || x >= p.W) continue;
        const size_t pix = (size_t)y * p.W + x;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = 32 * mt + 8 * (e >> 2) + (e & 3) + 4 * h;
            float v = acc[r][e];
            if (p.relu) v = relu0(v);
            if (p.res) v += p.res[(size_t)b * p.res_bs + (size_t)co * plane + pix];
            if (p.out) __builtin_nontemporal_store(v, &p.out[(size_t)b * p.out_bs + (size_t)co * plane + pix]);
            acc[r][e] = v;
        }
        if (p.xs_out) {
            u32x2_* __restrict__ dst = reinterpret_cast<u32x2_*>(p.xs_out + (size_t)b * p.xs_bs);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x4 vh, vl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[r][4 * q + i];
                    const __bf16 a = (__bf16)v;
                    vh[i] = a;
                    vl[i] = (__bf16)(v - (float)a);
                }
                const size_t g8 = (size_t)(p.xs_out_g8 + 4 * mt + q);
                __builtin_nontemporal_store(__builtin_bit_cast(u32x2_, vh), &dst[((2 * g8 + 0) * plane + pix) * 2 + h]);
                __builtin_nontemporal_store(__builtin_bit_cast(u32x2_, vl), &dst[((2 * g8 + 1) * plane + pix) * 2 + h]);
            }
        }
    }
}

// (The outputs of both kernels leave through non-temporal stores: as plain stores they sit dirty in the L2 until the
// kernel's end-of-kernel write-back -- per trunk at 256x256 8.30 -> 8.00 ms, 512x512 32.5 -> 32.2.  The fp32 Winograd kernel
// does not gain from the same change: 11.77 vs 11.75 ms.)

// ---------------------------------------------------------------------------------------------------------------------
// conv1x1_x3_kernel: a block's 1x1 fusion layer (rdn.py:34, 576 -> 64 plus the block input) in the same arithmetic, reading
// the split format directly: without a halo the B fragment of pixel x, k-half h, channel group g IS the 16-byte vector
// stored for it, so a lane loads its fragments straight from memory (512 contiguous bytes per half wave) -- no LDS, no
// barrier, the waves are independent.  A wave owns 64 consecutive pixels (two N-tiles) x 64 outputs; per group of 16
// channels 4 fragment loads + 4 weight pieces feed 12 MFMAs, one group ahead in registers (L1-bound: 8 KiB per 12 MFMAs
// and wave; as fast as the fp32 streaming kernel it replaces -- what it buys is that the dense layers need not write
// their outputs as fp32 planes any more, half of their store burst).  Outputs: the planes of the next block's input and of
// the global fusion input (fp32, residual added from the block input's planes), and the next block's channel groups 0..7
// in the split format IN PLACE (a pixel's vectors are read and written by the same lane only).
struct Conv1X3Params {
    u32x4* xs;               // [b][72 groups][hi, lo][H*W] x 16 bytes
    long long xs_bs;
    const float* wx;         // [group Cin/16][M-tile 2][hi, lo][lane 64][8 bf16]
    const float* bias;
    const float* res;        // block input planes [B,64,H,W] (batch stride res_bs)
    float* o0;               // next block's input planes
    float* o1;               // slice of the global fusion input
    long long res_bs, o0_bs, o1_bs;
    int G, B, H, W;
};

__global__ __launch_bounds__(256, 2) void conv1x1_x3_kernel(const Conv1X3Params p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, px = lane & 31;
    const int b = blockIdx.y;
    const size_t plane = (size_t)p.H * p.W;
    const size_t base = (size_t)blockIdx.x * 256 + 64 * wave;
    if (base >= plane) return;
    size_t pix[2];
    bool in[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const size_t q = base + 32 * t + px;
        in[t] = q < plane;
        pix[t] = in[t] ? q : plane - 1;
    }
    const u32x4* __restrict__ xs_b = p.xs + (size_t)b * p.xs_bs;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wx, 0, p.G * 4 * PIECE_BYTES, 0x00020000);
    const int lane_off = lane * 16;
    u32x4 Bf[2][2][2];            // [set][tile][hi, lo]
    f32x4 Wt[2][4];
    auto fetch = [&](const int set, const int g) {
        const u32x4* __restrict__ src = xs_b + (size_t)(2 * (2 * g + h)) * plane;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Bf[set][t][0] = src[pix[t]];
            Bf[set][t][1] = src[plane + pix[t]];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) Wt[set][i] = ld_piece(wrs, lane_off + i * PIECE_BYTES, g * 4 * PIECE_BYTES);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float bv = p.bias[32 * mt + 8 * (e >> 2) + (e & 3) + 4 * h];
            acc[0][mt][e] = bv;
            acc[1][mt][e] = bv;
        }
    auto mma = [&](const int set) {
        const bf16x8 wh0 = __builtin_bit_cast(bf16x8, Wt[set][0]), wl0 = __builtin_bit_cast(bf16x8, Wt[set][1]);
        const bf16x8 wh1 = __builtin_bit_cast(bf16x8, Wt[set][2]), wl1 = __builtin_bit_cast(bf16x8, Wt[set][3]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, Bf[set][t][0]), xl = __builtin_bit_cast(bf16x8, Bf[set][t][1]);
            acc[t][0] = MFMA_BF16(wl0, xh, acc[t][0]);
            acc[t][1] = MFMA_BF16(wl1, xh, acc[t][1]);
            acc[t][0] = MFMA_BF16(wh0, xl, acc[t][0]);
            acc[t][1] = MFMA_BF16(wh1, xl, acc[t][1]);
            acc[t][0] = MFMA_BF16(wh0, xh, acc[t][0]);
            acc[t][1] = MFMA_BF16(wh1, xh, acc[t][1]);
        }
    };
    fetch(0, 0);
    for (int g = 0; g < p.G; g += 2) {                           // G is even (576 / 16 = 36)
        fetch(1, g + 1);
        mma(0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(0, g + 2 < p.G ? g + 2 : g + 1);
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_* __restrict__ dst = reinterpret_cast<u32x2_*>(p.xs + (size_t)b * p.xs_bs);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (!in[t]) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = 32 * mt + 8 * (e >> 2) + (e & 3) + 4 * h;
                const float v = acc[t][mt][e] + p.res[(size_t)b * p.res_bs + (size_t)co * plane + pix[t]];
                __builtin_nontemporal_store(v, &p.o0[(size_t)b * p.o0_bs + (size_t)co * plane + pix[t]]);
                __builtin_nontemporal_store(v, &p.o1[(size_t)b * p.o1_bs + (size_t)co * plane + pix[t]]);
                acc[t][mt][e] = v;
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x4 vh, vl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[t][mt][4 * q + i];
                    const __bf16 a = (__bf16)v;
                    vh[i] = a;
                    vl[i] = (__bf16)(v - (float)a);
                }
                const size_t g8 = (size_t)(4 * mt + q);
                __builtin_nontemporal_store(__builtin_bit_cast(u32x2_, vh), &dst[((2 * g8 + 0) * plane + pix[t]) * 2 + h]);
                __builtin_nontemporal_store(__builtin_bit_cast(u32x2_, vl), &dst[((2 * g8 + 1) * plane + pix[t]) * 2 + h]);
            }
    }
}

extern "C" {

static int launch_conv_x3(void* stream, const ConvX3Params& p) {
    // two pixel rows per wave (every weight piece feeds 3 MFMAs).  Below two such workgroups per CU the eight-wave form
    // (conv3x3_x3m_kernel; per trunk 192x192 8.2 -> 7.2 ms, 256x256 9.1 -> 8.8, 320x320 16.2 -> 14.7; from 384x384 on the
    // four-wave ring form is faster: 21.3 vs 22.5, 512x512 32.9 vs 33.7).  DIINN_ENC_X3_ROWS forces 1 row / 2 rows (ring) /
    // 3 = four waves + a group's weights in registers (layers not in the split format use this one) / 4 = eight waves
    const long long wg2 = (long long)((p.W + CX_TX - 1) / CX_TX) * ((p.H + 7) / 8) * p.B;
    const long long force = knob(diinn_knobs().enc_x3_rows);
    const bool two = force ? force >= 2 : wg2 >= 64;
    const bool ahead = force ? force >= 3 : (two && wg2 < 512);
    const dim3 grid((unsigned)((p.W + CX_TX - 1) / CX_TX), (unsigned)((p.H + (two ? 7 : 3)) / (two ? 8 : 4)), (unsigned)p.B);
    const hipStream_t st = (hipStream_t)stream;
    if (p.xs_in) {
        if (ahead && force != 3) hipLaunchKernelGGL(conv3x3_x3m_kernel, grid, dim3(512), 0, st, p);   // eight waves: two per SIMD
        else if (ahead) hipLaunchKernelGGL((conv3x3_x3_kernel<2, true, true>), grid, dim3(256), 0, st, p);
        else if (two) hipLaunchKernelGGL((conv3x3_x3_kernel<2, true, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3_x3_kernel<1, true, false>), grid, dim3(256), 0, st, p);
    } else {
        if (ahead) hipLaunchKernelGGL((conv3x3_x3_kernel<2, false, true>), grid, dim3(256), 0, st, p);
        else if (two) hipLaunchKernelGGL((conv3x3_x3_kernel<2, false, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3_x3_kernel<1, false, false>), grid, dim3(256), 0, st, p);
    }
    return hip_status(hipGetLastError());
}

// one 3x3 layer (64 outputs, Cin a multiple of 16) in split-bf16 arithmetic; `wx` as diinn_rdn_x3_packed_floats describes
int diinn_conv3x3_x3(void* stream, const float* in_dev, long long in_bs, int Cin, const float* wx_dev, const float* bias_dev,
                     const float* res_dev, long long res_bs, float* out_dev, long long out_bs, int relu, int B, int H, int W) {
    if (!in_dev || !wx_dev || !bias_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin < 16 || Cin % 16 || Cin > 1024) return DIINN_ERR_UNSUPPORTED;
    if ((long long)H * W > 0x7fffffffLL / 4) return DIINN_ERR_TOO_LARGE;
    ConvX3Params p{in_dev, wx_dev, bias_dev, res_dev, out_dev, in_bs, out_bs, res_bs, Cin, B, H, W, relu, nullptr, nullptr, 0, 0};
    return launch_conv_x3(stream, p);
}

// the same layer inside the trunk (diinn_rdn_forward_x3): inputs from the planes `in_dev`, or -- in_dev null -- all Cin channels
// from the split-format buffer xs; outputs to the planes `out_dev` (may be null) and, when xs_out_g8 >= 0, to channel groups
// xs_out_g8 .. + 7 of xs
__attribute__((visibility("hidden")))
int diinn_conv3x3_x3_split(void* stream, const float* in_dev, long long in_bs, float* xs_dev, long long xs_bs16, int xs_out_g8,
                           int Cin, const float* wx_dev, const float* bias_dev, float* out_dev, long long out_bs, int relu,
                           int B, int H, int W) {
    if (Cin < 16 || Cin % 16 || Cin > 1024) return DIINN_ERR_UNSUPPORTED;
    if ((long long)H * W > 0x7fffffffLL / 4) return DIINN_ERR_TOO_LARGE;
    ConvX3Params p{in_dev, wx_dev, bias_dev, nullptr, out_dev, in_bs, out_bs, 0, Cin, B, H, W, relu,
                   in_dev ? nullptr : reinterpret_cast<const u32x4*>(xs_dev),
                   xs_out_g8 >= 0 ? reinterpret_cast<u32x4*>(xs_dev) : nullptr, xs_bs16, xs_out_g8 >= 0 ? xs_out_g8 : 0};
    return launch_conv_x3(stream, p);
}

// a block's 1x1 fusion layer from the split-format buffer (Cin % 32 == 0): see conv1x1_x3_kernel
__attribute__((visibility("hidden")))
int diinn_conv1x1_x3_split(void* stream, float* xs_dev, long long xs_bs16, int Cin, const float* wx_dev, const float* bias_dev,
                           const float* res_dev, long long res_bs, float* o0_dev, long long o0_bs, float* o1_dev, long long o1_bs,
                           int B, int H, int W) {
    if (Cin < 32 || Cin % 32 || Cin > 1024) return DIINN_ERR_UNSUPPORTED;
    Conv1X3Params p{reinterpret_cast<u32x4*>(xs_dev), xs_bs16, wx_dev, bias_dev, res_dev, o0_dev, o1_dev, res_bs, o0_bs, o1_bs,
                    Cin / 16, B, H, W};
    const long long plane = (long long)H * W;
    hipLaunchKernelGGL(conv1x1_x3_kernel, dim3((unsigned)((plane + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

}  // extern "C"
