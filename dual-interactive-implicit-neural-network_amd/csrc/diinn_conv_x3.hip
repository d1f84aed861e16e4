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
};

// CX_ROWS = pixel rows per wave.  2: every weight piece feeds 3 MFMAs on average (L1: 43 B/clk per CU), 256 workgroups
// on a 256x256 map; 1: twice the workgroups (two per CU there, which cover each other's patch loads), 1.5 MFMAs per
// piece.  Measured per layer at 256x256 (tools/conv_x3_time.py): 64 inputs 25.9 / 29.5 us, 512 inputs 131.4 / 124.0 us.
template <int CX_ROWS>
__global__ __launch_bounds__(256, 2) void conv3x3_x3_kernel(const ConvX3Params p) {
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
    float sv[CX_ITERS][8];
    auto stage_load = [&](const int g) {
#pragma unroll
        for (int i = 0; i < CX_ITERS; ++i) {
            const float* __restrict__ src = in_b + (size_t)(16 * g + 8 * t_half[i]) * plane;
#pragma unroll
            for (int c = 0; c < 8; ++c) sv[i][c] = t_off[i] >= 0 ? src[(size_t)c * plane + t_off[i]] : 0.0f;
        }
    };
    auto stage_store = [&](const int buf) {
        bf16x8* __restrict__ hi = &stage[buf][0][0][0][0];
        bf16x8* __restrict__ lo = &stage[buf][1][0][0][0];
#pragma unroll
        for (int i = 0; i < CX_ITERS; ++i) {
            if (t_lds[i] < 0) continue;
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
    };

    // ---- weights: 4 pieces per (group, tap): [M-tile][hi, lo], one tap ahead in a three-slot ring (tap t sits in slot
    // t % 3, and 9 % 3 == 0 keeps that true across groups).  Two or three workgroups share a CU: while one waits for its
    // patch (the vector-memory counter is in order: the first weight piece behind the patch loads waits for them too)
    // the others run their MFMAs.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wx, 0, (int)((size_t)ngroups * 9 * 4 * PIECE_BYTES), 0x00020000);
    const int lane_off = lane * 16;
    f32x4 A[3][4];
    auto load_w = [&](const int slot, const int gt) {            // gt = group * 9 + tap
#pragma unroll
        for (int i = 0; i < 4; ++i) A[slot][i] = ld_piece(wrs, lane_off + i * PIECE_BYTES, gt * 4 * PIECE_BYTES);
    };

    f32x16 acc[CX_ROWS][2];       // [pixel row of the wave][M-tile]
#pragma unroll
    for (int r = 0; r < CX_ROWS; ++r)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][mt][e] = 0.0f;

    load_w(0, 0);
    stage_load(0);
    stage_store(0);
    __syncthreads();
    const int total = ngroups * 9;
    for (int g = 0; g < ngroups; ++g) {
        const int buf = g & 1;
        load_w(1, g * 9 + 1);
        if (g + 1 < ngroups) stage_load(g + 1);                  // in flight behind this group's MFMAs
        const bf16x8* __restrict__ sh = &stage[buf][0][h][CX_ROWS * wave][px];
        const bf16x8* __restrict__ sl = &stage[buf][1][h][CX_ROWS * wave][px];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int gt = g * 9 + tap;
            if (tap > 0) load_w((tap + 1) % 3, gt + 1 < total ? gt + 1 : gt);
            const bf16x8 wh0 = __builtin_bit_cast(bf16x8, A[tap % 3][0]), wl0 = __builtin_bit_cast(bf16x8, A[tap % 3][1]);
            const bf16x8 wh1 = __builtin_bit_cast(bf16x8, A[tap % 3][2]), wl1 = __builtin_bit_cast(bf16x8, A[tap % 3][3]);
#pragma unroll
            for (int r = 0; r < CX_ROWS; ++r) {
                const bf16x8 xh = sh[(r + ky) * CX_PW + kx], xl = sl[(r + ky) * CX_PW + kx];
                acc[r][0] = MFMA_BF16(wl0, xh, acc[r][0]);
                acc[r][1] = MFMA_BF16(wl1, xh, acc[r][1]);
                acc[r][0] = MFMA_BF16(wh0, xl, acc[r][0]);
                acc[r][1] = MFMA_BF16(wh1, xl, acc[r][1]);
                acc[r][0] = MFMA_BF16(wh0, xh, acc[r][0]);
                acc[r][1] = MFMA_BF16(wh1, xh, acc[r][1]);
            }
        }
        if (g + 1 < ngroups) stage_store(buf ^ 1);
        __syncthreads();
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
                p.out[(size_t)b * p.out_bs + (size_t)co * plane + pix] = v;
            }
    }
}

extern "C" {

// one 3x3 layer (64 outputs, Cin a multiple of 16) in split-bf16 arithmetic; `wx` as diinn_conv_x3_packed_floats describes
int diinn_conv3x3_x3(void* stream, const float* in_dev, long long in_bs, int Cin, const float* wx_dev, const float* bias_dev,
                     const float* res_dev, long long res_bs, float* out_dev, long long out_bs, int relu, int B, int H, int W) {
    if (!in_dev || !wx_dev || !bias_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin < 16 || Cin % 16 || Cin > 1024) return DIINN_ERR_UNSUPPORTED;
    if ((long long)H * W > 0x7fffffffLL / 4) return DIINN_ERR_TOO_LARGE;
    ConvX3Params p{in_dev, wx_dev, bias_dev, res_dev, out_dev, in_bs, out_bs, res_bs, Cin, B, H, W, relu};
    // two rows per wave where the chip stays full with them or the layer is long; DIINN_ENC_X3_ROWS forces 1 / 2
    const long long wg2 = (long long)((W + CX_TX - 1) / CX_TX) * ((H + 7) / 8) * B;
    const long long force = knob(diinn_knobs().enc_x3_rows);
    const bool two = force ? force == 2 : (wg2 >= 512 || Cin > 192);
    if (two) {
        const dim3 grid((unsigned)((W + CX_TX - 1) / CX_TX), (unsigned)((H + 7) / 8), (unsigned)B);
        hipLaunchKernelGGL(conv3x3_x3_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        const dim3 grid((unsigned)((W + CX_TX - 1) / CX_TX), (unsigned)((H + 3) / 4), (unsigned)B);
        hipLaunchKernelGGL(conv3x3_x3_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    return hip_status(hipGetLastError());
}

}  // extern "C"
