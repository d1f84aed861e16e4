// conv_s16_kernel.hip -- round 6, NEGATIVE RESULT, not part of the library (kept with its integration patch,
// conv_s16_integration.patch: C ABI entry, trunk dispatch, packing, tests -- all of which passed on the GPU).
//
// Question: can the 3x3 layers of SMALL maps (the reference's own timing protocol, runtime_test.py:13,31-33:
// 1 x 3 x 48 x 48, where the trunk is 96 % of a forward) be made faster by a different unit of work?  conv_ksplit_kernel gives a
// 48 x 48 map 144 workgroups of (32 pixels, 32 outputs, all of Cin); a layer costs one such workgroup's time on its CU.
//
// This kernel: workgroups of (64 pixels = 4 rows x 16 columns, 16 outputs, all of Cin split over 8 waves) on
// v_mfma_f32_16x16x4_f32, direct fp32 sum; inputs by wave-private LDS-DMA (a group of 16 channels x 6 halo rows x 24 columns
// = 576 16-byte pieces = exactly 9 wave instructions whose lane-linear deposit IS the conflict-free layout the MFMAs read:
// channel pitch 144 floats = 16 banks mod 32), two stages per wave refilled behind the wave's own counted vmcnt, weights
// [quarter][group][tap][lane][4] double-buffered in registers a whole group ahead, one ds_read_b32 per (k-step, halo row, tap
// column) feeding up to three output rows, one LDS reduction of the 8 partial sums at the end.  140 VGPRs, no scratch.
// Correct: 4.2e-7 of max|out| against float64 on 15 shapes (every Cin of a dense block, ragged blocks, batches, strided
// views), bit-identical over 20 launches, whole trunk equal to the split-K trunk within 4e-6.
//
// Result (profiles/r06_small_map_s16.txt): NOT faster -- 48 x 48: 6.6 / 8.7 / 13.4 / 20.9 us at Cin 64 / 128 / 256 / 448 against
// the shipped kernel's 6.8 / 8.3 / 12.8 / 20.0; 64 x 64 the same picture.  Why, in one line: 64 pixels x 16 outputs IS 32
// pixels x 32 outputs -- the unit count (144) and therefore the MFMA work per workgroup (9.4 MFLOP at Cin 512 = 15.4 us at one
// CU's fp32 MFMA peak) do not change; only the intake mix does (295 + 295 KB against 590 + 123 KB), and the shipped kernel
// is not intake-bound.  (The plan this was built on took the unit for half the arithmetic: an error in the estimate, found
// by the measurement.)  What would move small maps is MORE units than CUs with balanced load, which at 2,304 pixels means
// units of 16 pixels x 16 outputs (576: 3 on the busiest CU = 0.75 of today's MFMA time) whose weight + input intake
// (3 x 369 KB per busiest CU at the 46-70 GB/s a CU gets from its L2 = 16-24 us) costs more than the MFMA time saved, or a
// reduction split across CUs, which round 5 measured (8-18 us per layer of slab hand-off).  A persistent trunk kernel with
// neighbour flags instead of 147 kernel boundaries keeps the dependent chain per layer (flag ~1 us, input latency ~1 us,
// reduction 0.5, store drain ~1): ~3.5 us against today's ~4.9 us of per-layer overhead = ~0.2 ms of 1.95 ms.  Not built.
#include "diinn_device.h"

constexpr int S16_WAVES = 8;
constexpr int S16_HROWS = 6, S16_PIECES = 6;                   // halo rows; 16-byte pieces per halo row
constexpr int S16_CP = S16_HROWS * S16_PIECES * 4;             // 144 floats per staged channel
constexpr int S16_GROUP = 16;                                  // input channels per stage / weight run
constexpr int S16_STAGE = S16_GROUP * S16_CP;                  // 2,304 floats = 9 KiB = 9 LDS-DMA instructions
constexpr int S16_LDS_FLOATS = S16_WAVES * 2 * S16_STAGE;      // 147,456 bytes
static_assert(S16_STAGE * 4 == 9 * 1024, "a stage is nine 1 KiB deposits");
static_assert(S16_WAVES * 16 * 64 <= S16_LDS_FLOATS, "the reduction buffer lies over the stages");
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct ConvS16Params {
    const float* in;         // input channel planes: in + b*in_bs + c*H*W
    const float* w;          // packed: [quarter 4][group Cin/16][tap 9][lane 64][4]
    const float* bias;       // [64]
    const float* res;        // optional residual [B,64,H,W] (batch stride res_bs), added after bias (no ReLU with it)
    float* out;              // out + b*out_bs + co*H*W
    long long in_bs, out_bs, res_bs;
    int Cin, B, H, W, relu;
};

__global__ __launch_bounds__(512, 2) void conv_s16_kernel(const ConvS16Params p) {
    __shared__ __attribute__((aligned(16))) float lds[S16_LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blocks_x = (p.W + 15) >> 4, blocks_y = (p.H + 3) >> 2;
    // every XCD takes a contiguous run of units (blocks b and b + 8 share an L2); the four output quarters of a pixel block
    // are neighbours in that run, so the second to fourth reader of a block's input rows find them in the L2
    const int per_xcd = (int)gridDim.x >> 3;                   // a multiple of 4
    int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int quarter = t & 3;
    t >>= 2;
    if (t >= p.B * blocks_x * blocks_y) return;
    const int b = __builtin_amdgcn_readfirstlane(t / (blocks_x * blocks_y));
    t -= b * blocks_x * blocks_y;
    const int by = t / blocks_x, bx = t - by * blocks_x;
    const int y0 = 4 * by, x0 = 16 * bx;
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_b = (unsigned)(plane * sizeof(float));
    const int G = p.Cin / S16_GROUP;                            // groups of 16 input channels
    const int my = wave < G ? (G - wave + S16_WAVES - 1) / S16_WAVES : 0;   // this wave's groups: wave, wave + 8, ...
    constexpr unsigned OUTSIDE = 0x80000000u;

    f32x4 acc[4];                                              // output row t of the block: D[out 4 (lane >> 4) + r][column lane & 15]
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    if (my > 0) {
        float* __restrict__ stage0 = lds + wave * 2 * S16_STAGE;
        const float* __restrict__ in_b = p.in + (size_t)b * p.in_bs;
        // piece i * 64 + lane of a group's 576: (channel, halo row, piece of the row) -> where it comes from
        unsigned voff[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int pi = i * 64 + lane;
            const int ch = pi / 36, rem = pi - ch * 36;
            const int hr = rem / 6, pc = rem - hr * 6;
            const int y = y0 - 1 + hr, x = x0 - 4 + 4 * pc;
            voff[i] = (y >= 0 && y < p.H && x >= 0 && x < p.W) ? (unsigned)ch * plane_b + (unsigned)(y * p.W + x) * 4u : OUTSIDE;
        }
        auto dma = [&](int st, int g) {
            const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(in_b + (size_t)S16_GROUP * g * plane), 0, (int)((unsigned)S16_GROUP * plane_b), 0x00020000);
            float* dst = stage0 + st * S16_STAGE;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(dst + i * 256), 16, (int)voff[i], 0, 0, 0);
        };
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.w + (size_t)quarter * G * (9 * 256)), 0, G * 9 * 1024, 0x00020000);
        const int lane_off = lane * 16;
        f32x4 wr[2][9];
        auto wload = [&](auto PAR_, int g) {
            constexpr int PAR = decltype(PAR_)::value;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wr[PAR][tap] = ld_piece(wrs, lane_off, (g * 9 + tap) * 1024);
        };
        // B operand of (k-step s, halo row hr, tap column kx): channel 4 s + (lane >> 4) at column (lane & 15) + kx - 1
        const float* __restrict__ bbase = stage0 + (lane >> 4) * S16_CP + (lane & 15) + 3;
        auto compute = [&](auto PAR_, int st) {
            constexpr int PAR = decltype(PAR_)::value;
            const float* __restrict__ bs = bbase + st * S16_STAGE;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int hr = 0; hr < S16_HROWS; ++hr)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float bv = bs[s * 4 * S16_CP + hr * 24 + kx];
#pragma unroll
                        for (int ky = 2; ky >= 0; --ky) {          // the oldest output row first
                            const int tr = hr - ky;
                            if (tr < 0 || tr >= 4) continue;
                            acc[tr] = MFMA16(wr[PAR][3 * ky + kx][s], bv, acc[tr]);
                        }
                    }
        };
        auto grp = [&](int j) { return wave + S16_WAVES * j; };
        // group j: inputs in stage j & 1, weights in wr[j & 1].  In flight at the wait of iteration j: the inputs and weights
        // of group j + 1 (18 requests) where there is one.  LDS-DMA data is ordered for this wave's own ds_reads by its
        // counted vmcnt (MI355X_MICROARCH.md, Two waves per SIMD, item 7); a stage is overwritten only after lgkmcnt(0).
        dma(0, grp(0));
        wload(IC<0>{}, grp(0));
        if (my > 1) dma(1, grp(1));
        auto iter = [&](auto PAR_, int j) {
            constexpr int PAR = decltype(PAR_)::value;
            if (j + 1 < my) {
                wload(IC<PAR ^ 1>{}, grp(j + 1));
                asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            compute(PAR_, PAR);
            if (j + 2 < my) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of this stage has returned
                dma(PAR, grp(j + 2));
            }
        };
        int j = 0;
        for (; j + 1 < my; j += 2) {
            iter(IC<0>{}, j);
            iter(IC<1>{}, j + 1);
        }
        if (j < my) iter(IC<0>{}, j);
    }
    // ---- the 8 partial sums meet in LDS (over the stages: every wave's LDS-DMA has landed -- the last iteration waited for
    // vmcnt(0) -- and its reads have returned into the MFMAs above)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int tr = 0; tr < 4; ++tr)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(wave * 16 + tr * 4 + r) * 64 + lane] = acc[tr][r];
    __syncthreads();
    float* __restrict__ ob = p.out + (size_t)b * p.out_bs;
    const float* __restrict__ rb = p.res ? p.res + (size_t)b * p.res_bs : nullptr;
#pragma unroll
    for (int e2 = 0; e2 < 2; ++e2) {
        const int e = (int)threadIdx.x + 512 * e2;                // (row tr, register r, lane l) of the 1,024 outputs
        const int rr = e >> 6, l = e & 63;
        float v = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < S16_WAVES; ++w8) v += lds[(w8 * 16 + rr) * 64 + l];
        const int co = 16 * quarter + 4 * (l >> 4) + (rr & 3);
        const int y = y0 + (rr >> 2), x = x0 + (l & 15);
        v += p.bias[co];
        if (p.relu) v = relu0(v);
        if (y < p.H && x < p.W) {
            const size_t o = (size_t)co * plane + (size_t)y * p.W + x;
            if (rb) v += rb[o];
            ob[o] = v;
        }
    }
}

extern "C" {

size_t diinn_rdn_s16_packed_floats(void) {
    // the 130 3x3 layers of the trunk in execution order, 9 floats per (output, input) pair: SFENet2, 16 x 8 dense convs, GFF.1
    size_t n = (size_t)2 * 64 * 64 * 9;
    for (int c = 0; c < 8; ++c) n += (size_t)16 * 64 * (64 + 64 * c) * 9;
    return n;
}

int diinn_conv_s16_applies(int B, int H, int W) {
    // the small-map kernel's maps: whole 16-byte pieces per row, and below the Winograd kernels' threshold
    if (B <= 0 || H <= 0 || W <= 0 || (W & 3)) return 0;
    return (long long)B * H * W < knob(diinn_knobs().enc_wino_min) && knob(diinn_knobs().enc_no_s16) == 0;
}

int diinn_conv_s16(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                   const float* packed_s16_dev, const float* bias_dev, const float* res_dev, long long res_batch_stride,
                   float* out_dev, long long out_batch_stride, int relu, int B, int H, int W) {
    if (!in_dev || !packed_s16_dev || !bias_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin <= 0 || Cin % S16_GROUP || (W & 3)) return DIINN_ERR_UNSUPPORTED;
    if ((((size_t)in_dev) & 15) || (((size_t)packed_s16_dev) & 15) || (in_batch_stride & 3)) return DIINN_ERR_INVALID_ARG;
    const long long units = 4LL * B * ((W + 15) / 16) * ((H + 3) / 4);
    if (units > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    if ((long long)H * W * 4 * S16_GROUP > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;   // a group's planes are addressed with 32-bit byte offsets
    if ((long long)Cin / S16_GROUP * 9 * 1024 > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    ConvS16Params p;
    p.in = in_dev; p.w = packed_s16_dev; p.bias = bias_dev; p.res = res_dev; p.out = out_dev;
    p.in_bs = in_batch_stride; p.out_bs = out_batch_stride; p.res_bs = res_batch_stride;
    p.Cin = Cin; p.B = B; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
    const unsigned grid = (unsigned)((units + 31) / 32 * 32);   // 8 XCD runs of whole pixel blocks (4 quarters each)
    hipLaunchKernelGGL(conv_s16_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

}  // extern "C"
