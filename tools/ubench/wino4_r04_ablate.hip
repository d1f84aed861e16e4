// diinn_winograd4.hip -- the 3x3 convolutions of the RDN trunk as Winograd F(4x4, 3x3) on the fp32 MFMA (gfx950).
// (part of libdiinn_hip.so; shared definitions in diinn_device.h)
//
// Reference: src/models/components/rdn.py:9-35,90-105 (130 of the trunk's 147 convolutions: 3x3, stride 1, zero
// padding 1, 64 outputs, 64..512 inputs).  F(4x4, 3x3) computes a 4x4 output block from a 6x6 input patch with 36
// multiplies per (input, output) channel pair instead of 144:
//     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A            (Lavin & Gray's matrices, interpolation points 0, +-1, +-2, inf)
// i.e. 36 independent GEMMs (one per position of the transformed 6x6 tile) with 2.25 multiplies per output against
// 4 in F(2x2, 3x3) (diinn_winograd.hip): 1.78x fewer MFMAs.  fp32 throughout; G g G^T is computed in float64 on the host
// and rounded once.  The price is accuracy: B^T holds 4 and 5, A^T up to 8, so the sums cancel more -- measured over the
// whole trunk (tools/enc_wino43_error.py, against float64): max error 2.9e-6 of max|feat| 1.6 against 3e-7 for
// F(2x2, 3x3) and 3.6e-7 for the direct sum; the trunk's parity bound is 2e-5 x max|ref| (tests/test_encoder_trunk.py).
//
// Work split (every step measured: profiles/r04_wino4_ablation.txt): a work item is a block of 32
// consecutive Winograd tiles of the row-major tile grid (128 x 4 output pixels) = one 32-wide MFMA N-tile, and ONE half of the 64 outputs (a 256 x 256 map is 128
// blocks x 2 halves = one workgroup per CU).  A workgroup has 16 waves.  Twelve MFMA waves own positions 3 w .. 3 w + 2 of the
// 36: three accumulators each, weights 1 KiB per position and chunk of 8 input channels straight from L2 into a register
// ring of two, B operands from LDS.  Four transform waves compute B^T d B ONCE per (tile, channel) -- one thread per patch,
// ~110 VALU instructions -- into a ring of three LDS buffers two chunks ahead, so one barrier per chunk orders everything.
// Their input arrives by LDS-DMA (buffer_load ... lds, range-checked: out-of-range lanes deposit zero = the padding), six
// 1-KiB rows + one edge load per wave and chunk, requested two chunks ahead; no patch lives in registers.
// The transform waves are the waves with (wave & 3) == 3: all on SIMD 3, the MFMA waves on SIMDs 0..2 -- the fp32 MFMA
// and the VALU are one pipe, and a transform beside three MFMA waves took 2.7x its time and set the iteration.
// (W4_ABL_* / W4_STAMPS / W4_MFMA_FETCH are timing-ablation and diagnosis hooks for tools/ubench/wino4_bench.hip: never in
// the library, W4_ABL_* give wrong results.)
// Epilogue: the 36 positions meet through LDS (144 KiB, over ring and raw slots), one thread per (tile, output channel):
// A^T (.) A, bias, ReLU / residual, 16-byte stores.
#include "../../dual-interactive-implicit-neural-network_amd/csrc/diinn_device.h"

constexpr int W4_TX = 32;                            // Winograd tiles per block: 32 = one MFMA N-tile; consecutive tiles of the row-major tile grid (128 x 4 output pixels; a block may wrap into the next tile row)
constexpr int W4_THREADS = 1024;                      // 12 MFMA waves + 4 transform waves
#ifdef W4_ABL_MIXED
constexpr int W4_MFMA_WAVES = 12;
#endif
constexpr int W4_VBUF = 36 * 256;                    // floats of one chunk's transformed data: [pos 36][e 4][h 2][tile 32]
constexpr int W4_RAW0 = 3 * W4_VBUF;                     // raw input slots behind the ring: [slot 2][transform wave 4][6 rows x 256 + 64 edge values]
constexpr int W4_RAW_WAVE = 6 * 256 + 64;
constexpr int W4_LDS_FLOATS = W4_RAW0 + 2 * 4 * W4_RAW_WAVE;   // 161,792 bytes; the epilogue exchange (36 * 1024 floats) lies over ring and raw slots
static_assert(W4_LDS_FLOATS >= 36 * 1024 && W4_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");             // epilogue exchange [pos 36][accumulator register 16][lane 64]; the ring uses 3 * W4_VBUF of it
constexpr int W4_PIECE_BYTES = 1024;                 // one A piece: 64 lanes x 4 k-steps
static_assert(3 * W4_VBUF <= W4_LDS_FLOATS, "ring inside the exchange buffer");

#pragma clang diagnostic ignored "-Winline-asm"          // the LDS-DMA requests below set M0 (a reserved register) in inline asm
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
// a raw buffer descriptor as four plain SGPRs (for inline asm operands): base, stride 0, bytes, the flags diinn_device.h uses
__device__ __forceinline__ i32x4 w4_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

struct ConvWino4Params {
    const float* in;         // input channel planes: in + b*in_bs + c*H*W
    const float* wu;         // packed transformed weight: [wave 12][half 2][chunk Cin/8][q 3][lane 64][4]
    const float* bias;       // [64]
    const float* res;        // optional residual [B,64,H,W] (batch stride res_bs), added after bias (no ReLU with it)
    float* out;              // out + b*out_bs + co*H*W
    long long in_bs, out_bs, res_bs;
    int Cin, B, H, W, relu;
#ifdef W4_STAMPS
    unsigned long long* stamps;   // tools/ubench/wino4_bench.hip -DW4_STAMPS: s_memtime of workgroup 0's waves 0 and 12, [wave 2][iteration 80][4]
#endif
};
#ifdef W4_STAMPS
#define W4_STAMP(role, it, i)                                                                        \
    do {                                                                                             \
        if (blockIdx.x == 0 && lane == 0 && (it) < 80) {                                             \
            unsigned long long t_;                                                                   \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            p.stamps[((role) * 80 + (it)) * 4 + (i)] = t_;                                           \
        }                                                                                            \
    } while (0)
#else
#define W4_STAMP(role, it, i) do {} while (0)
#endif

// one dimension of B^T (6 -> 6) and of A^T (6 -> 4); T = float or f32x2 (two columns at a time: v_pk_* instructions)
template <typename T>
__device__ __forceinline__ void w4_bt(const T d0, const T d1, const T d2, const T d3, const T d4, const T d5,
                                      T& r0, T& r1, T& r2, T& r3, T& r4, T& r5) {
    const T a = __builtin_elementwise_fma(T(-4.0f), d2, d4), b = __builtin_elementwise_fma(T(-4.0f), d1, d3);
    const T c = d4 - d2, e = d3 - d1;
    r0 = __builtin_elementwise_fma(T(4.0f), d0, __builtin_elementwise_fma(T(-5.0f), d2, d4));
    r1 = a + b;
    r2 = a - b;
    r3 = __builtin_elementwise_fma(T(2.0f), e, c);
    r4 = __builtin_elementwise_fma(T(-2.0f), e, c);
    r5 = __builtin_elementwise_fma(T(4.0f), d1, __builtin_elementwise_fma(T(-5.0f), d3, d5));
}
template <typename T>
__device__ __forceinline__ void w4_at(const T m0, const T m1, const T m2, const T m3, const T m4, const T m5,
                                      T& y0, T& y1, T& y2, T& y3) {
    const T s = m1 + m2, d = m1 - m2, u = m3 + m4, v = m3 - m4;
    y0 = (m0 + s) + u;
    y1 = __builtin_elementwise_fma(T(2.0f), v, d);
    y2 = __builtin_elementwise_fma(T(4.0f), u, s);
    y3 = __builtin_elementwise_fma(T(8.0f), v, d) + m5;
}

__device__ __forceinline__ void conv_wino4_body(const ConvWino4Params& p, float* __restrict__ lds, int b, int blk, int hh0) {
    // the block's tiles: blk * 32 .. + 31 of the image's row-major tile grid (a run that reaches the end of a tile row goes
    // on in the next one: no work item is left part empty by the map's width); tiles past the last one load zeros and store nothing
    const int tiles_x = (p.W + 3) / 4, tiles_n = tiles_x * ((p.H + 3) / 4);
    auto tile_xy = [&](int tile, int& tx, int& ty) {
        ty = tile / tiles_x;
        tx = tile - ty * tiles_x;
        return tile < tiles_n;
    };
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..15: the role follows from wave & 3 (below)
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_b = (unsigned)(plane * sizeof(float));
    const int n = p.Cin / 8;                                     // chunks of 8 input channels
    constexpr unsigned OUTSIDE = 0x80000000u;
    const bool ragged = (p.W & 3) != 0;                          // uniform: only then can a 16-byte access cross a row's end

    // Roles by SIMD (a workgroup's wave i runs on SIMD i % 4): the four transform waves share SIMD 3, the twelve MFMA waves
    // SIMDs 0..2.  The fp32 MFMA and the VALU are one pipe: beside three MFMA waves a transform took 3,500 cycles (1,670
    // alone) and set the iteration time (W4_ABL_MIXED keeps that layout: one transform wave per SIMD).
#ifdef W4_ABL_MIXED
    const bool producer = wave >= W4_MFMA_WAVES;
    const int wt_ = wave - W4_MFMA_WAVES, mw = wave;
#else
    const bool producer = (wave & 3) == 3;
    const int wt_ = wave >> 2, mw = (wave >> 2) * 3 + (wave & 3);
#endif
    if (threadIdx.x == 0) W4_STAMP(0, 76, 0);                     // body start
    if (producer) {
        // ---- transform waves: one (tile, channel of the chunk) patch per thread: wave wt takes the chunk's channels
        // 2 wt and 2 wt + 1 (= k-step wt of the MFMAs), a lane one of the block's 32 tiles.  Its input comes through LDS
        // (raw slots, filled by LDS-DMA): per chunk and wave six 1-KiB rows [row k][channel 2][128 columns] + the 24
        // values left and right of the block (6 rows x 2 channels x 2 sides); no patch is held in registers.
        const int wt = wt_;
#ifndef W4_ABL_NOPRIO
        __builtin_amdgcn_s_setprio(3);                           // the chunk's critical path: ahead of the MFMA waves' issue
#endif
        const int th = lane >> 5, tm = lane & 31;
        int ptx, pty;
        const bool pin = tile_xy(blk * W4_TX + tm, ptx, pty);
        (void)pin;
        const bool first = tm == 0, last = tm == W4_TX - 1;
        const bool row_start = ptx == 0, row_end = 4 * ptx + 4 >= p.W;   // the map's border: the outer column is padding (a wrapped block's neighbour lane holds another row's tile)
        const bool ok1 = 4 * ptx + 1 < p.W, ok2 = 4 * ptx + 2 < p.W, ok3 = 4 * ptx + 3 < p.W;
        float* __restrict__ raw = lds + W4_RAW0 + wt * W4_RAW_WAVE;           // + slot * 4 * W4_RAW_WAVE
#ifndef W4_MFMA_FETCH
        // The wave requests exactly the input it transforms, by LDS-DMA (buffer_load ... lds: no registers; range-checked
        // -- an out-of-range lane deposits zero, which is the zero padding), two chunks ahead: six 16-byte loads per chunk
        // (one per patch row: 2 channels x 512 contiguous bytes) and one 4-byte load whose lanes 0..23 fetch the columns
        // left and right of the block for 6 rows x 2 channels.  On their own SIMD the transform waves idle more than half
        // of an iteration; the requests cost the MFMA waves nothing there.
        const float* __restrict__ in_b = p.in + (size_t)b * p.in_bs;
        unsigned voff[6], voffe;
        {
            const int ek = lane % 6, ew = lane / 6;              // the edge load: lane < 24 -> row ek, (channel, side) ew
            int etx, ety;                                        // the block's first (left side) / last (right side) tile
            const bool ein = tile_xy(blk * W4_TX + ((ew & 1) ? W4_TX - 1 : 0), etx, ety);
            const int ex = (ew & 1) ? 4 * etx + 4 : 4 * etx - 1;
            const int ey = 4 * ety - 1 + ek;
            voffe = (lane < 24 && ein && ey >= 0 && ey < p.H && ex >= 0 && ex < p.W)
                        ? (unsigned)(2 * wt + (ew >> 1)) * plane_b + (unsigned)(ey * p.W + ex) * 4u : OUTSIDE;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int y = 4 * pty - 1 + k;
            voff[k] = (pin && y >= 0 && y < p.H) ? (unsigned)(2 * wt + th) * plane_b + (unsigned)(y * p.W + 4 * ptx) * 4u : OUTSIDE;
        }
        auto fetch = [&](int slot, int c) {
#ifndef W4_ABL_NOPATCH
            const __amdgpu_buffer_rsrc_t irs =
                __builtin_amdgcn_make_buffer_rsrc((void*)(in_b + (size_t)8 * c * plane), 0, (int)(8u * plane_b), 0x00020000);
            float* dst = raw + slot * 4 * W4_RAW_WAVE;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(dst + k * 256), 16, (int)voff[k], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(dst + 1536), 4, (int)voffe, 0, 0, 0);
#endif
        };
        auto chunk_of = [&](int k) { return k < n ? k : n - 1; };
#endif
        // the outer columns: the neighbouring tiles' values in the row, for the block's first / last tile the edge values
        const int la = first ? 1536 + 12 * th : th * 128 + tm * 4 - 1, lstep = first ? 1 : 256;
        const int ra = last ? 1536 + 12 * th + 6 : th * 128 + tm * 4 + 4, rstep = last ? 1 : 256;
        auto transform = [&](int slot, float* __restrict__ vb) {
            const float* __restrict__ src = raw + slot * 4 * W4_RAW_WAVE;
            // B^T d down the columns, two columns at a time, then (.) B along each row
            f32x2 d[6][3], t[6][3];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(src + k * 256 + th * 128 + tm * 4);
                float c1 = c4[1], c2 = c4[2], c3 = c4[3];
                if (ragged) {                                    // the 16-byte load ran past the row's end
                    c1 = ok1 ? c1 : 0.0f;
                    c2 = ok2 ? c2 : 0.0f;
                    c3 = ok3 ? c3 : 0.0f;
                }
                d[k][0] = f32x2{row_start ? 0.0f : src[la + k * lstep], c4[0]};
                d[k][1] = f32x2{c1, c2};
                d[k][2] = f32x2{c3, row_end ? 0.0f : src[ra + k * rstep]};
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                w4_bt<f32x2>(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], t[5][j]);
            float* __restrict__ dst = vb + (2 * wt + th) * 32 + tm;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float v0, v1, v2, v3, v4, v5;
                w4_bt<float>(t[i][0][0], t[i][0][1], t[i][1][0], t[i][1][1], t[i][2][0], t[i][2][1], v0, v1, v2, v3, v4, v5);
                dst[(6 * i + 0) * 256] = v0;
                dst[(6 * i + 1) * 256] = v1;
                dst[(6 * i + 2) * 256] = v2;
                dst[(6 * i + 3) * 256] = v3;
                dst[(6 * i + 4) * 256] = v4;
                dst[(6 * i + 5) * 256] = v5;
            }
        };
#ifdef W4_MFMA_FETCH
        // chunk k arrives in raw slot k & 1 (requested by the MFMA waves one iteration ahead) and is transformed into ring
        // slot k % 3; iteration c (the MFMA waves compute chunk c) transforms chunk c + 2
        __builtin_amdgcn_s_barrier();                            // P1: chunks 0 and 1 have landed
        transform(0, lds);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // P2: raw slot 0 is free again
        transform(1, lds + W4_VBUF);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // P3: chunks 0 and 1 transformed, chunk 2 landed
        int slot2 = 2;
        for (int c = 0; c < n; ++c) {
            if (wt == 0) W4_STAMP(1, c, 0);
#ifndef W4_ABL_NOTRANSFORM
            if (c + 2 < n) transform(c & 1, lds + slot2 * W4_VBUF);
#endif
            slot2 = slot2 == 2 ? 0 : slot2 + 1;
            if (wt == 0) W4_STAMP(1, c, 1);
#ifndef W4_ABL_NOBAR
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the raw slot is read, the transformed data stored
#endif
            if (wt == 0) W4_STAMP(1, c, 2);
        }
#else
        // chunk k is requested into raw slot k & 1 and transformed into ring slot k % 3; iteration c (the MFMA waves compute
        // chunk c) transforms chunk c + 2, then requests chunk c + 4 into the raw slot just read (7 requests per chunk,
        // always issued -- past the end the last chunk again -- so that "all but the newest 7" names a chunk) and retires
        // chunk c + 3 BEFORE the barrier: LDS-DMA data is ordered for a ds_read only by the issuing wave's counted vmcnt
        // followed by a barrier the reader has passed (cdna_hip_programming.md: "read a staged buffer one phase after the
        // wait that retires it") -- a read right behind the wait passes every check whenever the data happens to land first.
        fetch(0, 0);
        fetch(1, chunk_of(1));
        asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");     // P0: chunk 0 has landed
        transform(0, lds);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // raw slot 0 is read:
        fetch(0, chunk_of(2));                                   // chunk 2 lands while chunk 1 is transformed
        asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");     // P1: chunk 1 has landed
        transform(1, lds + W4_VBUF);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fetch(1, chunk_of(3));
        asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");     // P: chunks 0 and 1 are transformed, chunk 2 has landed
        int slot2 = 2;
        for (int c = 0; c < n; ++c) {
            if (wt == 0) W4_STAMP(1, c, 0);
#ifndef W4_ABL_NOTRANSFORM
            if (c + 2 < n) transform(c & 1, lds + slot2 * W4_VBUF);
#endif
            slot2 = slot2 == 2 ? 0 : slot2 + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the raw slot is read, the transformed data stored
            fetch(c & 1, chunk_of(c + 4));
            if (wt == 0) W4_STAMP(1, c, 1);
#ifndef W4_ABL_NOBAR
            asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");  // chunk c + 3 has landed (chunk c + 4 may be in flight)
#endif
            if (wt == 0) W4_STAMP(1, c, 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing may land in the raw slots any more: the exchange buffer lies over them
#endif
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    } else {
        // ---- MFMA waves: positions 3 wave .. 3 wave + 2; A = weights (32 outputs of half hh0 x 2 channels), B =
        // transformed data (2 channels x 32 tiles)
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.wu + (size_t)(mw * 2 + hh0) * n * (3 * W4_PIECE_BYTES / 4)), 0, n * 3 * W4_PIECE_BYTES, 0x00020000);
        const int lane_off = lane * 16;
#ifdef W4_MFMA_FETCH
        // The raw input of chunk c + 3 is requested here, one iteration ahead, by LDS-DMA (buffer_load ... lds: no
        // registers; range-checked -- an out-of-range lane deposits zero, which is the zero padding): wave w takes rows
        // 2 w and 2 w + 1 of the chunk's 24 (channel pair cp, patch row k) rows -- one 16-byte load per tile: 2 channels
        // x 512 contiguous bytes per request -- and waves 0..3 the edge values of channel pair w (one 4-byte load whose
        // lanes 0..23 fetch the columns left and right of the block for 6 rows x 2 channels).  Inline asm: with the
        // builtin the compiler drains every outstanding load (vmcnt(0)) at the next use of a weight register.
        const float* __restrict__ in_b = p.in + (size_t)b * p.in_bs;
        unsigned dvoff[2], dvoffe = OUTSIDE;
        unsigned dlds[2], dldse;
        {
            const int th = lane >> 5;
            int ptx, pty;
            const bool pin = tile_xy(blk * W4_TX + (lane & 31), ptx, pty);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = 2 * mw + i, cp = idx / 6, k = idx - 6 * cp;
                const int y = 4 * pty - 1 + k;
                dvoff[i] = (pin && y >= 0 && y < p.H) ? (unsigned)(2 * cp + th) * plane_b + (unsigned)(y * p.W + 4 * ptx) * 4u : OUTSIDE;
                dlds[i] = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + W4_RAW0 + cp * W4_RAW_WAVE + k * 256);
            }
            const int ek = lane % 6, ew = lane / 6;              // the edge load: lane < 24 -> row ek, (channel, side) ew
            int etx, ety;
            const bool ein = tile_xy(blk * W4_TX + ((ew & 1) ? W4_TX - 1 : 0), etx, ety);
            const int ex = (ew & 1) ? 4 * etx + 4 : 4 * etx - 1;
            const int ey = 4 * ety - 1 + ek;
            if (mw < 4 && lane < 24 && ein && ey >= 0 && ey < p.H && ex >= 0 && ex < p.W)
                dvoffe = (unsigned)(2 * mw + (ew >> 1)) * plane_b + (unsigned)(ey * p.W + ex) * 4u;
            dldse = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + W4_RAW0 + (mw & 3) * W4_RAW_WAVE + 1536);
        }
        auto fetch = [&](int slot, int c) {
#ifndef W4_ABL_NOPATCH
            const i32x4 irs = w4_rsrc(in_b + (size_t)8 * c * plane, 8u * plane_b);
            const unsigned so = (unsigned)slot * (4 * W4_RAW_WAVE * 4);
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(dlds[0] + so), "v"(dvoff[0]), "s"(irs) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(dlds[1] + so), "v"(dvoff[1]), "s"(irs) : "memory", "m0");
            if (mw < 4)
                asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(dldse + so), "v"(dvoffe), "s"(irs) : "memory", "m0");
#endif
        };
        auto chunk_of = [&](int k) { return k < n ? k : n - 1; };
#endif
        const float* __restrict__ bsrc = lds + 3 * mw * 256 + lane;      // + 64 e: [pos][e][h][tile]
        f32x16 acc[3];
#ifndef W4_EMAJOR
        f32x4 A[2][3], Bf[1][3]; // weights of chunks c, c + 1 (ring of two: a request has a whole iteration to arrive); B operands of chunk c
#else
        f32x4 A[2][3], Bf[2][3]; // weights and B operands of chunks c, c + 1 (rings of two: a request has a whole iteration to arrive)
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
#ifdef W4_MFMA_FETCH
        fetch(0, 0);
        fetch(1, chunk_of(1));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // P1: chunks 0 and 1 have landed
#else
        __builtin_amdgcn_s_barrier();                            // P0, P1: (the transform waves' first requests have landed)
        __builtin_amdgcn_s_barrier();
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            A[0][q] = ld_piece(wrs, lane_off, q * W4_PIECE_BYTES);
            A[1][q] = ld_piece(wrs, lane_off, ((n > 1 ? 3 : 0) + q) * W4_PIECE_BYTES);
        }
        __builtin_amdgcn_s_barrier();                            // P (P2): chunks 0 and 1 are transformed (raw slot 0 is free again)
#ifdef W4_MFMA_FETCH
        fetch(0, chunk_of(2));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // P3: chunks 0 and 1 transformed, chunk 2 landed
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) Bf[0][q] = f32x4{bsrc[q * 256], bsrc[q * 256 + 64], bsrc[q * 256 + 128], bsrc[q * 256 + 192]};

        if (mw == 0) W4_STAMP(0, 76, 1);                          // prologue done
        int slot1 = 1;                                           // ring slot of chunk c + 1
        // One iteration = one chunk: per position its four MFMAs, behind them the request of the position's weights of
        // chunk c + 2 into the registers just used and the read of its B operands of chunk c + 1; one barrier.  The
        // barrier waits for all but the two newest LDS operations (the last position's reads of slot c + 1, which
        // nobody writes before the NEXT barrier): no wave sits behind an LDS round trip with the matrix core idle.
        // (W4_EMAJOR: the twelve MFMAs k-step by k-step over double-buffered B registers, so that consecutive MFMAs of a
        // wave are independent: measured 2.5 % SLOWER -- 3,447 against 3,362 cycles per iteration; four waves per SIMD keep
        // the pipe busy either way.)
        auto iter = [&](auto PAR_, int c) {
            constexpr int PAR = decltype(PAR_)::value;
            const int c2 = c + 2 < n ? c + 2 : n - 1;
            if (mw == 0) W4_STAMP(0, c, 0);
#ifdef W4_MFMA_FETCH
            fetch((c + 1) & 1, chunk_of(c + 3));                 // always issued (past the end: the last chunk again): the count below relies on it
            W4_SB();
#endif
            if (mw == 0) W4_STAMP(0, c, 1);
#ifndef W4_EMAJOR
            // per position its four MFMAs, behind them the request of the position's weights of chunk c + 2 into the
            // registers just used and the read of its B operands of chunk c + 1
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q] = MFMA32(A[PAR][q][e], Bf[0][q][e], acc[q]);
                W4_SB();
                A[PAR][q] = ld_piece(wrs, lane_off, (c2 * 3 + q) * W4_PIECE_BYTES);
                {
                    const float* __restrict__ bq = bsrc + slot1 * W4_VBUF + q * 256;
                    Bf[0][q] = f32x4{bq[0], bq[64], bq[128], bq[192]};
                }
                W4_SB();
            }
#else
            // the B operands of chunk c + 1 first (their ring slot is complete since the last barrier; used next iteration),
            // then the twelve MFMAs k-step by k-step, behind a position's last MFMA the request of its weights of chunk c + 2
#ifndef W4_ABL_NOBREAD
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float* __restrict__ bq = bsrc + slot1 * W4_VBUF + q * 256;
                Bf[PAR ^ 1][q] = f32x4{bq[0], bq[64], bq[128], bq[192]};
            }
            W4_SB();
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
#ifndef W4_ABL_NOMFMA
                    acc[q] = MFMA32(A[PAR][q][e], Bf[PAR][q][e], acc[q]);
#else
                    acc[q][e] += A[PAR][q][e] * Bf[PAR][q][e];
#endif
#ifndef W4_ABL_NOW
                    if (e == 3) {
                        W4_SB();
                        A[PAR][q] = ld_piece(wrs, lane_off, (c2 * 3 + q) * W4_PIECE_BYTES);
                        W4_SB();
                    }
#endif
                }
#endif
#ifndef W4_ABL_NOBAR
            if (mw == 0) W4_STAMP(0, c, 2);
#ifdef W4_MFMA_FETCH
            // all but the three weight requests of this iteration: the raw rows requested above have landed
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(2)\n\ts_barrier" ::: "memory");
#else
#ifndef W4_EMAJOR
            asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
#else
            __builtin_amdgcn_s_barrier();                        // (this wave's LDS reads were issued at the top of the iteration)
#endif
#endif
#endif
            if (mw == 0) W4_STAMP(0, c, 3);
            slot1 = slot1 == 2 ? 0 : slot1 + 1;
        };
        int c = 0;
        for (; c + 1 < n; c += 2) {
            iter(IC<0>{}, c);
            iter(IC<1>{}, c + 1);
        }
        if (c < n) iter(IC<0>{}, c);
#ifdef W4_MFMA_FETCH
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // nothing may land in the raw slots any more: the exchange buffer lies over them
#else
        __builtin_amdgcn_s_barrier();
#endif

        if (mw == 0) W4_STAMP(0, 76, 2);                          // loop done
        // the 36 positions meet through LDS: [pos][accumulator register 16][lane]
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[((3 * mw + q) * 16 + r) * 64 + lane] = acc[q][r];
    }
    __syncthreads();
    {
        // ---- A^T (.) A, one (tile, output channel) per thread: wave = accumulator register r; lane (h, m): tile m,
        // output channel 32 hh0 + 8 (r >> 2) + 4 h + (r & 3)
        const int h = lane >> 5, m = lane & 31, r = wave;
        int tx, ty;
        const bool tin = tile_xy(blk * W4_TX + m, tx, ty);
        const float* __restrict__ src = lds + r * 64 + lane;
        float z[6][4], y[4][4];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float mm[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) mm[j] = src[(6 * i + j) * 1024];
            w4_at<float>(mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], z[i][0], z[i][1], z[i][2], z[i][3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w4_at<float>(z[0][j], z[1][j], z[2][j], z[3][j], z[4][j], z[5][j], y[0][j], y[1][j], y[2][j], y[3][j]);
        const int ox = 4 * tx, oy0 = 4 * ty;
        const bool vec = !ragged && (p.out_bs & 3) == 0 && (((size_t)p.out) & 15) == 0 &&
                         (!p.res || ((p.res_bs & 3) == 0 && (((size_t)p.res) & 15) == 0));
        const int co = 32 * hh0 + 8 * (r >> 2) + 4 * h + (r & 3);
        const float bias = p.bias[co];
        float* __restrict__ op = p.out + (size_t)b * p.out_bs + (size_t)co * plane;
        const float* __restrict__ rp = p.res ? p.res + (size_t)b * p.res_bs + (size_t)co * plane : nullptr;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = oy0 + a;
            if (!tin || oy >= p.H || ox >= p.W) continue;
            f32x4 v = {y[a][0] + bias, y[a][1] + bias, y[a][2] + bias, y[a][3] + bias};
            if (p.relu) v = f32x4{relu0(v[0]), relu0(v[1]), relu0(v[2]), relu0(v[3])};
            const size_t o = (size_t)oy * p.W + ox;
            if (vec) {
                if (rp) v += *reinterpret_cast<const f32x4*>(rp + o);
                *reinterpret_cast<f32x4*>(op + o) = v;
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if (ox + x < p.W) op[o + x] = rp ? v[x] + rp[o + x] : v[x];
            }
        }
    }
    if (threadIdx.x == 0) W4_STAMP(0, 76, 3);                     // wave 0 stored its outputs
}

__global__ __launch_bounds__(W4_THREADS) void conv_wino4_kernel(const ConvWino4Params p) {
    __shared__ __attribute__((aligned(16))) float lds[W4_LDS_FLOATS];
    const int nblk = (((p.W + 3) / 4) * ((p.H + 3) / 4) + W4_TX - 1) / W4_TX;   // blocks of 32 consecutive tiles per image
    const int total = p.B * nblk * 2;
    // every XCD takes a contiguous run of work items (as conv_wino_entry): neighbours share patch rows in one L2, and
    // the two halves of a block sit next to each other
    const int wg_per_xcd = gridDim.x >> 3;
    const int per_xcd = (total + 7) >> 3;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    for (int k = idx; k < per_xcd; k += wg_per_xcd) {
        int t = xcd * per_xcd + k;
        if (t >= total) break;
        const int hh0 = t & 1;
        t >>= 1;
        const int b = __builtin_amdgcn_readfirstlane(t / nblk);
        conv_wino4_body(p, lds, b, t - b * nblk, hh0);
        __syncthreads();                                         // the exchange buffer is free again
    }
}

#ifdef W4_STAMPS
unsigned long long* g_w4_stamps = nullptr;
#endif

extern "C" {

int diinn_conv_wino4(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                     const float* packed_u_dev, const float* bias_dev, const float* res_dev, long long res_batch_stride,
                     float* out_dev, long long out_batch_stride, int relu, int B, int H, int W) {
    if (!in_dev || !packed_u_dev || !bias_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin <= 0 || Cin % 8) return DIINN_ERR_UNSUPPORTED;
    if ((((size_t)in_dev) & 3) || (((size_t)packed_u_dev) & 15) || (((size_t)bias_dev) & 3))
        return DIINN_ERR_INVALID_ARG;
    const long long blocks = (((long long)((W + 3) / 4) * ((H + 3) / 4) + W4_TX - 1) / W4_TX) * B;
    if (2 * blocks > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    if ((long long)H * W * 4 * 64 > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;     // planes are addressed with 32-bit byte offsets
    if ((long long)Cin / 8 * 3 * W4_PIECE_BYTES > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    ConvWino4Params p;
    p.in = in_dev; p.wu = packed_u_dev; p.bias = bias_dev; p.res = res_dev; p.out = out_dev;
    p.in_bs = in_batch_stride; p.out_bs = out_batch_stride; p.res_bs = res_batch_stride;
    p.Cin = Cin; p.B = B; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
#ifdef W4_STAMPS
    p.stamps = g_w4_stamps;
#endif
    hipLaunchKernelGGL(conv_wino4_kernel, dim3((unsigned)((2 * blocks + 7) / 8 * 8)), dim3(W4_THREADS), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

}  // extern "C"
