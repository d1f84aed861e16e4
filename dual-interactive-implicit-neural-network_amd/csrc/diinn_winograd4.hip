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
// (The timing-ablation builds of round 4 -- W4_ABL_*, the MFMA-wave fetch, the k-step-major order, one transform wave per
// SIMD -- live on in tools/ubench/wino4_r04_ablate.hip, the round-4 text of this file; W4_STAMPS here is the one
// diagnosis hook left: s_memtime stamps of workgroup 0 for tools/ubench/wino4_bench.hip, never in the library.)
// Epilogue: the 36 positions meet through LDS (144 KiB, over ring and raw slots), one thread per (tile, output channel):
// A^T (.) A, bias, ReLU / residual, 16-byte stores.
//
// Round 5 -- the last, partly filled round of work items is split over the INPUT CHANNELS ("stream-K"): a launch of I work
// items on a chip of N CUs runs floor(I / N) * N items whole, as before, and cuts the chunk sequence of the R = I mod N items
// left into N equal runs, one per workgroup (a run may end inside one item and go on in the next).  A workgroup that computed
// only part of an item's channels transforms its partial sums to output space (A^T (.) A is linear), stores the 32 tiles x 16
// pixels x 32 outputs) and draws a ticket from the item's counter.  Not the last ticket: it stores the partial outputs as
// a 64 KiB slab with write-through (sc1) stores and counts the slab ready.  The last ticket: it waits until the other parts'
// slabs are counted ready -- workgroups that are past their arithmetic and wait for nobody themselves, so the wait is finite
// whatever the dispatch order or residency -- and adds the parts IN PART ORDER, its own from registers (the sum does not
// depend on who came last), then bias, ReLU / residual and the stores.  Counters are zero on entry and left zero (the last
// arriver resets its item's).  A 192 x 192 map (144 items on 256 CUs) then costs 0.56 + of a round instead of a whole one.
//
// Why there is no acquire fence (round 6: stated, not implied).  The hand-off is row 1 of the guide's table "Hand-offs measured
// with sc1 loads in place of the acquire" (MI355X_MICROARCH.md, Workgroup dispatch, XCD placement & inter-workgroup
// visibility, "Valid forms", Consumer bullet conditions (1)-(4)), matched in EVERY cell:
//   who signals     ONE lane (thread 0) of each storing workgroup, for ALL that workgroup's stores, by an agent-scope atomic
//                   add to the item's READY counter, issued after every storing wave's `s_waitcnt vmcnt(0)` and the workgroup
//                   barrier behind those waits (condition (3), Guideline 16 R1);
//   how learned     an sc1 load poll of that counter (`__hip_atomic_load(relaxed, agent)` = `global_load_dword sc1`) by thread 0
//                   of the last arriver;
//   between         the polling wave loads only after its poll matched; the other waves load behind the workgroup barrier that
//                   wave then joins (the __syncthreads() after the spin);
//   memory          hipMalloc'ed workspace; one workgroup per CU (1,024 threads, 158 KiB of LDS);
//   stores          16-byte buffer stores, all sc1 (aux 16), whole 1 KiB per wave instruction (condition (2));
//   loads           16-byte buffer loads to registers, all sc1 (never flat_, never through LDS-DMA) (condition (1)).
// The table is a measurement on gfx950 / ROCm 7.2, not an architectural guarantee, which is why the wait is BOUNDED and
// why giving up is LOUD: the last arriver that runs out of spins writes NaN to every output of its item instead of a sum of
// stale slabs, and sets the workspace's sticky STATUS word (1023), which no forward re-zeroes; from then on every split item
// computed with that workspace is NaN too, until the host has looked (diinn_conv_wino4_ws_status) and cleared it.  The
// library's rule elsewhere (the packed image's validity word) is the same: a loud NaN, never a plausible wrong picture.
#include "diinn_device.h"

constexpr int W4_TX = 32;                            // Winograd tiles per block: 32 = one MFMA N-tile; consecutive tiles of the row-major tile grid (128 x 4 output pixels; a block may wrap into the next tile row)
constexpr int W4_THREADS = 1024;                      // 12 MFMA waves + 4 transform waves
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
    unsigned m_tiles_x, m_nblk;   // tile / tiles_x and item / blocks-per-image as multiplications (div_m; 0: divide)
    // split of the last round over the input channels (0 items: none)
    float* sk_slabs;         // [2 sk_wgs][4][1024][4]: a split workgroup's partial outputs of its first / second item
    unsigned* sk_cnt;        // [1024] words: per item arrival tickets and ready counts, zero on entry, zero on exit
    int sk_first;            // work items [0, sk_first) run whole, [sk_first, sk_first + sk_items) split
    int sk_items;
    int sk_u;                // length of a run: pair q (workgroups 2 q, 2 q + 1: output halves 0, 1) takes units [q sk_u, (q + 1) sk_u)
                             // of a sequence that gives each of the sk_items / 2 blocks sk_e overhead units, then its Cin/8 chunks
    int sk_e;                // overhead units in front of a block's chunks: what a run pays in time for entering one more block
    int sk_wgs;              // split workgroups: blockIdx.x < sk_wgs (a multiple of 16)
    unsigned sk_spin_max;    // polls (with s_sleep 8 between them) before the last arriver gives up: 2^24 = seconds
    int sk_fault;            // test only (DIINN_ENC_WINO4_FAULT): 1 = the parts never count their slab ready
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
// the same instant on the constant 100 MHz counter (slot 79): shader clock over a phase = d(s_memtime) / d(s_memrealtime) x 100 MHz
#define W4_STAMP_RT(i)                                                                               \
    do {                                                                                             \
        if (blockIdx.x == 0 && lane == 0) {                                                          \
            unsigned long long t_;                                                                   \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            p.stamps[(0 * 80 + 79) * 4 + (i)] = t_;                                                  \
        }                                                                                            \
    } while (0)
#else
#define W4_STAMP(role, it, i) do {} while (0)
#define W4_STAMP_RT(i) do {} while (0)
#endif
constexpr int W4_SLAB_FLOATS = 4 * 1024 * 4;        // a split workgroup's partial outputs: [output row 4][thread 1024] f32x4 = 64 KiB
constexpr int W4_FLAG = 36 * 1024;                   // LDS word behind the exchange buffer: the ticket drawn by thread 0
static_assert(W4_FLAG < W4_LDS_FLOATS, "flag word inside the LDS array");
constexpr int W4_READY = 256, W4_STATUS = 1023;      // counter words: [item] tickets, [256 + item] slabs stored: zero on entry, zero on exit (bytes
static_assert(2 * W4_READY * 4 == DIINN_WINO4_COUNTER_BYTES, ""); // [0, 2048): what a forward re-zeroes); [1023] STICKY status: 1 = a wait gave up in this workspace
static_assert(DIINN_WINO4_MAX_SPLIT_WGS <= W4_READY, "one ticket and one ready word per split item");

// what a workgroup knows about the item it computes a channel range of (parts == 1: the whole item, nothing below is used)
struct W4Split { int parts, item, slab, first_pair, pair, blk_item; };   // item: counter index (2 blk_item + half); pairs: see the kernel

// 16-byte write-through store / L1-bypassing load (aux 16 = sc1) for the slabs: the store forms the guide's hand-off asks for
__device__ __forceinline__ void w4_st_sc1(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)voff, 0, 16);
}
__device__ __forceinline__ f32x4 w4_ld_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 16));
}

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

// chunks [c0, c0 + m) of work item (b, blk, hh0); sk.parts > 1: the item's other chunks are other workgroups' (see the header)
__device__ __forceinline__ void conv_wino4_body(const ConvWino4Params& p, float* __restrict__ lds, int b, int blk, int hh0,
                                                int c0, int m, const W4Split sk) {
    // the block's tiles: blk * 32 .. + 31 of the image's row-major tile grid (a run that reaches the end of a tile row goes
    // on in the next one: no work item is left part empty by the map's width); tiles past the last one load zeros and store nothing
    const int tiles_x = (p.W + 3) / 4, tiles_n = tiles_x * ((p.H + 3) / 4);
    auto tile_xy = [&](int tile, int& tx, int& ty) {
        ty = div_m(tile, tiles_x, p.m_tiles_x);
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
    // alone) and set the iteration time.
    const bool producer = (wave & 3) == 3;
    const int wt_ = wave >> 2, mw = (wave >> 2) * 3 + (wave & 3);
    if (threadIdx.x == 0) W4_STAMP(0, 76, 0);                     // body start
    if (producer) {
        // ---- transform waves: one (tile, channel of the chunk) patch per thread: wave wt takes the chunk's channels
        // 2 wt and 2 wt + 1 (= k-step wt of the MFMAs), a lane one of the block's 32 tiles.  Its input comes through LDS
        // (raw slots, filled by LDS-DMA): per chunk and wave six 1-KiB rows [row k][channel 2][128 columns] + the 24
        // values left and right of the block (6 rows x 2 channels x 2 sides); no patch is held in registers.
        const int wt = wt_;
        __builtin_amdgcn_s_setprio(3);                           // the chunk's critical path: ahead of the MFMA waves' issue
        const int th = lane >> 5, tm = lane & 31;
        int ptx, pty;
        const bool pin = tile_xy(blk * W4_TX + tm, ptx, pty);
        (void)pin;
        const bool first = tm == 0, last = tm == W4_TX - 1;
        const bool row_start = ptx == 0, row_end = 4 * ptx + 4 >= p.W;   // the map's border: the outer column is padding (a wrapped block's neighbour lane holds another row's tile)
        const bool ok1 = 4 * ptx + 1 < p.W, ok2 = 4 * ptx + 2 < p.W, ok3 = 4 * ptx + 3 < p.W;
        float* __restrict__ raw = lds + W4_RAW0 + wt * W4_RAW_WAVE;           // + slot * 4 * W4_RAW_WAVE
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
            const __amdgpu_buffer_rsrc_t irs =
                __builtin_amdgcn_make_buffer_rsrc((void*)(in_b + (size_t)8 * c * plane), 0, (int)(8u * plane_b), 0x00020000);
            float* dst = raw + slot * 4 * W4_RAW_WAVE;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(dst + k * 256), 16, (int)voff[k], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(dst + 1536), 4, (int)voffe, 0, 0, 0);
        };
        auto chunk_of = [&](int r) { return c0 + (r < m ? r : m - 1); };   // r-th chunk of this workgroup's range
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
        // (chunk numbers below are relative to c0.)
        // chunk k is requested into raw slot k & 1 and transformed into ring slot k % 3; iteration c (the MFMA waves compute
        // chunk c) transforms chunk c + 2, then requests chunk c + 4 into the raw slot just read (7 requests per chunk,
        // always issued -- past the end the last chunk again -- so that "all but the newest 7" names a chunk) and retires
        // chunk c + 3 BEFORE the barrier: LDS-DMA data is ordered for a ds_read only by the issuing wave's counted vmcnt
        // followed by a barrier the reader has passed (cdna_hip_programming.md: "read a staged buffer one phase after the
        // wait that retires it") -- a read right behind the wait passes every check whenever the data happens to land first.
        fetch(0, chunk_of(0));
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
        for (int c = 0; c < m; ++c) {
            if (wt == 0) W4_STAMP(1, c, 0);
            if (c + 2 < m) transform(c & 1, lds + slot2 * W4_VBUF);
            slot2 = slot2 == 2 ? 0 : slot2 + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the raw slot is read, the transformed data stored
            fetch(c & 1, chunk_of(c + 4));
            if (wt == 0) W4_STAMP(1, c, 1);
            asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");  // chunk c + 3 has landed (chunk c + 4 may be in flight)
            if (wt == 0) W4_STAMP(1, c, 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing may land in the raw slots any more: the exchange buffer lies over them
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    } else {
        // ---- MFMA waves: positions 3 wave .. 3 wave + 2; A = weights (32 outputs of half hh0 x 2 channels), B =
        // transformed data (2 channels x 32 tiles)
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.wu + (size_t)(mw * 2 + hh0) * n * (3 * W4_PIECE_BYTES / 4)), 0, n * 3 * W4_PIECE_BYTES, 0x00020000);
        const int lane_off = lane * 16;
        const float* __restrict__ bsrc = lds + 3 * mw * 256 + lane;      // + 64 e: [pos][e][h][tile]
        f32x16 acc[3];
        f32x4 A[2][3], Bf[1][3]; // weights of chunks c, c + 1 (ring of two: a request has a whole iteration to arrive); B operands of chunk c
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
        __builtin_amdgcn_s_barrier();                            // P0, P1: (the transform waves' first requests have landed)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            A[0][q] = ld_piece(wrs, lane_off, (c0 * 3 + q) * W4_PIECE_BYTES);
            A[1][q] = ld_piece(wrs, lane_off, ((c0 + (m > 1 ? 1 : 0)) * 3 + q) * W4_PIECE_BYTES);
        }
        __builtin_amdgcn_s_barrier();                            // P (P2): chunks 0 and 1 are transformed (raw slot 0 is free again)
#pragma unroll
        for (int q = 0; q < 3; ++q) Bf[0][q] = f32x4{bsrc[q * 256], bsrc[q * 256 + 64], bsrc[q * 256 + 128], bsrc[q * 256 + 192]};

        if (mw == 0) W4_STAMP(0, 76, 1);                          // prologue done
        if (mw == 0) W4_STAMP_RT(1);
        int slot1 = 1;                                           // ring slot of chunk c + 1
        // One iteration = one chunk: per position its four MFMAs, behind them the request of the position's weights of
        // chunk c + 2 into the registers just used and the read of its B operands of chunk c + 1; one barrier.  The
        // barrier waits for all but the two newest LDS operations (the last position's reads of slot c + 1, which
        // nobody writes before the NEXT barrier): no wave sits behind an LDS round trip with the matrix core idle.
        // (The twelve MFMAs k-step by k-step over double-buffered B registers, so that consecutive MFMAs of a wave are
        // independent, measured 2.5 % SLOWER -- 3,447 against 3,362 cycles per iteration; four waves per SIMD keep the pipe
        // busy either way.)  c counts from c0.
        auto iter = [&](auto PAR_, int c) {
            constexpr int PAR = decltype(PAR_)::value;
            const int c2 = c0 + (c + 2 < m ? c + 2 : m - 1);
            if (mw == 0) W4_STAMP(0, c, 0);
            if (mw == 0) W4_STAMP(0, c, 1);
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
            if (mw == 0) W4_STAMP(0, c, 2);
            asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
            if (mw == 0) W4_STAMP(0, c, 3);
            slot1 = slot1 == 2 ? 0 : slot1 + 1;
        };
        int c = 0;
        for (; c + 1 < m; c += 2) {
            iter(IC<0>{}, c);
            iter(IC<1>{}, c + 1);
        }
        if (c < m) iter(IC<0>{}, c);
        __builtin_amdgcn_s_barrier();

        if (mw == 0) W4_STAMP(0, 76, 2);                          // loop done
        if (mw == 0) W4_STAMP_RT(2);
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
        if (sk.parts > 1) {
            // ---- part of an item.  Thread 0 draws the item's ticket.  Not the last ticket: the partial outputs go to this
            // workgroup's slab (write-through, whole 1 KiB per wave instruction), every wave drains its stores, thread 0
            // counts the slab ready.  The last ticket: every other part has at least begun its stores -- wait until all
            // are counted ready (finite: those workgroups are past their arithmetic and wait for nobody), then sum the
            // parts IN PART ORDER, this workgroup's from its registers: whoever comes last, the same sum.
            const unsigned last_tk = (unsigned)sk.parts - 1u;
            if (threadIdx.x == 0) W4_STAMP(0, 77, 0);             // partial outputs in registers
            if (threadIdx.x == 0) {
                lds[W4_FLAG] = __builtin_bit_cast(float, __hip_atomic_fetch_add(p.sk_cnt + sk.item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                // the sticky status: a wait gave up in this workspace before and the host has not cleared it -> poison
                lds[W4_FLAG + 1] = __builtin_bit_cast(float, __hip_atomic_load(p.sk_cnt + W4_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            __syncthreads();
            const unsigned tk = __builtin_bit_cast(unsigned, *(volatile float*)(lds + W4_FLAG));
            const unsigned toff = threadIdx.x * 16u;
            if (threadIdx.x == 0) W4_STAMP(0, 77, 1);             // ticket known
            if (tk != last_tk) {
                const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(p.sk_slabs + (size_t)sk.slab * W4_SLAB_FLOATS), 0, W4_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int a = 0; a < 4; ++a) w4_st_sc1(f32x4{y[a][0], y[a][1], y[a][2], y[a][3]}, srs, a * 16384u + toff);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0 && !p.sk_fault)
                    __hip_atomic_fetch_add(p.sk_cnt + W4_READY + sk.item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (threadIdx.x == 0) W4_STAMP(0, 77, 2);         // slab stored, drained and counted
                return;
            }
            if (threadIdx.x == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(p.sk_cnt + W4_READY + sk.item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != last_tk) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > p.sk_spin_max) {               // seconds: something is broken.  LOUD: the sticky status word
                        __hip_atomic_store(p.sk_cnt + W4_STATUS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // and NaN outputs
                        lds[W4_FLAG + 1] = __builtin_bit_cast(float, 1u);
                        break;
                    }
                }
                // everybody has arrived and stored (or the item is poisoned): both counters are ready for the next launch
                __hip_atomic_store(p.sk_cnt + sk.item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.sk_cnt + W4_READY + sk.item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            const bool poisoned = __builtin_bit_cast(unsigned, *(volatile float*)(lds + W4_FLAG + 1)) != 0u;
            if (threadIdx.x == 0) W4_STAMP(0, 78, 0);             // the other parts are ready
            const int mine = sk.pair - sk.first_pair;            // this workgroup's part number
            auto part = [&](int k, f32x4 (&v)[4]) {
                if (k == mine) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) v[a] = f32x4{y[a][0], y[a][1], y[a][2], y[a][3]};
                } else {
                    const int pk = sk.first_pair + k;            // the k-th workgroup pair with chunks of this block; the block
                    const int jk = sk.blk_item - (pk * p.sk_u) / (n + p.sk_e);   // is the first (0) or second (1) one of its run
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(p.sk_slabs + (size_t)(2 * (2 * pk + hh0) + jk) * W4_SLAB_FLOATS), 0, W4_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
                    for (int a = 0; a < 4; ++a) v[a] = w4_ld_sc1(rs, a * 16384u + toff);
                }
            };
            f32x4 s[4], nxt[4];
            part(0, s);
            for (int k = 1; k < sk.parts; ++k) {                 // (the next part's loads are in flight while this one is added)
                part(k, nxt);
#pragma unroll
                for (int a = 0; a < 4; ++a) s[a] += nxt[a];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int x = 0; x < 4; ++x) y[a][x] = poisoned ? __builtin_nanf("") : s[a][x];   // never a sum of slabs nobody vouched for
            if (threadIdx.x == 0) W4_STAMP(0, 78, 1);             // parts summed
        }
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
    const int n = p.Cin / 8;
    if ((int)blockIdx.x < p.sk_wgs) {
        // ---- the split items (they come first in the grid: the whole items fill in behind as these workgroups end, so
        // the different lengths of their runs -- one or two prologues -- cost nothing at the launch's end).  The two halves
        // of a block are split alike and run side by side: workgroups 2 q and 2 q + 1 (a PAIR) take the same run of chunks
        // of the same blocks for output half 0 / 1, so the second reader of every raw input row finds it in the L2
        // (workgroups b and b + 8 share an XCD; each XCD takes a contiguous run of pairs).  Runs that followed the items one
        // after the other -- half 0's channels, then half 1's -- read every input row twice from memory: 15 % more cycles
        // per chunk (profiles/r05_wino4_split.txt).
        const int g = (int)(blockIdx.x & 7) * (p.sk_wgs >> 3) + (int)(blockIdx.x >> 3);
        const int pair = g >> 1, hh0 = g & 1;
        // A run is a stretch of a unit sequence in which every block contributes sk_e OVERHEAD units and then its chunks: a run
        // that enters a second block pays that block's prologue, epilogue and slab hand-off, so it gets that many chunks fewer
        // than a run that stays inside one block -- equal TIME per run, not equal chunks (the overhead units of a run's own
        // first block are what every run pays once).
        const int ne = n + p.sk_e;
        const int vtotal = (p.sk_items >> 1) * ne;
        int v0 = pair * p.sk_u;
        const int v1 = v0 + p.sk_u < vtotal ? v0 + p.sk_u : vtotal;
        for (int j = 0; v0 < v1;) {
            const int it = v0 / ne;
            const int rs = it * ne + p.sk_e, be = (it + 1) * ne;  // the block's chunks in the unit sequence
            const int a = v0 > rs ? v0 : rs, e = v1 < be ? v1 : be;
            if (a < e) {
                const int first = rs / p.sk_u, last = (be - 1) / p.sk_u;   // the pairs with chunks of this block
                const int t = (p.sk_first >> 1) + it;
                const int b = __builtin_amdgcn_readfirstlane(div_m(t, nblk, p.m_nblk));
                conv_wino4_body(p, lds, b, t - b * nblk, hh0, a - rs, e - a, W4Split{last - first + 1, 2 * it + hh0, 2 * g + j, first, pair, it});
                __syncthreads();                                 // the exchange buffer is free again
                ++j;
            }
            v0 = be;
        }
        return;
    }
    // ---- whole items: every XCD takes a contiguous run of them (as conv_wino_entry): neighbours share patch rows in one
    // L2, and the two halves of a block sit next to each other
    const int total = p.sk_first;
    const int bid = (int)blockIdx.x - p.sk_wgs;
    const int wg_per_xcd = ((int)gridDim.x - p.sk_wgs) >> 3;
    const int per_xcd = (total + 7) >> 3;
    const int xcd = bid & 7, idx = bid >> 3;
    for (int k = idx; k < per_xcd; k += wg_per_xcd) {
        int t = xcd * per_xcd + k;
        if (t >= total) break;
        const int hh0 = t & 1;
        t >>= 1;
        const int b = __builtin_amdgcn_readfirstlane(div_m(t, nblk, p.m_nblk));
        conv_wino4_body(p, lds, b, t - b * nblk, hh0, 0, n, W4Split{1, 0, 0, 0, 0, 0});
        __syncthreads();                                         // the exchange buffer is free again
    }
}

#ifdef W4_STAMPS
unsigned long long* g_w4_stamps = nullptr;
#endif

// ---- which items of a launch are split, and how (host).  Measured on 100 .. 384-pixel maps (profiles/r05_wino4_split.txt,
// tools/ubench/wino4_bench.hip): a last round of whole items costs 10 + 1.46 n us (n chunks: prologue and epilogue, then
// one iteration per chunk, R < N workgroups on the chip); the same items split into runs of u chunks 17.5 + 1.75 u us: one or
// two prologues and epilogues (a run crosses an item boundary more often than not; since the runs are balanced in TIME -- W4_SPLIT_E
// overhead units per block entered, see the kernel -- the longest run is 10.5 + 1.75 u' with u' its units), the slab round trip, and a chunk that
// takes longer because all N compute units are busy -- the part holds ~2.0 GHz with 256 of these workgroups resident and
// ~2.27 GHz with 144, at equal cycles per chunk.  That is why a 192 x 192 map (0.56 of a round) gains 15-25 % from the
// split, not 44 %: the chip is power-limited, and time follows the arithmetic done more than the workgroups in flight.
struct W4Plan { int first, items, u, wgs, e; };
// the compute units the F(4x4) kernel plans its rounds and its split for: the device's, clipped ONCE here to the workgroups
// the slab area holds and made even (workgroup pairs) -- w4_plan, w4_rounds and through them diinn_rdn_wino4_applies all
// describe the split the kernel runs, also on a part with more than 256 CUs
static int w4_cus() {
    int ncu = device_cus();
    if (ncu > DIINN_WINO4_MAX_SPLIT_WGS) ncu = DIINN_WINO4_MAX_SPLIT_WGS;
    return ncu & ~1;
}
constexpr int W4_SPLIT_E = 4;                                    // overhead units per block entered (prologue + epilogue + slab hand-off, in chunk times;
                                                                 // 2 / 3 / 6 / 8 measured: 192x192 trunk 6.83 / 6.76 / 6.80 / 6.77 ms against 6.65)
static inline double w4_whole_us(int n) { return 10.0 + 1.46 * n; }
static inline double w4_split_us(int u, int e) { return 17.5 - 1.75 * e + 1.75 * u; }   // u = run length in units (a run's e overhead units included)
static W4Plan w4_plan(long long total, int n, int ncu, bool have_ws) {
    W4Plan pl{(int)total, 0, 0, 0, 0};
    const long long mode = knob(diinn_knobs().enc_wino4_split);           // 0 never, 1 by the cost model, 2 whenever a round is partly filled
    if (!have_ws || mode == 0 || ncu < 8) return pl;
    const int R = (int)(total % ncu);                            // even: total is
    if (R == 0) return pl;
    int e = W4_SPLIT_E, ne = n + e;
    int u = (int)(((long long)(R / 2) * ne + ncu / 2 - 1) / (ncu / 2));
    if (u <= e) {                                                // runs too short to carry a block's overhead units: plain equal runs of chunks
        e = 0; ne = n;
        u = (int)(((long long)(R / 2) * ne + ncu / 2 - 1) / (ncu / 2));
    }
    if (u >= ne) return pl;                                      // a workgroup per item anyway
    if (mode == 1 && w4_split_us(u, e) > w4_whole_us(n) - 1.0) return pl;
    const int wgs = 2 * (int)(((long long)(R / 2) * ne + u - 1) / u);
    pl.first = (int)(total - R);
    pl.items = R;
    pl.u = u;
    pl.e = e;
    pl.wgs = (wgs + 15) / 16 * 16;
    return pl;
}

// the kernel's time for a map of `total` work items over the whole trunk, in rounds of whole items: the whole rounds + what a
// last round filled to r = R / N costs once it is split, 0.27 + 0.86 r of a round (measured on 112 .. 208-pixel maps:
// profiles/r05_enc_trunk_times.txt); diinn_rdn_wino4_applies compares it with the F(2x2) kernel's rounds
__attribute__((visibility("hidden"))) double w4_rounds(long long total, bool have_ws) {
    const int ncu = w4_cus();
    if (ncu < 2) return (double)total;
    const long long whole = total / ncu, R = total % ncu;
    if (R == 0) return (double)whole;
    const W4Plan pl = w4_plan(total, 36, ncu, have_ws);          // (would the trunk's average layer be split at all?)
    const double last = pl.items ? 0.27 + 0.86 * (double)R / ncu : 1.0;
    return (double)whole + (last < 1.0 ? last : 1.0);
}

extern "C" {

size_t diinn_conv_wino4_workspace_floats(void) {
    // [counters 1024 words][2 slabs per split workgroup]
    return (size_t)1024 + (size_t)2 * (DIINN_WINO4_MAX_SPLIT_WGS + 8) * W4_SLAB_FLOATS;
}

int diinn_conv_wino4_ws(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                        const float* packed_u_dev, const float* bias_dev, const float* res_dev, long long res_batch_stride,
                        float* out_dev, long long out_batch_stride, int relu, int B, int H, int W,
                        float* ws_dev, size_t ws_floats) {
    if (!in_dev || !packed_u_dev || !bias_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin <= 0 || Cin % 8) return DIINN_ERR_UNSUPPORTED;
    if ((((size_t)in_dev) & 3) || (((size_t)packed_u_dev) & 15) || (((size_t)bias_dev) & 3) || (((size_t)ws_dev) & 15))
        return DIINN_ERR_INVALID_ARG;
    if (ws_dev && ws_floats < diinn_conv_wino4_workspace_floats()) return DIINN_ERR_INVALID_ARG;
    const long long blocks = (((long long)((W + 3) / 4) * ((H + 3) / 4) + W4_TX - 1) / W4_TX) * B;
    if (2 * blocks > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    if ((long long)H * W * 4 * 64 > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;     // planes are addressed with 32-bit byte offsets
    if ((long long)Cin / 8 * 3 * W4_PIECE_BYTES > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    ConvWino4Params p;
    p.in = in_dev; p.wu = packed_u_dev; p.bias = bias_dev; p.res = res_dev; p.out = out_dev;
    p.in_bs = in_batch_stride; p.out_bs = out_batch_stride; p.res_bs = res_batch_stride;
    p.Cin = Cin; p.B = B; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
    {
        const long long tiles_x = (W + 3) / 4, nblk = (tiles_x * ((H + 3) / 4) + W4_TX - 1) / W4_TX;
        p.m_tiles_x = magic_m(tiles_x, nblk * W4_TX);            // tile numbers run up to a block's last (possibly past the map's)
        p.m_nblk = magic_m(nblk, blocks);
    }
    const W4Plan pl = w4_plan(2 * blocks, Cin / 8, w4_cus(), ws_dev != nullptr);
    p.sk_fault = knob(diinn_knobs().enc_wino4_fault) == 1 ? 1 : 0;
    p.sk_spin_max = p.sk_fault ? (1u << 10) : (1u << 24);
    p.sk_first = pl.first; p.sk_items = pl.items; p.sk_u = pl.u; p.sk_e = pl.e; p.sk_wgs = pl.wgs;
    p.sk_cnt = (unsigned*)ws_dev;
    p.sk_slabs = ws_dev ? ws_dev + 1024 : nullptr;
#ifdef W4_STAMPS
    p.stamps = g_w4_stamps;
#endif
    const unsigned grid = (unsigned)pl.wgs + (unsigned)((pl.first + 7) / 8 * 8);
    hipLaunchKernelGGL(conv_wino4_kernel, dim3(grid), dim3(W4_THREADS), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

// the workspace's sticky status word: 1 = a last arriver gave up waiting for another part's slab in some launch that used this
// workspace (its outputs, and those of every split item since, are NaN).  SYNCHRONISES `stream` (a 4-byte copy to the host):
// a query for after a forward, for tests and for bench.py -- not a launch function.  clear != 0: the whole counter area
// (4 KiB: tickets, ready counts, status) is zeroed behind the read.
int diinn_conv_wino4_ws_status(void* stream, float* ws_dev, int clear, int* status) {
    if (!ws_dev || !status) return DIINN_ERR_INVALID_ARG;
    unsigned word = 0;
    int st = hip_status(hipMemcpyAsync(&word, (const unsigned*)ws_dev + W4_STATUS, sizeof(word), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (st) return st;
    if (clear) {
        st = hip_status(hipMemsetAsync(ws_dev, 0, 4096, (hipStream_t)stream));
        if (st) return st;
    }
    st = hip_status(hipStreamSynchronize((hipStream_t)stream));
    if (st) return st;
    *status = word != 0u ? 1 : 0;
    return DIINN_OK;
}

int diinn_conv_wino4(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                     const float* packed_u_dev, const float* bias_dev, const float* res_dev, long long res_batch_stride,
                     float* out_dev, long long out_batch_stride, int relu, int B, int H, int W) {
    return diinn_conv_wino4_ws(stream, in_dev, in_batch_stride, Cin, packed_u_dev, bias_dev, res_dev, res_batch_stride,
                               out_dev, out_batch_stride, relu, B, H, W, nullptr, 0);
}

// how diinn_conv_wino4_ws would run a layer: info[0] work items, [1] items that run whole, [2] split workgroups, [3] chunks per split workgroup
int diinn_conv_wino4_plan(int Cin, int B, int H, int W, int with_workspace, int info[4]) {
    if (!info || Cin <= 0 || Cin % 8 || check_dims(B, H, W)) return DIINN_ERR_INVALID_ARG;
    const long long blocks = (((long long)((W + 3) / 4) * ((H + 3) / 4) + W4_TX - 1) / W4_TX) * B;
    const W4Plan pl = w4_plan(2 * blocks, Cin / 8, w4_cus(), with_workspace != 0);
    info[0] = (int)(2 * blocks); info[1] = pl.first; info[2] = pl.wgs; info[3] = pl.u;
    return DIINN_OK;
}

}  // extern "C"
