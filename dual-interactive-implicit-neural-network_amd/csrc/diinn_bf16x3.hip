// diinn_bf16x3.hip -- the optional split-bf16 decode kernel (DIINN_COMPUTE_BF16X3)
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
#include "diinn_device.h"
#include <type_traits>

// ---------------------------------------------------------------------------------
// decode_bf16x3_kernel: the per-pixel layers 1..3 (diinn.py:128-137) on v_mfma_f32_32x32x16_bf16 at the fp32 path's
// tolerance.  Every operand of the two 256 x 256 products of a layer is carried as TWO bf16 numbers, hi = bf16(v) and
// lo = bf16(v - hi) (16 significant bits together), and a product is evaluated as
//     w . q  ~=  w_lo . q_hi  +  w_hi . q_lo  +  w_hi . q_hi          (fp32 accumulation in the MFMA)
// -- the dropped term w_lo . q_lo is 2^-16 of the result.  Any two of the three products miss the bound on the
// stress weights by 100x (tools/bf16x3_error.py), all three meet it everywhere.  Three bf16 MFMAs (32 clocks each)
// replace the eight fp32 MFMAs (64 clocks each) of the same k-range: 5.3x fewer matrix-core clocks than decode_kernel.
// Structure: decode_bf16_kernel's -- one wave owns 32 pixels, packs its activation to B fragments straight from the
// epilogue (accumulator registers 8s..8s+7 of a tile are the fragment of k-step 2m+s) -- with
//   * weights: packed section 14 (WLX), the four pieces k_hi, q_hi, k_lo, q_lo of a k-step contiguous: one scalar
//     offset per k-step, the pieces by immediate offsets, through a register ring four k-steps deep;
//   * the next layer's hi/lo fragments parked in a wave-private LDS slab (each lane re-reads only what it wrote:
//     no barrier), as in decode_bf16x2_kernel;
//   * layer 0 as in decode_kernel: the Q0 rows in revolutions with t = fma(Q0r, ratio, bQ0) folded once per
//     workgroup, and the head rows, in an LDS table; P, seeds, biases, sine and the RGB head in fp32; the head is
//     accumulated in the last layer's epilogue on the unsplit activation.
// What bounds it: at one wave per SIMD the instruction stream of a wave is serial -- the
// four 1 KiB weight loads of a k-step and layer 0's VALU work add to the six MFMAs' 192 clocks instead of hiding
// behind them (ablations: no weight loads -0.47 ms, no layer 0 -0.27 ms of 2.06 at c2), and a second wave per SIMD
// does not fit (the activation alone is 128 registers).  Moving the weight stream into LDS (shared by the four
// waves) or layer 0 into the last layer of a persistent workgroup moved the cost, not the total: both forms were
// built, bit-identical, and measured equal (r03 history).
// ---------------------------------------------------------------------------------
#ifndef DECODE_BF16X3_PREFETCH
#define DECODE_BF16X3_PREFETCH 4                // ring depth in k-steps (4 pieces, 6 MFMAs each)
#endif

// issue order of a k-step (one wave per SIMD: whatever sits between two MFMAs delays the second): a weight load
// behind each of the first four MFMAs, the epilogue's VALU spread over all six; the region ends at the k-step
// (a scheduling region over the whole unrolled layer does not finish compiling).  A/B on one box: -2.7 %.
#define X3_KSTEP_ORDER()                                                              \
    do {                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
            if (i_ < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                        \
        }                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)

__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);               // exact difference: v and hi share sign and exponent range
}

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_bf16x3_kernel(const DecodeParams p) {
    __shared__ __attribute__((aligned(16))) bf16x8 park[4][2][16][64];     // [wave][hi, lo][fragment][lane] = 128 KiB
    __shared__ __attribute__((aligned(16))) float tab[6 * HID + 4];        // Q0h, Q0w (revolutions), t, L[3], bL | validity
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float* __restrict__ Wt = p.Wt;
    {   // the tables every pixel shares (as decode_kernel): t = fma(Q0r, ratio, bQ0) once per workgroup
        const float* __restrict__ Q0s = Wt + OFF_Q0R + 4 * lane;
        if (wave == 0) {
            *(f32x4*)(tab + 0 * HID + 4 * lane) = *(const f32x4*)(Q0s + 0 * HID);
            *(f32x4*)(tab + 1 * HID + 4 * lane) = *(const f32x4*)(Q0s + 1 * HID);
        } else if (wave == 1) {
            const f32x4 wr = *(const f32x4*)(Q0s + 2 * HID), bq = *(const f32x4*)(Q0s + 3 * HID);
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(wr[e], p.ratio, bq[e]);
            *(f32x4*)(tab + 2 * HID + 4 * lane) = t;
        } else if (wave == 2) {
            *(f32x4*)(tab + 3 * HID + 4 * lane) = *(const f32x4*)(Wt + OFF_L + 0 * HID + 4 * lane);
            *(f32x4*)(tab + 4 * HID + 4 * lane) = *(const f32x4*)(Wt + OFF_L + 1 * HID + 4 * lane);
        } else {
            *(f32x4*)(tab + 5 * HID + 4 * lane) = *(const f32x4*)(Wt + OFF_L + 2 * HID + 4 * lane);
            if (lane == 0) {                                     // bL + the image's validity word (an image without the
                const f32x4 bl = *(const f32x4*)(Wt + OFF_BL);   // derived sections decodes to NaN)
                *(f32x4*)(tab + 6 * HID) = or_bits(bl, derived_nan_mask(Wt));
            }
        }
    }
    const int x = p.x0 + blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int y = p.y0 + blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.x1) && (y < p.y1);
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.y1 ? y : p.y1 - 1;
    int iy, ix;
    float relh, relw;
    axis_eval(p.ah, yc, iy, relh);
    axis_eval(p.aw, xc, ix, relw);
    const float* __restrict__ Pc = p.P + (((size_t)b * p.Prows + (iy - p.Prow0)) * p.W + ix) * PCH + 4 * h;

    // ---- layer 0 (fp32), split into hi/lo fragments: register r = 4g+e of tile m -> q[2m + (r>>3)][r&7]
    bf16x8 qh[16], ql[16];
    {
        const float* __restrict__ Q0 = tab + 4 * h;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 pv = *(const f32x4*)(Pc + c0);
                const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                const f32x4 tq = *(const f32x4*)(Q0 + 2 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = __builtin_fmaf(ww[e], relw, tq[e]);
                    a = __builtin_fmaf(wh[e], relh, a);
                    __bf16 vh, vl;
                    split_bf16(relu0(pv[e]) * dsin_rev<SIN_MODE>(a), vh, vl);
                    qh[2 * m + (g >> 1)][4 * (g & 1) + e] = vh;
                    ql[2 * m + (g >> 1)][4 * (g & 1) + e] = vl;
                }
            }
        }
    }

    constexpr int PF = DECODE_BF16X3_PREFETCH;
    static_assert(16 % PF == 0, "ring index must be static");
    constexpr int KS_BYTES = (int)(WLX_KSTEP * sizeof(float));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    // the four pieces of a k-step by immediate offsets from one scalar offset
    auto ld_kstep = [&](f32x4& kh, f32x4& qhi, f32x4& kl, f32x4& qlo, const int soff) {
        kh = ld_piece(wrs, lane_off, soff);
        qhi = ld_piece(wrs, lane_off + 1 * PIECE_BYTES, soff);
        kl = ld_piece(wrs, lane_off + 2 * PIECE_BYTES, soff);
        qlo = ld_piece(wrs, lane_off + 3 * PIECE_BYTES, soff);
    };
    int wp = (int)(OFF_WLX * sizeof(float));
    f32x4 rkh[PF], rkl[PF], rqh[PF], rql[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) ld_kstep(rkh[d], rqh[d], rkl[d], rql[d], wp + d * KS_BYTES);
    f32x4 sk[4], sq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sk[g] = *(const f32x4*)(Pc + HID + 8 * g);
        sq[g] = *(const f32x4*)(Wt + OFF_BQR + 4 * h + 8 * g);
    }
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    bf16x8 (*mine)[16][64] = park[wave];

    // the layer loop is fully unrolled: LAST is a compile-time constant per copy and fuses the RGB head
    // (diinn.py:138) into the epilogue, on the unsplit activation
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        const bool LAST = layer == 2;
        const int nl = layer < 2 ? layer + 1 : 2;
        const float* __restrict__ Pl = Pc + (layer + 1) * HID;
        const float* __restrict__ Bq = Wt + OFF_BQR + layer * HID + 4 * h;
        const float* __restrict__ Pn = Pc + (nl + 1) * HID;
        const float* __restrict__ Bn = Wt + OFF_BQR + nl * HID + 4 * h;
        const float* __restrict__ L = tab + 3 * HID + 4 * h;
        f32x16 pk, ps;
        bf16x8 fh, fl;
        f32x4 l0[4], l1[4], l2[4];
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
                const bf16x8 wkh = __builtin_bit_cast(bf16x8, rkh[s % PF]);
                const bf16x8 wqh = __builtin_bit_cast(bf16x8, rqh[s % PF]);
                // the two small products first, the leading one last (the order the oracle's emulation adds them in)
                ak = MFMA_BF16(__builtin_bit_cast(bf16x8, rkl[s % PF]), qh[ks], ak);
                as = MFMA_BF16(__builtin_bit_cast(bf16x8, rql[s % PF]), qh[ks], as);
                ak = MFMA_BF16(wkh, ql[ks], ak);
                as = MFMA_BF16(wqh, ql[ks], as);
                ak = MFMA_BF16(wkh, qh[ks], ak);
                as = MFMA_BF16(wqh, qh[ks], as);
                ld_kstep(rkh[s % PF], rqh[s % PF], rkl[s % PF], rql[s % PF], wp + (s + PF) * KS_BYTES);
                if (ks == 2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        sk[g] = *(const f32x4*)((m < 7 ? Pl + 32 * (m + 1) : Pn) + 8 * g);
                        sq[g] = *(const f32x4*)((m < 7 ? Bq + 32 * (m + 1) : Bn) + 8 * g);
                    }
                }
                if (m > 0) {                                      // one epilogue element of tile m-1 per k-step
                    float v = relu0(pk[ks]) * dsin_rev<SIN_MODE>(ps[ks]);
                    asm volatile("" : "+v"(v));                   // the element stays behind its k-step (see decode_bf16x2_kernel)
                    if (LAST) {
                        o0 = __builtin_fmaf(l0[ks >> 2][ks & 3], v, o0);
                        o1 = __builtin_fmaf(l1[ks >> 2][ks & 3], v, o1);
                        o2 = __builtin_fmaf(l2[ks >> 2][ks & 3], v, o2);
                    } else {
                        __bf16 vh, vl;
                        split_bf16(v, vh, vl);
                        fh[ks & 7] = vh;
                        fl[ks & 7] = vl;
                        if ((ks & 7) == 7) {
                            mine[0][2 * (m - 1) + (ks >> 3)][lane] = fh;
                            mine[1][2 * (m - 1) + (ks >> 3)][lane] = fl;
                        }
                    }
                }
                X3_KSTEP_ORDER();
            }
            pk = ak;
            ps = as;
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
        for (int r = 0; r < 16; ++r) {
            const float v = relu0(pk[r]) * dsin_rev<SIN_MODE>(ps[r]);
            if (LAST) {
                o0 = __builtin_fmaf(l0[r >> 2][r & 3], v, o0);
                o1 = __builtin_fmaf(l1[r >> 2][r & 3], v, o1);
                o2 = __builtin_fmaf(l2[r >> 2][r & 3], v, o2);
            } else {
                __bf16 vh, vl;
                split_bf16(v, vh, vl);
                fh[r & 7] = vh;
                fl[r & 7] = vl;
                if ((r & 7) == 7) {
                    mine[0][14 + (r >> 3)][lane] = fh;
                    mine[1][14 + (r >> 3)][lane] = fl;
                }
            }
        }
        if (!LAST) {                                              // the parked activation becomes the next layer's B operand
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                qh[i] = mine[0][i][lane];
                ql[i] = mine[1][i][lane];
            }
        }
        wp += (int)(WLX_LAYER * sizeof(float));
    }

    o0 += __shfl_xor(o0, 32);
    o1 += __shfl_xor(o1, 32);
    o2 += __shfl_xor(o2, 32);
    if (valid && h == 0) {
        const long long plane = p.o_ps;
        float* o = out_px(p, b, y, x);
        o[0] = o0 + tab[6 * HID + 0];
        o[plane] = o1 + tab[6 * HID + 1];
        o[2 * plane] = o2 + tab[6 * HID + 2];
    }
}

// ---------------------------------------------------------------------------------
// decode_bf16x3h_kernel: persistent workgroups, the hi weight pieces SHARED through LDS.
// decode_bf16x3_kernel pulls 16 KiB per k-step and CU through the vector L1 (every wave its own copy of every piece):
// 256 clocks of a 64 B/clk L1 against the 192 clocks of the six MFMAs -- the L1 sets the pace
// (tools/ubench/mfma_issue.hip: 6 MFMAs + 4 loads of 1 KiB per wave = 280 clocks per k-step on a full chip).  Here
//   * the hi pieces (k_hi, q_hi: two MFMAs each) cross the L1 once per workgroup: the four waves fetch a quarter of a
//     stage (4 k-steps x 2 pieces = 8 KiB) each, four stages ahead, pass it on through a three-slot LDS ring and read
//     their A fragments from there one k-step ahead; one workgroup barrier per stage;
//   * the lo pieces (one MFMA each) stay on the per-wave register ring.
//   L1: 8 + 2 KiB per k-step (160 clocks), LDS: 8 KiB read + 2 written (80 clocks of 128 B/clk): both under the MFMAs.
//   * workgroups are persistent (X3H_PGRID, one per CU) and evaluate the NEXT block's layer 0 inside the current
//     block's last layer, one element per k-step, into the LDS slab no activation occupies then; the rings run on
//     across the block boundary.
// Per pixel the operations and their order are decode_bf16x3_kernel's: bit-identical output (tests).
// ---------------------------------------------------------------------------------
struct X3Pixel {
    int x, y, b;
    bool valid;
    const float* Pc;       // P row of the pixel's LR cell (+ 4 * lane half)
    float relh, relw;
};

__device__ __forceinline__ X3Pixel x3_locate(const DecodeParams& p, const int blk, const int wave, const int j, const int h) {
    const int bx = blk % p.pg[0], t = blk / p.pg[0], by = t % p.pg[1], bz = t / p.pg[1];
    X3Pixel r;
    r.x = p.x0 + bx * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    r.y = p.y0 + by * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    r.b = bz;
    r.valid = (r.x < p.x1) && (r.y < p.y1);
    const int xc = r.x < p.Wu ? r.x : p.Wu - 1;
    const int yc = r.y < p.y1 ? r.y : p.y1 - 1;
    int iy, ix;
    axis_eval(p.ah, yc, iy, r.relh);
    axis_eval(p.aw, xc, ix, r.relw);
    r.Pc = p.P + (((size_t)bz * p.Prows + (iy - p.Prow0)) * p.W + ix) * PCH + 4 * h;
    return r;
}

constexpr int X3H_PGRID = 256;                  // persistent workgroups of a launch at most: one per CU (the launch asks the device for its count)
constexpr int X3H_STAGE = 4;                    // k-steps per ring stage
constexpr int X3H_NSTAGE = 384 / X3H_STAGE;     // stages per block: a multiple of the 3 slots and of the 2 fetch buffers
// (lgkmcnt only: the global loads in flight stay in flight across the barrier)
#define X3H_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// issue order of a k-step: the two A reads of the NEXT k-step at once behind the first MFMA -- they have the whole
// k-step to arrive, so the lgkmcnt(0) in front of a stage's barrier finds nothing outstanding
#define X3H_KSTEP_ORDER()                                                             \
    do {                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                            \
        _Pragma("unroll") for (int i_ = 1; i_ < 6; ++i_) {                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
            if (i_ < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                        \
        }                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_bf16x3h_kernel(const DecodeParams p) {
    __shared__ __attribute__((aligned(16))) bf16x8 park[4][2][16][64];     // [wave][hi, lo][fragment][lane] = 128 KiB
    __shared__ __attribute__((aligned(16))) f32x4 wring[3][2 * X3H_STAGE][64];   // [slot][k-step x (k_hi, q_hi)][lane] = 24 KiB
    __shared__ __attribute__((aligned(16))) float tab[6 * HID + 4];        // Q0h, Q0w (revolutions), t, L[3], bL | validity
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));              // opaque: per-lane addresses are rebuilt per block, not hoisted and spilled
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float* __restrict__ Wt = p.Wt;
    {   // the tables every pixel shares (as decode_kernel): t = fma(Q0r, ratio, bQ0) once per workgroup
        const float* __restrict__ Q0s = Wt + OFF_Q0R + 4 * lane;
        if (wave == 0) {
            *(f32x4*)(tab + 0 * HID + 4 * lane) = *(const f32x4*)(Q0s + 0 * HID);
            *(f32x4*)(tab + 1 * HID + 4 * lane) = *(const f32x4*)(Q0s + 1 * HID);
        } else if (wave == 1) {
            const f32x4 wr = *(const f32x4*)(Q0s + 2 * HID), bq = *(const f32x4*)(Q0s + 3 * HID);
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(wr[e], p.ratio, bq[e]);
            *(f32x4*)(tab + 2 * HID + 4 * lane) = t;
        } else if (wave == 2) {
            *(f32x4*)(tab + 3 * HID + 4 * lane) = *(const f32x4*)(Wt + OFF_L + 0 * HID + 4 * lane);
            *(f32x4*)(tab + 4 * HID + 4 * lane) = *(const f32x4*)(Wt + OFF_L + 1 * HID + 4 * lane);
        } else {
            *(f32x4*)(tab + 5 * HID + 4 * lane) = *(const f32x4*)(Wt + OFF_L + 2 * HID + 4 * lane);
            if (lane == 0) {
                const f32x4 bl = *(const f32x4*)(Wt + OFF_BL);
                *(f32x4*)(tab + 6 * HID) = or_bits(bl, derived_nan_mask(Wt));
            }
        }
    }
    const int nblk = p.pg[0] * p.pg[1] * p.pg[2];
    int blk = blockIdx.x;
    X3Pixel cur = x3_locate(p, blk, wave, j, h);

    constexpr int KS_BYTES = (int)(WLX_KSTEP * sizeof(float));
    constexpr int WP0 = (int)(OFF_WLX * sizeof(float));
    constexpr int LAYER_BYTES = (int)(WLX_LAYER * sizeof(float));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    // ---- the shared hi stream.  Stage t (0..95 per block: layer, M-tile, quarter) = 4 k-steps x (k_hi, q_hi); wave w
    // fetches the two pieces of k-step 4 (t % 32) + w:
    //   top of stage t:  barrier | pieces of stage t + 2 (fetched in stage t - 2) -> slot (t + 2) % 3 | fetch stage t + 4
    // (slot (t + 2) % 3 = (t - 1) % 3 was read in stage t - 1, which every wave has left; what stage t reads was stored
    // during stages t - 2 and t - 1, before the barrier)
    f32x4 gl[2][2], A[2][2];
    const int wave_ks = wave * KS_BYTES;
    auto gl_issue = [&](const int t, const int lo16) {           // t: stage of the block, compile-time after inlining
        int c = wave_ks;                                         // opaque: one s_add at the load instead of a table of
        asm volatile("" : "+s"(c));                              // every stage's offset in spilled SGPRs
        const int off = WP0 + (t >> 5) * LAYER_BYTES + X3H_STAGE * (t & 31) * KS_BYTES + c;
        gl[t & 1][0] = ld_piece(wrs, lo16, off);
        gl[t & 1][1] = ld_piece(wrs, lo16 + PIECE_BYTES, off);
    };
    auto gl_store = [&](const int t, const int l) {              // the pieces of stage t into their slot
        wring[t % 3][2 * wave + 0][l] = gl[t & 1][0];
        wring[t % 3][2 * wave + 1][l] = gl[t & 1][1];
    };
    auto a_read = [&](const int par, const int slot, const int kk, const int l) {
        A[par][0] = wring[slot][2 * kk + 0][l];
        A[par][1] = wring[slot][2 * kk + 1][l];
    };
    gl_issue(0, lane * 16);
    gl_issue(1, lane * 16);
    gl_store(0, lane);
    gl_store(1, lane);
    gl_issue(2, lane * 16);
    gl_issue(3, lane * 16);
    // ---- the lo pieces: per-wave register ring, as decode_bf16x3_kernel
    constexpr int PF = DECODE_BF16X3_PREFETCH;
    f32x4 rkl[PF], rql[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        rkl[d] = ld_piece(wrs, lane * 16 + 2 * PIECE_BYTES, WP0 + d * KS_BYTES);
        rql[d] = ld_piece(wrs, lane * 16 + 3 * PIECE_BYTES, WP0 + d * KS_BYTES);
    }
    __syncthreads();                                             // tables and the first two stages are in LDS
    a_read(0, 0, 0, lane);

    auto l0_value = [&](const float pv, const float wh, const float ww, const float tq, const float relh, const float relw) -> float {
        float a = __builtin_fmaf(ww, relw, tq);
        a = __builtin_fmaf(wh, relh, a);
        return relu0(pv) * dsin_rev<SIN_MODE>(a);
    };
    // ---- layer 0 of the first block, as in decode_bf16x3_kernel
    bf16x8 qh[16], ql[16];
    {
        const float* __restrict__ Q0 = tab + 4 * h;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 pv = *(const f32x4*)(cur.Pc + c0);
                const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
                const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
                const f32x4 tq = *(const f32x4*)(Q0 + 2 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 vh, vl;
                    split_bf16(l0_value(pv[e], wh[e], ww[e], tq[e], cur.relh, cur.relw), vh, vl);
                    qh[2 * m + (g >> 1)][4 * (g & 1) + e] = vh;
                    ql[2 * m + (g >> 1)][4 * (g & 1) + e] = vl;
                }
            }
        }
    }
    f32x4 sk[4], sq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sk[g] = *(const f32x4*)(cur.Pc + HID + 8 * g);
        sq[g] = *(const f32x4*)(Wt + OFF_BQR + 4 * h + 8 * g);
    }
    bf16x8 (*mine)[16][64] = park[wave];

    for (;;) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                              // (the same for the LDS and weight-ring addresses)
        const int lane_off = ln * 16;
        const int nb = blk + (int)gridDim.x;
        const bool more = nb < nblk;                              // the same for every wave of the workgroup
        X3Pixel nxt = cur;
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
        // three copies of the layer body (a generic lambda: `#pragma unroll` on a loop of this size inside the block
        // loop is declined): the layer and LAST are compile-time constants per copy
        auto do_layer = [&](auto layer_tag) {
            constexpr int layer = decltype(layer_tag)::value;
            constexpr bool LAST = layer == 2;
            constexpr int nl = layer < 2 ? layer + 1 : 0;         // layer whose first seeds the last M-tile fetches
            constexpr int wp = WP0 + layer * LAYER_BYTES;
            // table addresses are rebuilt from an opaque lane half in every layer copy: as invariants of the block loop
            // they would be hoisted in front of it and spilled
            int hb = h;
            asm volatile("" : "+v"(hb));
            const float* __restrict__ Q0 = tab + 4 * hb;
            const float* __restrict__ L = tab + 3 * HID + 4 * hb;
            const float* __restrict__ Pl = cur.Pc + (layer + 1) * HID;
            const float* __restrict__ Bq = Wt + OFF_BQR + layer * HID + 4 * hb;
            const float* __restrict__ Bn = Wt + OFF_BQR + nl * HID + 4 * hb;
            f32x16 pk, ps;
            bf16x8 fh, fl, nh, nw;
            f32x4 l0[4], l1[4], l2[4];
            f32x4 cpv, cwh, cww, ctq, npv, nwh, nww, ntq;         // LAST: layer-0 inputs of the next block, a group ahead
            if (LAST) {
                nxt = x3_locate(p, more ? nb : blk, wave, j, h);
                cpv = *(const f32x4*)(nxt.Pc);
                cwh = *(const f32x4*)(Q0 + 0 * HID);
                cww = *(const f32x4*)(Q0 + 1 * HID);
                ctq = *(const f32x4*)(Q0 + 2 * HID);
            }
            const float* __restrict__ Pn = LAST ? nxt.Pc + HID : cur.Pc + (nl + 1) * HID;
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
                if (LAST && m > 0) {                              // head rows of the tile being finished
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
                    const int tt = (128 * layer + s) / X3H_STAGE;   // stage of the block
                    if (s % X3H_STAGE == 0) {                     // top of a stage
                        X3H_BARRIER();
                        gl_store((tt + 2) % X3H_NSTAGE, ln);      // fetched two stages ago
                        gl_issue((tt + 4) % X3H_NSTAGE, lane_off);
                    }
                    // A fragments of the next k-step (the first of the next stage: stored before the barrier above)
                    if (s % X3H_STAGE < X3H_STAGE - 1) a_read((s + 1) & 1, tt % 3, s % X3H_STAGE + 1, ln);
                    else a_read((s + 1) & 1, (tt + 1) % 3, 0, ln);
                    const bf16x8 wkh = __builtin_bit_cast(bf16x8, A[s & 1][0]);
                    const bf16x8 wqh = __builtin_bit_cast(bf16x8, A[s & 1][1]);
                    ak = MFMA_BF16(__builtin_bit_cast(bf16x8, rkl[s % PF]), qh[ks], ak);
                    as = MFMA_BF16(__builtin_bit_cast(bf16x8, rql[s % PF]), qh[ks], as);
                    ak = MFMA_BF16(wkh, ql[ks], ak);
                    as = MFMA_BF16(wqh, ql[ks], as);
                    ak = MFMA_BF16(wkh, qh[ks], ak);
                    as = MFMA_BF16(wqh, qh[ks], as);
                    {   // lo refill: k-step s + PF of this layer; past the last layer's end, the first layer's (next block)
                        const int rp = (LAST && s + PF >= 128) ? WP0 + (s + PF - 128) * KS_BYTES : wp + (s + PF) * KS_BYTES;
                        rkl[s % PF] = ld_piece(wrs, lane_off + 2 * PIECE_BYTES, rp);
                        rql[s % PF] = ld_piece(wrs, lane_off + 3 * PIECE_BYTES, rp);
                    }
                    if (ks == 2) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            sk[g] = *(const f32x4*)((m < 7 ? Pl + 32 * (m + 1) : Pn) + 8 * g);
                            sq[g] = *(const f32x4*)((m < 7 ? Bq + 32 * (m + 1) : Bn) + 8 * g);
                        }
                    }
                    if (m > 0) {                                  // one epilogue element of tile m-1 per k-step
                        float v = relu0(pk[ks]) * dsin_rev<SIN_MODE>(ps[ks]);
                        asm volatile("" : "+v"(v));               // the element stays behind its k-step (see decode_bf16x2_kernel)
                        if (LAST) {
                            o0 = __builtin_fmaf(l0[ks >> 2][ks & 3], v, o0);
                            o1 = __builtin_fmaf(l1[ks >> 2][ks & 3], v, o1);
                            o2 = __builtin_fmaf(l2[ks >> 2][ks & 3], v, o2);
                        } else {
                            __bf16 vh, vl;
                            split_bf16(v, vh, vl);
                            fh[ks & 7] = vh;
                            fl[ks & 7] = vl;
                            if ((ks & 7) == 7) {
                                mine[0][2 * (m - 1) + (ks >> 3)][ln] = fh;
                                mine[1][2 * (m - 1) + (ks >> 3)][ln] = fl;
                            }
                        }
                    }
                    if (LAST) {                                   // layer 0 of the next block: element ks of tile m
                        if ((ks & 3) == 0 && s + 4 < 128) {       // inputs of the next group of four
                            const int c0 = 32 * ((s + 4) >> 4) + 8 * (((s + 4) >> 2) & 3);
                            npv = *(const f32x4*)(nxt.Pc + c0);
                            nwh = *(const f32x4*)(Q0 + 0 * HID + c0);
                            nww = *(const f32x4*)(Q0 + 1 * HID + c0);
                            ntq = *(const f32x4*)(Q0 + 2 * HID + c0);
                        }
                        const int e = ks & 3;
                        float v0 = l0_value(cpv[e], cwh[e], cww[e], ctq[e], nxt.relh, nxt.relw);
                        asm volatile("" : "+v"(v0));
                        __bf16 vh, vl;
                        split_bf16(v0, vh, vl);
                        nh[ks & 7] = vh;
                        nw[ks & 7] = vl;
                        if ((ks & 7) == 7) {
                            mine[0][2 * m + (ks >> 3)][ln] = nh;
                            mine[1][2 * m + (ks >> 3)][ln] = nw;
                        }
                        if (e == 3) {
                            cpv = npv; cwh = nwh; cww = nww; ctq = ntq;
                        }
                    }
                    X3H_KSTEP_ORDER();
                }
                pk = ak;
                ps = as;
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
            for (int r = 0; r < 16; ++r) {
                const float v = relu0(pk[r]) * dsin_rev<SIN_MODE>(ps[r]);
                if (LAST) {
                    o0 = __builtin_fmaf(l0[r >> 2][r & 3], v, o0);
                    o1 = __builtin_fmaf(l1[r >> 2][r & 3], v, o1);
                    o2 = __builtin_fmaf(l2[r >> 2][r & 3], v, o2);
                } else {
                    __bf16 vh, vl;
                    split_bf16(v, vh, vl);
                    fh[r & 7] = vh;
                    fl[r & 7] = vl;
                    if ((r & 7) == 7) {
                        mine[0][14 + (r >> 3)][ln] = fh;
                        mine[1][14 + (r >> 3)][ln] = fl;
                    }
                }
            }
            // the parked activation -- the next layer's, or after a last layer the next block's layer 0 -- becomes
            // the B operand
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                qh[i] = mine[0][i][ln];
                ql[i] = mine[1][i][ln];
            }
        };
        do_layer(std::integral_constant<int, 0>{});
        do_layer(std::integral_constant<int, 1>{});
        do_layer(std::integral_constant<int, 2>{});

        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        if (cur.valid && h == 0) {
            const long long plane = p.o_ps;
            float* o = out_px(p, cur.b, cur.y, cur.x);
            o[0] = o0 + tab[6 * HID + 0];
            o[plane] = o1 + tab[6 * HID + 1];
            o[2 * plane] = o2 + tab[6 * HID + 2];
        }
        if (!more) break;
        cur = nxt;
        blk = nb;
    }
}

int launch_decode_bf16x3(void* stream, const DecodeParams& p, int gx, int gy, int gz, int sin_mode) {
    const dim3 grid(gx, gy, gz);
    const long long nblk = (long long)gx * gy * gz;
    if (nblk > 0x7fffffffLL) return DIINN_ERR_TOO_LARGE;
    // persistent workgroups with the shared hi stream from two blocks per CU up (below that there is no next block
    // whose layer 0 could be overlapped); DIINN_X3_KERNEL = 1 / 2 forces the one-block / the persistent form.
    // Bit-identical results.
    // (the two forms are bit-identical, so the choice may depend on the launch -- band or tile -- at hand)
    const int force = (int)knob(diinn_knobs().x3_kernel);
    const int pgrid = device_cus() < X3H_PGRID ? device_cus() : X3H_PGRID;   // one persistent workgroup per compute unit
    if (force ? force == 2 : nblk >= 2 * pgrid) {
        DecodeParams pc = p;
        pc.pg[0] = gx; pc.pg[1] = gy; pc.pg[2] = gz;
        const dim3 gridp((unsigned)(nblk < pgrid ? nblk : pgrid));
        if (sin_mode == DIINN_SIN_HW)
            hipLaunchKernelGGL(decode_bf16x3h_kernel<DIINN_SIN_HW>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
        else if (sin_mode == DIINN_SIN_HW_REDUCED)
            hipLaunchKernelGGL(decode_bf16x3h_kernel<DIINN_SIN_HW_REDUCED>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
        else
            hipLaunchKernelGGL(decode_bf16x3h_kernel<DIINN_SIN_ACCURATE>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
        return hip_status(hipGetLastError());
    }
    if (sin_mode == DIINN_SIN_HW)
        hipLaunchKernelGGL(decode_bf16x3_kernel<DIINN_SIN_HW>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (sin_mode == DIINN_SIN_HW_REDUCED)
        hipLaunchKernelGGL(decode_bf16x3_kernel<DIINN_SIN_HW_REDUCED>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(decode_bf16x3_kernel<DIINN_SIN_ACCURATE>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}
