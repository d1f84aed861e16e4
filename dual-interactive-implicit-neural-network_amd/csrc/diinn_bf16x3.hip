// diinn_bf16x3.hip -- the optional split-bf16 decode kernel (DIINN_COMPUTE_BF16X3)
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
#include "diinn_device.h"
#include <type_traits>

// ---------------------------------------------------------------------------------
// decode_bf16x3_kernel: the per-pixel layers 1..3 (diinn.py:128-137) on v_mfma_f32_32x32x16_bf16 at fp32 accuracy.
// Every operand of the two 256 x 256 products of a layer is carried as TWO bf16 numbers, hi = bf16(v) and
// lo = bf16(v - hi) (16 significant bits together), and a product is evaluated as
//     w . q  ~=  w_hi . q_hi  +  w_lo . q_hi  +  w_hi . q_lo          (fp32 accumulation in the MFMA)
// -- the dropped term w_lo . q_lo is 2^-16 of the result, below the fp32 rounding of the 256-term sum itself.
// Three bf16 MFMAs (32 cycles each) replace the eight fp32 MFMAs (64 cycles each) of the same k-range: 5.3x fewer
// matrix-core cycles than decode_kernel.  Measured error against the reference form (tools/bf16x3_error.py, the
// GPU parity tests): that of the fp32 kernels at default-init weights, 3e-5 x |out| on the x3 stress weights.
// Structure: decode_bf16_kernel's -- one wave owns 32 pixels, packs its activation to B fragments straight from
// the epilogue (accumulator registers 8s..8s+7 of a tile are the fragment of k-step 2m+s) -- with
//   * weights: hi parts from packed section 7 (WLB), lo parts from section 14 (WLBL), the same piece order; four
//     1 KiB pieces per k-step through a register ring (0.67 KiB per MFMA: less L1 traffic per MFMA than the
//     one-tile bf16 kernel's 1 KiB);
//   * the next layer's hi/lo fragments parked in a wave-private LDS slab (each lane re-reads only what it wrote:
//     no barrier), as in decode_bf16x2_kernel;
//   * P, seeds, biases, sine, layer 0 and the RGB head in fp32; the head is accumulated in the last layer's
//     epilogue on the unsplit activation.
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
            if (i_ < 4) __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                        \
        }                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)

#define X3_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);               // exact difference: v and hi share sign and exponent range
}

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_bf16x3_kernel(const DecodeParams p) {
    __shared__ __attribute__((aligned(16))) bf16x8 park[4][2][16][64];     // [wave][hi, lo][fragment][lane] = 128 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int x = blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int y = p.y0 + blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.Wu) && (y < p.y1);
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.y1 ? y : p.y1 - 1;
    int iy, ix;
    float relh, relw;
    axis_eval(p.ah, yc, iy, relh);
    axis_eval(p.aw, xc, ix, relw);
    const float* __restrict__ Wt = p.Wt;
    const float* __restrict__ Pc = p.P + (((size_t)b * p.Prows + (iy - p.Prow0)) * p.W + ix) * PCH + 4 * h;

#ifdef X3_WARM
    // the seeds of layers 1..3 (3 KiB of the cell's P row) are first touched in the main loop, where a miss to HBM
    // holds up every weight piece behind it (one in-order vmcnt): fetch their 24 lines into the L2 now
    float warm[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
        asm volatile("global_load_dword %0, %1, off" : "=v"(warm[i]) : "v"((const char*)(Pc - 4 * h) + 1024 + (2 * i + h) * 128) : "memory");
#endif
    // ---- layer 0 (fp32), split into hi/lo fragments: register r = 4g+e of tile m -> q[2m + (r>>3)][r&7]
    bf16x8 qh[16], ql[16];
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
                const f32x4 wr = *(const f32x4*)(Q0 + 2 * HID + c0);
                const f32x4 bq = *(const f32x4*)(Q0 + 3 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = __builtin_fmaf(wr[e], p.ratio, bq[e]);
                    a = __builtin_fmaf(ww[e], relw, a);
                    a = __builtin_fmaf(wh[e], relh, a);
                    __bf16 vh, vl;
                    split_bf16(relu0(pv[e]) * dsin<SIN_MODE>(a), vh, vl);
                    qh[2 * m + (g >> 1)][4 * (g & 1) + e] = vh;
                    ql[2 * m + (g >> 1)][4 * (g & 1) + e] = vl;
                }
            }
        }
    }

#ifdef X3_WARM
    asm volatile("" :: "v"(warm[0]), "v"(warm[1]), "v"(warm[2]), "v"(warm[3]), "v"(warm[4]), "v"(warm[5]), "v"(warm[6]),
                 "v"(warm[7]), "v"(warm[8]), "v"(warm[9]), "v"(warm[10]), "v"(warm[11]));   // older than layer 0's loads: arrived
#endif
    constexpr int PF = DECODE_BF16X3_PREFETCH;
    static_assert(16 % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    constexpr int LO = (int)((OFF_WLBL - OFF_WLB) * sizeof(float));      // hi piece -> its lo piece
    int wp = (int)(OFF_WLB * sizeof(float));
    f32x4 rkh[PF], rkl[PF], rqh[PF], rql[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        rkh[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        rqh[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
        rkl[d] = ld_piece(wrs, lane_off, wp + LO + (2 * d + 0) * PIECE_BYTES);
        rql[d] = ld_piece(wrs, lane_off, wp + LO + (2 * d + 1) * PIECE_BYTES);
    }
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
        const float* __restrict__ L = Wt + OFF_L + 4 * h;
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
#ifdef X3_HAND
                // the k-step's issue order, fixed by hand (one wave per SIMD: what stands between two MFMAs delays the
                // second unless it fits the 32 clocks the first one runs; tools/ubench/mfma_issue.hip): a weight piece
                // and a quarter of the epilogue element behind each of the first four MFMAs, the seeds behind the last two
                const bf16x8 wkh = __builtin_bit_cast(bf16x8, rkh[s % PF]);
                const bf16x8 wqh = __builtin_bit_cast(bf16x8, rqh[s % PF]);
                const int sp = s - 1, rs_ = (sp + PF) % PF;
                const bool refill = layer > 0 || s > 0;
                const int rp = wp + 2 * (sp + PF) * PIECE_BYTES;
                float er = 0.0f, et = 0.0f, ev = 0.0f;
                __bf16 vh, vl;
                X3_SB();
                ak = MFMA_BF16(__builtin_bit_cast(bf16x8, rkl[s % PF]), qh[ks], ak);
                X3_SB();
                if (m > 0) er = relu0(pk[ks]);
                X3_SB();
                as = MFMA_BF16(__builtin_bit_cast(bf16x8, rql[s % PF]), qh[ks], as);
                X3_SB();
                if (refill) rkl[rs_] = ld_piece(wrs, lane_off, rp + LO);
                if (m > 0) et = dsin_rev<SIN_MODE>(ps[ks]);
                X3_SB();
                ak = MFMA_BF16(wkh, ql[ks], ak);
                X3_SB();
                if (refill) rql[rs_] = ld_piece(wrs, lane_off, rp + LO + PIECE_BYTES);
                if (m > 0) {
                    ev = er * et;
                    if (LAST) {
                        o0 = __builtin_fmaf(l0[ks >> 2][ks & 3], ev, o0);
                        o1 = __builtin_fmaf(l1[ks >> 2][ks & 3], ev, o1);
                        o2 = __builtin_fmaf(l2[ks >> 2][ks & 3], ev, o2);
                    } else {
                        split_bf16(ev, vh, vl);
                    }
                }
                X3_SB();
                as = MFMA_BF16(wqh, ql[ks], as);
                X3_SB();
                if (m > 0 && !LAST) {
                    fh[ks & 7] = vh;
                    fl[ks & 7] = vl;
                    if ((ks & 7) == 7) {
                        mine[0][2 * (m - 1) + (ks >> 3)][lane] = fh;
                        mine[1][2 * (m - 1) + (ks >> 3)][lane] = fl;
                    }
                }
                if (ks >= 2 && ks < 6) {                          // seeds of the next M-tile, two loads per k-step here ...
                    const int g = ks - 2;
                    sk[g] = *(const f32x4*)((m < 7 ? Pl + 32 * (m + 1) : Pn) + 8 * g);
                }
                X3_SB();
                ak = MFMA_BF16(wkh, qh[ks], ak);
                X3_SB();
                if (refill) rkh[rs_] = ld_piece(wrs, lane_off, rp);
                if (ks >= 2 && ks < 6) {                          // ... and here
                    const int g = ks - 2;
                    sq[g] = *(const f32x4*)((m < 7 ? Bq + 32 * (m + 1) : Bn) + 8 * g);
                }
                X3_SB();
                as = MFMA_BF16(wqh, qh[ks], as);
                X3_SB();
                if (refill) rqh[rs_] = ld_piece(wrs, lane_off, rp + PIECE_BYTES);
                X3_SB();
#else
                const bf16x8 wkh = __builtin_bit_cast(bf16x8, rkh[s % PF]);
                const bf16x8 wqh = __builtin_bit_cast(bf16x8, rqh[s % PF]);
                // the two small products first, the leading one last (the order the oracle's emulation adds them in)
                ak = MFMA_BF16(__builtin_bit_cast(bf16x8, rkl[s % PF]), qh[ks], ak);
                as = MFMA_BF16(__builtin_bit_cast(bf16x8, rql[s % PF]), qh[ks], as);
                ak = MFMA_BF16(wkh, ql[ks], ak);
                as = MFMA_BF16(wqh, ql[ks], as);
                ak = MFMA_BF16(wkh, qh[ks], ak);
                as = MFMA_BF16(wqh, qh[ks], as);
#ifndef ABL_X3_NOLOAD
                // refill the slot the PREVIOUS k-step used (k-step s - 1 + PF): no MFMA of this k-step reads it, so
                // the four loads can sit anywhere between this k-step's MFMAs (X3_KSTEP_ORDER)
                if (layer > 0 || s > 0) {
                    const int sp = s - 1;
                    rkh[(sp + PF) % PF] = ld_piece(wrs, lane_off, wp + (2 * (sp + PF) + 0) * PIECE_BYTES);
                    rqh[(sp + PF) % PF] = ld_piece(wrs, lane_off, wp + (2 * (sp + PF) + 1) * PIECE_BYTES);
                    rkl[(sp + PF) % PF] = ld_piece(wrs, lane_off, wp + LO + (2 * (sp + PF) + 0) * PIECE_BYTES);
                    rql[(sp + PF) % PF] = ld_piece(wrs, lane_off, wp + LO + (2 * (sp + PF) + 1) * PIECE_BYTES);
                }
#endif
#ifndef ABL_X3_NOSEED
                if (ks == 2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        sk[g] = *(const f32x4*)((m < 7 ? Pl + 32 * (m + 1) : Pn) + 8 * g);
                        sq[g] = *(const f32x4*)((m < 7 ? Bq + 32 * (m + 1) : Bn) + 8 * g);
                    }
                }
#endif
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
#endif
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
        wp += (int)(WLB_LAYER * sizeof(float));
    }

    o0 += __shfl_xor(o0, 32);
    o1 += __shfl_xor(o1, 32);
    o2 += __shfl_xor(o2, 32);
    if (valid && h == 0) {
        const size_t plane = (size_t)p.Orows * p.Wu;
        float* o = p.out + (size_t)b * 3 * plane + (size_t)(y - p.Orow0) * p.Wu + x;
        const unsigned nanm = derived_nan_mask(Wt);
        o[0] = o0 + or_bits(Wt[OFF_BL + 0], nanm);
        o[plane] = o1 + or_bits(Wt[OFF_BL + 1], nanm);
        o[2 * plane] = o2 + or_bits(Wt[OFF_BL + 2], nanm);
    }
}

// ---------------------------------------------------------------------------------
// decode_bf16x3p_kernel: the same arithmetic with PERSISTENT workgroups.  Layer 0 is pure VALU work (128 sines per
// lane, ~13 % of decode_bf16x3_kernel's time at one wave per SIMD, where nothing else can cover it); here a workgroup
// walks blocks blockIdx.x, + gridDim.x, ... and evaluates the NEXT block's layer 0 inside the current block's last
// layer -- one element per k-step in the issue slots the six MFMAs of a k-step leave free -- into the LDS slab, which
// no activation occupies during a last layer.  The weight ring runs on across the block boundary (the refills of the
// last layer's final k-steps fetch the first layer's pieces) and the first seeds of the next block are loaded in the
// last M-tile.  Per pixel the operations and their order are those of decode_bf16x3_kernel: bit-identical output.
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
    r.x = bx * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    r.y = p.y0 + by * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    r.b = bz;
    r.valid = (r.x < p.Wu) && (r.y < p.y1);
    const int xc = r.x < p.Wu ? r.x : p.Wu - 1;
    const int yc = r.y < p.y1 ? r.y : p.y1 - 1;
    int iy, ix;
    axis_eval(p.ah, yc, iy, r.relh);
    axis_eval(p.aw, xc, ix, r.relw);
    r.Pc = p.P + (((size_t)bz * p.Prows + (iy - p.Prow0)) * p.W + ix) * PCH + 4 * h;
    return r;
}

constexpr int X3_PGRID = 256;                   // persistent workgroups of a launch: one per CU (128 KiB of LDS each)

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_bf16x3p_kernel(const DecodeParams p) {
    __shared__ __attribute__((aligned(16))) bf16x8 park[4][2][16][64];     // [wave][hi, lo][fragment][ln] = 128 KiB
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));              // opaque: per-lane addresses are rebuilt per block, not hoisted and spilled
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int nblk = p.pg[0] * p.pg[1] * p.pg[2];
    int blk = blockIdx.x;
    X3Pixel cur = x3_locate(p, blk, wave, j, h);
    const float* __restrict__ Wt = p.Wt;
    auto l0_value = [&](const float pv, const float wh, const float ww, const float wr, const float bq,
                        const float relh, const float relw) -> float {
        float a = __builtin_fmaf(wr, p.ratio, bq);
        a = __builtin_fmaf(ww, relw, a);
        a = __builtin_fmaf(wh, relh, a);
        return relu0(pv) * dsin<SIN_MODE>(a);
    };

    // ---- layer 0 of the first block, as in decode_bf16x3_kernel
    bf16x8 qh[16], ql[16];
    const float* __restrict__ Q0 = Wt + OFF_Q0 + 4 * h;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 32 * m + 8 * g;
            const f32x4 pv = *(const f32x4*)(cur.Pc + c0);
            const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
            const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
            const f32x4 wr = *(const f32x4*)(Q0 + 2 * HID + c0);
            const f32x4 bq = *(const f32x4*)(Q0 + 3 * HID + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 vh, vl;
                split_bf16(l0_value(pv[e], wh[e], ww[e], wr[e], bq[e], cur.relh, cur.relw), vh, vl);
                qh[2 * m + (g >> 1)][4 * (g & 1) + e] = vh;
                ql[2 * m + (g >> 1)][4 * (g & 1) + e] = vl;
            }
        }
    }

    constexpr int PF = DECODE_BF16X3_PREFETCH;
    static_assert(16 % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    constexpr int LO = (int)((OFF_WLBL - OFF_WLB) * sizeof(float));      // hi piece -> its lo piece
    constexpr int WP0 = (int)(OFF_WLB * sizeof(float));
    f32x4 rkh[PF], rkl[PF], rqh[PF], rql[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        rkh[d] = ld_piece(wrs, lane * 16, WP0 + (2 * d + 0) * PIECE_BYTES);
        rqh[d] = ld_piece(wrs, lane * 16, WP0 + (2 * d + 1) * PIECE_BYTES);
        rkl[d] = ld_piece(wrs, lane * 16, WP0 + LO + (2 * d + 0) * PIECE_BYTES);
        rql[d] = ld_piece(wrs, lane * 16, WP0 + LO + (2 * d + 1) * PIECE_BYTES);
    }
    f32x4 sk[4], sq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sk[g] = *(const f32x4*)(cur.Pc + HID + 8 * g);
        sq[g] = *(const f32x4*)(Wt + OFF_BQR + 4 * h + 8 * g);
    }
    bf16x8 (*mine)[16][64] = park[wave];
    const unsigned nanm = derived_nan_mask(Wt);

    for (;;) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                              // (the same for the LDS and weight-ring addresses)
        const int lane_off = ln * 16;
        const int nb = blk + (int)gridDim.x;
        const bool more = nb < nblk;                              // the same for every wave of the launch's workgroup
        X3Pixel nxt = cur;
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
        // three copies of the layer body (a generic lambda: `#pragma unroll` on a loop of this size inside the block
        // loop is declined): the layer and LAST are compile-time constants per copy
        auto do_layer = [&](auto layer_tag) {
            constexpr int layer = decltype(layer_tag)::value;
            constexpr bool LAST = layer == 2;
            constexpr int nl = layer < 2 ? layer + 1 : 0;         // layer whose first seeds the last M-tile fetches
            constexpr int wp = WP0 + layer * (int)(WLB_LAYER * sizeof(float));
            // table addresses are rebuilt from an opaque lane half in every layer copy: as invariants of the block loop
            // they would be hoisted in front of it, sixty-odd 64-bit values, and spilled
            int hb = h;
            asm volatile("" : "+v"(hb));
            const float* __restrict__ Q0 = Wt + OFF_Q0 + 4 * hb;
            const float* __restrict__ Pl = cur.Pc + (layer + 1) * HID;
            const float* __restrict__ Bq = Wt + OFF_BQR + layer * HID + 4 * hb;
            const float* __restrict__ Bn = Wt + OFF_BQR + nl * HID + 4 * hb;
            const float* __restrict__ L = Wt + OFF_L + 4 * hb;
            f32x16 pk, ps;
            bf16x8 fh, fl, nh, nw;
            f32x4 l0[4], l1[4], l2[4];
            f32x4 cpv, cwh, cww, cwr, cbq, npv, nwh, nww, nwr, nbq;   // LAST: layer-0 inputs of the next block, a group ahead
            if (LAST) {
                nxt = x3_locate(p, more ? nb : blk, wave, j, h);
                cpv = *(const f32x4*)(nxt.Pc);
                cwh = *(const f32x4*)(Q0 + 0 * HID);
                cww = *(const f32x4*)(Q0 + 1 * HID);
                cwr = *(const f32x4*)(Q0 + 2 * HID);
                cbq = *(const f32x4*)(Q0 + 3 * HID);
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
                    const bf16x8 wkh = __builtin_bit_cast(bf16x8, rkh[s % PF]);
                    const bf16x8 wqh = __builtin_bit_cast(bf16x8, rqh[s % PF]);
                    ak = MFMA_BF16(__builtin_bit_cast(bf16x8, rkl[s % PF]), qh[ks], ak);
                    as = MFMA_BF16(__builtin_bit_cast(bf16x8, rql[s % PF]), qh[ks], as);
                    ak = MFMA_BF16(wkh, ql[ks], ak);
                    as = MFMA_BF16(wqh, ql[ks], as);
                    ak = MFMA_BF16(wkh, qh[ks], ak);
                    as = MFMA_BF16(wqh, qh[ks], as);
                    {   // refill: k-step s + PF of this layer; past the last layer's end, the first layer's (next block)
                        const int rp = (LAST && s + PF >= 128) ? WP0 + 2 * (s + PF - 128) * PIECE_BYTES : wp + 2 * (s + PF) * PIECE_BYTES;
                        rkh[s % PF] = ld_piece(wrs, lane_off, rp);
                        rqh[s % PF] = ld_piece(wrs, lane_off, rp + PIECE_BYTES);
                        rkl[s % PF] = ld_piece(wrs, lane_off, rp + LO);
                        rql[s % PF] = ld_piece(wrs, lane_off, rp + LO + PIECE_BYTES);
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
#ifdef ABL_X3P_WARMP
                            npv = *(const f32x4*)(cur.Pc + c0);
#else
                            npv = *(const f32x4*)(nxt.Pc + c0);
#endif
                            nwh = *(const f32x4*)(Q0 + 0 * HID + c0);
                            nww = *(const f32x4*)(Q0 + 1 * HID + c0);
                            nwr = *(const f32x4*)(Q0 + 2 * HID + c0);
                            nbq = *(const f32x4*)(Q0 + 3 * HID + c0);
                        }
                        const int e = ks & 3;
                        float v0 = l0_value(cpv[e], cwh[e], cww[e], cwr[e], cbq[e], nxt.relh, nxt.relw);
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
                            cpv = npv; cwh = nwh; cww = nww; cwr = nwr; cbq = nbq;
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
            const size_t plane = (size_t)p.Orows * p.Wu;
            float* o = p.out + (size_t)cur.b * 3 * plane + (size_t)(cur.y - p.Orow0) * p.Wu + cur.x;
            o[0] = o0 + or_bits(Wt[OFF_BL + 0], nanm);
            o[plane] = o1 + or_bits(Wt[OFF_BL + 1], nanm);
            o[2 * plane] = o2 + or_bits(Wt[OFF_BL + 2], nanm);
        }
        if (!more) break;
        cur = nxt;
        blk = nb;
    }
}

// ---------------------------------------------------------------------------------
// decode_bf16x3s_kernel: the persistent form with the weight stream SHARED by the four waves of a workgroup.
// In the kernels above every wave pulls its own copy of every weight piece through the vector L1: 16 KiB per
// k-step and CU, 256 clocks of a 64 B/clk L1 against 192 clocks of MFMA work -- the L1 sets the pace (ablations in
// no refills -0.47 ms of 2.06).  Here a piece crosses the L1 once: the waves fetch a quarter
// of a stage (4 k-steps x 4 pieces = 16 KiB) each, three stages ahead, pass it on through a three-slot LDS ring and
// read their A fragments from there one k-step ahead (ds_read_b128, conflict-free); one workgroup barrier per stage.
// L1: 4 KiB per k-step (64 clocks); LDS: 16 KiB read + 4 KiB written (160 clocks of 128 B/clk).
// LDS budget: the ring takes 48 KiB, so only the hi parts of the next activation are parked in LDS (64 KiB); the lo
// parts stay in registers.  Operations per pixel and their order are unchanged: bit-identical output.
// ---------------------------------------------------------------------------------
#ifndef X3S_STAGE_KS
#define X3S_STAGE_KS 2
#endif
constexpr int X3S_STAGE = X3S_STAGE_KS;         // k-steps per ring stage (4: the lo parts of the next activation stay in registers)
constexpr bool X3S_LO_IN_LDS = X3S_STAGE == 2;
constexpr int X3S_NSTAGE = 384 / X3S_STAGE;     // stages per block (a multiple of the three slots)
constexpr int X3S_SLOTS = 3;
// (lgkmcnt only: the stage's own global loads stay in flight across the barrier)
#define X3S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// issue order of a k-step: the four A reads of the NEXT k-step at once behind the first MFMA -- they have the whole
// k-step to arrive, so the lgkmcnt(0) in front of the next stage's barrier finds nothing outstanding
#define X3S_KSTEP_ORDER()                                                             \
    do {                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                            \
        _Pragma("unroll") for (int i_ = 1; i_ < 6; ++i_) {                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
            if (i_ < 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                        \
        }                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_bf16x3s_kernel(const DecodeParams p) {
    __shared__ __attribute__((aligned(16))) f32x4 wring[X3S_SLOTS][4 * X3S_STAGE][64];   // [slot][k-step x (k_hi, q_hi, k_lo, q_lo)][lane] = 48 KiB
    __shared__ __attribute__((aligned(16))) bf16x8 park[4][X3S_LO_IN_LDS ? 32 : 16][64];   // [wave][hi 0..15, lo 16..31][lane]: the next activation
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));              // opaque: per-lane addresses are rebuilt per block, not hoisted and spilled
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int nblk = p.pg[0] * p.pg[1] * p.pg[2];
    int blk = blockIdx.x;
    X3Pixel cur = x3_locate(p, blk, wave, j, h);
    const float* __restrict__ Wt = p.Wt;
    auto l0_value = [&](const float pv, const float wh, const float ww, const float wr, const float bq,
                        const float relh, const float relw) -> float {
        float a = __builtin_fmaf(wr, p.ratio, bq);
        a = __builtin_fmaf(ww, relw, a);
        a = __builtin_fmaf(wh, relh, a);
        return relu0(pv) * dsin<SIN_MODE>(a);
    };

    // ---- layer 0 of the first block, as in decode_bf16x3_kernel
    bf16x8 qh[16], ql[16];
    const float* __restrict__ Q0 = Wt + OFF_Q0 + 4 * h;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 32 * m + 8 * g;
            const f32x4 pv = *(const f32x4*)(cur.Pc + c0);
            const f32x4 wh = *(const f32x4*)(Q0 + 0 * HID + c0);
            const f32x4 ww = *(const f32x4*)(Q0 + 1 * HID + c0);
            const f32x4 wr = *(const f32x4*)(Q0 + 2 * HID + c0);
            const f32x4 bq = *(const f32x4*)(Q0 + 3 * HID + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 vh, vl;
                split_bf16(l0_value(pv[e], wh[e], ww[e], wr[e], bq[e], cur.relh, cur.relw), vh, vl);
                qh[2 * m + (g >> 1)][4 * (g & 1) + e] = vh;
                ql[2 * m + (g >> 1)][4 * (g & 1) + e] = vl;
            }
        }
    }

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    constexpr int LO = (int)((OFF_WLBL - OFF_WLB) * sizeof(float));      // hi piece -> its lo piece
    constexpr int WP0 = (int)(OFF_WLB * sizeof(float));
    // ---- the shared weight stream: stage t (0..95 per block: layer, M-tile, quarter) = 4 k-steps x 4 pieces.  Wave w
    // fetches the four pieces of k-step 4 (t % 32) + w three stages ahead and stores them one stage later:
    //   top of stage t:  barrier | pieces of stage t + 2 (fetched in stage t - 2) -> slot (t + 2) % 3 | fetch stage t + 4
    // (slot (t + 2) % 3 = (t - 1) % 3 was read in stage t - 1, which every wave has left; what stage t reads was stored
    // during stage t - 2 and t - 1, before the barrier)
    f32x4 gl[2][X3S_STAGE], A[2][4];
    // piece pi = 4 kk + c of a stage (c: k_hi, q_hi, k_lo, q_lo); wave w fetches pieces X3S_STAGE w .. + X3S_STAGE - 1
    int wave_piece[X3S_STAGE];                                   // byte offsets of this wave's pieces inside a stage
#pragma unroll
    for (int i = 0; i < X3S_STAGE; ++i) {
        const int pi = X3S_STAGE * wave + i;
        wave_piece[i] = (2 * (pi >> 2) + (pi & 1)) * PIECE_BYTES + ((pi >> 1) & 1) * LO;
    }
    auto gl_issue = [&](const int t, const int lo16) {           // t: stage of the block, compile-time after inlining
        constexpr int SPL = 128 / X3S_STAGE;                     // stages per layer
        const int base = WP0 + (t / SPL) * (int)(WLB_LAYER * sizeof(float)) + 2 * X3S_STAGE * (t % SPL) * PIECE_BYTES;
#pragma unroll
        for (int i = 0; i < X3S_STAGE; ++i) {
            int c = wave_piece[i];                               // opaque: base + c is one s_add at the load, not a table of
            asm volatile("" : "+s"(c));                          // every stage's offset kept in spilled SGPRs
            gl[t & 1][i] = ld_piece(wrs, lo16, base + c);
        }
    };
    auto gl_store = [&](const int t, const int l) {              // the pieces of stage t into their slot
#pragma unroll
        for (int i = 0; i < X3S_STAGE; ++i) wring[t % 3][X3S_STAGE * wave + i][l] = gl[t & 1][i];
    };
    auto a_read = [&](const int par, const int slot, const int kk, const int l) {
#pragma unroll
        for (int i = 0; i < 4; ++i) A[par][i] = wring[slot][4 * kk + i][l];
    };
    gl_issue(0, lane * 16);
    gl_issue(1, lane * 16);
    gl_store(0, lane);
    gl_store(1, lane);
    gl_issue(2, lane * 16);
    gl_issue(3, lane * 16);
    X3S_BARRIER();
    a_read(0, 0, 0, lane);
    f32x4 sk[4], sq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sk[g] = *(const f32x4*)(cur.Pc + HID + 8 * g);
        sq[g] = *(const f32x4*)(Wt + OFF_BQR + 4 * h + 8 * g);
    }
    bf16x8 (*mine)[64] = park[wave];
    const unsigned nanm = derived_nan_mask(Wt);

    for (;;) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                              // (the same for the LDS and weight-ring addresses)
        const int lane_off = ln * 16;
        const int nb = blk + (int)gridDim.x;
        const bool more = nb < nblk;                              // the same for every wave of the launch's workgroup
        X3Pixel nxt = cur;
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
        // three copies of the layer body (a generic lambda: `#pragma unroll` on a loop of this size inside the block
        // loop is declined): the layer and LAST are compile-time constants per copy
        auto do_layer = [&](auto layer_tag) {
            constexpr int layer = decltype(layer_tag)::value;
            constexpr bool LAST = layer == 2;
            constexpr int nl = layer < 2 ? layer + 1 : 0;         // layer whose first seeds the last M-tile fetches
            constexpr int wp = WP0 + layer * (int)(WLB_LAYER * sizeof(float));
            // table addresses are rebuilt from an opaque lane half in every layer copy: as invariants of the block loop
            // they would be hoisted in front of it, sixty-odd 64-bit values, and spilled
            int hb = h;
            asm volatile("" : "+v"(hb));
            const float* __restrict__ Q0 = Wt + OFF_Q0 + 4 * hb;
            const float* __restrict__ Pl = cur.Pc + (layer + 1) * HID;
            const float* __restrict__ Bq = Wt + OFF_BQR + layer * HID + 4 * hb;
            const float* __restrict__ Bn = Wt + OFF_BQR + nl * HID + 4 * hb;
            const float* __restrict__ L = Wt + OFF_L + 4 * hb;
            f32x16 pk, ps;
            bf16x8 fh, fl, nh, nw;
            bf16x8 nlo[16];                                       // lo parts of the next activation (the hi parts go to LDS)
            f32x4 hl[3], hn[3];                                   // LAST: head rows of four elements, and of the next four
            f32x4 cpv, cwh, cww, cwr, cbq, npv, nwh, nww, nwr, nbq;   // LAST: layer-0 inputs of the next block, a group ahead
            if (LAST) {
                nxt = x3_locate(p, more ? nb : blk, wave, j, h);
                cpv = *(const f32x4*)(nxt.Pc);
                cwh = *(const f32x4*)(Q0 + 0 * HID);
                cww = *(const f32x4*)(Q0 + 1 * HID);
                cwr = *(const f32x4*)(Q0 + 2 * HID);
                cbq = *(const f32x4*)(Q0 + 3 * HID);
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
                if (LAST && m == 1) {                             // head rows of the first four finished elements
                    hl[0] = *(const f32x4*)(L + 0 * HID);
                    hl[1] = *(const f32x4*)(L + 1 * HID);
                    hl[2] = *(const f32x4*)(L + 2 * HID);
                }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const int s = m * 16 + ks;
                    const int tt = (128 * layer + s) / X3S_STAGE;   // stage of the block
                    if (s % X3S_STAGE == 0) {                     // top of a stage
                        X3S_BARRIER();
                        gl_store((tt + 2) % X3S_NSTAGE, ln);          // fetched two stages ago
                        gl_issue((tt + 4) % X3S_NSTAGE, lane_off);
                    }
                    // A fragments of the next k-step (the first of the next stage: stored before the barrier above)
                    if (s % X3S_STAGE < X3S_STAGE - 1) a_read((s + 1) & 1, tt % 3, s % X3S_STAGE + 1, ln);
                    else a_read((s + 1) & 1, (tt + 1) % 3, 0, ln);
                    const bf16x8 wkh = __builtin_bit_cast(bf16x8, A[s & 1][0]);
                    const bf16x8 wqh = __builtin_bit_cast(bf16x8, A[s & 1][1]);
                    ak = MFMA_BF16(__builtin_bit_cast(bf16x8, A[s & 1][2]), qh[ks], ak);
                    as = MFMA_BF16(__builtin_bit_cast(bf16x8, A[s & 1][3]), qh[ks], as);
                    ak = MFMA_BF16(wkh, ql[ks], ak);
                    as = MFMA_BF16(wqh, ql[ks], as);
                    ak = MFMA_BF16(wkh, qh[ks], ak);
                    as = MFMA_BF16(wqh, qh[ks], as);
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
                            if ((ks & 3) == 0) {                  // rows of the next group (tile m - 1, or m: the tail)
                                const int c1 = 32 * (m - 1) + 8 * (ks >> 2) + 8 + (ks == 12 ? 32 - 32 : 0);
                                const int cn = ks == 12 ? 32 * m : c1;
                                hn[0] = *(const f32x4*)(L + 0 * HID + cn);
                                hn[1] = *(const f32x4*)(L + 1 * HID + cn);
                                hn[2] = *(const f32x4*)(L + 2 * HID + cn);
                            }
                            o0 = __builtin_fmaf(hl[0][ks & 3], v, o0);
                            o1 = __builtin_fmaf(hl[1][ks & 3], v, o1);
                            o2 = __builtin_fmaf(hl[2][ks & 3], v, o2);
                            if ((ks & 3) == 3) {
                                hl[0] = hn[0]; hl[1] = hn[1]; hl[2] = hn[2];
                            }
                        } else {
                            __bf16 vh, vl;
                            split_bf16(v, vh, vl);
                            fh[ks & 7] = vh;
                            fl[ks & 7] = vl;
                            if ((ks & 7) == 7) {
                                mine[2 * (m - 1) + (ks >> 3)][ln] = fh;
                                if (X3S_LO_IN_LDS) mine[16 + 2 * (m - 1) + (ks >> 3)][ln] = fl; else nlo[2 * (m - 1) + (ks >> 3)] = fl;
                            }
                        }
                    }
                    if (LAST) {                                   // layer 0 of the next block: element ks of tile m
                        if ((ks & 3) == 0 && s + 4 < 128) {       // inputs of the next group of four
                            const int c0 = 32 * ((s + 4) >> 4) + 8 * (((s + 4) >> 2) & 3);
#ifdef ABL_X3P_WARMP
                            npv = *(const f32x4*)(cur.Pc + c0);
#else
                            npv = *(const f32x4*)(nxt.Pc + c0);
#endif
                            nwh = *(const f32x4*)(Q0 + 0 * HID + c0);
                            nww = *(const f32x4*)(Q0 + 1 * HID + c0);
                            nwr = *(const f32x4*)(Q0 + 2 * HID + c0);
                            nbq = *(const f32x4*)(Q0 + 3 * HID + c0);
                        }
                        const int e = ks & 3;
                        float v0 = l0_value(cpv[e], cwh[e], cww[e], cwr[e], cbq[e], nxt.relh, nxt.relw);
                        asm volatile("" : "+v"(v0));
                        __bf16 vh, vl;
                        split_bf16(v0, vh, vl);
                        nh[ks & 7] = vh;
                        nw[ks & 7] = vl;
                        if ((ks & 7) == 7) {
                            mine[2 * m + (ks >> 3)][ln] = nh;
                            if (X3S_LO_IN_LDS) mine[16 + 2 * m + (ks >> 3)][ln] = nw; else nlo[2 * m + (ks >> 3)] = nw;
                        }
                        if (e == 3) {
                            cpv = npv; cwh = nwh; cww = nww; cwr = nwr; cbq = nbq;
                        }
                    }
                    X3S_KSTEP_ORDER();
                }
                pk = ak;
                ps = as;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = relu0(pk[r]) * dsin_rev<SIN_MODE>(ps[r]);
                if (LAST) {
                    if ((r & 3) == 0 && r < 12) {
                        hn[0] = *(const f32x4*)(L + 0 * HID + 32 * 7 + 8 * (r >> 2) + 8);
                        hn[1] = *(const f32x4*)(L + 1 * HID + 32 * 7 + 8 * (r >> 2) + 8);
                        hn[2] = *(const f32x4*)(L + 2 * HID + 32 * 7 + 8 * (r >> 2) + 8);
                    }
                    o0 = __builtin_fmaf(hl[0][r & 3], v, o0);
                    o1 = __builtin_fmaf(hl[1][r & 3], v, o1);
                    o2 = __builtin_fmaf(hl[2][r & 3], v, o2);
                    if ((r & 3) == 3) {
                        hl[0] = hn[0]; hl[1] = hn[1]; hl[2] = hn[2];
                    }
                } else {
                    __bf16 vh, vl;
                    split_bf16(v, vh, vl);
                    fh[r & 7] = vh;
                    fl[r & 7] = vl;
                    if ((r & 7) == 7) {
                        mine[14 + (r >> 3)][ln] = fh;
                        if (X3S_LO_IN_LDS) mine[16 + 14 + (r >> 3)][ln] = fl; else nlo[14 + (r >> 3)] = fl;
                    }
                }
            }
            // the parked activation -- the next layer's, or after a last layer the next block's layer 0 -- becomes
            // the B operand
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                qh[i] = mine[i][ln];
                ql[i] = X3S_LO_IN_LDS ? mine[16 + i][ln] : nlo[i];
            }
        };
        do_layer(std::integral_constant<int, 0>{});
        do_layer(std::integral_constant<int, 1>{});
        do_layer(std::integral_constant<int, 2>{});

        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        if (cur.valid && h == 0) {
            const size_t plane = (size_t)p.Orows * p.Wu;
            float* o = p.out + (size_t)cur.b * 3 * plane + (size_t)(cur.y - p.Orow0) * p.Wu + cur.x;
            o[0] = o0 + or_bits(Wt[OFF_BL + 0], nanm);
            o[plane] = o1 + or_bits(Wt[OFF_BL + 1], nanm);
            o[2 * plane] = o2 + or_bits(Wt[OFF_BL + 2], nanm);
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
    // persistent workgroups from two blocks per CU up (below that there is no next block whose layer 0 could be
    // overlapped); DIINN_X3_KERNEL = 1 / 2 forces the one-block / the persistent form.  Bit-identical results.
    const int force = (int)knob(diinn_knobs().x3_kernel);
    if (force ? force >= 2 : nblk >= 2 * X3_PGRID) {
        DecodeParams pc = p;
        pc.pg[0] = gx; pc.pg[1] = gy; pc.pg[2] = gz;
        const dim3 gridp((unsigned)(nblk < X3_PGRID ? nblk : X3_PGRID));
        if (force == 3) {
            if (sin_mode == DIINN_SIN_HW)
                hipLaunchKernelGGL(decode_bf16x3s_kernel<DIINN_SIN_HW>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
            else if (sin_mode == DIINN_SIN_HW_REDUCED)
                hipLaunchKernelGGL(decode_bf16x3s_kernel<DIINN_SIN_HW_REDUCED>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
            else
                hipLaunchKernelGGL(decode_bf16x3s_kernel<DIINN_SIN_ACCURATE>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
            return hip_status(hipGetLastError());
        }
        if (sin_mode == DIINN_SIN_HW)
            hipLaunchKernelGGL(decode_bf16x3p_kernel<DIINN_SIN_HW>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
        else if (sin_mode == DIINN_SIN_HW_REDUCED)
            hipLaunchKernelGGL(decode_bf16x3p_kernel<DIINN_SIN_HW_REDUCED>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
        else
            hipLaunchKernelGGL(decode_bf16x3p_kernel<DIINN_SIN_ACCURATE>, gridp, dim3(256), 0, (hipStream_t)stream, pc);
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
