// diinn_precompute.hip -- the hoisted 3x3 convolution P = Wx . unfold3x3(feat) + bK (fp32 and bf16 operands)
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
#include "diinn_device.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------
// P kernel: P[b,y,x, i*256+ch] = sum_{c,ky,kx} Wx_i[ch,c,ky,kx] * feat[b,c,y+ky-1,x+kx-1] + bK_i[ch]
// (zero padding; diinn.py:168 unfold + the feature columns of K[i], diinn.py:133,136)
// An implicit-im2col GEMM [cells x 576] . [576 x 1024] on v_mfma_f32_32x32x2_f32.
// Workgroup = 4 waves = a 4-row x 32-column block of LR cells; its 6 x 34 x 64 feature halo
// tile is staged once in LDS (zero padded), and each wave (one row of 32 cells) streams the
// whole packed WP image past it two M-tiles at a time: A operands from the packed image
// (identical for the 4 waves -> one L1 fill), B operands by ds_read_b32 at an immediate
// offset per (tap, channel).  Few registers -> 2 workgroups per CU hide each other's waits.
// ---------------------------------------------------------------------------------
struct PParams {
    const float* feat;   // [B,64,Frows,W] = LR rows [Frow0, Frow0+Frows) of the [B,64,H,W] map (full map: 0, H)
    const float* Wt;
    float* P;            // [B,Prows,W,1024] = LR rows [Prow0, Prow0+Prows) of [B,H,W,1024]
    int B, H, W, r0, r1;
    int Frow0, Frows, Prow0, Prows;
    int msplit;          // the M-tile pairs are divided over `msplit` workgroups (blockIdx.z = b*msplit + part)
    int mp_total;        // M-tile pairs (64 channels each) to compute: 16 = all 1024 channels; LIIF needs the first 4
    int stream_stores;   // P is far larger than the caches: write it with streaming (nt) stores
};

constexpr int PT_ROWS = 4, PT_COLS = 32;                 // cells per workgroup: 4 x 32
constexpr int PT_LR = PT_ROWS + 2, PT_LC = PT_COLS + 2;  // with the 3x3 halo: 6 x 34
constexpr int PT_CH = PT_LR * PT_LC;                     // 204 floats per channel
constexpr int PT_LDS_FLOATS = C_IN * PT_CH;              // 13,056 floats = 52,224 B

constexpr int PT_TR_PITCH = 36;                           // floats per cell in the store transpose (32 + 4: rotates banks)

__global__ __launch_bounds__(256, 2) void precompute_P_kernel(const PParams p) {
    __shared__ __attribute__((aligned(16))) float tile[PT_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float tr[4][32 * PT_TR_PITCH];   // per wave: one 32 x 32 result tile
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int b = blockIdx.z / p.msplit;
    const int part = blockIdx.z - b * p.msplit;
    const int mp_count = p.mp_total / p.msplit, mp_begin = part * mp_count;
    const int x0 = blockIdx.x * PT_COLS;
    const int y0 = p.r0 + blockIdx.y * PT_ROWS;

    // L2 warm-up.  Inside a decode step this kernel starts right after decode_kernel has streamed hundreds of MB of
    // P through every L2, so the 2.25 MiB weight image is gone; all waves of an XCD then walk it in lock-step and
    // the whole first round of workgroups advances at HBM-latency pace (67 % MFMA utilisation at c2 against 80 % warm).
    // Every workgroup touches a slice of the image first so the fills run while the feature tile is staged.  The
    // loaded values are dead: they go to a dead LDS area by LDS-DMA (round 6: l2_touch, diinn_device.h, says why not to registers).
    __shared__ float warm_sink[4][64];
    {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned per_xcd = (nwg + 7) / 8, slot = wg >> 3;      // workgroups are dealt to the 8 XCDs round-robin
        constexpr unsigned LINES = (unsigned)(SZ_WP * sizeof(float) / 128);   // 128-byte lines of the WP image
        const unsigned share = (LINES + per_xcd - 1) / per_xcd;       // lines this workgroup touches (at most 1024)
        const __amdgpu_buffer_rsrc_t wpb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wt + OFF_WP), 0, (int)(SZ_WP * sizeof(float)), 0x00020000);
        float* sink = warm_sink[__builtin_amdgcn_readfirstlane(wave)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned k = threadIdx.x + 256 * i;
            unsigned line = slot * share + k;
            line = (k < share && line < LINES) ? line : 0;            // surplus lanes re-touch line 0
            l2_touch(wpb, line * 128u, sink);
        }
    }

    // stage feat[b, :, y0-1 .. y0+4, x0-1 .. x0+32] (zeros outside the map)
    const float* __restrict__ fb = p.feat + (size_t)b * C_IN * p.Frows * p.W;
    const int fy_hi = p.Frow0 + p.Frows - 1;              // rows of the map the feature window holds: [Frow0, fy_hi]
    static_assert(PT_LDS_FLOATS % 256 == 0, "staging loop has a fixed trip count");
    // fixed trip count, unrolled in batches so that many loads are in flight (a rolled loop would pay
    // one memory latency per element)
#pragma unroll 17
    for (int it = 0; it < PT_LDS_FLOATS / 256; ++it) {
        const int idx = it * 256 + threadIdx.x;
        const int c = idx / PT_CH;
        const int rem = idx - c * PT_CH;
        const int ly = rem / PT_LC, lx = rem - ly * PT_LC;
        const int yy = y0 + ly - 1, xx = x0 + lx - 1;
        const bool ok = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
        // unconditional load from a clamped address, then select: a load under `ok ? .. : 0` compiles
        // to a branch and a vmcnt(0) per element (51 serial memory round trips per workgroup)
        // rows of the map outside the window are only ever read for cells past the band (never stored)
        const int yc = (yy < p.Frow0 ? p.Frow0 : (yy > fy_hi ? fy_hi : yy)) - p.Frow0;
        const int xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
        const float v = fb[((size_t)c * p.Frows + yc) * p.W + xc];
        tile[idx] = ok ? v : 0.0f;
    }
    __syncthreads();

    const int x = x0 + j, y = y0 + wave;
    const bool store = (x < p.W) && (y < p.r1);
    // a wave whose row is past the band still runs (cheap at the band edge) -- no barrier follows,
    // so it may simply leave.
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(store) == 0ull))) return;

    // B operand of k-step kk = 32*t + cp (tap t = ky*3+kx, channel 2*cp + h):
    //   tile[(2cp + h) * PT_CH + (wave + ky) * PT_LC + j + kx]
    const int tb_off = h * PT_CH + wave * PT_LC + j;

    constexpr int PF = P_PREFETCH;
    static_assert(WP_KG % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);   // reads past the end return 0
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WP * sizeof(float)) + mp_begin * (WP_KG * 2 * PIECE_BYTES);   // advances one M-tile pair per iteration
    const float* __restrict__ Bk = p.Wt + OFF_BK + 4 * h;
    // Results leave through a per-wave LDS transpose so that every store instruction writes whole 128-byte lines (the
    // accumulator holds 16 bytes of 8 different lines per lane; storing those directly leaves the merging to L2 -- on
    // the bf16 kernel that cost 15 % of the run), and the stores of M-tile pair mp-1 are spread over the MFMAs of pair
    // mp.  Store instruction i covers cells 8i .. 8i+7: lane L writes chunk L & 7 (4 channels) of cell 8i + (L >> 3).
    float* const trw = tr[wave];
    float* const Prow = p.P + (((size_t)b * p.Prows + (y - p.Prow0)) * p.W + x0) * PCH;   // y < r1 for every storing wave
    f32x16 pa0 = {}, pa1 = {};                                   // the previous pair's results
    auto tr_write = [&](const f32x16& a) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a[4 * g + e];
            *(f32x4*)(trw + j * PT_TR_PITCH + 4 * (2 * g + h)) = v;
        }
    };
    auto store_i = [&](const int i, const int ch0) {
        const int cell = 8 * i + (lane >> 3), q = lane & 7;
        const f32x4 v = *(const f32x4*)(trw + cell * PT_TR_PITCH + 4 * q);
        if (y < p.r1 && x0 + cell < p.W) {
            float* dst = Prow + (size_t)cell * PCH + ch0 + 4 * q;
            if (p.stream_stores) __builtin_nontemporal_store(v, (f32x4*)dst);
            else *(f32x4*)dst = v;
        }
    };
    f32x4 r0v[PF], r1v[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        r0v[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        r1v[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }
#pragma unroll 1
    for (int mp = mp_begin; mp < mp_begin + mp_count; ++mp) {
        // the B operands do not depend on mp: hide the base from LICM, or all 288 LDS reads are
        // hoisted out of this loop and live (spilled) across it
        int off = tb_off;
        asm volatile("" : "+v"(off));
        const float* tbm = tile + off;          // still an LDS (ds_read) address
        const bool prev = mp > mp_begin;
        f32x16 a0, a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 s0 = *(const f32x4*)(Bk + 64 * mp + 8 * g);
            const f32x4 s1 = *(const f32x4*)(Bk + 64 * mp + 32 + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[4 * g + e] = s0[e];
                a1[4 * g + e] = s1[e];
            }
        }
#pragma unroll
        for (int kg = 0; kg < WP_KG; ++kg) {
            const f32x4 u0 = r0v[kg % PF], u1 = r1v[kg % PF];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = 4 * kg + e;
                const int t = kk >> 5, cp = kk & 31;
                const float bv = tbm[(2 * cp) * PT_CH + (t / 3) * PT_LC + (t % 3)];
                a0 = MFMA32(u0[e], bv, a0);
                a1 = MFMA32(u1[e], bv, a1);
            }
            r0v[kg % PF] = ld_piece(wrs, lane_off, wp + (2 * (kg + PF) + 0) * PIECE_BYTES);
            r1v[kg % PF] = ld_piece(wrs, lane_off, wp + (2 * (kg + PF) + 1) * PIECE_BYTES);
            if (prev) {                                          // the previous pair: transpose and store, spread out
                if (kg == 4) tr_write(pa0);
                if (kg >= 8 && kg < 40 && (kg & 7) == 0) store_i((kg - 8) >> 3, 64 * (mp - 1));
                if (kg == 36) tr_write(pa1);
                if (kg >= 40 && kg < 72 && (kg & 7) == 0) store_i((kg - 40) >> 3, 64 * (mp - 1) + 32);
            }
        }
        pa0 = a0;
        pa1 = a1;
        wp += WP_KG * 2 * PIECE_BYTES;
    }
    {
        const int mpl = mp_begin + mp_count - 1;
        tr_write(pa0);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_i(i, 64 * mpl);
        tr_write(pa1);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_i(i, 64 * mpl + 32);
    }
}

// ---------------------------------------------------------------------------------
// precompute_P_bf16_kernel (DIINN_COMPUTE_BF16_FULL): the hoisted 3x3 conv on v_mfma_f32_32x32x16_bf16.
// Same tiling as precompute_P_kernel (4 x 32 cells per workgroup, a wave per cell row, two M-tiles
// advancing together), with the feature halo tile converted to bf16 while it is staged and laid out
// channel-innermost in LDS: [6 x 34 pixels][64 channels + 8 pad] -> the B fragment of a k-step (16
// channels of one tap, 8 per lane-half) is one ds_read_b128, and the 144-byte pixel pitch spreads the
// 32 lanes of a row over all banks.  Accumulation, bias seeds and the stored P stay fp32.
// Bound: the bf16 weight stream through L1 (1 KiB per MFMA per wave) and the 4 KiB/cell store of P.
// ---------------------------------------------------------------------------------
constexpr int PB_PITCH = C_IN + 8;                        // bf16 elements per staged pixel (144 bytes)
constexpr int PB_LDS = PT_CH * PB_PITCH;                  // 14,688 bf16 = 29,376 B
constexpr int PB_ITEMS = PT_CH * (C_IN / 2);              // staged as channel pairs: 6,528 32-bit items

__global__ __launch_bounds__(256, 2) void precompute_P_bf16_kernel(const PParams p) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[PB_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int b = blockIdx.z / p.msplit;
    const int part = blockIdx.z - b * p.msplit;
    const int mp_count = p.mp_total / p.msplit, mp_begin = part * mp_count;
    const int x0 = blockIdx.x * PT_COLS;
    const int y0 = p.r0 + blockIdx.y * PT_ROWS;

    const float* __restrict__ fb = p.feat + (size_t)b * C_IN * p.Frows * p.W;
    const int fy_hi = p.Frow0 + p.Frows - 1;
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll 13
    for (int it = 0; it < (PB_ITEMS + 255) / 256; ++it) {
        const int idx = it * 256 + threadIdx.x;                   // (channel pair, pixel), pixel fastest
        const int cp = idx / PT_CH;
        const int pix = idx - cp * PT_CH;
        const int ly = pix / PT_LC, lx = pix - ly * PT_LC;
        const int yy = y0 + ly - 1, xx = x0 + lx - 1;
        const bool ok = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W) && (idx < PB_ITEMS);
        const int yc = (yy < p.Frow0 ? p.Frow0 : (yy > fy_hi ? fy_hi : yy)) - p.Frow0;
        const int xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
        const int cc = cp < C_IN / 2 ? cp : C_IN / 2 - 1;         // last iteration runs past the item count
        const float v0 = fb[((size_t)(2 * cc) * p.Frows + yc) * p.W + xc];
        const float v1 = fb[((size_t)(2 * cc + 1) * p.Frows + yc) * p.W + xc];
        bf16x2 pk;
        pk[0] = (__bf16)(ok ? v0 : 0.0f);
        pk[1] = (__bf16)(ok ? v1 : 0.0f);
        if (idx < PB_ITEMS) *(bf16x2*)(tile + pix * PB_PITCH + 2 * cp) = pk;
    }
    __syncthreads();

    const int x = x0 + j, y = y0 + wave;
    const bool store = (x < p.W) && (y < p.r1);
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(store) == 0ull))) return;

    // B fragment of k-step ks = 4*tap + cg: tile[((wave + ky) * 34 + j + kx) * 72 + 16cg + 8h .. +7]
    const int tb_off = (wave * PT_LC + j) * PB_PITCH + 8 * h;

    constexpr int PF = P_PREFETCH;
    static_assert(WPB_KS % PF == 0, "ring index must be static");
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);   // reads past the end return 0
    const int lane_off = lane * 16;
    int wp = (int)(OFF_WPB * sizeof(float)) + mp_begin * (WPB_KS * 2 * PIECE_BYTES);
    const float* __restrict__ Bk = p.Wt + OFF_BK + 4 * h;
    const unsigned nanm = derived_nan_mask(p.Wt);              // section WPB missing -> NaN into every P value
    float* __restrict__ Pout = p.P + (((size_t)b * p.Prows + ((y < p.r1 ? y : p.r1 - 1) - p.Prow0)) * p.W + (x < p.W ? x : p.W - 1)) * PCH + 4 * h;
    f32x4 r0v[PF], r1v[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        r0v[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        r1v[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }
#pragma unroll 1
    for (int mp = mp_begin; mp < mp_begin + mp_count; ++mp) {
        int off = tb_off;                                         // hide the base from LICM (see precompute_P_kernel)
        asm volatile("" : "+v"(off));
        const __bf16* tbm = tile + off;
        f32x16 a0, a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 s0 = or_bits(*(const f32x4*)(Bk + 64 * mp + 8 * g), nanm);
            const f32x4 s1 = or_bits(*(const f32x4*)(Bk + 64 * mp + 32 + 8 * g), nanm);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[4 * g + e] = s0[e];
                a1[4 * g + e] = s1[e];
            }
        }
#pragma unroll
        for (int ks = 0; ks < WPB_KS; ++ks) {
            const int t = ks >> 2, cg = ks & 3;
            const bf16x8 bv = *(const bf16x8*)(tbm + ((t / 3) * PT_LC + (t % 3)) * PB_PITCH + 16 * cg);
            a0 = MFMA_BF16(__builtin_bit_cast(bf16x8, r0v[ks % PF]), bv, a0);
            a1 = MFMA_BF16(__builtin_bit_cast(bf16x8, r1v[ks % PF]), bv, a1);
            r0v[ks % PF] = ld_piece(wrs, lane_off, wp + (2 * (ks + PF) + 0) * PIECE_BYTES);
            r1v[ks % PF] = ld_piece(wrs, lane_off, wp + (2 * (ks + PF) + 1) * PIECE_BYTES);
        }
        if (store) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v0, v1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = a0[4 * g + e];
                    v1[e] = a1[4 * g + e];
                }
                *(f32x4*)(Pout + 64 * mp + 8 * g) = v0;
                *(f32x4*)(Pout + 64 * mp + 32 + 8 * g) = v1;
            }
        }
        wp += WPB_KS * 2 * PIECE_BYTES;
    }
}

// ---------------------------------------------------------------------------------
// precompute_P_bf16_wide_kernel (DIINN_COMPUTE_BF16_FULL, large maps): the same bf16 convolution with the weights
// REUSED.  precompute_P_bf16_kernel streams one 1 KiB weight fragment per MFMA through the vector-memory path, and a
// vector-memory instruction blocks its wave for ~60 cycles against the MFMA's 32 (21 % utilisation at c5).  Here a
// workgroup covers 8 rows x 32 columns of LR cells and its four waves split the 32 M-tiles instead of the rows:
// wave w owns M-tiles w, w+4, .., keeps the 36 fragments of the current one in registers and runs all 8 cell rows
// past them, so a fragment is fetched once per 8 MFMAs; the B fragments come from the staged halo tile
// (ds_read_b128 per MFMA, conflict-free as in the narrow kernel), the bias seeds from an LDS table, and the 4 KiB
// rows of P are stored under the next tile's MFMAs.  The halo tile is staged with 16-byte loads.
// ---------------------------------------------------------------------------------
constexpr int PW_ROWS = 8;                                    // cell rows per workgroup
constexpr int PW_LR = PW_ROWS + 2;                            // with the 3x3 halo: 10 x 34 pixels
constexpr int PW_PIX = PW_LR * PT_LC;                         // 340 staged pixels
constexpr int PW_LDS = PW_PIX * PB_PITCH;                     // bf16 elements: 48,960 B
constexpr int PW_XQ = 10;                                     // 16-byte pieces per staged row: columns x0-4 .. x0+35
constexpr int PW_TR_PITCH = 36;                               // floats per cell in the store transpose (32 + 4: rotates banks)
#ifndef PWB_RING_N
#define PWB_RING_N 4                                          // B-fragment ring of the wide kernel (register slots; read-ahead = slots - 1)
#endif

__global__ __launch_bounds__(256, 2) void precompute_P_bf16_wide_kernel(const PParams p) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[PW_LDS];
    __shared__ __attribute__((aligned(16))) float bk[PCH];       // bK, all 1024 channels
    __shared__ __attribute__((aligned(16))) float tr[4][32 * PW_TR_PITCH];   // per wave: one result tile, for the transpose
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * PT_COLS;
    const int y0 = p.r0 + blockIdx.y * PW_ROWS;

    // ---- stage feat[b, :, y0-1 .. y0+8, x0-1 .. x0+32] as bf16, channel-innermost (zeros outside the map)
    const float* __restrict__ fb = p.feat + (size_t)b * C_IN * p.Frows * p.W;
    const int fy_hi = p.Frow0 + p.Frows - 1;
    *(f32x4*)(bk + 4 * threadIdx.x) = or_bits(*(const f32x4*)(p.Wt + OFF_BK + 4 * threadIdx.x), derived_nan_mask(p.Wt));   // section WPB missing -> NaN
    // item = (channel c, staged row ly, piece q): 4 consecutive columns x0 - 4 + 4q .. +3 of one row (16-byte aligned:
    // the launch requires W % 4 == 0); the tile keeps columns x0-1 .. x0+32, i.e. lx = 4q - 3 + e
    constexpr int ITEMS = C_IN * PW_LR * PW_XQ;                  // 6,400 = 25 per thread
    static_assert(ITEMS % 256 == 0, "staging loop has a fixed trip count");
#pragma unroll 5
    for (int it = 0; it < ITEMS / 256; ++it) {
        const int idx = it * 256 + threadIdx.x;
        const int q = idx % PW_XQ;
        const int rest = idx / PW_XQ;
        const int ly = rest % PW_LR, c = rest / PW_LR;
        const int yy = y0 + ly - 1, xx = x0 - 4 + 4 * q;
        const bool rowok = (yy >= 0) && (yy < p.H);
        const int yc = (yy < p.Frow0 ? p.Frow0 : (yy > fy_hi ? fy_hi : yy)) - p.Frow0;
        const int xc = xx < 0 ? 0 : (xx > p.W - 4 ? p.W - 4 : xx);
        const f32x4 v = *(const f32x4*)(fb + ((size_t)c * p.Frows + yc) * p.W + xc);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lx = 4 * q - 3 + e;
            if (lx >= 0 && lx < PT_LC) {
                const bool ok = rowok && (xx + e >= 0) && (xx + e < p.W);
                tile[(ly * PT_LC + lx) * PB_PITCH + c] = (__bf16)(ok ? v[e] : 0.0f);
            }
        }
    }
    __syncthreads();

    // B fragment of (cell row t, k-step ks = 4*tap + cg): tile[((t + ky) * 34 + j + kx) * 72 + 16cg + 8h .. +7]
    const __bf16* __restrict__ tb = tile + j * PB_PITCH + 8 * h;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);
    const int lane_off = lane * 16;
    // WPB piece of (M-tile mo, k-step ks): ((mo >> 1) * 36 + ks) * 2 + (mo & 1)
    auto piece = [&](const int mo, const int ks) {
        return (int)(OFF_WPB * sizeof(float)) + ((((mo >> 1) * WPB_KS + ks) * 2) + (mo & 1)) * PIECE_BYTES;
    };
    f32x4 A[WPB_KS];
#pragma unroll
    for (int ks = 0; ks < WPB_KS; ++ks) A[ks] = ld_piece(wrs, lane_off, piece(wave, ks));

    // Results leave through a per-wave LDS transpose so that every store instruction writes whole 128-byte lines:
    // the accumulator holds, per lane (h, cell j), four 16-byte groups of the cell's 32 channels (chunk q = 2g + h);
    // store instruction i covers cells 8i .. 8i+7, lane L writing chunk L & 7 of cell 8i + (L >> 3).  (Storing the
    // groups straight from the accumulator layout writes 32 bytes per line per instruction and leaves the merging to
    // L2: 1.66 ms against 0.88 ms without stores at c5.)
    float* const trw = tr[wave];
    int pmo = 0, prow = 0;
    bool prowok = false;
    auto tr_write = [&](const f32x16& r) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r[4 * g + e];
            *(f32x4*)(trw + j * PW_TR_PITCH + 4 * (2 * g + h)) = v;
        }
    };
    auto store_i = [&](const int i) {
        int ln = lane;                                           // opaque: the address arithmetic stays at the store (with the row
        asm volatile("" : "+v"(ln));                             // loop unrolled, 8 x 4 destination pointers would be hoisted and spilled)
        const int cell = 8 * i + (ln >> 3), q = ln & 7;
        const f32x4 v = *(const f32x4*)(trw + cell * PW_TR_PITCH + 4 * q);
        if (prowok && x0 + cell < p.W) {
            float* dst = p.P + (((size_t)b * p.Prows + prow) * p.W + x0 + cell) * PCH + 32 * pmo + 4 * q;
            // streaming store: P is hundreds of MB and is read back by the next launch only after all of it has been
            // written, so the lines need not stay in L2 (1.43 -> 1.20 ms at c5; the packed weights stay resident)
            __builtin_nontemporal_store(v, (f32x4*)dst);
        }
    };
    // r03: the staged halo tile is walked ROW by row and every B fragment feeds up to THREE MFMAs -- halo row r is
    // tap row ky of output row r - ky -- into a rolling window of three accumulators (output rows r, r - 1, r - 2).
    // Round 2 read one 1 KiB fragment from LDS per MFMA (8 waves per CU x 1 KiB per 32-cycle MFMA = twice the 128 B/clk
    // of the LDS) into ONE dependent accumulation chain; now 120 LDS reads per 288 MFMAs and three independent chains.
    // Per output value the products are added in the same order as before (ky, kx, channel group = ascending k-step):
    // bit-identical results.  Measured effect on the kernel's time: none (DESIGN.md section 3.4 has the ablations: the
    // loop alone 0.66-0.83 ms, staging +0.1-0.2, stores +0.1-0.25 at c5, and they add instead of overlapping) -- kept for
    // the LDS traffic it removes.  The HBM write stream is NOT the bound it was taken for in round 2: a plain fill of the
    // same 3.7 GB runs at 6.8 TB/s on the same box (tools/hbm_write_roof.py), this kernel writes at ~3.
#pragma unroll 1
    for (int mi = 0; mi < 8; ++mi) {
        const int mo = wave + 4 * mi;                            // this wave's M-tile: channels 32 mo .. 32 mo + 31
        f32x16 acc[3];
        auto seed = [&](f32x16& a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sd = *(const f32x4*)(bk + 32 * mo + 4 * h + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[4 * g + e] = sd[e];
            }
        };
        seed(acc[0]);
        // the B fragments of the 120 (halo row, kx, channel group) steps pass through a register ring, read PWB_AHEAD
        // steps before their MFMAs (an LDS read takes ~100 cycles, three MFMAs)
        constexpr int PWB_RING = PWB_RING_N, PWB_AHEAD = PWB_RING_N - 1, NSTEP = PW_LR * 12;
        bf16x8 bv[PWB_RING];
        auto bfrag = [&](const int n) -> bf16x8 {
            const int r = n / 12, kc = n % 12;
            return *(const bf16x8*)(tb + (r * PT_LC + (kc >> 2)) * PB_PITCH + 16 * (kc & 3));
        };
#pragma unroll
        for (int n = 0; n < PWB_AHEAD; ++n) bv[n % PWB_RING] = bfrag(n);
#pragma unroll
        for (int r = 0; r < PW_LR; ++r)                          // halo row r = map row y0 - 1 + r
#pragma unroll
        for (int kc = 0; kc < 12; ++kc) {                        // (kx, channel group)
            const int n = 12 * r + kc;
            const int kx = kc >> 2, cg = kc & 3;
            if (n + PWB_AHEAD < NSTEP) bv[(n + PWB_AHEAD) % PWB_RING] = bfrag(n + PWB_AHEAD);
            __builtin_amdgcn_sched_barrier(0);                   // the read is issued HERE (left free it sinks to its use and every
            const bf16x8 cur = bv[n % PWB_RING];                 // step waits out the LDS latency)
#pragma unroll
            for (int ky = 2; ky >= 0; --ky) {                    // the oldest output row first: it finishes with this halo row
                const int t = r - ky;
                if (t < 0 || t >= PW_ROWS) continue;
                const int ks = 4 * (3 * ky + kx) + cg;
                acc[t % 3] = MFMA_BF16(__builtin_bit_cast(bf16x8, A[ks]), cur, acc[t % 3]);
                if (t == PW_ROWS - 1 && mi < 7) A[ks] = ld_piece(wrs, lane_off, piece(mo + 4, ks));   // last use of this fragment
            }
            // the previous output row leaves under this halo row's MFMAs: its four stores, spread out
            if (r >= 3 || (r == 0 && mi > 0)) {                  // (row 0: the last output row of the previous M-tile)
                if ((kc % 3) == 1) store_i(kc / 3);
            }
            asm volatile("" ::: "memory");                       // memory operations stay where they are written
            if (kc == 11) {                                      // end of halo row r
                if (r >= 2) {                                    // output row t = r - 2 is complete: into the transpose buffer
                    const int t = r - 2;
                    tr_write(acc[t % 3]);
                    pmo = mo;
                    prow = y0 + t - p.Prow0;
                    prowok = y0 + t < p.r1;
                }
                if (r + 1 < PW_ROWS) seed(acc[(r + 1) % 3]);     // output row r + 1 starts with the next halo row, in the slot just freed
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) store_i(i);
}

// fp32, all 1024 channels, derived sections present: the Winograd form.  One definition for the launch and for
// diinn_p_launch_info (what bench.py reports).
static bool p_uses_winograd(int B, int H, int W, int mp_total, bool bf16, bool derived_ok) {
    const int forcep = (int)knob(diinn_knobs().p_kernel);
    const long long wino_min = knob(diinn_knobs().p_wino_min);
    return derived_ok && !bf16 && mp_total == 16 && forcep != 1 && (forcep == 2 || (long long)B * H * W >= wino_min);
}

// the split-bf16 form of the hoisted conv: blocks of 32 x 8 cells, so from about half a block per CU on; like the Winograd
// choice it depends on the whole map, never on the band (bands stay bit-identical to the same rows of a full launch)
static bool p_uses_x3(int B, int H, int W, int mp_total, int arith, bool derived_ok) {
    return derived_ok && arith == 2 && mp_total == 16 && (long long)B * H * W >= knob(diinn_knobs().p_x3_min);
}

int launch_P(void* stream, const float* feat_dev, const float* packed_dev, float* P_dev,
             int B, int H, int W, int r0, int r1, int mp_total, int arith, const RowWin* feat_win, const RowWin* p_win,
             bool derived_ok) {
    const bool bf16 = arith == 1;
    if (!feat_dev || !packed_dev || !P_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (r0 < 0 || r1 > H || r0 >= r1) return DIINN_ERR_INVALID_ARG;
    const RowWin fw = feat_win ? *feat_win : RowWin{0, H};
    const RowWin pw = p_win ? *p_win : RowWin{0, H};
    // the feature window must hold the band's rows and their 3x3 halo (clipped to the map: zero padding beyond)
    st = check_window(fw.row0, fw.rows, H, r0 > 0 ? r0 - 1 : 0, r1 < H ? r1 + 1 : H);
    if (st) return st;
    st = check_window(pw.row0, pw.rows, H, r0, r1);
    if (st) return st;
    // fp32, all 1024 channels: the Winograd form (diinn_precompute_wino.hip; 2.25x fewer MFMAs, faster from 16x16 cells
    // up: tools/p_time.py).  The choice never depends on the band, so a band stays bit-identical to the same rows of a
    // full launch.  derived_ok: the caller's packed image holds the
    // derived sections (the gather-packed image of a training step does not).  DIINN_P_KERNEL = 1 direct, 2 Winograd.
    {
        if (p_uses_x3(B, H, W, mp_total, arith, derived_ok)) return launch_P_x3(stream, feat_dev, packed_dev, P_dev, B, H, W, r0, r1, fw, pw);
        if (p_uses_winograd(B, H, W, mp_total, bf16, derived_ok)) return launch_P_wino(stream, feat_dev, packed_dev, P_dev, B, H, W, r0, r1, fw, pw);
    }
    // Small maps: split the 1024 output channels over up to 16 workgroups per cell block so the
    // launch still fills the chip (2 workgroups/CU resident -> aim for >= 2 rounds of 512).
    const long long blocks = (long long)((W + PT_COLS - 1) / PT_COLS) * ((r1 - r0 + PT_ROWS - 1) / PT_ROWS) * B;
    int msplit = 1;
    while (msplit < mp_total && blocks * msplit < 1024) msplit *= 2;
    if ((long long)B * msplit > 65535) return DIINN_ERR_TOO_LARGE;
    // streaming stores once P no longer fits beside anything in the 256 MiB last-level cache
    const int stream_stores = (double)B * (r1 - r0) * W * PCH * 4.0 >= 128.0 * 1024 * 1024;
    PParams p{feat_dev, packed_dev, P_dev, B, H, W, r0, r1, fw.row0, fw.rows, pw.row0, pw.rows, msplit, mp_total, stream_stores};
    const dim3 grid((W + PT_COLS - 1) / PT_COLS, (r1 - r0 + PT_ROWS - 1) / PT_ROWS, B * msplit);
    // diagnostic override (tests / A-B timing): DIINN_PBF16_KERNEL = 1 narrow, 2 wide
    const int force = (int)knob(diinn_knobs().pbf16_kernel);
    const dim3 gridw((W + PT_COLS - 1) / PT_COLS, (r1 - r0 + PW_ROWS - 1) / PW_ROWS, B);
    const bool wide_ok = bf16 && mp_total == 16 && W % 4 == 0 && ((uintptr_t)feat_dev % 16) == 0 && B <= 65535;
    const bool wide = wide_ok && (force ? force == 2 : (long long)gridw.x * gridw.y * gridw.z >= 256);
    if (wide)
        hipLaunchKernelGGL(precompute_P_bf16_wide_kernel, gridw, dim3(256), 0, (hipStream_t)stream, p);
    else if (bf16)
        hipLaunchKernelGGL(precompute_P_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(precompute_P_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

extern "C" {

int diinn_p_launch_info(int B, int H, int W, int r0, int r1, int compute, int* algorithm) {
    if (!algorithm || r0 < 0 || r1 > H || r0 >= r1) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (!compute_ok(compute))
        return DIINN_ERR_UNSUPPORTED;
    const bool bf16 = compute == DIINN_COMPUTE_BF16_FULL;
    *algorithm = p_uses_x3(B, H, W, 16, p_arith(compute), true) ? DIINN_P_ALGO_DIRECT_BF16X3
                 : p_uses_winograd(B, H, W, 16, bf16, true) ? DIINN_P_ALGO_WINOGRAD
                 : bf16 ? DIINN_P_ALGO_DIRECT_BF16 : DIINN_P_ALGO_DIRECT;
    return DIINN_OK;
}

int diinn_precompute_P(void* stream, const float* feat_dev, const float* packed_dev,
                       float* P_dev, int B, int H, int W, int r0, int r1) {
    return launch_P(stream, feat_dev, packed_dev, P_dev, B, H, W, r0, r1, 16);
}

int diinn_precompute_P_wpu(void* stream, const float* feat_dev, const float* packed_dev,
                           float* P_dev, int B, int H, int W, int r0, int r1) {
    // the fp32 Winograd form on an image whose validity word is DIINN_PACKED_MAGIC or DIINN_PACKED_MAGIC_WPU (a training image
    // whose WPU section the caller has filled on the device); any other word: NaN into every P value
    if (!feat_dev || !packed_dev || !P_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (r0 < 0 || r1 > H || r0 >= r1) return DIINN_ERR_INVALID_ARG;
    return launch_P_wino(stream, feat_dev, packed_dev, P_dev, B, H, W, r0, r1, RowWin{0, H}, RowWin{0, H}, true);
}

int diinn_precompute_P_ex(void* stream, const float* feat_dev, const float* packed_dev,
                          float* P_dev, int B, int H, int W, int r0, int r1, int compute) {
    if (!compute_ok(compute))
        return DIINN_ERR_UNSUPPORTED;
    return launch_P(stream, feat_dev, packed_dev, P_dev, B, H, W, r0, r1, 16, p_arith(compute), nullptr, nullptr, true);
}

int diinn_precompute_P_win(void* stream, const float* feat_win_dev, int feat_row0, int feat_rows,
                           const float* packed_dev, float* P_win_dev, int p_row0, int p_rows,
                           int B, int H, int W, int r0, int r1, int compute) {
    if (!compute_ok(compute))
        return DIINN_ERR_UNSUPPORTED;
    const RowWin fw{feat_row0, feat_rows}, pw{p_row0, p_rows};
    return launch_P(stream, feat_win_dev, packed_dev, P_win_dev, B, H, W, r0, r1, 16, p_arith(compute),
                    &fw, &pw, true);
}

}  // extern "C"
