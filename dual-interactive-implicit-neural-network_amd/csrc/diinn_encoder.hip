// diinn_encoder.hip -- the RDN encoder trunk on gfx950.
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
//
// Reference: src/models/components/rdn.py:9-105 (RDB_Conv, RDB, RDN config 'B': 16 blocks x 8 dense 3x3 convs,
// growth 64, 1x1 local / global feature fusion).  Every convolution of the trunk has 64 outputs and 64..1024
// inputs: a short M, a long K.  conv_ksplit_kernel splits the REDUCTION inside the workgroup instead of only
// the output across workgroups: a workgroup owns one 32-pixel tile (8 x 4) and its 8 waves each reduce 1/8 of
// the input channels from a slice of the 3x3 halo tile they stage privately in LDS (8 KiB per wave, refilled
// in chunks; no barrier while computing; two workgroups per CU so one's staging overlaps the other's MFMAs),
// then the 8 partial accumulators are summed through LDS and the epilogue (bias, ReLU, residual, up to two
// destinations: the dense-buffer slice and the global-fusion input) is applied once.
//   * small maps (48x48: the reference's runtime_test.py and training patches, 72 tiles): a workgroup per
//     (tile, 32-output half) -- a library convolution is launch/latency-bound there (~35 us per layer);
//   * maps with >= 512 tiles: both output halves per workgroup (half the staging and LDS reads per MFMA).
// This file: that split-K kernel (round 2: every layer of maps up to 8,192 pixels, and the fallback), the streaming
// kernel for the 1x1 layers of bigger maps (conv1x1_stream_kernel, below), and the trunk drivers.  From 8,192 pixels
// on the 3x3 layers run as Winograd F(2x2,3x3) in csrc/diinn_winograd.hip.
// Measured against MIOpen (PyTorch-ROCm), whole encoder: 2.0 vs 7.4 ms at 48x48, 3.8 vs 7.9 at 128x128, 11.7 vs 28.1
// at 256x256, 46.0 vs 108.1 at 512x512 (tools/enc_trunk_time.py; round 1, this kernel alone: 2.1 / 6.5 / 23.9 / 96.9).
// diinn_rdn_forward[_wino] runs the whole trunk (everything after SFENet1) as 147 launches from C++.
#include "diinn_device.h"

constexpr int CS_WAVES = 8;                       // K-split: waves per workgroup, each 1/8 of the input channels
constexpr int CS_TW = 8, CS_TH = 4;               // pixel tile 8 x 4 = 32 = one MFMA N-tile
constexpr int CS_HALO = (CS_TH + 2) * (CS_TW + 2);   // 60 staged pixels per channel for a 3x3 convolution
constexpr int CS_STAGE_FLOATS = 2048;             // per-wave LDS slice (8 KiB): a chunk of 32 channels x 60 (3x3) or 64 channels x 32 (1x1)

struct TagFalse { static constexpr bool value = false; };
struct TagTrue { static constexpr bool value = true; };

struct ConvKsplitParams {
    const float* in;         // input channel planes: in + b*in_bs + c*H*W
    const float* w;          // packed: [half 2][wave 8][tap][channel group of 8][lane 64][4]
    const float* bias;       // [64]
    const float* res;        // optional residual [B,64,H,W] (batch stride res_bs), added after bias (no ReLU with it)
    float* out0;             // destination 0: out0 + b*out0_bs + co*H*W
    float* out1;             // optional destination 1
    long long in_bs, out0_bs, out1_bs, res_bs;
    int Cin, B, H, W, relu;
    // set by the launch: divisions of the workgroup's tile number as multiplications (x / d == (x * m) >> 32 for x, d < 2^16 with
    // m = 2^32 / d + 1; 0: divide) -- an integer division in a kernel's preamble costs every launch ~0.2 us before its first load
    unsigned m_per_image, m_tiles_x;
#ifdef DIINN_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/stamp_report_enc.py)
#endif
};

// NH = 1: a workgroup computes one 32-output half (blockIdx.y) -- twice the workgroups, for maps with few tiles.
// NH = 2: both halves from the same staged features (half the staging and half the LDS reads per MFMA), for maps
//         with enough tiles to fill the chip anyway.
// LAT (3x3, NH = 1: small maps, about one wave per SIMD, so nobody covers a wave's waits): the weight ring holds a
// whole chunk (9 taps) and every request goes out a chunk ahead, B values are read one tap ahead, and the issue order
// is pinned with scheduling barriers -- left alone the compiler sinks each weight request to one MFMA before its use.
template <int TAPS, int NH, bool LAT = false>
__device__ __forceinline__ void conv_ksplit_body(const ConvKsplitParams& p) {
    static_assert(!LAT || (TAPS == 9 && NH == 1), "the latency variant is the small-map 3x3 kernel");
    // staging slices of the 8 waves, then (one output half at a time) their partial sums: 3x3 layers stage 8 channels
    // at a time (32 KiB per workgroup: three workgroups per CU), 1x1 layers 32 or 64 (64 KiB)
    constexpr int SLICE = TAPS == 9 ? 1024 : CS_STAGE_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[CS_WAVES * SLICE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int tiles_x = (p.W + CS_TW - 1) / CS_TW, tiles_y = (p.H + CS_TH - 1) / CS_TH;
    // workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2): give every XCD a contiguous
    // run of tiles, so the workgroups that share 128-byte lines of the feature planes (a tile row is 32-40 bytes)
    // and halo rows meet in ONE L2 instead of fetching the line into four (r02: 23.7 -> 22.1 ms per trunk at 256x256)
    const int per_xcd = gridDim.x >> 3;
    int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (t >= p.B * tiles_x * tiles_y) return;
#ifdef DIINN_STAMPS
    const size_t stamp_base = ((size_t)blockIdx.x * CS_WAVES + wave) * 8;
    if (p.stamps && lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        p.stamps[stamp_base + 6] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    STAMP(0);
    const int b = div_m(t, tiles_x * tiles_y, p.m_per_image);
    t -= b * tiles_x * tiles_y;
    const int ty = div_m(t, tiles_x, p.m_tiles_x), tx = t - ty * tiles_x;
    const int half = NH == 1 ? (int)blockIdx.y : 0;
    const int y0 = ty * CS_TH, x0 = tx * CS_TW;
    const int cw = p.Cin / CS_WAVES;                             // channels reduced by this wave
    const size_t plane = (size_t)p.H * p.W;
    const float* __restrict__ src = p.in + (size_t)b * p.in_bs + (size_t)(wave * cw) * plane;
    float* __restrict__ mine = lds + wave * SLICE;

    // The wave's channels go through its 8 KiB LDS slice in chunks of CH channels (two workgroups fit a CU, so one's
    // staging overlaps the other's MFMAs).  A pieces (weights) of one (chunk, tap): up to GQ groups of 8 channels, all
    // in flight at once; the next (chunk, tap)'s pieces are fetched while this one's MFMAs run.
    // LDS layout of a staged channel: 3x3: 6 halo rows at a pitch of 16 floats, channel pitch 104; 1x1: 32 pixels.
    // An MFMA B read takes, for lane-half h, channel 2s+h at the lane's pixel: with these pitches the 32 lanes
    // of h = 0 fall on banks {0-7, 16-23, 32-39, 48-55} (+ the tap's column shift) and h = 1, 104 = 40 (mod 64)
    // floats further, on the other 32 banks: conflict-free (a dense 6 x 10 tile made every read 2-way conflicted).
    constexpr int NPOS = TAPS == 9 ? CS_HALO : CS_TW * CS_TH;    // staged positions per channel (lanes that stage)
    constexpr int LW = TAPS == 9 ? 16 : CS_TW;
    constexpr int PIX = TAPS == 9 ? 104 : CS_TW * CS_TH;         // channel pitch in floats
    constexpr int OFF = TAPS == 9 ? 1 : 0;
    constexpr int CH = TAPS == 9 ? 16 : (NH == 2 ? 32 : 64);     // 16 x 104 / 64 x 32 floats <= CS_STAGE_FLOATS; NH = 2 holds
                                                                 // weights of both halves: fewer groups per chunk to stay in 128 VGPRs
    constexpr int GQ = CH / 8;
    static_assert(CH * PIX <= CS_STAGE_FLOATS, "chunk must fit the wave's LDS slice");
    const int groups = cw / 8;                                   // pieces (4 k-steps = 8 channels) per tap
    // weight pieces and staged features are fetched through buffer descriptors: descriptor base + scalar byte offset
    // + one constant per-lane offset, i.e. no per-load VALU address arithmetic (the fp32 MFMA shares its issue slot
    // with the VALU; this kernel ran 4.7 VALU instructions per MFMA with plain pointers)
    const int wbytes = TAPS * groups * PIECE_BYTES;              // this wave's slice of the packed weight
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.w + ((size_t)(half * CS_WAVES + wave) * TAPS * groups) * WL_PIECE), 0,
        NH == 1 ? wbytes : CS_WAVES * wbytes + wbytes, 0x00020000);
    const int lane_off = lane * 16;
    const int half_bytes = CS_WAVES * wbytes;                    // distance to the same wave's slice of the other output half
    constexpr bool PIPE = TAPS == 9;                             // 3x3 layers: the fully pipelined path below
    f32x4 a[2][NH][GQ];
    if constexpr (!PIPE) {
        const int g0n = groups < GQ ? groups : GQ;
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
            for (int g = 0; g < GQ; ++g)
                if (g < g0n) a[0][hh][g] = ld_piece(wrs, lane_off, hh * half_bytes + g * PIECE_BYTES);
    }

    // a lane keeps one staged position for the whole kernel (3x3: lanes 0..59 = the 6 x 10 halo of one channel per
    // load; 1x1: 32 pixels x 2 channels per load), so staging is one load and one LDS store per element
    constexpr int CPI = TAPS == 9 ? 1 : 2;                       // channels staged per load instruction
    const int pp = TAPS == 9 ? (lane < NPOS ? lane : NPOS - 1) : (lane & 31);
    const int csel = TAPS == 9 ? 0 : (lane >> 5);
    const int ly = TAPS == 9 ? pp / (CS_TW + 2) : pp / CS_TW;
    const int lx = TAPS == 9 ? pp - ly * (CS_TW + 2) : pp - ly * CS_TW;
    const int yy = y0 + ly - OFF, xx = x0 + lx - OFF;
    const bool ok = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
    const bool mine_lane = TAPS == 9 ? lane < NPOS : true;
    const int yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
    const int xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
    const unsigned plane_b = (unsigned)(plane * sizeof(float));
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)((unsigned)cw * plane_b), 0x00020000);
    const unsigned lsrc_off = (unsigned)csel * plane_b + (unsigned)(yc * p.W + xc) * 4u;
    float* __restrict__ ldst = mine + csel * PIX + (TAPS == 9 ? ly * LW + lx : pp);
    const int pix_off = (j / CS_TW) * LW + (j % CS_TW);          // this lane's pixel inside the staged tile (tap (0,0))

    f32x16 acc[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[hh][r] = 0.0f;
    if constexpr (LAT) {
        f32x4 ring[9];                                           // tap t of the chunk in flight sits in slot t
        float vpre[8];
#define CS_SB() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) ring[tap] = ld_piece(wrs, lane_off, (tap * groups + 0) * PIECE_BYTES);
#pragma unroll
        for (int u = 0; u < 8; ++u) vpre[u] = ld_act(irs, lsrc_off, (unsigned)u * plane_b);
        CS_SB();
        for (int chunk = 0; chunk < groups; ++chunk) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (mine_lane) ldst[u * PIX] = ok ? vpre[u] : 0.0f;
#ifdef DIINN_STAMPS
            if (chunk == 0) STAMP(1);
#endif
            CS_SB();
#pragma unroll
            for (int u = 0; u < 8; ++u) vpre[u] = ld_act(irs, lsrc_off, (unsigned)(8 * (chunk + 1) + u) * plane_b);
            const float* __restrict__ bbase = mine + h * PIX + pix_off;
            float bv[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[0][e] = bbase[(2 * e) * PIX];
            CS_SB();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)                  // channel 8 chunk + 2e + h under tap + 1
                        bv[(tap + 1) & 1][e] = bbase[(2 * e) * PIX + ((tap + 1) / 3) * LW + ((tap + 1) % 3)];
                }
                f32x4 a4 = ring[tap];
                CS_SB();
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0] = MFMA32(a4[e], bv[tap & 1][e], acc[0]);
                // the same tap of the next chunk (past the wave's slice: other slices' data or zeros, never used)
                ring[tap] = ld_piece(wrs, lane_off, (tap * groups + chunk + 1) * PIECE_BYTES);
                CS_SB();
            }
        }
#undef CS_SB
    } else
    if constexpr (PIPE) {
        // 3x3 layers, everything prefetched.  A chunk is 8 channels = one weight piece per (tap, half):
        //  * weight pieces live in a ring of 3 taps and are requested two taps (16 NH MFMAs) before use; 9 taps per
        //    chunk keep the ring phase equal from chunk to chunk, so every index is static and nothing is copied;
        //  * the next chunk's 8 feature values per lane are requested before this chunk's MFMAs and stored to LDS
        //    after them.
        // Requests past the wave's slice are harmless: the descriptors' range checks return 0 / other slices' data
        // that is never used.
        f32x4 ring[3][NH];
        float vpre[8];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            ring[0][hh] = ld_piece(wrs, lane_off, hh * half_bytes + (0 * groups + 0) * PIECE_BYTES);
            ring[1][hh] = ld_piece(wrs, lane_off, hh * half_bytes + (1 * groups + 0) * PIECE_BYTES);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) vpre[u] = ld_act(irs, lsrc_off, (unsigned)u * plane_b);
        for (int chunk = 0; chunk < groups; ++chunk) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (mine_lane) ldst[u * PIX] = ok ? vpre[u] : 0.0f;
#ifdef DIINN_STAMPS
            if (chunk == 0) STAMP(1);
#endif
#pragma unroll
            for (int u = 0; u < 8; ++u) vpre[u] = ld_act(irs, lsrc_off, (unsigned)(8 * (chunk + 1) + u) * plane_b);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int nt = tap + 2;                          // tap fetched now: (chunk, nt) or (chunk + 1, nt - 9)
#pragma unroll
                for (int hh = 0; hh < NH; ++hh)
                    ring[nt % 3][hh] = ld_piece(wrs, lane_off, hh * half_bytes +
                                                ((nt < 9 ? nt : nt - 9) * groups + (nt < 9 ? chunk : chunk + 1)) * PIECE_BYTES);
                const float* __restrict__ bsrc = mine + h * PIX + pix_off + (tap / 3) * LW + (tap % 3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bv = bsrc[(2 * e) * PIX];        // channel 8 chunk + 2e + h
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh) acc[hh] = MFMA32(ring[tap % 3][hh][e], bv, acc[hh]);
                }
            }
        }
    } else
    for (int cbase = 0; cbase < cw; cbase += CH) {
        const int cc = cw - cbase < CH ? cw - cbase : CH;        // channels in this chunk (multiple of 8)
        const int gcount = cc / 8, gbase = cbase / 8;
        const int nc = cw - cbase - CH;                          // channels left after this chunk
        const int gnext = nc <= 0 ? 0 : (nc < CH ? nc / 8 : GQ);
        // ---- stage the chunk (wave-private; the previous chunk's LDS reads were issued earlier in program order)
        for (int c0 = 0; c0 < cc; c0 += 8) {                     // 8 (3x3) or 4 (1x1) independent loads in flight
            float v[8 / CPI];
#pragma unroll
            for (int u = 0; u < 8 / CPI; ++u) v[u] = ld_act(irs, lsrc_off, (unsigned)(cbase + c0 + u * CPI) * plane_b);
#pragma unroll
            for (int u = 0; u < 8 / CPI; ++u)
                if (mine_lane) ldst[(c0 + u * CPI) * PIX] = ok ? v[u] : 0.0f;
        }
        // ---- this chunk's share of the reduction: k-step = (tap, channel pair).  The pieces of tap t sit in
        // a[t & 1] and tap t+1's are fetched into the other buffer meanwhile (static indices: no register copies
        // inside a chunk); FULL chunks (all GQ groups present, the common case) run without per-group branches.
        auto taps = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                f32x4 (&cur)[NH][GQ] = a[tap & 1];
                f32x4 (&nxt)[NH][GQ] = a[(tap + 1) & 1];
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) {
                    if (tap + 1 < TAPS) {
#pragma unroll
                        for (int g = 0; g < GQ; ++g)
                            if (FULL || g < gcount)
                                nxt[hh][g] = ld_piece(wrs, lane_off, hh * half_bytes + ((tap + 1) * groups + gbase + g) * PIECE_BYTES);
                    } else {                                     // last tap: tap 0 of the next chunk
#pragma unroll
                        for (int g = 0; g < GQ; ++g)
                            if (g < gnext) nxt[hh][g] = ld_piece(wrs, lane_off, hh * half_bytes + (gbase + GQ + g) * PIECE_BYTES);
                    }
                }
                const int toff = TAPS == 9 ? (tap / 3) * LW + (tap % 3) : 0;
                const float* __restrict__ bsrc = mine + h * PIX + pix_off + toff;
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    if (FULL || g < gcount) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float bv = bsrc[(8 * g + 2 * e) * PIX];                                // channel cbase + 8g + 2e + h
#pragma unroll
                            for (int hh = 0; hh < NH; ++hh) acc[hh] = MFMA32(cur[hh][g][e], bv, acc[hh]);
                        }
                    }
                }
            }
        };
        if (gcount == GQ) taps(TagTrue{});
        else taps(TagFalse{});
        if ((TAPS & 1) != 0) {                                   // odd tap count: the next chunk's tap 0 landed in a[1]
#pragma unroll
            for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                for (int g = 0; g < GQ; ++g) a[0][hh][g] = a[1][hh][g];
        }
    }

    STAMP(2);
    // ---- sum the 8 partial tiles through LDS (one output half per pass: 32 KiB), then the epilogue
    const int y = y0 + j / CS_TW, x = x0 + j % CS_TW;
    const bool inside = (y < p.H) && (x < p.W);
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        __syncthreads();                                         // every wave is done with the LDS contents of the previous phase
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[(wave * 16 + r) * 64 + lane] = acc[hh][r];
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {                         // wave w finishes accumulator rows 2w, 2w+1 of this half
            const int r = 2 * wave + rr;
            float v = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < CS_WAVES; ++w8) v += lds[(w8 * 16 + r) * 64 + lane];
            const int co = 32 * (NH == 1 ? half : hh) + (r & 3) + 8 * (r >> 2) + 4 * h;
            v += p.bias[co];
            if (p.relu) v = v > 0.0f ? v : 0.0f;
            if (inside) {
                const size_t o = (size_t)co * plane + (size_t)y * p.W + x;
                if (p.res) v += p.res[(size_t)b * p.res_bs + o];
                p.out0[(size_t)b * p.out0_bs + o] = v;
                if (p.out1) p.out1[(size_t)b * p.out1_bs + o] = v;
            }
        }
#ifdef DIINN_STAMPS
        if (hh == 0) STAMP(3);
#endif
    }
    STAMP(4);
}

// 3x3 layers: 32 KiB of LDS and <= 80 VGPRs -> three workgroups (6 waves per SIMD) share a CU, which covers one
// workgroup's prologue and reduction phases with the others' MFMAs; 1x1 layers: two workgroups per CU.
template <int NH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6)))
void conv_ksplit_kernel_3x3(const ConvKsplitParams p) { conv_ksplit_body<9, NH>(p); }

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))
void conv_ksplit_kernel_3x3_lat(const ConvKsplitParams p) { conv_ksplit_body<9, 1, true>(p); }

template <int NH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))
void conv_ksplit_kernel_1x1(const ConvKsplitParams p) { conv_ksplit_body<1, NH>(p); }

// ---------------------------------------------------------------------------------------------------------------
// 1x1 layers on big maps (LFF 576 -> 64, GFF.0 1024 -> 64): conv1x1_stream_kernel.  A 1x1 convolution is a plain GEMM
// over the flattened plane, so nothing needs staging: a workgroup owns 128 consecutive pixels, lane j holds pixels
// 4j..4j+3 (one 16-byte load per channel; pixel 4j+q belongs to MFMA N-tile q), and each of the 4 waves (one per
// SIMD) reduces a quarter of the input channels into 4 N-tiles x 2 output halves = 8 accumulators with operands
// streamed through a three-stage register ring: 6 vector-memory instructions and no VALU per 32 MFMAs.  The four
// partial sums meet once in LDS; wave w then finishes channels 8w..8w+3 (+4h, +32 half) of all 128 pixels, so every
// store is 16 bytes per lane, 512 bytes contiguous.  (conv_ksplit_kernel_1x1 ran these layers at 44 % of the MFMA
// peak: its 32-pixel tiles stage 64 channels per wave through LDS and wait for every chunk.)
constexpr int S1_PIX = 128;                           // pixels per workgroup
constexpr int S1_PITCH = 20;                          // LDS floats per (wave, q, lane): 16 + 4 padding
constexpr int S1_LDS_FLOATS = 4 * 4 * 64 * S1_PITCH;  // one output half of the four partial sums: 80 KiB

__global__ __launch_bounds__(256, 1) void conv1x1_stream_kernel(const ConvKsplitParams p) {
    __shared__ __attribute__((aligned(16))) float red[S1_LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_b = (unsigned)(plane * sizeof(float));
    const int blocks_per = (int)((plane + S1_PIX - 1) / S1_PIX);
    const int per_xcd = gridDim.x >> 3;                          // every XCD a contiguous run of blocks
    int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (t >= p.B * blocks_per) return;
    const int b = __builtin_amdgcn_readfirstlane(div_m(t, blocks_per, p.m_per_image));
    const int blk = t - b * blocks_per;
    const int g8 = p.Cin / 64;                                   // pieces per part of the packed weight
    const int n = 2 * g8;                                        // chunks of 8 channels this wave reduces (two parts)
    const int lane_off = lane * 16;
    // packed weight [half 2][part 8][group g8][lane 64][4]: this wave walks parts 2w and 2w + 1, which are contiguous
    const int half_bytes = CS_WAVES * g8 * PIECE_BYTES;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.w + (size_t)(2 * wave) * g8 * WL_PIECE), 0, half_bytes + n * PIECE_BYTES, 0x00020000);
    const float* __restrict__ in_w = p.in + (size_t)b * p.in_bs + (size_t)(wave * (p.Cin / 4)) * plane;
    constexpr unsigned OUTSIDE = 0x80000000u;
    const size_t pix0 = (size_t)blk * S1_PIX + 4 * j;           // this lane's first pixel
    const bool pix_in = pix0 < plane;                            // plane % 4 == 0: a lane's 4 pixels are in or out together
    const unsigned roff = pix_in ? (unsigned)h * plane_b + (unsigned)pix0 * 4u : OUTSIDE;

    f32x4 A[3][2], R[3][4];
    f32x16 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][hh][r] = 0.0f;
#define S1_SB() __builtin_amdgcn_sched_barrier(0)
    auto chunk_rsrc = [&](int c) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(in_w + (size_t)8 * c * plane), 0, (int)(8u * plane_b), 0x00020000);
    };
    // request #IDX (0..5) of chunk c into stage S: 2 weight pieces, 4 pixel rows (channels 8c + 2e + h)
    auto request = [&](auto S_, auto IDX_, int c, __amdgpu_buffer_rsrc_t irs) {
        constexpr int S = decltype(S_)::value, IDX = decltype(IDX_)::value;
        if constexpr (IDX < 2) A[S][IDX] = ld_piece(wrs, lane_off, IDX * half_bytes + c * PIECE_BYTES);
        else R[S][IDX - 2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, (int)roff, (int)((unsigned)(2 * (IDX - 2)) * plane_b), 0));
    };
    auto chunk = [&](auto S_, int cload) {
        constexpr int S = decltype(S_)::value;
        constexpr int SL = (S + 2) % 3;
        const __amdgpu_buffer_rsrc_t irs = chunk_rsrc(cload);
        auto estep = [&](auto E_) {
            constexpr int E = decltype(E_)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q][0] = MFMA32(A[S][0][E], R[S][E][q], acc[q][0]);
                S1_SB();
                acc[q][1] = MFMA32(A[S][1][E], R[S][E][q], acc[q][1]);
                if (q == 0) {                                    // 6 requests over the chunk's 16 MFMA pairs
                    if constexpr (E == 0) request(IC<SL>{}, IC<0>{}, cload, irs);
                    if constexpr (E == 1) request(IC<SL>{}, IC<2>{}, cload, irs);
                    if constexpr (E == 2) request(IC<SL>{}, IC<4>{}, cload, irs);
                }
                if (q == 2) {
                    if constexpr (E == 0) request(IC<SL>{}, IC<1>{}, cload, irs);
                    if constexpr (E == 1) request(IC<SL>{}, IC<3>{}, cload, irs);
                    if constexpr (E == 2) request(IC<SL>{}, IC<5>{}, cload, irs);
                }
                S1_SB();
            }
        };
        estep(IC<0>{}); estep(IC<1>{}); estep(IC<2>{}); estep(IC<3>{});
    };
    auto request_all = [&](auto S_, int c) {
        const __amdgpu_buffer_rsrc_t irs = chunk_rsrc(c);
        request(S_, IC<0>{}, c, irs); request(S_, IC<1>{}, c, irs); request(S_, IC<2>{}, c, irs);
        request(S_, IC<3>{}, c, irs); request(S_, IC<4>{}, c, irs); request(S_, IC<5>{}, c, irs);
    };
    request_all(IC<0>{}, 0);
    request_all(IC<1>{}, 1);
    S1_SB();
    auto ahead = [&](int cc) { return cc + 2 < n ? cc + 2 : n - 1; };
    int c = 0;
    for (; c + 3 <= n; c += 3) {
        chunk(IC<0>{}, ahead(c));
        chunk(IC<1>{}, ahead(c + 1));
        chunk(IC<2>{}, ahead(c + 2));
    }
    if (c < n) chunk(IC<0>{}, n - 1);
    if (c + 1 < n) chunk(IC<1>{}, n - 1);
#undef S1_SB

    // ---- the four partial sums meet in LDS, one output half per pass; wave w finishes accumulator registers 4w..4w+3
    // (channels 32 half + 8 w + e + 4 h) of all four N-tiles = the lane's 4 consecutive pixels
    const __amdgpu_buffer_rsrc_t o0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out0 + (size_t)b * p.out0_bs), 0, (int)(64u * plane_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t o1 = __builtin_amdgcn_make_buffer_rsrc((void*)((p.out1 ? p.out1 : p.out0) + (size_t)b * (p.out1 ? p.out1_bs : p.out0_bs)), 0, (int)(64u * plane_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)((p.res ? p.res : p.in) + (size_t)b * (p.res ? p.res_bs : p.in_bs)), 0, (int)(64u * plane_b), 0x00020000);
    const unsigned ooff = pix_in ? (unsigned)(4 * h) * plane_b + (unsigned)pix0 * 4u : OUTSIDE;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if (hh) __syncthreads();                                 // the previous pass has been read
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[q][hh][4 * g + e];
                *reinterpret_cast<f32x4*>(red + (((wave * 4 + q) * 64 + lane) * S1_PITCH + 4 * g)) = v;
            }
        __syncthreads();
        f32x4 y[4];                                              // [q] x components e
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 s = *reinterpret_cast<const f32x4*>(red + (((0 * 4 + q) * 64 + lane) * S1_PITCH + 4 * wave));
#pragma unroll
            for (int w4 = 1; w4 < 4; ++w4) s += *reinterpret_cast<const f32x4*>(red + (((w4 * 4 + q) * 64 + lane) * S1_PITCH + 4 * wave));
            y[q] = s;
        }
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + 32 * hh + 8 * wave + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned soff = (unsigned)(32 * hh + 8 * wave + e) * plane_b;
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[q] = y[q][e] + bias4[e];
                if (p.relu) o[q] = relu0(o[q]);
            }
            if (p.res) o += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)ooff, (int)soff, 0));
            st_b128(o, o0, ooff, soff);
            if (p.out1) st_b128(o, o1, ooff, soff);
        }
    }
}

static bool conv1x1_stream_ok(const ConvKsplitParams& p) {
    const long long plane = (long long)p.H * p.W;
    auto a16 = [](const void* q) { return (((size_t)q) & 15) == 0; };
    const long long min_blocks = knob(diinn_knobs().enc_s1_min_blocks);
    return plane % 4 == 0 && p.Cin % 64 == 0 && (long long)p.B * ((plane + S1_PIX - 1) / S1_PIX) >= min_blocks &&
           plane * 4 * 64 <= 0x7FFFFFFFLL && a16(p.in) && a16(p.w) && a16(p.bias) && a16(p.out0) && a16(p.out1) && a16(p.res) &&
           p.in_bs % 4 == 0 && p.out0_bs % 4 == 0 && p.out1_bs % 4 == 0 && p.res_bs % 4 == 0;
}

// ---------------------------------------------------------------------------------------------------------------
// SFENet1 (rdn.py:96): 3x3, n_colors (<= 4) -> 64, zero padding.  0.2 % of the encoder's arithmetic and no use for the
// matrix cores (K = 27): a workgroup owns 64 pixels, wave q of 4 the outputs 16q .. 16q+15; a thread keeps its pixel's
// 9 * Cin inputs in registers and walks its 16 outputs with the weights as scalar operands (wave-uniform addresses:
// fetched through the scalar cache); stores are 256 contiguous bytes per wave and output channel.  It exists so that
// the whole encoder runs through this library (no library convolution, no find pass on the first call).
struct Sfe1Params {
    const float* in;     // [B,Cin,H,W]
    const float* w;      // [64,Cin,3,3], the reference's own layout
    const float* bias;   // [64]
    float* out;          // [B,64,H,W]
    int B, H, W;
};

template <int CIN>
__global__ __launch_bounds__(256) void sfe1_conv_kernel(const Sfe1Params p) {
    const long long plane = (long long)p.H * p.W;
    const long long pix = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int og = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // this wave's 16 outputs
    const int b = blockIdx.y;
    if (pix >= plane) return;
    const int y = (int)(pix / p.W), x = (int)(pix - (long long)y * p.W);
    float v[CIN * 9];
    const float* __restrict__ ib = p.in + (size_t)b * CIN * plane;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const int yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy), xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
            const float u = ib[(size_t)c * plane + (size_t)yc * p.W + xc];   // clamped address, then select: no branch per element
            v[c * 9 + t] = ok ? u : 0.0f;
        }
    float* __restrict__ ob = p.out + (size_t)b * 64 * plane + pix;
#pragma unroll 4
    for (int oo = 0; oo < 16; ++oo) {
        const int o = 16 * og + oo;
        const float* __restrict__ wo = p.w + o * CIN * 9;        // wave-uniform: scalar loads
        float acc = p.bias[o];
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) acc = __builtin_fmaf(wo[k], v[k], acc);
        ob[(size_t)o * plane] = acc;
    }
}

static int launch_conv_ksplit(void* stream, const ConvKsplitParams& p_in, int taps) {
    const int tiles = ((p_in.W + CS_TW - 1) / CS_TW) * ((p_in.H + CS_TH - 1) / CS_TH) * p_in.B;
    // both output halves per workgroup once the tiles alone give every CU two workgroups; otherwise one half each
    const bool both = tiles >= 512;
    const dim3 grid((tiles + 7) / 8 * 8, both ? 1 : 2);          // a multiple of 8: see the XCD mapping in the kernel
    ConvKsplitParams p = p_in;
#ifdef DIINN_STAMPS
    p.stamps = g_stamps;
#endif
    const bool no_stream = knob(diinn_knobs().enc_no_stream1x1) != 0;   // A/B switches for tools/
    const int lat_max = (int)knob(diinn_knobs().enc_lat_max_tiles);
    const bool lat = tiles <= lat_max;                           // <= 2 workgroups per CU: about one wave per SIMD
    const long long tiles_x = (p.W + CS_TW - 1) / CS_TW, per_image = tiles_x * ((p.H + CS_TH - 1) / CS_TH);
    p.m_per_image = magic_m(per_image, (long long)grid.x);
    p.m_tiles_x = magic_m(tiles_x, per_image);
    if (taps == 1 && !no_stream && conv1x1_stream_ok(p)) {
        const long long blocks_per = ((long long)p.H * p.W + S1_PIX - 1) / S1_PIX, blocks = (long long)p.B * blocks_per;
        p.m_per_image = magic_m(blocks_per, (blocks + 7) / 8 * 8);
        hipLaunchKernelGGL(conv1x1_stream_kernel, dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream, p);
    } else if (taps == 9) {
        if (both) hipLaunchKernelGGL(conv_ksplit_kernel_3x3<2>, grid, dim3(512), 0, (hipStream_t)stream, p);
        else if (lat) hipLaunchKernelGGL(conv_ksplit_kernel_3x3_lat, grid, dim3(512), 0, (hipStream_t)stream, p);
        else      hipLaunchKernelGGL(conv_ksplit_kernel_3x3<1>, grid, dim3(512), 0, (hipStream_t)stream, p);
    } else {
        if (both) hipLaunchKernelGGL(conv_ksplit_kernel_1x1<2>, grid, dim3(512), 0, (hipStream_t)stream, p);
        else      hipLaunchKernelGGL(conv_ksplit_kernel_1x1<1>, grid, dim3(512), 0, (hipStream_t)stream, p);
    }
    return hip_status(hipGetLastError());
}

extern "C" {

int diinn_conv_ksplit(void* stream, const float* in_dev, long long in_batch_stride, int Cin, int taps,
                     const float* packed_w_dev, const float* bias_dev,
                     const float* res_dev, long long res_batch_stride,
                     float* out0_dev, long long out0_batch_stride, float* out1_dev, long long out1_batch_stride,
                     int relu, int B, int H, int W) {
    if (!in_dev || !packed_w_dev || !bias_dev || !out0_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (taps != 9 && taps != 1) return DIINN_ERR_UNSUPPORTED;
    if (Cin <= 0 || Cin % 64) return DIINN_ERR_UNSUPPORTED;
    if ((long long)((W + CS_TW - 1) / CS_TW) * ((H + CS_TH - 1) / CS_TH) * B > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    if ((long long)(Cin / CS_WAVES) * H * W * 4 > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;   // a wave's channel slice is addressed with 32-bit byte offsets
    ConvKsplitParams p;
    p.in = in_dev; p.w = packed_w_dev; p.bias = bias_dev; p.res = res_dev; p.out0 = out0_dev; p.out1 = out1_dev;
    p.in_bs = in_batch_stride; p.out0_bs = out0_batch_stride; p.out1_bs = out1_batch_stride; p.res_bs = res_batch_stride;
    p.Cin = Cin; p.B = B; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
    return launch_conv_ksplit(stream, p, taps);
}

int diinn_sfe1_forward(void* stream, const float* x_dev, int Cin, const float* w_dev, const float* bias_dev, float* out_dev,
                       int B, int H, int W) {
    if (!x_dev || !w_dev || !bias_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin < 1 || Cin > 4) return DIINN_ERR_UNSUPPORTED;
    Sfe1Params p{x_dev, w_dev, bias_dev, out_dev, B, H, W};
    const long long plane = (long long)H * W;
    const dim3 grid((unsigned)((plane + 63) / 64), (unsigned)B);
    switch (Cin) {
        case 1: hipLaunchKernelGGL(sfe1_conv_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
        case 2: hipLaunchKernelGGL(sfe1_conv_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
        case 3: hipLaunchKernelGGL(sfe1_conv_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
        default: hipLaunchKernelGGL(sfe1_conv_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p); break;
    }
    return hip_status(hipGetLastError());
}

size_t diinn_rdn_packed_floats(void) {
    // SFENet2 64*64*9; 16 x [8 dense convs 64*(64..512)*9 + LFF 64*576]; GFF 64*1024 + 64*64*9
    size_t n = (size_t)64 * 64 * 9;
    for (int c = 0; c < 8; ++c) n += (size_t)16 * 64 * (64 + 64 * c) * 9;
    n += (size_t)16 * 64 * 576 + (size_t)64 * 1024 + (size_t)64 * 64 * 9;
    return n;
}

size_t diinn_rdn_workspace_floats(int B, int H, int W) {
    // the four one-algorithm entry points (kept for one ABI version): [the F(4x4,3x3) kernel's split area (counters first: they
    // are zeroed at the start of every forward)][two dense buffers 576][global fusion input 1024][64]
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return diinn_conv_wino4_workspace_floats() + (size_t)B * H * W * (2 * 576 + 1024 + 64);
}

size_t diinn_rdn_planes_floats(int algo, int B, int H, int W) {
    // diinn_rdn_forward_ex's planes: [two dense buffers 576][global fusion input 1024][64] (+ for DIINN_RDN_ALGO_X3 the
    // split-format copy of one dense buffer: [B][72 groups][hi, lo][H][W] x 16 bytes); the F(4x4) split area is a buffer of its own there
    if (B <= 0 || H <= 0 || W <= 0 || algo < DIINN_RDN_ALGO_AUTO || algo > DIINN_RDN_ALGO_X3) return 0;
    return (size_t)B * H * W * (2 * 576 + 1024 + 64 + (algo == DIINN_RDN_ALGO_X3 ? 576 : 0));
}

size_t diinn_rdn_wino_packed_floats(void) {
    // the 3x3 layers only, 16 floats per (output, input) pair: SFENet2, 16 x 8 dense convs, GFF.1
    size_t n = (size_t)2 * 64 * 64 * 16;
    for (int c = 0; c < 8; ++c) n += (size_t)16 * 64 * (64 + 64 * c) * 16;
    return n;
}

size_t diinn_rdn_wino4_packed_floats(void) {
    // the 3x3 layers only, 36 floats per (output, input) pair (F(4x4, 3x3): csrc/diinn_winograd4.hip)
    return diinn_rdn_wino_packed_floats() / 16 * 36;
}

size_t diinn_rdn_x3_workspace_floats(int B, int H, int W) {
    // diinn_rdn_workspace_floats + the split-format copy of one dense buffer ([B][72 groups][hi, lo][H][W] x 16 bytes)
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return diinn_rdn_workspace_floats(B, H, W) + (size_t)B * 576 * H * W;
}

size_t diinn_rdn_x3_packed_floats(void) {
    // the 3x3 layers, 9 taps x (hi + lo) bf16 = 9 floats per (output, input) pair: SFENet2, 16 x 8 dense convs, GFF.1 in
    // execution order; then the 16 local-fusion 1x1 layers, 1 float per pair
    size_t n = (size_t)2 * 64 * 64 * 9;
    for (int c = 0; c < 8; ++c) n += (size_t)16 * 64 * (64 + 64 * c) * 9;
    return n + (size_t)16 * 64 * 576;
}

int diinn_rdn_wino4_applies(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const long long px = (long long)B * H * W;
    if (px < knob(diinn_knobs().enc_wino_min)) return 0;         // the split-K kernel's maps
    const long long forced = knob(diinn_knobs().enc_wino4_min);
    if (forced >= 0) return px >= forced;
    const long long items4 = 2LL * B * (((long long)((W + 3) / 4) * ((H + 3) / 4) + 31) / 32);   // blocks of 32 consecutive tiles x 2 output halves
    const long long blocks2 = (long long)B * (((W + 1) / 2 + 7) / 8) * (((H + 1) / 2 + 3) / 4);
    const int ncu = device_cus();
    const double r4 = 1.40 * w4_rounds(items4, true);            // (the trunk gives the kernel its workspace: a partly filled last round is split)
    const double r2w = (double)((blocks2 + ncu - 1) / ncu), r2h = 0.57 * (double)((2 * blocks2 + ncu - 1) / ncu);
    return r4 < 0.97 * (r2w < r2h ? r2w : r2h);
}

// planes: diinn_rdn_planes_floats; w4ws: the F(4x4) kernel's split area (diinn_conv_wino4_workspace_floats; NULL: no layer is split)
static int rdn_forward_impl(void* stream, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                            const float* packed_wino4_dev, const float* packed_x3_dev,
                            const float* biases_dev, float* planes, float* w4ws, float* out_dev, int B, int H, int W,
                            bool zero_status = false) {
    if (!sfe1_dev || !packed_dev || !biases_dev || !planes || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    const long long hw = (long long)H * W;
    // Winograd blocks are 16 x 8 pixels and a workgroup walks all input channels: faster than the split-K kernel from
    // about 90 x 90 pixels up (tools/r02_ab_env.sh: 96x96 3.6 vs 4.5 ms, 64x64 3.4 vs 2.1 ms per trunk)
    const long long wino_min = knob(diinn_knobs().enc_wino_min);
    const bool wino = (packed_wino_dev || packed_wino4_dev) && (long long)B * hw >= wino_min;
    // F(4x4, 3x3) (csrc/diinn_winograd4.hip): 1.78x fewer MFMAs again, in work items of 128 x 4 pixels x one output half.
    // Both kernels run in rounds of one workgroup per CU, and measured over 192 .. 512-pixel maps one F(4x4) round
    // costs 1.40 F(2x2) rounds of whole blocks (16 x 8 pixels x both halves; a round of halves 0.57; r04 measured 1.44, r05 1.40): the cheaper one
    // by that count runs.  DIINN_ENC_WINO4_MIN = n >= 0 replaces the rule by "from n pixels on".
    const bool wino4 = packed_wino4_dev && wino && diinn_rdn_wino4_applies(B, H, W);
    if (wino && !wino4 && !packed_wino_dev) return DIINN_ERR_INVALID_ARG;   // this map runs F(2x2): its image is needed
    // the F(4x4) kernel's split area leads the workspace; its arrival counters (the first 2 KiB) are zeroed here, once per
    // forward, whatever a caller or an aborted launch left there (a memset node: capture-safe).  NOT the sticky status
    // word behind them (word 1023 of the first 4 KiB, which the workspace's owner zeroes once when it allocates): a
    // hand-off that gave up stays visible -- NaN features from then on -- until diinn_conv_wino4_ws_status has cleared it
    const size_t w4ws_floats = w4ws ? diinn_conv_wino4_workspace_floats() : 0;
    if (wino4 && w4ws) {
        st = hip_status(hipMemsetAsync(w4ws, 0, zero_status ? 4096 : DIINN_WINO4_COUNTER_BYTES, (hipStream_t)stream));
        if (st) return st;
    }
    float* buf[2] = {planes, planes + (size_t)B * 576 * hw};                   // dense buffers [B,576,H,W]
    float* gff_in = planes + (size_t)2 * B * 576 * hw;                           // [B,1024,H,W]
    float* tmp = gff_in + (size_t)B * 1024 * hw;                                 // [B,64,H,W]
    // split-bf16 3x3 layers (csrc/diinn_conv_x3.hip; optional): blocks of 32 x 8 pixels; they pay from about 0.8 blocks
    // per CU on (measured per trunk, x3 vs Winograd: 160x160 7.8 vs 6.2 ms, 192x192 7.2 vs 9.3, 224x224 7.5 vs 11.4, 256x256 8.6 vs 11.8,
    // 384x384 20.8 vs 28.0, 512x512 32.9 vs 46.5)
    const bool x3 = packed_x3_dev && (long long)B * hw >= knob(diinn_knobs().enc_x3_min);
    // small maps (csrc/diinn_conv_t16.hip): the direct sum in units of (1 .. 3 rows x 16 pixels, 16 outputs) where the split-K
    // kernel's units leave compute units idle; the same weight image (per trunk: 48x48 1.95 -> 1.85 ms, 40x40 1.92 -> 1.44, 32x32 1.87 -> 1.05)
    const bool t16 = !wino && !x3 && diinn_conv_t16_applies(B, H, W) != 0 &&
                     ((((size_t)sfe1_dev) | ((size_t)planes) | ((size_t)packed_dev)) & 15) == 0;   // (its 16-byte LDS-DMA pieces; else the split-K kernel, as before)
    const float* w = packed_dev;
    const float* wu = packed_wino_dev;
    const float* wu4 = packed_wino4_dev;
    const float* wx = packed_x3_dev;
    const float* bias = biases_dev;
    auto conv = [&](const float* in, long long in_bs, int cin, int taps, const float* res, long long res_bs,
                    float* o0, long long o0_bs, float* o1, long long o1_bs, int relu) {
        const int s = (x3 && taps == 9 && !o1)
            ? diinn_conv3x3_x3(stream, in, in_bs, cin, wx, bias, res, res_bs, o0, o0_bs, relu, B, H, W)
            : (wino4 && taps == 9 && !o1)
            ? diinn_conv_wino4_ws(stream, in, in_bs, cin, wu4, bias, res, res_bs, o0, o0_bs, relu, B, H, W, w4ws, w4ws_floats)
            : (wino && taps == 9 && !o1)
            ? diinn_conv_wino(stream, in, in_bs, cin, wu, bias, res, res_bs, o0, o0_bs, relu, B, H, W)
            : (t16 && taps == 9 && !o1)
            ? diinn_conv_t16(stream, in, in_bs, cin, w, bias, res, res_bs, o0, o0_bs, relu, B, H, W)
            : (t16 && taps == 1 && cin <= 640)
            ? diinn_conv1x1_t16(stream, in, in_bs, cin, w, bias, res, res_bs, o0, o0_bs, o1, o1_bs, relu, B, H, W)
            : diinn_conv_ksplit(stream, in, in_bs, cin, taps, w, bias, res, res_bs, o0, o0_bs, o1, o1_bs, relu, B, H, W);
        w += (size_t)64 * cin * taps;
        if (taps == 9 && wu) wu += (size_t)64 * cin * 16;
        if (taps == 9 && wu4) wu4 += (size_t)64 * cin * 36;
        if (taps == 9 && wx) wx += (size_t)64 * cin * 9;
        bias += 64;
        return s;
    };
    float* xs = tmp + (size_t)B * 64 * hw;                                       // x3 only: the block's channels in split format
    const long long xs_bs16 = 144 * hw;                                          // 72 groups x (hi, lo) planes of 16-byte pixels
    size_t n3 = (size_t)2 * 64 * 64 * 9;
    for (int c = 0; c < 8; ++c) n3 += (size_t)16 * 64 * (64 + 64 * c) * 9;
    const float* wx1 = packed_x3_dev ? packed_x3_dev + n3 : nullptr;             // the fusion layers' split-bf16 weights
    // SFENet2: 64 -> 64 into channels [0,64) of the first dense buffer (rdn.py:97); x3: also groups 0..7 of the split buffer
    if (x3) {
        st = diinn_conv3x3_x3_split(stream, sfe1_dev, 64 * hw, xs, xs_bs16, 0, 64, wx, bias, buf[0], 576 * hw, 0, B, H, W);
        w += (size_t)64 * 64 * 9; wu += (size_t)64 * 64 * 16; wx += (size_t)64 * 64 * 9; bias += 64;
        if (wu4) wu4 += (size_t)64 * 64 * 36;
    } else {
        st = conv(sfe1_dev, 64 * hw, 64, 9, nullptr, 0, buf[0], 576 * hw, nullptr, 0, 0);
    }
    if (st) return st;
    for (int d = 0; d < 16; ++d) {
        float* cur = buf[d & 1];
        float* nxt = buf[(d + 1) & 1];
        if (x3) {
            // inside the trunk the layers exchange their activations already split (csrc/diinn_conv_x3.hip): every dense conv
            // reads ALL its inputs from the split buffer (two 16-byte copies per staged pixel instead of eight loads and a
            // conversion, redone by each of the up to eight layers that read a channel) and appends its outputs to it --
            // ONLY to it: the fp32 planes of the dense channels have no reader left, which halves the store burst at the end
            // of every layer (at 256x256 the epilogue's 33 MB were 10 of a layer's 23-92 us); the fusion layer reads the
            // split buffer whole and writes the next block's first 64 channels back in place, next to the planes
            for (int c = 0; c < 8; ++c) {
                st = diinn_conv3x3_x3_split(stream, nullptr, 0, xs, xs_bs16, 8 * (c + 1), 64 * (c + 1), wx, bias,
                                            nullptr, 0, 1, B, H, W);
                if (st) return st;
                w += (size_t)64 * 64 * (c + 1) * 9;
                wu += (size_t)64 * 64 * (c + 1) * 16;
                if (wu4) wu4 += (size_t)64 * 64 * (c + 1) * 36;
                wx += (size_t)64 * 64 * (c + 1) * 9;
                bias += 64;
            }
            st = diinn_conv1x1_x3_split(stream, xs, xs_bs16, 576, wx1 + (size_t)d * 64 * 576, bias, cur, 576 * hw, nxt, 576 * hw,
                                        gff_in + (size_t)64 * d * hw, 1024 * hw, B, H, W);
            if (st) return st;
            w += (size_t)64 * 576;
            bias += 64;
            continue;
        } else
        for (int c = 0; c < 8; ++c) {                            // dense 3x3 convs: read channels [0, 64(c+1)), append 64 (rdn.py:15-17)
            st = conv(cur, 576 * hw, 64 * (c + 1), 9, nullptr, 0, cur + (size_t)64 * (c + 1) * hw, 576 * hw, nullptr, 0, 1);
            if (st) return st;
        }
        // LFF 1x1 576 -> 64 plus the block input (rdn.py:34): next block's input and the d-th slice of the global fusion input
        st = conv(cur, 576 * hw, 576, 1, cur, 576 * hw, nxt, 576 * hw, gff_in + (size_t)64 * d * hw, 1024 * hw, 0);
        if (st) return st;
    }
    // GFF: 1x1 1024 -> 64, then 3x3 64 -> 64, plus the shallow features (rdn.py:100-103)
    st = conv(gff_in, 1024 * hw, 1024, 1, nullptr, 0, tmp, 64 * hw, nullptr, 0, 0);
    if (st) return st;
    return conv(tmp, 64 * hw, 64, 9, sfe1_dev, 64 * hw, out_dev, 64 * hw, nullptr, 0, 0);
}

// ---- ONE trunk entry point (ABI v9): which kernel family the 3x3 layers may take is an argument, the images it may read are
// the ones given.  AUTO = the fastest fp32 form the given images allow (the rules above); DIRECT / WINO / WINO4 cap the family
// (the result of the four former entry points); X3 = split bf16 on large maps, asked for explicitly (optional arithmetic).
static int rdn_forward_algo(void* stream, int algo, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                            const float* packed_wino4_dev, const float* packed_x3_dev, const float* biases_dev,
                            float* planes_dev, float* w4ws_dev, float* out_dev, int B, int H, int W, bool zero_status) {
    switch (algo) {
        case DIINN_RDN_ALGO_AUTO: packed_x3_dev = nullptr; break;
        case DIINN_RDN_ALGO_DIRECT: packed_wino_dev = packed_wino4_dev = packed_x3_dev = nullptr; break;
        case DIINN_RDN_ALGO_WINO:
            if (!packed_wino_dev) return DIINN_ERR_INVALID_ARG;
            packed_wino4_dev = packed_x3_dev = nullptr;
            break;
        case DIINN_RDN_ALGO_WINO4:
            // packed_wino_dev may be null when diinn_rdn_wino4_applies(B, H, W): every 3x3 layer then runs F(4x4) and the F(2x2)
            // image is never read (a caller that only sees such maps need not build or keep it: 152 MB)
            if (!packed_wino4_dev) return DIINN_ERR_INVALID_ARG;
            packed_x3_dev = nullptr;
            break;
        case DIINN_RDN_ALGO_X3:
            if (!packed_wino_dev || !packed_x3_dev) return DIINN_ERR_INVALID_ARG;
            packed_wino4_dev = nullptr;
            break;
        default: return DIINN_ERR_INVALID_ARG;
    }
    if ((((size_t)planes_dev) & 15) || (((size_t)w4ws_dev) & 15)) return DIINN_ERR_INVALID_ARG;
    return rdn_forward_impl(stream, sfe1_dev, packed_dev, packed_wino_dev, packed_wino4_dev, packed_x3_dev, biases_dev, planes_dev,
                            w4ws_dev, out_dev, B, H, W, zero_status);
}

int diinn_rdn_forward_ex(void* stream, int algo, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                         const float* packed_wino4_dev, const float* packed_x3_dev, const float* biases_dev,
                         float* planes_dev, float* w4ws_dev, float* out_dev, int B, int H, int W) {
    return rdn_forward_algo(stream, algo, sfe1_dev, packed_dev, packed_wino_dev, packed_wino4_dev, packed_x3_dev, biases_dev, planes_dev,
                            w4ws_dev, out_dev, B, H, W, false);
}

// ---- the four one-algorithm entry points of ABI <= 8: thin wrappers, to be dropped with the next ABI number.  Their single
// workspace is [F(4x4) split area][planes] (diinn_rdn_workspace_floats / diinn_rdn_x3_workspace_floats).
static int rdn_forward_v8(void* stream, int algo, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                          const float* packed_wino4_dev, const float* packed_x3_dev, const float* biases_dev,
                          float* workspace_dev, float* out_dev, int B, int H, int W) {
    if (!workspace_dev) return DIINN_ERR_INVALID_ARG;
    // ABI <= 8 promised "the first 4 KiB are zeroed by the forward itself" (callers hand in uninitialised workspaces): all control
    // words, the status word included -- so these entry points have no sticky status (a give-up is still NaN in that forward's output)
    return rdn_forward_algo(stream, algo, sfe1_dev, packed_dev, packed_wino_dev, packed_wino4_dev, packed_x3_dev, biases_dev,
                            workspace_dev + diinn_conv_wino4_workspace_floats(), workspace_dev, out_dev, B, H, W, true);
}

int diinn_rdn_forward(void* stream, const float* sfe1_dev, const float* packed_dev, const float* biases_dev,
                      float* workspace_dev, float* out_dev, int B, int H, int W) {
    return rdn_forward_v8(stream, DIINN_RDN_ALGO_DIRECT, sfe1_dev, packed_dev, nullptr, nullptr, nullptr, biases_dev, workspace_dev, out_dev, B, H, W);
}

int diinn_rdn_forward_wino(void* stream, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                           const float* biases_dev, float* workspace_dev, float* out_dev, int B, int H, int W) {
    return rdn_forward_v8(stream, DIINN_RDN_ALGO_WINO, sfe1_dev, packed_dev, packed_wino_dev, nullptr, nullptr, biases_dev, workspace_dev, out_dev, B, H, W);
}

int diinn_rdn_forward_wino4(void* stream, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                            const float* packed_wino4_dev, const float* biases_dev, float* workspace_dev, float* out_dev,
                            int B, int H, int W) {
    return rdn_forward_v8(stream, DIINN_RDN_ALGO_WINO4, sfe1_dev, packed_dev, packed_wino_dev, packed_wino4_dev, nullptr, biases_dev, workspace_dev,
                          out_dev, B, H, W);
}

int diinn_rdn_forward_x3(void* stream, const float* sfe1_dev, const float* packed_dev, const float* packed_wino_dev,
                         const float* packed_x3_dev, const float* biases_dev, float* workspace_dev, float* out_dev,
                         int B, int H, int W) {
    return rdn_forward_v8(stream, DIINN_RDN_ALGO_X3, sfe1_dev, packed_dev, packed_wino_dev, nullptr, packed_x3_dev, biases_dev, workspace_dev, out_dev, B, H, W);
}

}  // extern "C"
