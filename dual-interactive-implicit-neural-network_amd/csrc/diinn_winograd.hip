// diinn_winograd.hip -- the 3x3 convolutions of the RDN trunk as Winograd F(2x2, 3x3) on the fp32 MFMA (gfx950).
// (part of libdiinn_hip.so; shared definitions in diinn_device.h)
//
// Reference: src/models/components/rdn.py:9-35,90-105 -- 130 of the trunk's 147 convolutions are 3x3, stride 1,
// zero padding 1, 64 outputs, 64..512 inputs: 93 % of the encoder's arithmetic.  F(2x2, 3x3) computes a 2x2 output
// block from a 4x4 input patch with 16 multiplies per (input, output) channel pair instead of 36:
//     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A
// so the 64 x Cin reduction becomes 16 independent GEMMs (one per position of the transformed 4x4 tile) of 2.25x
// fewer MFMAs in total; only additions are added (B^T, A^T hold 0 / +-1), and G's halves are folded into the packed
// weight on the host.  fp32 throughout; the result differs from the direct sum by reassociation only (~1e-6
// relative; the library convolutions PyTorch calls use the same algorithm for fp32 3x3 layers).
//
// Work split: a workgroup owns a block of 8 x 4 Winograd tiles (16 x 8 output pixels) = one 32-wide MFMA N-tile;
// wave i (of 4, one per SIMD, the whole register file each) owns ROW i of the transformed tile: positions (i, 0..3),
// both 32-output halves -> 8 accumulators.  Per pair of input channels a wave loads two rows of every tile's 4x4
// patch straight from the feature planes (one 16-byte load per row; L1 serves the overlap between tiles and
// between waves), combines them (row i of B^T d: one fma per element), applies the column transform (4 adds) and
// feeds the 4 values as B operands to 8 MFMAs.  No LDS, no barrier, no cross-wave reduction until the end, where
// the four rows meet once through LDS for A^T (.) A, bias, ReLU, residual and the store.
// Operand streams: weights (8 KiB per wave and 8 channels) and patch rows are requested TWO chunks of 8 channels
// ahead into a three-stage register ring, one request after every second MFMA (a 1 KiB vector-memory instruction
// holds its wave's issue for ~60 cycles -- an MFMA's own length -- so they are spread, not batched), and the
// transform of k-step e+1 is issued between the MFMAs of k-step e.
#include "diinn_device.h"

constexpr int WN_TX = 8, WN_TY = 4;                  // Winograd tiles per block (x, y): 32 = one MFMA N-tile
constexpr int WN_PIECE_BYTES = 1024;                 // one A piece: 64 lanes x 4 k-steps
constexpr int WN_CHUNK_BYTES = 8 * WN_PIECE_BYTES;   // per wave and chunk of 8 channels: 4 positions x 2 halves
// WN_STAGES = 4 deepens the register ring (measured: no gain).  (The WN_ABL_* timing-ablation hooks behind
// profiles/r02_enc_*.txt were removed in round 5: git show 84fe2c6:<this file>.)
#ifndef WN_STAGES
#define WN_STAGES 3
#endif
constexpr int WN_NS = WN_STAGES;                     // register ring: chunks in flight + the one being computed
constexpr int WN_ZS_PITCH = 20;                      // a lane's 16 values + 4 floats of padding: 16-byte LDS accesses without bank conflicts
constexpr int WN_ZS_FLOATS = 4 * 2 * 2 * 64 * WN_ZS_PITCH;   // epilogue exchange: [row i][half][q][lane][acc reg]

struct ConvWinoParams {
    const float* in;         // input channel planes: in + b*in_bs + c*H*W
    const float* wu;         // packed transformed weight: [row i 4][chunk Cin/8][col j 4][half 2][lane 64][4], column 2 negated
    const float* bias;       // [64]
    const float* res;        // optional residual [B,64,H,W] (batch stride res_bs), added after bias (no ReLU with it)
    float* out;              // out + b*out_bs + co*H*W
    long long in_bs, out_bs, res_bs;
    int Cin, B, H, W, relu;
    unsigned m_per_image, m_bx_n;   // block / blocks-per-image and block / blocks-per-row as multiplications (div_m; 0: divide)
#ifdef DIINN_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/stamp_report_enc.py)
#endif
};


__device__ __forceinline__ f32x4 ld_row(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

#define WN_SB() __builtin_amdgcn_sched_barrier(0)

// NH = 2: both 32-output halves per workgroup (8 accumulators per wave, one workgroup per CU): big maps.
// NH = 1: one half (hh0) per workgroup: twice the workgroups at half the registers (two per CU, which cover each
//         other's prologue and epilogue) at the price of loading and transforming every patch row twice: maps whose
//         blocks alone do not fill the chip.
template <bool EDGE, int NH>
__device__ __forceinline__ void conv_wino_body(const ConvWinoParams& p, float* __restrict__ zs, int b, int tx0, int ty0, int hh0) {
    constexpr int NW = 4 * NH;                                   // weight pieces per chunk and wave
    constexpr int NREQ = NW + 8;                                 // + 8 patch rows
    constexpr int RPK = NH + 2;                                  // requests per k-step (4 k-steps per chunk)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = row i of the transformed tile
    const int h = lane >> 5, m = lane & 31;
    const int tx = tx0 + (m & (WN_TX - 1)), ty = ty0 + m / WN_TX;
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_b = (unsigned)(plane * sizeof(float));
    const float* __restrict__ in_b = p.in + (size_t)b * p.in_bs;
    const int n = p.Cin / 8;                                     // chunks of 8 input channels
    const int lane_off = lane * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wu + (size_t)wave * n * (WN_CHUNK_BYTES / 4)), 0, n * WN_CHUNK_BYTES, 0x00020000);
    // rows of B^T:  i = 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3   ->   t = d[ra] + sgn d[rb]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 2 ? 1 : (wave == 1 ? 2 : 3));
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    // patch of tile (tx, ty): rows 2 ty - 1 .. 2 ty + 2, columns 2 tx - 1 .. 2 tx + 2.  Rows outside the map (and
    // tiles outside it) get an offset the descriptor's range check answers with zeros; column -1 is handled by
    // loading from column 0 and shifting, columns >= W by selects (EDGE blocks only).
    const bool tile_in = 2 * tx < p.W && 2 * ty < p.H;
    const bool left = EDGE && tx == 0;
    const bool ok2 = 2 * tx + 1 < p.W, ok3 = 2 * tx + 2 < p.W;
    const int xc = left ? 0 : 2 * tx - 1;
    const int ya = 2 * ty - 1 + ra, yb = 2 * ty - 1 + rb;
    constexpr unsigned OUTSIDE = 0x80000000u;
    const unsigned offa = (tile_in && ya >= 0 && ya < p.H) ? (unsigned)h * plane_b + (unsigned)(ya * p.W + xc) * 4u : OUTSIDE;
    const unsigned offb = (tile_in && yb >= 0 && yb < p.H) ? (unsigned)h * plane_b + (unsigned)(yb * p.W + xc) * 4u : OUTSIDE;

    f32x4 A[WN_NS][NW];      // weight pieces of the ring's chunks: [stage][col j * NH + half], components = k-steps
    f32x4 R[WN_NS][8];       // patch rows of the ring's chunks: [stage][2 e + {row a, row b}], components = patch columns
    f32x16 acc[4][NH];       // [col j][half]
    f32x2 V01, V23;          // the B operands of the upcoming k-step: the transformed patch row, columns 0 1 | -2 3
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][hh][r] = 0.0f;

    // the descriptor of a chunk's 8 planes, rebased per chunk: offsets inside stay small whatever Cin * H * W is
    auto chunk_rsrc = [&](int c) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(in_b + (size_t)8 * c * plane), 0, (int)(8u * plane_b), 0x00020000);
    };
    // request #IDX (0..NREQ-1) of chunk c into stage S: the weight pieces (col j, half), then 8 patch rows
    auto request = [&](auto S_, auto IDX_, int c, __amdgpu_buffer_rsrc_t irs) {
        constexpr int S = decltype(S_)::value, IDX = decltype(IDX_)::value;
        if constexpr (IDX < NW) {
            A[S][IDX] = ld_piece(wrs, lane_off, c * WN_CHUNK_BYTES + (NH == 2 ? IDX : 2 * IDX + hh0) * WN_PIECE_BYTES);
        } else if constexpr (IDX < NREQ) {
            R[S][IDX - NW] = ld_row(irs, ((IDX - NW) & 1) ? offb : offa, (unsigned)(2 * ((IDX - NW) >> 1)) * plane_b);
        }
    };
    // The transform of one channel's two patch rows, 4 packed-fp32 instructions (each VALU instruction between two
    // fp32 MFMAs costs ~3 cycles of MFMA issue):
    //   row i of B^T d:  t = d[ra] + sgn d[rb]                                        (2 v_pk_fma)
    //   (B^T d) B:       v0 = t0 - t2, v1 = t1 + t2 | -v2 = t1 - t2, v3 = t1 - t3      (v_pk_fma with (-1, 1), v_pk_add;
    //                    the sign of column 2 lives in the packed weight)
    // Border columns (EDGE blocks): the patch's column -1 is padding (the row was loaded from column 0 and is shifted
    // here), columns >= W are padding.
    f32x2 pm = {-1.0f, 1.0f};
    asm volatile("" : "+v"(pm));                                 // a register pair: (-1, 1) is no inline constant
    const f32x2 sgn2 = {sgn, sgn};
    auto transform = [&](const f32x4& a, const f32x4& bb, f32x2& v01, f32x2& v23) {
        f32x2 t01 = __builtin_elementwise_fma(sgn2, f32x2{bb[0], bb[1]}, f32x2{a[0], a[1]});
        f32x2 t23 = __builtin_elementwise_fma(sgn2, f32x2{bb[2], bb[3]}, f32x2{a[2], a[3]});
        if constexpr (EDGE) {
            const float d0 = left ? 0.0f : t01[0];
            const float d1 = left ? t01[0] : t01[1];
            float d2 = left ? t01[1] : t23[0];
            float d3 = left ? t23[0] : t23[1];
            d2 = ok2 ? d2 : 0.0f;
            d3 = ok3 ? d3 : 0.0f;
            t01 = f32x2{d0, d1};
            t23 = f32x2{d2, d3};
        }
        v01 = __builtin_elementwise_fma(f32x2{t23[0], t23[0]}, pm, t01);
        v23 = f32x2{t01[1], t01[1]} - t23;
    };
    // One k-step (a pair of input channels) of the chunk in stage S: 8 MFMAs; between them the B operands of the next
    // k-step are prepared (the chunk's last k-step prepares the next chunk's first, from stage S+1) and four of the
    // sixteen requests of chunk cload go out into stage S+2.
    auto kstep = [&](auto S_, auto E_, int cload, __amdgpu_buffer_rsrc_t irs) {
        constexpr int S = decltype(S_)::value, E = decltype(E_)::value;
        constexpr int SL = (S + WN_NS - 1) % WN_NS, SN = (S + 1) % WN_NS;
        const f32x2 c01 = V01, c23 = V23;
        const f32x4& ra4 = E < 3 ? R[S][2 * ((E + 1) & 3)] : R[SN][0];
        const f32x4& rb4 = E < 3 ? R[S][2 * ((E + 1) & 3) + 1] : R[SN][1];
        const float cj[4] = {c01[0], c01[1], c23[0], c23[1]};
#pragma unroll
        for (int q = 0; q < NW; ++q) {                           // MFMA q: col q / NH, half q % NH
            acc[q / NH][q % NH] = MFMA32(A[S][q][E], cj[q / NH], acc[q / NH][q % NH]);
            if (q == 0) {
                transform(ra4, rb4, V01, V23);
            }
            // RPK requests per k-step: after every second MFMA of 8, or after each of the first three of 4
            if constexpr (NH == 2) {
                if (q == 1) request(IC<SL>{}, IC<RPK * E + 0>{}, cload, irs);
                if (q == 3) request(IC<SL>{}, IC<RPK * E + 1>{}, cload, irs);
                if (q == 5) request(IC<SL>{}, IC<RPK * E + 2>{}, cload, irs);
                if (q == 7) request(IC<SL>{}, IC<RPK * E + 3>{}, cload, irs);
            } else {
                if (q == 0) request(IC<SL>{}, IC<RPK * E + 0>{}, cload, irs);
                if (q == 1) request(IC<SL>{}, IC<RPK * E + 1>{}, cload, irs);
                if (q == 2) request(IC<SL>{}, IC<RPK * E + 2>{}, cload, irs);
            }
            WN_SB();
        }
    };
    auto chunk = [&](auto S_, int cload) {
        const __amdgpu_buffer_rsrc_t irs = chunk_rsrc(cload);
        kstep(S_, IC<0>{}, cload, irs);
        kstep(S_, IC<1>{}, cload, irs);
        kstep(S_, IC<2>{}, cload, irs);
        kstep(S_, IC<3>{}, cload, irs);
    };
    auto request_all = [&](auto S_, int c) {
        const __amdgpu_buffer_rsrc_t irs = chunk_rsrc(c);
        request(S_, IC<0>{}, c, irs);  request(S_, IC<1>{}, c, irs);  request(S_, IC<2>{}, c, irs);  request(S_, IC<3>{}, c, irs);
        request(S_, IC<4>{}, c, irs);  request(S_, IC<5>{}, c, irs);  request(S_, IC<6>{}, c, irs);  request(S_, IC<7>{}, c, irs);
        request(S_, IC<8>{}, c, irs);  request(S_, IC<9>{}, c, irs);  request(S_, IC<10>{}, c, irs); request(S_, IC<11>{}, c, irs);
        request(S_, IC<12>{}, c, irs); request(S_, IC<13>{}, c, irs); request(S_, IC<14>{}, c, irs); request(S_, IC<15>{}, c, irs);
    };

#ifdef DIINN_STAMPS
    const size_t stamp_base = ((size_t)blockIdx.x * 4 + wave) * 8;
#endif
    STAMP(0);
    request_all(IC<0>{}, 0);
    request_all(IC<1>{}, n > 1 ? 1 : 0);
    if constexpr (WN_NS > 3) request_all(IC<2>{}, n > 2 ? 2 : n - 1);
    WN_SB();
    transform(R[0][0], R[0][1], V01, V23);
    WN_SB();
    STAMP(1);
    // chunk c sits in stage c % NS; while it is computed chunk c + NS - 1 is requested (past the end: the last chunk
    // again, never used)
    auto ahead = [&](int cc) { return cc + WN_NS - 1 < n ? cc + WN_NS - 1 : n - 1; };
    int c = 0;
    for (; c + WN_NS <= n; c += WN_NS) {
        chunk(IC<0>{}, ahead(c));
        chunk(IC<1>{}, ahead(c + 1));
        chunk(IC<2>{}, ahead(c + 2));
        if constexpr (WN_NS > 3) chunk(IC<3>{}, ahead(c + 3));
    }
    if (c < n) chunk(IC<0>{}, n - 1);
    if (c + 1 < n) chunk(IC<1>{}, n - 1);
    if constexpr (WN_NS > 3) {
        if (c + 2 < n) chunk(IC<2>{}, n - 1);
    }

    STAMP(2);
    // ---- A^T (.) A: the column half in registers, the row half through LDS (the only barrier of the kernel).
    // A lane's 16 values of one (row, half, q) are contiguous (pitch 20 floats): 16-byte LDS accesses, no conflicts.
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 z0, z1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const float m0 = acc[0][hh][r], m1 = acc[1][hh][r], m2 = acc[2][hh][r], m3 = acc[3][hh][r];
                z0[e] = (m0 + m1) + m2;
                z1[e] = (m1 - m2) - m3;
            }
            *reinterpret_cast<f32x4*>(zs + ((((wave * NH + hh) * 2 + 0) * 64 + lane) * WN_ZS_PITCH + 4 * g)) = z0;
            *reinterpret_cast<f32x4*>(zs + ((((wave * NH + hh) * 2 + 1) * 64 + lane) * WN_ZS_PITCH + 4 * g)) = z1;
        }
    __syncthreads();
    STAMP(3);
    // wave: output row pr of the 2x2 block and (NH = 2) output half wave >> 1, all 16 accumulator registers, or (NH = 1)
    // registers 8 (wave >> 1) .. + 7 of the workgroup's half; a lane: both columns of its tile
    // (co = 32 hh + 8 g + e + 4 h for accumulator register 4 g + e)
    const int pr = wave & 1;
    const int hl = NH == 2 ? wave >> 1 : 0;                      // the half's index in the exchange buffer
    const int hh = NH == 2 ? wave >> 1 : hh0;                    // .. and among the output channels
    const int g0 = NH == 2 ? 0 : 2 * (wave >> 1);
    const float s2 = pr == 0 ? 1.0f : -1.0f;                    // row 0: Z0 + Z1 + Z2, row 1: Z1 - Z2 - Z3
    const int oy = 2 * ty + pr, ox = 2 * tx;
    const bool in0 = oy < p.H && ox < p.W, in1 = oy < p.H && ox + 1 < p.W;
    const bool pair = (p.W & 1) == 0 && (p.out_bs & 1) == 0 && (((size_t)p.out) & 7) == 0;   // 8-byte stores stay aligned
    // stores through a descriptor over this batch element's 64 planes: lane offset = plane 4 h + pixel, scalar offset =
    // plane 32 hh + 8 g + e; pixels outside the map get an offset the range check drops
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)b * p.out_bs), 0,
                                                                         (int)(64u * plane_b), 0x00020000);
    const unsigned pix_b = (unsigned)(4 * h) * plane_b + (unsigned)(oy * p.W + ox) * 4u;
    const unsigned st0 = in0 ? pix_b : OUTSIDE, st1 = in1 ? pix_b + 4u : OUTSIDE;
#pragma unroll
    for (int gi = 0; gi < 2 * NH; ++gi) {
        const int g = g0 + gi;
        f32x4 y[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 z0 = *reinterpret_cast<const f32x4*>(zs + (((((pr + 0) * NH + hl) * 2 + q) * 64 + lane) * WN_ZS_PITCH + 4 * g));
            const f32x4 z1 = *reinterpret_cast<const f32x4*>(zs + (((((pr + 1) * NH + hl) * 2 + q) * 64 + lane) * WN_ZS_PITCH + 4 * g));
            const f32x4 z2 = *reinterpret_cast<const f32x4*>(zs + (((((pr + 2) * NH + hl) * 2 + q) * 64 + lane) * WN_ZS_PITCH + 4 * g));
#pragma unroll
            for (int e = 0; e < 4; ++e) y[q][e] = __builtin_fmaf(s2, z2[e], __builtin_fmaf(s2, z1[e], z0[e]));
        }
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + 32 * hh + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y0 = y[0][e] + bias4[e], y1 = y[1][e] + bias4[e];
            if (p.relu) y0 = relu0(y0), y1 = relu0(y1);
            const int cs = 32 * hh + 8 * g + e;                  // the scalar part of the output channel
            if (p.res) {
                const float* __restrict__ rp = p.res + (size_t)b * p.res_bs + (size_t)(cs + 4 * h) * plane + (size_t)oy * p.W + ox;
                if (in0) y0 += rp[0];
                if (in1) y1 += rp[1];
            }
            if (pair) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{y0, y1}), ors, (int)st0, (int)((unsigned)cs * plane_b), 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0), ors, (int)st0, (int)((unsigned)cs * plane_b), 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1), ors, (int)st1, (int)((unsigned)cs * plane_b), 0);
            }
        }
    }
    STAMP(4);
}

template <int NH>
__device__ __forceinline__ void conv_wino_entry(const ConvWinoParams& p, float* zs) {
    const int tiles_x = (p.W + 1) / 2, tiles_y = (p.H + 1) / 2;
    const int bx_n = (tiles_x + WN_TX - 1) / WN_TX, by_n = (tiles_y + WN_TY - 1) / WN_TY;
    const int total = p.B * bx_n * by_n * (3 - NH);
    // Every XCD takes a contiguous run of work items (blocks b and b + 8 share an L2: neighbours share patch rows).  The
    // grid may be smaller than the work (persistent workgroups: conv_wino_kernel on big maps): round k of an XCD's
    // workgroups covers the next gridDim.x / 8 items of its run.
    const int wg_per_xcd = gridDim.x >> 3;
    const int per_xcd = (total + 7) >> 3;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    for (int k = idx; k < per_xcd; k += wg_per_xcd) {
        int t = xcd * per_xcd + k;
        if (t >= total) break;
        int hh0 = 0;
        if constexpr (NH == 1) {                                 // the two halves of a block are neighbours: they share its patch rows
            hh0 = t & 1;
            t >>= 1;
        }
        // (integer division runs on the vector unit: pin the results back into scalar registers, or every descriptor
        // derived from them is treated as divergent and each load becomes a waterfall loop)
        const int b = __builtin_amdgcn_readfirstlane(div_m(t, bx_n * by_n, p.m_per_image));
        t -= b * bx_n * by_n;
        const int by = __builtin_amdgcn_readfirstlane(div_m(t, bx_n, p.m_bx_n)), bx = t - by * bx_n;
        const int tx0 = bx * WN_TX, ty0 = by * WN_TY;
        const bool edge = tx0 == 0 || 2 * (tx0 + WN_TX - 1) + 2 >= p.W;
        if (edge) conv_wino_body<true, NH>(p, zs, b, tx0, ty0, hh0);
        else      conv_wino_body<false, NH>(p, zs, b, tx0, ty0, hh0);
        __syncthreads();                                         // the exchange buffer is free again
    }
}

__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvWinoParams p) {
    __shared__ __attribute__((aligned(16))) float zs[WN_ZS_FLOATS];
    conv_wino_entry<2>(p, zs);
}

__global__ __launch_bounds__(256, 2) void conv_wino_half_kernel(const ConvWinoParams p) {
    __shared__ __attribute__((aligned(16))) float zs[WN_ZS_FLOATS / 2];
    conv_wino_entry<1>(p, zs);
}

extern "C" {

int diinn_conv_wino(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                    const float* packed_u_dev, const float* bias_dev, const float* res_dev, long long res_batch_stride,
                    float* out_dev, long long out_batch_stride, int relu, int B, int H, int W) {
    if (!in_dev || !packed_u_dev || !bias_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Cin <= 0 || Cin % 8) return DIINN_ERR_UNSUPPORTED;
    if ((((size_t)in_dev) & 3) || (((size_t)packed_u_dev) & 15) || (((size_t)bias_dev) & 15)) return DIINN_ERR_INVALID_ARG;
    const long long blocks = (long long)(((W + 1) / 2 + WN_TX - 1) / WN_TX) * (((H + 1) / 2 + WN_TY - 1) / WN_TY) * B;
    if (blocks > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    if ((long long)H * W * 4 * 64 > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;     // the 64 output planes (and a chunk of 8 input planes) are addressed with 32-bit byte offsets
    if ((long long)Cin * WN_CHUNK_BYTES / 8 * 4 > 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    ConvWinoParams p;
    p.in = in_dev; p.wu = packed_u_dev; p.bias = bias_dev; p.res = res_dev; p.out = out_dev;
    p.in_bs = in_batch_stride; p.out_bs = out_batch_stride; p.res_bs = res_batch_stride;
    p.Cin = Cin; p.B = B; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
    {
        const long long bx_n = ((W + 1) / 2 + WN_TX - 1) / WN_TX, per_image = bx_n * (((H + 1) / 2 + WN_TY - 1) / WN_TY);
        p.m_per_image = magic_m(per_image, blocks);
        p.m_bx_n = magic_m(bx_n, per_image);
    }
#ifdef DIINN_STAMPS
    p.stamps = g_stamps;
#endif
    // Both halves per workgroup (one per CU at a time) or one half (two per CU, 0.57x the work each, every patch row
    // loaded and transformed twice)?  Whichever gives the busiest CU less to do: a CU ends up with ceil(b / 256) whole
    // blocks or ceil(2 b / 256) halves (measured on 176 .. 320-pixel maps, tools/r02_ab_env.sh).
    // DIINN_ENC_WINO_HALF_MAX = n forces halves below n blocks and whole blocks from there on.
    const long long half_max = knob(diinn_knobs().enc_wino_half_max);
    const double cost_whole = (double)((blocks + 255) / 256), cost_half = 0.57 * (double)((2 * blocks + 255) / 256);
    const bool halves = half_max >= 0 ? blocks < half_max : cost_half < cost_whole;
    if (halves)
        hipLaunchKernelGGL(conv_wino_half_kernel, dim3((unsigned)((2 * blocks + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream, p);
    else {
        // persistent workgroups, one per CU: the second and later blocks of a workgroup start without a dispatch
        const long long persist = knob(diinn_knobs().enc_wino_persist);
        const long long grid = persist > 0 && blocks > persist ? persist : (blocks + 7) / 8 * 8;
        hipLaunchKernelGGL(conv_wino_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
    }
    return hip_status(hipGetLastError());
}

}  // extern "C"
