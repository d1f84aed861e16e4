// diinn_training.hip -- backward pass of the decoder (training): per-pixel chain, weight-gradient GEMMs, per-cell sums
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
#include "diinn_device.h"

// ---------------------------------------------------------------------------------
// backward pass of the per-pixel layers (training; reference: autograd through step(), diinn.py:132-139)
//
// With q_i = k_i * sin(s_i) and the planes k_i, s_i saved by decode_kernel<SAVE>:
//     g_a,i = g_q,i * sin(s_i) * [k_i > 0]      (gradient at the modulation pre-activation)
//     g_s,i = g_q,i * k_i * cos(s_i)            (gradient at the sine argument)
//     g_q,i-1 = Wq_i^T g_a,i + Qw_i^T g_s,i     (stacked [256 x 512] GEMM per pixel)
// bwd_head_kernel  : g_q,3 = L^T g_out, gates of layer 3 (elementwise, HBM-bound).
// bwd_layer_kernel : one launch per layer i = 3, 2, 1.  A wave owns one plane tile (32 pixels), loads
//                    their 512 gate gradients G_i = (g_a,i ; g_s,i) into registers as the MFMA B operand
//                    (the rows are read in accumulator order, so no shuffle is needed), streams the
//                    transposed weights (WLT section) exactly like the forward kernel streams WL, and
//                    its epilogue applies the gates of layer i-1 and writes G_{i-1} and q_{i-1}.
// plane_gemm_lds_kernel / plane_rowdot_kernel : the parameter gradients, GEMMs over the pixel axis of the
//                    planes written here (dW_i = G_i q_{i-1}^T ...).
// All planes are tiled (see PLANE_TILE above): acts, G [4][ntiles][512][32]; Q [4][ntiles][256][32].
// ---------------------------------------------------------------------------------
// Cache policy of the training planes (round 6, A/B in ABBA order on one box: step 14.44 -> 14.31 ms): they are written once and read once
// or twice by LATER kernels, gigabytes at a time -- streaming (nt) loads in bwd_layer_kernel, plane_rowdot_kernel and cell_sum_kernel
// like the nt stores of st_act leave the L2 to the weight stream.  (nt on the plane GEMM's LDS-DMA pieces as well: no further change.)
#ifndef PLANE_LD_AUX
#define PLANE_LD_AUX 2
#endif
#ifndef PG_A_AUX
#define PG_A_AUX 0
#endif
#ifndef PG_B_AUX
#define PG_B_AUX 0
#endif
#ifndef STREAM_LD_NT
#define STREAM_LD_NT 1
#endif
__device__ __forceinline__ float ld_plane(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, PLANE_LD_AUX));
}
struct BwdParams {
    const float* Wt;         // packed image
    const float* acts;       // k_i (rows 0..255), s_i (rows 256..511)
    const float* gout;       // [3][npix] plain planes: d loss / d out
    float* G;                // g_a,i (rows 0..255), g_s,i (rows 256..511)
    float* Q;                // q_i
    long long npix, ntiles;
    int layer;               // bwd_layer_kernel: consumes G_layer, produces G_{layer-1}, Q_{layer-1}
};

__device__ __forceinline__ void dsincos(float x, float& sn, float& cs) {
    constexpr float C_HI = 0.15915494309189533577f;
    constexpr float C_LO = 6.4206383650924e-09f;
    const float k = __builtin_rintf(x * C_HI);
    float r = __builtin_fmaf(x, C_HI, -k);
    r = __builtin_fmaf(x, C_LO, r);
    sn = __builtin_amdgcn_sinf(r);
    cs = __builtin_amdgcn_cosf(r);
}


__global__ __launch_bounds__(256) void bwd_head_kernel(const BwdParams p) {
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= p.npix) return;
    const size_t np = (size_t)p.npix;
    const float g0 = p.gout[pix], g1 = p.gout[np + pix], g2 = p.gout[2 * np + pix];
    const float* __restrict__ L = p.Wt + OFF_L;
    const size_t tile = (size_t)(pix >> 5), lane = (size_t)(pix & 31);
    const size_t a0 = ((size_t)3 * p.ntiles + tile) * ACT_ROWS * PLANE_TILE + lane;   // layer 3 tile, row 0
    const size_t q0 = ((size_t)3 * p.ntiles + tile) * HID * PLANE_TILE + lane;
    const int c0 = blockIdx.y * 16;
#pragma unroll 4
    for (int c = c0; c < c0 + 16; ++c) {
        float g = L[c] * g0;
        g = __builtin_fmaf(L[HID + c], g1, g);
        g = __builtin_fmaf(L[2 * HID + c], g2, g);
        const float kv = p.acts[a0 + (size_t)c * PLANE_TILE];
        const float sv = p.acts[a0 + (size_t)(HID + c) * PLANE_TILE];
        float sn, cs;
        dsincos(sv, sn, cs);
        p.G[a0 + (size_t)c * PLANE_TILE] = kv > 0.0f ? g * sn : 0.0f;
        p.G[a0 + (size_t)(HID + c) * PLANE_TILE] = g * kv * cs;
        p.Q[q0 + (size_t)c * PLANE_TILE] = kv * sn;
    }
}

// HEAD (round 6; layer 3 only): the wave computes its own B operand -- g_q,3 = L^T g_out and the gates of layer 3 from the saved
// k_3, s_3 -- in place of bwd_head_kernel, which wrote G_3 and q_3 (1.8 GB at B = 16, 192 x 192) only for this kernel to read G_3
// back (1.2 GB): the loads that fetched G_3's groups fetch (k_3, s_3) instead, a group is turned into (g_a,3 ; g_s,3) right before
// its first MFMAs and stored from there (G_3 and q_3 are still needed: the weight-gradient GEMM, the cell sums, dL).  Same
// formulas in the same order as bwd_head_kernel: bit-identical planes.
template <bool HEAD>
__global__ __launch_bounds__(256, 1) void bwd_layer_kernel(const BwdParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= p.ntiles) return;                                // wave-uniform
    const bool valid = tile * PLANE_TILE + j < p.npix;

    const int li = p.layer;                                     // 1..3
    const size_t agroup = (size_t)p.ntiles * ACT_ROWS * PLANE_TILE;     // floats per layer of acts / G
    const size_t qgroup = (size_t)p.ntiles * HID * PLANE_TILE;
    // lanes past the end: offset outside the descriptor, loads return 0 and stores are dropped
    const unsigned voff = valid ? 4u * j + 4u * h * PLANE_ROW_BYTES : 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t inG = tile_rsrc(p.G + (size_t)li * agroup, tile, ACT_ROWS);
    const __amdgpu_buffer_rsrc_t act = tile_rsrc(p.acts + (size_t)(li - 1) * agroup, tile, ACT_ROWS);
    const __amdgpu_buffer_rsrc_t outG = tile_rsrc(p.G + (size_t)(li - 1) * agroup, tile, ACT_ROWS);
    const __amdgpu_buffer_rsrc_t outQ = tile_rsrc(p.Q + (size_t)(li - 1) * qgroup, tile, HID);

    // B operand: register kk = 16m + r of lane-half h holds channel chan_of(kk, h) of this lane's pixel.
    // Only the first BLD k-groups are fetched up front; the rest stream in BLD groups ahead of the
    // MFMAs of the first output tile (which walks all 32 k-groups), so the 64 KiB a wave reads
    // hide behind its own arithmetic instead of in front of it.
    constexpr int BLD = 8;
    float ga[128], gs[128];
    // HEAD: the saved planes of THIS layer (k in ga, s in gs until head_group turns them into the gate gradients)
    const __amdgpu_buffer_rsrc_t act_li = tile_rsrc(p.acts + (size_t)li * agroup, tile, ACT_ROWS);
    const __amdgpu_buffer_rsrc_t outG_li = tile_rsrc(p.G + (size_t)li * agroup, tile, ACT_ROWS);
    const __amdgpu_buffer_rsrc_t outQ_li = tile_rsrc(p.Q + (size_t)li * qgroup, tile, HID);
    auto load_group = [&](int kg) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = 4 * kg + e;
            const unsigned so = (unsigned)(32 * (kk >> 4) + (kk & 3) + 8 * ((kk & 15) >> 2)) * PLANE_ROW_BYTES;
            ga[kk] = ld_plane(HEAD ? act_li : inG, voff, so);
            gs[kk] = ld_plane(HEAD ? act_li : inG, voff, so + HID * PLANE_ROW_BYTES);
        }
    };
    float go0 = 0.0f, go1 = 0.0f, go2 = 0.0f;                    // HEAD: d loss / d out of this lane's pixel
    if constexpr (HEAD) {
        const long long pix = tile * PLANE_TILE + j;
        if (valid) {
            go0 = p.gout[pix];
            go1 = p.gout[(size_t)p.npix + pix];
            go2 = p.gout[2 * (size_t)p.npix + pix];
        }
    }
    // HEAD: (k, s) of group kg -> (g_a, g_s), stored with q (bwd_head_kernel's arithmetic, term for term)
    auto head_group = [&](int kg) {
        const float* __restrict__ L = p.Wt + OFF_L;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = 4 * kg + e;
            const int c0 = 32 * (kk >> 4) + (kk & 3) + 8 * ((kk & 15) >> 2);      // channel of lane-half 0; half 1: + 4
            const unsigned so = (unsigned)c0 * PLANE_ROW_BYTES;
            const float l0 = h ? L[c0 + 4] : L[c0], l1 = h ? L[HID + c0 + 4] : L[HID + c0], l2 = h ? L[2 * HID + c0 + 4] : L[2 * HID + c0];
            float g = l0 * go0;
            g = __builtin_fmaf(l1, go1, g);
            g = __builtin_fmaf(l2, go2, g);
            const float kv = ga[kk], sv = gs[kk];
            float sn, cs;
            dsincos(sv, sn, cs);
            ga[kk] = kv > 0.0f ? g * sn : 0.0f;
            gs[kk] = g * kv * cs;
            st_act(outQ_li, voff, so, kv * sn);                  // q_3 leaves now; (g_a,3 ; g_s,3) stay in registers as the B operand
        }                                                         // and are stored later, one k-group per 8 MFMAs of M-tiles 1..4
    };
    auto head_store_group = [&](int kg) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = 4 * kg + e;
            const unsigned so = (unsigned)(32 * (kk >> 4) + (kk & 3) + 8 * ((kk & 15) >> 2)) * PLANE_ROW_BYTES;
            st_act(outG_li, voff, so, ga[kk]);
            st_act(outG_li, voff, so + HID * PLANE_ROW_BYTES, gs[kk]);
        }
    };
#pragma unroll
    for (int kg = 0; kg < BLD; ++kg) load_group(kg);

    constexpr int PF = DECODE_PREFETCH;                          // (a ring of 8 or 16 steps: no change, 14.65-14.76 ms -- round 6)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wt, 0, (int)(PACKED_FLOATS * sizeof(float)), 0x00020000);   // reads past the end return 0
    const int lane_off = lane * 16;
    const int wp = (int)((OFF_WLT + (size_t)(li - 1) * WL_LAYER) * sizeof(float));
    f32x4 rk[PF], rq[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        rk[d] = ld_piece(wrs, lane_off, wp + (2 * d + 0) * PIECE_BYTES);
        rq[d] = ld_piece(wrs, lane_off, wp + (2 * d + 1) * PIECE_BYTES);
    }

    // gates of layer li-1 for one finished element: g = d loss / d q_{li-1}[channel, pixel]
    auto gate_store = [&](int mt, int r, float g, float kv, float sv) {
        const unsigned so = (unsigned)(32 * mt + (r & 3) + 8 * (r >> 2)) * PLANE_ROW_BYTES;
        float sn, cs;
        dsincos(sv, sn, cs);
        st_act(outG, voff, so, kv > 0.0f ? g * sn : 0.0f);
        st_act(outG, voff, so + HID * PLANE_ROW_BYTES, g * kv * cs);
        st_act(outQ, voff, so, kv * sn);
    };

    f32x16 pg;                                                   // finished g_q tile (sum of the two accumulators)
    float kt[16], st[16];                                        // saved k, s of the tile being finished
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 ak, as;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[r] = 0.0f; as[r] = 0.0f; }
#pragma unroll
        for (int kg = 0; kg < WL_KG; ++kg) {
            const int s = m * WL_KG + kg;
            const f32x4 wk = rk[s % PF];
            const f32x4 wq = rq[s % PF];
            if constexpr (HEAD) {
                if (m == 0) head_group(kg);                       // (its loads went out BLD groups = 64 MFMAs ago)
                if (m == 1) head_store_group(kg);                 // 8 stores behind 8 MFMAs (in M-tile 0 they would sit beside the
            }                                                     // group's 8 loads, ~76 VALU instructions and 4 q_3 stores)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ak = MFMA32(wk[e], ga[4 * kg + e], ak);
                as = MFMA32(wq[e], gs[4 * kg + e], as);
            }
            rk[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 0) * PIECE_BYTES);
            rq[s % PF] = ld_piece(wrs, lane_off, wp + (2 * (s + PF) + 1) * PIECE_BYTES);
            if (m == 0 && kg + BLD < WL_KG) load_group(kg + BLD); // rest of the B operand, BLD groups ahead
            if (m > 0 && kg == 0) {                               // saved planes of tile m-1, used from kg = 8 on
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned so = (unsigned)(32 * (m - 1) + (r & 3) + 8 * (r >> 2)) * PLANE_ROW_BYTES;
                    kt[r] = ld_plane(act, voff, so);
                    st[r] = ld_plane(act, voff, so + HID * PLANE_ROW_BYTES);
                }
            }
            if (m > 0 && kg >= 8 && kg < 24) {                    // one epilogue element of tile m-1 every 8 MFMAs
                const int r = kg - 8;
                gate_store(m - 1, r, pg[r], kt[r], st[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) pg[r] = ak[r] + as[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned so = (unsigned)(32 * 7 + (r & 3) + 8 * (r >> 2)) * PLANE_ROW_BYTES;
        kt[r] = ld_plane(act, voff, so);
        st[r] = ld_plane(act, voff, so + HID * PLANE_ROW_BYTES);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) gate_store(7, r, pg[r], kt[r], st[r]);
}

// ---------------------------------------------------------------------------------
// cell_sum_kernel (training backward): dP_i[b, ch, cy, cx] = sum of g_a,i over the HR pixels whose
// nearest LR cell is (cy, cx) -- the adjoint of the nearest-exact replication (diinn.py:168).  The
// index tables are monotone, so a cell's pixels are the rectangle [seg_h[cy], seg_h[cy+1]) x
// [seg_w[cx], seg_w[cx+1]).  One thread per output element, cx fastest: neighbouring lanes read
// neighbouring column segments of the same HR rows.  Fixed summation order (no atomics).  Output is
// NCHW [B][1024][H][W], channel = 256 i + ch: the layout the 3x3 convolution's weight/input
// gradients are taken in.  HBM-bound (reads the g_a rows of G once).
// ---------------------------------------------------------------------------------
// Round 6: one thread per (cell, plane) with the cells of an image FLATTENED over the threads (a 48-wide map left a quarter of
// a 64-lane row idle), and a cell's row segment fetched as ONE 16- or 8-byte load where it is 4 or 2 pixels wide and aligned
// (the integer scales x4 / x2 of the reference's training batches; the rows of a cell are independent loads in flight at
// once) -- 16 dependent 4-byte loads per thread before: 1.00 -> ~0.6 ms at B = 16, 48 x 48 -> 192 x 192 (2.4 GB read once).
// The summation order is unchanged (left to right inside a row, rows top to bottom): bit-identical results.
// dP_tiled (optional): the same values as a tiled plane group [ceil(B H W / 32)][1024][32] over the CELL axis -- the A operand
// of the hoisted conv's weight-gradient GEMM (diinn_plane_gemm_nt), written here instead of by a transposing copy.
// ---------------------------------------------------------------------------------
struct CellSumParams {
    const float* G;          // tiled [4][ntiles][512][32]; rows 0..255 (g_a) are summed
    float* dP;               // [B][1024][H][W]
    float* dP_tiled;         // optional: [ceil(B*H*W / 32)][1024][32]
    const int* seg_h;        // [H+1] first HR row of every LR row (seg_h[H] = Hu)
    const int* seg_w;        // [W+1]
    int B, H, W, Hu, Wu;
    long long ntiles;
};

__global__ __launch_bounds__(256) void cell_sum_kernel(const CellSumParams p) {
    const int cell = blockIdx.x * 256 + threadIdx.x;                 // cy * W + cx
    if (cell >= p.H * p.W) return;
    const int b = blockIdx.y;
    const int cy = cell / p.W, cx = cell - cy * p.W;
    const int plane = blockIdx.z;                                    // 256 i + ch
    const float* __restrict__ src = p.G + ((size_t)(plane >> 8) * p.ntiles * ACT_ROWS + (plane & 255)) * PLANE_TILE;
    const int y0 = p.seg_h[cy], y1 = p.seg_h[cy + 1];
    const int x0 = p.seg_w[cx], x1 = p.seg_w[cx + 1];
    const int wd = x1 - x0;
    auto at = [&](long long pix) { return src + (size_t)(pix >> 5) * (ACT_ROWS * PLANE_TILE) + (size_t)(pix & 31); };
    float acc = 0.0f;
    // (Wu % 4 == 0 and x0 % 4 == 0: the segment's first pixel is a multiple of 4 in the flattened index, so the 4 pixels lie in
    // one 32-pixel tile row, 16-byte aligned; likewise for 2)
    if (wd == 4 && ((p.Wu | x0) & 3) == 0) {
        for (int y = y0; y < y1; y += 4) {                       // up to four rows in flight
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (y + k < y1) v[k] = STREAM_LD_NT ? __builtin_nontemporal_load((const f32x4*)at(((long long)b * p.Hu + y + k) * p.Wu + x0))
                                                      : *(const f32x4*)at(((long long)b * p.Hu + y + k) * p.Wu + x0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (y + k < y1) acc += ((v[k][0] + v[k][1]) + v[k][2]) + v[k][3];
        }
    } else if (wd == 2 && ((p.Wu | x0) & 1) == 0) {
        for (int y = y0; y < y1; y += 4) {
            f32x2 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (y + k < y1) v[k] = *(const f32x2*)at(((long long)b * p.Hu + y + k) * p.Wu + x0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (y + k < y1) acc += v[k][0] + v[k][1];
        }
    } else {
        for (int y = y0; y < y1; ++y) {
            const long long rowpix = ((long long)b * p.Hu + y) * p.Wu;
            float r = 0.0f;
            for (int x = x0; x < x1; ++x) r += *at(rowpix + x);
            acc += r;
        }
    }
    p.dP[(((size_t)b * PCH + plane) * p.H + cy) * p.W + cx] = acc;
    if (p.dP_tiled) {
        const long long n = (long long)b * p.H * p.W + cell;
        p.dP_tiled[((size_t)(n >> 5) * PCH + plane) * PLANE_TILE + (size_t)(n & 31)] = acc;
    }
}

// ---------------------------------------------------------------------------------
// unfold_tiled_kernel (training backward, weight gradient of the hoisted 3x3 conv): the reference's F.unfold(feat, 3, padding=1)
// (diinn.py:168; row = c * 9 + ky * 3 + kx) written directly as a tiled plane group over the CELL axis,
// [ceil(B H W / 32)][rows][32] with rows >= 576 (the rows past 576 zero: diinn_plane_gemm_nt wants M % 128 == 0) -- the B
// operand of dWx^T = unfold . dP^T.  A gather of 9.4 MB into 94 MB: one thread per element, 128-byte store runs.
// ---------------------------------------------------------------------------------
struct UnfoldTiledParams {
    const float* feat;       // [B][64][H][W]
    float* out;              // [ceil(B*H*W / 32)][rows][32]
    int B, H, W, rows;
};

__global__ __launch_bounds__(256) void unfold_tiled_kernel(const UnfoldTiledParams p) {
    const long long tile = blockIdx.x;
    const int lane = threadIdx.x & 31;
    const long long n = tile * PLANE_TILE + lane;                    // flattened cell (b, cy, cx)
    const long long cells = (long long)p.B * p.H * p.W;
    const bool in = n < cells;
    const int hw = p.H * p.W;
    const int b = in ? (int)(n / hw) : 0;
    const int c0 = in ? (int)(n - (long long)b * hw) : 0;
    const int cy = c0 / p.W, cx = c0 - cy * p.W;
    float* __restrict__ dst = p.out + (size_t)tile * p.rows * PLANE_TILE + lane;
    for (int row = (int)(threadIdx.x >> 5) + 8 * (int)blockIdx.y; row < p.rows; row += 8 * (int)gridDim.y) {
        float v = 0.0f;
        if (in && row < C_IN * 9) {
            const int c = row / 9, tap = row - 9 * c;
            const int y = cy + tap / 3 - 1, x = cx + tap % 3 - 1;
            if (y >= 0 && y < p.H && x >= 0 && x < p.W) v = p.feat[(((size_t)b * C_IN + c) * p.H + y) * p.W + x];
        }
        dst[(size_t)row * PLANE_TILE] = v;
    }
}

// ---------------------------------------------------------------------------------
// plane GEMM (training backward, weight gradients): C[M x Nc] = A[M x npix] . B[Nc x npix]^T, A and B being rows
// [a_row0, a_row0+M) / [b_row0, b_row0+Nc) of tiled plane groups, i.e. a GEMM whose reduction axis is the pixel axis.
// Split-K: workgroup (block, ks) reduces the plane tiles of chunk ks for a 128 x 256 output block and writes its partial
// product to part[ks]; the caller adds the ksplit partials (sum_parts_kernel: fixed order, no atomics).  4 waves = 2 (M) x 2 (N),
// wave tile 64 x 128 = 2 x 4 MFMA tiles (128 accumulator registers).  With tiled planes a 32-row x 32-pixel MFMA panel is one
// contiguous 4 KiB block; the MFMA k-pair (pixel e, pixel 4+e) is the same for A and B, and the sum over pixels does not care
// about the order.  Optional extra column Nc: row sums of A (bias gradients).
// (Rounds 1-5 streamed the operands global -> registers per wave, plane_gemm_kernel<NB>: 110 TFLOP/s; deleted in round 6 for
// the LDS-staged kernel below -- profiles/r06_train.txt has the comparison and the timing ablations.)
// ---------------------------------------------------------------------------------
struct PlaneGemmParams {
    const float* A;          // tiled group, a_rows rows per tile; rows [a_row0, a_row0 + M) are used
    const float* Bm;         // tiled group, b_rows rows per tile; rows [b_row0, b_row0 + Nc)
    float* part;             // [ksplit][M][ldc]
    long long npix;
    int a_rows, a_row0, b_rows, b_row0;
    int M, Nc, ldc, tiles_per_split, with_rowsum;
};

// ---------------------------------------------------------------------------------
// plane_gemm_lds_kernel: the product above with the operands staged through LDS.
// What bounded the register-streaming kernel at 0.72 of the fp32 peak (PMC: MfmaUtil 76 % at 2.26 GHz) was how its operands arrive: every
// wave fetched its own 64 x 32 and 128 x 32 panels with loads that touch 32 different 128-byte lines per instruction, 32 bytes
// of each (an MFMA fragment is a COLUMN of the row-major panel), twice per workgroup (two waves share every panel), held in 192
// registers beside the 128 accumulators (3 VALU instructions per MFMA of AGPR staging); with the loads re-ordered to hold one
// tile (no AGPR staging: 0.6 VALU per MFMA) the time did not move -- the four loads of a line were then 32 MFMAs apart and
// the 32 KiB L1 did not keep it.  Here a tile's 384 rows x 128 B (48 KiB) are copied ONCE per workgroup, as 48 fully
// coalesced 1 KiB LDS-DMA pieces (12 per wave, no registers), two stages; the column reads are ds_read_b128.  The deposit of
// an LDS-DMA is lane-linear, so the bank swizzle is applied on the SOURCE side: lane (row r of 8, position p of 8) of a piece
// fetches 16-byte chunk p ^ r of its row, i.e. chunk c of row rr sits at position c ^ (rr & 7) -- a column read then spreads
// over 8 positions (2-way conflicts instead of 32-way).  One barrier per tile: [own pieces landed: vmcnt(0)] -> barrier (every
// wave's pieces landed, every wave done with the other stage) -> request the next tile into the other stage -> 128 MFMAs.
// ---------------------------------------------------------------------------------
constexpr int PG_STAGE_FLOATS = (128 + 256) * PLANE_TILE;        // 12,288 floats = 48 KiB

__global__ __launch_bounds__(256, 1) void plane_gemm_lds_kernel(const PlaneGemmParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * PG_STAGE_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int nblk = p.Nc / 256;
    const int mb = blockIdx.x / nblk, nb = blockIdx.x % nblk;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = mb * 128 + wm * 64, n0 = nb * 256 + wn * 128;
    const int ks = blockIdx.y;
    const long long ntiles = (p.npix + PLANE_TILE - 1) / PLANE_TILE;
    const long long t0 = (long long)ks * p.tiles_per_split;
    long long t1 = t0 + p.tiles_per_split;
    if (t1 > ntiles) t1 = ntiles;
    const int nt = t1 > t0 ? (int)(t1 - t0) : 0;

    const unsigned a_pitch = (unsigned)p.a_rows * PLANE_ROW_BYTES, b_pitch = (unsigned)p.b_rows * PLANE_ROW_BYTES;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + ((size_t)t0 * p.a_rows + p.a_row0 + mb * 128) * PLANE_TILE), 0, (int)(nt * a_pitch), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.Bm + ((size_t)t0 * p.b_rows + p.b_row0 + nb * 256) * PLANE_TILE), 0, (int)(nt * b_pitch), 0x00020000);
    // piece i (0..47) of a stage = rows 8 i .. 8 i + 7 of the sequence [A rows 0..127][B rows 0..255]; its lane (r, pos) fetches
    // chunk pos ^ r of row 8 i + r: the same per-lane offset for every piece
    const unsigned dma_off = (unsigned)(lane >> 3) * PLANE_ROW_BYTES + (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
    auto dma_piece = [&](int st, int t, int k) {                 // this wave's k-th piece (0..11) of tile t into stage st
        float* dst = lds + st * PG_STAGE_FLOATS;
        const int i = wave * 12 + k;                              // wave-uniform: wave 0 and a third of wave 1 copy A, the rest B
        if (i < 16)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dst + i * 256), 16, (int)dma_off,
                                                     (int)((unsigned)t * a_pitch + (unsigned)i * 1024u), 0, PG_A_AUX);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dst + i * 256), 16, (int)dma_off,
                                                     (int)((unsigned)t * b_pitch + (unsigned)(i - 16) * 1024u), 0, PG_B_AUX);
    };
    auto dma = [&](int st, int t) {
#pragma unroll
        for (int k = 0; k < 12; ++k) dma_piece(st, t, k);
    };
    // fragment (panel rows 32 P .. 32 P + 31, k-piece q) of lane (j, h): chunk 2 q + h of row 32 P + j
    int cq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cq[q] = j * PLANE_TILE + (((2 * q + h) ^ (j & 7)) * 4);
    const float* __restrict__ abase = lds + (2 * wm) * 32 * PLANE_TILE;
    const float* __restrict__ bbase = lds + (128 + 4 * wn * 32) * PLANE_TILE;

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float rs[2] = {0.0f, 0.0f};
    const bool sums = p.with_rowsum && n0 == 0;                  // wave-uniform

    if (nt > 0) dma(0, 0);
    for (int t = 0; t < nt; ++t) {
        const int st = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of tile t have landed
        __syncthreads();                                         // ... everybody's have; and everybody is done with the other stage
        const bool more = t + 1 < nt;
        const bool ragged = (t0 + t) * PLANE_TILE + PLANE_TILE > p.npix;   // the last tile's padding was never written
        const long long pix0 = (t0 + t) * PLANE_TILE;
        f32x4 fa[2][2], fb[2][4];                                 // [q parity][panel]
        auto frag = [&](int par, int q) {
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[par][a] = *(const f32x4*)(abase + st * PG_STAGE_FLOATS + a * 32 * PLANE_TILE + cq[q]);
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[par][b] = *(const f32x4*)(bbase + st * PG_STAGE_FLOATS + b * 32 * PLANE_TILE + cq[q]);
        };
        frag(0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int par = q & 1;
            if (q + 1 < 4) frag(par ^ 1, q + 1);                  // the next piece's reads go out in front of this piece's MFMAs
            if (ragged) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = pix0 + 8 * q + 4 * h + e < p.npix;
#pragma unroll
                    for (int a = 0; a < 2; ++a) fa[par][a][e] = in ? fa[par][a][e] : 0.0f;
#pragma unroll
                    for (int b = 0; b < 4; ++b) fb[par][b][e] = in ? fb[par][b][e] : 0.0f;
                }
            }
            // the next tile's 12 pieces go out ONE AT A TIME, each behind 8 MFMAs: an LDS-DMA instruction holds its wave's
            // issue for ~60-100 cycles -- free behind a 64-cycle fp32 MFMA, 12 of them back to back cost 11 MFMA slots per tile
            // (measured: all twelve in front of the tile's MFMAs 1.376 ms, none 1.232; profiles/r06_train.txt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = MFMA32(fa[par][a][e], fb[par][b][e], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
                if (more && 4 * q + e < 12) dma_piece(st ^ 1, t + 1, 4 * q + e);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (sums) {
#pragma unroll
                for (int a = 0; a < 2; ++a) rs[a] += (fa[par][a][0] + fa[par][a][1]) + (fa[par][a][2] + fa[par][a][3]);
            }
        }
    }

    float* __restrict__ dst = p.part + (size_t)ks * p.M * p.ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                dst[(size_t)row * p.ldc + n0 + 32 * b + j] = acc[a][b][r];
            }
    if (sums) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float v = rs[a] + __shfl_xor(rs[a], 32);
            if (h == 0) dst[(size_t)(m0 + 32 * a + j) * p.ldc + p.Nc] = v;
        }
    }
}

// ---------------------------------------------------------------------------------
// plane_rowdot_kernel (training backward, the two skinny products): C[M x 4] = A[M x npix] . S[4 x npix]^T
// with A rows of a tiled group (M = 256 or 512) and S a tiled 4-row group.  Used for
//   layer 0: (g_a,0 ; g_s,0) . (rel_h, rel_w, ratio, 1)^T  -> dbK_0, dQ0, dbQ0          (diinn.py:133-134,165-167)
//   head   : q_3 . (g_out0, g_out1, g_out2, 0)^T            -> d last_layer.weight      (diinn.py:138)
// HBM-bound (reads A once); split over the tiles like the plane GEMM, partials added by the caller.
// ---------------------------------------------------------------------------------
struct RowDotParams {
    const float* A;          // tiled, a_rows per tile, rows [0, M)
    const float* S;          // tiled [ntiles][4][32]
    float* part;             // [splits][M][4]
    long long npix;
    int a_rows, M, tiles_per_split;
};

// Round 6: a lane owns one 16-byte piece of one row -- lane = (row r of 8, piece q of 8) -- so a wave instruction reads 8 rows x
// 128 B = 1 KiB CONTIGUOUS of the tile (one thread per row read 64 different lines per instruction, 16 bytes of each, and leaned
// on the L1 to keep them until its 8th load: ~3 TB/s).  A wave takes 8-row groups wave, wave + 4, ...; each lane keeps the partial
// dot products of its 4 pixels against the 4 columns for every group it owns (M / 32 groups: 16 for M = 512) and the 8 lanes of
// a row meet once at the end (3 butterfly steps).  Sums over a row's pixels are taken in another order than before (per piece,
// then across pieces): results differ from the round-5 kernel by fp32 reassociation only (the test bounds it against float64).
__global__ __launch_bounds__(256) void plane_rowdot_kernel(const RowDotParams p) {
    const long long ntiles = (p.npix + PLANE_TILE - 1) / PLANE_TILE;
    const long long t0 = (long long)blockIdx.x * p.tiles_per_split;
    long long t1 = t0 + p.tiles_per_split;
    if (t1 > ntiles) t1 = ntiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane >> 3, q = lane & 7;
    constexpr int MAXG = 16;                                     // row groups per wave: M <= 4 waves x 16 groups x 8 rows = 512
    const int groups = p.M / 8;                                  // M % 8 == 0 (launch check)
    f32x4 c[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) c[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (long long t = t0; t < t1; ++t) {
        const float* __restrict__ a = p.A + (size_t)t * p.a_rows * PLANE_TILE;
        const f32x4* __restrict__ s = (const f32x4*)(p.S + (size_t)t * 4 * PLANE_TILE);
        const int left = (int)(p.npix - t * PLANE_TILE < PLANE_TILE ? p.npix - t * PLANE_TILE : PLANE_TILE);
        const f32x4 s0 = s[q], s1 = s[8 + q], s2 = s[16 + q], s3 = s[24 + q];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int row = 8 * (wave + 4 * g) + r;
            if (wave + 4 * g < groups) {
                f32x4 av = STREAM_LD_NT ? __builtin_nontemporal_load((const f32x4*)(a + (size_t)row * PLANE_TILE + 4 * q))
                                        : *(const f32x4*)(a + (size_t)row * PLANE_TILE + 4 * q);
                if (left < PLANE_TILE) {                          // ragged last tile: padding was never written
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = 4 * q + e < left ? av[e] : 0.0f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    c[g][0] = __builtin_fmaf(av[e], s0[e], c[g][0]);
                    c[g][1] = __builtin_fmaf(av[e], s1[e], c[g][1]);
                    c[g][2] = __builtin_fmaf(av[e], s2[e], c[g][2]);
                    c[g][3] = __builtin_fmaf(av[e], s3[e], c[g][3]);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        if (wave + 4 * g >= groups) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = c[g][k];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            c[g][k] = v;
        }
        if (q == 0) {
            const int row = 8 * (wave + 4 * g) + r;
            *(f32x4*)(p.part + ((size_t)blockIdx.x * p.M + row) * 4) = c[g];
        }
    }
}

// ---------------------------------------------------------------------------------
// sum_parts_kernel: out[i] = sum over the slices of part[k][i] in a FIXED order (deterministic), the add of the split partial
// products the GEMM / rowdot launches leave: 101 MB of GEMM partials at B = 16, 192 x 192 (3 x 64 slices of 512 x 257) and the
// rowdot partials (1,024 slices of 1,024 .. 2,048 floats; torch.sum over their slice axis: 137 us).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, long long n4) {
    // a workgroup = 32 column quads x 8 slice groups: thread (quad, g) adds slices g, g + 8, ... in order, the 8 group sums meet in LDS
    // and are added in group order -- a fixed order whatever the launch, and enough threads for SHORT rows with MANY slices too (the
    // rowdot partials: 1,024 slices of 1,024 .. 2,048 floats, which one thread per column walked as a 1,024-step chain)
    __shared__ f32x4 red[8][32];
    const int cq = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + cq;             // f32x4 index inside a slice
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    if (i < n4) {
        const f32x4* __restrict__ src = (const f32x4*)part + (size_t)blockIdx.y * nparts * n4 + i;
        int k = g;
        for (; k + 24 < nparts; k += 32) {                       // four slices in flight
            const f32x4 a = src[(size_t)k * n4], b = src[(size_t)(k + 8) * n4], c = src[(size_t)(k + 16) * n4], d = src[(size_t)(k + 24) * n4];
            s = (((s + a) + b) + c) + d;
        }
        for (; k < nparts; k += 8) s += src[(size_t)k * n4];
    }
    red[g][cq] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
        f32x4 t = red[0][cq];
#pragma unroll
        for (int gg = 1; gg < 8; ++gg) t += red[gg][cq];
        ((f32x4*)out)[(size_t)blockIdx.y * n4 + i] = t;
    }
}

extern "C" {

int diinn_sum_parts(void* stream, const float* part_dev, float* out_dev, int groups, int nparts, long long n) {
    // part_dev [groups][nparts][n] -> out_dev [groups][n]; n % 4 == 0, 16-byte aligned buffers
    if (!part_dev || !out_dev || groups <= 0 || groups > 65535 || nparts <= 0 || n <= 0) return DIINN_ERR_INVALID_ARG;
    if ((n & 3) || (((size_t)part_dev) & 15) || (((size_t)out_dev) & 15)) return DIINN_ERR_INVALID_ARG;
    const long long n4 = n / 4;
    if ((n4 + 31) / 32 > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n4 + 31) / 32), (unsigned)groups), dim3(256), 0, (hipStream_t)stream,
                       part_dev, out_dev, nparts, n4);
    return hip_status(hipGetLastError());
}

long long diinn_training_plane_floats(long long npix, int rows) {
    if (npix <= 0 || rows <= 0 || npix > DIINN_TRAIN_MAX_PIXELS) return -1;
    return (npix + PLANE_TILE - 1) / PLANE_TILE * rows * PLANE_TILE;
}

int diinn_backward_data(void* stream, const float* gout_planes_dev, const float* acts_dev,
                        const float* packed_dev, float* G_dev, float* Q_dev, long long npix) {
    if (!gout_planes_dev || !acts_dev || !packed_dev || !G_dev || !Q_dev) return DIINN_ERR_INVALID_ARG;
    const int stp = check_npix(npix);
    if (stp) return stp;
    BwdParams p;
    p.Wt = packed_dev; p.acts = acts_dev; p.gout = gout_planes_dev; p.G = G_dev; p.Q = Q_dev;
    p.npix = npix; p.ntiles = (npix + PLANE_TILE - 1) / PLANE_TILE; p.layer = 0;
    // DIINN_TRAIN_SPLIT_HEAD = 1 (test / A-B only): bwd_head_kernel as a launch of its own, as before round 6
    const bool split_head = knob(diinn_knobs().train_split_head) != 0;
    hipError_t e;
    if (split_head) {
        hipLaunchKernelGGL(bwd_head_kernel, dim3((unsigned)((npix + 255) / 256), HID / 16), dim3(256), 0,
                           (hipStream_t)stream, p);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_status(e);
    }
    const unsigned blocks = (unsigned)((p.ntiles + 3) / 4);
    for (int layer = 3; layer >= 1; --layer) {
        p.layer = layer;
        if (layer == 3 && !split_head) hipLaunchKernelGGL(bwd_layer_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(bwd_layer_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_status(e);
    }
    return DIINN_OK;
}

int diinn_plane_gemm_nt(void* stream, const float* A_dev, int a_rows, int a_row0, const float* B_dev, int b_rows,
                        int b_row0, float* part_dev, int M, int Nc, long long npix, int ksplit, int with_rowsum) {
    if (!A_dev || !B_dev || !part_dev || M <= 0 || Nc <= 0 || ksplit <= 0 || a_row0 < 0 || b_row0 < 0 ||
        a_row0 + M > a_rows || b_row0 + Nc > b_rows)
        return DIINN_ERR_INVALID_ARG;
    const int stp = check_npix(npix);
    if (stp) return stp;
    if (M % 128 || Nc % 256) return DIINN_ERR_UNSUPPORTED;
    if ((((size_t)A_dev) | ((size_t)B_dev)) & 15) return DIINN_ERR_INVALID_ARG;          // the tiles are copied in 16-byte pieces
    if (ksplit > 65535) return DIINN_ERR_TOO_LARGE;
    PlaneGemmParams p;
    p.A = A_dev; p.Bm = B_dev; p.part = part_dev; p.npix = npix;
    p.a_rows = a_rows; p.a_row0 = a_row0; p.b_rows = b_rows; p.b_row0 = b_row0;
    p.M = M; p.Nc = Nc;
    p.ldc = Nc + (with_rowsum ? 1 : 0);
    p.with_rowsum = with_rowsum ? 1 : 0;
    const long long ntiles = (npix + PLANE_TILE - 1) / PLANE_TILE;
    const long long per = (ntiles + ksplit - 1) / ksplit;
    // offsets inside one split are 32-bit: tiles_per_split * rows * 128 bytes must stay below 2 GiB
    if (per * (long long)(a_rows > b_rows ? a_rows : b_rows) * PLANE_ROW_BYTES >= 0x7FFFFFFFLL) return DIINN_ERR_TOO_LARGE;
    p.tiles_per_split = (int)per;
    hipLaunchKernelGGL(plane_gemm_lds_kernel, dim3((M / 128) * (Nc / 256), ksplit), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_plane_rowdot(void* stream, const float* A_dev, int a_rows, const float* S_dev, float* part_dev,
                       int M, long long npix, int splits) {
    if (!A_dev || !S_dev || !part_dev || M <= 0 || M > a_rows || splits <= 0) return DIINN_ERR_INVALID_ARG;
    const int stp = check_npix(npix);
    if (stp) return stp;
    if (M % 8 || M > 512) return DIINN_ERR_UNSUPPORTED;          // 8-row groups, at most 16 per wave
    if ((((size_t)A_dev) & 15) || (((size_t)S_dev) & 15) || (((size_t)part_dev) & 15)) return DIINN_ERR_INVALID_ARG;
    RowDotParams p;
    p.A = A_dev; p.S = S_dev; p.part = part_dev; p.npix = npix; p.a_rows = a_rows; p.M = M;
    const long long ntiles = (npix + PLANE_TILE - 1) / PLANE_TILE;
    p.tiles_per_split = (int)((ntiles + splits - 1) / splits);
    hipLaunchKernelGGL(plane_rowdot_kernel, dim3(splits), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_backward_cell_sum_ex(void* stream, const float* G_dev, const int32_t* seg_h_dev, const int32_t* seg_w_dev,
                               float* dP_dev, float* dP_tiled_dev, int B, int H, int W, int Hu, int Wu) {
    if (!G_dev || !seg_h_dev || !seg_w_dev || !dP_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Hu <= 0 || Wu <= 0) return DIINN_ERR_INVALID_ARG;
    const long long npix = (long long)B * Hu * Wu;
    st = check_npix(npix);
    if (st) return st;
    if (B > 65535 || (long long)H * W > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    if ((((size_t)G_dev) & 15)) return DIINN_ERR_INVALID_ARG;
    CellSumParams p{G_dev, dP_dev, dP_tiled_dev, seg_h_dev, seg_w_dev, B, H, W, Hu, Wu, (npix + PLANE_TILE - 1) / PLANE_TILE};
    hipLaunchKernelGGL(cell_sum_kernel, dim3((unsigned)(((long long)H * W + 255) / 256), (unsigned)B, PCH), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_backward_cell_sum(void* stream, const float* G_dev, const int32_t* seg_h_dev, const int32_t* seg_w_dev,
                            float* dP_dev, int B, int H, int W, int Hu, int Wu) {
    return diinn_backward_cell_sum_ex(stream, G_dev, seg_h_dev, seg_w_dev, dP_dev, nullptr, B, H, W, Hu, Wu);
}

int diinn_unfold_tiled(void* stream, const float* feat_dev, float* out_tiled_dev, int rows, int B, int H, int W) {
    if (!feat_dev || !out_tiled_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (rows < C_IN * 9 || rows > 65535) return DIINN_ERR_INVALID_ARG;
    const long long tiles = ((long long)B * H * W + PLANE_TILE - 1) / PLANE_TILE;
    if (tiles > 2147483000LL) return DIINN_ERR_TOO_LARGE;
    UnfoldTiledParams p{feat_dev, out_tiled_dev, B, H, W, rows};
    hipLaunchKernelGGL(unfold_tiled_kernel, dim3((unsigned)tiles, 4), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

}  // extern "C"
