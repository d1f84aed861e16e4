// PROTOTYPE (microbenchmark only, not in the library): conv_wino4_kernel without dedicated transform waves -- twelve waves on
// all four SIMDs, each with its three positions' MFMAs AND a third of one channel pair's data transform (two of the six
// output rows of B^T d B; the vertical pass of its rows recomputed from five raw rows: +25 % transform arithmetic, no
// exchange).  Question: does MFMA work on four SIMDs with the transform interleaved in the same waves beat three MFMA SIMDs
// + one transform SIMD (csrc/diinn_winograd4.hip)?   Built by tools/r04_wino4_abl.sh as variant SYM.
#include "../../dual-interactive-implicit-neural-network_amd/csrc/diinn_device.h"

constexpr int W4_TX = 32;
constexpr int W4_THREADS = 768;
constexpr int W4_VBUF = 36 * 256;
constexpr int W4_RAW0 = 3 * W4_VBUF;
constexpr int W4_RAW_WAVE = 6 * 256 + 64;
constexpr int W4_LDS_FLOATS = W4_RAW0 + 2 * 4 * W4_RAW_WAVE;
constexpr int W4_PIECE_BYTES = 1024;
#pragma clang diagnostic ignored "-Winline-asm"
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 w4_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
struct ConvWino4Params {
    const float* in; const float* wu; const float* bias; const float* res; float* out;
    long long in_bs, out_bs, res_bs;
    int Cin, B, H, W, relu;
};
template <typename T>
__device__ __forceinline__ void w4_bt(const T d0, const T d1, const T d2, const T d3, const T d4, const T d5,
                                      T& r0, T& r1, T& r2, T& r3, T& r4, T& r5) {
    const T a = __builtin_elementwise_fma(T(-4.0f), d2, d4), b = __builtin_elementwise_fma(T(-4.0f), d1, d3);
    const T c = d4 - d2, e = d3 - d1;
    r0 = __builtin_elementwise_fma(T(4.0f), d0, __builtin_elementwise_fma(T(-5.0f), d2, d4));
    r1 = a + b; r2 = a - b;
    r3 = __builtin_elementwise_fma(T(2.0f), e, c);
    r4 = __builtin_elementwise_fma(T(-2.0f), e, c);
    r5 = __builtin_elementwise_fma(T(4.0f), d1, __builtin_elementwise_fma(T(-5.0f), d3, d5));
}
template <typename T>
__device__ __forceinline__ void w4_at(const T m0, const T m1, const T m2, const T m3, const T m4, const T m5, T& y0, T& y1, T& y2, T& y3) {
    const T s = m1 + m2, d = m1 - m2, u = m3 + m4, v = m3 - m4;
    y0 = (m0 + s) + u;
    y1 = __builtin_elementwise_fma(T(2.0f), v, d);
    y2 = __builtin_elementwise_fma(T(4.0f), u, s);
    y3 = __builtin_elementwise_fma(T(8.0f), v, d) + m5;
}

__device__ __forceinline__ void conv_wino4_body(const ConvWino4Params& p, float* __restrict__ lds, int b, int blk, int hh0) {
    const int tiles_x = (p.W + 3) / 4, tiles_n = tiles_x * ((p.H + 3) / 4);
    auto tile_xy = [&](int tile, int& tx, int& ty) {
        ty = tile / tiles_x;
        tx = tile - ty * tiles_x;
        return tile < tiles_n;
    };
    const int lane = threadIdx.x & 63;
    const int mw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // 0..11
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_b = (unsigned)(plane * sizeof(float));
    const int n = p.Cin / 8;
    constexpr unsigned OUTSIDE = 0x80000000u;
    const bool ragged = (p.W & 3) != 0;
    const float* __restrict__ in_b = p.in + (size_t)b * p.in_bs;

    // ---- transform share: channel pair cp, output rows 2 rp, 2 rp + 1 of the transformed tile
    const int cp = mw & 3, rp = mw >> 2;
    const int th = lane >> 5, tm = lane & 31;
    int ptx, pty;
    const bool pin = tile_xy(blk * W4_TX + tm, ptx, pty);
    const bool first = tm == 0, last = tm == W4_TX - 1;
    const bool row_start = ptx == 0, row_end = 4 * ptx + 4 >= p.W;
    const bool ok1 = 4 * ptx + 1 < p.W, ok2 = 4 * ptx + 2 < p.W, ok3 = 4 * ptx + 3 < p.W;
    const float* __restrict__ raw = lds + W4_RAW0 + cp * W4_RAW_WAVE;
    const int k0 = rp == 0 ? 0 : 1;                              // raw rows k0 .. k0 + 4
    const int la = (first ? 1536 + 12 * th : th * 128 + tm * 4 - 1) + k0 * (first ? 1 : 256), lstep = first ? 1 : 256;
    const int ra = (last ? 1536 + 12 * th + 6 : th * 128 + tm * 4 + 4) + k0 * (last ? 1 : 256), rstep = last ? 1 : 256;
    auto tshare = [&](int slot, float* __restrict__ vb) {
        const float* __restrict__ src = raw + slot * 4 * W4_RAW_WAVE;
        f32x2 d[5][3];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(src + (k0 + k) * 256 + th * 128 + tm * 4);
            float c1 = c4[1], c2 = c4[2], c3 = c4[3];
            if (ragged) { c1 = ok1 ? c1 : 0.0f; c2 = ok2 ? c2 : 0.0f; c3 = ok3 ? c3 : 0.0f; }
            d[k][0] = f32x2{row_start ? 0.0f : src[la + k * lstep], c4[0]};
            d[k][1] = f32x2{c1, c2};
            d[k][2] = f32x2{c3, row_end ? 0.0f : src[ra + k * rstep]};
        }
        f32x2 t[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (rp == 0) {                                       // rows 0, 1 of B^T from d0..d4 = d[0..4]
                t[0][j] = __builtin_elementwise_fma(f32x2(4.0f), d[0][j], __builtin_elementwise_fma(f32x2(-5.0f), d[2][j], d[4][j]));
                t[1][j] = __builtin_elementwise_fma(f32x2(-4.0f), d[2][j], d[4][j]) + __builtin_elementwise_fma(f32x2(-4.0f), d[1][j], d[3][j]);
            } else if (rp == 1) {                                // rows 2, 3 from d1..d4 = d[0..3]
                t[0][j] = __builtin_elementwise_fma(f32x2(-4.0f), d[1][j], d[3][j]) - __builtin_elementwise_fma(f32x2(-4.0f), d[0][j], d[2][j]);
                t[1][j] = __builtin_elementwise_fma(f32x2(2.0f), d[2][j] - d[0][j], d[3][j] - d[1][j]);
            } else {                                             // rows 4, 5 from d1..d5 = d[0..4]
                t[0][j] = __builtin_elementwise_fma(f32x2(-2.0f), d[2][j] - d[0][j], d[3][j] - d[1][j]);
                t[1][j] = __builtin_elementwise_fma(f32x2(4.0f), d[0][j], __builtin_elementwise_fma(f32x2(-5.0f), d[2][j], d[4][j]));
            }
        }
        float* __restrict__ dst = vb + (2 * cp + th) * 32 + tm + 12 * rp * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v0, v1, v2, v3, v4, v5;
            w4_bt<float>(t[i][0][0], t[i][0][1], t[i][1][0], t[i][1][1], t[i][2][0], t[i][2][1], v0, v1, v2, v3, v4, v5);
            dst[(6 * i + 0) * 256] = v0; dst[(6 * i + 1) * 256] = v1; dst[(6 * i + 2) * 256] = v2;
            dst[(6 * i + 3) * 256] = v3; dst[(6 * i + 4) * 256] = v4; dst[(6 * i + 5) * 256] = v5;
        }
    };

    // ---- raw input requests (LDS-DMA, inline asm): wave w rows 2 w, 2 w + 1 of the chunk's 24; waves 0..3 the edge values
    unsigned dvoff[2], dvoffe = OUTSIDE, dlds[2], dldse;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = 2 * mw + i, dcp = idx / 6, k = idx - 6 * dcp;
        const int y = 4 * pty - 1 + k;
        dvoff[i] = (pin && y >= 0 && y < p.H) ? (unsigned)(2 * dcp + th) * plane_b + (unsigned)(y * p.W + 4 * ptx) * 4u : OUTSIDE;
        dlds[i] = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + W4_RAW0 + dcp * W4_RAW_WAVE + k * 256);
    }
    {
        const int ek = lane % 6, ew = lane / 6;
        int etx, ety;
        const bool ein = tile_xy(blk * W4_TX + ((ew & 1) ? W4_TX - 1 : 0), etx, ety);
        const int ex = (ew & 1) ? 4 * etx + 4 : 4 * etx - 1, ey = 4 * ety - 1 + ek;
        if (mw < 4 && lane < 24 && ein && ey >= 0 && ey < p.H && ex >= 0 && ex < p.W)
            dvoffe = (unsigned)(2 * mw + (ew >> 1)) * plane_b + (unsigned)(ey * p.W + ex) * 4u;
        dldse = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + W4_RAW0 + (mw & 3) * W4_RAW_WAVE + 1536);
    }
    auto fetch = [&](int slot, int c) {
        const i32x4 irs = w4_rsrc(in_b + (size_t)8 * c * plane, 8u * plane_b);
        const unsigned so = (unsigned)slot * (4 * W4_RAW_WAVE * 4);
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(dlds[0] + so), "v"(dvoff[0]), "s"(irs) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(dlds[1] + so), "v"(dvoff[1]), "s"(irs) : "memory", "m0");
        if (mw < 4)
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(dldse + so), "v"(dvoffe), "s"(irs) : "memory", "m0");
    };
    auto chunk_of = [&](int k) { return k < n ? k : n - 1; };

    // ---- MFMA role
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wu + (size_t)(mw * 2 + hh0) * n * (3 * W4_PIECE_BYTES / 4)), 0, n * 3 * W4_PIECE_BYTES, 0x00020000);
    const int lane_off = lane * 16;
    const float* __restrict__ bsrc = lds + 3 * mw * 256 + lane;
    f32x16 acc[3];
    f32x4 A[2][3], Bf[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    fetch(0, 0);
    fetch(1, chunk_of(1));
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        A[0][q] = ld_piece(wrs, lane_off, q * W4_PIECE_BYTES);
        A[1][q] = ld_piece(wrs, lane_off, ((n > 1 ? 3 : 0) + q) * W4_PIECE_BYTES);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // chunks 0 and 1 have landed
    tshare(0, lds);
    tshare(1, lds + W4_VBUF);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // ... are transformed; the raw slots are free
    fetch(0, chunk_of(2));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // chunk 2 has landed
#pragma unroll
    for (int q = 0; q < 3; ++q) Bf[q] = f32x4{bsrc[q * 256], bsrc[q * 256 + 64], bsrc[q * 256 + 128], bsrc[q * 256 + 192]};

    int slot1 = 1, slot2 = 2;
    auto iter = [&](auto PAR_, int c) {
        constexpr int PAR = decltype(PAR_)::value;
        const int c2 = c + 2 < n ? c + 2 : n - 1;
        fetch((c + 1) & 1, chunk_of(c + 3));                     // lands by this iteration's barrier; read in the next
        W4_SB();
#ifndef SYM_NOTRANSFORM
        if (c + 2 < n) tshare(c & 1, lds + slot2 * W4_VBUF);     // chunk c + 2: raw slot c & 1 (landed by the last barrier)
#endif
#ifdef SYM_FENCE
        W4_SB();
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q] = MFMA32(A[PAR][q][e], Bf[q][e], acc[q]);
            W4_SB();
            A[PAR][q] = ld_piece(wrs, lane_off, (c2 * 3 + q) * W4_PIECE_BYTES);
            {
                const float* __restrict__ bq = bsrc + slot1 * W4_VBUF + q * 256;
                Bf[q] = f32x4{bq[0], bq[64], bq[128], bq[192]};
            }
            W4_SB();
        }
        // all but the three weight requests: the raw rows have landed; all but the last position's two reads: the stores are done
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(2)\n\ts_barrier" ::: "memory");
        slot1 = slot1 == 2 ? 0 : slot1 + 1;
        slot2 = slot2 == 2 ? 0 : slot2 + 1;
    };
    int c = 0;
    for (; c + 1 < n; c += 2) {
        iter(IC<0>{}, c);
        iter(IC<1>{}, c + 1);
    }
    if (c < n) iter(IC<0>{}, c);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[((3 * mw + q) * 16 + r) * 64 + lane] = acc[q][r];
    __syncthreads();
    for (int r = mw; r < 16; r += 12) {
        const int h = lane >> 5, m = lane & 31;
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
        const bool vec = !ragged && (p.out_bs & 3) == 0 && (((size_t)p.out) & 15) == 0;
        const int co = 32 * hh0 + 8 * (r >> 2) + 4 * h + (r & 3);
        const float bias = p.bias[co];
        float* __restrict__ op = p.out + (size_t)b * p.out_bs + (size_t)co * plane;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = oy0 + a;
            if (!tin || oy >= p.H || ox >= p.W) continue;
            f32x4 v = {y[a][0] + bias, y[a][1] + bias, y[a][2] + bias, y[a][3] + bias};
            if (p.relu) v = f32x4{relu0(v[0]), relu0(v[1]), relu0(v[2]), relu0(v[3])};
            const size_t o = (size_t)oy * p.W + ox;
            if (vec) *reinterpret_cast<f32x4*>(op + o) = v;
            else
                for (int x = 0; x < 4; ++x)
                    if (ox + x < p.W) op[o + x] = v[x];
        }
    }
}

__global__ __launch_bounds__(W4_THREADS) void conv_wino4_kernel(const ConvWino4Params p) {
    __shared__ __attribute__((aligned(16))) float lds[W4_LDS_FLOATS];
    const int nblk = (((p.W + 3) / 4) * ((p.H + 3) / 4) + W4_TX - 1) / W4_TX;
    const int total = p.B * nblk * 2;
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
        __syncthreads();
    }
}

extern "C" int diinn_conv_wino4(void* stream, const float* in_dev, long long in_batch_stride, int Cin,
                                const float* packed_u_dev, const float* bias_dev, const float* res_dev, long long res_batch_stride,
                                float* out_dev, long long out_batch_stride, int relu, int B, int H, int W) {
    const long long blocks = (((long long)((W + 3) / 4) * ((H + 3) / 4) + W4_TX - 1) / W4_TX) * B;
    ConvWino4Params p;
    p.in = in_dev; p.wu = packed_u_dev; p.bias = bias_dev; p.res = res_dev; p.out = out_dev;
    p.in_bs = in_batch_stride; p.out_bs = out_batch_stride; p.res_bs = res_batch_stride;
    p.Cin = Cin; p.B = B; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
    hipLaunchKernelGGL(conv_wino4_kernel, dim3((unsigned)((2 * blocks + 7) / 8 * 8)), dim3(W4_THREADS), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}
