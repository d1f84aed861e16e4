// diinn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the DIINN implicit decoder
// and the launch half of the C ABI (include/diinn_hip.h).
//
// Reference path replaced: ImplicitDecoder.forward, mode 3
//   /root/reference/src/models/components/diinn.py:163-173 (+ :94-110, :132-139, :149-160)
//
// Two kernels (DESIGN.md has the derivation and the roofline of each):
//   precompute_P_kernel : per LR cell, P_i = Wx_i . unfold3x3(feat) + bK_i, i=0..3
//                         (implicit-im2col GEMM 576 -> 1024 on v_mfma_f32_32x32x2_f32)
//   decode_kernel       : per HR pixel, the dual-branch MLP.  One wave owns 32 pixels
//                         and keeps their 256-channel activation in registers for the
//                         whole network: the accumulator layout of one layer IS the
//                         B-operand layout of the next (diinn_layout.h), so activations
//                         never touch LDS or HBM.  Weights stream from the packed image.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (explicit fmaf where wanted)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diinn_hip.h"
#include "diinn_layout.h"

using namespace diinn;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------
// sine (reference: torch.sin via SineAct, diinn.py:21-26)
// ---------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float dsin(float x);

// Cody-Waite reduction to [-pi/2, pi/2] by multiples of pi, odd minimax polynomial.
// Absolute error <= ~2.5e-7 for |x| <= 1e4 (arguments here are O(1)..O(100)).
template <>
__device__ __forceinline__ float dsin<DIINN_SIN_ACCURATE>(float x) {
    const float k = __builtin_rintf(x * 0.31830988618379067154f);
    float r = __builtin_fmaf(k, -3.14159274101257324219f, x);      // pi, fp32 head
    r = __builtin_fmaf(k, 8.74227765734758577309e-08f, r);          // -(pi - head)
    const float s = r * r;
    // flip sign for odd k: (-1)^k
    const int ki = (int)k;
    r = __builtin_bit_cast(float, __builtin_bit_cast(int, r) ^ (ki << 31));
    float u = 2.6083159809786593541503e-06f;
    u = __builtin_fmaf(u, s, -0.0001981069071916863322258f);
    u = __builtin_fmaf(u, s, 0.00833307858556509017944336f);
    u = __builtin_fmaf(u, s, -0.166666597127914428710938f);
    return __builtin_fmaf(s, u * r, r);
}

// v_sin_f32 takes revolutions; fract keeps it inside the instruction's valid domain.
template <>
__device__ __forceinline__ float dsin<DIINN_SIN_HW>(float x) {
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189533577f));
}

// ---------------------------------------------------------------------------------
// decode kernel
// ---------------------------------------------------------------------------------
struct DecodeParams {
    const float* P;        // [B,H,W,1024]
    const float* Wt;       // packed image
    float* out;            // [B,3,Hu,Wu]
    int B, H, W, Hu, Wu, y0, y1;
    float ratio;           // fp32(H*W / (Hu*Wu))   (diinn.py:166)
    Axis ah, aw;
};

constexpr int TILE_W = 8, TILE_H = 4;           // one wave: 8x4 HR pixels
constexpr int WG_TILES_X = 2, WG_TILES_Y = 2;   // 4 waves -> 16x8 HR pixels per workgroup

template <int SIN_MODE>
__global__ __launch_bounds__(256, 1) void decode_kernel(const DecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;

    const int x = blockIdx.x * (TILE_W * WG_TILES_X) + (wave & 1) * TILE_W + (j & (TILE_W - 1));
    const int y = p.y0 + blockIdx.y * (TILE_H * WG_TILES_Y) + (wave >> 1) * TILE_H + (j / TILE_W);
    const int b = blockIdx.z;
    const bool valid = (x < p.Wu) && (y < p.y1);
    // whole wave outside the band/image: nothing to do (wave-uniform, no barriers in this kernel)
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(valid) == 0ull))) return;
    const int xc = x < p.Wu ? x : p.Wu - 1;
    const int yc = y < p.y1 ? y : p.y1 - 1;

    int iy, ix;
    float relh, relw;
    axis_eval(p.ah, yc, iy, relh);
    axis_eval(p.aw, xc, ix, relw);

    const float* __restrict__ Wt = p.Wt;
    const float* __restrict__ Pc = p.P + (((size_t)b * p.H + iy) * p.W + ix) * PCH + 4 * h;

    // ---- layer 0: q0 = relu(P_0[cell]) * sin(Q0 . (rel_h, rel_w, ratio) + bQ0)   (diinn.py:133-134)
    float q[128];
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
                    q[16 * m + 4 * g + e] = __builtin_fmaxf(pv[e], 0.0f) * dsin<SIN_MODE>(a);
                }
            }
        }
    }

    // ---- layers 1..3: [k;s] = [Wq_i;Qw_i] . q + [P_i[cell]; bQ_i];  q = relu(k) * sin(s)   (diinn.py:135-137)
#pragma unroll 1
    for (int layer = 0; layer < 3; ++layer) {
        const float* __restrict__ Wl = Wt + OFF_WL + (size_t)layer * WL_LAYER + lane * 4;
        const float* __restrict__ Pl = Pc + (layer + 1) * HID;
        const float* __restrict__ Bq = Wt + OFF_BQ + layer * HID + 4 * h;
        float qn[128];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x16 ak, as;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 pv = *(const f32x4*)(Pl + 32 * m + 8 * g);
                const f32x4 bv = *(const f32x4*)(Bq + 32 * m + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ak[4 * g + e] = pv[e];
                    as[4 * g + e] = bv[e];
                }
            }
#pragma unroll
            for (int kg = 0; kg < WL_KG; ++kg) {
                const f32x4 wk = *(const f32x4*)(Wl + ((size_t)(m * WL_KG + kg) * 2 + 0) * WL_PIECE);
                const f32x4 wq = *(const f32x4*)(Wl + ((size_t)(m * WL_KG + kg) * 2 + 1) * WL_PIECE);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ak = MFMA32(wk[e], q[4 * kg + e], ak);
                    as = MFMA32(wq[e], q[4 * kg + e], as);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                qn[16 * m + r] = __builtin_fmaxf(ak[r], 0.0f) * dsin<SIN_MODE>(as[r]);
        }
#pragma unroll
        for (int i = 0; i < 128; ++i) q[i] = qn[i];
    }

    // ---- head: out = L . q3 + bL   (diinn.py:138)
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    {
        const float* __restrict__ L = Wt + OFF_L + 4 * h;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m + 8 * g;
                const f32x4 l0 = *(const f32x4*)(L + 0 * HID + c0);
                const f32x4 l1 = *(const f32x4*)(L + 1 * HID + c0);
                const f32x4 l2 = *(const f32x4*)(L + 2 * HID + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = q[16 * m + 4 * g + e];
                    o0 = __builtin_fmaf(l0[e], v, o0);
                    o1 = __builtin_fmaf(l1[e], v, o1);
                    o2 = __builtin_fmaf(l2[e], v, o2);
                }
            }
        }
    }
    o0 += __shfl_xor(o0, 32);
    o1 += __shfl_xor(o1, 32);
    o2 += __shfl_xor(o2, 32);
    if (valid && h == 0) {
        const size_t plane = (size_t)p.Hu * p.Wu;
        float* o = p.out + (size_t)b * 3 * plane + (size_t)y * p.Wu + x;
        o[0] = o0 + Wt[OFF_BL + 0];
        o[plane] = o1 + Wt[OFF_BL + 1];
        o[2 * plane] = o2 + Wt[OFF_BL + 2];
    }
}

// ---------------------------------------------------------------------------------
// P kernel: P[b,y,x, i*256+ch] = sum_{c,ky,kx} Wx_i[ch,c,ky,kx] * feat[b,c,y+ky-1,x+kx-1] + bK_i[ch]
// (zero padding; diinn.py:168 unfold + the feature columns of K[i], diinn.py:133,136)
// Workgroup = 4 waves on the same 32 LR cells of one row; wave i produces P_i.
// ---------------------------------------------------------------------------------
struct PParams {
    const float* feat;   // [B,64,H,W]
    const float* Wt;
    float* P;            // [B,H,W,1024]
    int B, H, W, r0, r1;
};

__global__ __launch_bounds__(256, 1) void precompute_P_kernel(const PParams p) {
    const int lane = threadIdx.x & 63;
    const int i = threadIdx.x >> 6;          // which P_i this wave produces
    const int h = lane >> 5, j = lane & 31;
    const int x = blockIdx.x * 32 + j;
    const int y = p.r0 + blockIdx.y;
    const int b = blockIdx.z;

    // B operands: the 3x3 neighbourhood of this lane's cell, all 64 channels.
    // k-step kk = 32*t + cp: tap t, channel 2*cp + h.
    float fb[WP_KSTEPS];
    const float* __restrict__ fbase = p.feat + ((size_t)b * C_IN + h) * p.H * p.W;
    const size_t cstride = (size_t)2 * p.H * p.W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool ok = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
        const size_t off = ok ? ((size_t)yy * p.W + xx) : 0;
#pragma unroll
        for (int cp = 0; cp < 32; ++cp) {
            const float v = fbase[cp * cstride + off];
            fb[32 * t + cp] = ok ? v : 0.0f;
        }
    }

    const float* __restrict__ Wp = p.Wt + OFF_WP + lane * 4;
    const float* __restrict__ Bk = p.Wt + OFF_BK + i * HID + 4 * h;
    float* __restrict__ Pout = p.P + (((size_t)b * p.H + y) * p.W + x) * PCH + i * HID + 4 * h;
    const bool store = x < p.W;

#pragma unroll 1
    for (int m = 0; m < 8; m += 2) {
        f32x16 a0, a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b0 = *(const f32x4*)(Bk + 32 * m + 8 * g);
            const f32x4 b1 = *(const f32x4*)(Bk + 32 * (m + 1) + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[4 * g + e] = b0[e];
                a1[4 * g + e] = b1[e];
            }
        }
        const float* __restrict__ w0 = Wp + (size_t)(i * 8 + m) * WP_KG * WL_PIECE;
        const float* __restrict__ w1 = w0 + (size_t)WP_KG * WL_PIECE;
#pragma unroll
        for (int kg = 0; kg < WP_KG; ++kg) {
            const f32x4 u0 = *(const f32x4*)(w0 + (size_t)kg * WL_PIECE);
            const f32x4 u1 = *(const f32x4*)(w1 + (size_t)kg * WL_PIECE);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0 = MFMA32(u0[e], fb[4 * kg + e], a0);
                a1 = MFMA32(u1[e], fb[4 * kg + e], a1);
            }
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
                *(f32x4*)(Pout + 32 * m + 8 * g) = v0;
                *(f32x4*)(Pout + 32 * (m + 1) + 8 * g) = v1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// tables kernel (tests): the device evaluation of axis_eval
// ---------------------------------------------------------------------------------
__global__ void axis_tables_kernel(Axis a, int n_out, int32_t* idx, float* rel) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    int id;
    float r;
    axis_eval(a, j, id, r);
    if (idx) idx[j] = id;
    if (rel) rel[j] = r;
}

// ---------------------------------------------------------------------------------
// C ABI: launch functions
// ---------------------------------------------------------------------------------
static thread_local int g_last_hip_error = 0;

static int hip_status(hipError_t e) {
    if (e == hipSuccess) return DIINN_OK;
    g_last_hip_error = (int)e;
    return DIINN_ERR_HIP;
}

extern "C" {

int diinn_last_hip_error(void) { return g_last_hip_error; }

int diinn_make_axis_tables_device(void* stream, int n_in, int n_out, int small_output,
                                  int32_t* idx_dev, float* rel_dev) {
    if (n_in <= 0 || n_out <= 0) return DIINN_ERR_INVALID_ARG;
    const Axis a = make_axis(n_in, n_out, small_output ? 1 : 0);
    hipLaunchKernelGGL(axis_tables_kernel, dim3((n_out + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, a, n_out, idx_dev, rel_dev);
    return hip_status(hipGetLastError());
}

static int check_dims(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return DIINN_ERR_INVALID_ARG;
    if ((double)B * H * W * PCH >= 9.0e18 || B > 65535 || H > 65535) return DIINN_ERR_TOO_LARGE;
    return DIINN_OK;
}

int diinn_precompute_P(void* stream, const float* feat_dev, const float* packed_dev,
                       float* P_dev, int B, int H, int W, int r0, int r1) {
    if (!feat_dev || !packed_dev || !P_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (r0 < 0 || r1 > H || r0 >= r1) return DIINN_ERR_INVALID_ARG;
    PParams p{feat_dev, packed_dev, P_dev, B, H, W, r0, r1};
    const dim3 grid((W + 31) / 32, r1 - r0, B);
    hipLaunchKernelGGL(precompute_P_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_decode_launch_info(int B, int Hu, int Wu, int y0, int y1,
                             int* grid_x, int* grid_y, int* grid_z, int* block) {
    if (B <= 0 || Hu <= 0 || Wu <= 0 || y0 < 0 || y1 > Hu || y0 >= y1) return DIINN_ERR_INVALID_ARG;
    if (grid_x) *grid_x = (Wu + TILE_W * WG_TILES_X - 1) / (TILE_W * WG_TILES_X);
    if (grid_y) *grid_y = (y1 - y0 + TILE_H * WG_TILES_Y - 1) / (TILE_H * WG_TILES_Y);
    if (grid_z) *grid_z = B;
    if (block) *block = 256;
    return DIINN_OK;
}

int diinn_decode_band(void* stream, const float* P_dev, const float* packed_dev,
                      float* out_dev, int B, int H, int W, int Hu, int Wu,
                      int y0, int y1, int sin_mode) {
    if (!P_dev || !packed_dev || !out_dev) return DIINN_ERR_INVALID_ARG;
    int st = check_dims(B, H, W);
    if (st) return st;
    if (Hu <= 0 || Wu <= 0 || y0 < 0 || y1 > Hu || y0 >= y1) return DIINN_ERR_INVALID_ARG;
    if ((double)Hu * Wu >= 2.0e9) return DIINN_ERR_TOO_LARGE;
    if (sin_mode != DIINN_SIN_ACCURATE && sin_mode != DIINN_SIN_HW) return DIINN_ERR_UNSUPPORTED;
    int gx, gy, gz, blk;
    diinn_decode_launch_info(B, Hu, Wu, y0, y1, &gx, &gy, &gz, &blk);
    if (gy > 65535 * 16) return DIINN_ERR_TOO_LARGE;
    DecodeParams p;
    p.P = P_dev; p.Wt = packed_dev; p.out = out_dev;
    p.B = B; p.H = H; p.W = W; p.Hu = Hu; p.Wu = Wu; p.y0 = y0; p.y1 = y1;
    p.ratio = (float)(((double)H * (double)W) / ((double)Hu * (double)Wu));
    const int small = diinn_uses_small_output_kernel(Hu, Wu);
    p.ah = make_axis(H, Hu, small);
    p.aw = make_axis(W, Wu, small);
    const dim3 grid(gx, gy, gz);
    if (sin_mode == DIINN_SIN_HW)
        hipLaunchKernelGGL(decode_kernel<DIINN_SIN_HW>, grid, dim3(blk), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(decode_kernel<DIINN_SIN_ACCURATE>, grid, dim3(blk), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

int diinn_decode(void* stream, const float* feat_dev, const float* packed_dev,
                 float* workspace_dev, float* out_dev,
                 int B, int H, int W, int Hu, int Wu, int y0, int y1, int sin_mode) {
    if (!workspace_dev) return DIINN_ERR_INVALID_ARG;
    int r0, r1;
    int st = diinn_lr_rows_for_band(H, Hu, Wu, y0, y1, &r0, &r1);
    if (st) return st;
    st = diinn_precompute_P(stream, feat_dev, packed_dev, workspace_dev, B, H, W, r0, r1);
    if (st) return st;
    return diinn_decode_band(stream, workspace_dev, packed_dev, out_dev, B, H, W, Hu, Wu, y0, y1, sin_mode);
}

}  // extern "C"
