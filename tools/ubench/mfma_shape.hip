// mfma_shape.hip -- r04: the inner structure of decode_bf16_coop8p_kernel (8 waves per CU, two per SIMD; a layer's
// weights of the wave in 128 registers, B fragments re-read from LDS, optional relu*sin epilogue + LDS write-back) on
// v_mfma_f32_32x32x16_bf16 versus v_mfma_f32_16x16x32_bf16, random operands, equal FLOPs.  Question (VERDICT r03 item 1a,
// MI355X_MICROARCH.md "DVFS give-back" (7)): does the 16x16x32 shape run the same work faster once the chip limits its clock?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float epi(float k, float s) {
    return __builtin_elementwise_maximum(k, 0.0f) * __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(s));
}

// SHAPE 0: 32x32x16, the wave owns one 32-row M-tile of both branches (k, s): per 32-pixel tile 16 k-steps x 2 MFMAs.
// SHAPE 1: 16x16x32, the wave owns four 16-row M-tiles (k lo/hi, s lo/hi): per 16-pixel tile 8 k-steps x 4 MFMAs.
// EPI: 0 none (accumulators summed), 1 relu*sin + pack + LDS write (as the real layer)
template <int SHAPE, int EPI>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ w, float* sink, int blocks, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) f32x4 qa[2][4 * 16 * 64];     // 2 x 64 KiB: B fragments [tile][k-step][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2;
    for (int i = threadIdx.x; i < 2 * 4 * 16 * 64; i += 512) {
        const f32x4 v = *(const f32x4*)(w + 4 * ((i * 37 + blockIdx.x) & 16383));
        (&qa[0][0])[i] = v;
    }
    f32x4 A[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) A[i] = *(const f32x4*)(w + 4 * ((wave * 32 + i) * 64 + lane));
    __syncthreads();
    float acc_sum = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int b = 0; b < blocks; ++b) {
#pragma unroll 1
        for (int layer = 0; layer < 3; ++layer) {
            const int cur = layer & 1;
            const f32x4* __restrict__ qin = &qa[cur][lane];
            f32x4* __restrict__ qout = &qa[1 - cur][lane];
            if constexpr (SHAPE == 0) {
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    const int t = ti ^ (2 * grp);
                    f32x16 ak, as;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { ak[r] = 0.01f * r; as[r] = 0.02f * r; }
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const bf16x8 bv = __builtin_bit_cast(bf16x8, qin[(t * 16 + ks) * 64]);
                        ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[2 * ks]), bv, ak, 0, 0, 0);
                        as = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[2 * ks + 1]), bv, as, 0, 0, 0);
                    }
                    if constexpr (EPI) {
                        u32x4 fr;
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const f32x2 v = {epi(ak[r], as[r]), epi(ak[r + 1], as[r + 1])};
                            fr[(r >> 1) & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                            if ((r & 7) == 6) qout[(t * 16 + 2 * wave + (r >> 3)) * 64] = __builtin_bit_cast(f32x4, fr);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc_sum += ak[r] + as[r];
                    }
                }
            } else {
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) {                  // 16-pixel tiles; B fragment (tile, k-step) at [ti*8 + ks]
                    const int t = ti ^ (4 * grp);
                    f32x4 c[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) c[m] = f32x4{0.01f * m, 0.02f, 0.03f, 0.04f};
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const bf16x8 bv = __builtin_bit_cast(bf16x8, qin[(t * 8 + ks) * 64]);
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[4 * ks + m]), bv, c[m], 0, 0, 0);
                    }
                    if constexpr (EPI) {
                        u32x4 fr;
#pragma unroll
                        for (int i = 0; i < 4; i += 2) {
                            const f32x2 lo = {epi(c[0][i], c[2][i]), epi(c[0][i + 1], c[2][i + 1])};
                            const f32x2 hi = {epi(c[1][i], c[3][i]), epi(c[1][i + 1], c[3][i + 1])};
                            fr[i >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
                            fr[2 + (i >> 1)] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
                        }
                        qout[(t * 8 + wave) * 64] = __builtin_bit_cast(f32x4, fr);
                    } else {
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc_sum += c[m][0] + c[m][1] + c[m][2] + c[m][3];
                    }
                }
            }
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc_sum == 1234.5f) sink[0] = acc_sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}


// SHAPE 2 (pixel-stationary): every wave owns 32 pixels whose activation stays in registers (16 B fragments); the layer's
// weights stream through a 3-stage LDS ring (stage = one M-tile of both branches = 32 KiB) that all 8 waves read; each wave
// fetches 1/8 of a stage two stages ahead (global -> registers -> LDS); one barrier per stage.  No activation ever touches LDS.
template <int EPI>
__global__ __launch_bounds__(512, 2) void k2(const float* __restrict__ w, float* sink, int blocks, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) f32x4 ring[3][32 * 64];       // 3 x 32 KiB: [stage][piece 2*ks+part][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 65536 * 4, 0x00020000);
    bf16x8 qb[16], qn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) qb[i] = __builtin_bit_cast(bf16x8, *(const f32x4*)(w + 4 * ((i * 64 + lane + 7 * wave) & 16383)));
    f32x4 g[4];
    auto fetch = [&](int st) {                                   // this wave's four pieces of stage st (of 24 per block)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            g[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((st * 32 + wave * 4 + i) * 1024) & 0x3ffff, 0));
    };
    auto put = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[slot][(wave * 4 + i) * 64 + lane] = g[i];
    };
    fetch(0); put(0); fetch(1);
    __syncthreads();
    float acc_sum = 0.0f;
    int st = 0;                                                   // running stage number
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int b = 0; b < blocks; ++b) {
#pragma unroll 1
        for (int layer = 0; layer < 3; ++layer) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                // top of stage: publish stage st+1 (fetched during the previous stage), start fetching st+2
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                put((m + 1) % 3 == 0 ? 0 : (m + 1) % 3 == 1 ? 1 : 2);     // slot (st+1)%3, with 24 stages per block = 0 mod 3
                fetch(st + 2);
                const f32x4* __restrict__ A = &ring[m % 3][lane];
                f32x16 ak, as;
#pragma unroll
                for (int r = 0; r < 16; ++r) { ak[r] = 0.01f * r; as[r] = 0.02f * r; }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[(2 * ks) * 64]), qb[ks], ak, 0, 0, 0);
                    as = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[(2 * ks + 1) * 64]), qb[ks], as, 0, 0, 0);
                }
                if constexpr (EPI) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) qn[2 * m + (r >> 3)][r & 7] = (__bf16)epi(ak[r], as[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_sum += ak[r] + as[r];
                }
                ++st;
            }
            if constexpr (EPI) {
#pragma unroll
                for (int i = 0; i < 16; ++i) qb[i] = qn[i];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if constexpr (EPI) for (int i = 0; i < 16; ++i) acc_sum += (float)qb[i][0];
    if (acc_sum == 1234.5f) sink[0] = acc_sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int EPI>
static void run(const char* name, const float* w, float* sink, unsigned long long* cyc, int blocks) {
    hipLaunchKernelGGL((k<SHAPE, EPI>), dim3(256), dim3(512), 0, 0, w, sink, 20, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, tot = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, EPI>), dim3(256), dim3(512), 0, 0, w, sink, blocks, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; tot += ms;
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto x : h) c += (double)x; c /= 256;
    // one block = 3 layers x 8 waves x 4 tiles x 32 MFMA(32x32x16) = 3072 MFMA-equivalents = 100.7 MFLOP; ideal cycles/block/SIMD = 24576
    const double flop = 256.0 * blocks * 3072.0 * 32768.0;
    printf("%-34s best %.3f ms  mean %.3f ms  %.0f TFLOP/s  counter(100MHz?) per block %.0f  -> ns/block %.1f\n", name, best, tot / 5,
           flop / (best * 1e-3) / 1e12, c / blocks, best * 1e6 / blocks);
}

template <int EPI>
static void run2(const char* name, const float* w, float* sink, unsigned long long* cyc, int blocks) {
    hipLaunchKernelGGL((k2<EPI>), dim3(256), dim3(512), 0, 0, w, sink, 20, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, tot = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k2<EPI>), dim3(256), dim3(512), 0, 0, w, sink, blocks, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; tot += ms;
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto x : h) c += (double)x; c /= 256;
    // one "block" here = 8 waves x 32 px = 256 px x 3 layers = 2 x the MFMAs of the 128-px block above
    const double flop = 256.0 * blocks * 2 * 3072.0 * 32768.0;
    printf("%-34s best %.3f ms  mean %.3f ms  %.0f TFLOP/s  counter per 128 px %.0f  -> ns per 128 px %.1f\n", name, best, tot / 5,
           flop / (best * 1e-3) / 1e12, c / blocks / 2, best * 1e6 / blocks / 2);
}

int main() {
    float *w, *sink; unsigned long long* cyc;
    std::vector<float> hw(65536 + 64);
    srand(1);
    for (auto& x : hw) { unsigned short hi = 0x3f00 + (rand() & 0xff), lo = 0x3f00 + (rand() & 0xff); unsigned u = ((unsigned)hi << 16) | lo; if (rand() & 1) u ^= 0x80008000u; memcpy(&x, &u, 4); }
    hipMalloc(&w, hw.size() * 4); hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 300;                      // ~ the real kernel's blocks per persistent workgroup at c5
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("32x32x16  mfma + lds reads", w, sink, cyc, blocks);
        run<1, 0>("16x16x32  mfma + lds reads", w, sink, cyc, blocks);
        run<0, 1>("32x32x16  + epilogue", w, sink, cyc, blocks);
        run<1, 1>("16x16x32  + epilogue", w, sink, cyc, blocks);
        run2<0>("pixel-stationary, weights via LDS", w, sink, cyc, blocks / 2);
        run2<1>("pixel-stationary + epilogue", w, sink, cyc, blocks / 2);
    }
    return 0;
}
