// Microbenchmark: v_mfma_f32_32x32x2_f32 issue rate with NACC interleaved accumulators,
// one wave per SIMD (256 threads/WG, 1 WG per CU), operands in registers, plus EV VALU ops
// (spread over CH independent chains) between every group of NACC MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NACC, int EV, int CH, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    float v[CH];
    for (int c = 0; c < CH; ++c) v[c] = a0 + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = MFMA32(a, b, acc[i]);
#pragma unroll
            for (int e = 0; e < EV; ++e) v[e % CH] = __builtin_fmaf(v[e % CH], 1.0001f, 0.5f);
        }
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += v[c];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int NACC, int EV, int CH, int THREADS = 256>
void run() {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4 * 8);
    const int iters = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NACC, EV, CH, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NACC, EV, CH, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double mfmas = 256.0 * (THREADS / 64) * iters * 64 * NACC;
    double tf = mfmas * 4096 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD=%d NACC=%d valu/MFMA=%.2f chains=%d : %.3f ms %.1f TF (%.1f%%)  cyc/MFMA@2.4GHz=%.1f\n", THREADS / 256, NACC, (double)EV / NACC, CH, ms, tf,
           tf / 157.3 * 100, ms * 1e-3 * 2.4e9 / (iters * 64.0 * NACC * (THREADS / 256)));
    (void)hipFree(out);
}

int main() {
    run<1, 0, 1>(); run<2, 0, 1>(); run<4, 0, 1>(); run<8, 0, 1>();
    run<2, 2, 2>(); run<2, 4, 4>(); run<2, 8, 8>(); run<2, 16, 8>(); run<2, 32, 8>(); run<2, 64, 8>();
    run<2, 8, 1>(); run<4, 16, 8>(); run<4, 32, 8>();
    // two waves per SIMD (512-thread workgroups)
    run<1, 0, 1, 512>(); run<2, 0, 1, 512>(); run<2, 2, 2, 512>(); run<2, 4, 4, 512>(); run<2, 8, 8, 512>();
    return 0;
}
