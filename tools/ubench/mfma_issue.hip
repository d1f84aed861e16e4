// mfma_issue.hip -- what a single wave per SIMD can issue beside six bf16 MFMAs (the k-step of decode_bf16x3_kernel):
// clocks per k-step for variants of the body.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

template <int NLOAD, int NLDS, int NVALU, int DEP, int NSIN>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ w, unsigned long long* out, int iters, float* sink) {
    __shared__ f32x4 lds[32][64];
    __shared__ float pad[24 * 1024];            // 128 KiB total: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) lds[i >> 6][i & 63] = f32x4{1.f, 2.f, 3.f, 4.f};
    pad[threadIdx.x] = 0.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 20, 0x00020000);
    f32x16 ak = {}, as = {};
    f32x4 A[4] = {f32x4{1, 1, 1, 1}, f32x4{1, 1, 1, 1}, f32x4{1, 1, 1, 1}, f32x4{1, 1, 1, 1}};
    f32x4 R[8];
    for (int i = 0; i < 8; ++i) R[i] = f32x4{0, 0, 0, 0};
    bf16x8 qh = __builtin_bit_cast(bf16x8, f32x4{1, 1, 1, 1}), ql = qh;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane * 0.001f + i;
    int off = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {           // four k-steps per iteration
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, A[0] + R[(2 * u) & 7]), a1 = __builtin_bit_cast(bf16x8, A[1] + R[(2 * u + 1) & 7]);
            ak = MFMA(a0, qh, ak);
            as = MFMA(a1, qh, as);
            ak = MFMA(a0, ql, ak);
            as = MFMA(a1, ql, as);
            ak = MFMA(a1, qh, ak);
            as = MFMA(a0, qh, as);
#pragma unroll
            for (int i = 0; i < NLOAD; ++i)
                R[(4 * u + i) & 7] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, off + (4 * u + i) * 1024, 0));
#pragma unroll
            for (int i = 0; i < NLDS; ++i) R[(4 * u + i) & 7] = lds[(4 * u + i + off) & 31][lane];
            if (DEP) {
#pragma unroll
                for (int i = 0; i < NVALU; ++i) v[0] = __builtin_fmaf(v[0], 1.0001f, 0.5f);
            } else {
#pragma unroll
                for (int i = 0; i < NVALU; ++i) v[i & 15] = __builtin_fmaf(v[i & 15], 1.0001f, 0.5f);
            }
#pragma unroll
            for (int i = 0; i < NSIN; ++i) v[i & 15] = __builtin_amdgcn_sinf(v[i & 15]);
            __builtin_amdgcn_sched_barrier(0);
        }
        off = (off + 16 * 1024) & (256 * 1024 - 1);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 16; ++i) sum += ak[i] + as[i] + v[i];
    for (int i = 0; i < 8; ++i) sum += R[i][0];
    if (sum == 1234.5f) sink[0] = sum;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NLOAD, int NLDS, int NVALU, int DEP, int NSIN>
void run(const char* name, const float* w, unsigned long long* out, float* sink, int grid) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NLOAD, NLDS, NVALU, DEP, NSIN>), dim3(grid), dim3(256), 0, 0, w, out, iters, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NLOAD, NLDS, NVALU, DEP, NSIN>), dim3(grid), dim3(256), 0, 0, w, out, iters, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    // readcyclecounter = s_memtime: constant 100 MHz? report both the counter and the event time
    printf("%-34s grid %3d  counter/k-step %8.2f   event ns/k-step %7.2f\n", name, grid, s / h.size() / (iters * 4.0), ms * 1e6 / (iters * 4.0));
}


int main() {
    float *w, *sink; unsigned long long* out;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20); hipMalloc(&sink, 64); hipMalloc(&out, 256 * 4 * 8);
    for (int grid : {1, 256}) {
        run<0, 0, 0, 0, 0>("6 MFMA", w, out, sink, grid);
        run<4, 0, 0, 0, 0>("6 MFMA + 4 vmem b128", w, out, sink, grid);
        run<2, 0, 0, 0, 0>("6 MFMA + 2 vmem b128", w, out, sink, grid);
        run<0, 4, 0, 0, 0>("6 MFMA + 4 ds_read_b128", w, out, sink, grid);
        run<0, 0, 16, 0, 0>("6 MFMA + 16 valu indep", w, out, sink, grid);
        run<0, 0, 32, 0, 0>("6 MFMA + 32 valu indep", w, out, sink, grid);
        run<0, 0, 16, 1, 0>("6 MFMA + 16 valu chain", w, out, sink, grid);
        run<0, 0, 0, 0, 2>("6 MFMA + 2 v_sin", w, out, sink, grid);
        run<0, 0, 0, 0, 8>("6 MFMA + 8 v_sin", w, out, sink, grid);
        run<4, 0, 16, 0, 2>("6 MFMA + 4 vmem + 16 valu + 2 sin", w, out, sink, grid);
        run<0, 4, 16, 0, 2>("6 MFMA + 4 lds + 16 valu + 2 sin", w, out, sink, grid);
        run<1, 4, 16, 0, 2>("6 MFMA + 1 vmem + 4 lds + 16 valu + 2 sin", w, out, sink, grid);
    }
    return 0;
}
