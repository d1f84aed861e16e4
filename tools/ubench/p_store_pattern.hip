// Microbenchmark (round 6): what does the STORE PATTERN of the bf16 P kernel cost by itself?
// precompute_P_bf16_wide_kernel writes P[cell][1024] fp32 (4 KiB per LR cell, 3.77 GB at c5) as 128-byte pieces 4 KiB
// apart: a wave owns one 32-channel M-tile at a time and a store instruction covers 8 cells x 128 B.  This program writes
// the same bytes from registers -- no arithmetic, no LDS -- in that pattern and in wider ones, with the kernel's geometry
// (workgroups of 8 rows x 32 cells, 4 waves, 2 workgroups per CU, M-tile order wave + 4 mi, rows inside an M-tile), so
// the pattern's own HBM write rate can be read off:
//   A  8 cells x 128 B per instruction  (the shipped pattern; nontemporal)
//   B  the same with plain stores
//   C  2 cells x 512 B per instruction  (the four waves' M-tiles 4 mi .. 4 mi + 3 gathered: what a cross-wave transpose gives)
//   D  1 cell x 1 KiB per instruction   (8 M-tiles of one wave gathered)
//   E  1 cell's whole 4 KiB by 4 consecutive instructions of one wave (a cell-major kernel)
// build: hipcc --offload-arch=gfx950 -O3 -o p_store_pattern p_store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 720, W = 1280, PCH = 1024;

template <int PAT>
__global__ __launch_bounds__(256, 2) void k(float* __restrict__ P, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8;
    const f32x4 v = {seed + lane, seed, seed + wave, seed + 1.0f};
    if (PAT == 0 || PAT == 1) {
        for (int mi = 0; mi < 8; ++mi) {
            const int mo = wave + 4 * mi;
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cell = 8 * i + (lane >> 3), q = lane & 7;
                    float* dst = P + ((size_t)(y0 + t) * W + x0 + cell) * PCH + 32 * mo + 4 * q;
                    if (PAT == 0) __builtin_nontemporal_store(v, (f32x4*)dst);
                    else *(f32x4*)dst = v;
                }
        }
    } else if (PAT == 2) {
        // per (mi, row): the workgroup's 32 cells x 512 B = 16 instructions, 4 per wave: wave w takes cells 8 w .. 8 w + 7
        for (int mi = 0; mi < 8; ++mi)
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cell = 8 * wave + 2 * i + (lane >> 5), q = lane & 31;
                    float* dst = P + ((size_t)(y0 + t) * W + x0 + cell) * PCH + 128 * mi + 4 * q;
                    __builtin_nontemporal_store(v, (f32x4*)dst);
                }
    } else if (PAT == 3) {
        // wave w owns M-tiles 8 w .. 8 w + 7 (1 KiB per cell), one cell per instruction: 32 instructions per row
        for (int t = 0; t < 8; ++t)
            for (int cell = 0; cell < 32; ++cell) {
                float* dst = P + ((size_t)(y0 + t) * W + x0 + cell) * PCH + 256 * wave + 4 * lane;
                __builtin_nontemporal_store(v, (f32x4*)dst);
            }
    } else {
        // cell-major: wave w takes rows 2 w, 2 w + 1; per cell 4 instructions = 4 KiB contiguous
        for (int t = 2 * wave; t < 2 * wave + 2; ++t)
            for (int cell = 0; cell < 32; ++cell)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* dst = P + ((size_t)(y0 + t) * W + x0 + cell) * PCH + 256 * i + 4 * lane;
                    __builtin_nontemporal_store(v, (f32x4*)dst);
                }
    }
}

template <int PAT>
void run(float* P, const char* what) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const dim3 grid(W / 32, H / 8);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<PAT>, grid, dim3(256), 0, 0, P, 1.0f + w);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<PAT>, grid, dim3(256), 0, 0, P, 2.0f + w);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = (double)H * W * PCH * 4;
    printf("%-62s %.3f ms  %.2f TB/s\n", what, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    float* P;
    if (hipMalloc(&P, (size_t)H * W * PCH * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    run<0>(P, "A  8 cells x 128 B per instruction, nontemporal (shipped)");
    run<1>(P, "B  8 cells x 128 B per instruction, plain");
    run<2>(P, "C  2 cells x 512 B per instruction, nontemporal");
    run<3>(P, "D  1 cell x 1 KiB per instruction, nontemporal");
    run<4>(P, "E  cell-major, 4 KiB by 4 consecutive instructions, nontemporal");
    run<0>(P, "A  again");
    (void)hipFree(P);
    return 0;
}
