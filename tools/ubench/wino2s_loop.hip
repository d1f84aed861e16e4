// Feasibility probe for a small-map F(2x2,3x3) kernel on v_mfma_f32_16x16x4_f32 (round 5): the K loop alone -- 144 workgroups (36 pixel
// tiles of 8 x 8 x 4 output quarters of a 48 x 48 map), 8 waves each reducing Cin / 8 channels, per group of 4 channels 16 MFMAs (one per
// transformed position) fed by 4 KiB of weights per wave (A operands, 16 floats per lane) -- with the B operand faked from registers: how
// fast can a CU stream 16 x 16 x Cin x 4 bytes of weights (524 KB at Cin = 512) while issuing Cin / 2 MFMAs per wave?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/wino2s_loop.hip -o tools/ubench/wino2s_loop && ./tools/ubench/wino2s_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int RING>
__global__ __launch_bounds__(512, 2) void loop_kernel(const float* w, float* out, int groups, int wgs_per_quarter_stride) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x & 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(w + (size_t)(q * 8 + wave) * groups * 1024), 0, groups * 4096, 0x00020000);
    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0, 0, 0, 0};
    f32x4 A[RING][4];
    const int lo = lane * 16;
#pragma unroll
    for (int r = 0; r < RING - 1; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) A[r][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, (r * 4 + j) * 1024, 0));
    float b = (float)lane * 0.001f;
    for (int g0 = 0; g0 < groups; g0 += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            const int g = g0 + r;
            const int gn = g + RING - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                A[(r + RING - 1) % RING][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, (gn * 4 + j) * 1024, 0));
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[p] = MFMA16(A[r][p >> 2][p & 3], b + (float)p, acc[p]);
            b += 0.5f;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 16; ++p) s += acc[p];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    float *w, *out;
    const int cin_max = 512;
    const size_t wfloats = (size_t)4 * 8 * (cin_max / 32) * 1024;
    hipMalloc(&w, wfloats * 4); hipMalloc(&out, 144 * 512 * 4 * 4);
    std::vector<float> h(wfloats);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(w, h.data(), wfloats * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {36, 144, 256, 400}) {
        for (int cin : {64, 128, 256, 512}) {
            const int groups = cin / 32;                        // per wave: (cin / 8) channels / 4
            auto run = [&](int ring) {
                if (ring == 2) hipLaunchKernelGGL(loop_kernel<2>, dim3(wgs), dim3(512), 0, nullptr, w, out, groups, 0);
                else hipLaunchKernelGGL(loop_kernel<4>, dim3(wgs), dim3(512), 0, nullptr, w, out, groups, 0);
            };
            for (int ring : {2, 4}) {
                if (groups % ring) { continue; }
                for (int i = 0; i < 5; ++i) run(ring);
                hipEventRecord(e0);
                for (int i = 0; i < 50; ++i) run(ring);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double us = ms / 50 * 1e3, bytes = 16.0 * 16 * cin * 4;
                printf("wgs %3d Cin %3d ring %d: %6.2f us per launch   (%5.1f KB of weights per workgroup: %5.1f GB/s per workgroup; MFMA floor %5.2f us at 2.4 GHz)\n",
                       wgs, cin, ring, us, bytes / 1e3, bytes / us / 1e3, (cin / 2) * 32.0 * 2 / 2400.0);
            }
        }
    }
    return 0;
}
