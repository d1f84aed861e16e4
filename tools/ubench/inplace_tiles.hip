// inplace_tiles.hip -- r04: would a bigger block pay for decode_bf16_coop8p_kernel?  The kernel's layer structure (8 waves
// per CU, a wave = one M-tile of both branches in 128 registers, B fragments from LDS, relu*sin epilogue written back as
// the next layer's B fragments, the layer's weights RE-LOADED from global memory per layer and block as the last tile
// consumes the old ones) in three forms, equal FLOPs, random operands:
//   TILES = 4, double-buffered images (2 x 64 KiB): the shipped structure, one barrier per layer;
//   TILES = 6 / 8, ONE image updated IN PLACE (96 / 128 KiB): a wave overwrites a tile's slots only after all eight waves
//     have read them -- per-tile arrival counters in LDS, polled -- and the layer still ends with one barrier.
// Reports ns per 128 pixels.   hipcc --offload-arch=gfx950 -O3 -w -o inplace_tiles inplace_tiles.hip
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
__device__ __forceinline__ f32x4 ldw(__amdgpu_buffer_rsrc_t rs, int lane, int piece) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (piece * 1024) & 0xfffff, 0));
}

template <int TILES, bool INPLACE>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ w, float* sink, int blocks, unsigned long long* cyc) {
    constexpr int IMG = TILES * 16 * 64;                                  // f32x4 elements per image
    __shared__ __attribute__((aligned(16))) f32x4 qa[(INPLACE ? 1 : 2) * IMG];
    __shared__ unsigned cnt[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2;
    for (int i = threadIdx.x; i < (INPLACE ? 1 : 2) * IMG; i += 512) qa[i] = *(const f32x4*)(w + 4 * ((i * 37 + blockIdx.x) & 16383));
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 20, 0x00020000);
    f32x4 A[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) A[i] = ldw(rs, lane, wave * 32 + i);
    __syncthreads();
    unsigned target = 0;
    int wbase = 256 + wave * 32;                                          // piece number of the next layer's fragments
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int b = 0; b < blocks; ++b) {
#pragma unroll 1
        for (int layer = 0; layer < 3; ++layer) {
            const int cur = INPLACE ? 0 : (layer & 1);
            const f32x4* __restrict__ qin = &qa[cur * IMG + lane];
            f32x4* __restrict__ qout = &qa[(INPLACE ? 0 : 1 - cur) * IMG + lane];
            target += 8;
#pragma unroll (INPLACE ? 1 : TILES)
            for (int ti = 0; ti < TILES; ++ti) {
                const int t = INPLACE ? ti : (ti ^ (2 * grp)) % TILES;    // in place: both groups walk the tiles in one order
                f32x16 ak, as;
#pragma unroll
                for (int r = 0; r < 16; ++r) { ak[r] = 0.01f * r; as[r] = 0.02f * r; }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const bf16x8 bv = __builtin_bit_cast(bf16x8, qin[(t * 16 + ks) * 64]);
                    ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[2 * ks]), bv, ak, 0, 0, 0);
                    as = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[2 * ks + 1]), bv, as, 0, 0, 0);
                    if (ti == TILES - 1) {                                // last use: the next layer's fragments
                        A[2 * ks] = ldw(rs, lane, wbase + 2 * ks);
                        A[2 * ks + 1] = ldw(rs, lane, wbase + 2 * ks + 1);
                    }
                }
                if (INPLACE) {                                            // this wave has read tile t (its reads are in registers)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add(&cnt[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                u32x4 fr0, fr1;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = {epi(ak[r], as[r]), epi(ak[r + 1], as[r + 1])};
                    const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                    if (r < 8) fr0[(r >> 1) & 3] = pk; else fr1[(r >> 1) & 3] = pk;
                }
                if (INPLACE) {                                            // all eight waves must have read tile t
                    while (__hip_atomic_load(&cnt[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
                }
                qout[(t * 16 + 2 * wave) * 64] = __builtin_bit_cast(f32x4, fr0);
                qout[(t * 16 + 2 * wave + 1) * 64] = __builtin_bit_cast(f32x4, fr1);
            }
            wbase = (wbase + 256) & 1023;
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.0f;
    for (int i = 0; i < 32; ++i) acc += A[i][0];
    if (acc == 1234.5f) sink[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int TILES, bool INPLACE>
static void run(const char* name, const float* w, float* sink, unsigned long long* cyc) {
    const int blocks = 1200 / TILES;                                      // the same pixels for every form
    hipLaunchKernelGGL((k<TILES, INPLACE>), dim3(256), dim3(512), 0, 0, w, sink, 8, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, tot = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<TILES, INPLACE>), dim3(256), dim3(512), 0, 0, w, sink, blocks, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; tot += ms;
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto x : h) c += (double)x; c /= 256;
    const double px128 = blocks * TILES / 4.0;
    printf("%-44s best %.3f ms  mean %.3f ms  cycles per 128 px %.0f  ns per 128 px %.1f\n", name, best, tot / 5, c / px128, best * 1e6 / px128);
}

int main() {
    float *w, *sink; unsigned long long* cyc;
    std::vector<float> hw((1 << 18) + 64);
    srand(1);
    for (auto& x : hw) { unsigned short hi = 0x3f00 + (rand() & 0xff), lo = 0x3f00 + (rand() & 0xff); unsigned u = ((unsigned)hi << 16) | lo; if (rand() & 1) u ^= 0x80008000u; memcpy(&x, &u, 4); }
    hipMalloc(&w, hw.size() * 4); hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<4, false>("4 tiles, two images (shipped structure)", w, sink, cyc);
        run<4, true>("4 tiles, in place (counters)", w, sink, cyc);
        run<6, true>("6 tiles, in place (counters)", w, sink, cyc);
        run<8, true>("8 tiles, in place (counters)", w, sink, cyc);
    }
    return 0;
}
