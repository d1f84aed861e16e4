// Does a range-checked LDS-DMA (buffer_load ... lds) write ZERO into LDS for lanes whose offset is out of range?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_oor.hip -o lds_dma_oor && ./lds_dma_oor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* out, int n) {
    __shared__ float lds[64 * 4 + 64];
    for (int i = threadIdx.x; i < 320; i += 64) lds[i] = 7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    const unsigned oor = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16,
                                             (threadIdx.x & 1) ? oor : threadIdx.x * 16, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 4,
                                             (threadIdx.x % 3) ? oor - 4u : threadIdx.x * 4, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 320; i += 64) out[i] = lds[i];
}
int main() {
    float *in, *out, h[320], src[256];
    for (int i = 0; i < 256; ++i) src[i] = 100.0f + i;
    hipMalloc(&in, 1024); hipMalloc(&out, 1280);
    hipMemcpy(in, src, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(in, out, 256);
    hipMemcpy(h, out, 1280, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 4; ++j) { float want = (l & 1) ? 0.0f : 100.0f + 4 * l + j; if (h[4 * l + j] != want) { if (bad++ < 8) printf("x4 lane %d comp %d: %g want %g\n", l, j, h[4 * l + j], want); } }
        float want = (l % 3) ? 0.0f : 100.0f + l; if (h[256 + l] != want) { if (bad++ < 16) printf("x1 lane %d: %g want %g\n", l, h[256 + l], want); }
    }
    printf("LDS-DMA out-of-range lanes write zero: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return 0;
}
