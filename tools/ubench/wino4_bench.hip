// Times conv_wino4_kernel alone (random data, no checking) for timing ablations of its phases:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include [-DW4_ABL_...] tools/ubench/wino4_bench.hip -o wino4_bench
//   ./wino4_bench [SIZE=256]
#include "../../dual-interactive-implicit-neural-network_amd/csrc/diinn_winograd4.hip"
thread_local int g_last_hip_error = 0;
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 256;
    const size_t hw = (size_t)S * S;
    float *in, *w, *bias, *out;
    hipMalloc(&in, 576 * hw * 4); hipMalloc(&w, (size_t)36 * 64 * 512 * 4); hipMalloc(&bias, 256); hipMalloc(&out, 64 * hw * 4);
    std::vector<float> h(576 * hw);
    for (auto& v : h) v = (float)rand() / RAND_MAX;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    h.resize((size_t)36 * 64 * 512);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cin : {64, 128, 512}) {
        for (int i = 0; i < 3; ++i) diinn_conv_wino4(nullptr, in, 576 * hw, cin, w, bias, nullptr, 0, out, 64 * hw, 1, 1, S, S);
        hipEventRecord(e0);
        const int n = 20;
        for (int i = 0; i < n; ++i) diinn_conv_wino4(nullptr, in, 576 * hw, cin, w, bias, nullptr, 0, out, 64 * hw, 1, 1, S, S);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  Cin %3d: %7.1f us", cin, ms / n * 1e3);
    }
    printf("\n");
    return 0;
}
