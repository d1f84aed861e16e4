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
#ifdef W4_STAMPS
    hipMalloc(&g_w4_stamps, 2 * 80 * 4 * 8);
    hipMemset(g_w4_stamps, 0, 2 * 80 * 4 * 8);
#endif
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
#ifdef W4_STAMPS
    {   // phases of workgroup 0 (the XCD map puts work item 0 there) in the last launch (Cin = 512, 64 iterations), in shader cycles
        std::vector<unsigned long long> st(2 * 80 * 4);
        hipMemcpy(st.data(), g_w4_stamps, st.size() * 8, hipMemcpyDeviceToHost);
        double m[4] = {0, 0, 0, 0}, pr[3] = {0, 0, 0};
        const int i0 = 8, i1 = 56;
        for (int i = i0; i < i1; ++i) {
            const unsigned long long* a = &st[(0 * 80 + i) * 4];
            const unsigned long long* an = &st[(0 * 80 + i + 1) * 4];
            m[0] += a[1] - a[0]; m[1] += a[2] - a[1]; m[2] += a[3] - a[2]; m[3] += an[0] - a[0];
            const unsigned long long* q = &st[(1 * 80 + i) * 4];
            const unsigned long long* qn = &st[(1 * 80 + i + 1) * 4];
            pr[0] += q[1] - q[0]; pr[1] += q[2] - q[1]; pr[2] += qn[0] - q[0];
        }
        const unsigned long long* f = &st[(0 * 80 + 76) * 4];
        printf("  workgroup 0, wave 0 (Cin = 512 launch): prologue %llu cycles, loop %llu, epilogue %llu\n", f[1] - f[0], f[2] - f[1], f[3] - f[2]);
        const double k = 1.0 / (i1 - i0);
        printf("  stamps (s_memtime ticks per iteration, 100 MHz?): MFMA wave 0: fetch issue %.0f, MFMAs+requests %.0f, barrier wait %.0f, iteration %.0f | transform wave: transform %.0f, barrier wait %.0f, iteration %.0f\n",
               m[0] * k, m[1] * k, m[2] * k, m[3] * k, pr[0] * k, pr[1] * k, pr[2] * k);
    }
#endif
    return 0;
}
