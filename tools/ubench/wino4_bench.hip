// Times conv_wino4_kernel alone (random data) and checks it against a naive reference; round 5: with and without the split of
// the last round over the input channels (DIINN_ENC_WINO4_SPLIT), per map size.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I include [-DW4_STAMPS] tools/ubench/wino4_bench.hip \
//         dual-interactive-implicit-neural-network_amd/csrc/diinn_host.cpp -o tools/ubench/wino4_bench
//   ./wino4_bench [SIZE ...]            (default 256; W4_CHECK=1: reference + determinism checks of the split form)
// The round-4 timing ablations (-DW4_ABL_*, -DW4_SYM_KERNEL ...): -DW4_R04 builds the frozen round-4 kernel text
// (tools/ubench/wino4_r04_ablate.hip) with the round-4 harness below it.
#ifdef W4_SYM_KERNEL
#include "wino4_sym_kernel.hip"
#define W4_R04
#elif defined(W4_R04)
#include "wino4_r04_ablate.hip"
#else
#include "../../dual-interactive-implicit-neural-network_amd/csrc/diinn_winograd4.hip"
int device_cus() { int n = 256; hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, 0); if (getenv("W4_NCU")) n = atoi(getenv("W4_NCU")); return n; }
#endif
thread_local int g_last_hip_error = 0;
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

// naive reference in the Winograd domain (the harness feeds random TRANSFORMED weights): one thread per (tile, output channel)
__global__ void w4_reference(const float* in, const float* wu, float* out, int Cin, int S) {
    const int tiles_x = S / 4, t = blockIdx.x * blockDim.x + threadIdx.x, co = blockIdx.y;
    if (t >= tiles_x * tiles_x) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const size_t hw = (size_t)S * S;
    const int n = Cin / 8, half = co >> 5, m = co & 31;
    static const float BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
    static const float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
    double M[36];
    for (int i = 0; i < 36; ++i) M[i] = 0;
    for (int c = 0; c < Cin; ++c) {
        float d[6][6], tmp[6][6];
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) {
                const int y = 4 * ty - 1 + a, x = 4 * tx - 1 + b;
                d[a][b] = (y >= 0 && y < S && x >= 0 && x < S) ? in[c * hw + (size_t)y * S + x] : 0.0f;
            }
        for (int i = 0; i < 6; ++i)
            for (int b = 0; b < 6; ++b) { float v = 0; for (int a = 0; a < 6; ++a) v += BT[i][a] * d[a][b]; tmp[i][b] = v; }
        const int chunk = c >> 3, e = (c & 7) >> 1, h = c & 1;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                float v = 0;
                for (int b = 0; b < 6; ++b) v += tmp[i][b] * BT[j][b];
                const int pos = 6 * i + j, wave = pos / 3, q = pos % 3;
                const float u = wu[((((size_t)(wave * 2 + half) * n + chunk) * 3 + q) * 64 + (h * 32 + m)) * 4 + e];
                M[pos] += (double)u * v;
            }
    }
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            double v = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) v += AT[a][i] * M[6 * i + j] * AT[b][j];
            out[co * hw + (size_t)(4 * ty + a) * S + 4 * tx + b] = v > 0 ? (float)v : 0.0f;
        }
}

#ifndef W4_R04
int main(int argc, char** argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes.push_back(256);
    const bool check = getenv("W4_CHECK") != nullptr;
    int smax = 0;
    for (int s : sizes) smax = s > smax ? s : smax;
    const size_t hwmax = (size_t)smax * smax;
    float *in, *w, *bias, *out, *out2, *ws;
    const size_t wsf = diinn_conv_wino4_workspace_floats();
    hipMalloc(&in, 576 * hwmax * 4); hipMalloc(&w, (size_t)36 * 64 * 512 * 4); hipMalloc(&bias, 256);
    hipMalloc(&out, 64 * hwmax * 4); hipMalloc(&out2, 64 * hwmax * 4); hipMalloc(&ws, wsf * 4);
    hipMemset(ws, 0, wsf * 4);
    std::vector<float> h(576 * hwmax);
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
    auto run = [&](int S, int cin, float* o) {
        return diinn_conv_wino4_ws(nullptr, in, 576 * (size_t)S * S, cin, w, bias, nullptr, 0, o, 64 * (size_t)S * S, 1, 1, S, S, ws, wsf);
    };
    auto time_us = [&](int S, int cin) {
        for (int i = 0; i < 3; ++i) run(S, cin, out);
        hipEventRecord(e0);
        const int n = 20;
        for (int i = 0; i < n; ++i) run(S, cin, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms / n * 1e3;
    };
    for (int S : sizes) {
        const size_t hw = (size_t)S * S;
        printf("%dx%d\n", S, S);
        double sum[3] = {0, 0, 0};
        for (int cin = 64; cin <= 512; cin += 64) {
            int info[4];
            diinn_debug_set("DIINN_ENC_WINO4_SPLIT", 2);
            diinn_conv_wino4_plan(cin, 1, S, S, 1, info);
            printf("  Cin %3d: items %d (whole %d, split workgroups %d x %d chunks):", cin, info[0], info[1], info[2], info[3]);
            for (int mode = 0; mode < 3; ++mode) {
                if (getenv("W4_ONLY_MODE") && atoi(getenv("W4_ONLY_MODE")) != mode) continue;
                diinn_debug_set("DIINN_ENC_WINO4_SPLIT", mode);
                const double t = time_us(S, cin);
                sum[mode] += t;
                printf("  %s %7.1f us", mode == 0 ? "whole" : mode == 1 ? "model" : "split", t);
            }
            if (check && S % 4 == 0) {
                diinn_debug_set("DIINN_ENC_WINO4_SPLIT", 2);
                run(S, cin, out);
                std::vector<float> o(64 * hw), r(64 * hw), o2(64 * hw);
                hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
                hipLaunchKernelGGL(w4_reference, dim3((S / 4 * (S / 4) + 63) / 64, 64), dim3(64), 0, nullptr, in, w, out2, cin, S);
                hipMemcpy(r.data(), out2, r.size() * 4, hipMemcpyDeviceToHost);
                double mx = 0, worst_e = 0; size_t nbad = 0;
                for (size_t i = 0; i < r.size(); ++i) if (fabs(r[i]) > mx) mx = fabs(r[i]);
                for (size_t i = 0; i < r.size(); ++i) { const double e_ = fabs((double)o[i] - r[i]); if (e_ > worst_e) worst_e = e_; if (e_ > 1e-3 * mx) ++nbad; }
                printf("  [split vs naive: max err %.2e of %.2e, %zu bad]", worst_e, mx, nbad);
                size_t worst = 0;
                for (int rep = 0; rep < 20; ++rep) {
                    run(S, cin, out2);
                    hipMemcpy(o2.data(), out2, o2.size() * 4, hipMemcpyDeviceToHost);
                    size_t nd = 0;
                    for (size_t i = 0; i < o.size(); ++i) nd += o[i] != o2[i];
                    if (nd > worst) worst = nd;
                }
                std::vector<unsigned> cnt(1024);
                hipMemcpy(cnt.data(), ws, 4096, hipMemcpyDeviceToHost);
                size_t nz = 0;
                for (unsigned c : cnt) nz += c != 0;
                printf(" [repeats: worst %zu differing; counters left non-zero: %zu]", worst, nz);
            }
            printf("\n");
        }
        printf("  dense block (Cin 64 .. 512): whole %.1f us, model %.1f us, split %.1f us\n", sum[0], sum[1], sum[2]);
    }
#ifdef W4_STAMPS
    {
        std::vector<unsigned long long> st(2 * 80 * 4);
        hipMemcpy(st.data(), g_w4_stamps, st.size() * 8, hipMemcpyDeviceToHost);
        const unsigned long long* f = &st[(0 * 80 + 76) * 4];
        const unsigned long long* g7 = &st[(0 * 80 + 77) * 4];
        const unsigned long long* g8 = &st[(0 * 80 + 78) * 4];
        printf("  workgroup 0's LAST task (last launch; s_memtime ticks = shader cycles): prologue %llu, loop %llu, exchange + A^T.A %lld | ticket %lld, "
               "slab store + drain + count %lld | last arriver: wait %lld, sum %lld, final stores (from sum) %lld\n",
               f[1] - f[0], f[2] - f[1], (long long)(g7[0] - f[2]), (long long)(g7[1] - g7[0]), (long long)(g7[2] - g7[1]),
               (long long)(g8[0] - g7[1]), (long long)(g8[1] - g8[0]), (long long)(f[3] - g8[1]));
        const unsigned long long* rt = &st[(0 * 80 + 79) * 4];
        printf("  shader clock over workgroup 0's loop (last launch): %llu cycles / %llu ticks of the 100 MHz counter = %.3f GHz\n",
               f[2] - f[1], rt[2] - rt[1], (double)(f[2] - f[1]) / (double)(rt[2] - rt[1]) * 0.1);
    }
#endif
    return 0;
}
#else
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
        if (cin == 512 && getenv("W4_CHECKSUM")) {                // compare builds: same seeded input and weights
            std::vector<float> o(64 * hw);
            hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
            double s1 = 0, s2 = 0;
            for (size_t i = 0; i < o.size(); ++i) { s1 += o[i]; s2 += (double)o[i] * o[i] * (1 + (i % 7)); }
            printf("  [checksum %.6e %.6e]", s1, s2);
            float* out2; hipMalloc(&out2, 64 * hw * 4);
            if (S % 4 == 0) {
                hipLaunchKernelGGL(w4_reference, dim3((S / 4 * (S / 4) + 63) / 64, 64), dim3(64), 0, nullptr, in, w, out2, cin, S);
                std::vector<float> r(64 * hw);
                hipMemcpy(r.data(), out2, r.size() * 4, hipMemcpyDeviceToHost);
                double mx = 0, worst_e = 0; size_t nbad = 0;
                for (size_t i = 0; i < r.size(); ++i) { if (fabs(r[i]) > mx) mx = fabs(r[i]); }
                for (size_t i = 0; i < r.size(); ++i) { const double e_ = fabs((double)o[i] - r[i]); if (e_ > worst_e) worst_e = e_; if (e_ > 1e-3 * mx) ++nbad; }
                printf("  [vs naive reference: max err %.3e of max %.3e, %zu outputs beyond 1e-3]", worst_e, mx, nbad);
            }
            size_t worst = 0;
            for (int rep = 0; rep < 30; ++rep) {
                diinn_conv_wino4(nullptr, in, 576 * hw, cin, w, bias, nullptr, 0, out2, 64 * hw, 1, 1, S, S);
                std::vector<float> o2(64 * hw);
                hipMemcpy(o2.data(), out2, o2.size() * 4, hipMemcpyDeviceToHost);
                size_t nd = 0, firstd = 0;
                for (size_t i = 0; i < o.size(); ++i) if (o[i] != o2[i]) { if (!nd) firstd = i; ++nd; }
                if (nd) printf("\n    rep %d: %zu outputs differ from the first result, first at channel %zu row %zu col %zu", rep, nd, firstd / hw, (firstd % hw) / S, firstd % S);
                if (nd > worst) worst = nd;
            }
            printf("  [in-process repeats: worst %zu differing outputs]", worst);
        }
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

#endif
