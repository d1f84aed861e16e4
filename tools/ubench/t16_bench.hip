// Where does a conv_t16_kernel launch spend its time?  Builds the shipped kernel text with -DT16_STAMPS (s_memrealtime, the constant
// 100 MHz counter, per wave at: 0 body entered, 1 first group's loads landed, 2 last MFMA issued, 3 surplus loads landed, 4 every wave
// of the workgroup at the barrier, 5 stores issued, 6 stores acknowledged), launches it back to back on a 48 x 48 map and prints,
// for the last launch: the spread of the workgroups' starts, the phases (medians over waves that had work), the kernel's span and
// the gap to the launch before it.
//   hipcc -O2 -std=c++17 -w -I include -x c++ -c dual-interactive-implicit-neural-network_amd/csrc/diinn_host.cpp -o /tmp/diinn_host_ub.o
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I include -DT16_STAMPS -c tools/ubench/t16_bench.hip -o /tmp/t16_bench.o
//   hipcc --offload-arch=gfx950 /tmp/t16_bench.o /tmp/diinn_host_ub.o -o tools/ubench/t16_bench
//   ./t16_bench [SIZE=48] [CIN ...]
#include "../../dual-interactive-implicit-neural-network_amd/csrc/diinn_conv_t16.hip"
int device_cus() { int n = 256; (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, 0); return n; }
thread_local int g_last_hip_error = 0;
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double med(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 48;
    std::vector<int> cins;
    for (int i = 2; i < argc; ++i) cins.push_back(atoi(argv[i]));
    if (cins.empty()) cins = {64, 128, 256, 512};
    const size_t hw = (size_t)S * S;
    float *in, *w, *bias, *out;
    (void)hipMalloc(&in, 576 * hw * 4); (void)hipMalloc(&w, (size_t)64 * 512 * 9 * 4); (void)hipMalloc(&bias, 256); (void)hipMalloc(&out, 64 * hw * 4);
    (void)hipMemset(in, 0, 576 * hw * 4); (void)hipMemset(w, 0, (size_t)64 * 512 * 9 * 4); (void)hipMemset(bias, 0, 256);
    const int NL = 8, WGS = 256, PER = WGS * 8 * 8;
    unsigned long long* st;
    (void)hipMalloc(&st, (size_t)NL * PER * 8);
    std::vector<unsigned long long> h((size_t)NL * PER);
    for (int cin : cins) {
        (void)hipMemset(st, 0, (size_t)NL * PER * 8);
        for (int l = 0; l < NL; ++l) {
            g_t16_stamps = st + (size_t)l * PER;
            const int rc = diinn_conv_t16(nullptr, in, 576 * hw, cin, w, bias, nullptr, 0, out, 64 * hw, 1, 1, S, S);
            if (rc) { printf("diinn_conv_t16 -> %d\n", rc); return 1; }
        }
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), st, (size_t)NL * PER * 8, hipMemcpyDeviceToHost);
        auto at = [&](int l, int wg, int wave, int i) { return h[(size_t)l * PER + ((size_t)wg * 8 + wave) * 8 + i]; };
        const int L = NL - 1;
        unsigned long long s_min = ~0ull, s_max = 0, e_max = 0, prev_end = 0, prev_start = ~0ull;
        std::vector<double> ph[6], start_of;
        for (int wg = 0; wg < WGS; ++wg)
            for (int wave = 0; wave < 8; ++wave) {
                const unsigned long long t0 = at(L, wg, wave, 0);
                if (!t0) continue;
                s_min = std::min(s_min, t0); s_max = std::max(s_max, t0);
                e_max = std::max(e_max, at(L, wg, wave, 6));
                if (at(L - 1, wg, wave, 6)) { prev_end = std::max(prev_end, at(L - 1, wg, wave, 6)); prev_start = std::min(prev_start, at(L - 1, wg, wave, 0)); }
                if (at(L, wg, wave, 1)) {                          // a wave with groups to reduce
                    ph[0].push_back((double)(at(L, wg, wave, 1) - t0));
                    ph[1].push_back((double)(at(L, wg, wave, 2) - at(L, wg, wave, 1)));
                    ph[2].push_back((double)(at(L, wg, wave, 3) - at(L, wg, wave, 2)));
                    ph[3].push_back((double)(at(L, wg, wave, 4) - at(L, wg, wave, 3)));
                }
                ph[4].push_back((double)(at(L, wg, wave, 5) - at(L, wg, wave, 4)));
                ph[5].push_back((double)(at(L, wg, wave, 6) - at(L, wg, wave, 5)));
                start_of.push_back((double)(t0 - 0));
            }
        for (auto& x : start_of) x -= (double)s_min;
        printf("%dx%d Cin %3d: span %.2f us (first wave in .. last store acknowledged), launch period %.2f us, gap to the launch before %.2f us | starts spread %.2f (median %.2f) |"
               " loads landed +%.2f, MFMAs issued +%.2f, surplus landed +%.2f, barrier +%.2f, stores issued +%.2f, acknowledged +%.2f  (medians, us)\n",
               S, S, cin, (e_max - s_min) / 100.0, (s_min - prev_start) / 100.0, ((double)s_min - (double)prev_end) / 100.0, (s_max - s_min) / 100.0, med(start_of) / 100.0,
               med(ph[0]) / 100.0, med(ph[1]) / 100.0, med(ph[2]) / 100.0, med(ph[3]) / 100.0, med(ph[4]) / 100.0, med(ph[5]) / 100.0);
        // the workgroup that finished last: its waves one by one (us after the workgroup's first wave entered)
        int worst = 0; unsigned long long worst_end = 0;
        for (int wg = 0; wg < WGS; ++wg) if (at(L, wg, 0, 6) > worst_end) { worst_end = at(L, wg, 0, 6); worst = wg; }
        unsigned long long w0 = ~0ull;
        for (int wave = 0; wave < 8; ++wave) w0 = std::min(w0, at(L, worst, wave, 0));
        printf("    workgroup %d (last to finish; started %.2f us after the first):\n", worst, (w0 - s_min) / 100.0);
        for (int wave = 0; wave < 8; ++wave) {
            const unsigned hw = (unsigned)at(L, worst, wave, 7);
            printf("      wave %d simd %u cu %u se %u: in %.2f, loads landed %.2f, MFMAs issued %.2f, surplus landed %.2f, barrier %.2f, stores acknowledged %.2f\n", wave,
                   (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, (at(L, worst, wave, 0) - w0) / 100.0, at(L, worst, wave, 1) ? (at(L, worst, wave, 1) - w0) / 100.0 : -1.0,
                   at(L, worst, wave, 2) ? (at(L, worst, wave, 2) - w0) / 100.0 : -1.0, at(L, worst, wave, 3) ? (at(L, worst, wave, 3) - w0) / 100.0 : -1.0,
                   (at(L, worst, wave, 4) - w0) / 100.0, (at(L, worst, wave, 6) - w0) / 100.0);
        }
    }
    return 0;
}
