/*
 * diinn_oracle_c.c -- plain-C restatement of the reference's decode path.  TEST INFRASTRUCTURE ONLY
 * (an independent second opinion beside oracle/diinn_oracle.py; never linked into the product).
 *
 * Follows /root/reference/src/models/components/diinn.py as written, pixel by pixel, with no
 * algebraic restructuring:
 *   diinn_oracle_axis    -> _make_pos_encoding (diinn.py:94-110) + ATen's CPU nearest-exact index
 *                           (generic kernel: fused fma rounding; small-output kernel, Hu+Wu <= 128:
 *                           double product -- see oracle/diinn_oracle.py: nearest_exact_index)
 *   diinn_oracle_decode  -> forward (diinn.py:163-173): unfold 3x3 with zero padding (:168),
 *                           nearest-exact replication (:168), syn_inp = (rel_h, rel_w, ratio)
 *                           (:165-167), step() mode 3 (:132-139) with K[i] = conv1x1 + ReLU,
 *                           Q[i] = conv1x1 + sin (:73-78), last_layer (:92)
 * Weights arrive in the reference's own layouts (state_dict order).
 * Parity status: pinned by tests/test_oracle_golden.py against tests/golden (outputs of the real
 * reference), tolerance 5e-6 (dot products are summed in index order here, mkldnn blocks them).
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define C_IN 64
#define UNF 576
#define HID 256

void diinn_oracle_axis(int n_in, int n_out, int small_output, int32_t* idx, float* rel) {
    const float c0_in = (float)(-1.0 + 1.0 / (double)n_in), c1_in = (float)(2.0 / (double)n_in);
    const float c0_out = (float)(-1.0 + 1.0 / (double)n_out), c1_out = (float)(2.0 / (double)n_out);
    const float scale = (float)n_in / (float)n_out;
    for (int j = 0; j < n_out; ++j) {
        int id;
        if (small_output) {
            id = (int)floorf((float)(((double)j + 0.5) * (double)scale));
        } else {
            float r = fmaf(scale, (float)j + 0.5f, -0.5f);
            if (r < 0.0f) r = 0.0f;
            id = (int)floorf((float)((double)r + 0.5));
        }
        if (id > n_in - 1) id = n_in - 1;
        const float g_out = c1_out * (float)j + c0_out;
        const float g_in = c1_in * (float)id + c0_in;
        if (idx) idx[j] = id;
        if (rel) rel[j] = (g_out - g_in) * (float)n_in;
    }
}

static float dotf(const float* w, const float* x, int n) {
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc += w[i] * x[i];
    return acc;
}

/* w: K0w[256*576] K0b[256] K1w[256*832] K1b K2w K2b K3w K3b Q0w[256*3] Q0b Q1w[256*256] Q1b Q2w Q2b Q3w Q3b
 *    Lw[3*256] Lb[3]   -- 18 pointers in that order */
int diinn_oracle_decode(const float* feat, int B, int H, int W, int Hu, int Wu,
                        const float* const* w, float* out, int y0, int y1) {
    const int small = (Hu + Wu) <= 128;
    int32_t* ih = (int32_t*)malloc(sizeof(int32_t) * Hu);
    int32_t* iw = (int32_t*)malloc(sizeof(int32_t) * Wu);
    float* rh = (float*)malloc(sizeof(float) * Hu);
    float* rw = (float*)malloc(sizeof(float) * Wu);
    if (!ih || !iw || !rh || !rw) return 1;
    diinn_oracle_axis(H, Hu, small, ih, rh);
    diinn_oracle_axis(W, Wu, small, iw, rw);
    const float ratio = (float)(((double)H * (double)W) / ((double)Hu * (double)Wu));
    const float *K0w = w[0], *K0b = w[1], *Q0w = w[8], *Q0b = w[9], *Lw = w[16], *Lb = w[17];
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int b = 0; b < B; ++b) {
        for (int y = y0; y < y1; ++y) {
            float X[UNF], cat[HID + UNF], k[HID], q[HID], qn[HID];
            for (int x = 0; x < Wu; ++x) {
                const int cy = ih[y], cx = iw[x];
                for (int c = 0; c < C_IN; ++c)            /* unfold: channel c*9 + ky*3 + kx, zero padding */
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int yy = cy + ky - 1, xx = cx + kx - 1;
                            X[c * 9 + ky * 3 + kx] = (yy >= 0 && yy < H && xx >= 0 && xx < W)
                                ? feat[(((size_t)b * C_IN + c) * H + yy) * W + xx] : 0.0f;
                        }
                const float syn[3] = {rh[y], rw[x], ratio};
                for (int o = 0; o < HID; ++o) {
                    const float kv = dotf(K0w + (size_t)o * UNF, X, UNF) + K0b[o];
                    k[o] = kv > 0.0f ? kv : 0.0f;
                    q[o] = k[o] * sinf(dotf(Q0w + o * 3, syn, 3) + Q0b[o]);
                }
                for (int i = 1; i < 4; ++i) {
                    const float *Kw = w[2 * i], *Kb = w[2 * i + 1], *Qw = w[8 + 2 * i], *Qb = w[9 + 2 * i];
                    memcpy(cat, q, sizeof(float) * HID);               /* torch.cat([q, x]) */
                    memcpy(cat + HID, X, sizeof(float) * UNF);
                    for (int o = 0; o < HID; ++o) {
                        const float kv = dotf(Kw + (size_t)o * (HID + UNF), cat, HID + UNF) + Kb[o];
                        const float sv = dotf(Qw + (size_t)o * HID, q, HID) + Qb[o];
                        qn[o] = (kv > 0.0f ? kv : 0.0f) * sinf(sv);
                    }
                    memcpy(q, qn, sizeof(float) * HID);
                }
                for (int o = 0; o < 3; ++o)
                    out[(((size_t)b * 3 + o) * (y1 - y0) + (y - y0)) * Wu + x] = dotf(Lw + o * HID, q, HID) + Lb[o];
            }
        }
    }
    free(ih); free(iw); free(rh); free(rw);
    return 0;
}
