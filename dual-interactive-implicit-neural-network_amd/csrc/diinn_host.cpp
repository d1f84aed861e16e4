// diinn_host.cpp -- host-only half of the C ABI (include/diinn_hip.h):
// weight repacking, coordinate/index tables, size queries.  No HIP calls here.
// Build with -ffp-contract=off (axis_eval must round every fp32 op separately).
#include <cmath>
#include <cstring>
#include <cstdint>

#include <cstdlib>

#include "../../include/diinn_hip.h"
#include "diinn_layout.h"
#include "diinn_knobs.h"

using namespace diinn;

// ---- diagnostic overrides (diinn_knobs.h): one table, read from the environment once ----------------------------
namespace {
struct KnobDef { const char* name; std::atomic<long long> DiinnKnobs::*field; long long dflt; bool presence; };
const KnobDef KNOBS[] = {
    {"DIINN_F32_KERNEL", &DiinnKnobs::f32_kernel, 0, false},
    {"DIINN_BF16_KERNEL", &DiinnKnobs::bf16_kernel, 0, false},
    {"DIINN_X3_KERNEL", &DiinnKnobs::x3_kernel, 0, false},
    {"DIINN_PBF16_KERNEL", &DiinnKnobs::pbf16_kernel, 0, false},
    {"DIINN_P_KERNEL", &DiinnKnobs::p_kernel, 0, false},
    {"DIINN_P_X3_MIN", &DiinnKnobs::p_x3_min, 32768, false},
    {"DIINN_P_WINO_MIN", &DiinnKnobs::p_wino_min, 0, false},
    {"DIINN_ENC_S1_MIN_BLOCKS", &DiinnKnobs::enc_s1_min_blocks, 128, false},
    {"DIINN_ENC_NO_STREAM1X1", &DiinnKnobs::enc_no_stream1x1, 0, true},
    {"DIINN_ENC_LAT_MAX_TILES", &DiinnKnobs::enc_lat_max_tiles, 256, false},
    {"DIINN_ENC_WINO_MIN", &DiinnKnobs::enc_wino_min, 8192, false},
    {"DIINN_ENC_WINO4_MIN", &DiinnKnobs::enc_wino4_min, -1, false},
    {"DIINN_ENC_X3_MIN", &DiinnKnobs::enc_x3_min, 32768, false},
    {"DIINN_ENC_X3_ROWS", &DiinnKnobs::enc_x3_rows, 0, false},
    {"DIINN_ENC_WINO_HALF_MAX", &DiinnKnobs::enc_wino_half_max, -1, false},
    {"DIINN_ENC_WINO_PERSIST", &DiinnKnobs::enc_wino_persist, 256, false},
    {"DIINN_ENC_WINO4_SPLIT", &DiinnKnobs::enc_wino4_split, 1, false},
    {"DIINN_DEBUG_NCU", &DiinnKnobs::debug_ncu, 0, false},
    {"DIINN_ENC_WINO4_FAULT", &DiinnKnobs::enc_wino4_fault, 0, false},
    {"DIINN_TRAIN_SPLIT_HEAD", &DiinnKnobs::train_split_head, 0, false},
    {"DIINN_ENC_NO_T16", &DiinnKnobs::enc_no_t16, 0, false},
};
}  // namespace

DiinnKnobs& diinn_knobs() {
    static DiinnKnobs k;
    static const bool once = [] {
        for (const KnobDef& d : KNOBS) {
            const char* e = std::getenv(d.name);
            (k.*(d.field)).store(e ? (d.presence ? 1LL : std::atoll(e)) : d.dflt, std::memory_order_relaxed);
        }
        return true;
    }();
    (void)once;
    return k;
}

static const float INV_2PI = 0.15915494309189533577f;      // fp32(1 / (2 pi))

// fp32 -> bf16, round to nearest even (weights are finite; NaN is kept a NaN)
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

static inline float bf16_to_f32(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

extern "C" {

int diinn_abi_version(void) { return DIINN_ABI_VERSION; }

const char* diinn_status_string(int status) {
    switch (status) {
        case DIINN_OK: return "ok";
        case DIINN_ERR_INVALID_ARG: return "invalid argument";
        case DIINN_ERR_UNSUPPORTED: return "unsupported decoder variant";
        case DIINN_ERR_HIP: return "HIP runtime error";
        case DIINN_ERR_TOO_LARGE: return "extent too large for 32-bit indexing";
        default: return "unknown status";
    }
}

size_t diinn_packed_weight_floats(void) { return PACKED_FLOATS; }

int diinn_debug_set(const char* name, long long value) {
    if (!name) return DIINN_ERR_INVALID_ARG;
    DiinnKnobs& k = diinn_knobs();
    for (const KnobDef& d : KNOBS)
        if (!std::strcmp(name, d.name)) { (k.*(d.field)).store(value, std::memory_order_relaxed); return DIINN_OK; }
    return DIINN_ERR_INVALID_ARG;
}

int diinn_debug_get(const char* name, long long* value) {
    if (!name || !value) return DIINN_ERR_INVALID_ARG;
    DiinnKnobs& k = diinn_knobs();
    for (const KnobDef& d : KNOBS)
        if (!std::strcmp(name, d.name)) { *value = (k.*(d.field)).load(std::memory_order_relaxed); return DIINN_OK; }
    return DIINN_ERR_INVALID_ARG;
}

int diinn_liif_make_axis_tables(int n_in, int n_out, int v, int32_t* idx, float* rel, float* rel_cell) {
    if (n_in <= 0 || n_out <= 0 || (v != -1 && v != 1) || !idx || !rel) return DIINN_ERR_INVALID_ARG;
    const LiifAxis a = make_liif_axis(n_in, n_out);
    for (int j = 0; j < n_out; ++j) liif_axis_eval(a, j, v > 0 ? 1 : 0, idx[j], rel[j]);
    if (rel_cell) *rel_cell = a.rel_cell;
    return DIINN_OK;
}

size_t diinn_metasr_packed_floats(void) { return MS_PACKED_FLOATS; }

int diinn_metasr_pack_weights(const float* W1, const float* b1, const float* W2, const float* b2, float* packed) {
    if (!W1 || !b1 || !W2 || !b2 || !packed) return DIINN_ERR_INVALID_ARG;
    for (int o = 0; o < 3; ++o)
        for (int mm = 0; mm < MS_MM; ++mm)
            for (int kg = 0; kg < WL_KG; ++kg) {
                float* dst = packed + MS_OFF_W2 + (((size_t)o * MS_MM + mm) * WL_KG + kg) * WL_PIECE;
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = 3 * (32 * mm + (lane & 31)) + o;          // row of imnet.layers.2.weight
                    const int h = lane >> 5;
                    for (int e = 0; e < 4; ++e) dst[lane * 4 + e] = W2[(size_t)n * HID + chan_of(4 * kg + e, h)];
                }
            }
    float* q0 = packed + MS_OFF_Q0;
    for (int ch = 0; ch < HID; ++ch) {
        q0[0 * HID + ch] = W1[ch * 3 + 0];   // rel_h
        q0[1 * HID + ch] = W1[ch * 3 + 1];   // rel_w
        q0[2 * HID + ch] = W1[ch * 3 + 2];   // r_rev
        q0[3 * HID + ch] = b1[ch];
    }
    for (int o = 0; o < 3; ++o)
        for (int k = 0; k < MS_K; ++k) packed[MS_OFF_B2 + (size_t)o * MS_K + k] = b2[3 * k + o];
    return DIINN_OK;
}

int diinn_metasr_make_axis_tables(int n_in, int n_out, int32_t* idx, float* rel, float* r_rev) {
    if (n_in <= 0 || n_out <= 0 || !idx || !rel) return DIINN_ERR_INVALID_ARG;
    const MetaAxis a = make_meta_axis(n_in, n_out);
    for (int j = 0; j < n_out; ++j) meta_axis_eval(a, j, idx[j], rel[j]);
    if (r_rev) *r_rev = a.r_rev;
    return DIINN_OK;
}

size_t diinn_metasr_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * H * W * MS_K * sizeof(float);
}

int diinn_packed_section(int section, size_t* offset_floats, size_t* size_floats) {
    static const size_t off[17] = {OFF_WL, OFF_WP, OFF_BK, OFF_Q0, OFF_BQ, OFF_L, OFF_BL, OFF_WLB, OFF_WLT, OFF_WPB, OFF_BQR,
                                   OFF_Q0R, OFF_WLR, OFF_WPU, OFF_WLX, OFF_WPX, OFF_WL16};
    static const size_t sz[17]  = {SZ_WL, SZ_WP, 4 * HID, 4 * HID, 3 * HID, 3 * HID, 4, SZ_WLB, SZ_WLT, SZ_WPB, 3 * HID,
                                   4 * HID, SZ_WL, SZ_WPU, SZ_WLX, SZ_WPX, SZ_WL16};
    if (section < 0 || section > 16 || !offset_floats || !size_floats) return DIINN_ERR_INVALID_ARG;
    *offset_floats = off[section];
    *size_floats = sz[section];
    return DIINN_OK;
}

// Reference layouts:
//   K.i weight [256, 832]: input channel 0..255 = q, 256..831 = unfolded feature c*9+ky*3+kx
//   (torch.cat([q, x]) at diinn.py:136; unfold order at diinn.py:168)
int diinn_pack_weights(const float* K0w, const float* K0b,
                       const float* const Kw[3], const float* const Kb[3],
                       const float* Q0w, const float* Q0b,
                       const float* const Qw[3], const float* const Qb[3],
                       const float* Lw, const float* Lb,
                       float* packed) {
    if (!K0w || !K0b || !Kw || !Kb || !Q0w || !Q0b || !Qw || !Qb || !Lw || !Lb || !packed)
        return DIINN_ERR_INVALID_ARG;
    for (int i = 0; i < 3; ++i)
        if (!Kw[i] || !Kb[i] || !Qw[i] || !Qb[i]) return DIINN_ERR_INVALID_ARG;

    // WL: [layer][m][kg][part][lane][e]
    for (int i = 0; i < 3; ++i) {
        const float* wk = Kw[i];   // [256][832], q half = columns 0..255
        const float* wq = Qw[i];   // [256][256]
        for (int m = 0; m < 8; ++m)
            for (int kg = 0; kg < WL_KG; ++kg)
                for (int part = 0; part < 2; ++part) {
                    float* dst = packed + OFF_WL + (size_t)i * WL_LAYER +
                                 (((size_t)m * WL_KG + kg) * 2 + part) * WL_PIECE;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int out = 32 * m + (lane & 31);
                        const int h = lane >> 5;
                        for (int e = 0; e < 4; ++e) {
                            const int in = chan_of(4 * kg + e, h);
                            dst[lane * 4 + e] = part == 0 ? wk[(size_t)out * (HID + UNF) + in]
                                                          : wq[(size_t)out * HID + in];
                        }
                    }
                }
    }
    // WLR: WL with the synthesis pieces in revolutions
    for (size_t pc = 0; pc < SZ_WL / WL_PIECE; ++pc) {
        const float* src = packed + OFF_WL + pc * WL_PIECE;
        float* dst = packed + OFF_WLR + pc * WL_PIECE;
        const bool synth = (pc & 1) != 0;                       // [..][part 2][lane][e]: odd pieces are part 1
        for (size_t i = 0; i < WL_PIECE; ++i) dst[i] = synth ? src[i] * INV_2PI : src[i];
    }
    // WL16: A operands of v_mfma_f32_16x16x4_f32, [layer][wave][i][half][lane][T]; synthesis rows in revolutions
    for (int il = 0; il < 3; ++il)
        for (int w = 0; w < 4; ++w)
            for (int i = 0; i < 64; ++i)
                for (int half = 0; half < 2; ++half) {
                    float* dst = packed + OFF_WL16 + (size_t)il * WL16_LAYER + (size_t)w * WL16_WAVE + (size_t)i * WL16_KSTEP +
                                 (size_t)half * WL_PIECE;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int pos = 4 * i + (lane >> 4);
                        const int in = chan_of(pos >> 1, pos & 1);
                        for (int T = 0; T < 4; ++T) {
                            const int out = 64 * w + 16 * T + (lane & 15);
                            dst[lane * 4 + T] = half == 0 ? Kw[il][(size_t)out * (HID + UNF) + in]
                                                          : Qw[il][(size_t)out * HID + in] * INV_2PI;
                        }
                    }
                }
    // WLT: WL with the two channel indices swapped (backward pass)
    for (int i = 0; i < 3; ++i) {
        const float* wk = Kw[i];
        const float* wq = Qw[i];
        for (int m = 0; m < 8; ++m)
            for (int kg = 0; kg < WL_KG; ++kg)
                for (int part = 0; part < 2; ++part) {
                    float* dst = packed + OFF_WLT + (size_t)i * WL_LAYER +
                                 (((size_t)m * WL_KG + kg) * 2 + part) * WL_PIECE;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int in = 32 * m + (lane & 31);
                        const int h = lane >> 5;
                        for (int e = 0; e < 4; ++e) {
                            const int out = chan_of(4 * kg + e, h);
                            dst[lane * 4 + e] = part == 0 ? wk[(size_t)out * (HID + UNF) + in]
                                                          : wq[(size_t)out * HID + in];
                        }
                    }
                }
    }
    // WP: [mp][kg][t][lane][e], mo = 2mp+t;  Wx_i[ch][c*9 + tap] = feature half of K.i
    for (int mo = 0; mo < 32; ++mo) {
        const int i = mo >> 3;
        const float* w = (i == 0) ? K0w : Kw[i - 1];
        const size_t ld = (i == 0) ? (size_t)UNF : (size_t)(HID + UNF);
        const size_t col0 = (i == 0) ? 0 : (size_t)HID;
        for (int kg = 0; kg < WP_KG; ++kg) {
            float* dst = packed + OFF_WP + ((((size_t)(mo >> 1) * WP_KG + kg) * 2) + (mo & 1)) * WL_PIECE;
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = 32 * (mo & 7) + (lane & 31);
                const int h = lane >> 5;
                for (int e = 0; e < 4; ++e) {
                    const int kk = 4 * kg + e;
                    const int t = kk >> 5;                 // tap ky*3+kx
                    const int c = 2 * (kk & 31) + h;       // feature channel
                    dst[lane * 4 + e] = w[(size_t)ch * ld + col0 + (size_t)c * 9 + t];
                }
            }
        }
    }
    // WPU: U = G Wx G^T per (output, input) pair in float64, rounded once; column 2 negated (diinn_layout.h)
    {
        static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
        for (int mt = 0; mt < 32; ++mt) {
            const int il = mt >> 3;
            const float* w = (il == 0) ? K0w : Kw[il - 1];
            const size_t ld = (il == 0) ? (size_t)UNF : (size_t)(HID + UNF);
            const size_t col0 = (il == 0) ? 0 : (size_t)HID;
            for (int sg = 0; sg < 8; ++sg)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int ch = 32 * (mt & 7) + (lane & 31);
                        const int c = 8 * sg + 2 * e + (lane >> 5);
                        const float* g = w + (size_t)ch * ld + col0 + (size_t)c * 9;       // g[ky*3 + kx]
                        double t[4][3];                                                     // G g
                        for (int i = 0; i < 4; ++i)
                            for (int b = 0; b < 3; ++b)
                                t[i][b] = G[i][0] * g[b] + G[i][1] * g[3 + b] + G[i][2] * g[6 + b];
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j) {
                                const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                                float* dst = packed + OFF_WPU + ((((size_t)mt * 4 + i) * 8 + sg) * 4 + j) * WL_PIECE;
                                dst[lane * 4 + e] = (float)(j == 2 ? -u : u);
                            }
                    }
        }
    }
    // small tables
    float* bk = packed + OFF_BK;
    std::memcpy(bk, K0b, HID * sizeof(float));
    for (int i = 0; i < 3; ++i) std::memcpy(bk + (i + 1) * HID, Kb[i], HID * sizeof(float));
    float* q0 = packed + OFF_Q0;
    for (int ch = 0; ch < HID; ++ch) {
        q0[0 * HID + ch] = Q0w[ch * 3 + 0];   // rel_h
        q0[1 * HID + ch] = Q0w[ch * 3 + 1];   // rel_w
        q0[2 * HID + ch] = Q0w[ch * 3 + 2];   // ratio
        q0[3 * HID + ch] = Q0b[ch];
    }
    for (int i = 0; i < 3; ++i) std::memcpy(packed + OFF_BQ + i * HID, Qb[i], HID * sizeof(float));
    for (int i = 0; i < 3; ++i)
        for (int ch = 0; ch < HID; ++ch) packed[OFF_BQR + i * HID + ch] = Qb[i][ch] * INV_2PI;
    for (int i = 0; i < 4 * HID; ++i) packed[OFF_Q0R + i] = q0[i] * INV_2PI;
    std::memcpy(packed + OFF_L, Lw, 3 * HID * sizeof(float));
    float* bl = packed + OFF_BL;
    bl[0] = Lb[0]; bl[1] = Lb[1]; bl[2] = Lb[2];
    const uint32_t magic = DIINN_PACKED_MAGIC;                  // validity word: this image holds its derived sections
    std::memcpy(bl + 3, &magic, 4);
    // WLB: [layer][m][ks][part][lane][j] bf16; WLX (split-bf16 arithmetic): [layer][m][ks][k_hi, q_hi, k_lo, q_lo][lane][j],
    // hi as in WLB, lo = bf16(w - hi)
    uint16_t* wlb = reinterpret_cast<uint16_t*>(packed + OFF_WLB);
    uint16_t* wlx = reinterpret_cast<uint16_t*>(packed + OFF_WLX);
    for (int i = 0; i < 3; ++i)
        for (int m = 0; m < 8; ++m)
            for (int ks = 0; ks < 16; ++ks)
                for (int part = 0; part < 2; ++part) {
                    uint16_t* dst = wlb + ((((size_t)i * 8 + m) * 16 + ks) * 2 + part) * (64 * 8);
                    uint16_t* xhi = wlx + ((((size_t)i * 8 + m) * 16 + ks) * 4 + part) * (64 * 8);
                    uint16_t* xlo = xhi + 2 * (64 * 8);
                    for (int lane = 0; lane < 64; ++lane) {
                        const int out = 32 * m + (lane & 31);
                        const int h = lane >> 5;
                        for (int j = 0; j < 8; ++j) {
                            const int in = chan_of_bf16(ks, h, j);
                            // synthesis rows in revolutions (one fp32 multiply, then the bf16 rounding)
                            const float w = part == 0 ? Kw[i][(size_t)out * (HID + UNF) + in]
                                                      : Qw[i][(size_t)out * HID + in] * INV_2PI;
                            const uint16_t hi = f32_to_bf16(w);
                            dst[lane * 8 + j] = hi;
                            xhi[lane * 8 + j] = hi;
                            xlo[lane * 8 + j] = f32_to_bf16(w - bf16_to_f32(hi));   // exact difference, then one rounding
                        }
                    }
                }
    // WPB: [mp][ks][t][lane][j] bf16
    uint16_t* wpb = reinterpret_cast<uint16_t*>(packed + OFF_WPB);
    for (int mo = 0; mo < 32; ++mo) {
        const int i = mo >> 3;
        const float* w = (i == 0) ? K0w : Kw[i - 1];
        const size_t ld = (i == 0) ? (size_t)UNF : (size_t)(HID + UNF);
        const size_t col0 = (i == 0) ? 0 : (size_t)HID;
        for (int ks = 0; ks < WPB_KS; ++ks) {
            uint16_t* dst = wpb + ((((size_t)(mo >> 1) * WPB_KS + ks) * 2) + (mo & 1)) * (64 * 8);
            const int tap = ks >> 2, cg = ks & 3;
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = 32 * (mo & 7) + (lane & 31);
                for (int j = 0; j < 8; ++j) {
                    const int c = 16 * cg + 8 * (lane >> 5) + j;
                    dst[lane * 8 + j] = f32_to_bf16(w[(size_t)ch * ld + col0 + (size_t)c * 9 + tap]);
                }
            }
        }
    }
    // WPX: [og][group][tap][mt][hi, lo][lane][j] bf16 (split-bf16 hoisted conv)
    uint16_t* wpx = reinterpret_cast<uint16_t*>(packed + OFF_WPX);
    for (int og = 0; og < 16; ++og) {
        const int i = og >> 2;                                   // 256 channels per layer = 4 output groups of 64
        const float* w = (i == 0) ? K0w : Kw[i - 1];
        const size_t ld = (i == 0) ? (size_t)UNF : (size_t)(HID + UNF);
        const size_t col0 = (i == 0) ? 0 : (size_t)HID;
        for (int g = 0; g < 4; ++g)
            for (int tap = 0; tap < 9; ++tap)
                for (int mt = 0; mt < 2; ++mt) {
                    uint16_t* dhi = wpx + (((((size_t)og * 4 + g) * 9 + tap) * 2 + mt) * 2 + 0) * (64 * 8);
                    uint16_t* dlo = dhi + 64 * 8;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int ch = 64 * (og & 3) + 32 * mt + (lane & 31);
                        for (int j = 0; j < 8; ++j) {
                            const int c = 16 * g + 8 * (lane >> 5) + j;
                            const float v = w[(size_t)ch * ld + col0 + (size_t)c * 9 + tap];
                            const uint16_t hi = f32_to_bf16(v);
                            dhi[lane * 8 + j] = hi;
                            dlo[lane * 8 + j] = f32_to_bf16(v - bf16_to_f32(hi));
                        }
                    }
                }
    }
    return DIINN_OK;
}

int diinn_uses_small_output_kernel(int Hu, int Wu) { return (Hu + Wu) <= 128 ? 1 : 0; }

int diinn_make_axis_tables(int n_in, int n_out, int small_output, int32_t* idx, float* rel) {
    if (n_in <= 0 || n_out <= 0) return DIINN_ERR_INVALID_ARG;
    const Axis a = make_axis(n_in, n_out, small_output ? 1 : 0);
    for (int j = 0; j < n_out; ++j) {
        int id; float r;
        axis_eval(a, j, id, r);
        if (idx) idx[j] = id;
        if (rel) rel[j] = r;
    }
    return DIINN_OK;
}

size_t diinn_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * H * W * PCH * sizeof(float);
}

int diinn_lr_rows_for_band(int H, int Hu, int Wu, int y0, int y1, int* r0, int* r1) {
    if (H <= 0 || Hu <= 0 || Wu <= 0 || y0 < 0 || y1 > Hu || y0 >= y1 || !r0 || !r1)
        return DIINN_ERR_INVALID_ARG;
    const Axis a = make_axis(H, Hu, diinn_uses_small_output_kernel(Hu, Wu));
    int lo, hi; float rel;
    axis_eval(a, y0, lo, rel);       // idx is monotone non-decreasing in j
    axis_eval(a, y1 - 1, hi, rel);
    *r0 = lo;
    *r1 = hi + 1;
    return DIINN_OK;
}

int diinn_window_rows(int H, int Hu, int Wu, int y0, int y1,
                      int* feat_row0, int* feat_rows, int* p_row0, int* p_rows) {
    int r0, r1;
    const int st = diinn_lr_rows_for_band(H, Hu, Wu, y0, y1, &r0, &r1);
    if (st) return st;
    const int a0 = r0 > 0 ? r0 - 1 : 0, a1 = r1 < H ? r1 + 1 : H;     // + the 3x3 halo, clipped to the map
    if (feat_row0) *feat_row0 = a0;
    if (feat_rows) *feat_rows = a1 - a0;
    if (p_row0) *p_row0 = r0;
    if (p_rows) *p_rows = r1 - r0;
    return DIINN_OK;
}

}  // extern "C"
