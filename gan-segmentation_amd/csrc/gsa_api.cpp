// Host side of the C ABI (include/gsa.h): context, weight repacking, workspace, and the
// stream-ordered launch sequences of the generator and the decoder.
//
// Reference call sites replaced (reference file:line):
//   gsa_generator_*  Generator.__init__/load_parameters/hybrid_forward   networks_stylegan.py:78-197,
//                                                                        image_generator.py:20-22
//   gsa_decoder_*    Decoder.__init__/load_parameters/hybrid_forward     networks_seg.py:51-113,
//                                                                        seg_solver.py:307-349
//   gsa_generate     the per-batch body of `main.py generate`            main.py:97-99
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gsa.h"
#include "gsa_kernels.h"

using namespace gsa;

namespace {

constexpr int kMaxLevels = 12;
thread_local std::string g_create_error;

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> dims;
};

struct GenBlockDev {
    int C = 0, Cin = 0, R = 0;
    bool has_conv1 = false, is_deconv = false;
    float* w1 = nullptr;  // packed conv_1 / deconv_1
    float* blur = nullptr;
    float* nscale[2] = {nullptr, nullptr};
    float* nbias[2] = {nullptr, nullptr};
    float* w2 = nullptr;
    float* w2u = nullptr; // conv_2 in Winograd form (layers the static rule selects), or null
    float* gamma[2] = {nullptr, nullptr};
    float* beta[2] = {nullptr, nullptr};
    int style_off[2] = {0, 0};  // column offset of this layer's 2C styles
};

struct DecLevelDev {
    int F = 0, I = 0, in_c = 0, cs = 0;
    bool is_last = false, has_sc = false;
    float *cvt_w = nullptr, *cvt_s = nullptr, *cvt_beta = nullptr;      // *_s, *_beta: bias + BatchNorm folded (load_bn)
    float *cvt_u = nullptr, *b_u = nullptr;   // Winograd forms of cvt / conv b, or null
    float *a_w = nullptr, *a_s = nullptr, *a_beta = nullptr;
    float *b_w = nullptr, *b_s = nullptr, *b_beta = nullptr;
    float *sc_w = nullptr, *sc_b = nullptr;
    float *f_w = nullptr, *f_b = nullptr;
};

struct ProfEntry {
    std::string name;
    double ms = 0, flops = 0, bytes = 0, alg_flops = 0;   // flops: executed by the kernel; alg_flops: of the reference's formulation
    int64_t launches = 0;
};

struct ProfEvent {
    hipEvent_t a, b;
    int entry;
};

}  // namespace

struct gsa_ctx {
    int device = 0;
    std::string err;
    std::vector<void*> g_allocs, d_allocs, ws_allocs;

    // generator
    bool g_init = false, g_ready = false;
    gsa_generator_config gc{};
    std::map<std::string, HostTensor> gparams;
    int nlev = 0;
    int ch[kMaxLevels] = {0};
    float* map_wt[8] = {nullptr};
    float* map_b[8] = {nullptr};
    float *latent_avg = nullptr, *psi = nullptr, *constant = nullptr;
    float *style_wt = nullptr, *style_b = nullptr;
    int* style_col_layer = nullptr;
    int style_cols = 0;
    GenBlockDev blk[kMaxLevels];
    float *rgb_w = nullptr, *rgb_b = nullptr;

    // decoder
    bool d_init = false, d_ready = false;
    int d_n = 0, d_s0 = 0, d_bn = 1;
    int d_feat[kMaxLevels + 1] = {0}, d_inch[kMaxLevels] = {0};
    std::map<std::string, HostTensor> dparams;
    DecLevelDev dl[kMaxLevels];

    // workspace
    int max_batch = 0;
    float* lat[2] = {nullptr, nullptr};
    float* styles = nullptr;
    float *t_raw = nullptr, *x1 = nullptr;
    float* x2[kMaxLevels] = {nullptr};
    Aff* aff1 = nullptr;
    Aff* aff2[kMaxLevels] = {nullptr};
    StatPart* partials = nullptr;
    StatPart* stat_acc = nullptr;
    unsigned* stat_tickets = nullptr;
    unsigned* map_ctl = nullptr;            // fused mapping network: launch number, error word
    unsigned long long* map_ll[2] = {nullptr, nullptr};   // its {value, tag} exchange buffers
    float* zeros = nullptr;                 // 256 bytes of zeros (gsa_create): the out-of-image source of conv3x3_wino43's LDS-DMA halo gather
    unsigned long long* stamps = nullptr;   // diagnostic build only
    float* din[kMaxLevels] = {nullptr};
    float* cvt[kMaxLevels] = {nullptr};
    float *ya[kMaxLevels] = {nullptr}, *scb[kMaxLevels] = {nullptr}, *prev[kMaxLevels] = {nullptr};

    // second stream of gsa_generate: the decoder's low-resolution levels run beside the synthesis
    hipStream_t side = nullptr;
    hipEvent_t ev_level[kMaxLevels] = {nullptr};
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int bf16 = 0;                 // gsa_set_precision: 1 = bf16 MFMA operands (fixed once weights are committed)
    int prio = 0;                 // GSA_PRIO experiment switch
    int dbg = 0;                  // GSA_DBG, read once at gsa_create (only the diagnostic build looks at it)
    int side_levels = -1;         // decoder levels 0..side_levels-1 go to the side stream; -1 = by batch size (GSA_SIDE_LEVELS)
    int fault = 0;                // gsa_debug_inject (tests only; never armed by the environment): 1 = the fused mapping network is launched one
                                  // workgroup short; 2 = the next generator pass returns an error between a statistics producer and its finalize
    int fault_range_after = -1;   // gsa_debug_inject kind 3: passes left until the statistics-range word is set on the pass's stream (-1 = off)
    size_t partials_bytes = 0, stat_acc_bytes = 0, ticket_bytes = 0;
    bool stats_dirty = false;     // a generator pass failed between a statistics producer and its finalize: the rows are re-zeroed by the next pass

    // profiling
    int prof = 0;
    std::vector<ProfEntry> prof_entries;
    std::vector<ProfEvent> prof_events;
    std::vector<hipEvent_t> event_pool;
};

namespace {

int fail(gsa_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

int hip_fail(gsa_ctx* c, hipError_t e, const char* what) {
    return fail(c, GSA_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(c, e_, #expr); } while (0)

int nf(const gsa_generator_config& g, int r) {
    // reference networks_stylegan.py:114-116
    int fmaps = (int)(g.fmap_base / std::pow(2.0, (r - 1) * g.fmap_decay));
    return fmaps < g.fmap_max ? fmaps : g.fmap_max;
}

// (W*std)*lr_mult -- two fp32 roundings, reference networks_stylegan.py:407-412,513-518
inline float eff(float w, float std, bool use_std, float lr) {
    float v = use_std ? w * std : w;
    return v * lr;
}

int upload(gsa_ctx* c, const std::vector<float>& h, float** out, std::vector<void*>& track) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, h.size() * sizeof(float) + 16));
    track.push_back(d);
    HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = (float*)d;
    return GSA_OK;
}

// round-to-nearest-even fp32 -> bf16 (what v_cvt_pk_bf16_f32 does to the activations)
inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Upload of an MFMA weight pack (a sequence of 256-float [ci][16][cg] chunks, one per tap and
// (16 couts, 16 channels) pair).  bf16 mode: each chunk becomes [kq][16][4] bf16 with channel = 4*kq+j,
// the k order of v_mfma_f32_16x16x16_bf16 -- half the bytes, addressed in the same 4-byte slots.
int upload_mfma(gsa_ctx* c, const std::vector<float>& h, float** out, std::vector<void*>& track) {
    if (!c->bf16) return upload(c, h, out, track);
    std::vector<float> packed(h.size() / 2);
    uint16_t* o = reinterpret_cast<uint16_t*>(packed.data());
    for (size_t i = 0; i < h.size(); ++i) o[i] = bf16_rne(h[i]);      // [kq][16][j], channel 4kq+j: the fp32 pack order already
    return upload(c, packed, out, track);
}

template <typename T>
int dev_alloc(gsa_ctx* c, size_t count, T** out, std::vector<void*>& track) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, count * sizeof(T) + 256));
    track.push_back(d);
    *out = (T*)d;
    return GSA_OK;
}

void free_all(std::vector<void*>& v) {
    for (void* p : v) (void)hipFree(p);
    v.clear();
}

// conv OIHW (O,I,3,3) -> [O/16][I/16][tap][ci][16][cg]; channel = 16cb+4ci+cg (ci = k slot of the MFMA, cg = which of the
// four MFMAs of the (tap, block)), cout = 16g+n
std::vector<float> pack_conv3(const float* w, int O, int I, float std, bool us, float lr) {
    const int ct = 16;
    std::vector<float> out((size_t)O * I * 9);
    const int nblk = I / 16, G = O / ct;
    for (int g = 0; g < G; ++g)
        for (int cb = 0; cb < nblk; ++cb)
            for (int t = 0; t < 9; ++t)
                for (int ci = 0; ci < 4; ++ci)
                    for (int n = 0; n < ct; ++n)
                        for (int cg = 0; cg < 4; ++cg) {
                            const int o = g * ct + n, ch = cb * 16 + ci * 4 + cg;
                            out[((((((size_t)g * nblk + cb) * 9 + t) * 4 + ci) * ct + n) * 4) + cg] =
                                eff(w[((size_t)o * I + ch) * 9 + t], std, us, lr);
                        }
    return out;
}

// deconv IOHW (I,O,4,4) -> [O/16][I/16][tap16][ci][16][cg]
std::vector<float> pack_deconv(const float* w, int I, int O, float std, bool us, float lr) {
    const int ct = 16;
    std::vector<float> out((size_t)O * I * 16);
    const int nblk = I / 16, G = O / ct;
    for (int g = 0; g < G; ++g)
        for (int cb = 0; cb < nblk; ++cb)
            for (int t = 0; t < 16; ++t)
                for (int ci = 0; ci < 4; ++ci)
                    for (int n = 0; n < ct; ++n)
                        for (int cg = 0; cg < 4; ++cg) {
                            const int o = g * ct + n, ch = cb * 16 + ci * 4 + cg;
                            out[((((((size_t)g * nblk + cb) * 16 + t) * 4 + ci) * ct + n) * 4) + cg] =
                                eff(w[((size_t)ch * O + o) * 16 + t], std, us, lr);
                        }
    return out;
}

// nearest-x2 + conv3x3 (OIHW (O,I,3,3)) in sub-pixel form: the equivalent stride-2 transposed
// 4x4 kernel Wd[a][b] = sum_{ky in S(a)} sum_{kx in S(b)} W[ky][kx], S(0)={2} S(1)={1,2}
// S(2)={0,1} S(3)={0}; fp32 sums, ky then kx ascending, left to right (canonical order,
// DESIGN.md).  Packed like a deconv: [O/16][I/16][tap16][ci][16][cg].
std::vector<float> pack_upconv(const float* w, int O, int I, float std, bool us, float lr) {
    static const int S[4][2] = {{2, -1}, {1, 2}, {0, 1}, {0, -1}};
    std::vector<float> out((size_t)O * I * 16);
    const int nblk = I / 16, G = O / 16;
    for (int g = 0; g < G; ++g)
        for (int cb = 0; cb < nblk; ++cb)
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b)
                    for (int ci = 0; ci < 4; ++ci)
                        for (int n = 0; n < 16; ++n)
                            for (int cg = 0; cg < 4; ++cg) {
                                const int o = g * 16 + n, ch = cb * 16 + ci * 4 + cg;
                                const float* wk = w + ((size_t)o * I + ch) * 9;
                                float sum = 0.0f;
                                bool first = true;
                                for (int i = 0; i < 2; ++i)
                                    for (int j = 0; j < 2; ++j) {
                                        if (S[a][i] < 0 || S[b][j] < 0) continue;
                                        const float e = eff(wk[S[a][i] * 3 + S[b][j]], std, us, lr);
                                        sum = first ? e : sum + e;
                                        first = false;
                                    }
                                out[((((((size_t)g * nblk + cb) * 16 + a * 4 + b) * 4 + ci) * 16 + n) * 4) + cg] = sum;
                            }
    return out;
}

// 1x1 shortcut (O,I,1,1) -> [O/16][I/16][ci][16][cg]
std::vector<float> pack_conv1(const float* w, int O, int I) {
    const int ct = 16;
    std::vector<float> out((size_t)O * I);
    const int nblk = I / 16, G = O / ct;
    for (int g = 0; g < G; ++g)
        for (int cb = 0; cb < nblk; ++cb)
            for (int ci = 0; ci < 4; ++ci)
                for (int n = 0; n < ct; ++n)
                    for (int cg = 0; cg < 4; ++cg)
                        out[(((((size_t)g * nblk + cb) * 4 + ci) * ct + n) * 4) + cg] =
                            w[(size_t)(g * ct + n) * I + cb * 16 + ci * 4 + cg];
    return out;
}

// Winograd F(2x2,3x3) weights of a 3x3 conv OIHW (O,I,3,3): U = G g G^T evaluated in double on the effective fp32
// weights and rounded once (canonical arithmetic; the oracle's pack_wino is the same code path restated),
// packed like a 16-tap kernel: [O/16][I/16][f = 4i+j][ci][16][cg]
std::vector<float> pack_wino(const float* w, int O, int I, float std, bool us, float lr) {
    std::vector<float> out((size_t)O * I * 16);
    const int nblk = I / 16, G = O / 16;
    for (int g = 0; g < G; ++g)
        for (int cb = 0; cb < nblk; ++cb)
            for (int ci = 0; ci < 4; ++ci)
                for (int n = 0; n < 16; ++n)
                    for (int cg = 0; cg < 4; ++cg) {
                        const int o = g * 16 + n, ch = cb * 16 + ci * 4 + cg;
                        const float* wk = w + ((size_t)o * I + ch) * 9;
                        double k[3][3], r[4][3], u[4][4];
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b) k[a][b] = (double)eff(wk[a * 3 + b], std, us, lr);
                        for (int b = 0; b < 3; ++b) {
                            r[0][b] = k[0][b];
                            r[1][b] = 0.5 * ((k[0][b] + k[1][b]) + k[2][b]);
                            r[2][b] = 0.5 * ((k[0][b] - k[1][b]) + k[2][b]);
                            r[3][b] = k[2][b];
                        }
                        for (int a = 0; a < 4; ++a) {
                            u[a][0] = r[a][0];
                            u[a][1] = 0.5 * ((r[a][0] + r[a][1]) + r[a][2]);
                            u[a][2] = 0.5 * ((r[a][0] - r[a][1]) + r[a][2]);
                            u[a][3] = r[a][2];
                        }
                        for (int f = 0; f < 16; ++f)
                            out[((((((size_t)g * nblk + cb) * 16 + f) * 4 + ci) * 16 + n) * 4) + cg] = (float)u[f >> 2][f & 3];
                    }
    return out;
}

// Winograd F(4x4,3x3) weights (round 4): U = G g G^T with Lavin & Gray's 6x3 G, evaluated in double on the effective fp32 weights
// and rounded once (the oracle's pack_wino43 restated), packed per 8-CHANNEL block [O/16][I/8][f = 6i+j][kq][16][j2] with
// channel = 8b + 2kq + j2 (the K order of these layers: conv3x3_wino43): a lane's weight pair is one 8-byte LDS read
std::vector<float> pack_wino43(const float* w, int O, int I, float std, bool us, float lr) {
    static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    std::vector<float> out((size_t)O * I * 36);
    const int nblk = I / 8, NG = O / 16;
    for (int g = 0; g < NG; ++g)
        for (int cb = 0; cb < nblk; ++cb)
            for (int kq = 0; kq < 4; ++kq)
                for (int n = 0; n < 16; ++n)
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int o = g * 16 + n, ch = cb * 8 + kq * 2 + j2;
                        const float* wk = w + ((size_t)o * I + ch) * 9;
                        double k[3][3], r[6][3];
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b) k[a][b] = (double)eff(wk[a * 3 + b], std, us, lr);
                        for (int i = 0; i < 6; ++i)
                            for (int b = 0; b < 3; ++b) r[i][b] = (G[i][0] * k[0][b] + G[i][1] * k[1][b]) + G[i][2] * k[2][b];
                        for (int i = 0; i < 6; ++i)
                            for (int j = 0; j < 6; ++j)
                                out[((((((size_t)g * nblk + cb) * 36 + i * 6 + j) * 4 + kq) * 16 + n) * 2) + j2] =
                                    (float)((r[i][0] * G[j][0] + r[i][1] * G[j][1]) + r[i][2] * G[j][2]);
                    }
    return out;
}
// static rule of the F(4x4,3x3) form (gsa_kernels.hip conv_uses_wino43): a Winograd layer with >= 64 input channels and >= 32 px
inline bool wino43_layer(const gsa_ctx* c, int R, int Cin, int Cout);

// the static rule of the Winograd form (gsa_kernels.hip conv_uses_wino): plain 3x3 convs with outputs >= 64 px, or >= 32 px with
// at least 64 output channels (fewer tiles than that leave the chip idle: the direct small-tile kernels are faster there), fp32 mode
inline bool wino_layer(const gsa_ctx* c, int R, int Cout) { return !c->bf16 && (R >= 64 || (R >= 32 && Cout >= 64) || (R >= 16 && Cout >= 256)); }
inline bool wino43_layer(const gsa_ctx* c, int R, int Cin, int Cout) { return wino43_enabled() && wino_layer(c, R, Cout) && Cin >= 64 && R >= 32; }

// final conv (K,I,3,3) -> [cb][tap][c16][K]
std::vector<float> pack_final(const float* w, int K, int I) {
    std::vector<float> out((size_t)K * I * 9);
    for (int cb = 0; cb < I / 16; ++cb)
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < 16; ++ci)
                for (int o = 0; o < K; ++o)
                    out[(((size_t)cb * 9 + t) * 16 + ci) * K + o] = w[((size_t)o * I + cb * 16 + ci) * 9 + t];
    return out;
}

const HostTensor* find(const std::map<std::string, HostTensor>& m, const std::string& name) {
    auto it = m.find(name);
    return it == m.end() ? nullptr : &it->second;
}

int need(gsa_ctx* c, const std::map<std::string, HostTensor>& m, const std::string& name, size_t count,
         const float** out) {
    const HostTensor* t = find(m, name);
    if (!t) return fail(c, GSA_ERR_MISSING_PARAM, "parameter %s was not set", name.c_str());
    if (t->data.size() != count)
        return fail(c, GSA_ERR_INVALID, "parameter %s has %zu elements, expected %zu", name.c_str(), t->data.size(), count);
    *out = t->data.data();
    return GSA_OK;
}

#define NEED(map, name, count, ptr) do { int rc_ = need(c, map, name, count, ptr); if (rc_) return rc_; } while (0)

int get_std(gsa_ctx* c, const std::string& prefix, float* std) {
    *std = 1.0f;
    if (!c->gc.use_wscale) return GSA_OK;
    const float* p;
    NEED(c->gparams, prefix + "_std", 1, &p);
    *std = p[0];
    return GSA_OK;
}

bool known_generator_name(const gsa_ctx* c, const char* name) {
    int R = 0, k = 0, i = 0;
    char tail[64], t2[64];
    if (!strcmp(name, "constant_tensor") || !strcmp(name, "latent_avg") || !strcmp(name, "truncation_psi")) return true;
    if (sscanf(name, "mp_dense_%d_%63s", &i, tail) == 2)
        return i >= 0 && i < 8 && (!strcmp(tail, "weight") || !strcmp(tail, "bias") || !strcmp(tail, "std"));
    if (sscanf(name, "%d_%63s", &R, tail) != 2) return false;
    int r = 0;
    while ((1 << r) < R) ++r;
    if ((1 << r) != R || r < 2 || r > c->gc.max_res_log2) return false;
    if (!strcmp(tail, "conv_1_weight") || !strcmp(tail, "conv_1_std")) return r > 2 && r < 7;
    if (!strcmp(tail, "deconv_1_weight") || !strcmp(tail, "deconv_1_std")) return r >= 7;
    if (!strcmp(tail, "blur_1_w_kernel")) return r > 2;
    if (!strcmp(tail, "conv_2_weight") || !strcmp(tail, "conv_2_std")) return true;
    if (!strcmp(tail, "conv_to_rgb_weight") || !strcmp(tail, "conv_to_rgb_bias") || !strcmp(tail, "conv_to_rgb_std"))
        return r == c->gc.max_res_log2;
    if (sscanf(tail, "noise_%d_%63s", &k, t2) == 2) return (k == 1 || k == 2) && !strcmp(t2, "scale_factors");
    if (sscanf(tail, "bias_%d_%63s", &k, t2) == 2) return (k == 1 || k == 2) && !strcmp(t2, "bias");
    if (sscanf(tail, "adain_%d_%63s", &k, t2) == 2)
        return (k == 1 || k == 2) &&
               (!strcmp(t2, "dense_affine_weight") || !strcmp(t2, "dense_affine_bias") || !strcmp(t2, "dense_affine_std") ||
                !strcmp(t2, "norm_gamma") || !strcmp(t2, "norm_beta"));
    return false;
}

int set_param(gsa_ctx* c, std::map<std::string, HostTensor>& m, const char* name, const float* data, int ndim,
              const int64_t* dims) {
    if (!name || !data || ndim < 0 || ndim > 8) return fail(c, GSA_ERR_INVALID, "bad parameter tensor %s", name ? name : "(null)");
    HostTensor t;
    size_t cnt = 1;
    for (int i = 0; i < ndim; ++i) {
        if (dims[i] < 0) return fail(c, GSA_ERR_INVALID, "negative dimension in %s", name);
        t.dims.push_back(dims[i]);
        cnt *= (size_t)dims[i];
    }
    t.data.assign(data, data + cnt);
    m[name] = std::move(t);
    return GSA_OK;
}

void reset_generator_dev(gsa_ctx* c) {
    c->g_ready = false;
}

// ---------------------------------------------------------------- profiling wrapper

struct Launch {
    gsa_ctx* c;
    hipStream_t s;
    int entry = -1;
    hipEvent_t a = nullptr, b = nullptr;
    std::string kname_s;
    const char* kname = "";
    const char* label = nullptr;
    // flops = FLOP the kernel executes; alg = FLOP of the same layer in the reference's formulation (2*MACs of the direct
    // 9-tap / 16-tap convolution) when the kernel uses a cheaper form (sub-pixel, Winograd); < 0: the same
    Launch(gsa_ctx* ctx, hipStream_t st, const char* kernel, const char* layer, double flops, double bytes, double alg = -1.0) : c(ctx), s(st) {
        kname_s = kernel; kname = kname_s.c_str(); label = layer;
        if (!c->prof) return;
        std::string key = kernel;
        if (c->prof > 1 && layer) { key += " | "; key += layer; }
        for (size_t i = 0; i < c->prof_entries.size(); ++i)
            if (c->prof_entries[i].name == key) { entry = (int)i; break; }
        if (entry < 0) {
            ProfEntry e;
            e.name = key;
            c->prof_entries.push_back(e);
            entry = (int)c->prof_entries.size() - 1;
        }
        c->prof_entries[entry].flops += flops;
        c->prof_entries[entry].alg_flops += alg < 0 ? flops : alg;
        // bf16 mode: the activation tensors are 2 bytes per element (the fp32 noise planes and the parameters are a few per cent)
        c->prof_entries[entry].bytes += (c->bf16 && layer && strncmp(layer, "g.mapping", 9) && strncmp(layer, "g.styles", 8) && !strstr(layer, "finalize")) ? 0.5 * bytes : bytes;
        c->prof_entries[entry].launches += 1;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get(); b = get();
        (void)hipEventRecord(a, s);
    }
    ~Launch() {
#ifdef GSA_STAMP
        if (c->stamps && label && (strstr(kname, "conv") || strstr(kname, "subpixel"))) {   // diagnostic build: per-phase wave-cycle sums
            (void)hipStreamSynchronize(s);
            unsigned long long h[16];
            (void)hipMemcpy(h, c->stamps, sizeof h, hipMemcpyDeviceToHost);
            (void)hipMemset(c->stamps, 0, sizeof h);
            if (h[15] && strstr(kname, "_ws")) {
                const double w = (double)h[15];   // per wave (4 MFMA + 4 memory waves per workgroup)
                fprintf(stderr, "STAMPWS %-24s %-16s waves %6llu  mfma: work %9.0f barrier %9.0f | mem: write %9.0f load %9.0f epi %9.0f barrier %9.0f\n",
                        kname, label, h[15], h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w);
            } else if (h[15] && h[8]) {           // double-buffered persistent loop: cycles per ITEM per wave
                const double w = (double)h[12];
                fprintf(stderr, "STAMPDB %-70s %-14s waves %6llu items/wave %5.1f | per item: write %6.0f load %6.0f (epi-loads %5.0f index %5.0f bulk %5.0f) mfma %6.0f epi %6.0f barrier %6.0f\n",
                        kname, label, h[15], h[12] / (double)h[15], h[6] / w, h[7] / w, h[0] / w, h[1] / w, h[2] / w, h[8] / w, h[9] / w, h[10] / w);
                if (strstr(kname, "subpixel"))     // per wave index: the sums are over a quarter of the waves
                    fprintf(stderr, "STAMPSW %-14s barrier by wave: %6.0f %6.0f %6.0f %6.0f | write wave0 %6.0f wave3 %6.0f\n", label, 4 * h[0] / w, 4 * h[1] / w, 4 * h[2] / w, 4 * h[3] / w, 4 * h[4] / w, 4 * h[5] / w);
            } else if (h[15]) {
                const double w = (double)h[15];
                fprintf(stderr, "STAMP %-40s %-16s waves %8llu  load %7.0f  write %6.0f  bar %6.0f  blocks %8.0f  epi %7.0f\n", kname,
                        label, h[15], h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w);
            }
        }
#endif
        if (entry < 0) return;
        (void)hipEventRecord(b, s);
        c->prof_events.push_back(ProfEvent{a, b, entry});
    }
};

const char* conv_kernel_name(const ConvParams& cp, int n, int epi, bool sc) { return conv3x3_kernel_name(cp, epi, sc, n); }

}  // namespace

// =========================================================================================

extern "C" {

// the compiler is part of the build string: the s_nop padding around the inline-asm packed adds (valu_settle / mfma_settle,
// gsa_kernels.hip) is correct for the instruction order THIS hipcc emits; the bit-exact parity tests are what re-validates it
#define GSA_STR2(x) #x
#define GSA_STR(x) GSA_STR2(x)
const char* gsa_version(void) {
#ifndef GSA_EXPERIMENTS
#define GSA_EXPERIMENTS 0
#endif
    return "gsa-hip 0.5"
#if GSA_EXPERIMENTS
           "+experiments"
#endif
           " (gfx950, v_mfma_f32_16x16x4_f32 implicit-GEMM convs; built with hipcc = clang " __clang_version__
           ", HIP " GSA_STR(HIP_VERSION_MAJOR) "." GSA_STR(HIP_VERSION_MINOR) "." GSA_STR(HIP_VERSION_PATCH) ")";
}

int gsa_create(int device, gsa_ctx** out) {
    if (!out) return fail(nullptr, GSA_ERR_INVALID, "gsa_create: out is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, GSA_ERR_HIP, "no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(nullptr, GSA_ERR_INVALID, "device %d out of range (%d devices)", device, count);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, GSA_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    gsa_ctx* c = new gsa_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < kMaxLevels; ++i) e = hipEventCreateWithFlags(&c->ev_level[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) {
        const int rc = fail(nullptr, GSA_ERR_HIP, "creating the side stream: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    if (e == hipSuccess) e = hipMalloc((void**)&c->zeros, 256);
    if (e == hipSuccess) e = hipMemset(c->zeros, 0, 256);
    if (e != hipSuccess) {
        const int rc = fail(nullptr, GSA_ERR_HIP, "allocating the zero page: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    if (const char* v = getenv("GSA_SIDE_LEVELS")) c->side_levels = atoi(v);
    if (const char* v = getenv("GSA_DBG")) c->dbg = atoi(v);
    if (const char* v = getenv("GSA_PRIO")) c->prio = atoi(v);
    *out = c;
    return GSA_OK;
}

void gsa_destroy(gsa_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    free_all(c->g_allocs);
    free_all(c->d_allocs);
    free_all(c->ws_allocs);
    for (auto& ev : c->prof_events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < kMaxLevels; ++i) if (c->ev_level[i]) (void)hipEventDestroy(c->ev_level[i]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->zeros) (void)hipFree(c->zeros);
    delete c;
}

const char* gsa_last_error(const gsa_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

// ------------------------------------------------------------------------ generator setup

int gsa_generator_init(gsa_ctx* c, const gsa_generator_config* g) {
    if (!c || !g) return GSA_ERR_INVALID;
    if (g->max_res_log2 < 2 || g->max_res_log2 > kMaxLevels) return fail(c, GSA_ERR_INVALID, "max_res_log2 %d out of range", g->max_res_log2);
    if (g->latent_size != 512) return fail(c, GSA_ERR_INVALID, "latent_size must be 512 (latent_avg is (512,) in the reference)");
    if (g->channels < 1 || g->channels > 4) return fail(c, GSA_ERR_INVALID, "channels must be 1..4");
    c->gc = *g;
    c->nlev = g->max_res_log2 - 1;
    for (int l = 0; l < c->nlev; ++l) {
        c->ch[l] = nf(*g, l + 2);
        if (c->ch[l] % 16 || c->ch[l] <= 0)
            return fail(c, GSA_ERR_INVALID, "feature maps at %d px = %d: the MFMA kernels need multiples of 16", 4 << l, c->ch[l]);
    }
    c->gparams.clear();
    c->g_init = true;
    reset_generator_dev(c);
    return GSA_OK;
}

int gsa_generator_set_param(gsa_ctx* c, const char* name, const float* data, int32_t ndim, const int64_t* dims) {
    if (!c || !c->g_init) return fail(c, GSA_ERR_STATE, "gsa_generator_init first");
    if (!name) return fail(c, GSA_ERR_INVALID, "null parameter name");
    if (!known_generator_name(c, name)) return 1;  // ignore_extra=True
    c->g_ready = false;
    return set_param(c, c->gparams, name, data, ndim, dims);
}

int gsa_generator_commit(gsa_ctx* c) {
    if (!c || !c->g_init) return fail(c, GSA_ERR_STATE, "gsa_generator_init first");
    HIP_TRY(hipSetDevice(c->device));
    const auto& P = c->gparams;
    const int L = c->gc.latent_size;
    const bool us = c->gc.use_wscale != 0;
    const float *w, *b;
    float std;
    char nm[128];
    std::vector<float> h;
    HIP_TRY(hipDeviceSynchronize());
    c->g_ready = false;
    free_all(c->g_allocs);
    auto& T = c->g_allocs;

    const int C0 = c->ch[0];
    NEED(P, "constant_tensor", (size_t)C0 * 16, &w);
    h.assign((size_t)C0 * 16, 0.f);
    for (int ch = 0; ch < C0; ++ch)
        for (int p = 0; p < 16; ++p) h[(size_t)p * C0 + ch] = w[ch * 16 + p];
    if (int rc = upload(c, h, &c->constant, T)) return rc;
    NEED(P, "latent_avg", 512, &w);
    h.assign(w, w + 512);
    if (int rc = upload(c, h, &c->latent_avg, T)) return rc;
    NEED(P, "truncation_psi", (size_t)2 * c->nlev, &w);
    h.assign(w, w + 2 * c->nlev);
    if (int rc = upload(c, h, &c->psi, T)) return rc;

    for (int i = 0; i < 8; ++i) {
        snprintf(nm, sizeof nm, "mp_dense_%d", i);
        if (int rc = get_std(c, nm, &std)) return rc;
        NEED(P, std::string(nm) + "_weight", (size_t)L * L, &w);
        NEED(P, std::string(nm) + "_bias", (size_t)L, &b);
        h.assign((size_t)L * L, 0.f);
        for (int j = 0; j < L; ++j)
            for (int k = 0; k < L; ++k) h[(size_t)k * L + j] = eff(w[(size_t)j * L + k], std, us, 0.01f);  // lr_mult 0.01, reference :135
        if (int rc = upload(c, h, &c->map_wt[i], T)) return rc;
        h.assign((size_t)L, 0.f);
        for (int j = 0; j < L; ++j) h[j] = b[j] * 0.01f;
        if (int rc = upload(c, h, &c->map_b[i], T)) return rc;
    }

    // style affines of all 2*nlev layers concatenated on the output axis: WT [L][J], b [J]
    int J = 0;
    for (int l = 0; l < c->nlev; ++l) J += 4 * c->ch[l];
    c->style_cols = J;
    std::vector<float> swt((size_t)L * J), sb((size_t)J);
    std::vector<int> col_layer((size_t)J);
    int col = 0;
    for (int l = 0; l < c->nlev; ++l) {
        GenBlockDev& B = c->blk[l];
        const int r = l + 2, R = 1 << r, C = c->ch[l], Cin = l ? c->ch[l - 1] : C;
        B.C = C; B.Cin = Cin; B.R = R;
        B.has_conv1 = r > 2; B.is_deconv = r >= 7;   // reference networks_stylegan.py:154
        if (B.has_conv1) {
            snprintf(nm, sizeof nm, "%d_%s", R, B.is_deconv ? "deconv_1" : "conv_1");
            if (int rc = get_std(c, nm, &std)) return rc;
            if (B.is_deconv) {
                NEED(P, std::string(nm) + "_weight", (size_t)Cin * C * 16, &w);
                h = pack_deconv(w, Cin, C, std, us, 1.0f);
            } else {
                NEED(P, std::string(nm) + "_weight", (size_t)Cin * C * 9, &w);
                h = R >= 16 ? pack_upconv(w, C, Cin, std, us, 1.0f) : pack_conv3(w, C, Cin, std, us, 1.0f);
            }
            if (int rc = upload_mfma(c, h, &B.w1, T)) return rc;
            snprintf(nm, sizeof nm, "%d_blur_1_w_kernel", R);
            NEED(P, nm, (size_t)C * 9, &w);
            h.assign(w, w + (size_t)C * 9);
            if (int rc = upload(c, h, &B.blur, T)) return rc;
        }
        snprintf(nm, sizeof nm, "%d_conv_2", R);
        if (int rc = get_std(c, nm, &std)) return rc;
        NEED(P, std::string(nm) + "_weight", (size_t)C * C * 9, &w);
        h = pack_conv3(w, C, C, std, us, 1.0f);
        if (int rc = upload_mfma(c, h, &B.w2, T)) return rc;
        B.w2u = nullptr;
        if (wino_layer(c, R, C)) {
            h = wino43_layer(c, R, C, C) ? pack_wino43(w, C, C, std, us, 1.0f) : pack_wino(w, C, C, std, us, 1.0f);
            if (int rc = upload(c, h, &B.w2u, T)) return rc;
        }
        for (int k = 0; k < 2; ++k) {
            snprintf(nm, sizeof nm, "%d_noise_%d_scale_factors", R, k + 1);
            NEED(P, nm, (size_t)C, &w); h.assign(w, w + C);
            if (int rc = upload(c, h, &B.nscale[k], T)) return rc;
            snprintf(nm, sizeof nm, "%d_bias_%d_bias", R, k + 1);
            NEED(P, nm, (size_t)C, &w); h.assign(w, w + C);
            if (int rc = upload(c, h, &B.nbias[k], T)) return rc;
            snprintf(nm, sizeof nm, "%d_adain_%d_norm_gamma", R, k + 1);
            NEED(P, nm, (size_t)C, &w); h.assign(w, w + C);
            if (int rc = upload(c, h, &B.gamma[k], T)) return rc;
            snprintf(nm, sizeof nm, "%d_adain_%d_norm_beta", R, k + 1);
            NEED(P, nm, (size_t)C, &w); h.assign(w, w + C);
            if (int rc = upload(c, h, &B.beta[k], T)) return rc;
            snprintf(nm, sizeof nm, "%d_adain_%d_dense_affine", R, k + 1);
            if (int rc = get_std(c, nm, &std)) return rc;
            NEED(P, std::string(nm) + "_weight", (size_t)2 * C * L, &w);
            NEED(P, std::string(nm) + "_bias", (size_t)2 * C, &b);
            B.style_off[k] = col;
            for (int j = 0; j < 2 * C; ++j) {
                for (int q = 0; q < L; ++q) swt[(size_t)q * J + col + j] = eff(w[(size_t)j * L + q], std, us, 1.0f);
                sb[col + j] = b[j] * 1.0f;
                col_layer[col + j] = 2 * l + k;
            }
            col += 2 * C;
        }
    }
    if (int rc = upload(c, swt, &c->style_wt, T)) return rc;
    if (int rc = upload(c, sb, &c->style_b, T)) return rc;
    {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, sizeof(int) * J));
        T.push_back(d);
        HIP_TRY(hipMemcpy(d, col_layer.data(), sizeof(int) * J, hipMemcpyHostToDevice));
        c->style_col_layer = (int*)d;
    }
    {
        const int R = 1 << c->gc.max_res_log2, C = c->ch[c->nlev - 1], nc = c->gc.channels;
        snprintf(nm, sizeof nm, "%d_conv_to_rgb", R);
        if (int rc = get_std(c, nm, &std)) return rc;
        NEED(P, std::string(nm) + "_weight", (size_t)nc * C, &w);
        NEED(P, std::string(nm) + "_bias", (size_t)nc, &b);
        h.assign((size_t)nc * C, 0.f);
        for (size_t i = 0; i < (size_t)nc * C; ++i) h[i] = eff(w[i], std, us, 1.0f);
        if (int rc = upload(c, h, &c->rgb_w, T)) return rc;
        h.assign(b, b + nc);
        if (int rc = upload(c, h, &c->rgb_b, T)) return rc;
    }
    c->g_ready = true;
    return GSA_OK;
}

// ------------------------------------------------------------------------ decoder setup

int gsa_decoder_init(gsa_ctx* c, const gsa_decoder_config* d) {
    if (!c || !d || !d->features || !d->in_channels) return GSA_ERR_INVALID;
    if (d->num_feats < 1 || d->num_feats > kMaxLevels) return fail(c, GSA_ERR_INVALID, "num_feats %d out of range", d->num_feats);
    // start_res: the first feature the decoder consumes (reference networks_seg.py:56,64,81,102)
    if (d->start_res < 0 || d->start_res >= d->num_feats) return fail(c, GSA_ERR_INVALID, "start_res %d not in 0..%d", d->start_res, d->num_feats - 1);
    c->d_n = d->num_feats;
    c->d_s0 = d->start_res;
    c->d_bn = d->use_bn;
    for (int i = 0; i <= d->num_feats; ++i) c->d_feat[i] = d->features[i];
    for (int i = 0; i < d->num_feats; ++i) {
        c->d_inch[i] = d->in_channels[i];
        if (c->d_inch[i] % 16 || c->d_feat[i] % 16 || c->d_inch[i] <= 0 || c->d_feat[i] <= 0)
            return fail(c, GSA_ERR_INVALID, "decoder level %d: channel counts must be positive multiples of 16", i);
    }
    const int ncls = c->d_feat[d->num_feats];
    if (ncls < 1 || ncls > 8) return fail(c, GSA_ERR_INVALID, "num_classes %d not in 1..8", ncls);
    if (d->num_feats < 3) return fail(c, GSA_ERR_INVALID, "the final resolution must be at least 16 px (num_feats >= 3)");
    c->dparams.clear();
    c->d_init = true;
    c->d_ready = false;
    return GSA_OK;
}

int gsa_decoder_set_param(gsa_ctx* c, const char* name, const float* data, int32_t ndim, const int64_t* dims) {
    if (!c || !c->d_init) return fail(c, GSA_ERR_STATE, "gsa_decoder_init first");
    c->d_ready = false;
    return set_param(c, c->dparams, name, data, ndim, dims);
}

// conv bias + inference BatchNorm folded into one fma per element (canonical arithmetic; reference networks_seg.py:14-32,
// 68-76): s = gamma / sqrtf(running_var + 1e-5), k = fmaf(bias - running_mean, s, beta); use_bn=False: s = 1, k = bias
static int load_bn(gsa_ctx* c, const std::string& prefix, int C, const float* bias, float** s, float** k) {
    std::vector<float> hs((size_t)C, 1.0f), hk(bias, bias + C);
    if (c->d_bn) {
        const float *g, *b, *m, *v;
        NEED(c->dparams, prefix + ".gamma", (size_t)C, &g);
        NEED(c->dparams, prefix + ".beta", (size_t)C, &b);
        NEED(c->dparams, prefix + ".running_mean", (size_t)C, &m);
        NEED(c->dparams, prefix + ".running_var", (size_t)C, &v);
        for (int i = 0; i < C; ++i) {
            hs[i] = g[i] / std::sqrt(v[i] + 1e-5f);   // fp32: sqrtf, then one division
            hk[i] = std::fmaf(bias[i] - m[i], hs[i], b[i]);
        }
    }
    if (int rc = upload(c, hs, s, c->d_allocs)) return rc;
    return upload(c, hk, k, c->d_allocs);
}

int gsa_decoder_commit(gsa_ctx* c) {
    if (!c || !c->d_init) return fail(c, GSA_ERR_STATE, "gsa_decoder_init first");
    HIP_TRY(hipSetDevice(c->device));
    const auto& P = c->dparams;
    const int n = c->d_n;
    char nm[160];
    const float *w, *b;
    std::vector<float> h;
    HIP_TRY(hipDeviceSynchronize());
    c->d_ready = false;
    free_all(c->d_allocs);
    auto& T = c->d_allocs;
    for (int i = c->d_s0; i < n; ++i) {
        DecLevelDev& d = c->dl[i];
        d.F = c->d_feat[i]; d.I = c->d_inch[i];
        d.cs = c->d_feat[i + 1];
        d.in_c = d.F * (i > c->d_s0 ? 2 : 1);
        d.is_last = i == n - 1;
        snprintf(nm, sizeof nm, "cvt_block_%d.0", i);
        NEED(P, std::string(nm) + ".weight", (size_t)d.F * d.I * 9, &w);
        NEED(P, std::string(nm) + ".bias", (size_t)d.F, &b);
        h = pack_conv3(w, d.F, d.I, 1.0f, false, 1.0f);
        if (int rc = upload_mfma(c, h, &d.cvt_w, T)) return rc;
        d.cvt_u = d.b_u = nullptr;
        if (wino_layer(c, 4 << i, d.F)) {
            h = wino43_layer(c, 4 << i, d.I, d.F) ? pack_wino43(w, d.F, d.I, 1.0f, false, 1.0f) : pack_wino(w, d.F, d.I, 1.0f, false, 1.0f);
            if (int rc = upload(c, h, &d.cvt_u, T)) return rc;
        }
        snprintf(nm, sizeof nm, "cvt_block_%d.1", i);
        if (int rc = load_bn(c, nm, d.F, b, &d.cvt_s, &d.cvt_beta)) return rc;
        if (!d.is_last) {
            const int second = c->d_bn ? 3 : 2;
            const std::string pf = "main_block_" + std::to_string(i) + ".1.base_layers";
            NEED(P, pf + ".0.weight", (size_t)d.cs * d.in_c * 9, &w);
            NEED(P, pf + ".0.bias", (size_t)d.cs, &b);
            h = (8 << i) >= 16 ? pack_upconv(w, d.cs, d.in_c, 1.0f, false, 1.0f) : pack_conv3(w, d.cs, d.in_c, 1.0f, false, 1.0f);
            if (int rc = upload_mfma(c, h, &d.a_w, T)) return rc;
            if (int rc = load_bn(c, pf + ".1", d.cs, b, &d.a_s, &d.a_beta)) return rc;
            NEED(P, pf + "." + std::to_string(second) + ".weight", (size_t)d.cs * d.cs * 9, &w);
            NEED(P, pf + "." + std::to_string(second) + ".bias", (size_t)d.cs, &b);
            h = pack_conv3(w, d.cs, d.cs, 1.0f, false, 1.0f);
            if (int rc = upload_mfma(c, h, &d.b_w, T)) return rc;
            if (wino_layer(c, 8 << i, d.cs)) {
                h = pack_wino(w, d.cs, d.cs, 1.0f, false, 1.0f);      // conv b carries the residual: never the F(4x4,3x3) form (conv_uses_wino43)
                if (int rc = upload(c, h, &d.b_u, T)) return rc;
            }
            if (int rc = load_bn(c, pf + "." + std::to_string(second + 1), d.cs, b, &d.b_s, &d.b_beta)) return rc;
            d.has_sc = d.cs != d.in_c;
            if (d.has_sc) {
                const std::string sc = "main_block_" + std::to_string(i) + ".1.shortcut.0";
                NEED(P, sc + ".weight", (size_t)d.cs * d.in_c, &w);
                NEED(P, sc + ".bias", (size_t)d.cs, &b);
                h = pack_conv1(w, d.cs, d.in_c);
                if (int rc = upload_mfma(c, h, &d.sc_w, T)) return rc;
                h.assign(b, b + d.cs);
                if (int rc = upload(c, h, &d.sc_b, T)) return rc;
            }
        } else {
            const std::string pf = "main_block_" + std::to_string(i) + ".0";
            NEED(P, pf + ".weight", (size_t)d.cs * d.in_c * 9, &w);
            NEED(P, pf + ".bias", (size_t)d.cs, &b);
            h = pack_final(w, d.cs, d.in_c);
            if (int rc = upload(c, h, &d.f_w, T)) return rc;
            h.assign(b, b + d.cs);
            if (int rc = upload(c, h, &d.f_b, T)) return rc;
        }
    }
    c->d_ready = true;
    return GSA_OK;
}

// ------------------------------------------------------------------------ workspace

// Reads and clears the sticky device words (map_ctl[0]: instance-norm statistics out of range, map_ctl[1]: the mapping
// network's exchange timed out).  The device must be idle (the callers synchronise first).
static int read_device_status(gsa_ctx* c) {
    if (!c->map_ctl) return GSA_OK;
    unsigned w[2] = {0u, 0u};
    HIP_TRY(hipMemcpy(w, c->map_ctl, sizeof w, hipMemcpyDeviceToHost));
    if (!w[0] && !w[1]) return GSA_OK;
    HIP_TRY(hipMemset(c->map_ctl, 0, sizeof w));
    return fail(c, GSA_ERR_DEVICE, "device-side check failed since the last clean check:%s%s -- discard the results of those steps",
                w[1] ? " the fused mapping network timed out waiting for a partner workgroup (its workgroups were not co-resident)" : "",
                w[0] ? " an instance-norm statistic left the range of its 64-bit fixed-point sum (activations beyond rms ~1.4e3 at 1024^2; include/gsa.h)" : "");
}

int gsa_check(gsa_ctx* c) {
    if (!c) return GSA_ERR_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    return read_device_status(c);
}

int gsa_status_snapshot(gsa_ctx* c, void* stream, uint32_t* host_words) {
    if (!c || !host_words) return fail(c, GSA_ERR_INVALID, "gsa_status_snapshot: null argument");
    if (!c->map_ctl) {
        // no workspace yet: no step of this context has run, so there is nothing to report -- a replica of an in-process device list
        // that has not received a slice so far (ImageGenerator(gpu_ids=[a, b]) with one sample: split_sizes drops empty slices)
        host_words[0] = host_words[1] = 0u;
        return GSA_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(host_words, c->map_ctl, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return GSA_OK;
}

int gsa_debug_inject(gsa_ctx* c, int32_t kind, int32_t arg) {
    if (!c) return GSA_ERR_INVALID;
    // doubly gated: the explicit call AND GSA_TEST_HOOKS=1 in the environment of the process (the tests set it; neither alone arms anything)
    const char* hooks = getenv("GSA_TEST_HOOKS");
    if (!hooks || atoi(hooks) != 1) return fail(c, GSA_ERR_STATE, "gsa_debug_inject: test hooks are off (GSA_TEST_HOOKS=1 enables them)");
    switch (kind) {
    case 0: c->fault = 0; c->fault_range_after = -1; return GSA_OK;
    case 1: case 2: c->fault = kind; return GSA_OK;
    case 3: if (arg < 0) return fail(c, GSA_ERR_INVALID, "gsa_debug_inject(3): arg < 0"); c->fault_range_after = arg; return GSA_OK;
    default: return fail(c, GSA_ERR_INVALID, "gsa_debug_inject: unknown kind %d", kind);
    }
}

int gsa_reserve(gsa_ctx* c, int32_t max_batch) {
    if (!c) return GSA_ERR_INVALID;
    if (max_batch < 1) return fail(c, GSA_ERR_INVALID, "max_batch must be >= 1");
    if (!c->g_ready && !c->d_ready) return fail(c, GSA_ERR_STATE, "commit a generator or decoder before gsa_reserve");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = read_device_status(c)) return rc;      // a condition recorded by earlier steps is reported before its words are freed
    c->map_ctl = nullptr;
    free_all(c->ws_allocs);
    c->max_batch = 0;
    const size_t N = (size_t)max_batch;
    auto& T = c->ws_allocs;
    size_t prow_elems = 0;   // max over layers of rows*C
    if (c->g_ready) {
        const int L = c->gc.latent_size;
        for (int i = 0; i < 2; ++i)
            if (int rc = dev_alloc(c, N * L, &c->lat[i], T)) return rc;
        if (int rc = dev_alloc(c, N * c->style_cols, &c->styles, T)) return rc;
        if (int rc = dev_alloc(c, 2 + kMapSlices, &c->map_ctl, T)) return rc;
        HIP_TRY(hipMemset(c->map_ctl, 0, (2 + kMapSlices) * sizeof(unsigned)));
        for (int i = 0; i < 2; ++i) {
            if (int rc = dev_alloc(c, N * L, &c->map_ll[i], T)) return rc;
            HIP_TRY(hipMemset(c->map_ll[i], 0, N * L * sizeof(unsigned long long)));     // tag 0 never matches a launch
        }
        size_t maxact = 0;
        int maxC = 0;
        for (int l = 0; l < c->nlev; ++l) {
            const size_t R = (size_t)4 << l, C = (size_t)c->ch[l];
            maxact = std::max(maxact, R * R * C);
            maxC = std::max(maxC, c->ch[l]);
            if (int rc = dev_alloc(c, N * R * R * C, &c->x2[l], T)) return rc;
            if (int rc = dev_alloc(c, N * C, &c->aff2[l], T)) return rc;
            for (int nb = 1; nb <= max_batch; ++nb)
                prow_elems = std::max(prow_elems, (size_t)conv_stat_rows((int)R, (int)R, (int)C, nb) * C);
            prow_elems = std::max(prow_elems, (size_t)post_prow((int)R, (int)R, (int)C) * C);
        }
        if (int rc = dev_alloc(c, N * maxact, &c->t_raw, T)) return rc;
        if (int rc = dev_alloc(c, N * maxact, &c->x1, T)) return rc;
        if (int rc = dev_alloc(c, N * maxC, &c->aff1, T)) return rc;
        prow_elems = std::max(prow_elems, (size_t)64 * maxC);
        if (int rc = dev_alloc(c, N * prow_elems, &c->partials, T)) return rc;
        c->partials_bytes = N * prow_elems * sizeof(StatPart);
        c->stat_acc_bytes = N * maxC * sizeof(StatPart);
        c->ticket_bytes = N * ((maxC + 63) / 64) * sizeof(unsigned);
        c->stats_dirty = false;
        HIP_TRY(hipMemset(c->partials, 0, N * prow_elems * sizeof(StatPart)));     // all zero between layers: finalize_kernel clears what it read
        if (int rc = dev_alloc(c, N * maxC, &c->stat_acc, T)) return rc;
        HIP_TRY(hipMemset(c->stat_acc, 0, N * maxC * sizeof(StatPart)));
        if (int rc = dev_alloc(c, N * ((maxC + 63) / 64), &c->stat_tickets, T)) return rc;
        HIP_TRY(hipMemset(c->stat_tickets, 0, N * ((maxC + 63) / 64) * sizeof(unsigned)));
    }
    {
        if (int rc = dev_alloc(c, 16, &c->stamps, T)) return rc;
        HIP_TRY(hipMemset(c->stamps, 0, 16 * sizeof(unsigned long long)));
    }
    if (c->d_ready) {
        for (int i = c->d_s0; i < c->d_n; ++i) {
            const DecLevelDev& d = c->dl[i];
            const size_t R = (size_t)4 << i;
            if (int rc = dev_alloc(c, N * R * R * d.I, &c->din[i], T)) return rc;
            if (int rc = dev_alloc(c, N * R * R * d.F, &c->cvt[i], T)) return rc;
            if (!d.is_last) {
                if (int rc = dev_alloc(c, N * 4 * R * R * d.cs, &c->ya[i], T)) return rc;
                if (int rc = dev_alloc(c, N * 4 * R * R * d.cs, &c->prev[i], T)) return rc;
                if (d.has_sc)   // shortcut at the INPUT resolution when the sub-pixel kernel produces it (outputs >= 16 px)
                    if (int rc = dev_alloc(c, N * (2 * R >= 16 ? 1 : 4) * R * R * d.cs, &c->scb[i], T)) return rc;
            }
        }
    }
    c->max_batch = max_batch;
    return GSA_OK;
}

// ------------------------------------------------------------------------ forward passes

static int run_generator_pass(gsa_ctx* c, hipStream_t s, int n, const float* z, const float* const* noise, float* rgb,
                              uint8_t* img, float* const* feats, bool record_levels);

// The statistics rely on `partials` / `stat_acc` / the tickets being all zero between layers (the producers ADD to their rows,
// finalize_kernel clears what it read).  A pass that fails between a producer and its finalize (a launch error, a null noise
// plane) would leave rows dirty and every later pass of the context would add them into its instance-norm sums: the context
// remembers that a pass did not complete and the next one re-zeroes the three buffers on its stream first.
static int run_generator(gsa_ctx* c, hipStream_t s, int n, const float* z, const float* const* noise, float* rgb,
                         uint8_t* img, float* const* feats, bool record_levels = false) {
    if (c->stats_dirty) {
        HIP_TRY(hipMemsetAsync(c->partials, 0, c->partials_bytes, s));
        HIP_TRY(hipMemsetAsync(c->stat_acc, 0, c->stat_acc_bytes, s));
        HIP_TRY(hipMemsetAsync(c->stat_tickets, 0, c->ticket_bytes, s));
    }
    c->stats_dirty = true;
    const int rc = run_generator_pass(c, s, n, z, noise, rgb, img, feats, record_levels);
    if (rc == GSA_OK) c->stats_dirty = false;
    if (rc == GSA_OK && c->fault_range_after >= 0 && c->fault_range_after-- == 0) {
        // test hook: from this pass on the statistics-range word reads as if an instance norm had overflowed (it is sticky)
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c->map_ctl, 1, 1, s));
    }
    return rc;
}

static int run_generator_pass(gsa_ctx* c, hipStream_t s, int n, const float* z, const float* const* noise, float* rgb,
                              uint8_t* img, float* const* feats, bool record_levels) {
    const int L = c->gc.latent_size, nlev = c->nlev;
    const double N = n;
    // mapping network: PixelNorm, 8 x (dense + LeakyReLU)
    int cur = 0;
    if (mapping_fused(L, c->device)) {
        Launch lp(c, s, "mapping_kernel", "g.mapping", 3.0 * N * L + 16.0 * N * L * L, 4.0 * (8.0 * L * (double)L + 18 * N * L));
        HIP_TRY(launch_mapping(z, c->map_wt, c->map_b, c->map_ll, c->lat[0], c->map_ctl, n, L, c->device, s, c->fault == 1 ? 1 : 0));
    } else {
        { Launch lp(c, s, "pixelnorm_kernel", "g.mapping.pixelnorm", 3.0 * N * L, 8.0 * N * L);
          HIP_TRY(launch_pixelnorm(z, c->lat[0], n, L, s)); }
        for (int i = 0; i < 8; ++i) {
            Launch lp(c, s, "dense_kernel", "g.mapping.dense", 2.0 * N * L * L, 4.0 * (L * (double)L + 2 * N * L));
            HIP_TRY(launch_dense(c->lat[cur], c->map_wt[i], c->map_b[i], c->lat[cur ^ 1], n, L, L, 1, s));
            cur ^= 1;
        }
    }
    const float* w = c->lat[cur];
    { Launch lp(c, s, "styles_kernel", "g.styles", 2.0 * N * L * c->style_cols, 4.0 * ((double)L * c->style_cols + N * c->style_cols));
      HIP_TRY(launch_styles(w, c->latent_avg, c->psi, c->style_wt, c->style_b, c->style_col_layer, c->styles, n, L, c->style_cols, s)); }

    char layer[64];
    for (int l = 0; l < nlev; ++l) {
        const GenBlockDev& B = c->blk[l];
        const int R = B.R, C = B.C, Cin = B.Cin;
        const double px = N * R * R;
        for (int k = 0; k < 2; ++k) {
            const float* nz = noise[2 * l + k];
            if (!nz) return fail(c, GSA_ERR_INVALID, "noise plane %d is null", 2 * l + k);
            int prow = 0;
            if (k == 0) {
                PostParams pp{};
                pp.noise = nz; pp.nscale = B.nscale[0]; pp.nbias = B.nbias[0];
                pp.out = c->x1; pp.partials = c->partials; pp.H = R; pp.W = R; pp.C = C; pp.bf16 = c->bf16;
                if (!B.has_conv1) {
                    pp.src = c->constant; pp.src_per_sample = 0; pp.blur = nullptr;
                } else {
                    ConvParams cp{}; cp.stamps = c->stamps; cp.zeros = c->zeros; cp.dbg = c->dbg; cp.bf16 = c->bf16; cp.device = c->device; cp.prio = c->prio;
                    cp.src0 = c->x2[l - 1]; cp.aff0 = c->aff2[l - 1]; cp.C0 = Cin;
                    cp.Hs = R / 2; cp.Ws = R / 2; cp.H = R; cp.W = R;
                    cp.wpk = B.w1; cp.Cout = C; cp.out = c->t_raw;
                    if (B.is_deconv || R >= 16) {
                        // Deconvolution 4x4 s2, or nearest-x2 + conv3x3 in sub-pixel form (same kernel)
                        snprintf(layer, sizeof layer, B.is_deconv ? "g.%d.deconv_1" : "g.%d.conv_1", R);
                        static thread_local char kn[128];
                        snprintf(kn, sizeof kn, "%s", subpixel_kernel_name(cp, EPI_RAW, false, n));
                        // executed: 4 taps per output, or 9 products per 2x2 class outputs in the Winograd F(2x2,2x2) form; algorithmic: the
                        // reference's operator (16-tap transposed conv = 4 taps per output; 9-tap conv on the upsampled image)
                        Launch lp(c, s, kn, layer, 2.0 * px * C * Cin * (subpixel_uses_wino(cp) ? 2.25 : 4.0), 4.0 * (px / 4 * Cin + px * C), 2.0 * px * C * Cin * (B.is_deconv ? 4 : 9));
                        HIP_TRY(launch_subpixel(cp, EPI_RAW, false, n, s));
                    } else {
                        cp.up = 1;
                        snprintf(layer, sizeof layer, "g.%d.conv_1", R);
                        Launch lp(c, s, conv_kernel_name(cp, n, EPI_RAW, false), layer, 2.0 * px * C * Cin * 9, 4.0 * (px / 4 * Cin + px * C));
                        HIP_TRY(launch_conv3x3(cp, EPI_RAW, false, n, s));
                    }
                    pp.src = c->t_raw; pp.src_per_sample = 1; pp.blur = B.blur;
                }
                snprintf(layer, sizeof layer, "g.%d.post_1", R);
                if (post_fuses_finalize(pp)) {       // small plane: post pass + finalize in ONE launch (the workgroup holds the whole plane)
                    FinalizeParams fp{};
                    fp.partials = c->partials; fp.prow = 0; fp.HW = R * R; fp.C = C; fp.acc = c->stat_acc; fp.tickets = c->stat_tickets;
                    fp.style = c->styles + B.style_off[0]; fp.style_stride = c->style_cols;
                    fp.gamma = B.gamma[0]; fp.beta = B.beta[0];
                    fp.aff = c->aff1;
                    fp.flags = c->map_ctl;
                    Launch lp(c, s, pp.blur ? "post_fin_kernel<blur>" : "post_fin_kernel<const>", layer, 0.0, 4.0 * (2 * px * C + px));
                    HIP_TRY(launch_post_fin(pp, fp, n, s));
                    continue;
                }
                Launch lp(c, s, pp.blur ? "post_kernel<blur>" : "post_kernel<const>", layer, 0.0, 4.0 * (2 * px * C + px));
#ifdef GSA_DBG_HOOKS
                // diagnostic build only (`make dbg`, GSA_DBG bit 3; WRONG results): leave out the blur / noise / bias / LeakyReLU / statistics
                // pass of the 512^2 and 1024^2 levels -- what fusing it into the stride-2 convolution could win AT MOST, measured on the
                // overlapped step rather than priced from the serialized kernel times
                if (!((c->dbg & 8) && R >= 512))
#endif
                HIP_TRY(launch_post(pp, n, s));
                prow = post_rows_used(pp);
            } else {
                ConvParams cp{}; cp.stamps = c->stamps; cp.zeros = c->zeros; cp.dbg = c->dbg; cp.bf16 = c->bf16; cp.device = c->device; cp.prio = c->prio;
                cp.src0 = c->x1; cp.aff0 = c->aff1; cp.C0 = C;
                cp.Hs = R; cp.Ws = R; cp.H = R; cp.W = R;
                cp.wpk = B.w2; cp.wino = B.w2u; cp.Cout = C; cp.out = c->x2[l];
                cp.noise = nz; cp.nscale = B.nscale[1]; cp.nbias = B.nbias[1]; cp.partials = c->partials;
                snprintf(layer, sizeof layer, "g.%d.conv_2", R);
                int rows = conv_stat_rows(R, R, C, n);
                cp.stat_rows_host = &rows;            // the launcher reports the partial rows it used
                const bool fused_fin = conv_fuses_finalize(cp, EPI_SYNTH, false);
                if (fused_fin) {                      // the workgroup holds the whole plane: it writes the coefficients itself, no finalize launch
                    cp.fin_style = c->styles + B.style_off[1]; cp.fin_style_stride = c->style_cols;
                    cp.fin_gamma = B.gamma[1]; cp.fin_beta = B.beta[1]; cp.fin_aff = c->aff2[l]; cp.fin_flags = c->map_ctl;
                }
                Launch lp(c, s, conv_kernel_name(cp, n, EPI_SYNTH, false), layer, 2.0 * px * C * C * (conv_uses_wino43(cp, EPI_SYNTH, false) ? 2.25 : conv_uses_wino(cp, EPI_SYNTH, false) ? 4 : 9), 4.0 * (2 * px * C + px), 2.0 * px * C * C * 9);
                HIP_TRY(launch_conv3x3(cp, EPI_SYNTH, false, n, s));
                prow = rows;
                if (fused_fin) continue;
            }
            if (c->fault == 2 && l == 3 && k == 1) {      // fault injection (tests): ONE pass dies between a producer and its finalize (32 px conv_2: a level whose
                                                          // statistics still travel through partial rows -- the 4 / 8 px layers finalize in their producer)
                c->fault = 0;
                return fail(c, GSA_ERR_HIP, "injected fault (gsa_debug_inject 2) between a statistics producer and its finalize");
            }
            FinalizeParams fp{};
            fp.partials = c->partials; fp.prow = prow; fp.HW = R * R; fp.C = C; fp.acc = c->stat_acc; fp.tickets = c->stat_tickets;
            fp.style = c->styles + B.style_off[k]; fp.style_stride = c->style_cols;
            fp.gamma = B.gamma[k]; fp.beta = B.beta[k];
            fp.aff = k == 0 ? c->aff1 : c->aff2[l];
            fp.flags = c->map_ctl;      // word 0: statistics range check
            snprintf(layer, sizeof layer, "g.%d.finalize_%d", R, k + 1);
#ifdef GSA_DBG_HOOKS
            // diagnostic build only (GSA_DBG bit 5; WRONG results): without the finalize launches a fusion into their small producers would remove
            if ((c->dbg & 32) && ((k == 0 && R <= 32) || (k == 1 && R <= 8))) continue;
            if ((c->dbg & 64)) continue;      // bit 6: without ANY finalize launch (the bound of every such fusion)
#endif
            Launch lp(c, s, "finalize_kernel", layer, 0.0, 16.0 * N * prow * C);
            HIP_TRY(launch_finalize(fp, n, s));
        }
        if (record_levels) HIP_TRY(hipEventRecord(c->ev_level[l], s));   // feature l (x2, aff2) is complete
        if (feats && feats[l]) {
            snprintf(layer, sizeof layer, "g.%d.export", R);
            Launch lp(c, s, "export_nchw_kernel", layer, 0.0, 8.0 * px * C);
            HIP_TRY(launch_export_nchw(c->x2[l], c->aff2[l], feats[l], n, R, R, C, c->bf16, s));
        }
    }
    if (rgb || img) {
        const int l = nlev - 1, R = c->blk[l].R, C = c->blk[l].C, nc = c->gc.channels;
        const double px = N * R * R;
        Launch lp(c, s, C <= 16 ? "torgb_direct_kernel" : "torgb_kernel", "g.torgb", 2.0 * px * C * nc, px * (4.0 * C + (rgb ? 4.0 * nc : 0) + (img ? nc : 0)));
        HIP_TRY(launch_torgb(c->x2[l], c->aff2[l], c->rgb_w, c->rgb_b, rgb, img, n, R, R, C, nc, c->bf16, s));
    }
    return GSA_OK;
}

// feats_nhwc[i] / feat_aff[i]: decoder inputs in kernel layout (aff may be null)
// the cvt convolution of decoder level i (networks_seg.py:64-79): conv3x3 + bias -> BN -> LeakyReLU on feature i
static ConvParams cvt_params(gsa_ctx* c, int i, const float* src, const Aff* aff) {
    const DecLevelDev& d = c->dl[i];
    const int R = 4 << i;
    ConvParams cp{}; cp.stamps = c->stamps; cp.zeros = c->zeros; cp.dbg = c->dbg; cp.bf16 = c->bf16; cp.device = c->device; cp.prio = c->prio;
    cp.src0 = src; cp.aff0 = aff; cp.C0 = d.I;
    cp.Hs = R; cp.Ws = R; cp.H = R; cp.W = R;
    cp.wpk = d.cvt_w; cp.wino = d.cvt_u; cp.Cout = d.F; cp.out = c->cvt[i];
    cp.bn_s = d.cvt_s; cp.bn_beta = d.cvt_beta;
    return cp;
}

// feats_nhwc[i] / feat_aff[i]: decoder inputs in kernel layout (aff may be null).  rgb_img (gsa_generate only, the caller checked
// wino_lean_fuses_torgb): the LAST level's cvt convolution also writes toRGB's uint8 image -- it stages the very tensor toRGB reads.
static int run_decoder(gsa_ctx* c, hipStream_t s, int n, const float* const* fsrc, const Aff* const* faff, float* logits,
                       uint8_t* mask, int i_begin = 0, int i_end = -1, bool wait_levels = false, uint8_t* rgb_img = nullptr) {
    const int nl = i_end < 0 ? c->d_n : i_end;
    const double N = n;
    char layer[64];
    const int s0 = c->d_s0;   // levels below start_res have no blocks
    for (int i = std::max(i_begin, s0); i < nl; ++i) {
        const DecLevelDev& d = c->dl[i];
        const int R = 4 << i;
        const double px = N * R * R;
        if (wait_levels) HIP_TRY(hipStreamWaitEvent(s, c->ev_level[i], 0));   // generator feature i is ready
        {   // cvt_block: conv3x3+bias -> BN -> LeakyReLU (Dropout is identity at inference)
            ConvParams cp = cvt_params(c, i, fsrc[i], faff ? faff[i] : nullptr);
            const bool with_rgb = rgb_img != nullptr && i == c->d_n - 1;
            const int nc = c->gc.channels;
            if (with_rgb) { cp.rgb_w = c->rgb_w; cp.rgb_b = c->rgb_b; cp.rgb_img = rgb_img; }
            snprintf(layer, sizeof layer, with_rgb ? "d.cvt_%d+torgb" : "d.cvt_%d", i);
            Launch lp(c, s, conv_kernel_name(cp, n, EPI_DEC, false), layer,
                      2.0 * px * d.F * d.I * (conv_uses_wino43(cp, EPI_DEC, false) ? 2.25 : conv_uses_wino(cp, EPI_DEC, false) ? 4 : 9) + (with_rgb ? 2.0 * px * d.I * nc : 0.0),
                      4.0 * px * (d.I + d.F) + (with_rgb ? px * nc : 0.0), 2.0 * px * d.F * d.I * 9 + (with_rgb ? 2.0 * px * d.I * nc : 0.0));
            HIP_TRY(launch_conv3x3(cp, EPI_DEC, false, n, s));
        }
        if (!d.is_last) {
            const int R2 = 2 * R;
            const double px2 = 4 * px;
            {   // ResBlock conv a (+ fused 1x1 shortcut) on nearest-x2(concat(prev, cvt))
                ConvParams cp{}; cp.stamps = c->stamps; cp.zeros = c->zeros; cp.dbg = c->dbg; cp.bf16 = c->bf16; cp.device = c->device; cp.prio = c->prio;
                if (i > s0) { cp.src0 = c->prev[i - 1]; cp.C0 = d.F; cp.src1 = c->cvt[i]; cp.C1 = d.F; }
                else { cp.src0 = c->cvt[i]; cp.C0 = d.F; }
                cp.Hs = R; cp.Ws = R; cp.up = 1; cp.H = R2; cp.W = R2;
                cp.wpk = d.a_w; cp.Cout = d.cs; cp.out = c->ya[i];
                cp.bn_s = d.a_s; cp.bn_beta = d.a_beta;
                if (d.has_sc) { cp.wsc = d.sc_w; cp.sc_bias = d.sc_b; cp.out_sc = c->scb[i]; }
                snprintf(layer, sizeof layer, "d.main_%d.a", i);
                if (R2 >= 16) {   // sub-pixel form: 4 taps per output instead of 9
                    static thread_local char kn[128];
                    snprintf(kn, sizeof kn, "%s", subpixel_kernel_name(cp, EPI_DEC, d.has_sc, n));
                    Launch lp(c, s, kn, layer, 2.0 * px2 * d.cs * d.in_c * (subpixel_uses_wino(cp) ? 2.25 : 4.0) + (d.has_sc ? 2.0 * px * d.cs * d.in_c : 0.0),
                              4.0 * (px * d.in_c + px2 * d.cs + (d.has_sc ? px * d.cs : 0.0)), 2.0 * px2 * d.cs * d.in_c * (9 + (d.has_sc ? 1 : 0)));
                    cp.up = 0;
                    HIP_TRY(launch_subpixel(cp, EPI_DEC, d.has_sc, n, s));
                } else {
                    Launch lp(c, s, conv_kernel_name(cp, n, EPI_DEC, d.has_sc), layer,
                              2.0 * px2 * d.cs * d.in_c * (9 + (d.has_sc ? 1 : 0)), 4.0 * (px * d.in_c + px2 * d.cs * (d.has_sc ? 2 : 1)));
                    HIP_TRY(launch_conv3x3(cp, EPI_DEC, d.has_sc, n, s));
                }
            }
            {   // ResBlock conv b, + shortcut
                ConvParams cp{}; cp.stamps = c->stamps; cp.zeros = c->zeros; cp.dbg = c->dbg; cp.bf16 = c->bf16; cp.device = c->device; cp.prio = c->prio;
                cp.src0 = c->ya[i]; cp.C0 = d.cs;
                cp.Hs = R2; cp.Ws = R2; cp.H = R2; cp.W = R2;
                cp.wpk = d.b_w; cp.wino = d.b_u; cp.Cout = d.cs; cp.out = c->prev[i];
                cp.bn_s = d.b_s; cp.bn_beta = d.b_beta;
                if (d.has_sc) { cp.resid = c->scb[i]; cp.resid_up = R2 >= 16 ? 1 : 0; }   // sub-pixel conv a stores the shortcut at input resolution
                else if (i == s0) { cp.resid = c->cvt[i]; cp.resid_up = 1; }   // identity shortcut: the upsampled input itself
                else { cp.resid = c->prev[i - 1]; cp.resid1 = c->cvt[i]; cp.res_c0 = d.F; cp.resid_up = 1; }   // ... over concat(prev, cvt)
                snprintf(layer, sizeof layer, "d.main_%d.b", i);
                Launch lp(c, s, conv_kernel_name(cp, n, EPI_DEC, false), layer, 2.0 * px2 * d.cs * d.cs * (conv_uses_wino43(cp, EPI_DEC, false) ? 2.25 : conv_uses_wino(cp, EPI_DEC, false) ? 4 : 9), 4.0 * px2 * d.cs * 3, 2.0 * px2 * d.cs * d.cs * 9);
                HIP_TRY(launch_conv3x3(cp, EPI_DEC, false, n, s));
            }
        } else {
            snprintf(layer, sizeof layer, "d.final_%d", i);
            Launch lp(c, s, "final_conv_kernel", layer, 2.0 * px * d.cs * d.in_c * 9, px * (4.0 * d.in_c + (logits ? 4.0 * d.cs : 0) + (mask ? 1 : 0)));
            HIP_TRY(launch_final_conv(i > s0 ? c->prev[i - 1] : nullptr, i > s0 ? d.F : 0, c->cvt[i], d.F, d.f_w, d.f_b, logits, mask, n, R, R, d.cs, c->bf16, s));
        }
    }
    return GSA_OK;
}

static int check_batch(gsa_ctx* c, int n) {
    if (n < 1) return fail(c, GSA_ERR_INVALID, "batch size %d < 1", n);
    if (n > c->max_batch) return fail(c, GSA_ERR_STATE, "batch %d exceeds the reserved workspace (%d): call gsa_reserve", n, c->max_batch);
    return GSA_OK;
}

int gsa_generator_forward(gsa_ctx* c, void* stream, int32_t n, const float* z, const float* const* noise, int32_t num_noise,
                          float* rgb, uint8_t* img, float* const* feats, int32_t num_feats) {
    if (!c) return GSA_ERR_INVALID;
    if (!c->g_ready) return fail(c, GSA_ERR_STATE, "gsa_generator_commit first");
    if (!z || !noise) return fail(c, GSA_ERR_INVALID, "z and noise must not be null");
    if (num_noise != 2 * c->nlev) return fail(c, GSA_ERR_INVALID, "%d noise planes passed, this generator has %d", num_noise, 2 * c->nlev);
    if (feats && num_feats != c->nlev) return fail(c, GSA_ERR_INVALID, "%d feature pointers passed, this generator yields %d", num_feats, c->nlev);
    if (int rc = check_batch(c, n)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    return run_generator(c, (hipStream_t)stream, n, z, noise, rgb, img, feats);
}

int gsa_decoder_forward(gsa_ctx* c, void* stream, int32_t n, const float* const* feats, int32_t num_feats, float* logits, uint8_t* mask) {
    if (!c) return GSA_ERR_INVALID;
    if (!c->d_ready) return fail(c, GSA_ERR_STATE, "gsa_decoder_commit first");
    if (!feats) return fail(c, GSA_ERR_INVALID, "feats must not be null");
    if (num_feats != c->d_n) return fail(c, GSA_ERR_INVALID, "%d feature pointers passed, this decoder takes %d", num_feats, c->d_n);
    if (int rc = check_batch(c, n)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const float* fsrc[kMaxLevels];
    for (int i = c->d_s0; i < c->d_n; ++i) {   // entries below start_res are not read (they may be null)
        if (!feats[i]) return fail(c, GSA_ERR_INVALID, "feature %d is null", i);
        const int R = 4 << i;
        Launch lp(c, s, "import_nhwc_kernel", "d.import", 0.0, 8.0 * n * R * R * c->dl[i].I);
        HIP_TRY(launch_import_nhwc(feats[i], c->din[i], n, R, R, c->dl[i].I, c->bf16, s));
        fsrc[i] = c->din[i];
    }
    return run_decoder(c, s, n, fsrc, nullptr, logits, mask);
}

int gsa_generate(gsa_ctx* c, void* stream, int32_t n, const float* z, const float* const* noise, int32_t num_noise, uint8_t* img,
                 uint8_t* mask) {
    if (!c) return GSA_ERR_INVALID;
    if (!c->g_ready || !c->d_ready) return fail(c, GSA_ERR_STATE, "commit both generator and decoder first");
    if (!z || !noise) return fail(c, GSA_ERR_INVALID, "z and noise must not be null");
    if (num_noise != 2 * c->nlev) return fail(c, GSA_ERR_INVALID, "%d noise planes passed, this generator has %d", num_noise, 2 * c->nlev);
    if (c->d_n != c->nlev) return fail(c, GSA_ERR_INVALID, "decoder expects %d features, the generator yields %d", c->d_n, c->nlev);
    for (int l = 0; l < c->nlev; ++l)
        if (c->d_inch[l] != c->ch[l]) return fail(c, GSA_ERR_INVALID, "decoder in_channels[%d]=%d but the generator feature has %d", l, c->d_inch[l], c->ch[l]);
    if (int rc = check_batch(c, n)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const float* fsrc[kMaxLevels];
    const Aff* faff[kMaxLevels];
    for (int l = 0; l < c->nlev; ++l) { fsrc[l] = c->x2[l]; faff[l] = c->aff2[l]; }
    // Decoder level i only needs the generator feature of level i, so the decoder can run on a second
    // stream beside the synthesis of the higher levels (fork/join through events, no host
    // synchronisation, graph-capturable): its short, latency-bound low-resolution kernels then hide behind
    // the synthesis kernels and the tails of either side are filled by the other.
    // Measured on MI355X, round-2 kernels (Winograd / K-split forms: shorter kernels with more
    // non-MFMA time for a neighbour to fill): FFHQ fp32 +18 % at 1 sample, +5.7 % at 4, +2.3 % at 8 -- so the decoder
    // levels below the last always go to the side stream (gsa_set_overlap / GSA_SIDE_LEVELS override it).
    const int want = c->side_levels < 0 ? c->d_n - 1 : c->side_levels;
    const int ns = want > c->d_n - 1 ? c->d_n - 1 : want;
    if (ns > 0) {
        HIP_TRY(hipEventRecord(c->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    }
    // toRGB reads the generator's last feature with its AdaIN coefficients -- exactly what the decoder's last cvt convolution stages.  Where
    // that convolution is the lean 16 -> 16 kernel it writes the uint8 image as well (round 5): one 0.5 GB read per FFHQ batch of 8 less.
    // The last decoder level always runs on the caller's stream, behind the generator.
    bool fuse_rgb = false;
    if (img && ns <= c->d_n - 1) {
        const int last = c->d_n - 1;
        ConvParams cp = cvt_params(c, last, fsrc[last], faff[last]);
        fuse_rgb = conv_fuses_torgb(cp, EPI_DEC, false, c->gc.channels);
    }
    if (int rc = run_generator(c, s, n, z, noise, nullptr, fuse_rgb ? nullptr : img, nullptr, ns > 0)) return rc;
    if (ns > 0) {
        if (int rc = run_decoder(c, c->side, n, fsrc, faff, nullptr, nullptr, 0, ns, true)) return rc;
        HIP_TRY(hipEventRecord(c->ev_join, c->side));
        HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    return run_decoder(c, s, n, fsrc, faff, nullptr, mask, ns, -1, false, fuse_rgb ? img : nullptr);
}

int gsa_fill_inputs(gsa_ctx* c, void* stream, int32_t n, uint64_t seed, uint64_t first_index, float* z, float* const* noise,
                    int32_t num_noise) {
    if (!c) return GSA_ERR_INVALID;
    if (!c->g_init) return fail(c, GSA_ERR_STATE, "gsa_generator_init first");
    if (n <= 0 || (!z && !noise)) return fail(c, GSA_ERR_INVALID, "gsa_fill_inputs: bad argument");
    if (noise && num_noise != 2 * (c->gc.max_res_log2 - 1))
        return fail(c, GSA_ERR_INVALID, "%d noise planes passed, this generator has %d", num_noise, 2 * (c->gc.max_res_log2 - 1));
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (z) HIP_TRY(launch_fill_normal(z, c->gc.latent_size, n, first_index, 0xFFFFu, seed, s));
    if (noise) {
        const int planes = 2 * (c->gc.max_res_log2 - 1);
        for (int l = 0; l < planes; ++l) {
            if (!noise[l]) return fail(c, GSA_ERR_INVALID, "gsa_fill_inputs: null noise plane");
            const int R = 4 << (l / 2);
            HIP_TRY(launch_fill_normal(noise[l], R * R, n, first_index, (unsigned)l, seed, s));
        }
    }
    return GSA_OK;
}

int gsa_segmentation_eval(gsa_ctx* c, void* stream, int32_t n, int32_t classes, int32_t H, int32_t W, const float* logits,
                          const int8_t* labels, uint64_t* confusion, uint64_t* loss_fixed) {
    if (!c) return GSA_ERR_INVALID;
    if (n <= 0 || H <= 0 || W <= 0 || !logits || !labels || !confusion || !loss_fixed)
        return fail(c, GSA_ERR_INVALID, "gsa_segmentation_eval: bad argument");
    if (classes < 2 || classes > 8) return fail(c, GSA_ERR_INVALID, "gsa_segmentation_eval: 2..8 classes supported");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_seg_eval(logits, labels, n, classes, H, W, reinterpret_cast<unsigned long long*>(confusion),
                            reinterpret_cast<unsigned long long*>(loss_fixed), (hipStream_t)stream));
    return GSA_OK;
}

int gsa_set_overlap(gsa_ctx* c, int32_t levels) {
    if (!c) return GSA_ERR_INVALID;
    c->side_levels = levels < 0 ? -1 : levels;
    return GSA_OK;
}

int gsa_set_precision(gsa_ctx* c, int32_t mode) {
    if (!c) return GSA_ERR_INVALID;
    if (mode != GSA_PREC_F32 && mode != GSA_PREC_BF16) return fail(c, GSA_ERR_INVALID, "gsa_set_precision: unknown mode");
    if ((c->g_ready || c->d_ready) && mode != c->bf16)
        return fail(c, GSA_ERR_STATE, "gsa_set_precision must precede gsa_generator_commit / gsa_decoder_commit (weights are packed per mode)");
    c->bf16 = mode;
    return GSA_OK;
}

// ------------------------------------------------------------------------ measurement hooks

int gsa_profile_enable(gsa_ctx* c, int32_t on) {
    if (!c) return GSA_ERR_INVALID;
    c->prof = on;
    return GSA_OK;
}

int gsa_profile_collect(gsa_ctx* c) {
    if (!c) return GSA_ERR_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    for (auto& ev : c->prof_events) {
        HIP_TRY(hipEventSynchronize(ev.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        c->prof_entries[ev.entry].ms += ms;
        c->event_pool.push_back(ev.a);
        c->event_pool.push_back(ev.b);
    }
    c->prof_events.clear();
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = read_device_status(c)) return rc;
    return (int)c->prof_entries.size();
}

int gsa_profile_entry(gsa_ctx* c, int32_t i, const char** name, double* ms, int64_t* launches, double* flops, double* bytes,
                      double* alg_flops) {
    if (!c || i < 0 || i >= (int)c->prof_entries.size()) return GSA_ERR_INVALID;
    const ProfEntry& e = c->prof_entries[i];
    if (name) *name = e.name.c_str();
    if (ms) *ms = e.ms;
    if (launches) *launches = e.launches;
    if (flops) *flops = e.flops;
    if (bytes) *bytes = e.bytes;
    if (alg_flops) *alg_flops = e.alg_flops;
    return GSA_OK;
}

int gsa_profile_reset(gsa_ctx* c) {
    if (!c) return GSA_ERR_INVALID;
    int rc = gsa_profile_collect(c);
    if (rc < 0) return rc;
    c->prof_entries.clear();
    return GSA_OK;
}

}  // extern "C"
