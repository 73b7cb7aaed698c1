/*
 * gsa_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Explicit-loop CPU restatement of the reference's `generate` hot path in the CANONICAL
 * fp32 evaluation order of this project (DESIGN.md "Canonical arithmetic").  The HIP
 * kernels are required to reproduce its outputs BIT FOR BIT; oracle/ref_semantic.py
 * (torch functionals, the reference's own op order) pins it to within rounding.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path never does.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no tests or golden vectors and its
 * arithmetic lives in Apache MXNet 1.5.1 (reference README.md:9), absent here.
 *
 * What is restated (reference file:line):
 *   mapping, PixelNorm, DenseW ........ networks_stylegan.py:128-139, 479-524, 558-565
 *   truncation lerp ................... networks_stylegan.py:158-163, 170-189
 *   StyleGeneratorBlock ............... networks_stylegan.py:6-73
 *   Conv2DW / Conv2DTransposeW ........ networks_stylegan.py:354-476 (weight*std*lr_mult :407-412)
 *   Blur .............................. networks_stylegan.py:200-236
 *   AddNoise, Bias, LeakyReLU(0.2) .... networks_stylegan.py:267-305, 534-545, 40, 51
 *   AdaIN (DenseW affine + InstanceNorm) networks_stylegan.py:239-264
 *   toRGB ............................. networks_stylegan.py:118-126, 194-195
 *   _transform_gan_back ............... image_generator.py:76-84
 *   Decoder, DecoderResBlock .......... networks_seg.py:7-113
 *   argmax of SegSolver.predict ....... seg_solver.py:326
 *
 * Canonical arithmetic (all fp32 round-to-nearest, no contraction except explicit fmaf):
 *   up+conv (nearest x2 then 3x3, outputs >= 16 px): sub-pixel form, see pack_upconv()
 *   conv   acc=0; for cb in Cin/16: for ky: for kx: for c in (0,4,8,12,1,5,9,13,2,6,10,14,3,7,11,15): acc=fmaf(in[16cb+c],w,acc)
 *          (channels in blocks of 16, the taps inside a block; out-of-image taps skipped,
 *          which equals adding +0; 1x1 convs and dense layers are plain k-ordered chains)
 *   stats  per (n,c), per aligned quad of 4 consecutive x:
 *          s=(v0+v1)+(v2+v3), q=(v0*v0+v1*v1)+(v2*v2+v3*v3) (every product and sum rounded);
 *          I1 += rint(s*2^28), I2 += rint(q*2^20) as wrapping 64-bit integers
 *          (order independent, hence tiling independent)
 *   AdaIN  out = fmaf(x - mean, A, B) with A,B from finalize() below
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GSAO_API __attribute__((visibility("default")))

enum { GSA_OK = 0, GSA_ERR_INVALID = -1, GSA_ERR_STATE = -2, GSA_ERR_MISSING_PARAM = -3, GSA_ERR_NOMEM = -5 };
enum { GSA_PREC_F32 = 0, GSA_PREC_BF16 = 1 };

typedef struct {
    int32_t max_res_log2, fmap_base;
    double fmap_decay;
    int32_t fmap_max, latent_size, channels, use_wscale;
} gsa_generator_config;

typedef struct {
    int32_t num_feats, start_res, use_bn;
    const int32_t* features;
    const int32_t* in_channels;
} gsa_decoder_config;

#define MAX_LEVELS 12
#define MAX_PARAMS 512

typedef struct {
    char name[96];
    float* data;
    int64_t count;
    int ndim;
    int64_t dims[6];
} raw_param;

typedef struct {
    raw_param p[MAX_PARAMS];
    int n;
} param_table;

typedef struct {
    int C, Cin, R;
    int has_conv1, is_deconv;
    float* w1;      /* packed conv_1 / deconv_1 (sub-pixel deconv packing when R >= 16) */
    float* blur;    /* [C][9] */
    float* nscale[2];
    float* nbias[2];
    float* w2;      /* packed conv_2 */
    float* w2u;     /* conv_2 in Winograd form (pack_wino), used when use_wino() */
    float* aff_w[2];/* effective (2C, L) row-major */
    float* aff_b[2];
    float* gamma[2];
    float* beta[2];
} gen_block;

typedef struct {
    int F, I;        /* cvt: I -> F */
    float* cvt_w;    /* packed */
    float* cvt_u;    /* Winograd form */
    float* cvt_b; float *cvt_s, *cvt_rm, *cvt_beta; /* bias; BN scale, running_mean, beta */
    int in_c, cs;    /* main block */
    int is_last, has_sc;
    float *a_w, *a_b, *a_s, *a_rm, *a_beta;
    float *b_w, *b_b, *b_s, *b_rm, *b_beta;
    float* b_u;      /* conv b in Winograd form */
    float *sc_w, *sc_b;  /* sc_w packed [c][o] */
    float *f_w, *f_b;    /* final conv packed */
} dec_level;

typedef struct gsao_ctx {
    char err[512];
    /* generator */
    int g_init, g_ready;
    gsa_generator_config gc;
    param_table gp;
    int nlev;               /* max_res_log2 - 1 */
    int ch[MAX_LEVELS];     /* channels per level */
    float* map_w[8];        /* effective (L,L) */
    float* map_b[8];
    float* latent_avg; float* psi; float* constant; /* constant as NHWC [4][4][C] */
    gen_block blk[MAX_LEVELS];
    float* rgb_w; float* rgb_b;  /* [ch][C] effective, bias */
    /* decoder */
    int bf16;               /* gsao_set_precision: operands of the MFMA convolutions rounded to bf16 */
    int d_init, d_ready;
    int d_n, d_s0, d_bn, d_feat[MAX_LEVELS + 1], d_inch[MAX_LEVELS];
    param_table dp;
    dec_level dl[MAX_LEVELS];
} gsao_ctx;

static char g_err[512];

static int fail(gsao_ctx* c, int code, const char* fmt, const char* a, long b) {
    char* dst = c ? c->err : g_err;
    snprintf(dst, 512, fmt, a, b);
    return code;
}

/* ---------------------------------------------------------------- parameters */

static raw_param* find_param(param_table* t, const char* name) {
    for (int i = 0; i < t->n; ++i)
        if (!strcmp(t->p[i].name, name)) return &t->p[i];
    return NULL;
}

static int put_param(gsao_ctx* c, param_table* t, const char* name, const float* data, int ndim,
                     const int64_t* dims) {
    if (ndim < 0 || ndim > 6 || strlen(name) >= 96) return fail(c, GSA_ERR_INVALID, "bad parameter %s (%ld)", name, ndim);
    raw_param* p = find_param(t, name);
    if (!p) {
        if (t->n >= MAX_PARAMS) return fail(c, GSA_ERR_NOMEM, "too many parameters at %s (%ld)", name, t->n);
        p = &t->p[t->n++];
        memset(p, 0, sizeof *p);
        strcpy(p->name, name);
    }
    int64_t cnt = 1;
    for (int i = 0; i < ndim; ++i) { p->dims[i] = dims[i]; cnt *= dims[i]; }
    p->ndim = ndim;
    p->count = cnt;
    free(p->data);
    p->data = (float*)malloc(sizeof(float) * (size_t)(cnt > 0 ? cnt : 1));
    if (!p->data) return fail(c, GSA_ERR_NOMEM, "out of memory for %s (%ld)", name, cnt);
    memcpy(p->data, data, sizeof(float) * (size_t)cnt);
    return GSA_OK;
}

static void free_table(param_table* t) {
    for (int i = 0; i < t->n; ++i) free(t->p[i].data);
    t->n = 0;
}

static int nf(const gsa_generator_config* g, int r) {
    /* reference networks_stylegan.py:114-116 */
    int fmaps = (int)(g->fmap_base / pow(2.0, (r - 1) * g->fmap_decay));
    return fmaps < g->fmap_max ? fmaps : g->fmap_max;
}

/* (W*std)*lr_mult, reference networks_stylegan.py:407-412 / 513-518: two fp32 roundings */
static inline float eff(float w, float std, int use_std, float lr) {
    float v = use_std ? w * std : w;
    return v * lr;
}

/* conv OIHW (O,I,K,K) -> packed [(cb*K*K + tap)*CB + c][O]; I % CB == 0 */
#define CB 16
/* canonical order of the 16 channels of a block inside the MFMA convolutions: 0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15
 * (MFMA j of a (tap, block) multiplies the channels {j, 4+j, 8+j, 12+j}: k slot kq <-> channel 4kq+j, the natural
 * 16-byte chunks of the NHWC tensors) */
#define CPERM(k) ((((k) & 3) << 2) | ((k) >> 2))
static float* pack_conv(const float* w, int O, int I, int K, float std, int use_std, float lr) {
    float* out = (float*)malloc(sizeof(float) * (size_t)O * I * K * K);
    for (int cb = 0; cb < I / CB; ++cb)
        for (int t = 0; t < K * K; ++t)
            for (int ci = 0; ci < CB; ++ci)
                for (int o = 0; o < O; ++o)
                    out[(((size_t)cb * K * K + t) * CB + ci) * O + o] =
                        eff(w[((size_t)o * I + cb * CB + ci) * K * K + t], std, use_std, lr);
    return out;
}

/* deconv IOHW (I,O,4,4) -> packed [(cb*16 + tap)*CB + c][O] */
static float* pack_deconv(const float* w, int I, int O, float std, int use_std, float lr) {
    float* out = (float*)malloc(sizeof(float) * (size_t)O * I * 16);
    for (int cb = 0; cb < I / CB; ++cb)
        for (int t = 0; t < 16; ++t)
            for (int ci = 0; ci < CB; ++ci)
                for (int o = 0; o < O; ++o)
                    out[(((size_t)cb * 16 + t) * CB + ci) * O + o] =
                        eff(w[((size_t)(cb * CB + ci) * O + o) * 16 + t], std, use_std, lr);
    return out;
}

/* nearest-x2 upsample followed by a 3x3 conv (reference networks_stylegan.py:22-27,
 * networks_seg.py:86-88) in its SUB-PIXEL form: the output pixel (2y+dy, 2x+dx) only sees the
 * 2x2 input pixels {y-1+dy, y+dy} x {x-1+dx, x+dx}, with the 3x3 taps that fall on the same
 * input pixel pre-summed.  That is exactly a stride-2 transposed conv with the 4x4 kernel
 *   Wd[a][b] = sum_{ky in S(a)} sum_{kx in S(b)} W[ky][kx],  S(0)={2} S(1)={1,2} S(2)={0,1} S(3)={0}
 * (fp32 sums, ky then kx ascending, left to right), 2.25x fewer MACs than the 9-tap form.
 * Used for outputs of 16 px and larger; exact algebra, rounding differs by a few ulp.
 * conv OIHW (O,I,3,3) -> deconv packing [(cb*16 + tap16)*CB + c][O] */
static float* pack_upconv(const float* w, int O, int I, float std, int use_std, float lr) {
    static const int S[4][2] = {{2, -1}, {1, 2}, {0, 1}, {0, -1}};
    float* out = (float*)malloc(sizeof(float) * (size_t)O * I * 16);
    for (int cb = 0; cb < I / CB; ++cb)
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
                for (int ci = 0; ci < CB; ++ci)
                    for (int o = 0; o < O; ++o) {
                        const float* wk = w + ((size_t)o * I + cb * CB + ci) * 9;
                        float sum = 0.0f;
                        int first = 1;
                        for (int i = 0; i < 2; ++i)
                            for (int j = 0; j < 2; ++j) {
                                if (S[a][i] < 0 || S[b][j] < 0) continue;
                                const float e = eff(wk[S[a][i] * 3 + S[b][j]], std, use_std, lr);
                                sum = first ? e : sum + e;
                                first = 0;
                            }
                        out[(((size_t)cb * 16 + a * 4 + b) * CB + ci) * O + o] = sum;
                    }
    return out;
}

/* ---- Winograd F(2x2, 3x3) form of the plain 3x3 convolutions (Lavin & Gray 2016) -------------------------------
 * Rule (static, by layer shape only -- never by batch size): a 3x3 stride-1 convolution WITHOUT upsample-on-read
 * whose output is >= 64 px on a side (or >= 32 px with at least 64 output channels) is evaluated, in fp32 mode, as
 *     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          per 2x2 output tile, 16 products per (tile, c, o)
 * instead of 36: 2.25x fewer multiplications.  Canonical arithmetic (every fp32 op rounded, nothing contracted):
 *   weights    U = G g G^T evaluated in DOUBLE on the effective fp32 weights, rounded to fp32 once;
 *   input      t = B^T d (rows), V = t B (columns):  t0=d0-d2, t1=d1+d2, t2=d2-d1, t3=d1-d3  (d = AdaIN-applied,
 *              zero outside the image), the same four forms along the columns;
 *   products   M[f] = fmaf chain over the input channels (16-channel blocks ascending, CPERM order inside), per
 *              frequency f = 4i+j separately -- bitwise what v_mfma_f32_16x16x4_f32 produces;
 *   output     s0=(M0+M1)+M2, s1=(M1-M2)-M3 along the rows, the same two forms along the columns.
 * The result differs from the 9-tap chain by a few fp32 ulps of the intermediate magnitudes; oracle/ref_semantic.py
 * keeps the reference's 9-tap order and bounds the difference (tests: <= 1e-3 on rgb and logits at full size). */
static int g_wino_enabled = -1;
static int use_wino(int H, int W, int Cout, int up, int bf) {
    if (g_wino_enabled < 0) { const char* e = getenv("GSAO_WINO"); g_wino_enabled = !(e && atoi(e) == 0); }
    return g_wino_enabled && !bf && !up && H == W && (H >= 64 || (H >= 32 && Cout >= 64) || (H >= 16 && Cout >= 256)) && H % 16 == 0;
}

/* conv OIHW (O,I,3,3) -> U packed [(cb*16 + f)*CB + c][O], f = 4*i + j */
static float* pack_wino(const float* w, int O, int I, float std, int use_std, float lr) {
    float* out = (float*)malloc(sizeof(float) * (size_t)O * I * 16);
    for (int cb = 0; cb < I / CB; ++cb)
        for (int ci = 0; ci < CB; ++ci)
            for (int o = 0; o < O; ++o) {
                const float* wk = w + ((size_t)o * I + cb * CB + ci) * 9;
                double g[3][3], r[4][3], u[4][4];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) g[a][b] = (double)eff(wk[a * 3 + b], std, use_std, lr);
                for (int b = 0; b < 3; ++b) {
                    r[0][b] = g[0][b];
                    r[1][b] = 0.5 * ((g[0][b] + g[1][b]) + g[2][b]);
                    r[2][b] = 0.5 * ((g[0][b] - g[1][b]) + g[2][b]);
                    r[3][b] = g[2][b];
                }
                for (int a = 0; a < 4; ++a) {
                    u[a][0] = r[a][0];
                    u[a][1] = 0.5 * ((r[a][0] + r[a][1]) + r[a][2]);
                    u[a][2] = 0.5 * ((r[a][0] - r[a][1]) + r[a][2]);
                    u[a][3] = r[a][2];
                }
                for (int f = 0; f < 16; ++f) out[(((size_t)cb * 16 + f) * CB + ci) * O + o] = (float)u[f >> 2][f & 3];
            }
    return out;
}

/* ---- Winograd F(4x4, 3x3) form (round 4) -------------------------------------------------------------------------
 * Rule (static, by layer shape only): a layer WITHOUT a residual epilogue (synthesis conv_2, decoder cvt -- never ResBlock conv b)
 * that takes the Winograd form above, has at least 64 input channels and an output of at least 32 px (FFHQ: synthesis conv_2 at
 * 32^2-256^2, decoder cvt at 64^2-256^2) is evaluated per 4x4 OUTPUT tile from its
 * 6x6 input patch: 36 products per 16 outputs instead of 64 (F(2x2,3x3)) or 144 (direct).  Lavin & Gray's matrices
 *     B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
 *     G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
 *     A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
 * Canonical arithmetic (every fp32 op rounded; fmaf where written, every constant a power of two or 5):
 *   weights    U = G g G^T in DOUBLE on the effective fp32 weights, rounded to fp32 once;
 *   input      per 6-vector d (first down the columns of the patch, then along the rows of the result):
 *                a = fmaf(-4, d2, d4), b = fmaf(-4, d1, d3), c = d4 - d2, e = d3 - d1,
 *                t0 = fmaf(4, d0, fmaf(-5, d2, d4)), t1 = a + b, t2 = a - b, t3 = fmaf(2, e, c), t4 = fmaf(-2, e, c),
 *                t5 = fmaf(4, d1, fmaf(-5, d3, d5));
 *   products   M[f] = one fmaf chain over the input channels per frequency f = 6i+j, in the K order of THESE layers: 8-channel
 *              blocks ascending, inside a block channels 0,2,4,6, 1,3,5,7 (two MFMAs whose k slot kq holds channel 8b + 2kq + j);
 *   output     per 6-vector m (first down the columns of M, then along the rows):
 *                p = m1 + m2, q = m1 - m2, r = m3 + m4, s = m3 - m4,
 *                y0 = (m0 + p) + r, y1 = fmaf(2, s, q), y2 = fmaf(4, r, p), y3 = fmaf(8, s, q) + m5. */
#ifndef OC
#define OC 16
#endif
static int g_wino43 = -1;
static int use_wino43(int H, int W, int Cin, int Cout, int bf) {
    if (g_wino43 < 0) { const char* e = getenv("GSAO_WINO43"); g_wino43 = e ? atoi(e) : 0;      /* opt-in (GSAO_WINO43=1): measured slower than F(2x2,3x3) on MI355X, DESIGN.md section 4 round 4 */ }
    return g_wino43 && use_wino(H, W, Cout, 0, bf) && Cin >= 64 && H >= 32;
}

/* conv OIHW (O,I,3,3) -> U packed [f*I + c][O], f = 6*i + j */
static float* pack_wino43(const float* w, int O, int I, float std, int use_std, float lr) {
    static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    float* out = (float*)malloc(sizeof(float) * (size_t)O * I * 36);
    for (int ch = 0; ch < I; ++ch)
            for (int o = 0; o < O; ++o) {
                const float* wk = w + ((size_t)o * I + ch) * 9;
                double g[3][3], r[6][3];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) g[a][b] = (double)eff(wk[a * 3 + b], std, use_std, lr);
                for (int i = 0; i < 6; ++i)
                    for (int b = 0; b < 3; ++b) r[i][b] = (G[i][0] * g[0][b] + G[i][1] * g[1][b]) + G[i][2] * g[2][b];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j)
                        out[((size_t)(i * 6 + j) * I + ch) * O + o] = (float)((r[i][0] * G[j][0] + r[i][1] * G[j][1]) + r[i][2] * G[j][2]);
            }
    return out;
}

static inline void wino43_in(const float d[6], float t[6]) {
    const float a = fmaf(-4.0f, d[2], d[4]), b = fmaf(-4.0f, d[1], d[3]), c = d[4] - d[2], e = d[3] - d[1];
    t[0] = fmaf(4.0f, d[0], fmaf(-5.0f, d[2], d[4]));
    t[1] = a + b;
    t[2] = a - b;
    t[3] = fmaf(2.0f, e, c);
    t[4] = fmaf(-2.0f, e, c);
    t[5] = fmaf(4.0f, d[1], fmaf(-5.0f, d[3], d[5]));
}
static inline void wino43_out(const float m[6], float y[4]) {
    const float p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = (m[0] + p) + r;
    y[1] = fmaf(2.0f, s, q);
    y[2] = fmaf(4.0f, r, p);
    y[3] = fmaf(8.0f, s, q) + m[5];
}

/* 3x3 conv, pad 1, NHWC, Winograd F(4x4,3x3) form.  in: [H][W][Cin] affine-applied; U from pack_wino43. */
static void conv3x3_wino43(const float* in, int H, int W, int Cin, const float* U, int Cout, float* out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int ty = 0; ty < H / 4; ++ty) {
        float* V = (float*)malloc(sizeof(float) * 36 * (size_t)Cin);       /* [f][c] of the current tile */
        for (int tx = 0; tx < W / 4; ++tx) {
            for (int c = 0; c < Cin; ++c) {
                float d[6][6], t[6][6], col[6], res[6];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) {
                        const int yy = 4 * ty - 1 + i, xx = 4 * tx - 1 + j;
                        d[i][j] = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.0f : in[((size_t)yy * W + xx) * Cin + c];
                    }
                for (int j = 0; j < 6; ++j) {            /* down the columns: t = B^T d */
                    for (int i = 0; i < 6; ++i) col[i] = d[i][j];
                    wino43_in(col, res);
                    for (int i = 0; i < 6; ++i) t[i][j] = res[i];
                }
                for (int i = 0; i < 6; ++i) {            /* along the rows: V = t B */
                    wino43_in(t[i], res);
                    for (int j = 0; j < 6; ++j) V[(i * 6 + j) * Cin + c] = res[j];
                }
            }
            for (int o0 = 0; o0 < Cout; o0 += OC) {
                const int on = Cout - o0 < OC ? Cout - o0 : OC;
                static _Thread_local float M[36][OC];
                for (int f = 0; f < 36; ++f) {
                    float acc[OC];
                    for (int o = 0; o < OC; ++o) acc[o] = 0.0f;
                    /* K order of the F(4x4,3x3) layers: 8-channel blocks ascending; inside a block the two MFMAs j = 0, 1, each a
                     * k-ordered chain over the k slots kq = 0..3 holding channel 8b + 2kq + j: channels 0,2,4,6, 1,3,5,7 */
                    for (int cb = 0; cb < Cin / 8; ++cb)
                        for (int kk = 0; kk < 8; ++kk) {
                            const int ch = cb * 8 + ((kk & 3) << 1) + (kk >> 2);
                            const float a = V[f * Cin + ch];
                            const float* wrow = U + ((size_t)f * Cin + ch) * Cout + o0;
                            for (int o = 0; o < on; ++o) acc[o] = fmaf(a, wrow[o], acc[o]);
                        }
                    for (int o = 0; o < OC; ++o) M[f][o] = acc[o];
                }
                for (int o = 0; o < on; ++o) {
                    float sr[4][6], col[6], res[4];
                    for (int j = 0; j < 6; ++j) {        /* down the columns of M */
                        for (int i = 0; i < 6; ++i) col[i] = M[i * 6 + j][o];
                        wino43_out(col, res);
                        for (int i = 0; i < 4; ++i) sr[i][j] = res[i];
                    }
                    for (int i = 0; i < 4; ++i) {        /* along the rows */
                        wino43_out(sr[i], res);
                        for (int j = 0; j < 4; ++j) out[((size_t)(4 * ty + i) * W + 4 * tx + j) * Cout + o0 + o] = res[j];
                    }
                }
            }
        }
        free(V);
    }
}

static float* pack_wino_any(const float* w, int O, int I, float std, int use_std, float lr, int R, int bf) {
    return use_wino43(R, R, I, O, bf) ? pack_wino43(w, O, I, std, use_std, lr) : pack_wino(w, O, I, std, use_std, lr);
}

static float* copy_scaled(const float* w, int64_t n, float std, int use_std, float lr) {
    float* out = (float*)malloc(sizeof(float) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) out[i] = eff(w[i], std, use_std, lr);
    return out;
}

static float* copy_plain(const float* w, int64_t n) {
    float* out = (float*)malloc(sizeof(float) * (size_t)n);
    memcpy(out, w, sizeof(float) * (size_t)n);
    return out;
}

/* ---------------------------------------------------------------- context */

GSAO_API int gsao_create(int device, gsao_ctx** out) {
    (void)device;
    gsao_ctx* c = (gsao_ctx*)calloc(1, sizeof(gsao_ctx));
    if (!c) return fail(NULL, GSA_ERR_NOMEM, "out of memory%s (%ld)", "", 0);
    *out = c;
    return GSA_OK;
}

static void free_generator(gsao_ctx* c) {
    for (int i = 0; i < 8; ++i) { free(c->map_w[i]); free(c->map_b[i]); c->map_w[i] = c->map_b[i] = NULL; }
    free(c->latent_avg); free(c->psi); free(c->constant); free(c->rgb_w); free(c->rgb_b);
    c->latent_avg = c->psi = c->constant = c->rgb_w = c->rgb_b = NULL;
    for (int l = 0; l < MAX_LEVELS; ++l) {
        gen_block* b = &c->blk[l];
        free(b->w1); free(b->blur); free(b->w2); free(b->w2u);
        for (int k = 0; k < 2; ++k) { free(b->nscale[k]); free(b->nbias[k]); free(b->aff_w[k]); free(b->aff_b[k]); free(b->gamma[k]); free(b->beta[k]); }
        memset(b, 0, sizeof *b);
    }
    c->g_ready = 0;
}

static void free_decoder(gsao_ctx* c) {
    for (int l = 0; l < MAX_LEVELS; ++l) {
        dec_level* d = &c->dl[l];
        float* ptrs[] = {d->cvt_w, d->cvt_b, d->cvt_s, d->cvt_rm, d->cvt_beta, d->a_w, d->a_b, d->a_s, d->a_rm, d->a_beta,
                         d->b_w, d->b_b, d->b_s, d->b_rm, d->b_beta, d->sc_w, d->sc_b, d->f_w, d->f_b, d->cvt_u, d->b_u};
        for (size_t i = 0; i < sizeof ptrs / sizeof *ptrs; ++i) free(ptrs[i]);
        memset(d, 0, sizeof *d);
    }
    c->d_ready = 0;
}

GSAO_API void gsao_destroy(gsao_ctx* c) {
    if (!c) return;
    free_generator(c);
    free_decoder(c);
    free_table(&c->gp);
    free_table(&c->dp);
    free(c);
}

GSAO_API const char* gsao_last_error(const gsao_ctx* c) { return c ? c->err : g_err; }
GSAO_API int gsao_set_precision(gsao_ctx* c, int32_t mode) {
    if (!c || (mode != GSA_PREC_F32 && mode != GSA_PREC_BF16)) return GSA_ERR_INVALID;
    c->bf16 = mode;
    return GSA_OK;
}

GSAO_API const char* gsao_version(void) { return "gsa-oracle 0.1 (canonical fp32 CPU restatement)"; }

/* ---------------------------------------------------------------- generator setup */

GSAO_API int gsao_generator_init(gsao_ctx* c, const gsa_generator_config* g) {
    if (!c || !g) return GSA_ERR_INVALID;
    if (g->max_res_log2 < 2 || g->max_res_log2 > MAX_LEVELS) return fail(c, GSA_ERR_INVALID, "max_res_log2 out of range%s (%ld)", "", g->max_res_log2);
    free_generator(c);
    free_table(&c->gp);
    c->gc = *g;
    c->nlev = g->max_res_log2 - 1;
    for (int l = 0; l < c->nlev; ++l) {
        c->ch[l] = nf(g, l + 2);
        if (c->ch[l] % 16) return fail(c, GSA_ERR_INVALID, "channel count must be a multiple of 16%s (%ld)", "", c->ch[l]);
    }
    c->g_init = 1;
    return GSA_OK;
}

static int known_generator_name(gsao_ctx* c, const char* name) {
    int R, k, i;
    char tail[64];
    if (!strcmp(name, "constant_tensor") || !strcmp(name, "latent_avg") || !strcmp(name, "truncation_psi")) return 1;
    if (sscanf(name, "mp_dense_%d_%63s", &i, tail) == 2) return i >= 0 && i < 8 && (!strcmp(tail, "weight") || !strcmp(tail, "bias") || !strcmp(tail, "std"));
    if (sscanf(name, "%d_%63s", &R, tail) != 2) return 0;
    int r = 0;
    while ((1 << r) < R) ++r;
    if ((1 << r) != R || r < 2 || r > c->gc.max_res_log2) return 0;
    if (!strcmp(tail, "conv_1_weight") || !strcmp(tail, "conv_1_std")) return r > 2 && r < 7;
    if (!strcmp(tail, "deconv_1_weight") || !strcmp(tail, "deconv_1_std")) return r >= 7;
    if (!strcmp(tail, "blur_1_w_kernel")) return r > 2;
    if (!strcmp(tail, "conv_2_weight") || !strcmp(tail, "conv_2_std")) return 1;
    if (!strcmp(tail, "conv_to_rgb_weight") || !strcmp(tail, "conv_to_rgb_bias") || !strcmp(tail, "conv_to_rgb_std")) return r == c->gc.max_res_log2;
    char t2[64];
    if (sscanf(tail, "noise_%d_%63s", &k, t2) == 2) return (k == 1 || k == 2) && !strcmp(t2, "scale_factors");
    if (sscanf(tail, "bias_%d_%63s", &k, t2) == 2) return (k == 1 || k == 2) && !strcmp(t2, "bias");
    if (sscanf(tail, "adain_%d_%63s", &k, t2) == 2)
        return (k == 1 || k == 2) && (!strcmp(t2, "dense_affine_weight") || !strcmp(t2, "dense_affine_bias") ||
                                      !strcmp(t2, "dense_affine_std") || !strcmp(t2, "norm_gamma") || !strcmp(t2, "norm_beta"));
    return 0;
}

GSAO_API int gsao_generator_set_param(gsao_ctx* c, const char* name, const float* data, int32_t ndim, const int64_t* dims) {
    if (!c || !c->g_init) return fail(c, GSA_ERR_STATE, "generator_init first%s (%ld)", "", 0);
    if (!known_generator_name(c, name)) return 1; /* ignore_extra=True */
    c->g_ready = 0;
    return put_param(c, &c->gp, name, data, ndim, dims);
}

static int need(gsao_ctx* c, param_table* t, const char* name, int64_t count, float** out) {
    raw_param* p = find_param(t, name);
    if (!p) return fail(c, GSA_ERR_MISSING_PARAM, "parameter %s was not set (%ld)", name, 0);
    if (p->count != count) return fail(c, GSA_ERR_INVALID, "parameter %s has %ld elements, shape mismatch", name, p->count);
    *out = p->data;
    return GSA_OK;
}

#define NEED(tab, nm, cnt, ptr) do { int rc_ = need(c, tab, nm, cnt, ptr); if (rc_) return rc_; } while (0)

static int get_std(gsao_ctx* c, param_table* t, const char* prefix, float* std) {
    char nm[128];
    *std = 1.0f;
    if (!c->gc.use_wscale) return GSA_OK;
    snprintf(nm, sizeof nm, "%s_std", prefix);
    float* p;
    NEED(t, nm, 1, &p);
    *std = p[0];
    return GSA_OK;
}

GSAO_API int gsao_generator_commit(gsao_ctx* c) {
    if (!c || !c->g_init) return fail(c, GSA_ERR_STATE, "generator_init first%s (%ld)", "", 0);
    free_generator(c);
    param_table* t = &c->gp;
    const int L = c->gc.latent_size, us = c->gc.use_wscale;
    char nm[128], pf[96];
    float *w, *b, std;
    int C0 = c->ch[0];
    NEED(t, "constant_tensor", (int64_t)C0 * 16, &w);
    c->constant = (float*)malloc(sizeof(float) * 16 * C0);
    for (int ch = 0; ch < C0; ++ch)
        for (int p = 0; p < 16; ++p) c->constant[p * C0 + ch] = w[ch * 16 + p];
    NEED(t, "latent_avg", 512, &w);
    if (L != 512) return fail(c, GSA_ERR_INVALID, "latent_size must be 512 (latent_avg is (512,) in the reference)%s (%ld)", "", L);
    c->latent_avg = copy_plain(w, 512);
    NEED(t, "truncation_psi", 2 * c->nlev, &w);
    c->psi = copy_plain(w, 2 * c->nlev);
    for (int i = 0; i < 8; ++i) {
        snprintf(pf, sizeof pf, "mp_dense_%d", i);
        int rc = get_std(c, t, pf, &std); if (rc) return rc;
        snprintf(nm, sizeof nm, "%s_weight", pf); NEED(t, nm, (int64_t)L * L, &w);
        snprintf(nm, sizeof nm, "%s_bias", pf); NEED(t, nm, L, &b);
        c->map_w[i] = copy_scaled(w, (int64_t)L * L, std, us, 0.01f); /* lr_mult 0.01, reference :135 */
        c->map_b[i] = copy_scaled(b, L, 1.0f, 0, 0.01f);
    }
    for (int l = 0; l < c->nlev; ++l) {
        gen_block* B = &c->blk[l];
        int r = l + 2, R = 1 << r, C = c->ch[l], Cin = l ? c->ch[l - 1] : C;
        B->C = C; B->Cin = Cin; B->R = R;
        B->has_conv1 = r > 2; B->is_deconv = r >= 7;
        if (B->has_conv1) {
            snprintf(pf, sizeof pf, "%d_%s", R, B->is_deconv ? "deconv_1" : "conv_1");
            int rc = get_std(c, t, pf, &std); if (rc) return rc;
            snprintf(nm, sizeof nm, "%s_weight", pf);
            if (B->is_deconv) { NEED(t, nm, (int64_t)Cin * C * 16, &w); B->w1 = pack_deconv(w, Cin, C, std, us, 1.0f); }
            else { NEED(t, nm, (int64_t)Cin * C * 9, &w); B->w1 = R >= 16 ? pack_upconv(w, C, Cin, std, us, 1.0f) : pack_conv(w, C, Cin, 3, std, us, 1.0f); }
            snprintf(nm, sizeof nm, "%d_blur_1_w_kernel", R); NEED(t, nm, (int64_t)C * 9, &w);
            B->blur = copy_plain(w, (int64_t)C * 9);
        }
        snprintf(pf, sizeof pf, "%d_conv_2", R);
        { int rc = get_std(c, t, pf, &std); if (rc) return rc; }
        snprintf(nm, sizeof nm, "%s_weight", pf); NEED(t, nm, (int64_t)C * C * 9, &w);
        B->w2 = pack_conv(w, C, C, 3, std, us, 1.0f);
        B->w2u = pack_wino_any(w, C, C, std, us, 1.0f, R, c->bf16);
        for (int k = 0; k < 2; ++k) {
            snprintf(nm, sizeof nm, "%d_noise_%d_scale_factors", R, k + 1); NEED(t, nm, C, &w); B->nscale[k] = copy_plain(w, C);
            snprintf(nm, sizeof nm, "%d_bias_%d_bias", R, k + 1); NEED(t, nm, C, &w); B->nbias[k] = copy_plain(w, C);
            snprintf(pf, sizeof pf, "%d_adain_%d_dense_affine", R, k + 1);
            int rc = get_std(c, t, pf, &std); if (rc) return rc;
            snprintf(nm, sizeof nm, "%s_weight", pf); NEED(t, nm, (int64_t)2 * C * L, &w);
            B->aff_w[k] = copy_scaled(w, (int64_t)2 * C * L, std, us, 1.0f);
            snprintf(nm, sizeof nm, "%s_bias", pf); NEED(t, nm, 2 * C, &w); B->aff_b[k] = copy_scaled(w, 2 * C, 1.0f, 0, 1.0f);
            snprintf(nm, sizeof nm, "%d_adain_%d_norm_gamma", R, k + 1); NEED(t, nm, C, &w); B->gamma[k] = copy_plain(w, C);
            snprintf(nm, sizeof nm, "%d_adain_%d_norm_beta", R, k + 1); NEED(t, nm, C, &w); B->beta[k] = copy_plain(w, C);
        }
    }
    {
        int R = 1 << c->gc.max_res_log2, C = c->ch[c->nlev - 1], nc = c->gc.channels;
        snprintf(pf, sizeof pf, "%d_conv_to_rgb", R);
        int rc = get_std(c, t, pf, &std); if (rc) return rc;
        snprintf(nm, sizeof nm, "%s_weight", pf); NEED(t, nm, (int64_t)nc * C, &w);
        c->rgb_w = copy_scaled(w, (int64_t)nc * C, std, us, 1.0f);
        snprintf(nm, sizeof nm, "%s_bias", pf); NEED(t, nm, nc, &w); c->rgb_b = copy_scaled(w, nc, 1.0f, 0, 1.0f);
    }
    c->g_ready = 1;
    return GSA_OK;
}

/* ---------------------------------------------------------------- canonical kernels */

static inline float lrelu(float v) { return v > 0.0f ? v : 0.2f * v; }

/* rint(v * 2^shift) as a wrapping 64-bit integer via the 1.5*2^52 magic constant */
static inline uint64_t to_fixed(float v, double scale) {
    const double magic = 6755399441055744.0;
    double t = fma((double)v, scale, magic);
    uint64_t bits, mbits;
    memcpy(&bits, &t, 8);
    memcpy(&mbits, &magic, 8);
    return bits - mbits;
}

/* fixed-point scales of the statistics: quad sum at 2^-28; quad sum of squares in units of 2^-S2 with S2 a static function of the
 * plane size alone (round 4): S2 = clamp(40 - ceil(log2(H*W)), 20, 26) -- 20 at 1024^2 (the 64-bit sum of a plane then holds
 * rms(x) < 2.9e3 whatever the size), 22 / 24 at 512^2 / 256^2, 26 from 128^2 down.  A small plane has few quads to average the
 * rounding of rint(q * 2^S2) over, and a plane whose values sit on a bias far above their spread needs E[x^2] - mean^2 to survive
 * the cancellation: at 2^-20 a 4x4 plane with values 0.1 +- 0.003 had its variance off by 10 %.  A quad with q >= 2^(50-S2) does
 * not fit the 1.5*2^52 conversion at that unit: it is rounded at 2^-20 and shifted into the unit (still an integer multiple of
 * 2^-S2 and a pure function of q: the sum stays order-independent, the per-value range stays |x| < 2.3e4). */
#define STAT_SCALE1 268435456.0
static int stat_s2(int HW) {
    int lg = 0;
    while ((1 << lg) < HW) ++lg;
    int s2 = 40 - lg;
    if (s2 < 20) s2 = 20;
    if (s2 > 26) s2 = 26;
    if (getenv("GSAO_STAT_S2")) s2 = atoi(getenv("GSAO_STAT_S2"));   /* experiment switch (tests/test_oracle.py) */
    return s2;
}
static double stat_scale2(int HW) { return (double)(1ull << stat_s2(HW)); }
static inline uint64_t to_fixed_sq(float q, int s2) {
    const int big = q >= (float)(1ull << (50 - s2));
    const uint64_t k = to_fixed(q, (double)(1ull << (big ? 20 : s2)));
    return k << (big ? s2 - 20 : 0);
}

typedef struct { float mean, A, B; } affine3;

/* per-(n,c) fixed-point statistics of an NHWC plane set x[H][W][C] */
static void plane_stats(const float* x, int H, int W, int C, uint64_t* I1, uint64_t* I2) {
    const int s2 = stat_s2(H * W);
    for (int c = 0; c < C; ++c) I1[c] = I2[c] = 0;
    for (int y = 0; y < H; ++y)
        for (int x0 = 0; x0 < W; x0 += 4) {
            const float* p = x + ((size_t)y * W + x0) * C;
            for (int c = 0; c < C; ++c) {
                float v0 = p[c], v1 = p[C + c], v2 = p[2 * C + c], v3 = p[3 * C + c];
                float s = (v0 + v1) + (v2 + v3);
                float q = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                I1[c] += to_fixed(s, STAT_SCALE1);
                I2[c] += to_fixed_sq(q, s2);
            }
        }
}

/* InstanceNorm(eps 1e-5, biased variance) folded with the AdaIN style (reference
 * networks_stylegan.py:250-264):  out = IN(x)*(ys+1)+yb = fmaf(x, A, B) with the mean folded into the shift,
 * B = fmaf(-mean, A, beta*(ys+1)+yb) (round 3: one operation per element for every consumer instead of two) */
static void finalize(const uint64_t* I1, const uint64_t* I2, int HW, int C, const float* style /*2C*/,
                     const float* gamma, const float* beta, affine3* out) {
    const double inv_hw = 1.0 / (double)HW; /* HW is a power of two: exact */
    const double STAT_SCALE2 = stat_scale2(HW);
    for (int c = 0; c < C; ++c) {
        double m = (double)(int64_t)I1[c] * (1.0 / STAT_SCALE1) * inv_hw;
        double e2 = (double)(int64_t)I2[c] * (1.0 / STAT_SCALE2) * inv_hw;
        double var = fma(-m, m, e2);
        if (!(var > 0.0)) var = 0.0;
        float mean_f = (float)m, var_f = (float)var;
        float inv = 1.0f / sqrtf(var_f + 1e-5f);
        float g = gamma[c] * inv;
        float s1 = style[c] + 1.0f;
        out[c].mean = mean_f;
        out[c].A = g * s1;
        out[c].B = fmaf(-mean_f, out[c].A, fmaf(beta[c], s1, style[C + c]));
    }
}

static void apply_affine(const float* x, size_t npix, int C, const affine3* a, float* out) {
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npix; ++p)
        for (int c = 0; c < C; ++c) out[p * C + c] = fmaf(x[p * C + c], a[c].A, a[c].B);
}

/* y[j] = (chain_k fmaf(x[k], W[j][k], 0)) + b[j] */
static void dense(const float* x, const float* W, const float* b, int J, int K, float* y) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < J; ++j) {
        float acc = 0.0f;
        const float* w = W + (size_t)j * K;
        for (int k = 0; k < K; ++k) acc = fmaf(x[k], w[k], acc);
        y[j] = acc + b[j];
    }
}

#define PX 4
#define OC 16

/* bf16 mode (include/gsa.h gsa_set_precision): conv inputs (after AdaIN) and weights are rounded to bf16,
 * round-to-nearest-even -- what v_cvt_pk_bf16_f32 and the host packer of the HIP library do.  Products of
 * two bf16 are exact in fp32; the accumulation below stays the canonical fp32 chain, while the matrix core
 * adds its 16 products per instruction in an order/alignment of its own (tools/probe/), so parity with the
 * HIP path in this mode is a tolerance, not bit equality. */
static inline float bf16r(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}
#define OPND(v) (bf ? bf16r(v) : (v))

/* bf16 mode, round 2: every activation tensor that lives in HBM is bf16 -- the producer rounds what it stores (statistics
 * are taken from the fp32 values first), consumers widen exactly */
static void store_bf16(float* x, size_t n, int bf) {
    if (!bf) return;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) x[i] = bf16r(x[i]);
}

/* 3x3 conv, pad 1, NHWC.  in: [Hs][Ws][Cin] (already affine-applied); when up!=0 the
 * logical input is the nearest-x2 upsample of `in` (UpSampling, reference :308-315).
 * out: raw accumulators [H][W][Cout], H = Hs<<up. */
/* K split (static rule, by layer shape only): a direct 3x3 convolution over >= 64 input channels (a multiple of 64) whose
 * output is <= 8 px, or <= 32 px with at most 32 output channels (too few tiles to fill the chip otherwise), is
 * accumulated as FOUR independent chains, chain q over the q-th contiguous quarter of the 16-channel blocks, combined as
 * (s0 + s1) + (s2 + s3).  On the GPU the four waves of a workgroup each take one quarter: the 512-channel layers at
 * 4^2-32^2 were ONE dependent chain of 1152 MFMAs behind 32 load -> LDS -> barrier rounds before. */
static int use_ksplit(int Cin, int H, int Cout) { return Cin >= 64 && Cin % 64 == 0 && (H <= 8 || (H <= 32 && Cout <= 32)); }

static void conv3x3(const float* in, int Hs, int Ws, int Cin, int up, const float* Wp, int Cout, float* out, int bf, int perm) {
    const int H = Hs << up, W = Ws << up;
    const int nblk = Cin / CB;
    const int nq = (perm && use_ksplit(Cin, H, Cout)) ? 4 : 1, bq = nblk / nq;      /* perm = 0: the final conv (vector ALU): one chain */
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < H; ++y)
        for (int x0 = 0; x0 < W; x0 += PX)
            for (int o0 = 0; o0 < Cout; o0 += OC) {
                const int on = Cout - o0 < OC ? Cout - o0 : OC;
                float part[4][PX][OC];
                for (int q = 0; q < nq; ++q) {
                    float (*acc)[OC] = part[q];
                    for (int p = 0; p < PX; ++p)
                        for (int o = 0; o < OC; ++o) acc[p][o] = 0.0f;
                    for (int cb = q * bq; cb < (q + 1) * bq; ++cb)
                        for (int ky = 0; ky < 3; ++ky) {
                            const int yy = y + ky - 1;
                            if (yy < 0 || yy >= H) continue;
                            for (int kx = 0; kx < 3; ++kx)
                                for (int kk = 0; kk < CB; ++kk) {
                                    const int ci = perm ? CPERM(kk) : kk;     /* perm = 0: ascending channels */
                                    const float* wsrc = Wp + (((size_t)cb * 9 + ky * 3 + kx) * CB + ci) * Cout + o0;
                                    float wrow[OC];
                                    for (int o = 0; o < on; ++o) wrow[o] = OPND(wsrc[o]);
                                    for (int o = on; o < OC; ++o) wrow[o] = 0.0f;
                                    for (int p = 0; p < PX; ++p) {
                                        const int xx = x0 + p + kx - 1;
                                        if (xx < 0 || xx >= W) continue;
                                        const float a = OPND(in[((size_t)(yy >> up) * Ws + (xx >> up)) * Cin + cb * CB + ci]);
                                        if (on == OC) {
                                            for (int o = 0; o < OC; ++o) acc[p][o] = fmaf(a, wrow[o], acc[p][o]);
                                        } else {
                                            for (int o = 0; o < on; ++o) acc[p][o] = fmaf(a, wrow[o], acc[p][o]);
                                        }
                                    }
                                }
                        }
                }
                for (int p = 0; p < PX; ++p)
                    for (int o = 0; o < on; ++o)
                        out[((size_t)y * W + x0 + p) * Cout + o0 + o] =
                            nq == 4 ? (part[0][p][o] + part[1][p][o]) + (part[2][p][o] + part[3][p][o]) : part[0][p][o];
            }
}

/* Deconvolution k4 s2 p1 (reference networks_stylegan.py:460-476, A.3):
 * out[oy][ox][o] = sum in[iy][ix][i] * W[i][o][ky][kx] over oy = 2*iy - 1 + ky.
 * Canonical order: 16-channel block, then valid ky ascending, valid kx ascending, channel. */
static void deconv4x4s2(const float* in, int Hs, int Ws, int Cin, const float* Wd, int Cout, float* out, int bf) {
    const int H = Hs * 2, W = Ws * 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int oy = 0; oy < H; ++oy)
        for (int ox = 0; ox < W; ++ox)
            for (int o0 = 0; o0 < Cout; o0 += OC) {
                const int on = Cout - o0 < OC ? Cout - o0 : OC;
                float acc[OC];
                for (int o = 0; o < OC; ++o) acc[o] = 0.0f;
                for (int cb = 0; cb < Cin / CB; ++cb)
                    for (int ky = (oy + 1) & 1; ky < 4; ky += 2) {
                        const int iy = (oy + 1 - ky) / 2;
                        if (oy + 1 - ky < 0 || iy >= Hs) continue;
                        for (int kx = (ox + 1) & 1; kx < 4; kx += 2) {
                            const int ix = (ox + 1 - kx) / 2;
                            if (ox + 1 - kx < 0 || ix >= Ws) continue;
                            for (int kk = 0; kk < CB; ++kk) {
                                const int ci = CPERM(kk);
                                const float a = OPND(in[((size_t)iy * Ws + ix) * Cin + cb * CB + ci]);
                                const float* wrow = Wd + (((size_t)cb * 16 + ky * 4 + kx) * CB + ci) * Cout + o0;
                                for (int o = 0; o < on; ++o) acc[o] = fmaf(a, OPND(wrow[o]), acc[o]);
                            }
                        }
                    }
                for (int o = 0; o < on; ++o) out[((size_t)oy * W + ox) * Cout + o0 + o] = acc[o];
            }
}

/* ---- Winograd F(2x2, 2x2) form of the stride-2 layers (round 3) --------------------------------------------------
 * Rule (static, by mode only): in fp32 mode EVERY layer evaluated in the stride-2 form -- Deconvolution k4 s2 p1
 * (reference networks_stylegan.py:460-476) and nearest-x2 + conv3x3 with outputs >= 16 px (the pre-summed 4x4 kernel of
 * pack_upconv; reference :22-27, networks_seg.py:86-88) -- is evaluated per output parity class (py, px) as a 2x2-tap
 * stride-1 convolution in Winograd F(2x2, 2x2) form: 9 products per 2x2 class outputs instead of 16.
 *   class filter   g[a][b] = Wd[ky(a)][kx(b)],  ky(a) = 3 - py - 2a,  kx(b) = 3 - px - 2b   (a, b = 0: the input row /
 *                  column i - 1 + py, 1: the next one);  y[i] = g0 d[i-1+py] + g1 d[i+py]
 *   weights        U = G g G^T in fp32, every add rounded:  u01 = g00+g01, u21 = g10+g11, u10 = g00+g10, u12 = g01+g11,
 *                  u11 = u01 + u21, the corners are the taps themselves
 *   input          per tile (2x2 class outputs = input rows / columns 2t-1+py .. 2t+1+py, zero outside the image):
 *                  rows t0 = d0 - d1, t1 = d1, t2 = d2 - d1, then the same three forms along the columns
 *   products       M[f] = fmaf chain over the input channels (16-channel blocks ascending, CPERM order inside), one chain
 *                  per frequency f = 3r + c -- bitwise what v_mfma_f32_16x16x4_f32 produces
 *   output         rows s0 = m0 + m1, s1 = m1 + m2, then the same two forms along the columns.
 * Exact algebra; the result differs from the 4-tap chain by a few fp32 ulps (oracle/ref_semantic.py keeps the
 * reference's operators and bounds the difference).  GSAO_WINO22=0 selects the direct form (A/B timing only). */
static int g_wino22_enabled = -1;
static int use_wino22(int bf) {
    if (g_wino22_enabled < 0) { const char* e = getenv("GSAO_WINO22"); g_wino22_enabled = !(e && atoi(e) == 0); }
    return g_wino22_enabled && !bf;
}

static void deconv4x4s2_wino(const float* in, int Hs, int Ws, int Cin, const float* Wd, int Cout, float* out) {
    const int W = Ws * 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int ty = 0; ty < Hs / 2; ++ty) {
        float* V = (float*)malloc(sizeof(float) * 9 * (size_t)Cin);       /* [f][c] of the current (tile, class) */
        for (int tx = 0; tx < Ws / 2; ++tx)
            for (int cls = 0; cls < 4; ++cls) {
                const int py = cls >> 1, px = cls & 1;
                for (int c = 0; c < Cin; ++c) {
                    float d[3][3], t[3][3];
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) {
                            const int yy = 2 * ty - 1 + py + i, xx = 2 * tx - 1 + px + j;
                            d[i][j] = (yy < 0 || yy >= Hs || xx < 0 || xx >= Ws) ? 0.0f : in[((size_t)yy * Ws + xx) * Cin + c];
                        }
                    for (int j = 0; j < 3; ++j) {
                        t[0][j] = d[0][j] - d[1][j];
                        t[1][j] = d[1][j];
                        t[2][j] = d[2][j] - d[1][j];
                    }
                    for (int i = 0; i < 3; ++i) {
                        V[(i * 3 + 0) * Cin + c] = t[i][0] - t[i][1];
                        V[(i * 3 + 1) * Cin + c] = t[i][1];
                        V[(i * 3 + 2) * Cin + c] = t[i][2] - t[i][1];
                    }
                }
                const int k0y = 3 - py, k1y = 1 - py, k0x = 3 - px, k1x = 1 - px;     /* taps of g[a][b] */
                for (int o0 = 0; o0 < Cout; o0 += OC) {
                    const int on = Cout - o0 < OC ? Cout - o0 : OC;
                    float M[9][OC];
                    for (int f = 0; f < 9; ++f)
                        for (int o = 0; o < OC; ++o) M[f][o] = 0.0f;
                    for (int cb = 0; cb < Cin / CB; ++cb)
                        for (int kk = 0; kk < CB; ++kk) {
                            const int ci = CPERM(kk);
                            const float* g00 = Wd + (((size_t)cb * 16 + k0y * 4 + k0x) * CB + ci) * Cout + o0;
                            const float* g01 = Wd + (((size_t)cb * 16 + k0y * 4 + k1x) * CB + ci) * Cout + o0;
                            const float* g10 = Wd + (((size_t)cb * 16 + k1y * 4 + k0x) * CB + ci) * Cout + o0;
                            const float* g11 = Wd + (((size_t)cb * 16 + k1y * 4 + k1x) * CB + ci) * Cout + o0;
                            for (int o = 0; o < on; ++o) {
                                float u[9];
                                u[0] = g00[o]; u[2] = g01[o]; u[6] = g10[o]; u[8] = g11[o];
                                u[1] = g00[o] + g01[o];
                                u[7] = g10[o] + g11[o];
                                u[3] = g00[o] + g10[o];
                                u[5] = g01[o] + g11[o];
                                u[4] = u[1] + u[7];
                                for (int f = 0; f < 9; ++f) M[f][o] = fmaf(V[f * Cin + cb * CB + ci], u[f], M[f][o]);
                            }
                        }
                    for (int o = 0; o < on; ++o) {
                        float sr[2][3];
                        for (int j = 0; j < 3; ++j) {
                            sr[0][j] = M[0 + j][o] + M[3 + j][o];
                            sr[1][j] = M[3 + j][o] + M[6 + j][o];
                        }
                        for (int i = 0; i < 2; ++i) {
                            const size_t row = (size_t)(2 * (2 * ty + i) + py) * W;
                            out[(row + 2 * (2 * tx) + px) * Cout + o0 + o] = sr[i][0] + sr[i][1];
                            out[(row + 2 * (2 * tx + 1) + px) * Cout + o0 + o] = sr[i][1] + sr[i][2];
                        }
                    }
                }
            }
        free(V);
    }
}

/* 3x3 conv, pad 1, NHWC, Winograd F(2x2,3x3) form (see pack_wino).  in: [H][W][Cin] affine-applied; U packed. */
static void conv3x3_wino(const float* in, int H, int W, int Cin, const float* U, int Cout, float* out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int ty = 0; ty < H / 2; ++ty) {
        float* V = (float*)malloc(sizeof(float) * 16 * (size_t)Cin);       /* [f][c] of the current tile */
        for (int tx = 0; tx < W / 2; ++tx) {
            for (int c = 0; c < Cin; ++c) {
                float d[4][4], t[4][4];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        const int yy = 2 * ty - 1 + i, xx = 2 * tx - 1 + j;
                        d[i][j] = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.0f : in[((size_t)yy * W + xx) * Cin + c];
                    }
                for (int j = 0; j < 4; ++j) {
                    t[0][j] = d[0][j] - d[2][j];
                    t[1][j] = d[1][j] + d[2][j];
                    t[2][j] = d[2][j] - d[1][j];
                    t[3][j] = d[1][j] - d[3][j];
                }
                for (int i = 0; i < 4; ++i) {
                    V[(i * 4 + 0) * Cin + c] = t[i][0] - t[i][2];
                    V[(i * 4 + 1) * Cin + c] = t[i][1] + t[i][2];
                    V[(i * 4 + 2) * Cin + c] = t[i][2] - t[i][1];
                    V[(i * 4 + 3) * Cin + c] = t[i][1] - t[i][3];
                }
            }
            for (int o0 = 0; o0 < Cout; o0 += OC) {
                const int on = Cout - o0 < OC ? Cout - o0 : OC;
                float M[16][OC];
                for (int f = 0; f < 16; ++f) {
                    float acc[OC];
                    for (int o = 0; o < OC; ++o) acc[o] = 0.0f;
                    for (int cb = 0; cb < Cin / CB; ++cb)
                        for (int kk = 0; kk < CB; ++kk) {
                            const int ci = CPERM(kk);
                            const float a = V[f * Cin + cb * CB + ci];
                            const float* wrow = U + (((size_t)cb * 16 + f) * CB + ci) * Cout + o0;
                            for (int o = 0; o < on; ++o) acc[o] = fmaf(a, wrow[o], acc[o]);
                        }
                    for (int o = 0; o < OC; ++o) M[f][o] = acc[o];
                }
                for (int o = 0; o < on; ++o) {
                    float sr[2][4];
                    for (int j = 0; j < 4; ++j) {
                        sr[0][j] = (M[0 + j][o] + M[4 + j][o]) + M[8 + j][o];
                        sr[1][j] = (M[4 + j][o] - M[8 + j][o]) - M[12 + j][o];
                    }
                    for (int i = 0; i < 2; ++i) {
                        const float y0 = (sr[i][0] + sr[i][1]) + sr[i][2];
                        const float y1 = (sr[i][1] - sr[i][2]) - sr[i][3];
                        out[((size_t)(2 * ty + i) * W + 2 * tx) * Cout + o0 + o] = y0;
                        out[((size_t)(2 * ty + i) * W + 2 * tx + 1) * Cout + o0 + o] = y1;
                    }
                }
            }
        }
        free(V);
    }
}

/* depthwise 3x3 blur, zero pad (reference networks_stylegan.py:229-236) */
static void blur3x3(const float* t, int H, int W, int C, const float* wk /*[C][9]*/, float* out) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < C; ++c) {
                float b = 0.0f;
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    if (yy < 0 || yy >= H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = x + kx - 1;
                        if (xx < 0 || xx >= W) continue;
                        b = fmaf(t[((size_t)yy * W + xx) * C + c], wk[c * 9 + ky * 3 + kx], b);
                    }
                }
                out[((size_t)y * W + x) * C + c] = b;
            }
}

/* AddNoise -> Bias -> LeakyReLU(0.2), in place on NHWC (reference :302-304, :544, :40) */
static void noise_bias_act(float* x, int H, int W, int C, const float* noise /*[H][W]*/, const float* sf, const float* bias) {
#pragma omp parallel for schedule(static)
    for (int p = 0; p < H * W; ++p)
        for (int c = 0; c < C; ++c) {
            float nz = sf[c] * noise[p];
            float v = (x[(size_t)p * C + c] + nz) + bias[c];
            x[(size_t)p * C + c] = lrelu(v);
        }
}

static void nhwc_to_nchw(const float* in, int H, int W, int C, float* out) {
    for (int c = 0; c < C; ++c)
        for (int p = 0; p < H * W; ++p) out[(size_t)c * H * W + p] = in[(size_t)p * C + c];
}

static void nchw_to_nhwc(const float* in, int H, int W, int C, float* out) {
    for (int c = 0; c < C; ++c)
        for (int p = 0; p < H * W; ++p) out[(size_t)p * C + c] = in[(size_t)c * H * W + p];
}

/* ---------------------------------------------------------------- generator forward */

GSAO_API int gsao_reserve(gsao_ctx* c, int32_t max_batch) { (void)c; (void)max_batch; return GSA_OK; }

GSAO_API int gsao_generator_forward(gsao_ctx* c, void* stream, int32_t n, const float* z, const float* const* noise,
                                    int32_t num_noise, float* rgb, uint8_t* img, float* const* feats, int32_t num_feats) {
    (void)stream;
    if (!c || !c->g_ready) return fail(c, GSA_ERR_STATE, "generator_commit first%s (%ld)", "", 0);
    if (n < 0 || !z || !noise) return fail(c, GSA_ERR_INVALID, "bad arguments to generator_forward%s (%ld)", "", n);
    if (num_noise != 2 * c->nlev || (feats && num_feats != c->nlev))
        return fail(c, GSA_ERR_INVALID, "noise / feature pointer counts do not match this generator%s (%ld)", "", num_noise);
    const int L = c->gc.latent_size, nlev = c->nlev, nc = c->gc.channels;
    const int Cmax = c->ch[0] > 0 ? 512 : 0;
    (void)Cmax;
    int maxC = 0;
    for (int l = 0; l < nlev; ++l) if (c->ch[l] > maxC) maxC = c->ch[l];
    size_t maxact = 0;
    for (int l = 0; l < nlev; ++l) {
        size_t a = (size_t)c->ch[l] << (2 * (l + 2));
        if (a > maxact) maxact = a;
    }
    float* xa = (float*)malloc(sizeof(float) * maxact);   /* current raw activation */
    float* xb = (float*)malloc(sizeof(float) * maxact);   /* affine-applied / temp */
    float* xc = (float*)malloc(sizeof(float) * maxact);   /* conv output */
    float* w = (float*)malloc(sizeof(float) * L);
    float* w2 = (float*)malloc(sizeof(float) * L);
    float* dl = (float*)malloc(sizeof(float) * L);
    float* style = (float*)malloc(sizeof(float) * 2 * maxC);
    uint64_t* I1 = (uint64_t*)malloc(sizeof(uint64_t) * maxC);
    uint64_t* I2 = (uint64_t*)malloc(sizeof(uint64_t) * maxC);
    affine3* aff = (affine3*)malloc(sizeof(affine3) * maxC);
    if (!xa || !xb || !xc || !w || !w2 || !dl || !style || !I1 || !I2 || !aff) return fail(c, GSA_ERR_NOMEM, "out of memory in generator_forward%s (%ld)", "", 0);

    for (int s = 0; s < n; ++s) {
        /* mapping: PixelNorm then 8 x (dense, lrelu) -- reference :128-139, :558-565 */
        const float* zs = z + (size_t)s * L;
        float ss = 0.0f;
        for (int k = 0; k < L; ++k) ss = fmaf(zs[k], zs[k], ss);
        float rn = 1.0f / sqrtf(ss / (float)L + 1e-8f);
        for (int k = 0; k < L; ++k) w[k] = zs[k] * rn;
        for (int i = 0; i < 8; ++i) {
            dense(w, c->map_w[i], c->map_b[i], L, L, w2);
            for (int k = 0; k < L; ++k) w[k] = lrelu(w2[k]);
        }
        for (int l = 0; l < nlev; ++l) {
            const gen_block* B = &c->blk[l];
            const int C = B->C, R = B->R, Cin = B->Cin;
            const size_t npix = (size_t)R * R;
            for (int k = 0; k < 2; ++k) {
                const int li = 2 * l + k;
                const float* nz = noise[li] + (size_t)s * npix;
                /* truncation lerp, reference :158-163 */
                const float psi = c->psi[li], om = 1.0f - psi;
                for (int q = 0; q < L; ++q) dl[q] = c->latent_avg[q] * om + w[q] * psi;
                dense(dl, B->aff_w[k], B->aff_b[k], 2 * C, L, style);
                if (k == 0) {
                    if (!B->has_conv1) {
                        memcpy(xa, c->constant, sizeof(float) * npix * C); /* broadcast const, reference :178 */
                    } else {
                        /* xb holds the affine-applied previous feature */
                        if ((B->is_deconv || R >= 16) && use_wino22(c->bf16)) deconv4x4s2_wino(xb, R / 2, R / 2, Cin, B->w1, C, xc);
                        else if (B->is_deconv) deconv4x4s2(xb, R / 2, R / 2, Cin, B->w1, C, xc, c->bf16);
                        else if (R >= 16) deconv4x4s2(xb, R / 2, R / 2, Cin, B->w1, C, xc, c->bf16);   /* sub-pixel up+conv */
                        else conv3x3(xb, R / 2, R / 2, Cin, 1, B->w1, C, xc, c->bf16, 1);
                        store_bf16(xc, npix * C, c->bf16);      /* the raw conv_1 output is a stored tensor */
                        blur3x3(xc, R, R, C, B->blur, xa);
                    }
                } else {
                    if (use_wino43(R, R, C, C, c->bf16)) conv3x3_wino43(xb, R, R, C, B->w2u, C, xa);
                    else if (use_wino(R, R, C, 0, c->bf16)) conv3x3_wino(xb, R, R, C, B->w2u, C, xa);
                    else conv3x3(xb, R, R, C, 0, B->w2, C, xa, c->bf16, 1);
                }
                noise_bias_act(xa, R, R, C, nz, B->nscale[k], B->nbias[k]);
                plane_stats(xa, R, R, C, I1, I2);
                store_bf16(xa, npix * C, c->bf16);              /* x1 / x2: stored after the statistics were taken */
                finalize(I1, I2, R * R, C, style, B->gamma[k], B->beta[k], aff);
                apply_affine(xa, npix, C, aff, xb);
            }
            if (feats && feats[l]) nhwc_to_nchw(xb, R, R, C, feats[l] + (size_t)s * npix * C);
        }
        /* toRGB 1x1 conv + bias (reference :118-126), then _transform_gan_back */
        {
            const int l = nlev - 1, C = c->ch[l], R = 1 << (l + 2);
            const size_t npix = (size_t)R * R;
            for (size_t p = 0; p < npix; ++p)
                for (int o = 0; o < nc; ++o) {
                    float acc = 0.0f;
                    for (int ch = 0; ch < C; ++ch) acc = fmaf(xb[p * C + ch], c->rgb_w[o * C + ch], acc);
                    float v = acc + c->rgb_b[o];
                    if (rgb) rgb[((size_t)s * nc + o) * npix + p] = v;
                    if (img) {
                        float t = (v + 1.0f) * 0.5f;
                        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
                        t = 255.0f * t;
                        img[((size_t)s * npix + p) * nc + o] = (uint8_t)t;
                    }
                }
        }
    }
    free(xa); free(xb); free(xc); free(w); free(w2); free(dl); free(style); free(I1); free(I2); free(aff);
    return GSA_OK;
}

/* ---------------------------------------------------------------- decoder */

GSAO_API int gsao_decoder_init(gsao_ctx* c, const gsa_decoder_config* d) {
    if (!c || !d) return GSA_ERR_INVALID;
    if (d->num_feats < 1 || d->num_feats > MAX_LEVELS) return fail(c, GSA_ERR_INVALID, "num_feats out of range%s (%ld)", "", d->num_feats);
    free_decoder(c);
    free_table(&c->dp);
    c->d_n = d->num_feats; c->d_s0 = d->start_res; c->d_bn = d->use_bn;
    for (int i = 0; i <= d->num_feats; ++i) c->d_feat[i] = d->features[i];
    for (int i = 0; i < d->num_feats; ++i) {
        c->d_inch[i] = d->in_channels[i];
        if (c->d_inch[i] % 16 || c->d_feat[i] % 16) return fail(c, GSA_ERR_INVALID, "decoder channels must be multiples of 16%s (%ld)", "", i);
    }
    /* start_res: the first feature the decoder consumes (reference networks_seg.py:56,64,81,102); levels below it
     * have no blocks and their features are ignored */
    if (d->start_res < 0 || d->start_res >= d->num_feats) return fail(c, GSA_ERR_INVALID, "start_res out of range%s (%ld)", "", d->start_res);
    c->d_init = 1;
    return GSA_OK;
}

GSAO_API int gsao_decoder_set_param(gsao_ctx* c, const char* name, const float* data, int32_t ndim, const int64_t* dims) {
    if (!c || !c->d_init) return fail(c, GSA_ERR_STATE, "decoder_init first%s (%ld)", "", 0);
    c->d_ready = 0;
    return put_param(c, &c->dp, name, data, ndim, dims);
}

/* BatchNorm at inference (A.13): (y - rm)/sqrt(rv+eps)*gamma + beta = fmaf(y - rm, s, beta) */
static int load_bn(gsao_ctx* c, const char* prefix, int C, float** s, float** rm, float** beta) {
    char nm[128];
    float *g, *b, *m, *v;
    if (!c->d_bn) {
        *s = (float*)malloc(sizeof(float) * C); *rm = (float*)calloc(C, sizeof(float)); *beta = (float*)calloc(C, sizeof(float));
        for (int i = 0; i < C; ++i) (*s)[i] = 1.0f;
        return GSA_OK;
    }
    snprintf(nm, sizeof nm, "%s.gamma", prefix); NEED(&c->dp, nm, C, &g);
    snprintf(nm, sizeof nm, "%s.beta", prefix); NEED(&c->dp, nm, C, &b);
    snprintf(nm, sizeof nm, "%s.running_mean", prefix); NEED(&c->dp, nm, C, &m);
    snprintf(nm, sizeof nm, "%s.running_var", prefix); NEED(&c->dp, nm, C, &v);
    *s = (float*)malloc(sizeof(float) * C);
    for (int i = 0; i < C; ++i) (*s)[i] = g[i] / sqrtf(v[i] + 1e-5f);
    *rm = copy_plain(m, C);
    *beta = copy_plain(b, C);
    return GSA_OK;
}

GSAO_API int gsao_decoder_commit(gsao_ctx* c) {
    if (!c || !c->d_init) return fail(c, GSA_ERR_STATE, "decoder_init first%s (%ld)", "", 0);
    free_decoder(c);
    param_table* t = &c->dp;
    char nm[128], pf[96];
    float *w, *b;
    const int n = c->d_n;
    for (int i = c->d_s0; i < n; ++i) {
        dec_level* d = &c->dl[i];
        d->F = c->d_feat[i]; d->I = c->d_inch[i];
        snprintf(nm, sizeof nm, "cvt_block_%d.0.weight", i); NEED(t, nm, (int64_t)d->F * d->I * 9, &w);
        d->cvt_w = pack_conv(w, d->F, d->I, 3, 1.0f, 0, 1.0f);
        d->cvt_u = pack_wino_any(w, d->F, d->I, 1.0f, 0, 1.0f, 4 << i, c->bf16);
        snprintf(nm, sizeof nm, "cvt_block_%d.0.bias", i); NEED(t, nm, d->F, &b); d->cvt_b = copy_plain(b, d->F);
        snprintf(pf, sizeof pf, "cvt_block_%d.1", i);
        { int rc = load_bn(c, pf, d->F, &d->cvt_s, &d->cvt_rm, &d->cvt_beta); if (rc) return rc; }
        d->cs = c->d_feat[i + 1];
        d->in_c = d->F * (i > c->d_s0 ? 2 : 1);
        d->is_last = i == n - 1;
        if (!d->is_last) {
            const int second = c->d_bn ? 3 : 2;
            snprintf(pf, sizeof pf, "main_block_%d.1.base_layers", i);
            snprintf(nm, sizeof nm, "%s.0.weight", pf); NEED(t, nm, (int64_t)d->cs * d->in_c * 9, &w);
            d->a_w = (8 << i) >= 16 ? pack_upconv(w, d->cs, d->in_c, 1.0f, 0, 1.0f) : pack_conv(w, d->cs, d->in_c, 3, 1.0f, 0, 1.0f);
            snprintf(nm, sizeof nm, "%s.0.bias", pf); NEED(t, nm, d->cs, &b); d->a_b = copy_plain(b, d->cs);
            snprintf(nm, sizeof nm, "%s.1", pf);
            { int rc = load_bn(c, nm, d->cs, &d->a_s, &d->a_rm, &d->a_beta); if (rc) return rc; }
            snprintf(nm, sizeof nm, "%s.%d.weight", pf, second); NEED(t, nm, (int64_t)d->cs * d->cs * 9, &w);
            d->b_w = pack_conv(w, d->cs, d->cs, 3, 1.0f, 0, 1.0f);
            d->b_u = pack_wino(w, d->cs, d->cs, 1.0f, 0, 1.0f);      /* conv b (residual epilogue) keeps F(2x2,3x3) */
            snprintf(nm, sizeof nm, "%s.%d.bias", pf, second); NEED(t, nm, d->cs, &b); d->b_b = copy_plain(b, d->cs);
            snprintf(nm, sizeof nm, "%s.%d", pf, second + 1);
            { int rc = load_bn(c, nm, d->cs, &d->b_s, &d->b_rm, &d->b_beta); if (rc) return rc; }
            d->has_sc = d->cs != d->in_c;
            if (d->has_sc) {
                snprintf(nm, sizeof nm, "main_block_%d.1.shortcut.0.weight", i); NEED(t, nm, (int64_t)d->cs * d->in_c, &w);
                d->sc_w = pack_conv(w, d->cs, d->in_c, 1, 1.0f, 0, 1.0f);
                snprintf(nm, sizeof nm, "main_block_%d.1.shortcut.0.bias", i); NEED(t, nm, d->cs, &b); d->sc_b = copy_plain(b, d->cs);
            }
        } else {
            snprintf(nm, sizeof nm, "main_block_%d.0.weight", i); NEED(t, nm, (int64_t)d->cs * d->in_c * 9, &w);
            d->f_w = pack_conv(w, d->cs, d->in_c, 3, 1.0f, 0, 1.0f);
            snprintf(nm, sizeof nm, "main_block_%d.0.bias", i); NEED(t, nm, d->cs, &b); d->f_b = copy_plain(b, d->cs);
        }
    }
    c->d_ready = 1;
    return GSA_OK;
}

/* conv bias -> BatchNorm(inference) -> LeakyReLU, in place */
static void bias_bn_act(float* x, size_t npix, int C, const float* bias, const float* s, const float* rm, const float* beta) {
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npix; ++p)
        for (int c = 0; c < C; ++c) {
            /* conv bias + inference BatchNorm folded into ONE fma per element (round 3): k = fmaf(bias - running_mean, s, beta)
             * per channel, y = lrelu(fmaf(x, s, k)); reference networks_seg.py:14-32, 68-76 (use_bn=False: s = 1, k = bias) */
            const float k = fmaf(bias[c] - rm[c], s[c], beta[c]);
            x[p * C + c] = lrelu(fmaf(x[p * C + c], s[c], k));
        }
}

GSAO_API int gsao_decoder_forward(gsao_ctx* c, void* stream, int32_t n, const float* const* feats, int32_t num_feats, float* logits,
                                  uint8_t* mask) {
    (void)stream;
    if (!c || !c->d_ready) return fail(c, GSA_ERR_STATE, "decoder_commit first%s (%ld)", "", 0);
    if (n < 0 || !feats) return fail(c, GSA_ERR_INVALID, "bad arguments to decoder_forward%s (%ld)", "", n);
    if (num_feats != c->d_n) return fail(c, GSA_ERR_INVALID, "feature pointer count does not match this decoder%s (%ld)", "", num_feats);
    const int nl = c->d_n;
    /* feature i is at 4*2^i pixels (the generator's 4..2^max ladder, reference networks_stylegan.py:184-192) */
    size_t maxbuf = 0;
    for (int i = c->d_s0; i < nl; ++i) {
        const size_t R = (size_t)4 << i;
        size_t a = R * R * (size_t)c->dl[i].I;
        size_t b2 = 4 * R * R * (size_t)(c->dl[i].cs > c->dl[i].in_c ? c->dl[i].cs : c->dl[i].in_c);
        if (a > maxbuf) maxbuf = a;
        if (b2 > maxbuf) maxbuf = b2;
    }
    float* fin = (float*)malloc(sizeof(float) * maxbuf);   /* NHWC feature */
    float* cat = (float*)malloc(sizeof(float) * maxbuf);   /* concat(prev, cvt) at R_i */
    float* ya = (float*)malloc(sizeof(float) * maxbuf);
    float* yb = (float*)malloc(sizeof(float) * maxbuf);
    float* prev = (float*)malloc(sizeof(float) * maxbuf);
    if (!fin || !cat || !ya || !yb || !prev) return fail(c, GSA_ERR_NOMEM, "out of memory in decoder_forward%s (%ld)", "", 0);
    for (int s = 0; s < n; ++s) {
        for (int i = c->d_s0; i < nl; ++i) {
            const dec_level* d = &c->dl[i];
            const int R = 4 << i;
            const size_t npix = (size_t)R * R;
            nchw_to_nhwc(feats[i] + (size_t)s * npix * d->I, R, R, d->I, fin);
            /* cvt_block: conv3x3+bias -> BN -> LeakyReLU -> Dropout(identity), reference networks_seg.py:64-79 */
            if (use_wino43(R, R, d->I, d->F, c->bf16)) conv3x3_wino43(fin, R, R, d->I, d->cvt_u, d->F, ya);
            else if (use_wino(R, R, d->F, 0, c->bf16)) conv3x3_wino(fin, R, R, d->I, d->cvt_u, d->F, ya);
            else conv3x3(fin, R, R, d->I, 0, d->cvt_w, d->F, ya, c->bf16, 1);
            bias_bn_act(ya, npix, d->F, d->cvt_b, d->cvt_s, d->cvt_rm, d->cvt_beta);
            store_bf16(ya, npix * d->F, c->bf16);
            /* concat(prev, cvt) on channels, reference :108-109 */
            if (i > c->d_s0) {
                for (size_t p = 0; p < npix; ++p) {
                    memcpy(cat + p * d->in_c, prev + p * d->F, sizeof(float) * d->F);
                    memcpy(cat + p * d->in_c + d->F, ya + p * d->F, sizeof(float) * d->F);
                }
            } else {
                memcpy(cat, ya, sizeof(float) * npix * d->F);
            }
            if (!d->is_last) {
                /* main_block: nearest x2 -> DecoderResBlock, reference :7-46, :86-88 */
                const int R2 = 2 * R;
                const size_t np2 = (size_t)R2 * R2;
                if (R2 >= 16 && use_wino22(c->bf16)) deconv4x4s2_wino(cat, R, R, d->in_c, d->a_w, d->cs, ya);
                else if (R2 >= 16) deconv4x4s2(cat, R, R, d->in_c, d->a_w, d->cs, ya, c->bf16);   /* sub-pixel up+conv */
                else conv3x3(cat, R, R, d->in_c, 1, d->a_w, d->cs, ya, c->bf16, 1);
                bias_bn_act(ya, np2, d->cs, d->a_b, d->a_s, d->a_rm, d->a_beta);
                store_bf16(ya, np2 * d->cs, c->bf16);
                if (use_wino(R2, R2, d->cs, 0, c->bf16)) conv3x3_wino(ya, R2, R2, d->cs, d->b_u, d->cs, yb);
                else conv3x3(ya, R2, R2, d->cs, 0, d->b_w, d->cs, yb, c->bf16, 1);
                bias_bn_act(yb, np2, d->cs, d->b_b, d->b_s, d->b_rm, d->b_beta);
#pragma omp parallel for schedule(static)
                for (int y = 0; y < R2; ++y)
                    for (int x = 0; x < R2; ++x) {
                        const float* src = cat + ((size_t)(y >> 1) * R + (x >> 1)) * d->in_c;
                        float* dst = prev + ((size_t)y * R2 + x) * d->cs;
                        const float* yv = yb + ((size_t)y * R2 + x) * d->cs;
                        for (int o = 0; o < d->cs; ++o) {
                            float sc;
                            if (d->has_sc) {
                                float acc = 0.0f;
                                const int bf = c->bf16;
                                for (int k0 = 0; k0 < d->in_c; ++k0) {      /* the MFMA order of the 16-channel blocks */
                                    const int ch = (k0 & ~15) | CPERM(k0 & 15);
                                    acc = fmaf(OPND(src[ch]), OPND(d->sc_w[(size_t)ch * d->cs + o]), acc);
                                }
                                sc = acc + d->sc_b[o];
                                if (c->bf16) sc = bf16r(sc);       /* the shortcut is a stored tensor of its own */
                            } else {
                                sc = src[o];
                            }
                            dst[o] = sc + yv[o];
                            if (c->bf16) dst[o] = bf16r(dst[o]);
                        }
                    }
            } else {
                /* final conv3x3 + bias -> logits, then argmax (first maximum), reference :91-92, seg_solver.py:326 */
                const int nc = d->cs;
                conv3x3(cat, R, R, d->in_c, 0, d->f_w, nc, ya, 0, 0);   /* final conv: fp32 in both modes, ascending channel order */
                for (size_t p = 0; p < npix; ++p) {
                    int best = 0;
                    float bv = 0.0f;
                    for (int o = 0; o < nc; ++o) {
                        float v = ya[p * nc + o] + d->f_b[o];
                        if (logits) logits[((size_t)s * nc + o) * npix + p] = v;
                        if (o == 0 || v > bv) { bv = v; best = o; }
                    }
                    if (mask) mask[(size_t)s * npix + p] = (uint8_t)best;
                }
            }
        }
    }
    free(fin); free(cat); free(ya); free(yb); free(prev);
    return GSA_OK;
}

GSAO_API int gsao_generate(gsao_ctx* c, void* stream, int32_t n, const float* z, const float* const* noise, int32_t num_noise,
                           uint8_t* img, uint8_t* mask) {
    if (!c || !c->g_ready || !c->d_ready) return fail(c, GSA_ERR_STATE, "commit generator and decoder first%s (%ld)", "", 0);
    const int nlev = c->nlev;
    if (num_noise != 2 * nlev) return fail(c, GSA_ERR_INVALID, "noise plane count does not match this generator%s (%ld)", "", num_noise);
    if (c->d_n != nlev) return fail(c, GSA_ERR_INVALID, "decoder expects %s%ld features, generator yields a different count", "", c->d_n);
    float* feats[MAX_LEVELS];
    const float* cf[MAX_LEVELS];
    int rc = GSA_OK;
    for (int s = 0; s < n && rc == GSA_OK; ++s) {
        const float* nz[2 * MAX_LEVELS];
        for (int l = 0; l < nlev; ++l) {
            if (c->d_inch[l] != c->ch[l]) return fail(c, GSA_ERR_INVALID, "decoder in_channels do not match generator features%s (%ld)", "", l);
            const size_t npix = (size_t)1 << (2 * (l + 2));
            feats[l] = (float*)malloc(sizeof(float) * npix * c->ch[l]);
            cf[l] = feats[l];
            nz[2 * l] = noise[2 * l] + (size_t)s * npix;
            nz[2 * l + 1] = noise[2 * l + 1] + (size_t)s * npix;
        }
        const size_t R = (size_t)1 << c->gc.max_res_log2;
        rc = gsao_generator_forward(c, stream, 1, z + (size_t)s * c->gc.latent_size, nz, 2 * nlev, NULL,
                                    img ? img + (size_t)s * R * R * c->gc.channels : NULL, feats, nlev);
        if (rc == GSA_OK) rc = gsao_decoder_forward(c, stream, 1, cf, nlev, NULL, mask ? mask + (size_t)s * R * R : NULL);
        for (int l = 0; l < nlev; ++l) free(feats[l]);
    }
    return rc;
}
