// Kernel launch interface between the host orchestration (gsa_api.cpp) and the HIP kernels
// (gsa_kernels.hip).  Internal; the public boundary is include/gsa.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsa {

// Per-(sample, channel) AdaIN coefficients: out = fmaf(x, A, B), B = fmaf(-mean, A, beta*(ys+1)+yb)
// (InstanceNorm + style of reference networks_stylegan.py:250-264, folded; `mean` is kept for reference only).
struct Aff { float mean, A, B, pad; };

// 64-bit fixed-point partial statistics of one (sample, tile-row, channel)
struct StatPart { unsigned long long s1, s2; };

constexpr double kStatScale1 = 268435456.0;  // 2^28: quad sums
// quad sums of squares: 2^S2 with S2 = clamp(40 - ceil(log2(H*W)), 20, 26), a static function of the plane size (round 4):
// 20 at 1024^2 -- where the 64-bit sum of a plane holds rms(x) < 2.9e3 -- 22 / 24 at 512^2 / 256^2, 26 from 128^2 down.
// Small planes have few quads to average the rounding of rint(q * 2^S2) over, and E[x^2] - mean^2 must survive the cancellation
// when a plane's values sit on a bias far above their spread (device function stat_scale2 in gsa_kernels.hip; oracle: stat_scale2)

enum Epilogue { EPI_RAW = 0, EPI_SYNTH = 1, EPI_DEC = 2 };

struct ConvParams {
    int device;      // HIP device of the launching context (per-device launch state in the launchers)
    // input: up to two NHWC sources concatenated on channels (C0 then C1), both multiples of 16
    const float* src0; const Aff* aff0; int C0;   // aff0 may be null (identity)
    const float* src1; int C1;                    // src1 may be null
    int Hs, Ws;      // spatial size of the sources
    int up;          // 1: logical input is the nearest x2 upsample of the sources
    int H, W;        // output size
    const float* wpk;   // packed weights [Cout/16][Cin/16][tap][ci][16][cg]
    const float* wino;  // or null: the same conv in Winograd F(2x2,3x3) form, U packed [Cout/16][Cin/16][f16][ci][16][cg]
                        // (F(4x4,3x3) layers, conv_uses_wino43: [Cout/16][Cin/8][f36][kq][16][2], channel 8b + 2kq + j)
    int Cout;           // total output channels
    float* out;         // NHWC
    // EPI_SYNTH: AddNoise -> Bias -> LeakyReLU -> statistics
    const float* noise; const float* nscale; const float* nbias;
    StatPart* partials; int prow;      // prow = partial rows per sample
    // EPI_DEC: conv bias -> BatchNorm(inference) -> LeakyReLU [-> + resid], folded on the host into ONE fma per element:
    // y = lrelu(fmaf(v, bn_s, bn_beta)) with bn_s = gamma / sqrt(running_var + eps), bn_beta = fmaf(bias - running_mean, bn_s, beta)
    const float* bn_s; const float* bn_beta;
    const float* resid; int resid_up;   // resid_up: residual lives at half resolution (identity shortcut)
    const float* resid1; int res_c0;    // residual = concat(resid [res_c0 channels], resid1 [Cout - res_c0]); resid1 null: one tensor
    // fused 1x1 shortcut of DecoderResBlock (second output)
    const float* wsc; const float* sc_bias; float* out_sc;
    int tiles_x, tiles_y, groups, total_tiles;   // filled by the launcher
    int stats_direct;                            // filled by the launcher: EPI_SYNTH sums go to acc with atomics (no partial rows)
    int* stat_rows_host;                         // host pointer or null: the launcher reports the partial rows it used (0 = direct)
    int w_resident;                              // filled by the launcher: whole weight panel LDS-resident (conv3x3 DB form)
    int group_minor;                             // filled by the launcher (conv3x3_wino): 1-D grid, the channel groups of a tile consecutive on one XCD
    unsigned long long* stamps;   // diagnostic build (-DGSA_STAMP) only: per-phase cycle sums
    int dbg;                      // diagnostic build only: bit0 = stage pixel 0 everywhere (timing of a cache-resident input)
    int prio;                     // experiment (GSA_PRIO): 1 = the wave OUTSIDE its MFMA phase gets the higher issue priority
    int bf16;                     // 1: bf16 MFMA mode -- wpk/wsc hold bf16 packs [..][tap][kq][16][4], operands rounded at staging
    // EPI_SYNTH in a kernel whose workgroup holds a WHOLE plane (conv3x3_ksplit at 4 and 8 px: conv_fuses_finalize): the workgroup writes
    // the AdaIN coefficients itself instead of partial rows -- the finalize launch of the layer is skipped (fin_aff null: not fused)
    const float* fin_style; int fin_style_stride; const float* fin_gamma; const float* fin_beta; Aff* fin_aff; unsigned* fin_flags;
    const float* zeros;           // >= 64 bytes of zeros in device memory: what conv3x3_wino43's LDS-DMA reads for a pixel outside the image
    // toRGB fused into the convolution that stages the SAME tensor with the SAME AdaIN (round 5: the decoder's last cvt conv reads the generator's
    // last feature, as toRGB does): rgb_img non-null = also write the uint8 image (N, H, W, 3) from the staged, AdaIN-applied tile
    const float* rgb_w; const float* rgb_b; uint8_t* rgb_img;
};

struct PostParams {
    const float* src;      // raw conv_1 output NHWC, or the constant tensor [H][W][C]
    int src_per_sample;    // 1: src indexed by sample; 0: broadcast (constant tensor)
    const float* blur;     // [C][9] or null (no blur)
    const float* noise; const float* nscale; const float* nbias;
    float* out; StatPart* partials; int prow;
    int H, W, C;
    int bf16;              // 1: the per-sample source and the output are bf16 tensors (bf16 mode)
    int row_groups;        // filled by the launcher (post_rows_kernel<4>): groups of 4 rows a thread walks
};

struct FinalizeParams {
    const StatPart* partials; int prow; int HW; int C;
    StatPart* acc;     // [n][C] zero-initialised accumulators, cleared again by finalize
    unsigned* tickets; // [n][ceil(C/64)] zero-initialised arrival counters, cleared again by finalize
    const float* style; int style_stride;  // style[n*style_stride + c] = ys, [.. + C + c] = yb
    const float* gamma; const float* beta;
    Aff* aff;   // [n][C]
    unsigned* flags;   // sticky device word or null: set to 1 when a sum is within a factor 4 of its 64-bit wrap or the variance is negative
};

// launches (all stream-ordered, no sync)
hipError_t launch_conv3x3(const ConvParams& p, int epi, bool shortcut, int n, hipStream_t s);
bool conv_uses_ws(const ConvParams& p, int epi, bool shortcut, int n);
bool conv_uses_ksplit(const ConvParams& p, bool shortcut);         // true: 4-way K split form (static rule: layer shape only)
bool conv_uses_wino43(const ConvParams& p, int epi, bool shortcut); // true: Winograd F(4x4,3x3) form (static rule: layer shape only; p.wino then holds the 36-frequency panel)
bool wino43_enabled();                                              // GSA_WINO43 != 0
bool conv_uses_wino(const ConvParams& p, int epi, bool shortcut);   // true: Winograd form (static rule: layer shape only)   // true: wave-specialised kernel, no partial rows
// gsa_wino_lean.hip (round 5): the Winograd layers with one 16-channel input block and one 16-channel output group in a leaner
// instruction stream -- speed only, the same arithmetic and bits as conv3x3_wino (GSA_WINO_LEAN=0 keeps conv3x3_wino)
bool wino_lean_applies(const ConvParams& p, int epi);
const char* wino_lean_name(const ConvParams& p, int epi, int n);
hipError_t launch_wino_lean(const ConvParams& p, int epi, int n, hipStream_t s);
bool wino_lean_fuses_torgb(const ConvParams& p, int epi, int nc);      // the lean kernel's own conditions
bool conv_fuses_torgb(const ConvParams& p, int epi, bool shortcut, int nc);      // true: launch_conv3x3 on p (rgb_* set) also produces toRGB's uint8 image
// gsa_bf16_lean.hip (round 5): the bf16 mode's 3x3 convolutions from 32 px on in the lean form (GSA_BF16_LEAN=0: conv3x3_mfma<..., true>)
bool bf16_lean_applies(const ConvParams& p, int epi, bool shortcut);
const char* bf16_lean_name(const ConvParams& p, int epi, int n);
hipError_t launch_bf16_lean(const ConvParams& p, int epi, int n, hipStream_t s);
// gsa_sub_lean.hip (round 5): subpixel_res<..., WINO> (fp32) in a leaner instruction stream -- speed only, same bits (GSA_SUB_LEAN=0: subpixel_res)
bool subpixel_lean_applies(const ConvParams& p, int nt, int epi, bool sc, int kb, bool wst);
const char* subpixel_lean_name(const ConvParams& p, int nt, int epi, bool sc, int kb, bool wst);
hipError_t launch_subpixel_lean(const ConvParams& q, int nt, int epi, bool sc, int kb, bool wst, dim3 grid, hipStream_t s);
hipError_t launch_subpixel(const ConvParams& p, int epi, bool shortcut, int n, hipStream_t s);   // deconv4x4s2 / sub-pixel up+conv
bool subpixel_uses_wino(const ConvParams& p);                      // true: Winograd F(2x2,2x2) form (static rule: fp32 mode): 9 products per 2x2 class outputs instead of 163x3
hipError_t launch_post(const PostParams& p, int n, hipStream_t s);
// gsa_post_lean.hip (round 5): post_rows_kernel<4> in packed fp32 arithmetic (same bits).  GSA_POST_PK: 0 = off, 1 = packed (default), 2 = + non-temporal stores
int post_pk_mode();
bool post_dma_applies(const PostParams& p);      // GSA_POST_DMA: the LDS-DMA ring form (16 / 32 channels, fp32, blurred)
int post_dma_band(const PostParams& p);
int post_dma_blocks(const PostParams& p);
hipError_t launch_post_dma(const PostParams& p, int n, hipStream_t s);
hipError_t launch_post_pk(const PostParams& q, dim3 grid, size_t lds, hipStream_t s);
hipError_t launch_finalize(const FinalizeParams& p, int n, hipStream_t s);
bool conv_fuses_finalize(const ConvParams& p, int epi, bool shortcut);   // true: launch_conv3x3 writes the layer's AdaIN coefficients itself when p.fin_aff is set (whole-plane K-split tiles)
bool post_fuses_finalize(const PostParams& p);      // true: launch_post_fin does the post pass AND the finalize of this plane in one launch (planes <= 32 x 32)
hipError_t launch_post_fin(const PostParams& p, const FinalizeParams& f, int n, hipStream_t s);
hipError_t launch_pixelnorm(const float* z, float* out, int n, int L, hipStream_t s);
// PixelNorm + the eight mapping layers in one launch (result in lat[0]); mapping_fused: whether it applies to this latent size
constexpr int kMapSlices = 8;     // most sample slices of the fused mapping network (launch-number words ctl[2 .. 2 + kMapSlices))
bool mapping_fused(int L, int device);
hipError_t launch_mapping(const float* z, float* const* wt, float* const* b, unsigned long long* const* ll, float* out, unsigned* ctl,
                          int n, int L, int device, hipStream_t s, int drop_workgroups = 0);   // drop_workgroups: fault injection (tests)
hipError_t launch_dense(const float* x, const float* WT, const float* b, float* y, int n, int K, int J,
                        int lrelu, hipStream_t s);
hipError_t launch_styles(const float* w, const float* avg, const float* psi, const float* WT, const float* b,
                         const int* col_layer, float* styles, int n, int K, int J, hipStream_t s);
hipError_t launch_torgb(const float* x, const Aff* aff, const float* w, const float* b, float* rgb,
                        uint8_t* img, int n, int H, int W, int C, int nc, int bf16, hipStream_t s);
hipError_t launch_export_nchw(const float* x, const Aff* aff, float* out, int n, int H, int W, int C, int bf16, hipStream_t s);
hipError_t launch_import_nhwc(const float* in, float* out, int n, int H, int W, int C, int bf16, hipStream_t s);
hipError_t launch_final_conv(const float* src0, int C0, const float* src1, int C1, const float* wpk,
                             const float* bias, float* logits, uint8_t* mask, int n, int H, int W, int ncls, int bf16,
                             hipStream_t s);

// geometry the launchers pick (weights are packed per 16 output channels, so it is free to vary)
int conv_stat_rows(int H, int W, int Cout, int n);   // statistic partial rows per sample written by conv3x3 EPI_SYNTH
int post_prow(int H, int W, int C);                  // ... written by the post kernel (upper bound: the one-row form)
int post_rows_used(const PostParams& p);             // ... by the form launch_post picks for p
const char* subpixel_kernel_name(const ConvParams& p, int epi, bool sc, int n);
hipError_t launch_fill_normal(float* out, int per_sample, int n, unsigned long long first_index, unsigned plane,
                              unsigned long long seed, hipStream_t s);
hipError_t launch_seg_eval(const float* logits, const int8_t* labels, int n, int classes, int H, int W,
                           unsigned long long* confusion, unsigned long long* loss_fixed, hipStream_t s);
const char* conv3x3_kernel_name(const ConvParams& p, int epi, bool sc, int n);
const char* conv_geom_name(int H, int W, int Cout, int n);   // "tile16,cout64" -- for profile labels
const char* subpixel_geom_name(int H, int W, int Cout, int n);

}  // namespace gsa
