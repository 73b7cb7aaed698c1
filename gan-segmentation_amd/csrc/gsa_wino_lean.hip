// Lean Winograd F(2x2,3x3) kernels for the layers with ONE 16-channel input block and ONE 16-channel output group
// (round 5): g.1024.conv_2, d.cvt_8, d.main_7.b of the FFHQ path (16 -> 16 channels at 1024^2) and their twins in other
// configurations.  Reference operators: Conv2DW 3x3 + AddNoise + Bias + LeakyReLU (networks_stylegan.py:354-457, 267-305,
// 534-545), nn.Conv2D + BatchNorm + LeakyReLU [+ residual] (networks_seg.py:7-46, 64-79).
//
// Same canonical arithmetic as conv3x3_wino (gsa_kernels.hip; oracle/c/gsa_oracle.c conv3x3_wino): same transforms in the same
// order, the same k-ordered MFMA chains, the same epilogue operations -- the bits do not change.  What changes is the
// instruction stream.  The ISA of conv3x3_wino<2, 1, false, true, 1> (d.cvt_8) carries ~480 vector-ALU instructions, ~250 scalar
// ones and 96 LDS reads per (tile, wave) item beside its 64 MFMAs, and f32 MFMAs share the SIMD's issue time with the vector ALU
// (DESIGN.md section 4): the kernel is bound by its own instruction count, not by occupancy.  This file spends them only where
// the shape needs them:
//   * the shape is a compile-time fact (one channel block, one output group, weight panel resident): no block loop, no group
//     index, no streamed-weight path, an item IS a tile and every item ends in its epilogue;
//   * staging by 16-byte chunks with the thread's four AdaIN coefficient pairs in registers for the whole sample (a thread's
//     channels never change): two packed fma per chunk, no coefficient table in LDS, no table reads;
//   * global addresses are a wave-uniform base (scalar ALU) plus per-thread constants; zero padding costs instructions only on
//     the tiles that touch the image border (one packed 24-bit flag word per thread);
//   * the MFMAs take the weights as the A operand and the transformed patch as B (a*b = b*a, same k order: same bits), so a
//     lane ends up with FOUR CONSECUTIVE OUTPUT CHANNELS of its own Winograd tile: the 2x2 output pixels of the tile are four
//     16-byte NHWC stores as they are -- no quad transpose (64 DPP / select instructions per item in conv3x3_wino), the
//     per-channel constants are packed pairs, bias / BatchNorm / LeakyReLU run as packed operations;
//   * the first MFMA of every chain starts from the inline constant 0: the accumulators are never cleared.
#include "gsa_kernels.h"
#include "gsa_dev.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace gsa {
namespace lean {

constexpr int LH = 18, LW = 18, RS = LW * 16 + 4;      // halo image: 18 rows of 18 pixels x 16 channels, row stride 292 floats (bank-conflict free patch reads)
constexpr int IMG = LH * RS;                           // floats per image buffer (21 KB)
constexpr int SEG = 16 * 256;                          // U floats of the (16 couts, 16 channels) panel: [f][kq][16][cg]
constexpr int NPIX = LH * LW;                          // 324 halo pixels = 1296 chunks = 5 rounds of 256 threads + 16 chunks

struct Tile { int n, y0, x0; };

// EPI_DEC: y = lrelu(fmaf(v, bn_s, bn_beta)) [+ residual at half resolution (RES)];  EPI_SYNTH: y = lrelu((v + nscale * noise) + nbias) and the
// instance-norm statistics of y (direct form: per-wave integer sums, one atomic per channel and sample range);  AFF: the source carries AdaIN coefficients.
// NB = 16-channel input blocks (1: 16 input channels, 2: 32), GW = 16-channel output groups per workgroup = all of the layer's (1: 16 output
// channels, 4 waves; 2: 32, 8 waves -- waves 0-3 / 4-7 multiply the SAME staged image by the two groups' panels, as conv3x3_wino<..., GW = 2>).
// An item is (tile, block); the accumulators live across the NB items of a tile, the last one ends in the epilogue.
// DB = false (NB = 1, GW = 1 only): ONE image buffer and two barriers per item -- the patch is read and transformed into registers, then the
// buffer is refilled with the next item while the MFMAs run out of registers: 37 KB of LDS per workgroup, so THREE workgroups per CU when
// the kernel also fits 168 registers (more waves instead of bigger items).
// PF = register sets of the activation prefetch: 1 = item it+2 is loaded while item it is multiplied (one item time between a load and its
// use); 2 = item it+3 (two item times).  The 16 -> 16 layers at 1024^2 read and write 1.07 GB per launch -- they sit on the HBM roof, and
// the stamps of the PF = 1 form (profiles/r05_stamps_lean.txt: write phase ~2100 of ~8100 cycles per item for six chunk stores) show the
// staging writes WAITING for loads issued one item earlier: under a saturated memory system one item time (~3.5 us) does not cover a load.
// RGB (16 -> 16, decoder epilogue, AdaIN source): toRGB + _transform_gan_back (reference networks_stylegan.py:118-126, image_generator.py:76-84) of
// the SAME tile out of the staged LDS image -- the decoder's last cvt conv and toRGB both read the generator's last feature with the same AdaIN
// coefficients, and the staged value fmaf(x, A, B) is exactly what toRGB's chain starts from; thread = pixel of the 16x16 tile, the canonical
// k-ascending fmaf chain per colour, the uint8 bytes of four pixels packed into three dwords per lane quad.  Saves a 0.5 GB read per launch.
template <int EPI, bool AFF, bool RES, int NB, int GW, bool DB = true, int PF = 1, bool RGB = false>
__global__ __launch_bounds__(256 * GW, GW == 1 ? (DB ? 2 : 3) : 1) void conv3x3_wino_lean(ConvParams p) {
    static_assert(!RGB || (NB == 1 && GW == 1 && DB && AFF && EPI == EPI_DEC), "fused toRGB: the 16 -> 16 decoder kernel with an AdaIN source");
    static_assert(DB || (NB == 1 && GW == 1), "single-buffered form: one block, one group");
    static_assert(PF == 1 || (PF == 2 && DB), "two prefetch sets: the double-buffered form");
    constexpr int NTHR = 256 * GW, CIN = 16 * NB, COUT = 16 * GW;
    constexpr int NCH = (NPIX * 4 + NTHR - 1) / NTHR;      // staging rounds per item: 6 (5 full + 16 chunks) with 256 threads, 3 (2 full + 272) with 512
    constexpr bool kLastPartial = (NPIX * 4) % NTHR != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sA = smem;                 // [DB ? 2 : 1][IMG]
    float* const sB = smem + (DB ? 2 : 1) * IMG;       // [GW][NB][SEG]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, gsel = wave8 >> 2;
    const int i16 = lane & 15, kq = lane >> 4, part = tid & 3;
    const int H = p.H, W = p.W;
    const int per = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int w_begin = xcd_block(blockIdx.x, gridDim.x) * per;
    const int w_end = min(p.total_tiles, w_begin + per);
    if (w_begin >= w_end) return;
    const int items = (w_end - w_begin) * NB;

    {   // the whole weight panel of the layer ([Cout/16][Cin/16][SEG], 16 KB per (group, block)), once
        const f32x4* src = reinterpret_cast<const f32x4*>(p.wpk);
        f32x4 r[4];
#pragma unroll
        for (int h = 0; h < NB; ++h) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = src[tid + NTHR * (4 * h + j)];
#pragma unroll
            for (int j = 0; j < 4; ++j) reinterpret_cast<f32x4*>(sB)[tid + NTHR * (4 * h + j)] = r[j];
        }
    }

    // ---- staging: chunk q = tid + NTHR k = (halo pixel q >> 2, channels 4 * part .. of the item's block); byte offsets from the halo origin
    unsigned s_off[NCH], eflags = 0;
    int l_off[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int pix = (tid + NTHR * k) >> 2;
        const bool real = pix < NPIX;
        const int ly = real ? pix / LW : 1, lx = real ? pix % LW : 1;
        s_off[k] = (unsigned)(((ly * W + lx) * CIN + part * 4) * 4);
        l_off[k] = real ? ly * RS + lx * 16 + part * 4 : -1;
        eflags |= (unsigned)((ly == 0 ? 1 : 0) | (ly == LH - 1 ? 2 : 0) | (lx == 0 ? 4 : 0) | (lx == LW - 1 ? 8 : 0)) << (4 * k);
    }
    const unsigned safe_off = (unsigned)((((W + 1) * CIN) + part * 4) * 4);      // the tile's own first pixel: always inside the image
    auto edge_code = [&](const Tile& t) { return (t.y0 == 0 ? 1 : 0) | (t.y0 + 16 == H ? 2 : 0) | (t.x0 == 0 ? 4 : 0) | (t.x0 + 16 == W ? 8 : 0); };
    auto advance = [&](Tile& t) { t.x0 += 16; if (t.x0 == W) { t.x0 = 0; t.y0 += 16; if (t.y0 == H) { t.y0 = 0; t.n += 1; } } };

    f32x4 ra[PF][NCH];                        // the item(s) in flight (loaded, not yet written to LDS); item j lives in set j % PF
    f32x4 cA[NB], cB[NB];                     // AdaIN A / B of this thread's four channels of every block, for the sample being staged
#pragma unroll
    for (int h = 0; h < NB; ++h) cA[h] = cB[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    int n_coef = -1;
    auto load_item = [&](auto set_tag, const Tile& t, int e, int cb) {
        constexpr int S = decltype(set_tag)::value;
        const char* hb = reinterpret_cast<const char*>(p.src0) + ((((long)(t.n * H + t.y0) * W + t.x0) - (W + 1)) * CIN + cb * 16) * 4;
        // the last round holds fewer real chunks than threads; everybody issues it -- the idle lanes read the tile's first pixel
        if (e) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const unsigned bad = (eflags >> (4 * k)) & (unsigned)e;
                const unsigned off = s_off[k] + (bad ? safe_off - s_off[k] : 0u);
                ra[S][k] = *reinterpret_cast<const f32x4*>(hb + off);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) ra[S][k] = *reinterpret_cast<const f32x4*>(hb + s_off[k]);
        }
    };
    auto write_item = [&](auto set_tag, const Tile& t, int e, int cb, int buf) {
        constexpr int S = decltype(set_tag)::value;
        float* img = sA + buf * IMG;
        if (AFF && t.n != n_coef) {           // wave-uniform: the first item and sample changes (a workgroup's range spans at most a few samples)
#pragma unroll
            for (int h = 0; h < NB; ++h) {
                const f32x4* a = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)t.n * CIN + h * 16 + part * 4);      // (mean, A, B, -) x 4
                const f32x4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
                cA[h] = f32x4{a0[1], a1[1], a2[1], a3[1]};
                cB[h] = f32x4{a0[2], a1[2], a2[2], a3[2]};
            }
            n_coef = t.n;
        }
        const f32x4 kA = NB == 1 || cb == 0 ? cA[0] : cA[NB - 1], kB = NB == 1 || cb == 0 ? cB[0] : cB[NB - 1];
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (e) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const f32x4 v = AFF ? fma4(ra[S][k], kA, kB) : ra[S][k];
                const bool bad = ((eflags >> (4 * k)) & (unsigned)e) != 0;
                if (!kLastPartial || k < NCH - 1 || l_off[NCH - 1] >= 0) *reinterpret_cast<f32x4*>(img + l_off[k]) = bad ? z : v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (!kLastPartial || k < NCH - 1 || l_off[NCH - 1] >= 0) *reinterpret_cast<f32x4*>(img + l_off[k]) = AFF ? fma4(ra[S][k], kA, kB) : ra[S][k];
        }
    };

    // ---- operands: lane (i16, kq) owns Winograd tile (ty, tx) = (2 b3 + b1, 2 b2 + b0) of its wave's 8x8 quadrant and k slot kq
    const int wty = ((i16 >> 3) & 1) * 2 + ((i16 >> 1) & 1), wtx = ((i16 >> 2) & 1) * 2 + (i16 & 1);
    const int qy = wave >> 1, qx = wave & 1;
    const int pbase = (qy * 8 + 2 * wty) * RS + (qx * 8 + 2 * wtx) * 16 + kq * 4;
    const int bbase = gsel * (NB * SEG) + (kq * 16 + i16) * 4;
    // ---- epilogue: the lane's tile = output pixels (2 wty + i, 2 wtx + j) of the quadrant, channels 16 gsel + 4 kq .. + 3
    const int co4 = gsel * 16 + kq * 4;
    const unsigned out_off = (unsigned)((((qy * 8 + 2 * wty) * W + qx * 8 + 2 * wtx) * COUT + co4) * 4);      // bytes from the tile origin
    const unsigned res_off = RES ? (unsigned)((((qy * 4 + wty) * (W >> 1) + qx * 4 + wtx) * COUT + co4) * 4) : 0u;
    f32x4 e2 = {0.f, 0.f, 0.f, 0.f}, e3 = e2;
    if (EPI == EPI_DEC) {
        e2 = *reinterpret_cast<const f32x4*>(p.bn_s + co4);
        e3 = *reinterpret_cast<const f32x4*>(p.bn_beta + co4);
    }
    // EPI_SYNTH: v = lrelu((v + nscale * noise) + nbias), then the instance-norm statistics of the stored values
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0;
    if (EPI == EPI_SYNTH) {
        e0 = *reinterpret_cast<const f32x4*>(p.nscale + co4);
        e1 = *reinterpret_cast<const f32x4*>(p.nbias + co4);
    }
    const unsigned nz_off = (unsigned)((((qy * 8 + 2 * wty) * W + qx * 8 + 2 * wtx)) * 4);      // bytes from the tile origin in the (N, 1, H, W) noise plane
    f32x2 nz[2] = {{0.f, 0.f}, {0.f, 0.f}};                                                       // noise of tile row i at x = 2 wtx, 2 wtx + 1
    const int s2 = stat_s2(H * W);
    const bool odd = (lane & 1) != 0;
    unsigned long long dI1[4] = {0ull, 0ull, 0ull, 0ull}, dI2[4] = {0ull, 0ull, 0ull, 0ull};      // direct statistics of the lane's four channels
    auto flush_stats = [&](int n) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned long long I1 = dI1[c], I2 = dI2[c];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { I1 += shfl_xor_u64(I1, m); I2 += shfl_xor_u64(I2, m); }
            if (i16 == 0) {
                StatPart* a = p.partials + ((size_t)n * p.prow + (blockIdx.x & (kDirectRows - 1))) * COUT + co4 + c;
                atomicAdd(&a->s1, I1);
                atomicAdd(&a->s2, I2);
            }
            dI1[c] = dI2[c] = 0ull;
        }
    };
    const f32x2 k02 = {0.2f, 0.2f};
    f32x4 rr = {0.f, 0.f, 0.f, 0.f};
    auto epilogue_loads = [&](const Tile& t) {
        if (EPI == EPI_SYNTH) {
            const char* nb = reinterpret_cast<const char*>(p.noise) + ((long)(t.n * H + t.y0) * W + t.x0) * 4;
            nz[0] = *reinterpret_cast<const f32x2*>(nb + nz_off);
            nz[1] = *reinterpret_cast<const f32x2*>(nb + (long)W * 4 + nz_off);
        }
        if (RES) {
            const char* rb = reinterpret_cast<const char*>(p.resid) + ((long)(t.n * (H >> 1) + (t.y0 >> 1)) * (W >> 1) + (t.x0 >> 1)) * (COUT * 4);
            rr = *reinterpret_cast<const f32x4*>(rb + res_off);
        }
    };

    auto torgb = [&](const Tile& t, int buf) {      // after the item's multiply, before its closing barrier: the image buffer still holds the tile
        const float* px = sA + buf * IMG + ((tid >> 4) + 1) * RS + ((tid & 15) + 1) * 16;
        f32x4 f4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f4[k] = *reinterpret_cast<const f32x4*>(px + 4 * k);
        // the 48 weights and 3 biases are read through the constant address space: wave-uniform scalar loads (as plain global pointers next to
        // the kernel's own stores they became twelve vector loads per item)
        const __attribute__((address_space(4))) float* cw = (const __attribute__((address_space(4))) float*)p.rgb_w;
        const __attribute__((address_space(4))) float* cb = (const __attribute__((address_space(4))) float*)p.rgb_b;
        float a3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int o = 0; o < 3; ++o) a3[o] = fmaf(f4[c >> 2][c & 3], cw[o * 16 + c], a3[o]);
        unsigned pk = 0;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float u = ((a3[o] + cb[o]) + 1.0f) * 0.5f;
            u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
            pk |= (unsigned)(uint8_t)(255.0f * u) << (8 * o);
        }
        // four pixels of a lane quad = 12 bytes = three dwords: lane j of the quad stores dword j = (P_j >> 8j) | (P_{j+1} << (24 - 8j))
        const unsigned nxt = (unsigned)__builtin_amdgcn_mov_dpp((int)pk, 0xF9, 0xF, 0xF, true);      // quad_perm [1, 2, 3, 3]
        const int j = tid & 3;
        const unsigned dw = (pk >> (8 * j)) | (nxt << (24 - 8 * j));
        if (j < 3) {
            const long pix = ((long)(t.n * H + t.y0 + (tid >> 4)) * W + t.x0 + (tid & 12));      // first pixel of the quad
            *reinterpret_cast<unsigned*>(p.rgb_img + pix * 3 + 4 * j) = dw;
        }
    };
    f32x4 acc[16], V[16];
    auto transform = [&](int buf) {
        const float* a_img = sA + buf * IMG + pbase;
        // input transform V = B^T d B: rows t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3; then the same along the columns
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(a_img + 0 * RS + c * 16);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(a_img + 1 * RS + c * 16);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(a_img + 2 * RS + c * 16);
            const f32x4 d3 = *reinterpret_cast<const f32x4*>(a_img + 3 * RS + c * 16);
            V[0 * 4 + c] = sub4(d0, d2);
            V[1 * 4 + c] = add4(d1, d2);
            V[2 * 4 + c] = sub4(d2, d1);
            V[3 * 4 + c] = sub4(d1, d3);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 t0 = V[i * 4 + 0], t1 = V[i * 4 + 1], t2 = V[i * 4 + 2], t3 = V[i * 4 + 3];
            V[i * 4 + 0] = sub4(t0, t2);
            V[i * 4 + 1] = add4(t1, t2);
            V[i * 4 + 2] = sub4(t2, t1);
            V[i * 4 + 3] = sub4(t1, t3);
            valu_settle(V[i * 4 + 0], V[i * 4 + 1], V[i * 4 + 2], V[i * 4 + 3]);
        }
    };
    auto mfmas = [&](auto first_tag, int cb) {
        constexpr bool FIRST = decltype(first_tag)::value;      // the tile's first block: every chain starts from the inline constant 0
        const float* b_img = sB + bbase + cb * SEG;
        // sixteen chains, k ascending (channel 4 kq + cg at k slot kq of MFMA cg); weights = A operand, patch = B operand
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            f32x4 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const f32x4*>(b_img + (fb * 4 + j) * 256);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = fb * 4 + j;
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[j][cg], V[f][cg], FIRST && cg == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[f], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto multiply = [&](auto first_tag, int buf, int cb) { transform(buf); mfmas(first_tag, cb); };
    auto epilogue = [&](const Tile& t) {
        // output transform Y = A^T M A: rows s0 = (M0 + M1) + M2, s1 = (M1 - M2) - M3, then the same along the columns
        mfma_settle(acc);
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = add4(add4(acc[j], acc[4 + j]), acc[8 + j]);
            s1[j] = sub4(sub4(acc[4 + j], acc[8 + j]), acc[12 + j]);
        }
        f32x4 y[2][2];
        y[0][0] = add4(add4(s0[0], s0[1]), s0[2]); y[0][1] = sub4(sub4(s0[1], s0[2]), s0[3]);
        y[1][0] = add4(add4(s1[0], s1[1]), s1[2]); y[1][1] = sub4(sub4(s1[1], s1[2]), s1[3]);
        char* ob = reinterpret_cast<char*>(p.out) + ((long)(t.n * H + t.y0) * W + t.x0) * (COUT * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = y[i][j];
                if (EPI == EPI_DEC) {
                    v = lrelu4(fma4(v, e2, e3), k02);
                    if (RES) v = add4(rr, v);
                }
                if (EPI == EPI_SYNTH) {      // t = nscale * noise (rounded), (v + t) + nbias, LeakyReLU
                    const f32x2 tlo = j == 0 ? pk_mul2_lo(e0.xy, nz[i]) : pk_mul2_hi(e0.xy, nz[i]);
                    const f32x2 thi = j == 0 ? pk_mul2_lo(e0.zw, nz[i]) : pk_mul2_hi(e0.zw, nz[i]);
                    v = lrelu4(add4(add4(v, f32x4{tlo.x, tlo.y, thi.x, thi.y}), e1), k02);
                    y[i][j] = v;
                }
                *reinterpret_cast<f32x4*>(ob + (long)i * W * (COUT * 4) + out_off + j * (COUT * 4)) = v;
            }
        if (EPI == EPI_SYNTH) {
            // Statistics per aligned x-quad and channel: s = (v0 + v1) + (v2 + v3), q = (v0 v0 + v1 v1) + (v2 v2 + v3 v3).  The quad of tile
            // row i spans this lane's tile (x = 2 wtx, 2 wtx + 1) and its x-neighbour's (lane ^ 1): each lane forms its own pair sums, the
            // pair exchanges them by DPP, the even lane finishes row 0 and the odd lane row 1 -- a + b = b + a bit for bit, so which lane
            // adds does not matter; every quad is counted once.
            f32x4 ps[2], pq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ps[i] = add4(y[i][0], y[i][1]);
                const f32x2 a0 = pk_mul2(y[i][0].xy, y[i][0].xy), a1 = pk_mul2(y[i][0].zw, y[i][0].zw);
                const f32x2 b0 = pk_mul2(y[i][1].xy, y[i][1].xy), b1 = pk_mul2(y[i][1].zw, y[i][1].zw);
                pq[i] = add4(f32x4{a0.x, a0.y, a1.x, a1.y}, f32x4{b0.x, b0.y, b1.x, b1.y});
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float keep_s = odd ? ps[1][c] : ps[0][c], send_s = odd ? ps[0][c] : ps[1][c];
                const float keep_q = odd ? pq[1][c] : pq[0][c], send_q = odd ? pq[0][c] : pq[1][c];
                const float sq = keep_s + dpp_quad<0xB1>(send_s);
                const float qq = keep_q + dpp_quad<0xB1>(send_q);
                dI1[c] += to_fixed(sq, kStatScale1);
                dI2[c] += to_fixed_sq(qq, s2);
            }
        }
    };

    // ---- pipeline over the items (tile, block): item `it` is multiplied out of LDS buffer it & 1, item it+1 sits in ra, item it+2 is being loaded
    Tile tc, tr;
    {
        const int tx = w_begin % p.tiles_x, r = w_begin / p.tiles_x;
        tc.x0 = tx * 16; tc.y0 = (r % p.tiles_y) * 16; tc.n = r / p.tiles_y;
    }
    int ec = edge_code(tc), er, cbc = 0, cbr = 0;
    auto next_item = [&](Tile& t, int& e, int& cb) {
        if (NB == 1 || ++cb == NB) { cb = 0; advance(t); e = edge_code(t); }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, PF - 1>;
    // diagnostic build (make stamp) only: wave-cycle sums per phase -> ConvParams::stamps[6..10] (write, loads, transform + MFMA, epilogue, barrier)
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, sw = 0, sl = 0, sm = 0, se = 0, sb = 0;
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)k5; (void)sw; (void)sl; (void)sm; (void)se; (void)sb;
    load_item(S0{}, tc, ec, 0);
    write_item(S0{}, tc, ec, 0, 0);
    tr = tc; er = ec;
    if constexpr (PF == 2) {
        // ---- two prefetch sets: item `it` in LDS buffer it & 1, items it+1 and it+2 in the register sets (it+1) & 1 and it & 1, item it+3 being loaded
        Tile t2 = tc, t3; int e2c = ec, cb2 = 0, e3c, cb3;
        if (items > 1) { next_item(tr, er, cbr); load_item(S1{}, tr, er, cbr); t2 = tr; e2c = er; cb2 = cbr; }
        if (items > 2) { next_item(t2, e2c, cb2); load_item(S0{}, t2, e2c, cb2); }
        __syncthreads();
        auto body = [&](auto set_tag, int it) {      // set_tag: the register set of item it+1 (and of item it+3)
            const bool has_next = it + 1 < items;
            TICK(k0);
            if (has_next) write_item(set_tag, tr, er, cbr, (it + 1) & 1);
            TICK(k1);
            if (cbc == NB - 1) epilogue_loads(tc);
            t3 = t2; e3c = e2c; cb3 = cb2;
            if (it + 3 < items) { next_item(t3, e3c, cb3); load_item(set_tag, t3, e3c, cb3); }
            TICK(k2);
            if (NB == 1 || cbc == 0) multiply(std::true_type{}, it & 1, 0);
            else multiply(std::false_type{}, it & 1, cbc);
            TICK(k3);
            if (cbc == NB - 1) {
                epilogue(tc);
                if (EPI == EPI_SYNTH && (!has_next || tr.n != tc.n)) flush_stats(tc.n);
            }
            TICK(k4);
            __syncthreads();
            TICK(k5);
            TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4); TSUM(sb, k4, k5);
            tc = tr; ec = er; cbc = cbr; tr = t2; er = e2c; cbr = cb2; t2 = t3; e2c = e3c; cb2 = cb3;
        };
        for (int it = 0; it < items; it += 2) {
            body(S1{}, it);
            if (it + 1 < items) body(S0{}, it + 1);
        }
        TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
        TFLUSH(12, (unsigned long long)items); TFLUSH(15, 1ull);
        return;
    }
    if (items > 1) { next_item(tr, er, cbr); load_item(S0{}, tr, er, cbr); }
    __syncthreads();
    if constexpr (DB) {
        for (int it = 0; it < items; ++it) {
            const bool has_next = it + 1 < items;
            TICK(k0);
            if (has_next) write_item(S0{}, tr, er, cbr, (it + 1) & 1);
            TICK(k1);
            if (cbc == NB - 1) epilogue_loads(tc);
            Tile t2 = tr; int e2c = er, cb2 = cbr;
            if (it + 2 < items) { next_item(t2, e2c, cb2); load_item(S0{}, t2, e2c, cb2); }
            TICK(k2);
            if (NB == 1 || cbc == 0) multiply(std::true_type{}, it & 1, 0);
            else multiply(std::false_type{}, it & 1, cbc);
            TICK(k3);
            if constexpr (RGB) torgb(tc, it & 1);
            if (cbc == NB - 1) {
                epilogue(tc);
                if (EPI == EPI_SYNTH && (!has_next || tr.n != tc.n)) flush_stats(tc.n);
            }
            TICK(k4);
            __syncthreads();
            TICK(k5);
            TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4); TSUM(sb, k4, k5);
            tc = tr; ec = er; cbc = cbr; tr = t2; er = e2c; cbr = cb2;
        }
        TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
        TFLUSH(12, (unsigned long long)items); TFLUSH(15, 1ull);
    } else {
        for (int it = 0; it < items; ++it) {
            const bool has_next = it + 1 < items;
            transform(0);                       // the patch of item `it` out of the one buffer, into registers
            __syncthreads();                    // every wave has read it
            if (has_next) write_item(S0{}, tr, er, 0, 0);
            epilogue_loads(tc);
            Tile t2 = tr; int e2c = er, cb2 = 0;
            if (it + 2 < items) { next_item(t2, e2c, cb2); load_item(S0{}, t2, e2c, cb2); }
            mfmas(std::true_type{}, 0);
            epilogue(tc);
            if (EPI == EPI_SYNTH && (!has_next || tr.n != tc.n)) flush_stats(tc.n);
            __syncthreads();                    // item it + 1 is complete in the buffer
            tc = tr; ec = er; tr = t2; er = e2c;
        }
    }
}

// ---- anti-phase start (experiment) ----------------------------------------------------------------------------------------------------------
// Two workgroups share a CU; they start together, do the same work and therefore stay IN PHASE: both stage while the matrix pipe idles, both
// multiply while it is shared (the stamps' MFMA column is twice the MFMAs' own time; the timing-only decomposition of tools/dbg_stream.sh shows
// the phases adding up instead of overlapping).  Nothing re-phases them either -- an offset given once persists.  So every second workgroup
// to arrive on a CU (a ticket from a per-CU counter, CU = XCC / SE / SH / CU of HW_ID) sleeps `cycles64` x 64 cycles before its first item.
__device__ unsigned g_cu_tickets[4096];
__device__ __forceinline__ void antiphase_start(int cycles64, float* lds_word) {
    if (cycles64 <= 0) return;
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned cu = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
        *reinterpret_cast<unsigned*>(lds_word) = atomicAdd(&g_cu_tickets[cu & 4095u], 1u);
    }
    __syncthreads();
    const unsigned ticket = *reinterpret_cast<volatile unsigned*>(lds_word);
    __syncthreads();
    if (ticket & 1u)
        for (int i = 0; i < cycles64; i += 16) __builtin_amdgcn_s_sleep(16);
}

// ---- the streamed-weight layers (>= 64 input channels: g.16 ... g.256.conv_2, d.cvt_4 ... 6 of the FFHQ path) in the same lean form ----------
// One workgroup = 4 waves on 16x16 tiles x ONE 16-channel output group (blockIdx.y), an item = (tile, 16-channel block).  As conv3x3_wino<EPI, 1,
// true, AFF, 1>: double-buffered image and weight block, activations prefetched two items ahead through registers, accumulators across the
// blocks of a tile, direct statistics.  What is leaner: the item's 16 KB weight block goes to LDS by LDS-DMA (global_load_lds_dwordx4: four
// 1 KB pieces per wave, issued one item ahead into the buffer the previous item has left, waited for with a counted vmcnt before the item's
// closing barrier) -- no staging registers, no LDS stores, and room for the accumulators beside the prefetch; the sample's AdaIN coefficients sit in
// a 4 KB LDS table as (A x 16 | B x 16) per block -- two LDS reads per item instead of four global loads of (mean, A, B, -) entries -- and are
// applied with packed fma; wave-uniform global bases; border flags in one word; weights as the A operand, so the epilogue of the tile's last
// block is the transposition-free packed one of conv3x3_wino_lean; the chains of a tile's first block start from the inline constant 0.
// NT = 2: the workgroup owns TWO output groups (blockIdx.y = group pair): one staged image, one input transform, 128 MFMAs per item and wave.
// 128 accumulator registers: one wave per SIMD with the whole register file (LDS: 106 KB, one workgroup per CU).
// IL: the item's vector-memory instructions (the next item's weight DMA, the loads of the item after it) are issued INSIDE the MFMA stream
// of the current item instead of in phases of their own.  The stamps of the phased form charge 1.3-2.0 k cycles per item to those phases
// although nothing in them waits for data: every wave of the CU issues its 1 KB instructions at the same moment and the texture-address
// path takes them at 64 B per clock (an item's 37-53 KB are 600-850 cycles per CU) -- time in which no MFMA is issued.  Issued between the
// MFMAs of frequency rows 0 and 2 that time is the matrix pipe's; the weight reads of a row are issued one row ahead by hand, because the
// scheduling barriers that pin the memory instructions also stop the compiler from hoisting them.
// (A further form, IL = 2 -- the item's barrier moved between the staging stores and the MFMAs so that the next item's patch reads could be
// issued in front of the MFMAs and transformed behind them -- was bit-exact and 12 % slower without spills, 55 % slower with them: DESIGN.md 5.)
template <int EPI, bool AFF, int NT = 1, int IL = 0>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void conv3x3_wino_stream(ConvParams p) {
    constexpr int NCH = 6;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sA = smem;                       // [2][IMG]
    float* const sB = smem + 2 * IMG;             // [2][NT][SEG]
    float* const sE = sB + 2 * NT * SEG;          // [NT][4][16]: nscale | nbias | bn_s | bn_beta of the group's 16 output channels
    float* const sC = sE + 64 * NT;               // [nblk][32]: A of the block's 16 channels, then B (the sample being staged)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4, part = tid & 3;
    const int H = p.H, W = p.W, CIN = p.C0, COUT = p.Cout, nblk = p.C0 >> 4;
    const int g = blockIdx.y;
    const int per = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int w_begin = xcd_block(blockIdx.x, gridDim.x) * per;
    const int w_end = min(p.total_tiles, w_begin + per);
    if (w_begin >= w_end) return;
    const int items = (w_end - w_begin) * nblk;
#if GSA_EXPERIMENTS
    antiphase_start(p.group_minor, sC);      // experiments build: group_minor carries GSA_STAGGER (64-cycle units; 0 = off)
#endif

    unsigned s_off[NCH], eflags = 0;
    int l_off[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int pix = (tid + 256 * k) >> 2;
        const bool real = pix < NPIX;
        const int ly = real ? pix / LW : 1, lx = real ? pix % LW : 1;
        s_off[k] = (unsigned)(((ly * W + lx) * CIN + part * 4) * 4);
        l_off[k] = real ? ly * RS + lx * 16 + part * 4 : -1;
        eflags |= (unsigned)((ly == 0 ? 1 : 0) | (ly == LH - 1 ? 2 : 0) | (lx == 0 ? 4 : 0) | (lx == LW - 1 ? 8 : 0)) << (4 * k);
    }
    const unsigned safe_off = (unsigned)((((W + 1) * CIN) + part * 4) * 4);
    auto edge_code = [&](const Tile& t) { return (t.y0 == 0 ? 1 : 0) | (t.y0 + 16 == H ? 2 : 0) | (t.x0 == 0 ? 4 : 0) | (t.x0 + 16 == W ? 8 : 0); };
    auto advance = [&](Tile& t) { t.x0 += 16; if (t.x0 == W) { t.x0 = 0; t.y0 += 16; if (t.y0 == H) { t.y0 = 0; t.n += 1; } } };

    f32x4 ra[NCH];                                // the item in flight: this thread's activation chunks
    int dbg_items = 0; (void)dbg_items;           // diagnostic build: the first two items always run in full (valid data in both LDS buffers)
    const float* const wgrp = p.wpk + (size_t)g * NT * nblk * SEG + lane * 4;
    // the item's weight block(s): pieces wave, wave + 4, wave + 8, wave + 12 of 1 KB each per group, straight into the LDS panel
    auto dma_weights = [&](int cb, int buf) {
#ifdef GSA_DBG_HOOKS
        if (p.dbg & 16) cb = 0;      // diagnostic build only (WRONG results): the weight stream stays in L2
        if ((p.dbg & 1024) && dbg_items > 2) return;      // ... no weight DMA
#endif
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const float* wb = wgrp + ((size_t)ct * nblk + cb) * SEG;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int piece = wave + 4 * j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + piece * 256),
                                                 (__attribute__((address_space(3))) void*)(sB + (buf * NT + ct) * SEG + piece * 256), 16, 0, 0);
            }
        }
    };
    auto load_item = [&](const Tile& t, int e, int cb) {
        const char* hb = reinterpret_cast<const char*>(p.src0) + ((((long)(t.n * H + t.y0) * W + t.x0) - (W + 1)) * CIN + cb * 16) * 4;
#ifdef GSA_DBG_HOOKS
        if (p.dbg & 32) hb = reinterpret_cast<const char*>(p.src0) + ((long)(W + 1) * CIN) * 4;      // diagnostic build only (WRONG results): one input tile, block 0
        if ((p.dbg & 512) && dbg_items > 2) return;       // ... no activation loads (the registers keep an earlier item)
#endif
        if (e) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const unsigned bad = (eflags >> (4 * k)) & (unsigned)e;
                const unsigned off = s_off[k] + (bad ? safe_off - s_off[k] : 0u);
                ra[k] = *reinterpret_cast<const f32x4*>(hb + off);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) ra[k] = *reinterpret_cast<const f32x4*>(hb + s_off[k]);
        }
    };
    int n_coef = -1;
    auto coefficients = [&](int n) {              // wave-uniform and rare: the sample changes (first item; a range spans few samples)
        __syncthreads();                          // nobody still reads the old table
        for (int e = tid; e < CIN; e += 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.aff0 + (size_t)n * CIN + e);      // (mean, A, B, -)
            sC[(e >> 4) * 32 + (e & 15)] = a[1];
            sC[(e >> 4) * 32 + 16 + (e & 15)] = a[2];
        }
        __syncthreads();
    };
    auto write_item = [&](const Tile& t, int e, int cb, int buf) {
        float* img = sA + buf * IMG;
        if (AFF && t.n != n_coef) { coefficients(t.n); n_coef = t.n; }
#ifdef GSA_DBG_HOOKS
        if (++dbg_items > 2 && (p.dbg & 128)) return;      // diagnostic build only (WRONG results): no staging (AdaIN + LDS stores)
#endif
        f32x4 kA = {0.f, 0.f, 0.f, 0.f}, kB = kA;
        if (AFF) {
            kA = *reinterpret_cast<const f32x4*>(sC + cb * 32 + part * 4);
            kB = *reinterpret_cast<const f32x4*>(sC + cb * 32 + 16 + part * 4);
        }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (e) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const f32x4 v = AFF ? fma4(ra[k], kA, kB) : ra[k];
                const bool bad = ((eflags >> (4 * k)) & (unsigned)e) != 0;
                if (k < NCH - 1 || l_off[NCH - 1] >= 0) *reinterpret_cast<f32x4*>(img + l_off[k]) = bad ? z : v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (k < NCH - 1 || l_off[NCH - 1] >= 0) *reinterpret_cast<f32x4*>(img + l_off[k]) = AFF ? fma4(ra[k], kA, kB) : ra[k];
        }
    };

    const int wty = ((i16 >> 3) & 1) * 2 + ((i16 >> 1) & 1), wtx = ((i16 >> 2) & 1) * 2 + (i16 & 1);
    const int qy = wave >> 1, qx = wave & 1;
    const int pbase = (qy * 8 + 2 * wty) * RS + (qx * 8 + 2 * wtx) * 16 + kq * 4;
    const int bbase = (kq * 16 + i16) * 4;
    const int co4 = g * NT * 16 + kq * 4;
    const unsigned out_off = (unsigned)((((qy * 8 + 2 * wty) * W + qx * 8 + 2 * wtx) * COUT + co4) * 4);
    if (tid < 16 * NT) {      // the groups' per-channel epilogue constants: read from LDS in the epilogue (one per tile), not held in registers
        float* e = sE + (tid >> 4) * 64 + (tid & 15);
        const int co = g * NT * 16 + tid;
        e[0] = EPI == EPI_SYNTH ? p.nscale[co] : 0.f;
        e[16] = EPI == EPI_SYNTH ? p.nbias[co] : 0.f;
        e[32] = EPI == EPI_DEC ? p.bn_s[co] : 0.f;
        e[48] = EPI == EPI_DEC ? p.bn_beta[co] : 0.f;
    }
    const unsigned nz_off = (unsigned)((((qy * 8 + 2 * wty) * W + qx * 8 + 2 * wtx)) * 4);
    f32x2 nz[2] = {{0.f, 0.f}, {0.f, 0.f}};
    const int s2 = stat_s2(H * W);
    const bool odd = (lane & 1) != 0;
    unsigned long long dI1[NT][4], dI2[NT][4];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int c = 0; c < 4; ++c) dI1[ct][c] = dI2[ct][c] = 0ull;
    auto flush_stats = [&](int n) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned long long I1 = dI1[ct][c], I2 = dI2[ct][c];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { I1 += shfl_xor_u64(I1, m); I2 += shfl_xor_u64(I2, m); }
                if (i16 == 0) {
                    StatPart* a = p.partials + ((size_t)n * p.prow + (blockIdx.x & (kDirectRows - 1))) * COUT + co4 + ct * 16 + c;
                    atomicAdd(&a->s1, I1);
                    atomicAdd(&a->s2, I2);
                }
                dI1[ct][c] = dI2[ct][c] = 0ull;
            }
    };
    const f32x2 k02 = {0.2f, 0.2f};
    auto epilogue_loads = [&](const Tile& t) {
        if (EPI == EPI_SYNTH) {
            const char* nb = reinterpret_cast<const char*>(p.noise) + ((long)(t.n * H + t.y0) * W + t.x0) * 4;
            nz[0] = *reinterpret_cast<const f32x2*>(nb + nz_off);
            nz[1] = *reinterpret_cast<const f32x2*>(nb + (long)W * 4 + nz_off);
        }
    };

    f32x4 accs[NT][16];
    auto multiply = [&](auto first_tag, int buf, auto&& vm0, auto&& vm1) {
        constexpr bool FIRST = decltype(first_tag)::value;
#ifdef GSA_DBG_HOOKS
        if ((p.dbg & 256) && dbg_items > 2) {      // diagnostic build only (WRONG results): no patch reads, transform, weight reads, MFMAs
            if (FIRST) for (int ct = 0; ct < NT; ++ct) for (int f = 0; f < 16; ++f) accs[ct][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
#endif
        const float* a_img = sA + buf * IMG + pbase;
        const float* b_img = sB + buf * NT * SEG + bbase;
        f32x4 V[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(a_img + 0 * RS + c * 16);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(a_img + 1 * RS + c * 16);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(a_img + 2 * RS + c * 16);
            const f32x4 d3 = *reinterpret_cast<const f32x4*>(a_img + 3 * RS + c * 16);
            V[0 * 4 + c] = sub4(d0, d2);
            V[1 * 4 + c] = add4(d1, d2);
            V[2 * 4 + c] = sub4(d2, d1);
            V[3 * 4 + c] = sub4(d1, d3);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 t0 = V[i * 4 + 0], t1 = V[i * 4 + 1], t2 = V[i * 4 + 2], t3 = V[i * 4 + 3];
            V[i * 4 + 0] = sub4(t0, t2);
            V[i * 4 + 1] = add4(t1, t2);
            V[i * 4 + 2] = sub4(t2, t1);
            V[i * 4 + 3] = sub4(t1, t3);
            valu_settle(V[i * 4 + 0], V[i * 4 + 1], V[i * 4 + 2], V[i * 4 + 3]);
        }
        if constexpr (IL == 1) {
            f32x4 u[2][NT][4];
            auto read_row = [&](int fb) {
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[fb & 1][ct][j] = *reinterpret_cast<const f32x4*>(b_img + ct * SEG + (fb * 4 + j) * 256);
            };
            read_row(0);
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                if (fb < 3) read_row(fb + 1);
                if (fb == 0) vm0();
                if (fb == 2) vm1();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) {
                            const int f = fb * 4 + j;
                            accs[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[fb & 1][ct][j][cg], V[f][cg], FIRST && cg == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : accs[ct][f], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            return;
        }
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            f32x4 u[NT][4];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) u[ct][j] = *reinterpret_cast<const f32x4*>(b_img + ct * SEG + (fb * 4 + j) * 256);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) {
                        const int f = fb * 4 + j;
                        accs[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[ct][j][cg], V[f][cg], FIRST && cg == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : accs[ct][f], 0, 0, 0);
                    }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue_ct = [&](const Tile& t, auto ct_tag) {
        constexpr int ct = decltype(ct_tag)::value;
        f32x4 (&acc)[16] = accs[ct];
        mfma_settle(acc);
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = add4(add4(acc[j], acc[4 + j]), acc[8 + j]);
            s1[j] = sub4(sub4(acc[4 + j], acc[8 + j]), acc[12 + j]);
        }
        f32x4 y[2][2];
        y[0][0] = add4(add4(s0[0], s0[1]), s0[2]); y[0][1] = sub4(sub4(s0[1], s0[2]), s0[3]);
        y[1][0] = add4(add4(s1[0], s1[1]), s1[2]); y[1][1] = sub4(sub4(s1[1], s1[2]), s1[3]);
        char* ob = reinterpret_cast<char*>(p.out) + (((long)(t.n * H + t.y0) * W + t.x0) * COUT + ct * 16) * 4;
        const float* sEc = sE + ct * 64 + kq * 4;
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(sEc), e1 = *reinterpret_cast<const f32x4*>(sEc + 16);
        const f32x4 e2 = *reinterpret_cast<const f32x4*>(sEc + 32), e3 = *reinterpret_cast<const f32x4*>(sEc + 48);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = y[i][j];
                if (EPI == EPI_DEC) v = lrelu4(fma4(v, e2, e3), k02);
                if (EPI == EPI_SYNTH) {
                    const f32x2 tlo = j == 0 ? pk_mul2_lo(e0.xy, nz[i]) : pk_mul2_hi(e0.xy, nz[i]);
                    const f32x2 thi = j == 0 ? pk_mul2_lo(e0.zw, nz[i]) : pk_mul2_hi(e0.zw, nz[i]);
                    v = lrelu4(add4(add4(v, f32x4{tlo.x, tlo.y, thi.x, thi.y}), e1), k02);
                    y[i][j] = v;
                }
#ifdef GSA_DBG_HOOKS
                if (!(p.dbg & 64))        // diagnostic build only (WRONG results): no output stores
#endif
                *reinterpret_cast<f32x4*>(ob + ((long)i * W + j) * COUT * 4 + out_off) = v;
            }
        if (EPI == EPI_SYNTH) {      // statistics: see conv3x3_wino_lean
            f32x4 ps[2], pq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ps[i] = add4(y[i][0], y[i][1]);
                const f32x2 a0 = pk_mul2(y[i][0].xy, y[i][0].xy), a1 = pk_mul2(y[i][0].zw, y[i][0].zw);
                const f32x2 b0 = pk_mul2(y[i][1].xy, y[i][1].xy), b1 = pk_mul2(y[i][1].zw, y[i][1].zw);
                pq[i] = add4(f32x4{a0.x, a0.y, a1.x, a1.y}, f32x4{b0.x, b0.y, b1.x, b1.y});
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float keep_s = odd ? ps[1][c] : ps[0][c], send_s = odd ? ps[0][c] : ps[1][c];
                const float keep_q = odd ? pq[1][c] : pq[0][c], send_q = odd ? pq[0][c] : pq[1][c];
                dI1[ct][c] += to_fixed(keep_s + dpp_quad<0xB1>(send_s), kStatScale1);
                dI2[ct][c] += to_fixed_sq(keep_q + dpp_quad<0xB1>(send_q), s2);
            }
        }
    };
    auto epilogue = [&](const Tile& t) {
        epilogue_ct(t, std::integral_constant<int, 0>{});
        if constexpr (NT == 2) epilogue_ct(t, std::integral_constant<int, 1>{});
    };

    Tile tc, tr;
    {
        const int tx = w_begin % p.tiles_x, r = w_begin / p.tiles_x;
        tc.x0 = tx * 16; tc.y0 = (r % p.tiles_y) * 16; tc.n = r / p.tiles_y;
    }
    int ec = edge_code(tc), er, cbc = 0, cbr = 0;
    auto next_item = [&](Tile& t, int& e, int& cb) {
        if (++cb == nblk) { cb = 0; advance(t); e = edge_code(t); }
    };
    dma_weights(0, 0);
    load_item(tc, ec, 0);
    write_item(tc, ec, 0, 0);
    tr = tc; er = ec;
    if (items > 1) {
        next_item(tr, er, cbr);
        load_item(tr, er, cbr);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // the four DMA pieces are older than the six loads of the second item
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, sw = 0, sl = 0, sm = 0, se = 0, sb = 0;      // diagnostic build only (make stamp)
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)k5; (void)sw; (void)sl; (void)sm; (void)se; (void)sb;
    for (int it = 0; it < items; ++it) {
        const bool has_next = it + 1 < items;
        TICK(k0);
        if (has_next) {
            if (!IL) dma_weights(cbr, (it + 1) & 1);      // the panel buffer item it - 1 used: every wave is past that item's closing barrier
            write_item(tr, er, cbr, (it + 1) & 1);
        }
        TICK(k1);
        if (cbc == nblk - 1) epilogue_loads(tc);
        Tile t2 = tr; int e2c = er, cb2 = cbr;
        if (it + 2 < items) { next_item(t2, e2c, cb2); if (!IL) load_item(t2, e2c, cb2); }
        TICK(k2);
        auto vm0 = [&]() { if (IL == 1 && has_next) dma_weights(cbr, (it + 1) & 1); };      // same order as the phased form: the DMA pieces are the older operations
        auto vm1 = [&]() { if (IL == 1 && it + 2 < items) load_item(t2, e2c, cb2); };
        if (cbc == 0) multiply(std::true_type{}, it & 1, vm0, vm1);
        else multiply(std::false_type{}, it & 1, vm0, vm1);
        TICK(k3);
        if (cbc == nblk - 1) {
            epilogue(tc);
            if (EPI == EPI_SYNTH && (!has_next || tr.n != tc.n)) flush_stats(tc.n);
        }
        TICK(k4);
        // this wave's DMA pieces have landed before anybody passes the barrier: they are older than the six loads of item it + 2 (in-order
        // completion), so at most six outstanding operations means the pieces are in LDS and the younger loads stay in flight; an
        // iteration that issued no such loads (the last two) waits for everything
        if (it + 2 < items) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        TICK(k5);
        TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4); TSUM(sb, k4, k5);
        tc = tr; ec = er; cbc = cbr; tr = t2; er = e2c; cbr = cb2;
    }
    TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
    TFLUSH(12, (unsigned long long)items); TFLUSH(15, 1ull);
}

constexpr int kMaxDev = 64;
struct LeanState { bool attr_done = false; int cus = 0; };
static std::mutex g_mu;

template <int EPI, bool AFF, bool RES, int NB, int GW, bool DB = true, int PF = 1, bool RGB = false>
hipError_t launch_t(const ConvParams& p, int n, hipStream_t s) {
    static LeanState st[kMaxDev];
    auto kern = conv3x3_wino_lean<EPI, AFF, RES, NB, GW, DB, PF, RGB>;
    const size_t lds = sizeof(float) * ((DB ? 2 : 1) * IMG + GW * NB * SEG);
    if (p.device < 0 || p.device >= kMaxDev) return hipErrorInvalidDevice;
    int cus;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        LeanState& d = st[p.device];
        if (!d.attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, p.device) != hipSuccess) d.cus = 256;
            d.attr_done = true;
        }
        cus = d.cus;
    }
    ConvParams q = p;
    q.wpk = p.wino;
    q.tiles_x = p.W / 16;
    q.tiles_y = p.H / 16;
    q.groups = 1;
    q.total_tiles = q.tiles_x * q.tiles_y * n;
    if (EPI == EPI_SYNTH) {      // direct statistics: the sums go to kDirectRows zeroed rows per sample (finalize_kernel clears what it read)
        q.stats_direct = 1;
        q.prow = kDirectRows;
        if (p.stat_rows_host) *p.stat_rows_host = q.prow;
    }
    // persistent workgroups: two 4-wave workgroups per CU (58 KB of LDS each) or one 8-wave workgroup (106 KB), each a contiguous tile range
    const int gx = std::min(q.total_tiles, cus * (GW == 1 ? (DB ? 2 : 3) : 1));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256 * GW), lds, s, q);
    return hipGetLastError();
}

// GSA_WINO_LEAN_SB=1: the single-buffered three-workgroups-per-CU form for the 16 -> 16 decoder layers (A/B; same bits)
bool single_buffered() {
    static const bool on = getenv("GSA_WINO_LEAN_SB") && atoi(getenv("GSA_WINO_LEAN_SB")) != 0;
    return on;
}

// GSA_WINO_LEAN_PF=1 | 2: register sets of the activation prefetch of the 16 -> 16 kernels (speed only, same bits).  Measured equal (same box,
// FFHQ batch 8: d.cvt_8 0.228 / 0.232 ms, d.main_7.b 0.260 / 0.257, g.1024.conv_2 0.257 / 0.259 with one / two sets): the wait in front of
// the staging writes is the memory system's throughput (1.1-1.2 GB of traffic in 0.23 ms), not the distance between a load and its use.
// Default: one set (24 registers fewer).
int prefetch_sets() {
    static const int pf = getenv("GSA_WINO_LEAN_PF") ? atoi(getenv("GSA_WINO_LEAN_PF")) : 1;
    return pf == 2 ? 2 : 1;
}

template <int NB, int GW>
hipError_t launch_shape(const ConvParams& p, int epi, int n, hipStream_t s) {
    if constexpr (NB == 1 && GW == 1) {
        if (epi == EPI_DEC && single_buffered()) {
            if (p.resid) return p.aff0 ? launch_t<EPI_DEC, true, true, 1, 1, false>(p, n, s) : launch_t<EPI_DEC, false, true, 1, 1, false>(p, n, s);
            return p.aff0 ? launch_t<EPI_DEC, true, false, 1, 1, false>(p, n, s) : launch_t<EPI_DEC, false, false, 1, 1, false>(p, n, s);
        }
    }
    if constexpr (NB == 1 && GW == 1) {
        if (p.rgb_img != nullptr) {      // the caller checked wino_lean_fuses_torgb
            if (epi != EPI_DEC || !p.aff0 || p.resid) return hipErrorInvalidValue;
            return launch_t<EPI_DEC, true, false, 1, 1, true, 1, true>(p, n, s);
        }
        if (prefetch_sets() == 2) {      // the HBM-bound 16 -> 16 layers: loads two items ahead of their use
            if (epi == EPI_DEC) {
                if (p.resid) return p.aff0 ? launch_t<EPI_DEC, true, true, 1, 1, true, 2>(p, n, s) : launch_t<EPI_DEC, false, true, 1, 1, true, 2>(p, n, s);
                return p.aff0 ? launch_t<EPI_DEC, true, false, 1, 1, true, 2>(p, n, s) : launch_t<EPI_DEC, false, false, 1, 1, true, 2>(p, n, s);
            }
            if (epi == EPI_SYNTH) return p.aff0 ? launch_t<EPI_SYNTH, true, false, 1, 1, true, 2>(p, n, s) : launch_t<EPI_SYNTH, false, false, 1, 1, true, 2>(p, n, s);
        }
    }
    if (epi == EPI_DEC) {
        if (p.resid) return p.aff0 ? launch_t<EPI_DEC, true, true, NB, GW>(p, n, s) : launch_t<EPI_DEC, false, true, NB, GW>(p, n, s);
        return p.aff0 ? launch_t<EPI_DEC, true, false, NB, GW>(p, n, s) : launch_t<EPI_DEC, false, false, NB, GW>(p, n, s);
    }
    if (epi == EPI_SYNTH) return p.aff0 ? launch_t<EPI_SYNTH, true, false, NB, GW>(p, n, s) : launch_t<EPI_SYNTH, false, false, NB, GW>(p, n, s);
    return hipErrorInvalidValue;
}

template <int EPI, bool AFF, int NT = 1, int IL = 0>
hipError_t launch_stream_t(const ConvParams& p, int n, hipStream_t s) {
    static LeanState st[kMaxDev];
    auto kern = conv3x3_wino_stream<EPI, AFF, NT, IL>;
    const int nblk = p.C0 / 16;
    const size_t lds = sizeof(float) * (2 * IMG + 2 * NT * SEG + 64 * NT + nblk * 32);
    if (p.device < 0 || p.device >= kMaxDev) return hipErrorInvalidDevice;
    int cus;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        LeanState& d = st[p.device];
        if (!d.attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, p.device) != hipSuccess) d.cus = 256;
            d.attr_done = true;
        }
        cus = d.cus;
    }
    ConvParams q = p;
    q.wpk = p.wino;
    q.tiles_x = p.W / 16;
    q.tiles_y = p.H / 16;
    q.groups = p.Cout / (16 * NT);                     // workgroup columns: a group (NT = 2: a pair of groups) each
#if GSA_EXPERIMENTS
    static const int stagger = getenv("GSA_STAGGER") ? atoi(getenv("GSA_STAGGER")) : 0;
    q.group_minor = stagger;
#else
    q.group_minor = 0;
#endif
    q.total_tiles = q.tiles_x * q.tiles_y * n;         // per output-channel group
    if (EPI == EPI_SYNTH) {
        q.stats_direct = 1;
        q.prow = kDirectRows;
        if (p.stat_rows_host) *p.stat_rows_host = q.prow;
    }
    // persistent workgroups, two per CU (NT = 2: one), each inside its channel group; workgroup (x, g) has the linear index x + g * gx: with gx a
    // multiple of 8 the groups of a tile range meet in one XCD's L2 (conv3x3_wino's launch shape)
    const int slots = std::max(1, cus * (NT == 1 ? 2 : 1) / q.groups);
    const int gx = std::min(q.total_tiles, slots);
    hipLaunchKernelGGL(kern, dim3(gx, q.groups), dim3(256), lds, s, q);
    return hipGetLastError();
}

}  // namespace lean
using namespace lean;

// the layers this file takes: Winograd form (conv_uses_wino), 16 -> 16 or 32 -> 32 channels, output >= 32 px, synthesis epilogue with direct
// statistics or decoder epilogue with the residual absent or one half-resolution tensor.  GSA_WINO_LEAN=0 keeps conv3x3_wino (same bits).
bool wino_lean_applies(const ConvParams& p, int epi) {
    static const int enabled = getenv("GSA_WINO_LEAN") ? atoi(getenv("GSA_WINO_LEAN")) : 7;      // bit 0: 16 -> 16, bit 1: 32 -> 32
    if (p.H != p.W || p.H % 16) return false;
    if ((enabled & 4) && p.C0 >= 64 && p.C0 <= 512 && p.resid == nullptr && p.H >= 16)      // bit 2: the streamed-weight layers
        return epi == EPI_SYNTH ? (p.partials != nullptr && p.noise != nullptr && p.fin_aff == nullptr) : epi == EPI_DEC;
    const bool shape = (p.C0 == 16 && p.Cout == 16 && (enabled & 1)) || (p.C0 == 32 && p.Cout == 32 && (enabled & 2));
    if (!shape || p.H < 32) return false;
    if (epi == EPI_SYNTH) return p.partials != nullptr && p.noise != nullptr && p.fin_aff == nullptr && p.resid == nullptr;
    if (epi != EPI_DEC) return false;
    if (p.resid != nullptr && (p.resid_up != 1 || p.resid1 != nullptr)) return false;
    return true;
}

// toRGB rides on the decoder's last cvt conv when that conv is the 16 -> 16 lean kernel reading an AdaIN source, 3 colours (GSA_FUSE_RGB=0: never)
bool wino_lean_fuses_torgb(const ConvParams& p, int epi, int nc) {
    static const bool enabled = !(getenv("GSA_FUSE_RGB") && atoi(getenv("GSA_FUSE_RGB")) == 0);
    return enabled && nc == 3 && epi == EPI_DEC && p.C0 == 16 && p.Cout == 16 && p.aff0 != nullptr && p.resid == nullptr && !p.bf16 &&
           wino_lean_applies(p, epi) && !single_buffered();
}

// GSA_WINO_NT2=1 (experiments build; 2: on every streamed layer): two output groups per workgroup in the streamed-weight kernel -- one wave per
// SIMD, 128 MFMAs per staged item -- where the layer still gives every CU a workgroup.  Same bits; measured EQUAL per layer (g.32 ... g.256.conv_2
// 0.224/0.221/0.218/0.228 -> 0.227/0.220/0.218/0.232 ms), with the memory instructions inside the MFMA stream (GSA_WINO_IL=1) -4 % on three of them.
bool stream_pairs(const ConvParams& p, int n) {
#if GSA_EXPERIMENTS
    static const int on = getenv("GSA_WINO_NT2") ? atoi(getenv("GSA_WINO_NT2")) : 0;
    if (!on || p.Cout % 32) return false;
    const long wgs = (long)(p.W / 16) * (p.H / 16) * n * (p.Cout / 32);
    return wgs >= (on == 2 ? 1 : 256);
#else
    (void)p; (void)n;
    return false;
#endif
}

// GSA_WINO_IL=1 (experiments build): the paired streamed-weight kernel issues its memory instructions inside the MFMA stream -- speed only, same bits
int stream_interleaved() {
#if GSA_EXPERIMENTS
    static const int il = getenv("GSA_WINO_IL") ? atoi(getenv("GSA_WINO_IL")) : 0;
    return il == 1 ? 1 : 0;
#else
    return 0;
#endif
}

const char* wino_lean_name(const ConvParams& p, int epi, int n) {
    static thread_local char buf[112];
    if (p.C0 >= 64) {
        const bool pairs = stream_pairs(p, n);
        const int il = pairs ? stream_interleaved() : 0;
        snprintf(buf, sizeof buf, "void gsa::lean::conv3x3_wino_stream<%d, %s, %d, %d>(gsa::ConvParams)", epi, p.aff0 ? "true" : "false", pairs ? 2 : 1, il);
        return buf;
    }
    const bool sb = p.C0 == 16 && epi == EPI_DEC && single_buffered();
    const bool rgbf = p.rgb_img != nullptr;
    snprintf(buf, sizeof buf, "void gsa::lean::conv3x3_wino_lean<%d, %s, %s, %d, %d, %s, %d, %s>(gsa::ConvParams)", epi, p.aff0 ? "true" : "false",
             p.resid ? "true" : "false", p.C0 / 16, p.Cout / 16, sb ? "false" : "true", p.C0 == 16 && !sb && !rgbf ? prefetch_sets() : 1, rgbf ? "true" : "false");
    return buf;
}

hipError_t launch_wino_lean(const ConvParams& p, int epi, int n, hipStream_t s) {
    if (p.C0 >= 64) {
#if GSA_EXPERIMENTS
        if (stream_pairs(p, n)) {
            const int il = stream_interleaved();
#define GSA_STREAM2(IL_) \
            if (il == IL_) { \
                if (epi == EPI_SYNTH) return p.aff0 ? launch_stream_t<EPI_SYNTH, true, 2, IL_>(p, n, s) : launch_stream_t<EPI_SYNTH, false, 2, IL_>(p, n, s); \
                if (epi == EPI_DEC) return p.aff0 ? launch_stream_t<EPI_DEC, true, 2, IL_>(p, n, s) : launch_stream_t<EPI_DEC, false, 2, IL_>(p, n, s); \
                return hipErrorInvalidValue; \
            }
            GSA_STREAM2(0) GSA_STREAM2(1)
#undef GSA_STREAM2
            return hipErrorInvalidValue;
        }
#endif
        if (epi == EPI_SYNTH) return p.aff0 ? launch_stream_t<EPI_SYNTH, true>(p, n, s) : launch_stream_t<EPI_SYNTH, false>(p, n, s);
        if (epi == EPI_DEC) return p.aff0 ? launch_stream_t<EPI_DEC, true>(p, n, s) : launch_stream_t<EPI_DEC, false>(p, n, s);
        return hipErrorInvalidValue;
    }
    if (p.C0 == 16 && p.Cout == 16) return launch_shape<1, 1>(p, epi, n, s);
    if (p.C0 == 32 && p.Cout == 32) return launch_shape<2, 2>(p, epi, n, s);
    return hipErrorInvalidValue;
}

}  // namespace gsa
