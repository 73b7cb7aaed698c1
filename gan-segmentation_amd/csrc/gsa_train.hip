// Decoder-training operators for gfx950 behind the C ABI include/gsa_train.h.
//
// Training the reference's decoder is a few annotated images at batch size 1 for 24 epochs
// (seg_solver.py:83-132).  The kernels work in the reference's own NCHW / OIHW layouts: one convolution kernel
// serves the forward pass and, with the weight read transposed and flipped, the input gradient; a second one
// the weight gradient.  Both run on the matrix cores (v_mfma_f32_16x16x4_f32) straight from global memory when
// W % 4 == 0 and H*W % 16 == 0 (every decoder layer); the LDS-tiled vector-ALU forms remain for other shapes.
// Nothing here is shared with the inference kernels, and nothing here is order-canonical.
//
//   conv_mfma_kernel<K> / conv_kernel<K>     nn.Conv2D 3x3 / 1x1 (+ concat, + nearest x2 on read) and its dgrad   networks_seg.py:14-41,68,91
//   wgrad_mfma_kernel<K> / wgrad_kernel<K>   dL/dW, dL/db of the same convolution                                  (autograd)
//   bn_*                nn.BatchNorm (training mode) + LeakyReLU(0.2) + Dropout mask          networks_seg.py:17-32,69-78
//   softmax_ce_kernel   SoftmaxCELoss(axis=1) with sample weights                             seg_solver.py:395-407
//   adam_kernel         mx.optimizer.Adam                                                     seg_solver.py:203-219
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>

#include "../../include/gsa_train.h"

namespace {

constexpr int GSA_OK_ = 0, GSA_ERR_INVALID_ = -1, GSA_ERR_HIP_ = -4;

#define TRY_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "gsa_train: %s -> %s\n", #expr, hipGetErrorString(e_)); return GSA_ERR_HIP_; } } while (0)

__device__ __forceinline__ float lrelu02(float v) { return v > 0.0f ? v : 0.2f * v; }

// ---- convolution (forward / input gradient) ---------------------------------------------------
struct ConvArgs {
    const float* x0; const float* x1; int C0, C1, Hs, Ws, up, H, W;
    const float* w; int Cout, transposed; const float* bias;
    float* out0; int Cout0; float* out1; int accumulate;
};

template <int K>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
    constexpr int P = K / 2, LT = 16 + 2 * P, CI_B = 8, CO_B = 16, KK = K * K;
    __shared__ float tile[CI_B][LT][LT + 1];
    __shared__ float wsm[CI_B][KK][CO_B];
    const int tiles_x = (a.W + 15) / 16;
    const int ty0 = (blockIdx.x / tiles_x) * 16, tx0 = (blockIdx.x % tiles_x) * 16;
    const int o0 = blockIdx.y * CO_B, n = blockIdx.z;
    const int tid = threadIdx.x, ly = tid / 16, lx = tid % 16;
    const int Cin = a.C0 + a.C1;
    float acc[CO_B];
#pragma unroll
    for (int o = 0; o < CO_B; ++o) acc[o] = 0.0f;
    for (int c0 = 0; c0 < Cin; c0 += CI_B) {
        for (int i = tid; i < CI_B * LT * LT; i += 256) {
            const int c = i / (LT * LT), r = i % (LT * LT), yy = r / LT, xx = r % LT;
            const int gy = ty0 - P + yy, gx = tx0 - P + xx, cc = c0 + c;
            float v = 0.0f;
            if (cc < Cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const bool first = cc < a.C0;
                const float* src = first ? a.x0 : a.x1;
                const int Cs = first ? a.C0 : a.C1, cs = first ? cc : cc - a.C0;
                v = src[(((size_t)n * Cs + cs) * a.Hs + (gy >> a.up)) * a.Ws + (gx >> a.up)];
            }
            tile[c][yy][xx] = v;
        }
        for (int i = tid; i < CI_B * KK * CO_B; i += 256) {
            const int o = i % CO_B, t = (i / CO_B) % KK, c = i / (CO_B * KK);
            const int oo = o0 + o, cc = c0 + c;
            float v = 0.0f;
            if (oo < a.Cout && cc < Cin)
                v = a.transposed ? a.w[((size_t)cc * a.Cout + oo) * KK + (KK - 1 - t)] : a.w[((size_t)oo * Cin + cc) * KK + t];
            wsm[c][t][o] = v;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CI_B; ++c)
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                const float v = tile[c][ly + t / K][lx + t % K];
#pragma unroll
                for (int o = 0; o < CO_B; ++o) acc[o] = fmaf(v, wsm[c][t][o], acc[o]);
            }
        __syncthreads();
    }
    const int y = ty0 + ly, x = tx0 + lx;
    if (y >= a.H || x >= a.W) return;
#pragma unroll
    for (int o = 0; o < CO_B; ++o) {
        const int oo = o0 + o;
        if (oo >= a.Cout) break;
        float v = acc[o] + (a.bias ? a.bias[oo] : 0.0f);
        float* dst = oo < a.Cout0 ? a.out0 + (((size_t)n * a.Cout0 + oo) * a.H + y) * a.W + x
                                  : a.out1 + (((size_t)n * (a.Cout - a.Cout0) + (oo - a.Cout0)) * a.H + y) * a.W + x;
        if (a.accumulate) v += *dst;
        *dst = v;
    }
}

// The same convolution on the matrix cores (v_mfma_f32_16x16x4_f32) straight from NCHW: M = 16 consecutive pixels of
// an output row, N = 16 output channels, K = input channels (4 per MFMA) x taps.  A lane (m = lane & 15, q = lane >> 4)
// feeds pixel x0+m of channel 4i+q (one coalesced dword load per tap; concat / nearest-x2 resolved in the address) and
// the weight of output channel o0+m from an LDS panel [tap][channel][16]; D comes out as 4 consecutive pixels of one
// output channel per lane -> one 16-byte store.  A block (4 waves = 64 pixels of a row) walks many row segments with the
// panel staged once (Cin <= 64) or per 64-channel slice.  W % 4 == 0 and H*W % 16 == 0 (chunks may span rows).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    constexpr int P = K / 2, KK = K * K, CB = 64;
    __shared__ float wl[KK][CB][16];
    const int Cin = a.C0 + a.C1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 15, q = lane >> 4;
    const int o0 = blockIdx.y * 16;
    const int HW = a.H * a.W, chunks = HW / 16, groups = (chunks + 3) / 4, units = gridDim.z * groups;
    const bool single = Cin <= CB;
    auto stage = [&](int cblk) {
        const int cend = min(CB, Cin - cblk);
        for (int i = tid; i < KK * cend * 16; i += 256) {
            const int o = i & 15, cl = (i >> 4) % cend, t = i / (16 * cend);
            const int oo = o0 + o, cc = cblk + cl;
            float v = 0.0f;
            if (oo < a.Cout)
                v = a.transposed ? a.w[((size_t)cc * a.Cout + oo) * KK + (KK - 1 - t)] : a.w[((size_t)oo * Cin + cc) * KK + t];
            wl[t][cl][o] = v;
        }
        for (int i = tid; i < KK * 3 * 16; i += 256) {     // zero the (up to 3) channels past cend that a 4-group may touch
            const int o = i & 15, cl = cend + (i >> 4) % 3, t = i / 48;
            if (cl < CB) wl[t][cl][o] = 0.0f;
        }
    };
    if (single) { stage(0); __syncthreads(); }
    // a unit = 64 consecutive pixels of a sample's H*W plane (rows are contiguous, so 16-pixel chunks may span rows when
    // W < 16): wave w takes chunk 4*group + w
    for (int u = blockIdx.x + gridDim.x * blockIdx.z; u < units; u += gridDim.x * gridDim.z) {   // gridDim.z = batch size
        const int n = u / groups, grp = u - n * groups;
        const int chunk = grp * 4 + wave;
        const bool active = chunk < chunks;
        const int L0 = (active ? chunk : 0) * 16;
        const int L = L0 + m, y = L / a.W, x = L - y * a.W;        // this lane's pixel as the A operand
        int off[KK];
        bool ok[KK];
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int xx = x + kx - P, yy = y + ky - P;
                ok[ky * K + kx] = xx >= 0 && xx < a.W && yy >= 0 && yy < a.H;
                off[ky * K + kx] = (min(max(yy, 0), a.H - 1) >> a.up) * a.Ws + (min(max(xx, 0), a.W - 1) >> a.up);
            }
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int cblk = 0; cblk < Cin; cblk += CB) {
            if (!single) { __syncthreads(); stage(cblk); __syncthreads(); }
            const int cend = min(CB, Cin - cblk);
#pragma unroll 2
            for (int cl = 0; cl < cend; cl += 4) {
                const int c = cblk + cl + q;
                const bool cok = c < Cin;
                const bool first = c < a.C0 || !cok;          // lanes past the last channel read channel 0 (valid address)
                const float* src = first ? a.x0 : a.x1;
                const int Cs = first ? a.C0 : a.C1, cs = !cok ? 0 : (first ? c : c - a.C0);
                const float* plane = src + ((size_t)n * Cs + cs) * a.Hs * a.Ws;
                float v[KK];
#pragma unroll
                for (int t = 0; t < KK; ++t) v[t] = plane[off[t]];          // all loads of the group in flight together
#pragma unroll
                for (int t = 0; t < KK; ++t)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((ok[t] && cok) ? v[t] : 0.0f, wl[t][cl + q][m], acc, 0, 0, 0);
            }
        }
        const int oo = o0 + m;                                  // D: column = output channel lane&15, rows = pixels 4q..4q+3
        if (active && oo < a.Cout) {
            const float b = a.bias ? a.bias[oo] : 0.0f;
            float* dst = (oo < a.Cout0 ? a.out0 + ((size_t)n * a.Cout0 + oo) * HW
                                       : a.out1 + ((size_t)n * (a.Cout - a.Cout0) + (oo - a.Cout0)) * HW) + L0 + 4 * q;
            float4 v = make_float4(acc[0] + b, acc[1] + b, acc[2] + b, acc[3] + b);
            if (a.accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(dst);
                v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
            }
            *reinterpret_cast<float4*>(dst) = v;
        }
    }
}

// ---- weight / bias gradient ---------------------------------------------------------------------
struct WgradArgs {
    const float* x0; const float* x1; int C0, C1, Hs, Ws, up, H, W;
    const float* dy; int Cout; float* dw; float* db; int n;
};

template <int K>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    constexpr int P = K / 2, LT = 16 + 2 * P, KK = K * K, OB = 16, CB = 16;
    __shared__ float dyt[OB][256];
    __shared__ float xt[CB][LT][LT + 1];
    const int tiles_x = (a.W + 15) / 16, tiles = tiles_x * ((a.H + 15) / 16);
    const int Cin = a.C0 + a.C1, ncb = (Cin + CB - 1) / CB;
    const int o0 = (blockIdx.y / ncb) * OB, c0 = (blockIdx.y % ncb) * CB;
    const int tid = threadIdx.x;
    const int o = tid / CB, c = tid % CB;
    float acc[KK], bsum = 0.0f;
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = 0.0f;
    // a block walks many (sample, tile) pairs and keeps its 16x16xKK partial sums in registers: one atomic per
    // weight and block at the end (one block per tile made 4096 blocks contend on the same 2304 addresses)
    for (int work = blockIdx.x; work < tiles * a.n; work += gridDim.x) {
        const int n = work / tiles, tile = work % tiles;
        const int ty0 = (tile / tiles_x) * 16, tx0 = (tile % tiles_x) * 16;
        __syncthreads();
        for (int i = tid; i < OB * 256; i += 256) {
            const int oo_l = i / 256, p = i % 256, y = ty0 + p / 16, x = tx0 + p % 16, oo = o0 + oo_l;
            dyt[oo_l][p] = (oo < a.Cout && y < a.H && x < a.W) ? a.dy[(((size_t)n * a.Cout + oo) * a.H + y) * a.W + x] : 0.0f;
        }
        for (int i = tid; i < CB * LT * LT; i += 256) {
            const int cl = i / (LT * LT), r = i % (LT * LT), yy = r / LT, xx = r % LT;
            const int gy = ty0 - P + yy, gx = tx0 - P + xx, cc = c0 + cl;
            float v = 0.0f;
            if (cc < Cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const bool first = cc < a.C0;
                const float* src = first ? a.x0 : a.x1;
                const int Cs = first ? a.C0 : a.C1, cs = first ? cc : cc - a.C0;
                v = src[(((size_t)n * Cs + cs) * a.Hs + (gy >> a.up)) * a.Ws + (gx >> a.up)];
            }
            xt[cl][yy][xx] = v;
        }
        __syncthreads();
        // four consecutive pixels at a time: a row segment of 4+K-1 inputs serves all K horizontal taps (0.6 LDS
        // reads per FMA instead of 1.1)
        for (int py = 0; py < 16; ++py)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = dyt[o][py * 16 + q * 4 + j];
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    float xr[4 + K - 1];
#pragma unroll
                    for (int j = 0; j < 4 + K - 1; ++j) xr[j] = xt[c][py + ky][q * 4 + j];
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ky * K + kx] = fmaf(g[j], xr[j + kx], acc[ky * K + kx]);
                }
            }
        if (a.db && c0 == 0)      // bias gradient: the 16 c-lanes of an output channel split the pixels
            for (int p = c; p < 256; p += CB) bsum += dyt[o][p];
    }
    if (o0 + o < a.Cout && c0 + c < Cin) {
#pragma unroll
        for (int t = 0; t < KK; ++t) atomicAdd(&a.dw[((size_t)(o0 + o) * Cin + c0 + c) * KK + t], acc[t]);
    }
    if (a.db && c0 == 0 && o0 + o < a.Cout) atomicAdd(&a.db[o0 + o], bsum);
}

// Weight gradient on the matrix cores (v_mfma_f32_16x16x4_f32), straight from the NCHW tensors: dW[o][c][tap] is a
// GEMM with M = 16 output channels, N = 16 input channels, K = pixels, and in NCHW both operands are contiguous along K.
// A lane (m = lane & 15, q = lane >> 4) loads 4 consecutive pixels of dy[o0+m] and of the tap-shifted x[c0+m] (one
// aligned 16-byte load per input row plus the two neighbours for the horizontal taps); MFMA j multiplies element j of
// both, so the four MFMAs of a 16-pixel chunk cover pixels {4q+j} -- a permutation of K, which a sum does not see.
// A wave keeps the K*K 16x16 accumulators in registers over its share of 64-pixel row segments; the four waves of a
// block are added in LDS and each block issues one float atomic per weight.  W % 4 == 0 and H*W % 16 == 0.

template <int K>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradArgs a) {
    constexpr int P = K / 2, KK = K * K;
    __shared__ float red[4][KK * 4 * 64];
    const int Cin = a.C0 + a.C1, ncb = (Cin + 15) / 16;
    const int o0 = (blockIdx.y / ncb) * 16, c0 = (blockIdx.y % ncb) * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 15, q = lane >> 4;
    const int o = o0 + m, c = c0 + m;
    const bool o_ok = o < a.Cout, c_ok = c < Cin;
    // lanes past the last input / output channel load from channel 0 of the first source (a valid address; x1 may be
    // null) and are zeroed afterwards
    const bool first = c < a.C0 || !c_ok;
    const float* xsrc = first ? a.x0 : a.x1;
    const int Cs = first ? a.C0 : a.C1, cs = !c_ok ? 0 : (first ? c : c - a.C0);
    f32x4 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const int HW = a.H * a.W, segs = (HW + 63) / 64, units = a.n * segs;
    // a unit = 64 consecutive pixels of a sample's H*W plane; a lane's 4 pixels lie in one row (W % 4 == 0)
    for (int u = blockIdx.x * 4 + wave; u < units; u += gridDim.x * 4) {
        const int n = u / segs, L0 = (u - n * segs) * 64;
        const float* dyplane = a.dy + ((size_t)n * a.Cout + (o_ok ? o : 0)) * HW;
        const float* xplane = xsrc + ((size_t)n * Cs + cs) * a.Hs * a.Ws;
        const int Le = min(L0 + 64, HW);
        for (int Lc = L0; Lc < Le; Lc += 16) {
            const int Lp = Lc + q * 4, y = Lp / a.W, x = Lp - y * a.W;
            float4 g = *reinterpret_cast<const float4*>(dyplane + Lp);
            if (!o_ok) g = make_float4(0.f, 0.f, 0.f, 0.f);
            const float gv[4] = {g.x, g.y, g.z, g.w};
            if (a.db) bsum += (g.x + g.y) + (g.z + g.w);
            float v[K][4 + 2 * P];                                   // pixels x-P .. x+3+P of the K input rows
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int yy = y + ky - P;
                const bool yok = yy >= 0 && yy < a.H && c_ok;
                const int yc = min(max(yy, 0), a.H - 1);
                if (a.up) {
                    const float* row = xplane + (size_t)(yc >> 1) * a.Ws;
                    const int s2 = x >> 1;
                    const float2 t2 = *reinterpret_cast<const float2*>(row + s2);
                    v[ky][P] = t2.x; v[ky][P + 1] = t2.x; v[ky][P + 2] = t2.y; v[ky][P + 3] = t2.y;
                    if (K == 3) {
                        v[ky][0] = row[max(s2 - 1, 0)];
                        v[ky][5] = row[min(s2 + 2, a.Ws - 1)];
                    }
                } else {
                    const float* row = xplane + (size_t)yc * a.Ws;
                    const float4 t4 = *reinterpret_cast<const float4*>(row + x);
                    v[ky][P] = t4.x; v[ky][P + 1] = t4.y; v[ky][P + 2] = t4.z; v[ky][P + 3] = t4.w;
                    if (K == 3) {
                        v[ky][0] = row[max(x - 1, 0)];
                        v[ky][5] = row[min(x + 4, a.W - 1)];
                    }
                }
                if (K == 3) {
                    if (x == 0) v[ky][0] = 0.0f;                    // zero padding left / right of the image
                    if (x + 4 >= a.W) v[ky][5] = 0.0f;
                }
                if (!yok) {
#pragma unroll
                    for (int j = 0; j < 4 + 2 * P; ++j) v[ky][j] = 0.0f;
                }
            }
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[ky * K + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[j], v[ky][j + kx], acc[ky * K + kx], 0, 0, 0);
        }
    }
    // D layout: lane holds rows (output channels) 4*(lane>>4)+r, column (input channel) lane&15
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(t * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    for (int e = tid; e < KK * 256; e += 256) {
        const float sum = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        const int l = e & 63, tr = e >> 6, t = tr >> 2, r = tr & 3;
        const int oo = o0 + 4 * (l >> 4) + r, cc = c0 + (l & 15);
        if (oo < a.Cout && cc < Cin) atomicAdd(&a.dw[((size_t)oo * Cin + cc) * KK + t], sum);
    }
    if (a.db && c0 == 0) {                                          // bias gradient: sum of dy over the pixels
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (q == 0 && o_ok) atomicAdd(&a.db[o], bsum);
    }
}

// ---- BatchNorm (training) + LeakyReLU + Dropout -------------------------------------------------
// per-channel double-precision sums over (n, HW): sums[c] += sum f1, sums[C + c] += sum f2
template <int MODE>   // 0: f1 = v, f2 = v*v;  1: f1 = dz, f2 = dz * xhat
__global__ __launch_bounds__(256) void bn_reduce_kernel(int n, int C, int HW, const float* v, const float* g, const float* gamma,
                                                        const float* beta, float eps, const float* mean, const float* var,
                                                        const uint8_t* mask, float drop_scale, double* sums) {
    __shared__ double s1[256], s2[256];
    const int c = blockIdx.x;
    const long total = (long)n * HW;
    double a1 = 0.0, a2 = 0.0;
    float mu = 0.f, inv = 0.f, ga = 0.f, be = 0.f;
    if (MODE == 1) { mu = mean[c]; inv = 1.0f / sqrtf(var[c] + eps); ga = gamma[c]; be = beta[c]; }
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        const size_t idx = ((size_t)(i / HW) * C + c) * HW + i % HW;
        const float x = v[idx];
        if (MODE == 0) { a1 += x; a2 += (double)x * x; }
        else {
            const float xh = (x - mu) * inv, z = ga * xh + be;
            float dz = g[idx] * (z > 0.0f ? 1.0f : 0.2f);
            if (mask) dz *= mask[idx] ? drop_scale : 0.0f;
            a1 += dz; a2 += (double)dz * xh;
        }
    }
    s1[threadIdx.x] = a1; s2[threadIdx.x] = a2;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { s1[threadIdx.x] += s1[threadIdx.x + k]; s2[threadIdx.x] += s2[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&sums[c], s1[0]); atomicAdd(&sums[C + c], s2[0]); }
}

__global__ void bn_stats_finish_kernel(int C, double count, const double* sums, float momentum, float* mean, float* var,
                                       float* running_mean, float* running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] / count;
    double vv = sums[C + c] / count - m * m;
    if (vv < 0.0) vv = 0.0;
    mean[c] = (float)m; var[c] = (float)vv;
    if (running_mean) running_mean[c] = running_mean[c] * momentum + (float)m * (1.0f - momentum);
    if (running_var) running_var[c] = running_var[c] * momentum + (float)vv * (1.0f - momentum);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(long total, int C, int HW, const float* v, const float* gamma, const float* beta,
                                                       float eps, const float* mean, const float* var, const uint8_t* mask,
                                                       float drop_scale, float* y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        const float xh = (v[i] - mean[c]) / sqrtf(var[c] + eps);
        float o = lrelu02(gamma[c] * xh + beta[c]);
        if (mask) o *= mask[i] ? drop_scale : 0.0f;
        y[i] = o;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(long total, int C, int HW, double count, const float* v, const float* gamma,
                                                           const float* beta, float eps, const float* mean, const float* var,
                                                           const uint8_t* mask, float drop_scale, const double* sums, float* g) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        const float inv = 1.0f / sqrtf(var[c] + eps);
        const float xh = (v[i] - mean[c]) * inv, z = gamma[c] * xh + beta[c];
        float dz = g[i] * (z > 0.0f ? 1.0f : 0.2f);
        if (mask) dz *= mask[i] ? drop_scale : 0.0f;
        const float dbeta = (float)(sums[c] / count), dgamma = (float)(sums[C + c] / count);
        g[i] = gamma[c] * inv * (dz - dbeta - xh * dgamma);
    }
}

__global__ void bn_param_grad_kernel(int C, const double* sums, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] += (float)sums[c];
    dgamma[c] += (float)sums[C + c];
}

// ---- loss ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_ce_kernel(int K, int HW, const float* logits, const int8_t* labels, float* loss,
                                                         float* dlogits, float grad_scale) {
    __shared__ float ssum[256];
    const int n = blockIdx.y;
    float my = 0.0f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const int l = labels[(size_t)n * HW + p];
        float m = -3.4e38f;
        for (int o = 0; o < K; ++o) m = fmaxf(m, logits[((size_t)n * K + o) * HW + p]);
        float se = 0.0f;
        for (int o = 0; o < K; ++o) se += expf(logits[((size_t)n * K + o) * HW + p] - m);
        const float w = l > -1 ? 1.0f : 0.0f;
        const int lc = l < 0 ? 0 : (l >= K ? K - 1 : l);
        for (int o = 0; o < K; ++o) {
            const float sm = expf(logits[((size_t)n * K + o) * HW + p] - m) / se;
            dlogits[((size_t)n * K + o) * HW + p] = (sm - (o == lc ? 1.0f : 0.0f)) * w * grad_scale / (float)HW;
        }
        my += w * ((m + logf(se)) - logits[((size_t)n * K + lc) * HW + p]);
    }
    ssum[threadIdx.x] = my;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) ssum[threadIdx.x] += ssum[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(&loss[n], ssum[0] / (float)HW);
}

// ---- small elementwise pieces -------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(long total, int Hs, int Ws, const float* dy, float* dx, int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % Ws), y = (int)((i / Ws) % Hs);
        const long plane = i / ((long)Hs * Ws);
        const float* s = dy + (plane * 2 * Hs + 2 * y) * 2 * Ws + 2 * x;
        const float v = (s[0] + s[1]) + (s[2 * Ws] + s[2 * Ws + 1]);
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

__global__ __launch_bounds__(256) void add_kernel(long total, const float* a, const float* b, float* out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) out[i] = a[i] + b[i];
}

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(long total, unsigned long long seed, unsigned stream_id, float keep, uint8_t* mask) {
    const long quads = (total + 3) / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
        unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), stream_id, 0x44524F50u /* "DROP" */};
        philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long i = 4 * q + j;
            if (i < total) mask[i] = (((float)(c[j] >> 8) + 0.5f) * 5.9604644775390625e-8f) < keep ? 1 : 0;
        }
    }
}

__global__ __launch_bounds__(256) void adam_kernel(long total, float* w, const float* g, float* m, float* v, float lr_t, float b1,
                                                   float b2, float eps, float rescale, float wd) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float gg = g[i] * rescale + wd * w[i];
        const float mm = b1 * m[i] + (1.0f - b1) * gg;
        const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        w[i] -= lr_t * mm / (sqrtf(vv) + eps);
    }
}

// per-device scratch for the BatchNorm reductions (2 * 8192 doubles), zeroed on the caller's stream before use
double* bn_scratch() {
    static std::mutex mu;
    static double* buf[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!buf[dev] && hipMalloc(reinterpret_cast<void**>(&buf[dev]), sizeof(double) * 2 * 8192) != hipSuccess) return nullptr;
    return buf[dev];
}

inline int grid_for(long total) { long g = (total + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

}  // namespace

extern "C" {

int gsa_train_conv(void* stream, int32_t n, const float* x0, int32_t C0, const float* x1, int32_t C1, int32_t Hs, int32_t Ws,
                   int32_t up, const float* w, int32_t Cout, int32_t K, int32_t transposed, const float* bias, float* out0,
                   int32_t Cout0, float* out1, int32_t accumulate) {
    if (n <= 0 || !x0 || C0 <= 0 || C1 < 0 || (C1 > 0 && !x1) || Hs <= 0 || Ws <= 0 || (up != 0 && up != 1) || !w || Cout <= 0 ||
        (K != 1 && K != 3) || !out0 || Cout0 <= 0 || Cout0 > Cout || (Cout0 < Cout && !out1))
        return GSA_ERR_INVALID_;
    ConvArgs a{x0, x1, C0, C1, Hs, Ws, up, Hs << up, Ws << up, w, Cout, transposed, bias, out0, Cout0, out1, accumulate};
    const uintptr_t align = reinterpret_cast<uintptr_t>(out0) | reinterpret_cast<uintptr_t>(out1);
    if (a.W % 4 == 0 && (a.H * a.W) % 16 == 0 && (align & 15) == 0) {   // matrix-core path (16-byte stores along the rows)
        const int groups = (a.H * a.W / 16 + 3) / 4, otiles = (Cout + 15) / 16;
        int gx = (2048 + otiles * n - 1) / (otiles * n);   // ~2048 blocks in all; z carries the batch size
        gx = gx > groups ? groups : gx;
        const dim3 mgrid(gx, otiles, n);
        if (K == 3) hipLaunchKernelGGL(conv_mfma_kernel<3>, mgrid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(conv_mfma_kernel<1>, mgrid, dim3(256), 0, (hipStream_t)stream, a);
        TRY_HIP(hipGetLastError());
        return GSA_OK_;
    }
    const dim3 grid(((a.H + 15) / 16) * ((a.W + 15) / 16), (Cout + 15) / 16, n);
    if (K == 3) hipLaunchKernelGGL(conv_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_conv_wgrad(void* stream, int32_t n, const float* x0, int32_t C0, const float* x1, int32_t C1, int32_t Hs, int32_t Ws,
                         int32_t up, const float* dy, int32_t Cout, int32_t K, float* dw, float* db) {
    if (n <= 0 || !x0 || C0 <= 0 || C1 < 0 || (C1 > 0 && !x1) || Hs <= 0 || Ws <= 0 || (up != 0 && up != 1) || !dy || Cout <= 0 ||
        (K != 1 && K != 3) || !dw)
        return GSA_ERR_INVALID_;
    WgradArgs a{x0, x1, C0, C1, Hs, Ws, up, Hs << up, Ws << up, dy, Cout, dw, db, n};
    const int Cin = C0 + C1;
    const int pairs = ((Cout + 15) / 16) * ((Cin + 15) / 16), work = ((a.H + 15) / 16) * ((a.W + 15) / 16) * n;
    const uintptr_t align = reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(dy);
    if (a.W % 4 == 0 && (a.H * a.W) % 16 == 0 && (align & 15) == 0) {   // matrix-core path (16-byte loads along the rows)
        const int units = n * ((a.H * a.W + 63) / 64);
        int gx = (1024 + pairs - 1) / pairs;      // ~1024 blocks (4096 waves) in all
        gx = gx < 1 ? 1 : (gx > (units + 3) / 4 ? (units + 3) / 4 : gx);
        const dim3 grid(gx, pairs, 1);
        if (K == 3) hipLaunchKernelGGL(wgrad_mfma_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(wgrad_mfma_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
        TRY_HIP(hipGetLastError());
        return GSA_OK_;
    }
    int gx = (1024 + pairs - 1) / pairs;          // ~1024 blocks in all: enough to fill the chip, few atomics per address
    gx = gx < 1 ? 1 : (gx > work ? work : gx);
    const dim3 grid(gx, pairs, 1);
    if (K == 3) hipLaunchKernelGGL(wgrad_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_bn_lrelu_fwd(void* stream, int32_t n, int32_t C, int32_t HW, const float* v, const float* gamma, const float* beta,
                           float eps, float momentum, float* mean, float* var, float* running_mean, float* running_var,
                           const uint8_t* mask, float drop_scale, float* y) {
    if (n <= 0 || C <= 0 || C > 8192 || HW <= 0 || !v || !gamma || !beta || !mean || !var || !y) return GSA_ERR_INVALID_;
    hipStream_t s = (hipStream_t)stream;
    double* sums = bn_scratch();
    if (!sums) return GSA_ERR_HIP_;
    TRY_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s));
    const long total = (long)n * HW;
    const int chunks = (int)((total + 256 * 64 - 1) / (256 * 64));
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(C, chunks < 1 ? 1 : (chunks > 256 ? 256 : chunks)), dim3(256), 0, s, n, C, HW, v,
                       (const float*)nullptr, gamma, beta, eps, (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, 1.0f, sums);
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, s, C, (double)total, sums, momentum, mean, var,
                       running_mean, running_var);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total * C)), dim3(256), 0, s, total * C, C, HW, v, gamma, beta, eps, mean, var, mask,
                       drop_scale, y);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_bn_lrelu_bwd(void* stream, int32_t n, int32_t C, int32_t HW, const float* v, const float* gamma, const float* beta,
                           float eps, const float* mean, const float* var, const uint8_t* mask, float drop_scale, float* g,
                           float* dgamma, float* dbeta) {
    if (n <= 0 || C <= 0 || C > 8192 || HW <= 0 || !v || !gamma || !beta || !mean || !var || !g || !dgamma || !dbeta) return GSA_ERR_INVALID_;
    hipStream_t s = (hipStream_t)stream;
    double* sums = bn_scratch();
    if (!sums) return GSA_ERR_HIP_;
    TRY_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s));
    const long total = (long)n * HW;
    const int chunks = (int)((total + 256 * 64 - 1) / (256 * 64));
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(C, chunks < 1 ? 1 : (chunks > 256 ? 256 : chunks)), dim3(256), 0, s, n, C, HW, v, (const float*)g,
                       gamma, beta, eps, mean, var, mask, drop_scale, sums);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total * C)), dim3(256), 0, s, total * C, C, HW, (double)total, v, gamma, beta, eps, mean,
                       var, mask, drop_scale, (const double*)sums, g);
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 63) / 64), dim3(64), 0, s, C, (const double*)sums, dgamma, dbeta);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

// ---- SyncBatchNorm: the same kernels with the per-channel sums handed to the host between the two halves (include/gsa_train.h)
static int bn_launch_reduce(hipStream_t s, int mode, int32_t n, int32_t C, int32_t HW, const float* v, const float* g, const float* gamma,
                            const float* beta, float eps, const float* mean, const float* var, const uint8_t* mask, float drop_scale,
                            double* sums) {
    TRY_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s));
    const long total = (long)n * HW;
    const int chunks = (int)((total + 256 * 64 - 1) / (256 * 64));
    const dim3 grid(C, chunks < 1 ? 1 : (chunks > 256 ? 256 : chunks));
    if (mode == 0) hipLaunchKernelGGL(bn_reduce_kernel<0>, grid, dim3(256), 0, s, n, C, HW, v, g, gamma, beta, eps, mean, var, mask, drop_scale, sums);
    else hipLaunchKernelGGL(bn_reduce_kernel<1>, grid, dim3(256), 0, s, n, C, HW, v, g, gamma, beta, eps, mean, var, mask, drop_scale, sums);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_bn_sums(void* stream, int32_t n, int32_t C, int32_t HW, const float* v, double* sums) {
    if (n <= 0 || C <= 0 || C > 8192 || HW <= 0 || !v || !sums) return GSA_ERR_INVALID_;
    return bn_launch_reduce((hipStream_t)stream, 0, n, C, HW, v, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, 1.0f, sums);
}

int gsa_train_bn_lrelu_fwd_sums(void* stream, int32_t n, int32_t C, int32_t HW, double count, const float* v, const float* gamma,
                                const float* beta, float eps, float momentum, const double* sums, float* mean, float* var,
                                float* running_mean, float* running_var, const uint8_t* mask, float drop_scale, float* y) {
    if (n <= 0 || C <= 0 || C > 8192 || HW <= 0 || !(count >= (double)n * HW) || !v || !gamma || !beta || !sums || !mean || !var || !y)
        return GSA_ERR_INVALID_;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)n * HW;
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, s, C, count, sums, momentum, mean, var, running_mean,
                       running_var);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total * C)), dim3(256), 0, s, total * C, C, HW, v, gamma, beta, eps, mean, var, mask,
                       drop_scale, y);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_bn_bwd_sums(void* stream, int32_t n, int32_t C, int32_t HW, const float* v, const float* gamma, const float* beta, float eps,
                          const float* mean, const float* var, const uint8_t* mask, float drop_scale, const float* g, double* sums) {
    if (n <= 0 || C <= 0 || C > 8192 || HW <= 0 || !v || !gamma || !beta || !mean || !var || !g || !sums) return GSA_ERR_INVALID_;
    return bn_launch_reduce((hipStream_t)stream, 1, n, C, HW, v, g, gamma, beta, eps, mean, var, mask, drop_scale, sums);
}

int gsa_train_bn_lrelu_bwd_sums(void* stream, int32_t n, int32_t C, int32_t HW, double count, const float* v, const float* gamma,
                                const float* beta, float eps, const float* mean, const float* var, const uint8_t* mask, float drop_scale,
                                const double* sums_all, const double* sums_own, float* g, float* dgamma, float* dbeta) {
    if (n <= 0 || C <= 0 || C > 8192 || HW <= 0 || !(count >= (double)n * HW) || !v || !gamma || !beta || !mean || !var || !sums_all ||
        !sums_own || !g || !dgamma || !dbeta)
        return GSA_ERR_INVALID_;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)n * HW;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total * C)), dim3(256), 0, s, total * C, C, HW, count, v, gamma, beta, eps, mean, var,
                       mask, drop_scale, sums_all, g);
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 63) / 64), dim3(64), 0, s, C, sums_own, dgamma, dbeta);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_softmax_ce(void* stream, int32_t n, int32_t classes, int32_t HW, const float* logits, const int8_t* labels, float* loss,
                         float* dlogits, float grad_scale) {
    if (n <= 0 || classes < 2 || classes > 64 || HW <= 0 || !logits || !labels || !loss || !dlogits) return GSA_ERR_INVALID_;
    hipStream_t s = (hipStream_t)stream;
    TRY_HIP(hipMemsetAsync(loss, 0, sizeof(float) * n, s));
    const int gx = (HW + 255) / 256;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(gx > 1024 ? 1024 : gx, n), dim3(256), 0, s, classes, HW, logits, labels, loss, dlogits, grad_scale);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_upsample2_bwd(void* stream, int32_t n, int32_t C, int32_t Hs, int32_t Ws, const float* dy_up, float* dx, int32_t accumulate) {
    if (n <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || !dy_up || !dx) return GSA_ERR_INVALID_;
    const long total = (long)n * C * Hs * Ws;
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, total, Hs, Ws, dy_up, dx, accumulate);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_add(void* stream, int64_t count, const float* a, const float* b, float* out) {
    if (count <= 0 || !a || !b || !out) return GSA_ERR_INVALID_;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, (long)count, a, b, out);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_dropout_mask(void* stream, int64_t count, uint64_t seed, uint32_t stream_id, float keep_prob, uint8_t* mask) {
    if (count <= 0 || !mask || !(keep_prob > 0.0f && keep_prob <= 1.0f)) return GSA_ERR_INVALID_;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((count + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (long)count,
                       (unsigned long long)seed, stream_id, keep_prob, mask);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

int gsa_train_adam(void* stream, int64_t count, float* w, const float* g, float* m, float* v, float lr_t, float beta1, float beta2,
                   float eps, float rescale, float wd) {
    if (count <= 0 || !w || !g || !m || !v) return GSA_ERR_INVALID_;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, (long)count, w, g, m, v, lr_t, beta1, beta2, eps,
                       rescale, wd);
    TRY_HIP(hipGetLastError());
    return GSA_OK_;
}

}  // extern "C"
