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

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace gsa {
namespace lean {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int LH = 18, LW = 18, RS = LW * 16 + 4;      // halo image: 18 rows of 18 pixels x 16 channels, row stride 292 floats (bank-conflict free patch reads)
constexpr int IMG = LH * RS;                           // floats per image buffer (21 KB)
constexpr int SEG = 16 * 256;                          // U floats of the (16 couts, 16 channels) panel: [f][kq][16][cg]
constexpr int NPIX = LH * LW;                          // 324 halo pixels = 1296 chunks = 5 rounds of 256 threads + 16 chunks

__device__ __forceinline__ int xcd_block(int b, int nb) { return (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b; }

// packed fp32 operations (IEEE, the same bits as the scalar forms); hazards towards / from MFMAs are settled by hand below
__device__ __forceinline__ f32x2 pk_add2(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 m, f32x2 b) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(m), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_mul2(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 add4(const f32x4& a, const f32x4& b) { const f32x2 lo = pk_add2(a.xy, b.xy), hi = pk_add2(a.zw, b.zw); return f32x4{lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ f32x4 sub4(const f32x4& a, const f32x4& b) { const f32x2 lo = pk_sub2(a.xy, b.xy), hi = pk_sub2(a.zw, b.zw); return f32x4{lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ f32x4 fma4(const f32x4& a, const f32x4& m, const f32x4& b) {
    const f32x2 lo = pk_fma2(a.xy, m.xy, b.xy), hi = pk_fma2(a.zw, m.zw, b.zw);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
// LeakyReLU(0.2) = max(v, 0.2 v) (gsa_kernels.hip lrelu: the same bits as the select form)
__device__ __forceinline__ f32x4 lrelu4(const f32x4& v, f32x2 k02) {
    const f32x2 lo = pk_mul2(v.xy, k02), hi = pk_mul2(v.zw, k02);
    return f32x4{fmaxf(v.x, lo.x), fmaxf(v.y, lo.y), fmaxf(v.z, hi.x), fmaxf(v.w, hi.y)};
}
// 2 wait states between a vector-ALU result and the MFMA that reads it (the hazard recognizer does not see through inline asm)
__device__ __forceinline__ void valu_settle(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
// 11 wait states between an 8-pass MFMA result and a vector-ALU read of it
__device__ __forceinline__ void mfma_settle(f32x4 (&a)[16]) {
    asm("s_nop 7\n\ts_nop 3"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
          "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}

struct Tile { int n, y0, x0; };

// EPI_DEC: y = lrelu(fmaf(v, bn_s, bn_beta)) [+ residual at half resolution (RES)];  AFF: the source carries AdaIN coefficients.
template <int EPI, bool AFF, bool RES>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_c16(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sA = smem;                 // [2][IMG]
    float* const sB = smem + 2 * IMG;       // [SEG]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4, part = tid & 3;
    const int H = p.H, W = p.W;
    const int per = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int w_begin = xcd_block(blockIdx.x, gridDim.x) * per;
    const int w_end = min(p.total_tiles, w_begin + per);
    if (w_begin >= w_end) return;
    const int items = w_end - w_begin;

    {   // the 16 KB weight panel, once
        const f32x4* src = reinterpret_cast<const f32x4*>(p.wpk);
        f32x4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = src[tid + 256 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) reinterpret_cast<f32x4*>(sB)[tid + 256 * j] = r[j];
    }

    // ---- staging: chunk q = tid + 256 k = (halo pixel q >> 2, channels 4 * part ..); byte offsets from the halo origin
    unsigned s_off[6], eflags = 0;
    int l_off[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int pix = (tid + 256 * k) >> 2;
        const bool real = pix < NPIX;
        const int ly = real ? pix / LW : 1, lx = real ? pix % LW : 1;
        s_off[k] = (unsigned)(((ly * W + lx) * 16 + part * 4) * 4);
        l_off[k] = real ? ly * RS + lx * 16 + part * 4 : -1;
        eflags |= (unsigned)((ly == 0 ? 1 : 0) | (ly == LH - 1 ? 2 : 0) | (lx == 0 ? 4 : 0) | (lx == LW - 1 ? 8 : 0)) << (4 * k);
    }
    const unsigned safe_off = (unsigned)((((W + 1) * 16) + part * 4) * 4);      // the tile's own first pixel: always inside the image
    auto edge_code = [&](const Tile& t) { return (t.y0 == 0 ? 1 : 0) | (t.y0 + 16 == H ? 2 : 0) | (t.x0 == 0 ? 4 : 0) | (t.x0 + 16 == W ? 8 : 0); };
    auto advance = [&](Tile& t) { t.x0 += 16; if (t.x0 == W) { t.x0 = 0; t.y0 += 16; if (t.y0 == H) { t.y0 = 0; t.n += 1; } } };

    f32x4 ra[6];                              // the item in flight (loaded, not yet written to LDS)
    f32x4 cA, cB, nA, nB;                     // AdaIN A / B of this thread's four channels: current sample, sample of the item in flight
    cA = cB = nA = nB = f32x4{0.f, 0.f, 0.f, 0.f};
    int n_coef = -1, n_next = -1;
    auto load_item = [&](const Tile& t, int e) {
        const char* hb = reinterpret_cast<const char*>(p.src0) + (((long)(t.n * H + t.y0) * W + t.x0) - (W + 1)) * 64;
        // round 5 holds 16 real chunks (lanes 0..15 of wave 0); everybody issues it -- the idle lanes read the tile's first pixel
        if (e) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const unsigned bad = (eflags >> (4 * k)) & (unsigned)e;
                const unsigned off = s_off[k] + (bad ? safe_off - s_off[k] : 0u);
                ra[k] = *reinterpret_cast<const f32x4*>(hb + off);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) ra[k] = *reinterpret_cast<const f32x4*>(hb + s_off[k]);
        }
        if (AFF && t.n != n_next) {           // wave-uniform: first item and sample changes
            const f32x4* a = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)t.n * 16 + part * 4);      // (mean, A, B, -) x 4
            const f32x4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
            nA = f32x4{a0[1], a1[1], a2[1], a3[1]};
            nB = f32x4{a0[2], a1[2], a2[2], a3[2]};
            n_next = t.n;
        }
    };
    auto write_item = [&](const Tile& t, int e, int buf) {
        float* img = sA + buf * IMG;
        if (AFF && t.n != n_coef) { cA = nA; cB = nB; n_coef = t.n; }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (e) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const f32x4 v = AFF ? fma4(ra[k], cA, cB) : ra[k];
                const bool bad = ((eflags >> (4 * k)) & (unsigned)e) != 0;
                if (k < 5 || l_off[5] >= 0) *reinterpret_cast<f32x4*>(img + l_off[k]) = bad ? z : v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k < 5 || l_off[5] >= 0) *reinterpret_cast<f32x4*>(img + l_off[k]) = AFF ? fma4(ra[k], cA, cB) : ra[k];
        }
    };

    // ---- operands: lane (i16, kq) owns Winograd tile (ty, tx) = (2 b3 + b1, 2 b2 + b0) of its wave's 8x8 quadrant and k slot kq
    const int wty = ((i16 >> 3) & 1) * 2 + ((i16 >> 1) & 1), wtx = ((i16 >> 2) & 1) * 2 + (i16 & 1);
    const int qy = wave >> 1, qx = wave & 1;
    const int pbase = (qy * 8 + 2 * wty) * RS + (qx * 8 + 2 * wtx) * 16 + kq * 4;
    const int bbase = (kq * 16 + i16) * 4;
    // ---- epilogue: the lane's tile = output pixels (2 wty + i, 2 wtx + j) of the quadrant, channels 4 kq .. 4 kq + 3
    const unsigned out_off = (unsigned)((((qy * 8 + 2 * wty) * W + qx * 8 + 2 * wtx) * 16 + kq * 4) * 4);      // bytes from the tile origin
    const unsigned res_off = RES ? (unsigned)((((qy * 4 + wty) * (W >> 1) + qx * 4 + wtx) * 16 + kq * 4) * 4) : 0u;
    f32x4 e2 = {0.f, 0.f, 0.f, 0.f}, e3 = e2;
    if (EPI == EPI_DEC) {
        e2 = *reinterpret_cast<const f32x4*>(p.bn_s + kq * 4);
        e3 = *reinterpret_cast<const f32x4*>(p.bn_beta + kq * 4);
    }
    const f32x2 k02 = {0.2f, 0.2f};
    f32x4 rr = {0.f, 0.f, 0.f, 0.f};
    auto epilogue_loads = [&](const Tile& t) {
        if (RES) {
            const char* rb = reinterpret_cast<const char*>(p.resid) + ((long)(t.n * (H >> 1) + (t.y0 >> 1)) * (W >> 1) + (t.x0 >> 1)) * 64;
            rr = *reinterpret_cast<const f32x4*>(rb + res_off);
        }
    };

    f32x4 acc[16];
    auto multiply = [&](int buf) {
        const float* a_img = sA + buf * IMG + pbase;
        const float* b_img = sB + bbase;
        // input transform V = B^T d B: rows t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3; then the same along the columns
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
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[j][cg], V[f][cg], cg == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[f], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
    };
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
        char* ob = reinterpret_cast<char*>(p.out) + ((long)(t.n * H + t.y0) * W + t.x0) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = y[i][j];
                if (EPI == EPI_DEC) {
                    v = lrelu4(fma4(v, e2, e3), k02);
                    if (RES) v = add4(rr, v);
                }
                *reinterpret_cast<f32x4*>(ob + (long)i * W * 64 + out_off + j * 64) = v;
            }
    };

    // ---- pipeline: item `it` is multiplied out of LDS buffer it & 1, item it+1 sits in ra, item it+2 is being loaded
    Tile tc, tr;
    {
        const int tx = w_begin % p.tiles_x, r = w_begin / p.tiles_x;
        tc.x0 = tx * 16; tc.y0 = (r % p.tiles_y) * 16; tc.n = r / p.tiles_y;
    }
    int ec = edge_code(tc), er;
    load_item(tc, ec);
    write_item(tc, ec, 0);
    tr = tc; er = ec;
    if (items > 1) { advance(tr); er = edge_code(tr); load_item(tr, er); }
    __syncthreads();
    for (int it = 0; it < items; ++it) {
        const bool has_next = it + 1 < items;
        if (has_next) write_item(tr, er, (it + 1) & 1);
        epilogue_loads(tc);
        Tile t2 = tr; int e2c = er;
        if (it + 2 < items) { advance(t2); e2c = edge_code(t2); load_item(t2, e2c); }
        multiply(it & 1);
        epilogue(tc);
        __syncthreads();
        tc = tr; ec = er; tr = t2; er = e2c;
    }
}

constexpr int kMaxDev = 64;
struct LeanState { bool attr_done = false; int cus = 0; };
static std::mutex g_mu;

template <int EPI, bool AFF, bool RES>
hipError_t launch_t(const ConvParams& p, int n, hipStream_t s) {
    static LeanState st[kMaxDev];
    auto kern = conv3x3_wino_c16<EPI, AFF, RES>;
    const size_t lds = sizeof(float) * (2 * IMG + SEG);
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
    const int gx = std::min(q.total_tiles, cus * 2);      // persistent: two workgroups per CU (58 KB of LDS each)
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, s, q);
    return hipGetLastError();
}

}  // namespace lean
using namespace lean;

// the layers this file takes: Winograd form (conv_uses_wino), 16 -> 16 channels, decoder epilogue, residual absent or one half-resolution tensor
bool wino_lean_applies(const ConvParams& p, int epi) {
    static const bool enabled = !(getenv("GSA_WINO_LEAN") && atoi(getenv("GSA_WINO_LEAN")) == 0);
    if (!enabled || p.C0 != 16 || p.Cout != 16 || p.H < 32 || p.H != p.W || p.H % 16) return false;
    if (epi != EPI_DEC) return false;
    if (p.resid != nullptr && (p.resid_up != 1 || p.resid1 != nullptr)) return false;
    return true;
}

const char* wino_lean_name(const ConvParams& p, int epi) {
    static thread_local char buf[96];
    snprintf(buf, sizeof buf, "void gsa::lean::conv3x3_wino_c16<%d, %s, %s>(gsa::ConvParams)", epi, p.aff0 ? "true" : "false",
             p.resid ? "true" : "false");
    return buf;
}

hipError_t launch_wino_lean(const ConvParams& p, int epi, int n, hipStream_t s) {
    if (epi == EPI_DEC) {
        if (p.resid) return p.aff0 ? launch_t<EPI_DEC, true, true>(p, n, s) : launch_t<EPI_DEC, false, true>(p, n, s);
        return p.aff0 ? launch_t<EPI_DEC, true, false>(p, n, s) : launch_t<EPI_DEC, false, false>(p, n, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace gsa
