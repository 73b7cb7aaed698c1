// The 3x3 convolutions of the bf16 MFMA mode (BASELINE config 5; reference networks_stylegan.py:354-457, networks_seg.py:7-46,64-79) from
// 32 px on, in the lean form of gsa_wino_lean.hip (round 5).  conv3x3_mfma<..., BF = true> -- the general kernel of rounds 1-3 -- spends 10-15 k
// cycles per (tile, 16-channel block) item on these layers for 36 bf16 MFMAs (~300 cycles): whole-pixel staging with a per-channel AdaIN
// table, 64-bit address arithmetic per load, a quad transpose per output row, both border paths, accumulator clearing.  Here:
//   * one workgroup = 4 waves on 16x16 tiles x ONE 16-channel output group (blockIdx.y), an item = (tile, 16-channel block); image and weight
//     block double-buffered in LDS, activations prefetched two items ahead through registers, the item's 4.5 KB weight block by LDS-DMA;
//   * staging by 16-byte chunks (8 bf16 channels) on every thread: unpack, one packed fma per channel pair with the AdaIN coefficients of the
//     block out of a per-sample LDS table, round to bf16 (RNE) -- the operand rule of the bf16 mode, unchanged; wave-uniform bases, one border
//     flag word, executed on border tiles only;
//   * weights as the MFMA's A operand: a lane ends up with four consecutive output channels of ONE pixel -- 8-byte NHWC stores without a
//     transpose, packed epilogue arithmetic, statistics of an aligned x-quad finished between four lanes with two DPP steps;
//   * the chains of a tile's first block start from the inline constant 0.
// The products, their order inside a chain (block, tap, 16 channels per MFMA) and every rounding are conv3x3_mfma<..., true>'s: the same values.
#include "gsa_kernels.h"
#include "gsa_dev.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace gsa {
namespace lean {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BPX = 8;                    // 4-byte slots per staged pixel: 16 bf16 channels
constexpr int BRS = 18 * BPX + 4;         // row stride of the halo image (conv3x3_mfma's: conflict-free 8-byte patch reads)
constexpr int BIMG = 18 * BRS;            // slots per image buffer (10.4 KB)
constexpr int BSEG = 9 * 128;             // slots of a (16 couts, 16 channels) weight block: [tap][kq][16][4 bf16]
constexpr int BNPIX = 18 * 18;            // 324 halo pixels = 648 chunks of 8 channels

struct BTile { int n, y0, x0; };

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// EPI_SYNTH: y = lrelu((v + nscale * noise) + nbias) and the instance-norm statistics of y (direct form);  EPI_DEC: y = lrelu(fmaf(v, bn_s, bn_beta))
// [+ the residual, one tensor at half resolution (RES)];  AFF: the source carries AdaIN coefficients.  Output and residual are bf16 tensors.
// NT = output groups per workgroup (blockIdx.y = group NT-tuple): the MFMAs are an eighth of the fp32 form's, so an item is its staging -- one
// staged image feeds 16 NT output channels.
template <int EPI, bool AFF, bool RES, int NT>
__global__ __launch_bounds__(256, (EPI == EPI_SYNTH || NT == 4 || (RES && NT == 2)) ? 2 : 3) void conv3x3_bf16_lean(ConvParams p) {
    constexpr int NCH = 3;                // staging rounds per item: 648 chunks on 256 threads (2 full + 136)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sA = smem;                       // [2][BIMG]
    float* const sB = smem + 2 * BIMG;            // [2][NT][BSEG]
    float* const sE = sB + 2 * NT * BSEG;         // [NT][4][16]: nscale | nbias | bn_s | bn_beta of the groups' output channels
    float* const sC = sE + 64 * NT;               // [nblk][32]: A of the block's 16 channels, then B (the sample being staged)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4, half = tid & 1;
    const int H = p.H, W = p.W, CIN = p.C0, COUT = p.Cout, nblk = p.C0 >> 4;
    const int g = blockIdx.y;
    const int per = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int w_begin = xcd_block(blockIdx.x, gridDim.x) * per;
    const int w_end = min(p.total_tiles, w_begin + per);
    if (w_begin >= w_end) return;
    const int items = (w_end - w_begin) * nblk;

    // ---- staging: chunk q = tid + 256 k = (halo pixel q >> 1, channels 8 * half .. of the item's block); byte offsets from the halo origin
    unsigned s_off[NCH], eflags = 0;
    int l_off[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int pix = (tid + 256 * k) >> 1;
        const bool real = pix < BNPIX;
        const int ly = real ? pix / 18 : 1, lx = real ? pix % 18 : 1;
        s_off[k] = (unsigned)(((ly * W + lx) * CIN + half * 8) * 2);
        l_off[k] = real ? ly * BRS + lx * BPX + half * 4 : -1;
        eflags |= (unsigned)((ly == 0 ? 1 : 0) | (ly == 17 ? 2 : 0) | (lx == 0 ? 4 : 0) | (lx == 17 ? 8 : 0)) << (4 * k);
    }
    const unsigned safe_off = (unsigned)((((W + 1) * CIN) + half * 8) * 2);      // the tile's own first pixel: always inside the image
    auto edge_code = [&](const BTile& t) { return (t.y0 == 0 ? 1 : 0) | (t.y0 + 16 == H ? 2 : 0) | (t.x0 == 0 ? 4 : 0) | (t.x0 + 16 == W ? 8 : 0); };
    auto advance = [&](BTile& t) { t.x0 += 16; if (t.x0 == W) { t.x0 = 0; t.y0 += 16; if (t.y0 == H) { t.y0 = 0; t.n += 1; } } };

    u32x4 ra[1][NCH];                             // the item in flight: this thread's chunks (8 bf16 each)
    // the item's weight blocks: NT x 288 chunks of 16 bytes, chunk c = tid + 256 j of the LDS panel [NT][BSEG] (linear: c * 16 bytes) comes from
    // group c / 288 -- a wave's 64 chunks are consecutive in LDS whatever their sources
    constexpr int WR = (NT * 288 + 255) / 256;      // rounds: 2 / 3 / 5, the last one partial (whole waves but for NT = 1)
    const float* const wgrp = p.wpk + (size_t)g * NT * nblk * BSEG;
    unsigned w_src[WR];
#pragma unroll
    for (int j = 0; j < WR; ++j) {
        const int c = min(tid + 256 * j, NT * 288 - 1);
        w_src[j] = (unsigned)(((c / 288) * nblk * BSEG + (c % 288) * 4) * 4);
    }
    auto dma_weights = [&](int cb, int buf) {
#ifdef GSA_DBG_HOOKS
        if (p.dbg & 1024) return;     // diagnostic build only (WRONG results): no weight DMA
#endif
        const char* wb = reinterpret_cast<const char*>(wgrp + (size_t)cb * BSEG);
#pragma unroll
        for (int j = 0; j < WR; ++j)
            if (tid + 256 * j < NT * 288)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + w_src[j]),
                                                 (__attribute__((address_space(3))) void*)(sB + buf * NT * BSEG + (wave * 64 + 256 * j) * 4), 16, 0, 0);
    };
    auto load_item = [&](auto set_tag, const BTile& t, int e, int cb) {
        constexpr int S = decltype(set_tag)::value;
#ifdef GSA_DBG_HOOKS
        if (p.dbg & 512) return;      // diagnostic build only (WRONG results): no activation loads
#endif
        const char* hb = reinterpret_cast<const char*>(p.src0) + ((((long)(t.n * H + t.y0) * W + t.x0) - (W + 1)) * CIN + cb * 16) * 2;
        if (e) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const unsigned bad = (eflags >> (4 * k)) & (unsigned)e;
                const unsigned off = s_off[k] + (bad ? safe_off - s_off[k] : 0u);
                ra[S][k] = *reinterpret_cast<const u32x4*>(hb + off);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) ra[S][k] = *reinterpret_cast<const u32x4*>(hb + s_off[k]);
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
    auto write_item = [&](auto set_tag, const BTile& t, int e, int cb, int buf) {
        constexpr int S = decltype(set_tag)::value;
#ifdef GSA_DBG_HOOKS
        if (p.dbg & 128) return;      // diagnostic build only (WRONG results): no staging (AdaIN + LDS stores)
#endif
        float* img = sA + buf * BIMG;
        if (AFF && t.n != n_coef) { coefficients(t.n); n_coef = t.n; }
        f32x4 kA0 = {0.f, 0.f, 0.f, 0.f}, kA1 = kA0, kB0 = kA0, kB1 = kA0;
        if (AFF) {
            const float* tab = sC + cb * 32 + half * 8;
            kA0 = *reinterpret_cast<const f32x4*>(tab); kA1 = *reinterpret_cast<const f32x4*>(tab + 4);
            kB0 = *reinterpret_cast<const f32x4*>(tab + 16); kB1 = *reinterpret_cast<const f32x4*>(tab + 20);
        }
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            u32x4 v = ra[S][k];
            if (AFF) {      // the bf16 mode's operand rule: fmaf(x, A, B) in fp32 on the stored bf16 value, rounded to bf16 (RNE)
                const f32x4 lo = {bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1])};
                const f32x4 hi = {bf16_lo(v[2]), bf16_hi(v[2]), bf16_lo(v[3]), bf16_hi(v[3])};
                const f32x4 a = fma4(lo, kA0, kB0), b = fma4(hi, kA1, kB1);
                v = u32x4{pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])};
            }
            if (e && ((eflags >> (4 * k)) & (unsigned)e)) v = z;
            if (k < NCH - 1 || l_off[NCH - 1] >= 0) *reinterpret_cast<u32x4*>(img + l_off[k]) = v;
        }
    };

    // ---- operands: wave (qy, qx) owns an 8x8 quadrant = four 4x4 patches; lane (i16, kq) = pixel i16 of a patch (B operand) / output channel
    // i16 (A operand), channels 4 kq .. 4 kq + 3
    const int qy = wave >> 1, qx = wave & 1;
    const int py = i16 >> 2, px = i16 & 3;
    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) abase[m] = (qy * 8 + (m >> 1) * 4 + py) * BRS + (qx * 8 + (m & 1) * 4 + px) * BPX + kq * 2;
    const int bbase = (kq * 16 + i16) * 2;
    const int co4 = g * NT * 16 + kq * 4;
    if (tid < 16 * NT) {
        float* e = sE + (tid >> 4) * 64 + (tid & 15);
        const int co = g * NT * 16 + tid;
        e[0] = EPI == EPI_SYNTH ? p.nscale[co] : 0.f;
        e[16] = EPI == EPI_SYNTH ? p.nbias[co] : 0.f;
        e[32] = EPI == EPI_DEC ? p.bn_s[co] : 0.f;
        e[48] = EPI == EPI_DEC ? p.bn_beta[co] : 0.f;
    }
    // pixel of patch m inside the tile: (qy * 8 + (m >> 1) * 4 + py, qx * 8 + (m & 1) * 4 + px)
    unsigned out_off[4], nz_off[4], res_off[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int ty = qy * 8 + (m >> 1) * 4 + py, tx = qx * 8 + (m & 1) * 4 + px;
        out_off[m] = (unsigned)(((ty * W + tx) * COUT + co4) * 2);
        nz_off[m] = (unsigned)((ty * W + tx) * 4);
        res_off[m] = (unsigned)((((ty >> 1) * (W >> 1) + (tx >> 1)) * COUT + co4) * 2);
    }
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    const int s2 = stat_s2(H * W);
    // statistics: the four lanes of a quad hold one aligned x-quad; after the two DPP steps every one of them has the quad's sums of its four
    // channels -- lane px converts and keeps channel 4 kq + px only (one conversion pair per lane and patch instead of four)
    constexpr int NS = EPI == EPI_SYNTH ? NT : 1;
    unsigned long long dI1[NS], dI2[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) dI1[q] = dI2[q] = 0ull;
    auto flush_stats = [&](int n) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            unsigned long long I1 = dI1[q], I2 = dI2[q];
            I1 += shfl_xor_u64(I1, 4); I2 += shfl_xor_u64(I2, 4);      // over the four rows of the patch (py)
            I1 += shfl_xor_u64(I1, 8); I2 += shfl_xor_u64(I2, 8);
            if (py == 0) {
                StatPart* a = p.partials + ((size_t)n * p.prow + (blockIdx.x & (kDirectRows - 1))) * COUT + co4 + q * 16 + px;
                atomicAdd(&a->s1, I1);
                atomicAdd(&a->s2, I2);
            }
            dI1[q] = dI2[q] = 0ull;
        }
    };
    const f32x2 k02 = {0.2f, 0.2f};
    u32x2 rres[RES ? NT : 1][4];
    // the tile's noise / residual values are requested IN FRONT of the loads of item it + 2: completion is in order, so waiting for them in the
    // epilogue does not wait for those loads
    auto epilogue_loads = [&](const BTile& t) {
        if (EPI == EPI_SYNTH) {
            const char* nb = reinterpret_cast<const char*>(p.noise) + ((long)(t.n * H + t.y0) * W + t.x0) * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) nz[m] = *reinterpret_cast<const float*>(nb + nz_off[m]);
        }
        if (RES) {
            const char* rb = reinterpret_cast<const char*>(p.resid) + ((long)(t.n * (H >> 1) + (t.y0 >> 1)) * (W >> 1) + (t.x0 >> 1)) * COUT * 2;
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int m = 0; m < 4; ++m) rres[q][m] = *reinterpret_cast<const u32x2*>(rb + res_off[m] + q * 32);
        }
    };

    f32x4 acc[NT][4];
    auto multiply = [&](auto first_tag, int buf, int wslot) {
        constexpr bool FIRST = decltype(first_tag)::value;
#ifdef GSA_DBG_HOOKS
        if (p.dbg & 256) { if (FIRST) for (int q = 0; q < NT; ++q) for (int m = 0; m < 4; ++m) acc[q][m] = f32x4{0.f, 0.f, 0.f, 0.f}; return; }      // no operand reads, no MFMAs
#endif
        const float* a_img = sA + buf * BIMG;
        const float* b_img = sB + wslot * NT * BSEG + bbase;
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * BRS + (tap % 3) * BPX;
            s16x4 w[NT], x[4];
#pragma unroll
            for (int q = 0; q < NT; ++q) w[q] = *reinterpret_cast<const s16x4*>(b_img + q * BSEG + tap * 128);
#pragma unroll
            for (int m = 0; m < 4; ++m) x[m] = *reinterpret_cast<const s16x4*>(a_img + abase[m] + toff);
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w[q], x[m], FIRST && tap == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[q][m], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue = [&](const BTile& t) {
        char* ob = reinterpret_cast<char*>(p.out) + ((long)(t.n * H + t.y0) * W + t.x0) * COUT * 2;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
        const float* sEq = sE + q * 64 + kq * 4;
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(sEq), e1 = *reinterpret_cast<const f32x4*>(sEq + 16);
        const f32x4 e2 = *reinterpret_cast<const f32x4*>(sEq + 32), e3 = *reinterpret_cast<const f32x4*>(sEq + 48);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 v = acc[q][m];
            if (EPI == EPI_DEC) {
                v = lrelu4(fma4(v, e2, e3), k02);
                if (RES) v = add4(f32x4{bf16_lo(rres[q][m][0]), bf16_hi(rres[q][m][0]), bf16_lo(rres[q][m][1]), bf16_hi(rres[q][m][1])}, v);
            }
            if (EPI == EPI_SYNTH) {
                const f32x2 n2 = {nz[m], nz[m]};
                const f32x2 tlo = pk_mul2(e0.xy, n2), thi = pk_mul2(e0.zw, n2);
                v = lrelu4(add4(add4(v, f32x4{tlo.x, tlo.y, thi.x, thi.y}), e1), k02);
            }
#ifdef GSA_DBG_HOOKS
            if (!(p.dbg & 64))        // diagnostic build only (WRONG results): no output stores
#endif
            *reinterpret_cast<u32x2*>(ob + out_off[m] + q * 32) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            if (EPI == EPI_SYNTH) {
                // statistics per aligned x-quad (the four lanes of a quad hold x .. x + 3 of one row): s = (v0 + v1) + (v2 + v3), q likewise
                const f32x2 a0 = pk_mul2(v.xy, v.xy), a1 = pk_mul2(v.zw, v.zw);
                const f32x4 sq = {a0.x, a0.y, a1.x, a1.y};
                f32x4 sv, qv;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sv[c] = v[c] + dpp_quad<0xB1>(v[c]);
                    sv[c] = sv[c] + dpp_quad<0x4E>(sv[c]);
                    qv[c] = sq[c] + dpp_quad<0xB1>(sq[c]);
                    qv[c] = qv[c] + dpp_quad<0x4E>(qv[c]);
                }
                const float s_mine = px == 0 ? sv[0] : (px == 1 ? sv[1] : (px == 2 ? sv[2] : sv[3]));
                const float q_mine = px == 0 ? qv[0] : (px == 1 ? qv[1] : (px == 2 ? qv[2] : qv[3]));
                dI1[EPI == EPI_SYNTH ? q : 0] += to_fixed(s_mine, kStatScale1);
                dI2[EPI == EPI_SYNTH ? q : 0] += to_fixed_sq(q_mine, s2);
            }
        }
        }
    };

    BTile tc, tr;
    {
        const int tx = w_begin % p.tiles_x, r = w_begin / p.tiles_x;
        tc.x0 = tx * 16; tc.y0 = (r % p.tiles_y) * 16; tc.n = r / p.tiles_y;
    }
    int ec = edge_code(tc), er, cbc = 0, cbr = 0;
    auto next_item = [&](BTile& t, int& e, int& cb) {
        if (++cb == nblk) { cb = 0; advance(t); e = edge_code(t); }
    };
    using S0 = std::integral_constant<int, 0>;
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, sw = 0, sl = 0, sm = 0, se = 0, sb = 0;      // diagnostic build only (make stamp)
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)k5; (void)sw; (void)sl; (void)sm; (void)se; (void)sb;
    dma_weights(0, 0);
    load_item(S0{}, tc, ec, 0);
    write_item(S0{}, tc, ec, 0, 0);
    tr = tc; er = ec;
    if (items > 1) {
        next_item(tr, er, cbr);
        load_item(S0{}, tr, er, cbr);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // the DMA pieces are older than the three loads of the second item
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int it = 0; it < items; ++it) {
        const bool has_next = it + 1 < items;
        TICK(k0);
        if (has_next) {
            dma_weights(cbr, (it + 1) & 1);      // the buffer item it - 1 used: every wave is past that item's closing barrier
            write_item(S0{}, tr, er, cbr, (it + 1) & 1);
        }
        TICK(k1);
        if (cbc == nblk - 1) epilogue_loads(tc);
        BTile t2 = tr; int e2c = er, cb2 = cbr;
        if (it + 2 < items) { next_item(t2, e2c, cb2); load_item(S0{}, t2, e2c, cb2); }
        TICK(k2);
        if (cbc == 0) multiply(std::true_type{}, it & 1, it & 1);
        else multiply(std::false_type{}, it & 1, it & 1);
        TICK(k3);
        const bool stored = cbc == nblk - 1;
        const bool flushed = stored && EPI == EPI_SYNTH && (!has_next || tr.n != tc.n);
        if (stored) {
            epilogue(tc);
            if (flushed) flush_stats(tc.n);
        }
        TICK(k4);
        // this wave's DMA pieces have landed before anybody passes the barrier: they are older than the three loads of item it + 2 and the
        // tile's stores (in-order completion); an iteration that issued no such loads (the last two) waits for everything
        if (it + 2 < items) {
            if (stored && !flushed) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 + 4 * NT) : "memory");      // (a trip that also issued the statistics' atomics keeps the stricter wait)
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        TICK(k5);
        TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4); TSUM(sb, k4, k5);
        tc = tr; ec = er; cbc = cbr; tr = t2; er = e2c; cbr = cb2;
    }
    TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
    TFLUSH(12, (unsigned long long)items); TFLUSH(15, 1ull);
}

constexpr int kMaxDevB = 64;
struct BState { bool attr_done = false; int cus = 0; };
static std::mutex g_bmu;

template <int EPI, bool AFF, bool RES, int NT>
hipError_t launch_bf16_t(const ConvParams& p, int n, hipStream_t s) {
    static BState st[kMaxDevB];
    auto kern = conv3x3_bf16_lean<EPI, AFF, RES, NT>;
    const int nblk = p.C0 / 16;
    const size_t lds = sizeof(float) * (2 * BIMG + 2 * NT * BSEG + 64 * NT + nblk * 32);
    if (p.device < 0 || p.device >= kMaxDevB) return hipErrorInvalidDevice;
    int cus;
    {
        std::lock_guard<std::mutex> lk(g_bmu);
        BState& d = st[p.device];
        if (!d.attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, p.device) != hipSuccess) d.cus = 256;
            d.attr_done = true;
        }
        cus = d.cus;
    }
    ConvParams q = p;
    q.tiles_x = p.W / 16;
    q.tiles_y = p.H / 16;
    q.groups = p.Cout / (16 * NT);                     // workgroup columns
    q.total_tiles = q.tiles_x * q.tiles_y * n;         // per output-channel group
    if (EPI == EPI_SYNTH) {
        q.stats_direct = 1;
        q.prow = kDirectRows;
        if (p.stat_rows_host) *p.stat_rows_host = q.prow;
    }
    // persistent workgroups, three per CU (synthesis epilogue: two), each inside its channel group (conv3x3_wino_stream's launch shape)
    const int per_cu = std::min((EPI == EPI_SYNTH || NT == 4 || (RES && NT == 2)) ? 2 : 3, (int)(160 * 1024 / lds));
    const int slots = std::max(1, cus * std::max(per_cu, 1) / q.groups);
    const int gx = std::min(q.total_tiles, slots);
    hipLaunchKernelGGL(kern, dim3(gx, q.groups), dim3(256), lds, s, q);
    return hipGetLastError();
}

}  // namespace lean
using namespace lean;

// the layers this file takes in bf16 mode: plain 3x3 convolutions from 32 px on, one source of a multiple of 16 channels up to 512, synthesis
// epilogue with direct statistics or decoder epilogue with the residual absent or one half-resolution tensor.  GSA_BF16_LEAN=0: conv3x3_mfma.
bool bf16_lean_applies(const ConvParams& p, int epi, bool sc) {
    static const bool enabled = !(getenv("GSA_BF16_LEAN") && atoi(getenv("GSA_BF16_LEAN")) == 0);
    if (!enabled || !p.bf16 || sc || p.up || p.C1 != 0 || p.src1 != nullptr) return false;
    if (p.H != p.W || p.H % 16 || p.H < 32 || p.Hs != p.H || p.C0 % 16 || p.C0 > 512 || p.Cout % 16) return false;
    if (epi == EPI_SYNTH) return p.partials != nullptr && p.noise != nullptr && p.fin_aff == nullptr && p.resid == nullptr;
    if (epi != EPI_DEC) return false;
    if (p.resid != nullptr && (p.resid_up != 1 || p.resid1 != nullptr)) return false;
    return true;
}

// output groups per workgroup: as many (4, 2, 1) as still leave two workgroups per CU of a 256-CU chip.
// (A resident-panel form with the activations prefetched three items ahead through three register sets was built and measured: bit-identical,
// cars bf16 batch 4 3344 -> 3242 pairs/s -- these layers are not waiting for their loads: DESIGN.md section 5.)
static int bf16_lean_nt(const ConvParams& p, int n) {
    const int groups = p.Cout / 16;
    const long tile_groups = (long)(p.H / 16) * (p.W / 16) * n * groups;
    for (int nt = 4; nt > 1; nt >>= 1)
        if (groups % nt == 0 && tile_groups / nt >= 512) return nt;
    return 1;
}

const char* bf16_lean_name(const ConvParams& p, int epi, int n) {
    static thread_local char buf[128];
    snprintf(buf, sizeof buf, "void gsa::lean::conv3x3_bf16_lean<%d, %s, %s, %d>(gsa::ConvParams)", epi, p.aff0 ? "true" : "false", p.resid ? "true" : "false",
             bf16_lean_nt(p, n));
    return buf;
}

template <int NT>
static hipError_t launch_bf16_nt(const ConvParams& p, int epi, int n, hipStream_t s) {
    if (epi == EPI_SYNTH) return p.aff0 ? launch_bf16_t<EPI_SYNTH, true, false, NT>(p, n, s) : launch_bf16_t<EPI_SYNTH, false, false, NT>(p, n, s);
    if (epi == EPI_DEC) {
        if (p.resid) return p.aff0 ? launch_bf16_t<EPI_DEC, true, true, NT>(p, n, s) : launch_bf16_t<EPI_DEC, false, true, NT>(p, n, s);
        return p.aff0 ? launch_bf16_t<EPI_DEC, true, false, NT>(p, n, s) : launch_bf16_t<EPI_DEC, false, false, NT>(p, n, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_bf16_lean(const ConvParams& p, int epi, int n, hipStream_t s) {
    switch (bf16_lean_nt(p, n)) {
        case 4: return launch_bf16_nt<4>(p, epi, n, s);
        case 2: return launch_bf16_nt<2>(p, epi, n, s);
        default: return launch_bf16_nt<1>(p, epi, n, s);
    }
}

}  // namespace gsa
