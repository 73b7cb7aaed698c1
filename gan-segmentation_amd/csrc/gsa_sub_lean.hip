// Lean form of the persistent stride-2 kernels (round 5): Deconvolution 4x4 s2 p1 (reference networks_stylegan.py:460-476) and nearest-x2 +
// conv3x3 in its sub-pixel form (networks_stylegan.py:308-315 + 354-457; networks_seg.py:7-46 with the fused 1x1 shortcut), every parity
// class in Winograd F(2x2,2x2) form -- the layers subpixel_res<..., WINO> takes (gsa_kernels.hip; canonical arithmetic: oracle/c/gsa_oracle.c
// deconv4x4s2_wino).  Same workgroup shape -- 512 threads = two halves of four waves (one wave per output parity class), each half walking
// its own tiles through its own double-buffered 10x10 input image, both reading ONE weight panel (resident, or a two-block ring when the panel
// does not fit) -- same transforms, same k-ordered MFMA chains, same epilogue operations: the same bits.  What is leaner:
//   * staging by 16-byte chunks on ALL 256 threads of a half (400 chunks = 2 rounds) instead of whole pixels on 100 of them; the AdaIN
//     coefficients of the sample sit in an LDS table as (A x 16 | B x 16) per block (two LDS reads and two packed fma per chunk instead of
//     16 table reads + 16 fma per pixel); global addresses are a wave-uniform base plus per-thread constants; border flags in one word;
//   * streamed weights go to the LDS ring by LDS-DMA (no staging registers, no LDS stores), waited for with a counted vmcnt before the
//     iteration's closing barrier;
//   * the chains of a tile's first block start from the inline constant 0 (no accumulator clearing); the epilogue is packed.
#include "gsa_kernels.h"
#include "gsa_dev.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace gsa {
namespace lean {

constexpr int SLW = 10, SRS = SLW * 16 + 4;      // 10x10 input image of a 16x16 output tile, row stride 164 floats (conflict-free 3x3 patch reads)
constexpr int SIMG = 10 * SRS;                   // floats per block image (6.4 KB)
constexpr int SSEG = 16 * 256;                   // the packed 4x4 kernel of (16 couts, 16 channels): [tap16][kq][16][cg]
constexpr int STS = 256;                         // the 1x1 shortcut block: [kq][16][cg]

struct STile { int n, y0, x0; };

// 2 wait states between packed adds (inline asm) and the MFMAs that read them; 11 between MFMA results and packed adds (see gsa_dev.h)
__device__ __forceinline__ void settle8(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f, f32x4& g, f32x4& h) {
    asm("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ void settle5(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e) { asm("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)); }
__device__ __forceinline__ void mfma_settle9(f32x4 (&a)[9]) {
    asm("s_nop 7\n\ts_nop 3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]));
}

// CNT (resident panel only): the two halves of the workgroup share nothing but the read-only panel, so nothing forces them through ONE
// eight-wave barrier per item.  Each half hands its image buffers over through two LDS counters of its own (arrivals after an item's
// staging stores / after its last LDS read) and the images form a ring of THREE: the buffer a wave fills was read two items ago, the buffer
// it reads was filled one item ago -- both conditions are a whole item old when they are checked, so a wave waits only for a partner that
// lags a full item behind, and the four waves of a half drift apart instead of meeting at the slowest one every item.
template <int NT, int EPI, bool SC, bool AFF, int KB, bool WST, bool CNT = false>
__global__ __launch_bounds__(512, 2) void subpixel_lean(ConvParams p) {
    static_assert(!WST || KB == 1, "streamed weights: one channel block per item");
    static_assert(!CNT || !WST, "counter hand-over: the resident-panel form");
    constexpr int NBUF = CNT ? 3 : 2;
    constexpr int NSTORES = 4 * NT + (SC ? NT : 0);      // global stores of one epilogue per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nblk0 = p.C0 >> 4, nblk = (p.C0 + p.C1) >> 4, nitem = nblk / KB;
    const int wblk = WST ? 2 : nblk;
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int g = WST ? (int)blockIdx.y : 0;
    float* const sW = smem;                                           // [wblk][NT][SSEG]
    float* const sS = sW + wblk * NT * SSEG;                           // SC: [wblk][NT][STS]
    float* const sAall = sS + (SC ? wblk * NT * STS : 0);
    float* const sA = sAall + half * (NBUF * KB * SIMG);               // this half: [NBUF][KB][SIMG]
    const int t8 = threadIdx.x & 255, lane = t8 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t8 >> 6);          // the wave's parity class
    const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // AFF: [2 (sample parity)][nblk0][A x 16 | B x 16] per half -- per WAVE with CNT (each wave fills its own copy: no hand-over between waves)
    float* const sC = sAall + 2 * NBUF * KB * SIMG + (CNT ? wave8 : half) * (2 * nblk0 * 32);
    unsigned* const sCnt = reinterpret_cast<unsigned*>(sAall + 2 * NBUF * KB * SIMG + (AFF ? (CNT ? 8 : 2) * 2 * nblk0 * 32 : 0)) + half * 8;   // CNT: {written[4], read[4]}
    const int py = wave >> 1, px = wave & 1;
    const int i16 = lane & 15, kq = lane >> 4, part = t8 & 3;
    const int H = p.H, W = p.W, Hs = p.Hs, Ws = p.Ws, COUT = p.Cout;

    // contiguous range of tiles (n, ty, tx), split between the halves
    const int per = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    int w_begin = xcd_block(blockIdx.x, gridDim.x) * per;
    int w_end = min(p.total_tiles, w_begin + per);
    if (w_begin >= w_end) return;                  // whole workgroup
    const int first_half = (w_end - w_begin + 1) >> 1;
    const int iters = first_half * nitem;          // loop trips of the longer half: both halves run the same barriers
    if (half == 0) w_end = w_begin + first_half; else w_begin += first_half;
    const int total_items = max(w_end - w_begin, 0) * nitem;

    // ---- staging: chunk q = t8 + 256 k = (input-tile pixel q >> 2, channels 4 * part ..), byte offsets per source from the halo origin
    unsigned off0[2], off1[2], eflags = 0;
    int l_off[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pix = (t8 + 256 * k) >> 2;
        const bool real = pix < 100;
        const int ly = real ? pix / SLW : 1, lx = real ? pix % SLW : 1;
        off0[k] = (unsigned)(((ly * Ws + lx) * p.C0 + part * 4) * 4);
        off1[k] = (unsigned)(((ly * Ws + lx) * p.C1 + part * 4) * 4);
        l_off[k] = real ? ly * SRS + lx * 16 + part * 4 : -1;
        eflags |= (unsigned)((ly == 0 ? 1 : 0) | (ly == 9 ? 2 : 0) | (lx == 0 ? 4 : 0) | (lx == 9 ? 8 : 0)) << (4 * k);
    }
    const unsigned safe0 = (unsigned)((((Ws + 1) * p.C0) + part * 4) * 4), safe1 = (unsigned)((((Ws + 1) * p.C1) + part * 4) * 4);
    auto edge_code = [&](const STile& t) { return (t.y0 == 0 ? 1 : 0) | (t.y0 + 16 == H ? 2 : 0) | (t.x0 == 0 ? 4 : 0) | (t.x0 + 16 == W ? 8 : 0); };
    auto advance = [&](STile& t) { t.x0 += 16; if (t.x0 == W) { t.x0 = 0; t.y0 += 16; if (t.y0 == H) { t.y0 = 0; t.n += 1; } } };

    f32x4 ra[KB][2];
    auto load_item = [&](const STile& t, int e, int ci) {
        const long pix0 = ((long)(t.n * Hs + (t.y0 >> 1)) * Ws + (t.x0 >> 1)) - (Ws + 1);      // halo origin (pixels)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int cb = ci * KB + kb;
            const bool first = cb < nblk0;
            const char* hb = first ? reinterpret_cast<const char*>(p.src0) + (pix0 * p.C0 + cb * 16) * 4
                                   : reinterpret_cast<const char*>(p.src1) + (pix0 * p.C1 + (cb - nblk0) * 16) * 4;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned real_off = first ? off0[k] : off1[k], safe = first ? safe0 : safe1;
                const unsigned bad = (eflags >> (4 * k)) & (unsigned)e;
                ra[kb][k] = *reinterpret_cast<const f32x4*>(hb + (real_off + (bad ? safe - real_off : 0u)));
            }
        }
    };
    // AdaIN coefficients of sample n -> this half's table slot n & 1 (filled one item ahead of their first use: the barrier that closes the
    // iteration publishes them; the slot of the previous sample is still being read meanwhile)
    auto fill_coefficients = [&](int n) {
        float* tab = sC + (n & 1) * (nblk0 * 32);
        for (int e = CNT ? lane : t8; e < p.C0; e += CNT ? 64 : 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.aff0 + (size_t)n * p.C0 + e);      // (mean, A, B, -)
            tab[(e >> 4) * 32 + (e & 15)] = a[1];
            tab[(e >> 4) * 32 + 16 + (e & 15)] = a[2];
        }
    };
    auto write_item = [&](const STile& t, int e, int ci, int buf) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int cb = ci * KB + kb;
            float* img = sA + (buf * KB + kb) * SIMG;
            f32x4 kA = z, kB = z;
            const bool aff = AFF && cb < nblk0;
            if (aff) {
                const float* tab = sC + (t.n & 1) * (nblk0 * 32) + cb * 32 + part * 4;
                kA = *reinterpret_cast<const f32x4*>(tab);
                kB = *reinterpret_cast<const f32x4*>(tab + 16);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4 v = aff ? fma4(ra[kb][k], kA, kB) : ra[kb][k];
                if (e && ((eflags >> (4 * k)) & (unsigned)e)) v = z;
                if (k == 0 || l_off[1] >= 0) *reinterpret_cast<f32x4*>(img + l_off[k]) = v;
            }
        }
    };

    // ---- weights: the whole panel (and the shortcut's) -> LDS once, or (WST) one block per iteration by LDS-DMA into the two-slot ring
    auto dma_block = [&](int cbk, int slot) {
#pragma unroll
        for (int j = 0; j < 2 * NT; ++j) {
            const int pc = wave8 + 8 * j, q = pc >> 4, r = pc & 15;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.wpk + ((size_t)(g * NT + q) * nblk + cbk) * SSEG + r * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(sW + (slot * NT + q) * SSEG + r * 256), 16, 0, 0);
        }
        if (SC && wave8 < NT)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.wsc + ((size_t)(g * NT + wave8) * nblk + cbk) * STS + lane * 4),
                                             (__attribute__((address_space(3))) void*)(sS + (slot * NT + wave8) * STS), 16, 0, 0);
    };
    if (WST) {
        dma_block(0, 0);
    } else {
        for (int i = threadIdx.x; i < nblk * NT * (SSEG / 4); i += 512) {
            const int cbk = i / (NT * (SSEG / 4)), r = i % (NT * (SSEG / 4)), q = r / (SSEG / 4), o = r % (SSEG / 4);
            reinterpret_cast<f32x4*>(sW)[i] = reinterpret_cast<const f32x4*>(p.wpk + ((size_t)(g * NT + q) * nblk + cbk) * SSEG)[o];
        }
        if (SC)
            for (int i = threadIdx.x; i < nblk * NT * (STS / 4); i += 512) {
                const int cbk = i / (NT * (STS / 4)), r = i % (NT * (STS / 4)), q = r / (STS / 4), o = r % (STS / 4);
                reinterpret_cast<f32x4*>(sS)[i] = reinterpret_cast<const f32x4*>(p.wsc + ((size_t)(g * NT + q) * nblk + cbk) * STS)[o];
            }
    }

    // ---- operands: lane (i16, kq) owns Winograd tile (wty, wtx) of the 4x4 grid of 2x2-input blocks; its 3x3 patch is shifted by the class
    const int wty = (i16 & 1) * 2 + ((i16 >> 1) & 1), wtx = ((i16 >> 2) & 1) * 2 + ((i16 >> 3) & 1);
    const int pbase = (2 * wty + py) * SRS + (2 * wtx + px) * 16 + kq * 4;
    const int bbase = (kq * 16 + i16) * 4;
    const int asc = ((wave >> 1) * 4 + (i16 >> 2) + 1) * SRS + ((wave & 1) * 4 + (i16 & 3) + 1) * 16 + kq * 4;      // shortcut: this wave's 4x4 input pixels
    // taps of the class filter g[a][b] = Wd[3 - py - 2a][3 - px - 2b]
    const int t00 = ((3 - py) * 4 + (3 - px)) * 256, t01 = ((3 - py) * 4 + (1 - px)) * 256;
    const int t10 = ((1 - py) * 4 + (3 - px)) * 256, t11 = ((1 - py) * 4 + (1 - px)) * 256;
    f32x4 acc[9][NT], accs[SC ? NT : 1];
    auto mfma_block = [&](auto first_tag, const float* a_img, int wslot) {
        constexpr bool FIRST = decltype(first_tag)::value;      // the tile's first block: every chain starts from the inline constant 0
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 V[9];
        {
            const float* ap = a_img + pbase;
            f32x4 d[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) d[r][c] = *reinterpret_cast<const f32x4*>(ap + r * SRS + c * 16);
            // rows t0 = d0 - d1, t1 = d1, t2 = d2 - d1; then the same three forms along the columns
#pragma unroll
            for (int c = 0; c < 3; ++c) { d[0][c] = sub4(d[0][c], d[1][c]); d[2][c] = sub4(d[2][c], d[1][c]); }
#pragma unroll
            for (int r = 0; r < 3; ++r) { V[3 * r] = sub4(d[r][0], d[r][1]); V[3 * r + 1] = d[r][1]; V[3 * r + 2] = sub4(d[r][2], d[r][1]); }
            settle8(V[0], V[1], V[2], V[3], V[5], V[6], V[7], V[8]);       // V[4] = d[1][1] comes straight from LDS
        }
        const float* bp = sW + wslot * (NT * SSEG) + bbase;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 U[9];
            U[0] = *reinterpret_cast<const f32x4*>(bp + nt * SSEG + t00);
            U[2] = *reinterpret_cast<const f32x4*>(bp + nt * SSEG + t01);
            U[6] = *reinterpret_cast<const f32x4*>(bp + nt * SSEG + t10);
            U[8] = *reinterpret_cast<const f32x4*>(bp + nt * SSEG + t11);
            U[1] = add4(U[0], U[2]);
            U[7] = add4(U[6], U[8]);
            U[3] = add4(U[0], U[6]);
            U[5] = add4(U[2], U[8]);
            U[4] = add4(U[1], U[7]);
            settle5(U[1], U[3], U[4], U[5], U[7]);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int f = 0; f < 9; ++f)
                    acc[f][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[f][cg], V[f][cg], FIRST && cg == 0 ? zero : acc[f][nt], 0, 0, 0);
        }
        if constexpr (SC) {
            const f32x4 as = *reinterpret_cast<const f32x4*>(a_img + asc);
            f32x4 bs[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bs[nt] = *reinterpret_cast<const f32x4*>(sS + (wslot * NT + nt) * STS + bbase);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    accs[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bs[nt][cg], as[cg], FIRST && cg == 0 ? zero : accs[nt], 0, 0, 0);
        }
    };

    // ---- epilogue: a lane holds four consecutive output channels (4 kq ..) of its tile's 2x2 class outputs
    const int cq4 = 4 * kq;
    const unsigned lane_out = (unsigned)(((4 * wty * W + 4 * wtx) * COUT + cq4) * 4);
    const unsigned lane_sc = (unsigned)((((i16 >> 2) * Ws + (i16 & 3)) * COUT + cq4) * 4);
    f32x4 e2[NT], e3[NT], scb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = g * 16 * NT + nt * 16 + cq4;
        e2[nt] = e3[nt] = scb[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_DEC) { e2[nt] = *reinterpret_cast<const f32x4*>(p.bn_s + co); e3[nt] = *reinterpret_cast<const f32x4*>(p.bn_beta + co); }
        if (SC) scb[nt] = *reinterpret_cast<const f32x4*>(p.sc_bias + co);
    }
    const f32x2 k02 = {0.2f, 0.2f};
    auto epilogue = [&](const STile& t) {
        char* ob = reinterpret_cast<char*>(p.out) + (((long)(t.n * H + t.y0 + py) * W + t.x0 + px) * COUT + g * 16 * NT) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // output transform Y = A^T M A: rows s0 = m0 + m1, s1 = m1 + m2, then along the columns
            f32x4 m[9];
#pragma unroll
            for (int f = 0; f < 9; ++f) m[f] = acc[f][nt];
            mfma_settle9(m);
            f32x4 s0[3], s1[3], y[2][2];
#pragma unroll
            for (int j = 0; j < 3; ++j) { s0[j] = add4(m[j], m[3 + j]); s1[j] = add4(m[3 + j], m[6 + j]); }
            y[0][0] = add4(s0[0], s0[1]); y[0][1] = add4(s0[1], s0[2]);
            y[1][0] = add4(s1[0], s1[1]); y[1][1] = add4(s1[1], s1[2]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 v = y[a][b];
                    if (EPI == EPI_DEC) v = lrelu4(fma4(v, e2[nt], e3[nt]), k02);
                    *reinterpret_cast<f32x4*>(ob + ((long)(2 * a) * W + 2 * b) * COUT * 4 + nt * 64 + lane_out) = v;
                }
            if constexpr (SC) {
                char* sb = reinterpret_cast<char*>(p.out_sc) +
                           (((long)(t.n * Hs + (t.y0 >> 1) + (wave >> 1) * 4) * Ws + (t.x0 >> 1) + (wave & 1) * 4) * COUT + g * 16 * NT + nt * 16) * 4;
                f32x4 sv[1] = {accs[nt]};
                asm("s_nop 7\n\ts_nop 3" : "+v"(sv[0]));
                *reinterpret_cast<f32x4*>(sb + lane_sc) = add4(sv[0], scb[nt]);
            }
        }
    };

    // ---- items (tile, item of KB blocks): item `it` is multiplied out of LDS buffer it & 1, item it+1 sits in ra, item it+2 is being loaded
    STile tc, tr;
    {
        const int w0 = min(w_begin, p.total_tiles - 1);
        const int tx = w0 % p.tiles_x, r = w0 / p.tiles_x;
        tc.x0 = tx * 16; tc.y0 = (r % p.tiles_y) * 16; tc.n = r / p.tiles_y;
    }
    int ec = edge_code(tc), er, cic = 0, cir = 0, n_tab = tc.n;
    auto next_item = [&](STile& t, int& e, int& ci) {
        if (++ci == nitem) { ci = 0; advance(t); e = edge_code(t); }
    };
    if (AFF) fill_coefficients(tc.n);
    if (total_items > 0) load_item(tc, ec, 0);
    if (WST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ring slot 0
    if (CNT && t8 < 8) sCnt[t8] = t8 < 4 ? 1u : 0u;                   // item 0 counts as staged (the barrier below publishes it)
    __syncthreads();                                                  // weight panel and coefficient table visible
    if (total_items > 0) write_item(tc, ec, 0, 0);
    tr = tc; er = ec;
    if (total_items > 1) {
        next_item(tr, er, cir);
        if (AFF && tr.n != n_tab) { fill_coefficients(tr.n); n_tab = tr.n; }
        load_item(tr, er, cir);
    }
    __syncthreads();
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, sw = 0, sl = 0, sm = 0, se = 0, sb = 0;      // diagnostic build only (make stamp)
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)k5; (void)sw; (void)sl; (void)sm; (void)se; (void)sb;
    if constexpr (CNT) {
        // one progress word per wave and kind (a sum over the waves would let a wave that runs ahead stand in for one that lags):
        // written[w] = items wave w has staged, read[w] = items it has finished reading; a condition holds when the slowest wave meets it
        unsigned n_written = 1, n_read = 0;
        auto publish = [&](unsigned* c, unsigned v) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's LDS stores / reads are done before it says so
            if (lane == 0) __hip_atomic_store(c + wave, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto wait_for = [&](unsigned* c, int target) {
            for (;;) {
                const u32x4 v = *reinterpret_cast<volatile u32x4*>(c);
                const int lo = (int)min(min(v[0], v[1]), min(v[2], v[3]));
                if (lo >= target) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        for (int it = 0; it < total_items; ++it) {
            TICK(k0);
            if (it + 1 < total_items) {
                wait_for(sCnt + 4, it - 1);                          // items <= it - 2 are read by everybody: buffer (it + 1) % 3 is free
                write_item(tr, er, cir, (it + 1) % 3);
                publish(sCnt, ++n_written);
            }
            TICK(k1);
            STile t2 = tr; int e2c = er, ci2 = cir;
            if (it + 2 < total_items) {
                next_item(t2, e2c, ci2);
                if (AFF && t2.n != n_tab) { fill_coefficients(t2.n); n_tab = t2.n; }
                load_item(t2, e2c, ci2);
            }
            TICK(k2);
            wait_for(sCnt, it + 1);                                  // item it is complete in its buffer (it was staged one item ago)
            const float* a_buf = sA + ((it % 3) * KB) * SIMG;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int wslot = cic * KB + kb;
                if (kb == 0 && cic == 0) mfma_block(std::true_type{}, a_buf + kb * SIMG, wslot);
                else mfma_block(std::false_type{}, a_buf + kb * SIMG, wslot);
            }
            publish(sCnt + 4, ++n_read);
            TICK(k3);
            if (cic == nitem - 1) epilogue(tc);
            TICK(k4);
            TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4);
            tc = tr; ec = er; cic = cir; tr = t2; er = e2c; cir = ci2;
        }
        TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
        TFLUSH(12, (unsigned long long)total_items); TFLUSH(15, 1ull);
        return;
    }
    for (int it = 0; it < iters; ++it) {
        TICK(k0);
        if (WST && it + 1 < iters) dma_block((it + 1) % nblk, (it + 1) & 1);      // the slot iteration it - 1 read: every wave is past its closing barrier
        bool stored = false, loaded = false;
        if (it < total_items) {
            if (it + 1 < total_items) write_item(tr, er, cir, (it + 1) & 1);
            TICK(k1);
            STile t2 = tr; int e2c = er, ci2 = cir;
            if (it + 2 < total_items) {
                loaded = true;
                next_item(t2, e2c, ci2);
                if (AFF && t2.n != n_tab) { fill_coefficients(t2.n); n_tab = t2.n; }
                load_item(t2, e2c, ci2);
            }
            TICK(k2);
            const float* a_buf = sA + ((it & 1) * KB) * SIMG;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int wslot = WST ? (it & 1) : cic * KB + kb;
                if (kb == 0 && cic == 0) mfma_block(std::true_type{}, a_buf + kb * SIMG, wslot);
                else mfma_block(std::false_type{}, a_buf + kb * SIMG, wslot);
            }
            TICK(k3);
            if (cic == nitem - 1) { epilogue(tc); stored = true; }
            TICK(k4);
            tc = tr; ec = er; cic = cir; tr = t2; er = e2c; cir = ci2;
        }
        if (WST) {
            // this wave's DMA pieces are older than the loads of item it + 2 and the epilogue's stores (in-order completion): with at most that
            // many operations outstanding the pieces are in LDS, the younger operations stay in flight
            // (an iteration that issued fewer of them waits for correspondingly more)
            if (stored && loaded) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * KB + NSTORES) : "memory");
            else if (loaded) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * KB) : "memory");
            else if (stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTORES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        TICK(k5);
        if (it < total_items) { TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4); TSUM(sb, k4, k5); }
    }
    TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
    TFLUSH(12, (unsigned long long)total_items); TFLUSH(15, 1ull);
}

constexpr int kMaxDev = 64;
struct SubState { bool attr_done = false; };
static std::mutex g_sub_mu;

// LDS bytes of an instantiation (cnt: the three-buffer counter form)
size_t sub_lds(const ConvParams& q, int nt, bool sc, bool aff, int kb, bool wst, bool cnt) {
    const int nblk0 = q.C0 / 16, nblk = (q.C0 + q.C1) / 16, wblk = wst ? 2 : nblk;
    return sizeof(float) * ((size_t)wblk * nt * SSEG + (sc ? (size_t)wblk * nt * STS : 0) + 2 * (cnt ? 3 : 2) * kb * SIMG +
                            (aff ? (cnt ? 8 : 2) * 2 * nblk0 * 32 : 0)) + (cnt ? 64 : 0);
}

// GSA_SUB_CNT=1 (experiments build): per-half progress words with a ring of three images instead of the eight-wave barrier (resident
// panel; same bits).  Measured 2.3 % slower over the whole step (1290 against 1320 pairs/s on one box): see DESIGN.md section 5.
bool sub_counters(const ConvParams& q, int nt, bool sc, int kb, bool wst) {
#if GSA_EXPERIMENTS
    static const bool cnt_on = getenv("GSA_SUB_CNT") && atoi(getenv("GSA_SUB_CNT")) != 0;
    return cnt_on && !wst && sub_lds(q, nt, sc, q.aff0 != nullptr, kb, false, true) <= 160 * 1024;
#else
    (void)q; (void)nt; (void)sc; (void)kb; (void)wst;
    return false;
#endif
}

template <int NT, int EPI, bool SC, bool AFF, int KB, bool WST, bool CNT = false>
hipError_t launch_k(const ConvParams& q, dim3 grid, hipStream_t s) {
    static SubState st[kMaxDev];
    auto kern = subpixel_lean<NT, EPI, SC, AFF, KB, WST, CNT>;
    const size_t lds = sub_lds(q, NT, SC, AFF, KB, WST, CNT);
    if (lds > 160 * 1024 || q.device < 0 || q.device >= kMaxDev) return hipErrorInvalidValue;
    {
        std::lock_guard<std::mutex> lk(g_sub_mu);
        if (!st[q.device].attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            st[q.device].attr_done = true;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, q);
    return hipGetLastError();
}

template <int NT, int EPI, bool SC, bool AFF>
hipError_t launch_shape(const ConvParams& q, int kb, bool wst, dim3 grid, hipStream_t s) {
    if (wst) return launch_k<NT, EPI, SC, AFF, 1, true>(q, grid, s);
#if GSA_EXPERIMENTS
    if (sub_counters(q, NT, SC, kb, false))
        return kb == 2 ? launch_k<NT, EPI, SC, AFF, 2, false, true>(q, grid, s) : launch_k<NT, EPI, SC, AFF, 1, false, true>(q, grid, s);
#endif
    return kb == 2 ? launch_k<NT, EPI, SC, AFF, 2, false>(q, grid, s) : launch_k<NT, EPI, SC, AFF, 1, false>(q, grid, s);
}

}  // namespace lean

// the launches of subpixel_res<..., WINO> (fp32) this file takes: 16 or 32 output channels per workgroup, epilogue RAW or DEC, AdaIN only on
// a single source.  GSA_SUB_LEAN=0 keeps subpixel_res (same bits).
bool subpixel_lean_applies(const ConvParams& p, int nt, int epi, bool sc, int kb, bool wst) {
    static const bool enabled = !(getenv("GSA_SUB_LEAN") && atoi(getenv("GSA_SUB_LEAN")) == 0);
    if (!enabled || p.bf16 || (nt != 1 && nt != 2) || (epi != EPI_RAW && epi != EPI_DEC) || (sc && epi != EPI_DEC)) return false;
    if (p.aff0 != nullptr && p.C1 != 0) return false;
    return lean::sub_lds(p, nt, sc, p.aff0 != nullptr, kb, wst, false) <= 160 * 1024;
}

const char* subpixel_lean_name(const ConvParams& p, int nt, int epi, bool sc, int kb, bool wst) {
    static thread_local char buf[112];
    snprintf(buf, sizeof buf, "void gsa::lean::subpixel_lean<%d, %d, %s, %s, %d, %s, %s>(gsa::ConvParams)", nt, epi, sc ? "true" : "false",
             p.aff0 ? "true" : "false", kb, wst ? "true" : "false", lean::sub_counters(p, nt, sc, kb, wst) ? "true" : "false");
    return buf;
}

// q: the ConvParams the subpixel_res launcher prepared (tiles_x, tiles_y, groups, total_tiles filled), grid: its grid
hipError_t launch_subpixel_lean(const ConvParams& q, int nt, int epi, bool sc, int kb, bool wst, dim3 grid, hipStream_t s) {
    using namespace lean;
    const bool aff = q.aff0 != nullptr;
#define GSA_SL(NT, EPI, SC) \
    if (nt == NT && epi == EPI && sc == SC) return aff ? launch_shape<NT, EPI, SC, true>(q, kb, wst, grid, s) : launch_shape<NT, EPI, SC, false>(q, kb, wst, grid, s);
    GSA_SL(1, EPI_RAW, false) GSA_SL(2, EPI_RAW, false) GSA_SL(1, EPI_DEC, false) GSA_SL(2, EPI_DEC, false) GSA_SL(1, EPI_DEC, true) GSA_SL(2, EPI_DEC, true)
#undef GSA_SL
    return hipErrorInvalidValue;
}

}  // namespace gsa
