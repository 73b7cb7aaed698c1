// HIP kernels of the `generate` hot path for gfx950 (MI355X, CDNA4).
//
// Layout: activations are NHWC fp32 in HBM.  Instance-norm + style (AdaIN) is never
// materialised: a producer stores its post-LeakyReLU tensor plus fixed-point statistics,
// a tiny finalize kernel turns them into per-(sample,channel) coefficients (mean, A, B) and
// every consumer applies fmaf(x - mean, A, B) while staging its input tile into LDS.
//
// All convolutions are implicit GEMMs on the exact-fp32 matrix cores
// (v_mfma_f32_16x16x4_f32): M = a 4x4 patch of output pixels, N = 16 output channels,
// K = 4 input channels per instruction.  The MFMA is bitwise a k-ordered fmaf chain, so with
// the K order fixed to (16-channel block, tap, channel) the kernels reproduce the canonical
// arithmetic of oracle/c/gsa_oracle.c bit for bit (DESIGN.md "Canonical arithmetic").
//
// Reference operators replaced (reference file:line):
//   conv3x3_mfma    Conv2DW 3x3 (+UpSample nearest)   networks_stylegan.py:354-457, 308-315
//                   nn.Conv2D+BatchNorm+LeakyReLU       networks_seg.py:14-46, 68-76
//   deconv4x4_mfma  Conv2DTransposeW k4 s2 p1           networks_stylegan.py:460-476
//   post_kernel     Blur, AddNoise, Bias, LeakyReLU     networks_stylegan.py:200-236, 267-305, 534-545
//   finalize_kernel InstanceNorm + AdaIN style          networks_stylegan.py:239-264
//   dense / styles  PixelNorm, DenseW, lerp             networks_stylegan.py:128-139, 158-163, 479-524, 558-565
//   torgb_kernel    toRGB + _transform_gan_back         networks_stylegan.py:118-126; image_generator.py:76-84
//   final_conv      final Conv2D + argmax               networks_seg.py:91-92; seg_solver.py:326
#include "gsa_kernels.h"
#include "gsa_dev.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>

// wave priority during the MFMA phase of conv3x3_mfma (the wave multiplying wins the SIMD's issue slot over the wave
// staging): +0.5-1 % on the step in A/B runs; the same in the stride-2 kernels measured -0.5 % and is not used
#ifndef GSA_MFMA_PRIO
#define GSA_MFMA_PRIO 2
#endif
#ifndef GSA_OPERAND_PIPELINE
#define GSA_OPERAND_PIPELINE 1
#endif
#ifndef GSA_NO_XCD_REMAP
#define GSA_NO_XCD_REMAP 0
#endif
// double-buffered (prefetch two items ahead, one barrier per item) form also for the 8x8 / 4x4 tiles of the 4^2-16^2 layers:
// their 32 channel blocks were 32 dependent load -> LDS -> barrier rounds with ONE block in flight per workgroup
#ifndef GSA_POST_OCC
#define GSA_POST_OCC 2      // waves per SIMD the blur pass is compiled for (3 = 168 registers: 260-370 bytes of scratch, g.512.post_1 0.130 -> 0.343 ms)
#endif
#ifndef GSA_WINO_SOA
#define GSA_WINO_SOA 0
#endif
#ifndef GSA_DB_SMALL
#define GSA_DB_SMALL 1
#endif
#ifndef GSA_EXPERIMENTS
#define GSA_EXPERIMENTS 0      // 1 (`make experiments` -> libgsa_hip_exp.so): also build the measured-slower kernels of round 4 behind their switches
#endif


namespace gsa {

// ------------------------------------------------------------------------------------------
// Shared pieces of the two MFMA convolutions.
//
// LDS image of an input tile (1-pixel halo): [row][pixel][16 channels] in natural channel order.  ONE ds_read_b128
// per lane (pixel i, k slot kq) at chunk kq yields channels 4kq..4kq+3 = the A operands of 4 consecutive MFMAs:
// MFMA j of a (tap, block) multiplies the channels {j, 4+j, 8+j, 12+j} (k slot kq <-> channel 4kq+j), so the
// canonical order of a 16-channel block is 0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15 (the oracle walks the
// same order).  A row stride = 8 (mod 16) floats makes those reads bank-conflict free.  Weights are packed in HBM
// per (16 output channels, 16-channel block) as [tap][kq][n][j] (channel 4kq+j) and copied verbatim.
//
// Staging is split (load to registers early / write to LDS late) so that the global loads of
// block cb+1 are in flight while the MFMAs of block cb run.

struct TilePixel {   // one staged pixel of this thread
    int pix;         // pixel index into the source tensors ((n*Hs+sy)*Ws+sx), or -1: zero padding
    int lds;         // float offset in the LDS image, or -1: this thread stages nothing
};

// ---- activation tensors: fp32, or (bf16 mode, BF = true) bf16 in the same NHWC order ------------------------------------
// In bf16 mode every activation tensor that lives in HBM is bf16 (round 2; round 1 kept fp32 there and only rounded the MFMA
// operands): half the bytes of kernels that are HBM-bound in that mode.  Pointers stay typed `float*` in the parameter
// structs; the element index is the same in both modes, the element size is not.  Arithmetic stays fp32: a consumer widens
// (exact), applies AdaIN in fp32 and rounds to bf16 (RNE) once more for the MFMA operand; statistics are taken from the fp32
// values before the producer rounds them for storage.
// Unconditional loads (padding / idle threads read pixel 0 and discard it): no branch sits
// between a global load and its use, so the compiler keeps every load of a block in flight.
// BF: the 16 channels of the block are 32 bytes; they travel as raw bits in v[0], v[1].
template <bool BF = false>
__device__ __forceinline__ void load_pixel(f32x4 (&v)[4], const float* src, int Cs, int coff, const TilePixel& tp) {
    const int pix = tp.pix >= 0 ? tp.pix : 0;
    if constexpr (BF) {
        const f32x4* ptr = reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(src) + (size_t)pix * Cs + coff);
        v[0] = ptr[0]; v[1] = ptr[1];
    } else {
        const f32x4* ptr = reinterpret_cast<const f32x4*>(src + (size_t)pix * Cs + coff);
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) v[cg] = ptr[cg];
    }
}

// AdaIN on the fly (zero padding stays zero), then the transposed 4x4 store.
// HAS_AFF is wave-uniform; the coefficients of the sample were copied to LDS (saff, one float4
// per input channel) at kernel start, so the write phase issues no global load.
// BF (bf16 MFMA mode): the 16 values are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) AFTER the fp32 AdaIN and
// stored in natural channel order, 32 bytes per pixel: the k slot of a 16x16x16 MFMA is 4 consecutive channels.
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// MASK = false: the caller knows every staged pixel of the tile lies inside the image (interior tile), so the
// 16 zero-padding selects per pixel are not emitted.
template <bool HAS_AFF, bool BF = false, bool MASK = true>
__device__ __forceinline__ void store_pixel(float* sA, f32x4 (&v)[4], const float4* saff, const TilePixel& tp) {
    const bool inside = !MASK || tp.pix >= 0;
    if constexpr (BF && !HAS_AFF) {      // bf16 tensor, no AdaIN: the LDS image IS the tensor's 32 bytes (zero outside the image)
        if (tp.lds >= 0) {
            f32x4* dst = reinterpret_cast<f32x4*>(sA + tp.lds);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            dst[0] = inside ? v[0] : z;
            dst[1] = inside ? v[1] : z;
        }
        return;
    }
    float f[16];
    if constexpr (BF) {                  // widen the 16 bf16 channels (exact)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned u = __float_as_uint(v[k >> 2][k & 3]);
            f[2 * k] = bf16_lo(u); f[2 * k + 1] = bf16_hi(u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = v[k >> 2][k & 3];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float t = f[c];
        if (HAS_AFF) {
            const float4 a = saff[c];    // (mean, A, B, -)
            t = fmaf(t, a.y, a.z);
        }
        f[c] = inside ? t : 0.0f;
    }
    if (BF) {
        if (tp.lds >= 0) {
            u32x4* dst = reinterpret_cast<u32x4*>(sA + tp.lds);
            dst[0] = u32x4{pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7])};
            dst[1] = u32x4{pack_bf16(f[8], f[9]), pack_bf16(f[10], f[11]), pack_bf16(f[12], f[13]), pack_bf16(f[14], f[15])};
        }
        return;
    }
    if (tp.lds >= 0) {      // natural channel order: the 16-byte chunk kq holds channels 4kq..4kq+3
        f32x4* dst = reinterpret_cast<f32x4*>(sA + tp.lds);
        dst[0] = f32x4{f[0], f[1], f[2], f[3]};
        dst[1] = f32x4{f[4], f[5], f[6], f[7]};
        dst[2] = f32x4{f[8], f[9], f[10], f[11]};
        dst[3] = f32x4{f[12], f[13], f[14], f[15]};
    }
}

// store_pixel with the block's AdaIN coefficients as two structure-of-arrays register sets (A and B of channels 4j..4j+3 in cA[j] /
// cB[j], read once per ITEM from a [2][16]-float LDS table): one packed fma per two channels and no per-channel table read -- the
// array-of-structures table of store_pixel costs a 16-byte LDS read and a scalar fma per channel (same values, same bits)
template <bool MASK>
__device__ __forceinline__ void store_pixel_soa(float* sA, const f32x4 (&v)[4], const f32x4 (&cA)[4], const f32x4 (&cB)[4], const TilePixel& tp) {
    const bool inside = !MASK || tp.pix >= 0;
    if (tp.lds < 0) return;
    f32x4* dst = reinterpret_cast<f32x4*>(sA + tp.lds);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = __builtin_elementwise_fma(v[j], cA[j], cB[j]);
        dst[j] = inside ? t : z;
    }
}

// ---- staging by 16-byte chunks --------------------------------------------------------------------------
// One chunk = 4 consecutive channels (part*4..+3 of the item's 16-channel block) of one tile pixel: ONE global load,
// 8 vector-ALU instructions of AdaIN, ONE LDS store.  Thread t of a workgroup stages the chunks q = t + k*NTHR
// (pixel q >> 2, part q & 3 = t & 3 since NTHR % 4 == 0): a tile of P pixels costs ceil(4P / NTHR) rounds instead of
// ceil(P / NTHR) rounds of four loads / stores each with most threads idle in the last one (324 pixels on 256 threads:
// 6 chunk rounds instead of 8; 100 pixels: 2 instead of 4).  A thread's four channels never change, so its AdaIN
// coefficients are four registers loaded with the item (64 contiguous bytes of Aff) -- no LDS table, no table barrier.
struct Chunk {
    int pix;         // pixel index into the source tensor ((n*Hs+sy)*Ws+sx), or -1: zero padding
    int lds;         // 4-byte slot offset of the chunk in the LDS image, or -1: this thread stages nothing
};

__device__ __forceinline__ f32x4 load_chunk(const float* src, int Cs, int coff, const Chunk& ch) {
    const int pix = ch.pix >= 0 ? ch.pix : 0;      // unconditional load (padding reads pixel 0 and discards it)
    return *reinterpret_cast<const f32x4*>(src + (size_t)pix * Cs + coff);
}

// the four Aff entries (mean, A, B, -) of channels c0..c0+3 of sample n
__device__ __forceinline__ void load_aff4(f32x4 (&aff)[4], const Aff* table, size_t first) {
    const f32x4* ptr = reinterpret_cast<const f32x4*>(table + first);
#pragma unroll
    for (int c = 0; c < 4; ++c) aff[c] = ptr[c];
}

template <bool HAS_AFF, bool BF, bool MASK>
__device__ __forceinline__ void store_chunk(float* sA, const f32x4& v, const f32x4 (&aff)[4], const Chunk& ch) {
    const bool inside = !MASK || ch.pix >= 0;
    float f[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float t = f[c];
        if (HAS_AFF) t = fmaf(t, aff[c][1], aff[c][2]);
        f[c] = inside ? t : 0.0f;
    }
    if (ch.lds < 0) return;
    if (BF) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(sA + ch.lds) = u32x2{pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3])};
    } else {
        *reinterpret_cast<f32x4*>(sA + ch.lds) = f32x4{f[0], f[1], f[2], f[3]};
    }
}

// ------------------------------------------------------------------------------------------
// conv3x3 (pad 1) as implicit GEMM on v_mfma_f32_16x16x4_f32.
//
// Workgroup = WM*WN waves; one work tile = TH x TW output pixels x COUT_T = 16*NT*WN channels
// of one sample.  Workgroups are PERSISTENT: each walks a contiguous range of work tiles and
// software-pipelines across (tile, 16-channel block) items -- the global loads of item i+1 are
// in flight during the MFMAs and the epilogue stores of item i.  (One workgroup per tile made
// whole generations of workgroups alternate between "everyone loads" and "everyone computes":
// in-kernel stamps showed 57k-cycle load phases and 7x-stretched MFMA phases on the 16-channel
// layers.)
struct WorkTile { int n, g, y0, x0, row; };   // row = tile index inside the image

//
// BF = bf16 MFMA mode (BASELINE config 5): operands are rounded to bf16 when they are staged (inputs after
// the fp32 AdaIN, weights on the host), one v_mfma_f32_16x16x16_bf16 per (tap, 16-channel block) replaces four
// fp32 MFMAs, accumulation / epilogue / statistics stay fp32.  LDS offsets are in 4-byte slots in both modes:
// a staged pixel is PX slots, a (tap, 16 couts, 16 channels) weight chunk TS slots.
template <int TH, int TW, int WM, int WN, int NT, int EPI, bool SC, bool BF>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv3x3_mfma(ConvParams p) {
    constexpr int NW = WM * WN, NTHR = 64 * NW;
    constexpr int PW = TW / 4, MT = (TH / 4) * PW / WM;
    constexpr int LH = TH + 2, LW = TW + 2;
    constexpr int PX = BF ? 8 : 16, TS = BF ? 128 : 256, KQ = BF ? 2 : 4;
    constexpr int RS = LW * PX + (BF ? 4 : 8);   // slots; fp32: RS % 16 == 8 (conflict-free ds_read_b128)
    constexpr int Q = NT * WN;                   // 16-channel output groups per workgroup
    constexpr int COUT_T = 16 * Q;
    constexpr int SEG = 9 * TS;                  // weight slots per (16 couts, 16-channel block)
    constexpr int NB4 = Q * SEG / 4;             // 16-byte pieces of weights per block
    constexpr int AIT = (LH * LW + NTHR - 1) / NTHR;
    constexpr int BIT = (NB4 + NTHR - 1) / NTHR;
    constexpr int SIT = (Q * TS / 4 + NTHR - 1) / NTHR;   // shortcut weights: Q*TS slots
    constexpr int FIT = (512 + NTHR - 1) / NTHR;      // AdaIN table: up to 512 input channels
    // DB: double-buffered LDS images for the narrow channel tiles (<= 32 output channels): the LDS
    // write of item i+1 no longer has to wait for the readers of item i, so it sits in front of
    // the MFMAs of item i in the same wave (one barrier per item instead of two) and its VALU/LDS
    // instructions issue in the gaps of the MFMA stream
    constexpr bool DB = Q <= 2 && !SC && (TH == 16 || GSA_DB_SMALL);
    constexpr int NBUF = DB ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                            // [NBUF][LH*RS]
    float* sB = sA + NBUF * LH * RS;             // [NBUF][q][tap][ci][16][cg]
    float* sS = sB + ((DB && p.w_resident) ? ((p.C0 + p.C1) >> 4) : NBUF) * Q * SEG;   // SC: [q][ci][16][cg]; resident weights: all blocks
    f32x4* sAff = reinterpret_cast<f32x4*>(sS + (SC ? Q * TS : 0));   // [C0] (mean, A, B, -)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int i16 = lane & 15, kq = lane >> 4;

    // contiguous range of work tiles of this workgroup; order (n, g, ty, tx), tx fastest
    const int chunk = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int w_begin = xcd_block(blockIdx.x, gridDim.x) * chunk;
    const int w_end = min(p.total_tiles, w_begin + chunk);
    if (w_begin >= w_end) return;
    // Index arithmetic is kept off the vector ALU: f32 MFMAs and VALU instructions do not co-execute on a SIMD
    // (SQ_VALU_MFMA_COEXEC_CYCLES = 0 in the PMC passes), so every VALU instruction of one wave is taken from
    // the MFMA stream of the other.  Only the first tile of a workgroup is decoded with divisions (wave-uniform);
    // the walk (tx, ty, g, n) is incremental, and everything per thread that does not depend on the tile
    // (tile-local pixel coordinates, weight offsets) is computed once.
    auto decode = [&](int w) {
        WorkTile t;
        const int tx = w % p.tiles_x;
        int r = w / p.tiles_x;
        const int ty = r % p.tiles_y;
        r /= p.tiles_y;
        t.g = r % p.groups;
        t.n = r / p.groups;
        t.y0 = ty * TH; t.x0 = tx * TW;
        t.row = ty * p.tiles_x + tx;
        return t;
    };
    auto advance = [&](const WorkTile& t) {      // the tile after t in (n, g, ty, tx) order
        WorkTile u = t;
        u.x0 += TW; u.row += 1;
        if (u.x0 == p.W) {
            u.x0 = 0; u.y0 += TH;
            if (u.y0 == p.H) {
                u.y0 = 0; u.row = 0; u.g += 1;
                if (u.g == p.groups) { u.g = 0; u.n += 1; }
            }
        }
        return u;
    };
    int t_ly[AIT], t_lx[AIT], t_lds[AIT];        // tile-local coordinates / LDS offset of the pixels this thread stages
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
        const int idx = tid + it * NTHR;
        t_ly[it] = idx / LW - 1; t_lx[it] = idx % LW - 1;
        t_lds[it] = idx < LH * LW ? (idx / LW) * RS + (idx % LW) * PX : -1;
    }
    auto tile_pixels = [&](const WorkTile& t, TilePixel (&tp)[AIT]) {
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int gy = t.y0 + t_ly[it], gx = t.x0 + t_lx[it];
            const int lx = t_lx[it] + 1; (void)lx;
            const bool inside = t_lds[it] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            tp[it].lds = t_lds[it];
            tp[it].pix = inside ? (t.n * p.Hs + (gy >> p.up)) * p.Ws + (gx >> p.up) : -1;
#ifdef GSA_DBG_HOOKS
            if ((p.dbg & 1) && inside) tp[it].pix = lx & 1;      // timing-only: cache-resident input
#endif
        }
    };

    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pidx = wm * MT + mt, pr = pidx / PW, pc = pidx % PW;
        abase[mt] = (pr * 4 + (i16 >> 2)) * RS + (pc * 4 + (i16 & 3)) * PX + kq * KQ;
    }
    const int bbase = wn * NT * SEG + (kq * 16 + i16) * KQ;
    const int sbase = wn * NT * TS + (kq * 16 + i16) * KQ;

    f32x4 acc[MT][NT];
    f32x4 accs[SC ? MT : 1][SC ? NT : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (SC) accs[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    const int nblk0 = p.C0 >> 4, nblk = (p.C0 + p.C1) >> 4;
    const bool has_aff = p.aff0 != nullptr;
    // Resident weights (launcher: one channel group, all nblk blocks fit): the workgroup copies the whole weight
    // panel into LDS once and the items stage activations only -- for the 16- and 32-channel layers at 512^2 and
    // 1024^2 the 9-18 KB weight block was a third to a half of every item's staging traffic and instructions.
    const bool wres = DB && p.w_resident;
    f32x4 ra[AIT][4], rb[BIT], rs[SC ? SIT : 1], rf[FIT];
    STAMP_DECL;
    STAMP(0);

    int wofs[BIT];                               // this thread's 16-byte pieces of a weight block: q*nblk*SEG + 4r
#pragma unroll
    for (int j = 0; j < BIT; ++j) {
        const int i = min(tid + j * NTHR, NB4 - 1);    // clamped: duplicates rewrite the same value
        wofs[j] = (i / (SEG / 4)) * nblk * SEG + (i % (SEG / 4)) * 4;
    }
    // prefetch of one (tile, block) item into registers -- every load unconditional
    auto load_item = [&](const WorkTile& t, int cb, const TilePixel (&tp)[AIT]) {
        const bool first = cb < nblk0;
        const float* src = first ? p.src0 : p.src1;
        const int Cs = first ? p.C0 : p.C1;
        const int coff = (first ? cb : cb - nblk0) * 16;
#pragma unroll
        for (int it = 0; it < AIT; ++it) load_pixel<BF>(ra[it], src, Cs, coff, tp[it]);
        if (!wres) {
            const float* wblk = p.wpk + ((size_t)t.g * Q * nblk + cb) * SEG;      // wave-uniform
#pragma unroll
            for (int j = 0; j < BIT; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wblk + wofs[j]);
        }
        if (SC) {
#pragma unroll
            for (int j = 0; j < SIT; ++j) {
                const int i = min(tid + j * NTHR, Q * TS / 4 - 1);
                const int q = i / (TS / 4), r = i % (TS / 4);
                rs[j] = reinterpret_cast<const f32x4*>(p.wsc + ((size_t)(t.g * Q + q) * nblk + cb) * TS)[r];
            }
        }
        if (DB) {          // the 16 AdaIN entries of this item's channel block travel with the item (lanes 0-15 matter)
            if (has_aff && cb < nblk0) rf[0] = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)t.n * p.C0 + cb * 16)[tid & 15];
        } else if (has_aff) {     // whole AdaIN table of the item's sample (copied to LDS when the sample changes)
#pragma unroll
            for (int j = 0; j < FIT; ++j)
                rf[j] = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)t.n * p.C0)[min(tid + j * NTHR, p.C0 - 1)];
        }
    };
    auto write_aff_item = [&](int slot) {     // DB form: sAff = [2][16] entries
        if (has_aff && tid < 16) sAff[slot * 16 + tid] = rf[0];
    };
    auto write_aff = [&]() {
#pragma unroll
        for (int j = 0; j < FIT; ++j) sAff[min(tid + j * NTHR, p.C0 - 1)] = rf[j];
    };
    auto is_edge = [&](const WorkTile& t) { return t.y0 == 0 || t.x0 == 0 || t.y0 + TH == p.H || t.x0 + TW == p.W; };
    auto write_item = [&](int cb, const TilePixel (&tp)[AIT], bool edge, int buf = 0) {
        float* a_img = sA + buf * (LH * RS);
        const float4* tab = reinterpret_cast<const float4*>(sAff) + (DB ? buf : cb) * 16;
        if (cb < nblk0 && has_aff) {      // wave-uniform branches
            if (edge) {
#pragma unroll
                for (int it = 0; it < AIT; ++it) store_pixel<true, BF, true>(a_img, ra[it], tab, tp[it]);
            } else {
#pragma unroll
                for (int it = 0; it < AIT; ++it) store_pixel<true, BF, false>(a_img, ra[it], tab, tp[it]);
            }
        } else if (edge) {
#pragma unroll
            for (int it = 0; it < AIT; ++it) store_pixel<false, BF, true>(a_img, ra[it], tab, tp[it]);
        } else {
#pragma unroll
            for (int it = 0; it < AIT; ++it) store_pixel<false, BF, false>(a_img, ra[it], tab, tp[it]);
        }
        if (!wres) {
#pragma unroll
            for (int j = 0; j < BIT; ++j) reinterpret_cast<f32x4*>(sB + buf * (Q * SEG))[min(tid + j * NTHR, NB4 - 1)] = rb[j];
        }
        if (SC) {
#pragma unroll
            for (int j = 0; j < SIT; ++j) reinterpret_cast<f32x4*>(sS)[min(tid + j * NTHR, Q * TS / 4 - 1)] = rs[j];
        }
    };

    // Epilogue inputs live in registers of their own so that their loads can be ISSUED BEFORE the bulk prefetch
    // of the item after next: a wave's loads return in order, and behind ~11 bulk loads per lane the few epilogue
    // loads waited out the whole memory time of that prefetch (in-kernel timers: 6-9k of 17-27k cycles per item).
    const int prow_in_patch = lane >> 4;
    const int xj = lane & 3, cq4 = ((lane >> 2) & 3) * 4;   // quad-transposed layout: lane -> (x = lane&3, channels 4*((lane>>2)&3)..+3)
    float4 nzs[EPI == EPI_SYNTH ? MT : 1];
    float e0[NT], e1[NT], e2[NT], e3[NT], scb[NT];
    f32x4 rr[EPI == EPI_DEC ? MT : 1][EPI == EPI_DEC ? NT : 1];
    const bool has_resid = EPI == EPI_DEC && p.resid != nullptr;
    // addresses = wave-uniform base of the patch (scalar ALU) + a per-lane 32-bit offset that never changes
    const int ru = p.resid_up;      // 1: the residual lives at half resolution
    const unsigned lane_out = (unsigned)((prow_in_patch * p.W + xj) * p.Cout + cq4);
    const int res_cs0 = p.resid1 ? p.res_c0 : p.Cout, res_cs1 = p.Cout - p.res_c0;   // channel strides of the residual source(s)
    const unsigned lane_rpix = (unsigned)((prow_in_patch >> ru) * (p.W >> ru) + (xj >> ru));
    const unsigned lane_res0 = lane_rpix * res_cs0 + cq4, lane_res1 = lane_rpix * res_cs1 + cq4;
    const unsigned lane_nz = (unsigned)(prow_in_patch * p.W);
    auto epilogue_loads = [&](const WorkTile& tc) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pidx = wm * MT + mt, pr = pidx / PW, pc = pidx % PW;
            if (EPI == EPI_SYNTH)
                nzs[mt] = *reinterpret_cast<const float4*>(p.noise + ((size_t)(tc.n * p.H + tc.y0 + pr * 4) * p.W + tc.x0 + pc * 4) + lane_nz);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = tc.g * COUT_T + (wn * NT + nt) * 16 + i16;
            e0[nt] = e1[nt] = e2[nt] = e3[nt] = scb[nt] = 0.f;
            if (EPI == EPI_SYNTH) { e0[nt] = p.nscale[co]; e1[nt] = p.nbias[co]; }
            if (EPI == EPI_DEC) { e2[nt] = p.bn_s[co]; e3[nt] = p.bn_beta[co]; }      // BatchNorm folded: y = lrelu(fmaf(v, s, k))
            if (SC) scb[nt] = p.sc_bias[co];
        }
        if (EPI == EPI_DEC && has_resid) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int pidx = wm * MT + mt, pr = pidx / PW, pc = pidx % PW;
                const size_t rpix = (size_t)(tc.n * (p.H >> ru) + ((tc.y0 + pr * 4) >> ru)) * (p.W >> ru) + ((tc.x0 + pc * 4) >> ru);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    // the residual may be the channel concatenation of two tensors (identity shortcut over concat(prev, cvt)):
                    // channels [0, res_c0) live in resid, the rest in resid1; a 16-channel group never straddles the split
                    const int cch = tc.g * COUT_T + (wn * NT + nt) * 16;
                    const bool second = p.resid1 != nullptr && cch >= p.res_c0;
                    rr[mt][nt] = act_load4<BF>(second ? p.resid1 : p.resid, rpix * (second ? res_cs1 : res_cs0) + (second ? cch - p.res_c0 : cch) +
                                                                             (second ? lane_res1 : lane_res0));
                }
            }
        }
    };

    // Direct statistics (launcher: persistent double-buffered EPI_SYNTH launches): a wave keeps its fixed-point
    // sums in registers across the tiles of one (sample, channel group) and adds them to acc[n][c] with one
    // 64-bit atomic per channel when the group changes -- instead of a 16-byte partial row per tile and wave
    // (16384 rows per sample at 1024^2, which finalize_kernel then needed 53 us to re-read).  The atomics go to
    // kDirectRows zero-initialised rows per sample (row = workgroup & 63): all 3072 waves adding to ONE row made
    // the 1024^2 layer 2.4x slower at batch 1 (same-address atomics serialise).
    const bool stats_direct = EPI == EPI_SYNTH && p.stats_direct;
    unsigned long long dI1[NT], dI2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dI1[nt] = dI2[nt] = 0ull;
    auto flush_stats = [&](const WorkTile& t) {      // t: any tile of the (sample, group) the sums belong to
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            unsigned long long I1 = dI1[nt], I2 = dI2[nt];
            I1 += shfl_xor_u64(I1, 16); I2 += shfl_xor_u64(I2, 16);
            I1 += shfl_xor_u64(I1, 32); I2 += shfl_xor_u64(I2, 32);
            if (lane < 16) {      // kDirectRows zeroed rows per sample: workgroups spread over them, finalize_kernel sums them
                StatPart* a = p.partials + ((size_t)t.n * p.prow + (blockIdx.x & (kDirectRows - 1))) * p.Cout + t.g * COUT_T + (wn * NT + nt) * 16 + i16;
                atomicAdd(&a->s1, I1);
                atomicAdd(&a->s2, I2);
            }
            dI1[nt] = dI2[nt] = 0ull;
        }
    };

    auto epilogue = [&](const WorkTile& tc) {
            // ---- epilogue of tile tc (after epilogue_loads(tc)).  C layout: lane -> (channel = lane&15,
            // patch row = lane>>4), reg -> patch column.
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = tc.g * COUT_T + (wn * NT + nt) * 16 + i16;
                unsigned long long I1 = 0, I2 = 0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int pidx = wm * MT + mt, pr = pidx / PW, pc = pidx % PW;
                    const size_t ubase = ((size_t)(tc.n * p.H + tc.y0 + pr * 4) * p.W + tc.x0 + pc * 4) * p.Cout + tc.g * COUT_T + (wn * NT + nt) * 16;
                    float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
                    if (EPI == EPI_SYNTH) {
                        const float nzv[4] = {nzs[mt].x, nzs[mt].y, nzs[mt].z, nzs[mt].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float t = e0[nt] * nzv[r];
                            v[r] = lrelu((v[r] + t) + e1[nt]);
                        }
                        const float s = (v[0] + v[1]) + (v[2] + v[3]);
                        const float q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                        I1 += to_fixed(s, kStatScale1);
                        I2 += to_fixed_sq(q, stat_s2(p.H * p.W));
                    }
                    if (EPI == EPI_DEC) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = lrelu(fmaf(v[r], e2[nt], e3[nt]));
                        }
                    }
                    f32x4 vt = quad_transpose(v[0], v[1], v[2], v[3], xj);
                    if (EPI == EPI_DEC && has_resid) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) vt[r] = rr[mt][nt][r] + vt[r];
                    }
#ifdef GSA_DBG_HOOKS
                    if (!(p.dbg & 4))      // timing-only: no epilogue stores
#endif
                    act_store4<BF>(p.out, ubase + lane_out, vt);
                    if (SC) {
                        act_store4<BF>(p.out_sc, ubase + lane_out,
                            quad_transpose(accs[mt][nt][0] + scb[nt], accs[mt][nt][1] + scb[nt], accs[mt][nt][2] + scb[nt],
                                           accs[mt][nt][3] + scb[nt], xj));
                    }
                    acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (SC) accs[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (EPI == EPI_SYNTH && stats_direct) {
                    dI1[nt] += I1; dI2[nt] += I2;
                } else if (EPI == EPI_SYNTH) {
                    I1 += shfl_xor_u64(I1, 16); I2 += shfl_xor_u64(I2, 16);
                    I1 += shfl_xor_u64(I1, 32); I2 += shfl_xor_u64(I2, 32);
                    if (lane < 16) {
                        StatPart sp; sp.s1 = I1; sp.s2 = I2;
                        p.partials[((size_t)tc.n * p.prow + tc.row * WM + wm) * p.Cout + co] = sp;
                    }
                }
            }
            };

    auto mfma_item = [&](int buf, int cb_res = 0) {
#ifdef GSA_DBG_HOOKS
        if (p.dbg & 16) return;      // timing-only: no MFMA phase
#endif
        if (p.prio) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(GSA_MFMA_PRIO);      // the wave in its MFMA phase wins the SIMD's issue slot
        const float* a_img = sA + buf * (LH * RS);
        const float* b_img = sB + (wres ? cb_res : buf) * (Q * SEG);
        if constexpr (BF) {
            // ---- bf16: one 16x16x16 MFMA per (tap, patch, cout group); k slot kq = channels 4kq..4kq+3
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = (tap / 3) * RS + (tap % 3) * PX;
                s16x4 a[MT], b[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const s16x4*>(a_img + abase[mt] + toff);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const s16x4*>(b_img + bbase + nt * SEG + tap * TS);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                if (SC && tap == 4) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const s16x4 bs = *reinterpret_cast<const s16x4*>(sS + sbase + nt * TS);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            accs[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[mt], bs, accs[mt][nt], 0, 0, 0);
                    }
                }
            }
            return;
        }
        // ---- MFMA: K order (tap, cg, ci) inside the block
        if constexpr (!SC && NT <= 2 && GSA_OPERAND_PIPELINE) {
            // operands of tap t+1 are read from LDS before the MFMAs of tap t are issued (explicit double buffer)
            f32x4 a2[2][MT], b2[2][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a2[0][mt] = *reinterpret_cast<const f32x4*>(a_img + abase[mt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b2[0][nt] = *reinterpret_cast<const f32x4*>(b_img + bbase + nt * SEG);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) {
                    const int toff = ((tap + 1) / 3) * RS + ((tap + 1) % 3) * 16;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a2[(tap + 1) & 1][mt] = *reinterpret_cast<const f32x4*>(a_img + abase[mt] + toff);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b2[(tap + 1) & 1][nt] = *reinterpret_cast<const f32x4*>(b_img + bbase + nt * SEG + (tap + 1) * 256);
                }
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tap & 1][mt][cg], b2[tap & 1][nt][cg], acc[mt][nt], 0, 0, 0);
            }
            if (p.prio) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
            return;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * RS + (tap % 3) * 16;
            f32x4 a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(a_img + abase[mt] + toff);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(b_img + bbase + nt * SEG + tap * 256);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][cg], b[nt][cg], acc[mt][nt], 0, 0, 0);
            if (SC && tap == 4) {  // 1x1 shortcut on the centre tap, natural channel order
                f32x4 bs[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bs[nt] = *reinterpret_cast<const f32x4*>(sS + sbase + nt * 256);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            accs[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][cg], bs[nt][cg], accs[mt][nt], 0, 0, 0);
            }
        }
        if (p.prio) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    };
    if (p.prio) __builtin_amdgcn_s_setprio(3);

    if constexpr (DB) {
        // items in flight: `tc/cb` is being multiplied out of LDS buffer it&1, `tr/cbr` sits in the
        // prefetch registers (written to buffer (it+1)&1 at the top of the next iteration)
        const int total_items = (w_end - w_begin) * nblk;
        // item i+1 of (t, cbi), or the same item again when i+1 is past the end (a harmless reload)
        auto next_item = [&](int i, WorkTile& t, int& cbi, TilePixel (&tp)[AIT]) {
            if (i + 1 >= total_items) return;
            if (++cbi == nblk) {
                cbi = 0;
                t = advance(t);
                tile_pixels(t, tp);
            }
        };
        WorkTile tc, tr;
        int cb = 0, cbr = 0;
        TilePixel tpr[AIT];
        tc = decode(w_begin);
        tile_pixels(tc, tpr);
        if (wres) {                       // the whole weight panel of the (single) channel group -> LDS, once
            for (int cbk = 0; cbk < nblk; ++cbk) {
                const float* wblk = p.wpk + (size_t)cbk * SEG;
#pragma unroll
                for (int j = 0; j < BIT; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wblk + wofs[j]);
#pragma unroll
                for (int j = 0; j < BIT; ++j) reinterpret_cast<f32x4*>(sB + cbk * (Q * SEG))[min(tid + j * NTHR, NB4 - 1)] = rb[j];
            }
        }
        load_item(tc, cb, tpr);
        if (has_aff) {
            write_aff_item(0);
            __syncthreads();
        }
        write_item(cb, tpr, is_edge(tc), 0);
        tr = tc; cbr = cb;
        next_item(0, tr, cbr, tpr);
        load_item(tr, cbr, tpr);
        write_aff_item(1);                 // entries of item 1 -> slot 1 (read after the barrier below)
        __syncthreads();
        unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, sw = 0, sl = 0, sm = 0, se = 0, sb = 0, ka = 0, kb = 0, s0 = 0, s1 = 0, s2 = 0;
        (void)ka; (void)kb; (void)s0; (void)s1; (void)s2; (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)k5; (void)sw; (void)sl; (void)sm; (void)se; (void)sb;
        for (int it = 0; it < total_items; ++it) {
            const bool has_next = it + 1 < total_items;
            TICK(k0);
#ifdef GSA_DBG_HOOKS
            if (!(p.dbg & 8))        // timing-only: no LDS staging writes in the steady state
#endif
            if (has_next) write_item(cbr, tpr, is_edge(tr), (it + 1) & 1);  // item it+1: registers -> the other LDS buffer
            TICK(k1);
            if (cb == nblk - 1) epilogue_loads(tc);            // ahead of the bulk prefetch: loads return in order
            TICK(ka);
            WorkTile t2 = tr; int cb2 = cbr;
            next_item(it + 1, t2, cb2, tpr);
            TICK(kb);
            load_item(t2, cb2, tpr);                           // item it+2 -> registers, lands during the MFMAs
            TICK(k2);
            TSUM(s0, k1, ka); TSUM(s1, ka, kb); TSUM(s2, kb, k2);
            mfma_item(it & 1, cb);
            TICK(k3);
            if (cb == nblk - 1) {
                epilogue(tc);
                if (stats_direct && (it + 1 >= total_items || tr.n != tc.n || tr.g != tc.g)) flush_stats(tc);
            }
            TICK(k4);
            write_aff_item(it & 1);        // entries of item it+2 (loaded above) -> the slot item it used; read after the barrier
            __syncthreads();
            TICK(k5);
            TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4); TSUM(sb, k4, k5);
            tc = tr; cb = cbr; tr = t2; cbr = cb2;
        }
        TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
        TFLUSH(0, s0); TFLUSH(1, s1); TFLUSH(2, s2);
        TFLUSH(12, (unsigned long long)total_items); TFLUSH(15, 1ull);
        return;
    }

    WorkTile tc = decode(w_begin);
    TilePixel tp[AIT], tpn[AIT];
    tile_pixels(tc, tp);
    load_item(tc, 0, tp);
    int n_aff = tc.n;
    if (has_aff) {
        write_aff();
        __syncthreads();
    }
#ifdef GSA_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP(1);                                      // [0,1): prologue + first global loads landed
    write_item(0, tp, is_edge(tc));
    STAMP(2);                                      // [1,2): first LDS write
    __syncthreads();
    STAMP(3);                                      // [2,3): barrier

    int w = w_begin, cb = 0;
    while (true) {
        // ---- the next item (or a harmless reload of this one when it is the last)
        int w2 = w, cb2 = cb + 1;
        if (cb2 == nblk) { cb2 = 0; w2 = w + 1; }
        const bool has_next = w2 < w_end;
        if (!has_next) { w2 = w; cb2 = cb; }
        WorkTile tn = tc;
        if (w2 != w) {
            tn = advance(tc);
            tile_pixels(tn, tpn);
        } else {
#pragma unroll
            for (int it = 0; it < AIT; ++it) tpn[it] = tp[it];
        }
        load_item(tn, cb2, tpn);                   // in flight during the MFMAs and the epilogue below
        __builtin_amdgcn_sched_barrier(0);         // keep the consumers of those loads below the MFMAs
        mfma_item(0);
        __builtin_amdgcn_sched_barrier(0);

        if (cb == nblk - 1) { epilogue_loads(tc); epilogue(tc); }
        if (!has_next) break;
        __syncthreads();              // every wave has finished reading this item's LDS image
        if (has_aff && tn.n != n_aff) {   // wave-uniform: the next item belongs to another sample
            write_aff();
            n_aff = tn.n;
            __syncthreads();
        }
#ifdef GSA_DBG_HOOKS
        if (!(p.dbg & 8))
#endif
        write_item(cb2, tpn, is_edge(tn));
        __syncthreads();
        tc = tn; w = w2; cb = cb2;
#pragma unroll
        for (int it = 0; it < AIT; ++it) tp[it] = tpn[it];
    }
    STAMP(4);                                      // [3,4): all items (MFMA, staging, epilogues)
#ifdef GSA_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP(5);                                      // [4,5): last stores retired
    STAMP_FLUSH(5);
}

// ------------------------------------------------------------------------------------------
// conv3x3 with a 4-WAY K SPLIT for the low-resolution layers (>= 64 input channels, one source; outputs <= 8 px, or <= 32 px
// with at most 32 output channels: the layers whose tiles cannot fill the chip).
//
// The 512-channel layers at 4^2-32^2 used to be ONE dependent chain of 1152 MFMAs per output tile behind 32
// load -> LDS -> barrier rounds (one wave per SIMD, every phase serialised: ~50 us whatever the batch).  Here the four waves
// of a workgroup each accumulate one contiguous QUARTER of the 16-channel blocks over ALL patches of the tile
// (4 independent chains per wave at 8x8), an item stages four blocks (one per wave) per barrier, and the partial sums are
// combined through LDS in the fixed order (s0 + s1) + (s2 + s3) -- the canonical arithmetic of these layers (DESIGN.md;
// oracle conv3x3 use_ksplit), so the result is still bit-exact and batch-independent.
// Workgroup = one TH x TH tile (TH = 4, 8) x 16 output channels of one sample.
// PS = 2 (8x8 tiles): EIGHT waves -- wave (quarter, half) multiplies its quarter over two of the four patches, and the 400 staged
// (pixel, block) units of an item are one per thread instead of four: per item a wave issues 72 MFMAs instead of 144 and
// stages a quarter of what it did, with two waves per SIMD to overlap the phases.  Same chains, same combine: same bits.
__device__ __forceinline__ void finalize_one(const FinalizeParams& p, int n, int c, unsigned long long I1, unsigned long long I2);
template <int TH, int EPI, bool BF, int PS>
__global__ __launch_bounds__(256 * PS) void conv3x3_ksplit(ConvParams p) {
    constexpr int NTHR = 256 * PS, MT = (TH / 4) * (TH / 4), PW = TH / 4;
    constexpr int MTW = MT / PS;                      // patches per wave
    constexpr int LH = TH + 2, LW = TH + 2, NPX = LH * LW;
    constexpr int PX = BF ? 8 : 16, TS = BF ? 128 : 256, KQ = BF ? 2 : 4;
    constexpr int RS = LW * PX + (BF ? 4 : 8);
    constexpr int SEG = 9 * TS, IMG = LH * RS;
    constexpr int NB4 = 4 * SEG / 4, BIT = (NB4 + NTHR - 1) / NTHR;         // weights of an item: four blocks
    constexpr int QPT = PS == 1 ? 4 : 1;              // (pixel, block) units a staging thread handles per item
    static_assert(PS == 1 || (PS == 2 && MT == 4 && 4 * NPX <= NTHR), "patch split: 8x8 tiles, one staged unit per thread");
    static_assert(NPX <= NTHR, "one staged pixel per thread and block");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                 // [2][4][IMG]
    float* sB = sA + 2 * 4 * IMG;                     // [2][4][SEG]
    f32x4* sAff = reinterpret_cast<f32x4*>(sB + 2 * 4 * SEG);     // [2][4][16]
    f32x4* sP = sAff + 2 * 4 * 16;                    // [4 quarters][MT][64 lanes] partial accumulators
    // fused finalize (p.fin_aff): the plane's sums over the MT finishing waves + their arrival count, behind sP (dynamic LDS too: a
    // static __shared__ object would make the 160 KB dynamic-size attribute of prepare_kernel invalid)
    unsigned long long (*sfin)[16] = reinterpret_cast<unsigned long long (*)[16]>(sP + 4 * MT * 64);
    unsigned& sfin_cnt = *reinterpret_cast<unsigned*>(sfin + 2);
    if (EPI == EPI_SYNTH && p.fin_aff != nullptr && threadIdx.x < 33) {
        if (threadIdx.x < 32) sfin[threadIdx.x >> 4][threadIdx.x & 15] = 0ull; else sfin_cnt = 0u;      // visible after the first barrier of the item loop
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quarter = wave & 3, half = wave >> 2;   // K quarter, patch half (PS == 2)
    const int i16 = lane & 15, kq = lane >> 4;
    const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
    const int g = blockIdx.y, n = blockIdx.z;
    const int y0 = ty * TH, x0 = tx * TH;
    const int nblk = p.C0 >> 4, nq = nblk >> 2;       // items per tile = blocks per quarter
    const bool has_aff = p.aff0 != nullptr;

    const int sq0 = PS == 1 ? 0 : tid / NPX;          // first (only) block of the item this thread stages
    TilePixel tp;
    {
        const int st = PS == 1 ? tid : tid % NPX;
        const int ly = st / LW, lx = st % LW;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        const bool stage = PS == 1 ? tid < NPX : tid < 4 * NPX;
        const bool inside = stage && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        tp.lds = stage ? ly * RS + lx * PX : -1;
        tp.pix = inside ? (n * p.Hs + (gy >> p.up)) * p.Ws + (gx >> p.up) : -1;
    }
    int abase[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int mt = half * MTW + m;
        abase[m] = ((mt / PW) * 4 + (i16 >> 2)) * RS + ((mt % PW) * 4 + (i16 & 3)) * PX + kq * KQ;
    }
    const int bbase = (kq * 16 + i16) * KQ;

    f32x4 acc[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[QPT][4], rb[BIT], rf;
    rf = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wgrp = p.wpk + (size_t)g * nblk * SEG;
    auto load_item = [&](int it) {                    // blocks q*nq + it, q = 0..3
#pragma unroll
        for (int u = 0; u < QPT; ++u) load_pixel<BF>(ra[u], p.src0, p.C0, ((min(sq0, 3) + u) * nq + it) * 16, tp);
#pragma unroll
        for (int j = 0; j < BIT; ++j) {
            const int i = min(tid + j * NTHR, NB4 - 1);
            const int q = i / (SEG / 4), r = i % (SEG / 4);
            rb[j] = reinterpret_cast<const f32x4*>(wgrp + (size_t)(q * nq + it) * SEG)[r];
        }
        if (has_aff && tid < 64) rf = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)n * p.C0 + ((tid >> 4) * nq + it) * 16)[tid & 15];
    };
    auto write_aff = [&](int buf) {
        if (has_aff && tid < 64) sAff[buf * 64 + tid] = rf;
    };
    auto write_item = [&](int buf) {
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            const int q = min(sq0, 3) + u;
            float* a_img = sA + (buf * 4 + q) * IMG;
            const float4* tab = reinterpret_cast<const float4*>(sAff) + (buf * 4 + q) * 16;
            if (has_aff) store_pixel<true, BF, true>(a_img, ra[u], tab, tp);
            else store_pixel<false, BF, true>(a_img, ra[u], tab, tp);
        }
#pragma unroll
        for (int j = 0; j < BIT; ++j) reinterpret_cast<f32x4*>(sB + buf * 4 * SEG)[min(tid + j * NTHR, NB4 - 1)] = rb[j];
    };
    auto mfma_item = [&](int buf) {                   // this wave's block of the item
        const float* a_img = sA + (buf * 4 + quarter) * IMG;
        const float* b_img = sB + (buf * 4 + quarter) * SEG + bbase;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * RS + (tap % 3) * PX;
            if constexpr (BF) {
                const s16x4 b = *reinterpret_cast<const s16x4*>(b_img + tap * TS);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const s16x4*>(a_img + abase[m] + toff), b, acc[m], 0, 0, 0);
            } else {
                f32x4 a[MTW];
#pragma unroll
                for (int m = 0; m < MTW; ++m) a[m] = *reinterpret_cast<const f32x4*>(a_img + abase[m] + toff);
                const f32x4 b = *reinterpret_cast<const f32x4*>(b_img + tap * 256);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int m = 0; m < MTW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][cg], b[cg], acc[m], 0, 0, 0);
            }
        }
    };

    // ---- items: double-buffered LDS, item it+1 in registers, item it+2 in flight
    load_item(0);
    write_aff(0);
    __syncthreads();
    write_item(0);
    load_item(min(1, nq - 1));
    write_aff(1);
    __syncthreads();
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, sw = 0, sl = 0, sm = 0, sb = 0;
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)sw; (void)sl; (void)sm; (void)sb;
    for (int it = 0; it < nq; ++it) {
        TICK(k0);
        if (it + 1 < nq) write_item((it + 1) & 1);
        TICK(k1);
        load_item(min(it + 2, nq - 1));
        TICK(k2);
        mfma_item(it & 1);
        TICK(k3);
        write_aff(it & 1);            // entries of item it+2 -> the slot item it used (its reader ran in iteration it-1); read after the barrier
        __syncthreads();              // readers of buffer it&1 done; buffer (it+1)&1 complete
        TICK(k4);
        TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(sb, k3, k4);
    }
    TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(10, sb); TFLUSH(12, (unsigned long long)nq); TFLUSH(15, 1ull);
    // ---- combine the four quarters in the fixed order (s0 + s1) + (s2 + s3); wave w finishes patch w
#pragma unroll
    for (int m = 0; m < MTW; ++m) sP[(quarter * MT + half * MTW + m) * 64 + lane] = acc[m];
    __syncthreads();
    if (wave >= MT) return;
    const int mt = wave;
    const f32x4 s01 = sP[(0 * MT + mt) * 64 + lane] + sP[(1 * MT + mt) * 64 + lane];
    const f32x4 s23 = sP[(2 * MT + mt) * 64 + lane] + sP[(3 * MT + mt) * 64 + lane];
    const f32x4 tot = s01 + s23;
    float v[4] = {tot[0], tot[1], tot[2], tot[3]};
    // ---- epilogue of patch mt: C layout lane -> (channel = lane & 15, patch row = lane >> 4), reg -> x
    const int pr = mt / PW, pc = mt % PW, prow = lane >> 4;
    const int co = g * 16 + i16;
    const int y = y0 + pr * 4 + prow, xb = x0 + pc * 4;
    const int xj = lane & 3, cq4 = ((lane >> 2) & 3) * 4;
    if (EPI == EPI_SYNTH) {
        const float4 nz = *reinterpret_cast<const float4*>(p.noise + ((size_t)(n * p.H + y) * p.W + xb));
        const float nzv[4] = {nz.x, nz.y, nz.z, nz.w};
        const float e0 = p.nscale[co], e1 = p.nbias[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = e0 * nzv[r];
            v[r] = lrelu((v[r] + t) + e1);
        }
        const float sq = (v[0] + v[1]) + (v[2] + v[3]);
        const float qq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        unsigned long long I1 = to_fixed(sq, kStatScale1), I2 = to_fixed_sq(qq, stat_s2(p.H * p.W));
        I1 += shfl_xor_u64(I1, 16); I2 += shfl_xor_u64(I2, 16);
        I1 += shfl_xor_u64(I1, 32); I2 += shfl_xor_u64(I2, 32);
        if (p.fin_aff != nullptr) {
            // the tile is the whole plane: the MT finishing waves add their integer sums in LDS, the last one writes the coefficients
            // (finalize_one, the tail of finalize_kernel: same bits) -- no partial rows, no finalize launch (round 4)
            if (lane < 16) { atomicAdd(&sfin[0][i16], I1); atomicAdd(&sfin[1][i16], I2); }
            __threadfence_block();
            unsigned arrived = 0;
            if (lane == 0) arrived = atomicAdd(&sfin_cnt, 1u);
            arrived = __shfl(arrived, 0);
            if (arrived == (unsigned)MT - 1u && lane < 16) {
                __threadfence_block();
                FinalizeParams f{};
                f.HW = p.H * p.W; f.C = p.Cout; f.style = p.fin_style; f.style_stride = p.fin_style_stride;
                f.gamma = p.fin_gamma; f.beta = p.fin_beta; f.aff = p.fin_aff; f.flags = p.fin_flags;
                finalize_one(f, n, co, atomicAdd(&sfin[0][i16], 0ull), atomicAdd(&sfin[1][i16], 0ull));
            }
        } else if (lane < 16) {
            StatPart sp; sp.s1 = I1; sp.s2 = I2;
            p.partials[((size_t)n * p.prow + blockIdx.x * MT + mt) * p.Cout + co] = sp;
        }
    }
    if (EPI == EPI_DEC) {
        const float e2 = p.bn_s[co], e3 = p.bn_beta[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = lrelu(fmaf(v[r], e2, e3));
        }
    }
    f32x4 vt = quad_transpose(v[0], v[1], v[2], v[3], xj);
    if (EPI == EPI_DEC && p.resid != nullptr) {
        const int ru = p.resid_up;
        const size_t rpix = (size_t)(n * (p.H >> ru) + (y >> ru)) * (p.W >> ru) + ((xb + xj) >> ru);
        const int cch = g * 16;
        const bool second = p.resid1 != nullptr && cch >= p.res_c0;
        const int rcs = second ? p.Cout - p.res_c0 : (p.resid1 ? p.res_c0 : p.Cout);
        const f32x4 rr = act_load4<BF>((second ? p.resid1 : p.resid), rpix * rcs + (second ? cch - p.res_c0 : cch) + cq4);
#pragma unroll
        for (int r = 0; r < 4; ++r) vt[r] = rr[r] + vt[r];
    }
    act_store4<BF>(p.out, ((size_t)(n * p.H + y) * p.W + xb + xj) * p.Cout + g * 16 + cq4, vt);
}

// Packed fp32 adds for the Winograd transforms: one v_pk_add_f32 per two additions (neg modifiers for a subtraction; the
// same IEEE results).  hipcc scalarises a <4 x float> add whose only users are element extracts -- each component feeds its
// own MFMA -- into four v_add_f32 / v_sub_f32 (and folds every other spelling of the subtraction back into one), so the
// instruction is written out.  f32 MFMAs and vector-ALU instructions share the SIMD's issue time:
// every instruction saved is MFMA time gained.
// Hazards: the compiler's hazard recot form needs 144.  Canonical arithmetic: oracle/c/gsa_oracle.c conv3x3_wino (the transforms
// are plain fp32 adds in a fixed order, each M_f is one k-ordered fmaf chain = the MFMA, U is computed in double on
// the host and rounded once), so the result is reproduced BIT FOR BIT; DESIGN.md states the rule that selects the
// form (layer shape only: plain 3x3 convs with outputs >= 32 px, fp32 mode).
//
// Workgroup = 4 waves on a 16x16 output tile x 16 output channels; wave w owns the 8x8 quadrant (w>>1, w&1) = 16
// Winograd tiles.  Lane (i16, kq): tile (ty, tx) = (2*b3 + b1, 2*b2 + b0) of the quadrant (b = bits of i16) -- with
// the row stride 18*16+4 floats every ds_read_b128 of the 4x4 input patch is bank-conflict free -- and k slot kq.
// The C layout then gives a lane (channel = lane & 15, group = lane >> 4) the 2x2 tiles of ONE 4x4 output patch in
// its four accumulator registers: after the output transform a lane holds four aligned x-quads, exactly the shape
// the epilogue of conv3x3_mfma works on (in-lane statistics, quad transpose, 16-byte stores).
// Staging (AdaIN on read, zero padding, double-buffered LDS image, prefetch two items ahead, resident or streamed
// weight panel, direct statistics, XCD-aware tile walk) is the structure of conv3x3_mfma's double-buffered form.
// blockIdx.y = output-channel group, so a resident panel never changes under a persistent workgroup.
// CHUNK = staging by 16-byte chunks with the AdaIN coefficients in registers (see "staging by 16-byte chunks" above): measured
// faster for the layers with >= 64 input channels (g.64 / g.128 / g.256.conv_2 -4..-9 %), slower for the resident-weight layers
// with <= 32 (d.main_6.b +11 %, d.cvt_8 +5 %), which therefore keep whole-pixel staging with a two-slot LDS coefficient table.
// GW = output-channel groups per workgroup (round 3): GW = 2 is a 512-thread workgroup whose waves 0-3 / 4-7 multiply the SAME staged
// activation image by the weights of two neighbouring 16-channel groups -- a layer with several groups stages (loads, AdaIN, LDS
// writes) every input tile once per group, so sharing the image between two groups halves that work per MFMA; occupancy is
// unchanged (one 8-wave workgroup per CU instead of two 4-wave ones).  Same arithmetic per output: same bits.
// TW = 2 (round 4, VERDICT r3 item 5 in the only form that keeps the occupancy): ONE 512-thread workgroup on a 16x32 tile -- waves 0-3 the left
// 16x16 half, waves 4-7 the right one, ONE staged 18x34 halo image (612 instead of 2 x 324 pixels), one barrier per item for both halves;
// resident-weight layers only (both halves multiply by the same LDS panel), GW = 1.  Same arithmetic per output: same bits.
template <int EPI, int NT, bool CHUNK, bool AFF, int GW = 1, int TW = 1>      // AFF: the source carries AdaIN coefficients (p.aff0 != null) -- compile time, so that the
__global__ __launch_bounds__(256 * GW * TW, NT == 1 ? 2 / TW : 1) void conv3x3_wino(ConvParams p) {      // instantiations without it carry none of its code
    static_assert(TW == 1 || (TW == 2 && GW == 1 && NT == 1), "two tiles per workgroup: one channel group, 16 output channels");
    constexpr int NTHR = 256 * GW * TW, LH = 18, LW = 16 * TW + 2, RS = LW * 16 + 4;
    constexpr int SEG = 16 * 256;                // U floats per (16 couts, 16-channel block): 16 frequencies x [ci][16][cg]
    constexpr int NB4 = SEG / 4, BIT = NT * NB4 / 256;      // 16-byte pieces of one (16 couts, block) segment; pieces per thread of a block
    constexpr int NCH = CHUNK ? (LH * LW * 4 + NTHR - 1) / NTHR : (LH * LW + NTHR - 1) / NTHR;     // staging rounds per item (chunks or pixels)
    constexpr int RW = CHUNK ? 1 : 4;                        // 16-byte registers per staged unit
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nblk = p.C0 >> 4;
    const bool wres = p.w_resident != 0;
    float* sA = smem;                            // [2][LH*RS]
    float* sB = sA + 2 * LH * RS;                // resident: [GW][nblk][NT][SEG]; streamed: [2][GW][NT][SEG]
    f32x4* sAff = reinterpret_cast<f32x4*>(sB + (wres ? nblk : 2) * GW * NT * SEG);   // !CHUNK: [2][16] (mean, A, B, -)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;                         // quadrant of the (half) tile
    const int gsel = TW == 2 ? 0 : wave8 >> 2;          // which of the workgroup's GW channel groups
    const int tsel = TW == 2 ? wave8 >> 2 : 0;          // which 16x16 half of a 16x32 tile
    const int tid8 = tid & 255;                         // thread index inside its group's 256 threads (weight staging)
    const int i16 = lane & 15, kq = lane >> 4;
    // group_minor (launcher: one tile per workgroup, several channel groups, all weight panels together small enough for an
    // XCD's L2): a 1-D grid in which the G groups of ONE tile are consecutive workgroups of ONE XCD (workgroup b runs on XCD
    // b % 8), so the tile's input is fetched from HBM once and the other G-1 groups find it in that L2 -- in group-major order
    // every group re-read the whole input from HBM / MALL (rocprofv3: 1.58x the algorithmic bytes).
    int g, w_begin, w_end;
    if (p.group_minor) {
        const int b = blockIdx.x, j = b >> 3;
        g = (j % p.groups) * GW + gsel;
        w_begin = (j / p.groups) * 8 + (b & 7);
        w_end = w_begin + 1;
    } else {
        // contiguous range of the group's tiles, order (n, ty, tx)
        g = (int)blockIdx.y * GW + gsel;
        const int chunk = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
        w_begin = xcd_block(blockIdx.x, gridDim.x) * chunk;
        w_end = min(p.total_tiles, w_begin + chunk);
    }
    if (w_begin >= w_end) return;
    struct Tile { int n, y0, x0, row; };
    auto advance = [&](const Tile& t) {
        Tile u = t;
        u.x0 += 16 * TW; u.row += 1;
        if (u.x0 == p.W) { u.x0 = 0; u.y0 += 16; if (u.y0 == p.H) { u.y0 = 0; u.row = 0; u.n += 1; } }
        return u;
    };
    auto is_edge = [&](const Tile& t) { return t.y0 == 0 || t.x0 == 0 || t.y0 + 16 == p.H || t.x0 + 16 * TW == p.W; };
    const int part = tid & 3;                    // CHUNK: this thread's four channels of every 16-channel block
    int t_ly[NCH], t_lx[NCH], t_lds[NCH];        // tile-local coordinates / LDS offset of the units this thread stages
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int idx = CHUNK ? (tid + k * NTHR) >> 2 : tid + k * NTHR;   // pixel of the halo tile
        t_ly[k] = idx / LW - 1; t_lx[k] = idx % LW - 1;
        t_lds[k] = idx < LH * LW ? (idx / LW) * RS + (idx % LW) * 16 + (CHUNK ? part * 4 : 0) : -1;
    }
    int t_rel[NCH];                              // pixel offset from the tile origin (0 for an idle chunk: any valid pixel)
#pragma unroll
    for (int k = 0; k < NCH; ++k) t_rel[k] = t_lds[k] >= 0 ? t_ly[k] * p.W + t_lx[k] : 0;
    auto tile_chunks = [&](const Tile& t, Chunk (&tp)[NCH]) {
        const int base = (t.n * p.H + t.y0) * p.W + t.x0;
        if (t.y0 == 0 || t.x0 == 0 || t.y0 + 16 == p.H || t.x0 + 16 * TW == p.W) {      // wave-uniform: only edge tiles test their pixels
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int gy = t.y0 + t_ly[k], gx = t.x0 + t_lx[k];
                const bool inside = t_lds[k] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                tp[k].lds = t_lds[k];
                tp[k].pix = inside ? base + t_rel[k] : -1;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) { tp[k].lds = t_lds[k]; tp[k].pix = base + t_rel[k]; }
        }
    };
    // A operand: the 4x4 input patch of this lane's Winograd tile (halo coordinates), k slot kq
    const int wty = ((i16 >> 3) & 1) * 2 + ((i16 >> 1) & 1), wtx = ((i16 >> 2) & 1) * 2 + (i16 & 1);
    const int pbase = ((wave >> 1) * 8 + 2 * wty) * RS + (tsel * 16 + (wave & 1) * 8 + 2 * wtx) * 16 + kq * 4;
    const int bbase = (kq * 16 + i16) * 4;

    f32x4 acc[16][NT];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr bool has_aff = AFF;
    // whole-pixel staging with the coefficients as structure-of-arrays registers (8 packed fma per pixel instead of 16 table reads +
    // 16 scalar fma, store_pixel_soa): measured 3 % SLOWER on the four layers that use it (d.cvt_7/8 0.265/0.292 -> 0.273/0.306 ms,
    // g.512/g.1024.conv_2 0.277/0.333 -> 0.287/0.337): packed fp32 operations cost more issue time beside the MFMA stream than
    // the scalar pair.  Kept for A/B builds (-DGSA_WINO_SOA=1); same bits.
    constexpr bool kSoA = GSA_WINO_SOA != 0;
    f32x4 ra[NCH][RW], rb[BIT], raff[4];      // raff: CHUNK: this thread's 4 AdaIN entries; else raff[0] = one entry of the item's 16 (lanes 0-15)
#pragma unroll
    for (int c = 0; c < 4; ++c) raff[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wgrp = p.wpk + (size_t)g * NT * nblk * SEG;      // U panels of this output-channel group: [q][cb][SEG]
    int wsrc[BIT];                                   // this thread's pieces of a block: q*nblk*SEG + 4r floats
#pragma unroll
    for (int j = 0; j < BIT; ++j) {
        const int i = tid8 + j * 256;
        wsrc[j] = (i / NB4) * nblk * SEG + (i % NB4) * 4;
    }
    auto load_item = [&](const Tile& t, int cb, const Chunk (&tp)[NCH]) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if constexpr (CHUNK) ra[k][0] = load_chunk(p.src0, p.C0, cb * 16 + part * 4, tp[k]);
            else load_pixel(ra[k], p.src0, p.C0, cb * 16, TilePixel{tp[k].pix, tp[k].lds});
        }
        if (!wres) {
#pragma unroll
            for (int j = 0; j < BIT; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wgrp + (size_t)cb * SEG + wsrc[j]);
        }
        if (has_aff) {      // travels with the item
            if constexpr (CHUNK) load_aff4(raff, p.aff0, (size_t)t.n * p.C0 + cb * 16 + part * 4);
            else raff[0] = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)t.n * p.C0 + cb * 16)[tid & 15];
        }
    };
    auto write_aff_item = [&](int slot) {        // !CHUNK: the item's 16 entries -> LDS slot (read after the next barrier)
        if constexpr (!CHUNK && kSoA) {      // structure of arrays: [slot][A of the 16 channels | B of the 16 channels]
            if (has_aff && tid < 16) {
                float* tabf = reinterpret_cast<float*>(sAff) + slot * 32;
                tabf[tid] = raff[0][1]; tabf[16 + tid] = raff[0][2];
            }
        } else if constexpr (!CHUNK) {
            if (has_aff && tid < 16) sAff[slot * 16 + tid] = raff[0];
        }
    };
    auto stage_unit = [&](auto aff_tag, auto mask_tag, float* a_img, const float4* tab, int k, const Chunk& u) {
        constexpr bool A = decltype(aff_tag)::value, M = decltype(mask_tag)::value;
        if constexpr (CHUNK) store_chunk<A, false, M>(a_img, ra[k][0], raff, u);
        else store_pixel<A, false, M>(a_img, ra[k], tab, TilePixel{u.pix, u.lds});
    };
    auto write_item = [&](const Chunk (&tp)[NCH], bool edge, int buf) {
        float* a_img = sA + buf * (LH * RS);
        const float4* tab = reinterpret_cast<const float4*>(sAff) + buf * 16;
        bool staged = false;
        if constexpr (!CHUNK && kSoA) {
            if (has_aff) {               // whole pixels: the item's coefficients as packed operands, one table read per item
                f32x4 cA[4], cB[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { cA[j] = sAff[buf * 8 + j]; cB[j] = sAff[buf * 8 + 4 + j]; }
                if (edge) {
#pragma unroll
                    for (int k = 0; k < NCH; ++k) store_pixel_soa<true>(a_img, ra[k], cA, cB, TilePixel{tp[k].pix, tp[k].lds});
                } else {
#pragma unroll
                    for (int k = 0; k < NCH; ++k) store_pixel_soa<false>(a_img, ra[k], cA, cB, TilePixel{tp[k].pix, tp[k].lds});
                }
                staged = true;
            }
        }
        if (staged) {
        } else if (has_aff) {
            if (edge) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) stage_unit(std::true_type{}, std::true_type{}, a_img, tab, k, tp[k]);
            } else {
#pragma unroll
                for (int k = 0; k < NCH; ++k) stage_unit(std::true_type{}, std::false_type{}, a_img, tab, k, tp[k]);
            }
        } else if (edge) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) stage_unit(std::false_type{}, std::true_type{}, a_img, tab, k, tp[k]);
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) stage_unit(std::false_type{}, std::false_type{}, a_img, tab, k, tp[k]);
        }
        if (!wres) {
#pragma unroll
            for (int j = 0; j < BIT; ++j) reinterpret_cast<f32x4*>(sB + (buf * GW + gsel) * (NT * SEG))[tid8 + j * 256] = rb[j];
        }
    };

    // ---- epilogue geometry: lane -> (channel = lane & 15, 4x4 output patch (b3, b2) = lane >> 4 of the wave's quadrant);
    // registers after the output transform: row r of the patch as an x-quad
    const int xj = lane & 3, cq4 = ((lane >> 2) & 3) * 4;     // quad-transposed store layout
    const int prow0 = (wave >> 1) * 8 + ((lane >> 5) & 1) * 4, pcol0 = tsel * 16 + (wave & 1) * 8 + ((lane >> 4) & 1) * 4;
    const int co0 = g * (16 * NT);                        // first output channel of the group
    const unsigned lane_out = (unsigned)((prow0 * p.W + pcol0 + xj) * p.Cout + co0 + cq4);
    const unsigned lane_nz = (unsigned)(prow0 * p.W + pcol0);
    const bool has_resid = EPI == EPI_DEC && p.resid != nullptr;
    const int ru = p.resid_up;
    const int res_cs0 = p.resid1 ? p.res_c0 : p.Cout, res_cs1 = p.Cout - p.res_c0;
    float4 nzs[EPI == EPI_SYNTH ? 4 : 1];
    f32x4 rr[EPI == EPI_DEC ? 4 : 1][EPI == EPI_DEC ? NT : 1];
    float e0[NT], e1[NT], e2[NT], e3[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + nt * 16 + i16;
        e0[nt] = e1[nt] = e2[nt] = e3[nt] = 0.f;
        if (EPI == EPI_SYNTH) { e0[nt] = p.nscale[co]; e1[nt] = p.nbias[co]; }
        if (EPI == EPI_DEC) { e2[nt] = p.bn_s[co]; e3[nt] = p.bn_beta[co]; }      // BatchNorm folded: y = lrelu(fmaf(v, s, k))
    }
    auto epilogue_loads = [&](const Tile& tc) {
        if (EPI == EPI_SYNTH) {
            const float* nz = p.noise + ((size_t)(tc.n * p.H + tc.y0) * p.W + tc.x0) + lane_nz;
#pragma unroll
            for (int r = 0; r < 4; ++r) nzs[r] = *reinterpret_cast<const float4*>(nz + r * p.W);
        }
        if (EPI == EPI_DEC && has_resid) {
            const int Wr = p.W >> ru;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t rpix = (size_t)(tc.n * (p.H >> ru) + ((tc.y0 + prow0 + r) >> ru)) * Wr + ((tc.x0 + pcol0 + xj) >> ru);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    // the residual may be concat(resid [res_c0 channels], resid1): a 16-channel group never straddles the split
                    const int cch = co0 + nt * 16;
                    const bool second = p.resid1 != nullptr && cch >= p.res_c0;
                    const float* rsrc = (second ? p.resid1 : p.resid) + rpix * (second ? res_cs1 : res_cs0) + (second ? cch - p.res_c0 : cch);
                    rr[r][nt] = *reinterpret_cast<const f32x4*>(rsrc + cq4);
                }
            }
        }
    };
    const bool stats_direct = EPI == EPI_SYNTH && p.stats_direct;
    unsigned long long dI1[NT], dI2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dI1[nt] = dI2[nt] = 0ull;
    auto flush_stats = [&](const Tile& t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            unsigned long long I1 = dI1[nt], I2 = dI2[nt];
            I1 += shfl_xor_u64(I1, 16); I2 += shfl_xor_u64(I2, 16);
            I1 += shfl_xor_u64(I1, 32); I2 += shfl_xor_u64(I2, 32);
            if (lane < 16) {
                StatPart* a = p.partials + ((size_t)t.n * p.prow + (blockIdx.x & (kDirectRows - 1))) * p.Cout + co0 + nt * 16 + i16;
                atomicAdd(&a->s1, I1);
                atomicAdd(&a->s2, I2);
            }
            dI1[nt] = dI2[nt] = 0ull;
        }
    };
    auto epilogue = [&](const Tile& tc) {
        const size_t ubase = ((size_t)(tc.n * p.H + tc.y0) * p.W + tc.x0) * p.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // output transform Y = A^T M A: rows s0 = (M0+M1)+M2, s1 = (M1-M2)-M3, then the same along the columns;
            // register component = Winograd tile (b1, b0) of the lane's 4x4 patch
            f32x4 s0[4], s1[4], m[16];
#pragma unroll
            for (int f = 0; f < 16; ++f) m[f] = acc[f][nt];
            mfma_settle(m);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = add4(add4(m[j], m[4 + j]), m[8 + j]);
                s1[j] = sub4(sub4(m[4 + j], m[8 + j]), m[12 + j]);
            }
            const f32x4 y00 = add4(add4(s0[0], s0[1]), s0[2]), y01 = sub4(sub4(s0[1], s0[2]), s0[3]);     // tile row 0: x = 0, 1
            const f32x4 y10 = add4(add4(s1[0], s1[1]), s1[2]), y11 = sub4(sub4(s1[1], s1[2]), s1[3]);     // tile row 1
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned long long I1 = 0, I2 = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // patch row r = 2*b1 + i: tiles (b1, 0) and (b1, 1), tile row i
                const int b1 = r >> 1;
                const f32x4& ya = (r & 1) ? y10 : y00;
                const f32x4& yb = (r & 1) ? y11 : y01;
                float v[4] = {ya[2 * b1], yb[2 * b1], ya[2 * b1 + 1], yb[2 * b1 + 1]};
                if (EPI == EPI_SYNTH) {
                    const float nzv[4] = {nzs[r].x, nzs[r].y, nzs[r].z, nzs[r].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = e0[nt] * nzv[k];
                        v[k] = lrelu((v[k] + t) + e1[nt]);
                    }
                    const float sq = (v[0] + v[1]) + (v[2] + v[3]);
                    const float qq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    I1 += to_fixed(sq, kStatScale1);
                    I2 += to_fixed_sq(qq, stat_s2(p.H * p.W));
                }
                if (EPI == EPI_DEC) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = lrelu(fmaf(v[k], e2[nt], e3[nt]));
                    }
                }
                f32x4 vt = quad_transpose(v[0], v[1], v[2], v[3], xj);
                if (EPI == EPI_DEC && has_resid) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) vt[k] = rr[r][nt][k] + vt[k];
                }
                *reinterpret_cast<f32x4*>(p.out + ubase + (size_t)r * p.W * p.Cout + nt * 16 + lane_out) = vt;
            }
            if (EPI == EPI_SYNTH && stats_direct) {
                dI1[nt] += I1; dI2[nt] += I2;
            } else if (EPI == EPI_SYNTH) {
                I1 += shfl_xor_u64(I1, 16); I2 += shfl_xor_u64(I2, 16);
                I1 += shfl_xor_u64(I1, 32); I2 += shfl_xor_u64(I2, 32);
                if (lane < 16) {
                    StatPart sp; sp.s1 = I1; sp.s2 = I2;
                    p.partials[((size_t)tc.n * p.prow + tc.row * (4 * TW) + (TW == 2 ? wave8 : wave)) * p.Cout + co0 + nt * 16 + i16] = sp;
                }
            }
        }
    };

    auto wino_item = [&](int buf, int cb_res) {
        __builtin_amdgcn_s_setprio(GSA_MFMA_PRIO);
        const float* a_img = sA + buf * (LH * RS) + pbase;
        const float* b_img = sB + (wres ? gsel * nblk + cb_res : buf * GW + gsel) * (NT * SEG) + bbase;
        // input transform V = B^T d B on the lane's 4x4 patch, four channels (cg) per vector:
        //   rows  t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3;   columns: the same four forms
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
        // sixteen GEMMs: per frequency one chain over the block's channels (cg ascending), four frequencies interleaved
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            f32x4 b[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[j][nt] = *reinterpret_cast<const f32x4*>(b_img + nt * SEG + (fb * 4 + j) * 256);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[fb * 4 + j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[fb * 4 + j][cg], b[j][nt][cg], acc[fb * 4 + j][nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- items (tile, channel block): item `it` is multiplied out of LDS buffer it & 1, item it+1 sits in the
    // prefetch registers, item it+2 is being loaded
    const int total_items = (w_end - w_begin) * nblk;
    Tile tc, tr;
    {
        const int tx = w_begin % p.tiles_x, r = w_begin / p.tiles_x;
        tc.x0 = tx * 16 * TW; tc.y0 = (r % p.tiles_y) * 16; tc.n = r / p.tiles_y; tc.row = (r % p.tiles_y) * p.tiles_x + tx;
    }
    int cb = 0, cbr = 0;
    Chunk tpr[NCH];
    auto next_item = [&](int i, Tile& t, int& cbi, Chunk (&tp)[NCH]) {
        if (i + 1 >= total_items) return;
        if (++cbi == nblk) { cbi = 0; t = advance(t); tile_chunks(t, tp); }
    };
    tile_chunks(tc, tpr);
    if (wres) {
        for (int cbk = 0; cbk < nblk; ++cbk) {
#pragma unroll
            for (int j = 0; j < BIT; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wgrp + (size_t)cbk * SEG + wsrc[j]);
#pragma unroll
            for (int j = 0; j < BIT; ++j) reinterpret_cast<f32x4*>(sB + (gsel * nblk + cbk) * (NT * SEG))[tid8 + j * 256] = rb[j];
        }
    }
    load_item(tc, cb, tpr);
    if (!CHUNK && has_aff) {
        write_aff_item(0);
        __syncthreads();
    }
    write_item(tpr, is_edge(tc), 0);
    tr = tc; cbr = cb;
    next_item(0, tr, cbr, tpr);
    load_item(tr, cbr, tpr);
    write_aff_item(1);                 // entries of item 1 -> slot 1 (read after the barrier below)
    __syncthreads();
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, sw = 0, sl = 0, sm = 0, se = 0, sb = 0;
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)k5; (void)sw; (void)sl; (void)sm; (void)se; (void)sb;
    for (int it = 0; it < total_items; ++it) {
        const bool has_next = it + 1 < total_items;
        TICK(k0);
        if (has_next) write_item(tpr, is_edge(tr), (it + 1) & 1);
        TICK(k1);
        if (cb == nblk - 1) epilogue_loads(tc);
        Tile t2 = tr; int cb2 = cbr;
        next_item(it + 1, t2, cb2, tpr);
        load_item(t2, cb2, tpr);
        TICK(k2);
        wino_item(it & 1, cb);
        TICK(k3);
        if (cb == nblk - 1) {
            epilogue(tc);
            if (stats_direct && (!has_next || tr.n != tc.n)) flush_stats(tc);
        }
        TICK(k4);
        write_aff_item(it & 1);        // entries of item it+2 (loaded above) -> the slot item it used; read after the barrier
        __syncthreads();
        TICK(k5);
        TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4); TSUM(sb, k4, k5);
        tc = tr; cb = cbr; tr = t2; cbr = cb2;
    }
    TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
    TFLUSH(12, (unsigned long long)total_items); TFLUSH(15, 1ull);
}

#if GSA_EXPERIMENTS      // the measured-slower kernels of round 4 (conv3x3_wino_dma, conv3x3_wino43): `make experiments` builds them into libgsa_hip_exp.so
// Every barrier of these two kernels publishes LDS-DMA data: the issuing wave waits for its own DMA explicitly (vmcnt) before the
// barrier -- gfx950's s_barrier does not wait for outstanding vector-memory operations, and relying on the compiler's conservative
// placement of its own waits tied the kernels to one hipcc.
#define GSA_DMA_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
// ------------------------------------------------------------------------------------------
// conv3x3_wino with ALL staging by LDS-DMA (round 4) -- the streamed-weight layers (>= 64 input channels, one output-channel group
// per workgroup): same workgroup geometry, same LDS image addresses, same MFMAs, same epilogue, same bits as conv3x3_wino<EPI, 1,
// true, AFF, 1>; what changes is how the operands reach LDS.  There a thread carries the next-but-one item through 24 + 16
// prefetch registers (six chunk loads, four weight loads), applies AdaIN and writes LDS (stamps: write ~1600 + load ~1750 of ~8600
// cycles per item and wave, 256 registers and 20 bytes of scratch); here global_load_lds_dwordx4 moves both operands (21 + 16
// instructions of 1 KB per item and workgroup), no staging register exists, AdaIN is ONE packed fma per patch vector after the LDS
// read (the patch of an edge tile masks its zero padding: a pixel outside the image must stay 0, not become B), and the next item's
// DMA is in flight while this item is multiplied.  MEASURED SLOWER than the register-staged kernel (DESIGN.md section 4, round 4:
// per item and wave DMA issue ~2400 + barrier ~1800 cycles against write ~1600 + load ~1750 + barrier ~570): opt-in (GSA_WINO_DMA=1).
// LDS image: the addresses conv3x3_wino reads (row stride 18*16+4 floats = 73 units of 16 bytes); DMA round k of the workgroup fills
// units 256k .. 256k+255 (wave w: 64 consecutive units), unit u = 73*row + 4*pixel + chunk (u % 73 == 72: the row's padding unit,
// u >= 1314: the buffer's tail -- both read the page of zeros, as does a pixel outside the image).
template <int EPI, bool AFF>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_dma(ConvParams p) {
    constexpr int LH = 18, LW = 18, RS = LW * 16 + 4, RU = RS / 4;          // floats / 16-byte units per image row
    constexpr int NU = LH * RU, NRD = (NU + 63) / 64;                        // 1314 units, 21 DMA instructions per item
    constexpr int IMGF = NRD * 256;                                          // floats per image buffer
    constexpr int SEG = 16 * 256;                                            // U floats per (16 couts, 16-channel block)
    constexpr int NRT = (NRD + 3) / 4;                                       // DMA rounds per wave (6, the last one for wave 0 only)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nblk = p.C0 >> 4;
    float* sA = smem;                            // [2][IMGF]
    float* sB = sA + 2 * IMGF;                   // [2][SEG]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int g = (int)blockIdx.y;
    const int chunk = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int w_begin = xcd_block(blockIdx.x, gridDim.x) * chunk;
    const int w_end = min(p.total_tiles, w_begin + chunk);
    if (w_begin >= w_end) return;
    struct Tile { int n, y0, x0, row; };
    auto advance = [&](const Tile& t) {
        Tile u = t;
        u.x0 += 16; u.row += 1;
        if (u.x0 == p.W) { u.x0 = 0; u.y0 += 16; if (u.y0 == p.H) { u.y0 = 0; u.row = 0; u.n += 1; } }
        return u;
    };
    auto is_edge = [&](const Tile& t) { return t.y0 == 0 || t.x0 == 0 || t.y0 + 16 == p.H || t.x0 + 16 == p.W; };
    // ---- DMA geometry of this thread: round k of its wave fills unit (4k + wave) * 64 + lane
    int st_rel[NRT];                             // float offset of the unit's source from the halo origin
    int st_yx[NRT];                              // (ly << 8) | lx, or -1: no pixel behind this unit
#pragma unroll
    for (int k = 0; k < NRT; ++k) {
        const int u = (4 * k + wave) * 64 + lane, ly = u / RU, w = u - ly * RU, lx = w >> 2, ch = w & 3;
        const bool valid = u < NU && w < 72;
        st_yx[k] = valid ? (ly << 8) | lx : -1;
        st_rel[k] = valid ? (ly * p.W + lx) * p.C0 + ch * 4 : 0;
    }
    auto tile_mask = [&](const Tile& t) {        // bit k: the unit of round k is a pixel inside the image
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < NRT; ++k) {
            const int ly = st_yx[k] >> 8, lx = st_yx[k] & 0xff;
            if (st_yx[k] >= 0 && (unsigned)(t.y0 + ly - 1) < (unsigned)p.H && (unsigned)(t.x0 + lx - 1) < (unsigned)p.W) m |= 1u << k;
        }
        return m;
    };
    const float* wgrp = p.wpk + (size_t)g * nblk * SEG;
    auto dma_item = [&](const Tile& t, unsigned mask, int cb, int buf) {
        const float* src = p.src0 + ((size_t)((t.n * p.H + t.y0 - 1) * p.W + t.x0 - 1)) * p.C0 + cb * 16;      // halo origin (only masked units use it)
#pragma unroll
        for (int k = 0; k < NRT; ++k) {
            if (4 * k + wave < NRD) {            // wave-uniform
                const float* ptr = (mask >> k) & 1u ? src + st_rel[k] : p.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ptr,
                                                 (__attribute__((address_space(3))) void*)(sA + buf * IMGF + (4 * k + wave) * 256), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {            // the item's 16 KB weight block: 16 pieces, four per wave
            const int piece = wave + 4 * j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wgrp + (size_t)cb * SEG + piece * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(sB + buf * SEG + piece * 256), 16, 0, 0);
        }
    };
    // A operand: the 4x4 input patch of this lane's Winograd tile (halo coordinates), k slot kq -- as in conv3x3_wino
    const int wty = ((i16 >> 3) & 1) * 2 + ((i16 >> 1) & 1), wtx = ((i16 >> 2) & 1) * 2 + (i16 & 1);
    const int prow = (wave >> 1) * 8 + 2 * wty, pcol = (wave & 1) * 8 + 2 * wtx;      // halo row / column of the patch origin
    const int pbase = prow * RS + pcol * 16 + kq * 4;
    const int bbase = (kq * 16 + i16) * 4;

    f32x4 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue geometry and code: conv3x3_wino's, NT = 1
    const int xj = lane & 3, cq4 = ((lane >> 2) & 3) * 4;
    const int prow0 = (wave >> 1) * 8 + ((lane >> 5) & 1) * 4, pcol0 = (wave & 1) * 8 + ((lane >> 4) & 1) * 4;
    const int co0 = g * 16;
    const unsigned lane_out = (unsigned)((prow0 * p.W + pcol0 + xj) * p.Cout + co0 + cq4);
    const unsigned lane_nz = (unsigned)(prow0 * p.W + pcol0);
    float4 nzs[EPI == EPI_SYNTH ? 4 : 1];
    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
    if (EPI == EPI_SYNTH) { e0 = p.nscale[co0 + i16]; e1 = p.nbias[co0 + i16]; }
    if (EPI == EPI_DEC) { e2 = p.bn_s[co0 + i16]; e3 = p.bn_beta[co0 + i16]; }
    auto epilogue_loads = [&](const Tile& tc) {
        if (EPI == EPI_SYNTH) {
            const float* nz = p.noise + ((size_t)(tc.n * p.H + tc.y0) * p.W + tc.x0) + lane_nz;
#pragma unroll
            for (int r = 0; r < 4; ++r) nzs[r] = *reinterpret_cast<const float4*>(nz + r * p.W);
        }
    };
    const int s2 = stat_s2(p.H * p.W);
    unsigned long long dI1 = 0ull, dI2 = 0ull;
    auto flush_stats = [&](const Tile& t) {
        unsigned long long I1 = dI1, I2 = dI2;
        I1 += shfl_xor_u64(I1, 16); I2 += shfl_xor_u64(I2, 16);
        I1 += shfl_xor_u64(I1, 32); I2 += shfl_xor_u64(I2, 32);
        if (lane < 16) {
            StatPart* a = p.partials + ((size_t)t.n * p.prow + (blockIdx.x & (kDirectRows - 1))) * p.Cout + co0 + i16;
            atomicAdd(&a->s1, I1);
            atomicAdd(&a->s2, I2);
        }
        dI1 = dI2 = 0ull;
    };
    auto epilogue = [&](const Tile& tc) {
        const size_t ubase = ((size_t)(tc.n * p.H + tc.y0) * p.W + tc.x0) * p.Cout;
        f32x4 s0[4], s1[4], m[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) m[f] = acc[f];
        mfma_settle(m);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = add4(add4(m[j], m[4 + j]), m[8 + j]);
            s1[j] = sub4(sub4(m[4 + j], m[8 + j]), m[12 + j]);
        }
        const f32x4 y00 = add4(add4(s0[0], s0[1]), s0[2]), y01 = sub4(sub4(s0[1], s0[2]), s0[3]);
        const f32x4 y10 = add4(add4(s1[0], s1[1]), s1[2]), y11 = sub4(sub4(s1[1], s1[2]), s1[3]);
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b1 = r >> 1;
            const f32x4& ya = (r & 1) ? y10 : y00;
            const f32x4& yb = (r & 1) ? y11 : y01;
            float v[4] = {ya[2 * b1], yb[2 * b1], ya[2 * b1 + 1], yb[2 * b1 + 1]};
            if (EPI == EPI_SYNTH) {
                const float nzv[4] = {nzs[r].x, nzs[r].y, nzs[r].z, nzs[r].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = e0 * nzv[k];
                    v[k] = lrelu((v[k] + t) + e1);
                }
                const float sq = (v[0] + v[1]) + (v[2] + v[3]);
                const float qq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                dI1 += to_fixed(sq, kStatScale1);
                dI2 += to_fixed_sq(qq, s2);
            }
            if (EPI == EPI_DEC) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = lrelu(fmaf(v[k], e2, e3));
            }
            const f32x4 vt = quad_transpose(v[0], v[1], v[2], v[3], xj);
            *reinterpret_cast<f32x4*>(p.out + ubase + (size_t)r * p.W * p.Cout + lane_out) = vt;
        }
    };

    auto wino_item = [&](auto edge_tag, int buf, const f32x4& cA, const f32x4& cB, unsigned rowin, unsigned colin) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        __builtin_amdgcn_s_setprio(GSA_MFMA_PRIO);
        const float* a_img = sA + buf * IMGF + pbase;
        const float* b_img = sB + buf * SEG + bbase;
        f32x4 V[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[r] = *reinterpret_cast<const f32x4*>(a_img + r * RS + c * 16);
                if (AFF) {                       // AdaIN: fmaf(x, A, B) on the lane's four channels -- what the staging did before the LDS write
                    const f32x2 lo = fma2v(d[r].xy, cA.xy, cB.xy), hi = fma2v(d[r].zw, cA.zw, cB.zw);
                    d[r] = f32x4{lo.x, lo.y, hi.x, hi.y};
                    if (EDGE) { if (!((rowin >> r) & (colin >> c) & 1u)) d[r] = f32x4{0.f, 0.f, 0.f, 0.f}; }      // zero padding stays zero
                }
            }
            V[0 * 4 + c] = sub4(d[0], d[2]);
            V[1 * 4 + c] = add4(d[1], d[2]);
            V[2 * 4 + c] = sub4(d[2], d[1]);
            V[3 * 4 + c] = sub4(d[1], d[3]);
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
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            f32x4 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_img + (fb * 4 + j) * 256);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[fb * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[fb * 4 + j][cg], b[j][cg], acc[fb * 4 + j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- items (tile, channel block): item `it` is multiplied out of buffer it & 1 while the DMA of item it+1 fills the other
    const int total_items = (w_end - w_begin) * nblk;
    Tile tc;
    {
        const int tx = w_begin % p.tiles_x, r = w_begin / p.tiles_x;
        tc.x0 = tx * 16; tc.y0 = (r % p.tiles_y) * 16; tc.n = r / p.tiles_y; tc.row = (r % p.tiles_y) * p.tiles_x + tx;
    }
    int cb = 0;
    unsigned mask_c = tile_mask(tc);
    dma_item(tc, mask_c, 0, 0);
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, sl = 0, sm = 0, se = 0, sb = 0;
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)sl; (void)sm; (void)se; (void)sb;
    for (int it = 0; it < total_items; ++it) {
        TICK(k0);
        const bool has_next = it + 1 < total_items;
        // AdaIN coefficients of the lane's four channels 16cb + 4kq .. +3 (tiny, L2-resident; issued before the barrier's wait)
        f32x4 cA = f32x4{1.f, 1.f, 1.f, 1.f}, cB = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned rowin = 0xf, colin = 0xf;
        if (AFF) {
            const f32x4* ap = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)tc.n * p.C0 + cb * 16 + kq * 4);
            const f32x4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
            cA = f32x4{a0[1], a1[1], a2[1], a3[1]};
            cB = f32x4{a0[2], a1[2], a2[2], a3[2]};
        }
        const bool edge = is_edge(tc);
        if (AFF && edge) {
            rowin = 0; colin = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((unsigned)(tc.y0 + prow + r - 1) < (unsigned)p.H) rowin |= 1u << r;
                if ((unsigned)(tc.x0 + pcol + r - 1) < (unsigned)p.W) colin |= 1u << r;
            }
        }
        GSA_DMA_SYNC();                         // item `it` has landed; every wave has left the buffer the next DMA overwrites
        TICK(k1);
        Tile tn = tc; int cbn = cb + 1; unsigned mask_n = mask_c;
        if (cbn == nblk) { cbn = 0; if (has_next) { tn = advance(tc); mask_n = tile_mask(tn); } }
        if (has_next) dma_item(tn, mask_n, cbn, (it + 1) & 1);
        if (cb == nblk - 1) epilogue_loads(tc);
        TICK(k2);
        if (AFF && edge) wino_item(std::true_type{}, it & 1, cA, cB, rowin, colin);
        else wino_item(std::false_type{}, it & 1, cA, cB, rowin, colin);
        TICK(k3);
        if (cb == nblk - 1) {
            epilogue(tc);
            if (EPI == EPI_SYNTH && (!has_next || tn.n != tc.n)) flush_stats(tc);
        }
        TICK(k4);
        TSUM(sb, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4);
        tc = tn; cb = cbn; mask_c = mask_n;
    }
    TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
    TFLUSH(12, (unsigned long long)total_items); TFLUSH(15, 1ull);
}

// ------------------------------------------------------------------------------------------
// conv3x3 (pad 1, stride 1, one source) in WINOGRAD F(4x4, 3x3) form on v_mfma_f32_16x16x4_f32 (round 4).
//
//     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      per 4x4 OUTPUT tile from its 6x6 input patch:
//     36 products per 16 outputs (2.25 per output) where F(2x2,3x3) needs 64 (4 per output) and the direct form 144 (9).
//
// Thirty-six GEMMs  M_f[cout][tile] = sum_c U_f[c][cout] V_f[tile][c],  f = 6i+j:  the weights are the A operand (M = 16 output
// channels), the transformed patches the B operand (N = the 16 tiles of a 16x16 output region), K = input channels -- 72 MFMAs
// per 8-channel block and 256 output pixels where conv3x3_wino issues 128.  Canonical arithmetic: oracle/c/gsa_oracle.c
// conv3x3_wino43 (transform op order with its fmaf forms; U = G g G^T in double on the host, rounded once; each M_f one k-ordered
// fmaf chain = the MFMA; K order of these layers: 8-channel blocks): reproduced BIT FOR BIT.  Static rule: conv_uses_wino43.
// OPT-IN (GSA_WINO43=1): measured 15-20 % slower than conv3x3_wino on MI355X -- DESIGN.md section 4 "Round 4" has the five forms
// this kernel went through, the stamps and the reasons; the body below is the last and fastest of them.
//
// One wave = one 16x16 output region; lane (i16, kq) owns tile (ty, tx) = (i16 >> 2, i16 & 3) and k slot kq, and -- weights being
// the A operand -- leaves the MFMAs with FOUR CONSECUTIVE OUTPUT CHANNELS (4*kq + r) of ITS tile in an accumulator vector: the
// output transform works on channel vectors, every output pixel is one 16-byte NHWC store, the statistics' x-quads are in-lane.
// 36 accumulator vectors (144 registers) do not leave room for two waves per SIMD: the workgroup is FOUR waves, one per SIMD, with
// the whole register file (launch bound 1); hipcc keeps the accumulators in the accumulation registers and the transforms in the
// vector registers only when nothing else competes for them, hence NO staging registers: both operands reach LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KB per wave-instruction), the image wave-private, the weight panel shared by the four waves, both
// double-buffered, ONE barrier per item; the next item's DMA is issued between the MFMA groups of the current one.
// blockIdx.y = output-channel group; a workgroup walks a contiguous (XCD-aware) range of 4-tile items of its group.
#ifndef GSA_W43_PK
#define GSA_W43_PK 1
#endif
__device__ __forceinline__ f32x2 pk_fma2s(f32x2 a, f32x2 c, f32x2 b) {      // a * c + b, c = a constant pair in scalar registers
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c), "v"(b));
    return r;
}
// The F(4x4,3x3) transforms on channel PAIRS (v_pk_*_f32: one instruction per two channels; constants as a scalar register pair).
// c is a power of two or 5; fmaf(c, a, b), a + b, a - b exactly as the oracle's wino43_in / wino43_out write them.
__device__ __forceinline__ f32x2 fma2c(float c, f32x2 a, f32x2 b) {
#if GSA_W43_PK
    return pk_fma2s(a, f32x2{c, c}, b);
#else
    return f32x2{fmaf(c, a.x, b.x), fmaf(c, a.y, b.y)};
#endif
}
__device__ __forceinline__ f32x2 add2c(f32x2 a, f32x2 b) {
#if GSA_W43_PK
    return pk_add2(a, b);
#else
    return a + b;
#endif
}
__device__ __forceinline__ f32x2 sub2c(f32x2 a, f32x2 b) {
#if GSA_W43_PK
    return pk_sub2(a, b);
#else
    return a - b;
#endif
}
// input transform of one 6-vector (oracle wino43_in): 12 operations
__device__ __forceinline__ void w43_in(const f32x2 (&d)[6], f32x2 (&t)[6]) {
    const f32x2 a = fma2c(-4.0f, d[2], d[4]), b = fma2c(-4.0f, d[1], d[3]);
    const f32x2 c = sub2c(d[4], d[2]), e = sub2c(d[3], d[1]);
    t[0] = fma2c(4.0f, d[0], fma2c(-5.0f, d[2], d[4]));
    t[1] = add2c(a, b);
    t[2] = sub2c(a, b);
    t[3] = fma2c(2.0f, e, c);
    t[4] = fma2c(-2.0f, e, c);
    t[5] = fma2c(4.0f, d[1], fma2c(-5.0f, d[3], d[5]));
}
// output transform of one 6-vector (oracle wino43_out): 10 operations; on the lane's FOUR output channels (two pairs)
__device__ __forceinline__ void w43_out2(const f32x2 (&m)[6], f32x2 (&y)[4]) {
    const f32x2 pp = add2c(m[1], m[2]), qq = sub2c(m[1], m[2]), rr = add2c(m[3], m[4]), ss = sub2c(m[3], m[4]);
    y[0] = add2c(add2c(m[0], pp), rr);
    y[1] = fma2c(2.0f, ss, qq);
    y[2] = fma2c(4.0f, rr, pp);
    y[3] = add2c(fma2c(8.0f, ss, qq), m[5]);
}
__device__ __forceinline__ void w43_out(const f32x4 (&m)[6], f32x4 (&y)[4]) {
    f32x2 lo[6], hi[6], ylo[4], yhi[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) { lo[i] = m[i].xy; hi[i] = m[i].zw; }
    w43_out2(lo, ylo);
    w43_out2(hi, yhi);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = f32x4{ylo[i].x, ylo[i].y, yhi[i].x, yhi[i].y};
}
__device__ __forceinline__ void valu_settle6(f32x2 (&v)[6]) {
#if GSA_W43_PK
    asm("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));      // VALU result -> MFMA operand: 2 wait states
#endif
}
__device__ __forceinline__ void mfma_settle18(f32x4* a) {      // the accumulators stay where the MFMAs left them (accumulation registers)
#if GSA_W43_PK
    asm("s_nop 7\n\ts_nop 3"
        : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7]), "+a"(a[8]),
          "+a"(a[9]), "+a"(a[10]), "+a"(a[11]), "+a"(a[12]), "+a"(a[13]), "+a"(a[14]), "+a"(a[15]), "+a"(a[16]), "+a"(a[17]));
#endif
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void conv3x3_wino43(ConvParams p) {
    // Items are 8-CHANNEL blocks (the F(4x4,3x3) layers have a K order of their own: 8-channel blocks ascending, inside a block MFMA
    // j = 0, 1 with k slot kq <-> channel 8b + 2kq + j -- oracle conv3x3_wino43): a lane's operand pair is then ONE contiguous 8-byte
    // piece of the NHWC tensor, a pixel of a block is two 16-byte units, and BOTH operands reach LDS by LDS-DMA with no staging
    // registers at all -- what the 144 accumulators leave of the register file goes to the transforms.
    //   image  (per wave, double-buffered): 16-byte unit u(ly, lx, half) = half * 378 + ly * 21 + (lx & 3) * 5 + (lx >> 2): two planes
    //          (the two 16-byte halves of a pixel's 32 bytes), a row stored as four classes lx & 3 of five cells, row stride 21 units.
    //          A tile step in x is then ONE unit and a tile step in y 84 = 4 (mod 16) units: the 16 tiles of an 8-byte patch read
    //          fall on 16 different 16-byte slots of the 256-byte bank line -- every read conflict-free (the linear [pixel][32 B]
    //          image was 8-way conflicted: a tile step of 4 pixels = 128 B = half the banks; measured 0.44 ms on g.64.conv_2).
    //          One DMA instruction fills 64 consecutive units from 64 per-lane source addresses (the layout costs address
    //          arithmetic once per kernel); a pixel outside the image reads a page of zeros.
    //          RAW values: AdaIN is applied after the LDS read (a pixel outside the image must stay 0: edge tiles mask it);
    //   weights (per workgroup, double-buffered): [f36][kq][16 couts][2] = 18 KB per (16 couts, 8-channel block).
    constexpr int LH = 18, LW = 18, SU = 21, PLU = LH * SU;  // units per image row / per plane
    constexpr int NRI = (2 * PLU + 63) / 64;                 // 12 image DMA instructions per item and wave
    constexpr int IMG = NRI * 64 * 4;                        // floats per image buffer (768 units)
    constexpr int SEG = 36 * 128;                            // U floats per (16 couts, 8-channel block)
    constexpr int NWP = SEG * 4 / 1024;                      // 18 weight pieces of 1 KB per item, shared by the four waves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    float* sA = smem + wave * (2 * IMG);          // [2][IMG]
    float* sB = smem + 4 * (2 * IMG);             // [2][SEG]
    const int nblk = p.C0 >> 3;
    const int g = blockIdx.y;
    // contiguous range of the group's 4-tile items, order (n, ty, tx)
    const int witems = (p.total_tiles + 3) >> 2;
    const int wchunk = (witems + (int)gridDim.x - 1) / (int)gridDim.x;
    const int w_begin = xcd_block(blockIdx.x, gridDim.x) * wchunk;
    const int w_end = min(witems, w_begin + wchunk);
    if (w_begin >= w_end) return;
    const int tpi = p.tiles_x * p.tiles_y;
    const bool has_aff = p.aff0 != nullptr;

    // ---- DMA geometry: round k fills units 64k .. 64k+63; this lane's unit 64k + lane holds (ly, lx, half) by the inverse of u(...)
    int st_rel[NRI];                              // float offset of the unit's source from the halo origin
    int st_yx[NRI];                               // (ly << 8) | lx, or -1: the unit holds no pixel (row padding, plane tail)
#pragma unroll
    for (int k = 0; k < NRI; ++k) {
        const int u = 64 * k + lane, half = u >= PLU ? 1 : 0, w = u - half * PLU;
        const int ly = w / SU, xw = w - ly * SU, cls = xw / 5, cell = xw - cls * 5, lx = 4 * cell + cls;
        const bool valid = u < 2 * PLU && xw < 20 && lx < LW;
        st_yx[k] = valid ? (ly << 8) | lx : -1;
        st_rel[k] = valid ? (ly * p.W + lx) * p.C0 + half * 4 : 0;
    }
    auto tile_mask = [&](int y0, int x0) {        // bit k: the unit of round k is a pixel inside the image
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < NRI; ++k) {
            const int ly = st_yx[k] >> 8, lx = st_yx[k] & 0xff;
            if (st_yx[k] >= 0 && (unsigned)(y0 + ly - 1) < (unsigned)p.H && (unsigned)(x0 + lx - 1) < (unsigned)p.W) m |= 1u << k;
        }
        return m;
    };
    // ---- patch geometry: lane (tile (ty, tx), k slot kq) reads 8 bytes of pixel (4ty + r, 4tx + c)
    const int ty = i16 >> 2, tx = i16 & 3;
    const int abase = ((kq >> 1) * PLU + (4 * ty) * SU + tx) * 4 + (kq & 1) * 2;      // + (r * SU + (c & 3) * 5 + (c >> 2)) * 4 per patch pixel
    const int bbase = (kq * 16 + i16) * 2;
    const float* wgrp = p.wpk + (size_t)g * nblk * SEG;

    f32x4 acc[36];
#pragma unroll
    for (int f = 0; f < 36; ++f) asm volatile("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0"
                                              : "=a"(acc[f][0]), "=a"(acc[f][1]), "=a"(acc[f][2]), "=a"(acc[f][3]));

    struct WT { int n, y0, x0; bool on, edge; unsigned mask; unsigned rowin, colin; };
    auto tile_of = [&](int wi) {
        WT t;
        const int idx = wi * 4 + wave;
        t.on = idx < p.total_tiles;
        const int id = t.on ? idx : 0;
        t.n = id / tpi;
        const int r = id - t.n * tpi;
        t.y0 = (r / p.tiles_x) * 16; t.x0 = (r % p.tiles_x) * 16;
        t.edge = t.y0 == 0 || t.x0 == 0 || t.y0 + 16 == p.H || t.x0 + 16 == p.W;      // wave-uniform
        t.mask = tile_mask(t.y0, t.x0);
        t.rowin = 0; t.colin = 0;                 // bits r / c: row 4ty + r / column 4tx + c of this lane's patch lies inside the image
#pragma unroll
        for (int r6 = 0; r6 < 6; ++r6) {
            if ((unsigned)(t.y0 + 4 * ty + r6 - 1) < (unsigned)p.H) t.rowin |= 1u << r6;
            if ((unsigned)(t.x0 + 4 * tx + r6 - 1) < (unsigned)p.W) t.colin |= 1u << r6;
        }
        return t;
    };
    // Everything of item (t, cb) reaches buffer `buf` by LDS-DMA in NDMA = 17 steps: 12 image rounds (the wave's own halo tile) and this
    // wave's 5 (of 18) weight pieces.  Issued back to back they cost ~3500 cycles per item (stamps: ~210 cycles per instruction with
    // nothing beside them); the steady state therefore issues them three at a time BETWEEN the MFMA groups of the previous item,
    // where the matrix pipe works on while the wave issues memory instructions.
    constexpr int NDMA = NRI + (NWP + 3) / 4;
    auto dma_step = [&](const WT& t, int cb, int buf, int sidx) {
        if (sidx < NRI) {
            if (!t.on) return;
            const float* src = p.src0 + ((size_t)((t.n * p.H + t.y0 - 1) * p.W + t.x0 - 1)) * p.C0 + cb * 8;      // halo origin (may lie before the tensor: only masked units use it)
            const float* ptr = (t.mask >> sidx) & 1u ? src + st_rel[sidx] : p.zeros;
#ifdef GSA_DBG_HOOKS
            if (p.dbg & 16) ptr = p.src0 + (size_t)((t.n * p.H + t.y0) * p.W) * p.C0 + (sidx * 64 + lane) * 4;      // timing only: a DENSE 1 KB read per instruction (no partial cache lines)
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ptr,
                                             (__attribute__((address_space(3))) void*)(sA + buf * IMG + sidx * 256), 16, 0, 0);
        } else {
            const int piece = wave + 4 * (sidx - NRI);
            if (piece < NWP)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wgrp + (size_t)cb * SEG + lane * 4 + piece * 256),
                                                 (__attribute__((address_space(3))) void*)(sB + buf * SEG + piece * 256), 16, 0, 0);
        }
    };
    auto load_item = [&](const WT& t, int cb, int buf) {
#pragma unroll
        for (int sidx = 0; sidx < NDMA; ++sidx) dma_step(t, cb, buf, sidx);
    };

    // ---- epilogue: lane -> tile (ty, tx), output channels co4 .. co4+3
    const int co4 = g * 16 + kq * 4;
    f32x4 e0 = f32x4{0.f, 0.f, 0.f, 0.f}, e1 = e0;
    if (EPI == EPI_SYNTH) { e0 = *reinterpret_cast<const f32x4*>(p.nscale + co4); e1 = *reinterpret_cast<const f32x4*>(p.nbias + co4); }
    if (EPI == EPI_DEC) { e0 = *reinterpret_cast<const f32x4*>(p.bn_s + co4); e1 = *reinterpret_cast<const f32x4*>(p.bn_beta + co4); }
    const int s2 = stat_s2(p.H * p.W);
    unsigned long long dI1[4], dI2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dI1[r] = dI2[r] = 0ull;
    auto flush_stats = [&](int n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned long long I1 = dI1[r], I2 = dI2[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { I1 += shfl_xor_u64(I1, m); I2 += shfl_xor_u64(I2, m); }
            if (i16 == 0) {
                StatPart* a = p.partials + ((size_t)n * p.prow + (blockIdx.x & (kDirectRows - 1))) * p.Cout + co4 + r;
                atomicAdd(&a->s1, I1);
                atomicAdd(&a->s2, I2);
            }
            dI1[r] = dI2[r] = 0ull;
        }
    };
    auto epilogue = [&](const WT& t) {
        float4 nzs[EPI == EPI_SYNTH ? 4 : 1];
        if (EPI == EPI_SYNTH) {
            const float* nz = p.noise + ((size_t)(t.n * p.H + t.y0 + 4 * ty) * p.W + t.x0 + 4 * tx);
#pragma unroll
            for (int i = 0; i < 4; ++i) nzs[i] = *reinterpret_cast<const float4*>(nz + i * p.W);
        }
        // output transform Y = A^T M A: down the columns of M, then along the rows; vectors = the lane's four output channels
        mfma_settle18(acc);
        mfma_settle18(acc + 18);
        f32x4 sr[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x4 m[6] = {acc[j], acc[6 + j], acc[12 + j], acc[18 + j], acc[24 + j], acc[30 + j]};
            f32x4 y[4];
            w43_out(m, y);
#pragma unroll
            for (int i = 0; i < 4; ++i) sr[i][j] = y[i];
        }
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0"
                                                  : "=a"(acc[f][0]), "=a"(acc[f][1]), "=a"(acc[f][2]), "=a"(acc[f][3]));
        float* orow = p.out + ((size_t)(t.n * p.H + t.y0 + 4 * ty) * p.W + t.x0 + 4 * tx) * p.Cout + co4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v[4];
            w43_out(sr[i], v);
            if (EPI == EPI_SYNTH) {
                const float nzv[4] = {nzs[i].x, nzs[i].y, nzs[i].z, nzs[i].w};
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float tn = e0[r] * nzv[x];
                        v[x][r] = lrelu((v[x][r] + tn) + e1[r]);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sq = (v[0][r] + v[1][r]) + (v[2][r] + v[3][r]);
                    const float qq = (v[0][r] * v[0][r] + v[1][r] * v[1][r]) + (v[2][r] * v[2][r] + v[3][r] * v[3][r]);
                    dI1[r] += to_fixed(sq, kStatScale1);
                    dI2[r] += to_fixed_sq(qq, s2);
                }
            }
            if (EPI == EPI_DEC) {
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[x][r] = lrelu(fmaf(v[x][r], e0[r], e1[r]));
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) *reinterpret_cast<f32x4*>(orow + ((size_t)i * p.W + x) * p.Cout) = v[x];
        }
    };

    // The MFMAs are inline asm with the accumulator as an in/out ACCUMULATION register ("+a"): left to itself hipcc put the 144
    // accumulators in vector registers and copied every result out.  Hazards by hand, as in conv3x3_wino: V goes through valu_settle6
    // (2 wait states between a vector-ALU result and the MFMA reading it); the two MFMAs of a frequency (channels j = 0, 1: a dependent
    // chain through the accumulator) are issued six MFMAs apart; the accumulators go through mfma_settle18 before the epilogue reads them.
#ifndef GSA_W43_ASM_MFMA
#define GSA_W43_ASM_MFMA 0      // measured: the builtin is faster (0.44 vs 0.62 ms on g.64.conv_2) once nothing spills; the asm form needs settles of its own
#endif
    auto mfma43 = [&](f32x4& c, float a, float b) {
#if GSA_W43_ASM_MFMA
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#else
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
    };
    auto compute = [&](auto aff_tag, auto edge_tag, int buf, const WT& t, f32x2 cA, f32x2 cB, bool has_next, const WT& tn, int cbn) {
        constexpr bool AFFc = decltype(aff_tag)::value, EDGE = decltype(edge_tag)::value;
        __builtin_amdgcn_s_setprio(GSA_MFMA_PRIO);
        const float* a_img = sA + buf * IMG + abase;
        const float* b_img = sB + buf * SEG + bbase;
        // input transform V = B^T d B: down the columns of the 6x6 patch (column c+1 is read while column c is transformed), then along the rows
        f32x2 tt[6][6];
        {
            f32x2 d[2][6];
#pragma unroll
            for (int r = 0; r < 6; ++r) d[0][r] = *reinterpret_cast<const f32x2*>(a_img + (r * SU) * 4);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if (c < 5) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) d[(c + 1) & 1][r] = *reinterpret_cast<const f32x2*>(a_img + (r * SU + ((c + 1) & 3) * 5 + ((c + 1) >> 2)) * 4);
                }
                f32x2 x[6], col[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    x[r] = d[c & 1][r];
                    if (AFFc) x[r] = fma2v(x[r], cA, cB);                                   // AdaIN: fmaf(x, A, B) per channel
                    if (AFFc && EDGE) { if (!((t.rowin >> r) & (t.colin >> c) & 1u)) x[r] = f32x2{0.f, 0.f}; }      // zero padding stays zero
                }
                w43_in(x, col);
#pragma unroll
                for (int i = 0; i < 6; ++i) tt[i][c] = col[i];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the weights of frequency row i+1 are read while row i is multiplied (a lone wave per SIMD has nobody to hide its LDS latency)
        f32x2 bq[2][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) bq[0][j] = *reinterpret_cast<const f32x2*>(b_img + j * 128);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x2 V[6];
            w43_in(tt[i], V);
            valu_settle6(V);
            if (i < 5) {
#pragma unroll
                for (int j = 0; j < 6; ++j) bq[(i + 1) & 1][j] = *reinterpret_cast<const f32x2*>(b_img + ((i + 1) * 6 + j) * 128);
            }
#if GSA_W43_ASM_MFMA
            // hipcc moves accumulators between the two register files around these statements with copies of its own and does not know
            // that the statements are MFMAs: its v_accvgpr_write must have settled before they read the accumulator (2 wait states),
            // and its v_accvgpr_read must not come within 11 wait states of the last of them -- found as wrong results, not in the ISA manual
            asm volatile("s_nop 1" : "+a"(acc[i * 6]), "+a"(acc[i * 6 + 1]), "+a"(acc[i * 6 + 2]), "+a"(acc[i * 6 + 3]), "+a"(acc[i * 6 + 4]), "+a"(acc[i * 6 + 5]));
#endif
#pragma unroll
            for (int j = 0; j < 6; ++j) mfma43(acc[i * 6 + j], bq[i & 1][j].x, V[j].x);
#pragma unroll
            for (int j = 0; j < 6; ++j) mfma43(acc[i * 6 + j], bq[i & 1][j].y, V[j].y);
#if GSA_W43_ASM_MFMA
            asm volatile("s_nop 7\n\ts_nop 3" : "+a"(acc[i * 6]), "+a"(acc[i * 6 + 1]), "+a"(acc[i * 6 + 2]), "+a"(acc[i * 6 + 3]), "+a"(acc[i * 6 + 4]), "+a"(acc[i * 6 + 5]));
#endif
            if (has_next) {                      // the next item's DMA, three instructions behind each MFMA group (wave-uniform branch)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (3 * i + q < NDMA) dma_step(tn, cbn, buf ^ 1, 3 * i + q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- items (4-tile item, 8-channel block).  Item `it` is multiplied out of buffer it & 1 while the DMA of item it+1 fills the other.
    const int total_items = (w_end - w_begin) * nblk;
    int wi = w_begin, cb = 0;
    WT tc = tile_of(wi);
    load_item(tc, 0, 0);
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, sw = 0, sl = 0, sm = 0, se = 0, sb = 0;
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)sw; (void)sl; (void)sm; (void)se; (void)sb;
    for (int it = 0; it < total_items; ++it) {
        TICK(k0);
        // AdaIN coefficients (A, B) of the lane's channel pair 8cb + 2kq, +1 (tiny, L2-resident; issued before the barrier's wait)
        f32x2 cA = f32x2{1.f, 1.f}, cB = f32x2{0.f, 0.f};
        if (has_aff && tc.on) {
            const f32x4* ap = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)tc.n * p.C0 + cb * 8 + kq * 2);
            const f32x4 a0 = ap[0], a1 = ap[1];
            cA = f32x2{a0[1], a1[1]}; cB = f32x2{a0[2], a1[2]};
        }
        GSA_DMA_SYNC();                         // item `it` has landed (the compiler waits for this wave's DMA before the barrier); buffer (it+1) & 1 is free
        TICK(k1);
        int wi2 = wi, cb2 = cb + 1;
        WT tn = tc;
        if (cb2 == nblk) { cb2 = 0; ++wi2; if (wi2 < w_end) tn = tile_of(wi2); }
        const bool has_next = it + 1 < total_items;
        TICK(k2);
        if (tc.on) {
            if (!has_aff) compute(std::false_type{}, std::false_type{}, it & 1, tc, cA, cB, has_next, tn, cb2);
            else if (tc.edge) compute(std::true_type{}, std::true_type{}, it & 1, tc, cA, cB, has_next, tn, cb2);
            else compute(std::true_type{}, std::false_type{}, it & 1, tc, cA, cB, has_next, tn, cb2);
        } else if (has_next) {
            load_item(tn, cb2, (it + 1) & 1);      // an idle wave (no tile in this item) still brings its share of the weights
        }
        TICK(k3);
        if (cb == nblk - 1 && tc.on) {
            epilogue(tc);
            if (EPI == EPI_SYNTH && (!has_next || !tn.on || tn.n != tc.n)) flush_stats(tc.n);
        }
        TICK(k4);
        TSUM(sb, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4);
        wi = wi2; cb = cb2; tc = tn;
    }
    TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
    TFLUSH(12, (unsigned long long)total_items); TFLUSH(15, 1ull);
}

// ------------------------------------------------------------------------------------------
#endif      // GSA_EXPERIMENTS

// Winograd F(2x2, 2x2) form of the stride-2 parity-class convolutions (round 3; canonical arithmetic:
// oracle/c/gsa_oracle.c deconv4x4s2_wino).  A parity class (py, px) of the 4x4 stride-2 transposed
// convolution is a 2x2-tap stride-1 convolution; per 2x2 class outputs the Winograd form needs 9 products instead of 16:
//     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A,   B^T = [1 -1 0; 0 1 0; 0 -1 1],  G = [1 0; 1 1; 0 1],  A^T = [1 1 0; 0 1 1]
// i.e. nine GEMMs M_f[tile][o] = sum_c V_f[tile][c] U_f[c][o] with M = 16 Winograd tiles (the 4x4 grid of 2x2 input blocks of
// an 8x8 input tile), N = 16 output channels, K = channels: 36 MFMAs per 16-channel block where the direct form issues 64.
// Every coefficient is +-1, and U = G g G^T is FIVE fp32 adds of the class's four taps (every add rounded: the canonical
// definition), so the weights in HBM / LDS stay the packed 4x4 kernel -- no second panel, the resident / streamed panel
// sizes of the direct form hold -- and U is formed in registers per (block, 16 output channels).
// Lane (i16, kq): Winograd tile (ty, tx) = (2*b0 + b1, 2*b2 + b3) (b = bits of i16): with a row stride of 10*16+4 floats the
// nine ds_read_b128 of the 3x3 patch are bank-conflict free (brute-forced over the ds_read_b128 lane groups); k slot kq.
// The MFMAs take the weights as the A operand and the tiles as B (D^T): a lane holds FOUR CONSECUTIVE OUTPUT CHANNELS
// (4*(lane>>4) + r) of its tile, so after the output transform each of its 2x2 outputs is one 16-byte NHWC store.
__device__ __forceinline__ int wino22_tile_y(int i16) { return (i16 & 1) * 2 + ((i16 >> 1) & 1); }
__device__ __forceinline__ int wino22_tile_x(int i16) { return ((i16 >> 2) & 1) * 2 + ((i16 >> 3) & 1); }

// one 16-channel block: ap = the lane's patch origin in the LDS image (row py / column px of its tile's halo patch, k slot
// included), bp = the block's weights of this lane ([tap16][kq][16][cg] + (kq*16 + i16)*4), SEG = floats per 16 output channels
// 2 wait states between the packed adds (inline asm: invisible to the compiler's hazard recognizer) and the MFMAs that read
// their results; not volatile -- ordered by its operands only (see valu_settle)
__device__ __forceinline__ void valu_settle8(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f, f32x4& g, f32x4& h) {
    asm("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ void valu_settle5(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e) {
    asm("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}

#ifndef GSA_W22_PK
#define GSA_W22_PK 1
#endif
template <int NT, int SEG>
__device__ __forceinline__ void wino22_block(const float* ap, int RS, const float* bp, int py, int px, f32x4 (&acc)[9][NT]) {
    f32x4 V[9];
    {
        f32x4 d[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) d[r][c] = *reinterpret_cast<const f32x4*>(ap + r * RS + c * 16);
        // rows t0 = d0 - d1, t1 = d1, t2 = d2 - d1; then the same three forms along the columns (packed adds: 24 instructions)
#if GSA_W22_PK
#pragma unroll
        for (int c = 0; c < 3; ++c) { d[0][c] = sub4(d[0][c], d[1][c]); d[2][c] = sub4(d[2][c], d[1][c]); }
#pragma unroll
        for (int r = 0; r < 3; ++r) { V[3 * r] = sub4(d[r][0], d[r][1]); V[3 * r + 1] = d[r][1]; V[3 * r + 2] = sub4(d[r][2], d[r][1]); }
        valu_settle8(V[0], V[1], V[2], V[3], V[5], V[6], V[7], V[8]);       // V[4] = d[1][1] comes straight from LDS
#else
#pragma unroll
        for (int c = 0; c < 3; ++c) { d[0][c] = d[0][c] - d[1][c]; d[2][c] = d[2][c] - d[1][c]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { V[3 * r] = d[r][0] - d[r][1]; V[3 * r + 1] = d[r][1]; V[3 * r + 2] = d[r][2] - d[r][1]; }
#endif
    }
    // taps of the class filter g[a][b] = Wd[3 - py - 2a][3 - px - 2b]
    const int t00 = ((3 - py) * 4 + (3 - px)) * 256, t01 = ((3 - py) * 4 + (1 - px)) * 256;
    const int t10 = ((1 - py) * 4 + (3 - px)) * 256, t11 = ((1 - py) * 4 + (1 - px)) * 256;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x4 U[9];
        U[0] = *reinterpret_cast<const f32x4*>(bp + nt * SEG + t00);
        U[2] = *reinterpret_cast<const f32x4*>(bp + nt * SEG + t01);
        U[6] = *reinterpret_cast<const f32x4*>(bp + nt * SEG + t10);
        U[8] = *reinterpret_cast<const f32x4*>(bp + nt * SEG + t11);
#if GSA_W22_PK
        U[1] = add4(U[0], U[2]);
        U[7] = add4(U[6], U[8]);
        U[3] = add4(U[0], U[6]);
        U[5] = add4(U[2], U[8]);
        U[4] = add4(U[1], U[7]);
        valu_settle5(U[1], U[3], U[4], U[5], U[7]);
#else
        U[1] = U[0] + U[2];
        U[7] = U[6] + U[8];
        U[3] = U[0] + U[6];
        U[5] = U[2] + U[8];
        U[4] = U[1] + U[7];
        __builtin_amdgcn_sched_barrier(0);      // the whole transform ahead of one MFMA stream (no vector-ALU / MFMA switches inside it)
#endif
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int f = 0; f < 9; ++f)
                acc[f][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[f][cg], V[f][cg], acc[f][nt], 0, 0, 0);
    }
}

// output transform Y = A^T M A of one 16-channel output tile: rows s0 = m0 + m1, s1 = m1 + m2, then along the columns
__device__ __forceinline__ void wino22_output(const f32x4 (&m)[9], f32x4 (&y)[2][2]) {
    f32x4 s0[3], s1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { s0[j] = m[j] + m[3 + j]; s1[j] = m[3 + j] + m[6 + j]; }
    y[0][0] = s0[0] + s0[1]; y[0][1] = s0[1] + s0[2];
    y[1][0] = s1[0] + s1[1]; y[1][1] = s1[1] + s1[2];
}

// LeakyReLU(0.2) on four channels (the scalar lrelu per element)
__device__ __forceinline__ f32x4 lrelu4(const f32x4& v) {
    const f32x4 t = v * 0.2f;
    return f32x4{fmaxf(v[0], t[0]), fmaxf(v[1], t[1]), fmaxf(v[2], t[2]), fmaxf(v[3], t[3])};
}

// ------------------------------------------------------------------------------------------
// Stride-2 "parity class" convolution: Deconvolution 4x4 s2 p1 (the reference's fused upscale)
// AND nearest-x2 + conv3x3 in its sub-pixel form (weights pre-summed on the host into the same
// 4x4 stride-2 kernel: 2.25x fewer MACs than 9 taps on the upsampled image).
// A 16x16 output tile splits into 4 parity classes (oy&1, ox&1); each is a 2x2-tap convolution
// over the 10x10 input tile with its own weights.  One wave per class, 4 patches of 4x4 outputs.
template <int NT, int EPI, bool SC, bool BF, bool WINO = false>
__global__ __launch_bounds__(256) void subpixel_mfma(ConvParams p) {
    static_assert(!(WINO && BF), "the Winograd form is fp32 only");
    constexpr int PX = BF ? 8 : 16, TS = BF ? 128 : 256, KQ = BF ? 2 : 4;     // 4-byte slots, as in conv3x3_mfma
    constexpr int LH = 10, LW = 10, RS = LW * PX + (BF || WINO ? 4 : 8);     // WINO: conflict-free 3x3 patch reads at stride-2 tiles
    constexpr int NA = WINO ? 9 : 4;                  // accumulator tiles per 16 output channels: frequencies / 4x4 patches
    constexpr int COUT_T = 16 * NT, SEG = 16 * TS, NB4 = NT * SEG / 4, BIT = (NB4 + 255) / 256;
    constexpr int SIT = (NT * TS / 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = sA + LH * RS;                     // [q][tap16][ci][16][cg]
    float* sS = sB + NT * SEG;                    // SC: [q][ci][16][cg]
    float4* sAff = reinterpret_cast<float4*>(sS + (SC ? NT * TS : 0));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = wave >> 1, px = wave & 1;
    const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x % p.tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;           // output coordinates
    const int iy0 = y0 / 2 - 1, ix0 = x0 / 2 - 1;   // input tile origin (with halo)
    const int g = blockIdx.y, n = blockIdx.z;
    const int i16 = lane & 15, kq = lane >> 4;
    TilePixel tp;
    {
        const int ly = tid / LW, lx = tid % LW;
        const int gy = iy0 + ly, gx = ix0 + lx;
        const bool stage = tid < LH * LW;
        const bool inside = stage && gy >= 0 && gy < p.Hs && gx >= 0 && gx < p.Ws;
        tp.lds = stage ? ly * RS + lx * PX : -1;
        tp.pix = inside ? (n * p.Hs + gy) * p.Ws + gx : -1;
#ifdef GSA_DBG_HOOKS
        if ((p.dbg & 1) && inside) tp.pix = lx & 1;              // timing-only: cache-resident input
#endif
    }
    int abase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        abase[mt] = ((mt >> 1) * 4 + (i16 >> 2) + 1) * RS + ((mt & 1) * 4 + (i16 & 3) + 1) * PX + kq * KQ;
    const int bbase = (kq * 16 + i16) * KQ;
    // fused 1x1 shortcut: on a nearest-upsampled input it is the same value for the 4 sub-pixels of
    // an input pixel, so it is computed ONCE per input pixel and stored at the input resolution
    // (the consumer reads it with resid_up).  The 8x8 input pixels of the tile are 4 patches; wave w
    // takes patch w (A operand = the offset-(0,0) tap of that patch).
    const int asc = ((wave >> 1) * 4 + (i16 >> 2) + 1) * RS + ((wave & 1) * 4 + (i16 & 3) + 1) * PX + kq * KQ;
    // WINO: the lane's 3x3 patch (halo coordinates) of Winograd tile (wty, wtx), shifted by the class
    const int wty = wino22_tile_y(i16), wtx = wino22_tile_x(i16);
    const int pbase = (2 * wty + py) * RS + (2 * wtx + px) * 16 + kq * 4;
    f32x4 acc[NA][NT];
    f32x4 accs[SC ? NT : 1];
#pragma unroll
    for (int mt = 0; mt < NA; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (SC) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accs[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int nblk0 = p.C0 >> 4, nblk = (p.C0 + p.C1) >> 4;
    f32x4 ra[4], rb[BIT], rs[SC ? SIT : 1];
    auto load_block = [&](int cb) {
        const bool first = cb < nblk0;
        const float* src = first ? p.src0 : p.src1;
        const int Cs = first ? p.C0 : p.C1;
        const int coff = (first ? cb : cb - nblk0) * 16;
        load_pixel<BF>(ra, src, Cs, coff, tp);
#pragma unroll
        for (int j = 0; j < BIT; ++j) {
            const int i = min(tid + j * 256, NB4 - 1);
            const int q = i / (SEG / 4), r = i % (SEG / 4);
            rb[j] = reinterpret_cast<const f32x4*>(p.wpk + ((size_t)(g * NT + q) * nblk + cb) * SEG)[r];
        }
        if (SC) {
#pragma unroll
            for (int j = 0; j < SIT; ++j) {
                const int i = min(tid + j * 256, NT * TS / 4 - 1);
                const int q = i / (TS / 4), r = i % (TS / 4);
                rs[j] = reinterpret_cast<const f32x4*>(p.wsc + ((size_t)(g * NT + q) * nblk + cb) * TS)[r];
            }
        }
    };
    auto write_block = [&](int cb) {
        if (cb < nblk0 && p.aff0) store_pixel<true, BF>(sA, ra, sAff + cb * 16, tp);
        else store_pixel<false, BF>(sA, ra, sAff, tp);
#pragma unroll
        for (int j = 0; j < BIT; ++j) reinterpret_cast<f32x4*>(sB)[min(tid + j * 256, NB4 - 1)] = rb[j];
        if (SC) {
#pragma unroll
            for (int j = 0; j < SIT; ++j) reinterpret_cast<f32x4*>(sS)[min(tid + j * 256, NT * TS / 4 - 1)] = rs[j];
        }
    };
    load_block(0);
    if (p.aff0) {
        const float4* ga = reinterpret_cast<const float4*>(p.aff0 + (size_t)n * p.C0);
        for (int i = tid; i < p.C0; i += 256) sAff[i] = ga[i];
        __syncthreads();
    }
    write_block(0);
    __syncthreads();
    for (int cb = 0; cb < nblk; ++cb) {
        load_block(min(cb + 1, nblk - 1));
        __builtin_amdgcn_sched_barrier(0);
        // valid taps of this parity class, ascending ky then kx:
        //   py==0: ky=1 (dy 0), ky=3 (dy -1);   py==1: ky=0 (dy +1), ky=2 (dy 0)
#ifdef GSA_DBG_HOOKS
        if (!(p.dbg & 16)) {
#endif
        if constexpr (WINO) {
            wino22_block<NT, SEG>(sA + pbase, RS, sB + bbase, py, px, acc);
        } else {
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) {
            const int ky = (py ? 0 : 1) + 2 * jy;
            const int dy = py ? (1 - jy) : -jy;
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                const int kx = (px ? 0 : 1) + 2 * jx;
                const int dx = px ? (1 - jx) : -jx;
                const int toff = dy * RS + dx * PX;
                const int tap = ky * 4 + kx;
                if constexpr (BF) {
                    s16x4 a[4], b[NT];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const s16x4*>(sA + abase[mt] + toff);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const s16x4*>(sB + bbase + nt * SEG + tap * TS);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                } else {
                f32x4 a[4], b[NT];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(sA + abase[mt] + toff);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(sB + bbase + nt * SEG + tap * 256);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][cg], b[nt][cg], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        }
#ifdef GSA_DBG_HOOKS
        }
#endif
        if constexpr (SC && BF) {
            const s16x4 as = *reinterpret_cast<const s16x4*>(sA + asc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                accs[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as, *reinterpret_cast<const s16x4*>(sS + bbase + nt * TS), accs[nt], 0, 0, 0);
        } else if constexpr (SC) {   // natural channel order: block, then 4 k-slots of 4 channels
            const f32x4 as = *reinterpret_cast<const f32x4*>(sA + asc);
            f32x4 bs[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bs[nt] = *reinterpret_cast<const f32x4*>(sS + bbase + nt * 256);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    accs[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[cg], bs[nt][cg], accs[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (cb + 1 < nblk) {
            __syncthreads();
#ifdef GSA_DBG_HOOKS
            if (!(p.dbg & 8))
#endif
            write_block(cb + 1);
            __syncthreads();
        }
    }
    if constexpr (WINO) {
        // transposed accumulators: lane = (tile i16, output channels 4*(lane>>4)..+3); each of the tile's 2x2 class outputs is one
        // 16-byte NHWC store
        const int cq = 4 * (lane >> 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = g * COUT_T + nt * 16 + cq;
            f32x4 m[9], y[2][2];
#pragma unroll
            for (int f = 0; f < 9; ++f) m[f] = acc[f][nt];
            wino22_output(m, y);
            f32x4 c2 = {0.f, 0.f, 0.f, 0.f}, c3 = c2;
            if (EPI == EPI_DEC) {
                c2 = *reinterpret_cast<const f32x4*>(p.bn_s + co); c3 = *reinterpret_cast<const f32x4*>(p.bn_beta + co);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 vt = y[a][b];
                    if (EPI == EPI_DEC) vt = lrelu4(__builtin_elementwise_fma(vt, c2, c3));
                    const int oy = y0 + 2 * (2 * wty + a) + py, ox = x0 + 2 * (2 * wtx + b) + px;
                    act_store4<false>(p.out, ((size_t)(n * p.H + oy) * p.W + ox) * p.Cout + co, vt);
                }
        }
    }
    float e0[NT], e1[NT], e2[NT], e3[NT], scb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = g * COUT_T + nt * 16 + i16;
        e0[nt] = e1[nt] = e2[nt] = e3[nt] = scb[nt] = 0.f;
        if (EPI == EPI_DEC) { e2[nt] = p.bn_s[co]; e3[nt] = p.bn_beta[co]; }      // BatchNorm folded: y = lrelu(fmaf(v, s, k))
        if (SC) scb[nt] = p.sc_bias[co];
    }
    // stores in the quad-transposed layout: lane -> (x = lane&3, channels 4*((lane>>2)&3)..+3), 16 bytes each
    const int xj = lane & 3, cq4 = ((lane >> 2) & 3) * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cot = g * COUT_T + nt * 16 + cq4;
        if constexpr (!WINO) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = y0 + 2 * ((mt >> 1) * 4 + (lane >> 4)) + py;
            const int ox = x0 + 2 * ((mt & 1) * 4 + xj) + px;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[mt][nt][r];
                if (EPI == EPI_DEC) {
                    v[r] = lrelu(fmaf(v[r], e2[nt], e3[nt]));
                }
            }
            const f32x4 vt = quad_transpose(v[0], v[1], v[2], v[3], xj);
#ifdef GSA_DBG_HOOKS
            if (!(p.dbg & 4))
#endif
            act_store4<BF>(p.out, ((size_t)(n * p.H + oy) * p.W + ox) * p.Cout + cot, vt);
        }
        }
        if (SC) {
            const int iy = y0 / 2 + (wave >> 1) * 4 + (lane >> 4);
            const int ix = x0 / 2 + (wave & 1) * 4 + xj;
            act_store4<BF>(p.out_sc, ((size_t)(n * p.Hs + iy) * p.Ws + ix) * p.Cout + cot,
                quad_transpose(accs[nt][0] + scb[nt], accs[nt][1] + scb[nt], accs[nt][2] + scb[nt], accs[nt][3] + scb[nt], xj));
        }
    }
}

// ------------------------------------------------------------------------------------------
// subpixel_mfma for layers with ONE output-channel group whose whole weight panel fits in LDS (the 512^2 ->
// 1024^2 layers: 16 output channels, 32-64 input channels).  There the 16 KB weight block was 2.5x the 6.4 KB
// activation block of every (tile, channel-block) item, re-read from L2 by each of 32768 tiles.  Here the
// workgroups are persistent: the panel is copied to LDS once, a workgroup walks a contiguous range of tiles,
// the activation image is double buffered (one barrier per item) and the loads of item i+2 fly during the
// MFMAs of item i -- the structure of conv3x3_mfma's double-buffered form.  Arithmetic order per output is
// unchanged (bit-identical results).
// KB = 16-channel blocks per item (2 when the block count is even: half as many workgroup barriers per MFMA).
// WST = the panel does not fit: weights are STREAMED through a two-block LDS ring shared by both halves (all 512
// threads stage block it+1 while block it is multiplied; both halves are at the same channel block in every
// iteration by construction), and blockIdx.y selects the output-channel group.  A weight block is then read once
// per two tiles instead of once per tile.
template <int NT, int EPI, bool SC, bool BF, int KB, bool WST, bool WINO = false>
__global__ __launch_bounds__(512, 2) void subpixel_res(ConvParams p) {
    static_assert(!WST || KB == 1, "streamed weights: one channel block per item");
    static_assert(!(WINO && BF), "the Winograd form is fp32 only");
    constexpr int PX = BF ? 8 : 16, TS = BF ? 128 : 256, KQ = BF ? 2 : 4;
    constexpr int LH = 10, LW = 10, RS = LW * PX + (BF || WINO ? 4 : 8);     // WINO: conflict-free 3x3 patch reads at stride-2 tiles
    constexpr int NA = WINO ? 9 : 4;                  // accumulator tiles per 16 output channels: frequencies / 4x4 patches
    constexpr int SEG = 16 * TS, NB4 = NT * SEG / 4, BIT = (NB4 + 511) / 512;
    constexpr int SIT = (NT * TS / 4 + 511) / 512;
    const int nblk0 = p.C0 >> 4, nblk = (p.C0 + p.C1) >> 4;
    const int nitem = nblk / KB;                   // items per tile
    constexpr int AB = KB * LH * RS;               // one activation buffer: KB block images
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // The workgroup is two halves of 4 waves (one CU holds one workgroup, every SIMD one wave of each half): each
    // half walks its own tiles through its own double-buffered activation image, both read the ONE weight panel.
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int wblk = WST ? 2 : nblk;               // weight blocks held in LDS
    const int g = WST ? (int)blockIdx.y : 0;       // output-channel group of this workgroup
    float* sW = smem;                              // [wblk][q][tap16][ci][16][cg]
    float* sS = sW + wblk * NT * SEG;              // SC: [wblk][q][ci][16][cg]
    float* sA = sS + (SC ? wblk * NT * TS : 0) + half * (2 * AB);            // per half [2][KB][LH*RS]
    f32x4* sAff = reinterpret_cast<f32x4*>(sS + (SC ? wblk * NT * TS : 0) + 4 * AB) + half * (2 * KB * 16);   // per half [2][KB][16]
    const int tid = threadIdx.x & 255, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = wave >> 1, px = wave & 1;
    const int i16 = lane & 15, kq = lane >> 4;
    const bool has_aff = p.aff0 != nullptr;

    // contiguous range of tiles, order (n, ty, tx)
    const int chunk = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    int w_begin = xcd_block(blockIdx.x, gridDim.x) * chunk;
    int w_end = min(p.total_tiles, w_begin + chunk);
    if (w_begin >= w_end) return;                  // whole workgroup
    const int first_half = (w_end - w_begin + 1) >> 1;
    const int iters = first_half * nitem;          // loop trips of the longer half: both halves run the same barriers
    if (half == 0) w_end = w_begin + first_half; else w_begin += first_half;
    struct Tile { int n, y0, x0; };
    auto advance = [&](const Tile& t) {
        Tile u = t;
        u.x0 += 16;
        if (u.x0 == p.W) { u.x0 = 0; u.y0 += 16; if (u.y0 == p.H) { u.y0 = 0; u.n += 1; } }
        return u;
    };
    auto is_edge = [&](const Tile& t) { return t.y0 == 0 || t.x0 == 0 || t.y0 + 16 == p.H || t.x0 + 16 == p.W; };
    const int t_ly = tid / LW - 1, t_lx = tid % LW - 1;                 // input-tile coordinates of the staged pixel
    const int t_lds = tid < LH * LW ? (tid / LW) * RS + (tid % LW) * PX : -1;
    auto tile_pixel = [&](const Tile& t) {
        TilePixel tp;
        const int gy = (t.y0 >> 1) + t_ly, gx = (t.x0 >> 1) + t_lx;
        const bool inside = t_lds >= 0 && (unsigned)gy < (unsigned)p.Hs && (unsigned)gx < (unsigned)p.Ws;
        tp.lds = t_lds;
        tp.pix = inside ? (t.n * p.Hs + gy) * p.Ws + gx : -1;
        return tp;
    };
    int abase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        abase[mt] = ((mt >> 1) * 4 + (i16 >> 2) + 1) * RS + ((mt & 1) * 4 + (i16 & 3) + 1) * PX + kq * KQ;
    const int bbase = (kq * 16 + i16) * KQ;
    const int asc = ((wave >> 1) * 4 + (i16 >> 2) + 1) * RS + ((wave & 1) * 4 + (i16 & 3) + 1) * PX + kq * KQ;
    const int wty = wino22_tile_y(i16), wtx = wino22_tile_x(i16);       // WINO: the lane's Winograd tile; its 3x3 patch shifted by the class
    const int pbase = (2 * wty + py) * RS + (2 * wtx + px) * 16 + kq * 4;
    f32x4 acc[NA][NT];
    f32x4 accs[SC ? NT : 1];
#pragma unroll
    for (int mt = 0; mt < NA; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (SC) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accs[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- weights: the whole panel (and the shortcut's) -> LDS once, or (WST) block 0 now and block 1 into registers
    f32x4 rbw[BIT], rsw[SC ? SIT : 1];
    auto load_w = [&](int cbk) {
#pragma unroll
        for (int j = 0; j < BIT; ++j) {
            const int i = min((int)threadIdx.x + j * 512, NB4 - 1);
            const int q = i / (SEG / 4), r = i % (SEG / 4);
            rbw[j] = reinterpret_cast<const f32x4*>(p.wpk + ((size_t)(g * NT + q) * nblk + cbk) * SEG)[r];
        }
        if (SC) {
#pragma unroll
            for (int j = 0; j < SIT; ++j) {
                const int i = min((int)threadIdx.x + j * 512, NT * TS / 4 - 1);
                const int q = i / (TS / 4), r = i % (TS / 4);
                rsw[j] = reinterpret_cast<const f32x4*>(p.wsc + ((size_t)(g * NT + q) * nblk + cbk) * TS)[r];
            }
        }
    };
    auto store_w = [&](int slot) {
#pragma unroll
        for (int j = 0; j < BIT; ++j) reinterpret_cast<f32x4*>(sW + slot * NT * SEG)[min((int)threadIdx.x + j * 512, NB4 - 1)] = rbw[j];
        if (SC) {
#pragma unroll
            for (int j = 0; j < SIT; ++j) reinterpret_cast<f32x4*>(sS + slot * NT * TS)[min((int)threadIdx.x + j * 512, NT * TS / 4 - 1)] = rsw[j];
        }
    };
    if (WST) {
        load_w(0);
        store_w(0);
        load_w(1 % nblk);
    } else {
        for (int cbk = 0; cbk < nblk; ++cbk) { load_w(cbk); store_w(cbk); }
    }
    // per-channel epilogue constants: one channel group, so they never change
    // The MFMAs are issued with the operands SWAPPED (weights as A, pixels as B): D^T, i.e. a lane holds FOUR CONSECUTIVE OUTPUT
    // CHANNELS (4*(lane>>4) + r) of ONE pixel (lane & 15 -> patch row (lane&15)>>2, column lane&3) -- exactly a 16-byte NHWC
    // store, no quad transpose (8 DPP moves + 8 selects per accumulator tile).  a*b == b*a and the k order is unchanged: same bits.
    f32x4 e0[NT], e1[NT], e2[NT], e3[NT], scb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = g * 16 * NT + nt * 16 + 4 * (lane >> 4);
        e0[nt] = e1[nt] = e2[nt] = e3[nt] = scb[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_DEC) {
            e2[nt] = *reinterpret_cast<const f32x4*>(p.bn_s + co); e3[nt] = *reinterpret_cast<const f32x4*>(p.bn_beta + co);
        }
        if (SC) scb[nt] = *reinterpret_cast<const f32x4*>(p.sc_bias + co);
    }
    const int pyl = (lane & 15) >> 2, pxl = lane & 3, cq4 = 4 * (lane >> 4);     // pixel of the 4x4 patch, first channel of the lane
    const unsigned lane_out = WINO ? (unsigned)((4 * wty * p.W + 4 * wtx) * p.Cout + cq4) : (unsigned)((2 * pyl * p.W + 2 * pxl) * p.Cout + cq4);
    const unsigned lane_sc = (unsigned)((pyl * p.Ws + pxl) * p.Cout + cq4);

    f32x4 ra[KB][4], raff[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) raff[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_item = [&](const Tile& t, int ci, const TilePixel& tp) {      // item ci of tile t = blocks ci*KB .. +KB-1
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int cb = ci * KB + kb;
            const bool first = cb < nblk0;
            load_pixel<BF>(ra[kb], first ? p.src0 : p.src1, first ? p.C0 : p.C1, (first ? cb : cb - nblk0) * 16, tp);
            if (has_aff && first) raff[kb] = reinterpret_cast<const f32x4*>(p.aff0 + (size_t)t.n * p.C0 + cb * 16)[tid & 15];
        }
    };
    auto write_aff_item = [&](int slot) {
        if (has_aff && tid < 16) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) sAff[(slot * KB + kb) * 16 + tid] = raff[kb];
        }
    };
    auto write_item = [&](int ci, const TilePixel& tp, bool edge, int buf) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int cb = ci * KB + kb;
            float* a_img = sA + buf * AB + kb * (LH * RS);
            const float4* tab = reinterpret_cast<const float4*>(sAff) + (buf * KB + kb) * 16;
            if (cb < nblk0 && has_aff) {
                if (edge) store_pixel<true, BF, true>(a_img, ra[kb], tab, tp);
                else store_pixel<true, BF, false>(a_img, ra[kb], tab, tp);
            } else if (edge) store_pixel<false, BF, true>(a_img, ra[kb], tab, tp);
            else store_pixel<false, BF, false>(a_img, ra[kb], tab, tp);
        }
    };
    auto mfma_block = [&](int buf, int kb, int cb) {      // cb: weight slot in LDS
        const float* a_img = sA + buf * AB + kb * (LH * RS);
        const float* b_img = sW + cb * (NT * SEG);
        // valid taps of this parity class, ascending ky then kx:
        //   py==0: ky=1 (dy 0), ky=3 (dy -1);   py==1: ky=0 (dy +1), ky=2 (dy 0)
        if constexpr (WINO) {
            wino22_block<NT, SEG>(a_img + pbase, RS, b_img + bbase, py, px, acc);
        } else {
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) {
            const int ky = (py ? 0 : 1) + 2 * jy;
            const int dy = py ? (1 - jy) : -jy;
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                const int kx = (px ? 0 : 1) + 2 * jx;
                const int dx = px ? (1 - jx) : -jx;
                const int toff = dy * RS + dx * PX;
                const int tap = ky * 4 + kx;
                if constexpr (BF) {
                    s16x4 a[4], b[NT];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const s16x4*>(a_img + abase[mt] + toff);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const s16x4*>(b_img + bbase + nt * SEG + tap * TS);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b[nt], a[mt], acc[mt][nt], 0, 0, 0);
                } else {
                    f32x4 a[4], b[NT];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(a_img + abase[mt] + toff);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(b_img + bbase + nt * SEG + tap * 256);
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt][cg], a[mt][cg], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        }
        if constexpr (SC && BF) {
            const s16x4 as = *reinterpret_cast<const s16x4*>(a_img + asc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                accs[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const s16x4*>(sS + cb * (NT * TS) + bbase + nt * TS), as, accs[nt], 0, 0, 0);
        } else if constexpr (SC) {
            const f32x4 as = *reinterpret_cast<const f32x4*>(a_img + asc);
            f32x4 bs[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bs[nt] = *reinterpret_cast<const f32x4*>(sS + cb * (NT * TS) + bbase + nt * 256);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    accs[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bs[nt][cg], as[cg], accs[nt], 0, 0, 0);
        }
    };
    auto mfma_item = [&](int buf, int ci, int it) {
        if (p.prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) mfma_block(buf, kb, WST ? (it & 1) : ci * KB + kb);
        if (p.prio) __builtin_amdgcn_s_setprio(3);
    };
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    auto epilogue = [&](const Tile& t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (WINO) {
                f32x4 m[9], y[2][2];
#pragma unroll
                for (int f = 0; f < 9; ++f) { m[f] = acc[f][nt]; acc[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                wino22_output(m, y);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {       // the tile's 2x2 class outputs (a, b) = (mt >> 1, mt & 1)
                    f32x4 vt = y[mt >> 1][mt & 1];
                    if (EPI == EPI_DEC) {
                        vt = lrelu4(__builtin_elementwise_fma(vt, e2[nt], e3[nt]));
                    }
                    const size_t ubase = ((size_t)(t.n * p.H + t.y0 + 2 * (mt >> 1) + py) * p.W + t.x0 + 2 * (mt & 1) + px) * p.Cout + g * 16 * NT + nt * 16;
                    act_store4<false>(p.out, ubase + lane_out, vt);
                }
            } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 vt = acc[mt][nt];
                if (EPI == EPI_DEC) {
                    vt = lrelu4(__builtin_elementwise_fma(vt, e2[nt], e3[nt]));
                }
                const size_t ubase = ((size_t)(t.n * p.H + t.y0 + 8 * (mt >> 1) + py) * p.W + t.x0 + 8 * (mt & 1) + px) * p.Cout + g * 16 * NT + nt * 16;
                act_store4<BF>(p.out, ubase + lane_out, vt);
                acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            }
            if (SC) {
                const size_t ubase = ((size_t)(t.n * p.Hs + (t.y0 >> 1) + (wave >> 1) * 4) * p.Ws + (t.x0 >> 1) + (wave & 1) * 4) * p.Cout + g * 16 * NT + nt * 16;
                act_store4<BF>(p.out_sc, ubase + lane_sc, accs[nt] + scb[nt]);
                accs[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // ---- items (tile, channel block): `tc/cb` is multiplied out of LDS buffer it&1, `tr/cbr` sits in registers
    const int total_items = max(w_end - w_begin, 0) * nitem;     // of this half (the second one may have none)
    Tile tc, tr;
    {
        const int tiles_y = p.H >> 4, w0 = min(w_begin, p.total_tiles - 1);
        const int tx = w0 % p.tiles_x, r = w0 / p.tiles_x;
        tc.x0 = tx * 16; tc.y0 = (r % tiles_y) * 16; tc.n = r / tiles_y;
    }
    int cb = 0, cbr = 0;
    TilePixel tpr = tile_pixel(tc);
    auto next_item = [&](int i, Tile& t, int& cbi, TilePixel& tp) {
        if (i + 1 >= total_items) return;
        if (++cbi == nitem) { cbi = 0; t = advance(t); tp = tile_pixel(t); }
    };
    load_item(tc, cb, tpr);
    write_aff_item(0);
    __syncthreads();                       // weight panel + entries of item 0 visible
    if (total_items > 0) write_item(cb, tpr, is_edge(tc), 0);
    tr = tc; cbr = cb;
    next_item(0, tr, cbr, tpr);
    load_item(tr, cbr, tpr);
    write_aff_item(1);
    __syncthreads();
    unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, sw = 0, sl = 0, sm = 0, se = 0, sb = 0;
    (void)k0; (void)k1; (void)k2; (void)k3; (void)k4; (void)k5; (void)sw; (void)sl; (void)sm; (void)se; (void)sb;
    for (int it = 0; it < iters; ++it) {
        TICK(k0);
        if (WST && it + 1 < iters) {        // every thread of the workgroup: weight block of iteration it+1 -> the other slot,
            store_w((it + 1) & 1);          // then the block of iteration it+2 -> registers
            load_w((it + 2) % nblk);
        }
        if (it < total_items) {
            if (it + 1 < total_items) write_item(cbr, tpr, is_edge(tr), (it + 1) & 1);
            TICK(k1);
            Tile t2 = tr; int cb2 = cbr;
            next_item(it + 1, t2, cb2, tpr);
            load_item(t2, cb2, tpr);
            TICK(k2);
            mfma_item(it & 1, cb, it);
            TICK(k3);
            if (cb == nitem - 1) epilogue(tc);
            write_aff_item(it & 1);
            TICK(k4);
            tc = tr; cb = cbr; tr = t2; cbr = cb2;
        }
        __syncthreads();
        TICK(k5);
        if (it < total_items) { TSUM(sw, k0, k1); TSUM(sl, k1, k2); TSUM(sm, k2, k3); TSUM(se, k3, k4); TSUM(sb, k4, k5); }
    }
    TFLUSH(6, sw); TFLUSH(7, sl); TFLUSH(8, sm); TFLUSH(9, se); TFLUSH(10, sb);
    TFLUSH(wave, sb);             // diagnostic: barrier wait by wave index (0-3) ...
    if (wave == 0) TFLUSH(4, sw); // ... and the write phase of waves 0 and 3
    if (wave == 3) TFLUSH(5, sw);
    TFLUSH(12, (unsigned long long)total_items); TFLUSH(15, 1ull);
}

// ------------------------------------------------------------------------------------------
// Blur (depthwise 3x3, zero pad) -> AddNoise -> Bias -> LeakyReLU -> statistics.
// One thread = one aligned quad of 4 consecutive x for 4 consecutive channels.
// One element of the post pass -- blur (or the constant tensor), noise, bias, LeakyReLU for 4 consecutive x and 4 channels at (n, y, x0, c) --
// stored to p.out and returned in v[x][channel] for the statistics (post_kernel and post_fin_kernel share it: same operations, same bits)
template <bool BF>
__device__ __forceinline__ void post_element(const PostParams& p, int n, int c, int x0, int y, float (&v)[4][4]) {
            // per-sample source = an activation tensor (bf16 in bf16 mode); the broadcast constant tensor is an fp32 parameter
            const size_t sbase = p.src_per_sample ? (size_t)n * p.H * p.W * p.C : 0;
                    if (p.blur) {
                float wk[4][9];
    #pragma unroll
                for (int j = 0; j < 4; ++j)
    #pragma unroll
                    for (int t = 0; t < 9; ++t) wk[j][t] = p.blur[(c + j) * 9 + t];
    #pragma unroll
                for (int r = 0; r < 4; ++r)
    #pragma unroll
                    for (int j = 0; j < 4; ++j) v[r][j] = 0.0f;
                // all 18 loads are unconditional (clamped coordinates) and issued before the first use; taps outside the
                // image are zeroed values: fmaf(0, w, b) == b bit for bit (b is never -0), i.e. the same as skipping them
                float4 row[3][6];
    #pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    const bool vy = yy >= 0 && yy < p.H;
                    const int yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
    #pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int xx = x0 - 1 + k;
                        const int xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
                        const f32x4 ld = act_load4<BF>(p.src, sbase + ((size_t)yc * p.W + xc) * p.C + c);
                        row[ky][k] = make_float4(ld[0], ld[1], ld[2], ld[3]);
                        if (!(vy && xx >= 0 && xx < p.W)) row[ky][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
    #pragma unroll
                for (int ky = 0; ky < 3; ++ky)
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
    #pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float4 t = row[ky][r + kx];
                            v[r][0] = fmaf(t.x, wk[0][ky * 3 + kx], v[r][0]);
                            v[r][1] = fmaf(t.y, wk[1][ky * 3 + kx], v[r][1]);
                            v[r][2] = fmaf(t.z, wk[2][ky * 3 + kx], v[r][2]);
                            v[r][3] = fmaf(t.w, wk[3][ky * 3 + kx], v[r][3]);
                        }
            } else {
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 t = *reinterpret_cast<const float4*>(p.src + sbase + ((size_t)y * p.W + x0 + r) * p.C + c);      // no blur: the fp32 constant tensor
                    v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
                }
            }
            const float4 nz = *reinterpret_cast<const float4*>(p.noise + ((size_t)n * p.H + y) * p.W + x0);
            const float nzv[4] = {nz.x, nz.y, nz.z, nz.w};
            const float4 sf = *reinterpret_cast<const float4*>(p.nscale + c);
            const float4 nb = *reinterpret_cast<const float4*>(p.nbias + c);
            const float sfv[4] = {sf.x, sf.y, sf.z, sf.w}, nbv[4] = {nb.x, nb.y, nb.z, nb.w};
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = sfv[j] * nzv[r];
                    v[r][j] = lrelu((v[r][j] + t) + nbv[j]);
                }
                act_store4<BF>(p.out, (((size_t)n * p.H + y) * p.W + x0 + r) * p.C + c, f32x4{v[r][0], v[r][1], v[r][2], v[r][3]});
            }
}

template <bool BF>
__global__ __launch_bounds__(256) void post_kernel(PostParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sstat[];   // [2][C]
    const int n = blockIdx.y;
    const int C4 = p.C >> 2, W4 = p.W >> 2;
    const int total = p.H * W4 * C4;
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) sstat[i] = 0ull;
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < total) {
        // channel quads fastest, then 8 rows of one x-quad column, then the next column: a workgroup covers a
        // patch of ~8 rows, so the 3-row blur window is re-read from its own L1 instead of by workgroups that
        // sit on other XCDs (HBM fetch was 2.1x the tensor with row-major blocks)
        const int bh = p.H < 8 ? p.H : 8;
        const int cq = idx % C4, t = idx / C4;
        const int xq = (t / bh) % W4, y = (t / (bh * W4)) * bh + t % bh;
        const int c = cq * 4, x0 = xq * 4;
        float v[4][4];   // [x][channel]
        post_element<BF>(p, n, c, x0, y, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = (v[0][j] + v[1][j]) + (v[2][j] + v[3][j]);
            const float q = (v[0][j] * v[0][j] + v[1][j] * v[1][j]) + (v[2][j] * v[2][j] + v[3][j] * v[3][j]);
            atomicAdd(&sstat[c + j], to_fixed(s, kStatScale1));
            atomicAdd(&sstat[p.C + c + j], to_fixed_sq(q, stat_s2(p.H * p.W)));
        }
    }
    __syncthreads();
    if ((int)gridDim.x > p.prow) {       // more blocks than rows: the blocks ADD to row (block mod rows) of the all-zero partials
        for (int i = threadIdx.x; i < p.C; i += 256) {
            StatPart* a = p.partials + ((size_t)n * p.prow + (blockIdx.x & (kDirectRows - 1))) * p.C + i;
            atomicAdd(&a->s1, sstat[i]);
            atomicAdd(&a->s2, sstat[p.C + i]);
        }
        return;
    }
    for (int i = threadIdx.x; i < p.C; i += 256) {
        StatPart sp; sp.s1 = sstat[i]; sp.s2 = sstat[p.C + i];
        p.partials[((size_t)n * p.prow + blockIdx.x) * p.C + i] = sp;
    }
}

// post_kernel for the large planes: one thread = 4 consecutive x, 4 consecutive channels and RPT consecutive rows.  The 3-row
// blur window slides down in registers, so a row of inputs is loaded once per RPT+2 output rows... per thread (6 float4 per
// new row instead of 18 per output row: the first form was bound by its load instructions, 3.5-4.1 TB/s), the loads of row
// r+2 are in flight while row r is computed, and the statistics of the RPT rows stay in registers until the end.
// Same arithmetic in the same order per output (9-tap fmaf chain ky, kx ascending; zero padding = zeroed values).
template <int RPT, bool BF>
__global__ __launch_bounds__(256, GSA_POST_OCC) void post_rows_kernel(PostParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sstat[];   // [2][C]
    const int n = blockIdx.y;
    const int C4 = p.C >> 2, W4 = p.W >> 2;
    // row_groups (launcher; RPT = 4 only: the four-row ring then repeats per group): a thread walks row_groups * RPT consecutive rows
    // with the window carried over -- RPT + 2 rows were loaded per RPT output rows, now RPT * NG + 2 per RPT * NG (6 loads per new row in
    // both: 1.5x -> 1.125x load instructions at NG = 4) and a quarter of the threads pay the cold start of their first three rows
    const int NG = p.row_groups > 1 ? p.row_groups : 1;
    const int total = (p.H / (RPT * NG)) * W4 * C4;
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) sstat[i] = 0ull;
    __syncthreads();
    // XCD-aware block order (workgroup b runs on XCD b % 8): XCD x takes the x-th contiguous eighth of the row groups, so the two
    // halo rows a row group shares with its neighbours come out of that XCD's L2 instead of being fetched once more from HBM by
    // another XCD (rocprofv3, round 2: 1.29x the algorithmic bytes with row groups dealt round-robin)
    const int idx = xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (idx < total) {
        const int cq = idx % C4, t = idx / C4;
        const int xq = t % W4, y0 = (t / W4) * (RPT * NG);
        const int c = cq * 4, x0 = xq * 4;
        const size_t sbase = (size_t)n * p.H * p.W * p.C + c;
        float wk[4][9];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) wk[j][tp] = p.blur[(c + j) * 9 + tp];
        const float4 sf = *reinterpret_cast<const float4*>(p.nscale + c);
        const float4 nb = *reinterpret_cast<const float4*>(p.nbias + c);
        const float sfv[4] = {sf.x, sf.y, sf.z, sf.w}, nbv[4] = {nb.x, nb.y, nb.z, nb.w};
        int xoff[6];
        bool xin[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int xx = x0 - 1 + k;
            xin[k] = xx >= 0 && xx < p.W;
            xoff[k] = (xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx)) * p.C;
        }
        auto load_row = [&](float4 (&row)[6], int yy) {      // unconditional loads (clamped), padding = zeroed values
            const bool vy = yy >= 0 && yy < p.H;
            const int yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
            const size_t rp = sbase + (size_t)yc * p.W * p.C;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const f32x4 ld = act_load4<BF>(p.src, rp + xoff[k]);
                row[k] = make_float4(ld[0], ld[1], ld[2], ld[3]);
                if (!(vy && xin[k])) row[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        float4 win[4][6];                                   // rows y-1, y, y+1 and the prefetched y+2 (ring of four)
        load_row(win[0], y0 - 1);
        load_row(win[1], y0);
        load_row(win[2], y0 + 1);
        unsigned long long I1[4] = {0, 0, 0, 0}, I2[4] = {0, 0, 0, 0};
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int y = y0 + gi * RPT + r;
            if (r + 1 < RPT || gi + 1 < NG) load_row(win[(r + 3) & 3], y + 2);      // in flight during this row's arithmetic
            const float4 nz = *reinterpret_cast<const float4*>(p.noise + ((size_t)n * p.H + y) * p.W + x0);
            const float nzv[4] = {nz.x, nz.y, nz.z, nz.w};
            float v[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[q][j] = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 tv = win[(r + ky) & 3][q + kx];
                        v[q][0] = fmaf(tv.x, wk[0][ky * 3 + kx], v[q][0]);
                        v[q][1] = fmaf(tv.y, wk[1][ky * 3 + kx], v[q][1]);
                        v[q][2] = fmaf(tv.z, wk[2][ky * 3 + kx], v[q][2]);
                        v[q][3] = fmaf(tv.w, wk[3][ky * 3 + kx], v[q][3]);
                    }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float tn = sfv[j] * nzv[q];
                    v[q][j] = lrelu((v[q][j] + tn) + nbv[j]);
                }
                act_store4<BF>(p.out, (((size_t)n * p.H + y) * p.W + x0 + q) * p.C + c, f32x4{v[q][0], v[q][1], v[q][2], v[q][3]});
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sq = (v[0][j] + v[1][j]) + (v[2][j] + v[3][j]);
                const float qq = (v[0][j] * v[0][j] + v[1][j] * v[1][j]) + (v[2][j] * v[2][j] + v[3][j] * v[3][j]);
                I1[j] += to_fixed(sq, kStatScale1);
                I2[j] += to_fixed_sq(qq, stat_s2(p.H * p.W));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&sstat[c + j], I1[j]);
            atomicAdd(&sstat[p.C + c + j], I2[j]);
        }
    }
    __syncthreads();
    if ((int)gridDim.x > p.prow) {       // more blocks than rows: the blocks ADD to row (block mod rows) of the all-zero partials
        for (int i = threadIdx.x; i < p.C; i += 256) {
            StatPart* a = p.partials + ((size_t)n * p.prow + (blockIdx.x & (kDirectRows - 1))) * p.C + i;
            atomicAdd(&a->s1, sstat[i]);
            atomicAdd(&a->s2, sstat[p.C + i]);
        }
        return;
    }
    for (int i = threadIdx.x; i < p.C; i += 256) {
        StatPart sp; sp.s1 = sstat[i]; sp.s2 = sstat[p.C + i];
        p.partials[((size_t)n * p.prow + blockIdx.x) * p.C + i] = sp;
    }
}

// ------------------------------------------------------------------------------------------
// InstanceNorm statistics -> AdaIN coefficients in ONE launch.  Many blocks per (sample,
// 64-channel group) sum the fixed-point partial rows into acc[n][C] with 64-bit integer atomics
// (order independent, hence deterministic); the block that draws the last ticket of its group
// folds InstanceNorm(eps 1e-5) with the style and clears acc and the ticket for the next layer.
// Hand-off between blocks: device-scope atomics on both sides (the adds, the ticket, and the
// final reads via atomic exchange), never plain loads of another block's data.
// InstanceNorm + AdaIN coefficients of channel c of sample n from its two fixed-point sums (the tail of finalize_kernel; also called by
// the producers that hold a whole plane in one workgroup: post_fin_kernel)
__device__ __forceinline__ void finalize_one(const FinalizeParams& p, int n, int c, unsigned long long I1, unsigned long long I2) {
    const double inv_hw = 1.0 / (double)p.HW;   // HW is a power of two
    const double m = (double)(long long)I1 * (1.0 / kStatScale1) * inv_hw;
    const double e2 = (double)(long long)I2 * (1.0 / stat_scale2(p.HW)) * inv_hw;
    double var = fma(-m, m, e2);
    // range check of the fixed-point sums (include/gsa.h): a sum within a factor 4 of the 64-bit wrap, or a variance that is
    // negative beyond rounding (what a wrapped sum of squares produces), sets the sticky word gsa_check reports
    {
        const long long s1 = (long long)I1;
        const bool near_wrap = (s1 < 0 ? -s1 : s1) >= (1ll << 61) || I2 >= (1ull << 61);
        if (p.flags && (near_wrap || var < -1e-6 * (e2 + m * m) - 1e-30)) atomicOr(p.flags, 1u);
    }
    if (!(var > 0.0)) var = 0.0;
    const float mean_f = (float)m, var_f = (float)var;
    const float inv = 1.0f / sqrtf(var_f + 1e-5f);
    const float gsc = p.gamma[c] * inv;
    const float* st = p.style + (size_t)n * p.style_stride;
    const float s1 = st[c] + 1.0f;
    Aff a;
    a.mean = mean_f;
    a.A = gsc * s1;
    a.B = fmaf(-mean_f, a.A, fmaf(p.beta[c], s1, st[p.C + c]));      // the mean folded into the shift: consumers apply ONE fmaf(x, A, B)
    a.pad = 0.0f;
    p.aff[(size_t)n * p.C + c] = a;
}

__global__ __launch_bounds__(256) void finalize_kernel(FinalizeParams p, int rows_per_block) {
    __shared__ unsigned long long sh[2][256];
    __shared__ int is_last;
    const int n = blockIdx.y;
    const int cblk = p.C >= 64 ? 64 : p.C, nrg = 256 / cblk;
    const int cl = threadIdx.x % cblk, rg = threadIdx.x / cblk;
    const int c = blockIdx.x * 64 + cl;
    unsigned long long I1 = 0, I2 = 0;
    if (rg < nrg && c < p.C) {
        const int r0 = blockIdx.z * rows_per_block;
        const int r1 = min(p.prow, r0 + rows_per_block);
        const StatPart* base = p.partials + (size_t)n * p.prow * p.C + c;
#pragma unroll 16
        for (int r = r0 + rg; r < r1; r += nrg) {      // independent loads, all in flight together (one memory round trip)
            const StatPart sp = base[(size_t)r * p.C];
            I1 += sp.s1; I2 += sp.s2;
        }
        // the rows go back to zero: the producers that ADD to their rows (64-bit atomics on at most kDirectRows rows) rely on
        // `partials` being all zero between layers -- no memset launch per layer
        StatPart zero; zero.s1 = 0ull; zero.s2 = 0ull;
        StatPart* wbase = const_cast<StatPart*>(base);
        for (int r = r0 + rg; r < r1; r += nrg) wbase[(size_t)r * p.C] = zero;
    }
    sh[0][threadIdx.x] = I1; sh[1][threadIdx.x] = I2;
    __syncthreads();
    const bool single = gridDim.z == 1;      // one block per channel group: no accumulator round trip, no ticket
    if (rg == 0 && c < p.C) {
        for (int k = 1; k < nrg; ++k) { I1 += sh[0][k * cblk + cl]; I2 += sh[1][k * cblk + cl]; }
        if (!single) {
            atomicAdd(&p.acc[(size_t)n * p.C + c].s1, I1);
            atomicAdd(&p.acc[(size_t)n * p.C + c].s2, I2);
        }
    }
    if (!single) {     // several blocks per channel group: the last arriver finalizes
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* ticket = p.tickets + n * gridDim.x + blockIdx.x;
            const unsigned t = atomicAdd(ticket, 1u);
            is_last = t == gridDim.z - 1;
            if (is_last) atomicExch(ticket, 0u);
        }
        __syncthreads();
        if (!is_last) return;
        __threadfence();
    }
    if (rg == 0 && c < p.C) {
        if (!single) {
            StatPart* ap = p.acc + (size_t)n * p.C + c;
            I1 = atomicExch(&ap->s1, 0ull);      // read the total and clear it for the next layer
            I2 = atomicExch(&ap->s2, 0ull);
        }
        finalize_one(p, n, c, I1, I2);
    }
}

// Post pass AND finalize in one launch for small planes (round 4): one workgroup = one (sample, 16-channel group) plane, so its LDS
// sums are the plane's complete statistics and the coefficients follow at once -- no partial rows, no finalize launch.  Where the
// small launches are the critical path (bf16 mode, small batches: a finalize launch costs ~7 us between two kernel boundaries) this
// removes one launch per level up to 32 px; the large-batch fp32 step hides them anyway.  Same bits:
// integer sums, the same finalize_one.
template <bool BF>
__global__ __launch_bounds__(256) void post_fin_kernel(PostParams p, FinalizeParams f) {
    __shared__ unsigned long long sstat[2][16];
    const int n = blockIdx.y, c0 = blockIdx.x * 16;
    const int W4 = p.W >> 2, total = p.H * W4 * 4;
    if (threadIdx.x < 32) sstat[threadIdx.x >> 4][threadIdx.x & 15] = 0ull;
    __syncthreads();
    const int s2 = stat_s2(p.H * p.W);
    for (int e = threadIdx.x; e < total; e += 256) {
        const int cq = e & 3, t = e >> 2, xq = t % W4, y = t / W4;
        float v[4][4];
        post_element<BF>(p, n, c0 + cq * 4, xq * 4, y, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = (v[0][j] + v[1][j]) + (v[2][j] + v[3][j]);
            const float q = (v[0][j] * v[0][j] + v[1][j] * v[1][j]) + (v[2][j] * v[2][j] + v[3][j] * v[3][j]);
            atomicAdd(&sstat[0][cq * 4 + j], to_fixed(s, kStatScale1));
            atomicAdd(&sstat[1][cq * 4 + j], to_fixed_sq(q, s2));
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) finalize_one(f, n, c0 + threadIdx.x, sstat[0][threadIdx.x], sstat[1][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// Mapping network pieces.  Every output is one k-ordered fmaf chain (canonical order).
__global__ void pixelnorm_kernel(const float* z, float* out, int n, int L) {
    // one wave per sample: the row goes to LDS, lane 0 runs the canonical chain, all lanes scale
    extern __shared__ __attribute__((aligned(16))) float srow[];
    __shared__ float rn;
    const int s = blockIdx.x;
    const float* zs = z + (size_t)s * L;
    for (int k = threadIdx.x; k < L; k += blockDim.x) srow[k] = zs[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        float ss = 0.0f;
        for (int k = 0; k < L; ++k) ss = fmaf(srow[k], srow[k], ss);
        rn = 1.0f / sqrtf(ss / (float)L + 1e-8f);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < L; k += blockDim.x) out[(size_t)s * L + k] = srow[k] * rn;
}

// Dense layer(s) with the weight panel staged through LDS: one workgroup = 64 output columns
// x up to 16 samples.  STYLE: column j belongs to style layer col_layer[j] and its input is the
// truncated latent x_k = latent_avg[k]*(1-psi_l) + w[k]*psi_l (reference :158-163), formed on
// the fly.  Every output is one k-ordered fmaf chain (canonical order); the K loop only walks
// LDS, so the 512-step chain costs ~2 us instead of 512 dependent global-load round trips.
// JB = output columns per workgroup: 64 for the wide style affines, 16 for the 512-wide mapping layers (32 workgroups
// instead of 8: the eight layers are a serial latency chain at the head of every step)
template <bool STYLE, int JB, int KC>      // KC = K rows per LDS pass
__global__ __launch_bounds__(256) void dense_lds_kernel(const float* x, const float* WT, const float* b, float* y,
                                                        int n, int K, int J, int act, const float* avg,
                                                        const float* psi, const int* col_layer) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int G = 256 / JB, SPT = 16 / G;       // sample groups, samples per thread (16 samples per pass)
    float* sW = smem;                               // [KC][JB]
    float* sX = sW + KC * JB;                       // [16][KC]
    float* sAvg = sX + 16 * KC;                     // [KC] (STYLE)
    const int tid = threadIdx.x, jl = tid % JB, ng = tid / JB;
    const int j0 = blockIdx.x * JB, j = j0 + jl;
    const int jc = j < J ? j : J - 1;
    float ps = 1.0f, om = 0.0f;
    if (STYLE) { ps = psi[col_layer[jc]]; om = 1.0f - ps; }
    for (int n0 = 16 * (int)blockIdx.y; n0 < n; n0 += 16 * (int)gridDim.y) {      // blockIdx.y: 16-sample chunks side by side
        const int nn = min(16, n - n0);
        float acc[SPT];                             // samples n0 + ng + G*i
#pragma unroll
        for (int i = 0; i < SPT; ++i) acc[i] = 0.f;
        for (int k0 = 0; k0 < K; k0 += KC) {
            __syncthreads();
            // weight panel: KC x JB floats, coalesced rows of 4*JB bytes
            constexpr int Q4 = JB / 4, NW4 = KC * Q4 / 256;
            f32x4 rw[NW4];
#pragma unroll
            for (int i = 0; i < NW4; ++i) {
                const int idx = tid + i * 256, kr = idx / Q4, c4 = (idx % Q4) * 4;
                const int col = min(j0 + c4, J - 4);
                rw[i] = *reinterpret_cast<const f32x4*>(WT + (size_t)(k0 + kr) * J + col);
            }
            for (int idx = tid; idx < nn * KC; idx += 256) sX[idx] = x[(size_t)(n0 + idx / KC) * K + k0 + idx % KC];
            if (STYLE) for (int idx = tid; idx < KC; idx += 256) sAvg[idx] = avg[k0 + idx];
#pragma unroll
            for (int i = 0; i < NW4; ++i) reinterpret_cast<f32x4*>(sW)[tid + i * 256] = rw[i];
            // STYLE: when every column of the workgroup has the same psi (the rule: psi changes at ONE column, a multiple of
            // 64), the truncated latent x' = avg*(1-psi) + w*psi is formed ONCE per (sample, k) here instead of once per
            // (sample, k, column) in the chains below -- the same three roundings either way
            bool pre = false;
            if (STYLE) {
                pre = __syncthreads_and(ps == psi[col_layer[min(j0, J - 1)]]) != 0;
                if (pre) {
                    for (int idx = tid; idx < nn * KC; idx += 256) {
                        const float t0 = sAvg[idx % KC] * om, t1 = sX[idx] * ps;
                        sX[idx] = t0 + t1;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SPT; ++i) {
                const int s = ng + G * i;
                if (s < nn) {
                    float a = acc[i];
                    const float* xs = sX + s * KC;
                    if (STYLE && !pre) {
#pragma unroll 8
                        for (int k = 0; k < KC; ++k) {
                            const float t0 = sAvg[k] * om, t1 = xs[k] * ps;
                            a = fmaf(t0 + t1, sW[k * JB + jl], a);
                        }
                    } else {
#pragma unroll 8
                        for (int k = 0; k < KC; ++k) a = fmaf(xs[k], sW[k * JB + jl], a);
                    }
                    acc[i] = a;
                }
            }
        }
        if (j < J) {
            const float bj = b[j];
#pragma unroll
            for (int i = 0; i < SPT; ++i) {
                const int s = ng + G * i;
                if (s < nn) {
                    const float v = acc[i] + bj;
                    y[(size_t)(n0 + s) * J + j] = act ? lrelu(v) : v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The whole mapping network in ONE launch: PixelNorm + 8 x (dense 512->512 + LeakyReLU) were ten dependent launches of
// ~10 us each at the head of every step (a quarter of a batch-1 step's latency chain).  Here L/16 workgroups each own 16
// output columns of every layer and hand their activations to the others through global memory WITHOUT a barrier: every
// value travels as one naturally aligned 64-bit word {fp32 bits, tag}, written and read with relaxed device-scope atomics,
// where tag = 8 * launch number + layer + 1.  A consumer simply re-reads a word until it carries the tag it expects --
// value and flag arrive together, so no fence, no arrival counter and no cache maintenance is involved (a grid barrier
// was built first: its five dependent device-scope round trips per layer across the eight L2s made the fused kernel no
// faster than the ten launches).  The launch number lives in device memory and is bumped by workgroup 0 when it has
// finished (every other workgroup has read it by then: workgroup 0 cannot finish a layer without their outputs), so the
// kernel is replayable from a captured graph; stale words of earlier launches never match.  Two ping-pong buffers
// suffice: a workgroup can only be two layers ahead of another one after that one has consumed the older buffer.
// Arithmetic per output is unchanged: the canonical k-ordered fmaf chains of pixelnorm_kernel / dense_lds_kernel.
// All L/16 workgroups must be resident together (the launcher keeps the grid within half the CU count); a wait gives up
// after 0.25 s of wall time instead of spinning forever if that is ever violated: the sticky error word ctl[1] is set, the
// results of the step are garbage, and the host reports it at its next synchronisation point (gsa_check, include/gsa.h).
struct MappingParams {
    const float* z;                  // [n][L]
    const float* wt[8];              // [K = L][J = L]
    const float* b[8];
    unsigned long long* ll[2];       // ping-pong {value, tag} words [n][L]; layer i writes ll[(i + 1) & 1]
    float* out;                      // [n][L] the dlatents (plain fp32)
    unsigned* ctl;                   // [1] error word, [2 + slice] launch number of the slice (kMapSlices slices at most)
    int n, L;
};

__global__ __launch_bounds__(256) void mapping_kernel(MappingParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.L, tid = threadIdx.x, jl = tid & 15, ng = tid >> 4;
    const int WS = L + 4;              // row stride of the transposed weight slice: 16 lanes x 16-byte reads hit 64 distinct banks
    float* sW = smem;                  // [16 columns][WS]: column-major, so that a thread reads 4 consecutive k of ITS column at once
    float* sX = sW + 16 * WS;          // [16 samples][L]
    float* rn = sX + 16 * L;           // [16]
    const int j0 = blockIdx.x * 16;
    const int nw4 = L * 4 / 256;       // float4s of the weight slice per thread (L a multiple of 64, <= 512)
    // blockIdx.y = slice: the 16-sample chunks y, y + gridDim.y, ... form an independent copy of the exchange (samples never
    // mix), with its own launch-number word ctl[2 + y]
    unsigned* ctl = p.ctl + 2 + blockIdx.y;
    const unsigned tag0 = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 8u;
    f32x4 rw[8];
    auto load_w = [&](int layer) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nw4) {
                const int idx = tid + i * 256, kr = idx >> 2, c4 = (idx & 3) * 4;
                rw[i] = *reinterpret_cast<const f32x4*>(p.wt[layer] + (size_t)kr * L + j0 + c4);
            }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nw4) {
                const int idx = tid + i * 256, kr = idx >> 2, c4 = (idx & 3) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) sW[(c4 + c) * WS + kr] = rw[i][c];
            }
    };
    // a 16-sample chunk of the previous layer's output -> sX: all words are requested at once, the ones that do not carry
    // the expected tag yet are requested again
    auto fetch_x = [&](const unsigned long long* src, int n0, int nn, unsigned tag) {
        const int cnt = nn * L;                          // <= 8192 words: 32 per thread
        unsigned pending = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (tid + i * 256 < cnt) pending |= 1u << i;
        int spins = 0;
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        while (pending) {
            unsigned long long v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (pending >> i & 1) v[i] = __hip_atomic_load(src + (size_t)n0 * L + tid + i * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if ((pending >> i & 1) && (unsigned)(v[i] >> 32) == tag) {
                    sX[tid + i * 256] = __uint_as_float((unsigned)v[i]);
                    pending &= ~(1u << i);
                }
            if (pending) {
                __builtin_amdgcn_s_sleep(1);
                // never expected (see the header).  Bounded by ELAPSED TIME: s_memrealtime ticks at 100 MHz -- 0.25 s; a partner
                // that already gave up (ctl[1] set) ends the wait at once.  ctl[1] is the sticky error word gsa_check reports.
                if ((++spins & 255) == 0) {
                    const bool late = __builtin_amdgcn_s_memrealtime() - t_start > 25000000ull;
                    if (late || __hip_atomic_load(p.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { atomicExch(p.ctl + 1, 1u); break; }
                }
            }
        }
    };
    load_w(0);
    for (int layer = 0; layer < 8; ++layer) {
        if (layer) __syncthreads();                        // every thread is done with sW / sX of the previous layer
        store_w();
        if (layer + 1 < 8) load_w(layer + 1);
        const float bj = p.b[layer][j0 + jl];
        const int nfirst = 16 * (int)blockIdx.y, nstep = 16 * (int)gridDim.y;
        for (int n0 = nfirst; n0 < p.n; n0 += nstep) {
            const int nn = min(16, p.n - n0);
            if (n0 != nfirst) __syncthreads();
            if (layer == 0) {
                for (int idx = tid; idx < nn * L / 4; idx += 256)
                    reinterpret_cast<f32x4*>(sX)[idx] = *reinterpret_cast<const f32x4*>(p.z + (size_t)n0 * L + idx * 4);
            } else {
                fetch_x(p.ll[layer & 1], n0, nn, tag0 + (unsigned)layer);
            }
            __syncthreads();
            if (layer == 0) {                               // PixelNorm of this chunk (pixelnorm_kernel's chain)
                if (tid < nn) {
                    float ss = 0.0f;
                    const f32x4* r = reinterpret_cast<const f32x4*>(sX + tid * L);
#pragma unroll 4
                    for (int k4 = 0; k4 < L / 4; ++k4) {
                        const f32x4 v = r[k4];
                        ss = fmaf(v[0], v[0], ss); ss = fmaf(v[1], v[1], ss); ss = fmaf(v[2], v[2], ss); ss = fmaf(v[3], v[3], ss);
                    }
                    rn[tid] = 1.0f / sqrtf(ss / (float)L + 1e-8f);
                }
                __syncthreads();
                for (int idx = tid; idx < nn * L; idx += 256) sX[idx] = sX[idx] * rn[idx / L];
                __syncthreads();
            }
            if (ng < nn) {
                float a = 0.f;
                const f32x4* xs = reinterpret_cast<const f32x4*>(sX + ng * L);
                const f32x4* ws = reinterpret_cast<const f32x4*>(sW + jl * WS);
#pragma unroll 4
                for (int k4 = 0; k4 < L / 4; ++k4) {
                    const f32x4 xv = xs[k4], wv = ws[k4];
                    a = fmaf(xv[0], wv[0], a);
                    a = fmaf(xv[1], wv[1], a);
                    a = fmaf(xv[2], wv[2], a);
                    a = fmaf(xv[3], wv[3], a);
                }
                const float yv = lrelu(a + bj);
                const size_t o = (size_t)(n0 + ng) * L + j0 + jl;
                if (layer == 7) p.out[o] = yv;
                else __hip_atomic_store(p.ll[(layer + 1) & 1] + o, ((unsigned long long)(tag0 + (unsigned)layer + 1u) << 32) | __float_as_uint(yv),
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (tid == 0) atomicAdd(ctl, 1u);
    }
}

// ------------------------------------------------------------------------------------------
// toRGB (1x1 conv + bias) and the uint8 image of _transform_gan_back.
// One thread per pixel reading its own channels: right for C <= 16 (one 64-byte line per pixel).
template <int CT, bool BF>      // CT = 16: the channel count is known, all four 16-byte loads of a pixel are issued up front
__global__ __launch_bounds__(256) void torgb_direct_kernel(const float* x, const Aff* aff, const float* w, const float* b,
                                                    float* rgb, uint8_t* img, int HW, int Crt, int nc) {
    const int C = CT > 0 ? CT : Crt;
    const int n = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const size_t px = ((size_t)n * HW + pix) * C;
    const Aff* a = aff + (size_t)n * C;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 pre[CT > 0 ? CT / 4 : 1];
    if (CT > 0) {
#pragma unroll
        for (int k = 0; k < CT / 4; ++k) pre[k] = act_load4<BF>(x, px + 4 * k);
    }
#pragma unroll
    for (int c = 0; c < C; c += 4) {
        const f32x4 v = CT > 0 ? pre[c / 4] : act_load4<BF>(x, px + c);
        const float f[4] = {fmaf(v[0], a[c].A, a[c].B), fmaf(v[1], a[c + 1].A, a[c + 1].B),
                            fmaf(v[2], a[c + 2].A, a[c + 2].B), fmaf(v[3], a[c + 3].A, a[c + 3].B)};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < nc) acc[o] = fmaf(f[j], w[o * C + c + j], acc[o]);
    }
    for (int o = 0; o < nc; ++o) {
        const float v = acc[o] + b[o];
        if (rgb) rgb[((size_t)n * nc + o) * HW + pix] = v;
        if (img) {
            float t = (v + 1.0f) * 0.5f;
            t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
            t = 255.0f * t;
            img[((size_t)n * HW + pix) * nc + o] = (uint8_t)t;
        }
    }
}

// A workgroup owns 256 consecutive pixels.  Their channels are read 32 at a time as ONE contiguous block (256 x 128
// bytes, fully coalesced float4 loads), parked in LDS with a row stride of 36 floats (conflict-free 16-byte reads),
// and each thread then walks ITS pixel's channels in order -- the canonical k-ordered fmaf chain -- out of LDS.
// (One thread reading its own pixel straight from HBM touched 64 different cache lines per load instruction and
// fetched the tensor 6.4 times at C = 32.)
template <bool BF>
__global__ __launch_bounds__(256) void torgb_kernel(const float* x, const Aff* aff, const float* w, const float* b,
                                                    float* rgb, uint8_t* img, int HW, int C, int nc) {
    constexpr int CH = 32, LS = CH + 4;
    __shared__ __attribute__((aligned(16))) float tile[256 * LS];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int p0 = blockIdx.x * 256;
    const int npx = min(256, HW - p0);
    const int pix = p0 + tid;
    const Aff* a = aff + (size_t)n * C;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += CH) {
        const int cw = min(CH, C - c0), q4 = cw >> 2;          // channels / float4s per pixel in this chunk
        if (c0) __syncthreads();
        for (int i = tid; i < npx * q4; i += 256) {
            const int pl = i / q4, cq = i - pl * q4;
            *reinterpret_cast<f32x4*>(&tile[pl * LS + cq * 4]) = act_load4<BF>(x, ((size_t)n * HW + p0 + pl) * C + c0 + cq * 4);
        }
        __syncthreads();
        if (tid < npx) {
            for (int c = 0; c < cw; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(&tile[tid * LS + c]);
                const Aff* ac = a + c0 + c;
                const float f[4] = {fmaf(v.x, ac[0].A, ac[0].B), fmaf(v.y, ac[1].A, ac[1].B),
                                    fmaf(v.z, ac[2].A, ac[2].B), fmaf(v.w, ac[3].A, ac[3].B)};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o < nc) acc[o] = fmaf(f[j], w[o * C + c0 + c + j], acc[o]);
            }
        }
    }
    if (tid >= npx) return;
    for (int o = 0; o < nc; ++o) {
        const float v = acc[o] + b[o];
        if (rgb) rgb[((size_t)n * nc + o) * HW + pix] = v;
        if (img) {
            float t = (v + 1.0f) * 0.5f;
            t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
            t = 255.0f * t;
            img[((size_t)n * HW + pix) * nc + o] = (uint8_t)t;
        }
    }
}

// feature export: NHWC (+AdaIN) -> NCHW fp32, the layout the reference returns
template <bool BF>
__global__ __launch_bounds__(256) void export_nchw_kernel(const float* x, const Aff* aff, float* out, int HW, int C) {
    __shared__ float tile[64][17];
    const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 16;
    {   // read 64 pixels x 16 channels, channel fastest
        const int pl = threadIdx.x >> 2, cq = (threadIdx.x & 3) * 4;
        if (p0 + pl < HW) {
            const f32x4 v = act_load4<BF>(x, ((size_t)n * HW + p0 + pl) * C + c0 + cq);
            float f[4] = {v[0], v[1], v[2], v[3]};
            if (aff) {
                const Aff* a = aff + (size_t)n * C + c0 + cq;
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = fmaf(f[j], a[j].A, a[j].B);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[pl][cq + j] = f[j];
        }
    }
    __syncthreads();
    {   // write pixel fastest
        const int pl = threadIdx.x & 63, cr = threadIdx.x >> 6;
        if (p0 + pl < HW)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = cr * 4 + k;
                out[((size_t)n * C + c0 + c) * HW + p0 + pl] = tile[pl][c];
            }
    }
}

// feature import: NCHW fp32 -> NHWC
template <bool BF>
__global__ __launch_bounds__(256) void import_nhwc_kernel(const float* in, float* out, int HW, int C) {
    __shared__ float tile[64][17];
    const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 16;
    {
        const int pl = threadIdx.x & 63, cr = threadIdx.x >> 6;
        if (p0 + pl < HW)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = cr * 4 + k;
                tile[pl][c] = in[((size_t)n * C + c0 + c) * HW + p0 + pl];
            }
    }
    __syncthreads();
    {
        const int pl = threadIdx.x >> 2, cq = (threadIdx.x & 3) * 4;
        if (p0 + pl < HW)
            act_store4<BF>(out, ((size_t)n * HW + p0 + pl) * C + c0 + cq, f32x4{tile[pl][cq], tile[pl][cq + 1], tile[pl][cq + 2], tile[pl][cq + 3]});
    }
}

// ------------------------------------------------------------------------------------------
// Final conv3x3 (in_c -> NCLS classes) + bias + argmax (first maximum wins).  NCLS is tiny
// (2 in the reference), so the contraction runs on the vector ALU: one thread per pixel,
// input tile staged in LDS, weights through uniform (scalar) loads.
template <int NCLS, bool BF, bool PK = false>
__global__ __launch_bounds__(256) void final_conv_kernel(const float* src0, int C0, const float* src1, int C1,
                                                         const float* wpk, const float* bias, float* logits,
                                                         uint8_t* mask, int H, int W, int tiles_x) {
    constexpr int LW = 18, CP = 20, RS = 384;   // pixel stride 20 floats, row stride 384 floats
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): XCD x takes the x-th contiguous eighth of the tiles, so the halo rows of
    // vertically neighbouring tiles come out of that XCD's L2 (rocprofv3, round 3: 1.34 GB fetched per launch against 1.07 GB of
    // input with the tiles dealt round-robin -- and this kernel sits on the HBM roof)
    const int bt = xcd_block(blockIdx.x, gridDim.x);
    const int ty = bt / tiles_x, tx = bt % tiles_x, n = blockIdx.y;
    const int y0 = ty * 16, x0 = tx * 16;
    const int ly = tid >> 4, lx = tid & 15;
    float acc[NCLS];
#pragma unroll
    for (int o = 0; o < NCLS; ++o) acc[o] = 0.0f;
    const int nblk0 = C0 >> 4, nblk = (C0 + C1) >> 4;
    // register prefetch: the 16-channel block cb+1 is loaded (unconditional, clamped; zero padding by select) while
    // block cb is multiplied out of LDS
    float4 pre[2][4];
    auto load_block = [&](int cb) {
        const bool first = cb < nblk0;
        const float* src = first ? src0 : src1;
        const int Cs = first ? C0 : C1;
        const int coff = (first ? cb : cb - nblk0) * 16;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = min(tid + it * 256, 18 * LW - 1);
            const int sy = idx / LW, sx = idx % LW;
            const int gy = y0 - 1 + sy, gx = x0 - 1 + sx;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            const size_t eidx = ((size_t)(n * H + cy) * W + cx) * Cs + coff;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 ld = act_load4<BF>(src, eidx + 4 * k);
                pre[it][k] = make_float4(ld[0], ld[1], ld[2], ld[3]);
                if (!inside) pre[it][k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    load_block(0);
    for (int cb = 0; cb < nblk; ++cb) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * 256;
            if (idx < 18 * LW) {
                float4* dst = reinterpret_cast<float4*>(smem + (idx / LW) * RS + (idx % LW) * CP);
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[k] = pre[it][k];
            }
        }
        __syncthreads();
        if (cb + 1 < nblk) load_block(cb + 1);
        const float* wb = wpk + (size_t)cb * 9 * 16 * NCLS;   // [tap][c][NCLS]
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float4* a4 = reinterpret_cast<const float4*>(smem + (ly + tap / 3) * RS + (lx + tap % 3) * CP);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 a = a4[k];
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (NCLS % 2 == 0 && PK) {
                        // round 5: the chains of two classes as ONE v_pk_fma_f32 (the value broadcast, the class pair's weights as a scalar
                        // register pair): the same fmaf per class in the same order, half the vector-ALU instructions of a kernel that was
                        // half vector-ALU bound (576 fma per pixel at 1024^2 beside 1.1 GB of reads)
#pragma unroll
                        for (int o = 0; o < NCLS; o += 2) {
                            const f32x2 w2 = {wb[(tap * 16 + k * 4 + j) * NCLS + o], wb[(tap * 16 + k * 4 + j) * NCLS + o + 1]};
                            const f32x2 r = __builtin_elementwise_fma(f32x2{av[j], av[j]}, w2, f32x2{acc[o], acc[o + 1]});
                            acc[o] = r.x; acc[o + 1] = r.y;
                        }
                    } else {
#pragma unroll
                        for (int o = 0; o < NCLS; ++o) acc[o] = fmaf(av[j], wb[(tap * 16 + k * 4 + j) * NCLS + o], acc[o]);
                    }
                }
            }
        }
        __syncthreads();
    }
    const int y = y0 + ly, x = x0 + lx;
    const size_t pix = (size_t)y * W + x, HW = (size_t)H * W;
    int best = 0;
    float bv = 0.0f;
#pragma unroll
    for (int o = 0; o < NCLS; ++o) {
        const float v = acc[o] + bias[o];
        if (logits) logits[((size_t)n * NCLS + o) * HW + pix] = v;
        if (o == 0 || v > bv) { bv = v; best = o; }
    }
    if (mask) mask[(size_t)n * HW + pix] = (uint8_t)best;
}

// ------------------------------------------------------------------------------------------
// Evaluation: confusion counts of argmax(logits) against integer labels (-1 = ignore) and the
// weighted softmax cross-entropy of SegSolver.evaluate_for_data (weight 1 on labelled pixels, 0 on ignored).
// Counts are exact integers; the loss is summed as rint(err * 2^32) in wrapping 64-bit integers, so the
// result does not depend on the order of the atomics.
template <int K>
__global__ __launch_bounds__(256) void seg_eval_kernel(const float* logits, const int8_t* labels, int HW,
                                                       unsigned long long* confusion, unsigned long long* loss_fixed) {
    __shared__ unsigned cnt[K * K];
    __shared__ unsigned long long lsum;
    const int n = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < K * K; i += 256) cnt[i] = 0u;
    if (tid == 0) lsum = 0ull;
    __syncthreads();
    unsigned long long my = 0ull;
    for (int pix = blockIdx.x * 256 + tid; pix < HW; pix += gridDim.x * 256) {
        const int l = labels[(size_t)n * HW + pix];
        float v[K];
#pragma unroll
        for (int o = 0; o < K; ++o) v[o] = logits[((size_t)n * K + o) * HW + pix];
        int best = 0;
        float m = v[0];
#pragma unroll
        for (int o = 1; o < K; ++o)
            if (v[o] > m) { m = v[o]; best = o; }
        if (l >= 0 && l < K) {
            atomicAdd(&cnt[l * K + best], 1u);
            float se = 0.0f;
#pragma unroll
            for (int o = 0; o < K; ++o) se += expf(v[o] - m);
            float vl = v[0];
#pragma unroll
            for (int o = 1; o < K; ++o) vl = l == o ? v[o] : vl;
            const float err = (m + logf(se)) - vl;              // -log_softmax(v)[l]
            my += to_fixed(err, 4294967296.0);
        }
    }
    atomicAdd(&lsum, my);
    __syncthreads();
    for (int i = tid; i < K * K; i += 256)
        if (cnt[i]) atomicAdd(&confusion[i], (unsigned long long)cnt[i]);
    if (tid == 0 && lsum) atomicAdd(&loss_fixed[n], lsum);
}

// ------------------------------------------------------------------------------------------
// Counter-based N(0,1) inputs: latents and AddNoise planes as a pure function of
// (seed, global sample index, plane, element), so that a sample is the same bytes whatever batch, rank or GPU
// count produces it.  Philox4x32-10 (Salmon et al., SC'11) + Box-Muller; the reference draws both from MXNet's
// global RNG (image_generator.py:94, networks_stylegan.py:297-300), which has no such property.
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

// out: (n, per_sample) fp32, per_sample % 4 == 0; counter = (element quad, plane, sample index lo, hi), key = seed
__global__ __launch_bounds__(256) void fill_normal_kernel(float* out, int per_sample, int n, unsigned long long first_index,
                                                          unsigned plane, unsigned long long seed) {
    const int quads = per_sample >> 2;
    const long total = (long)n * quads;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int smp = (int)(i / quads), q = (int)(i - (long)smp * quads);
        const unsigned long long idx = first_index + (unsigned long long)smp;
        unsigned c[4] = {(unsigned)q, plane, (unsigned)idx, (unsigned)(idx >> 32)};
        philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
        float v[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * 5.9604644775390625e-8f;       // (0,1), 24 bits
            const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * 5.9604644775390625e-8f;
            const float r = sqrtf(-2.0f * logf(u1));
            const float a = 6.283185307179586f * u2;
            v[2 * h] = r * cosf(a);
            v[2 * h + 1] = r * sinf(a);
        }
        *reinterpret_cast<float4*>(out + (size_t)smp * per_sample + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ========================================================================================
// host-side launchers

// ---- per-device launch state -------------------------------------------------------------
// hipFuncSetAttribute, the CU count and the occupancy answers belong to ONE device: a process may hold contexts
// on several GPUs (ImageGenerator(gpu_ids=[0, 1, ...]), reference image_generator.py:17), so every launcher keeps
// one LaunchState per (kernel instantiation, device), and a mutex makes the first-use path safe for the "one
// context per (device, host thread)" contract of include/gsa.h.
constexpr int kMaxDevices = 64;
struct LaunchState {
    bool attr_done = false;
    size_t occ_lds[8] = {0};
    int occ_k[8] = {0}, occ_n = 0;
};
static std::mutex g_launch_mu;

static int device_cus(int dev) {     // caller holds g_launch_mu
    static int cus[kMaxDevices] = {0};
    if (dev < 0 || dev >= kMaxDevices) return 256;
    if (!cus[dev] && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus[dev] = 256;
    return cus[dev];
}

// first launch of `kern` on device `dev`: allow the full 160 KB of dynamic LDS
// few statistic rows (GSA_FEWROWS=0: one row per workgroup / wave as in round 1): producers with more workgroups than
// kDirectRows ADD to row (workgroup mod kDirectRows) of the all-zero `partials` instead of writing a row each, so that
// finalize_kernel always finds at most kDirectRows rows per sample: one block per channel group, every row load in flight at
// once, no accumulator round trip and no ticket (it took 9-18 us per launch behind the 1024-row producers, ~3 us now)
static bool few_rows() {
    static const bool enabled = !(getenv("GSA_FEWROWS") && atoi(getenv("GSA_FEWROWS")) == 0);
    return enabled;
}

template <class K>
static hipError_t prepare_kernel(K kern, LaunchState& st) {
    if (st.attr_done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) st.attr_done = true;
    return e;
}

// ---- conv3x3 geometry selection --------------------------------------------------------
// Candidates per output size, most efficient first; the first one that fills the chip
// (>= 2 workgroups per CU) wins, otherwise the one with the most workgroups.
struct ConvGeom { int th, wm, wn, nt; };
static const ConvGeom kGeom4[] = {{4, 1, 4, 1}, {4, 1, 2, 1}, {4, 1, 1, 1}};
static const ConvGeom kGeom8[] = {{8, 4, 1, 4}, {8, 4, 1, 2}, {8, 4, 1, 1}};
static const ConvGeom kGeom16[] = {{16, 4, 1, 4}, {16, 4, 1, 2}, {16, 4, 1, 1}, {8, 4, 1, 4}, {8, 4, 1, 2}, {8, 4, 1, 1}};

static ConvGeom pick_geom(int H, int W, int Cout, int n) {
    const ConvGeom* cand = H == 4 ? kGeom4 : (H == 8 ? kGeom8 : kGeom16);
    const int ncand = H == 4 ? 3 : (H == 8 ? 3 : 6);
    ConvGeom best = cand[ncand - 1];
    long best_wgs = -1;
    for (int i = 0; i < ncand; ++i) {
        const ConvGeom& c = cand[i];
        const int ct = 16 * c.nt * c.wn;
        if (Cout % ct) continue;
        const long wgs = (long)(H / c.th) * (W / c.th) * (Cout / ct) * n;
        if (wgs >= 512) return c;
        if (wgs > best_wgs) { best_wgs = wgs; best = c; }
    }
    return best;
}

int conv_stat_rows(int H, int W, int Cout, int n) {   // rows of the non-specialised kernel (upper bound for the workspace)
    const ConvGeom c = pick_geom(H, W, Cout, n);
    return (H / c.th) * (W / c.th) * c.wm;
}

int post_prow(int H, int W, int C) { return (H * (W / 4) * (C / 4) + 255) / 256; }

// template arguments of the instantiation the launcher picks: "16, 16, 4, 1, 1" -- profile labels
// spell the kernel exactly as rocprofv3 prints it
const char* conv_geom_name(int H, int W, int Cout, int n) {
    static thread_local char buf[48];
    const ConvGeom c = pick_geom(H, W, Cout, n);
    snprintf(buf, sizeof buf, "%d, %d, %d, %d, %d", c.th, c.th, c.wm, c.wn, c.nt);
    return buf;
}

template <int TH, int TW, int WM, int WN, int NT, int EPI, bool SC, bool BF>
static hipError_t launch_conv_t(const ConvParams& p, int n, hipStream_t s) {
    constexpr int Q = NT * WN, COUT_T = 16 * Q;
    constexpr int TS = BF ? 128 : 256;
    constexpr int RS = (TW + 2) * (BF ? 8 : 16) + (BF ? 4 : 8);
    constexpr int NBUF = (Q <= 2 && !SC && (TH == 16 || GSA_DB_SMALL)) ? 2 : 1;   // must match the kernel's DB
    // resident weights: one channel group and a whole panel of at most 40 KB (see the kernel)
    const int nblk_all = (p.C0 + p.C1) / 16;
    static const bool wres_enabled = !(getenv("GSA_WRES") && atoi(getenv("GSA_WRES")) == 0);
    const bool wres = wres_enabled && NBUF == 2 && p.Cout == COUT_T && (size_t)nblk_all * Q * 9 * TS * sizeof(float) <= 40 * 1024;
    if (NBUF == 1 && p.C0 > 512 && p.aff0) return hipErrorInvalidValue;   // AdaIN table registers sized for <= 512 channels
    const int wslots = wres ? nblk_all * Q * 9 * TS : NBUF * Q * 9 * TS;
    const size_t lds = sizeof(float) * (NBUF * (TH + 2) * RS + wslots + (SC ? Q * TS : 0)) +
                       (p.aff0 ? sizeof(float4) * (NBUF == 2 ? 32 : p.C0) : 0);   // double-buffered form: 2 x 16 AdaIN entries
    auto kern = conv3x3_mfma<TH, TW, WM, WN, NT, EPI, SC, BF>;
    if (p.device < 0 || p.device >= kMaxDevices) return hipErrorInvalidDevice;
    static LaunchState states[kMaxDevices];
    int num_cus = 0, wgs_per_cu = 0;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        LaunchState& st = states[p.device];
        hipError_t e = prepare_kernel(kern, st);
        if (e != hipSuccess) return e;
        num_cus = device_cus(p.device);
        // resident workgroups per CU for this LDS footprint (a few distinct footprints per instantiation: cached)
        for (int i = 0; i < st.occ_n; ++i)
            if (st.occ_lds[i] == lds) wgs_per_cu = st.occ_k[i];
        if (!wgs_per_cu) {
            int k = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&k, reinterpret_cast<const void*>(kern), 64 * WM * WN, lds);
            if (e != hipSuccess) return e;
            wgs_per_cu = k < 1 ? 1 : (k > 8 ? 8 : k);
            if (st.occ_n < 8) { st.occ_lds[st.occ_n] = lds; st.occ_k[st.occ_n] = wgs_per_cu; ++st.occ_n; }
            if (getenv("GSA_VERBOSE"))
                fprintf(stderr, "gsa: conv3x3_mfma<%d,%d,%d,%d,%d,%d,%d,%d> lds %zu B%s -> %d workgroups/CU\n", TH, TW, WM, WN, NT, EPI, (int)SC,
                        (int)BF, lds, wres ? " (resident weights)" : "", k);
        }
    }
    ConvParams q = p;
    q.w_resident = wres ? 1 : 0;
    q.tiles_x = p.W / TW;
    q.tiles_y = p.H / TH;
    q.groups = p.Cout / COUT_T;
    q.prow = q.tiles_x * q.tiles_y * WM;
    q.total_tiles = q.tiles_x * q.tiles_y * q.groups * n;
    // persistent workgroups pay off for short tiles (<= 2 channel blocks: the cross-tile prefetch hides the
    // prologue); long tiles pipeline inside the tile already and run ~8 % faster one tile per workgroup
    const bool persistent = (p.C0 + p.C1) <= 32 && q.total_tiles > num_cus * wgs_per_cu;
    const int grid = persistent ? num_cus * wgs_per_cu : q.total_tiles;
    static const bool direct_enabled = !(getenv("GSA_STATS_DIRECT") && atoi(getenv("GSA_STATS_DIRECT")) == 0);
    q.stats_direct = (direct_enabled && EPI == EPI_SYNTH && NBUF == 2 && persistent && p.partials != nullptr) ? 1 : 0;
    if (q.stats_direct) q.prow = kDirectRows;      // the rows are all zero between layers (finalize_kernel clears what it read)
    if (p.stat_rows_host) *p.stat_rows_host = q.prow;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, s, q);
    return hipGetLastError();
}

template <int TH, int TW, int WM, int WN, int NT, bool BF>
static hipError_t launch_conv_b(const ConvParams& p, int epi, bool sc, int n, hipStream_t s) {
    if (sc) {
        if (epi != EPI_DEC) return hipErrorInvalidValue;
        return launch_conv_t<TH, TW, WM, WN, NT, EPI_DEC, true, BF>(p, n, s);
    }
    switch (epi) {
        case EPI_RAW: return launch_conv_t<TH, TW, WM, WN, NT, EPI_RAW, false, BF>(p, n, s);
        case EPI_SYNTH: return launch_conv_t<TH, TW, WM, WN, NT, EPI_SYNTH, false, BF>(p, n, s);
        case EPI_DEC: return launch_conv_t<TH, TW, WM, WN, NT, EPI_DEC, false, BF>(p, n, s);
    }
    return hipErrorInvalidValue;
}

template <int TH, int TW, int WM, int WN, int NT>
static hipError_t launch_conv_e(const ConvParams& p, int epi, bool sc, int n, hipStream_t s) {
    return p.bf16 ? launch_conv_b<TH, TW, WM, WN, NT, true>(p, epi, sc, n, s)
                  : launch_conv_b<TH, TW, WM, WN, NT, false>(p, epi, sc, n, s);
}

// ---- 4-way K split (static rule: the same one the oracle applies) -------------------------------------------------------
bool conv_uses_ksplit(const ConvParams& p, bool sc) {
    static const bool enabled = !(getenv("GSA_KSPLIT") && atoi(getenv("GSA_KSPLIT")) == 0);
    return enabled && !sc && p.src1 == nullptr && p.C1 == 0 && p.C0 >= 64 && p.C0 % 64 == 0 && (p.H <= 8 || (p.H <= 32 && p.Cout <= 32)) && p.H == p.W && p.Cout % 16 == 0;
}

template <int TH, int EPI, bool BF, int PS>
static hipError_t launch_ksplit_t(const ConvParams& p, int n, hipStream_t s) {
    constexpr int MT = (TH / 4) * (TH / 4), PX = BF ? 8 : 16, TS = BF ? 128 : 256;
    constexpr int RS = (TH + 2) * PX + (BF ? 4 : 8), IMG = (TH + 2) * RS, SEG = 9 * TS;
    const size_t lds = sizeof(float) * (2 * 4 * IMG + 2 * 4 * SEG) + sizeof(float4) * (2 * 4 * 16 + 4 * MT * 64) + 2 * 16 * sizeof(unsigned long long) + 16;
    auto kern = conv3x3_ksplit<TH, EPI, BF, PS>;
    if (p.device < 0 || p.device >= kMaxDevices) return hipErrorInvalidDevice;
    static LaunchState states[kMaxDevices];
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        hipError_t e = prepare_kernel(kern, states[p.device]);
        if (e != hipSuccess) return e;
    }
    ConvParams q = p;
    q.tiles_x = p.W / TH;
    q.tiles_y = p.H / TH;
    q.groups = p.Cout / 16;
    q.prow = q.tiles_x * q.tiles_y * MT;
    q.stats_direct = 0;
    if (p.stat_rows_host) *p.stat_rows_host = q.prow;
    hipLaunchKernelGGL(kern, dim3(q.tiles_x * q.tiles_y, q.groups, n), dim3(256 * PS), lds, s, q);
    return hipGetLastError();
}

// waves per workgroup of the 8x8 form: 8 (patch split) unless GSA_KSPLIT_PS=1 (speed only, same bits)
static int ksplit_ps() {
    static const int forced = getenv("GSA_KSPLIT_PS") ? atoi(getenv("GSA_KSPLIT_PS")) : 2;
    return forced == 1 ? 1 : 2;
}

static hipError_t launch_ksplit(const ConvParams& p, int epi, int n, hipStream_t s) {
#define GSA_KS(TH, BF, PS) \
    if ((p.H == 4 ? 4 : 8) == TH && (p.bf16 != 0) == BF && (TH == 4 ? 1 : ksplit_ps()) == PS) { \
        if (epi == EPI_RAW) return launch_ksplit_t<TH, EPI_RAW, BF, PS>(p, n, s); \
        if (epi == EPI_SYNTH) return launch_ksplit_t<TH, EPI_SYNTH, BF, PS>(p, n, s); \
        return launch_ksplit_t<TH, EPI_DEC, BF, PS>(p, n, s); \
    }
    GSA_KS(4, false, 1) GSA_KS(8, false, 1) GSA_KS(8, false, 2) GSA_KS(4, true, 1) GSA_KS(8, true, 1) GSA_KS(8, true, 2)
#undef GSA_KS
    return hipErrorInvalidValue;
}

// ---- Winograd form ---------------------------------------------------------------------------------------
// The rule is static (layer shape and arithmetic mode only, never the batch size): it is part of the canonical
// arithmetic and the oracle applies the same one (oracle/c/gsa_oracle.c use_wino).
bool conv_uses_wino(const ConvParams& p, int epi, bool sc) {
    static const bool enabled = !(getenv("GSA_WINO") && atoi(getenv("GSA_WINO")) == 0);
    return enabled && p.wino != nullptr && !p.bf16 && !sc && !p.up && p.src1 == nullptr && p.C1 == 0 && epi != EPI_RAW &&
           (p.H >= 64 || (p.H >= 32 && p.Cout >= 64) || (p.H >= 16 && p.Cout >= 256)) && p.H == p.W && p.H % 16 == 0 && p.Cout % 16 == 0 && p.C0 % 16 == 0;
}

// output channels per workgroup = 16*NT.  NT = 2 (one workgroup per CU with the whole register file, the input transform and
// the staging shared by 32 output channels) measured SLOWER than two NT = 1 workgroups per CU (g.512.conv_2: 0.35 vs 0.30 ms):
// a lone wave per SIMD does not hide its own LDS / MFMA latencies.  Kept for A/B runs (GSA_WINO_NT=2); speed only, same bits.
static int wino_nt(const ConvParams& p) {
    static const int forced = getenv("GSA_WINO_NT") ? atoi(getenv("GSA_WINO_NT")) : 1;
    return (forced >= 2 && p.Cout % 32 == 0) ? 2 : 1;
}

template <int EPI, int NT, bool CHUNK, bool AFF, int GW, int TW = 1>
static hipError_t launch_wino_t(const ConvParams& p, int n, hipStream_t s) {
    constexpr int RS = (16 * TW + 2) * 16 + 4, SEG = 16 * 256;
    const int nblk = p.C0 / 16;
    const bool wres = (size_t)nblk * NT * SEG * sizeof(float) <= (NT == 1 ? 36 : 72) * 1024;      // whole panel of the group resident (<= 32 input channels)
    const size_t lds = sizeof(float) * (2 * 18 * RS + (wres ? nblk : 2) * GW * NT * SEG) + (CHUNK ? 0 : 32 * sizeof(float4));
    auto kern = conv3x3_wino<EPI, NT, CHUNK, AFF, GW, TW>;
    if (TW == 2 && !wres) return hipErrorInvalidValue;      // two tiles per workgroup: resident weight panel only
    if (p.device < 0 || p.device >= kMaxDevices) return hipErrorInvalidDevice;
    static LaunchState states[kMaxDevices];
    int num_cus = 0, wgs_per_cu = 0;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        LaunchState& st = states[p.device];
        hipError_t e = prepare_kernel(kern, st);
        if (e != hipSuccess) return e;
        num_cus = device_cus(p.device);
        for (int i = 0; i < st.occ_n; ++i)
            if (st.occ_lds[i] == lds) wgs_per_cu = st.occ_k[i];
        if (!wgs_per_cu) {
            int k = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&k, reinterpret_cast<const void*>(kern), 256 * GW * TW, lds);
            if (e != hipSuccess) return e;
            wgs_per_cu = k < 1 ? 1 : (k > 8 ? 8 : k);
            if (st.occ_n < 8) { st.occ_lds[st.occ_n] = lds; st.occ_k[st.occ_n] = wgs_per_cu; ++st.occ_n; }
            if (getenv("GSA_VERBOSE")) fprintf(stderr, "gsa: conv3x3_wino<%d,%d,%d,%d,%d> lds %zu B%s -> %d workgroups/CU\n", EPI, NT, (int)CHUNK, (int)AFF, GW, lds, wres ? " (resident weights)" : "", k);
        }
    }
    ConvParams q = p;
    q.wpk = p.wino;
    q.w_resident = wres ? 1 : 0;
    q.tiles_x = p.W / (16 * TW);
    q.tiles_y = p.H / 16;
    q.groups = p.Cout / (16 * NT * GW);      // workgroup-level groups: GW channel groups of 16*NT each per workgroup
    q.prow = q.tiles_x * q.tiles_y * 4 * TW;
    q.total_tiles = q.tiles_x * q.tiles_y * n;         // per output-channel group
    // persistent workgroups; a workgroup stays inside its channel group
    const int slots = std::max(1, num_cus * wgs_per_cu / q.groups);
    // Round 3: persistent for EVERY channel count (round 2: only tiles of <= 2 channel blocks).  A one-tile workgroup pays its cold
    // start -- first loads out of HBM, nothing to overlap them with -- per tile; a workgroup that walks a contiguous tile range keeps
    // its two-item prefetch full across tiles: g.256.conv_2 0.292 -> 0.249 ms, d.cvt_6 0.150 -> 0.133, g.128.conv_2 0.247 -> 0.233,
    // g.64 / g.32.conv_2 0.233 / 0.229 -> 0.224 / 0.225 (same box).  The groups of a tile range still meet in one XCD's L2: workgroup
    // (x, g) has the linear index x + g * gx and gx is a multiple of 8.  GSA_WINO_PERS=<max input channels> restores a limit.
    static const int pers_c = getenv("GSA_WINO_PERS") ? atoi(getenv("GSA_WINO_PERS")) : 1 << 30;
    const bool persistent = p.C0 <= pers_c && q.total_tiles > slots;
    const int gx = persistent ? slots : q.total_tiles;
    static const bool direct_enabled = !(getenv("GSA_STATS_DIRECT") && atoi(getenv("GSA_STATS_DIRECT")) == 0);
    // direct statistics: always for persistent workgroups; for one-tile workgroups when the layer would otherwise write
    // more than kDirectRows rows (few_rows).  The rows are all zero between layers (finalize_kernel clears what it read).
    q.stats_direct = (direct_enabled && EPI == EPI_SYNTH && p.partials != nullptr && (persistent || (few_rows() && q.prow > kDirectRows))) ? 1 : 0;
    if (q.stats_direct) q.prow = kDirectRows;
    if (p.stat_rows_host) *p.stat_rows_host = q.prow;
    // groups of a tile side by side on one XCD when the layer's whole U (16 * Cin * Cout floats) fits comfortably in a 4 MB L2
    static const bool gm_enabled = !(getenv("GSA_WINO_GM") && atoi(getenv("GSA_WINO_GM")) == 0);
    q.group_minor = (gm_enabled && !persistent && q.groups > 1 && q.total_tiles % 8 == 0 &&
                     (size_t)16 * p.C0 * p.Cout * sizeof(float) <= (size_t)2 << 20) ? 1 : 0;
    if (q.group_minor) hipLaunchKernelGGL(kern, dim3(q.total_tiles * q.groups), dim3(256 * GW * TW), lds, s, q);
    else hipLaunchKernelGGL(kern, dim3(gx, q.groups), dim3(256 * GW * TW), lds, s, q);
    return hipGetLastError();
}

// ---- Winograd F(4x4,3x3) form (round 4).  Static rule (oracle/c/gsa_oracle.c use_wino43): a layer without a residual epilogue
// that takes the Winograd form, has at least 64 input channels (the streamed-weight layers) and an output of at least 32 px.
// OPT-IN: GSA_WINO43=1 selects it -- a different canonical arithmetic (the oracle has GSAO_WINO43=1), bit-exact against that oracle
// and within the 1e-3 tolerance, but measured 15-20 % SLOWER than F(2x2,3x3) on these layers: the
// product keeps F(2x2,3x3).
#if GSA_EXPERIMENTS
bool wino43_enabled() {
    static const bool enabled = getenv("GSA_WINO43") && atoi(getenv("GSA_WINO43")) != 0;
    return enabled;
}
bool conv_uses_wino43(const ConvParams& p, int epi, bool sc) {
    return wino43_enabled() && conv_uses_wino(p, epi, sc) && p.C0 >= 64 && p.H >= 32 && p.resid == nullptr;
}

template <int EPI>
static hipError_t launch_wino43_t(const ConvParams& p, int n, hipStream_t s) {
    constexpr int IMG = 12 * 64 * 4, SEG = 36 * 128;      // conv3x3_wino43: image buffer of a wave, weight panel of an item (8-channel blocks)
    const size_t lds = sizeof(float) * (4 * 2 * IMG + 2 * SEG);
    if (!p.zeros) return hipErrorInvalidValue;
    auto kern = conv3x3_wino43<EPI>;
    if (p.device < 0 || p.device >= kMaxDevices) return hipErrorInvalidDevice;
    static LaunchState states[kMaxDevices];
    int num_cus = 0;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        hipError_t e = prepare_kernel(kern, states[p.device]);
        if (e != hipSuccess) return e;
        num_cus = device_cus(p.device);
    }
    ConvParams q = p;
    q.wpk = p.wino;
    q.tiles_x = p.W / 16;
    q.tiles_y = p.H / 16;
    q.groups = p.Cout / 16;
    q.total_tiles = q.tiles_x * q.tiles_y * n;         // 16x16 wave tiles per output-channel group
    const int witems = (q.total_tiles + 3) / 4;        // a workgroup item = 4 wave tiles
    // one workgroup per CU (the whole register file per wave); a workgroup stays inside its channel group and walks a contiguous
    // range of items; workgroup (x, g) has the linear index x + g * gx: with gx a multiple of 8 the groups of a range share an XCD's L2
    const int slots = std::max(1, num_cus / q.groups);
    const int gx = std::min(witems, slots);
    q.stats_direct = 1;
    q.prow = kDirectRows;      // the sums go straight to kDirectRows zeroed rows per sample (finalize_kernel clears what it read)
    if (p.stat_rows_host) *p.stat_rows_host = q.prow;
    hipLaunchKernelGGL(kern, dim3(gx, q.groups), dim3(256), lds, s, q);
    return hipGetLastError();
}

static hipError_t launch_wino43(const ConvParams& p, int epi, int n, hipStream_t s) {
    if (epi == EPI_SYNTH) return launch_wino43_t<EPI_SYNTH>(p, n, s);
    return launch_wino43_t<EPI_DEC>(p, n, s);
}

#else
bool wino43_enabled() { return false; }
bool conv_uses_wino43(const ConvParams&, int, bool) { return false; }
#endif

// staging form of the Winograd kernel: 16-byte chunks from 64 input channels on, whole pixels below (measured; speed only)
static bool wino_chunk(const ConvParams& p) {
    static const int forced = getenv("GSA_WINO_CHUNK") ? atoi(getenv("GSA_WINO_CHUNK")) : -1;
    return forced >= 0 ? forced != 0 : p.C0 >= 64;
}

// channel groups per workgroup: 2 (one staged image multiplied by two groups' weights) for the layers with an LDS-resident weight
// panel (<= 32 input channels) and an even number of 16-channel groups -- measured (FFHQ batch 8, same box): d.cvt_7 0.262 -> 0.233 ms,
// g.512.conv_2 0.279 -> 0.255, d.main_6.b unchanged; the streamed-weight layers (chunk staging, >= 64 input channels) were 2-10 %
// SLOWER with it (g.32...g.256.conv_2 0.234/0.237/0.248/0.294 -> 0.256/0.257/0.253/0.298: their eight waves then wait on one
// barrier for a 32 KB weight block per item) and keep one group per workgroup.  GSA_WINO_GW=1 / 2 force it (speed only, same bits).
static int wino_gw(const ConvParams& p) {
    static const int forced = getenv("GSA_WINO_GW") ? atoi(getenv("GSA_WINO_GW")) : 0;
    if (wino_nt(p) != 1 || p.Cout % 32 != 0 || forced == 1) return 1;
    return (forced >= 2 || p.C0 <= 32) ? 2 : 1;
}

#if GSA_EXPERIMENTS
// conv3x3_wino_dma: the streamed-weight layers (>= 64 input channels) with every operand staged by LDS-DMA.  Speed only -- the same
// arithmetic as conv3x3_wino, same bits -- and measured 8-10 % SLOWER (g.64.conv_2 0.250 vs 0.227 ms: ~260 cycles per DMA instruction
// with eight waves per CU, and every wave waits at the barrier for the slowest wave's DMA): opt-in, GSA_WINO_DMA=1, for A/B runs.
static bool wino_dma(const ConvParams& p, int epi) {
    static const bool enabled = getenv("GSA_WINO_DMA") && atoi(getenv("GSA_WINO_DMA")) != 0;
    return enabled && p.C0 >= 64 && p.zeros != nullptr && p.resid == nullptr && (epi == EPI_SYNTH ? p.partials != nullptr : true);
}

template <int EPI, bool AFF>
static hipError_t launch_wino_dma_t(const ConvParams& p, int n, hipStream_t s) {
    constexpr int IMGF = 21 * 256, SEG = 16 * 256;
    const size_t lds = sizeof(float) * (2 * IMGF + 2 * SEG);
    auto kern = conv3x3_wino_dma<EPI, AFF>;
    if (p.device < 0 || p.device >= kMaxDevices) return hipErrorInvalidDevice;
    static LaunchState states[kMaxDevices];
    int num_cus = 0;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        hipError_t e = prepare_kernel(kern, states[p.device]);
        if (e != hipSuccess) return e;
        num_cus = device_cus(p.device);
    }
    ConvParams q = p;
    q.wpk = p.wino;
    q.tiles_x = p.W / 16;
    q.tiles_y = p.H / 16;
    q.groups = p.Cout / 16;
    q.total_tiles = q.tiles_x * q.tiles_y * n;
    // persistent workgroups, two per CU, each inside its channel group; (x, g) -> linear index x + g * gx: the groups of a tile range share an XCD's L2
    const int slots = std::max(1, num_cus * 2 / q.groups);
    const int gx = std::min(q.total_tiles, slots);
    q.stats_direct = 1;
    q.prow = kDirectRows;
    if (p.stat_rows_host) *p.stat_rows_host = q.prow;
    hipLaunchKernelGGL(kern, dim3(gx, q.groups), dim3(256), lds, s, q);
    return hipGetLastError();
}

static hipError_t launch_wino_dma(const ConvParams& p, int epi, int n, hipStream_t s) {
    if (epi == EPI_SYNTH) return p.aff0 ? launch_wino_dma_t<EPI_SYNTH, true>(p, n, s) : launch_wino_dma_t<EPI_SYNTH, false>(p, n, s);
    return p.aff0 ? launch_wino_dma_t<EPI_DEC, true>(p, n, s) : launch_wino_dma_t<EPI_DEC, false>(p, n, s);
}

// two 16x16 tiles per 8-wave workgroup (conv3x3_wino<..., TW = 2>): the layers with ONE output-channel group and a resident panel
// (16 output channels from <= 32 inputs: g.1024.conv_2, d.cvt_8, d.main_7.b), whole-pixel staging.  GSA_WINO_TW=2 selects it (speed only).
static int wino_tw(const ConvParams& p) {
    static const int forced = getenv("GSA_WINO_TW") ? atoi(getenv("GSA_WINO_TW")) : 1;
    return (forced == 2 && p.Cout == 16 && p.C0 <= 32 && p.W % 32 == 0 && wino_nt(p) == 1 && !wino_chunk(p)) ? 2 : 1;
}

#else
static bool wino_dma(const ConvParams&, int) { return false; }
static int wino_tw(const ConvParams&) { return 1; }
#endif

static hipError_t launch_wino(const ConvParams& p, int epi, int n, hipStream_t s) {
#if GSA_EXPERIMENTS      // an explicitly selected experiment wins over the lean kernels
    if (wino_dma(p, epi) && wino_nt(p) == 1) return launch_wino_dma(p, epi, n, s);
    if (wino_tw(p) == 2) {
        if (epi == EPI_SYNTH) return p.aff0 ? launch_wino_t<EPI_SYNTH, 1, false, true, 1, 2>(p, n, s) : launch_wino_t<EPI_SYNTH, 1, false, false, 1, 2>(p, n, s);
        return p.aff0 ? launch_wino_t<EPI_DEC, 1, false, true, 1, 2>(p, n, s) : launch_wino_t<EPI_DEC, 1, false, false, 1, 2>(p, n, s);
    }
#endif
    if (wino_lean_applies(p, epi)) return launch_wino_lean(p, epi, n, s);
    const int nt = wino_nt(p);
    const int gw = wino_gw(p);
    const bool ch = wino_chunk(p);
#define GSA_W(EPI, AFF) \
    if (nt == 2) return ch ? launch_wino_t<EPI, 2, true, AFF, 1>(p, n, s) : launch_wino_t<EPI, 2, false, AFF, 1>(p, n, s); \
    if (gw == 2) return ch ? launch_wino_t<EPI, 1, true, AFF, 2>(p, n, s) : launch_wino_t<EPI, 1, false, AFF, 2>(p, n, s); \
    return ch ? launch_wino_t<EPI, 1, true, AFF, 1>(p, n, s) : launch_wino_t<EPI, 1, false, AFF, 1>(p, n, s);
    if (epi == EPI_SYNTH) {
        if (p.aff0) { GSA_W(EPI_SYNTH, true) }
        GSA_W(EPI_SYNTH, false)
    }
    if (p.aff0) { GSA_W(EPI_DEC, true) }
    GSA_W(EPI_DEC, false)
#undef GSA_W
}

// true: launch_conv3x3(p, epi, sc) with p.rgb_* set also writes toRGB's uint8 image -- i.e. the call reaches the lean 16 -> 16 kernel (an
// experiment switch that takes the layer elsewhere, or the F(4x4,3x3) form, rules it out: they know nothing of rgb_img)
bool conv_fuses_torgb(const ConvParams& p, int epi, bool sc, int nc) {
    if (!conv_uses_wino(p, epi, sc) || conv_uses_wino43(p, epi, sc)) return false;
    if ((wino_dma(p, epi) && wino_nt(p) == 1) || wino_tw(p) == 2) return false;
    return wino_lean_fuses_torgb(p, epi, nc);
}

// exact C++ name of the instantiation launch_conv3x3 picks (profile labels spell kernels as rocprofv3 prints them)
const char* conv3x3_kernel_name(const ConvParams& p, int epi, bool sc, int n) {
    static thread_local char buf[128];
    if (conv_uses_wino43(p, epi, sc)) {
        snprintf(buf, sizeof buf, "void gsa::conv3x3_wino43<%d>(gsa::ConvParams)", epi);
        return buf;
    }
    if (conv_uses_wino(p, epi, sc) && !(wino_dma(p, epi) && wino_nt(p) == 1) && wino_tw(p) != 2 && wino_lean_applies(p, epi)) return wino_lean_name(p, epi, n);
    if (conv_uses_wino(p, epi, sc) && wino_dma(p, epi) && wino_nt(p) == 1) {
        snprintf(buf, sizeof buf, "void gsa::conv3x3_wino_dma<%d, %s>(gsa::ConvParams)", epi, p.aff0 ? "true" : "false");
        return buf;
    }
    if (conv_uses_wino(p, epi, sc) && wino_tw(p) == 2) {
        snprintf(buf, sizeof buf, "void gsa::conv3x3_wino<%d, 1, false, %s, 1, 2>(gsa::ConvParams)", epi, p.aff0 ? "true" : "false");
        return buf;
    }
    if (conv_uses_wino(p, epi, sc)) {
        snprintf(buf, sizeof buf, "void gsa::conv3x3_wino<%d, %d, %s, %s, %d>(gsa::ConvParams)", epi, wino_nt(p), wino_chunk(p) ? "true" : "false", p.aff0 ? "true" : "false", wino_gw(p));
        return buf;
    }
    if (conv_uses_ksplit(p, sc)) {
        snprintf(buf, sizeof buf, "void gsa::conv3x3_ksplit<%d, %d, %s, %d>(gsa::ConvParams)", p.H == 4 ? 4 : 8, epi, p.bf16 ? "true" : "false", p.H == 4 ? 1 : ksplit_ps());
        return buf;
    }
    if (bf16_lean_applies(p, epi, sc)) return bf16_lean_name(p, epi, n);
    const ConvGeom c = pick_geom(p.H, p.W, p.Cout, n);
    snprintf(buf, sizeof buf, "void gsa::conv3x3_mfma<%d, %d, %d, %d, %d, %d, %s, %s>(gsa::ConvParams)", c.th, c.th, c.wm, c.wn, c.nt,
             epi, sc ? "true" : "false", p.bf16 ? "true" : "false");
    return buf;
}

hipError_t launch_conv3x3(const ConvParams& p, int epi, bool sc, int n, hipStream_t s) {
    if (p.H != p.W || (p.H & (p.H - 1)) || p.H < 4 || p.Cout % 16 || p.C0 % 16 || p.C1 % 16) return hipErrorInvalidValue;
#if GSA_EXPERIMENTS
    if (conv_uses_wino43(p, epi, sc)) return launch_wino43(p, epi, n, s);
#endif
    if (conv_uses_wino(p, epi, sc)) return launch_wino(p, epi, n, s);
    if (conv_uses_ksplit(p, sc)) return launch_ksplit(p, epi, n, s);
    if (bf16_lean_applies(p, epi, sc)) return launch_bf16_lean(p, epi, n, s);
    const ConvGeom c = pick_geom(p.H, p.W, p.Cout, n);
#define GSA_GEOM(TH, WM, WN, NT) \
    if (c.th == TH && c.wm == WM && c.wn == WN && c.nt == NT) return launch_conv_e<TH, TH, WM, WN, NT>(p, epi, sc, n, s);
    GSA_GEOM(4, 1, 4, 1) GSA_GEOM(4, 1, 2, 1) GSA_GEOM(4, 1, 1, 1)
    GSA_GEOM(8, 4, 1, 4) GSA_GEOM(8, 4, 1, 2) GSA_GEOM(8, 4, 1, 1)
    GSA_GEOM(16, 4, 1, 4) GSA_GEOM(16, 4, 1, 2) GSA_GEOM(16, 4, 1, 1)
#undef GSA_GEOM
    return hipErrorInvalidValue;
}

// Winograd F(2x2,2x2) form of the stride-2 layers: every such layer in fp32 mode (the static rule of the canonical arithmetic,
// oracle/c/gsa_oracle.c use_wino22).  GSA_WINO22=0 selects the direct 4-tap kernels -- a different arithmetic, timing only.
static bool sub_wino(const ConvParams& p) {
    static const bool enabled = !(getenv("GSA_WINO22") && atoi(getenv("GSA_WINO22")) == 0);
    return enabled && !p.bf16;
}
bool subpixel_uses_wino(const ConvParams& p) { return sub_wino(p); }
static int sub_rs(const ConvParams& p) { return p.bf16 ? 10 * 8 + 4 : 10 * 16 + (sub_wino(p) ? 4 : 8); }      // LDS row stride in 4-byte slots

template <int NT, int EPI, bool SC, bool BF, bool WINO = false>
static hipError_t launch_subpixel_t(const ConvParams& p, int n, hipStream_t s) {
    constexpr int COUT_T = 16 * NT, TS = BF ? 128 : 256;
    const size_t lds = sizeof(float) * (10 * (10 * (BF ? 8 : 16) + (BF || WINO ? 4 : 8)) + NT * 16 * TS + (SC ? NT * TS : 0)) + (p.aff0 ? sizeof(float4) * p.C0 : 0);
    auto kern = subpixel_mfma<NT, EPI, SC, BF, WINO>;
    if (p.device < 0 || p.device >= kMaxDevices) return hipErrorInvalidDevice;
    static LaunchState states[kMaxDevices];
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        hipError_t e = prepare_kernel(kern, states[p.device]);
        if (e != hipSuccess) return e;
    }
    ConvParams q = p;
    q.tiles_x = p.W / 16;
    dim3 grid((p.H / 16) * (p.W / 16), p.Cout / COUT_T, n);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, q);
    return hipGetLastError();
}

// channel blocks per item of subpixel_res: 2 when the block count is even and the doubled activation buffers still fit
static int subpixel_res_kb(const ConvParams& p);

// persistent form with the LDS-resident weight panel (subpixel_res): one 512-thread workgroup per CU
template <int NT, int EPI, bool SC, bool BF, int KB, bool WINO>
static hipError_t launch_subpixel_res_k(const ConvParams& p, int n, size_t lds, hipStream_t s) {
    auto kern = subpixel_res<NT, EPI, SC, BF, KB, false, WINO>;
    if (p.device < 0 || p.device >= kMaxDevices) return hipErrorInvalidDevice;
    static LaunchState states[kMaxDevices];
    int num_cus = 0;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        hipError_t e = prepare_kernel(kern, states[p.device]);
        if (e != hipSuccess) return e;
        num_cus = device_cus(p.device);
    }
    ConvParams q = p;
    q.tiles_x = p.W / 16;
    q.tiles_y = p.H / 16;
    q.groups = 1;
    q.total_tiles = q.tiles_x * q.tiles_y * n;
    const int grid = std::min(num_cus, (q.total_tiles + 1) / 2);
    if (WINO && !BF && subpixel_lean_applies(q, NT, EPI, SC, KB, false)) return launch_subpixel_lean(q, NT, EPI, SC, KB, false, dim3(grid), s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, q);
    return hipGetLastError();
}

// streamed-weights form: grid (workgroups per channel group, channel groups)
template <int NT, int EPI, bool SC, bool BF, bool WINO = false>
static hipError_t launch_subpixel_wst_t(const ConvParams& p, int n, size_t lds, int wgs_per_g, hipStream_t s) {
    auto kern = subpixel_res<NT, EPI, SC, BF, 1, true, WINO>;
    if (p.device < 0 || p.device >= kMaxDevices) return hipErrorInvalidDevice;
    static LaunchState states[kMaxDevices];
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        hipError_t e = prepare_kernel(kern, states[p.device]);
        if (e != hipSuccess) return e;
    }
    ConvParams q = p;
    q.tiles_x = p.W / 16;
    q.tiles_y = p.H / 16;
    q.groups = p.Cout / (16 * NT);
    q.total_tiles = q.tiles_x * q.tiles_y * n;       // per channel group
    if (WINO && !BF && subpixel_lean_applies(q, NT, EPI, SC, 1, true)) return launch_subpixel_lean(q, NT, EPI, SC, 1, true, dim3(wgs_per_g, q.groups), s);
    hipLaunchKernelGGL(kern, dim3(wgs_per_g, q.groups), dim3(512), lds, s, q);
    return hipGetLastError();
}

// Streamed form: LDS bytes and workgroups per channel group, or 0 when it does not pay (fewer than 2 tiles per half)
static size_t subpixel_wst_lds(const ConvParams& p, int ct, bool sc, int n, int* wgs_per_g) {
    static const bool enabled = !(getenv("GSA_SUBWST") && atoi(getenv("GSA_SUBWST")) == 0);
    int num_cus;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        num_cus = device_cus(p.device);
    }
    if (!enabled || p.Cout % ct) return 0;
    const int ts = p.bf16 ? 128 : 256, rs = sub_rs(p);
    const int nt = ct / 16, groups = p.Cout / ct;
    const size_t lds = sizeof(float) * ((size_t)2 * nt * 16 * ts + (sc ? (size_t)2 * nt * ts : 0) + 4 * 10 * rs) + 64 * sizeof(float4);
    const long tiles = (long)(p.H / 16) * (p.W / 16) * n;
    const int wgs = std::max(1, num_cus / groups);
    if (lds > 160 * 1024 || tiles < 4L * wgs) return 0;
    *wgs_per_g = wgs;
    return lds;
}

template <int NT, int EPI, bool SC, bool BF, bool WINO = false>
static hipError_t launch_subpixel_res_t(const ConvParams& p, int n, size_t lds, hipStream_t s) {
    return subpixel_res_kb(p) == 2 ? launch_subpixel_res_k<NT, EPI, SC, BF, 2, WINO>(p, n, lds, s)
                                   : launch_subpixel_res_k<NT, EPI, SC, BF, 1, WINO>(p, n, lds, s);
}

// LDS bytes of subpixel_res for this layer, or 0 when the layer does not qualify (several channel groups, a
// weight panel beyond ~100 KB, too few tiles to keep every CU busy)
static size_t subpixel_res_lds(const ConvParams& p, int ct, bool sc, int n) {
    static const bool enabled = !(getenv("GSA_SUBRES") && atoi(getenv("GSA_SUBRES")) == 0);
    if (!enabled || ct != p.Cout || ct > 32) return 0;      // instantiated for 16 and 32 output channels
    const int ts = p.bf16 ? 128 : 256, rs = sub_rs(p);
    const int nblk = (p.C0 + p.C1) / 16, nt = ct / 16;
    const size_t fixed = sizeof(float) * ((size_t)nblk * nt * 16 * ts + (sc ? (size_t)nblk * nt * ts : 0));
    const long tiles = (long)(p.H / 16) * (p.W / 16) * n;
    if (tiles < 4096) return 0;
    const int kb_min_blocks = sub_wino(p) ? 2 : 4;      // Winograd form: 36 MFMAs per block instead of 64 -- a two-block tile is one item
    for (int kb = (nblk % 2 == 0 && nblk >= kb_min_blocks) ? 2 : 1; kb >= 1; --kb) {     // two blocks per item pay off from 4 blocks on
        const size_t lds = fixed + sizeof(float) * 4 * kb * 10 * rs + 64 * kb * sizeof(float4);
        if (lds <= 160 * 1024) return lds;
    }
    return 0;
}

static int subpixel_res_kb(const ConvParams& p) {
    const int ts = p.bf16 ? 128 : 256, rs = sub_rs(p);
    const int nblk = (p.C0 + p.C1) / 16, nt = p.Cout / 16;
    const bool sc = p.wsc != nullptr;
    const size_t fixed = sizeof(float) * ((size_t)nblk * nt * 16 * ts + (sc ? (size_t)nblk * nt * ts : 0));
    if (nblk % 2 || nblk < (sub_wino(p) ? 2 : 4)) return 1;
    return fixed + sizeof(float) * 4 * 2 * 10 * rs + 128 * sizeof(float4) <= 160 * 1024 ? 2 : 1;
}

// widest channel tile that still gives the chip >= 2 workgroups per CU (else the narrowest)
static int subpixel_cout_tile(int H, int W, int Cout, int n, bool wino = false) {
    const long tiles = (long)(H / 16) * (W / 16) * n;
    for (int ct = wino ? 32 : 64; ct >= 16; ct /= 2)      // Winograd form: nine accumulator vectors per 16 channels -- at most 32 per workgroup
        if (Cout % ct == 0 && (tiles * (Cout / ct) >= 512 || ct == 16)) return ct;
    return 16;
}

// exact C++ name of the instantiation launch_subpixel picks (profile labels)
const char* subpixel_kernel_name(const ConvParams& p, int epi, bool sc, int n) {
    static thread_local char buf[128];
    const char* wn = sub_wino(p) ? "true" : "false";
    const int ct = subpixel_cout_tile(p.H, p.W, p.Cout, n, sub_wino(p));
    const bool res = subpixel_res_lds(p, ct, sc, n) != 0;
    int wgs_per_g = 0;
    if (res && sub_wino(p) && !p.bf16 && subpixel_lean_applies(p, ct / 16, epi, sc, subpixel_res_kb(p), false))
        return subpixel_lean_name(p, ct / 16, epi, sc, subpixel_res_kb(p), false);
    if (!res && sub_wino(p) && !p.bf16 && subpixel_wst_lds(p, ct, sc, n, &wgs_per_g) && subpixel_lean_applies(p, ct / 16, epi, sc, 1, true))
        return subpixel_lean_name(p, ct / 16, epi, sc, 1, true);
    if (res)
        snprintf(buf, sizeof buf, "void gsa::subpixel_res<%d, %d, %s, %s, %d, false, %s>(gsa::ConvParams)", ct / 16, epi, sc ? "true" : "false",
                 p.bf16 ? "true" : "false", subpixel_res_kb(p), wn);
    else if (subpixel_wst_lds(p, ct, sc, n, &wgs_per_g))
        snprintf(buf, sizeof buf, "void gsa::subpixel_res<%d, %d, %s, %s, 1, true, %s>(gsa::ConvParams)", ct / 16, epi, sc ? "true" : "false",
                 p.bf16 ? "true" : "false", wn);
    else
        snprintf(buf, sizeof buf, "void gsa::subpixel_mfma<%d, %d, %s, %s, %s>(gsa::ConvParams)", ct / 16, epi, sc ? "true" : "false",
                 p.bf16 ? "true" : "false", wn);
    return buf;
}

const char* subpixel_geom_name(int H, int W, int Cout, int n) {
    static thread_local char buf[32];
    snprintf(buf, sizeof buf, "%d", subpixel_cout_tile(H, W, Cout, n) / 16);   // NT (direct form)
    return buf;
}

// deconv 4x4 s2 p1, or nearest-x2 + conv3x3 with host-presummed weights (same kernel)
hipError_t launch_subpixel(const ConvParams& p, int epi, bool sc, int n, hipStream_t s) {
    if (p.H != 2 * p.Hs || p.W != 2 * p.Ws || p.H % 16 || p.W % 16 || p.Cout % 16 || p.C0 % 16 || p.C1 % 16) return hipErrorInvalidValue;
    if (sc && epi != EPI_DEC) return hipErrorInvalidValue;
    const bool wino = sub_wino(p);
    const int ct = subpixel_cout_tile(p.H, p.W, p.Cout, n, wino);
    if (const size_t rlds = subpixel_res_lds(p, ct, sc, n)) {
#define GSA_SUBR(NT, BF, WN) \
        if (ct == 16 * NT && (p.bf16 != 0) == BF && wino == WN) { \
            if (sc) return launch_subpixel_res_t<NT, EPI_DEC, true, BF, WN>(p, n, rlds, s); \
            if (epi == EPI_DEC) return launch_subpixel_res_t<NT, EPI_DEC, false, BF, WN>(p, n, rlds, s); \
            if (epi == EPI_RAW) return launch_subpixel_res_t<NT, EPI_RAW, false, BF, WN>(p, n, rlds, s); \
            return hipErrorInvalidValue; \
        }
        GSA_SUBR(1, false, false) GSA_SUBR(2, false, false) GSA_SUBR(1, true, false) GSA_SUBR(2, true, false)
        GSA_SUBR(1, false, true) GSA_SUBR(2, false, true)
#undef GSA_SUBR
    }
    int wgs_per_g = 0;
    if (const size_t wlds = subpixel_wst_lds(p, ct, sc, n, &wgs_per_g)) {
#define GSA_SUBW(NT, BF, WN) \
        if (ct == 16 * NT && (p.bf16 != 0) == BF && wino == WN) { \
            if (sc) return launch_subpixel_wst_t<NT, EPI_DEC, true, BF, WN>(p, n, wlds, wgs_per_g, s); \
            if (epi == EPI_DEC) return launch_subpixel_wst_t<NT, EPI_DEC, false, BF, WN>(p, n, wlds, wgs_per_g, s); \
            if (epi == EPI_RAW) return launch_subpixel_wst_t<NT, EPI_RAW, false, BF, WN>(p, n, wlds, wgs_per_g, s); \
            return hipErrorInvalidValue; \
        }
        GSA_SUBW(1, false, false) GSA_SUBW(2, false, false) GSA_SUBW(4, false, false) GSA_SUBW(1, true, false) GSA_SUBW(2, true, false) GSA_SUBW(4, true, false)
        GSA_SUBW(1, false, true) GSA_SUBW(2, false, true)
#undef GSA_SUBW
    }
#define GSA_SUB(NT) \
    if (ct == 16 * NT && wino && NT <= 2) { \
        if (sc) return launch_subpixel_t<(NT <= 2 ? NT : 1), EPI_DEC, true, false, true>(p, n, s); \
        if (epi == EPI_DEC) return launch_subpixel_t<(NT <= 2 ? NT : 1), EPI_DEC, false, false, true>(p, n, s); \
        if (epi == EPI_RAW) return launch_subpixel_t<(NT <= 2 ? NT : 1), EPI_RAW, false, false, true>(p, n, s); \
        return hipErrorInvalidValue; \
    } \
    if (ct == 16 * NT && !p.bf16) { \
        if (sc) return launch_subpixel_t<NT, EPI_DEC, true, false>(p, n, s); \
        if (epi == EPI_DEC) return launch_subpixel_t<NT, EPI_DEC, false, false>(p, n, s); \
        if (epi == EPI_RAW) return launch_subpixel_t<NT, EPI_RAW, false, false>(p, n, s); \
        return hipErrorInvalidValue; \
    } \
    if (ct == 16 * NT && p.bf16) { \
        if (sc) return launch_subpixel_t<NT, EPI_DEC, true, true>(p, n, s); \
        if (epi == EPI_DEC) return launch_subpixel_t<NT, EPI_DEC, false, true>(p, n, s); \
        if (epi == EPI_RAW) return launch_subpixel_t<NT, EPI_RAW, false, true>(p, n, s); \
        return hipErrorInvalidValue; \
    }
    GSA_SUB(4) GSA_SUB(2) GSA_SUB(1)
#undef GSA_SUB
    return hipErrorInvalidValue;
}

// rows per thread of the blurred form: 4 from 64 px on (enough threads to fill the chip), else the one-row kernel
static int post_rpt(const PostParams& p) {
    static const int forced = getenv("GSA_POST_RPT") ? atoi(getenv("GSA_POST_RPT")) : -1;
    if (!p.blur || !p.src_per_sample || p.H % 8) return 1;
    if (forced >= 0) return forced == 8 ? 8 : (forced == 4 ? 4 : (forced == 2 ? 2 : 1));
    return p.H >= 64 ? 4 : 1;
}

// workgroups per sample of the form launch_post picks
// row groups per thread of post_rows_kernel<4> (GSA_POST_NG; default 1).  Measured, FFHQ batch 8, ms per launch at 1024 / 512 / 256 /
// 128 px, kernels serialized: 1 group 0.274 / 0.133 / 0.061 / 0.031; 2: 0.260 / 0.127 / 0.058 / 0.030; 4: 0.257 / 0.117 / 0.058 / 0.032;
// 8: 0.253 / 0.107 / 0.077 / 0.046 -- the best choice per size saves 0.04 ms of serialized kernel time per step, but the STEP (decoder
// running beside the synthesis) was 0-1 % slower with it in four of four same-box comparisons (1257-1265 vs 1254-1256 pairs/s): fewer,
// longer-lived workgroups leave the concurrent stream less room.  Not adopted.
static int post_row_groups(const PostParams& p) {
    static const int forced = getenv("GSA_POST_NG") ? atoi(getenv("GSA_POST_NG")) : -1;
    if (post_rpt(p) != 4) return 1;
    int ng = forced > 0 ? forced : 1;
    while (ng > 1 && p.H % (4 * ng)) ng >>= 1;
    return ng;
}

static int post_blocks(const PostParams& p) {
    if (post_dma_applies(p)) return post_dma_blocks(p);
    const int rpt = post_rpt(p);
    return rpt == 1 ? post_prow(p.H, p.W, p.C) : ((p.H / (rpt * post_row_groups(p))) * (p.W / 4) * (p.C / 4) + 255) / 256;
}

int post_rows_used(const PostParams& p) {
    const int blocks = post_blocks(p);
    return few_rows() ? std::min(blocks, kDirectRows) : blocks;
}

hipError_t launch_post(const PostParams& p, int n, hipStream_t s) {
    if (p.W % 4 || p.C % 4) return hipErrorInvalidValue;
    PostParams q = p;
    const int rpt = post_rpt(p);
    q.prow = post_rows_used(p);
    q.row_groups = post_row_groups(p);
    if (post_dma_applies(p)) return launch_post_dma(q, n, s);
    dim3 grid(post_blocks(p), n);
    const size_t lds = sizeof(unsigned long long) * 2 * p.C;
    // the packed-arithmetic form (gsa_post_lean.hip): four rows per thread, a wave's 64 threads inside one row group
    if (rpt == 4 && post_pk_mode() > 0 && ((p.W / 4) * (p.C / 4)) % 64 == 0)
        return launch_post_pk(q, grid, lds + sizeof(float) * 9 * p.C, s);
#define GSA_POST(BF) \
    if (rpt == 8) hipLaunchKernelGGL((post_rows_kernel<8, BF>), grid, dim3(256), lds, s, q); \
    else if (rpt == 4) hipLaunchKernelGGL((post_rows_kernel<4, BF>), grid, dim3(256), lds, s, q); \
    else if (rpt == 2) hipLaunchKernelGGL((post_rows_kernel<2, BF>), grid, dim3(256), lds, s, q); \
    else hipLaunchKernelGGL(post_kernel<BF>, grid, dim3(256), lds, s, q);
    if (p.bf16) { GSA_POST(true) } else { GSA_POST(false) }
#undef GSA_POST
    return hipGetLastError();
}

// whole-plane K-split tiles (4 and 8 px) finalize in the kernel when the caller supplies the finalize operands (GSA_FUSEFIN=0: never)
bool conv_fuses_finalize(const ConvParams& p, int epi, bool sc) {
    static const bool enabled = !(getenv("GSA_FUSEFIN") && atoi(getenv("GSA_FUSEFIN")) == 0);
    return enabled && epi == EPI_SYNTH && conv_uses_ksplit(p, sc) && p.H == p.W && p.H <= 8 && !conv_uses_wino(p, epi, sc);
}
// post_fin_kernel: planes of at most 32 x 32 pixels with a multiple of 16 channels (GSA_FUSEFIN=0: the two separate launches)
bool post_fuses_finalize(const PostParams& p) {
    static const bool enabled = !(getenv("GSA_FUSEFIN") && atoi(getenv("GSA_FUSEFIN")) == 0);
    return enabled && p.H * p.W <= 1024 && p.C % 16 == 0 && p.W % 4 == 0;
}
hipError_t launch_post_fin(const PostParams& p, const FinalizeParams& f, int n, hipStream_t s) {
    if (!post_fuses_finalize(p)) return hipErrorInvalidValue;
    dim3 grid(p.C / 16, n);
    if (p.bf16) hipLaunchKernelGGL(post_fin_kernel<true>, grid, dim3(256), 0, s, p, f);
    else hipLaunchKernelGGL(post_fin_kernel<false>, grid, dim3(256), 0, s, p, f);
    return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeParams& p, int n, hipStream_t s) {
    // ~16 blocks per (sample, channel group): more blocks contend on the same few atomics (C = 16: 16 addresses),
    // fewer leave each thread a long serial chain of row loads
    const int rpb = std::min(256, std::max(64, (p.prow + 15) / 16));
    const int zb = (p.prow + rpb - 1) / rpb;   // prow == 0: the producer already summed into acc
    dim3 grid((p.C + 63) / 64, n, zb < 1 ? 1 : zb);
    hipLaunchKernelGGL(finalize_kernel, grid, dim3(256), 0, s, p, rpb);
    return hipGetLastError();
}

hipError_t launch_pixelnorm(const float* z, float* out, int n, int L, hipStream_t s) {
    hipLaunchKernelGGL(pixelnorm_kernel, dim3(n), dim3(64), sizeof(float) * L, s, z, out, n, L);
    return hipGetLastError();
}

hipError_t launch_dense(const float* x, const float* WT, const float* b, float* y, int n, int K, int J, int act, hipStream_t s) {
    // 256 K rows per pass: the eight mapping layers are a serial chain at the head of every step, each pass a global load ->
    // LDS -> barrier -> 256-step fmaf chain round trip (two passes per layer instead of four: 11.3 -> ~7 us per layer)
    if (K % 256 || J % 4) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * (256 * 16 + 16 * 256 + 256);
    hipLaunchKernelGGL((dense_lds_kernel<false, 16, 256>), dim3((J + 15) / 16), dim3(256), lds, s, x, WT, b, y, n, K, J, act,
                       (const float*)nullptr, (const float*)nullptr, (const int*)nullptr);
    return hipGetLastError();
}

// the fused mapping network applies when its L/16 workgroups are certainly co-resident (see mapping_kernel)
bool mapping_fused(int L, int device) {
    static const bool enabled = !(getenv("GSA_MAPFUSE") && atoi(getenv("GSA_MAPFUSE")) == 0);
    if (!enabled || L % 64 || L > 512 || L < 64) return false;
    int num_cus;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        num_cus = device_cus(device);
    }
    return L / 16 <= num_cus / 2;
}

hipError_t launch_mapping(const float* z, float* const* wt, float* const* b, unsigned long long* const* ll, float* out, unsigned* ctl,
                          int n, int L, int device, hipStream_t s, int drop_workgroups) {
    if (!mapping_fused(L, device)) return hipErrorInvalidValue;
    MappingParams p;
    p.z = z;
    for (int i = 0; i < 8; ++i) { p.wt[i] = wt[i]; p.b[i] = b[i]; }
    p.ll[0] = ll[0]; p.ll[1] = ll[1];
    p.out = out; p.ctl = ctl; p.n = n; p.L = L;
    auto kern = mapping_kernel;
    static LaunchState states[kMaxDevices];
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        hipError_t e = prepare_kernel(kern, states[device]);
        if (e != hipSuccess) return e;
    }
    const size_t lds = sizeof(float) * (size_t)(16 * (L + 4) + 16 * L + 16);
    // slices of 16-sample chunks side by side, as many as stay co-resident (the exchange spins)
    int num_cus;
    {
        std::lock_guard<std::mutex> lk(g_launch_mu);
        num_cus = device_cus(device);
    }
    const int slices = std::max(1, std::min(std::min((n + 15) / 16, kMapSlices), (num_cus / 2) / (L / 16)));
    // drop_workgroups > 0 (fault injection, tests only): the last workgroups are not launched, their columns never arrive and
    // every other workgroup's wait times out -- the condition gsa_check must report
    hipLaunchKernelGGL(kern, dim3(std::max(1, L / 16 - drop_workgroups), slices), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_styles(const float* w, const float* avg, const float* psi, const float* WT, const float* b,
                         const int* col_layer, float* styles, int n, int K, int J, hipStream_t s) {
    if (K % 128 || J % 4) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * (128 * 64 + 16 * 128 + 128);
    hipLaunchKernelGGL((dense_lds_kernel<true, 64, 128>), dim3((J + 63) / 64, std::min((n + 15) / 16, 8)), dim3(256), lds, s, w, WT, b, styles, n, K, J, 0, avg, psi, col_layer);
    return hipGetLastError();
}

hipError_t launch_torgb(const float* x, const Aff* aff, const float* w, const float* b, float* rgb, uint8_t* img,
                        int n, int H, int W, int C, int nc, int bf16, hipStream_t s) {
    if (nc > 4 || C % 4) return hipErrorInvalidValue;
    const int HW = H * W;
    const dim3 grid((HW + 255) / 256, n);
#define GSA_RGB(BF) \
    if (C == 16) hipLaunchKernelGGL((torgb_direct_kernel<16, BF>), grid, dim3(256), 0, s, x, aff, w, b, rgb, img, HW, C, nc); \
    else if (C < 16) hipLaunchKernelGGL((torgb_direct_kernel<0, BF>), grid, dim3(256), 0, s, x, aff, w, b, rgb, img, HW, C, nc); \
    else hipLaunchKernelGGL(torgb_kernel<BF>, grid, dim3(256), 0, s, x, aff, w, b, rgb, img, HW, C, nc);
    if (bf16) { GSA_RGB(true) } else { GSA_RGB(false) }
#undef GSA_RGB
    return hipGetLastError();
}

hipError_t launch_export_nchw(const float* x, const Aff* aff, float* out, int n, int H, int W, int C, int bf16, hipStream_t s) {
    const int HW = H * W;
    if (bf16) hipLaunchKernelGGL(export_nchw_kernel<true>, dim3((HW + 63) / 64, C / 16, n), dim3(256), 0, s, x, aff, out, HW, C);
    else hipLaunchKernelGGL(export_nchw_kernel<false>, dim3((HW + 63) / 64, C / 16, n), dim3(256), 0, s, x, aff, out, HW, C);
    return hipGetLastError();
}

hipError_t launch_import_nhwc(const float* in, float* out, int n, int H, int W, int C, int bf16, hipStream_t s) {
    const int HW = H * W;
    if (bf16) hipLaunchKernelGGL(import_nhwc_kernel<true>, dim3((HW + 63) / 64, C / 16, n), dim3(256), 0, s, in, out, HW, C);
    else hipLaunchKernelGGL(import_nhwc_kernel<false>, dim3((HW + 63) / 64, C / 16, n), dim3(256), 0, s, in, out, HW, C);
    return hipGetLastError();
}

// class pairs as packed fma in the fp32 final conv (speed only, same bits; GSA_FINAL_PK=0: the scalar chains)
static bool final_pk() {
    static const bool on = !(getenv("GSA_FINAL_PK") && atoi(getenv("GSA_FINAL_PK")) == 0);
    return on;
}

template <int NCLS>
static hipError_t launch_final_t(const float* src0, int C0, const float* src1, int C1, const float* wpk, const float* bias,
                                 float* logits, uint8_t* mask, int n, int H, int W, int bf16, hipStream_t s) {
    const size_t lds = sizeof(float) * 18 * 384;
    if (bf16) hipLaunchKernelGGL((final_conv_kernel<NCLS, true>), dim3((H / 16) * (W / 16), n), dim3(256), lds, s, src0, C0, src1, C1, wpk,
                                 bias, logits, mask, H, W, W / 16);
    else if (NCLS % 2 == 0 && final_pk()) hipLaunchKernelGGL((final_conv_kernel<NCLS, false, true>), dim3((H / 16) * (W / 16), n), dim3(256), lds, s, src0, C0, src1, C1, wpk,
                            bias, logits, mask, H, W, W / 16);
    else hipLaunchKernelGGL((final_conv_kernel<NCLS, false>), dim3((H / 16) * (W / 16), n), dim3(256), lds, s, src0, C0, src1, C1, wpk,
                            bias, logits, mask, H, W, W / 16);
    return hipGetLastError();
}

hipError_t launch_final_conv(const float* src0, int C0, const float* src1, int C1, const float* wpk, const float* bias,
                             float* logits, uint8_t* mask, int n, int H, int W, int ncls, int bf16, hipStream_t s) {
    if (H % 16 || W % 16 || C0 % 16 || C1 % 16) return hipErrorInvalidValue;
    switch (ncls) {
        case 1: return launch_final_t<1>(src0, C0, src1, C1, wpk, bias, logits, mask, n, H, W, bf16, s);
        case 2: return launch_final_t<2>(src0, C0, src1, C1, wpk, bias, logits, mask, n, H, W, bf16, s);
        case 3: return launch_final_t<3>(src0, C0, src1, C1, wpk, bias, logits, mask, n, H, W, bf16, s);
        case 4: return launch_final_t<4>(src0, C0, src1, C1, wpk, bias, logits, mask, n, H, W, bf16, s);
        case 5: return launch_final_t<5>(src0, C0, src1, C1, wpk, bias, logits, mask, n, H, W, bf16, s);
        case 6: return launch_final_t<6>(src0, C0, src1, C1, wpk, bias, logits, mask, n, H, W, bf16, s);
        case 7: return launch_final_t<7>(src0, C0, src1, C1, wpk, bias, logits, mask, n, H, W, bf16, s);
        case 8: return launch_final_t<8>(src0, C0, src1, C1, wpk, bias, logits, mask, n, H, W, bf16, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_seg_eval(const float* logits, const int8_t* labels, int n, int classes, int H, int W,
                           unsigned long long* confusion, unsigned long long* loss_fixed, hipStream_t s) {
    const int HW = H * W;
    const dim3 grid(std::min((HW + 255) / 256, 1024), n);
#define GSA_EVAL(K) \
    if (classes == K) { hipLaunchKernelGGL(seg_eval_kernel<K>, grid, dim3(256), 0, s, logits, labels, HW, confusion, loss_fixed); return hipGetLastError(); }
    GSA_EVAL(2) GSA_EVAL(3) GSA_EVAL(4) GSA_EVAL(5) GSA_EVAL(6) GSA_EVAL(7) GSA_EVAL(8)
#undef GSA_EVAL
    return hipErrorInvalidValue;
}

hipError_t launch_fill_normal(float* out, int per_sample, int n, unsigned long long first_index, unsigned plane,
                              unsigned long long seed, hipStream_t s) {
    if (per_sample % 4 || per_sample <= 0 || n <= 0) return hipErrorInvalidValue;
    const long total = (long)n * (per_sample / 4);
    const int grid = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(fill_normal_kernel, dim3(grid), dim3(256), 0, s, out, per_sample, n, first_index, plane, seed);
    return hipGetLastError();
}

}  // namespace gsa
