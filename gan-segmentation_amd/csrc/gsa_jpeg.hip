// gsa_jpeg.hip -- baseline JPEG encoder on the GPU for the dataset writer (include/gsa_jpeg.h).
//
// Replaces cv2.imwrite("img_%06d.jpg") of reference main.py:100-101 (libjpeg defaults: quality 95, 4:2:0, integer
// "islow" DCT, Annex K Huffman tables) with the same integer arithmetic on the image gsa_generate left in HBM.
// HBM-bound byte/integer work on the vector ALU -- nothing here is GEMM-shaped:
//   jpeg_transform_kernel  one wave per 16x16-px MCU: 16-byte loads of the RGB rows into LDS, colour conversion +
//                          2x2 chroma box filter in registers (row partner by wave shuffle), the two 8-point DCT
//                          passes through LDS (one row / column per lane), quantisation, 16-byte coalesced stores of
//                          the six zigzag-ordered coefficient blocks            (3 B/px read, 3 B/px written)
//   jpeg_entropy_wave_kernel  one wave per restart interval, a lane per coefficient: zero runs from the ballot of
//                          non-zeros, per-lane bit strings placed by a wave scan into an LDS bit buffer, byte stuffing
//                          into a private scratch segment; DC prediction restarts with the interval, which is what
//                          makes the waves independent (jpeg_entropy_kernel: one lane per interval, for long intervals)
//   jpeg_offsets_kernel    per image: exclusive scan of the segment lengths, EOI marker, total length
//   jpeg_gather_kernel     one wave per segment: scratch -> its final position in the image's scan
// Bit-exact contract: oracle/c/jpeg_oracle.c (itself byte-identical to libjpeg-turbo via Pillow).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/gsa.h"
#include "../../include/gsa_jpeg.h"

namespace {

constexpr int kBlockCap = 448;   // bytes one 8x8 block can need at most: 64 x (16-bit code + 11 value bits), all bytes stuffed

constexpr uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ITU-T T.81 Annex K.1 quantisation tables (natural order) and K.3 Huffman tables (counts per code length, symbols)
constexpr uint8_t kBaseLuma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                   14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                   18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
constexpr uint8_t kBaseChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                     99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
constexpr uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
constexpr uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
constexpr uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
constexpr uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
constexpr uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
constexpr uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
constexpr uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

// Canonical Huffman codes (T.81 Annex C) built at compile time: entry = (code << 5) | length
struct Huff { uint32_t e[256]; };
constexpr Huff make_huff(const uint8_t* bits, const uint8_t* vals) {
    Huff h{};
    unsigned code = 0;
    int k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < bits[len - 1]; ++i, ++k) h.e[vals[k]] = (code++ << 5) | (unsigned)len;
        code <<= 1;
    }
    return h;
}
struct ZigInv { uint8_t p[64]; };
constexpr ZigInv make_ziginv() {
    ZigInv z{};
    for (int k = 0; k < 64; ++k) z.p[kZigzag[k]] = (uint8_t)k;
    return z;
}

__device__ const Huff kHuffDev[4] = {make_huff(kDcLumaBits, kDcVals), make_huff(kAcLumaBits, kAcLumaVals),
                                     make_huff(kDcChromaBits, kDcVals), make_huff(kAcChromaBits, kAcChromaVals)};
__device__ const ZigInv kZigInvDev = make_ziginv();   // natural index -> zigzag position

struct QDiv { uint16_t d[2][64]; };   // quantiser step << 3 (the DCT output carries a factor 8), natural order

// IJG quality scaling: q < 50 -> 5000/q, else 200 - 2q; table = clamp((base*scale + 50)/100, 1, 255)
void quant_table(const uint8_t* base, int quality, uint8_t* out) {
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        const long v = ((long)base[i] * scale + 50) / 100;
        out[i] = (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
    }
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// One 8-point pass of the Loeffler-Ligtenberg-Moschytz DCT in 13-bit fixed point (constants = round(x * 2^13)).
// PASS 0 (rows) leaves 2 extra fraction bits, PASS 1 (columns) removes them: the result is 8 x the true DCT.
template <int PASS>
__device__ __forceinline__ void dct8(int (&v)[8]) {
    const int t0 = v[0] + v[7], t7 = v[0] - v[7], t1 = v[1] + v[6], t6 = v[1] - v[6];
    const int t2 = v[2] + v[5], t5 = v[2] - v[5], t3 = v[3] + v[4], t4 = v[3] - v[4];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int SH = PASS == 0 ? 13 - 2 : 13 + 2;
    if (PASS == 0) { v[0] = (t10 + t11) << 2; v[4] = (t10 - t11) << 2; }
    else { v[0] = descale(t10 + t11, 2); v[4] = descale(t10 - t11, 2); }
    int z1 = (t12 + t13) * 4433;
    v[2] = descale(z1 + t13 * 6270, SH);
    v[6] = descale(z1 - t12 * 15137, SH);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * 9633;
    const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
    z1 *= -7373; z2 *= -20995;
    z3 = z3 * -16069 + z5;
    z4 = z4 * -3196 + z5;
    v[7] = descale(a4 + z1 + z3, SH);
    v[5] = descale(a5 + z2 + z4, SH);
    v[3] = descale(a6 + z2 + z3, SH);
    v[1] = descale(a7 + z1 + z4, SH);
}

// coef: [image][MCU][6 blocks: Y00 Y01 Y10 Y11 Cb Cr][64 zigzag] int16
__global__ __launch_bounds__(256) void jpeg_transform_kernel(const uint8_t* __restrict__ rgb, int H, int W, int mcus_x,
                                                             int mcus_per_img, int total_mcus, QDiv q,
                                                             int16_t* __restrict__ coef) {
    __shared__ __attribute__((aligned(16))) uint8_t raw[4][768];
    __shared__ int comp[4][6][64];
    __shared__ __attribute__((aligned(16))) int16_t zz[4][384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int base = blockIdx.x * 4; base < total_mcus; base += gridDim.x * 4) {   // trip count uniform over the block
        const int m = base + wave;
        const bool active = m < total_mcus;
        const int img = active ? m / mcus_per_img : 0;
        const int mi = active ? m - img * mcus_per_img : 0;
        const int my = mi / mcus_x, mx = mi - my * mcus_x;
        if (active && lane < 48) {   // 16 rows x 48 bytes, 16 bytes per lane
            const int row = lane / 3, part = lane - row * 3;
            const uint8_t* src = rgb + ((size_t)((size_t)img * H + my * 16 + row) * W + mx * 16) * 3 + part * 16;
            *reinterpret_cast<uint4*>(&raw[wave][row * 48 + part * 16]) = *reinterpret_cast<const uint4*>(src);
        }
        __syncthreads();
        if (active) {   // a lane converts 4 neighbouring pixels of one row
            const int y = lane >> 2, x0 = (lane & 3) * 4;
            const uint8_t* px = &raw[wave][(y * 16 + x0) * 3];
            int cb[4], cr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = px[3 * j], g = px[3 * j + 1], b = px[3 * j + 2];
                // 16-bit fixed point: Y = .299R+.587G+.114B; Cb = -.16874R-.33126G+.5B+128; Cr = .5R-.41869G-.08131B+128
                const int Y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
                cb[j] = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
                cr[j] = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
                comp[wave][(y >> 3) * 2 + (x0 >> 3)][(y & 7) * 8 + ((x0 + j) & 7)] = Y - 128;
            }
            // 2x2 box filter: horizontal pairs here, the row below sits 4 lanes up; rounding bias alternates 1, 2
            const int b01 = cb[0] + cb[1], b23 = cb[2] + cb[3], r01 = cr[0] + cr[1], r23 = cr[2] + cr[3];
            const int ob01 = __shfl_down(b01, 4), ob23 = __shfl_down(b23, 4), or01 = __shfl_down(r01, 4), or23 = __shfl_down(r23, 4);
            if (!(y & 1)) {
                const int o = (y >> 1) * 8 + (lane & 3) * 2;
                comp[wave][4][o] = ((b01 + ob01 + 1) >> 2) - 128;
                comp[wave][4][o + 1] = ((b23 + ob23 + 2) >> 2) - 128;
                comp[wave][5][o] = ((r01 + or01 + 1) >> 2) - 128;
                comp[wave][5][o + 1] = ((r23 + or23 + 2) >> 2) - 128;
            }
        }
        __syncthreads();
        const int blk = lane >> 3, rc = lane & 7;
        if (active && lane < 48) {   // rows
            int v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = comp[wave][blk][rc * 8 + i];
            dct8<0>(v);
#pragma unroll
            for (int i = 0; i < 8; ++i) comp[wave][blk][rc * 8 + i] = v[i];
        }
        __syncthreads();
        if (active && lane < 48) {   // columns, quantisation (round half away from zero), zigzag
            int v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = comp[wave][blk][i * 8 + rc];
            dct8<1>(v);
            const uint16_t* d = q.d[blk < 4 ? 0 : 1];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int nat = i * 8 + rc;
                const unsigned dv = d[nat];
                const unsigned a = (unsigned)(v[i] < 0 ? -v[i] : v[i]);
                const int qv = (int)((a + (dv >> 1)) / dv);
                zz[wave][blk * 64 + kZigInvDev.p[nat]] = (int16_t)(v[i] < 0 ? -qv : qv);
            }
        }
        __syncthreads();
        if (active && lane < 48)
            reinterpret_cast<uint4*>(coef + (size_t)m * 384)[lane] = reinterpret_cast<const uint4*>(&zz[wave][0])[lane];
    }
}

struct BitSink {
    uint8_t* p;
    int n;
    unsigned long long acc;
    int nbits;
    __device__ __forceinline__ void put(unsigned code, int size) {
        acc = (acc << size) | code;
        nbits += size;
        while (nbits >= 8) {
            const unsigned c = (unsigned)(acc >> (nbits - 8)) & 255u;
            p[n++] = (uint8_t)c;
            if (c == 255u) p[n++] = 0;   // byte stuffing
            nbits -= 8;
        }
    }
};

// One 8x8 block: 64 zigzag-ordered coefficients held in registers (8 x 16 bytes).
__device__ __forceinline__ void encode_block(BitSink& o, const uint4 (&u)[8], int& pred, const uint32_t* dc, const uint32_t* ac) {
    int run = 0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const uint32_t w[4] = {u[ch].x, u[ch].y, u[ch].z, u[ch].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int v = (int)(int16_t)(w[j >> 1] >> ((j & 1) * 16));
            if (ch == 0 && j == 0) {                     // DC: difference to the previous block of the component
                const int diff = v - pred;
                pred = v;
                const int a = diff < 0 ? -diff : diff;
                const int nb = 32 - __clz(a);
                const uint32_t e = dc[nb];
                const unsigned vb = (unsigned)(diff < 0 ? diff - 1 : diff) & ((1u << nb) - 1u);
                o.put(((e >> 5) << nb) | vb, (int)(e & 31u) + nb);
            } else if (v == 0) {
                ++run;
            } else {
                while (run > 15) { const uint32_t z = ac[0xF0]; o.put(z >> 5, (int)(z & 31u)); run -= 16; }
                const int a = v < 0 ? -v : v;
                const int nb = 32 - __clz(a);
                const uint32_t e = ac[(run << 4) + nb];
                const unsigned vb = (unsigned)(v < 0 ? v - 1 : v) & ((1u << nb) - 1u);
                o.put(((e >> 5) << nb) | vb, (int)(e & 31u) + nb);
                run = 0;
            }
        }
    }
    if (run) { const uint32_t e = ac[0]; o.put(e >> 5, (int)(e & 31u)); }   // end of block
}

// One lane per restart interval (`restart` MCUs): sequential Huffman coding of its blocks into scratch + t*segcap.
// The lane is latency-bound (a wave per CU at most): the 128 bytes of a block are fetched with eight independent
// 16-byte loads, the next block's while the current one is coded, and the code tables sit in LDS.
__global__ __launch_bounds__(64) void jpeg_entropy_kernel(const int16_t* __restrict__ coef, int mcus_per_img, int restart,
                                                          int segs_per_img, int total_segs, int segcap,
                                                          uint8_t* __restrict__ scratch, int* __restrict__ seglen) {
    __shared__ uint32_t huff[4][256];
    for (int i = threadIdx.x; i < 1024; i += 64) huff[i >> 8][i & 255] = kHuffDev[i >> 8].e[i & 255];
    __syncthreads();
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= total_segs) return;
    const int img = t / segs_per_img, s = t - img * segs_per_img;
    const int m0 = s * restart, m1 = min(m0 + restart, mcus_per_img);
    BitSink o{scratch + (size_t)t * segcap, 0, 0ull, 0};
    int pred[3] = {0, 0, 0};
    const uint4* c = reinterpret_cast<const uint4*>(coef + ((size_t)img * mcus_per_img + m0) * 384);   // 8 x uint4 per block
    const int nblocks = (m1 - m0) * 6;
    uint4 cur[8], nxt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cur[k] = c[k];
    for (int bi = 0; bi < nblocks; ++bi) {
        const int pf = bi + 1 < nblocks ? bi + 1 : bi;          // unconditional (clamped) prefetch of the next block
#pragma unroll
        for (int k = 0; k < 8; ++k) nxt[k] = c[pf * 8 + k];
        const int b = bi % 6;
        if (b < 4) encode_block(o, cur, pred[0], huff[0], huff[1]);
        else encode_block(o, cur, pred[b - 3], huff[2], huff[3]);
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
    }
    if (o.nbits) o.put((1u << (8 - o.nbits)) - 1u, 8 - o.nbits);   // pad the last byte with ones
    if (s != segs_per_img - 1) {                                    // RSTm in front of the next interval
        o.p[o.n++] = 0xFF;
        o.p[o.n++] = (uint8_t)(0xD0 + (s & 7));
    }
    seglen[t] = o.n;
}

// The same coding with one WAVE per restart interval (used for intervals of up to kWaveRestartMax MCUs): a lane per
// coefficient of a block.  The zero run in front of a non-zero coefficient comes from the ballot of non-zeros, every lane
// builds its own bit string ([ZRL..] code value-bits, lane 0 the DC difference, the last non-zero lane also the EOB), a
// wave scan of the lengths gives its bit offset, and the strings are OR-ed into a zeroed LDS bit buffer.  Byte stuffing
// is a second pass over that buffer: 64 bytes at a time, the 0xFF bytes in front of a lane counted by ballot + popcount.
constexpr int kWaveRestartMax = 8;
constexpr int kRawWordsPerBlock = 54;        // 64 coefficients x 27 bits at most = 216 bytes, before stuffing

__device__ __forceinline__ void deposit(unsigned* buf, int o, unsigned long long str, int len) {   // MSB first at bit offset o
    const int w = o >> 5, shift = 64 - (o & 31) - len;          // the 64-bit window over words w, w+1
    unsigned long long hi;
    unsigned lo = 0;
    if (shift >= 0) hi = str << shift;
    else { hi = str >> (-shift); lo = (unsigned)(str << (32 + shift)); }
    const unsigned h1 = (unsigned)(hi >> 32), h0 = (unsigned)hi;
    if (h1) atomicOr(&buf[w], h1);
    if (h0) atomicOr(&buf[w + 1], h0);
    if (lo) atomicOr(&buf[w + 2], lo);
}

__global__ __launch_bounds__(256) void jpeg_entropy_wave_kernel(const int16_t* __restrict__ coef, int mcus_per_img, int restart,
                                                                int segs_per_img, int total_segs, int segcap, int words_per_wave,
                                                                uint8_t* __restrict__ scratch, int* __restrict__ seglen) {
    extern __shared__ unsigned raw[];
    __shared__ uint32_t huff[4][256];
    for (int i = threadIdx.x; i < 1024; i += 256) huff[i >> 8][i & 255] = kHuffDev[i >> 8].e[i & 255];
    __syncthreads();                                   // the only block-wide barrier: waves are independent from here on
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= total_segs) return;
    unsigned* buf = raw + wave * words_per_wave;
    for (int i = lane; i < words_per_wave; i += 64) buf[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int img = t / segs_per_img, s = t - img * segs_per_img;
    const int m0 = s * restart, m1 = min(m0 + restart, mcus_per_img);
    const int16_t* c = coef + ((size_t)img * mcus_per_img + m0) * 384;
    const int nblocks = (m1 - m0) * 6;
    int bitpos = 0, pred[3] = {0, 0, 0};
    short v = c[lane];
    for (int bi = 0; bi < nblocks; ++bi) {
        const short vnext = c[(bi + 1 < nblocks ? bi + 1 : bi) * 64 + lane];      // prefetch the next block
        const int b = bi % 6, ci = b < 4 ? 0 : b - 3;
        const uint32_t* dc = huff[b < 4 ? 0 : 2];
        const uint32_t* ac = huff[b < 4 ? 1 : 3];
        const int dcv = __builtin_amdgcn_readlane((int)v, 0);
        const int diff = dcv - pred[ci];
        pred[ci] = dcv;
        const unsigned long long nzmask = __ballot(v != 0) & ~1ull;              // non-zero AC coefficients
        const int val = lane == 0 ? diff : (int)v;
        const int a = val < 0 ? -val : val;
        const int nb = 32 - __clz(a);
        const unsigned vb = (unsigned)(val < 0 ? val - 1 : val) & ((1u << nb) - 1u);
        unsigned long long str = 0;
        int len = 0;
        if (lane == 0) {
            const uint32_t e = dc[nb];
            str = ((unsigned long long)(e >> 5) << nb) | vb;
            len = (int)(e & 31u) + nb;
        } else if (v != 0) {
            const unsigned long long lower = nzmask & ((1ull << lane) - 1ull);
            const int prevpos = lower ? 63 - __clzll((long long)lower) : 0;
            const int run = lane - prevpos - 1;
            const uint32_t z = ac[0xF0], e = ac[((run & 15) << 4) + nb];
            for (int k = run >> 4; k > 0; --k) { str = (str << (z & 31u)) | (z >> 5); len += (int)(z & 31u); }
            const int cl = (int)(e & 31u) + nb;
            str = (str << cl) | ((unsigned long long)(e >> 5) << nb) | vb;
            len += cl;
        }
        const int lastnz = nzmask ? 63 - __clzll((long long)nzmask) : 0;         // uniform; 0 = no AC coefficient at all
        if (lastnz != 63 && lane == lastnz) {                                      // end of block
            const uint32_t e = ac[0];
            str = (str << (e & 31u)) | (e >> 5);
            len += (int)(e & 31u);
        }
        int incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (len) deposit(buf, bitpos + incl - len, str, len);
        bitpos += __builtin_amdgcn_readlane(incl, 63);
        v = vnext;
    }
    if ((bitpos & 7) && lane == 0) {                                               // pad the last byte with ones
        const int pad = 8 - (bitpos & 7);
        deposit(buf, bitpos, (1ull << pad) - 1ull, pad);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int nbytes = (bitpos + 7) >> 3;
    uint8_t* dst = scratch + (size_t)t * segcap;
    int outpos = 0;
    for (int j0 = 0; j0 < nbytes; j0 += 64) {                                      // byte stuffing: 0xFF -> 0xFF 0x00
        const int j = j0 + lane;
        const bool valid = j < nbytes;
        const unsigned byte = valid ? (buf[j >> 2] >> (24 - 8 * (j & 3))) & 255u : 0u;
        const bool ff = valid && byte == 255u;
        const unsigned long long m = __ballot(ff);
        if (valid) {
            const int pos = outpos + lane + __popcll(m & ((1ull << lane) - 1ull));
            dst[pos] = (uint8_t)byte;
            if (ff) dst[pos + 1] = 0;
        }
        outpos += min(64, nbytes - j0) + __popcll(m);
    }
    if (lane == 0) {
        if (s != segs_per_img - 1) {                                               // RSTm in front of the next interval
            dst[outpos] = 0xFF;
            dst[outpos + 1] = (uint8_t)(0xD0 + (s & 7));
            outpos += 2;
        }
        seglen[t] = outpos;
    }
}

// One block per image: exclusive scan of its segment lengths; EOI marker and the total (negative when it does not fit).
__global__ __launch_bounds__(256) void jpeg_offsets_kernel(const int* __restrict__ seglen, int segs_per_img,
                                                           int* __restrict__ segoff, int* __restrict__ lengths,
                                                           uint8_t* __restrict__ out, long long out_stride) {
    __shared__ int wsum[4];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;                                      // same value in every thread
    for (int base = 0; base < segs_per_img; base += 256) {
        const int i = base + tid;
        const int v = i < segs_per_img ? seglen[(size_t)img * segs_per_img + i] : 0;
        int incl = v;                                   // inclusive scan inside the wave by shuffles
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { before += w < wave ? wsum[w] : 0; total += wsum[w]; }
        if (i < segs_per_img) segoff[(size_t)img * segs_per_img + i] = carry + before + incl - v;
        carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        const long long need = (long long)carry + 2;
        if (need <= out_stride) {
            out[(size_t)img * out_stride + carry] = 0xFF;
            out[(size_t)img * out_stride + carry + 1] = 0xD9;
            lengths[img] = (int)need;
        } else {
            lengths[img] = (int)-need;
        }
    }
}

// One wave per segment: scratch -> final position.
__global__ __launch_bounds__(256) void jpeg_gather_kernel(const uint8_t* __restrict__ scratch, const int* __restrict__ seglen,
                                                          const int* __restrict__ segoff, int segs_per_img, int total_segs,
                                                          int segcap, uint8_t* __restrict__ out, long long out_stride) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= total_segs) return;
    const int img = t / segs_per_img;
    const long long off = segoff[t];
    long long len = seglen[t];
    if (off + len > out_stride) len = out_stride > off ? out_stride - off : 0;
    const uint8_t* src = scratch + (size_t)t * segcap;
    uint8_t* dst = out + (size_t)img * out_stride + off;
    for (int i = lane; i < (int)len; i += 64) dst[i] = src[i];
}

struct Geometry {
    int mcus_x, mcus_per_img, segs_per_img, segcap;
    size_t coef_bytes, scratch_bytes, seg_ints;
};

bool geometry(int n, int H, int W, int restart, Geometry* g) {
    if (n < 1 || H < 16 || W < 16 || H % 16 || W % 16 || H > 65535 || W > 65535 || restart < 1 || restart > 65535) return false;
    g->mcus_x = W / 16;
    g->mcus_per_img = (H / 16) * (W / 16);
    g->segs_per_img = (g->mcus_per_img + restart - 1) / restart;
    g->segcap = (restart * 6 * kBlockCap + 16 + 15) & ~15;
    g->coef_bytes = (size_t)n * g->mcus_per_img * 768;
    g->scratch_bytes = (size_t)n * g->segs_per_img * g->segcap;
    g->seg_ints = (size_t)n * g->segs_per_img;
    return true;
}

struct HostSink {
    uint8_t* p;
    int64_t n, cap;
    void put(int b) { if (n < cap) p[n] = (uint8_t)b; ++n; }
    void put16(int v) { put(v >> 8); put(v & 255); }
    void dht(int id, const uint8_t* bits, const uint8_t* vals) {
        int cnt = 0;
        for (int i = 0; i < 16; ++i) cnt += bits[i];
        put16(0xFFC4); put16(2 + 1 + 16 + cnt); put(id);
        for (int i = 0; i < 16; ++i) put(bits[i]);
        for (int i = 0; i < cnt; ++i) put(vals[i]);
    }
};

}  // namespace

extern "C" {

int64_t gsa_jpeg_header(int32_t H, int32_t W, int32_t quality, int32_t restart, uint8_t* host_buf, int64_t cap) {
    if (H < 1 || W < 1 || H > 65535 || W > 65535 || restart < 0 || restart > 65535 || (!host_buf && cap > 0)) return GSA_ERR_INVALID;
    uint8_t ql[64], qc[64];
    quant_table(kBaseLuma, quality, ql);
    quant_table(kBaseChroma, quality, qc);
    HostSink s{host_buf, 0, cap};
    s.put16(0xFFD8);                                                      // SOI
    s.put16(0xFFE0); s.put16(16);                                         // APP0: JFIF 1.01, aspect ratio 1:1, no thumbnail
    for (const char* c = "JFIF"; *c; ++c) s.put(*c);
    s.put(0); s.put(1); s.put(1); s.put(0); s.put16(1); s.put16(1); s.put(0); s.put(0);
    for (int t = 0; t < 2; ++t) {                                         // DQT, zigzag order
        s.put16(0xFFDB); s.put16(67); s.put(t);
        for (int i = 0; i < 64; ++i) s.put((t ? qc : ql)[kZigzag[i]]);
    }
    s.put16(0xFFC0); s.put16(17); s.put(8); s.put16(H); s.put16(W); s.put(3);   // SOF0: Y 2x2, Cb 1x1, Cr 1x1
    s.put(1); s.put(0x22); s.put(0);
    s.put(2); s.put(0x11); s.put(1);
    s.put(3); s.put(0x11); s.put(1);
    s.dht(0x00, kDcLumaBits, kDcVals);
    s.dht(0x10, kAcLumaBits, kAcLumaVals);
    s.dht(0x01, kDcChromaBits, kDcVals);
    s.dht(0x11, kAcChromaBits, kAcChromaVals);
    if (restart > 0) { s.put16(0xFFDD); s.put16(4); s.put16(restart); }  // DRI
    s.put16(0xFFDA); s.put16(12); s.put(3);                               // SOS
    s.put(1); s.put(0x00); s.put(2); s.put(0x11); s.put(3); s.put(0x11);
    s.put(0); s.put(63); s.put(0);
    return s.n;
}

int64_t gsa_jpeg_workspace_bytes(int32_t n, int32_t H, int32_t W, int32_t restart) {
    Geometry g;
    if (!geometry(n, H, W, restart, &g)) return GSA_ERR_INVALID;
    return (int64_t)(g.coef_bytes + g.scratch_bytes + 2 * g.seg_ints * sizeof(int) + 64);
}

int64_t gsa_jpeg_max_scan_bytes(int32_t H, int32_t W, int32_t restart) {
    Geometry g;
    if (!geometry(1, H, W, restart, &g)) return GSA_ERR_INVALID;
    return (int64_t)g.mcus_per_img * 6 * kBlockCap + (int64_t)g.segs_per_img * 3 + 2;
}

int gsa_jpeg_encode(void* stream, int32_t n, int32_t H, int32_t W, const uint8_t* rgb, int32_t quality, int32_t restart,
                    void* workspace, int64_t workspace_bytes, uint8_t* out, int64_t out_stride, int32_t* lengths) {
    Geometry g;
    if (!geometry(n, H, W, restart, &g)) return GSA_ERR_INVALID;
    if (!rgb || !workspace || !out || !lengths || out_stride < 2 || out_stride > 0x7fffffffll) return GSA_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(rgb) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return GSA_ERR_INVALID;
    if (workspace_bytes < gsa_jpeg_workspace_bytes(n, H, W, restart)) return GSA_ERR_INVALID;
    if ((int64_t)n * g.mcus_per_img > 0x7fffffffll / 4) return GSA_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    uint8_t ql[64], qc[64];
    quant_table(kBaseLuma, quality, ql);
    quant_table(kBaseChroma, quality, qc);
    QDiv q;
    for (int i = 0; i < 64; ++i) { q.d[0][i] = (uint16_t)(ql[i] << 3); q.d[1][i] = (uint16_t)(qc[i] << 3); }
    uint8_t* w = static_cast<uint8_t*>(workspace);
    int16_t* coef = reinterpret_cast<int16_t*>(w);
    uint8_t* scratch = w + g.coef_bytes;
    int* seglen = reinterpret_cast<int*>(w + g.coef_bytes + ((g.scratch_bytes + 15) & ~(size_t)15));
    int* segoff = seglen + g.seg_ints;
    const int total_mcus = n * g.mcus_per_img, total_segs = n * g.segs_per_img;
    const int tgrid = (total_mcus + 3) / 4 < 8192 ? (total_mcus + 3) / 4 : 8192;
    hipLaunchKernelGGL(jpeg_transform_kernel, dim3(tgrid), dim3(256), 0, s, rgb, H, W, g.mcus_x, g.mcus_per_img, total_mcus, q, coef);
    if (restart <= kWaveRestartMax) {
        const int words = restart * 6 * kRawWordsPerBlock + 4;
        hipLaunchKernelGGL(jpeg_entropy_wave_kernel, dim3((total_segs + 3) / 4), dim3(256), 4 * words * sizeof(unsigned), s, coef,
                           g.mcus_per_img, restart, g.segs_per_img, total_segs, g.segcap, words, scratch, seglen);
    } else {
        hipLaunchKernelGGL(jpeg_entropy_kernel, dim3((total_segs + 63) / 64), dim3(64), 0, s, coef, g.mcus_per_img, restart,
                           g.segs_per_img, total_segs, g.segcap, scratch, seglen);
    }
    hipLaunchKernelGGL(jpeg_offsets_kernel, dim3(n), dim3(256), 0, s, seglen, g.segs_per_img, segoff, lengths, out,
                       (long long)out_stride);
    hipLaunchKernelGGL(jpeg_gather_kernel, dim3((total_segs + 3) / 4), dim3(256), 0, s, scratch, seglen, segoff,
                       g.segs_per_img, total_segs, g.segcap, out, (long long)out_stride);
    return hipGetLastError() == hipSuccess ? GSA_OK : GSA_ERR_HIP;
}

}  // extern "C"
