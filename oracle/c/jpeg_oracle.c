/*
 * jpeg_oracle.c -- TEST INFRASTRUCTURE ONLY: scalar restatement of the baseline JPEG encoder the dataset writer's GPU
 * path implements (include/gsa_jpeg.h).
 *
 * The reference writes its images with cv2.imwrite(".jpg") (reference main.py:100-101), i.e. libjpeg(-turbo) at
 * its defaults: quality 95, YCbCr 4:2:0, the "islow" integer DCT, the Annex K Huffman tables.  That encoder lives in a
 * third-party dependency (OpenCV 4.0.0.21 -> libjpeg-turbo; requirements.txt), absent from /root/reference.  This file
 * restates the published algorithm (ITU-T T.81 + the integer arithmetic of the IJG encoder: 16-bit fixed-point colour
 * conversion, 2x2 box downsampling with alternating rounding bias, the Loeffler-Ligtenberg-Moschytz 13-bit integer
 * DCT, round-half-away quantisation) one sample at a time.  It is PINNED: tests/test_jpeg.py compares its output
 * byte for byte with Pillow's libjpeg-turbo (the same library cv2 links) for the same quality and restart interval.
 *
 * Only tests/ may load this library; the product path is csrc/gsa_jpeg.hip and has no CPU fallback.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* T.81 Annex K.1 example quantisation tables (natural order) */
static const uint8_t kBaseLuma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                      14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                      18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                      49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kBaseChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                        99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                        99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* T.81 Annex K.3 typical Huffman tables: code counts per length 1..16, then the symbols */
static const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct { uint16_t code[256]; uint8_t size[256]; } huff;

/* T.81 Annex C: canonical codes from (bits, vals) */
static void build_huff(const uint8_t* bits, const uint8_t* vals, huff* h) {
    memset(h, 0, sizeof *h);
    unsigned code = 0;
    int k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < bits[len - 1]; ++i, ++k) {
            h->code[vals[k]] = (uint16_t)code++;
            h->size[vals[k]] = (uint8_t)len;
        }
        code <<= 1;
    }
}

/* IJG quality scaling: q < 50 -> 5000/q, else 200 - 2q; table = clamp((base*scale + 50)/100, 1, 255) */
static void quant_table(const uint8_t* base, int quality, uint8_t* out) {
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        long v = ((long)base[i] * scale + 50) / 100;
        out[i] = (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
    }
}

typedef struct { uint8_t* p; size_t n, cap; } sink;
static void put(sink* s, int b) { if (s->n < s->cap) s->p[s->n] = (uint8_t)b; s->n++; }
static void put16(sink* s, int v) { put(s, v >> 8); put(s, v & 255); }

static void emit_dht(sink* s, int tc_th, const uint8_t* bits, const uint8_t* vals) {
    int n = 0;
    for (int i = 0; i < 16; ++i) n += bits[i];
    put16(s, 0xFFC4); put16(s, 2 + 1 + 16 + n); put(s, tc_th);
    for (int i = 0; i < 16; ++i) put(s, bits[i]);
    for (int i = 0; i < n; ++i) put(s, vals[i]);
}

/* Everything in front of the entropy-coded data: SOI, JFIF APP0, DQT x2, SOF0, DHT x4, [DRI], SOS */
static void write_header(sink* s, int H, int W, const uint8_t* ql, const uint8_t* qc, int restart) {
    put16(s, 0xFFD8);
    put16(s, 0xFFE0); put16(s, 16); put(s, 'J'); put(s, 'F'); put(s, 'I'); put(s, 'F'); put(s, 0);
    put(s, 1); put(s, 1); put(s, 0); put16(s, 1); put16(s, 1); put(s, 0); put(s, 0);
    for (int t = 0; t < 2; ++t) {
        put16(s, 0xFFDB); put16(s, 67); put(s, t);
        for (int i = 0; i < 64; ++i) put(s, (t ? qc : ql)[kZigzag[i]]);
    }
    put16(s, 0xFFC0); put16(s, 17); put(s, 8); put16(s, H); put16(s, W); put(s, 3);
    put(s, 1); put(s, 0x22); put(s, 0);
    put(s, 2); put(s, 0x11); put(s, 1);
    put(s, 3); put(s, 0x11); put(s, 1);
    emit_dht(s, 0x00, kDcLumaBits, kDcVals);
    emit_dht(s, 0x10, kAcLumaBits, kAcLumaVals);
    emit_dht(s, 0x01, kDcChromaBits, kDcVals);
    emit_dht(s, 0x11, kAcChromaBits, kAcChromaVals);
    if (restart > 0) { put16(s, 0xFFDD); put16(s, 4); put16(s, restart); }
    put16(s, 0xFFDA); put16(s, 12); put(s, 3);
    put(s, 1); put(s, 0x00); put(s, 2); put(s, 0x11); put(s, 3); put(s, 0x11);
    put(s, 0); put(s, 63); put(s, 0);
}

/* 13-bit fixed-point constants of the LLM 8-point DCT: round(x * 2^13) */
#define C_0_298631336 2446
#define C_0_390180644 3196
#define C_0_541196100 4433
#define C_0_765366865 6270
#define C_0_899976223 7373
#define C_1_175875602 9633
#define C_1_501321110 12299
#define C_1_847759065 15137
#define C_1_961570560 16069
#define C_2_053119869 16819
#define C_2_562915447 20995
#define C_3_072711026 25172

static inline int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

/* One 8-point pass; sh0 = shift of outputs 0/4 (left shift when negative), sh = right shift of the others */
static void dct8(const int32_t* in, int stride, int32_t* out, int pass) {
    const int32_t t0 = in[0] + in[7 * stride], t7 = in[0] - in[7 * stride];
    const int32_t t1 = in[stride] + in[6 * stride], t6 = in[stride] - in[6 * stride];
    const int32_t t2 = in[2 * stride] + in[5 * stride], t5 = in[2 * stride] - in[5 * stride];
    const int32_t t3 = in[3 * stride] + in[4 * stride], t4 = in[3 * stride] - in[4 * stride];
    const int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    const int sh = pass == 0 ? 13 - 2 : 13 + 2;
    if (pass == 0) { out[0] = (t10 + t11) << 2; out[4 * stride] = (t10 - t11) << 2; }
    else { out[0] = descale(t10 + t11, 2); out[4 * stride] = descale(t10 - t11, 2); }
    int32_t z1 = (t12 + t13) * C_0_541196100;
    out[2 * stride] = descale(z1 + t13 * C_0_765366865, sh);
    out[6 * stride] = descale(z1 - t12 * C_1_847759065, sh);
    z1 = t4 + t7;
    int32_t z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int32_t z5 = (z3 + z4) * C_1_175875602;
    const int32_t a4 = t4 * C_0_298631336, a5 = t5 * C_2_053119869, a6 = t6 * C_3_072711026, a7 = t7 * C_1_501321110;
    z1 *= -C_0_899976223; z2 *= -C_2_562915447;
    z3 = z3 * -C_1_961570560 + z5;
    z4 = z4 * -C_0_390180644 + z5;
    out[7 * stride] = descale(a4 + z1 + z3, sh);
    out[5 * stride] = descale(a5 + z2 + z4, sh);
    out[3 * stride] = descale(a6 + z2 + z3, sh);
    out[1 * stride] = descale(a7 + z1 + z4, sh);
}

/* samples (already level-shifted) -> quantised coefficients in zigzag order */
static void fdct_quant(const int32_t* blk, const uint8_t* q, int16_t* zz) {
    int32_t ws[64], co[64];
    for (int r = 0; r < 8; ++r) dct8(blk + 8 * r, 1, ws + 8 * r, 0);
    for (int c = 0; c < 8; ++c) dct8(ws + c, 8, co + c, 1);
    for (int k = 0; k < 64; ++k) {
        const int i = kZigzag[k];
        const int32_t d = (int32_t)q[i] << 3;            /* the DCT output carries a factor 8 */
        int32_t v = co[i];
        v = v < 0 ? -((-v + (d >> 1)) / d) : (v + (d >> 1)) / d;
        zz[k] = (int16_t)v;
    }
}

typedef struct { sink* s; uint32_t acc; int nbits; } bitw;
static void bits_put(bitw* b, unsigned code, int size) {
    if (!size) return;
    b->acc = (b->acc << size) | (code & ((1u << size) - 1));
    b->nbits += size;
    while (b->nbits >= 8) {
        const int c = (b->acc >> (b->nbits - 8)) & 255;
        put(b->s, c);
        if (c == 255) put(b->s, 0);
        b->nbits -= 8;
    }
}
static void bits_flush(bitw* b) { if (b->nbits) bits_put(b, 0x7F, 8 - b->nbits); b->acc = 0; b->nbits = 0; }

static int bit_length(int v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

static void encode_block(bitw* b, const int16_t* zz, int* pred, const huff* dc, const huff* ac) {
    int t = zz[0] - *pred, t2 = t;
    *pred = zz[0];
    if (t < 0) { t = -t; t2--; }
    int nb = bit_length(t);
    bits_put(b, dc->code[nb], dc->size[nb]);
    bits_put(b, (unsigned)t2, nb);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        t = zz[k];
        if (!t) { ++run; continue; }
        while (run > 15) { bits_put(b, ac->code[0xF0], ac->size[0xF0]); run -= 16; }
        t2 = t;
        if (t < 0) { t = -t; t2--; }
        nb = bit_length(t);
        bits_put(b, ac->code[(run << 4) + nb], ac->size[(run << 4) + nb]);
        bits_put(b, (unsigned)t2, nb);
        run = 0;
    }
    if (run) bits_put(b, ac->code[0], ac->size[0]);
}

/* Header only (what gsa_jpeg_header must produce).  Returns the byte count (may exceed cap: nothing is written past it). */
API int64_t gsao_jpeg_header(int32_t H, int32_t W, int32_t quality, int32_t restart, uint8_t* out, int64_t cap) {
    uint8_t ql[64], qc[64];
    quant_table(kBaseLuma, quality, ql);
    quant_table(kBaseChroma, quality, qc);
    sink s = {out, 0, (size_t)cap};
    write_header(&s, H, W, ql, qc, restart);
    return (int64_t)s.n;
}

/* rgb: H x W x 3 bytes, H and W multiples of 16.  restart = MCUs per restart interval (0 = none).
 * Returns the file size (may exceed cap), or -1 for bad arguments. */
API int64_t gsao_jpeg_encode(int32_t H, int32_t W, const uint8_t* rgb, int32_t quality, int32_t restart, uint8_t* out, int64_t cap) {
    if (H <= 0 || W <= 0 || H % 16 || W % 16 || H > 65535 || W > 65535 || restart < 0 || restart > 65535 || !rgb) return -1;
    uint8_t ql[64], qc[64];
    quant_table(kBaseLuma, quality, ql);
    quant_table(kBaseChroma, quality, qc);
    huff dcl, dcc, acl, acc;
    build_huff(kDcLumaBits, kDcVals, &dcl);
    build_huff(kDcChromaBits, kDcVals, &dcc);
    build_huff(kAcLumaBits, kAcLumaVals, &acl);
    build_huff(kAcChromaBits, kAcChromaVals, &acc);
    sink s = {out, 0, (size_t)cap};
    write_header(&s, H, W, ql, qc, restart);
    bitw b = {&s, 0, 0};
    int pred[3] = {0, 0, 0};
    int togo = restart, rst = 0;
    for (int my = 0; my < H / 16; ++my)
        for (int mx = 0; mx < W / 16; ++mx) {
            if (restart && togo == 0) {          /* restart marker in front of the MCU that opens a new interval */
                bits_flush(&b);
                put16(&s, 0xFFD0 + rst);
                rst = (rst + 1) & 7;
                pred[0] = pred[1] = pred[2] = 0;
                togo = restart;
            }
            int32_t Y[16][16], Cb[16][16], Cr[16][16];
            for (int y = 0; y < 16; ++y)
                for (int x = 0; x < 16; ++x) {
                    const uint8_t* px = rgb + ((size_t)(my * 16 + y) * W + mx * 16 + x) * 3;
                    const int32_t r = px[0], g = px[1], bl = px[2];
                    /* 16-bit fixed point: Y = .299R+.587G+.114B, Cb = -.16874R-.33126G+.5B+128, Cr = .5R-.41869G-.08131B+128 */
                    Y[y][x] = (19595 * r + 38470 * g + 7471 * bl + 32768) >> 16;
                    Cb[y][x] = (-11059 * r - 21709 * g + 32768 * bl + (128 << 16) + 32767) >> 16;
                    Cr[y][x] = (32768 * r - 27439 * g - 5329 * bl + (128 << 16) + 32767) >> 16;
                }
            int32_t blk[64];
            int16_t zz[64];
            for (int by = 0; by < 2; ++by)
                for (int bx = 0; bx < 2; ++bx) {
                    for (int y = 0; y < 8; ++y)
                        for (int x = 0; x < 8; ++x) blk[8 * y + x] = Y[by * 8 + y][bx * 8 + x] - 128;
                    fdct_quant(blk, ql, zz);
                    encode_block(&b, zz, &pred[0], &dcl, &acl);
                }
            for (int c = 1; c < 3; ++c) {
                int32_t (*P)[16] = c == 1 ? Cb : Cr;
                for (int y = 0; y < 8; ++y)
                    for (int x = 0; x < 8; ++x)       /* 2x2 box, rounding bias alternating 1,2 along the row */
                        blk[8 * y + x] = ((P[2 * y][2 * x] + P[2 * y][2 * x + 1] + P[2 * y + 1][2 * x] + P[2 * y + 1][2 * x + 1] + 1 + (x & 1)) >> 2) - 128;
                fdct_quant(blk, qc, zz);
                encode_block(&b, zz, &pred[c], &dcc, &acc);
            }
            if (restart) --togo;
        }
    bits_flush(&b);
    put16(&s, 0xFFD9);
    return (int64_t)s.n;
}
