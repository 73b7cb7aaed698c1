// gsa_png.hip -- zlib/deflate stream of the mask PNGs on the GPU (include/gsa_png.h).
//
// Replaces the compression inside cv2.imwrite("mask_%06d.png") of reference main.py:102-103.  Byte work on the vector
// ALU, HBM-bound by nature (1 B/px read, ~0.01 B/px written):
//   png_rows_kernel     one wave per group of 4 rows: PNG filter "Up" (row minus the row above, 16 pixels per lane and
//                       load, bytes subtracted four at a time), run boundaries found in parallel, run-length tokens
//                       (a literal + matches of distance 1) emitted for the few boundaries, fixed-Huffman deflate
//                       block + empty stored block (byte alignment) into a private segment; Adler-32 partial sums
//   png_offsets_kernel  per mask: scan of the segment lengths, zlib header, final block, Adler-32 combination
//   png_gather_kernel   one wave per segment: scratch -> its final position in the mask's stream
// Contract: lossless -- zlib inflates the stream to the filtered scanlines, a PNG reader returns the mask.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/gsa.h"
#include "../../include/gsa_png.h"

namespace {

constexpr int kRowsPerSeg = 4;
constexpr unsigned kAdlerMod = 65521u;

// deflate packs bits LSB first; Huffman codes go in MSB first, i.e. bit-reversed
struct LsbSink {
    uint8_t* p;
    int n;
    unsigned long long acc;
    int nbits;
    bool writer;                 // every lane of the wave keeps the same state; one of them stores
    __device__ __forceinline__ void byte(unsigned b) {
        if (writer) p[n] = (uint8_t)b;
        ++n;
    }
    __device__ __forceinline__ void put(unsigned bits, int size) {
        acc |= (unsigned long long)bits << nbits;
        nbits += size;
        while (nbits >= 8) { byte((unsigned)(acc & 255ull)); acc >>= 8; nbits -= 8; }
    }
    __device__ __forceinline__ void code(unsigned c, int len) { put(__brev(c) >> (32 - len), len); }
    // RFC 1951 3.2.6 fixed code: literals 0-143 -> 8 bits from 00110000, 144-255 -> 9 bits from 110010000
    __device__ __forceinline__ void literal(unsigned v) {
        if (v < 144u) code(0x30u + v, 8);
        else code(0x190u + (v - 144u), 9);
    }
    // match of length 3..258 at distance 1: length symbol (+ extra bits), distance code 0 (5 bits)
    __device__ __forceinline__ void match1(int len) {
        int sym, ebits = 0;
        unsigned extra = 0;
        if (len == 258) sym = 285;
        else if (len <= 10) sym = 254 + len;
        else {
            const unsigned x = (unsigned)len - 3u;                 // 8..254
            ebits = (31 - __clz(x)) - 2;                           // 1..5
            sym = 257 + 4 * (ebits + 1) + (int)(x >> ebits) - 4;
            extra = x & ((1u << ebits) - 1u);
        }
        if (sym < 280) code((unsigned)(sym - 256), 7);             // 256-279: 7 bits from 0000000
        else code(0xC0u + (unsigned)(sym - 280), 8);               // 280-287: 8 bits from 11000000
        if (ebits) put(extra, ebits);
        put(0u, 5);                                                // distance 1
    }
    __device__ __forceinline__ void run(unsigned v, int n) {
        literal(v);
        --n;
        while (n >= 3) { const int m = n < 258 ? n : 258; match1(m); n -= m; }
        while (n-- > 0) literal(v);
    }
};

// One WAVE per group of 4 rows.  The filtered bytes are produced 1024 at a time (16 per lane), run boundaries are found
// in parallel (a byte differs from its predecessor; the predecessor of a lane's first byte comes from the lane below),
// and only the boundaries -- a handful per row of a mask -- are walked serially, with wave-uniform values, to emit the
// tokens; lane 0 stores the bytes.  The Adler-32 sums need no order: byte i of a segment of L bytes weighs (L - i).
struct RunState {
    unsigned rv;        // value of the open run
    int start;          // its first position in the segment's byte sequence (-1 = none yet)
};

__device__ __forceinline__ void boundary(LsbSink& o, RunState& st, unsigned v, int pos) {   // a new run of value v starts at pos
    if (st.start >= 0) o.run(st.rv, pos - st.start);
    st.rv = v;
    st.start = pos;
}

__global__ __launch_bounds__(256) void png_rows_kernel(const uint8_t* __restrict__ mask, int H, int W, int segs_per_img,
                                                       int total_segs, int segcap, uint8_t* __restrict__ scratch,
                                                       int* __restrict__ seglen, unsigned* __restrict__ adler) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= total_segs) return;                        // whole waves leave together
    const int img = t / segs_per_img, s = t - img * segs_per_img;
    const int y0 = s * kRowsPerSeg, y1 = min(y0 + kRowsPerSeg, H);
    const int Lseg = (y1 - y0) * (W + 1);
    LsbSink o{scratch + (size_t)t * segcap, 0, 0ull, 0, lane == 0};
    RunState st{0u, -1};
    unsigned long long s1 = 0, s2 = 0;
    o.put(2u, 3);                                       // BFINAL = 0, BTYPE = 01 (fixed Huffman)
    const uint8_t* base = mask + (size_t)img * H * W;
    for (int y = y0; y < y1; ++y) {
        const int rowpos = (y - y0) * (W + 1);          // position of this row's filter byte in the segment
        if (st.start < 0 || st.rv != 2u) boundary(o, st, 2u, rowpos);   // filter type 2 (Up)
        if (lane == 0) { s1 += 2u; s2 += 2ull * (unsigned long long)(Lseg - rowpos); }
        unsigned last = 2u;                             // the byte in front of the strip's first pixel
        const uint8_t* cur = base + (size_t)y * W;
        const uint8_t* up = base + (size_t)(y > 0 ? y - 1 : 0) * W;
        for (int x0 = 0; x0 < W; x0 += 1024) {
            const int px = x0 + lane * 16;
            const bool valid = px < W;
            uint4 c = make_uint4(0u, 0u, 0u, 0u), u = c;
            if (valid) {
                c = *reinterpret_cast<const uint4*>(cur + px);
                if (y > 0) u = *reinterpret_cast<const uint4*>(up + px);
            }
            const unsigned cw[4] = {c.x, c.y, c.z, c.w}, uw[4] = {u.x, u.y, u.z, u.w};
            unsigned dw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)                 // four byte-wise differences at once (no borrow across bytes)
                dw[k] = ((cw[k] | 0x80808080u) - (uw[k] & 0x7F7F7F7Fu)) ^ ((cw[k] ^ ~uw[k]) & 0x80808080u);
            unsigned prevb = __shfl_up(dw[3] >> 24, 1);
            if (lane == 0) prevb = last;
            unsigned bm = 0;                            // bit j: byte j starts a new run
            const int q = rowpos + 1 + px;              // position of this lane's first byte
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned before = (dw[k] << 8) | (k ? dw[k - 1] >> 24 : prevb);
                const unsigned x = dw[k] ^ before;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned byte = (dw[k] >> (8 * j)) & 255u;
                    if ((x >> (8 * j)) & 255u) bm |= 1u << (4 * k + j);
                    s1 += byte;
                    s2 += (unsigned long long)byte * (unsigned long long)(Lseg - (q + 4 * k + j));
                }
            }
            if (!valid) bm = 0;
            unsigned long long lanes = __ballot(bm != 0);
            while (lanes) {                             // wave-uniform walk over the boundaries of this strip
                const int L = __ffsll((long long)lanes) - 1;
                lanes &= lanes - 1;
                unsigned m = (unsigned)__builtin_amdgcn_readlane((int)bm, L);
                const unsigned w0 = (unsigned)__builtin_amdgcn_readlane((int)dw[0], L), w1 = (unsigned)__builtin_amdgcn_readlane((int)dw[1], L);
                const unsigned w2 = (unsigned)__builtin_amdgcn_readlane((int)dw[2], L), w3 = (unsigned)__builtin_amdgcn_readlane((int)dw[3], L);
                while (m) {
                    const int j = __ffs((int)m) - 1;
                    m &= m - 1;
                    const unsigned w = j < 4 ? w0 : (j < 8 ? w1 : (j < 12 ? w2 : w3));
                    boundary(o, st, (w >> (8 * (j & 3))) & 255u, rowpos + 1 + x0 + L * 16 + j);
                }
            }
            const int lastlane = min(63, (W - x0) / 16 - 1);
            last = (unsigned)__builtin_amdgcn_readlane((int)dw[3], lastlane) >> 24;
        }
    }
    o.run(st.rv, Lseg - st.start);
    o.put(0u, 7);                                       // end of block (symbol 256)
    o.put(0u, 3);                                       // empty stored block: BFINAL = 0, BTYPE = 00 ...
    if (o.nbits) o.put(0u, 8 - o.nbits);                // ... padded to a byte boundary ...
    o.byte(0x00); o.byte(0x00); o.byte(0xFF); o.byte(0xFF);   // ... LEN = 0, NLEN = ~0
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s1 += __shfl_xor(s1, d);
        s2 += __shfl_xor(s2, d);
    }
    if (lane == 0) {
        seglen[t] = o.n;
        adler[2 * t] = (unsigned)(s1 % kAdlerMod);
        adler[2 * t + 1] = (unsigned)(s2 % kAdlerMod);
    }
}

__global__ __launch_bounds__(256) void png_offsets_kernel(const int* __restrict__ seglen, const unsigned* __restrict__ adler,
                                                          int segs_per_img, int H, int W, int* __restrict__ segoff,
                                                          int* __restrict__ lengths, uint8_t* __restrict__ out,
                                                          long long out_stride) {
    __shared__ int wsum[4];
    __shared__ unsigned long long wsum1[4], wterm[4];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 2;                                      // the segments follow the 2-byte zlib header
    // Adler-32 of the concatenation without a serial pass: with S1_s = sum of segment s's bytes, S2_s = its own weighted
    // sum and L_s its length,  A = 1 + sum S1_s,  B = L_total + sum_s [ S2_s + L_s * (S1_0 + .. + S1_{s-1}) ]   (mod 65521)
    unsigned long long carry1 = 0, bsum = 0;            // sum of S1 so far; running sum of the bracket
    for (int base = 0; base < segs_per_img; base += 256) {
        const int i = base + tid;
        const bool in = i < segs_per_img;
        const size_t g = (size_t)img * segs_per_img + (in ? i : 0);
        const int v = in ? seglen[g] : 0;
        const unsigned long long a1 = in ? adler[2 * g] : 0ull, a2 = in ? adler[2 * g + 1] : 0ull;
        int incl = v;
        unsigned long long incl1 = a1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            const unsigned long long up1 = __shfl_up(incl1, d);
            if (lane >= d) { incl += up; incl1 += up1; }
        }
        if (lane == 63) { wsum[wave] = incl; wsum1[wave] = incl1; }
        __syncthreads();
        int before = 0, total = 0;
        unsigned long long before1 = 0, total1 = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            before += w < wave ? wsum[w] : 0; total += wsum[w];
            before1 += w < wave ? wsum1[w] : 0ull; total1 += wsum1[w];
        }
        if (in) segoff[g] = carry + before + incl - v;
        const unsigned long long rows = in ? (unsigned long long)min(kRowsPerSeg, H - i * kRowsPerSeg) : 0ull;
        const unsigned long long Ls = rows * (unsigned long long)(W + 1);
        const unsigned long long prefix1 = (carry1 + before1 + incl1 - a1) % kAdlerMod;        // S1 of all earlier segments
        unsigned long long term = in ? (a2 + (Ls % kAdlerMod) * prefix1 + Ls) % kAdlerMod : 0ull;   // + L_s: the leading 1 of A
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) term += __shfl_xor(term, d);
        __syncthreads();                                // wsum / wsum1 have been read
        if (lane == 0) wterm[wave] = term;
        __syncthreads();
        bsum += (wterm[0] + wterm[1]) + (wterm[2] + wterm[3]);
        carry += total;
        carry1 += total1;
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long A = (1ull + carry1) % kAdlerMod, B = bsum % kAdlerMod;
        const long long need = (long long)carry + 6;
        uint8_t* o = out + (size_t)img * out_stride;
        if (need <= out_stride) {
            o[0] = 0x78; o[1] = 0x01;                   // zlib header: deflate, 32 KiB window, fastest
            o[carry] = 0x03; o[carry + 1] = 0x00;       // final block: fixed Huffman, immediately end of block
            o[carry + 2] = (uint8_t)(B >> 8); o[carry + 3] = (uint8_t)(B & 255);
            o[carry + 4] = (uint8_t)(A >> 8); o[carry + 5] = (uint8_t)(A & 255);
            lengths[img] = (int)need;
        } else {
            lengths[img] = (int)-need;
        }
    }
}

__global__ __launch_bounds__(256) void png_gather_kernel(const uint8_t* __restrict__ scratch, const int* __restrict__ seglen,
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
    int segs_per_img, segcap;
    size_t scratch_bytes, seg_count;
};

bool geometry(int n, int H, int W, Geometry* g) {
    if (n < 1 || H < 1 || W < 16 || W % 16 || H > 65535 || W > 65535) return false;
    g->segs_per_img = (H + kRowsPerSeg - 1) / kRowsPerSeg;
    // worst case: every filtered byte a 9-bit literal, plus block header, end of block and the stored block
    g->segcap = ((kRowsPerSeg * (W + 1) * 9 + 7) / 8 + 24 + 15) & ~15;
    g->seg_count = (size_t)n * g->segs_per_img;
    g->scratch_bytes = g->seg_count * g->segcap;
    return true;
}

}  // namespace

extern "C" {

int64_t gsa_png_workspace_bytes(int32_t n, int32_t H, int32_t W) {
    Geometry g;
    if (!geometry(n, H, W, &g)) return GSA_ERR_INVALID;
    return (int64_t)(g.scratch_bytes + g.seg_count * (2 * sizeof(int) + 2 * sizeof(unsigned)) + 64);
}

int64_t gsa_png_max_stream_bytes(int32_t H, int32_t W) {
    Geometry g;
    if (!geometry(1, H, W, &g)) return GSA_ERR_INVALID;
    return (int64_t)g.segs_per_img * g.segcap + 8;
}

int gsa_png_encode(void* stream, int32_t n, int32_t H, int32_t W, const uint8_t* mask, void* workspace,
                   int64_t workspace_bytes, uint8_t* out, int64_t out_stride, int32_t* lengths) {
    Geometry g;
    if (!geometry(n, H, W, &g)) return GSA_ERR_INVALID;
    if (!mask || !workspace || !out || !lengths || out_stride < 8 || out_stride > 0x7fffffffll) return GSA_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(mask) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return GSA_ERR_INVALID;
    if (workspace_bytes < gsa_png_workspace_bytes(n, H, W) || g.seg_count > 0x7fffffffull / 4) return GSA_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    uint8_t* w = static_cast<uint8_t*>(workspace);
    uint8_t* scratch = w;
    int* seglen = reinterpret_cast<int*>(w + ((g.scratch_bytes + 15) & ~(size_t)15));
    int* segoff = seglen + g.seg_count;
    unsigned* adler = reinterpret_cast<unsigned*>(segoff + g.seg_count);
    const int total_segs = (int)g.seg_count;
    hipLaunchKernelGGL(png_rows_kernel, dim3((total_segs + 3) / 4), dim3(256), 0, s, mask, H, W, g.segs_per_img, total_segs,
                       g.segcap, scratch, seglen, adler);
    hipLaunchKernelGGL(png_offsets_kernel, dim3(n), dim3(256), 0, s, seglen, adler, g.segs_per_img, H, W, segoff, lengths, out,
                       (long long)out_stride);
    hipLaunchKernelGGL(png_gather_kernel, dim3((total_segs + 3) / 4), dim3(256), 0, s, scratch, seglen, segoff, g.segs_per_img,
                       total_segs, g.segcap, out, (long long)out_stride);
    return hipGetLastError() == hipSuccess ? GSA_OK : GSA_ERR_HIP;
}

}  // extern "C"
