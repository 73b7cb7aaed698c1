// Blur + AddNoise + Bias + LeakyReLU + statistics of the large planes (reference networks_stylegan.py:200-236, 267-305, 534-545) in packed
// fp32 arithmetic (round 5): post_rows_kernel<4>'s thread shape -- 4 consecutive x, 4 consecutive channels, 4 rows per group with the 3-row
// blur window sliding in registers -- with every per-element operation issued as v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 on channel
// pairs: 72 packed fma per output row instead of 144 scalar ones, 6 instead of 10 instructions per noise / bias / LeakyReLU pair, 10 instead
// of 20 for the quad sums.  The same operations in the same order per element (IEEE packed forms): the same bits as post_rows_kernel.
// NT: the output is written with non-temporal stores (it is re-read by the next kernel from HBM anyway: 0.5 GB per launch at 1024^2
// against 32 MB of L2), so the lines the halo rows of the neighbouring row groups are re-read from stay in L2.
#include "gsa_kernels.h"
#include "gsa_dev.h"

#include <cstdlib>
#include <mutex>

namespace gsa {
namespace lean {

// BF: the source and the output are bf16 tensors (bf16 mode: half the bytes; the arithmetic and the statistics stay fp32, the output is
// rounded to bf16 (RNE) when it is stored -- post_rows_kernel<4, true>'s rule)
template <bool NT, bool BF>
__global__ __launch_bounds__(256, 2) void post_rows_pk(PostParams p) {
    constexpr int RPT = 4, ES = BF ? 2 : 4;      // bytes per activation element
    extern __shared__ __attribute__((aligned(16))) unsigned long long sstat[];   // [2][C], then the blur taps [9][C] as floats
    float* const sW = reinterpret_cast<float*>(sstat + 2 * p.C);
    const int n = blockIdx.y;
    const int C4 = p.C >> 2, W4 = p.W >> 2;
    const int NG = p.row_groups > 1 ? p.row_groups : 1;
    const int total = (p.H / (RPT * NG)) * W4 * C4;
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) sstat[i] = 0ull;
    // the taps live in LDS, one 16-byte read per tap and row (the four channels of the thread), not in 36 registers: with them in
    // registers the packed form spills (288 bytes of scratch); the reads are volatile so that they are not hoisted back out of the loop
    for (int i = threadIdx.x; i < 9 * p.C; i += 256) sW[(i % 9) * p.C + i / 9] = p.blur[i];
    __syncthreads();
    const int idx = xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (idx < total) {
        const int cq = idx % C4, t = idx / C4;
        const int xq = t % W4;
        // the launcher guarantees W4 * C4 % 64 == 0: the 64 threads of a wave share their rows, so every row base is a scalar
        const int y0 = __builtin_amdgcn_readfirstlane((t / W4) * (RPT * NG));
        const int c = cq * 4, x0 = xq * 4;
        const char* const sb = reinterpret_cast<const char*>(p.src) + (size_t)n * p.H * p.W * p.C * ES;      // this sample: wave-uniform
        char* const ob = reinterpret_cast<char*>(p.out) + (size_t)n * p.H * p.W * p.C * ES;
        const char* const nzb = reinterpret_cast<const char*>(p.noise + (size_t)n * p.H * p.W);
        const volatile f32x4* wtap = reinterpret_cast<const volatile f32x4*>(sW + c);      // tap t of channels c .. c+3 at wtap[t * C4]
        const f32x4 sf = *reinterpret_cast<const f32x4*>(p.nscale + c);
        const f32x4 nb = *reinterpret_cast<const f32x4*>(p.nbias + c);
        const f32x2 k02 = {0.2f, 0.2f};
        const unsigned rowb = (unsigned)(p.W * p.C * ES);     // bytes per row
        unsigned xoff[6], xin = 0;                            // byte offset of (x0 - 1 + k, c) inside a row (clamped), inside-the-image bits
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int xx = x0 - 1 + k;
            if (xx >= 0 && xx < p.W) xin |= 1u << k;
            xoff[k] = (unsigned)(((xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx)) * p.C + c) * ES);
        }
        auto load_row = [&](f32x4 (&row)[6], int yy) {      // unconditional loads (clamped), padding = zeroed values; yy is wave-uniform
            const bool vy = yy >= 0 && yy < p.H;
            const int yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
            const char* rp = sb + (size_t)yc * rowb;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                row[k] = act_load4<BF>(reinterpret_cast<const float*>(rp + xoff[k]), 0);
                if (!(vy && ((xin >> k) & 1u))) row[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        f32x4 win[4][6];                                    // rows y-1, y, y+1 and the prefetched y+2 (ring of four)
        load_row(win[0], y0 - 1);
        load_row(win[1], y0);
        load_row(win[2], y0 + 1);
        unsigned long long I1[4] = {0, 0, 0, 0}, I2[4] = {0, 0, 0, 0};
        const int s2 = stat_s2(p.H * p.W);
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int y = y0 + gi * RPT + r;
            if (r + 1 < RPT || gi + 1 < NG) load_row(win[(r + 3) & 3], y + 2);      // in flight during this row's arithmetic
            const f32x4 nz = *reinterpret_cast<const f32x4*>(nzb + (size_t)y * (p.W * 4) + (unsigned)(x0 * 4));
            f32x2 v01[4], v23[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v01[q] = v23[q] = f32x2{0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                f32x4 wr[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) wr[kx] = wtap[(ky * 3 + kx) * C4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 tv = win[(r + ky) & 3][q + kx];
                        v01[q] = __builtin_elementwise_fma(tv.xy, wr[kx].xy, v01[q]);      // <2 x float> fma: v_pk_fma_f32 (same IEEE bits as two fmaf)
                        v23[q] = __builtin_elementwise_fma(tv.zw, wr[kx].zw, v23[q]);
                    }
            }
            char* orow = ob + (size_t)y * rowb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // tn = nscale * noise (rounded), (v + tn) + nbias, LeakyReLU
                const f32x2 nzq = {nz[q], nz[q]};
                const f32x2 t01 = sf.xy * nzq, t23 = sf.zw * nzq;
                f32x2 a = (v01[q] + t01) + nb.xy, b = (v23[q] + t23) + nb.zw;
                const f32x2 la = a * k02, lb = b * k02;
                a = f32x2{max1(a.x, la.x), max1(a.y, la.y)};
                b = f32x2{max1(b.x, lb.x), max1(b.y, lb.y)};
                v01[q] = a; v23[q] = b;
                const f32x4 o = {a.x, a.y, b.x, b.y};
                char* dst = orow + xoff[1 + q];      // x0 + q is inside the image: its clamped offset is the real one
                if constexpr (BF) act_store4<true>(reinterpret_cast<float*>(dst), 0, o);
                else if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dst));
                else *reinterpret_cast<f32x4*>(dst) = o;
            }
            // statistics per aligned x-quad and channel: s = (v0 + v1) + (v2 + v3), q = (v0 v0 + v1 v1) + (v2 v2 + v3 v3)
            const f32x2 s01 = (v01[0] + v01[1]) + (v01[2] + v01[3]);
            const f32x2 s23 = (v23[0] + v23[1]) + (v23[2] + v23[3]);
            const f32x2 q01 = (v01[0] * v01[0] + v01[1] * v01[1]) + (v01[2] * v01[2] + v01[3] * v01[3]);
            const f32x2 q23 = (v23[0] * v23[0] + v23[1] * v23[1]) + (v23[2] * v23[2] + v23[3] * v23[3]);
            I1[0] += to_fixed(s01.x, kStatScale1); I1[1] += to_fixed(s01.y, kStatScale1);
            I1[2] += to_fixed(s23.x, kStatScale1); I1[3] += to_fixed(s23.y, kStatScale1);
            I2[0] += to_fixed_sq(q01.x, s2); I2[1] += to_fixed_sq(q01.y, s2);
            I2[2] += to_fixed_sq(q23.x, s2); I2[3] += to_fixed_sq(q23.y, s2);
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


// ---- the same pass fed by LDS-DMA (round 5): bytes in flight without registers --------------------------------------------------------
// post_rows_pk holds its rows in registers: 252 of them at two waves per SIMD are 49 KB in flight per CU, and the pass runs at 0.73 of the
// copy rate.  Here ONE wave owns a strip of XT = 1024 / C pixels x all C channels (a row of it: 4 KB contiguous in the NHWC plane plus a
// halo pixel either side) and walks a band of BH rows: rows go from global memory straight into a wave-private ring of eight LDS rows
// (global_load_lds_dwordx4: five instructions per row, the row's 4 x-quads of noise riding in the fifth), seven rows ahead of the row
// being computed; the wave waits for its own DMA with counted s_waitcnt vmcnt (in-order completion: the row it needs is older than
// everything issued in the last five rows) -- no barrier at all.  The arithmetic is post_rows_pk's, operation for operation.
// OST: the output row goes through a wave-private 4 KB LDS row so that every store instruction writes 1 KB of consecutive bytes (lane =
// consecutive 16-byte chunks of the row) instead of sixteen 64-byte pieces 256 bytes apart.
template <int C, bool OST>
__global__ __launch_bounds__(256, 1) void post_dma(PostParams p) {
    constexpr int CPP = C / 4, XT = 1024 / C, NPX = XT + 2, ICH = NPX * CPP, NCH = XT / 4, ROWF = (ICH + NCH) * 4, R = 8;
    static_assert(ICH > 256 && ICH + NCH <= 320, "five DMA instructions per ring row");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned long long* const sstat = reinterpret_cast<unsigned long long*>(smem + 4 * R * ROWF + (OST ? 4 * 1024 : 0));      // [2][C]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ring = smem + wave * R * ROWF;
    float* const orow_lds = smem + 4 * R * ROWF + wave * 1024;      // OST: this wave's output row (XT pixels x C channels = 1024 floats)
    const int n = blockIdx.y, H = p.H, W = p.W, BH = p.row_groups;
    const int strips = W / XT, nbands = (H / BH) * strips;
    const int wb = __builtin_amdgcn_readfirstlane((int)xcd_block(blockIdx.x, gridDim.x) * 4 + wave);
    for (int i = tid; i < 2 * C; i += 256) sstat[i] = 0ull;
    __syncthreads();
    if (wb < nbands) {
        const int x0 = (wb % strips) * XT, yb = (wb / strips) * BH;
        const char* const sb = reinterpret_cast<const char*>(p.src) + (size_t)n * H * W * C * 4;
        char* const ob = reinterpret_cast<char*>(p.out) + (size_t)n * H * W * C * 4;
        const char* const nzb = reinterpret_cast<const char*>(p.noise + (size_t)n * H * W);
        const unsigned rowb = (unsigned)(W * C * 4);
        // DMA: chunk i = lane + 64 k of a ring row = 16 bytes of pixel x0 - 1 + i / CPP (clamped into the image; the compute side zeroes
        // what lies outside), or of the row's noise
        unsigned goff[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = lane + 64 * k;
            const int px = x0 - 1 + i / CPP;
            goff[k] = (unsigned)(((px < 0 ? 0 : (px >= W ? W - 1 : px)) * C + (i % CPP) * 4) * 4);
        }
        const bool tail_img = lane < ICH - 256, tail_nz = lane >= ICH - 256 && lane < ICH - 256 + NCH;
        const unsigned nzoff = (unsigned)((x0 + (lane - (ICH - 256)) * 4) * 4);
        auto dma_row = [&](int yy, int slot) {      // yy: wave-uniform, any value (clamped); slot: ring row
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            const char* rp = sb + (size_t)yc * rowb;
            float* dst = ring + slot * ROWF;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rp + goff[k]),
                                                 (__attribute__((address_space(3))) void*)(dst + k * 256), 16, 0, 0);
            const char* tp = tail_img ? rp + goff[4] : nzb + (size_t)yc * (W * 4) + nzoff;
            if (tail_img || tail_nz)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)tp,
                                                 (__attribute__((address_space(3))) void*)(dst + 4 * 256), 16, 0, 0);
        };
        // compute: thread = x-quad xq x channel quad cq of the strip
        const int cq = lane % CPP, xq = lane / CPP;
        const int c = cq * 4;
        unsigned xin = 0;                                  // inside-the-image bits of pixels x0 + 4 xq - 1 + k
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int xx = x0 + 4 * xq - 1 + k;
            if (xx >= 0 && xx < W) xin |= 1u << k;
        }
        const int lpix = (4 * xq * C + c);                 // float offset of the thread's first pixel (strip pixel 4 xq = image pixel x0 + 4 xq - 1)
        auto read_row = [&](f32x4 (&row)[6], int yy, int slot) {
            const bool vy = yy >= 0 && yy < H;
            const float* rp = ring + slot * ROWF + lpix;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                row[k] = *reinterpret_cast<const f32x4*>(rp + k * C);
                if (!(vy && ((xin >> k) & 1u))) row[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        f32x4 wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = f32x4{p.blur[(c + 0) * 9 + t], p.blur[(c + 1) * 9 + t], p.blur[(c + 2) * 9 + t], p.blur[(c + 3) * 9 + t]};
        const f32x4 sf = *reinterpret_cast<const f32x4*>(p.nscale + c);
        const f32x4 nb = *reinterpret_cast<const f32x4*>(p.nbias + c);
        const f32x2 k02 = {0.2f, 0.2f};
        const unsigned ooff = (unsigned)(((x0 + 4 * xq) * C + c) * 4);
        // the constants above are in registers before the first DMA is issued: a wait for them later would be a wait for every row in flight
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wt[0]), "+v"(wt[1]), "+v"(wt[2]), "+v"(wt[3]), "+v"(wt[4]), "+v"(wt[5]), "+v"(wt[6]), "+v"(wt[7]), "+v"(wt[8]));
        // ring slot of row y: (y - (yb - 1)) & 7
        for (int j = 0; j < R; ++j) dma_row(yb - 1 + j, j);                    // rows yb - 1 ... yb + 6
        f32x4 win[4][6];
        asm volatile("s_waitcnt vmcnt(25)" ::: "memory");                       // the first three rows: five rows of five are younger
        read_row(win[0], yb - 1, 0);
        read_row(win[1], yb, 1);
        read_row(win[2], yb + 1, 2);
        unsigned long long I1[4] = {0, 0, 0, 0}, I2[4] = {0, 0, 0, 0};
        const int s2 = stat_s2(H * W);
        for (int g4 = 0; g4 < BH; g4 += 4)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = g4 + r, y = yb + j;
            // row y + 2 has landed: it is older than the rows y + 3 ... y + 6 (prologue) or y + 3 ... y + 7 with the stores between them
            if (j < 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            read_row(win[(r + 3) & 3], y + 2, (j + 3) & 7);
            const f32x4 nz = *reinterpret_cast<const f32x4*>(ring + ((j + 1) & 7) * ROWF + ICH * 4 + xq * 4);
            // the slot of row y - 1 (read three rows ago) takes row y + 7; past the band's last halo row the same row again (an L2 hit that
            // keeps the count of operations in flight, and with it the wait above, the same to the end)
            dma_row(y + 7 <= yb + BH ? y + 7 : yb + BH, j & 7);
            f32x2 v01[4], v23[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v01[q] = v23[q] = f32x2{0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 tv = win[(r + ky) & 3][q + kx];
                        v01[q] = __builtin_elementwise_fma(tv.xy, wt[ky * 3 + kx].xy, v01[q]);
                        v23[q] = __builtin_elementwise_fma(tv.zw, wt[ky * 3 + kx].zw, v23[q]);
                    }
            char* orow = ob + (size_t)y * rowb + ooff;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 nzq = {nz[q], nz[q]};
                const f32x2 t01 = sf.xy * nzq, t23 = sf.zw * nzq;
                f32x2 a = (v01[q] + t01) + nb.xy, b = (v23[q] + t23) + nb.zw;
                const f32x2 la = a * k02, lb = b * k02;
                a = f32x2{max1(a.x, la.x), max1(a.y, la.y)};
                b = f32x2{max1(b.x, lb.x), max1(b.y, lb.y)};
                v01[q] = a; v23[q] = b;
                if constexpr (OST) *reinterpret_cast<f32x4*>(orow_lds + (4 * xq + q) * C + c) = f32x4{a.x, a.y, b.x, b.y};
                else *reinterpret_cast<f32x4*>(orow + q * (C * 4)) = f32x4{a.x, a.y, b.x, b.y};
            }
            if constexpr (OST) {
                char* ocont = ob + (size_t)y * rowb + (unsigned)(x0 * C * 4) + lane * 16;
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(ocont + k * 1024) = *reinterpret_cast<const f32x4*>(orow_lds + k * 256 + lane * 4);
            }
            const f32x2 s01 = (v01[0] + v01[1]) + (v01[2] + v01[3]);
            const f32x2 s23 = (v23[0] + v23[1]) + (v23[2] + v23[3]);
            const f32x2 q01 = (v01[0] * v01[0] + v01[1] * v01[1]) + (v01[2] * v01[2] + v01[3] * v01[3]);
            const f32x2 q23 = (v23[0] * v23[0] + v23[1] * v23[1]) + (v23[2] * v23[2] + v23[3] * v23[3]);
            I1[0] += to_fixed(s01.x, kStatScale1); I1[1] += to_fixed(s01.y, kStatScale1);
            I1[2] += to_fixed(s23.x, kStatScale1); I1[3] += to_fixed(s23.y, kStatScale1);
            I2[0] += to_fixed_sq(q01.x, s2); I2[1] += to_fixed_sq(q01.y, s2);
            I2[2] += to_fixed_sq(q23.x, s2); I2[3] += to_fixed_sq(q23.y, s2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA of this wave is in flight when its LDS is released
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&sstat[c + j], I1[j]);
            atomicAdd(&sstat[C + c + j], I2[j]);
        }
    }
    __syncthreads();
    if ((int)gridDim.x > p.prow) {
        for (int i = threadIdx.x; i < C; i += 256) {
            StatPart* a = p.partials + ((size_t)n * p.prow + (blockIdx.x & (kDirectRows - 1))) * C + i;
            atomicAdd(&a->s1, sstat[i]);
            atomicAdd(&a->s2, sstat[C + i]);
        }
        return;
    }
    for (int i = threadIdx.x; i < C; i += 256) {
        StatPart sp; sp.s1 = sstat[i]; sp.s2 = sstat[C + i];
        p.partials[((size_t)n * p.prow + blockIdx.x) * C + i] = sp;
    }
}

}  // namespace lean

// the fp32 planes post_rows_kernel<4> takes (launch_post decides that); GSA_POST_PK: 0 = post_rows_kernel, 1 = packed, 2 = packed + non-temporal stores
int post_pk_mode() {
    static const int mode = getenv("GSA_POST_PK") ? atoi(getenv("GSA_POST_PK")) : 1;
    return mode;
}

// rows per wave: 64, or 32 where that is what gives every SIMD of the chip two bands
int post_dma_band(const PostParams& p) {
    static const int forced = getenv("GSA_POST_DMA_BH") ? atoi(getenv("GSA_POST_DMA_BH")) : 0;      // A/B only
    if (forced >= 8 && forced % 4 == 0 && p.H % forced == 0) return forced;
    return (long)(p.H / 64) * (p.W / (1024 / p.C)) * 8 >= 2048 ? 64 : 32;
}

// the LDS-DMA form for the blurred fp32 planes with 16 or 32 channels (1024^2 and 512^2 of the FFHQ path); GSA_POST_DMA=0: post_rows_pk (same bits).
// Same box, FFHQ batch 8: 0.243 -> 0.225 ms at 1024^2, 0.116 -> 0.108 at 512^2 (gpurun_out/r5/pdma*); band height 32-64 rows equal, 16 and 128 slower.
bool post_dma_applies(const PostParams& p) {
    static const bool on = !(getenv("GSA_POST_DMA") && atoi(getenv("GSA_POST_DMA")) == 0);
    if (!on || p.bf16 || !p.blur || !p.src_per_sample || (p.C != 16 && p.C != 32)) return false;
    return p.W % (1024 / p.C) == 0 && p.H % post_dma_band(p) == 0 && p.H >= 64;
}

int post_dma_blocks(const PostParams& p) { return ((p.H / post_dma_band(p)) * (p.W / (1024 / p.C)) + 3) / 4; }

hipError_t launch_post_dma(const PostParams& p, int n, hipStream_t s) {
    PostParams q = p;
    q.row_groups = post_dma_band(p);
    const dim3 grid(post_dma_blocks(p), n);
    static std::mutex mu;
    static bool attr_done[64][2] = {};
    static const bool ost = !(getenv("GSA_POST_DMA_OST") && atoi(getenv("GSA_POST_DMA_OST")) == 0);
    const size_t lds = sizeof(float) * 4 * 8 * ((1024 / p.C + 2) * (p.C / 4) + 1024 / p.C / 4) * 4 + (ost ? 16384 : 0) + sizeof(unsigned long long) * 2 * p.C;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    {
        std::lock_guard<std::mutex> lk(mu);
        bool& done = attr_done[dev][p.C == 32];
        if (!done) {
            const void* fs[4] = {reinterpret_cast<const void*>(lean::post_dma<16, false>), reinterpret_cast<const void*>(lean::post_dma<32, false>),
                                 reinterpret_cast<const void*>(lean::post_dma<16, true>), reinterpret_cast<const void*>(lean::post_dma<32, true>)};
            for (const void* f : fs) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
            done = true;
        }
    }
    if (p.C == 16) { if (ost) hipLaunchKernelGGL((lean::post_dma<16, true>), grid, dim3(256), lds, s, q); else hipLaunchKernelGGL((lean::post_dma<16, false>), grid, dim3(256), lds, s, q); }
    else { if (ost) hipLaunchKernelGGL((lean::post_dma<32, true>), grid, dim3(256), lds, s, q); else hipLaunchKernelGGL((lean::post_dma<32, false>), grid, dim3(256), lds, s, q); }
    return hipGetLastError();
}

hipError_t launch_post_pk(const PostParams& q, dim3 grid, size_t lds, hipStream_t s) {
    if (q.bf16) hipLaunchKernelGGL((lean::post_rows_pk<false, true>), grid, dim3(256), lds, s, q);
    else if (post_pk_mode() >= 2) hipLaunchKernelGGL((lean::post_rows_pk<true, false>), grid, dim3(256), lds, s, q);
    else hipLaunchKernelGGL((lean::post_rows_pk<false, false>), grid, dim3(256), lds, s, q);
    return hipGetLastError();
}

}  // namespace gsa
