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

}  // namespace lean

// the fp32 planes post_rows_kernel<4> takes (launch_post decides that); GSA_POST_PK: 0 = post_rows_kernel, 1 = packed, 2 = packed + non-temporal stores
int post_pk_mode() {
    static const int mode = getenv("GSA_POST_PK") ? atoi(getenv("GSA_POST_PK")) : 1;
    return mode;
}

hipError_t launch_post_pk(const PostParams& q, dim3 grid, size_t lds, hipStream_t s) {
    if (q.bf16) hipLaunchKernelGGL((lean::post_rows_pk<false, true>), grid, dim3(256), lds, s, q);
    else if (post_pk_mode() >= 2) hipLaunchKernelGGL((lean::post_rows_pk<true, false>), grid, dim3(256), lds, s, q);
    else hipLaunchKernelGGL((lean::post_rows_pk<false, false>), grid, dim3(256), lds, s, q);
    return hipGetLastError();
}

}  // namespace gsa
