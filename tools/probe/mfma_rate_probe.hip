// Probe (diagnostic, not product): what does v_mfma_f32_16x16x4_f32 sustain on MI355X when it is fed like the convolution kernels feed it?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate_probe.hip -o tools/probe/mfma_rate_probe && tools/probe/mfma_rate_probe
//
// Every f32 convolution kernel of this repository ends up near 64 shader cycles per MFMA and SIMD (0.5 of the 32-cycle issue rate) whatever
// its vector-ALU count, occupancy or memory traffic (round 5: two output groups per wave and cache-resident operands both left the time
// unchanged).  This probe isolates the matrix pipe: per loop trip 64 MFMAs on 16 accumulator tiles (the kernels' chain order: four
// k-steps per tile, tiles round-robin), operands
//   same      one A and one B register for all of them (what mfma_valu_probe measured: 32 cycles)
//   regs      16 x 4 distinct A and B registers, loaded once
//   lds       A operands re-read from LDS every trip (16 ds_read_b128 per 64 MFMAs, as the kernels' weight reads), B from registers
//   lds2      A and B re-read from LDS every trip (32 ds_read_b128 per 64 MFMAs)
// with one or two waves per SIMD, on ONE workgroup or on every CU (does the rate drop when the whole chip runs MFMAs?).
// Output: shader cycles (s_memtime) per MFMA and SIMD, and the clock implied by wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(int iters, const float* in, float* out, unsigned long long* cycles) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 16 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 16 * 256; i += blockDim.x) lds[i] = in[i & 1023];
    __syncthreads();
    f32x4 acc[16], a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        a[i] = *reinterpret_cast<const f32x4*>(lds + i * 256 + lane * 4);
        b[i] = *reinterpret_cast<const f32x4*>(lds + 4096 + i * 256 + lane * 4);
    }
    // explicit ds_read_b128 (the compiler would split or hoist plain loads), software-pipelined as the kernels' schedule is: the reads of
    // frequency row fb + 1 are issued in front of the 16 MFMAs of row fb and waited for behind them
    const unsigned la = (unsigned)(size_t)(lds) + lane * 16, lb = la + 16384;
    auto issue = [&](int fb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[fb * 4 + j]) : "v"(la), "n"(0), "i"((fb * 4 + j) * 1024));
            if (MODE >= 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[fb * 4 + j]) : "v"(lb), "n"(0), "i"((fb * 4 + j) * 1024));
        }
    };
    auto landed = [&](int fb) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[fb * 4]), "+v"(a[fb * 4 + 1]), "+v"(a[fb * 4 + 2]), "+v"(a[fb * 4 + 3]),
                                              "+v"(b[fb * 4]), "+v"(b[fb * 4 + 1]), "+v"(b[fb * 4 + 2]), "+v"(b[fb * 4 + 3]));
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE >= 2) issue(0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            if (MODE >= 2) { landed(fb); issue((fb + 1) & 3); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = fb * 4 + j;
                    if (MODE == 0) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][0], b[0][0], acc[f], 0, 0, 0);
                    else acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f][cg], b[f][cg], acc[f], 0, 0, 0);
                }
            // the row just multiplied is overwritten by the reads issued three rows later: its MFMAs have long read their operands
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (MODE >= 2) landed(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) cycles[(size_t)blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int blocks, const float* in, float* out, unsigned long long* cyc) {
    const int iters = 2000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, 10, in, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, iters, in, out, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(8); CK(hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost));
    // wall time at the nominal 2.4 GHz per MFMA a SIMD issued (the launch's few microseconds are inside); wave 0's own counter beside it
    const double per_simd = ms * 1e-3 * 2.4e9 / ((double)iters * 64.0 * waves_per_simd);
    printf("%-5s waves/SIMD %d  workgroups %4d | %6.1f cycles per MFMA and SIMD (wall, 2.4 GHz) | wave 0 saw %.1f cycles per own MFMA\n",
           name, waves_per_simd, blocks, per_simd, (double)h[0] / (iters * 64.0));
    fflush(stdout);
}

int main() {
    float *in, *out; unsigned long long* cyc;
    CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 64));
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1.0f / (1 + i % 7);
    CK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
    for (int blocks : {1, 256}) {
        for (int w : {1, 2}) {
            run<0>("same", w, blocks, in, out, cyc);
            run<1>("regs", w, blocks, in, out, cyc);
            run<2>("lds", w, blocks, in, out, cyc);
            run<3>("lds2", w, blocks, in, out, cyc);
        }
    }
    return 0;
}
