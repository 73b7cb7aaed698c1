// Probe (diagnostic, not product): how do f32 MFMAs share a SIMD with vector-ALU / LDS instructions on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_probe.hip -o tools/probe/mfma_valu_probe && tools/probe/mfma_valu_probe
// Every workgroup = 8 waves (two per SIMD) or 4 waves (one per SIMD); one workgroup per CU.  Roles by wave id.
// Prints shader cycles (s_memtime) per loop iteration for the MFMA waves and for the partner waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VALU1(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(ka), "v"(kb))

// ROLE_A / ROLE_B: 0 idle, 1 = 8x mfma16x16x4 per iter, 2 = 4x mfma32x32x2 per iter (same FLOP), 3 = 32 VALU per iter,
// 4 = 8 ds_write_b128 + 8 ds_read_b128 per iter, 5 = per iter 8x (mfma16 + K valu) interleaved in ONE wave, 6 = 4x(mfma32 + 2K valu)
template <int ROLE, int K>
__device__ __forceinline__ void body(int iters, float ka, float kb, float* lds, float& sink) {
    f32x4 acc[8];
    f32x16 big[4];
    float v[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0, 0, 0, 0}; v[i] = ka + i; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    f32x4 w = {ka, kb, ka, kb};
    for (int it = 0; it < iters; ++it) {
        if constexpr (ROLE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka, kb, acc[i], 0, 0, 0);
        } else if constexpr (ROLE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka, kb, big[i], 0, 0, 0);
        } else if constexpr (ROLE == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) VALU1(v[i]);
        } else if constexpr (ROLE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(lds + (threadIdx.x * 4 + i * 2048) % 8192) = w;
#pragma unroll
            for (int i = 0; i < 8; ++i) { f32x4 t = *reinterpret_cast<volatile f32x4*>(lds + (threadIdx.x * 4 + i * 2048) % 8192); v[i] += t[0]; }
        } else if constexpr (ROLE == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka, kb, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) VALU1(v[(i + k) & 7]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (ROLE == 6) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka, kb, big[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 2 * K; ++k) VALU1(v[(i + k) & 7]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i];
    for (int i = 0; i < 4; ++i) s += big[i][0] + big[i][15];
    sink = s;
}

template <int RA, int RB, int K>
__global__ __launch_bounds__(512) void probe(int iters, float ka, float kb, unsigned long long* cyc, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    float sink = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) body<RA, K>(iters, ka, kb, lds, sink); else body<RB, K>(iters, ka, kb, lds, sink);
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (sink == 123.456f) out[threadIdx.x] = sink;
}

template <int RA, int RB, int K>
static void run(const char* name, int threads) {
    const int iters = 2000, blocks = 256;
    unsigned long long* d; float* o;
    hipMalloc(&d, sizeof(unsigned long long) * blocks * 8); hipMalloc(&o, 4096);
    hipMemset(d, 0, sizeof(unsigned long long) * blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<RA, RB, K><<<blocks, threads>>>(10, 1.0f, 0.5f, d, o);
    hipEventRecord(e0);
    probe<RA, RB, K><<<blocks, threads>>>(iters, 1.0f, 0.5f, d, o);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < blocks; ++i) { for (int w = 0; w < 4; ++w) a += h[i * 8 + w]; for (int w = 4; w < 8; ++w) b += h[i * 8 + w]; }
    a /= blocks * 4.0 * iters; b /= blocks * 4.0 * iters;
    printf("%-64s waves0-3 %8.1f cyc/iter   waves4-7 %8.1f cyc/iter   kernel %.3f ms\n", name, a, threads > 256 ? b : 0.0, ms);
    hipFree(d); hipFree(o);
}

int main() {
    printf("per iteration: an MFMA wave issues 8x mfma_f32_16x16x4 (nominal 8*32 = 256 cyc) or 4x mfma_f32_32x32x2 (4*64 = 256 cyc); a VALU wave 32 v_fma (nominal 128 cyc alone)\n");
    run<1, 0, 0>("A: mfma16 alone (1 wave/SIMD)", 256);
    run<2, 0, 0>("A: mfma32 alone (1 wave/SIMD)", 256);
    run<3, 0, 0>("A: 32 valu alone (1 wave/SIMD)", 256);
    run<4, 0, 0>("A: 8 ds_write_b128 + 8 ds_read_b128 alone", 256);
    run<1, 1, 0>("A: mfma16 | B: mfma16", 512);
    run<1, 3, 0>("A: mfma16 | B: 32 valu", 512);
    run<2, 3, 0>("A: mfma32 | B: 32 valu", 512);
    run<1, 4, 0>("A: mfma16 | B: 8 ds_write + 8 ds_read b128", 512);
    run<2, 4, 0>("A: mfma32 | B: 8 ds_write + 8 ds_read b128", 512);
    run<5, 0, 1>("A: 8x(mfma16 + 1 valu) one wave/SIMD", 256);
    run<5, 0, 2>("A: 8x(mfma16 + 2 valu) one wave/SIMD", 256);
    run<5, 0, 4>("A: 8x(mfma16 + 4 valu) one wave/SIMD", 256);
    run<5, 0, 6>("A: 8x(mfma16 + 6 valu) one wave/SIMD", 256);
    run<6, 0, 1>("A: 4x(mfma32 + 2 valu) one wave/SIMD", 256);
    run<6, 0, 2>("A: 4x(mfma32 + 4 valu) one wave/SIMD", 256);
    run<6, 0, 4>("A: 4x(mfma32 + 8 valu) one wave/SIMD", 256);
    run<6, 0, 6>("A: 4x(mfma32 + 12 valu) one wave/SIMD", 256);
    run<5, 5, 2>("A,B: both 8x(mfma16 + 2 valu), two waves/SIMD", 512);
    run<5, 3, 2>("A: 8x(mfma16 + 2 valu) | B: 32 valu", 512);
    return 0;
}
