// Probe (diagnostic, not product): which operand path feeds a barrier-free Winograd F(2x2,3x3) wave fastest on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/patch_load_probe.hip -o tools/probe/patch_load_probe && tools/probe/patch_load_probe
// Workload = the A operand of g.1024.conv_2 / d.cvt_8 / d.main_7.b: an NHWC fp32 tensor (8, 1024, 1024, 16); one wave owns an 8x8
// output quadrant = 16 Winograd tiles; lane (i16, kq) needs the 4x4 input patch of tile i16, channels 4kq..4kq+3 (16 x 16 bytes).
//   PATH 0: reference -- a plain coalesced read of every byte once (4 x 1 KB per wave item)
//   PATH 1: the patch straight from global memory into registers (16 global_load_dwordx4 per lane, every byte fetched ~2.5x, scattered
//           16-byte pieces at a 128-byte lane stride) -- no LDS, no barrier
//   PATH 2: coalesced 16-byte chunk loads of the 10x10 halo image -> ds_write_b128 into a WAVE-PRIVATE LDS image -> 16 ds_read_b128
//   PATH 3: the same image by LDS-DMA (global_load_lds_dwordx4, 7 x 1 KB per wave) -> 16 ds_read_b128
// MFMA = 1 adds the 64 v_mfma_f32_16x16x4_f32 of the item (operands = the patch registers), OCC = waves per SIMD the kernel is built for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 8, H = 1024, W = 1024, C = 16;
constexpr int RS = 10 * 16 + 4;      // LDS row stride of the 10x10 wave image in floats (bank-conflict free for the patch reads)
constexpr int IMG = 448 * 4;         // floats per wave image (7 rounds of 64 16-byte units >= 10 rows x 41 units)

template <int PATH, int MFMA, int OCC>
__global__ __launch_bounds__(256, OCC) void probe(const float* __restrict__ src, float* __restrict__ out, int tiles_per_wg, int total_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[PATH >= 2 ? 4 * IMG : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int ty = ((i16 >> 1) & 1) * 2 + ((i16 >> 2) & 1), tx = ((i16 >> 3) & 1) * 2 + (i16 & 1);      // (1,2,3,0): conflict-free with RS = 164
    const int nb = gridDim.x, b = blockIdx.x;
    const int xb = (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;      // XCD-contiguous tile ranges
    const int t0 = xb * tiles_per_wg, t1 = min(total_tiles, t0 + tiles_per_wg);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* img = smem + (PATH >= 2 ? wave * IMG : 0);
    // PATH 2 / 3: unit u = 64 k + lane of the wave image: row u / 41, 16-byte column unit u % 41 (unit 40 of a row = padding)
    int st_src[7], st_lds[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int u = 64 * k + lane, row = u / 41, cu = u % 41;
        const bool real = row < 10 && cu < 40;
        st_src[k] = real ? (row * W + (cu >> 2)) * C + (cu & 3) * 4 : 0;
        st_lds[k] = real ? row * RS + cu * 4 : -1;
    }
    const int poff = (2 * ty * W + 2 * tx) * C + kq * 4;               // PATH 1: patch origin relative to the halo origin
    const int pl = 2 * ty * RS + 2 * tx * 16 + kq * 4;                  // PATH 2 / 3: the same in the LDS image
    for (int t = t0; t < t1; ++t) {
        const int n = t / (64 * 64), r = t % (64 * 64), y0 = (r / 64) * 16 + (wave >> 1) * 8, x0 = (r % 64) * 16 + (wave & 1) * 8;
        const int hy = min(max(y0 - 1, 0), H - 10), hx = min(max(x0 - 1, 0), W - 10);      // halo origin, kept inside the image (probe)
        const float* base = src + ((size_t)(n * H + hy) * W + hx) * C;
        f32x4 d[16];
        if constexpr (PATH == 0) {
            const float* q = src + ((size_t)t * 4 + wave) * 1024 + lane * 4;      // 4 KB per wave item: every byte of the tensor once
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = *reinterpret_cast<const f32x4*>(q + k * 256);
#pragma unroll
            for (int k = 4; k < 16; ++k) d[k] = d[k & 3];
        } else if constexpr (PATH == 1) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[rr * 4 + c] = *reinterpret_cast<const f32x4*>(base + poff + (rr * W + c) * C);
        } else {
            if constexpr (PATH == 2) {
                f32x4 v[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) v[k] = *reinterpret_cast<const f32x4*>(base + st_src[k]);
#pragma unroll
                for (int k = 0; k < 7; ++k) if (st_lds[k] >= 0) *reinterpret_cast<f32x4*>(img + st_lds[k]) = v[k];
            } else {
#pragma unroll
                for (int k = 0; k < 7; ++k)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + st_src[k]),
                                                     (__attribute__((address_space(3))) void*)(img + k * 256), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if constexpr (PATH == 3) {
                // DMA image is linear in units (row stride 41 units = 164 floats): the same addresses as PATH 2
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[rr * 4 + c] = *reinterpret_cast<const volatile f32x4*>(img + pl + rr * RS + c * 16);
        }
        if constexpr (MFMA) {
#pragma unroll
            for (int f = 0; f < 16; ++f)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, d[f][cg], acc[f], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += d[k];
        }
    }
#pragma unroll
    for (int f = 0; f < 16; ++f) sum += acc[f];
    if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) out[tid] = sum[0];      // keeps everything alive
}

template <int PATH, int MFMA, int OCC>
static void run(const float* src, float* out, int cus) {
    const int total_tiles = N * 64 * 64;
    const int wgs = cus * OCC;
    const int per = (total_tiles + wgs - 1) / wgs;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<PATH, MFMA, OCC>), dim3(wgs), dim3(256), 0, 0, src, out, per, total_tiles);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<PATH, MFMA, OCC>), dim3(wgs), dim3(256), 0, 0, src, out, per, total_tiles);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)N * H * W * C * 4;
    printf("path %d mfma %d occ %d: %.4f ms  %.2f TB/s of unique input bytes\n", PATH, MFMA, OCC, ms, bytes / ms * 1e-9);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t elems = (size_t)N * H * W * C;
    float *src, *out;
    CHECK(hipMalloc(&src, elems * 4 + (1 << 20)));
    CHECK(hipMalloc(&out, 4096));
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-4f - 3.f;
    for (size_t o = 0; o < elems; o += h.size()) CHECK(hipMemcpy(src + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("%s, %d CUs; tensor %d x %d x %d x %d fp32 = %.0f MB\n", prop.name, cus, N, H, W, C, elems * 4e-6);
    run<0, 0, 2>(src, out, cus); run<0, 0, 3>(src, out, cus);
    run<1, 0, 2>(src, out, cus); run<1, 0, 3>(src, out, cus); run<1, 0, 4>(src, out, cus);
    run<2, 0, 2>(src, out, cus); run<2, 0, 3>(src, out, cus); run<2, 0, 4>(src, out, cus);
    run<3, 0, 2>(src, out, cus); run<3, 0, 3>(src, out, cus); run<3, 0, 4>(src, out, cus);
    run<0, 1, 2>(src, out, cus); run<0, 1, 3>(src, out, cus);
    run<1, 1, 2>(src, out, cus); run<1, 1, 3>(src, out, cus);
    run<2, 1, 2>(src, out, cus); run<2, 1, 3>(src, out, cus);
    run<3, 1, 2>(src, out, cus); run<3, 1, 3>(src, out, cus);
    return 0;
}
