// Probe (diagnostic, not product): what does the memory system of one MI355X deliver to a pass that READS a tensor and WRITES one of the same
// size (the shape of the post pass: 0.54 GB in, 0.54 GB out at 1024^2), against read-only and write-only streams of the same bytes?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/stream_probe.hip -o tools/probe/stream_probe && tools/probe/stream_probe
// The guide's "6.29 TB/s (float4 copy)" is the yardstick the HBM-bound kernels are priced against; this prints the same figure measured here,
// for several launch shapes, so that 0.75 of it can be told from 0.95 of what a read + write pass can reach at all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0 copy, 1 read-only (sum), 2 write-only; UNR = independent 16-byte accesses per thread and trip
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4, float* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * UNR) {
        f32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t j = i + u * stride;
            if (MODE != 2) v[u] = j < n4 ? in[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            else v[u] = f32x4{1.f, 2.f, 3.f, 4.f};
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t j = i + u * stride;
            if (MODE == 1) acc += v[u];
            else if (j < n4) out[j] = v[u];
        }
    }
    if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

template <int MODE, int UNR>
float run(const char* name, int blocks, const f32x4* in, f32x4* out, size_t n4, float* sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rep = 10;
    hipLaunchKernelGGL((stream_kernel<MODE, UNR>), dim3(blocks), dim3(256), 0, 0, in, out, n4, sink);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < rep; ++r) hipLaunchKernelGGL((stream_kernel<MODE, UNR>), dim3(blocks), dim3(256), 0, 0, in, out, n4, sink);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= rep;
    const double bytes = (double)n4 * 16 * (MODE == 0 ? 2 : 1);
    printf("%-10s unroll %d  workgroups %6d | %.3f ms  %.2f TB/s (bytes moved: %s)\n", name, UNR, blocks, ms, bytes / ms / 1e9,
           MODE == 0 ? "read + written" : MODE == 1 ? "read" : "written");
    fflush(stdout);
    return ms;
}

int main() {
    const size_t bytes = (size_t)8 * 1024 * 1024 * 16 * 4;      // 0.54 GB: one 1024^2 x 16-channel fp32 tensor of a batch of 8
    const size_t n4 = bytes / 16;
    f32x4 *in, *out; float* sink;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 0, bytes)); CK(hipMemset(out, 0, bytes));
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; ++r) CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("hipMemcpyDtoD                          | %.3f ms  %.2f TB/s (read + written)\n", ms, 2.0 * bytes / ms / 1e9);
    }
    for (int blocks : {1024, 2048, 4096, 16384, (int)(n4 / 256)}) {
        run<0, 1>("copy", blocks, in, out, n4, sink);
        run<0, 4>("copy", blocks, in, out, n4, sink);
    }
    for (int blocks : {2048, 16384}) {
        run<1, 4>("read", blocks, in, out, n4, sink);
        run<2, 4>("write", blocks, in, out, n4, sink);
    }
    return 0;
}
