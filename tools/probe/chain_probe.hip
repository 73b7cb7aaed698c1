// Probe (diagnostic, not product): what does a dependent kernel boundary cost on MI355X, and what does the grid-resident alternative cost?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/chain_probe.hip -o tools/probe/chain_probe && tools/probe/chain_probe
//
// The question behind VERDICT r4 item 2 ("one persistent launch for everything at <= 32 px"): the <= 32 px part of a step is a chain of
// 35-45 small DEPENDENT kernels (a few dozen workgroups each, every one reading what other workgroups of its predecessor wrote).  A chain of
// S stages is run three ways over the same data -- G workgroups x 256 threads, workgroup g reads the 16 KB slice workgroup (g + 1) % G
// wrote in the previous stage, adds one, writes its own slice (so the result proves the hand-over was coherent):
//   launches   S kernel launches on one stream (what the product does), eager and as a replayed hipGraph
//   barrier    ONE launch, a grid-wide barrier between stages: release fence + one atomic counter + spin + acquire fence
//   flags      ONE launch, point-to-point: a workgroup publishes its stage number in its own word and waits only for its producer's word
// plus each with `work` dependent fma per element in the stage body (16 elements per thread: 32 -> ~0.9 us, 128 -> ~3.4 us of arithmetic per
// stage), to see what the boundary costs beside real work.
// Every spin loop is bounded (a stuck chain sets an error word and leaves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int SLICE = 4096;      // floats per workgroup and stage (16 KB)
constexpr int THREADS = 256;
constexpr long SPIN_LIMIT = 4000000;

__device__ __forceinline__ float burn(float v, int work) {
    // `work` dependent fma per element (16 elements per thread): ~4 cycles each, opaque to the compiler
    const float one = 1.0f, zero = 0.0f;
    for (int i = 0; i < work; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(one), "v"(zero));
    return v;
}

__device__ __forceinline__ void stage_body(const float* in, float* out, int g, int G, int work) {
    const float* src = in + (size_t)((g + 1) % G) * SLICE;
    float* dst = out + (size_t)g * SLICE;
    for (int i = threadIdx.x; i < SLICE; i += THREADS) dst[i] = burn(src[i], work) + 1.0f;
}

__global__ __launch_bounds__(THREADS) void stage_kernel(const float* in, float* out, int G, int work) {
    stage_body(in, out, blockIdx.x, G, work);
}

// mode 0: counter barrier; mode 1: producer flag
__global__ __launch_bounds__(THREADS) void chain_kernel(float* a, float* b, int G, int stages, int work, int mode, unsigned* sync, unsigned* err) {
    const int g = blockIdx.x;
    unsigned* counter = sync;            // mode 0
    unsigned* flags = sync + 64;         // mode 1: one word per workgroup, 64 B apart
    for (int s = 0; s < stages; ++s) {
        const float* in = (s & 1) ? b : a;
        float* out = (s & 1) ? a : b;
        if (s > 0) {
            if (threadIdx.x == 0) {
                long spins = 0;
                if (mode == 0) {
                    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * s)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT) { atomicExch(err, 1u); break; }
                    }
                } else {
                    // the producer of my input finished stage s - 1; the consumer of my previous output (g - 1) finished READING it when it
                    // published stage s - 1 too (it read my stage s - 2 output, which stage s overwrites) -- wait for both
                    const unsigned* pf = flags + ((g + 1) % G) * 16;
                    const unsigned* cf = flags + ((g + G - 1) % G) * 16;
                    while (__hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s ||
                           __hip_atomic_load(cf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT) { atomicExch(err, 1u); break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        stage_body(in, out, g, G, work);
        __syncthreads();                 // every thread's stores are issued
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (mode == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(flags + g * 16, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static bool check(const float* d, int G, int stages, const char* what) {
    std::vector<float> h((size_t)G * SLICE);
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i)
        if (h[i] != (float)stages) { printf("  %s: WRONG value %g at %zu (expected %d)\n", what, h[i], i, stages); return false; }
    return true;
}

int main() {
    const int S = 40, REP = 20;
    float *a, *b; unsigned *sync, *err;
    CK(hipMalloc(&a, 256 * SLICE * sizeof(float))); CK(hipMalloc(&b, 256 * SLICE * sizeof(float)));
    CK(hipMalloc(&sync, 64 * 1024)); CK(hipMalloc(&err, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("chain of %d dependent stages, 16 KB per workgroup and stage, microseconds PER STAGE (median-free mean of %d chains after a warm-up)\n", S, REP);
    printf("%5s %6s | %9s %9s | %9s %9s\n", "G", "work", "launches", "graph", "barrier", "flags");
    for (int G : {8, 32, 64, 128, 256}) {
        for (int work : {0, 32, 128}) {
            float t_launch, t_graph, t_bar, t_flag;
            bool ok = true;
            // ---- launches
            auto run_chain = [&](hipStream_t s) {
                for (int i = 0; i < S; ++i) hipLaunchKernelGGL(stage_kernel, dim3(G), dim3(THREADS), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, G, work);
            };
            CK(hipMemsetAsync(a, 0, 256 * SLICE * sizeof(float), st));
            run_chain(st); CK(hipStreamSynchronize(st));
            ok &= check(a, G, S, "launches");
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < REP; ++r) run_chain(st);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&t_launch, e0, e1));
            // ---- graph
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); run_chain(st); CK(hipStreamEndCapture(st, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(exec, st));
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&t_graph, e0, e1));
            CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
            // ---- persistent
            float* tp[2] = {&t_bar, &t_flag};
            for (int mode = 0; mode < 2; ++mode) {
                auto one = [&]() {
                    CK(hipMemsetAsync(sync, 0, 64 * 1024, st));
                    hipLaunchKernelGGL(chain_kernel, dim3(G), dim3(THREADS), 0, st, a, b, G, S, work, mode, sync, err);
                };
                CK(hipMemsetAsync(a, 0, 256 * SLICE * sizeof(float), st)); CK(hipMemsetAsync(err, 0, 4, st));
                one(); CK(hipStreamSynchronize(st));
                ok &= check(a, G, S, mode ? "flags" : "barrier");
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < REP; ++r) one();
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(tp[mode], e0, e1));
                unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                if (herr) { printf("  spin limit hit (mode %d)\n", mode); ok = false; }
            }
            const float k = 1000.0f / (REP * S);
            printf("%5d %6d | %9.2f %9.2f | %9.2f %9.2f %s\n", G, work, t_launch * k, t_graph * k, t_bar * k, t_flag * k, ok ? "" : " (result check FAILED)");
            fflush(stdout);
        }
    }
    return 0;
}
