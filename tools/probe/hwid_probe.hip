// Probe (diagnostic): do 512 workgroups of 256 threads with ~75 KB of LDS land two per CU, and does HW_ID / XCC_ID identify the CU?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/hwid_probe.hip -o tools/probe/hwid_probe && tools/probe/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, unsigned long long* t) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
        t[blockIdx.x] = __builtin_readcyclecounter();
    }
    lds[threadIdx.x] = 1.f;
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(64);      // ~50 us: everybody is resident together
    if (lds[(threadIdx.x + 1) & 255] == 3.f) out[0] = 0;
}
int main() {
    unsigned* d; unsigned long long* t; hipMalloc(&d, 512 * 8); hipMalloc(&t, 512 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 75 * 1024, 0, d, t);
    std::vector<unsigned> h(1024); std::vector<unsigned long long> ht(512);
    hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, 4096, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cus;
    for (int b = 0; b < 512; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1];
        const unsigned cu = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
        cus[cu].push_back(b);
    }
    printf("distinct CU ids: %zu\n", cus.size());
    int hist[8] = {0};
    for (auto& kv : cus) hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
    for (int i = 0; i < 8; ++i) if (hist[i]) printf("  CUs holding %d workgroups: %d\n", i, hist[i]);
    int shown = 0;
    for (auto& kv : cus) { if (shown++ >= 6) break; printf("  cu %03x:", kv.first); for (int b : kv.second) printf(" wg %d (t %llu)", b, ht[b] - ht[0]); printf("\n"); }
    printf("raw hw[0]=%08x xcc[0]=%08x hw[1]=%08x xcc[1]=%08x\n", h[0], h[1], h[2], h[3]);
    return 0;
}
