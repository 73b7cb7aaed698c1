// Probe: what does v_mfma_f32_16x16x16_bf16 compute, bit for bit?  Writes T trials of
// (A bf16 16x16, B bf16 16x16, C f32 16x16, D f32 16x16) to stdout as raw binary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const uint16_t* A, const uint16_t* B, const float* C, float* D, int T) {
    const int lane = threadIdx.x;
    for (int t = 0; t < T; ++t) {
        const uint16_t* a = A + t * 256; const uint16_t* b = B + t * 256;
        const float* c = C + t * 256; float* d = D + t * 256;
        s16x4 av, bv; f32x4 cv;
        for (int j = 0; j < 4; ++j) {
            av[j] = (short)a[(lane & 15) * 16 + 4 * (lane >> 4) + j];      // A[i][k]
            bv[j] = (short)b[(4 * (lane >> 4) + j) * 16 + (lane & 15)];    // B[k][n]
            cv[j] = c[(4 * (lane >> 4) + j) * 16 + (lane & 15)];           // C[row][col]
        }
        f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, cv, 0, 0, 0);
        for (int j = 0; j < 4; ++j) d[(4 * (lane >> 4) + j) * 16 + (lane & 15)] = dv[j];
    }
}
int main(int argc, char** argv) {
    const int T = 256;
    std::vector<uint16_t> A(T * 256), B(T * 256); std::vector<float> C(T * 256), D(T * 256);
    srand(1234);
    auto rb = [&](int spread) { // random bf16 with exponent in [127-spread, 127+spread]
        uint16_t s = rand() & 1, e = 127 - spread + rand() % (2 * spread + 1), m = rand() & 0x7f;
        return (uint16_t)((s << 15) | (e << 7) | m);
    };
    for (int t = 0; t < T; ++t) {
        const int spread = 1 + t % 12;
        for (int i = 0; i < 256; ++i) {
            A[t * 256 + i] = rb(spread); B[t * 256 + i] = rb(spread);
            uint32_t cb = ((uint32_t)rb(spread) << 16) | (rand() & 0xffff);
            float cf; memcpy(&cf, &cb, 4); C[t * 256 + i] = (t % 3 == 0) ? 0.f : cf;
        }
    }
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC, dD, T);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    FILE* f = fopen(argc > 1 ? argv[1] : "probe.bin", "wb");
    fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(C.data(), 4, C.size(), f); fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    return 0;
}
