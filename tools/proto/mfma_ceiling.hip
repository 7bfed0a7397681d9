// What the bf16 matrix cores of this chip sustain on non-trivial data with nothing else going on: register operands only, one or two
// waves per SIMD, v_mfma_f32_32x32x16_bf16 back to back on four accumulators (random bf16 operands; all-zero operands for contrast).
// The spec peak (2516.6 TFLOP/s at 2.4 GHz) is not reachable under load: the chip lowers its clock.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_ceiling mfma_ceiling.hip ;  run: ./mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ ops, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint4 va = ops[(i * 64 + lane)], vb = ops[((i + 6) * 64 + lane)];
        __builtin_memcpy(&a[i], &va, 16);
        __builtin_memcpy(&b[i], &vb, 16);
    }
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(t + q) % 6], b[(t * 2 + q) % 6], acc[q], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks_per_cu[2] = {1, 2};
    std::vector<uint16_t> h(12 * 64 * 8);
    uint4* d;
    float* o;
    CK(hipMalloc(&d, h.size() * 2));
    CK(hipMalloc(&o, 4096 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int zero = 0; zero < 2; ++zero) {
        uint64_t s = 88172645463325252ull;
        for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = zero ? 0 : (uint16_t)((s >> 33) & 0xbfff); }   // random sign / mantissa, exponent below 2^64
        CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (int bi = 0; bi < 2; ++bi) {
            const int blocks = 256 * blocks_per_cu[bi], iters = 20000;
            mfma_loop<<<blocks, 256>>>(d, o, 1000);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < 5; ++rep) mfma_loop<<<blocks, 256>>>(d, o, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = 5.0 * blocks * 4.0 * iters * 24.0 * 32.0 * 32.0 * 16.0 * 2.0;
            printf("%s operands, %d workgroup(s) of 4 waves per CU: %.0f TFLOP/s of bf16 MFMA (%.1f ms)\n", zero ? "all-zero" : "random  ", blocks_per_cu[bi],
                   flops / (ms * 1e-3) / 1e12, ms);
        }
    }
    return 0;
}
