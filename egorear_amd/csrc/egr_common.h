// Shared device/host helpers for the egorear HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "egorear_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define EGR_WAVE 64

static inline int egr_launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

__device__ __forceinline__ int64_t egr_map(const egr_nmap& m, int n) {
    int o = n / m.n_inner;
    int i = n - o * m.n_inner;
    return (int64_t)i * m.stride_inner + (int64_t)o * m.stride_outer;
}

__device__ __forceinline__ float egr_act(float v, int act) {
    if (act == EGR_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == EGR_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
