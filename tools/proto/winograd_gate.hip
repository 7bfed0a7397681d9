// Round 5 decision gate for Winograd on the stride-1 3x3 launches of the fp16 scheme ("Stage A").
//   1. ACCURACY against fp64, six operand distributions x K = 576 / 1152 / 4608 (cin 64 / 128 / 512), 3x3 / stride 1 / pad 1:
//        fp32 fma chain (direct)                                    - the yard-stick
//        f16 x 2 planes, 3 products, direct                         - what ships
//        Winograd F(2x2,3x3): weights G g G^T in fp64 -> per (position, output channel) power-of-two scale -> two fp16 planes;
//          input transform B^T d B in fp32 (adds only), pre-scale lowered by 2 bits for the 4x growth, split on the fly;
//          16 position GEMMs on v_mfma_f32_32x32x16_f16 (3 products), descaled fp32 accumulators -> A^T M A in fp32
//        the same with the position GEMMs as fp32 fma chains        - Winograd's own error without the split
//        Winograd F(2,3) along W only (1-D: 4 positions x 3 kernel rows, K = 3 cin per position; pre-scale lowered by 1 bit)
//      GO if the rms error is <= 2 x the fp32 fma chain's.
//   2. STRUCTURE: what the matrix pipe sustains when it is fed the way a Winograd tile would feed it inside the role-split
//      workgroup of conv_tapx_kernel (512 threads, 256 registers per lane => <= 128 accumulators per multiplying wave, one barrier
//      per 16-channel chunk, loading waves that transform / split / write the planes): MFMA pipe-busy of
//        direct 128 x 64 wave tile (9 taps / chunk, 0.5 KB of operands per MFMA)                - the shipped FN = 2 K loop
//        direct 128 x 32 (0.83 KB / MFMA)                                                       - the shipped FN = 1 K loop
//        F(2,3) 1-D: 64 pairs x 32 channels x 4 positions, 12 (position, row) steps / chunk    (1 KB / MFMA, 2x the plane bytes)
//        F(2x2,3x3): 4 positions per wave, 32 tiles x 64 channels (24 MFMAs / chunk / wave, 1 KB / MFMA, 32-KB planes / chunk)
//        F(2x2,3x3): 4 positions per wave, 64 tiles x 32 channels (64-KB planes / chunk: does not fit beside the staging area)
//      A variant pays if   (MFMAs per output of direct / of the variant) x busy(variant) / busy(direct) is well above 1.
//   build: hipcc --offload-arch=gfx950 -O3 -o winograd_gate winograd_gate.hip ;  run: ./winograd_gate
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split2_f16(float x0, float x1, float s, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}

// C[M][N] = A[M][K] W[N][K]^T, one wave per 32 x 32 tile (correctness harness).  MODE 0: fp32 fma chain in k order (W as fp32);
// MODE 1: A split on the fly (scale sa), W given as two fp16 planes (already scaled by sw[n]); the result times 1 / (sa sw[n]).
template <int MODE>
__global__ __launch_bounds__(64) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ W, const uint16_t* __restrict__ Wh,
                                                   const uint16_t* __restrict__ Wl, float* __restrict__ C, int M, int N, int K, float sa,
                                                   const float* __restrict__ sw) {
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int tn = blockIdx.x % (N / 32), tm = blockIdx.x / (N / 32);
    if (MODE == 0) {
        for (int r = 0; r < 16; ++r) {
            const float* a = A + (int64_t)(tm * 32 + half * 16 + r) * K;
            const float* w = W + (int64_t)(tn * 32 + l31) * K;
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(a[k], w[k], s);
            C[(int64_t)(tm * 32 + half * 16 + r) * N + tn * 32 + l31] = s;
        }
        return;
    }
    const float* ar = A + (int64_t)(tm * 32 + l31) * K + half * 8;
    const int64_t wo = (int64_t)(tn * 32 + l31) * K + half * 8;
    f32x16 acc = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = ar[k0 + j];
        unsigned ah[4], al[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) split2_f16(a[2 * p], a[2 * p + 1], sa, ah[p], al[p]);
        const h8 Ah = __builtin_bit_cast(h8, u32x4{ah[0], ah[1], ah[2], ah[3]}), Al = __builtin_bit_cast(h8, u32x4{al[0], al[1], al[2], al[3]});
        const h8 Whv = *reinterpret_cast<const h8*>(Wh + wo + k0), Wlv = *reinterpret_cast<const h8*>(Wl + wo + k0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Whv, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Wlv, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Whv, acc, 0, 0, 0);
    }
    const float inv = 1.f / (sa * sw[tn * 32 + l31]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        C[(int64_t)(tm * 32 + row) * N + tn * 32 + l31] = acc[r] * inv;
    }
}

// ------------------------------------------------------------------------------------------------ 2. the feed probe
// 512 threads.  Waves 0-3 multiply: per chunk STEPS steps; a step reads FM A fragments x 2 planes from the chunk's plane buffer in LDS
// (a different 1-KB slot per step and fragment) and has FN B fragments x 2 planes arrive from a 2-MiB L2-resident image (requested AHEAD
// steps ahead, AHEAD + 1 register sets), then issues 3 FM FN v_mfma_f32_32x32x16_f16 into accumulator set (step % NACC).  Waves 4-7 stand for the loading waves:
// per chunk they request GLD 16-byte loads per lane from a 64-MiB buffer (HBM / L2), run VAL v_fma_mix-class instructions and write
// LWR 16-byte slots per lane into the other plane buffer.  One s_barrier per chunk for all eight waves.
template <int FM, int FN, int STEPS, int NACC, int GLD, int VAL, int LWR, int PLANE_KB, int AHEAD>
__global__ __launch_bounds__(512, 1) void feed_kernel(const uint8_t* __restrict__ wimg, const uint8_t* __restrict__ act, float* __restrict__ out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int PB = PLANE_KB * 1024;                 // one chunk buffer (both planes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * PB / 16; i += 512) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0x3c003c00u + (unsigned)i, 0x38003c00u, 0x34003c00u ^ (unsigned)(i & 1023), 0x3c003400u};
    __syncthreads();
    float s = 0.f;
    if (wave < 4) {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wimg), 0, 2 << 20, 0x00020000);
        constexpr int NSET = AHEAD + 1;
        static_assert(STEPS % NSET == 0, "a step's weight set is a compile-time constant");
        h8 af[FM][2], bf[NSET][FN][2];
        f32x16 acc[NACC][FM][FN];
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;
        auto load_b = [&](int gstep, int par) {
            const int so = ((gstep * (FN * 2048) + wave * 65536) & ((2 << 20) - 1)) & ~1023;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bf[par][j][pl] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rb, (j * 2 + pl) * 1024 + lane * 16, so, 0));
        };
        auto read_a = [&](int buf, int step) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int slot = (((step * FM + i) * 2 + pl) * 4 + wave) * 1024 % PB;
                    af[i][pl] = *reinterpret_cast<const h8*>(lds + buf * PB + slot + lane * 16);
                }
        };
#pragma unroll
        for (int t = 0; t < AHEAD; ++t) load_b(t, t);
        int g = 0;
        for (int c = 0; c < chunks; ++c) {
#pragma unroll
            for (int st = 0; st < STEPS; ++st, ++g) {
                load_b(g + AHEAD, (st + AHEAD) % NSET);
                read_a(c & 1, st);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PA[3] = {1, 0, 0}, PBt[3] = {0, 1, 0};
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[st % NACC][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][PA[t]], bf[st % NSET][j][PBt[t]], acc[st % NACC][i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += acc[q][i][j][r];
    } else {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(act), 0, 64 << 20, 0x00020000);
        const int lt = tid - 256;
        u32x4 v[GLD > 0 ? GLD : 1];
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = 1.f + i + lane;
        unsigned u = 0;
        for (int c = 0; c < chunks; ++c) {
            const int so = (int)(((unsigned)(blockIdx.x * 977 + c) * 40960u) & ((64u << 20) - 1)) & ~4095;
#pragma unroll
            for (int i = 0; i < GLD; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, (i * 256 + lt) * 16, so, 0);
#pragma unroll
            for (int i = 0; i < VAL; ++i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(u) : "v"(f[i & 7]), "v"(f[(i + 3) & 7]));
#pragma unroll
            for (int i = 0; i < LWR; ++i) {
                u32x4 w = v[GLD > 0 ? i % GLD : 0];
                w.x ^= u;
                *reinterpret_cast<u32x4*>(lds + ((c + 1) & 1) * PB + ((i * 256 + lt) * 16) % PB) = w;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        s = __uint_as_float(u);
    }
    out[blockIdx.x * 512 + tid] = s;
}

template <int FM, int FN, int STEPS, int NACC, int GLD, int VAL, int LWR, int PLANE_KB, int AHEAD>
static double run_feed(const char* what, const uint8_t* w, const uint8_t* act, float* out, double bare_tf, double mfma_per_out_rel) {
    const int blocks = 256, chunks = 3 * 4096 / STEPS;
    auto kern = feed_kernel<FM, FN, STEPS, NACC, GLD, VAL, LWR, PLANE_KB, AHEAD>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PLANE_KB * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 2 * PLANE_KB * 1024, 0, w, act, out, chunks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double flops = 2.0 * 32 * 32 * 16 * 3.0 * FM * FN * STEPS * (double)chunks * 4.0 * blocks;
    const double tf = flops / ms * 1e-9;
    printf("%-100s %7.1f TFLOP/s executed = %4.1f %% of the bare loop; %4.2f KB operands / MFMA; %.2f x fewer MFMAs per output than direct\n",
           what, tf, 100.0 * tf / bare_tf, (FM + FN) * 2.0 / (3.0 * FM * FN), mfma_per_out_rel);
    return tf;
}

template <int NPR>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ ops, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    uint4 a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { a[i] = ops[i * 64 + lane]; b[i] = ops[(i + 6) * 64 + lane]; }
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NPR; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[(t + q) % 6]), __builtin_bit_cast(h8, b[(t * 2 + q) % 6]), acc[q], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------ host helpers
static uint64_t rng_s = 88172645463325252ull;
static double urand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (double)(rng_s >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand(), v = urand(); if (u < 1e-300) u = 1e-300; return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }
static double rne_f16(double x) {
    if (x == 0.0 || !isfinite(x)) return x;
    int e;
    frexp(fabs(x), &e);
    int q = e - 11;
    if (q < -24) q = -24;
    const double r = ldexp(nearbyint(ldexp(x, -q)), q);
    return fabs(r) > 65504.0 ? copysign(INFINITY, x) : r;
}
static uint16_t f16_bits(double v) {           // v is on the fp16 grid
    if (v == 0.0) return signbit(v) ? 0x8000 : 0;
    const uint16_t s = v < 0 ? 0x8000 : 0;
    const double a = fabs(v);
    int e;
    const double f = frexp(a, &e);             // a = f 2^e, f in [0.5, 1)
    if (e - 1 < -14) return s | (uint16_t)nearbyint(ldexp(a, 24));             // subnormal
    return s | (uint16_t)(((e - 1 + 15) << 10) | ((int)nearbyint(ldexp(f, 11)) - 1024));
}
static float pow2_scale_for(float amax, int lower_bits) {
    if (!(amax > 0.f) || !isfinite(amax)) return 1.f;
    int e;
    frexpf(amax, &e);
    int k = 15 - e - lower_bits;
    if (k > 100) k = 100;
    if (k < -100) k = -100;
    return ldexpf(1.f, k);
}
// rows of W (fp64) -> per-row power-of-two scale, two fp16 planes of the scaled value
static void split_rows(const std::vector<double>& W, int N, int K, std::vector<uint16_t>& h, std::vector<uint16_t>& l, std::vector<float>& sw) {
    h.resize((size_t)N * K); l.resize((size_t)N * K); sw.resize(N);
    for (int n = 0; n < N; ++n) {
        double mx = 0;
        for (int k = 0; k < K; ++k) mx = fmax(mx, fabs(W[(size_t)n * K + k]));
        sw[n] = pow2_scale_for((float)mx, 0);
        for (int k = 0; k < K; ++k) {
            const double v = W[(size_t)n * K + k] * (double)sw[n];
            const double hh = rne_f16(v), ll = rne_f16(v - hh);
            h[(size_t)n * K + k] = f16_bits(hh);
            l[(size_t)n * K + k] = f16_bits(ll);
        }
    }
}

struct Dev {
    float *A = 0, *W = 0, *C = 0, *sw = 0; uint16_t *Wh = 0, *Wl = 0;
    size_t ca = 0, cw = 0, cc = 0;
    void need(size_t a, size_t w, size_t c) {
        if (a > ca) { if (A) CK(hipFree(A)); CK(hipMalloc(&A, a * 4)); ca = a; }
        if (w > cw) { if (W) { CK(hipFree(W)); CK(hipFree(Wh)); CK(hipFree(Wl)); } CK(hipMalloc(&W, w * 4)); CK(hipMalloc(&Wh, w * 2)); CK(hipMalloc(&Wl, w * 2)); cw = w; }
        if (c > cc) { if (C) CK(hipFree(C)); CK(hipMalloc(&C, c * 4)); cc = c; }
        if (!sw) CK(hipMalloc(&sw, 4096 * 4));
    }
};
// C = A W^T on the device.  mode 0: fp32 chain with W rounded to fp32; mode 1: fp16 scheme with W's planes made from the fp64 rows
static void gemm(Dev& d, int mode, const std::vector<float>& A, const std::vector<double>& W, int M, int N, int K, float sa, std::vector<float>& C) {
    d.need(A.size(), W.size(), (size_t)M * N);
    CK(hipMemcpy(d.A, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    if (mode == 0) {
        std::vector<float> wf(W.size());
        for (size_t i = 0; i < W.size(); ++i) wf[i] = (float)W[i];
        CK(hipMemcpy(d.W, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
        gemm_kernel<0><<<(M / 32) * (N / 32), 64>>>(d.A, d.W, d.Wh, d.Wl, d.C, M, N, K, sa, d.sw);
    } else {
        std::vector<uint16_t> h, l; std::vector<float> sw;
        split_rows(W, N, K, h, l, sw);
        CK(hipMemcpy(d.Wh, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d.Wl, l.data(), l.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d.sw, sw.data(), N * 4, hipMemcpyHostToDevice));
        gemm_kernel<1><<<(M / 32) * (N / 32), 64>>>(d.A, d.W, d.Wh, d.Wl, d.C, M, N, K, sa, d.sw);
    }
    CK(hipDeviceSynchronize());
    C.resize((size_t)M * N);
    CK(hipMemcpy(C.data(), d.C, C.size() * 4, hipMemcpyDeviceToHost));
}

int main() {
    Dev dev;
    // ---------------------------------------------------------------- 1. accuracy
    struct Dist { const char* name; int relu; double a_scale, tail, w_scale; };
    const Dist dists[] = {
        {"N(0,1) ReLU activations, N(0,1/K) weights", 1, 1.0, 0.0, 1.0},
        {"activations x 1e-3", 1, 1e-3, 0.0, 1.0},
        {"activations x 1e+3, weights x 1e-2", 1, 1e3, 0.0, 1e-2},
        {"heavy-tailed activations: N(0,1) exp(2 N(0,1))", 0, 1.0, 2.0, 1.0},
        {"very heavy tail: N(0,1) exp(4 N(0,1)), weights heavy too", 0, 1.0, 4.0, -2.0},
        {"activations x 1e-6", 1, 1e-6, 0.0, 1.0},
    };
    const int H = 16, Wd = 16, NB = 4, N = 64;
    const int cins[3] = {64, 128, 512};
    double worst_ratio[3] = {0, 0, 0};   // f16 direct, Winograd 2-D f16, Winograd 1-D f16: worst rms / rms(fp32 chain)
    for (int cin : cins) {
        const int K = 9 * cin, M = NB * H * Wd, T = M / 4, P = M / 2;
        for (auto& ds : dists) {
            // input NHWC with zero padding handled by index checks; weights g[n][ky][kx][c]
            std::vector<float> x((size_t)NB * H * Wd * cin), g((size_t)N * 9 * cin);
            float amax = 0.f;
            for (auto& v : x) {
                double t = nrand();
                if (ds.relu && t < 0) t = 0;
                if (ds.tail > 0) t *= exp(ds.tail * nrand());
                v = (float)(t * ds.a_scale);
                amax = fmaxf(amax, fabsf(v));
            }
            for (auto& v : g) {
                double t = nrand() / sqrt((double)K);
                if (ds.w_scale < 0) t *= exp(-ds.w_scale * nrand()); else t *= ds.w_scale;
                v = (float)t;
            }
            auto X = [&](int b, int y, int xx, int c) -> float { return (y < 0 || y >= H || xx < 0 || xx >= Wd) ? 0.f : x[(((size_t)b * H + y) * Wd + xx) * cin + c]; };
            // fp64 reference, rows m = (b, y, x)
            std::vector<double> ref((size_t)M * N);
            std::vector<float> A((size_t)M * K);
            for (int b = 0; b < NB; ++b) for (int y = 0; y < H; ++y) for (int xx = 0; xx < Wd; ++xx) {
                const size_t m = ((size_t)b * H + y) * Wd + xx;
                for (int t = 0; t < 9; ++t) for (int c = 0; c < cin; ++c) A[m * K + t * cin + c] = X(b, y + t / 3 - 1, xx + t % 3 - 1, c);
            }
            for (size_t m = 0; m < (size_t)M; ++m) for (int n = 0; n < N; ++n) {
                double s = 0; const float* a = &A[m * K]; const float* w = &g[(size_t)n * K];
                for (int k = 0; k < K; ++k) s += (double)a[k] * (double)w[k];
                ref[m * N + n] = s;
            }
            double rr = 0; for (double v : ref) rr += v * v;
            const double ref_rms = sqrt(rr / ref.size());
            auto err = [&](const std::vector<float>& Y, double& mx) {
                double se = 0; mx = 0;
                for (size_t i = 0; i < Y.size(); ++i) { const double e = fabs((double)Y[i] - ref[i]); se += e * e; if (e > mx) mx = e; }
                mx /= ref_rms;
                return sqrt(se / Y.size()) / ref_rms;
            };
            std::vector<double> Wd64(g.begin(), g.end());
            std::vector<float> Y;
            double mx, rms[6];
            printf("cin %d (K %d), %d pixels, %d channels  %s\n", cin, K, M, N, ds.name);
            gemm(dev, 0, A, Wd64, M, N, K, 1.f, Y); rms[0] = err(Y, mx);
            printf("   %-58s: max err %.3e  rms err %.3e\n", "fp32 fma chain, direct", mx, rms[0]);
            gemm(dev, 1, A, Wd64, M, N, K, pow2_scale_for(amax, 0), Y); rms[1] = err(Y, mx);
            printf("   %-58s: max err %.3e  rms err %.3e  (%.2f x fp32)\n", "f16 x2 planes, 3 products, direct (shipped)", mx, rms[1], rms[1] / rms[0]);
            // ---- Winograd F(2x2,3x3)
            {
                const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
                std::vector<std::vector<double>> U(16, std::vector<double>((size_t)N * cin));
                for (int n = 0; n < N; ++n) for (int c = 0; c < cin; ++c) {
                    double gg[3][3], t[4][3];
                    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gg[i][j] = g[(size_t)n * K + (i * 3 + j) * cin + c];
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * gg[0][j] + G[i][1] * gg[1][j] + G[i][2] * gg[2][j];
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) U[i * 4 + j][(size_t)n * cin + c] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                }
                // input transform in fp32: tile (b, ty, tx) covers outputs (2ty..2ty+1, 2tx..2tx+1), input rows 2ty-1 .. 2ty+2
                std::vector<std::vector<float>> V(16, std::vector<float>((size_t)T * cin));
                for (int b = 0; b < NB; ++b) for (int ty = 0; ty < H / 2; ++ty) for (int tx = 0; tx < Wd / 2; ++tx) {
                    const size_t t = ((size_t)b * (H / 2) + ty) * (Wd / 2) + tx;
                    for (int c = 0; c < cin; ++c) {
                        float d[4][4], r[4][4];
                        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) d[i][j] = X(b, 2 * ty - 1 + i, 2 * tx - 1 + j, c);
                        for (int j = 0; j < 4; ++j) { r[0][j] = d[0][j] - d[2][j]; r[1][j] = d[1][j] + d[2][j]; r[2][j] = d[2][j] - d[1][j]; r[3][j] = d[1][j] - d[3][j]; }
                        for (int i = 0; i < 4; ++i) {
                            V[i * 4 + 0][t * cin + c] = r[i][0] - r[i][2]; V[i * 4 + 1][t * cin + c] = r[i][1] + r[i][2];
                            V[i * 4 + 2][t * cin + c] = r[i][2] - r[i][1]; V[i * 4 + 3][t * cin + c] = r[i][1] - r[i][3];
                        }
                    }
                }
                for (int mode = 1; mode >= 0; --mode) {
                    std::vector<std::vector<float>> Mp(16);
                    for (int p = 0; p < 16; ++p) gemm(dev, mode, V[p], U[p], T, N, cin, pow2_scale_for(amax, 2), Mp[p]);
                    Y.assign((size_t)M * N, 0.f);
                    for (int b = 0; b < NB; ++b) for (int ty = 0; ty < H / 2; ++ty) for (int tx = 0; tx < Wd / 2; ++tx) {
                        const size_t t = ((size_t)b * (H / 2) + ty) * (Wd / 2) + tx;
                        for (int n = 0; n < N; ++n) {
                            float m[4][4], q[2][4];
                            for (int p = 0; p < 16; ++p) m[p / 4][p % 4] = Mp[p][t * N + n];
                            for (int j = 0; j < 4; ++j) { q[0][j] = (m[0][j] + m[1][j]) + m[2][j]; q[1][j] = (m[1][j] - m[2][j]) - m[3][j]; }
                            for (int i = 0; i < 2; ++i) {
                                const size_t m0 = ((size_t)b * H + 2 * ty + i) * Wd + 2 * tx;
                                Y[m0 * N + n] = (q[i][0] + q[i][1]) + q[i][2];
                                Y[(m0 + 1) * N + n] = (q[i][1] - q[i][2]) - q[i][3];
                            }
                        }
                    }
                    rms[mode ? 2 : 3] = err(Y, mx);
                    printf("   %-58s: max err %.3e  rms err %.3e  (%.2f x fp32)\n",
                           mode ? "Winograd F(2x2,3x3), f16 x2 planes, 3 products" : "Winograd F(2x2,3x3), fp32 fma chains per position", mx, rms[mode ? 2 : 3], rms[mode ? 2 : 3] / rms[0]);
                }
            }
            // ---- Winograd F(2,3) along W: positions j = 0..3, K = 3 cin (kernel rows concatenated)
            {
                const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
                std::vector<std::vector<double>> U(4, std::vector<double>((size_t)N * 3 * cin));
                for (int n = 0; n < N; ++n) for (int ky = 0; ky < 3; ++ky) for (int c = 0; c < cin; ++c)
                    for (int j = 0; j < 4; ++j) {
                        double s = 0;
                        for (int kx = 0; kx < 3; ++kx) s += G[j][kx] * g[(size_t)n * K + (ky * 3 + kx) * cin + c];
                        U[j][(size_t)n * 3 * cin + ky * cin + c] = s;
                    }
                std::vector<std::vector<float>> V(4, std::vector<float>((size_t)P * 3 * cin));
                for (int b = 0; b < NB; ++b) for (int y = 0; y < H; ++y) for (int px = 0; px < Wd / 2; ++px) {
                    const size_t p = ((size_t)b * H + y) * (Wd / 2) + px;
                    for (int ky = 0; ky < 3; ++ky) for (int c = 0; c < cin; ++c) {
                        float d[4];
                        for (int j = 0; j < 4; ++j) d[j] = X(b, y + ky - 1, 2 * px - 1 + j, c);
                        const size_t o = p * 3 * cin + ky * cin + c;
                        V[0][o] = d[0] - d[2]; V[1][o] = d[1] + d[2]; V[2][o] = d[2] - d[1]; V[3][o] = d[1] - d[3];
                    }
                }
                for (int mode = 1; mode >= 0; --mode) {
                    std::vector<std::vector<float>> Mp(4);
                    for (int j = 0; j < 4; ++j) gemm(dev, mode, V[j], U[j], P, N, 3 * cin, pow2_scale_for(amax, 1), Mp[j]);
                    Y.assign((size_t)M * N, 0.f);
                    for (size_t p = 0; p < (size_t)P; ++p) for (int n = 0; n < N; ++n) {
                        const float m0 = Mp[0][p * N + n], m1 = Mp[1][p * N + n], m2 = Mp[2][p * N + n], m3 = Mp[3][p * N + n];
                        Y[(2 * p) * N + n] = (m0 + m1) + m2;
                        Y[(2 * p + 1) * N + n] = (m1 - m2) - m3;
                    }
                    rms[mode ? 4 : 5] = err(Y, mx);
                    printf("   %-58s: max err %.3e  rms err %.3e  (%.2f x fp32)\n",
                           mode ? "Winograd F(2,3) along W, f16 x2 planes, 3 products" : "Winograd F(2,3) along W, fp32 fma chains per position", mx, rms[mode ? 4 : 5], rms[mode ? 4 : 5] / rms[0]);
                }
            }
            const int ci = cin == 64 ? 0 : (cin == 128 ? 1 : 2);
            (void)ci;
            worst_ratio[0] = fmax(worst_ratio[0], rms[1] / rms[0]);
            worst_ratio[1] = fmax(worst_ratio[1], rms[2] / rms[0]);
            worst_ratio[2] = fmax(worst_ratio[2], rms[4] / rms[0]);
        }
    }
    printf("ACCURACY GATE (rms error <= 2 x the fp32 fma chain's, worst of 18 cases): direct f16 scheme %.2f x | Winograd F(2x2,3x3) %.2f x -> %s | F(2,3) 1-D %.2f x -> %s\n",
           worst_ratio[0], worst_ratio[1], worst_ratio[1] <= 2.0 ? "GO" : "NO-GO", worst_ratio[2], worst_ratio[2] <= 2.0 ? "GO" : "NO-GO");

    // ---------------------------------------------------------------- 2. feed probe
    {
        std::vector<uint16_t> h(12 * 64 * 8);
        for (auto& v : h) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; v = (uint16_t)((rng_s >> 33) & 0x9fff); }
        uint4* d; float* o;
        CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, 4096 * 512 * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = 20000;
        mfma_loop<3><<<256, 256>>>(d, o, 1000);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) mfma_loop<3><<<256, 256>>>(d, o, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bare = 5.0 * 256 * 4.0 * iters * 12.0 * 32.0 * 32.0 * 16.0 * 2.0 / (ms * 1e-3) / 1e12;
        printf("bare f16 MFMA loop (register operands, one wave per SIMD): %.0f TFLOP/s\n", bare);
        uint8_t *w, *act;
        CK(hipMalloc(&w, 2 << 20)); CK(hipMalloc(&act, 64 << 20));
        std::vector<uint16_t> wi((2 << 20) / 2);
        for (size_t i = 0; i < wi.size(); ++i) wi[i] = 0x3c00 ^ (uint16_t)((i * 2654435761u) >> 22);
        CK(hipMemcpy(w, wi.data(), 2 << 20, hipMemcpyHostToDevice));
        CK(hipMemset(act, 0x3c, 64 << 20));
        // loader figures per chunk and lane (256 loading lanes): GLD = 16-byte loads, VAL = convert-class VALU, LWR = 16-byte LDS writes
        //   direct 256-px tile, 16 channels: 264 x 16 x 4 B = 17 KB -> 4 loads; 2 instr / value -> 33; 2 planes x 8.5 KB -> 4 writes
        const double t0 = run_feed<4, 2, 9, 1, 4, 34, 4, 20, 2>("direct, wave tile 128 x 64 (FN 2), 9 taps / chunk, weights 2 taps ahead, tile 256 px x 128 ch", w, act, o, bare, 1.0);
        const double t1 = run_feed<4, 1, 9, 1, 4, 34, 4, 20, 2>("direct, wave tile 128 x 32 (FN 1), 9 taps / chunk, weights 2 taps ahead, tile 256 px x 64 ch", w, act, o, bare, 1.0);
        //   F(2,3) 1-D: tile 256 px = 128 pairs x 64 ch; planes 2x: 4 positions x 128 pairs x 3.. rows share planes: ~40 KB / chunk
        const double t2 = run_feed<2, 1, 12, 4, 4, 100, 8, 40, 3>("F(2,3) along W: 64 pairs x 32 ch x 4 positions / wave, 12 steps / chunk, weights 3 steps ahead", w, act, o, bare, 1.5);
        //   F(2x2,3x3), wave = 4 positions: tile 32 tiles (128 px) x 64 ch: input 4 rows x 66 px x 16 ch x 4 B = 17 KB -> 4 loads;
        //   8192 transformed values / 256 lanes = 32 per lane x (2 adds + 2 split) = 128 VALU; planes 32 KB -> 8 writes
        const double t3 = run_feed<1, 2, 4, 4, 4, 128, 8, 32, 3>("F(2x2,3x3): 4 positions / wave, 32 tiles x 64 ch, 4 steps / chunk, weights 3 steps ahead (64 registers)", w, act, o, bare, 2.25);
        const double t4 = run_feed<2, 1, 4, 4, 8, 256, 16, 64, 3>("F(2x2,3x3): 4 positions / wave, 64 tiles x 32 ch, 4 steps / chunk (64-KB planes: no room for staging)", w, act, o, bare, 2.25);
        const double t5 = run_feed<1, 2, 4, 4, 0, 0, 0, 32, 3>("F(2x2,3x3): 32 tiles x 64 ch, loading waves idle (barriers only)", w, act, o, bare, 2.25);
        const double t7 = run_feed<1, 2, 4, 4, 4, 128, 8, 32, 1>("F(2x2,3x3): 32 tiles x 64 ch, weights ONE step ahead (16 registers)", w, act, o, bare, 2.25);
        (void)t7;
        printf("STRUCTURE GATE: rate per output relative to the direct 128 x 64 K loop:  FN 1 direct %.2f | F(2,3) 1-D %.2f | F(2x2,3x3) 32x64 %.2f | 64x32 %.2f | 32x64 idle loaders %.2f\n",
               t1 / t0, 1.5 * t2 / t0, 2.25 * t3 / t0, 2.25 * t4 / t0, 2.25 * t5 / t0);
    }
    return 0;
}
