// Prototype: fp32 GEMM C[M][N] = A[M][K] * W[N][K]^T on the bf16 matrix cores with fp32-equivalent accuracy.
// Every fp32 operand is the exact sum of three bf16 numbers (hi, mid, lo: 8 significant bits each); of the nine partial
// products the six of order <= 2 are kept (error of the dropped ones ~2^-26 relative, below fp32 rounding), each exact in
// fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  A is split on the fly while it is staged (VALU work that hides
// under the MFMAs), W is split once on the host into the LDS image the fragments are read from.
//   build:  hipcc --offload-arch=gfx950 -O3 -o gemm_bf16x6 gemm_bf16x6.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int UNIT = 1024;                 // bytes of one (k16 step, fragment, plane): 64 lanes x 16 B
constexpr int OPB = 2 * 4 * 3 * UNIT;      // one operand tile of a stage: 24 KiB
constexpr int STAGE = 2 * OPB;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float hi_of(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ float lo_of(unsigned p) { return __uint_as_float(p << 16); }

// 8 fp32 -> three packed-bf16 quads (element 2p in the low half of dword p)
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& m, u32x4& l) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned hp = cvt_pk(a, b);
        const float ra = a - lo_of(hp), rb = b - hi_of(hp);
        const unsigned mp = cvt_pk(ra, rb);
        const float sa = ra - lo_of(mp), sb = rb - hi_of(mp);
        h[p] = hp;
        m[p] = mp;
        l[p] = cvt_pk(sa, sb);
    }
}

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// KS = k16 steps per LDS stage (1: 24 KiB per stage, 3 workgroups per CU; 2: 48 KiB, 1 per CU)
template <int TERMS, int KS, int PAD = 0>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const uint8_t* __restrict__ Wimg, float* __restrict__ C, int M, int N, int K) {
    constexpr int OP = KS * 4 * 3 * UNIT;   // bytes of one operand tile per stage
    constexpr int ST = 2 * OP;
    constexpr int BKS = 16 * KS;
    constexpr int NU = KS;                  // A staging units per thread (a unit = 8 consecutive k of one row)
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * ST + PAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = N / BN;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int ktiles = K / BKS;
    if (PAD && K < 0) lds[2 * ST + tid] = 1;  // keep the padding allocated

    const float* ap[NU];
    int awo[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = tid + 256 * i, r = u / (2 * KS), g = u % (2 * KS);
        ap[i] = A + (int64_t)(tm * BM + r) * K + g * 8;
        awo[i] = (((g >> 1) * 4 + (r >> 5)) * 3) * UNIT + ((r & 31) + 32 * (g & 1)) * 16;
    }
    const uint8_t* wb = Wimg + (int64_t)tn * ktiles * OP + tid * 16;

    f32x4 xr[2][NU][2];
    auto load_a = [&](int kt, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            xr[S][i][0] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BKS);
            xr[S][i][1] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BKS + 4);
        }
    };
    auto dma_b = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < OP / 4096; ++i) glds16(wb + (int64_t)kt * OP + i * 4096, lds + buf * ST + OP + i * 4096 + wave * 1024);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    // one stage: MFMAs of chunk kt from buffer BUF; behind them, in slices pinned between the matrix instructions, the split of
    // chunk kt+1 (registers of set BUF^1, loaded one iteration ago) and its ds_writes into the other buffer
    auto stage = [&](int kt, auto buf_tag, bool more) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr int NX = BUF ^ 1;
        const uint8_t* st = lds + BUF * ST;
        uint8_t* nx = lds + NX * ST;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    af[i][p] = *reinterpret_cast<const bf16x8*>(st + ((s * 4 + wm * 2 + i) * 3 + p) * UNIT + lane * 16);
                    bf[i][p] = *reinterpret_cast<const bf16x8*>(st + OP + ((s * 4 + wn * 2 + i) * 3 + p) * UNIT + lane * 16);
                }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            const float xs[8] = {xr[NX][s][0].x, xr[NX][s][0].y, xr[NX][s][0].z, xr[NX][s][0].w, xr[NX][s][1].x, xr[NX][s][1].y, xr[NX][s][1].z, xr[NX][s][1].w};
            u32x4 h, m, l;
            float ra[4], rb[4];
            int n = 0;
#pragma unroll
            for (int t = 6 - TERMS; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j, ++n) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
                        if (more) {
                            if (n >= 1 && n <= 4) {  // hi parts and first residuals of pair n-1
                                const int p = n - 1;
                                h[p] = cvt_pk(xs[2 * p], xs[2 * p + 1]);
                                ra[p] = xs[2 * p] - lo_of(h[p]);
                                rb[p] = xs[2 * p + 1] - hi_of(h[p]);
                                __builtin_amdgcn_sched_barrier(0);
                            } else if (n >= 5 && n <= 8) {  // mid and lo parts
                                const int p = n - 5;
                                m[p] = cvt_pk(ra[p], rb[p]);
                                l[p] = cvt_pk(ra[p] - lo_of(m[p]), rb[p] - hi_of(m[p]));
                                __builtin_amdgcn_sched_barrier(0);
                            } else if (n == 9) {
                                *reinterpret_cast<u32x4*>(nx + awo[s]) = h;
                                *reinterpret_cast<u32x4*>(nx + awo[s] + UNIT) = m;
                                *reinterpret_cast<u32x4*>(nx + awo[s] + 2 * UNIT) = l;
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
        }
    };

    // prologue: chunk 0 staged synchronously, chunk 1's A rows already on their way
    load_a(0, S1{});
    dma_b(0, 0);
    {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 h, m, l;
            split8(xr[1][s][0], xr[1][s][1], h, m, l);
            *reinterpret_cast<u32x4*>(lds + awo[s]) = h;
            *reinterpret_cast<u32x4*>(lds + awo[s] + UNIT) = m;
            *reinterpret_cast<u32x4*>(lds + awo[s] + 2 * UNIT) = l;
        }
    }
    if (ktiles > 1) load_a(1, S1{});
    __syncthreads();

    int kt = 0;
    for (; kt + 1 < ktiles; kt += 2) {
        // even chunk: buffer 0; chunk kt+1 is in register set 1 -> buffer 1; chunk kt+2 goes to set 0
        dma_b(kt + 1, 1);
        if (kt + 2 < ktiles) load_a(kt + 2, S0{});
        stage(kt, S0{}, true);
        __syncthreads();
        const bool more = kt + 2 < ktiles;
        if (more) dma_b(kt + 2, 0);
        if (kt + 3 < ktiles) load_a(kt + 3, S1{});
        stage(kt + 1, S1{}, more);
        __syncthreads();
    }
    if (kt < ktiles) stage(kt, S0{}, false);

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = tn * BN + wn * 64 + j * 32 + (lane & 31);
                C[(int64_t)row * N + col] = acc[i][j][r];
            }
}

// ---- variant: v_mfma_f32_16x16x32_bf16 (the guide measures it at a higher held clock than 32x32x16), one K32 chunk per stage.
// Fragment = 16 rows; lane l holds row l & 15, k = 8 (l >> 4) + j.  Tile 128 x TN, waves 2 x 2 (wave tile 64 x TN/2).
// TN = 64: 72 KiB of LDS (two workgroups per CU); TN = 128: 96 KiB (one).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int TN>
__global__ __launch_bounds__(256) void gemm16_kernel(const float* __restrict__ A, const uint8_t* __restrict__ Wimg, float* __restrict__ C, int M, int N, int K) {
    constexpr int NFA = BM / 16, NFB = TN / 16;
    constexpr int A_B = NFA * 3 * 1024, B_B = NFB * 3 * 1024, ST = A_B + B_B;
    constexpr int FM = 4, FN = TN / 32;
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * ST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = N / TN;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int ktiles = K / 32;
    const float* ap0 = A + (int64_t)(tm * BM + (tid >> 2)) * K + (tid & 3) * 8;          // unit 0: row tid/4, k8 group tid%4
    const float* ap1 = ap0 + (int64_t)64 * K;                                            // unit 1: row + 64
    const int awo0 = (((tid >> 2) >> 4) * 3) * 1024 + (((tid >> 2) & 15) + 16 * (tid & 3)) * 16;
    const int awo1 = awo0 + 4 * 3 * 1024;
    const uint8_t* wb = Wimg + (int64_t)tn * ktiles * B_B + tid * 16;
    f32x4 x0[2][2], x1[2][2];   // [set][half]
    auto load_a = [&](int kt, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        x0[S][0] = *reinterpret_cast<const f32x4*>(ap0 + kt * 32);
        x0[S][1] = *reinterpret_cast<const f32x4*>(ap0 + kt * 32 + 4);
        x1[S][0] = *reinterpret_cast<const f32x4*>(ap1 + kt * 32);
        x1[S][1] = *reinterpret_cast<const f32x4*>(ap1 + kt * 32 + 4);
    };
    auto dma_b = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < B_B / 4096; ++i) glds16(wb + (int64_t)kt * B_B + i * 4096, lds + buf * ST + A_B + i * 4096 + wave * 1024);
    };
    f32x4v acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto stage = [&](auto buf_tag, auto more_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr int NX = BUF ^ 1;
        constexpr bool MORE = decltype(more_tag)::value;
        const uint8_t* st = lds + BUF * ST;
        uint8_t* nx = lds + NX * ST;
        bf16x8 af[FM][3], bf[FN][3];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(st + ((wm * FM + i) * 3 + p) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(st + A_B + ((wn * FN + j) * 3 + p) * 1024 + lane * 16);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        constexpr int NM = 6 * FM * FN;
        u32x4 h0, m0, l0, h1, m1, l1;
        int n = 0;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j, ++n) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
                    if constexpr (MORE) {
                        if (n == NM / 4) {
                            split8(x0[NX][0], x0[NX][1], h0, m0, l0);
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (n == NM / 4 + 2) {
                            *reinterpret_cast<u32x4*>(nx + awo0) = h0;
                            *reinterpret_cast<u32x4*>(nx + awo0 + 1024) = m0;
                            *reinterpret_cast<u32x4*>(nx + awo0 + 2048) = l0;
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (n == NM / 2) {
                            split8(x1[NX][0], x1[NX][1], h1, m1, l1);
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (n == NM / 2 + 2) {
                            *reinterpret_cast<u32x4*>(nx + awo1) = h1;
                            *reinterpret_cast<u32x4*>(nx + awo1 + 1024) = m1;
                            *reinterpret_cast<u32x4*>(nx + awo1 + 2048) = l1;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
    };
    load_a(0, S1{});
    dma_b(0, 0);
    {
        u32x4 h, m, l;
        split8(x0[1][0], x0[1][1], h, m, l);
        *reinterpret_cast<u32x4*>(lds + awo0) = h; *reinterpret_cast<u32x4*>(lds + awo0 + 1024) = m; *reinterpret_cast<u32x4*>(lds + awo0 + 2048) = l;
        split8(x1[1][0], x1[1][1], h, m, l);
        *reinterpret_cast<u32x4*>(lds + awo1) = h; *reinterpret_cast<u32x4*>(lds + awo1 + 1024) = m; *reinterpret_cast<u32x4*>(lds + awo1 + 2048) = l;
    }
    if (ktiles > 1) load_a(1, S1{});
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < ktiles; kt += 2) {
        dma_b(kt + 1, 1);
        load_a(kt + 2, S0{});
        stage(S0{}, std::true_type{});
        __syncthreads();
        dma_b(kt + 2, 0);
        if (kt + 3 < ktiles) load_a(kt + 3, S1{});
        stage(S1{}, std::true_type{});
        __syncthreads();
    }
    if (kt + 1 < ktiles) {
        dma_b(kt + 1, 1);
        stage(S0{}, std::true_type{});
        __syncthreads();
        stage(S1{}, std::false_type{});
    } else if (kt < ktiles) {
        stage(S0{}, std::false_type{});
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tm * BM + (wm * FM + i) * 16 + (lane >> 4) * 4 + r;
                const int col = tn * TN + (wn * FN + j) * 16 + (lane & 15);
                C[(int64_t)row * N + col] = acc[i][j][r];
            }
}

// ---- larger wave tile: workgroup 256 x 128, wave tile 128 x 64 (4 x 2 fragments): 18 ds_read_b128 per 48 MFMAs instead of 12 per 24.
// In a power-limited loop fewer LDS bytes per MFMA should buy clock.  One k16 step per stage, A split on the fly (2 units per thread).
__global__ __launch_bounds__(256) void gemm_big_kernel(const float* __restrict__ A, const uint8_t* __restrict__ Wimg, float* __restrict__ C, int M, int N, int K) {
    constexpr int BMB = 256, FM = 4, FN = 2;
    constexpr int A_B = 8 * 3 * UNIT, B_B = 4 * 3 * UNIT, ST = A_B + B_B;     // 36 KiB per stage
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * ST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = N / BN;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int ktiles = K / 16;
    const float* ap0 = A + (int64_t)(tm * BMB + (tid >> 1)) * K + (tid & 1) * 8;
    const float* ap1 = ap0 + (int64_t)128 * K;
    const int awo0 = (((tid >> 1) >> 5) * 3) * UNIT + (((tid >> 1) & 31) + 32 * (tid & 1)) * 16;
    const int awo1 = awo0 + 4 * 3 * UNIT;
    const uint8_t* wb = Wimg + (int64_t)tn * ktiles * B_B + tid * 16;
    f32x4 x0[2][2], x1[2][2];
    auto load_a = [&](int kt, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        x0[S][0] = *reinterpret_cast<const f32x4*>(ap0 + kt * 16);
        x0[S][1] = *reinterpret_cast<const f32x4*>(ap0 + kt * 16 + 4);
        x1[S][0] = *reinterpret_cast<const f32x4*>(ap1 + kt * 16);
        x1[S][1] = *reinterpret_cast<const f32x4*>(ap1 + kt * 16 + 4);
    };
    auto dma_b = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < B_B / 4096; ++i) glds16(wb + (int64_t)kt * B_B + i * 4096, lds + buf * ST + A_B + i * 4096 + wave * 1024);
    };
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto stage = [&](auto buf_tag, auto more_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr int NX = BUF ^ 1;
        constexpr bool MORE = decltype(more_tag)::value;
        const uint8_t* st = lds + BUF * ST;
        uint8_t* nx = lds + NX * ST;
        bf16x8 af[FM][3], bf[FN][3];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(st + ((wm * FM + i) * 3 + p) * UNIT + lane * 16);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(st + A_B + ((wn * FN + j) * 3 + p) * UNIT + lane * 16);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        constexpr int NM = 6 * FM * FN;
        u32x4 h0, m0, l0, h1, m1, l1;
        int n = 0;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j, ++n) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
                    if constexpr (MORE) {
                        if (n == 4) { split8(x0[NX][0], x0[NX][1], h0, m0, l0); __builtin_amdgcn_sched_barrier(0); }
                        else if (n == 8) {
                            *reinterpret_cast<u32x4*>(nx + awo0) = h0; *reinterpret_cast<u32x4*>(nx + awo0 + UNIT) = m0; *reinterpret_cast<u32x4*>(nx + awo0 + 2 * UNIT) = l0;
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (n == 16) { split8(x1[NX][0], x1[NX][1], h1, m1, l1); __builtin_amdgcn_sched_barrier(0); }
                        else if (n == 20) {
                            *reinterpret_cast<u32x4*>(nx + awo1) = h1; *reinterpret_cast<u32x4*>(nx + awo1 + UNIT) = m1; *reinterpret_cast<u32x4*>(nx + awo1 + 2 * UNIT) = l1;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
    };
    load_a(0, S1{});
    dma_b(0, 0);
    {
        u32x4 h, m, l;
        split8(x0[1][0], x0[1][1], h, m, l);
        *reinterpret_cast<u32x4*>(lds + awo0) = h; *reinterpret_cast<u32x4*>(lds + awo0 + UNIT) = m; *reinterpret_cast<u32x4*>(lds + awo0 + 2 * UNIT) = l;
        split8(x1[1][0], x1[1][1], h, m, l);
        *reinterpret_cast<u32x4*>(lds + awo1) = h; *reinterpret_cast<u32x4*>(lds + awo1 + UNIT) = m; *reinterpret_cast<u32x4*>(lds + awo1 + 2 * UNIT) = l;
    }
    if (ktiles > 1) load_a(1, S1{});
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < ktiles; kt += 2) {
        dma_b(kt + 1, 1);
        load_a(kt + 2, S0{});
        stage(S0{}, std::true_type{});
        __syncthreads();
        dma_b(kt + 2, 0);
        if (kt + 3 < ktiles) load_a(kt + 3, S1{});
        stage(S1{}, std::true_type{});
        __syncthreads();
    }
    if (kt + 1 < ktiles) {
        dma_b(kt + 1, 1);
        stage(S0{}, std::true_type{});
        __syncthreads();
        stage(S1{}, std::false_type{});
    } else if (kt < ktiles) {
        stage(S0{}, std::false_type{});
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * BMB + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = tn * BN + wn * 64 + j * 32 + (lane & 31);
                C[(int64_t)row * N + col] = acc[i][j][r];
            }
}

// ---- bound: BOTH operands pre-split in fragment order (as if the producer of A had written bf16 planes): LDS-DMA only, no VALU
// in the loop.  What the on-the-fly split costs, and what a pre-splitting epilogue could gain.
__global__ __launch_bounds__(256) void gemm_presplit_kernel(const uint8_t* __restrict__ Aimg, const uint8_t* __restrict__ Wimg, float* __restrict__ C, int M, int N, int K) {
    constexpr int OP = 4 * 3 * UNIT, ST = 2 * OP;
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * ST + 22 * 1024];   // padded to two workgroups per CU like the conv kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = N / BN;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int ktiles = K / 16;
    if (K < 0) lds[2 * ST + tid] = 1;
    const uint8_t* ab = Aimg + (int64_t)tm * ktiles * OP + tid * 16;
    const uint8_t* wb = Wimg + (int64_t)tn * ktiles * OP + tid * 16;
    auto dma = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < OP / 4096; ++i) {
            glds16(ab + (int64_t)kt * OP + i * 4096, lds + buf * ST + i * 4096 + wave * 1024);
            glds16(wb + (int64_t)kt * OP + i * 4096, lds + buf * ST + OP + i * 4096 + wave * 1024);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    dma(0, 0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) dma(kt + 1, cur ^ 1);
        const uint8_t* st = lds + cur * ST;
        bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[i][p] = *reinterpret_cast<const bf16x8*>(st + ((wm * 2 + i) * 3 + p) * UNIT + lane * 16);
                bf[i][p] = *reinterpret_cast<const bf16x8*>(st + OP + ((wn * 2 + i) * 3 + p) * UNIT + lane * 16);
            }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = tn * BN + wn * 64 + j * 32 + (lane & 31);
                C[(int64_t)row * N + col] = acc[i][j][r];
            }
}

// ---- deeper pipeline on the pre-split variant: ring of NB k16 buffers, DMA issued DIST stages ahead, raw barrier that leaves the
// youngest DIST-1 stages of DMA in flight (s_waitcnt vmcnt((DIST-1)*6)).  Tests whether the one-stage-ahead staging (768 MFMA
// cycles of cover against ~2000+ cycles of global->LDS latency) is what keeps the matrix pipe at ~55 % in the K loop.
template <int NB, int DIST>
__global__ __launch_bounds__(256) void gemm_ring_kernel(const uint8_t* __restrict__ Aimg, const uint8_t* __restrict__ Wimg, float* __restrict__ C, int M, int N, int K) {
    constexpr int OP = 4 * 3 * UNIT, ST = 2 * OP;
    __shared__ __attribute__((aligned(16))) uint8_t lds[NB * ST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = N / BN;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int ktiles = K / 16;
    const uint8_t* ab = Aimg + (int64_t)tm * ktiles * OP + tid * 16;
    const uint8_t* wb = Wimg + (int64_t)tn * ktiles * OP + tid * 16;
    auto dma = [&](int kt, int buf) {   // 6 DMA instructions per thread
#pragma unroll
        for (int i = 0; i < OP / 4096; ++i) {
            glds16(ab + (int64_t)kt * OP + i * 4096, lds + buf * ST + i * 4096 + wave * 1024);
            glds16(wb + (int64_t)kt * OP + i * 4096, lds + buf * ST + OP + i * 4096 + wave * 1024);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < ktiles) dma(d, d % NB);
    int cur = 0, nxt = DIST % NB;
    for (int kt = 0; kt < ktiles; ++kt) {
        // stage kt must have landed: everything but the youngest (DIST-1) stages' DMA (fewer near the end: then wait for all)
        if (kt + DIST - 1 < ktiles) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((DIST - 1) * 6) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + DIST < ktiles) dma(kt + DIST, nxt);     // its buffer was read DIST... stages ago (NB > DIST)
        const uint8_t* st = lds + cur * ST;
        bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[i][p] = *reinterpret_cast<const bf16x8*>(st + ((wm * 2 + i) * 3 + p) * UNIT + lane * 16);
                bf[i][p] = *reinterpret_cast<const bf16x8*>(st + OP + ((wn * 2 + i) * 3 + p) * UNIT + lane * 16);
            }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
        cur = (cur + 1 == NB) ? 0 : cur + 1;
        nxt = (nxt + 1 == NB) ? 0 : nxt + 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = tn * BN + wn * 64 + j * 32 + (lane & 31);
                C[(int64_t)row * N + col] = acc[i][j][r];
            }
}

// fp32 FMA-chain reference on the device (what an fp32 MFMA / any fp32 kernel delivers), one thread per output
__global__ void ref_f32_kernel(const float* A, const float* W, float* C, int M, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int row = idx / N, col = idx % N;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(int64_t)row * K + k], W[(int64_t)col * K + k], s);
    C[idx] = s;
}

static uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 131072, N = argc > 2 ? atoi(argv[2]) : 128, K = argc > 3 ? atoi(argv[3]) : 1152;
    int reps = argc > 4 ? atoi(argv[4]) : 20;
    if (M % BM || N % BN || K % BK) { printf("sizes must be multiples of the tile\n"); return 1; }
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 40) / 16777216.0 * 2.0 - 1.0); };
    for (auto& v : hA) v = rnd() * (1.f + 3.f * fabsf(rnd()));
    for (auto& v : hW) v = rnd() * 0.05f;
    const int tilesN = N / BN;
    auto make_image = [&](int KS) {
        const int ktiles = K / (16 * KS), OP = KS * 4 * 3 * UNIT;
        std::vector<uint8_t> img((size_t)tilesN * ktiles * OP);
        for (int tn = 0; tn < tilesN; ++tn)
            for (int kt = 0; kt < ktiles; ++kt)
                for (int sidx = 0; sidx < KS; ++sidx)
                    for (int f = 0; f < 4; ++f)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int col = tn * BN + f * 32 + (l & 31), k = kt * 16 * KS + sidx * 16 + 8 * (l >> 5) + j;
                                const float w = hW[(size_t)col * K + k];
                                const uint16_t h = bf16_rne(w);
                                const float r1 = w - bf16_to_f(h);
                                const uint16_t m = bf16_rne(r1);
                                const float r2 = r1 - bf16_to_f(m);
                                const uint16_t lo = bf16_rne(r2);
                                uint8_t* base = img.data() + ((size_t)tn * ktiles + kt) * OP + ((sidx * 4 + f) * 3) * UNIT + l * 16 + j * 2;
                                memcpy(base, &h, 2);
                                memcpy(base + UNIT, &m, 2);
                                memcpy(base + 2 * UNIT, &lo, 2);
                            }
        return img;
    };
    auto make_image16 = [&](int TN) {
        const int ktiles = K / 32, NFB = TN / 16, B_B = NFB * 3 * 1024, tiles = N / TN;
        std::vector<uint8_t> img((size_t)tiles * ktiles * B_B);
        for (int tn = 0; tn < tiles; ++tn)
            for (int kt = 0; kt < ktiles; ++kt)
                for (int f = 0; f < NFB; ++f)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int col = tn * TN + f * 16 + (l & 15), k = kt * 32 + 8 * (l >> 4) + j;
                            const float w = hW[(size_t)col * K + k];
                            const uint16_t h = bf16_rne(w);
                            const float r1 = w - bf16_to_f(h);
                            const uint16_t m = bf16_rne(r1);
                            const float r2 = r1 - bf16_to_f(m);
                            const uint16_t lo = bf16_rne(r2);
                            uint8_t* base = img.data() + ((size_t)tn * ktiles + kt) * B_B + (f * 3) * 1024 + l * 16 + j * 2;
                            memcpy(base, &h, 2);
                            memcpy(base + 1024, &m, 2);
                            memcpy(base + 2048, &lo, 2);
                        }
        return img;
    };
    std::vector<uint8_t> img16a = make_image16(64), img16b = make_image16(128);
    std::vector<uint8_t> imgA;
    {
        const int ktiles = K / 16, OP = 4 * 3 * UNIT, tilesM = M / BM;
        imgA.resize((size_t)tilesM * ktiles * OP);
        for (int tmi = 0; tmi < tilesM; ++tmi)
            for (int kt = 0; kt < ktiles; ++kt)
                for (int f = 0; f < 4; ++f)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int row = tmi * BM + f * 32 + (l & 31), k = kt * 16 + 8 * (l >> 5) + j;
                            const float w = hA[(size_t)row * K + k];
                            const uint16_t h = bf16_rne(w);
                            const float r1 = w - bf16_to_f(h);
                            const uint16_t m = bf16_rne(r1);
                            const float r2 = r1 - bf16_to_f(m);
                            const uint16_t lo = bf16_rne(r2);
                            uint8_t* base = imgA.data() + ((size_t)tmi * ktiles + kt) * OP + (f * 3) * UNIT + l * 16 + j * 2;
                            memcpy(base, &h, 2);
                            memcpy(base + UNIT, &m, 2);
                            memcpy(base + 2 * UNIT, &lo, 2);
                        }
    }
    std::vector<uint8_t> img1 = make_image(1), img2 = make_image(2);
    float *dA, *dW, *dC, *dR;
    uint8_t *dI1, *dI2;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dW, hW.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dR, (size_t)M * N * 4));
    CK(hipMalloc(&dI1, img1.size()));
    CK(hipMalloc(&dI2, img2.size()));
    uint8_t *dIA, *dJa, *dJb;
    CK(hipMalloc(&dJa, img16a.size()));
    CK(hipMalloc(&dJb, img16b.size()));
    CK(hipMemcpy(dJa, img16a.data(), img16a.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dJb, img16b.data(), img16b.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&dIA, imgA.size()));
    CK(hipMemcpy(dIA, imgA.data(), imgA.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dI1, img1.data(), img1.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dI2, img2.data(), img2.size(), hipMemcpyHostToDevice));
    const int blocks = (M / BM) * tilesN;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> hC((size_t)M * N), hR((size_t)M * N);
    ref_f32_kernel<<<(unsigned)(((int64_t)M * N + 255) / 256), 256>>>(dA, dW, dR, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost));
    // fp64 truth on a sample of rows
    const int nsample = 64;
    std::vector<double> truth((size_t)nsample * N);
    std::vector<int> rows(nsample);
    for (int i = 0; i < nsample; ++i) {
        rows[i] = (int)(((int64_t)i * 2654435761ll) % M);
        for (int c = 0; c < N; ++c) {
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)rows[i] * K + k] * (double)hW[(size_t)c * K + k];
            truth[(size_t)i * N + c] = acc;
        }
    }
    auto err_vs_truth = [&](const std::vector<float>& c, double& maxe, double& rmse) {
        maxe = 0; rmse = 0; double scale = 0;
        for (int i = 0; i < nsample; ++i)
            for (int cc = 0; cc < N; ++cc) {
                const double t = truth[(size_t)i * N + cc], e = fabs((double)c[(size_t)rows[i] * N + cc] - t);
                if (e > maxe) maxe = e;
                rmse += e * e;
                scale += t * t;
            }
        rmse = sqrt(rmse / (nsample * N));
        scale = sqrt(scale / (nsample * N));
        maxe /= scale; rmse /= scale;
    };
    double me, re;
    err_vs_truth(hR, me, re);
    printf("M %d N %d K %d  blocks %d\n", M, N, K, blocks);
    printf("fp32 fma chain      : max err %.3e  rms err %.3e (relative to rms of the result)\n", me, re);
    for (int ks : {3, 32, 77})
        for (int terms : {6}) {
            auto launch = [&]() {
                if (ks == 77) gemm_big_kernel<<<(M / 256) * (N / BN), 256>>>(dA, dI1, dC, M, N, K);   // label 77: workgroup 256x128, wave tile 128x64
                else if (ks == 32) gemm_ring_kernel<3, 2><<<blocks, 256>>>(dIA, dI1, dC, M, N, K);        // label 32: ring of 3 buffers, DMA 2 stages ahead (72 KiB: 2 per CU)
                else if (ks == 43) gemm_ring_kernel<4, 3><<<blocks, 256>>>(dIA, dI1, dC, M, N, K);   // label 43: ring of 4, 3 ahead (96 KiB: 1 per CU)
                else if (ks == 63) gemm_ring_kernel<6, 3><<<blocks, 256>>>(dIA, dI1, dC, M, N, K);   // label 63: ring of 6, 3 ahead (144 KiB: 1 per CU)
                else if (ks == 16) gemm16_kernel<64><<<(M / BM) * (N / 64), 256>>>(dA, dJa, dC, M, N, K);          // label 16: 16x16x32, tile 128x64
                else if (ks == 17) gemm16_kernel<128><<<(M / BM) * (N / 128), 256>>>(dA, dJb, dC, M, N, K);   // label 17: 16x16x32, tile 128x128
                else if (ks == 9) gemm_presplit_kernel<<<blocks, 256>>>(dIA, dI1, dC, M, N, K);   // label 9 = both operands pre-split
                else if (ks == 3) gemm_kernel<6, 1, 22 * 1024><<<blocks, 256>>>(dA, dI1, dC, M, N, K);  // 70 KiB: 2 workgroups per CU
                else if (ks == 1 && terms == 6) gemm_kernel<6, 1><<<blocks, 256>>>(dA, dI1, dC, M, N, K);
                else if (ks == 1) gemm_kernel<3, 1><<<blocks, 256>>>(dA, dI1, dC, M, N, K);
                else if (terms == 6) gemm_kernel<6, 2><<<blocks, 256>>>(dA, dI2, dC, M, N, K);
                else gemm_kernel<3, 2><<<blocks, 256>>>(dA, dI2, dC, M, N, K);
            };
            CK(hipMemset(dC, 0, (size_t)M * N * 4));
            launch();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
            err_vs_truth(hC, me, re);
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
            printf("bf16x%d k16-steps/stage %d : max err %.3e  rms err %.3e   %.3f ms  %.1f TFLOP/s fp32-equivalent (%.0f TF of bf16 MFMA work)\n", terms, ks, me, re, ms,
                   tf, tf * terms);
        }
    return 0;
}
