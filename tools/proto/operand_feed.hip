// How fast the bf16 matrix cores run when their operands are FED the way the tap-sharing conv kernels feed them - and the way a
// larger wave tile would.  Synthetic (random operands, the result is only summed so nothing is eliminated): per "tap" a wave reads
// FM x 3 A fragments from LDS (ds_read_b128, shifted windows of a 50 KiB plane area) and FN x 3 B fragments from a 2 MiB weight image in
// global memory (L2-resident; buffer_load_dwordx4, one tap ahead, two register sets) or from LDS, and issues 6 x FM x FN
// v_mfma_f32_32x32x16_bf16.  Configurations: wave tile 64 x 64 at two workgroups per CU (the shipped kernels), 128 x 64 and 128 x 128 at
// one workgroup per CU (512 registers per lane: accumulators in AGPRs).
//   build: hipcc --offload-arch=gfx950 -O3 -o operand_feed operand_feed.hip ;  run: ./operand_feed
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int PLANE = 264 * 32;          // bytes of one bf16 plane of the halo area (the 128-row tap-sharing tile)
constexpr int ABYTES = 2 * 3 * PLANE;    // two halo buffers
constexpr int WIMG = 2 << 20;            // weight image bytes walked by the B loads

// BSRC: 0 = B from global memory (one tap ahead), 1 = B from LDS, 2 = no B loads (registers only), AOFF: 1 = no A reads
template <int FM, int FN, int OCC, int BSRC, int AOFF>
__global__ __launch_bounds__(256, OCC) void feed_kernel(const uint8_t* __restrict__ wimg, float* __restrict__ out, int taps) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[ABYTES + 24576];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (ABYTES + 24576) / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0x3f803f80u + i, 0x3f003f80u, 0x3e803f80u ^ (unsigned)i, 0x3f803e80u};
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wimg), 0, WIMG, 0x00020000);
    int abase[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) abase[i] = (((wave >> 1) * FM + i) % 4 * 33 + (lane & 31)) * 32 + (lane >> 5) * 16;
    const int bvo = ((wave & 1) * FN) * 3072 + lane * 16;
    bf16x8 af[FM][3], bf[2][FN][3];
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto read_a = [&](int tap, int pl) {
        const int to = ((tap % 9) / 3 * 66 + (tap % 9) % 3) * 32 + ((tap / 9) & 1) * 3 * PLANE;
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i][pl] = *reinterpret_cast<const bf16x8*>(lds + pl * PLANE + abase[i] + to);
    };
    auto load_b = [&](int tap, int par) {
        const int so = ((tap * 18432) & (WIMG - 1)) & ~1023;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                if (BSRC == 0) bf[par][j][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, bvo + j * 3072 + pl * 1024, so, 0));
                else if (BSRC == 1) bf[par][j][pl] = *reinterpret_cast<const bf16x8*>(lds + ABYTES + ((j * 3 + pl) * 1024 + lane * 16 + (tap & 1) * 4096) % 16384);
            }
    };
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[par][j][pl] = *reinterpret_cast<const bf16x8*>(lds + ((par * FN + j) * 3 + pl) * 1024 + lane * 16);
    load_b(0, 0);
    read_a(0, 2); read_a(0, 0); read_a(0, 1);
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    // BSRC == 3: the tap's weights (4 fragments x 3 planes = 12 KiB for the 128 columns of the workgroup) go through LDS: every wave
    // fetches a quarter (3 KiB) two taps ahead, writes it one tap ahead, ONE BARRIER PER TAP hands it over, every wave reads its FN
    // fragments back.  (The real cost of sharing the weights between waves: the synchronisation.)
    u32x4 stage[3];
    auto fetch_q = [&](int tap) {
        const int so = ((tap * 18432) & (WIMG - 1)) & ~1023;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) stage[pl] = __builtin_amdgcn_raw_buffer_load_b128(rb, wave * 3072 + pl * 1024 + lane * 16, so, 0);
    };
    auto write_q = [&](int tap) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(lds + ABYTES + (tap & 1) * 12288 + wave * 3072 + pl * 1024 + lane * 16) = stage[pl];
    };
    auto read_b = [&](int tap, int par) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[par][j][pl] = *reinterpret_cast<const bf16x8*>(lds + ABYTES + (tap & 1) * 12288 + (((wave & 1) * FN + j) % 4) * 3072 + pl * 1024 + lane * 16);
    };
    if (BSRC == 3) { fetch_q(1); }
    for (int tap = 0; tap < taps; tap += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (BSRC == 3) {
                write_q(tap + u + 1);            // the quarter fetched during the previous tap
                fetch_q(tap + u + 2);
            }
            if (BSRC != 2 && BSRC != 3) load_b(tap + u + 1, u ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[u][j][PB[t]], acc[i][j], 0, 0, 0);
                if (!AOFF && (t == 0 || t == 3 || t == 5)) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_a(tap + u + 1, t == 0 ? 2 : (t == 3 ? 1 : 0));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (BSRC == 3 && t == 3) {       // two thirds into the tap: hand the next tap's weights over, read them behind the rest
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    read_b(tap + u + 1, u ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if ((tap % 18) == 16) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the per-chunk barrier
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int FM, int FN, int OCC, int BSRC, int AOFF>
static void run(const char* what, const uint8_t* w, float* out) {
    const int blocks = 256 * OCC, taps = 18 * 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((feed_kernel<FM, FN, OCC, BSRC, AOFF>), dim3(blocks), dim3(256), 0, 0, w, out, taps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 2.0 * 32 * 32 * 16 * 6.0 * FM * FN * taps * 4.0 * blocks;
    printf("%-78s %7.1f TFLOP/s executed (%5.1f fp32-equivalent), %.3f ms\n", what, flops / ms * 1e-9, flops / ms * 1e-9 / 6.0, ms);
}

int main() {
    uint8_t* w; float* out;
    CK(hipMalloc(&w, WIMG)); CK(hipMalloc(&out, 1 << 22));
    std::vector<uint16_t> h(WIMG / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f80 ^ (uint16_t)((i * 2654435761u) >> 23);
    CK(hipMemcpy(w, h.data(), WIMG, hipMemcpyHostToDevice));
    run<2, 2, 2, 0, 0>("wave tile  64 x  64, 2 workgroups / CU, A from LDS, B from L2 (shipped)", w, out);
    run<2, 2, 2, 1, 0>("wave tile  64 x  64, 2 workgroups / CU, A from LDS, B from LDS", w, out);
    run<2, 2, 2, 2, 0>("wave tile  64 x  64, 2 workgroups / CU, A from LDS, no B loads", w, out);
    run<2, 2, 2, 0, 1>("wave tile  64 x  64, 2 workgroups / CU, no A reads, B from L2", w, out);
    run<2, 2, 2, 2, 1>("wave tile  64 x  64, 2 workgroups / CU, registers only", w, out);
    run<4, 2, 1, 0, 0>("wave tile 128 x  64, 1 workgroup  / CU, A from LDS, B from L2", w, out);
    run<4, 2, 1, 1, 0>("wave tile 128 x  64, 1 workgroup  / CU, A from LDS, B from LDS", w, out);
    run<4, 2, 1, 3, 0>("wave tile 128 x  64, 1 workgroup  / CU, A from LDS, B staged through LDS (barrier per tap)", w, out);
    run<2, 2, 2, 3, 0>("wave tile  64 x  64, 2 workgroups / CU, A from LDS, B staged through LDS (barrier per tap)", w, out);
    run<4, 4, 1, 0, 0>("wave tile 128 x 128, 1 workgroup  / CU, A from LDS, B from L2", w, out);
    run<4, 4, 1, 1, 0>("wave tile 128 x 128, 1 workgroup  / CU, A from LDS, B from LDS", w, out);
    run<4, 4, 1, 2, 1>("wave tile 128 x 128, 1 workgroup  / CU, registers only", w, out);
    return 0;
}
