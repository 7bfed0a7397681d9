// Frame pre-processing on the GPU: uint8 HWC camera frames -> the model-contract input.
// Replaces PIL.Image.resize([256,256], BICUBIC) + ToTensor + Normalize of the reference's CPU loader
// (datasets/ego4view_syn/ego4view_syn_pose3d.py:41-44,159-162).
//
// Pillow's resampling is integer arithmetic (22-bit fixed-point weights, int32 accumulation from 2^21, >> 22,
// clip to [0,255]) in two passes, horizontal then vertical, with a uint8 intermediate — restated exactly, so the
// resized uint8 image is bit-identical to Pillow's.  The window bounds / fixed-point weights come from the host
// (egorear_amd/preprocess.py computes them in float64 exactly as Pillow's precompute_coeffs does).
// Both passes are HBM-bound: one read of the 2.28 MB frame, a 0.67 MB uint8 intermediate, one 0.79 MB fp32 write.
#include "egr_common.h"

namespace {

constexpr int PRECISION_BITS = 22;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: (n, h, w, 3) u8 -> (n, h, ow, 3) u8; one thread per (n, y, ox).
// The 3*cnt (<= 48) source bytes of a window are fetched as aligned dwords (neighbouring lanes overlap heavily: L1 hits)
// and unpacked with shifts, instead of 45 byte loads.
constexpr int MAXK = 24;  // taps per output (872 -> 256 needs 15)

__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* src, uint8_t* tmp, int n, int h, int w, int ow,
                                                       const int32_t* bounds, const int32_t* coef, int ksize) {
    // requires ksize <= 16 and (h*w*3) % 4 == 0 (host-checked): 48 window bytes + up to 3 bytes of misalignment = 13 dwords
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * h * ow) return;
    int ox = (int)(idx % ow);
    int64_t row = idx / ow;  // n*h + y
    int xmin = bounds[2 * ox], cnt = bounds[2 * ox + 1];
    const int32_t* k = coef + (int64_t)ox * ksize;
    const int64_t b0 = (row * w + xmin) * 3;          // first source byte
    const int64_t a0 = b0 & ~(int64_t)3;               // aligned start
    const unsigned sh = (unsigned)(b0 - a0);           // 0..3
    const int64_t total = (int64_t)n * h * w * 3;
    const uint32_t* p32 = reinterpret_cast<const uint32_t*>(src + a0);
    uint32_t raw[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) raw[i] = (a0 + 4 * i < total) ? p32[i] : 0u;
    uint32_t al[12];  // window realigned to byte 0 (v_alignbyte_b32)
#pragma unroll
    for (int i = 0; i < 12; ++i) al[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], sh);
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
        int kk = (x < cnt && x < ksize) ? k[x < ksize ? x : 0] : 0;
        const int bb = 3 * x;
        s0 += (int)((al[bb >> 2] >> (8 * (bb & 3))) & 0xff) * kk;
        s1 += (int)((al[(bb + 1) >> 2] >> (8 * ((bb + 1) & 3))) & 0xff) * kk;
        s2 += (int)((al[(bb + 2) >> 2] >> (8 * ((bb + 2) & 3))) & 0xff) * kk;
    }
    uint8_t* o = tmp + idx * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

// horizontal pass through LDS: a workgroup owns RH = 4 consecutive source rows (contiguous in memory, 16-byte aligned when the
// row count per image is a multiple of 4 and a row is a multiple of 4 bytes), stages them with 16-byte loads, computes the
// 4 x ow outputs from LDS (a thread = one output column of all four rows: its 13 window dwords per row come from LDS, the taps are
// read once), and writes the 4 x ow x 3 result bytes back as 16-byte stores.  One global read of every source byte instead of ~5
// overlapping window reads per lane, no byte stores: 269 -> ~70 us for 64 frames of 872 x 872.
constexpr int RH = 4;
__global__ __launch_bounds__(256) void resize_h_lds_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, int64_t rows_total, int w,
                                                           int ow, const int32_t* __restrict__ bounds, const int32_t* __restrict__ coef, int ksize) {
    extern __shared__ __attribute__((aligned(16))) uint8_t hl[];
    const int row_b = w * 3;                          // bytes per source row
    const int in_b = RH * row_b;                      // multiple of 16 (host-checked)
    const int out_row_b = ow * 3, out_b = RH * out_row_b;
    uint32_t* in32 = reinterpret_cast<uint32_t*>(hl);
    uint8_t* outl = hl + ((in_b + 64 + 15) & ~15);    // + 64: the last window may read up to 13 dwords past its start
    const int64_t r0 = (int64_t)blockIdx.x * RH;
    const uint4* g = reinterpret_cast<const uint4*>(src + r0 * row_b);
    const int nvec = in_b >> 4;
    for (int i = threadIdx.x; i < nvec; i += 256) reinterpret_cast<uint4*>(hl)[i] = g[i];
    for (int i = threadIdx.x; i < 16; i += 256) in32[(in_b >> 2) + i] = 0u;   // padding read by the last windows (weights there are 0)
    __syncthreads();
    for (int ox = threadIdx.x; ox < ow; ox += 256) {
        const int xmin = bounds[2 * ox], cnt = bounds[2 * ox + 1];
        const int32_t* k = coef + (int64_t)ox * ksize;
        int kk[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) kk[x] = (x < cnt && x < ksize) ? k[x < ksize ? x : 0] : 0;
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            const int b0 = r * row_b + xmin * 3;
            const int a0 = b0 >> 2;
            const unsigned sh = (unsigned)(b0 & 3);
            uint32_t raw[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) raw[i] = in32[a0 + i];
            uint32_t al[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) al[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], sh);
            int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int bb = 3 * x;
                s0 += (int)((al[bb >> 2] >> (8 * (bb & 3))) & 0xff) * kk[x];
                s1 += (int)((al[(bb + 1) >> 2] >> (8 * ((bb + 1) & 3))) & 0xff) * kk[x];
                s2 += (int)((al[(bb + 2) >> 2] >> (8 * ((bb + 2) & 3))) & 0xff) * kk[x];
            }
            uint8_t* o = outl + r * out_row_b + ox * 3;
            o[0] = (uint8_t)clip8(s0);
            o[1] = (uint8_t)clip8(s1);
            o[2] = (uint8_t)clip8(s2);
        }
    }
    __syncthreads();
    uint4* go = reinterpret_cast<uint4*>(tmp + r0 * out_row_b);
    for (int i = threadIdx.x; i < (out_b >> 4); i += 256) go[i] = reinterpret_cast<const uint4*>(outl)[i];
}

// generic horizontal pass (any ksize / size): byte loads
__global__ __launch_bounds__(256) void resize_h_generic_kernel(const uint8_t* src, uint8_t* tmp, int n, int h, int w, int ow,
                                                               const int32_t* bounds, const int32_t* coef, int ksize) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * h * ow) return;
    int ox = (int)(idx % ow);
    int64_t row = idx / ow;
    int xmin = bounds[2 * ox], cnt = bounds[2 * ox + 1];
    const int32_t* k = coef + (int64_t)ox * ksize;
    const uint8_t* p = src + (row * w + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        int kk = k[x];
        s0 += p[3 * x + 0] * kk;
        s1 += p[3 * x + 1] * kk;
        s2 += p[3 * x + 2] * kk;
    }
    uint8_t* o = tmp + idx * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

// vertical pass + ToTensor + Normalize: (n, h, ow, 3) u8 -> (n, 3, oh, ow) f32; one thread per (n, oy, 4 consecutive ox):
// 12 source bytes per row = 3 aligned dwords, 4 consecutive floats per colour plane = one 16-byte store each.
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* tmp, float* dst, int n, int h, int ow, int oh,
                                                            const int32_t* bounds, const int32_t* coef, int ksize, float m0,
                                                            float m1, float m2, float d0, float d1, float d2, uint8_t* u8out) {
    const int owq = ow >> 2;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * oh * owq) return;
    int oxq = (int)(idx % owq);
    int64_t r = idx / owq;
    int oy = (int)(r % oh);
    int img = (int)(r / oh);
    int ymin = bounds[2 * oy], cnt = bounds[2 * oy + 1];
    const int32_t* k = coef + (int64_t)oy * ksize;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(tmp + (((int64_t)img * h + ymin) * ow + oxq * 4) * 3);
    const int rowdw = ow * 3 / 4;
    int acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) {
        int kk = k[y];
        uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        p += rowdw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] += (int)((w0 >> (8 * i)) & 0xff) * kk;
            acc[4 + i] += (int)((w1 >> (8 * i)) & 0xff) * kk;
            acc[8 + i] += (int)((w2 >> (8 * i)) & 0xff) * kk;
        }
    }
    int c8[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) c8[i] = clip8(acc[i]);
    if (u8out) {  // optional: Pillow's uint8 result, HWC
        uint8_t* u = u8out + (((int64_t)img * oh + oy) * ow + oxq * 4) * 3;
#pragma unroll
        for (int i = 0; i < 12; ++i) u[i] = (uint8_t)c8[i];
    }
    const int64_t plane = (int64_t)oh * ow;
    float* o = dst + (int64_t)img * 3 * plane + (int64_t)oy * ow + oxq * 4;
    // ToTensor: x / 255 (fp32 division); Normalize: (x - mean) / std — same operations, same order, as torchvision
    f32x4 r0, r1, r2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r0[i] = ((float)c8[3 * i + 0] / 255.0f - m0) / d0;
        r1[i] = ((float)c8[3 * i + 1] / 255.0f - m1) / d1;
        r2[i] = ((float)c8[3 * i + 2] / 255.0f - m2) / d2;
    }
    *reinterpret_cast<f32x4*>(o) = r0;
    *reinterpret_cast<f32x4*>(o + plane) = r1;
    *reinterpret_cast<f32x4*>(o + 2 * plane) = r2;
}

// ------------------------------------------------------------------ both passes in one kernel
// A workgroup (512 threads) owns OB = 32 output rows of one image.  Their vertical windows cover a band of ~3.4 * 32 + 12 source
// rows; the band's horizontally resized uint8 rows (Pillow's intermediate image) live in LDS only: the source rows are staged
// eight at a time (two 16-byte-aligned 4-row blocks, 16-byte loads), every thread computes one output column of four of them
// (resize_h_lds_kernel's arithmetic), and the vertical pass + ToTensor + Normalize (resize_v_norm_kernel's arithmetic) reads the
// band back from LDS.  The 0.67 MB intermediate per image never reaches HBM; neighbouring bands recompute ~12 shared rows (+11 %).
constexpr int FOB = 32, FTH = 512, FRB = 8;
__global__ __launch_bounds__(FTH) void resize_fused_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, uint8_t* __restrict__ u8out, int h,
                                                           int w, int oh, int ow, const int32_t* __restrict__ bounds_h,
                                                           const int32_t* __restrict__ coef_h, int ksize_h, const int32_t* __restrict__ bounds_v,
                                                           const int32_t* __restrict__ coef_v, int ksize_v, int band_rows, float m0, float m1, float m2,
                                                           float d0, float d1, float d2) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fl[];
    const int row_b = w * 3, out_row_b = ow * 3;
    const int in_b = FRB * row_b;                              // one staged block (multiple of 16: host-checked)
    uint32_t* const in32 = reinterpret_cast<uint32_t*>(fl);
    uint8_t* const mid = fl + ((in_b + 64 + 15) & ~15);        // [band_rows][ow * 3] uint8
    const int bands = (oh + FOB - 1) / FOB;
    const int img = blockIdx.x / bands, band = blockIdx.x - img * bands;
    const int oy0 = band * FOB, oy1 = min(oh, oy0 + FOB);
    const int r_begin = bounds_v[2 * oy0];
    const int r_end = bounds_v[2 * (oy1 - 1)] + bounds_v[2 * (oy1 - 1) + 1];   // windows are monotone in oy
    const uint8_t* const simg = src + (int64_t)img * h * row_b;
    const int tid = threadIdx.x;

    // ---- horizontal pass into LDS, FRB source rows per round
    const int ox = tid % ow, rq = tid / ow;                    // (ow = 256: two row quads per round)
    int kk[16];
    int xmin = 0;
    if (tid < 2 * ow) {
        xmin = bounds_h[2 * ox];
        const int cnt = bounds_h[2 * ox + 1];
        const int32_t* k = coef_h + (int64_t)ox * ksize_h;
#pragma unroll
        for (int x = 0; x < 16; ++x) kk[x] = (x < cnt && x < ksize_h) ? k[x < ksize_h ? x : 0] : 0;
    }
    // the next block's 16-byte pieces are requested into registers before the current block is resampled (their latency hides
    // under ~500 VALU operations per thread) and parked in LDS afterwards
    constexpr int NV = 4;                                      // pieces per thread: (8 * 872 * 3 / 16 + 4) / 512 <= 4 (host-checked)
    const int slots = (in_b >> 4) + 4;
    uint4 pv[NV];
    auto fetch = [&](int rb) {
        const uint4* g = reinterpret_cast<const uint4*>(simg + (int64_t)rb * row_b);
        const int nvec = (min(FRB, h - rb) * row_b) >> 4;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i = tid + FTH * u;
            pv[u] = (i < nvec) ? g[i] : uint4{0u, 0u, 0u, 0u};
        }
    };
    fetch(r_begin & ~3);
    for (int rb = r_begin & ~3; rb < r_end; rb += FRB) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i = tid + FTH * u;
            if (i < slots) reinterpret_cast<uint4*>(fl)[i] = pv[u];
        }
        __syncthreads();
        if (rb + FRB < r_end) fetch(rb + FRB);
        if (tid < 2 * ow) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = rq * 4 + r4, sr = rb + r;
                if (sr < r_begin || sr >= r_end) continue;
                const int b0 = r * row_b + xmin * 3;
                const int a0 = b0 >> 2;
                const unsigned sh = (unsigned)(b0 & 3);
                uint32_t raw[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) raw[i] = in32[a0 + i];
                uint32_t al[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) al[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], sh);
                int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int bb = 3 * x;
                    s0 += (int)((al[bb >> 2] >> (8 * (bb & 3))) & 0xff) * kk[x];
                    s1 += (int)((al[(bb + 1) >> 2] >> (8 * ((bb + 1) & 3))) & 0xff) * kk[x];
                    s2 += (int)((al[(bb + 2) >> 2] >> (8 * ((bb + 2) & 3))) & 0xff) * kk[x];
                }
                uint8_t* o = mid + (sr - r_begin) * out_row_b + ox * 3;
                o[0] = (uint8_t)clip8(s0);
                o[1] = (uint8_t)clip8(s1);
                o[2] = (uint8_t)clip8(s2);
            }
        }
        __syncthreads();
    }
    // ---- vertical pass + ToTensor + Normalize out of LDS: one item = (output row, 4 consecutive columns)
    const int owq = ow >> 2;
    const int64_t plane = (int64_t)oh * ow;
    for (int item = tid; item < (oy1 - oy0) * owq; item += FTH) {
        const int oxq = item % owq, oy = oy0 + item / owq;
        const int ymin = bounds_v[2 * oy], cnt = bounds_v[2 * oy + 1];
        const int32_t* k = coef_v + (int64_t)oy * ksize_v;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(mid + (ymin - r_begin) * out_row_b + oxq * 12);
        const int rowdw = out_row_b >> 2;
        int acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < cnt; ++y) {
            const int kv = k[y];
            const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
            p += rowdw;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] += (int)((w0 >> (8 * i)) & 0xff) * kv;
                acc[4 + i] += (int)((w1 >> (8 * i)) & 0xff) * kv;
                acc[8 + i] += (int)((w2 >> (8 * i)) & 0xff) * kv;
            }
        }
        int c8[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) c8[i] = clip8(acc[i]);
        if (u8out) {
            uint8_t* u = u8out + (((int64_t)img * oh + oy) * ow + oxq * 4) * 3;
#pragma unroll
            for (int i = 0; i < 12; ++i) u[i] = (uint8_t)c8[i];
        }
        float* o = dst + (int64_t)img * 3 * plane + (int64_t)oy * ow + oxq * 4;
        f32x4 r0, r1, r2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r0[i] = ((float)c8[3 * i + 0] / 255.0f - m0) / d0;
            r1[i] = ((float)c8[3 * i + 1] / 255.0f - m1) / d1;
            r2[i] = ((float)c8[3 * i + 2] / 255.0f - m2) / d2;
        }
        *reinterpret_cast<f32x4*>(o) = r0;
        *reinterpret_cast<f32x4*>(o + plane) = r1;
        *reinterpret_cast<f32x4*>(o + 2 * plane) = r2;
    }
}

}  // namespace

// largest source band (rows) any 32-row output band of the vertical pass touches; the tables are host arrays here
extern "C" int egr_preprocess_band_rows(const int32_t* bounds_v_host, int32_t oh) {
    if (!bounds_v_host || oh <= 0) return 0;
    int best = 0;
    for (int oy0 = 0; oy0 < oh; oy0 += FOB) {
        const int oy1 = (oy0 + FOB < oh ? oy0 + FOB : oh) - 1;
        const int span = bounds_v_host[2 * oy1] + bounds_v_host[2 * oy1 + 1] - bounds_v_host[2 * oy0];
        if (span > best) best = span;
    }
    return best;
}

// One launch for both passes (the uint8 intermediate stays in LDS).  band_rows: egr_preprocess_band_rows of the vertical table.
// Falls back (returns EGR_EINVAL, nothing launched) when the shape does not meet the kernel's alignment / LDS limits: the
// caller then uses egr_preprocess_u8_f32.
extern "C" int egr_preprocess_fused_u8_f32(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t oh, int32_t ow, const int32_t* bounds_h,
                                           const int32_t* coef_h, int32_t ksize_h, const int32_t* bounds_v, const int32_t* coef_v, int32_t ksize_v,
                                           int32_t band_rows, const float* mean, const float* stdv, float* dst, uint8_t* u8out, void* stream) {
    if (!src || !bounds_h || !coef_h || !bounds_v || !coef_v || !mean || !stdv || !dst) return EGR_ENULL;
    if (n <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || ksize_h <= 0 || ksize_v <= 0 || band_rows <= 0) return EGR_EINVAL;
    const int in_b = FRB * w * 3;
    const size_t lds_b = (size_t)((in_b + 64 + 15) & ~15) + (size_t)band_rows * ow * 3;
    if (ow != FTH / 2 || ksize_h > 16 || ksize_v > MAXK || h % 4 != 0 || (in_b >> 4) + 4 > 4 * FTH || (4 * w * 3) % 16 != 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15) ||
        ((int64_t)h * w * 3) % 16 != 0 || lds_b > 150 * 1024)
        return EGR_EINVAL;
    static bool allowed[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return EGR_EINVAL;
    if (!allowed[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resize_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return EGR_EINVAL;
        allowed[dev] = true;
    }
    const int bands = (oh + FOB - 1) / FOB;
    hipLaunchKernelGGL(resize_fused_kernel, dim3((unsigned)(n * bands)), dim3(FTH), lds_b, (hipStream_t)stream, src, dst, u8out, h, w, oh, ow, bounds_h,
                       coef_h, ksize_h, bounds_v, coef_v, ksize_v, band_rows, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    return egr_launch_status();
}

extern "C" int egr_preprocess_u8_f32(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t oh, int32_t ow,
                                     const int32_t* bounds_h, const int32_t* coef_h, int32_t ksize_h,
                                     const int32_t* bounds_v, const int32_t* coef_v, int32_t ksize_v, const float* mean,
                                     const float* stdv, uint8_t* tmp, float* dst, uint8_t* u8out, void* stream) {
    if (!src || !bounds_h || !coef_h || !bounds_v || !coef_v || !mean || !stdv || !tmp || !dst) return EGR_ENULL;
    if (n <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || ksize_h <= 0 || ksize_v <= 0) return EGR_EINVAL;
    if (ow % 4 != 0 || ksize_h > MAXK || ksize_v > MAXK || ((uintptr_t)src & 3) || ((uintptr_t)tmp & 3) || ((uintptr_t)dst & 15))
        return EGR_EINVAL;  // 4 output columns per thread, aligned dword / 16-byte accesses
    hipStream_t s = (hipStream_t)stream;
    int64_t t1 = (int64_t)n * h * ow, t2 = (int64_t)n * oh * (ow / 4);
    if (t1 >= (1LL << 31) * 256 || t2 >= (1LL << 31) * 256) return EGR_EINVAL;
    const int in_b = RH * w * 3, out_b = RH * ow * 3;
    const size_t lds_b = (size_t)((in_b + 64 + 15) & ~15) + (size_t)out_b;
    if (ksize_h <= 16 && h % RH == 0 && in_b % 16 == 0 && out_b % 16 == 0 && lds_b <= 64 * 1024 && ((uintptr_t)src & 15) == 0 &&
        ((uintptr_t)tmp & 15) == 0)
        hipLaunchKernelGGL(resize_h_lds_kernel, dim3((unsigned)((int64_t)n * h / RH)), dim3(256), lds_b, s, src, tmp, (int64_t)n * h, w, ow,
                           bounds_h, coef_h, ksize_h);
    else if (ksize_h <= 16 && ((int64_t)h * w * 3) % 4 == 0)
        hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, src, tmp, n, h, w, ow, bounds_h,
                           coef_h, ksize_h);
    else
        hipLaunchKernelGGL(resize_h_generic_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, src, tmp, n, h, w, ow,
                           bounds_h, coef_h, ksize_h);
    int rc = egr_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, tmp, dst, n, h, ow, oh,
                       bounds_v, coef_v, ksize_v, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], u8out);
    return egr_launch_status();
}
