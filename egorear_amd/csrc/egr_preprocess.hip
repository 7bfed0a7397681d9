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

// horizontal pass: (n, h, w, 3) u8 -> (n, h, ow, 3) u8; one thread per (n, y, ox)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* src, uint8_t* tmp, int n, int h, int w, int ow,
                                                       const int32_t* bounds, const int32_t* coef, int ksize) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * h * ow) return;
    int ox = (int)(idx % ow);
    int64_t row = idx / ow;  // n*h + y
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

// vertical pass + ToTensor + Normalize: (n, h, ow, 3) u8 -> (n, 3, oh, ow) f32; one thread per (n, oy, ox)
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* tmp, float* dst, int n, int h, int ow, int oh,
                                                            const int32_t* bounds, const int32_t* coef, int ksize, float m0,
                                                            float m1, float m2, float d0, float d1, float d2, uint8_t* u8out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * oh * ow) return;
    int ox = (int)(idx % ow);
    int64_t r = idx / ow;
    int oy = (int)(r % oh);
    int img = (int)(r / oh);
    int ymin = bounds[2 * oy], cnt = bounds[2 * oy + 1];
    const int32_t* k = coef + (int64_t)oy * ksize;
    const uint8_t* p = tmp + (((int64_t)img * h + ymin) * ow + ox) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < cnt; ++y) {
        int kk = k[y];
        const uint8_t* q = p + (int64_t)y * ow * 3;
        s0 += q[0] * kk;
        s1 += q[1] * kk;
        s2 += q[2] * kk;
    }
    int c0 = clip8(s0), c1 = clip8(s1), c2 = clip8(s2);
    if (u8out) {  // optional: Pillow's uint8 result, HWC
        uint8_t* u = u8out + idx * 3;
        u[0] = (uint8_t)c0; u[1] = (uint8_t)c1; u[2] = (uint8_t)c2;
    }
    int64_t plane = (int64_t)oh * ow;
    float* o = dst + (int64_t)img * 3 * plane + (int64_t)oy * ow + ox;
    // ToTensor: x / 255 (fp32 division); Normalize: (x - mean) / std — same operations, same order, as torchvision
    o[0] = ((float)c0 / 255.0f - m0) / d0;
    o[plane] = ((float)c1 / 255.0f - m1) / d1;
    o[2 * plane] = ((float)c2 / 255.0f - m2) / d2;
}

}  // namespace

extern "C" int egr_preprocess_u8_f32(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t oh, int32_t ow,
                                     const int32_t* bounds_h, const int32_t* coef_h, int32_t ksize_h,
                                     const int32_t* bounds_v, const int32_t* coef_v, int32_t ksize_v, const float* mean,
                                     const float* stdv, uint8_t* tmp, float* dst, uint8_t* u8out, void* stream) {
    if (!src || !bounds_h || !coef_h || !bounds_v || !coef_v || !mean || !stdv || !tmp || !dst) return EGR_ENULL;
    if (n <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || ksize_h <= 0 || ksize_v <= 0) return EGR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int64_t t1 = (int64_t)n * h * ow, t2 = (int64_t)n * oh * ow;
    if (t1 >= (1LL << 31) * 256 || t2 >= (1LL << 31) * 256) return EGR_EINVAL;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, src, tmp, n, h, w, ow, bounds_h,
                       coef_h, ksize_h);
    int rc = egr_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, tmp, dst, n, h, ow, oh,
                       bounds_v, coef_v, ksize_v, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], u8out);
    return egr_launch_status();
}
