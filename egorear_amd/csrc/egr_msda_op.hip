// Multi-scale deformable attention with the operand layout of the reference's only native call,
// mmcv==2.2.0 MultiScaleDeformableAttnFunction (models/utils/deform_attn.py:155-162):
//     value (N, Lin, heads, D), spatial_shapes i64 (L, 2) = (H_l, W_l), level_start_index i64 (L),
//     sampling_locations (N, Lq, heads, L, P, 2) normalised (x, y), attention_weights (N, Lq, heads, L, P)
//     -> out (N, Lq, heads*D)
// and its backward (grad_value by scatter, grad_sampling_loc, grad_attn_weight).  The product model path does not
// call these (it uses the sample-then-project kernels in egr_attn.hip / egr_msda_bwd.hip); they exist so that a
// maintainer who only wants to replace the mmcv extension can bind the same operation.
//
// Mapping: a group of GW lanes (power of two, <= 64) owns one (n, q, head); a lane carries VEC channels of the head per
// pass, so the reads of a corner are one contiguous GW*VEC*4-byte segment of the value row.  Locations and weights are
// the same address for the whole group (a broadcast read).  Gradients of the location / weight are dot products over
// the head's channels: partial sums per lane, xor-shuffles inside the group.
#include "egr_common.h"

namespace {

struct Corner {
    int64_t i[4];  // element offsets of the four corner rows (n, token, head, 0), -1 when outside
    float w[4];    // bilinear weights hh*hw, hh*lw, lh*hw, lh*lw
    float lh, lw, hh, hw;
    bool inside;
};

__device__ __forceinline__ Corner corner_setup(float lx, float ly, int H, int W, int64_t start, int64_t lin, int64_t nbase, int heads,
                                               int D, int h) {
    Corner c;
    const float w_im = lx * (float)W - 0.5f;
    const float h_im = ly * (float)H - 0.5f;
    c.inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float hl = floorf(h_im), wl = floorf(w_im);
    c.lh = h_im - hl;
    c.lw = w_im - wl;
    c.hh = 1.f - c.lh;
    c.hw = 1.f - c.lw;
    const int h_low = (int)hl, w_low = (int)wl;
    const int h_high = h_low + 1, w_high = w_low + 1;
    c.w[0] = c.hh * c.hw;
    c.w[1] = c.hh * c.lw;
    c.w[2] = c.lh * c.hw;
    c.w[3] = c.lh * c.lw;
    const bool oky[2] = {h_low >= 0, h_high <= H - 1};
    const bool okx[2] = {w_low >= 0, w_high <= W - 1};
    const int ys[2] = {h_low, h_high}, xs[2] = {w_low, w_high};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t tok = start + (int64_t)ys[k >> 1] * W + xs[k & 1];
        const bool ok = c.inside && oky[k >> 1] && okx[k & 1] && tok >= 0 && tok < lin;  // tok < lin: inconsistent shapes never fault
        c.i[k] = ok ? ((nbase + tok) * heads + h) * D : -1;
    }
    return c;
}

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    typedef f32x4 T;
    static __device__ __forceinline__ T load(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void store(float* p, T v) { *(f32x4*)p = v; }
    static __device__ __forceinline__ float dot(T a, T b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
    static __device__ __forceinline__ void atomic_add(float* p, T v) {
        atomicAdd(p + 0, v.x);
        atomicAdd(p + 1, v.y);
        atomicAdd(p + 2, v.z);
        atomicAdd(p + 3, v.w);
    }
};
template <>
struct Vec<1> {
    typedef float T;
    static __device__ __forceinline__ T load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, T v) { *p = v; }
    static __device__ __forceinline__ float dot(T a, T b) { return a * b; }
    static __device__ __forceinline__ void atomic_add(float* p, T v) { atomicAdd(p, v); }
};

template <int VEC>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ starts, const float* __restrict__ loc,
                                                       const float* __restrict__ attn, int N, int64_t lin, int heads, int D, int Lq,
                                                       int L, int P, int gw, float* __restrict__ out) {
    typedef typename Vec<VEC>::T V;
    const int gpb = blockDim.x / gw;
    const int64_t grp = (int64_t)blockIdx.x * gpb + threadIdx.x / gw;
    const int lane = threadIdx.x % gw;
    const int64_t groups = (int64_t)N * Lq * heads;
    if (grp >= groups) return;
    const int h = (int)(grp % heads);
    const int64_t nq = grp / heads;
    const int n = (int)(nq / Lq);
    const float* lp = loc + grp * L * P * 2;
    const float* ap = attn + grp * L * P;
    for (int c0 = lane * VEC; c0 < D; c0 += gw * VEC) {
        V acc = V(0.f);
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const int64_t st = starts[l];
            for (int p = 0; p < P; ++p) {
                const float a = ap[l * P + p];
                const Corner c = corner_setup(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W, st, lin, (int64_t)n * lin, heads, D, h);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c.i[k] >= 0) acc += (a * c.w[k]) * Vec<VEC>::load(value + c.i[k] + c0);
            }
        }
        Vec<VEC>::store(out + grp * D + c0, acc);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ starts, const float* __restrict__ loc,
                                                       const float* __restrict__ attn, const float* __restrict__ gout, int N,
                                                       int64_t lin, int heads, int D, int Lq, int L, int P, int gw,
                                                       float* __restrict__ gvalue, float* __restrict__ gloc,
                                                       float* __restrict__ gattn) {
    typedef typename Vec<VEC>::T V;
    const int gpb = blockDim.x / gw;
    const int64_t grp = (int64_t)blockIdx.x * gpb + threadIdx.x / gw;
    const int lane = threadIdx.x % gw;
    const int64_t groups = (int64_t)N * Lq * heads;
    if (grp >= groups) return;  // gw divides 64: a wave holds whole groups, the shuffles below stay inside live groups
    const int h = (int)(grp % heads);
    const int64_t nq = grp / heads;
    const int n = (int)(nq / Lq);
    const float* lp = loc + grp * L * P * 2;
    const float* ap = attn + grp * L * P;
    const float* go = gout + grp * D;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const int64_t st = starts[l];
        for (int p = 0; p < P; ++p) {
            const float a = ap[l * P + p];
            const Corner c = corner_setup(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W, st, lin, (int64_t)n * lin, heads, D, h);
            float da = 0.f, dw = 0.f, dh = 0.f;
            if (c.inside) {  // group-uniform
                for (int c0 = lane * VEC; c0 < D; c0 += gw * VEC) {
                    const V g = Vec<VEC>::load(go + c0);
                    float d[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        d[k] = 0.f;
                        if (c.i[k] >= 0) {
                            d[k] = Vec<VEC>::dot(g, Vec<VEC>::load(value + c.i[k] + c0));
                            Vec<VEC>::atomic_add(gvalue + c.i[k] + c0, (a * c.w[k]) * g);
                        }
                    }
                    da += c.w[0] * d[0] + c.w[1] * d[1] + c.w[2] * d[2] + c.w[3] * d[3];
                    dw += -c.hh * d[0] + c.hh * d[1] - c.lh * d[2] + c.lh * d[3];
                    dh += -c.hw * d[0] - c.lw * d[1] + c.hw * d[2] + c.lw * d[3];
                }
            }
            for (int o = gw >> 1; o > 0; o >>= 1) {
                da += __shfl_xor(da, o, 64);
                dw += __shfl_xor(dw, o, 64);
                dh += __shfl_xor(dh, o, 64);
            }
            if (lane == 0) {
                gattn[grp * L * P + l * P + p] = da;
                gloc[(grp * L * P + l * P + p) * 2 + 0] = (float)W * a * dw;
                gloc[(grp * L * P + l * P + p) * 2 + 1] = (float)H * a * dh;
            }
        }
    }
}

int pick_group(int D, int* vec) {
    *vec = (D % 4 == 0) ? 4 : 1;
    const int per = D / *vec;
    int gw = 1;
    while (gw < per && gw < 64) gw <<= 1;
    return gw;
}

bool bad_dims(int32_t n, int64_t lin, int32_t heads, int32_t d, int32_t lq, int32_t levels, int32_t points) {
    return n < 0 || lin < 0 || heads <= 0 || d <= 0 || lq < 0 || levels <= 0 || points <= 0;
}

}  // namespace

extern "C" int egr_msda_fwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const float* sampling_loc, const float* attn_weight, int32_t n, int64_t lin, int32_t heads, int32_t d,
                                int32_t lq, int32_t levels, int32_t points, float* out, void* stream) {
    if (bad_dims(n, lin, heads, d, lq, levels, points)) return EGR_EINVAL;
    const int64_t groups = (int64_t)n * lq * heads;
    if (groups == 0) return 0;
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out) return EGR_EINVAL;
    int vec;
    const int gw = pick_group(d, &vec);
    const int gpb = 256 / gw;
    const int64_t blocks = (groups + gpb - 1) / gpb;
    if (blocks > 0x7fffffffLL) return EGR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (vec == 4)
        hipLaunchKernelGGL(msda_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, value, spatial_shapes, level_start_index, sampling_loc,
                           attn_weight, n, lin, heads, d, lq, levels, points, gw, out);
    else
        hipLaunchKernelGGL(msda_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, value, spatial_shapes, level_start_index, sampling_loc,
                           attn_weight, n, lin, heads, d, lq, levels, points, gw, out);
    return egr_launch_status();
}

extern "C" int egr_msda_bwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const float* sampling_loc, const float* attn_weight, const float* grad_out, int32_t n, int64_t lin,
                                int32_t heads, int32_t d, int32_t lq, int32_t levels, int32_t points, float* grad_value,
                                float* grad_sampling_loc, float* grad_attn_weight, void* stream) {
    if (bad_dims(n, lin, heads, d, lq, levels, points)) return EGR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t vbytes = (int64_t)n * lin * heads * d * (int64_t)sizeof(float);
    if (vbytes > 0) {
        if (!grad_value) return EGR_EINVAL;
        hipError_t e = hipMemsetAsync(grad_value, 0, (size_t)vbytes, s);
        if (e != hipSuccess) return (int)e;
    }
    const int64_t groups = (int64_t)n * lq * heads;
    if (groups == 0) return 0;
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_out || !grad_sampling_loc ||
        !grad_attn_weight)
        return EGR_EINVAL;
    int vec;
    const int gw = pick_group(d, &vec);
    const int gpb = 256 / gw;
    const int64_t blocks = (groups + gpb - 1) / gpb;
    if (blocks > 0x7fffffffLL) return EGR_EINVAL;
    if (vec == 4)
        hipLaunchKernelGGL(msda_bwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, value, spatial_shapes, level_start_index, sampling_loc,
                           attn_weight, grad_out, n, lin, heads, d, lq, levels, points, gw, grad_value, grad_sampling_loc, grad_attn_weight);
    else
        hipLaunchKernelGGL(msda_bwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, value, spatial_shapes, level_start_index, sampling_loc,
                           attn_weight, grad_out, n, lin, heads, d, lq, levels, points, gw, grad_value, grad_sampling_loc, grad_attn_weight);
    return egr_launch_status();
}
