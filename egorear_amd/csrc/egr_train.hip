// Training-step kernels other than the convolution gradients (include/egorear_train.h): BatchNorm in training mode and
// its backward, ReLU / GELU / LayerNorm / max-pool / bilinear-upsample backward, the joint-attention core backward,
// boundary layout changes, the wrapper's losses, gradient norm and fused AdamW.  All HBM-bound: 16-byte accesses across
// channels, one pass per tensor, fp64 only inside per-channel / scalar reductions.
#include "egr_common.h"
#include "egorear_train.h"

namespace {

inline unsigned nblocks(int64_t total) { return (unsigned)((total + 255) / 256); }

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ BatchNorm (training mode)
// Partial per-channel sums over a slab of rows.  FWD: (sum x, sum x^2).  BWD: (sum dz, sum dz*xhat), dz = dy*[y>0].
// 256 threads = (256 / (c/4)) row lanes x (c/4) channel quads; fp64 accumulators; LDS reduce over the row lanes.
// mm (optional): per (group, slab) and channel the extremes the abs-max BOUNDS of the outputs are made from - forward: min and max
// of x; backward: max |dy masked| (second half unused).  [g][slab][2][c] floats.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* x, const float* dy, const float* y, const float* mean,
                                                         const float* invstd, int64_t rpg, int c, int nblk, double* ws, float* mm) {
    __shared__ double red[256 * 8];
    float* const mred = reinterpret_cast<float*>(red);      // the extremes go through the same words after the sums (no extra LDS: occupancy)
    float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const int g = blockIdx.y, blk = blockIdx.x;
    const int c4 = c >> 2;
    const int rpi = 256 / c4;
    const int cl = threadIdx.x % c4, rl = threadIdx.x / c4;
    const int64_t chunk = (rpg + nblk - 1) / nblk;
    const int64_t r0 = (int64_t)blk * chunk, r1 = (r0 + chunk < rpg) ? r0 + chunk : rpg;
    const int64_t base = (int64_t)g * rpg;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    f32x4 mu = {0, 0, 0, 0}, is = {0, 0, 0, 0};
    if (BWD) {
        mu = *reinterpret_cast<const f32x4*>(mean + g * c + cl * 4);
        is = *reinterpret_cast<const f32x4*>(invstd + g * c + cl * 4);
    }
    for (int64_t r = r0 + rl; r < r1; r += rpi) {
        const int64_t o = (base + r) * c + cl * 4;
        f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
        if (!BWD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i] += (double)xv[i];
                q[i] += (double)xv[i] * (double)xv[i];
                lo[i] = fminf(lo[i], xv[i]);
                hi[i] = fmaxf(hi[i], xv[i]);
            }
        } else {
            f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
            if (y) {
                f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = yv[i] > 0.f ? d[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i] += (double)d[i];
                q[i] += (double)d[i] * (double)((xv[i] - mu[i]) * is[i]);
                hi[i] = fmaxf(hi[i], fabsf(d[i]));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[threadIdx.x * 8 + i] = s[i];
        red[threadIdx.x * 8 + 4 + i] = q[i];
    }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rpi; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i] += red[(k * c4 + cl) * 8 + i];
                q[i] += red[(k * c4 + cl) * 8 + 4 + i];
            }
    }
    if (mm) {        // (uniform: every thread of the workgroup)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mred[threadIdx.x * 8 + i] = lo[i];
            mred[threadIdx.x * 8 + 4 + i] = hi[i];
        }
        __syncthreads();
        if (rl == 0) {
            for (int k = 1; k < rpi; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] = fminf(lo[i], mred[(k * c4 + cl) * 8 + i]);
                    hi[i] = fmaxf(hi[i], mred[(k * c4 + cl) * 8 + 4 + i]);
                }
            float* m_ = mm + ((int64_t)(g * nblk + blk) * 2) * c + cl * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m_[i] = lo[i];
                m_[c + i] = hi[i];
            }
        }
    }
    if (rl == 0) {
        double* o = ws + ((int64_t)(g * nblk + blk) * 2) * c + cl * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = s[i];
            o[c + i] = q[i];
        }
    }
}

// Sum of the per-slab partials: BN_CH channel lanes x BN_SL slab lanes per block (independent loads in flight), LDS reduce in
// lane order (deterministic).  4 x 64: the conv epilogues leave one slab per M tile - thousands for the early layers -, so the slab
// direction gets the threads (16 x 16 lanes spent 0.6 ms per step in these kernels).
constexpr int BN_CH = 4, BN_SL = 64;
__device__ __forceinline__ void bn_sum_partials(const double* ws, int nblk, int c, int g, int ch, int sl, double* red, double& s, double& q) {
    s = 0;
    q = 0;
    if (ch < c)
        for (int b = sl; b < nblk; b += BN_SL) {
            const double* p = ws + ((int64_t)(g * nblk + b) * 2) * c;
            s += p[ch];
            q += p[c + ch];
        }
    red[threadIdx.x * 2] = s;
    red[threadIdx.x * 2 + 1] = q;
    __syncthreads();
    if (sl == 0)
        for (int k = 1; k < BN_SL; ++k) {
            s += red[(k * BN_CH + (threadIdx.x % BN_CH)) * 2];
            q += red[(k * BN_CH + (threadIdx.x % BN_CH)) * 2 + 1];
        }
}

// extremes over the slabs (same lane pattern as bn_sum_partials; call from every thread)
__device__ __forceinline__ void bn_minmax_partials(const float* mm, int nblk, int c, int g, int ch, int sl, float* red, float& lo, float& hi) {
    lo = INFINITY;
    hi = -INFINITY;
    if (ch < c)
        for (int b = sl; b < nblk; b += BN_SL) {
            const float* p = mm + ((int64_t)(g * nblk + b) * 2) * c;
            lo = fminf(lo, p[ch]);
            hi = fmaxf(hi, p[c + ch]);
        }
    red[threadIdx.x * 2] = lo;
    red[threadIdx.x * 2 + 1] = hi;
    __syncthreads();
    if (sl == 0)
        for (int k = 1; k < BN_SL; ++k) {
            lo = fminf(lo, red[(k * BN_CH + (threadIdx.x % BN_CH)) * 2]);
            hi = fmaxf(hi, red[(k * BN_CH + (threadIdx.x % BN_CH)) * 2 + 1]);
        }
}

// max over a 64-slot abs-max record (one thread)
__device__ __forceinline__ float record_max(const unsigned* rec) {
    unsigned m = 0;
    for (int i = 0; i < 64; ++i) m = rec[i] > m ? rec[i] : m;
    return __uint_as_float(m);
}

// a workgroup's per-channel bounds -> one atomic max into the record (bounds are >= 0: float bits order like the values)
__device__ __forceinline__ void bound_to_record(float bound, bool valid, unsigned* rec, float* red16) {
    __syncthreads();
    if (threadIdx.x < BN_CH) red16[threadIdx.x] = valid ? bound : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int i = 0; i < BN_CH; ++i) m = fmaxf(m, red16[i]);
        if (m > 0.f) __hip_atomic_fetch_max(rec + ((blockIdx.x + 7 * blockIdx.y) & 63), __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// mm / xhat_max / amax_out (all optional): the abs-max BOUND of the normalised output y = act(gamma xhat + beta [+ res]) from the batch
// extremes: |y| <= |gamma| max|xhat| + |beta| [+ max|res|], max|xhat| = max(max x - mean, mean - min x) invstd (exact per channel).
// An upper bound is all a consumer's pre-scale needs; no pass over y is spent on it.
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const double* ws, int nblk, int c, int64_t rpg, const float* gamma,
                                                              const float* beta, float* rmean, float* rvar, float momentum,
                                                              float eps, float* mean, float* invstd, float* alpha,
                                                              float* shift, const float* mm, float* xhat_max, const unsigned* amax_res,
                                                              unsigned* amax_out) {
    __shared__ double red[512];
    __shared__ float fred[512];
    __shared__ float red16[16];
    const int g = blockIdx.y;
    const int ch = blockIdx.x * BN_CH + (threadIdx.x % BN_CH), sl = threadIdx.x / BN_CH;
    double s, q;
    bn_sum_partials(ws, nblk, c, g, ch, sl, red, s, q);
    float lo = 0.f, hi = 0.f;
    if (mm) bn_minmax_partials(mm, nblk, c, g, ch, sl, fred, lo, hi);
    const bool valid = sl == 0 && ch < c;
    float bound = 0.f;
    if (valid) {
        const double n = (double)rpg;
        const double m = s / n;
        double var = q / n - m * m;
        if (var < 0) var = 0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const int o = g * c + ch;
        mean[o] = (float)m;
        invstd[o] = (float)is;
        const float a = gamma[o] * (float)is;
        alpha[o] = a;
        shift[o] = beta[o] - (float)m * a;
        if (rmean) {  // nn.BatchNorm2d buffer update: momentum blend, unbiased variance
            const double unb = (rpg > 1) ? var * n / (n - 1.0) : var;
            rmean[o] = (float)((1.0 - (double)momentum) * (double)rmean[o] + (double)momentum * m);
            rvar[o] = (float)((1.0 - (double)momentum) * (double)rvar[o] + (double)momentum * unb);
        }
        if (mm) {
            const float xh = fmaxf(hi - (float)m, (float)m - lo) * (float)is * 1.000001f;      // (one ulp of slack for the fp32 evaluation of y)
            if (xhat_max) xhat_max[o] = xh;
            bound = fabsf(gamma[o]) * xh + fabsf(beta[o]);
            if (amax_res) bound += record_max(amax_res);
            bound *= 1.000001f;
        }
    }
    if (mm && amax_out) bound_to_record(bound, valid, amax_out, red16);
}

// amax_dx (optional): the bound of dx = alpha (d - dbeta / n - xhat dgamma / n):  |dx| <= |alpha| (max|d| + |dbeta| / n + max|xhat| |dgamma| / n)
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const double* ws, int nblk, int c, float* dgamma, float* dbeta, int64_t rpg,
                                                              const float* alpha, const float* mm, const float* xhat_max, unsigned* amax_dx) {
    __shared__ double red[512];
    __shared__ float fred[512];
    __shared__ float red16[16];
    const int g = blockIdx.y;
    const int ch = blockIdx.x * BN_CH + (threadIdx.x % BN_CH), sl = threadIdx.x / BN_CH;
    double s, q;
    bn_sum_partials(ws, nblk, c, g, ch, sl, red, s, q);
    float lo = 0.f, hi = 0.f;
    const bool rec = mm && xhat_max && amax_dx;
    if (rec) bn_minmax_partials(mm, nblk, c, g, ch, sl, fred, lo, hi);
    const bool valid = sl == 0 && ch < c;
    float bound = 0.f;
    if (valid) {
        dbeta[g * c + ch] = (float)s;
        dgamma[g * c + ch] = (float)q;
        if (rec) {
            const float inv_n = 1.0f / (float)rpg;
            bound = fabsf(alpha[g * c + ch]) * (fmaxf(hi, 0.f) + fabsf((float)s) * inv_n + xhat_max[g * c + ch] * fabsf((float)q) * inv_n) * 1.00001f;
        }
    }
    if (rec) bound_to_record(bound, valid, amax_dx, red16);
}

// out[0] max= sa * max(a) [+ sb * max(b)]: the record of a tensor bounded by its inputs' records (a sum, an up-sampling gradient ...)
// dst (cols, rows) = src (rows, cols)^T through 64 x 64 LDS tiles: 256-byte row segments on both sides.  The data-gradient operand of the
// one very large Linear of the path (mlp_pred.0: 2048 x 32768) after every optimiser update - a strided torch copy ran at 1.75 TB/s.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float t[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 16 * k, c = c0 + 4 * tx;
        if (r < rows && c + 3 < cols) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)r * cols + c);
            t[ty + 16 * k][4 * tx] = v[0]; t[ty + 16 * k][4 * tx + 1] = v[1]; t[ty + 16 * k][4 * tx + 2] = v[2]; t[ty + 16 * k][4 * tx + 3] = v[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) t[ty + 16 * k][4 * tx + e] = (r < rows && c + e < cols) ? src[(int64_t)r * cols + c + e] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 16 * k, r = r0 + 4 * tx;      // output row c, output columns r .. r + 3
        if (c >= cols) continue;
        const f32x4 v = {t[4 * tx][ty + 16 * k], t[4 * tx + 1][ty + 16 * k], t[4 * tx + 2][ty + 16 * k], t[4 * tx + 3][ty + 16 * k]};
        if (r + 3 < rows && (rows & 3) == 0) *reinterpret_cast<f32x4*>(dst + (int64_t)c * rows + r) = v;
        else
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (r + e < rows) dst[(int64_t)c * rows + r + e] = v[e];
    }
}

__global__ __launch_bounds__(64) void record_bound_kernel(const unsigned* a, const unsigned* b, float sa, float sb, unsigned* out) {
    unsigned ma = a[threadIdx.x], mb = b ? b[threadIdx.x] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned oa = (unsigned)__shfl_xor((int)ma, o, 64), ob = (unsigned)__shfl_xor((int)mb, o, 64);
        ma = oa > ma ? oa : ma;
        mb = ob > mb ? ob : mb;
    }
    if (threadIdx.x == 0) {
        const float v = (sa * __uint_as_float(ma) + sb * __uint_as_float(mb)) * 1.000001f;
        if (v > 0.f) __hip_atomic_fetch_max(out, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void scale_shift_kernel(const float* x, const float* alpha, const float* shift,
                                                          const float* res, float* y, int64_t rpg, int c4, int64_t total4,
                                                          int relu) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int cq = (int)(idx % c4);
    const int g = (int)((idx / c4) / rpg);
    const f32x4 a = *reinterpret_cast<const f32x4*>(alpha + (g * c4 + cq) * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(shift + (g * c4 + cq) * 4);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + idx * 4);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = v[i] * a[i] + b[i];
    if (res) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(res + idx * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += r[i];
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = o[i] > 0.f ? o[i] : 0.f;
    }
    *reinterpret_cast<f32x4*>(y + idx * 4) = o;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, const float* y, const float* x, const float* mean,
                                                           const float* invstd, const float* alpha, const float* dgamma,
                                                           const float* dbeta, float* dx, float* dz_out, int64_t rpg, int c4,
                                                           int64_t total4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int cq = (int)(idx % c4);
    const int g = (int)((idx / c4) / rpg);
    const int po = (g * c4 + cq) * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + po), is = *reinterpret_cast<const f32x4*>(invstd + po);
    const f32x4 a = *reinterpret_cast<const f32x4*>(alpha + po);
    const f32x4 dg = *reinterpret_cast<const f32x4*>(dgamma + po), db = *reinterpret_cast<const f32x4*>(dbeta + po);
    const float inv_n = 1.0f / (float)rpg;
    f32x4 d = *reinterpret_cast<const f32x4*>(dy + idx * 4);
    if (y) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + idx * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = yv[i] > 0.f ? d[i] : 0.f;
    }
    if (dz_out) *reinterpret_cast<f32x4*>(dz_out + idx * 4) = d;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx * 4);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xh = (xv[i] - mu[i]) * is[i];
        o[i] = a[i] * (d[i] - db[i] * inv_n - xh * (dg[i] * inv_n));
    }
    *reinterpret_cast<f32x4*>(dx + idx * 4) = o;
}

// ------------------------------------------------------------------ element-wise
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* dy, const float* y, float* dx, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    f32x4 d = *reinterpret_cast<const f32x4*>(dy + idx * 4);
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + idx * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = yv[i] > 0.f ? d[i] : 0.f;
    *reinterpret_cast<f32x4*>(dx + idx * 4) = d;
}

__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* y, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    f32x4 u = *reinterpret_cast<const f32x4*>(a + idx * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(b + idx * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] += v[i];
    *reinterpret_cast<f32x4*>(y + idx * 4) = u;
}

__global__ __launch_bounds__(256) void fill_kernel(float* x, float v, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    f32x4 o = {v, v, v, v};
    *reinterpret_cast<f32x4*>(x + idx * 4) = o;
}

__global__ __launch_bounds__(256) void gelu_kernel(const float* z, float* h, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(z + idx * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = egr_act(v[i], EGR_ACT_GELU);
    *reinterpret_cast<f32x4*>(h + idx * 4) = v;
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* dh, const float* z, float* dz, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    const f32x4 d = *reinterpret_cast<const f32x4*>(dh + idx * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(z + idx * 4);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // d/dz [0.5 z (1 + erf(z/sqrt2))] = Phi(z) + z phi(z)
        const float cdf = 0.5f * (1.0f + erff(v[i] * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * v[i] * v[i]);
        o[i] = d[i] * (cdf + v[i] * pdf);
    }
    *reinterpret_cast<f32x4*>(dz + idx * 4) = o;
}

__global__ __launch_bounds__(256) void rowmask_kernel(float* x, const uint8_t* mask, int64_t total4, int c4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    if (mask[idx / c4]) return;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(x + idx * 4) = z;
}

// ------------------------------------------------------------------ MaxPool2d with recorded arg-max slot, and backward
__global__ __launch_bounds__(256) void maxpool_train_kernel(const float* x, float* y, uint8_t* slot, int n, int h, int w, int c4,
                                                            int ho, int wo, int k, int stride, int pad) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * ho * wo * c4;
    if (idx >= total) return;
    const int cq = (int)(idx % c4);
    int64_t p = idx / c4;
    const int ox = (int)(p % wo);
    p /= wo;
    const int oy = (int)(p % ho);
    const int img = (int)(p / ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bs[4] = {0, 0, 0, 0};
    bool first = true;
    for (int dy = 0; dy < k; ++dy) {
        const int iy = oy * stride - pad + dy;
        if (iy < 0 || iy >= h) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int ix = ox * stride - pad + dx;
            if (ix < 0 || ix >= w) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((int64_t)img * h + iy) * w + ix) * (c4 * 4) + cq * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (first || v[i] > best[i]) {  // first maximum in scan order (ATen max_pool2d)
                    best[i] = v[i];
                    bs[i] = dy * k + dx;
                }
            first = false;
        }
    }
    *reinterpret_cast<f32x4*>(y + idx * 4) = best;
    *reinterpret_cast<uint32_t*>(slot + idx * 4) = (uint32_t)bs[0] | ((uint32_t)bs[1] << 8) | ((uint32_t)bs[2] << 16) | ((uint32_t)bs[3] << 24);
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* dy, const uint8_t* slot, float* dx, int n, int h, int w,
                                                          int c4, int ho, int wo, int k, int stride, int pad) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * h * w * c4;
    if (idx >= total) return;
    const int cq = (int)(idx % c4);
    int64_t p = idx / c4;
    const int ix = (int)(p % w);
    p /= w;
    const int iy = (int)(p % h);
    const int img = (int)(p / h);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // windows (oy, ox) that contain (iy, ix): oy*stride - pad <= iy <= oy*stride - pad + k - 1
    int oy0 = (iy + pad - k + 1 + stride - 1);
    oy0 = oy0 <= 0 ? 0 : oy0 / stride;
    int ox0 = (ix + pad - k + 1 + stride - 1);
    ox0 = ox0 <= 0 ? 0 : ox0 / stride;
    const int oy1 = min((iy + pad) / stride, ho - 1), ox1 = min((ix + pad) / stride, wo - 1);
    for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) {
            const int me = (iy - (oy * stride - pad)) * k + (ix - (ox * stride - pad));
            const int64_t o = (((int64_t)img * ho + oy) * wo + ox) * c4 + cq;
            const uint32_t s = *reinterpret_cast<const uint32_t*>(slot + o * 4);
            const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((int)((s >> (8 * i)) & 255u) == me) acc[i] += d[i];
        }
    *reinterpret_cast<f32x4*>(dx + idx * 4) = acc;
}

// ------------------------------------------------------------------ the stem's BatchNorm + ReLU + MaxPool2d without its full-resolution tensors (round 6)
// resnet.py:16-17 in training mode used to be four passes over the (n, 128, 128, 64) tensor forward (statistics, normalise + ReLU ->
// y, max-pool -> pooled + slot) and seven backward (max-pool adjoint -> dy, then the two BatchNorm passes over dy, y, x).  The normalised
// tensor y and the scattered gradient dy never need to exist: the pooling kernel normalises what it reads, the reverse pass gathers a
// pixel's gradient from the (at most (k / stride)^2) windows that chose it and recomputes the ReLU mask from x - the same arithmetic, in
// the same order, as scale_shift_kernel / maxpool_train_kernel / maxpool_bwd_kernel / bn_partial_kernel<true> / bn_bwd_apply_kernel
// (the results are bit-identical to the separate launches; tests/test_gpu_train_kernels.py).
struct pool_geom {
    int h, w, ho, wo, k, stride, pad, ipg;      // ipg: images per BatchNorm group
    int w_shift, hw_shift, c4_shift;            // FAST launches: w, h * w and c / 4 are powers of two
};

// FAST = MaxPool2d(3, 2, 1) on power-of-two image sides and channel counts (the stem): window bounds and pixel coordinates by shifts and
// compile-time constants; otherwise the general arithmetic of maxpool_train_kernel / maxpool_bwd_kernel.  Same values either way.
template <bool FAST>
struct pool_ops {
    static __device__ __forceinline__ int K(const pool_geom& P) { return FAST ? 3 : P.k; }
    static __device__ __forceinline__ int S(const pool_geom& P) { return FAST ? 2 : P.stride; }
    static __device__ __forceinline__ int PAD(const pool_geom& P) { return FAST ? 1 : P.pad; }
    // row (pixel index over all images) -> image, y, x
    static __device__ __forceinline__ void pixel(int64_t row, const pool_geom& P, int& img, int& iy, int& ix) {
        if (FAST) {
            const int r = (int)row;
            img = r >> P.hw_shift;
            const int pix = r & ((1 << P.hw_shift) - 1);
            iy = pix >> P.w_shift;
            ix = pix & ((1 << P.w_shift) - 1);
        } else {
            const int hw = P.h * P.w;
            img = (int)(row / hw);
            const int pix = (int)(row - (int64_t)img * hw);
            iy = pix / P.w;
            ix = pix - iy * P.w;
        }
    }
};

// normalised + rectified value exactly as scale_shift_kernel stores it
__device__ __forceinline__ float bn_relu_value(float x, float a, float b) {
    const float t = x * a + b;
    return t > 0.f ? t : 0.f;
}

template <bool FAST>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* x, const float* alpha, const float* shift, float* y, uint8_t* slot,
                                                              int n, int c4, pool_geom P) {
    typedef pool_ops<FAST> O;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * P.ho * P.wo * c4;
    if (idx >= total) return;
    int cq, ox, oy, img;
    if (FAST) {          // (ho, wo = h / 2, w / 2: powers of two as well)
        const int i = (int)idx;
        cq = i & (c4 - 1);
        const int p = i >> P.c4_shift;
        ox = p & (P.wo - 1);
        oy = (p >> (P.w_shift - 1)) & (P.ho - 1);
        img = p >> (P.hw_shift - 2);
    } else {
        cq = (int)(idx % c4);
        int64_t p = idx / c4;
        ox = (int)(p % P.wo);
        p /= P.wo;
        oy = (int)(p % P.ho);
        img = (int)(p / P.ho);
    }
    const int g = img / P.ipg;
    const f32x4 a = *reinterpret_cast<const f32x4*>(alpha + (g * c4 + cq) * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(shift + (g * c4 + cq) * 4);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bs[4] = {0, 0, 0, 0};
    bool first = true;
    const int k = O::K(P), st = O::S(P), pd = O::PAD(P);
#pragma unroll
    for (int dy = 0; dy < (FAST ? 3 : k); ++dy) {
        const int iy = oy * st - pd + dy;
        if (iy < 0 || iy >= P.h) continue;
#pragma unroll
        for (int dx = 0; dx < (FAST ? 3 : k); ++dx) {
            const int ix = ox * st - pd + dx;
            if (ix < 0 || ix >= P.w) continue;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (((int64_t)img * P.h + iy) * P.w + ix) * (c4 * 4) + cq * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = bn_relu_value(xv[i], a[i], b[i]);
                if (first || v > best[i]) {  // first maximum in scan order (ATen max_pool2d), as maxpool_train_kernel
                    best[i] = v;
                    bs[i] = dy * k + dx;
                }
            }
            first = false;
        }
    }
    *reinterpret_cast<f32x4*>(y + idx * 4) = best;
    *reinterpret_cast<uint32_t*>(slot + idx * 4) = (uint32_t)bs[0] | ((uint32_t)bs[1] << 8) | ((uint32_t)bs[2] << 16) | ((uint32_t)bs[3] << 24);
}

// gradient of the normalised + rectified tensor at one pixel, gathered from the pooled gradient (maxpool_bwd_kernel's loop, its order)
template <bool FAST>
__device__ __forceinline__ f32x4 pool_gather(const float* dpool, const uint8_t* slot, int img, int iy, int ix, int cq, int c4, const pool_geom& P) {
    typedef pool_ops<FAST> O;
    const int k = O::K(P), st = O::S(P), pd = O::PAD(P);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int oy0 = (iy + pd - k + 1 + st - 1);
    oy0 = oy0 <= 0 ? 0 : oy0 / st;
    int ox0 = (ix + pd - k + 1 + st - 1);
    ox0 = ox0 <= 0 ? 0 : ox0 / st;
    const int oy1 = min((iy + pd) / st, P.ho - 1), ox1 = min((ix + pd) / st, P.wo - 1);
    for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) {
            const int me = (iy - (oy * st - pd)) * k + (ix - (ox * st - pd));
            const int64_t o = (((int64_t)img * P.ho + oy) * P.wo + ox) * c4 + cq;
            const uint32_t sl = *reinterpret_cast<const uint32_t*>(slot + o * 4);
            const f32x4 d = *reinterpret_cast<const f32x4*>(dpool + o * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((int)((sl >> (8 * i)) & 255u) == me) acc[i] += d[i];
        }
    return acc;
}

// bn_partial_kernel<true> with d = [x alpha + shift > 0] * (gathered pooled gradient): same lanes, same slabs, same accumulation order
template <bool FAST>
__global__ __launch_bounds__(256) void bn_pool_partial_kernel(const float* x, const float* dpool, const uint8_t* slot, const float* mean,
                                                              const float* invstd, const float* alpha, const float* shift, int64_t rpg, int c,
                                                              int nblk, double* ws, float* mm, pool_geom P) {
    __shared__ double red[256 * 8];
    float* const mred = reinterpret_cast<float*>(red);
    float hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const int g = blockIdx.y, blk = blockIdx.x;
    const int c4 = c >> 2;
    const int rpi = 256 / c4;
    const int cl = threadIdx.x % c4, rl = threadIdx.x / c4;
    const int64_t chunk = (rpg + nblk - 1) / nblk;
    const int64_t r0 = (int64_t)blk * chunk, r1 = (r0 + chunk < rpg) ? r0 + chunk : rpg;
    const int64_t base = (int64_t)g * rpg;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * c + cl * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + g * c + cl * 4);
    const f32x4 a = *reinterpret_cast<const f32x4*>(alpha + g * c + cl * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(shift + g * c + cl * 4);
    for (int64_t r = r0 + rl; r < r1; r += rpi) {
        const int64_t row = base + r;
        const int64_t o = row * c + cl * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
        int img, iy, ix;
        pool_ops<FAST>::pixel(row, P, img, iy, ix);
        f32x4 d = pool_gather<FAST>(dpool, slot, img, iy, ix, cl, c4, P);
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = bn_relu_value(xv[i], a[i], b[i]) > 0.f ? d[i] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] += (double)d[i];
            q[i] += (double)d[i] * (double)((xv[i] - mu[i]) * is[i]);
            hi[i] = fmaxf(hi[i], fabsf(d[i]));
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[threadIdx.x * 8 + i] = s[i];
        red[threadIdx.x * 8 + 4 + i] = q[i];
    }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rpi; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i] += red[(k * c4 + cl) * 8 + i];
                q[i] += red[(k * c4 + cl) * 8 + 4 + i];
            }
    }
    if (mm) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mred[threadIdx.x * 8 + i] = INFINITY;
            mred[threadIdx.x * 8 + 4 + i] = hi[i];
        }
        __syncthreads();
        if (rl == 0) {
            for (int k = 1; k < rpi; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) hi[i] = fmaxf(hi[i], mred[(k * c4 + cl) * 8 + 4 + i]);
            float* m_ = mm + ((int64_t)(g * nblk + blk) * 2) * c + cl * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m_[i] = INFINITY;
                m_[c + i] = hi[i];
            }
        }
    }
    if (rl == 0) {
        double* o = ws + ((int64_t)(g * nblk + blk) * 2) * c + cl * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = s[i];
            o[c + i] = q[i];
        }
    }
}

// bn_bwd_apply_kernel with the same recomputed d
template <bool FAST>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float* dpool, const uint8_t* slot, const float* x, const float* mean,
                                                                const float* invstd, const float* alpha, const float* shift, const float* dgamma,
                                                                const float* dbeta, float* dx, int64_t rpg, int c4, int64_t total4, pool_geom P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    int cq;
    int64_t row;
    if (FAST) {
        cq = (int)idx & (c4 - 1);
        row = idx >> P.c4_shift;
    } else {
        cq = (int)(idx % c4);
        row = idx / c4;
    }
    int img, iy, ix;
    pool_ops<FAST>::pixel(row, P, img, iy, ix);
    const int g = img / P.ipg;
    const int po = (g * c4 + cq) * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + po), is = *reinterpret_cast<const f32x4*>(invstd + po);
    const f32x4 a = *reinterpret_cast<const f32x4*>(alpha + po), b = *reinterpret_cast<const f32x4*>(shift + po);
    const f32x4 dg = *reinterpret_cast<const f32x4*>(dgamma + po), db = *reinterpret_cast<const f32x4*>(dbeta + po);
    const float inv_n = 1.0f / (float)rpg;
    f32x4 d = pool_gather<FAST>(dpool, slot, img, iy, ix, cq, c4, P);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = bn_relu_value(xv[i], a[i], b[i]) > 0.f ? d[i] : 0.f;
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xh = (xv[i] - mu[i]) * is[i];
        o[i] = a[i] * (d[i] - db[i] * inv_n - xh * (dg[i] * inv_n));
    }
    *reinterpret_cast<f32x4*>(dx + idx * 4) = o;
}

// ------------------------------------------------------------------ bilinear x2 (align_corners=True) backward: exact adjoint
// of upsample2x_kernel in gather form (no atomics): every source pixel collects the destination pixels whose (i0, i1)
// pair - computed with the forward's own arithmetic - touches it.
__device__ __forceinline__ float up_weight(int o, int i, int in, float scale) {
    const float f = scale * (float)o;
    const int i0 = (int)f;
    const int i1 = min(i0 + 1, in - 1);
    const float l1 = fminf(fmaxf(f - (float)i0, 0.f), 1.f);
    float wgt = 0.f;
    if (i0 == i) wgt += 1.f - l1;
    if (i1 == i) wgt += l1;
    return wgt;
}

__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* dy, const float* y, float* dx, int n, int h, int w,
                                                             int c4) {
    const int ho = 2 * h, wo = 2 * w;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n * h * w * c4;
    if (idx >= total) return;
    const int cq = (int)(idx % c4);
    int64_t p = idx / c4;
    const int ix = (int)(p % w);
    p /= w;
    const int iy = (int)(p % h);
    const int img = (int)(p / h);
    const float sh = (ho > 1) ? (float)(h - 1) / (float)(ho - 1) : 0.f;
    const float sw = (wo > 1) ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // candidate output rows / columns 2 i - 2 .. 2 i + 3 and their weights first (no memory), then the loads of a row all together:
    // a loop of load -> weight test -> fma per candidate was one memory round trip per contributing output (up to 16 in a row)
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int oy = 2 * iy - 2 + k, ox = 2 * ix - 2 + k;
        wy[k] = (oy >= 0 && oy < ho) ? up_weight(oy, iy, h, sh) : 0.f;
        wx[k] = (ox >= 0 && ox < wo) ? up_weight(ox, ix, w, sw) : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        if (wy[a] == 0.f) continue;
        const int oy = 2 * iy - 2 + a;
        f32x4 d[6], yv[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            d[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            yv[b] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (wx[b] != 0.f) {
                const int64_t o = ((((int64_t)img * ho + oy) * wo + (2 * ix - 2 + b)) * c4 + cq) * 4;
                d[b] = *reinterpret_cast<const f32x4*>(dy + o);
                if (y) yv[b] = *reinterpret_cast<const f32x4*>(y + o);
            }
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            if (wx[b] == 0.f) continue;
            const float wgt = wy[a] * wx[b];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(wgt, yv[b][i] > 0.f ? d[b][i] : 0.f, acc[i]);
        }
    }
    *reinterpret_cast<f32x4*>(dx + idx * 4) = acc;
}

// ------------------------------------------------------------------ boundary layouts
__global__ __launch_bounds__(256) void planes_to_nhwc_kernel(const float* planes, egr_nmap map, float* y, int n, int c, int hw,
                                                             int cpad) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * hw * cpad) return;
    const int ch = (int)(idx % cpad);
    const int64_t r = idx / cpad;
    const int p = (int)(r % hw), img = (int)(r / hw);
    y[idx] = ch < c ? planes[egr_map(map, img) + (int64_t)ch * hw + p] : 0.f;
}

__global__ __launch_bounds__(256) void nhwc_to_planes_kernel(const float* x, float* planes, egr_nmap map, int n, int c, int hw,
                                                             int cpad) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * c * hw) return;
    const int p = (int)(idx % hw);
    const int64_t r = idx / hw;
    const int ch = (int)(r % c), img = (int)(r / c);
    planes[egr_map(map, img) + (int64_t)ch * hw + p] = x[((int64_t)img * hw + p) * cpad + ch];
}

__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* img, egr_nmap map, int n, int h, int w, float* cols) {
    const int ho = h / 2, wo = w / 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * ho * wo * 160) return;
    const int col = (int)(idx % 160);
    int64_t r = idx / 160;
    const int ox = (int)(r % wo);
    r /= wo;
    const int oy = (int)(r % ho);
    const int im = (int)(r / ho);
    float v = 0.f;
    if (col < 147) {  // OIHW flatten order of the 7x7 weight: (c, ky, kx)
        const int ch = col / 49, ky = (col % 49) / 7, kx = col % 7;
        const int iy = 2 * oy - 3 + ky, ix = 2 * ox - 3 + kx;
        if (iy >= 0 && iy < h && ix >= 0 && ix < w) v = img[egr_map(map, im) + ((int64_t)ch * h + iy) * w + ix];
    }
    cols[idx] = v;
}

// ------------------------------------------------------------------ LayerNorm backward
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* dy, const float* pre, const float* gamma, float* ds,
                                                            float* rowstats, int rows, float eps, int rpg) {
    const int c = VPL * 64;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    if (rpg > 0) gamma += (row / rpg) * c;
    float v[VPL], gd[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = pre[(int64_t)row * c + i * 64 + lane];
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float dlt = v[i] - mean;
        q += dlt * dlt;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)c + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int ch = i * 64 + lane;
        v[i] = (v[i] - mean) * rstd;  // xhat
        gd[i] = dy[(int64_t)row * c + ch] * gamma[ch];
        m1 += gd[i];
        m2 += gd[i] * v[i];
    }
    m1 = wave_sum(m1) / (float)c;
    m2 = wave_sum(m2) / (float)c;
#pragma unroll
    for (int i = 0; i < VPL; ++i) ds[(int64_t)row * c + i * 64 + lane] = rstd * (gd[i] - m1 - v[i] * m2);
    if (lane == 0) {
        rowstats[row * 2 + 0] = mean;
        rowstats[row * 2 + 1] = rstd;
    }
}

__global__ __launch_bounds__(256) void layernorm_param_grad_kernel(const float* dy, const float* pre, const float* rowstats,
                                                                   float* dgamma, float* dbeta, int rows_per_group, int c) {
    // block = 16 channel lanes x 16 row lanes; every lane walks its rows with stride 16
    __shared__ float rg[256], rb[256];
    const int g = blockIdx.y;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cl;
    float sg = 0.f, sb = 0.f;
    if (ch < c)
        for (int r = rl; r < rows_per_group; r += 16) {
            const int64_t row = (int64_t)g * rows_per_group + r;
            const float d = dy[row * c + ch];
            sg = fmaf(d, (pre[row * c + ch] - rowstats[row * 2]) * rowstats[row * 2 + 1], sg);
            sb += d;
        }
    rg[threadIdx.x] = sg;
    rb[threadIdx.x] = sb;
    __syncthreads();
    if (rl == 0 && ch < c) {
        for (int k = 1; k < 16; ++k) {
            sg += rg[k * 16 + cl];
            sb += rb[k * 16 + cl];
        }
        dgamma[g * c + ch] = sg;
        dbeta[g * c + ch] = sb;
    }
}

// ------------------------------------------------------------------ joint-attention core backward: one wave per (b, head)
__global__ __launch_bounds__(64) void joint_mha_bwd_kernel(const float* qkv, const float* dout, float* dqkv, int J, int heads,
                                                           int d, float scale) {
    __shared__ float sq[16 * 64], sk[16 * 64], sv[16 * 64], sdo[16 * 64], sp[16 * 16], sds[16 * 16];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x;
    const int c = heads * d;
    for (int i = lane; i < J * d; i += 64) {
        const int t = i / d, dd = i % d;
        const float* r = qkv + ((int64_t)b * J + t) * 3 * c + h * d + dd;
        sq[t * d + dd] = r[0];
        sk[t * d + dd] = r[c];
        sv[t * d + dd] = r[2 * c];
        sdo[t * d + dd] = dout[((int64_t)b * J + t) * c + h * d + dd];
    }
    __syncthreads();
    const int i = lane >> 2, gq = lane & 3;
    float s[4], dp[4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int jj = gq + 4 * t;
        float dot = 0.f, dpv = 0.f;
        if (i < J && jj < J)
            for (int dd = 0; dd < d; ++dd) {
                dot = fmaf(sq[i * d + dd], sk[jj * d + dd], dot);
                dpv = fmaf(sdo[i * d + dd], sv[jj * d + dd], dpv);   // dP = dO V^T
            }
        s[t] = (i < J && jj < J) ? dot * scale : -INFINITY;
        dp[t] = dpv;
        mx = fmaxf(mx, s[t]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s[t] = (s[t] == -INFINITY) ? 0.f : expf(s[t] - mx);
        sum += s[t];
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    float rowdot = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s[t] = (i < J) ? s[t] / sum : 0.f;   // P
        rowdot += s[t] * dp[t];
    }
    rowdot += __shfl_xor(rowdot, 1, 64);
    rowdot += __shfl_xor(rowdot, 2, 64);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        sp[i * 16 + gq + 4 * t] = s[t];
        sds[i * 16 + gq + 4 * t] = s[t] * (dp[t] - rowdot) * scale;   // dS (softmax backward), scale folded in
    }
    __syncthreads();
    for (int idx = lane; idx < J * d; idx += 64) {
        const int t = idx / d, dd = idx % d;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int jj = 0; jj < J; ++jj) {
            dq = fmaf(sds[t * 16 + jj], sk[jj * d + dd], dq);    // dQ = dS K
            dk = fmaf(sds[jj * 16 + t], sq[jj * d + dd], dk);    // dK = dS^T Q
            dv = fmaf(sp[jj * 16 + t], sdo[jj * d + dd], dv);    // dV = P^T dO
        }
        float* o = dqkv + ((int64_t)b * J + t) * 3 * c + h * d + dd;
        o[0] = dq;
        o[c] = dk;
        o[2 * c] = dv;
    }
}

// ------------------------------------------------------------------ small reductions
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int64_t ld, int64_t rows, int c, const float* scale,
                                                     float* out, int accumulate, int64_t gx, int64_t gs, int cps, int64_t sstride) {
    __shared__ float red[256];
    const int g = blockIdx.y;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cl;
    x += (int64_t)g * gx;
    if (scale) scale += (int64_t)g * gs + (cps > 0 ? (int64_t)(min(ch, c - 1) / cps) * sstride : 0);
    float s = 0.f;
    if (ch < c)
        for (int64_t r = rl; r < rows; r += 16) s = fmaf(scale ? scale[r] : 1.f, x[r * ld + ch], s);
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && ch < c) {
        for (int k = 1; k < 16; ++k) s += red[k * 16 + cl];
        float* o = out + (int64_t)g * c + ch;
        *o = accumulate ? *o + s : s;
    }
}

__global__ __launch_bounds__(256) void fold_rows_kernel(const float* x, float* y, int64_t total, int fold, int c) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % c);
    const int64_t r = idx / c;
    float s = 0.f;
    for (int k = 0; k < fold; ++k) s += x[(r * fold + k) * c + ch];
    y[idx] = s;
}

__global__ __launch_bounds__(256) void jqa_sum_bwd_kernel(const float* dx, float* d_embed, float* d_bfb, int b, int j, int c,
                                                          int bpg) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int groups = b / bpg;
    const int64_t n_embed = (int64_t)groups * j * c, n_bfb = (int64_t)b * c;
    if (idx < n_embed) {
        const int ch = (int)(idx % c);
        const int64_t r = idx / c;
        const int jj = (int)(r % j), g = (int)(r / j);
        float s = 0.f;
        for (int bb = 0; bb < bpg; ++bb) s += dx[(((int64_t)g * bpg + bb) * j + jj) * c + ch];
        d_embed[idx] = s;
    } else if (idx < n_embed + n_bfb) {
        const int64_t k = idx - n_embed;
        const int ch = (int)(k % c);
        const int bb = (int)(k / c);
        float s = 0.f;
        for (int jj = 0; jj < j; ++jj) s += dx[((int64_t)bb * j + jj) * c + ch];
        d_bfb[k] = s;
    }
}

// ------------------------------------------------------------------ losses: one wave per row
__global__ __launch_bounds__(256) void rownorm_loss_kernel(const float* pred, const float* gt, int64_t rows, int d, int inner,
                                                           int64_t ldp, int64_t ldg, float coef, double* loss, float* dpred) {
    __shared__ double part[4];
    const int lane = threadIdx.x & 63;
    double mine = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int64_t po = (row / inner) * ldp + (row % inner) * d, go = (row / inner) * ldg + (row % inner) * d;
        float diff = 0.f;
        if (lane < d) diff = pred[po + lane] - gt[go + lane];
        const float nrm = sqrtf(wave_sum(diff * diff));
        if (lane < d && dpred) dpred[po + lane] = nrm > 0.f ? coef * diff / nrm : 0.f;
        mine += (double)nrm;
    }
    if (lane == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (double)coef * ((part[0] + part[1]) + (part[2] + part[3])));
}

// nn.MSELoss(reduction="mean") * weight and its gradient (the heat-map stages' loss, pl_wrappers/egoposeformer/heatmap.py:215-218):
// loss += coef * sum (pred - gt)^2, dpred = 2 coef (pred - gt), coef = weight / n
__global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int64_t n4, float coef,
                                                       double* loss, float* __restrict__ dpred) {
    __shared__ double part[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(pred + i * 4), g = *reinterpret_cast<const f32x4*>(gt + i * 4);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[e] = p[e] - g[e];
            s += (double)d[e] * d[e];
            d[e] *= 2.f * coef;
        }
        if (dpred) *reinterpret_cast<f32x4*>(dpred + i * 4) = d;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (double)coef * ((part[0] + part[1]) + (part[2] + part[3])));
}

// ------------------------------------------------------------------ optimiser
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, int64_t n4, double* out) {
    __shared__ double part[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
        s += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

__global__ void zero_double_kernel(double* p) { *p = 0.0; }
__global__ void set4_kernel(float* dst, float a, float b, float c, float d) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }

// torch.optim.AdamW single-tensor update: p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps), with g pre-scaled by the clip_grad_norm_ coefficient.
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n4, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const double* sumsq, float clip, const float* hyper) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }
    float coef = 1.f;
    if (sumsq) {
        const float total = (float)sqrt(*sumsq);
        coef = fminf(clip / (total + 1e-6f), 1.0f);
    }
    f32x4 pv = *reinterpret_cast<const f32x4*>(p + idx * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + idx * 4);
    f32x4 mv = *reinterpret_cast<const f32x4*>(m + idx * 4);
    f32x4 vv = *reinterpret_cast<const f32x4*>(v + idx * 4);
    const float step_size = lr / bc1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float gi = gv[i] * coef;
        pv[i] = pv[i] * (1.f - lr * wd);
        mv[i] = mv[i] + (gi - mv[i]) * (1.f - b1);
        vv[i] = vv[i] * b2 + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vv[i]) / bc2_sqrt + eps;
        pv[i] = pv[i] - step_size * (mv[i] / denom);
    }
    *reinterpret_cast<f32x4*>(p + idx * 4) = pv;
    *reinterpret_cast<f32x4*>(m + idx * 4) = mv;
    *reinterpret_cast<f32x4*>(v + idx * 4) = vv;
}

// ------------------------------------------------------------------ multi-tensor re-packing
__global__ __launch_bounds__(256) void repack_kernel(const egr_repack_desc* table, const int64_t* blocks) {
    const egr_repack_desc d = table[blocks[2 * blockIdx.x]];
    const int64_t first = blocks[2 * blockIdx.x + 1];
    const int tk = d.taps * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = first + u * 256 + threadIdx.x;
        if (i >= d.total) return;
        float v = 0.f;
        if (d.kind == EGR_REPACK_COPYPAD) {
            if (i < d.rows) v = d.src[i];
            d.dst[i] = v;
        } else if (d.kind == EGR_REPACK_FWD) {
            const int Kp = d.cin_pad * d.taps;
            const int row = (int)(i / Kp), k = (int)(i - (int64_t)row * Kp);
            const int cb = k / tk, rem = k - cb * tk, tap = rem >> 5, ci = cb * 32 + (rem & 31);
            if (row < d.rows && ci < d.cin) v = d.src[((int64_t)row * d.cin_tot + d.ci0 + ci) * d.taps + tap];
            d.dst[i] = v;
        } else if (d.kind == EGR_REPACK_DGRAD) {
            const int Kt = d.rows_pad * d.taps;   // this descriptor's share of the K side (k_off is a multiple of 32 when taps > 1)
            const int ci = (int)(i / Kt), k = (int)(i - (int64_t)ci * Kt);
            const int cob = k / tk, rem = k - cob * tk, tap = rem >> 5, co = cob * 32 + (rem & 31);
            if (ci < d.cin && co < d.rows) v = d.src[((int64_t)co * d.cin_tot + d.ci0 + ci) * d.taps + tap];
            d.dst[((int64_t)ci * d.k_tot + d.k_off) * d.taps + k] = v;
        } else {  // UNPACK: i enumerates the destination slice (row, ci, tap)
            const int per_row = d.cin * d.taps;
            const int row = (int)(i / per_row), r2 = (int)(i - (int64_t)row * per_row);
            const int ci = r2 / d.taps, tap = r2 - ci * d.taps;
            v = d.src[(int64_t)row * d.cin_pad * d.taps + ((ci >> 5) * d.taps + tap) * 32 + (ci & 31)];
            d.dst[((int64_t)row * d.cin_tot + d.ci0 + ci) * d.taps + tap] = v;
        }
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

extern "C" int egr_repack_f32(const egr_repack_desc* table, const int64_t* blocks, int32_t n_blocks, void* stream) {
    if (!table || !blocks) return EGR_ENULL;
    if (n_blocks <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, table, blocks);
    return egr_launch_status();
}

extern "C" int32_t egr_bn_blocks(int64_t rows_per_group) {
    // >= 32 rows per slab: the deep layers (4096 rows per group) still spread over 128 workgroups per group instead of 8
    int64_t n = rows_per_group / 32;
    if (n < 1) n = 1;
    if (n > 512) n = 512;
    return (int32_t)n;
}

static int bn_shape_ok(int64_t rpg, int c, int groups) {
    if (rpg <= 0 || groups <= 0 || groups > 65535 || c < 64 || c > 1024 || (c & (c - 1)) != 0) return 0;
    return 1;
}

extern "C" int egr_bn_stats_ex_f32(const float* x, int64_t rows_per_group, int32_t c, int32_t groups, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                   float* mean, float* invstd, float* alpha, float* shift, double* workspace,
                                   size_t workspace_doubles, float* xhat_max, const uint32_t* amax_res, uint32_t* amax_out, void* stream) {
    if (!x || !gamma || !beta || !mean || !invstd || !alpha || !shift || !workspace) return EGR_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EGR_ENULL;
    if (!bn_shape_ok(rows_per_group, c, groups)) return EGR_EINVAL;
    if ((((uintptr_t)amax_res) | ((uintptr_t)amax_out)) & 3) return EGR_EINVAL;
    const int nblk = egr_bn_blocks(rows_per_group);
    const bool ext = xhat_max || amax_out;                 // the batch extremes go behind the sums: one more third of the workspace
    const size_t sums = (size_t)groups * nblk * 2 * c;
    if (workspace_doubles < sums + (ext ? sums / 2 : 0)) return EGR_EWORKSPACE;
    float* mm = ext ? reinterpret_cast<float*>(workspace + sums) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<false>, dim3(nblk, groups), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr,
                       rows_per_group, c, nblk, workspace, mm);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((c + BN_CH - 1) / BN_CH, groups), dim3(256), 0, s, workspace, nblk, c,
                       rows_per_group, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, alpha, shift, mm, xhat_max,
                       amax_res, amax_out);
    return egr_launch_status();
}

extern "C" int egr_bn_stats_f32(const float* x, int64_t rows_per_group, int32_t c, int32_t groups, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                float* mean, float* invstd, float* alpha, float* shift, double* workspace,
                                size_t workspace_doubles, void* stream) {
    return egr_bn_stats_ex_f32(x, rows_per_group, c, groups, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, alpha,
                               shift, workspace, workspace_doubles, nullptr, nullptr, nullptr, stream);
}

extern "C" int egr_bn_finalize_f32(const double* partials, int32_t slabs, int64_t rows_per_group, int32_t c, int32_t groups, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                   float* invstd, float* alpha, float* shift, float* xhat_max, const uint32_t* amax_res, uint32_t* amax_out,
                                   void* stream) {
    if (!partials || !gamma || !beta || !mean || !invstd || !alpha || !shift) return EGR_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EGR_ENULL;
    if (!bn_shape_ok(rows_per_group, c, groups) || slabs <= 0 || ((((uintptr_t)amax_res) | ((uintptr_t)amax_out)) & 3)) return EGR_EINVAL;
    const float* mm = reinterpret_cast<const float*>(partials + (size_t)groups * slabs * 2 * c);      // the extremes behind the sums
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((c + BN_CH - 1) / BN_CH, groups), dim3(256), 0, (hipStream_t)stream, partials, slabs, c,
                       rows_per_group, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, alpha, shift, mm, xhat_max,
                       amax_res, amax_out);
    return egr_launch_status();
}

extern "C" int egr_transpose_f32(const float* src, int32_t rows, int32_t cols, float* dst, void* stream) {
    if (!src || !dst) return EGR_ENULL;
    if (rows <= 0 || cols <= 0 || ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) || src == dst) return EGR_EINVAL;
    if ((cols + 63) / 64 > 65535 * 32 || (rows + 63) / 64 > 65535) return EGR_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       rows, cols);
    return egr_launch_status();
}

extern "C" int egr_record_bound_f32(const uint32_t* a, const uint32_t* b, float scale_a, float scale_b, uint32_t* out, void* stream) {
    if (!a || !out) return EGR_ENULL;
    if (((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 3) || !(scale_a >= 0.f) || !(scale_b >= 0.f)) return EGR_EINVAL;
    hipLaunchKernelGGL(record_bound_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, scale_a, scale_b, out);
    return egr_launch_status();
}

extern "C" int egr_scale_shift_f32(const float* x, const float* alpha, const float* shift, const float* res, float* y,
                                   int64_t rows_per_group, int32_t c, int32_t groups, int32_t relu, void* stream) {
    if (!x || !alpha || !shift || !y) return EGR_ENULL;
    if (rows_per_group <= 0 || groups <= 0 || c <= 0 || c % 4 != 0) return EGR_EINVAL;
    const int64_t total4 = (int64_t)groups * rows_per_group * (c / 4);
    hipLaunchKernelGGL(scale_shift_kernel, dim3(nblocks(total4)), dim3(256), 0, (hipStream_t)stream, x, alpha, shift, res, y,
                       rows_per_group, c / 4, total4, relu);
    return egr_launch_status();
}

extern "C" int egr_bn_backward_ex_f32(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                      const float* alpha, int64_t rows_per_group, int32_t c, int32_t groups, float* dgamma,
                                      float* dbeta, float* dx, float* dz_out, double* workspace, size_t workspace_doubles,
                                      const float* xhat_max, uint32_t* amax_dx, void* stream);

extern "C" int egr_bn_backward_f32(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                   const float* alpha, int64_t rows_per_group, int32_t c, int32_t groups, float* dgamma,
                                   float* dbeta, float* dx, float* dz_out, double* workspace, size_t workspace_doubles,
                                   void* stream) {
    return egr_bn_backward_ex_f32(dy, y, x, mean, invstd, alpha, rows_per_group, c, groups, dgamma, dbeta, dx, dz_out, workspace,
                                  workspace_doubles, nullptr, nullptr, stream);
}

extern "C" int egr_bn_backward_ex_f32(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                      const float* alpha, int64_t rows_per_group, int32_t c, int32_t groups, float* dgamma,
                                      float* dbeta, float* dx, float* dz_out, double* workspace, size_t workspace_doubles,
                                      const float* xhat_max, uint32_t* amax_dx, void* stream) {
    if (!dy || !x || !mean || !invstd || !alpha || !dgamma || !dbeta || !dx || !workspace) return EGR_ENULL;
    if (!bn_shape_ok(rows_per_group, c, groups)) return EGR_EINVAL;
    if (((uintptr_t)amax_dx & 3) || ((amax_dx != nullptr) != (xhat_max != nullptr))) return EGR_EINVAL;
    const int nblk = egr_bn_blocks(rows_per_group);
    const size_t sums = (size_t)groups * nblk * 2 * c;
    if (workspace_doubles < sums + (amax_dx ? sums / 2 : 0)) return EGR_EWORKSPACE;
    float* mm = amax_dx ? reinterpret_cast<float*>(workspace + sums) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<true>, dim3(nblk, groups), dim3(256), 0, s, x, dy, y, mean, invstd, rows_per_group, c,
                       nblk, workspace, mm);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((c + BN_CH - 1) / BN_CH, groups), dim3(256), 0, s, workspace, nblk, c, dgamma,
                       dbeta, rows_per_group, alpha, mm, xhat_max, amax_dx);
    const int64_t total4 = (int64_t)groups * rows_per_group * (c / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblocks(total4)), dim3(256), 0, s, dy, y, x, mean, invstd, alpha, dgamma, dbeta,
                       dx, dz_out, rows_per_group, c / 4, total4);
    return egr_launch_status();
}

static int log2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

// fast: MaxPool2d(3, 2, 1), even power-of-two sides, power-of-two channel quads, 32-bit element indices
static int pool_geom_ok(int n, int h, int w, int c, int groups, int k, int stride, int pad, pool_geom& P, bool& fast) {
    if (n <= 0 || groups <= 0 || n % groups != 0 || h <= 0 || w <= 0 || c % 4 != 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0 || 2 * pad > k) return 0;
    P.h = h; P.w = w; P.k = k; P.stride = stride; P.pad = pad; P.ipg = n / groups;
    P.ho = (h + 2 * pad - k) / stride + 1;
    P.wo = (w + 2 * pad - k) / stride + 1;
    P.w_shift = log2_exact(w);
    P.hw_shift = (P.w_shift >= 0 && log2_exact(h) >= 0) ? P.w_shift + log2_exact(h) : -1;
    P.c4_shift = log2_exact(c / 4);
    fast = k == 3 && stride == 2 && pad == 1 && h >= 2 && w >= 2 && P.w_shift >= 1 && P.hw_shift >= 2 && P.c4_shift >= 0 &&
           (int64_t)n * h * w * (c / 4) < (1LL << 31);
    return P.ho > 0 && P.wo > 0 && (int64_t)n * h * w < (1LL << 31);
}

extern "C" int egr_bn_relu_maxpool_f32(const float* x, const float* alpha, const float* shift, float* y, uint8_t* slot, int32_t n, int32_t h,
                                       int32_t w, int32_t c, int32_t groups, int32_t k, int32_t stride, int32_t pad, void* stream) {
    if (!x || !alpha || !shift || !y || !slot) return EGR_ENULL;
    pool_geom P;
    bool fast = false;
    if (!pool_geom_ok(n, h, w, c, groups, k, stride, pad, P, fast)) return EGR_EINVAL;
    if ((((uintptr_t)x) | ((uintptr_t)alpha) | ((uintptr_t)shift) | ((uintptr_t)y)) & 15 || ((uintptr_t)slot & 3)) return EGR_EINVAL;
    const int64_t total = (int64_t)n * P.ho * P.wo * (c / 4);
    if (fast) hipLaunchKernelGGL(bn_relu_maxpool_kernel<true>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, alpha, shift, y, slot, n, c / 4, P);
    else hipLaunchKernelGGL(bn_relu_maxpool_kernel<false>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, alpha, shift, y, slot, n, c / 4, P);
    return egr_launch_status();
}

extern "C" int egr_bn_pool_backward_f32(const float* dpool, const uint8_t* slot, const float* x, const float* mean, const float* invstd,
                                        const float* alpha, const float* shift, int32_t n, int32_t h, int32_t w, int32_t c, int32_t groups,
                                        int32_t k, int32_t stride, int32_t pad, float* dgamma, float* dbeta, float* dx, double* workspace,
                                        size_t workspace_doubles, const float* xhat_max, uint32_t* amax_dx, void* stream) {
    if (!dpool || !slot || !x || !mean || !invstd || !alpha || !shift || !dgamma || !dbeta || !dx || !workspace) return EGR_ENULL;
    pool_geom P;
    bool fast = false;
    if (!pool_geom_ok(n, h, w, c, groups, k, stride, pad, P, fast)) return EGR_EINVAL;
    const int64_t rpg = (int64_t)P.ipg * h * w;
    if (!bn_shape_ok(rpg, c, groups)) return EGR_EINVAL;
    if (((uintptr_t)amax_dx & 3) || ((amax_dx != nullptr) != (xhat_max != nullptr))) return EGR_EINVAL;
    if ((((uintptr_t)dpool) | ((uintptr_t)x) | ((uintptr_t)dx)) & 15 || ((uintptr_t)slot & 3)) return EGR_EINVAL;
    const int nblk = egr_bn_blocks(rpg);
    const size_t sums = (size_t)groups * nblk * 2 * c;
    if (workspace_doubles < sums + (amax_dx ? sums / 2 : 0)) return EGR_EWORKSPACE;
    float* mm = amax_dx ? reinterpret_cast<float*>(workspace + sums) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (fast) hipLaunchKernelGGL(bn_pool_partial_kernel<true>, dim3(nblk, groups), dim3(256), 0, s, x, dpool, slot, mean, invstd, alpha, shift, rpg, c, nblk, workspace, mm, P);
    else hipLaunchKernelGGL(bn_pool_partial_kernel<false>, dim3(nblk, groups), dim3(256), 0, s, x, dpool, slot, mean, invstd, alpha, shift, rpg, c, nblk, workspace, mm, P);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((c + BN_CH - 1) / BN_CH, groups), dim3(256), 0, s, workspace, nblk, c, dgamma, dbeta, rpg,
                       alpha, mm, xhat_max, amax_dx);
    const int64_t total4 = (int64_t)groups * rpg * (c / 4);
    if (fast) hipLaunchKernelGGL(bn_pool_bwd_apply_kernel<true>, dim3(nblocks(total4)), dim3(256), 0, s, dpool, slot, x, mean, invstd, alpha, shift, dgamma, dbeta, dx, rpg, c / 4, total4, P);
    else hipLaunchKernelGGL(bn_pool_bwd_apply_kernel<false>, dim3(nblocks(total4)), dim3(256), 0, s, dpool, slot, x, mean, invstd, alpha, shift, dgamma, dbeta, dx, rpg, c / 4, total4, P);
    return egr_launch_status();
}

#define EGR_ELTWISE_CHECK(n, ...)                                         \
    do {                                                                  \
        const void* ptrs_[] = {__VA_ARGS__};                              \
        for (const void* q_ : ptrs_) {                                    \
            if (!q_) return EGR_ENULL;                                    \
            if (!aligned16(q_)) return EGR_EINVAL;                        \
        }                                                                 \
        if ((n) <= 0 || (n) % 4 != 0) return EGR_EINVAL;                  \
    } while (0)

extern "C" int egr_relu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    EGR_ELTWISE_CHECK(n, dy, y, dx);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n / 4);
    return egr_launch_status();
}

extern "C" int egr_add_f32(const float* a, const float* b, float* y, int64_t n, void* stream) {
    EGR_ELTWISE_CHECK(n, a, b, y);
    hipLaunchKernelGGL(add_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, y, n / 4);
    return egr_launch_status();
}

extern "C" int egr_gelu_f32(const float* z, float* h, int64_t n, void* stream) {
    EGR_ELTWISE_CHECK(n, z, h);
    hipLaunchKernelGGL(gelu_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, z, h, n / 4);
    return egr_launch_status();
}

extern "C" int egr_gelu_bwd_f32(const float* dh, const float* z, float* dz, int64_t n, void* stream) {
    EGR_ELTWISE_CHECK(n, dh, z, dz);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, dh, z, dz, n / 4);
    return egr_launch_status();
}

extern "C" int egr_fill_f32(float* x, float v, int64_t n, void* stream) {
    EGR_ELTWISE_CHECK(n, x);
    hipLaunchKernelGGL(fill_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, x, v, n / 4);
    return egr_launch_status();
}

extern "C" int egr_rowmask_f32(float* x, const uint8_t* mask, int64_t rows, int32_t c, void* stream) {
    if (!x || !mask) return EGR_ENULL;
    if (rows <= 0 || c <= 0 || c % 4 != 0 || !aligned16(x)) return EGR_EINVAL;
    const int64_t total4 = rows * (c / 4);
    hipLaunchKernelGGL(rowmask_kernel, dim3(nblocks(total4)), dim3(256), 0, (hipStream_t)stream, x, mask, total4, c / 4);
    return egr_launch_status();
}

extern "C" int egr_maxpool_train_f32(const float* x, float* y, uint8_t* slot, int32_t n, int32_t h, int32_t w, int32_t c,
                                     int32_t k, int32_t stride, int32_t pad, void* stream) {
    if (!x || !y || !slot) return EGR_ENULL;
    if (n <= 0 || c % 4 != 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0 || 2 * pad > k) return EGR_EINVAL;
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    const int64_t total = (int64_t)n * ho * wo * (c / 4);
    if (total <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(maxpool_train_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, slot, n, h, w, c / 4,
                       ho, wo, k, stride, pad);
    return egr_launch_status();
}

extern "C" int egr_maxpool_bwd_f32(const float* dy, const uint8_t* slot, float* dx, int32_t n, int32_t h, int32_t w, int32_t c,
                                   int32_t k, int32_t stride, int32_t pad, void* stream) {
    if (!dy || !slot || !dx) return EGR_ENULL;
    if (n <= 0 || c % 4 != 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0 || 2 * pad > k) return EGR_EINVAL;
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    const int64_t total = (int64_t)n * h * w * (c / 4);
    if (total <= 0 || ho <= 0 || wo <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, dy, slot, dx, n, h, w, c / 4,
                       ho, wo, k, stride, pad);
    return egr_launch_status();
}

extern "C" int egr_upsample2x_bwd_f32(const float* dy, const float* y, float* dx, int32_t n, int32_t h, int32_t w, int32_t c,
                                      void* stream) {
    if (!dy || !dx) return EGR_ENULL;
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 4 != 0) return EGR_EINVAL;
    const int64_t total = (int64_t)n * h * w * (c / 4);
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, h, w, c / 4);
    return egr_launch_status();
}

extern "C" int egr_planes_to_nhwc_f32(const float* planes, int32_t n_inner, int64_t stride_inner, int64_t stride_outer, float* y,
                                      int32_t n, int32_t c, int32_t hw, int32_t cpad, void* stream) {
    if (!planes || !y) return EGR_ENULL;
    if (n <= 0 || c <= 0 || hw <= 0 || cpad < c || n_inner <= 0) return EGR_EINVAL;
    egr_nmap map{n_inner, stride_inner, stride_outer};
    hipLaunchKernelGGL(planes_to_nhwc_kernel, dim3(nblocks((int64_t)n * hw * cpad)), dim3(256), 0, (hipStream_t)stream, planes,
                       map, y, n, c, hw, cpad);
    return egr_launch_status();
}

extern "C" int egr_nhwc_to_planes_f32(const float* x, float* planes, int32_t n_inner, int64_t stride_inner, int64_t stride_outer,
                                      int32_t n, int32_t c, int32_t hw, int32_t cpad, void* stream) {
    if (!planes || !x) return EGR_ENULL;
    if (n <= 0 || c <= 0 || hw <= 0 || cpad < c || n_inner <= 0) return EGR_EINVAL;
    egr_nmap map{n_inner, stride_inner, stride_outer};
    hipLaunchKernelGGL(nhwc_to_planes_kernel, dim3(nblocks((int64_t)n * c * hw)), dim3(256), 0, (hipStream_t)stream, x, planes, map,
                       n, c, hw, cpad);
    return egr_launch_status();
}

extern "C" int egr_stem_im2col_f32(const float* img, int32_t n_inner, int64_t stride_inner, int64_t stride_outer, int32_t n,
                                   int32_t h, int32_t w, float* cols, void* stream) {
    if (!img || !cols) return EGR_ENULL;
    if (n <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || n_inner <= 0) return EGR_EINVAL;
    egr_nmap map{n_inner, stride_inner, stride_outer};
    const int64_t total = (int64_t)n * (h / 2) * (w / 2) * 160;
    if (total >= (1LL << 31) * 256) return EGR_EINVAL;
    hipLaunchKernelGGL(stem_im2col_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, img, map, n, h, w, cols);
    return egr_launch_status();
}

extern "C" int egr_layernorm_bwd_f32(const float* dy, const float* pre, const float* gamma, float* ds, float* dgamma,
                                     float* dbeta, float* rowstats, int32_t rows, int32_t c, float eps, int32_t rows_per_group,
                                     void* stream) {
    if (!dy || !pre || !gamma || !ds || !dgamma || !dbeta || !rowstats) return EGR_ENULL;
    if (rows <= 0) return EGR_EINVAL;
    const int rpg = rows_per_group > 0 ? rows_per_group : rows;
    if (rows % rpg != 0) return EGR_EINVAL;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (c) {
        case 64: hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, block, 0, s, dy, pre, gamma, ds, rowstats, rows, eps, rows_per_group); break;
        case 128: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, block, 0, s, dy, pre, gamma, ds, rowstats, rows, eps, rows_per_group); break;
        case 256: hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, block, 0, s, dy, pre, gamma, ds, rowstats, rows, eps, rows_per_group); break;
        case 512: hipLaunchKernelGGL(layernorm_bwd_kernel<8>, grid, block, 0, s, dy, pre, gamma, ds, rowstats, rows, eps, rows_per_group); break;
        default: return EGR_EINVAL;
    }
    hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3((c + 15) / 16, rows / rpg), dim3(256), 0, s, dy, pre, rowstats, dgamma,
                       dbeta, rpg, c);
    return egr_launch_status();
}

extern "C" int egr_joint_mha_bwd_f32(const float* qkv, const float* dout, float* dqkv, int32_t b, int32_t j, int32_t heads,
                                     int32_t d, float scale, void* stream) {
    if (!qkv || !dout || !dqkv) return EGR_ENULL;
    if (b <= 0 || j <= 0 || j > 16 || heads <= 0 || d <= 0 || d > 64) return EGR_EINVAL;
    hipLaunchKernelGGL(joint_mha_bwd_kernel, dim3((unsigned)(b * heads)), dim3(64), 0, (hipStream_t)stream, qkv, dout, dqkv, j,
                       heads, d, scale);
    return egr_launch_status();
}

extern "C" int egr_colsum_f32(const float* x, int64_t ld, int64_t rows, int32_t c, const float* scale, float* out,
                              int32_t accumulate, int32_t groups, int64_t gx, int64_t gs, int32_t cols_per_scale, int64_t scale_stride,
                              void* stream) {
    if (!x || !out) return EGR_ENULL;
    if (rows <= 0 || c <= 0 || ld < c || groups <= 0 || groups > 65535) return EGR_EINVAL;
    hipLaunchKernelGGL(colsum_kernel, dim3((c + 15) / 16, groups), dim3(256), 0, (hipStream_t)stream, x, ld, rows, c, scale, out,
                       accumulate, gx, gs, cols_per_scale, scale_stride);
    return egr_launch_status();
}

extern "C" int egr_fold_rows_f32(const float* x, float* y, int64_t rows_out, int32_t fold, int32_t c, void* stream) {
    if (!x || !y) return EGR_ENULL;
    if (rows_out <= 0 || fold <= 0 || c <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(fold_rows_kernel, dim3(nblocks(rows_out * c)), dim3(256), 0, (hipStream_t)stream, x, y, rows_out * c, fold, c);
    return egr_launch_status();
}

extern "C" int egr_jqa_sum_bwd_f32(const float* dx, float* d_embed, float* d_bfb, int32_t b, int32_t j, int32_t c,
                                   int32_t b_per_group, void* stream) {
    if (!dx || !d_embed || !d_bfb) return EGR_ENULL;
    if (b <= 0 || j <= 0 || c <= 0 || b_per_group <= 0 || b % b_per_group != 0) return EGR_EINVAL;
    const int64_t total = (int64_t)(b / b_per_group) * j * c + (int64_t)b * c;
    hipLaunchKernelGGL(jqa_sum_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, dx, d_embed, d_bfb, b, j, c,
                       b_per_group);
    return egr_launch_status();
}

extern "C" int egr_rownorm_loss_f32(const float* pred, const float* gt, int64_t rows, int32_t d, int32_t inner, int64_t ld_pred,
                                    int64_t ld_gt, float weight, double* loss, float* dpred, void* stream) {
    if (!pred || !gt || !loss) return EGR_ENULL;
    if (rows <= 0 || d <= 0 || d > 64 || inner <= 0 || ld_pred < (int64_t)inner * d || ld_gt < (int64_t)inner * d) return EGR_EINVAL;
    const float coef = weight / (float)rows;
    hipLaunchKernelGGL(rownorm_loss_kernel, dim3((unsigned)(((rows + 3) / 4) < 1024 ? ((rows + 3) / 4) : 1024)), dim3(256), 0, (hipStream_t)stream, pred, gt, rows, d,
                       inner, ld_pred, ld_gt, coef, loss, dpred);
    return egr_launch_status();
}

extern "C" int egr_mse_loss_f32(const float* pred, const float* gt, int64_t n, float weight, double* loss, float* dpred, void* stream) {
    if (!pred || !gt || !loss) return EGR_ENULL;
    if (n <= 0 || n % 4 != 0 || !aligned16(pred) || !aligned16(gt) || (dpred && !aligned16(dpred))) return EGR_EINVAL;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mse_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, gt, n / 4, weight / (float)n, loss, dpred);
    return egr_launch_status();
}

extern "C" int egr_sumsq_f32(const float* g, int64_t n, double* out, int32_t accumulate, void* stream) {
    if (!g || !out) return EGR_ENULL;
    if (n <= 0 || n % 4 != 0 || !aligned16(g)) return EGR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) hipLaunchKernelGGL(zero_double_kernel, dim3(1), dim3(1), 0, s, out);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, n / 4, out);
    return egr_launch_status();
}

extern "C" int egr_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int32_t step, const double* grad_sumsq, float clip, void* stream) {
    if (!p || !g || !m || !v) return EGR_ENULL;
    if (n <= 0 || n % 4 != 0 || step <= 0 || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return EGR_EINVAL;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2_sqrt, grad_sumsq, clip, (const float*)nullptr);
    return egr_launch_status();
}

extern "C" int egr_set4_f32(float* dst, float a, float b, float c, float d, void* stream) {
    if (!dst) return EGR_ENULL;
    hipLaunchKernelGGL(set4_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, a, b, c, d);
    return egr_launch_status();
}

extern "C" int egr_adamw_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                                 float beta2, float eps, float weight_decay, const double* grad_sumsq, float clip, void* stream) {
    if (!p || !g || !m || !v || !hyper) return EGR_ENULL;
    if (n <= 0 || n % 4 != 0 || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return EGR_EINVAL;
    hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, 0.f, beta1, beta2,
                       eps, weight_decay, 1.f, 1.f, grad_sumsq, clip, hyper);
    return egr_launch_status();
}
