// HBM-bound helper kernels of the hot path: pooling, bilinear upsampling, LayerNorm, argmax,
// tiny dense layers.  All are channels-last with 16-byte vector accesses across channels
// (coalesced: consecutive lanes touch consecutive channel quads of one pixel), one pass over the data.
#include "egr_common.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------ MaxPool2d (NHWC)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* x, float* y, int n, int h, int w, int c4, int ho,
                                                      int wo, int k, int stride, int pad) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)n * ho * wo * c4;
    if (idx >= total) return;
    int cq = (int)(idx % c4);
    int64_t p = idx / c4;
    int ox = (int)(p % wo);
    p /= wo;
    int oy = (int)(p % ho);
    int img = (int)(p / ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dy = 0; dy < k; ++dy) {
        int iy = oy * stride - pad + dy;
        if (iy < 0 || iy >= h) continue;
        for (int dx = 0; dx < k; ++dx) {
            int ix = ox * stride - pad + dx;
            if (ix < 0 || ix >= w) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(x + (((int64_t)img * h + iy) * w + ix) * (c4 * 4) + cq * 4);
            best[0] = fmaxf(best[0], v[0]);
            best[1] = fmaxf(best[1], v[1]);
            best[2] = fmaxf(best[2], v[2]);
            best[3] = fmaxf(best[3], v[3]);
        }
    }
    *reinterpret_cast<f32x4*>(y + idx * 4) = best;
}

// ------------------------------------------------------------------ bilinear x2, align_corners=True (NHWC)
// index / weight arithmetic follows ATen's area_pixel_compute_source_index(align_corners=True):
// src = dst * (in-1)/(out-1) in fp32, i0 = (int)src, i1 = min(i0+1, in-1), l1 = src - i0, l0 = 1 - l1.
// Two output rows (img = blockIdx.z, oy = 2 blockIdx.y, + 1) per block row: the vertical source rows / weights are wave-uniform,
// the thread index only splits into (ox, channel quad) with 32-bit arithmetic (a flat 64-bit index cost three 64-bit divisions
// per thread - more instructions than the memory system needed time: 232 us for 0.67 GB), and a thread's two outputs share their
// horizontal neighbours' columns and weights - up to eight 16-byte loads in flight per thread before the first use.
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* x, int ldx, float* y, int ldy, int n, int h, int w,
                                                         int c4, int c4_shift, int relu, int nt) {
    const int ho = 2 * h, wo = 2 * w;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= wo * c4) return;
    const int ox = c4_shift >= 0 ? (t >> c4_shift) : t / c4;
    const int cq = t - ox * c4;
    const int oy = 2 * blockIdx.y, img = blockIdx.z;
    const float sh = (ho > 1) ? (float)(h - 1) / (float)(ho - 1) : 0.f;
    const float sw = (wo > 1) ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    const float fx = sw * (float)ox;
    const int x0 = (int)fx, x1 = min(x0 + 1, w - 1);
    const float lx1 = fminf(fmaxf(fx - (float)x0, 0.f), 1.f), lx0 = 1.f - lx1;
    const float* base = x + (int64_t)img * h * w * ldx + cq * 4;
    f32x4 v[2][4];
    float ly0[2], ly1[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float fy = sh * (float)(oy + r);
        const int y0 = (int)fy, y1 = min(y0 + 1, h - 1);
        ly1[r] = fminf(fmaxf(fy - (float)y0, 0.f), 1.f);
        ly0[r] = 1.f - ly1[r];
        v[r][0] = *reinterpret_cast<const f32x4*>(base + (y0 * w + x0) * ldx);
        v[r][1] = *reinterpret_cast<const f32x4*>(base + (y0 * w + x1) * ldx);
        v[r][2] = *reinterpret_cast<const f32x4*>(base + (y1 * w + x0) * ldx);
        v[r][3] = *reinterpret_cast<const f32x4*>(base + (y1 * w + x1) * ldx);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t_ = ly0[r] * (lx0 * v[r][0][i] + lx1 * v[r][1][i]) + ly1[r] * (lx0 * v[r][2][i] + lx1 * v[r][3][i]);
            o[i] = (relu && t_ < 0.f) ? 0.f : t_;
        }
        // outputs far beyond the last-level cache: non-temporal stores (no write-allocate; 171 -> 103 us for the 537 MB launch)
        if (nt) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(y + (((int64_t)img * ho + oy + r) * wo + ox) * ldy + cq * 4));
        else *reinterpret_cast<f32x4*>(y + (((int64_t)img * ho + oy + r) * wo + ox) * ldy + cq * 4) = o;
    }
}

// ------------------------------------------------------------------ global average pool (NHWC)
__global__ __launch_bounds__(256) void avgpool_kernel(const float* x, float* y, int n, int hw, int c) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * c) return;
    int img = (int)(idx / c), ch = (int)(idx % c);
    float s = 0.f;
    const float* xp = x + (int64_t)img * hw * c + ch;
    int p = 0;
    for (; p + 8 <= hw; p += 8) {        // eight loads in flight, summed in pixel order (the same chain as the plain loop)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[(int64_t)(p + u) * c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < hw; ++p) s += xp[(int64_t)p * c];
    y[idx] = s / (float)hw;
}

// ------------------------------------------------------------------ LayerNorm (+ residual), one wave per row
template <int VPL>  // values per lane = c / 64
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* res, const float* gamma,
                                                        const float* beta, float* y, int rows, float eps, int rpg) {
    const int c = VPL * 64;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    if (rpg > 0) {  // grouped: per-group affine parameters
        int g = row / rpg;
        gamma += g * c;
        beta += g * c;
    }
    float v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int ch = i * 64 + lane;
        float t = x[(int64_t)row * c + ch];
        if (res) t += res[(int64_t)row * c + ch];
        v[i] = t;
        s += t;
    }
    float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float dlt = v[i] - mean;
        q += dlt * dlt;
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) / (float)c + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int ch = i * 64 + lane;
        y[(int64_t)row * c + ch] = (v[i] - mean) * rstd * gamma[ch] + beta[ch];
    }
}

// ------------------------------------------------------------------ argmax over hw, one wave per row
__global__ __launch_bounds__(256) void argmax_kernel(const float* hm, int rows, int hgt, int wid, float thr,
                                                     float* anchors, float* maxvals, uint8_t* valid, int32_t* index) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int hw = hgt * wid;
    const float* p = hm + (int64_t)row * hw;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    // each lane walks its indices in increasing order, so '>' keeps the first maximum it sees
    for (int base = lane * 4; base < hw; base += 256) {
        if (base + 3 < hw) {
            f32x4 v = *reinterpret_cast<const f32x4*>(p + base);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (v[i] > best || bidx == 0x7fffffff) { best = v[i]; bidx = base + i; }
        } else {
            for (int i = 0; i < 4 && base + i < hw; ++i) {
                float v = p[base + i];
                if (v > best || bidx == 0x7fffffff) { best = v; bidx = base + i; }
            }
        }
    }
    // cross-lane: larger value wins, ties go to the smaller flat index (torch.max returns the first)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) {
        anchors[row * 2 + 0] = (float)(bidx % wid) / (float)wid;
        anchors[row * 2 + 1] = (float)(bidx / wid) / (float)hgt;
        maxvals[row] = best;
        valid[row] = best >= thr ? 1 : 0;
        index[row] = bidx;
    }
}

// ------------------------------------------------------------------ tiny dense layer, K not a multiple of 32
// One thread per output.  The weight rows are k floats apart: read straight from global memory a wave's 64 columns touch ~30
// cache lines per k step (74 us for the 65536 x 64 x 15 head layer, all of it in the texture path).  With k <= 16 and n <= 256
// the group's filter is staged once per workgroup, transposed ([k][n]: consecutive lanes, consecutive banks), and a workgroup
// walks SK_ROWS rows, their x values staged as well.  Used from 16384 rows (small launches have too few workgroups for it:
// 6 -> 27 us at 1024 rows).  Same left-to-right fma chain either way.
constexpr int SK_ROWS = 64, SK_MAXK = 16, SK_MAXN = 256;
__global__ __launch_bounds__(256) void linear_smallk_lds_kernel(const float* x, int64_t sxm, int64_t sxk, const float* w, const float* bias,
                                                                float* y, int m, int n, int k, int act, int rpg) {
    __shared__ float s_w[SK_MAXK * SK_MAXN];
    __shared__ float s_b[SK_MAXN];
    __shared__ float s_x[SK_ROWS * SK_MAXK];
    const int row0 = blockIdx.x * SK_ROWS;              // a workgroup's rows lie inside one group (the host checks rpg % SK_ROWS == 0)
    if (rpg > 0) {
        const int g = row0 / rpg;
        w += (int64_t)g * n * k;
        if (bias) bias += g * n;
    }
    for (int i = threadIdx.x; i < n * k; i += 256) {
        const int col = i / k, kk = i - col * k;
        s_w[kk * n + col] = w[i];
    }
    for (int i = threadIdx.x; i < n; i += 256) s_b[i] = bias ? bias[i] : 0.f;
    const int rows = min(SK_ROWS, m - row0);
    for (int i = threadIdx.x; i < rows * k; i += 256) {
        const int r = i / k, kk = i - r * k;
        s_x[r * SK_MAXK + kk] = x[(int64_t)(row0 + r) * sxm + kk * sxk];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < rows * n; idx += 256) {
        const int r = idx / n, col = idx - r * n;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < SK_MAXK; ++i)
            if (i < k) s = fmaf(s_x[r * SK_MAXK + i], s_w[i * n + col], s);
        y[(int64_t)(row0 + r) * n + col] = egr_act(s + (bias ? s_b[col] : 0.f), act);
    }
}

__global__ __launch_bounds__(256) void linear_smallk_kernel(const float* x, int64_t sxm, int64_t sxk, const float* w,
                                                            const float* bias, float* y, int m, int n, int k, int act, int rpg) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)m * n) return;
    int row = (int)(idx / n), col = (int)(idx % n);
    if (rpg > 0) {
        int g = row / rpg;
        w += (int64_t)g * n * k;
        if (bias) bias += g * n;
    }
    float s = 0.f;
    for (int i = 0; i < k; ++i) s = fmaf(x[row * sxm + i * sxk], w[(int64_t)col * k + i], s);
    if (bias) s += bias[col];
    y[idx] = egr_act(s, act);
}

__global__ __launch_bounds__(256) void jqa_sum_kernel(const float* hm_embed, const float* embed, const float* bfb, float* y,
                                                      int b, int j, int c, int bpg) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)b * j * c) return;
    int ch = (int)(idx % c);
    int64_t r = idx / c;
    int jj = (int)(r % j), bb = (int)(r / j);
    if (bpg > 0) embed += (int64_t)(bb / bpg) * j * c;
    // (joint_query_embed + bfb) + heatmap_embed, the reference's association order
    y[idx] = (embed[jj * c + ch] + bfb[(int64_t)bb * c + ch]) + hm_embed[idx];
}

__global__ __launch_bounds__(256) void tokens_to_nhwc_kernel(const float* x, float* y, int b, int j, int hw) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)b * j * hw) return;
    int jj = (int)(idx % j);
    int64_t r = idx / j;
    int p = (int)(r % hw), bb = (int)(r / hw);
    y[idx] = x[((int64_t)bb * j + jj) * hw + p];
}


// ------------------------------------------------------------------ heat-map head tail: up x2 (+ReLU) -> 1x1 conv (cin -> <=16) -> planes
// The last two steps of every heat-map head (heatmap_mvf_ex.py:101-126, 570-584) in one pass: the 128-channel tensor at
// 64 x 64 that would sit between them (written by the upsample, read back by a 15-output conv that is pure HBM traffic)
// never exists.  A workgroup owns 8 x 32 output pixels and stages their (<= 6 x 18) low-resolution source pixels in LDS
// (pixel stride cin + 4 floats: the 16 pixels of an MFMA group hit different banks).  The 1x1 conv runs on
// v_mfma_f32_16x16x4_f32 with N = 16 output channels: lane l supplies a[pixel l%16][k = l/16] of step s = channel
// 16 (s/4) + 4 (l/16) + s%4, computed on the fly (four 16-byte source reads per 16-channel block, the upsample kernel's
// interpolation arithmetic, ReLU), and b from registers (the filter,
// 32 floats per lane, loaded once); D[pixel][co] leaves as one 16-byte store of 4 consecutive pixels per lane.
#ifndef HT_ROWS
#define HT_ROWS 8     // (4-row tiles, four workgroups per CU instead of two: 141 us per launch against 119)
#endif
constexpr int HT_H = HT_ROWS, HT_W = 32, HT_SRC_H = HT_ROWS / 2 + 2, HT_SRC_W = 18, HT_MAXCO = 16, HT_MAXCIN = 128;
constexpr int HT_RPW = HT_H / 4, HT_NG = 2 * HT_RPW;       // output rows / 16-pixel groups per wave
typedef float f32x4_mfma __attribute__((ext_vector_type(4)));

// CIN > 0: the channel count at compile time (the path's heads: 128) - the 32-step K loop is straight-line code the compiler
// can software-pipeline (with a run-time trip count it stayed a rolled loop of read -> wait -> interpolate -> MFMA: 160 us
// per launch).  CIN == 0: any multiple of 16 up to HT_MAXCIN.
template <int CIN>
__global__ __launch_bounds__(256) void up2_relu_head_kernel(const float* lo, int h, int w, int cin_rt, const float* wgt, const float* bias,
                                                            int cout, float* planes, egr_nmap map, int npg, int64_t gy) {
    extern __shared__ __attribute__((aligned(16))) float s_src[];   // [HT_SRC_H * HT_SRC_W][cin + 4]
    const int cin = CIN > 0 ? CIN : cin_rt;
    const int ld = cin + 4;
    const int ho = 2 * h, wo = 2 * w;
    const int tiles_x = wo / HT_W, tiles_y = ho / HT_H;
    int t = blockIdx.x;
    const int img = t / (tiles_x * tiles_y);
    t -= img * tiles_x * tiles_y;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * HT_H, ox0 = tx * HT_W;
    const int grp = img / npg;
    const float sh = (ho > 1) ? (float)(h - 1) / (float)(ho - 1) : 0.f;
    const float sw = (wo > 1) ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    const int sy0 = (int)(sh * (float)oy0), sx0 = (int)(sw * (float)ox0);   // first source row / column of the tile
    const int c4n = cin >> 2;
    const float* src = lo + (int64_t)img * h * w * cin;
    // all of a thread's source loads in flight at once, then the LDS writes (a load -> store loop waited for every load in turn:
    // 14 memory latencies per workgroup, most of the kernel's time)
    constexpr int NLD = (HT_SRC_H * HT_SRC_W * (HT_MAXCIN / 4) + 255) / 256;
    f32x4 stg[NLD];
    const int nsrc = HT_SRC_H * HT_SRC_W * c4n;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int ii = i < nsrc ? i : 0;
        const int cq = ii % c4n, p = ii / c4n;
        const int py = p / HT_SRC_W, px = p - py * HT_SRC_W;
        const int iy = min(sy0 + py, h - 1), ix = min(sx0 + px, w - 1);
        stg[u] = *reinterpret_cast<const f32x4*>(src + ((int64_t)iy * w + ix) * cin + cq * 4);
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < nsrc) {
            const int cq = i % c4n, p = i / c4n;
            *reinterpret_cast<f32x4*>(&s_src[p * ld + cq * 4]) = stg[u];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pj = lane & 15, kq = lane >> 4;                   // MFMA roles: A row / B column pj, k index kq
    // K order: step s = 4j + r multiplies channel 16j + 4kq + r of lane kq (any bijection of (step, kq) onto the channels serves - the
    // sum runs over all of them): a lane's four steps of a block j are FOUR CONSECUTIVE channels, so every source corner is one 16-byte
    // LDS read per block instead of four 4-byte ones (round 5: the loop was bound by LDS instruction issue, not bytes)
    // filter: b[s] = W[co = pj][channel 16 (s / 4) + 4 kq + s % 4] (zero rows above cout)
    const float* wg = wgt + (int64_t)grp * cout * cin;
    float bw[HT_MAXCIN / 4];
    const int ksteps = cin >> 2;
#pragma unroll
    for (int s_ = 0; s_ < HT_MAXCIN / 4; ++s_) bw[s_] = (s_ < ksteps && pj < cout) ? wg[pj * cin + 16 * (s_ >> 2) + 4 * kq + (s_ & 3)] : 0.f;
    __syncthreads();

    const float* bg = bias ? bias + grp * cout : nullptr;
    const float bco = (bg && pj < cout) ? bg[pj] : 0.f;        // D column = co = pj
    float* outg = planes + grp * gy + egr_map(map, img - grp * npg);
    // wave: HT_RPW rows, 2 groups of 16 columns per row = independent accumulator chains advanced together
    const float* q00[HT_NG]; const float* q01[HT_NG]; const float* q10[HT_NG]; const float* q11[HT_NG];
    float wy0[HT_NG], wy1[HT_NG], wx0[HT_NG], wx1[HT_NG];
#pragma unroll
    for (int mg = 0; mg < HT_NG; ++mg) {
        const int oy = oy0 + HT_RPW * wave + (mg >> 1);
        const int ox = ox0 + (mg & 1) * 16 + pj;                // this lane's A-row pixel
        // ATen's align_corners=True source index / weights, as in upsample2x_kernel
        const float fy = sh * (float)oy, fx = sw * (float)ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        wy1[mg] = fminf(fmaxf(fy - (float)y0, 0.f), 1.f); wx1[mg] = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
        wy0[mg] = 1.f - wy1[mg]; wx0[mg] = 1.f - wx1[mg];
        q00[mg] = s_src + ((y0 - sy0) * HT_SRC_W + (x0 - sx0)) * ld + 4 * kq;
        q01[mg] = s_src + ((y0 - sy0) * HT_SRC_W + (x1 - sx0)) * ld + 4 * kq;
        q10[mg] = s_src + ((y1 - sy0) * HT_SRC_W + (x0 - sx0)) * ld + 4 * kq;
        q11[mg] = s_src + ((y1 - sy0) * HT_SRC_W + (x1 - sx0)) * ld + 4 * kq;
    }
    f32x4_mfma acc[HT_NG];
#pragma unroll
    for (int mg = 0; mg < HT_NG; ++mg) acc[mg] = f32x4_mfma{0.f, 0.f, 0.f, 0.f};
    // the four source quads of every chain, one 16-channel block ahead of their use (pinned: left to itself the scheduler consumed
    // each LDS read right behind its issue and the wave sat through the LDS latency every block)
    f32x4 sv[2][HT_NG][4];
    auto fetch = [&](int j, int par) {
#pragma unroll
        for (int mg = 0; mg < HT_NG; ++mg) {
            sv[par][mg][0] = *reinterpret_cast<const f32x4*>(q00[mg] + 16 * j);
            sv[par][mg][1] = *reinterpret_cast<const f32x4*>(q01[mg] + 16 * j);
            sv[par][mg][2] = *reinterpret_cast<const f32x4*>(q10[mg] + 16 * j);
            sv[par][mg][3] = *reinterpret_cast<const f32x4*>(q11[mg] + 16 * j);
        }
    };
    const int kblocks = ksteps >> 2;          // (cin % 16 == 0 is checked by the host)
    fetch(0, 0);
#pragma unroll
    for (int j = 0; j < HT_MAXCIN / 16; ++j) {
        if (j < kblocks) {
            if (j + 1 < kblocks) fetch(j + 1, (j + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 u[HT_NG];
#pragma unroll
            for (int mg = 0; mg < HT_NG; ++mg) {
                const f32x4* v = sv[j & 1][mg];
                u[mg] = wy0[mg] * (wx0[mg] * v[0] + wx1[mg] * v[1]) + wy1[mg] * (wx0[mg] * v[2] + wx1[mg] * v[3]);
            }
            // ReLU as one v_max (NaN -> 0, like the `t > 0 ? t : 0` of the conv epilogues); the four steps of a chain are dependent
            // MFMAs: issued step by step over the chains, a chain's next MFMA is four instructions away
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mg = 0; mg < HT_NG; ++mg)
                    acc[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaxf(u[mg][r], 0.f), bw[4 * j + r], acc[mg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D[row = 4*kq + r][col = pj]: rows are the pixels oxg + 4*kq + r (contiguous in a plane), col the output channel
    if (pj < cout) {
#pragma unroll
        for (int mg = 0; mg < HT_NG; ++mg) {
            const int oy = oy0 + HT_RPW * wave + (mg >> 1);
            const int oxg = ox0 + (mg & 1) * 16;
            f32x4 v = {acc[mg][0] + bco, acc[mg][1] + bco, acc[mg][2] + bco, acc[mg][3] + bco};
            *reinterpret_cast<f32x4*>(outg + (int64_t)pj * ho * wo + (int64_t)oy * wo + oxg + 4 * kq) = v;
        }
    }
}

// Round 5: the same pass as a PERSISTENT workgroup of eight waves (one output row each) over the same 8 x 32 output pixels.  The tile kernel above spends most of a workgroup's life waiting for its source pixels (a workgroup lives
// ~15 us of which ~3 are arithmetic: two per CU cannot cover that); here the NEXT tile's source pixels are requested into registers
// before the current tile's arithmetic and parked in LDS behind it - the load latency runs under the matrix instructions.
// Same arithmetic per output (same interpolation expression, same K order, same MFMA chain): bit-identical to the tile kernel.
constexpr int PT_H = 8, PT_W = 32, PT_SRC_H = PT_H / 2 + 2, PT_SRC_W = 18, PT_NG = 2;   // a wave: one row = 2 groups of 16 columns
__global__ __launch_bounds__(512) void up2_relu_head_p_kernel(const float* lo, int h, int w, const float* wgt, const float* bias, int cout,
                                                              float* planes, egr_nmap map, int npg, int64_t gy, int ntiles) {
    constexpr int CIN = 128, LD = CIN + 4, C4N = CIN / 4;
    extern __shared__ __attribute__((aligned(16))) float s_src[];   // [PT_SRC_H * PT_SRC_W][LD]
    const int ho = 2 * h, wo = 2 * w;
    const int tiles_x = wo / PT_W, tiles_y = ho / PT_H;
    const float sh = (ho > 1) ? (float)(h - 1) / (float)(ho - 1) : 0.f;
    const float sw = (wo > 1) ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    constexpr int NSRC = PT_SRC_H * PT_SRC_W * C4N;
    constexpr int NLD = (NSRC + 511) / 512;
    f32x4 stg[NLD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pj = lane & 15, kq = lane >> 4;
    auto decode = [&](int t, int& img, int& oy0, int& ox0) {
        img = t / (tiles_x * tiles_y);
        t -= img * tiles_x * tiles_y;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        oy0 = ty * PT_H;
        ox0 = tx * PT_W;
    };
    auto request = [&](int t) {
        int img, oy0, ox0;
        decode(t, img, oy0, ox0);
        const int sy0 = (int)(sh * (float)oy0), sx0 = (int)(sw * (float)ox0);
        const float* src = lo + (int64_t)img * h * w * CIN;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = threadIdx.x + 512 * u;
            const int ii = i < NSRC ? i : 0;
            const int cq = ii % C4N, p = ii / C4N;
            const int py = p / PT_SRC_W, px = p - py * PT_SRC_W;
            const int iy = min(sy0 + py, h - 1), ix = min(sx0 + px, w - 1);
            stg[u] = *reinterpret_cast<const f32x4*>(src + ((int64_t)iy * w + ix) * CIN + cq * 4);
        }
    };
    float bw[CIN / 4];
    float bco = 0.f;
    int cur_grp = -1;
    int tile = blockIdx.x;
    if (tile < ntiles) request(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = threadIdx.x + 512 * u;
            if (i < NSRC) {
                const int cq = i % C4N, p = i / C4N;
                *reinterpret_cast<f32x4*>(&s_src[p * LD + cq * 4]) = stg[u];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);      // in flight under this tile's arithmetic
        int img, oy0, ox0;
        decode(tile, img, oy0, ox0);
        const int grp = img / npg;
        if (grp != cur_grp) {          // (workgroup-uniform) the group's filter: b[s] = W[co = pj][channel 16 (s / 4) + 4 kq + s % 4]
            cur_grp = grp;
            const float* wg = wgt + (int64_t)grp * cout * CIN;
#pragma unroll
            for (int s_ = 0; s_ < CIN / 4; ++s_) bw[s_] = pj < cout ? wg[pj * CIN + 16 * (s_ >> 2) + 4 * kq + (s_ & 3)] : 0.f;
            bco = (bias && pj < cout) ? bias[grp * cout + pj] : 0.f;
        }
        const int sy0 = (int)(sh * (float)oy0), sx0 = (int)(sw * (float)ox0);
        float* outg = planes + grp * gy + egr_map(map, img - grp * npg);
        const float* q00[PT_NG]; const float* q01[PT_NG]; const float* q10[PT_NG]; const float* q11[PT_NG];
        float wy0[PT_NG], wy1[PT_NG], wx0[PT_NG], wx1[PT_NG];
#pragma unroll
        for (int mg = 0; mg < PT_NG; ++mg) {
            const int oy = oy0 + wave;
            const int ox = ox0 + mg * 16 + pj;
            const float fy = sh * (float)oy, fx = sw * (float)ox;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
            wy1[mg] = fminf(fmaxf(fy - (float)y0, 0.f), 1.f); wx1[mg] = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
            wy0[mg] = 1.f - wy1[mg]; wx0[mg] = 1.f - wx1[mg];
            q00[mg] = s_src + ((y0 - sy0) * PT_SRC_W + (x0 - sx0)) * LD + 4 * kq;
            q01[mg] = s_src + ((y0 - sy0) * PT_SRC_W + (x1 - sx0)) * LD + 4 * kq;
            q10[mg] = s_src + ((y1 - sy0) * PT_SRC_W + (x0 - sx0)) * LD + 4 * kq;
            q11[mg] = s_src + ((y1 - sy0) * PT_SRC_W + (x1 - sx0)) * LD + 4 * kq;
        }
        f32x4_mfma acc[PT_NG];
#pragma unroll
        for (int mg = 0; mg < PT_NG; ++mg) acc[mg] = f32x4_mfma{0.f, 0.f, 0.f, 0.f};
        f32x4 sv[2][PT_NG][4];
        auto fetch = [&](int j, int par) {
#pragma unroll
            for (int mg = 0; mg < PT_NG; ++mg) {
                sv[par][mg][0] = *reinterpret_cast<const f32x4*>(q00[mg] + 16 * j);
                sv[par][mg][1] = *reinterpret_cast<const f32x4*>(q01[mg] + 16 * j);
                sv[par][mg][2] = *reinterpret_cast<const f32x4*>(q10[mg] + 16 * j);
                sv[par][mg][3] = *reinterpret_cast<const f32x4*>(q11[mg] + 16 * j);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int j = 0; j < CIN / 16; ++j) {
            if (j + 1 < CIN / 16) fetch(j + 1, (j + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 u[PT_NG];
#pragma unroll
            for (int mg = 0; mg < PT_NG; ++mg) {
                const f32x4* v = sv[j & 1][mg];
                u[mg] = wy0[mg] * (wx0[mg] * v[0] + wx1[mg] * v[1]) + wy1[mg] * (wx0[mg] * v[2] + wx1[mg] * v[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mg = 0; mg < PT_NG; ++mg)
                    acc[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaxf(u[mg][r], 0.f), bw[4 * j + r], acc[mg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pj < cout) {
#pragma unroll
            for (int mg = 0; mg < PT_NG; ++mg) {
                const int oy = oy0 + wave;
                const int oxg = ox0 + mg * 16;
                f32x4 v = {acc[mg][0] + bco, acc[mg][1] + bco, acc[mg][2] + bco, acc[mg][3] + bco};
                *reinterpret_cast<f32x4*>(outg + (int64_t)pj * ho * wo + (int64_t)oy * wo + oxg + 4 * kq) = v;
            }
        }
        __syncthreads();               // every wave is done with the tile's source pixels before the next ones are parked
    }
}

inline unsigned nblocks(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int egr_maxpool_nhwc_f32(const float* x, float* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t k,
                                    int32_t stride, int32_t pad, void* stream) {
    if (!x || !y) return EGR_ENULL;
    if (n <= 0 || c % 4 != 0 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k) return EGR_EINVAL;
    int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    int64_t total = (int64_t)n * ho * wo * (c / 4);
    if (total <= 0 || total >= (1LL << 31) * 256) return EGR_EINVAL;
    hipLaunchKernelGGL(maxpool_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, n, h, w, c / 4, ho,
                       wo, k, stride, pad);
    return egr_launch_status();
}

extern "C" int egr_upsample2x_nhwc_f32(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t n, int32_t h,
                                       int32_t w, int32_t c, int32_t relu, void* stream) {
    if (!x || !y) return EGR_ENULL;
    if (n <= 0 || h <= 0 || w <= 0 || c % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || ldx < c || ldy < c) return EGR_EINVAL;
    const int c4 = c / 4;
    int c4_shift = 0;
    while ((1 << c4_shift) < c4) ++c4_shift;
    if ((1 << c4_shift) != c4) c4_shift = -1;
    if (2 * h > 65535 || (int64_t)h * w * ldx >= (1LL << 31) || (int64_t)2 * w * c4 >= (1LL << 31)) return EGR_EINVAL;
    static const double nt_mb = getenv("EGR_NT_STORE_MB") ? atof(getenv("EGR_NT_STORE_MB")) : 256.0;
    const int nt = ((double)n * 4 * h * w * c * 4.0 >= nt_mb * 1e6) ? 1 : 0;
    for (int n0 = 0; n0 < n; n0 += 65535) {       // grid.z is limited to 65535 images per launch
        const int nn = n - n0 < 65535 ? n - n0 : 65535;
        hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((2 * w * c4 + 255) / 256), (unsigned)h, (unsigned)nn), dim3(256), 0,
                           (hipStream_t)stream, x + (int64_t)n0 * h * w * ldx, ldx, y + (int64_t)n0 * 4 * h * w * ldy, ldy, nn, h, w, c4, c4_shift, relu, nt);
    }
    return egr_launch_status();
}

static int g_head_persist = getenv("EGR_HEAD_PERSIST") ? atoi(getenv("EGR_HEAD_PERSIST")) != 0 : 1;
extern "C" int egr_head_set_persist(int on) {
    const int old = g_head_persist;
    if (on >= 0) g_head_persist = on != 0;
    return old;
}

extern "C" int egr_up2_relu_head_f32(const float* lo, int32_t n, int32_t h, int32_t w, int32_t cin, const float* wgt, const float* bias,
                                     int32_t cout, float* planes, int32_t n_inner, int64_t stride_inner, int64_t stride_outer,
                                     int32_t groups, int64_t gy, void* stream) {
    if (!lo || !wgt || !planes) return EGR_ENULL;
    if (n <= 0 || groups <= 0 || n % groups != 0 || h <= 0 || w <= 0 || cin <= 0 || cin % 16 != 0 || cin > HT_MAXCIN || cout <= 0 ||
        cout > HT_MAXCO || (2 * h) % HT_H != 0 || (2 * w) % HT_W != 0 || n_inner <= 0 || ((uintptr_t)planes & 15) ||
        ((stride_inner | stride_outer | gy) % 4 != 0))
        return EGR_EINVAL;
    egr_nmap map{n_inner, stride_inner, stride_outer};
    const size_t lds = (size_t)(HT_SRC_H * HT_SRC_W * (cin + 4)) * sizeof(float);
    const int64_t blocks = (int64_t)n * ((2 * h) / HT_H) * ((2 * w) / HT_W);
    if (blocks >= (1LL << 31) || lds > 64 * 1024) return EGR_EINVAL;
    // the persistent form (EGR_HEAD_PERSIST=0: the tile kernel): 128 channels, 16-row tiles, enough tiles to give every CU several
    if (g_head_persist && cin == 128 && (2 * h) % PT_H == 0 && (2 * w) % PT_W == 0) {
        const int64_t nt = (int64_t)n * ((2 * h) / PT_H) * ((2 * w) / PT_W);
        int dev = 0, cus = 0;
        if (nt >= 1024 && nt < (1LL << 31) && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
            const size_t lds_p = (size_t)(PT_SRC_H * PT_SRC_W * (128 + 4)) * sizeof(float);
            static bool allowed[64] = {};
            if (!allowed[dev]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(up2_relu_head_p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p) !=
                    hipSuccess)
                    return EGR_EINVAL;
                allowed[dev] = true;
            }
            const unsigned grid = (unsigned)(nt < cus ? nt : cus);
            hipLaunchKernelGGL(up2_relu_head_p_kernel, dim3(grid), dim3(512), lds_p, (hipStream_t)stream, lo, h, w, wgt, bias, cout, planes, map,
                               n / groups, gy, (int)nt);
            return egr_launch_status();
        }
    }
    if (cin == 128)
        hipLaunchKernelGGL(up2_relu_head_kernel<128>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, lo, h, w, cin, wgt, bias, cout,
                           planes, map, n / groups, gy);
    else
        hipLaunchKernelGGL(up2_relu_head_kernel<0>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, lo, h, w, cin, wgt, bias, cout,
                           planes, map, n / groups, gy);
    return egr_launch_status();
}

// abs-max record of a dense tensor (see egr_conv2d_nhwc_ex_f32): for tensors that reach the path from outside (no producing launch)
// Few, fat workgroups: the record's slots are agent-scope atomics that serialise per slot (about 0.6 us each, measured) - 1024
// workgroups spent more time in that tail than reading; 256 workgroups with eight 16-byte loads in flight per thread keep HBM busy
// and send four atomics per slot.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n4, int64_t n, unsigned* __restrict__ rec) {
    float m = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + 4 * (i + u * stride));
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
    }
    for (; i < n4; i += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));
    __shared__ float s_m[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        if (m > 0.f) __hip_atomic_fetch_max(rec + (blockIdx.x & 63), __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

extern "C" int egr_absmax_f32(const float* x, int64_t n, uint32_t* record, void* stream) {
    if (!x || !record) return EGR_ENULL;
    if (n <= 0 || ((uintptr_t)x & 15) || ((uintptr_t)record & 3)) return EGR_EINVAL;
    const int64_t n4 = n / 4;
    static const int max_blocks = getenv("EGR_ABSMAX_BLOCKS") ? atoi(getenv("EGR_ABSMAX_BLOCKS")) : 256;     // (tuning)
    int64_t blocks = (n4 + 2047) / 2048;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, n, record);
    return egr_launch_status();
}

extern "C" int egr_avgpool_nhwc_f32(const float* x, float* y, int32_t n, int32_t hw, int32_t c, void* stream) {
    if (!x || !y) return EGR_ENULL;
    if (n <= 0 || hw <= 0 || c <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(avgpool_kernel, dim3(nblocks((int64_t)n * c)), dim3(256), 0, (hipStream_t)stream, x, y, n, hw, c);
    return egr_launch_status();
}

extern "C" int egr_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                 int32_t rows, int32_t c, float eps, int32_t rows_per_group, void* stream) {
    if (!x || !gamma || !beta || !y) return EGR_ENULL;
    if (rows <= 0) return EGR_EINVAL;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (c) {
        case 64: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, res, gamma, beta, y, rows, eps, rows_per_group); break;
        case 128: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, res, gamma, beta, y, rows, eps, rows_per_group); break;
        case 256: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, res, gamma, beta, y, rows, eps, rows_per_group); break;
        case 512: hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, res, gamma, beta, y, rows, eps, rows_per_group); break;
        case 1024: hipLaunchKernelGGL(layernorm_kernel<16>, grid, block, 0, s, x, res, gamma, beta, y, rows, eps, rows_per_group); break;
        default: return EGR_EINVAL;
    }
    return egr_launch_status();
}

extern "C" int egr_argmax_rows_f32(const float* hm, int32_t rows, int32_t hgt, int32_t wid, float thr, float* anchors,
                                   float* maxvals, uint8_t* valid, int32_t* index, void* stream) {
    if (!hm || !anchors || !maxvals || !valid || !index) return EGR_ENULL;
    if (rows <= 0 || hgt <= 0 || wid <= 0 || (int64_t)hgt * wid >= (1 << 24) || ((hgt * wid) % 4 != 0)) return EGR_EINVAL;
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, hm, rows, hgt,
                       wid, thr, anchors, maxvals, valid, index);
    return egr_launch_status();
}

extern "C" int egr_linear_smallk_f32(const float* x, int64_t sxm, int64_t sxk, const float* w, const float* bias,
                                     float* y, int32_t m, int32_t n, int32_t k, int32_t act, int32_t rows_per_group,
                                     void* stream) {
    if (!x || !w || !y) return EGR_ENULL;
    if (m <= 0 || n <= 0 || k <= 0) return EGR_EINVAL;
    if (k <= SK_MAXK && n <= SK_MAXN && m >= 256 * SK_ROWS && (rows_per_group <= 0 || rows_per_group % SK_ROWS == 0))
        hipLaunchKernelGGL(linear_smallk_lds_kernel, dim3((unsigned)((m + SK_ROWS - 1) / SK_ROWS)), dim3(256), 0, (hipStream_t)stream, x, sxm,
                           sxk, w, bias, y, m, n, k, act, rows_per_group);
    else
        hipLaunchKernelGGL(linear_smallk_kernel, dim3(nblocks((int64_t)m * n)), dim3(256), 0, (hipStream_t)stream, x, sxm, sxk,
                           w, bias, y, m, n, k, act, rows_per_group);
    return egr_launch_status();
}

extern "C" int egr_jqa_sum_f32(const float* hm_embed, const float* embed, const float* bfb, float* y, int32_t b, int32_t j,
                               int32_t c, int32_t b_per_group, void* stream) {
    if (!hm_embed || !embed || !bfb || !y) return EGR_ENULL;
    if (b <= 0 || j <= 0 || c <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(jqa_sum_kernel, dim3(nblocks((int64_t)b * j * c)), dim3(256), 0, (hipStream_t)stream, hm_embed,
                       embed, bfb, y, b, j, c, b_per_group);
    return egr_launch_status();
}

extern "C" int egr_tokens_to_nhwc_f32(const float* x, float* y, int32_t b, int32_t j, int32_t hw, void* stream) {
    if (!x || !y) return EGR_ENULL;
    if (b <= 0 || j <= 0 || hw <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(tokens_to_nhwc_kernel, dim3(nblocks((int64_t)b * j * hw)), dim3(256), 0, (hipStream_t)stream, x, y,
                       b, j, hw);
    return egr_launch_status();
}

extern "C" const char* egr_version(void) { return "egorear_hip 0.3 (gfx950: fp32 MFMA, two-plane fp16 scheme, exact bf16x3 split)"; }

extern "C" int egr_device_arch(char* buf, int32_t buflen) {
    if (!buf || buflen <= 0) return EGR_ENULL;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    int i = 0;
    for (; i < buflen - 1 && prop.gcnArchName[i]; ++i) buf[i] = prop.gcnArchName[i];
    buf[i] = 0;
    return 0;
}
