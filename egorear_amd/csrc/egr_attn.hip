// Joint-query attention kernels: deformable sampling (sample-then-project form), the 15/16-token
// joint-to-joint attention core, and the fisheye anchor reprojection.
#include "egr_common.h"
#include "egr_fisheye.h"

namespace {

// 1: the gather reads feature rows by half waves and positional slices by quarter waves (EGR_GATHER_WIDE=0: one corner per instruction)
static const bool g_gather_wide_env = getenv("EGR_GATHER_WIDE") ? atoi(getenv("EGR_GATHER_WIDE")) != 0 : true;

#ifndef GATHER_UNROLL
#define GATHER_UNROLL 4    // (8 / 16 rows in flight measured no faster: the loop runs at ~60 % of the L1 rate, 21 TB/s of corner bytes at batch 64)
#endif
constexpr int NPTS = 16;  // n_points of MSDeformAttn (heatmap_mvf_ex.py:772, egoposeformer_mvf_ex.py:460)

// ------------------------------------------------------------------ deformable sampling
// One workgroup per (b, joint, view) row, one wave per head.  Lanes 0..15 own the 16 sampling points:
// they form the softmax over the logits with wave shuffles and the four bilinear corners (mmcv
// ms_deform_attn_im2col_bilinear: pixel = loc*size - 0.5, zero padding).  The wave then walks the
// (point, corner) list; every step is one fully coalesced read of a feature row (cf floats: 512 B for
// cf = 128, a float2 per lane) and, when a positional table is given, of the head's dh-float slice.
// The value projection is linear, so sampling first and projecting the 960 sampled rows per frame
// afterwards (egr_conv2d on g, with the per-row in-bounds mass sigma scaling the bias) gives the same
// result as projecting all 4096 tokens per view and then sampling — at ~1/70 of the FLOPs.
template <int CPL, bool POS, bool WIDE>  // feature channels per lane = cf / 64; POS: the positional table is read as well
__global__ __launch_bounds__(256) void msda_gather_kernel(const float* feat, const float* pos, int dh,
                                                          const float* offs_logits, const float* anchors,
                                                          const uint8_t* valid, int B, int V, int J, int heads, int hgt,
                                                          int wid, float* g, float* e, float* sigma, uint8_t* rowmask) {
    const int cf = CPL * 64;
    const int row = blockIdx.x;  // (b, j, v)
    {   // grouped launch: query sets (e.g. the four refiners) share feat / anchors / valid, own everything else
        const int grp = blockIdx.y;
        const int64_t rows = gridDim.x;
        if (pos) pos += (int64_t)grp * V * hgt * wid * heads * dh;
        offs_logits += (int64_t)grp * B * J * heads * NPTS * 3;
        g += grp * rows * heads * cf;
        if (e) e += grp * rows * heads * dh;
        sigma += grp * rows * heads;
        if (grp) rowmask = nullptr;  // identical for every group: written once
    }
    const int v = row % V;
    const int bj = row / V;
    const int j = bj % J, b = bj / J;
    const int lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    const bool ok = valid[((int64_t)b * V + v) * J + j] != 0;
    if (threadIdx.x == 0 && rowmask) rowmask[row] = ok ? 1 : 0;
    const int hw = hgt * wid;
    const float ax = anchors[(((int64_t)b * V + v) * J + j) * 2 + 0];
    const float ay = anchors[(((int64_t)b * V + v) * J + j) * 2 + 1];
    const float* fbase = feat + ((int64_t)v * B + b) * hw * cf;
    const int stride_ol = heads * NPTS * 3;

    for (int h = threadIdx.x >> 6; h < heads; h += nw) {
        float accf[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) accf[i] = 0.f;
        float acce = 0.f, sig = 0.f;
        if (ok) {  // wave-uniform: a masked row is overwritten with zeros after output_proj anyway
            // ---- per-point setup on lanes 0..15 (other lanes mirror lane&15, harmless)
            const int p = lane & 15;
            const float* ol = offs_logits + (int64_t)bj * stride_ol;
            float ox = ol[(h * NPTS + p) * 2 + 0], oy = ol[(h * NPTS + p) * 2 + 1];
            float lg = ol[heads * NPTS * 2 + h * NPTS + p];
            float mx = lg;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float ex = expf(lg - mx);
            float sm = ex;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
            const float aw = ex / sm;
            const float locx = ax + ox / (float)wid, locy = ay + oy / (float)hgt;
            const float w_im = locx * (float)wid - 0.5f, h_im = locy * (float)hgt - 0.5f;
            const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)hgt) && (w_im < (float)wid);
            const float hl = floorf(h_im), wl = floorf(w_im);
            const float lh = h_im - hl, lw = w_im - wl, hh = 1.f - lh, hwt = 1.f - lw;
            const int h0 = (int)hl, w0 = (int)wl, h1 = h0 + 1, w1 = w0 + 1;
            float cw[4];
            int ci[4];
            cw[0] = (inside && h0 >= 0 && w0 >= 0) ? hh * hwt : 0.f;
            cw[1] = (inside && h0 >= 0 && w1 <= wid - 1) ? hh * lw : 0.f;
            cw[2] = (inside && h1 <= hgt - 1 && w0 >= 0) ? lh * hwt : 0.f;
            cw[3] = (inside && h1 <= hgt - 1 && w1 <= wid - 1) ? lh * lw : 0.f;
            ci[0] = h0 * wid + w0; ci[1] = h0 * wid + w1; ci[2] = h1 * wid + w0; ci[3] = h1 * wid + w1;
            float mass = (cw[0] + cw[1]) + (cw[2] + cw[3]);
            float sp = aw * mass;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sp += __shfl_xor(sp, o, 64);
            sig = sp;
#pragma unroll
            for (int c = 0; c < 4; ++c) cw[c] *= aw;
            // ---- gather.  WIDE form (round 5; the path's shapes: cf a multiple of 64, dh = 64 with the positional table): the loop was
            // bound by the NUMBER of vector-memory instructions (64 + 64 per head, 8 / 4 bytes per lane), not by bytes - so a feature
            // row is read by HALF a wave (16 bytes per lane at cf = 128) and every instruction fetches TWO corners, a positional slice by
            // a QUARTER wave and every instruction fetches FOUR: 32 + 16 instructions per head.  The partial sums of the halves / quarters are added at the end (fixed order).
            const float* pbase = pos ? pos + (int64_t)v * hw * (heads * dh) + h * dh : nullptr;
            if (WIDE && (!POS || dh == 64)) {
                constexpr int VPL = 2 * CPL;                     // feature floats per lane: the half wave covers the row
                const int half = lane >> 5, l32 = lane & 31, quarter = lane >> 4, l16 = lane & 15;
                // the lane that broadcasts corner (point q, corner c) to a reader in half `hf` / quarter `qt` is lane q of the SAME half /
                // quarter (every lane holds the corners of point lane & 15), with the corner picked by its own half / quarter
                float selw[2], selq;
                int seli[2], selqi;
#pragma unroll
                for (int t = 0; t < 2; ++t) { selw[t] = half ? cw[2 * t + 1] : cw[2 * t]; seli[t] = half ? ci[2 * t + 1] : ci[2 * t]; }
                selq = quarter == 0 ? cw[0] : (quarter == 1 ? cw[1] : (quarter == 2 ? cw[2] : cw[3]));
                selqi = quarter == 0 ? ci[0] : (quarter == 1 ? ci[1] : (quarter == 2 ? ci[2] : ci[3]));
                float af[VPL];
#pragma unroll
                for (int i = 0; i < VPL; ++i) af[i] = 0.f;
                // (zero-weight corners - outside the map - read pixel 0 instead of being skipped: without a branch in the loop several row
                // reads stay in flight; with one every read waited for the read before it)
#pragma unroll
                for (int t = 0; t < 2; ++t) seli[t] = (selw[t] == 0.f) ? 0 : seli[t];
                selqi = (selq == 0.f) ? 0 : selqi;
#pragma unroll GATHER_UNROLL
                for (int st = 0; st < 2 * NPTS; ++st) {          // corner 2 st + half: point st / 2, corner 2 (st & 1) + half
                    const int src = (st >> 1) + 32 * half;
                    const float wgt = __shfl((st & 1) ? selw[1] : selw[0], src, 64);
                    const int idx = __shfl((st & 1) ? seli[1] : seli[0], src, 64);
                    {
                        const float* fr = fbase + (int64_t)idx * cf + l32 * VPL;
                        if constexpr (VPL == 4) {
                            const f32x4 t = *reinterpret_cast<const f32x4*>(fr);
#pragma unroll
                            for (int i = 0; i < 4; ++i) af[i] = fmaf(wgt, t[i], af[i]);
                        } else if constexpr (VPL == 8) {
                            const f32x4 t0 = *reinterpret_cast<const f32x4*>(fr), t1 = *reinterpret_cast<const f32x4*>(fr + 4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) { af[i] = fmaf(wgt, t0[i], af[i]); af[4 + i] = fmaf(wgt, t1[i], af[4 + i]); }
                        } else {
                            const f32x2 t = *reinterpret_cast<const f32x2*>(fr);
                            af[0] = fmaf(wgt, t[0], af[0]);
                            af[1] = fmaf(wgt, t[1], af[1]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < VPL; ++i) af[i] += __shfl_xor(af[i], 32, 64);     // (both halves end with the row's sum)
                f32x4 ae = {0.f, 0.f, 0.f, 0.f};
                if constexpr (POS) {
#pragma unroll GATHER_UNROLL
                    for (int st = 0; st < NPTS; ++st) {          // corner 4 st + quarter: point st, corner = quarter
                        const int src = st + 16 * quarter;
                        const float wgt = __shfl(selq, src, 64);
                        const int idx = __shfl(selqi, src, 64);
                        {
                            const f32x4 t = *reinterpret_cast<const f32x4*>(pbase + (int64_t)idx * (heads * dh) + l16 * 4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) ae[i] = fmaf(wgt, t[i], ae[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ae[i] += __shfl_xor(ae[i], 16, 64);
                        ae[i] += __shfl_xor(ae[i], 32, 64);
                    }
                }
                if (half == 0) {
                    float* go = g + ((int64_t)row * heads + h) * cf + l32 * VPL;
#pragma unroll
                    for (int i = 0; i < VPL; ++i) go[i] = af[i];
                }
                if (POS && quarter == 0) *reinterpret_cast<f32x4*>(e + (int64_t)row * heads * dh + h * dh + l16 * 4) = ae;
                if (lane == 0) sigma[(int64_t)h * gridDim.x + row] = sig;
                continue;
            }
            auto corner = [&](float wgt, int idx) {
                const float* fr = fbase + (int64_t)idx * cf + lane * CPL;
                if constexpr (CPL == 2) {
                    f32x2 t = *reinterpret_cast<const f32x2*>(fr);
                    accf[0] = fmaf(wgt, t[0], accf[0]);
                    accf[1] = fmaf(wgt, t[1], accf[1]);
                } else if constexpr (CPL == 4) {
                    f32x4 t = *reinterpret_cast<const f32x4*>(fr);
#pragma unroll
                    for (int i = 0; i < 4; ++i) accf[i] = fmaf(wgt, t[i], accf[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < CPL; ++i) accf[i] = fmaf(wgt, fr[i], accf[i]);
                }
            };
            if constexpr (POS) {
                // (with the positional table: zero-weight corners are skipped - the branch-free form below measured slower here,
                // 173 -> 231 us for the refiners' launch at batch 64)
                for (int q = 0; q < NPTS; ++q) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float wgt = __shfl(cw[c], q, 64);
                        int idx = __shfl(ci[c], q, 64);
                        if (wgt == 0.f) continue;  // wave-uniform (broadcast value)
                        corner(wgt, idx);
                        if (lane < dh) acce = fmaf(wgt, pbase[(int64_t)idx * (heads * dh) + lane], acce);
                    }
                }
            } else {
                // corners with zero weight (outside the map) read pixel 0 instead of being skipped: without a branch in the loop
                // the compiler keeps several of the 64 row reads in flight (with it every read waited for the one before - a chain
                // of 64 L2 latencies per head: the lifting head's launches 77 -> 52 us at batch 64, 30 -> 10 us at batch 1)
#pragma unroll
                for (int c = 0; c < 4; ++c) ci[c] = (cw[c] == 0.f) ? 0 : ci[c];
#pragma unroll 4
                for (int q = 0; q < NPTS; ++q) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) corner(__shfl(cw[c], q, 64), __shfl(ci[c], q, 64));
                }
            }
        }
        float* go = g + ((int64_t)row * heads + h) * cf + lane * CPL;
#pragma unroll
        for (int i = 0; i < CPL; ++i) go[i] = accf[i];
        if (e && lane < dh) e[(int64_t)row * heads * dh + h * dh + lane] = acce;
        if (lane == 0) sigma[(int64_t)h * gridDim.x + row] = sig;  // (heads, rows): one contiguous rowscale vector per head
    }
}

// ------------------------------------------------------------------ joint-to-joint attention core
// One wave per (b, head); q, k, v head slices staged in LDS; 4 lanes share a score row.
__global__ __launch_bounds__(64) void joint_mha_kernel(const float* qkv, float* out, int J, int heads, int d, float scale) {
    __shared__ float sq[16 * 64], sk[16 * 64], sv[16 * 64], sp[16 * 16];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x;
    const int c = heads * d;
    for (int i = lane; i < J * d; i += 64) {
        int t = i / d, dd = i % d;
        const float* r = qkv + ((int64_t)b * J + t) * 3 * c + h * d + dd;
        sq[t * d + dd] = r[0];
        sk[t * d + dd] = r[c];
        sv[t * d + dd] = r[2 * c];
    }
    __syncthreads();
    const int i = lane >> 2, gq = lane & 3;  // row i, columns gq, gq+4, gq+8, gq+12
    float s[4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int jj = gq + 4 * t;
        float dot = 0.f;
        if (i < J && jj < J)
            for (int dd = 0; dd < d; ++dd) dot = fmaf(sq[i * d + dd], sk[jj * d + dd], dot);
        s[t] = (i < J && jj < J) ? dot * scale : -INFINITY;
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
#pragma unroll
    for (int t = 0; t < 4; ++t) sp[i * 16 + gq + 4 * t] = (i < J) ? s[t] / sum : 0.f;
    __syncthreads();
    for (int idx = lane; idx < J * d; idx += 64) {
        int t = idx / d, dd = idx % d;
        float o = 0.f;
        for (int jj = 0; jj < J; ++jj) o = fmaf(sp[t * 16 + jj], sv[jj * d + dd], o);
        out[((int64_t)b * J + t) * c + h * d + dd] = o;
    }
}

// ------------------------------------------------------------------ fisheye reprojection
using egrf::CAM_REC;

// pts_out receives the anchors the later layers add their offsets to: the mutated points in syn mode, a copy in rw mode (may be pts)
__global__ __launch_bounds__(256) void fisheye_kernel(const float* pts, float* pts_out, const float* ctm, const float* cams, int B, int J,
                                                      float* anchors, uint8_t* valid, float* q4) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * J) return;
    const int b = idx / J, j = idx % J;
    float x = pts[idx * 3 + 0], y = pts[idx * 3 + 1], z = pts[idx * 3 + 2];
    egrf::fisheye_joint(x, y, z, ctm, cams, b, j, J, anchors, valid);
    if (!ctm || pts_out != pts) { pts_out[idx * 3 + 0] = x; pts_out[idx * 3 + 1] = y; pts_out[idx * 3 + 2] = z; }
    q4[idx * 4 + 0] = (float)(j + 1) / (float)J;
    q4[idx * 4 + 1] = x; q4[idx * 4 + 2] = y; q4[idx * 4 + 3] = z;
}

}  // namespace

extern "C" int egr_msda_gather_f32(const float* feat, int32_t cf, const float* pos, int32_t dh, const float* offs_logits,
                                   const float* anchors, const uint8_t* valid, int32_t b, int32_t views, int32_t joints,
                                   int32_t heads, int32_t hgt, int32_t wid, float* g, float* e, float* sigma,
                                   uint8_t* rowmask, int32_t groups, void* stream) {
    if (groups <= 0 || groups > 65535) return EGR_EINVAL;
    if (!feat || !offs_logits || !anchors || !valid || !g || !sigma || !rowmask) return EGR_ENULL;
    if ((pos != nullptr) != (e != nullptr)) return EGR_ENULL;
    if (b <= 0 || views <= 0 || joints <= 0 || heads <= 0 || heads > 16 || hgt <= 0 || wid <= 0) return EGR_EINVAL;
    if (pos && (dh <= 0 || dh > 64)) return EGR_EINVAL;
    int64_t rows = (int64_t)b * joints * views;
    if (rows >= (1LL << 31)) return EGR_EINVAL;
    dim3 grid((unsigned)rows, (unsigned)groups), block(64 * (heads < 4 ? heads : 4));
    hipStream_t s = (hipStream_t)stream;
    // the wide form needs 16-byte aligned rows / slices
    const bool wide = g_gather_wide_env && !(((uintptr_t)feat | (uintptr_t)pos | (uintptr_t)e) & 15) && (!pos || (dh == 64 && (heads * dh) % 4 == 0));
#define EGR_GATHER(CPL_)                                                                                                            \
    do {                                                                                                                            \
        if (pos && wide)                                                                                                            \
            hipLaunchKernelGGL((msda_gather_kernel<CPL_, true, true>), grid, block, 0, s, feat, pos, dh, offs_logits, anchors, valid, b, views, \
                               joints, heads, hgt, wid, g, e, sigma, rowmask);                                                      \
        else if (pos)                                                                                                               \
            hipLaunchKernelGGL((msda_gather_kernel<CPL_, true, false>), grid, block, 0, s, feat, pos, dh, offs_logits, anchors, valid, b, views, \
                               joints, heads, hgt, wid, g, e, sigma, rowmask);                                                      \
        else if (wide)                                                                                                              \
            hipLaunchKernelGGL((msda_gather_kernel<CPL_, false, true>), grid, block, 0, s, feat, pos, dh, offs_logits, anchors, valid, b, views, \
                               joints, heads, hgt, wid, g, e, sigma, rowmask);                                                      \
        else                                                                                                                        \
            hipLaunchKernelGGL((msda_gather_kernel<CPL_, false, false>), grid, block, 0, s, feat, pos, dh, offs_logits, anchors, valid, b, views, \
                               joints, heads, hgt, wid, g, e, sigma, rowmask);                                                      \
    } while (0)
    if (cf == 128) EGR_GATHER(2);
    else if (cf == 64) EGR_GATHER(1);
    else if (cf == 256) EGR_GATHER(4);
    else return EGR_EINVAL;
#undef EGR_GATHER
    return egr_launch_status();
}

extern "C" int egr_joint_mha_f32(const float* qkv, float* out, int32_t b, int32_t j, int32_t heads, int32_t d, float scale,
                                 void* stream) {
    if (!qkv || !out) return EGR_ENULL;
    if (b <= 0 || j <= 0 || j > 16 || heads <= 0 || d <= 0 || d > 64) return EGR_EINVAL;
    hipLaunchKernelGGL(joint_mha_kernel, dim3((unsigned)(b * heads)), dim3(64), 0, (hipStream_t)stream, qkv, out, j, heads, d,
                       scale);
    return egr_launch_status();
}

extern "C" int egr_fisheye_project2_f32(const float* pts, float* pts_out, const float* ctm, const float* cams, int32_t b, int32_t joints,
                                        float* anchors, uint8_t* valid, float* q4, void* stream) {
    if (!pts || !pts_out || !cams || !anchors || !valid || !q4) return EGR_ENULL;
    if (b <= 0 || joints <= 0) return EGR_EINVAL;
    hipLaunchKernelGGL(fisheye_kernel, dim3((unsigned)((b * joints + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts, pts_out,
                       ctm, cams, b, joints, anchors, valid, q4);
    return egr_launch_status();
}

extern "C" int egr_fisheye_project_f32(float* pts, const float* ctm, const float* cams, int32_t b, int32_t joints,
                                       float* anchors, uint8_t* valid, float* q4, void* stream) {
    return egr_fisheye_project2_f32(pts, pts, ctm, cams, b, joints, anchors, valid, q4, stream);
}
