// Backward of the deformable sampling in sample-then-project form (forward: msda_gather_kernel in egr_attn.hip).
//
// Forward, per (group, row = (b, joint, view), head h):
//     a_h = Wfold_h g_h + cfold_h sigma_h + e_h,      g_h = sum_p w_p S_p(f_v),  e_h = sum_p w_p S_p(posproj_v)[h],
//     sigma_h = sum_p w_p (in-bounds bilinear mass),  w = softmax_16(logits),    S_p = bilinear sample at anchor + offset_p
// The caller turns d a_h into  dg_h = Wfold_h^T da_h  (a 1x1 data-gradient conv); this kernel produces the gradients
// w.r.t. the 16 offsets and logits of the head (the mmcv backward's grad_sampling_loc / grad_attn_weight followed by
// the softmax backward), and scatters into the feature map / positional table gradients.  For an in-bounds corner c of
// point p the (projected) value seen by the reference is  Wfold f_c + cfold + posproj_c, so with
//     val_c = dg_h . f_c + da_h . posproj_c[h] + da_h . cfold_h
//     d w_p     = sum_c bw_c val_c                       (bw = bilinear weights, zero for out-of-bounds corners)
//     d w_im    = w_p sum_c (d bw_c / d w_im) val_c      (same for h_im; offsets are in pixels, so d off = d pixel)
// One workgroup per (row, group), one wave per head, lane p < 16 owns point p (softmax, bilinear setup), all 64 lanes
// share the channel dot products (CPL channels each) and reduce with shuffles: 3 reductions per point.
#include "egr_common.h"
#include "egorear_train.h"

namespace {

constexpr int NPTS = 16;

template <int CPL>
__global__ __launch_bounds__(256) void msda_gather_bwd_kernel(const float* feat, const float* pos, int dh, const float* offs_logits,
                                                              const float* anchors, const uint8_t* valid, int B, int V, int J,
                                                              int heads, int hgt, int wid, const float* dg, const float* da,
                                                              const float* cfold, float* dol, float* dfeat, float* dpos) {
    const int cf = CPL * 64;
    const int row = blockIdx.x;
    const int grp = blockIdx.y;
    const int64_t rows = gridDim.x;
    const int C = heads * dh;
    const int stride_ol = heads * NPTS * 3;
    if (pos) pos += (int64_t)grp * V * hgt * wid * C;
    if (dpos) dpos += (int64_t)grp * V * hgt * wid * C;
    offs_logits += (int64_t)grp * B * J * stride_ol;
    dg += (int64_t)grp * rows * heads * cf;
    da += (int64_t)grp * rows * C;
    cfold += (int64_t)grp * C;
    dol += (int64_t)grp * rows * stride_ol;
    const int v = row % V;
    const int bj = row / V;
    const int j = bj % J, b = bj / J;
    const int lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    const bool ok = valid[((int64_t)b * V + v) * J + j] != 0;
    const int hw = hgt * wid;
    const float ax = anchors[(((int64_t)b * V + v) * J + j) * 2 + 0];
    const float ay = anchors[(((int64_t)b * V + v) * J + j) * 2 + 1];
    const int64_t fimg = ((int64_t)v * B + b) * hw;
    float* drow = dol + (int64_t)row * stride_ol;

    for (int h = threadIdx.x >> 6; h < heads; h += nw) {
        const int p = lane & 15;
        if (!ok) {  // masked row: no gradient flows (wave-uniform)
            if (lane < NPTS) {
                drow[(h * NPTS + p) * 2 + 0] = 0.f;
                drow[(h * NPTS + p) * 2 + 1] = 0.f;
                drow[heads * NPTS * 2 + h * NPTS + p] = 0.f;
            }
            continue;
        }
        // ---- per-point setup, identical arithmetic to the forward
        const float* ol = offs_logits + (int64_t)bj * stride_ol;
        const float ox = ol[(h * NPTS + p) * 2 + 0], oy = ol[(h * NPTS + p) * 2 + 1];
        const float lg = ol[heads * NPTS * 2 + h * NPTS + p];
        float mx = lg;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float ex = expf(lg - mx);
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
        bool cin[4];
        cin[0] = inside && h0 >= 0 && w0 >= 0;
        cin[1] = inside && h0 >= 0 && w1 <= wid - 1;
        cin[2] = inside && h1 <= hgt - 1 && w0 >= 0;
        cin[3] = inside && h1 <= hgt - 1 && w1 <= wid - 1;
        float bw[4], gw[4], gh[4];   // bilinear weight and its derivatives w.r.t. w_im / h_im
        bw[0] = hh * hwt; gw[0] = -hh; gh[0] = -hwt;
        bw[1] = hh * lw;  gw[1] = hh;  gh[1] = -lw;
        bw[2] = lh * hwt; gw[2] = -lh; gh[2] = hwt;
        bw[3] = lh * lw;  gw[3] = lh;  gh[3] = lw;
        int ci[4];
        ci[0] = h0 * wid + w0; ci[1] = h0 * wid + w1; ci[2] = h1 * wid + w0; ci[3] = h1 * wid + w1;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (!cin[c]) { bw[c] = 0.f; gw[c] = 0.f; gh[c] = 0.f; ci[c] = -1; }
        // ---- this head's upstream gradients, spread over the lanes
        float dgl[CPL];
        const float* dgp = dg + ((int64_t)row * heads + h) * cf + lane * CPL;
#pragma unroll
        for (int i = 0; i < CPL; ++i) dgl[i] = dgp[i];
        const float dal = lane < dh ? da[(int64_t)row * C + h * dh + lane] : 0.f;
        const float dsig = wave_sum(lane < dh ? dal * cfold[h * dh + lane] : 0.f);
        const float* pbase = pos ? pos + (int64_t)v * hw * C + h * dh : nullptr;
        float* dpbase = dpos ? dpos + (int64_t)v * hw * C + h * dh : nullptr;
        float my_daw = 0.f, my_dw = 0.f, my_dh = 0.f;
        for (int q = 0; q < NPTS; ++q) {
            float s_aw = 0.f, s_w = 0.f, s_h = 0.f;   // per-lane partial dot products
            const float awq = __shfl(aw, q, 64);
            // the four corners' rows are requested together (a skipped corner reads row 0), then consumed: with the loads behind the
            // `continue` every corner waited for its own row
            int cidx[4];
            float frv[4][CPL], pv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                cidx[c] = __shfl(ci[c], q, 64);
                const int ic = cidx[c] < 0 ? 0 : cidx[c];
                const float* fr = feat + (fimg + ic) * cf + lane * CPL;
#pragma unroll
                for (int i = 0; i < CPL; ++i) frv[c][i] = fr[i];
                pv[c] = pbase ? pbase[(int64_t)ic * C + (lane < dh ? lane : 0)] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int idx = cidx[c];
                if (idx < 0) continue;  // wave-uniform
                const float b_c = __shfl(bw[c], q, 64), gw_c = __shfl(gw[c], q, 64), gh_c = __shfl(gh[c], q, 64);
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; ++i) part = fmaf(dgl[i], frv[c][i], part);
                if (pbase && lane < dh) part = fmaf(dal, pv[c], part);
                if (lane == 0) part += dsig;  // the bias term of an in-bounds corner
                s_aw = fmaf(b_c, part, s_aw);
                s_w = fmaf(gw_c, part, s_w);
                s_h = fmaf(gh_c, part, s_h);
                const float sc = awq * b_c;   // scatter: this corner's weight in the forward
                if (dfeat) {
                    float* df = dfeat + (fimg + idx) * cf + lane * CPL;
#pragma unroll
                    for (int i = 0; i < CPL; ++i) atomicAdd(df + i, sc * dgl[i]);
                }
                if (dpbase && lane < dh) atomicAdd(dpbase + (int64_t)idx * C + lane, sc * dal);
            }
            s_aw = wave_sum(s_aw);
            s_w = wave_sum(s_w);
            s_h = wave_sum(s_h);
            if (lane == q) { my_daw = s_aw; my_dw = awq * s_w; my_dh = awq * s_h; }
        }
        // ---- softmax backward over the 16 points (lanes 0..15 hold them; lanes >= 16 mirror lane & 15 harmlessly)
        const float daw = __shfl(my_daw, p, 64);
        float dotp = aw * daw;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dotp += __shfl_xor(dotp, o, 64);
        if (lane < NPTS) {
            drow[(h * NPTS + p) * 2 + 0] = my_dw;
            drow[(h * NPTS + p) * 2 + 1] = my_dh;
            drow[heads * NPTS * 2 + h * NPTS + p] = aw * (daw - dotp);
        }
    }
}

}  // namespace

extern "C" int egr_msda_gather_bwd_f32(const float* feat, int32_t cf, const float* pos, int32_t dh, const float* offs_logits,
                                       const float* anchors, const uint8_t* valid, int32_t b, int32_t views, int32_t joints,
                                       int32_t heads, int32_t hgt, int32_t wid, const float* dg, const float* da,
                                       const float* cfold, float* dol, float* dfeat, float* dpos, int32_t groups, void* stream) {
    if (groups <= 0 || groups > 65535) return EGR_EINVAL;
    if (!feat || !offs_logits || !anchors || !valid || !dg || !da || !cfold || !dol) return EGR_ENULL;
    if (dpos && !pos) return EGR_ENULL;
    if (b <= 0 || views <= 0 || joints <= 0 || heads <= 0 || heads > 16 || hgt <= 0 || wid <= 0 || dh <= 0 || dh > 64)
        return EGR_EINVAL;
    const int64_t rows = (int64_t)b * joints * views;
    if (rows >= (1LL << 31)) return EGR_EINVAL;
    dim3 grid((unsigned)rows, (unsigned)groups), block(64 * (heads < 4 ? heads : 4));
    hipStream_t s = (hipStream_t)stream;
    if (cf == 128)
        hipLaunchKernelGGL(msda_gather_bwd_kernel<2>, grid, block, 0, s, feat, pos, dh, offs_logits, anchors, valid, b, views,
                           joints, heads, hgt, wid, dg, da, cfold, dol, dfeat, dpos);
    else if (cf == 64)
        hipLaunchKernelGGL(msda_gather_bwd_kernel<1>, grid, block, 0, s, feat, pos, dh, offs_logits, anchors, valid, b, views,
                           joints, heads, hgt, wid, dg, da, cfold, dol, dfeat, dpos);
    else if (cf == 256)
        hipLaunchKernelGGL(msda_gather_bwd_kernel<4>, grid, block, 0, s, feat, pos, dh, offs_logits, anchors, valid, b, views,
                           joints, heads, hgt, wid, dg, da, cfold, dol, dfeat, dpos);
    else
        return EGR_EINVAL;
    return egr_launch_status();
}
