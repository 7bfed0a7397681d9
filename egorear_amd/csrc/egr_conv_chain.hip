// Two 1x1 / stride-1 convolutions back to back in ONE launch of the fp16 scheme (EGR_W_F16X2):
//     y = act2( W2 . act1( W1 . x + b1 ) + b2 [+ res] )
// for the chains of the path whose 128-channel intermediate used to make a round trip through HBM (round 4's PMC pass:
// profiles/r04_v3_pmc_traffic_per_launch.txt #28 -> #30, #24 -> #26, #45 -> #46): the FPN's lateral conv -> fuse conv
// (models/backbones/resnet.py:96-110, 127-133 in the reference; the fuse conv's up-sampled half enters as EGR_RES_UP2_BEFORE_ACT) and
// the refiners' frame_feat_refined_proj_layers (models/estimator/egoposeformer_heatmap_mvf_ex.py:553-563, 715).
//
// The streaming kernel's structure (conv_pw_x6_kernel, egr_conv.hip): operand roles swapped - A = weights (rows = output channels,
// stationary in LDS for the whole launch), B = activations (columns = pixels); lane (p = lane & 31, h = lane >> 5) of a wave owns
// pixel p of the wave's 32-pixel tile, its B operand of a k16 step is 2 x 16 bytes of the pixel's own NHWC row.  What makes the
// chain free of any transposition: the accumulator layout of the FIRST product (register quad g of fragment cf = channels
// 32 cf + 8 g + 4 h .. + 3 of the lane's pixel) IS the B-operand layout of the second (k16 step s of the second product = quads
// 2 (s & 1), 2 (s & 1) + 1 of fragment s / 2) - so bias + ReLU run on the accumulator registers, the values are split into their
// two fp16 planes where they stand and go straight back into the matrix cores.  No LDS, no barrier between the two products.
//
// Pre-scale of the intermediate: it has no abs-max record (it never exists as a tensor), and does not need one - a column of the
// second GEMM is ONE pixel, so each lane scales its own pixel by the power of two that puts the pixel's largest magnitude (over its
// 128 channels: the lane's 64 values and those of lane ^ 32) into [2^14, 2^15), and multiplies the pixel's accumulators by the inverse
// (exact).  Finer than the per-tensor scale of the single launches: every pixel keeps its own 22 bits.
//
// Registers: both accumulator sets (64 + 64) and two sets of weight fragments (the next k step's are read from LDS in front of the
// current step's MFMAs) live at the same time; the NEXT tile's activations are requested behind the second product, when the
// first product's registers are dead, and have the whole epilogue to land.
// LDS: W1 (cin 64: 32 KB, 128: 64 KB) + W2 (64 KB) + the channel vectors + one 16-row epilogue patch per wave (rows come back as whole
// 128-byte lines, as in conv_pw_x6_kernel; 16 rows instead of 32 so that the 128 -> 128 -> 128 chain fits 160 KB).
#include "egr_conv_shared.h"

using namespace egrc;

namespace {

#ifndef CHAIN_SB_G
#define CHAIN_SB_G 1
#endif
#ifndef CHAIN_SB_E
#define CHAIN_SB_E 1
#endif
#define SB_G() do { if (CHAIN_SB_G) __builtin_amdgcn_sched_barrier(0); } while (0)
#define SB_E() do { if (CHAIN_SB_E) __builtin_amdgcn_sched_barrier(0); } while (0)

struct ChainArgs {
    ConvArgs a;              // the chain as ONE 1x1 conv cin -> cout (geometry, maps, act = act2, residual, groups); a.w = W1's image, a.wds = its descale
    const void* w2;          // egr_pack_wh2_f32 image of the second conv (cout x cmid)
    const float* wds2;       // its descale, (groups,) 128, group stride a.d.gp
    const float* shift1;     // bias of the first conv, (groups,) 128, group stride gp1; NULL = 0
    int64_t gw2, gp1;        // group strides: 16-bit elements of w2; floats of shift1 / W1's descale
    int act1;
};

constexpr int OOB = (int)0x80000000;

// KS1 = cin / 16 (4 / 8); the intermediate and the output have 128 channels (four 32-row fragments each).
// RESK: 0 no residual, 1 residual of the output's shape, 2 half-resolution residual up-sampled on the fly (EGR_RES_UP2_BEFORE_ACT)
template <int KS1, int RESK>
__global__ __launch_bounds__(512) void conv_pw_chain_kernel(const ChainArgs ca) {
    constexpr int NCF = 4, KS2 = 8, NPL = 2;
    constexpr bool PIPE = true;
    constexpr int W1B = NCF * KS1 * NPL * 1024, W2B = NCF * KS2 * NPL * 1024;
    constexpr int PR = 16, PATCH = PR * 144;          // per-wave epilogue patch: 16 pixels x 32 channels, rows padded to 144 bytes
    constexpr int VEC = 4 * 128 * 4;                  // sc1, sh1, sc2, sh2
    static_assert(W1B + W2B + VEC + 8 * PATCH <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) uint8_t lds[W1B + W2B + VEC + 8 * PATCH];
    float* const s_sc1 = reinterpret_cast<float*>(lds + W1B + W2B);
    float* const s_sh1 = s_sc1 + 128;
    float* const s_sc2 = s_sh1 + 128;
    float* const s_sh2 = s_sc2 + 128;
    const ConvArgs& a = ca.a;
    const egr_conv_desc& d = a.d;
    const int grp = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    float sa, ads;
    act_prescale(a.amax_in, lane, sa, ads);
    float amx = 0.f;
    // ds_read / ds_write take a 16-bit immediate offset: one opaque base register per 64-KB window of the LDS image keeps every weight
    // fragment read at base + immediate (left to itself the compiler materialised one address register per fragment above 64 KB: 30 registers)
    unsigned lb0 = lane * 16, lb1 = lane * 16 + 65536;
    asm volatile("" : "+v"(lb0), "+v"(lb1));
    auto wfrag = [&](const int off) __attribute__((always_inline)) {      // 16 bytes of this lane at byte offset `off` (a literal) + 16 lane
        return off < 65536 ? *reinterpret_cast<const u32x4*>(lds + lb0 + off) : *reinterpret_cast<const u32x4*>(lds + lb1 + (off - 65536));
    };
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.x + grp * d.gx), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.y + grp * d.gy), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.res ? a.res + grp * d.gr : a.x), 0, 0x80000000u, 0x00020000);
    const int HoWo = d.ho * d.wo;
    const int T = (a.M + 31) >> 5, NW = gridDim.x * 8;
    int t = blockIdx.x * 8 + wave;

    auto x_off = [&](int tile) {      // byte offset of the lane's pixel in x (OOB past the last pixel: the loads return zeros)
        const int m = tile * 32 + p;
        int n, pix;
        if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
        else { n = fdiv(m, a.dHoWo); pix = m - n * HoWo; }
        const int xb = a.x_plain ? n * (int)d.xmap.stride_inner : (int)fmap(d.xmap, a.dXin, n);
        return (tile < T && m < a.M) ? (xb + pix * d.ldx) * 4 + h * 16 : OOB;
    };
    // weights -> LDS, both images, permuted to the lane's k order (lane half h holds k = {4h .. 4h+3} u {8+4h .. 8+4h+3} of a k16 step:
    // a 16-byte piece of the image (lane (p, q): k 8q .. 8q+7) goes as two 8-byte halves to the new lanes (p, 0) and (p, 1), slot q)
    u32x4 raw[2 * KS1];
    {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        // (all of a round's 16-byte pieces in flight at once: a load -> store loop pays one memory latency per piece - 16 of them in
        // front of a launch whose waves only see a handful of tiles each)
        auto stage = [&](const uint8_t* img, auto ks_tag, uint8_t* dst0) __attribute__((always_inline)) {
            constexpr int KS = decltype(ks_tag)::value;
            constexpr int NV = NCF * KS * NPL * 64 / 512;           // 16-byte pieces per thread
            u32x4 wv[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = tid + 512 * i;
                const int cf = v / (KS * 64 * NPL), rem = v - cf * (KS * 64 * NPL);
                wv[i] = *reinterpret_cast<const u32x4*>(img + ((int64_t)cf * (KS / 2) * 2 * NPL) * 1024 + rem * 16);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = tid + 512 * i;
                const int lo = v & 63, blk = v >> 6;
                uint8_t* const dst = dst0 + blk * 1024 + (lo & 31) * 16 + (lo >> 5) * 8;
                *reinterpret_cast<u32x2*>(dst) = u32x2{wv[i][0], wv[i][1]};
                *reinterpret_cast<u32x2*>(dst + 512) = u32x2{wv[i][2], wv[i][3]};
            }
        };
        const int xo = x_off(t);
#pragma unroll
        for (int i = 0; i < 2 * KS1; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo + i * 32, 0, 0);
        stage(reinterpret_cast<const uint8_t*>(a.w) + (int64_t)grp * d.gw * 2, std::integral_constant<int, KS1>{}, lds);
        stage(reinterpret_cast<const uint8_t*>(ca.w2) + (int64_t)grp * ca.gw2 * 2, std::integral_constant<int, KS2>{}, lds + W1B);
        if (tid < 128) {
            // (the exact power-of-two descales ride on the channel scales: first conv = activation pre-scale x W1's row scale; second
            // conv = W2's row scale - the per-pixel scale of the intermediate is undone in the registers)
            s_sc1[tid] = ads * a.wds[grp * ca.gp1 + tid];
            s_sh1[tid] = ca.shift1 ? ca.shift1[grp * ca.gp1 + tid] : 0.f;
            s_sc2[tid] = (tid < d.cout) ? ca.wds2[grp * d.gp + tid] * ((a.scale && tid < d.cout) ? a.scale[grp * d.gp + tid] : 1.f) : 1.f;
            s_sh2[tid] = (a.shift && tid < d.cout) ? a.shift[grp * d.gp + tid] : 0.f;
        }
    }
    __syncthreads();

    const float floor1 = (ca.act1 == EGR_ACT_RELU) ? 0.f : -__builtin_inff();
    const float floor_ = (d.act == EGR_ACT_RELU) ? 0.f : -__builtin_inff();
    const bool res_before = d.res_mode == EGR_RES_BEFORE_ACT;
    constexpr int PW[3] = {1, 0, 0}, PX[3] = {0, 1, 0};      // (l,h) (h,l) (h,h): smallest products first
    for (; t < T; t += NW) {
#ifdef CHAIN_EXP_NOLOAD      // (CHAIN_EXP_*: elimination builds for tools/chain_micro.py - timing only, the results are wrong)
        const int xo_n = OOB;
#else
        const int xo_n = x_off(t + NW);
#endif
        // ---------------------------------------------------------------- first product: acc1 = W1 . x
        // (PIPE: the weight fragments of k step s + 1 are read from LDS in front of step s's MFMAs - two register sets - so that no step
        // starts with an exposed LDS round trip; the 128 -> 128 -> 128 chain with the up-sampled residual has no registers for the second set)
        f32x16 acc1[NCF];
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[cf][r] = 0.f;
        u32x4 wf[PIPE ? 2 : 1][NCF][NPL];
        auto read_w1 = [&](const int set, const int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    wf[PIPE ? set : 0][cf][pl] = wfrag(((cf * KS1 + ks) * NPL + pl) * 1024);
        };
        auto read_w2 = [&](const int set, const int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    wf[PIPE ? set : 0][cf][pl] = wfrag(W1B + ((cf * KS2 + ks) * NPL + pl) * 1024);
        };
        if constexpr (PIPE) read_w1(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            unsigned xh[4], xl[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32x4& src = raw[2 * ks + (e >> 1)];
                split2_f16(__uint_as_float(src[2 * (e & 1)]), __uint_as_float(src[2 * (e & 1) + 1]), sa, xh[e], xl[e]);
            }
            u32x4 xb[NPL];
            xb[0] = u32x4{xh[0], xh[1], xh[2], xh[3]};
            xb[1] = u32x4{xl[0], xl[1], xl[2], xl[3]};
            if constexpr (PIPE) {
                if (ks + 1 < KS1) read_w1((ks + 1) & 1, ks + 1);
                else read_w2((ks + 1) & 1, 0);            // the second product's first step
            } else {
                read_w1(0, ks);
            }
            SB_G();
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf)
#ifdef CHAIN_EXP_NOMFMA1
                    if (t3 == 0 && ks == 0)
#endif
                    acc1[cf] = mfma_split<NPL>(wf[PIPE ? (ks & 1) : 0][cf][PW[t3]], xb[PX[t3]], acc1[cf]);
            SB_G();
        }
        // ---------------------------------------------------------------- bias + activation on the accumulators, the pixel's maximum
        float pmax = 0.f;
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc1 + cf * 32 + 8 * g + 4 * h);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(s_sh1 + cf * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc1[cf][4 * g + e] * sc[e] + sh[e];
                    v = v > floor1 ? v : floor1;
                    acc1[cf][4 * g + e] = v;
                    pmax = fmaxf(pmax, fabsf(v));
                }
            }
        pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
        // 2^k with pmax 2^k in [2^14, 2^15), k clamped to +-60 (act_prescale's rule, per pixel); NaN / Inf propagate through the products
        int k2 = 141 - (int)(__float_as_uint(pmax) >> 23);
        k2 = k2 > 60 ? 60 : (k2 < -60 ? -60 : k2);
        const float sa2 = __uint_as_float((unsigned)(127 + k2) << 23), inv2 = __uint_as_float((unsigned)(127 - k2) << 23);
        // ---------------------------------------------------------------- second product: acc2 = W2 . y1 (y1 = acc1, in place)
        SB_G();
        f32x16 acc2[NCF];
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[cf][r] = 0.f;
        constexpr int S0 = KS1 & 1;              // register set holding step 0 of the second product (read behind the first product's last step)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const int cf1 = ks >> 1, g0 = 2 * (ks & 1);
            unsigned xh[4], xl[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r0 = 4 * (g0 + (e >> 1)) + 2 * (e & 1);
                split2_f16(acc1[cf1][r0], acc1[cf1][r0 + 1], sa2, xh[e], xl[e]);
            }
            u32x4 xb[NPL];
            xb[0] = u32x4{xh[0], xh[1], xh[2], xh[3]};
            xb[1] = u32x4{xl[0], xl[1], xl[2], xl[3]};
            if constexpr (PIPE) {
                if (ks + 1 < KS2) read_w2((S0 + ks + 1) & 1, ks + 1);
            } else {
                read_w2(0, ks);
            }
            SB_G();
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf)
#ifdef CHAIN_EXP_NOMFMA2
                    if (t3 == 0 && ks == 0)
#endif
                    acc2[cf] = mfma_split<NPL>(wf[PIPE ? ((S0 + ks) & 1) : 0][cf][PW[t3]], xb[PX[t3]], acc2[cf]);
            SB_G();
        }
        // the next tile's activations: requested now that the first product's registers are dead (the whole epilogue to land)
        SB_G();
#pragma unroll
        for (int i = 0; i < 2 * KS1; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo_n + i * 32, 0, 0);
        SB_G();
        // ---------------------------------------------------------------- epilogue (conv_pw_x6_kernel's, on 16-row patches)
        const int m = t * 32 + p;
        int n, pix;
        if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
        else { n = fdiv(m, a.dHoWo); pix = m - n * HoWo; }
        const bool live = m < a.M;
        const int yo_p = live ? ((a.y_plain ? n * (int)d.ymap.stride_inner : (int)fmap(d.ymap, a.dYin, n)) + pix * d.ldy) * 4 : OOB;
        int ro_p = OOB, ox_p = 0, oy_p = 0;
        float lx1_p = 0.f, ly1_p = 0.f;
        if constexpr (RESK == 1) {
            if (live) ro_p = ((a.r_plain ? n * (int)d.rmap.stride_inner : (int)fmap(d.rmap, a.dRin, n)) + pix * d.ldr) * 4;
        }
        if constexpr (RESK == 2) {
            // bilinear x2 (align_corners = True, ATen arithmetic as in upsample2x_kernel) of a half-resolution tensor
            const int ho = (a.wo_shift >= 0) ? (pix >> a.wo_shift) : fdiv(pix, a.dWo), wo = pix - ho * d.wo;
            const int hl = d.ho >> 1, wl = d.wo >> 1;
            const float shh = (d.ho > 1) ? (float)(hl - 1) / (float)(d.ho - 1) : 0.f;
            const float sww = (d.wo > 1) ? (float)(wl - 1) / (float)(d.wo - 1) : 0.f;
            const float fy = shh * (float)ho, fx = sww * (float)wo;
            const int y0 = (int)fy, x0 = (int)fx;
            ly1_p = fminf(fmaxf(__builtin_fmaf(shh, (float)ho, -(float)y0), 0.f), 1.f);
            lx1_p = fminf(fmaxf(__builtin_fmaf(sww, (float)wo, -(float)x0), 0.f), 1.f);
            if (live) ro_p = ((a.r_plain ? n * (int)d.rmap.stride_inner : (int)fmap(d.rmap, a.dRin, n)) + (y0 * wl + x0) * d.ldr) * 4;
            ox_p = (x0 + 1 > wl - 1) ? 0 : d.ldr * 4;
            oy_p = (y0 + 1 > hl - 1) ? 0 : wl * d.ldr * 4;
        }
        const int qd = lane & 7, rsub = lane >> 3;
        uint8_t* const patch = lds + W1B + W2B + VEC + wave * PATCH;
        // Eight steps (pixel half hp = it / 4, fragment cf = it % 4).  The residual quads of step it + 1 are requested BEFORE step it's
        // stores: vector-memory operations retire in order, so a residual load issued behind the previous step's stores would wait for
        // those stores to be acknowledged as well (measured on the single launches: one exposed round trip per step, 15 us per tile).
        int yo[2][2], ro[2][2], ox[2][2], oy[2][2];       // [hp][row i]: the two rows this lane stores in each half
        float lx1[2][2], ly1[2][2];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int src = hp * 16 + i * 8 + rsub;       // parameters come from the lane that owns that pixel
                yo[hp][i] = __shfl(yo_p, src) + qd * 16;
                if constexpr (RESK != 0) ro[hp][i] = __shfl(ro_p, src) + qd * 16;
                if constexpr (RESK == 2) {
                    ox[hp][i] = __shfl(ox_p, src); oy[hp][i] = __shfl(oy_p, src);
                    lx1[hp][i] = __shfl(lx1_p, src); ly1[hp][i] = __shfl(ly1_p, src);
                }
            }
        constexpr int NR = RESK == 2 ? 4 : (RESK == 1 ? 1 : 0);
        f32x4 rq[2][NR > 0 ? NR : 1];                        // [parity of the step][corner]
        // sixteen steps: (pixel half hp, fragment cf, row i of the two rows a lane stores per half)
        auto res_issue = [&](const int st, const int par) __attribute__((always_inline)) {
            if constexpr (RESK != 0) {
                const int hp = st >> 3, cf = (st >> 1) & 3, i = st & 1;
#ifdef CHAIN_EXP_NORES
                if (st > 0) return;
#endif
                // RESK == 1: the residual has the OUTPUT's shape - a fragment past cout (cout < 128) lies outside its row (and, at the
                // last pixel, outside the tensor): not requested, like the stores of those channels
                const int r0 = (RESK == 1 && cf * 32 + 4 * qd >= d.cout) ? OOB : ro[hp][i] + cf * 128;
                rq[par][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, r0, 0, 0));
                if constexpr (RESK == 2) {
                    rq[par][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro[hp][i] + ox[hp][i] + cf * 128, 0, 0));
                    rq[par][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro[hp][i] + oy[hp][i] + cf * 128, 0, 0));
                    rq[par][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro[hp][i] + oy[hp][i] + ox[hp][i] + cf * 128, 0, 0));
                }
            }
        };
        auto patch_write = [&](const int u) __attribute__((always_inline)) {
            const int hp = u >> 2, cf = u & 3;
            if ((p >> 4) == hp) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(patch + (p & 15) * 144 + g * 32 + h * 16) =
                        f32x4{acc2[cf][4 * g] * inv2, acc2[cf][4 * g + 1] * inv2, acc2[cf][4 * g + 2] * inv2, acc2[cf][4 * g + 3] * inv2};
            }
        };
        res_issue(0, 0);
        patch_write(0);
        // eight units (pixel half hp, fragment cf) of two rows each: both rows of a unit are read back, then the NEXT unit's accumulators go
        // into the patch in front of this unit's arithmetic and stores - the write's LDS latency hides behind them
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#ifdef CHAIN_EXP_NOEPI
            if (u > 0) break;
#endif
            const int hp = u >> 2, cf = u & 3;
            __builtin_amdgcn_wave_barrier();
            f32x4 vv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) vv[i] = *reinterpret_cast<const f32x4*>(patch + (i * 8 + rsub) * 144 + qd * 16);
            const int c = cf * 32 + 4 * qd;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc2 + c), sh = *reinterpret_cast<const f32x4*>(s_sh2 + c);
            const bool cok = c < d.cout;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (u + 1 < 8) patch_write(u + 1);
            SB_E();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int st = 2 * u + i, par = st & 1;
                if (st + 1 < 16) res_issue(st + 1, par ^ 1);
                SB_E();
                f32x4 v = vv[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float tt = v[e] * sc[e] + sh[e];
                    if constexpr (RESK == 1) tt += res_before ? rq[par][0][e] : 0.f;
                    if constexpr (RESK == 2) {
                        const float lx0 = 1.f - lx1[hp][i], ly0 = 1.f - ly1[hp][i];
                        tt += ly0 * (lx0 * rq[par][0][e] + lx1[hp][i] * rq[par][1][e]) + ly1[hp][i] * (lx0 * rq[par][2][e] + lx1[hp][i] * rq[par][3][e]);
                    }
                    tt = tt > floor_ ? tt : floor_;
                    if constexpr (RESK == 1) tt += res_before ? 0.f : rq[par][0][e];
                    v[e] = tt;
                }
                if (cok && yo[hp][i] >= 0) amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
#ifdef CHAIN_EXP_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, (v[0] == 12345.678f) ? yo[hp][i] + cf * 128 : OOB, 0, 0);
#else
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, cok ? yo[hp][i] + cf * 128 : OOB, 0, 0);
#endif
                SB_E();
            }
        }
    }
    if (a.amax_out) amax_flush(a.amax_out, amx, (int)blockIdx.x * 8 + wave);
}

// ------------------------------------------------------------------ 256 -> 256 -> <= 128 (round 5): both weight matrices STREAMED
// The heat-map heads' Conv2d(256, 256, 1) + ReLU -> [Upsample ->] Conv2d(256, 128, 1) (egoposeformer_heatmap_mvf_ex.py:101-126, 570-584):
// W1 alone is 256 KB as two fp16 planes - nothing stays in LDS.  Same lane-owns-a-pixel structure and register hand-over between the
// products as conv_pw_chain_kernel, but
//   * the eight waves of a workgroup walk their 32-pixel tiles in LOCKSTEP through one sequence of 48 weight chunks per tile round
//     (8 KB each: four 32-row fragments x two planes of one k16 step), every thread fetches 16 bytes of a chunk four steps ahead
//     (L2-resident: 384 KB per group), parks it in a four-slot LDS ring one step ahead, one barrier per step;
//   * the 256-channel intermediate is produced and consumed in two halves of 128 channels (mh): first product W1[128 mh ..] . x over
//     all 256 input channels (the pixel's row is streamed from L1 / L2 a second time for mh = 1), then the second product's K range
//     128 mh .. 128 mh + 127 on the registers - 64 + 64 accumulators as in the resident kernel;
//   * each half scales its own pixel maximum into [2^14, 2^15); between the halves the second product's accumulators move to the new
//     scale by an exact power of two (clamped to 2^+-40 of the first: the other half is then below 2^-40 of the sum).
// NWV waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU, each with its own ring and barriers: while one waits for its
// pixel rows the other multiplies - eight waves in lockstep stall together)
template <int RESK, int NWV>
__global__ __launch_bounds__(64 * NWV, 8 / NWV) void conv_pw2_kernel(const ChainArgs ca) {
    static_assert(RESK == 0 && (NWV == 4 || NWV == 8), "the heads' chains carry no residual");
    constexpr int NTHR = 64 * NWV, PPT = 512 / NTHR;          // 16-byte pieces of a chunk per thread
    constexpr int NCF = 4, NPL = 2, KS1 = 16, KS2H = 8, CH = NCF * NPL * 1024;      // a chunk: 8 KB
    constexpr int NCH = 2 * (KS1 + KS2H);                                           // 48 chunks per tile round
    constexpr int R = 4, D = NWV == 8 ? 4 : 3;                                                     // ring slots; global requests this many steps ahead
    constexpr int PR = 16, PATCH = PR * 144;
    constexpr int VEC = (256 + 256 + 128 + 128) * 4;
    __shared__ __attribute__((aligned(16))) uint8_t lds[R * CH + VEC + NWV * PATCH];
    float* const s_sc1 = reinterpret_cast<float*>(lds + R * CH);
    float* const s_sh1 = s_sc1 + 256;
    float* const s_sc2 = s_sh1 + 256;
    float* const s_sh2 = s_sc2 + 128;
    const ConvArgs& a = ca.a;
    const egr_conv_desc& d = a.d;
    const int grp = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    float sa, ads;
    act_prescale(a.amax_in, lane, sa, ads);
    float amx = 0.f;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.x + grp * d.gx), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.y + grp * d.gy), 0, 0x80000000u, 0x00020000);
    const int HoWo = d.ho * d.wo;
    const int T = (a.M + 31) >> 5, NW = gridDim.x * NWV;
    const int rounds = (T + NW - 1) / NW;                 // workgroup-uniform: every wave passes every barrier

    auto x_off = [&](int tile) {      // byte offset of the lane's pixel in x (OOB past the last pixel: the loads return zeros)
        const int m = tile * 32 + p;
        int n, pix;
        if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
        else { n = fdiv(m, a.dHoWo); pix = m - n * HoWo; }
        const int xb = a.x_plain ? n * (int)d.xmap.stride_inner : (int)fmap(d.xmap, a.dXin, n);
        // (the first product takes its operands in the image's own k order - lane half h: k 8h .. 8h+7 of a k16 step - so a lane reads 32
        // CONTIGUOUS bytes of its pixel's row per step and the two lanes of a pixel share a 64-byte line: half the address work of the
        // {4h..4h+3} u {8+4h..} order that the register hand-over imposes on the second product only)
        return (tile < T && m < a.M) ? (xb + pix * d.ldx) * 4 + h * 32 : OOB;
    };
    // this thread's 16 bytes of chunk c (c: a literal): W1 image [cf 8][ks 16][pl][lane][16 B], W2 image [cf 4][ks 16][pl][lane][16 B]
    // (through buffer descriptors: ONE per-thread offset register - fragment cfl, plane pl_, lane - and the chunk's position as the
    // instruction's scalar offset; as 48 flat pointers the addresses took ~100 registers and the kernel spilled)
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(reinterpret_cast<const uint8_t*>(a.w) + (int64_t)grp * d.gw * 2), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(reinterpret_cast<const uint8_t*>(ca.w2) + (int64_t)grp * ca.gw2 * 2), 0, 0x80000000u, 0x00020000);
    struct Piece { u32x4 v[PPT]; };
    const int cfl = tid >> 7, pl_ = (tid >> 6) & 1;          // piece j of this thread: fragment cfl + (4 / PPT) j
    const int woff = (cfl * 16 * NPL + pl_) * 1024 + lane * 16;
    auto chunk_load = [&](const int c) __attribute__((always_inline)) {
        const int mh = c / (KS1 + KS2H), r = c % (KS1 + KS2H);
        Piece q;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int fo = j * (NCF / PPT) * 16 * NPL * 1024;          // the fragment step between a thread's pieces
            q.v[j] = r < KS1 ? __builtin_amdgcn_raw_buffer_load_b128(rw1, woff, (mh * NCF * 16 + r) * NPL * 1024 + fo, 0)
                             : __builtin_amdgcn_raw_buffer_load_b128(rw2, woff, (mh * KS2H + (r - KS1)) * NPL * 1024 + fo, 0);
        }
        return q;
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // a 16-byte piece of the image (lane (p, q): k 8q .. 8q+7) goes as two 8-byte halves to the lanes (p, 0) and (p, 1), slot q (the
    // resident kernel's permutation: a lane's k order is {4h .. 4h+3} u {8+4h .. 8+4h+3})
    uint8_t* const park_dst = lds + (cfl * NPL + pl_) * 1024 + (lane & 31) * 16 + (lane >> 5) * 8;
    uint8_t* const park_nat = lds + (cfl * NPL + pl_) * 1024 + lane * 16;
    // (chunk c, a literal, is parked in slot c % R: first-product chunks as they come, second-product chunks permuted)
    auto park = [&](const int c, const Piece& q) __attribute__((always_inline)) {
        const int slot = c % R;
        const bool nat = (c % (KS1 + KS2H)) < KS1;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int fo = j * (NCF / PPT) * NPL * 1024;
            if (nat) {
                *reinterpret_cast<u32x4*>(park_nat + slot * CH + fo) = q.v[j];
            } else {
                *reinterpret_cast<u32x2*>(park_dst + slot * CH + fo) = u32x2{q.v[j][0], q.v[j][1]};
                *reinterpret_cast<u32x2*>(park_dst + slot * CH + fo + 512) = u32x2{q.v[j][2], q.v[j][3]};
            }
        }
    };
    for (int i = tid; i < 256; i += NTHR) {
        s_sc1[i] = ads * a.wds[grp * ca.gp1 + i];
        s_sh1[i] = ca.shift1 ? ca.shift1[grp * ca.gp1 + i] : 0.f;
    }
    if (tid < 128) {
        s_sc2[tid] = (tid < d.cout) ? ca.wds2[grp * d.gp + tid] * ((a.scale && tid < d.cout) ? a.scale[grp * d.gp + tid] : 1.f) : 1.f;
        s_sh2[tid] = (a.shift && tid < d.cout) ? a.shift[grp * d.gp + tid] : 0.f;
    }
    // Pipeline of a step c: request chunk c + 4, park chunk c + 2 (visible from step c + 1 on: one barrier per step), multiply fragments
    // 0-1 of chunk c while fragments 2-3 are read from LDS, then multiply 2-3 while fragments 0-1 of chunk c + 1 are read - the LDS
    // reads of the eight waves (64 KB per step: 512 cycles of the LDS pipe) run under the matrix instructions instead of in front of them
    Piece wst[D];
#pragma unroll
    for (int c = 0; c < D; ++c) wst[c] = chunk_load(c);
    park(0, wst[0]);
    park(1, wst[1]);
    wst[0] = chunk_load(D);                                // (step c requests chunk c + 2 + D into the register set it has just parked)
    wst[1] = chunk_load(D + 1);
    __syncthreads();
    u32x4 wfa[2][NPL], wfb[2][NPL];                       // fragments 0-1 / 2-3 of the current chunk
    auto read_half = [&](const int c, const int hf, u32x4 (&w)[2][NPL]) __attribute__((always_inline)) {
#pragma unroll
        for (int cf = 0; cf < 2; ++cf)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                w[cf][pl] = *reinterpret_cast<const u32x4*>(lds + (c % R) * CH + ((2 * hf + cf) * NPL + pl) * 1024 + lane * 16);
    };
    read_half(0, 0, wfa);

    const float floor1 = (ca.act1 == EGR_ACT_RELU) ? 0.f : -__builtin_inff();
    const float floor_ = (d.act == EGR_ACT_RELU) ? 0.f : -__builtin_inff();
    constexpr int PW[3] = {1, 0, 0}, PX[3] = {0, 1, 0};      // (l,h) (h,l) (h,h): smallest products first
    constexpr int PF = 2;                                     // k16 steps of the pixel's row in flight (3: no faster, and the kernel spills)
    u32x4 raw[PF][2];
    {
        const int xo0 = x_off((int)blockIdx.x * NWV + wave);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            raw[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo0 + i * 64, 0, 0);
            raw[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo0 + i * 64 + 16, 0, 0);
        }
    }
    for (int rd = 0; rd < rounds; ++rd) {
        const int t = (rd * (int)gridDim.x + (int)blockIdx.x) * NWV + wave;
        const int xo = x_off(t);
        f32x16 acc2[NCF];
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[cf][r] = 0.f;
        float inv2 = 1.f;
        int k2_prev = 0;
        // the memory side of step c (see above)
        auto step_io = [&](const int c) __attribute__((always_inline)) {
#ifndef PW2_EXP_NOW          // (PW2_EXP_*: elimination builds for tools/probes/chain_big_micro.py - timing only, the results are wrong)
            park((c + 2) % NCH, wst[(c + 2) % D]);            // (requested D steps ago)
            wst[(c + 2) % D] = chunk_load((c + 2 + D) % NCH);
#endif
            read_half(c, 1, wfb);
        };
        auto step_mm = [&](const int c, f32x16 (&acc)[NCF], const u32x4 (&xb)[NPL]) __attribute__((always_inline)) {
            SB_G();
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                for (int cf = 0; cf < 2; ++cf)
#ifdef PW2_EXP_NOMFMA
                    if (t3 == 0 && c % 8 == 0)
#endif
                    acc[cf] = mfma_split<NPL>(wfa[cf][PW[t3]], xb[PX[t3]], acc[cf]);
            SB_G();
            read_half(c + 1, 0, wfa);                     // (chunk c + 1: parked during step c - 1)
            SB_G();
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                for (int cf = 0; cf < 2; ++cf)
#ifdef PW2_EXP_NOMFMA
                    if (t3 == 0 && c % 8 == 0)
#endif
                    acc[2 + cf] = mfma_split<NPL>(wfb[cf][PW[t3]], xb[PX[t3]], acc[2 + cf]);
            SB_G();
#ifndef PW2_EXP_NOBAR
            // (LDS traffic only: __syncthreads() also waits for every outstanding global load - the chunk requested four steps ahead and the
            // pixel rows would be waited for at every step)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        };
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            // ------------------------------------------------------------ first product, this half of the intermediate: acc1 = W1[128 mh ..] . x
            f32x16 acc1[NCF];
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[cf][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int c = mh * (KS1 + KS2H) + ks;
                step_io(c);
                unsigned xh[4], xl[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32x4& src = raw[ks % PF][e >> 1];
                    split2_f16(__uint_as_float(src[2 * (e & 1)]), __uint_as_float(src[2 * (e & 1) + 1]), sa, xh[e], xl[e]);
                }
                u32x4 xb[NPL];
                xb[0] = u32x4{xh[0], xh[1], xh[2], xh[3]};
                xb[1] = u32x4{xl[0], xl[1], xl[2], xl[3]};
                // the row's step ks + PF (the same pixel's next pass when this one is done) takes the registers just consumed
                {
                    const int kn = ks + PF;
#ifdef PW2_EXP_NOX
                    if (false) {
#else
                    if (kn < KS1) {
#endif
                        raw[ks % PF][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo + kn * 64, 0, 0);
                        raw[ks % PF][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo + kn * 64 + 16, 0, 0);
                    }
                }
                step_mm(c, acc1, xb);
            }
            // the row's first PF steps again: the second half's pass (mh = 0) or the next round's tile (mh = 1) - in flight under the second product
            {
                const int xn = mh == 0 ? xo : x_off(((rd + 1) * (int)gridDim.x + (int)blockIdx.x) * NWV + wave);
                if (mh == 0 || rd + 1 < rounds) {
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        raw[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, xn + i * 64, 0, 0);
                        raw[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, xn + i * 64 + 16, 0, 0);
                    }
                }
            }
            // ------------------------------------------------------------ bias + activation on the accumulators, the pixel's maximum over this half
            float pmax = 0.f;
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc1 + mh * 128 + cf * 32 + 8 * g + 4 * h);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(s_sh1 + mh * 128 + cf * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc1[cf][4 * g + e] * sc[e] + sh[e];
                        v = v > floor1 ? v : floor1;
                        acc1[cf][4 * g + e] = v;
                        pmax = fmaxf(pmax, fabsf(v));
                    }
                }
            pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
            int k2 = 141 - (int)(__float_as_uint(pmax) >> 23);
            k2 = k2 > 60 ? 60 : (k2 < -60 ? -60 : k2);
            if (mh == 1) {
                // the accumulators of the first half move to this half's scale by an exact power of two.  Only the UPPER side is
                // clamped (a second half below 2^-40 of the first keeps k2_prev + 40: acc2 * 2^40 stays far inside fp32).  A second
                // half that is LARGER keeps its natural scale - forcing it up towards the first half's (an all-zero first half has
                // k2_prev = 60) would push y1 * 2^k2 past 65504 and split2_f16 would return +inf / -inf planes; mv then goes down to
                // 2^-120 (biased exponent 7: a valid float) and a negligible first half underflows harmlessly.
                k2 = k2 > k2_prev + 40 ? k2_prev + 40 : k2;
                const float mv = __uint_as_float((unsigned)(127 + k2 - k2_prev) << 23);
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[cf][r] *= mv;
            }
            k2_prev = k2;
            const float sa2 = __uint_as_float((unsigned)(127 + k2) << 23);
            inv2 = __uint_as_float((unsigned)(127 - k2) << 23);
            // ------------------------------------------------------------ second product over this half's 128 channels: acc2 += W2[:, 128 mh ..] . y1
#pragma unroll
            for (int ks = 0; ks < KS2H; ++ks) {
                const int c = mh * (KS1 + KS2H) + KS1 + ks;
                step_io(c);
                const int cf1 = ks >> 1, g0 = 2 * (ks & 1);
                unsigned xh[4], xl[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r0 = 4 * (g0 + (e >> 1)) + 2 * (e & 1);
                    split2_f16(acc1[cf1][r0], acc1[cf1][r0 + 1], sa2, xh[e], xl[e]);
                }
                u32x4 xb[NPL];
                xb[0] = u32x4{xh[0], xh[1], xh[2], xh[3]};
                xb[1] = u32x4{xl[0], xl[1], xl[2], xl[3]};
                step_mm(c, acc2, xb);
            }
        }
        // ---------------------------------------------------------------- epilogue (conv_pw_chain_kernel's, without a residual)
        const int m = t * 32 + p;
        int n, pix;
        if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
        else { n = fdiv(m, a.dHoWo); pix = m - n * HoWo; }
        const bool live = t < T && m < a.M;
        const int yo_p = live ? ((a.y_plain ? n * (int)d.ymap.stride_inner : (int)fmap(d.ymap, a.dYin, n)) + pix * d.ldy) * 4 : OOB;
        const int qd = lane & 7, rsub = lane >> 3;
        uint8_t* const patch = lds + R * CH + VEC + wave * PATCH;
        int yo[2][2];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int i = 0; i < 2; ++i) yo[hp][i] = __shfl(yo_p, hp * 16 + i * 8 + rsub) + qd * 16;
        auto patch_write = [&](const int u) __attribute__((always_inline)) {
            const int hp = u >> 2, cf = u & 3;
            if ((p >> 4) == hp) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(patch + (p & 15) * 144 + g * 32 + h * 16) =
                        f32x4{acc2[cf][4 * g] * inv2, acc2[cf][4 * g + 1] * inv2, acc2[cf][4 * g + 2] * inv2, acc2[cf][4 * g + 3] * inv2};
            }
        };
        patch_write(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int hp = u >> 2, cf = u & 3;
            __builtin_amdgcn_wave_barrier();
            f32x4 vv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) vv[i] = *reinterpret_cast<const f32x4*>(patch + (i * 8 + rsub) * 144 + qd * 16);
            const int c = cf * 32 + 4 * qd;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc2 + c), sh = *reinterpret_cast<const f32x4*>(s_sh2 + c);
            const bool cok = c < d.cout;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (u + 1 < 8) patch_write(u + 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 v = vv[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float tt = v[e] * sc[e] + sh[e];
                    tt = tt > floor_ ? tt : floor_;
                    v[e] = tt;
                }
                if (cok && yo[hp][i] >= 0) amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, cok ? yo[hp][i] + cf * 128 : OOB, 0, 0);
            }
        }
    }
    if (a.amax_out) amax_flush(a.amax_out, amx, (int)blockIdx.x * NWV + wave);
}

int g_chain_blocks = getenv("EGR_CONV_CHAIN_BLOCKS") ? atoi(getenv("EGR_CONV_CHAIN_BLOCKS")) : 256;    // resident workgroups (one per CU)

}  // namespace

extern "C" int egr_conv1x1_chain_f32(const egr_conv_desc* dd, const float* x, const void* w1, const float* shift2, const float* res, float* y,
                                     const egr_conv_aux* aux, const egr_chain_aux* chain, void* stream) {
    if (!dd || !x || !w1 || !y || !aux || !chain || !aux->w_descale || !aux->amax_in || !chain->w2 || !chain->w2_descale) return EGR_ENULL;
    ChainArgs ca;
    ConvArgs& a = ca.a;
    a = ConvArgs{};
    a.d = *dd;
    egr_conv_desc& d = a.d;
    if (d.groups <= 0) d.groups = 1;
    if (d.kh != 1 || d.kw != 1 || d.stride != 1 || d.pad != 0 || d.h != d.ho || d.w != d.wo || d.transposed || d.out_nchw || d.split_k > 1) return EGR_EINVAL;
    const bool big = d.cin == 256 && chain->cmid == 256;      // both matrices streamed (conv_pw2_kernel): no residual
    if (!big && ((d.cin != 64 && d.cin != 128) || chain->cmid != 128)) return EGR_EINVAL;
    if (d.cout <= 0 || d.cout > 128 || d.cout % 4 != 0) return EGR_EINVAL;
    if (big && d.res_mode != EGR_RES_NONE) return EGR_EINVAL;
    if (d.w_format != EGR_W_F16X2 || d.n <= 0 || d.ho <= 0 || d.wo <= 0 || d.groups > 65535) return EGR_EINVAL;
    if ((d.act != EGR_ACT_NONE && d.act != EGR_ACT_RELU) || (chain->act1 != EGR_ACT_NONE && chain->act1 != EGR_ACT_RELU)) return EGR_EINVAL;
    if (d.res_mode != EGR_RES_NONE && !res) return EGR_ENULL;
    if (d.res_mode == EGR_RES_UP2_BEFORE_ACT && ((d.ho | d.wo) & 1)) return EGR_EINVAL;
    if (d.ldx % 4 != 0 || d.ldy % 4 != 0 || (d.res_mode && d.ldr % 4 != 0)) return EGR_EINVAL;
    if (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)chain->w2 | (uintptr_t)y | (uintptr_t)res) & 15) return EGR_EINVAL;
    if (((uintptr_t)aux->amax_in | (uintptr_t)aux->amax_out) & 3) return EGR_EINVAL;
    if (d.xmap.n_inner <= 0 || d.ymap.n_inner <= 0 || (d.res_mode && d.rmap.n_inner <= 0)) return EGR_EINVAL;
    if (((d.xmap.stride_inner | d.xmap.stride_outer | d.ymap.stride_inner | d.ymap.stride_outer) % 4) != 0) return EGR_EINVAL;
    if (d.res_mode && ((d.rmap.stride_inner | d.rmap.stride_outer) % 4) != 0) return EGR_EINVAL;
    if (d.groups > 1 && (((d.gx | d.gp | d.gy | d.gr | chain->gp1) % 4) != 0 || (d.gw | chain->gw2) % 8 != 0)) return EGR_EINVAL;
    const int64_t M64 = (int64_t)d.n * d.ho * d.wo;
    if (M64 >= (1LL << 31)) return EGR_EINVAL;
    auto span = [](const egr_nmap& m, int n) {
        const int o = (n - 1) / m.n_inner, i = (n - 1 < m.n_inner ? n - 1 : m.n_inner - 1);
        return (int64_t)i * m.stride_inner + (int64_t)o * m.stride_outer;
    };
    // 32-bit BYTE offsets inside a group's window
    if ((span(d.xmap, d.n) + (int64_t)d.h * d.w * d.ldx) * 4 + 64 >= (1LL << 31)) return EGR_EINVAL;
    if ((span(d.ymap, d.n) + (int64_t)d.ho * d.wo * d.ldy) * 4 >= (1LL << 31)) return EGR_EINVAL;
    if (d.res_mode && (span(d.rmap, d.n) + (int64_t)d.ho * d.wo * d.ldr) * 4 >= (1LL << 31)) return EGR_EINVAL;
    a.x = x; a.w = static_cast<const float*>(w1); a.shift = shift2; a.res = res; a.y = y;
    a.wds = aux->w_descale; a.amax_in = aux->amax_in; a.amax_out = aux->amax_out;
    a.M = (int)M64;
    a.Npad = 128;
    a.K = d.cin;
    auto log2_exact = [](int v) { int l = 0; while ((1 << l) < v) ++l; return ((1 << l) == v) ? l : -1; };
    a.howo_shift = log2_exact(d.ho * d.wo);
    a.wo_shift = log2_exact(d.wo);
    a.x_plain = d.xmap.n_inner >= d.n;
    a.y_plain = d.ymap.n_inner >= d.n;
    a.r_plain = d.res_mode ? (d.rmap.n_inner >= d.n) : 1;
    a.dHoWo = make_fastdiv(d.ho * d.wo);
    a.dWo = make_fastdiv(d.wo);
    a.dXin = make_fastdiv(d.xmap.n_inner);
    a.dYin = make_fastdiv(d.ymap.n_inner);
    a.dRin = make_fastdiv(d.res_mode ? d.rmap.n_inner : 1);
    a.cblocks = d.cin / BK;
    a.taps = 1;
    a.ktiles = a.cblocks;
    ca.w2 = chain->w2; ca.wds2 = chain->w2_descale; ca.shift1 = chain->shift1; ca.gw2 = chain->gw2; ca.gp1 = chain->gp1; ca.act1 = chain->act1;
    int nblk = g_chain_blocks / d.groups;
    if (nblk < 1) nblk = 1;
    const int t32 = (a.M + 31) / 32;
    if (nblk * 8 > t32) nblk = (t32 + 7) / 8;
    const dim3 grid((unsigned)nblk, 1, (unsigned)d.groups);
    const int resk = d.res_mode == EGR_RES_NONE ? 0 : (d.res_mode == EGR_RES_UP2_BEFORE_ACT ? 2 : 1);
    hipStream_t s = (hipStream_t)stream;
    if (big) {
        static const int nwv = getenv("EGR_CONV_CHAIN_BIG_WAVES") ? atoi(getenv("EGR_CONV_CHAIN_BIG_WAVES")) : 8;   // (4: two workgroups per CU - measured the same)
        if (nwv == 8) {
            hipLaunchKernelGGL((conv_pw2_kernel<0, 8>), grid, dim3(512), 0, s, ca);
        } else {
            int nb4 = 2 * g_chain_blocks / d.groups;          // two four-wave workgroups per CU
            if (nb4 < 1) nb4 = 1;
            if (nb4 * 4 > t32) nb4 = (t32 + 3) / 4;
            hipLaunchKernelGGL((conv_pw2_kernel<0, 4>), dim3((unsigned)nb4, 1, (unsigned)d.groups), dim3(256), 0, s, ca);
        }
        return egr_launch_status();
    }
    if (d.cin == 64) {
        if (resk == 0) hipLaunchKernelGGL((conv_pw_chain_kernel<4, 0>), grid, dim3(512), 0, s, ca);
        else if (resk == 1) hipLaunchKernelGGL((conv_pw_chain_kernel<4, 1>), grid, dim3(512), 0, s, ca);
        else hipLaunchKernelGGL((conv_pw_chain_kernel<4, 2>), grid, dim3(512), 0, s, ca);
    } else {
        if (resk == 0) hipLaunchKernelGGL((conv_pw_chain_kernel<8, 0>), grid, dim3(512), 0, s, ca);
        else if (resk == 1) hipLaunchKernelGGL((conv_pw_chain_kernel<8, 1>), grid, dim3(512), 0, s, ca);
        else hipLaunchKernelGGL((conv_pw_chain_kernel<8, 2>), grid, dim3(512), 0, s, ca);
    }
    return egr_launch_status();
}
