// 3x3 / pad 1 convolutions of the fp16 scheme (EGR_W_F16X2) on ROLE-SPLIT, persistent workgroups whose epilogue runs
// under the next tile's K loop.  Forward launches, stride 1 and stride 2 (models/backbones/resnet.py:43-74, 121-137;
// models/estimator/egoposeformer_heatmap_mvf_ex.py:101-126, 525-532, 570-584 in the reference); of the training step (config 5)
// the trunk's conv -> train-mode BatchNorm launches (statistics in the epilogue) and the stride-1 data gradients (template
// parameter TR below).
//
// Why (profiles/r03_v2_tap_kernel_experiments.txt): vector-memory operations of a wave complete in order (one vmcnt).  In
// conv_igemm_tap_kernel every wave requests the next chunk's activations (HBM, microseconds under load) and, behind them, the next
// taps' weight fragments (L2): no weight fragment can be consumed before the older activation loads have landed, which costs that
// kernel a fifth of its time.  Round 3's role-split prototype put the two streams into different waves and its K loop ran at 88 %
// of the matrix pipe - but with every CU in lockstep its 128-KB tile epilogue was an exposed, HBM-write-bound phase.  Here:
//   * waves 0-3 MULTIPLY (one per SIMD): wave tile 128 x 64 (4 x 2 accumulators of v_mfma_f32_32x32x16_f16, three products per
//     fp32 product), weights from L2 straight into the B operand two taps ahead, A fragments from the fp16 planes in LDS.  At the
//     end of a tile they park the accumulators (descaled) in a 64-KB LDS staging area, half a tile at a time, and go on with the
//     next tile: ~3 k cycles per tile instead of the whole epilogue;
//   * waves 4-7 LOAD: the activations of chunk c + 2 are requested while chunk c is multiplied, chunk c + 1 is split into the
//     fp16 planes of the other LDS buffer - AND they run the previous tile's epilogue (BatchNorm scale / shift, residual,
//     activation, abs-max record, 16-byte row stores) in slices behind the first four chunks of the current tile, the first half
//     out of registers (so that the staging area is free for the second half after a few hundred cycles), the second out of LDS.
//     The stores of a CU are spread over four chunk periods instead of arriving as one burst from all CUs.
// One workgroup per CU walks tiles b, b + grid, ... in an XCD-contiguous order: the N tiles of one M tile run at the same time on
// neighbouring CUs of ONE XCD, so the activations they share are fetched into one L2 once.
// Stride 2: the nine taps fall into four classes by the parity of the input pixel they read (see conv_igemm_tap2_kernel); the
// four class planes of a chunk are staged together and the taps are unit shifts inside their class plane - the multiplying waves
// run the same code with another tap table.
#include "egr_conv_shared.h"

using namespace egrc;

namespace {

#ifdef TAPX_STAMPS       // diagnostic build (tools/tapx_stamps.py): per-wave s_memtime sums of the phases
#define TAPX_DBG true
#else
#define TAPX_DBG false
#endif
#define TAPX_T() (TAPX_DBG ? __builtin_amdgcn_s_memtime() : 0ull)

#ifndef TAPX_LD_AUX
#define TAPX_LD_AUX 0      // cache policy bits of the loading waves' activation / residual loads and of the output stores (2 = nt)
#endif
#ifndef TAPX_ST_AUX
#define TAPX_ST_AUX 0
#endif
#ifndef TAPX_PRIO
#define TAPX_PRIO 0        // s_setprio of the multiplying waves
#endif
constexpr int XOOB = (int)0x80000000;   // buffer offset beyond every descriptor's range: loads return 0, stores are dropped

// pixels of one 16-bit plane of a chunk
//   stride 1: whole image rows ((BM / wo + 2) x (wo + 2)) or whole small images, wo in {8, 16, 32, 64}
//   stride 2: four class planes of (BM / wo + 1) x (wo + 1) pixels (or whole small images), wo in {8, 16, 32}
constexpr int tapx_cls(int bm) { return bm == 128 ? 168 : 336; }
constexpr int tapx_hp(int bm, int stride) { return stride == 2 ? 4 * tapx_cls(bm) : (bm == 128 ? 264 : (bm == 256 ? 400 : 664)); }

// FN = 2: wave tile 128 x 64 (128 accumulator registers), the tile goes through the 64-KB staging area in two halves, the first
//         of which the loading waves take into registers at once (three barriers per tile);
// FN = 1: wave tile 128 x 32 (64 accumulator registers), the WHOLE tile fits the staging area: one barrier per tile, no register
//         copy, half the registers in both roles - and tiles of 128 x 128 / 256 x 64 for layers with few pixels.
// RES: the launch adds a residual (its quads are requested one drain step ahead: eight / four more quads of registers per loading thread).
// STRIDE = 0: the same machinery for 1x1 / stride 1 convolutions with >= 256 input channels (the heads' and refiners' 256 -> 256,
// 256 -> 128, 512 -> 128: 169 TFLOP/s / 2.7-3.7 TB/s on the tiled kernel): a chunk is 64 channels of the tile's 128 pixels, its four
// k16 steps play the part of the taps (four weight register sets, requested three steps ahead), no halo.
// TR (training launches, narrow wave tile only): 1 = the raw output feeds a train-mode BatchNorm - every tile leaves its per-channel
// sum / sum of squares (double) and extremes in a slab (egr_conv_aux.bn_partials, the layout of conv_igemm's epilogue: one slab per
// M tile); 2 = data gradient behind a ReLU: dx = (acc [+ res]) * [mask > 0], the mask quads requested like the residual's.
// d.transposed (stride 1): the data gradient of a 3x3 / pad 1 conv - the same launch with the taps' windows mirrored.
template <int WM, int WN, int FN, int STRIDE, bool RES, int TR = 0>
__global__ __launch_bounds__(512) void conv_tapx_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4 && (STRIDE == 0 || STRIDE == 1 || STRIDE == 2) && (FN == 1 || FN == 2), "four multiplying waves");
    static_assert(TR == 0 || (STRIDE != 0 && (FN == 1 || (TR == 1 && STRIDE == 1))), "training epilogues: the narrow wave tile (registers), statistics also on the wide one");
    constexpr bool BNST = TR == 1, MASK = TR == 2;
    constexpr int NPL = 2, NPR = 3, FM = 4;
    constexpr bool PW = STRIDE == 0;
    constexpr bool WHOLE = FN == 1;                           // the staging area holds the whole tile
    constexpr int BM = WM * 128, BN = WN * 32 * FN, NFB = BN / 32;
    constexpr int NT = PW ? 4 : 9;                            // "taps" of a chunk: filter taps of 16 channels, or k16 steps of 64 channels
    constexpr int SEGS = PW ? 16 : 4;                         // 4-channel units per pixel and chunk
    constexpr int HPX = PW ? BM : tapx_hp(BM, STRIDE), CLS = tapx_cls(BM);
    constexpr int PLANE = HPX * 32, HBUF = (PW ? NT : 1) * NPL * PLANE;
    constexpr int STG = (WHOLE ? BM : BM / 2) * BN * 4;       // the tile (FN = 1) or half of it (FN = 2) in fp32: 64 KB
    constexpr int NUH = (HPX * SEGS + 255) / 256;             // staging units (4 channels of one pixel) per loading thread and chunk
    constexpr int QPR = BN / 4, RPE = 256 / QPR;              // epilogue: channel quads per row, rows covered by the 256 loading threads per step
    constexpr int NQ = WHOLE ? 4 : 8;                         // quads per thread and drain step (four steps per tile)
    static_assert(STG == 65536, "staging area");
    constexpr int RED = BNST ? 4 * QPR * 96 : 0;              // statistics: eight doubles + eight floats per loading wave and channel quad
    static_assert(2 * HBUF + STG + RED <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * HBUF + STG + RED];
    // the staging area first: every one of a wave's parking stores is then one base register + an immediate offset below 64 KB
    // (behind the planes the compiler needed a dozen address registers for them - spilled, and every reload waited for vmcnt(0))
    float* const stg = reinterpret_cast<float*>(lds);
    uint8_t* const lb = lds + STG;

    const egr_conv_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int HoWo = d.ho * d.wo, wo = d.wo;
    const int WP = wo + (STRIDE == 1 ? 2 : 1);
    const int NI = HoWo >= BM ? 1 : BM / HoWo, RTI = HoWo >= BM ? BM / wo : d.ho;
    const int HPI = (RTI + (STRIDE == 1 ? 2 : 1)) * WP, HP = NI * HPI, PPI = RTI * wo;
    const int NC = PW ? a.cblocks / 2 : a.cblocks * 2;        // chunks per tile (>= 4, even)
    const int T_all = a.ntiles * d.groups;
    // tile order: XCD x owns the contiguous run [x T/8, (x + 1) T/8) and deals it to its workgroups in order (blocks b and b + 8 share
    // an XCD); the N tiles of an M tile are adjacent in a run.  Falls back to b, b + grid, ... when the counts do not divide.
    const int grid = (int)gridDim.x, bid = (int)blockIdx.x;
    const bool xmap = ((grid | T_all) & 7) == 0;
    const int t_first = xmap ? (bid & 7) * (T_all >> 3) + (bid >> 3) : bid;
    const int t_step = xmap ? (grid >> 3) : grid;
    const int t_end = xmap ? ((bid & 7) + 1) * (T_all >> 3) : T_all;
    if (t_first >= t_end) return;
    const int my_tiles = (t_end - t_first + t_step - 1) / t_step;
    const int abias = (d.w + 1) * d.ldx * 4;                  // the window starts one row + one pixel early: activation offsets >= 0

    struct Tile { int grp, tm, tn, n0, y0, xbase; };
    auto tile_of = [&](int t) __attribute__((always_inline)) {
        Tile T;
        T.grp = t / a.ntiles;
        const int rem = t - T.grp * a.ntiles;
        T.tm = (a.tilesN == 1) ? rem : fdiv(rem, a.dTilesN);
        T.tn = rem - T.tm * a.tilesN;
        const int m0 = T.tm * BM;
        int pix0;
        T.n0 = m0 >> a.howo_shift;                            // (the host sends power-of-two image sizes only: shifts, no division)
        pix0 = m0 & (HoWo - 1);
        T.y0 = PW ? 0 : pix0 >> a.wo_shift;
        T.xbase = (int)fmap(d.xmap, a.dXin, T.n0);
        return T;
    };
    const auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (wave >= 4) {
        // ------------------------------------------------------------------ loading waves
        const int lt = tid - 256;
        float sa, ads;
        act_prescale(a.amax_in, lane, sa, ads);
        int hvo[NUH];
        // chunks of activations in flight: two register sets (requested two chunk periods before they are split), or one (one period -
        // a chunk of the wide wave tile is 3.3 us) where eleven units per thread and the parked half tile do not leave room for two
        constexpr int NXS = ((NUH > 8 || BNST) && !WHOLE) ? 1 : 2;      // (the statistics' 24 registers: one set on the wide tile)
        u32x4 xr[NXS][NUH];
        __amdgpu_buffer_rsrc_t ra;
        int so_tile = 0;
        auto setup = [&](int t) __attribute__((always_inline)) {    // the load cursor enters tile t
            const Tile T = tile_of(t);
            ra = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(reinterpret_cast<const char*>(a.x + T.grp * d.gx) - (PW ? 0 : abias)), 0, 0x80000000u, 0x00020000);
            so_tile = PW ? 0 : __builtin_amdgcn_readfirstlane((T.xbase + STRIDE * T.y0 * d.w * d.ldx) * 4);
#pragma unroll
            for (int i = 0; i < NUH; ++i) {
                if constexpr (PW) {       // pixel p of the tile = output pixel tm BM + p, read at the same place; 16 units of 4 channels per pixel
                    const int u = lt + 256 * i, m = T.tm * BM + (u >> 4);
                    const int n = m >> a.howo_shift, pix = m & (HoWo - 1);
                    hvo[i] = ((int)fmap(d.xmap, a.dXin, n) + pix * d.ldx) * 4 + (u & 15) * 16;
                    continue;
                }
                // unit -> pixel of the chunk's planes: class | image of the tile | row | column
                const int seg = (lt + 256 * i) & 3, pxl = (lt + 256 * i) >> 2;
                const int q = STRIDE == 2 ? pxl / CLS : 0, hp = STRIDE == 2 ? pxl - q * CLS : pxl;
                const int il = hp / HPI, hq = hp - il * HPI, hr = hq / WP, hc = hq - hr * WP;
                // input pixel relative to (first output row's input row, column 0)
                const int dy = STRIDE == 1 ? hr - 1 : 2 * hr - (q >> 1), ix = STRIDE == 1 ? hc - 1 : 2 * hc - (q & 1);
                const bool ok = hp < HP && q < 4 && (unsigned)(STRIDE * T.y0 + dy) < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
                const int ioff = (int)fmap(d.xmap, a.dXin, T.n0 + il) - T.xbase;
                hvo[i] = ok ? (ioff + (dy * d.w + ix) * d.ldx) * 4 + seg * 16 + abias : XOOB;
            }
        };
        auto issue = [&](const int SET, int ck) __attribute__((always_inline)) {      // (SET: a literal at every call site)
            const int so = so_tile + ck * (PW ? 256 : 64);
#pragma unroll
            for (int i = 0; i < NUH; ++i) xr[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, hvo[i], so, TAPX_LD_AUX);
        };
        auto convert = [&](const int SET, int buf) __attribute__((always_inline)) {   // register set -> the fp16 planes of LDS buffer `buf`
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < NUH; ++i) {
                unsigned h0, l0, h1, l1;
                split4_f16(__uint_as_float(xr[SET][i][0]), __uint_as_float(xr[SET][i][1]), __uint_as_float(xr[SET][i][2]), __uint_as_float(xr[SET][i][3]), sa, h0, l0, h1, l1);
                if (lt + 256 * i < SEGS * HPX) {
                    // plane layout [8-channel half][pixel][8 channels]: the lanes of an A fragment (consecutive pixels, one half) read
                    // consecutive 16-byte slots - no bank conflicts (pixel-major 32-byte rows were 2-way for ds_read_b128).
                    // 1x1: one (h, l) plane pair per k16 step of the chunk
                    const int u = lt + 256 * i, px = PW ? u >> 4 : u >> 2, sg = PW ? u & 15 : u & 3;
                    uint8_t* dst = lb + buf * HBUF + (sg >> 2) * (NPL * PLANE) + ((sg >> 1) & 1) * (PLANE / 2) + px * 16 + (sg & 1) * 8;
                    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(dst + PLANE) = u32x2{l0, l1};
                }
            }
        };
        // load cursor (tile k of this workgroup, chunk); saturates at the very last chunk, which is then requested again (the same
        // vector-memory operations on every path keep the compiler's vmcnt counts exact)
        int lk = 0, lck = 0;
        auto advance = [&]() __attribute__((always_inline)) {
            if (lck + 1 < NC) ++lck;
            else if (lk + 1 < my_tiles) { lck = 0; ++lk; setup(t_first + lk * t_step); }
        };

        // ---- epilogue state: the tile whose accumulators are parked (P), this thread's channel quad and its 16 + 16 rows
        struct Pend { int grp, tm, tn, valid; };
        Pend P = {0, 0, 0, 0};
        f32x4 hold[WHOLE ? 1 : 16];                           // FN = 2: the first half of the parked tile
#pragma unroll
        for (int e = 0; e < (WHOLE ? 1 : 16); ++e) hold[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        // residual quads, requested ahead of the drain step that adds them: one step on the wide wave tile (registers), two on the narrow
        // one (layer1 at 64 channels is HBM-heavy - one chunk period of 3456 cycles did not cover the loads: the loading waves spent 60 % of
        // their time in the slices and the multiplying waves 8.8 % at the chunk barriers)
        constexpr int RD = WHOLE ? 2 : 1;
        f32x4 rr[RES ? RD : 1][RES ? NQ : 1];
#pragma unroll
        for (int q = 0; q < (RES ? RD : 1); ++q)
#pragma unroll
            for (int e = 0; e < (RES ? NQ : 1); ++e) rr[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 mk[MASK ? RD : 1][MASK ? NQ : 1];             // TR = 2: the ReLU mask's quads, like rr
#pragma unroll
        for (int q = 0; q < (MASK ? RD : 1); ++q)
#pragma unroll
            for (int e = 0; e < (MASK ? NQ : 1); ++e) mk[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        // TR = 1: this thread's share of the parked tile's statistics (its channel quad, its rows)
        double bs[BNST ? 4 : 1], bq[BNST ? 4 : 1];
        float blo[BNST ? 4 : 1], bhi[BNST ? 4 : 1];
        auto stats_clear = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < (BNST ? 4 : 1); ++c) { bs[c] = 0.0; bq[c] = 0.0; blo[c] = INFINITY; bhi[c] = -INFINITY; }
        };
        stats_clear();
        const int cq = lt % QPR, sr0 = lt / QPR;
        float amx = 0.f;
        __amdgpu_buffer_rsrc_t ry, rr_, rm_;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, dsq = {1.f, 1.f, 1.f, 1.f};
        int co = 0;
        bool live = false;
        auto pend_setup = [&](const Pend& Q) __attribute__((always_inline)) {
            ry = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.y + Q.grp * d.gy), 0, 0x80000000u, 0x00020000);
            rr_ = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.res ? a.res + Q.grp * d.gr : a.y), 0, 0x80000000u, 0x00020000);
            if constexpr (MASK) rm_ = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.mask + Q.grp * d.gy), 0, 0x80000000u, 0x00020000);
            co = Q.tn * BN + cq * 4;
            live = Q.valid && co < d.cout;
            sc = f32x4{1.f, 1.f, 1.f, 1.f};
            sh = f32x4{0.f, 0.f, 0.f, 0.f};
            // the accumulators arrive with both pre-scales on them: the inverse powers of two (activations: ads, weights: per channel)
            dsq = *reinterpret_cast<const f32x4*>(a.wds + Q.grp * d.gp + Q.tn * BN + cq * 4) * ads;
            if (live) {
                if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + Q.grp * d.gp + co);
                if (a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + Q.grp * d.gp + co);
            }
        };
        // row of quad step e (0..15) of half H inside the tile, and its output / residual byte offsets
        auto row_offsets = [&](int H, int e, int& yo, int& ro) __attribute__((always_inline)) {
            const int sr = e * RPE + sr0;                                  // row of the staged (half) tile
            const int R = WHOLE ? sr : (sr >> 6) * 128 + H * 64 + (sr & 63);
            const int m = P.tm * BM + R;
            const int n = m >> a.howo_shift, pix = m & (HoWo - 1);
            yo = live ? ((int)fmap(d.ymap, a.dYin, n) + pix * d.ldy + co) * 4 : XOOB;
            ro = (RES && live) ? ((int)fmap(d.rmap, a.dRin, n) + pix * d.ldr + co) * 4 : XOOB;
        };
        auto res_load = [&](int H, int es) __attribute__((always_inline)) {      // residual quad of step es of half H
            int yo, ro;
            row_offsets(H, es, yo, ro);
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr_, ro, 0, TAPX_LD_AUX));
        };
        auto mask_load = [&](int H, int es) __attribute__((always_inline)) {     // (the mask is laid out like y)
            int yo, ro;
            row_offsets(H, es, yo, ro);
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm_, yo, 0, TAPX_LD_AUX));
        };
        auto finish = [&](f32x4 v, const f32x4& rr, const f32x4& mq, int yo) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = (v[c] * dsq[c]) * sc[c] + sh[c];
                if (RES && d.res_mode == EGR_RES_BEFORE_ACT) t += rr[c];
                if (d.act == EGR_ACT_RELU) t = t > 0.f ? t : 0.f;
                if (RES && d.res_mode == EGR_RES_AFTER_ACT) t += rr[c];
                if (MASK) t = mq[c] > 0.f ? t : 0.f;
                v[c] = t;
                if constexpr (BNST) {
                    bs[c] += (double)t;
                    bq[c] += (double)t * (double)t;
                    blo[c] = fminf(blo[c], t);
                    bhi[c] = fmaxf(bhi[c], t);
                }
            }
            if (live) amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, yo, 0, TAPX_ST_AUX);
        };
        // drain step K of the parked tile: K = 0, 1 out of the registers (first half), K = 2, 3 out of the staging area (second half);
        // behind every quad the residual of the same slot of the NEXT step is requested (a whole chunk period to land)
        auto drain = [&](const int K) __attribute__((always_inline)) {
#ifdef TAPX_EXP_NOLOADER
            return;
#endif
#pragma unroll
            for (int e = 0; e < NQ; ++e) {
                const int H = WHOLE ? 0 : K >> 1, es = WHOLE ? K * NQ + e : (K & 1) * 8 + e;
                int yo, ro;
                row_offsets(H, es, yo, ro);
                f32x4 v;
                if (!WHOLE && K < 2) v = hold[es];
                else v = *reinterpret_cast<const f32x4*>(stg + (es * RPE + sr0) * BN + cq * 4);
                finish(v, rr[RES ? K % RD : 0][RES ? e : 0], mk[MASK ? K % RD : 0][MASK ? e : 0], yo);
                if constexpr (RES) {
                    if (K + RD < 4) rr[K % RD][e] = WHOLE ? res_load(0, (K + RD) * NQ + e) : res_load((K + RD) >> 1, ((K + RD) & 1) * 8 + e);
                }
                if constexpr (MASK) {
                    if (K + RD < 4) mk[K % RD][e] = mask_load(0, (K + RD) * NQ + e);
                }
                __builtin_amdgcn_sched_barrier(0);        // one quad at a time: interleaving the quads only costs registers here (spills = vmcnt(0) waits)
            }
        };

        // TR = 1: after the last drain step the lanes of a wave that share a channel quad add up their shares (butterfly over the lane
        // bits above the quad index: the same sum in every lane) and the wave leaves one entry per quad in LDS (stats_publish, in front
        // of a barrier); behind the barrier the first QPR threads add the four waves' entries in wave order and write the slab
        double* const red_d = reinterpret_cast<double*>(lds + STG + 2 * HBUF);
        float* const red_f = reinterpret_cast<float*>(lds + STG + 2 * HBUF + 4 * QPR * 64);
        auto stats_publish = [&]() __attribute__((always_inline)) {
            if constexpr (BNST) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int m = 32; m >= QPR; m >>= 1) {
                        bs[c] += __shfl_xor(bs[c], m);
                        bq[c] += __shfl_xor(bq[c], m);
                        blo[c] = fminf(blo[c], __shfl_xor(blo[c], m));
                        bhi[c] = fmaxf(bhi[c], __shfl_xor(bhi[c], m));
                    }
                }
                if (lane < QPR) {
                    const int slot = ((wave - 4) * QPR + lane) * 8;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        red_d[slot + c] = bs[c];
                        red_d[slot + 4 + c] = bq[c];
                        red_f[slot + c] = blo[c];
                        red_f[slot + 4 + c] = bhi[c];
                    }
                }
            }
        };
        auto stats_reduce = [&]() __attribute__((always_inline)) {
            if constexpr (BNST) {
                if (P.valid && lt < QPR) {          // (wave 4: its own entry is in the registers)
#pragma unroll
                    for (int k = 1; k < 4; ++k)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            bs[c] += red_d[(k * QPR + lt) * 8 + c];
                            bq[c] += red_d[(k * QPR + lt) * 8 + 4 + c];
                            blo[c] = fminf(blo[c], red_f[(k * QPR + lt) * 8 + c]);
                            bhi[c] = fmaxf(bhi[c], red_f[(k * QPR + lt) * 8 + 4 + c]);
                        }
                    const int64_t slab = ((int64_t)P.grp * a.tilesM + P.tm) * 2 * d.cout + co;
                    float* const mm = reinterpret_cast<float*>(a.bn_part + (int64_t)d.groups * a.tilesM * 2 * d.cout);   // the extremes sit behind the sums
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        a.bn_part[slab + c] = bs[c];
                        a.bn_part[slab + d.cout + c] = bq[c];
                        mm[slab + c] = blo[c];
                        mm[slab + d.cout + c] = bhi[c];
                    }
                }
                stats_clear();
            }
        };

        pend_setup(P);                                        // no tile parked yet: valid descriptors, every offset out of range (the first tile's drains)
        setup(t_first);
        issue(0, 0);                                          // chunks 0, 1 -> sets 0, 1
        if constexpr (NXS == 2) { advance(); issue(1, lck); }
        convert(0, 0);
        advance(); issue(0, lck);                             // chunk 2 (one set: chunk 1)
        barrier();                                            // chunk 0 staged
        // chunk step (the multiplying waves work on chunk ck of the current tile): the chunk after it (register set SET) -> the other
        // LDS buffer, then the set is reloaded with the chunk three after it
        int it = 0;
        const int total = my_tiles * NC;
        auto step = [&](const int SET_) __attribute__((always_inline)) {
            const int SET = NXS == 1 ? 0 : SET_;
#ifndef TAPX_EXP_NOLOADER
            if (it + 1 < total) convert(SET, (it + 1) & 1);
            advance(); issue(SET, lck);
#endif
            ++it;
        };
        unsigned long long c_work = 0, c_drain = 0, c_bar = 0, c_hand = 0, c_t0 = TAPX_T(), ts = 0;
        auto tbar = [&]() __attribute__((always_inline)) {      // barrier, its wait booked apart from the work in front of it
            const unsigned long long t1 = TAPX_T();
            c_work += t1 - ts;
            barrier();
            ts = TAPX_T();
            c_bar += ts - t1;
        };
        ts = TAPX_T();
        for (int k = 0; k < my_tiles; ++k) {
            // the first four chunks of a tile carry the parked tile's epilogue (NC is even: chunk ck converts set (ck + 1) & 1)
            unsigned long long d0;
            step(1); d0 = TAPX_T(); drain(0); c_drain += TAPX_T() - d0; tbar();
            step(0); d0 = TAPX_T(); drain(1); c_drain += TAPX_T() - d0; tbar();
            step(1); d0 = TAPX_T(); drain(2); c_drain += TAPX_T() - d0; tbar();
            step(0); d0 = TAPX_T(); drain(3); c_drain += TAPX_T() - d0; stats_publish(); tbar();
            stats_reduce();
            for (int ck = 4; ck < NC; ck += 2) {
                step(1); tbar();
                step(0); tbar();
            }
            const unsigned long long h0 = TAPX_T();
            // ---- tile k is complete: its accumulators arrive through the staging area, half a tile at a time
            const Tile Tk = tile_of(t_first + k * t_step);
            P = Pend{Tk.grp, Tk.tm, Tk.tn, 1};
            pend_setup(P);
            barrier();                                        // X1: the tile (FN = 2: its first half) is staged
            if constexpr (!WHOLE) {
#pragma unroll
                for (int e = 0; e < 16; ++e) hold[e] = *reinterpret_cast<const f32x4*>(stg + (e * RPE + sr0) * BN + cq * 4);
            }
            if constexpr (RES) {
#pragma unroll
                for (int q = 0; q < RD; ++q)
#pragma unroll
                    for (int e = 0; e < NQ; ++e) rr[q][e] = WHOLE ? res_load(0, q * NQ + e) : res_load(0, e);
            }
            if constexpr (MASK) {
#pragma unroll
                for (int q = 0; q < RD; ++q)
#pragma unroll
                    for (int e = 0; e < NQ; ++e) mk[q][e] = mask_load(0, q * NQ + e);
            }
            if constexpr (!WHOLE) {
                barrier();                                    // X2: staging area free again
                barrier();                                    // X3: second half staged
            }
            ts = TAPX_T();
            c_hand += ts - h0;
        }
        if (TAPX_DBG && a.dbg && lane == 0) {
            unsigned long long* o = a.dbg + ((int64_t)bid * 8 + wave) * 8;
            o[0] = c_work; o[1] = c_bar; o[2] = c_hand; o[3] = TAPX_T() - c_t0; o[4] = (unsigned long long)my_tiles * NC; o[5] = c_drain;
        }
        drain(0);
        drain(1);
        drain(2);
        drain(3);
        if constexpr (BNST) {
            stats_publish();
            barrier();                                        // (the multiplying waves come to this one too)
            stats_reduce();
        }
        if (a.amax_out) amax_flush(a.amax_out, amx, bid * 4 + wave);
        return;
    }

    // ---------------------------------------------------------------------- multiplying waves
    if (TAPX_PRIO) __builtin_amdgcn_s_setprio(TAPX_PRIO);
    const int wm = wave / WN, wn = wave % WN;
    int abase[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * 128 + i * 32 + l31, il = ml / PPI, mq = ml - il * PPI, r = mq / wo, c = mq - r * wo;
        abase[i] = (PW ? ml : il * HPI + r * WP + c) * 16 + half * (PLANE / 2);
    }
    // the nine taps in the order they are multiplied: byte offset of the tap's window inside a plane, index of its weights
    //   stride 1: tap (kh, kw) = the window shifted by (kh, kw) pixels
    //   stride 2: class order (even, even) | (even, odd) x 2 | (odd, even) x 2 | (odd, odd) x 4; unit shifts inside the class plane
    constexpr int TID[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
    constexpr int TCL[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
    constexpr int TDR[9] = {0, 0, 0, 0, 1, 0, 0, 1, 1};
    constexpr int TDC[9] = {0, 0, 1, 0, 0, 0, 1, 0, 1};
    // (data gradient, stride 1: tap (kh, kw) reads dy at (y + 1 - kh, x + 1 - kw) - the mirrored window)
    const int tmir = (STRIDE == 1 && d.transposed) ? (2 * WP + 2) * 16 : 0, tsgn = (STRIDE == 1 && d.transposed) ? -1 : 1;
    auto tap_off = [&](int tap) __attribute__((always_inline)) {
        if (PW) return tap * NPL * PLANE;
        return STRIDE == 1 ? tmir + tsgn * ((tap / 3) * WP + (tap % 3)) * 16 : TCL[tap] * CLS * 16 + (TDR[tap] * WP + TDC[tap]) * 16;
    };
    auto tap_w = [&](int tap) __attribute__((always_inline)) { return STRIDE == 1 ? tap : TID[tap]; };
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // weight fragments: NSET register sets (a divisor of the nine taps, so that a tap's set is a compile-time constant), requested
    // AHEAD taps before they are multiplied.  L2 latency under load is above one microsecond: two taps of the narrow wave tile
    // (2 x 12 MFMAs = 768 cycles) do not cover it.
    constexpr int NSET = PW ? 4 : (FN == 1 ? 9 : 3);
#ifndef TAPX_AHEAD1
#define TAPX_AHEAD1 5
#endif
    constexpr int AHEAD = PW ? 3 : (FN == 1 ? TAPX_AHEAD1 : 2);
    static_assert(AHEAD < NSET && NT % NSET == 0, "weight register sets");
    u32x4 af[FM][NPL], bf[NSET][FN][NPL];

    const int FSTR = a.ktiles * 2 * NPL * 1024;              // bytes between column fragments of the weight image
    struct WTile { __amdgpu_buffer_rsrc_t rb; int bvo; };
    auto wtile_of = [&](const Tile& T) __attribute__((always_inline)) {
        WTile W;
        W.rb = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(reinterpret_cast<const uint8_t*>(a.w) + (int64_t)T.grp * d.gw * 2), 0, 0x80000000u, 0x00020000);
        W.bvo = (T.tn * NFB + wn * FN) * FSTR + lane * 16;      // (NFB = BN / 32 column fragments per tile, FN of them per wave)
        return W;
    };
    auto load_b_plane = [&](const WTile& W, int ck, int tap, int set, const int pl) __attribute__((always_inline)) {
        const int so = PW ? (ck * 4 + tap) * NPL * 1024 : (((ck >> 1) * 9 + tap_w(tap)) * 2 * NPL + (ck & 1) * NPL) * 1024;
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[set][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(W.rb, W.bvo + j * FSTR + pl * 1024, so, 0);
    };
    auto load_b = [&](const WTile& W, int ck, int tap, int set) __attribute__((always_inline)) {
        load_b_plane(W, ck, tap, set, 0);
        load_b_plane(W, ck, tap, set, 1);
    };
    // EARLY (experiment, off): a weight plane of tap T + NSET requested the moment its last product of tap T has been issued - products
    // in the order (l,h) (h,h) (h,l), so the h plane is free after the second product: 2 1/3 taps of lead instead of 2 for the wide wave
    // tile, whose registers stop at three sets.  Measured in the pipeline: no change (3x3 + wide 1x1 launches 6.95-7.03 ms per forward with,
    // 7.00-7.03 without) - like the nine-set variant of the narrow tile: the weight path is not latency-bound.
#ifndef TAPX_EARLY
#define TAPX_EARLY 0
#endif
    constexpr bool EARLY = TAPX_EARLY && FN == 2;
    auto read_a = [&](int base, int tap, int pl) __attribute__((always_inline)) {
        const int to = tap_off(tap);
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i][pl] = *reinterpret_cast<const u32x4*>(lb + base + pl * PLANE + abase[i] + to);
    };
    // one chunk out of LDS buffer `base`; the weights of the position two taps ahead are requested in front of each tap's MFMAs
    // (X: the tile the chunk after this one belongs to, ckn its index there; behind the very last chunk they re-read the first
    // weights, harmlessly)
    auto chunk = [&](int base, int ck, const WTile& W, const WTile& X, int ckn) __attribute__((always_inline)) {
        read_a(base, 0, 1);
        read_a(base, 0, 0);
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, EARLY ? 0 : 1, EARLY ? 1 : 0};
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int pc = tap % NSET, pn = (tap + AHEAD) % NSET;
#ifndef TAPX_EXP_NOB       // (TAPX_EXP_*: elimination builds for tools/tapx_stamps.py - timing only, the results are wrong)
            if constexpr (!EARLY) {
                if (tap + AHEAD < NT) load_b(W, ck, tap + AHEAD, pn);
                else load_b(X, ckn, tap + AHEAD - NT, pn);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NPR; ++t) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
#ifdef TAPX_EXP_SHAPE16     // timing experiment (wrong results): the same MACs as two v_mfma_f32_16x16x32_f16 - which clock does the chip hold on that shape?
                        {
                            typedef float f32x4m __attribute__((ext_vector_type(4)));
                            f32x4m c0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}, c1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
                            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i][PA[t]]), __builtin_bit_cast(f16x8, bf[pc][j][PB[t]]), c0, 0, 0, 0);
                            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i][PA[t]]), __builtin_bit_cast(f16x8, bf[pc][j][PB[t]]), c1, 0, 0, 0);
                            acc[i][j][0] = c0[0]; acc[i][j][1] = c0[1]; acc[i][j][2] = c0[2]; acc[i][j][3] = c0[3];
                            acc[i][j][4] = c1[0]; acc[i][j][5] = c1[1]; acc[i][j][6] = c1[2]; acc[i][j][7] = c1[3];
                        }
#else
                        acc[i][j] = mfma_split<NPL>(af[i][PA[t]], bf[pc][j][PB[t]], acc[i][j]);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
#ifndef TAPX_EXP_NOB
                if constexpr (EARLY) {
                    if (t >= 1) {       // plane t - 1 of this tap's set is dead: the same plane of tap + NSET
                        if (tap + NSET < NT) load_b_plane(W, ck, tap + NSET, pc, t - 1);
                        else load_b_plane(X, ckn, tap + NSET - NT, pc, t - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#endif
#ifndef TAPX_EXP_NOA
                if (tap + 1 < NT && split_free_a(NPL, t) >= 0) {
#else
                if (false) {
#endif
                    read_a(base, tap + 1, split_free_a(NPL, t));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    // park the accumulators in the staging area (raw: the loading waves undo the pre-scales) and clear them: FN = 2 half H of them
    // (fragment rows 2H, 2H + 1 of every wave), FN = 1 all four fragment rows
    auto park = [&](const int H) __attribute__((always_inline)) {
        float* const sp = stg + (wm * (WHOLE ? 128 : 64) + 4 * half) * BN + wn * 32 * FN + l31;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int ii = 0; ii < (WHOLE ? 4 : 2); ++ii)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = WHOLE ? ii : 2 * H + ii;
#ifdef TAPX_EXP_SHAPE16     // (registers 8-15 never accumulate in that build: park copies of 0-7 so that the data downstream keeps its statistics)
                    sp[(ii * 32 + (r & 3) + 8 * (r >> 2)) * BN + j * 32] = acc[i][j][r & 7];
#else
                    sp[(ii * 32 + (r & 3) + 8 * (r >> 2)) * BN + j * 32] = acc[i][j][r];
#endif
                    acc[i][j][r] = 0.f;
                }
    };

    Tile T = tile_of(t_first);
    WTile W = wtile_of(T);
#pragma unroll
    for (int t = 0; t < (EARLY ? NSET : AHEAD); ++t) load_b(W, 0, t, t);
    barrier();                                                // chunk 0 staged
    int it = 0;
    unsigned long long c_mul = 0, c_bar = 0, c_park = 0, c_t0 = TAPX_T();
    for (int k = 0; k < my_tiles; ++k) {
        const bool has_next = k + 1 < my_tiles;
        const Tile Tn = has_next ? tile_of(t_first + (k + 1) * t_step) : T;
        const WTile Wn = wtile_of(Tn);
        for (int ck = 0; ck + 1 < NC; ++ck) {
            const unsigned long long s0 = TAPX_T();
            chunk((it & 1) * HBUF, ck, W, W, ck + 1);
            const unsigned long long s1 = TAPX_T();
            barrier();
            if (TAPX_DBG) { c_mul += s1 - s0; c_bar += TAPX_T() - s1; }
            ++it;
        }
        const unsigned long long s0 = TAPX_T();
        chunk((it & 1) * HBUF, NC - 1, W, Wn, 0);
        const unsigned long long s1 = TAPX_T();
        barrier();
        const unsigned long long s2 = TAPX_T();
        ++it;
        park(0);
        barrier();                                            // X1
        if constexpr (!WHOLE) {
            barrier();                                        // X2: the loading waves hold the first half in registers
            park(1);
            barrier();                                        // X3
        }
        if (TAPX_DBG) { c_mul += s1 - s0; c_bar += s2 - s1; c_park += TAPX_T() - s2; }
        T = Tn;
        W = Wn;
    }
    if constexpr (BNST) barrier();                            // the loading waves' last statistics hand-over
    if (TAPX_DBG && a.dbg && lane == 0) {
        unsigned long long* o = a.dbg + ((int64_t)bid * 8 + wave) * 8;
        o[0] = c_mul; o[1] = c_bar; o[2] = c_park; o[3] = TAPX_T() - c_t0; o[4] = (unsigned long long)my_tiles * NC;
    }
}

int g_tapx = getenv("EGR_CONV_TAPX") ? atoi(getenv("EGR_CONV_TAPX")) : 1;                          // 0: the 3x3 launches stay on conv_igemm_tap[2]_kernel
int g_tapx_min_tiles = getenv("EGR_CONV_TAPX_MIN_TILES") ? atoi(getenv("EGR_CONV_TAPX_MIN_TILES")) : 256; // tiles (all groups) from which the role-split kernel is used
int g_tapx_blocks = getenv("EGR_CONV_TAPX_BLOCKS") ? atoi(getenv("EGR_CONV_TAPX_BLOCKS")) : 256;    // resident workgroups (one per CU)
int g_tapx_pw = getenv("EGR_CONV_TAPX_PW") ? atoi(getenv("EGR_CONV_TAPX_PW")) : 1;                  // 0: 1x1 launches with >= 256 input channels stay on the tiled kernel
int g_tapx_tpw = getenv("EGR_CONV_TAPX_TPW") ? atoi(getenv("EGR_CONV_TAPX_TPW")) : 0;               // > 0: tiles per workgroup of a non-persistent launch (experiment)
int g_tapx_train = getenv("EGR_CONV_TAPX_TRAIN") ? atoi(getenv("EGR_CONV_TAPX_TRAIN")) : 1;         // 0: statistics-epilogue / masked launches stay on conv_igemm_tap_kernel
int g_tapx_fn = getenv("EGR_CONV_TAPX_FN") ? atoi(getenv("EGR_CONV_TAPX_FN")) : 0;                  // wave tile: 0 by shape, 1: 128 x 32, 2: 128 x 64 wherever it exists

}  // namespace

namespace egrc {

int tapx_set(int on, int min_tiles, int blocks) {
    if (on >= 0) { g_tapx = on != 0; g_tapx_fn = on >= 2 ? on - 1 : 0; }      // on = 2 / 3: the 128 x 32 / 128 x 64 wave tile wherever it exists (tests)
    if (min_tiles >= 0) g_tapx_min_tiles = min_tiles;
    if (blocks > 0) g_tapx_blocks = blocks;
    return 0;
}

// Launch conv_tapx_kernel if the problem is one it covers: returns TAPX_NO (nothing launched) or the launch status.
// `a` arrives from conv_run with the geometry fields filled in (M, Npad, K, cblocks, ktiles, *_shift, *_plain, vec_ok, cls_mode).
int tapx_try(ConvArgs& a, int64_t yspan_floats, int64_t rspan_floats, hipStream_t stream) {
    egr_conv_desc& d = a.d;
    const bool pw = d.kh == 1 && d.kw == 1;
    if (!g_tapx || d.w_format != EGR_W_F16X2 || !(pw ? (d.pad == 0 && d.stride == 1 && g_tapx_pw) : (d.kh == 3 && d.kw == 3 && d.pad == 1)) ||
        (d.transposed && (pw || d.stride != 1)) || a.cls_mode || d.split_k > 1 ||
        d.out_nchw || a.rowscale || a.rowmask || (a.mask && a.bn_part) || !a.vec_ok || d.cout % 4 != 0 || d.cin < 64 ||
        (d.act != EGR_ACT_NONE && d.act != EGR_ACT_RELU) || d.res_mode == EGR_RES_UP2_BEFORE_ACT || (a.dbg && !TAPX_DBG))
        return TAPX_NO;
    if (yspan_floats * 4 >= (1LL << 31) || (d.res_mode && rspan_floats * 4 >= (1LL << 31))) return TAPX_NO;   // 32-bit byte offsets in the epilogue
    // training launches (statistics epilogue: raw output of a 3x3 conv; masked data gradient: 3x3 / stride 1): the narrow wave tile
    const int tr = a.bn_part ? 1 : (a.mask ? 2 : 0);
    if (tr && (pw || !g_tapx_train)) return TAPX_NO;
    if (tr == 1 && (d.act != EGR_ACT_NONE || d.res_mode != EGR_RES_NONE || d.transposed || d.cout % 64 != 0)) return TAPX_NO;
    if (tr == 2 && (d.stride != 1 || a.scale || a.shift || d.act != EGR_ACT_NONE || d.res_mode == EGR_RES_AFTER_ACT)) return TAPX_NO;
    // (the wide wave tile has no registers left for the mask quads; at >= 128 channels the narrow one is no faster than
    // conv_igemm_tap_kernel - 345 against 340 TFLOP/s at 128 channels, 346 against 358 at 256 - so masked launches come here for 64-channel
    // outputs only.  The statistics epilogue exists on both wave tiles for stride 1, on the narrow one for stride 2.)
    if (tr == 2 && a.Npad % 128 == 0 && g_tapx_train < 2 && g_tapx_min_tiles > 1) return TAPX_NO;     // (forced from one tile up: tests)
    const int fn = (tr == 2 || (tr == 1 && d.stride == 2)) ? 1 : g_tapx_fn;
    const int P = d.ho * d.wo;
    if (a.howo_shift < 0 || a.wo_shift < 0) return TAPX_NO;
    const int ext = d.stride == 1 ? 2 : 1;
    auto fits = [&](int bm, int bn) {        // tiles of whole image rows / whole small images whose planes fit the kernel's LDS buffers
        if (a.Npad % bn != 0 || a.M % bm != 0) return false;
        if (!pw) {
            if (!((P % bm == 0) || (bm % P == 0)) || (bm / P) > 15) return false;
            const int hp = P >= bm ? (bm / d.wo + ext) * (d.wo + ext) : (bm / P) * (d.ho + ext) * (d.wo + ext);
            if (hp > (d.stride == 1 ? tapx_hp(bm, 1) : tapx_cls(bm))) return false;
        }
        // enough tiles for every CU - and, below two tiles per CU, a K loop long enough to carry the epilogue that then runs exposed
        // (one tile per workgroup: 128 -> 128 at 32 x 32 pixels measured 0.070 ms against 0.063 on the tap kernel, layer4 0.551 against 0.565)
        const int64_t tiles = (int64_t)(a.M / bm) * (a.Npad / bn) * d.groups;
        // training launches (config 5 at batch 32 has 256-1024 tiles per trunk launch): from four tiles per CU - measured per launch,
        // statistics epilogue, role-split against tap-sharing kernel: 64 channels / 2048 tiles 0.538 against 0.573 ms, 128 channels /
        // 1024 narrow tiles 0.448 against 0.471 (512 wide tiles: 0.478), 256 channels / 512 tiles 0.335-0.349 against 0.324
        if (tr) return tiles >= 4 * (int64_t)g_tapx_min_tiles || g_tapx_min_tiles <= 1;
        // below 2048 rows the alternative is the generic split kernel (the tap-sharing ones start there): a quarter of the CUs on this
        // kernel is still faster - the refiners' 256 -> 512 stride-2 conv at batch 1 (128 tiles) 72 us there
        if (!pw && a.M < 2048 && tiles >= g_tapx_min_tiles / 4 && a.cblocks >= 4) return true;
        return tiles >= g_tapx_min_tiles && (tiles >= 2 * (int64_t)g_tapx_min_tiles || (!pw && a.cblocks >= 16) || g_tapx_min_tiles <= 1);
    };
    // tiles (rows x columns): the 128 x 64 wave tile (FN = 2) wherever the channel count allows it - stride 1: 256 x 128, stride 2:
    // 128 x 256 - else the 128 x 32 one (FN = 1): stride 1: 256 x 64 (64 / 192 channels; measured in the pipeline at batch 64:
    // layer1 346 TFLOP/s against 291 on conv_igemm_tap_kernel and 245 on 512 x 64 tiles) or 128 x 128, stride 2: 128 x 128
    int cfg = -1;
    if (pw) {      // 1x1 with >= 256 input channels (chunks of 64): 128 x 256 tiles, else 128 x 128
        if (d.cin % 128 != 0 || d.cin < 256 || d.h != d.ho || d.w != d.wo) return TAPX_NO;      // (an even number of 64-channel chunks, at least four)
        cfg = (g_tapx_fn != 1 && fits(128, 256)) ? 6 : (g_tapx_fn != 2 && fits(128, 128) ? 7 : (fits(128, 256) ? 6 : -1));
    } else if (d.stride == 1) {
        if (!(d.wo == 8 || d.wo == 16 || d.wo == 32 || d.wo == 64) || d.ho != d.h || d.wo != d.w) return TAPX_NO;
        if (a.Npad % 128 == 0) cfg = (fn != 1 && fits(256, 128)) ? 0 : (fn != 2 && fits(128, 128) ? 1 : (fn != 1 && fits(256, 128) ? 0 : -1));
        else cfg = fits(256, 64) ? 3 : -1;
    } else if (d.stride == 2) {
        if (!(d.wo == 8 || d.wo == 16 || d.wo == 32) || d.h != 2 * d.ho || d.w != 2 * d.wo) return TAPX_NO;
        if (a.M < 2048 && fn == 0) cfg = fits(128, 128) ? 5 : (fits(128, 256) ? 4 : -1);       // (few rows: the narrower tile = twice the workgroups)
        else cfg = (fn != 1 && fits(128, 256)) ? 4 : (fn != 2 && fits(128, 128) ? 5 : (fn != 1 && fits(128, 256) ? 4 : -1));
    }
    if (cfg < 0) return TAPX_NO;
    static const int kbm[8] = {256, 128, 512, 256, 128, 128, 128, 128}, kbn[8] = {128, 128, 64, 64, 256, 128, 256, 128};
    const int bm = kbm[cfg], bn = kbn[cfg];
    const int64_t tiles = (int64_t)(a.M / bm) * (a.Npad / bn) * d.groups;
    if (tiles >= (1 << 30)) return TAPX_NO;
    d.split_k = 1;
    a.ktiles_per_split = a.ktiles;
    a.tilesM = a.M / bm;
    a.tilesN = a.Npad / bn;
    a.dTilesN = make_fastdiv(a.tilesN);
    a.ntiles = a.tilesM * a.tilesN;
    if (const int rcb = bn_slabs(a)) return rcb;
    // workgroups: one per CU walking tiles / blocks tiles each - or, with g_tapx_tpw > 0 (experiment), more workgroups of about that many
    // tiles each, handed to CUs as they free up: measured 6647 frames/s persistent, 6588 / 6511-6520 / 6334 with 8 / 4 / 2 tiles per workgroup
    int64_t wgs = g_tapx_blocks;
    if (g_tapx_tpw > 0 && tiles / g_tapx_tpw > wgs) wgs = (tiles / g_tapx_tpw + 7) / 8 * 8;
    const unsigned grid = (unsigned)(tiles < wgs ? tiles : wgs);
    auto launch = [&](auto res_tag) {
        constexpr bool R = decltype(res_tag)::value;
        switch (cfg) {
            case 0: hipLaunchKernelGGL((conv_tapx_kernel<2, 2, 2, 1, R>), dim3(grid), dim3(512), 0, stream, a); break;
            case 1: hipLaunchKernelGGL((conv_tapx_kernel<1, 4, 1, 1, R>), dim3(grid), dim3(512), 0, stream, a); break;
            case 3: hipLaunchKernelGGL((conv_tapx_kernel<2, 2, 1, 1, R>), dim3(grid), dim3(512), 0, stream, a); break;
            case 4: hipLaunchKernelGGL((conv_tapx_kernel<1, 4, 2, 2, R>), dim3(grid), dim3(512), 0, stream, a); break;
            case 5: hipLaunchKernelGGL((conv_tapx_kernel<1, 4, 1, 2, R>), dim3(grid), dim3(512), 0, stream, a); break;
            case 6: hipLaunchKernelGGL((conv_tapx_kernel<1, 4, 2, 0, R>), dim3(grid), dim3(512), 0, stream, a); break;
            default: hipLaunchKernelGGL((conv_tapx_kernel<1, 4, 1, 0, R>), dim3(grid), dim3(512), 0, stream, a); break;
        }
    };
    // (false: no instantiation for this tile / epilogue pair - nothing was launched and the caller's other kernels take the problem)
    auto launch_tr = [&](auto res_tag, auto tr_tag) -> bool {
        constexpr bool R = decltype(res_tag)::value;
        constexpr int T = decltype(tr_tag)::value;
        switch (cfg) {
            case 0:
                if constexpr (T == 1) { hipLaunchKernelGGL((conv_tapx_kernel<2, 2, 2, 1, R, T>), dim3(grid), dim3(512), 0, stream, a); return true; }
                return false;
            case 1: hipLaunchKernelGGL((conv_tapx_kernel<1, 4, 1, 1, R, T>), dim3(grid), dim3(512), 0, stream, a); return true;
            case 3: hipLaunchKernelGGL((conv_tapx_kernel<2, 2, 1, 1, R, T>), dim3(grid), dim3(512), 0, stream, a); return true;
            case 5:
                if constexpr (T == 1) { hipLaunchKernelGGL((conv_tapx_kernel<1, 4, 1, 2, R, T>), dim3(grid), dim3(512), 0, stream, a); return true; }
                return false;
            default: return false;
        }
    };
    bool launched = true;
    if (tr == 1) launched = launch_tr(std::false_type{}, std::integral_constant<int, 1>{});
    else if (tr == 2 && d.res_mode != EGR_RES_NONE) launched = launch_tr(std::true_type{}, std::integral_constant<int, 2>{});
    else if (tr == 2) launched = launch_tr(std::false_type{}, std::integral_constant<int, 2>{});
    else if (d.res_mode != EGR_RES_NONE) launch(std::true_type{});
    else launch(std::false_type{});
    if (!launched) return TAPX_NO;
    return egr_launch_status();
}

}  // namespace egrc
