// Implicit-GEMM convolution / linear layer on the fp32 matrix cores of gfx950.
//
//   GEMM view:  M = n*ho*wo output pixels,  N = cout,  K = kh*kw*cin,  k = (ci/32, kh, kw, ci%32)
//   A[m][k]  = x[n, ho*s-p+kh, wo*s-p+kw, ci]   (NHWC gather, zero outside the image)
//   B[k][co] = w[co][k]                           (weights pre-packed [cout_pad][cin/32][kh*kw][32])
//
// Structure:
//  * a workgroup (4 waves) owns a BM x BN output tile and walks K in chunks of 32 = one filter tap x 32
//    input channels, so every A row of a chunk is 128 contiguous bytes of one pixel;
//  * staging is DIRECT-TO-LDS (global_load_lds_dwordx4, no VGPR round trip, no ds_write): chunk t+1 streams
//    into the second LDS buffer while chunk t is multiplied; one barrier per chunk.  A wave-instruction
//    writes 1 KiB = 8 rows x 128 B linearly, so rows are unpadded; bank conflicts of the ds_read_b128
//    fragment reads are removed by an XOR swizzle applied on the SOURCE side (lane (row, seg) fetches
//    logical 16-byte segment seg ^ ((row>>1)&7)) and undone by the same XOR on the read.  Pixels outside
//    the image (conv halo) and weight rows beyond cout_pad fetch from a 16-byte zero buffer instead;
//  * each wave owns FM x FN accumulator tiles of v_mfma_f32_32x32x2_f32; one ds_read_b128 gives a lane four
//    consecutive k of its row, feeding four MFMA k-steps (lane half h covers k = kk+4h+t in step t — A and B
//    use the same permutation, so each of the chunk's 32 products is summed exactly once);
//  * the epilogue goes through LDS once more so that global stores (and residual loads) are 16 bytes per
//    lane and whole 512-byte rows per wave: scale/shift (BatchNorm or bias), per-row bias scale, residual
//    before/after the activation, ReLU / exact-erf GELU, row mask; NHWC (any pixel stride / channel offset)
//    or channel-major planes.
// f32 MFMA is a k-ordered fp32 fma chain: results are deterministic and fp32-exact in the reference's sense.
//  * X6 = true (EGR_W_BF16X3): the same GEMM on the bf16 matrix cores without giving up that exactness: every
//    fp32 operand is the exact sum of three bf16 numbers, the six partial products of order <= 2 are accumulated in fp32
//    (the dropped ones are below fp32 rounding).  Weights arrive pre-split in fragment order (egr_pack_w6_f32), the
//    activations are split while they are staged.  Same row table, modes and epilogue as the fp32 main loop.
#include <mutex>
#include <type_traits>

#include <cstdlib>

#include "egr_conv_shared.h"

using namespace egrc;

namespace {


#ifndef X6_OCC_SMALL
#define X6_OCC_SMALL 4
#endif
#ifndef X6_OCC_BIG
#define X6_OCC_BIG 2
#endif

// resident workgroup slots a persistent split-bf16 launch is sized for (2 per CU x 256 CUs); 0 = one workgroup per tile
int g_persist = getenv("EGR_CONV_PERSIST") ? atoi(getenv("EGR_CONV_PERSIST")) : 512;
int g_persist_ktiles = getenv("EGR_CONV_PERSIST_KTILES") ? atoi(getenv("EGR_CONV_PERSIST_KTILES")) : 4;   // K <= 128: persistent

__device__ __attribute__((aligned(16))) float egr_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// halo pixels of a tap-sharing tile of BM output pixels: whole rows of one image ((BM / wo + 2) x (wo + 2)) or whole small images
// (BM / (ho wo) x (ho + 2) x (wo + 2)); wo in {8, 16, 32, 64}: 128 -> 4 x 66, 256 -> 6 x 66
constexpr int tap_hpmax(int bm) { return bm == 256 ? 396 : 264; }
constexpr int tap_plane(int bm) { return tap_hpmax(bm) * 32; }      // bytes per 16-bit plane (16 channels per pixel)
constexpr int tap_hbuf(int bm, int npl) { return npl * tap_plane(bm); }        // bytes per halo buffer (hi, mid, lo / h, l)

template <int BM, int BN, bool X6 = false, bool PERSIST_ = false, bool TAP = false, int NPL = 3>
struct LdsPlan {
    // floats per stage.  fp32 path: BK = 32 deep rows of both operands.  Split-bf16 path: one k16 step of (hi, mid, lo) bf16
    // planes in fragment order, 1 KiB per (32-row fragment, plane)
    static constexpr int TILE = X6 ? (BM / 32 + BN / 32) * NPL * 256 : (BM + BN) * BK;
    static constexpr int CS = BN + 4;                       // epilogue staging row stride
    static constexpr int STAGES = TAP ? 2 * tap_hbuf(BM, NPL) / 4 : 2 * TILE;          // floats of the two stage buffers
    static constexpr int ROWOFF = (STAGES > BM * CS) ? STAGES : BM * CS;      // row offsets (y, res) live past both
    static constexpr bool PERSIST = X6 && PERSIST_;                           // persistent workgroups, see the tile loop of the split kernel
    static constexpr int TABLES = PERSIST ? 2 : 1;                            // the next tile's table is decoded under the epilogue
    // row table columns: y / res offset, [x offset, tap mask - not in the tap-sharing kernel,] two bilinear weights (EGR_RES_UP2_BEFORE_ACT)
    static constexpr int TCOLS = TAP ? 4 : 6, COL_LX = TAP ? 2 : 4, COL_LY = COL_LX + 1;
    static constexpr int FLOATS = ROWOFF + TCOLS * BM * TABLES;
};


template <int BM, int BN, int WM, int WN, bool X6, bool PERSIST = false, bool TAP = false, bool TAP2 = false, int NPL = 3>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int NT = 256;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int IA = BM / 32, IB = BN / 32;  // 32 rows per load pass (8 rows per wave-instruction)
    using P = LdsPlan<BM, BN, X6, PERSIST, TAP, NPL>;
    constexpr int NPR = split_npr(NPL);      // matrix instructions per fp32 product (split kernels)
    static_assert(FM >= 1 && FN >= 1, "wave tile must be >= 32x32");

    __shared__ __attribute__((aligned(16))) float lds[P::FLOATS];

    const egr_conv_desc& d = a.d;
    int stamp_tile = blockIdx.x;   // persistent launches: the tile being worked on (stamps are per tile)
    auto stamp = [&](int slot) {
        if (a.dbg && threadIdx.x == 0)
            a.dbg[((int64_t)blockIdx.z * a.ntiles + stamp_tile) * 8 + slot] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    if (a.dbg && threadIdx.x == 0)  // physical placement, for co-residency analysis (tools/conv_stamps.py)
        a.dbg[((int64_t)blockIdx.z * a.ntiles + blockIdx.x) * 8 + 6] =
            ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) << 16) |
            (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu);
    const int grp = blockIdx.z;  // grouped launch: same shape, own operands
    const float* const xg = a.x + grp * d.gx;
    const float* const wg = a.w + grp * d.gw;
    const float* const scg = a.scale ? a.scale + grp * d.gp : nullptr;
    const float* const shg = a.shift ? a.shift + grp * d.gp : nullptr;
    const float* const resg = a.res ? a.res + grp * d.gr : nullptr;
    const float* const rsg = a.rowscale ? a.rowscale + grp * d.grs : nullptr;
    const uint8_t* const rmg = a.rowmask ? a.rowmask + grp * d.grm : nullptr;
    float* const yg = a.y + grp * d.gy;
    float* const wsg = a.ws ? a.ws + (int64_t)grp * d.split_k * a.M * a.Npad : nullptr;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    // EGR_W_F16X2: activation pre-scale and its inverse (uniform), the weights' per-channel descale; max |y| stored by this thread
    float sa = 1.f, ads = 1.f;
    if constexpr (X6 && NPL == 2) act_prescale(a.amax_in, lane, sa, ads);
    const float* const wdsg = (X6 && NPL == 2) ? a.wds + grp * d.gp : nullptr;
    float amx = 0.f;
    auto track4 = [&](const f32x4& v) { amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3]))); };

    // XCD-aware tile order: blocks b and b+8 share an L2; hand each XCD a contiguous run of tiles,
    // with the N tiles of one M tile adjacent so the activation tile is fetched into one L2 only.
    // A persistent launch (a.ntiles > gridDim.x, split kernel only) walks tiles blockIdx.x, + gridDim.x, ...: gridDim.x is a
    // multiple of 8 then, so a workgroup's tiles stay on its XCD's run.
    const int nb = a.ntiles;
    auto tile_of = [&](int bid, int& tm_, int& tn_) {
        if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
        tm_ = (a.tilesN == 1) ? bid : fdiv(bid, a.dTilesN);
        tn_ = bid - tm_ * a.tilesN;
    };
    int tm, tn;
    tile_of((int)blockIdx.x, tm, tn);
    // Stride-2 data gradient: output pixel (ho, wo) only sees the taps with kh = (ho+pad) mod 2, kw = (wo+pad) mod 2, so the
    // launch is split into the four output-parity classes (blockIdx.y).  A class enumerates its (ho/2, wo/2) sub-grid
    // and walks only its own taps: no structurally-zero MFMA work (1 + 2 + 2 + 4 of the 9 taps of a 3x3 kernel).
    const int cls = a.cls_mode ? (int)blockIdx.y : 0;
    const int ph = cls >> 1, pw = cls & 1;
    const int kstep = a.cls_mode ? 2 : 1;
    const int kh0 = a.cls_mode ? ((ph + d.pad) & 1) : 0, kw0 = a.cls_mode ? ((pw + d.pad) & 1) : 0;
    const int nkh = a.cls_mode ? max(0, (d.kh - kh0 + 1) >> 1) : d.kh, nkw = a.cls_mode ? max(0, (d.kw - kw0 + 1) >> 1) : d.kw;
    const int taps_blk = nkh * nkw;
    const int split = a.cls_mode ? 0 : (int)blockIdx.y;
    const int kt0 = split * a.ktiles_per_split;
    const int kt1 = a.cls_mode ? taps_blk * a.cblocks : min(a.ktiles, kt0 + a.ktiles_per_split);
    const int HoWo = a.cls_mode ? (d.ho >> 1) * (d.wo >> 1) : d.ho * d.wo;   // pixels per image in the enumeration of m
    const int Wenum = a.cls_mode ? (d.wo >> 1) : d.wo;

    // ---- per-lane staging roles: piece p covers tile rows p*32 + wave*8 + (lane>>3); physical segment lane&7.
    // Row geometry is decoded ONCE per row (thread r < BM handles row r) into a small LDS table: element offset of
    // tap (0,0) of the row's receptive field, bitmask of the taps that fall inside the image, output / residual
    // offsets.  Per chunk a lane then adds the wave-uniform tap offset and tests one mask bit.
    const int rsub = wave * 8 + (lane >> 3);
    const int pseg = lane & 7;
    int* const tab0 = reinterpret_cast<int*>(lds + P::ROWOFF);   // table `slot`: [yoff | roff | xoff | mask | lx | ly] x BM
    const unsigned fullmask = (d.kh * d.kw >= 32) ? 0xffffffffu : ((1u << (d.kh * d.kw)) - 1u);
    auto decode = [&](int tm_, int slot) {
    int* const s_yoff = tab0 + slot * P::TCOLS * BM;
    int* const s_roff = s_yoff + BM;
    int* const s_xoff = s_roff + BM;
    unsigned* const s_mask = reinterpret_cast<unsigned*>(s_xoff + BM);
    for (int r = tid; r < BM; r += NT) {
        const int m = tm_ * BM + r;
        int xo = 0, yo = -1, ro = 0;
        unsigned mk = 0u;
        if (m < a.M) {
            // (n, ho, wo) of output pixel m: shifts when the geometry is a power of two, multiply-high otherwise
            int n, pix, ho, wo;
            if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
            else { n = fdiv(m, a.dHoWo); pix = m - n * HoWo; }
            if (a.wo_shift >= 0) { ho = pix >> a.wo_shift; wo = pix & (Wenum - 1); }
            else { ho = fdiv(pix, a.dWo); wo = pix - ho * Wenum; }
            if (a.cls_mode) { ho = 2 * ho + ph; wo = 2 * wo + pw; pix = ho * d.wo + wo; }
            const int xbase = a.x_plain ? n * (int)d.xmap.stride_inner : (int)fmap(d.xmap, a.dXin, n);
            if (!d.transposed) {
                const int hi0 = ho * d.stride - d.pad, wi0 = wo * d.stride - d.pad;
                // 32-bit element offsets (the host bounds every operand below 2^31 elements); may be negative for halo rows
                xo = xbase + (hi0 * d.w + wi0) * d.ldx;
                // taps inside the image: kh in [kh_lo, kh_hi), kw in [kw_lo, kw_hi)
                const int kh_lo = max(0, -hi0), kh_hi = min(d.kh, d.h - hi0);
                const int kw_lo = max(0, -wi0), kw_hi = min(d.kw, d.w - wi0);
                const unsigned rowbits = (kw_hi > kw_lo) ? (((1u << (kw_hi - kw_lo)) - 1u) << kw_lo) : 0u;
                if (d.kh <= 3) {  // branch-free for the 1x1 / 3x3 layers of the path
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) mk |= (kh >= kh_lo && kh < kh_hi) ? (rowbits << (kh * d.kw)) : 0u;
                } else {
                    for (int kh = kh_lo; kh < kh_hi; ++kh) mk |= rowbits << (kh * d.kw);
                }
            } else {
                // data gradient: output pixel (ho, wo) is an INPUT pixel of the forward conv; tap (kh, kw) reads
                // dy[(ho+pad-kh)/stride, (wo+pad-kw)/stride] when both divisions are exact and in range.  Base = floor
                // of the tap-0 position; tap (kh, kw) sits (kh/stride, kw/stride) pixels before it (see issue_piece).
                const int th = ho + d.pad, tw = wo + d.pad;
                const int bh = th / d.stride, bw = tw / d.stride;          // th, tw >= 0
                xo = xbase + (bh * d.w + bw) * d.ldx;
                for (int kh = 0; kh < d.kh; ++kh) {
                    const int sh_ = th - kh;
                    if (sh_ < 0 || sh_ % d.stride != 0 || sh_ / d.stride >= d.h) continue;
                    for (int kw = 0; kw < d.kw; ++kw) {
                        const int sw_ = tw - kw;
                        if (sw_ < 0 || sw_ % d.stride != 0 || sw_ / d.stride >= d.w) continue;
                        mk |= 1u << (kh * d.kw + kw);
                    }
                }
            }
            mk &= fullmask;
            yo = (a.y_plain ? n * (int)d.ymap.stride_inner : (int)fmap(d.ymap, a.dYin, n)) + (d.out_nchw ? pix : pix * d.ldy);
            if (d.res_mode == EGR_RES_BEFORE_ACT || d.res_mode == EGR_RES_AFTER_ACT) ro = (a.r_plain ? n * (int)d.rmap.stride_inner : (int)fmap(d.rmap, a.dRin, n)) + pix * d.ldr;
            if (d.res_mode == EGR_RES_UP2_BEFORE_ACT) {
                // residual = bilinear x2 upsampling (align_corners=True, ATen arithmetic as in upsample2x_kernel) of a HALF-resolution
                // tensor: offset of the upper-left neighbour, the two weights; sign bit set = the right / lower neighbour is the
                // same pixel (last column / row)
                const int hl = d.ho >> 1, wl = d.wo >> 1;
                const float shh = (d.ho > 1) ? (float)(hl - 1) / (float)(d.ho - 1) : 0.f;
                const float sww = (d.wo > 1) ? (float)(wl - 1) / (float)(d.wo - 1) : 0.f;
                const float fy = shh * (float)ho, fx = sww * (float)wo;
                const int y0 = (int)fy, x0 = (int)fx;
                const float ly1 = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
                ro = (a.r_plain ? n * (int)d.rmap.stride_inner : (int)fmap(d.rmap, a.dRin, n)) + (y0 * wl + x0) * d.ldr;
                s_yoff[P::COL_LX * BM + r] = (int)(__float_as_uint(lx1) | (x0 + 1 > wl - 1 ? 0x80000000u : 0u));
                s_yoff[P::COL_LY * BM + r] = (int)(__float_as_uint(ly1) | (y0 + 1 > hl - 1 ? 0x80000000u : 0u));
            }
        } else if (d.res_mode == EGR_RES_UP2_BEFORE_ACT) {
            s_yoff[P::COL_LX * BM + r] = 0;
            s_yoff[P::COL_LY * BM + r] = 0;
        }
        if constexpr (!TAP) {
            s_xoff[r] = xo;
            s_mask[r] = mk;
        }
        s_yoff[r] = yo;
        s_roff[r] = ro;
    }
    };
    decode(tm, 0);
    int* s_yoff = tab0;
    int* s_roff = s_yoff + BM;
    int* s_xoff = s_roff + BM;
    unsigned* s_mask = reinterpret_cast<unsigned*>(s_xoff + BM);
    // wave-uniform position of a chunk in K: (channel chunk cb, tap (kh, kw)); advanced incrementally
    struct KPos { int cb, kh, kw; };
    auto kpos_of = [&](int kt) {
        KPos p;
        const int tb = taps_blk > 0 ? taps_blk : 1, nw = nkw > 0 ? nkw : 1;
        p.cb = kt / tb;
        const int t = kt - p.cb * tb, th = t / nw;
        p.kh = kh0 + kstep * th;
        p.kw = kw0 + kstep * (t - th * nw);
        return p;
    };
    auto kpos_next = [&](KPos p) {
        p.kw += kstep;
        if (p.kw >= d.kw) { p.kw = kw0; p.kh += kstep; if (p.kh >= d.kh) { p.kh = kh0; ++p.cb; } }
        return p;
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- epilogue of one tile (tm_, tn_) whose row table is (s_yoff, s_roff); `hook` runs on every thread right after the
    // accumulators are staged
    auto epilogue = [&](const int tm, const int tn, const int* const s_yoff, const int* const s_roff, auto&& hook) {
    stamp(3);  // k loop done
    // Split launches with a residual: the tile's residual quads are requested HERE, before the accumulators are staged, and held in
    // registers (the operand registers of the K loop are dead) - one HBM round trip that overlaps the staging and its barrier, instead of
    // two batches of eight loads behind it (layer1's conv2: 192 us with the residual from HBM, 141 with it in L2, 130 without one).
    constexpr int E_QPR = BN / 4, E_RPI = NT / E_QPR, E_IT = BM / E_RPI;
    constexpr bool RPRE = X6 && !PERSIST;      // (the persistent variant holds the next tile's operands across its epilogue: no room)
    constexpr bool MPRE = RPRE && (TAP || TAP2);   // (mask + residual: 2 x E_IT quads - the tap kernels have the registers, the 128-register tiles do not)
    f32x4 rpre[RPRE ? E_IT : 1], mpre[MPRE ? E_IT : 1];
    if constexpr (RPRE) {
        const int co_e = tn * BN + (tid % E_QPR) * 4;
        const bool pre = (d.res_mode == EGR_RES_BEFORE_ACT || d.res_mode == EGR_RES_AFTER_ACT) && d.split_k <= 1 && !d.out_nchw && !a.mask &&
                         !a.bn_part && a.vec_ok && co_e + 3 < d.cout && !rsg && !rmg;      // = the conditions of the fast path below
        if (pre) {
#pragma unroll
            for (int it = 0; it < E_IT; ++it) {
                const int row = tid / E_QPR + it * E_RPI;
                const int ro = s_yoff[row] >= 0 ? s_roff[row] : 0;          // (padded rows read element 0: no branch around the loads)
                rpre[it] = *reinterpret_cast<const f32x4*>(resg + (int64_t)ro + co_e);
            }
        }
        if (MPRE && a.mask && d.split_k <= 1 && co_e < d.cout) {   // masked data gradient (training): the mask quads and the residual likewise
            const float* const maskg = a.mask + grp * d.gy;
#pragma unroll
            for (int it = 0; it < E_IT; ++it) {
                const int row = tid / E_QPR + it * E_RPI;
                const int yo = s_yoff[row];
                mpre[it] = *reinterpret_cast<const f32x4*>(maskg + (int64_t)(yo >= 0 ? yo : 0) + co_e);
                if (d.res_mode) rpre[it] = *reinterpret_cast<const f32x4*>(resg + (int64_t)(yo >= 0 ? s_roff[row] : 0) + co_e);
            }
        }
    }
    // ---- epilogue: accumulators -> LDS [BM][BN+4] -> 16-byte row-contiguous global accesses
    float* sC = lds;
    float dsc[FN];   // EGR_W_F16X2: the accumulators carry both pre-scales; undone here by an exact power-of-two factor per column
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        dsc[j] = 1.f;
        if constexpr (X6 && NPL == 2) {
            const int col = tn * BN + wn * TN + j * 32 + l31;
            dsc[j] = ads * (col < a.Npad ? wdsg[col] : 1.f);
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                int row = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                sC[row * P::CS + wn * TN + j * 32 + l31] = (X6 && NPL == 2) ? acc[i][j][r] * dsc[j] : acc[i][j][r];
            }
    __syncthreads();
    stamp(4);  // accumulators staged
    hook();    // (split kernel, persistent: the next tile's first operands are requested here and fly under the stores)

    if (d.split_k > 1) {  // raw partial sums
        constexpr int QPR = BN / 4;
        if (!a.cnt) {     // the epilogue runs in splitk_reduce_kernel
            for (int idx = tid; idx < BM * QPR; idx += NT) {
                int row = idx / QPR, cq = idx - row * QPR;
                int m = tm * BM + row, co = tn * BN + cq * 4;
                if (m < a.M && co < a.Npad)
                    *reinterpret_cast<f32x4*>(&wsg[((int64_t)split * a.M + m) * a.Npad + co]) =
                        *reinterpret_cast<const f32x4*>(&sC[row * P::CS + cq * 4]);
            }
            return;
        }
        // The K slices of a tile arrive in any order; the LAST one to arrive sums all slabs in slice order (the arithmetic of
        // splitk_reduce_kernel: deterministic, independent of the arrival order) and applies the epilogue - no second launch.
        // The slices run on any XCD, whose L2s are not coherent with each other for ordinary accesses; agent-scope FENCES would
        // write back / invalidate the whole L2 per workgroup (measured: the forward 15.3 -> 16.1 ms, batch 1 1.8 -> 3.7 ms).
        // Instead the slab accesses themselves are agent-scope atomic (relaxed: `sc1` stores that write through, `sc1` loads that do
        // not hit stale lines); ordering: every thread waits for its own slab stores (vmcnt(0)) before the barrier that precedes
        // the counter increment, and the slab loads follow the barrier behind the counter read.
        for (int idx = tid; idx < BM * BN; idx += NT) {
            const int row = idx / BN, c = idx - row * BN;
            const int m = tm * BM + row, co = tn * BN + c;
            if (m < a.M && co < a.Npad)
                __hip_atomic_store(&wsg[((int64_t)split * a.M + m) * a.Npad + co], sC[row * P::CS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // every slab store of this workgroup has completed; sC is dead
        int* const flag = reinterpret_cast<int*>(lds);
        if (tid == 0) {
            int* const c = a.cnt + (int64_t)grp * a.ntiles + tm * a.tilesN + tn;
            const int last = atomicAdd(c, 1) == d.split_k - 1;
            if (last) atomicExch(c, 0);                    // ready for the next launch
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        for (int idx = tid; idx < BM * BN; idx += NT) {
            const int row = idx / BN, c = idx - row * BN;
            const int m = tm * BM + row, co = tn * BN + c;
            const int yo = s_yoff[row];
            if (yo < 0 || co >= d.cout) continue;
            float sum = 0.f;
            {   // four slab loads in flight, added in slice order (one dependent load per slice is a latency chain)
                const int64_t slab = (int64_t)a.M * a.Npad;
                const float* const p0 = &wsg[(int64_t)m * a.Npad + co];
                int sp = 0;
                for (; sp + 4 <= d.split_k; sp += 4) {
                    float t4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) t4[u] = __hip_atomic_load(p0 + (int64_t)(sp + u) * slab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int u = 0; u < 4; ++u) sum += t4[u];
                }
                for (; sp < d.split_k; ++sp) sum += __hip_atomic_load(p0 + (int64_t)sp * slab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            float v = sum * (scg ? scg[co] : 1.f) + (shg ? shg[co] : 0.f) * (rsg ? rsg[m] : 1.f);
            if (d.res_mode == EGR_RES_BEFORE_ACT) v += resg[(int64_t)s_roff[row] + co];
            v = egr_act(v, d.act);
            if (d.res_mode == EGR_RES_AFTER_ACT) v += resg[(int64_t)s_roff[row] + co];
            if (rmg && !rmg[m]) v = 0.f;
            amx = fmaxf(amx, fabsf(v));
            yg[(int64_t)yo + (d.out_nchw ? (int64_t)co * HoWo : (int64_t)co)] = v;
        }
        return;
    }

    if (d.out_nchw) {  // channel-major planes: lanes run along pixels (contiguous within a plane)
        for (int c = 0; c < BN; ++c) {
            int co = tn * BN + c;
            if (co >= d.cout) break;
            float sc = scg ? scg[co] : 1.f;
            float sh = shg ? shg[co] : 0.f;
            for (int row = tid; row < BM; row += NT) {
                int yo = s_yoff[row];
                if (yo < 0) continue;
                int m = tm * BM + row;
                float v = sC[row * P::CS + c] * sc + sh * (rsg ? rsg[m] : 1.f);
                if (d.res_mode == EGR_RES_BEFORE_ACT) v += resg[(int64_t)s_roff[row] + co];
                v = egr_act(v, d.act);
                if (d.res_mode == EGR_RES_AFTER_ACT) v += resg[(int64_t)s_roff[row] + co];
                if (rmg && !rmg[m]) v = 0.f;
                yg[(int64_t)yo + (int64_t)co * HoWo] = v;
            }
        }
        return;
    }

    constexpr int QPR = BN / 4;          // channel quads per tile row
    constexpr int RPI = NT / QPR;        // rows covered per iteration; a thread keeps one channel quad throughout
    static_assert(NT % QPR == 0 && BM % RPI == 0, "epilogue mapping");
    const int cq = tid % QPR, row0 = tid / QPR;
    const int co = tn * BN + cq * 4;
    if (co >= d.cout) return;
    const bool vec = a.vec_ok && co + 3 < d.cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (co + e < d.cout) {
            if (scg) sc[e] = scg[co + e];
            if (shg) sh[e] = shg[co + e];
        }

    if (d.res_mode == EGR_RES_UP2_BEFORE_ACT) {
        // the residual is interpolated here instead of being materialised by a separate pass (the FPN top-down path): 4 gathers
        // per output, parameters from the row table.  Straight-line code - the gathers of padded rows read offset 0, only the
        // store is predicated (a branch around the loads would make every row wait for its own gathers)
        const int wl = d.wo >> 1;
        const int dxo = d.ldr, dyo = wl * d.ldr;
        const float* const rb = resg + co;
#pragma unroll 8
        for (int it = 0; it < BM / RPI; ++it) {
            const int row = row0 + it * RPI;
            const int yo = s_yoff[row];
            const int ro = s_roff[row];
            const unsigned bx = (unsigned)s_yoff[P::COL_LX * BM + row], by = (unsigned)s_yoff[P::COL_LY * BM + row];
            const float lx1 = __uint_as_float(bx & 0x7fffffffu), ly1 = __uint_as_float(by & 0x7fffffffu);
            const float lx0 = 1.f - lx1, ly0 = 1.f - ly1;
            const int ox = (bx >> 31) ? 0 : dxo, oy = (by >> 31) ? 0 : dyo;
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(rb + ro);
            const f32x4 v01 = *reinterpret_cast<const f32x4*>(rb + ro + ox);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(rb + ro + oy);
            const f32x4 v11 = *reinterpret_cast<const f32x4*>(rb + ro + oy + ox);
            f32x4 v = *reinterpret_cast<const f32x4*>(&sC[row * P::CS + cq * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e] * sc[e] + sh[e];
                t += ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
                v[e] = (d.act == EGR_ACT_RELU) ? (t > 0.f ? t : 0.f) : t;
            }
            if (yo >= 0) {
                *reinterpret_cast<f32x4*>(yg + (int64_t)yo + co) = v;
                track4(v);
            }
        }
        return;
    }

    if (a.mask) {  // dx = (acc [+ res]) * [mask > 0]  (the host guarantees the 16-byte path, no scale/shift/activation)
        const float* const maskg = a.mask + grp * d.gy;
#pragma unroll
        for (int it = 0; it < BM / RPI; ++it) {
            const int row = row0 + it * RPI;
            const int yo = s_yoff[row];
            if (yo < 0) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(&sC[row * P::CS + cq * 4]);
            f32x4 mk;
            if constexpr (MPRE) {
                if (d.res_mode) v += rpre[it];
                mk = mpre[it];
            } else {
                if (d.res_mode) v += *reinterpret_cast<const f32x4*>(resg + (int64_t)s_roff[row] + co);
                mk = *reinterpret_cast<const f32x4*>(maskg + (int64_t)yo + co);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
            track4(v);
            *reinterpret_cast<f32x4*>(yg + (int64_t)yo + co) = v;
        }
        return;
    }

    if (vec && !rsg && !rmg) {
        // fast path (every conv of the CNN stages): 16-byte accesses, activation / residual mode resolved at compile
        // time so the row loop is branch-free straight-line code
        if (a.bn_part) {
            // The raw output of a conv that feeds a train-mode BatchNorm (host: no activation / residual, cout % BN == 0, no split-K):
            // the statistics pass over y (egr_bn_stats_f32's first kernel: one read of the tensor) is folded in here - this tile's
            // per-channel sums in double and extremes, reduced over the threads that share a channel quad through the staging area.
            double ds[4] = {0.0, 0.0, 0.0, 0.0}, dq[4] = {0.0, 0.0, 0.0, 0.0};
            float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll 8
            for (int it = 0; it < BM / RPI; ++it) {
                const int row = row0 + it * RPI;
                const int yo = s_yoff[row];
                if (yo < 0) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(&sC[row * P::CS + cq * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] * sc[e] + sh[e];
                    ds[e] += (double)v[e];
                    dq[e] += (double)v[e] * (double)v[e];
                    lo[e] = fminf(lo[e], v[e]);
                    hi[e] = fmaxf(hi[e], v[e]);
                }
                track4(v);
                *reinterpret_cast<f32x4*>(yg + (int64_t)yo + co) = v;
            }
            __syncthreads();                               // the staged tile has been read by everybody: its words are free
            double* const rd = reinterpret_cast<double*>(sC);
            float* const rf = reinterpret_cast<float*>(sC);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rd[tid * 8 + e] = ds[e];
                rd[tid * 8 + 4 + e] = dq[e];
            }
            __syncthreads();
            if (row0 == 0)
                for (int k = 1; k < RPI; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ds[e] += rd[(k * QPR + cq) * 8 + e];
                        dq[e] += rd[(k * QPR + cq) * 8 + 4 + e];
                    }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rf[tid * 8 + e] = lo[e];
                rf[tid * 8 + 4 + e] = hi[e];
            }
            __syncthreads();
            if (row0 == 0) {
                for (int k = 1; k < RPI; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = fminf(lo[e], rf[(k * QPR + cq) * 8 + e]);
                        hi[e] = fmaxf(hi[e], rf[(k * QPR + cq) * 8 + 4 + e]);
                    }
                const int64_t slab = ((int64_t)grp * a.tilesM + tm) * 2 * d.cout + co;
                float* const mm = reinterpret_cast<float*>(a.bn_part + (int64_t)d.groups * a.tilesM * 2 * d.cout);   // the extremes sit behind the sums
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a.bn_part[slab + e] = ds[e];
                    a.bn_part[slab + d.cout + e] = dq[e];
                    mm[slab + e] = lo[e];
                    mm[slab + d.cout + e] = hi[e];
                }
            }
            __syncthreads();                               // (persistent launches stage the next tile into the same words)
            return;
        }
        auto rows_fast = [&](auto act_tag, auto res_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            constexpr int RES = decltype(res_tag)::value;
#pragma unroll
            for (int it = 0; it < BM / RPI; ++it) {
                const int row = row0 + it * RPI;
                const int yo = s_yoff[row];
                if (yo < 0) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(&sC[row * P::CS + cq * 4]);
                f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                if constexpr (RES != EGR_RES_NONE) {
                    if constexpr (RPRE) rr = rpre[it];
                    else rr = *reinterpret_cast<const f32x4*>(resg + (int64_t)s_roff[row] + co);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[e] * sc[e] + sh[e];
                    if constexpr (RES == EGR_RES_BEFORE_ACT) t += rr[e];
                    if constexpr (ACT == EGR_ACT_RELU) t = t > 0.f ? t : 0.f;
                    if constexpr (ACT == EGR_ACT_GELU) t = 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f));
                    if constexpr (RES == EGR_RES_AFTER_ACT) t += rr[e];
                    v[e] = t;
                }
                track4(v);
#ifdef X6_EXP_NOSTORE
                if (v[0] == 12345.678f)
#endif
                *reinterpret_cast<f32x4*>(yg + (int64_t)yo + co) = v;
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        const int key = d.act * 3 + d.res_mode;
        switch (key) {
            case 0: rows_fast(I0{}, I0{}); break;
            case 1: rows_fast(I0{}, I1{}); break;
            case 2: rows_fast(I0{}, I2{}); break;
            case 3: rows_fast(I1{}, I0{}); break;
            case 4: rows_fast(I1{}, I1{}); break;
            case 5: rows_fast(I1{}, I2{}); break;
            case 6: rows_fast(I2{}, I0{}); break;
            case 7: rows_fast(I2{}, I1{}); break;
            default: rows_fast(I2{}, I2{}); break;
        }
        stamp(5);  // stores issued
        return;
    }

    // generic path: ragged channel counts (cout = 15, 3, 48 ...), unaligned outputs, per-row scale / mask
    for (int it = 0; it < BM / RPI; ++it) {
        const int row = row0 + it * RPI;
        const int yo = s_yoff[row];
        if (yo < 0) continue;
        const int m = tm * BM + row;
        f32x4 v = *reinterpret_cast<const f32x4*>(&sC[row * P::CS + cq * 4]);
        const float rs = rsg ? rsg[m] : 1.f;
        const bool keep = !(rmg && !rmg[m]);
        f32x4 rr = {0.f, 0.f, 0.f, 0.f};
        if (d.res_mode) {
            const float* rp = resg + (int64_t)s_roff[row] + co;
            if (vec) rr = *reinterpret_cast<const f32x4*>(rp);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < d.cout) rr[e] = rp[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v[e] * sc[e] + sh[e] * rs;
            if (d.res_mode == EGR_RES_BEFORE_ACT) t += rr[e];
            t = egr_act(t, d.act);
            if (d.res_mode == EGR_RES_AFTER_ACT) t += rr[e];
            v[e] = keep ? t : 0.f;
            if (co + e < d.cout) amx = fmaxf(amx, fabsf(v[e]));
        }
        float* yp = yg + (int64_t)yo + co;
        if (vec) *reinterpret_cast<f32x4*>(yp) = v;
        else
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (co + e < d.cout) yp[e] = v[e];
    }
    stamp(5);  // stores issued
    };

    if constexpr (TAP2) {
        // ---- split-bf16, 3x3 / STRIDE 2 / pad 1, taps shared by input-pixel parity class (forward).  Output pixel (y, x), tap (kh, kw)
        // reads input (2y + kh - 1, 2x + kw - 1): kh = 1 -> even rows (index y), kh = 0 / 2 -> odd rows (index y - 1 / y); the same for
        // columns.  So the nine taps fall into four classes by the parities of the input pixel they read - (even, even): 1 tap,
        // (even, odd): 2, (odd, even): 2, (odd, odd): 4 - and inside a class they are unit shifts of ONE plane of (rows + 1) x (wo + 1)
        // input pixels.  Per 16-channel chunk four such planes are fetched, split and staged (one barrier each) instead of nine
        // 128 x 16 blocks with nine barriers: activation work and barriers / 2.25.  Loads run two classes ahead (the single-tap
        // class is too short to cover a memory latency), the conversion one class ahead, weights one tap ahead straight into the
        // B operand, A fragments refilled plane by plane - as in the stride-1 branch below.
        static_assert(X6 && !PERSIST && BM == 128, "stride-2 tap-sharing tile");
        constexpr int HPM = 192;                                 // pixels per class plane: (128 / wo + 1) x (wo + 1) for wo in {8, 16, 32}: <= 165
        constexpr int T2_PLANE = HPM * 32, T2_HBUF = NPL * T2_PLANE;
        static_assert(2 * T2_HBUF <= P::ROWOFF * 4, "class planes fit under the staged tile");
        constexpr int NFB = BN / 32;
        constexpr int NUH = (HPM * 4 + NT - 1) / NT;             // staging units (4 channels of one pixel) per thread and class
        uint8_t* const lb = reinterpret_cast<uint8_t*>(lds);
        const int wo = d.wo, WP = wo + 1;
        const int NI = HoWo >= BM ? 1 : BM / HoWo, RTI = HoWo >= BM ? BM / wo : d.ho;
        const int HPI = (RTI + 1) * WP, HP = NI * HPI, PPI = RTI * wo;
        const int m0 = tm * BM;
        int n0, pix0;
        if (a.howo_shift >= 0) { n0 = m0 >> a.howo_shift; pix0 = m0 & (HoWo - 1); }
        else { n0 = fdiv(m0, a.dHoWo); pix0 = m0 - n0 * HoWo; }
        const int y0 = (a.wo_shift >= 0) ? (pix0 >> a.wo_shift) : fdiv(pix0, a.dWo);
        const int xbase = a.x_plain ? n0 * (int)d.xmap.stride_inner : (int)fmap(d.xmap, a.dXin, n0);
        const int abias = (d.w + 1) * d.ldx * 4;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(reinterpret_cast<const char*>(xg) - abias), 0, 0x80000000u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(reinterpret_cast<const uint8_t*>(a.w) + (int64_t)grp * d.gw * 2), 0, 0x80000000u, 0x00020000);
        const int so_tile = __builtin_amdgcn_readfirstlane((xbase + 2 * y0 * d.w * d.ldx) * 4);
        // class q = 2 * (odd row) + (odd column); plane pixel (pr, pc) = input (2 (y0 + pr) - odd row, 2 pc - odd column)
        int hvo[4][NUH];
#pragma unroll
        for (int i = 0; i < NUH; ++i) {
            const int u = tid + NT * i, hp = u >> 2, seg = u & 3;
            const int il = hp / HPI, hq = hp - il * HPI;
            const int pr = hq / WP, pc = hq - pr * WP;
            const int ioff = (il == 0) ? 0 : ((a.x_plain ? (n0 + il) * (int)d.xmap.stride_inner : (int)fmap(d.xmap, a.dXin, n0 + il)) - xbase);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int dy = 2 * pr - (q >> 1), ix = 2 * pc - (q & 1);
                const bool ok = hp < HP && (unsigned)(2 * y0 + dy) < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
                hvo[q][i] = ok ? (ioff + (dy * d.w + ix) * d.ldx) * 4 + seg * 16 + abias : (int)0x80000000;
            }
        }
        int abase[FM], bvo[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int ml = wm * TM + i * 32 + l31, il = ml / PPI, mq = ml - il * PPI, r = mq / wo, c = mq - r * wo;
            abase[i] = (il * HPI + r * WP + c) * 32 + half * 16;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) bvo[j] = ((tn * NFB + wn * FN + j) * a.ktiles * 2 * NPL) * 1024 + lane * 16;
        const int NC = a.cblocks * 2;            // 16-channel chunks
        // the nine taps in class order: weight tap index, class, row / column shift inside the class plane
        constexpr int TID[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
        constexpr int TCL[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
        constexpr int TDR[9] = {0, 0, 0, 0, 1, 0, 0, 1, 1};
        constexpr int TDC[9] = {0, 0, 1, 0, 0, 0, 1, 0, 1};
        u32x4 xr[2][NUH];
        auto load_class = [&](int ck, auto q_tag) {    // plane of class q of chunk ck -> register set q & 1
            constexpr int Q = decltype(q_tag)::value;
            const int so = so_tile + ck * 64;
#pragma unroll
            for (int i = 0; i < NUH; ++i) xr[Q & 1][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, hvo[Q][i], so, 0);
        };
        unsigned ch_[NUH][2], cm_[NUH][2], cl_[NUH][2];     // converted pairs per plane: [unit][pair]
        auto cslice = [&](auto set_tag, int base, int k) {
            constexpr int SET = decltype(set_tag)::value;
            const int u = k / 3, q = k % 3;
            if (q < 2) {
                const float v0 = __uint_as_float(xr[SET][u][2 * q]), v1 = __uint_as_float(xr[SET][u][2 * q + 1]);
                split_pair<NPL>(v0, v1, sa, ch_[u][q], cm_[u][q], cl_[u][q]);
            } else if (tid + NT * u < 4 * HP) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                uint8_t* dst = lb + base + (tid + NT * u) * 8;
                *reinterpret_cast<u32x2*>(dst) = u32x2{ch_[u][0], ch_[u][1]};
                *reinterpret_cast<u32x2*>(dst + T2_PLANE) = u32x2{cm_[u][0], cm_[u][1]};
                if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(dst + 2 * T2_PLANE) = u32x2{cl_[u][0], cl_[u][1]};
            }
        };
        constexpr int NSL = 3 * NUH;
        u32x4 af[FM][NPL], bf[2][FN][NPL];
        auto read_a = [&](int base, int t9, int pl) {       // fragment plane pl of the t9-th tap (class order)
            const int to = (TDR[t9] * WP + TDC[t9]) * 32;
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i][pl] = *reinterpret_cast<const u32x4*>(lb + base + pl * T2_PLANE + abase[i] + to);
        };
        auto read_a_all = [&](int base, int t9) {           // every plane, in the order of first use
            if constexpr (NPL == 3) { read_a(base, t9, 2); read_a(base, t9, 0); read_a(base, t9, 1); }
            else { read_a(base, t9, 1); read_a(base, t9, 0); }
        };
        auto load_b = [&](int ck, int t9, int par) {
            const int so = (((ck >> 1) * 9 + TID[t9]) * 2 * NPL + (ck & 1) * NPL) * 1024;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    bf[par][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rb, bvo[j] + pl * 1024, so, 0);
        };
        using Q0 = std::integral_constant<int, 0>;
        using Q1 = std::integral_constant<int, 1>;
        using Q2 = std::integral_constant<int, 2>;
        using Q3 = std::integral_constant<int, 3>;
        // one chunk = four classes.  Class q multiplies out of buffer q & 1 while class q + 1's plane (register set (q + 1) & 1, loaded
        // during class q - 1) is converted into buffer (q + 1) & 1 and class q + 2's plane is requested into set q & 1.
        // PAR: B register set of the chunk's first tap; LAST: no chunk follows.
        // one tap (compile-time position T9 in class order) of class Q
        auto tap_step = [&](auto par_tag, auto last_tag, auto q_tag, auto t9_tag, int ck, int& n, int& done) {
            constexpr int PAR = decltype(par_tag)::value, Q = decltype(q_tag)::value, T9 = decltype(t9_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr int PA[6] = {NPL == 3 ? 2 : 1, 0, NPL == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0};
            constexpr int NMT = NPR * FM * FN;
            constexpr int FIRST[4] = {0, 1, 3, 5}, COUNT[4] = {1, 2, 2, 4};
            constexpr bool conv = !(LAST && Q == 3);              // a next class exists
            constexpr int cur = (Q & 1) * T2_HBUF, nxt = T2_HBUF - cur;
            constexpr int nm = COUNT[Q] * NMT;
            constexpr int pc = (PAR + T9) & 1, pn = pc ^ 1;
            constexpr bool more_in_class = T9 + 1 < FIRST[Q] + COUNT[Q];
            if constexpr (T9 + 1 < 9) load_b(ck, T9 + 1, pn);
            else if constexpr (!LAST) load_b(ck + 1, 0, pn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NPR; ++t) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j, ++n) {
                        acc[i][j] = mfma_split<NPL>(af[i][PA[t]], bf[pc][j][PB[t]], acc[i][j]);
                        if constexpr (conv) {
                            // (several slices behind one MFMA when a class is shorter than the conversion: one tap of the fp16 scheme on a
                            // 64-wide tile is 6 MFMAs against 9 slices)
                            const int upto = ((n + 1) * NSL + nm - 1) / nm;
#pragma unroll
                            for (int k = 0; k < NSL; ++k)
                                if (k >= done && k < upto) cslice(std::integral_constant<int, (Q + 1) & 1>{}, nxt, k);
                            done = upto > done ? upto : done;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                if constexpr (more_in_class) {                    // next tap of the same class: same buffer
                    if (split_free_a(NPL, t) >= 0) {
                        read_a(cur, T9 + 1, split_free_a(NPL, t));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        // one class of a chunk: request the plane of the class after next, multiply this class's taps (converting the next class's
        // plane behind them), hand over
        auto cls = [&](auto par_tag, auto last_tag, auto q_tag, int ck) {
            constexpr int Q = decltype(q_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr int FIRST[4] = {0, 1, 3, 5};
            constexpr bool conv = !(LAST && Q == 3);
            constexpr int nxt = T2_HBUF - (Q & 1) * T2_HBUF;
            if constexpr (Q == 0) load_class(ck, Q2{});
            if constexpr (Q == 1) load_class(ck, Q3{});
            if constexpr (Q == 2 && !LAST) load_class(ck + 1, Q0{});
            if constexpr (Q == 3 && !LAST) load_class(ck + 1, Q1{});
            int n = 0, done = 0;
            using I = std::integral_constant<int, FIRST[Q]>;
            tap_step(par_tag, last_tag, q_tag, I{}, ck, n, done);
            if constexpr (Q >= 1) tap_step(par_tag, last_tag, q_tag, std::integral_constant<int, FIRST[Q] + 1>{}, ck, n, done);
            if constexpr (Q == 3) {
                tap_step(par_tag, last_tag, q_tag, std::integral_constant<int, FIRST[Q] + 2>{}, ck, n, done);
                tap_step(par_tag, last_tag, q_tag, std::integral_constant<int, FIRST[Q] + 3>{}, ck, n, done);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (conv) {
                constexpr int t9n = (Q == 3) ? 0 : FIRST[(Q + 1) & 3];
                read_a_all(nxt, t9n);
            }
        };
        // PAR: B register set of the chunk's first tap (nine taps flip it, so chunks come in pairs); LAST: no chunk follows
        auto chunk = [&](auto par_tag, int ck, auto last_tag) {
            cls(par_tag, last_tag, Q0{}, ck);
            cls(par_tag, last_tag, Q1{}, ck);
            cls(par_tag, last_tag, Q2{}, ck);
            cls(par_tag, last_tag, Q3{}, ck);
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        using TT = std::true_type;
        using FF = std::false_type;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // row table visible
        stamp(1);
        load_class(0, Q0{});
        load_class(0, Q1{});
        load_b(0, 0, 0);
#pragma unroll
        for (int k = 0; k < NSL; ++k) cslice(Q0{}, 0, k);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_a_all(0, 0);
        stamp(2);
        int ck = 0;
        for (; ck + 2 < NC; ck += 2) {           // NC is even; nine taps flip the B register parity, so chunks come in pairs
            chunk(T0{}, ck, FF{});
            chunk(T1{}, ck + 1, FF{});
        }
        chunk(T0{}, ck, FF{});
        chunk(T1{}, ck + 1, TT{});
        epilogue(tm, tn, s_yoff, s_roff, [] {});
        if (a.amax_out) amax_flush(a.amax_out, amx, (int)blockIdx.x + wave);
    } else if constexpr (TAP) {
        // ---- split-bf16, 3x3 / stride 1 / pad 1, taps SHARED (forward and data gradient).  In the generic split loop below every (tap, 16-channel)
        // stage fetches and splits its own BM x 16 activation block: each input value is loaded and split nine times, and every
        // stage ends in a barrier.  Here the tile is BM consecutive output pixels = BM / wo whole rows of one image (or whole small
        // images); for one 16-channel chunk the input rows around them (+ 1 pixel all around, zeros outside the image) are split ONCE
        // into pixel-major bf16 planes [pixel][16 channels] in LDS, and the nine taps are nine pixel-shifted ds_read_b128 windows
        // of those planes.  One barrier, one conversion and one activation fetch per 9 x 6 x FM x FN MFMAs; the weights go from
        // global memory (L2) straight into the B operand registers, one tap ahead.
        static_assert(X6 && !PERSIST && (BM == 64 || BM == 128 || BM == 256), "tap-sharing tile");
        constexpr int TAP_PLANE = tap_plane(BM), TAP_HBUF = tap_hbuf(BM, NPL);
        constexpr int NFB = BN / 32;
        constexpr int NUH = (tap_hpmax(BM) * 4 + NT - 1) / NT;   // halo staging units (4 channels of one pixel) per thread
        uint8_t* const lb = reinterpret_cast<uint8_t*>(lds);
        // tile geometry: RTI rows of NI images (one image's rows, or several whole small images)
        const int wo = d.wo, WP = wo + 2;
        const int NI = HoWo >= BM ? 1 : BM / HoWo, RTI = HoWo >= BM ? BM / wo : d.ho;
        const int HPI = (RTI + 2) * WP, HP = NI * HPI, PPI = RTI * wo;
        // the tile's first image and row (wave-uniform)
        const int m0 = tm * BM;
        int n0, pix0;
        if (a.howo_shift >= 0) { n0 = m0 >> a.howo_shift; pix0 = m0 & (HoWo - 1); }
        else { n0 = fdiv(m0, a.dHoWo); pix0 = m0 - n0 * HoWo; }
        const int y0 = (a.wo_shift >= 0) ? (pix0 >> a.wo_shift) : fdiv(pix0, a.dWo);
        const int xbase = a.x_plain ? n0 * (int)d.xmap.stride_inner : (int)fmap(d.xmap, a.dXin, n0);
        const int abias = (d.w + 1) * d.ldx * 4;                 // the window starts one row + one pixel early: halo offsets >= 0
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(reinterpret_cast<const char*>(xg) - abias), 0, 0x80000000u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(reinterpret_cast<const uint8_t*>(a.w) + (int64_t)grp * d.gw * 2), 0, 0x80000000u, 0x00020000);
        const int so_tile = __builtin_amdgcn_readfirstlane((xbase + y0 * d.w * d.ldx) * 4);
        int hvo[NUH];
#pragma unroll
        for (int i = 0; i < NUH; ++i) {
            const int u = tid + NT * i, hp = u >> 2, seg = u & 3;
            const int il = hp / HPI, hq = hp - il * HPI;
            const int hr = hq / WP, hc = hq - hr * WP;
            const bool ok = hp < HP && (unsigned)(y0 + hr - 1) < (unsigned)d.h && (unsigned)(hc - 1) < (unsigned)d.w;
            const int ioff = (il == 0) ? 0 : ((a.x_plain ? (n0 + il) * (int)d.xmap.stride_inner : (int)fmap(d.xmap, a.dXin, n0 + il)) - xbase);
            hvo[i] = ok ? (ioff + ((hr - 1) * d.w + (hc - 1)) * d.ldx) * 4 + seg * 16 + abias : (int)0x80000000;
        }
        int abase[FM], bvo[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int ml = wm * TM + i * 32 + l31, il = ml / PPI, mq = ml - il * PPI, r = mq / wo, c = mq - r * wo;
            abase[i] = (il * HPI + r * WP + c) * 32 + half * 16;
        }
#pragma unroll
#ifdef TAP_EXP_BUNIFORM
        for (int j = 0; j < FN; ++j) bvo[j] = ((tn * NFB + wn * FN + j) * a.ktiles * 2 * NPL) * 1024;
#else
        for (int j = 0; j < FN; ++j) bvo[j] = ((tn * NFB + wn * FN + j) * a.ktiles * 2 * NPL) * 1024 + lane * 16;
#endif
        const int NC = a.cblocks * 2;            // 16-channel chunks
        u32x4 xr[NUH];
        auto load_halo = [&](int ck) {           // chunk ck = (32-channel block ck >> 1, half ck & 1)
            const int so = so_tile + ck * 64;
#pragma unroll
            for (int i = 0; i < NUH; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, hvo[i], so, 0);
        };
        // slice k of the conversion: unit k / 3; 0 / 1 = the planes of the unit's first / second pair, 2 = the 8-byte writes (one per plane)
        // (fp16 scheme: two slices per unit - the unit's four values converted as two interleaved dependency chains, then the writes)
        unsigned ch_[NUH][2], cm_[NUH][2], cl_[NUH][2];     // converted pairs per plane: [unit][pair]
        constexpr int SPU = NPL == 3 ? 3 : 2;               // slices per unit
        auto cslice = [&](int base, int k) {
            const int u = k / SPU, q = k % SPU;
#ifdef TAP_EXP_NOCVT
            if (q < SPU - 1) return;
#endif
            if (NPL == 2 && q == 0) {
                split4_f16(__uint_as_float(xr[u][0]), __uint_as_float(xr[u][1]), __uint_as_float(xr[u][2]), __uint_as_float(xr[u][3]), sa,
                           ch_[u][0], cm_[u][0], ch_[u][1], cm_[u][1]);
            } else if (NPL == 3 && q < 2) {
                const float v0 = __uint_as_float(xr[u][2 * q]), v1 = __uint_as_float(xr[u][2 * q + 1]);
                split_pair<NPL>(v0, v1, sa, ch_[u][q], cm_[u][q], cl_[u][q]);
            } else if (tid + NT * u < 4 * HP) {           // LDS byte (tid + 256 u) * 8 of the plane: pixel hp = unit >> 2, 8 bytes per unit
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                uint8_t* dst = lb + base + (tid + NT * u) * 8;
                *reinterpret_cast<u32x2*>(dst) = u32x2{ch_[u][0], ch_[u][1]};
                *reinterpret_cast<u32x2*>(dst + TAP_PLANE) = u32x2{cm_[u][0], cm_[u][1]};
                if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(dst + 2 * TAP_PLANE) = u32x2{cl_[u][0], cl_[u][1]};
            }
        };
        constexpr int NSL = SPU * NUH;
        // A fragments: ONE register set, refilled plane by plane behind the last product that uses the plane (lo after product 0,
        // mid after product 3, hi after product 5) - the next tap's LDS reads overlap this tap's remaining MFMAs at no register cost.
        // B fragments come from global memory (L2 latency > one tap): two sets, one tap ahead.  (Wave layouts 1 x 4 / 2 x 2 instead of
        // 2 x 2 / 4 x 1, i.e. fewer redundant weight fetches per workgroup, measured the same within 1 %.)
#ifndef TAP_BAHEAD_H2
#define TAP_BAHEAD_H2 2
#endif
        // Taps the weight loads run ahead (1: two register sets by parity; 2: three sets, set = tap % 3).  Vector-memory loads return in
        // order (one vmcnt): a weight load issued BEHIND the next chunk's halo loads (HBM latency) cannot be consumed before they have
        // landed.  With the fp16 scheme's short taps (12 MFMAs) one tap of distance made every chunk wait for its successor's halo;
        // two taps ahead + the halo issued behind tap 0's weight loads + the conversion in the chunk's second half give the halo
        // three taps to arrive.
        constexpr int BAHEAD = NPL == 2 ? TAP_BAHEAD_H2 : 1;
        u32x4 af[FM][NPL], bf[BAHEAD + 1][FN][NPL];
        auto read_a = [&](int base, int tap, int pl) {
            // (data gradient: tap (kh, kw) reads dy at (y + 1 - kh, x + 1 - kw) - the mirrored window)
            const int to = d.transposed ? ((2 - tap / 3) * WP + (2 - tap % 3)) * 32 : ((tap / 3) * WP + (tap % 3)) * 32;
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i][pl] = *reinterpret_cast<const u32x4*>(lb + base + pl * TAP_PLANE + abase[i] + to);
        };
        auto read_a_all = [&](int base, int tap) {          // every plane, in the order of first use
            if constexpr (NPL == 3) { read_a(base, tap, 2); read_a(base, tap, 0); read_a(base, tap, 1); }
            else { read_a(base, tap, 1); read_a(base, tap, 0); }
        };
        auto load_b = [&](int ck, int tap, int par) {
            const int so = (((ck >> 1) * 9 + tap) * 2 * NPL + (ck & 1) * NPL) * 1024;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    bf[par][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rb, bvo[j] + pl * 1024, so, 0);
        };
        // one chunk: nine taps out of halo buffer `cur`; the next chunk's halo is converted into `nxt` behind the taps 2 .. 7.
        // PAR: B register set of tap 0 (nine taps flip it, so chunks alternate)
        auto chunk = [&](auto par_tag, int ck, int cur, int nxt, auto conv_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            // (TAP_EXP_*: elimination builds for tools/build_variant.py - timing only, the results are wrong)
#ifdef TAP_EXP_NOCONV
            constexpr bool conv = false;
#else
            constexpr bool conv = decltype(conv_tag)::value;
#endif
#ifndef TAP_EXP_NOHALO
            if constexpr (conv && BAHEAD == 1) load_halo(ck + 1);
#endif
            constexpr int PA[6] = {NPL == 3 ? 2 : 1, 0, NPL == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0};
            constexpr int NMT = NPR * FM * FN;               // MFMAs per tap
            constexpr int W0 = (BAHEAD == 1 ? 2 : BAHEAD + 1) * NMT, W1 = (BAHEAD == 1 ? 8 : 9) * NMT;   // conversion window (in MFMAs of the chunk)
            int n = 0, done = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int pc = BAHEAD == 1 ? ((PAR + tap) & 1) : tap % 3, pn = BAHEAD == 1 ? (pc ^ 1) : (tap + 2) % 3;
#ifndef TAP_EXP_NOB
                if (tap + BAHEAD < 9) load_b(ck, tap + BAHEAD, pn);
                else if (conv) load_b(ck + 1, tap + BAHEAD - 9, pn);
#endif
#ifndef TAP_EXP_NOHALO
                if constexpr (conv && BAHEAD > 1) { if (tap == 0) load_halo(ck + 1); }
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NPR; ++t) {
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j, ++n) {
                            acc[i][j] = mfma_split<NPL>(af[i][PA[t]], bf[pc][j][PB[t]], acc[i][j]);
                            if constexpr (conv) {
                                const int upto = (n + 1 <= W0) ? 0 : (n + 1 >= W1 ? NSL : ((n + 1 - W0) * NSL + (W1 - W0) - 1) / (W1 - W0));
                                static_assert(NSL <= W1 - W0, "at most one conversion slice behind an MFMA");
                                if (done < upto) { cslice(nxt, done); ++done; }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
#ifndef TAP_EXP_NOA
                    if (tap + 1 < 9 && split_free_a(NPL, t) >= 0) {
#else
                    if (false) {
#endif
                        read_a(cur, tap + 1, split_free_a(NPL, t));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (conv) {
                read_a_all(nxt, 0);
            }
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // row table visible
        stamp(1);
        load_halo(0);
        load_b(0, 0, 0);
        if constexpr (BAHEAD == 2) load_b(0, 1, 1);
#pragma unroll
        for (int k = 0; k < NSL; ++k) cslice(0, k);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_a_all(0, 0);
        stamp(2);
        int ck = 0;
        for (; ck + 2 < NC; ck += 2) {           // NC is even: chunk pairs, buffers 0 / 1, register parity 0 / 1
            chunk(T0{}, ck, 0, TAP_HBUF, std::true_type{});
            chunk(T1{}, ck + 1, TAP_HBUF, 0, std::true_type{});
        }
        chunk(T0{}, ck, 0, TAP_HBUF, std::true_type{});
        chunk(T1{}, ck + 1, TAP_HBUF, 0, std::false_type{});
        epilogue(tm, tn, s_yoff, s_roff, [] {});
        if (a.amax_out) amax_flush(a.amax_out, amx, (int)blockIdx.x + wave);
    } else if constexpr (!X6) {
        const float* wrow[IB];
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            int r = i * 32 + rsub;
            int co = tn * BN + r;
            wrow[i] = (co < a.Npad) ? wg + (int64_t)co * a.K + (pseg ^ ((r >> 1) & 7)) * 4 : nullptr;
        }
        const float* arow[IA];
        unsigned amask[IA];

        // one 1-KiB DMA piece (8 rows x 128 B) of a stage: pieces [0, IA) are A rows, [IA, IA+IB) weight rows.
        // The stage buffer is a compile-time constant so LDS addresses fold into instruction immediates.
        const uint32_t zlo = (uint32_t)(uint64_t)egr_zero16, zhi = (uint32_t)((uint64_t)egr_zero16 >> 32);
        auto issue_piece = [&](int kt, KPos kp, auto buf_tag, int piece) {
            constexpr int BUF = decltype(buf_tag)::value;
            float* sA = lds + BUF * P::TILE;
            if (piece < IA) {
                const int tap = kp.kh * d.kw + kp.kw;
                // wave-uniform, 32-bit.  Transposed mode: th - kh = stride*(bh - kh/stride) + (th%stride - kh%stride), and the mask
                // bit is set only where the remainders agree, so the source pixel is (bh - kh/stride, bw - kw/stride).
                const int toff = d.transposed ? -((kp.kh / d.stride) * d.w + (kp.kw / d.stride)) * d.ldx + kp.cb * BK
                                              : (kp.kh * d.w + kp.kw) * d.ldx + kp.cb * BK;
                const uint64_t pa = (uint64_t)(arow[piece] + toff);
                const bool ok = (amask[piece] >> tap) & 1u;
                // select between the pixel row and the zero buffer with two 32-bit v_cndmask (no branch, no 64-bit logic)
                const uint32_t lo = ok ? (uint32_t)pa : zlo, hi = ok ? (uint32_t)(pa >> 32) : zhi;
                glds16(reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo), sA + (piece * 32 + wave * 8) * BK);
            } else {
                const int i = piece - IA;
                const float* p = wrow[i] ? wrow[i] + (kp.cb * a.taps + kp.kh * d.kw + kp.kw) * BK : egr_zero16;   // chunk index in K
                glds16(p, sA + BM * BK + (i * 32 + wave * 8) * BK);
            }
        };

        // fragment read offsets (floats) inside a stage for the four k-groups of a chunk, computed once:
        // row * 32 + ((q ^ swz(row)) * 4) with q = 2*g + half
        int afo[FM][4], bfo[FN][4];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int r = wm * TM + i * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) afo[i][g] = r * BK + (((2 * g + half) ^ ((r >> 1) & 7)) << 2);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int r = wn * TN + j * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) bfo[j][g] = BM * BK + r * BK + (((2 * g + half) ^ ((r >> 1) & 7)) << 2);
        }

        constexpr int NPIECE = IA + IB;
        constexpr int DMA_EVERY = (FM * FN >= 4) ? 2 : 1;  // big tiles: one DMA piece behind every 2nd MFMA (measured +2.5 %)
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;

        // multiply one staged chunk; when ISSUE, the next chunk's DMA pieces go out one per MFMA behind the first
        // matrix instructions (pinned with sched_barrier), so their issue slots hide under the 64-cycle MFMAs
        auto chunk = [&](int kt, auto buf_tag, KPos kp_next, auto issue_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
            constexpr bool ISSUE = decltype(issue_tag)::value;
            using NB = std::integral_constant<int, BUF ^ 1>;
            const float* st = lds + BUF * P::TILE;
            f32x4 av[2][FM], bv[2][FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(&st[afo[i][0]]);
#pragma unroll
            for (int j = 0; j < FN; ++j) bv[0][j] = *reinterpret_cast<const f32x4*>(&st[bfo[j][0]]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cur = g & 1, nxt = cur ^ 1;
                if (g < 3) {  // fragments of the next k-group are fetched under this group's MFMAs
#pragma unroll
                    for (int i = 0; i < FM; ++i) av[nxt][i] = *reinterpret_cast<const f32x4*>(&st[afo[i][g + 1]]);
#pragma unroll
                    for (int j = 0; j < FN; ++j) bv[nxt][j] = *reinterpret_cast<const f32x4*>(&st[bfo[j][g + 1]]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i][t], bv[cur][j][t], acc[i][j], 0, 0, 0);
                            if constexpr (ISSUE) {
                                const int n = ((g * 4 + t) * FM + i) * FN + j;
                                if ((n % DMA_EVERY) == 0 && (n / DMA_EVERY) < NPIECE) {
                                    issue_piece(kt + 1, kp_next, NB{}, n / DMA_EVERY);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
            }
        };

        KPos kp = kpos_of(kt0);
        if (kt0 < kt1) {  // weight pieces of the first chunk need no row geometry: their latency overlaps the decode
#pragma unroll
            for (int pc = IA; pc < NPIECE; ++pc) issue_piece(kt0, kp, B0{}, pc);
        }
        __syncthreads();  // row table visible (also drains the weight DMA; it had the whole decode to land)
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int r = i * 32 + rsub;
            arow[i] = xg + (s_xoff[r] + (pseg ^ ((r >> 1) & 7)) * 4);
            amask[i] = s_mask[r];
        }
        stamp(1);  // row decode done
        if (kt0 < kt1) {
#pragma unroll
            for (int pc = 0; pc < IA; ++pc) issue_piece(kt0, kp, B0{}, pc);
        }
        __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes stage 0
        stamp(2);  // first chunk landed
        int kt = kt0;
        for (; kt + 2 < kt1; kt += 2) {  // two chunks per trip: stage indices are compile-time constants
            KPos k1 = kpos_next(kp);
            chunk(kt, B0{}, k1, std::true_type{});
            __syncthreads();
            KPos k2 = kpos_next(k1);
            chunk(kt + 1, B1{}, k2, std::true_type{});
            __syncthreads();
            kp = k2;
        }
        if (kt + 1 < kt1) {  // two chunks left
            KPos k1 = kpos_next(kp);
            chunk(kt, B0{}, k1, std::true_type{});
            __syncthreads();
            chunk(kt + 1, B1{}, k1, std::false_type{});
            __syncthreads();
        } else if (kt < kt1) {  // one chunk left
            chunk(kt, B0{}, kp, std::false_type{});
            __syncthreads();
        }
        epilogue(tm, tn, s_yoff, s_roff, [] {});
        if (a.amax_out) amax_flush(a.amax_out, amx, (int)blockIdx.x + wave);
    } else {
        // ---- split-bf16 main loop: fp32 operands as exact sums of three bf16 (hi, mid, lo); the six partial products of order
        // <= 2 on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  The dropped ones are ~2^-26 relative, below fp32 rounding.
        // A stage is ONE k16 step: a chunk (tap x 32 channels) is two stages; stage s multiplies out of LDS buffer s % 2.
        //   A: a thread owns `unit` = (tile row, 8-channel group).  Its 32 bytes arrive by two buffer loads two stages ahead (a
        //      halo tap / padded row is an out-of-range offset: the load returns zeros and touches no memory), are split by VALU
        //      work pinned into the gaps behind the MFMAs one stage ahead, and written in fragment order (ds_write_b128 x 3);
        //   B: the weights were split once (egr_pack_w6_f32) into fragment order; a wave fetches its 1-KiB pieces into registers
        //      one stage ahead and writes them to LDS at the top of the next stage.
        // Everything in the loop is straight-line code with wave-uniform control (exact s_waitcnt vmcnt(N) from the compiler: a
        // load is only waited for when it is consumed), the stage hand-over is `s_waitcnt lgkmcnt(0); s_barrier` - loads stay in
        // flight across it - and the last chunk is peeled so that the accumulators never move between registers.
        constexpr int AU = 2 * BM;                    // staging units per stage
        constexpr int NU = (AU + NT - 1) / NT;        // per thread
        constexpr int NFA = BM / 32, NFB = BN / 32;
        constexpr int A_BYTES = NFA * NPL * 1024;
        constexpr int STB = P::TILE * 4;              // bytes per stage
        constexpr int NPB = NFB * NPL;                // weight pieces per stage
        constexpr int NBJ = (NPB + 3) / 4;            // per wave
        uint8_t* const lb = reinterpret_cast<uint8_t*>(lds);
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        // buffer resources (raw, 2 GiB window): A = the activations of this group shifted back by `abias` bytes so that every
        // row offset is non-negative; B = the weight image of this group
        // (a data-gradient launch walks its taps backwards: `tbias` keeps the scalar tap offset non-negative as well)
        const int abias = (d.kh * d.w + d.kw + 1) * d.ldx * 4;
        const int tbias = d.transposed ? (((d.kh - 1) / d.stride) * d.w + (d.kw - 1) / d.stride) * d.ldx * 4 : 0;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(reinterpret_cast<const char*>(xg) - abias - tbias), 0, 0x80000000u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(reinterpret_cast<const uint8_t*>(a.w) + (int64_t)grp * d.gw * 2), 0, 0x80000000u, 0x00020000);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // row table visible
        stamp(1);  // row decode done
        int uoff[NU], uinv[NU], uwo[NU], bvo[NBJ];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + NT * i, r = (u >> 1) % BM, g = u & 1;
            uwo[i] = ((r >> 5) * NPL) * 1024 + ((r & 31) + 32 * g) * 16;
        }
        // a tile's staging roles: row offsets / dead-tap masks from table `slot`, weight piece offsets of column tile tn_
        auto unit_setup = [&](int slot, int tn_) {
            const int* const t_xoff = tab0 + slot * P::TCOLS * BM + 2 * BM;
            const int* const t_mask = t_xoff + BM;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int u = tid + NT * i, r = (u >> 1) % BM, g = u & 1;
                const bool live = u < AU;
                uoff[i] = (live ? t_xoff[r] * 4 : 0) + g * 32 + abias;
                uinv[i] = live ? ~t_mask[r] : -1;     // bit t set: tap t of this row reads zeros
            }
#pragma unroll
            for (int j = 0; j < NBJ; ++j) {
                const int pc = (uw + 4 * j) % NPB;
                bvo[j] = ((tn_ * NFB + pc / NPL) * a.ktiles * 2 * NPL + pc % NPL) * 1024 + lane * 16;
            }
        };
        unit_setup(0, tn);
        // weight piece (uw + 4 j) mod NPB: when the pieces do not divide evenly over the four waves, some are fetched twice (the
        // same bytes to the same place) - cheaper than a wave-dependent branch, behind which the compiler waits for every load
        int lds_b[NBJ];
#pragma unroll
        for (int j = 0; j < NBJ; ++j) lds_b[j] = A_BYTES + ((uw + 4 * j) % NPB) * 1024 + lane * 16;
        u32x4 xr[2][NU][2], breg[NBJ];
        auto load_a = [&](KPos kp, int sidx, auto set_tag) {
            constexpr int SET = decltype(set_tag)::value;
            const int tap = kp.kh * d.kw + kp.kw;
            const int toff = ((d.transposed ? -((kp.kh / d.stride) * d.w + (kp.kw / d.stride)) * d.ldx + kp.cb * BK
                                            : (kp.kh * d.w + kp.kw) * d.ldx + kp.cb * BK) + sidx * 16) * 4 + tbias;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int dead = __builtin_amdgcn_sbfe(uinv[i], tap, 1);               // -1 when the tap is outside the image
                const int vo = (dead & (int)0x80000000) | uoff[i];                      // beyond the window: the load returns zeros
                xr[SET][i][0] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo, toff, 0);
                xr[SET][i][1] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo + 16, toff, 0);
            }
        };
        auto load_b = [&](KPos kp, int sidx) {
            const int soff = ((kp.cb * a.taps + kp.kh * d.kw + kp.kw) * 2 * NPL + sidx * NPL) * 1024;
#pragma unroll
            for (int j = 0; j < NBJ; ++j) breg[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, bvo[j], soff, 0);
        };
        auto write_b = [&](auto buf_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
            for (int j = 0; j < NBJ; ++j) *reinterpret_cast<u32x4*>(lb + BUF * STB + lds_b[j]) = breg[j];
        };
        // one slice of the split of a unit.  bf16 x 3: 0-3 hi halves + first residuals of pair k, 4-7 mid / lo halves, 8 the three
        // writes; fp16 x 2: 0-3 both planes of pair k (four instructions), 4 the two writes
        u32x4 sh_[NU], sm_[NU], sl_[NU];
        float ra_[NU][4], rb_[NU][4];
        constexpr int SPU = NPL == 3 ? 9 : 5;   // slices per unit
        auto slice = [&](auto set_tag, auto buf_tag, int k) {
            constexpr int SET = decltype(set_tag)::value;
            constexpr int BUF = decltype(buf_tag)::value;
            const int i = k / SPU, q = k % SPU;
            if (q < 4) {
                const float v0 = __uint_as_float(xr[SET][i][q >> 1][(q & 1) * 2]), v1 = __uint_as_float(xr[SET][i][q >> 1][(q & 1) * 2 + 1]);
                if constexpr (NPL == 3) {
                    sh_[i][q] = cvt_pk_bf16(v0, v1);
                    ra_[i][q] = v0 - bf16_lo_f32(sh_[i][q]);
                    rb_[i][q] = v1 - bf16_hi_f32(sh_[i][q]);
                } else {
                    unsigned h_, l_;
                    split2_f16(v0, v1, sa, h_, l_);
                    sh_[i][q] = h_;
                    sm_[i][q] = l_;
                }
            } else if (NPL == 3 && q < 8) {
                const int t = q - 4;
                sm_[i][t] = cvt_pk_bf16(ra_[i][t], rb_[i][t]);
                sl_[i][t] = cvt_pk_bf16(ra_[i][t] - bf16_lo_f32(sm_[i][t]), rb_[i][t] - bf16_hi_f32(sm_[i][t]));
            } else if (AU >= NT * (i + 1) || tid + NT * i < AU) {
                uint8_t* dst = lb + BUF * STB + uwo[i];
                *reinterpret_cast<u32x4*>(dst) = sh_[i];
                *reinterpret_cast<u32x4*>(dst + 1024) = sm_[i];
                if constexpr (NPL == 3) *reinterpret_cast<u32x4*>(dst + 2048) = sl_[i];
            }
        };
        constexpr int NS = SPU * NU;        // slices per stage
        constexpr int NM = NPR * FM * FN;   // MFMAs per stage
        // MFMAs in front of the first slice (its operands are the youngest loads but two).  Measured neutral-to-worse: starting the
        // slices behind MFMA 2 or 8, finishing them 2 / 4 / 8 MFMAs before the barrier.
        constexpr int S0 = (NM >= 12) ? NM / 6 : 0;
        // One stage.  BUF: the LDS buffer multiplied.  CONV: the registers of set BUF^1 / breg hold the next stage's operands and
        // go to buffer BUF^1.  LOADS: the operands of the stage after next are requested (A into set BUF, B into breg).
        auto stage = [&](auto buf_tag, auto conv_tag, auto loads_tag, KPos kl, int sl) {
            constexpr int BUF = decltype(buf_tag)::value;
            constexpr bool conv = decltype(conv_tag)::value;
            constexpr bool loads = decltype(loads_tag)::value;
            using NB = std::integral_constant<int, BUF ^ 1>;
            using CB = std::integral_constant<int, BUF>;
            const uint8_t* st = lb + BUF * STB;
            u32x4 af[FM][NPL], bf[FN][NPL];
            // fragments in the order of their first use: (lo, hi) (hi, lo) (mid, mid)  /  (l, h) (h, l)
            constexpr int RA[3] = {NPL == 3 ? 2 : 1, 0, 1}, RB[3] = {0, NPL == 3 ? 2 : 1, 1};
            auto read_q = [&](int q) {
#pragma unroll
                for (int i = 0; i < FM; ++i) af[i][RA[q]] = *reinterpret_cast<const u32x4*>(st + ((wm * FM + i) * NPL + RA[q]) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[j][RB[q]] = *reinterpret_cast<const u32x4*>(st + A_BYTES + ((wn * FN + j) * NPL + RB[q]) * 1024 + lane * 16);
            };
            // (lo, hi) and (hi, lo) operands first, pinned in this order; the (mid, mid) ones follow behind the first product's
            // MFMAs (+1-4 %: the first MFMA of a stage waits for 4 reads instead of 9 - the compiler had shuffled them)
            read_q(0);
            __builtin_amdgcn_sched_barrier(0);
            read_q(1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (conv) write_b(NB{});
            if constexpr (loads) {
                load_b(kl, sl);
#ifndef X6_EXP_NOA
                load_a(kl, sl, CB{});
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            // smallest terms first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)  /  (l,h) (h,l) (h,h)
            constexpr int PA[6] = {NPL == 3 ? 2 : 1, 0, NPL == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0};
            int n = 0, done = 0;
#pragma unroll
            for (int t = 0; t < NPR; ++t)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j, ++n) {
                        acc[i][j] = mfma_split<NPL>(af[i][PA[t]], bf[j][PB[t]], acc[i][j]);
                        if (NPL == 3 && n == FM * FN - 1) {   // the (mid, mid) fragments are first needed by the third product
                            __builtin_amdgcn_sched_barrier(0);
                            read_q(2);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (conv) {
                            // spread the slices evenly behind the matrix instructions S0 .. NM-1
                            const int upto = (n + 1 <= S0) ? 0 : ((n + 1 - S0) * NS + (NM - S0) - 1) / (NM - S0);
#pragma unroll
                            for (int k = 0; k < NS; ++k)
#ifndef X6_EXP_NOA
                                if (k >= done && k < upto) slice(NB{}, NB{}, k);
#else
                                if (k >= done && k < upto && (k % SPU) == SPU - 1) slice(NB{}, NB{}, k);   // the LDS writes only (stale registers)
#endif
                            done = upto > done ? upto : done;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#ifndef X6_EXP_NOBAR
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        };
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        using TT = std::true_type;
        using FF = std::false_type;
        // Tiles of this workgroup: blockIdx.x, + gridDim.x, ...  While tile i is stored, tile i+1 is already under way: its
        // row table is decoded before the accumulators are staged, and its first operands (weights of stage 0, rows of stages 0
        // and 1) are requested right after - they arrive under the stores.
        const KPos kp0 = kpos_of(kt0);
        const bool work = kt0 < kt1;
        auto first_loads = [&]() {
            load_b(kp0, 0);
            load_a(kp0, 0, B0{});
            load_a(kp0, 1, B1{});
        };
        if (work) first_loads();
        int slot = 0;
        for (int vb = blockIdx.x;;) {
            if (work) {
                KPos kp = kp0;
                write_b(B0{});
#pragma unroll
                for (int k = 0; k < NS; ++k) slice(B0{}, B0{}, k);
                load_b(kp, 1);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                stamp(2);  // first stage staged
                for (int kt = kt0; kt + 1 < kt1; ++kt) {
                    const KPos kn = kpos_next(kp);
                    stage(B0{}, TT{}, TT{}, kn, 0);
                    stage(B1{}, TT{}, TT{}, kn, 1);
                    kp = kn;
                }
                stage(B0{}, TT{}, FF{}, kp, 0);
                stage(B1{}, FF{}, FF{}, kp, 0);
            }
            const int vn = vb + (int)gridDim.x;
            const bool more = P::PERSIST && vn < a.ntiles;
            int tmn = 0, tnn = 0;
            if (more) {
                tile_of(vn, tmn, tnn);
                decode(tmn, slot ^ 1);
            }
            const int* const t_yoff = tab0 + slot * P::TCOLS * BM;
            epilogue(tm, tn, t_yoff, t_yoff + BM, [&] {
                if (more) {
                    unit_setup(slot ^ 1, tnn);
                    if (work) first_loads();
                }
            });
            if (!more) break;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the staged tile
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            vb = vn; tm = tmn; tn = tnn; slot ^= 1;
            stamp_tile = vb;
            stamp(0);  // (persistent: the tile's turn begins; its rows were requested during the previous tile's stores)
        }
        if (a.amax_out) amax_flush(a.amax_out, amx, (int)blockIdx.x + wave);
    }

}

// split-K second pass: sum the partial slabs in fixed order (deterministic) and apply the epilogue.
constexpr int SPLITK_RED_EPT = 4;      // outputs per thread of splitk_reduce_kernel

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a) {
    const egr_conv_desc& d = a.d;
    const int grp = blockIdx.y;
    const float* const scg = a.scale ? a.scale + grp * d.gp : nullptr;
    const float* const shg = a.shift ? a.shift + grp * d.gp : nullptr;
    const float* const resg = a.res ? a.res + grp * d.gr : nullptr;
    const float* const rsg = a.rowscale ? a.rowscale + grp * d.grs : nullptr;
    const uint8_t* const rmg = a.rowmask ? a.rowmask + grp * d.grm : nullptr;
    float* const yg = a.y + grp * d.gy;
    const float* const wsg = a.ws + (int64_t)grp * d.split_k * a.M * a.Npad;
    const int64_t total = (int64_t)a.M * d.cout;
    const int HoWo = d.ho * d.wo;
    float amx = 0.f;
    // SPLITK_RED_EPT outputs per thread, a block apart (coalesced): fewer, longer blocks - and one abs-max atomic per BLOCK: the
    // record's 64 slots are agent-scope atomics that serialise (one per wave of a 2048-wave launch cost 20 us at batch 1).
    // The slab loads of a thread's outputs are issued four slices at a time (16 loads in flight) and added in slice order: a loop of
    // one dependent load per slice is bound by the load latency (split 16: ~24 us per launch at batch 1, measured).
    constexpr int E = SPLITK_RED_EPT;
    const int64_t slab = (int64_t)a.M * a.Npad;
    bool ok[E];
    int mm[E], cc[E];
    const float* pp[E];
    float ss[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int64_t idx = ((int64_t)blockIdx.x * E + e) * 256 + threadIdx.x;
        ok[e] = idx < total;
        mm[e] = ok[e] ? (int)(idx / d.cout) : 0;
        cc[e] = ok[e] ? (int)(idx - (int64_t)mm[e] * d.cout) : 0;
        pp[e] = wsg + (int64_t)mm[e] * a.Npad + cc[e];      // (a valid address for the idle lanes of the last block, too)
        ss[e] = 0.f;
    }
    int sp = 0;
    for (; sp + 4 <= d.split_k; sp += 4) {
        float t[4][E];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) t[u][e] = pp[e][(int64_t)(sp + u) * slab];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) ss[e] += t[u][e];
    }
    for (; sp < d.split_k; ++sp) {
        float t[E];
#pragma unroll
        for (int e = 0; e < E; ++e) t[e] = pp[e][(int64_t)sp * slab];
#pragma unroll
        for (int e = 0; e < E; ++e) ss[e] += t[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (!ok[e]) continue;
        const int m = mm[e], co = cc[e];
        const float s = ss[e];
        int n = m / HoWo, pix = m - n * HoWo;
        float sc = scg ? scg[co] : 1.f;
        float sh = shg ? shg[co] : 0.f;
        float rs = rsg ? rsg[m] : 1.f;
        float v = s * sc + sh * rs;
        int64_t ro = d.res_mode ? egr_map(d.rmap, n) + (int64_t)pix * d.ldr + co : 0;
        if (d.res_mode == EGR_RES_BEFORE_ACT) v += resg[ro];
        v = egr_act(v, d.act);
        if (d.res_mode == EGR_RES_AFTER_ACT) v += resg[ro];
        if (rmg && !rmg[m]) v = 0.f;
        int64_t yo = egr_map(d.ymap, n) + (d.out_nchw ? ((int64_t)co * HoWo + pix) : ((int64_t)pix * d.ldy + co));
        yg[yo] = v;
        amx = fmaxf(amx, fabsf(v));
    }
    if (a.amax_out) {       // (every thread of the block reaches this point)
        __shared__ float s_amx[4];
        amx = wave_max(amx);
        if ((threadIdx.x & 63) == 0) s_amx[threadIdx.x >> 6] = amx;
        __syncthreads();
        if (threadIdx.x == 0) {
            amx = fmaxf(fmaxf(s_amx[0], s_amx[1]), fmaxf(s_amx[2], s_amx[3]));
            if (amx > 0.f) __hip_atomic_fetch_max(a.amax_out + (blockIdx.x & 63), __float_as_uint(amx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// ---- small 1x1 launches (Linear layers / point-wise convs on few rows: the heads' MLPs, the transformer layers' projections in the
// training step, everything at batch 1).  On the tiled kernels above such a launch is bound by the per-stage hand-over of its K loop
// (about 0.45 us per 16-32 k: M 2048 x N 128 x K 512 = 16 us, measured inside a hipGraph) while it occupies a quarter of the CUs.
// Here a workgroup owns ONE 32 x 32 output tile and its four waves split K four ways; a wave reads its operands straight from
// global memory (L2) into registers - lane (row r, half h) takes 16 bytes of row r at k offset 8 c + 4 h of both operands, the four
// v_mfma_f32_32x32x2_f32 of a block use element t of both (the same k permutation on both sides) - all of a wave's loads of a K block of
// 64 in flight at once, no LDS, no barrier in the K loop.  The four partial tiles are summed in wave order through LDS (deterministic)
// and the epilogue (scale / shift / row scale / residual / activation / row mask, abs-max record) runs on 16-byte quads.
// fp32 matrix cores, fp32 operands: exact in the sense of the fp32 tiled kernel (another, fixed, summation order).
__global__ __launch_bounds__(256) void linear_small_kernel(const ConvArgs a) {
    const egr_conv_desc& d = a.d;
    __shared__ __attribute__((aligned(16))) float s_part[4][32][36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int grp = blockIdx.z;
    const int tm = blockIdx.x / a.tilesN, tn = blockIdx.x - tm * a.tilesN;
    const float* const xg = a.x + grp * d.gx;
    const float* const wg = a.w + grp * d.gw;
    const int HoWo = d.ho * d.wo;
    // this lane's operand rows
    const int m = tm * 32 + r;
    const bool live = m < a.M;
    int64_t xo = 0;
    if (live) {
        const int n = (HoWo == 1) ? m : m / HoWo, pix = m - n * HoWo;
        xo = egr_map(d.xmap, n) + (int64_t)pix * d.ldx;
    }
    const int col = tn * 32 + r;                      // < Npad: the packed weight matrix has Npad rows
    const int kq = a.K >> 2;                          // K % 32 == 0: a multiple of 8 per wave
    const float* pa = xg + xo + wave * kq + 4 * h;
    const float* pb = wg + (int64_t)col * a.K + wave * kq + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 64 <= kq; k += 64) {                   // eight blocks of 8 k: sixteen 16-byte loads in flight per lane
        f32x4 av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = live ? *reinterpret_cast<const f32x4*>(pa + k + 8 * u) : zero4;
            bv[u] = *reinterpret_cast<const f32x4*>(pb + k + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc, 0, 0, 0);
    }
    if (k + 32 <= kq) {                               // (K = 128 .. 255 per launch: one round trip, not four)
        f32x4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = live ? *reinterpret_cast<const f32x4*>(pa + k + 8 * u) : zero4;
            bv[u] = *reinterpret_cast<const f32x4*>(pb + k + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc, 0, 0, 0);
        k += 32;
    }
    if (k + 16 <= kq) {
        f32x4 av[2], bv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            av[u] = live ? *reinterpret_cast<const f32x4*>(pa + k + 8 * u) : zero4;
            bv[u] = *reinterpret_cast<const f32x4*>(pb + k + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc, 0, 0, 0);
        k += 16;
    }
    if (k < kq) {
        const f32x4 av = live ? *reinterpret_cast<const f32x4*>(pa + k) : zero4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(pb + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 16; ++i) s_part[wave][(i & 3) + 8 * (i >> 2) + 4 * h][r] = acc[i];
    __syncthreads();
    // thread -> (row, column quad): sum of the four K quarters in wave order, then the epilogue
    const int row = tid >> 3, cq = tid & 7;
    const int mo = tm * 32 + row, co = tn * 32 + cq * 4;
    float amx = 0.f;
    if (mo < a.M && co < d.cout) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&s_part[0][row][cq * 4]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(&s_part[w][row][cq * 4]);
            v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
        }
        const float* const scg = a.scale ? a.scale + grp * d.gp : nullptr;
        const float* const shg = a.shift ? a.shift + grp * d.gp : nullptr;
        const float rs = a.rowscale ? a.rowscale[grp * d.grs + mo] : 1.f;
        const bool keep = !(a.rowmask && !a.rowmask[grp * d.grm + mo]);
        const int n = (HoWo == 1) ? mo : mo / HoWo, pix = mo - n * HoWo;
        float* const yp = a.y + grp * d.gy + egr_map(d.ymap, n) + (int64_t)pix * d.ldy + co;
        const float* const rp = d.res_mode ? a.res + grp * d.gr + egr_map(d.rmap, n) + (int64_t)pix * d.ldr + co : nullptr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (co + e < d.cout) {
                float t = v[e] * (scg ? scg[co + e] : 1.f) + (shg ? shg[co + e] : 0.f) * rs;
                if (d.res_mode == EGR_RES_BEFORE_ACT) t += rp[e];
                t = egr_act(t, d.act);
                if (d.res_mode == EGR_RES_AFTER_ACT) t += rp[e];
                if (!keep) t = 0.f;
                v[e] = t;
                amx = fmaxf(amx, fabsf(t));
            }
        }
        if (a.vec_ok && co + 3 < d.cout) *reinterpret_cast<f32x4*>(yp) = v;
        else
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (co + e < d.cout) yp[e] = v[e];
    }
    if (a.amax_out) amax_flush(a.amax_out, amx, (int)blockIdx.x + wave);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    conv_igemm_body<BM, BN, WM, WN, false>(a);
}

// split-bf16 launches: at least two workgroups per CU (their barriers and fixed phases overlap), so at most 256 registers per
// lane; the narrower tiles hold fewer accumulators, are not persistent and are bounded for four (128 registers: measured
// 131 TFLOP/s on the 64-channel 3x3 layers against 122 at three and 112-118 at two workgroups per CU)
template <int BM, int BN, int WM, int WN, int NPL = 3>
__global__ __launch_bounds__(256, (BM * BN <= 128 * 64) ? X6_OCC_SMALL : X6_OCC_BIG) void conv_igemm_x6_kernel(const ConvArgs a) {
    conv_igemm_body<BM, BN, WM, WN, true, false, false, false, NPL>(a);
}

// 3x3 / stride 1 / pad 1 with the taps shared (see the TAP branch of conv_igemm_body)
template <int BM, int BN, int WM, int WN, int NPL = 3>
__global__ __launch_bounds__(256, 2) void conv_igemm_tap_kernel(const ConvArgs a) {
    conv_igemm_body<BM, BN, WM, WN, true, false, true, false, NPL>(a);
}

// 3x3 / stride 2 / pad 1 with the taps shared by parity class (see the TAP2 branch of conv_igemm_body)
template <int BM, int BN, int WM, int WN, int NPL = 3>
__global__ __launch_bounds__(256, 2) void conv_igemm_tap2_kernel(const ConvArgs a) {
    conv_igemm_body<BM, BN, WM, WN, true, false, false, true, NPL>(a);
}

// ---- 1x1 / stride 1 split-bf16 launches with short K (cin = 64 / 128) over many pixels: the STREAMING kernel.
// These layers are HBM-bound (K = N = 128: 1 KiB of traffic per pixel against 0.4 us of bf16 MFMA time per 1000 pixels), and
// in the tiled kernels above they spend their time in per-tile fixed costs: 8 staging barriers, a row-table decode and an LDS
// round trip of the accumulators for every 64 KiB of input.  Here nothing is tiled across waves:
//   * the WEIGHTS are stationary: a workgroup (8 waves, one per CU: 96 KiB of LDS) copies the split image of its 64 / 128 output
//     channels into LDS once and keeps it for the whole launch;
//   * the operand roles are swapped - A = weights (rows = output channels), B = activations (columns = pixels): lane
//     (p = lane & 31, h = lane >> 5) of a wave owns pixel p of the wave's 32-pixel tile.  Its B operand of a 16-deep k step is 2 x 16
//     bytes of the pixel's own NHWC row, so the activations go global memory -> registers -> (split) -> MFMA with no LDS and no
//     barrier, and the accumulator layout (4 consecutive channels of the lane's pixel per register quad) is stored with 16-byte
//     stores straight from the registers;
//   * a wave walks pixel tiles wave_id, + waves, ... on its own; the next tile's 2 * KS loads are issued register by register as the
//     current tile's k steps release them, a full tile (KS * NCF * 6 MFMAs) ahead of their use.
// k order inside a step: lane half h holds k = {4h .. 4h+3} u {8+4h .. 8+4h+3}, so one load instruction covers 32 contiguous bytes
// per pixel; the weight image (pack_w6 order: k = 8h + j) is permuted to match while it is copied into LDS.
// RESK: 0 no residual, 1 residual of the output's shape (before / behind the activation: run time), 2 half-resolution residual
// up-sampled on the fly (EGR_RES_UP2_BEFORE_ACT); the ReLU is a run-time clamp bound.
template <int KS, int NCF, int RESK, int NPL = 3>
__global__ __launch_bounds__(512) void conv_pw_x6_kernel(const ConvArgs a) {
    constexpr int WBYTES = NCF * KS * NPL * 1024;
    constexpr int NPR = split_npr(NPL);
    constexpr int PATCH = 32 * 144;        // per-wave epilogue patch: 32 pixels x 32 channels, rows padded to 144 bytes
    __shared__ __attribute__((aligned(16))) uint8_t lds[WBYTES + NCF * 32 * 8 + 8 * PATCH];
    float* const s_sc = reinterpret_cast<float*>(lds + WBYTES);
    float* const s_sh = s_sc + NCF * 32;
    const egr_conv_desc& d = a.d;
    const int grp = blockIdx.z, tn = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    const float* const scg = a.scale ? a.scale + grp * d.gp : nullptr;
    const float* const shg = a.shift ? a.shift + grp * d.gp : nullptr;
    float sa = 1.f, ads = 1.f;        // EGR_W_F16X2: activation pre-scale and its inverse
    if constexpr (NPL == 2) act_prescale(a.amax_in, lane, sa, ads);
    const float* const wdsg = (NPL == 2) ? a.wds + grp * d.gp : nullptr;
    float amx = 0.f;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.x + grp * d.gx), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.y + grp * d.gy), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.res ? a.res + grp * d.gr : a.x), 0, 0x80000000u, 0x00020000);
    const bool masked = a.mask != nullptr;       // data gradient behind a ReLU: dx = (acc [+ res]) * [mask > 0], mask laid out like y
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(masked ? a.mask + grp * d.gy : a.x), 0, 0x80000000u, 0x00020000);
    const int HoWo = d.ho * d.wo;
    constexpr int OOB = (int)0x80000000;
    const int T = (a.M + 31) >> 5, NW = gridDim.x * 8;
    int t = blockIdx.x * 8 + wave;

    // byte offset of the lane's pixel in x (OOB past the last pixel: the loads return zeros)
    auto x_off = [&](int tile) {
        const int m = tile * 32 + p;
        int n, pix;
        if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
        else { n = fdiv(m, a.dHoWo); pix = m - n * HoWo; }
        const int xb = a.x_plain ? n * (int)d.xmap.stride_inner : (int)fmap(d.xmap, a.dXin, n);
        int ipix = pix;
        if (d.stride == 2) {             // 1x1 / stride 2 (the residual blocks' down-sampling convs): input pixel (2 ho, 2 wo)
            const int ho = (a.wo_shift >= 0) ? (pix >> a.wo_shift) : fdiv(pix, a.dWo), wo = pix - ho * d.wo;
            ipix = 2 * ho * d.w + 2 * wo;
        }
        return (tile < T && m < a.M) ? (xb + ipix * d.ldx) * 4 + h * 16 : OOB;
    };
    // weights -> LDS: 16-byte loads of the image (lane (p, q) of a fragment = k 8q .. 8q+7), each written as two 8-byte pieces:
    // its half hh goes to the new lane (p, hh), position q  (new lane (p, h): k {4h .. 4h+3} u {8+4h .. 8+4h+3})
    u32x4 raw[2 * KS];
    {
        const uint8_t* const wimg = reinterpret_cast<const uint8_t*>(a.w) + (int64_t)grp * d.gw * 2;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        constexpr int NV = WBYTES / 16 / 512;
        static_assert(NV * 512 * 16 == WBYTES, "whole staging rounds");
        u32x4 wv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + 512 * i;
            const int cf = v / (KS * 64 * NPL), rem = v - cf * (KS * 64 * NPL);
            wv[i] = *reinterpret_cast<const u32x4*>(wimg + ((int64_t)(tn * NCF + cf) * a.ktiles * 2 * NPL) * 1024 + rem * 16);
        }
        const int xo = x_off(t);
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo + i * 32, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + 512 * i;
            const int lo = v & 63, blk = v >> 6;            // (fragment, k step, plane) block of 1 KiB; its lane
            uint8_t* const dst = lds + blk * 1024 + (lo & 31) * 16 + (lo >> 5) * 8;
            *reinterpret_cast<u32x2*>(dst) = u32x2{wv[i][0], wv[i][1]};
            *reinterpret_cast<u32x2*>(dst + 512) = u32x2{wv[i][2], wv[i][3]};
        }
        for (int c = tid; c < NCF * 32; c += 512) {
            const int co = tn * NCF * 32 + c;
            // (EGR_W_F16X2: the exact power-of-two descale of the accumulators rides on the channel scale)
            s_sc[c] = ((scg && co < d.cout) ? scg[co] : 1.f) * ((NPL == 2) ? ads * wdsg[co] : 1.f);
            s_sh[c] = (shg && co < d.cout) ? shg[co] : 0.f;
        }
    }
    __syncthreads();

    const float floor_ = (d.act == EGR_ACT_RELU) ? 0.f : -__builtin_inff();
    const bool res_before = d.res_mode == EGR_RES_BEFORE_ACT;
    {
        for (; t < T; t += NW) {
#ifdef PW_EXP_NOLOAD
            const int xo_n = OOB;
#else
            const int xo_n = x_off(t + NW);
#endif
            f32x16 acc[NCF];
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cf][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // split the lane's 8 values of this k step (hi + mid + lo, exact  /  h + l of the pre-scaled value)
                unsigned xh[4], xm[4], xl[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32x4& src = raw[2 * ks + (e >> 1)];
                    split_pair<NPL>(__uint_as_float(src[2 * (e & 1)]), __uint_as_float(src[2 * (e & 1) + 1]), sa, xh[e], xm[e], xl[e]);
                }
                u32x4 xb[NPL];
                xb[0] = u32x4{xh[0], xh[1], xh[2], xh[3]};
                xb[1] = u32x4{xm[0], xm[1], xm[2], xm[3]};
                if constexpr (NPL == 3) xb[2] = u32x4{xl[0], xl[1], xl[2], xl[3]};
                // the registers of a k-step pair (one 128-byte line of the pixel's row) are free: request the same pair of the wave's
                // next tile - four loads back to back, so the line is fetched once
#ifndef PW_EXP_LOAD2
                if (ks & 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        raw[2 * ks - 2 + i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo_n + (ks - 1) * 64 + i * 32, 0, 0);
                }
#else
                raw[2 * ks] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo_n + ks * 64, 0, 0);
                raw[2 * ks + 1] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo_n + ks * 64 + 32, 0, 0);
#endif
                u32x4 wf[NCF][NPL];
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        wf[cf][pl] = *reinterpret_cast<const u32x4*>(lds + ((cf * KS + ks) * NPL + pl) * 1024 + lane * 16);
                constexpr int PW[6] = {NPL == 3 ? 2 : 1, 0, NPL == 3 ? 1 : 0, 1, 0, 0}, PX[6] = {0, NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0};   // smallest products first
#pragma unroll
                for (int t6 = 0; t6 < NPR; ++t6)
#pragma unroll
                    for (int cf = 0; cf < NCF; ++cf)
#ifdef PW_EXP_NOMFMA
                        if (t6 == 0)
#endif
                        acc[cf] = mfma_split<NPL>(wf[cf][PW[t6]], xb[PX[t6]], acc[cf]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue.  Register quad g of fragment cf = channels cf*32 + 8g + 4h .. +3 of pixel p: stored as it is, an instruction
            // would write 32-byte pieces of 32 rows (measured: 4.2 TB/s write-only against 5.8 TB/s for the reads).  So each fragment goes
            // through a wave-private 32 x 32 LDS patch (rows padded to 144 bytes; no barrier - one wave's LDS operations execute in
            // order) and comes back row-major: lane = (row i*8 + lane/8, channel quad lane%8), an instruction writes 8 whole 128-byte lines.
            const int m = t * 32 + p;
            int n, pix;
            if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
            else { n = fdiv(m, a.dHoWo); pix = m - n * HoWo; }
            const bool live = m < a.M;
            const int yo_p = live ? ((a.y_plain ? n * (int)d.ymap.stride_inner : (int)fmap(d.ymap, a.dYin, n)) + pix * d.ldy + tn * NCF * 32) * 4 : OOB;
            int ro_p = OOB, ox_p = 0, oy_p = 0;
            float lx1_p = 0.f, ly1_p = 0.f;
            if constexpr (RESK == 1) {
                if (live) ro_p = ((a.r_plain ? n * (int)d.rmap.stride_inner : (int)fmap(d.rmap, a.dRin, n)) + pix * d.ldr + tn * NCF * 32) * 4;
            }
            if constexpr (RESK == 2) {
                // bilinear x2 (align_corners = True, ATen arithmetic as in upsample2x_kernel) of a half-resolution tensor
                const int ho = (a.wo_shift >= 0) ? (pix >> a.wo_shift) : fdiv(pix, a.dWo), wo = pix - ho * d.wo;
                const int hl = d.ho >> 1, wl = d.wo >> 1;
                const float shh = (d.ho > 1) ? (float)(hl - 1) / (float)(d.ho - 1) : 0.f;
                const float sww = (d.wo > 1) ? (float)(wl - 1) / (float)(d.wo - 1) : 0.f;
                const float fy = shh * (float)ho, fx = sww * (float)wo;
                const int y0 = (int)fy, x0 = (int)fx;
                // (fused multiply-subtract, as the compiler contracts the same expressions of the tiled kernels' row decode)
                ly1_p = fminf(fmaxf(__builtin_fmaf(shh, (float)ho, -(float)y0), 0.f), 1.f);
                lx1_p = fminf(fmaxf(__builtin_fmaf(sww, (float)wo, -(float)x0), 0.f), 1.f);
                if (live) ro_p = ((a.r_plain ? n * (int)d.rmap.stride_inner : (int)fmap(d.rmap, a.dRin, n)) + (y0 * wl + x0) * d.ldr + tn * NCF * 32) * 4;
                ox_p = (x0 + 1 > wl - 1) ? 0 : d.ldr * 4;
                oy_p = (y0 + 1 > hl - 1) ? 0 : wl * d.ldr * 4;
            }
            // the four rows this lane stores: their parameters come from the lanes that own those pixels
            const int qd = lane & 7, rsub = lane >> 3;
            int yo[4], ro[4], ox[4], oy[4];
            float lx1[4], ly1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int src = i * 8 + rsub;
                yo[i] = __shfl(yo_p, src) + qd * 16;
                if constexpr (RESK != 0) ro[i] = __shfl(ro_p, src) + qd * 16;
                if constexpr (RESK == 2) {
                    ox[i] = __shfl(ox_p, src); oy[i] = __shfl(oy_p, src);
                    lx1[i] = __shfl(lx1_p, src); ly1[i] = __shfl(ly1_p, src);
                }
            }
            uint8_t* const patch = lds + WBYTES + NCF * 32 * 8 + wave * PATCH;
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) {
                f32x4 r00[4], r01[4], r10[4], r11[4], mk[4];
                if constexpr (RESK != 2) {
                    if (masked) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            mk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, yo[i] + cf * 128, 0, 0));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) mk[i] = f32x4{1.f, 1.f, 1.f, 1.f};
                    }
                }
                if constexpr (RESK != 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        r00[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro[i] + cf * 128, 0, 0));
                        if constexpr (RESK == 2) {
                            r01[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro[i] + ox[i] + cf * 128, 0, 0));
                            r10[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro[i] + oy[i] + cf * 128, 0, 0));
                            r11[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro[i] + oy[i] + ox[i] + cf * 128, 0, 0));
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(patch + p * 144 + g * 32 + h * 16) =
                        f32x4{acc[cf][4 * g], acc[cf][4 * g + 1], acc[cf][4 * g + 2], acc[cf][4 * g + 3]};
                __builtin_amdgcn_wave_barrier();
                const int c = cf * 32 + 4 * qd;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc + c), sh = *reinterpret_cast<const f32x4*>(s_sh + c);
                const bool cok = tn * NCF * 32 + c < d.cout;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(patch + (i * 8 + rsub) * 144 + qd * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tt = v[e] * sc[e] + sh[e];
                        if constexpr (RESK == 1) tt += res_before ? r00[i][e] : 0.f;
                        if constexpr (RESK == 2) {
                            const float lx0 = 1.f - lx1[i], ly0 = 1.f - ly1[i];
                            tt += ly0 * (lx0 * r00[i][e] + lx1[i] * r01[i][e]) + ly1[i] * (lx0 * r10[i][e] + lx1[i] * r11[i][e]);
                        }
                        tt = tt > floor_ ? tt : floor_;
                        if constexpr (RESK == 1) tt += res_before ? 0.f : r00[i][e];
                        if constexpr (RESK != 2) tt = mk[i][e] > 0.f ? tt : 0.f;
                        v[e] = tt;
                    }
                    if (cok && yo[i] >= 0) amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
#ifdef PW_EXP_NOSTORE
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, (v[0] == 12345.678f) ? yo[i] + cf * 128 : OOB, 0, 0);
#else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, cok ? yo[i] + cf * 128 : OOB, 0, 0);
#endif
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (a.amax_out) amax_flush(a.amax_out, amx, (int)blockIdx.x * 8 + wave);
}

// persistent variant (short K: a tile is mostly fixed cost and HBM traffic - the next tile's decode and first loads overlap the
// stores; measured 106 -> 121 TFLOP/s on 1x1 128 -> 128 at 64x64 pixels, -1.5 % on the long-K layers, which keep the plain launch)
template <int BM, int BN, int WM, int WN, int NPL = 3>
__global__ __launch_bounds__(256, 2) void conv_igemm_x6p_kernel(const ConvArgs a) {
    conv_igemm_body<BM, BN, WM, WN, true, true, false, false, NPL>(a);
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(ConvArgs& a, hipStream_t s) {
    a.tilesM = (a.M + BM - 1) / BM;
    a.tilesN = (a.Npad + BN - 1) / BN;
    a.dTilesN = make_fastdiv(a.tilesN);
    a.ntiles = a.tilesM * a.tilesN;
    if (const int rc = bn_slabs(a)) return rc;
    const int ny = a.cls_mode ? 4 : a.d.split_k;
    int gx = a.ntiles;
    bool persist = false;
    const bool h2 = a.d.w_format == EGR_W_F16X2, x6 = h2 || a.d.w_format == EGR_W_BF16X3;
    if (x6 && g_persist && BM * BN > 128 * 64 && a.ktiles <= g_persist_ktiles) {
        // persistent launch: as many workgroups as stay resident (2 per CU, launch bounds), each walking `rounds` tiles; a
        // multiple of 8 so that a workgroup's tiles keep its XCD (the tile order hands each XCD a contiguous run)
        const int slots = (g_persist / (ny * a.d.groups)) & ~7;
        if (slots >= 8 && a.ntiles > slots) {
            const int rounds = (a.ntiles + slots - 1) / slots;
            gx = ((a.ntiles + rounds - 1) / rounds + 7) & ~7;
            persist = true;
            a.cnt = nullptr;    // (short K: never split in practice) the persistent tile loop keeps the two-pass reduction
        }
    }
    dim3 grid((unsigned)gx, (unsigned)ny, (unsigned)a.d.groups);
    if (persist) {
        if constexpr (BM * BN > 128 * 64) {
            if (h2) hipLaunchKernelGGL((conv_igemm_x6p_kernel<BM, BN, WM, WN, 2>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv_igemm_x6p_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, s, a);
        }
    } else if (h2)
        hipLaunchKernelGGL((conv_igemm_x6_kernel<BM, BN, WM, WN, 2>), grid, dim3(256), 0, s, a);
    else if (x6)
        hipLaunchKernelGGL((conv_igemm_x6_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, s, a);
    return egr_launch_status();
}

// fp32 weight matrix -> (hi, mid, lo) bf16 planes in MFMA-fragment order (layout: egorear_hip.h, egr_pack_w6_f32).
// One workgroup per (32-column fragment, 32-deep chunk): thread (row = tid / 8, seg = tid % 8) reads 16 bytes of its weight row
// (a row's 128 bytes are one coalesced segment) and writes 8 bytes per plane; the two threads of an 8-channel group and the 32
// rows of a lane half fill 512 contiguous bytes of the image.
__device__ __forceinline__ void pack_w6_block(const float* __restrict__ w, int npad, int K, int cfp, uint8_t* __restrict__ img, int blk, int g) {
    const int KC = K / 32;
    const int chunk = blk % KC, cf = blk / KC;
    const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
    const int col = cf * 32 + row;
    unsigned h[2] = {0u, 0u}, m[2] = {0u, 0u}, l[2] = {0u, 0u};
    if (col < npad) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(w + ((int64_t)g * npad + col) * K + chunk * 32 + seg * 4);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float a = x[2 * q], b = x[2 * q + 1];
            h[q] = cvt_pk_bf16(a, b);
            const float ra = a - bf16_lo_f32(h[q]), rb = b - bf16_hi_f32(h[q]);
            m[q] = cvt_pk_bf16(ra, rb);
            l[q] = cvt_pk_bf16(ra - bf16_lo_f32(m[q]), rb - bf16_hi_f32(m[q]));
        }
    }
    const int step = seg >> 2, half = (seg >> 1) & 1, lane = row + 32 * half;
    uint8_t* dst = img + (((int64_t)g * cfp + cf) * KC + chunk) * 6144 + step * 3072 + lane * 16 + (seg & 1) * 8;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(dst) = u32x2{h[0], h[1]};
    *reinterpret_cast<u32x2*>(dst + 1024) = u32x2{m[0], m[1]};
    *reinterpret_cast<u32x2*>(dst + 2048) = u32x2{l[0], l[1]};
}

__global__ __launch_bounds__(256) void pack_w6_kernel(const float* __restrict__ w, int npad, int K, int cfp, uint8_t* __restrict__ img) {
    pack_w6_block(w, npad, K, cfp, img, (int)blockIdx.x, (int)blockIdx.y);
}

// many matrices in one launch (the training step re-splits ~90 images after every update): a workgroup finds its matrix by
// binary search in the table's running workgroup count
__global__ __launch_bounds__(256) void pack_w6_many_kernel(const egr_w6_job* __restrict__ jobs, int count) {
    int lo = 0, hi = count - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {   // last job with first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const egr_w6_job j = jobs[lo];
    const int cfp = (j.npad / 32 + 3) / 4 * 4;
    const int per_group = cfp * (j.k / 32);
    const int64_t local = b - j.first_block;
    pack_w6_block(j.w, j.npad, j.k, cfp, reinterpret_cast<uint8_t*>(j.img), (int)(local % per_group), (int)(local / per_group));
}

// ---- EGR_W_F16X2 image of a packed fp32 weight matrix: per output channel (row) the power of two that puts the row's largest
// magnitude into [2^14, 2^15), then h = f16(w s), l = f16(w s - h) in MFMA-fragment order (layout: egorear_hip.h, egr_pack_wh2_f32)
__device__ __forceinline__ void wh2_rowscale_block(const float* __restrict__ w, int rows, int K, float* __restrict__ descale, int blk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blk * 4 + wave;                        // over groups * npad rows, one wave each
    if (row >= rows) return;
    const float* r = w + (int64_t)row * K;
    float m = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(r + k);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    if (lane == 0) {
        const int e = (int)(__float_as_uint(m) >> 23);
        int k = 141 - e;
        k = k > 60 ? 60 : (k < -60 ? -60 : k);
        descale[row] = __uint_as_float((unsigned)(127 - k) << 23);
    }
}

__global__ __launch_bounds__(256) void wh2_rowscale_kernel(const float* __restrict__ w, int rows, int K, float* __restrict__ descale) {
    wh2_rowscale_block(w, rows, K, descale, (int)blockIdx.x);
}

__device__ __forceinline__ void pack_wh2_block(const float* __restrict__ w, const float* __restrict__ descale, int npad, int K, int cfp,
                                               uint8_t* __restrict__ img, int blk, int g) {
    const int KC = K / 32;
    const int chunk = blk % KC, cf = blk / KC;
    const int row = threadIdx.x >> 3, seg = threadIdx.x & 7;
    const int col = cf * 32 + row;
    unsigned h[2] = {0u, 0u}, l[2] = {0u, 0u};
    if (col < npad) {
        const float s = 1.f / descale[(int64_t)g * npad + col];      // exact: a power of two
        const f32x4 x = *reinterpret_cast<const f32x4*>(w + ((int64_t)g * npad + col) * K + chunk * 32 + seg * 4);
#pragma unroll
        for (int q = 0; q < 2; ++q) split2_f16(x[2 * q], x[2 * q + 1], s, h[q], l[q]);
    }
    const int step = seg >> 2, half = (seg >> 1) & 1, lane = row + 32 * half;
    uint8_t* dst = img + (((int64_t)g * cfp + cf) * KC + chunk) * 4096 + step * 2048 + lane * 16 + (seg & 1) * 8;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(dst) = u32x2{h[0], h[1]};
    *reinterpret_cast<u32x2*>(dst + 1024) = u32x2{l[0], l[1]};
}

__global__ __launch_bounds__(256) void pack_wh2_kernel(const float* __restrict__ w, const float* __restrict__ descale, int npad, int K, int cfp,
                                                        uint8_t* __restrict__ img) {
    pack_wh2_block(w, descale, npad, K, cfp, img, (int)blockIdx.x, (int)blockIdx.y);
}

// every image of a job table in two launches (the training step re-splits all its weight operands after every update)
__global__ __launch_bounds__(256) void wh2_rowscale_many_kernel(const egr_wh2_job* __restrict__ jobs, int count) {
    int lo = 0, hi = count - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {   // last job with first_rblock <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_rblock <= b) lo = mid; else hi = mid - 1;
    }
    const egr_wh2_job j = jobs[lo];
    wh2_rowscale_block(j.w, j.groups * j.npad, j.k, j.descale, (int)(b - j.first_rblock));
}

__global__ __launch_bounds__(256) void pack_wh2_many_kernel(const egr_wh2_job* __restrict__ jobs, int count) {
    int lo = 0, hi = count - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_pblock <= b) lo = mid; else hi = mid - 1;
    }
    const egr_wh2_job j = jobs[lo];
    const int cfp = (j.npad / 32 + 3) / 4 * 4;
    const int per_group = cfp * (j.k / 32);
    const int64_t local = b - j.first_pblock;
    pack_wh2_block(j.w, j.descale, j.npad, j.k, cfp, reinterpret_cast<uint8_t*>(j.img), (int)(local % per_group), (int)(local / per_group));
}

enum { CFG_AUTO = -1, CFG_128x128 = 0, CFG_256x64 = 1, CFG_64x64 = 2, CFG_128x32 = 3, CFG_128x64 = 4, CFG_COUNT = 5 };
const int kBM[CFG_COUNT] = {128, 256, 64, 128, 128};
const int kBN[CFG_COUNT] = {128, 64, 64, 32, 64};
int g_force_cfg = CFG_AUTO;
int g_tap = getenv("EGR_CONV_TAP") ? atoi(getenv("EGR_CONV_TAP")) : 1;   // 0: the generic split kernel everywhere (egr_conv_set_tap)
int g_tap2 = getenv("EGR_CONV_TAP2") ? atoi(getenv("EGR_CONV_TAP2")) : 1; // 0: stride-2 3x3 launches stay on the generic split kernel
int g_last_conv_kernel = 0;   // diagnostic (tests): 0 fp32 MFMA, 1 split-bf16 generic, 2 / 3 split-bf16 tap-sharing (stride 1 / 2), 4 1x1 streaming
int g_pw = getenv("EGR_CONV_PW") ? atoi(getenv("EGR_CONV_PW")) : 1;     // 0: short-K 1x1 split launches stay on the tiled kernels
const int g_tap64_env = getenv("EGR_CONV_TAP64") ? atoi(getenv("EGR_CONV_TAP64")) : 1;
int g_tap64 = g_tap64_env;    // 0: no 64-row tiles for the tap-sharing kernel's small launches
int g_small = getenv("EGR_CONV_SMALL") ? atoi(getenv("EGR_CONV_SMALL")) : 1;        // 0: small fp32 1x1 launches stay on the tiled kernel
int g_small_k = getenv("EGR_CONV_SMALL_K") ? atoi(getenv("EGR_CONV_SMALL_K")) : 1024;            // longest K (longer: split-K on the tiled kernel)
int g_small_tiles = getenv("EGR_CONV_SMALL_TILES") ? atoi(getenv("EGR_CONV_SMALL_TILES")) : 256;  // most 32 x 32 tiles (all groups) for K > 64
int g_small_rows = getenv("EGR_CONV_SMALL_ROWS") ? atoi(getenv("EGR_CONV_SMALL_ROWS")) : 8192;   // rows x groups up to which linear_small_kernel is used
int g_pw_min_rows = getenv("EGR_CONV_PW_MIN_ROWS") ? atoi(getenv("EGR_CONV_PW_MIN_ROWS")) : 65536;   // rows x groups from which the streaming kernel is used
int g_pw_blocks = getenv("EGR_CONV_PW_BLOCKS") ? atoi(getenv("EGR_CONV_PW_BLOCKS")) : 256;          // resident workgroups (one per CU)
unsigned long long* g_dbg = nullptr;
// split-K arrival counters: one region per WORKSPACE (the slab buffer the K slices are written to).  Two launches that share a
// workspace can never overlap (they would race on the slabs themselves), two launches with different workspaces may - the graphs of
// two engine lanes replaying side by side own one workspace each.  The key survives graph capture (round 4 keyed the region by the
// launch stream: every graph torch captures is recorded on the same capture stream, so both lanes' graphs shared a region).
// Zero at load, every launch leaves its region zero again.  Regions are never re-assigned (a captured graph may hold one): the
// 65th workspace of a process gets none and its launches take the second pass.
constexpr int SPLITK_REGION = 2048, SPLITK_REGIONS = 64;
__device__ int g_splitk_cnt[SPLITK_REGION * SPLITK_REGIONS];
// 1: the last-arriving K slice of a tile reduces it (no second launch).  Correct and deterministic, but measured SLOWER than the
// second pass (batch 1: 2.36 against 1.76 ms over 25 split launches; batch 64: 15.35 against 15.26 ms): the slabs must then be
// written and read with agent-scope (`sc1`) accesses that go to memory, and one workgroup sums a tile that the second pass
// spreads over the chip.  Opt-in (egr_conv_set_splitk_fused / EGR_SPLITK_FUSED=1).
int g_splitk_fused = getenv("EGR_SPLITK_FUSED") ? atoi(getenv("EGR_SPLITK_FUSED")) : 0;
const void* g_splitk_keys[SPLITK_REGIONS];
int g_splitk_nkeys = 0;
std::mutex g_splitk_mu;

int splitk_region_of(const void* workspace) {   // -1: every region belongs to another workspace
    std::lock_guard<std::mutex> lk(g_splitk_mu);
    for (int i = 0; i < g_splitk_nkeys; ++i)
        if (g_splitk_keys[i] == workspace) return i;
    if (g_splitk_nkeys == SPLITK_REGIONS) return -1;
    g_splitk_keys[g_splitk_nkeys] = workspace;
    return g_splitk_nkeys++;
}

}  // namespace

extern "C" int64_t egr_w6_elems(int32_t npad, int32_t k) {
    if (npad <= 0 || k <= 0 || npad % 32 != 0 || k % 32 != 0) return 0;
    const int64_t cfp = (npad / 32 + 3) / 4 * 4;
    return cfp * (k / 32) * 3072;   // 6 KiB per (fragment, chunk)
}

extern "C" int egr_pack_w6_f32(const float* w, int32_t npad, int32_t k, int32_t groups, void* img, void* stream) {
    if (!w || !img) return EGR_ENULL;
    if (npad <= 0 || k <= 0 || npad % 32 != 0 || k % 32 != 0 || groups <= 0 || groups > 65535) return EGR_EINVAL;
    if (((uintptr_t)w & 15) || ((uintptr_t)img & 15)) return EGR_EINVAL;
    const int cfp = (npad / 32 + 3) / 4 * 4;
    const int64_t blocks = (int64_t)cfp * (k / 32);
    if (blocks > 0x7fffffffLL) return EGR_EINVAL;
    hipLaunchKernelGGL(pack_w6_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, w, npad, k, cfp, (uint8_t*)img);
    return egr_launch_status();
}

extern "C" int egr_pack_w6_many_f32(const egr_w6_job* jobs, int32_t count, int64_t total_blocks, void* stream) {
    if (count <= 0 || total_blocks <= 0) return 0;
    if (!jobs) return EGR_ENULL;
    if (total_blocks > 0x7fffffffLL) return EGR_EINVAL;
    hipLaunchKernelGGL(pack_w6_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, count);
    return egr_launch_status();
}

extern "C" int64_t egr_wh2_elems(int32_t npad, int32_t k) {
    if (npad <= 0 || k <= 0 || npad % 32 != 0 || k % 32 != 0) return 0;
    const int64_t cfp = (npad / 32 + 3) / 4 * 4;
    return cfp * (k / 32) * 2048;   // 4 KiB per (fragment, chunk)
}

extern "C" int egr_pack_wh2_f32(const float* w, int32_t npad, int32_t k, int32_t groups, void* img, float* descale, void* stream) {
    if (!w || !img || !descale) return EGR_ENULL;
    if (npad <= 0 || k <= 0 || npad % 32 != 0 || k % 32 != 0 || groups <= 0 || groups > 65535) return EGR_EINVAL;
    if (((uintptr_t)w & 15) || ((uintptr_t)img & 15)) return EGR_EINVAL;
    const int cfp = (npad / 32 + 3) / 4 * 4;
    const int64_t blocks = (int64_t)cfp * (k / 32), rows = (int64_t)groups * npad;
    if (blocks > 0x7fffffffLL || rows > 0x7fffffffLL) return EGR_EINVAL;
    hipLaunchKernelGGL(wh2_rowscale_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, (int)rows, k, descale);
    hipLaunchKernelGGL(pack_wh2_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, w, descale, npad, k, cfp, (uint8_t*)img);
    return egr_launch_status();
}

extern "C" int egr_pack_wh2_many_f32(const egr_wh2_job* jobs, int32_t count, int64_t total_rblocks, int64_t total_pblocks, void* stream) {
    if (count <= 0 || total_rblocks <= 0 || total_pblocks <= 0) return 0;
    if (!jobs) return EGR_ENULL;
    if (total_rblocks > 0x7fffffffLL || total_pblocks > 0x7fffffffLL) return EGR_EINVAL;
    hipLaunchKernelGGL(wh2_rowscale_many_kernel, dim3((unsigned)total_rblocks), dim3(256), 0, (hipStream_t)stream, jobs, count);
    hipLaunchKernelGGL(pack_wh2_many_kernel, dim3((unsigned)total_pblocks), dim3(256), 0, (hipStream_t)stream, jobs, count);
    return egr_launch_status();
}

extern "C" int egr_conv_debug_stamps(unsigned long long* buf) {  // diagnostic: 8 x u64 per workgroup, NULL = off
    g_dbg = buf;
    return 0;
}

// diagnostic / test knob: resident workgroup slots a persistent split-bf16 launch is sized for (0 = never persistent) and the
// largest K (in 32-deep chunks) that takes the persistent kernel.  Negative values leave a setting unchanged.  Defaults: 512, 4.
extern "C" int egr_conv_set_persist(int slots, int max_ktiles) {
    if (slots >= 0) g_persist = slots;
    if (max_ktiles >= 0) g_persist_ktiles = max_ktiles;
    return 0;
}

extern "C" int egr_conv_last_kernel(void) { return g_last_conv_kernel; }
extern "C" int egr_conv_set_splitk_fused(int on) { g_splitk_fused = on; return 0; }
extern "C" int egr_conv_set_tap(int on) { g_tap = on & 1; g_tap64 = (on & 2) ? 0 : g_tap64_env; return 0; }     // (bit 1: no 64-row tiles; clear: what EGR_CONV_TAP64 chose)
extern "C" int egr_conv_set_tapx(int32_t on, int32_t min_tiles, int32_t blocks) { return tapx_set(on, min_tiles, blocks); }

extern "C" int egr_conv_force_config(int cfg) {
    if (cfg < CFG_AUTO || cfg >= CFG_COUNT) return EGR_EINVAL;
    g_force_cfg = cfg;
    return 0;
}

static int conv_run(const egr_conv_desc* dd, const float* x, const float* w, const float* scale, const float* shift, const float* res,
                    const float* rowscale, const uint8_t* rowmask, const float* mask, float* y, float* workspace,
                    size_t workspace_floats, const egr_conv_aux* aux, void* stream) {
    if (!dd || !x || !w || !y) return EGR_ENULL;
    ConvArgs a;
    a.d = *dd;
    egr_conv_desc& d = a.d;
    if (d.cin <= 0 || d.cin % BK != 0 || d.cout <= 0 || d.kh <= 0 || d.kw <= 0 || d.stride <= 0) return EGR_EINVAL;
    if (d.n <= 0 || d.ho <= 0 || d.wo <= 0 || d.kh * d.kw > 32) return EGR_EINVAL;
    if (d.ldx % 4 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return EGR_EINVAL;  // 16-byte A/B loads
    if (d.res_mode != EGR_RES_NONE && !res) return EGR_ENULL;
    if (d.xmap.n_inner <= 0 || d.ymap.n_inner <= 0 || (d.res_mode && d.rmap.n_inner <= 0)) return EGR_EINVAL;
    if ((d.xmap.stride_inner | d.xmap.stride_outer) % 4 != 0) return EGR_EINVAL;
    if (d.groups <= 0) d.groups = 1;
    if (d.groups > 1 && ((d.gx | d.gw | d.gp | d.gy | d.gr) % 4 != 0)) return EGR_EINVAL;  // keep 16-byte alignment per group
    if (d.groups > 65535) return EGR_EINVAL;
    if (d.w_format != EGR_W_F32 && d.w_format != EGR_W_BF16X3 && d.w_format != EGR_W_F16X2) return EGR_EINVAL;
    const bool h2 = d.w_format == EGR_W_F16X2, x6 = h2 || d.w_format == EGR_W_BF16X3;   // split kernels (bf16 x 3 / fp16 x 2)
    if (x6 && d.groups > 1 && d.gw % 8 != 0) return EGR_EINVAL;
    // the fp16 scheme needs the weights' descale and the activations' abs-max record; forward launches only for now
    if (h2 && (!aux || !aux->w_descale || !aux->amax_in)) return EGR_ENULL;
    if (h2 && (((uintptr_t)aux->amax_in) & 3)) return EGR_EINVAL;
    if (aux && aux->amax_out && d.out_nchw) return EGR_EINVAL;   // the channel-major epilogue does not record max |y|
    int64_t M64 = (int64_t)d.n * d.ho * d.wo;
    if (M64 >= (1LL << 31)) return EGR_EINVAL;
    // 32-bit offsets inside the kernel: bound the furthest element each operand can touch
    auto span = [](const egr_nmap& m, int n) {
        int o = (n - 1) / m.n_inner, i = (n - 1 < m.n_inner ? n - 1 : m.n_inner - 1);
        return (int64_t)i * m.stride_inner + (int64_t)o * m.stride_outer;
    };
    if (span(d.xmap, d.n) + (int64_t)d.h * d.w * d.ldx >= (1LL << 31)) return EGR_EINVAL;
    int64_t ypix = d.out_nchw ? (int64_t)d.cout * d.ho * d.wo : (int64_t)d.ho * d.wo * d.ldy;
    if (span(d.ymap, d.n) + ypix >= (1LL << 31)) return EGR_EINVAL;
    if (d.res_mode && span(d.rmap, d.n) + (int64_t)d.ho * d.wo * d.ldr >= (1LL << 31)) return EGR_EINVAL;   // (upper bound for the half-size mode too)
    // split-bf16 launches address the activations through a 2-GiB buffer window (byte offsets, shifted by the halo bias)
    if (x6 &&
        (span(d.xmap, d.n) + (int64_t)d.h * d.w * d.ldx + 2 * (int64_t)(d.kh * d.w + d.kw + 1) * d.ldx) * 4 + 64 >= (1LL << 31))
        return EGR_EINVAL;

    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = res; a.rowscale = rowscale; a.rowmask = rowmask;
    a.y = y; a.ws = workspace;
    a.mask = mask;
    a.wds = aux ? aux->w_descale : nullptr;
    a.amax_in = aux ? aux->amax_in : nullptr;
    a.amax_out = aux ? aux->amax_out : nullptr;
    // train-mode BatchNorm statistics in the epilogue (egr_conv_aux.bn_partials): the plain 16-byte epilogue of a raw conv output only
    const bool bnst = aux && aux->bn_partials;
    a.bn_part = bnst ? aux->bn_partials : nullptr;
    a.bn_tiles_host = bnst ? aux->bn_tiles_out : nullptr;
    a.bn_cap = bnst ? (size_t)aux->bn_capacity : 0;
    if (bnst) {
        if (!aux->bn_tiles_out) return EGR_ENULL;
        if (mask || d.transposed || d.out_nchw || d.act != EGR_ACT_NONE || d.res_mode != EGR_RES_NONE || rowscale || rowmask ||
            d.cout % 64 != 0 || ((uintptr_t)aux->bn_partials & 15))
            return EGR_EINVAL;
        d.split_k = 1;        // (the statistics are taken where the tile is stored, not in a split-K reduction)
    }
    a.cnt = nullptr;
    a.dbg = g_dbg;
    a.M = (int)M64;
    a.Npad = (d.cout + 31) / 32 * 32;
    a.K = d.kh * d.kw * d.cin;
    auto log2_exact = [](int v) { int l = 0; while ((1 << l) < v) ++l; return ((1 << l) == v) ? l : -1; };
    // stride-2 data gradient with even output size: four parity classes, each enumerating a (ho/2, wo/2) grid
    a.cls_mode = (d.transposed && d.stride == 2 && (d.ho % 2 == 0) && (d.wo % 2 == 0) && !d.out_nchw && !rowscale && !rowmask &&
                  d.split_k <= 1) ? 1 : 0;
    const int eho = a.cls_mode ? d.ho / 2 : d.ho, ewo = a.cls_mode ? d.wo / 2 : d.wo;
    if (a.cls_mode) a.M = d.n * eho * ewo;
    a.howo_shift = log2_exact(eho * ewo);
    a.wo_shift = log2_exact(ewo);
    a.x_plain = d.xmap.n_inner >= d.n;
    a.y_plain = d.ymap.n_inner >= d.n;
    a.r_plain = d.res_mode ? (d.rmap.n_inner >= d.n) : 1;
    a.dHoWo = make_fastdiv(eho * ewo);
    a.dWo = make_fastdiv(ewo);
    a.dXin = make_fastdiv(d.xmap.n_inner);
    a.dYin = make_fastdiv(d.ymap.n_inner);
    a.dRin = make_fastdiv(d.res_mode ? d.rmap.n_inner : 1);
    a.cblocks = d.cin / BK;
    a.taps = d.kh * d.kw;
    a.ktiles = a.taps * a.cblocks;
    // 16-byte epilogue accesses need every (row, channel-quad) address aligned
    a.vec_ok = !d.out_nchw && (d.ldy % 4 == 0) && (((uintptr_t)y & 15) == 0) &&
               ((d.ymap.stride_inner | d.ymap.stride_outer) % 4 == 0) &&
               (!scale || ((uintptr_t)scale & 15) == 0) && (!shift || ((uintptr_t)shift & 15) == 0);
    if (d.res_mode)
        a.vec_ok = a.vec_ok && (d.ldr % 4 == 0) && (((uintptr_t)res & 15) == 0) &&
                   ((d.rmap.stride_inner | d.rmap.stride_outer) % 4 == 0);
    if (d.res_mode == EGR_RES_UP2_BEFORE_ACT) {  // residual upsampled on the fly: 16-byte path, even output size, ReLU / none
        if (!a.vec_ok || d.cout % 4 != 0 || rowscale || rowmask || mask || d.out_nchw || d.transposed || (d.ho & 1) || (d.wo & 1) ||
            d.act == EGR_ACT_GELU)
            return EGR_EINVAL;
        d.split_k = 1;
    }
    if (bnst && !a.vec_ok) return EGR_EINVAL;
    if (mask) {  // masked data gradient: plain 16-byte epilogue only
        if (!a.vec_ok || d.cout % 4 != 0 || scale || shift || rowscale || rowmask || d.act != EGR_ACT_NONE || d.out_nchw ||
            d.res_mode == EGR_RES_AFTER_ACT || ((uintptr_t)mask & 15))
            return EGR_EINVAL;
        d.split_k = 1;   // the mask is applied in the tile epilogue, not in the split-K reduction
    }

    // ---- small fp32 1x1 / stride 1 launches: one 32 x 32 tile per workgroup, K split over its four waves (linear_small_kernel)
    if (g_small && g_force_cfg == CFG_AUTO && !x6 && d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad == 0 && d.h == d.ho && d.w == d.wo &&
        !d.out_nchw && !mask && !a.cls_mode && !bnst && d.res_mode != EGR_RES_UP2_BEFORE_ACT && a.K % 32 == 0 && a.K <= g_small_k && d.split_k <= 1 &&
        (int64_t)a.M * d.groups <= g_small_rows) {
        a.tilesM = (a.M + 31) / 32;
        a.tilesN = a.Npad / 32;
        const int64_t tiles = (int64_t)a.tilesM * a.tilesN;
        // (measured inside a hipGraph, per launch: 64 tiles K 128 4.1 us against 6.5 tiled, 256 tiles K 512 12.2 against 16.2, 1024 tiles
        // K 64 6.0 against 10.6 - but 1024 tiles K 128 25 against 12, and M 32 x K 32768 0.34 ms against 0.11 with split-K)
        if (tiles * d.groups <= g_small_tiles || a.K <= 64) {
            hipLaunchKernelGGL(linear_small_kernel, dim3((unsigned)tiles, 1, (unsigned)d.groups), dim3(256), 0, (hipStream_t)stream, a);
            g_last_conv_kernel = 5;
            return egr_launch_status();
        }
    }

    // ---- the fp16 scheme's 3x3 (and wide 1x1) launches with enough tiles - forward, and the training step's statistics-epilogue and
    // stride-1 data-gradient (plain / masked) ones: role-split persistent workgroups (egr_conv_tapx.hip)
    if (g_tap && g_force_cfg == CFG_AUTO && h2 && ((d.kh == 3 && d.kw == 3) || (d.kh == 1 && d.kw == 1 && d.cin >= 256))) {
        const int rc = tapx_try(a, span(d.ymap, d.n) + ypix, d.res_mode ? span(d.rmap, d.n) + (int64_t)d.ho * d.wo * d.ldr : 0, (hipStream_t)stream);
        if (rc != TAPX_NO) {
            g_last_conv_kernel = 6;
            return rc;
        }
    }
    // ---- 3x3 / stride 1 / pad 1 split launches whose tiles are whole image rows: the tap-sharing kernel
    if (g_tap && g_force_cfg == CFG_AUTO && x6 && d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad == 1 &&
        !a.cls_mode && d.split_k <= 1 && (d.wo == 8 || d.wo == 16 || d.wo == 32 || d.wo == 64) && d.ho == d.h && d.wo == d.w &&
        a.Npad % 64 == 0 && a.M >= 2048) {
        // 256 x 64 tiles for 64 / 192 output channels (as many MFMAs per tap as 128 x 128), 128 x 128 when that fills the chip, else 128 x 64
        const int P = d.ho * d.wo;
        auto fits_tile = [&](int bm) {
            if (a.M % bm != 0 || !((P % bm == 0) || (bm % P == 0))) return false;
            const int hp = P >= bm ? (bm / d.wo + 2) * (d.wo + 2) : (bm / P) * (d.ho + 2) * (d.wo + 2);
            return hp <= tap_hpmax(bm);
        };
        int bm = 0, bn = 0;
        if (a.Npad % 128 == 0 && fits_tile(128) && (int64_t)(a.M / 128) * (a.Npad / 128) * d.groups >= 256) { bm = 128; bn = 128; }
        else if (fits_tile(256) && (int64_t)(a.M / 256) * (a.Npad / 64) * d.groups >= 256) { bm = 256; bn = 64; }
        else if (fits_tile(128)) { bm = 128; bn = 64; }
        // few rows (batch 1: layer1 has 128 tiles of 128 x 64, layer2 64): 64-row tiles put twice the workgroups on the chip
        if (g_tap64 && h2 && bm == 128 && bn == 64 && !bnst && fits_tile(64) && (int64_t)(a.M / 128) * (a.Npad / 64) * d.groups < 256) bm = 64;
        if (bm) {
            d.split_k = 1;
            a.ktiles_per_split = a.ktiles;
            a.tilesM = a.M / bm;
            a.tilesN = a.Npad / bn;
            a.dTilesN = make_fastdiv(a.tilesN);
            a.ntiles = a.tilesM * a.tilesN;
            if (const int rcb = bn_slabs(a)) return rcb;
            dim3 grid((unsigned)a.ntiles, 1, (unsigned)d.groups);
            if (h2) {
                if (bm == 64) hipLaunchKernelGGL((conv_igemm_tap_kernel<64, 64, 2, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
                else if (bm == 256) hipLaunchKernelGGL((conv_igemm_tap_kernel<256, 64, 4, 1, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
                else if (bn == 128) hipLaunchKernelGGL((conv_igemm_tap_kernel<128, 128, 2, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
                else hipLaunchKernelGGL((conv_igemm_tap_kernel<128, 64, 2, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
            } else if (bm == 256) hipLaunchKernelGGL((conv_igemm_tap_kernel<256, 64, 4, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
            else if (bn == 128) hipLaunchKernelGGL((conv_igemm_tap_kernel<128, 128, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((conv_igemm_tap_kernel<128, 64, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
            g_last_conv_kernel = 2;
            return egr_launch_status();
        }
    }
    // ---- 3x3 / stride 2 / pad 1 split launches on even images: taps shared by parity class
    if (g_tap && g_tap2 && g_force_cfg == CFG_AUTO && x6 && d.kh == 3 && d.kw == 3 && d.stride == 2 && d.pad == 1 &&
        !d.transposed && !a.cls_mode && d.split_k <= 1 && !mask && (d.wo == 8 || d.wo == 16 || d.wo == 32) &&
        d.h == 2 * d.ho && d.w == 2 * d.wo && a.Npad % 64 == 0 && a.M >= 2048 && a.M % 128 == 0) {
        const int P = d.ho * d.wo;
        const int hp = P >= 128 ? (128 / d.wo + 1) * (d.wo + 1) : (128 / P) * (d.ho + 1) * (d.wo + 1);
        if (((P % 128 == 0) || (128 % P == 0)) && hp <= 192) {
            const int bn = (a.Npad % 128 == 0 && (int64_t)(a.M / 128) * (a.Npad / 128) * d.groups >= 256) ? 128 : 64;
            d.split_k = 1;
            a.ktiles_per_split = a.ktiles;
            a.tilesM = a.M / 128;
            a.tilesN = a.Npad / bn;
            a.dTilesN = make_fastdiv(a.tilesN);
            a.ntiles = a.tilesM * a.tilesN;
            if (const int rcb = bn_slabs(a)) return rcb;
            dim3 grid((unsigned)a.ntiles, 1, (unsigned)d.groups);
            if (h2) {
                if (bn == 128) hipLaunchKernelGGL((conv_igemm_tap2_kernel<128, 128, 2, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
                else hipLaunchKernelGGL((conv_igemm_tap2_kernel<128, 64, 2, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
            } else if (bn == 128) hipLaunchKernelGGL((conv_igemm_tap2_kernel<128, 128, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((conv_igemm_tap2_kernel<128, 64, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
            g_last_conv_kernel = 3;
            return egr_launch_status();
        }
    }
    // ---- 1x1 / stride 1 split launches with cin = 64 / 128 over many pixels: weights stationary in LDS, activations streamed
    if (g_tap && g_pw && !bnst && g_force_cfg == CFG_AUTO && x6 && d.kh == 1 && d.kw == 1 && d.pad == 0 &&
        (d.stride == 1 || (d.stride == 2 && !d.transposed && d.ho == (d.h - 1) / 2 + 1 && d.wo == (d.w - 1) / 2 + 1)) &&
        !a.cls_mode && d.split_k <= 1 && !d.out_nchw && !rowscale && !rowmask && a.vec_ok && d.cout % 4 == 0 &&
        d.act != EGR_ACT_GELU && (d.cin == 64 || d.cin == 128) && a.Npad % 64 == 0 && (d.stride == 2 || (d.h == d.ho && d.w == d.wo)) &&
        (int64_t)a.M * d.groups >= g_pw_min_rows && (span(d.ymap, d.n) + ypix) * 4 < (1LL << 31) &&
        (!d.res_mode || (span(d.rmap, d.n) + (int64_t)d.ho * d.wo * d.ldr) * 4 < (1LL << 31))) {
        const int ncf = (a.Npad % 128 == 0) ? 4 : 2;
        const int tiles_n = a.Npad / (ncf * 32);
        int nblk = g_pw_blocks / (tiles_n * d.groups);
        if (nblk < 1) nblk = 1;
        const int t32 = (a.M + 31) / 32;
        if (nblk * 8 > t32) nblk = (t32 + 7) / 8;
        dim3 grid((unsigned)nblk, (unsigned)tiles_n, (unsigned)d.groups);
        const int resk = d.res_mode == EGR_RES_NONE ? 0 : (d.res_mode == EGR_RES_UP2_BEFORE_ACT ? 2 : 1);
        auto launch = [&](auto ks_tag, auto ncf_tag) {
            constexpr int KS = decltype(ks_tag)::value, NCF = decltype(ncf_tag)::value;
            if (h2) {
                if (resk == 0) hipLaunchKernelGGL((conv_pw_x6_kernel<KS, NCF, 0, 2>), grid, dim3(512), 0, (hipStream_t)stream, a);
                else if (resk == 1) hipLaunchKernelGGL((conv_pw_x6_kernel<KS, NCF, 1, 2>), grid, dim3(512), 0, (hipStream_t)stream, a);
                else hipLaunchKernelGGL((conv_pw_x6_kernel<KS, NCF, 2, 2>), grid, dim3(512), 0, (hipStream_t)stream, a);
            } else if (resk == 0) hipLaunchKernelGGL((conv_pw_x6_kernel<KS, NCF, 0>), grid, dim3(512), 0, (hipStream_t)stream, a);
            else if (resk == 1) hipLaunchKernelGGL((conv_pw_x6_kernel<KS, NCF, 1>), grid, dim3(512), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((conv_pw_x6_kernel<KS, NCF, 2>), grid, dim3(512), 0, (hipStream_t)stream, a);
        };
        using C2 = std::integral_constant<int, 2>;
        using C4 = std::integral_constant<int, 4>;
        using C8 = std::integral_constant<int, 8>;
        if (d.cin == 128 && ncf == 4) launch(C8{}, C4{});
        else if (d.cin == 128) launch(C8{}, C2{});
        else if (ncf == 4) launch(C4{}, C4{});
        else launch(C4{}, C2{});
        g_last_conv_kernel = 4;
        return egr_launch_status();
    }
    g_last_conv_kernel = x6 ? 1 : 0;

    // ---- tile configuration
    int cfg = g_force_cfg;
    if (cfg == CFG_AUTO) {
        if (a.Npad == 32) cfg = CFG_128x32;
        else if (a.M <= 4096) cfg = CFG_64x64;
        else if (x6) {
            // split-bf16 launches: a stage is only 6 MFMAs per 32x32 fragment, so the wave tile must be at least 64x32 to keep
            // the barrier count down (N = 64 / 192: 128x64, measured 123 vs 94 TF on 64x64), and 128x128 needs two resident
            // workgroups per CU to overlap its barriers (layer4 alone: 256 workgroups ran 91 TF, as 128x64 121 TF)
            const int64_t blocks128 = (int64_t)((a.M + 127) / 128) * ((a.Npad + 127) / 128) * d.groups;
            cfg = (a.Npad % 128 == 0 && blocks128 >= 256) ? CFG_128x128 : CFG_128x64;   // (256 workgroups of 128x128: 186 TF, as 128x64: 170)
        } else if (a.Npad % 128 == 0) cfg = CFG_128x128;
        else cfg = CFG_64x64;  // N = 64 / 192: 128x64 wins the isolated micro-benchmark (+15 %) but not the pipeline (26.2 vs 26.0 ms); 256x64 runs at 1 workgroup/CU
    }
    const int bm = kBM[cfg], bn = kBN[cfg];

    // ---- split-K: auto (0) fills the chip for skinny GEMMs with long K
    int blocks = ((a.M + bm - 1) / bm) * ((a.Npad + bn - 1) / bn) * d.groups;
    if (a.cls_mode) d.split_k = 1;
    if (d.split_k <= 0) {
        d.split_k = 1;
        // (also: a few hundred blocks each walking a very long K alone - heatmap_proj.0: 240 blocks x 128 chunks, latency-bound)
        // (small batches: 128 .. 511 blocks walking 18+ chunks alone leave most CUs idle and are latency-bound as well - layer1 / layer2 at
        // batch 1: 42 -> ~25 us with the K range split 2-4 ways)
        static const int mid_kt = getenv("EGR_SPLITK_MID_KT") ? atoi(getenv("EGR_SPLITK_MID_KT")) : 16;   // tuning knob
        if ((blocks < 128 ? a.ktiles >= 32 : (blocks < 512 && !x6 && (a.ktiles >= 64 || (mid_kt > 0 && a.ktiles >= mid_kt)))) && workspace) {
            // skinny GEMM streaming a long weight matrix (mlp_pred.0: 268 MB): a block's two-stage pipeline moves ~8 GB/s,
            // so the HBM rate is set by how many blocks stream at once -> aim at 4 per CU
            static const int target = getenv("EGR_SPLITK_TARGET") ? atoi(getenv("EGR_SPLITK_TARGET")) : 1024;   // tuning knob
            int s = target / blocks;
            if (s > a.ktiles / 8) s = a.ktiles / 8;
            if (s > 32) s = 32;
            while (s > 1 && (size_t)s * a.M * a.Npad * d.groups > workspace_floats) --s;
            if (s > 1) d.split_k = s;
        }
    }
    if (d.split_k > a.ktiles) d.split_k = a.ktiles;
    if (d.split_k > 1) {
        if (!workspace) return EGR_ENULL;
        if ((size_t)d.split_k * a.M * a.Npad * d.groups > workspace_floats) return EGR_EWORKSPACE;
        if ((uintptr_t)workspace & 15) return EGR_EINVAL;
    }
    a.ktiles_per_split = (a.ktiles + d.split_k - 1) / d.split_k;
    d.split_k = (a.ktiles + a.ktiles_per_split - 1) / a.ktiles_per_split;  // no empty splits

    hipStream_t s = (hipStream_t)stream;
    int rc;
    a.cnt = nullptr;
    {
        const int64_t tiles_all = (int64_t)((a.M + bm - 1) / bm) * ((a.Npad + bn - 1) / bn) * d.groups;
        if (d.split_k > 1 && g_splitk_fused && tiles_all <= SPLITK_REGION) {
            static int* base = nullptr;
            if (!base && hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_splitk_cnt)) != hipSuccess) base = nullptr;
            const int region = base ? splitk_region_of(workspace) : -1;
            if (region >= 0) a.cnt = base + region * SPLITK_REGION;
        }
    }
    switch (cfg) {
        case CFG_128x128: rc = launch_cfg<128, 128, 2, 2>(a, s); break;
        case CFG_256x64: rc = launch_cfg<256, 64, 4, 1>(a, s); break;
        case CFG_64x64: rc = launch_cfg<64, 64, 2, 2>(a, s); break;
        case CFG_128x64: rc = launch_cfg<128, 64, 2, 2>(a, s); break;
        default: rc = launch_cfg<128, 32, 4, 1>(a, s); break;
    }
    if (rc) return rc;
    if (d.split_k > 1 && !a.cnt) {
        int64_t total = (int64_t)a.M * d.cout;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 256 * SPLITK_RED_EPT - 1) / (256 * SPLITK_RED_EPT)), (unsigned)d.groups), dim3(256), 0, s, a);
        rc = egr_launch_status();
    }
    return rc;
}

extern "C" int egr_conv2d_nhwc_f32(const egr_conv_desc* dd, const float* x, const float* w, const float* scale,
                                   const float* shift, const float* res, const float* rowscale,
                                   const uint8_t* rowmask, float* y, float* workspace, size_t workspace_floats,
                                   void* stream) {
    return conv_run(dd, x, w, scale, shift, res, rowscale, rowmask, nullptr, y, workspace, workspace_floats, nullptr, stream);
}

extern "C" int egr_conv2d_nhwc_ex_f32(const egr_conv_desc* dd, const float* x, const void* w, const float* scale,
                                      const float* shift, const float* res, const float* rowscale,
                                      const uint8_t* rowmask, float* y, float* workspace, size_t workspace_floats,
                                      const egr_conv_aux* aux, void* stream) {
    return conv_run(dd, x, static_cast<const float*>(w), scale, shift, res, rowscale, rowmask, nullptr, y, workspace, workspace_floats, aux, stream);
}

extern "C" int egr_conv2d_masked_f32(const egr_conv_desc* dd, const float* x, const float* w, const float* res, const float* mask,
                                     float* y, float* workspace, size_t workspace_floats, void* stream) {
    if (!mask) return EGR_ENULL;
    return conv_run(dd, x, w, nullptr, nullptr, res, nullptr, nullptr, mask, y, workspace, workspace_floats, nullptr, stream);
}

extern "C" int egr_conv2d_masked_ex_f32(const egr_conv_desc* dd, const float* x, const void* w, const float* res, const float* mask, float* y,
                                        float* workspace, size_t workspace_floats, const egr_conv_aux* aux, void* stream) {
    if (!mask) return EGR_ENULL;
    return conv_run(dd, x, static_cast<const float*>(w), nullptr, nullptr, res, nullptr, nullptr, mask, y, workspace, workspace_floats, aux, stream);
}
