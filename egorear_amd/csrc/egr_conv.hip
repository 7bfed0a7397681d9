// Implicit-GEMM convolution / linear layer on the fp32 matrix cores of gfx950.
//
//   GEMM view:  M = n*ho*wo output pixels,  N = cout,  K = kh*kw*cin  (k = (kh,kw,ci), ci fastest)
//   A[m][k]  = x[n, ho*s-p+kh, wo*s-p+kw, ci]   (NHWC gather, zero outside the image)
//   B[k][co] = w[co][k]                           (weights pre-packed [cout_pad][K])
//
// Tiling: a workgroup owns a BM x BN output tile and walks K in chunks of 32
// (one filter tap x 32 input channels, so every A row of a chunk is 128 contiguous bytes).
// Chunk t+1 is fetched global->registers (16-byte loads) while chunk t is multiplied out of LDS;
// LDS rows are padded to 36 floats so the ds_read_b128 fragment reads are bank-conflict free.
// Each wave owns FM x FN accumulator tiles of v_mfma_f32_32x32x2_f32; a ds_read_b128 gives a lane
// four consecutive k of its row, which feed four MFMA k-steps (lane half h covers k = kk+4h+t in
// step t — A and B use the same permutation, so the chunk's 32 products are each summed once).
// f32 MFMA is a k-ordered fp32 fma chain, so results are deterministic and fp32-exact in the
// reference's sense (no reduced-precision path exists on gfx950, and none is wanted: argmax
// indices must match the reference bit for bit).
#include "egr_common.h"

namespace {

struct ConvArgs {
    egr_conv_desc d;
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    const float* rowscale;
    const uint8_t* rowmask;
    float* y;
    float* ws;
    int M, Npad, K;
    int ktiles, ktiles_per_split;
    int tilesM, tilesN;
    int cblocks;  // cin / 32
};

constexpr int BK = 32;
constexpr int LDSS = 36;  // padded LDS row stride (floats)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int RPT = NT / 8;  // rows covered per load pass
    constexpr int IA = BM / RPT, IB = BN / RPT;
    static_assert(BM % RPT == 0 && BN % RPT == 0, "tile/threads mismatch");
    static_assert(FM >= 1 && FN >= 1, "wave tile must be >= 32x32");

    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDSS];
    __shared__ int s_yoff[BM];
    __shared__ int s_roff[BM];
    float* sA = lds;
    float* sB = lds + BM * LDSS;

    const egr_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an L2; hand each XCD a contiguous run of tiles,
    // with the N tiles of one M tile adjacent so the activation tile is fetched into one L2 only.
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
    const int tn = bid % a.tilesN, tm = bid / a.tilesN;
    const int split = blockIdx.y;
    const int kt0 = split * a.ktiles_per_split;
    const int kt1 = min(a.ktiles, kt0 + a.ktiles_per_split);

    const int HoWo = d.ho * d.wo;
    const int seg = tid & 7, r0 = tid >> 3;

    // ---- per-thread row descriptors for the A gather
    int xb[IA], hi0[IA], wi0[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        int m = tm * BM + r0 + i * RPT;
        if (m < a.M) {
            int n = m / HoWo;
            int pix = m - n * HoWo;
            int ho = pix / d.wo;
            int wo = pix - ho * d.wo;
            xb[i] = (int)egr_map(d.xmap, n);
            hi0[i] = ho * d.stride - d.pad;
            wi0[i] = wo * d.stride - d.pad;
        } else {
            xb[i] = 0;
            hi0[i] = -(1 << 20);
            wi0[i] = 0;
        }
    }
    // ---- output row offsets (y / res) into LDS, read back in the epilogue
    for (int r = tid; r < BM; r += NT) {
        int m = tm * BM + r;
        int yo = -1, ro = 0;
        if (m < a.M) {
            int n = m / HoWo;
            int pix = m - n * HoWo;
            yo = (int)egr_map(d.ymap, n) + (d.out_nchw ? pix : pix * d.ldy);
            if (d.res_mode) ro = (int)egr_map(d.rmap, n) + pix * d.ldr;
        }
        s_yoff[r] = yo;
        s_roff[r] = ro;
    }

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[IA], rb[IB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tiles = [&](int kt) {
        int tap = kt / a.cblocks;
        int c0 = (kt - tap * a.cblocks) * BK;
        int kh = tap / d.kw, kw = tap - kh * d.kw;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            int hi = hi0[i] + kh, wi = wi0[i] + kw;
            bool ok = (hi >= 0) & (hi < d.h) & (wi >= 0) & (wi < d.w);
            const float* p = a.x + (int64_t)xb[i] + (int64_t)(hi * d.w + wi) * d.ldx + c0 + seg * 4;
            ra[i] = ok ? *reinterpret_cast<const f32x4*>(p) : zero4;
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            int co = tn * BN + r0 + i * RPT;
            const float* p = a.w + (int64_t)co * a.K + kt * BK + seg * 4;
            rb[i] = (co < a.Npad) ? *reinterpret_cast<const f32x4*>(p) : zero4;
        }
    };

    if (kt0 < kt1) load_tiles(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
        for (int i = 0; i < IA; ++i) *reinterpret_cast<f32x4*>(&sA[(r0 + i * RPT) * LDSS + seg * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < IB; ++i) *reinterpret_cast<f32x4*>(&sB[(r0 + i * RPT) * LDSS + seg * 4]) = rb[i];
        __syncthreads();
        if (kt + 1 < kt1) load_tiles(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            f32x4 av[FM], bv[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                av[i] = *reinterpret_cast<const f32x4*>(&sA[(wm * TM + i * 32 + l31) * LDSS + kk + 4 * half]);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                bv[j] = *reinterpret_cast<const f32x4*>(&sB[(wn * TN + j * 32 + l31) * LDSS + kk + 4 * half]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t], bv[j][t], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    if (d.split_k > 1) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                int co = tn * BN + wn * TN + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    int m = tm * BM + row;
                    if (m < a.M && co < a.Npad) a.ws[((int64_t)split * a.M + m) * a.Npad + co] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        int co = tn * BN + wn * TN + j * 32 + l31;
        bool cok = co < d.cout;
        float sc = (cok && a.scale) ? a.scale[co] : 1.f;
        float sh = (cok && a.shift) ? a.shift[co] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                int yo = s_yoff[row];
                if (!cok || yo < 0) continue;
                int m = tm * BM + row;
                float rs = a.rowscale ? a.rowscale[m] : 1.f;
                float v = acc[i][j][r] * sc + sh * rs;
                if (d.res_mode == EGR_RES_BEFORE_ACT) v += a.res[(int64_t)s_roff[row] + co];
                v = egr_act(v, d.act);
                if (d.res_mode == EGR_RES_AFTER_ACT) v += a.res[(int64_t)s_roff[row] + co];
                if (a.rowmask && !a.rowmask[m]) v = 0.f;
                int64_t o = d.out_nchw ? ((int64_t)yo + (int64_t)co * HoWo) : ((int64_t)yo + co);
                a.y[o] = v;
            }
        }
    }
}

// split-K second pass: sum the partial slabs in fixed order (deterministic) and apply the epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a) {
    const egr_conv_desc& d = a.d;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)a.M * d.cout;
    if (idx >= total) return;
    int m = (int)(idx / d.cout);
    int co = (int)(idx - (int64_t)m * d.cout);
    float s = 0.f;
    for (int sp = 0; sp < d.split_k; ++sp) s += a.ws[((int64_t)sp * a.M + m) * a.Npad + co];
    const int HoWo = d.ho * d.wo;
    int n = m / HoWo, pix = m - n * HoWo;
    float sc = a.scale ? a.scale[co] : 1.f;
    float sh = a.shift ? a.shift[co] : 0.f;
    float rs = a.rowscale ? a.rowscale[m] : 1.f;
    float v = s * sc + sh * rs;
    int64_t ro = d.res_mode ? egr_map(d.rmap, n) + (int64_t)pix * d.ldr + co : 0;
    if (d.res_mode == EGR_RES_BEFORE_ACT) v += a.res[ro];
    v = egr_act(v, d.act);
    if (d.res_mode == EGR_RES_AFTER_ACT) v += a.res[ro];
    if (a.rowmask && !a.rowmask[m]) v = 0.f;
    int64_t yo = egr_map(d.ymap, n) + (d.out_nchw ? ((int64_t)co * HoWo + pix) : ((int64_t)pix * d.ldy + co));
    a.y[yo] = v;
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(ConvArgs& a, hipStream_t s) {
    a.tilesM = (a.M + BM - 1) / BM;
    a.tilesN = (a.Npad + BN - 1) / BN;
    dim3 grid((unsigned)(a.tilesM * a.tilesN), (unsigned)a.d.split_k, 1);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), 0, s, a);
    return egr_launch_status();
}

}  // namespace

extern "C" int egr_conv2d_nhwc_f32(const egr_conv_desc* dd, const float* x, const float* w, const float* scale,
                                   const float* shift, const float* res, const float* rowscale,
                                   const uint8_t* rowmask, float* y, float* workspace, size_t workspace_floats,
                                   void* stream) {
    if (!dd || !x || !w || !y) return EGR_ENULL;
    ConvArgs a;
    a.d = *dd;
    egr_conv_desc& d = a.d;
    if (d.cin <= 0 || d.cin % BK != 0 || d.cout <= 0 || d.kh <= 0 || d.kw <= 0 || d.stride <= 0) return EGR_EINVAL;
    if (d.n <= 0 || d.ho <= 0 || d.wo <= 0) return EGR_EINVAL;
    if (d.ldx % 4 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return EGR_EINVAL;  // 16-byte A/B loads
    if (d.res_mode != EGR_RES_NONE && !res) return EGR_ENULL;
    if (d.xmap.n_inner <= 0 || d.ymap.n_inner <= 0 || (d.res_mode && d.rmap.n_inner <= 0)) return EGR_EINVAL;
    if ((d.xmap.stride_inner | d.xmap.stride_outer) % 4 != 0) return EGR_EINVAL;
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
    if (d.res_mode && span(d.rmap, d.n) + (int64_t)d.ho * d.wo * d.ldr >= (1LL << 31)) return EGR_EINVAL;

    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = res; a.rowscale = rowscale; a.rowmask = rowmask;
    a.y = y; a.ws = workspace;
    a.M = (int)M64;
    a.Npad = (d.cout + 31) / 32 * 32;
    a.K = d.kh * d.kw * d.cin;
    a.cblocks = d.cin / BK;
    a.ktiles = d.kh * d.kw * a.cblocks;

    // ---- tile configuration
    enum { CFG_128x128, CFG_256x64, CFG_64x64, CFG_128x32 } cfg;
    int bm, bn;
    if (a.Npad == 32) { cfg = CFG_128x32; bm = 128; bn = 32; }
    else if (a.M <= 4096) { cfg = CFG_64x64; bm = 64; bn = 64; }
    else if (a.Npad % 128 == 0) { cfg = CFG_128x128; bm = 128; bn = 128; }
    else { cfg = CFG_256x64; bm = 256; bn = 64; }

    // ---- split-K: auto (0) fills the chip for skinny GEMMs with long K
    int blocks = ((a.M + bm - 1) / bm) * ((a.Npad + bn - 1) / bn);
    if (d.split_k <= 0) {
        d.split_k = 1;
        if (blocks < 128 && a.ktiles >= 32 && workspace) {
            int s = 256 / blocks;
            if (s > a.ktiles / 8) s = a.ktiles / 8;
            if (s > 32) s = 32;
            while (s > 1 && (size_t)s * a.M * a.Npad > workspace_floats) --s;
            if (s > 1) d.split_k = s;
        }
    }
    if (d.split_k > a.ktiles) d.split_k = a.ktiles;
    if (d.split_k > 1) {
        if (!workspace) return EGR_ENULL;
        if ((size_t)d.split_k * a.M * a.Npad > workspace_floats) return EGR_EWORKSPACE;
    }
    a.ktiles_per_split = (a.ktiles + d.split_k - 1) / d.split_k;
    d.split_k = (a.ktiles + a.ktiles_per_split - 1) / a.ktiles_per_split;  // no empty splits

    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (cfg) {
        case CFG_128x128: rc = launch_cfg<128, 128, 2, 2>(a, s); break;
        case CFG_256x64: rc = launch_cfg<256, 64, 4, 1>(a, s); break;
        case CFG_64x64: rc = launch_cfg<64, 64, 2, 2>(a, s); break;
        default: rc = launch_cfg<128, 32, 4, 1>(a, s); break;
    }
    if (rc) return rc;
    if (d.split_k > 1) {
        int64_t total = (int64_t)a.M * d.cout;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        rc = egr_launch_status();
    }
    return rc;
}
