// ResNet stem: conv 7x7 / stride 2 / pad 3, 3 -> 64 channels, + BatchNorm(eval) + ReLU.
// Reads the model-contract input (NCHW fp32, 3 channels) and writes NHWC.
//
// K = 3*7*7 = 147 is too ragged for the generic 32-channel-chunk implicit GEMM, so the stem has its
// own kernel: a workgroup stages the input patch of an 8 x 32 output-pixel tile (21 x 69 x 3 floats)
// and the whole 64 x 147 filter bank in LDS once, then runs 74 K=2 steps of v_mfma_f32_32x32x2_f32
// with the A operand gathered straight out of the staged patch (the (ci,kh,kw) -> LDS offset of every
// step is a compile-time constant; the two lane halves take the even / odd k of the step).
#include "egr_common.h"
#include "egr_stem_pool.h"

namespace {

constexpr int TH = 8, TW = 32;           // output tile
constexpr int PH = 2 * TH + 5;           // 21 input rows
constexpr int PW = 2 * TW + 5;           // 69 input cols
constexpr int PWS = 72;                  // padded patch row stride
constexpr int KTOT = 147, KPAD = 148;   // the filter bank is staged k-major: s_w[k][64 channels]

struct StemArgs {
    const float* x;
    egr_nmap xmap;
    int n, h, w, ho, wo;
    const float* wpack;
    const float* scale;
    const float* shift;
    float* y;
    int tiles_x, tiles_y;
    int64_t gx;
};

__device__ __forceinline__ constexpr int patch_off(int k) {
    // k = ci*49 + kh*7 + kw  (natural OIHW flattening); k == 147 is the zero-weight pad column
    int kk = k > 146 ? 146 : k;
    int ci = kk / 49, r = kk % 49;
    return ci * PH * PWS + (r / 7) * PWS + (r % 7);
}

constexpr int PATCH = 3 * PH * PWS;                       // floats per staged patch
constexpr int PLOADS = (3 * PH * PW + 255) / 256;         // patch elements per thread (17)

// Persistent workgroups: the 64 x 147 filter bank is staged once per workgroup, which then walks tiles with stride
// gridDim.x.  The next tile's input patch is fetched into registers before the MFMA loop of the current tile and
// parked in the other LDS buffer afterwards, so its global-load latency hides under 296 MFMAs.
// POOL: MaxPool2d(3, 2, 1) in the epilogue (egr_stem_pool.h) - the (n, h/2, w/2, 64) tensor is never written.
template <bool POOL>
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemArgs a) {
    __shared__ float s_patch[2][PATCH];
    __shared__ __attribute__((aligned(16))) float s_w[KPAD * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int grp = blockIdx.y;  // grouped launch (e.g. the two stereo estimators)
    const int tpi = a.tiles_x * a.tiles_y;
    const int total = a.n * tpi;
    const float* xg = a.x + grp * a.gx;
    const float* wpack = a.wpack + grp * 64 * KPAD;
    const bool raw = a.scale == nullptr;   // training mode: the bare convolution (BatchNorm on batch statistics follows)
    const float* scale = raw ? nullptr : a.scale + grp * 64;
    const float* shift = raw ? nullptr : a.shift + grp * 64;
    float* y = a.y + (int64_t)grp * a.n * (POOL ? (a.ho >> 1) * (a.wo >> 1) : a.ho * a.wo) * 64;

    // per-thread patch slots: element i = tid + 256*u -> (ci, py, px) and its LDS offset (tile independent)
    int p_lds[PLOADS], p_ci[PLOADS], p_py[PLOADS], p_px[PLOADS];
#pragma unroll
    for (int u = 0; u < PLOADS; ++u) {
        const int i = tid + 256 * u;
        const int ci = i / (PH * PW);
        const int r = i - ci * PH * PW;
        p_ci[u] = ci; p_py[u] = r / PW; p_px[u] = r - p_py[u] * PW;
        p_lds[u] = (i < 3 * PH * PW) ? ci * PH * PWS + p_py[u] * PWS + p_px[u] : -1;
    }
    auto fetch = [&](int tile, float (&v)[PLOADS]) {
        const int n = tile / tpi;
        const int t = tile - n * tpi;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int iy0 = 2 * ty * TH - 3, ix0 = 2 * tx * TW - 3;
        const float* img = xg + egr_map(a.xmap, n);
#pragma unroll
        for (int u = 0; u < PLOADS; ++u) {
            const int iy = iy0 + p_py[u], ix = ix0 + p_px[u];
            const bool ok = p_lds[u] >= 0 && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
            v[u] = ok ? img[((int64_t)p_ci[u] * a.h + iy) * a.w + ix] : 0.f;
        }
    };
    auto park = [&](int buf, const float (&v)[PLOADS]) {
#pragma unroll
        for (int u = 0; u < PLOADS; ++u)
            if (p_lds[u] >= 0) s_patch[buf][p_lds[u]] = v[u];
    };

    for (int i = tid; i < 64 * KPAD; i += 256) {
        int co = i / KPAD, k = i - co * KPAD;
        s_w[k * 64 + co] = wpack[i];
    }
    float pv[PLOADS];
    int tile = blockIdx.x;
    if (tile < total) {
        fetch(tile, pv);
        park(0, pv);
    }
    __syncthreads();

    // wave w owns tile rows 2w and 2w+1 (fm = 0/1), lane -> column
    const int abase0 = (2 * (2 * wave)) * PWS + 2 * l31;
    const int abase1 = (2 * (2 * wave + 1)) * PWS + 2 * l31;
    // fragment j covers the channels co = 2*l31 + j: a lane's two B operands are adjacent (one ds_read_b64), and so are
    // its two outputs per pixel (one 8-byte store, 256 contiguous bytes per pixel and half-wave)
    float sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { sc[j] = raw ? 1.f : scale[2 * l31 + j]; sh[j] = raw ? 0.f : shift[2 * l31 + j]; }

    int buf = 0;
    for (; tile < total; tile += gridDim.x, buf ^= 1) {
        const int next = tile + gridDim.x;
        if (next < total) fetch(next, pv);        // in flight during the MFMA loop below
        const float* sp = s_patch[buf];
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KPAD / 2; ++s) {
            const int ao = half ? patch_off(2 * s + 1) : patch_off(2 * s);
            const int k = 2 * s + half;
            float a0 = sp[abase0 + ao], a1 = sp[abase1 + ao];
            const f32x2 bb = *reinterpret_cast<const f32x2*>(&s_w[k * 64 + 2 * l31]);
            const float b0 = bb[0], b1 = bb[1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        const int n = tile / tpi;
        const int t = tile - n * tpi;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        if constexpr (POOL) {
            StemPool<TH / 2> pool;
            pool.reduce(acc, sc, sh, half);
            __syncthreads();                         // every wave is done with the patch: its buffer is the exchange area now
            pool.publish(s_patch[buf], wave, l31, half);
            __syncthreads();
            pool.finish(s_patch[buf], wave, l31, half, y, n, oy0, ox0, a.ho >> 1, a.wo >> 1);
        } else
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = oy0 + 2 * wave + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                f32x2 v;
                v[0] = acc[i][0][r] * sc[0] + sh[0];
                v[1] = acc[i][1][r] * sc[1] + sh[1];
                if (!raw) {
                    v[0] = v[0] > 0.f ? v[0] : 0.f;
                    v[1] = v[1] > 0.f ? v[1] : 0.f;
                }
                *reinterpret_cast<f32x2*>(&y[(((int64_t)n * a.ho + oy) * a.wo + ox) * 64 + 2 * l31]) = v;
            }
        }
        if (next < total) park(buf ^ 1, pv);
        __syncthreads();                           // next patch visible; everybody is done reading the current one
    }
}

// ------------------------------------------------------------------ stem weight gradient
//   dW[co][k] = sum over output pixels of dy[pix][co] * patch(pix)[k],   k = ci*49 + kh*7 + kw (the forward's K order).
// Same tiling and patch staging as the forward; here the pixel is the MFMA reduction index: per k-step of two pixels
//   a[i = co][pix],  b[pix][j = k column]  ->  D[co][k] += dy[pix][co] * patch[pix][k].
// dy is read straight from global memory (every element exactly once, 128 contiguous bytes per half-wave); the B operand
// is gathered from the staged patch with per-lane K offsets.  A wave owns a quarter of the tile's pixels and all
// 2 x 5 output fragments (64 channels x 160 K columns); persistent workgroups accumulate over their tiles, reduce the
// four waves through LDS and write ONE partial slab each; a second kernel sums the slabs in fixed order.
struct StemWgradArgs {
    const float* x;
    egr_nmap xmap;
    int n, h, w, ho, wo;
    const float* dy;
    float* ws;     // [groups][nblk][64][160]
    float* dw;     // [groups][64][147]
    int tiles_x, tiles_y, nblk;
    int64_t gx;
};

__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const StemWgradArgs a) {
    __shared__ float s_patch[2][PATCH];
    __shared__ float s_red[64 * 160];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int grp = blockIdx.y;
    const int tpi = a.tiles_x * a.tiles_y;
    const int total = a.n * tpi;
    const float* xg = a.x + grp * a.gx;
    const float* dyg = a.dy + (int64_t)grp * a.n * a.ho * a.wo * 64;

    // per-thread patch slots, one packed word each: (ci << 16) | (py << 8) | px, -1 past the patch (four separate arrays cost 120 registers
    // beside the 160 accumulators: the kernel spilled)
    int p_geo[PLOADS];
#pragma unroll
    for (int u = 0; u < PLOADS; ++u) {
        const int i = tid + 256 * u;
        const int ci = i / (PH * PW);
        const int r = i - ci * PH * PW;
        const int py = r / PW;
        p_geo[u] = (i < 3 * PH * PW) ? ((ci << 16) | (py << 8) | (r - py * PW)) : -1;
    }
    auto fetch = [&](int tile, float (&v)[PLOADS]) {
        const int n = tile / tpi;
        const int t = tile - n * tpi;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int iy0 = 2 * ty * TH - 3, ix0 = 2 * tx * TW - 3;
        const float* img = xg + egr_map(a.xmap, n);
#pragma unroll
        for (int u = 0; u < PLOADS; ++u) {
            int g = p_geo[u];
            asm volatile("" : "+v"(g));          // (unpacked per use: hoisted out of the tile loop the fields are the 120 registers again)
            const int ci = g >> 16, py = (g >> 8) & 255, px = g & 255;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = g >= 0 && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
            v[u] = ok ? img[((int64_t)ci * a.h + iy) * a.w + ix] : 0.f;
        }
    };
    auto park = [&](int buf, const float (&v)[PLOADS]) {
#pragma unroll
        for (int u = 0; u < PLOADS; ++u) {
            int g = p_geo[u];
            asm volatile("" : "+v"(g));
            if (g >= 0) s_patch[buf][(g >> 16) * PH * PWS + ((g >> 8) & 255) * PWS + (g & 255)] = v[u];
        }
    };
    // K offsets of this lane's five B columns (columns >= 147 point at a valid slot and are dropped at the end)
    int koff[5];
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) koff[kb] = patch_off(min(kb * 32 + l31, 147));

    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float pv[PLOADS];
    int tile = blockIdx.x;
    if (tile < total) {
        fetch(tile, pv);
        park(0, pv);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < total; tile += gridDim.x, buf ^= 1) {
        const int next = tile + gridDim.x;
        if (next < total) fetch(next, pv);
        const float* sp = s_patch[buf];
        const int n = tile / tpi;
        const int t = tile - n * tpi;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        // wave w owns tile rows 2w, 2w+1: 64 pixels = 32 k-steps of two horizontally adjacent pixels (lane half = which one)
        const float* dyt = dyg + (((int64_t)n * a.ho + ty * TH) * a.wo + tx * TW) * 64;
        // (round 5: the output gradients of a group of four steps are requested one group ahead of their MFMAs - the loop used to issue a
        // group's eight loads and wait for them in front of its forty matrix instructions)
        constexpr int SG = 4;
        float av0[SG][2], av1[SG][2];
        auto dy_load = [&](int g, float (&av)[SG][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < SG; ++u) {
                const int s = g * SG + u;
                const int prow = 2 * wave + (s >> 4), pcol = 2 * (s & 15) + half;
                const float* dp = dyt + ((int64_t)prow * a.wo + pcol) * 64;
                av[u][0] = dp[l31];
                av[u][1] = dp[32 + l31];
            }
        };
        auto group = [&](int g, const float (&av)[SG][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < SG; ++u) {
                const int s = g * SG + u;
                const int prow = 2 * wave + (s >> 4), pcol = 2 * (s & 15) + half;
                const int pb = (2 * prow) * PWS + 2 * pcol;
                float b[5];
#pragma unroll
                for (int kb = 0; kb < 5; ++kb) b[kb] = sp[pb + koff[kb]];
#pragma unroll
                for (int kb = 0; kb < 5; ++kb) {
                    acc[0][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][0], b[kb], acc[0][kb], 0, 0, 0);
                    acc[1][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][1], b[kb], acc[1][kb], 0, 0, 0);
                }
            }
        };
        dy_load(0, av0);
#pragma unroll 1
        for (int g = 0; g < 32 / SG; g += 2) {
            dy_load(g + 1, av1);
            group(g, av0);
            if (g + 2 < 32 / SG) dy_load(g + 2, av0);
            group(g + 1, av1);
        }
        if (next < total) park(buf ^ 1, pv);
        __syncthreads();
    }
    // reduce the four waves (fixed order) and write this workgroup's slab: D row = co, col = k
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kb = 0; kb < 5; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const int idx = co * 160 + kb * 32 + l31;
                        s_red[idx] = (w == 0) ? acc[i][kb][r] : s_red[idx] + acc[i][kb][r];
                    }
        }
        __syncthreads();
    }
    float* slab = a.ws + ((int64_t)grp * a.nblk + blockIdx.x) * (64 * 160);
    for (int i = tid; i < 64 * 160; i += 256) slab[i] = s_red[i];
}

__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* ws, int nblk, float* dw) {
    // one thread per (group, co, k < 147); slabs summed in order
    const int grp = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 64 * 147) return;
    const int co = idx / 147, k = idx - co * 147;
    const float* p = ws + (int64_t)grp * nblk * (64 * 160) + co * 160 + k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < nblk; b += 4) {
        s0 += p[(int64_t)b * 10240];
        s1 += p[(int64_t)(b + 1) * 10240];
        s2 += p[(int64_t)(b + 2) * 10240];
        s3 += p[(int64_t)(b + 3) * 10240];
    }
    for (; b < nblk; ++b) s0 += p[(int64_t)b * 10240];
    dw[(int64_t)grp * 64 * 147 + idx] = (s0 + s1) + (s2 + s3);
}

}  // namespace

extern "C" int egr_stem_conv7x7_f32(const float* x, egr_nmap xmap, int32_t n, int32_t h, int32_t w, const float* wpack,
                                    const float* scale, const float* shift, float* y, int32_t groups, int64_t gx,
                                    void* stream) {
    if (!x || !wpack || !y || ((scale == nullptr) != (shift == nullptr))) return EGR_ENULL;   // scale == shift == NULL: raw conv
    if (groups <= 0 || groups > 65535) return EGR_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0 || h % (2 * TH) != 0 || w % (2 * TW) != 0 || xmap.n_inner <= 0) return EGR_EINVAL;
    StemArgs a;
    a.x = x; a.xmap = xmap; a.n = n; a.h = h; a.w = w; a.ho = h / 2; a.wo = w / 2;
    a.wpack = wpack; a.scale = scale; a.shift = shift; a.y = y;
    a.tiles_x = a.wo / TW; a.tiles_y = a.ho / TH;
    a.gx = gx;
    int64_t tiles = (int64_t)n * a.tiles_x * a.tiles_y;
    if (tiles >= (1LL << 31)) return EGR_EINVAL;
    // persistent: two workgroups per CU (74 KiB LDS each) across the 256 CUs, shared by the groups
    int64_t blocks = 512 / groups;
    if (blocks < 1) blocks = 1;
    if (blocks > tiles) blocks = tiles;
    hipLaunchKernelGGL(stem_kernel<false>, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, a);
    return egr_launch_status();
}

extern "C" int egr_stem_conv7x7_pool_f32(const float* x, egr_nmap xmap, int32_t n, int32_t h, int32_t w, const float* wpack,
                                         const float* scale, const float* shift, float* y, int32_t groups, int64_t gx,
                                         void* stream) {
    if (!x || !wpack || !y || !scale || !shift) return EGR_ENULL;
    if (groups <= 0 || groups > 65535) return EGR_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0 || h % (2 * TH) != 0 || w % (2 * TW) != 0 || xmap.n_inner <= 0) return EGR_EINVAL;
    StemArgs a;
    a.x = x; a.xmap = xmap; a.n = n; a.h = h; a.w = w; a.ho = h / 2; a.wo = w / 2;
    a.wpack = wpack; a.scale = scale; a.shift = shift; a.y = y;
    a.tiles_x = a.wo / TW; a.tiles_y = a.ho / TH;
    a.gx = gx;
    const int64_t tiles = (int64_t)n * a.tiles_x * a.tiles_y;
    if (tiles >= (1LL << 31)) return EGR_EINVAL;
    stem_pool_init<TH / 2>(y, (int64_t)groups * n, a.ho / 2, a.wo / 2, (hipStream_t)stream);
    int64_t blocks = 512 / groups;
    if (blocks < 1) blocks = 1;
    if (blocks > tiles) blocks = tiles;
    hipLaunchKernelGGL(stem_kernel<true>, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, a);
    return egr_launch_status();
}

extern "C" int egr_stem_wgrad_f32(const float* x, int32_t n_inner, int64_t stride_inner, int64_t stride_outer, int32_t n, int32_t h,
                                  int32_t w, const float* dy, float* dw, float* workspace, size_t workspace_floats, int32_t groups,
                                  int64_t gx, void* stream) {
    egr_nmap xmap{n_inner, stride_inner, stride_outer};
    if (!x || !dy || !dw || !workspace) return EGR_ENULL;
    if (groups <= 0 || groups > 65535) return EGR_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0 || h % (2 * TH) != 0 || w % (2 * TW) != 0 || xmap.n_inner <= 0) return EGR_EINVAL;
    StemWgradArgs a;
    a.x = x; a.xmap = xmap; a.n = n; a.h = h; a.w = w; a.ho = h / 2; a.wo = w / 2;
    a.dy = dy; a.ws = workspace; a.dw = dw; a.gx = gx;
    a.tiles_x = a.wo / TW; a.tiles_y = a.ho / TH;
    int64_t tiles = (int64_t)n * a.tiles_x * a.tiles_y;
    if (tiles >= (1LL << 31)) return EGR_EINVAL;
    int64_t blocks = 512 / groups;
    if (blocks < 1) blocks = 1;
    if (blocks > tiles) blocks = tiles;
    a.nblk = (int)blocks;
    if ((size_t)groups * blocks * 64 * 160 > workspace_floats) return EGR_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, s, a);
    int rc = egr_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((64 * 147 + 255) / 256, (unsigned)groups), dim3(256), 0, s, workspace, a.nblk, dw);
    return egr_launch_status();
}
