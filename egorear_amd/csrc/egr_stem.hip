// ResNet stem: conv 7x7 / stride 2 / pad 3, 3 -> 64 channels, + BatchNorm(eval) + ReLU.
// Reads the model-contract input (NCHW fp32, 3 channels) and writes NHWC.
//
// K = 3*7*7 = 147 is too ragged for the generic 32-channel-chunk implicit GEMM, so the stem has its
// own kernel: a workgroup stages the input patch of an 8 x 32 output-pixel tile (21 x 69 x 3 floats)
// and the whole 64 x 147 filter bank in LDS once, then runs 74 K=2 steps of v_mfma_f32_32x32x2_f32
// with the A operand gathered straight out of the staged patch (the (ci,kh,kw) -> LDS offset of every
// step is a compile-time constant; the two lane halves take the even / odd k of the step).
#include "egr_common.h"

namespace {

constexpr int TH = 8, TW = 32;           // output tile
constexpr int PH = 2 * TH + 5;           // 21 input rows
constexpr int PW = 2 * TW + 5;           // 69 input cols
constexpr int PWS = 72;                  // padded patch row stride
constexpr int KTOT = 147, KPAD = 148, WS = 149;  // weight row stride 149: conflict-free column reads

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

__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
    __shared__ float s_patch[3 * PH * PWS];
    __shared__ float s_w[64 * WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int bid = blockIdx.x;
    const int grp = blockIdx.y;  // grouped launch (e.g. the two stereo estimators)
    const int tpi = a.tiles_x * a.tiles_y;
    const int n = bid / tpi;
    int t = bid - n * tpi;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const float* img = a.x + grp * a.gx + egr_map(a.xmap, n);
    const float* wpack = a.wpack + grp * 64 * KPAD;
    const float* scale = a.scale + grp * 64;
    const float* shift = a.shift + grp * 64;
    float* y = a.y + (int64_t)grp * a.n * a.ho * a.wo * 64;

    for (int i = tid; i < 3 * PH * PW; i += 256) {
        int ci = i / (PH * PW);
        int r = i - ci * PH * PW;
        int py = r / PW, px = r - py * PW;
        int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) v = img[((int64_t)ci * a.h + iy) * a.w + ix];
        s_patch[ci * PH * PWS + py * PWS + px] = v;
    }
    for (int i = tid; i < 64 * KPAD; i += 256) {
        int co = i / KPAD, k = i - co * KPAD;
        s_w[co * WS + k] = wpack[i];
    }
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // wave w owns tile rows 2w and 2w+1 (fm = 0/1), lane -> column
    const int abase0 = (2 * (2 * wave)) * PWS + 2 * l31;
    const int abase1 = (2 * (2 * wave + 1)) * PWS + 2 * l31;
    const int bbase0 = l31 * WS, bbase1 = (32 + l31) * WS;

#pragma unroll
    for (int s = 0; s < KPAD / 2; ++s) {
        const int ao = half ? patch_off(2 * s + 1) : patch_off(2 * s);
        const int k = 2 * s + half;
        float a0 = s_patch[abase0 + ao], a1 = s_patch[abase1 + ao];
        float b0 = s_w[bbase0 + k], b1 = s_w[bbase1 + k];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int co = j * 32 + l31;
        float sc = scale[co], sh = shift[co];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int oy = oy0 + 2 * wave + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[i][j][r] * sc + sh;
                v = v > 0.f ? v : 0.f;
                y[(((int64_t)n * a.ho + oy) * a.wo + ox) * 64 + co] = v;
            }
        }
    }
}

}  // namespace

extern "C" int egr_stem_conv7x7_f32(const float* x, egr_nmap xmap, int32_t n, int32_t h, int32_t w, const float* wpack,
                                    const float* scale, const float* shift, float* y, int32_t groups, int64_t gx,
                                    void* stream) {
    if (!x || !wpack || !scale || !shift || !y) return EGR_ENULL;
    if (groups <= 0 || groups > 65535) return EGR_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0 || h % (2 * TH) != 0 || w % (2 * TW) != 0 || xmap.n_inner <= 0) return EGR_EINVAL;
    StemArgs a;
    a.x = x; a.xmap = xmap; a.n = n; a.h = h; a.w = w; a.ho = h / 2; a.wo = w / 2;
    a.wpack = wpack; a.scale = scale; a.shift = shift; a.y = y;
    a.tiles_x = a.wo / TW; a.tiles_y = a.ho / TH;
    a.gx = gx;
    int64_t blocks = (int64_t)n * a.tiles_x * a.tiles_y;
    if (blocks >= (1LL << 31)) return EGR_EINVAL;
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, a);
    return egr_launch_status();
}
