// One joint-query transformer layer behind its deformable sampling, fused into ONE kernel.
//
// Reference ops replaced (per layer, after MSDeformAttn's sampling = egr_msda_gather_f32):
//   value projection of the sampled rows + output_proj + masked_fill(~valid)      models/utils/deform_attn.py:122-168,
//                                                                                 egoposeformer_heatmap_mvf_ex.py:779-796 / egoposeformer_mvf_ex.py:455-478
//   cat over views -> fuse_mlp -> +residual -> LayerNorm                          heatmap_mvf_ex.py:896-911 / mvf_ex.py:555-570
//   q/k/v projections, 15/16-token softmax attention, out_proj, +res, LayerNorm   heatmap_mvf_ex.py:799-817, 913-917; models/utils/transformer.py:36-93
//   FFN (Linear, exact-erf GELU, Linear), +res, LayerNorm                         models/utils/transformer.py:8-33; heatmap_mvf_ex.py:919-922
//   optional tails: the NEXT layer's sampling_offsets / attention_weights Linear (deform_attn.py:122-135), post_norm
//   (heatmap_mvf_ex.py:707 / mvf_ex.py:411), the 3-D regression MLP + anchor (mvf_ex.py:255-262, 412-418)
// i.e. the ~14 launch-latency-bound small launches per layer of round 1 (56 per forward, 1.1 ms at batch 64).
//
// One workgroup (4 waves) per (query set, frame): J = 15 / 16 joint tokens, padded to one 16-row MFMA block; the J*V = 60 / 64
// sampled rows are handled as two 32-row halves.  Every contraction runs on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: a
// k-ordered fma chain, exact in the reference's sense): the activations sit in LDS ([rows][K + 4]), a lane reads 16 bytes of its
// row per 16-deep k block (ds_read_b128) and 16 bytes of its weight row straight from global memory (the weights of a layer are
// 0.9 - 3.4 MB and L2-resident across the frames of a query set: workgroups are mapped so that an XCD serves one query set), and
// the four MFMA k-steps of the block use element t of both - the same k permutation on both operands, so every product is
// summed exactly once.  Weight loads run one 8-block chunk ahead of the MFMAs.  A wave owns 16-column output blocks.
// LayerNorm is the arithmetic of layernorm_kernel (same lane -> channel map, same shuffle tree), the attention core that of
// joint_mha_kernel (one wave per head).
//
// w_packed = 2 (round 4, the shipped path under EGR_W_FORMAT=f16x2): the contractions run in the fp16 scheme of the conv launches
//  instead - half of a workgroup's time was the fp32 matrix pipe itself (64 FLOP / cycle / SIMD: 8 x 32 cycles per
// 32-deep block; the three v_mfma_f32_16x16x32_f16 products take 3 x 16).  Weights: egr_pack_layer_wh2_f32 (per-row power-of-two
// scale, two fp16 planes in fragment order, the descales behind the image).  Activations: every tile that feeds a contraction is
// produced inside this kernel, its producers fold max |v| into an LDS slot (one atomic per wave), the consumer scales by the
// power of two that puts that maximum into [2^14, 2^15) and splits its A fragments on the fly (8 floats of a row -> 2 x 8 fp16).
#include <cstdlib>
#include <type_traits>

#include "egr_common.h"
#include "egr_conv_shared.h"
#include "egr_fisheye.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct LayerArgs {
    egr_layer_desc d;
};

constexpr int PAD = 4;   // LDS row padding (floats)
constexpr int NW = 8;    // waves per workgroup (two per SIMD: one's weight fetches hide under the other's MFMAs)
constexpr int NTH = NW * 64;

__device__ __forceinline__ float gelu_erf(float t) { return 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f)); }

// Column blocks nb = nb0, nb0 + nstep, ... < nb1 of  out[rb*16 .. +16][nb*16 .. +16] = A . W^T, as ONE software pipeline over
// the flattened (column block, 128-deep chunk) sequence: the weights of step s+1 are requested before the MFMAs of step s, also
// across column blocks, so a weight fetch is exposed once per call, not once per block.  NSEG (A, W) segment pairs of depth K
// each: A segment s starts at A + s*a_seg (row stride lda), W segment s at W + s*w_seg (row stride ldw); K % 128 == 0.
// epi(nb, acc) consumes a finished block.
// packed: W is in fragment order (egr_pack_layer_w_f32: [16-column block][128-deep chunk][16-deep k block][lane][4 floats]) - a
// wave's weight load is then 1 KiB contiguous instead of 16 rows x 64 bytes (the layer kernel streams 0.9 - 3.4 MB of weights per
// workgroup: -14 % time); the segments of a multi-segment call must be adjacent in K (w_seg == K), which they are.
// A first weight chunk requested AHEAD of its product (gemm_first_*): the request goes out before the barrier / staging that precedes
// the product, so its latency runs under them instead of behind them.
struct WPre {
    f32x4_t b[8];
};
__device__ __forceinline__ void gemm_first_f32(const float* __restrict__ W, int ldw, int K, int nb0, int lane, bool packed, WPre& pre) {
    const int i = lane & 15, q = lane >> 4;
    if (packed) {
        const float* p = W + (int64_t)nb0 * (K / 128) * 8 * 256 + lane * 4;
#pragma unroll
        for (int u = 0; u < 8; ++u) pre.b[u] = *reinterpret_cast<const f32x4_t*>(p + 256 * u);
    } else {
        const float* p = W + (int64_t)(nb0 * 16 + i) * ldw + 4 * q;
#pragma unroll
        for (int u = 0; u < 8; ++u) pre.b[u] = *reinterpret_cast<const f32x4_t*>(p + 16 * u);
    }
}
// (fp16 scheme: the image's first chunk of column block nb0; K = the matrix' whole depth)
__device__ __forceinline__ void gemm_first_h2(const float* __restrict__ W, int K, int nb0, int lane, WPre& pre) {
    const float* p = W + (int64_t)nb0 * (K / 128) * 8 * 256 + lane * 4;
#pragma unroll
    for (int u = 0; u < 8; ++u) pre.b[u] = *reinterpret_cast<const f32x4_t*>(p + 256 * u);
}

template <int RB, int NSEG, typename Epi>
__device__ __forceinline__ void gemm_cols(const float* __restrict__ A, int lda, int a_seg, const float* __restrict__ W, int ldw, int w_seg, int K,
                                          int nb0, int nstep, int nb1, int lane, bool packed, Epi&& epi, const WPre* pre = nullptr) {
    const int i = lane & 15, q = lane >> 4;
    constexpr int CH = 8;      // 16-deep k blocks per chunk (128 k)
    const int cps = K / (16 * CH);          // chunks per segment
    const int cpb = NSEG * cps;             // chunks per column block
    const int nblk = (nb1 - nb0 + nstep - 1) / nstep;
    if (nblk <= 0) return;
    const int steps = nblk * cpb;
    const float* const arow = A + i * lda + 4 * q;
    const float* const wbase = W + (int64_t)i * ldw + 4 * q;
    f32x4_t b0[CH], b1[CH];
    f32x4_t acc[RB];
    // step -> (block ordinal, chunk in block), advanced incrementally
    int l_blk = 0, l_c = 0;                 // position of the next LOAD
    auto load = [&](f32x4_t (&b)[CH]) {
        const int seg = l_c / cps, cc = l_c - seg * cps;
        if (packed) {
            const float* p = W + ((int64_t)((nb0 + l_blk * nstep) * cpb + l_c) * CH) * 256 + lane * 4;
#pragma unroll
            for (int u = 0; u < CH; ++u) b[u] = *reinterpret_cast<const f32x4_t*>(p + 256 * u);
        } else {
            const float* p = wbase + (int64_t)((nb0 + l_blk * nstep) * 16) * ldw + seg * w_seg + cc * (16 * CH);
#pragma unroll
            for (int u = 0; u < CH; ++u) b[u] = *reinterpret_cast<const f32x4_t*>(p + 16 * u);
        }
        if (++l_c == cpb) { l_c = 0; ++l_blk; }
    };
    int c_blk = 0, c_c = 0;                 // position of the next COMPUTE
    auto compute = [&](const f32x4_t (&b)[CH]) {
        if (c_c == 0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        const int seg = c_c / cps, cc = c_c - seg * cps;
        const float* p = arow + seg * a_seg + cc * (16 * CH);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const f32x4_t af = *reinterpret_cast<const f32x4_t*>(p + rb * 16 * lda + 16 * u);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], b[u][t], acc[rb], 0, 0, 0);
            }
        }
        if (++c_c == cpb) {
            epi(nb0 + c_blk * nstep, acc);
            c_c = 0;
            ++c_blk;
        }
    };
    if (pre) {
#pragma unroll
        for (int u = 0; u < CH; ++u) b0[u] = pre->b[u];
        if (++l_c == cpb) { l_c = 0; ++l_blk; }
    } else {
        load(b0);
    }
    for (int s = 0; s < steps; s += 2) {
        if (s + 1 < steps) load(b1);
        compute(b0);
        if (s + 2 < steps) load(b0);
        if (s + 1 < steps) compute(b1);
    }
}

// The same pipeline in the fp16 scheme: W is an egr_pack_layer_wh2_f32 image ([16-column block][128-deep chunk][32-deep k block]
// [plane h | l][lane][8 fp16] - the eight 1-KiB units of a chunk where the fp32 order has its eight 16-deep blocks), wds the per-row
// descales behind it; sa / inv the tile's pre-scale and its inverse.  Lane (i, q) holds A[row i][32 kb + 8 q ..+7] and
// W[column i][the same k]; products (l,h) (h,l) (h,h) as in the conv kernels.  The accumulators reach epi() descaled.
template <int RB, int NSEG, typename Epi>
__device__ __forceinline__ void gemm_cols_h2(const float* __restrict__ A, int lda, int a_seg, const float* __restrict__ W, const float* __restrict__ wds,
                                             int K, int nb0, int nstep, int nb1, int lane, float sa, float inv, Epi&& epi, const WPre* pre = nullptr) {
    using namespace egrc;
    const int i = lane & 15, q = lane >> 4;
    constexpr int CH = 8;
    const int cps = K / 128, cpb = NSEG * cps;
    const int nblk = (nb1 - nb0 + nstep - 1) / nstep;
    if (nblk <= 0) return;
    const int steps = nblk * cpb;
    const float* const arow = A + i * lda + 8 * q;
    u32x4 b0[CH], b1[CH];
    f32x4_t acc[RB];
    int l_blk = 0, l_c = 0;
    auto load = [&](u32x4 (&b)[CH]) {
        const float* p = W + ((int64_t)((nb0 + l_blk * nstep) * cpb + l_c) * CH) * 256 + lane * 4;
#pragma unroll
        for (int u = 0; u < CH; ++u) b[u] = *reinterpret_cast<const u32x4*>(p + 256 * u);
        if (++l_c == cpb) { l_c = 0; ++l_blk; }
    };
    int c_blk = 0, c_c = 0;
    auto compute = [&](const u32x4 (&b)[CH]) {
        if (c_c == 0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        const int seg = c_c / cps, cc = c_c - seg * cps;
        const float* p = arow + seg * a_seg + cc * 128;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(p + rb * 16 * lda + 32 * kb);
                const f32x4_t x1 = *reinterpret_cast<const f32x4_t*>(p + rb * 16 * lda + 32 * kb + 4);
                unsigned h0, l0, h1, l1, h2, l2, h3, l3;
                split4_f16(x0[0], x0[1], x0[2], x0[3], sa, h0, l0, h1, l1);
                split4_f16(x1[0], x1[1], x1[2], x1[3], sa, h2, l2, h3, l3);
                const u32x4 ah = {h0, h1, h2, h3}, al = {l0, l1, l2, l3};
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, b[2 * kb]), acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b[2 * kb + 1]), acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b[2 * kb]), acc[rb], 0, 0, 0);
            }
        }
        if (++c_c == cpb) {
            const int nb = nb0 + c_blk * nstep;
            const float dsc = wds[nb * 16 + i] * inv;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] *= dsc;
            epi(nb, acc);
            c_c = 0;
            ++c_blk;
        }
    };
    if (pre) {
#pragma unroll
        for (int u = 0; u < CH; ++u) b0[u] = __builtin_bit_cast(u32x4, pre->b[u]);
        if (++l_c == cpb) { l_c = 0; ++l_blk; }
    } else {
        load(b0);
    }
    for (int s = 0; s < steps; s += 2) {
        if (s + 1 < steps) load(b1);
        compute(b0);
        if (s + 2 < steps) load(b0);
        if (s + 1 < steps) compute(b1);
    }
}

// abs-max slot of a tile (LDS, float bits): producers fold their values in, the consumer derives the power-of-two pre-scale
__device__ __forceinline__ void tile_track(unsigned* slot, float m) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) __hip_atomic_fetch_max(slot, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void tile_prescale(unsigned bits, float& sa, float& inv) {
    const int e = (int)(__builtin_amdgcn_readfirstlane(bits) >> 23);
    int k = 141 - e;
    k = k > 60 ? 60 : (k < -60 ? -60 : k);
    sa = __uint_as_float((unsigned)(127 + k) << 23);
    inv = __uint_as_float((unsigned)(127 - k) << 23);
}

// rows of a 16-row tile owned by this wave: y = LayerNorm(t [+ res]) * gamma + beta, lane -> channels i*64 + lane (layernorm_kernel)
template <int VPL, int NW>
__device__ __forceinline__ void ln_rows(const float* t, int ldt, const float* res, int ldr, const float* gamma, const float* beta, float eps,
                                        float* y, int ldy, int wave, int lane, unsigned* slot = nullptr) {
    constexpr int c = VPL * 64;
    float amx = 0.f;
    for (int rr = 0; rr < 16 / NW; ++rr) {
        const int row = wave * (16 / NW) + rr;
        float v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int ch = i * 64 + lane;
            float x = t[row * ldt + ch];
            if (res) x += res[row * ldr + ch];
            v[i] = x;
            s += x;
        }
        const float mean = wave_sum(s) / (float)c;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float dlt = v[i] - mean;
            qq += dlt * dlt;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(qq) / (float)c + eps);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int ch = i * 64 + lane;
            const float o = (v[i] - mean) * rstd * gamma[ch] + beta[ch];
            y[row * ldy + ch] = o;
            amx = fmaxf(amx, fabsf(o));
        }
    }
    if (slot) tile_track(slot, amx);
}

// ---- tail: the refiner's head offset (heatmap_mvf_ex.py:707-711 + TransformerHeadLayer.head[0..2]): the J x (s*s) token
// matrix IS an s x s image with the joints as channels (pixel p = channel p of xn); Conv2d(J, HN, 1) + ReLU on it (the fma chain of
// linear_smallk_kernel: k left to right, then the bias), then Upsample(x2, bilinear, align_corners) with upsample2x_kernel's
// arithmetic, straight to (2s, 2s, HN) NHWC - the launches egr_tokens_to_nhwc_f32, egr_linear_smallk_f32, egr_upsample2x_nhwc_f32
template <int LC>
__device__ __forceinline__ void head_offset_tail(const egr_layer_desc& d, int grp, int fb, int B, int J, const float* tile0, float* bufO, float* bufG,
                                                 int tid, int lane) {
    constexpr int C = 256;
    constexpr int S = 16, HN = 64;
    float* const h0 = bufO;                       // [C pixels][HN]
    float* const wT = bufG;                       // [J][HN] + bias [HN]
    const float* const w_h0 = d.w_h0 + (int64_t)grp * HN * J;
    for (int idx = tid; idx < HN * J; idx += NTH) {
        const int n = idx / J, jj = idx - n * J;
        wT[jj * HN + n] = w_h0[idx];
    }
    if (tid < HN) wT[J * HN + tid] = d.b_h0[grp * HN + tid];
    __syncthreads();
    float amx = 0.f;
    {
        // lane = output channel (its filter row in registers), a wave walks pixel quads: one broadcast 16-byte LDS read per
        // (joint, quad) instead of two 4-byte reads per multiply-add
        const int n = tid & (HN - 1);
        const float bb = wT[J * HN + n];
        float wr[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) wr[jj] = jj < J ? wT[jj * HN + n] : 0.f;
        for (int p0 = 4 * (tid / HN); p0 < C; p0 += 4 * (NTH / HN)) {
            float sacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < 16; ++jj)
                if (jj < J) {
                    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(tile0 + jj * LC + p0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sacc[i] = fmaf(t[i], wr[jj], sacc[i]);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = sacc[i] + bb;
                v = v > 0.f ? v : 0.f;
                h0[(p0 + i) * HN + n] = v;
                amx = fmaxf(amx, v);
            }
        }
    }
    if (d.amax_h0) {        // an upper bound of |h0_out| (interpolation weights are non-negative and sum to 1): one atomic per wave
        amx = wave_max(amx);
        if (lane == 0 && amx > 0.f) atomicMax(d.amax_h0 + (blockIdx.x & 63), __float_as_uint(amx));
    }
    __syncthreads();
    constexpr int SO = 2 * S;
    const float sc = (float)(S - 1) / (float)(SO - 1);
    const int cq = tid & (HN / 4 - 1), ox = tid / (HN / 4);          // 16 channel quads x 32 columns = 512 threads
    static_assert(NTH == (HN / 4) * SO, "one thread per (column, channel quad)");
    const float fx = sc * (float)ox;
    const int x0 = (int)fx, x1 = min(x0 + 1, S - 1);
    const float lx1 = fminf(fmaxf(fx - (float)x0, 0.f), 1.f), lx0 = 1.f - lx1;
    float* const out = d.h0_out + ((int64_t)grp * B + fb) * SO * SO * HN;
    for (int oy = 0; oy < SO; ++oy) {
        const float fy = sc * (float)oy;
        const int y0 = (int)fy, y1 = min(y0 + 1, S - 1);
        const float ly1 = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), ly0 = 1.f - ly1;
        const f32x4_t v00 = *reinterpret_cast<const f32x4_t*>(h0 + (y0 * S + x0) * HN + cq * 4);
        const f32x4_t v01 = *reinterpret_cast<const f32x4_t*>(h0 + (y0 * S + x1) * HN + cq * 4);
        const f32x4_t v10 = *reinterpret_cast<const f32x4_t*>(h0 + (y1 * S + x0) * HN + cq * 4);
        const f32x4_t v11 = *reinterpret_cast<const f32x4_t*>(h0 + (y1 * S + x1) * HN + cq * 4);
        f32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = ly0 * (lx0 * v00[i] + lx1 * v01[i]) + ly1 * (lx0 * v10[i] + lx1 * v11[i]);
        *reinterpret_cast<f32x4_t*>(out + ((int64_t)oy * SO + ox) * HN + cq * 4) = o;
    }
}

#ifdef LAYER_STAMPS      // diagnostic build (tools/probes/layer_stamps.py): s_memtime of workgroup 0 at the phase boundaries
__device__ unsigned long long g_layer_stamps[64];
#define LSTAMP(n) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_layer_stamps[n] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LSTAMP(n) do { } while (0)
#endif

// (query set, frame) of a workgroup: blocks b and b + 8 share an XCD - keep a query set's frames (same weights) on the same XCDs
__device__ __forceinline__ void wg_group_frame(int bid, int G, int B, int& grp, int& fb) {
    if (G > 1 && (8 % G) == 0 && ((G * B) % 8) == 0) {
        const int per = 8 / G;
        grp = (bid & 7) / per;
        fb = (bid >> 3) * per + ((bid & 7) % per);
    } else {
        grp = bid / B;
        fb = bid - grp * B;
    }
}


template <int C>
struct LayerLds {
    static constexpr int HEADS = 4, DH = C / HEADS, CF = 128, FF = 512;
    static constexpr int LC = C + PAD, LG = CF + PAD, LQ = 3 * C + PAD, LF = FF + PAD;
    static constexpr int T = 16 * LC;                         // one 16-row tile of C-wide activations
    static constexpr int PW = (NW * 16 < C) ? NW * 16 : C;    // output columns of one value-projection pass
    static constexpr int HPP = PW / DH;                       // heads staged per pass
    static constexpr int BUFA = 2 * T;                        // a_half (32 rows) | later tile0 / tile1
    static constexpr int M1 = 64 * LC, M2 = 16 * LQ, M3 = 16 * LF;
    static constexpr int BUFO = (M1 > M2 ? (M1 > M3 ? M1 : M3) : (M2 > M3 ? M2 : M3));
    static constexpr int BUFG = (HPP * 32 * LG > T ? HPP * 32 * LG : T);
    static constexpr int SIDE = 64 * HEADS + 64 + 16;         // sigma of the 64 sampled rows per head, their row mask, the tiles' abs-max slots
    static constexpr size_t BYTES = sizeof(float) * (size_t)(BUFA + BUFO + BUFG + SIDE);
};

template <int C, bool H2>
__global__ __launch_bounds__(NTH) void joint_layer_kernel(const LayerArgs a) {
    const egr_layer_desc& d = a.d;
    using L = LayerLds<C>;
    constexpr int HEADS = L::HEADS, DH = L::DH, CF = L::CF, FF = L::FF, VPL = C / 64;
    constexpr int LC = L::LC, LG = L::LG, LQ = L::LQ, LF = L::LF, T = L::T, PW = L::PW, HPP = L::HPP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const bufA = lds;
    float* const bufO = bufA + L::BUFA;
    float* const bufG = bufO + L::BUFO;
    float* const tile0 = bufA;
    float* const tile1 = bufA + T;
    float* const s_sig = bufG + L::BUFG;               // [HEADS][64]
    float* const s_keep = s_sig + 64 * HEADS;          // [64]: 1 = valid sampled row
    // fp16 scheme: abs-max (float bits) of every tile that feeds a contraction - slots 0-3 the staged sampled features per (half,
    // pass), 4-5 the value projection per half, 6 output_proj, 7 norm_cross, 8 attention, 9 norm_spatial, 10 FFN hidden, 11 norm_ffn,
    // 12 post_norm
    unsigned* const s_amax = reinterpret_cast<unsigned*>(s_keep + 64);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q4 = lane >> 4;
    const int B = d.B, J = d.J, V = d.V, G = d.groups;
    // (query set, frame) of this workgroup: blocks b and b + 8 share an XCD - keep a query set's frames (same weights) on the
    // same XCDs when the sets divide the eight evenly
    int grp, fb;
    {
        const int bid = blockIdx.x;
        if (G > 1 && (8 % G) == 0 && ((G * B) % 8) == 0) {
            const int per = 8 / G;
            grp = (bid & 7) / per;
            fb = (bid >> 3) * per + ((bid & 7) % per);
        } else {
            grp = bid / B;
            fb = bid - grp * B;
        }
    }
    const int rows_all = B * J * V;                    // sampled rows per query set
    const int row0 = fb * J * V;                       // first sampled row of this frame
    const int nrow = J * V;                            // valid sampled rows (60 / 64)
    const int64_t xrow0 = ((int64_t)grp * B + fb) * J; // first token row
    const float* const gq = d.g + (int64_t)grp * rows_all * HEADS * CF;
    const float* const eq = d.e ? d.e + (int64_t)grp * rows_all * C : nullptr;
    const float* const sg = d.sigma + (int64_t)grp * HEADS * rows_all;
    // a matrix of `rows` x k: rows * k floats; in the fp16 scheme the same bytes as two fp16 planes, the rows' descales behind them
    auto wsz = [](int rows, int k) { return (int64_t)rows * k + (H2 ? rows : 0); };
    const float* const w_fold = d.w_fold + grp * wsz(C, CF);
    const float* const c_fold = d.c_fold + grp * C;
    const float* const w_out = d.w_out + grp * wsz(C, C);
    const float* const b_out = d.b_out + grp * C;
    const float* const w_fuse = d.w_fuse + grp * wsz(C, 4 * C);
    const float* const b_fuse = d.b_fuse + grp * C;
    const float* const w_qkv = d.w_qkv + grp * wsz(3 * C, C);
    const float* const b_qkv = d.b_qkv + grp * 3 * C;
    const float* const w_mo = d.w_mo + grp * wsz(C, C);
    const float* const b_mo = d.b_mo + grp * C;
    const float* const w_f0 = d.w_f0 + grp * wsz(FF, C);
    const float* const b_f0 = d.b_f0 + grp * FF;
    const float* const w_f1 = d.w_f1 + grp * wsz(C, FF);
    const float* const b_f1 = d.b_f1 + grp * C;

    const bool wpk = d.w_packed != 0;
    LSTAMP(0);
    // per-row side inputs of the sampled-row phase, once (they were global loads inside the GEMM epilogues)
    if (tid < 64 * HEADS) {
        const int h = tid >> 6, rl = tid & 63;
        s_sig[tid] = rl < nrow ? sg[(int64_t)h * rows_all + row0 + rl] : 0.f;
    }
    if (tid < 64) s_keep[tid] = (tid < nrow && d.rowmask[row0 + tid] != 0) ? 1.f : 0.f;
    if constexpr (H2) {
        if (tid < 16) s_amax[tid] = 0u;
        __syncthreads();                                // (the first producers' atomics must find the zeros)
    }
    // one contraction in either arithmetic: `rows` x `ktot` is the whole weight matrix (its descales sit behind the image), `slot` the
    // abs-max slot of the A tile
    auto gemm = [&](auto rb_tag, auto nseg_tag, const float* A, int lda, int a_seg, const float* W, int rows, int ktot, int ldw, int w_seg, int K,
                    int nb0, int nstep, int nb1, int slot, auto&& epi, const WPre* pre = nullptr) __attribute__((always_inline)) {
        constexpr int RB = decltype(rb_tag)::value, NSEG = decltype(nseg_tag)::value;
        if constexpr (H2) {
            float sa, inv;
            tile_prescale(s_amax[slot], sa, inv);
            gemm_cols_h2<RB, NSEG>(A, lda, a_seg, W, W + (int64_t)rows * ktot, K, nb0, nstep, nb1, lane, sa, inv, epi, pre);
        } else {
            gemm_cols<RB, NSEG>(A, lda, a_seg, W, ldw, w_seg, K, nb0, nstep, nb1, lane, wpk, epi, pre);
        }
    };
    // the first weight chunk of column block nb0 of a (rows x ktot) matrix, requested ahead (WPre)
    auto first = [&](const float* W, int ktot, int nb0, WPre& pre) __attribute__((always_inline)) {
        if constexpr (H2) gemm_first_h2(W, ktot, nb0, lane, pre);
        else gemm_first_f32(W, ktot, ktot, nb0, lane, wpk, pre);
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    // plain "Linear on a 16-row tile": out[16][ldo] = act(A . W^T + bias)
    auto linear16 = [&](const float* A, int lda, const float* W, int K, const float* bias, int N, float* out, int ldo, bool gelu, int slot_in,
                        int slot_out) {
        float amx = 0.f;
        gemm(I1{}, I1{}, A, lda, 0, W, N, K, K, 0, K, wave, NW, N / 16, slot_in, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = bias[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[0][r] + bb;
                v = gelu ? gelu_erf(v) : v;
                out[(4 * q4 + r) * ldo + col] = v;
                amx = fmaxf(amx, fabsf(v));
            }
        });
        if (H2 && slot_out >= 0) tile_track(s_amax + slot_out, amx);
    };

    // ---- value projection of the sampled rows (sample-then-project) + output_proj, in two 32-row halves.
    // Round 5: software-pipelined over the (half, pass) sequence - every global request of a pass (its weight chunk, its positional
    // terms, the NEXT pass's sampled features) is on its way before the barrier in front of the product instead of being waited for
    // one after the other behind it (per pass: three exposed round trips -> one).
    {
        constexpr int NPS = C / PW;                    // passes per half: PW output columns each, one 16-column block per wave
        constexpr int NST = HPP * 32 * (CF / 4) / NTH;
        static_assert(HPP * 32 * (CF / 4) % NTH == 0, "staging units per thread");
        f32x4_t stg[NST];
        // the sampled features of the heads behind a pass' columns: bufG[hp][32][LG] (all of a thread's loads first, then the LDS writes)
        auto stage_load = [&](int hf, int ps) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int idx = tid + it * NTH;
                const int cq = idx % (CF / 4), r = (idx / (CF / 4)) % 32, hp = idx / (32 * (CF / 4));
                const int rl = hf * 32 + r, h = ps * HPP + hp;
                stg[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (rl < nrow) stg[it] = *reinterpret_cast<const f32x4_t*>(gq + ((int64_t)(row0 + rl) * HEADS + h) * CF + cq * 4);
            }
        };
        stage_load(0, 0);
        for (int hf = 0; hf < 2; ++hf) {
            WPre pre_out;
            for (int ps = 0; ps < NPS; ++ps) {
                float samx = 0.f;
#pragma unroll
                for (int it = 0; it < NST; ++it) {
                    const int idx = tid + it * NTH;
                    const int cq = idx % (CF / 4), r = (idx / (CF / 4)) % 32, hp = idx / (32 * (CF / 4));
                    *reinterpret_cast<f32x4_t*>(bufG + (hp * 32 + r) * LG + cq * 4) = stg[it];
                    samx = fmaxf(fmaxf(samx, fmaxf(fabsf(stg[it][0]), fabsf(stg[it][1]))), fmaxf(fabsf(stg[it][2]), fabsf(stg[it][3])));
                }
                if constexpr (H2) tile_track(s_amax + hf * 2 + ps, samx);
                const bool mine = wave * 16 < PW;
                const int n0 = ps * PW + wave * 16;    // this wave's output columns [n0, n0 + 16)
                const int h = n0 / DH, hp = h - ps * HPP;
                WPre pre;
                float ev[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (mine) {
                    first(w_fold, CF, n0 / 16, pre);
                    if (eq) {
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int rl = hf * 32 + rb * 16 + 4 * q4 + r;
                                if (rl < nrow) ev[rb][r] = eq[(int64_t)(row0 + rl) * C + n0 + i16];
                            }
                    }
                }
                __syncthreads();
                // the next pass's sampled features travel under this pass's product (the next HALF's first pass is requested behind
                // output_proj instead: in flight across it they cost registers the fp16 variant does not have)
                if (ps + 1 < NPS) stage_load(hf, ps + 1);
                if (mine) {
                    float amx = 0.f;
                    gemm(I2{}, I1{}, bufG + hp * 32 * LG, LG, 0, w_fold, C, CF, CF, 0, CF, n0 / 16, 1, n0 / 16 + 1, hf * 2 + ps, [&](int nb, const f32x4_t (&acc)[2]) {
                        const int col = nb * 16 + i16;
                        const float cf_ = c_fold[col];
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int rl32 = rb * 16 + 4 * q4 + r, rl = hf * 32 + rl32;
                                float v = 0.f;
                                if (rl < nrow) {
                                    v = acc[rb][r] + cf_ * s_sig[h * 64 + rl];
                                    if (eq) v += ev[rb][r];
                                }
                                bufA[rl32 * LC + col] = v;
                                amx = fmaxf(amx, fabsf(v));
                            }
                    }, &pre);
                    if constexpr (H2) tile_track(s_amax + 4 + hf, amx);
                }
                if (ps == NPS - 1) first(w_out, C, wave, pre_out);     // output_proj's first chunk under the barrier
                __syncthreads();
            }
            // output_proj on the 32 rows of this half; masked_fill(~valid) AFTER it (the bias is zeroed too)
            {
                float amx = 0.f;
                gemm(I2{}, I1{}, bufA, LC, 0, w_out, C, C, C, 0, C, wave, NW, C / 16, 4 + hf, [&](int nb, const f32x4_t (&acc)[2]) {
                    const int col = nb * 16 + i16;
                    const float bo = b_out[col];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int rl = hf * 32 + rb * 16 + 4 * q4 + r;
                            const bool keep = s_keep[rl] != 0.f;
                            const float v = keep ? acc[rb][r] + bo : 0.f;
                            bufO[rl * LC + col] = v;
                            amx = fmaxf(amx, fabsf(v));
                        }
                }, &pre_out);
                if constexpr (H2) tile_track(s_amax + 6, amx);
            }
            if (hf == 0) stage_load(1, 0);
            __syncthreads();
            LSTAMP(16 + hf);
        }
    }
    LSTAMP(1);
    // ---- cat over views -> fuse_mlp: token j = rows 4j .. 4j+3 of bufO side by side (V segments of K = C)
    gemm(I1{}, I4{}, bufO, V * LC, LC, w_fuse, C, 4 * C, V * C, C, C, wave, NW, C / 16, 6, [&](int nb, const f32x4_t (&acc)[1]) {
        const int col = nb * 16 + i16;
        const float bb = b_fuse[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) tile0[(4 * q4 + r) * LC + col] = acc[0][r] + bb;
    });
    LSTAMP(2);
    // ---- + residual (the layer input, from global memory) -> norm_cross
    // residual rows of tokens >= J do not exist: point them at token 0 (their results are never stored)
    {
        constexpr int NX4 = 16 * C / 4 / NTH;       // float4 per thread (1 or 2)
        static_assert(16 * C / 4 % NTH == 0, "residual tile");
        f32x4_t xr[NX4];
#pragma unroll
        for (int it = 0; it < NX4; ++it) {
            const int idx = tid + it * NTH, r = idx / (C / 4), c4 = idx - r * (C / 4);
            xr[it] = *reinterpret_cast<const f32x4_t*>(d.x + (xrow0 + (r < J ? r : 0)) * C + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < NX4; ++it) {
            const int idx = tid + it * NTH, r = idx / (C / 4), c4 = idx - r * (C / 4);
            *reinterpret_cast<f32x4_t*>(bufG + r * LC + c4 * 4) = xr[it];
        }
    }
    __syncthreads();
    ln_rows<VPL, NW>(tile0, LC, bufG, LC, d.ln1_g + grp * C, d.ln1_b + grp * C, d.eps, tile1, LC, wave, lane, H2 ? s_amax + 7 : nullptr);
    __syncthreads();
    LSTAMP(3);
    // ---- q/k/v projections -> bufO [16][3C]
    linear16(tile1, LC, w_qkv, C, b_qkv, 3 * C, bufO, LQ, false, 7, -1);
    __syncthreads();
    LSTAMP(4);
    // ---- joint-to-joint attention (joint_mha_kernel's arithmetic): scores by waves 0-3 (one per head), PV by all waves -> tile0
    if (wave < HEADS) {
        const int h = wave;
        float* const sp = bufG + h * 256;             // this head's 16 x 16 probabilities
        const float scale = d.mha_scale;
        const int i = lane >> 2, gq_ = lane & 3;
        float s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int jj = gq_ + 4 * t;
            float dot = 0.f;
            if (i < J && jj < J)
                for (int dd = 0; dd < DH; ++dd) dot = fmaf(bufO[i * LQ + h * DH + dd], bufO[jj * LQ + C + h * DH + dd], dot);
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
        for (int t = 0; t < 4; ++t) sp[i * 16 + gq_ + 4 * t] = (i < J) ? s[t] / sum : 0.f;
    }
    __syncthreads();
    {
        const int h = wave & (HEADS - 1), part = wave / HEADS;      // NW / HEADS waves share a head's 16 x DH outputs
        const float* const sp = bufG + h * 256;
        float amx = 0.f;
        for (int idx = part * 64 + lane; idx < 16 * DH; idx += 64 * (NW / HEADS)) {
            const int t = idx / DH, dd = idx - t * DH;
            float o = 0.f;
            for (int jj = 0; jj < J; ++jj) o = fmaf(sp[t * 16 + jj], bufO[jj * LQ + 2 * C + h * DH + dd], o);
            tile0[t * LC + h * DH + dd] = o;
            amx = fmaxf(amx, fabsf(o));
        }
        if constexpr (H2) tile_track(s_amax + 8, amx);
    }
    __syncthreads();
    LSTAMP(5);
    // ---- out_proj -> bufG, + residual (tile1) -> norm_spatial -> tile0
    linear16(tile0, LC, w_mo, C, b_mo, C, bufG, LC, false, 8, -1);
    __syncthreads();
    ln_rows<VPL, NW>(bufG, LC, tile1, LC, d.ln2_g + grp * C, d.ln2_b + grp * C, d.eps, tile0, LC, wave, lane, H2 ? s_amax + 9 : nullptr);
    __syncthreads();
    LSTAMP(6);
    // ---- FFN: Linear + GELU -> bufO [16][512]; Linear -> bufG; + residual (tile0) -> norm_ffn -> tile1
    linear16(tile0, LC, w_f0, C, b_f0, FF, bufO, LF, true, 9, 10);
    __syncthreads();
    linear16(bufO, LF, w_f1, FF, b_f1, C, bufG, LC, false, 10, -1);
    __syncthreads();
    ln_rows<VPL, NW>(bufG, LC, tile0, LC, d.ln3_g + grp * C, d.ln3_b + grp * C, d.eps, tile1, LC, wave, lane, H2 ? s_amax + 11 : nullptr);
    __syncthreads();
    LSTAMP(7);
    // ---- the layer's output tokens
    for (int idx = tid; idx < J * C; idx += NTH) {
        const int r = idx / C, ch = idx - r * C;
        d.x_out[(xrow0 + r) * C + ch] = tile1[r * LC + ch];
    }
    // ---- tail: the next layer's sampling offsets / attention logits from these tokens (straight to global memory)
    if (d.w_ol) {
        const float* const w_ol = d.w_ol + grp * wsz(d.ol_n, C);
        const float* const b_ol = d.b_ol + grp * d.ol_n;
        const int oln = d.ol_n;
        gemm(I1{}, I1{}, tile1, LC, 0, w_ol, oln, C, C, 0, C, wave, NW, oln / 16, 11, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = b_ol[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q4 + r;
                if (row < J) d.ol_out[(xrow0 + row) * oln + col] = acc[0][r] + bb;
            }
        });
    }
    LSTAMP(8);
    // ---- tail: post_norm [+ regression MLP + anchor]
    if (d.lnp_g) {
        ln_rows<VPL, NW>(tile1, LC, nullptr, 0, d.lnp_g + grp * C, d.lnp_b + grp * C, d.eps, tile0, LC, wave, lane, H2 ? s_amax + 12 : nullptr);
        __syncthreads();
        if (d.xn_out)
            for (int idx = tid; idx < J * C; idx += NTH) {
                const int r = idx / C, ch = idx - r * C;
                d.xn_out[(xrow0 + r) * C + ch] = tile0[r * LC + ch];
            }
        if constexpr (C == 256) {
            if (d.w_h0) head_offset_tail<LC>(d, grp, fb, B, J, tile0, bufO, bufG, tid, lane);
        }
        if (d.w_r0) {
            linear16(tile0, LC, d.w_r0 + grp * wsz(C, C), C, d.b_r0 + grp * C, C, bufG, LC, true, 12, -1);
            __syncthreads();
            // reg_mlp[2]: C -> 3, + init_anchors_3d; one thread per (token, coordinate), sequential k like the GEMM's chain
            if (tid < J * 3) {
                const int r = tid / 3, o = tid - r * 3;
                const float* w = d.w_r2 + ((int64_t)grp * 3 + o) * C;
                float s = 0.f;
                for (int k = 0; k < C; ++k) s = fmaf(bufG[r * LC + k], w[k], s);
                s += d.b_r2[grp * 3 + o];
                d.pred_out[(xrow0 + r) * 3 + o] = s + d.anchors3d[(xrow0 + r) * 3 + o];
            }
        }
    }
    LSTAMP(9);
}

// ------------------------------------------------------------------ the layer with its A tiles split ONCE (round 5)
// joint_layer_kernel<C, true> splits a row's 8 floats into their fp16 planes inside the product loop - every one of the eight waves
// does that for the SAME 16 (or 32) rows, and the conversion (2 x 10 vector instructions per 3 matrix instructions) is what a
// product step costs.  Here a tile that feeds a contraction is split once, by all threads, into fragment-ordered planes in a
// dedicated LDS region ([128-deep chunk][32-deep block][row block][plane h | l][lane][8 fp16]); the product loop is two 16-byte
// LDS reads and three MFMAs per block.  Same scales, same products in the same order: bit-identical to joint_layer_kernel<C, true>
// (tested).  The first weight chunk of every product is requested before the barriers in front of it.
template <int C>
struct PlaneLds {
    static constexpr int HEADS = 4, DH = C / HEADS, CF = 128, FF = 512;
    static constexpr int LC = C + PAD, LQ = 3 * C + PAD, LF = FF + PAD;
    static constexpr int T = 16 * LC;
    static constexpr int PW = (NW * 16 < C) ? NW * 16 : C;
    static constexpr int HPP = PW / DH;
    static constexpr int BUFA = 2 * T;
    static constexpr int M1 = 64 * LC, M2 = 16 * LQ, M3 = 16 * LF;
    static constexpr int BUFO = (M1 > M2 ? (M1 > M3 ? M1 : M3) : (M2 > M3 ? M2 : M3));
    static constexpr int BUFG = T > 2048 ? T : 2048;           // x residual / small products / attention probabilities / head filter
    // plane region (bytes): the sampled rows of a pass (HPP heads x 32 rows x 128), output_proj's 32 x C, the FFN's 16 x 512; fuse_mlp's
    // 16 x 4C takes the region and, where that is too small (C = 256), the idle bufA behind it
    static constexpr int PV = HPP * 16384, PO = 32 * C * 4, PF = 16 * FF * 4;
    static constexpr int PBYTES = (PV > PO ? (PV > PF ? PV : PF) : (PO > PF ? PO : PF));
    static constexpr int FUSE_CH = 4 * C / 128;                // chunks of fuse_mlp's A
    static constexpr int FUSE_IN_P = (PBYTES / 8192 < FUSE_CH) ? PBYTES / 8192 : FUSE_CH;
    static_assert((FUSE_CH - FUSE_IN_P) * 8192 <= BUFA * 4, "fuse_mlp planes: region + bufA");
    static constexpr int SIDE = 64 * HEADS + 64 + 16;
    static constexpr size_t BYTES = sizeof(float) * (size_t)(BUFA + BUFO + BUFG + SIDE) + PBYTES;
};

// 16 RB rows x K of fp32 (element (row, k) at src(row, k), 8 consecutive k contiguous) -> planes: chunk c < nsplit in P0, the rest in P1
template <int RB, typename Src>
__device__ __forceinline__ void split_tile(Src&& src, int K, char* P0, char* P1, int nsplit, float sa, int tid) {
    const int oct = K >> 3, units = RB * 16 * oct;
    constexpr int CHB = 8192 * RB;
    for (int u = tid; u < units; u += NTH) {
        const int i = u & 15, t = u >> 4, o = t % oct, rb = t / oct;
        const float* p = src(rb * 16 + i, o * 8);
        const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(p), x1 = *reinterpret_cast<const f32x4_t*>(p + 4);
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        egrc::split4_f16(x0[0], x0[1], x0[2], x0[3], sa, h0, l0, h1, l1);
        egrc::split4_f16(x1[0], x1[1], x1[2], x1[3], sa, h2, l2, h3, l3);
        const int c = o >> 4, kb = (o >> 2) & 3, q = o & 3;
        char* d = (c < nsplit ? P0 + c * CHB : P1 + (c - nsplit) * CHB) + ((kb * RB + rb) * 2) * 1024 + (16 * q + i) * 16;
        *reinterpret_cast<egrc::u32x4*>(d) = egrc::u32x4{h0, h1, h2, h3};
        *reinterpret_cast<egrc::u32x4*>(d + 1024) = egrc::u32x4{l0, l1, l2, l3};
    }
}

// gemm_cols_h2's pipeline with the A fragments read from planes (chunk c of a column block: P0 + c CHB for c < nsplit, else P1)
template <int RB, typename Epi>
__device__ __forceinline__ void gemm_planes(const char* P0, const char* P1, int nsplit, const float* __restrict__ W, const float* __restrict__ wds,
                                            int cpb, int nb0, int nstep, int nb1, int lane, float inv, Epi&& epi, const WPre* pre = nullptr) {
    using namespace egrc;
    constexpr int CH = 8, CHB = 8192 * RB;
    const int i = lane & 15;
    const int nblk = (nb1 - nb0 + nstep - 1) / nstep;
    if (nblk <= 0) return;
    const int steps = nblk * cpb;
    u32x4 b0[CH], b1[CH];
    f32x4_t acc[RB];
    int l_blk = 0, l_c = 0;
    auto load = [&](u32x4 (&b)[CH]) {
        const float* p = W + ((int64_t)((nb0 + l_blk * nstep) * cpb + l_c) * CH) * 256 + lane * 4;
#pragma unroll
        for (int u = 0; u < CH; ++u) b[u] = *reinterpret_cast<const u32x4*>(p + 256 * u);
        if (++l_c == cpb) { l_c = 0; ++l_blk; }
    };
    int c_blk = 0, c_c = 0;
    auto compute = [&](const u32x4 (&b)[CH]) {
        if (c_c == 0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        const char* p = (c_c < nsplit ? P0 + c_c * CHB : P1 + (c_c - nsplit) * CHB) + lane * 16;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const u32x4 ah = *reinterpret_cast<const u32x4*>(p + ((kb * RB + rb) * 2) * 1024);
                const u32x4 al = *reinterpret_cast<const u32x4*>(p + ((kb * RB + rb) * 2 + 1) * 1024);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, b[2 * kb]), acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b[2 * kb + 1]), acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b[2 * kb]), acc[rb], 0, 0, 0);
            }
        }
        if (++c_c == cpb) {
            const int nb = nb0 + c_blk * nstep;
            const float dsc = wds[nb * 16 + i] * inv;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] *= dsc;
            epi(nb, acc);
            c_c = 0;
            ++c_blk;
        }
    };
    if (pre) {
#pragma unroll
        for (int u = 0; u < CH; ++u) b0[u] = __builtin_bit_cast(u32x4, pre->b[u]);
        if (++l_c == cpb) { l_c = 0; ++l_blk; }
    } else {
        load(b0);
    }
    for (int s = 0; s < steps; s += 2) {
        if (s + 1 < steps) load(b1);
        compute(b0);
        if (s + 2 < steps) load(b0);
        if (s + 1 < steps) compute(b1);
    }
}

template <int C>
__global__ __launch_bounds__(NTH) void joint_layer_p_kernel(const LayerArgs a) {
    const egr_layer_desc& d = a.d;
    using L = PlaneLds<C>;
    constexpr int HEADS = L::HEADS, DH = L::DH, CF = L::CF, FF = L::FF, VPL = C / 64;
    constexpr int LC = L::LC, LQ = L::LQ, LF = L::LF, T = L::T, PW = L::PW, HPP = L::HPP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const bufA = lds;
    float* const bufO = bufA + L::BUFA;
    float* const bufG = bufO + L::BUFO;
    char* const P = reinterpret_cast<char*>(bufG + L::BUFG);
    float* const s_sig = reinterpret_cast<float*>(P + L::PBYTES);      // [HEADS][64]
    float* const s_keep = s_sig + 64 * HEADS;                          // [64]
    unsigned* const s_amax = reinterpret_cast<unsigned*>(s_keep + 64); // slots as in joint_layer_kernel
    float* const tile0 = bufA;
    float* const tile1 = bufA + T;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q4 = lane >> 4;
    const int B = d.B, J = d.J, V = d.V, G = d.groups;
    int grp, fb;
    wg_group_frame(blockIdx.x, G, B, grp, fb);
    const int rows_all = B * J * V, row0 = fb * J * V, nrow = J * V;
    const int64_t xrow0 = ((int64_t)grp * B + fb) * J;
    const float* const gq = d.g + (int64_t)grp * rows_all * HEADS * CF;
    const float* const eq = d.e ? d.e + (int64_t)grp * rows_all * C : nullptr;
    const float* const sg = d.sigma + (int64_t)grp * HEADS * rows_all;
    auto wsz = [](int rows, int k) { return (int64_t)rows * k + rows; };
    const float* const w_fold = d.w_fold + grp * wsz(C, CF);
    const float* const c_fold = d.c_fold + grp * C;
    const float* const w_out = d.w_out + grp * wsz(C, C);
    const float* const b_out = d.b_out + grp * C;
    const float* const w_fuse = d.w_fuse + grp * wsz(C, 4 * C);
    const float* const b_fuse = d.b_fuse + grp * C;
    const float* const w_qkv = d.w_qkv + grp * wsz(3 * C, C);
    const float* const b_qkv = d.b_qkv + grp * 3 * C;
    const float* const w_mo = d.w_mo + grp * wsz(C, C);
    const float* const b_mo = d.b_mo + grp * C;
    const float* const w_f0 = d.w_f0 + grp * wsz(FF, C);
    const float* const b_f0 = d.b_f0 + grp * FF;
    const float* const w_f1 = d.w_f1 + grp * wsz(C, FF);
    const float* const b_f1 = d.b_f1 + grp * C;

    LSTAMP(0);
    if (tid < 64 * HEADS) {
        const int h = tid >> 6, rl = tid & 63;
        s_sig[tid] = rl < nrow ? sg[(int64_t)h * rows_all + row0 + rl] : 0.f;
    }
    if (tid < 64) s_keep[tid] = (tid < nrow && d.rowmask[row0 + tid] != 0) ? 1.f : 0.f;
    if (tid < 16) s_amax[tid] = 0u;
    __syncthreads();

    // a 16-row fp32 tile -> planes (behind the barrier that completed its abs-max slot); returns the inverse scale
    auto split16 = [&](const float* src, int lda, int K, int slot) __attribute__((always_inline)) {
        float sa, inv;
        tile_prescale(s_amax[slot], sa, inv);
        split_tile<1>([&](int row, int k) { return src + row * lda + k; }, K, P, P, 1 << 20, sa, tid);
        return inv;
    };
    // out[16][ldo] = act(planes . W^T + bias) for the wave's column blocks
    auto linear_p = [&](const float* W, int K, const float* bias, int N, float inv, float* out, int ldo, bool gelu, int slot_out, const WPre* pre)
                        __attribute__((always_inline)) {
        float amx = 0.f;
        gemm_planes<1>(P, P, 1 << 20, W, W + (int64_t)N * K, K / 128, wave, NW, N / 16, lane, inv, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = bias[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[0][r] + bb;
                v = gelu ? gelu_erf(v) : v;
                out[(4 * q4 + r) * ldo + col] = v;
                amx = fmaxf(amx, fabsf(v));
            }
        }, pre);
        if (slot_out >= 0) tile_track(s_amax + slot_out, amx);
    };

    // ---- value projection of the sampled rows + output_proj, in two 32-row halves
    {
        constexpr int NPS = C / PW;
        constexpr int NU = HPP * 32 * (CF / 8) / NTH;          // octets of sampled features per thread and pass
        static_assert(HPP * 32 * (CF / 8) % NTH == 0, "staging units per thread");
        f32x4_t stg[NU][2];
        // unit u = tid + it NTH: row i = u & 15, octet o = (u >> 4) & 15, row block rb = (u >> 8) & 1, head hp = u >> 9
        auto stage_load = [&](int hf, int ps) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < NU; ++it) {
                const int u = tid + it * NTH;
                const int i = u & 15, o = (u >> 4) & 15, rb = (u >> 8) & 1, hp = u >> 9;
                const int rl = hf * 32 + rb * 16 + i, h = ps * HPP + hp;
                stg[it][0] = stg[it][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (rl < nrow) {
                    const float* p = gq + ((int64_t)(row0 + rl) * HEADS + h) * CF + o * 8;
                    stg[it][0] = *reinterpret_cast<const f32x4_t*>(p);
                    stg[it][1] = *reinterpret_cast<const f32x4_t*>(p + 4);
                }
            }
        };
        stage_load(0, 0);
        for (int hf = 0; hf < 2; ++hf) {
            WPre pre_out;
            for (int ps = 0; ps < NPS; ++ps) {
                float samx = 0.f;
#pragma unroll
                for (int it = 0; it < NU; ++it)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
                        samx = fmaxf(fmaxf(samx, fmaxf(fabsf(stg[it][hh][0]), fabsf(stg[it][hh][1]))), fmaxf(fabsf(stg[it][hh][2]), fabsf(stg[it][hh][3])));
                tile_track(s_amax + hf * 2 + ps, samx);
                const bool mine = wave * 16 < PW;
                const int n0 = ps * PW + wave * 16;
                const int h = n0 / DH, hp = h - ps * HPP;
                WPre pre;
                float ev[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (mine) {
                    gemm_first_h2(w_fold, CF, n0 / 16, lane, pre);
                    if (eq) {
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int rl = hf * 32 + rb * 16 + 4 * q4 + r;
                                if (rl < nrow) ev[rb][r] = eq[(int64_t)(row0 + rl) * C + n0 + i16];
                            }
                    }
                }
                __syncthreads();                                   // the pass' abs-max slot is complete
                float sa, inv;
                tile_prescale(s_amax[hf * 2 + ps], sa, inv);
#pragma unroll
                for (int it = 0; it < NU; ++it) {                  // the features go from the registers straight into their planes
                    const int u = tid + it * NTH;
                    const int i = u & 15, o = (u >> 4) & 15, rb = (u >> 8) & 1, hp2 = u >> 9;
                    unsigned h0, l0, h1, l1, h2, l2, h3, l3;
                    egrc::split4_f16(stg[it][0][0], stg[it][0][1], stg[it][0][2], stg[it][0][3], sa, h0, l0, h1, l1);
                    egrc::split4_f16(stg[it][1][0], stg[it][1][1], stg[it][1][2], stg[it][1][3], sa, h2, l2, h3, l3);
                    char* dst = P + hp2 * 16384 + ((((o >> 2) & 3) * 2 + rb) * 2) * 1024 + (16 * (o & 3) + i) * 16;
                    *reinterpret_cast<egrc::u32x4*>(dst) = egrc::u32x4{h0, h1, h2, h3};
                    *reinterpret_cast<egrc::u32x4*>(dst + 1024) = egrc::u32x4{l0, l1, l2, l3};
                }
                __syncthreads();
                if (ps + 1 < NPS) stage_load(hf, ps + 1);          // the next pass's features travel under this pass's product
                if (mine) {
                    float amx = 0.f;
                    gemm_planes<2>(P + hp * 16384, P, 1 << 20, w_fold, w_fold + (int64_t)C * CF, 1, n0 / 16, 1, n0 / 16 + 1, lane, inv,
                                   [&](int nb, const f32x4_t (&acc)[2]) {
                        const int col = nb * 16 + i16;
                        const float cf_ = c_fold[col];
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int rl32 = rb * 16 + 4 * q4 + r, rl = hf * 32 + rl32;
                                float v = 0.f;
                                if (rl < nrow) {
                                    v = acc[rb][r] + cf_ * s_sig[h * 64 + rl];
                                    if (eq) v += ev[rb][r];
                                }
                                bufA[rl32 * LC + col] = v;
                                amx = fmaxf(amx, fabsf(v));
                            }
                    }, &pre);
                    tile_track(s_amax + 4 + hf, amx);
                }
                if (ps == NPS - 1) gemm_first_h2(w_out, C, wave, lane, pre_out);
                __syncthreads();
            }
            // output_proj on the 32 rows of this half; masked_fill(~valid) AFTER it (the bias is zeroed too)
            {
                float sa, inv;
                tile_prescale(s_amax[4 + hf], sa, inv);
                split_tile<2>([&](int row, int k) { return bufA + row * LC + k; }, C, P, P, 1 << 20, sa, tid);
                __syncthreads();
                float amx = 0.f;
                gemm_planes<2>(P, P, 1 << 20, w_out, w_out + (int64_t)C * C, C / 128, wave, NW, C / 16, lane, inv, [&](int nb, const f32x4_t (&acc)[2]) {
                    const int col = nb * 16 + i16;
                    const float bo = b_out[col];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int rl = hf * 32 + rb * 16 + 4 * q4 + r;
                            const bool keep = s_keep[rl] != 0.f;
                            const float v = keep ? acc[rb][r] + bo : 0.f;
                            bufO[rl * LC + col] = v;
                            amx = fmaxf(amx, fabsf(v));
                        }
                }, &pre_out);
                tile_track(s_amax + 6, amx);
            }
            if (hf == 0) stage_load(1, 0);
            __syncthreads();
            LSTAMP(16 + hf);
        }
    }
    LSTAMP(1);
    // ---- cat over views -> fuse_mlp: token j = rows 4j .. 4j+3 of bufO side by side (k = view * C + channel)
    constexpr int NX4 = 16 * C / 4 / NTH;
    static_assert(16 * C / 4 % NTH == 0, "residual tile");
    f32x4_t xr[NX4];
    {
        WPre pre;
        gemm_first_h2(w_fuse, 4 * C, wave, lane, pre);
        // the layer input (residual of norm_cross) is requested here and parked in tile0 behind the product
#pragma unroll
        for (int it = 0; it < NX4; ++it) {
            const int idx = tid + it * NTH, r = idx / (C / 4), c4 = idx - r * (C / 4);
            xr[it] = *reinterpret_cast<const f32x4_t*>(d.x + (xrow0 + (r < J ? r : 0)) * C + c4 * 4);
        }
        float sa, inv;
        tile_prescale(s_amax[6], sa, inv);
        split_tile<1>([&](int row, int k) { return bufO + row * (V * LC) + (k / C) * LC + (k % C); }, 4 * C, P, reinterpret_cast<char*>(bufA), L::FUSE_IN_P, sa, tid);
        __syncthreads();
        gemm_planes<1>(P, reinterpret_cast<const char*>(bufA), L::FUSE_IN_P, w_fuse, w_fuse + (int64_t)C * 4 * C, 4 * C / 128, wave, NW, C / 16, lane, inv,
                       [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = b_fuse[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) bufG[(4 * q4 + r) * LC + col] = acc[0][r] + bb;
        }, &pre);
    }
    WPre pre_qkv;
    gemm_first_h2(w_qkv, C, wave, lane, pre_qkv);
    __syncthreads();
    LSTAMP(2);
    // ---- + residual -> norm_cross -> tile1
#pragma unroll
    for (int it = 0; it < NX4; ++it) {
        const int idx = tid + it * NTH, r = idx / (C / 4), c4 = idx - r * (C / 4);
        *reinterpret_cast<f32x4_t*>(tile0 + r * LC + c4 * 4) = xr[it];
    }
    __syncthreads();
    ln_rows<VPL, NW>(bufG, LC, tile0, LC, d.ln1_g + grp * C, d.ln1_b + grp * C, d.eps, tile1, LC, wave, lane, s_amax + 7);
    __syncthreads();
    LSTAMP(3);
    // ---- q/k/v projections -> bufO [16][3C]
    {
        const float inv = split16(tile1, LC, C, 7);
        __syncthreads();
        linear_p(w_qkv, C, b_qkv, 3 * C, inv, bufO, LQ, false, -1, &pre_qkv);
    }
    WPre pre_mo;
    gemm_first_h2(w_mo, C, wave, lane, pre_mo);
    __syncthreads();
    LSTAMP(4);
    // ---- joint-to-joint attention (joint_mha_kernel's arithmetic)
    if (wave < HEADS) {
        const int h = wave;
        float* const sp = bufG + h * 256;
        const float scale = d.mha_scale;
        const int i = lane >> 2, gq_ = lane & 3;
        float sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int jj = gq_ + 4 * t;
            float dot = 0.f;
            if (i < J && jj < J)
                for (int dd = 0; dd < DH; ++dd) dot = fmaf(bufO[i * LQ + h * DH + dd], bufO[jj * LQ + C + h * DH + dd], dot);
            sc[t] = (i < J && jj < J) ? dot * scale : -INFINITY;
            mx = fmaxf(mx, sc[t]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sc[t] = (sc[t] == -INFINITY) ? 0.f : expf(sc[t] - mx);
            sum += sc[t];
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) sp[i * 16 + gq_ + 4 * t] = (i < J) ? sc[t] / sum : 0.f;
    }
    __syncthreads();
    {
        const int h = wave & (HEADS - 1), part = wave / HEADS;
        const float* const sp = bufG + h * 256;
        float amx = 0.f;
        for (int idx = part * 64 + lane; idx < 16 * DH; idx += 64 * (NW / HEADS)) {
            const int t = idx / DH, dd = idx - t * DH;
            float o = 0.f;
            for (int jj = 0; jj < J; ++jj) o = fmaf(sp[t * 16 + jj], bufO[jj * LQ + 2 * C + h * DH + dd], o);
            tile0[t * LC + h * DH + dd] = o;
            amx = fmaxf(amx, fabsf(o));
        }
        tile_track(s_amax + 8, amx);
    }
    __syncthreads();
    LSTAMP(5);
    // ---- out_proj -> bufG, + residual (tile1) -> norm_spatial -> tile0
    {
        const float inv = split16(tile0, LC, C, 8);
        __syncthreads();
        linear_p(w_mo, C, b_mo, C, inv, bufG, LC, false, -1, &pre_mo);
    }
    WPre pre_f0;
    gemm_first_h2(w_f0, C, wave, lane, pre_f0);
    __syncthreads();
    ln_rows<VPL, NW>(bufG, LC, tile1, LC, d.ln2_g + grp * C, d.ln2_b + grp * C, d.eps, tile0, LC, wave, lane, s_amax + 9);
    __syncthreads();
    LSTAMP(6);
    // ---- FFN: Linear + GELU -> bufO [16][512]; Linear -> bufG; + residual (tile0) -> norm_ffn -> tile1
    {
        const float inv = split16(tile0, LC, C, 9);
        __syncthreads();
        linear_p(w_f0, C, b_f0, FF, inv, bufO, LF, true, 10, &pre_f0);
    }
    {
        WPre pre_f1;
        gemm_first_h2(w_f1, FF, wave, lane, pre_f1);
        __syncthreads();
        const float inv = split16(bufO, LF, FF, 10);
        __syncthreads();
        linear_p(w_f1, FF, b_f1, C, inv, bufG, LC, false, -1, &pre_f1);
    }
    __syncthreads();
    ln_rows<VPL, NW>(bufG, LC, tile0, LC, d.ln3_g + grp * C, d.ln3_b + grp * C, d.eps, tile1, LC, wave, lane, s_amax + 11);
    __syncthreads();
    LSTAMP(7);
    // ---- the layer's output tokens
    for (int idx = tid; idx < J * C; idx += NTH) {
        const int r = idx / C, ch = idx - r * C;
        d.x_out[(xrow0 + r) * C + ch] = tile1[r * LC + ch];
    }
    // ---- tail: the next layer's sampling offsets / attention logits from these tokens (straight to global memory)
    if (d.w_ol) {
        const float* const w_ol = d.w_ol + grp * wsz(d.ol_n, C);
        const float* const b_ol = d.b_ol + grp * d.ol_n;
        const int oln = d.ol_n;
        const float inv = split16(tile1, LC, C, 11);
        __syncthreads();
        gemm_planes<1>(P, P, 1 << 20, w_ol, w_ol + (int64_t)oln * C, C / 128, wave, NW, oln / 16, lane, inv, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = b_ol[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q4 + r;
                if (row < J) d.ol_out[(xrow0 + row) * oln + col] = acc[0][r] + bb;
            }
        });
    }
    LSTAMP(8);
    // ---- tail: post_norm [+ head offset | regression MLP + anchor]
    if (d.lnp_g) {
        ln_rows<VPL, NW>(tile1, LC, nullptr, 0, d.lnp_g + grp * C, d.lnp_b + grp * C, d.eps, tile0, LC, wave, lane, s_amax + 12);
        __syncthreads();
        if (d.xn_out)
            for (int idx = tid; idx < J * C; idx += NTH) {
                const int r = idx / C, ch = idx - r * C;
                d.xn_out[(xrow0 + r) * C + ch] = tile0[r * LC + ch];
            }
        if constexpr (C == 256) {
            if (d.w_h0) head_offset_tail<LC>(d, grp, fb, B, J, tile0, bufO, bufG, tid, lane);
        }
        if (d.w_r0) {
            const float* const w_r0 = d.w_r0 + grp * wsz(C, C);
            const float inv = split16(tile0, LC, C, 12);     // (the planes' last readers - the offsets tail - are behind post_norm's barrier)
            __syncthreads();
            linear_p(w_r0, C, d.b_r0 + grp * C, C, inv, bufG, LC, true, -1, nullptr);
            __syncthreads();
            if (tid < J * 3) {
                const int r = tid / 3, o = tid - r * 3;
                const float* w = d.w_r2 + ((int64_t)grp * 3 + o) * C;
                float sum = 0.f;
                for (int k = 0; k < C; ++k) sum = fmaf(bufG[r * LC + k], w[k], sum);
                sum += d.b_r2[grp * 3 + o];
                d.pred_out[(xrow0 + r) * 3 + o] = sum + d.anchors3d[(xrow0 + r) * 3 + o];
            }
        }
    }
    LSTAMP(9);
}

// ------------------------------------------------------------------ token chains in front of the layers (round 5)
// The small launches between the big ones - the JQA query of a refiner and the decoder query of the lifting head - as ONE launch
// each, built from the layer kernel's pieces: one workgroup per (query set, frame), the <= 16 tokens as one MFMA row block in LDS,
// weights streamed in fragment order (w_packed as in egr_joint_layer_f32), intermediates never leave the CU.
template <bool H2>
struct TileGemm {
    int lane, wave;
    bool wpk;
    unsigned* s_amax;
    // epi(nb, acc) for the 16-column blocks nb = wave, wave + NW, ... of  A[16][K] . W^T (W: rows x K, its descales behind the image)
    template <typename Epi>
    __device__ __forceinline__ void run(const float* A, int lda, const float* W, int rows, int K, int slot, Epi&& epi) const {
        if constexpr (H2) {
            float sa, inv;
            tile_prescale(s_amax[slot], sa, inv);
            gemm_cols_h2<1, 1>(A, lda, 0, W, W + (int64_t)rows * K, K, wave, NW, rows / 16, lane, sa, inv, epi);
        } else {
            gemm_cols<1, 1>(A, lda, 0, W, K, 0, K, wave, NW, rows / 16, lane, wpk, epi);
        }
    }
};

struct JqaArgs {
    egr_jqa_query_desc d;
};

// JQA query of HeatmapMVF.forward_feat_only (egoposeformer_heatmap_mvf_ex.py:655-665) behind heatmap_proj[0] + ReLU:
//   hm_embed = heatmap_proj[2](t);  bfb = fc_bfb(adaptive_avg_pool2d(backbone_feat_bottom, 1));
//   x = fc_query(ReLU)((joint_query_embed + bfb) + hm_embed);  and the layer's sampling_offsets / attention_weights Linear of x
// = the launches egr_avgpool_nhwc_f32, 4 x egr_conv2d_nhwc_f32 (small), egr_jqa_sum_f32.
template <int C, bool H2>
__global__ __launch_bounds__(NTH) void jqa_query_kernel(const JqaArgs a) {
    const egr_jqa_query_desc& d = a.d;
    constexpr int LC = C + PAD, KB = 512;
    __shared__ __attribute__((aligned(16))) float tA[16 * LC];
    __shared__ __attribute__((aligned(16))) float tB[16 * LC];
    __shared__ __attribute__((aligned(16))) float tC[16 * LC];
    __shared__ __attribute__((aligned(16))) float tP[KB];
    __shared__ unsigned s_amax[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q4 = lane >> 4;
    const int B = d.B, J = d.J, G = d.groups;
    int grp, fb;
    wg_group_frame(blockIdx.x, G, B, grp, fb);
    const int64_t frame = (int64_t)grp * B + fb, xrow0 = frame * J;
    auto wsz = [](int rows, int k) { return (int64_t)rows * k + (H2 ? rows : 0); };
    const float* const w_hp2 = d.w_hp2 + grp * wsz(C, C);
    const float* const w_bfb = d.w_bfb + grp * wsz(C, KB);
    const float* const w_q = d.w_q + grp * wsz(C, C);
    const float* const w_ol = d.w_ol + grp * wsz(d.ol_n, C);
    const float* const b_hp2 = d.b_hp2 + grp * C;
    const float* const b_bfb = d.b_bfb + grp * C;
    const float* const b_q = d.b_q + grp * C;
    const float* const b_ol = d.b_ol + grp * d.ol_n;
    const float* const embed = d.embed + (int64_t)grp * J * C;
    if (tid < 8) s_amax[tid] = 0u;
    __syncthreads();
    const TileGemm<H2> mm{lane, wave, d.w_packed != 0, s_amax};
    // ---- the token rows behind heatmap_proj[0] (rows >= J: zeros) and the pooled stride-32 features (avgpool_kernel's chain)
    {
        float amx = 0.f;
        for (int idx = tid; idx < 16 * (C / 4); idx += NTH) {
            const int r = idx / (C / 4), c4 = idx - r * (C / 4);
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (r < J) v = *reinterpret_cast<const f32x4_t*>(d.t + (xrow0 + r) * C + c4 * 4);
            *reinterpret_cast<f32x4_t*>(tA + r * LC + c4 * 4) = v;
            amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        if constexpr (H2) tile_track(s_amax + 0, amx);
        const int hw = d.pool_hw;
        const float* const xp = d.s32 + frame * hw * KB;
        float pmx = 0.f;
        for (int ch = tid; ch < KB; ch += NTH) {
            float sum = 0.f;
            int p = 0;
            for (; p + 8 <= hw; p += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = xp[(int64_t)(p + u) * KB + ch];
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += v[u];
            }
            for (; p < hw; ++p) sum += xp[(int64_t)p * KB + ch];
            sum = sum / (float)hw;
            tP[ch] = sum;
            pmx = fmaxf(pmx, fabsf(sum));
        }
        if constexpr (H2) tile_track(s_amax + 1, pmx);
    }
    __syncthreads();
    // ---- hm_embed -> tB;  bfb -> tC (row stride 0: the one pooled row stands for all sixteen, every row of tC is fc_bfb's output)
    mm.run(tA, LC, w_hp2, C, C, 0, [&](int nb, const f32x4_t (&acc)[1]) {
        const int col = nb * 16 + i16;
        const float bb = b_hp2[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) tB[(4 * q4 + r) * LC + col] = acc[0][r] + bb;
    });
    mm.run(tP, 0, w_bfb, C, KB, 1, [&](int nb, const f32x4_t (&acc)[1]) {
        const int col = nb * 16 + i16;
        const float bb = b_bfb[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) tC[(4 * q4 + r) * LC + col] = acc[0][r] + bb;
    });
    __syncthreads();
    // ---- (joint_query_embed + bfb) + heatmap_embed, the reference's association order (jqa_sum_kernel)
    {
        float amx = 0.f;
        for (int idx = tid; idx < 16 * C; idx += NTH) {
            const int r = idx / C, ch = idx - r * C;
            const float v = r < J ? (embed[r * C + ch] + tC[r * LC + ch]) + tB[r * LC + ch] : 0.f;
            tA[r * LC + ch] = v;
            amx = fmaxf(amx, fabsf(v));
        }
        if constexpr (H2) tile_track(s_amax + 2, amx);
    }
    __syncthreads();
    // ---- fc_query + ReLU -> tB and x_out
    {
        float amx = 0.f;
        mm.run(tA, LC, w_q, C, C, 2, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = b_q[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q4 + r;
                float v = acc[0][r] + bb;
                v = v > 0.f ? v : 0.f;
                tB[row * LC + col] = v;
                amx = fmaxf(amx, v);
                if (row < J) d.x_out[(xrow0 + row) * C + col] = v;
            }
        });
        if constexpr (H2) tile_track(s_amax + 3, amx);
    }
    __syncthreads();
    // ---- the layer's sampling offsets / attention logits from the query
    {
        const int oln = d.ol_n;
        mm.run(tB, LC, w_ol, oln, C, 3, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = b_ol[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q4 + r;
                if (row < J) d.ol_out[(xrow0 + row) * oln + col] = acc[0][r] + bb;
            }
        });
    }
}

struct PoseQArgs {
    egr_pose_query_desc d;
};

// Lifting head between mlp_pred[1] and the first decoder layer (egoposeformer_mvf_ex.py:255-262, 317-322, 340-348, 400-410): one
// workgroup per frame - mlp_pred[2] (the 3-D proposal), the fisheye reprojection of its joints into the four views, query_gen_mlp
// (Linear(4, c) + ReLU, Linear + ReLU, Linear) and the first layer's sampling_offsets / attention_weights Linear
// = the launches 4 x egr_conv2d_nhwc_f32 (small), egr_fisheye_project_f32, egr_linear_smallk_f32.
template <int C, bool H2>
__global__ __launch_bounds__(NTH) void pose_query_kernel(const PoseQArgs a) {
    const egr_pose_query_desc& d = a.d;
    constexpr int LC = C + PAD;
    __shared__ __attribute__((aligned(16))) float tA[16 * LC];
    __shared__ __attribute__((aligned(16))) float tB[16 * LC];
    __shared__ __attribute__((aligned(16))) float tH[C];
    __shared__ float s_pred[48], s_q4[16 * 4];
    __shared__ unsigned s_amax[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q4 = lane >> 4;
    const int J = d.J, fb = blockIdx.x;
    const int64_t xrow0 = (int64_t)fb * J;
    if (tid < 8) s_amax[tid] = 0u;
    __syncthreads();
    const TileGemm<H2> mm{lane, wave, d.w_packed != 0, s_amax};
    {
        float amx = 0.f;
        for (int ch = tid; ch < C; ch += NTH) {
            const float v = d.h1[(int64_t)fb * C + ch];
            tH[ch] = v;
            amx = fmaxf(amx, fabsf(v));
        }
        if constexpr (H2) tile_track(s_amax + 0, amx);
    }
    __syncthreads();
    // ---- mlp_pred[2]: C -> 3 J (every row of the product is the frame's one row; row 0 is kept)
    mm.run(tH, 0, d.w_m2, 3 * J, C, 0, [&](int nb, const f32x4_t (&acc)[1]) {
        const int col = nb * 16 + i16;
        if (q4 == 0) s_pred[col] = acc[0][0] + d.b_m2[col];
    });
    __syncthreads();
    // ---- proposal out, fisheye reprojection, decoder query input [ (j + 1) / J, mutated point ]
    if (tid < J) {
        const int j = tid;
        float x = s_pred[3 * j], y = s_pred[3 * j + 1], z = s_pred[3 * j + 2];
        float* const po = d.pred_out + (xrow0 + j) * 3;
        po[0] = x; po[1] = y; po[2] = z;
        egrf::fisheye_joint(x, y, z, d.ctm, d.cams, fb, j, J, d.anchors2d_out, d.valid_out);
        float* const ao = d.anchors3d_out + (xrow0 + j) * 3;
        ao[0] = x; ao[1] = y; ao[2] = z;
        s_q4[j * 4 + 0] = (float)(j + 1) / (float)J;
        s_q4[j * 4 + 1] = x; s_q4[j * 4 + 2] = y; s_q4[j * 4 + 3] = z;
    }
    __syncthreads();
    // ---- query_gen_mlp[0]: Linear(4, C) + ReLU (linear_smallk_kernel's chain)
    {
        float amx = 0.f;
        for (int idx = tid; idx < 16 * C; idx += NTH) {
            const int r = idx / C, n = idx - r * C;
            float v = 0.f;
            if (r < J) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) s = fmaf(s_q4[r * 4 + k], d.w_qg0[n * 4 + k], s);
                s += d.b_qg0[n];
                v = s > 0.f ? s : 0.f;
            }
            tA[r * LC + n] = v;
            amx = fmaxf(amx, v);
        }
        if constexpr (H2) tile_track(s_amax + 1, amx);
    }
    __syncthreads();
    {
        float amx = 0.f;
        mm.run(tA, LC, d.w_qg2, C, C, 1, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = d.b_qg2[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[0][r] + bb;
                v = v > 0.f ? v : 0.f;
                tB[(4 * q4 + r) * LC + col] = v;
                amx = fmaxf(amx, v);
            }
        });
        if constexpr (H2) tile_track(s_amax + 2, amx);
    }
    __syncthreads();
    {
        float amx = 0.f;
        mm.run(tB, LC, d.w_qg4, C, C, 2, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = d.b_qg4[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q4 + r;
                const float v = acc[0][r] + bb;
                tA[row * LC + col] = v;
                amx = fmaxf(amx, fabsf(v));
                if (row < J) d.x_out[(xrow0 + row) * C + col] = v;
            }
        });
        if constexpr (H2) tile_track(s_amax + 3, amx);
    }
    __syncthreads();
    {
        const int oln = d.ol_n;
        mm.run(tA, LC, d.w_ol, oln, C, 3, [&](int nb, const f32x4_t (&acc)[1]) {
            const int col = nb * 16 + i16;
            const float bb = d.b_ol[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q4 + r;
                if (row < J) d.ol_out[(xrow0 + row) * oln + col] = acc[0][r] + bb;
            }
        });
    }
}

// (rows, k) row-major -> fragment order; one thread per 16 bytes of the output
__global__ __launch_bounds__(256) void pack_layer_w_kernel(const float* __restrict__ w, float* __restrict__ out, int rows, int k, int64_t total4) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total4) return;
    const int64_t per = (int64_t)rows * k / 4;          // float4 per matrix
    const int64_t m = o / per;
    int64_t r = o - m * per;
    const int lane = (int)(r & 63); r >>= 6;
    const int u = (int)(r & 7); r >>= 3;
    const int kc = (int)(r % (k / 128)), nb = (int)(r / (k / 128));
    const int i = lane & 15, q = lane >> 4;
    *reinterpret_cast<f32x4_t*>(out + o * 4) =
        *reinterpret_cast<const f32x4_t*>(w + (m * rows + nb * 16 + i) * k + kc * 128 + u * 16 + 4 * q);
}

// fp16 scheme: per matrix row the power of two that puts max |w| into [2^14, 2^15) (its inverse goes behind the image) ...
__global__ __launch_bounds__(256) void layer_wh2_rowscale_kernel(const float* __restrict__ w, float* __restrict__ out, int rows, int k, int64_t rows_all) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows_all) return;
    const float* r = w + row * k;
    float m = 0.f;
    for (int c = lane * 4; c < k; c += 256) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(r + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    if (lane == 0) {
        const int e = (int)(__float_as_uint(m) >> 23);
        int kk = 141 - e;
        kk = kk > 60 ? 60 : (kk < -60 ? -60 : kk);
        const int64_t mtx = row / rows;
        out[mtx * ((int64_t)rows * k + rows) + (int64_t)rows * k + (row - mtx * rows)] = __uint_as_float((unsigned)(127 - kk) << 23);
    }
}
// ... and the two fp16 planes in fragment order; one thread per 16 bytes of the image
__global__ __launch_bounds__(256) void pack_layer_wh2_kernel(const float* __restrict__ w, float* __restrict__ out, int rows, int k, int64_t total4) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total4) return;
    const int64_t per = (int64_t)rows * k / 4;          // 16-byte units per matrix
    const int64_t m = o / per;
    int64_t r = o - m * per;
    const int lane = (int)(r & 63); r >>= 6;
    const int u = (int)(r & 7); r >>= 3;
    const int kc = (int)(r % (k / 128)), nb = (int)(r / (k / 128));
    const int i = lane & 15, q = lane >> 4, kb = u >> 1, plane = u & 1;
    const int64_t mbase = m * ((int64_t)rows * k + rows);
    const float s = 1.f / out[mbase + (int64_t)rows * k + nb * 16 + i];     // exact: a power of two
    const float* src = w + (m * rows + nb * 16 + i) * k + kc * 128 + kb * 32 + q * 8;
    const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(src), x1 = *reinterpret_cast<const f32x4_t*>(src + 4);
    unsigned h[4], l[4];
    egrc::split2_f16(x0[0], x0[1], s, h[0], l[0]);
    egrc::split2_f16(x0[2], x0[3], s, h[1], l[1]);
    egrc::split2_f16(x1[0], x1[1], s, h[2], l[2]);
    egrc::split2_f16(x1[2], x1[3], s, h[3], l[3]);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    *reinterpret_cast<u32x4_t*>(out + mbase + (o - m * per) * 4) = plane ? u32x4_t{l[0], l[1], l[2], l[3]} : u32x4_t{h[0], h[1], h[2], h[3]};
}

}  // namespace

#ifdef LAYER_STAMPS
extern "C" int egr_layer_stamps(unsigned long long* host64) {
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_layer_stamps), 64 * sizeof(unsigned long long));
}
#endif

extern "C" int egr_pack_layer_wh2_f32(const float* w, int32_t matrices, int32_t rows, int32_t k, float* out, void* stream) {
    if (!w || !out) return EGR_ENULL;
    if (matrices <= 0 || rows <= 0 || k <= 0 || rows % 16 != 0 || k % 128 != 0 || (((uintptr_t)w | (uintptr_t)out) & 15)) return EGR_EINVAL;
    const int64_t total4 = (int64_t)matrices * rows * k / 4, rows_all = (int64_t)matrices * rows;
    if ((total4 + 255) / 256 >= (1LL << 31)) return EGR_EINVAL;
    hipLaunchKernelGGL(layer_wh2_rowscale_kernel, dim3((unsigned)((rows_all + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, out, rows, k, rows_all);
    hipLaunchKernelGGL(pack_layer_wh2_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, rows, k, total4);
    return egr_launch_status();
}

extern "C" int egr_pack_layer_w_f32(const float* w, int32_t matrices, int32_t rows, int32_t k, float* out, void* stream) {
    if (!w || !out) return EGR_ENULL;
    if (matrices <= 0 || rows <= 0 || k <= 0 || rows % 16 != 0 || k % 128 != 0 || (((uintptr_t)w | (uintptr_t)out) & 15)) return EGR_EINVAL;
    const int64_t total4 = (int64_t)matrices * rows * k / 4;
    if ((total4 + 255) / 256 >= (1LL << 31)) return EGR_EINVAL;
    hipLaunchKernelGGL(pack_layer_w_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, rows, k, total4);
    return egr_launch_status();
}

// fp16 scheme: 1 = joint_layer_p_kernel (A tiles split once), 0 = joint_layer_kernel<C, true> (split inside the product loop)
static int g_layer_planes = getenv("EGR_LAYER_PLANES") ? atoi(getenv("EGR_LAYER_PLANES")) != 0 : 1;
extern "C" int egr_layer_set_planes(int on) {
    const int old = g_layer_planes;
    if (on >= 0) g_layer_planes = on != 0;
    return old;
}

extern "C" int egr_joint_layer_f32(const egr_layer_desc* dd, void* stream) {
    if (!dd) return EGR_ENULL;
    const egr_layer_desc& d = *dd;
    if (!d.x || !d.g || !d.sigma || !d.rowmask || !d.x_out) return EGR_ENULL;
    if (!d.w_fold || !d.c_fold || !d.w_out || !d.b_out || !d.w_fuse || !d.b_fuse || !d.ln1_g || !d.ln1_b || !d.w_qkv || !d.b_qkv || !d.w_mo ||
        !d.b_mo || !d.ln2_g || !d.ln2_b || !d.w_f0 || !d.b_f0 || !d.w_f1 || !d.b_f1 || !d.ln3_g || !d.ln3_b)
        return EGR_ENULL;
    if (d.B <= 0 || d.groups <= 0 || d.J <= 0 || d.J > 16 || d.V != 4 || d.heads != 4 || d.cf != 128 || d.ffn_dim != 512) return EGR_EINVAL;
    if (d.C != 128 && d.C != 256) return EGR_EINVAL;
    if (d.w_ol && (!d.b_ol || !d.ol_out || d.ol_n <= 0 || d.ol_n % 16 != 0)) return EGR_EINVAL;
    if (d.lnp_g && !d.lnp_b) return EGR_ENULL;
    if ((d.xn_out || d.w_r0 || d.w_h0) && !d.lnp_g) return EGR_EINVAL;
    if (d.w_h0 && (!d.b_h0 || !d.h0_out)) return EGR_ENULL;
    if (d.w_h0 && (d.C != 256 || d.h0_n != 64 || d.w_r0 || ((uintptr_t)d.h0_out & 15))) return EGR_EINVAL;   // s * s == c tokens-as-image; 16-byte stores
    if (d.w_r0 && (!d.b_r0 || !d.w_r2 || !d.b_r2 || !d.anchors3d || !d.pred_out)) return EGR_ENULL;
    if ((int64_t)d.B * d.groups >= (1LL << 31) || (int64_t)d.B * d.J * d.V * d.heads * d.cf >= (1LL << 31)) return EGR_EINVAL;
    const uintptr_t al = (uintptr_t)d.g | (uintptr_t)d.w_fold | (uintptr_t)d.w_out | (uintptr_t)d.w_fuse | (uintptr_t)d.w_qkv | (uintptr_t)d.w_mo |
                         (uintptr_t)d.w_f0 | (uintptr_t)d.w_f1 | (uintptr_t)(d.w_ol ? d.w_ol : d.w_out) | (uintptr_t)(d.w_r0 ? d.w_r0 : d.w_out);
    if (al & 15) return EGR_EINVAL;   // 16-byte weight / feature loads
    LayerArgs a;
    a.d = d;
    const dim3 grid((unsigned)(d.B * d.groups)), block(NTH);
    hipStream_t s = (hipStream_t)stream;
    // more than 64 KiB of dynamic LDS must be allowed per kernel and per device (a host-side attribute, not a stream operation)
    static bool allowed[4][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return EGR_EINVAL;
    if (d.w_packed < 0 || d.w_packed > 2) return EGR_EINVAL;
    auto run = [&](auto c_tag, auto h2_tag, int slot) {
        constexpr int CC = decltype(c_tag)::value;
        constexpr bool HH = decltype(h2_tag)::value;
        if (!allowed[slot][dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(joint_layer_kernel<CC, HH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LayerLds<CC>::BYTES) != hipSuccess)
                return EGR_EINVAL;
            allowed[slot][dev] = true;
        }
        hipLaunchKernelGGL((joint_layer_kernel<CC, HH>), grid, block, LayerLds<CC>::BYTES, s, a);
        return egr_launch_status();
    };
    using C256 = std::integral_constant<int, 256>;
    using C128 = std::integral_constant<int, 128>;
    // fp16 scheme: the kernel that splits its A tiles once (EGR_LAYER_PLANES=0: the round-4 kernel that splits inside the product loop)
    if (d.w_packed == 2 && g_layer_planes) {
        static bool allowed_p[2][64] = {};
        auto run_p = [&](auto c_tag, int slot) {
            constexpr int CC = decltype(c_tag)::value;
            if (!allowed_p[slot][dev]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(joint_layer_p_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)PlaneLds<CC>::BYTES) != hipSuccess)
                    return EGR_EINVAL;
                allowed_p[slot][dev] = true;
            }
            hipLaunchKernelGGL((joint_layer_p_kernel<CC>), grid, block, PlaneLds<CC>::BYTES, s, a);
            return egr_launch_status();
        };
        return d.C == 256 ? run_p(C256{}, 0) : run_p(C128{}, 1);
    }
    if (d.C == 256) return d.w_packed == 2 ? run(C256{}, std::true_type{}, 0) : run(C256{}, std::false_type{}, 1);
    return d.w_packed == 2 ? run(C128{}, std::true_type{}, 2) : run(C128{}, std::false_type{}, 3);
}

extern "C" int egr_jqa_query_f32(const egr_jqa_query_desc* dd, void* stream) {
    if (!dd) return EGR_ENULL;
    const egr_jqa_query_desc& d = *dd;
    if (!d.t || !d.s32 || !d.w_hp2 || !d.b_hp2 || !d.w_bfb || !d.b_bfb || !d.embed || !d.w_q || !d.b_q || !d.w_ol || !d.b_ol || !d.x_out || !d.ol_out)
        return EGR_ENULL;
    if (d.B <= 0 || d.groups <= 0 || d.J <= 0 || d.J > 16 || d.C != 256 || d.kb != 512 || d.pool_hw <= 0 || d.ol_n <= 0 || d.ol_n % 16 != 0) return EGR_EINVAL;
    if (d.w_packed < 0 || d.w_packed > 2) return EGR_EINVAL;
    if ((int64_t)d.B * d.groups >= (1LL << 31)) return EGR_EINVAL;
    if (((uintptr_t)d.t | (uintptr_t)d.w_hp2 | (uintptr_t)d.w_bfb | (uintptr_t)d.w_q | (uintptr_t)d.w_ol) & 15) return EGR_EINVAL;
    JqaArgs a;
    a.d = d;
    const dim3 grid((unsigned)(d.B * d.groups)), block(NTH);
    if (d.w_packed == 2) hipLaunchKernelGGL((jqa_query_kernel<256, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((jqa_query_kernel<256, false>), grid, block, 0, (hipStream_t)stream, a);
    return egr_launch_status();
}

extern "C" int egr_pose_query_f32(const egr_pose_query_desc* dd, void* stream) {
    if (!dd) return EGR_ENULL;
    const egr_pose_query_desc& d = *dd;
    if (!d.h1 || !d.w_m2 || !d.b_m2 || !d.cams || !d.w_qg0 || !d.b_qg0 || !d.w_qg2 || !d.b_qg2 || !d.w_qg4 || !d.b_qg4 || !d.w_ol || !d.b_ol ||
        !d.pred_out || !d.anchors3d_out || !d.anchors2d_out || !d.valid_out || !d.x_out || !d.ol_out)
        return EGR_ENULL;
    if (d.B <= 0 || d.J != 16 || d.C != 128 || d.ol_n <= 0 || d.ol_n % 16 != 0 || d.w_packed < 0 || d.w_packed > 2) return EGR_EINVAL;
    if (((uintptr_t)d.w_m2 | (uintptr_t)d.w_qg2 | (uintptr_t)d.w_qg4 | (uintptr_t)d.w_ol) & 15) return EGR_EINVAL;
    PoseQArgs a;
    a.d = d;
    const dim3 grid((unsigned)d.B), block(NTH);
    if (d.w_packed == 2) hipLaunchKernelGGL((pose_query_kernel<128, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((pose_query_kernel<128, false>), grid, block, 0, (hipStream_t)stream, a);
    return egr_launch_status();
}
