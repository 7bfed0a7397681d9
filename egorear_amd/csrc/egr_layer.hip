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
#include "egr_common.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct LayerArgs {
    egr_layer_desc d;
};

constexpr int PAD = 4;   // LDS row padding (floats)

__device__ __forceinline__ float gelu_erf(float t) { return 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f)); }

// acc[rb] (+)= A[rb*16 .. +16][0..K) . W[nb*16 .. +16][0..K)^T for NSEG (A, W) segment pairs: A segment s starts at A + s*a_seg
// (row stride lda), W segment s at W + s*w_seg (row stride ldw).  K % 128 == 0.
template <int RB, int NSEG>
__device__ __forceinline__ void mfma_block(f32x4_t (&acc)[RB], const float* __restrict__ A, int lda, int a_seg, const float* __restrict__ W, int ldw,
                                           int w_seg, int K, int nb, int lane) {
    const int i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const float* wrow = W + (int64_t)(nb * 16 + i) * ldw + 4 * q;
    const float* arow = A + i * lda + 4 * q;
    constexpr int CH = 8;      // 16-deep k blocks per chunk (128 k)
    const int chunks = NSEG * (K / (16 * CH));
    const int cps = K / (16 * CH);   // chunks per segment
    auto wptr = [&](int c) { return wrow + (c / cps) * w_seg + (c % cps) * (16 * CH); };
    auto aptr = [&](int c) { return arow + (c / cps) * a_seg + (c % cps) * (16 * CH); };
    f32x4_t b0[CH], b1[CH];
    auto load = [&](f32x4_t (&b)[CH], int c) {
        const float* p = wptr(c);
#pragma unroll
        for (int u = 0; u < CH; ++u) b[u] = *reinterpret_cast<const f32x4_t*>(p + 16 * u);
    };
    auto compute = [&](const f32x4_t (&b)[CH], int c) {
        const float* p = aptr(c);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const f32x4_t af = *reinterpret_cast<const f32x4_t*>(p + rb * 16 * lda + 16 * u);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], b[u][t], acc[rb], 0, 0, 0);
            }
        }
    };
    load(b0, 0);
    for (int c = 0; c < chunks; c += 2) {
        if (c + 1 < chunks) load(b1, c + 1);
        compute(b0, c);
        if (c + 2 < chunks) load(b0, c + 2);
        if (c + 1 < chunks) compute(b1, c + 1);
    }
}

// rows [r0, r0 + 4) of a 16-row tile: y = LayerNorm(t [+ res]) * gamma + beta, lane -> channels i*64 + lane (layernorm_kernel)
template <int VPL>
__device__ __forceinline__ void ln_rows(const float* t, int ldt, const float* res, int ldr, const float* gamma, const float* beta, float eps,
                                        float* y, int ldy, int wave, int lane) {
    constexpr int c = VPL * 64;
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr;
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
            y[row * ldy + ch] = (v[i] - mean) * rstd * gamma[ch] + beta[ch];
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void joint_layer_kernel(const LayerArgs a) {
    const egr_layer_desc& d = a.d;
    constexpr int HEADS = 4, DH = C / HEADS, CF = 128, FF = 512, VPL = C / 64;
    constexpr int LC = C + PAD, LG = CF + PAD, LQ = 3 * C + PAD, LF = FF + PAD;
    constexpr int T = 16 * LC;                         // one 16-row tile of C-wide activations
    constexpr int HPP = 64 / DH;                       // heads staged per 64-column pass of the value projection
    constexpr int BUFA = 2 * T;                        // a_half (32 rows) | later tile0 / tile1
    constexpr int BUFO = (64 * LC > 16 * LQ ? (64 * LC > 16 * LF ? 64 * LC : 16 * LF) : (16 * LQ > 16 * LF ? 16 * LQ : 16 * LF));
    constexpr int BUFG = (HPP * 32 * LG > T ? HPP * 32 * LG : T);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const bufA = lds;
    float* const bufO = bufA + BUFA;
    float* const bufG = bufO + BUFO;
    float* const tile0 = bufA;
    float* const tile1 = bufA + T;

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
    const float* const w_fold = d.w_fold + (int64_t)grp * C * CF;
    const float* const c_fold = d.c_fold + grp * C;
    const float* const w_out = d.w_out + (int64_t)grp * C * C;
    const float* const b_out = d.b_out + grp * C;
    const float* const w_fuse = d.w_fuse + (int64_t)grp * C * V * C;
    const float* const b_fuse = d.b_fuse + grp * C;
    const float* const w_qkv = d.w_qkv + (int64_t)grp * 3 * C * C;
    const float* const b_qkv = d.b_qkv + grp * 3 * C;
    const float* const w_mo = d.w_mo + (int64_t)grp * C * C;
    const float* const b_mo = d.b_mo + grp * C;
    const float* const w_f0 = d.w_f0 + (int64_t)grp * FF * C;
    const float* const b_f0 = d.b_f0 + grp * FF;
    const float* const w_f1 = d.w_f1 + (int64_t)grp * C * FF;
    const float* const b_f1 = d.b_f1 + grp * C;

    // ---- value projection of the sampled rows (sample-then-project) + output_proj, in two 32-row halves
    for (int hf = 0; hf < 2; ++hf) {
        for (int ps = 0; ps < C / 64; ++ps) {          // 64 output columns per pass: one 16-column block per wave
            // stage the sampled features of the heads behind these columns: bufG[hp][32][LG]
            for (int idx = tid; idx < HPP * 32 * (CF / 4); idx += 256) {
                const int cq = idx % (CF / 4), r = (idx / (CF / 4)) % 32, hp = idx / (32 * (CF / 4));
                const int rl = hf * 32 + r, h = ps * HPP + hp;
                f32x4_t v = {0.f, 0.f, 0.f, 0.f};
                if (rl < nrow) v = *reinterpret_cast<const f32x4_t*>(gq + ((int64_t)(row0 + rl) * HEADS + h) * CF + cq * 4);
                *reinterpret_cast<f32x4_t*>(bufG + (hp * 32 + r) * LG + cq * 4) = v;
            }
            __syncthreads();
            {
                const int n0 = ps * 64 + wave * 16;    // this wave's output columns [n0, n0 + 16)
                const int h = n0 / DH, hp = h - ps * HPP;
                f32x4_t acc[2];
                mfma_block<2, 1>(acc, bufG + hp * 32 * LG, LG, 0, w_fold, CF, 0, CF, n0 / 16, lane);
                const int col = n0 + i16;
                const float cf_ = c_fold[col];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rl32 = rb * 16 + 4 * q4 + r, rl = hf * 32 + rl32;
                        float v = 0.f;
                        if (rl < nrow) {
                            v = acc[rb][r] * 1.0f + cf_ * sg[(int64_t)h * rows_all + row0 + rl];
                            if (eq) v += eq[(int64_t)(row0 + rl) * C + col];
                        }
                        bufA[rl32 * LC + col] = v;
                    }
            }
            __syncthreads();
        }
        // output_proj on the 32 rows of this half; masked_fill(~valid) AFTER it (the bias is zeroed too)
        for (int nb = wave; nb < C / 16; nb += 4) {
            f32x4_t acc[2];
            mfma_block<2, 1>(acc, bufA, LC, 0, w_out, C, 0, C, nb, lane);
            const int col = nb * 16 + i16;
            const float bo = b_out[col];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = hf * 32 + rb * 16 + 4 * q4 + r;
                    const bool keep = rl < nrow && d.rowmask[row0 + rl] != 0;
                    bufO[rl * LC + col] = keep ? acc[rb][r] + bo : 0.f;
                }
        }
        __syncthreads();
    }
    // ---- cat over views -> fuse_mlp: token j = rows 4j .. 4j+3 of bufO side by side (V segments of K = C)
    for (int nb = wave; nb < C / 16; nb += 4) {
        f32x4_t acc[1];
        mfma_block<1, 4>(acc, bufO, V * LC, LC, w_fuse, V * C, C, C, nb, lane);
        const int col = nb * 16 + i16;
        const float bb = b_fuse[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) tile0[(4 * q4 + r) * LC + col] = acc[0][r] + bb;
    }
    __syncthreads();
    // ---- + residual (the layer input, from global memory) -> norm_cross
    {
        // residual rows of tokens >= J do not exist: point them at token 0 (their results are never stored)
        for (int idx = tid; idx < 16 * C; idx += 256) {
            const int r = idx / C, ch = idx - r * C;
            bufG[r * LC + ch] = d.x[(xrow0 + (r < J ? r : 0)) * C + ch];
        }
        __syncthreads();
        ln_rows<VPL>(tile0, LC, bufG, LC, d.ln1_g + grp * C, d.ln1_b + grp * C, d.eps, tile1, LC, wave, lane);
    }
    __syncthreads();
    // ---- q/k/v projections -> bufO [16][3C]
    for (int nb = wave; nb < 3 * C / 16; nb += 4) {
        f32x4_t acc[1];
        mfma_block<1, 1>(acc, tile1, LC, 0, w_qkv, C, 0, C, nb, lane);
        const int col = nb * 16 + i16;
        const float bb = b_qkv[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) bufO[(4 * q4 + r) * LQ + col] = acc[0][r] + bb;
    }
    __syncthreads();
    // ---- joint-to-joint attention, one wave per head (joint_mha_kernel's arithmetic) -> tile0
    {
        const int h = wave;
        float* const sp = bufG + wave * 256;          // this head's 16 x 16 probabilities
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
        __syncthreads();
        for (int idx = lane; idx < 16 * DH; idx += 64) {
            const int t = idx / DH, dd = idx - t * DH;
            float o = 0.f;
            for (int jj = 0; jj < J; ++jj) o = fmaf(sp[t * 16 + jj], bufO[jj * LQ + 2 * C + h * DH + dd], o);
            tile0[t * LC + h * DH + dd] = o;
        }
    }
    __syncthreads();
    // ---- out_proj -> bufG, + residual (tile1) -> norm_spatial -> tile0
    for (int nb = wave; nb < C / 16; nb += 4) {
        f32x4_t acc[1];
        mfma_block<1, 1>(acc, tile0, LC, 0, w_mo, C, 0, C, nb, lane);
        const int col = nb * 16 + i16;
        const float bb = b_mo[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) bufG[(4 * q4 + r) * LC + col] = acc[0][r] + bb;
    }
    __syncthreads();
    ln_rows<VPL>(bufG, LC, tile1, LC, d.ln2_g + grp * C, d.ln2_b + grp * C, d.eps, tile0, LC, wave, lane);
    __syncthreads();
    // ---- FFN: Linear + GELU -> bufO [16][512]; Linear -> bufG; + residual (tile0) -> norm_ffn -> tile1
    for (int nb = wave; nb < FF / 16; nb += 4) {
        f32x4_t acc[1];
        mfma_block<1, 1>(acc, tile0, LC, 0, w_f0, C, 0, C, nb, lane);
        const int col = nb * 16 + i16;
        const float bb = b_f0[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) bufO[(4 * q4 + r) * LF + col] = gelu_erf(acc[0][r] + bb);
    }
    __syncthreads();
    for (int nb = wave; nb < C / 16; nb += 4) {
        f32x4_t acc[1];
        mfma_block<1, 1>(acc, bufO, LF, 0, w_f1, FF, 0, FF, nb, lane);
        const int col = nb * 16 + i16;
        const float bb = b_f1[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) bufG[(4 * q4 + r) * LC + col] = acc[0][r] + bb;
    }
    __syncthreads();
    ln_rows<VPL>(bufG, LC, tile0, LC, d.ln3_g + grp * C, d.ln3_b + grp * C, d.eps, tile1, LC, wave, lane);
    __syncthreads();
    // ---- the layer's output tokens
    for (int idx = tid; idx < J * C; idx += 256) {
        const int r = idx / C, ch = idx - r * C;
        d.x_out[(xrow0 + r) * C + ch] = tile1[r * LC + ch];
    }
    // ---- tail: the next layer's sampling offsets / attention logits from these tokens (straight to global memory)
    if (d.w_ol) {
        const float* const w_ol = d.w_ol + (int64_t)grp * d.ol_n * C;
        const float* const b_ol = d.b_ol + grp * d.ol_n;
        for (int nb = wave; nb < d.ol_n / 16; nb += 4) {
            f32x4_t acc[1];
            mfma_block<1, 1>(acc, tile1, LC, 0, w_ol, C, 0, C, nb, lane);
            const int col = nb * 16 + i16;
            const float bb = b_ol[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q4 + r;
                if (row < J) d.ol_out[(xrow0 + row) * d.ol_n + col] = acc[0][r] + bb;
            }
        }
    }
    // ---- tail: post_norm [+ regression MLP + anchor]
    if (d.lnp_g) {
        ln_rows<VPL>(tile1, LC, nullptr, 0, d.lnp_g + grp * C, d.lnp_b + grp * C, d.eps, tile0, LC, wave, lane);
        __syncthreads();
        if (d.xn_out)
            for (int idx = tid; idx < J * C; idx += 256) {
                const int r = idx / C, ch = idx - r * C;
                d.xn_out[(xrow0 + r) * C + ch] = tile0[r * LC + ch];
            }
        if (d.w_r0) {
            const float* const w_r0 = d.w_r0 + (int64_t)grp * C * C;
            const float* const b_r0 = d.b_r0 + grp * C;
            for (int nb = wave; nb < C / 16; nb += 4) {
                f32x4_t acc[1];
                mfma_block<1, 1>(acc, tile0, LC, 0, w_r0, C, 0, C, nb, lane);
                const int col = nb * 16 + i16;
                const float bb = b_r0[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) bufG[(4 * q4 + r) * LC + col] = gelu_erf(acc[0][r] + bb);
            }
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
}

template <int C>
constexpr size_t layer_lds_bytes() {
    constexpr int LC = C + PAD, LG = 128 + PAD, LQ = 3 * C + PAD, LF = 512 + PAD, T = 16 * LC, HPP = 64 / (C / 4);
    constexpr int BUFA = 2 * T;
    constexpr int BUFO = (64 * LC > 16 * LQ ? (64 * LC > 16 * LF ? 64 * LC : 16 * LF) : (16 * LQ > 16 * LF ? 16 * LQ : 16 * LF));
    constexpr int BUFG = (HPP * 32 * LG > T ? HPP * 32 * LG : T);
    return sizeof(float) * (size_t)(BUFA + BUFO + BUFG);
}

}  // namespace

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
    if ((d.xn_out || d.w_r0) && !d.lnp_g) return EGR_EINVAL;
    if (d.w_r0 && (!d.b_r0 || !d.w_r2 || !d.b_r2 || !d.anchors3d || !d.pred_out)) return EGR_ENULL;
    if ((int64_t)d.B * d.groups >= (1LL << 31) || (int64_t)d.B * d.J * d.V * d.heads * d.cf >= (1LL << 31)) return EGR_EINVAL;
    const uintptr_t al = (uintptr_t)d.g | (uintptr_t)d.w_fold | (uintptr_t)d.w_out | (uintptr_t)d.w_fuse | (uintptr_t)d.w_qkv | (uintptr_t)d.w_mo |
                         (uintptr_t)d.w_f0 | (uintptr_t)d.w_f1 | (uintptr_t)(d.w_ol ? d.w_ol : d.w_out) | (uintptr_t)(d.w_r0 ? d.w_r0 : d.w_out);
    if (al & 15) return EGR_EINVAL;   // 16-byte weight / feature loads
    LayerArgs a;
    a.d = d;
    const dim3 grid((unsigned)(d.B * d.groups)), block(256);
    hipStream_t s = (hipStream_t)stream;
    // more than 64 KiB of dynamic LDS must be allowed per kernel and per device (a host-side attribute, not a stream operation)
    static bool allowed[2][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return EGR_EINVAL;
    if (d.C == 256) {
        if (!allowed[0][dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(joint_layer_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)layer_lds_bytes<256>()) != hipSuccess)
                return EGR_EINVAL;
            allowed[0][dev] = true;
        }
        hipLaunchKernelGGL(joint_layer_kernel<256>, grid, block, layer_lds_bytes<256>(), s, a);
    } else {
        if (!allowed[1][dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(joint_layer_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)layer_lds_bytes<128>()) != hipSuccess)
                return EGR_EINVAL;
            allowed[1][dev] = true;
        }
        hipLaunchKernelGGL(joint_layer_kernel<128>, grid, block, layer_lds_bytes<128>(), s, a);
    }
    return egr_launch_status();
}
