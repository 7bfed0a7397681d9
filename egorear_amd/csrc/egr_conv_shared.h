// Shared by the translation units of the implicit-GEMM convolution (egr_conv.hip, egr_conv_tapx.hip): launch arguments,
// invariant-divisor helpers, the operand formats of the split kernels and their conversion / record helpers.
#pragma once
#include <type_traits>
#include <cstdlib>

#include "egr_common.h"

namespace egrc {

// q = n / d for 0 <= n < 2^31 by multiply-high: l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1,
// q = (umulhi(m, n) + n) >> l.
struct FastDiv {
    uint32_t mul, shift, d;
};
inline FastDiv make_fastdiv(int dd) {
    FastDiv f;
    uint32_t d = (uint32_t)dd;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.shift = l;
    f.d = d;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
    return (int)((__umulhi(f.mul, (uint32_t)n) + (uint32_t)n) >> f.shift);
}
__device__ __forceinline__ int64_t fmap(const egr_nmap& m, const FastDiv& f, int n) {
    int o = fdiv(n, f);
    int i = n - o * (int)f.d;
    return (int64_t)i * m.stride_inner + (int64_t)o * m.stride_outer;
}

struct ConvArgs {
    egr_conv_desc d;
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    const float* rowscale;
    const uint8_t* rowmask;
    const float* mask;   // data gradient behind a ReLU: output *= [mask > 0]; same layout / offsets as y (NULL = off)
    float* y;
    float* ws;
    int M, Npad, K;
    int ktiles, ktiles_per_split;
    int tilesM, tilesN;
    int ntiles;   // tilesM * tilesN: tiles of one group (gridDim.x of a non-persistent launch)
    int cblocks;  // cin / 32
    int taps;     // kh * kw
    FastDiv dHoWo, dWo, dXin, dYin, dRin, dTilesN;  // invariant-divisor division (no integer divide in the kernel)
    int howo_shift, wo_shift;        // >= 0 when ho*wo / wo are powers of two (every layer of the path): shifts, no division
    int x_plain, y_plain, r_plain;   // image map is a plain batch (n_inner >= n): offset = n * stride_inner
    unsigned long long* dbg;  // diagnostic only: per-block phase stamps (s_memtime), NULL in normal operation
    int vec_ok;   // NHWC output / residual addresses are 16-byte aligned for every (row, channel quad)
    int cls_mode; // stride-2 data gradient split into the four output-parity classes (blockIdx.y): M, dHoWo, dWo, *_shift describe ONE class
    int* cnt;     // split-K: arrival counters (groups x tiles, zero between launches) - the last slice of a tile reduces it; NULL = separate pass
    // EGR_W_F16X2 (two-way fp16 operand split, three products; see SplitFmt below): per-output-channel descale of the weights
    // ((groups,) Npad floats, exact powers of two) and the abs-max record of the activations (64 slots of float bits: the launch
    // scales x by the power of two that puts max |x| into [2^14, 2^15))
    const float* wds;
    const unsigned* amax_in;
    unsigned* amax_out;   // any format: max |y| of this launch is folded into the 64 slots (atomic max on the float bits); NULL = off
    // train-mode BatchNorm behind this conv (egr_conv_aux.bn_partials): every tile leaves the per-channel sum / sum of squares (double)
    // and min / max (float) of the rows it stores - the slab layout of egr_bn_stats_f32's first pass with one slab per M tile:
    // bn_part [groups][tilesM][2][cout] doubles, then [groups][tilesM][2][cout] floats
    double* bn_part;
    int* bn_tiles_host;      // HOST pointers (conv_run): where the launch reports its M tiles per group, capacity of bn_part in doubles
    size_t bn_cap;
};

// host side, wherever a launch path has fixed its tile height: report the slab count and check the partials fit
static inline int bn_slabs(const ConvArgs& a) {
    if (!a.bn_part) return 0;
    if ((size_t)a.d.groups * a.tilesM * 3 * a.d.cout > a.bn_cap) return EGR_EWORKSPACE;
    *a.bn_tiles_host = a.tilesM;
    return 0;
}

constexpr int BK = 32;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- operand formats of the split kernels (template parameter NPL = planes per operand)
//   NPL = 3 (EGR_W_BF16X3): x = hi + mid + lo, three bf16 (exact); the six products of order <= 2 on v_mfma_f32_32x32x16_bf16
//   NPL = 2 (EGR_W_F16X2):  x 2^e = h + l, two fp16 (22 significant bits); the three products (l,h) (h,l) (h,h) on
//            v_mfma_f32_32x32x16_f16.  The dropped (l,l) product and the representation error are ~2^-22 of a product, below the
//            accumulated rounding of an fp32 fma chain (tools/proto/f16x3_accuracy.hip: rms 3.8e-7 against 4.4e-7 for the chain and
//            5.0e-7 for the bf16 scheme).  fp16's range is narrow, hence the exact power-of-two pre-scales: the activations by
//            2^e chosen per launch from their recorded abs-max (amax_in), the weights per output channel at pack time; the
//            accumulators are multiplied by the inverse before the epilogue.  Half the matrix instructions and two thirds of
//            the operand bytes of the bf16 scheme.
constexpr int split_npr(int npl) { return npl == 3 ? 6 : 3; }
// plane of the A operand whose last use is product t (-1: none): its registers can be refilled for the next tap behind it
constexpr int split_free_a(int npl, int t) { return npl == 3 ? (t == 0 ? 2 : (t == 3 ? 1 : (t == 5 ? 0 : -1))) : (t == 0 ? 1 : (t == 2 ? 0 : -1)); }

template <int NPL>
__device__ __forceinline__ f32x16 mfma_split(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (NPL == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// (v0, v1) * s -> packed fp16 pair h = f16(v s) and l = f16(v s - h), round to nearest even, one instruction per value and plane:
// v_fma_mix{lo,hi}_f16 computes the fma exactly (s is a power of two, v s - h is representable) and rounds once
__device__ __forceinline__ void split2_f16(float v0, float v1, float s, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v1), "v"(s), "v"(h));
}

// two fp32 -> packed bf16 (round to nearest even); element 0 in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16_hi_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ float bf16_lo_f32(unsigned p) { return __uint_as_float(p << 16); }

// two fp32 values -> one packed 16-bit pair per plane (s: the fp16 scheme's pre-scale, unused by the bf16 scheme)
// four values at once: the two dependency chains (h, then l = f16(v s - h)) interleaved, so that no instruction waits for its predecessor
__device__ __forceinline__ void split4_f16(float v0, float v1, float v2, float v3, float s, unsigned& h01, unsigned& l01, unsigned& h23, unsigned& l23) {
    asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %6, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
        "v_fma_mixhi_f16 %2, %7, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h01), "=&v"(l01), "=&v"(h23), "=&v"(l23)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(s));
}

// (planes as separate scalars: as one [..][NPL] register array the tap-sharing kernels' conversion state spilled)
template <int NPL>
__device__ __forceinline__ void split_pair(float v0, float v1, float s, unsigned& p0, unsigned& p1, unsigned& p2) {
    if constexpr (NPL == 3) {
        p0 = cvt_pk_bf16(v0, v1);
        const float r0 = v0 - bf16_lo_f32(p0), r1 = v1 - bf16_hi_f32(p0);
        p1 = cvt_pk_bf16(r0, r1);
        p2 = cvt_pk_bf16(r0 - bf16_lo_f32(p1), r1 - bf16_hi_f32(p1));
    } else {
        split2_f16(v0, v1, s, p0, p1);
    }
}

// EGR_W_F16X2: pre-scale 2^k of the activations from their abs-max record (64 slots of float bits) - k puts the largest
// magnitude into [2^14, 2^15) (fp16 overflows at 65504), clamped to +-60 - and its inverse.  Wave-uniform.
__device__ __forceinline__ void act_prescale(const unsigned* amax_in, int lane, float& sa, float& inv) {
    unsigned am = amax_in[lane & 63];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)am, o, 64);
        am = other > am ? other : am;
    }
    const int e = (int)(__builtin_amdgcn_readfirstlane(am) >> 23);     // biased exponent (the sign bit is never set)
    int k = 141 - e;                                                   // 2^(e-127) <= amax < 2^(e-126)  ->  2^14 <= amax 2^k < 2^15
    k = k > 60 ? 60 : (k < -60 ? -60 : k);
    sa = __uint_as_float((unsigned)(127 + k) << 23);
    inv = __uint_as_float((unsigned)(127 - k) << 23);
}

// max |y| of a launch: a thread folds what it stores into `amx`; at the end one atomic per wave into one of the 64 slots
__device__ __forceinline__ void amax_flush(unsigned* amax_out, float amx, int slot) {
    // (fire and forget: reading the slot first to skip redundant atomics was measured slower - every wave then waits for an agent-scope load)
    amx = wave_max(amx);
    if ((threadIdx.x & 63) == 0 && amx > 0.f)
        __hip_atomic_fetch_max(amax_out + (slot & 63), __float_as_uint(amx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a wave-uniform pointer computed with vector instructions (64-bit multiplies have no scalar form), back in scalar registers
__device__ __forceinline__ void* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (void*)(((uint64_t)hi << 32) | lo);
}

// ---- egr_conv_tapx.hip: role-split persistent workgroups for the fp16 scheme's 3x3 forward launches
constexpr int TAPX_NO = -1000;    // tapx_try: not a launch this kernel covers, nothing was launched
// yspan / rspan: furthest float a (group's) output / residual access can touch - the epilogue addresses them with 32-bit byte offsets
int tapx_try(ConvArgs& a, int64_t yspan_floats, int64_t rspan_floats, hipStream_t stream);
int tapx_set(int on, int min_tiles, int blocks);

}  // namespace egrc
