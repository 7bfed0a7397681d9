// Weight gradient of a convolution / linear layer on fp32 MFMA (training row, SURVEY.md §8f rank 2 — a brick, the
// training step is not assembled yet).
//
//   dW[co][k] = sum_m dy[m][co] * A[m][k],   A[m][k] = x[n, ho*s-p+kh, wo*s-p+kw, ci]  (the forward im2col row),
//   m = (n, ho, wo) over all output pixels,  k = (ci/32, kh, kw, ci%32) — the forward kernel's K order, so dW comes out in
//   the packed weight layout [cout][cin/32][kh*kw][32].
//
// Here the pixel index m is the REDUCTION dimension.  A workgroup owns a BCO x 128 tile of dW and walks its share of the
// pixels in stages of 32 rows: the dy rows (BCO contiguous floats each) and the gathered x rows (4 K-chunks of 128
// contiguous bytes each, zero outside the image) stream into two LDS stages by global_load_lds; the MFMA operands are
// read k-major straight from those row-major tiles (lane = channel, so ds_read_b32 is conflict free):
//   a[i = co][k = m],  b[k = m][j = kcol]  ->  D[co][kcol] += dy[m][co] * A[m][kcol].
// The pixels are split over gridDim.y; every split writes its partial tile to a slab and a second kernel sums the slabs
// in fixed order (deterministic, no float atomics).
#include <type_traits>

#include "egr_common.h"

namespace {

struct WgradArgs {
    const float* x;
    const float* dy;
    float* ws;   // [splits][cout][K]
    float* dw;   // [cout][K]
    egr_nmap xmap, ymap;
    int n, h, w, cin, cout, kh, kw, stride, pad, ho, wo, ldx, ldy;
    int M, K, chunks, taps;
    int splits, rows_per_split;
    int tilesCO, tilesK;
    int accumulate;  // dw += sum instead of dw = sum
    int howo_shift, wo_shift;  // >= 0: ho*wo / wo are powers of two (every layer of the path) -> shifts instead of divisions
    int groups;      // same-shape problems (blockIdx.z): element strides below
    int64_t gx, gy, gw, gb;
    float* db;
    const unsigned* amax_x;    // the fp16 scheme (NPL = 2 kernels): abs-max records of x and dy (64 slots each)
    const unsigned* amax_dy;
    float* bias_ws;            // generic split kernel: per-split column sums of dy [groups][splits][cout] (NULL: the bias gradient is a separate pass)
};

__device__ __attribute__((aligned(16))) float egr_wg_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void wg_glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

constexpr int RS = 32;    // rows (pixels) per stage
constexpr int BKO = 128;  // K columns per tile = 4 chunks

template <int BCO>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int TM = BCO / 2, FM = TM / 32, FN = 2;        // waves 2 (co) x 2 (k); wave tile TM x 64
    constexpr int STAGE = RS * (BCO + BKO);                  // floats
    constexpr int DY_ROWS_PER_PIECE = 256 / BCO;             // 1 KiB = 256 floats
    constexpr int DY_PIECES = RS / DY_ROWS_PER_PIECE;        // per stage
    constexpr int A_PIECES = RS / 2;                         // 2 rows x 4 chunks x 128 B per piece
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    __shared__ int s_xoff[2][RS];
    __shared__ int s_yoff[2][RS];
    __shared__ unsigned s_mask[2][RS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float* xg = a.x + (int64_t)blockIdx.z * a.gx;
    const float* dyg = a.dy + (int64_t)blockIdx.z * a.gy;
    float* wsg = a.ws + (int64_t)blockIdx.z * a.splits * a.cout * a.K;
    const int l31 = lane & 31, half = lane >> 5;
    const int tk = blockIdx.x % a.tilesK, tco = blockIdx.x / a.tilesK;
    const int co0 = tco * BCO, chunk0 = tk * 4;
    const int split = blockIdx.y;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nstages = (m_end > m_begin) ? (m_end - m_begin + RS - 1) / RS : 0;
    const int HoWo = a.ho * a.wo;

    // per-lane constants of the A pieces: lane -> (row parity, chunk, 16-byte segment)
    const int a_chunk = (lane >> 3) & 3, a_seg = lane & 7, a_rsub = lane >> 5;
    const int chunk = chunk0 + a_chunk;
    const bool chunk_ok = chunk < a.chunks;
    int a_tap = 0, a_toff = 0;
    if (chunk_ok) {
        int cb = chunk / a.taps;
        a_tap = chunk - cb * a.taps;
        int kh = a_tap / a.kw, kw = a_tap - kh * a.kw;
        a_toff = (kh * a.w + kw) * a.ldx + cb * 32 + a_seg * 4;
    }
    // dy pieces: lane -> (row within piece, 16-byte segment)
    const int d_rsub = lane / (BCO / 4), d_seg = lane % (BCO / 4);
    const bool d_ok = (co0 + d_seg * 4) < a.cout;  // cout % 4 == 0 is required by the host

    auto decode = [&](int stage, int tb) {  // rows of `stage` -> table tb; lanes 0..7 of every wave: 32 rows
        if (lane < 8) {
            int r = wave * 8 + lane;
            int m = m_begin + stage * RS + r;
            int xo = 0, yo = -1;
            unsigned mk = 0u;
            if (stage < nstages && m < m_end) {
                int n, pix, ho, wo;
                if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
                else { n = m / HoWo; pix = m - n * HoWo; }
                if (a.wo_shift >= 0) { ho = pix >> a.wo_shift; wo = pix & (a.wo - 1); }
                else { ho = pix / a.wo; wo = pix - ho * a.wo; }
                const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
                xo = (int)egr_map(a.xmap, n) + (hi0 * a.w + wi0) * a.ldx;
                yo = (int)egr_map(a.ymap, n) + pix * a.ldy;
                // taps inside the image: kh in [kh_lo, kh_hi), kw in [kw_lo, kw_hi)
                const int kh_lo = max(0, -hi0), kh_hi = min(a.kh, a.h - hi0);
                const int kw_lo = max(0, -wi0), kw_hi = min(a.kw, a.w - wi0);
                const unsigned rowbits = (kw_hi > kw_lo) ? (((1u << (kw_hi - kw_lo)) - 1u) << kw_lo) : 0u;
                if (a.kh <= 3) {   // branch-free for the 1x1 / 3x3 layers of the path (this runs on wave 0 inside every stage)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) mk |= (kh >= kh_lo && kh < kh_hi) ? (rowbits << (kh * a.kw)) : 0u;
                } else {
                    for (int kh = kh_lo; kh < kh_hi; ++kh) mk |= rowbits << (kh * a.kw);
                }
            }
            s_xoff[tb][r] = xo;
            s_yoff[tb][r] = yo;
            s_mask[tb][r] = mk;
        }
    };

    auto issue = [&](int tb, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        float* sDy = lds + BUF * STAGE;
        float* sA = sDy + RS * BCO;
#pragma unroll
        for (int j = 0; j < DY_PIECES / 4; ++j) {
            const int piece = wave * (DY_PIECES / 4) + j;
            const int r = piece * DY_ROWS_PER_PIECE + d_rsub;
            const int yo = s_yoff[tb][r];
            const float* p = (yo >= 0 && d_ok) ? dyg + yo + co0 + d_seg * 4 : egr_wg_zero16;
            wg_glds16(p, sDy + piece * 256);
        }
#pragma unroll
        for (int j = 0; j < A_PIECES / 4; ++j) {
            const int piece = wave * (A_PIECES / 4) + j;
            const int r = piece * 2 + a_rsub;
            const bool ok = chunk_ok && s_yoff[tb][r] >= 0 && ((s_mask[tb][r] >> a_tap) & 1u);
            const float* p = ok ? xg + (s_xoff[tb][r] + a_toff) : egr_wg_zero16;
            wg_glds16(p, sA + piece * 256);
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment i of a wave covers the channels co = wm*TM + l31*FM + i (interleaved), fragment j the K columns
    // wn*64 + l31*2 + j: a lane's FM (2) operands are adjacent floats, fetched with ONE ds_read_b64 per operand side.
    auto compute = [&](auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        const float* sDy = lds + BUF * STAGE;
        const float* sA = sDy + RS * BCO;
#pragma unroll
        for (int ks = 0; ks < RS / 2; ++ks) {
            const int row = 2 * ks + half;
            float av[FM], bv[FN];
            if constexpr (FM == 2) {
                const f32x2 t = *reinterpret_cast<const f32x2*>(&sDy[row * BCO + wm * TM + l31 * 2]);
                av[0] = t[0]; av[1] = t[1];
            } else {
                av[0] = sDy[row * BCO + wm * TM + l31];
            }
            {
                const f32x2 t = *reinterpret_cast<const f32x2*>(&sA[row * BKO + wn * 64 + l31 * 2]);
                bv[0] = t[0]; bv[1] = t[1];
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // pipeline: table t+1 is decoded while stage t is multiplied; DMA of stage t+1 is issued right after the barrier
    decode(0, 0);
    __syncthreads();
    if (nstages > 0) issue(0, B0{});
    decode(1, 1);
    for (int t = 0; t < nstages; t += 2) {
        __syncthreads();                       // stage t landed (vmcnt(0)), table t+1 visible
        if (t + 1 < nstages) issue(1, B1{});
        decode(t + 2, 0);
        compute(B0{});
        if (t + 1 >= nstages) break;
        __syncthreads();                       // stage t+1 landed, table t+2 visible
        if (t + 2 < nstages) issue(0, B0{});
        decode(t + 3, 1);
        compute(B1{});
    }

    // partial tile -> slab [split][co][k]; C/D map: col = lane&31 (k column), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (co)
    // (a lane's two K columns are adjacent: one 8-byte store; K is a multiple of 32, so the pair is in range together)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int kcol = chunk0 * 32 + wn * 64 + l31 * 2;
        if (kcol >= a.K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * TM + ((r & 3) + 8 * (r >> 2) + 4 * half) * FM + i;
            if (co < a.cout) {
                f32x2 v = {acc[i][0][r], acc[i][1][r]};
                *reinterpret_cast<f32x2*>(&wsg[((int64_t)split * a.cout + co) * a.K + kcol]) = v;
            }
        }
    }
}

// ---- the same weight gradient on the bf16 matrix cores, fp32-exact: both operands are activations here, so
// both are split on the fly into (hi, mid, lo) bf16 planes while they are staged, and the six products of order <= 2 go to
// v_mfma_f32_32x32x16_bf16.  The reduction runs over pixels, so a lane's operand fragment is 8 PIXELS of one channel: the
// planes are stored pixel-major ([16 pixels][channels] bf16, 8-byte writes) and read with ds_read_b64_tr_b16, the transposing
// LDS read (a 16-lane group fetches 4 pixel rows x 16 channels and each lane receives one channel's 4 pixels).  Row stride =
// channels * 2 + 64 bytes: the four rows of a transposed read fall into four different 64-byte bank windows.
// A stage is 16 pixels; rows are requested two stages ahead, split one stage ahead behind the MFMAs (as in conv_igemm_kernel<.., true>).
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned wg_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned wg_cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float wg_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ float wg_lo(unsigned p) { return __uint_as_float(p << 16); }

// ---- the fp16 scheme for the same kernels, NPL = 2: both operands as two fp16 planes of the value times a power of
// two taken from its abs-max record (x: the forward launch's record, dy: the record of the gradient), three products (l,h) (h,l) (h,h)
// on v_mfma_f32_32x32x16_f16, the accumulators multiplied by both inverse powers when the partial tile is written.
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void wg_split2_f16(float v0, float v1, float s, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v1), "v"(s), "v"(h));
}
__device__ __forceinline__ void wg_prescale(const unsigned* rec, int lane, float& s, float& inv) {
    unsigned am = rec[lane & 63];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)am, o, 64);
        am = other > am ? other : am;
    }
    int k = 141 - (int)(__builtin_amdgcn_readfirstlane(am) >> 23);     // largest magnitude -> [2^14, 2^15)
    k = k > 60 ? 60 : (k < -60 ? -60 : k);
    s = __uint_as_float((unsigned)(127 + k) << 23);
    inv = __uint_as_float((unsigned)(127 - k) << 23);
}
template <int NPL>
__device__ __forceinline__ f32x16 wg_mfma(const wg_bf16x8& x, const wg_bf16x8& y, const f32x16& c) {
    if constexpr (NPL == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f16x8, x), __builtin_bit_cast(wg_f16x8, y), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
}

// a wave-uniform pointer computed with vector instructions (64-bit multiplies have no scalar form), back in scalar registers
__device__ __forceinline__ void* wg_uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (void*)(((uint64_t)hi << 32) | lo);
}

template <int BCO, int NPL = 3>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x6_kernel(const WgradArgs a) {
    constexpr int NPR = NPL == 2 ? 3 : 6, SPU = NPL == 2 ? 3 : 5;   // matrix instructions per fp32 product; staging steps per unit
    constexpr int PS = 16;                                       // pixels per stage = one k16 step
    constexpr int TM = BCO / 2, FM = TM / 32, FN = 2;            // waves 2 (co) x 2 (k); wave tile TM x 64
    constexpr int ROW_DY = 2 * BCO + 64, ROW_A = 2 * BKO + 64;   // bytes per pixel row of a plane
    constexpr int PL_DY = PS * ROW_DY, PL_A = PS * ROW_A;
    constexpr int STB = NPL * (PL_DY + PL_A);                    // bytes per stage (30 KiB for BCO = 128, three planes)
    constexpr int SEG_DY = BCO / 4;                              // 16-byte segments per dy row
    constexpr int ND = PS * SEG_DY / 256, NA = PS * 32 / 256;    // staging units (4 floats of one pixel) per thread
    constexpr int NUN = ND + NA;
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STB];
    __shared__ int s_xoff[4][PS];
    __shared__ int s_yoff[4][PS];
    __shared__ unsigned s_mask[4][PS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    float s_dy = 1.f, s_x = 1.f, dsc = 1.f;
    if constexpr (NPL == 2) {
        float i_dy, i_x;
        wg_prescale(a.amax_dy, lane, s_dy, i_dy);
        wg_prescale(a.amax_x, lane, s_x, i_x);
        dsc = i_dy * i_x;
    }
    const float* xg = a.x + (int64_t)blockIdx.z * a.gx;
    const float* dyg = a.dy + (int64_t)blockIdx.z * a.gy;
    float* wsg = a.ws + (int64_t)blockIdx.z * a.splits * a.cout * a.K;
    const int l31 = lane & 31, half = lane >> 5;
    const int tk = blockIdx.x % a.tilesK, tco = blockIdx.x / a.tilesK;
    const int co0 = tco * BCO, chunk0 = tk * 4;
    const int split = blockIdx.y;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nstages = (m_end > m_begin) ? (m_end - m_begin + PS - 1) / PS : 0;
    const int HoWo = a.ho * a.wo;

    // staging roles.  dy unit i: pixel d_pix + (256 / SEG_DY) * i, segment d_seg; A unit i: pixel a_pix + 8 i, chunk a_chunk, segment a_seg
    const int d_pix = tid / SEG_DY, d_seg = tid % SEG_DY;
    const bool d_ok = (co0 + d_seg * 4) < a.cout;
    const int a_pix = tid >> 5, a_chunk = (tid >> 3) & 3, a_seg = tid & 7;
    const int chunk = chunk0 + a_chunk;
    const bool chunk_ok = chunk < a.chunks;
    int a_tap = 0, a_toff = 0;
    if (chunk_ok) {
        int cb = chunk / a.taps;
        a_tap = chunk - cb * a.taps;
        int kh = a_tap / a.kw, kw = a_tap - kh * a.kw;
        a_toff = (kh * a.w + kw) * a.ldx + cb * 32 + a_seg * 4;
    }

    auto decode = [&](int stage) {  // rows of `stage` -> table stage & 3; 16 lanes of wave 0
        if (tid < PS) {
            const int r = tid, tb = stage & 3;
            int m = m_begin + stage * PS + r;
            int xo = 0, yo = -1;
            unsigned mk = 0u;
            if (stage < nstages && m < m_end) {
                int n, pix, ho, wo;
                if (a.howo_shift >= 0) { n = m >> a.howo_shift; pix = m & (HoWo - 1); }
                else { n = m / HoWo; pix = m - n * HoWo; }
                if (a.wo_shift >= 0) { ho = pix >> a.wo_shift; wo = pix & (a.wo - 1); }
                else { ho = pix / a.wo; wo = pix - ho * a.wo; }
                const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
                xo = (int)egr_map(a.xmap, n) + (hi0 * a.w + wi0) * a.ldx;
                yo = (int)egr_map(a.ymap, n) + pix * a.ldy;
                const int kh_lo = max(0, -hi0), kh_hi = min(a.kh, a.h - hi0);
                const int kw_lo = max(0, -wi0), kw_hi = min(a.kw, a.w - wi0);
                const unsigned rowbits = (kw_hi > kw_lo) ? (((1u << (kw_hi - kw_lo)) - 1u) << kw_lo) : 0u;
                if (a.kh <= 3) {   // branch-free for the 1x1 / 3x3 layers of the path (this runs on wave 0 inside every stage)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) mk |= (kh >= kh_lo && kh < kh_hi) ? (rowbits << (kh * a.kw)) : 0u;
                } else {
                    for (int kh = kh_lo; kh < kh_hi; ++kh) mk |= rowbits << (kh * a.kw);
                }
            }
            s_xoff[tb][r] = xo;
            s_yoff[tb][r] = yo;
            s_mask[tb][r] = mk;
        }
    };

    // Both operands come through 2-GiB buffer windows: a row that does not exist (beyond M, beyond cout, a tap outside the image)
    // is an out-of-range offset - the load returns zeros and touches no memory; no pointer selects, no zero buffer.  The activation
    // window starts `abias` bytes before the group's base so that halo rows have non-negative offsets.
    const int abias = (a.pad * a.w + a.pad + 1) * a.ldx * 4;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(wg_uniform_ptr(dyg), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxa = __builtin_amdgcn_make_buffer_rsrc(wg_uniform_ptr(reinterpret_cast<const char*>(xg) - abias), 0, 0x80000000u,
                                                                         0x00020000);
    const int d_col = (co0 + d_seg * 4) * 4;          // byte offset of this thread's dy channels
    const int a_col = a_toff * 4 + abias;              // byte offset of this thread's tap / channel chunk
    f32x4 xr[2][NUN];
    auto request = [&](int stage, auto set_tag) {   // global loads of `stage` (its table must be visible) into register set SET
        constexpr int SET = decltype(set_tag)::value;
        const int tb = stage & 3;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int yo = s_yoff[tb][d_pix + (256 / SEG_DY) * i];
            const int vo = (yo >= 0 && d_ok) ? yo * 4 + d_col : (int)0x80000000;
            xr[SET][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, vo, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = a_pix + 8 * i;
            const bool ok = chunk_ok && s_yoff[tb][r] >= 0 && ((s_mask[tb][r] >> a_tap) & 1u);
            const int vo = ok ? s_xoff[tb][r] * 4 + a_col : (int)0x80000000;
            xr[SET][ND + i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxa, vo, 0, 0));
        }
    };
    // slice k of the staging work: unit k / 5; step 0/1 = hi parts + residuals of the unit's two pairs, 2/3 = mid + lo, 4 = the writes
    unsigned sh_[NUN][2], sm_[NUN][2], sl_[NUN][2];
    float ra_[NUN][2], rb_[NUN][2];
    // bias gradient folded in: the tiles of the first K column add up the dy rows they stage anyway (fp32 per thread, the threads of a
    // channel quad through LDS at the end, the splits in double by wgrad_bias_reduce_kernel) - no second pass over dy
    const bool do_bias = a.bias_ws != nullptr && tk == 0;
    f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
    auto slice = [&](auto set_tag, auto buf_tag, int k) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr int BUF = decltype(buf_tag)::value;
        const int u = k / SPU, q = k % SPU;
        if (q == 0 && u < ND && do_bias) bacc += xr[SET][u];
        if constexpr (NPL == 2) {      // steps 0/1: the unit's two pairs, 2: the writes
            if (q < 2) {
                wg_split2_f16(xr[SET][u][2 * q], xr[SET][u][2 * q + 1], u < ND ? s_dy : s_x, sh_[u][q], sl_[u][q]);
            } else {
                uint8_t* dst;
                int pl;
                if (u < ND) { dst = lds + BUF * STB + (d_pix + (256 / SEG_DY) * u) * ROW_DY + d_seg * 8; pl = PL_DY; }
                else { dst = lds + BUF * STB + NPL * PL_DY + (a_pix + 8 * (u - ND)) * ROW_A + (a_chunk * 32 + a_seg * 4) * 2; pl = PL_A; }
                *reinterpret_cast<wg_u32x2*>(dst) = wg_u32x2{sh_[u][0], sh_[u][1]};
                *reinterpret_cast<wg_u32x2*>(dst + pl) = wg_u32x2{sl_[u][0], sl_[u][1]};
            }
        } else if (q < 2) {
            const float v0 = xr[SET][u][2 * q], v1 = xr[SET][u][2 * q + 1];
            sh_[u][q] = wg_cvt_pk(v0, v1);
            ra_[u][q] = v0 - wg_lo(sh_[u][q]);
            rb_[u][q] = v1 - wg_hi(sh_[u][q]);
        } else if (q < 4) {
            const int t = q - 2;
            sm_[u][t] = wg_cvt_pk(ra_[u][t], rb_[u][t]);
            sl_[u][t] = wg_cvt_pk(ra_[u][t] - wg_lo(sm_[u][t]), rb_[u][t] - wg_hi(sm_[u][t]));
        } else {
            uint8_t* dst;
            int pl;
            if (u < ND) { dst = lds + BUF * STB + (d_pix + (256 / SEG_DY) * u) * ROW_DY + d_seg * 8; pl = PL_DY; }
            else { dst = lds + BUF * STB + NPL * PL_DY + (a_pix + 8 * (u - ND)) * ROW_A + (a_chunk * 32 + a_seg * 4) * 2; pl = PL_A; }
            *reinterpret_cast<wg_u32x2*>(dst) = wg_u32x2{sh_[u][0], sh_[u][1]};
            *reinterpret_cast<wg_u32x2*>(dst + pl) = wg_u32x2{sm_[u][0], sm_[u][1]};
            *reinterpret_cast<wg_u32x2*>(dst + 2 * pl) = wg_u32x2{sl_[u][0], sl_[u][1]};
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read address of this lane inside a plane: row (pixel) 8*half + 4*rd + q, 4 channels starting at 16*((lane>>4)&1) + 4*p
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    auto frag = [&](const uint8_t* plane, int row_bytes, int c0) {
        wg_bf16x8 v;
        const uint8_t* p = plane + (8 * half + tq) * row_bytes + (c0 + 16 * tg + 4 * tp) * 2;
        const wg_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(p));
        const wg_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(p + 4 * row_bytes));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 both = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        __builtin_memcpy(&v, &both, 16);
        return v;
    };
    constexpr int NS = SPU * NUN, NM = NPR * FM * FN;
    auto stage = [&](auto buf_tag, auto conv_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool conv = decltype(conv_tag)::value;   // compile time: the slice bookkeeping below must fold to constants
        using NB = std::integral_constant<int, BUF ^ 1>;
        const uint8_t* st = lds + BUF * STB;
        wg_bf16x8 af[FM][NPL], bf[FN][NPL];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) af[i][pl] = frag(st + pl * PL_DY, ROW_DY, wm * TM + 32 * i);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bf[j][pl] = frag(st + NPL * PL_DY + pl * PL_A, ROW_A, wn * 64 + 32 * j);
        constexpr int PA[6] = {NPL == 2 ? 1 : 2, 0, NPL == 2 ? 0 : 1, 1, 0, 0}, PB[6] = {0, NPL == 2 ? 1 : 2, NPL == 2 ? 0 : 1, 0, 1, 0};
        int n = 0, done = 0;
#pragma unroll
        for (int t = 0; t < NPR; ++t)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j, ++n) {
                    acc[i][j] = wg_mfma<NPL>(af[i][PA[t]], bf[j][PB[t]], acc[i][j]);
                    if constexpr (conv) {
                        const int upto = ((n + 1) * NS + NM - 1) / NM;
#pragma unroll
                        for (int k = 0; k < NS; ++k)
                            if (k >= done && k < upto) slice(NB{}, NB{}, k);
                        done = upto > done ? upto : done;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
    };

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // stage t lives in buffer t & 1 and register set t & 1; its rows are requested during stage t-2, split during stage t-1.
    // Hand-over between stages: this wave's LDS traffic done, then the workgroup barrier.  NOT __syncthreads(), which also waits
    // for vmcnt(0), i.e. for the rows just requested for the stage after next (their latency would be exposed in every stage).
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    decode(0);
    decode(1);
    decode(2);
    lds_barrier();
    if (nstages > 0) {
        request(0, B0{});
#pragma unroll
        for (int k = 0; k < NS; ++k) slice(B0{}, B0{}, k);
        request(1, B1{});
    }
    lds_barrier();
    int t = 0;
    // full pairs of stages as branch-free straight-line code (every stage of the pair splits its successor and requests the
    // one after): with the end-of-range tests inside, the two variants of each stage met in a phi and the compiler moved the
    // 64 accumulator registers twice per pair
    for (; t + 3 < nstages; t += 2) {
        decode(t + 3);
        request(t + 2, B0{});
        stage(B0{}, std::true_type{});
        lds_barrier();
        decode(t + 4);
        request(t + 3, B1{});
        stage(B1{}, std::true_type{});
        lds_barrier();
    }
    for (; t < nstages; t += 2) {   // the last one to three stages
        // even stage: buffer 0; set 1 (stage t+1) -> buffer 1; set 0 requests stage t+2
        decode(t + 3);
        if (t + 2 < nstages) request(t + 2, B0{});
        if (t + 1 < nstages) stage(B0{}, std::true_type{});
        else stage(B0{}, std::false_type{});
        lds_barrier();
        if (t + 1 >= nstages) break;
        decode(t + 4);
        if (t + 3 < nstages) request(t + 3, B1{});
        if (t + 2 < nstages) stage(B1{}, std::true_type{});
        else stage(B1{}, std::false_type{});
        lds_barrier();
    }

    if (do_bias) {          // (the stage buffers are free: every wave is behind the last hand-over)
        float* const sb = reinterpret_cast<float*>(lds);
        *reinterpret_cast<f32x4*>(sb + tid * 4) = bacc;
        __syncthreads();
        if (tid < SEG_DY) {
            f32x4 v = *reinterpret_cast<const f32x4*>(sb + tid * 4);
            for (int p = 1; p < 256 / SEG_DY; ++p) v += *reinterpret_cast<const f32x4*>(sb + (p * SEG_DY + tid) * 4);
            float* const dst = a.bias_ws + ((int64_t)blockIdx.z * a.splits + split) * a.cout + co0 + tid * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (co0 + tid * 4 + e < a.cout) dst[e] = v[e];
        }
    }
    // partial tile -> slab [split][co][k]; C/D map: col = lane&31 (k column), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (co)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int kcol = chunk0 * 32 + wn * 64 + 32 * j + l31;
            if (kcol >= a.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < a.cout) wsg[((int64_t)split * a.cout + co) * a.K + kcol] = NPL == 2 ? acc[i][j][r] * dsc : acc[i][j][r];
            }
        }
}

// ---- 3x3 / stride 1 / pad 1 on the bf16 matrix cores with the taps SHARED.  In conv_wgrad_x6_kernel every (tap, pixel) row of
// the im2col matrix is fetched and split on its own: each input pixel is split nine times, and with only 64 output channels to
// amortise that over, the kernel is VALU-bound (96 TFLOP/s on layer1 against 150-170 on the wide layers).  Here a stage is 16
// consecutive output pixels of ONE image row; the three input rows around them (18 pixels each, zeros outside the image) are split
// once into pixel-major bf16 planes, and the nine taps are nine row-shifted windows of those planes - a shift by whole rows of
// the transposing LDS read, free.  (WS = 8: images 8 pixels wide - a stage is two image rows, four input rows of 10 pixels.)  A workgroup owns CO_T output channels x CB_T 32-channel input chunks x all 9 taps =
// 36 accumulator fragments of 32 x 32 (9 per wave: one dy fragment against the 9 taps of one input chunk), i.e. 3.6 split
// instructions per MFMA instead of 11.  Rows are requested at the start of a stage and split behind its second third.
template <int CO_T, int CB_T, int WS, int NPL = 3>
__global__ __launch_bounds__(256, 2) void conv_wgrad3_x6_kernel(const WgradArgs a) {
    constexpr int NPR = NPL == 2 ? 3 : 6, SPU = NPL == 2 ? 3 : 5;
    constexpr int PS = 16;                                       // output pixels per stage: PS / WS image rows of WS pixels
    constexpr int XP = WS + 2, XR = PS / WS + 2;                 // input pixels per row / input rows, with the halo
    static_assert(WS == 16 || WS == 8, "16 pixels of one row, or two rows of an 8-pixel-wide image");
    constexpr int CX = CB_T * 32;                                // input channels of the tile
    constexpr int ROW_DY = 2 * CO_T + 64, ROW_X = 2 * CX + 32;   // bytes per pixel row of a plane (padding: see frag())
    constexpr int PL_DY = PS * ROW_DY, PL_X = XR * XP * ROW_X;
    constexpr int STB = NPL * (PL_DY + PL_X);                    // bytes per stage
    constexpr int SEG_DY = CO_T / 4, SEG_X = CX / 4;
    constexpr int UD = PS * SEG_DY, UX = XR * XP * SEG_X;         // staging units (4 floats of one pixel)
    constexpr int ND = UD / 256, NX = (UX + 255) / 256, NUN = ND + NX;
    static_assert(UD % 256 == 0 && (CO_T / 32) * CB_T == 4, "tile shape");
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s_dy = 1.f, s_x = 1.f, dsc = 1.f;
    if constexpr (NPL == 2) {
        float i_dy, i_x;
        wg_prescale(a.amax_dy, lane, s_dy, i_dy);
        wg_prescale(a.amax_x, lane, s_x, i_x);
        dsc = i_dy * i_x;
    }
    const int mfrag = wave / CB_T, cbl = wave % CB_T;
    const int l31 = lane & 31, half = lane >> 5;
    const int tilesCB = a.cin / CX;
    const int tcb = blockIdx.x % tilesCB, tco = blockIdx.x / tilesCB;
    const int co0 = tco * CO_T, cb0 = tcb * CB_T;
    const int split = blockIdx.y;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nstages = (m_end > m_begin) ? (m_end - m_begin) / PS : 0;      // M and rows_per_split are multiples of 16
    const int HoWo = a.ho * a.wo;
    const float* xg = a.x + (int64_t)blockIdx.z * a.gx;
    const float* dyg = a.dy + (int64_t)blockIdx.z * a.gy;
    float* wsg = a.ws + (int64_t)blockIdx.z * a.splits * a.cout * a.K;

    // staging roles (stage independent): element offset relative to the stage's first output pixel, LDS destination
    int d_off[ND], d_lds[ND], x_off[NX], x_lds[NX], x_row[NX], x_px[NX];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int u = tid + 256 * i, dp = u / SEG_DY, ds = u % SEG_DY;
        d_off[i] = (dp * a.ldy + co0 + ds * 4) * 4;
        d_lds[i] = dp * ROW_DY + ds * 8;
    }
    const int abias = (a.w + 1) * a.ldx * 4;       // the activation window starts one row + one pixel early: halo offsets stay >= 0
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int u = tid + 256 * i, xs = u % SEG_X, xp = (u / SEG_X) % XP, xr_ = u / (SEG_X * XP);
        x_row[i] = (u < UX) ? xr_ - 1 : -100000;       // never inside the image
        x_px[i] = xp - 1;
        x_off[i] = (((xr_ - 1) * a.w + (xp - 1)) * a.ldx + cb0 * 32 + xs * 4) * 4 + abias;
        x_lds[i] = (xr_ * XP + xp) * ROW_X + xs * 8;
    }
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(wg_uniform_ptr(dyg), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxa = __builtin_amdgcn_make_buffer_rsrc(wg_uniform_ptr(reinterpret_cast<const char*>(xg) - abias), 0, 0x80000000u,
                                                                         0x00020000);
    f32x4 xr[NUN];
    auto request = [&](int stage) {     // everything about the stage itself is wave-uniform
        const int m0 = m_begin + stage * PS;
        int n, pix, y, x0;
        if (a.howo_shift >= 0) { n = m0 >> a.howo_shift; pix = m0 & (HoWo - 1); }
        else { n = m0 / HoWo; pix = m0 - n * HoWo; }
        if (a.wo_shift >= 0) { y = pix >> a.wo_shift; x0 = pix & (a.wo - 1); }
        else { y = pix / a.wo; x0 = pix - y * a.wo; }
        const int so_y = __builtin_amdgcn_readfirstlane(((int)egr_map(a.ymap, n) + pix * a.ldy) * 4);
        const int so_x = __builtin_amdgcn_readfirstlane(((int)egr_map(a.xmap, n) + (y * a.w + x0) * a.ldx) * 4);
#pragma unroll
        for (int i = 0; i < ND; ++i) xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, d_off[i], so_y, 0));
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool ok = (unsigned)(y + x_row[i]) < (unsigned)a.h && (unsigned)(x0 + x_px[i]) < (unsigned)a.w;
            xr[ND + i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxa, ok ? x_off[i] : (int)0x80000000, so_x, 0));
        }
    };
    // slice k of the staging work: unit k / 5; step 0/1 = hi parts + residuals of the unit's two pairs, 2/3 = mid + lo, 4 = the writes
    unsigned sh_[NUN][2], sm_[NUN][2], sl_[NUN][2];
    float ra_[NUN][2], rb_[NUN][2];
    auto slice = [&](int base, int k) {
        const int u = k / SPU, q = k % SPU;
        if constexpr (NPL == 2) {
            if (q < 2) {
                wg_split2_f16(xr[u][2 * q], xr[u][2 * q + 1], u < ND ? s_dy : s_x, sh_[u][q], sl_[u][q]);
            } else if (u < ND || NX * 256 == UX || tid + 256 * (u - ND) < UX) {
                uint8_t* dst;
                int pl;
                if (u < ND) { dst = lds + base + d_lds[u]; pl = PL_DY; }
                else { dst = lds + base + NPL * PL_DY + x_lds[u - ND]; pl = PL_X; }
                *reinterpret_cast<wg_u32x2*>(dst) = wg_u32x2{sh_[u][0], sh_[u][1]};
                *reinterpret_cast<wg_u32x2*>(dst + pl) = wg_u32x2{sl_[u][0], sl_[u][1]};
            }
        } else if (q < 2) {
            const float v0 = xr[u][2 * q], v1 = xr[u][2 * q + 1];
            sh_[u][q] = wg_cvt_pk(v0, v1);
            ra_[u][q] = v0 - wg_lo(sh_[u][q]);
            rb_[u][q] = v1 - wg_hi(sh_[u][q]);
        } else if (q < 4) {
            const int t = q - 2;
            sm_[u][t] = wg_cvt_pk(ra_[u][t], rb_[u][t]);
            sl_[u][t] = wg_cvt_pk(ra_[u][t] - wg_lo(sm_[u][t]), rb_[u][t] - wg_hi(sm_[u][t]));
        } else if (u < ND || NX * 256 == UX || tid + 256 * (u - ND) < UX) {
            uint8_t* dst;
            int pl;
            if (u < ND) { dst = lds + base + d_lds[u]; pl = PL_DY; }
            else { dst = lds + base + NPL * PL_DY + x_lds[u - ND]; pl = PL_X; }
            *reinterpret_cast<wg_u32x2*>(dst) = wg_u32x2{sh_[u][0], sh_[u][1]};
            *reinterpret_cast<wg_u32x2*>(dst + pl) = wg_u32x2{sm_[u][0], sm_[u][1]};
            *reinterpret_cast<wg_u32x2*>(dst + 2 * pl) = wg_u32x2{sl_[u][0], sl_[u][1]};
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read address of this lane inside a plane: row (pixel) 8*half + 4*rd + q, 4 channels starting at 16*((lane>>4)&1) + 4*p.
    // Row strides of 2*channels + 64 / + 32 bytes put the four 32-byte rows of one 16-lane read into four different bank ranges.
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    auto frag = [&](const uint8_t* plane, int row_bytes, int c0, int hrows) {   // hrows: LDS rows between pixel 0 and pixel 8 of the stage
        wg_bf16x8 v;
        const uint8_t* p = plane + (hrows * half + tq) * row_bytes + (c0 + 16 * tg + 4 * tp) * 2;
        const wg_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(p));
        const wg_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(p + 4 * row_bytes));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 both = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        __builtin_memcpy(&v, &both, 16);
        return v;
    };
    constexpr int HB = (WS == 16) ? 8 : XP;      // pixel 8 of the stage: 8 rows further in the same image row, or the start of the next one
    constexpr int NS = SPU * NUN, NM = 9 * NPR, S0 = NM / 3;
    static_assert(NS <= NM - S0, "one slice per MFMA");
    auto stage = [&](int cur, int nxt, auto conv_tag) {
        constexpr bool conv = decltype(conv_tag)::value;
        const uint8_t* st = lds + cur;
        wg_bf16x8 af[NPL], bf[2][NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) af[pl] = frag(st + pl * PL_DY, ROW_DY, mfrag * 32, 8);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bf[0][pl] = frag(st + NPL * PL_DY + pl * PL_X, ROW_X, cbl * 32, HB);
        constexpr int PA[6] = {NPL == 2 ? 1 : 2, 0, NPL == 2 ? 0 : 1, 1, 0, 0}, PB[6] = {0, NPL == 2 ? 1 : 2, NPL == 2 ? 0 : 1, 0, 1, 0};
        int n = 0, done = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) {
                const int kh = (tap + 1) / 3, kw = (tap + 1) % 3;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) bf[(tap + 1) & 1][pl] = frag(st + NPL * PL_DY + pl * PL_X + (kh * XP + kw) * ROW_X, ROW_X, cbl * 32, HB);
            }
#pragma unroll
            for (int t = 0; t < NPR; ++t, ++n) {
                acc[tap] = wg_mfma<NPL>(af[PA[t]], bf[tap & 1][PB[t]], acc[tap]);
                if constexpr (conv) {
                    // (n, hence upto and done, are compile-time values once the tap / product loops are unrolled; NS <= NM - S0:
                    // at most one slice per MFMA - no inner loop over the slices, which the unroller gives up on at this size)
                    const int upto = (n + 1 <= S0) ? 0 : ((n + 1 - S0) * NS + (NM - S0) - 1) / (NM - S0);
                    if (done < upto) { slice(nxt, done); ++done; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    if (nstages > 0) {
        request(0);
#pragma unroll
        for (int k = 0; k < NS; ++k) slice(0, k);
    }
    lds_barrier();
    int cur = 0;
    for (int t = 0; t + 1 < nstages; ++t) {
        request(t + 1);
        stage(cur, STB - cur, std::true_type{});
        lds_barrier();
        cur = STB - cur;
    }
    if (nstages > 0) stage(cur, 0, std::false_type{});

    // partial tile -> slab [split][co][k], k = (chunk, tap, ci % 32); C/D map: col = lane&31 (k column), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (co)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int kcol = ((cb0 + cbl) * 9 + tap) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mfrag * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            wsg[((int64_t)split * a.cout + co) * a.K + kcol] = NPL == 2 ? acc[tap][r] * dsc : acc[tap][r];
        }
    }
}

// ---- small 1x1 weight gradients (the Linear layers of the transformer layers and the heads: a few hundred to a few thousand rows).
// On the tiled kernel above such a problem is three launches (split tiles into slabs, the slab sum, the bias column sum) of 23 + 7 + 5 us
// each - 53 of them in a training step.  Here ONE launch: a workgroup owns a 32 x 32 tile of dW, its eight waves take the row pairs
// round-robin; lane (r, h) reads dy[m = 2p + h][co0 + r] and x[m][k0 + r] straight from global memory (two 128-byte row segments per
// wave load), v_mfma_f32_32x32x2_f32 sums the pair, sixteen pairs of a wave in flight.  The eight partial tiles are added in wave
// order through LDS (deterministic) and written to dW (or added to it); the tiles of the first K column also sum their dy columns in
// double (the bias gradient, the arithmetic of colsum_small_kernel up to the order of the rows).
constexpr int WSM_NW = 8;
__global__ __launch_bounds__(64 * WSM_NW) void wgrad_small_kernel(const WgradArgs a) {
    __shared__ float s_part[WSM_NW][32][33];
    __shared__ double s_db[WSM_NW][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int tilesK = a.K >> 5;
    const int tco = blockIdx.x / tilesK, tk = blockIdx.x - tco * tilesK;
    const int grp = blockIdx.z;
    const float* const xg = a.x + (int64_t)grp * a.gx + tk * 32 + r;
    const int co = tco * 32 + r;
    const bool co_ok = co < a.cout;
    const float* const dyg = a.dy + (int64_t)grp * a.gy + (co_ok ? co : 0);
    const int HoWo = a.ho * a.wo;
    const bool plain = HoWo == 1 && a.xmap.n_inner >= a.n && a.ymap.n_inner >= a.n;    // Linear layers: row m is image m
    const bool want_db = a.db != nullptr && tk == 0;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    double bsum = 0.0;
    const int npairs = (a.M + 1) >> 1;
    constexpr int U = 16;
    for (int p0 = wave; p0 < npairs; p0 += WSM_NW * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = 2 * (p0 + WSM_NW * u) + h;
            av[u] = 0.f;
            bv[u] = 0.f;
            if (m < a.M) {
                int64_t xo, yo;
                if (plain) {
                    xo = (int64_t)m * a.xmap.stride_inner;
                    yo = (int64_t)m * a.ymap.stride_inner;
                } else {
                    const int n = (a.howo_shift >= 0) ? (m >> a.howo_shift) : m / HoWo, pix = m - n * HoWo;
                    xo = egr_map(a.xmap, n) + (int64_t)pix * a.ldx;
                    yo = egr_map(a.ymap, n) + (int64_t)pix * a.ldy;
                }
                bv[u] = xg[xo];
                if (co_ok) av[u] = dyg[yo];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            if (want_db) bsum += (double)av[u];
        }
    }
    // C/D map of the 32x32 MFMA: col (k) = lane & 31, row (co) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 16; ++i) s_part[wave][(i & 3) + 8 * (i >> 2) + 4 * h][r] = acc[i];
    if (want_db) {
        bsum += __shfl_xor(bsum, 32, 64);          // the two rows of a pair
        if (h == 0) s_db[wave][r] = bsum;
    }
    __syncthreads();
    float* const dwg = a.dw + (int64_t)grp * a.gw;
    for (int e = tid; e < 32 * 32; e += 64 * WSM_NW) {
        const int row = e >> 5, c = e & 31;
        const int o = tco * 32 + row;
        if (o >= a.cout) continue;
        float v = s_part[0][row][c];
#pragma unroll
        for (int w = 1; w < WSM_NW; ++w) v += s_part[w][row][c];
        float* const dst = dwg + (int64_t)o * a.K + tk * 32 + c;
        *dst = a.accumulate ? *dst + v : v;
    }
    if (want_db && tid < 32 && tco * 32 + tid < a.cout) {
        double v = s_db[0][tid];
#pragma unroll
        for (int w = 1; w < WSM_NW; ++w) v += s_db[w][tid];
        float* const dst = a.db + (int64_t)grp * a.gb + tco * 32 + tid;
        *dst = a.accumulate ? *dst + (float)v : (float)v;
    }
}

// Sum of the split slabs in a fixed order (deterministic).  A block owns 16 float4 outputs; 16 split lanes walk the slabs
// with stride 16 (independent loads in flight instead of one dependent chain of `splits` loads), then an LDS tree in
// lane order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a) {
    __shared__ f32x4 red[256];
    const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t total = (int64_t)a.cout * a.K;
    const int64_t idx = ((int64_t)blockIdx.x * 16 + ol) * 4;
    const float* wsg = a.ws + (int64_t)blockIdx.z * a.splits * total;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (idx < total) {      // four slab loads in flight, added in slab order (a loop of one dependent load per slab is a latency chain)
        int sp = sl;
        for (; sp + 48 < a.splits; sp += 64) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(wsg + (int64_t)(sp + 16 * u) * total + idx);
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; sp < a.splits; sp += 16) s += *reinterpret_cast<const f32x4*>(wsg + (int64_t)sp * total + idx);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && idx < total) {
        for (int k = 1; k < 16; ++k) s += red[k * 16 + ol];
        f32x4* o = reinterpret_cast<f32x4*>(a.dw + (int64_t)blockIdx.z * a.gw + idx);
        if (a.accumulate) s += *o;
        *o = s;
    }
}

// bias gradient from the per-split column sums the generic split kernel left: one thread per channel, the splits in order, in double
__global__ __launch_bounds__(256) void wgrad_bias_reduce_kernel(const WgradArgs a) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.cout) return;
    const float* const p = a.bias_ws + (int64_t)blockIdx.z * a.splits * a.cout + c;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int sp = 0;
    for (; sp + 4 <= a.splits; sp += 4) {           // four loads in flight; four interleaved chains summed at the end (fixed order)
        const float v0 = p[(int64_t)sp * a.cout], v1 = p[(int64_t)(sp + 1) * a.cout], v2 = p[(int64_t)(sp + 2) * a.cout], v3 = p[(int64_t)(sp + 3) * a.cout];
        s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
    }
    for (; sp < a.splits; ++sp) s0 += (double)p[(int64_t)sp * a.cout];
    const double s = (s0 + s1) + (s2 + s3);
    float* const o = a.db + (int64_t)blockIdx.z * a.gb + c;
    *o = a.accumulate ? *o + (float)s : (float)s;
}

// per-channel sum over rows (bias gradient).  256 threads = (256 / (c/4)) row lanes x (c/4) float4 column lanes over a
// 1024-channel column block (blockIdx.y); four independent rows in flight per lane; LDS reduce over the row lanes.
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int64_t rows, int c, int ld, float* partial, int rows_per_block,
                                                     int64_t gx) {
    __shared__ f32x4 red[256];
    x += (int64_t)blockIdx.z * gx;
    partial += (int64_t)blockIdx.z * gridDim.x * c;
    const int c0 = blockIdx.y * 1024;
    const int cw = min(1024, c - c0);              // channels of this column block (multiple of 4)
    int c4 = cw >> 2;
    int lanes = 1;
    while (lanes < c4) lanes <<= 1;                // power of two >= c4, <= 256
    const int rpi = 256 / lanes;
    const int cl = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (cl < c4) {
        const float* p = x + c0 + cl * 4;
        int64_t r = r0 + rl;
        for (; r + 3 * rpi < r1; r += 4 * rpi) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + r * ld);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (r + rpi) * ld);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (r + 2 * rpi) * ld);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (r + 3 * rpi) * ld);
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; r < r1; r += rpi) s0 += *reinterpret_cast<const f32x4*>(p + r * ld);
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && cl < c4) {
        f32x4 t = red[cl];
        for (int k = 1; k < rpi; ++k) t += red[k * lanes + cl];
        *reinterpret_cast<f32x4*>(partial + (int64_t)blockIdx.x * c + c0 + cl * 4) = t;
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int nblk, int c, float* out, int accumulate, int64_t gb) {
    __shared__ double red[256];
    const int ch = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;   // 16 channels x 16 slab lanes
    partial += (int64_t)blockIdx.y * nblk * c;
    out += (int64_t)blockIdx.y * gb;
    double s = 0.0;
    if (ch < c)
        for (int b = sl; b < nblk; b += 16) s += partial[(int64_t)b * c + ch];
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && ch < c) {
        for (int k = 1; k < 16; ++k) s += red[k * 16 + (threadIdx.x & 15)];
        out[ch] = (float)s + (accumulate ? out[ch] : 0.f);
    }
}

// bias gradient of a few rows (<= 4096: the token-sized Linear layers) in ONE launch: 16 channels x 16 row lanes per workgroup,
// double accumulation like colsum_final_kernel
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* x, int rows, int c, int ld, float* out, int accumulate, int64_t gx, int64_t gb) {
    __shared__ double red[256];
    const int ch = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
    x += (int64_t)blockIdx.y * gx;
    out += (int64_t)blockIdx.y * gb;
    double s = 0.0;
    if (ch < c) {
        int r = sl;
        for (; r + 48 < rows; r += 64) {
            const float v0 = x[(int64_t)r * ld + ch], v1 = x[(int64_t)(r + 16) * ld + ch], v2 = x[(int64_t)(r + 32) * ld + ch], v3 = x[(int64_t)(r + 48) * ld + ch];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; r < rows; r += 16) s += x[(int64_t)r * ld + ch];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && ch < c) {
        for (int k = 1; k < 16; ++k) s += red[k * 16 + (threadIdx.x & 15)];
        out[ch] = (float)s + (accumulate ? out[ch] : 0.f);
    }
}

int g_last_kernel = 0;
int g_last_h2 = 0;

}  // namespace

// diagnostic (tests): which kernel the last egr_conv2d_wgrad_f32 call launched - 0 fp32 MFMA, 1 split-bf16 generic,
// 2 split-bf16 3x3 tap-sharing 64 x 2 chunks, 3 the same 128 x 1 chunk, 4 the small 1x1 kernel (one launch, fp32)
extern "C" int egr_wgrad_last_kernel(void) { return g_last_kernel; }
extern "C" int egr_wgrad_last_h2(void) { return g_last_h2; }

extern "C" int egr_conv2d_wgrad_ex_f32(const egr_conv_desc* dd, const float* x, const float* dy, float* dw, float* db, float* workspace,
                                       size_t workspace_floats, int32_t accumulate, const uint32_t* amax_x, const uint32_t* amax_dy, void* stream);

extern "C" int egr_conv2d_wgrad_f32(const egr_conv_desc* dd, const float* x, const float* dy, float* dw, float* db,
                                    float* workspace, size_t workspace_floats, int32_t accumulate, void* stream) {
    if (dd && (dd->w_format & EGR_W_F16X2)) return EGR_EINVAL;      // (the fp16 scheme needs the records: egr_conv2d_wgrad_ex_f32)
    return egr_conv2d_wgrad_ex_f32(dd, x, dy, dw, db, workspace, workspace_floats, accumulate, nullptr, nullptr, stream);
}

extern "C" int egr_conv2d_wgrad_ex_f32(const egr_conv_desc* dd, const float* x, const float* dy, float* dw, float* db, float* workspace,
                                       size_t workspace_floats, int32_t accumulate, const uint32_t* amax_x, const uint32_t* amax_dy, void* stream) {
    if (!dd || !x || !dy || !dw || !workspace) return EGR_ENULL;
    if ((dd->w_format & EGR_W_F16X2) && (!amax_x || !amax_dy)) return EGR_ENULL;
    if ((dd->w_format & EGR_W_F16X2) && ((((uintptr_t)amax_x) | ((uintptr_t)amax_dy)) & 3)) return EGR_EINVAL;
    const egr_conv_desc& d = *dd;
    if (d.groups < 1 || d.groups > 65535 || d.transposed || d.out_nchw) return EGR_EINVAL;
    const int G = d.groups;
    if (d.cin <= 0 || d.cin % 32 != 0 || d.cout <= 0 || d.cout % 4 != 0 || d.kh * d.kw > 32 || d.kh <= 0 || d.kw <= 0) return EGR_EINVAL;
    if (d.ldx % 4 != 0 || d.ldy % 4 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dw & 15) || ((uintptr_t)workspace & 15))
        return EGR_EINVAL;
    if (d.xmap.n_inner <= 0 || d.ymap.n_inner <= 0) return EGR_EINVAL;
    WgradArgs a;
    a.x = x; a.dy = dy; a.ws = workspace; a.dw = dw;
    a.xmap = d.xmap; a.ymap = d.ymap;
    a.n = d.n; a.h = d.h; a.w = d.w; a.cin = d.cin; a.cout = d.cout; a.kh = d.kh; a.kw = d.kw; a.stride = d.stride; a.pad = d.pad;
    a.ho = d.ho; a.wo = d.wo; a.ldx = d.ldx; a.ldy = d.ldy;
    int64_t M64 = (int64_t)d.n * d.ho * d.wo;
    if (M64 <= 0 || M64 >= (1LL << 31)) return EGR_EINVAL;
    a.M = (int)M64;
    a.taps = d.kh * d.kw;
    a.chunks = a.taps * (d.cin / 32);
    a.K = a.chunks * 32;
    a.accumulate = accumulate;
    { auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return ((1 << l) == v) ? l : -1; };
      a.howo_shift = lg(d.ho * d.wo); a.wo_shift = lg(d.wo); }
    a.groups = G; a.gx = d.gx; a.gy = d.gy; a.gw = d.gw; a.gb = d.gp; a.db = db;
    a.amax_x = amax_x; a.amax_dy = amax_dy;
    a.bias_ws = nullptr;
    // large problems run on the bf16 matrix cores with exact three-way operand splits (same result class as the fp32 kernel);
    // w_format == EGR_W_BF16X3 requests it, small ones stay on the fp32 kernel (latency-bound)
    // (its operands are addressed through 2-GiB buffer windows: larger tensors stay on the fp32 kernel)
    auto span = [](const egr_nmap& m, int n) {
        const int o = (n - 1) / m.n_inner, i = (n - 1 < m.n_inner ? n - 1 : m.n_inner - 1);
        return (int64_t)i * m.stride_inner + (int64_t)o * m.stride_outer;
    };
    const bool fits = (span(d.xmap, d.n) + (int64_t)(d.h * d.w + 2 * (d.pad * d.w + d.pad + 1)) * d.ldx) * 4 + 64 < (1LL << 31) &&
                      (span(d.ymap, d.n) + (int64_t)d.ho * d.wo * d.ldy) * 4 + 64 < (1LL << 31);
    const bool x6 = fits && (d.w_format & EGR_W_BF16X3) &&
                    ((d.w_format & EGR_W_FORCE) || (a.M >= 1024 && 2.0 * (double)a.M * d.cout * a.K * G >= 4e9));
    // small 1x1 problems on the fp32 path: one launch of wgrad_small_kernel (weights and bias gradient, no slabs)
    static const int g_small = getenv("EGR_WGRAD_SMALL") ? atoi(getenv("EGR_WGRAD_SMALL")) : 1;
    // (a workgroup of the small kernel walks ALL rows of its group: ~14 us per 1000 rows - beyond EGR_WGRAD_SMALL_ROWS per group the split kernel
    // with its slab reduction is faster: the FPN's top-level 1x1 convs at batch 32, 4096 rows, took 120 us each)
    static const int g_small_rows = getenv("EGR_WGRAD_SMALL_ROWS") ? atoi(getenv("EGR_WGRAD_SMALL_ROWS")) : 2048;
    if (g_small && !x6 && d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad == 0 && d.h == d.ho && d.w == d.wo && (int64_t)a.M * G <= 8192 &&
        a.M <= g_small_rows && d.gw >= (int64_t)d.cout * a.K) {
        const int64_t tiles32 = (int64_t)((d.cout + 31) / 32) * (a.K / 32);
        if (tiles32 * G <= 256 &&      // (1024 tiles of M 480 x K 4096: 65 us against 38 on the tiled kernel, measured)
             !(db && (d.ymap.n_inner < d.n || d.ymap.stride_inner != (int64_t)d.ho * d.wo * d.ldy))) {
            hipLaunchKernelGGL(wgrad_small_kernel, dim3((unsigned)tiles32, 1, (unsigned)G), dim3(64 * WSM_NW), 0, (hipStream_t)stream, a);
            g_last_kernel = 4;
            g_last_h2 = 0;
            return egr_launch_status();
        }
    }
    const int bco = (d.cout > 64) ? 128 : 64;
    a.tilesCO = (d.cout + bco - 1) / bco;
    a.tilesK = (a.chunks + 3) / 4;
    int tiles = a.tilesCO * a.tilesK;
    // 3x3 / stride 1 / pad 1 split launches: the tap-sharing kernel (a stage = 16 pixels of one image row)
    static const int g_wg3 = getenv("EGR_WGRAD3") ? atoi(getenv("EGR_WGRAD3")) : 1;   // diagnostic: 0 = the generic kernel everywhere
    int wg3 = 0;          // 1: 64 channels x 2 chunks, 2: 128 channels x 1 chunk
    const bool narrow = d.wo == 8 && d.ho % 2 == 0;     // 8-pixel-wide images: a stage is two rows
    if (g_wg3 && d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad == 1 && d.ho == d.h && d.wo == d.w && (d.wo % 16 == 0 || narrow) && x6) {
        if (d.cout % 128 == 0) wg3 = 2;
        else if (d.cout % 64 == 0 && d.cin % 64 == 0) wg3 = 1;
    }
    if (wg3) tiles = (wg3 == 2) ? (d.cout / 128) * (d.cin / 32) : (d.cout / 64) * (d.cin / 64);
    const int stage_rows = wg3 ? 16 : RS;
    // Two workgroups fit a CU (64 KiB LDS each): 512 run at once, and equal-sized blocks finish in whole rounds.  Pick the
    // split count that minimises rounds x (rows per block + a fixed per-block cost of ~4 stages: prologue, slab write).
    const int max_splits = (a.M + 4 * stage_rows - 1) / (4 * stage_rows);  // at least 4 stages per split
    int splits = 1;
    {
        double best = 1e30;
        const int64_t tg = (int64_t)tiles * G;
        for (int sp = 1; sp <= max_splits && sp <= 512; ++sp) {
            const int64_t rounds = (tg * sp + 511) / 512;
            const double cost = (double)rounds * ((double)((a.M + sp - 1) / sp) + 4.0 * stage_rows);
            if (cost < best * 0.97) { best = cost; splits = sp; }   // prefer fewer splits unless clearly better
            if (tg * sp > 4096) break;
        }
    }
    const size_t per_split = (size_t)d.cout * a.K;
    if ((size_t)splits * per_split * G > workspace_floats) splits = (int)(workspace_floats / (per_split * G));
    if (splits < 1) return EGR_EWORKSPACE;
    a.rows_per_split = ((a.M + splits - 1) / splits + RS - 1) / RS * RS;
    a.splits = (a.M + a.rows_per_split - 1) / a.rows_per_split;
    hipStream_t s = (hipStream_t)stream;
    const bool direct = (a.splits == 1 && !accumulate && d.gw == (int64_t)d.cout * a.K);   // the slab IS the result
    if (direct) a.ws = dw;
    // generic split kernel: the bias gradient rides along (per-split column sums behind the dW slabs in the workspace)
    static const int g_fold_bias = getenv("EGR_WGRAD_FOLD_BIAS") ? atoi(getenv("EGR_WGRAD_FOLD_BIAS")) : 1;
    a.bias_ws = nullptr;
    if (g_fold_bias && db && x6 && !wg3 && d.ymap.n_inner >= d.n && d.ymap.stride_inner == (int64_t)d.ho * d.wo * d.ldy) {
        const size_t slabs = (size_t)a.splits * per_split * G, need = (size_t)a.splits * d.cout * G;
        if (slabs + need <= workspace_floats) a.bias_ws = workspace + slabs;
    }
    dim3 grid((unsigned)tiles, (unsigned)a.splits, (unsigned)G);
    g_last_kernel = x6 ? (wg3 ? 1 + wg3 : 1) : 0;
    const bool h2 = x6 && (d.w_format & EGR_W_F16X2);        // the split launch in the fp16 scheme (two planes, three products)
    g_last_h2 = h2 ? 1 : 0;
    if (h2 && wg3) {
        if (wg3 == 2 && narrow) hipLaunchKernelGGL((conv_wgrad3_x6_kernel<128, 1, 8, 2>), grid, dim3(256), 0, s, a);
        else if (wg3 == 2) hipLaunchKernelGGL((conv_wgrad3_x6_kernel<128, 1, 16, 2>), grid, dim3(256), 0, s, a);
        else if (narrow) hipLaunchKernelGGL((conv_wgrad3_x6_kernel<64, 2, 8, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_wgrad3_x6_kernel<64, 2, 16, 2>), grid, dim3(256), 0, s, a);
    } else if (h2) {
        if (bco == 128) hipLaunchKernelGGL((conv_wgrad_x6_kernel<128, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_wgrad_x6_kernel<64, 2>), grid, dim3(256), 0, s, a);
    } else if (x6 && wg3) {
        if (wg3 == 2 && narrow) hipLaunchKernelGGL((conv_wgrad3_x6_kernel<128, 1, 8>), grid, dim3(256), 0, s, a);
        else if (wg3 == 2) hipLaunchKernelGGL((conv_wgrad3_x6_kernel<128, 1, 16>), grid, dim3(256), 0, s, a);
        else if (narrow) hipLaunchKernelGGL((conv_wgrad3_x6_kernel<64, 2, 8>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_wgrad3_x6_kernel<64, 2, 16>), grid, dim3(256), 0, s, a);
    } else if (x6) {
        if (bco == 128) hipLaunchKernelGGL(conv_wgrad_x6_kernel<128>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv_wgrad_x6_kernel<64>, grid, dim3(256), 0, s, a);
    } else if (bco == 128) hipLaunchKernelGGL(conv_wgrad_kernel<128>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(conv_wgrad_kernel<64>, grid, dim3(256), 0, s, a);
    int rc = egr_launch_status();
    if (rc) return rc;
    const int64_t total = (int64_t)d.cout * a.K;
    if (!direct) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total / 4 + 15) / 16), 1, (unsigned)G), dim3(256), 0, s, a);
    rc = egr_launch_status();
    if (rc || !db) return rc;
    if (a.bias_ws) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((d.cout + 255) / 256), 1, (unsigned)G), dim3(256), 0, s, a);
        return egr_launch_status();
    }
    // bias gradient: column sums of dy (plain batch only)
    if (d.ymap.n_inner < d.n || d.ymap.stride_inner != (int64_t)d.ho * d.wo * d.ldy) return EGR_EINVAL;
    if (M64 <= 4096) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((unsigned)((d.cout + 15) / 16), (unsigned)G), dim3(256), 0, s, dy, (int)M64, d.cout, d.ldy, db,
                           accumulate, d.gy, d.gp);
        return egr_launch_status();
    }
    int nblk = (int)((M64 + 255) / 256);          // >= 256 rows per block, at most 256 blocks
    if (nblk > 256) nblk = 256;
    const int rpb = (int)((M64 + nblk - 1) / nblk);
    nblk = (int)((M64 + rpb - 1) / rpb);
    if ((size_t)nblk * d.cout * G > workspace_floats) return EGR_EWORKSPACE;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nblk, (unsigned)((d.cout + 1023) / 1024), (unsigned)G), dim3(256), 0, s, dy, M64,
                       d.cout, d.ldy, workspace, rpb, d.gy);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((d.cout + 15) / 16), (unsigned)G), dim3(256), 0, s, workspace, nblk,
                       d.cout, db, accumulate, d.gp);
    return egr_launch_status();
}
