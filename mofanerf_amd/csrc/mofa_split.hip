// OPT-IN experiment (MOFA_GEMM=bf16x3 | bf16x6 | fp16x3; default OFF — the shipped path is the exact fp32 MFMA of mofa_layer.h):
// the fp32 products of a layer emulated by partial products of 16-bit pieces on the 16-bit matrix pipe.  Its own translation unit:
// nothing here is reachable unless the host passes split weights / a piece count to mofa_net_forward or calls
// mofa_layer_forward_split directly.  DESIGN.md section 3.6 has the numbers and the accuracy contract.
#include <atomic>

#include "mofa_layer.h"

extern "C" {
int mofa_internal_prof_open(void* stream, int kind);                       // mofa_mlp.hip: 1 = session open (event recorded), 0 = closed, < 0 = error
void mofa_internal_prof_close(void* stream, int kind, double flops);
}

namespace mofa {
namespace {

// ======================================================================================================
// OPT-IN split-product layer kernel (MOFA_GEMM=bf16x3 | bf16x6; default OFF — the shipped path is exact fp32 MFMA).
// Every fp32 operand is split EXACTLY into bf16 pieces by truncation (a = a1 + a2 + a3, 8+8+8 significand bits) and the
// product a*b is replaced by the partial products with piece index i + j <= P-1 on the 16x faster bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation): P = 3 -> 6 products (drops terms < 2^-23 |ab|: fp32-equivalent,
// measured 3.6e-7 on RGB, tools/split_precision_study.py), P = 2 -> 3 products (~2^-15 |ab|; 1.2e-5 on RGB).
// Activations stay fp32 panels in HBM/LDS and are split in registers (5.5 VALU ops per element, hidden under the
// other wave's MFMAs); weights are pre-split into P bf16 planes (mofa_net_pack_split).  Both operands use the same
// (lane, element) -> k assignment, so the instruction's internal k ordering is irrelevant.  C/D layout = the fp32 kernel's.
// ======================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SplitArgs {
    LayerArgs base;              // x1/x2/bias/y/... as for k_layer (base.w unused)
    const unsigned short* ws;    // split weights: [panel][plane][n_padded][16] bf16, 16-B chunks swizzled by (row>>3)&1
};

__device__ __forceinline__ unsigned pack_hi16(unsigned x0, unsigned x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// 8 fp32 values -> P bf16x8 pieces (exact truncation split)
template <int P>
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, bf16x8 (&out)[P]) {
    float r[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int p = 0; p < P; ++p) {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_hi16(__float_as_uint(r[2 * i]), __float_as_uint(r[2 * i + 1]));
        out[p] = __builtin_bit_cast(bf16x8, w);
        if (p + 1 < P) {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = r[i] - __uint_as_float(__float_as_uint(r[i]) & 0xFFFF0000u);
        }
    }
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// 8 fp32 values -> 2 fp16x8 pieces (round-to-nearest split: x = h1 + h2 + O(2^-22 |x|); needs |x| < 65504)
__device__ __forceinline__ void split8_f16(const f32x4 lo, const f32x4 hi, f16x8 (&out)[2]) {
    const float r[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 h1 = (_Float16)r[i];
        out[0][i] = h1;
        out[1][i] = (_Float16)(r[i] - (float)h1);
    }
}

template <int P, bool F16>
struct SplitFrag {
    using type = bf16x8;
};
template <int P>
struct SplitFrag<P, true> {
    using type = f16x8;
};

template <int P, bool F16, typename Frag>
__device__ __forceinline__ void split_any(const f32x4 lo, const f32x4 hi, Frag (&out)[P]) {
    if constexpr (F16) split8_f16(lo, hi, out);
    else split8<P>(lo, hi, out);
}

template <bool F16, typename Frag>
__device__ __forceinline__ f32x16 mfma_split(const Frag& a, const Frag& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int BN, int P, bool F16 = false, bool HH = false>
__global__ __launch_bounds__(256, 2) void k_layer_split(const SplitArgs sa) {
    static_assert(!HH || (F16 && P == 2), "pre-split activation panels exist for the fp16x3 mode only");
    using Frag = typename SplitFrag<P, F16>::type;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LayerArgs& a = sa.base;
    constexpr int BM = kRowTile;
    constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N;
    constexpr int NI = 2, NJ = (BM / WAVES_M) / 32;
    constexpr int WPLANE = BN * 8;                       // floats (= BN rows x 32 B) of one weight plane tile
    constexpr int STAGE = BM * 16 + P * WPLANE;          // floats per pipeline stage
    constexpr int XR = BM / 64;

    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= a.total_tiles) return;
    const int mt = logical / a.n_tiles, nt = logical - mt * a.n_tiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int KT = a.k1p + a.k2p;
    const int lr = lane & 31, g = lane >> 5, sw = (lane >> 2) & 3;

    auto stage_issue = [&](int buf, int kt) {
        float* xs = smem + buf * STAGE;
        const float* src = (kt < a.k1p ? a.x1 + ((long long)kt * a.m_padded + m0) * 16
                                       : a.x2 + ((long long)(kt - a.k1p) * a.m_padded + m0) * 16);
#pragma unroll
        for (int r = 0; r < XR; ++r) glds16(src + (r * 256 + tid) * 4, xs + (r * 256 + wave * 64) * 4);
        // weight planes: BN rows x 32 B each = BN*8 floats; 256 threads x 16 B = 1024 floats per round
        const float* wsrc = (const float*)(sa.ws + (((long long)kt * P) * a.n_padded + n0) * 16);
        float* ws = xs + BM * 16;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < WPLANE / 1024; ++r)
                glds16(wsrc + (long long)p * a.n_padded * 8 + (r * 256 + tid) * 4, ws + p * WPLANE + (r * 256 + wave * 64) * 4);
    };

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    stage_issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) stage_issue(cur ^ 1, kt + 1);
        const float* xs = smem + cur * STAGE;
        const float* ws = xs + BM * 16;
        // weight fragments: lane (row, g) holds k = 8g .. 8g+7 of each plane (one 16-B read per plane)
        Frag wf[NI][P];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = wn * 64 + 32 * i + lr;
            const int chunk = g ^ ((row >> 3) & 1);
#pragma unroll
            for (int p = 0; p < P; ++p) wf[i][p] = *(const Frag*)(ws + p * WPLANE + row * 8 + chunk * 4);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wm * (32 * NJ) + 32 * j + lr;
            const f32x4 lo = *(const f32x4*)(xs + row * 16 + (((2 * g) ^ sw) << 2));
            const f32x4 hi = *(const f32x4*)(xs + row * 16 + (((2 * g + 1) ^ sw) << 2));
            Frag xf[P];
            if constexpr (HH) xf[0] = __builtin_bit_cast(Frag, lo), xf[1] = __builtin_bit_cast(Frag, hi);   // pieces as stored
            else split_any<P, F16>(lo, hi, xf);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                // smallest partial products first
#pragma unroll
                for (int t = P - 1; t >= 0; --t)
#pragma unroll
                    for (int pw = 0; pw <= t; ++pw)
                        acc[i][j] = mfma_split<F16>(wf[i][pw], xf[t - pw], acc[i][j]);
            }
        }
        __syncthreads();
    }

    // epilogue: k_layer's forward epilogue (bias + ReLU + panel store), see the F16 / HH notes inline
    f32x4 bv[NI][4];
    int boff = n0 + wn * 64 + 4 * g;
    asm volatile("" : "+v"(boff));
    if (!a.bias_row_div) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[i][q] = *(const f32x4*)(a.bias + boff + 32 * i + 8 * q);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long m = m0 + wm * (32 * NJ) + 32 * j + lr;
        if (a.bias_row_div) {
            long long brow = m / a.bias_row_div;
            if (brow >= a.bias_rows) brow = a.bias_rows - 1;
            const float* bias = a.bias + brow * a.n_padded + n0 + wn * 64 + 4 * g;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[i][q] = *(const f32x4*)(bias + 32 * i + 8 * q);
        }
        const int msw = (int)(m >> 2) & 3;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + 32 * i + 8 * q + 4 * g;
                f32x4 v;
                v.x = acc[i][j][4 * q + 0] + bv[i][q].x;
                v.y = acc[i][j][4 * q + 1] + bv[i][q].y;
                v.z = acc[i][j][4 * q + 2] + bv[i][q].z;
                v.w = acc[i][j][4 * q + 3] + bv[i][q].w;
                if (a.relu) {
                    // relu_np propagates NaN: an operand beyond the fp16 range splits into (+Inf, -Inf), the sum of its
                    // products is NaN, and that stays visible down to the image (tests/test_gpu_edge.py)
                    v.x = relu_np(v.x), v.y = relu_np(v.y), v.z = relu_np(v.z), v.w = relu_np(v.w);
                }
                if constexpr (HH) store_quad_hh(a.y, a.m_padded, n, m, msw, v);
                else *(f32x4*)(a.y + (long long)(n >> 4) * a.m_padded * 16 + m * 16 + ((((n >> 2) & 3) ^ msw) << 2)) = v;
            }
        }
    }
}

// v2 of the opt-in split-product kernel: BOTH operands stay fp32 panels (the ordinary packed weights) and are split in
// registers; 3-stage LDS ring (24 KiB / stage, 72 KiB / workgroup -> two workgroups per CU) with a prefetch distance of two
// panels: LDS-DMA loads stay in flight ACROSS the barrier (counted s_waitcnt vmcnt + raw s_barrier), because a bf16x6
// panel lasts only ~1.5k MFMA cycles per wave — shorter than an L2/HBM round trip.
template <int BN, int P>
__global__ __launch_bounds__(256, 2) void k_layer_split2(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = kRowTile;
    constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N;
    constexpr int NI = 2, NJ = (BM / WAVES_M) / 32;
    constexpr int STAGE = (BM + BN) * 16;
    constexpr int XR = BM / 64, WR = BN / 64;
    constexpr int LOADS = XR + WR;                     // LDS-DMA instructions per thread per stage

    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= a.total_tiles) return;
    const int mt = logical / a.n_tiles, nt = logical - mt * a.n_tiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int KT = a.k1p + a.k2p;
    const int lr = lane & 31, g = lane >> 5, sw = (lane >> 2) & 3;

    auto stage_issue = [&](int buf, int kt) {
        float* xs = smem + buf * STAGE;
        float* ws = xs + BM * 16;
        const float* src = (kt < a.k1p ? a.x1 + ((long long)kt * a.m_padded + m0) * 16
                                       : a.x2 + ((long long)(kt - a.k1p) * a.m_padded + m0) * 16);
#pragma unroll
        for (int r = 0; r < XR; ++r) glds16(src + (r * 256 + tid) * 4, xs + (r * 256 + wave * 64) * 4);
        const float* wsrc = a.w + ((long long)kt * a.n_padded + n0) * 16;
#pragma unroll
        for (int r = 0; r < WR; ++r) glds16(wsrc + (r * 256 + tid) * 4, ws + (r * 256 + wave * 64) * 4);
    };

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    stage_issue(0, 0);
    if (KT > 1) stage_issue(1, 1);
    int cur = 0, nxt2 = 2;                              // ring positions of panel kt and panel kt+2
    for (int kt = 0; kt < KT; ++kt) {
        // this wave's loads of panel kt have landed once at most the LOADS newer ones (panel kt+1) are still in flight
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // ... and every other wave's too; also: everyone is done with panel kt-1
        if (kt + 2 < KT) stage_issue(nxt2, kt + 2);     // refill the buffer panel kt-1 just vacated
        const float* xs = smem + cur * STAGE;
        const float* ws = xs + BM * 16;
        bf16x8 wf[NI][P];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = wn * 64 + 32 * i + lr;
            const f32x4 lo = *(const f32x4*)(ws + row * 16 + (((2 * g) ^ sw) << 2));
            const f32x4 hi = *(const f32x4*)(ws + row * 16 + (((2 * g + 1) ^ sw) << 2));
            split8<P>(lo, hi, wf[i]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wm * (32 * NJ) + 32 * j + lr;
            const f32x4 lo = *(const f32x4*)(xs + row * 16 + (((2 * g) ^ sw) << 2));
            const f32x4 hi = *(const f32x4*)(xs + row * 16 + (((2 * g + 1) ^ sw) << 2));
            bf16x8 xf[P];
            split8<P>(lo, hi, xf);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                // smallest partial products first
#pragma unroll
                for (int t = P - 1; t >= 0; --t)
#pragma unroll
                    for (int pw = 0; pw <= t; ++pw)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][pw], xf[t - pw], acc[i][j], 0, 0, 0);
            }
        }
        cur = (cur == 2) ? 0 : cur + 1;
        nxt2 = (nxt2 == 2) ? 0 : nxt2 + 1;
    }

    f32x4 bv[NI][4];
    int boff = n0 + wn * 64 + 4 * g;
    asm volatile("" : "+v"(boff));
    if (!a.bias_row_div) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[i][q] = *(const f32x4*)(a.bias + boff + 32 * i + 8 * q);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long m = m0 + wm * (32 * NJ) + 32 * j + lr;
        if (a.bias_row_div) {
            long long brow = m / a.bias_row_div;
            if (brow >= a.bias_rows) brow = a.bias_rows - 1;
            const float* bias = a.bias + brow * a.n_padded + n0 + wn * 64 + 4 * g;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[i][q] = *(const f32x4*)(bias + 32 * i + 8 * q);
        }
        const int msw = (int)(m >> 2) & 3;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + 32 * i + 8 * q + 4 * g;
                f32x4 v;
                v.x = acc[i][j][4 * q + 0] + bv[i][q].x;
                v.y = acc[i][j][4 * q + 1] + bv[i][q].y;
                v.z = acc[i][j][4 * q + 2] + bv[i][q].z;
                v.w = acc[i][j][4 * q + 3] + bv[i][q].w;
                if (a.relu) {
                    v.x = relu_np(v.x), v.y = relu_np(v.y), v.z = relu_np(v.z), v.w = relu_np(v.w);
                }
                *(f32x4*)(a.y + (long long)(n >> 4) * a.m_padded * 16 + m * 16 + ((((n >> 2) & 3) ^ msw) << 2)) = v;
            }
        }
    }
}

// weights -> P bf16 planes: dst[((panel0+panel)*P + plane)*rows_padded + row][16], chunk (k/8) swizzled by (row>>3)&1
__global__ __launch_bounds__(256) void k_pack_split(const float* __restrict__ w, int n_out, int ld, int col0, int ncols,
                                                    unsigned short* __restrict__ dst, int rows_padded, int panel0,
                                                    int k_padded, int P, int fp16) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;       // over rows_padded * k_padded
    if (idx >= (long long)rows_padded * k_padded) return;
    const int e = idx & 7, gph = (idx >> 3) & 1;
    const long long rowpanel = idx >> 4;
    const int row = (int)(rowpanel % rows_padded), panel = (int)(rowpanel / rows_padded);
    const int k = panel * 16 + 8 * (gph ^ ((row >> 3) & 1)) + e;
    float r = (row < n_out && k < ncols) ? w[(long long)row * ld + col0 + k] : 0.f;
    for (int p = 0; p < P; ++p) {
        const long long o = (((long long)(panel0 + panel) * P + p) * rows_padded + row) * 16 + gph * 8 + e;
        if (fp16) {
            const _Float16 h = (_Float16)r;
            dst[o] = __builtin_bit_cast(unsigned short, h);
            r = r - (float)h;
        } else {
            const unsigned bits = __float_as_uint(r) & 0xFFFF0000u;
            dst[o] = (unsigned short)(bits >> 16);
            r = r - __uint_as_float(bits);
        }
    }
}

template <int P, bool F16 = false, bool HH = false>
int launch_layer_split(LayerArgs a, const unsigned short* ws, hipStream_t st) {
    constexpr int BN = 128;
    a.n_tiles = a.n_padded / BN;
    const long long total = (a.m_padded / kRowTile) * a.n_tiles;
    MOFA_REQUIRE(total > 0 && total < (1ll << 30), "layer_split: tile count %lld out of range", total);
    a.total_tiles = (int)total;
    SplitArgs sa{a, ws};
    const int prof = mofa_internal_prof_open(st, 0);       // measurement session open? (bench.py --gemm ...)
    if (prof < 0) return MOFA_EHIP;
    if (!F16 && a.w && !(ws && config().split_v == 1)) {  // v2: fp32 weight panels split in registers, 3-stage ring (bf16 only)
        const size_t lds2 = 3 * (size_t)(kRowTile + BN) * 16 * sizeof(float);
        static std::atomic<bool> attr_set[kMaxDevices];     // the attribute is per device (per loaded code object)
        const int dev = current_device();
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute((const void*)k_layer_split2<BN, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) !=
                hipSuccess)
                return check_launch("hipFuncSetAttribute(k_layer_split2)");
            attr_set[dev].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL((k_layer_split2<BN, P>), dim3((unsigned)round_up(total, 8)), dim3(256), lds2, st, a);
    } else {
        const size_t lds = 2 * (size_t)(kRowTile * 16 + P * BN * 8) * sizeof(float);
        MOFA_REQUIRE(ws, "layer_split: this mode needs the pre-split weight planes");
        hipLaunchKernelGGL((k_layer_split<BN, P, F16, HH>), dim3((unsigned)round_up(total, 8)), dim3(256), lds, st, sa);
    }
    if (prof) mofa_internal_prof_close(st, 0, 2.0 * (double)a.m_padded * (double)a.n_padded * 16.0 * (double)(a.k1p + a.k2p));
    return check_launch("k_layer_split");
}


inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace
}  // namespace mofa

using namespace mofa;

extern "C" {

/* OPT-IN split-product variant of mofa_layer_forward (pieces = 2: bf16x3, 3: bf16x6); w_split from mofa_pack_split. */
int mofa_layer_forward_split(const float* x1, int32_t k1, const float* x2, int32_t k2, const uint16_t* w_split,
                             const float* w_packed, int32_t pieces, const float* bias, int32_t bias_row_div,
                             int64_t bias_rows, float* y, int64_t m_padded, int32_t n_padded, int32_t relu, void* stream) {
    MOFA_REQUIRE(x1 && (w_split || w_packed) && bias && y, "layer_forward_split: null pointer");
    MOFA_REQUIRE(k1 > 0 && k1 % 16 == 0 && k2 >= 0 && k2 % 16 == 0 && (k2 == 0 || x2), "layer_forward_split: bad K");
    MOFA_REQUIRE(n_padded % 128 == 0 && m_padded % kRowTile == 0 && (pieces == 2 || pieces == 3 || pieces == -2),
                 "layer_forward_split: needs n_padded %% 128 == 0 and pieces in {2, 3, -2} (got %d, %d)", n_padded, pieces);
    LayerArgs a{};
    a.x1 = x1, a.x2 = x2, a.bias = bias, a.y = y, a.w = w_packed;   // w_packed != NULL -> v2 (operands split in registers)
    a.k1p = k1 / 16, a.k2p = k2 / 16, a.n_padded = n_padded, a.m_padded = m_padded;
    a.bias_row_div = bias_row_div, a.bias_rows = bias_rows, a.relu = relu;
    if (pieces == -2) return launch_layer_split<2, true>(a, w_split, (hipStream_t)stream);   // fp16x3
    return pieces == 3 ? launch_layer_split<3>(a, w_split, (hipStream_t)stream) : launch_layer_split<2>(a, w_split, (hipStream_t)stream);
}

int mofa_pack_split(const float* w, int32_t n_out, int32_t ld, int32_t col0, int32_t ncols, uint16_t* dst,
                    int32_t rows_padded, int32_t panel0, int32_t k_padded, int32_t pieces, void* stream) {
    MOFA_REQUIRE(w && dst && (pieces == 2 || pieces == 3 || pieces == -2), "pack_split: bad arguments");
    MOFA_REQUIRE(rows_padded >= n_out && k_padded % 16 == 0 && k_padded >= ncols && col0 >= 0 && col0 + ncols <= ld,
                 "pack_split: bad shape");
    const long long total = (long long)rows_padded * k_padded;
    hipLaunchKernelGGL(k_pack_split, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, w, n_out, ld, col0, ncols,
                       dst, rows_padded, panel0, k_padded, pieces < 0 ? -pieces : pieces, pieces < 0 ? 1 : 0);
    return check_launch("k_pack_split");
}

int mofa_internal_layer_split_hh(const float* x1, int32_t k1, const float* x2, int32_t k2, const uint16_t* w_split,
                                 const float* bias, int32_t bias_row_div, int64_t bias_rows, float* y, int64_t m_padded,
                                 int32_t n_padded, int32_t relu, void* stream) {
    MOFA_REQUIRE(x1 && w_split && bias && y && n_padded % 128 == 0 && m_padded % kRowTile == 0 && k1 % 16 == 0 && k2 % 16 == 0,
                 "layer_split_hh: bad arguments");
    LayerArgs a{};
    a.x1 = x1, a.x2 = x2, a.bias = bias, a.y = y;
    a.k1p = k1 / 16, a.k2p = x2 ? k2 / 16 : 0, a.n_padded = n_padded, a.m_padded = m_padded;
    a.bias_row_div = bias_row_div, a.bias_rows = bias_rows, a.relu = relu, a.y_hh = 1;
    return launch_layer_split<2, true, true>(a, w_split, (hipStream_t)stream);
}

}  // extern "C"
