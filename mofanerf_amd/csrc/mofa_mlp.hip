// Fused positional-encode -> conditioned-MLP kernels for gfx950 (MI355X).
//
// Replaces run_network/batchify/NeRF.forward of the reference (models/render_class.py:69-109,
// models/model.py:121-137, :202-230): every Linear+bias+ReLU is one launch of k_layer, an
// LDS-tiled fp32 MFMA (v_mfma_f32_32x32x2_f32 — exact fp32, bitwise an fmaf chain) GEMM whose
// operands arrive as ready-made, bank-swizzled LDS images ("panels") by direct global->LDS DMA.
//
// Formulation.  For a tile of 256 points (rows m) and BN output features (rows n):
//     D[n][m] = sum_k Wp[n][k] * X[m][k]          (weights are the MFMA "A" operand, points "B")
// so each lane ends up with 4 CONSECUTIVE features of ONE point per accumulator quad — one 16-byte
// store per quad straight into the next layer's panel layout (bias + ReLU fused).
//
// Work decomposition: workgroup = 4 waves (256 threads), tile 256 (m) x BN (n); BN = 128 -> waves
// 2(n) x 2(m), wave tile 64 x 128 (8 accumulators of 32x32 = 128 AGPRs); BN = 64 -> waves 1 x 4,
// wave tile 64 x 64.  K is walked in 16-wide panels, double-buffered in LDS (24 KiB / stage at
// BN=128 => 48 KiB / workgroup, 2 workgroups per CU so one's epilogue hides under the other's MFMAs).
// blockIdx -> tile is XCD-aware: block b runs on XCD b%8, and each XCD walks a contiguous range of
// point tiles across all feature tiles, so an X tile is fetched into ONE L2 and the (<= 8 MiB) weight
// slab stays resident in the 256 MiB Infinity Cache.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "mofa_common.h"

namespace mofa {
namespace {

struct LayerArgs {
    const float* x1;      // panels [k1p][m_padded][16]
    const float* x2;      // optional second source (skip concat [x | h]), panels [k2p][m_padded][16]
    const float* w;       // packed weights, panels [(k1p+k2p)][n_padded][16]
    const float* bias;    // [bias_rows][n_padded]
    float* y;             // panels [n_padded/16][m_padded][16]
    const float* mask;    // backward epilogue: saved forward activation with y's geometry; y *= (mask > 0)
    int accumulate;       // backward epilogue: y = (y_old + acc) [* mask]
    // layer-0 (positional encoding prologue) inputs
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    long long z_row_stride;
    long long n_points;
    long long m_padded;
    long long bias_rows;
    int k1p, k2p;         // number of 16-wide K panels per source
    int n_padded;
    int bias_row_div;     // 0: one bias row; else bias row = m / bias_row_div (per-ray bias)
    int relu;
    int S;
    int n_tiles;          // n_padded / BN
    int total_tiles;
    int y_hh;             // opt-in fp16x3 mode only: write y as pre-split fp16 piece panels (store_quad_hh)
    // layer-0 camera mode (mofa_layer0_forward_cam): rays are built in the prologue from (K, c2w, pixel) instead of being read
    const float* cam_c2w;   // 12 floats [3,4] (device) or NULL = read rays_o / rays_d
    const int* cam_pix;     // flat pixel index per ray, or NULL = pixel cam_pix0 + ray
    long long cam_pix0;
    float fx, fy, cx, cy;
    int cam_w;
#ifdef MOFA_TIMELINE   // measurement build only (tools/timeline_layer.py): per-workgroup time stamps, 8 x u64 per tile
    unsigned long long* timeline;
#endif
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// "hh" activation panels (opt-in fp16x3 mode): same bytes and swizzle as an fp32 panel row (64 B = four
// 16-B chunks per point and 16 features), but the chunks hold PRE-SPLIT fp16 pieces: chunk 2G = h1 of features 8G..8G+7,
// chunk 2G+1 = h2 of the same features (x = h1 + h2 + O(2^-23 |x|), both round-to-nearest).  The consuming kernel's
// operand fragment is then exactly the two 16-B reads it already makes - no conversion work per use.
// `v` = features n..n+3 (n % 4 == 0) of point m; msw = (m >> 2) & 3.
__device__ __forceinline__ void store_quad_hh(float* __restrict__ y, long long m_padded, int n, long long m, int msw,
                                              const f32x4 v) {
    f16x4 h1, h2;
    h1.x = (_Float16)v.x, h1.y = (_Float16)v.y, h1.z = (_Float16)v.z, h1.w = (_Float16)v.w;
    h2.x = (_Float16)(v.x - (float)h1.x), h2.y = (_Float16)(v.y - (float)h1.y);
    h2.z = (_Float16)(v.z - (float)h1.z), h2.w = (_Float16)(v.w - (float)h1.w);
    const int G = (n >> 3) & 1, half = (n >> 2) & 1;
    float* row = y + (long long)(n >> 4) * m_padded * 16 + m * 16 + half * 2;
    *(f16x4*)(row + (((2 * G) ^ msw) << 2)) = h1;
    *(f16x4*)(row + (((2 * G + 1) ^ msw) << 2)) = h2;
}

__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    // 16 B per lane, LDS destination = wave-uniform base + lane*16 (LDS-DMA, no VGPR round trip)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// positional-encoding feature k of a 3-vector: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]
// (models/model.py:24-45; frequency-major blocks of 3).  k is wave-uniform => no divergence.
__device__ __forceinline__ float pe_feature(int k, float x0, float x1, float x2, int nfeat) {
    if (k >= nfeat) return 0.0f;
    if (k < 3) return k == 0 ? x0 : (k == 1 ? x1 : x2);
    const int j = k - 3;
    const int f = j / 6;
    const int r = j - 6 * f;
    const int d = r >= 3 ? r - 3 : r;
    const float x = d == 0 ? x0 : (d == 1 ? x1 : x2);
    const float arg = x * (float)(1 << f);  // exact (power of two), like x * freq in the reference
    return r < 3 ? sinf(arg) : cosf(arg);
}

#ifndef MOFA_SETPRIO
#define MOFA_SETPRIO 0
#endif
template <int NI, int NJ>
__device__ __forceinline__ void mma_panel(const float* __restrict__ Xt, const float* __restrict__ Wt, int xrow0,
                                          int wrow0, int lane, f32x16 (&acc)[NI][NJ]) {
    const int lr = lane & 31, g = lane >> 5, sw = (lane >> 2) & 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = ((2 * h + g) ^ sw) << 2;  // swizzled 16-B chunk holding k = 8h + 4g .. +3
        f32x4 a[NI], b[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = *(const f32x4*)(Wt + (wrow0 + 32 * i + lr) * 16 + p);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = *(const f32x4*)(Xt + (xrow0 + 32 * j + lr) * 16 + p);
#if MOFA_SETPRIO   // A/B arm (tools/ab_layer.py): raise the wave's issue priority while its MFMA block issues
        __builtin_amdgcn_s_setprio(MOFA_SETPRIO);
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
#if MOFA_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
}

// ---- software-pipelined K loop (k_layer<.., PIPE = true>) ----------------------------------------------------------------
// Same tile, same two LDS stages, same MFMA order as the plain loop (bit-identical results) - only the PLACEMENT of the
// loop's memory instructions differs.  For the plain loop hipcc emits, per 16-wide panel and wave: one block of ~30 SALU + 6
// LDS-DMA requests with no MFMA in flight, 6 fragment reads followed by a full lgkmcnt(0) wait, 32 MFMAs, 6 fragment reads
// + wait + barrier, 32 MFMAs; the LDS-DMA of panel kt+1 is requested half a panel before the barrier that waits for it.  An
// LDS-DMA request costs 60-185 issue cycles (MI355X micro-architecture guide), so a wave that is alone on its SIMD leaves
// the matrix pipe idle for ~10 % of every panel.  Here every half panel (32 MFMAs) carries the memory instructions of the
// NEXT one in its shadow (`sched_group_barrier` pins the interleaving):
//   half A(kt):  fragment reads of the second half of panel kt            between the MFMAs of its first half
//   wait + barrier: panel kt+1 has landed everywhere, everyone is done reading panel kt's stage
//   half B(kt):  fragment reads of the first half of panel kt+1, THEN the LDS-DMA requests of panel kt+2 (into panel kt's
//                stage), one per MOFA_PIPE_GAP MFMAs                       between the MFMAs of the second half
// so a request is waited for a full panel after it was made and nothing sits between two MFMA blocks.  Measured (M = 196608,
// K = N = 1024, interleaved A/B): 139.3 -> 145.6 TFLOP/s; reads-before-requests and a gap of 4 matter (requests first: 141).
// Needs an even number of panels >= 4 (unrolled by two: stage addresses are compile-time constants); launch_layer checks.
#ifndef MOFA_PIPE_GAP
#define MOFA_PIPE_GAP 0      // MFMAs between two LDS-DMA requests; 0 = as many as the half panel allows after its reads
#endif                       // (4 for the 128-feature tile: 32 MFMAs, 6 reads, 6 requests; 2 for the 64-feature tile: 16 / 4 / 5)
// xb / x2b / wb: the tile's first panel in the two activation sources (x2b is only dereferenced when KT > k1p) and in the
// weight pack; xstep / wstep: floats between consecutive panels; xrow0 / wrow0: this wave's first row in the staged X / W
// tile; `wave`: index of the wave's 1 KiB slot inside each 4 KiB staging round.
template <int NI, int NJ, int BM, int BN>
__device__ __forceinline__ void kloop_pipelined(const float* xb, const float* x2b, const float* wb, long long xstep, long long wstep,
                                                int k1p, int KT, float* smem, int tid, int wave, int lane, int xrow0, int wrow0,
                                                f32x16 (&acc)[NI][NJ], unsigned long long* pstamp = nullptr) {
    constexpr int STAGE = (BM + BN) * 16, XR = BM / 64, WR = BN / 64;
    const int lr = lane & 31, g = lane >> 5, sw = (lane >> 2) & 3;
    int pq = 0;                                  // panel xb / wb point at
    const unsigned toff = (unsigned)tid * 4u;
    float* const lds_wave = smem + wave * 256;   // this wave's 1 KiB slot inside each 4 KiB round

    struct Frag {
        f32x4 a[NI], b[NJ];
    };
    auto request = [&](int stage) {              // LDS-DMA of panel pq into `stage`, then step to panel pq + 1
        float* xs = lds_wave + stage * STAGE;
        float* ws = xs + BM * 16;
#pragma unroll
        for (int r = 0; r < XR; ++r) glds16(xb + (r * 1024u + toff), xs + r * 1024);
#pragma unroll
        for (int r = 0; r < WR; ++r) glds16(wb + (r * 1024u + toff), ws + r * 1024);
        ++pq;
        wb += wstep;
        xb = pq == k1p ? x2b : xb + xstep;
    };
    auto read = [&](int stage, int h, Frag& f) {
        const float* Xt = smem + stage * STAGE;
        const float* Wt = Xt + BM * 16;
        const int p = ((2 * h + g) ^ sw) << 2;
#pragma unroll
        for (int i = 0; i < NI; ++i) f.a[i] = *(const f32x4*)(Wt + (wrow0 + 32 * i + lr) * 16 + p);
#pragma unroll
        for (int j = 0; j < NJ; ++j) f.b[j] = *(const f32x4*)(Xt + (xrow0 + 32 * j + lr) * 16 + p);
    };
    auto mfma_half = [&](const Frag& f) {       // the same (e, i, j) order as mma_panel
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };
    // scheduling masks: 0x008 MFMA, 0x100 LDS read, 0x020 VMEM read (the LDS-DMA request)
    auto half_a = [&](int stage, Frag& cur, Frag& nxt) {       // MFMAs of the first half, reads of the second
        __builtin_amdgcn_sched_barrier(0);
        read(stage, 1, nxt);
        mfma_half(cur);
#pragma unroll
        for (int q = 0; q < NI + NJ; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI * NJ, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync_point = [&]() {   // my own requests have landed and my reads are done; then everybody's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef MOFA_TIMELINE   // measurement build: one 100 MHz stamp per panel (tools/timeline_layer.py --panels), kept in LDS until the end
        if (pstamp && tid == 0) *pstamp++ = wall_clock64();   // (a global store here would sit in front of the next vmcnt(0) wait)
#endif
    };
    auto half_b = [&](int stage, bool do_request, bool do_read, Frag& cur, Frag& nxt) {   // MFMAs of the second half
        __builtin_amdgcn_sched_barrier(0);
        if (do_read) read(stage ^ 1, 0, nxt);
        if (do_request) request(stage);
        mfma_half(cur);
        if (do_read) {
#pragma unroll
            for (int q = 0; q < NI + NJ; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        if (do_request) {
            constexpr int GAP = MOFA_PIPE_GAP > 0 ? MOFA_PIPE_GAP : (4 * NI * NJ - (NI + NJ)) / (XR + WR);
            static_assert(GAP >= 1, "the half panel has too few MFMAs to carry its memory instructions");
#pragma unroll
            for (int q = 0; q < XR + WR; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI * NJ, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    Frag fa, fb;
    request(0);
    request(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XR + WR) : "memory");   // panel 0 (the older requests) has landed
    __builtin_amdgcn_s_barrier();
#ifdef MOFA_TIMELINE
    if (pstamp && tid == 0) pstamp[63] = wall_clock64();              // slot 63 is never a panel stamp (KT <= 64): panel 0 landed
#endif
    read(0, 0, fa);
    for (int kt = 0; kt + 2 < KT; kt += 2) {
        half_a(0, fa, fb);
        sync_point();
        half_b(0, true, true, fb, fa);
        half_a(1, fa, fb);
        sync_point();
        half_b(1, true, true, fb, fa);
    }
    half_a(0, fa, fb);
    sync_point();
    half_b(0, false, true, fb, fa);
    half_a(1, fa, fb);
    half_b(1, false, false, fb, fa);
}

// Forward epilogue of one wave tile (NI x NJ accumulators of 32x32): bias + ReLU, one 16-byte store per accumulator quad
// straight into the next layer's panels.  PERRAY (the view layer: bias row = ray of the point) is a TEMPLATE parameter on
// purpose: with the per-ray bias loads inside the point loop under a RUN-TIME `if`, hipcc must assume at the join that the
// loads may still be in flight and brackets every store with `s_waitcnt vmcnt(7)` — and because loads and stores share the
// in-order vmcnt on gfx9, that also limits every wave of the ordinary layers to 7 stores in flight: the 32 stores per lane
// of a tile then take 4-5 store-acknowledge round trips (the "12 us to issue the epilogue stores" of DESIGN.md 3.1) instead
// of being fire-and-forget.  With PERRAY = false the bias is fetched once, waited for once, and the 32 stores issue back to
// back with no wait between them.
template <int NI>
__device__ __forceinline__ void bias_fetch(const float* __restrict__ bias_base, int n_first, int lane, f32x4 (&bv)[NI][4]) {
    int boff = n_first + 4 * (lane >> 5);
    asm volatile("" : "+v"(boff));  // opaque AFTER the K loop: keeps hipcc from hoisting the 8 bias loads (32 VGPRs) above it
#pragma unroll
    for (int i = 0; i < NI; ++i)    // one bias row for every point: fetch it once, all 8 loads in flight together
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[i][q] = *(const f32x4*)(bias_base + boff + 32 * i + 8 * q);
}

// FETCH = false: `bv` was filled by bias_fetch earlier (the persistent kernel fetches it BEFORE it requests the next tile's
// first panel, so that the in-order vmcnt wait for the bias does not also wait for that panel).
template <int NI, int NJ, bool PERRAY, bool HH, bool FETCH = true>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[NI][NJ], const float* __restrict__ bias_base, long long bias_rows,
                                           int bias_row_div, int n_padded, float* __restrict__ y, long long m_padded,
                                           long long m_first, int n_first, int relu, int lane, f32x4 (&bv)[NI][4]) {
    const int lr = lane & 31, g = lane >> 5;
    int boff = n_first + 4 * g;
    if constexpr (!PERRAY && FETCH) bias_fetch<NI>(bias_base, n_first, lane, bv);
    if constexpr (PERRAY) asm volatile("" : "+v"(boff));
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long m = m_first + 32 * j + lr;
        if constexpr (PERRAY) {     // per-ray bias (view layer): row = ray of this point
            long long brow = m / bias_row_div;
            if (brow >= bias_rows) brow = bias_rows - 1;
            const float* bias = bias_base + brow * n_padded + boff;
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
                const int n = n_first + 32 * i + 8 * q + 4 * g;
                f32x4 v;
                v.x = acc[i][j][4 * q + 0] + bv[i][q].x;
                v.y = acc[i][j][4 * q + 1] + bv[i][q].y;
                v.z = acc[i][j][4 * q + 2] + bv[i][q].z;
                v.w = acc[i][j][4 * q + 3] + bv[i][q].w;
                if (relu) {
                    v.x = relu_np(v.x), v.y = relu_np(v.y), v.z = relu_np(v.z), v.w = relu_np(v.w);
                }
                if constexpr (HH) store_quad_hh(y, m_padded, n, m, msw, v);
                else *(f32x4*)(y + (long long)(n >> 4) * m_padded * 16 + m * 16 + ((((n >> 2) & 3) ^ msw) << 2)) = v;
            }
        }
    }
}

#ifndef MOFA_STAGED_EPILOGUE
#define MOFA_STAGED_EPILOGUE 1   // 0: A/B arm with store_tile's 16-byte strided stores everywhere
#endif
// The forward epilogue of the ordinary layers (one bias row, fp32 panels) with CONTIGUOUS stores.  In store_tile a wave-store is 64 lanes x 16 B at a
// 64-byte stride (a lane owns a point), which the memory pipeline issues at ~7 B/clk/CU (store-issue-bound); here every wave
// passes its tile through a PRIVATE 4 KiB LDS window in the panels' own (swizzled) row layout — 64 rows x 64 B per slice, written
// as 16-byte fragments, read back as 1 KiB contiguous wave rows — so that each global store is 1 KiB of consecutive bytes.
// No barrier: the window is wave-private and a wave's LDS operations execute in order.  `win` must not be read or written by
// anyone else (the pipelined K loop's stage 0 is free for all waves after its last barrier).  Same values as store_tile
// (bit-identical).  Measured against it (interleaved A/B): +0.4 % at K = N = 1024, +4 % at 256, k_mlp_fused 133.2 -> 135.5 TFLOP/s.
template <int NI, int NJ, bool RELU>
__device__ __forceinline__ void store_tile_staged(const f32x16 (&acc)[NI][NJ], const float* __restrict__ bias, float* __restrict__ y,
                                                  long long m_padded, long long m_first, int n_first, int lane, float* win) {
    static_assert(NJ % 2 == 0, "row halves of 64 points");
    const int lr = lane & 31, g = lane >> 5, msw = (lr >> 2) & 3;   // m_first + 32 j is a multiple of 32: the row swizzle is the lane's
    f32x4 bv[NI][4];
    bias_fetch<NI>(bias, n_first, lane, bv);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
            float* __restrict__ panel = y + ((long long)((n_first >> 4) + 2 * i + qh) * m_padded + m_first) * 16;
#pragma unroll
            for (int jh = 0; jh < NJ / 2; ++jh) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int j = 2 * jh + jj, q = 2 * qh + qq;
                        f32x4 v;
                        v.x = acc[i][j][4 * q + 0] + bv[i][q].x, v.y = acc[i][j][4 * q + 1] + bv[i][q].y;
                        v.z = acc[i][j][4 * q + 2] + bv[i][q].z, v.w = acc[i][j][4 * q + 3] + bv[i][q].w;
                        if constexpr (RELU) v.x = relu_np(v.x), v.y = relu_np(v.y), v.z = relu_np(v.z), v.w = relu_np(v.w);
                        *(f32x4*)(win + (32 * jj + lr) * 16 + (((2 * qq + g) ^ msw) << 2)) = v;     // logical chunk 2 qq + g
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f32x4 v = *(const f32x4*)(win + it * 256 + lane * 4);
                    *(f32x4*)(panel + jh * 1024 + it * 256 + lane * 4) = v;
                }
            }
        }
}

// Backward-data epilogue through the same wave-private LDS window: dX = (acc [+ dX_old]) [* (saved activation > 0)].  Staging
// first turns the accumulator fragments into 1 KiB contiguous wave rows, so the optional reads of dX_old and of the saved
// activation are fully coalesced 1 KiB loads (all four of a slice in flight together) instead of 16 B per lane at a 64-byte
// stride, and ACC / MASK are compile-time: no wait sits between a load and the next one.  Same arithmetic as the strided form.
template <int NI, int NJ, bool ACC, bool MASK>
__device__ __forceinline__ void store_tile_staged_bwd(const f32x16 (&acc)[NI][NJ], float* __restrict__ y, const float* __restrict__ mask,
                                                      long long m_padded, long long m_first, int n_first, int lane, float* win) {
    static_assert(NJ % 2 == 0, "row halves of 64 points");
    const int lr = lane & 31, g = lane >> 5, msw = (lr >> 2) & 3;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
            const long long poff = ((long long)((n_first >> 4) + 2 * i + qh) * m_padded + m_first) * 16;
#pragma unroll
            for (int jh = 0; jh < NJ / 2; ++jh) {
                const long long off = poff + jh * 1024 + lane * 4;
                f32x4 old[4], act[4];
                if constexpr (ACC) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) old[it] = *(const f32x4*)(y + off + it * 256);
                }
                if constexpr (MASK) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) act[it] = *(const f32x4*)(mask + off + it * 256);
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int j = 2 * jh + jj, q = 2 * qh + qq;
                        f32x4 v;
                        v.x = acc[i][j][4 * q + 0], v.y = acc[i][j][4 * q + 1], v.z = acc[i][j][4 * q + 2], v.w = acc[i][j][4 * q + 3];
                        *(f32x4*)(win + (32 * jj + lr) * 16 + (((2 * qq + g) ^ msw) << 2)) = v;
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    f32x4 v = *(const f32x4*)(win + it * 256 + lane * 4);
                    if constexpr (ACC) v.x += old[it].x, v.y += old[it].y, v.z += old[it].z, v.w += old[it].w;
                    if constexpr (MASK)
                        v.x = act[it].x > 0.f ? v.x : 0.f, v.y = act[it].y > 0.f ? v.y : 0.f, v.z = act[it].z > 0.f ? v.z : 0.f,
                        v.w = act[it].w > 0.f ? v.w : 0.f;
                    *(f32x4*)(y + off + it * 256) = v;
                }
            }
        }
}

// BN: feature-tile height; L0: X tile is generated (positional encoding) instead of loaded;
// GLDS: stage operands with LDS-DMA (true) or through registers (false; kept as the A/B arm).
// BWD: backward-data epilogue (no bias/ReLU; optional accumulate into y and ReLU mask from the saved activation):
//      dX[m][k] = sum_n G[m][n] * W[n][k]  is the same GEMM with the transposed weight pack as "Wp".
#ifndef MOFA_LAYER_WAVES
#define MOFA_LAYER_WAVES 2  // min waves per SIMD the register allocator must leave room for (= workgroups per CU)
#endif
template <int BN, bool L0, bool GLDS, bool BWD = false, bool HH = false, bool PERRAY = false, bool PIPE = false>
__global__ __launch_bounds__(256, MOFA_LAYER_WAVES) void k_layer(const LayerArgs a) {
    static_assert(!PIPE || (GLDS && !L0 && BN == 128), "the pipelined K loop stages both operands by LDS-DMA at the 128-feature tile");
    // (measured on the 64-feature tile too - 4 workgroups per CU, 16 MFMAs per half panel carrying 4 reads + 5 requests: 132 against
    //  136 TFLOP/s for its plain loop, so that tile keeps the plain loop)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = kRowTile;
    constexpr int WAVES_N = BN / 64;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int NI = 2;
    constexpr int NJ = (BM / WAVES_M) / 32;
    constexpr int STAGE = (BM + BN) * 16;  // floats per pipeline stage
    constexpr int XR = BM / 64;            // 4 KiB rounds per X stage
    constexpr int WR = BN / 64;

    // XCD-aware tile order (block b -> XCD b % 8; grid is padded to a multiple of 8)
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= a.total_tiles) return;
    const int mt = logical / a.n_tiles, nt = logical - mt * a.n_tiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int KT = a.k1p + a.k2p;

    // layer 0: this thread owns point row m0+tid
    float px = 0.f, py = 0.f, pz = 0.f;
    if constexpr (L0) {
        long long m = m0 + tid;
        if (m >= a.n_points) m = a.n_points - 1;
        if (a.pts) {
            px = a.pts[m * 3 + 0], py = a.pts[m * 3 + 1], pz = a.pts[m * 3 + 2];
        } else {
            const long long r = m / a.S;
            const int s = (int)(m - r * a.S);
            const float zz = a.z[r * a.z_row_stride + s];
            float ro[3], rd[3];
            if (a.cam_c2w) {   // the ray itself comes from (K, c2w, pixel): get_rays folded into the prologue
                const long long pix = a.cam_pix ? (long long)a.cam_pix[r] : a.cam_pix0 + r;
                const int pj = (int)(pix / a.cam_w), pi = (int)(pix - (long long)pj * a.cam_w);
                pinhole_ray(pi, pj, a.fx, a.fy, a.cx, a.cy, a.cam_c2w, ro, rd);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) ro[c] = a.rays_o[r * 3 + c], rd[c] = a.rays_d[r * 3 + c];
            }
            // pts = o + d * z with a separately rounded multiply and add (render_class.py:315)
            px = __fadd_rn(ro[0], __fmul_rn(rd[0], zz));
            py = __fadd_rn(ro[1], __fmul_rn(rd[1], zz));
            pz = __fadd_rn(ro[2], __fmul_rn(rd[2], zz));
        }
    }

    auto x_src = [&](int kt) -> const float* {
        const float* base = kt < a.k1p ? a.x1 : a.x2;
        const int kk = kt < a.k1p ? kt : kt - a.k1p;
        return base + ((long long)kk * a.m_padded + m0) * 16;
    };
    auto w_src = [&](int kt) -> const float* { return a.w + ((long long)kt * a.n_padded + n0) * 16; };

    f32x4 sx[GLDS ? 1 : XR], sw_[GLDS ? 1 : WR];  // register staging (GLDS == false only)

    auto stage_issue = [&](int buf, int kt) {
        float* xs = smem + buf * STAGE;
        float* ws = xs + BM * 16;
        if constexpr (L0) {
            const int swz = (tid >> 2) & 3;
#pragma unroll 1
            for (int kk = 0; kk < 16; ++kk) {
                const float v = pe_feature(kt * 16 + kk, px, py, pz, 3 + 6 * MOFA_PE_POINT_FREQS);
                xs[tid * 16 + ((((kk >> 2) & 3) ^ swz) << 2) + (kk & 3)] = v;
            }
        } else if constexpr (GLDS) {
            const float* src = x_src(kt);
#pragma unroll
            for (int r = 0; r < XR; ++r) glds16(src + (r * 256 + tid) * 4, xs + (r * 256 + wave * 64) * 4);
        } else {
            const float* src = x_src(kt);
#pragma unroll
            for (int r = 0; r < XR; ++r) sx[r] = *(const f32x4*)(src + (r * 256 + tid) * 4);
        }
        const float* wsrc = w_src(kt);
        if constexpr (GLDS) {
#pragma unroll
            for (int r = 0; r < WR; ++r) glds16(wsrc + (r * 256 + tid) * 4, ws + (r * 256 + wave * 64) * 4);
        } else {
#pragma unroll
            for (int r = 0; r < WR; ++r) sw_[r] = *(const f32x4*)(wsrc + (r * 256 + tid) * 4);
        }
    };
    auto stage_commit = [&](int buf) {  // register-staged arm: write the tile once the MFMAs are issued
        if constexpr (!GLDS) {
            float* xs = smem + buf * STAGE;
            float* ws = xs + BM * 16;
            if constexpr (!L0) {
#pragma unroll
                for (int r = 0; r < XR; ++r) *(f32x4*)(xs + (r * 256 + tid) * 4) = sx[r];
            }
#pragma unroll
            for (int r = 0; r < WR; ++r) *(f32x4*)(ws + (r * 256 + tid) * 4) = sw_[r];
        }
    };

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

#ifdef MOFA_TIMELINE
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tc1 = 0, tc2 = 0;
    if (a.timeline && tid == 0) ts0 = wall_clock64();
#endif
    if constexpr (PIPE) {
#ifdef MOFA_TIMELINE
        if (a.timeline && tid == 0) ts1 = wall_clock64(), tc1 = clock64();   // (includes the first two panels' fetch)
#endif
        kloop_pipelined<NI, NJ, BM, BN>(a.x1 + m0 * 16, a.k2p ? a.x2 + m0 * 16 : nullptr, a.w + (long long)n0 * 16, a.m_padded * 16,
                                        (long long)a.n_padded * 16, a.k1p, KT, smem, tid, wave, lane, wm * (32 * NJ), wn * 64, acc
#ifdef MOFA_TIMELINE
                                        , (a.timeline && KT <= 64) ? (unsigned long long*)(smem + 2 * STAGE) : nullptr   // 512 B behind the stages
#endif
        );
#ifdef MOFA_TIMELINE
        __builtin_amdgcn_s_barrier();
#endif
    } else {
        stage_issue(0, 0);
        stage_commit(0);
        __syncthreads();
#ifdef MOFA_TIMELINE
        if (a.timeline && tid == 0) ts1 = wall_clock64(), tc1 = clock64();   // first operand panel has landed: the K loop starts
#endif
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) stage_issue(cur ^ 1, kt + 1);
            const float* xs = smem + cur * STAGE;
            mma_panel<NI, NJ>(xs, xs + BM * 16, wm * (32 * NJ), wn * 64, lane, acc);
            if (kt + 1 < KT) stage_commit(cur ^ 1);
            __syncthreads();
        }
    }
#ifdef MOFA_TIMELINE
    if (a.timeline && tid == 0) ts2 = wall_clock64(), tc2 = clock64();   // K loop done (all four waves): the epilogue starts
#endif

    const int lr = lane & 31, g = lane >> 5;
    if constexpr (BWD && PIPE && MOFA_STAGED_EPILOGUE) {
        float* win = smem + wave * 1024;
        const long long mf = m0 + wm * (32 * NJ);
        const int nf = n0 + wn * 64;
        if (a.accumulate) {
            if (a.mask) store_tile_staged_bwd<NI, NJ, true, true>(acc, a.y, a.mask, a.m_padded, mf, nf, lane, win);
            else store_tile_staged_bwd<NI, NJ, true, false>(acc, a.y, a.mask, a.m_padded, mf, nf, lane, win);
        } else {
            if (a.mask) store_tile_staged_bwd<NI, NJ, false, true>(acc, a.y, a.mask, a.m_padded, mf, nf, lane, win);
            else store_tile_staged_bwd<NI, NJ, false, false>(acc, a.y, a.mask, a.m_padded, mf, nf, lane, win);
        }
        return;
    }
    if constexpr (BWD) {
        // backward-data epilogue: (acc [+ y_old]) [* (saved activation > 0)]
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const long long m = m0 + wm * (32 * NJ) + 32 * j + lr;
            const int msw = (int)(m >> 2) & 3;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + 32 * i + 8 * q + 4 * g;
                    const long long off = (long long)(n >> 4) * a.m_padded * 16 + m * 16 + ((((n >> 2) & 3) ^ msw) << 2);
                    f32x4 v;
                    v.x = acc[i][j][4 * q + 0], v.y = acc[i][j][4 * q + 1], v.z = acc[i][j][4 * q + 2],
                    v.w = acc[i][j][4 * q + 3];
                    if (a.accumulate) {
                        const f32x4 o = *(const f32x4*)(a.y + off);
                        v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                    }
                    if (a.mask) {
                        const f32x4 k = *(const f32x4*)(a.mask + off);
                        v.x = k.x > 0.f ? v.x : 0.f, v.y = k.y > 0.f ? v.y : 0.f, v.z = k.z > 0.f ? v.z : 0.f,
                        v.w = k.w > 0.f ? v.w : 0.f;
                    }
                    *(f32x4*)(a.y + off) = v;
                }
            }
        }
        return;
    }
    // epilogue: bias + ReLU, one 16-B store per accumulator quad into the next layer's panels
#ifdef MOFA_ABLATE_EPILOGUE   // timing-only ablation build (wrong results): what a free epilogue would be worth
    float s_ = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) s_ += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
    if (s_ == 123.456f) a.y[0] = s_;
    return;
#endif
    if constexpr (MOFA_STAGED_EPILOGUE && PIPE && !PERRAY && !HH) {
        float* win = smem + wave * 1024;      // 4 KiB per wave inside stage 0 (free for everybody after the K loop's last barrier)
        if (a.relu) store_tile_staged<NI, NJ, true>(acc, a.bias, a.y, a.m_padded, m0 + wm * (32 * NJ), n0 + wn * 64, lane, win);
        else store_tile_staged<NI, NJ, false>(acc, a.bias, a.y, a.m_padded, m0 + wm * (32 * NJ), n0 + wn * 64, lane, win);
    } else {
        f32x4 bv[NI][4];
        store_tile<NI, NJ, PERRAY, HH>(acc, a.bias, a.bias_rows, a.bias_row_div, a.n_padded, a.y, a.m_padded, m0 + wm * (32 * NJ),
                                       n0 + wn * 64, a.relu, lane, bv);
    }
#ifdef MOFA_TIMELINE
    if (a.timeline && tid == 0) {      // wave 0: its 32 stores per lane are ISSUED (not acknowledged)
        unsigned long long* t = a.timeline + (long long)logical * 8;
        t[0] = ts0, t[1] = ts1, t[2] = ts2, t[3] = wall_clock64();
        t[4] = __builtin_amdgcn_s_getreg(GETREG_IMMED(32 - 1, 0, HW_ID));
        t[5] = __builtin_amdgcn_s_getreg(GETREG_IMMED(4 - 1, 0, 20));        // XCC_ID
        t[6] = tc2 - tc1;                                                     // clock64() (s_memtime) ticks spent in the K loop
        if constexpr (PIPE) {
            if (KT <= 64) {
                const unsigned long long* ps = (const unsigned long long*)(smem + 2 * STAGE);
                unsigned long long* pd = a.timeline + (long long)a.total_tiles * 8 + (long long)logical * 64;
                for (int i = 0; i < 64; ++i) pd[i] = ps[i];
            }
        }
    }
#endif
}

// ---- 3-stage-ring twin of k_layer<128,false,true> (MOFA_RING3=1; A/B arm) --------------------------------------------------
// The timeline says a workgroup that is ALONE in its K loop drives the pipe at 75 %, a pair at 94 %.  In k_layer
// the next panel's LDS-DMA is requested half a panel (2,048 MFMA cycles of ONE wave) before the barrier that waits for it; alone
// on its SIMD a wave then sits out the rest of an L2 round trip every panel.  Here the ring has three stages (72 KiB per
// workgroup, still two per CU): panel kt+2 is requested at the top of panel kt and waited for two panels later with a COUNTED
// vmcnt (panel kt+1's loads may stay in flight across the barrier), one barrier per panel.  Same tiles, same arithmetic order,
// bit-identical results.
template <bool PERRAY>
__global__ __launch_bounds__(256, MOFA_LAYER_WAVES) void k_layer_ring3(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 128, BM = kRowTile, NI = 2, NJ = 4;
    constexpr int STAGE = (BM + BN) * 16, XR = BM / 64, WR = BN / 64, LOADS = XR + WR;
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= a.total_tiles) return;
    const int mt = logical / a.n_tiles, nt = logical - mt * a.n_tiles;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int KT = a.k1p + a.k2p;

    auto stage_issue = [&](int buf, int kt) {
        float* xs = smem + buf * STAGE;
        float* ws = xs + BM * 16;
        const float* base = kt < a.k1p ? a.x1 : a.x2;
        const int kk = kt < a.k1p ? kt : kt - a.k1p;
        const float* src = base + ((long long)kk * a.m_padded + m0) * 16;
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
    int cur = 0, nxt2 = 2;                                  // ring positions of panel kt and of panel kt + 2
    for (int kt = 0; kt < KT; ++kt) {
        // this wave's loads of panel kt have landed once at most the LOADS newer ones (panel kt + 1) are still in flight
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // ... and every other wave's; also: everyone is done reading panel kt - 1
        if (kt + 2 < KT) stage_issue(nxt2, kt + 2);         // refill the stage panel kt - 1 just vacated
        const float* xs = smem + cur * STAGE;
        mma_panel<NI, NJ>(xs, xs + BM * 16, wm * (32 * NJ), wn * 64, lane, acc);
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }
    f32x4 bv[NI][4];
    store_tile<NI, NJ, PERRAY, false>(acc, a.bias, a.bias_rows, a.bias_row_div, a.n_padded, a.y, a.m_padded, m0 + wm * (32 * NJ),
                                      n0 + wn * 64, a.relu, lane, bv);
}


// ---- persistent twin of k_layer<128,false,true> (MOFA_PERSIST=1; A/B arm) ------------------------------
// Same tile, same panels, same K loop, same epilogue, bit-identical results.  What changes is the SCHEDULE: the grid is
// 2 workgroups per CU and every workgroup WALKS its share of the tiles instead of exiting after one, so that
//   (a) the 5-7 us a freed slot waits for the dispatcher's next workgroup disappears (12 rounds per 196,608-point launch),
//   (b) the next tile's first operand panel is requested BEFORE the epilogue's 32 stores per lane are issued, so its
//       ~2.6 us first-fetch latency overlaps the store burst instead of following it,
//   (c) optionally (MOFA_DEPHASE=1) the 8 feature-tile workgroups of one point tile start late together by a
//       group-specific fraction of a tile time, so that the chip's workgroups are no longer all in their epilogue at once.
// XCD-aware walk: block b runs on XCD b % 8; XCD x owns the contiguous logical tile range [x*per, (x+1)*per) and its
// G/8 workgroups sweep it side by side, so the 8 feature tiles of a point tile are in flight together in ONE L2.
__global__ __launch_bounds__(256, MOFA_LAYER_WAVES) void k_layer_persist(const LayerArgs a, int per_xcd_tiles, int dephase) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 128, BM = kRowTile, NI = 2, NJ = 4;
    constexpr int STAGE = (BM + BN) * 16, XR = BM / 64, WR = BN / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int KT = a.k1p + a.k2p;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;

    auto tile_of = [&](int it, long long& m0, int& n0) -> bool {
        const int local = w + it * wg_per_xcd;
        const int logical = xcd * per_xcd_tiles + local;
        if (local >= per_xcd_tiles || logical >= a.total_tiles) return false;
        const int mt = logical / a.n_tiles;
        m0 = (long long)mt * BM, n0 = (logical - mt * a.n_tiles) * BN;
        return true;
    };
    auto stage_issue = [&](int buf, int kt, long long m0, int n0) {
        float* xs = smem + buf * STAGE;
        float* ws = xs + BM * 16;
        const float* base = kt < a.k1p ? a.x1 : a.x2;
        const int kk = kt < a.k1p ? kt : kt - a.k1p;
        const float* src = base + ((long long)kk * a.m_padded + m0) * 16;
#pragma unroll
        for (int r = 0; r < XR; ++r) glds16(src + (r * 256 + tid) * 4, xs + (r * 256 + wave * 64) * 4);
        const float* wsrc = a.w + ((long long)kt * a.n_padded + n0) * 16;
#pragma unroll
        for (int r = 0; r < WR; ++r) glds16(wsrc + (r * 256 + tid) * 4, ws + (r * 256 + wave * 64) * 4);
    };

    long long m0 = 0;
    int n0 = 0;
    if (!tile_of(0, m0, n0)) return;
    if (dephase) {
        // the workgroups of one point tile (consecutive w) share a phase; 64 groups chip-wide -> phases k/64 of a tile time
        const int grp = (w / a.n_tiles) * 8 + xcd;
        const int units = ((grp * 37) & 63) * KT / 64;          // one unit = s_sleep 127 = 8128 cycles ~ one K panel of a shared SIMD
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);
    }
    stage_issue(0, 0, m0, n0);
    for (int it = 0;; ++it) {
        f32x16 acc[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        // Panel 0 of this tile has landed?  It was requested BEFORE the previous tile's NI*NJ*4 = 32 stores per lane, and on
        // gfx9 vector-memory operations of one wave retire IN ORDER (loads and stores share vmcnt; hipcc itself emits
        // vmcnt(N > 0) across younger stores), so vmcnt(32) = "everything older than the last 32 stores" = the panel, WITHOUT
        // waiting for the store acknowledgements of the chip-wide write burst (a plain __syncthreads() would: vmcnt(0)).
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) stage_issue(cur ^ 1, kt + 1, m0, n0);
            const float* xs = smem + cur * STAGE;
            mma_panel<NI, NJ>(xs, xs + BM * 16, wm * (32 * NJ), wn * 64, lane, acc);
            __syncthreads();
        }
        // (1) bias + ReLU applied IN PLACE to the accumulators (the wait for the bias happens here, before anything else is in
        // flight), (2) request the next tile's first panel, (3) this tile's 32 stores per lane, which need no wait at all.
        // (No wave reads LDS any more: the K loop ended on a barrier.)
        const bool perray = a.bias_row_div != 0;
        if (!perray) {
            f32x4 bv[NI][4];
            bias_fetch<NI>(a.bias, n0 + wn * 64, lane, bv);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[i][j][4 * q + e] + bv[i][q][e];
                            acc[i][j][4 * q + e] = a.relu ? relu_np(v) : v;
                        }
        }
        long long m1 = 0;
        int n1 = 0;
        const bool more = tile_of(it + 1, m1, n1);
        if (more) stage_issue(0, 0, m1, n1);
        if (perray) {
            f32x4 bv[NI][4];
            store_tile<NI, NJ, true, false>(acc, a.bias, a.bias_rows, a.bias_row_div, a.n_padded, a.y, a.m_padded,
                                            m0 + wm * (32 * NJ), n0 + wn * 64, a.relu, lane, bv);
        } else {
            const int lr = lane & 31, g = lane >> 5;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const long long m = m0 + wm * (32 * NJ) + 32 * j + lr;
                const int msw = (int)(m >> 2) & 3;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + wn * 64 + 32 * i + 8 * q + 4 * g;
                        f32x4 v;
                        v.x = acc[i][j][4 * q + 0], v.y = acc[i][j][4 * q + 1], v.z = acc[i][j][4 * q + 2], v.w = acc[i][j][4 * q + 3];
                        *(f32x4*)(a.y + (long long)(n >> 4) * a.m_padded * 16 + m * 16 + ((((n >> 2) & 3) ^ msw) << 2)) = v;
                    }
            }
        }
        if (!more) break;
        m0 = m1, n0 = n1;
    }
}

// ---- measurement aid: what the fp32 matrix pipe sustains with NO memory traffic, barriers or epilogue -----------------------
// 8 independent 32x32 accumulators per wave (the layer kernel's register blocking), iters x 64 MFMAs each.
// tools/microbench_layer.py --peak turns the time into TFLOP/s: 156 = 99 % of 157.3, with one OR two waves per SIMD.
__global__ __launch_bounds__(256, 2) void k_mfma_peak_probe(float* __restrict__ out, int iters, int random_operands) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // random_operands: every lane gets its own pseudo-random operand values (|v| ~ 1e-3 .. 1) and their mantissa bits are
    // re-scrambled with integer ops once per 64 MFMAs, so that the multiplier inputs toggle like real data instead of sitting
    // at two constants - the question being whether the pipe's sustained rate depends on the data
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h = h * 1664525u + 1013904223u;
        a[e] = random_operands ? __uint_as_float(0x3A000000u | (h & 0x05FFFFFFu) | ((h >> 3) & 0x80000000u)) : (float)threadIdx.x * 1e-3f;
        h = h * 1664525u + 1013904223u;
        b[e] = random_operands ? __uint_as_float(0x3A000000u | (h & 0x05FFFFFFu) | ((h >> 5) & 0x80000000u)) : (float)blockIdx.x * 1e-3f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[(e + i) & 7], acc[i], 0, 0, 0);
        }
        if (random_operands) {
            // random_operands == 2: the control - identical instruction stream, but the scramble keeps only bits that are
            // already set (mask 0), so the operands stay what they were
            const unsigned keep = random_operands == 2 ? 0u : 0x007FFFFFu;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h = h * 1664525u + 1013904223u;
                a[e] = __uint_as_float((__float_as_uint(a[e]) & ~keep) | (h & keep));
                b[e] = __uint_as_float((__float_as_uint(b[e]) & ~keep) | (((h >> 7) | (h << 3)) & keep));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(a[e]), "+v"(b[e]));
        }
    }
    float s_ = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_ += acc[i][r];
    if (s_ == 123.456f) out[0] = s_;   // keeps the accumulators alive
}

// ---- heads: sigma = sigmaCodes . w + b (model.py:130), rgb = v . W3 + b3 (model.py:134) ---------
__global__ __launch_bounds__(256) void k_head(const float* __restrict__ x, int kp, long long m_padded,
                                              const float* __restrict__ w, const float* __restrict__ b, int n_out,
                                              float* __restrict__ raw, int raw_off, long long n_points, int hh) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_points) return;
    const int sw = (int)(m >> 2) & 3;
    const int K = kp * 16;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < kp; ++kt) {
        const f32x4* row = (const f32x4*)(x + ((long long)kt * m_padded + m) * 16);
        float xv[16];
        if (hh) {   // pre-split fp16 piece panels (store_quad_hh): x = h1 + h2
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                const f16x8 h1 = __builtin_bit_cast(f16x8, row[(2 * G) ^ sw]);
                const f16x8 h2 = __builtin_bit_cast(f16x8, row[(2 * G + 1) ^ sw]);
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[8 * G + e] = (float)h1[e] + (float)h2[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 t = row[c ^ sw];
                xv[4 * c] = t.x, xv[4 * c + 1] = t.y, xv[4 * c + 2] = t.z, xv[4 * c + 3] = t.w;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 v;
            v.x = xv[4 * c], v.y = xv[4 * c + 1], v.z = xv[4 * c + 2], v.w = xv[4 * c + 3];
            const int k = kt * 16 + 4 * c;
            for (int o = 0; o < n_out; ++o) {
                const float* wo = w + (long long)o * K + k;
                acc[o] = fmaf(v.x, wo[0], acc[o]);
                acc[o] = fmaf(v.y, wo[1], acc[o]);
                acc[o] = fmaf(v.z, wo[2], acc[o]);
                acc[o] = fmaf(v.w, wo[3], acc[o]);
            }
        }
    }
    for (int o = 0; o < n_out; ++o) raw[m * 4 + raw_off + o] = acc[o] + b[o];
}

// ---- per-ray bias of the view layer: b + W[:, :27] @ PE(viewdir)  (model.py:133, render_class.py:88-90)
__global__ __launch_bounds__(256) void k_view_bias(const float* __restrict__ viewdirs, long long n_rays,
                                                   const float* __restrict__ w, int n_out, int ld,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   int n_padded) {
    constexpr int RPB = 8, NF = 3 + 6 * MOFA_PE_VIEW_FREQS;
    __shared__ float pe[RPB][NF + 1];
    const long long r0 = (long long)blockIdx.x * RPB;
    for (int t = threadIdx.x; t < RPB * NF; t += 256) {
        const int rr = t / NF, k = t - rr * NF;
        long long r = r0 + rr;
        if (r >= n_rays) r = n_rays - 1;
        pe[rr][k] = pe_feature(k, viewdirs[r * 3], viewdirs[r * 3 + 1], viewdirs[r * 3 + 2], NF);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < n_padded; n += 256) {
        float acc[RPB];
        const float b = n < n_out ? bias[n] : 0.f;
#pragma unroll
        for (int rr = 0; rr < RPB; ++rr) acc[rr] = 0.f;
        if (n < n_out) {
            for (int k = 0; k < NF; ++k) {
                const float wk = w[(long long)n * ld + k];
#pragma unroll
                for (int rr = 0; rr < RPB; ++rr) acc[rr] = fmaf(wk, pe[rr][k], acc[rr]);
            }
        }
#pragma unroll
        for (int rr = 0; rr < RPB; ++rr)
            if (r0 + rr < n_rays) out[(r0 + rr) * n_padded + n] = acc[rr] + b;
    }
}

// ---- folded per-call bias: out[n] = bias[n] + sum_c W[n, col0 + c] * code[c] -----------------------
__global__ __launch_bounds__(256) void k_fold_bias(const float* __restrict__ w, int n_out, int ld, int col0,
                                                   int ncols, const float* __restrict__ code,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   int n_padded) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_padded) return;
    float acc = 0.f;
    if (n < n_out) {
        for (int c = 0; c < ncols; ++c) acc = fmaf(w[(long long)n * ld + col0 + c], code[c], acc);
        acc += bias[n];
    }
    out[n] = acc;
}

// ---- weight / activation repacking -------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_panels(const float* __restrict__ w, int n_out, int ld, int col0,
                                                     int ncols, float* __restrict__ dst, int rows_padded,
                                                     int panel0, int k_padded) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)rows_padded * k_padded;
    if (idx >= total) return;
    const int e = idx & 3, p = (idx >> 2) & 3;
    const long long rowpanel = idx >> 4;
    const int row = (int)(rowpanel % rows_padded), panel = (int)(rowpanel / rows_padded);
    const int k = panel * 16 + ((p ^ ((row >> 2) & 3)) << 2) + e;
    const float v = (row < n_out && k < ncols) ? w[(long long)row * ld + col0 + k] : 0.f;
    dst[(long long)panel0 * rows_padded * 16 + idx] = v;
}

// transposed pack for the backward-data GEMM: dst rows = forward INPUT features (k), contraction = forward outputs (n):
//   dst(row = k, kk = n) = w[n, col0 + k]
__global__ __launch_bounds__(256) void k_pack_panels_t(const float* __restrict__ w, int n_out, int ld, int col0,
                                                       int ncols, float* __restrict__ dst, int rows_padded,
                                                       int k_padded) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)rows_padded * k_padded;
    if (idx >= total) return;
    const int e = idx & 3, p = (idx >> 2) & 3;
    const long long rowpanel = idx >> 4;
    const int row = (int)(rowpanel % rows_padded), panel = (int)(rowpanel / rows_padded);
    const int n = panel * 16 + ((p ^ ((row >> 2) & 3)) << 2) + e;
    dst[idx] = (row < ncols && n < n_out) ? w[(long long)n * ld + col0 + row] : 0.f;
}

__global__ __launch_bounds__(256) void k_to_panels(const float* __restrict__ x, long long rows, int k_in,
                                                   float* __restrict__ dst, long long rows_padded, int k_padded) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows_padded * k_padded) return;
    const int e = idx & 3, p = (idx >> 2) & 3;
    const long long rowpanel = idx >> 4;
    const long long row = rowpanel % rows_padded;
    const int panel = (int)(rowpanel / rows_padded);
    const int k = panel * 16 + ((p ^ ((int)(row >> 2) & 3)) << 2) + e;
    dst[idx] = (row < rows && k < k_in) ? x[row * k_in + k] : 0.f;
}

__global__ __launch_bounds__(256) void k_from_panels(const float* __restrict__ src, long long rows_padded,
                                                     long long rows, int k_out, float* __restrict__ x) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * k_out) return;
    const long long row = idx / k_out;
    const int k = (int)(idx - row * k_out);
    x[idx] = src[panel_index(rows_padded, row, k)];
}

__global__ __launch_bounds__(256) void k_dense_rows(const float* __restrict__ w, int n_out, int ld, int col0,
                                                    int ncols, float* __restrict__ dst, int k_padded) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out * k_padded) return;
    const int o = idx / k_padded, k = idx - o * k_padded;
    dst[idx] = k < ncols ? w[(long long)o * ld + col0 + k] : 0.f;
}

__global__ __launch_bounds__(256) void k_positional_encode(const float* __restrict__ x, long long n, int n_freqs,
                                                           float* __restrict__ out) {
    const int nf = 3 + 6 * n_freqs;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * nf) return;
    const long long r = idx / nf;
    const int k = (int)(idx - r * nf);
    out[idx] = pe_feature(k, x[r * 3], x[r * 3 + 1], x[r * 3 + 2], nf);
}

// ======================================================================================================
// Persistent whole-network kernel for widths <= 256 (coarse net; fine net of the 256x8 variant).
// One workgroup owns ALL features of a 128-point half tile, so consecutive layers of that tile depend only on
// this workgroup's own stores: the 2D+5 MFMA layers run back to back in ONE launch (no per-layer launch ramp/tail, no
// inter-workgroup synchronisation), each workgroup looping over its point tiles.  Same panels, same LDS-DMA pipeline and
// bit-identical arithmetic as k_layer; activations still round-trip through (L2-resident) global panels because a
// 256 x 256 fp32 tile (256 KiB) does not fit in the 160 KiB LDS.
// ======================================================================================================
constexpr int kMaxFusedLayers = 40;

struct FusedLayer {
    long long x1_off, x2_off, y_off;  // float offsets into the activation arena; x1_off < 0: layer 0 (positional encoding)
    long long w_off;                  // into the packed weights
    long long bias_off;               // into `folded` (bias_row_div == 0) or into `view_bias_rows`
    int k1p, k2p, n_padded, bias_row_div;
};

struct FusedArgs {
    const float* arena;   // activation buffers (workspace or tape)
    float* arena_w;
    const float* packed;
    const float* folded;
    const float* view_bias_rows;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    long long z_row_stride, n_points, m_padded, bias_rows;
    int S, n_layers, m_tiles;
    int pipe;             // 1: full 256-feature blocks use kloop_pipelined (MOFA_PIPE != 0)
    FusedLayer L[kMaxFusedLayers];
};

__global__ __launch_bounds__(256, 2) void k_mlp_fused(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // 4 waves side by side over the (<= 256) features, each 64 features x 128 points (8 accumulators of 32x32): a workgroup
    // owns ALL features of a 128-point half tile.  48 KiB of LDS and <= 256 VGPRs -> two INDEPENDENT workgroups per CU, so
    // one's barrier / LDS-latency bubbles hide under the other's MFMAs (an 8-wave, 256-point variant measured 2 % slower).
    constexpr int TM = 128, BNMAX = 256, NI = 2, NJ = 4;
    constexpr int STAGE = (TM + BNMAX) * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half_tiles = a.m_tiles * 2;

    for (int ht = blockIdx.x; ht < half_tiles; ht += gridDim.x) {
        const long long m0 = (long long)ht * TM;
        // layer 0: thread t generates features [8*(t>>7), +8) of each 16-wide panel for point row t & 127
        float px = 0.f, py = 0.f, pz = 0.f;
        {
            long long m = m0 + (tid & (TM - 1));
            if (m >= a.n_points) m = a.n_points - 1;
            if (a.pts) {
                px = a.pts[m * 3 + 0], py = a.pts[m * 3 + 1], pz = a.pts[m * 3 + 2];
            } else {
                const long long r = m / a.S;
                const int s = (int)(m - r * a.S);
                const float zz = a.z[r * a.z_row_stride + s];
                px = __fadd_rn(a.rays_o[r * 3 + 0], __fmul_rn(a.rays_d[r * 3 + 0], zz));
                py = __fadd_rn(a.rays_o[r * 3 + 1], __fmul_rn(a.rays_d[r * 3 + 1], zz));
                pz = __fadd_rn(a.rays_o[r * 3 + 2], __fmul_rn(a.rays_d[r * 3 + 2], zz));
            }
        }
        for (int li = 0; li < a.n_layers; ++li) {
            const FusedLayer& l = a.L[li];
            const bool l0 = l.x1_off < 0;
            const int KT = l.k1p + l.k2p;
            const int np = l.n_padded;
            const float* wbase = a.packed + l.w_off;
            // feature blocks of 256 (one for widths <= 256; a 1024-wide layer walks four, re-streaming its X half tile from L2)
            for (int nb = 0; nb * BNMAX < np; ++nb) {
            const int nbase = nb * BNMAX;
            const bool active = nbase + wn * 64 < np;   // wave-uniform: this wave's feature rows exist in this layer

            auto stage_issue = [&](int buf, int kt) {
                float* xs = smem + buf * STAGE;
                float* ws = xs + TM * 16;
                if (l0) {
                    const int row = tid & (TM - 1), k0 = (tid >> 7) * 8;
                    const int swz = (row >> 2) & 3;
#pragma unroll 1
                    for (int kk = k0; kk < k0 + 8; ++kk) {
                        const float v = pe_feature(kt * 16 + kk, px, py, pz, 3 + 6 * MOFA_PE_POINT_FREQS);
                        xs[row * 16 + ((((kk >> 2) & 3) ^ swz) << 2) + (kk & 3)] = v;
                    }
                } else {
                    const float* src = (kt < l.k1p ? a.arena + l.x1_off + ((long long)kt * a.m_padded + m0) * 16
                                                   : a.arena + l.x2_off + ((long long)(kt - l.k1p) * a.m_padded + m0) * 16);
#pragma unroll
                    for (int r = 0; r < 2; ++r) glds16(src + (r * 256 + tid) * 4, xs + (r * 256 + wn * 64) * 4);
                }
                const float* wsrc = wbase + ((long long)kt * np + nbase) * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((r * 256 + wn * 64) * 4 < (np - nbase) * 16)   // wave-uniform guard for (block remainders of) layers narrower than 256
                        glds16(wsrc + (r * 256 + tid) * 4, ws + (r * 256 + wn * 64) * 4);
            };

            f32x16 acc[NI][NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

            if (a.pipe && !l0 && np - nbase >= BNMAX && KT >= 4 && !(KT & 1)) {
                // full 256-feature block of an ordinary layer: the software-pipelined K loop of k_layer (bit-identical)
                kloop_pipelined<NI, NJ, TM, BNMAX>(a.arena + l.x1_off + m0 * 16, l.k2p ? a.arena + l.x2_off + m0 * 16 : nullptr, wbase + (long long)nbase * 16,
                                                   a.m_padded * 16, (long long)np * 16, l.k1p, KT, smem, tid, wn, lane, 0, wn * 64, acc);
                __syncthreads();    // every wave is done reading the stages before the next block / layer requests into them
            } else {
                stage_issue(0, 0);
                __syncthreads();
                for (int kt = 0; kt < KT; ++kt) {
                    const int cur = kt & 1;
                    if (kt + 1 < KT) stage_issue(cur ^ 1, kt + 1);
                    const float* xs = smem + cur * STAGE;
                    if (active) mma_panel<NI, NJ>(xs, xs + TM * 16, 0, wn * 64, lane, acc);
                    __syncthreads();
                }
            }
            if (active) {
                float* y = a.arena_w + l.y_off;
                f32x4 bv[NI][4];
                if (l.bias_row_div)
                    store_tile<NI, NJ, true, false>(acc, a.view_bias_rows + l.bias_off, a.bias_rows, l.bias_row_div, np, y, a.m_padded, m0,
                                                    nbase + wn * 64, 1, lane, bv);
                else {
#if MOFA_STAGED_EPILOGUE   // both loops end with a workgroup barrier: stage 0 is free, 4 KiB of it per wave
                    store_tile_staged<NI, NJ, true>(acc, a.folded + l.bias_off, y, a.m_padded, m0, nbase + wn * 64, lane, smem + wn * 1024);
#else
                    store_tile<NI, NJ, false, false>(acc, a.folded + l.bias_off, 1, 0, np, y, a.m_padded, m0, nbase + wn * 64, 1, lane, bv);
#endif
                }
            }
            // Another feature block of this layer follows (layers wider than 256): its first LDS-DMA requests land in stage 0, where the
            // staged epilogue's wave-private windows live — a faster wave must not overwrite a window its neighbour is still reading.
            if (nbase + BNMAX < np) __syncthreads();
            }   // feature blocks
            // this workgroup's stores of layer li feed its own loads of layer li+1
            __threadfence_block();
            __syncthreads();
        }
    }
}

int launch_fused(const FusedArgs& a, hipStream_t st) {
    const int cus = compute_units(current_device());
    const int half_tiles = a.m_tiles * 2;
    const int grid = half_tiles < 2 * cus ? half_tiles : 2 * cus;      // two resident workgroups per CU
    const size_t lds = 2 * (size_t)(128 + 256) * 16 * sizeof(float);   // 48 KiB
    hipLaunchKernelGGL(k_mlp_fused, dim3(grid), dim3(256), lds, st, a);
    return check_launch("k_mlp_fused");
}

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
#ifndef MOFA_SPLIT_PIPELINED
#define MOFA_SPLIT_PIPELINED 0   // 1: software-pipelined split + sched_group_barrier + alternating accumulators (A/B arm)
#endif
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
#if defined(MOFA_SPLIT_FAKE)      // measurement arm only (wrong numbers): zero-VALU "split" = upper bound of a producer-side split
    out[0] = __builtin_bit_cast(f16x8, lo);
    out[1] = __builtin_bit_cast(f16x8, hi);
    return;
#endif
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
#if !MOFA_SPLIT_PIPELINED
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
#else
        // software pipeline over the point blocks: split block j+1 (VALU) while block j's MFMAs issue
        f32x4 lo[NJ], hi[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wm * (32 * NJ) + 32 * j + lr;
            lo[j] = *(const f32x4*)(xs + row * 16 + (((2 * g) ^ sw) << 2));
            hi[j] = *(const f32x4*)(xs + row * 16 + (((2 * g + 1) ^ sw) << 2));
        }
        Frag xf[2][P];
        split_any<P, F16>(lo[0], hi[0], xf[0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // smallest partial products first; the two feature blocks alternate so that consecutive MFMAs never wait on
            // the accumulator the previous one is still producing
#pragma unroll
            for (int t = P - 1; t >= 0; --t)
#pragma unroll
                for (int pw = 0; pw <= t; ++pw)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        acc[i][j] = mfma_split<F16>(wf[i][pw], xf[j & 1][t - pw], acc[i][j]);
            if (j + 1 < NJ) {
                split_any<P, F16>(lo[j + 1], hi[j + 1], xf[(j + 1) & 1]);
                // ask the scheduler to interleave: 1 MFMA, then 4 VALU of the next block's split, ...
#pragma unroll
                for (int q = 0; q < NI * P * (P + 1) / 2; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
        }
#endif
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
#if !MOFA_SPLIT_PIPELINED
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
#else
        // software pipeline over the point blocks: split block j+1 (VALU) while block j's MFMAs issue
        f32x4 lo[NJ], hi[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wm * (32 * NJ) + 32 * j + lr;
            lo[j] = *(const f32x4*)(xs + row * 16 + (((2 * g) ^ sw) << 2));
            hi[j] = *(const f32x4*)(xs + row * 16 + (((2 * g + 1) ^ sw) << 2));
        }
        bf16x8 xf[2][P];
        split8<P>(lo[0], hi[0], xf[0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // smallest partial products first; the two feature blocks alternate so that consecutive MFMAs never wait on
            // the accumulator the previous one is still producing
#pragma unroll
            for (int t = P - 1; t >= 0; --t)
#pragma unroll
                for (int pw = 0; pw <= t; ++pw)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][pw], xf[j & 1][t - pw], acc[i][j], 0, 0, 0);
            if (j + 1 < NJ) {
                split8<P>(lo[j + 1], hi[j + 1], xf[(j + 1) & 1]);
                // ask the scheduler to interleave: 1 MFMA, then 4 VALU of the next block's split, ...
#pragma unroll
                for (int q = 0; q < NI * P * (P + 1) / 2; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
        }
#endif
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

inline int stage_mode() { return config().stage_glds; }  // MOFA_STAGE=reg selects the register-staged A/B arm

// Optional per-launch timing of the dominant kernel (k_layer<128, false, *>) with HIP events recorded on the launch
// stream; used by bench.py for the live roofline figure.  Off by default (no events, no overhead).
// The measurement session is explicit state the HOST opens and closes (mofa_prof_begin/end); it is kept per device and
// guarded by a mutex, so two devices or two host threads in one process do not share or corrupt it.  When no session is
// open the launch paths only read one relaxed atomic.
constexpr int kProfKinds = MOFA_PROF_KINDS;   // 0: k_layer<128,false,true> (forward), 1: k_mlp_fused, 2: k_layer<BWD>, 3: k_wgrad, 4: k_layer<..PERRAY> (view layer)
struct ProfState {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<int> kind;
    size_t used = 0;
    double flops[kProfKinds] = {};
};
ProfState g_prof[kMaxDevices];
std::atomic<bool> g_prof_on[kMaxDevices];
std::mutex g_prof_mu;

inline bool prof_enabled() { return g_prof_on[current_device()].load(std::memory_order_relaxed); }

inline int prof_open(hipStream_t st, int kind) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfState& P = g_prof[current_device()];
    if (P.used == P.ev.size()) {
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return check_launch("hipEventCreate");
        P.ev.emplace_back(e0, e1);
        P.kind.push_back(kind);
    }
    P.kind[P.used] = kind;
    (void)hipEventRecord(P.ev[P.used].first, st);
    return MOFA_OK;
}
inline void prof_close(hipStream_t st, int kind, double flops) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfState& P = g_prof[current_device()];
    (void)hipEventRecord(P.ev[P.used].second, st);
    P.used++;
    P.flops[kind] += flops;
}

#ifdef MOFA_TIMELINE
unsigned long long* g_timeline = nullptr;   // measurement build only: set by mofa_internal_set_timeline
#endif

template <int BN, bool L0, bool BWD = false>
int launch_layer(LayerArgs a, hipStream_t st) {
#ifdef MOFA_TIMELINE
    a.timeline = (BN == 128 && !L0 && !BWD) ? g_timeline : nullptr;
#endif
    a.n_tiles = a.n_padded / BN;
    const long long mt = a.m_padded / kRowTile;
    const long long total = mt * a.n_tiles;
    MOFA_REQUIRE(total > 0 && total < (1ll << 30), "layer: tile count %lld out of range", total);
    a.total_tiles = (int)total;
    const unsigned grid = (unsigned)round_up(total, 8);
    size_t lds = 2 * (size_t)(kRowTile + BN) * 16 * sizeof(float) + (size_t)(config().lds_pad > 0 ? config().lds_pad : 0);
#ifdef MOFA_TIMELINE
    lds += 512;     // per-panel stamps of the measurement build
#endif
    const bool prof = BN == 128 && !L0 && prof_enabled();
    const int pkind = BWD ? 2 : (a.bias_row_div ? 4 : 0);   // the view layer's per-ray-bias instantiation is its own kernel
    if (prof && prof_open(st, pkind) != MOFA_OK) return MOFA_EHIP;
    bool launched = false;
    if constexpr (L0 && BN == 128) {
        if (a.y_hh) {   // opt-in fp16x3 mode: the first layer feeds a split-product layer, so it writes piece panels
            hipLaunchKernelGGL((k_layer<BN, true, true, false, true>), dim3(grid), dim3(256), lds, st, a);
            launched = true;
        }
    }
    if constexpr (BN == 128 && !L0 && !BWD) {
        const Config& cfg = config();
        if (cfg.ring3 == 1 && stage_mode()) {   // MOFA_RING3=1: the 3-stage-ring twin (bit-identical; A/B arm)
            const size_t lds3 = 3 * (size_t)(kRowTile + BN) * 16 * sizeof(float);
            static std::atomic<bool> attr3[kMaxDevices];
            const int dev = current_device();
            if (!attr3[dev].load(std::memory_order_acquire)) {
                if (hipFuncSetAttribute((const void*)k_layer_ring3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess ||
                    hipFuncSetAttribute((const void*)k_layer_ring3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess)
                    return check_launch("hipFuncSetAttribute(k_layer_ring3)");
                attr3[dev].store(true, std::memory_order_release);
            }
            if (a.bias_row_div) hipLaunchKernelGGL((k_layer_ring3<true>), dim3(grid), dim3(256), lds3, st, a);
            else hipLaunchKernelGGL((k_layer_ring3<false>), dim3(grid), dim3(256), lds3, st, a);
            launched = true;
        }
        // MOFA_PERSIST=1: the persistent twin (bit-identical; A/B arm until it is measured faster)
        if (!launched && cfg.persist == 1 && stage_mode()) {
            const int cus = compute_units(current_device());
            long long G = 2LL * cus / 8 * 8;                                  // two resident workgroups per CU, multiple of 8 XCDs
            const int per_xcd_tiles = (int)((total + 7) / 8);
            if (G > 8LL * per_xcd_tiles) G = 8LL * per_xcd_tiles;
            hipLaunchKernelGGL(k_layer_persist, dim3((unsigned)G), dim3(256), lds, st, a, per_xcd_tiles, cfg.dephase);
            launched = true;
        }
    }
    // software-pipelined K loop (kloop_pipelined; bit-identical to the plain loop): every 128-feature layer whose panel count
    // is even and >= 4, unless MOFA_PIPE=0 or the register-staged arm is selected
    bool pipe = false;
    if constexpr (BN == 128 && !L0) pipe = config().pipe != 0 && stage_mode() && (a.k1p + a.k2p) >= 4 && ((a.k1p + a.k2p) & 1) == 0;
    if (launched) {
    } else if constexpr (BWD) {
        if constexpr (BN == 128) {
            if (pipe) {
                hipLaunchKernelGGL((k_layer<BN, false, true, true, false, false, true>), dim3(grid), dim3(256), lds, st, a);
                launched = true;
            }
        }
        if (!launched) hipLaunchKernelGGL((k_layer<BN, false, true, true>), dim3(grid), dim3(256), lds, st, a);
    } else if (!L0 && a.bias_row_div) {  // per-ray bias (the view layer): its own instantiation, see store_tile
        if constexpr (BN == 128 && !L0) {
            if (pipe) {
                hipLaunchKernelGGL((k_layer<BN, false, true, false, false, true, true>), dim3(grid), dim3(256), lds, st, a);
                launched = true;
            }
        }
        if (!launched) hipLaunchKernelGGL((k_layer<BN, false, true, false, false, true>), dim3(grid), dim3(256), lds, st, a);
    } else if (pipe) {
        if constexpr (BN == 128 && !L0)
            hipLaunchKernelGGL((k_layer<BN, false, true, false, false, false, true>), dim3(grid), dim3(256), lds, st, a);
    }
    else if (stage_mode())
        hipLaunchKernelGGL((k_layer<BN, L0, true>), dim3(grid), dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL((k_layer<BN, L0, false>), dim3(grid), dim3(256), lds, st, a);
    if (prof) prof_close(st, pkind, 2.0 * (double)a.m_padded * (double)a.n_padded * 16.0 * (double)(a.k1p + a.k2p));
    return check_launch(BWD ? "k_layer<BWD>" : (L0 ? "k_layer<L0>" : "k_layer"));
}

template <int P, bool F16 = false, bool HH = false>
int launch_layer_split(LayerArgs a, const unsigned short* ws, hipStream_t st) {
    constexpr int BN = 128;
    a.n_tiles = a.n_padded / BN;
    const long long total = (a.m_padded / kRowTile) * a.n_tiles;
    MOFA_REQUIRE(total > 0 && total < (1ll << 30), "layer_split: tile count %lld out of range", total);
    a.total_tiles = (int)total;
    SplitArgs sa{a, ws};
    const bool prof = prof_enabled();
    if (prof && prof_open(st, 0) != MOFA_OK) return MOFA_EHIP;
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
    if (prof) prof_close(st, 0, 2.0 * (double)a.m_padded * (double)a.n_padded * 16.0 * (double)(a.k1p + a.k2p));
    return check_launch("k_layer_split");
}

int dispatch_layer_bwd(LayerArgs a, hipStream_t st) {
    MOFA_REQUIRE(a.m_padded > 0 && a.m_padded % kRowTile == 0, "m_padded=%lld must be a positive multiple of %d",
                 a.m_padded, kRowTile);
    MOFA_REQUIRE(a.n_padded > 0 && a.n_padded % 64 == 0, "n_padded=%d must be a positive multiple of 64", a.n_padded);
    if (a.n_padded % 128 == 0) return launch_layer<128, false, true>(a, st);
    return launch_layer<64, false, true>(a, st);
}

int dispatch_layer(LayerArgs a, bool l0, hipStream_t st) {
    MOFA_REQUIRE(a.m_padded > 0 && a.m_padded % kRowTile == 0, "m_padded=%lld must be a positive multiple of %d",
                 a.m_padded, kRowTile);
    MOFA_REQUIRE(a.n_padded > 0 && a.n_padded % 64 == 0, "n_padded=%d must be a positive multiple of 64", a.n_padded);
    const bool force64 = config().bn64 != 0;   // measurement knob (tools/microbench_layer.py)
    if (a.n_padded % 128 == 0 && !force64) return l0 ? launch_layer<128, true>(a, st) : launch_layer<128, false>(a, st);
    return l0 ? launch_layer<64, true>(a, st) : launch_layer<64, false>(a, st);
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace
}  // namespace mofa

using namespace mofa;

extern "C" {

size_t mofa_panel_floats(int64_t rows, int32_t k) { return (size_t)rows * (size_t)round_up(k, 16); }

int mofa_pack_panels(const float* w, int32_t n_out, int32_t ld, int32_t col0, int32_t ncols, float* dst,
                     int32_t rows_padded, int32_t panel0, int32_t k_padded, void* stream) {
    MOFA_REQUIRE(w && dst, "pack_panels: null pointer");
    MOFA_REQUIRE(rows_padded >= n_out && k_padded % 16 == 0 && k_padded >= ncols && col0 >= 0 && col0 + ncols <= ld,
                 "pack_panels: bad shape n_out=%d ld=%d col0=%d ncols=%d rows_padded=%d k_padded=%d", n_out, ld, col0,
                 ncols, rows_padded, k_padded);
    const long long total = (long long)rows_padded * k_padded;
    hipLaunchKernelGGL(k_pack_panels, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, w, n_out, ld, col0,
                       ncols, dst, rows_padded, panel0, k_padded);
    return check_launch("k_pack_panels");
}

int mofa_to_panels(const float* x, int64_t rows, int32_t k, float* dst, int64_t rows_padded, void* stream) {
    MOFA_REQUIRE(x && dst && rows > 0 && rows_padded >= rows && k > 0, "to_panels: bad arguments");
    const int kp = (int)round_up(k, 16);
    hipLaunchKernelGGL(k_to_panels, dim3(blocks_for(rows_padded * kp)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)rows, k, dst, (long long)rows_padded, kp);
    return check_launch("k_to_panels");
}

int mofa_from_panels(const float* src, int64_t rows_padded, int64_t rows, int32_t k, float* x, void* stream) {
    MOFA_REQUIRE(x && src && rows > 0 && rows_padded >= rows && k > 0, "from_panels: bad arguments");
    hipLaunchKernelGGL(k_from_panels, dim3(blocks_for(rows * k)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)rows_padded, (long long)rows, k, x);
    return check_launch("k_from_panels");
}

int mofa_layer_forward(const float* x1, int32_t k1, const float* x2, int32_t k2, const float* w_packed,
                       const float* bias, int32_t bias_row_div, int64_t bias_rows, float* y, int64_t m_padded,
                       int32_t n_padded, int32_t relu, void* stream) {
    MOFA_REQUIRE(x1 && w_packed && bias && y, "layer_forward: null pointer");
    MOFA_REQUIRE(k1 > 0 && k1 % 16 == 0 && k2 >= 0 && k2 % 16 == 0 && (k2 == 0 || x2),
                 "layer_forward: k1=%d k2=%d must be multiples of 16 (x2 required when k2>0)", k1, k2);
    MOFA_REQUIRE(bias_row_div >= 0 && (bias_row_div == 0 || bias_rows > 0), "layer_forward: bad bias rows");
    LayerArgs a{};
    a.x1 = x1, a.x2 = x2, a.w = w_packed, a.bias = bias, a.y = y;
    a.k1p = k1 / 16, a.k2p = k2 / 16, a.n_padded = n_padded, a.m_padded = m_padded;
    a.bias_row_div = bias_row_div, a.bias_rows = bias_rows, a.relu = relu;
    return dispatch_layer(a, false, (hipStream_t)stream);
}

/* dX = G @ W (optionally += and * ReLU mask): g panels [n_padded_fwd/16][Mp][16], wt_packed = transposed pack
 * (rows = forward input features padded to k_out_padded, contraction = g_k), dx panels [k_out_padded/16][Mp][16]. */
int mofa_layer_backward_data(const float* g, int32_t g_k, const float* wt_packed, const float* mask, int32_t accumulate,
                             float* dx, int64_t m_padded, int32_t k_out_padded, void* stream) {
    MOFA_REQUIRE(g && wt_packed && dx, "layer_backward_data: null pointer");
    MOFA_REQUIRE(g_k > 0 && g_k % 16 == 0, "layer_backward_data: g_k=%d must be a positive multiple of 16", g_k);
    LayerArgs a{};
    a.x1 = g, a.w = wt_packed, a.y = dx, a.mask = mask, a.accumulate = accumulate;
    a.k1p = g_k / 16, a.k2p = 0, a.n_padded = k_out_padded, a.m_padded = m_padded;
    return dispatch_layer_bwd(a, (hipStream_t)stream);
}

int mofa_pack_panels_t(const float* w, int32_t n_out, int32_t ld, int32_t col0, int32_t ncols, float* dst,
                       int32_t rows_padded, int32_t k_padded, void* stream) {
    MOFA_REQUIRE(w && dst, "pack_panels_t: null pointer");
    MOFA_REQUIRE(rows_padded >= ncols && k_padded % 16 == 0 && k_padded >= n_out && col0 >= 0 && col0 + ncols <= ld,
                 "pack_panels_t: bad shape n_out=%d ld=%d col0=%d ncols=%d rows_padded=%d k_padded=%d", n_out, ld, col0,
                 ncols, rows_padded, k_padded);
    const long long total = (long long)rows_padded * k_padded;
    hipLaunchKernelGGL(k_pack_panels_t, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, w, n_out, ld, col0,
                       ncols, dst, rows_padded, k_padded);
    return check_launch("k_pack_panels_t");
}

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

int mofa_layer0_forward(const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                        const float* pts, int64_t n_points, int32_t S, const float* w_packed, const float* bias,
                        float* y, int64_t m_padded, int32_t n_padded, void* stream) {
    MOFA_REQUIRE(w_packed && bias && y, "layer0_forward: null pointer");
    MOFA_REQUIRE(pts || (rays_o && rays_d && z && S > 0), "layer0_forward: need pts or (rays_o, rays_d, z, S)");
    MOFA_REQUIRE(n_points > 0 && n_points <= m_padded, "layer0_forward: n_points=%lld m_padded=%lld",
                 (long long)n_points, (long long)m_padded);
    LayerArgs a{};
    a.w = w_packed, a.bias = bias, a.y = y, a.rays_o = rays_o, a.rays_d = rays_d, a.z = z, a.pts = pts;
    a.z_row_stride = z_row_stride, a.n_points = n_points, a.S = S > 0 ? S : 1;
    a.k1p = 4, a.k2p = 0, a.n_padded = n_padded, a.m_padded = m_padded, a.relu = 1;
    return dispatch_layer(a, true, (hipStream_t)stream);
}

int mofa_layer0_forward_cam(int32_t img_w, float fx, float fy, float cx, float cy, const float* c2w, const int32_t* pixels,
                            int64_t pix0, const float* z, int64_t z_row_stride, int64_t n_points, int32_t S,
                            const float* w_packed, const float* bias, float* y, int64_t m_padded, int32_t n_padded, void* stream) {
    MOFA_REQUIRE(c2w && z && w_packed && bias && y && img_w > 0 && S > 0, "layer0_forward_cam: bad arguments");
    MOFA_REQUIRE(n_points > 0 && n_points <= m_padded, "layer0_forward_cam: n_points=%lld m_padded=%lld", (long long)n_points,
                 (long long)m_padded);
    LayerArgs a{};
    a.w = w_packed, a.bias = bias, a.y = y, a.z = z, a.z_row_stride = z_row_stride, a.n_points = n_points, a.S = S;
    a.cam_c2w = c2w, a.cam_pix = (const int*)pixels, a.cam_pix0 = pix0, a.fx = fx, a.fy = fy, a.cx = cx, a.cy = cy, a.cam_w = img_w;
    a.k1p = 4, a.k2p = 0, a.n_padded = n_padded, a.m_padded = m_padded, a.relu = 1;
    return dispatch_layer(a, true, (hipStream_t)stream);
}

int mofa_head_forward(const float* x, int32_t k_padded, int64_t m_padded, const float* w_dense, const float* b,
                      int32_t n_out, float* raw, int32_t raw_off, int64_t n_points, void* stream) {
    MOFA_REQUIRE(x && w_dense && b && raw, "head_forward: null pointer");
    MOFA_REQUIRE(k_padded % 16 == 0 && n_out >= 1 && n_out <= 4 && raw_off >= 0 && raw_off + n_out <= 4 &&
                     n_points <= m_padded,
                 "head_forward: bad shape");
    hipLaunchKernelGGL(k_head, dim3(blocks_for(n_points)), dim3(256), 0, (hipStream_t)stream, x, k_padded / 16,
                       (long long)m_padded, w_dense, b, n_out, raw, raw_off, (long long)n_points, 0);
    return check_launch("k_head");
}

int mofa_view_bias(const float* viewdirs, int64_t n_rays, const float* w, int32_t n_out, int32_t ld,
                   const float* bias, float* out, int32_t n_padded, void* stream) {
    MOFA_REQUIRE(viewdirs && w && bias && out && n_rays > 0 && n_padded >= n_out, "view_bias: bad arguments");
    hipLaunchKernelGGL(k_view_bias, dim3((unsigned)((n_rays + 7) / 8)), dim3(256), 0, (hipStream_t)stream, viewdirs,
                       (long long)n_rays, w, n_out, ld, bias, out, n_padded);
    return check_launch("k_view_bias");
}

int mofa_positional_encode(const float* x, int64_t n, int32_t n_freqs, float* out, void* stream) {
    MOFA_REQUIRE(x && out && n > 0 && n_freqs >= 0 && n_freqs <= 16, "positional_encode: bad arguments");
    hipLaunchKernelGGL(k_positional_encode, dim3(blocks_for(n * (3 + 6 * n_freqs))), dim3(256), 0,
                       (hipStream_t)stream, x, (long long)n, n_freqs, out);
    return check_launch("k_positional_encode");
}

int mofa_prof_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const int dev = current_device();
    ProfState& P = g_prof[dev];
    P.used = 0;
    for (int k = 0; k < kProfKinds; ++k) P.flops[k] = 0.0;
    g_prof_on[dev].store(true, std::memory_order_relaxed);
    return MOFA_OK;
}

/* arrays of MOFA_PROF_KINDS: [0] the per-layer forward MFMA kernel k_layer<128,false,true,false,false,false> (or its
 * persistent twin), [1] the persistent network kernel k_mlp_fused, [2] the backward-data kernel k_layer<128,..,BWD>, [3] the
 * weight-gradient kernel k_wgrad, [4] the per-ray-bias instantiation k_layer<128,false,true,false,false,true> (view layer).
 * Session of the CURRENT device. */
int mofa_prof_end(double* total_ms, int64_t* launches, double* padded_flops) {
    MOFA_REQUIRE(total_ms && launches && padded_flops, "prof_end: null pointer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const int dev = current_device();
    ProfState& P = g_prof[dev];
    g_prof_on[dev].store(false, std::memory_order_relaxed);
    for (int k = 0; k < kProfKinds; ++k) total_ms[k] = 0.0, launches[k] = 0;
    for (size_t i = 0; i < P.used; ++i) {
        if (hipEventSynchronize(P.ev[i].second) != hipSuccess) return check_launch("hipEventSynchronize");
        float t = 0.f;
        if (hipEventElapsedTime(&t, P.ev[i].first, P.ev[i].second) != hipSuccess)
            return check_launch("hipEventElapsedTime");
        total_ms[P.kind[i]] += (double)t, launches[P.kind[i]] += 1;
    }
    for (int k = 0; k < kProfKinds; ++k) padded_flops[k] = P.flops[k];
    P.used = 0;
    return MOFA_OK;
}

/* internal (mofa_bwd.hip): bracket a k_wgrad launch with events when a measurement session is open */
int mofa_internal_prof_open(void* stream, int kind) { return prof_enabled() ? (prof_open((hipStream_t)stream, kind) == MOFA_OK ? 1 : MOFA_EHIP) : 0; }
void mofa_internal_prof_close(void* stream, int kind, double flops) { prof_close((hipStream_t)stream, kind, flops); }

// internal (used by mofa_net.hip): run a list of MFMA layers of one network (all widths <= 256) as ONE persistent launch
// ---- opt-in fp16x3 mode with pre-split ("hh") activation panels: internal to mofa_net_forward --------------------------
int mofa_internal_layer0_forward_hh(const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                                    const float* pts, int64_t n_points, int32_t S, const float* w_packed, const float* bias,
                                    float* y, int64_t m_padded, int32_t n_padded, void* stream) {
    MOFA_REQUIRE(n_padded % 128 == 0, "layer0_forward_hh: n_padded %% 128 != 0");
    LayerArgs a{};
    a.w = w_packed, a.bias = bias, a.y = y, a.rays_o = rays_o, a.rays_d = rays_d, a.z = z, a.pts = pts;
    a.z_row_stride = z_row_stride, a.n_points = n_points, a.S = S > 0 ? S : 1;
    a.k1p = 4, a.k2p = 0, a.n_padded = n_padded, a.m_padded = m_padded, a.relu = 1, a.y_hh = 1;
    return launch_layer<128, true>(a, (hipStream_t)stream);
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

int mofa_internal_head_forward_hh(const float* x, int32_t k_padded, int64_t m_padded, const float* w_dense, const float* b,
                                  int32_t n_out, float* raw, int32_t raw_off, int64_t n_points, void* stream) {
    hipLaunchKernelGGL(k_head, dim3(blocks_for(n_points)), dim3(256), 0, (hipStream_t)stream, x, k_padded / 16,
                       (long long)m_padded, w_dense, b, n_out, raw, raw_off, (long long)n_points, 1);
    return check_launch("k_head(hh)");
}

#ifdef MOFA_TIMELINE
/* measurement build only: per-tile stamps [tiles][8] u64 = {entry, first panel landed, K loop done, stores issued, HW_ID, XCC_ID,
 * clock64 ticks in the K loop, -} */
int mofa_internal_set_timeline(unsigned long long* buf) {
    g_timeline = buf;
    return MOFA_OK;
}
#endif

/* measurement aid (tools/microbench_layer.py --peak): `blocks` workgroups of 4 waves running iters x 64 fp32 MFMAs each */
int mofa_internal_mfma_peak_probe(float* out, int32_t blocks, int32_t iters, int32_t random_operands, void* stream) {
    hipLaunchKernelGGL(k_mfma_peak_probe, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, iters, random_operands);
    return check_launch("k_mfma_peak_probe");
}

int mofa_internal_fused_forward(const float* arena, float* arena_w, const float* packed, const float* folded,
                                const float* view_bias_rows, long long bias_rows, const float* rays_o, const float* rays_d,
                                const float* z, long long z_row_stride, const float* pts, long long n_points, int S,
                                long long m_padded, int n_layers, const long long* x1_off, const long long* x2_off,
                                const long long* y_off, const long long* w_off, const long long* bias_off, const int* k1p,
                                const int* k2p, const int* n_padded, const int* bias_row_div, void* stream) {
    MOFA_REQUIRE(n_layers > 0 && n_layers <= kMaxFusedLayers, "fused_forward: %d layers (max %d)", n_layers, kMaxFusedLayers);
    MOFA_REQUIRE(m_padded > 0 && m_padded % kRowTile == 0, "fused_forward: m_padded=%lld", m_padded);
    FusedArgs a{};
    a.arena = arena, a.arena_w = arena_w, a.packed = packed, a.folded = folded, a.view_bias_rows = view_bias_rows;
    a.rays_o = rays_o, a.rays_d = rays_d, a.z = z, a.pts = pts;
    a.z_row_stride = z_row_stride, a.n_points = n_points, a.m_padded = m_padded, a.bias_rows = bias_rows;
    a.S = S > 0 ? S : 1, a.n_layers = n_layers, a.m_tiles = (int)(m_padded / kRowTile);
    a.pipe = config().pipe != 0 ? 1 : 0;
    for (int i = 0; i < n_layers; ++i) {
        MOFA_REQUIRE(n_padded[i] > 0 && n_padded[i] % 64 == 0, "fused_forward: layer %d has n_padded=%d", i, n_padded[i]);
        a.L[i] = FusedLayer{x1_off[i], x2_off[i], y_off[i], w_off[i], bias_off[i], k1p[i], k2p[i], n_padded[i], bias_row_div[i]};
    }
    hipStream_t st = (hipStream_t)stream;
    if (!prof_enabled()) return launch_fused(a, st);
    double flops = 0.0;
    for (int i = 0; i < n_layers; ++i) flops += 2.0 * (double)m_padded * (double)n_padded[i] * 16.0 * (double)(k1p[i] + k2p[i]);
    if (prof_open(st, 1) != MOFA_OK) return MOFA_EHIP;
    const int rc = launch_fused(a, st);
    prof_close(st, 1, flops);
    return rc;
}

// internal (used by mofa_net.hip)
int mofa_internal_fold_bias(const float* w, int n_out, int ld, int col0, int ncols, const float* code,
                            const float* bias, float* out, int n_padded, void* stream) {
    hipLaunchKernelGGL(k_fold_bias, dim3(blocks_for(n_padded)), dim3(256), 0, (hipStream_t)stream, w, n_out, ld, col0,
                       ncols, code, bias, out, n_padded);
    return check_launch("k_fold_bias");
}

int mofa_internal_dense_rows(const float* w, int n_out, int ld, int col0, int ncols, float* dst, int k_padded,
                             void* stream) {
    hipLaunchKernelGGL(k_dense_rows, dim3(blocks_for((long long)n_out * k_padded)), dim3(256), 0, (hipStream_t)stream,
                       w, n_out, ld, col0, ncols, dst, k_padded);
    return check_launch("k_dense_rows");
}

}  // extern "C"
