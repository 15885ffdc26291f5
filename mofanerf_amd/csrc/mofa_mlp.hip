// Product translation unit of the network kernels (gfx950): the per-layer launcher of k_layer (mofa_layer.h) with the shipped
// policy only, the persistent whole-network kernel k_mlp_fused for widths <= 256, the chained launch k_net_chain of the wider networks (the
// layer kernel's tiles behind per-XCD queues and row-tile dependency counters), the small kernels around them (heads, per-ray
// view bias, folded biases, panel packing, positional encoding) and their C ABI.  No measurement arms live here: scheduling
// experiments, time-stamp builds and ablations are csrc/measure/mofa_measure.hip (built only by tools/build_measure.py into
// its own library).
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "mofa_layer.h"
#include "mofa_wgrad.h"

namespace mofa {
namespace {

// ---- heads: sigma = sigmaCodes . w + b (model.py:130), rgb = v . W3 + b3 (model.py:134) ---------
__global__ __launch_bounds__(256) void k_head(const float* __restrict__ x, int kp, long long m_padded,
                                              const float* __restrict__ w, const float* __restrict__ b, int n_out,
                                              float* __restrict__ raw, int raw_off, long long n_points) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_points) return;
    const int sw = (int)(m >> 2) & 3;
    const int K = kp * 16;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < kp; ++kt) {
        const f32x4* row = (const f32x4*)(x + ((long long)kt * m_padded + m) * 16);
        float xv[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 t = row[c ^ sw];
            xv[4 * c] = t.x, xv[4 * c + 1] = t.y, xv[4 * c + 2] = t.z, xv[4 * c + 3] = t.w;
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
                                                   int n_padded, int NF) {
    constexpr int RPB = 8, NFMAX = 3 + 6 * MOFA_MAX_PE_FREQS;
    __shared__ float pe[RPB][NFMAX + 1];
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

// ---- mask-only tape, generic writer: bits of (y > 0) for a whole panel buffer (layers whose epilogue is not the contiguous-store
// one: the first layer, the per-ray-bias view layer, narrow / short-K launches).  One wavefront per KiB block, see mask_store_block.
__global__ __launch_bounds__(256) void k_mask_pack(const float* __restrict__ y, long long n_blocks, unsigned long long* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    for (long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_blocks; b += (long long)gridDim.x * 4) {
        const f32x4 v = *(const f32x4*)(y + b * 256 + lane * 4);
        mask_store_block(bits, b * 256, lane, v);
    }
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
    long long mask_off;               // mask-only tape: 64-bit word offset of this layer's bits (MASKW kernels only)
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
    int pe_feats;         // 3 + 6 * multires
    unsigned long long* mask_bits;   // mask-only tape (MASKW) or NULL
    FusedLayer L[kMaxFusedLayers];
};

template <bool MASKW>      // MASKW: also leave (y > 0) of every layer as bits (fitting's mask-only tape); the inference kernel is MASKW = false
__global__ __launch_bounds__(256, 2) void k_mlp_fused_generic(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // 4 waves side by side over the (<= 256) features, each 64 features x 128 points (8 accumulators of 32x32): a workgroup
    // owns ALL features of a 128-point half tile.  48 KiB of LDS and <= 256 VGPRs -> two INDEPENDENT workgroups per CU, so
    // one's barrier / LDS-latency bubbles hide under the other's MFMAs (an 8-wave, 256-point variant measured 2 % slower).
    constexpr int TM = 128, BNMAX = 256, NI = 2, NJ = 4;
    constexpr int STAGE = (TM + BNMAX) * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half_tiles = a.m_tiles * 2;
    ShippedPolicy::Probe probe(LayerArgs{}, 0, tid, smem, 0);      // (empty hooks; the K loop takes one by reference)

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
                        const float v = pe_feature(kt * 16 + kk, px, py, pz, a.pe_feats);
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
                kloop_pipelined<NI, NJ, TM, BNMAX, ShippedPolicy, 0, false>(a.arena + l.x1_off + m0 * 16, l.k2p ? a.arena + l.x2_off + m0 * 16 : nullptr, wbase + (long long)nbase * 16,
                                                   a.m_padded * 16, (long long)np * 16, l.k1p, KT, smem, tid, wn, lane, 0, wn * 64, acc, probe);
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
                    store_tile<NI, NJ, true>(acc, a.view_bias_rows + l.bias_off, a.bias_rows, l.bias_row_div, np, y, a.m_padded, m0,
                                                    nbase + wn * 64, 1, lane, bv);
                else      // both loops end with a workgroup barrier: stage 0 is free, 4 KiB of it per wave
                    store_tile_staged<NI, NJ, true, MASKW>(acc, a.folded + l.bias_off, y, a.m_padded, m0, nbase + wn * 64, lane, smem + wn * 1024,
                                                           MASKW ? a.mask_bits + l.mask_off : nullptr);
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


// ======================================================================================================
// k_mlp_fused: the persistent kernel for the shape the north star names — every ordinary layer 256 wide (width 193..256 pads to
// it), a 128-wide view layer last, the generated-operand layer first — with the layer BOUNDARIES taken off the critical path.
//
// In the generic kernel above a boundary is a chain of exposed latencies: the epilogue's bias fetch (one L2 round trip), the drain
// of its 128 KiB of stores (`__threadfence_block`), the barrier, and the next layer's first two operand panels (another L2 round
// trip) with nothing in flight — 21 times per 128-point tile, ~15 % of the kernel (PMC).  Here:
//   * the next layer's weight panel 0 is requested inside the LAST half panel of this layer's K loop (its stage is free by then),
//     panel 1 and the next layer's bias row right after the loop's closing barrier — all before the epilogue starts;
//   * the next layer's activation panels 0 and 1 never travel: they are the first two 16-feature slices of THIS layer's output,
//     which the contiguous-store epilogue already forms in LDS in exactly the operand layout — wave 0 forms them directly in the
//     two stages' X regions (and stores them to global from there like every other slice);
//   * the epilogue reads its bias row from LDS (requested one layer earlier);
//   * so the next K loop starts on resident operands straight after one barrier, and nothing waits for the stores: the loop's
//     first `vmcnt(0)` + barrier (half a panel later, before the first global activation panel is requested) is what orders
//     this workgroup's stores before its own loads — the same guarantee the fence gave;
//   * the view layer maps its 128 features x 128 points onto 2 x 2 waves (the generic kernel leaves two waves idle there) and
//     runs the same pipelined loop.
// Same panels, same MFMA order, same epilogue arithmetic: bit-identical to the generic kernel and to per-layer launches.
// LDS: two 24 KiB stages + four 4 KiB wave-private windows + two 1 KiB bias rows = 66 KiB, two workgroups per CU.
// ======================================================================================================
constexpr int kFsStage = (128 + 256) * 16;         // floats per stage: X [128][16] then W [<= 256][16]
constexpr int kFsWin = 2 * kFsStage;               // float offset of the four wave-private 1024-float windows
constexpr int kFsBias = kFsWin + 4 * 1024;         // float offset of the two 256-float bias rows (layer parity)
constexpr int kFsFloats = kFsBias + 2 * 256;       // 16,896 floats = 67,584 B

// Pipelined K loop on PRE-LOADED first panels: panels 0 and 1 are resident in stages 0 / 1 on entry (weights by LDS-DMA,
// activations by LDS-DMA or straight from the previous epilogue), everything else is kloop_pipelined (mofa_layer.h) — same stages,
// same MFMA order.  The tail's first half panel additionally carries the requests of the NEXT layer's panel 0 into stage 0 (free
// after the tail's barrier): NWR rounds of weights (4: 256 rows, 2: 128 rows, 0: none) and NXR rounds of activations (2 when
// the next layer's first source is not this layer's output — the skip layers — else 0).
// What the merged tail needs to write a layer's output (EPI = true).
struct FusedEpi {
    const float* bias_lds;            // this wave's 64 bias floats in LDS
    float* y;                         // the layer's output panels
    long long m_padded, m_first;
    int n_first;
    float* win;                       // wave-private 1024-float window
    float* pass0;                     // wave 0 (when the next layer reads this output first): the two stages' X regions, where output
    float* pass1;                     //   slices 0 / 1 are formed — they ARE the next layer's operand panels 0 / 1; else nullptr
    unsigned long long* mask_out;     // MASKW: this layer's bits in the mask-only tape
};

// EPI = true (the ordinary layers): the epilogue runs INSIDE the tail.  After the last panel's first half every wave has read its
// last fragments, so one barrier frees both stages; the requests for the next layer's panel 1 and bias row go out, and the last 32
// MFMAs are issued accumulator pair by accumulator pair (each accumulator still sees its k-updates in the same order: bit-identical)
// with the bias + ReLU + staging + stores of the pair BEFORE in their shadow — a timing-only build without any epilogue says the
// epilogue is worth 5.3 % of this kernel when it runs by itself after the loop (profiles/r04_ab_fused_epilogue.txt).
template <int NI, int NJ, int BN, int NWR, int NXR, bool EPI = false, class Pre = int, bool MASKW = false>
__device__ __forceinline__ void kloop_fused(const float* xb, const float* x2b, const float* wb, long long xstep, long long wstep, int k1p,
                                            int KT, float* smem, int tid, int wave, int lane, int xrow0, int wrow0,
                                            f32x16 (&acc)[NI][NJ], const float* nwb, const float* nxb, const FusedEpi* ep = nullptr,
                                            Pre pre = Pre()) {
    constexpr int BM = 128, STAGE = kFsStage, XR = BM / 64, WR = BN / 64;
    const int lr = lane & 31, g = lane >> 5, sw = (lane >> 2) & 3;
    unsigned toff = (unsigned)tid * 4u;
    asm volatile("" : "+v"(toff));     // per-lane offsets are formed per call (hoisted out of the tile loop they cost registers for the whole kernel)
    float* const lds_wave = smem + wave * 256;
    int pq = 2;                                  // panels 0 and 1 are resident
    xb = (k1p > 2 || !x2b) ? xb + 2 * xstep : x2b + (2 - k1p) * xstep;
    wb += 2 * wstep;

    struct Frag {
        f32x4 a[NI], b[NJ];
    };
    // Byte offsets of this lane's 16 B inside round r of a panel, ONE VGPR each and opaque to the optimiser, and the weight-panel pointer kept
    // in SGPRs across its per-panel step: every LDS-DMA request of the loop is then `global_load_lds_dwordx4 v_off, s[base]`.  Left to itself
    // hipcc forms the addresses with 64-bit vector adds inside the MFMA stream (16 `v_lshl_add_u64` per 128 MFMAs here), and a vector
    // instruction in the shadow of an MFMA takes 6-12 cycles of the matrix pipe on this part (profiles/r06_probe_dual_issue.md): this kernel
    // 0.894 -> 0.922 of the peak (profiles/r06_ab_kloop_addr.md).
    unsigned foff[WR > XR ? WR : XR];
#pragma unroll
    for (int r = 0; r < (WR > XR ? WR : XR); ++r) {
        foff[r] = ((unsigned)r * 1024u + (unsigned)tid * 4u) * 4u;
        asm volatile("" : "+v"(foff[r]));
    }
    auto request = [&](int stage) {
        float* xs = lds_wave + stage * STAGE;
        float* ws = xs + BM * 16;
#pragma unroll
        for (int r = 0; r < (WR > XR ? WR : XR); ++r) asm volatile("" : "+v"(foff[r]));   // (per request: a zero-extension hoisted out of the
                                                                                           //  loop would hide the 32-bit offset from isel)
#pragma unroll
        for (int r = 0; r < XR; ++r) glds16((const float*)((const char*)xb + foff[r]), xs + r * 1024);
#pragma unroll
        for (int r = 0; r < WR; ++r) glds16((const float*)((const char*)wb + foff[r]), ws + r * 1024);
        ++pq;
        wb += wstep;
        asm volatile("" : "+s"(wb));              // (a scalar add per panel instead of base + step + offset on the vector pipe per request)
        xb = pq == k1p ? x2b : xb + xstep;
    };
    auto request_next = [&]() {                  // the next layer's panel 0 into stage 0
        float* xs = lds_wave;
        float* ws = xs + BM * 16;
#pragma unroll
        for (int r = 0; r < NXR; ++r) glds16(nxb + (r * 1024u + toff), xs + r * 1024);
#pragma unroll
        for (int r = 0; r < NWR; ++r) glds16(nwb + (r * 1024u + toff), ws + r * 1024);
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
    auto mfma_half = [&](const Frag& f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };
    // the layer's FIRST half panel starts the accumulators: the e = 0 products take a constant-zero C operand (an inline constant of the
    // MFMA instruction) instead of 128 registers zeroed by 128 VALU moves — VALU issue is time the matrix pipe does not get on this chip
    auto mfma_half_init = [&](const Frag& f) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][0], f.b[j][0], zero, 0, 0, 0);
#pragma unroll
        for (int e = 1; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };
    auto half_a = [&](int stage, Frag& cur, Frag& nxt, auto init_c) {
        __builtin_amdgcn_sched_barrier(0);
        read(stage, 1, nxt);
        if constexpr (decltype(init_c)::value) mfma_half_init(cur);
        else mfma_half(cur);
#pragma unroll
        for (int q = 0; q < NI + NJ; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI * NJ, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync_point = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // kind 0: no requests; 1: this layer's panel pq; 2: the next layer's panel 0
    auto half_b = [&](int stage, auto kind_c, bool do_read, Frag& cur, Frag& nxt) {
        constexpr int kind = decltype(kind_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if (do_read) read(stage ^ 1, 0, nxt);
        if constexpr (kind == 1) request(stage);
        if constexpr (kind == 2) request_next();
        mfma_half(cur);
        if (do_read) {
#pragma unroll
            for (int q = 0; q < NI + NJ; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        constexpr int nreq = kind == 1 ? XR + WR : (kind == 2 ? NXR + NWR : 0);
        if constexpr (nreq > 0) {
            constexpr int gap = (4 * NI * NJ - (NI + NJ)) / nreq;
            static_assert(gap >= 1, "the half panel has too few MFMAs to carry its memory instructions");
#pragma unroll
            for (int q = 0; q < nreq; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, gap, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI * NJ, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // (the loop is rotated by one half panel against kloop_pipelined so that the peeled first half is the initialising one; the
    //  sequence of halves, barriers and requests is the same)
    Frag fa, fb;
    constexpr std::false_type kAcc{};
    read(0, 0, fa);
    half_a(0, fa, fb, std::true_type{});
    for (int kt = 0; kt + 2 < KT; kt += 2) {
        sync_point();
        half_b(0, std::integral_constant<int, 1>{}, true, fb, fa);
        half_a(1, fa, fb, kAcc);
        sync_point();
        half_b(1, std::integral_constant<int, 1>{}, true, fb, fa);
        half_a(0, fa, fb, kAcc);
    }
    sync_point();
    half_b(0, std::integral_constant<int, ((NWR + NXR) > 0 ? 2 : 0)>{}, true, fb, fa);
    half_a(1, fa, fb, kAcc);
    if constexpr (!EPI) {
        half_b(1, std::integral_constant<int, 0>{}, false, fb, fa);
    } else {
        static_assert(NI == 2 && NJ == 4, "merged tail: 64 features x 128 points per wave");
        // every wave holds its last fragments in registers: both stages are free after this barrier (and the next layer's panel 0,
        // requested half a panel ago, has landed)
        sync_point();
        pre();                                   // the next layer's panel 1 + bias row: LDS-DMA requests under the MFMAs below
        const int wg = lane >> 5, msw = ((lane & 31) >> 2) & 3;
        int wo0 = (lane & 31) * 16 + (((0 + wg) ^ msw) << 2), wo1 = (lane & 31) * 16 + (((2 + wg) ^ msw) << 2), ro = lane * 4;
        asm volatile("" : "+v"(wo0), "+v"(wo1), "+v"(ro));
        // the wave's bias quads, fetched (LDS) before the MFMAs they hide under: the epilogue below is a chain of LDS round trips, and
        // every wait in it would stall the in-order MFMA issue behind it
        f32x4 bv[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[i][q] = *(const f32x4*)(ep->bias_lds + 4 * wg + 32 * i + 8 * q);
        // output slices of accumulator pair G = (i, jh): features 32 i + 16 qh .. + 15 (qh = 0, 1), points 64 jh .. + 63
        auto slices = [&](int i, int jh) {
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
                float* __restrict__ panel = ep->y + ((long long)((ep->n_first >> 4) + 2 * i + qh) * ep->m_padded + ep->m_first) * 16 + jh * 1024;
                float* const pass = (i == 0) ? (qh == 0 ? ep->pass0 : ep->pass1) : nullptr;          // wave-uniform
                float* const w = pass ? pass + jh * 1024 : ep->win;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int j = 2 * jh + jj, q = 2 * qh + qq;
                        f32x4 v;
                        v.x = acc[i][j][4 * q + 0] + bv[i][q].x, v.y = acc[i][j][4 * q + 1] + bv[i][q].y;
                        v.z = acc[i][j][4 * q + 2] + bv[i][q].z, v.w = acc[i][j][4 * q + 3] + bv[i][q].w;
                        v.x = relu_np(v.x), v.y = relu_np(v.y), v.z = relu_np(v.z), v.w = relu_np(v.w);
                        *(f32x4*)(w + 32 * jj * 16 + (qq ? wo1 : wo0)) = v;
                    }
                f32x4 r[4];                      // all four read-backs in flight together, then the four stores
#pragma unroll
                for (int it = 0; it < 4; ++it) r[it] = *(const f32x4*)(w + it * 256 + ro);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    *(f32x4*)(panel + it * 256 + ro) = r[it];
                }
                if constexpr (MASKW) mask_store_blocks4(ep->mask_out, panel - ep->y, lane, r);
            }
        };
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            const int i = G >> 1, jh = G & 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[i][2 * jh + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.a[i][e], fb.b[2 * jh + jj][e], acc[i][2 * jh + jj], 0, 0, 0);
            if (G > 0) slices((G - 1) >> 1, (G - 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        slices(1, 1);
    }
}

// The contiguous-store epilogue of mofa_layer.h (same values, same stores) with two additions: the bias row may come from LDS
// (LDSBIAS), and the slices (i = 0, qh = 0 / 1) — features [0,16) and [16,32) of the wave's 64 — may be formed in the caller's
// regions pass0 / pass1 instead of the wave's window: for wave 0 those are the next layer's operand panels 0 and 1.
template <int NI, int NJ, bool LDSBIAS, bool MASKW>
__device__ __forceinline__ void store_tile_fused(const f32x16 (&acc)[NI][NJ], const float* __restrict__ bias, float* __restrict__ y,
                                                 long long m_padded, long long m_first, int n_first, int lane, float* win, float* pass0,
                                                 float* pass1, unsigned long long* __restrict__ mask_out) {
    static_assert(NJ % 2 == 0, "row halves of 64 points");
    const int lr = lane & 31, g = lane >> 5, msw = (lr >> 2) & 3;
    // per-lane LDS offsets, formed per call: hoisted out of the layer / tile loops (they are invariant) the address of every
    // (window, fragment) pair would live in a register for the whole kernel
    int wo0 = lr * 16 + (((0 + g) ^ msw) << 2), wo1 = lr * 16 + (((2 + g) ^ msw) << 2), ro = lane * 4;
    asm volatile("" : "+v"(wo0), "+v"(wo1), "+v"(ro));
    f32x4 bv[NI][4];
    if constexpr (LDSBIAS) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[i][q] = *(const f32x4*)(bias + 4 * g + 32 * i + 8 * q);   // `bias`: this wave's 64 floats in LDS
    } else {
        bias_fetch<NI>(bias, n_first, lane, bv);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
            float* __restrict__ panel = y + ((long long)((n_first >> 4) + 2 * i + qh) * m_padded + m_first) * 16;
            float* const pass = (i == 0) ? (qh == 0 ? pass0 : pass1) : nullptr;      // wave-uniform
#pragma unroll
            for (int jh = 0; jh < NJ / 2; ++jh) {
                float* const w = pass ? pass + jh * 1024 : win;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int j = 2 * jh + jj, q = 2 * qh + qq;
                        f32x4 v;
                        v.x = acc[i][j][4 * q + 0] + bv[i][q].x, v.y = acc[i][j][4 * q + 1] + bv[i][q].y;
                        v.z = acc[i][j][4 * q + 2] + bv[i][q].z, v.w = acc[i][j][4 * q + 3] + bv[i][q].w;
                        v.x = relu_np(v.x), v.y = relu_np(v.y), v.z = relu_np(v.z), v.w = relu_np(v.w);
                        *(f32x4*)(w + 32 * jj * 16 + (qq ? wo1 : wo0)) = v;
                    }
                f32x4 r[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    r[it] = *(const f32x4*)(w + it * 256 + ro);
                    *(f32x4*)(panel + jh * 1024 + it * 256 + ro) = r[it];
                }
                if constexpr (MASKW) mask_store_blocks4(mask_out, (panel - y) + jh * 1024, lane, r);
            }
        }
}

template <bool MASKW>      // MASKW: also leave (y > 0) of every layer as bits (the fitting forward's mask-only tape)
__global__ __launch_bounds__(256, 2) void k_mlp_fused(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 128, NI = 2, STAGE = kFsStage;
    const int tid0 = threadIdx.x;
    const int wn = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int half_tiles = a.m_tiles * 2;
    const int nl = a.n_layers;
    float* const win = smem + kFsWin + wn * 1024;
    float* const lds_wave = smem + wn * 256;

    // everything of layer `nx` that can be requested BEFORE this layer's epilogue, once both stages are free: its weight panel 1
    // (panel 0 went out in the K loop's tail; `both`: layer 0 has no such tail, so panel 0 goes here too), its activation panels when
    // they do not come from this layer's output, and its bias row (ordinary layers; wave 0)
    // Per-lane indices are re-derived from an OPAQUE copy of the thread id inside every block below: hoisted out of the tile loop
    // (they are invariant) they would occupy registers — or spill slots — for the whole kernel next to 128 accumulators.
    auto thread_id = [&]() {
        int t = tid0;
        asm volatile("" : "+v"(t));
        return t;
    };
    auto prefetch = [&](const FusedLayer& nx, int nxi, bool both) {
        unsigned toff = (unsigned)thread_id() * 4u;
        const int wr = nx.n_padded >> 6;                                  // 4 (256 rows) or 2 (the 128-row view layer): wave-uniform
        const float* wsrc = a.packed + nx.w_off;
        for (int p = both ? 0 : 1; p < 2; ++p) {
            float* xs = lds_wave + p * STAGE;
            float* ws = xs + TM * 16;
            const float* wp = wsrc + (long long)p * nx.n_padded * 16;
            for (int r = 0; r < wr; ++r) glds16(wp + (r * 1024u + toff), ws + r * 1024);
        }
        if (wn == 0 && !nx.bias_row_div) glds16(a.folded + nx.bias_off + toff, smem + kFsBias + (nxi & 1) * 256);   // (wave 0: toff = lane * 4)
    };

    for (int ht = blockIdx.x; ht < half_tiles; ht += gridDim.x) {
        const long long m0 = (long long)ht * TM;
        // ---------------- layer 0: operand generated from the point (positional encoding), plain double-buffered loop ----------------
        {
            const int tid = thread_id(), lane = tid & 63;
            float px = 0.f, py = 0.f, pz = 0.f;
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
            const FusedLayer& l = a.L[0];
            const float* wbase = a.packed + l.w_off;
            auto stage_issue = [&](int buf, int kt) {
                float* xs = smem + buf * STAGE;
                float* ws = xs + TM * 16;
                const int row = tid & (TM - 1), k0 = (tid >> 7) * 8;
                const int swz = (row >> 2) & 3;
#pragma unroll 1
                for (int kk = k0; kk < k0 + 8; ++kk) {
                    const float v = pe_feature(kt * 16 + kk, px, py, pz, a.pe_feats);
                    xs[row * 16 + ((((kk >> 2) & 3) ^ swz) << 2) + (kk & 3)] = v;
                }
                const float* wsrc = wbase + (long long)kt * 256 * 16;
                const unsigned toff = (unsigned)tid * 4u;
#pragma unroll
                for (int r = 0; r < 4; ++r) glds16(wsrc + (r * 1024u + toff), ws + (r * 256 + wn * 64) * 4);
            };
            f32x16 acc[NI][4];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            stage_issue(0, 0);
            __syncthreads();
            for (int kt = 0; kt < l.k1p; ++kt) {
                const int cur = kt & 1;
                if (kt + 1 < l.k1p) stage_issue(cur ^ 1, kt + 1);
                const float* xs = smem + cur * STAGE;
                mma_panel<NI, 4>(xs, xs + TM * 16, 0, wn * 64, lane, acc);
                __syncthreads();
            }
            prefetch(a.L[1], 1, true);
            store_tile_fused<NI, 4, false, MASKW>(acc, a.folded + l.bias_off, const_cast<float*>(a.arena) + l.y_off, a.m_padded, m0, wn * 64, lane, win,
                                                  wn == 0 ? smem : nullptr, wn == 0 ? smem + STAGE : nullptr, MASKW ? a.mask_bits + l.mask_off : nullptr);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // layer 1's panels 0 / 1 (and this tile's first stores): once per tile
            __builtin_amdgcn_s_barrier();
        }
        // ---------------- the ordinary layers: 256 features, resident first panels, boundary work before the epilogue ----------------
        for (int li = 1; li < nl - 1; ++li) {
            const FusedLayer& l = a.L[li];
            const FusedLayer& nx = a.L[li + 1];
            const int tid = thread_id(), lane = tid & 63;
            const int KT = l.k1p + l.k2p;
            f32x16 acc[NI][4];                                              // (started by the loop's first half panel: no zeroing here)
            const float* xb = a.arena + l.x1_off + m0 * 16;
            const float* x2b = l.k2p ? a.arena + l.x2_off + m0 * 16 : nullptr;
            const float* wb = a.packed + l.w_off;
            const float* nwb = a.packed + nx.w_off;
            // every layer's first source is the previous layer's output (the skip layers contract [h | x], make_plan), so the next
            // layer's operand panels 0 / 1 always come straight from this epilogue: wave 0 forms them in the stages' X regions
            const bool pass = wn == 0;
            const FusedEpi ep{smem + kFsBias + (li & 1) * 256 + wn * 64, const_cast<float*>(a.arena) + l.y_off, a.m_padded, m0, wn * 64, win,
                              pass ? smem : nullptr, pass ? smem + STAGE : nullptr, MASKW ? a.mask_bits + l.mask_off : nullptr};
            auto pre = [&]() { prefetch(nx, li + 1, false); };
            // ONE instantiation for every ordinary layer (three would meet in register copies of the 128 accumulators): the tail always
            // requests four 1 KiB rounds of the next layer's weight panel 0 — for the 128-row view layer the upper two land in rows the
            // view layer never reads (they are its panel 1, valid memory)
            kloop_fused<NI, 4, 256, 4, 0, true, decltype(pre), MASKW>(xb, x2b, wb, a.m_padded * 16, 256 * 16, l.k1p, KT, smem, tid, wn, lane, 0, wn * 64, acc, nwb,
                                                                      nullptr, &ep, pre);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // my LDS writes (the handed-over panels) are done; the stores drain
            __builtin_amdgcn_s_barrier();                                 // behind the next loop's first half panel (its vmcnt(0) + barrier)
        }
        // ---------------- the view layer: 128 features x 128 points on 2 x 2 waves, per-ray bias rows ----------------
        {
            const FusedLayer& l = a.L[nl - 1];
            const int tid = thread_id(), lane = tid & 63;
            f32x16 acc[NI][2];
            const int wn2 = wn & 1, wm2 = wn >> 1;
            kloop_fused<NI, 2, 128, 0, 0>(a.arena + l.x1_off + m0 * 16, nullptr, a.packed + l.w_off, a.m_padded * 16, 128 * 16, l.k1p, l.k1p, smem,
                                          tid, wn, lane, wm2 * 64, wn2 * 64, acc, nullptr, nullptr);
            f32x4 bv[NI][4];
            store_tile<NI, 2, true>(acc, a.view_bias_rows + l.bias_off, a.bias_rows, l.bias_row_div, 128, const_cast<float*>(a.arena) + l.y_off, a.m_padded,
                                    m0 + wm2 * 64, wn2 * 64, 1, lane, bv);
            __threadfence_block();
            __syncthreads();          // the stages are free for the next tile's first layer
        }
    }
}

// can this launch take the pipelined persistent kernel?  (the shape the north star names; everything else runs the generic one)
bool fused_fast_shape(const FusedArgs& a) {
    if (!a.pipe || a.n_layers < 3) return false;
    const FusedLayer& f = a.L[0];
    if (f.x1_off >= 0 || f.n_padded != 256 || f.k2p != 0 || f.bias_row_div != 0) return false;
    for (int i = 1; i < a.n_layers - 1; ++i) {
        const FusedLayer& l = a.L[i];
        if (l.x1_off != a.L[i - 1].y_off || l.n_padded != 256 || l.k1p != 16 || (l.k2p != 0 && l.k2p != 16) || l.bias_row_div != 0) return false;
    }
    const FusedLayer& v = a.L[a.n_layers - 1];
    return v.x1_off >= 0 && v.n_padded == 128 && v.k1p == 16 && v.k2p == 0 && v.bias_row_div != 0 && v.x1_off == a.L[a.n_layers - 2].y_off;
}
std::atomic<int> g_fused_attr[kMaxDevices];
int set_fused_attributes(int lds) {
    if (hipFuncSetAttribute((const void*)k_mlp_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mlp_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return check_launch("hipFuncSetAttribute(k_mlp_fused)");
    return MOFA_OK;
}

int launch_fused(const FusedArgs& a, hipStream_t st) {
    const int dev = current_device();
    const int cus = compute_units(dev);
    const int half_tiles = a.m_tiles * 2;
    const int grid = half_tiles < 2 * cus ? half_tiles : 2 * cus;      // two resident workgroups per CU
    if (fused_fast_shape(a)) {
        const size_t lds = (size_t)kFsFloats * sizeof(float);          // 66 KiB: above the 64 KiB default limit of dynamic LDS
        if (!g_fused_attr[dev].load(std::memory_order_acquire)) {      // one-time function attribute per device
            const int rc = set_fused_attributes((int)lds);
            if (rc != MOFA_OK) return rc;
            g_fused_attr[dev].store(1, std::memory_order_release);
        }
        if (a.mask_bits) hipLaunchKernelGGL(k_mlp_fused<true>, dim3(grid), dim3(256), lds, st, a);      // the fitting forward: + mask bits
        else hipLaunchKernelGGL(k_mlp_fused<false>, dim3(grid), dim3(256), lds, st, a);
        return check_launch("k_mlp_fused");
    }
    const size_t lds = 2 * (size_t)(128 + 256) * 16 * sizeof(float);   // 48 KiB
    if (a.mask_bits) hipLaunchKernelGGL(k_mlp_fused_generic<true>, dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(k_mlp_fused_generic<false>, dim3(grid), dim3(256), lds, st, a);
    return check_launch("k_mlp_fused_generic");
}


// ======================================================================================================
// k_net_chain: the GEMMs of a WIDE network (widths > 256: every layer is a grid of 256-point x 128-feature tiles of the layer kernel)
// as ONE launch — the forward layers of a sub-batch (inference, or keeping a tape: fp32 or mask bits), or the backward-data products
// of a fitting step.  A per-layer launch ends when its slowest CU has finished its last tile — at 12 tile rounds per launch the fill
// and the drain cost 2.5 % of every one of the ~26 launches of a sub-batch (profiles/r02_timeline_*.md; a 4x larger netchunk measures
// 0.949 instead of 0.938 for exactly that reason).  Here the tiles of ALL steps form one queue per XCD and the only thing a tile waits
// for is what it really depends on: the tiles of the earlier steps over ITS OWN 256 point rows (every step is row-wise: Linear + bias
// + ReLU or dX = G W^T with a row-aligned mask, concatenations / accumulations of row-aligned tensors).  So a CU that is done with
// step l starts step l + 1 on rows whose inputs are complete while other CUs still finish step l, and the launch drains once per
// sub-batch instead of once per layer.
//
//   * Queue: workgroups are persistent (two per CU); each pulls tile numbers from the head counter of the XCD it RUNS on
//     (`s_getreg HW_REG_XCC_ID`, not an assumed blockIdx -> XCD map; mofa_device_init() takes a census per device that all eight XCDs
//     get workgroups, else the per-layer launches run).  XCD x owns the point-row tiles [x m/8, (x+1) m/8) of every step, in
//     step-major order — the driver lists the steps in a topological order, pulled in increasing order, so a tile only ever waits for
//     tiles that RUNNING workgroups hold: no deadlock for any residency.  The next tile number is drawn one tile ahead.
//   * Dependency: one counter per point-row tile, incremented once per finished tile of any step; tile (step s, rows m, *) waits
//     until done[m] has reached the number of tiles the steps before s have over those rows.  Conservative for a step list that is
//     not a chain (the backward's skip branches), and the same rule covers the buffers the plan recycles or accumulates into (a step
//     overwrites rows only after every earlier reader / writer of those rows is complete).
//   * Visibility: producer and consumer of a row tile are by construction on the same XCD, whose L2 is their point of coherence:
//     the producer's plain stores are acknowledged by that L2 (`s_waitcnt vmcnt(0)` in every wave, then a barrier) before one lane
//     bumps the counter; the consumer polls the counter with relaxed agent-scope loads and requests its operand panels with
//     `sc1` LDS-DMA loads, which the L2 serves and this CU's vector L1 (never refreshed by other CUs' stores) cannot; the backward's
//     accumulate epilogue reads the running sum with agent-scope loads for the same reason.  Weights, biases, masks and the per-ray
//     bias rows were written by earlier launches and use the default policy.
//   * The tile itself is k_layer<128, .., PIPE>'s: same panels, same K loop, same epilogues — bit-identical to per-layer launches.
//   * Failure is LOUD.  A dependency wait that exceeds its poll budget (seconds) — or that sees another
//     workgroup's time-out flag — sets status bit 0 and the workgroup ABANDONS the launch: it computes nothing further (never a tile
//     on incomplete inputs), so the launch ends with fewer finished tiles than it has.  An XCD that received no workgroups (a CU-masked
//     stream, a partition change after the census) leaves its queue unworked with the same result.  k_chain_verify, enqueued behind
//     every chained launch (after the heads), compares {time-out flag, finished tiles} with what the launch must have produced and on
//     any difference overwrites the launch's outputs with NaN — the image / the gradients cannot look plausible — and raises the
//     caller's sticky verdict words, which the host layer reads back asynchronously and turns into MofaError.
// State (zeroed by a memset node ahead of every launch): 8 heads at a 128-byte stride, two status words, the per-row-tile counters.
// ======================================================================================================
constexpr int kMaxChainSteps = MOFA_MAX_CHAIN_STEPS;
constexpr int kChainHeadStride = 32;                 // unsigned words between two XCDs' queue heads (one 128-byte line each)
constexpr int kChainStatus = 8 * kChainHeadStride;   // [0] bit 0: a dependency wait timed out; [1]: tiles finished (all XCDs)
constexpr int kChainDone = kChainStatus + 32;        // done[m_tiles]

struct ChainArgs {
    unsigned* state;
    long long m_padded, bias_rows;
    int m_tiles, n_steps, tiles_per_m;
    unsigned spin_limit;              // polls (each >= ~1 us with its s_sleep) before a wait gives up: seconds, never a hang
    int skip_xcd;                     // tests only (mofa_test_hooks): workgroups on this XCD leave at once — what a CU-masked stream
                                      // that starves an XCD looks like; -1 = none
    ChainStep S[kMaxChainSteps];
};
static_assert(sizeof(ChainArgs) <= 4096, "kernel arguments are limited to 4 KiB");

__global__ void k_xcc_census(unsigned* counts) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicAdd(counts + (xcc & 7u), 1u);
}

// Behind every chained launch: did every tile of every step run, with no wait abandoned?  If not, the outputs of the launch are
// overwritten with NaN (up to four buffers) and the sticky verdict words of the caller are raised:
//   verdict[0] |= 1 (a wait timed out) | 2 (tiles missing);  [1] += 1 (launches verified);  [2], [3], [4] = this launch's flags,
//   finished tiles, expected tiles;  [5] += 1 per bad launch.
__global__ __launch_bounds__(256) void k_chain_verify(const unsigned* __restrict__ status, unsigned expect, unsigned* __restrict__ verdict,
                                                      float* p0, long long n0, float* p1, long long n1, float* p2, long long n2, float* p3, long long n3) {
    const unsigned flags = status[0], finished = status[1];
    const bool bad = flags != 0u || finished != expect;
    if (verdict && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(verdict + 1, 1u);                      // (atomics: two streams may verify launches of one network at the same time)
        verdict[2] = flags, verdict[3] = finished, verdict[4] = expect;
        if (bad) atomicOr(verdict, (flags ? 1u : 0u) | (finished != expect ? 2u : 0u)), atomicAdd(verdict + 5, 1u);
    }
    if (!bad) return;
    const float nan = __builtin_nanf("");
    const long long stride = (long long)gridDim.x * 256, t0 = (long long)blockIdx.x * 256 + threadIdx.x;
    for (long long i = t0; i < n0; i += stride) p0[i] = nan;
    for (long long i = t0; i < n1; i += stride) p1[i] = nan;
    for (long long i = t0; i < n2; i += stride) p2[i] = nan;
    for (long long i = t0; i < n3; i += stride) p3[i] = nan;
}

// The same look at a chained launch's status for MANY output buffers (the training backward's weight gradients: 2D + 7 tensors the
// second-stage sums were written into): an incomplete launch turns every one of them into NaN too (the verdict words are k_chain_verify's).
constexpr int kMaxPoison = 80;
struct PoisonArgs {
    float* p[kMaxPoison];
    long long n[kMaxPoison];
    int count;
};
__global__ __launch_bounds__(256) void k_chain_poison(const unsigned* __restrict__ status, unsigned expect, const PoisonArgs a) {
    if (status[0] == 0u && status[1] == expect) return;
    const float nan = __builtin_nanf("");
    const long long stride = (long long)gridDim.x * 256, t0 = (long long)blockIdx.x * 256 + threadIdx.x;
    for (int b = 0; b < a.count; ++b)
        for (long long i = t0; i < a.n[b]; i += stride) a.p[b][i] = nan;
}

// What a workgroup of k_net_chain carries from tile to tile.  It rides into the K loop as the policy's `Probe` (the hook the loop calls
// after every panel's wait + barrier), because two things belong BEHIND the next tile's first panel rather than between two tiles:
//   * the completion signal of the tile just finished — the loop's first `s_waitcnt vmcnt(0)` + barrier is the point where every
//     wave's stores of the PREVIOUS tile are known to be in the L2, so nobody waits for the store acknowledgements at the tile
//     boundary (2-4 us of a 234 us tile when it was done there);
//   * the look at the NEXT tile's dependency counter — the queue ticket drawn at the top of this tile has returned by then, and the
//     counter's value travels while this tile computes; the poll only spins (at the next tile's top) in the rare case it was too early.
struct ChainPolicy : ShippedPolicy {
    struct Probe {
        const ChainArgs& a;
        unsigned* done;
        unsigned* status;
        int m_lo, m_cnt, total, tid;
        int prev_mt = -1;        // row tile whose completion is still to be signalled
        bool first = false;      // the current tile's first panel has not been passed yet
        int s_next = 0;          // (thread 0) step pointer of the next tile
        int qn = 0;              // (thread 0) next tile number
        unsigned seen = 0, need = 0;   // (thread 0) the next tile's counter as seen early / what it must reach
        __device__ __forceinline__ Probe(const ChainArgs& a_, unsigned* done_, unsigned* status_, int m_lo_, int m_cnt_, int total_, int tid_)
            : a(a_), done(done_), status(status_), m_lo(m_lo_), m_cnt(m_cnt_), total(total_), tid(tid_) {}
        __device__ __forceinline__ void signal_prev() {       // (after a vmcnt(0) + barrier that covers the previous tile's stores)
            if (prev_mt >= 0 && tid == 0) {
                __hip_atomic_fetch_add(done + prev_mt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            prev_mt = -1;
        }
        __device__ __forceinline__ void entry() {}
        __device__ __forceinline__ void kloop_begin() {}
        __device__ __forceinline__ void first_panel_landed() {}
        __device__ __forceinline__ void kloop_end() {}
        __device__ __forceinline__ void stores_issued() {}
        __device__ __forceinline__ void panel() {
            if (!first) return;
            first = false;
            signal_prev();
            if (tid == 0 && qn < total) {
                while (qn >= m_cnt * (a.S[s_next].tiles_before + a.S[s_next].n_tiles)) ++s_next;
                const int r = qn - m_cnt * a.S[s_next].tiles_before;
                need = (unsigned)a.S[s_next].tiles_before;
                seen = __hip_atomic_load(done + m_lo + r / a.S[s_next].n_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
};

// Backward-data epilogue of a chained step whose running sum (ACC) was written by ANOTHER workgroup of the same launch: the same
// arithmetic as store_tile_staged_bwd, with the old value read at agent scope (served by the XCD's L2, see "Visibility").
template <int NI, int NJ, int MASK>
__device__ __forceinline__ void chain_store_bwd_acc(const f32x16 (&acc)[NI][NJ], float* __restrict__ y, const float* __restrict__ mask,
                                                    long long m_padded, long long m_first, int n_first, int lane, float* win,
                                                    const unsigned long long* __restrict__ mask_bits) {
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
                unsigned long long mb[4][4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const float* o = y + off + it * 256;
                    old[it].x = __hip_atomic_load(o + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    old[it].y = __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    old[it].z = __hip_atomic_load(o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    old[it].w = __hip_atomic_load(o + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if constexpr (MASK == 1) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) act[it] = *(const f32x4*)(mask + off + it * 256);
                }
                if constexpr (MASK == 2) {
                    const unsigned long long* w = mask_bits + ((poff + jh * 1024) >> 8) * 4;
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int c = 0; c < 4; ++c) mb[it][c] = w[it * 4 + c];
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
                    v.x += old[it].x, v.y += old[it].y, v.z += old[it].z, v.w += old[it].w;
                    if constexpr (MASK == 1)
                        v.x = act[it].x > 0.f ? v.x : 0.f, v.y = act[it].y > 0.f ? v.y : 0.f, v.z = act[it].z > 0.f ? v.z : 0.f,
                        v.w = act[it].w > 0.f ? v.w : 0.f;
                    if constexpr (MASK == 2)
                        v.x = ((mb[it][0] >> lane) & 1ull) ? v.x : 0.f, v.y = ((mb[it][1] >> lane) & 1ull) ? v.y : 0.f,
                        v.z = ((mb[it][2] >> lane) & 1ull) ? v.z : 0.f, v.w = ((mb[it][3] >> lane) & 1ull) ? v.w : 0.f;
                    *(f32x4*)(y + off + it * 256) = v;
                }
            }
        }
}

// MODE: kChainForward (inference and fp32-tape forwards: the same epilogues, the outputs simply are tape slots), kChainForwardMask (the
// contiguous-store epilogue also writes (y > 0) as bits), kChainBackward (backward-data epilogues).
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_net_chain(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = kRowTile, BN = 128, NI = 2, NJ = 4, STAGE = (BM + BN) * 16;
    int* const slot = (int*)(smem + 2 * STAGE);       // [0] next tile number, [1] "its inputs are known to be complete", [2] "the wait ended well": thread 0 -> workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;

    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if ((int)xcc == a.skip_xcd) return;
    const int mpx = (a.m_tiles + 7) >> 3;
    const int m_lo = (int)xcc * mpx;
    const int m_cnt = m_lo < a.m_tiles ? (a.m_tiles - m_lo < mpx ? a.m_tiles - m_lo : mpx) : 0;
    const int total = m_cnt * a.tiles_per_m;
    unsigned* const head = a.state + xcc * kChainHeadStride;
    unsigned* const status = a.state + kChainStatus;
    unsigned* const done = a.state + kChainDone;
    ChainPolicy::Probe hook(a, done, status, m_lo, m_cnt, total, tid);

    if (tid == 0) slot[0] = (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), slot[1] = 0;
    __syncthreads();
    int q = slot[0], ready = 0;
    int s = 0;
    while (q < total) {
        while (q >= m_cnt * (a.S[s].tiles_before + a.S[s].n_tiles)) ++s;          // tile numbers only grow: the step pointer only advances
        const ChainStep& st = a.S[s];
        const int r = q - m_cnt * st.tiles_before;
        const int mt = m_lo + r / st.n_tiles, nt = r - (r / st.n_tiles) * st.n_tiles;
        if (!ready) {
            // the early look did not find this tile's inputs complete (or there was none: the first tile).  The previous tile's signal
            // must go out BEFORE waiting — this tile may depend on it — so its stores are waited for here instead of behind the first panel
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            hook.signal_prev();
            if (tid == 0) {
                const unsigned need = (unsigned)st.tiles_before;
                unsigned spins = 0;
                int ok = 1;
                while (__hip_atomic_load(done + mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(8);
                    ++spins;
                    // never a hang, never a tile on incomplete inputs: out of budget — or somebody else already is, so the producer this
                    // tile waits for may never come — flag it and abandon the launch (k_chain_verify turns that into NaN + a verdict)
                    if (spins >= a.spin_limit || ((spins & 255u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
                slot[2] = ok;
            }
            __syncthreads();                             // the inputs of this tile are complete — for every wave
            if (!slot[2]) break;
        }
        if (tid == 0) hook.qn = (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the ticket after this one
        hook.first = true;

        const long long m0 = (long long)mt * BM;
        const int n0 = nt * BN;
        f32x16 acc[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        kloop_pipelined<NI, NJ, BM, BN, ChainPolicy, 16>(st.x1 + m0 * 16, st.k2p ? st.x2 + m0 * 16 : nullptr, st.w + (long long)n0 * 16,
                                                         a.m_padded * 16, (long long)st.n_padded * 16, st.k1p, st.k1p + st.k2p, smem, tid, wave,
                                                         lane, wm * (32 * NJ), wn * 64, acc, hook);
        float* const y = st.y;
        const long long mf = m0 + wm * (32 * NJ);
        const int nf = n0 + wn * 64;
        float* const win = smem + wave * 1024;           // wave-private window inside stage 0 (free after the loop's last barrier)
        if constexpr (MODE == kChainBackward) {
            if (st.flags & 1) {                          // += the running sum another step of this launch left there
                if (st.aux) chain_store_bwd_acc<NI, NJ, 1>(acc, y, st.aux, a.m_padded, mf, nf, lane, win, nullptr);
                else if (st.bits) chain_store_bwd_acc<NI, NJ, 2>(acc, y, nullptr, a.m_padded, mf, nf, lane, win, st.bits);
                else chain_store_bwd_acc<NI, NJ, 0>(acc, y, nullptr, a.m_padded, mf, nf, lane, win, nullptr);
            } else {
                if (st.aux) store_tile_staged_bwd<NI, NJ, false, 1>(acc, y, st.aux, a.m_padded, mf, nf, lane, win);
                else if (st.bits) store_tile_staged_bwd<NI, NJ, false, 2>(acc, y, nullptr, a.m_padded, mf, nf, lane, win, st.bits);
                else store_tile_staged_bwd<NI, NJ, false, 0>(acc, y, nullptr, a.m_padded, mf, nf, lane, win);
            }
        } else if (st.bias_row_div) {                    // the view layer: per-ray bias rows (its mask bits, if any, come from a pass over y)
            f32x4 bv[NI][4];
            store_tile<NI, NJ, true>(acc, st.aux, a.bias_rows, st.bias_row_div, st.n_padded, y, a.m_padded, mf, nf, st.flags & 1, lane, bv);
        } else if constexpr (MODE == kChainForwardMask) {
            store_tile_staged<NI, NJ, true, true>(acc, st.aux, y, a.m_padded, mf, nf, lane, win, st.bits);
        } else {
            if (st.flags & 1) store_tile_staged<NI, NJ, true>(acc, st.aux, y, a.m_padded, mf, nf, lane, win);
            else store_tile_staged<NI, NJ, false>(acc, st.aux, y, a.m_padded, mf, nf, lane, win);
        }
        hook.prev_mt = mt;                               // signalled behind the next tile's first panel (or below / above when there is a wait)
        if (tid == 0) slot[0] = hook.qn, slot[1] = (hook.qn >= total || hook.seen >= hook.need) ? 1 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                 // the windows and stages are free for the next tile's requests
        q = slot[0], ready = slot[1];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last tile of this workgroup
    __syncthreads();
    hook.signal_prev();
}

// ======================================================================================================
// k_net_chain_train (round 6): the TRAINING backward of a wide network's sub-batch — backward-data products AND weight gradients — behind
// the same per-XCD queues.  Until round 5 this was the last per-layer path (two launches per layer and sub-batch, 65 % of a training
// step): every layer's gradient G feeds a weight-gradient GEMM dW = G^T X between two backward-data products, and G's buffer is
// recycled two products later.  Here the weight gradient's units are QUEUE ENTRIES too:
//   * a step is either a backward-data product (tiles of 256 rows x 128 features, exactly k_net_chain<backward>'s) or a weight gradient
//     (flags bit 1): its units are the [128 x 256] output tiles of k_wgrad over ONE SPLIT of the points — wg_split's plan (mofa_common.h):
//     whole row tiles, never straddling an XCD's row range, the same splits the per-layer kernel sums, so the partial sums are
//     bit-identical and the same deterministic second stage (k_wgrad_reduce, after the launch) finishes them;
//   * the dependency rule is unchanged — a tile of step s over rows R starts when every tile of the steps before s over R has finished —
//     with R a RANGE of row tiles for a weight-gradient unit: it waits for the counters of all its rows and bumps them all when done.
//     That one rule orders everything the per-layer stream order did: dW_l reads G_l after the product that wrote it; the product that
//     overwrites G_l's buffer waits for dW_l's units over those rows;
//   * queue order per XCD stays step-major (the driver lists the steps in the per-layer form's own order), so a unit only ever waits for
//     entries RUNNING workgroups hold: no deadlock for any residency; a weight-gradient unit is ~6 tiles' worth of work, drawn like any
//     other entry;
//   * G panels were written by other workgroups of the same launch: requested with sc1 (served by the XCD's L2), like the activation
//     panels of the products; X panels (the tape), masks and weights come from earlier launches;
//   * the same loud failure: bounded waits, abandon, k_chain_verify behind the launch.
// ======================================================================================================
struct TrainChainArgs {
    unsigned* state;
    long long m_padded, n_points;
    int m_tiles, n_steps;
    unsigned spin_limit;
    int skip_xcd;                     // tests only (mofa_test_hooks)
    int pipe;                         // weight-gradient units: software-pipelined chunk loop (MOFA_PIPE; the products always are)
    ChainStep S[kMaxChainSteps];
};
static_assert(sizeof(TrainChainArgs) <= 4096, "kernel arguments are limited to 4 KiB");

struct TrainChainPolicy : ShippedPolicy {
    // ChainPolicy::Probe generalised to entries that own a RANGE of row tiles: the signal bumps every counter of the finished entry's
    // rows, the early look (and the wait) is wave 0's — lane i takes counters i, i + 64, ... of the next entry's rows.
    struct Probe {
        const TrainChainArgs& a;
        unsigned* done;
        unsigned* status;
        int m_lo, m_cnt, total, tid;
        int prev_first = -1, prev_cnt = 0;   // rows whose completion is still to be signalled
        bool first = false;                  // the current entry's first panel / chunk has not been passed yet
        int s_next = 0, qlo_next = 0;        // (wave 0) the next entry's step and that step's first entry number
        int qn = 0;                          // (wave 0, uniform) next entry number
        unsigned seen = 0xffffffffu, need = 0;   // (wave 0) this lane's minimum over its share of the next entry's counters, seen early / what they must reach
        __device__ __forceinline__ Probe(const TrainChainArgs& a_, unsigned* done_, unsigned* status_, int m_lo_, int m_cnt_, int total_, int tid_)
            : a(a_), done(done_), status(status_), m_lo(m_lo_), m_cnt(m_cnt_), total(total_), tid(tid_) {}
        // entries of step s in this XCD's queue: a product owns the XCD's row tiles, a weight gradient its splits of them
        __device__ __forceinline__ int cnt(int s) const {
            const ChainStep& st = a.S[s];
            return st.n_tiles * ((st.flags & kChainStepWgrad) ? (m_cnt + st.spt - 1) / st.spt : m_cnt);
        }
        // rows [mf, mf + mc) of entry r of step st
        __device__ __forceinline__ void rows(const ChainStep& st, int r, int& mf, int& mc) const {
            if (st.flags & kChainStepWgrad) {
                const int j = r / st.n_tiles;
                mf = m_lo + j * st.spt;
                mc = m_cnt - j * st.spt < st.spt ? m_cnt - j * st.spt : st.spt;
            } else {
                mf = m_lo + r / st.n_tiles, mc = 1;
            }
        }
        __device__ __forceinline__ unsigned look(int mf, int mc) const {      // (wave 0) this lane's minimum over its share of the counters
            unsigned v = 0xffffffffu;
            for (int i = tid; i < mc; i += 64) {
                const unsigned c = __hip_atomic_load(done + mf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = c < v ? c : v;
            }
            return v;
        }
        __device__ __forceinline__ void signal_prev() {       // (after a vmcnt(0) + barrier that covers the previous entry's stores)
            if (prev_first >= 0 && tid < 64) {
                for (int i = tid; i < prev_cnt; i += 64) __hip_atomic_fetch_add(done + prev_first + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tid == 0) __hip_atomic_fetch_add(status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            prev_first = -1;
        }
        __device__ __forceinline__ void entry() {}
        __device__ __forceinline__ void kloop_begin() {}
        __device__ __forceinline__ void first_panel_landed() {}
        __device__ __forceinline__ void kloop_end() {}
        __device__ __forceinline__ void stores_issued() {}
        __device__ __forceinline__ void panel() {
            if (!first) return;
            first = false;
            signal_prev();
            if (tid < 64 && qn < total) {
                while (qn >= qlo_next + cnt(s_next)) qlo_next += cnt(s_next), ++s_next;
                int mf, mc;
                rows(a.S[s_next], qn - qlo_next, mf, mc);
                need = (unsigned)a.S[s_next].tiles_before;
                seen = look(mf, mc);
            }
        }
    };
};

__global__ __launch_bounds__(256, 2) void k_net_chain_train(const TrainChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = kRowTile, BN = 128, NI = 2, NJ = 4, STAGE = (BM + BN) * 16;
    constexpr int kWgLds = 2 * (128 / 16 + 256 / 16) * (WgCfg<128, 256>::MC * 16 + 16);     // floats of the weight-gradient unit's two stages (> 2 * STAGE)
    static_assert(kWgLds >= 2 * STAGE, "the slot words sit behind the larger of the two tile forms");
    int* const slot = (int*)(smem + kWgLds);          // [0] next entry number, [1] "its inputs are known to be complete", [2] "the wait ended well"
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;

    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if ((int)xcc == a.skip_xcd) return;
    const int mpx = (a.m_tiles + 7) >> 3;
    const int m_lo = (int)xcc * mpx;
    const int m_cnt = m_lo < a.m_tiles ? (a.m_tiles - m_lo < mpx ? a.m_tiles - m_lo : mpx) : 0;
    unsigned* const head = a.state + xcc * kChainHeadStride;
    unsigned* const status = a.state + kChainStatus;
    unsigned* const done = a.state + kChainDone;
    TrainChainPolicy::Probe hook(a, done, status, m_lo, m_cnt, 0, tid);
    int total = 0;
    for (int i = 0; i < a.n_steps; ++i) total += hook.cnt(i);
    hook.total = total;

    if (tid == 0) slot[0] = (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), slot[1] = 0;
    __syncthreads();
    // (entry numbers and everything derived from them are workgroup-uniform: say so, or they live in vector registers next to 128 accumulators)
    int q = __builtin_amdgcn_readfirstlane(slot[0]), ready = 0;
    int s = 0, qlo = 0;
    while (q < total) {
        while (q >= qlo + hook.cnt(s)) qlo += hook.cnt(s), ++s;          // entry numbers only grow: the step pointer only advances
        const ChainStep& st = a.S[s];
        const int r = q - qlo;
        int mf, mc;
        hook.rows(st, r, mf, mc);
        if (!ready) {
            // (as in k_net_chain: the previous entry's signal must go out BEFORE waiting — this entry may depend on it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            hook.signal_prev();
            if (tid < 64) {
                const unsigned need = (unsigned)st.tiles_before;
                unsigned spins = 0;
                int ok = 1;
                while (!__all(hook.look(mf, mc) >= need ? 1 : 0)) {
                    __builtin_amdgcn_s_sleep(8);
                    ++spins;
                    if (spins >= a.spin_limit || ((spins & 255u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        if (tid == 0) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
                if (tid == 0) slot[2] = ok;
            }
            __syncthreads();                             // the inputs of this entry are complete — for every wave
            if (!__builtin_amdgcn_readfirstlane(slot[2])) break;
        }
        if (tid < 64) {                                  // the ticket after this one (wave 0 keeps it uniform: its lanes share the early look)
            unsigned t = 0;
            if (tid == 0) t = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hook.qn = (int)__builtin_amdgcn_readfirstlane(t);
        }
        hook.first = true;

        if (st.flags & kChainStepWgrad) {
            // one [128 x 256] tile of dW over the points of split j: k_wgrad's unit (mofa_wgrad.h), G requested with sc1
            const int u = r - (r / st.n_tiles) * st.n_tiles, j = r / st.n_tiles;
            const int n_tiles_n = st.n_padded >> 7;
            const int nt = u % n_tiles_n, kt = u / n_tiles_n;
            const int k_padded = st.k1p * 16;
            const long long gs = (long long)xcc * ((mpx + st.spt - 1) / st.spt) + j;        // the split's global index (wg_split: nspx per full range)
            const long long total_chunks = (a.n_points + 15) / 16;
            const long long c_begin = (long long)mf * (kRowTile / 16);
            long long c_end = c_begin + (long long)mc * (kRowTile / 16);
            if (c_end > total_chunks) c_end = total_chunks;
            wgrad_unit<128, 256, TrainChainPolicy::Probe, 16>(st.x1, st.x2, a.m_padded, a.n_points, st.n_padded, k_padded, nt * 128, kt * 256, c_begin, c_end,
                                                              st.y + gs * st.n_padded * k_padded,
                                                              st.aux ? const_cast<float*>(st.aux) + gs * st.n_padded : nullptr, a.pipe, smem, hook);
        } else {
            const int nt = r - (r / st.n_tiles) * st.n_tiles;
            int lane_t = lane, tid_t = tid;              // (this form's lane constants are formed per tile, not kept across the other form's accumulators)
            asm volatile("" : "+v"(lane_t), "+v"(tid_t));
            const long long m0 = (long long)mf * BM;
            const int n0 = nt * BN;
            f32x16 acc[NI][NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
            kloop_pipelined<NI, NJ, BM, BN, TrainChainPolicy, 16>(st.x1 + m0 * 16, nullptr, st.w + (long long)n0 * 16, a.m_padded * 16,
                                                                  (long long)st.n_padded * 16, st.k1p, st.k1p, smem, tid_t, wave, lane_t, wm * (32 * NJ), wn * 64,
                                                                  acc, hook);
            float* const y = st.y;
            const long long mrow = m0 + wm * (32 * NJ);
            const int nf = n0 + wn * 64;
            float* const win = smem + wave * 1024;       // wave-private window inside stage 0 (free after the loop's last barrier)
            if (st.flags & 1) {                          // += the running sum another step of this launch left there
                if (st.aux) chain_store_bwd_acc<NI, NJ, 1>(acc, y, st.aux, a.m_padded, mrow, nf, lane_t, win, nullptr);
                else chain_store_bwd_acc<NI, NJ, 0>(acc, y, nullptr, a.m_padded, mrow, nf, lane_t, win, nullptr);
            } else {
                if (st.aux) store_tile_staged_bwd<NI, NJ, false, 1>(acc, y, st.aux, a.m_padded, mrow, nf, lane_t, win);
                else store_tile_staged_bwd<NI, NJ, false, 0>(acc, y, nullptr, a.m_padded, mrow, nf, lane_t, win);
            }
        }
        if (hook.first) {                                // (an entry without a single panel / chunk: cannot happen for the shapes the launcher admits)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            hook.panel();
        }
        hook.prev_first = mf, hook.prev_cnt = mc;        // signalled behind the next entry's first panel (or below / above when there is a wait)
        if (tid < 64) {
            const int ok = __all(hook.seen >= hook.need ? 1 : 0);
            if (tid == 0) slot[0] = hook.qn, slot[1] = (hook.qn >= total || ok) ? 1 : 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                 // the windows and stages are free for the next entry's requests
        q = __builtin_amdgcn_readfirstlane(slot[0]), ready = __builtin_amdgcn_readfirstlane(slot[1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last entry of this workgroup
    __syncthreads();
    hook.signal_prev();
}

// Optional per-launch timing of the dominant kernels with HIP events recorded on the launch stream; used by bench.py for the
// live roofline figure.  Off by default (no events, no overhead).  The measurement session is explicit state the HOST opens and
// closes (mofa_prof_begin/end); it is kept per device and guarded by a mutex, so two devices or two host threads in one process
// do not share or corrupt it.  When no session is open the launch paths only read one relaxed atomic.
constexpr int kProfKinds = MOFA_PROF_KINDS;   // 0: k_layer<128,..,PIPE> (forward), 1: k_mlp_fused, 2: k_layer<BWD>, 3: k_wgrad, 4: k_layer<..PERRAY> (view layer), 5: k_net_chain<0> (forward),
                                              // 6: k_net_chain<2> (backward-data), 7: k_net_chain<1> (forward + mask bits); the HBM-bound ray kernels (work = rays): 8: k_composite<1>, 9: k_composite<2>,
                                              // 10: k_sample_pdf_merge; 11: k_net_chain_train (training backward: products + weight gradients)
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

// One Linear(+bias+ReLU) / backward-data launch.  Which instantiation runs is decided by the shape alone (plus MOFA_PIPE=0, which
// selects the plain K loop — bit-identical, kept as the reference form of the loop): 128-feature tile when the width allows it,
// the software-pipelined K loop for an even number of panels >= 4, the per-ray-bias instantiation for the view layer.
int launch_mask_pack(const float* y, long long n_floats, unsigned long long* bits, hipStream_t st) {
    const long long blocks = n_floats / 256;                 // panel buffers are multiples of 256 rows x 16 floats
    long long grid = (blocks + 3) / 4;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_mask_pack, dim3((unsigned)grid), dim3(256), 0, st, y, blocks, bits);
    return check_launch("k_mask_pack");
}

template <int BN, bool L0, bool BWD = false>
int launch_layer(LayerArgs a, hipStream_t st) {
    a.n_tiles = a.n_padded / BN;
    const long long mt = a.m_padded / kRowTile;
    const long long total = mt * a.n_tiles;
    MOFA_REQUIRE(total > 0 && total < (1ll << 30), "layer: tile count %lld out of range", total);
    a.total_tiles = (int)total;
    const dim3 grid((unsigned)round_up(total, 8)), block(256);
    const size_t lds = 2 * (size_t)(kRowTile + BN) * 16 * sizeof(float);
    const bool prof = BN == 128 && !L0 && prof_enabled();
    const int pkind = BWD ? 2 : (a.bias_row_div ? 4 : 0);   // the view layer's per-ray-bias instantiation is its own kernel
    if (prof && prof_open(st, pkind) != MOFA_OK) return MOFA_EHIP;
    bool pipe = false;
    if constexpr (BN == 128 && !L0) pipe = config().pipe != 0 && (a.k1p + a.k2p) >= 4 && ((a.k1p + a.k2p) & 1) == 0;
    if constexpr (L0) {
        hipLaunchKernelGGL((k_layer<BN, true>), grid, block, lds, st, a);
    } else if constexpr (BWD) {
        if constexpr (BN == 128) {
            if (pipe) hipLaunchKernelGGL((k_layer<BN, false, true, false, true>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((k_layer<BN, false, true>), grid, block, lds, st, a);
        } else {
            hipLaunchKernelGGL((k_layer<BN, false, true>), grid, block, lds, st, a);
        }
    } else if (a.bias_row_div) {   // per-ray bias (the view layer): its own instantiation, see store_tile
        if constexpr (BN == 128) {
            if (pipe) hipLaunchKernelGGL((k_layer<BN, false, false, true, true>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((k_layer<BN, false, false, true>), grid, block, lds, st, a);
        } else {
            hipLaunchKernelGGL((k_layer<BN, false, false, true>), grid, block, lds, st, a);
        }
    } else {
        if constexpr (BN == 128) {
            if (pipe) hipLaunchKernelGGL((k_layer<BN, false, false, false, true>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((k_layer<BN, false>), grid, block, lds, st, a);
        } else {
            hipLaunchKernelGGL((k_layer<BN, false>), grid, block, lds, st, a);
        }
    }
    if (prof) prof_close(st, pkind, 2.0 * (double)a.m_padded * (double)a.n_padded * 16.0 * (double)(a.k1p + a.k2p));
    if (!BWD && a.mask_out && !(pipe && !a.bias_row_div)) {   // every epilogue but the contiguous-store one: bits from a pass over y
        const int rc = check_launch("k_layer");
        if (rc != MOFA_OK) return rc;
        return launch_mask_pack(a.y, a.m_padded * a.n_padded, a.mask_out, st);
    }
    return check_launch(BWD ? "k_layer<BWD>" : (L0 ? "k_layer<L0>" : "k_layer"));
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
    if (a.n_padded % 128 == 0) return l0 ? launch_layer<128, true>(a, st) : launch_layer<128, false>(a, st);
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
    return mofa_layer_forward_masked(x1, k1, x2, k2, w_packed, bias, bias_row_div, bias_rows, y, m_padded, n_padded, relu, nullptr, stream);
}

int mofa_layer_forward_masked(const float* x1, int32_t k1, const float* x2, int32_t k2, const float* w_packed,
                              const float* bias, int32_t bias_row_div, int64_t bias_rows, float* y, int64_t m_padded,
                              int32_t n_padded, int32_t relu, uint64_t* mask_bits_out, void* stream) {
    MOFA_REQUIRE(x1 && w_packed && bias && y, "layer_forward: null pointer");
    MOFA_REQUIRE(k1 > 0 && k1 % 16 == 0 && k2 >= 0 && k2 % 16 == 0 && (k2 == 0 || x2),
                 "layer_forward: k1=%d k2=%d must be multiples of 16 (x2 required when k2>0)", k1, k2);
    MOFA_REQUIRE(bias_row_div >= 0 && (bias_row_div == 0 || bias_rows > 0), "layer_forward: bad bias rows");
    LayerArgs a{};
    a.x1 = x1, a.x2 = x2, a.w = w_packed, a.bias = bias, a.y = y;
    a.k1p = k1 / 16, a.k2p = k2 / 16, a.n_padded = n_padded, a.m_padded = m_padded;
    a.bias_row_div = bias_row_div, a.bias_rows = bias_rows, a.relu = relu;
    MOFA_REQUIRE(!mask_bits_out || relu, "layer_forward: a mask tape records the ReLU of the layer (relu must be 1)");
    a.mask_out = (unsigned long long*)mask_bits_out;
    return dispatch_layer(a, false, (hipStream_t)stream);
}

/* dX = G @ W (optionally += and * ReLU mask): g panels [n_padded_fwd/16][Mp][16], wt_packed = transposed pack
 * (rows = forward input features padded to k_out_padded, contraction = g_k), dx panels [k_out_padded/16][Mp][16]. */
static int layer_backward_data(const float* g, int32_t g_k, const float* wt_packed, const float* mask, const uint64_t* mask_bits,
                               int32_t accumulate, float* dx, int64_t m_padded, int32_t k_out_padded, void* stream);
int mofa_layer_backward_data(const float* g, int32_t g_k, const float* wt_packed, const float* mask, int32_t accumulate,
                             float* dx, int64_t m_padded, int32_t k_out_padded, void* stream) {
    return layer_backward_data(g, g_k, wt_packed, mask, nullptr, accumulate, dx, m_padded, k_out_padded, stream);
}
int mofa_layer_backward_data_bits(const float* g, int32_t g_k, const float* wt_packed, const uint64_t* mask_bits, int32_t accumulate,
                                  float* dx, int64_t m_padded, int32_t k_out_padded, void* stream) {
    MOFA_REQUIRE(mask_bits, "layer_backward_data_bits: null mask");
    return layer_backward_data(g, g_k, wt_packed, nullptr, mask_bits, accumulate, dx, m_padded, k_out_padded, stream);
}
static int layer_backward_data(const float* g, int32_t g_k, const float* wt_packed, const float* mask, const uint64_t* mask_bits,
                               int32_t accumulate, float* dx, int64_t m_padded, int32_t k_out_padded, void* stream) {
    MOFA_REQUIRE(g && wt_packed && dx, "layer_backward_data: null pointer");
    MOFA_REQUIRE(g_k > 0 && g_k % 16 == 0, "layer_backward_data: g_k=%d must be a positive multiple of 16", g_k);
    LayerArgs a{};
    a.x1 = g, a.w = wt_packed, a.y = dx, a.mask = mask, a.mask_bits = (const unsigned long long*)mask_bits, a.accumulate = accumulate;
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

int mofa_pe_k_padded(int32_t n_freqs) {
    return (n_freqs >= 0 && n_freqs <= MOFA_MAX_PE_FREQS) ? (int)round_up(3 + 6 * n_freqs, 64) : MOFA_EINVAL;
}

int mofa_layer0_forward(const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                        const float* pts, int64_t n_points, int32_t S, int32_t n_freqs, const float* w_packed, const float* bias,
                        float* y, int64_t m_padded, int32_t n_padded, uint64_t* mask_bits_out, void* stream) {
    MOFA_REQUIRE(w_packed && bias && y, "layer0_forward: null pointer");
    MOFA_REQUIRE(n_freqs >= 0 && n_freqs <= MOFA_MAX_PE_FREQS, "layer0_forward: n_freqs=%d out of [0, %d]", n_freqs, MOFA_MAX_PE_FREQS);
    MOFA_REQUIRE(pts || (rays_o && rays_d && z && S > 0), "layer0_forward: need pts or (rays_o, rays_d, z, S)");
    MOFA_REQUIRE(n_points > 0 && n_points <= m_padded, "layer0_forward: n_points=%lld m_padded=%lld",
                 (long long)n_points, (long long)m_padded);
    LayerArgs a{};
    a.w = w_packed, a.bias = bias, a.y = y, a.rays_o = rays_o, a.rays_d = rays_d, a.z = z, a.pts = pts;
    a.z_row_stride = z_row_stride, a.n_points = n_points, a.S = S > 0 ? S : 1;
    a.pe_feats = 3 + 6 * n_freqs, a.k1p = mofa_pe_k_padded(n_freqs) / 16;
    a.k2p = 0, a.n_padded = n_padded, a.m_padded = m_padded, a.relu = 1, a.mask_out = (unsigned long long*)mask_bits_out;
    return dispatch_layer(a, true, (hipStream_t)stream);
}

int mofa_layer0_forward_cam(int32_t img_w, float fx, float fy, float cx, float cy, const float* c2w, const int32_t* pixels,
                            int64_t pix0, const float* z, int64_t z_row_stride, int64_t n_points, int32_t S, int32_t n_freqs,
                            const float* w_packed, const float* bias, float* y, int64_t m_padded, int32_t n_padded, void* stream) {
    MOFA_REQUIRE(c2w && z && w_packed && bias && y && img_w > 0 && S > 0, "layer0_forward_cam: bad arguments");
    MOFA_REQUIRE(n_freqs >= 0 && n_freqs <= MOFA_MAX_PE_FREQS, "layer0_forward_cam: n_freqs=%d out of [0, %d]", n_freqs, MOFA_MAX_PE_FREQS);
    MOFA_REQUIRE(n_points > 0 && n_points <= m_padded, "layer0_forward_cam: n_points=%lld m_padded=%lld", (long long)n_points,
                 (long long)m_padded);
    LayerArgs a{};
    a.w = w_packed, a.bias = bias, a.y = y, a.z = z, a.z_row_stride = z_row_stride, a.n_points = n_points, a.S = S;
    a.cam_c2w = c2w, a.cam_pix = (const int*)pixels, a.cam_pix0 = pix0, a.fx = fx, a.fy = fy, a.cx = cx, a.cy = cy, a.cam_w = img_w;
    a.pe_feats = 3 + 6 * n_freqs, a.k1p = mofa_pe_k_padded(n_freqs) / 16;
    a.k2p = 0, a.n_padded = n_padded, a.m_padded = m_padded, a.relu = 1;
    return dispatch_layer(a, true, (hipStream_t)stream);
}

int mofa_head_forward(const float* x, int32_t k_padded, int64_t m_padded, const float* w_dense, const float* b,
                      int32_t n_out, float* raw, int32_t raw_off, int64_t n_points, void* stream) {
    MOFA_REQUIRE(x && w_dense && b && raw, "head_forward: null pointer");
    MOFA_REQUIRE(k_padded % 16 == 0 && n_out >= 1 && n_out <= 4 && raw_off >= 0 && raw_off + n_out <= 4 &&
                     n_points <= m_padded,
                 "head_forward: bad shape");
    hipLaunchKernelGGL(k_head, dim3(blocks_for(n_points)), dim3(256), 0, (hipStream_t)stream, x, k_padded / 16,
                       (long long)m_padded, w_dense, b, n_out, raw, raw_off, (long long)n_points);
    return check_launch("k_head");
}

int mofa_view_bias(const float* viewdirs, int64_t n_rays, int32_t n_freqs, const float* w, int32_t n_out, int32_t ld,
                   const float* bias, float* out, int32_t n_padded, void* stream) {
    MOFA_REQUIRE(viewdirs && w && bias && out && n_rays > 0 && n_padded >= n_out, "view_bias: bad arguments");
    MOFA_REQUIRE(n_freqs >= 0 && n_freqs <= MOFA_MAX_PE_FREQS && 3 + 6 * n_freqs <= ld, "view_bias: n_freqs=%d (ld=%d)", n_freqs, ld);
    hipLaunchKernelGGL(k_view_bias, dim3((unsigned)((n_rays + 7) / 8)), dim3(256), 0, (hipStream_t)stream, viewdirs,
                       (long long)n_rays, w, n_out, ld, bias, out, n_padded, 3 + 6 * n_freqs);
    return check_launch("k_view_bias");
}

int mofa_positional_encode(const float* x, int64_t n, int32_t n_freqs, float* out, void* stream) {
    MOFA_REQUIRE(x && out && n > 0 && n_freqs >= 0 && n_freqs <= MOFA_MAX_PE_FREQS, "positional_encode: bad arguments");
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

/* arrays of MOFA_PROF_KINDS (include/mofanerf_hip.h lists the kinds).  Session of the CURRENT device. */
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
int mofa_internal_fused_forward(const float* arena, float* arena_w, const float* packed, const float* folded,
                                const float* view_bias_rows, long long bias_rows, const float* rays_o, const float* rays_d,
                                const float* z, long long z_row_stride, const float* pts, long long n_points, int S,
                                long long m_padded, int n_layers, const long long* x1_off, const long long* x2_off,
                                const long long* y_off, const long long* w_off, const long long* bias_off, const int* k1p,
                                const int* k2p, const int* n_padded, const int* bias_row_div, int pe_feats,
                                unsigned long long* mask_bits, const long long* mask_off, void* stream) {
    MOFA_REQUIRE(n_layers > 0 && n_layers <= kMaxFusedLayers, "fused_forward: %d layers (max %d)", n_layers, kMaxFusedLayers);
    MOFA_REQUIRE(m_padded > 0 && m_padded % kRowTile == 0, "fused_forward: m_padded=%lld", m_padded);
    FusedArgs a{};
    a.arena = arena, a.arena_w = arena_w, a.packed = packed, a.folded = folded, a.view_bias_rows = view_bias_rows;
    a.rays_o = rays_o, a.rays_d = rays_d, a.z = z, a.pts = pts;
    a.z_row_stride = z_row_stride, a.n_points = n_points, a.m_padded = m_padded, a.bias_rows = bias_rows;
    a.S = S > 0 ? S : 1, a.n_layers = n_layers, a.m_tiles = (int)(m_padded / kRowTile);
    a.pipe = config().pipe != 0 ? 1 : 0;
    a.pe_feats = pe_feats, a.mask_bits = mask_bits;
    for (int i = 0; i < n_layers; ++i) {
        MOFA_REQUIRE(n_padded[i] > 0 && n_padded[i] % 64 == 0, "fused_forward: layer %d has n_padded=%d", i, n_padded[i]);
        a.L[i] = FusedLayer{x1_off[i], x2_off[i], y_off[i], w_off[i], bias_off[i], mask_bits ? mask_off[i] : 0, k1p[i], k2p[i], n_padded[i], bias_row_div[i]};
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

// ---- device initialisation: the ONE place of this library that allocates and synchronises ---------------------------------------
// k_net_chain gives every XCD the queue of its own number and relies on all eight having workgroups (the default mode of the part:
// one device, 8 XCDs, round-robin dispatch).  A device whose workgroups land on fewer XCDs (compute partitions) would leave queues
// unworked, so the chained launch is only taken on a device whose CENSUS found eight populated XCDs: a 512-workgroup launch that
// counts HW_REG_XCC_ID (one tiny allocation, one launch, one stream synchronisation).  It is taken HERE, at an explicit point the
// host layer calls when it binds a network to a device (HipNet.__init__) — never inside a forward: without it
// mofa_net_forward / mofa_net_backward use the per-layer launches, which need no census and are bit-identical.
// The result is keyed on the device of `stream`.  (A stream that later restricts the CUs — hipExtStreamCreateWithCUMask — is not
// covered by the census; k_chain_verify catches what that does: unworked queues -> NaN outputs + verdict.)
static std::atomic<int> g_chain_census[kMaxDevices];       // 0: not taken, 1: eight XCDs seen AND the self-check passed, 2: fewer / it did not
extern "C" int mofa_internal_chain_selfcheck(void* stream, int* ok, char* why, size_t why_len);      // mofa_net.hip

static int stream_device(hipStream_t st) {
    hipDevice_t d = 0;
    if (st && hipStreamGetDevice(st, &d) == hipSuccess && d >= 0) return d < kMaxDevices ? (int)d : kMaxDevices - 1;
    return current_device();
}

int mofa_device_init(void* stream, int32_t* xcd_workgroups, int32_t* chain_selfcheck) {
    hipStream_t st = (hipStream_t)stream;
    const int dev = stream_device(st);
    unsigned* counts = nullptr;
    unsigned host[8] = {};
    hipError_t e = hipMalloc((void**)&counts, sizeof(host));
    if (e == hipSuccess) e = hipMemsetAsync(counts, 0, sizeof(host), st);
    if (e == hipSuccess) {      // (hipLaunchKernel reports THIS launch's status; an earlier call's pending error is not consumed here)
        void* args[] = {(void*)&counts};
        e = hipLaunchKernel((const void*)k_xcc_census, dim3(2 * compute_units(dev)), dim3(64), args, 0, st);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, counts, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (counts) (void)hipFree(counts);
    if (e != hipSuccess) {
        set_error("device_init: %s", hipGetErrorString(e));
        return MOFA_EHIP;
    }
    int populated = 0;
    for (int i = 0; i < 8; ++i) {
        populated += host[i] > 0;
        if (xcd_workgroups) xcd_workgroups[i] = (int32_t)host[i];
    }
    // the persistent 256-wide kernel's 66 KiB of dynamic LDS needs a function attribute once per device (launch_fused would set it on its
    // first launch otherwise: not a synchronisation, but it belongs here)
    if (dev == current_device() && !g_fused_attr[dev].load(std::memory_order_acquire)) {
        if (set_fused_attributes((int)((size_t)kFsFloats * sizeof(float))) != MOFA_OK) return MOFA_EHIP;
        g_fused_attr[dev].store(1, std::memory_order_release);
    }
    // Eight populated XCDs are necessary for the chained launch, not sufficient: its visibility contract (producer's plain stores seen
    // by the consumer's sc1 loads through the XCD's L2) is checked HERE, once per device — chained against per-layer launches of a
    // small fixed network, bit for bit (mofa_net.hip).  A device that fails takes the per-layer launches, and says so.
    int ok = -1;
    if (populated == 8) {
        char why[320] = "";
        const int rc = mofa_internal_chain_selfcheck(st, &ok, why, sizeof(why));
        if (rc != MOFA_OK) return rc;
        if (!ok) set_error("device_init: device %d takes the per-layer launches: %s", dev, why);
    } else {
        set_error("device_init: device %d takes the per-layer launches: the census found workgroups on %d of 8 XCDs", dev, populated);
    }
    if (chain_selfcheck) *chain_selfcheck = ok;
    g_chain_census[dev].store(populated == 8 && ok == 1 ? 1 : 2, std::memory_order_release);
    return MOFA_OK;
}

// internal (used by mofa_net.hip): 1 = this device may take the chained launch, 0 = census found fewer than eight XCDs, -1 = no census yet
int mofa_internal_chain_capable(void* stream) {
    const int c = g_chain_census[stream_device((hipStream_t)stream)].load(std::memory_order_acquire);
    return c == 0 ? -1 : (c == 1 ? 1 : 0);
}

// internal (used by mofa_net.hip): the steps of one chained launch (k_net_chain).  `state`: at least
// mofa_internal_chain_state_words(m_padded) unsigned words inside the caller's workspace.  *tiles_out = tiles the launch must finish.
size_t mofa_internal_chain_state_words(long long m_padded) { return (size_t)kChainDone + (size_t)(m_padded / kRowTile) + 32; }

int mofa_internal_chain_launch(int mode, const mofa::ChainStep* steps, int n_steps, long long m_padded, long long bias_rows, unsigned* state,
                               long long* tiles_out, void* stream) {
    MOFA_REQUIRE(n_steps > 0 && n_steps <= kMaxChainSteps, "chain_launch: %d steps (max %d)", n_steps, kMaxChainSteps);
    MOFA_REQUIRE(m_padded > 0 && m_padded % kRowTile == 0 && m_padded / kRowTile < (1 << 24), "chain_launch: m_padded=%lld", m_padded);
    MOFA_REQUIRE(steps && state && mode >= kChainForward && mode <= kChainBackward, "chain_launch: bad arguments");
    ChainArgs a{};
    a.state = state;
    a.m_padded = m_padded, a.bias_rows = bias_rows, a.m_tiles = (int)(m_padded / kRowTile), a.n_steps = n_steps;
    a.spin_limit = hook_chain_spin();         // (the shipped values unless a TEST called mofa_test_hooks(): nothing in the environment)
    a.skip_xcd = hook_chain_skip_xcd();
    double flops = 0.0;
    int before = 0;
    for (int i = 0; i < n_steps; ++i) {
        ChainStep s = steps[i];
        const int kt = s.k1p + s.k2p;
        MOFA_REQUIRE(s.x1 && s.y && s.w && (s.k2p == 0 || s.x2), "chain_launch: step %d has a null operand", i);
        MOFA_REQUIRE(s.n_padded > 0 && s.n_padded % 128 == 0 && kt >= 4 && (kt & 1) == 0 && s.k1p > 0,
                     "chain_launch: step %d (n_padded=%d, %d K panels) does not fit the pipelined 128-feature tile", i, s.n_padded, kt);
        MOFA_REQUIRE(mode == kChainBackward || s.aux, "chain_launch: forward step %d has no bias", i);
        MOFA_REQUIRE(mode != kChainForwardMask || (s.flags & 1), "chain_launch: a mask tape records the ReLU of the layer (step %d has none)", i);
        MOFA_REQUIRE(mode != kChainForwardMask || s.bits || s.bias_row_div, "chain_launch: step %d has no place for its mask bits", i);
        MOFA_REQUIRE(mode != kChainBackward || !(s.aux && s.bits), "chain_launch: step %d has both an fp32 mask and mask bits", i);
        s.n_tiles = s.n_padded / 128, s.tiles_before = before;
        a.S[i] = s;
        before += s.n_tiles;
        flops += 2.0 * (double)m_padded * (double)s.n_padded * 16.0 * (double)kt;
    }
    a.tiles_per_m = before;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(state, 0, mofa_internal_chain_state_words(m_padded) * sizeof(unsigned), st) != hipSuccess) return check_launch("hipMemsetAsync(chain state)");
    const int dev = stream_device(st);
    const long long tiles = (long long)a.m_tiles * before;
    if (tiles_out) *tiles_out = tiles;
    const int slots = 2 * compute_units(dev);
    const int grid = tiles < slots ? (int)round_up(tiles, 8) : slots;       // two resident workgroups per CU
    const size_t lds = 2 * (size_t)(kRowTile + 128) * 16 * sizeof(float) + 64;
    const bool prof = prof_enabled();
    const int pkind = mode == kChainBackward ? 6 : (mode == kChainForwardMask ? 7 : 5);
    if (prof && prof_open(st, pkind) != MOFA_OK) return MOFA_EHIP;
    if (mode == kChainForward) hipLaunchKernelGGL(k_net_chain<kChainForward>, dim3(grid), dim3(256), lds, st, a);
    else if (mode == kChainForwardMask) hipLaunchKernelGGL(k_net_chain<kChainForwardMask>, dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(k_net_chain<kChainBackward>, dim3(grid), dim3(256), lds, st, a);
    if (prof) prof_close(st, pkind, flops);
    return check_launch("k_net_chain");
}

// internal (used by mofa_net.hip): the steps of one chained TRAINING-backward launch (k_net_chain_train): backward-data products and weight
// gradients (flags bit 1, `spt` = wg_split's plan for that product).  *tiles_out = entries the launch must finish.
int mofa_internal_chain_train_launch(const mofa::ChainStep* steps, int n_steps, long long m_padded, long long n_points, unsigned* state,
                                     long long* tiles_out, void* stream) {
    MOFA_REQUIRE(n_steps > 0 && n_steps <= kMaxChainSteps, "chain_train_launch: %d steps (max %d)", n_steps, kMaxChainSteps);
    MOFA_REQUIRE(m_padded > 0 && m_padded % kRowTile == 0 && m_padded / kRowTile < (1 << 24) && n_points > 0 && n_points <= m_padded &&
                     m_padded - n_points < kRowTile,
                 "chain_train_launch: m_padded=%lld n_points=%lld", m_padded, n_points);
    MOFA_REQUIRE(steps && state, "chain_train_launch: bad arguments");
    TrainChainArgs a{};
    a.state = state;
    a.m_padded = m_padded, a.n_points = n_points, a.m_tiles = (int)(m_padded / kRowTile), a.n_steps = n_steps;
    a.spin_limit = hook_chain_spin();
    a.skip_xcd = hook_chain_skip_xcd();
    a.pipe = config().pipe != 0 ? 1 : 0;
    double flops = 0.0;
    int before = 0;
    for (int i = 0; i < n_steps; ++i) {
        ChainStep s = steps[i];
        if (s.flags & kChainStepWgrad) {
            MOFA_REQUIRE(s.x1 && s.x2 && s.y, "chain_train_launch: weight-gradient step %d has a null operand", i);
            MOFA_REQUIRE(s.n_padded > 0 && s.n_padded % 128 == 0 && s.k1p > 0 && (s.k1p * 16) % 256 == 0,
                         "chain_train_launch: weight-gradient step %d (%d x %d) does not fit the 128 x 256 tile", i, s.n_padded, s.k1p * 16);
            s.n_tiles = (s.n_padded / 128) * (s.k1p * 16 / 256);
            MOFA_REQUIRE(s.spt == wg_split(m_padded, s.n_tiles).spt, "chain_train_launch: weight-gradient step %d does not carry wg_split's plan", i);
            flops += 2.0 * (double)n_points * (double)s.n_padded * 16.0 * (double)s.k1p;
        } else {
            MOFA_REQUIRE(s.x1 && s.y && s.w && s.k2p == 0, "chain_train_launch: step %d has a null operand / a second source", i);
            MOFA_REQUIRE(s.n_padded > 0 && s.n_padded % 128 == 0 && s.k1p >= 4 && (s.k1p & 1) == 0,
                         "chain_train_launch: step %d (n_padded=%d, %d K panels) does not fit the pipelined 128-feature tile", i, s.n_padded, s.k1p);
            MOFA_REQUIRE(!s.bits, "chain_train_launch: step %d carries mask bits (training keeps the fp32 tape)", i);
            s.n_tiles = s.n_padded / 128;
            flops += 2.0 * (double)m_padded * (double)s.n_padded * 16.0 * (double)s.k1p;
        }
        s.tiles_before = before;
        a.S[i] = s;
        before += s.n_tiles;
    }
    // entries per XCD: products own their row tiles, weight gradients their splits (the kernel's own arithmetic)
    const int mpx = (a.m_tiles + 7) >> 3;
    long long tiles = 0;
    for (int x = 0; x < 8; ++x) {
        const int m_lo = x * mpx, m_cnt = m_lo < a.m_tiles ? (a.m_tiles - m_lo < mpx ? a.m_tiles - m_lo : mpx) : 0;
        for (int i = 0; i < n_steps; ++i)
            tiles += (long long)a.S[i].n_tiles * ((a.S[i].flags & kChainStepWgrad) ? (m_cnt + a.S[i].spt - 1) / a.S[i].spt : m_cnt);
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(state, 0, mofa_internal_chain_state_words(m_padded) * sizeof(unsigned), st) != hipSuccess) return check_launch("hipMemsetAsync(chain state)");
    if (tiles_out) *tiles_out = tiles;
    const int slots = 2 * compute_units(stream_device(st));
    const int grid = tiles < slots ? (int)round_up(tiles, 8) : slots;       // two resident workgroups per CU
    constexpr size_t lds = 2 * (size_t)(128 / 16 + 256 / 16) * (WgCfg<128, 256>::MC * 16 + 16) * sizeof(float) + 64;
    const bool prof = prof_enabled();
    if (prof && prof_open(st, 11) != MOFA_OK) return MOFA_EHIP;
    hipLaunchKernelGGL(k_net_chain_train, dim3(grid), dim3(256), lds, st, a);
    if (prof) prof_close(st, 11, flops);
    return check_launch("k_net_chain_train");
}

// internal (used by mofa_net.hip): behind a chained launch (and whatever consumed its outputs into p0 / p1 / p2): verify, poison, verdict
int mofa_internal_chain_verify(const unsigned* state, long long tiles, unsigned* verdict, float* p0, long long n0, float* p1, long long n1,
                               float* p2, long long n2, float* p3, long long n3, void* stream) {
    n0 = p0 ? n0 : 0, n1 = p1 ? n1 : 0, n2 = p2 ? n2 : 0, n3 = p3 ? n3 : 0;
    long long n = n0 > n1 ? n0 : n1;
    n = n > n2 ? n : n2;
    n = n > n3 ? n : n3;
    long long grid = (n + 255) / 256;
    grid = grid < 1 ? 1 : (grid > 1024 ? 1024 : grid);
    hipLaunchKernelGGL(k_chain_verify, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, state + kChainStatus, (unsigned)tiles, verdict, p0, n0, p1,
                       n1, p2, n2, p3, n3);
    return check_launch("k_chain_verify");
}

// internal (used by mofa_net.hip): NaN into `count` more buffers if the chained launch behind `state` ended incomplete
int mofa_internal_chain_poison(const unsigned* state, long long tiles, float* const* ptrs, const long long* sizes, int count, void* stream) {
    for (int b0 = 0; b0 < count; b0 += kMaxPoison) {
        PoisonArgs a{};
        a.count = count - b0 < kMaxPoison ? count - b0 : kMaxPoison;
        for (int i = 0; i < a.count; ++i) a.p[i] = ptrs[b0 + i], a.n[i] = ptrs[b0 + i] ? sizes[b0 + i] : 0;
        hipLaunchKernelGGL(k_chain_poison, dim3(256), dim3(256), 0, (hipStream_t)stream, state + kChainStatus, (unsigned)tiles, a);
    }
    return check_launch("k_chain_poison");
}

// internal (used by mofa_net.hip): bits of (y > 0) for a panel buffer of n_floats (a multiple of 256) floats
int mofa_internal_mask_pack(const float* y, long long n_floats, unsigned long long* bits, void* stream) {
    return launch_mask_pack(y, n_floats, bits, (hipStream_t)stream);
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
