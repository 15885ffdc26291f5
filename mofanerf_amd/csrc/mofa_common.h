// Shared device/host helpers for libmofanerf_hip.so (gfx950 only — no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mofanerf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace mofa {

constexpr int kRowTile = MOFA_ROW_TILE;  // activation rows per workgroup tile
constexpr int kPanelK = 16;              // floats per panel row (64 B)

__host__ __device__ inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// element (row,k) of a panel matrix with `rows` rows (see include/mofanerf_hip.h, "Panel layout")
__host__ __device__ inline int64_t panel_index(int64_t rows, int64_t row, int k) {
    return (int64_t)(k >> 4) * rows * kPanelK + row * kPanelK + ((((k >> 2) & 3) ^ ((int)(row >> 2) & 3)) << 2) +
           (k & 3);
}

// One pinhole ray (tools/run_nerf_helpers.py:153-168) with the reference's operation order and separately rounded ops:
// dirs = [(i - cx) / fx, -(j - cy) / fy, -1];  rays_d[a] = (dirs0 * c[a][0] + dirs1 * c[a][1]) + dirs2 * c[a][2];  rays_o = c[:, 3].
// Shared by k_get_rays (full frames / pixel lists) and the layer-0 prologue's camera mode, so both are bit-identical.
__device__ __forceinline__ void pinhole_ray(int i, int j, float fx, float fy, float cx, float cy, const float* __restrict__ c2w,
                                            float (&ro)[3], float (&rd)[3]) {
    const float d0 = __fdiv_rn((float)i - cx, fx);
    const float d1 = -__fdiv_rn((float)j - cy, fy);
    const float d2 = -1.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        rd[a] = __fadd_rn(__fadd_rn(__fmul_rn(d0, c2w[a * 4 + 0]), __fmul_rn(d1, c2w[a * 4 + 1])), __fmul_rn(d2, c2w[a * 4 + 2]));
        ro[a] = c2w[a * 4 + 3];
    }
}

// ReLU with torch's NaN behaviour (F.relu(nan) = nan; fmaxf(nan, 0) would be 0): a NaN that enters the network (bad
// input) must reach the image exactly as it does in the reference.
// gfx950's v_maximum3_f32 is IEEE-754-2019 `maximum` (NaN-propagating): ONE instruction instead of compare + select.
__device__ __forceinline__ float relu_np(float v) { return __builtin_elementwise_maximum(v, 0.0f); }

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Run-time knobs.  Each selects between BIT-IDENTICAL forms of the exact-fp32 path (plain / pipelined K loop; persistent or chained /
// per-layer network launch) — there is no reduced-precision mode in this library; measurement arms live in csrc/measure/.
// The environment is read ONCE (at library load, and again only when the host calls mofa_config_reload()) into an immutable
// snapshot, so no launch path calls getenv and concurrent host threads see one consistent configuration.
// -1 = "not set: use the built-in heuristic".
struct Config {
    int fused = -1;       // MOFA_FUSED=0/1: persistent whole-network kernel off / on
    int pipe = -1;        // MOFA_PIPE=0: the plain K loops (layer kernel, persistent kernel, weight gradient) instead of the pipelined ones
    int chain = -1;       // MOFA_CHAIN=0: per-layer launches for the wide networks instead of the chained launch (k_net_chain)
    int chain_train = -1; // MOFA_CHAIN_TRAIN=1: the TRAINING backward (products + weight gradients) as chained launches too (k_net_chain_train).
                          // Off by default: bit-identical, measured 0.2-0.4 % SLOWER than the per-layer launches it replaces
};
const Config& config();

// Test hooks (mofa_test_hooks(), include/mofanerf_hip.h): the ONLY way to reach k_net_chain's failure paths and the self-check's
// mismatch path on purpose.  Nothing in the environment sets them (round 5 read two of them from MOFA_* variables: a stray variable
// in production turned every wide-network launch into NaN + MofaError); they are process-wide atomics only an explicit call changes.
constexpr unsigned kChainSpinDefault = 1u << 22;   // polls (each >= ~1 us with its s_sleep) before a dependency wait gives up: seconds
unsigned hook_chain_spin();        // polls before a dependency wait of k_net_chain gives up
int hook_chain_skip_xcd();         // k_net_chain's workgroups on this XCD leave at once (an unworked queue); -1 = none
int hook_selfcheck_poison();       // mofa_device_init's chained-vs-per-layer self-check sees one flipped bit (forces its fallback)

constexpr int kMaxDevices = 64;
#define MOFA_MAX_CHAIN_STEPS 40    /* MFMA layers one chained launch (k_net_chain) can hold: its step table travels as kernel arguments */
// One GEMM of a chained launch (k_net_chain, mofa_mlp.hip): a forward layer (Linear + bias + ReLU) or a backward-data product, as the
// driver (mofa_net.hip) lists them in a topological order.  Plain pointers: the buffers of one launch live in several allocations
// (workspace, tape, packed weights, folded biases).
struct ChainStep {
    const float* x1;            // operand panels [k1p][m_padded][16]
    const float* x2;            // second source of a skip layer's contraction, or NULL
    float* y;                   // output panels [n_padded / 16][m_padded][16]
    const float* w;             // packed weights (forward) / transposed pack (backward), panels [(k1p + k2p)][n_padded][16]
    const float* aux;           // forward: the bias row (bias_row_div == 0) or the per-ray bias rows; backward: the saved fp32 activation the
                                // result is masked with, or NULL
    unsigned long long* bits;   // forward with a mask tape: where (y > 0) goes as bits (NULL: none); backward: the mask bits (NULL: none)
    int k1p, k2p, n_padded, n_tiles;
    int bias_row_div;           // forward: 0, or points per bias row (the view layer's per-ray rows)
    int flags;                  // forward: bit 0 = ReLU; backward: bit 0 = accumulate into y; bit 1 (k_net_chain_train only) = this step is a
                                // WEIGHT GRADIENT dW = G^T X: x1 = G panels [n_padded / 16][m_padded][16], x2 = X panels [k1p][m_padded][16],
                                // y = partial sums [split][n_padded][16 k1p], aux = bias partial sums [split][n_padded] or NULL,
                                // n_tiles = (n_padded / 128) * (16 k1p / 256) output tiles per split
    int tiles_before;           // filled by the launcher: tiles of the earlier steps over one row tile
    int spt;                    // weight-gradient steps: row tiles per split of the points (wg_split of THIS product; 0 otherwise)
};
enum ChainMode { kChainForward = 0, kChainForwardMask = 1, kChainBackward = 2 };
constexpr int kChainStepWgrad = 2;   // ChainStep::flags bit

// How the weight gradient dW = G^T X splits its contraction over the points (training).  ONE plan for the per-layer kernel (k_wgrad:
// grid.y = splits) and for the chained training backward (k_net_chain_train: a split's units are queue entries of the XCD that owns its
// rows), so that both sum the same points in the same order — bit-identical partial sums, one deterministic second stage.  Splits are whole
// row tiles (256 points) and never straddle the row ranges k_net_chain gives the XCDs (XCD x owns row tiles [x mpx, (x + 1) mpx)).
struct WgSplit {
    int m_tiles;   // row tiles of the batch
    int mpx;       // row tiles per XCD range: ceil(m_tiles / 8)
    int spt;       // row tiles per split
    int nspx;      // splits per full XCD range: ceil(mpx / spt)
    int total;     // splits in all (the last populated range may hold fewer)
};
__host__ __device__ inline WgSplit wg_split(long long m_padded, int out_tiles /* output tiles of the product: (N / TN) * (K / TK) */) {
    WgSplit w;
    w.m_tiles = (int)(m_padded / kRowTile);
    w.mpx = (w.m_tiles + 7) >> 3;
    int want = (128 + out_tiles - 1) / out_tiles;      // ~1024 units per product: two rounds of the chip's 512 workgroup slots
    if (want < 1) want = 1;
    w.spt = (w.mpx + want - 1) / want;
    if (w.spt < 1) w.spt = 1;
    w.nspx = (w.mpx + w.spt - 1) / w.spt;
    const int full = w.m_tiles / w.mpx, rem = w.m_tiles - full * w.mpx;
    w.total = full * w.nspx + (rem + w.spt - 1) / w.spt;
    return w;
}
// row tiles [first, first + count) of split s
__host__ __device__ inline void wg_split_rows(const WgSplit& w, int s, int& first, int& count) {
    const int x = s / w.nspx, j = s - x * w.nspx;
    first = x * w.mpx + j * w.spt;
    int end = first + w.spt, xend = (x + 1) * w.mpx;
    if (xend > w.m_tiles) xend = w.m_tiles;
    if (end > xend) end = xend;
    count = end - first;
}

int current_device();              // hipGetDevice, clamped to [0, kMaxDevices)
int compute_units(int device);     // multiProcessorCount, cached per device

}  // namespace mofa

#define MOFA_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            mofa::set_error(__VA_ARGS__);    \
            return MOFA_EINVAL;              \
        }                                    \
    } while (0)
