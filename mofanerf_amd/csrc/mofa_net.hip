// Host-side driver of the conditioned MLP: layer plan, weight packing, per-call bias folding and the
// launch sequence of one run_network call (models/render_class.py:69-94 + models/model.py:121-137).
// Pure launch code: no allocation, no host synchronisation, every buffer is the caller's.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <utility>
#include <vector>

#include "mofa_common.h"

extern "C" {
int mofa_internal_fold_bias(const float* w, int n_out, int ld, int col0, int ncols, const float* code,
                            const float* bias, float* out, int n_padded, void* stream);
int mofa_internal_dense_rows(const float* w, int n_out, int ld, int col0, int ncols, float* dst, int k_padded,
                             void* stream);
int mofa_internal_raw_colsum(const float* d_raw, long long n_points, float* out_rgb, float* out_sigma, float* scratch, void* stream);
int mofa_internal_fused_forward(const float* arena, float* arena_w, const float* packed, const float* folded,
                                const float* view_bias_rows, long long bias_rows, const float* rays_o, const float* rays_d,
                                const float* z, long long z_row_stride, const float* pts, long long n_points, int S,
                                long long m_padded, int n_layers, const long long* x1_off, const long long* x2_off,
                                const long long* y_off, const long long* w_off, const long long* bias_off, const int* k1p,
                                const int* k2p, const int* n_padded, const int* bias_row_div, int pe_feats,
                                unsigned long long* mask_bits, const long long* mask_off, void* stream);
int mofa_internal_mask_pack(const float* y, long long n_floats, unsigned long long* bits, void* stream);
size_t mofa_internal_chain_state_words(long long m_padded);
int mofa_internal_chain_capable(void* stream);
int mofa_internal_chain_launch(int mode, const mofa::ChainStep* steps, int n_steps, long long m_padded, long long bias_rows, unsigned* state,
                               long long* tiles_out, void* stream);
int mofa_internal_chain_verify(const unsigned* state, long long tiles, unsigned* verdict, float* p0, long long n0, float* p1, long long n1,
                               float* p2, long long n2, float* p3, long long n3, void* stream);
int mofa_internal_bias_grad_split(const float* g, long long m_padded, long long n_points, int n_padded, float* out, float* workspace, void* stream);
int mofa_internal_head_weight_grad_split(const float* d_raw, int32_t raw_off, int32_t n_out, const float* x, int32_t k_padded,
                                         int64_t m_padded, int64_t n_points, int32_t ncols, float* dst, int32_t ld, float* workspace,
                                         void* stream);
int mofa_internal_chain_selfcheck(void* stream, int* ok, char* why, size_t why_len);
int mofa_internal_chain_train_launch(const mofa::ChainStep* steps, int n_steps, long long m_padded, long long n_points, unsigned* state,
                                     long long* tiles_out, void* stream);
int mofa_internal_chain_poison(const unsigned* state, long long tiles, float* const* ptrs, const long long* sizes, int count, void* stream);
int mofa_internal_wgrad_reduce(const float* partial, int splits, int n_padded, int k_padded, int n_out, int ncols, float* dst, int ld,
                               int col0, float* bias_out, void* stream);
}

namespace mofa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- configuration snapshot + per-device caches (no getenv / no device-0 assumptions on the launch paths) ----------------
namespace {
Config read_env() {
    Config c;
    auto tri = [](const char* name) { const char* e = getenv(name); return e ? (e[0] == '1' ? 1 : 0) : -1; };
    c.fused = tri("MOFA_FUSED"), c.pipe = tri("MOFA_PIPE"), c.chain = tri("MOFA_CHAIN"), c.chain_train = tri("MOFA_CHAIN_TRAIN");
    return c;
}
std::atomic<unsigned> g_hook_spin{kChainSpinDefault};
std::atomic<int> g_hook_skip_xcd{-1};
std::atomic<int> g_hook_poison{0};
// two slots + an atomic index: readers never see a half-written snapshot, reload is rare and host-side only
Config g_cfg[2] = {read_env(), Config{}};
std::atomic<int> g_cfg_cur{0};
std::atomic<int> g_cus[kMaxDevices];
}  // namespace

const Config& config() { return g_cfg[g_cfg_cur.load(std::memory_order_acquire)]; }
unsigned hook_chain_spin() { return g_hook_spin.load(std::memory_order_relaxed); }
int hook_chain_skip_xcd() { return g_hook_skip_xcd.load(std::memory_order_relaxed); }
int hook_selfcheck_poison() { return g_hook_poison.load(std::memory_order_relaxed); }

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < kMaxDevices ? dev : kMaxDevices - 1;
}

int compute_units(int device) {
    int v = g_cus[device].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    g_cus[device].store(v, std::memory_order_relaxed);
    return v;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MOFA_EHIP;
    }
    return MOFA_OK;
}

namespace {

enum Fold { kNone = 0, kExp, kShape, kTex, kView };

struct Layer {
    int n_out, ld;        // PyTorch weight [n_out, ld]
    int col0[2], ncols[2];// per-point-varying column ranges (second one only for skip layers)
    int nsrc;
    int fold;             // which per-call code feeds the constant columns [0, fold_cols) or [63,93)
    int fold_col0, fold_cols;
    int n_padded, k_padded[2];
    bool head;            // dense head (alpha / rgb) instead of an MFMA layer
    size_t packed_off, folded_off;
    size_t packed_t_off[2];   // transposed packs for the backward-data GEMMs (one per source part)
    size_t tape_cols;         // sum of n_padded of the MFMA layers before this one (tape slot = Mp * tape_cols)
};

struct Plan {
    int D, W, Wp, Hp;
    int pe_k;             // K of layer 0's operand panels: roundup(3 + 6 * multires, 64)
    std::vector<Layer> L;
    size_t packed_floats = 0, folded_floats = 0, packed_t_floats = 0, tape_cols = 0;
    // indices into L
    int xyz0, bim0, bim_skip, uv0, uv_skip, view, alpha, rgb;
};

Plan make_plan(MofaNetShape s) {
    Plan p;
    p.D = s.D, p.W = s.W;
    const int W = s.W, Wp = (int)round_up(W, 64), Hp = (int)round_up(W / 2, 64);
    p.Wp = Wp, p.Hp = Hp;
    const int PE = 3 + 6 * s.pe_point_freqs, PV = 3 + 6 * s.pe_view_freqs;
    p.pe_k = (int)round_up(PE, 64);
    auto plain = [&](int n_out, int ld) {
        Layer l{};
        l.n_out = n_out, l.ld = ld, l.nsrc = 1, l.col0[0] = 0, l.ncols[0] = ld, l.fold = kNone;
        l.n_padded = (int)round_up(n_out, 64), l.k_padded[0] = (int)round_up(ld, 64), l.head = false;
        return l;
    };
    // xyzEncode: skipMLP(D=3, skip=None) -> Linear0..3 (models/model.py:97, :220-223)
    {
        Layer l = plain(W, PE + s.ch_exp);
        l.ncols[0] = PE, l.k_padded[0] = p.pe_k, l.fold = kExp, l.fold_col0 = PE, l.fold_cols = s.ch_exp;
        p.xyz0 = (int)p.L.size();
        p.L.push_back(l);
        for (int i = 1; i < 4; ++i) p.L.push_back(plain(W, W));
    }
    auto cond_stack = [&](int cin, int fold, int& first, int& skip) {
        Layer l0 = plain(W, cin + W);
        l0.col0[0] = cin, l0.ncols[0] = W, l0.k_padded[0] = Wp, l0.fold = fold, l0.fold_col0 = 0, l0.fold_cols = cin;
        first = (int)p.L.size();
        p.L.push_back(l0);
        for (int i = 1; i <= 4; ++i) p.L.push_back(plain(W, W));
        // input [code | x | h] (models/model.py:215,229).  The contraction walks h FIRST, then x (part 0 = the h columns, part 1 = the
        // x columns): h is the previous layer's output — still on chip in the persistent kernels — while x comes back from memory.
        // (Which half is summed first is a free choice — the reference's GEMM fixes none — but it is ONE choice for every kernel here.)
        Layer ls = plain(W, cin + 2 * W);
        ls.nsrc = 2, ls.col0[0] = cin + W, ls.ncols[0] = W, ls.col0[1] = cin, ls.ncols[1] = W;
        ls.k_padded[0] = Wp, ls.k_padded[1] = Wp, ls.fold = fold, ls.fold_col0 = 0, ls.fold_cols = cin;
        skip = (int)p.L.size();
        p.L.push_back(ls);
        for (int i = 1; i < s.D - 5; ++i) p.L.push_back(plain(W, W));
    };
    cond_stack(s.ch_shape, kShape, p.bim0, p.bim_skip);
    cond_stack(s.ch_tex, kTex, p.uv0, p.uv_skip);
    {
        Layer l = plain(W / 2, PV + W);
        l.col0[0] = PV, l.ncols[0] = W, l.k_padded[0] = Wp, l.fold = kView, l.fold_col0 = 0, l.fold_cols = PV;
        l.n_padded = Hp;
        p.view = (int)p.L.size();
        p.L.push_back(l);
        Layer a = plain(1, W);
        a.head = true, a.k_padded[0] = Wp, a.n_padded = 4;
        p.alpha = (int)p.L.size();
        p.L.push_back(a);
        Layer r = plain(3, W / 2);
        r.head = true, r.k_padded[0] = Hp, r.n_padded = 4;
        p.rgb = (int)p.L.size();
        p.L.push_back(r);
    }
    for (auto& l : p.L) {
        l.packed_off = p.packed_floats;
        l.folded_off = p.folded_floats;
        const size_t kp = (size_t)l.k_padded[0] + (l.nsrc > 1 ? l.k_padded[1] : 0);
        p.packed_floats += l.head ? (size_t)l.n_out * kp : (size_t)l.n_padded * kp;
        p.packed_floats = (size_t)round_up((int64_t)p.packed_floats, 64);
        p.folded_floats += (l.fold == kView) ? 0 : (size_t)l.n_padded;
        l.tape_cols = p.tape_cols;
        l.packed_t_off[0] = l.packed_t_off[1] = 0;
        if (!l.head) {
            p.tape_cols += (size_t)l.n_padded;
            for (int part = 0; part < l.nsrc; ++part) {
                l.packed_t_off[part] = p.packed_t_floats;
                p.packed_t_floats += (size_t)l.k_padded[part] * (size_t)l.n_padded;
            }
        }
    }
    return p;
}

bool shape_ok(MofaNetShape s) {
    return s.D >= 6 && s.D <= 64 && s.W >= 2 && s.W <= 8192 && s.W % 2 == 0 && s.pe_point_freqs >= 0 && s.pe_point_freqs <= MOFA_MAX_PE_FREQS &&
           s.pe_view_freqs >= 0 && s.pe_view_freqs <= MOFA_MAX_PE_FREQS && s.ch_exp >= 0 && s.ch_exp <= MOFA_MAX_CODE && s.ch_shape >= 0 &&
           s.ch_shape <= MOFA_MAX_CODE && s.ch_tex >= 0 && s.ch_tex <= MOFA_MAX_CODE;
}
// ---- the chained TRAINING backward (k_net_chain_train; opt-in: MOFA_CHAIN_TRAIN=1): which shapes take it, and what its weight-gradient
// partial sums need.  Every product of the network must fit the chain's two tile forms: backward-data 256 x 128 (outputs Wp wide, contraction Wp or Hp in an
// even number >= 4 of 16-panels) and the weight gradient's 128 x 256 (N in {Wp, Hp} a multiple of 128, K = Wp a multiple of 256); both
// launches' step tables (products + weight gradients) must fit MOFA_MAX_CHAIN_STEPS.
bool train_chain_shape(const Plan& p, int D) {
    const int n2 = D - 5;
    return config().chain_train == 1 && p.Wp % 256 == 0 && p.Hp % 128 == 0 && 2 * (n2 + 9) <= MOFA_MAX_CHAIN_STEPS;
}
// partial sums [splits][N][K + 1] of one weight-gradient step (wg_split's plan for this product)
size_t train_partial_floats(int64_t m_padded, int n_padded, int k_padded) {
    const WgSplit sp = wg_split(m_padded, (n_padded / 128) * (k_padded / 256));
    return (size_t)round_up((int64_t)((size_t)sp.total * n_padded * ((size_t)k_padded + 1)), 64);
}
// the weight-gradient scratch of a training backward: one product's partials (per-layer launches), or — for the shapes the chained
// form takes — the partials of EVERY weight gradient of the larger of its two launches (shape stack + xyzEncode 3..1: D + 4 products of
// Wp x Wp), which are reduced behind the launch
size_t train_wws_floats(const Plan& p, int D, int64_t n_points) {
    const int64_t mp = round_up(n_points, kRowTile);
    size_t n = mofa_weight_grad_workspace_floats(n_points, p.Wp, p.Wp);
    if (train_chain_shape(p, D)) {
        const size_t chain = (size_t)(D + 4) * train_partial_floats(mp, p.Wp, p.Wp);
        if (chain > n) n = chain;
    }
    return n;
}

// kernels of the chained launch's self-check (mofa_internal_chain_selfcheck below)
__global__ __launch_bounds__(256) void k_selfcheck_fill(float* __restrict__ p, long long n, unsigned seed, float scale, float offset) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = ((unsigned)i * 2654435761u) ^ seed;
    h ^= h >> 16, h *= 0x85ebca6bu, h ^= h >> 13, h *= 0xc2b2ae35u, h ^= h >> 16;
    p[i] = ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale + offset;      // uniform in [-scale, scale) + offset
}
// out[0] += words that differ, out[1] += non-finite values of a, out[2] += non-zero values of a
__global__ __launch_bounds__(256) void k_selfcheck_compare(const unsigned* __restrict__ a, const unsigned* __restrict__ b, long long n,
                                                           unsigned* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned x = a[i], y = b[i];
    if (x != y) atomicAdd(out, 1u);
    if ((x & 0x7f800000u) == 0x7f800000u) atomicAdd(out + 1, 1u);
    if ((x & 0x7fffffffu) != 0u) atomicAdd(out + 2, 1u);
}
__global__ void k_selfcheck_poison(unsigned* p) { p[0] ^= 1u; }

#define MOFA_SHAPE_FMT "D=%d W=%d multires=%d multires_views=%d ch_exp=%d ch_shape=%d ch_tex=%d"
#define MOFA_SHAPE_ARGS(s) (s).D, (s).W, (s).pe_point_freqs, (s).pe_view_freqs, (s).ch_exp, (s).ch_shape, (s).ch_tex

}  // namespace
}  // namespace mofa

using namespace mofa;

extern "C" {

int mofa_abi_version(void) { return MOFA_ABI_VERSION; }

int mofa_config_reload(void) {
    const int next = 1 - g_cfg_cur.load(std::memory_order_acquire);
    g_cfg[next] = read_env();
    g_cfg_cur.store(next, std::memory_order_release);
    return MOFA_OK;
}
const char* mofa_last_error(void) { return g_err; }

int mofa_test_hooks(uint32_t chain_spin_limit, int32_t chain_skip_xcd, int32_t selfcheck_poison) {
    MOFA_REQUIRE(chain_skip_xcd >= -1 && chain_skip_xcd < 8, "test_hooks: chain_skip_xcd=%d (want -1 .. 7)", chain_skip_xcd);
    g_hook_spin.store(chain_spin_limit ? chain_spin_limit : kChainSpinDefault, std::memory_order_relaxed);
    g_hook_skip_xcd.store(chain_skip_xcd, std::memory_order_relaxed);
    g_hook_poison.store(selfcheck_poison ? 1 : 0, std::memory_order_relaxed);
    return MOFA_OK;
}

int mofa_net_num_layers(MofaNetShape s) { return shape_ok(s) ? 2 * s.D + 7 : MOFA_EINVAL; }

int mofa_net_layer_dims(MofaNetShape s, int32_t li, int32_t* n_out, int32_t* n_in) {
    MOFA_REQUIRE(shape_ok(s), "net_layer_dims: unsupported shape " MOFA_SHAPE_FMT, MOFA_SHAPE_ARGS(s));
    MOFA_REQUIRE(n_out && n_in && li >= 0 && li < 2 * s.D + 7, "net_layer_dims: layer %d of %d", li, 2 * s.D + 7);
    const Plan p = make_plan(s);
    *n_out = p.L[li].n_out, *n_in = p.L[li].ld;
    return MOFA_OK;
}
size_t mofa_net_packed_floats(MofaNetShape s) { return shape_ok(s) ? make_plan(s).packed_floats : 0; }
size_t mofa_net_folded_floats(MofaNetShape s) { return shape_ok(s) ? make_plan(s).folded_floats : 0; }

size_t mofa_net_workspace_floats(MofaNetShape s, int64_t n_points, int64_t n_rays) {
    if (!shape_ok(s) || n_points <= 0 || n_rays <= 0) return 0;
    const Plan p = make_plan(s);
    const size_t mp = (size_t)round_up(n_points, kRowTile);
    return 4 * mp * (size_t)p.Wp + (size_t)n_rays * (size_t)p.Hp + 64 + mofa_internal_chain_state_words((long long)mp);   // (+ k_net_chain's queue state)
}

int mofa_net_pack(MofaNetShape s, const float* const* weights, float* packed, void* stream) {
    MOFA_REQUIRE(shape_ok(s), "net_pack: unsupported shape " MOFA_SHAPE_FMT, MOFA_SHAPE_ARGS(s));
    MOFA_REQUIRE(weights && packed, "net_pack: null pointer");
    const Plan p = make_plan(s);
    for (size_t li = 0; li < p.L.size(); ++li) {
        const Layer& l = p.L[li];
        MOFA_REQUIRE(weights[li], "net_pack: weights[%zu] is null", li);
        float* dst = packed + l.packed_off;
        int rc;
        if (l.head) {
            rc = mofa_internal_dense_rows(weights[li], l.n_out, l.ld, 0, l.ncols[0], dst, l.k_padded[0], stream);
        } else {
            rc = mofa_pack_panels(weights[li], l.n_out, l.ld, l.col0[0], l.ncols[0], dst, l.n_padded, 0,
                                  l.k_padded[0], stream);
            if (rc == MOFA_OK && l.nsrc > 1)
                rc = mofa_pack_panels(weights[li], l.n_out, l.ld, l.col0[1], l.ncols[1], dst, l.n_padded,
                                      l.k_padded[0] / 16, l.k_padded[1], stream);
        }
        if (rc != MOFA_OK) return rc;
    }
    return MOFA_OK;
}

int mofa_net_fold(MofaNetShape s, const float* const* weights, const float* const* biases, const float* exp_code,
                  const float* shape_code, const float* tex_code, float* folded, void* stream) {
    MOFA_REQUIRE(shape_ok(s), "net_fold: unsupported shape " MOFA_SHAPE_FMT, MOFA_SHAPE_ARGS(s));
    MOFA_REQUIRE(weights && biases && folded && (exp_code || !s.ch_exp) && (shape_code || !s.ch_shape) && (tex_code || !s.ch_tex),
                 "net_fold: null pointer");
    const Plan p = make_plan(s);
    for (size_t li = 0; li < p.L.size(); ++li) {
        const Layer& l = p.L[li];
        if (l.fold == kView) continue;  // per-ray: mofa_view_bias inside mofa_net_forward
        MOFA_REQUIRE(weights[li] && biases[li], "net_fold: layer %zu has a null weight/bias", li);
        const float* code = l.fold == kExp ? exp_code : l.fold == kShape ? shape_code : l.fold == kTex ? tex_code : nullptr;
        const int rc = mofa_internal_fold_bias(weights[li], l.n_out, l.ld, l.fold_col0, code ? l.fold_cols : 0,
                                               code ? code : biases[li], biases[li], folded + l.folded_off, l.n_padded,
                                               stream);
        if (rc != MOFA_OK) return rc;
    }
    return MOFA_OK;
}

size_t mofa_net_packed_t_floats(MofaNetShape s) { return shape_ok(s) ? make_plan(s).packed_t_floats : 0; }

size_t mofa_net_tape_floats(MofaNetShape s, int64_t n_points) {
    if (!shape_ok(s) || n_points <= 0) return 0;
    return (size_t)round_up(n_points, kRowTile) * make_plan(s).tape_cols;
}

size_t mofa_net_mask_tape_words(MofaNetShape s, int64_t n_points) {
    if (!shape_ok(s) || n_points <= 0) return 0;
    return (size_t)round_up(n_points, kRowTile) * make_plan(s).tape_cols / 64;     // one bit per tape float, in 64-bit words
}

size_t mofa_net_backward_workspace_floats(MofaNetShape s, int64_t n_points, int32_t with_weight_grads) {
    if (!shape_ok(s) || n_points <= 0) return 0;
    const Plan p = make_plan(s);
    const size_t mp = (size_t)round_up(n_points, kRowTile);
    // both forms: four gradient buffers + the encoding gradient + weight-gradient scratch (fitting: the row-split bias sums' partials live
    // there) + the queue state of two chained launches
    const size_t head = mp * (4 * (size_t)p.Wp + (size_t)p.pe_k) + 64;
    const size_t tail = 2 * (size_t)round_up((int64_t)mofa_internal_chain_state_words((long long)mp), 32) + 64;
    // training: the scratch holds the partial sums of every weight gradient of one chained launch (train_wws_floats); no further buffers
    if (with_weight_grads) return head + train_wws_floats(p, s.D, n_points) + 64 + tail;
    // fitting: three more gradient buffers — the bias-gradient inputs its chained backward keeps
    return head + mofa_weight_grad_workspace_floats(n_points, p.Wp, p.Wp) + 64 + 3 * mp * (size_t)p.Wp + tail;
}

int mofa_net_pack_t(MofaNetShape s, const float* const* weights, float* packed_t, void* stream) {
    MOFA_REQUIRE(shape_ok(s), "net_pack_t: unsupported shape " MOFA_SHAPE_FMT, MOFA_SHAPE_ARGS(s));
    MOFA_REQUIRE(weights && packed_t, "net_pack_t: null pointer");
    const Plan p = make_plan(s);
    for (size_t li = 0; li < p.L.size(); ++li) {
        const Layer& l = p.L[li];
        if (l.head) continue;
        MOFA_REQUIRE(weights[li], "net_pack_t: weights[%zu] is null", li);
        for (int part = 0; part < l.nsrc; ++part) {
            const int rc = mofa_pack_panels_t(weights[li], l.n_out, l.ld, l.col0[part], l.ncols[part],
                                              packed_t + l.packed_t_off[part], l.k_padded[part], l.n_padded, stream);
            if (rc != MOFA_OK) return rc;
        }
    }
    return MOFA_OK;
}

#define MOFA_TRY(expr) \
    if ((rc = (expr)) != MOFA_OK) return rc

// force_chain: -1 = by configuration + census (every caller but the self-check); 0 = the per-layer launches, 1 = the chained launch —
// whatever MOFA_* says and whatever the census found (mofa_device_init's self-check runs both forms on the same inputs)
static int net_forward(MofaNetShape s, const float* packed, const float* folded, const float* view_w,
                       const float* view_b, const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                       const float* pts, const float* viewdirs, int64_t n_rays, int32_t S, float* workspace,
                       float* raw_out, float* tape, uint64_t* mask_tape, const float* view_bias_rows, uint32_t* verdict, void* stream,
                       int force_chain) {
    MOFA_REQUIRE(shape_ok(s), "net_forward: unsupported shape " MOFA_SHAPE_FMT, MOFA_SHAPE_ARGS(s));
    MOFA_REQUIRE(packed && folded && workspace && raw_out, "net_forward: null pointer");
    MOFA_REQUIRE(!(tape && mask_tape), "net_forward: give the fp32 tape OR the mask-only tape, not both");
    MOFA_REQUIRE(view_bias_rows || (view_w && view_b && viewdirs),
                 "net_forward: need view_bias_rows or (view_w, view_b, viewdirs)");
    MOFA_REQUIRE(n_rays > 0 && S > 0, "net_forward: n_rays=%lld S=%d", (long long)n_rays, S);
    MOFA_REQUIRE(pts || (rays_o && rays_d && z), "net_forward: need pts or (rays_o, rays_d, z)");
    const Plan p = make_plan(s);
    const int64_t M = n_rays * S, Mp = round_up(M, kRowTile);
    const size_t act = (size_t)Mp * p.Wp;
    float* bufA = workspace;             // xyz_code, later rgbCodes
    float* bufB = workspace + act;       // sigmaCodes
    float* t0 = workspace + 2 * act;
    float* t1 = workspace + 3 * act;
    float* vbias = workspace + 4 * act;  // [n_rays, Hp]
    // with a tape every layer output is kept (training); otherwise 4 buffers are recycled — and with a mask-only tape (fitting)
    // every layer additionally leaves (output > 0) as one bit per activation (mofa_layer.h, mask_store_block)
    auto slot = [&](int li, float* fallback) -> float* {
        return tape ? tape + (size_t)Mp * p.L[li].tape_cols : fallback;
    };
    unsigned long long* const mbits = (unsigned long long*)mask_tape;
    auto mword = [&](int li) -> long long { return (long long)((size_t)Mp * p.L[li].tape_cols / 64); };   // 4 words per 256 floats
    auto mslot = [&](int li) -> uint64_t* { return mask_tape ? mask_tape + mword(li) : nullptr; };
    // ---- the MFMA layers in execution order: (layer, inputs, output) -------------------------------------------
    struct Step {
        int li;
        const float* x1;
        const float* x2;
        float* y;
    };
    std::vector<Step> steps;
    steps.reserve(p.L.size());
    float* y0 = slot(p.xyz0, t0);
    steps.push_back({p.xyz0, nullptr, nullptr, y0});                       // layer 0: positional encoding prologue
    float* y1 = slot(p.xyz0 + 1, t1);
    steps.push_back({p.xyz0 + 1, y0, nullptr, y1});
    float* y2 = slot(p.xyz0 + 2, t0);
    steps.push_back({p.xyz0 + 2, y1, nullptr, y2});
    float* xyz = slot(p.xyz0 + 3, bufA);
    steps.push_back({p.xyz0 + 3, y2, nullptr, xyz});
    // one conditioned skipMLP: x -> linears1 (5 layers) -> [x | h] -> linears2 (D-5 layers) -> out
    auto cond = [&](int first, int skip, const float* x, float* out_fb, float* pa, float* pb) -> float* {
        const float* cur = x;
        float* pp[2] = {pa, pb};
        int w = 0;
        for (int li = first; li < skip; ++li) {
            float* y = slot(li, pp[w]);
            steps.push_back({li, cur, nullptr, y});
            cur = y, w ^= 1;
        }
        const int last = skip + (s.D - 5) - 1;
        for (int li = skip; li <= last; ++li) {
            float* y = slot(li, (li == last) ? out_fb : pp[w]);
            steps.push_back({li, cur, li == skip ? x : nullptr, y});        // skip layer: [h | x] (see make_plan)
            cur = y, w ^= 1;
        }
        return const_cast<float*>(cur);
    };
    float* sigma = cond(p.bim0, p.bim_skip, xyz, bufB, t0, t1);
    float* rgbc = cond(p.uv0, p.uv_skip, sigma, bufA, t0, t1);  // without a tape rgbCodes reuses xyz_code's buffer
    float* v = slot(p.view, t0);
    steps.push_back({p.view, rgbc, nullptr, v});

    int rc;
    unsigned* chain_state = nullptr;      // set when the chained launch ran: its status words are verified behind the heads
    long long chain_tiles = 0;
    if (!view_bias_rows) {
        // per-ray bias b + W[:, :27] @ PE(viewdir) from the ORIGINAL (unpacked) view-layer tensors
        const Layer& l = p.L[p.view];
        MOFA_TRY(mofa_view_bias(viewdirs, n_rays, s.pe_view_freqs, view_w, l.n_out, l.ld, view_b, vbias, l.n_padded, stream));
        view_bias_rows = vbias;
    }
    // the chained launch takes a forward — inference, or keeping either tape — whose every layer fits the pipelined 128-feature tile, on a
    // device mofa_device_init() found eight populated XCDs on (no census yet: per-layer launches; nothing here allocates or synchronises).
    // (MOFA_CHAIN=0: per-layer launches — the bit-identical reference form; MOFA_PIPE=0 implies it)
    auto chain_ok = [&]() {
        if (force_chain == 0 || steps.size() > MOFA_MAX_CHAIN_STEPS) return false;
        if (force_chain < 0 && (config().chain == 0 || config().pipe == 0)) return false;
        for (const Step& st : steps) {
            const Layer& l = p.L[st.li];
            const int kt = l.k_padded[0] / 16 + (st.x2 ? l.k_padded[1] / 16 : 0);
            if (l.n_padded % 128 != 0 || kt < 4 || (kt & 1)) return false;
            if (!st.x1 && (l.n_padded < 512 || st.y == t1)) return false;     // layer 0 through k_pe_panels, as below
        }
        return force_chain == 1 || mofa_internal_chain_capable(stream) == 1;
    };
    // ---- dispatch: one persistent launch for widths <= 256 (every layer of a point tile lives in one workgroup); wider networks: one
    //      chained launch over the tiles of every layer (inference), else one launch per layer.  MOFA_FUSED=0/1 and MOFA_CHAIN=0 override
    //      the choice (tests / A-B).
    const Config& cfg = config();
    const bool fused = force_chain < 0 && (cfg.fused >= 0 ? cfg.fused == 1 : (p.Wp <= 256 && Mp / kRowTile >= 128));
    if (fused) {
        const float* arena = tape ? tape : workspace;
        const int n = (int)steps.size();
        std::vector<long long> x1(n), x2(n), yo(n), wo(n), bo(n), mo(n);
        std::vector<int> k1(n), k2(n), np(n), div(n);
        for (int i = 0; i < n; ++i) {
            const Layer& l = p.L[steps[i].li];
            x1[i] = steps[i].x1 ? steps[i].x1 - arena : -1;
            x2[i] = steps[i].x2 ? steps[i].x2 - arena : 0;
            yo[i] = steps[i].y - arena;
            wo[i] = (long long)l.packed_off;
            const bool view = steps[i].li == p.view;
            bo[i] = view ? 0 : (long long)l.folded_off;
            k1[i] = l.k_padded[0] / 16, k2[i] = steps[i].x2 ? l.k_padded[1] / 16 : 0;
            np[i] = l.n_padded, div[i] = view ? S : 0;
            mo[i] = mword(steps[i].li);
        }
        MOFA_TRY(mofa_internal_fused_forward(arena, const_cast<float*>(arena), packed, folded, view_bias_rows, n_rays, rays_o,
                                             rays_d, z, z_row_stride, pts, M, S, Mp, n, x1.data(), x2.data(), yo.data(),
                                             wo.data(), bo.data(), k1.data(), k2.data(), np.data(), div.data(), 3 + 6 * s.pe_point_freqs,
                                             mbits, mo.data(), stream));
        if (mask_tape) {  // the view layer's per-ray-bias epilogue is not the contiguous-store one: its bits come from a pass over its output
            MOFA_TRY(mofa_internal_mask_pack(v, (long long)Mp * p.L[p.view].n_padded, mbits + mword(p.view), stream));
        }
    } else if (chain_ok()) {
        // Wide network: ONE chained launch (k_net_chain, mofa_mlp.hip) over the tiles of every layer — the same tiles as the per-layer
        // launches below (bit-identical), without their ~26 launch boundaries per sub-batch.  Layer 0 reads the encoding panels
        // k_pe_panels leaves in t1 (without a tape layer 1 overwrites them row tile by row tile, after layer 0 is done with those rows).
        // With a tape the outputs simply are the tape's slots; with a mask tape the contiguous-store epilogues also write the bits
        // (the view layer's per-ray-bias epilogue does not: a pass over its output below, as with per-layer launches).
        MOFA_TRY(mofa_pe_panels(rays_o, rays_d, z, z_row_stride, pts, M, S, s.pe_point_freqs, Mp, t1, stream));
        std::vector<ChainStep> cs(steps.size());
        for (size_t i = 0; i < steps.size(); ++i) {
            const Layer& l = p.L[steps[i].li];
            const bool view = steps[i].li == p.view;
            ChainStep c{};
            c.x1 = steps[i].x1 ? steps[i].x1 : t1, c.x2 = steps[i].x2, c.y = steps[i].y;
            c.w = packed + l.packed_off, c.aux = view ? view_bias_rows : folded + l.folded_off;
            c.bits = (mask_tape && !view) ? mbits + mword(steps[i].li) : nullptr;
            c.k1p = l.k_padded[0] / 16, c.k2p = steps[i].x2 ? l.k_padded[1] / 16 : 0;
            c.n_padded = l.n_padded, c.bias_row_div = view ? S : 0, c.flags = 1;
            cs[i] = c;
        }
        chain_state = (unsigned*)(vbias + (size_t)n_rays * p.Hp + 64);
        MOFA_TRY(mofa_internal_chain_launch(mask_tape ? kChainForwardMask : kChainForward, cs.data(), (int)cs.size(), Mp, n_rays, chain_state,
                                            &chain_tiles, stream));
        if (mask_tape) MOFA_TRY(mofa_internal_mask_pack(v, (long long)Mp * p.L[p.view].n_padded, mbits + mword(p.view), stream));
    } else {
        for (const Step& st : steps) {
            const Layer& l = p.L[st.li];
            if (!st.x1 && l.n_padded % 128 == 0 && l.n_padded >= 512 && st.y != t1 && config().pipe != 0) {
                // Wide first layer: the encoding features of every point are computed ONCE into panels (t1 is free until layer 1
                // writes it) and the layer runs as an ordinary K = pe_k launch of the pipelined kernel.  The generated-operand kernel
                // (k_layer<.., L0>) re-derives them in each of the n_padded / 128 feature-tile workgroups of a point tile — 8 times
                // at width 1024, which bounded that launch at 62 TFLOP/s (414 us per 196,608 points).  Same
                // features (pe_feature's formula, separately rounded o + d z), same MFMA order, same epilogue: bit-identical for every
                // point row (the padding rows m >= n_points hold relu(bias) here and a copy of the last point there; no consumer reads
                // them: heads, compositing and every backward kernel stop at n_points).
                MOFA_TRY(mofa_pe_panels(rays_o, rays_d, z, z_row_stride, pts, M, S, s.pe_point_freqs, Mp, t1, stream));
                MOFA_TRY(mofa_layer_forward_masked(t1, l.k_padded[0], nullptr, 0, packed + l.packed_off, folded + l.folded_off, 0, 1, st.y, Mp,
                                            l.n_padded, 1, mslot(st.li), stream));
            } else if (!st.x1) {
                MOFA_TRY(mofa_layer0_forward(rays_o, rays_d, z, z_row_stride, pts, M, S, s.pe_point_freqs, packed + l.packed_off,
                                             folded + l.folded_off, st.y, Mp, l.n_padded, mslot(st.li), stream));
            } else if (st.li == p.view) {
                MOFA_TRY(mofa_layer_forward_masked(st.x1, l.k_padded[0], nullptr, 0, packed + l.packed_off, view_bias_rows, S,
                                            n_rays, st.y, Mp, l.n_padded, 1, mslot(st.li), stream));
            } else {
                MOFA_TRY(mofa_layer_forward_masked(st.x1, l.k_padded[0], st.x2, st.x2 ? l.k_padded[1] : 0, packed + l.packed_off,
                                            folded + l.folded_off, 0, 1, st.y, Mp, l.n_padded, 1, mslot(st.li), stream));
            }
        }
    }
    // ---- heads: sigma from sigmaCodes, rgb from the view layer's output ------------------------------------------
    {
        const Layer& l = p.L[p.alpha];
        auto head = mofa_head_forward;
        MOFA_TRY(head(sigma, l.k_padded[0], Mp, packed + l.packed_off, folded + l.folded_off, 1, raw_out, 3, M, stream));
        const Layer& r = p.L[p.rgb];
        MOFA_TRY(head(v, r.k_padded[0], Mp, packed + r.packed_off, folded + r.folded_off, 3, raw_out, 0, M, stream));
    }
    // a chained launch that did not finish every tile (a wait abandoned, an unworked queue) must not look like a result: NaN + verdict
    if (chain_state) MOFA_TRY(mofa_internal_chain_verify(chain_state, chain_tiles, verdict, raw_out, M * 4, nullptr, 0, nullptr, 0, nullptr, 0, stream));
    return MOFA_OK;
}

int mofa_net_forward(MofaNetShape s, const float* packed, const float* folded, const float* view_w,
                     const float* view_b, const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                     const float* pts, const float* viewdirs, int64_t n_rays, int32_t S, float* workspace,
                     float* raw_out, float* tape, uint64_t* mask_tape, const float* view_bias_rows, uint32_t* verdict, void* stream) {
    return net_forward(s, packed, folded, view_w, view_b, rays_o, rays_d, z, z_row_stride, pts, viewdirs, n_rays, S, workspace, raw_out, tape,
                       mask_tape, view_bias_rows, verdict, stream, -1);
}

// ---- the chained launch's self-check (mofa_device_init) ---------------------------------------------------------
// k_chain_verify sees an INCOMPLETE chained launch; it cannot see a launch that finished every tile on STALE inputs.  The protocol's
// visibility leg (plain stores + vmcnt(0) + barrier + a relaxed agent-scope counter bump; consumers fetch with sc1 LDS-DMA loads
// served by the XCD's L2) is a micro-architectural contract of this part, not a release / acquire pair the language guarantees — if a
// driver, firmware or partition mode ever breaks it, every tile finishes, the verdict is clean and the pixels are plausibly wrong.
// So the contract is CHECKED once per device and process, where the census is taken: a fixed 10 x 512 network on 4,096 points (16 row
// tiles — two per XCD, so nearly every tile waits on a dependency and fetches what another workgroup has just stored) runs twice
// chained and twice per layer on two point sets through the SAME recycled workspace (the second chained run reads buffers that hold
// the first run's different activations: a stale line cannot go unnoticed), and the raw outputs are compared bit for bit on the
// device.  Any difference, any NaN, a raised verdict: this device takes the per-layer launches (bit-identical, ~1 % slower) and
// mofa_device_init reports it.  Weights / biases / points are generated on the device (hashed uniform values, Xavier-scaled).
// internal (mofa_device_init): *ok = 1 the chained launch reproduces the per-layer launches bit for bit on this device, 0 = it does not
// (the text says how).  Allocates, launches, synchronises `stream`, frees.
int mofa_internal_chain_selfcheck(void* stream, int* ok, char* why, size_t why_len) {
    hipStream_t st = (hipStream_t)stream;
    const MofaNetShape s{10, 512, MOFA_DEFAULT_PE_POINT_FREQS, MOFA_DEFAULT_PE_VIEW_FREQS, MOFA_DEFAULT_CH_EXP, MOFA_DEFAULT_CH_SHAPE, MOFA_DEFAULT_CH_TEX};
    const Plan p = make_plan(s);
    const int64_t R = 32, S = 128, M = R * S;                      // 16 row tiles
    const size_t n_packed = (size_t)round_up((int64_t)p.packed_floats, 64), n_folded = (size_t)round_up((int64_t)p.folded_floats, 64);
    const size_t n_ws = (size_t)round_up((int64_t)mofa_net_workspace_floats(s, M, R), 64), n_vb = (size_t)R * p.Hp, n_pts = (size_t)M * 3, n_raw = (size_t)M * 4;
    const size_t total = n_packed + n_folded + n_ws + n_vb + 2 * n_pts + 4 * n_raw + 64;
    float* base = nullptr;
    *ok = 0;
    if (hipMalloc((void**)&base, total * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(why, why_len, "the self-check could not allocate %zu MiB", total * sizeof(float) >> 20);
        return MOFA_OK;                                              // (not an error of the device: the per-layer launches need nothing)
    }
    float* packed = base;
    float* folded = packed + n_packed;
    float* ws = folded + n_folded;
    float* vb = ws + n_ws;
    float* pts[2] = {vb + n_vb, vb + n_vb + n_pts};
    float* raw[4] = {pts[1] + n_pts, pts[1] + n_pts + n_raw, pts[1] + n_pts + 2 * n_raw, pts[1] + n_pts + 3 * n_raw};   // per-layer A, B; chained A, B
    unsigned* words = (unsigned*)(raw[3] + n_raw);                 // [0..7] a verdict, [8..15] compare A, [16..23] compare B
    auto fill = [&](float* q, size_t n, unsigned seed, float scale, float off) {
        hipLaunchKernelGGL(k_selfcheck_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q, (long long)n, seed, scale, off);
    };
    int rc = MOFA_OK;
    fill(packed, n_packed, 0x1234u, 0.108f, 0.f);                  // ~ Xavier-uniform with the ReLU gain at fan-in = fan-out = 512
    fill(folded, n_folded, 0x2345u, 0.05f, 0.05f);
    fill(vb, n_vb, 0x3456u, 0.05f, 0.05f);
    fill(pts[0], n_pts, 0x4567u, 3.0f, 0.f);
    fill(pts[1], n_pts, 0x5678u, 3.0f, 0.f);
    if (hipMemsetAsync(words, 0, 64 * sizeof(unsigned), st) != hipSuccess) rc = check_launch("hipMemsetAsync(self-check)");
    // per-layer A, chained B, chained A (its buffers hold B's activations), per-layer B
    const int order[4][3] = {{0, 0, 0}, {1, 1, 3}, {1, 0, 2}, {0, 1, 1}};   // {force_chain, point set, raw slot}
    for (int i = 0; i < 4 && rc == MOFA_OK; ++i)
        rc = net_forward(s, packed, folded, nullptr, nullptr, nullptr, nullptr, nullptr, 0, pts[order[i][1]], nullptr, R, (int32_t)S, ws,
                         raw[order[i][2]], nullptr, nullptr, vb, words, st, order[i][0]);
    if (rc == MOFA_OK) {
        if (hook_selfcheck_poison()) hipLaunchKernelGGL(k_selfcheck_poison, dim3(1), dim3(1), 0, st, (unsigned*)raw[2] + 5);
        const unsigned blocks = (unsigned)((n_raw + 255) / 256);
        hipLaunchKernelGGL(k_selfcheck_compare, dim3(blocks), dim3(256), 0, st, (const unsigned*)raw[2], (const unsigned*)raw[0], (long long)n_raw, words + 8);
        hipLaunchKernelGGL(k_selfcheck_compare, dim3(blocks), dim3(256), 0, st, (const unsigned*)raw[3], (const unsigned*)raw[1], (long long)n_raw, words + 16);
        rc = check_launch("k_selfcheck_compare");
    }
    unsigned host[24] = {};
    if (rc == MOFA_OK && (hipMemcpyAsync(host, words, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
        rc = check_launch("self-check read-back");
    else if (rc != MOFA_OK) (void)hipStreamSynchronize(st);
    (void)hipFree(base);
    if (rc != MOFA_OK) return rc;
    const unsigned diff = host[8] + host[16], nonfinite = host[9] + host[17], alive = host[10] < host[18] ? host[10] : host[18];
    if (host[0] != 0u || host[1] != 2u)
        snprintf(why, why_len, "the self-check's chained launches did not complete (verdict %u, %u of 2 verified)", host[0], host[1]);
    else if (nonfinite != 0u || alive < n_raw / 2)
        snprintf(why, why_len, "the self-check's network produced %u non-finite and %u non-zero of %zu outputs", nonfinite, alive, n_raw);
    else if (diff != 0u)
        snprintf(why, why_len, "the chained launch differs from the per-layer launches in %u of %zu output words (inter-workgroup visibility "
                               "through the XCD's L2 does not hold on this device)", diff, 2 * n_raw);
    else
        *ok = 1;
    return MOFA_OK;
}

int mofa_net_backward(MofaNetShape s, const float* packed, const float* packed_t, const float* tape, const uint64_t* mask_tape,
                      const float* d_raw, const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                      const float* pts, int64_t n_rays, int32_t S, float* workspace, float* d_folded,
                      float* d_view_bias_rows, float* d_rays_o, float* d_rays_d, float* d_pts, float* const* d_weights, uint32_t* verdict,
                      void* stream) {
    MOFA_REQUIRE(shape_ok(s), "net_backward: unsupported shape " MOFA_SHAPE_FMT, MOFA_SHAPE_ARGS(s));
    MOFA_REQUIRE(packed && packed_t && d_raw && workspace && d_folded && d_view_bias_rows, "net_backward: null pointer");
    MOFA_REQUIRE((tape != nullptr) != (mask_tape != nullptr), "net_backward: need the fp32 tape OR the mask-only tape");
    MOFA_REQUIRE(!(mask_tape && d_weights), "net_backward: weight gradients need the fp32 tape (the layer inputs), not the mask-only tape");
    MOFA_REQUIRE(pts ? d_pts != nullptr : (rays_o && rays_d && z && d_rays_o && d_rays_d),
                 "net_backward: need (pts, d_pts) or (rays_o, rays_d, z, d_rays_o, d_rays_d)");
    MOFA_REQUIRE(n_rays > 0 && S > 0, "net_backward: n_rays=%lld S=%d", (long long)n_rays, S);
    const Plan p = make_plan(s);
    const int64_t M = n_rays * S, Mp = round_up(M, kRowTile);
    const size_t act = (size_t)Mp * p.Wp;
    float* g0 = workspace;
    float* g1 = workspace + act;
    float* gS = workspace + 2 * act;   // accumulates d sigmaCodes
    float* gX = workspace + 3 * act;   // accumulates d xyz_code
    float* dpe = workspace + 4 * act;  // [Mp, pe_k]
    float* wws = dpe + (size_t)Mp * p.pe_k + 64;  // split-M partial sums of the weight-gradient GEMMs (training: of a whole chained launch)
    float* extra = wws + (d_weights ? train_wws_floats(p, s.D, M) : mofa_weight_grad_workspace_floats(M, p.Wp, p.Wp)) + 64;   // fitting: three more gradient buffers (chained form only)
    unsigned* cstate[2];        // queue state of the two chained launches, each on a 128-byte boundary of the workspace (the queue heads own a line each)
    cstate[0] = (unsigned*)(workspace + round_up((extra + (d_weights ? 0 : 3 * act)) - workspace, 32));
    cstate[1] = cstate[0] + round_up((int64_t)mofa_internal_chain_state_words((long long)Mp), 32);
    // (output > 0) of layer li: the saved fp32 activation itself, or its bits in the mask-only tape — never both
    struct Mask {
        const float* act;
        const uint64_t* bits;
    };
    const Mask kNoMask{nullptr, nullptr};
    auto T = [&](int li) -> const float* { return tape ? tape + (size_t)Mp * p.L[li].tape_cols : nullptr; };
    auto Mk = [&](int li) -> Mask {
        return tape ? Mask{T(li), nullptr} : Mask{nullptr, mask_tape + (size_t)Mp * p.L[li].tape_cols / 64};
    };
    int rc;
    const int n2 = s.D - 5;
    // ---- the chained form (fitting: no weight gradients).  The backward-data products of the wide network run as TWO chained launches
    // (k_net_chain<backward>: view layer + texture stack | shape stack + xyzEncode 3..1) instead of 26 per-layer launches — same tiles,
    // same epilogues, bit-identical.  What sits between the products in the per-layer form moves around them: the sigma head's
    // contribution (an elementwise accumulate into d sigmaCodes) separates the two launches, and the five bias-gradient column sums
    // the fitting needs (the code-conditioned layers) run after the launch that produced their input — which therefore must survive
    // the launch: those gradients are kept in buffers the chain does not recycle (three more than the per-layer form's four).
    // Training (weight gradients; round 6, opt-in MOFA_CHAIN_TRAIN=1): the same two launches as k_net_chain_train — every layer's gradient
    // feeds a weight-gradient GEMM between two products, so the weight gradients' units are queue entries too (wg_split's plan: the
    // per-layer kernel's own splits, so the partial sums are bit-identical); their partials live in the scratch until the second-stage
    // sums behind the launch.  Built, verified bit-identical, and measured 0.2-0.4 % slower than the per-layer launches (the queues mix
    // products and weight gradients in time: 2.6 TB/s of steady fabric traffic, waves parked 10.6 % against 3.4 / 4.4 %), so the default
    // training backward stays per-layer (profiles/r06_ab_chain_train.md).
    const bool chain_knobs = config().chain != 0 && config().pipe != 0;
    const bool chain = chain_knobs && mofa_internal_chain_capable(stream) == 1 &&
                       (d_weights ? train_chain_shape(p, s.D)
                                  : (p.Wp % 128 == 0 && p.Wp >= 64 && p.Hp % 32 == 0 && p.Hp >= 64 && n2 + 7 <= MOFA_MAX_CHAIN_STEPS && n2 + 9 <= MOFA_MAX_CHAIN_STEPS));
    bool chaining = false;                       // products are being recorded (between begin_chain() and flush())
    std::vector<ChainStep> seg;
    struct Deferred {
        int li;
        const float* g;
    };
    std::vector<Deferred> deferred;             // bias gradients whose input the open segment produces
    std::vector<const float*> kept;             // ... and the buffers they read: not to be overwritten before flush()
    struct Reduce {                             // (training) second-stage sums of the weight gradients the open segment's launch leaves as partials
        int li, part, splits;
        const float* partial;
    };
    std::vector<Reduce> reduces;
    float* wpart = wws;                         // next free partial region of the open segment
    long long ctiles[2] = {0, 0};
    int nseg = 0;
    float *cur = nullptr, *spare = nullptr;
    auto is_kept = [&](const float* b) {
        for (const float* k : kept)
            if (k == b) return true;
        return false;
    };
    // the next buffer a product may overwrite: in the chained form never one a deferred bias gradient still has to read
    auto fix_spare = [&]() {
        if (!chaining || (!is_kept(spare) && spare != cur)) return;
        float* cand[5] = {g0, g1, extra, extra + act, extra + 2 * act};
        for (float* c : cand)
            if (!is_kept(c) && c != cur) {
                spare = c;
                return;
            }
    };
    auto flush = [&]() -> int {
        chaining = false;
        if (seg.empty()) return MOFA_OK;
        MOFA_REQUIRE(nseg < 2, "net_backward: internal error (more than two chained segments)");
        if (d_weights) {
            MOFA_TRY(mofa_internal_chain_train_launch(seg.data(), (int)seg.size(), Mp, M, cstate[nseg], &ctiles[nseg], stream));
        } else {
            MOFA_TRY(mofa_internal_chain_launch(kChainBackward, seg.data(), (int)seg.size(), Mp, 1, cstate[nseg], &ctiles[nseg], stream));
        }
        ++nseg;
        for (const Deferred& d : deferred)
            MOFA_TRY(mofa_internal_bias_grad_split(d.g, Mp, M, p.L[d.li].n_padded, d_folded + p.L[d.li].folded_off, wws, stream));
        for (const Reduce& r : reduces) {       // the deterministic second stage, in the per-layer form's own order
            const Layer& l = p.L[r.li];
            MOFA_TRY(mofa_internal_wgrad_reduce(r.partial, r.splits, l.n_padded, l.k_padded[r.part], l.n_out, l.ncols[r.part], d_weights[r.li], l.ld,
                                                l.col0[r.part], (r.part == 0 && l.fold != kView) ? d_folded + l.folded_off : nullptr, stream));
        }
        seg.clear(), deferred.clear(), kept.clear(), reduces.clear();
        wpart = wws;
        return MOFA_OK;
    };
    // dW[li][:, col0[part] : +ncols[part]] = G^T X   (training only: d_weights != NULL; the constant columns are the host's)
    auto wgrad = [&](int li, int part, const float* g, const float* x) -> int {
        if (!d_weights) return MOFA_OK;
        const Layer& l = p.L[li];
        MOFA_REQUIRE(d_weights[li], "net_backward: d_weights[%d] is null", li);
        if (chaining) {                          // a queue entry of the open segment's launch; summed behind it (flush)
            const WgSplit sp = wg_split(Mp, (l.n_padded / 128) * (l.k_padded[part] / 256));
            const bool bias = part == 0 && l.fold != kView;
            ChainStep c{};
            c.x1 = g, c.x2 = x, c.y = wpart, c.aux = bias ? wpart + (size_t)sp.total * l.n_padded * l.k_padded[part] : nullptr;
            c.k1p = l.k_padded[part] / 16, c.n_padded = l.n_padded, c.flags = kChainStepWgrad, c.spt = sp.spt;
            seg.push_back(c);
            reduces.push_back({li, part, sp.total, wpart});
            wpart += train_partial_floats(Mp, l.n_padded, l.k_padded[part]);
            return MOFA_OK;
        }
        return mofa_weight_grad(g, l.n_padded, x, l.k_padded[part], Mp, M, l.n_out, l.ncols[part], d_weights[li], l.ld,
                                l.col0[part], (part == 0 && l.fold != kView) ? d_folded + l.folded_off : nullptr, wws,
                                stream);
    };
    // bias gradient of layer li from its masked output gradient g.  Training: produced by the weight-gradient kernel's
    // pass over G (part 0).  Fitting: only the five code-conditioned layers need it (the others' biases are not optimised;
    // their slots are zeroed below).
    auto bgrad = [&](int li, const float* g) -> int {
        if (d_weights || p.L[li].fold == kNone) return MOFA_OK;
        if (chaining) {
            deferred.push_back({li, g}), kept.push_back(g);
            return MOFA_OK;
        }
        // (row-split form: 64 workgroups — one per panel — are a quarter of the chip; the weight-gradient partials' region is free in fitting)
        return mofa_internal_bias_grad_split(g, Mp, M, p.L[li].n_padded, d_folded + p.L[li].folded_off, wws, stream);
    };
    // dX = G @ W[:, part]
    auto bdata = [&](int li, int part, const float* g, Mask mask, int accumulate, float* dx) -> int {
        const Layer& l = p.L[li];
        if (chaining) {
            ChainStep c{};
            c.x1 = g, c.y = dx, c.w = packed_t + l.packed_t_off[part], c.aux = mask.act, c.bits = (unsigned long long*)mask.bits;
            c.k1p = l.n_padded / 16, c.n_padded = l.k_padded[part], c.flags = accumulate ? 1 : 0;
            seg.push_back(c);
            return MOFA_OK;
        }
        if (mask.bits)
            return mofa_layer_backward_data_bits(g, l.n_padded, packed_t + l.packed_t_off[part], mask.bits, accumulate, dx, Mp,
                                                 l.k_padded[part], stream);
        return mofa_layer_backward_data(g, l.n_padded, packed_t + l.packed_t_off[part], mask.act, accumulate, dx, Mp, l.k_padded[part], stream);
    };
    auto hbwd = [&](const float* dr, int off, int n, const float* w, int kp, Mask mask, int accumulate, float* dx) -> int {
        if (mask.bits) return mofa_head_backward_bits(dr, off, n, w, kp, mask.bits, accumulate, dx, Mp, M, stream);
        return mofa_head_backward(dr, off, n, w, kp, mask.act, accumulate, dx, Mp, M, stream);
    };
    if (!d_weights && hipMemsetAsync(d_folded, 0, p.folded_floats * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return check_launch("hipMemsetAsync(d_folded)");
    // heads' bias gradients
    MOFA_TRY(mofa_internal_raw_colsum(d_raw, M, d_folded + p.L[p.rgb].folded_off, d_folded + p.L[p.alpha].folded_off, wws, stream));
    // rgb head -> gradient at the view layer's output (masked by its ReLU); its per-ray sums are d(view bias rows)
    {
        const Layer& r = p.L[p.rgb];
        MOFA_TRY(hbwd(d_raw, 0, 3, packed + r.packed_off, r.k_padded[0], Mk(p.view), 0, g0));
        MOFA_TRY(mofa_bias_grad_rays(g0, Mp, n_rays, S, p.L[p.view].n_padded, d_view_bias_rows, stream));
        if (d_weights) {
            MOFA_REQUIRE(d_weights[p.rgb] && d_weights[p.alpha], "net_backward: head d_weights are null");
            MOFA_TRY(mofa_internal_head_weight_grad_split(d_raw, 0, 3, T(p.view), r.k_padded[0], Mp, M, r.ld, d_weights[p.rgb], r.ld, wws,
                                                          stream));
            const Layer& a = p.L[p.alpha];
            MOFA_TRY(mofa_internal_head_weight_grad_split(d_raw, 3, 1, T(p.bim_skip + n2 - 1), a.k_padded[0], Mp, M, a.ld,
                                                          d_weights[p.alpha], a.ld, wws, stream));
        }
    }
    chaining = chain;                                            // ---- first chained launch: view layer + texture stack
    MOFA_TRY(wgrad(p.view, 0, g0, T(p.uv_skip + n2 - 1)));
    // view layer -> d rgbCodes, masked by the last uv layer's ReLU
    MOFA_TRY(bdata(p.view, 0, g0, Mk(p.uv_skip + n2 - 1), 0, g1));
    cur = g1, spare = g0;
    // One conditioned stack, walked backwards.  `cur` = masked gradient at its output.  The gradient w.r.t. the stack's
    // input x has two contributions (the skip concat and linears1.Linear0): the first overwrites gx, the second
    // accumulates and applies `final_mask` (the ReLU of the layer that produced x) if given.
    auto stack_bwd = [&](int first, int skip, const float* xin, float* gx, Mask final_mask) -> int {
        const int last = skip + n2 - 1;
        for (int li = last; li > skip; --li) {
            MOFA_TRY(bgrad(li, cur));
            MOFA_TRY(wgrad(li, 0, cur, T(li - 1)));
            MOFA_TRY(bdata(li, 0, cur, Mk(li - 1), 0, spare));
            std::swap(cur, spare);
            fix_spare();
        }
        MOFA_TRY(bgrad(skip, cur));
        MOFA_TRY(wgrad(skip, 0, cur, T(skip - 1)));                  // part 0 = the h columns, part 1 = the x columns (make_plan)
        MOFA_TRY(wgrad(skip, 1, cur, xin));
        MOFA_TRY(bdata(skip, 1, cur, kNoMask, 0, gx));               // x part of [x | h]
        MOFA_TRY(bdata(skip, 0, cur, Mk(skip - 1), 0, spare));       // h part, masked by linears1's last ReLU
        std::swap(cur, spare);
        fix_spare();
        for (int li = skip - 1; li > first; --li) {
            MOFA_TRY(bgrad(li, cur));
            MOFA_TRY(wgrad(li, 0, cur, T(li - 1)));
            MOFA_TRY(bdata(li, 0, cur, Mk(li - 1), 0, spare));
            std::swap(cur, spare);
            fix_spare();
        }
        MOFA_TRY(bgrad(first, cur));
        MOFA_TRY(wgrad(first, 0, cur, xin));
        MOFA_TRY(bdata(first, 0, cur, final_mask, 1, gx));
        return MOFA_OK;
    };
    // uv stack (input sigmaCodes); the sigma head adds the third contribution and applies the bim stack's last ReLU
    MOFA_TRY(stack_bwd(p.uv0, p.uv_skip, T(p.bim_skip + n2 - 1), gS, kNoMask));
    MOFA_TRY(flush());
    {
        const Layer& a = p.L[p.alpha];
        MOFA_TRY(hbwd(d_raw, 3, 1, packed + a.packed_off, a.k_padded[0], Mk(p.bim_skip + n2 - 1), 1, gS));
    }
    chaining = chain;                                            // ---- second chained launch: shape stack + xyzEncode Linear3..1
    // bim stack (input xyz_code)
    cur = gS, spare = g0;
    MOFA_TRY(stack_bwd(p.bim0, p.bim_skip, T(p.xyz0 + 3), gX, Mk(p.xyz0 + 3)));
    // xyzEncode Linear3..1, then Linear0 -> gradient w.r.t. the encoding features -> rays
    cur = gX, spare = g0;
    fix_spare();
    for (int li = p.xyz0 + 3; li > p.xyz0; --li) {
        MOFA_TRY(bgrad(li, cur));
        MOFA_TRY(wgrad(li, 0, cur, T(li - 1)));
        MOFA_TRY(bdata(li, 0, cur, Mk(li - 1), 0, spare));
        std::swap(cur, spare);
        if (spare == gX) spare = g1;
        fix_spare();
    }
    MOFA_TRY(flush());                                            // (Linear0's product has a 64-wide output: its own launch)
    MOFA_TRY(bgrad(p.xyz0, cur));
    if (d_weights) {   // layer 0's input is the positional encoding itself: regenerate it as panels, then reuse the buffer
        MOFA_TRY(mofa_pe_panels(rays_o, rays_d, z, z_row_stride, pts, M, S, s.pe_point_freqs, Mp, dpe, stream));
        MOFA_TRY(wgrad(p.xyz0, 0, cur, dpe));
    }
    MOFA_TRY(bdata(p.xyz0, 0, cur, kNoMask, 0, dpe));
    if (pts) {
        MOFA_TRY(mofa_pe_backward_points(dpe, Mp, pts, M, s.pe_point_freqs, d_pts, stream));
    } else {
        MOFA_TRY(mofa_pe_backward(dpe, Mp, rays_o, rays_d, z, z_row_stride, n_rays, S, s.pe_point_freqs, d_rays_o, d_rays_d, stream));
    }
    // chained launches that did not finish every tile must not look like gradients: NaN into everything this call returns + verdict
    for (int i = 0; i < nseg; ++i) {
        MOFA_TRY(mofa_internal_chain_verify(cstate[i], ctiles[i], verdict, d_folded, (long long)p.folded_floats, d_view_bias_rows,
                                            (long long)n_rays * p.L[p.view].n_padded, pts ? d_pts : d_rays_o, pts ? M * 3 : n_rays * 3,
                                            pts ? nullptr : d_rays_d, n_rays * 3, stream));
        if (d_weights) {                         // ... and into every weight gradient (their second-stage sums read the launch's partials)
            std::vector<float*> wp(p.L.size());
            std::vector<long long> wn(p.L.size());
            for (size_t li = 0; li < p.L.size(); ++li) wp[li] = d_weights[li], wn[li] = (long long)p.L[li].n_out * p.L[li].ld;
            MOFA_TRY(mofa_internal_chain_poison(cstate[i], ctiles[i], wp.data(), wn.data(), (int)wp.size(), stream));
        }
    }
    return MOFA_OK;
}
#undef MOFA_TRY

}  // extern "C"
