// Host-side driver of the conditioned MLP: layer plan, weight packing, per-call bias folding and the
// launch sequence of one run_network call (models/render_class.py:69-94 + models/model.py:121-137).
// Pure launch code: no allocation, no host synchronisation, every buffer is the caller's.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "mofa_common.h"

extern "C" {
int mofa_internal_fold_bias(const float* w, int n_out, int ld, int col0, int ncols, const float* code,
                            const float* bias, float* out, int n_padded, void* stream);
int mofa_internal_dense_rows(const float* w, int n_out, int ld, int col0, int ncols, float* dst, int k_padded,
                             void* stream);
}

namespace mofa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
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
};

struct Plan {
    int D, W, Wp, Hp;
    std::vector<Layer> L;
    size_t packed_floats = 0, folded_floats = 0;
    // indices into L
    int xyz0, bim0, bim_skip, uv0, uv_skip, view, alpha, rgb;
};

Plan make_plan(MofaNetShape s) {
    Plan p;
    p.D = s.D, p.W = s.W;
    const int W = s.W, Wp = (int)round_up(W, 64), Hp = (int)round_up(W / 2, 64);
    p.Wp = Wp, p.Hp = Hp;
    const int PE = 3 + 6 * MOFA_PE_POINT_FREQS, PV = 3 + 6 * MOFA_PE_VIEW_FREQS;
    auto plain = [&](int n_out, int ld) {
        Layer l{};
        l.n_out = n_out, l.ld = ld, l.nsrc = 1, l.col0[0] = 0, l.ncols[0] = ld, l.fold = kNone;
        l.n_padded = (int)round_up(n_out, 64), l.k_padded[0] = (int)round_up(ld, 64), l.head = false;
        return l;
    };
    // xyzEncode: skipMLP(D=3, skip=None) -> Linear0..3 (models/model.py:97, :220-223)
    {
        Layer l = plain(W, PE + MOFA_CH_EXP);
        l.ncols[0] = PE, l.k_padded[0] = 64, l.fold = kExp, l.fold_col0 = PE, l.fold_cols = MOFA_CH_EXP;
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
        Layer ls = plain(W, cin + 2 * W);  // input [code | x | h]  (models/model.py:215,229)
        ls.nsrc = 2, ls.col0[0] = cin, ls.ncols[0] = W, ls.col0[1] = cin + W, ls.ncols[1] = W;
        ls.k_padded[0] = Wp, ls.k_padded[1] = Wp, ls.fold = fold, ls.fold_col0 = 0, ls.fold_cols = cin;
        skip = (int)p.L.size();
        p.L.push_back(ls);
        for (int i = 1; i < s.D - 5; ++i) p.L.push_back(plain(W, W));
    };
    cond_stack(MOFA_CH_SHAPE, kShape, p.bim0, p.bim_skip);
    cond_stack(MOFA_CH_TEX, kTex, p.uv0, p.uv_skip);
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
    }
    return p;
}

bool shape_ok(MofaNetShape s) { return s.D >= 6 && s.D <= 64 && s.W >= 2 && s.W <= 8192 && s.W % 2 == 0; }

}  // namespace
}  // namespace mofa

using namespace mofa;

extern "C" {

int mofa_abi_version(void) { return MOFA_ABI_VERSION; }
const char* mofa_last_error(void) { return g_err; }

int mofa_net_num_layers(MofaNetShape s) { return shape_ok(s) ? 2 * s.D + 7 : MOFA_EINVAL; }
size_t mofa_net_packed_floats(MofaNetShape s) { return shape_ok(s) ? make_plan(s).packed_floats : 0; }
size_t mofa_net_folded_floats(MofaNetShape s) { return shape_ok(s) ? make_plan(s).folded_floats : 0; }

size_t mofa_net_workspace_floats(MofaNetShape s, int64_t n_points, int64_t n_rays) {
    if (!shape_ok(s) || n_points <= 0 || n_rays <= 0) return 0;
    const Plan p = make_plan(s);
    const size_t mp = (size_t)round_up(n_points, kRowTile);
    return 4 * mp * (size_t)p.Wp + (size_t)n_rays * (size_t)p.Hp + 64;
}

int mofa_net_pack(MofaNetShape s, const float* const* weights, float* packed, void* stream) {
    MOFA_REQUIRE(shape_ok(s), "net_pack: unsupported shape D=%d W=%d", s.D, s.W);
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
    MOFA_REQUIRE(shape_ok(s), "net_fold: unsupported shape D=%d W=%d", s.D, s.W);
    MOFA_REQUIRE(weights && biases && exp_code && shape_code && tex_code && folded, "net_fold: null pointer");
    const Plan p = make_plan(s);
    for (size_t li = 0; li < p.L.size(); ++li) {
        const Layer& l = p.L[li];
        if (l.fold == kView) continue;  // per-ray: mofa_view_bias inside mofa_net_forward
        MOFA_REQUIRE(weights[li] && biases[li], "net_fold: layer %zu has a null weight/bias", li);
        const float* code = l.fold == kExp ? exp_code : l.fold == kShape ? shape_code : l.fold == kTex ? tex_code : nullptr;
        const int rc = mofa_internal_fold_bias(weights[li], l.n_out, l.ld, l.fold_col0, code ? l.fold_cols : 0,
                                               code ? code : exp_code, biases[li], folded + l.folded_off, l.n_padded,
                                               stream);
        if (rc != MOFA_OK) return rc;
    }
    return MOFA_OK;
}

int mofa_net_forward(MofaNetShape s, const float* packed, const float* folded, const float* view_w,
                     const float* view_b, const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                     const float* pts, const float* viewdirs, int64_t n_rays, int32_t S, float* workspace,
                     float* raw_out, void* stream) {
    MOFA_REQUIRE(shape_ok(s), "net_forward: unsupported shape D=%d W=%d", s.D, s.W);
    MOFA_REQUIRE(packed && folded && view_w && view_b && viewdirs && workspace && raw_out, "net_forward: null pointer");
    MOFA_REQUIRE(n_rays > 0 && S > 0, "net_forward: n_rays=%lld S=%d", (long long)n_rays, S);
    MOFA_REQUIRE(pts || (rays_o && rays_d && z), "net_forward: need pts or (rays_o, rays_d, z)");
    const Plan p = make_plan(s);
    const int64_t M = n_rays * S, Mp = round_up(M, kRowTile);
    const size_t act = (size_t)Mp * p.Wp;
    float* bufA = workspace;            // xyz_code, later reused
    float* bufB = workspace + act;      // sigmaCodes
    float* t0 = workspace + 2 * act;
    float* t1 = workspace + 3 * act;
    float* vbias = workspace + 4 * act;  // [n_rays, Hp]

    auto run = [&](int li, const float* x1, const float* x2, float* y) -> int {
        const Layer& l = p.L[li];
        return mofa_layer_forward(x1, l.k_padded[0], x2, x2 ? l.k_padded[1] : 0, packed + l.packed_off,
                                  folded + l.folded_off, 0, 1, y, Mp, l.n_padded, 1, stream);
    };
    int rc;
#define MOFA_TRY(expr) \
    if ((rc = (expr)) != MOFA_OK) return rc
    // xyzEncode
    {
        const Layer& l = p.L[p.xyz0];
        MOFA_TRY(mofa_layer0_forward(rays_o, rays_d, z, z_row_stride, pts, M, S, packed + l.packed_off,
                                     folded + l.folded_off, t0, Mp, l.n_padded, stream));
        MOFA_TRY(run(p.xyz0 + 1, t0, nullptr, t1));
        MOFA_TRY(run(p.xyz0 + 2, t1, nullptr, t0));
        MOFA_TRY(run(p.xyz0 + 3, t0, nullptr, bufA));
    }
    // one conditioned skipMLP: x -> linears1 (5 layers) -> [x | h] -> linears2 (D-5 layers) -> out
    auto cond = [&](int first, int skip, const float* x, float* out, float* pa, float* pb) -> int {
        const float* cur = x;
        float* pp[2] = {pa, pb};
        int w = 0;
        for (int li = first; li < skip; ++li) {
            MOFA_TRY(run(li, cur, nullptr, pp[w]));
            cur = pp[w], w ^= 1;
        }
        const int last = skip + (s.D - 5) - 1;
        for (int li = skip; li <= last; ++li) {
            float* y = (li == last) ? out : pp[w];
            MOFA_TRY(run(li, li == skip ? x : cur, li == skip ? cur : nullptr, y));
            cur = y, w ^= 1;
        }
        return MOFA_OK;
    };
    MOFA_TRY(cond(p.bim0, p.bim_skip, bufA, bufB, t0, t1));
    {
        const Layer& l = p.L[p.alpha];
        MOFA_TRY(mofa_head_forward(bufB, l.k_padded[0], Mp, packed + l.packed_off, folded + l.folded_off, 1, raw_out,
                                   3, M, stream));
    }
    MOFA_TRY(cond(p.uv0, p.uv_skip, bufB, bufA, t0, t1));  // rgbCodes -> bufA (xyz_code is dead by now)
    {
        const Layer& l = p.L[p.view];
        // per-ray bias b + W[:, :27] @ PE(viewdir) from the ORIGINAL (unpacked) view-layer tensors
        MOFA_TRY(mofa_view_bias(viewdirs, n_rays, view_w, l.n_out, l.ld, view_b, vbias, l.n_padded, stream));
        MOFA_TRY(mofa_layer_forward(bufA, l.k_padded[0], nullptr, 0, packed + l.packed_off, vbias, S, n_rays, t0, Mp,
                                    l.n_padded, 1, stream));
        const Layer& r = p.L[p.rgb];
        MOFA_TRY(mofa_head_forward(t0, r.k_padded[0], Mp, packed + r.packed_off, folded + r.folded_off, 3, raw_out, 0,
                                   M, stream));
    }
#undef MOFA_TRY
    return MOFA_OK;
}

}  // extern "C"
