/*
 * mofanerf_hip.h — C ABI of libmofanerf_hip.so, the MI355X (gfx950) implementation of MoFaNeRF's
 * ray-marching hot path.
 *
 * The reference (zhuhao-nju/mofanerf) is pure Python/PyTorch with no FFI layer; the drop-in boundary
 * is the Python class `myRenderer` (models/render_class.py:40).  This header is what that class's
 * methods bind instead of aten ops.  Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated; the caller owns every buffer
 *     (PyTorch's caching allocator in the shipped host layer); nothing here allocates or frees.
 *   - `stream` is a hipStream_t passed as void*; no entry point synchronises the host — with ONE exception, mofa_device_init(), the
 *     explicit per-device initialisation (and mofa_prof_end(), which collects a measurement).
 *   - return 0 on success, MOFA_EINVAL for a bad argument, MOFA_EHIP if a launch failed
 *     (hipGetLastError text via mofa_last_error()).
 *   - kernels are stateless and re-entrant per stream.
 *
 * Panel layout ("panels").  Activations and the per-point weight blocks are stored K-panel-major so
 * that one MFMA operand tile is a single contiguous, already bank-swizzled LDS image:
 *     a matrix [rows, K] (K padded to a multiple of 16) is K/16 panels, each [rows][16] floats;
 *     element (row, k) lives at
 *         (k/16)*rows*16 + row*16 + ((((k%16)/4) ^ ((row/4)%4)) * 4) + k%4 .
 * `rows` is the point count padded to 256 for activations, the feature count padded to 64 for weights.
 */
#ifndef MOFANERF_HIP_H
#define MOFANERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden; everything declared in this header — and nothing else — is exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define MOFA_ABI_VERSION 5 /* 2: MofaNetShape carries the encoding / code widths; explicit-point backward; mask tape; no split modes.
                             3: MOFA_PROF_KINDS = 6 (mofa_prof_end's arrays grew by the chained kernel's entry); larger mofa_net_workspace_floats
                             4: mofa_device_init(); `verdict` words of mofa_net_forward / mofa_net_backward; MOFA_PROF_KINDS = 7; larger
                                mofa_net_backward_workspace_floats
                             5: mofa_device_init() runs the chained launch's self-check (third argument); mofa_test_hooks() replaces the two
                                MOFA_CHAIN_* environment hooks; MOFA_PROF_KINDS = 12 (the mask-writing chained forward, the chained training backward and the HBM-bound ray
                                kernels have their own entries); mofa_net_backward_workspace_floats(.., with_weight_grads) */
#define MOFA_OK 0
#define MOFA_EINVAL (-1)
#define MOFA_EHIP (-2)

/* the shipped configuration (configs/exp_mofanerf.txt over tools/config_parser.py:51-56,113-118) — DEFAULTS, not limits:
 * every one of them is a field of MofaNetShape below */
#define MOFA_DEFAULT_PE_POINT_FREQS 10 /* multires       -> 3 + 6*10 = 63 features */
#define MOFA_DEFAULT_PE_VIEW_FREQS 4   /* multires_views -> 3 + 6*4  = 27 features */
#define MOFA_DEFAULT_CH_EXP 30         /* input_ch_expCodes     */
#define MOFA_DEFAULT_CH_SHAPE 50       /* input_ch_shapeCodes   */
#define MOFA_DEFAULT_CH_TEX 256        /* input_ch_textureCodes */
#define MOFA_MAX_PE_FREQS 16           /* 3 + 6*16 = 99 features */
#define MOFA_MAX_CODE 4096
#define MOFA_ROW_TILE 256 /* activation rows are padded to this */

int mofa_abi_version(void);
const char* mofa_last_error(void);   /* thread-local text of the calling thread's last failure */

/* Library-wide state is limited to what is listed here; everything else is in caller-owned buffers.
 *   - four run-time knobs, each choosing between BIT-IDENTICAL forms of the exact-fp32 path — MOFA_PIPE=0 (plain instead of
 *     software-pipelined K loops), MOFA_FUSED=0/1 (per-layer launches / persistent network kernel for widths <= 256), MOFA_CHAIN=0
 *     (per-layer launches instead of the chained launch of the wider networks), MOFA_CHAIN_TRAIN=1 (the TRAINING backward — products and
 *     weight gradients — as chained launches too; off by default: measured a tie); there is no reduced-precision
 *     mode: read from the environment ONCE when the library is loaded into an immutable snapshot;
 *     no launch path calls getenv.  mofa_config_reload() re-reads them (tests that change a knob inside one process call it explicitly).
 *     Measurement arms (scheduling experiments, time stamps, ablations) are NOT in this library: csrc/measure/, tools/build_measure.py.
 *   - per-device caches (CU count, a one-time function attribute) and the per-device measurement session below. */
int mofa_config_reload(void);

/* TEST HOOKS — for tests/ and tools/ only; the shipped host layer never calls this, and NOTHING in the environment reaches these
 * settings (round 5 read two of them from MOFA_CHAIN_* variables: a stray variable in production turned every wide-network launch into
 * NaN + MofaError).  Process-wide, effective from the next launch / the next mofa_device_init():
 *   chain_spin_limit  polls before a dependency wait of the chained launch (k_net_chain) gives up; 0 = the shipped budget (2^22 polls,
 *                     seconds).  1 forces the "wait out of budget" path: the launch ends incomplete -> NaN outputs + verdict.
 *   chain_skip_xcd    -1 (none), or 0..7: the chained launch's workgroups on that XCD leave at once — what a CU-masked stream that
 *                     starves an XCD looks like (an unworked tile queue -> NaN outputs + verdict "tiles missing").
 *   selfcheck_poison  non-zero: mofa_device_init()'s self-check sees one flipped bit in the chained result (forces its fallback). */
int mofa_test_hooks(uint32_t chain_spin_limit, int32_t chain_skip_xcd, int32_t selfcheck_poison);

/* Per-device initialisation — the ONE entry point that allocates (scratch of the checks below, ~70 MB, freed again) and synchronises
 * `stream`.  Call it once per device before the first mofa_net_forward (the shipped host layer does, when a network is bound to a
 * device).  It decides whether the wide networks of this device may take the chained launch (k_net_chain) — two checks, both needed:
 *   1. the XCD census: one tile queue per XCD, so all eight must receive workgroups of a 2-per-CU launch.
 *      xcd_workgroups: NULL, or 8 ints receiving the census.
 *   2. the SELF-CHECK (ABI 5): the chained launch replaces launch boundaries by an inter-workgroup protocol whose visibility leg
 *      (a producer's plain stores, acknowledged by the XCD's L2, are what the consumer's `sc1` loads return) is a property of this
 *      part, not a language guarantee — and the verification kernel behind every chained launch sees incompleteness, not staleness.
 *      So it is checked here: a fixed 10 x 512 network on 4,096 points (16 row tiles: nearly every tile waits on a dependency) runs
 *      twice chained and twice per layer through one recycled workspace and the outputs are compared bit for bit on the device
 *      (~4 ms).  chain_selfcheck: NULL, or receives 1 (identical), 0 (any difference / NaN / incomplete launch: this device takes the
 *      per-layer launches, mofa_last_error() says why), -1 (not run: the census already said no).
 * It also sets the persistent kernel's LDS attribute.  Returns MOFA_OK whenever the device is usable — a failed check is a fallback
 * (per-layer launches: bit-identical, ~1 % slower), not an error.  Without this call the wide networks run per layer too;
 * mofa_net_forward / mofa_net_backward themselves never allocate or synchronise. */
int mofa_device_init(void* stream, int32_t* xcd_workgroups, int32_t* chain_selfcheck);

/* Launch verdicts.  A chained launch (k_net_chain) replaces launch boundaries by an inter-workgroup protocol; if that protocol ever
 * fails — a dependency wait out of budget, a tile queue nobody worked (CU-masked stream) — the kernel does NOT compute on incomplete
 * inputs: the launch ends incomplete, a verification kernel behind it overwrites the call's outputs (raw_out; the gradients) with NaN
 * and raises these words.  `verdict`: NULL, or MOFA_VERDICT_WORDS uint32 on the device, zeroed ONCE by the caller and then sticky:
 *   [0] bit 0 = a wait timed out, bit 1 = tiles missing (0 = every chained launch so far was complete)
 *   [1] chained launches verified   [2] [3] [4] the last launch's flags / finished tiles / expected tiles   [5] bad launches.
 * The host reads them back whenever it likes (asynchronously in the shipped layer) — [0] != 0 is an error, never a result. */
#define MOFA_VERDICT_WORDS 8

/* ---- network description -------------------------------------------------------------------
 * NeRF(D, W, input_ch, input_ch_views, input_ch_textureCodes, input_ch_shapeCodes, use_viewdirs=True, skips=[4]) of
 * models/model.py:80-137 as tools/create_model_condition.py:16-34 builds it from the flags of tools/config_parser.py:51-56,113-118:
 *   input_ch       = (3 + 6*multires) + input_ch_expCodes      (get_embedder, models/model.py:48-63; i_embed = -1: multires -> 0)
 *   input_ch_views =  3 + 6*multires_views
 * `weights`/`biases` are the 2D+7 Linear layers in state-dict order (mofanerf_amd/schema.py::nerf_layers):
 *   xyzEncode.linears1.Linear0..3, linear_BiM_xyz.linears1.Linear0..4, .linears2.Linear0..D-6,
 *   linear_uv_xyzBiM.linears1.Linear0..4, .linears2.Linear0..D-6, linear_view_xyBMuv.0,
 *   alpha_linear.0, rgb_linear — each weight row-major [out, in] exactly as PyTorch stores it; mofa_net_layer_dims() states the
 * [out, in] every entry point below assumes for layer li, so a host layer can REFUSE a module that does not match. */
typedef struct MofaNetShape {
    int32_t D;              /* netdepth  (8 coarse / 10 fine)   */
    int32_t W;              /* netwidth  (256 coarse / 1024 fine) */
    int32_t pe_point_freqs; /* multires       (0 .. MOFA_MAX_PE_FREQS; 0 = the raw coordinates only, i_embed = -1) */
    int32_t pe_view_freqs;  /* multires_views (same range) */
    int32_t ch_exp;         /* expression-code columns behind the point encoding in xyzEncode.Linear0 (0 .. MOFA_MAX_CODE) */
    int32_t ch_shape;       /* shape-code columns in front of linear_BiM_xyz.linears{1,2}.Linear0 */
    int32_t ch_tex;         /* texture-code columns in front of linear_uv_xyzBiM.linears{1,2}.Linear0 */
} MofaNetShape;

int mofa_net_num_layers(MofaNetShape s);          /* 2D+7, or MOFA_EINVAL for an unsupported shape */
int mofa_net_layer_dims(MofaNetShape s, int32_t li, int32_t* n_out, int32_t* n_in);   /* PyTorch weight shape of layer li */
int mofa_pe_k_padded(int32_t n_freqs);            /* roundup(3 + 6*n_freqs, 64): K of the first layer's operand panels */
size_t mofa_net_packed_floats(MofaNetShape s);    /* size of the packed per-point weight blob */
size_t mofa_net_folded_floats(MofaNetShape s);    /* size of the per-call folded-bias blob    */
size_t mofa_net_workspace_floats(MofaNetShape s, int64_t n_points, int64_t n_rays);

/* Repack the per-point-varying weight columns of every layer into panels (+ dense head rows).
 * Replaces nothing in the reference (it multiplies the concatenated inputs, model.py:129-133); the
 * split is exact algebra — see SURVEY.md §7 "constant folding". */
int mofa_net_pack(MofaNetShape s, const float* const* weights, float* packed, void* stream);

/* Per-call folded biases: b' = b + W[:, const cols] @ code for the five conditioned layers
 * (expression [ch_exp] -> xyzEncode.L0; shape [ch_shape] -> BiM l1.L0 / l2.L0; texture [ch_tex] -> uv l1.L0 / l2.L0),
 * plus plain copies of every other bias.  Replaces the torch.cat of expanded codes in
 * render_class.py:74-85,104 and model.py:129,132.   exp_code is the ALREADY modulated code
 * (scale*sigma+bias, render_class.py:81).  A code of width 0 may be NULL. */
int mofa_net_fold(MofaNetShape s, const float* const* weights, const float* const* biases,
                  const float* exp_code, const float* shape_code, const float* tex_code, float* folded,
                  void* stream);

/* run_network + NeRF.forward for n_rays*S points (render_class.py:69-94, model.py:121-137):
 * positional encoding of pts = o + d*z (separately rounded mul and add, render_class.py:315),
 * 2D+5 fused Linear+bias+ReLU layers on MFMA, per-ray view-direction bias, sigma/rgb heads.
 *   rays_o, rays_d, viewdirs [n_rays,3]; z [n_rays,S] (z_row_stride = S) or one shared row (stride 0)
 *   pts: optional explicit [n_rays*S,3] points (then rays_o/rays_d/z may be NULL)
 *   view_w [W/2, (3+6*multires_views)+W], view_b [W/2]: the ORIGINAL linear_view_xyBMuv.0 tensors (their view-encoding
 *   columns become a per-ray bias, computed here from viewdirs)
 *   raw_out [n_rays,S,4] = (rgb pre-sigmoid, sigma pre-ReLU)
 *   workspace: mofa_net_workspace_floats(s, n_rays*S, n_rays) floats.
 *   tape: NULL, or mofa_net_tape_floats() floats that receive EVERY layer's output for mofa_net_backward WITH weight gradients
 *         (training; sized for 288 GB of HBM: no recomputation).  tape == mask_tape == NULL: inference, 4 recycled buffers.
 *   mask_tape: NULL, or mofa_net_mask_tape_words() 64-bit words that receive ONE BIT per layer output, (output > 0) — all the
 *         backward needs when no weight gradient is asked for (fitting: run_fit.py:305-313 never steps the networks).  The
 *         activations themselves are recycled as in inference: 1/32 of the tape, no recomputation.  Excludes `tape`.
 *   view_bias_rows: NULL (computed here from viewdirs), or caller-provided per-ray bias rows [n_rays, roundup(W/2,64)]
 *         (the autograd path computes them on the host so that gradients reach viewdirs and the 27 view columns)
 *   verdict: NULL or the caller's sticky launch-verdict words (above). */
int mofa_net_forward(MofaNetShape s, const float* packed, const float* folded, const float* view_w,
                     const float* view_b, const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                     const float* pts, const float* viewdirs, int64_t n_rays, int32_t S, float* workspace,
                     float* raw_out, float* tape, uint64_t* mask_tape, const float* view_bias_rows, uint32_t* verdict, void* stream);
/* ---- backward (run_fit.py:305-313 photometric fitting, run_train.py:333-357 training) -----------------------
 * Backward of mofa_net_forward given d_raw [n_rays,S,4] and the tape (fp32, or mask-only when d_weights == NULL) of that forward:
 *   d_folded  [mofa_net_folded_floats]: gradient w.r.t. every folded bias (sum over points of the ReLU-masked
 *             pre-activation gradient) — host autograd carries it on to the codes / raw biases / constant columns;
 *   d_view_bias_rows [n_rays, roundup(W/2,64)]: gradient w.r.t. the per-ray view bias rows;
 *   d_rays_o, d_rays_d [n_rays,3]: through the positional encoding and pts = o + d*z (z carries no gradient: the
 *             coarse z is constant and the fine z is detached, render_class.py:326);
 *   or, when the forward ran on explicit points (`pts` != NULL — run_network(inputs, viewdirs, fn) under autograd,
 *             render_class.py:69-94): d_pts [n_rays*S,3]; rays_o / rays_d / z / d_rays_o / d_rays_d are then unused (NULL).
 * packed_t: transposed weight panels from mofa_net_pack_t; workspace: mofa_net_backward_workspace_floats(). */
size_t mofa_net_packed_t_floats(MofaNetShape s);
size_t mofa_net_tape_floats(MofaNetShape s, int64_t n_points);
size_t mofa_net_mask_tape_words(MofaNetShape s, int64_t n_points);   /* uint64 words = tape floats / 64 */
size_t mofa_net_backward_workspace_floats(MofaNetShape s, int64_t n_points, int32_t with_weight_grads);   /* d_weights != NULL (training) or not (fitting): the two forms keep different buffers */
int mofa_net_pack_t(MofaNetShape s, const float* const* weights, float* packed_t, void* stream);
int mofa_net_backward(MofaNetShape s, const float* packed, const float* packed_t, const float* tape, const uint64_t* mask_tape,
                      const float* d_raw, const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                      const float* pts, int64_t n_rays, int32_t S, float* workspace, float* d_folded,
                      float* d_view_bias_rows, float* d_rays_o, float* d_rays_d, float* d_pts, float* const* d_weights, uint32_t* verdict,
                      void* stream);
/* d_weights: NULL (fitting: only codes/pose are optimised), or 2D+7 pointers to [out,in] gradient tensors in state-dict
 * order: the per-point column blocks are OVERWRITTEN with dW = G^T X (fp32 MFMA, contraction over the points, split over
 * M with a deterministic second-stage sum); the per-call-constant columns are left untouched (host autograd owns them). */
size_t mofa_weight_grad_workspace_floats(int64_t n_points, int32_t n_padded, int32_t k_padded);
int mofa_weight_grad(const float* g, int32_t n_padded, const float* x, int32_t k_padded, int64_t m_padded,
                     int64_t n_points, int32_t n_out, int32_t ncols, float* dst, int32_t ld, int32_t col0,
                     float* bias_out /* NULL or [n_padded] = sum_m G[m][:] from the same pass */, float* workspace,
                     void* stream);
int mofa_head_weight_grad(const float* d_raw, int32_t raw_off, int32_t n_out, const float* x, int32_t k_padded,
                          int64_t m_padded, int64_t n_points, int32_t ncols, float* dst, int32_t ld, void* stream);
/* positional-encoding features of every point (o + d z, or explicit `pts`) as panels [mofa_pe_k_padded(n_freqs)/16][m_padded][16] */
int mofa_pe_panels(const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride, const float* pts, int64_t n_points,
                   int32_t S, int32_t n_freqs, int64_t m_padded, float* out, void* stream);
/* pieces of mofa_net_backward (unit-testable) */
int mofa_pack_panels_t(const float* w, int32_t n_out, int32_t ld, int32_t col0, int32_t ncols, float* dst,
                       int32_t rows_padded, int32_t k_padded, void* stream);
/* mask: the saved fp32 activation (dx *= mask > 0).  _bits: the same mask as ONE BIT per activation — the mask-only tape's layout:
 * float offset o of the dx-shaped panel buffer <-> bit ((o & 255) >> 2) of 64-bit word (o >> 8) * 4 + (o & 3). */
int mofa_layer_backward_data(const float* g, int32_t g_k, const float* wt_packed, const float* mask, int32_t accumulate,
                             float* dx, int64_t m_padded, int32_t k_out_padded, void* stream);
int mofa_layer_backward_data_bits(const float* g, int32_t g_k, const float* wt_packed, const uint64_t* mask_bits, int32_t accumulate,
                                  float* dx, int64_t m_padded, int32_t k_out_padded, void* stream);
int mofa_head_backward(const float* d_raw, int32_t raw_off, int32_t n_out, const float* w_dense, int32_t k_padded,
                       const float* mask, int32_t accumulate, float* dx, int64_t m_padded, int64_t n_points, void* stream);
int mofa_head_backward_bits(const float* d_raw, int32_t raw_off, int32_t n_out, const float* w_dense, int32_t k_padded,
                            const uint64_t* mask_bits, int32_t accumulate, float* dx, int64_t m_padded, int64_t n_points, void* stream);
int mofa_bias_grad(const float* g, int64_t m_padded, int64_t n_points, int32_t n_padded, float* out, void* stream);
int mofa_bias_grad_rays(const float* g, int64_t m_padded, int64_t n_rays, int32_t S, int32_t n_padded, float* out,
                        void* stream);
int mofa_pe_backward(const float* dpe, int64_t m_padded, const float* rays_o, const float* rays_d, const float* z,
                     int64_t z_row_stride, int64_t n_rays, int32_t S, int32_t n_freqs, float* d_rays_o, float* d_rays_d, void* stream);
int mofa_pe_backward_points(const float* dpe, int64_t m_padded, const float* pts, int64_t n_points, int32_t n_freqs, float* d_pts,
                            void* stream);
/* raw2outputs backward: upstream gradients g_* (g_disp/g_acc/g_depth/g_weights may be NULL = zero) ->
 * d_raw [n_rays,S,4] and the |rays_d| contribution d_rays_d [n_rays,3] (may be NULL). */
int mofa_composite_backward(const float* raw, const float* z, int64_t z_row_stride, const float* rays_d,
                            const float* noise, int64_t n_rays, int32_t S, int32_t white_bkgd, const float* g_rgb,
                            const float* g_disp, const float* g_acc, const float* g_depth, const float* g_weights,
                            float* d_raw, float* d_rays_d, void* stream);

/* ---- single-layer entry points (unit-testable pieces of mofa_net_forward) -------------------- */
size_t mofa_panel_floats(int64_t rows, int32_t k); /* rows * roundup(k,16) */
int mofa_pack_panels(const float* w, int32_t n_out, int32_t ld, int32_t col0, int32_t ncols, float* dst,
                     int32_t rows_padded, int32_t panel0, int32_t k_padded, void* stream);
int mofa_to_panels(const float* x, int64_t rows, int32_t k, float* dst, int64_t rows_padded, void* stream);
int mofa_from_panels(const float* src, int64_t rows_padded, int64_t rows, int32_t k, float* x, void* stream);
/* y = act(x1|x2 @ Wpacked^T + bias); bias_row_div = 0: bias[Np]; else bias[(m / div), Np] (per ray).
 * _masked: mask_bits_out = NULL, or m_padded * n_padded / 64 words receiving (y > 0) as bits (mask-only tape; requires relu). */
int mofa_layer_forward(const float* x1, int32_t k1, const float* x2, int32_t k2, const float* w_packed,
                       const float* bias, int32_t bias_row_div, int64_t bias_rows, float* y, int64_t m_padded,
                       int32_t n_padded, int32_t relu, void* stream);
int mofa_layer_forward_masked(const float* x1, int32_t k1, const float* x2, int32_t k2, const float* w_packed,
                              const float* bias, int32_t bias_row_div, int64_t bias_rows, float* y, int64_t m_padded,
                              int32_t n_padded, int32_t relu, uint64_t* mask_bits_out, void* stream);
/* first layer with the positional encoding (n_freqs = multires) generated in the prologue (model.py:44-45 + Linear0);
 * w_packed: mofa_pe_k_padded(n_freqs) / 16 panels. */
int mofa_layer0_forward(const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride,
                        const float* pts, int64_t n_points, int32_t S, int32_t n_freqs, const float* w_packed, const float* bias,
                        float* y, int64_t m_padded, int32_t n_padded, uint64_t* mask_bits_out, void* stream);
/* Layer 0 with ray generation FOLDED INTO THE PROLOGUE: the ray of point m is built from
 * (intrinsics, c2w[3,4], pixel) by the arithmetic of mofa_get_rays — pixel = pixels[m / S] (flat row * img_w + col) or
 * pix0 + m / S when pixels == NULL — then pts = o + d * z as in mofa_layer0_forward.  Bit-identical to
 * mofa_get_rays(_at) followed by mofa_layer0_forward.  (The shipped renderer keeps the 24 B / ray arrays because compositing and
 * the positional-encoding backward read them as well; this entry is the kernel-level form.) */
int mofa_layer0_forward_cam(int32_t img_w, float fx, float fy, float cx, float cy, const float* c2w, const int32_t* pixels,
                            int64_t pix0, const float* z, int64_t z_row_stride, int64_t n_points, int32_t S, int32_t n_freqs,
                            const float* w_packed, const float* bias, float* y, int64_t m_padded, int32_t n_padded, void* stream);
int mofa_head_forward(const float* x, int32_t k_padded, int64_t m_padded, const float* w_dense, const float* b,
                      int32_t n_out, float* raw, int32_t raw_off, int64_t n_points, void* stream);
/* out[r][n] = bias[n] + sum_k w[n][k] * PE(viewdirs[r])[k], k < 3 + 6 * n_freqs (n_freqs = multires_views) */
int mofa_view_bias(const float* viewdirs, int64_t n_rays, int32_t n_freqs, const float* w, int32_t n_out, int32_t ld,
                   const float* bias, float* out, int32_t n_padded, void* stream);
int mofa_positional_encode(const float* x, int64_t n, int32_t n_freqs, float* out, void* stream);

/* ---- measurement hook ------------------------------------------------------------------------
 * A measurement session of the CALLING THREAD'S CURRENT DEVICE (state is per device, mutex-guarded; with no session
 * open the launch paths read one atomic flag).  Between mofa_prof_begin() and mofa_prof_end() every launch of the kernels below is
 * bracketed by hipEventRecord on its own stream.  mofa_prof_end() synchronises those events (host blocks) and fills three arrays of
 * length MOFA_PROF_KINDS: summed kernel time, launch count, WORK executed.
 *   MFMA kernels (work = FLOPs, 2*M*K*N of the padded shapes):
 *     [0] the per-layer forward kernel k_layer<128,..,PIPE> (128-feature tile, pipelined K loop)   [1] the persistent whole-network
 *     kernel k_mlp_fused (widths <= 256)   [2] the backward-data kernel k_layer<128,..,BWD>   [3] the weight-gradient kernel k_wgrad
 *     [4] the view layer's per-ray-bias instantiation of the forward kernel   [5] the chained wide-network kernel k_net_chain<0>
 *     (forward: inference or keeping the fp32 tape)   [6] k_net_chain<2> (the fitting backward's backward-data products)
 *     [7] k_net_chain<1> (forward, also writing the mask tape)   [11] k_net_chain_train (the training backward of a wide network as
     chained launches: backward-data products and weight gradients)
 *   HBM-bound ray kernels (work = RAYS; bench.py multiplies by SURVEY section 8d's algorithmic bytes per ray):
 *     [8] k_composite<1> (S <= 64: the coarse pass)   [9] k_composite<2> (S <= 128: the fine pass)   [10] k_sample_pdf_merge
 * Used by bench.py only. */
#define MOFA_PROF_KINDS 12
int mofa_prof_begin(void);
int mofa_prof_end(double* total_ms, int64_t* launches, double* padded_flops);

/* ---- ray-side kernels ------------------------------------------------------------------------ */
/* get_rays (tools/run_nerf_helpers.py:153-168) + viewdir normalisation (render_class.py:399-401) for
 * pixels [pix0, pix0+n) of an H x W image in row-major order.  c2w: 12 floats [3,4] (device). */
int mofa_get_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* c2w, int64_t pix0,
                  int64_t n, float* rays_o, float* rays_d, float* viewdirs, void* stream);

/* The same for a LIST of pixels (flat indices row * W + col, int32, device): the rays of a fitting / training batch without
 * building the H x W grid first (run_fit.py:281-293 builds get_rays_withGrad's full grid and gathers N_rand of it;
 * run_train.py:306-330).  Bit-identical to mofa_get_rays at those pixels. */
int mofa_get_rays_at(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* c2w, const int32_t* pixels,
                     int64_t n, float* rays_o, float* rays_d, float* viewdirs, void* stream);

/* Reverse of ray generation — the camera-pose gradient run_fit.py optimises (get_rays_withGrad, run_fit.py:116-127):
 *   d_c2w[a][b] = sum_rays d_rays_d[ray][a] * dirs[ray][b] (b < 3),  d_c2w[a][3] = sum_rays d_rays_o[ray][a];  d_c2w: 12 floats.
 * pixels == NULL: the contiguous pixel range [pix0, pix0 + n).  Deterministic (fixed-order double sums, no atomics). */
int mofa_rays_pose_backward(int32_t W, float fx, float fy, float cx, float cy, const int32_t* pixels, int64_t pix0, int64_t n,
                            const float* d_rays_o, const float* d_rays_d, float* d_c2w, void* stream);

/* raw2outputs (render_class.py:440-482): one wavefront per ray, exclusive prefix product over the
 * samples.  noise may be NULL; disp is NaN where acc == 0 exactly like the reference. */
int mofa_composite_forward(const float* raw, const float* z, int64_t z_row_stride, const float* rays_d,
                           const float* noise, int64_t n_rays, int32_t S, int32_t white_bkgd, float* rgb,
                           float* disp, float* acc, float* depth, float* weights, void* stream);

/* sample_pdf on (z_mid, weights[1:-1]) + sort(cat(z, z_samples)) + std(z_samples)
 * (render_class.py:324-328,345; tools/run_nerf_helpers.py:203-247).  u: [n_rays,Ni] (stride Ni) or a
 * shared row (stride 0) — linspace(0,1,Ni) for det. */
int mofa_sample_pdf_merge(const float* z, int64_t z_row_stride, const float* weights, const float* u,
                          int64_t u_row_stride, int64_t n_rays, int32_t S, int32_t Ni, float* z_samples,
                          float* z_fine, float* z_std, void* stream);

/* sample_pdf(bins, weights, N_samples, det / u) exactly as tools/run_nerf_helpers.py:203-247 takes it: bins [n_rays, n_bins]
 * (row stride given; 0 = one shared row), weights [n_rays, n_bins-1], u as above -> samples [n_rays, Ni].  Same kernel as
 * mofa_sample_pdf_merge without the mid-point / merge stages; this is the form the reference's own KATs are stated in. */
int mofa_sample_pdf(const float* bins, int64_t bins_row_stride, const float* weights, const float* u, int64_t u_row_stride,
                    int64_t n_rays, int32_t n_bins, int32_t Ni, float* samples, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MOFANERF_HIP_H */
