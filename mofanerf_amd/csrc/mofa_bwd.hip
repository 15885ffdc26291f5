// Backward kernels of the hot path (fitting / training: run_fit.py:305-313, run_train.py:333-357).
// The backward-data GEMMs reuse k_layer (mofa_mlp.hip, BWD epilogue); this file holds the HBM-bound pieces:
// head backward, bias-gradient column sums, positional-encoding backward and raw2outputs backward.
// Built with -ffp-contract=off like the forward.
#include <type_traits>

#include "mofa_layer.h"
#include "mofa_wgrad.h"

extern "C" {
int mofa_internal_prof_open(void* stream, int kind);
void mofa_internal_prof_close(void* stream, int kind, double flops);
int mofa_internal_wgrad_reduce(const float* partial, int splits, int n_padded, int k_padded, int n_out, int ncols, float* dst, int ld,
                               int col0, float* bias_out, void* stream);
}

namespace mofa {
namespace {

constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- head backward: dX[m][k] (+)= sum_o d_raw[m][off+o] * w[o][k], optionally * (saved > 0) -------------------
// one thread per (point, 16-byte chunk of features); dx in panels [kp][m_padded][16].
__global__ __launch_bounds__(256) void k_head_backward(const float* __restrict__ d_raw, int raw_off, int n_out,
                                                       const float* __restrict__ w, int k_padded,
                                                       const float* __restrict__ mask, const unsigned long long* __restrict__ mask_bits,
                                                       int accumulate, float* __restrict__ dx, long long m_padded, long long n_points) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over m_padded * (k_padded/4)
    const int chunks = k_padded >> 2;
    if (idx >= m_padded * chunks) return;
    // consecutive threads walk one panel row-major: (panel, m, physical chunk p)
    const int p = (int)(idx & 3);
    const long long rowpanel = idx >> 2;
    const long long m = rowpanel % m_padded;
    const int panel = (int)(rowpanel / m_padded);
    const int k = panel * 16 + ((p ^ ((int)(m >> 2) & 3)) << 2);
    const long long off = ((long long)panel * m_padded + m) * 16 + (p << 2);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < n_points) {
        for (int o = 0; o < n_out; ++o) {
            const float g = d_raw[m * 4 + raw_off + o];
            const float* wo = w + (long long)o * k_padded + k;
            v.x = fmaf(g, wo[0], v.x), v.y = fmaf(g, wo[1], v.y), v.z = fmaf(g, wo[2], v.z), v.w = fmaf(g, wo[3], v.w);
        }
    }
    if (accumulate) {
        const f32x4 o = *(const f32x4*)(dx + off);
        v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
    }
    if (mask) {
        const f32x4 s = *(const f32x4*)(mask + off);
        v.x = s.x > 0.f ? v.x : 0.f, v.y = s.y > 0.f ? v.y : 0.f, v.z = s.z > 0.f ? v.z : 0.f, v.w = s.w > 0.f ? v.w : 0.f;
    } else if (mask_bits) {
        bool keep[4];
        mask_load_quad(mask_bits, off, keep);
        v.x = keep[0] ? v.x : 0.f, v.y = keep[1] ? v.y : 0.f, v.z = keep[2] ? v.z : 0.f, v.w = keep[3] ? v.w : 0.f;
    }
    *(f32x4*)(dx + off) = v;
}

// ---- bias gradient: out[n] = sum_{m < n_points} G[m][n]; one block per 16-feature panel ----------------------
// One workgroup per 16-feature panel (64 for a 1024-wide layer), so each must keep a whole CU's memory pipeline busy on its
// own.  A panel is [rows][64 B]: lane l of a wave reads the 16-byte chunk (l & 3) of row (l >> 2), so one wave instruction is
// 1 KiB of CONTIGUOUS memory (a thread-per-row layout reads 16 B at a 64-byte stride: four quarter-filled requests per row —
// measured 1.5 TB/s over 64 workgroups instead of this form's rate).  The row swizzle ((row >> 2) & 3) is constant per lane
// when rows advance by multiples of 16, so every lane accumulates ONE logical chunk (4 features).  1024 threads, four rows in
// flight per thread; fixed combination order (deterministic).
__global__ __launch_bounds__(1024) void k_colsum(const float* __restrict__ g, long long m_padded, long long n_points,
                                                 float* __restrict__ out) {
    __shared__ f32x4 red[1024];
    const int panel = blockIdx.x, tid = threadIdx.x;
    const int chunk = tid & 3, r0 = tid >> 2;                 // physical chunk, first row; rows r0 + 256 i
    // this lane's LOGICAL chunk is chunk ^ ((r0 >> 2) & 3) for every row it visits ((256 i) >> 2 is a multiple of 4)
    const float* base = g + (long long)panel * m_padded * 16 + chunk * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long m = r0;
    for (; m + 3 * 256 < n_points; m += 4 * 256) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(base + (m + u * 256) * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
    }
    for (; m < n_points; m += 256) {
        const f32x4 v = *(const f32x4*)(base + m * 16);
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < 256) {      // threads 256 apart hold the same logical chunk
        f32x4 t = red[tid];
#pragma unroll
        for (int u = 1; u < 4; ++u) {
            const f32x4 v = red[tid + 256 * u];
            t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
        }
        red[tid] = t;
    }
    __syncthreads();
    if (tid < 16) {       // feature tid = logical chunk tid >> 2, component tid & 3: the 64 of those 256 threads that hold that chunk
        const int want = tid >> 2, comp = tid & 3;
        float t = 0.f;
        for (int i = 0; i < 256; ++i) {
            const int lg = (i & 3) ^ ((i >> 4) & 3);
            if (lg == want) t += red[i][comp];
        }
        out[panel * 16 + tid] = t;
    }
}

// The same sum split over the rows (fitting's five bias gradients, mofa_net_backward): 64 workgroups — one per panel of a 1024-wide
// layer — are a quarter of the chip, and the column sum of a 0.5 GB gradient buffer is HBM-bound work (175 us = 3 TB/s in that form).
// Here blockIdx.y walks `splits` row ranges (multiples of 1024 rows, so the per-lane swizzle argument above still holds), each writing
// a partial row; k_colsum_combine adds the partials in index order: deterministic, independent of the device.
__global__ __launch_bounds__(1024) void k_colsum_split(const float* __restrict__ g, long long m_padded, long long n_points, long long rows_per_split,
                                                       int n_padded, float* __restrict__ partial) {
    __shared__ f32x4 red[1024];
    const int panel = blockIdx.x, tid = threadIdx.x;
    const int chunk = tid & 3, r0 = tid >> 2;
    const long long lo = (long long)blockIdx.y * rows_per_split;
    long long hi = lo + rows_per_split;
    if (hi > n_points) hi = n_points;
    const float* base = g + (long long)panel * m_padded * 16 + chunk * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long m = lo + r0;
    for (; m + 3 * 256 < hi; m += 4 * 256) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(base + (m + u * 256) * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
    }
    for (; m < hi; m += 256) {
        const f32x4 v = *(const f32x4*)(base + m * 16);
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < 256) {
        f32x4 t = red[tid];
#pragma unroll
        for (int u = 1; u < 4; ++u) {
            const f32x4 v = red[tid + 256 * u];
            t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
        }
        red[tid] = t;
    }
    __syncthreads();
    if (tid < 16) {
        const int want = tid >> 2, comp = tid & 3;
        float t = 0.f;
        for (int i = 0; i < 256; ++i) {
            const int lg = (i & 3) ^ ((i >> 4) & 3);
            if (lg == want) t += red[i][comp];
        }
        partial[(long long)blockIdx.y * n_padded + panel * 16 + tid] = t;
    }
}

__global__ __launch_bounds__(256) void k_colsum_combine(const float* __restrict__ partial, int splits, int n_padded, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_padded) return;
    float t = 0.f;
    for (int s = 0; s < splits; ++s) t += partial[(long long)s * n_padded + n];
    out[n] = t;
}

__global__ __launch_bounds__(256) void k_colsum_rays(const float* __restrict__ g, long long m_padded, long long n_rays,
                                                     int S, int n_padded, float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int chunks = n_padded >> 2;
    if (idx >= n_rays * chunks) return;
    const long long r = idx / chunks;
    const int ch = (int)(idx - r * chunks);
    const int panel = ch >> 2, c = ch & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const long long m = r * S + s;
        const f32x4 v = *(const f32x4*)(g + ((long long)panel * m_padded + m) * 16 + ((c ^ ((int)(m >> 2) & 3)) << 2));
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    *(f32x4*)(out + r * n_padded + ch * 4) = acc;
}

// ---- positional-encoding backward + pts = o + d*z backward --------------------------------------------------
// gradient w.r.t. the point x from the gradient w.r.t. its 3 + 6 * n_freqs encoding features (panel layout, point row m)
__device__ __forceinline__ void pe_point_backward(const float* __restrict__ dpe, long long m_padded, long long m, const float (&x)[3],
                                                  int n_freqs, float (&gx)[3]) {
    const int sw = (int)(m >> 2) & 3;
    auto at = [&](int k) -> float {   // gradient w.r.t. encoding feature k of point m (panel layout)
        return dpe[(long long)(k >> 4) * m_padded * 16 + m * 16 + ((((k >> 2) & 3) ^ sw) << 2) + (k & 3)];
    };
#pragma unroll
    for (int d = 0; d < 3; ++d) gx[d] = at(d);                     // identity features
#pragma unroll 1
    for (int f = 0; f < n_freqs; ++f) {
        const float fr = (float)(1 << f);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float gs = at(3 + 6 * f + d), gc = at(6 + 6 * f + d);
            float sn, cs;
            sincosf(x[d] * fr, &sn, &cs);
            // d/dx sin(f x) = f cos(f x);  d/dx cos(f x) = -f sin(f x)
            gx[d] += fr * (gs * cs - gc * sn);
        }
    }
}

// explicit points (run_network(inputs, ...) under autograd, models/render_class.py:69-94): d_pts[m] per point, one thread each
__global__ __launch_bounds__(256) void k_pe_backward_pts(const float* __restrict__ dpe, long long m_padded, const float* __restrict__ pts,
                                                         long long n_points, int n_freqs, float* __restrict__ d_pts) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_points) return;
    const float x[3] = {pts[m * 3], pts[m * 3 + 1], pts[m * 3 + 2]};
    float gx[3];
    pe_point_backward(dpe, m_padded, m, x, n_freqs, gx);
    d_pts[m * 3] = gx[0], d_pts[m * 3 + 1] = gx[1], d_pts[m * 3 + 2] = gx[2];
}

// dpe: panels [pe_k_padded/16][m_padded][16] (gradient w.r.t. the 3 + 6 * n_freqs encoding features; the rest is padding).  One wavefront per ray.
__global__ __launch_bounds__(256) void k_pe_backward(const float* __restrict__ dpe, long long m_padded,
                                                     const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ z, long long z_row_stride,
                                                     long long n_rays, int S, int n_freqs, float* __restrict__ d_rays_o,
                                                     float* __restrict__ d_rays_d) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
    for (int s = lane; s < S; s += 64) {
        const long long m = ray * S + s;
        const float zz = z[ray * z_row_stride + s];
        const float x[3] = {__fadd_rn(ox, __fmul_rn(dx, zz)), __fadd_rn(oy, __fmul_rn(dy, zz)),
                            __fadd_rn(oz, __fmul_rn(dz, zz))};
        float gx[3];
        pe_point_backward(dpe, m_padded, m, x, n_freqs, gx);
#pragma unroll
        for (int c = 0; c < 3; ++c) go[c] += gx[c], gd[c] += gx[c] * zz;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = wave_sum(go[c]), b = wave_sum(gd[c]);
        if (lane == 0) d_rays_o[ray * 3 + c] = a, d_rays_d[ray * 3 + c] = b;
    }
}

// ---- raw2outputs backward (models/render_class.py:440-482) ---------------------------------------------------
// w_i = alpha_i T_i, T_i = prod_{j<i}(1 - alpha_j + 1e-10).  With G_i = dL/dw_i (collected from rgb, depth, acc, disp and
// the explicit weights gradient):  dL/dalpha_j = G_j T_j - (sum_{i>j} G_i w_i) / (1 - alpha_j + 1e-10).
template <int SPL>
__global__ __launch_bounds__(256) void k_composite_backward(
    const float* __restrict__ raw, const float* __restrict__ z, long long z_row_stride, const float* __restrict__ rays_d,
    const float* __restrict__ noise, long long n_rays, int S, int white_bkgd, const float* __restrict__ g_rgb,
    const float* __restrict__ g_disp, const float* __restrict__ g_acc, const float* __restrict__ g_depth,
    const float* __restrict__ g_weights, float* __restrict__ d_raw, float* __restrict__ d_rays_d) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float dnorm = __fsqrt_rn(dx * dx + dy * dy + dz * dz);
    const float* zr = z + ray * z_row_stride;
    const f32x4* rr = (const f32x4*)(raw + ray * (long long)S * 4);

    float zv[SPL + 1], alpha[SPL], sig[SPL], dlt[SPL], cr[SPL], cg[SPL], cb[SPL];
    const int s0 = lane * SPL;
#pragma unroll
    for (int t = 0; t <= SPL; ++t) zv[t] = (s0 + t < S) ? zr[s0 + t] : 0.f;
    float run = 1.0f;
#pragma unroll
    for (int t = 0; t < SPL; ++t) {
        const int s = s0 + t;
        if (s < S) {
            const f32x4 v = rr[s];
            dlt[t] = (s + 1 < S) ? (zv[t + 1] - zv[t]) : 1e10f;
            float sg = v.w;
            if (noise) sg = sg + noise[ray * (long long)S + s];
            sig[t] = sg;                                   // pre-ReLU (sign decides the gate)
            alpha[t] = 1.0f - expf(-relu_np(sg) * (dlt[t] * dnorm));
            cr[t] = 1.0f / (1.0f + expf(-v.x)), cg[t] = 1.0f / (1.0f + expf(-v.y)), cb[t] = 1.0f / (1.0f + expf(-v.z));
            run = run * ((1.0f - alpha[t]) + 1e-10f);
        } else {
            alpha[t] = 0.f, sig[t] = 0.f, dlt[t] = 0.f, cr[t] = cg[t] = cb[t] = 0.f;
        }
    }
    float incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(incl, o, 64);
        if (lane >= o) incl = incl * up;
    }
    float T0 = __shfl_up(incl, 1, 64);
    if (lane == 0) T0 = 1.0f;

    // forward sums needed by the disp gradient
    float T = T0, sd = 0.f, sa = 0.f;
    float w[SPL], Tt[SPL];
#pragma unroll
    for (int t = 0; t < SPL; ++t) {
        Tt[t] = T;
        w[t] = alpha[t] * T;
        if (s0 + t < S) sd += w[t] * zv[t], sa += w[t];
        T = T * ((1.0f - alpha[t]) + 1e-10f);
    }
    sd = wave_sum(sd), sa = wave_sum(sa);
    const float gr = g_rgb[ray * 3], gg = g_rgb[ray * 3 + 1], gb = g_rgb[ray * 3 + 2];
    float gdepth = g_depth ? g_depth[ray] : 0.f, gacc = g_acc ? g_acc[ray] : 0.f;
    const float gdisp = g_disp ? g_disp[ray] : 0.f;
    if (gdisp != 0.f) {                                     // disp = acc/depth where depth/acc > 1e-10
        const float q = __fdiv_rn(sd, sa);
        if (q > 1e-10f) gdepth += gdisp * (-sa / (sd * sd)), gacc += gdisp * (1.0f / sd);
        else if (q != q) gdepth += q, gacc += q;             // NaN propagates like autograd through 0/0
    }
    if (white_bkgd) gacc -= gr + gg + gb;                    // rgb += 1 - acc

    // G_i and the suffix sums of G_i w_i
    float G[SPL], local = 0.f;
#pragma unroll
    for (int t = 0; t < SPL; ++t) {
        const int s = s0 + t;
        G[t] = (s < S) ? ((g_weights ? g_weights[ray * (long long)S + s] : 0.f) + gr * cr[t] + gg * cg[t] + gb * cb[t] +
                          gdepth * zv[t] + gacc)
                       : 0.f;
        local += G[t] * w[t];
    }
    float suf = local;                                      // inclusive suffix sum over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float dn = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += dn;
    }
    float after = __shfl_down(suf, 1, 64);                  // sum over lanes > this one
    if (lane == 63) after = 0.f;

    float dn_acc = 0.f;                                     // d L / d |rays_d|
#pragma unroll
    for (int t = SPL - 1; t >= 0; --t) {
        const int s = s0 + t;
        if (s < S) {
            const float one_m = (1.0f - alpha[t]) + 1e-10f;
            const float dalpha = G[t] * Tt[t] - after / one_m;
            const float dist = dlt[t] * dnorm;
            const float keep = 1.0f - alpha[t];             // exp(-sigma dist)
            const float dsig = sig[t] > 0.f ? dalpha * dist * keep : 0.f;
            const float ddist = dalpha * relu_np(sig[t]) * keep;
            dn_acc += ddist * dlt[t];
            f32x4 o;
            o.x = w[t] * gr * cr[t] * (1.0f - cr[t]);
            o.y = w[t] * gg * cg[t] * (1.0f - cg[t]);
            o.z = w[t] * gb * cb[t] * (1.0f - cb[t]);
            o.w = dsig;
            *(f32x4*)(d_raw + (ray * (long long)S + s) * 4) = o;
            after += G[t] * w[t];
        }
    }
    dn_acc = wave_sum(dn_acc);
    if (lane == 0 && d_rays_d) {
        const float inv = dnorm > 0.f ? 1.0f / dnorm : 0.f;
        d_rays_d[ray * 3] = dn_acc * dx * inv, d_rays_d[ray * 3 + 1] = dn_acc * dy * inv, d_rays_d[ray * 3 + 2] = dn_acc * dz * inv;
    }
}

// Rays with more than 256 samples: the same backward walked in passes of 256 samples (lane l owns samples [256 p + 4 l, +4) of
// pass p).  Sweep 1 (front to back) carries the transmittance from pass to pass and leaves the value in front of every pass in LDS
// (plus the ray sums the disp gradient needs); sweep 2 (back to front) recomputes a pass's alphas from `raw` and carries the suffix
// sum  sum_{i > j} G_i w_i  from the passes behind it.  kMaxPasses * 256 samples per ray.
constexpr int kMaxPasses = 64;
__global__ __launch_bounds__(256) void k_composite_backward_long(
    const float* __restrict__ raw, const float* __restrict__ z, long long z_row_stride, const float* __restrict__ rays_d,
    const float* __restrict__ noise, long long n_rays, int S, int white_bkgd, const float* __restrict__ g_rgb,
    const float* __restrict__ g_disp, const float* __restrict__ g_acc, const float* __restrict__ g_depth,
    const float* __restrict__ g_weights, float* __restrict__ d_raw, float* __restrict__ d_rays_d) {
    constexpr int SPL = 4;
    __shared__ float s_carry[kWavesPerBlock][kMaxPasses];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * kWavesPerBlock + wv;
    if (ray >= n_rays) return;                              // wave-uniform; only wave-level synchronisation below
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float dnorm = __fsqrt_rn(dx * dx + dy * dy + dz * dz);
    const float* zr = z + ray * z_row_stride;
    const f32x4* rr = (const f32x4*)(raw + ray * (long long)S * 4);
    const int passes = (S + 64 * SPL - 1) / (64 * SPL);

    float zv[SPL + 1], alpha[SPL], sig[SPL], dlt[SPL], cr[SPL], cg[SPL], cb[SPL];
    auto load_pass = [&](int pass) -> float {               // fills the per-sample arrays; returns this lane's (1 - alpha + 1e-10) product
        const int s0 = pass * 64 * SPL + lane * SPL;
#pragma unroll
        for (int t = 0; t <= SPL; ++t) zv[t] = (s0 + t < S) ? zr[s0 + t] : 0.f;
        float run = 1.0f;
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
            const int s = s0 + t;
            if (s < S) {
                const f32x4 v = rr[s];
                dlt[t] = (s + 1 < S) ? (zv[t + 1] - zv[t]) : 1e10f;
                float sg = v.w;
                if (noise) sg = sg + noise[ray * (long long)S + s];
                sig[t] = sg;
                alpha[t] = 1.0f - expf(-relu_np(sg) * (dlt[t] * dnorm));
                cr[t] = 1.0f / (1.0f + expf(-v.x)), cg[t] = 1.0f / (1.0f + expf(-v.y)), cb[t] = 1.0f / (1.0f + expf(-v.z));
                run = run * ((1.0f - alpha[t]) + 1e-10f);
            } else {
                alpha[t] = 0.f, sig[t] = 0.f, dlt[t] = 0.f, cr[t] = cg[t] = cb[t] = 0.f;
            }
        }
        return run;
    };
    auto lane_T0 = [&](float run, float carry, float& total) -> float {   // transmittance in front of this lane's first sample
        float incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o, 64);
            if (lane >= o) incl = incl * up;
        }
        float T0 = __shfl_up(incl, 1, 64);
        if (lane == 0) T0 = 1.0f;
        total = __shfl(incl, 63, 64);
        return carry * T0;
    };

    // sweep 1: carries and the ray sums
    float carry = 1.0f, sd = 0.f, sa = 0.f;
    for (int pass = 0; pass < passes; ++pass) {
        const float run = load_pass(pass);
        if (lane == 0) s_carry[wv][pass] = carry;
        float total;
        float T = lane_T0(run, carry, total);
        carry = carry * total;
        const int s0 = pass * 64 * SPL + lane * SPL;
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
            const float w = alpha[t] * T;
            if (s0 + t < S) sd += w * zv[t], sa += w;
            T = T * ((1.0f - alpha[t]) + 1e-10f);
        }
    }
    __builtin_amdgcn_wave_barrier();
    sd = wave_sum(sd), sa = wave_sum(sa);
    const float gr = g_rgb[ray * 3], gg = g_rgb[ray * 3 + 1], gb = g_rgb[ray * 3 + 2];
    float gdepth = g_depth ? g_depth[ray] : 0.f, gacc = g_acc ? g_acc[ray] : 0.f;
    const float gdisp = g_disp ? g_disp[ray] : 0.f;
    if (gdisp != 0.f) {
        const float q = __fdiv_rn(sd, sa);
        if (q > 1e-10f) gdepth += gdisp * (-sa / (sd * sd)), gacc += gdisp * (1.0f / sd);
        else if (q != q) gdepth += q, gacc += q;
    }
    if (white_bkgd) gacc -= gr + gg + gb;

    // sweep 2: back to front
    float behind = 0.f, dn_acc = 0.f;                       // sum of G_i w_i over every LATER pass; d L / d |rays_d|
    for (int pass = passes - 1; pass >= 0; --pass) {
        const float run = load_pass(pass);
        float total;
        const float T0 = lane_T0(run, s_carry[wv][pass], total);
        const int s0 = pass * 64 * SPL + lane * SPL;
        float T = T0, w[SPL], Tt[SPL], G[SPL], local = 0.f;
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
            Tt[t] = T;
            w[t] = alpha[t] * T;
            T = T * ((1.0f - alpha[t]) + 1e-10f);
            const int s = s0 + t;
            G[t] = (s < S) ? ((g_weights ? g_weights[ray * (long long)S + s] : 0.f) + gr * cr[t] + gg * cg[t] + gb * cb[t] +
                              gdepth * zv[t] + gacc)
                           : 0.f;
            local += G[t] * w[t];
        }
        float suf = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float dn = __shfl_down(suf, o, 64);
            if (lane + o < 64) suf += dn;
        }
        float after = __shfl_down(suf, 1, 64);
        if (lane == 63) after = 0.f;
        after += behind;
        behind += __shfl(suf, 0, 64);
#pragma unroll
        for (int t = SPL - 1; t >= 0; --t) {
            const int s = s0 + t;
            if (s < S) {
                const float one_m = (1.0f - alpha[t]) + 1e-10f;
                const float dalpha = G[t] * Tt[t] - after / one_m;
                const float dist = dlt[t] * dnorm;
                const float keep = 1.0f - alpha[t];
                const float dsig = sig[t] > 0.f ? dalpha * dist * keep : 0.f;
                const float ddist = dalpha * relu_np(sig[t]) * keep;
                dn_acc += ddist * dlt[t];
                f32x4 o;
                o.x = w[t] * gr * cr[t] * (1.0f - cr[t]);
                o.y = w[t] * gg * cg[t] * (1.0f - cg[t]);
                o.z = w[t] * gb * cb[t] * (1.0f - cb[t]);
                o.w = dsig;
                *(f32x4*)(d_raw + (ray * (long long)S + s) * 4) = o;
                after += G[t] * w[t];
            }
        }
    }
    dn_acc = wave_sum(dn_acc);
    if (lane == 0 && d_rays_d) {
        const float inv = dnorm > 0.f ? 1.0f / dnorm : 0.f;
        d_rays_d[ray * 3] = dn_acc * dx * inv, d_rays_d[ray * 3 + 1] = dn_acc * dy * inv, d_rays_d[ray * 3 + 2] = dn_acc * dz * inv;
    }
}

// ---- head bias gradients: the four column sums of d_raw [n_points][4] (rgb head: columns 0..2, sigma head: column 3) in ONE pass ----
// Round 6: this was one 1,024-thread workgroup per head walking all the points — 95 us per call at a training sub-batch, twice per
// backward (0.16 % of a training step on a single CU).  Now up to 256 workgroups sum contiguous ranges of the points into partial rows
// and a second kernel adds the rows in index order: deterministic, independent of the device; both heads from one read of d_raw.
constexpr int kRawColsumRows = 16384;       // points per workgroup (a multiple of the 1,024-thread stride)
__global__ __launch_bounds__(1024) void k_raw_colsum(const float* __restrict__ d_raw, long long n_points, float* __restrict__ partial) {
    __shared__ float red[1024][4];
    const long long lo = (long long)blockIdx.x * kRawColsumRows;
    long long hi = lo + kRawColsumRows;
    if (hi > n_points) hi = n_points;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long m = lo + threadIdx.x; m < hi; m += 1024) {      // one 16-byte load per point: the 4 raw columns
        const f32x4 v = *(const f32x4*)(d_raw + m * 4);
        acc[0] += v.x, acc[1] += v.y, acc[2] += v.z, acc[3] += v.w;
    }
    for (int c = 0; c < 4; ++c) red[threadIdx.x][c] = acc[c];
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int c = 0; c < 4; ++c) red[threadIdx.x][c] += red[threadIdx.x + s][c];
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(long long)blockIdx.x * 4 + threadIdx.x] = red[0][threadIdx.x];
}
// out_rgb[0..3] = {sum col 0, 1, 2, 0}, out_sigma[0..3] = {sum col 3, 0, 0, 0} (the heads' folded-bias slots are 4 floats wide)
__global__ void k_raw_colsum_combine(const float* __restrict__ partial, int rows, float* __restrict__ out_rgb, float* __restrict__ out_sigma) {
    const int c = threadIdx.x;
    if (c >= 4) return;
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += partial[r * 4 + c];
    if (c < 3) out_rgb[c] = t;
    else out_rgb[3] = 0.f, out_sigma[0] = t, out_sigma[1] = 0.f, out_sigma[2] = 0.f, out_sigma[3] = 0.f;
}

inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace
}  // namespace mofa

using namespace mofa;

extern "C" {

static int head_backward(const float* d_raw, int32_t raw_off, int32_t n_out, const float* w_dense, int32_t k_padded,
                         const float* mask, const uint64_t* mask_bits, int32_t accumulate, float* dx, int64_t m_padded, int64_t n_points,
                         void* stream);
int mofa_head_backward(const float* d_raw, int32_t raw_off, int32_t n_out, const float* w_dense, int32_t k_padded,
                       const float* mask, int32_t accumulate, float* dx, int64_t m_padded, int64_t n_points, void* stream) {
    return head_backward(d_raw, raw_off, n_out, w_dense, k_padded, mask, nullptr, accumulate, dx, m_padded, n_points, stream);
}
int mofa_head_backward_bits(const float* d_raw, int32_t raw_off, int32_t n_out, const float* w_dense, int32_t k_padded,
                            const uint64_t* mask_bits, int32_t accumulate, float* dx, int64_t m_padded, int64_t n_points, void* stream) {
    MOFA_REQUIRE(mask_bits, "head_backward_bits: null mask");
    return head_backward(d_raw, raw_off, n_out, w_dense, k_padded, nullptr, mask_bits, accumulate, dx, m_padded, n_points, stream);
}
static int head_backward(const float* d_raw, int32_t raw_off, int32_t n_out, const float* w_dense, int32_t k_padded,
                         const float* mask, const uint64_t* mask_bits, int32_t accumulate, float* dx, int64_t m_padded, int64_t n_points,
                         void* stream) {
    MOFA_REQUIRE(d_raw && w_dense && dx, "head_backward: null pointer");
    MOFA_REQUIRE(k_padded % 16 == 0 && n_out >= 1 && raw_off >= 0 && raw_off + n_out <= 4 && n_points <= m_padded,
                 "head_backward: bad shape");
    hipLaunchKernelGGL(k_head_backward, dim3(blocks_for(m_padded * (k_padded / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, d_raw, raw_off, n_out, w_dense, k_padded, mask, (const unsigned long long*)mask_bits,
                       accumulate, dx, (long long)m_padded, (long long)n_points);
    return check_launch("k_head_backward");
}

// internal (mofa_net_backward): both heads' bias gradients; `scratch`: 4 floats per 16,384 points
int mofa_internal_raw_colsum(const float* d_raw, long long n_points, float* out_rgb, float* out_sigma, float* scratch, void* stream) {
    const int rows = (int)((n_points + kRawColsumRows - 1) / kRawColsumRows);
    hipLaunchKernelGGL(k_raw_colsum, dim3(rows), dim3(1024), 0, (hipStream_t)stream, d_raw, n_points, scratch);
    hipLaunchKernelGGL(k_raw_colsum_combine, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, rows, out_rgb, out_sigma);
    return check_launch("k_raw_colsum");
}

int mofa_bias_grad(const float* g, int64_t m_padded, int64_t n_points, int32_t n_padded, float* out, void* stream) {
    MOFA_REQUIRE(g && out && n_padded % 16 == 0 && n_points <= m_padded, "bias_grad: bad arguments");
    hipLaunchKernelGGL(k_colsum, dim3(n_padded / 16), dim3(1024), 0, (hipStream_t)stream, g, (long long)m_padded,
                       (long long)n_points, out);
    return check_launch("k_colsum");
}

// internal (mofa_net_backward, fitting): the same column sums over `splits` row ranges + a fixed-order combine; `workspace`: at least
// 8 * n_padded floats.  The number of ranges depends on the SHAPE only (never on the device), so results are reproducible everywhere.
int mofa_internal_bias_grad_split(const float* g, long long m_padded, long long n_points, int n_padded, float* out, float* workspace,
                                  void* stream) {
    MOFA_REQUIRE(g && out && workspace && n_padded % 16 == 0 && n_points > 0 && n_points <= m_padded, "bias_grad_split: bad arguments");
    const int panels = n_padded / 16;
    int splits = panels >= 256 ? 1 : 256 / panels;                 // ~ one workgroup per CU of a 256-CU part
    if (splits > 8) splits = 8;
    long long rows = (n_points + splits - 1) / splits;
    rows = (rows + 1023) / 1024 * 1024;                            // whole 1024-row strides per range
    splits = (int)((n_points + rows - 1) / rows);
    hipLaunchKernelGGL(k_colsum_split, dim3(panels, splits), dim3(1024), 0, (hipStream_t)stream, g, m_padded, n_points, rows, n_padded, workspace);
    int rc = check_launch("k_colsum_split");
    if (rc != MOFA_OK) return rc;
    hipLaunchKernelGGL(k_colsum_combine, dim3((n_padded + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace, splits, n_padded, out);
    return check_launch("k_colsum_combine");
}

int mofa_bias_grad_rays(const float* g, int64_t m_padded, int64_t n_rays, int32_t S, int32_t n_padded, float* out,
                        void* stream) {
    MOFA_REQUIRE(g && out && n_padded % 16 == 0 && n_rays * S <= m_padded, "bias_grad_rays: bad arguments");
    hipLaunchKernelGGL(k_colsum_rays, dim3(blocks_for(n_rays * (n_padded / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       g, (long long)m_padded, (long long)n_rays, S, n_padded, out);
    return check_launch("k_colsum_rays");
}

int mofa_pe_backward(const float* dpe, int64_t m_padded, const float* rays_o, const float* rays_d, const float* z,
                     int64_t z_row_stride, int64_t n_rays, int32_t S, int32_t n_freqs, float* d_rays_o, float* d_rays_d, void* stream) {
    MOFA_REQUIRE(dpe && rays_o && rays_d && z && d_rays_o && d_rays_d && n_rays * S <= m_padded,
                 "pe_backward: bad arguments");
    MOFA_REQUIRE(n_freqs >= 0 && n_freqs <= MOFA_MAX_PE_FREQS, "pe_backward: n_freqs=%d out of [0, %d]", n_freqs, MOFA_MAX_PE_FREQS);
    hipLaunchKernelGGL(k_pe_backward, dim3(blocks_for(n_rays, kWavesPerBlock)), dim3(256), 0, (hipStream_t)stream, dpe,
                       (long long)m_padded, rays_o, rays_d, z, (long long)z_row_stride, (long long)n_rays, S, n_freqs, d_rays_o,
                       d_rays_d);
    return check_launch("k_pe_backward");
}

int mofa_pe_backward_points(const float* dpe, int64_t m_padded, const float* pts, int64_t n_points, int32_t n_freqs, float* d_pts,
                            void* stream) {
    MOFA_REQUIRE(dpe && pts && d_pts && n_points > 0 && n_points <= m_padded, "pe_backward_points: bad arguments");
    MOFA_REQUIRE(n_freqs >= 0 && n_freqs <= MOFA_MAX_PE_FREQS, "pe_backward_points: n_freqs=%d out of [0, %d]", n_freqs, MOFA_MAX_PE_FREQS);
    hipLaunchKernelGGL(k_pe_backward_pts, dim3(blocks_for(n_points, 256)), dim3(256), 0, (hipStream_t)stream, dpe, (long long)m_padded,
                       pts, (long long)n_points, n_freqs, d_pts);
    return check_launch("k_pe_backward_pts");
}

int mofa_composite_backward(const float* raw, const float* z, int64_t z_row_stride, const float* rays_d,
                            const float* noise, int64_t n_rays, int32_t S, int32_t white_bkgd, const float* g_rgb,
                            const float* g_disp, const float* g_acc, const float* g_depth, const float* g_weights,
                            float* d_raw, float* d_rays_d, void* stream) {
    MOFA_REQUIRE(raw && z && rays_d && g_rgb && d_raw, "composite_backward: null pointer");
    MOFA_REQUIRE(n_rays > 0 && S >= 2 && S <= kMaxPasses * 256, "composite_backward: need 2 <= S <= %d (got %d)", kMaxPasses * 256, S);
    const dim3 grid(blocks_for(n_rays, kWavesPerBlock)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MOFA_CB(SPL)                                                                                                \
    hipLaunchKernelGGL((k_composite_backward<SPL>), grid, block, 0, st, raw, z, (long long)z_row_stride, rays_d, noise, \
                       (long long)n_rays, S, white_bkgd, g_rgb, g_disp, g_acc, g_depth, g_weights, d_raw, d_rays_d)
    if (S <= 64) MOFA_CB(1);
    else if (S <= 128) MOFA_CB(2);
    else if (S <= 256) MOFA_CB(4);
    else
        hipLaunchKernelGGL(k_composite_backward_long, grid, block, 0, st, raw, z, (long long)z_row_stride, rays_d, noise, (long long)n_rays,
                           S, white_bkgd, g_rgb, g_disp, g_acc, g_depth, g_weights, d_raw, d_rays_d);
#undef MOFA_CB
    return check_launch("k_composite_backward");
}

}  // extern "C"

// ======================================================================================================
// Weight gradient  dW[n][k] = sum_m G[m][n] * X[m][k]   (training, run_train.py:349) on fp32 MFMA.
// The contraction runs over POINTS, so both operands are read "down the rows" of their panels: a lane (i = l&31,
// g = l>>5) feeds A = G[m0+g][n0+i] and B = X[m0+g][k0+i] as single dwords (fp32 MFMA operands are one VGPR, so no
// packing constraint).  Work is split over M: one workgroup per (output tile, split of the points), placed XCD-aware; each reduces its
// slice of points into a [TN x TK] partial, a second kernel sums the partials (deterministic, no atomics).
// ======================================================================================================
namespace mofa {
namespace {

// The per-layer launch: one workgroup per (unit = output tile, split of the points — wg_split: whole row tiles, the chained form's plan)
template <int TN, int TK>
__global__ __launch_bounds__(256, 2) void k_wgrad(const float* __restrict__ g, const float* __restrict__ x,
                                                  long long m_padded, long long n_points, int n_padded, int k_padded,
                                                  WgSplit sp, float* __restrict__ partial,
                                                  float* __restrict__ bias_partial, int pipe) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MC = WgCfg<TN, TK>::MC;
    const int n_tiles = n_padded / TN;
    // XCD-aware placement (round 6): workgroup b runs on XCD b % 8 (round-robin dispatch).  XCD x takes the splits of ITS row range (the
    // plan's ranges are k_net_chain's), the units of one split on consecutive slots of that XCD: they start together and stream the
    // split's G / X panels in step, so a G panel is fetched once for its 4 readers and an X panel once for its 8 — measured at the
    // benchmark's training sub-batch: 7.78 -> 2.21 GB per launch through the L2s' fabric ports (algorithmic: 1.74), L2 hit 0.19 -> 0.76,
    // +0.3 % (profiles/r06_ab_wgrad_xcd.md).  Same units, same splits, same partial sums: placement only.
    const int units = n_tiles * (k_padded / TK);
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int j = local / units;
    const int unit = local - j * units, split = xcd * sp.nspx + j;
    const int nt = unit % n_tiles, kt = unit / n_tiles;
    int first, count;
    wg_split_rows(sp, split, first, count);
    if (count <= 0) return;
    const long long total_chunks = (n_points + MC - 1) / MC;
    const long long c_begin = (long long)first * (kRowTile / MC);
    long long c_end = c_begin + (long long)count * (kRowTile / MC);
    if (c_end > total_chunks) c_end = total_chunks;
    WgNoHook hook;
    wgrad_unit<TN, TK>(g, x, m_padded, n_points, n_padded, k_padded, nt * TN, kt * TK, c_begin, c_end,
                       partial + (long long)split * n_padded * k_padded, bias_partial ? bias_partial + (long long)split * n_padded : nullptr, pipe, smem, hook);
}

// dst[n][col0 + k] = sum_s partial[s][n][k]   for n < n_out, k < ncols  (natural PyTorch [out, in] gradient layout) — and, from the blocks
// behind those (round 6: it was a launch of its own, 112 per training step of 17 us each), the bias gradient bias_out[n] = sum_s
// bias_partial[s][n]: 64 columns per block, the splits dealt round-robin to 4 thread groups, combined through LDS in a fixed order
// (deterministic; a single thread walking all the splits of its column was a 60 us latency chain).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, int splits, int n_padded,
                                                      int k_padded, int n_out, int ncols, float* __restrict__ dst,
                                                      int ld, int col0, int main_blocks, const float* __restrict__ bias_partial,
                                                      float* __restrict__ bias_out) {
    if ((int)blockIdx.x >= main_blocks) {            // block-uniform: the bias columns
        __shared__ float red[4][64];
        const int col = ((int)blockIdx.x - main_blocks) * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
        float s = 0.f;
        if (col < n_padded)
            for (int sp = grp; sp < splits; sp += 4) s += bias_partial[(long long)sp * n_padded + col];
        red[grp][threadIdx.x & 63] = s;
        __syncthreads();
        if (grp == 0 && col < n_padded) bias_out[col] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        return;
    }
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)n_out * ncols) return;
    const int n = (int)(idx / ncols), k = (int)(idx - (long long)n * ncols);
    const float* p = partial + (long long)n * k_padded + k;
    const long long stride = (long long)n_padded * k_padded;
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= splits; sp += 8) {      // 8 independent loads in flight, summed in the fixed split order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(sp + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sp < splits; ++sp) s += p[(long long)sp * stride];
    dst[(long long)n * ld + col0 + k] = s;
}

// head weight gradient: dst[o][k] = sum_m d_raw[m][off+o] * X[m][k]; blockIdx.x = 16-feature panel of X, blockIdx.y = slice of the
// points.  gridDim.y == 1: results go straight to dst; otherwise to partial[slice][o][k_padded] and k_head_wgrad_reduce sums the
// slices in order (deterministic).
__global__ __launch_bounds__(256) void k_head_wgrad(const float* __restrict__ d_raw, int raw_off, int n_out,
                                                    const float* __restrict__ x, long long m_padded,
                                                    long long n_points, int ncols, float* __restrict__ dst, int ld,
                                                    float* __restrict__ partial, int k_padded) {
    __shared__ float red[256][17];
    const int panel = blockIdx.x;
    const float* base = x + (long long)panel * m_padded * 16;
    const long long per = ((n_points + gridDim.y - 1) / gridDim.y + 255) / 256 * 256;
    const long long m_lo = (long long)blockIdx.y * per, m_hi = (m_lo + per < n_points) ? m_lo + per : n_points;
    for (int o = 0; o < n_out; ++o) {
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.f;
        for (long long m = m_lo + threadIdx.x; m < m_hi; m += 256) {
            const float gw = d_raw[m * 4 + raw_off + o];
            const int sw = (int)(m >> 2) & 3;
            const f32x4* row = (const f32x4*)(base + m * 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 v = row[c ^ sw];
                acc[4 * c + 0] += gw * v.x, acc[4 * c + 1] += gw * v.y, acc[4 * c + 2] += gw * v.z, acc[4 * c + 3] += gw * v.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) red[threadIdx.x][c] = acc[c];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s)
#pragma unroll
                for (int c = 0; c < 16; ++c) red[threadIdx.x][c] += red[threadIdx.x + s][c];
            __syncthreads();
        }
        if (threadIdx.x < 16) {
            if (partial) partial[((long long)blockIdx.y * n_out + o) * k_padded + panel * 16 + threadIdx.x] = red[0][threadIdx.x];
            else if (panel * 16 + (int)threadIdx.x < ncols) dst[(long long)o * ld + panel * 16 + threadIdx.x] = red[0][threadIdx.x];
        }
    }
}

__global__ __launch_bounds__(256) void k_head_wgrad_reduce(const float* __restrict__ partial, int slices, int n_out, int k_padded,
                                                           int ncols, float* __restrict__ dst, int ld) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out * ncols) return;
    const int o = idx / ncols, k = idx - o * ncols;
    float s = 0.f;
    for (int sl = 0; sl < slices; ++sl) s += partial[((long long)sl * n_out + o) * k_padded + k];
    dst[(long long)o * ld + k] = s;
}

// positional-encoding features of every point as panels [k_padded/16][m_padded][16] (the X operand of layer 0's weight gradient, and of
// the wide networks' first layer); pts != NULL: explicit points instead of o + d z
__global__ __launch_bounds__(256) void k_pe_panels(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                   const float* __restrict__ z, long long z_row_stride, const float* __restrict__ pts,
                                                   long long n_points, int S, int pe_feats, int k_padded, long long m_padded,
                                                   float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over m_padded * k_padded
    if (idx >= m_padded * k_padded) return;
    const int e = idx & 3, p = (idx >> 2) & 3;
    const long long rowpanel = idx >> 4;
    const long long m = rowpanel % m_padded;
    const int panel = (int)(rowpanel / m_padded);
    const int k = panel * 16 + ((p ^ ((int)(m >> 2) & 3)) << 2) + e;
    float v = 0.f;
    if (m < n_points && k < pe_feats) {
        const int d = k < 3 ? k : ((k - 3) % 6) % 3;
        float xd;
        if (pts) xd = pts[m * 3 + d];
        else {
            const long long r = m / S;
            const int s = (int)(m - r * S);
            const float zz = z[r * z_row_stride + s];
            xd = __fadd_rn(rays_o[r * 3 + d], __fmul_rn(rays_d[r * 3 + d], zz));
        }
        if (k < 3) v = xd;
        else {
            const int j = k - 3, f = j / 6, rr = j - 6 * f;
            const float arg = xd * (float)(1 << f);
            v = rr < 3 ? sinf(arg) : cosf(arg);
        }
    }
    out[idx] = v;
}

template <int TN, int TK>
int launch_wgrad(const float* g, const float* x, long long m_padded, long long n_points, int n_padded, int k_padded,
                 WgSplit sp, float* partial, float* bias_partial, hipStream_t st) {
    constexpr int PSTR = WgCfg<TN, TK>::MC * 16 + 16;
    const size_t lds = 2 * (size_t)(TN / 16 + TK / 16) * PSTR * sizeof(float);
    const int units = (n_padded / TN) * (k_padded / TK);
    const dim3 grid(8 * sp.nspx * units);                  // (slots of XCD ranges that hold fewer splits — the last, the empty ones — return at once)
    const int prof = mofa_internal_prof_open(st, 3);       // measurement session open? (bench.py --mode train)
    if (prof < 0) return MOFA_EHIP;
    hipLaunchKernelGGL((k_wgrad<TN, TK>), grid, dim3(256), lds, st, g, x, m_padded, n_points, n_padded, k_padded,
                       sp, partial, bias_partial, config().pipe != 0 ? 1 : 0);
    if (prof) mofa_internal_prof_close(st, 3, 2.0 * (double)n_points * (double)n_padded * (double)k_padded);
    return check_launch("k_wgrad");
}

// tile / split plan shared by the workspace query and the launch
struct WgPlan {
    int tn, tk;
    WgSplit sp;
};
inline WgPlan wg_plan(long long m_padded, int n_padded, int k_padded) {
    WgPlan p;
    p.tn = n_padded % 128 == 0 ? 128 : 64;
    p.tk = (p.tn == 128 && k_padded % 256 == 0) ? 256 : (k_padded % 128 == 0 ? 128 : 64);
    p.sp = wg_split(m_padded, (n_padded / p.tn) * (k_padded / p.tk));
    return p;
}

}  // namespace
}  // namespace mofa

extern "C" {

/* partial-sum workspace (floats) needed by mofa_weight_grad for a [n_padded x k_padded] block over n_points */
size_t mofa_weight_grad_workspace_floats(int64_t n_points, int32_t n_padded, int32_t k_padded) {
    if (n_points <= 0 || n_padded <= 0 || k_padded <= 0) return 0;
    const WgPlan p = wg_plan(round_up(n_points, kRowTile), n_padded, k_padded);
    return (size_t)p.sp.total * n_padded * ((size_t)k_padded + 1);     // dW partials + bias partials
}

/* dst[n][col0 + k] = sum_m G[m][n] X[m][k]  (n < n_out, k < ncols); G panels [n_padded/16][Mp][16] (ReLU-masked output
 * gradient), X panels [k_padded/16][Mp][16] (the layer's input), dst row-major with leading dimension ld.
 * bias_out (may be NULL): [n_padded] = sum_m G[m][n], produced by the same pass over G. */
int mofa_weight_grad(const float* g, int32_t n_padded, const float* x, int32_t k_padded, int64_t m_padded,
                     int64_t n_points, int32_t n_out, int32_t ncols, float* dst, int32_t ld, int32_t col0,
                     float* bias_out, float* workspace, void* stream) {
    MOFA_REQUIRE(g && x && dst && workspace, "weight_grad: null pointer");
    MOFA_REQUIRE(n_padded % 64 == 0 && k_padded % 64 == 0 && n_out <= n_padded && ncols <= k_padded && n_points > 0 &&
                     n_points <= m_padded && m_padded % 256 == 0 && col0 >= 0 && col0 + ncols <= ld,
                 "weight_grad: bad shape n_padded=%d k_padded=%d n_out=%d ncols=%d", n_padded, k_padded, n_out, ncols);
    // (the splits cover the row tiles that hold points: the workspace query rounds n_points up the same way)
    const WgPlan p = wg_plan(round_up(n_points, kRowTile), n_padded, k_padded);
    float* bias_partial = bias_out ? workspace + (size_t)p.sp.total * n_padded * k_padded : nullptr;
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define MOFA_WG(TN, TK) \
    rc = launch_wgrad<TN, TK>(g, x, m_padded, n_points, n_padded, k_padded, p.sp, workspace, bias_partial, st)
    if (p.tn == 128 && p.tk == 256) MOFA_WG(128, 256);
    else if (p.tn == 128 && p.tk == 128) MOFA_WG(128, 128);
    else if (p.tn == 128) MOFA_WG(128, 64);
    else if (p.tk == 128) MOFA_WG(64, 128);
    else MOFA_WG(64, 64);
#undef MOFA_WG
    if (rc != MOFA_OK) return rc;
    return mofa_internal_wgrad_reduce(workspace, p.sp.total, n_padded, k_padded, n_out, ncols, dst, ld, col0, bias_out, stream);
}

/* internal (also mofa_net_backward's chained training form, which fills the partials inside its launch): the deterministic second stage */
int mofa_internal_wgrad_reduce(const float* partial, int splits, int n_padded, int k_padded, int n_out, int ncols, float* dst, int ld,
                               int col0, float* bias_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)n_out * ncols;
    const int main_blocks = (int)((total + 255) / 256), bias_blocks = bias_out ? (n_padded + 63) / 64 : 0;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)(main_blocks + bias_blocks)), dim3(256), 0, st, partial, splits,
                       n_padded, k_padded, n_out, ncols, dst, ld, col0, main_blocks, partial + (size_t)splits * n_padded * k_padded, bias_out);
    return check_launch("k_wgrad_reduce");
}

int mofa_head_weight_grad(const float* d_raw, int32_t raw_off, int32_t n_out, const float* x, int32_t k_padded,
                          int64_t m_padded, int64_t n_points, int32_t ncols, float* dst, int32_t ld, void* stream) {
    MOFA_REQUIRE(d_raw && x && dst && k_padded % 16 == 0 && ncols <= k_padded && n_out >= 1 && raw_off + n_out <= 4,
                 "head_weight_grad: bad arguments");
    hipLaunchKernelGGL(k_head_wgrad, dim3(k_padded / 16), dim3(256), 0, (hipStream_t)stream, d_raw, raw_off, n_out, x,
                       (long long)m_padded, (long long)n_points, ncols, dst, ld, (float*)nullptr, k_padded);
    return check_launch("k_head_wgrad");
}

/* internal (mofa_net_backward): the same with the points split over `slices` workgroup rows; `workspace` holds
 * slices * n_out * k_padded floats of partials (the weight-gradient scratch is free at that point) */
int mofa_internal_head_weight_grad_split(const float* d_raw, int32_t raw_off, int32_t n_out, const float* x, int32_t k_padded,
                                         int64_t m_padded, int64_t n_points, int32_t ncols, float* dst, int32_t ld, float* workspace,
                                         void* stream) {
    int slices = (int)((n_points + 16383) / 16384);          // >= 16 k points per slice; ~1000 workgroups at training sizes
    if (slices > 16) slices = 16;
    if (slices <= 1 || !workspace)
        return mofa_head_weight_grad(d_raw, raw_off, n_out, x, k_padded, m_padded, n_points, ncols, dst, ld, stream);
    hipLaunchKernelGGL(k_head_wgrad, dim3(k_padded / 16, slices), dim3(256), 0, (hipStream_t)stream, d_raw, raw_off, n_out, x,
                       (long long)m_padded, (long long)n_points, ncols, dst, ld, workspace, k_padded);
    hipLaunchKernelGGL(k_head_wgrad_reduce, dim3((unsigned)((n_out * ncols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace,
                       slices, n_out, k_padded, ncols, dst, ld);
    return check_launch("k_head_wgrad(split)");
}

int mofa_pe_panels(const float* rays_o, const float* rays_d, const float* z, int64_t z_row_stride, const float* pts, int64_t n_points,
                   int32_t S, int32_t n_freqs, int64_t m_padded, float* out, void* stream) {
    MOFA_REQUIRE(out && (pts || (rays_o && rays_d && z && S > 0)) && n_points <= m_padded, "pe_panels: bad arguments");
    MOFA_REQUIRE(n_freqs >= 0 && n_freqs <= MOFA_MAX_PE_FREQS, "pe_panels: n_freqs=%d out of [0, %d]", n_freqs, MOFA_MAX_PE_FREQS);
    const int kp = (int)round_up(3 + 6 * n_freqs, 64);
    hipLaunchKernelGGL(k_pe_panels, dim3((unsigned)((m_padded * kp + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, z, (long long)z_row_stride, pts, (long long)n_points, S > 0 ? S : 1, 3 + 6 * n_freqs, kp,
                       (long long)m_padded, out);
    return check_launch("k_pe_panels");
}

}  // extern "C"
