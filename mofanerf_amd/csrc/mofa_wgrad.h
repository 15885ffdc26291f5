// The weight-gradient tile  dW[n][k] = sum_m G[m][n] * X[m][k]  (training, run_train.py:349) on fp32 MFMA, as a device function shared by
// the per-layer kernel (k_wgrad, mofa_bwd.hip) and the chained training backward (k_net_chain_train, mofa_mlp.hip), whose weight-gradient
// queue entries are exactly these tiles — same stages, same MFMA order, same split of the points (wg_split, mofa_common.h): bit-identical.
// The contraction runs over POINTS, so both operands are read "down the rows" of their panels: a lane (i = l&31, g = l>>5) feeds
// A = G[m0+g][n0+i] and B = X[m0+g][k0+i] as single dwords (fp32 MFMA operands are one VGPR, so no packing constraint).  Work is split
// over M: every unit reduces its slice of points into a [TN x TK] partial, a second kernel sums the partials (deterministic, no atomics).
#pragma once
#include "mofa_common.h"

namespace mofa {
namespace {

__device__ __forceinline__ void glds16b(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same with a cache policy (AUX = 16: `sc1`, agent scope — served by the XCD's L2, never by this CU's vector L1): the form for G panels
// another workgroup of the SAME launch has just written (k_net_chain_train; mofa_layer.h glds16_policy says why)
template <int AUX>
__device__ __forceinline__ void glds16b_policy(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

// points per pipeline stage: 16 for the 128 x 256 tile (24 KiB / stage, like the forward kernel), 32 for the small tiles
template <int TN, int TK>
struct WgCfg {
    static constexpr int MC = (TN == 128 && TK == 256) ? 16 : 32;
};

// No-op hook (the per-layer kernel).  The chained training backward passes its queue probe: `panel()` is called behind every
// workgroup-wide "my requests have landed" point of the chunk loop (vmcnt(0) + barrier), the first of which is where the PREVIOUS
// tile's stores are known to have reached the L2 (k_net_chain_train, mofa_mlp.hip).
struct WgNoHook {
    __device__ __forceinline__ void panel() {}
};

// One [TN x TK] output tile (features [n0, n0 + TN) x inputs [k0, k0 + TK)) of dW = G^T X over the chunks [c_begin, c_end) of MC points:
// `out` = this split's partial [n_padded][k_padded], `bias_out` = this split's bias partial [n_padded] (NULL, or written by the k0 == 0 tile).
// GAUX: cache policy of the G panels' LDS-DMA requests (0 = default; 16 = sc1 for the chained launch).
template <int TN, int TK, class Hook, int GAUX = 0>
__device__ __forceinline__ void wgrad_unit(const float* __restrict__ g, const float* __restrict__ x, long long m_padded, long long n_points,
                                           int n_padded, int k_padded, int n0, int k0, long long c_begin, long long c_end,
                                           float* __restrict__ out, float* __restrict__ bias_out, int pipe, float* smem, Hook& hook) {
    constexpr int MC = WgCfg<TN, TK>::MC;
    constexpr int PSTR = MC * 16 + 16;              // LDS stride between 16-feature panels (+16: bank spread)
    constexpr int GP = TN / 16, XP = TK / 16;       // panels per stage
    constexpr int STAGE = (GP + XP) * PSTR;
    constexpr int NI = TN / 64, NJ = TK / 64;       // 2 x 2 waves, wave tile (TN/2) x (TK/2)
    const int tid = threadIdx.x;
    int lane = tid & 63;
    // inside the chained kernel this unit is one of two tile forms in a loop: keep its lane constants (fragment addresses, request offsets)
    // from being hoisted to the kernel's entry, where they would sit in vector registers across the OTHER form's 128 accumulators
    if constexpr (GAUX != 0) asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wk = wave >> 1;

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one stage = GP + XP panel pieces of MC rows x 64 B, moved as 1 KiB (16-row) wave-instructions.  Each wave owns every 4th
    // piece; its source pointers are formed ONCE and advanced by a constant per chunk (chunks are consumed in order), so staging
    // costs one 64-bit add per piece instead of a multiply-add chain.
    constexpr int PPP = MC / 16;                    // 1 KiB pieces per panel
    constexpr int PIECES = (GP + XP) * PPP;
    constexpr int NQ = (PIECES + 3) / 4;
    const float* srcq[NQ];
    int dstq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int pc = wave + 4 * q;
        const int panel = pc / PPP, part = pc % PPP;
        const float* src = (panel < GP) ? g + ((long long)(n0 / 16 + panel) * m_padded + part * 16) * 16
                                        : x + ((long long)(k0 / 16 + panel - GP) * m_padded + part * 16) * 16;
        srcq[q] = src + c_begin * (long long)(MC * 16) + lane * 4;
        dstq[q] = panel * PSTR + part * 256;
    }
    auto stage = [&](int buf) {                     // stages the NEXT chunk (called once per chunk, in order)
        float* base = smem + buf * STAGE;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (wave + 4 * q < PIECES) {
                if (GAUX != 0 && (wave + 4 * q) / PPP < GP) glds16b_policy<GAUX>(srcq[q], base + dstq[q]);
                else glds16b(srcq[q], base + dstq[q]);
            }
            srcq[q] += MC * 16;
        }
    };
    float bsum[NI];                                 // bias gradient rides along: sum_m G[m][n] for this lane's features
#pragma unroll
    for (int i = 0; i < NI; ++i) bsum[i] = 0.f;

    // Operand fragments.  An fp32 MFMA operand is ONE value per lane and the contraction index here is the point, so which
    // FEATURE a lane's row stands for is free: lane li of feature block i takes feature NI * li + i (and NJ * li + j on the X
    // side).  The NI (NJ) values a lane needs for one point are then CONTIGUOUS in the panel row — one ds_read_b64 / b128 per
    // operand and point instead of NI + NJ scalar reads (the scalar form spent ~20 address VALU ops per 8 MFMAs, and VALU time
    // adds to MFMA time on this chip: DESIGN.md 3.1).  The accumulation order over the points is unchanged: results are
    // bit-identical to the scalar-read kernel.
    typedef float fvecA __attribute__((ext_vector_type(NI)));
    typedef float fvecB __attribute__((ext_vector_type(NJ)));
    const int li = lane & 31, gsel = lane >> 5;
    const int nl0 = wn * (TN / 2) + NI * li, kl0 = wk * (TK / 2) + NJ * li;      // first feature of this lane's fragment
    // LDS float offsets of this lane's fragment for point `gsel` of a chunk, one per value of the row swizzle ((point >> 2) & 3 —
    // a compile-time constant per unrolled point pair): every read below is base[sw] + an IMMEDIATE, so the K loop carries no
    // address arithmetic (the buffer is a template constant too: the chunk loop is unrolled by two).
    int a_base[4], b_base[4];
#pragma unroll
    for (int sw = 0; sw < 4; ++sw) {
        a_base[sw] = (nl0 >> 4) * PSTR + (nl0 & 3) + gsel * 16 + ((((nl0 >> 2) & 3) ^ sw) << 2);
        b_base[sw] = GP * PSTR + (kl0 >> 4) * PSTR + (kl0 & 3) + gsel * 16 + ((((kl0 >> 2) & 3) ^ sw) << 2);
    }
    const bool want_bias = bias_out && k0 == 0 && wk == 0;        // wave-uniform: only these waves carry the bias sums
    // One loop body (two instantiations made hipcc keep the accumulators in two register sets and spill).  Rows beyond the
    // batch (only the launch's last chunk can have them) are zeroed IN LDS before the chunk is consumed, so the loop never masks;
    // the bias sums sit behind a wave-uniform scalar branch.
    auto compute = [&](int cur) {
        const float* base = smem + cur * STAGE;                       // 8 VALU adds per chunk fold this into the fragment addresses
        auto load = [&](int mp, float (&a)[NI], float (&b)[NJ]) {
            const int sw = (mp >> 1) & 3;                             // ((2 mp + gsel) >> 2) & 3: m0 is a multiple of 16
            if constexpr (NI == 1) a[0] = base[a_base[sw] + mp * 32];
            else {
                const fvecA v = *(const fvecA*)(base + a_base[sw] + mp * 32);
#pragma unroll
                for (int i = 0; i < NI; ++i) a[i] = v[i];
            }
            if constexpr (NJ == 1) b[0] = base[b_base[sw] + mp * 32];
            else {
                const fvecB v = *(const fvecB*)(base + b_base[sw] + mp * 32);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = v[j];
            }
        };
        // fragments of point pair mp + 1 are fetched while the MFMAs of pair mp issue; the scheduling barrier keeps hipcc from
        // hoisting all MC/2 fetches to the top of the chunk
        float a[2][NI], b[2][NJ];
        load(0, a[0], b[0]);
#pragma unroll
        for (int mp = 0; mp < MC / 2; ++mp) {
            const int cb = mp & 1;
            if (mp + 1 < MC / 2) load(mp + 1, a[cb ^ 1], b[cb ^ 1]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][i], b[cb][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (want_bias) {   // wave-uniform: the one column of waves that owns the bias sums re-reads its G fragments (LDS, 8 reads)
#pragma unroll
            for (int mp = 0; mp < MC / 2; ++mp) {
                const int sw = (mp >> 1) & 3;
#pragma unroll
                for (int i = 0; i < NI; ++i) bsum[i] += base[a_base[sw] + mp * 32 + i];
            }
        }
    };
    bool done = false;
    if constexpr (TN == 128 && TK == 256) {
        // Software-pipelined chunk loop (the forward kernel's kloop_pipelined, DESIGN.md 3.1, transposed to this contraction): a
        // chunk's 64 MFMAs are two halves of four point pairs; the first half carries the fragment reads of the second, the second
        // half carries the reads of the NEXT chunk's first half and the six LDS-DMA requests of the chunk after that (one per four
        // MFMAs, after the reads), with the workgroup barrier between the halves.  Same stages, same MFMA order: bit-identical.
        // Needs an even chunk count >= 4 and no ragged last chunk (the plain loop below handles everything else).
        static_assert(MC == 16 && PIECES % 4 == 0, "pipelined weight-gradient loop: 16-point chunks, whole rounds of pieces");
        const long long nch = c_end - c_begin;
        if (pipe && nch >= 4 && (nch & 1) == 0 && c_end * MC <= n_points) {
            struct WFrag {
                float a[4][NI], b[4][NJ];
            };
            auto readh = [&](int st, int h, WFrag& f) {
                const float* base = smem + st * STAGE;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int mp = 4 * h + t, sw = (mp >> 1) & 3;
                    const fvecA va = *(const fvecA*)(base + a_base[sw] + mp * 32);
                    const fvecB vb = *(const fvecB*)(base + b_base[sw] + mp * 32);
#pragma unroll
                    for (int i = 0; i < NI; ++i) f.a[t][i] = va[i];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) f.b[t][j] = vb[j];
                }
            };
            auto mfmah = [&](const WFrag& f) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t][i], f.b[t][j], acc[i][j], 0, 0, 0);
            };
            // wave-uniform source bases (SGPRs) + one per-lane offset: a request costs no vector address arithmetic
            const float* sq[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int pc = wave + 4 * q;
                sq[q] = (pc < GP ? g + (long long)(n0 / 16 + pc) * m_padded * 16 : x + (long long)(k0 / 16 + pc - GP) * m_padded * 16) +
                        c_begin * (long long)(MC * 16);
            }
            // (the lane's BYTE offset, opaque and re-opaqued per request: a zero-extension hipcc hoists out of the loop hides the 32-bit offset
            //  from the instruction selector, which then forms base + offset with 64-bit VECTOR adds in the MFMA stream — 18 per 128 MFMAs here
            //  — and a vector instruction in the shadow of an MFMA takes matrix-pipe time on this part: profiles/r06_probe_dual_issue.md;
            //  the bases are pinned in SGPRs across their per-chunk step for the same reason)
            unsigned loff = (unsigned)lane * 16u;
            asm volatile("" : "+v"(loff));
            auto request = [&](int st) {
                float* base = smem + st * STAGE;
                asm volatile("" : "+v"(loff));
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float* src = (const float*)((const char*)sq[q] + loff);
                    // (128 x 256 tile: pieces 0..7 are the G panels, i.e. q = 0, 1 for every wave — a compile-time choice)
                    if constexpr (GAUX != 0) {
                        static_assert(GP % 4 == 0 && PPP == 1, "G pieces must fill whole rounds of the four waves");
                        if (q < GP / 4) glds16b_policy<GAUX>(src, base + dstq[q]);
                        else glds16b(src, base + dstq[q]);
                    } else {
                        glds16b(src, base + dstq[q]);
                    }
                    sq[q] += MC * 16;
                    asm volatile("" : "+s"(sq[q]));
                }
            };
            auto bias_add = [&](const WFrag& f) {     // one column of waves owns the bias sums: a REAL wave-uniform branch
                if (want_bias) {
                    asm volatile("" ::: "memory");    // (not if-converted: the other waves must not pay for the adds)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < NI; ++i) bsum[i] += f.a[t][i];
                }
            };
            auto half_a = [&](int st, WFrag& cur, WFrag& nxt) {
                __builtin_amdgcn_sched_barrier(0);
                readh(st, 1, nxt);
                mfmah(cur);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
                __builtin_amdgcn_sched_barrier(0);
                bias_add(cur);
            };
            auto sync_point = [&]() {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                hook.panel();
            };
            auto half_b = [&](int st, bool do_request, bool do_read, WFrag& cur, WFrag& nxt) {
                __builtin_amdgcn_sched_barrier(0);
                if (do_read) readh(st ^ 1, 0, nxt);
                if (do_request) request(st);
                mfmah(cur);
                if (do_read) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if (do_request) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
                __builtin_amdgcn_sched_barrier(0);
                bias_add(cur);
            };
            WFrag fa, fb;
            request(0);
            request(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
            __builtin_amdgcn_s_barrier();
            readh(0, 0, fa);
            const int nch_u = __builtin_amdgcn_readfirstlane((int)nch);     // (workgroup-uniform: a scalar loop counter, not a vector compare per chunk pair)
            for (int c = 0; c + 2 < nch_u; c += 2) {
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
            done = true;
        }
    }
    if (!done && c_begin < c_end) {
        stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        hook.panel();
        for (long long c = c_begin; c < c_end; ++c) {
            const int cur = (int)((c - c_begin) & 1);
            if (c + 1 < c_end) stage(cur ^ 1);
            const long long m0 = c * MC;
            if (m0 + MC > n_points) {  // the batch ends inside this chunk (block-uniform, at most once per launch): zero the dead G rows
                float* gsm = smem + cur * STAGE;
                for (int t = tid; t < GP * MC * 16; t += 256) {
                    const int panel = t / (MC * 16), rem = t - panel * (MC * 16);
                    if (m0 + (rem >> 4) >= n_points) gsm[panel * PSTR + rem] = 0.f;
                }
                __syncthreads();
            }
            compute(cur);
            __syncthreads();
        }
    }
    if (want_bias) {       // one column of workgroups owns the bias partials [split][n_padded]
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);      // even + odd points
            if (lane < 32) bias_out[n0 + wn * (TN / 2) + NI * li + i] = tot;
        }
    }
    // partial[split][n][k], row-major [n_padded][k_padded]; accumulator row rr <-> feature NI * rr + i, column li <-> inputs
    // NJ * li .. + NJ - 1: one NJ-wide store per (feature, lane) — 32 lanes cover NJ * 128 contiguous bytes
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * gsel;
            const int n = n0 + wn * (TN / 2) + NI * rr + i;
            const int k = k0 + wk * (TK / 2) + NJ * li;
            float* dst = out + (long long)n * k_padded + k;
            if constexpr (NJ == 1) dst[0] = acc[i][0][r];
            else {
                fvecB v;
#pragma unroll
                for (int j = 0; j < NJ; ++j) v[j] = acc[i][j][r];
                *(fvecB*)dst = v;
            }
        }
}

}  // namespace
}  // namespace mofa
