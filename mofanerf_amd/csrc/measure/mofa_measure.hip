// MEASUREMENT-ONLY translation unit (gfx950).  Built by tools/build_measure.py into build_arms/libmofanerf_measure.so — never into
// libmofanerf_hip.so, never loaded by the product (mofanerf_amd/lib.py does not know it exists).  It instantiates the SAME layer
// kernel source as the product (../mofa_layer.h) under other policies, plus the scheduling arms earlier rounds measured and
// rejected, so that their numbers stay reproducible (DESIGN.md section 9 has the results):
//   shipped            k_layer<128,..,PIPE> with ShippedPolicy (the control: must reproduce the product bit for bit)
//   plain              the plain K loop (what MOFA_PIPE=0 selects in the product)
//   bn64               the 64-feature tile forced for every width (4 workgroups per CU)
//   unstaged           strided 16-byte epilogue stores instead of the LDS-window epilogue
//   gap2 / gap3        2 / 3 MFMAs between two LDS-DMA requests of the pipelined loop (shipped: derived, 4)
//   setprio1/3         s_setprio around every MFMA block of the plain loop
//   waves3             __launch_bounds__(256, 3): three workgroups per CU need <= 168 VGPRs; the pipelined loop's 197 then spill (1,166
//                      spilled VGPRs -> 19.6 TFLOP/s: the bound, not an option; round 2 measured the PLAIN loop at 3 per CU: +-0.5 %)
//   ring3              3-stage LDS ring twin of the plain loop (k_layer_ring3)
//   persist / persist_dephase   persistent per-layer twin walking the tiles (k_layer_persist)
//   persist_pipe       persistent twin of the SHIPPED kernel (pipelined loop + staged epilogue), no next-tile prefetch
//   persist_pipe_sink  the same with no epilogue at all (WRONG RESULTS): the bound of hiding the store drain in that schedule
//   timeline           per-workgroup / per-panel time stamps (mofa_measure_set_timeline)
//   timeline_sink      the same stamps on a launch whose epilogue stores nothing (WRONG RESULTS): the slot turnaround without a store drain
//   sink_epilogue      WRONG RESULTS BY DESIGN: the tile is discarded instead of stored — what a free epilogue would be worth
// and k_mfma_peak_probe (what the fp32 matrix pipe sustains with no memory traffic at all).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "../mofa_layer.h"

namespace mofa {

// the host-side helpers of mofa_common.h, local to this library (the product's live in mofa_net.hip and are not exported)
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
int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < kMaxDevices ? dev : kMaxDevices - 1;
}
int compute_units(int device) {
    hipDeviceProp_t prop;
    return (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
}
const Config& config() {
    static const Config c{};
    return c;
}

namespace {

// ---- policies ----------------------------------------------------------------------------------------------------------------
struct UnstagedPolicy : ShippedPolicy {
    static constexpr bool kStagedEpilogue = false;
};
template <int GAP>
struct GapPolicy : ShippedPolicy {
    static constexpr int kPipeGap = GAP;
};
template <int PRIO>
struct SetPrioPolicy : ShippedPolicy {
    static constexpr int kSetPrio = PRIO;
};
struct Waves3Policy : ShippedPolicy {
    static constexpr int kMinWaves = 3;
};
struct SinkPolicy : ShippedPolicy {       // timing only: results are NOT computed
    static constexpr bool kSinkEpilogue = true;
    template <class Acc>
    static __device__ __forceinline__ void sink(const Acc& acc, float* y) {
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < (int)(sizeof(acc[0]) / sizeof(acc[0][0])); ++j) s_ += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
        if (s_ == 123.456f) y[0] = s_;      // keeps the accumulators (and with them the K loop) alive
    }
};

// per-workgroup stamps [tiles][8] u64 = {entry, first panel landed (K loop starts), K loop done, stores issued, HW_ID, XCC_ID, clock64
// ticks in the K loop, -} followed by [tiles][64] per-panel stamps of the pipelined loop (kept in LDS until the end: a global store
// at a sync point would sit in front of the next vmcnt(0) wait).  wall_clock64() = the 100 MHz constant clock.
__device__ unsigned long long* g_timeline_dev = nullptr;
struct TimelinePolicy : ShippedPolicy {
    static constexpr int kExtraLds = 512;
    struct Probe {
        unsigned long long* tl;
        unsigned long long* pstamp;
        unsigned long long* pbase;
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tc1 = 0, tc2 = 0;
        int logical, total, tid;
        __device__ __forceinline__ Probe(const LayerArgs& a, int logical_, int tid_, float* lds_behind_stages, int KT)
            : tl(g_timeline_dev), logical(logical_), total(a.total_tiles), tid(tid_) {
            pbase = (tl && KT <= 64) ? (unsigned long long*)lds_behind_stages : nullptr;
            pstamp = pbase;
        }
        __device__ __forceinline__ void entry() {
            if (tl && tid == 0) ts0 = wall_clock64();
        }
        __device__ __forceinline__ void kloop_begin() {
            if (tl && tid == 0) ts1 = wall_clock64(), tc1 = clock64();
        }
        __device__ __forceinline__ void panel() {
            if (pstamp && tid == 0) *pstamp++ = wall_clock64();
        }
        __device__ __forceinline__ void first_panel_landed() {
            if (pbase && tid == 0) pbase[63] = wall_clock64();      // slot 63 is never a panel stamp (KT <= 64)
        }
        __device__ __forceinline__ void kloop_end() {
            if (tl) __builtin_amdgcn_s_barrier();                   // all four waves are out of the K loop
            if (tl && tid == 0) ts2 = wall_clock64(), tc2 = clock64();
        }
        __device__ __forceinline__ void stores_issued() {           // wave 0: its 32 stores per lane are ISSUED (not acknowledged)
            if (!(tl && tid == 0)) return;
            unsigned long long* t = tl + (long long)logical * 8;
            t[0] = ts0, t[1] = ts1, t[2] = ts2, t[3] = wall_clock64();
            t[4] = __builtin_amdgcn_s_getreg(GETREG_IMMED(32 - 1, 0, HW_ID));
            t[5] = __builtin_amdgcn_s_getreg(GETREG_IMMED(4 - 1, 0, 20));        // XCC_ID
            t[6] = tc2 - tc1;
            if (pbase) {
                unsigned long long* pd = tl + (long long)total * 8 + (long long)logical * 64;
                for (int i = 0; i < 64; ++i) pd[i] = pbase[i];
            }
        }
    };
};

// ---- 3-stage-ring twin of k_layer<128,false,true> (MOFA_RING3=1; A/B arm) --------------------------------------------------
// The timeline says a workgroup that is ALONE in its K loop drives the pipe at 75 %, a pair at 94 %.  In k_layer
// the next panel's LDS-DMA is requested half a panel (2,048 MFMA cycles of ONE wave) before the barrier that waits for it; alone
// on its SIMD a wave then sits out the rest of an L2 round trip every panel.  Here the ring has three stages (72 KiB per
// workgroup, still two per CU): panel kt+2 is requested at the top of panel kt and waited for two panels later with a COUNTED
// vmcnt (panel kt+1's loads may stay in flight across the barrier), one barrier per panel.  Same tiles, same arithmetic order,
// bit-identical results.
template <bool PERRAY>
__global__ __launch_bounds__(256, 2) void k_layer_ring3(const LayerArgs a) {
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
    store_tile<NI, NJ, PERRAY>(acc, a.bias, a.bias_rows, a.bias_row_div, a.n_padded, a.y, a.m_padded, m0 + wm * (32 * NJ),
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
__global__ __launch_bounds__(256, 2) void k_layer_persist(const LayerArgs a, int per_xcd_tiles, int dephase) {
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
            store_tile<NI, NJ, true>(acc, a.bias, a.bias_rows, a.bias_row_div, a.n_padded, a.y, a.m_padded,
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

// ---- measurement aid (round 6): does the VECTOR pipe issue packed-fp32 FMAs in the shadow of the fp32 MFMAs? ---------------------------
// The matrix pipe is the ceiling of every kernel in this library (157.3 TFLOP/s; the layer kernel sits at 0.95 of it), and the part's
// vector fp32 peak is the same 157.3 TFLOP/s on a different pipe.  A v_mfma_f32_32x32x2_f32 occupies the matrix core for 16 passes of 4
// cycles; this probe issues V independent v_pk_fma_f32 (64 lanes x 2 FMAs = 256 FLOP each) behind every MFMA of the peak probe's stream
// and reports both rates — the bound of a layer kernel whose workgroups compute a strip of their output tile on the vector pipe.
// KIND: which instruction rides behind the MFMAs — 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_add_u32, 3 v_mov_b32, 4 v_lshl_add_u32 (address
// arithmetic), 5 ds_read_b128 (LDS, waited for once per 64 MFMAs), 6 s_add_u32 (scalar pipe), 7 v_pk_fma_f32 issued as ONE cluster of 8 V
// instructions behind every 8th MFMA instead of V behind each (same count: does grouping save the switches?)
template <int V, int KIND = 0>
__global__ __launch_bounds__(256, 2) void k_mfma_valu_probe(float* __restrict__ out, int iters) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 vacc[16];
    unsigned iacc[16];
    f32x4 lacc[4];
    __shared__ f32x4 lds_src[256];
    lds_src[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const unsigned laddr = (unsigned)(threadIdx.x * 16);
    unsigned sacc = blockIdx.x;
#pragma unroll
    for (int j = 0; j < 16; ++j) vacc[j] = f32x2{0.f, 0.f}, iacc[j] = (unsigned)j;
#pragma unroll
    for (int j = 0; j < 4; ++j) lacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
    float a[8], b[8];
    f32x2 va[4], vb[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h = h * 1664525u + 1013904223u;
        a[e] = __uint_as_float(0x3A000000u | (h & 0x05FFFFFFu) | ((h >> 3) & 0x80000000u));
        h = h * 1664525u + 1013904223u;
        b[e] = __uint_as_float(0x3A000000u | (h & 0x05FFFFFFu) | ((h >> 5) & 0x80000000u));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) va[e] = f32x2{a[e], a[e + 4]}, vb[e] = f32x2{b[e], b[e + 4]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[e]), "v"(b[(e + i) & 7]));
                if constexpr (KIND == 7) {
                    if (i == 7) {
#pragma unroll
                        for (int v = 0; v < 8 * V; ++v)
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(vacc[v & 15]) : "v"(va[(e + v) & 3]), "v"(vb[v & 3]));
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const int j = (i * V + v) & 15;
                        if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(vacc[j]) : "v"(va[(e + v) & 3]), "v"(vb[(i + v) & 3]));
                        if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vacc[j].x) : "v"(a[(e + v) & 7]), "v"(b[(i + v) & 7]));
                        if constexpr (KIND == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(iacc[j]) : "v"(h));
                        if constexpr (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(iacc[j]) : "v"(h));
                        if constexpr (KIND == 4) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(iacc[j]) : "v"(h));
                        if constexpr (KIND == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(lacc[j & 3]) : "v"(laddr));
                        if constexpr (KIND == 6) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) : : "scc");
                    }
                }
            }
        }
        if constexpr (KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(a[e]), "+v"(b[e]));
    }
    float s_ = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_ += acc[i][r];
#pragma unroll
    for (int j = 0; j < 16; ++j) s_ += vacc[j].x + vacc[j].y + (float)iacc[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) s_ += lacc[j].x + lacc[j].w;
    s_ += (float)sacc;
    if (s_ == 123.456f) out[0] = s_;   // keeps the accumulators alive
}

// time stamps of a launch WITHOUT the epilogue's stores: what is left of the slot turnaround when there is nothing to drain
struct TimelineSinkPolicy : TimelinePolicy {
    static constexpr bool kSinkEpilogue = true;
    struct Probe : TimelinePolicy::Probe {
        using TimelinePolicy::Probe::Probe;
        __device__ __forceinline__ void kloop_end() {
            TimelinePolicy::Probe::kloop_end();
            TimelinePolicy::Probe::stores_issued();          // the kernel returns right after the sink: record here
        }
    };
    template <class Acc>
    static __device__ __forceinline__ void sink(const Acc& acc, float* y) { SinkPolicy::sink(acc, y); }
};

// ---- persistent twin of the SHIPPED kernel (pipelined K loop + staged epilogue): 2 workgroups per CU walk the tiles -----------------
// PREFETCH = false: after a tile's epilogue the next tile simply starts (its first LDS-DMA requests queue behind the 32 stores per lane
// in the wave's in-order vmcnt, so the first panel waits for the store drain — but no dispatcher is involved).
template <bool SINK>      // SINK: no epilogue at all (WRONG RESULTS) — the bound of what hiding the store drain in this schedule could reach
__global__ __launch_bounds__(256, 2) void k_layer_persist_pipe(const LayerArgs a, int per_xcd_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 128, BM = kRowTile, NI = 2, NJ = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int KT = a.k1p + a.k2p;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    ShippedPolicy::Probe probe(a, 0, tid, smem, KT);
    for (int it = 0;; ++it) {
        const int local = w + it * wg_per_xcd;
        const int logical = xcd * per_xcd_tiles + local;
        if (local >= per_xcd_tiles || logical >= a.total_tiles) break;
        const int mt = logical / a.n_tiles;
        const long long m0 = (long long)mt * BM;
        const int n0 = (logical - mt * a.n_tiles) * BN;
        f32x16 acc[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        kloop_pipelined<NI, NJ, BM, BN, ShippedPolicy>(a.x1 + m0 * 16, a.k2p ? a.x2 + m0 * 16 : nullptr, a.w + (long long)n0 * 16, a.m_padded * 16,
                                                       (long long)a.n_padded * 16, a.k1p, KT, smem, tid, wave, lane, wm * (32 * NJ), wn * 64, acc, probe);
        if constexpr (SINK) {
            SinkPolicy::sink(acc, a.y);
        } else {
            float* win = smem + wave * 1024;
            if (a.relu) store_tile_staged<NI, NJ, true>(acc, a.bias, a.y, a.m_padded, m0 + wm * (32 * NJ), n0 + wn * 64, lane, win);
            else store_tile_staged<NI, NJ, false>(acc, a.bias, a.y, a.m_padded, m0 + wm * (32 * NJ), n0 + wn * 64, lane, win);
        }
        // every wave is done with the stages (last panel's reads, the epilogue windows) before the next tile's requests overwrite them;
        // only LDS operations need to have completed — NOT the global stores (no vmcnt wait here)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

template <class P, bool PIPE, int BN = 128>
int launch_policy(LayerArgs a, hipStream_t st) {
    a.n_tiles = a.n_padded / BN;
    const long long total = (a.m_padded / kRowTile) * a.n_tiles;
    MOFA_REQUIRE(total > 0 && total < (1ll << 30), "measure: tile count %lld out of range", total);
    a.total_tiles = (int)total;
    const dim3 grid((unsigned)round_up(total, 8)), block(256);
    const size_t lds = 2 * (size_t)(kRowTile + BN) * 16 * sizeof(float) + P::kExtraLds;
    if (a.bias_row_div) hipLaunchKernelGGL((k_layer<BN, false, false, true, PIPE, P>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((k_layer<BN, false, false, false, PIPE, P>), grid, block, lds, st, a);
    return check_launch("k_layer(measure)");
}

int launch_ring3(LayerArgs a, hipStream_t st) {
    constexpr int BN = 128;
    a.n_tiles = a.n_padded / BN;
    const long long total = (a.m_padded / kRowTile) * a.n_tiles;
    a.total_tiles = (int)total;
    const size_t lds3 = 3 * (size_t)(kRowTile + BN) * 16 * sizeof(float);
    static std::atomic<bool> attr3[kMaxDevices];
    const int dev = current_device();
    if (!attr3[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)k_layer_ring3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_layer_ring3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess)
            return check_launch("hipFuncSetAttribute(k_layer_ring3)");
        attr3[dev].store(true, std::memory_order_release);
    }
    const dim3 grid((unsigned)round_up(total, 8)), block(256);
    if (a.bias_row_div) hipLaunchKernelGGL((k_layer_ring3<true>), grid, block, lds3, st, a);
    else hipLaunchKernelGGL((k_layer_ring3<false>), grid, block, lds3, st, a);
    return check_launch("k_layer_ring3");
}

int launch_persist_pipe(LayerArgs a, bool sink, hipStream_t st) {
    constexpr int BN = 128;
    a.n_tiles = a.n_padded / BN;
    const long long total = (a.m_padded / kRowTile) * a.n_tiles;
    a.total_tiles = (int)total;
    const size_t lds = 2 * (size_t)(kRowTile + BN) * 16 * sizeof(float);
    const int cus = compute_units(current_device());
    long long G = 2LL * cus / 8 * 8;
    const int per_xcd_tiles = (int)((total + 7) / 8);
    if (G > 8LL * per_xcd_tiles) G = 8LL * per_xcd_tiles;
    if (sink) hipLaunchKernelGGL(k_layer_persist_pipe<true>, dim3((unsigned)G), dim3(256), lds, st, a, per_xcd_tiles);
    else hipLaunchKernelGGL(k_layer_persist_pipe<false>, dim3((unsigned)G), dim3(256), lds, st, a, per_xcd_tiles);
    return check_launch("k_layer_persist_pipe");
}

int launch_persist(LayerArgs a, int dephase, hipStream_t st) {
    constexpr int BN = 128;
    a.n_tiles = a.n_padded / BN;
    const long long total = (a.m_padded / kRowTile) * a.n_tiles;
    a.total_tiles = (int)total;
    const size_t lds = 2 * (size_t)(kRowTile + BN) * 16 * sizeof(float);
    const int cus = compute_units(current_device());
    long long G = 2LL * cus / 8 * 8;                                  // two resident workgroups per CU, multiple of 8 XCDs
    const int per_xcd_tiles = (int)((total + 7) / 8);
    if (G > 8LL * per_xcd_tiles) G = 8LL * per_xcd_tiles;
    hipLaunchKernelGGL(k_layer_persist, dim3((unsigned)G), dim3(256), lds, st, a, per_xcd_tiles, dephase);
    return check_launch("k_layer_persist");
}

}  // namespace
}  // namespace mofa

using namespace mofa;

#define MOFA_MEASURE_API extern "C" __attribute__((visibility("default")))

MOFA_MEASURE_API const char* mofa_measure_arms(void) {
    return "shipped,plain,bn64,unstaged,gap2,gap3,setprio1,setprio3,waves3,ring3,persist,persist_dephase,persist_pipe,persist_pipe_sink,timeline,timeline_sink,sink_epilogue";
}
MOFA_MEASURE_API const char* mofa_measure_last_error(void) { return g_err; }

/* mofa_layer_forward's arguments behind the name of an arm (see the list at the top of this file) */
MOFA_MEASURE_API int mofa_measure_layer_forward(const char* arm, const float* x1, int32_t k1, const float* x2, int32_t k2,
                                                const float* w_packed, const float* bias, int32_t bias_row_div, int64_t bias_rows, float* y,
                                                int64_t m_padded, int32_t n_padded, int32_t relu, void* stream) {
    MOFA_REQUIRE(arm && x1 && w_packed && bias && y, "measure_layer_forward: null pointer");
    MOFA_REQUIRE(k1 > 0 && k1 % 16 == 0 && k2 >= 0 && k2 % 16 == 0 && (k2 == 0 || x2), "measure_layer_forward: bad K");
    MOFA_REQUIRE(m_padded > 0 && m_padded % kRowTile == 0 && n_padded > 0 && n_padded % 64 == 0, "measure_layer_forward: bad M / N");
    LayerArgs a{};
    a.x1 = x1, a.x2 = x2, a.w = w_packed, a.bias = bias, a.y = y;
    a.k1p = k1 / 16, a.k2p = k2 / 16, a.n_padded = n_padded, a.m_padded = m_padded;
    a.bias_row_div = bias_row_div, a.bias_rows = bias_rows, a.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    const int KT = a.k1p + a.k2p;
    const bool pipe_ok = KT >= 4 && (KT & 1) == 0;
    if (!strcmp(arm, "bn64")) return launch_policy<ShippedPolicy, false, 64>(a, st);
    MOFA_REQUIRE(n_padded % 128 == 0, "measure_layer_forward: arm %s needs n_padded %% 128 == 0", arm);
    if (!strcmp(arm, "plain")) return launch_policy<ShippedPolicy, false>(a, st);
    if (!strcmp(arm, "setprio1")) return launch_policy<SetPrioPolicy<1>, false>(a, st);
    if (!strcmp(arm, "setprio3")) return launch_policy<SetPrioPolicy<3>, false>(a, st);
    if (!strcmp(arm, "ring3")) return launch_ring3(a, st);
    if (!strcmp(arm, "persist")) return launch_persist(a, 0, st);
    if (!strcmp(arm, "persist_dephase")) return launch_persist(a, 1, st);
    MOFA_REQUIRE(pipe_ok, "measure_layer_forward: arm %s runs the pipelined loop: needs an even number of K panels >= 4 (got %d)", arm, KT);
    if (!strcmp(arm, "shipped")) return launch_policy<ShippedPolicy, true>(a, st);
    if (!strcmp(arm, "unstaged")) return launch_policy<UnstagedPolicy, true>(a, st);
    if (!strcmp(arm, "gap2")) return launch_policy<GapPolicy<2>, true>(a, st);
    if (!strcmp(arm, "gap3")) return launch_policy<GapPolicy<3>, true>(a, st);
    if (!strcmp(arm, "waves3")) return launch_policy<Waves3Policy, true>(a, st);
    if (!strcmp(arm, "timeline")) return launch_policy<TimelinePolicy, true>(a, st);
    if (!strcmp(arm, "timeline_sink")) return launch_policy<TimelineSinkPolicy, true>(a, st);
    if (!strcmp(arm, "persist_pipe")) {
        MOFA_REQUIRE(!a.bias_row_div, "measure_layer_forward: arm persist_pipe has no per-ray-bias form");
        return launch_persist_pipe(a, false, st);
    }
    if (!strcmp(arm, "persist_pipe_sink")) return launch_persist_pipe(a, true, st);
    if (!strcmp(arm, "sink_epilogue")) return launch_policy<SinkPolicy, true>(a, st);
    set_error("measure_layer_forward: unknown arm '%s' (have: %s)", arm, mofa_measure_arms());
    return MOFA_EINVAL;
}

/* the stamp buffer of the "timeline" arm: [tiles][8] + [tiles][64] u64 on the device, or NULL = stamps off */
MOFA_MEASURE_API int mofa_measure_set_timeline(unsigned long long* buf) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_timeline_dev), &buf, sizeof(buf)) != hipSuccess) return check_launch("hipMemcpyToSymbol(g_timeline_dev)");
    return MOFA_OK;
}

/* `blocks` workgroups of 4 waves running iters x 64 fp32 MFMAs each (tools/microbench_layer.py --peak) */
template <int KIND>
static int launch_kind_probe(float* out, int32_t blocks, int32_t iters, int32_t v, hipStream_t st) {
    const dim3 g((unsigned)blocks), b(256);
    switch (v) {
        case 1: hipLaunchKernelGGL((k_mfma_valu_probe<1, KIND>), g, b, 0, st, out, iters); break;
        case 2: hipLaunchKernelGGL((k_mfma_valu_probe<2, KIND>), g, b, 0, st, out, iters); break;
        case 4: hipLaunchKernelGGL((k_mfma_valu_probe<4, KIND>), g, b, 0, st, out, iters); break;
        default: set_error("mfma_kind_probe: per_mfma must be 1, 2 or 4"); return MOFA_EINVAL;
    }
    return check_launch("k_mfma_valu_probe<kind>");
}
MOFA_MEASURE_API int mofa_measure_mfma_kind_probe(float* out, int32_t blocks, int32_t iters, int32_t per_mfma, int32_t kind, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
        case 0: return launch_kind_probe<0>(out, blocks, iters, per_mfma, st);
        case 1: return launch_kind_probe<1>(out, blocks, iters, per_mfma, st);
        case 2: return launch_kind_probe<2>(out, blocks, iters, per_mfma, st);
        case 3: return launch_kind_probe<3>(out, blocks, iters, per_mfma, st);
        case 4: return launch_kind_probe<4>(out, blocks, iters, per_mfma, st);
        case 5: return launch_kind_probe<5>(out, blocks, iters, per_mfma, st);
        case 6: return launch_kind_probe<6>(out, blocks, iters, per_mfma, st);
        case 7: return launch_kind_probe<7>(out, blocks, iters, per_mfma, st);
        default: set_error("mfma_kind_probe: kind must be 0..7"); return MOFA_EINVAL;
    }
}

MOFA_MEASURE_API int mofa_measure_mfma_valu_probe(float* out, int32_t blocks, int32_t iters, int32_t valu_per_mfma, void* stream) {
    const dim3 g((unsigned)blocks), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (valu_per_mfma) {
        case 0: hipLaunchKernelGGL(k_mfma_valu_probe<0>, g, b, 0, st, out, iters); break;
        case 1: hipLaunchKernelGGL(k_mfma_valu_probe<1>, g, b, 0, st, out, iters); break;
        case 2: hipLaunchKernelGGL(k_mfma_valu_probe<2>, g, b, 0, st, out, iters); break;
        case 4: hipLaunchKernelGGL(k_mfma_valu_probe<4>, g, b, 0, st, out, iters); break;
        case 6: hipLaunchKernelGGL(k_mfma_valu_probe<6>, g, b, 0, st, out, iters); break;
        case 8: hipLaunchKernelGGL(k_mfma_valu_probe<8>, g, b, 0, st, out, iters); break;
        case 12: hipLaunchKernelGGL(k_mfma_valu_probe<12>, g, b, 0, st, out, iters); break;
        case 15: hipLaunchKernelGGL(k_mfma_valu_probe<15>, g, b, 0, st, out, iters); break;
        case 16: hipLaunchKernelGGL(k_mfma_valu_probe<16>, g, b, 0, st, out, iters); break;
        default: set_error("mfma_valu_probe: valu_per_mfma must be one of 0 1 2 4 6 8 12 15 16"); return MOFA_EINVAL;
    }
    return check_launch("k_mfma_valu_probe");
}

// Does a RESIDENT workgroup stay where it was dispatched while several PROCESSES share the device?  (round 6: profiles/r06_shared_device_chain.md —
// k_net_chain keeps producer and consumer of a row tile on one XCD, read once at entry.)  Every workgroup holds a slot like k_net_chain's (two per CU:
// 64 KiB of LDS) for `ticks` of the 100 MHz real-time counter and polls its XCC_ID / HW_ID registers and the clock; it reports
// {XCC_ID at entry, XCC_ID changes, HW_ID changes (CU / SE / SIMD ...), longest gap between two polls (= time off the chip), polls, last XCC_ID,
//  entry time, exit time}.
__global__ __launch_bounds__(256, 2) void k_xcc_watch(unsigned long long* __restrict__ out, unsigned long long ticks) {
    extern __shared__ float smem_watch[];
    if (threadIdx.x == 255) smem_watch[16 * 1024 - 1] = 0.f;      // (the allocation is what matters)
    unsigned x0, h0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x0));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h0));
    x0 &= 7u;
    unsigned xl = x0, hl = h0, xch = 0, hch = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long prev = t0, gap = 0, polls = 0, t = t0;
    while (t - t0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        t = __builtin_amdgcn_s_memrealtime();
        if (t - prev > gap) gap = t - prev;
        prev = t;
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        x &= 7u;
        if (x != xl) ++xch, xl = x;
        if (h != hl) ++hch, hl = h;
        ++polls;
    }
    if (threadIdx.x == 0) {
        unsigned long long* o = out + (size_t)blockIdx.x * 8;
        o[0] = x0, o[1] = xch, o[2] = hch, o[3] = gap, o[4] = polls, o[5] = xl, o[6] = t0, o[7] = t;
    }
}
MOFA_MEASURE_API int mofa_measure_xcc_watch(unsigned long long* out, int32_t blocks, unsigned long long ticks, void* stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)k_xcc_watch, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return check_launch("hipFuncSetAttribute(k_xcc_watch)");
        attr = true;
    }
    hipLaunchKernelGGL(k_xcc_watch, dim3((unsigned)blocks), dim3(256), 64 * 1024, (hipStream_t)stream, out, ticks);
    return check_launch("k_xcc_watch");
}

MOFA_MEASURE_API int mofa_measure_mfma_peak_probe(float* out, int32_t blocks, int32_t iters, int32_t random_operands, void* stream) {
    hipLaunchKernelGGL(k_mfma_peak_probe, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, iters, random_operands);
    return check_launch("k_mfma_peak_probe");
}
