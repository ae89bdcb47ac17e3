// mac_sys.hip -- k_mac_sys: the partition sums as an S-stage systolic chain that walks FORWARD in time.
//
//     y[t] = sum_{p < B} x[t - p] h[p]      per bin, partitions in the reference's order p = 0 .. B-1
//                                           (brutefir/brutefir.cpp:288-299; convolve / convolve_add,
//                                           fftw_convolver.cpp:1464-1525 / :2160-2220), one fused multiply-add chain per output.
//
// k_mac_stream (kernels.hip) keeps all 32 partitions of a bin AND 32 running sums in one lane (168 registers) and therefore
// has to walk the delay line backwards, a whole range of blocks at a time; the fp64 kernels share operands through LDS
// tiles or keep 16 running sums per lane.  All of them issue their FMAs with a fresh accumulator AND a fresh operand pair
// per instruction, the slowest operand pattern of the SIMD (scripts/ubench/fma_chain.hip: 1.35 ns per wave-FMA with 16 sums
// per wave, 0.95-1.05 with two to four; fp64 2.78 against 1.9).  Here S lanes of one 16-lane DPP row share a bin:
//
//     lane j of a group ("stage j") holds h[j PL .. (j+1) PL - 1] and a window of PL delay-line values, j (PL + 1) blocks
//     behind stage 0's.
//
// In slot t stage 0 starts y[t]'s chain from zero and runs its first PL links; stage j continues the chain of y[t - j]
// from the partial sum stage j-1 finished one slot earlier; stage S-1 finishes y[t - S + 1] and stores it.  Values move
// one stage up by DPP (row_shr -- an ordinary vector instruction, see SysLanes; a first build split a bin across the two
// HALVES of a wave and paid 5-6 FMA slots per v_permlane32_swap, profiles/r03_fusion_bound.txt): the partial sum (stage j -> j+1),
// and the delay-line value stage j drops from its window, which is exactly stage j+1's newest value of the next slot, so
// only stage 0 ever reads the delay line from memory.  Every output is ONE chain p = 0 .. S PL - 1 with the fma sequence
// of every other MAC kernel here: bit-identical sums.  Partitions >= nblk hold h = 0 (an exact no-op).
//
//   * two accumulators per wave: the SIMD's fast operand pattern; 2 PL (h) + 2 (PL + D) (window + prefetch) registers
//     (x 2 for double): 80-90 (fp32, PL = 16) instead of 168;
//   * the walk is forward: y[t] leaves S - 1 slots after x[t] has arrived, a run of R outputs costs R + S - 1 slots and
//     S (PL + 1) window loads of history: short runs are cheap;
//   * the same kernel serves fp32 (pairs layout) and fp64 (pairs, or the reference's grouped layout: re and im 32 bytes
//     apart), S PL >= B partitions: up to 256 (sixteen stages: a whole DPP row per bin).
// Bin 0 (DC | Nyquist: two independent real sums) is left to the first workgroups of the grid, as in k_mac_stream.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

#include "fft_lds.h"

// -DBFIR_SYS_EXP=bits (timing experiments only, results garbage): 1 no partial-sum hand-over, 2 no relay of x to the next
// stage, 4 no stores, 8 no loads, 16 no negated operand (all FMAs in the short encoding), 32 no tail test per slot.
// Never defined in the product library.
#ifndef BFIR_SYS_EXP
#define BFIR_SYS_EXP 0
#endif

namespace bfir {

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Addressing through buffer descriptors (4 SGPRs per stream, wave-uniform) + ONE 32-bit byte offset per lane + a scalar
// offset per access: a channel's delay line (ring x N reals) and product spectra stay below 4 GiB, and a 64-bit address
// pair per access would cost this kernel the registers that decide its occupancy.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sys_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long u = (unsigned long long)p;     // workgroup-uniform by construction
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

// One complex spectrum value of one bin.  (re, im) pairs (fp32 engines; fp64 engines on the run kernels): ONE access of 8 / 16
// bytes.  The reference's grouped layout (fftw_convolver.cpp:1583-1607: 4 re | 4 im per group; the other fp64 engines): two
// 8-byte accesses 32 bytes apart.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Cx { T x, y; };
template <typename T, bool ILV, int AUX> __device__ __forceinline__ Cx<T> sys_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    Cx<T> c;
    if constexpr (sizeof(T) == 4) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
        c.x = __uint_as_float(v.x); c.y = __uint_as_float(v.y);
    } else if constexpr (ILV) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
        c.x = __hiloint2double((int)v.y, (int)v.x); c.y = __hiloint2double((int)v.w, (int)v.z);
    } else {
        const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
        const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, voff + 32u, soff, AUX);
        c.x = __hiloint2double((int)a.y, (int)a.x); c.y = __hiloint2double((int)b.y, (int)b.x);
    }
    return c;
}
template <typename T, bool ILV, int AUX> __device__ __forceinline__ void sys_store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, Cx<T> c)
{
    if constexpr (sizeof(T) == 4) {
        u32x2 v; v.x = __float_as_uint(c.x); v.y = __float_as_uint(c.y);
        __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, AUX);
    } else if constexpr (ILV) {
        u32x4 v;
        v.x = (unsigned)__double2loint(c.x); v.y = (unsigned)__double2hiint(c.x);
        v.z = (unsigned)__double2loint(c.y); v.w = (unsigned)__double2hiint(c.y);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, AUX);
    } else {
        u32x2 a, b;
        a.x = (unsigned)__double2loint(c.x); a.y = (unsigned)__double2hiint(c.x);
        b.x = (unsigned)__double2loint(c.y); b.y = (unsigned)__double2hiint(c.y);
        __builtin_amdgcn_raw_buffer_store_b64(a, r, voff, soff, AUX);
        __builtin_amdgcn_raw_buffer_store_b64(b, r, voff + 32u, soff, AUX);
    }
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Which lane is which stage: the stages of a bin sit 16 / S lanes apart inside a 16-lane DPP row (stage-major: the first
// 16 / S lanes of a row are stage 0 of its 16 / S bins), so "from the stage below" is row_shr:(16 / S) and the two selects
// of the exchange come for free from the DPP controls: without bound_ctrl a lane whose source lies outside the row -- a
// stage 0 -- is not written and keeps the value it loaded itself; with bound_ctrl it reads zero, the start of a chain.
// One v_mov_b32_dpp per exchanged register and nothing else.  (The first build had the stages in ADJACENT lanes, row_shr:1,
// and paid a v_cndmask per register or an exec-mask round trip on top: cfg5's MAC 1.64 -> 1.55 ms, profiles/r03_mac_sys.txt.)
template <int S> struct SysLanes {
    static_assert(S >= 2 && S <= 16 && (S & (S - 1)) == 0, "stages per bin");
    static constexpr int BPR = 16 / S;                           // bins per row
    static constexpr int SHR = 0x110 + BPR;                      // dpp_ctrl row_shr:BPR
    static __device__ __forceinline__ int stage(int tid) { return (tid & 15) / BPR; }
    // bin inside the workgroup's column of 256 / S bins
    static __device__ __forceinline__ int bin(int tid) { return (tid >> 4) * BPR + (tid & (BPR - 1)); }
};

// stage 0 ? own : v of the stage below
template <int S> __device__ __forceinline__ unsigned relay1(unsigned own, unsigned v)
{
    return __builtin_amdgcn_update_dpp(own, v, SysLanes<S>::SHR, 0xf, 0xf, false);
}
// stage 0 ? 0 : v of the stage below
template <int S> __device__ __forceinline__ unsigned hand1(unsigned v)
{
    return __builtin_amdgcn_update_dpp(v, v, SysLanes<S>::SHR, 0xf, 0xf, true);
}
template <int S> __device__ __forceinline__ float relay_t(float own, float v)
{
    return __uint_as_float(relay1<S>(__float_as_uint(own), __float_as_uint(v)));
}
template <int S> __device__ __forceinline__ double relay_t(double own, double v)
{
    const unsigned lo = relay1<S>((unsigned)__double2loint(own), (unsigned)__double2loint(v));
    const unsigned hi = relay1<S>((unsigned)__double2hiint(own), (unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}
template <int S> __device__ __forceinline__ float hand_t(float v) { return __uint_as_float(hand1<S>(__float_as_uint(v))); }
template <int S> __device__ __forceinline__ double hand_t(double v)
{
    const unsigned lo = hand1<S>((unsigned)__double2loint(v)), hi = hand1<S>((unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}

// One group of G = PL + D slots.  The window and the prefetch queue are ONE rotating set of G register pairs: in slot u
// entry u holds the newest value, entry u - p the one p slots old, and the load issued in slot u goes into entry u + D,
// whose old content in stage j (p = PL slots old, last used in slot u - 1) is first handed to stage j + 1 as ITS entry
// u + 1: the newest value of its next slot.  All lanes of a group load the same address (the copies of the stages above
// 0 are overwritten by that hand-down before they are looked at), so the slot part of the address is a scalar.  No
// register copies; the entries rotate with the slot index, hence the unrolled group; only the first `rem` slots run
// (the last group of a run).  ONE instantiation on purpose: a second copy of the group (without the tail test) makes the
// register allocator reconcile the window registers at the joins and costs 16+ registers (measured); the tail test is a
// scalar compare-and-branch per slot.  tau0 = output index stage 0 works on in slot u = 0.
template <typename T, bool ILV, int S, int PL, int D>
__device__ __forceinline__ void sys_group(Cx<T> (&W)[PL + D], const T (&hr)[PL], const T (&hi)[PL], T &hand_r, T &hand_i,
                                          __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ry, unsigned &so, unsigned xstep,
                                          unsigned xwrap, unsigned kv, unsigned kvs, int tau0, int t_lo,
                                          bool store_lane, int rem, const MacArgs &a)
{
    constexpr int G = PL + D;
    static_assert(D >= 2, "entry u + 1 and entry u + D are different registers");
    static_assert(G >= S - 1, "only a run's first group has slots without a finished output");
    static_for<0, G>([&](auto U) {
        constexpr int u = decltype(U)::value;
#if !(BFIR_SYS_EXP & 32)
        if (u >= rem) return;                                    // wave-uniform
#endif
#if !(BFIR_SYS_EXP & 2)
        {   // stage j's x[newest - PL] -> stage j + 1's newest of the next slot; stage 0 keeps what it loaded
            W[(u + 1) % G].x = relay_t<S>(W[(u + 1) % G].x, W[(u + D) % G].x);
            W[(u + 1) % G].y = relay_t<S>(W[(u + 1) % G].y, W[(u + D) % G].y);
        }
#endif
#if !(BFIR_SYS_EXP & 8)
        W[(u + D) % G] = sys_load<T, ILV, (BFIR_NT_X & 2) ? 2 : 0>(rx, kv, so);   // stage 0's x[tau + D], D slots ahead
#else
        W[(u + D) % G].x += (T)1;
#endif
        so += xstep; so = so >= xwrap ? so - xwrap : so;         // scalar
        T ar = hand_r, ai = hand_i;                              // stage 0: zero; stage j: the partial sum from below
        static_for<0, PL>([&](auto P) {
            constexpr int p = decltype(P)::value;
            const Cx<T> w = W[(u - p + G) % G];                  // p slots old
            ar = fma_t(w.x, hr[p], ar); ai = fma_t(w.x, hi[p], ai);
#if !(BFIR_SYS_EXP & 16)
            ar = fma_t(-w.y, hi[p], ar); ai = fma_t(w.y, hr[p], ai);
#else
            ar = fma_t(w.y, hi[p], ar); ai = fma_t(w.y, hr[p], ai);
#endif
            // keep the two chains interleaved (an independent instruction between dependent ones); a scheduling
            // barrier, not an empty asm: behind inline asm the hazard recogniser pads with s_nop
            __builtin_amdgcn_sched_barrier(0);
        });
        // the top stage holds the finished y[tau - (S - 1)]
        const int ty = tau0 + u - (S - 1);
#if !(BFIR_SYS_EXP & 4)
        // the lanes below the top stage (and bin 0) carry a byte offset beyond the end of every product-spectra buffer
        // (kvs): the buffer range check drops their stores -- cheaper than an exec-mask round trip per slot
        // ty < t_lo only in the first S - 1 slots of a run's first group (the chain is still filling); ty < t_hi by the slot
        // count of the run.  A compile-time test for all other slots: every scalar instruction of a slot costs issue time here
        if (u >= S - 1 || tau0 != t_lo) {
            Cx<T> v; v.x = ar; v.y = ai;
            sys_store<T, ILV, (BFIR_NT_Y & 1) ? 2 : 0>(ry, kvs, (unsigned)BFIR_YSLOT(a, ty) * xstep, v);
        }
#else
        if (ty == -12345 && store_lane) { Cx<T> v; v.x = ar; v.y = ai; sys_store<T, ILV, 0>(ry, kv, 0, v); }
#endif
#if !(BFIR_SYS_EXP & 1)
        hand_r = hand_t<S>(ar); hand_i = hand_t<S>(ai);
#else
        hand_r = ar; hand_i = ai;
#endif
    });
}

// registers: 2 PL + 2 (PL + D) values (x 2 for double) + a dozen; the waves per SIMD the build is asked for
template <typename T, int PL, int D> constexpr int sys_min_waves()
{
    const int regs = (int)(sizeof(T) / 4) * (2 * PL + 2 * (PL + D)) + 16;
    return regs <= 64 ? 8 : regs <= 80 ? 6 : regs <= 96 ? 5 : regs <= 128 ? 4 : regs <= 168 ? 3 : 2;
}

template <typename T, bool ILV, int S, int PL, int D>
__global__ __launch_bounds__(256, (sys_min_waves<T, PL, D>())) void k_mac_sys(MacArgs a, int ncol, int nR, int R, int n_dc)
{
    static_assert(ILV || sizeof(T) == 8, "fp32 engines of this size keep (re, im) pairs");
    constexpr int NYQ = ILV ? 1 : 4;                             // where bin 0 keeps its second real (Nyquist): pairs / grouped
    constexpr int BPW = 256 / S;                                 // bins per workgroup
    const int N = a.N, ring = a.ring;
    if ((int)blockIdx.x < n_dc) {
        // DC / Nyquist: two independent real sums in partition order; one workgroup = one channel x 256 output
        // blocks: the delay-line values and the filter values go through LDS, every thread runs its own chains
        constexpr int PB = 32;
        __shared__ T s_x[2][256 + PB], s_hh[2][PB];
        const int ndt = (a.n_t + 255) / 256;
        const int gc = blockIdx.x / ndt, t0 = (blockIdx.x - gc * ndt) * 256;
        const T *__restrict__ X = (const T *)a.x + (long)gc * a.x_ch_stride;
        const T *__restrict__ H = (const T *)a.h + (long)gc * a.h_ch_stride;
        const int nb = a.nblk[gc];
        const int t = t0 + threadIdx.x;
        T dc = 0, ny = 0;
        for (int pb0 = 0; pb0 < nb; pb0 += PB) {                     // all partitions, PB at a time, in order
            if (pb0 > 0) __syncthreads();
            for (int i = threadIdx.x; i < 256 + PB - 1; i += 256) {  // s_x[.][i] = block t0 - pb0 - (PB-1) + i
                int sl = (a.base_slot + t0 - pb0 - (PB - 1) + i) % ring; if (sl < 0) sl += ring;
                s_x[0][i] = X[(long)sl * N]; s_x[1][i] = X[(long)sl * N + NYQ];
            }
            if (threadIdx.x < PB) {
                const int p = pb0 + threadIdx.x;
                s_hh[0][threadIdx.x] = p < nb ? H[(long)p * N] : (T)0;
                s_hh[1][threadIdx.x] = p < nb ? H[(long)p * N + NYQ] : (T)0;
            }
            __syncthreads();
            const int np = min(PB, nb - pb0);
            for (int p = 0; p < np; p++) {
                dc = fma_t(s_x[0][threadIdx.x + PB - 1 - p], s_hh[0][p], dc);
                ny = fma_t(s_x[1][threadIdx.x + PB - 1 - p], s_hh[1][p], ny);
            }
        }
        if (t < a.n_t) {
            T *yo = (T *)a.y + (long)gc * a.y_ch_stride + (long)BFIR_YSLOT(a, t) * N;
            yo[0] = dc; yo[NYQ] = ny;
        }
        return;
    }
    // XCD-aware bijective remap: (channel, bin column) major, run minor -- the runs of one column meet in one L2
    const int Wg = gridDim.x - n_dc, b = blockIdx.x - n_dc, xcd = b & 7, qn = Wg >> 3, rn = Wg & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / nR, r = w - s * nR;
    const int gc = s / ncol, col = s - gc * ncol;
    const int stage = SysLanes<S>::stage(threadIdx.x);
    const unsigned k = col * BPW + SysLanes<S>::bin(threadIdx.x);   // bin
    // byte offset of the bin's real part inside a spectrum (its imaginary part: + sizeof(T) in pairs, + 32 in groups of four)
    const unsigned kv = ILV ? k * 2u * (unsigned)sizeof(T) : ((k >> 2) * 8u + (k & 3u)) * 8u;
    const T *__restrict__ Hc = (const T *)a.h + (long)gc * a.h_ch_stride;
    const int nb = a.nblk[gc];
    const int t_lo = r * R, t_hi = min(a.n_t, t_lo + R);       // this run's outputs
    if (t_lo >= t_hi) return;
    // bytes per spectrum / per ring
    const unsigned xstep = (unsigned)N * (unsigned)sizeof(T), xwrap = (unsigned)ring * xstep;
    const __amdgpu_buffer_rsrc_t rx = sys_rsrc((const T *)a.x + (long)gc * a.x_ch_stride, xwrap);
#ifdef BFIR_EXPERIMENT_ALIAS
    const __amdgpu_buffer_rsrc_t ry = sys_rsrc((T *)a.y + (long)gc * a.y_ch_stride, (unsigned)min(a.n_t, a.y_alias) * xstep);
#else
    const __amdgpu_buffer_rsrc_t ry = sys_rsrc((T *)a.y + (long)gc * a.y_ch_stride, (unsigned)a.n_t * xstep);
#endif

    T hr[PL], hi[PL];
#pragma unroll
    for (int p = 0; p < PL; p++) {
        const int pg = stage * PL + p;
        if (pg < nb) {
            const T *hp = (const T *)((const char *)(Hc + (long)pg * N) + kv);
            hr[p] = hp[0]; hi[p] = hp[NYQ];
        } else { hr[p] = 0; hi[p] = 0; }
    }
    // Before the first slot every lane fetches its own history: in slot tau stage j has x[tau - j (PL + 1)] as its newest
    // value; entry G - p holds the value p slots old, entries 0 .. D-1 are in flight.  (From then on stage j + 1 is fed
    // by stage j, see sys_group; entry D = G - PL is the first value to move up.)
    const int s0 = t_lo - stage * (PL + 1);
    auto slot_of = [&](int t) { int v = (a.base_slot + t) % ring; if (v < 0) v += ring; return (unsigned)v; };
    constexpr int G = PL + D;
    Cx<T> W[G];
#pragma unroll
    for (int p = 1; p <= PL; p++) W[G - p] = sys_load<T, ILV, 0>(rx, slot_of(s0 - p) * xstep + kv, 0);
#pragma unroll
    for (int d = 0; d < D; d++) W[d] = sys_load<T, ILV, 0>(rx, slot_of(s0 + d) * xstep + kv, 0);
    unsigned so = __builtin_amdgcn_readfirstlane(slot_of(t_lo + D) * xstep);   // stage 0's time from here on: wave-uniform
    T hand_r = 0, hand_i = 0;
    const bool store_lane = stage == S - 1 && k != 0;          // bin 0 belongs to the DC / Nyquist workgroups
    const unsigned kvs = store_lane ? kv : 0x80000000u;        // out of range: mac_sys_supported keeps the buffers below 2 GiB
    int left = t_hi - t_lo + (S - 1);                           // slots: the last output leaves the top stage S - 1 slots later
    for (int tau = t_lo; left > 0; tau += G, left -= G)
        sys_group<T, ILV, S, PL, D>(W, hr, hi, hand_r, hand_i, rx, ry, so, xstep, xwrap, kv, kvs, tau, t_lo, store_lane, left, a);
}

}  // namespace

// runs per (channel, bin column): enough workgroups for the waves per SIMD the kernel is built for, in one round
// (BFIR_MAC_RANGE overrides the run length, in blocks; BFIR_SYS_WGS the workgroups in flight)
template <typename T, bool ILV, int S, int PL, int D> static void launch_sys(const MacArgs &a_, hipStream_t s)
{
    MacArgs a = a_;
#ifdef BFIR_EXPERIMENT_ALIAS
    if (const int xa = bfir_alias_env("BFIR_X_ALIAS")) { a.ring = xa; a.base_slot %= xa; }
    if (const int ya = bfir_alias_env("BFIR_Y_ALIAS")) a.y_alias = ya;
#endif
    const int ncol = a.N / 2 / (256 / S);
    const char *re_ = getenv("BFIR_MAC_RANGE");
    int R = re_ ? atoi(re_) : 0;
    if (R <= 0) {
        const long cols = (long)ncol * a.n_ch;
        const char *we = getenv("BFIR_SYS_WGS");
        const long wgs = we && atoi(we) > 0 ? atoi(we) : 256L * sys_min_waves<T, PL, D>();
        const long want = std::max<long>(1, wgs / cols);
        R = std::max(2 * PL, (int)((a.n_t + want - 1) / want));
    }
    const int nR = (a.n_t + R - 1) / R;
    const int n_dc = a.n_ch * ((a.n_t + 255) / 256);
    hipLaunchKernelGGL((k_mac_sys<T, ILV, S, PL, D>), dim3(n_dc + nR * ncol * a.n_ch), dim3(256), 0, s, a, ncol, nR, R, n_dc);
}

// fp32 on the pairs layout (whole columns of 256 / S bins: N / 2 >= 128), fp64 on the grouped layout (N / 2 >= 64 bins,
// i.e. the engines with N >= 512 either way); S PL >= B: up to BFIR_MAC_SYS_MAX_B = 256 partitions
bool mac_sys_supported(const MacArgs &a)
{
    if (a.N < 512) return false;
    // 32-bit byte offsets into one channel's delay line, and 2 GiB as the "nowhere" offset of the lanes that do not store
    const unsigned long long spec = (unsigned long long)a.N * (unsigned)a.realsize;
    if (spec * (unsigned)a.ring >= (1ull << 32) || spec * (unsigned)a.n_t > (1ull << 31)) return false;
    if (a.realsize == 4) return a.interleaved && a.B <= BFIR_MAC_SYS_MAX_B;
    return a.B <= BFIR_MAC_SYS_MAX_B;                        // fp64: either layout
}

// Stages x partitions per stage: the smallest S PL >= B the build holds.  Partitions beyond B are zeros the lanes multiply
// all the same (the stages of a bin are lanes of one wave), so the steps are kept at a quarter: PL = 12 between the powers of
// two (the plug-in cuts whatever impulse file it is given into 1024-sample partitions, foo_dsp_bfir.cpp:275-276: any count
// occurs).  fp64: prefetch depth 6 with 16 partitions per stage (160 registers, three waves per SIMD; measured best, cfg5
// 39.7 against 39.2 with 4 and 39.4 with 8, the plug-in's shape 43.3 / 40.1 / 43.7, profiles/r03_fp64.txt; BFIR_SYS_D
// overrides at 33 ... 64 partitions), 4 with 12 (128 registers: four waves).
template <bool ILV> static void launch_sys_f64(const MacArgs &a, hipStream_t s)
{
    if (a.B <= 16) launch_sys<double, ILV, 2, 8, 4>(a, s);
    else if (a.B <= 24) launch_sys<double, ILV, 2, 12, 4>(a, s);
    else if (a.B <= 32) launch_sys<double, ILV, 2, 16, 4>(a, s);
    else if (a.B <= 48) launch_sys<double, ILV, 4, 12, 4>(a, s);
    else if (a.B <= 64) {
        const char *de = getenv("BFIR_SYS_D");
        const int d = de ? atoi(de) : 6;
        if (d >= 8) launch_sys<double, ILV, 4, 16, 8>(a, s);
        else if (d == 6) launch_sys<double, ILV, 4, 16, 6>(a, s);
        else launch_sys<double, ILV, 4, 16, 4>(a, s);
    }
    else if (a.B <= 96) launch_sys<double, ILV, 8, 12, 4>(a, s);
    else if (a.B <= 128) launch_sys<double, ILV, 8, 16, 6>(a, s);    // eight stages: 65 ... 128 partitions
    else if (a.B <= 192) launch_sys<double, ILV, 16, 12, 4>(a, s);   // sixteen stages, one DPP row per bin: 129 ... 256 partitions
    else launch_sys<double, ILV, 16, 16, 6>(a, s);                   // (1024-sample blocks, an impulse of up to 5.9 s at 44.1 kHz)
}

void launch_mac_sys(const MacArgs &a, hipStream_t s)
{
    if (a.realsize == 4) {
        if (a.B <= 8) launch_sys<float, true, 2, 4, 4>(a, s);
        else if (a.B <= 16) launch_sys<float, true, 2, 8, 4>(a, s);
        else if (a.B <= 24) launch_sys<float, true, 2, 12, 4>(a, s);
        else if (a.B <= 32) launch_sys<float, true, 2, 16, 4>(a, s);
        else if (a.B <= 48) launch_sys<float, true, 4, 12, 4>(a, s);
        else if (a.B <= 64) launch_sys<float, true, 4, 16, 4>(a, s);
        else if (a.B <= 128) launch_sys<float, true, 8, 16, 4>(a, s);
        else launch_sys<float, true, 16, 16, 4>(a, s);
    } else if (a.interleaved) launch_sys_f64<true>(a, s);
    else launch_sys_f64<false>(a, s);
}

}  // namespace bfir
