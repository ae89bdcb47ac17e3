// mac_sys.hip -- k_mac_sys: the partition sums as a two-stage systolic chain that walks FORWARD in time.
//
//     y[t] = sum_{p < B} x[t - p] h[p]      per bin, partitions in the reference's order p = 0 .. B-1
//                                           (brutefir/brutefir.cpp:288-299; convolve / convolve_add,
//                                           fftw_convolver.cpp:1464-1525), one fused multiply-add chain per output.
//
// k_mac_stream (kernels.hip) keeps all 32 partitions of a bin AND 32 running sums in one lane (168 registers, three
// waves per SIMD) and therefore has to walk the delay line backwards, a whole range of blocks at a time: it needs the
// NEWEST block of its range first, and it cannot share a SIMD with the FFT kernels' workgroups (4 x 88..96 registers
// per SIMD leave 128..160).  Here TWO lanes of a wave share a bin:
//
//     lanes  0..31 ("A")  hold h[0 .. PL-1]     and the window x[t .. t-PL+1]
//     lanes 32..63 ("B")  hold h[PL .. 2PL-1]   and the window x[t-1-PL .. t-2PL]      (one slot behind A)
//
// In slot t lane A runs the first PL links of y[t]'s chain from zero, lane B the last PL links of y[t-1]'s chain
// starting from the partial sum A finished one slot earlier (handed over by ONE v_permlane32_swap per component:
// upper half <- lower half, lower half <- 0, which is exactly "B continues A's chain, A starts a new one").
// Every output is one chain p = 0 .. 2PL-1 with the same fma sequence as every other MAC kernel here: bit-identical
// sums.  Partitions >= nblk hold h = 0 (an exact no-op).
//
//   * 2 PL (h) + 2 PL (window) + queue registers: ~95 at PL = 16 instead of 168 -- five waves per SIMD, and two of
//     them fit beside two FFT workgroups;
//   * the walk is forward: y[t] leaves one slot after x[t] has arrived, a run of R outputs costs R + 1 slots (not
//     R + 31) and 2 PL - 1 window loads of history: short runs are cheap, which is what range-ordered producer /
//     consumer schedules need;
//   * 64 FMAs per lane and slot, both halves busy in every slot.  Only A's values come from memory (one 8-byte load
//     per slot, D slots ahead, wave-uniform slot address in an SGPR): the value A drops from its window in slot t
//     (x[t - PL]) is exactly B's newest value of slot t + 1, so it moves over with one more v_permlane32_swap per
//     component.  (Letting B re-read its values PL + 1 slots later doubled the kernel's traffic -- the L2 does not
//     hold a stream that long -- and made it memory-bound: 0.61 ms against 0.50 for k_mac_stream.)
// Bin 0 (DC | Nyquist: two independent real sums) is left to the first workgroups of the grid, as in k_mac_stream.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

#include "fft_lds.h"

// -DBFIR_SYS_EXP=bits (timing experiments only, results garbage): 1 no partial-sum hand-over, 2 no relay of x to
// the B lanes, 4 no stores, 8 no loads.  Never defined in the product library.
#ifndef BFIR_SYS_EXP
#define BFIR_SYS_EXP 0
#endif

namespace bfir {

namespace {

// Addressing through buffer descriptors (4 SGPRs per stream, wave-uniform) + ONE 32-bit byte offset per lane:
// a channel's delay line (ring x N floats) and product spectra stay below 4 GiB, and a 64-bit address pair per
// access would cost this kernel the registers that decide its occupancy.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sys_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long u = (unsigned long long)p;     // workgroup-uniform by construction
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float2 sys_ld_x(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, (BFIR_NT_X & 2) ? 2 : 0);
    float2 f; f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y);
    return f;
}
__device__ __forceinline__ void sys_st_y(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 f)
{
    u32x2 v; v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, (BFIR_NT_Y & 1) ? 2 : 0);
}

// upper 32 lanes <- v of the lower 32 lanes (same lane & 31), lower 32 lanes <- 0
__device__ __forceinline__ float hand_over(float v)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    // v_permlane32_swap: the odd row (lanes 32..63) of the first operand <-> the even row of the second
    const u2 r = __builtin_amdgcn_permlane32_swap(0u, __float_as_uint(v), false, false);
    return __uint_as_float(r.x);
}
// upper 32 lanes of dst <- lower 32 lanes of src; src's lower half is dead afterwards (it receives dst's upper half)
__device__ __forceinline__ void relay(float &dst, float &src)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(dst), __float_as_uint(src), false, false);
    dst = __uint_as_float(r.x); src = __uint_as_float(r.y);
}

// One group of G = PL + D slots.  The window and the prefetch queue are ONE rotating set of G register pairs: in
// slot u entry u holds the newest value, entry u - p the one p slots old, and the load issued in slot u goes into
// entry u + D, whose old content in the A lanes (x[tau - PL], last used in slot u - 1) is first moved into the B
// lanes of entry u + 1: B's newest value of the next slot.  Both halves load the same address (B's copy is
// overwritten by that move before it is looked at), so the slot part of the address is a scalar.  No register
// copies; the entries rotate with the slot index, hence the unrolled group; only the first `rem` slots run (the
// last group of a run).  ONE instantiation on purpose: a second copy of the group (without the tail test) makes
// the register allocator reconcile the 2 G window registers at the joins and costs 16+ registers (measured: 80 ->
// 96 + spills at PL = 16); the tail test is a scalar compare-and-branch per slot.
// tau0 = output index lane A works on in slot u = 0.
template <int PL, int D>
__device__ __forceinline__ void sys_group(float2 (&W)[PL + D], const float (&hr)[PL], const float (&hi)[PL],
                                          float &hand_r, float &hand_i, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ry,
                                          unsigned &so, unsigned xstep, unsigned xwrap, unsigned k8, int tau0,
                                          int t_lo, int t_hi, bool store_lane, int rem, const MacArgs &a)
{
    constexpr int G = PL + D;
    static_assert(D >= 2, "entry u + 1 and entry u + D are different registers");
    static_for<0, G>([&](auto U) {
        constexpr int u = decltype(U)::value;
        if (u >= rem) return;                                    // wave-uniform
#if !(BFIR_SYS_EXP & 2)
        relay(W[(u + 1) % G].x, W[(u + D) % G].x);               // A's x[tau - PL] -> B's newest of slot tau + 1
        relay(W[(u + 1) % G].y, W[(u + D) % G].y);
#endif
#if !(BFIR_SYS_EXP & 8)
        W[(u + D) % G] = sys_ld_x(rx, k8, so);                   // x[tau + D], D slots ahead
#else
        W[(u + D) % G].x += 1.f;
#endif
        so += xstep; so = so >= xwrap ? so - xwrap : so;         // scalar
        float ar = hand_r, ai = hand_i;                          // A: 0; B: A's partial sum of y[tau - 1]
        static_for<0, PL>([&](auto P) {
            constexpr int p = decltype(P)::value;
            const float2 w = W[(u - p + G) % G];                 // p slots old
            ar = fmaf(w.x, hr[p], ar); ai = fmaf(w.x, hi[p], ai);
            ar = fmaf(-w.y, hi[p], ar); ai = fmaf(w.y, hr[p], ai);
            // keep the two chains interleaved (an independent instruction between dependent ones); a scheduling
            // barrier, not an empty asm: behind inline asm the hazard recogniser pads with s_nop
            __builtin_amdgcn_sched_barrier(0);
        });
        // B lanes hold the finished y[tau - 1]
        const int ty = tau0 + u - 1;
#if !(BFIR_SYS_EXP & 4)
        if (ty >= t_lo && ty < t_hi && store_lane) {
            float2 v; v.x = ar; v.y = ai;
            sys_st_y(ry, k8, (unsigned)BFIR_YSLOT(a, ty) * xstep, v);
        }
#else
        if (ty == -12345 && store_lane) { float2 v; v.x = ar; v.y = ai; sys_st_y(ry, k8, 0, v); }
#endif
#if !(BFIR_SYS_EXP & 1)
        hand_r = hand_over(ar); hand_i = hand_over(ai);
#else
        hand_r = ar; hand_i = ai;
#endif
    });
}

template <int PL, int D>
__global__ __launch_bounds__(256, 5) void k_mac_sys(MacArgs a, int ncol, int nR, int R, int n_dc)
{
    const int N = a.N, ring = a.ring;
    if ((int)blockIdx.x < n_dc) {
        // DC / Nyquist: the two floats of bin 0 (pairs layout), two independent real sums in partition order;
        // one workgroup = one channel x 256 output blocks (the k_mac_stream code)
        constexpr int PB = 2 * PL;
        __shared__ float s_x[2][256 + PB], s_hh[2][PB];
        const int ndt = (a.n_t + 255) / 256;
        const int gc = blockIdx.x / ndt, t0 = (blockIdx.x - gc * ndt) * 256;
        const float *__restrict__ X = (const float *)a.x + (long)gc * a.x_ch_stride;
        const float *__restrict__ H = (const float *)a.h + (long)gc * a.h_ch_stride;
        const int nb = a.nblk[gc];
        const int t = t0 + threadIdx.x;
        float dc = 0.f, ny = 0.f;
        for (int pb0 = 0; pb0 < nb; pb0 += PB) {
            if (pb0 > 0) __syncthreads();
            for (int i = threadIdx.x; i < 256 + PB - 1; i += 256) {
                int sl = (a.base_slot + t0 - pb0 - (PB - 1) + i) % ring; if (sl < 0) sl += ring;
                s_x[0][i] = X[(long)sl * N]; s_x[1][i] = X[(long)sl * N + 1];
            }
            if (threadIdx.x < PB) {
                const int p = pb0 + threadIdx.x;
                s_hh[0][threadIdx.x] = p < nb ? H[(long)p * N] : 0.f;
                s_hh[1][threadIdx.x] = p < nb ? H[(long)p * N + 1] : 0.f;
            }
            __syncthreads();
            const int np = min(PB, nb - pb0);
            for (int p = 0; p < np; p++) {
                dc = fmaf(s_x[0][threadIdx.x + PB - 1 - p], s_hh[0][p], dc);
                ny = fmaf(s_x[1][threadIdx.x + PB - 1 - p], s_hh[1][p], ny);
            }
        }
        if (t < a.n_t) {
            float *yo = (float *)a.y + (long)gc * a.y_ch_stride + (long)BFIR_YSLOT(a, t) * N;
            yo[0] = dc; yo[1] = ny;
        }
        return;
    }
    // XCD-aware bijective remap: (channel, bin column) major, run minor -- the runs of one column meet in one L2
    const int Wg = gridDim.x - n_dc, b = blockIdx.x - n_dc, xcd = b & 7, qn = Wg >> 3, rn = Wg & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / nR, r = w - s * nR;
    const int gc = s / ncol, col = s - gc * ncol;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const unsigned k = col * 128 + (threadIdx.x >> 6) * 32 + (lane & 31);   // bin: (re, im) pair k of every spectrum
    const int N2 = N / 2;
    const float2 *__restrict__ Hc = (const float2 *)((const float *)a.h + (long)gc * a.h_ch_stride);
    const int nb = a.nblk[gc];
    const int t_lo = r * R, t_hi = min(a.n_t, t_lo + R);       // this run's outputs
    if (t_lo >= t_hi) return;
    // bytes per spectrum / per ring; a lane's offset slot * xstep + k8 is >= xwrap exactly when slot >= ring (k8 < xstep)
    const unsigned xstep = (unsigned)N * 4u, xwrap = (unsigned)ring * xstep, k8 = k * 8u;
    const __amdgpu_buffer_rsrc_t rx = sys_rsrc((const float *)a.x + (long)gc * a.x_ch_stride, xwrap);
#ifdef BFIR_EXPERIMENT_ALIAS
    const __amdgpu_buffer_rsrc_t ry = sys_rsrc((float *)a.y + (long)gc * a.y_ch_stride, (unsigned)min(a.n_t, a.y_alias) * xstep);
#else
    const __amdgpu_buffer_rsrc_t ry = sys_rsrc((float *)a.y + (long)gc * a.y_ch_stride, (unsigned)a.n_t * xstep);
#endif

    float hr[PL], hi[PL];
#pragma unroll
    for (int p = 0; p < PL; p++) {
        const int pg = half * PL + p;
        if (pg < nb) { const float2 h = (Hc + (long)pg * N2)[k]; hr[p] = h.x; hi[p] = h.y; }
        else { hr[p] = 0.f; hi[p] = 0.f; }
    }
    // Before the first slot every lane fetches its own history: in slot tau lane A has x[tau] as its newest value,
    // lane B x[tau - 1 - PL]; entry G - p holds the value p slots old, entries 0 .. D-1 are in flight.  (From then
    // on B is fed by A, see sys_group; A's entry D = G - PL is the first value to move over.)
    const int s0 = t_lo - half * (PL + 1);
    auto slot_of = [&](int t) { int v = (a.base_slot + t) % ring; if (v < 0) v += ring; return (unsigned)v; };
    constexpr int G = PL + D;
    float2 W[G];
#pragma unroll
    for (int p = 1; p <= PL; p++) W[G - p] = sys_ld_x(rx, slot_of(s0 - p) * xstep + k8, 0);
#pragma unroll
    for (int d = 0; d < D; d++) W[d] = sys_ld_x(rx, slot_of(s0 + d) * xstep + k8, 0);
    unsigned so = __builtin_amdgcn_readfirstlane(slot_of(t_lo + D) * xstep);   // A's time from here on: wave-uniform
    float hand_r = 0.f, hand_i = 0.f;
    const bool store_lane = half == 1 && k != 0;               // bin 0 belongs to the DC / Nyquist workgroups
    int left = t_hi - t_lo + 1;                                 // slots: the last output leaves lane B one slot later
    for (int tau = t_lo; left > 0; tau += G, left -= G)
        sys_group<PL, D>(W, hr, hi, hand_r, hand_i, rx, ry, so, xstep, xwrap, k8, tau, t_lo, t_hi, store_lane, left, a);
}

}  // namespace

// runs per (channel, 128-bin column): about five waves per SIMD in one round (BFIR_MAC_RANGE overrides, in blocks)
template <int PL, int D> static void launch_sys(const MacArgs &a_, hipStream_t s)
{
    MacArgs a = a_;
#ifdef BFIR_EXPERIMENT_ALIAS
    if (const int xa = bfir_alias_env("BFIR_X_ALIAS")) { a.ring = xa; a.base_slot %= xa; }
    if (const int ya = bfir_alias_env("BFIR_Y_ALIAS")) a.y_alias = ya;
#endif
    const int ncol = a.N / 2 / 128;
    const char *re_ = getenv("BFIR_MAC_RANGE");
    int R = re_ ? atoi(re_) : 0;
    if (R <= 0) {
        const long cols = (long)ncol * a.n_ch;
        const char *we = getenv("BFIR_SYS_WGS");                 // tuning aid: workgroups in flight the runs are cut for
        const long want = std::max<long>(1, (we && atoi(we) > 0 ? atoi(we) : 1280) / cols);   // 256 CUs x 5 workgroups
        R = std::max(2 * PL, (int)((a.n_t + want - 1) / want));
    }
    const int nR = (a.n_t + R - 1) / R;
    const int n_dc = a.n_ch * ((a.n_t + 255) / 256);
    hipLaunchKernelGGL((k_mac_sys<PL, D>), dim3(n_dc + nR * ncol * a.n_ch), dim3(256), 0, s, a, ncol, nR, R, n_dc);
}

bool mac_sys_supported(const MacArgs &a)
{
    return a.realsize == 4 && a.interleaved && a.B <= 32 && a.N >= 512;   // whole 128-bin columns
}

void launch_mac_sys(const MacArgs &a, hipStream_t s)
{
    if (a.B <= 8) launch_sys<4, 4>(a, s);
    else if (a.B <= 16) launch_sys<8, 8>(a, s);
    else {
        // prefetch depth (slots): what has to be in flight is bytes, and a load instruction of this kernel brings
        // 256 of them (both lane halves read the same address); BFIR_SYS_D = 4 / 8 / 12 (tuning aid)
        const char *de = getenv("BFIR_SYS_D");
        const int d = de ? atoi(de) : 12;
        if (d == 4) launch_sys<16, 4>(a, s);
        else if (d == 8) launch_sys<16, 8>(a, s);
        else launch_sys<16, 12>(a, s);
    }
}

}  // namespace bfir
