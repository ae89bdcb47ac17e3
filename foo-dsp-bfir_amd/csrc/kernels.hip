// kernels.hip -- hand-written gfx950 kernels of the partitioned-FIR hot path.
//
// One kernel per stage of brutefir::run (brutefir/brutefir.cpp:244-343), each
// launched once over a whole chunk of blocks x channels:
//   k_stage_in   a5   convolver_raw2cbuf            fftw_convolver.cpp:156-185
//   k_fwd        a6+a7 time2freq + mixnscale INPUT  :187-212, :883-907 / :1583-1607
//                a16  coeffs2cbuf (zero_first_half) :474-537
//   k_fwd_run    a5-a7 of fp64 engines on raw frames, a run of blocks per workgroup
//   k_mac        a8-a10 convolve / convolve_add / convolve_inplace
//                                                   :1429-1525 / :2125-2220
//   k_inv        a11+a12 mixnscale OUTPUT + freq2time :1163-1186, :350-375
//   k_inv_run    a11-a13 of fp64 engines on raw frames, a run of blocks per workgroup
//   k_stage_out  a13  convolver_cbuf2raw            :405-466, real2raw.cpp:321-420
// Stage-level buffers use the reference's grouped layout (4 real parts, then the
// 4 imaginary parts of the same bins; Nyquist in slot 4): one group is two
// 16-byte vectors, bit-compatible with a reference cbuf.  An ENGINE's internal
// spectra are (re, im) pairs instead where its kernels read them (fp32 from 512
// points; fp64 on the run kernels): FwdArgs / MacArgs / InvArgs .interleaved.
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_lds.h"

namespace bfir {

constexpr int BFIR_MAXCH = 8;   // BF_MAXCHANNELS, brutefir/global.h:21

#define BFIR_FOR_LOG2M(F) \
    F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14)

// ---------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------
template <int LOG2M> static void fill_tw(std::vector<long double> &tw)
{
    using F = LdsFft<float, LOG2M, -1>;
    tw.assign(2 * (size_t)(F::twsize() > 0 ? F::twsize() : 1), 0.0L);
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int s = 1; s < F::NP; s++) {
        const int R = F::radix(s), p = F::pprod(s);
        for (int r = 1; r < R; r++)
            for (int k = 0; k < p; k++) {
                long double ang = -two_pi * (long double)(r * k) / (long double)(p * R);
                size_t o = (size_t)F::twoff(s) + (size_t)(r - 1) * p + k;
                tw[2 * o] = cosl(ang);
                tw[2 * o + 1] = sinl(ang);
            }
    }
}

// twiddle bases of the persistent kernels (fft_lds.h): exp(-2 pi i e_j k0 / (p R)), k0 < min(p, NT)
template <int LOG2M> static void fill_twb(std::vector<long double> &tw)
{
    using F = LdsFft<float, LOG2M, -1>;
    tw.assign(2 * (size_t)(F::bsize() > 0 ? F::bsize() : 1), 0.0L);
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int s = 1; s < F::NP; s++) {
        const int R = F::radix(s), p = F::pprod(s), ks = F::kspan(s);
        for (int j = 0; j < tw_nbase(R); j++)
            for (int k0 = 0; k0 < ks; k0++) {
                long double ang = -two_pi * (long double)(tw_base_exp(R, j) * k0) / (long double)(p * R);
                size_t o = (size_t)F::boff(s) + (size_t)j * ks + k0;
                tw[2 * o] = cosl(ang);
                tw[2 * o + 1] = sinl(ang);
            }
    }
}

template <typename T> static void *upload(const std::vector<long double> &v)
{
    std::vector<T> h(v.size());
    for (size_t i = 0; i < v.size(); i++) h[i] = (T)v[i];
    void *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    return d;
}

int fft_threads(int log2m)
{
    switch (log2m) {
#define F(lg) case lg: return FftCfg<lg>::NT;
        BFIR_FOR_LOG2M(F)
#undef F
    }
    return 0;
}

int fft_plan_create(FftPlan *plan, int filter_length, int realsize)
{
    int lg = 0;
    while ((1 << lg) < filter_length) lg++;
    if ((1 << lg) != filter_length || lg < BFIR_MIN_LOG2M || lg > BFIR_MAX_LOG2M) return -1;
    if (realsize != 4 && realsize != 8) return -1;
    // one transform's LDS buffer must fit: M complex values
    if (((size_t)filter_length + (size_t)filter_length / 32) * 2 * (size_t)realsize > 160 * 1024) return -1;
    std::vector<long double> tw, ws, twb;
    switch (lg) {
#define F(lgv) case lgv: fill_tw<lgv>(tw); fill_twb<lgv>(twb); break;
        BFIR_FOR_LOG2M(F)
#undef F
    }
    const int M = filter_length, N = 2 * M;
    ws.resize(2 * (size_t)M);
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int k = 0; k < M; k++) {
        long double ang = -two_pi * (long double)k / (long double)N;
        ws[2 * k] = cosl(ang);
        ws[2 * k + 1] = sinl(ang);
    }
    plan->log2m = lg;
    plan->realsize = realsize;
    plan->tw = realsize == 4 ? upload<float>(tw) : upload<double>(tw);
    plan->ws = realsize == 4 ? upload<float>(ws) : upload<double>(ws);
    plan->twb = realsize == 4 ? upload<float>(twb) : upload<double>(twb);
    if (!plan->tw || !plan->ws || !plan->twb) { fft_plan_destroy(plan); return -2; }
    return 0;
}

void fft_plan_destroy(FftPlan *plan)
{
    if (plan->tw) (void)hipFree(plan->tw);
    if (plan->ws) (void)hipFree(plan->ws);
    if (plan->twb) (void)hipFree(plan->twb);
    plan->tw = plan->ws = plan->twb = nullptr;
}

// |v| as an integer whose ordering equals the ordering of the magnitudes
__device__ __forceinline__ unsigned int abs_bits(float v) { return __float_as_uint(fabsf(v)); }
__device__ __forceinline__ unsigned long long abs_bits(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }

// Two channels per workgroup (k_fwd / k_inv, CPW = 2): both LDS buffers must fit a CU and the workgroup 1024 threads.
template <typename T, int LOG2M> constexpr bool direct_stereo_fits()
{
    return FftCfg<LOG2M>::NT >= 64 && 2 * FftCfg<LOG2M>::NT <= 1024 &&
           2 * sizeof(T) * 2 * ((size_t(1) << LOG2M) + (size_t(1) << LOG2M) / 32) <= 128 * 1024;
}
bool direct_stereo_supported(int filter_length, int realsize)
{
    int lg = 0;
    while ((1 << lg) < filter_length) lg++;
    bool ok = false;
    switch (lg) {
#define F(lg_) case lg_: ok = realsize == 4 ? direct_stereo_fits<float, lg_>() : direct_stereo_fits<double, lg_>(); break;
        BFIR_FOR_LOG2M(F)
#undef F
    }
    return ok;
}
// float (or, for fp64 engines, double) frames of an engine with an even channel count: a workgroup takes the two channels
// of a PAIR.  Stereo frames are moved two at a time (four samples per lane: blocks on four-sample boundaries), wider frames
// two samples (one pair of one frame) per lane.
static bool direct_stereo_ok(int raw_bytes, int realsize, int C, const void *raw, long eng_stride_samples, long frame_off)
{
    if ((raw_bytes != 4 && !(raw_bytes == 8 && realsize == 8)) || C < 2 || (C & 1)) return false;
    if (C == 2) return ((uintptr_t)raw % (4 * raw_bytes)) == 0 && (eng_stride_samples % 4) == 0 && (frame_off % 2) == 0;
    return ((uintptr_t)raw % (2 * raw_bytes)) == 0 && (eng_stride_samples % 2) == 0;
}

// ---------------------------------------------------------------------------
// a6 + a7: forward real FFT into the grouped layout
// ---------------------------------------------------------------------------
// TR: type of the raw samples in direct mode (FwdArgs.raw_bytes), void for the planar source.
// CPW = 2 (direct mode, float frames -- or double frames of an fp64 engine -- with an even channel count): one workgroup of
// 2 NT threads transforms BOTH channels of a pair, each half in its own LDS buffer, so that the interleaved frames are
// moved four samples per lane (two whole stereo frames) instead of one sample at a stride.
template <typename T, int LOG2M, bool ILV, typename TR = void, int CPW = 1>
__global__ __launch_bounds__(FftCfg<LOG2M>::NT * CPW) void k_fwd(FwdArgs a,
                                                                 const typename Vec2<T>::type *__restrict__ tw,
                                                                 const typename Vec2<T>::type *__restrict__ ws)
{
    using F = LdsFft<T, LOG2M, -1>;
    using V2 = typename Vec2<T>::type;
    using V4 = typename Vec4<T>::type;
    constexpr int M = F::M, NT = F::NT, P = F::P, N = 2 * M;
    __shared__ __attribute__((aligned(16))) V2 lds_all[CPW][F::LDS_ELEMS];

    constexpr bool DIRECT = !std::is_void<TR>::value;
    static_assert(CPW == 1 || (CPW == 2 && DIRECT && (std::is_same<TR, float>::value || std::is_same<TR, T>::value)), "two channels per workgroup: float frames, or frames of the engine's own type");
    const int half = CPW == 1 ? 0 : (int)threadIdx.x / NT;         // which channel of the frame (wave-uniform: NT >= 64)
    const int tid = CPW == 1 ? (int)threadIdx.x : (int)threadIdx.x - half * NT;
    V2 *lds = lds_all[half];
    // direct mode: the channels of a block share input cache lines -> give every XCD a contiguous range
    int wi = blockIdx.x;
    if constexpr (DIRECT) {
        const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
        wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    }
    const int ncw = a.n_ch / CPW;
    const int t = wi / ncw, gc = (wi - t * ncw) * CPW + half;
    const T *__restrict__ cur = (const T *)a.src + (long)gc * a.src_ch_stride + (long)t * M;
    const T *__restrict__ old = (t == 0) ? (const T *)a.prev + (long)gc * a.prev_ch_stride : cur - M;
    T *__restrict__ dst =
        (T *)a.dst + (long)gc * a.dst_ch_stride + (long)((a.base_slot + t) % a.ring) * N;

    BFIR_STAMP(0, 0);
    // z[m] = x[2m] + i x[2m+1] over the window [previous block | this block]
    T re[P], im[P];
    const T ls = (T)a.load_scale;
    if constexpr (DIRECT && CPW == 2) {
      using RS = typename std::conditional<DIRECT, TR, float>::type;
      using R2 = typename Vec2<RS>::type;
      using R4 = typename Vec4<RS>::type;
      if (a.C == 2) {
        // stereo frames: quad m of a block = frames 2m, 2m+1 = (l, r, l, r); aligned to four samples (launcher)
        const int g = gc >> 1;
        const long ho = (long)g * a.hist_eng_stride;
        const R4 *__restrict__ rc4 = (const R4 *)((const RS *)a.raw + (long)g * a.raw_eng_stride + (a.frame_off + (long)t * M) * 2);
        const R4 *__restrict__ ro4 = (t == 0) ? (const R4 *)((const RS *)a.prev_raw + ho) : rc4 - M / 2;
#pragma unroll
        for (int e = 0; e < P; e++) {
            const int m = F::in_index(tid, e);
            if (m < M / 2) {                                   // frames 2m, 2m+1 of the previous block
                const R4 v = ro4[m];
                re[e] = (T)(half ? v.y : v.x) * ls; im[e] = (T)(half ? v.w : v.z) * ls;
                if (a.n_t == 1 && half == 0)                   // one-block chunk: the other history block moves on unchanged
                    ((R4 *)((RS *)a.save_prev + ho))[m] = ((const R4 *)((const RS *)a.carry + ho))[m];
            } else {                                           // frames of this block
                const R4 v = rc4[m - M / 2];
                re[e] = (T)(half ? v.y : v.x) * ls; im[e] = (T)(half ? v.w : v.z) * ls;
                // the engine's history: raw frames of the last two blocks of the chunk (whole frames: one half stores)
                if (t >= a.n_t - 2 && half == 0)
                    ((R4 *)((RS *)(t == a.n_t - 1 ? a.save_last : a.save_prev) + ho))[m - M / 2] = v;
            }
        }
      } else {
        // wider frames, even channel count (round 3): the workgroup's two channels are the pair (c0, c0 + 1) of every
        // frame, two samples per frame at the frame stride -- the access pattern of the fp32 pair kernels
        const int C = a.C, g = gc / C, c0 = gc - g * C - half;
        const long ho = (long)g * a.hist_eng_stride + c0;
        const RS *__restrict__ rc = (const RS *)a.raw + (long)g * a.raw_eng_stride + (a.frame_off + (long)t * M) * C + c0;
        const RS *__restrict__ ro = (t == 0) ? (const RS *)a.prev_raw + ho : rc - (long)M * C;
#pragma unroll
        for (int e = 0; e < P; e++) {
            const int m = F::in_index(tid, e);
            if (m < M / 2) {                                   // frames 2m, 2m+1 of the previous block
                const R2 v0 = *(const R2 *)(ro + (long)(2 * m) * C), v1 = *(const R2 *)(ro + (long)(2 * m + 1) * C);
                re[e] = (T)(half ? v0.y : v0.x) * ls; im[e] = (T)(half ? v1.y : v1.x) * ls;
                if (a.n_t == 1 && half == 0) {                 // one-block chunk: the other history block moves on unchanged
                    RS *sp = (RS *)a.save_prev + ho; const RS *cr = (const RS *)a.carry + ho;
                    *(R2 *)(sp + (long)(2 * m) * C) = *(const R2 *)(cr + (long)(2 * m) * C);
                    *(R2 *)(sp + (long)(2 * m + 1) * C) = *(const R2 *)(cr + (long)(2 * m + 1) * C);
                }
            } else {                                           // frames of this block
                const int n = 2 * m - M;
                const R2 v0 = *(const R2 *)(rc + (long)n * C), v1 = *(const R2 *)(rc + (long)(n + 1) * C);
                re[e] = (T)(half ? v0.y : v0.x) * ls; im[e] = (T)(half ? v1.y : v1.x) * ls;
                if (t >= a.n_t - 2 && half == 0) {             // the engine's history (one half stores the pair)
                    RS *kp = (RS *)(t == a.n_t - 1 ? a.save_last : a.save_prev) + ho;
                    *(R2 *)(kp + (long)n * C) = v0; *(R2 *)(kp + (long)(n + 1) * C) = v1;
                }
            }
        }
      }
    } else if constexpr (DIRECT) {
        using RS = typename std::conditional<DIRECT, TR, float>::type;
        const int C = a.C, g = gc / C, c = gc - g * C;
        const long ho = (long)g * a.hist_eng_stride + c;
        const RS *__restrict__ rc = (const RS *)a.raw + (long)g * a.raw_eng_stride + (a.frame_off + (long)t * M) * C + c;
        const RS *__restrict__ ro = (t == 0) ? (const RS *)a.prev_raw + ho : rc - (long)M * C;
#pragma unroll
        for (int e = 0; e < P; e++) {
            const int m = F::in_index(tid, e);
            if (m < M / 2) {                                   // frames 2m, 2m+1 of the previous block
                const RS x0 = ro[(long)(2 * m) * C], x1 = ro[(long)(2 * m + 1) * C];
                re[e] = (T)x0 * ls; im[e] = (T)x1 * ls;
                if (a.n_t == 1) {                              // one-block chunk: the other history block moves on unchanged
                    RS *sp = (RS *)a.save_prev + ho; const RS *cr = (const RS *)a.carry + ho;
                    sp[(long)(2 * m) * C] = cr[(long)(2 * m) * C]; sp[(long)(2 * m + 1) * C] = cr[(long)(2 * m + 1) * C];
                }
            } else {                                           // frames of this block
                const int n = 2 * m - M;
                const RS x0 = rc[(long)n * C], x1 = rc[(long)(n + 1) * C];
                re[e] = (T)x0 * ls; im[e] = (T)x1 * ls;
                // the engine's history: raw frames of the last two blocks of the chunk
                if (t >= a.n_t - 2) {
                    RS *kp = (RS *)(t == a.n_t - 1 ? a.save_last : a.save_prev) + ho;
                    kp[(long)n * C] = x0; kp[(long)(n + 1) * C] = x1;
                }
            }
        }
    } else
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int m = F::in_index(tid, e);
        if (m < M / 2) {
            if (a.zero_first_half) {
                re[e] = (T)0; im[e] = (T)0;
            } else {
                V2 v = *(const V2 *)(old + 2 * m);
                re[e] = v.x * ls; im[e] = v.y * ls;
            }
        } else {
            V2 v = *(const V2 *)(cur + (2 * m - M));
            re[e] = v.x * ls; im[e] = v.y * ls;
        }
    }

    BFIR_STAMP(0, 1);
    F::run(re, im, lds, tw, tid);

    // Z in natural order to LDS so every thread can fetch Z[M-k]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < P; e++) {
        V2 v; v.x = re[e]; v.y = im[e];
        lds[F::phys(F::out_index(tid, e))] = v;
    }
    __syncthreads();
    // X_k = E_k + W^k O_k,  E = (Z_k + conj Z_{M-k})/2,  O = -i (Z_k - conj Z_{M-k})/2
    const T os = (T)a.out_scale;
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int k = F::out_index(tid, e);
        V2 pz = lds[F::phys((M - k) & (M - 1))];
        V2 w = ws[k];
        T er = (T)0.5 * (re[e] + pz.x), ei = (T)0.5 * (im[e] - pz.y);
        T orr = (T)0.5 * (im[e] + pz.y), oi = (T)-0.5 * (re[e] - pz.x);
        T tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;
        T xr = er + tr, xi = ei + ti;
        if (k == 0) xi = re[e] - im[e];  // the imaginary slot of bin 0 carries Re X_{N/2} (slot 4 of group 0 in the grouped layout)
        re[e] = xr * os; im[e] = xi * os;
    }
    pin_registers(re, im);   // every Z[M-k] read happens before the barrier
    BFIR_STAMP(0, 9);
    __syncthreads();
    T *ldsr = (T *)lds;
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int k = F::out_index(tid, e);
        if constexpr (ILV) {               // (re, im) pairs: slot 2k | 2k+1 (bin 0: DC | Nyquist)
            V2 v; v.x = re[e]; v.y = im[e];
            ((V2 *)ldsr)[k] = v;
        } else {                           // the reference's groups: 4 re then 4 im
            ldsr[8 * (k >> 2) + (k & 3)] = re[e];
            ldsr[8 * (k >> 2) + 4 + (k & 3)] = im[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < P / 2; j++) {
        const int idx = tid + j * NT;
        ((V4 *)dst)[idx] = ((const V4 *)ldsr)[idx];
    }
    BFIR_STAMP(0, 10);
}

// ---------------------------------------------------------------------------
// k_fwd_run: the forward kernel of fp64 engines in direct mode, one channel per workgroup, as a RUN of blocks
// ---------------------------------------------------------------------------
// A 4096-point double transform holds 66 KB of LDS: two workgroups of four waves per CU, and nothing hides the latency of
// the frame loads in front of every transform (scripts/trace_phases_f64.py: 6.6 of a workgroup's 22.7 us in cfg5, the
// frames being 8 bytes at a 16-byte stride).  Here a workgroup walks run_len consecutive blocks of ONE channel:
//   * the window's first half is the block it transformed last -- kept in registers, never loaded again;
//   * block t + 1 is fetched under the split and the stores of block t.  Vector-memory operations return in order, so a
//     twiddle load issued behind that fetch would wait for it: the twiddles come from a handful of BASES instead, loaded once per
//     workgroup (the last pass's six in registers, the earlier passes' in 1.5 KB of LDS) and multiplied up as needed --
//     the persistent fp32 kernels' scheme (pair.hip, fft_lds.h butterflies_tb).  A derived twiddle is the rounded product
//     of two rounded roots of unity, so the spectra differ from k_fwd's in the last bit or two (1e-16 relative): every
//     launch of such an engine, whatever its size, goes through this kernel, so results do not depend on the chunking.
// Addressing: buffer descriptors (wave-uniform base, 32-bit lane offset, scalar offset per access) -- sixteen 64-bit
// addresses per stream would cost the registers that keep two workgroups on a CU.
typedef unsigned int run_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int run_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t run_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long u = (unsigned long long)p;     // workgroup-uniform by construction
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void run_load(float &v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void run_load(double &v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const run_u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    v = __hiloint2double((int)q.y, (int)q.x);
}
__device__ __forceinline__ void run_store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ void run_store(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    run_u32x2 q; q.x = (unsigned)__double2loint(v); q.y = (unsigned)__double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(q, r, voff, soff, 0);
}
__device__ __forceinline__ double2 run_load_v2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const run_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    double2 v; v.x = __hiloint2double((int)q.y, (int)q.x); v.y = __hiloint2double((int)q.w, (int)q.z);
    return v;
}
__device__ __forceinline__ void run_store_v2(double2 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    run_u32x4 q;
    q.x = (unsigned)__double2loint(v.x); q.y = (unsigned)__double2hiint(v.x);
    q.z = (unsigned)__double2loint(v.y); q.w = (unsigned)__double2hiint(v.y);
    __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, soff, 0);
}

// exp(-2 pi i j / 32), j < 16 (forward convention): the rotation between the split twiddles of a thread's points, which lie
// M / 16 bins apart (W_N^(k0 + j M / 16) = W_N^k0 exp(-2 pi i j / 32), N = 2 M).  The seven constants come from constant
// memory through SCALAR loads: as literals the compiler keeps them in fourteen vector registers across the transform loop,
// which k_fwd_run<12, double> does not have (it spilled, and a scratch reload waits behind the prefetch like any other
// vector-memory load).
__constant__ double k_w32_cos[8] = {1.0, 0.9807852804032304491262, 0.9238795325112867561282, 0.8314696123025452370788,
                                    0.7071067811865475244008, 0.5555702330196022247428, 0.3826834323650897717285,
                                    0.1950903220161282678483};
template <int J> __device__ __forceinline__ void mul_w32(double &re, double &im)
{
    static_assert(J >= 0 && J < 16, "a 32nd root of unity, lower half plane");
    if constexpr (J == 0) {
    } else if constexpr (J == 8) {               // -i
        const double t = re; re = im; im = -t;
    } else {
        // cos(2 pi J / 32), sin(2 pi J / 32) from the first-octant table: cos(x) = sin(pi / 2 - x), cos(pi - x) = -cos(x)
        constexpr int jc = J < 8 ? J : 16 - J, js = J < 8 ? 8 - J : J - 8;
        const double c = J < 8 ? k_w32_cos[jc] : -k_w32_cos[jc], sn = k_w32_cos[js];
        const double tr = re * c + im * sn, ti = im * c - re * sn;
        re = tr; im = ti;
    }
}

template <int LOG2M, typename TR, bool ILV>
__global__ __launch_bounds__(FftCfg<LOG2M>::NT, 2) void k_fwd_run(FwdArgs a, const double2 *__restrict__ twb,
                                                                  const double2 *__restrict__ ws, int run_len)
{
    using T = double;
    using F = LdsFft<T, LOG2M, -1>;
    using V2 = double2;
    constexpr int M = F::M, NT = F::NT, P = F::P, N = 2 * M, H = P / 2;
    static_assert(F::radix(0) == P, "in_index(tid, e) = tid + e NT: slots e < P / 2 are the window's first half");
    __shared__ __attribute__((aligned(16))) V2 lds[F::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) V2 ldsb[F::LDSB_ELEMS];
    const int tid = threadIdx.x;
    // (run, channel): the channels of a run share input cache lines -> neighbours on one XCD
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int r = wi / a.n_ch, gc = wi - r * a.n_ch;
    const int t0 = r * run_len, t1 = min(a.n_t, t0 + run_len);
    if (t0 >= t1) return;
    const int C = a.C, g = gc / C, c = gc - g * C;
    const long ho = (long)g * a.hist_eng_stride + c;
    const TR *__restrict__ raw = (const TR *)a.raw + (long)g * a.raw_eng_stride + a.frame_off * C + c;   // frame 0 of block 0
    const T ls = (T)a.load_scale, os = (T)a.out_scale;
    // frames 2 (tid + e NT), + 1 of a block: byte offsets of the lane, between a thread's points, between the two frames
    const unsigned blk_bytes = (unsigned)M * C * (unsigned)sizeof(TR);
    const unsigned fo = (unsigned)(2 * tid) * C * (unsigned)sizeof(TR), estep = (unsigned)(2 * NT) * C * (unsigned)sizeof(TR);
    const unsigned fstep = (unsigned)C * (unsigned)sizeof(TR);
    const V2 wb = ws[tid];                                     // W_N^tid; the other fifteen split twiddles of a thread: mul_w32

    V2 B[F::NBREG];
    F::load_bases(B, ldsb, twb, tid);                          // the first exchange's barrier publishes ldsb
    T oldr[H], oldi[H];
    TR nx0[H], nx1[H];
    {
        const __amdgpu_buffer_rsrc_t ro = run_rsrc((t0 == 0) ? (const TR *)a.prev_raw + ho : raw + (long)(t0 - 1) * M * C, blk_bytes);
#pragma unroll
        for (int e = 0; e < H; e++) {
            TR x0, x1;
            run_load(x0, ro, fo, e * estep); run_load(x1, ro, fo, e * estep + fstep);
            oldr[e] = (T)x0 * ls; oldi[e] = (T)x1 * ls;
        }
        if (a.n_t == 1) {                                      // one-block chunk: the other history block moves on unchanged
            const __amdgpu_buffer_rsrc_t rcr = run_rsrc((const TR *)a.carry + ho, blk_bytes), rsp = run_rsrc((TR *)a.save_prev + ho, blk_bytes);
#pragma unroll
            for (int e = 0; e < H; e++) {
                TR x0, x1;
                run_load(x0, rcr, fo, e * estep); run_load(x1, rcr, fo, e * estep + fstep);
                run_store(x0, rsp, fo, e * estep); run_store(x1, rsp, fo, e * estep + fstep);
            }
        }
        const __amdgpu_buffer_rsrc_t rc = run_rsrc(raw + (long)t0 * M * C, blk_bytes);
#pragma unroll
        for (int e = 0; e < H; e++) { run_load(nx0[e], rc, fo, e * estep); run_load(nx1[e], rc, fo, e * estep + fstep); }
    }
    T *__restrict__ dch = (T *)a.dst + (long)gc * a.dst_ch_stride;
    for (int t = t0; t < t1; t++) {
        T re[P], im[P];
        if (t >= a.n_t - 2) {                                  // the engine's history: raw frames of the chunk's last two blocks
            const __amdgpu_buffer_rsrc_t rk = run_rsrc((TR *)(t == a.n_t - 1 ? a.save_last : a.save_prev) + ho, blk_bytes);
#pragma unroll
            for (int e = 0; e < H; e++) { run_store(nx0[e], rk, fo, e * estep); run_store(nx1[e], rk, fo, e * estep + fstep); }
        }
#pragma unroll
        for (int e = 0; e < H; e++) {                          // window = [block t-1 | block t]
            re[e] = oldr[e]; im[e] = oldi[e];
            oldr[e] = (T)nx0[e] * ls; oldi[e] = (T)nx1[e] * ls;
            re[H + e] = oldr[e]; im[H + e] = oldi[e];
        }
        {
            V2 w0[1];
            F::template butterflies<0>(re, im, w0);
        }
        static_for<1, F::NP>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            // LDS addresses are formed afresh per phase from an opaque copy of the lane index: otherwise the compiler
            // hoists dozens of them out of the transform loop (and the registers are spoken for)
            int tl = tid; asm volatile("" : "+v"(tl));
            F::template exchange<S - 1>(re, im, lds, tl);
            F::template butterflies_tb<S>(re, im, B, ldsb, tl);
        });
        {
            // block t + 1 under the split and the stores of block t (behind the butterflies: their registers are free now);
            // past the end of the run a zero-byte descriptor (zeros, no branch)
            const __amdgpu_buffer_rsrc_t rn_ = run_rsrc(raw + (long)(t + 1) * M * C, t + 1 < t1 ? blk_bytes : 0u);
#pragma unroll
            for (int e = 0; e < H; e++) { run_load(nx0[e], rn_, fo, e * estep); run_load(nx1[e], rn_, fo, e * estep + fstep); }
        }
        // Z in natural order to LDS so every thread can fetch Z[M-k]; from here on k_fwd's own steps
        int tz = tid; asm volatile("" : "+v"(tz));
        __syncthreads();
#pragma unroll
        for (int e = 0; e < P; e++) {
            V2 v; v.x = re[e]; v.y = im[e];
            lds[F::phys(F::out_index(tz, e))] = v;
        }
        __syncthreads();
        // an opaque copy per transform: left alone the compiler forms all fifteen rotated twiddles once, in front of the loop,
        // and carries 60 registers through it
        T wbx = wb.x, wby = wb.y; asm volatile("" : "+v"(wbx), "+v"(wby));
        static_for<0, P>([&](auto E_) {
            constexpr int e = decltype(E_)::value;
            static_assert(F::out_index(0, e) % NT == 0 && F::out_index(0, e) / NT < 16, "a thread's bins lie M / 16 apart");
            const int k = F::out_index(tz, e);
            V2 pz = lds[F::phys((M - k) & (M - 1))];
            T wx = wbx, wy = wby;
            mul_w32<F::out_index(0, e) / NT>(wx, wy);
            T er = (T)0.5 * (re[e] + pz.x), ei = (T)0.5 * (im[e] - pz.y);
            T orr = (T)0.5 * (im[e] + pz.y), oi = (T)-0.5 * (re[e] - pz.x);
            T tr = orr * wx - oi * wy, ti = orr * wy + oi * wx;
            T xr = er + tr, xi = ei + ti;
            if (k == 0) xi = re[e] - im[e];
            re[e] = xr * os; im[e] = xi * os;
        });
        pin_registers(re, im);
        int ty = tid; asm volatile("" : "+v"(ty));
        const __amdgpu_buffer_rsrc_t rd = run_rsrc(dch + (long)((a.base_slot + t) % a.ring) * N, (unsigned)N * 8u);
        if constexpr (ILV) {
            // (re, im) pairs, bin k at 16 k bytes: a lane's sixteen bins leave straight from its registers, 16 bytes each,
            // consecutive lanes consecutive bins (bin 0: DC | Nyquist)
            static_for<0, P>([&](auto E_) {
                constexpr int e = decltype(E_)::value;
                V2 v; v.x = re[e]; v.y = im[e];
                run_store_v2(v, rd, (unsigned)ty * 16u, (unsigned)F::out_index(0, e) * 16u);
            });
        } else {
            // the reference's groups (4 re | 4 im): through LDS, 32 bytes per lane out
            __syncthreads();
            T *ldsr = (T *)lds;
#pragma unroll
            for (int e = 0; e < P; e++) {
                const int k = F::out_index(ty, e);
                ldsr[8 * (k >> 2) + (k & 3)] = re[e];
                ldsr[8 * (k >> 2) + 4 + (k & 3)] = im[e];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < P / 2; j++) {
                const V2 *sp = (const V2 *)ldsr + 2 * (ty + j * NT);
                run_store_v2(sp[0], rd, (unsigned)ty * 32u, (unsigned)(j * NT) * 32u);
                run_store_v2(sp[1], rd, (unsigned)ty * 32u, (unsigned)(j * NT) * 32u + 16u);
            }
        }
        // the next transform's first exchange starts with a barrier
    }
}

// blocks per workgroup of k_fwd_run / k_inv_run: one round of the 512 resident workgroups (256 CUs x 2) where the launch
// has that many blocks, never more than 8 (cfg5: 42.6 / 42.8 / 42.1 Gsamples/s at 4 / 8 / 16; short runs keep the three kernels of
// the pipeline close together in time);
// BFIR_RUN64 overrides (0 = the one-transform kernels)
constexpr int BFIR_RUN64_MIN_LOG2M = 10;   // 1024 points: one wave per transform (512: half a wave, not built)
// fp64 spectra as (re, im) pairs: up to 4096 points (at 8192 the forward run kernel would spill: 256 registers without the
// sixteen 16-byte stores from registers)
constexpr int BFIR_PAIRS64_MAX_LOG2M = 12;
bool run64_supported(int filter_length, int realsize)
{
    if (const char *e = getenv("BFIR_RUN64")) if (atoi(e) == 0) return false;
    return realsize == 8 && filter_length >= (1 << BFIR_RUN64_MIN_LOG2M) && filter_length <= 8192;
}
bool pairs64_supported(int filter_length, int realsize)
{
    return run64_supported(filter_length, realsize) && filter_length <= (1 << BFIR_PAIRS64_MAX_LOG2M);
}
static int run64_len(int n_t, int n_ch)
{
    if (const char *e = getenv("BFIR_RUN64")) return atoi(e);
    const int runs = std::max(1, 512 / std::max(1, n_ch));
    return std::min(std::max(1, (n_t + runs - 1) / runs), 8);
}

template <typename T, int LOG2M> static void launch_fwd_t(const FftPlan &plan, const FwdArgs &a, int items, hipStream_t s)
{
    using V2 = typename Vec2<T>::type;
    // a transform whose LDS buffer would not fit one CU is never instantiated
    if constexpr (sizeof(T) * 2 * ((size_t(1) << LOG2M) + (size_t(1) << LOG2M) / 32) <= 160 * 1024)
    {
        if (a.raw_bytes) {           // direct mode: raw float / double frames in
            const bool il = sizeof(T) == 4 && a.interleaved;
            // fp64, 1024 ... 8192 points: runs of blocks per workgroup, one channel each -- also where two channels per
            // workgroup would fit: with the loads hidden the forward kernel is faster alone (0.35 against 0.43 ms per 32768
            // blocks of the plug-in's shape) and in the pipeline (41.9 against 40.7 Gsamples/s at 8 channels); the INVERSE keeps
            // the channel pairs where it can, whole output frames beating stores at a stride (profiles/r03_fp64.txt)
            if constexpr (sizeof(T) == 8 && LOG2M >= BFIR_RUN64_MIN_LOG2M && LOG2M <= 13) {
                int len = plan.twb ? run64_len(a.n_t, a.n_ch) : 0;
                if (len <= 0 && a.interleaved && plan.twb) len = 8;   // an engine on the pairs layout: only this kernel writes it (BFIR_RUN64 is read at creation)
                if (len > 0) {
                    const int runs = (a.n_t + len - 1) / len;
#define BFIR_LAUNCH_FWD_RUN(TR_, IL_) hipLaunchKernelGGL((k_fwd_run<LOG2M, TR_, IL_>), dim3(runs * a.n_ch), dim3(FftCfg<LOG2M>::NT), 0, s, a, (const double2 *)plan.twb, (const double2 *)plan.ws, len)
                    if constexpr (LOG2M <= BFIR_PAIRS64_MAX_LOG2M) {
                        if (a.interleaved) { if (a.raw_bytes == 4) BFIR_LAUNCH_FWD_RUN(float, true); else BFIR_LAUNCH_FWD_RUN(double, true); return; }
                    }
                    if (a.raw_bytes == 4) BFIR_LAUNCH_FWD_RUN(float, false); else BFIR_LAUNCH_FWD_RUN(double, false);
#undef BFIR_LAUNCH_FWD_RUN
                    return;
                }
            }
            if constexpr (direct_stereo_fits<T, LOG2M>()) {
                if (direct_stereo_ok(a.raw_bytes, (int)sizeof(T), a.C, a.raw, a.raw_eng_stride, a.frame_off) && !il) {
                    if (a.raw_bytes == 4)
                        hipLaunchKernelGGL((k_fwd<T, LOG2M, false, float, 2>), dim3(items / 2), dim3(2 * FftCfg<LOG2M>::NT), 0, s, a,
                                           (const V2 *)plan.tw, (const V2 *)plan.ws);
                    else
                        hipLaunchKernelGGL((k_fwd<T, LOG2M, false, T, 2>), dim3(items / 2), dim3(2 * FftCfg<LOG2M>::NT), 0, s, a,
                                           (const V2 *)plan.tw, (const V2 *)plan.ws);
                    return;
                }
            }
#define BFIR_LAUNCH_FWD_RAW(IL_, TR_) hipLaunchKernelGGL((k_fwd<T, LOG2M, IL_, TR_>), dim3(items), dim3(FftCfg<LOG2M>::NT), 0, s, a, (const V2 *)plan.tw, (const V2 *)plan.ws)
            if constexpr (sizeof(T) == 4) {
                if (il) { if (a.raw_bytes == 4) BFIR_LAUNCH_FWD_RAW(true, float); else BFIR_LAUNCH_FWD_RAW(true, double); return; }
            }
            if (a.raw_bytes == 4) BFIR_LAUNCH_FWD_RAW(false, float); else BFIR_LAUNCH_FWD_RAW(false, double);
#undef BFIR_LAUNCH_FWD_RAW
            return;
        }
        if constexpr (sizeof(T) == 4 || (LOG2M >= BFIR_RUN64_MIN_LOG2M && LOG2M <= BFIR_PAIRS64_MAX_LOG2M)) {   // fp64 pairs: the run kernels' engines
            if (a.interleaved) {
                hipLaunchKernelGGL((k_fwd<T, LOG2M, true>), dim3(items), dim3(FftCfg<LOG2M>::NT), 0, s, a,
                                   (const V2 *)plan.tw, (const V2 *)plan.ws);
                return;
            }
        }
        hipLaunchKernelGGL((k_fwd<T, LOG2M, false>), dim3(items), dim3(FftCfg<LOG2M>::NT), 0, s, a,
                           (const V2 *)plan.tw, (const V2 *)plan.ws);
    }
}

void launch_fwd(const FftPlan &plan, const FwdArgs &a, hipStream_t s)
{
    const int items = a.n_t * a.n_ch;
    if (items <= 0) return;
    switch (plan.log2m) {
#define F(lg)                                                              \
    case lg:                                                               \
        if (plan.realsize == 4) launch_fwd_t<float, lg>(plan, a, items, s); \
        else launch_fwd_t<double, lg>(plan, a, items, s);                  \
        break;
        BFIR_FOR_LOG2M(F)
#undef F
    }
}

// ---------------------------------------------------------------------------
// a11 + a12: inverse real FFT from the grouped layout, valid half only
// ---------------------------------------------------------------------------
// CPW = 2: both channels of a pair in one workgroup (see k_fwd); the two valid halves meet in LDS as whole frames and
// leave four samples (stereo) or two (wider frames) per lane.
template <typename T, int LOG2M, bool ILV, typename TR = void, int CPW = 1>
__global__ __launch_bounds__(FftCfg<LOG2M>::NT * CPW) void k_inv(InvArgs a,
                                                                 const typename Vec2<T>::type *__restrict__ tw,
                                                                 const typename Vec2<T>::type *__restrict__ ws)
{
    using F = LdsFft<T, LOG2M, +1>;
    using V2 = typename Vec2<T>::type;
    using V4 = typename Vec4<T>::type;
    constexpr int M = F::M, NT = F::NT, P = F::P, N = 2 * M;
    __shared__ __attribute__((aligned(16))) V2 lds_all[CPW][F::LDS_ELEMS];

    constexpr bool DIRECT = !std::is_void<TR>::value;
    static_assert(CPW == 1 || (CPW == 2 && DIRECT && (std::is_same<TR, float>::value || std::is_same<TR, T>::value)), "two channels per workgroup: float frames, or frames of the engine's own type");
    const int half = CPW == 1 ? 0 : (int)threadIdx.x / NT;
    const int tid = CPW == 1 ? (int)threadIdx.x : (int)threadIdx.x - half * NT;
    V2 *lds = lds_all[half];
    int wi = blockIdx.x;
    if constexpr (DIRECT) {       // the channels of a block write the same cache lines: one XCD
        const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
        wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    }
    const int ncw = a.n_ch / CPW;
    const int t = wi / ncw, gc = (wi - t * ncw) * CPW + half;
    const T *__restrict__ src = (const T *)a.src + (long)gc * a.src_ch_stride + (long)t * N;
    T *__restrict__ dst =
        (T *)a.dst + (long)gc * a.dst_ch_stride + (long)t * (a.full_output ? N : M);

    BFIR_STAMP(1, 0);
    T *ldsr = (T *)lds;
#pragma unroll
    for (int j = 0; j < P / 2; j++) {
        const int idx = tid + j * NT;
        ((V4 *)ldsr)[idx] = ((const V4 *)src)[idx];
    }
    __syncthreads();
    BFIR_STAMP(1, 9);

    // Z_k = (X_k + conj X_{M-k}) + i conj(W^k) (X_k - conj X_{M-k})
    T re[P], im[P];
    const T sc = (T)a.in_scale;
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int k = F::in_index(tid, e);
        const int q = (k == 0) ? 0 : M - k;
        T xr, xi, yr, yi;
        if constexpr (ILV) {
            const V2 vx = ((const V2 *)ldsr)[k], vy = ((const V2 *)ldsr)[q];
            xr = vx.x * sc; xi = vx.y * sc; yr = vy.x * sc; yi = vy.y * sc;
        } else {
            xr = ldsr[8 * (k >> 2) + (k & 3)] * sc; xi = ldsr[8 * (k >> 2) + 4 + (k & 3)] * sc;
            yr = ldsr[8 * (q >> 2) + (q & 3)] * sc; yi = ldsr[8 * (q >> 2) + 4 + (q & 3)] * sc;
        }
        if (k == 0) { yr = xi; xi = (T)0; yi = (T)0; }  // X_0 = (DC, 0), X_M = (Nyquist, 0)
        V2 w = ws[k];
        T ar = xr + yr, ai = xi - yi, br = xr - yr, bi = xi + yi;
        T tr = br * w.x + bi * w.y, ti = bi * w.x - br * w.y;
        re[e] = ar - ti; im[e] = ai + tr;
    }
    pin_registers(re, im);   // every read of the staged spectrum happens before run()'s first barrier

    BFIR_STAMP(1, 1);
    F::run(re, im, lds, tw, tid);

    if constexpr (DIRECT) {
        // valid half -> raw output frames of this channel, with real2raw's bookkeeping
        using RS = typename std::conditional<DIRECT, TR, float>::type;
        using Bits = decltype(abs_bits((T)0));
        __shared__ Bits red_max_all[CPW][NT / 64 > 0 ? NT / 64 : 1];
        __shared__ unsigned int red_cnt_all[CPW][NT / 64 > 0 ? NT / 64 : 1];
        Bits *red_max = red_max_all[half];
        unsigned int *red_cnt = red_cnt_all[half];
        const int C = a.C, g = gc / C, c = gc - g * C;
        RS *__restrict__ out = (RS *)a.raw + (long)g * a.raw_eng_stride + (a.frame_off + (long)t * M) * C + c;
        // CPW = 2: the frames are assembled in the first LDS buffer (2 M raw samples <= its size), which the other
        // half's waves may still be reading for their last pass
        using R2 = typename Vec2<RS>::type;
        using R4 = typename Vec4<RS>::type;
        RS *stg = (RS *)lds_all[0];
        if constexpr (CPW == 2) __syncthreads();
        const T rmax = (T)a.max, rmin = (T)(-a.max);
        Bits mx = 0; unsigned int cnt = 0u;
#pragma unroll
        for (int e = 0; e < P; e++) {
            const int m = F::out_index(tid, e);
            if (m < M / 2) {
                const T v0 = re[e], v1 = im[e];
                if constexpr (CPW == 2) { stg[4 * m + half] = (RS)v0; stg[4 * m + 2 + half] = (RS)v1; }   // frames 2m, 2m+1: (l, r, l, r)
                else { out[(long)(2 * m) * C] = (RS)v0; out[(long)(2 * m + 1) * C] = (RS)v1; }
                // brutefir/real2raw.cpp:321-336: strict compares, NaN never counts
                cnt += ((v0 < (T)0) ? (v0 < rmin) : (v0 > rmax)) ? 1u : 0u;
                cnt += ((v1 < (T)0) ? (v1 < rmin) : (v1 > rmax)) ? 1u : 0u;
                const Bits b0 = (v0 == v0) ? abs_bits(v0) : (Bits)0, b1 = (v1 == v1) ? abs_bits(v1) : (Bits)0;
                mx = b0 > mx ? b0 : mx; mx = b1 > mx ? b1 : mx;
                // brutefir/brutefir.cpp:316-321: only sample 0 of each block is checked
                if (m == 0 && !isfinite((double)v0)) flag_bad(a, t);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const Bits om = __shfl_xor(mx, o);
            mx = om > mx ? om : mx;
            cnt += __shfl_xor(cnt, o);
        }
        if ((tid & 63) == 0) { red_max[tid >> 6] = mx; red_cnt[tid >> 6] = cnt; }
        __syncthreads();
        if constexpr (CPW == 2) {
            if (C == 2) {                                      // whole frames out: 2 M samples, four per lane
                R4 *__restrict__ out4 = (R4 *)((RS *)a.raw + (long)(gc >> 1) * a.raw_eng_stride + (a.frame_off + (long)t * M) * 2);
                const R4 *stg4 = (const R4 *)stg;
#pragma unroll
                for (int j = 0; j < P / 4; j++) {
                    const int idx = (int)threadIdx.x + j * 2 * NT;
                    out4[idx] = stg4[idx];
                }
            } else {                                           // the pair's two samples of every frame, at the frame stride
                RS *__restrict__ outp = (RS *)a.raw + (long)g * a.raw_eng_stride + (a.frame_off + (long)t * M) * C + (c - half);
                const R2 *stg2 = (const R2 *)stg;
#pragma unroll
                for (int j = 0; j < P / 2; j++) {
                    const int f = (int)threadIdx.x + j * 2 * NT;   // frame 0 .. M-1
                    *(R2 *)(outp + (long)f * C) = stg2[f];
                }
            }
        }
        if (tid == 0) {
            Bits m2 = 0; unsigned int n2 = 0u;
            for (int wv = 0; wv < (NT + 63) / 64; wv++) { m2 = red_max[wv] > m2 ? red_max[wv] : m2; n2 += red_cnt[wv]; }
            DevOverflow *of = of_shard(a.overflow, a.of_shard_stride) + gc;
            if (n2) atomicAdd(&of->n_overflows, n2);
            if ((unsigned long long)m2 > *(volatile unsigned long long *)&of->largest_bits)
                atomicMax(&of->largest_bits, (unsigned long long)m2);
        }
    } else
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int m = F::out_index(tid, e);
        if (a.full_output || m < M / 2) {
            V2 v; v.x = re[e]; v.y = im[e];
            *(V2 *)(dst + 2 * m) = v;
        }
    }
    BFIR_STAMP(1, 10);
}

// ---------------------------------------------------------------------------
// k_inv_run has no register to spare for flag_bad's second store (256 VGPRs: the ISA audit found 8 bytes of scratch), so on the
// latency path one thread turns its atomicMin verdict into the host flag afterwards -- what the 4-byte copy used to cost, for
// the engines on this kernel only (fp64 with odd channel counts).
__global__ void k_publish_bad(const int *__restrict__ bad_block, int block_base, int n_t, int *__restrict__ bad_host)
{
    const int b = *bad_block - block_base;
    if (b >= 0 && b < n_t) bad_host[b] = 1;
}

// k_inv_run: the inverse kernel of fp64 engines in direct mode, one channel per workgroup, as a RUN of blocks
// ---------------------------------------------------------------------------
// k_fwd_run's counterpart: the product spectrum of block t + 1 (N doubles: 32 bytes x P / 2 per lane) is fetched into
// registers under the output phase of block t -- behind the butterflies, whose registers are free by then -- twiddles from
// bases, the split twiddles from one table entry per thread times constant 32nd roots.  Steps, bookkeeping (overflow
// statistics, NaN verdict: real2raw.cpp:321-336, brutefir.cpp:316-321) and output addressing are k_inv's.
template <int LOG2M, typename TR, bool ILV>
__global__ __launch_bounds__(FftCfg<LOG2M>::NT, 2) void k_inv_run(InvArgs a, const double2 *__restrict__ twb,
                                                                  const double2 *__restrict__ ws, int run_len)
{
    using T = double;
    using F = LdsFft<T, LOG2M, +1>;
    using V2 = double2;
    constexpr int M = F::M, NT = F::NT, P = F::P, N = 2 * M;
    static_assert(F::radix(0) == P, "in_index(tid, e) = tid + e NT");
    __shared__ __attribute__((aligned(16))) V2 lds[F::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) V2 ldsb[F::LDSB_ELEMS];
    using Bits = unsigned long long;
    __shared__ Bits red_max[NT / 64 > 0 ? NT / 64 : 1];
    __shared__ unsigned int red_cnt[NT / 64 > 0 ? NT / 64 : 1];
    const int tid = threadIdx.x;
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int r = wi / a.n_ch, gc = wi - r * a.n_ch;
    const int t0 = r * run_len, t1 = min(a.n_t, t0 + run_len);
    if (t0 >= t1) return;
    const int C = a.C, g = gc / C, c = gc - g * C;
    const T *__restrict__ sch = (const T *)a.src + (long)gc * a.src_ch_stride;
    TR *__restrict__ raw = (TR *)a.raw + (long)g * a.raw_eng_stride + a.frame_off * C + c;   // frame 0 of block 0
    const unsigned blk_bytes = (unsigned)M * C * (unsigned)sizeof(TR);
    const unsigned fo = (unsigned)(2 * tid) * C * (unsigned)sizeof(TR), fstep = (unsigned)C * (unsigned)sizeof(TR);
    const T sc = (T)a.in_scale, rmax = (T)a.max, rmin = (T)(-a.max);
    const V2 wb = ws[tid];
    V2 B[F::NBREG];
    F::load_bases(B, ldsb, twb, tid);

    V2 nx[P];                                                  // the next block's spectrum as it lies in memory: quad tid + j NT
    {
        const __amdgpu_buffer_rsrc_t rs = run_rsrc(sch + (long)t0 * N, (unsigned)N * 8u);
#pragma unroll
        for (int j = 0; j < P / 2; j++) {
            nx[2 * j] = run_load_v2(rs, (unsigned)tid * 32u, (unsigned)(j * NT) * 32u);
            nx[2 * j + 1] = run_load_v2(rs, (unsigned)tid * 32u, (unsigned)(j * NT) * 32u + 16u);
        }
    }
    for (int t = t0; t < t1; t++) {
        // the spectrum to LDS (the last reads of this buffer lie in front of the barrier that closes the loop body)
        T *ldsr = (T *)lds;
        int tl0 = tid; asm volatile("" : "+v"(tl0));
#pragma unroll
        for (int j = 0; j < P / 2; j++) {
            ((V2 *)ldsr)[2 * (tl0 + j * NT)] = nx[2 * j];
            ((V2 *)ldsr)[2 * (tl0 + j * NT) + 1] = nx[2 * j + 1];
        }
        __syncthreads();
        // Z_k = (X_k + conj X_{M-k}) + i conj(W^k) (X_k - conj X_{M-k})
        T re[P], im[P];
        T wbx = wb.x, wby = wb.y; asm volatile("" : "+v"(wbx), "+v"(wby));
        static_for<0, P>([&](auto E_) {
            constexpr int e = decltype(E_)::value;
            static_assert(F::in_index(0, e) == e * NT, "a thread's bins lie M / 16 apart");
            const int k = tl0 + e * NT;
            const int q = (k == 0) ? 0 : M - k;
            T xr, xi, yr, yi;
            if constexpr (ILV) {
                const V2 vx = ((const V2 *)ldsr)[k], vy = ((const V2 *)ldsr)[q];
                xr = vx.x * sc; xi = vx.y * sc; yr = vy.x * sc; yi = vy.y * sc;
            } else {
                xr = ldsr[8 * (k >> 2) + (k & 3)] * sc; xi = ldsr[8 * (k >> 2) + 4 + (k & 3)] * sc;
                yr = ldsr[8 * (q >> 2) + (q & 3)] * sc; yi = ldsr[8 * (q >> 2) + 4 + (q & 3)] * sc;
            }
            if (k == 0) { yr = xi; xi = (T)0; yi = (T)0; }      // X_0 = (DC, 0), X_M = (Nyquist, 0)
            T wx = wbx, wy = wby;
            mul_w32<e>(wx, wy);
            T ar = xr + yr, ai = xi - yi, br = xr - yr, bi = xi + yi;
            T tr = br * wx + bi * wy, ti = bi * wx - br * wy;
            re[e] = ar - ti; im[e] = ai + tr;
        });
        pin_registers(re, im);   // every read of the staged spectrum happens before the first exchange's barrier
        {
            // block t + 1's spectrum under the transform and the output of block t (the output phase alone is too short); past the end of the run a zero-byte descriptor
            const __amdgpu_buffer_rsrc_t rs = run_rsrc(sch + (long)(t + 1) * N, t + 1 < t1 ? (unsigned)N * 8u : 0u);
#pragma unroll
            for (int j = 0; j < P / 2; j++) {
                nx[2 * j] = run_load_v2(rs, (unsigned)tid * 32u, (unsigned)(j * NT) * 32u);
                nx[2 * j + 1] = run_load_v2(rs, (unsigned)tid * 32u, (unsigned)(j * NT) * 32u + 16u);
            }
        }
        {
            V2 w0[1];
            F::template butterflies<0>(re, im, w0);
        }
        static_for<1, F::NP>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            int tl = tid; asm volatile("" : "+v"(tl));
            F::template exchange<S - 1>(re, im, lds, tl);
            F::template butterflies_tb<S>(re, im, B, ldsb, tl);
        });
        // valid half -> raw output frames of this channel, with real2raw's bookkeeping
        const __amdgpu_buffer_rsrc_t ro = run_rsrc(raw + (long)t * M * C, blk_bytes);
        Bits mx = 0; unsigned int cnt = 0u;
        static_for<0, P>([&](auto E_) {
            constexpr int e = decltype(E_)::value;
            constexpr int m0 = F::out_index(0, e);
            if constexpr (m0 < M / 2) {                         // point m = tid + m0: frames 2m, 2m+1
                const T v0 = re[e], v1 = im[e];
                run_store((TR)v0, ro, fo, (unsigned)(2 * m0) * fstep);
                run_store((TR)v1, ro, fo, (unsigned)(2 * m0 + 1) * fstep);
                // brutefir/real2raw.cpp:321-336: strict compares, NaN never counts
                cnt += ((v0 < (T)0) ? (v0 < rmin) : (v0 > rmax)) ? 1u : 0u;
                cnt += ((v1 < (T)0) ? (v1 < rmin) : (v1 > rmax)) ? 1u : 0u;
                const Bits b0 = (v0 == v0) ? abs_bits(v0) : (Bits)0, b1 = (v1 == v1) ? abs_bits(v1) : (Bits)0;
                mx = b0 > mx ? b0 : mx; mx = b1 > mx ? b1 : mx;
                // brutefir/brutefir.cpp:316-321: only sample 0 of each block is checked
                if (m0 == 0 && tid == 0 && !isfinite(v0)) atomicMin(a.bad_block, a.block_base + t);   // (no host flag here: see k_publish_bad)
            }
        });
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const Bits om = __shfl_xor(mx, o);
            mx = om > mx ? om : mx;
            cnt += __shfl_xor(cnt, o);
        }
        if ((tid & 63) == 0) { red_max[tid >> 6] = mx; red_cnt[tid >> 6] = cnt; }
        __syncthreads();                                       // also: every wave is through with the transform's LDS buffer
        if (tid == 0) {
            Bits m2 = 0; unsigned int n2 = 0u;
            for (int wv = 0; wv < (NT + 63) / 64; wv++) { m2 = red_max[wv] > m2 ? red_max[wv] : m2; n2 += red_cnt[wv]; }
            DevOverflow *of = of_shard(a.overflow, a.of_shard_stride) + gc;
            if (n2) atomicAdd(&of->n_overflows, n2);
            // no look at the current maximum first (k_inv does): a load here would wait for the prefetch
            if (m2) atomicMax(&of->largest_bits, (unsigned long long)m2);
        }
    }
}

template <typename T, int LOG2M> static void launch_inv_t(const FftPlan &plan, const InvArgs &a, int items, hipStream_t s)
{
    using V2 = typename Vec2<T>::type;
    // a transform whose LDS buffer would not fit one CU is never instantiated
    if constexpr (sizeof(T) * 2 * ((size_t(1) << LOG2M) + (size_t(1) << LOG2M) / 32) <= 160 * 1024)
    {
        if (a.raw_bytes) {           // direct mode: raw float / double frames out
            const bool il = sizeof(T) == 4 && a.interleaved;
            if constexpr (direct_stereo_fits<T, LOG2M>()) {
                if (direct_stereo_ok(a.raw_bytes, (int)sizeof(T), a.C, a.raw, a.raw_eng_stride, a.frame_off) && !il) {
#define BFIR_LAUNCH_INV_PAIR(TR_, IL_) hipLaunchKernelGGL((k_inv<T, LOG2M, IL_, TR_, 2>), dim3(items / 2), dim3(2 * FftCfg<LOG2M>::NT), 0, s, a, (const V2 *)plan.tw, (const V2 *)plan.ws)
                    if constexpr (sizeof(T) == 8 && LOG2M >= BFIR_RUN64_MIN_LOG2M && LOG2M <= BFIR_PAIRS64_MAX_LOG2M) {   // (re, im) pairs: the run kernels' engines
                        if (a.interleaved) { if (a.raw_bytes == 4) BFIR_LAUNCH_INV_PAIR(float, true); else BFIR_LAUNCH_INV_PAIR(T, true); return; }
                    }
                    if (a.raw_bytes == 4) BFIR_LAUNCH_INV_PAIR(float, false); else BFIR_LAUNCH_INV_PAIR(T, false);
#undef BFIR_LAUNCH_INV_PAIR
                    return;
                }
            }
            if constexpr (sizeof(T) == 8 && LOG2M >= BFIR_RUN64_MIN_LOG2M && LOG2M <= 13) {
                int len = plan.twb && !a.full_output ? run64_len(a.n_t, a.n_ch) : 0;
                if (len <= 0 && a.interleaved && plan.twb && !a.full_output) len = 8;   // pairs layout: the one-transform kernels below read groups
                if (len > 0) {
                    const int runs = (a.n_t + len - 1) / len;
#define BFIR_LAUNCH_INV_RUN(TR_, IL_) hipLaunchKernelGGL((k_inv_run<LOG2M, TR_, IL_>), dim3(runs * a.n_ch), dim3(FftCfg<LOG2M>::NT), 0, s, a, (const double2 *)plan.twb, (const double2 *)plan.ws, len)
                    if constexpr (LOG2M <= BFIR_PAIRS64_MAX_LOG2M) {
                        if (a.interleaved) {
                            if (a.raw_bytes == 4) BFIR_LAUNCH_INV_RUN(float, true); else BFIR_LAUNCH_INV_RUN(double, true);
                            if (a.bad_host) hipLaunchKernelGGL(k_publish_bad, dim3(1), dim3(1), 0, s, a.bad_block, a.block_base, a.n_t, a.bad_host);
                            return;
                        }
                    }
                    if (a.raw_bytes == 4) BFIR_LAUNCH_INV_RUN(float, false); else BFIR_LAUNCH_INV_RUN(double, false);
#undef BFIR_LAUNCH_INV_RUN
                    if (a.bad_host) hipLaunchKernelGGL(k_publish_bad, dim3(1), dim3(1), 0, s, a.bad_block, a.block_base, a.n_t, a.bad_host);
                    return;
                }
            }
#define BFIR_LAUNCH_INV_RAW(IL_, TR_) hipLaunchKernelGGL((k_inv<T, LOG2M, IL_, TR_>), dim3(items), dim3(FftCfg<LOG2M>::NT), 0, s, a, (const V2 *)plan.tw, (const V2 *)plan.ws)
            if constexpr (sizeof(T) == 4) {
                if (il) { if (a.raw_bytes == 4) BFIR_LAUNCH_INV_RAW(true, float); else BFIR_LAUNCH_INV_RAW(true, double); return; }
            }
            if (a.raw_bytes == 4) BFIR_LAUNCH_INV_RAW(false, float); else BFIR_LAUNCH_INV_RAW(false, double);
#undef BFIR_LAUNCH_INV_RAW
            return;
        }
        if constexpr (sizeof(T) == 4) {
            if (a.interleaved) {
                hipLaunchKernelGGL((k_inv<T, LOG2M, true>), dim3(items), dim3(FftCfg<LOG2M>::NT), 0, s, a,
                                   (const V2 *)plan.tw, (const V2 *)plan.ws);
                return;
            }
        }
        hipLaunchKernelGGL((k_inv<T, LOG2M, false>), dim3(items), dim3(FftCfg<LOG2M>::NT), 0, s, a,
                           (const V2 *)plan.tw, (const V2 *)plan.ws);
    }
}

void launch_inv(const FftPlan &plan, const InvArgs &a, hipStream_t s)
{
    const int items = a.n_t * a.n_ch;
    if (items <= 0) return;
    switch (plan.log2m) {
#define F(lg)                                                              \
    case lg:                                                               \
        if (plan.realsize == 4) launch_inv_t<float, lg>(plan, a, items, s); \
        else launch_inv_t<double, lg>(plan, a, items, s);                  \
        break;
        BFIR_FOR_LOG2M(F)
#undef F
    }
}

// ---------------------------------------------------------------------------
// a8-a10: streaming complex multiply-accumulate over the partitions
// ---------------------------------------------------------------------------
// One thread owns one group (4 bins: V4 of real parts + V4 of imaginary parts)
// of one channel for TT consecutive output blocks.  Walking the partitions
// i = 0..nb-1 in the reference's order, it streams H_i once per tile and keeps
// a sliding window of TT delay-line spectra in registers, so a spectrum is
// fetched once per tile instead of once per output block.
// Explicit fma() calls pin the rounding: every instantiation (any TT) produces
// bit-identical sums, so results do not depend on how a run is cut into chunks.
// fp32: written on 2-wide vectors so it is v_pk_fma_f32 by construction (the library
// is built without SLP vectorisation).  Measured on MI355X (scripts/ubench/valu_rate.hip,
// profiles/r01_valu_rate.txt): a wave64 v_fma_f32 issues every ~4 cycles per SIMD
// (59-75 TFLOP/s chip-wide for 1-8 waves/SIMD), a v_pk_fma_f32 every ~7 (71-90 TFLOP/s),
// so the packed form is worth 15-20 % here.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cmac4(float4 &ar, float4 &ai, const float4 &xr, const float4 &xi,
                                      const float4 &hr, const float4 &hi)
{
    const v2f xrl = {xr.x, xr.y}, xrh = {xr.z, xr.w}, xil = {xi.x, xi.y}, xih = {xi.z, xi.w};
    const v2f hrl = {hr.x, hr.y}, hrh = {hr.z, hr.w}, hil = {hi.x, hi.y}, hih = {hi.z, hi.w};
    v2f arl = {ar.x, ar.y}, arh = {ar.z, ar.w}, ail = {ai.x, ai.y}, aih = {ai.z, ai.w};
    arl = __builtin_elementwise_fma(xrl, hrl, arl); arl = __builtin_elementwise_fma(-xil, hil, arl);
    arh = __builtin_elementwise_fma(xrh, hrh, arh); arh = __builtin_elementwise_fma(-xih, hih, arh);
    ail = __builtin_elementwise_fma(xrl, hil, ail); ail = __builtin_elementwise_fma(xil, hrl, ail);
    aih = __builtin_elementwise_fma(xrh, hih, aih); aih = __builtin_elementwise_fma(xih, hrh, aih);
    ar.x = arl.x; ar.y = arl.y; ar.z = arh.x; ar.w = arh.y;
    ai.x = ail.x; ai.y = ail.y; ai.z = aih.x; ai.w = aih.y;
}

template <typename V4> __device__ __forceinline__ void cmac4(V4 &ar, V4 &ai, const V4 &xr, const V4 &xi,
                                                             const V4 &hr, const V4 &hi)
{
    ar.x = fma(xr.x, hr.x, ar.x); ar.x = fma(-xi.x, hi.x, ar.x); ai.x = fma(xr.x, hi.x, ai.x); ai.x = fma(xi.x, hr.x, ai.x);
    ar.y = fma(xr.y, hr.y, ar.y); ar.y = fma(-xi.y, hi.y, ar.y); ai.y = fma(xr.y, hi.y, ai.y); ai.y = fma(xi.y, hr.y, ai.y);
    ar.z = fma(xr.z, hr.z, ar.z); ar.z = fma(-xi.z, hi.z, ar.z); ai.z = fma(xr.z, hi.z, ai.z); ai.z = fma(xi.z, hr.z, ai.z);
    ar.w = fma(xr.w, hr.w, ar.w); ar.w = fma(-xi.w, hi.w, ar.w); ai.w = fma(xr.w, hi.w, ai.w); ai.w = fma(xi.w, hr.w, ai.w);
}

// The partition loop of one thread.  Partitions in the reference's order
// i = 0 .. nb-1, TT per trip so the rotating window keeps compile-time register
// names.  Operands are fetched D steps ahead into a small register queue: step i
// consumes H_i and the one new delay-line spectrum X[t0-i] (entering the window
// in place of the entry nobody needs any more) from queue slot i mod D, refills
// that slot with the operands of step i+D, then does its TT x 4 complex MACs, so
// D steps of arithmetic cover the load latency.  DCNY: also carry the two real
// sums of group 0 (DC, Nyquist); only the wave owning group 0 runs that version.
template <typename T, int TT, int D, bool DCNY, typename V4>
__device__ __forceinline__ void mac_partitions(V4 (&accr)[TT], V4 (&acci)[TT], T (&dc)[TT], T (&ny)[TT],
                                               V4 (&wr)[TT], V4 (&wi)[TT], const V4 *__restrict__ X,
                                               const V4 *__restrict__ H, int nb, int ring, int sl, long slot4)
{
    V4 qhr[D], qhi[D], qxr[D], qxi[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const int ih = d < nb ? d : nb - 1;
        qhr[d] = H[ih * slot4]; qhi[d] = H[ih * slot4 + 1];
        int sd = sl - d; if (sd < 0) sd += ring;       // slot of X[t0 - d]; d = 0 is never consumed
        qxr[d] = X[sd * slot4]; qxi[d] = X[sd * slot4 + 1];
    }
    int sp = sl - D; if (sp < 0) sp += ring;           // slot of X[t0 - (i + D)] for i = 0
    for (int i0 = 0; i0 < nb; i0 += TT) {
#pragma unroll
        for (int ii = 0; ii < TT; ii++) {
            const int i = i0 + ii;
            if (i < nb) {   // wave-uniform
                constexpr int TTc = TT;
                const V4 hr = qhr[ii % D], hi = qhi[ii % D];
                if (i > 0) { wr[(TTc - ii) % TTc] = qxr[ii % D]; wi[(TTc - ii) % TTc] = qxi[ii % D]; }
                const int ih = (i + D < nb) ? i + D : nb - 1;
                qhr[ii % D] = H[ih * slot4]; qhi[ii % D] = H[ih * slot4 + 1];
                qxr[ii % D] = X[sp * slot4]; qxi[ii % D] = X[sp * slot4 + 1];
                sp -= 1; if (sp < 0) sp += ring;
#pragma unroll
                for (int j = 0; j < TT; j++) {
                    const int idx = (j - ii + TTc) % TTc;  // window slot holding X[t0 + j - i]
                    cmac4(accr[j], acci[j], wr[idx], wi[idx], hr, hi);
                    if constexpr (DCNY) {
                        dc[j] = fma(wr[idx].x, hr.x, dc[j]);
                        ny[j] = fma(wi[idx].x, hi.x, ny[j]);
                    }
                }
            }
        }
    }
}

template <typename T, int TT, int WPE, int D>
__global__ __launch_bounds__(256, WPE) void k_mac(MacArgs a, int nbt, int nTT, int G)
{
    static_assert(D >= 1 && TT % D == 0, "prefetch depth must divide the time tile");
    using V4 = typename Vec4<T>::type;
    // XCD-aware, bijective block -> work remap: each XCD (blocks b, b+8, ...)
    // gets one contiguous range of work items, ordered (channel, bin tile)
    // major / time tile minor, so its L2 holds a slice of H for the whole
    // launch while consecutive time tiles re-use each other's spectra.
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / nTT, tt = w - s * nTT;
    const int gc = s / nbt, bt = s - gc * nbt;
    const int g = bt * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int t0 = tt * TT;
    const long slot4 = a.N / 4;  // V4 elements per spectrum
    const V4 *__restrict__ X = (const V4 *)((const T *)a.x + (long)gc * a.x_ch_stride) + 2 * g;
    const V4 *__restrict__ H = (const V4 *)((const T *)a.h + (long)gc * a.h_ch_stride) + 2 * g;
    const int nb = a.nblk[gc];
    const int ring = a.ring;
    // DC and Nyquist share group 0 as two independent reals; only the wave that
    // owns group 0 carries the two extra sums (a scalar, wave-uniform branch)
    const bool wave0 = __builtin_amdgcn_readfirstlane((int)((bt == 0) && (threadIdx.x < 64))) != 0;

    V4 accr[TT], acci[TT], wr[TT], wi[TT];
    T dc[TT], ny[TT];
    const int sl = (a.base_slot + t0) % ring;  // delay-line slot of block t0
#pragma unroll
    for (int j = 0; j < TT; j++) {
        accr[j] = V4{0, 0, 0, 0}; acci[j] = V4{0, 0, 0, 0};
        dc[j] = (T)0; ny[j] = (T)0;
        int sj = sl + j; if (sj >= ring) sj -= ring;
        wr[j] = X[sj * slot4]; wi[j] = X[sj * slot4 + 1];
    }
    if (wave0) mac_partitions<T, TT, D, true>(accr, acci, dc, ny, wr, wi, X, H, nb, ring, sl, slot4);
    else mac_partitions<T, TT, D, false>(accr, acci, dc, ny, wr, wi, X, H, nb, ring, sl, slot4);
    T *__restrict__ Y = (T *)a.y + (long)gc * a.y_ch_stride;
#pragma unroll
    for (int j = 0; j < TT; j++) {
        const int t = t0 + j;
        if (t < a.n_t) {
            if (g == 0) { accr[j].x = dc[j]; acci[j].x = ny[j]; }
            V4 *yo = (V4 *)(Y + (long)t * a.N) + 2 * g;
            yo[0] = accr[j]; yo[1] = acci[j];
        }
    }
}

// ---------------------------------------------------------------------------
// k_mac_lds: the same sums with the operands shared through LDS (fp32)
// ---------------------------------------------------------------------------
// A workgroup of 4 waves owns 64 groups (256 bins) of one channel for 32
// consecutive output blocks: wave w computes blocks tb+8w .. tb+8w+7 with the
// same rotating register window as k_mac.  Per partition step the workgroup
// needs H_i (shared by all four waves) and, per wave, one new delay-line
// spectrum X[tb+8w-i] -- which is exactly what wave w-1 fetched 8 steps
// earlier.  So the spectra live in a 32-entry LDS ring: only X[tb-i] and H_i
// are fetched from memory each step (4 KiB per workgroup-step instead of
// 16 KiB), one 16-byte load per lane, issued D steps ahead; everything else is
// ds_read_b128 of lane-contiguous data (conflict free).  One barrier per step.
// Sums are formed in the same order with the same fma chain as k_mac, so the
// two kernels give bit-identical results.
// Native 4-wide vectors (not HIP's float4 struct): they are register-tuple aligned, so their
// .lo / .hi halves feed v_pk_fma_f32 directly; scalarised struct members needed ~30 v_mov per step.
typedef float v4f __attribute__((ext_vector_type(4)));
// nhi = -hi, negated once per step by the caller (fma(-xi, hi, ar) == fma(xi, -hi, ar) exactly)
__device__ __forceinline__ void cmac4(v4f &ar, v4f &ai, const v4f &xr, const v4f &xi, const v4f &hr, const v4f &hi,
                                      const v4f &nhi)
{
    ar.lo = __builtin_elementwise_fma(xr.lo, hr.lo, ar.lo); ar.lo = __builtin_elementwise_fma(xi.lo, nhi.lo, ar.lo);
    ar.hi = __builtin_elementwise_fma(xr.hi, hr.hi, ar.hi); ar.hi = __builtin_elementwise_fma(xi.hi, nhi.hi, ar.hi);
    ai.lo = __builtin_elementwise_fma(xr.lo, hi.lo, ai.lo); ai.lo = __builtin_elementwise_fma(xi.lo, hr.lo, ai.lo);
    ai.hi = __builtin_elementwise_fma(xr.hi, hi.hi, ai.hi); ai.hi = __builtin_elementwise_fma(xi.hi, hr.hi, ai.hi);
}

// (re, im) pair layout: a group is A = (r0 i0 r1 i1), B = (r2 i2 r3 i3); the same four fused
// multiply-adds per bin as cmac4, on the components where they lie (no shuffle, same rounding)
__device__ __forceinline__ void cmac4_pairs(v4f &aA, v4f &aB, const v4f &xA, const v4f &xB, const v4f &hA, const v4f &hB)
{
    aA.x = fmaf(xA.x, hA.x, aA.x); aA.x = fmaf(xA.y, -hA.y, aA.x); aA.y = fmaf(xA.x, hA.y, aA.y); aA.y = fmaf(xA.y, hA.x, aA.y);
    aA.z = fmaf(xA.z, hA.z, aA.z); aA.z = fmaf(xA.w, -hA.w, aA.z); aA.w = fmaf(xA.z, hA.w, aA.w); aA.w = fmaf(xA.w, hA.z, aA.w);
    aB.x = fmaf(xB.x, hB.x, aB.x); aB.x = fmaf(xB.y, -hB.y, aB.x); aB.y = fmaf(xB.x, hB.y, aB.y); aB.y = fmaf(xB.y, hB.x, aB.y);
    aB.z = fmaf(xB.z, hB.z, aB.z); aB.z = fmaf(xB.w, -hB.w, aB.z); aB.w = fmaf(xB.z, hB.w, aB.w); aB.w = fmaf(xB.w, hB.z, aB.w);
}

// PAIRS: the spectra are (re, im) pairs (MacArgs.interleaved); "planes" 0 / 1 of a group are then its
// first / second 16 bytes (bins 0-1 / bins 2-3) instead of the real / imaginary parts.
template <int D, bool DCNY, bool PAIRS>
__device__ __forceinline__ void mac_lds_steps(v4f (&accr)[8], v4f (&acci)[8], float (&dc)[8], float (&ny)[8],
                                              v4f (&wr)[8], v4f (&wi)[8], v4f (*s_ring)[2][64],
                                              v4f (*s_h)[2][64], const v4f *__restrict__ duty_base,
                                              long duty_slot4, bool duty_is_h, int duty_plane, int nb, int ring,
                                              int sl_tb, int lane, int wv)
{
    // duty: this wave fetches one plane (re or im) of H_s (waves 0,1) or of
    // X[tb - s] (waves 2,3) for every step s, D steps ahead of its use
    v4f q[D];
    const v4f *__restrict__ dbase = duty_base;
    // The duty operand of step s sits at slot index dnext when s is the next one to fetch:
    // H waves walk up (clamped to the last partition), X waves walk down the ring with wrap.
    int dnext = duty_is_h ? 0 : sl_tb;
#define BFIR_DUTY_ADVANCE()                                                                         \
    do {                                                                                            \
        if (duty_is_h) { if (dnext < nb - 1) dnext += 1; }                                          \
        else { dnext -= 1; if (dnext < 0) dnext += ring; }                                          \
    } while (0)
#define BFIR_DUTY_LOAD() dbase[dnext * duty_slot4 + duty_plane]
#define BFIR_DUTY_STORE(s_, v_)                                                                     \
    do {                                                                                            \
        v4f *dst_ = duty_is_h ? &s_h[(s_) & 1][duty_plane][lane] : &s_ring[(-(s_)) & 31][duty_plane][lane]; \
        *dst_ = (v_);                                                                               \
    } while (0)
    // step 0: H_0 straight to LDS (X waves have no duty: the windows are loaded already)
    if (duty_is_h) { const v4f h0 = BFIR_DUTY_LOAD(); BFIR_DUTY_STORE(0, h0); }
#pragma unroll
    for (int d = 0; d < D; d++) { BFIR_DUTY_ADVANCE(); q[(1 + d) % D] = BFIR_DUTY_LOAD(); }   // steps 1 .. D
    __syncthreads();
    for (int i0 = 0; i0 < nb; i0 += 8) {
#pragma unroll
        for (int ii = 0; ii < 8; ii++) {
            const int i = i0 + ii;
            if (i < nb) {   // uniform over the workgroup
                const v4f hr = s_h[ii & 1][0][lane], hi = s_h[ii & 1][1][lane];
                const v4f nhi = -hi;
                if (i > 0) {
                    const int e = (8 * wv - i) & 31;             // ring entry holding X[tb + 8 wv - i]
                    wr[(8 - ii) % 8] = s_ring[e][0][lane]; wi[(8 - ii) % 8] = s_ring[e][1][lane];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int idx = (j - ii + 8) % 8;            // window slot holding X[t0 + j - i]
                    if constexpr (PAIRS) {
                        cmac4_pairs(accr[j], acci[j], wr[idx], wi[idx], hr, hi);
                        if constexpr (DCNY) {              // bin 0 = (DC, Nyquist): .x and .y of the first half
                            dc[j] = fma(wr[idx].x, hr.x, dc[j]);
                            ny[j] = fma(wr[idx].y, hr.y, ny[j]);
                        }
                    } else {
                        cmac4(accr[j], acci[j], wr[idx], wi[idx], hr, hi, nhi);
                        if constexpr (DCNY) {
                            dc[j] = fma(wr[idx].x, hr.x, dc[j]);
                            ny[j] = fma(wi[idx].x, hi.x, ny[j]);
                        }
                    }
                }
                // publish the operands of step i+1, refill the queue slot with those of step i+1+D
                BFIR_DUTY_STORE(i + 1, q[(ii + 1) % D]);
                BFIR_DUTY_ADVANCE();
                q[(ii + 1) % D] = BFIR_DUTY_LOAD();
                __syncthreads();
            }
        }
    }
#undef BFIR_DUTY_ADVANCE
#undef BFIR_DUTY_LOAD
#undef BFIR_DUTY_STORE
}

template <int D, bool PAIRS>
__global__ __launch_bounds__(256, 2) void k_mac_lds(MacArgs a, int nbt, int nTQ)
{
    __shared__ __attribute__((aligned(16))) v4f s_ring[32][2][64];
    __shared__ __attribute__((aligned(16))) v4f s_h[2][2][64];
    static_assert(8 % D == 0, "prefetch depth must divide the unroll");
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / nTQ, tq = w - s * nTQ;
    const int gc = s / nbt, bt = s - gc * nbt;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = bt * 64 + lane;
    const int tb = tq * 32, t0 = tb + 8 * wv;
    const long slot4 = a.N / 4;
    const v4f *__restrict__ X = (const v4f *)((const float *)a.x + (long)gc * a.x_ch_stride) + 2 * g;
    const v4f *__restrict__ H = (const v4f *)((const float *)a.h + (long)gc * a.h_ch_stride) + 2 * g;
    const int nb = a.nblk[gc];
    const int ring = a.ring;
    const int sl_tb = (a.base_slot + tb) % ring;   // delay-line slot of block tb
    v4f accr[8], acci[8], wr[8], wi[8];
    float dc[8], ny[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        accr[j] = v4f{0, 0, 0, 0}; acci[j] = v4f{0, 0, 0, 0};
        dc[j] = 0.f; ny[j] = 0.f;
        int sj = sl_tb + 8 * wv + j; if (sj >= ring) sj -= ring;
        wr[j] = X[sj * slot4]; wi[j] = X[sj * slot4 + 1];
        s_ring[8 * wv + j][0][lane] = wr[j]; s_ring[8 * wv + j][1][lane] = wi[j];
    }
    const bool duty_is_h = wv < 2;
    const int plane = wv & 1;
    if (bt == 0)
        mac_lds_steps<D, true, PAIRS>(accr, acci, dc, ny, wr, wi, s_ring, s_h, duty_is_h ? H : X, slot4, duty_is_h, plane,
                               nb, ring, sl_tb, lane, wv);
    else
        mac_lds_steps<D, false, PAIRS>(accr, acci, dc, ny, wr, wi, s_ring, s_h, duty_is_h ? H : X, slot4, duty_is_h, plane,
                                nb, ring, sl_tb, lane, wv);
    float *__restrict__ Y = (float *)a.y + (long)gc * a.y_ch_stride;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int t = t0 + j;
        if (t < a.n_t) {
            if (g == 0) {
                if constexpr (PAIRS) { accr[j].x = dc[j]; accr[j].y = ny[j]; }
                else { accr[j].x = dc[j]; acci[j].x = ny[j]; }
            }
            v4f *yo = (v4f *)(Y + (long)t * a.N) + 2 * g;
            yo[0] = accr[j]; yo[1] = acci[j];
        }
    }
}

// ---------------------------------------------------------------------------
// k_mac_lds_d: the LDS-shared MAC for fp64 (grouped layout)
// ---------------------------------------------------------------------------
// k_mac<double> has room for one wave per SIMD only (4 output blocks x 4 bins of accumulators and
// window are 128 registers before anything else) and fetches every operand from L2 itself.  This
// is k_mac_lds with half the time tile: 4 waves x 4 blocks = 16 blocks per workgroup, a 16-entry
// ring of 32-byte lanes (64 KiB) + the double-buffered H planes (8 KiB): two workgroups and two
// waves per SIMD per CU, one 32-byte duty load per lane and step.  Same fma chain as k_mac.
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cmac4d(v4d &ar, v4d &ai, const v4d &xr, const v4d &xi, const v4d &hr, const v4d &hi)
{
    ar = __builtin_elementwise_fma(xr, hr, ar); ar = __builtin_elementwise_fma(-xi, hi, ar);
    ai = __builtin_elementwise_fma(xr, hi, ai); ai = __builtin_elementwise_fma(xi, hr, ai);
}

template <int D, bool DCNY>
__device__ __forceinline__ void mac_lds_steps_d(v4d (&accr)[4], v4d (&acci)[4], double (&dc)[4], double (&ny)[4],
                                                v4d (&wr)[4], v4d (&wi)[4], v4d (*s_ring)[2][64], v4d (*s_h)[2][64],
                                                const v4d *__restrict__ dbase, long duty_slot4, bool duty_is_h,
                                                int duty_plane, int nb, int ring, int sl_tb, int lane, int wv)
{
    v4d q[D];
    int dnext = duty_is_h ? 0 : sl_tb;
#define BFIR_DUTY_ADVANCE()                                                                         \
    do {                                                                                            \
        if (duty_is_h) { if (dnext < nb - 1) dnext += 1; }                                          \
        else { dnext -= 1; if (dnext < 0) dnext += ring; }                                          \
    } while (0)
#define BFIR_DUTY_LOAD() dbase[dnext * duty_slot4 + duty_plane]
#define BFIR_DUTY_STORE(s_, v_)                                                                     \
    do {                                                                                            \
        v4d *dst_ = duty_is_h ? &s_h[(s_) & 1][duty_plane][lane] : &s_ring[(-(s_)) & 15][duty_plane][lane]; \
        *dst_ = (v_);                                                                               \
    } while (0)
    if (duty_is_h) { const v4d h0 = BFIR_DUTY_LOAD(); BFIR_DUTY_STORE(0, h0); }
#pragma unroll
    for (int d = 0; d < D; d++) { BFIR_DUTY_ADVANCE(); q[(1 + d) % D] = BFIR_DUTY_LOAD(); }   // steps 1 .. D
    __syncthreads();
    for (int i0 = 0; i0 < nb; i0 += 4) {
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            const int i = i0 + ii;
            if (i < nb) {   // uniform over the workgroup
                const v4d hr = s_h[ii & 1][0][lane], hi = s_h[ii & 1][1][lane];
                if (i > 0) {
                    const int e = (4 * wv - i) & 15;             // ring entry holding X[tb + 4 wv - i]
                    wr[(4 - ii) % 4] = s_ring[e][0][lane]; wi[(4 - ii) % 4] = s_ring[e][1][lane];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int idx = (j - ii + 4) % 4;            // window slot holding X[t0 + j - i]
                    cmac4d(accr[j], acci[j], wr[idx], wi[idx], hr, hi);
                    if constexpr (DCNY) {
                        dc[j] = fma(wr[idx].x, hr.x, dc[j]);
                        ny[j] = fma(wi[idx].x, hi.x, ny[j]);
                    }
                }
                BFIR_DUTY_STORE(i + 1, q[(ii + 1) % D]);
                BFIR_DUTY_ADVANCE();
                q[(ii + 1) % D] = BFIR_DUTY_LOAD();
                __syncthreads();
            }
        }
    }
#undef BFIR_DUTY_ADVANCE
#undef BFIR_DUTY_LOAD
#undef BFIR_DUTY_STORE
}

template <int D>
__global__ __launch_bounds__(256, 2) void k_mac_lds_d(MacArgs a, int nbt, int nTQ)
{
    __shared__ __attribute__((aligned(32))) v4d s_ring[16][2][64];
    __shared__ __attribute__((aligned(32))) v4d s_h[2][2][64];
    static_assert(4 % D == 0, "prefetch depth must divide the unroll");
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / nTQ, tq = w - s * nTQ;
    const int gc = s / nbt, bt = s - gc * nbt;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = bt * 64 + lane;
    const int tb = tq * 16, t0 = tb + 4 * wv;
    const long slot4 = a.N / 4;
    const v4d *__restrict__ X = (const v4d *)((const double *)a.x + (long)gc * a.x_ch_stride) + 2 * g;
    const v4d *__restrict__ H = (const v4d *)((const double *)a.h + (long)gc * a.h_ch_stride) + 2 * g;
    const int nb = a.nblk[gc];
    const int ring = a.ring;
    const int sl_tb = (a.base_slot + tb) % ring;   // delay-line slot of block tb
    v4d accr[4], acci[4], wr[4], wi[4];
    double dc[4], ny[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        accr[j] = v4d{0, 0, 0, 0}; acci[j] = v4d{0, 0, 0, 0};
        dc[j] = 0.0; ny[j] = 0.0;
        int sj = sl_tb + 4 * wv + j; if (sj >= ring) sj -= ring;
        wr[j] = X[sj * slot4]; wi[j] = X[sj * slot4 + 1];
        s_ring[4 * wv + j][0][lane] = wr[j]; s_ring[4 * wv + j][1][lane] = wi[j];
    }
    const bool duty_is_h = wv < 2;
    const int plane = wv & 1;
    if (bt == 0)
        mac_lds_steps_d<D, true>(accr, acci, dc, ny, wr, wi, s_ring, s_h, duty_is_h ? H : X, slot4, duty_is_h, plane, nb,
                                 ring, sl_tb, lane, wv);
    else
        mac_lds_steps_d<D, false>(accr, acci, dc, ny, wr, wi, s_ring, s_h, duty_is_h ? H : X, slot4, duty_is_h, plane, nb,
                                  ring, sl_tb, lane, wv);
    double *__restrict__ Y = (double *)a.y + (long)gc * a.y_ch_stride;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = t0 + j;
        if (t < a.n_t) {
            if (g == 0) { accr[j].x = dc[j]; acci[j].x = ny[j]; }
            v4d *yo = (v4d *)(Y + (long)t * a.N) + 2 * g;
            yo[0] = accr[j]; yo[1] = acci[j];
        }
    }
}

// ---------------------------------------------------------------------------
// k_mac_lds_d2: the fp64 LDS MAC with TWO bins per lane and EIGHT blocks per wave
// ---------------------------------------------------------------------------
// k_mac_lds_d reads 128 bytes of LDS per lane and step (H_i and one delay-line spectrum, 4 bins each) for
// 64 double FMAs; eight waves per CU doing that need the whole 256 B/clk of the LDS, which is what bounds it
// (a wave64 DFMA issues every 4 cycles).  With half a group per lane (2 bins: 16 bytes of real parts + 16 of
// imaginary parts per spectrum) the same registers hold 8 output blocks instead of 4, so a step is still 64
// FMAs per lane but on 64 bytes of LDS.  The data movement is then byte for byte that of the fp32 kernel
// k_mac_lds (16-byte planes, 32-entry ring, 32-block time tiles, duty waves 0/1 fetch H re/im, 2/3 fetch
// X[tb - s] re/im); lanes 2g and 2g+1 share group g of the grouped layout.  Same fma chain per bin and the
// same partition order as every other MAC kernel: bit-identical sums.
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cmac2d(v2d &ar, v2d &ai, const v2d &xr, const v2d &xi, const v2d &hr, const v2d &hi)
{
    ar = __builtin_elementwise_fma(xr, hr, ar); ar = __builtin_elementwise_fma(-xi, hi, ar);
    ai = __builtin_elementwise_fma(xr, hi, ai); ai = __builtin_elementwise_fma(xi, hr, ai);
}

template <int D, bool DCNY>
__device__ __forceinline__ void mac_lds_steps_d2(v2d (&accr)[8], v2d (&acci)[8], double (&dc)[8], double (&ny)[8],
                                                 v2d (&wr)[8], v2d (&wi)[8], v2d (*s_ring)[2][64], v2d (*s_h)[2][64],
                                                 const v2d *__restrict__ dbase, long duty_slot, bool duty_is_h,
                                                 int nb, int ring, int sl_tb, int lane, int wv, int duty_plane)
{
    // duty: this wave fetches one plane (re or im; dbase already points at it) of H_s (waves 0, 1) or of
    // X[tb - s] (waves 2, 3) for every step s, D steps ahead of its use
    v2d q[D];
    int dnext = duty_is_h ? 0 : sl_tb;
#define BFIR_DUTY_ADVANCE()                                                                         \
    do {                                                                                            \
        if (duty_is_h) { if (dnext < nb - 1) dnext += 1; }                                          \
        else { dnext -= 1; if (dnext < 0) dnext += ring; }                                          \
    } while (0)
#define BFIR_DUTY_LOAD() dbase[dnext * duty_slot]
#define BFIR_DUTY_STORE(s_, v_)                                                                     \
    do {                                                                                            \
        v2d *dst_ = duty_is_h ? &s_h[(s_) & 1][duty_plane][lane] : &s_ring[(-(s_)) & 31][duty_plane][lane]; \
        *dst_ = (v_);                                                                               \
    } while (0)
    if (duty_is_h) { const v2d h0 = BFIR_DUTY_LOAD(); BFIR_DUTY_STORE(0, h0); }
#pragma unroll
    for (int d = 0; d < D; d++) { BFIR_DUTY_ADVANCE(); q[(1 + d) % D] = BFIR_DUTY_LOAD(); }   // steps 1 .. D
    __syncthreads();
    for (int i0 = 0; i0 < nb; i0 += 8) {
#pragma unroll
        for (int ii = 0; ii < 8; ii++) {
            const int i = i0 + ii;
            if (i < nb) {   // uniform over the workgroup
                const v2d hr = s_h[ii & 1][0][lane], hi = s_h[ii & 1][1][lane];
                if (i > 0) {
                    const int e = (8 * wv - i) & 31;             // ring entry holding X[tb + 8 wv - i]
                    wr[(8 - ii) % 8] = s_ring[e][0][lane]; wi[(8 - ii) % 8] = s_ring[e][1][lane];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int idx = (j - ii + 8) % 8;            // window slot holding X[t0 + j - i]
                    cmac2d(accr[j], acci[j], wr[idx], wi[idx], hr, hi);
                    if constexpr (DCNY) {
                        dc[j] = fma(wr[idx].x, hr.x, dc[j]);
                        ny[j] = fma(wi[idx].x, hi.x, ny[j]);
                    }
                }
                // publish the operands of step i+1, refill the queue slot with those of step i+1+D
                BFIR_DUTY_STORE(i + 1, q[(ii + 1) % D]);
                BFIR_DUTY_ADVANCE();
                q[(ii + 1) % D] = BFIR_DUTY_LOAD();
                __syncthreads();
            }
        }
    }
#undef BFIR_DUTY_ADVANCE
#undef BFIR_DUTY_LOAD
#undef BFIR_DUTY_STORE
}

template <int D>
__global__ __launch_bounds__(256, 2) void k_mac_lds_d2(MacArgs a, int nbt, int nTQ)
{
    __shared__ __attribute__((aligned(16))) v2d s_ring[32][2][64];
    __shared__ __attribute__((aligned(16))) v2d s_h[2][2][64];
    static_assert(8 % D == 0, "prefetch depth must divide the unroll");
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / nTQ, tq = w - s * nTQ;
    const int gc = s / nbt, bt = s - gc * nbt;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = bt * 32 + (lane >> 1), hh = lane & 1;              // group of 4 bins, which half of it
    const int tb = tq * 32, t0 = tb + 8 * wv;
    const long slot = a.N / 2;                                        // v2d elements per spectrum
    // grouped layout: group g = doubles 8g .. 8g+7 = re0 re1 | re2 re3 | im0 im1 | im2 im3 as four v2d
    const v2d *__restrict__ X = (const v2d *)((const double *)a.x + (long)gc * a.x_ch_stride) + 4 * g + hh;
    const v2d *__restrict__ H = (const v2d *)((const double *)a.h + (long)gc * a.h_ch_stride) + 4 * g + hh;
    const int nb = a.nblk[gc];
    const int ring = a.ring;
    const int sl_tb = (a.base_slot + tb) % ring;   // delay-line slot of block tb
    v2d accr[8], acci[8], wr[8], wi[8];
    double dc[8], ny[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        accr[j] = v2d{0, 0}; acci[j] = v2d{0, 0};
        dc[j] = 0.0; ny[j] = 0.0;
        int sj = sl_tb + 8 * wv + j; if (sj >= ring) sj -= ring;
        wr[j] = X[sj * slot]; wi[j] = X[sj * slot + 2];
        s_ring[8 * wv + j][0][lane] = wr[j]; s_ring[8 * wv + j][1][lane] = wi[j];
    }
    const bool duty_is_h = wv < 2;
    const int plane = wv & 1;
    const v2d *dbase = (duty_is_h ? H : X) + 2 * plane;
    if (bt == 0)   // lane 0 of this tile holds bin 0: DC in the real slot, Nyquist in the imaginary one
        mac_lds_steps_d2<D, true>(accr, acci, dc, ny, wr, wi, s_ring, s_h, dbase, slot, duty_is_h, nb, ring, sl_tb, lane, wv, plane);
    else
        mac_lds_steps_d2<D, false>(accr, acci, dc, ny, wr, wi, s_ring, s_h, dbase, slot, duty_is_h, nb, ring, sl_tb, lane, wv, plane);
    double *__restrict__ Y = (double *)a.y + (long)gc * a.y_ch_stride;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int t = t0 + j;
        if (t < a.n_t) {
            if (g == 0 && hh == 0) { accr[j].x = dc[j]; acci[j].x = ny[j]; }
            v2d *yo = (v2d *)(Y + (long)t * a.N) + 4 * g + hh;
            yo[0] = accr[j]; yo[2] = acci[j];
        }
    }
}

// The same kernel with G partition steps per barrier.  k_mac_lds_d2 meets at a workgroup barrier after every
// partition (64 double FMAs per lane, ~0.3 us): at two waves per SIMD the barrier wait is a third of its time.
// Here the four duty waves publish the operands of G steps at a time (H_s into a 2G-deep buffer, X[tb - s] into
// the ring: an entry is overwritten 8 steps after its last reader, so up to 4 steps ahead is safe) and the
// workgroup meets once per G steps.  Same sums in the same order: bit-identical results.
template <int G, int PF, bool DCNY>
__device__ __forceinline__ void mac_lds_steps_d2g(v2d (&accr)[8], v2d (&acci)[8], double (&dc)[8], double (&ny)[8],
                                                  v2d (&wr)[8], v2d (&wi)[8], v2d (*s_ring)[2][64], v2d (*s_h)[2][64],
                                                  const v2d *__restrict__ dbase, long duty_slot, bool duty_is_h,
                                                  int nb, int ring, int sl_tb, int lane, int wv, int duty_plane)
{
    static_assert(G == 1 || G == 2 || G == 4, "G divides the window period 8; ring entries live 8 steps");
    constexpr int NH = 2 * G;
    // duty: this wave fetches one plane (re or im; dbase already points at it) of H_s (waves 0, 1) or of
    // X[tb - s] (waves 2, 3) for every step s; the loads run PF groups of G steps ahead of the group being
    // computed (PF = 2 or 4 measured no faster than 1: global latency is covered; what remains exposed is the LDS
    // read latency right after each barrier)
    static_assert((8 / G) % PF == 0, "queue slot of a group is a compile-time constant");
    v2d q[PF][G];
    int dnext = duty_is_h ? 0 : sl_tb;             // operand index of the step whose load comes next
    auto duty_load = [&]() -> v2d {
        const v2d v = dbase[dnext * duty_slot];
        if (duty_is_h) { if (dnext < nb - 1) dnext += 1; }
        else { dnext -= 1; if (dnext < 0) dnext += ring; }
        return v;
    };
    auto duty_store = [&](int s_, v2d v_) {
        // X[tb - 0] is the tile's own first block: already in the ring
        if (duty_is_h) s_h[s_ % NH][duty_plane][lane] = v_;
        else if (s_ > 0) s_ring[(-s_) & 31][duty_plane][lane] = v_;
    };
    // the ring entries these first stores reuse (31, 30, ..) are being filled with the tile's own blocks by
    // wave 3 right now: every wave's fill first
    __syncthreads();
#pragma unroll
    for (int u = 0; u < G; u++) { const v2d v = duty_load(); duty_store(u, v); }     // steps 0 .. G-1
#pragma unroll
    for (int f = 0; f < PF; f++)
#pragma unroll
        for (int u = 0; u < G; u++) q[f][u] = duty_load();                            // steps G .. (PF+1) G - 1
    __syncthreads();
    for (int i0 = 0; i0 < nb; i0 += 8) {
#pragma unroll
        for (int gi = 0; gi < 8 / G; gi++) {
            if (i0 + gi * G < nb) {   // uniform over the workgroup
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const int ii = gi * G + u, i = i0 + ii;
                    if (i < nb) {     // uniform
                        const v2d hr = s_h[ii % NH][0][lane], hi = s_h[ii % NH][1][lane];
                        if (i > 0) {
                            const int e = (8 * wv - i) & 31;             // ring entry holding X[tb + 8 wv - i]
                            wr[(8 - ii) % 8] = s_ring[e][0][lane]; wi[(8 - ii) % 8] = s_ring[e][1][lane];
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int idx = (j - ii + 8) % 8;            // window slot holding X[t0 + j - i]
                            cmac2d(accr[j], acci[j], wr[idx], wi[idx], hr, hi);
                            if constexpr (DCNY) {
                                dc[j] = fma(wr[idx].x, hr.x, dc[j]);
                                ny[j] = fma(wi[idx].x, hi.x, ny[j]);
                            }
                        }
                    }
                }
                // publish the operands of the next G steps, refill the queue with those of the G after them
#pragma unroll
                for (int u = 0; u < G; u++) {
                    duty_store(i0 + gi * G + G + u, q[gi % PF][u]);
                    q[gi % PF][u] = duty_load();
                }
                __syncthreads();
            }
        }
    }
}

template <int G, int PF>
__global__ __launch_bounds__(256, 2) void k_mac_lds_d2g(MacArgs a, int nbt, int nTQ)
{
    __shared__ __attribute__((aligned(16))) v2d s_ring[32][2][64];
    __shared__ __attribute__((aligned(16))) v2d s_h[2 * G][2][64];
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / nTQ, tq = w - s * nTQ;
    const int gc = s / nbt, bt = s - gc * nbt;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = bt * 32 + (lane >> 1), hh = lane & 1;              // group of 4 bins, which half of it
    const int tb = tq * 32, t0 = tb + 8 * wv;
    const long slot = a.N / 2;                                        // v2d elements per spectrum
    // grouped layout: group g = doubles 8g .. 8g+7 = re0 re1 | re2 re3 | im0 im1 | im2 im3 as four v2d
    const v2d *__restrict__ X = (const v2d *)((const double *)a.x + (long)gc * a.x_ch_stride) + 4 * g + hh;
    const v2d *__restrict__ H = (const v2d *)((const double *)a.h + (long)gc * a.h_ch_stride) + 4 * g + hh;
    const int nb = a.nblk[gc];
    const int ring = a.ring;
    const int sl_tb = (a.base_slot + tb) % ring;   // delay-line slot of block tb
    v2d accr[8], acci[8], wr[8], wi[8];
    double dc[8], ny[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        accr[j] = v2d{0, 0}; acci[j] = v2d{0, 0};
        dc[j] = 0.0; ny[j] = 0.0;
        int sj = sl_tb + 8 * wv + j; if (sj >= ring) sj -= ring;
        wr[j] = X[sj * slot]; wi[j] = X[sj * slot + 2];
        s_ring[8 * wv + j][0][lane] = wr[j]; s_ring[8 * wv + j][1][lane] = wi[j];
    }
    const bool duty_is_h = wv < 2;
    const int plane = wv & 1;
    const v2d *dbase = (duty_is_h ? H : X) + 2 * plane;
    if (bt == 0)   // lane 0 of this tile holds bin 0: DC in the real slot, Nyquist in the imaginary one
        mac_lds_steps_d2g<G, PF, true>(accr, acci, dc, ny, wr, wi, s_ring, s_h, dbase, slot, duty_is_h, nb, ring, sl_tb, lane, wv, plane);
    else
        mac_lds_steps_d2g<G, PF, false>(accr, acci, dc, ny, wr, wi, s_ring, s_h, dbase, slot, duty_is_h, nb, ring, sl_tb, lane, wv, plane);
    double *__restrict__ Y = (double *)a.y + (long)gc * a.y_ch_stride;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int t = t0 + j;
        if (t < a.n_t) {
            if (g == 0 && hh == 0) { accr[j].x = dc[j]; acci[j].x = ny[j]; }
            v2d *yo = (v2d *)(Y + (long)t * a.N) + 4 * g + hh;
            yo[0] = accr[j]; yo[2] = acci[j];
        }
    }
}

template <int D> static void launch_mac_lds_d2(const MacArgs &a, hipStream_t s)
{
    const int nbt = a.N / 8 / 32;              // bin tiles of 32 groups (two lanes per group)
    const int nTQ = (a.n_t + 31) / 32;         // time tiles of 32 blocks
    hipLaunchKernelGGL((k_mac_lds_d2<D>), dim3(nTQ * nbt * a.n_ch), dim3(256), 0, s, a, nbt, nTQ);
}

template <int G, int PF> static void launch_mac_lds_d2g(const MacArgs &a, hipStream_t s)
{
    const int nbt = a.N / 8 / 32;
    const int nTQ = (a.n_t + 31) / 32;
    hipLaunchKernelGGL((k_mac_lds_d2g<G, PF>), dim3(nTQ * nbt * a.n_ch), dim3(256), 0, s, a, nbt, nTQ);
}

template <int D> static void launch_mac_lds_d(const MacArgs &a, hipStream_t s)
{
    const int nbt = a.N / 8 / 64;              // bin tiles of 64 groups
    const int nTQ = (a.n_t + 15) / 16;         // time tiles of 16 blocks
    hipLaunchKernelGGL((k_mac_lds_d<D>), dim3(nTQ * nbt * a.n_ch), dim3(256), 0, s, a, nbt, nTQ);
}

// ---------------------------------------------------------------------------
// k_mac_stream: the time-streaming form of the same sums (fp32; B > PB in batches of PB partitions)
// ---------------------------------------------------------------------------
// One lane owns ONE bin of one channel for a range of output blocks and keeps
// all PB filter partitions of that bin (2 PB registers) plus PB rotating
// accumulators (2 PB registers) resident.  It walks the delay line backwards in
// time: each spectrum value x[t] is loaded exactly once (one 8-byte load per lane,
// 512 contiguous bytes per wave: the engine keeps X, H and Y as (re, im) pairs
// for this kernel -- MacArgs.interleaved) and feeds the PB outputs it contributes to,
//     y[t + p] += x[t] * h[p],   p = 0 .. PB-1,
// so output y[t'] receives p = 0 first and p = PB-1 last -- the reference's
// partition order with the same fma chain as k_mac, hence bit-identical sums --
// and is stored the moment its last partition has been added.  128 FMAs per
// 8 bytes loaded, no LDS, no barrier, loads D blocks ahead in a register queue.
// A range of R = ngrp*PB blocks costs exactly R*PB complex MACs: the newest
// group only feeds outputs inside the range (MODE 0, a triangle), the PB-1
// blocks before the range only feed its oldest outputs (MODE 2, the
// complementary triangle), the groups in between are full (MODE 1).
// Partitions p >= nblk[gc] hold h = 0 (an exact no-op on the sums).
// Bin 0 (DC | Nyquist, two independent real sums) is left to the first blocks
// of the grid (one thread per channel and output block); the streaming lanes
// never store it.

// ACC (partition batches after the first, B > PB): the sums continue from the partial results the
// previous batch left in Y; those are fetched D blocks ahead like the spectra (yq, ty_next).
__device__ __forceinline__ float2 mac_ld_x(const float2 *p)
{
#if BFIR_NT_X & 2
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v v = __builtin_nontemporal_load((const f2v *)p);
    float2 r; r.x = v.x; r.y = v.y; return r;
#else
    return *p;
#endif
}
__device__ __forceinline__ void mac_st_y(float2 *p, float2 v)
{
#if BFIR_NT_Y & 1
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, (f2v *)p);
#else
    *p = v;
#endif
}

template <int PB, int D, int MODE, bool ACC>
__device__ __forceinline__ void mac_stream_group(float (&ar)[PB], float (&ai)[PB], const float (&hr)[PB],
                                                 const float (&hi)[PB], float2 (&q)[D], float2 (&yq)[ACC ? D : 1],
                                                 const float2 *__restrict__ Xc, float2 *__restrict__ Yc, unsigned k,
                                                 int N2, int ring, int &sq, int &ty_next, int tg, int n_t,
                                                 bool store_lane, const MacArgs &a)
{
    static_assert(PB % D == 0, "queue depth must divide the group");
    constexpr int NU = MODE == 2 ? PB - 1 : PB;          // the block PB before the range feeds nothing
    static_for<0, NU>([&](auto U) {                      // spectrum of block t = tg + PB-1-u
        constexpr int u = decltype(U)::value;
        const float xr = q[u % D].x, xi = q[u % D].y;
        q[u % D] = mac_ld_x(Xc + (long)sq * N2 + k);     // uniform base + 32-bit lane offset, 8 bytes per lane
        sq -= 1; if (sq < 0) sq += ring;
        const float nxi = -xi;
        constexpr int plo = MODE == 2 ? u + 1 : 0, phi = MODE == 0 ? u : PB - 1;
        float y0r = 0.f, y0i = 0.f;                      // what the sums of y[t] start from
        if constexpr (ACC && plo == 0) {
            y0r = yq[u % D].x; y0i = yq[u % D].y;
            float2 v; v.x = 0.f; v.y = 0.f;
            if (ty_next >= 0 && ty_next < n_t) v = (Yc + (long)ty_next * N2)[k];
            yq[u % D] = v;
            ty_next -= 1;
        }
        static_for<plo, phi + 1>([&](auto P) {
            constexpr int p = decltype(P)::value;
            constexpr int sl = (PB - 1 - u + p) % PB;    // accumulator of y[t + p]
            if constexpr (p == 0) {
                ar[sl] = fmaf(xr, hr[0], y0r); ar[sl] = fmaf(nxi, hi[0], ar[sl]);
                ai[sl] = fmaf(xr, hi[0], y0i); ai[sl] = fmaf(xi, hr[0], ai[sl]);
            } else {
                ar[sl] = fmaf(xr, hr[p], ar[sl]); ar[sl] = fmaf(nxi, hi[p], ar[sl]);
                ai[sl] = fmaf(xr, hi[p], ai[sl]); ai[sl] = fmaf(xi, hr[p], ai[sl]);
            }
            // pin the sums here: otherwise the compiler sinks every chain down to its (conditional)
            // store, which keeps a whole group of spectra live and serialises each chain
            asm volatile("" : "+v"(ar[sl]), "+v"(ai[sl]));
        });
        if constexpr (phi == PB - 1) {                   // y[t + PB-1] is complete
            constexpr int sl = (2 * PB - 2 - u) % PB;
            const int ty = tg + 2 * PB - 2 - u;
            if (ty < n_t && store_lane) {
                float2 v; v.x = ar[sl]; v.y = ai[sl];
                mac_st_y(Yc + (long)BFIR_YSLOT(a, ty) * N2 + k, v);
            }
        }
    });
}

template <int PB, int D, bool ACC>
__global__ __launch_bounds__(256, ACC ? 2 : 3) void k_mac_stream(MacArgs a, int ncol, int nR, int ngrp, int n_dc, int p0)
{
    const int N = a.N, ring = a.ring;
    if ((int)blockIdx.x < n_dc) {
        // DC / Nyquist: the two floats of bin 0 (pairs layout), two independent real sums in partition order.
        // One workgroup = one channel x 256 output blocks: the 256+PB-1 delay-line values and the
        // PB filter values go through LDS in one round of (scattered, but concurrent) loads, then
        // every thread runs its own chain out of LDS.
        __shared__ float s_x[2][256 + PB], s_hh[2][PB];
        const int ndt = (a.n_t + 255) / 256;
        const int gc = blockIdx.x / ndt, t0 = (blockIdx.x - gc * ndt) * 256;
        const float *__restrict__ X = (const float *)a.x + (long)gc * a.x_ch_stride;
        const float *__restrict__ H = (const float *)a.h + (long)gc * a.h_ch_stride;
        const int nb = a.nblk[gc];
        const int t = t0 + threadIdx.x;
        float dc = 0.f, ny = 0.f;
        for (int pb0 = 0; pb0 < nb; pb0 += PB) {                     // all partitions, PB at a time, in order
            if (pb0 > 0) __syncthreads();
            for (int i = threadIdx.x; i < 256 + PB - 1; i += 256) {  // s_x[.][i] = block t0 - pb0 - (PB-1) + i
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
            float *yo = (float *)a.y + (long)gc * a.y_ch_stride + (long)t * N;
            yo[0] = dc; yo[1] = ny;
        }
        return;
    }
    // XCD-aware bijective remap (see k_mac): (channel, bin column) major, time range minor
    const int W = gridDim.x - n_dc, b = blockIdx.x - n_dc, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / nR, r = w - s * nR;
    const int gc = s / ncol, col = s - gc * ncol;
    const unsigned k = col * 256 + threadIdx.x;          // bin: (re, im) pair k of every spectrum
    const int N2 = N / 2;
    const float2 *__restrict__ Xc = (const float2 *)((const float *)a.x + (long)gc * a.x_ch_stride);
    const float2 *__restrict__ Hc = (const float2 *)((const float *)a.h + (long)gc * a.h_ch_stride);
    float2 *__restrict__ Yc = (float2 *)((float *)a.y + (long)gc * a.y_ch_stride);
    const int nb = a.nblk[gc];
    // this range: blocks ta .. ta+R-1; the last range of a launch only runs the groups it has blocks for
    const int ta = r * ngrp * PB;
    const int my_grp = min(ngrp, (a.n_t - ta + PB - 1) / PB);
    const int R = my_grp * PB;
    BFIR_STAMP(2, 0);

    float2 q[D];
    // newest block of the range first; batch p0 pairs output t with spectrum t - p0 - p
    int sq = ((a.base_slot + ta + R - 1 - p0) % ring + ring) % ring;
#pragma unroll
    for (int d = 0; d < D; d++) {
        q[d] = mac_ld_x(Xc + (long)sq * N2 + k);
        sq -= 1; if (sq < 0) sq += ring;
    }
    float hr[PB], hi[PB];
#pragma unroll
    for (int p = 0; p < PB; p++) {
        if (p0 + p < nb) { const float2 h = (Hc + (long)(p0 + p) * N2)[k]; hr[p] = h.x; hi[p] = h.y; }
        else { hr[p] = 0.f; hi[p] = 0.f; }
    }
    float2 yq[ACC ? D : 1];
    int ty_next = ta + R - 1;
    if constexpr (ACC) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            float2 v; v.x = 0.f; v.y = 0.f;
            if (ty_next >= 0 && ty_next < a.n_t) v = (Yc + (long)ty_next * N2)[k];
            yq[d] = v;
            ty_next -= 1;
        }
    }
    float ar[PB], ai[PB];
#pragma unroll
    for (int p = 0; p < PB; p++) { ar[p] = 0.f; ai[p] = 0.f; }
    const bool store_lane = k != 0;
    int tg = ta + R - PB;
    BFIR_STAMP(2, 1);
    mac_stream_group<PB, D, 0, ACC>(ar, ai, hr, hi, q, yq, Xc, Yc, k, N2, ring, sq, ty_next, tg, a.n_t, store_lane, a);
    BFIR_STAMP(2, 2);
    for (int g = my_grp - 2; g >= 0; g--) {
        tg -= PB;
        mac_stream_group<PB, D, 1, ACC>(ar, ai, hr, hi, q, yq, Xc, Yc, k, N2, ring, sq, ty_next, tg, a.n_t, store_lane, a);
    }
    BFIR_STAMP(2, 3);
    tg -= PB;
    mac_stream_group<PB, D, 2, ACC>(ar, ai, hr, hi, q, yq, Xc, Yc, k, N2, ring, sq, ty_next, tg, a.n_t, store_lane, a);
    BFIR_STAMP(2, 4);
}

// ngrp: groups of PB blocks per wave (BFIR_MAC_RANGE overrides, in blocks)
template <int PB, int D> static void launch_mac_stream(const MacArgs &a_, hipStream_t s)
{
    MacArgs a = a_;
#ifdef BFIR_EXPERIMENT_ALIAS
    if (const int xa = bfir_alias_env("BFIR_X_ALIAS")) { a.ring = xa; a.base_slot %= xa; }
    if (const int ya = bfir_alias_env("BFIR_Y_ALIAS")) a.y_alias = ya;
#endif
    // read per launch (a getenv is nanoseconds next to a launch) so tests can switch it in-process
    const char *re_ = getenv("BFIR_MAC_RANGE");
    const int range_env = re_ ? atoi(re_) : 0;
    const int ncol = a.N / 2 / 256;                      // workgroup columns of 256 bins
    int ngrp;
    if (range_env > 0) ngrp = std::max(1, range_env / PB);
    else {
        // about 2048 waves (two per SIMD, one round: measured best, profiles/r01_mac_ranges.txt),
        // never shorter than one group
        const long cols = (long)ncol * a.n_ch;
        const long want = std::max<long>(1, 512 / cols);
        ngrp = std::max(1, (int)((a.n_t + want * PB - 1) / (want * PB)));
    }
    const int R = ngrp * PB, nR = (a.n_t + R - 1) / R;
    const int n_dc = a.n_ch * ((a.n_t + 255) / 256);
    hipLaunchKernelGGL((k_mac_stream<PB, D, false>), dim3(n_dc + nR * ncol * a.n_ch), dim3(256), 0, s, a, ncol, nR, ngrp,
                       n_dc, 0);
    // more partitions than fit the registers: further batches of PB continue the sums left in Y
    // (same stream, so batch b sees batch b-1's stores; DC/Nyquist were finished by the first launch)
    for (int p0 = PB; p0 < a.B; p0 += PB)
        hipLaunchKernelGGL((k_mac_stream<PB, D, true>), dim3(nR * ncol * a.n_ch), dim3(256), 0, s, a, ncol, nR, ngrp, 0,
                           p0);
}

template <int D, bool PAIRS = false> static void launch_mac_lds(const MacArgs &a, hipStream_t s)
{
    const int nbt = a.N / 8 / 64;              // bin tiles of 64 groups
    const int nTQ = (a.n_t + 31) / 32;         // time tiles of 32 blocks
    hipLaunchKernelGGL((k_mac_lds<D, PAIRS>), dim3(nTQ * nbt * a.n_ch), dim3(256), 0, s, a, nbt, nTQ);
}

template <typename T, int TT, int WPE, int D> static void launch_mac_t(const MacArgs &a, hipStream_t s)
{
    const int G = a.N / 8;
    const int threads = G < 256 ? G : 256;
    const int nbt = (G + threads - 1) / threads;
    const int nTT = (a.n_t + TT - 1) / TT;
    const int W = nTT * nbt * a.n_ch;
    hipLaunchKernelGGL((k_mac<T, TT, WPE, D>), dim3(W), dim3(threads), 0, s, a, nbt, nTT, G);
}

// BFIR_MAC_VARIANT (tuning aid): pick another (time tile, waves/SIMD, prefetch depth)
// instantiation of the fp32 MAC kernel.  All variants give bit-identical results.
static int mac_variant()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("BFIR_MAC_VARIANT");
        v = e ? atoi(e) : 0;
        if (v < 0) v = 0;
    }
    return v;
}

// ---------------------------------------------------------------------------
// k_mac_tstream: partition-streaming MAC out of registers (fp64; grouped layout)
// ---------------------------------------------------------------------------
// The transpose of k_mac_stream: a lane owns ONE bin and keeps the sums of TT consecutive output blocks (2 TT
// values) plus a window of TT delay-line values (2 TT) in registers, and walks the PARTITIONS in order: step p
// loads h[p] and the one new window value x[t0 - p] (8-byte loads, D steps ahead in a register queue) and feeds
// all TT outputs, y[t0 + j] += x[t0 + j - p] h[p].  4 TT fused multiply-adds per 32 bytes loaded, no LDS, no
// barrier -- the LDS-tiled fp64 kernels (k_mac_lds_d2g) spend half their time behind barriers and LDS latency
// at the two waves per SIMD their 220 registers allow.  Price: the window is re-read by every time tile
// ((TT + B - 1) / TT spectra per output block instead of (32 + B - 1) / 32) and so is h -- out of L2: the XCD map
// keeps the time tiles of one (channel, 256-bin column) on one XCD.  Output y[t] receives p = 0 first, B-1
// last, with the fma chain of every other MAC kernel: bit-identical sums.
// Measured (profiles/r02_fp64_mac.txt): exactly as fast as the LDS-tiled kernel (0.151 vs 0.158 ms per 4096 blocks at
// the plug-in's shape, 0.60 vs 0.57 at cfg5's) -- a wave issues DFMAs 31 % of its life and waits for its queue
// 27 %; two waves per SIMD is all 208 registers allow, and a one-wave-per-SIMD build with 24 outputs and a
// 12-deep queue in AGPRs is slower (0.68), 12 outputs per lane at three waves per SIMD the same (0.154-0.163 / 0.63).
// Kept as BFIR_MAC64_VARIANT=12, not the default.
template <typename T, bool ILV, int TT, int D, bool DCNY>
__device__ __forceinline__ void mac_tstream_body(const MacArgs &a, int gc, int k, int t0)
{
    using V2 = typename Vec2<T>::type;
    const int N = a.N, ring = a.ring;
    const T *__restrict__ X = (const T *)a.x + (long)gc * a.x_ch_stride;
    const T *__restrict__ H = (const T *)a.h + (long)gc * a.h_ch_stride;
    T *__restrict__ Y = (T *)a.y + (long)gc * a.y_ch_stride;
    const int nb = a.nblk[gc];
    // (re, im) pairs (ILV: one load per value) or the reference's groups: 4 re then 4 im per 4 bins (two loads)
    const int ore = ILV ? 2 * k : 8 * (k >> 2) + (k & 3), oim = ILV ? ore + 1 : ore + 4;
    auto ld = [&](const T *base, T &re, T &im) {
        if constexpr (ILV) { const V2 v = *(const V2 *)(base + ore); re = v.x; im = v.y; }
        else { re = base[ore]; im = base[oim]; }
    };
    const int sl0 = (a.base_slot + t0) % ring;                    // delay-line slot of block t0
    T wr[TT], wi[TT], ar[TT], ai[TT];
#pragma unroll
    for (int j = 0; j < TT; j++) {
        int sj = sl0 + j; if (sj >= ring) sj -= ring;
        ld(X + (long)sj * N, wr[j], wi[j]);
        ar[j] = (T)0; ai[j] = (T)0;
    }
    // operand queue: step p needs h[p] and x[t0 - p] (p = 0: x[t0], already in the window; loaded anyway)
    T qhr[D], qhi[D], qxr[D], qxi[D];
    int ph = 0, sx = sl0;
    auto fetch = [&](int d) {
        const int pc = ph < nb ? ph : nb - 1;                     // clamped: in range, never used
        ld(H + (long)pc * N, qhr[d], qhi[d]);
        ld(X + (long)sx * N, qxr[d], qxi[d]);
        ph += 1; sx -= 1; if (sx < 0) sx += ring;
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d);
    for (int p0 = 0; p0 < nb; p0 += TT) {
        static_for<0, TT>([&](auto U_) {
            constexpr int u = decltype(U_)::value;                // = p mod TT
            if (p0 + u < nb) {                                    // uniform over the grid's channel
                const T hr = qhr[u % D], hi = qhi[u % D], xr = qxr[u % D], xi = qxi[u % D];
                fetch(u % D);
                if (p0 + u > 0) { wr[(TT - u) % TT] = xr; wi[(TT - u) % TT] = xi; }   // block t0 - p takes the slot of t0 + TT - p
                static_for<0, TT>([&](auto J_) {
                    constexpr int j = decltype(J_)::value, idx = (j - u + TT) % TT;   // slot holding x[t0 + j - p]
                    if constexpr (DCNY) {
                        if (k == 0) {                             // bin 0: DC | Nyquist, two independent real sums
                            ar[j] = fma(wr[idx], hr, ar[j]); ai[j] = fma(wi[idx], hi, ai[j]);
                        } else {
                            ar[j] = fma(wr[idx], hr, ar[j]); ar[j] = fma(-wi[idx], hi, ar[j]);
                            ai[j] = fma(wr[idx], hi, ai[j]); ai[j] = fma(wi[idx], hr, ai[j]);
                        }
                    } else {
                        ar[j] = fma(wr[idx], hr, ar[j]); ar[j] = fma(-wi[idx], hi, ar[j]);
                        ai[j] = fma(wr[idx], hi, ai[j]); ai[j] = fma(wi[idx], hr, ai[j]);
                    }
                });
            }
        });
    }
#pragma unroll
    for (int j = 0; j < TT; j++) {
        const int t = t0 + j;
        if (t < a.n_t) {
            if constexpr (ILV) { V2 v; v.x = ar[j]; v.y = ai[j]; *(V2 *)(Y + (long)t * N + ore) = v; }
            else { Y[(long)t * N + ore] = ar[j]; Y[(long)t * N + oim] = ai[j]; }
        }
    }
}

template <typename T, bool ILV, int TT, int D, int WPS>
__global__ __launch_bounds__(256, WPS) void k_mac_tstream(MacArgs a, int ncol, int ntile)
{
    static_assert(TT % D == 0, "the queue slot of a step is its index mod D in every round");
    // XCD-aware bijective remap: (channel, bin column) major, time tile minor
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    const int w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    const int s = w / ntile, tile = w - s * ntile;
    const int gc = s / ncol, col = s - gc * ncol;
    const int k = col * 256 + (int)threadIdx.x;
    if (col == 0) mac_tstream_body<T, ILV, TT, D, true>(a, gc, k, tile * TT);
    else mac_tstream_body<T, ILV, TT, D, false>(a, gc, k, tile * TT);
}

template <typename T, bool ILV, int TT, int D, int WPS> static void launch_mac_tstream(const MacArgs &a, hipStream_t s)
{
    const int ncol = a.N / 2 / 256, ntile = (a.n_t + TT - 1) / TT;     // N >= 512
    hipLaunchKernelGGL((k_mac_tstream<T, ILV, TT, D, WPS>), dim3(ncol * ntile * a.n_ch), dim3(256), 0, s, a, ncol, ntile);
}

// ---------------------------------------------------------------------------
// k_mac_small: the sums of a handful of output blocks (the one-block-per-call latency path)
// ---------------------------------------------------------------------------
// One lane owns ONE bin of one output block and runs the reference's chain over the partitions in order
// (same fma sequence as every other MAC kernel: bit-identical sums), U partitions' operands in flight at a
// time.  The throughput kernels are built around reuse over many blocks; called for one block they either
// walk a whole group of history twice (k_mac_stream: 63 spectra for 32 products) or run on two workgroups
// (k_mac<double>: 30 us for the plug-in's shape, half of the whole call).  Here one block of the plug-in's
// shape is 8 workgroups of independent loads.
template <typename T, bool ILV, int U>
__global__ __launch_bounds__(256) void k_mac_small(MacArgs a)
{
    const int N = a.N, N2 = N / 2, ring = a.ring;
    const int k = blockIdx.x * 256 + threadIdx.x;                 // bin; bin 0 carries DC | Nyquist
    const int t = blockIdx.y, gc = blockIdx.z;
    if (k >= N2) return;
    const T *__restrict__ X = (const T *)a.x + (long)gc * a.x_ch_stride;
    const T *__restrict__ H = (const T *)a.h + (long)gc * a.h_ch_stride;
    T *__restrict__ Y = (T *)a.y + (long)gc * a.y_ch_stride + (long)t * N;
    const int nb = a.nblk[gc];
    // (re, im) pairs: 2k, 2k+1.  The reference's groups: 4 re then 4 im per 4 bins.
    const int ore = ILV ? 2 * k : 8 * (k >> 2) + (k & 3), oim = ILV ? ore + 1 : ore + 4;
    int sl = (a.base_slot + t) % ring;                            // delay-line slot of X[t - p], p = 0
    T ar = (T)0, ai = (T)0;
    for (int p0 = 0; p0 < nb; p0 += U) {
        T xr[U], xi[U], hr[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int p = p0 + u < nb ? p0 + u : nb - 1;          // clamped: in range, never used
            const T *xs = X + (long)sl * N, *hs = H + (long)p * N;
            if constexpr (ILV) {
                using V2 = typename Vec2<T>::type;
                const V2 xv = *(const V2 *)(xs + ore), hv = *(const V2 *)(hs + ore);
                xr[u] = xv.x; xi[u] = xv.y; hr[u] = hv.x; hi[u] = hv.y;
            } else {
                xr[u] = xs[ore]; xi[u] = xs[oim]; hr[u] = hs[ore]; hi[u] = hs[oim];
            }
            sl -= 1; if (sl < 0) sl += ring;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (p0 + u < nb) {
                if (k == 0) {                                     // two independent real sums
                    ar = fma(xr[u], hr[u], ar); ai = fma(xi[u], hi[u], ai);
                } else {
                    ar = fma(xr[u], hr[u], ar); ar = fma(-xi[u], hi[u], ar);
                    ai = fma(xr[u], hi[u], ai); ai = fma(xi[u], hr[u], ai);
                }
            }
        }
    }
    Y[ore] = ar; Y[oim] = ai;
}

constexpr int BFIR_MAC_SMALL_MAX = 4;      // output blocks per launch up to which k_mac_small runs (engine.hip: kSmallRun)
static void launch_mac_small(const MacArgs &a, hipStream_t s)
{
    const dim3 grid((a.N / 2 + 255) / 256, a.n_t, a.n_ch), block(256);
    if (a.realsize == 4) {
        if (a.interleaved) hipLaunchKernelGGL((k_mac_small<float, true, 16>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_mac_small<float, false, 16>), grid, block, 0, s, a);
    } else {
        if (a.interleaved) hipLaunchKernelGGL((k_mac_small<double, true, 16>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_mac_small<double, false, 16>), grid, block, 0, s, a);
    }
}

void launch_mac(const MacArgs &a, hipStream_t s)
{
    if (a.n_t <= 0 || a.n_ch <= 0) return;
    const int tt = a.n_t;
    if (tt <= BFIR_MAC_SMALL_MAX && !getenv("BFIR_NO_MAC_SMALL")) { launch_mac_small(a, s); return; }   // env: A/B and tests
    {   // The forward-walking systolic kernel (mac_sys.hip: up to 256 partitions; fp32 on the pairs layout, fp64 on either):
        // the default for fp64 (cfg5 +14 %, its MAC -27 % against the LDS-tiled kernel) and for fp32 with 33 to
        // 64 partitions (+1..5 % against k_mac_lds; with eight lanes per bin, 65 to 128 partitions, it is 10 % SLOWER:
        // 44.7 against 49.6 Gsamples/s at 8 channels x 128 partitions of 1024); with up to 32 partitions k_mac_stream stays
        // ahead (0.49 against 0.59 ms per 4096 headline blocks: two lanes per bin double the vector-memory instructions
        // per FMA, profiles/r03_mac_sys.txt) -- except at 17 to 24 partitions, where it pays for a register batch of 32 and two
        // stages of twelve do not (MAC 0.49 -> 0.45 ms, 116.2 -> 118.3 Gsamples/s at the headline's shape with 24 partitions).
        // BFIR_MAC_SYS=1 / 0 forces / forbids it; read per launch (tests switch it).
        const char *ms = getenv("BFIR_MAC_SYS");
        const bool want = ms ? atoi(ms) != 0 : (a.realsize == 8 || (a.B > 32 && a.B <= 64) || (a.B > 16 && a.B <= 24)) && !getenv("BFIR_MAC64_VARIANT") && mac_variant() == 0;
        if (want && mac_sys_supported(a) && !getenv("BFIR_MAC_BATCHED")) { launch_mac_sys(a, s); return; }
        // an fp64 engine on (re, im) pairs (engine.hip picks that layout only where this kernel serves it, and looks at the
        // same switches when it does): no other fp64 MAC kernel reads pairs
        if (a.realsize == 8 && a.interleaved) {
            if (mac_sys_supported(a)) launch_mac_sys(a, s);
            else fprintf(stderr, "bfir: fp64 engine on the pairs layout outside the systolic MAC's range (B = %d): launch skipped\n", a.B);
            return;
        }
    }
    if (a.realsize == 4) {
        const int v = mac_variant();
        // time-streaming kernel: PB partitions of a bin in registers per batch, whole 256-bin columns
        const int pb = a.B <= 4 ? 4 : a.B <= 8 ? 8 : a.B <= 16 ? 16 : 32;
        const bool batched_only = getenv("BFIR_MAC_BATCHED") != nullptr;   // tuning aid / test hook, read per launch
        // (k_mac_tstream on the pair layout -- 24 or 32 outputs per lane -- was measured here: +1..5 % over the LDS kernel,
        // profiles/r02_fp64_mac.txt; not worth a minute of build time per instantiation, so not kept)
        if (a.interleaved && a.B > 32 && tt >= 32 && !batched_only) {
            // more partitions than one register batch: the LDS-shared kernel on the pair layout beats
            // re-reading X and Y once per batch of 32 (profiles/r01_other_configs.txt)
            launch_mac_lds<8, true>(a, s);
            return;
        }
        if (a.interleaved) {                          // the engine picked the pair layout (fp32, N >= 512)
            if (pb == 4) launch_mac_stream<4, 4>(a, s);
            else if (pb == 8) launch_mac_stream<8, 8>(a, s);
            else if (pb == 16) launch_mac_stream<16, 8>(a, s);
            else launch_mac_stream<32, 8>(a, s);
            return;
        }
        const bool lds_ok = a.N >= 512 && tt >= 32;   // 64-group tiles, 32-block time tiles
        if (lds_ok && (v == 0 || v == 8)) launch_mac_lds<8>(a, s);
        else if (lds_ok && v == 6) launch_mac_lds<4>(a, s);
        else if (lds_ok && v == 7) launch_mac_lds<2>(a, s);
        else if (tt >= 8 && v == 1) launch_mac_t<float, 8, 2, 1>(a, s);
        else if (tt >= 4 && v == 4) launch_mac_t<float, 4, 3, 2>(a, s);
        else if (tt >= 8) launch_mac_t<float, 8, 2, 2>(a, s);   // measured best (profiles/r01_mac_variants.txt)
        else if (tt >= 4) launch_mac_t<float, 4, 3, 1>(a, s);
        else if (tt >= 2) launch_mac_t<float, 2, 4, 1>(a, s);
        else launch_mac_t<float, 1, 4, 1>(a, s);
    } else {
        const int v64 = getenv("BFIR_MAC64_VARIANT") ? atoi(getenv("BFIR_MAC64_VARIANT")) : 0;   // tuning aid, read per launch
        if (a.N >= 512 && tt >= 32 && v64 == 0) launch_mac_lds_d2g<2, 1>(a, s);  // two bins per lane, 32-block tiles, two partitions per barrier
        else if (a.N >= 512 && tt >= 16 && v64 == 12) launch_mac_tstream<double, false, 16, 4, 2>(a, s);   // partition-streaming, registers only
        else if (a.N >= 512 && tt >= 32 && v64 == 7) launch_mac_lds_d2<2>(a, s);      // ... a barrier per partition
        else if (a.N >= 512 && tt >= 32 && v64 == 8) launch_mac_lds_d2g<4, 1>(a, s);  // ... four partitions per barrier
        else if (a.N >= 512 && tt >= 32 && v64 == 9) launch_mac_lds_d2g<2, 2>(a, s);  // ... two, loads two groups ahead (no gain:
                                                                                      // 0.564 vs 0.573 ms; four ahead 0.566; the loads are not what it waits for)
        else if (a.N >= 512 && tt >= 32 && v64 == 4) launch_mac_lds_d2<4>(a, s);
        else if (a.N >= 512 && tt >= 32 && v64 == 5) launch_mac_lds_d2<1>(a, s);
        else if (a.N >= 512 && tt >= 16 && (v64 == 0 || v64 == 6)) launch_mac_lds_d<2>(a, s);
        else if (a.N >= 512 && tt >= 16 && v64 == 2) launch_mac_lds_d<4>(a, s);
        else if (a.N >= 512 && tt >= 16 && v64 == 3) launch_mac_lds_d<1>(a, s);
        else if (tt >= 4) launch_mac_t<double, 4, 1, 1>(a, s);   // 1 wave/SIMD: the 2-wave build spills to scratch
        else if (tt >= 2) launch_mac_t<double, 2, 3, 1>(a, s);
        else launch_mac_t<double, 1, 4, 1>(a, s);
    }
}

// ---------------------------------------------------------------------------
// a5 / a13: staging between interleaved raw frames and planar time buffers
// ---------------------------------------------------------------------------
// One thread moves one whole frame (C samples, at most 64 bytes): the
// interleaved side is touched with the widest word the frame size and the
// buffer alignment allow (16 bytes for 8 x float32), so a wave covers a
// contiguous 2 KiB there, and the planar side lane-consecutively.  No LDS, no
// barrier: every thread is independent and keeps its loads in flight.
constexpr int STAGE_THREADS = 256;

template <int WB> struct StageWord;
template <> struct StageWord<4>  { using type = unsigned int; };
template <> struct StageWord<8>  { using type = uint2; };
template <> struct StageWord<16> { using type = uint4; };

__device__ __forceinline__ void words_from(unsigned int *w, unsigned int q) { w[0] = q; }
__device__ __forceinline__ void words_from(unsigned int *w, uint2 q) { w[0] = q.x; w[1] = q.y; }
__device__ __forceinline__ void words_from(unsigned int *w, uint4 q) { w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w; }
__device__ __forceinline__ void words_to(const unsigned int *w, unsigned int &q) { q = w[0]; }
__device__ __forceinline__ void words_to(const unsigned int *w, uint2 &q) { q.x = w[0]; q.y = w[1]; }
__device__ __forceinline__ void words_to(const unsigned int *w, uint4 &q) { q.x = w[0]; q.y = w[1]; q.z = w[2]; q.w = w[3]; }

template <typename TR> __device__ __forceinline__ TR sample_from_words(const unsigned int *w, int c);
template <> __device__ __forceinline__ float sample_from_words<float>(const unsigned int *w, int c) { return __uint_as_float(w[c]); }
template <> __device__ __forceinline__ double sample_from_words<double>(const unsigned int *w, int c)
{
    return __hiloint2double((int)w[2 * c + 1], (int)w[2 * c]);
}
__device__ __forceinline__ void sample_to_words(unsigned int *w, int c, float v) { w[c] = __float_as_uint(v); }
__device__ __forceinline__ void sample_to_words(unsigned int *w, int c, double v)
{
    w[2 * c] = (unsigned int)__double2loint(v); w[2 * c + 1] = (unsigned int)__double2hiint(v);
}

// Largest word (16 / 8 / 4 bytes) that divides the frame and every address the kernel forms.
static int stage_word_bytes(const void *raw, long eng_stride_bytes, long frame_off, int C, int spacing, int raw_bytes)
{
    if (spacing != C) return 4;   // a channel subset of a wider frame (stage API): sample by sample
    const long fb = (long)C * raw_bytes;
    const size_t base = (size_t)raw + (size_t)(frame_off * fb);
    for (int wb = 16; wb > 4; wb >>= 1)
        if (fb % wb == 0 && base % wb == 0 && eng_stride_bytes % wb == 0) return wb;
    return 4;
}

template <typename TR, typename T, int WB> __global__ __launch_bounds__(STAGE_THREADS) void k_stage_in(StageInArgs a)
{
    using W = typename StageWord<WB>::type;
    constexpr int WW = WB / 4;                       // 32-bit words per load
    const int C = a.C, e = blockIdx.y;
    const int fw = C * (int)sizeof(TR) / 4;          // 32-bit words per frame, <= 16
    const char *__restrict__ raw = (const char *)a.raw + (long)e * a.eng_stride_bytes +
                                   a.frame_off * a.spacing * (long)sizeof(TR);
    T *__restrict__ dst = (T *)a.dst + (long)e * C * a.dst_ch_stride + a.dst_off;
    const long step = (long)gridDim.x * STAGE_THREADS;
    for (long f = (long)blockIdx.x * STAGE_THREADS + threadIdx.x; f < a.n_frames; f += step) {
        unsigned int w[16];
        const W *p = (const W *)(raw + f * a.spacing * (long)sizeof(TR));
#pragma unroll
        for (int i = 0; i < 16 / WW; i++)
            if (i * WW < fw) words_from(w + i * WW, p[i]);
#pragma unroll
        for (int c = 0; c < BFIR_MAXCH; c++)
            if (c < C) dst[(long)c * a.dst_ch_stride + f] = (T)sample_from_words<TR>(w, c);
    }
}

template <typename TR, typename T> static void launch_stage_in_t(const StageInArgs &a, dim3 grid, int wb, hipStream_t s)
{
    dim3 block(STAGE_THREADS);
    if (wb == 16) hipLaunchKernelGGL((k_stage_in<TR, T, 16>), grid, block, 0, s, a);
    else if (wb == 8) hipLaunchKernelGGL((k_stage_in<TR, T, 8>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_stage_in<TR, T, 4>), grid, block, 0, s, a);
}

// ---- any of the eleven sample formats ---------------------
// Integer and big-endian formats take this byte-wise path: one thread per frame,
// each sample assembled from its bytes.  (raw2real.cpp:30-424.)  Integer samples
// keep their integer value; the 1/2^(bits-1) scale is applied by k_fwd.
template <typename T>
__device__ __forceinline__ T decode_sample(const unsigned char *p, int fmt, int bytes, bool be)
{
    unsigned long long u = 0;
    for (int k = 0; k < bytes; k++) u |= (unsigned long long)p[be ? bytes - 1 - k : k] << (8 * k);
    switch (fmt) {
    case 1: return (T)(int)(signed char)u;
    case 2: case 3: return (T)(int)(short)u;
    case 4: case 5: return (T)(((int)((unsigned int)u << 8)) >> 8);     // sign-extended 24 bit
    case 6: case 7: return (T)(int)(unsigned int)u;
    case 8: case 9: return (T)__uint_as_float((unsigned int)u);
    default: return (T)__longlong_as_double((long long)u);
    }
}

template <typename T> __global__ __launch_bounds__(STAGE_THREADS) void k_stage_in_fmt(StageInArgs a, int bytes, bool be)
{
    const int C = a.C, e = blockIdx.y;
    const unsigned char *__restrict__ raw = (const unsigned char *)a.raw + (long)e * a.eng_stride_bytes +
                                            a.frame_off * a.spacing * (long)bytes;
    T *__restrict__ dst = (T *)a.dst + (long)e * C * a.dst_ch_stride + a.dst_off;
    const long step = (long)gridDim.x * STAGE_THREADS;
    for (long f = (long)blockIdx.x * STAGE_THREADS + threadIdx.x; f < a.n_frames; f += step)
        for (int c = 0; c < C; c++)
            dst[(long)c * a.dst_ch_stride + f] = decode_sample<T>(raw + (f * a.spacing + c) * (long)bytes, a.fmt, bytes, be);
}

void launch_stage_in(const StageInArgs &a, hipStream_t s)
{
    if (a.n_frames <= 0) return;
    if (a.fmt != 0 && !fmt_is_native(a.fmt)) {
        const FmtInfo fi = fmt_info(a.fmt);
        const long nb = (a.n_frames + STAGE_THREADS - 1) / STAGE_THREADS;
        dim3 grid((unsigned)(nb < 2048 ? nb : 2048), a.n_eng), block(STAGE_THREADS);
        if (a.realsize == 4) hipLaunchKernelGGL(k_stage_in_fmt<float>, grid, block, 0, s, a, fi.bytes, fi.big_endian);
        else hipLaunchKernelGGL(k_stage_in_fmt<double>, grid, block, 0, s, a, fi.bytes, fi.big_endian);
        return;
    }
    const long nblk = (a.n_frames + STAGE_THREADS - 1) / STAGE_THREADS;
    dim3 grid((unsigned)(nblk < 2048 ? nblk : 2048), a.n_eng);
    const int wb = stage_word_bytes(a.raw, a.eng_stride_bytes, a.frame_off, a.C, a.spacing, a.raw_bytes);
    if (a.raw_bytes == 4 && a.realsize == 4) launch_stage_in_t<float, float>(a, grid, wb, s);
    else if (a.raw_bytes == 4) launch_stage_in_t<float, double>(a, grid, wb, s);
    else if (a.realsize == 4) launch_stage_in_t<double, float>(a, grid, wb, s);
    else launch_stage_in_t<double, double>(a, grid, wb, s);
}


template <typename T, typename TR, int WB> __global__ __launch_bounds__(STAGE_THREADS) void k_stage_out(StageOutArgs a)
{
    using W = typename StageWord<WB>::type;
    using Bits = decltype(abs_bits((T)0));
    constexpr int WW = WB / 4;
    __shared__ Bits red_max[STAGE_THREADS / 64][BFIR_MAXCH];
    __shared__ unsigned int red_cnt[STAGE_THREADS / 64][BFIR_MAXCH];
    const int tid = threadIdx.x, e = blockIdx.y, C = a.C;
    const int fw = C * (int)sizeof(TR) / 4;
    const T rmax = (T)a.max, rmin = (T)(-a.max);
    char *__restrict__ raw = (char *)a.raw + (long)e * a.eng_stride_bytes + a.frame_off * a.spacing * (long)sizeof(TR);
    const T *__restrict__ src = (const T *)a.src + (long)e * C * a.src_ch_stride;
    Bits mx[BFIR_MAXCH];
    unsigned int cnt[BFIR_MAXCH];
#pragma unroll
    for (int c = 0; c < BFIR_MAXCH; c++) { mx[c] = 0; cnt[c] = 0u; }
    const long step = (long)gridDim.x * STAGE_THREADS;
    for (long f = (long)blockIdx.x * STAGE_THREADS + tid; f < a.n_frames; f += step) {
        unsigned int w[16];
        const bool first_of_block = (f % a.L) == 0;
#pragma unroll
        for (int c = 0; c < BFIR_MAXCH; c++) {
            if (c < C) {
                const T v = src[(long)c * a.src_ch_stride + f];
                sample_to_words(w, c, (TR)v);
                // brutefir/real2raw.cpp:321-336: strict compares, NaN never counts
                cnt[c] += ((v < (T)0) ? (v < rmin) : (v > rmax)) ? 1u : 0u;
                const Bits bits = (v == v) ? abs_bits(v) : (Bits)0;
                mx[c] = bits > mx[c] ? bits : mx[c];
                // brutefir/brutefir.cpp:316-321: only sample 0 of each block is checked
                if (first_of_block && !isfinite((double)v)) flag_bad(a, (int)(f / a.L));
            }
        }
        W *p = (W *)(raw + f * a.spacing * (long)sizeof(TR));
#pragma unroll
        for (int i = 0; i < 16 / WW; i++)
            if (i * WW < fw) { W q; words_to(w + i * WW, q); p[i] = q; }
    }
    // one filtered atomic per block and channel: the peak only ever grows, so a
    // stale read of it can cost an extra atomic but never a wrong result
#pragma unroll
    for (int c = 0; c < BFIR_MAXCH; c++) {
        if (c < C) {
            Bits m = mx[c];
            unsigned int n = cnt[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const Bits om = __shfl_xor(m, o);
                m = om > m ? om : m;
                n += __shfl_xor(n, o);
            }
            if ((tid & 63) == 0) { red_max[tid >> 6][c] = m; red_cnt[tid >> 6][c] = n; }
        }
    }
    __syncthreads();
    if (tid < C) {
        Bits m = 0;
        unsigned int n = 0u;
        for (int w = 0; w < STAGE_THREADS / 64; w++) { m = red_max[w][tid] > m ? red_max[w][tid] : m; n += red_cnt[w][tid]; }
        DevOverflow *of = of_shard(a.overflow, a.of_shard_stride) + (e * C + tid);
        if (n) atomicAdd(&of->n_overflows, n);
        if ((unsigned long long)m > *(volatile unsigned long long *)&of->largest_bits)
            atomicMax(&of->largest_bits, (unsigned long long)m);
    }
}

template <typename T, typename TR> static void launch_stage_out_t(const StageOutArgs &a, dim3 grid, int wb, hipStream_t s)
{
    dim3 block(STAGE_THREADS);
    if (wb == 16) hipLaunchKernelGGL((k_stage_out<T, TR, 16>), grid, block, 0, s, a);
    else if (wb == 8) hipLaunchKernelGGL((k_stage_out<T, TR, 8>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_stage_out<T, TR, 4>), grid, block, 0, s, a);
}

// Any output format, no dither (real2raw.cpp:342-922, 947-1224).  Integer formats use
// dither{f,d}_real2int_no_dither (dither.cpp:196-262, 346-416): add 0.5, truncate toward
// zero, step negatives down by one, clip; n_overflows / largest count clipped samples,
// intlargest the largest unclipped |sample|.
template <typename T> __global__ __launch_bounds__(STAGE_THREADS) void k_stage_out_fmt(StageOutArgs a, int bytes, bool be, bool isfloat)
{
    using Bits = decltype(abs_bits((T)0));
    __shared__ Bits red_max[STAGE_THREADS / 64][BFIR_MAXCH];
    __shared__ unsigned int red_cnt[STAGE_THREADS / 64][BFIR_MAXCH];
    __shared__ int red_int[STAGE_THREADS / 64][BFIR_MAXCH];
    const int tid = threadIdx.x, e = blockIdx.y, C = a.C;
    const int bits = 8 * bytes;
    const int imin = (int)(0u - (1u << (bits - 1))), imax = (int)((1u << (bits - 1)) - 1u);
    const T rmax = isfloat ? (T)a.max : (T)imax, rmin = isfloat ? (T)(-a.max) : (T)imin;
    unsigned char *__restrict__ raw = (unsigned char *)a.raw + (long)e * a.eng_stride_bytes +
                                      a.frame_off * a.spacing * (long)bytes;
    const T *__restrict__ src = (const T *)a.src + (long)e * C * a.src_ch_stride;
    Bits mx[BFIR_MAXCH];
    unsigned int cnt[BFIR_MAXCH];
    int imx[BFIR_MAXCH];
#pragma unroll
    for (int c = 0; c < BFIR_MAXCH; c++) { mx[c] = 0; cnt[c] = 0u; imx[c] = 0; }
    const long step = (long)gridDim.x * STAGE_THREADS;
    for (long f = (long)blockIdx.x * STAGE_THREADS + tid; f < a.n_frames; f += step) {
        const bool first_of_block = (f % a.L) == 0;
#pragma unroll
        for (int c = 0; c < BFIR_MAXCH; c++) {
            if (c < C) {
                T v = src[(long)c * a.src_ch_stride + f];
                if (first_of_block && !isfinite((double)v)) flag_bad(a, (int)(f / a.L));
                unsigned long long u;
                if (isfloat) {
                    cnt[c] += ((v < (T)0) ? (v < rmin) : (v > rmax)) ? 1u : 0u;
                    const Bits b = (v == v) ? abs_bits(v) : (Bits)0;
                    mx[c] = b > mx[c] ? b : mx[c];
                    if (bytes == 4) u = __float_as_uint((float)v);
                    else u = (unsigned long long)__double_as_longlong((double)v);
                } else {
                    int smp;
                    v += (T)0.5;
                    if (v < (T)0) {
                        if (v <= rmin) { smp = imin; cnt[c] += 1u; const Bits b = abs_bits(v); mx[c] = b > mx[c] ? b : mx[c]; }
                        else { smp = (int)v - 1; imx[c] = -smp > imx[c] ? -smp : imx[c]; }
                    } else {
                        if (v > rmax) { smp = imax; cnt[c] += 1u; const Bits b = abs_bits(v); mx[c] = b > mx[c] ? b : mx[c]; }
                        else { smp = (int)v; imx[c] = smp > imx[c] ? smp : imx[c]; }
                    }
                    u = (unsigned int)smp;
                }
                unsigned char *p = raw + (f * a.spacing + c) * (long)bytes;
                for (int k = 0; k < bytes; k++) p[k] = (unsigned char)(u >> (8 * (be ? bytes - 1 - k : k)));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < BFIR_MAXCH; c++) {
        if (c < C) {
            Bits m = mx[c];
            unsigned int n = cnt[c];
            int im = imx[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const Bits om = __shfl_xor(m, o);
                m = om > m ? om : m;
                n += __shfl_xor(n, o);
                const int oi = __shfl_xor(im, o);
                im = oi > im ? oi : im;
            }
            if ((tid & 63) == 0) { red_max[tid >> 6][c] = m; red_cnt[tid >> 6][c] = n; red_int[tid >> 6][c] = im; }
        }
    }
    __syncthreads();
    if (tid < C) {
        Bits m = 0;
        unsigned int n = 0u;
        int im = 0;
        for (int w = 0; w < STAGE_THREADS / 64; w++) {
            m = red_max[w][tid] > m ? red_max[w][tid] : m; n += red_cnt[w][tid]; im = red_int[w][tid] > im ? red_int[w][tid] : im;
        }
        DevOverflow *of = of_shard(a.overflow, a.of_shard_stride) + (e * C + tid);
        if (n) atomicAdd(&of->n_overflows, n);
        if ((unsigned long long)m > *(volatile unsigned long long *)&of->largest_bits)
            atomicMax(&of->largest_bits, (unsigned long long)m);
        if (im > *(volatile int *)&of->intlargest) atomicMax(&of->intlargest, im);
    }
}

void launch_stage_out(const StageOutArgs &a, hipStream_t s)
{
    if (a.n_frames <= 0) return;
    if (a.dither_tab && !fmt_info(a.fmt).isfloat) { launch_stage_out_dither(a, s); return; }   // fftw_convolver.cpp:421, 444
    if (a.fmt != 0 && !fmt_is_native(a.fmt)) {
        const FmtInfo fi = fmt_info(a.fmt);
        const long nb = (a.n_frames + 4 * STAGE_THREADS - 1) / (4 * STAGE_THREADS);
        dim3 grid((unsigned)(nb < 2048 ? (nb > 0 ? nb : 1) : 2048), a.n_eng), block(STAGE_THREADS);
        if (a.realsize == 4)
            hipLaunchKernelGGL(k_stage_out_fmt<float>, grid, block, 0, s, a, fi.bytes, fi.big_endian, fi.isfloat);
        else
            hipLaunchKernelGGL(k_stage_out_fmt<double>, grid, block, 0, s, a, fi.bytes, fi.big_endian, fi.isfloat);
        return;
    }
    // a few frames per thread so the per-block reduction is amortised
    const long nblk = (a.n_frames + 4 * STAGE_THREADS - 1) / (4 * STAGE_THREADS);
    dim3 grid((unsigned)(nblk < 2048 ? (nblk > 0 ? nblk : 1) : 2048), a.n_eng);
    const int wb = stage_word_bytes(a.raw, a.eng_stride_bytes, a.frame_off, a.C, a.spacing, a.raw_bytes);
    if (a.raw_bytes == 4 && a.realsize == 4) launch_stage_out_t<float, float>(a, grid, wb, s);
    else if (a.raw_bytes == 4) launch_stage_out_t<double, float>(a, grid, wb, s);
    else if (a.realsize == 4) launch_stage_out_t<float, double>(a, grid, wb, s);
    else launch_stage_out_t<double, double>(a, grid, wb, s);
}

// ---------------------------------------------------------------------------
// mixnscale with one buffer on half-complex data (stage API only)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_reorder(const T *__restrict__ in, T *__restrict__ out, int n_fft, T sc, int to_grouped)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = n_fft >> 1;
    if (k >= half) return;
    const int gr = 8 * (k >> 2) + (k & 3);
    if (to_grouped) {  // brutefir/fftw_convolver.cpp:883-907
        out[gr] = in[k] * sc;
        out[gr + 4] = (k == 0 ? in[half] : in[n_fft - k]) * sc;
    } else {           // :1163-1186
        out[k] = in[gr] * sc;
        if (k == 0) out[half] = in[4] * sc;
        else out[n_fft - k] = in[gr + 4] * sc;
    }
}

void launch_reorder(const void *in, void *out, int n_fft, double scale, int to_grouped, int realsize,
                    hipStream_t s)
{
    const int half = n_fft / 2, threads = 256, blocks = (half + threads - 1) / threads;
    if (realsize == 4)
        hipLaunchKernelGGL(k_reorder<float>, dim3(blocks), dim3(threads), 0, s, (const float *)in,
                           (float *)out, n_fft, (float)scale, to_grouped);
    else
        hipLaunchKernelGGL(k_reorder<double>, dim3(blocks), dim3(threads), 0, s, (const double *)in,
                           (double *)out, n_fft, scale, to_grouped);
}

// ---------------------------------------------------------------------------
// stage API, SURVEY 8f row 3: N-input mixnscale, dirac_convolve, crossfade blend,
// finite check.  All keep the reference's operation order (separate roundings).
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T mul_rn(T a, T b);   // separately rounded (defined below)
template <typename T> __device__ __forceinline__ T add_rn(T a, T b);

template <typename T> __global__ void k_reorder_n(MixArgs m, T *__restrict__ out, int n_fft, int to_grouped)
{
    // brutefir/fftw_convolver.cpp:908-1156 (INPUT), :1187-1419 (OUTPUT): sum_i in_i * scale_i, left to right
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = n_fft >> 1;
    if (k >= half) return;
    const int gr = 8 * (k >> 2) + (k & 3);
    T a = (T)0, b = (T)0;
    for (int i = 0; i < m.n; i++) {
        const T *in = (const T *)m.in[i];
        const T sc = (T)m.scale[i];
        T pa, pb;
        if (to_grouped) { pa = mul_rn(in[k], sc); pb = mul_rn(k == 0 ? in[half] : in[n_fft - k], sc); }
        else { pa = mul_rn(in[gr], sc); pb = mul_rn(in[gr + 4], sc); }
        a = (i == 0) ? pa : add_rn(a, pa);
        b = (i == 0) ? pb : add_rn(b, pb);
    }
    if (to_grouped) { out[gr] = a; out[gr + 4] = b; }
    else { out[k] = a; if (k == 0) out[half] = b; else out[n_fft - k] = b; }
}

void launch_reorder_n(const MixArgs &m, void *out, int n_fft, int to_grouped, int realsize, hipStream_t s)
{
    const int half = n_fft / 2, threads = 256, blocks = (half + threads - 1) / threads;
    if (realsize == 4) hipLaunchKernelGGL(k_reorder_n<float>, dim3(blocks), dim3(threads), 0, s, m, (float *)out, n_fft, to_grouped);
    else hipLaunchKernelGGL(k_reorder_n<double>, dim3(blocks), dim3(threads), 0, s, m, (double *)out, n_fft, to_grouped);
}

template <typename T> __global__ void k_dirac(const T *in, T *out, int n_fft)
{
    // brutefir/fftw_convolver.cpp:1527-1556, 2222-2251
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_fft) return;
    const T fr = (T)(1.0 / (T)n_fft);
    out[k] = mul_rn(in[k], (k & 1) ? -fr : fr);
}

void launch_dirac(const void *in, void *out, int n_fft, int realsize, hipStream_t s)
{
    const int threads = 256, blocks = (n_fft + threads - 1) / threads;
    if (realsize == 4) hipLaunchKernelGGL(k_dirac<float>, dim3(blocks), dim3(threads), 0, s, (const float *)in, (float *)out, n_fft);
    else hipLaunchKernelGGL(k_dirac<double>, dim3(blocks), dim3(threads), 0, s, (const double *)in, (double *)out, n_fft);
}

// the time-domain blend of convolver_crossfade_inplace (brutefir/fftw_convolver.cpp:296-315)
__global__ void k_crossfade_blend_f(const float *cf, float *buf, int n_fft2)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_fft2) return;
    const float f = (float)(1.0 / (double)(float)(n_fft2 - 1));
    // crossfade[n] * (1.0 - f * (float)n) + buffer[n] * f * (float)n, with C's promotions
    const double a = mul_rn((double)cf[n], add_rn(1.0, -(double)mul_rn(f, (float)n)));
    const float b = mul_rn(mul_rn(buf[n], f), (float)n);
    buf[n] = (float)add_rn(a, (double)b);
}
__global__ void k_crossfade_blend_d(double *buf1, const double *buf2, int n_fft2)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_fft2) return;
    const double d = 1.0 / (double)(n_fft2 - 1);
    buf1[n] = add_rn(mul_rn(buf1[n], add_rn(1.0, -mul_rn(d, (double)n))), mul_rn(mul_rn(buf2[n], d), (double)n));
}

void launch_crossfade_blend(const void *crossfade_time, void *buffer_time, const void *buffer_tail, int n_fft2,
                            int realsize, hipStream_t s)
{
    const int threads = 256, blocks = (n_fft2 + threads - 1) / threads;
    if (realsize == 4)
        hipLaunchKernelGGL(k_crossfade_blend_f, dim3(blocks), dim3(threads), 0, s, (const float *)crossfade_time,
                           (float *)buffer_time, n_fft2);
    else
        hipLaunchKernelGGL(k_crossfade_blend_d, dim3(blocks), dim3(threads), 0, s, (double *)buffer_time,
                           (const double *)buffer_tail, n_fft2);
}

template <typename T> __global__ void k_check_finite(const T *buf, int n, int *bad)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && !isfinite((double)buf[k])) atomicExch(bad, 1);
}

void launch_check_finite(const void *buf, int n, int realsize, int *bad, hipStream_t s)
{
    const int threads = 256, blocks = (n + threads - 1) / threads;
    if (realsize == 4) hipLaunchKernelGGL(k_check_finite<float>, dim3(blocks), dim3(threads), 0, s, (const float *)buf, n, bad);
    else hipLaunchKernelGGL(k_check_finite<double>, dim3(blocks), dim3(threads), 0, s, (const double *)buf, n, bad);
}

// ---------------------------------------------------------------------------
// stage API: one complex multiply(-add) pass in the reference's operation order
// ---------------------------------------------------------------------------
// separately rounded multiply / add: the pragma strips the `contract` flag the
// -ffp-contract=fast build would otherwise put on these operations
template <typename T> __device__ __forceinline__ T mul_rn(T a, T b)
{
#pragma clang fp contract(off)
    return a * b;
}
template <typename T> __device__ __forceinline__ T add_rn(T a, T b)
{
#pragma clang fp contract(off)
    return a + b;
}

template <typename T>
__global__ void k_cmul_stage(const T *b, const T *c, T *d, int n_fft, int mode)
{
#pragma clang fp contract(off)   // keep every multiply and add separately rounded
    // one thread per bin; brutefir/fftw_convolver.cpp:1464-1525 (float), :2160-2220 (double)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (n_fft >> 1)) return;
    const int r = 8 * (k >> 2) + (k & 3), i = r + 4;
    const T br = b[r], bi = b[i], cr = c[r], ci = c[i];
    T pr, pi;
    if (k == 0) {  // DC and Nyquist: two independent real products
        pr = mul_rn(br, cr);
        pi = mul_rn(bi, ci);
    } else {
        pr = add_rn(mul_rn(br, cr), -mul_rn(bi, ci));
        pi = add_rn(mul_rn(br, ci), mul_rn(bi, cr));
    }
    if (mode == 1) { pr = add_rn(d[r], pr); pi = add_rn(d[i], pi); }
    d[r] = pr; d[i] = pi;
}

void launch_cmul_stage(const void *b, const void *c, void *d, int n_fft, int mode, int realsize,
                       hipStream_t s)
{
    const int half = n_fft / 2, threads = 256, blocks = (half + threads - 1) / threads;
    if (realsize == 4)
        hipLaunchKernelGGL(k_cmul_stage<float>, dim3(blocks), dim3(threads), 0, s, (const float *)b,
                           (const float *)c, (float *)d, n_fft, mode);
    else
        hipLaunchKernelGGL(k_cmul_stage<double>, dim3(blocks), dim3(threads), 0, s, (const double *)b,
                           (const double *)c, (double *)d, n_fft, mode);
}

}  // namespace bfir

#ifdef BFIR_TRACE
// tuning builds only: copy the phase stamps of kernel `kern` (0 k_fwd, 1 k_inv, 2 k_mac_lds)
extern "C" int bfir_debug_read_trace(int kern, unsigned long long *out, int n_wgs)
{
    if (kern < 0 || kern > 2 || n_wgs > BFIR_TRACE_WGS) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * n_wgs * BFIR_TRACE_SLOTS,
                                    sizeof(unsigned long long) * kern * BFIR_TRACE_WGS * BFIR_TRACE_SLOTS,
                                    hipMemcpyDeviceToHost);
}
#endif
