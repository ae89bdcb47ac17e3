// fft_lds.h -- workgroup-resident Stockham FFT for CDNA4 (gfx950), device side.
//
// One workgroup transforms one sequence of M = 2^LOG2M complex points that
// lives in registers (P = M / NT points per thread) and is exchanged between
// passes through ONE LDS buffer of M complex values.  This is the arithmetic
// that replaces the reference's FFTW r2r plans
// (brutefir/fftw_convolver.cpp:204-209 R2HC, :367-372 HC2R, plan creation
// :798-806); the real<->half-complex split steps live in kernels.hip.
//
// Pass s (radix R, p = product of the radices before it), butterfly i < M/R:
//     k      = i mod p
//     u[r]   = x[i + r*M/R] * exp(-+2 pi i r k / (p R))       r < R
//     v      = DFT_R(u)
//     y[(i-k) R + k + q p] = v[q]                              q < R
// After the last pass y is in natural order (Stockham autosort, no bit
// reversal).  Twiddles come from a per-plan table built on the host in
// double precision (see FftPlanHost in kernels.hip): tw[off(s) + (r-1) p + k].
//
// LDS layout: logical index i is stored at i + (i >> 5) (one pad element per
// 32).  With 8-byte (float2) or 16-byte (double2) elements every access of the
// passes is conflict-free on gfx950's 64-bank LDS except the stride-R writes
// of the first pass (2-way), and every address is a per-thread base plus a
// compile-time offset.  (An XOR swizzle i ^ ((i >> 4) & 15) is conflict-free
// everywhere but costs a VALU address computation per element; on a kernel
// that is bound by VALU issue -- 4 cycles per wave64 instruction, see
// profiles/r01_valu_rate.txt -- the padding is faster.)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

// -DBFIR_TRACE (tuning builds only, scripts/gpu_trace.sh): thread 0 of every workgroup stamps the
// 100 MHz wall clock at phase boundaries into a device array read back by bfir_debug_read_trace.
#ifdef BFIR_TRACE
#define BFIR_TRACE_SLOTS 24
#define BFIR_TRACE_WGS 4096
static __device__ unsigned long long g_trace[3][BFIR_TRACE_WGS * BFIR_TRACE_SLOTS];
#define BFIR_STAMP(kern, n)                                                                        \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                               \
        if (threadIdx.x == 0 && blockIdx.x < BFIR_TRACE_WGS)                                       \
            g_trace[kern][blockIdx.x * BFIR_TRACE_SLOTS + (n)] = wall_clock64();                   \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define BFIR_STAMP(kern, n) do { } while (0)
#endif

#ifndef BFIR_F64_EARLY_TW
#define BFIR_F64_EARLY_TW(lg) ((lg) >= 12)
#endif

namespace bfir {

// Force the values to exist in registers at this point of the program.  LLVM's sinking passes
// otherwise move an LDS read (and the arithmetic hanging off it) down to its first use even when
// that use sits behind a __syncthreads() in a conditional block -- observed on k_fwd<float,14>:
// ten of the sixteen Z[M-k] reads of the split step landed after the barrier that protects them,
// racing with the staging writes of other waves.  Costs no instruction.
template <typename T, int P> __device__ __forceinline__ void pin_registers(T (&re)[P], T (&im)[P])
{
#pragma unroll
    for (int e = 0; e < P; e++) asm volatile("" : "+v"(re[e]), "+v"(im[e]));
}

template <typename T> struct Vec2;
template <> struct Vec2<float>  { using type = float2; };
template <> struct Vec2<double> { using type = double2; };
template <typename T> struct Vec4;
template <> struct Vec4<float>  { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

// Threads per transform and radix sequence per size.  P = M / NT points per
// thread; every radix divides P.
template <int LOG2M> struct FftCfg;
#define BFIR_FFT_CFG(lg, nt, np, a, b, c, d)                                   \
    template <> struct FftCfg<lg> {                                            \
        static constexpr int NT = nt, NP = np, R0 = a, R1 = b, R2 = c, R3 = d; \
    };
BFIR_FFT_CFG(4, 4, 2, 4, 4, 1, 1)
BFIR_FFT_CFG(5, 4, 2, 8, 4, 1, 1)
BFIR_FFT_CFG(6, 8, 2, 8, 8, 1, 1)
BFIR_FFT_CFG(7, 16, 3, 8, 4, 4, 1)
BFIR_FFT_CFG(8, 16, 2, 16, 16, 1, 1)
BFIR_FFT_CFG(9, 32, 3, 16, 8, 4, 1)
BFIR_FFT_CFG(10, 64, 3, 16, 16, 4, 1)
BFIR_FFT_CFG(11, 128, 3, 16, 16, 8, 1)
BFIR_FFT_CFG(12, 256, 3, 16, 16, 16, 1)
BFIR_FFT_CFG(13, 512, 4, 16, 16, 8, 4)
BFIR_FFT_CFG(14, 1024, 4, 16, 16, 16, 4)
#undef BFIR_FFT_CFG

constexpr int BFIR_MIN_LOG2M = 4;
constexpr int BFIR_MAX_LOG2M = 14;

// ---- small complex helpers on split re/im registers -----------------------
template <typename T>
__device__ __forceinline__ void cmul(T &re, T &im, T wr, T wi)
{
    T tr = re * wr - im * wi;
    T ti = re * wi + im * wr;
    re = tr; im = ti;
}

// multiply by exp(SIGN * 2 pi i m / 16)  (SIGN = -1 forward, +1 inverse)
template <int MM, int SIGN, typename T>
__device__ __forceinline__ void mul_w16(T &re, T &im)
{
    constexpr int m = ((MM % 16) + 16) % 16;
    constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173,
                     RH = 0.70710678118654752440;
    constexpr double cs[16] = {1, C1, RH, S1, 0, -S1, -RH, -C1, -1, -C1, -RH, -S1, 0, S1, RH, C1};
    constexpr double sn[16] = {0, S1, RH, C1, 1, C1, RH, S1, 0, -S1, -RH, -C1, -1, -C1, -RH, -S1};
    if constexpr (m == 0) {
    } else if constexpr (m == 8) {
        re = -re; im = -im;
    } else if constexpr (m == 4) {          // exp(SIGN i pi/2) = SIGN * i
        T t = re;
        if constexpr (SIGN < 0) { re = im; im = -t; } else { re = -im; im = t; }
    } else if constexpr (m == 12) {         // -SIGN * i
        T t = re;
        if constexpr (SIGN < 0) { re = -im; im = t; } else { re = im; im = -t; }
    } else {
        cmul(re, im, (T)cs[m], (T)(SIGN * sn[m]));
    }
}

// In-register DFTs.  Output X[q] is left at register position pos(q) (inv(p) is
// the q held at position p): the radix-8/16 kernels finish in digit-swapped
// order and their consumers index through pos(), which is free (compile-time
// indices) where a transpose to natural order would cost 2R register moves.
template <int R, int SIGN, typename T> struct Dft;

template <int SIGN, typename T> struct Dft<2, SIGN, T> {
    __host__ __device__ static constexpr int pos(int q) { return q; }
    __host__ __device__ static constexpr int inv(int p) { return p; }
    __device__ __forceinline__ static void run(T *re, T *im)
    {
        T ar = re[0], ai = im[0];
        re[0] = ar + re[1]; im[0] = ai + im[1];
        re[1] = ar - re[1]; im[1] = ai - im[1];
    }
};

template <int SIGN, typename T> struct Dft<4, SIGN, T> {
    __host__ __device__ static constexpr int pos(int q) { return q; }
    __host__ __device__ static constexpr int inv(int p) { return p; }
    // stride lets the radix-8/16 kernels run it on interleaved sub-sequences
    template <int STRIDE = 1>
    __device__ __forceinline__ static void run(T *re, T *im)
    {
        T s0r = re[0] + re[2 * STRIDE], s0i = im[0] + im[2 * STRIDE];
        T d0r = re[0] - re[2 * STRIDE], d0i = im[0] - im[2 * STRIDE];
        T s1r = re[STRIDE] + re[3 * STRIDE], s1i = im[STRIDE] + im[3 * STRIDE];
        T d1r = re[STRIDE] - re[3 * STRIDE], d1i = im[STRIDE] - im[3 * STRIDE];
        re[0] = s0r + s1r;          im[0] = s0i + s1i;
        re[2 * STRIDE] = s0r - s1r; im[2 * STRIDE] = s0i - s1i;
        if constexpr (SIGN < 0) {   // X1 = d0 - i d1, X3 = d0 + i d1
            re[STRIDE] = d0r + d1i;     im[STRIDE] = d0i - d1r;
            re[3 * STRIDE] = d0r - d1i; im[3 * STRIDE] = d0i + d1r;
        } else {
            re[STRIDE] = d0r - d1i;     im[STRIDE] = d0i + d1r;
            re[3 * STRIDE] = d0r + d1i; im[3 * STRIDE] = d0i - d1r;
        }
    }
};

// 8 = 4 x 2:  r = 2 r1 + r2,  q = q1 + 4 q2
template <int SIGN, typename T> struct Dft<8, SIGN, T> {
    // X[q1 + 4 q2] ends at 2 q1 + q2
    __host__ __device__ static constexpr int pos(int q) { return 2 * (q & 3) + (q >> 2); }
    __host__ __device__ static constexpr int inv(int p) { return (p >> 1) + 4 * (p & 1); }
    __device__ __forceinline__ static void run(T *re, T *im)
    {
        Dft<4, SIGN, T>::template run<2>(re, im);          // r2 = 0: elements 0,2,4,6
        Dft<4, SIGN, T>::template run<2>(re + 1, im + 1);  // r2 = 1: elements 1,3,5,7
        // t[r2][q1] sits at 2 q1 + r2; twiddle w8^{q1} on r2 = 1
        mul_w16<2, SIGN>(re[3], im[3]);
        mul_w16<4, SIGN>(re[5], im[5]);
        mul_w16<6, SIGN>(re[7], im[7]);
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) {                   // 2-point DFT over r2, in place
            const T ar = re[2 * q1], ai = im[2 * q1];
            re[2 * q1] = ar + re[2 * q1 + 1];     im[2 * q1] = ai + im[2 * q1 + 1];
            re[2 * q1 + 1] = ar - re[2 * q1 + 1]; im[2 * q1 + 1] = ai - im[2 * q1 + 1];
        }
    }
};

// 16 = 4 x 4:  r = 4 r1 + r2,  q = q1 + 4 q2
template <int SIGN, typename T> struct Dft<16, SIGN, T> {
    // X[q1 + 4 q2] ends at 4 q1 + q2 (an involution)
    __host__ __device__ static constexpr int pos(int q) { return 4 * (q & 3) + (q >> 2); }
    __host__ __device__ static constexpr int inv(int p) { return 4 * (p & 3) + (p >> 2); }
    __device__ __forceinline__ static void run(T *re, T *im)
    {
        // step 1: for each r2, 4-point DFT over r1 (elements r2, r2+4, r2+8, r2+12):
        // result t[r2][q1] lands at 4 q1 + r2
        Dft<4, SIGN, T>::template run<4>(re, im);
        Dft<4, SIGN, T>::template run<4>(re + 1, im + 1);
        Dft<4, SIGN, T>::template run<4>(re + 2, im + 2);
        Dft<4, SIGN, T>::template run<4>(re + 3, im + 3);
        // step 2: t[r2][q1] *= w16^{r2 q1}
        mul_w16<1, SIGN>(re[5], im[5]);   mul_w16<2, SIGN>(re[6], im[6]);   mul_w16<3, SIGN>(re[7], im[7]);
        mul_w16<2, SIGN>(re[9], im[9]);   mul_w16<4, SIGN>(re[10], im[10]); mul_w16<6, SIGN>(re[11], im[11]);
        mul_w16<3, SIGN>(re[13], im[13]); mul_w16<6, SIGN>(re[14], im[14]); mul_w16<9, SIGN>(re[15], im[15]);
        // step 3: for each q1, 4-point DFT over r2 (contiguous 4 q1 .. 4 q1 + 3):
        // X[q1 + 4 q2] lands at 4 q1 + q2 and stays there (see pos())
        Dft<4, SIGN, T>::template run<1>(re, im);
        Dft<4, SIGN, T>::template run<1>(re + 4, im + 4);
        Dft<4, SIGN, T>::template run<1>(re + 8, im + 8);
        Dft<4, SIGN, T>::template run<1>(re + 12, im + 12);
    }
};

// compile-time loop: f(integral_constant<int, I>) for I = LO .. HI-1 (register arrays keep constant indices)
template <int LO, int HI, typename Fn> __device__ __forceinline__ void static_for(Fn &&f)
{
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        static_for<LO + 1, HI>(f);
    }
}

// Twiddle bases.  A pass of radix R multiplies input r of a butterfly by w^{r k}.  Instead of fetching
// all R-1 of them per transform, a PERSISTENT workgroup keeps w^{e k0} for a few exponents e in
// registers for its whole life (k0 = tid mod min(p, NT) never changes) and forms the others as the
// product of two of them: r = 4a + c -> w^{4a} w^{c} (radix 16: bases 1 2 3 4 8 12; radix 8: 1 2 3 4;
// radix 4: 1 2, w^3 = w w^2).
// One extra rounding per derived twiddle; no vector-memory traffic for twiddles in steady state.
__host__ __device__ constexpr int tw_nbase(int R) { return R == 16 ? 6 : R == 8 ? 4 : R == 4 ? 2 : 1; }
__host__ __device__ constexpr int tw_base_exp(int R, int j)
{
    return R == 16 ? (j < 4 ? j + 1 : 4 * (j - 2)) : j + 1;      // 16: 1 2 3 4 8 12; 8: 1 2 3 4; 4: 1 2; 2: 1
}

// ---- the workgroup transform ------------------------------------------------
template <typename T, int LOG2M, int SIGN> struct LdsFft {
    using Cfg = FftCfg<LOG2M>;
    using V2 = typename Vec2<T>::type;
    static constexpr int M = 1 << LOG2M;
    static constexpr int NT = Cfg::NT;
    static constexpr int P = M / NT;
    static constexpr int NP = Cfg::NP;

    __host__ __device__ static constexpr int radix(int s)
    {
        return s == 0 ? Cfg::R0 : s == 1 ? Cfg::R1 : s == 2 ? Cfg::R2 : Cfg::R3;
    }
    __host__ __device__ static constexpr int pprod(int s)
    {
        int p = 1;
        for (int j = 0; j < s; j++) p *= radix(j);
        return p;
    }
    // offset of pass s in the twiddle table (pass 0 needs none)
    __host__ __device__ static constexpr int twoff(int s)
    {
        int o = 0;
        for (int j = 1; j < s; j++) o += (radix(j) - 1) * pprod(j);
        return o;
    }
    __host__ __device__ static constexpr int twsize() { return twoff(NP); }

    // LDS placement of logical index i: one element of padding per 32.  Every
    // access pattern of the passes then is "per-thread base + compile-time offset"
    // (no per-element address arithmetic); all of them are bank-conflict free
    // except the stride-R writes of the first pass (2-way).
#ifdef BFIR_FFT_XOR_SWIZZLE   // A/B switch: conflict-free XOR layout, exactly M elements (5 workgroups per CU at M = 4096 fp32)
    static constexpr int LDS_ELEMS = M;
    __host__ __device__ __forceinline__ static constexpr int phys(int i) { return i ^ ((i >> 4) & 15); }
#else
    static constexpr int LDS_ELEMS = M + M / 32;
    __host__ __device__ __forceinline__ static constexpr int phys(int i) { return i + (i >> 5); }
#endif

    // logical index held in register slot e before pass 0
    __host__ __device__ __forceinline__ static constexpr int in_index(int tid, int e)
    {
        return (tid + (e / radix(0)) * NT) + (e % radix(0)) * (M / radix(0));
    }
    // logical index held in register slot e after the last pass
    __host__ __device__ __forceinline__ static constexpr int out_index(int tid, int e)
    {
        return (tid + (e / radix(NP - 1)) * NT) + Dft<radix(NP - 1), SIGN, T>::inv(e % radix(NP - 1)) * (M / radix(NP - 1));
    }

    // twiddles of pass S for this thread: (P/R) butterflies x (R-1) factors
    template <int S> static constexpr int ntw() { return S == 0 ? 1 : (P / radix(S)) * (radix(S) - 1); }

    // Fetch the twiddles of pass S.  Called BEFORE the exchange that feeds pass S (they do not
    // depend on the data), so the L2 latency of these loads hides behind the barriers and LDS
    // traffic of the exchange instead of stalling the first butterfly of the pass.
    template <int S>
    __device__ __forceinline__ static void load_twiddles(V2 (&w)[ntw<S>()], const V2 *__restrict__ tw, int tid)
    {
        constexpr int R = radix(S), p = pprod(S);
#pragma unroll
        for (int b = 0; b < P / R; b++) {
            const int k = (tid + b * NT) & (p - 1);
#pragma unroll
            for (int r = 1; r < R; r++) w[b * (R - 1) + r - 1] = tw[twoff(S) + (r - 1) * p + k];
        }
    }

    template <int S>
    __device__ __forceinline__ static void butterflies(T *re, T *im, const V2 (&w)[ntw<S>()])
    {
        constexpr int R = radix(S), p = pprod(S);
#pragma unroll
        for (int b = 0; b < P / R; b++) {
            if constexpr (p > 1) {
#pragma unroll
                for (int r = 1; r < R; r++) {
                    const V2 ww = w[b * (R - 1) + r - 1];
                    cmul(re[b * R + r], im[b * R + r], ww.x, (T)(SIGN < 0 ? ww.y : -ww.y));
                }
            }
            Dft<R, SIGN, T>::run(re + b * R, im + b * R);
        }
    }

    // one pass after the first: twiddle fetch, exchange, butterflies.  Small fp64 transforms have no
    // registers to spare for the early fetch (data alone is 2 x 2P registers; their occupancy is bound by
    // registers), they load late; from 4096 points on the LDS buffer (66 KB) allows two workgroups per CU
    // whatever the register count, and the fetch goes in front of the exchange as in fp32.
    template <int S>
    __device__ __forceinline__ static void pass(T *re, T *im, V2 *lds, const V2 *__restrict__ tw, int tid)
    {
        V2 w[ntw<S>()];
        if constexpr (sizeof(T) == 4 || BFIR_F64_EARLY_TW(LOG2M)) {
            load_twiddles<S>(w, tw, tid);
            exchange<S - 1>(re, im, lds, tid);
        } else {
            exchange<S - 1>(re, im, lds, tid);
            load_twiddles<S>(w, tw, tid);
        }
        butterflies<S>(re, im, w);
    }

    // Logical LDS index of output q of butterfly b after pass S / of input r of butterfly b before pass S+1.
    template <int S> __host__ __device__ static constexpr int widx(int tid, int b, int q)
    {
        const int R = radix(S), p = pprod(S);
        const int i = tid + b * NT, k = i & (p - 1);
        return (i - k) * R + k + q * p;
    }
    template <int S> __host__ __device__ static constexpr int ridx(int tid, int b, int r)
    {
        return (tid + b * NT) + r * (M / radix(S + 1));
    }
    // Is phys(index(tid, b, q)) = phys(index(tid, 0, 0)) + a compile-time constant for EVERY thread?  Then an
    // exchange costs one address per thread and direction, the rest are immediate offsets of the LDS
    // instructions.  (The compiler cannot see it through i + (i >> 5) and otherwise spends two VALU
    // instructions per access: a fifth of the persistent pair kernels' vector instructions.)  Checked
    // exhaustively at compile time; sizes that fail keep the per-access form.
    template <int S> __host__ __device__ static constexpr bool wr_affine()
    {
        for (int b = 0; b < P / radix(S); b++)
            for (int q = 0; q < radix(S); q++) {
                const int off = phys(widx<S>(0, b, q)) - phys(widx<S>(0, 0, 0));
                for (int t = 1; t < NT; t++)
                    if (phys(widx<S>(t, b, q)) - phys(widx<S>(t, 0, 0)) != off) return false;
            }
        return true;
    }
    template <int S> __host__ __device__ static constexpr bool rd_affine()
    {
        for (int b = 0; b < P / radix(S + 1); b++)
            for (int r = 0; r < radix(S + 1); r++) {
                const int off = phys(ridx<S>(0, b, r)) - phys(ridx<S>(0, 0, 0));
                for (int t = 1; t < NT; t++)
                    if (phys(ridx<S>(t, b, r)) - phys(ridx<S>(t, 0, 0)) != off) return false;
            }
        return true;
    }

    // write the outputs of pass S to LDS and fetch the inputs of pass S+1
    template <int S>
    __device__ __forceinline__ static void exchange(T *re, T *im, V2 *lds, int tid)
    {
        constexpr int R = radix(S), Rn = radix(S + 1);
        __syncthreads();  // earlier readers of lds are done
        if constexpr (wr_affine<S>()) {
            V2 *wp = lds + phys(widx<S>(tid, 0, 0));
            static_for<0, P / R>([&](auto B_) {
                constexpr int b = decltype(B_)::value;
                static_for<0, R>([&](auto Q_) {
                    constexpr int q = decltype(Q_)::value;
                    constexpr int off = phys(widx<S>(0, b, q)) - phys(widx<S>(0, 0, 0));
                    V2 v; v.x = re[b * R + Dft<R, SIGN, T>::pos(q)]; v.y = im[b * R + Dft<R, SIGN, T>::pos(q)];
                    wp[off] = v;
                });
            });
        } else {
#pragma unroll
            for (int b = 0; b < P / R; b++)
#pragma unroll
                for (int q = 0; q < R; q++) {
                    V2 v; v.x = re[b * R + Dft<R, SIGN, T>::pos(q)]; v.y = im[b * R + Dft<R, SIGN, T>::pos(q)];
                    lds[phys(widx<S>(tid, b, q))] = v;
                }
        }
        __syncthreads();
        if constexpr (rd_affine<S>()) {
            const V2 *rp = lds + phys(ridx<S>(tid, 0, 0));
            static_for<0, P / Rn>([&](auto B_) {
                constexpr int b = decltype(B_)::value;
                static_for<0, Rn>([&](auto R_) {
                    constexpr int r = decltype(R_)::value;
                    constexpr int off = phys(ridx<S>(0, b, r)) - phys(ridx<S>(0, 0, 0));
                    const V2 v = rp[off];
                    re[b * Rn + r] = v.x; im[b * Rn + r] = v.y;
                });
            });
        } else {
#pragma unroll
            for (int b = 0; b < P / Rn; b++)
#pragma unroll
                for (int r = 0; r < Rn; r++) {
                    V2 v = lds[phys(ridx<S>(tid, b, r))];
                    re[b * Rn + r] = v.x; im[b * Rn + r] = v.y;
                }
        }
        // the reads must not sink below the next barrier (see pin_registers)
#pragma unroll
        for (int e = 0; e < P; e++) asm volatile("" : "+v"(re[e]), "+v"(im[e]));
    }

    // ---- twiddle bases: table layout, registers, derived twiddles -------------------------------
    // table: for pass s >= 1, base j, k0 < kspan(s):  twb[boff(s) + j kspan(s) + k0] = exp(-2 pi i e_j k0 / (p R))
    __host__ __device__ static constexpr int kspan(int s) { return pprod(s) < NT ? pprod(s) : NT; }
    __host__ __device__ static constexpr int boff(int s)
    {
        int o = 0;
        for (int j = 1; j < s; j++) o += tw_nbase(radix(j)) * kspan(j);
        return o;
    }
    __host__ __device__ static constexpr int bsize() { return boff(NP); }
    // Where a persistent workgroup keeps them: the bases of the LAST pass (k0 = tid, the big set) in
    // NBREG registers per thread, those of the passes before it (k0 < 256: a few KB) in LDS, copied once.
    static constexpr int NBREG = NP > 1 ? tw_nbase(radix(NP - 1)) : 1;
    static constexpr int LDSB_ELEMS = NP > 2 ? boff(NP - 1) : 1;

    // once per workgroup; a barrier must follow before the first butterflies_tb
    __device__ __forceinline__ static void load_bases(V2 (&B)[NBREG], V2 *ldsb, const V2 *__restrict__ twb, int tid)
    {
        if constexpr (NP > 2)
            for (int i = tid; i < LDSB_ELEMS; i += NT) ldsb[i] = twb[i];
        if constexpr (NP > 1)
            static_for<0, NBREG>([&](auto J_) {
                constexpr int J = decltype(J_)::value;
                B[J] = twb[boff(NP - 1) + J * kspan(NP - 1) + (tid & (kspan(NP - 1) - 1))];
            });
    }

    template <int S, int J> __device__ __forceinline__ static V2 base(const V2 (&B)[NBREG], const V2 *ldsb, int tid)
    {
        if constexpr (S == NP - 1) return B[J];
        else return ldsb[boff(S) + J * kspan(S) + (tid & (kspan(S) - 1))];
    }

    // w^{r k0} of pass S in the forward convention exp(-i theta): a base or the product of two
    template <int S, int r> __device__ __forceinline__ static V2 base_twiddle(const V2 (&B)[NBREG], const V2 *ldsb, int tid)
    {
        constexpr int R = radix(S);
        if constexpr (R == 16 && (r & 3) != 0 && (r >> 2) != 0) {
            const V2 a = base<S, 2 + (r >> 2)>(B, ldsb, tid), c = base<S, (r & 3) - 1>(B, ldsb, tid);
            V2 w; w.x = a.x * c.x - a.y * c.y; w.y = a.x * c.y + a.y * c.x;
            return w;
        } else if constexpr (R == 16) {
            return base<S, ((r >> 2) == 0 ? r - 1 : 2 + (r >> 2))>(B, ldsb, tid);
        } else if constexpr (R == 8 && r > 4) {
            const V2 a = base<S, 3>(B, ldsb, tid), c = base<S, r - 5>(B, ldsb, tid);
            V2 w; w.x = a.x * c.x - a.y * c.y; w.y = a.x * c.y + a.y * c.x;
            return w;
        } else if constexpr (R == 4 && r == 3) {
            const V2 a = base<S, 0>(B, ldsb, tid), c = base<S, 1>(B, ldsb, tid);
            V2 w; w.x = a.x * c.x - a.y * c.y; w.y = a.x * c.y + a.y * c.x;
            return w;
        } else {
            return base<S, r - 1>(B, ldsb, tid);
        }
    }

    // pass S (S >= 1) on register data with twiddles from the bases.  Butterfly b of a thread has
    // k = (tid + b NT) mod p: the same k0 for every b when p <= NT, otherwise k0 + NT b' with
    // b' = b mod (p / NT), i.e. a further constant rotation by w^{r NT b'} = exp(-2 pi i r b' / 16)
    // (NT / (p R) = 1 / 16 whenever p > NT: such a pass is the last one and P = 16).
    template <int S>
    __device__ __forceinline__ static void butterflies_tb(T *re, T *im, const V2 (&B)[NBREG], const V2 *ldsb, int tid)
    {
        constexpr int R = radix(S), p = pprod(S);
        static_assert(p <= NT || 16 * NT == p * R, "constant rotations are sixteenth roots of unity");
        // one twiddle live at a time: derive w^{r k0}, apply it to input r of every butterfly, move on
        static_for<1, R>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            // radix 16: four twiddles at a time.  Left alone the scheduler derives all fifteen first (30
            // registers), which a persistent kernel with its prefetch registers cannot afford.
            if constexpr (R == 16 && (r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const V2 w = base_twiddle<S, r>(B, ldsb, tid);
            static_for<0, P / R>([&](auto B_) {
                constexpr int b = decltype(B_)::value;
                constexpr int bp = b % (p > NT ? p / NT : 1);            // 0 when p <= NT
                cmul(re[b * R + r], im[b * R + r], w.x, (T)(SIGN < 0 ? w.y : -w.y));
                if constexpr (bp != 0) mul_w16<(r * bp) % 16, SIGN>(re[b * R + r], im[b * R + r]);
            });
        });
        static_for<0, P / R>([&](auto B_) {
            constexpr int b = decltype(B_)::value;
            Dft<R, SIGN, T>::run(re + b * R, im + b * R);
        });
    }

    // Full transform of the P register points (in_index order in, out_index order out).
    __device__ __forceinline__ static void run(T *re, T *im, V2 *lds, const V2 *__restrict__ tw, int tid)
    {
        constexpr int TK = SIGN < 0 ? 0 : 1;   // trace array: forward / inverse
        {
            V2 w0[1];
            butterflies<0>(re, im, w0);
        }
        BFIR_STAMP(TK, 2);
        if constexpr (NP > 1) { pass<1>(re, im, lds, tw, tid); BFIR_STAMP(TK, 4); }
        if constexpr (NP > 2) { pass<2>(re, im, lds, tw, tid); BFIR_STAMP(TK, 6); }
        if constexpr (NP > 3) { pass<3>(re, im, lds, tw, tid); BFIR_STAMP(TK, 8); }
        (void)TK;
    }
};

}  // namespace bfir
