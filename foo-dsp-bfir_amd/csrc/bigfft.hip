// bigfft.hip -- FFTW-r2r-shaped plans of any power-of-two size on the GPU:
// bfir_fft_plan_* of include/bfir_hip.h.
//
// The reference hands out raw FFTW plans through fftw_convolver::create_fft_plan
// (brutefir/fftw_convolver.cpp:653-675) and equalizer.cpp executes one of them
// directly (FFTW_HC2R of 2*65536 points, equalizer.cpp:54-55, 262, 357).  A
// transform of that size does not fit one workgroup's LDS, so it is done in four
// steps over the workgroup-resident FFT of fft_lds.h:
//   M complex points, M = M1*M2, input index n = n1*M2 + n2, output k = k1 + M1*k2
//   1. transpose [M1][M2] -> [M2][M1]
//   2. M2 row FFTs of length M1, times exp(-+2 pi i n2 k1 / M)
//   3. transpose back
//   4. M1 row FFTs of length M2
//   5. transpose: X[k1 + M1 k2] lands in natural order
// Real transforms wrap a complex one of half the size with the same split step the
// convolution kernels use.  Init-time code: clarity over speed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/bfir_hip.h"
#include "fft_lds.h"
#include "kernels.h"

using namespace bfir;

void bfir_logf(const char *fmt, ...);

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            bfir_logf("HIP error %s at %s:%d", hipGetErrorString(_e), __FILE__, __LINE__); \
            return BFIR_ERR_HIP;                                                   \
        }                                                                          \
    } while (0)

namespace {

// exp(sign * 2 pi i num / den), evaluated in double whatever T is
template <typename T> __device__ __forceinline__ void unit_root(long num, long den, int sign, T &c, T &s)
{
    double sn, cs;
    sincospi(2.0 * (double)(num % den) / (double)den, &sn, &cs);
    c = (T)cs; s = (T)(sign < 0 ? -sn : sn);
}

// One workgroup = one row of M contiguous complex values; optional twiddle
// exp(sign 2 pi i row*k / m_total) on output element k.
template <typename T, int LOG2M, int SIGN>
__global__ __launch_bounds__(FftCfg<LOG2M>::NT) void k_cfft_rows(const typename Vec2<T>::type *__restrict__ in,
                                                                typename Vec2<T>::type *__restrict__ out,
                                                                const typename Vec2<T>::type *__restrict__ tw,
                                                                long m_total, int twiddle)
{
    using F = LdsFft<T, LOG2M, SIGN>;
    using V2 = typename Vec2<T>::type;
    constexpr int M = F::M, P = F::P;
    __shared__ __attribute__((aligned(16))) V2 lds[F::LDS_ELEMS];
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    const V2 *src = in + row * M;
    V2 *dst = out + row * M;
    T re[P], im[P];
#pragma unroll
    for (int e = 0; e < P; e++) { V2 v = src[F::in_index(tid, e)]; re[e] = v.x; im[e] = v.y; }
    F::run(re, im, lds, tw, tid);
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int k = F::out_index(tid, e);
        if (twiddle) { T c, s; unit_root<T>(row * (long)k, m_total, SIGN, c, s); cmul(re[e], im[e], c, s); }
        V2 v; v.x = re[e]; v.y = im[e];
        dst[k] = v;
    }
}

template <typename V2> __global__ void k_transpose(const V2 *__restrict__ in, V2 *__restrict__ out, int rows, int cols)
{
    __shared__ V2 tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[(long)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = bx + j, c = by + threadIdx.x;      // position in the transposed array [cols][rows]
        if (r < cols && c < rows) out[(long)r * rows + c] = tile[threadIdx.x][j];
    }
}

// half-complex (FFTW order) -> Z_k = (X_k + conj X_{M-k}) + i conj(W^k)(X_k - conj X_{M-k}), W = exp(-2 pi i / n)
template <typename T> __global__ void k_hc_to_z(const T *__restrict__ hc, typename Vec2<T>::type *__restrict__ z, int n)
{
    const int M = n >> 1, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    T xr, xi, yr, yi;
    if (k == 0) { xr = hc[0]; xi = 0; yr = hc[M]; yi = 0; }
    else { xr = hc[k]; xi = hc[n - k]; yr = hc[M - k]; yi = (M - k == k) ? hc[n - k] : hc[n - (M - k)]; }
    T c, s;
    unit_root<T>(k, n, -1, c, s);                         // W^k = (c, s); conj -> (c, -s)
    const T ar = xr + yr, ai = xi - yi, br = xr - yr, bi = xi + yi;
    const T tr = br * c + bi * s, ti = bi * c - br * s;
    typename Vec2<T>::type v; v.x = ar - ti; v.y = ai + tr;
    z[k] = v;
}

// Z (forward FFT of the packed real sequence) -> half-complex X
template <typename T> __global__ void k_z_to_hc(const typename Vec2<T>::type *__restrict__ z, T *__restrict__ hc, int n)
{
    const int M = n >> 1, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M / 2) return;
    if (k == 0) { hc[0] = z[0].x + z[0].y; hc[M] = z[0].x - z[0].y; return; }
    const int q = M - k;
    const T er = (T)0.5 * (z[k].x + z[q].x), ei = (T)0.5 * (z[k].y - z[q].y);
    const T orr = (T)0.5 * (z[k].y + z[q].y), oi = (T)-0.5 * (z[k].x - z[q].x);
    T c, s;
    unit_root<T>(k, n, -1, c, s);
    const T tr = orr * c - oi * s, ti = orr * s + oi * c;
    hc[k] = er + tr; hc[n - k] = ei + ti;
    if (q != k) { hc[q] = er - tr; hc[n - q] = -(ei - ti); }
}

// Transforms of 2 ... 16 reals (plans of order 1 ... 4: the td convolver of a filter of up to eight taps,
// fftw_convolver.cpp:726-727) are below the workgroup FFT's smallest size: one lane per output, the
// defining sums in double.
template <typename T> __global__ void k_small_r2hc(const T *__restrict__ x, T *__restrict__ hc, int n)
{
    const int k = threadIdx.x;
    if (k > n / 2) return;
    double re = 0.0, im = 0.0;
    for (int j = 0; j < n; j++) {
        double sn, cs;
        sincospi(2.0 * (double)((j * k) % n) / (double)n, &sn, &cs);
        re += (double)x[j] * cs; im -= (double)x[j] * sn;
    }
    hc[k] = (T)re;
    if (k > 0 && k < n / 2) hc[n - k] = (T)im;
}

template <typename T> __global__ void k_small_hc2r(const T *__restrict__ hc, T *__restrict__ x, int n)
{
    const int j = threadIdx.x;
    if (j >= n) return;
    double acc = (double)hc[0] + ((j & 1) ? -(double)hc[n / 2] : (double)hc[n / 2]);
    for (int k = 1; k < n / 2; k++) {
        double sn, cs;
        sincospi(2.0 * (double)((j * k) % n) / (double)n, &sn, &cs);
        acc += 2.0 * ((double)hc[k] * cs - (double)hc[n - k] * sn);
    }
    x[j] = (T)acc;
}

// every value times sc (convolver_td_new's normalisation, fftw_convolver.cpp:740-753)
template <typename T> __global__ void k_hc_scale(T *__restrict__ a, T sc, long n)
{
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] *= sc;
}

// convolve_inplace_ordered (fftw_convolver.cpp:819-856): half-complex product in place in b, the
// reference's separate multiplies and adds
template <typename T> __global__ void k_hc_mul_ordered(T *__restrict__ b, const T *__restrict__ c, long size)
{
#pragma clang fp contract(off)
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x, size2 = size >> 1;
    if (n > size2) return;
    if (n == 0 || n == size2) { b[n] *= c[n]; return; }
    const T a = b[n], bi = b[size - n];
    b[n] = a * c[n] - bi * c[size - n];
    b[size - n] = a * c[size - n] + bi * c[n];
}

#define BFIR_ROWS_CASE(lg)                                                                                  \
    case lg:                                                                                                \
        if (sign < 0) hipLaunchKernelGGL((k_cfft_rows<T, lg, -1>), dim3(rows), dim3(FftCfg<lg>::NT), 0, s,  \
                                         (const V2 *)in, (V2 *)out, (const V2 *)tw, m_total, twiddle);      \
        else hipLaunchKernelGGL((k_cfft_rows<T, lg, +1>), dim3(rows), dim3(FftCfg<lg>::NT), 0, s,           \
                                (const V2 *)in, (V2 *)out, (const V2 *)tw, m_total, twiddle);               \
        break;

template <typename T> void launch_rows(int lg, int sign, const void *in, void *out, const void *tw, int rows,
                                       long m_total, int twiddle, hipStream_t s)
{
    using V2 = typename Vec2<T>::type;
    switch (lg) {
        BFIR_ROWS_CASE(4) BFIR_ROWS_CASE(5) BFIR_ROWS_CASE(6) BFIR_ROWS_CASE(7) BFIR_ROWS_CASE(8)
        BFIR_ROWS_CASE(9) BFIR_ROWS_CASE(10) BFIR_ROWS_CASE(11) BFIR_ROWS_CASE(12)
    }
}

template <typename T> void launch_transpose(const void *in, void *out, int rows, int cols, hipStream_t s)
{
    using V2 = typename Vec2<T>::type;
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(k_transpose<V2>, grid, block, 0, s, (const V2 *)in, (V2 *)out, rows, cols);
}

}  // namespace

struct bfir_fft_plan {
    int device = 0, order = 0, invert = 0, realsize = 0;
    long n = 0;                 // real length
    int lm1 = 0, lm2 = 0;       // complex length 2^(lm1+lm2); lm2 == 0: one row pass
    FftPlan p1, p2;
    void *d_a = nullptr, *d_b = nullptr;   // two buffers of n reals (= n/2 complex)
    hipStream_t stream = nullptr;
};

// fftw_convolver::create_fft_plan / get_fft_plan (fftw_convolver.cpp:653-675, 779-817):
// a FFTW_R2HC (invert == 0) or FFTW_HC2R plan of 2^order reals.  `inplace` only selects a
// plan variant in FFTW; execute accepts in == out either way.
extern "C" bfir_fft_plan *bfir_fft_plan_create(int order, int invert, int inplace, int realsize, int device, int *err)
{
    (void)inplace;
    int dummy;
    if (!err) err = &dummy;
    *err = BFIR_OK;
    if ((realsize != 4 && realsize != 8) || order < 1 || order > 25) { *err = BFIR_ERR_ARG; return nullptr; }
    int ndev = bfir_device_count();
    if (ndev <= 0) { *err = BFIR_ERR_NO_DEVICE; return nullptr; }
    if (device < 0 || device >= ndev) { *err = BFIR_ERR_ARG; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *err = BFIR_ERR_HIP; return nullptr; }
    bfir_fft_plan *p = new bfir_fft_plan();
    p->device = device; p->order = order; p->invert = invert ? 1 : 0; p->realsize = realsize;
    p->n = 1L << order;
    const int lm = order - 1;                       // complex points
    if (order < 5) { p->lm1 = 0; p->lm2 = 0; }      // direct sums (k_small_*)
    else if (lm <= 12) { p->lm1 = lm; p->lm2 = 0; }
    else { p->lm1 = (lm + 1) / 2; p->lm2 = lm - p->lm1; }
    if (p->lm1 > 12 || (p->lm2 != 0 && p->lm2 < 4)) { *err = BFIR_ERR_UNSUPPORTED; delete p; return nullptr; }
    bool ok = order < 5 || fft_plan_create(&p->p1, 1 << p->lm1, realsize) == 0;
    if (ok && p->lm2) ok = fft_plan_create(&p->p2, 1 << p->lm2, realsize) == 0;
    ok = ok && hipMalloc(&p->d_a, (size_t)p->n * realsize) == hipSuccess;
    ok = ok && hipMalloc(&p->d_b, (size_t)p->n * realsize) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess;
    if (!ok) { *err = BFIR_ERR_HIP; bfir_fft_plan_destroy(p); return nullptr; }
    return p;
}

extern "C" void bfir_fft_plan_destroy(bfir_fft_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    fft_plan_destroy(&p->p1);
    fft_plan_destroy(&p->p2);
    if (p->d_a) (void)hipFree(p->d_a);
    if (p->d_b) (void)hipFree(p->d_b);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

// complex FFT of M = 2^(lm1+lm2) points from d_a; the result is returned in d_a
template <typename T> static void complex_fft(bfir_fft_plan *p, int sign)
{
    const int M1 = 1 << p->lm1, M2 = 1 << p->lm2;
    const long M = (long)M1 * M2;
    hipStream_t s = p->stream;
    if (p->lm2 == 0) {
        launch_rows<T>(p->lm1, sign, p->d_a, p->d_b, p->p1.tw, 1, M, 0, s);
        (void)hipMemcpyAsync(p->d_a, p->d_b, (size_t)M * 2 * sizeof(T), hipMemcpyDeviceToDevice, s);
        return;
    }
    launch_transpose<T>(p->d_a, p->d_b, M1, M2, s);                     // [M1][M2] -> [M2][M1]
    launch_rows<T>(p->lm1, sign, p->d_b, p->d_a, p->p1.tw, M2, M, 1, s);  // rows n2: FFT over n1, twiddle n2*k1
    launch_transpose<T>(p->d_a, p->d_b, M2, M1, s);                     // -> [M1][M2] (k1 major)
    launch_rows<T>(p->lm2, sign, p->d_b, p->d_a, p->p2.tw, M1, M, 0, s);  // rows k1: FFT over n2
    launch_transpose<T>(p->d_a, p->d_b, M1, M2, s);                     // -> [M2][M1]: index k1 + M1 k2
    (void)hipMemcpyAsync(p->d_a, p->d_b, (size_t)M * 2 * sizeof(T), hipMemcpyDeviceToDevice, s);
}

// FFTW_R2HC of the n reals in d_a -> half-complex in d_b (d_a is overwritten)
template <typename T> static void r2hc_dev(bfir_fft_plan *p)
{
    const int threads = 256, M = (int)(p->n >> 1);
    hipStream_t s = p->stream;
    if (p->order < 5) {
        hipLaunchKernelGGL(k_small_r2hc<T>, dim3(1), dim3(64), 0, s, (const T *)p->d_a, (T *)p->d_b, (int)p->n);
        return;
    }
    complex_fft<T>(p, -1);
    hipLaunchKernelGGL(k_z_to_hc<T>, dim3((M / 2 + 1 + threads - 1) / threads), dim3(threads), 0, s,
                       (const typename Vec2<T>::type *)p->d_a, (T *)p->d_b, (int)p->n);
}

// FFTW_HC2R (unnormalised) of the half-complex values in d_b -> n reals in d_a
template <typename T> static void hc2r_dev(bfir_fft_plan *p)
{
    const int threads = 256, M = (int)(p->n >> 1);
    hipStream_t s = p->stream;
    if (p->order < 5) {
        hipLaunchKernelGGL(k_small_hc2r<T>, dim3(1), dim3(64), 0, s, (const T *)p->d_b, (T *)p->d_a, (int)p->n);
        return;
    }
    hipLaunchKernelGGL(k_hc_to_z<T>, dim3((M + threads - 1) / threads), dim3(threads), 0, s, (const T *)p->d_b,
                       (typename Vec2<T>::type *)p->d_a, (int)p->n);
    complex_fft<T>(p, +1);                               // z[m] = (x[2m], x[2m+1])
}

template <typename T> static int execute_t(bfir_fft_plan *p, const void *in, void *out)
{
    const size_t bytes = (size_t)p->n * sizeof(T);
    hipStream_t s = p->stream;
    if (p->invert) {
        HIP_TRY(hipMemcpyAsync(p->d_b, in, bytes, hipMemcpyHostToDevice, s));
        hc2r_dev<T>(p);
        HIP_TRY(hipMemcpyAsync(out, p->d_a, bytes, hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(hipMemcpyAsync(p->d_a, in, bytes, hipMemcpyHostToDevice, s));
        r2hc_dev<T>(p);
        HIP_TRY(hipMemcpyAsync(out, p->d_b, bytes, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    return BFIR_OK;
}

// fftw[f]_execute_r2r(plan, in, out) on host buffers of 2^order reals (in == out allowed)
extern "C" int bfir_fft_plan_execute(bfir_fft_plan *p, const void *in, void *out)
{
    if (!p || !in || !out) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(p->device));
    return p->realsize == 4 ? execute_t<float>(p, in, out) : execute_t<double>(p, in, out);
}

extern "C" int64_t bfir_fft_plan_length(const bfir_fft_plan *p) { return p ? p->n : 0; }

// ---------------------------------------------------------------------------
// fftw_convolver::convolver_td_* (brutefir/fftw_convolver.hpp:157-166, fftw_convolver.cpp:697-777):
// the small one-block convolver the reference's (unused) delay class filters its sub-sample
// delays with (delay.cpp:174, 196, 240-257)
// ---------------------------------------------------------------------------
struct bfir_td_conv {
    bfir_fft_plan *plan = nullptr;   // 2 * blocklen reals; its two buffers carry both directions
    void *d_coeffs = nullptr;        // half-complex spectrum of [0 ... 0 | taps | 0 ...], times 1 / (2 blocklen)
    void *h_coeffs = nullptr;        // host copy: td_conv_t.coeffs
    int blocklen = 0;
};

// log2_roof (brutefir/log2.h:33-51) of n_coeffs as a power of two.  The reference's log2_roof(1) is -1 and it
// then shifts by that (undefined); here one tap is refused like zero taps.
extern "C" int bfir_td_block_length(int n_coeffs)
{
    if (n_coeffs < 2 || n_coeffs > (1 << 24)) return -1;
    int lg = 0;
    while ((1 << lg) < n_coeffs) lg++;
    return 1 << lg;
}

template <typename T> static int td_new_t(bfir_td_conv *t, const void *coeffs, int n_coeffs)
{
    bfir_fft_plan *p = t->plan;
    const size_t n = (size_t)p->n, bl = (size_t)t->blocklen;
    hipStream_t s = p->stream;
    HIP_TRY(hipMemsetAsync(p->d_a, 0, n * sizeof(T), s));
    HIP_TRY(hipMemcpyAsync((T *)p->d_a + bl, coeffs, (size_t)n_coeffs * sizeof(T), hipMemcpyHostToDevice, s));
    r2hc_dev<T>(p);
    const T sc = (T)(1.0 / (T)(t->blocklen << 1));
    hipLaunchKernelGGL(k_hc_scale<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (T *)p->d_b, sc, (long)n);
    HIP_TRY(hipMemcpyAsync(t->d_coeffs, p->d_b, n * sizeof(T), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(t->h_coeffs, p->d_b, n * sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    return BFIR_OK;
}

extern "C" void bfir_td_destroy(bfir_td_conv *t)
{
    if (!t) return;
    if (t->plan) {
        (void)hipSetDevice(t->plan->device);
        if (t->d_coeffs) { (void)hipDeviceSynchronize(); (void)hipFree(t->d_coeffs); }
        bfir_fft_plan_destroy(t->plan);
    }
    if (t->h_coeffs) bfir_aligned_free(t->h_coeffs);
    delete t;
}

extern "C" bfir_td_conv *bfir_td_new(const void *coeffs, int n_coeffs, int realsize, int device, int *err)
{
    int dummy;
    if (!err) err = &dummy;
    *err = BFIR_OK;
    const int blocklen = bfir_td_block_length(n_coeffs);
    if (!coeffs || blocklen < 0 || (realsize != 4 && realsize != 8)) { *err = BFIR_ERR_ARG; return nullptr; }
    int order = 1;
    while ((1 << order) < 2 * blocklen) order++;
    bfir_td_conv *t = new bfir_td_conv();
    t->blocklen = blocklen;
    t->plan = bfir_fft_plan_create(order, 0, 1, realsize, device, err);
    if (!t->plan) { delete t; return nullptr; }
    const size_t bytes = (size_t)2 * blocklen * realsize;
    t->h_coeffs = bfir_aligned_malloc(bytes, 16);
    if (!t->h_coeffs || hipMalloc(&t->d_coeffs, bytes) != hipSuccess) { *err = BFIR_ERR_HIP; bfir_td_destroy(t); return nullptr; }
    const int rc = realsize == 4 ? td_new_t<float>(t, coeffs, n_coeffs) : td_new_t<double>(t, coeffs, n_coeffs);
    if (rc != BFIR_OK) { *err = rc; bfir_td_destroy(t); return nullptr; }
    return t;
}

template <typename T> static int td_convolve_t(bfir_td_conv *t, void *overlap_block)
{
    bfir_fft_plan *p = t->plan;
    const size_t n = (size_t)p->n;
    hipStream_t s = p->stream;
    HIP_TRY(hipMemcpyAsync(p->d_a, overlap_block, n * sizeof(T), hipMemcpyHostToDevice, s));
    r2hc_dev<T>(p);
    hipLaunchKernelGGL(k_hc_mul_ordered<T>, dim3((unsigned)((n / 2 + 1 + 255) / 256)), dim3(256), 0, s, (T *)p->d_b,
                       (const T *)t->d_coeffs, (long)n);
    hc2r_dev<T>(p);
    HIP_TRY(hipMemcpyAsync(overlap_block, p->d_a, n * sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    return BFIR_OK;
}

// convolver_td_convolve: R2HC, ordered product with the filter's spectrum, HC2R, in place on the caller's
// 2 * blocklen reals
extern "C" int bfir_td_convolve(bfir_td_conv *t, void *overlap_block)
{
    if (!t || !overlap_block) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(t->plan->device));
    return t->plan->realsize == 4 ? td_convolve_t<float>(t, overlap_block) : td_convolve_t<double>(t, overlap_block);
}

extern "C" int bfir_td_blocklen(const bfir_td_conv *t) { return t ? t->blocklen : 0; }
extern "C" const void *bfir_td_coeffs(const bfir_td_conv *t) { return t ? t->h_coeffs : nullptr; }

// ---------------------------------------------------------------------------
// equalizer::render_f / render_d (brutefir/equalizer.cpp:211-299, 301-394)
// ---------------------------------------------------------------------------
namespace {

constexpr int EQ_MAX_BANDS = 40;   // BAND_COUNT + 2 = 33 in the reference (equalizer.hpp:14, 62-64)
struct EqBands { int count; double freq[EQ_MAX_BANDS], mag[EQ_MAX_BANDS], phase[EQ_MAX_BANDS]; };

// cosine_int_{f,d} (equalizer.cpp:182-204): T arguments, the arithmetic in double, a T result
template <typename T> __device__ __forceinline__ T cosine_int(T m1, T m2, T f1, T f2, T cur)
{
#pragma clang fp contract(off)
    return (T)((double)(m1 - m2) * 0.5 * cos(M_PI * (double)(cur - f1) / (double)(f2 - f1)) + (double)(m1 + m2) * 0.5);
}

// The half-complex spectrum the reference builds bin by bin (:238-259 / :328-349).  Every
// intermediate has the reference's type: with T = float the phase `rad` is a float, which is
// what quantises the rendered response there too.
template <typename T> __global__ void k_eq_fill(EqBands b, T *__restrict__ rbuf, int taps)
{
#pragma clang fp contract(off)
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = taps >> 1;
    if (n > half) return;
    const T scale = (T)(1.0 / (T)taps), divtaps = scale;
    const T tapspi = (T)(-(double)(T)taps * M_PI);
    if (n == 0) { rbuf[0] = (T)b.mag[0] * scale; return; }
    if (n == half) { rbuf[half] = (T)b.mag[b.count - 1] * scale; return; }
    const T cur = (T)n * divtaps;
    int i = 0;
    while (cur > (T)b.freq[i + 1]) i++;               // the reference's running index, found from scratch
    const T mag = cosine_int<T>((T)b.mag[i], (T)b.mag[i + 1], (T)b.freq[i], (T)b.freq[i + 1], cur) * scale;
    const T rad = tapspi * cur + cosine_int<T>((T)b.phase[i], (T)b.phase[i + 1], (T)b.freq[i], (T)b.freq[i + 1], cur);
    rbuf[n] = (T)(cos((double)rad) * (double)mag);
    rbuf[taps - n] = (T)(sin((double)rad) * (double)mag);
}

}  // namespace

// Render the smoothed band response to a taps/2-sample impulse: fill the half-complex
// spectrum, HC2R it with `ifftplan` (a plan of `taps` reals from bfir_fft_plan_create, as
// equalizer.cpp:54-55 obtains one), return the upper half (:274-282).  freq/mag/phase are the
// band tables AFTER equalizer::generate's conversion (:113-118): freq / sampling rate, linear
// magnitude, phase / (180 pi).
extern "C" int bfir_equalizer_render(bfir_fft_plan *ifftplan, int band_count, const double *freq, const double *mag,
                                     const double *phase, void *ir_out)
{
    if (!ifftplan || !ifftplan->invert || ifftplan->order < 5 || band_count < 2 || band_count > EQ_MAX_BANDS || !freq || !mag || !phase || !ir_out)
        return BFIR_ERR_ARG;
    bfir_fft_plan *p = ifftplan;
    HIP_TRY(hipSetDevice(p->device));
    EqBands b;
    b.count = band_count;
    for (int i = 0; i < band_count; i++) { b.freq[i] = freq[i]; b.mag[i] = mag[i]; b.phase[i] = phase[i]; }
    const int taps = (int)p->n, threads = 256, blocks = (taps / 2 + 1 + threads - 1) / threads;
    hipStream_t s = p->stream;
    if (p->realsize == 4) {
        hipLaunchKernelGGL(k_eq_fill<float>, dim3(blocks), dim3(threads), 0, s, b, (float *)p->d_b, taps);
        hipLaunchKernelGGL(k_hc_to_z<float>, dim3((taps / 2 + threads - 1) / threads), dim3(threads), 0, s,
                           (const float *)p->d_b, (float2 *)p->d_a, taps);
        complex_fft<float>(p, +1);
    } else {
        hipLaunchKernelGGL(k_eq_fill<double>, dim3(blocks), dim3(threads), 0, s, b, (double *)p->d_b, taps);
        hipLaunchKernelGGL(k_hc_to_z<double>, dim3((taps / 2 + threads - 1) / threads), dim3(threads), 0, s,
                           (const double *)p->d_b, (double2 *)p->d_a, taps);
        complex_fft<double>(p, +1);
    }
    const size_t half_bytes = (size_t)(taps / 2) * p->realsize;
    HIP_TRY(hipMemcpyAsync(ir_out, (char *)p->d_a + half_bytes, half_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    return BFIR_OK;
}
