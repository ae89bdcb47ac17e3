// dither.hip -- HP-TPDF dither on integer outputs.
//
// class dither (brutefir/dither.cpp): a table of int8 random numbers drawn once from a combined
// Tausworthe generator (:21-110, :419-449), per-channel positions in it 10 s of samples apart, and a
// quantiser with first-order high-pass error feedback (:141-194 float, :264-344 double) called per
// sample from real2raw{f,d}_hp_tpdf (brutefir/real2raw.cpp:38-317 and its double twin) after
// dither_preloop_real2int_hp_tpdf (:127-139) has placed the channel in the table.
//
// The recursion  v[n] += e[n-1] - e[n-2];  e[n] = v[n] - quantise(v[n] + dither[n])  has no parallel
// form (the quantiser sits inside it), so ONE LANE owns one channel and walks its samples in order;
// channels are independent and share a wave.  Integer outputs are off the measured path.
//
// Undefined in the reference, defined here: the map from byte
// differences to dither values has entries -256 .. 254 but a difference reaches +255, where the
// reference reads one element past its table; we continue the table's formula.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bfir_hip.h"
#include "kernels.h"

using namespace bfir;

void bfir_logf(const char *fmt, ...);

namespace bfir {

// ---- table (host) ------------------------------------------------------------------------------
namespace {
struct Taus {
    uint32_t s0, s1, s2;
    static uint32_t step(uint32_t s, int a, int b, uint32_t mask, int d) { return ((s & mask) << d) ^ (((s << a) ^ s) >> b); }
    uint32_t next()
    {
        s0 = step(s0, 13, 19, 0xFFFFFFFEu, 12);
        s1 = step(s1, 2, 25, 0xFFFFFFF8u, 4);
        s2 = step(s2, 3, 11, 0xFFFFFFF0u, 17);
        return s0 ^ s1 ^ s2;
    }
    explicit Taus(uint32_t seed)                  // tausinit, dither.cpp:437-449
    {
        if (!seed) seed = 1;
        s0 = 69069u * seed; s1 = 69069u * s0; s2 = 69069u * s1;
        for (int i = 0; i < 6; i++) (void)next();
    }
};
}  // namespace

int dither_spacing(int n_channels, int sample_rate, int max_size, int max_samples_per_loop)
{
    // dither.cpp:29-60
    int spacing = 10 * sample_rate;
    const int minspacing = sample_rate > max_samples_per_loop ? sample_rate : max_samples_per_loop;
    if (spacing < minspacing) spacing = minspacing;
    if (max_size > 0 && n_channels * spacing > max_size) spacing = max_size / n_channels;
    return spacing < minspacing ? -1 : spacing;
}

void dither_fill_table(std::vector<int8_t> &tab, int n_channels, int spacing)
{
    tab.resize((size_t)n_channels * spacing + 1);          // dither.cpp:62
    Taus g(0);
    for (auto &b : tab) b = (int8_t)(g.next() & 0xFFu);
}

// ---- device ------------------------------------------------------------------------------------
// dither_randmap[d] (dither.cpp:73-104), evaluated in double as there and narrowed to T
// The table entry is a rounded product added to a rounded sum: no fused multiply-add (the library is built with
// -ffp-contract=on, which would otherwise fuse this one expression; the reference is an SSE2 build, the CPU checker
// a gcc build without FMA).  Entry -256 is the table's exact -0.5 (unreachable from two int8 bytes, kept for the mirror).
template <typename T> __device__ __forceinline__ T dither_map(int d)
{
#pragma clang fp contract(off)
    if (d == 254) return (T)1.5;
    if (d <= -256) return (T)-0.5;
    const double prod = 1.0 / 255.0 * (double)d;
    return (T)((0.5 + 1.0 / 255.0) + prod);
}

template <typename T> struct DitherRun {
    T fb0, fb1;            // dither_state_t.sf / .sd
    double largest;        // bfoverflow_t.largest
    unsigned int n_over;
    int intlargest;
};

// One block: n samples at stride 1 from `src`, dither bytes tab[0..n) with `prev` = tab[-1].
template <typename T>
__device__ __forceinline__ void dither_block(DitherRun<T> &r, const T *__restrict__ src, unsigned char *__restrict__ raw,
                                             long raw_step, int n, const int8_t *__restrict__ tab, int prev, int bytes,
                                             bool be, int imin, int imax, T rmin, T rmax)
{
    for (int j = 0; j < n; j++) {
        const int cur = (int)tab[j];
        T v = src[j];
        v += r.fb0 - r.fb1;                                 // error feedback {1, -1}
        r.fb1 = r.fb0;
        const T dv = v + dither_map<T>(cur - prev);
        prev = cur;
        int s;
        if (dv < (T)0) {
            if (dv <= rmin) {
                s = imin; r.n_over++;
                if ((double)v < -r.largest) r.largest = (double)-dv;     // as written: tests v, stores dv
            } else {
                s = (int)dv - 1;
                if (s < -r.intlargest) r.intlargest = -s;
            }
        } else {
            if (dv > rmax) {
                s = imax; r.n_over++;
                if ((double)v > r.largest) r.largest = (double)dv;
            } else {
                s = (int)dv;
                if (s > r.intlargest) r.intlargest = s;
            }
        }
        r.fb0 = v - (T)s;
        const unsigned int u = (unsigned int)s;
        unsigned char *p = raw + (long)j * raw_step;
        for (int k = 0; k < bytes; k++) p[k] = (unsigned char)(u >> (8 * (be ? bytes - 1 - k : k)));
    }
}

template <typename T> __device__ __forceinline__ double largest_from_bits(unsigned long long b);
template <> __device__ __forceinline__ double largest_from_bits<float>(unsigned long long b) { return (double)__uint_as_float((unsigned int)b); }
template <> __device__ __forceinline__ double largest_from_bits<double>(unsigned long long b) { return __longlong_as_double((long long)b); }
template <typename T> __device__ __forceinline__ unsigned long long largest_to_bits(double v);
template <> __device__ __forceinline__ unsigned long long largest_to_bits<float>(double v) { return (unsigned long long)__float_as_uint((float)v); }
template <> __device__ __forceinline__ unsigned long long largest_to_bits<double>(double v) { return (unsigned long long)__double_as_longlong(v); }

// Engine form: lane gc walks blocks 0 .. n_blocks-1 of its channel, doing the preloop per block.
template <typename T>
__global__ __launch_bounds__(64) void k_stage_out_dither(StageOutArgs a, int bytes, bool be)
{
    const int gc = blockIdx.x * 64 + threadIdx.x;
    if (gc >= a.n_eng * a.C) return;
    const int e = gc / a.C, c = gc - e * a.C;
    const int bits = 8 * bytes;
    const int imin = (int)(0u - (1u << (bits - 1))), imax = (int)((1u << (bits - 1)) - 1u);
    const T rmin = (T)imin, rmax = (T)imax;
    DevDitherState *sp = (DevDitherState *)a.dither_state + gc;
    DevOverflow *of = a.overflow + gc;
    DitherRun<T> r;
    if constexpr (sizeof(T) == 4) { r.fb0 = sp->sf[0]; r.fb1 = sp->sf[1]; } else { r.fb0 = sp->sd[0]; r.fb1 = sp->sd[1]; }
    r.largest = largest_from_bits<T>(of->largest_bits);
    r.n_over = 0u; r.intlargest = of->intlargest;
    int ptr = sp->randtab_ptr;
    const int8_t *__restrict__ table = (const int8_t *)a.dither_tab;
    const T *__restrict__ src = (const T *)a.src + (long)gc * a.src_ch_stride;
    unsigned char *__restrict__ raw = (unsigned char *)a.raw + (long)e * a.eng_stride_bytes +
                                      (a.frame_off * a.spacing + c) * (long)bytes;
    const long raw_step = (long)a.spacing * bytes;
    const int L = a.L, n_blocks = (int)(a.n_frames / L);
    for (int t = 0; t < n_blocks; t++) {
        // brutefir.cpp:316-321: sample 0 of the block decides run()'s verdict
        if (!isfinite((double)src[(long)t * L])) flag_bad(a, t);
        // dither_preloop_real2int_hp_tpdf (dither.cpp:127-139).  The reference copies the last byte
        // used into table[0] on a wrap; only the wrapping channel ever reads it (as tab[-1] of this
        // very block), so the shared table stays read-only here and the byte is taken where it lies.
        int prev_idx = ptr - 1;
        if (ptr + L >= a.dither_size) ptr = 1;
        dither_block<T>(r, src + (long)t * L, raw + (long)t * L * raw_step, raw_step, L, table + ptr, (int)table[prev_idx],
                        bytes, be, imin, imax, rmin, rmax);
        ptr += L;
    }
    if constexpr (sizeof(T) == 4) { sp->sf[0] = r.fb0; sp->sf[1] = r.fb1; } else { sp->sd[0] = r.fb0; sp->sd[1] = r.fb1; }
    sp->randtab_ptr = ptr;
    of->largest_bits = largest_to_bits<T>(r.largest);
    of->n_overflows += r.n_over;
    of->intlargest = r.intlargest;
}

void launch_stage_out_dither(const StageOutArgs &a, hipStream_t s)
{
    const FmtInfo fi = fmt_info(a.fmt);
    const int gcs = a.n_eng * a.C;
    dim3 grid((gcs + 63) / 64), block(64);
    if (a.realsize == 4) hipLaunchKernelGGL(k_stage_out_dither<float>, grid, block, 0, s, a, fi.bytes, fi.big_endian);
    else hipLaunchKernelGGL(k_stage_out_dither<double>, grid, block, 0, s, a, fi.bytes, fi.big_endian);
}

// Stage form: one block of one channel, the preloop done by the host (bfir_dither_preloop...).
template <typename T>
__global__ void k_cbuf2raw_dither(const T *__restrict__ src, unsigned char *__restrict__ raw, long raw_step, int n,
                                  const int8_t *__restrict__ tab, int prev, int bytes, bool be, DitherRun<T> *state)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int bits = 8 * bytes;
    const int imin = (int)(0u - (1u << (bits - 1))), imax = (int)((1u << (bits - 1)) - 1u);
    DitherRun<T> r = *state;
    dither_block<T>(r, src, raw, raw_step, n, tab, prev, bytes, be, imin, imax, (T)imin, (T)imax);
    *state = r;
}

}  // namespace bfir

// ---- C ABI: class dither --------------------------------------------------------------------------
struct bfir_dither {
    int device = 0, realsize = 0, n_channels = 0, spacing = 0;
    std::vector<int8_t> tab;      // host copy; slot 0 is rewritten by the preloop exactly as in the reference
    int8_t *d_tab = nullptr;
    void *d_state = nullptr;      // one DitherRun<T>
};

extern "C" bfir_dither *bfir_dither_create(int n_channels, int sample_rate, int realsize, int max_size,
                                           int max_samples_per_loop, bfir_dither_state *dither_state, int device, int *err)
{
    int dummy;
    if (!err) err = &dummy;
    *err = BFIR_OK;
    if (n_channels < 1 || n_channels > BFIR_MAXCHANNELS || (realsize != 4 && realsize != 8) || !dither_state) {
        *err = BFIR_ERR_ARG; return nullptr;
    }
    const int spacing = dither_spacing(n_channels, sample_rate, max_size, max_samples_per_loop);
    if (spacing < 0) {                                     // dither.cpp:52-58
        bfir_logf("Maximum dither table size %d bytes is too small.\n", max_size);
        *err = BFIR_ERR_ARG; return nullptr;
    }
    if (bfir_device_count() <= 0) { *err = BFIR_ERR_NO_DEVICE; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *err = BFIR_ERR_HIP; return nullptr; }
    bfir_dither *d = new bfir_dither();
    d->device = device; d->realsize = realsize; d->n_channels = n_channels; d->spacing = spacing;
    dither_fill_table(d->tab, n_channels, spacing);
    bfir_logf("Dither table size is %d bytes.\nGenerating random numbers.", (int)d->tab.size());
    if (hipMalloc((void **)&d->d_tab, d->tab.size()) != hipSuccess ||
        hipMalloc(&d->d_state, sizeof(DitherRun<double>)) != hipSuccess ||
        hipMemcpy(d->d_tab, d->tab.data(), d->tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
        *err = BFIR_ERR_HIP; bfir_dither_destroy(d); return nullptr;
    }
    for (int n = 0; n < n_channels; n++) {                 // dither.cpp:105-109
        memset(&dither_state[n], 0, sizeof(bfir_dither_state));
        dither_state[n].randtab_ptr = n * spacing + 1;
    }
    return d;
}

extern "C" void bfir_dither_destroy(bfir_dither *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->d_tab) (void)hipFree(d->d_tab);
    if (d->d_state) (void)hipFree(d->d_state);
    delete d;
}

extern "C" int bfir_dither_table_size(const bfir_dither *d) { return d ? (int)d->tab.size() : 0; }
extern "C" const int8_t *bfir_dither_table(const bfir_dither *d) { return d ? d->tab.data() : nullptr; }

extern "C" void bfir_dither_preloop_real2int_hp_tpdf(bfir_dither *d, bfir_dither_state *state, int samples_per_loop)
{
    if (!d || !state) return;
    if (state->randtab_ptr + samples_per_loop >= (int)d->tab.size()) {     // dither.cpp:130-134
        d->tab[0] = d->tab[state->randtab_ptr - 1];
        state->randtab_ptr = 1;
    }
    state->randtab = d->tab.data() + state->randtab_ptr;
    state->randtab_ptr += samples_per_loop;
}

// real2raw{f,d}_hp_tpdf of one block (n samples of `d_src` on the device, working precision) into the
// strided device buffer d_raw; `state` and `overflow` are the caller's (host) structs.  The preloop
// must have been called for this block.  Used by bfir_convolver_cbuf2raw_dither (stage.hip).
int bfir_dither_realsize(const bfir_dither *d) { return d ? d->realsize : 0; }

int bfir_dither_run_block(bfir_dither *d, int realsize, const void *d_src, void *d_raw, int fmt, int spacing, int n,
                          bfir_dither_state *state, bfir_overflow *overflow, hipStream_t s)
{
    const FmtInfo fi = fmt_info(fmt);
    if (!fi.bytes || fi.isfloat || realsize != d->realsize) return BFIR_ERR_ARG;   // d_src holds `realsize`-byte reals
    const long off = state->randtab - d->tab.data();       // where the preloop put this block
    if (off < 1 || off + n > (long)d->tab.size()) return BFIR_ERR_ARG;
    const int prev = (int)d->tab[off - 1];                 // honours the host-side rewrite of slot 0
    auto run = [&](auto tag) -> int {
        using T = decltype(tag);
        DitherRun<T> r;
        if (sizeof(T) == 4) { r.fb0 = (T)state->sf[0]; r.fb1 = (T)state->sf[1]; } else { r.fb0 = (T)state->sd[0]; r.fb1 = (T)state->sd[1]; }
        r.largest = overflow->largest; r.n_over = 0u; r.intlargest = overflow->intlargest;
        if (hipMemcpyAsync(d->d_state, &r, sizeof(r), hipMemcpyHostToDevice, s) != hipSuccess) return BFIR_ERR_HIP;
        hipLaunchKernelGGL(k_cbuf2raw_dither<T>, dim3(1), dim3(64), 0, s, (const T *)d_src, (unsigned char *)d_raw,
                           (long)spacing * fi.bytes, n, (const int8_t *)d->d_tab + off, prev, fi.bytes, fi.big_endian,
                           (DitherRun<T> *)d->d_state);
        if (hipMemcpyAsync(&r, d->d_state, sizeof(r), hipMemcpyDeviceToHost, s) != hipSuccess) return BFIR_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return BFIR_ERR_HIP;
        if (sizeof(T) == 4) { state->sf[0] = (float)r.fb0; state->sf[1] = (float)r.fb1; } else { state->sd[0] = (double)r.fb0; state->sd[1] = (double)r.fb1; }
        overflow->largest = r.largest; overflow->n_overflows += r.n_over; overflow->intlargest = r.intlargest;
        return BFIR_OK;
    };
    return d->realsize == 4 ? run(float()) : run(double());
}
