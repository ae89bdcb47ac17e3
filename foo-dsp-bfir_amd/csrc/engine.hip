// engine.hip -- host side of the fused engine: bfir_engine_* of include/bfir_hip.h.
//
// One bfir_engine reproduces n_eng independent `brutefir` instances
// (brutefir/brutefir.hpp:15-128).  State the reference keeps in host memory
// (brutefir.cpp:738-810) lives in HBM for the life of the engine:
//   H     [GC][B][N]        partition spectra        (bfcoeff_t.data, coeff.cpp:292-354)
//   X     [GC][R][N]        delay line of spectra    (cbuf[n][B]; R = 2*chunk+B slots)
//   Y     [GC][chunk][N]    accumulated spectra      (ocbuf[n])
//   tin   2 x [GC][chunk*L] planar time input, double buffered (input_timecbuf)
//   tout  [GC][chunk*L]     planar time output
//   saved 2 x [GC][L]      history blocks kept across reset / reallocation
// GC = n_eng * C global channels.  A run of n blocks is cut into chunks of at
// most `chunk` blocks; each chunk is five launches, the first two on a side
// stream so that they overlap the last three of the chunk before.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/bfir_hip.h"
#include "kernels.h"

using namespace bfir;

// ---------------------------------------------------------------------------
// logging (pinfo.c:17-39 shape) and errors
// ---------------------------------------------------------------------------
static bfir_log_fn g_log = nullptr;

extern "C" void bfir_set_log_callback(bfir_log_fn fn) { g_log = fn; }

void bfir_logf(const char *fmt, ...)
{
    if (!g_log) return;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_log(buf);
}

extern "C" const char *bfir_strerror(int err)
{
    switch (err) {
    case BFIR_OK: return "ok";
    case BFIR_ERR_NONFINITE: return "NaN or Inf values in the system";
    case BFIR_ERR_COEFF: return "NaN or Inf value among coefficients";
    case BFIR_ERR_ARG: return "invalid argument";
    case BFIR_ERR_NO_DEVICE: return "no HIP device";
    case BFIR_ERR_HIP: return "HIP runtime error";
    case BFIR_ERR_STATE: return "engine not initialised";
    case BFIR_ERR_UNSUPPORTED: return "unsupported format or size";
    case BFIR_ERR_IO: return "file could not be opened";
    }
    return "unknown error";
}

extern "C" int bfir_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *bfir_version(void) { return "bfir-hip 0.1 (gfx950)"; }

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            bfir_logf("HIP error %s at %s:%d", hipGetErrorString(_e), __FILE__, __LINE__); \
            return BFIR_ERR_HIP;                                                   \
        }                                                                          \
    } while (0)

// brutefir::setup_sample_format (brutefir/brutefir.cpp:435-538): all eleven formats
static int fmt_bytes(int fmt) { return fmt_info(fmt).bytes; }

// A block of planar time samples somewhere in HBM: [GC][L] with a channel stride.
struct BlockRef {
    const void *ptr = nullptr;
    long ch_stride = 0;   // in reals
};

struct bfir_engine {
    int device = 0;
    int L = 0, N = 0, B = 0, s = 0, C = 0, n_eng = 1, GC = 0;
    int in_bytes = 0, out_bytes = 0, in_fmt = 0, out_fmt = 0;
    double in_scale = 1.0, out_scale = 1.0, of_max = 1.0;
    FftPlan plan;
    int chunk = 0, ring = 0;        // allocated geometry
    int want_chunk = 0;                    // blocks per launch (bfir_engine_set_chunk); 0 = automatic; buffers are sized lazily
    void *H = nullptr, *X = nullptr, *Y = nullptr, *tout = nullptr;
    void *tin[2] = {nullptr, nullptr};
    void *Yb[2] = {nullptr, nullptr};      // product spectra, one buffer per chunk parity
    void *saved[2] = {nullptr, nullptr};   // [GC][L] each: materialised history blocks
    // first halves of the reference's input_timecbuf[n][0/1] (brutefir.cpp:255-260):
    // where the block each of them holds currently lives
    BlockRef hist[2];
    int *d_nblk = nullptr;
    DevOverflow *d_of = nullptr;
    int *d_bad = nullptr;
    std::vector<int> nblk;          // host copy
    std::vector<char> eng_init;     // per engine: coefficients set
    unsigned long long blockcounter = 0;
    unsigned long long chunk_seq = 0;       // chunks queued since creation
    int curbuf = 0;
    // front (stage_in, fwd) runs on s_front, the MAC on s_mac, the back (inv, stage_out) on the
    // caller's stream, so fwd(k+1), mac(k) and inv/stage_out(k-1) are on the GPU together
    hipStream_t stream = nullptr, s_front = nullptr, s_mac = nullptr, s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_entry = nullptr, ev_fwd[2] = {nullptr, nullptr}, ev_mac[2] = {nullptr, nullptr};
    hipEvent_t ev_inv[2] = {nullptr, nullptr};
    bool pipe3 = true;                     // BFIR_PIPE=2: MAC on the caller's stream (two-stage schedule)
    // spectra (X, H, Y) as (re, im) pairs instead of the reference's 4 re | 4 im groups: the layout
    // of the fp32 fast MAC kernels; chosen once per engine (N >= 512, fp32)
    bool ilv = false;
    // pair path (pair.hip): FLOAT_LE in and out, even channel count, 512 <= L <= 8192 on top of ilv.
    // No planar time buffers; the engine's time history is the raw frames of the last two blocks,
    // tails[set][i] = [n_eng][L][C] floats, set alternating per chunk so a launch never reads and
    // writes the same copy; hist_raw[i] is where input_timecbuf[n][i]'s first half currently lives.
    bool pair = false;
    // ... with an odd channel count (or one channel) the pairs are blocks t, t + 1 of ONE channel (k_fwd_tp_ps / k_inv_tp_ps)
    bool pair_tp = false;
    // direct path: any other engine whose frames are FLOAT_LE / FLOAT64_LE in and out (fp64 arithmetic, odd
    // channel counts, partitions outside the pair kernels' range): k_fwd reads the raw frames itself and
    // k_inv writes them, one channel per transform; same history bookkeeping as the pair path (tails of raw
    // input frames), no staging kernels, no planar time buffers.  BFIR_DIRECT=0 / BFIR_PAIR=0 (tuning aids,
    // tests) keep the staging kernels.
    bool direct = false;
    FftPlan plan2;                         // transform of 2L complex points
    float *tails[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    const float *hist_raw[2] = {nullptr, nullptr};
    // HP-TPDF dither (integer output + apply_dither; dither.hip): the reference instance's random table
    // (every engine of a batch would build the same one) and dither_state_t per global channel
    int8_t *d_dither_tab = nullptr; int dither_size = 0;
    DevDitherState *d_dither_state = nullptr;
    // latency path (bfir_engine_run with a handful of blocks, the plug-in's one run() per block): everything
    // on one stream, no events, kernels read and write the pinned staging buffers across the host link
    bool inline_launch = false;            // set around run_chunk by run_small()
    bool async_pending = false;            // run_device queued work that nobody has waited for yet
    int *h_bad = nullptr;                  // pinned: one flag per block of a latency-path call (flag_bad, kernels.h)
    int *bad_host_cur = nullptr;           // ... what the kernels of the chunk being launched are given (null outside run_small)
    bool serial = false;                   // BFIR_PIPE=1: everything on the caller's stream (kernel timing runs)
    // host-pointer path: pinned + device staging, double buffered
    void *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr};
    void *dev_in[2] = {nullptr, nullptr}, *dev_out[2] = {nullptr, nullptr};
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
    size_t stage_bytes_in = 0, stage_bytes_out = 0;
    // profiling
    bool profiling = false;
    struct Span { int k; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[BFIR_K_COUNT] = {0, 0, 0, 0, 0};
    long prof_n[BFIR_K_COUNT] = {0, 0, 0, 0, 0};
};

static size_t cbuf_bytes(const bfir_engine *e) { return (size_t)e->N * (size_t)e->s; }

// Copy the two history blocks into the engine-owned `saved` buffers so they
// survive the work buffers they may point into.  A reference only ever points
// at its own saved[i] or into a time buffer, never at the other saved buffer.
static int materialise_history(bfir_engine *e)
{
    HIP_TRY(hipDeviceSynchronize());
    const size_t Ls = (size_t)e->L * e->s;
    for (int i = 0; i < 2; i++) {
        if (e->hist[i].ptr != e->saved[i])
            HIP_TRY(hipMemcpy2D(e->saved[i], Ls, e->hist[i].ptr, (size_t)e->hist[i].ch_stride * e->s, Ls, e->GC,
                                hipMemcpyDeviceToDevice));
        e->hist[i].ptr = e->saved[i];
        e->hist[i].ch_stride = e->L;
    }
    HIP_TRY(hipDeviceSynchronize());
    return BFIR_OK;
}

static void free_work(bfir_engine *e)
{
    void **bufs[] = {&e->X, &e->Yb[0], &e->Yb[1], &e->tin[0], &e->tin[1], &e->tout};
    for (void **b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
    for (int i = 0; i < 2; i++) {
        if (e->pin_in[i]) (void)hipHostFree(e->pin_in[i]);
        if (e->pin_out[i]) (void)hipHostFree(e->pin_out[i]);
        if (e->dev_in[i]) (void)hipFree(e->dev_in[i]);
        if (e->dev_out[i]) (void)hipFree(e->dev_out[i]);
        e->pin_in[i] = e->pin_out[i] = e->dev_in[i] = e->dev_out[i] = nullptr;
    }
    e->stage_bytes_in = e->stage_bytes_out = 0;
    e->chunk = e->ring = 0;
    if (e->h_bad) { (void)hipHostFree(e->h_bad); e->h_bad = nullptr; }
}

// (Re)allocate the chunk-sized work buffers.  The delay line is carried over
// slot by slot when the ring size changes; the time history is moved into
// `saved` first because it may live in the buffers being freed.
static int alloc_work(bfir_engine *e, int chunk)
{
    const size_t cb = cbuf_bytes(e);
    // fwd of chunk k+1 may run while mac of chunk k still reads its B-1 older slots
    const int ring = 2 * chunk + e->B;
    if (e->X) { int rc = materialise_history(e); if (rc != BFIR_OK) return rc; }
    void *X = nullptr, *Y0 = nullptr, *Y1 = nullptr, *tin0 = nullptr, *tin1 = nullptr, *tout = nullptr;
    HIP_TRY(hipMalloc(&X, (size_t)e->GC * ring * cb));
    HIP_TRY(hipMalloc(&Y0, (size_t)e->GC * chunk * cb));
    HIP_TRY(hipMalloc(&Y1, (size_t)e->GC * chunk * cb));
    if (!e->pair && !e->direct) {   // the pair and direct paths have no planar time buffers
        HIP_TRY(hipMalloc(&tin0, (size_t)e->GC * chunk * e->L * e->s));
        HIP_TRY(hipMalloc(&tin1, (size_t)e->GC * chunk * e->L * e->s));
        HIP_TRY(hipMalloc(&tout, (size_t)e->GC * chunk * e->L * e->s));
    }
    HIP_TRY(hipMemset(X, 0, (size_t)e->GC * ring * cb));
    if (e->X) {
        // keep the last B-1 spectra: absolute block j lives in slot j % ring
        const int keep = (int)std::min<unsigned long long>(e->blockcounter, (unsigned long long)(e->B - 1));
        for (int d = 1; d <= keep; d++) {
            const unsigned long long j = e->blockcounter - d;
            const size_t so = (size_t)(j % e->ring) * cb, dn = (size_t)(j % ring) * cb;
            HIP_TRY(hipMemcpy2D((char *)X + dn, (size_t)ring * cb, (char *)e->X + so, (size_t)e->ring * cb, cb,
                                e->GC, hipMemcpyDeviceToDevice));
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    free_work(e);
    e->X = X; e->Yb[0] = Y0; e->Yb[1] = Y1; e->tin[0] = tin0; e->tin[1] = tin1; e->tout = tout;
    e->chunk = chunk; e->ring = ring;
    return BFIR_OK;
}

extern "C" bfir_engine *bfir_engine_create_batch(int n_engines, int filter_length, int filter_blocks,
                                                 int realsize, int channels, int in_format,
                                                 int out_format, int sampling_rate, int apply_dither,
                                                 int device, int *err)
{
    int dummy;
    if (!err) err = &dummy;
    *err = BFIR_OK;
    // brutefir.cpp:652 (channel limit), fftw_convolver.cpp:64-74 (realsize, length)
    if (channels < 1 || channels > BFIR_MAXCHANNELS) {
        bfir_logf("Number of channels (%d) exceeds limit (%d).", channels, BFIR_MAXCHANNELS);
        *err = BFIR_ERR_ARG; return nullptr;
    }
    if (realsize != 4 && realsize != 8) { bfir_logf("Invalid real size %d.", realsize); *err = BFIR_ERR_ARG; return nullptr; }
    if (filter_length < 1 || (filter_length & (filter_length - 1))) {
        bfir_logf("Invalid length %d.", filter_length); *err = BFIR_ERR_ARG; return nullptr;
    }
    if (filter_blocks < 1 || n_engines < 1) { *err = BFIR_ERR_ARG; return nullptr; }
    if (!fmt_bytes(in_format) || !fmt_bytes(out_format)) { *err = BFIR_ERR_UNSUPPORTED; return nullptr; }
    int ndev = bfir_device_count();
    if (ndev <= 0) { *err = BFIR_ERR_NO_DEVICE; return nullptr; }
    if (device < 0 || device >= ndev) { *err = BFIR_ERR_ARG; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *err = BFIR_ERR_HIP; return nullptr; }

    bfir_engine *e = new bfir_engine();
    e->device = device;
    e->L = filter_length; e->N = 2 * filter_length; e->B = filter_blocks; e->s = realsize;
    e->C = channels; e->n_eng = n_engines; e->GC = n_engines * channels;
    e->in_bytes = fmt_bytes(in_format); e->out_bytes = fmt_bytes(out_format);
    e->in_fmt = in_format; e->out_fmt = out_format;
    // setup_input: normalised scale 1/2^(bits-1); setup_output: full scale; overflow max
    // 2^(bits-1)-1 for integers, 1.0 for floats (brutefir.cpp:395-420, 546-582, 672-684)
    e->in_scale = fmt_info(in_format).isfloat ? 1.0 : 1.0 / fmt_full_scale(in_format);
    e->out_scale = fmt_info(out_format).isfloat ? 1.0 : fmt_full_scale(out_format);
    e->of_max = fmt_info(out_format).isfloat ? 1.0 : fmt_full_scale(out_format) - 1.0;
    {   // BFIR_MAC_VARIANT != 0 (tuning aid) keeps the grouped layout and the other MAC kernels
        const char *mv = getenv("BFIR_MAC_VARIANT");
        e->ilv = realsize == 4 && e->N >= 512 && !(mv && atoi(mv) != 0);
    }
    {   // BFIR_PAIR=0 (tuning aid) keeps the planar staging kernels
        const char *pv = getenv("BFIR_PAIR");
        // the persistent kernels only (BFIR_PAIR_PERSIST=0 keeps odd channel counts on the general path)
        const char *pp = getenv("BFIR_PAIR_PERSIST"), *tv = getenv("BFIR_PAIR_TIME");
        const bool tp_ok = !(pp && atoi(pp) != 1) && !(tv && atoi(tv) == 0);
        e->pair_tp = (channels % 2) == 1 && tp_ok;
        e->pair = e->ilv && in_format == 8 && out_format == 8 && ((channels % 2) == 0 || e->pair_tp) &&
                  pair_supported(filter_length) && !(pv && atoi(pv) == 0);
        e->pair_tp = e->pair_tp && e->pair;
        const char *dv = getenv("BFIR_DIRECT");
        // worth it where a channel's samples are 8 bytes apart or wider units: FLOAT64 frames (any C), or one
        // channel (contiguous samples), or float frames with an even channel count (the reference plug-in's own shape:
        // fp64 arithmetic, float32 frames), which k_fwd / k_inv move a channel PAIR at a time with both channels in one
        // workgroup (stereo: two whole frames per lane; wider frames since round 3: 33.5 -> ~40 Gsamples/s at 4-8 channels).  Other 4-byte samples at a stride
        // (float frames, C > 2) are faster through the staging kernels (profiles/r02_other_configs.txt: one
        // channel per workgroup ran the plug-in's shape at 11.4 instead of 20.6 Gsamples/s).  BFIR_DIRECT=1
        // forces it (tests).
        const bool stereo = e->in_bytes == 4 && e->out_bytes == 4 && (channels % 2) == 0 && !e->ilv &&
                            direct_stereo_supported(filter_length, realsize);
        // ... and, since round 3, any float / double frames of an fp64 engine whose transform the run kernels take (k_fwd_run /
        // k_inv_run hide the strided loads under the transform: 3 / 5 channels of float32 frames 33 -> 41-42 Gsamples/s)
        const bool wide = (e->in_bytes == 8 && e->out_bytes == 8) || channels == 1 || stereo || run64_supported(filter_length, realsize);
        e->direct = !e->pair && fmt_is_native(in_format) && fmt_is_native(out_format) && !(pv && atoi(pv) == 0) &&
                    (dv ? atoi(dv) != 0 : wide);
        // fp64 engines whose transforms the run kernels take keep their spectra -- delay line, filter partitions, products --
        // as (re, im) PAIRS like the fp32 engines, not in the reference's groups of four: one 16-byte access per bin in the
        // MAC instead of two of 8, the forward kernel's spectrum straight from registers (no LDS staging), conflict-free reads
        // in the inverse.  Only where every kernel on the engine's way reads pairs: direct mode, the systolic MAC (up to 256
        // partitions) -- so the switches that pick other kernels keep the groups (all read HERE, at creation, for such engines).
        // BFIR_F64_PAIRS=0: off (A/B).  Same arithmetic either way: the same bits.
        {
            const char *fp = getenv("BFIR_F64_PAIRS"), *ms = getenv("BFIR_MAC_SYS"), *mv = getenv("BFIR_MAC_VARIANT");
            const bool other_mac = (ms && atoi(ms) == 0) || getenv("BFIR_MAC64_VARIANT") || getenv("BFIR_MAC_BATCHED") || (mv && atoi(mv) != 0);
            if (realsize == 8 && e->direct && pairs64_supported(filter_length, realsize) && filter_blocks <= BFIR_MAC_SYS_MAX_B && !other_mac &&
                !(fp && atoi(fp) == 0))
                e->ilv = true;
        }
    }
    if (const char *pm = getenv("BFIR_PIPE")) { e->pipe3 = atoi(pm) >= 3; e->serial = atoi(pm) == 1; }
    // fp64 engines: one stream.  Their kernels are bound by issue and latency, not by memory, each fills the GPU by itself, and
    // three of them side by side only get into each other's way: the plug-in's shape 42.8 -> 45.9 Gsamples/s, 8 channels 42.5 ->
    // 44.6, cfg5 43.1 either way (profiles/r03_fp64.txt).  The fp32 headline gains 10 % from the three-stream schedule.
    else if (realsize == 8) { e->pipe3 = false; e->serial = true; }
    e->nblk.assign(e->GC, 0);
    e->eng_init.assign(n_engines, 0);
    int rc = fft_plan_create(&e->plan, filter_length, realsize);
    if (rc != 0) { *err = (rc == -1) ? BFIR_ERR_UNSUPPORTED : BFIR_ERR_HIP; delete e; return nullptr; }
    auto fail = [&](int code) { *err = code; bfir_engine_destroy(e); return (bfir_engine *)nullptr; };
    if (e->pair || e->direct) {
        if (e->pair && fft_plan_create(&e->plan2, 2 * filter_length, 4) != 0) return fail(BFIR_ERR_HIP);
        const size_t tb = (size_t)e->n_eng * e->L * e->C * e->in_bytes;     // raw input frames of one block
        for (int st = 0; st < 2; st++)
            for (int i = 0; i < 2; i++) {
                if (hipMalloc((void **)&e->tails[st][i], tb) != hipSuccess) return fail(BFIR_ERR_HIP);
                (void)hipMemset(e->tails[st][i], 0, tb);   // input_timecbuf starts zeroed (brutefir.cpp:769)
            }
        // chunk 0 writes set 0, so the (zero) history it reads lives in set 1
        e->hist_raw[0] = e->tails[1][0]; e->hist_raw[1] = e->tails[1][1];
    }
    hipStream_t *streams[] = {&e->stream, &e->s_front, &e->s_mac, &e->s_in, &e->s_out};
    for (hipStream_t *st : streams)
        if (hipStreamCreateWithFlags(st, hipStreamNonBlocking) != hipSuccess) return fail(BFIR_ERR_HIP);
    hipEvent_t *events[] = {&e->ev_entry, &e->ev_fwd[0], &e->ev_fwd[1], &e->ev_mac[0], &e->ev_mac[1],
                            &e->ev_inv[0], &e->ev_inv[1],
                            &e->ev_h2d[0], &e->ev_h2d[1], &e->ev_comp[0], &e->ev_comp[1], &e->ev_d2h[0], &e->ev_d2h[1]};
    for (hipEvent_t *ev : events)
        if (hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) return fail(BFIR_ERR_HIP);
    const size_t cb = cbuf_bytes(e), Ls = (size_t)e->L * e->s;
    if (hipMalloc(&e->H, (size_t)e->GC * e->B * cb) != hipSuccess ||
        hipMalloc(&e->saved[0], (size_t)e->GC * Ls) != hipSuccess ||
        hipMalloc(&e->saved[1], (size_t)e->GC * Ls) != hipSuccess ||
        hipMalloc((void **)&e->d_nblk, sizeof(int) * e->GC) != hipSuccess ||
        hipMalloc((void **)&e->d_of, sizeof(DevOverflow) * e->GC * BFIR_OF_SHARDS) != hipSuccess ||
        hipMalloc((void **)&e->d_bad, sizeof(int)) != hipSuccess)
        return fail(BFIR_ERR_HIP);
    (void)hipMemset(e->H, 0, (size_t)e->GC * e->B * cb);
    for (int i = 0; i < 2; i++) {   // input_timecbuf starts zeroed (brutefir.cpp:769)
        (void)hipMemset(e->saved[i], 0, (size_t)e->GC * Ls);
        e->hist[i].ptr = e->saved[i]; e->hist[i].ch_stride = e->L;
    }
    (void)hipMemset(e->d_nblk, 0, sizeof(int) * e->GC);
    (void)hipMemset(e->d_of, 0, sizeof(DevOverflow) * e->GC * BFIR_OF_SHARDS);
    (void)hipMemset(e->d_bad, 0x7f, sizeof(int));
    if (apply_dither && !fmt_info(out_format).isfloat) {
        // dither::dither(n_channels, sampling_rate, realsize, max_dither_table_size = 0, filter_length, state)
        // as brutefir::init_convolver builds it (brutefir.cpp:709-714)
        const int spacing = dither_spacing(channels, sampling_rate, 0, filter_length);
        if (spacing < 0) return fail(BFIR_ERR_ARG);
        std::vector<int8_t> tab;
        dither_fill_table(tab, channels, spacing);
        std::vector<DevDitherState> st((size_t)e->GC);
        for (int gc = 0; gc < e->GC; gc++) { memset(&st[gc], 0, sizeof(DevDitherState)); st[gc].randtab_ptr = (gc % channels) * spacing + 1; }
        if (hipMalloc((void **)&e->d_dither_tab, tab.size()) != hipSuccess ||
            hipMalloc((void **)&e->d_dither_state, st.size() * sizeof(DevDitherState)) != hipSuccess ||
            hipMemcpy(e->d_dither_tab, tab.data(), tab.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e->d_dither_state, st.data(), st.size() * sizeof(DevDitherState), hipMemcpyHostToDevice) != hipSuccess)
            return fail(BFIR_ERR_HIP);
        e->dither_size = (int)tab.size();
        bfir_logf("Dither table size is %d bytes.", e->dither_size);
    }
    if (alloc_work(e, 1) != BFIR_OK) return fail(BFIR_ERR_HIP);
    if (hipDeviceSynchronize() != hipSuccess) return fail(BFIR_ERR_HIP);
    bfir_logf("bfir engine: %d x %d channels, partition %d, %d blocks, realsize %d on device %d.",
              n_engines, channels, filter_length, filter_blocks, realsize, device);
    return e;
}

extern "C" bfir_engine *bfir_engine_create(int filter_length, int filter_blocks, int realsize,
                                           int channels, int in_format, int out_format,
                                           int sampling_rate, int apply_dither, int device, int *err)
{
    return bfir_engine_create_batch(1, filter_length, filter_blocks, realsize, channels, in_format,
                                    out_format, sampling_rate, apply_dither, device, err);
}

extern "C" void bfir_engine_destroy(bfir_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    free_work(e);
    fft_plan_destroy(&e->plan);
    fft_plan_destroy(&e->plan2);
    for (int st = 0; st < 2; st++) for (int i = 0; i < 2; i++) if (e->tails[st][i]) (void)hipFree(e->tails[st][i]);
    void *bufs[] = {e->H, e->saved[0], e->saved[1], e->d_nblk, e->d_of, e->d_bad, e->d_dither_tab, e->d_dither_state};
    for (void *b : bufs) if (b) (void)hipFree(b);
    for (auto &sp : e->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    hipEvent_t events[] = {e->ev_entry, e->ev_fwd[0], e->ev_fwd[1], e->ev_mac[0], e->ev_mac[1], e->ev_inv[0], e->ev_inv[1],
                           e->ev_h2d[0], e->ev_h2d[1], e->ev_comp[0], e->ev_comp[1], e->ev_d2h[0], e->ev_d2h[1]};
    for (hipEvent_t ev : events) if (ev) (void)hipEventDestroy(ev);
    hipStream_t streams[] = {e->stream, e->s_front, e->s_mac, e->s_in, e->s_out};
    for (hipStream_t st : streams) if (st) (void)hipStreamDestroy(st);
    delete e;
}

extern "C" int bfir_engine_is_initialized(const bfir_engine *e)
{
    if (!e) return 0;
    for (char c : e->eng_init) if (!c) return 0;
    return 1;
}

extern "C" int bfir_engine_set_chunk(bfir_engine *e, int blocks_per_launch)
{
    if (!e || blocks_per_launch < 0) return BFIR_ERR_ARG;   // 0: automatic (the default)
    e->want_chunk = blocks_per_launch;
    return BFIR_OK;
}

// coeff::preprocess_coeff + convolver_coeffs2cbuf for the C channels of one engine.
extern "C" int bfir_engine_set_coeff_at(bfir_engine *e, int engine_index, const void *const *coeffs,
                                        int n_coeffs, int length, int coeff_blocks, double scale)
{
    if (!e || engine_index < 0 || engine_index >= e->n_eng || !coeffs || length < 0 || coeff_blocks < 1)
        return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    e->eng_init[engine_index] = 0;                              // free_coeff(), brutefir.cpp:188
    if (n_coeffs > e->C) n_coeffs = e->C;                       // brutefir.cpp:190-193
    const int nb = std::min(coeff_blocks, e->B);                // run() never looks past B blocks
    const size_t taps_pad = (size_t)nb * e->L;
    const size_t cb = cbuf_bytes(e);
    const int gc0 = engine_index * e->C;
    // zero padded impulse per channel; a block past the end is all zero
    // (coeff.cpp:315-339), taps are scaled in working precision (fftw_convolver.cpp:491,507)
    std::vector<char> host((size_t)e->C * taps_pad * e->s, 0);
    for (int n = 0; n < n_coeffs; n++) {
        if (!coeffs[n]) return BFIR_ERR_ARG;
        const size_t cnt = std::min((size_t)length, taps_pad);
        bool finite = true;
        if (e->s == 4) {
            const float *src = (const float *)coeffs[n];
            const float sc = (float)scale;
            for (size_t i = 0; i < cnt; i++) finite &= std::isfinite((double)(src[i] * sc));
        } else {
            const double *src = (const double *)coeffs[n];
            for (size_t i = 0; i < cnt; i++) finite &= std::isfinite(src[i] * scale);
        }
        if (!finite) {
            bfir_logf("NaN or Inf value among coefficients.");
            bfir_logf("Error preprocessing coefficient %d", n);
            return BFIR_ERR_COEFF;
        }
        memcpy(host.data() + (size_t)n * taps_pad * e->s, coeffs[n], cnt * e->s);
    }
    void *d_taps = nullptr;
    HIP_TRY(hipMalloc(&d_taps, host.size()));
    HIP_TRY(hipMemcpyAsync(d_taps, host.data(), host.size(), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync((char *)e->H + (size_t)gc0 * e->B * cb, 0, (size_t)e->C * e->B * cb, e->stream));
    FwdArgs fa;
    // window of block b = [L zeros | taps b*L .. b*L+L)
    fa.src = d_taps; fa.src_ch_stride = (long)taps_pad;
    fa.prev = nullptr; fa.prev_ch_stride = 0;
    fa.dst = (char *)e->H + (size_t)gc0 * e->B * cb;
    fa.dst_ch_stride = (long)e->B * e->N;
    fa.ring = e->B; fa.base_slot = 0;
    fa.n_t = nb; fa.n_ch = e->C;
    fa.load_scale = scale;
    fa.out_scale = 1.0 / (double)e->N;                          // fftw_convolver.cpp:520
    fa.zero_first_half = 1;
    fa.interleaved = e->ilv;
    launch_fwd(e->plan, fa, e->stream);
    for (int n = 0; n < e->C; n++) e->nblk[gc0 + n] = nb;
    HIP_TRY(hipMemcpyAsync(e->d_nblk + gc0, e->nblk.data() + gc0, sizeof(int) * e->C,
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipFree(d_taps));
    HIP_TRY(hipGetLastError());
    e->eng_init[engine_index] = 1;
    return BFIR_OK;
}

extern "C" int bfir_engine_set_coeff(bfir_engine *e, const void *const *coeffs, int n_coeffs,
                                     int length, int coeff_blocks, double scale)
{
    return bfir_engine_set_coeff_at(e, 0, coeffs, n_coeffs, length, coeff_blocks, scale);
}

extern "C" int bfir_engine_read_coeff(bfir_engine *e, int channel, int block, void *dst)
{
    if (!e || channel < 0 || channel >= e->GC || block < 0 || block >= e->B || !dst) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t cb = cbuf_bytes(e);
    HIP_TRY(hipMemcpy(dst, (char *)e->H + ((size_t)channel * e->B + block) * cb, cb, hipMemcpyDeviceToHost));
    if (e->ilv) {   // hand out the reference's grouped layout (fftw_convolver.cpp:883-907)
        auto regroup = [&](auto *o) {
            using R = typename std::remove_pointer<decltype(o)>::type;
            std::vector<R> tmp(o, o + e->N);
            for (int k = 0; k < e->N / 2; k++) {
                o[8 * (k >> 2) + (k & 3)] = tmp[2 * k];
                o[8 * (k >> 2) + 4 + (k & 3)] = tmp[2 * k + 1];
            }
        };
        if (e->s == 4) regroup((float *)dst); else regroup((double *)dst);
    }
    return BFIR_OK;
}

// ---------------------------------------------------------------------------
// profiling helpers
// ---------------------------------------------------------------------------
static hipEvent_t take_event(bfir_engine *e)
{
    if (!e->ev_pool.empty()) { hipEvent_t ev = e->ev_pool.back(); e->ev_pool.pop_back(); return ev; }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

struct ProfScope {
    bfir_engine *e; int k; hipStream_t st; hipEvent_t a = nullptr;
    ProfScope(bfir_engine *e_, int k_, hipStream_t st_) : e(e_), k(k_), st(st_)
    {
        if (e->profiling) { a = take_event(e); (void)hipEventRecord(a, st); }
    }
    ~ProfScope()
    {
        if (e->profiling) {
            hipEvent_t b = take_event(e);
            (void)hipEventRecord(b, st);
            e->spans.push_back({k, a, b});
        }
    }
};

static void drain_spans(bfir_engine *e)
{
    for (auto &sp : e->spans) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            e->prof_ms[sp.k] += ms;
            e->prof_n[sp.k] += 1;
        }
        e->ev_pool.push_back(sp.a);
        e->ev_pool.push_back(sp.b);
    }
    e->spans.clear();
}

extern "C" int bfir_engine_set_profiling(bfir_engine *e, int enable)
{
    if (!e) return BFIR_ERR_ARG;
    drain_spans(e);
    e->profiling = enable != 0;
    if (e->profiling && e->ev_pool.size() < 4096) {   // keep event creation out of timed regions
        for (int i = 0; i < 4096; i++) { hipEvent_t ev = nullptr; if (hipEventCreate(&ev) == hipSuccess) e->ev_pool.push_back(ev); }
    }
    for (int k = 0; k < BFIR_K_COUNT; k++) { e->prof_ms[k] = 0; e->prof_n[k] = 0; }
    return BFIR_OK;
}

extern "C" int bfir_engine_get_profile(bfir_engine *e, int kernel, double *total_ms, int64_t *launches)
{
    if (!e || kernel < 0 || kernel >= BFIR_K_COUNT) return BFIR_ERR_ARG;
    drain_spans(e);
    if (total_ms) *total_ms = e->prof_ms[kernel];
    if (launches) *launches = e->prof_n[kernel];
    return BFIR_OK;
}

// ---------------------------------------------------------------------------
// brutefir::run, chunked and software-pipelined over three streams
// ---------------------------------------------------------------------------
// Queue one chunk of tc blocks (frames frame_off .. of every engine's raw
// buffer).  `st` is the caller's stream: the input must be ready on it when
// this is called, and the output is complete on it when its work is.
//   s_front : stage_in(k) -> fwd(k)                      (k = chunk sequence number)
//   s_mac   : mac(k)                  waits fwd(k) and inv(k-2) (owner of Yb[k&1])
//   st      : inv(k) -> stage_out(k)  waits mac(k)
// fwd(k) writes delay-line slots that mac(k-2) may still read (ring = 2*chunk+B),
// and stage_in(k) rewrites the time buffer fwd(k-2) read; both are ordered by
// events / stream order.  So fwd(k+1), mac(k) and inv/stage_out(k-1) share the
// GPU: none of the kernels saturates VALU or HBM alone (load-latency phases,
// profiles/r01_phase_trace.txt), together they fill each other's gaps.
// The same chunk on the pair path: no staging kernels, no planar time buffers.
//   s_front : fwd_pair(k)             raw frames -> delay-line spectra of two channels per transform
//   s_mac   : mac(k)
//   st      : inv_pair(k)             product spectra -> raw frames + overflow statistics
static int run_chunk_pair(bfir_engine *e, const void *d_in, long in_stride, void *d_out, long out_stride,
                          long frame_off, int tc, int block_base, hipStream_t st, hipEvent_t input_ready)
{
    if (((uintptr_t)d_in | (uintptr_t)d_out | (uintptr_t)in_stride | (uintptr_t)out_stride) & (e->pair_tp ? 3 : 7)) {
        bfir_logf("bfir engine: frame buffers of the float fast path must be 8-byte aligned (4 with an odd channel count).");
        return BFIR_ERR_ARG;
    }
    const int par = (int)(e->chunk_seq & 1);
    const int base_slot = (int)(e->blockcounter % (unsigned long long)e->ring);
    const bool il = e->inline_launch;                                     // one stream, stream order is the only order
    hipStream_t sf = (e->serial || il) ? st : e->s_front;
    hipStream_t sm = (e->pipe3 && !il) ? e->s_mac : st;
    if (input_ready) HIP_TRY(hipStreamWaitEvent(sf, input_ready, 0));
    if (!il && e->chunk_seq >= 2) HIP_TRY(hipStreamWaitEvent(sf, e->ev_mac[par], 0));   // mac(k-2) is done with the ring
    // input_timecbuf bookkeeping as in run_chunk: block j of the chunk lands in buffer !(curbuf ^ (j & 1))
    const int idx_last = 1 ^ e->curbuf ^ ((tc - 1) & 1);
    {
        ProfScope ps(e, BFIR_K_FWD, sf);
        FwdPairArgs a;
        a.raw = (const float *)d_in; a.eng_stride = in_stride / 4; a.frame_off = frame_off;
        a.C = e->C; a.n_eng = e->n_eng; a.n_t = tc;
        a.prev = e->hist_raw[e->curbuf];
        a.save_last = e->tails[par][idx_last]; a.save_prev = e->tails[par][1 ^ idx_last];
        a.carry = e->hist_raw[1 ^ idx_last];
        a.hist_eng_stride = (long)e->L * e->C;
        a.dst = (float *)e->X; a.dst_ch_stride = (long)e->ring * e->N; a.ring = e->ring; a.base_slot = base_slot;
        a.scale = (float)e->in_scale;
        a.tp = e->pair_tp;
        launch_fwd_pair(e->plan2, a, sf);
    }
    e->hist_raw[0] = e->tails[par][0]; e->hist_raw[1] = e->tails[par][1];
    if (!il) {
        HIP_TRY(hipEventRecord(e->ev_fwd[par], sf));
        HIP_TRY(hipStreamWaitEvent(sm, e->ev_fwd[par], 0));
        if (e->pipe3 && e->chunk_seq >= 2) HIP_TRY(hipStreamWaitEvent(sm, e->ev_inv[par], 0));   // inv(k-2) has read Yb[par]
    }
    void *Y = e->Yb[e->pipe3 ? par : 0];
    {
        ProfScope ps(e, BFIR_K_MAC, sm);
        MacArgs a;
        a.x = e->X; a.x_ch_stride = (long)e->ring * e->N; a.ring = e->ring; a.base_slot = base_slot;
        a.h = e->H; a.h_ch_stride = (long)e->B * e->N;
        a.nblk = e->d_nblk;
        a.y = Y; a.y_ch_stride = (long)e->chunk * e->N;
        a.n_t = tc; a.n_ch = e->GC; a.N = e->N; a.realsize = e->s; a.B = e->B;
        a.interleaved = 1;
        launch_mac(a, sm);
    }
    if (!il) {
        HIP_TRY(hipEventRecord(e->ev_mac[par], sm));
        if (e->pipe3) HIP_TRY(hipStreamWaitEvent(st, e->ev_mac[par], 0));
    }
    {
        ProfScope ps(e, BFIR_K_INV, st);
        InvPairArgs a;
        a.y = (const float *)Y; a.y_ch_stride = (long)e->chunk * e->N;
        a.raw = (float *)d_out; a.eng_stride = out_stride / 4; a.frame_off = frame_off;
        a.C = e->C; a.n_eng = e->n_eng; a.n_t = tc;
        a.scale = (float)e->out_scale; a.max = (float)e->of_max;
        a.overflow = e->d_of; a.of_shard_stride = e->GC; a.bad_block = e->d_bad; a.block_base = block_base; a.bad_host = e->bad_host_cur;
        a.tp = e->pair_tp;
        launch_inv_pair(e->plan2, a, st);
    }
    if (e->pipe3 && !il) HIP_TRY(hipEventRecord(e->ev_inv[par], st));
    e->curbuf ^= (tc & 1);
    e->blockcounter += (unsigned long long)tc;
    e->chunk_seq += 1;
    return BFIR_OK;
}

// The same chunk on the direct path: k_fwd / k_inv of the general path in direct mode (kernels.h).
//   s_front : fwd(k)   raw frames -> delay-line spectrum, one channel per transform
//   s_mac   : mac(k)
//   st      : inv(k)   product spectrum -> raw frames + overflow statistics
static int run_chunk_direct(bfir_engine *e, const void *d_in, long in_stride, void *d_out, long out_stride,
                            long frame_off, int tc, int block_base, hipStream_t st, hipEvent_t input_ready)
{
    if (((uintptr_t)d_in | (uintptr_t)in_stride) % e->in_bytes || ((uintptr_t)d_out | (uintptr_t)out_stride) % e->out_bytes) {
        bfir_logf("bfir engine: frame buffers must be aligned to their sample size.");
        return BFIR_ERR_ARG;
    }
    const int par = (int)(e->chunk_seq & 1);
    const int base_slot = (int)(e->blockcounter % (unsigned long long)e->ring);
    const bool il = e->inline_launch;
    hipStream_t sf = (e->serial || il) ? st : e->s_front;
    hipStream_t sm = (e->pipe3 && !il) ? e->s_mac : st;
    if (input_ready) HIP_TRY(hipStreamWaitEvent(sf, input_ready, 0));
    if (!il && e->chunk_seq >= 2) HIP_TRY(hipStreamWaitEvent(sf, e->ev_mac[par], 0));   // mac(k-2) is done with the ring
    const int idx_last = 1 ^ e->curbuf ^ ((tc - 1) & 1);
    {
        ProfScope ps(e, BFIR_K_FWD, sf);
        FwdArgs a;
        a.src = nullptr; a.src_ch_stride = 0; a.prev = nullptr; a.prev_ch_stride = 0;
        a.dst = e->X; a.dst_ch_stride = (long)e->ring * e->N; a.ring = e->ring; a.base_slot = base_slot;
        a.n_t = tc; a.n_ch = e->GC;
        a.load_scale = 1.0; a.out_scale = e->in_scale; a.zero_first_half = 0; a.interleaved = e->ilv;
        a.raw_bytes = e->in_bytes; a.raw = d_in; a.raw_eng_stride = in_stride / e->in_bytes; a.frame_off = frame_off; a.C = e->C;
        a.prev_raw = e->hist_raw[e->curbuf];
        a.save_last = e->tails[par][idx_last]; a.save_prev = e->tails[par][1 ^ idx_last];
        a.carry = e->hist_raw[1 ^ idx_last];
        a.hist_eng_stride = (long)e->L * e->C;
        launch_fwd(e->plan, a, sf);
    }
    e->hist_raw[0] = e->tails[par][0]; e->hist_raw[1] = e->tails[par][1];
    if (!il) {
        HIP_TRY(hipEventRecord(e->ev_fwd[par], sf));
        HIP_TRY(hipStreamWaitEvent(sm, e->ev_fwd[par], 0));
        if (e->pipe3 && e->chunk_seq >= 2) HIP_TRY(hipStreamWaitEvent(sm, e->ev_inv[par], 0));   // inv(k-2) has read Yb[par]
    }
    void *Y = e->Yb[e->pipe3 ? par : 0];
    {
        ProfScope ps(e, BFIR_K_MAC, sm);
        MacArgs a;
        a.x = e->X; a.x_ch_stride = (long)e->ring * e->N; a.ring = e->ring; a.base_slot = base_slot;
        a.h = e->H; a.h_ch_stride = (long)e->B * e->N;
        a.nblk = e->d_nblk;
        a.y = Y; a.y_ch_stride = (long)e->chunk * e->N;
        a.n_t = tc; a.n_ch = e->GC; a.N = e->N; a.realsize = e->s; a.B = e->B;
        a.interleaved = e->ilv;
        launch_mac(a, sm);
    }
    if (!il) {
        HIP_TRY(hipEventRecord(e->ev_mac[par], sm));
        if (e->pipe3) HIP_TRY(hipStreamWaitEvent(st, e->ev_mac[par], 0));
    }
    {
        ProfScope ps(e, BFIR_K_INV, st);
        InvArgs a;
        a.src = Y; a.src_ch_stride = (long)e->chunk * e->N;
        a.dst = nullptr; a.dst_ch_stride = 0;
        a.n_t = tc; a.n_ch = e->GC;
        a.in_scale = e->out_scale; a.full_output = 0; a.interleaved = e->ilv;
        a.raw_bytes = e->out_bytes; a.raw = d_out; a.raw_eng_stride = out_stride / e->out_bytes; a.frame_off = frame_off; a.C = e->C;
        a.max = e->of_max; a.overflow = e->d_of; a.of_shard_stride = e->GC; a.bad_block = e->d_bad; a.block_base = block_base; a.bad_host = e->bad_host_cur;
        launch_inv(e->plan, a, st);
    }
    if (e->pipe3 && !il) HIP_TRY(hipEventRecord(e->ev_inv[par], st));
    e->curbuf ^= (tc & 1);
    e->blockcounter += (unsigned long long)tc;
    e->chunk_seq += 1;
    return BFIR_OK;
}

static int run_chunk(bfir_engine *e, const void *d_in, long in_stride, void *d_out, long out_stride,
                     long frame_off, int tc, int block_base, hipStream_t st, hipEvent_t input_ready)
{
    if (e->pair) return run_chunk_pair(e, d_in, in_stride, d_out, out_stride, frame_off, tc, block_base, st, input_ready);
    if (e->direct) return run_chunk_direct(e, d_in, in_stride, d_out, out_stride, frame_off, tc, block_base, st, input_ready);
    const int par = (int)(e->chunk_seq & 1);
    const long t_stride = (long)e->chunk * e->L;
    void *tin = e->tin[par];
    const int base_slot = (int)(e->blockcounter % (unsigned long long)e->ring);
    const bool il = e->inline_launch;                                     // one stream, stream order is the only order
    hipStream_t sf = (e->serial || il) ? st : e->s_front;

    // the front only waits for the input, never for the back of the chunk before
    if (input_ready) HIP_TRY(hipStreamWaitEvent(sf, input_ready, 0));
    {
        ProfScope ps(e, BFIR_K_STAGE_IN, sf);
        StageInArgs a;
        a.raw = d_in; a.eng_stride_bytes = in_stride; a.frame_off = frame_off;
        a.n_eng = e->n_eng; a.C = e->C; a.raw_bytes = e->in_bytes; a.spacing = e->C; a.fmt = e->in_fmt;
        a.n_frames = (long)tc * e->L;
        a.dst = tin; a.dst_ch_stride = t_stride; a.dst_off = 0;
        a.realsize = e->s;
        launch_stage_in(a, sf);
    }
    if (!il && e->chunk_seq >= 2) HIP_TRY(hipStreamWaitEvent(sf, e->ev_mac[par], 0));   // mac(k-2) is done with the ring
    {
        ProfScope ps(e, BFIR_K_FWD, sf);
        FwdArgs a;
        a.src = tin; a.src_ch_stride = t_stride;
        // the block before this chunk, as the reference sees it: first half of
        // input_timecbuf[n][curbuf] (fftw_convolver.cpp:184 left it there one call earlier)
        a.prev = e->hist[e->curbuf].ptr; a.prev_ch_stride = e->hist[e->curbuf].ch_stride;
        a.dst = e->X; a.dst_ch_stride = (long)e->ring * e->N;
        a.ring = e->ring; a.base_slot = base_slot;
        a.n_t = tc; a.n_ch = e->GC;
        a.load_scale = 1.0; a.out_scale = e->in_scale;
        a.zero_first_half = 0;
        a.interleaved = e->ilv;
        launch_fwd(e->plan, a, sf);
    }
    hipStream_t sm = (e->pipe3 && !il) ? e->s_mac : st;
    if (!il) {
        HIP_TRY(hipEventRecord(e->ev_fwd[par], sf));
        HIP_TRY(hipStreamWaitEvent(sm, e->ev_fwd[par], 0));
        if (e->pipe3 && e->chunk_seq >= 2) HIP_TRY(hipStreamWaitEvent(sm, e->ev_inv[par], 0));   // inv(k-2) has read Yb[par]
    }
    void *Y = e->Yb[e->pipe3 ? par : 0];
    {
        ProfScope ps(e, BFIR_K_MAC, sm);
        MacArgs a;
        a.x = e->X; a.x_ch_stride = (long)e->ring * e->N; a.ring = e->ring; a.base_slot = base_slot;
        a.h = e->H; a.h_ch_stride = (long)e->B * e->N;
        a.nblk = e->d_nblk;
        a.y = Y; a.y_ch_stride = (long)e->chunk * e->N;
        a.n_t = tc; a.n_ch = e->GC; a.N = e->N; a.realsize = e->s; a.B = e->B;
        a.interleaved = e->ilv;
        launch_mac(a, sm);
    }
    if (!il) {
        HIP_TRY(hipEventRecord(e->ev_mac[par], sm));
        if (e->pipe3) HIP_TRY(hipStreamWaitEvent(st, e->ev_mac[par], 0));
    }
    {
        ProfScope ps(e, BFIR_K_INV, st);
        InvArgs a;
        a.src = Y; a.src_ch_stride = (long)e->chunk * e->N;
        a.dst = e->tout; a.dst_ch_stride = t_stride;
        a.n_t = tc; a.n_ch = e->GC;
        a.in_scale = e->out_scale;
        a.full_output = 0;
        a.interleaved = e->ilv;
        launch_inv(e->plan, a, st);
    }
    if (e->pipe3 && !il) HIP_TRY(hipEventRecord(e->ev_inv[par], st));
    {
        ProfScope ps(e, BFIR_K_STAGE_OUT, st);
        StageOutArgs a;
        a.raw = d_out; a.eng_stride_bytes = out_stride; a.frame_off = frame_off;
        a.n_eng = e->n_eng; a.C = e->C; a.raw_bytes = e->out_bytes; a.spacing = e->C; a.fmt = e->out_fmt;
        a.n_frames = (long)tc * e->L;
        a.src = e->tout; a.src_ch_stride = t_stride;
        a.realsize = e->s; a.L = e->L; a.max = e->of_max;
        a.overflow = e->d_of; a.of_shard_stride = e->GC; a.bad_block = e->d_bad; a.block_base = block_base; a.bad_host = e->bad_host_cur;
        a.dither_tab = e->d_dither_tab; a.dither_size = e->dither_size; a.dither_state = e->d_dither_state;
        launch_stage_out(a, st);
    }
    // input_timecbuf bookkeeping: block j of the chunk lands in buffer
    // !(curbuf ^ (j & 1)).  Only the references move; the samples stay where
    // stage_in put them (this time buffer is not rewritten before chunk k+2,
    // by which time both references have moved on).
    const size_t Ls = (size_t)e->L * e->s;
    const int idx_last = 1 ^ e->curbuf ^ ((tc - 1) & 1);
    if (tc >= 2) { e->hist[1 ^ idx_last].ptr = (char *)tin + (size_t)(tc - 2) * Ls; e->hist[1 ^ idx_last].ch_stride = t_stride; }
    e->hist[idx_last].ptr = (char *)tin + (size_t)(tc - 1) * Ls; e->hist[idx_last].ch_stride = t_stride;
    e->curbuf ^= (tc & 1);
    e->blockcounter += (unsigned long long)tc;
    e->chunk_seq += 1;
    return BFIR_OK;
}

static int ensure_chunk(bfir_engine *e, int n_blocks)
{
    // automatic: as many blocks per launch as give a launch the work of 4096 blocks of the 8-channel, 4096-sample
    // headline shape (measured best for long jobs, profiles/r01_bench_chunk_sweep.jsonl) -- small engines need longer
    // launches to keep the three kernels of the pipeline overlapped (the stereo plug-in shape: +12 % fp64 / +19 % fp32
    // at 32768 blocks per launch against 4096, profiles/r03_chunk.txt) -- between 4096 and 32768, less when the
    // delay line of that many blocks would pass 4 GiB (many channels or engines)
    int limit = e->want_chunk;
    if (limit <= 0) {
        const long slots = (long)((4ull << 30) / ((size_t)e->GC * cbuf_bytes(e)));
        long want = (8L * 4096 * 4096) / ((long)e->GC * e->L);
        want = std::max(4096L, std::min(32768L, want));
        limit = (int)std::max(16L, std::min(want, (slots - e->B) / 2));
    }
    // HP-TPDF dither is a recursion over a channel's samples (dither.hip: one lane walks them in order), so a
    // launch's length is its run time: bound it (64 blocks: the serial walk stays in the milliseconds and the
    // other kernels of the pipeline get the GPU in between; integer outputs are off the measured path)
    if (e->d_dither_tab) limit = std::min(limit, 64);
    const int want = std::max(1, std::min(limit, n_blocks));
    if (want > e->chunk) return alloc_work(e, want);
    return BFIR_OK;
}

extern "C" int bfir_engine_run_device(bfir_engine *e, const void *d_in, int64_t in_stride_bytes,
                                      void *d_out, int64_t out_stride_bytes, int n_blocks, void *hip_stream)
{
    if (!e || !d_in || !d_out || n_blocks < 0) return BFIR_ERR_ARG;
    if (!bfir_engine_is_initialized(e)) return BFIR_ERR_STATE;
    if (n_blocks == 0) return BFIR_OK;
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    int rc = ensure_chunk(e, n_blocks);
    if (rc != BFIR_OK) return rc;
    // whatever produced d_in on the caller's stream must be done before the front reads it
    HIP_TRY(hipEventRecord(e->ev_entry, st));
    for (int c0 = 0; c0 < n_blocks; c0 += e->chunk) {
        const int tc = std::min(e->chunk, n_blocks - c0);
        rc = run_chunk(e, d_in, in_stride_bytes, d_out, out_stride_bytes, (long)c0 * e->L, tc, c0, st,
                       c0 == 0 ? e->ev_entry : nullptr);
        if (rc != BFIR_OK) return rc;
    }
    HIP_TRY(hipGetLastError());
    e->async_pending = true;
    return BFIR_OK;
}

extern "C" int bfir_engine_sync(bfir_engine *e)
{
    if (!e) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    e->async_pending = false;
    drain_spans(e);
    int bad = INT_MAX;
    HIP_TRY(hipMemcpy(&bad, e->d_bad, sizeof(int), hipMemcpyDeviceToHost));
    if (bad != 0x7f7f7f7f) {
        HIP_TRY(hipMemset(e->d_bad, 0x7f, sizeof(int)));
        bfir_logf("NaN or Inf values in the system! Invalid input? Aborting.\n");
        return BFIR_ERR_NONFINITE;
    }
    return BFIR_OK;
}

// memcpy between the caller's (pageable) buffers and the pinned staging buffers, split over a few
// threads when it is large: one core moves ~9 GB/s, the host link 63 GB/s.
static void copy_host(void *dst, const void *src, size_t n)
{
    constexpr size_t kMinPerThread = 4u << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), n / kMinPerThread);
    if (nt <= 1) { memcpy(dst, src, n); return; }
    const size_t per = ((n / nt) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (size_t i = 1; i < nt; i++) {
        const size_t o = i * per;
        if (o >= n) break;
        th.emplace_back([=]() { memcpy((char *)dst + o, (const char *)src + o, std::min(per, n - o)); });
    }
    memcpy(dst, src, std::min(per, n));
    for (auto &t : th) t.join();
}

// Is [p, p + n) page-locked host memory the copy engines can reach directly (bfir_pinned_malloc, or the caller's own
// hipHostMalloc / hipHostRegister)?  Then the staging memcpy -- one pass of a few host cores over every byte, the slowest leg
// of the host-pointer path -- is skipped for that buffer.
static bool is_pinned_host(const void *p, size_t n)
{
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // pageable: not known to the runtime
    if (a0.type != hipMemoryTypeHost) return false;
    if (n <= 1) return true;
    if (hipPointerGetAttributes(&a1, (const char *)p + n - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a1.type == hipMemoryTypeHost;
}

extern "C" void *bfir_pinned_malloc(size_t size)
{
    void *p = nullptr;
    if (bfir_device_count() <= 0) return nullptr;
    if (hipHostMalloc(&p, size ? size : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

extern "C" void bfir_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

// Host-path chunk: the link, not the GPU, bounds this path, so the pinned staging buffers stay small.
static int host_chunk(const bfir_engine *e) { return std::min(e->chunk, 512); }

static int ensure_staging(bfir_engine *e, int min_blocks = 0)
{
    const int hc = std::max(host_chunk(e), min_blocks);
    const size_t bin = (size_t)e->n_eng * hc * e->L * e->C * e->in_bytes;
    const size_t bout = (size_t)e->n_eng * hc * e->L * e->C * e->out_bytes;
    if (e->stage_bytes_in >= bin && e->stage_bytes_out >= bout) return BFIR_OK;
    for (int i = 0; i < 2; i++) {
        if (e->pin_in[i]) (void)hipHostFree(e->pin_in[i]);
        if (e->pin_out[i]) (void)hipHostFree(e->pin_out[i]);
        if (e->dev_in[i]) (void)hipFree(e->dev_in[i]);
        if (e->dev_out[i]) (void)hipFree(e->dev_out[i]);
        e->pin_in[i] = e->pin_out[i] = e->dev_in[i] = e->dev_out[i] = nullptr;
        HIP_TRY(hipHostMalloc(&e->pin_in[i], bin, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(&e->pin_out[i], bout, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&e->dev_in[i], bin));
        HIP_TRY(hipMalloc(&e->dev_out[i], bout));
    }
    e->stage_bytes_in = bin; e->stage_bytes_out = bout;
    return BFIR_OK;
}

// A handful of blocks per call -- the plug-in's pattern is ONE (foo_dsp_bfir.cpp:311-349): what
// counts is the latency of the call, not throughput.  No copy engine, no second stream, no event: the
// caller's frames go into the pinned staging buffer, the kernels read them from there and write the output
// frames into the other one straight across the host link (a block is a few KiB), launched back to back on
// one stream; one 4-byte copy brings the NaN verdict; one stream synchronise ends the call.
static constexpr int kSmallRun = 4;

// 16 bytes per lane between pinned host memory and HBM (whole cache lines across the host link)
__global__ __launch_bounds__(256) void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, long n16)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

static int run_small(bfir_engine *e, const void *inbuf, void *outbuf, int n_blocks)
{
    int rc = ensure_chunk(e, n_blocks);
    if (rc != BFIR_OK) return rc;
    rc = ensure_staging(e, kSmallRun);
    if (rc != BFIR_OK) return rc;
    if (e->async_pending) { HIP_TRY(hipDeviceSynchronize()); e->async_pending = false; }
    if (!e->h_bad) { HIP_TRY(hipHostMalloc((void **)&e->h_bad, sizeof(int) * kSmallRun, hipHostMallocDefault)); }
    for (int t = 0; t < kSmallRun; t++) e->h_bad[t] = 0;
    const size_t per_in = (size_t)n_blocks * e->L * e->C * e->in_bytes, per_out = (size_t)n_blocks * e->L * e->C * e->out_bytes;
    memcpy(e->pin_in[0], inbuf, per_in * e->n_eng);       // engine after engine, n_blocks * L frames each: same layout
    // Frames wider than what one workgroup of the fused FFT kernels consumes (a channel pair or one channel of
    // many): every workgroup would pull its 8 bytes of each frame across the host link on its own (measured:
    // 26 us per FFT kernel for one block of the 8-channel headline shape, against 8 us for stereo frames).
    // Those engines get the block into HBM and out of it by a copy kernel moving whole lines: two more
    // launches, ~35 us less per call.
    const bool bounce = (e->pair || e->direct) && (size_t)e->C * e->in_bytes > 8 && !getenv("BFIR_NO_BOUNCE");
    const void *src = e->pin_in[0];
    void *dst = e->pin_out[0];
    if (bounce) {
        const long n16 = (long)(per_in * e->n_eng / 16);
        hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, e->stream, (uint4 *)e->dev_in[0],
                           (const uint4 *)e->pin_in[0], n16);
        src = e->dev_in[0]; dst = e->dev_out[0];
    }
    e->inline_launch = true;
    for (int c0 = 0; c0 < n_blocks && rc == BFIR_OK; c0 += e->chunk) {     // the work buffers hold e->chunk blocks
        e->bad_host_cur = e->h_bad + c0;                                   // the NaN verdict of block c0 + t lands in h_bad[c0 + t]
        rc = run_chunk(e, src, (long)per_in, dst, (long)per_out, (long)c0 * e->L,
                       std::min(e->chunk, n_blocks - c0), c0, e->stream, nullptr);
    }
    e->bad_host_cur = nullptr;
    e->inline_launch = false;
    if (rc != BFIR_OK) return rc;
    if (bounce) {
        const long n16 = (long)(per_out * e->n_eng / 16);
        hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, e->stream, (uint4 *)e->pin_out[0],
                           (const uint4 *)e->dev_out[0], n16);
    }
    // no copy of the verdict: the kernels flagged bad blocks in pinned host memory themselves (one 4-byte copy was a blit
    // kernel of 3.5 us plus its launch, a tenth of the call)
    HIP_TRY(hipStreamSynchronize(e->stream));
    drain_spans(e);
    memcpy(outbuf, e->pin_out[0], per_out * e->n_eng);
    bool bad = false;
    for (int t = 0; t < n_blocks; t++) bad = bad || e->h_bad[t] != 0;
    if (bad) {
        HIP_TRY(hipMemset(e->d_bad, 0x7f, sizeof(int)));
        bfir_logf("NaN or Inf values in the system! Invalid input? Aborting.\n");
        return BFIR_ERR_NONFINITE;
    }
    return BFIR_OK;
}

// Host-pointer run: pinned double buffers, H2D on s_in, kernels on the engine's
// streams, D2H on s_out, so the copies of neighbouring chunks overlap the compute
// (the reference's raw2real / real2raw staging, fftw_convolver.cpp:156-185, 405-466,
// turned into pinned-host staging).
extern "C" int bfir_engine_run(bfir_engine *e, const void *inbuf, void *outbuf, int n_blocks)
{
    if (!e || !inbuf || !outbuf || n_blocks < 0) return BFIR_ERR_ARG;
    if (!bfir_engine_is_initialized(e)) return BFIR_ERR_STATE;
    if (n_blocks == 0) return BFIR_OK;
    HIP_TRY(hipSetDevice(e->device));
    if (n_blocks <= kSmallRun && !getenv("BFIR_NO_SMALL_RUN")) return run_small(e, inbuf, outbuf, n_blocks);
    int rc = ensure_chunk(e, std::min(n_blocks, 512));   // see host_chunk()
    if (rc != BFIR_OK) return rc;
    rc = ensure_staging(e);
    if (rc != BFIR_OK) return rc;
    const size_t fin = (size_t)e->C * e->in_bytes, fout = (size_t)e->C * e->out_bytes;  // bytes per frame
    const size_t eng_in = (size_t)n_blocks * e->L * fin, eng_out = (size_t)n_blocks * e->L * fout;
    const int hc = host_chunk(e);
    const int nchunks = (n_blocks + hc - 1) / hc;
    // page-locked caller buffers go straight to / from the copy engines (no staging memcpy); BFIR_NO_PINNED_DIRECT=1: A/B, tests
    const bool direct_ok = !getenv("BFIR_NO_PINNED_DIRECT");
    const bool in_direct = direct_ok && is_pinned_host(inbuf, eng_in * e->n_eng);
    const bool out_direct = direct_ok && is_pinned_host(outbuf, eng_out * e->n_eng);
    auto copy_out = [&](int k) -> int {
        const int b = k & 1, c0 = k * hc, tc = std::min(hc, n_blocks - c0);
        HIP_TRY(hipEventSynchronize(e->ev_d2h[b]));
        if (out_direct) return BFIR_OK;
        const size_t per = (size_t)tc * e->L * fout;
        for (int g = 0; g < e->n_eng; g++)
            copy_host((char *)outbuf + g * eng_out + (size_t)c0 * e->L * fout, (char *)e->pin_out[b] + g * per, per);
        return BFIR_OK;
    };
    for (int k = 0; k < nchunks; k++) {
        const int b = k & 1, c0 = k * hc, tc = std::min(hc, n_blocks - c0);
        if (k >= 2) { rc = copy_out(k - 2); if (rc != BFIR_OK) return rc; }
        const size_t per_in = (size_t)tc * e->L * fin, per_out = (size_t)tc * e->L * fout;
        if (in_direct) {
            for (int g = 0; g < e->n_eng; g++)
                HIP_TRY(hipMemcpyAsync((char *)e->dev_in[b] + g * per_in, (const char *)inbuf + g * eng_in + (size_t)c0 * e->L * fin,
                                       per_in, hipMemcpyHostToDevice, e->s_in));
        } else {
            for (int g = 0; g < e->n_eng; g++)
                copy_host((char *)e->pin_in[b] + g * per_in, (const char *)inbuf + g * eng_in + (size_t)c0 * e->L * fin, per_in);
            HIP_TRY(hipMemcpyAsync(e->dev_in[b], e->pin_in[b], per_in * e->n_eng, hipMemcpyHostToDevice, e->s_in));
        }
        HIP_TRY(hipEventRecord(e->ev_h2d[b], e->s_in));
        rc = run_chunk(e, e->dev_in[b], (long)per_in, e->dev_out[b], (long)per_out, 0, tc, c0, e->stream,
                       e->ev_h2d[b]);
        if (rc != BFIR_OK) return rc;
        HIP_TRY(hipEventRecord(e->ev_comp[b], e->stream));
        HIP_TRY(hipStreamWaitEvent(e->s_out, e->ev_comp[b], 0));
        if (out_direct) {
            for (int g = 0; g < e->n_eng; g++)
                HIP_TRY(hipMemcpyAsync((char *)outbuf + g * eng_out + (size_t)c0 * e->L * fout, (char *)e->dev_out[b] + g * per_out,
                                       per_out, hipMemcpyDeviceToHost, e->s_out));
        } else
            HIP_TRY(hipMemcpyAsync(e->pin_out[b], e->dev_out[b], per_out * e->n_eng, hipMemcpyDeviceToHost, e->s_out));
        HIP_TRY(hipEventRecord(e->ev_d2h[b], e->s_out));
        // the next H2D into dev_in[b] must not overtake this chunk's kernels
        HIP_TRY(hipStreamWaitEvent(e->s_in, e->ev_comp[b], 0));
    }
    for (int k = std::max(0, nchunks - 2); k < nchunks; k++) { rc = copy_out(k); if (rc != BFIR_OK) return rc; }
    return bfir_engine_sync(e);
}

extern "C" void bfir_engine_reset(bfir_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    // brutefir.cpp:346-367: counters only.  procblocks = 0 hides every
    // delay-line slot written before the reset (brutefir.cpp:292), which the
    // zeroed ring reproduces; the time-domain history is NOT cleared, so both
    // input_timecbuf halves are kept (copied out of the work buffers).
    (void)materialise_history(e);
    (void)hipMemset(e->d_of, 0, sizeof(DevOverflow) * e->GC * BFIR_OF_SHARDS);
    // block t < B-1 of the new run still reads slots (t - i) mod ring, i > t: the B-1 slots at the top of
    // every channel's ring.  Only those need to read as zero; the rest is rewritten before it is read.
    if (e->B > 1) {
        const size_t cb = cbuf_bytes(e);
        (void)hipMemset2D((char *)e->X + (size_t)(e->ring - (e->B - 1)) * cb, (size_t)e->ring * cb, 0,
                          (size_t)(e->B - 1) * cb, (size_t)e->GC);
    }
    (void)hipDeviceSynchronize();
    e->blockcounter = 0;
    e->curbuf = 0;
}

extern "C" int bfir_engine_get_overflow(bfir_engine *e, int channel, bfir_overflow *of)
{
    if (!e || !of || channel < 0 || channel >= e->GC) return BFIR_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    // the copies of the counters (kernels.h, BFIR_OF_SHARDS): counts add up, peaks are maxima (the bit patterns of
    // non-negative floats order like integers); the dither kernel keeps its running state in copy 0
    std::vector<DevOverflow> all((size_t)e->GC * BFIR_OF_SHARDS);
    HIP_TRY(hipMemcpy(all.data(), e->d_of, all.size() * sizeof(DevOverflow), hipMemcpyDeviceToHost));
    DevOverflow d = all[channel];
    for (int sh = 1; sh < BFIR_OF_SHARDS; sh++) {
        const DevOverflow &o = all[(size_t)sh * e->GC + channel];
        d.n_overflows += o.n_overflows;
        d.intlargest = std::max(d.intlargest, o.intlargest);
        d.largest_bits = std::max(d.largest_bits, o.largest_bits);
    }
    of->n_overflows = d.n_overflows;
    of->intlargest = d.intlargest;
    if (e->s == 4) {
        unsigned int u = (unsigned int)d.largest_bits;
        float f;
        memcpy(&f, &u, sizeof(f));
        of->largest = (double)f;
    } else {
        double f;
        memcpy(&f, &d.largest_bits, sizeof(f));
        of->largest = f;
    }
    of->max = e->of_max;
    return BFIR_OK;
}
