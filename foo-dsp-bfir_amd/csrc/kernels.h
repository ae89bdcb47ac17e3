// kernels.h -- host-callable launchers of the gfx950 kernels (internal API,
// C++; the public C ABI is include/bfir_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

namespace bfir {

// Per-channel output statistics kept on the device; folded into
// bfoverflow_t (brutefir/global.h:96-102) by the host.
struct DevOverflow {
    unsigned int n_overflows;          // samples with |y| > max
    int intlargest;                    // integer outputs: largest unclipped |sample|
    unsigned long long largest_bits;   // bit pattern of max |y| (float widened, or double)
};
// The engine keeps BFIR_OF_SHARDS copies of its per-channel counters and every workgroup updates the copy
// blockIdx.x mod BFIR_OF_SHARDS (args.of_shard_stride elements apart; 0 = a single copy): when the output clips,
// every workgroup adds to n_overflows, and same-address atomics from a hundred thousand workgroups ran the
// plug-in's own shape at half speed.  bfir_engine_get_overflow sums / maximises over the copies.
constexpr int BFIR_OF_SHARDS = 64;
__device__ __forceinline__ DevOverflow *of_shard(DevOverflow *base, long shard_stride)
{
    return base + (long)(blockIdx.x & (BFIR_OF_SHARDS - 1)) * shard_stride;
}

// brutefir.cpp:316-321's verdict: block t of this launch has a non-finite sample 0.  The first such block of a run is kept
// by atomicMin in HBM; the latency path (a handful of blocks per call, run_small) also gets one flag per block in pinned
// host memory -- plain, idempotent stores, visible with the stream's end -- so that no copy has to follow the kernels.
template <typename A> __device__ __forceinline__ void flag_bad(const A &a, int t)
{
    atomicMin(a.bad_block, a.block_base + t);
    if (a.bad_host) a.bad_host[t] = 1;
}

// Sample formats (brutefir/global.h:24-34; table of brutefir.cpp:435-538, little-endian host).
struct FmtInfo { int bytes; bool isfloat; bool big_endian; };
inline FmtInfo fmt_info(int fmt)
{
    static const int b[12] = {0, 1, 2, 2, 3, 3, 4, 4, 4, 4, 8, 8};
    FmtInfo f;
    f.bytes = (fmt >= 1 && fmt <= 11) ? b[fmt] : 0;
    f.isfloat = fmt >= 8 && fmt <= 11;
    f.big_endian = fmt == 3 || fmt == 5 || fmt == 7 || fmt == 9 || fmt == 11;
    return f;
}
// get_full_scale (brutefir.cpp:395-398) in the reference's int arithmetic: negative for 32 bits.
inline double fmt_full_scale(int fmt) { return (double)(int32_t)(1u << (8 * fmt_info(fmt).bytes - 1)); }
inline bool fmt_is_native(int fmt) { return fmt == 8 || fmt == 10; }   // FLOAT_LE, FLOAT64_LE: fast staging kernels

// Twiddle tables of one transform size / precision, resident in HBM.
struct FftPlan {
    int log2m = 0;        // M = L complex points; N = 2M reals
    int realsize = 0;     // 4 or 8
    void *tw = nullptr;   // per-pass twiddles  exp(-2 pi i r k / (p R))
    void *ws = nullptr;   // split twiddles     exp(-2 pi i k / N), k < M
    void *twb = nullptr;  // twiddle bases of the persistent kernels (fft_lds.h)
};

int  fft_plan_create(FftPlan *plan, int filter_length, int realsize);   // 0 or negative error
void fft_plan_destroy(FftPlan *plan);
int  fft_threads(int log2m);

// a5: interleaved raw frames -> planar working-precision time buffers.
// Engine e reads frames frame_off .. frame_off+n_frames-1 of the interleaved
// buffer at raw + e*eng_stride_bytes.  dst[gc][dst_off + f], gc = e*C + c.
struct StageInArgs {
    const void *raw; long eng_stride_bytes; long frame_off;
    int n_eng, C, raw_bytes;       // raw_bytes: bytes per raw sample
    int spacing;                   // samples between frames (buffer_format_t.sample_spacing)
    long n_frames;
    void *dst; long dst_ch_stride; long dst_off;   // in reals
    int realsize;
    int fmt = 0;                   // BF_SAMPLE_FORMAT_* code; 0 = FLOAT_LE / FLOAT64_LE by raw_bytes
};
void launch_stage_in(const StageInArgs &a, hipStream_t s);

// a13: planar time -> interleaved raw frames + overflow statistics + NaN guard.
struct StageOutArgs {
    void *raw; long eng_stride_bytes; long frame_off;
    int n_eng, C, raw_bytes;
    int spacing;
    long n_frames;
    const void *src; long src_ch_stride;           // in reals
    int realsize;
    int L;                         // block length: sample 0 of every block is NaN-checked
    double max;                    // bfoverflow_t.max
    DevOverflow *overflow;         // [n_eng*C]
    long of_shard_stride = 0;
    int *bad_block;                // atomicMin of the first block with a non-finite sample 0
    int *bad_host = nullptr;       // latency path: one flag per block of the launch in pinned host memory (flag_bad)
    int block_base;                // index of the chunk's first block within the run
    int fmt = 0;                   // BF_SAMPLE_FORMAT_* code; 0 = FLOAT_LE / FLOAT64_LE by raw_bytes
    // HP-TPDF dither (integer formats only, dither.hip): the shared random table and one state per global channel
    const void *dither_tab = nullptr; int dither_size = 0; void *dither_state = nullptr;
};
void launch_stage_out(const StageOutArgs &a, hipStream_t s);

// dither_state_t (brutefir/global.h:63-69) as the engine keeps it in HBM, one per global channel
struct DevDitherState { int randtab_ptr; int pad; float sf[2]; double sd[2]; };
// class dither's table (brutefir/dither.cpp:21-110): spacing between channels in samples (-1: budget too small)
int dither_spacing(int n_channels, int sample_rate, int max_size, int max_samples_per_loop);
void dither_fill_table(std::vector<int8_t> &tab, int n_channels, int spacing);
void launch_stage_out_dither(const StageOutArgs &a, hipStream_t s);

// a6 + a7 (and a16 with zero_first_half): real FFT of the N-sample window
// [block t-1 | block t] of channel gc, written in the grouped layout to
// dst + gc*dst_ch_stride + ((base_slot + t) % ring) * N.  Block t (t >= 0) is
// at src + gc*src_ch_stride + t*L; the block before block 0 is at
// prev + gc*prev_ch_stride.
struct FwdArgs {
    const void *src; long src_ch_stride;
    const void *prev; long prev_ch_stride;
    void *dst; long dst_ch_stride;
    int ring, base_slot;
    int n_t, n_ch;
    double load_scale, out_scale;
    int zero_first_half;
    int interleaved = 0;           // spectrum layout: 0 = the reference's groups (4 re | 4 im), 1 = (re, im) pairs
    // Direct mode (raw_bytes = 4 / 8, FLOAT_LE / FLOAT64_LE frames): the samples come straight from the
    // interleaved raw frames instead of a planar buffer -- no staging kernel, no planar time buffers.
    // Channel gc = g C + c reads frame f of engine g at raw + g raw_eng_stride + (frame_off + f) C + c (in
    // samples); the block before block 0 is the raw-frame block prev_raw [n_eng][L][C]; the raw frames of the
    // chunk's last two blocks are kept in save_last / save_prev as the pair path keeps them (pair.hip).
    int raw_bytes = 0;
    const void *raw = nullptr; long raw_eng_stride = 0, frame_off = 0; int C = 0;
    const void *prev_raw = nullptr, *carry = nullptr; void *save_last = nullptr, *save_prev = nullptr;
    long hist_eng_stride = 0;
};
void launch_fwd(const FftPlan &plan, const FwdArgs &a, hipStream_t s);

// a8/a9/a10: Y[gc][t] = sum_{i < nblk[gc]} X[gc][slot(t - i)] * H[gc][i].
struct MacArgs {
    const void *x; long x_ch_stride; int ring, base_slot;
    const void *h; long h_ch_stride;               // [gc][B][N]
    const int *nblk;                               // device [n_ch]
    void *y; long y_ch_stride;                     // [gc][n_t][N]
    int n_t, n_ch, N, realsize;
    int B = 0;                                     // partitions allocated per channel (max of nblk)
    int interleaved = 0;                           // layout of x, h and y, as in FwdArgs (fp32 streaming kernel only)
#ifdef BFIR_EXPERIMENT_ALIAS
    int y_alias = 1 << 30;                         // timing experiment: product spectrum t lives in slot t % y_alias
#endif
};
// Cache policy of the two big streams, the delay line X and the product spectra Y (each written once and read
// once, a gigabyte apart): bit 0 = stores nontemporal, bit 1 = loads nontemporal (-DBFIR_NT_X=n / -DBFIR_NT_Y=n;
// persistent pair kernels and the streaming MAC).  All four on is the product setting: +4 % on the headline
// pipeline (112.1 vs 107.5 Gsamples/s, same box; any three of them +1.5..3 %, profiles/r02_nt_policy.txt) --
// the lines of the interleaved input / output frames, of which every workgroup uses a quarter, then
// survive in L2 until the other three channel pairs have come by.  No effect on small launches.
#ifndef BFIR_NT_X
#define BFIR_NT_X 3
#endif
#ifndef BFIR_NT_Y
#define BFIR_NT_Y 3
#endif
#ifdef BFIR_EXPERIMENT_ALIAS
// Timing experiment (scripts/gpu_alias_exp.sh): BFIR_X_ALIAS / BFIR_Y_ALIAS fold the delay line / the product
// spectra into that many slots, so that they stay in the caches.  Results are garbage; instruction streams
// and launch geometry are those of the product build.  Never defined in the product library.
#define BFIR_YSLOT(a, t) ((t) % (a).y_alias)
inline int bfir_alias_env(const char *name) { const char *e = getenv(name); return e && atoi(e) > 0 ? atoi(e) : 0; }
#else
#define BFIR_YSLOT(a, t) (t)
#endif
void launch_mac(const MacArgs &a, hipStream_t s);
// mac_sys.hip: the forward-walking two-lanes-per-bin form of the fp32 pair-layout MAC (B <= 32)
constexpr int BFIR_MAC_SYS_MAX_B = 256;   // partitions the systolic MAC takes (sixteen stages of sixteen)
bool mac_sys_supported(const MacArgs &a);
void launch_mac_sys(const MacArgs &a, hipStream_t s);

// a11 + a12: inverse real FFT of Y[gc][t] (grouped layout, times in_scale),
// first L samples to dst + gc*dst_ch_stride + t*L.
struct InvArgs {
    const void *src; long src_ch_stride;           // [gc][n_t][N]
    void *dst; long dst_ch_stride;
    int n_t, n_ch;
    double in_scale;
    int full_output;                               // 1: write all N samples at stride N (stage API)
    int interleaved = 0;                           // layout of src, as in FwdArgs
    // Direct mode (raw_bytes = 4 / 8): the valid half goes straight into the interleaved raw output frames
    // (same addressing as FwdArgs) with the overflow statistics and the NaN guard of real2raw
    // (brutefir/real2raw.cpp:321-336, brutefir.cpp:316-321): no planar output buffer, no staging kernel.
    int raw_bytes = 0;
    void *raw = nullptr; long raw_eng_stride = 0, frame_off = 0; int C = 0;
    double max = 1.0; DevOverflow *overflow = nullptr; int *bad_block = nullptr; int block_base = 0;
    int *bad_host = nullptr;                       // as in StageOutArgs
    long of_shard_stride = 0;
};
void launch_inv(const FftPlan &plan, const InvArgs &a, hipStream_t s);

// Pair path (pair.hip): two adjacent channels per complex transform, straight from / to interleaved
// FLOAT_LE frames; fp32, even channel count, 512 <= L <= 8192.  `plan` is the plan of 2L points.
bool pair_supported(int filter_length);
// k_fwd / k_inv can take both channels of a stereo float-frame block in one workgroup (direct mode, whole frames)
bool direct_stereo_supported(int filter_length, int realsize);
// fp64 engines of 1024 ... 8192 points in direct mode: k_fwd_run / k_inv_run, runs of blocks per workgroup with the next block's
// frames / spectrum fetched under the current transform (BFIR_RUN64=0: off; =n: blocks per run)
bool run64_supported(int filter_length, int realsize);
// ... and such engines may keep their spectra as (re, im) pairs (engine.hip; 1024 ... 4096 points)
bool pairs64_supported(int filter_length, int realsize);
struct FwdPairArgs {
    const float *raw; long eng_stride; long frame_off;   // input frames; engine stride in floats
    int C, n_eng, n_t;
    const float *prev;                                   // the block before block 0: [n_eng][L][C] raw frames
    float *save_last, *save_prev;                        // where blocks n_t-1 / n_t-2 are kept for the next chunk
    const float *carry;                                  // n_t == 1: the history block that becomes save_prev
    long hist_eng_stride;                                // floats between engines in prev / save_* / carry
    float *dst; long dst_ch_stride; int ring, base_slot; // delay line, (re, im) pairs
    float scale;
    int tp = 0;                                          // pairs in TIME: blocks t, t+1 of one channel per transform (any C)
};
void launch_fwd_pair(const FftPlan &plan, const FwdPairArgs &a, hipStream_t s);
struct InvPairArgs {
    const float *y; long y_ch_stride;                    // [gc][n_t][N] product spectra, (re, im) pairs
    float *raw; long eng_stride; long frame_off;         // output frames
    int C, n_eng, n_t;
    float scale, max;
    DevOverflow *overflow; int *bad_block; int block_base;
    int *bad_host = nullptr;                             // as in StageOutArgs
    long of_shard_stride = 0;
    int tp = 0;                                          // pairs in time, as in FwdPairArgs
#ifdef BFIR_EXPERIMENT_ALIAS
    int y_alias = 1 << 30;
#endif
};
void launch_inv_pair(const FftPlan &plan, const InvPairArgs &a, hipStream_t s);

// mixnscale with one buffer (a7 / a11) on half-complex data, for the stage API.
void launch_reorder(const void *in, void *out, int n_fft, double scale, int to_grouped, int realsize,
                    hipStream_t s);

// SURVEY 8f row 3 (stage API only): N-input mixnscale, dirac_convolve, the blend of
// crossfade_inplace, finite check.
constexpr int BFIR_MAX_MIX = 32;
struct MixArgs {
    const void *in[BFIR_MAX_MIX];
    double scale[BFIR_MAX_MIX];
    int n;
};
void launch_reorder_n(const MixArgs &m, void *out, int n_fft, int to_grouped, int realsize, hipStream_t s);
void launch_dirac(const void *in, void *out, int n_fft, int realsize, hipStream_t s);
void launch_crossfade_blend(const void *crossfade_time, void *buffer_time, const void *buffer_tail, int n_fft2,
                            int realsize, hipStream_t s);
void launch_check_finite(const void *buf, int n, int realsize, int *bad, hipStream_t s);

// One convolve / convolve_add / convolve_inplace call of the stage API with the
// reference's exact operation order (separate multiplies and adds).
// mode 0: d = b*c, mode 1: d += b*c.  d may alias b (in-place form).
void launch_cmul_stage(const void *b, const void *c, void *d, int n_fft, int mode, int realsize,
                       hipStream_t s);

}  // namespace bfir
