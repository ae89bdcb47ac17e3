// pair.hip -- two channels per transform, straight from / to interleaved float frames.
//
// The engine's fast path for FLOAT_LE in and out, an even channel count and fp32 arithmetic
// (the plug-in's and the offline drivers' configuration: foo_dsp_bfir.cpp:283-284,
// preprocessor.cpp:292-293).  Channels 2c and 2c+1 of a frame are adjacent floats, so
//
//     z[n] = x_2c[n] + i x_2c+1[n],    n = 0 .. N-1   (N = 2L: previous block | this block)
//
// is ONE 8-byte load per point from the raw frames, and one complex FFT of N points yields both
// half-complex spectra (the classic two-for-one split, no twiddle):
//
//     X_a[k] = (Z[k] + conj Z[N-k]) / 2,      X_b[k] = (Z[k] - conj Z[N-k]) / (2i).
//
// The inverse runs the same identity backwards (Z = Y_a + i Y_b, Hermitian-extended), and
// Re z / Im z of the complex inverse are the two channels' samples, stored as 8-byte pairs
// into the raw output frames together with the overflow statistics and the NaN guard of
// real2raw (brutefir/real2raw.cpp:321-336, brutefir.cpp:316-321).
// This replaces raw2real + R2HC + mixnscale (fftw_convolver.cpp:156-209, a5-a7) and
// mixnscale + HC2R + real2raw (:360-466, a11-a13) for such engines and removes the planar time
// buffers with their 2 x 2 trips through HBM; the MAC between them is k_mac_stream.
// Spectra are (re, im) pairs (MacArgs.interleaved); bin 0 carries DC | Nyquist.
#include "kernels.h"

#include <algorithm>

#include "fft_lds.h"

namespace bfir {

namespace {

// XCD-aware bijective remap: blocks b, b+8, ... run on one XCD; give every XCD one contiguous
// range of work items so the channel pairs of a block (which share input cache lines) meet in one L2.
__device__ __forceinline__ int xcd_remap()
{
    const int W = gridDim.x, b = blockIdx.x, xcd = b & 7, qn = W >> 3, rn = W & 7;
    return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
}

template <int LOG2N>
__global__ __launch_bounds__(FftCfg<LOG2N>::NT) void k_fwd_pair(FwdPairArgs a, const float2 *__restrict__ tw)
{
    using F = LdsFft<float, LOG2N, -1>;
    constexpr int N = F::M, NT = F::NT, P = F::P, L = N / 2;
    __shared__ __attribute__((aligned(16))) float2 lds[F::LDS_ELEMS];
#ifdef BFIR_PAIR_LDS_PAD   // occupancy experiment: extra LDS so that fewer workgroups fit a CU
    __shared__ volatile float occ_pad[BFIR_PAIR_LDS_PAD / 4];
    if (a.n_t < 0) { occ_pad[threadIdx.x] = 1.f; a.scale += occ_pad[threadIdx.x ^ 1]; }
#endif

    const int tid = threadIdx.x;
    const int w = xcd_remap();
    const int half_c = a.C / 2, pairs = a.n_eng * half_c;
    const int t = w / pairs, pp = w - t * pairs;
    const int g = pp / half_c, cp = pp - g * half_c;
    const int C = a.C;
    // frames of block t of engine g, channel pair cp
    const float *__restrict__ cur = a.raw + (long)g * a.eng_stride + (a.frame_off + (long)t * L) * C + 2 * cp;
    const float *__restrict__ old = (t == 0) ? a.prev + (long)g * a.hist_eng_stride + 2 * cp : cur - (long)L * C;

    BFIR_STAMP(0, 0);
    float re[P], im[P];
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int n = F::in_index(tid, e);
        const float2 v = (n < L) ? *(const float2 *)(old + (long)n * C) : *(const float2 *)(cur + (long)(n - L) * C);
        // the engine's history: raw frames of the last two blocks of the chunk
        if (n >= L) {
            if (t == a.n_t - 1) *(float2 *)(a.save_last + (long)g * a.hist_eng_stride + (long)(n - L) * C + 2 * cp) = v;
            else if (t == a.n_t - 2) *(float2 *)(a.save_prev + (long)g * a.hist_eng_stride + (long)(n - L) * C + 2 * cp) = v;
        } else if (a.n_t == 1) {
            // one-block chunk: the other history block moves on unchanged
            const long o = (long)g * a.hist_eng_stride + (long)n * C + 2 * cp;
            *(float2 *)(a.save_prev + o) = *(const float2 *)(a.carry + o);
        }
        re[e] = v.x * a.scale; im[e] = v.y * a.scale;
    }
    BFIR_STAMP(0, 1);

    F::run(re, im, lds, tw, tid);

    // Z in natural order to LDS, then two-for-one split
    __syncthreads();
#pragma unroll
    for (int e = 0; e < P; e++) {
        float2 v; v.x = re[e]; v.y = im[e];
        lds[F::phys(F::out_index(tid, e))] = v;
    }
    __syncthreads();
    const long slot = (long)((a.base_slot + t) % a.ring) * N;          // N floats per spectrum
    float2 *__restrict__ da = (float2 *)(a.dst + (long)(g * C + 2 * cp) * a.dst_ch_stride + slot);
    float2 *__restrict__ db = (float2 *)(a.dst + (long)(g * C + 2 * cp + 1) * a.dst_ch_stride + slot);
#pragma unroll
    for (int j = 0; j < P / 2; j++) {
        const int k = tid + j * NT;                                    // bin 0 .. L-1
        const float2 zk = lds[F::phys(k)];
        const float2 zn = lds[F::phys((N - k) & (N - 1))];
        float2 xa, xb;
        xa.x = 0.5f * (zk.x + zn.x); xa.y = 0.5f * (zk.y - zn.y);
        xb.x = 0.5f * (zk.y + zn.y); xb.y = -0.5f * (zk.x - zn.x);
        if (k == 0) {                                                  // DC | Nyquist, both real
            const float2 zh = lds[F::phys(L)];
            xa.x = zk.x; xa.y = zh.x; xb.x = zk.y; xb.y = zh.y;
        }
        da[k] = xa; db[k] = xb;
    }
    BFIR_STAMP(0, 10);
}

template <int LOG2N>
__global__ __launch_bounds__(FftCfg<LOG2N>::NT) void k_inv_pair(InvPairArgs a, const float2 *__restrict__ tw)
{
    using F = LdsFft<float, LOG2N, +1>;
    constexpr int N = F::M, NT = F::NT, P = F::P, L = N / 2;
    __shared__ __attribute__((aligned(16))) float2 lds[F::LDS_ELEMS];
    __shared__ unsigned int red_max[NT / 64 > 0 ? NT / 64 : 1][2], red_cnt[NT / 64 > 0 ? NT / 64 : 1][2];
#ifdef BFIR_PAIR_LDS_PAD
    __shared__ volatile float occ_pad[BFIR_PAIR_LDS_PAD / 4];
    if (a.n_t < 0) { occ_pad[threadIdx.x] = 1.f; a.scale += occ_pad[threadIdx.x ^ 1]; }
#endif

    const int tid = threadIdx.x;
    const int w = xcd_remap();
    const int half_c = a.C / 2, pairs = a.n_eng * half_c;
    const int t = w / pairs, pp = w - t * pairs;
    const int g = pp / half_c, cp = pp - g * half_c;
    const int C = a.C;
    const int gc = g * C + 2 * cp;

    BFIR_STAMP(1, 0);
    // both spectra into LDS: Ya at [0, L), Yb at [L, 2L)  (float2 units), 16 bytes per lane
    {
        const float4 *__restrict__ ya = (const float4 *)(a.y + (long)gc * a.y_ch_stride + (long)t * N);
        const float4 *__restrict__ yb = (const float4 *)(a.y + (long)(gc + 1) * a.y_ch_stride + (long)t * N);
        float4 *l4 = (float4 *)lds;
#pragma unroll
        for (int j = 0; j < P / 4; j++) {
            const int idx = tid + j * NT;                              // < L/2 float4 per spectrum
            l4[idx] = ya[idx];
            l4[L / 2 + idx] = yb[idx];
        }
    }
    __syncthreads();
    BFIR_STAMP(1, 9);
    // Z[k] = Ya[k] + i Yb[k], Hermitian-extended to the full circle
    float re[P], im[P];
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int k = F::in_index(tid, e);
        const int kk = (k <= L) ? k : N - k;
        const float2 pa = lds[kk == L ? 0 : kk], pb = lds[L + (kk == L ? 0 : kk)];
        float zr, zi;
        if (k == 0)      { zr = pa.x; zi = pb.x; }                     // DC of both channels
        else if (k == L) { zr = pa.y; zi = pb.y; }                     // Nyquist of both channels
        else if (k < L)  { zr = pa.x - pb.y; zi = pa.y + pb.x; }
        else             { zr = pa.x + pb.y; zi = pb.x - pa.y; }       // conj Ya + i conj Yb
        re[e] = zr * a.scale; im[e] = zi * a.scale;
    }
    pin_registers(re, im);   // every read of the staged spectra happens before run()'s first barrier
    BFIR_STAMP(1, 1);

    F::run(re, im, lds, tw, tid);
    BFIR_STAMP(1, 10);

    // first L samples are the valid half (the taps sit in the upper half of their blocks)
    float *__restrict__ out = a.raw + (long)g * a.eng_stride + (a.frame_off + (long)t * L) * C + 2 * cp;
    const float rmax = a.max, rmin = -a.max;
    unsigned int mx0 = 0u, mx1 = 0u, c0 = 0u, c1 = 0u;
#pragma unroll
    for (int e = 0; e < P; e++) {
        const int n = F::out_index(tid, e);
        if (n < L) {
            float2 v; v.x = re[e]; v.y = im[e];
            *(float2 *)(out + (long)n * C) = v;
            // brutefir/real2raw.cpp:321-336: strict compares, NaN never counts
            c0 += ((v.x < 0.f) ? (v.x < rmin) : (v.x > rmax)) ? 1u : 0u;
            c1 += ((v.y < 0.f) ? (v.y < rmin) : (v.y > rmax)) ? 1u : 0u;
            const unsigned int b0 = (v.x == v.x) ? __float_as_uint(fabsf(v.x)) : 0u;
            const unsigned int b1 = (v.y == v.y) ? __float_as_uint(fabsf(v.y)) : 0u;
            mx0 = b0 > mx0 ? b0 : mx0; mx1 = b1 > mx1 ? b1 : mx1;
            // brutefir/brutefir.cpp:316-321: only sample 0 of each block is checked
            if (n == 0 && !(isfinite(v.x) && isfinite(v.y))) atomicMin(a.bad_block, a.block_base + t);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int m0 = __shfl_xor(mx0, o), m1 = __shfl_xor(mx1, o);
        mx0 = m0 > mx0 ? m0 : mx0; mx1 = m1 > mx1 ? m1 : mx1;
        c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o);
    }
    if ((tid & 63) == 0) { red_max[tid >> 6][0] = mx0; red_max[tid >> 6][1] = mx1; red_cnt[tid >> 6][0] = c0; red_cnt[tid >> 6][1] = c1; }
    __syncthreads();
    if (tid < 2) {
        unsigned int m = 0u, n = 0u;
        for (int wv = 0; wv < (NT + 63) / 64; wv++) { m = red_max[wv][tid] > m ? red_max[wv][tid] : m; n += red_cnt[wv][tid]; }
        DevOverflow *of = a.overflow + (gc + tid);
        if (n) atomicAdd(&of->n_overflows, n);
        // filtered: the peak only ever grows, a stale read costs an extra atomic, never a wrong result
        if ((unsigned long long)m > *(volatile unsigned long long *)&of->largest_bits)
            atomicMax(&of->largest_bits, (unsigned long long)m);
    }
    BFIR_STAMP(1, 11);
}

}  // namespace

#define BFIR_FOR_PAIR_LOG2N(F) F(10) F(11) F(12) F(13) F(14)

bool pair_supported(int filter_length)
{
    return filter_length >= 512 && filter_length <= 8192;             // N = 2L complex points, 2^10 .. 2^14 (whole waves)
}

void launch_fwd_pair(const FftPlan &plan, const FwdPairArgs &a, hipStream_t s)
{
    const int items = a.n_t * a.n_eng * (a.C / 2);
    if (items <= 0) return;
    switch (plan.log2m) {
#define F(lg) case lg: hipLaunchKernelGGL((k_fwd_pair<lg>), dim3(items), dim3(FftCfg<lg>::NT), 0, s, a, (const float2 *)plan.tw); break;
        BFIR_FOR_PAIR_LOG2N(F)
#undef F
    }
}

void launch_inv_pair(const FftPlan &plan, const InvPairArgs &a, hipStream_t s)
{
    const int items = a.n_t * a.n_eng * (a.C / 2);
    if (items <= 0) return;
    switch (plan.log2m) {
#define F(lg) case lg: hipLaunchKernelGGL((k_inv_pair<lg>), dim3(items), dim3(FftCfg<lg>::NT), 0, s, a, (const float2 *)plan.tw); break;
        BFIR_FOR_PAIR_LOG2N(F)
#undef F
    }
}

}  // namespace bfir

#ifdef BFIR_TRACE
// tuning builds only: phase stamps of this file's kernels (0 k_fwd_pair, 1 k_inv_pair)
extern "C" int bfir_debug_read_trace_pair(int kern, unsigned long long *out, int n_wgs)
{
    if (kern < 0 || kern > 1 || n_wgs > BFIR_TRACE_WGS) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * n_wgs * BFIR_TRACE_SLOTS,
                                    sizeof(unsigned long long) * kern * BFIR_TRACE_WGS * BFIR_TRACE_SLOTS,
                                    hipMemcpyDeviceToHost);
}
#endif
