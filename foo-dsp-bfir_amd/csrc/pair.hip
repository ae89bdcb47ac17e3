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
#include <cstdlib>

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
            if (n == 0 && !(isfinite(v.x) && isfinite(v.y))) flag_bad(a, t);
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
        DevOverflow *of = of_shard(a.overflow, a.of_shard_stride) + (gc + tid);
        if (n) atomicAdd(&of->n_overflows, n);
        // filtered: the peak only ever grows, a stale read costs an extra atomic, never a wrong result
        if ((unsigned long long)m > *(volatile unsigned long long *)&of->largest_bits)
            atomicMax(&of->largest_bits, (unsigned long long)m);
    }
    BFIR_STAMP(1, 11);
}


// ---- persistent forward kernel -------------------------------------------------------------------
// What k_fwd_pair waits for is memory, not arithmetic (profiles/r02_alias_and_bottleneck_experiments.txt:
// 0.235 ms of VALU + LDS under 0.33 ms of exposed load / store / twiddle time per 4096 blocks).
// Here ONE workgroup transforms a run of consecutive blocks of one channel pair:
//   * every block is loaded once: the half-window "current block" of transform t is the "previous
//     block" of transform t+1 and stays in registers (same thread: n -> n - L keeps the lane);
//   * the next block is fetched right after the first butterflies of the current transform and is
//     only waited for when that transform has stored its spectra, which in turn drain while the
//     next transform computes (a workgroup that ends with its stores keeps its CU slot until they land);
//   * twiddles never touch memory in steady state: the bases of the last pass live in registers, those
//     of the passes before it in 9 KB of LDS, for the workgroup's life (LdsFft::butterflies_tb), so
//     nothing queues behind the prefetch in vmcnt order.
// Same arithmetic as k_fwd_pair except that a derived twiddle carries one more rounding.
// Buffer addressing for the persistent kernels: a wave-uniform descriptor (4 SGPRs) + one 32-bit lane
// offset + a scalar offset per access, instead of a 64-bit address pair in VGPRs per access (the
// one-transform kernels spend a quarter of their vector instructions and ~20 registers on those).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    // the pointer is workgroup-uniform by construction; readfirstlane tells the compiler so
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    float2 f; f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y);
    return f;
}
template <int AUX = 0>   // cache policy bits of the instruction: 2 = nt
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 f)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 v; v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, AUX);
}
__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 f)
{
    u32x2 v; v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// Prefetch loads the compiler must not wait for.  hipcc's s_waitcnt bookkeeping gives up at the head of
// the transform loop ("vmcnt(0)": wait for everything, i.e. also for the spectrum stores of the previous
// transform, which is exactly the overlap this kernel exists for).  So the in-loop prefetch is issued from
// an asm statement (invisible to that bookkeeping) and waited for by hand with a counted vmcnt: vector
// memory operations retire in issue order, so "all but the N youngest" with N = the stores issued after
// the prefetch is precisely "the prefetch has landed".  (cdna_hip_programming.md 5.7 form (ii): "+v"
// operands on load and wait statements pin the order; the .s is audited for copies in between.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc_words(const void *p, unsigned bytes)
{
    const unsigned long long u = (unsigned long long)p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));   // stride 0
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
// 8 x 8 bytes per lane: d[e] <- buffer[voff + e * step]
__device__ __forceinline__ void asm_prefetch8x2(u32x2 (&d)[8], unsigned voff, i32x4 r, unsigned step)
{
    const unsigned s1 = __builtin_amdgcn_readfirstlane(step), s2 = 2 * s1, s3 = 3 * s1, s4 = 4 * s1, s5 = 5 * s1, s6 = 6 * s1, s7 = 7 * s1;
    asm volatile("s_nop 4\n\t"
                 "buffer_load_dwordx2 %0, %8, %9, 0 offen\n\t"
                 "buffer_load_dwordx2 %1, %8, %9, %10 offen\n\t"
                 "buffer_load_dwordx2 %2, %8, %9, %11 offen\n\t"
                 "buffer_load_dwordx2 %3, %8, %9, %12 offen\n\t"
                 "buffer_load_dwordx2 %4, %8, %9, %13 offen\n\t"
                 "buffer_load_dwordx2 %5, %8, %9, %14 offen\n\t"
                 "buffer_load_dwordx2 %6, %8, %9, %15 offen\n\t"
                 "buffer_load_dwordx2 %7, %8, %9, %16 offen"
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                 : "v"(voff), "s"(r), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7)
                 : "memory");
}
// 8 x 16 bytes per lane: a[j] <- buffer A[voff + j * step], b[j] <- buffer B[...]   (j < 4)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void asm_prefetch2x4x4(u32x4 (&a)[4], u32x4 (&b)[4], unsigned voff, i32x4 ra, i32x4 rb, unsigned step)
{
    const unsigned s1 = __builtin_amdgcn_readfirstlane(step), s2 = 2 * s1, s3 = 3 * s1;
#if BFIR_NT_Y & 2
#define BFIR_YLD_POLICY " nt"
#else
#define BFIR_YLD_POLICY ""
#endif
    asm volatile("s_nop 4\n\t"
                 "buffer_load_dwordx4 %0, %8, %9, 0 offen" BFIR_YLD_POLICY "\n\t"
                 "buffer_load_dwordx4 %4, %8, %10, 0 offen" BFIR_YLD_POLICY "\n\t"
                 "buffer_load_dwordx4 %1, %8, %9, %11 offen" BFIR_YLD_POLICY "\n\t"
                 "buffer_load_dwordx4 %5, %8, %10, %11 offen" BFIR_YLD_POLICY "\n\t"
                 "buffer_load_dwordx4 %2, %8, %9, %12 offen" BFIR_YLD_POLICY "\n\t"
                 "buffer_load_dwordx4 %6, %8, %10, %12 offen" BFIR_YLD_POLICY "\n\t"
                 "buffer_load_dwordx4 %3, %8, %9, %13 offen" BFIR_YLD_POLICY "\n\t"
                 "buffer_load_dwordx4 %7, %8, %10, %13 offen" BFIR_YLD_POLICY
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "v"(voff), "s"(ra), "s"(rb), "s"(s1), "s"(s2), "s"(s3)
                 : "memory");
}
// 2 x 8 x 4 bytes per lane (time pairs: one channel's samples of two blocks): b[e] <- buffer B[voff + e * step], c[e] <- buffer C[...]
__device__ __forceinline__ void asm_prefetch2x8x1(unsigned (&b)[8], unsigned (&c)[8], unsigned voff, i32x4 rb, i32x4 rc, unsigned step)
{
    const unsigned s1 = __builtin_amdgcn_readfirstlane(step), s2 = 2 * s1, s3 = 3 * s1, s4 = 4 * s1, s5 = 5 * s1, s6 = 6 * s1, s7 = 7 * s1;
    asm volatile("s_nop 4\n\t"
                 "buffer_load_dword %0, %16, %17, 0 offen\n\t"
                 "buffer_load_dword %8, %16, %18, 0 offen\n\t"
                 "buffer_load_dword %1, %16, %17, %19 offen\n\t"
                 "buffer_load_dword %9, %16, %18, %19 offen\n\t"
                 "buffer_load_dword %2, %16, %17, %20 offen\n\t"
                 "buffer_load_dword %10, %16, %18, %20 offen\n\t"
                 "buffer_load_dword %3, %16, %17, %21 offen\n\t"
                 "buffer_load_dword %11, %16, %18, %21 offen\n\t"
                 "buffer_load_dword %4, %16, %17, %22 offen\n\t"
                 "buffer_load_dword %12, %16, %18, %22 offen\n\t"
                 "buffer_load_dword %5, %16, %17, %23 offen\n\t"
                 "buffer_load_dword %13, %16, %18, %23 offen\n\t"
                 "buffer_load_dword %6, %16, %17, %24 offen\n\t"
                 "buffer_load_dword %14, %16, %18, %24 offen\n\t"
                 "buffer_load_dword %7, %16, %17, %25 offen\n\t"
                 "buffer_load_dword %15, %16, %18, %25 offen"
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                 : "v"(voff), "s"(rb), "s"(rc), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7)
                 : "memory");
}
template <int N> __device__ __forceinline__ void asm_wait_vmcnt(unsigned (&b)[8], unsigned (&c)[8])
{
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                 : "n"(N) : "memory");
}
// wait until all but the N youngest vector-memory operations of this wave are done; names the prefetch
// destinations so that no use of them is scheduled above it
template <int N, typename V, int K> __device__ __forceinline__ void asm_wait_vmcnt(V (&d)[K])
{
    static_assert(K == 8, "eight destinations");
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                 : "n"(N) : "memory");
}
template <int N, typename V> __device__ __forceinline__ void asm_wait_vmcnt(V (&a)[4], V (&b)[4])
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(N) : "memory");
}

template <int LOG2N>
__global__ __launch_bounds__(FftCfg<LOG2N>::NT, 4) void k_fwd_pair_ps(FwdPairArgs a, const float2 *__restrict__ twb, int run_len)
{
    using F = LdsFft<float, LOG2N, -1>;
    constexpr int N = F::M, NT = F::NT, P = F::P, L = N / 2, H = P / 2;   // H points per thread and block
    static_assert(F::radix(0) == 16 && P == 16, "in_index(tid, e) = tid + e * (N / 16)");
    static_assert(F::phys(32) == 33 && F::phys(N - 1) == N - 1 + N / 32 - 1, "the split step's addresses assume i + (i >> 5)");
    __shared__ __attribute__((aligned(16))) float2 lds[F::LDS_ELEMS + 1];   // + 1: see the split step
    __shared__ __attribute__((aligned(16))) float2 ldsb[F::LDSB_ELEMS];

    const int tid = threadIdx.x;
    const int w = xcd_remap();
    const int half_c = a.C / 2, pairs = a.n_eng * half_c;
    const int rr = w / pairs, pp = w - rr * pairs;                       // run, pair: the pairs of a run share an XCD
    // The two-for-one split halves every sum (Xa = (Z[k] + conj Z[N-k]) / 2 ...): the half goes into the input
    // scale instead, which is exact (a power of two commutes with every rounding of the transform), and
    // the two purely real bins, which the split takes unhalved, are doubled back (exact again).
    const float sc = 0.5f * a.scale;
    const int g = pp / half_c, cp = pp - g * half_c;
    const int C = a.C;
    const int t0 = rr * run_len, t1 = min(a.n_t, t0 + run_len);
    if (t0 >= t1) return;

    float2 B[F::NBREG];
    F::load_bases(B, ldsb, twb, tid);                                    // the first exchange's barrier publishes ldsb

    // frame n = tid + e NT of a block of frames sits at byte  blk + 4 (e NT C + tid C)  (+ 8 cp for the pair)
#ifdef BFIR_EXPERIMENT_PLANAR_IO
    // Timing experiment (profiles/r02_frame_io_experiment.txt): what would whole-line frame I/O be worth?  Each
    // pair's L x 2 floats of a block are taken from / put to a contiguous eighth... quarter of the block's bytes
    // (results garbage, byte counts and instruction streams those of the product kernel).
    const float *__restrict__ raw = a.raw + (long)g * a.eng_stride + a.frame_off * C + (long)cp * L * 2;
    const long hist = (long)g * a.hist_eng_stride + (long)cp * L * 2;
    const unsigned blk_bytes = (unsigned)L * 2 * 4u;
    const unsigned fo = (unsigned)tid * 2u * 4u;
    const unsigned estep = (unsigned)NT * 2 * 4u;
#else
    const float *__restrict__ raw = a.raw + (long)g * a.eng_stride + a.frame_off * C + 2 * cp;
    const long hist = (long)g * a.hist_eng_stride + 2 * cp;
    const unsigned blk_bytes = (unsigned)L * C * 4u;
    const unsigned fo = (unsigned)tid * (unsigned)C * 4u;                // lane offset into a block of frames
    const unsigned estep = (unsigned)NT * C * 4u;                        // bytes between a thread's consecutive points
#endif
    static_assert(H == 8, "the prefetch statement moves eight points per thread");
    float2 cur[H];
    u32x2 nxt[H];                                                        // raw frames of the next block, as loaded
    {
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((t0 == 0) ? a.prev + hist : raw + (long)(t0 - 1) * L * C, blk_bytes);
#pragma unroll
        for (int e = 0; e < H; e++) {
            const float2 v = buf_load2(ro, fo, e * estep);
            cur[e].x = v.x * sc; cur[e].y = v.y * sc;
        }
        if (a.n_t == 1) {                                                // one-block chunk: the other history block moves on unchanged
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.carry + hist, blk_bytes), rp = make_rsrc(a.save_prev + hist, blk_bytes);
#pragma unroll
            for (int e = 0; e < H; e++) buf_store2(rp, fo, e * estep, buf_load2(rc, fo, e * estep));
        }
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(raw + (long)t0 * L * C, blk_bytes);
#pragma unroll
        for (int e = 0; e < H; e++) nxt[e] = __builtin_amdgcn_raw_buffer_load_b64(rb, fo, e * estep, 0);
        // these (compiler-counted) loads are consumed before the loop, so inside it the compiler has no
        // pending load of its own to wait for
#pragma unroll
        for (int e = 0; e < H; e++) asm volatile("" : "+v"(nxt[e]));
    }
    float *__restrict__ da0 = a.dst + (long)(g * C + 2 * cp) * a.dst_ch_stride;
    for (int t = t0; t < t1; t++) {
        float re[P], im[P];
        // block t has landed when all but the 2 (P/4) spectrum stores issued after its prefetch are done
        asm_wait_vmcnt<2 * (P / 4)>(nxt);
        if (t >= a.n_t - 2) {                                            // uniform: the chunk's last two blocks
            // the engine's history: raw frames of the last two blocks of the chunk (stores only: nothing in
            // this branch for the compiler to wait for, and it lies outside the window the counted wait spans)
            const __amdgpu_buffer_rsrc_t rk = make_rsrc((t == a.n_t - 1 ? a.save_last : a.save_prev) + hist, blk_bytes);
#pragma unroll
            for (int e = 0; e < H; e++) __builtin_amdgcn_raw_buffer_store_b64(nxt[e], rk, fo, e * estep, 0);
        }
#pragma unroll
        for (int e = 0; e < H; e++) {
            float2 v; v.x = __uint_as_float(nxt[e].x); v.y = __uint_as_float(nxt[e].y);   // raw frames of block t
            re[e] = cur[e].x; im[e] = cur[e].y;                          // window = [block t-1 | block t]
            cur[e].x = v.x * sc; cur[e].y = v.y * sc;
            re[H + e] = cur[e].x; im[H + e] = cur[e].y;
        }
        {
            float2 w0[1];
            F::template butterflies<0>(re, im, w0);
        }
        static_for<1, F::NP>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            // LDS addresses are formed afresh per phase from an opaque copy of the lane index: otherwise
            // the compiler hoists a dozen of them out of the transform loop and spills them around it
            int tl = tid; asm volatile("" : "+v"(tl));
            F::template exchange<S - 1>(re, im, lds, tl);
            F::template butterflies_tb<S>(re, im, B, ldsb, tl);
            if constexpr (S == 1) {
                // fetch block t+1 under the remaining passes (after the widest butterfly); past the end of
                // the run the descriptor has zero bytes and the loads return zeros (no branch, see above)
                asm_prefetch8x2(nxt, fo, make_rsrc_words(raw + (long)(t + 1) * L * C, t + 1 < t1 ? blk_bytes : 0u), estep);
            }
        });

        // Z in natural order to LDS, then two-for-one split.  Every LDS address below is one per-thread base
        // plus a compile-time offset (i + (i >> 5) is additive for offsets that are multiples of 32).
        int tz = tid; asm volatile("" : "+v"(tz));
        __syncthreads();
        {
            float2 *zw = lds + F::phys(tz);
            static_for<0, P>([&](auto E_) {
                constexpr int e = decltype(E_)::value;
                static_assert(F::out_index(0, e) % 32 == 0, "additive padding");
                float2 v; v.x = re[e]; v.y = im[e];
                zw[F::phys(F::out_index(0, e))] = v;
            });
        }
        __syncthreads();
        const long slot = (long)((a.base_slot + t) % a.ring) * N;
        const __amdgpu_buffer_rsrc_t rxa = make_rsrc(da0 + slot, (unsigned)N * 4u), rxb = make_rsrc(da0 + a.dst_ch_stride + slot, (unsigned)N * 4u);
        // A thread splits two ADJACENT bins (k = 2 m, 2 m + 1; m = tid + j NT) so that each spectrum store is
        // 16 bytes per lane: vector stores are issue-bound per instruction on this chip (8-byte stores run at
        // about 7 B/clk/CU, MI355X_MICROARCH.md cycle table), and this kernel stores twice what it loads.
        // Z[k], Z[k+1] sit at zk[phys(2 NT j)], + 1.  Z[N-k-1] sits at zn1[-phys(2 NT j)]: N-1-2 tid has
        // its low five bits free of borrows; Z[N-k] one element further, two where N-k is a multiple of 32 (a pad
        // element lies between).  Lane 0 at j = 0 reads Z[N] = lds[LDS_ELEMS], the spare element: its
        // bin 0 is replaced below.
        const float2 *zk = lds + F::phys(2 * tz);
        constexpr int OMAX = F::phys(2 * NT * (P / 4 - 1));              // bases lowered so that every offset is >= 0
        const float2 *zn1 = lds + (F::phys(N - 1 - 2 * tz) - OMAX);
        const float2 *zn0 = zn1 + ((tz & 15) == 0 ? 2 : 1);
#pragma unroll
        for (int j = 0; j < P / 4; j++) {
            // one pair at a time: left alone the scheduler fetches all sixteen Z values first (32 registers
            // on top of the prefetch and the carried block)
            if (j > 0) __builtin_amdgcn_sched_barrier(0);
            const int o = F::phys(2 * NT * j);
            const float2 zk0 = zk[o], zk1 = zk[o + 1];
            const float2 zn0v = zn0[OMAX - o], zn1v = zn1[OMAX - o];
            float4 xa, xb;
            xa.x = zk0.x + zn0v.x; xa.y = zk0.y - zn0v.y;
            xb.x = zk0.y + zn0v.y; xb.y = zn0v.x - zk0.x;
            xa.z = zk1.x + zn1v.x; xa.w = zk1.y - zn1v.y;
            xb.z = zk1.y + zn1v.y; xb.w = zn1v.x - zk1.x;
            if (j == 0) {                                                // m = tid: bin 0 is DC | Nyquist, both real
                const float2 zh = lds[F::phys(L)];
                const bool k0 = tz == 0;
                xa.x = k0 ? 2.f * zk0.x : xa.x; xa.y = k0 ? 2.f * zh.x : xa.y;
                xb.x = k0 ? 2.f * zk0.y : xb.x; xb.y = k0 ? 2.f * zh.y : xb.y;
            }
            buf_store4<(BFIR_NT_X & 1) ? 2 : 0>(rxa, (unsigned)tz * 16u, (unsigned)(j * NT) * 16u, xa);
            buf_store4<(BFIR_NT_X & 1) ? 2 : 0>(rxb, (unsigned)tz * 16u, (unsigned)(j * NT) * 16u, xb);
        }
        // the next transform's first exchange starts with a barrier, which also protects these LDS reads
    }
    // The last prefetch (zero-byte descriptor: it returns zeros) is still in flight and will write nxt[]:
    // those registers must not be handed to anything else before it has landed.
    asm_wait_vmcnt<0>(nxt);
}

// ---- persistent forward kernel, pairs in TIME ------------------------------------------------------------
// The same two-for-one transform for engines whose channels cannot be paired (an odd channel count, one channel --
// the 8-GPU point of a channel-sharded stream): blocks t and t + 1 of ONE channel are the real and the imaginary part,
//
//     z[n] = w_t[n] + i w_{t+1}[n],     w_t = [block t-1 | block t]   (the window brutefir.cpp:255-263 transforms),
//
// so that X_a is block t's delay-line spectrum and X_b block t + 1's -- channels never mix (brutefir.cpp:252-334), and
// the split, the scaling and every rounding are those of the channel-pair kernel: the spectra are the SAME numbers
// whichever way the blocks are paired up.  One workgroup walks a run of an even number of blocks of one channel; a
// thread keeps its 8 samples of block t + 1 (the next window's "previous block") in registers, fetches blocks
// t + 2 and t + 3 (4-byte samples at the frame stride) under the passes, same hand-counted wait as above.  When the
// chunk ends on an odd block the imaginary part is zero (zero-byte descriptor) and its spectrum is not stored (ditto).
template <int LOG2N>
__global__ __launch_bounds__(FftCfg<LOG2N>::NT, 4) void k_fwd_tp_ps(FwdPairArgs a, const float2 *__restrict__ twb, int run_len)
{
    using F = LdsFft<float, LOG2N, -1>;
    constexpr int N = F::M, NT = F::NT, P = F::P, L = N / 2, H = P / 2;
    static_assert(F::radix(0) == 16 && P == 16, "in_index(tid, e) = tid + e * (N / 16)");
    static_assert(F::phys(32) == 33 && F::phys(N - 1) == N - 1 + N / 32 - 1, "the split step's addresses assume i + (i >> 5)");
    __shared__ __attribute__((aligned(16))) float2 lds[F::LDS_ELEMS + 1];
    __shared__ __attribute__((aligned(16))) float2 ldsb[F::LDSB_ELEMS];

    const int tid = threadIdx.x;
    const int w = xcd_remap();
    const int C = a.C, units = a.n_eng * C;
    const int rr = w / units, pp = w - rr * units;                       // run, channel: the channels of a run share an XCD
    const int g = pp / C, c = pp - g * C;
    const float sc = 0.5f * a.scale;                                     // the split's 1/2, see k_fwd_pair_ps
    const int t0 = rr * run_len, t1 = min(a.n_t, t0 + run_len);          // run_len is even
    if (t0 >= t1) return;

    float2 B[F::NBREG];
    F::load_bases(B, ldsb, twb, tid);

    const float *__restrict__ raw = a.raw + (long)g * a.eng_stride + a.frame_off * C + c;
    const long hist = (long)g * a.hist_eng_stride + c;
    const unsigned blk_bytes = (unsigned)L * C * 4u;
    const unsigned fo = (unsigned)tid * (unsigned)C * 4u;                // lane offset into a block of frames
    const unsigned estep = (unsigned)NT * C * 4u;                        // bytes between a thread's consecutive samples
    static_assert(H == 8, "the prefetch statement moves eight samples per thread and block");
    float prv[H];                                                        // block t - 1, scaled
    unsigned nb[H], nc[H];                                               // raw samples of blocks t and t + 1, as loaded
    {
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((t0 == 0) ? a.prev + hist : raw + (long)(t0 - 1) * L * C, blk_bytes);
#pragma unroll
        for (int e = 0; e < H; e++) prv[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, fo, e * estep, 0)) * sc;
        if (a.n_t == 1) {                                                // one-block chunk: the other history block moves on unchanged
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.carry + hist, blk_bytes), rp = make_rsrc(a.save_prev + hist, blk_bytes);
#pragma unroll
            for (int e = 0; e < H; e++)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_raw_buffer_load_b32(rc, fo, e * estep, 0), rp, fo, e * estep, 0);
        }
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(raw + (long)t0 * L * C, blk_bytes);
        const __amdgpu_buffer_rsrc_t rn = make_rsrc(raw + (long)(t0 + 1) * L * C, t0 + 1 < t1 ? blk_bytes : 0u);
#pragma unroll
        for (int e = 0; e < H; e++) {
            nb[e] = __builtin_amdgcn_raw_buffer_load_b32(rb, fo, e * estep, 0);
            nc[e] = __builtin_amdgcn_raw_buffer_load_b32(rn, fo, e * estep, 0);
        }
        // consumed before the loop, so inside it the compiler has no pending load of its own to wait for
#pragma unroll
        for (int e = 0; e < H; e++) asm volatile("" : "+v"(nb[e]), "+v"(nc[e]));
    }
    float *__restrict__ da0 = a.dst + (long)(g * C + c) * a.dst_ch_stride;
    for (int t = t0; t < t1; t += 2) {
        float re[P], im[P];
        // blocks t, t + 1 have landed when all but the 2 (P/4) spectrum stores issued after their prefetch are done
        asm_wait_vmcnt<2 * (P / 4)>(nb, nc);
        if (t + 1 >= a.n_t - 2) {                                        // uniform: one of the chunk's last two blocks is here
            // the engine's history: the raw samples of the last two blocks of the chunk (stores only, outside the
            // window the counted wait spans)
            if (t >= a.n_t - 2) {
                const __amdgpu_buffer_rsrc_t rk = make_rsrc((t == a.n_t - 1 ? a.save_last : a.save_prev) + hist, blk_bytes);
#pragma unroll
                for (int e = 0; e < H; e++) __builtin_amdgcn_raw_buffer_store_b32(nb[e], rk, fo, e * estep, 0);
            }
            if (t + 1 < a.n_t) {
                const __amdgpu_buffer_rsrc_t rk = make_rsrc((t + 1 == a.n_t - 1 ? a.save_last : a.save_prev) + hist, blk_bytes);
#pragma unroll
                for (int e = 0; e < H; e++) __builtin_amdgcn_raw_buffer_store_b32(nc[e], rk, fo, e * estep, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < H; e++) {
            const float vb = __uint_as_float(nb[e]) * sc, vc = __uint_as_float(nc[e]) * sc;
            re[e] = prv[e]; im[e] = vb;                                  // windows [t-1 | t] and [t | t+1]
            re[H + e] = vb; im[H + e] = vc;
            prv[e] = vc;
        }
        {
            float2 w0[1];
            F::template butterflies<0>(re, im, w0);
        }
        static_for<1, F::NP>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            int tl = tid; asm volatile("" : "+v"(tl));
            F::template exchange<S - 1>(re, im, lds, tl);
            F::template butterflies_tb<S>(re, im, B, ldsb, tl);
            if constexpr (S == 1) {
                // blocks t + 2 and t + 3 under the remaining passes; past the end of the run the descriptors have
                // zero bytes and the loads return zeros (no branch)
                asm_prefetch2x8x1(nb, nc, fo, make_rsrc_words(raw + (long)(t + 2) * L * C, t + 2 < t1 ? blk_bytes : 0u),
                                  make_rsrc_words(raw + (long)(t + 3) * L * C, t + 3 < t1 ? blk_bytes : 0u), estep);
            }
        });

        // Z in natural order to LDS, then two-for-one split (k_fwd_pair_ps)
        int tz = tid; asm volatile("" : "+v"(tz));
        __syncthreads();
        {
            float2 *zw = lds + F::phys(tz);
            static_for<0, P>([&](auto E_) {
                constexpr int e = decltype(E_)::value;
                static_assert(F::out_index(0, e) % 32 == 0, "additive padding");
                float2 v; v.x = re[e]; v.y = im[e];
                zw[F::phys(F::out_index(0, e))] = v;
            });
        }
        __syncthreads();
        const long slot_a = (long)((a.base_slot + t) % a.ring) * N, slot_b = (long)((a.base_slot + t + 1) % a.ring) * N;
        const __amdgpu_buffer_rsrc_t rxa = make_rsrc(da0 + slot_a, (unsigned)N * 4u);
        const __amdgpu_buffer_rsrc_t rxb = make_rsrc(da0 + slot_b, t + 1 < a.n_t ? (unsigned)N * 4u : 0u);   // odd end: dropped
        const float2 *zk = lds + F::phys(2 * tz);
        constexpr int OMAX = F::phys(2 * NT * (P / 4 - 1));
        const float2 *zn1 = lds + (F::phys(N - 1 - 2 * tz) - OMAX);
        const float2 *zn0 = zn1 + ((tz & 15) == 0 ? 2 : 1);
#pragma unroll
        for (int j = 0; j < P / 4; j++) {
            if (j > 0) __builtin_amdgcn_sched_barrier(0);
            const int o = F::phys(2 * NT * j);
            const float2 zk0 = zk[o], zk1 = zk[o + 1];
            const float2 zn0v = zn0[OMAX - o], zn1v = zn1[OMAX - o];
            float4 xa, xb;
            xa.x = zk0.x + zn0v.x; xa.y = zk0.y - zn0v.y;
            xb.x = zk0.y + zn0v.y; xb.y = zn0v.x - zk0.x;
            xa.z = zk1.x + zn1v.x; xa.w = zk1.y - zn1v.y;
            xb.z = zk1.y + zn1v.y; xb.w = zn1v.x - zk1.x;
            if (j == 0) {                                                // bin 0 is DC | Nyquist, both real
                const float2 zh = lds[F::phys(L)];
                const bool k0 = tz == 0;
                xa.x = k0 ? 2.f * zk0.x : xa.x; xa.y = k0 ? 2.f * zh.x : xa.y;
                xb.x = k0 ? 2.f * zk0.y : xb.x; xb.y = k0 ? 2.f * zh.y : xb.y;
            }
            buf_store4<(BFIR_NT_X & 1) ? 2 : 0>(rxa, (unsigned)tz * 16u, (unsigned)(j * NT) * 16u, xa);
            buf_store4<(BFIR_NT_X & 1) ? 2 : 0>(rxb, (unsigned)tz * 16u, (unsigned)(j * NT) * 16u, xb);
        }
    }
    // the last prefetch (zero-byte descriptors) is still in flight and will write nb / nc: see k_fwd_pair_ps
    asm_wait_vmcnt<0>(nb, nc);
}

// ---- persistent inverse kernel ---------------------------------------------------------------------
// The same treatment for the way back: one workgroup turns a run of consecutive product spectra of one
// channel pair into output frames.  The next block's two spectra (64 KiB per workgroup) are fetched into
// registers (8 x 16 bytes per thread) under the passes of the current transform and only staged into LDS
// when the passes are done with it; the 8-byte strided output stores drain under the next transform;
// twiddles from bases (LdsFft::butterflies_tb); overflow statistics reduced once per run, not per block.
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    float4 f; f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
    return f;
}

// Minimum waves per SIMD the inverse kernel is compiled for.  Four (128 registers) everywhere but N = 4096, whose
// last pass is a radix-16 one with six base twiddles in registers: at 128 it spills ONE register, and a spill is a
// scratch (vector-memory) operation inside the window the hand-counted vmcnt spans -- the wait would then cover
// one operation too few.  scripts/audit_ps_isa.py (tests/test_isa_audit.py) found it and checks every build.
template <int LOG2N> constexpr int inv_ps_min_waves() { return LOG2N == 12 ? 3 : 4; }

template <int LOG2N>
__global__ __launch_bounds__(FftCfg<LOG2N>::NT, inv_ps_min_waves<LOG2N>()) void k_inv_pair_ps(InvPairArgs a, const float2 *__restrict__ twb, int run_len)
{
    using F = LdsFft<float, LOG2N, +1>;
    constexpr int N = F::M, NT = F::NT, P = F::P, L = N / 2, Q = P / 4;   // Q 16-byte pieces per thread and spectrum
    __shared__ __attribute__((aligned(16))) float2 lds[F::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) float2 ldsb[F::LDSB_ELEMS];
    __shared__ unsigned int red_max[NT / 64 > 0 ? NT / 64 : 1][2], red_cnt[NT / 64 > 0 ? NT / 64 : 1][2];

    const int tid = threadIdx.x;
    const int w = xcd_remap();
    const int half_c = a.C / 2, pairs = a.n_eng * half_c;
    const int rr = w / pairs, pp = w - rr * pairs;
    const int g = pp / half_c, cp = pp - g * half_c;
    const int C = a.C;
    const int gc = g * C + 2 * cp;
    const int t0 = rr * run_len, t1 = min(a.n_t, t0 + run_len);
    if (t0 >= t1) return;

    float2 B[F::NBREG];
    F::load_bases(B, ldsb, twb, tid);

    const float *__restrict__ ya0 = a.y + (long)gc * a.y_ch_stride;
#ifdef BFIR_EXPERIMENT_PLANAR_IO
    float *__restrict__ out0 = a.raw + (long)g * a.eng_stride + a.frame_off * C + (long)cp * L * 2;
    const unsigned blk_bytes = (unsigned)L * 2 * 4u;
    const unsigned cio = 2u;
#else
    float *__restrict__ out0 = a.raw + (long)g * a.eng_stride + a.frame_off * C + 2 * cp;
    const unsigned blk_bytes = (unsigned)L * C * 4u;
    const unsigned cio = (unsigned)C;                                    // floats between a pair's consecutive frames
#endif
    static_assert(Q == 4, "the prefetch statement moves four 16-byte pieces per thread and spectrum");
    u32x4 qa[Q], qb[Q];
    {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(ya0 + (long)BFIR_YSLOT(a, t0) * N, (unsigned)N * 4u);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(ya0 + a.y_ch_stride + (long)BFIR_YSLOT(a, t0) * N, (unsigned)N * 4u);
#pragma unroll
        for (int j = 0; j < Q; j++) {
            qa[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)tid * 16u, (unsigned)(j * NT) * 16u, (BFIR_NT_Y & 2) ? 2 : 0);
            qb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)tid * 16u, (unsigned)(j * NT) * 16u, (BFIR_NT_Y & 2) ? 2 : 0);
        }
        // consumed before the loop: inside it the compiler has no pending load of its own (see k_fwd_pair_ps)
#pragma unroll
        for (int j = 0; j < Q; j++) asm volatile("" : "+v"(qa[j]), "+v"(qb[j]));
    }
    // brutefir/real2raw.cpp:321-336 counts v < 0 ? v < -max : v > max (strict, NaN never counts) and tracks the
    // largest magnitude: with symmetric limits that is |v| > max, and a float maximum that skips NaNs
    // (fmaxf) -- three instructions per sample instead of twelve.
    const float rmax = a.max;
    float pk0 = 0.f, pk1 = 0.f;
    unsigned int c0 = 0u, c1 = 0u;
    int bad = 0x7fffffff;
    for (int t = t0; t < t1; t++) {
        // both spectra into LDS: Ya at [0, L), Yb at [L, 2L)  (float2 units)
        int ts = tid; asm volatile("" : "+v"(ts));
        // block t's spectra have landed when all but the P/2 output stores issued after their prefetch are done
        asm_wait_vmcnt<P / 2>(qa, qb);
        __syncthreads();                                                 // the previous transform's readers are done
        {
            u32x4 *l4 = (u32x4 *)lds;
#pragma unroll
            for (int j = 0; j < Q; j++) { l4[ts + j * NT] = qa[j]; l4[L / 2 + ts + j * NT] = qb[j]; }
        }
        __syncthreads();
        // Z[k] = Ya[k] + i Yb[k], Hermitian-extended to the full circle
        float re[P], im[P];
        static_for<0, P>([&](auto E_) {
            constexpr int e = decltype(E_)::value;
            // k = base + tid with a compile-time base: which side of L a point lies on is known per e,
            // and only e with base 0 / base L can hit the two real bins (tid 0) -- selects, no branches
            constexpr int base = F::in_index(0, e);
            static_assert(base + NT <= L || base >= L, "a thread's points do not straddle L");
            const int k = base + ts;
            const int kk = (base < L) ? k : N - k;                       // kk == L only for base == L, tid == 0
            const bool edge = (base == 0 || base == L) && ts == 0;
            const float2 pa = lds[edge ? 0 : kk], pb = lds[L + (edge ? 0 : kk)];
            float zr, zi;
            if (base < L) { zr = pa.x - pb.y; zi = pa.y + pb.x; }
            else          { zr = pa.x + pb.y; zi = pb.x - pa.y; }         // conj Ya + i conj Yb
            if (base == 0) { zr = edge ? pa.x : zr; zi = edge ? pb.x : zi; }   // DC of both channels
            if (base == L) { zr = edge ? pa.y : zr; zi = edge ? pb.y : zi; }   // Nyquist of both channels
            re[e] = zr * a.scale; im[e] = zi * a.scale;
        });
        pin_registers(re, im);   // every read of the staged spectra happens before the first exchange's barrier
        {
            float2 w0[1];
            F::template butterflies<0>(re, im, w0);
        }
        static_for<1, F::NP>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            int tl = tid; asm volatile("" : "+v"(tl));
            F::template exchange<S - 1>(re, im, lds, tl);
            F::template butterflies_tb<S>(re, im, B, ldsb, tl);
            if constexpr (S == 1) {
                // the next block's spectra, under the remaining passes; past the end of the run the descriptors
                // have zero bytes and the loads return zeros (no branch in the loop, see k_fwd_pair_ps)
                const unsigned nbytes = t + 1 < t1 ? (unsigned)N * 4u : 0u;
                asm_prefetch2x4x4(qa, qb, (unsigned)tl * 16u, make_rsrc_words(ya0 + (long)BFIR_YSLOT(a, t + 1) * N, nbytes),
                                  make_rsrc_words(ya0 + a.y_ch_stride + (long)BFIR_YSLOT(a, t + 1) * N, nbytes), (unsigned)NT * 16u);
            }
        });

        // first L samples are the valid half (the taps sit in the upper half of their blocks)
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(out0 + (long)t * L * C, blk_bytes);
        int to = tid; asm volatile("" : "+v"(to));
#pragma unroll
        for (int e = 0; e < P; e++) {
            const int n = F::out_index(to, e);
            if (F::out_index(0, e) < L) {                                // compile time: out_index(tid, e) = tid + const, tid < NT <= L
                float2 v; v.x = re[e]; v.y = im[e];
                buf_store2(ro, (unsigned)to * cio * 4u, (unsigned)F::out_index(0, e) * cio * 4u, v);
                c0 += (fabsf(v.x) > rmax) ? 1u : 0u;
                c1 += (fabsf(v.y) > rmax) ? 1u : 0u;
                pk0 = fmaxf(pk0, fabsf(v.x)); pk1 = fmaxf(pk1, fabsf(v.y));
                // brutefir/brutefir.cpp:316-321: only sample 0 of each block is checked; the first bad block
                // of the run is reported once, after the loop (no atomic inside it)
                if (F::out_index(0, e) == 0) {
                    const bool nf = n == 0 && !(isfinite(v.x) && isfinite(v.y));
                    bad = (nf && t < bad) ? t : bad;
                }
            }
        }
    }
    // The last prefetch (zero-byte descriptors: it returns zeros) is still in flight and will write qa / qb.
    // Without this wait the compiler reuses those registers below -- for the ADDRESSES of the atomics, which the
    // late zeros then turn into a null pointer (seen as a sporadic "memory access fault on address (nil)").
    asm_wait_vmcnt<0>(qa, qb);
    if (bad != 0x7fffffff) flag_bad(a, bad);
    unsigned int mx0 = __float_as_uint(pk0), mx1 = __float_as_uint(pk1);   // non-negative floats order like their bits
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
        DevOverflow *of = of_shard(a.overflow, a.of_shard_stride) + (gc + tid);
        if (n) atomicAdd(&of->n_overflows, n);
        // filtered: the peak only ever grows, a stale read costs an extra atomic, never a wrong result
        if ((unsigned long long)m > *(volatile unsigned long long *)&of->largest_bits)
            atomicMax(&of->largest_bits, (unsigned long long)m);
    }
}

// ---- persistent inverse kernel, pairs in TIME --------------------------------------------------------------
// Z = Y_t + i Y_{t+1} of ONE channel: Re z is block t's output, Im z block t + 1's (k_fwd_tp_ps).  4-byte samples at
// the frame stride, 16 stores per transform (hence the wait for all but 16), both blocks' overflow statistics go to
// the one channel; at an odd chunk end Y_{t+1} reads as zero and its samples are not stored (zero-byte descriptors).
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float f)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(f), r, voff, soff, 0);
}

template <int LOG2N>
__global__ __launch_bounds__(FftCfg<LOG2N>::NT, inv_ps_min_waves<LOG2N>()) void k_inv_tp_ps(InvPairArgs a, const float2 *__restrict__ twb, int run_len)
{
    using F = LdsFft<float, LOG2N, +1>;
    constexpr int N = F::M, NT = F::NT, P = F::P, L = N / 2, Q = P / 4;
    __shared__ __attribute__((aligned(16))) float2 lds[F::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) float2 ldsb[F::LDSB_ELEMS];
    __shared__ unsigned int red_max[NT / 64 > 0 ? NT / 64 : 1][2], red_cnt[NT / 64 > 0 ? NT / 64 : 1][2];

    const int tid = threadIdx.x;
    const int w = xcd_remap();
    const int C = a.C, units = a.n_eng * C;
    const int rr = w / units, pp = w - rr * units;
    const int g = pp / C, c = pp - g * C;
    const int gc = g * C + c;
    const int t0 = rr * run_len, t1 = min(a.n_t, t0 + run_len);          // run_len is even
    if (t0 >= t1) return;

    float2 B[F::NBREG];
    F::load_bases(B, ldsb, twb, tid);

    const float *__restrict__ ya0 = a.y + (long)gc * a.y_ch_stride;
    float *__restrict__ out0 = a.raw + (long)g * a.eng_stride + a.frame_off * C + c;
    const unsigned blk_bytes = (unsigned)L * C * 4u;
    const unsigned cio = (unsigned)C;                                    // floats between a channel's consecutive samples
    static_assert(Q == 4, "the prefetch statement moves four 16-byte pieces per thread and spectrum");
    u32x4 qa[Q], qb[Q];
    {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(ya0 + (long)BFIR_YSLOT(a, t0) * N, (unsigned)N * 4u);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(ya0 + (long)BFIR_YSLOT(a, t0 + 1) * N, t0 + 1 < t1 ? (unsigned)N * 4u : 0u);
#pragma unroll
        for (int j = 0; j < Q; j++) {
            qa[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)tid * 16u, (unsigned)(j * NT) * 16u, (BFIR_NT_Y & 2) ? 2 : 0);
            qb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)tid * 16u, (unsigned)(j * NT) * 16u, (BFIR_NT_Y & 2) ? 2 : 0);
        }
#pragma unroll
        for (int j = 0; j < Q; j++) asm volatile("" : "+v"(qa[j]), "+v"(qb[j]));
    }
    const float rmax = a.max;
    float pk0 = 0.f, pk1 = 0.f;
    unsigned int c0 = 0u, c1 = 0u;
    int bad = 0x7fffffff;
    for (int t = t0; t < t1; t += 2) {
        int ts = tid; asm volatile("" : "+v"(ts));
        // the two spectra have landed when all but the P output stores issued after their prefetch are done
        asm_wait_vmcnt<P>(qa, qb);
        __syncthreads();                                                 // the previous transform's readers are done
        {
            u32x4 *l4 = (u32x4 *)lds;
#pragma unroll
            for (int j = 0; j < Q; j++) { l4[ts + j * NT] = qa[j]; l4[L / 2 + ts + j * NT] = qb[j]; }
        }
        __syncthreads();
        // Z[k] = Ya[k] + i Yb[k], Hermitian-extended to the full circle (k_inv_pair_ps)
        float re[P], im[P];
        static_for<0, P>([&](auto E_) {
            constexpr int e = decltype(E_)::value;
            constexpr int base = F::in_index(0, e);
            static_assert(base + NT <= L || base >= L, "a thread's points do not straddle L");
            const int k = base + ts;
            const int kk = (base < L) ? k : N - k;
            const bool edge = (base == 0 || base == L) && ts == 0;
            const float2 pa = lds[edge ? 0 : kk], pb = lds[L + (edge ? 0 : kk)];
            float zr, zi;
            if (base < L) { zr = pa.x - pb.y; zi = pa.y + pb.x; }
            else          { zr = pa.x + pb.y; zi = pb.x - pa.y; }
            if (base == 0) { zr = edge ? pa.x : zr; zi = edge ? pb.x : zi; }
            if (base == L) { zr = edge ? pa.y : zr; zi = edge ? pb.y : zi; }
            re[e] = zr * a.scale; im[e] = zi * a.scale;
        });
        pin_registers(re, im);
        {
            float2 w0[1];
            F::template butterflies<0>(re, im, w0);
        }
        static_for<1, F::NP>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            int tl = tid; asm volatile("" : "+v"(tl));
            F::template exchange<S - 1>(re, im, lds, tl);
            F::template butterflies_tb<S>(re, im, B, ldsb, tl);
            if constexpr (S == 1) {
                asm_prefetch2x4x4(qa, qb, (unsigned)tl * 16u,
                                  make_rsrc_words(ya0 + (long)BFIR_YSLOT(a, t + 2) * N, t + 2 < t1 ? (unsigned)N * 4u : 0u),
                                  make_rsrc_words(ya0 + (long)BFIR_YSLOT(a, t + 3) * N, t + 3 < t1 ? (unsigned)N * 4u : 0u), (unsigned)NT * 16u);
            }
        });

        // first L samples are the valid half: Re -> block t, Im -> block t + 1
        const bool has_b = t + 1 < a.n_t;
        const __amdgpu_buffer_rsrc_t roa = make_rsrc(out0 + (long)t * L * C, blk_bytes);
        const __amdgpu_buffer_rsrc_t rob = make_rsrc(out0 + (long)(t + 1) * L * C, has_b ? blk_bytes : 0u);
        int to = tid; asm volatile("" : "+v"(to));
#pragma unroll
        for (int e = 0; e < P; e++) {
            const int n = F::out_index(to, e);
            if (F::out_index(0, e) < L) {
                const float va = re[e], vb = has_b ? im[e] : 0.f;
                buf_store1(roa, (unsigned)to * cio * 4u, (unsigned)F::out_index(0, e) * cio * 4u, va);
                buf_store1(rob, (unsigned)to * cio * 4u, (unsigned)F::out_index(0, e) * cio * 4u, vb);
                c0 += (fabsf(va) > rmax) ? 1u : 0u;
                c1 += (fabsf(vb) > rmax) ? 1u : 0u;
                pk0 = fmaxf(pk0, fabsf(va)); pk1 = fmaxf(pk1, fabsf(vb));
                // brutefir/brutefir.cpp:316-321: only sample 0 of each block is checked
                if (F::out_index(0, e) == 0) {
                    const bool na = n == 0 && !isfinite(va), nbb = n == 0 && has_b && !isfinite(vb);
                    bad = (na && t < bad) ? t : bad;
                    bad = (nbb && t + 1 < bad) ? t + 1 : bad;
                }
            }
        }
    }
    asm_wait_vmcnt<0>(qa, qb);                                           // see k_inv_pair_ps
    if (bad != 0x7fffffff) flag_bad(a, bad);
    unsigned int mx0 = __float_as_uint(fmaxf(pk0, pk1)), mx1 = 0u;       // one channel: its peak over both blocks
    c0 += c1; c1 = 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int m0 = __shfl_xor(mx0, o);
        mx0 = m0 > mx0 ? m0 : mx0;
        c0 += __shfl_xor(c0, o);
    }
    if ((tid & 63) == 0) { red_max[tid >> 6][0] = mx0; red_cnt[tid >> 6][0] = c0; }
    __syncthreads();
    if (tid == 0) {
        unsigned int m = 0u, n = 0u;
        for (int wv = 0; wv < (NT + 63) / 64; wv++) { m = red_max[wv][0] > m ? red_max[wv][0] : m; n += red_cnt[wv][0]; }
        DevOverflow *of = of_shard(a.overflow, a.of_shard_stride) + gc;
        if (n) atomicAdd(&of->n_overflows, n);
        if ((unsigned long long)m > *(volatile unsigned long long *)&of->largest_bits)
            atomicMax(&of->largest_bits, (unsigned long long)m);
    }
    (void)mx1;
}

}  // namespace

#define BFIR_FOR_PAIR_LOG2N(F) F(10) F(11) F(12) F(13) F(14)

bool pair_supported(int filter_length)
{
    return filter_length >= 512 && filter_length <= 8192;             // N = 2L complex points, 2^10 .. 2^14 (whole waves)
}

// Blocks per workgroup of the persistent kernels.  Long runs are best for a kernel on its own (forward:
// 16 blocks) but the engine runs forward, MAC and inverse of neighbouring chunks concurrently, and a
// kernel made of a few hundred long-lived workgroups holds every CU slot until it ends, so the other
// two cannot slip in: the pipeline is fastest with SHORT runs (profiles/r02_pair_run_sweep.txt:
// 108 Gsamples/s at 4 blocks per workgroup against 95 at 32).  BFIR_PAIR_RUN_FWD / _INV override (tuning aid);
// BFIR_PAIR_PERSIST=0 keeps the one-transform-per-workgroup kernels.
static int pair_run_len(int n_t, int pairs, bool inverse)
{
    if (const char *e = getenv(inverse ? "BFIR_PAIR_RUN_INV" : "BFIR_PAIR_RUN_FWD")) return std::max(1, atoi(e));
    const int slots = 512;                                 // 256 CUs x 2 workgroups
    const int runs = std::max(1, slots / std::max(1, pairs));
    const int fill = std::max(1, (n_t + runs - 1) / runs); // what one round of workgroups would need
    return std::min(fill, 4);
}

void launch_fwd_pair(const FftPlan &plan, const FwdPairArgs &a_, hipStream_t s)
{
    FwdPairArgs a = a_;
#ifdef BFIR_EXPERIMENT_ALIAS
    if (const int xa = bfir_alias_env("BFIR_X_ALIAS")) { a.ring = xa; a.base_slot %= xa; }
#endif
    if (a.tp) {                                            // pairs in time: one unit per channel, runs of an even number of blocks
        const int units = a.n_eng * a.C;
        if (a.n_t <= 0 || units <= 0 || !plan.twb) return;
        // as many TRANSFORMS per workgroup as the channel-pair kernels take (a transform is two blocks here): 105.5 against
        // 103.4 Gsamples/s at 7 channels, 104.2 against 101.7 at one (profiles/r03_channels.txt)
        const int len = 2 * pair_run_len((a.n_t + 1) / 2, units, false), runs = (a.n_t + len - 1) / len;
        switch (plan.log2m) {
#define F(lg) case lg: hipLaunchKernelGGL((k_fwd_tp_ps<lg>), dim3(runs * units), dim3(FftCfg<lg>::NT), 0, s, a, (const float2 *)plan.twb, len); break;
            BFIR_FOR_PAIR_LOG2N(F)
#undef F
        }
        return;
    }
    const int items = a.n_t * a.n_eng * (a.C / 2);
    if (items <= 0) return;
    const char *pe = getenv("BFIR_PAIR_PERSIST");
    if (!(pe && atoi(pe) == 0) && plan.twb) {
        const int pairs = a.n_eng * (a.C / 2);
        const int len = pair_run_len(a.n_t, pairs, false), runs = (a.n_t + len - 1) / len;
        switch (plan.log2m) {
#define F(lg) case lg: hipLaunchKernelGGL((k_fwd_pair_ps<lg>), dim3(runs * pairs), dim3(FftCfg<lg>::NT), 0, s, a, (const float2 *)plan.twb, len); break;
            BFIR_FOR_PAIR_LOG2N(F)
#undef F
        }
        return;
    }
    switch (plan.log2m) {
#define F(lg) case lg: hipLaunchKernelGGL((k_fwd_pair<lg>), dim3(items), dim3(FftCfg<lg>::NT), 0, s, a, (const float2 *)plan.tw); break;
        BFIR_FOR_PAIR_LOG2N(F)
#undef F
    }
}

void launch_inv_pair(const FftPlan &plan, const InvPairArgs &a_, hipStream_t s)
{
    InvPairArgs a = a_;
#ifdef BFIR_EXPERIMENT_ALIAS
    if (const int ya = bfir_alias_env("BFIR_Y_ALIAS")) a.y_alias = ya;
#endif
    if (a.tp) {                                            // pairs in time (k_inv_tp_ps)
        const int units = a.n_eng * a.C;
        if (a.n_t <= 0 || units <= 0 || !plan.twb) return;
        const int len = 2 * pair_run_len((a.n_t + 1) / 2, units, true), runs = (a.n_t + len - 1) / len;
        switch (plan.log2m) {
#define F(lg) case lg: hipLaunchKernelGGL((k_inv_tp_ps<lg>), dim3(runs * units), dim3(FftCfg<lg>::NT), 0, s, a, (const float2 *)plan.twb, len); break;
            BFIR_FOR_PAIR_LOG2N(F)
#undef F
        }
        return;
    }
    const int items = a.n_t * a.n_eng * (a.C / 2);
    if (items <= 0) return;
    const char *pe = getenv("BFIR_PAIR_PERSIST");
    if (!(pe && (atoi(pe) == 0 || atoi(pe) == 2)) && plan.twb) {        // 2: persistent forward kernel only (A/B)
        const int pairs = a.n_eng * (a.C / 2);
        const int len = pair_run_len(a.n_t, pairs, true), runs = (a.n_t + len - 1) / len;
        switch (plan.log2m) {
#define F(lg) case lg: hipLaunchKernelGGL((k_inv_pair_ps<lg>), dim3(runs * pairs), dim3(FftCfg<lg>::NT), 0, s, a, (const float2 *)plan.twb, len); break;
            BFIR_FOR_PAIR_LOG2N(F)
#undef F
        }
        return;
    }
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
