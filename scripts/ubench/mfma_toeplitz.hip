// mfma_toeplitz.hip -- can the partition sums  y[t] = sum_{p < B} x[t - p] h[p]  (per bin; brutefir.cpp:288-299) run on the
// matrix cores?  A GO / NO-GO measurement for a later MAC kernel, not product code.
//
// The sums have no second matrix dimension (every bin has its own x AND its own h), so they are no GEMM.  But the batched
// outer-product form  v_mfma_f32_4x4x1_16B_f32  (16 independent 4x4 tiles  D_b += a_b (x) b_b  per wave instruction) fits a
// Toeplitz product exactly when the 16 tiles are 16 CONSECUTIVE TIME TILES OF ONE BIN:
//     a_b[i] = x[T0 - p + 4 b + i]   = x[T0 - p + lane]          (64 consecutive samples of the bin's time series)
//     b_b[j] = h[p + j]              = h[p + (lane & 3)]         (four partitions, the same for every tile)
//     D_b[i][j] += x[T0 - p + 4b + i] h[p + j]                   belongs to output  T0 + 4b + i + j  =  T0 + lane(4b + j) + i
// so after the B / 4 steps p = 0, 4, ...:   y[T0 + m] = sum_i D_i[lane m - i]   -- a Horner chain of three DPP wave_shr:1 adds per
// 64 outputs.  Every product is a needed one (no padding inside the tile); lanes 0 .. 2 miss the terms of the tile before and
// three terms fall off the end, so a wave advances by 60 outputs per pass (61 valid) and needs no carry.
// Complex: four real MFMAs per step (re += xr hr; re += xi (-hi); im += xr hi; im += xi hr).
// The time series of a bin is read from LDS (one ds_read_b32 per operand: consecutive lanes, consecutive words).  NOT covered
// here: the transposes a product kernel needs between the delay line's [block][bin] order and a bin's time series (through LDS,
// both ways), DC / Nyquist, the ring.  Yardstick: the product's k_mac_stream forms the same sums (8 flop per partition, bin
// and output) at 0.497 ms per 4096 blocks of the headline's shape = 69 TFLOP/s, k_mac_sys with 64 partitions at ~66.
// Numerics: each tile entry is a p-ordered fmaf chain, the entries of an output are then added: not the reference's single
// chain, so such a kernel would agree with the others to rounding, not bit for bit.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_toeplitz.hip -o /tmp/mfma_toeplitz && /tmp/mfma_toeplitz
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

// shift one lane up the wave, zero into lane 0 (DPP wave_shr:1, bound_ctrl)
__device__ __forceinline__ float shr1(float v)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x138, 0xf, 0xf, true));
}

constexpr int STEP = 60;        // outputs a wave advances per pass (lanes 3 .. 62 stored)
constexpr int RING = 128;       // LDS words per series and wave (>= 64 + B - 4 + STEP slack), power of two

// x, y: [bin][time] (re, im) pairs -- the time series of a bin contiguous (the layout the transposes would produce in LDS)
// h:    [bin][B] (re, im)
// MODE 0: the whole pass; 1: MFMAs only (operands loaded once); 2: no reduction / stores
template <int B, int MODE>
__global__ __launch_bounds__(256) void k_toeplitz(const float2 *__restrict__ x, const float2 *__restrict__ h, float2 *__restrict__ y,
                                                  int n_time, int hist)
{
    __shared__ float s_re[4][RING], s_im[4][RING];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long bin = (long)blockIdx.x * 4 + w;
    const float2 *xb = x + bin * (long)(n_time + hist) + hist;      // xb[t], t >= -hist
    float2 *yb = y + bin * (long)n_time;
    float hr[B / 4], hi[B / 4], nhi[B / 4];
#pragma unroll
    for (int k = 0; k < B / 4; k++) {
        const float2 v = h[bin * B + 4 * k + (lane & 3)];
        hr[k] = v.x; hi[k] = v.y; nhi[k] = -v.y;
    }
    float *sr = s_re[w], *si = s_im[w];
    // history: times -(B - 4) - 3 .. -1 go into the ring first
    for (int t = -RING + 64 + lane; t < 0; t += 64) {
        const float2 v = t >= -hist ? xb[t] : make_float2(0.f, 0.f);
        sr[t & (RING - 1)] = v.x; si[t & (RING - 1)] = v.y;
    }
    float ar0 = 0, ai0 = 0;
    for (int T0 = -3; T0 < n_time; T0 += STEP) {
        // this pass's new samples T0 + 4 .. T0 + 63 (the first pass also T0 .. T0 + 3)
        {
            const int t = T0 + lane;
            const float2 v = (t >= -hist && t < n_time) ? xb[t] : make_float2(0.f, 0.f);
            sr[t & (RING - 1)] = v.x; si[t & (RING - 1)] = v.y;
        }
        // one wave owns its ring and its LDS operations execute in order: no barrier
        f32x4 dre = {0, 0, 0, 0}, dim = {0, 0, 0, 0};
        if (MODE == 1) { ar0 = sr[lane]; ai0 = si[lane]; }
#pragma unroll
        for (int k = 0; k < B / 4; k++) {
            float ar, ai;
            if (MODE == 1) { ar = ar0; ai = ai0; }
            else { const int a = (T0 - 4 * k + lane) & (RING - 1); ar = sr[a]; ai = si[a]; }
            dre = __builtin_amdgcn_mfma_f32_4x4x1f32(ar, hr[k], dre, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f32_4x4x1f32(ar, hi[k], dim, 0, 0, 0);
            dre = __builtin_amdgcn_mfma_f32_4x4x1f32(ai, nhi[k], dre, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f32_4x4x1f32(ai, hr[k], dim, 0, 0, 0);
        }
        if (MODE == 2) { if (dre[0] == 123.456f && dim[3] == 1.f) yb[0] = make_float2(dre[1], dim[2]); continue; }
        float pr = dre[3], pi = dim[3];
        pr = shr1(pr) + dre[2]; pi = shr1(pi) + dim[2];
        pr = shr1(pr) + dre[1]; pi = shr1(pi) + dim[1];
        pr = shr1(pr) + dre[0]; pi = shr1(pi) + dim[0];
        const int t = T0 + lane;
        if (lane >= 3 && lane < 3 + STEP && t < n_time) yb[t] = make_float2(pr, pi);
    }
}

// The same with  v_mfma_f32_16x16x1_4B_f32  (4 tiles of 16 x 16 per instruction; scripts/ubench/mfma_f32_forms.hip: 15.5 ns per
// instruction and SIMD = 133-136 TFLOP/s even from one wave, against 4.5-6.7 ns = 78-117 for the 4x4x1 form whose issue cost
// shows).  Probed layout: A lane 16 b + i, B lane 16 b + j, D register 4 b + (i & 3), lane 16 (i >> 2) + j.  With
//     a_b[4 g + r] = x[T0 - p + 16 g + 4 b + r]      (the lane's b and g fields swapped: a permuted LDS read, still 64 consecutive words)
//     b_b[j]       = h[p + j]                        (sixteen partitions per step, the same for every tile)
// D[register 4 b + r][lane 16 g + j] belongs to output  T0 + lane + register:  y[T0 + m] = sum_reg D_reg[lane m - reg], a Horner chain
// of fifteen wave_shr:1 adds; lanes 0 .. 14 are incomplete and fifteen terms fall off the end: 49 valid outputs per pass.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int STEP16 = 49;
template <int B, int MODE>
__global__ __launch_bounds__(256) void k_toeplitz16(const float2 *__restrict__ x, const float2 *__restrict__ h, float2 *__restrict__ y,
                                                    int n_time, int hist)
{
    __shared__ float s_re[4][RING], s_im[4][RING];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long bin = (long)blockIdx.x * 4 + w;
    const float2 *xb = x + bin * (long)(n_time + hist) + hist;
    float2 *yb = y + bin * (long)n_time;
    float hr[B / 16], hi[B / 16], nhi[B / 16];
#pragma unroll
    for (int k = 0; k < B / 16; k++) {
        const float2 v = h[bin * B + 16 * k + (lane & 15)];
        hr[k] = v.x; hi[k] = v.y; nhi[k] = -v.y;
    }
    float *sr = s_re[w], *si = s_im[w];
    for (int t = -RING + 64 + lane; t < 0; t += 64) {
        const float2 v = t >= -hist ? xb[t] : make_float2(0.f, 0.f);
        sr[t & (RING - 1)] = v.x; si[t & (RING - 1)] = v.y;
    }
    const int perm = ((lane >> 2) & 3) * 16 + ((lane >> 4) & 3) * 4 + (lane & 3);
    float ar0 = 0, ai0 = 0;
    for (int T0 = -15; T0 < n_time; T0 += STEP16) {
        {
            const int t = T0 + lane;
            const float2 v = (t >= -hist && t < n_time) ? xb[t] : make_float2(0.f, 0.f);
            sr[t & (RING - 1)] = v.x; si[t & (RING - 1)] = v.y;
        }
        f32x16 dre, dim;
#pragma unroll
        for (int r = 0; r < 16; r++) { dre[r] = 0; dim[r] = 0; }
        if (MODE == 1) { ar0 = sr[lane]; ai0 = si[lane]; }
#pragma unroll
        for (int k = 0; k < B / 16; k++) {
            float ar, ai;
            if (MODE == 1) { ar = ar0; ai = ai0; }
            else { const int a = (T0 - 16 * k + perm) & (RING - 1); ar = sr[a]; ai = si[a]; }
            dre = __builtin_amdgcn_mfma_f32_16x16x1f32(ar, hr[k], dre, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f32_16x16x1f32(ar, hi[k], dim, 0, 0, 0);
            dre = __builtin_amdgcn_mfma_f32_16x16x1f32(ai, nhi[k], dre, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f32_16x16x1f32(ai, hr[k], dim, 0, 0, 0);
        }
        if (MODE == 2) { if (dre[0] == 123.456f && dim[3] == 1.f) yb[0] = make_float2(dre[1], dim[2]); continue; }
        float pr = dre[15], pi = dim[15];
#pragma unroll
        for (int r = 14; r >= 0; r--) { pr = shr1(pr) + dre[r]; pi = shr1(pi) + dim[r]; }
        const int t = T0 + lane;
        if (lane >= 15 && t < n_time) yb[t] = make_float2(pr, pi);
    }
}

template <int B, int FORM> static void run(int n_bins, int n_time)
{
    const int hist = B + 8;
    std::vector<float2> hx((size_t)n_bins * (n_time + hist)), hh((size_t)n_bins * B);
    srand(B);
    for (auto &v : hx) { v.x = rand() / (float)RAND_MAX - 0.5f; v.y = rand() / (float)RAND_MAX - 0.5f; }
    for (auto &v : hh) { v.x = (rand() / (float)RAND_MAX - 0.5f) / B; v.y = (rand() / (float)RAND_MAX - 0.5f) / B; }
    float2 *dx, *dh, *dy;
    CHECK(hipMalloc(&dx, hx.size() * 8)); CHECK(hipMalloc(&dh, hh.size() * 8)); CHECK(hipMalloc(&dy, (size_t)n_bins * n_time * 8));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dh, hh.data(), hh.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemset(dy, 0xff, (size_t)n_bins * n_time * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); for (int r = 0; r < 3; r++) launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
    };
    const double flop = 8.0 * B * (double)n_bins * n_time;
    const float t0 = time_it([&] { if (FORM == 4) hipLaunchKernelGGL((k_toeplitz<B, 0>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist); else hipLaunchKernelGGL((k_toeplitz16<B, 0>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist); });
    // correctness: a few bins, every time, against double
    std::vector<float2> hy((size_t)n_bins * n_time);
    CHECK(hipMemcpy(hy.data(), dy, hy.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0, peak = 0;
    for (int bi = 0; bi < 3; bi++) {
        const long bin = bi == 0 ? 0 : bi == 1 ? n_bins / 2 + 1 : n_bins - 1;
        for (int t = 0; t < n_time; t++) {
            double re = 0, im = 0;
            for (int p = 0; p < B; p++) {
                const float2 a = hx[bin * (n_time + hist) + hist + t - p], c = hh[bin * B + p];
                re += (double)a.x * c.x - (double)a.y * c.y; im += (double)a.x * c.y + (double)a.y * c.x;
            }
            const float2 g = hy[bin * n_time + t];
            worst = std::fmax(worst, std::fmax(std::fabs(g.x - re), std::fabs(g.y - im)));
            peak = std::fmax(peak, std::fmax(std::fabs(re), std::fabs(im)));
        }
    }
    const float t1 = time_it([&] { if (FORM == 4) hipLaunchKernelGGL((k_toeplitz<B, 1>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist); else hipLaunchKernelGGL((k_toeplitz16<B, 1>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist); });
    const float t2 = time_it([&] { if (FORM == 4) hipLaunchKernelGGL((k_toeplitz<B, 2>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist); else hipLaunchKernelGGL((k_toeplitz16<B, 2>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist); });
    printf("%s B = %3d, %d bins x %d outputs: MFMA pass %.3f ms = %.1f TFLOP/s (rel err %.2e of peak), MFMAs alone %.3f ms = %.1f TF, "
           "without reduction / stores %.3f ms = %.1f TF\n",
           FORM == 4 ? "4x4x1_16B " : "16x16x1_4B", B, n_bins, n_time, t0, flop / t0 * 1e-9, worst / peak, t1, flop / t1 * 1e-9, t2, flop / t2 * 1e-9);
    CHECK(hipFree(dx)); CHECK(hipFree(dh)); CHECK(hipFree(dy));
}

int main()
{
    // the headline's MAC: 8 channels x 4096 bins, 32 partitions; 1024 outputs here (a quarter launch)
    run<32, 4>(32768, 1020);
    run<64, 4>(16384, 1020);
    run<16, 4>(32768, 1020);
    run<32, 16>(32768, 4096);      // the headline's launch: 4096 blocks
    run<32, 16>(32768, 1020);
    run<64, 16>(16384, 1020);
    run<16, 16>(32768, 1020);
    return 0;
}
