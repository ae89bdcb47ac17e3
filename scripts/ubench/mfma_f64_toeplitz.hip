// mfma_f64_toeplitz.hip -- the fp64 partition sums  y[t] = sum_{p < B} x[t - p] h[p]  (per bin; brutefir.cpp:288-299) on
// v_mfma_f64_4x4x4_4B_f64 (4 tiles of 4 x 4, K = 4, one f64 per lane and operand; 7.7 ns per instruction and SIMD = 68 TFLOP/s,
// scripts/ubench/mfma_f32_forms.hip).  A GO / NO-GO measurement for a later fp64 MAC kernel, not product code; the fp32 companion
// (mfma_toeplitz.hip) has the idea: the tiles of one instruction are consecutive TIME tiles of ONE bin.  Here with K = 4:
//     tile b, row i, k:   a = x[T0 + 4 b - P - 4 k + i]            (gathered from the bin's time series in LDS)
//     k, column j:        h[P + 4 k + j]                            (sixteen partitions per instruction, the same for every tile)
//     D_b[i][j] += sum_k a b                                        belongs to output  T0 + 4 b + i + j
// Outputs T0 + 3 .. T0 + 15 are complete after the B / 16 steps (13 per pass; the skewed ends are recomputed by the neighbouring
// passes), each the sum of the four entries 4 b + i + j = const: read back through LDS by the lane that stores the output.
// Complex: four real MFMAs per step.  Layout of the instruction (probed below, checked for all 64 lanes): A lane = 16 k + 4 tile + row,
// B lane = 16 k + 4 tile + column, D lane = 16 row + 4 tile + column.
// NOT covered: the transposes between the delay line's [block][bin] order and a bin's time series, DC / Nyquist, the ring.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f64_toeplitz.hip -o /tmp/mfma_f64_toeplitz && /tmp/mfma_f64_toeplitz
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

__global__ void k_probe(double *out, int lb)
{
    const int l = threadIdx.x;
    double d = 0;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64((double)(l + 1), l == lb ? 1.0 : 0.0, d, 0, 0, 0);
    out[l] = d;
}

// lane -> (tile, k, row / column) of the A and B operands and (tile, row, column) of D, filled from the probe
struct Layout { int a_b[64], a_k[64], a_i[64], b_b[64], b_k[64], b_j[64], d_b[64], d_i[64], d_j[64]; };

constexpr int STEP = 13;

template <int B, int MODE>
__global__ __launch_bounds__(256) void k_toeplitz64(const double2 *__restrict__ x, const double2 *__restrict__ h, double2 *__restrict__ y,
                                                    int n_time, int hist, Layout L)
{
    constexpr int RING = B + 32 <= 128 ? 128 : 256;     // window B - 4 + 16 samples, power of two
    __shared__ double s_re[4][RING], s_im[4][RING], s_dre[4][64], s_dim[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long bin = (long)blockIdx.x * 4 + w;
    const double2 *xb = x + bin * (long)(n_time + hist) + hist;
    double2 *yb = y + bin * (long)n_time;
    double hr[B / 16], hi[B / 16], nhi[B / 16];
#pragma unroll
    for (int s = 0; s < B / 16; s++) {
        const double2 v = h[bin * B + 16 * s + 4 * L.b_k[lane] + L.b_j[lane]];
        hr[s] = v.x; hi[s] = v.y; nhi[s] = -v.y;
    }
    double *sr = s_re[w], *si = s_im[w];
    for (int t = -RING + 64 + lane; t < 0; t += 64) {
        const double2 v = t >= -hist ? xb[t] : make_double2(0., 0.);
        sr[t & (RING - 1)] = v.x; si[t & (RING - 1)] = v.y;
    }
    const int a_off = 4 * L.a_b[lane] - 4 * L.a_k[lane] + L.a_i[lane];
    // where this lane's D entry goes: slot (output offset 4 b + i + j, column j)
    const int d_slot = (4 * L.d_b[lane] + L.d_i[lane] + L.d_j[lane]) * 4 + L.d_j[lane];
    double ar0 = 0, ai0 = 0;
    // the next pass's sixteen samples are fetched one pass ahead (registers), so a pass never waits for memory
    auto fetch = [&](int T0) { const int t = T0 + (lane & 15); return (t >= -hist && t < n_time) ? xb[t] : make_double2(0., 0.); };
    double2 nxt = fetch(-3);
    for (int T0 = -3; T0 < n_time; T0 += STEP) {
        if (lane < 16) { const int t = T0 + lane; sr[t & (RING - 1)] = nxt.x; si[t & (RING - 1)] = nxt.y; }
        nxt = fetch(T0 + STEP);
        // four independent accumulator chains per wave (the instruction's dependent latency is 21.7 ns against 7.7 ns of issue)
        double dre = 0, dim = 0, dre2 = 0, dim2 = 0;
        if (MODE == 1) { ar0 = sr[lane]; ai0 = si[lane]; }
#pragma unroll
        for (int s = 0; s < B / 16; s++) {
            double ar, ai;
            if (MODE == 1) { ar = ar0; ai = ai0; }
            else { const int a = (T0 - 16 * s + a_off) & (RING - 1); ar = sr[a]; ai = si[a]; }
            dre = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, hr[s], dre, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, hi[s], dim, 0, 0, 0);
            dre2 = __builtin_amdgcn_mfma_f64_4x4x4f64(ai, nhi[s], dre2, 0, 0, 0);
            dim2 = __builtin_amdgcn_mfma_f64_4x4x4f64(ai, hr[s], dim2, 0, 0, 0);
        }
        dre += dre2; dim += dim2;
        if (MODE == 2) { if (dre == 123.456 && dim == 1.) yb[0] = make_double2(dre, dim); continue; }
        // entry (b, i, j) -> slot [4 b + i + j][j]: the four entries of an output side by side
        // (offsets 16 .. 18 are incomplete and dropped; the slots of offsets 0 .. 2 are never read)
        if (4 * L.d_b[lane] + L.d_i[lane] + L.d_j[lane] < 16) { s_dre[w][d_slot] = dre; s_dim[w][d_slot] = dim; }
        if (lane >= 3 && lane < 16) {
            const int t = T0 + lane;
            const double pr = (s_dre[w][4 * lane] + s_dre[w][4 * lane + 1]) + (s_dre[w][4 * lane + 2] + s_dre[w][4 * lane + 3]);
            const double pi = (s_dim[w][4 * lane] + s_dim[w][4 * lane + 1]) + (s_dim[w][4 * lane + 2] + s_dim[w][4 * lane + 3]);
            if (t < n_time) yb[t] = make_double2(pr, pi);
        }
    }
}

template <int B> static void run(int n_bins, int n_time, const Layout &L)
{
    const int hist = B + 8;
    std::vector<double2> hx((size_t)n_bins * (n_time + hist)), hh((size_t)n_bins * B);
    srand(B);
    for (auto &v : hx) { v.x = rand() / (double)RAND_MAX - 0.5; v.y = rand() / (double)RAND_MAX - 0.5; }
    for (auto &v : hh) { v.x = (rand() / (double)RAND_MAX - 0.5) / B; v.y = (rand() / (double)RAND_MAX - 0.5) / B; }
    double2 *dx, *dh, *dy;
    CHECK(hipMalloc(&dx, hx.size() * 16)); CHECK(hipMalloc(&dh, hh.size() * 16)); CHECK(hipMalloc(&dy, (size_t)n_bins * n_time * 16));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dh, hh.data(), hh.size() * 16, hipMemcpyHostToDevice));
    CHECK(hipMemset(dy, 0xff, (size_t)n_bins * n_time * 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); for (int r = 0; r < 3; r++) launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
    };
    const double flop = 8.0 * B * (double)n_bins * n_time;
    const float t0 = time_it([&] { hipLaunchKernelGGL((k_toeplitz64<B, 0>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist, L); });
    std::vector<double2> hy((size_t)n_bins * n_time);
    CHECK(hipMemcpy(hy.data(), dy, hy.size() * 16, hipMemcpyDeviceToHost));
    double worst = 0, peak = 0;
    for (int bi = 0; bi < 3; bi++) {
        const long bin = bi == 0 ? 0 : bi == 1 ? n_bins / 2 + 1 : n_bins - 1;
        for (int t = 0; t < n_time; t++) {
            long double re = 0, im = 0;
            for (int p = 0; p < B; p++) {
                const double2 a = hx[bin * (n_time + hist) + hist + t - p], c = hh[bin * B + p];
                re += (long double)a.x * c.x - (long double)a.y * c.y; im += (long double)a.x * c.y + (long double)a.y * c.x;
            }
            const double2 g = hy[bin * n_time + t];
            worst = std::fmax(worst, std::fmax(std::fabs((double)(g.x - re)), std::fabs((double)(g.y - im))));
            peak = std::fmax(peak, std::fmax(std::fabs((double)re), std::fabs((double)im)));
        }
    }
    const float t1 = time_it([&] { hipLaunchKernelGGL((k_toeplitz64<B, 1>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist, L); });
    const float t2 = time_it([&] { hipLaunchKernelGGL((k_toeplitz64<B, 2>), dim3(n_bins / 4), dim3(256), 0, 0, dx, dh, dy, n_time, hist, L); });
    printf("f64 4x4x4_4B B = %3d, %d bins x %d outputs: pass %.3f ms = %.1f TFLOP/s (rel err %.2e of peak), operands loaded once %.3f ms = %.1f TF, "
           "without reduction / stores %.3f ms = %.1f TF\n", B, n_bins, n_time, t0, flop / t0 * 1e-9, worst / peak, t1, flop / t1 * 1e-9, t2,
           flop / t2 * 1e-9);
    CHECK(hipFree(dx)); CHECK(hipFree(dh)); CHECK(hipFree(dy));
}

int main()
{
    // probe: A = lane + 1, B = 1 in lane lb -> the D lanes that light up are (tile of lb, rows 0..3, column of lb), their
    // values the A lanes (tile, k of lb, row)
    double *d; CHECK(hipMalloc(&d, 64 * 8));
    Layout L;
    int hit[64][4], from[64][4];
    for (int lb = 0; lb < 64; lb++) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, lb);
        double h[64]; CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        int n = 0;
        for (int l = 0; l < 64; l++) if (h[l] != 0.0 && n < 4) { hit[lb][n] = l; from[lb][n] = (int)h[l] - 1; n++; }
        if (n != 4) { printf("probe: B lane %d lights %d D lanes, expected 4\n", lb, n); return 1; }
    }
    printf("probe of v_mfma_f64_4x4x4_4B: B lane -> D lanes <- A lanes\n");
    for (int lb : {0, 1, 4, 5, 16, 21, 63})
        printf("  B %2d -> D %2d %2d %2d %2d <- A %2d %2d %2d %2d\n", lb, hit[lb][0], hit[lb][1], hit[lb][2], hit[lb][3], from[lb][0], from[lb][1],
               from[lb][2], from[lb][3]);
    // what the probe shows: A lane = 16 k + 4 tile + row, B lane = 16 k + 4 tile + column, D lane = 16 row + 4 tile + column
    for (int l = 0; l < 64; l++) {
        L.a_k[l] = l >> 4; L.a_b[l] = (l >> 2) & 3; L.a_i[l] = l & 3;
        L.b_k[l] = l >> 4; L.b_b[l] = (l >> 2) & 3; L.b_j[l] = l & 3;
        L.d_i[l] = l >> 4; L.d_b[l] = (l >> 2) & 3; L.d_j[l] = l & 3;
    }
    for (int lb = 0; lb < 64; lb++)         // ... checked against every probed B lane
        for (int r = 0; r < 4; r++)
            if (hit[lb][r] != 16 * r + 4 * L.b_b[lb] + L.b_j[lb] || from[lb][r] != 16 * L.b_k[lb] + 4 * L.b_b[lb] + r) {
                printf("probe: layout differs from the formula at B lane %d\n", lb); return 1;
            }
    printf("  lane: A (tile,k,row)  B (tile,k,col)  D (tile,row,col)\n");
    for (int l : {0, 1, 4, 5, 16, 21, 63})
        printf("  %2d:   (%d,%d,%d)  (%d,%d,%d)  (%d,%d,%d)\n", l, L.a_b[l], L.a_k[l], L.a_i[l], L.b_b[l], L.b_k[l], L.b_j[l], L.d_b[l], L.d_i[l], L.d_j[l]);
    CHECK(hipFree(d));
    run<64>(2048, 32760, L);       // the plug-in's shape: stereo x 1024 bins, 64 partitions, one 32768-block launch
    run<64>(16384, 4095, L);
    run<32>(16384, 4095, L);
    run<128>(2048, 32760, L);
    return 0;
}
