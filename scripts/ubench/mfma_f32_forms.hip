// mfma_f32_forms.hip -- the f32-input (and, rates only, the f64) MFMA forms of gfx950 that take ONE value per lane and operand: issue rate with
// independent accumulators, and the lane / register layout of the multi-block forms (probed, not assumed:
// A = lane + 1, B = 1 in ONE lane; every non-zero D entry names the A lane that met that B lane).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f32_forms.hip -o /tmp/mfma_f32_forms && /tmp/mfma_f32_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <int FORM, int NACC> __global__ __launch_bounds__(256) void k_rate(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    if constexpr (FORM == 0) {          // 4x4x1, 16 blocks
        f32x4 d[NACC];
        for (int c = 0; c < NACC; c++) d[c] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int c = 0; c < NACC; c++) d[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[c], 0, 0, 0);
        float s = 0; for (int c = 0; c < NACC; c++) s += d[c][0] + d[c][3];
        if (s == 1.2345f) out[0] = s;
    } else if constexpr (FORM == 1) {   // 16x16x1, 4 blocks
        f32x16 d[NACC];
        for (int c = 0; c < NACC; c++) for (int r = 0; r < 16; r++) d[c][r] = 0;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int c = 0; c < NACC; c++) d[c] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, d[c], 0, 0, 0);
        float s = 0; for (int c = 0; c < NACC; c++) s += d[c][0] + d[c][15];
        if (s == 1.2345f) out[0] = s;
    } else {                            // 16x16x4, one block
        f32x4 d[NACC];
        for (int c = 0; c < NACC; c++) d[c] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int c = 0; c < NACC; c++) d[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[c], 0, 0, 0);
        float s = 0; for (int c = 0; c < NACC; c++) s += d[c][0] + d[c][3];
        if (s == 1.2345f) out[0] = s;
    }
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
// the two f64 forms: 16x16x4 (one tile, four f64 results per lane) and 4x4x4 in 4 blocks (one result per lane)
template <int FORM, int NACC> __global__ __launch_bounds__(256) void k_rate64(float *out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    if constexpr (FORM == 0) {
        f64x4 d[NACC];
        for (int c = 0; c < NACC; c++) d[c] = f64x4{0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int c = 0; c < NACC; c++) d[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d[c], 0, 0, 0);
        double s = 0; for (int c = 0; c < NACC; c++) s += d[c][0] + d[c][3];
        if (s == 1.2345) out[0] = (float)s;
    } else {
        double d[NACC];
        for (int c = 0; c < NACC; c++) d[c] = 0;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int c = 0; c < NACC; c++) d[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[c], 0, 0, 0);
        double s = 0; for (int c = 0; c < NACC; c++) s += d[c];
        if (s == 1.2345) out[0] = (float)s;
    }
}

// v_mfma_f64_4x4x4_4B with operands that CHANGE from instruction to instruction (eight A and eight B register pairs in turn, as a
// Toeplitz pass uses them) and, optionally, each A operand fetched from LDS right before its use: what of the two halves the rate
// of scripts/ubench/mfma_f64_toeplitz.hip's loop?
template <int NACC, int LDS> __global__ __launch_bounds__(256) void k_rate64_ops(float *out, int iters)
{
    __shared__ double s_a[256 + 64];
    double a[8], b[8], d[NACC];
    for (int j = 0; j < 8; j++) { a[j] = threadIdx.x * 1e-3 + j; b[j] = 1.0 + threadIdx.x * 1e-4 - j; }
    for (int j = threadIdx.x; j < 256 + 64; j += 256) s_a[j] = j * 1e-2;
    __syncthreads();
    for (int c = 0; c < NACC; c++) d[c] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            double av = a[j];
            if (LDS) av = s_a[((threadIdx.x & 63) + 4 * j + (it & 63)) & 255];
#pragma unroll
            for (int c = 0; c < NACC; c++) d[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b[(j + c) & 7], d[c], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" : "+v"(a[j]), "+v"(b[j]));
    }
    double sum = 0; for (int c = 0; c < NACC; c++) sum += d[c];
    if (sum == 1.2345) out[0] = (float)sum;
}

template <int NACC, int LDS> static void rate64_ops(int wps)
{
    float *d; CHECK(hipMalloc(&d, 4));
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate64_ops<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, d, 50); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL((k_rate64_ops<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, d, iters); CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * 8 * NACC * wps;
    printf("v_mfma_f64_4x4x4_4B, changing operands%s, %d accumulators, %d wave(s) per SIMD: %.2f ns per MFMA per SIMD, %.1f TFLOP/s\n",
           LDS ? ", A from LDS" : "", NACC, wps, ms * 1e6 / n, 2.0 * 256 * n * 1024 / (ms * 1e-3) * 1e-12);
    CHECK(hipFree(d));
}

template <int FORM, int NACC> static void rate64(const char *name, double macs, int wps)
{
    float *d; CHECK(hipMalloc(&d, 4));
    const int iters = 20000, blocks = 256 * wps;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate64<FORM, NACC>), dim3(blocks), dim3(256), 0, 0, d, 100); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL((k_rate64<FORM, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters); CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * NACC * wps;
    printf("%-22s %d accumulators, %d wave(s) per SIMD: %.2f ns per MFMA per SIMD, %.1f TFLOP/s\n", name, NACC, wps, ms * 1e6 / n,
           2.0 * macs * n * 1024 / (ms * 1e-3) * 1e-12);
    CHECK(hipFree(d));
}

template <int FORM, int NACC> static void rate(const char *name, double macs, int wps)
{
    float *d; CHECK(hipMalloc(&d, 4));
    const int iters = 20000, blocks = 256 * wps;     // 256 CUs x wps workgroups of 4 waves = wps waves per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<FORM, NACC>), dim3(blocks), dim3(256), 0, 0, d, 100); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL((k_rate<FORM, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters); CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * NACC * wps;      // MFMAs per SIMD
    printf("%-22s %d accumulators, %d wave(s) per SIMD: %.2f ns per MFMA per SIMD, %.1f TFLOP/s\n", name, NACC, wps, ms * 1e6 / n,
           2.0 * macs * n * 1024 / (ms * 1e-3) * 1e-12);
    CHECK(hipFree(d));
}

__global__ void k_probe(float *out, int lb, int form)
{
    const int l = threadIdx.x;
    const float a = (float)(l + 1), b = l == lb ? 1.f : 0.f;
    if (form == 0) {
        f32x4 d = {0, 0, 0, 0};
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
        for (int r = 0; r < 4; r++) out[r * 64 + l] = d[r];
    } else {
        f32x16 d; for (int r = 0; r < 16; r++) d[r] = 0;
        d = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, d, 0, 0, 0);
        for (int r = 0; r < 16; r++) out[r * 64 + l] = d[r];
    }
}

static void probe(int form, int regs)
{
    float *d; CHECK(hipMalloc(&d, 16 * 64 * 4));
    std::vector<float> h(16 * 64);
    printf("layout of %s: B = 1 in lane lb only -> (register, lane) <- A lane\n", form == 0 ? "4x4x1 (16 blocks)" : "16x16x1 (4 blocks)");
    for (int lb : {0, 1, 5, 16, 17, 35}) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, lb, form);
        CHECK(hipMemcpy(h.data(), d, 16 * 64 * 4, hipMemcpyDeviceToHost));
        printf("  lb = %2d:", lb);
        int shown = 0;
        for (int r = 0; r < regs; r++)
            for (int l = 0; l < 64; l++)
                if (h[r * 64 + l] != 0.f && shown++ < 20) printf(" (r%d, l%d)<-a%d", r, l, (int)h[r * 64 + l] - 1);
        printf("\n");
    }
    CHECK(hipFree(d));
}

int main()
{
    for (int w : {1, 2, 4}) {
        rate<0, 4>("v_mfma_f32_4x4x1_16B", 256, w);
        rate<1, 2>("v_mfma_f32_16x16x1_4B", 1024, w);
        rate<2, 4>("v_mfma_f32_16x16x4", 1024, w);
    }
    rate<0, 1>("v_mfma_f32_4x4x1_16B", 256, 1);
    rate<0, 2>("v_mfma_f32_4x4x1_16B", 256, 1);
    rate<1, 1>("v_mfma_f32_16x16x1_4B", 1024, 1);
    for (int w : {1, 2, 4}) {
        rate64<0, 4>("v_mfma_f64_16x16x4", 1024, w);
        rate64<1, 4>("v_mfma_f64_4x4x4_4B", 256, w);
    }
    rate64<0, 1>("v_mfma_f64_16x16x4", 1024, 1);
    rate64<1, 1>("v_mfma_f64_4x4x4_4B", 256, 1);
    for (int w : {1, 2, 4}) { rate64_ops<4, 0>(w); rate64_ops<4, 1>(w); rate64_ops<2, 1>(w); }
    probe(0, 4);
    probe(1, 16);
    return 0;
}
