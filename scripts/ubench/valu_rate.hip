// valu_rate.hip -- how many cycles does one wave64 VALU FMA cost on gfx950?
// Runs ITER x 32 independent FMAs per thread with W waves per SIMD (block = 256 threads,
// blocks per CU = W) and prints FLOP/s and cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void k(float *out, int iters, float a, float b)
{
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; j++) acc[j] = threadIdx.x * 0.001f + j;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 32; j++) acc[j] = __builtin_fmaf(acc[j], a, b);
        } else {
#pragma unroll
            for (int j = 0; j < 32; j += 2) {
                v2f v = {acc[j], acc[j + 1]};
                v = __builtin_elementwise_fma(v, (v2f){a, a}, (v2f){b, b});
                acc[j] = v.x; acc[j + 1] = v.y;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) s += acc[j];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE> void run(const char *name, int wpe)
{
    float *d; hipMalloc(&d, 4);
    const int iters = 20000, blocks = 256 * wpe;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma_lane = (double)blocks * 256 * iters * 32;
    const double wave_instr_per_simd = (double)wpe * iters * (MODE == 0 ? 32 : 16);   // 1 wave per SIMD per block
    printf("%-12s waves/SIMD %d: %.3f ms  %.1f TFLOP/s  %.2f ns per wave-instr per SIMD (x clock GHz = cycles)\n",
           name, wpe, ms, 2 * fma_lane / ms / 1e9, ms * 1e6 / wave_instr_per_simd);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32", w);
    for (int w : {1, 2, 4, 8}) run<1>("v_pk_fma_f32", w);
    return 0;
}
