// valu_rate.hip -- how many cycles does one wave64 VALU op cost on gfx950?
// Runs ITER x 32 independent ops per thread with W waves per SIMD (block = 256 threads,
// blocks per CU = W); every operand is a VGPR (thread-dependent values), so the loop body is
// nothing but the instruction under test.  Prints FLOP/s and ns per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_add_f32, 3 v_pk_mul_f32, 4 v_add_f32, 5 v_fma_f64, 6 / 7 the MAC's operand pattern (f64 / f32)
template <int MODE> __global__ void k(float *out, int iters, float a0, float b0)
{
    float acc[32];
    const float a = a0 + threadIdx.x * 1e-9f, b = b0 + threadIdx.x * 1e-9f;
    const v2f a2 = {a, a + 1e-9f}, b2 = {b, b - 1e-9f};
#pragma unroll
    for (int j = 0; j < 32; j++) acc[j] = threadIdx.x * 0.001f + j;
    if (MODE == 6 || MODE == 7) {
        // the MAC's operand pattern: acc[j] = fma(x[j], h, acc[j]) -- one operand repeated, two fresh register pairs
        // per instruction (MODE 6: f64, MODE 7: f32 with 32 chains)
        if (MODE == 6) {
            double d[16], e[16];
            const double da = a;
#pragma unroll
            for (int j = 0; j < 16; j++) { d[j] = acc[j]; e[j] = acc[j + 16] * 1e-3; }
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int j = 0; j < 16; j++) d[j] = __builtin_fma(e[j], da, d[j]);
#pragma unroll
                for (int j = 0; j < 16; j++) asm volatile("" : "+v"(e[j]));
            }
#pragma unroll
            for (int j = 0; j < 16; j++) acc[j] = (float)d[j];
        } else {
            float e[32];
#pragma unroll
            for (int j = 0; j < 32; j++) e[j] = acc[j] * 1e-3f;
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int j = 0; j < 32; j++) acc[j] = __builtin_fmaf(e[j], a, acc[j]);
#pragma unroll
                for (int j = 0; j < 32; j++) asm volatile("" : "+v"(e[j]));
            }
        }
    } else if (MODE == 5) {
        double d[16];
        const double da = a, db = b;
#pragma unroll
        for (int j = 0; j < 16; j++) d[j] = acc[j];
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 16; j++) d[j] = __builtin_fma(d[j], da, db);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) acc[j] = (float)d[j];
    } else
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 32; j++) acc[j] = __builtin_fmaf(acc[j], a, b);
        } else if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 32; j++) acc[j] = acc[j] + a;
        } else {
#pragma unroll
            for (int j = 0; j < 32; j += 2) {
                v2f v = {acc[j], acc[j + 1]};
                if (MODE == 1) v = __builtin_elementwise_fma(v, a2, b2);
                if (MODE == 2) v = v + a2;
                if (MODE == 3) v = v * a2;
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
    const int instr = (MODE == 0 || MODE == 4 || MODE == 7) ? 32 : 16;
    const double lane_ops = (double)blocks * 256 * iters * ((MODE == 5 || MODE == 6) ? 16 : 32);
    const double wave_instr_per_simd = (double)wpe * iters * instr;   // 1 wave per SIMD per block
    printf("%-12s waves/SIMD %d: %.3f ms  %.1f Tops/s (x2 = FLOP/s for fma)  %.2f ns per wave-instr per SIMD\n",
           name, wpe, ms, lane_ops / ms / 1e9, ms * 1e6 / wave_instr_per_simd);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) run<0>("v_fma_f32", w);
    for (int w : {1, 2, 4}) run<4>("v_add_f32", w);
    for (int w : {1, 2, 4}) run<1>("v_pk_fma_f32", w);
    for (int w : {1, 2, 4}) run<2>("v_pk_add_f32", w);
    for (int w : {1, 2, 4}) run<3>("v_pk_mul_f32", w);
    for (int w : {1, 2, 4}) run<5>("v_fma_f64", w);
    for (int w : {1, 2, 3, 4}) run<6>("v_fma_f64 acc+=x*h", w);
    for (int w : {1, 2, 3, 4}) run<7>("v_fma_f32 acc+=x*h", w);
    return 0;
}
