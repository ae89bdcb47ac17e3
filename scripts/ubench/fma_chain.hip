// fma_chain.hip -- how many independent fused multiply-add chains does a SIMD of gfx950 need to issue v_fma_f32 at its full
// rate?  Each wave runs NC independent dependent chains acc_c = fma(x_j, h_j, acc_c) (operands from registers, 32
// different x / h pairs per pass: the MAC's operand pattern), W waves per SIMD.  Prints ns per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/ubench/fma_chain.hip -o /tmp/fma_chain && /tmp/fma_chain
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NC> __global__ __launch_bounds__(256) void k(float *out, int iters, float a0)
{
    float x[32], h[32], acc[NC];
#pragma unroll
    for (int j = 0; j < 32; j++) { x[j] = a0 + threadIdx.x * 1e-7f + j * 1e-3f; h[j] = 1.0f - j * 1e-4f; }
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = c;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 32; j += NC > 32 ? 1 : 1) {
#pragma unroll
            for (int c = 0; c < NC; c++) acc[c] = __builtin_fmaf(x[(j + c) & 31], h[(j + 2 * c) & 31], acc[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 32; j++) asm volatile("" : "+v"(x[j]));
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) s += acc[c];
    if (s == 12345.678f) out[0] = s;
}

// the same in double: NC dependent v_fma_f64 chains per wave (16 x / h pairs: 64 registers)
template <int NC> __global__ __launch_bounds__(256) void kd(float *out, int iters, double a0)
{
    double x[16], h[16], acc[NC];
#pragma unroll
    for (int j = 0; j < 16; j++) { x[j] = a0 + threadIdx.x * 1e-9 + j * 1e-3; h[j] = 1.0 - j * 1e-4; }
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = c;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
#pragma unroll
            for (int c = 0; c < NC; c++) acc[c] = __builtin_fma(x[(j + c) & 15], h[(j + 2 * c) & 15], acc[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("" : "+v"(x[j]));
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) s += acc[c];
    if (s == 12345.678) out[0] = (float)s;
}

template <int NC> void rund(int wpe)
{
    float *d; (void)hipMalloc(&d, 4);
    const int iters = 4000, blocks = 256 * wpe;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kd<NC>, dim3(blocks), dim3(256), 0, 0, d, 50, 1.0001);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kd<NC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)wpe * iters * 16 * NC;
    printf("f64 chains/wave %2d  waves/SIMD %d  chains/SIMD %3d : %.2f ns per wave-DFMA per SIMD   (%.1f T lane-DFMA/s)\n", NC, wpe, NC * wpe,
           ms * 1e6 / wave_instr_per_simd, (double)blocks * 256 * iters * 16 * NC / ms / 1e9);
    (void)hipFree(d);
}

template <int NC> void run(int wpe)
{
    float *d; (void)hipMalloc(&d, 4);
    const int iters = 4000, blocks = 256 * wpe;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(256), 0, 0, d, 50, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)wpe * iters * 32 * NC;
    printf("chains/wave %2d  waves/SIMD %d  chains/SIMD %3d : %.2f ns per wave-FMA per SIMD   (%.1f T lane-FMA/s)\n", NC, wpe, NC * wpe,
           ms * 1e6 / wave_instr_per_simd, (double)blocks * 256 * iters * 32 * NC / ms / 1e9);
    (void)hipFree(d);
}

int main()
{
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run<1>(w);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run<2>(w);
    for (int w : {1, 2, 3, 4, 5, 6}) run<4>(w);
    for (int w : {1, 2, 3, 4}) run<8>(w);
    for (int w : {1, 2, 3}) run<16>(w);
    for (int w : {1, 2, 3, 4}) rund<1>(w);
    for (int w : {1, 2, 3, 4}) rund<2>(w);
    for (int w : {1, 2, 3, 4}) rund<4>(w);
    for (int w : {1, 2, 3}) rund<8>(w);
    return 0;
}
