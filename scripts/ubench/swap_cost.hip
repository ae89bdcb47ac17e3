// swap_cost.hip -- what does a cross-half lane exchange cost next to FMA chains on gfx950?
// Per iteration a wave runs 64 FMAs (two interleaved dependent chains, operands from registers: the systolic MAC's
// slot) plus NS exchanges of one of four kinds:
//   0 v_permlane32_swap_b32     1 ds_bpermute_b32 (__shfl)     2 v_mov_b32 (plain copy, the yardstick)     3 v_mov_b32 DPP row_shr:1
// Five waves per SIMD.  Prints ns per iteration per SIMD and the cost of one exchange in FMA-issue equivalents.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/ubench/swap_cost.hip -o /tmp/swap_cost && /tmp/swap_cost
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND, int NS> __global__ __launch_bounds__(256) void k(float *out, int iters, float a0)
{
    float x[32], h[32], acc0 = 0.f, acc1 = 1.f, e[8];
#pragma unroll
    for (int j = 0; j < 32; j++) { x[j] = a0 + threadIdx.x * 1e-7f + j * 1e-3f; h[j] = 1.0f - j * 1e-4f; }
#pragma unroll
    for (int j = 0; j < 8; j++) e[j] = threadIdx.x + j;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            acc0 = __builtin_fmaf(x[j], h[j], acc0);
            acc1 = __builtin_fmaf(x[j], h[(j + 1) & 31], acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (KIND == 0) {
                typedef unsigned int u2 __attribute__((ext_vector_type(2)));
                const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(e[s]), __float_as_uint(e[(s + 1) % 8]), false, false);
                e[s] = __uint_as_float(r.x); e[(s + 1) % 8] = __uint_as_float(r.y);
            } else if (KIND == 1) {
                e[s] = __shfl(e[s], (int)(threadIdx.x ^ 32), 64);
            } else if (KIND == 2) {
                asm volatile("v_mov_b32 %0, %1" : "=v"(e[s]) : "v"(e[(s + 1) % 8]));
            } else {
                e[s] = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(e[s]), __float_as_uint(e[(s + 1) % 8]), 0x111, 0xf, 0xf, false));
            }
        }
        acc0 += e[0] * 1e-30f;                                    // the exchanged values feed the chains
#pragma unroll
        for (int j = 0; j < 32; j++) asm volatile("" : "+v"(x[j]));
    }
    float s = acc0 + acc1;
#pragma unroll
    for (int j = 0; j < 8; j++) s += e[j];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND, int NS> double run()
{
    float *d; (void)hipMalloc(&d, 4);
    const int iters = 4000, wpe = 5, blocks = 256 * wpe;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NS>), dim3(blocks), dim3(256), 0, 0, d, 50, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms * 1e6 / ((double)wpe * iters);                     // ns per iteration per SIMD
}

int main()
{
    const double base = run<2, 0>();
    printf("64 FMAs, no exchange:                 %7.1f ns per iteration per SIMD (%.2f ns per FMA)\n", base, base / 65);
    const char *names[4] = {"v_permlane32_swap_b32", "ds_bpermute_b32", "v_mov_b32", "v_mov_b32 dpp row_shr:1"};
    double t[4][3];
    t[0][0] = run<0, 2>(); t[0][1] = run<0, 4>(); t[0][2] = run<0, 8>();
    t[1][0] = run<1, 2>(); t[1][1] = run<1, 4>(); t[1][2] = run<1, 8>();
    t[2][0] = run<2, 2>(); t[2][1] = run<2, 4>(); t[2][2] = run<2, 8>();
    t[3][0] = run<3, 2>(); t[3][1] = run<3, 4>(); t[3][2] = run<3, 8>();
    for (int kd = 0; kd < 4; kd++)
        printf("%-26s + 2: %7.1f   + 4: %7.1f   + 8: %7.1f ns   => one exchange = %.1f FMA issue slots\n", names[kd], t[kd][0], t[kd][1],
               t[kd][2], (t[kd][2] - base) / 8 / (base / 65));
    return 0;
}
