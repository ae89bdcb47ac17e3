// hbm_mix.hip -- what mixed read / write streams reach on this GPU, and whether the cache policy of the accesses changes it.
// Variants of a float4 copy over 4 GiB + 4 GiB: plain, nontemporal loads, nontemporal stores, both; a 2:1 read:write mix
// (the pipeline's ratio is 1:1.1); unrolled x4 (more bytes in flight per lane).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/hbm_mix.hip -o /tmp/hbm_mix && /tmp/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS, int U>
__global__ void k_copy(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NTL ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NTS) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
    }
}
// two reads per write
template <bool NT>
__global__ void k_add(const f4 *__restrict__ a, const f4 *__restrict__ c, f4 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const f4 x = NT ? __builtin_nontemporal_load(a + i) : a[i], y = NT ? __builtin_nontemporal_load(c + i) : c[i];
        const f4 s = x + y;
        if (NT) __builtin_nontemporal_store(s, b + i); else b[i] = s;
    }
}

int main()
{
    const size_t bytes = (size_t)4 << 30, n = bytes / sizeof(f4);
    f4 *a, *b, *c;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess) return 1;
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 16;
    auto time = [&](auto launch, const char *name, double factor) {
        float ms = 0;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); if (rep > 0) ms += t / 3;
        }
        printf("%-44s %.0f GB/s (read + written bytes)\n", name, factor * bytes / ms / 1e6);
    };
    time([&] { hipLaunchKernelGGL((k_copy<false, false, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, "copy, plain", 2.0);
    time([&] { hipLaunchKernelGGL((k_copy<true, false, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, "copy, nt loads", 2.0);
    time([&] { hipLaunchKernelGGL((k_copy<false, true, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, "copy, nt stores", 2.0);
    time([&] { hipLaunchKernelGGL((k_copy<true, true, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, "copy, nt loads + stores", 2.0);
    time([&] { hipLaunchKernelGGL((k_copy<false, false, 4>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, "copy, plain, 4 x 16 B in flight per lane", 2.0);
    time([&] { hipLaunchKernelGGL((k_copy<true, true, 4>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, "copy, nt, 4 x 16 B in flight per lane", 2.0);
    time([&] { hipLaunchKernelGGL((k_add<false>), dim3(blocks), dim3(256), 0, 0, a, c, b, n); }, "b = a + c (2 reads : 1 write), plain", 3.0);
    time([&] { hipLaunchKernelGGL((k_add<true>), dim3(blocks), dim3(256), 0, 0, a, c, b, n); }, "b = a + c (2 reads : 1 write), nt", 3.0);
    return 0;
}
