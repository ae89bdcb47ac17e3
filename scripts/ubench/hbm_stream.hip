// hbm_stream.hip -- measured HBM stream rates on this GPU (quote the measured read-stream
// peak next to the 8 TB/s specification).  float4 per lane, grid-stride, buffer far larger than the
// 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_read(const float4 *__restrict__ p, size_t n, float *out)
{
    float4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ void k_write(float4 *__restrict__ p, size_t n)
{
    const float4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main()
{
    const size_t bytes = (size_t)4 << 30, n = bytes / sizeof(float4);
    float4 *a, *b; float *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256 * 8, 256 * 16, 256 * 32}) {
        float ms[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) {
            for (int rep = 0; rep < 4; rep++) {   // first repetition is warm-up
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, o);
                if (k == 1) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n);
                if (k == 2) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (rep > 0) ms[k] += t / 3;
            }
        }
        printf("%5d workgroups of 256: read %.0f GB/s   write %.0f GB/s   copy %.0f GB/s (read + written bytes)\n", blocks,
               bytes / ms[0] / 1e6, bytes / ms[1] / 1e6, 2.0 * bytes / ms[2] / 1e6);
    }
    return 0;
}
