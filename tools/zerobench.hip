// zero-fill variants: GB/s for 512 MiB and 64 MiB buffers.  hipcc --offload-arch=gfx950 -O3 tools/zerobench.hip -o tools/zerobench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void strideK(float4 *v, size_t nq)
{
    for (size_t q = size_t(blockIdx.x) * blockDim.x + threadIdx.x; q < nq; q += size_t(gridDim.x) * blockDim.x) v[q] = make_float4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void strideNtK(v4f *v, size_t nq)
{
    for (size_t q = size_t(blockIdx.x) * blockDim.x + threadIdx.x; q < nq; q += size_t(gridDim.x) * blockDim.x) __builtin_nontemporal_store(v4f{0, 0, 0, 0}, v + q);
}
__global__ __launch_bounds__(256) void blockK(float4 *v, size_t nq)  // one 4 KB piece per block, no loop
{
    const size_t q = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q < nq) v[q] = make_float4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void block4K(float4 *v, size_t nq)  // 16 KB contiguous per block
{
    const size_t base = size_t(blockIdx.x) * 1024 + threadIdx.x;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (base + m * 256 < nq) v[base + m * 256] = make_float4(0, 0, 0, 0);
}
int main()
{
    for (size_t mb : {512, 64}) {
        const size_t bytes = mb << 20, nq = bytes / 16;
        float4 *d;
        hipMalloc(&d, bytes);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        auto time = [&](const char *name, auto launch) {
            for (int w = 0; w < 3; ++w) launch();
            hipEventRecord(e0);
            for (int r = 0; r < 20; ++r) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%4zu MiB %-28s %7.1f us  %6.0f GB/s\n", mb, name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e9);
        };
        time("grid-stride 2048 blocks", [&] { strideK<<<2048, 256>>>(d, nq); });
        time("grid-stride 8192 blocks", [&] { strideK<<<8192, 256>>>(d, nq); });
        time("grid-stride nt 2048", [&] { strideNtK<<<2048, 256>>>((v4f *)d, nq); });
        time("one quad per thread", [&] { blockK<<<unsigned((nq + 255) / 256), 256>>>(d, nq); });
        time("4 quads per thread, contiguous", [&] { block4K<<<unsigned((nq + 1023) / 1024), 256>>>(d, nq); });
        time("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
        hipFree(d);
    }
    return 0;
}
