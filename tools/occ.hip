// occupancy probe: how many workgroups with S bytes of static LDS does a CU take?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *o)
{
    __shared__ float s[BYTES / 4];
    s[threadIdx.x] = o[threadIdx.x];
    __syncthreads();
    // spin a while so that residency can be observed
    float a = s[(threadIdx.x * 7) % (BYTES / 4)];
    for (int i = 0; i < 20000; ++i) a = a * 1.0001f + 0.5f;
    o[blockIdx.x * THREADS + threadIdx.x] = a;
}
template <int BYTES, int THREADS>
void probe(float *d)
{
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES, THREADS>, THREADS, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms1 = 0, ms8 = 0;
    k<BYTES, THREADS><<<256, THREADS>>>(d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<BYTES, THREADS><<<256, THREADS>>>(d);  // 1 block per CU
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms1, e0, e1);
    hipEventRecord(e0);
    k<BYTES, THREADS><<<256 * 8, THREADS>>>(d);  // 8 blocks per CU: time ratio = 8 / resident blocks
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms8, e0, e1);
    printf("LDS %6d B, %4d threads: API says %d blocks/CU; 1/CU %.3f ms, 8/CU %.3f ms -> ~%.1f resident\n", BYTES, THREADS, n, ms1, ms8,
           8.0 * ms1 / ms8);
}
int main()
{
    float *d;
    hipMalloc(&d, 256 * 8 * 1024 * 4);
    hipMemset(d, 0, 256 * 8 * 1024 * 4);
    probe<40304, 64>(d);
    probe<40304, 256>(d);
    probe<32768, 64>(d);
    probe<24576, 64>(d);
    probe<16384, 64>(d);
    probe<75856, 64>(d);
    probe<38912, 1024>(d);
    return 0;
}
