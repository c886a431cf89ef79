// Tuning tool: what does a workgroup COST on this chip before it has done anything?  The band stage launches one workgroup of
// 1024 threads with 64 KB of LDS per box (18 808 of them on the 1024^3 cube); this prices the skeleton of such a launch:
//   A  empty workgroups (one LDS store, one barrier) of T threads with L bytes of static LDS
//   B  A + a chain of dependent loads (info -> list -> value), one hop each, from a cold array
//   C  B in persistent form: 2 workgroups per CU walk the groups, no prefetch
//   D  C with the next group's info and list hop requested a group ahead
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int THREADS, int LDSB>
__global__ __launch_bounds__(THREADS) void emptyK(float *out)
{
    __shared__ float s[LDSB / 4];
    s[threadIdx.x] = float(blockIdx.x);
    __syncthreads();
    if (s[(threadIdx.x + 1) % THREADS] == -1.f) out[0] = 1.f;
}
// info[g] = offset of the group's list; list[off + t] = index into val; val[...] = float
template <int THREADS, int LDSB>
__global__ __launch_bounds__(THREADS) void chainK(const int *__restrict__ info, const int *__restrict__ list, const float *__restrict__ val, float *out, int hops)
{
    __shared__ float s[LDSB / 4];
    const int off = __builtin_amdgcn_readfirstlane(info[blockIdx.x * 16]);
    float v = 0.f;
    if (hops >= 2) {
        const int e = list[off + threadIdx.x];
        v = hops >= 3 ? val[e] : float(e);
    }
    s[threadIdx.x] = v;
    __syncthreads();
    if (s[(threadIdx.x + 1) % THREADS] == -1.f) out[0] = 1.f;
}
template <int THREADS, int LDSB, bool PREFETCH>
__global__ __launch_bounds__(THREADS) void persistK(const int *__restrict__ info, const int *__restrict__ list, const float *__restrict__ val, float *out, int ngroups)
{
    __shared__ float s[LDSB / 4];
    int g = blockIdx.x;
    int off = g < ngroups ? __builtin_amdgcn_readfirstlane(info[g * 16]) : 0;
    int e = g < ngroups ? list[off + threadIdx.x] : 0;
    for (; g < ngroups; g += gridDim.x) {
        const int gn = g + gridDim.x;
        int offn = 0, en = 0;
        if (PREFETCH && gn < ngroups) {
            offn = __builtin_amdgcn_readfirstlane(info[gn * 16]);
            en = list[offn + threadIdx.x];
        }
        const float v = val[e];
        s[threadIdx.x] = v;
        __syncthreads();
        if (s[(threadIdx.x + 1) % THREADS] == -1.f) out[0] = 1.f;
        __syncthreads();
        if (!PREFETCH && gn < ngroups) {
            offn = __builtin_amdgcn_readfirstlane(info[gn * 16]);
            en = list[offn + threadIdx.x];
        }
        off = offn;
        e = en;
    }
}
int main(int argc, char **argv)
{
    const int ng = argc > 1 ? atoi(argv[1]) : 18808;
    constexpr int T = 1024;
    int *info, *list;
    float *val, *out, *flush;
    const size_t nval = size_t(1) << 28;  // 1 GiB of floats: the value hop misses every cache
    CK(hipMalloc(&info, size_t(ng) * 16 * 4)); CK(hipMalloc(&list, size_t(ng) * T * 4)); CK(hipMalloc(&val, nval * 4)); CK(hipMalloc(&out, 64));
    CK(hipMalloc(&flush, size_t(1) << 30));
    {
        int *hi = (int *)malloc(size_t(ng) * 16 * 4), *hl = (int *)malloc(size_t(ng) * T * 4);
        for (int g = 0; g < ng; ++g) {
            hi[g * 16] = g * T;
            // a group's values: 128 runs of 8 consecutive floats, 4 KB apart (an x-face), groups far apart
            for (int t = 0; t < T; ++t) hl[size_t(g) * T + t] = int((size_t(g) * 131072 + size_t(t / 8) * 1024 + t % 8) % nval);
        }
        CK(hipMemcpy(info, hi, size_t(ng) * 16 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(list, hl, size_t(ng) * T * 4, hipMemcpyHostToDevice));
        CK(hipMemset(val, 0, nval * 4));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeIt = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(flush, rep, size_t(1) << 30));
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("  %-64s %8.1f us  %6.2f ns per group\n", name, best * 1e3, best * 1e6 / ng);
    };
    printf("%d groups\n", ng);
    timeIt("A empty, 1024 threads, 64 KB LDS", [&] { emptyK<1024, 65536><<<ng, 1024>>>(out); });
    timeIt("A empty, 1024 threads, 16 KB LDS", [&] { emptyK<1024, 16384><<<ng, 1024>>>(out); });
    timeIt("A empty, 512 threads, 32 KB LDS", [&] { emptyK<512, 32768><<<ng, 512>>>(out); });
    timeIt("A empty, 256 threads, 16 KB LDS", [&] { emptyK<256, 16384><<<ng, 256>>>(out); });
    timeIt("A empty, 256 threads, 16 KB LDS, 4 x the groups", [&] { emptyK<256, 16384><<<4 * ng, 256>>>(out); });
    timeIt("B 1 hop (info)", [&] { chainK<1024, 65536><<<ng, 1024>>>(info, list, val, out, 1); });
    timeIt("B 2 hops (info -> list)", [&] { chainK<1024, 65536><<<ng, 1024>>>(info, list, val, out, 2); });
    timeIt("B 3 hops (info -> list -> values, 8-float runs 4 KB apart)", [&] { chainK<1024, 65536><<<ng, 1024>>>(info, list, val, out, 3); });
    timeIt("C persistent 512 workgroups, 3 hops, no prefetch", [&] { persistK<1024, 65536, false><<<512, 1024>>>(info, list, val, out, ng); });
    timeIt("D persistent 512 workgroups, info + list a group ahead", [&] { persistK<1024, 65536, true><<<512, 1024>>>(info, list, val, out, ng); });
    return 0;
}
