// Tuning tool: what does the memory system charge for the x-face of a grid?  One 16-byte quad per x-row of an N^3 grid (x fastest):
// every access is its own 128-byte line, 4 N bytes from the next one.  A: one array;  B: two arrays (x and rhs);  C: the same
// rows with a read AND a 16-byte write into a third array (the band stage's pattern on an x-face: read x, read b, write out).
// D: a y-face for comparison (contiguous rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void faceRead(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ w, float *__restrict__ out, int n, int lo, int hi, int xq)
{
    const int span = hi - lo;
    const size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= size_t(span) * span) return;
    const int j = lo + int(t % span), k = lo + int(t / span);
    const size_t q = ((size_t(k) * n + j) * n) / 4 + xq;
    float4 v = a[q];
    if (b) { const float4 u = b[q]; v.x += u.x; v.y += u.y; }
    if (w) w[q] = v;
    else if (v.x == 123.f) out[0] = v.y;
}
__global__ __launch_bounds__(256) void yfaceRead(const float4 *__restrict__ a, const float4 *__restrict__ b, float *__restrict__ out, int n, int lo, int hi, int j)
{
    const int span = hi - lo, nq = span / 4;
    const size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= size_t(nq) * span * 3) return;
    const int i = int(t % nq), r = int((t / nq) % 3), k = lo + int(t / (size_t(nq) * 3));
    const size_t q = ((size_t(k) * n + j + r) * n + lo) / 4 + i;
    float4 v = a[q];
    if (b) { const float4 u = b[q]; v.x += u.x; }
    if (v.x == 123.f) out[0] = v.y;
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const int pad = n / 16, lo = pad, hi = n - pad;
    const size_t bytes = size_t(n) * n * n * 4;
    float4 *a, *b, *w; float *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&w, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(w, 0, bytes));
    const size_t rows = size_t(hi - lo) * (hi - lo);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int arrays, bool write, bool yface) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            // flush the caches between repetitions: stream over another array
            CK(hipMemsetAsync(w, 0, yface || !write ? bytes : 0));
            CK(hipEventRecord(e0));
            if (yface) yfaceRead<<<unsigned((size_t((hi - lo) / 4) * (hi - lo) * 3 + 255) / 256), 256>>>(a, arrays > 1 ? b : nullptr, out, n, lo, hi, lo);
            else {
                // both x-faces
                faceRead<<<unsigned((rows + 255) / 256), 256>>>(a, arrays > 1 ? b : nullptr, write ? w : nullptr, out, n, lo, hi, lo / 4);
                faceRead<<<unsigned((rows + 255) / 256), 256>>>(a, arrays > 1 ? b : nullptr, write ? w : nullptr, out, n, lo, hi, hi / 4 - 1);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double lines = yface ? double(hi - lo) * (hi - lo) * 3 * 4 / 128 * arrays : 2.0 * rows * (arrays + (write ? 1 : 0));
        printf("  %-44s %8.1f us  %7.1f M lines  %6.2f TB/s in whole lines\n", name, best * 1e3, lines / 1e6, lines * 128 / best / 1e9);
    };
    printf("N = %d, %zu rows per face\n", n, rows);
    run("A x-faces, one array", 1, false, false);
    run("B x-faces, two arrays", 2, false, false);
    run("C x-faces, two arrays read + one written", 2, true, false);
    run("D one y-face (3 rows deep), two arrays", 2, false, true);
    return 0;
}
