// Tuning tool: does a read of one 64-byte half of a 128-byte line cost the fabric / HBM 64 or 128 bytes?
// A: every thread reads 16 B, contiguous (all bytes of `lines` lines);  B: the first 64 B of every line only;
// C: 64 B runs in a 16^3-tile checkerboard pattern of an N^3 grid (what one Gauss-Seidel colour pass reads of x).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void readAll(const float4 *__restrict__ a, float *__restrict__ out, size_t nq)
{
    size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;
    float4 v = t < nq ? a[t] : make_float4(0, 0, 0, 0);
    if (v.x == 123.f) out[0] = v.y;
}
__global__ __launch_bounds__(256) void readHalf(const float4 *__restrict__ a, float *__restrict__ out, size_t nlines)
{
    size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;  // 4 threads per line
    size_t line = t >> 2;
    float4 v = line < nlines ? a[line * 8 + (t & 3)] : make_float4(0, 0, 0, 0);
    if (v.x == 123.f) out[0] = v.y;
}
// one colour of a 16^3 checkerboard: thread -> (tile, row, quad)
__global__ __launch_bounds__(256) void readChecker(const float4 *__restrict__ a, float *__restrict__ out, int n)
{
    const int nt = n / 16;
    size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;
    const int q = t & 3;
    size_t r = t >> 2;              // row index among the rows of one colour
    const int halfTilesX = nt / 2;  // tiles of the colour per tile row
    const int tx2 = r % halfTilesX; r /= halfTilesX;
    const int j = r % n; r /= n;
    const int k = int(r);
    if (k >= n) return;
    const int ty = j / 16, tz = k / 16;
    const int tx = 2 * tx2 + ((ty + tz) & 1);
    float4 v = a[((size_t(k) * n + j) * n + tx * 16) / 4 + q];
    if (v.x == 123.f) out[0] = v.y;
}
static float timeK(void (*f)(void *), void *ctx)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f(ctx);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) f(ctx);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 20 * 1e3f;
}
struct Ctx { float4 *a; float *out; size_t bytes; int n; };
int main()
{
    Ctx c; c.n = 1024; c.bytes = size_t(c.n) * c.n * c.n * 4;  // 4 GiB: beyond the 256 MiB Infinity Cache
    CK(hipMalloc(&c.a, c.bytes)); CK(hipMalloc(&c.out, 64)); CK(hipMemset(c.a, 0, c.bytes));
    const size_t lines = c.bytes / 128;
    float tA = timeK([](void *p) { Ctx *c = (Ctx *)p; size_t nq = c->bytes / 16; readAll<<<unsigned((nq + 255) / 256), 256>>>(c->a, c->out, nq); }, &c);
    float tB = timeK([](void *p) { Ctx *c = (Ctx *)p; size_t nl = c->bytes / 128; readHalf<<<unsigned((nl * 4 + 255) / 256), 256>>>(c->a, c->out, nl); }, &c);
    float tC = timeK([](void *p) { Ctx *c = (Ctx *)p; size_t thr = size_t(c->n) * c->n * (c->n / 32) * 4; readChecker<<<unsigned((thr + 255) / 256), 256>>>(c->a, c->out, c->n); }, &c);
    printf("%zu lines of 128 B (%.2f GB)\n", lines, c.bytes / 1e9);
    printf("  A all bytes          %8.1f us  %6.2f TB/s of bytes read\n", tA, c.bytes / tA / 1e6);
    printf("  B first 64 B of each %8.1f us  %6.2f TB/s of bytes asked for, %6.2f TB/s if whole lines move\n", tB, c.bytes / 2 / tB / 1e6, c.bytes / tB / 1e6);
    printf("  C 16^3 checkerboard  %8.1f us  %6.2f TB/s of bytes asked for, %6.2f TB/s if whole lines move\n", tC, c.bytes / 2 / tC / 1e6, c.bytes / tC / 1e6);
    return 0;
}
