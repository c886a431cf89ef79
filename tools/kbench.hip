// Kernel micro-benchmark used while tuning (not part of the product): times variants of the fine
// level Jacobi sweep against two streaming upper bounds on an N^3 interior cube.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench.hip -o tools/kbench && tools/kbench 512
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct G {
    int nx, ny, nz;
    const uint8_t *lab;
};

__device__ __forceinline__ unsigned remap(unsigned bid, unsigned nb)
{
    const unsigned per = nb / 8;
    if (per == 0 || bid >= per * 8) return bid;
    return (bid % 8) * per + bid / 8;
}

// ---- upper bounds ----
__global__ __launch_bounds__(256) void copyK(float4 *__restrict__ o, const float4 *__restrict__ a, size_t nq)
{
    size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (t < nq) o[t] = a[t];
}
// in place: a[t] += c (read and write the same lines: what the prolongation does to the fine iterate)
__global__ __launch_bounds__(256) void inplaceK(float4 *a, size_t nq)
{
    size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= nq) return;
    float4 v = a[t];
    a[t] = make_float4(v.x + 1.f, v.y + 1.f, v.z + 1.f, v.w + 1.f);
}
__global__ __launch_bounds__(256) void inplaceNtK(float4 *a, size_t nq)
{
    size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= nq) return;
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(a) + t);
    v += 1.f;
    __builtin_nontemporal_store(v, reinterpret_cast<v4 *>(a) + t);
}
__global__ __launch_bounds__(256) void stream3K(float4 *__restrict__ o, const float4 *__restrict__ x,
                                                const float4 *__restrict__ b, const uchar4 *__restrict__ l, size_t nq)
{
    size_t t = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= nq) return;
    float4 xv = x[t], bv = b[t];
    uchar4 lv = l[t];
    float s = (lv.x | lv.y | lv.z | lv.w) ? 0.5f : 1.f;
    o[t] = make_float4(xv.x + s * bv.x, xv.y + s * bv.y, xv.z + s * bv.z, xv.w + s * bv.w);
}

// ---- variant A/B: one quad per thread, all neighbours from cache ----
template <bool REMAP>
__global__ __launch_bounds__(256) void jacA(G g, float *__restrict__ out, const float *__restrict__ x,
                                            const float *__restrict__ b, float omega, unsigned nblocks)
{
    const unsigned nq = unsigned(g.nx) >> 2;
    const size_t rows = size_t(g.ny) * g.nz;
    const size_t total = size_t(nq) * rows;
    const unsigned block = REMAP ? remap(blockIdx.x, nblocks) : blockIdx.x;
    const size_t t = size_t(block) * 256 + threadIdx.x;
    if (t >= total) return;
    const unsigned q = unsigned(t % nq);
    const size_t row = t / nq;
    const int j = int(row % g.ny), k = int(row / g.ny);
    const int i = int(q) << 2;
    const size_t sy = g.nx, sz = size_t(g.nx) * g.ny;
    const size_t c = row * sy + i;
    const float4 xc = *reinterpret_cast<const float4 *>(x + c);
    const float4 ym = *reinterpret_cast<const float4 *>(x + (j > 0 ? c - sy : c));
    const float4 yp = *reinterpret_cast<const float4 *>(x + (j < g.ny - 1 ? c + sy : c));
    const float4 zm = *reinterpret_cast<const float4 *>(x + (k > 0 ? c - sz : c));
    const float4 zp = *reinterpret_cast<const float4 *>(x + (k < g.nz - 1 ? c + sz : c));
    const uchar4 lab = *reinterpret_cast<const uchar4 *>(g.lab + c);
    const float4 bc = *reinterpret_cast<const float4 *>(b + c);
    const int lane = threadIdx.x & 63;
    float left = __shfl_up(xc.w, 1), right = __shfl_down(xc.x, 1);
    if (lane == 0 || q == 0) left = i > 0 ? x[c - 1] : 0.f;
    if (lane == 63 || q == nq - 1) right = i + 4 < g.nx ? x[c + 4] : 0.f;
    const float xs[6] = {left, xc.x, xc.y, xc.z, xc.w, right};
    const float a1[4] = {ym.x, ym.y, ym.z, ym.w}, a2[4] = {yp.x, yp.y, yp.z, yp.w};
    const float a3[4] = {zm.x, zm.y, zm.z, zm.w}, a4[4] = {zp.x, zp.y, zp.z, zp.w};
    const float bs[4] = {bc.x, bc.y, bc.z, bc.w};
    const unsigned ls[4] = {lab.x, lab.y, lab.z, lab.w};
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lap = 6.f * xs[e + 1] - (xs[e] + xs[e + 2] + a1[e] + a2[e] + a3[e] + a4[e]);
        const float v = xs[e + 1] + omega * ((bs[e] - lap) * (1.f / 6.f));
        r[e] = ls[e] == 0 ? v : xs[e + 1];  // boundary cells ignored here: timing only
    }
    *reinterpret_cast<float4 *>(out + c) = make_float4(r[0], r[1], r[2], r[3]);
}

// ---- variant M: z-marching, x(k-1),x(k),x(k+1) in registers, y neighbours from cache ----
// block = 64 x TY threads, each thread one quad; grid = (nx/256, ny/TY, nz/ZC)
template <int TY, int ZC, bool REMAP>
__global__ __launch_bounds__(64 * TY) void jacM(G g, float *__restrict__ out, const float *__restrict__ x,
                                                const float *__restrict__ b, float omega, unsigned nbx, unsigned nby,
                                                unsigned nbz)
{
    unsigned bid = blockIdx.x;
    const unsigned nb = nbx * nby * nbz;
    if (REMAP) bid = remap(bid, nb);
    const unsigned bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = int(bx) * 256 + lane * 4, j = int(by) * TY + ty;
    if (i >= g.nx || j >= g.ny) return;
    const size_t sy = g.nx, sz = size_t(g.nx) * g.ny;
    const int k0 = int(bz) * ZC, k1 = min(k0 + ZC, g.nz);
    size_t c = (size_t(k0) * g.ny + j) * sy + i;
    const ptrdiff_t dym = j > 0 ? -ptrdiff_t(sy) : 0, dyp = j < g.ny - 1 ? ptrdiff_t(sy) : 0;
    const bool hasL = lane > 0, hasR = lane < 63 && i + 4 < g.nx;
    float4 xm = *reinterpret_cast<const float4 *>(x + (k0 > 0 ? c - sz : c));
    float4 xc = *reinterpret_cast<const float4 *>(x + c);
    float4 ym = *reinterpret_cast<const float4 *>(x + c + dym);
    float4 yp = *reinterpret_cast<const float4 *>(x + c + dyp);
    float4 bc = *reinterpret_cast<const float4 *>(b + c);
    uchar4 lc = *reinterpret_cast<const uchar4 *>(g.lab + c);
    float le = i > 0 ? x[c - 1] : 0.f, re = i + 4 < g.nx ? x[c + 4] : 0.f;
    for (int k = k0; k < k1; ++k) {
        const size_t cn = (k + 1 < g.nz) ? c + sz : c;
        // issue the next plane's loads first
        const float4 xp = *reinterpret_cast<const float4 *>(x + cn);
        float4 ymn, ypn, bn;
        uchar4 ln;
        float len = 0.f, ren = 0.f;
        const bool more = k + 1 < k1;
        if (more) {
            ymn = *reinterpret_cast<const float4 *>(x + cn + dym);
            ypn = *reinterpret_cast<const float4 *>(x + cn + dyp);
            bn = *reinterpret_cast<const float4 *>(b + cn);
            ln = *reinterpret_cast<const uchar4 *>(g.lab + cn);
            if (!hasL) len = i > 0 ? x[cn - 1] : 0.f;
            if (!hasR) ren = i + 4 < g.nx ? x[cn + 4] : 0.f;
        }
        float left = __shfl_up(xc.w, 1), right = __shfl_down(xc.x, 1);
        if (!hasL) left = le;
        if (!hasR) right = re;
        const float xs[6] = {left, xc.x, xc.y, xc.z, xc.w, right};
        const float a1[4] = {ym.x, ym.y, ym.z, ym.w}, a2[4] = {yp.x, yp.y, yp.z, yp.w};
        const float a3[4] = {xm.x, xm.y, xm.z, xm.w}, a4[4] = {xp.x, xp.y, xp.z, xp.w};
        const float bs[4] = {bc.x, bc.y, bc.z, bc.w};
        const unsigned ls[4] = {lc.x, lc.y, lc.z, lc.w};
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lap = 6.f * xs[e + 1] - (xs[e] + xs[e + 2] + a1[e] + a2[e] + a3[e] + a4[e]);
            const float v = xs[e + 1] + omega * ((bs[e] - lap) * (1.f / 6.f));
            r[e] = ls[e] == 0 ? v : xs[e + 1];
        }
        *reinterpret_cast<float4 *>(out + c) = make_float4(r[0], r[1], r[2], r[3]);
        xm = xc;
        xc = xp;
        if (more) {
            ym = ymn;
            yp = ypn;
            bc = bn;
            lc = ln;
            le = len;
            re = ren;
        }
        c = cn;
    }
}

// ---- variant L: z-marching with the current plane staged in LDS (1-cell halo), z in registers ----
// block = 64 x TY threads; LDS plane (TY+2) x (256+8) floats, double buffered
template <int TY, int ZC, bool REMAP>
__global__ __launch_bounds__(64 * TY) void jacL(G g, float *__restrict__ out, const float *__restrict__ x,
                                                const float *__restrict__ b, float omega, unsigned nbx, unsigned nby,
                                                unsigned nbz)
{
    constexpr int PITCH = 256 + 8;  // 4 floats of halo each side keeps 16-byte alignment
    __shared__ float plane[2][(TY + 2) * PITCH];
    unsigned bid = blockIdx.x;
    const unsigned nb = nbx * nby * nbz;
    if (REMAP) bid = remap(bid, nb);
    const unsigned bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = int(bx) * 256 + lane * 4, j = int(by) * TY + ty;
    const size_t sy = g.nx, sz = size_t(g.nx) * g.ny;
    const int k0 = int(bz) * ZC, k1 = min(k0 + ZC, g.nz);
    size_t c = (size_t(k0) * g.ny + j) * sy + i;
    const ptrdiff_t dym = j > 0 ? -ptrdiff_t(sy) : 0, dyp = j < g.ny - 1 ? ptrdiff_t(sy) : 0;
    float4 xm = *reinterpret_cast<const float4 *>(x + (k0 > 0 ? c - sz : c));
    float4 xc = *reinterpret_cast<const float4 *>(x + c);
    float4 bc = *reinterpret_cast<const float4 *>(b + c);
    uchar4 lc = *reinterpret_cast<const uchar4 *>(g.lab + c);
    // halo values this thread is responsible for in the current plane
    float4 hy = make_float4(0, 0, 0, 0);
    if (ty == 0) hy = *reinterpret_cast<const float4 *>(x + c + dym);
    if (ty == TY - 1) hy = *reinterpret_cast<const float4 *>(x + c + dyp);
    float hx = 0.f;
    if (lane == 0) hx = i > 0 ? x[c - 1] : 0.f;
    if (lane == 63) hx = i + 4 < g.nx ? x[c + 4] : 0.f;
    int buf = 0;
    for (int k = k0; k < k1; ++k) {
        float *pl = plane[buf];
        float *me = pl + (ty + 1) * PITCH + 4 + lane * 4;
        *reinterpret_cast<float4 *>(me) = xc;
        if (ty == 0) *reinterpret_cast<float4 *>(me - PITCH) = hy;
        if (ty == TY - 1) *reinterpret_cast<float4 *>(me + PITCH) = hy;
        if (lane == 0) me[-1] = hx;
        if (lane == 63) me[4] = hx;
        const size_t cn = (k + 1 < g.nz) ? c + sz : c;
        const float4 xp = *reinterpret_cast<const float4 *>(x + cn);
        const bool more = k + 1 < k1;
        float4 bn = bc, hyn = hy;
        uchar4 ln = lc;
        float hxn = hx;
        if (more) {
            bn = *reinterpret_cast<const float4 *>(b + cn);
            ln = *reinterpret_cast<const uchar4 *>(g.lab + cn);
            if (ty == 0) hyn = *reinterpret_cast<const float4 *>(x + cn + dym);
            if (ty == TY - 1) hyn = *reinterpret_cast<const float4 *>(x + cn + dyp);
            if (lane == 0) hxn = i > 0 ? x[cn - 1] : 0.f;
            if (lane == 63) hxn = i + 4 < g.nx ? x[cn + 4] : 0.f;
        }
        __syncthreads();
        const float4 ym = *reinterpret_cast<const float4 *>(me - PITCH);
        const float4 yp = *reinterpret_cast<const float4 *>(me + PITCH);
        const float left = me[-1], right = me[4];
        const float xs[6] = {left, xc.x, xc.y, xc.z, xc.w, right};
        const float a1[4] = {ym.x, ym.y, ym.z, ym.w}, a2[4] = {yp.x, yp.y, yp.z, yp.w};
        const float a3[4] = {xm.x, xm.y, xm.z, xm.w}, a4[4] = {xp.x, xp.y, xp.z, xp.w};
        const float bs[4] = {bc.x, bc.y, bc.z, bc.w};
        const unsigned ls[4] = {lc.x, lc.y, lc.z, lc.w};
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lap = 6.f * xs[e + 1] - (xs[e] + xs[e + 2] + a1[e] + a2[e] + a3[e] + a4[e]);
            const float v = xs[e + 1] + omega * ((bs[e] - lap) * (1.f / 6.f));
            r[e] = ls[e] == 0 ? v : xs[e + 1];
        }
        *reinterpret_cast<float4 *>(out + c) = make_float4(r[0], r[1], r[2], r[3]);
        xm = xc;
        xc = xp;
        bc = bn;
        lc = ln;
        hy = hyn;
        hx = hxn;
        c = cn;
        buf ^= 1;
    }
}

// ---- band gather variants ----
template <bool REMAP>
__global__ void bandK(G g, const float *__restrict__ x, const float *__restrict__ b, const int32_t *__restrict__ band,
                      int nband, float *__restrict__ tmp, float omega, unsigned nblocks)
{
    const unsigned block = REMAP ? remap(blockIdx.x, nblocks) : blockIdx.x;
    const int t = block * blockDim.x + threadIdx.x;
    if (t >= nband) return;
    const size_t c = size_t(band[t]);
    const size_t sy = size_t(g.nx), sz = size_t(g.nx) * g.ny;
    const float xc = x[c];
    const float lap = 6.f * xc - (x[c - 1] + x[c + 1] + x[c - sy] + x[c + sy] + x[c - sz] + x[c + sz]);
    const float v = xc + omega * ((b[c] - lap) * (1.f / 6.f));
    tmp[t] = g.lab[c] == 0 ? v : xc;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int pad = argc > 2 ? atoi(argv[2]) : 16;
    const int reps = 20;
    const size_t cells = size_t(n) * n * n;
    printf("kbench N=%d cells=%zu\n", n, cells);
    float *x, *b, *o, *tmp;
    uint8_t *lab;
    int32_t *band;
    CK(hipMalloc(&x, cells * 4));
    CK(hipMalloc(&b, cells * 4));
    CK(hipMalloc(&o, cells * 4));
    CK(hipMalloc(&lab, cells));
    std::vector<float> hx(cells), hb(cells);
    std::vector<uint8_t> hl(cells, 1);
    std::vector<int32_t> hband;
    uint32_t s = 12345;
    auto rnd = [&] {
        s = s * 1664525u + 1013904223u;
        return float(s >> 8) / float(1 << 24);
    };
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                const size_t c = (size_t(k) * n + j) * n + i;
                const int lo = pad + 1, hi = n - pad - 1;
                const bool in = i >= lo && i < hi && j >= lo && j < hi && k >= lo && k < hi;
                const bool shell = i >= pad && i < n - pad && j >= pad && j < n - pad && k >= pad && k < n - pad;
                hl[c] = in ? 0 : (shell ? 2 : 1);
                if (in) {
                    const bool edge = i == lo || i == hi - 1 || j == lo || j == hi - 1 || k == lo || k == hi - 1;
                    if (edge) hl[c] = 3;
                }
                hx[c] = in ? rnd() : 0.f;
                hb[c] = in ? rnd() : 0.f;
            }
    // band = 3 layers, tile order
    {
        const int T = 16, tn = n / T;
        for (int t = 0; t < tn * tn * tn; ++t) {
            const int ti = t % tn, tj = (t / tn) % tn, tk = t / (tn * tn);
            for (int k = tk * T; k < (tk + 1) * T; ++k)
                for (int j = tj * T; j < (tj + 1) * T; ++j)
                    for (int i = ti * T; i < (ti + 1) * T; ++i) {
                        const int lo = pad + 1, hi = n - pad - 1;
                        const bool in = i >= lo && i < hi && j >= lo && j < hi && k >= lo && k < hi;
                        if (!in) continue;
                        const int d = std::min({i - lo, hi - 1 - i, j - lo, hi - 1 - j, k - lo, hi - 1 - k});
                        if (d < 3) hband.push_back(int32_t((size_t(k) * n + j) * n + i));
                    }
        }
    }
    const int nband = int(hband.size());
    printf("band cells %d\n", nband);
    CK(hipMalloc(&band, size_t(nband) * 4));
    CK(hipMalloc(&tmp, size_t(nband) * 4));
    CK(hipMemcpy(x, hx.data(), cells * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), cells * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lab, hl.data(), cells, hipMemcpyHostToDevice));
    CK(hipMemcpy(band, hband.data(), size_t(nband) * 4, hipMemcpyHostToDevice));
    G g{n, n, n, lab};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ref(cells), got(cells);
    bool haveRef = false;
    auto run = [&](const char *name, double bytesPerCell, bool check, std::function<void()> launch) {
        CK(hipMemset(o, 0, cells * 4));
        for (int w = 0; w < 3; ++w) launch();
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms);
        }
        CK(hipGetLastError());
        std::sort(ts.begin(), ts.end());
        const float med = ts[ts.size() / 2], mn = ts[0];
        double maxdiff = -1;
        if (check) {
            CK(hipMemcpy(got.data(), o, cells * 4, hipMemcpyDeviceToHost));
            if (!haveRef) {
                ref = got;
                haveRef = true;
            }
            maxdiff = 0;
            for (size_t c = 0; c < cells; ++c) maxdiff = std::max(maxdiff, double(fabsf(got[c] - ref[c])));
        }
        printf("%-34s med %8.3f ms  min %8.3f ms  %8.1f GB/s (alg %.1f B/cell)  maxdiff %g\n", name, med, mn,
               bytesPerCell * double(cells) / (med * 1e-3) / 1e9, bytesPerCell, maxdiff);
    };
    const size_t nq = cells / 4;
    const unsigned nbq = unsigned((nq + 255) / 256);
    run("copy float4 (8 B/cell)", 8, false, [&] { copyK<<<nbq, 256>>>((float4 *)o, (const float4 *)x, nq); });
    run("in place a += c (8 B/cell)", 8, false, [&] { inplaceK<<<nbq, 256>>>((float4 *)o, nq); });
    run("in place, nontemporal (8 B/cell)", 8, false, [&] { inplaceNtK<<<nbq, 256>>>((float4 *)o, nq); });
    run("stream3 x,b,lab->out (13 B/cell)", 13, false,
        [&] { stream3K<<<nbq, 256>>>((float4 *)o, (const float4 *)x, (const float4 *)b, (const uchar4 *)lab, nq); });
    run("jacA remap", 13, true, [&] { jacA<true><<<nbq, 256>>>(g, o, x, b, 0.6666667f, nbq); });
    run("jacA noremap", 13, true, [&] { jacA<false><<<nbq, 256>>>(g, o, x, b, 0.6666667f, nbq); });
#define RUN_M(KER, TY, ZC, RM)                                                                       \
    {                                                                                                \
        const unsigned nbx = (n + 255) / 256, nby = (n + TY - 1) / TY, nbz = (n + ZC - 1) / ZC;      \
        run(#KER " TY=" #TY " ZC=" #ZC " remap=" #RM, 13, true, [&] {                                \
            KER<TY, ZC, RM><<<nbx * nby * nbz, 64 * TY>>>(g, o, x, b, 0.6666667f, nbx, nby, nbz);    \
        });                                                                                          \
    }
    RUN_M(jacM, 4, 16, true)
    RUN_M(jacM, 4, 32, true)
    RUN_M(jacM, 4, 64, true)
    RUN_M(jacM, 4, 32, false)
    RUN_M(jacM, 8, 32, true)
    RUN_M(jacM, 2, 32, true)
    RUN_M(jacM, 1, 32, true)
    RUN_M(jacL, 4, 32, true)
    RUN_M(jacL, 8, 32, true)
    RUN_M(jacL, 8, 64, true)
    RUN_M(jacL, 16, 32, true)
    RUN_M(jacL, 4, 32, false)
    {
        const unsigned nb = (nband + 255) / 256;
        auto runb = [&](const char *name, std::function<void()> launch) {
            for (int w = 0; w < 3; ++w) launch();
            CK(hipDeviceSynchronize());
            std::vector<float> ts;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf("%-34s med %8.3f ms  min %8.3f ms  (%d cells, %.1f cells/ns)\n", name, ts[reps / 2], ts[0], nband,
                   nband / (ts[reps / 2] * 1e6));
        };
        runb("band gather noremap", [&] { bandK<false><<<nb, 256>>>(g, x, b, band, nband, tmp, 0.66f, nb); });
        runb("band gather remap", [&] { bandK<true><<<nb, 256>>>(g, x, b, band, nband, tmp, 0.66f, nb); });
    }
    return 0;
}
