// The device-resident solver object and the C ABI (include/mgps.h): level storage in HBM, the
// V-cycle schedule of GeometricMultigridPoissonSolver::applyVCycle (MG.cpp:420-881), the PCG driver
// of solveGeometricConjugateGradient (CG.h:18-207).  Host orchestration only -- every arithmetic
// step is a HIP kernel from mgps_kernels.hip; there is no CPU fallback: without a HIP device
// mgps_create fails with MGPS_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mgps_internal.h"

using namespace mgps;

namespace {

struct DevLevel {
    Dims d;
    GridP g{};
    uint8_t *lab = nullptr;
    float *x = nullptr, *b = nullptr;  // coarse levels only; level 0 works on the caller's grids
    float *r = nullptr, *tmp = nullptr;
    int32_t *band = nullptr;  // device-ordered band list: BOUNDARY cells first
    int nband = 0;
    float *bandTmp = nullptr;
    float *rows = nullptr;    // 7 x numBoundary operator rows of the general BOUNDARY cells
    uint8_t *bandDiag = nullptr;
    // Gauss-Seidel tile lists per colour [0] = even, [1] = odd tiles
    int32_t *pure[2] = {nullptr, nullptr}, *mixed[2] = {nullptr, nullptr};
    int npure[2] = {0, 0}, nmixed[2] = {0, 0};
    int32_t *tileBndStart = nullptr;
};

}  // namespace

struct mgps_solver {
    mgps_hierarchy *hier = nullptr;
    mgps_options opt{};
    bool useGS = false;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<DevLevel> lv;
    float *w[3] = {nullptr, nullptr, nullptr};
    // coarsest-level dense inverse
    int cn = 0;
    float *cinv = nullptr, *cvec = nullptr;
    int32_t *ccells = nullptr;
    // reductions
    double *partials = nullptr, *resultDev = nullptr, *resultHost = nullptr;
    // PCG work grids (allocated on first use): r, p, z, t (CG.h:43, 67, 92, 96) and 1/diag
    float *pcg[4] = {nullptr, nullptr, nullptr, nullptr};
    float *dinv = nullptr;
    std::vector<void *> userGrids;
    // measurement hooks: event pairs around the fine-level full-domain smoother
    bool profiling = false;
    std::vector<hipEvent_t> profEvents;  // start/stop pairs
    size_t profUsed = 0;
    std::string lastError = "";
};

namespace {

int failH(mgps_solver *h, int code, const std::string &msg)
{
    if (h) h->lastError = msg;
    else setLastGlobalError(msg);
    return code;
}

#define MGPS_HIP(h, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return failH(h, MGPS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));              \
    } while (0)
#define MGPS_LAUNCH(h, call)                                                                               \
    do {                                                                                                   \
        int e_ = (call);                                                                                   \
        if (e_ != 0)                                                                                       \
            return failH(h, MGPS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(hipError_t(e_)));  \
    } while (0)
#define MGPS_TRY(call)                 \
    do {                               \
        int s_ = (call);               \
        if (s_ != MGPS_OK) return s_;  \
    } while (0)

template <class T>
int devAlloc(mgps_solver *h, T **p, size_t count, bool zero)
{
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
    if (e != hipSuccess) return failH(h, MGPS_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (zero) MGPS_HIP(h, hipMemsetAsync(*p, 0, count * sizeof(T), h->stream));
    return MGPS_OK;
}

template <class T>
int devUpload(mgps_solver *h, T **p, const std::vector<T> &v)
{
    MGPS_TRY(devAlloc(h, p, v.size(), false));
    if (!v.empty()) MGPS_HIP(h, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return MGPS_OK;
}

void freeAll(mgps_solver *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    for (auto &L : h->lv) {
        hipFree(L.lab);
        hipFree(L.x);
        hipFree(L.b);
        hipFree(L.r);
        hipFree(L.tmp);
        hipFree(L.band);
        hipFree(L.bandTmp);
        hipFree(L.rows);
        hipFree(L.bandDiag);
        for (int c = 0; c < 2; ++c) {
            hipFree(L.pure[c]);
            hipFree(L.mixed[c]);
        }
        hipFree(L.tileBndStart);
    }
    for (int a = 0; a < 3; ++a) hipFree(h->w[a]);
    hipFree(h->cinv);
    hipFree(h->cvec);
    hipFree(h->ccells);
    hipFree(h->partials);
    hipFree(h->resultDev);
    if (h->resultHost) hipHostFree(h->resultHost);
    for (int q = 0; q < 4; ++q) hipFree(h->pcg[q]);
    hipFree(h->dinv);
    for (void *p : h->userGrids) hipFree(p);
    for (hipEvent_t e : h->profEvents) hipEventDestroy(e);
    mgps_hierarchy_destroy(h->hier);
    delete h;
}

int checkLevel(mgps_solver *h, int level, const char *who)
{
    if (!h) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, std::string(who) + ": NULL handle");
    if (level < 0 || level >= int(h->lv.size()))
        return failH(h, MGPS_ERR_INVALID_ARGUMENT, std::string(who) + ": level out of range");
    hipSetDevice(h->device);
    return MGPS_OK;
}

// ---- level operators ---------------------------------------------------------------------------

int bandPasses(mgps_solver *h, int l, float *x, const float *b)
{
    DevLevel &L = h->lv[l];
    for (int it = 0; it < h->opt.band_iterations; ++it)
        MGPS_LAUNCH(h, launchBandJacobi(h->stream, L.g, x, b, L.band, L.nband, L.bandTmp, h->opt.jacobi_weight));
    return MGPS_OK;
}

int gsHalfSweep(mgps_solver *h, int l, float *x, const float *b, int odd, int forward)
{
    DevLevel &L = h->lv[l];
    MGPS_LAUNCH(h, launchTiledGS(h->stream, L.g, x, b, L.pure[odd], L.npure[odd], L.mixed[odd], L.nmixed[odd],
                                 L.tileBndStart, forward));
    return MGPS_OK;
}

// 3 x band Jacobi -> full-domain smoother -> 3 x band Jacobi (MG.cpp:445-513 down, 806-879 up).
// Jacobi runs out of place: `cur` holds the current iterate, `other` the spare grid; they swap.
int smoothStroke(mgps_solver *h, int l, float *&cur, float *&other, const float *b, bool down)
{
    DevLevel &L = h->lv[l];
    MGPS_TRY(bandPasses(h, l, cur, b));
    const bool timed = h->profiling && l == 0;
    if (timed) {
        if (h->profUsed + 2 > h->profEvents.size()) {
            hipEvent_t e0, e1;
            MGPS_HIP(h, hipEventCreate(&e0));
            MGPS_HIP(h, hipEventCreate(&e1));
            h->profEvents.push_back(e0);
            h->profEvents.push_back(e1);
        }
        MGPS_HIP(h, hipEventRecord(h->profEvents[h->profUsed], h->stream));
    }
    if (h->useGS) {
        if (down) {  // odd tiles forward, then even tiles forward (MG.cpp:466-479)
            MGPS_TRY(gsHalfSweep(h, l, cur, b, 1, 1));
            MGPS_TRY(gsHalfSweep(h, l, cur, b, 0, 1));
        } else {  // even tiles backward, then odd tiles backward (MG.cpp:740-751)
            MGPS_TRY(gsHalfSweep(h, l, cur, b, 0, 0));
            MGPS_TRY(gsHalfSweep(h, l, cur, b, 1, 0));
        }
    } else {
        MGPS_LAUNCH(h, launchStencil(h->stream, OP_JACOBI, L.g, other, cur, b, h->opt.jacobi_weight));
        std::swap(cur, other);
    }
    if (timed) {
        MGPS_HIP(h, hipEventRecord(h->profEvents[h->profUsed + 1], h->stream));
        h->profUsed += 2;
    }
    MGPS_TRY(bandPasses(h, l, cur, b));
    return MGPS_OK;
}

int vcycle(mgps_solver *h, float *x, const float *b, bool useInitialGuess)
{
    const int L = int(h->lv.size());
    std::vector<float *> cur(L), other(L);
    cur[0] = x;
    other[0] = h->lv[0].tmp;
    if (!useInitialGuess) MGPS_HIP(h, hipMemsetAsync(x, 0, h->lv[0].d.cells() * sizeof(float), h->stream));  // MG.cpp:439
    MGPS_TRY(smoothStroke(h, 0, cur[0], other[0], b, true));
    if (L > 1) {
        const float *rhs = b;
        for (int l = 0; l < L - 1; ++l) {  // MG.cpp:519-553 (fine), 557-667 (coarser)
            DevLevel &F = h->lv[l], &C = h->lv[l + 1];
            if (l > 0) {
                cur[l] = F.x;
                other[l] = F.tmp;
                rhs = F.b;
                MGPS_HIP(h, hipMemsetAsync(F.x, 0, F.d.cells() * sizeof(float), h->stream));  // MG.cpp:566
                MGPS_TRY(smoothStroke(h, l, cur[l], other[l], rhs, true));
            }
            MGPS_LAUNCH(h, launchStencil(h->stream, OP_RESIDUAL, F.g, F.r, cur[l], rhs, 0.f));
            MGPS_LAUNCH(h, launchRestrict(h->stream, C.g, C.b, F.r));
        }
        DevLevel &B = h->lv[L - 1];  // direct solve, MG.cpp:669-692
        MGPS_LAUNCH(h, launchCoarseSolve(h->stream, h->cn, h->cinv, h->ccells, B.x, B.b, h->cvec));
        cur[L - 1] = B.x;
        for (int l = L - 2; l >= 0; --l) {  // MG.cpp:695-784 (coarser), 787-880 (fine)
            DevLevel &F = h->lv[l];
            MGPS_LAUNCH(h, launchProlongAdd(h->stream, F.g, cur[l], cur[l + 1]));
            MGPS_TRY(smoothStroke(h, l, cur[l], other[l], l == 0 ? b : F.b, false));
        }
    }
    if (cur[0] != x)  // single-level Jacobi cycle: the iterate ended in the spare grid
        MGPS_HIP(h, hipMemcpyAsync(x, cur[0], h->lv[0].d.cells() * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return MGPS_OK;
}

int reduceToHost(mgps_solver *h, int kind, int level, const float *a, const float *b, double *out)
{
    DevLevel &L = h->lv[level];
    MGPS_LAUNCH(h, launchReduce(h->stream, kind, L.g, a, b, h->partials, h->resultDev));
    MGPS_HIP(h, hipMemcpyAsync(h->resultHost, h->resultDev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    *out = *h->resultHost;
    return MGPS_OK;
}

int ensurePcgGrids(mgps_solver *h, bool needDiag)
{
    const size_t n = h->lv[0].d.cells();
    for (int q = 0; q < 4; ++q)
        if (!h->pcg[q]) MGPS_TRY(devAlloc(h, &h->pcg[q], n, true));
    if (needDiag && !h->dinv) {
        MGPS_TRY(devAlloc(h, &h->dinv, n, false));
        MGPS_LAUNCH(h, launchDiagInverse(h->stream, h->lv[0].g, h->dinv));
    }
    return MGPS_OK;
}

int pcg(mgps_solver *h, float *x, const float *b, double tol, int maxIt, bool useMG, mgps_pcg_stats *st)
{
    DevLevel &F = h->lv[0];
    const size_t bytes = F.d.cells() * sizeof(float);
    mgps_pcg_stats local{};
    if (!st) st = &local;
    std::memset(st, 0, sizeof(*st));
    MGPS_TRY(ensurePcgGrids(h, !useMG));
    float *r = h->pcg[0], *p = h->pcg[1], *z = h->pcg[2], *t = h->pcg[3];
    hipEvent_t e0, e1;
    MGPS_HIP(h, hipEventCreate(&e0));
    MGPS_HIP(h, hipEventCreate(&e1));
    MGPS_HIP(h, hipEventRecord(e0, h->stream));
    auto finish = [&](int outcome) {
        st->outcome = outcome;
        hipEventRecord(e1, h->stream);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        st->solve_ms = ms;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        return MGPS_OK;
    };
    auto precondition = [&](float *dst, const float *src) -> int {
        if (useMG) return vcycle(h, dst, src, false);  // Plug.cpp:468-472
        MGPS_LAUNCH(h, launchMulMasked(h->stream, F.g, dst, src, h->dinv));  // Plug.cpp:555-606
        return MGPS_OK;
    };

    double rhs2 = 0;
    MGPS_TRY(reduceToHost(h, 1, 0, b, nullptr, &rhs2));  // CG.h:35
    st->rhs_norm2 = rhs2;
    if (rhs2 == 0) return finish(MGPS_PCG_RHS_ZERO);  // CG.h:36-40
    MGPS_LAUNCH(h, launchStencil(h->stream, OP_RESIDUAL, F.g, r, x, b, 0.f));  // CG.h:50-51
    double res2 = 0;
    MGPS_TRY(reduceToHost(h, 1, 0, r, nullptr, &res2));  // CG.h:57
    const double threshold = tol * tol * rhs2;           // CG.h:58
    if (res2 < threshold) {                              // CG.h:60-64
        st->rel_residual = st->rel_residual_recomputed = std::sqrt(res2 / rhs2);
        return finish(MGPS_PCG_ALREADY_CONVERGED);
    }
    MGPS_HIP(h, hipMemsetAsync(p, 0, bytes, h->stream));  // CG.h:69
    MGPS_TRY(precondition(p, r));                         // CG.h:75
    double absNew = 0;
    MGPS_TRY(reduceToHost(h, 0, 0, p, r, &absNew));  // CG.h:86
    MGPS_HIP(h, hipMemsetAsync(z, 0, bytes, h->stream));
    MGPS_HIP(h, hipMemsetAsync(t, 0, bytes, h->stream));
    int it = 0;
    bool converged = false;
    for (; it < maxIt; ++it) {
        if (h->opt.interrupt && h->opt.interrupt(h->opt.interrupt_user)) {
            finish(MGPS_PCG_MAX_ITERATIONS);
            st->iterations = it;
            return failH(h, MGPS_ERR_INTERRUPTED, "mgps_solve_pcg: interrupted");
        }
        MGPS_LAUNCH(h, launchStencil(h->stream, OP_APPLY, F.g, t, p, nullptr, 0.f));  // CG.h:110
        double pAp = 0;
        MGPS_TRY(reduceToHost(h, 0, 0, p, t, &pAp));
        const double alpha = absNew / pAp;                                                // CG.h:121
        MGPS_LAUNCH(h, launchAxpy(h->stream, F.g, x, p, nullptr, float(alpha), 1.f));    // CG.h:132
        MGPS_LAUNCH(h, launchAxpy(h->stream, F.g, r, t, nullptr, float(alpha), -1.f));   // CG.h:143
        MGPS_TRY(reduceToHost(h, 1, 0, r, nullptr, &res2));                              // CG.h:153
        if (h->opt.print_stats) std::printf("  Iteration: %d  Relative error: %.10g\n", it, std::sqrt(res2 / rhs2));
        if (res2 < threshold) {  // CG.h:161 -- the counter is not advanced on the exit pass
            converged = true;
            break;
        }
        MGPS_TRY(precondition(z, r));  // CG.h:168
        const double absOld = absNew;
        MGPS_TRY(reduceToHost(h, 0, 0, z, r, &absNew));  // CG.h:180
        const double beta = absNew / absOld;
        MGPS_LAUNCH(h, launchXpay(h->stream, F.g, p, z, p, nullptr, float(beta)));  // CG.h:191
    }
    st->iterations = it;
    st->rel_residual = std::sqrt(res2 / rhs2);                                  // CG.h:199
    MGPS_LAUNCH(h, launchStencil(h->stream, OP_RESIDUAL, F.g, r, x, b, 0.f));  // CG.h:203-204
    double rec2 = 0;
    MGPS_TRY(reduceToHost(h, 1, 0, r, nullptr, &rec2));
    st->rel_residual_recomputed = std::sqrt(rec2 / rhs2);  // CG.h:205
    return finish(converged ? MGPS_PCG_CONVERGED : MGPS_PCG_MAX_ITERATIONS);
}

}  // namespace

extern "C" {

const char *mgps_last_error(const mgps_solver *h) { return h ? h->lastError.c_str() : lastGlobalError(); }

int mgps_device_count(int *count)
{
    if (!count) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_device_count: NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return MGPS_OK;
}

int mgps_create(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels_host, const float *wx_host,
                const float *wy_host, const float *wz_host, int mg_levels, int use_gauss_seidel,
                const mgps_options *opt)
{
    if (!out) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create: out is NULL");
    *out = nullptr;
    if (!labels_host || !wx_host || !wy_host || !wz_host)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create: labels and the three weight grids are required");
    mgps_options o;
    mgps_default_options(&o);
    if (opt) {
        if (opt->struct_size != int(sizeof(mgps_options)))
            return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_options.struct_size mismatch");
        o = *opt;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return failH(nullptr, MGPS_ERR_NO_DEVICE, "mgps_create: no HIP device is visible (this library has no CPU path)");
    int device = o.device;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) return failH(nullptr, MGPS_ERR_NO_DEVICE, "mgps_create: device ordinal out of range");

    mgps_hierarchy *hier = nullptr;
    MGPS_TRY(mgps_hierarchy_create(&hier, nx, ny, nz, labels_host, mg_levels, &o));
    {  // the fine-level invariants the reference asserts in debug builds (MG.cpp:234)
        int pass = 0;
        mgps_check_boundary_cells(labels_host, wx_host, wy_host, wz_host, nx, ny, nz, &pass);
        if (!pass) {
            mgps_hierarchy_destroy(hier);
            return failH(nullptr, MGPS_ERR_HIERARCHY,
                         "labels/weights violate the BOUNDARY-cell rules (unitTestBoundaryCells): run mgps_set_boundary_labels");
        }
    }
    auto *h = new mgps_solver();
    h->hier = hier;
    h->opt = o;
    h->useGS = use_gauss_seidel != 0;
    h->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        freeAll(h);
        return failH(nullptr, MGPS_ERR_NO_DEVICE, "mgps_create: hipSetDevice failed");
    }
    auto bail = [&](int code) {
        setLastGlobalError(h->lastError);
        freeAll(h);
        return code;
    };
#define CREATE_TRY(call)                        \
    do {                                        \
        int s_ = (call);                        \
        if (s_ != MGPS_OK) return bail(s_);     \
    } while (0)

    const Dims d0{nx, ny, nz};
    const size_t wn[3] = {size_t(nx + 1) * ny * nz, size_t(nx) * (ny + 1) * nz, size_t(nx) * ny * (nz + 1)};
    const float *wh[3] = {wx_host, wy_host, wz_host};
    for (int a = 0; a < 3; ++a) {
        CREATE_TRY(devAlloc(h, &h->w[a], wn[a], false));
        if (hipMemcpy(h->w[a], wh[a], wn[a] * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return bail(failH(h, MGPS_ERR_HIP, "mgps_create: weight upload failed"));
    }
    h->lv.resize(hier->levels);
    buildBoundaryRows(hier->lv[0], wx_host, wy_host, wz_host);  // fine level: the real face weights
    for (int l = 0; l < hier->levels; ++l) {
        const HostLevel &HL = hier->lv[l];
        DevLevel &L = h->lv[l];
        L.d = HL.d;
        CREATE_TRY(devUpload(h, &L.lab, HL.codes));
        CREATE_TRY(devUpload(h, &L.bandDiag, HL.bandDiag));
        CREATE_TRY(devUpload(h, &L.band, HL.bandDev));
        CREATE_TRY(devUpload(h, &L.rows, HL.rows));
        L.nband = int(HL.bandDev.size());
        CREATE_TRY(devAlloc(h, &L.bandTmp, HL.band.size(), false));
        CREATE_TRY(devUpload(h, &L.pure[0], HL.pureEven));
        CREATE_TRY(devUpload(h, &L.pure[1], HL.pureOdd));
        CREATE_TRY(devUpload(h, &L.mixed[0], HL.mixedEven));
        CREATE_TRY(devUpload(h, &L.mixed[1], HL.mixedOdd));
        L.npure[0] = int(HL.pureEven.size());
        L.npure[1] = int(HL.pureOdd.size());
        L.nmixed[0] = int(HL.mixedEven.size());
        L.nmixed[1] = int(HL.mixedOdd.size());
        CREATE_TRY(devUpload(h, &L.tileBndStart, HL.tileBndStart));
        if (l > 0) {
            CREATE_TRY(devAlloc(h, &L.x, L.d.cells(), true));
            CREATE_TRY(devAlloc(h, &L.b, L.d.cells(), true));
        }
        CREATE_TRY(devAlloc(h, &L.r, L.d.cells(), true));
        CREATE_TRY(devAlloc(h, &L.tmp, L.d.cells(), true));
        L.g = GridP{L.d.nx, L.d.ny, L.d.nz, L.lab, l == 0 ? h->w[0] : nullptr, l == 0 ? h->w[1] : nullptr,
                    l == 0 ? h->w[2] : nullptr, L.band, L.rows, int(HL.numBoundary), L.bandDiag};
    }
    (void)d0;
    hier->buildDenseInverse();
    h->cn = hier->coarseN;
    CREATE_TRY(devUpload(h, &h->cinv, hier->coarseInverse));
    CREATE_TRY(devUpload(h, &h->ccells, hier->coarseCell));
    CREATE_TRY(devAlloc(h, &h->cvec, size_t(h->cn), true));
    CREATE_TRY(devAlloc(h, &h->partials, size_t(kReducePartials), true));
    CREATE_TRY(devAlloc(h, &h->resultDev, 1, true));
    if (hipHostMalloc(reinterpret_cast<void **>(&h->resultHost), sizeof(double)) != hipSuccess)
        return bail(failH(h, MGPS_ERR_ALLOC, "mgps_create: pinned allocation failed"));
    if (hipDeviceSynchronize() != hipSuccess) return bail(failH(h, MGPS_ERR_HIP, "mgps_create: device synchronize failed"));
#undef CREATE_TRY
    *out = h;
    return MGPS_OK;
}

void mgps_destroy(mgps_solver *h) { freeAll(h); }
int mgps_levels(const mgps_solver *h) { return h ? int(h->lv.size()) : 0; }
const mgps_hierarchy *mgps_get_hierarchy(const mgps_solver *h) { return h ? h->hier : nullptr; }

int mgps_level_dims(const mgps_solver *h, int level, int out_dims[3])
{
    if (!h) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_level_dims: NULL handle");
    return mgps_hierarchy_level_dims(h->hier, level, out_dims);
}

int mgps_set_stream(mgps_solver *h, void *hip_stream)
{
    if (!h) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_set_stream: NULL handle");
    h->stream = static_cast<hipStream_t>(hip_stream);
    return MGPS_OK;
}

int mgps_synchronize(mgps_solver *h)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_synchronize"));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    return MGPS_OK;
}

int mgps_grid_alloc(mgps_solver *h, int level, float **out_dev)
{
    MGPS_TRY(checkLevel(h, level, "mgps_grid_alloc"));
    if (!out_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_grid_alloc: out is NULL");
    MGPS_TRY(devAlloc(h, out_dev, h->lv[level].d.cells(), true));
    h->userGrids.push_back(*out_dev);
    return MGPS_OK;
}

int mgps_grid_free(mgps_solver *h, float *dev)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_grid_free"));
    auto it = std::find(h->userGrids.begin(), h->userGrids.end(), static_cast<void *>(dev));
    if (it == h->userGrids.end()) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_grid_free: not a grid of this solver");
    h->userGrids.erase(it);
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    MGPS_HIP(h, hipFree(dev));
    return MGPS_OK;
}

int mgps_grid_upload(mgps_solver *h, int level, float *dst_dev, const float *src_host)
{
    MGPS_TRY(checkLevel(h, level, "mgps_grid_upload"));
    if (!dst_dev || !src_host) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_grid_upload: NULL pointer");
    MGPS_HIP(h, hipMemcpyAsync(dst_dev, src_host, h->lv[level].d.cells() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    return MGPS_OK;
}

int mgps_grid_download(mgps_solver *h, int level, float *dst_host, const float *src_dev)
{
    MGPS_TRY(checkLevel(h, level, "mgps_grid_download"));
    if (!dst_host || !src_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_grid_download: NULL pointer");
    MGPS_HIP(h, hipMemcpyAsync(dst_host, src_dev, h->lv[level].d.cells() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    return MGPS_OK;
}

int mgps_apply_vcycle(mgps_solver *h, float *x_dev, const float *b_dev, int use_initial_guess)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_apply_vcycle"));
    if (!x_dev || !b_dev || x_dev == b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_apply_vcycle: bad grid pointers");
    return vcycle(h, x_dev, b_dev, use_initial_guess != 0);
}

int mgps_jacobi_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev)
{
    MGPS_TRY(checkLevel(h, level, "mgps_jacobi_smooth"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_jacobi_smooth: NULL grid");
    DevLevel &L = h->lv[level];
    MGPS_LAUNCH(h, launchStencil(h->stream, OP_JACOBI, L.g, L.tmp, x_dev, b_dev, h->opt.jacobi_weight));
    MGPS_HIP(h, hipMemcpyAsync(x_dev, L.tmp, L.d.cells() * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return MGPS_OK;
}

int mgps_tiled_gs_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev, int smooth_odd_tiles,
                         int smooth_forward)
{
    MGPS_TRY(checkLevel(h, level, "mgps_tiled_gs_smooth"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_tiled_gs_smooth: NULL grid");
    return gsHalfSweep(h, level, x_dev, b_dev, smooth_odd_tiles ? 1 : 0, smooth_forward != 0);
}

int mgps_boundary_jacobi_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev)
{
    MGPS_TRY(checkLevel(h, level, "mgps_boundary_jacobi_smooth"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_boundary_jacobi_smooth: NULL grid");
    DevLevel &L = h->lv[level];
    MGPS_LAUNCH(h, launchBandJacobi(h->stream, L.g, x_dev, b_dev, L.band, L.nband, L.bandTmp, h->opt.jacobi_weight));
    return MGPS_OK;
}

int mgps_apply_poisson(mgps_solver *h, int level, float *y_dev, const float *x_dev)
{
    MGPS_TRY(checkLevel(h, level, "mgps_apply_poisson"));
    if (!y_dev || !x_dev || y_dev == x_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_apply_poisson: bad grid pointers");
    MGPS_LAUNCH(h, launchStencil(h->stream, OP_APPLY, h->lv[level].g, y_dev, x_dev, nullptr, 0.f));
    return MGPS_OK;
}

int mgps_residual(mgps_solver *h, int level, float *r_dev, const float *x_dev, const float *b_dev)
{
    MGPS_TRY(checkLevel(h, level, "mgps_residual"));
    if (!r_dev || !x_dev || !b_dev || r_dev == x_dev)
        return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_residual: bad grid pointers");
    MGPS_LAUNCH(h, launchStencil(h->stream, OP_RESIDUAL, h->lv[level].g, r_dev, x_dev, b_dev, 0.f));
    return MGPS_OK;
}

int mgps_downsample(mgps_solver *h, int fine_level, float *coarse_dev, const float *fine_dev)
{
    MGPS_TRY(checkLevel(h, fine_level + 1, "mgps_downsample"));
    if (fine_level < 0 || !coarse_dev || !fine_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_downsample: bad arguments");
    MGPS_LAUNCH(h, launchRestrict(h->stream, h->lv[fine_level + 1].g, coarse_dev, fine_dev));
    return MGPS_OK;
}

int mgps_upsample_add(mgps_solver *h, int fine_level, float *fine_dev, const float *coarse_dev)
{
    MGPS_TRY(checkLevel(h, fine_level + 1, "mgps_upsample_add"));
    if (fine_level < 0 || !coarse_dev || !fine_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_upsample_add: bad arguments");
    MGPS_LAUNCH(h, launchProlongAdd(h->stream, h->lv[fine_level].g, fine_dev, coarse_dev));
    return MGPS_OK;
}

int mgps_coarse_solve(mgps_solver *h, float *x_dev, const float *b_dev)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_coarse_solve"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_coarse_solve: NULL grid");
    MGPS_LAUNCH(h, launchCoarseSolve(h->stream, h->cn, h->cinv, h->ccells, x_dev, b_dev, h->cvec));
    return MGPS_OK;
}

int mgps_dot(mgps_solver *h, int level, const float *a_dev, const float *b_dev, double *out)
{
    MGPS_TRY(checkLevel(h, level, "mgps_dot"));
    if (!a_dev || !b_dev || !out) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_dot: NULL pointer");
    return reduceToHost(h, 0, level, a_dev, b_dev, out);
}

int mgps_squared_l2_norm(mgps_solver *h, int level, const float *a_dev, double *out)
{
    MGPS_TRY(checkLevel(h, level, "mgps_squared_l2_norm"));
    if (!a_dev || !out) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_squared_l2_norm: NULL pointer");
    return reduceToHost(h, 1, level, a_dev, nullptr, out);
}

int mgps_l2_norm(mgps_solver *h, int level, const float *a_dev, double *out)
{
    MGPS_TRY(mgps_squared_l2_norm(h, level, a_dev, out));
    *out = std::sqrt(*out);  // Ops.h:1202
    return MGPS_OK;
}

int mgps_inf_norm(mgps_solver *h, int level, const float *a_dev, int reference_signed_max, double *out)
{
    MGPS_TRY(checkLevel(h, level, "mgps_inf_norm"));
    if (!a_dev || !out) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_inf_norm: NULL pointer");
    return reduceToHost(h, reference_signed_max ? 2 : 3, level, a_dev, nullptr, out);
}

int mgps_add_to_vector(mgps_solver *h, int level, float *dst_dev, const float *src_dev, double scale)
{
    MGPS_TRY(checkLevel(h, level, "mgps_add_to_vector"));
    if (!dst_dev || !src_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_add_to_vector: NULL grid");
    MGPS_LAUNCH(h, launchAxpy(h->stream, h->lv[level].g, dst_dev, src_dev, nullptr, float(scale), 1.f));
    return MGPS_OK;
}

int mgps_add_vectors(mgps_solver *h, int level, float *dst_dev, const float *a_dev, const float *scaled_dev, double scale)
{
    MGPS_TRY(checkLevel(h, level, "mgps_add_vectors"));
    if (!dst_dev || !a_dev || !scaled_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_add_vectors: NULL grid");
    MGPS_LAUNCH(h, launchXpay(h->stream, h->lv[level].g, dst_dev, a_dev, scaled_dev, nullptr, float(scale)));
    return MGPS_OK;
}

int mgps_scale_vector(mgps_solver *h, int level, float *v_dev, double scale)
{
    MGPS_TRY(checkLevel(h, level, "mgps_scale_vector"));
    if (!v_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_scale_vector: NULL grid");
    MGPS_LAUNCH(h, launchScale(h->stream, h->lv[level].g, v_dev, float(scale)));
    return MGPS_OK;
}

int mgps_solve_pcg(mgps_solver *h, float *x_dev, const float *b_dev, double tolerance, int max_iterations,
                   int use_mg_preconditioner, mgps_pcg_stats *stats)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_solve_pcg"));
    if (!x_dev || !b_dev || x_dev == b_dev || !(tolerance >= 0) || max_iterations < 0)
        return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_solve_pcg: bad arguments");
    return pcg(h, x_dev, b_dev, tolerance, max_iterations, use_mg_preconditioner != 0, stats);
}

int mgps_profile_enable(mgps_solver *h, int enable)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_profile_enable"));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    h->profiling = enable != 0;
    h->profUsed = 0;
    return MGPS_OK;
}

int mgps_profile_read(mgps_solver *h, double *fine_smoother_ms, int *fine_smoother_launches)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_profile_read"));
    if (!fine_smoother_ms || !fine_smoother_launches)
        return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_profile_read: NULL pointer");
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    double total = 0;
    for (size_t q = 0; q + 1 < h->profUsed; q += 2) {
        float ms = 0.f;
        MGPS_HIP(h, hipEventElapsedTime(&ms, h->profEvents[q], h->profEvents[q + 1]));
        total += ms;
    }
    *fine_smoother_ms = total;
    *fine_smoother_launches = int(h->profUsed / 2);
    h->profUsed = 0;
    return MGPS_OK;
}

static int withHostGrids(mgps_solver *h, float *x_host, const float *b_host, bool uploadX,
                         int (*body)(mgps_solver *, float *, const float *, void *), void *ctx)
{
    if (!x_host || !b_host) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "host form: NULL pointer");
    float *xd = nullptr, *bd = nullptr;
    MGPS_TRY(mgps_grid_alloc(h, 0, &xd));
    int rc = mgps_grid_alloc(h, 0, &bd);
    if (rc == MGPS_OK) rc = mgps_grid_upload(h, 0, bd, b_host);
    if (rc == MGPS_OK && uploadX) rc = mgps_grid_upload(h, 0, xd, x_host);
    if (rc == MGPS_OK) rc = body(h, xd, bd, ctx);
    if (rc == MGPS_OK) rc = mgps_grid_download(h, 0, x_host, xd);
    const std::string keep = h->lastError;
    if (bd) mgps_grid_free(h, bd);
    mgps_grid_free(h, xd);
    if (rc != MGPS_OK) h->lastError = keep;
    return rc;
}

int mgps_apply_vcycle_host(mgps_solver *h, float *x_host, const float *b_host, int use_initial_guess)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_apply_vcycle_host"));
    int guess = use_initial_guess;
    return withHostGrids(
        h, x_host, b_host, use_initial_guess != 0,
        [](mgps_solver *hh, float *xd, const float *bd, void *c) { return mgps_apply_vcycle(hh, xd, bd, *static_cast<int *>(c)); },
        &guess);
}

struct PcgHostCtx {
    double tol;
    int maxIt, useMG;
    mgps_pcg_stats *stats;
};

int mgps_solve_pcg_host(mgps_solver *h, float *x_host, const float *b_host, double tolerance, int max_iterations,
                        int use_mg_preconditioner, mgps_pcg_stats *stats)
{
    MGPS_TRY(checkLevel(h, 0, "mgps_solve_pcg_host"));
    PcgHostCtx ctx{tolerance, max_iterations, use_mg_preconditioner, stats};
    return withHostGrids(
        h, x_host, b_host, true,
        [](mgps_solver *hh, float *xd, const float *bd, void *c) {
            auto *p = static_cast<PcgHostCtx *>(c);
            return mgps_solve_pcg(hh, xd, bd, p->tol, p->maxIt, p->useMG, p->stats);
        },
        &ctx);
}

}  // extern "C"
