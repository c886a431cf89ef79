// ASan/UBSan driver for the host-side set-up code (no GPU): hierarchy, band lists, band groups, slab levels + halos
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "mgps_internal.h"
using namespace mgps;
static std::vector<uint8_t> cube(int n, int pad, std::mt19937 &rng, bool noisy)
{
    std::vector<uint8_t> lab(size_t(n) * n * n, MGPS_EXTERIOR_CELL);
    std::uniform_real_distribution<float> u(0, 1);
    for (int k = pad; k < n - pad; ++k)
        for (int j = pad; j < n - pad; ++j)
            for (int i = pad; i < n - pad; ++i) {
                const bool shell = k == pad || j == pad || i == pad || k == n - pad - 1 || j == n - pad - 1 || i == n - pad - 1;
                uint8_t l = shell ? MGPS_DIRICHLET_CELL : MGPS_INTERIOR_CELL;
                if (!shell && noisy && u(rng) < 0.02f) l = u(rng) < 0.5f ? MGPS_EXTERIOR_CELL : MGPS_DIRICHLET_CELL;
                lab[(size_t(k) * n + j) * n + i] = l;
            }
    return lab;
}
int main()
{
    std::mt19937 rng(5);
    for (int trial = 0; trial < 6; ++trial) {
        const int n = trial < 3 ? 64 : 96, levels = trial < 3 ? 3 : 4, pad = 1 << (levels - 1);
        std::vector<uint8_t> lab = cube(n, pad, rng, trial % 3 != 0);
        const size_t nn = lab.size();
        std::vector<float> wx(size_t(n + 1) * n * n, 1.f), wy(wx.size(), 1.f), wz(wx.size(), 1.f);
        mgps_set_boundary_labels(lab.data(), wx.data(), wy.data(), wz.data(), n, n, n);
        mgps_options o;
        mgps_default_options(&o);
        o.band_width = 1 + trial % 4;
        mgps_hierarchy *H = nullptr;
        int rc = mgps_hierarchy_create(&H, n, n, n, lab.data(), levels, &o);
        if (rc != 0) { std::printf("trial %d: create rc %d (%s)\n", trial, rc, lastGlobalError()); continue; }
        for (int l = 0; l < mgps_hierarchy_levels(H); ++l)
            for (int depth = 1; depth <= 4; ++depth) {  // the box form of the fused band stage: builder + bit-exact replay
                int64_t g = 0, cells = 0, general = 0;
                rc = mgps_hierarchy_check_band_boxes(H, l, depth, l == 0 ? wx.data() : nullptr, l == 0 ? wy.data() : nullptr, l == 0 ? wz.data() : nullptr, &g, &cells,
                                                     &general);
                if (rc != 0) { std::printf("trial %d level %d depth %d: rc %d (%s)\n", trial, l, depth, rc, lastGlobalError()); return 1; }
            }
        // slab levels of the host builder (the checker of the device-side slab set-up) for 2 and 4 ranks
        for (int P : {2, 4}) {
            const int nzl = n / P;
            if (nzl % 16) continue;
            for (int r = 0; r < P; ++r) {
                HostLevel L;
                buildSlabLevel(H->lv[0], r * nzl, (r + 1) * nzl, nullptr, nullptr, nullptr, L);
            }
        }
        std::printf("trial %d ok: levels %d band0 %lld nn %zu\n", trial, mgps_hierarchy_levels(H), (long long)mgps_hierarchy_band_count(H, 0), nn);
        mgps_hierarchy_destroy(H);
    }
    return 0;
}
