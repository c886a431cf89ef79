// hipMalloc / hipFree cost by size (what a solver that is rebuilt every sub-step pays for its grids): hipcc -O2 tools/allocbench.hip -o tools/allocbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(nullptr);
    for (size_t mb : {1, 16, 256, 1024, 4096}) {
        for (int rep = 0; rep < 3; ++rep) {
            void *p = nullptr;
            double t0 = now();
            hipMalloc(&p, mb << 20);
            double t1 = now();
            hipMemset(p, 0, mb << 20);
            hipDeviceSynchronize();
            double t2 = now();
            hipFree(p);
            double t3 = now();
            std::printf("%5zu MB: malloc %8.2f ms  first memset %8.2f ms  free %8.2f ms\n", mb, t1 - t0, t2 - t1, t3 - t2);
        }
    }
    // ten 4 GB blocks at once, as a 1024^3 solver holds them
    std::vector<void *> ps(10);
    double t0 = now();
    for (auto &p : ps) hipMalloc(&p, size_t(4096) << 20);
    double t1 = now();
    for (auto &p : ps) hipFree(p);
    double t2 = now();
    std::printf("10 x 4096 MB: malloc %8.2f ms  free %8.2f ms\n", t1 - t0, t2 - t1);
    t0 = now();
    for (auto &p : ps) hipMalloc(&p, size_t(4096) << 20);
    t1 = now();
    for (auto &p : ps) hipFree(p);
    t2 = now();
    std::printf("again       : malloc %8.2f ms  free %8.2f ms\n", t1 - t0, t2 - t1);
    return 0;
}
