// flattenGrid / unflattenGrid: the conversion between a tiled voxel array (Houdini's UT_VoxelArray<T>, 16^3 tiles,
// constant-tile compressed) and the dense x-fastest arrays of the C ABI (include/mgps.h: index = (k*ny + j)*nx + i,
// the logical order UT_VoxelArray exposes).  Header only and HDK-free: written against the four members of
// UT_VoxelArray the conversion needs --
//     int  getXRes() / getYRes() / getZRes() const;
//     T    getValue(int x, int y, int z) const;
//     void setValue(int x, int y, int z, T value);
//     void size(int xres, int yres, int zres);          (unflattenGrid with resize only)
// -- so the same template serves UT_VoxelArray<int>, UT_VoxelArray<fpreal32 / fpreal64> (SIM_RawField::field(),
// SIM_RawIndexField::field()) in the plugin and a plain stand-in in tests/cpp/flatten_roundtrip.cpp.
// Reference use: the labels / weights / rhs / solution grids of Plug.cpp:297-418 are UT_VoxelArray<int | double>.
#pragma once

#include <cstddef>
#include <thread>
#include <vector>

namespace mgps {

// run fn(k0, k1) over [0, nz) on a few host threads (the HDK build may swap this for UTparallelFor; plain threads keep
// the header free of HDK includes)
template <class Fn>
inline void forEachPlaneRange(int nz, Fn fn)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = int(hw == 0 ? 1 : (hw > 16 ? 16 : hw));
    if (nt <= 1 || nz < 2 * nt) {
        fn(0, nz);
        return;
    }
    std::vector<std::thread> pool;
    const int chunk = (nz + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int k0 = t * chunk, k1 = k0 + chunk < nz ? k0 + chunk : nz;
        if (k0 >= k1) break;
        pool.emplace_back([=] { fn(k0, k1); });
    }
    for (auto &th : pool) th.join();
}

// out[(k*ny + j)*nx + i] = U(grid(i, j, k)) into a buffer of nx*ny*nz entries the caller provides (e.g. page-locked memory
// from mgps_host_alloc: uploads from it run at PCIe speed).  U may differ from the array's value type (int labels ->
// uint8_t, double -> float).
template <class U, class VoxelArray>
inline void flattenGrid(U *out, const VoxelArray &grid)
{
    const int nx = grid.getXRes(), ny = grid.getYRes(), nz = grid.getZRes();
    forEachPlaneRange(nz, [&grid, out, nx, ny](int k0, int k1) {
        for (int k = k0; k < k1; ++k)
            for (int j = 0; j < ny; ++j) {
                U *row = out + (size_t(k) * ny + j) * nx;
                for (int i = 0; i < nx; ++i) row[i] = U(grid.getValue(i, j, k));
            }
    });
}
// the same into a vector, which is resized
template <class U, class VoxelArray>
inline void flattenGrid(std::vector<U> &flat, const VoxelArray &grid)
{
    flat.resize(size_t(grid.getXRes()) * grid.getYRes() * grid.getZRes());
    flattenGrid(flat.data(), grid);
}

// the inverse: grid(i, j, k) = T(flat[(k*ny + j)*nx + i]).  The array must already have the extents of `flat`
// (nx * ny * nz entries); returns false, touching nothing, if it does not.  Writes go plane range by plane range: a
// 16^3 tile of UT_VoxelArray spans 16 planes, so ranges are cut at multiples of 16 -- two threads never write one tile
// (UT_VoxelArray::setValue may decompress the tile it writes into).
template <class VoxelArray, class U>
inline bool unflattenGrid(VoxelArray &grid, const U *in, size_t count)
{
    const int nx = grid.getXRes(), ny = grid.getYRes(), nz = grid.getZRes();
    if (count != size_t(nx) * ny * nz) return false;
    using T = decltype(grid.getValue(0, 0, 0));
    const int tilePlanes = 16, ntiles = (nz + tilePlanes - 1) / tilePlanes;
    forEachPlaneRange(ntiles, [&grid, in, nx, ny, nz, tilePlanes](int t0, int t1) {
        const int k1 = t1 * tilePlanes < nz ? t1 * tilePlanes : nz;
        for (int k = t0 * tilePlanes; k < k1; ++k)
            for (int j = 0; j < ny; ++j) {
                const U *row = in + (size_t(k) * ny + j) * nx;
                for (int i = 0; i < nx; ++i) grid.setValue(i, j, k, T(row[i]));
            }
    });
    return true;
}
template <class VoxelArray, class U>
inline bool unflattenGrid(VoxelArray &grid, const std::vector<U> &flat)
{
    return unflattenGrid(grid, flat.data(), flat.size());
}

}  // namespace mgps
