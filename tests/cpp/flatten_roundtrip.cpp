// Round trip of host/mgps_voxel_flatten.hpp on a stand-in for UT_VoxelArray (the four members the header names, stored
// in 16^3 tiles like the real thing so that index bugs do not cancel): flatten -> unflatten -> flatten must reproduce the
// values, with type conversion (int labels -> uint8_t, double -> float) and ragged extents.  Exit code 0 = pass.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "mgps_voxel_flatten.hpp"

template <class T>
class TiledArray  // test double: NOT the HDK class, only its call surface
{
public:
    void size(int x, int y, int z)
    {
        nx_ = x, ny_ = y, nz_ = z;
        tx_ = (x + 15) / 16, ty_ = (y + 15) / 16, tz_ = (z + 15) / 16;
        data_.assign(size_t(tx_) * ty_ * tz_ * 4096, T());
    }
    int getXRes() const { return nx_; }
    int getYRes() const { return ny_; }
    int getZRes() const { return nz_; }
    T getValue(int x, int y, int z) const { return data_[at(x, y, z)]; }
    void setValue(int x, int y, int z, T v) { data_[at(x, y, z)] = v; }

private:
    size_t at(int x, int y, int z) const
    {
        const size_t tile = (size_t(z / 16) * ty_ + y / 16) * tx_ + x / 16;
        return tile * 4096 + (size_t(z % 16) * 16 + y % 16) * 16 + x % 16;
    }
    int nx_ = 0, ny_ = 0, nz_ = 0, tx_ = 0, ty_ = 0, tz_ = 0;
    std::vector<T> data_;
};

int main()
{
    const int shapes[3][3] = {{64, 64, 64}, {37, 21, 50}, {5, 3, 2}};
    for (const auto &s : shapes) {
        TiledArray<int> labels;
        TiledArray<double> field;
        labels.size(s[0], s[1], s[2]);
        field.size(s[0], s[1], s[2]);
        for (int k = 0; k < s[2]; ++k)
            for (int j = 0; j < s[1]; ++j)
                for (int i = 0; i < s[0]; ++i) {
                    labels.setValue(i, j, k, (i * 7 + j * 3 + k) % 4);
                    field.setValue(i, j, k, 0.25 * i - 1.5 * j + 3.0 * k);  // exactly representable in float
                }
        std::vector<uint8_t> flatLabels;
        std::vector<float> flatField;
        mgps::flattenGrid(flatLabels, labels);
        mgps::flattenGrid(flatField, field);
        for (int k = 0; k < s[2]; ++k)
            for (int j = 0; j < s[1]; ++j)
                for (int i = 0; i < s[0]; ++i) {
                    const size_t c = (size_t(k) * s[1] + j) * s[0] + i;  // the ABI's order
                    if (flatLabels[c] != uint8_t((i * 7 + j * 3 + k) % 4) || flatField[c] != float(0.25 * i - 1.5 * j + 3.0 * k)) {
                        std::printf("flatten mismatch at %d %d %d\n", i, j, k);
                        return 1;
                    }
                }
        TiledArray<int> labels2;
        TiledArray<double> field2;
        labels2.size(s[0], s[1], s[2]);
        field2.size(s[0], s[1], s[2]);
        if (!mgps::unflattenGrid(labels2, flatLabels) || !mgps::unflattenGrid(field2, flatField)) return 2;
        for (int k = 0; k < s[2]; ++k)
            for (int j = 0; j < s[1]; ++j)
                for (int i = 0; i < s[0]; ++i)
                    if (labels2.getValue(i, j, k) != labels.getValue(i, j, k) || field2.getValue(i, j, k) != field.getValue(i, j, k)) {
                        std::printf("round trip mismatch at %d %d %d\n", i, j, k);
                        return 3;
                    }
        TiledArray<double> wrong;
        wrong.size(s[0] + 1, s[1], s[2]);
        if (mgps::unflattenGrid(wrong, flatField)) return 4;  // extents must match
    }
    std::printf("flatten round trip ok\n");
    return 0;
}
