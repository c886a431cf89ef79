// HDK mock (tests/hdk_mock/README.md): declarations only.
#pragma once
#include <cstdint>
typedef float fpreal32;
typedef double fpreal64;
typedef double fpreal;
typedef int64_t exint;
struct UT_Vector3 {
    float v[3];
    float &operator()(int i) { return v[i]; }
    float maxComponent() const;
};
struct UT_Vector3I {
    int v[3];
    int operator[](int i) const { return v[i]; }
};
enum UT_ErrorSeverity { UT_ERROR_NONE, UT_ERROR_MESSAGE, UT_ERROR_WARNING, UT_ERROR_ABORT };
template <class T>
class UT_VoxelArray
{
public:
    int getXRes() const;
    int getYRes() const;
    int getZRes() const;
    T getValue(int x, int y, int z) const;
    void setValue(int x, int y, int z, T value);
    void size(int xres, int yres, int zres);
    bool isConstant(T *value = nullptr) const;
};
typedef UT_VoxelArray<fpreal32> UT_VoxelArrayF;
