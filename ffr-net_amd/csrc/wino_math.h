// Vector types and the F(4x4,3x3) transform helpers shared by the fused Winograd kernels (wino_fused.hip, wino_mixed.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ffr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// v = B^T d for any vector width, 12 operations (shared sub-expressions of the F(4x4,3x3) input transform)
template <typename V>
__device__ __forceinline__ void bt6t(const V d[6], V v[6]) {
    const V p = d[4] - 4.f * d[2], q = d[3] - 4.f * d[1];
    const V t0 = d[4] - d[2], t1 = d[3] - d[1];
    v[0] = 4.f * d[0] + (d[4] - 5.f * d[2]);
    v[1] = p + q;
    v[2] = p - q;
    v[3] = t0 + 2.f * t1;
    v[4] = t0 - 2.f * t1;
    v[5] = 4.f * d[1] + (d[5] - 5.f * d[3]);
}

// v = B^T d (vector form, as in winograd.hip)
__device__ __forceinline__ void bt6v(const f32x4 d[6], f32x4 v[6]) {
    v[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    v[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    v[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    v[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    v[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    v[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// y = A^T m on a channel pair (packed fp32)
__device__ __forceinline__ void at6p(const f32x2 m[6], f32x2 y[4]) {
    const f32x2 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

// y = A^T m on four channels
__device__ __forceinline__ void at6q(const f32x4 m[6], f32x4 y[4]) {
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

// y = A^T m (scalar form)
__device__ __forceinline__ void at6s(const float m[6], float y[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

}  // namespace ffr
