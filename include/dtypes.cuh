// Point types for the MI355X spheroid-cell engine.
//
// API parity with ya||a `include/dtypes.cuh:1-217`: `Is_vector<Pt>`,
// `MAKE_PT(Name, fields...)`, `Po_cell`, and the vector-space operators
// + += - -= unary- * *= / /= on every point type.  Implementation notes:
//
//  * Under HIP `float3` / `float4` are `HIP_vector_type`s that already carry
//    component-wise + - * += -= *= ==, so (unlike the reference, lines 10-47)
//    nothing is re-declared for them except division: the reference divides
//    a point by a scalar as `a *= 1. / b` (dtypes.cuh:202-208, double
//    reciprocal rounded to float, then one multiply per component).  HIP's
//    built-in `/` is a true per-component division, ≤1 ulp different and three
//    divisions instead of one, so non-template overloads for float3/float4
//    restore the reference's arithmetic (a non-template beats HIP's template
//    `operator/(const HIP_vector_type<T,n>&, U)` in overload resolution).
//
//  * A MAKE_PT type is `struct { float x, y, z, fields...; }`.  All members
//    are float, so the operators treat a point as sizeof(Pt)/4 packed floats;
//    loops are fully unrolled by the compiler and live in VGPRs.  No 50-arity
//    MAP macro is needed.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

template<typename Pt>
struct Is_vector : public std::false_type {};

template<>
struct Is_vector<float3> : public std::true_type {};
template<>
struct Is_vector<float4> : public std::true_type {};

namespace ya {
// True for the points whose operators this header must supply (MAKE_PT types).
template<typename Pt>
struct Is_made_pt : public std::false_type {};

template<typename Pt>
struct N_floats {
    static_assert(sizeof(Pt) % sizeof(float) == 0, "points are packed floats");
    static constexpr int value = sizeof(Pt) / sizeof(float);
};

template<typename Pt>
__host__ __device__ __forceinline__ float& field(Pt& a, int k)
{
    return reinterpret_cast<float*>(&a)[k];
}
template<typename Pt>
__host__ __device__ __forceinline__ const float& field(const Pt& a, int k)
{
    return reinterpret_cast<const float*>(&a)[k];
}

template<typename Pt>
__host__ __device__ __forceinline__ void add_to(Pt& a, const Pt& b)
{
#pragma unroll
    for (int k = 0; k < N_floats<Pt>::value; k++) field(a, k) += field(b, k);
}
template<typename Pt>
__host__ __device__ __forceinline__ void scale(Pt& a, const float b)
{
#pragma unroll
    for (int k = 0; k < N_floats<Pt>::value; k++) field(a, k) *= b;
}
template<typename Pt>
__host__ __device__ __forceinline__ Pt zero()
{
    Pt z;
#pragma unroll
    for (int k = 0; k < N_floats<Pt>::value; k++) field(z, k) = 0.f;
    return z;
}

// 1 / b rounded to nearest: the factor in the reference's `Pt / b` (dtypes.cuh:202-217
// multiplies by 1. / b).  On the device this is the compiler's own expansion of
// 1.0f / b (reciprocal estimate, two refinements, final residual correction)
// without the rescaling and special-case fix-up, which only matter outside
// 2^-64 <= |b| <= 2^64; arguments out there take the library path.  Verified against
// 1.0f / b for EVERY binary32 argument by tests/test_parity_gpu.py
// (ya::check_reciprocal_all): seven instructions instead of fourteen per pair.
__host__ __device__ __forceinline__ float reciprocal(const float b)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(YA_ARITH_FAST)
    // fast-arithmetic tier (see ya::exact_sqrt in solvers.cuh): the bare v_rcp_f32, <= 1 ulp
    return __builtin_amdgcn_rcpf(b);
#elif defined(__HIP_DEVICE_COMPILE__)
    // The common path runs unconditionally; out-of-range arguments replace its result
    // afterwards (one skipped branch per call instead of a two-sided one).
    const float a = __builtin_fabsf(b);
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float e1 = __builtin_fmaf(-b, r1, 1.0f);
    const float q1 = __builtin_fmaf(e1, r1, r1);
    const float e2 = __builtin_fmaf(-b, q1, 1.0f);
    float out = __builtin_fmaf(e2, r1, q1);
    if (__builtin_expect(!(a >= 0x1p-64f && a <= 0x1p+64f), 0)) out = 1.0f / b;
    return out;
#else
    return 1. / b;
#endif
}
}  // namespace ya

// float3 / float4 division with the reference's reciprocal-multiply
// arithmetic (dtypes.cuh:202-217).
__host__ __device__ __forceinline__ float3 operator/(const float3& a, const float b)
{
    const float inv = ya::reciprocal(b);
    return float3{a.x * inv, a.y * inv, a.z * inv};
}
__host__ __device__ __forceinline__ float3 operator/(const float3& a, const double b)
{
    return a / static_cast<float>(b);
}
__host__ __device__ __forceinline__ float3 operator/(const float3& a, const int b)
{
    return a / static_cast<float>(b);
}
__host__ __device__ __forceinline__ float4 operator/(const float4& a, const float b)
{
    const float inv = ya::reciprocal(b);
    return float4{a.x * inv, a.y * inv, a.z * inv, a.w * inv};
}
__host__ __device__ __forceinline__ float4 operator/(const float4& a, const double b)
{
    return a / static_cast<float>(b);
}
__host__ __device__ __forceinline__ float4 operator/(const float4& a, const int b)
{
    return a / static_cast<float>(b);
}

// MAKE_PT(Name, fields...): dtypes.cuh:58-75.
#define MAKE_PT(Name, ...)                                                     \
    struct Name {                                                              \
        float x, y, z, __VA_ARGS__;                                            \
        friend __host__ __device__ __forceinline__ Name operator+=(            \
            Name& a, const Name& b)                                            \
        {                                                                      \
            ya::add_to(a, b);                                                  \
            return a;                                                          \
        }                                                                      \
        friend __host__ __device__ __forceinline__ Name operator*=(            \
            Name& a, const float b)                                            \
        {                                                                      \
            ya::scale(a, b);                                                   \
            return a;                                                          \
        }                                                                      \
    };                                                                         \
    template<>                                                                 \
    struct Is_vector<Name> : public std::true_type {};                         \
    template<>                                                                 \
    struct ya::Is_made_pt<Name> : public std::true_type {}

// Polarized cell, dtypes.cuh:147
MAKE_PT(Po_cell, theta, phi);

// + -= - unary- * /= / for MAKE_PT types, built from += and *= exactly as
// dtypes.cuh:150-217 builds them (so `a - b` is `a + (-1 * b)` and `a / b` is
// `a * float(1. / b)`).
#define YA_PT_OP(ret)                               \
    template<typename Pt>                           \
    __host__ __device__ __forceinline__             \
        typename std::enable_if<ya::Is_made_pt<Pt>::value, ret>::type

YA_PT_OP(Pt) operator*(const Pt& a, const float b)
{
    Pt p = a;
    p *= b;
    return p;
}
YA_PT_OP(Pt) operator*(const float b, const Pt& a)
{
    Pt p = a;
    p *= b;
    return p;
}
YA_PT_OP(Pt) operator+(const Pt& a, const Pt& b)
{
    Pt s = a;
    s += b;
    return s;
}
YA_PT_OP(Pt) operator-=(Pt& a, const Pt& b)
{
    a += -1 * b;
    return a;
}
YA_PT_OP(Pt) operator-(const Pt& a, const Pt& b)
{
    Pt d = a;
    d -= b;
    return d;
}
YA_PT_OP(Pt) operator-(const Pt& a) { return -1 * a; }
YA_PT_OP(Pt) operator/=(Pt& a, const float b)
{
    a *= ya::reciprocal(b);
    return a;
}
YA_PT_OP(Pt) operator/(const Pt& a, const float b)
{
    Pt q = a;
    q /= b;
    return q;
}
#undef YA_PT_OP
