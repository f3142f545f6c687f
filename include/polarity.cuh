// Forces for polarization, single-point-layer epithelia, and migration.
//
// API parity with ya||a `include/polarity.cuh:1-164`: Polarity, pol_to_float3,
// pt_to_pol (2 overloads), pol_dot_product, unidirectional_ /
// bidirectional_polarization_force, bending_force, apical_constriction_force,
// orthonormal, migration_force -- same template parameters (member pointers
// selecting which fields hold the polarity), same arithmetic.  These are pure
// `__device__ __host__` functions called from inside model functors; they do
// not touch memory, so there is nothing MI355X-specific to do except keep them
// inlinable.  A polarity is a unit vector p given by 0 <= theta < pi and
// -pi <= phi <= pi.
#pragma once

#include <math.h>

#include <type_traits>

#ifndef YA_ORACLE
#include "utils.cuh"  // dot_product
#endif


struct Polarity {
    float theta, phi;
};

// Unit vector of a polarity (polarity.cuh:14-22).
template<typename Pt, float Pt::*theta = &Pt::theta, float Pt::*phi = &Pt::phi>
__device__ __host__ float3 pol_to_float3(Pt p)
{
    const float sin_theta = sinf(p.*theta);
    float3 vec;
    vec.x = sin_theta * cosf(p.*phi);
    vec.y = sin_theta * sinf(p.*phi);
    vec.z = cosf(p.*theta);
    return vec;
}

// Direction of r as a polarity (polarity.cuh:24-40).
template<typename Pt>
__device__ __host__ Polarity pt_to_pol(Pt r, float dist)
{
    Polarity pol{acosf(r.z / dist), atan2f(r.y, r.x)};
    return pol;
}

template<typename Pt>
__device__ __host__ Polarity pt_to_pol(Pt r)
{
    const float dist = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z);
    return pt_to_pol(r, dist);
}

// p_a . p in spherical coordinates (polarity.cuh:42-47).
template<typename Pt, float Pt::*theta = &Pt::theta, float Pt::*phi = &Pt::phi>
__device__ __host__ float pol_dot_product(Pt a, Polarity p)
{
    return sinf(a.*theta) * sinf(p.theta) * cosf(a.*phi - p.phi) +
           cosf(a.*theta) * cosf(p.theta);
}

// Aligning force from the potential U = - sum(p_i . p_j): all polarities end up
// pointing the same way (polarity.cuh:49-62).  Only the two polarity fields of
// the result are non-zero.
template<typename Pt, float Pt::*theta = &Pt::theta, float Pt::*phi = &Pt::phi>
__device__ __host__ Pt unidirectional_polarization_force(Pt Xi, Polarity p)
{
    Pt dF{0};
    const float sin_i = sinf(Xi.*theta);
    dF.*theta = cosf(Xi.*theta) * sinf(p.theta) * cosf(Xi.*phi - p.phi) - sin_i * cosf(p.theta);
    if (fabs(sin_i) > 1e-10) dF.*phi = -sinf(p.theta) * sinf(Xi.*phi - p.phi) / sin_i;
    return dF;
}

// Aligning force from U_Pol = - sum(p_i . p_j)^2 / 2: polarities end up
// parallel or anti-parallel (polarity.cuh:64-72).
template<typename Pt, float Pt::*theta = &Pt::theta, float Pt::*phi = &Pt::phi>
__device__ __host__ Pt bidirectional_polarization_force(Pt Xi, Polarity p)
{
    const float prod = pol_dot_product<Pt, theta, phi>(Xi, p);
    return prod * unidirectional_polarization_force<Pt, theta, phi>(Xi, p);
}

// The spelling the reference's own tests/test_polarity.cu:25,43,65-71 and examples/polarization.cu:29 still
// use (they predate `Polarity`): the partner given as a whole point, of which only the polarity
// fields count.  Pinned by that test's known answer: bidirectional_polarization_force(i, j).theta = 0.126,
// .phi = 0.215 for the two Po_cells at test_polarity.cu:22-23.
namespace ya {
template<typename Pb>
struct Whole_point {  // any point type but Polarity itself, which the functions above take
    static constexpr bool value = !std::is_same<Pb, Polarity>::value;
};
template<typename Pb>
__device__ __host__ Polarity polarity_of(const Pb& b)
{
    return Polarity{b.theta, b.phi};
}
}  // namespace ya

template<typename Pa, typename Pb>
__device__ __host__ typename std::enable_if<ya::Whole_point<Pb>::value, float>::type pol_dot_product(Pa a, Pb b)
{
    return pol_dot_product(a, ya::polarity_of(b));
}

template<typename Pt, typename Pb>
__device__ __host__ typename std::enable_if<ya::Whole_point<Pb>::value, Pt>::type unidirectional_polarization_force(
    Pt Xi, Pb Xj)
{
    return unidirectional_polarization_force(Xi, ya::polarity_of(Xj));
}

template<typename Pt, typename Pb>
__device__ __host__ typename std::enable_if<ya::Whole_point<Pb>::value, Pt>::type bidirectional_polarization_force(
    Pt Xi, Pb Xj)
{
    return bidirectional_polarization_force(Xi, ya::polarity_of(Xj));
}

namespace ya {
// -(c/dist) p + c^2/dist^2 r: one polarity's share of the position part of the
// bending-type forces below.
template<typename Pt>
__device__ __host__ inline float3 bending_share(float3 p, Pt r, float dist, float c)
{
    float3 s;
    s.x = -c / dist * p.x + powf(c, 2) / powf(dist, 2) * r.x;
    s.y = -c / dist * p.y + powf(c, 2) / powf(dist, 2) * r.y;
    s.z = -c / dist * p.z + powf(c, 2) / powf(dist, 2) * r.z;
    return s;
}
}  // namespace ya

// Resistance to bending from U_Epi = sum(p_i . r_ij/r)^2 / 2 (polarity.cuh:75-97).
// r = Xi - Xj carries the difference of the polarities in its polarity fields,
// which is how the neighbour's polarity is recovered.
template<typename Pt, float Pt::*theta = &Pt::theta, float Pt::*phi = &Pt::phi>
__device__ __host__ Pt bending_force(Pt Xi, Pt r, float dist)
{
    const float3 pi = pol_to_float3<Pt, theta, phi>(Xi);
    const float prodi = (pi.x * r.x + pi.y * r.y + pi.z * r.z) / dist;
    const Polarity r_hat = pt_to_pol(r, dist);
    Pt dF = -prodi * unidirectional_polarization_force<Pt, theta, phi>(Xi, r_hat);
    const float3 from_i = ya::bending_share(pi, r, dist, prodi);
    dF.x = from_i.x;
    dF.y = from_i.y;
    dF.z = from_i.z;

    // Contribution from (p_j . r_ji/r)^2/2
    const Polarity Xj{Xi.*theta - r.*theta, Xi.*phi - r.*phi};
    const float3 pj = pol_to_float3(Xj);
    const float prodj = (pj.x * r.x + pj.y * r.y + pj.z * r.z) / dist;
    const float3 from_j = ya::bending_share(pj, r, dist, prodj);
    dF.x += from_j.x;
    dF.y += from_j.y;
    dF.z += from_j.z;
    return dF;
}

// Bending force whose preferred angle between p_i and r_ij is pref_angle
// instead of 90 degrees (wedge-shaped cells; pi/2 gives a flat epithelium)
// (polarity.cuh:99-124).
template<typename Pt>
__device__ __host__ Pt apical_constriction_force(Pt Xi, Pt r, float dist, float pref_angle)
{
    const float3 pi = pol_to_float3(Xi);
    const float prodi = (pi.x * r.x + pi.y * r.y + pi.z * r.z) / dist + cosf(pref_angle);
    const Polarity r_hat = pt_to_pol(r, dist);
    Pt dF = -prodi * unidirectional_polarization_force(Xi, r_hat);
    const float3 from_i = ya::bending_share(pi, r, dist, prodi);
    dF.x = from_i.x;
    dF.y = from_i.y;
    dF.z = from_i.z;

    const Polarity Xj{Xi.theta - r.theta, Xi.phi - r.phi};
    const float3 pj = pol_to_float3(Xj);
    const float prodj = (pj.x * r.x + pj.y * r.y + pj.z * r.z) / dist - cosf(pref_angle);
    const float3 from_j = ya::bending_share(pj, r, dist, prodj);
    dF.x += from_j.x;
    dF.y += from_j.y;
    dF.z += from_j.z;
    return dF;
}

// Unit vector orthogonal to p in the plane of r and p (polarity.cuh:128-134).
template<typename Pt>
__device__ __host__ float3 orthonormal(Pt r, float3 p)
{
    const float3 r3{r.x, r.y, r.z};
    const float3 normal = r3 - dot_product(r3, p) * p;
    return normal / sqrt(dot_product(normal, normal));
}

// Mono-polar migration force, after
// https://doi.org/10.1016/B978-0-12-405926-9.00016-2 (polarity.cuh:136-164).
template<typename Pt, float Pt::*theta = &Pt::theta, float Pt::*phi = &Pt::phi>
__device__ __host__ Pt migration_force(Pt Xi, Pt r, float dist)
{
    Pt dF{0};

    // Pulling around j
    const Polarity r_hat = pt_to_pol(r, dist);
    if ((Xi.phi != 0) or (Xi.theta != 0)) {
        if (pol_dot_product<Pt, theta, phi>(Xi, r_hat) <= -0.15) {
            const float3 pi = pol_to_float3<Pt, theta, phi>(Xi);
            const float3 pi_T = orthonormal(r, pi);
            dF.x = 0.6 * pi.x + 0.8 * pi_T.x;
            dF.y = 0.6 * pi.y + 0.8 * pi_T.y;
            dF.z = 0.6 * pi.z + 0.8 * pi_T.z;
        }
    }

    // Getting pushed aside by j
    const Polarity Xj{Xi.*theta - r.*theta, Xi.*phi - r.*phi};
    if ((Xj.phi > 1e-10) or (Xj.theta > 1e-10)) {
        if (pol_dot_product(Xj, r_hat) >= 0.15) {
            const float3 pj = pol_to_float3(Xj);
            const float3 pj_T = orthonormal(-r, pj);
            dF.x -= 0.6 * pj.x + 0.8 * pj_T.x;
            dF.y -= 0.6 * pj.y + 0.8 * pj_T.y;
            dF.z -= 0.6 * pj.z + 0.8 * pj_T.z;
        }
    }

    return dF;
}
