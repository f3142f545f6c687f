// Initial states.  API parity with ya||a `include/inits.cuh` for the functions
// on or next to the step path: random_disk (:14-31), random_sphere (:33-51),
// random_cuboid (:53-75), relu_force (:78-93), relaxed_sphere (:95-125),
// relaxed_cuboid (:127-155).  Each random_* takes an optional trailing `seed`
// (0 = seed from std::random_device, the reference's behaviour) so runs can be
// reproduced; draws use glibc rand() in the reference's order and arithmetic.
#pragma once

#include <assert.h>
#include <math.h>
#include <stdlib.h>

#include <iostream>
#include <random>

#include "dtypes.cuh"

template<typename Pt, template<typename> class Solver>
class Solution;

namespace ya {
// seed 0 = "any seed": std::random_device as in the reference (inits.cuh:19-20,38-39), or,
// when the environment variable YALLA_SEED is set to a non-zero number, that number plus the
// count of such calls so far -- every initial condition of a run is then reproducible
// (used to run the reference's own statistical tests deterministically).
inline void seed_rand(unsigned seed)
{
    if (seed == 0) {
        static unsigned calls = 0;
        const char* pinned = getenv("YALLA_SEED");
        const unsigned base = pinned ? (unsigned)strtoul(pinned, nullptr, 10) : 0u;
        if (base != 0) {
            seed = base + calls++;
        } else {
            std::random_device rd;
            seed = rd();
        }
    }
    srand(seed);
}
inline double unit_rand() { return rand() / (RAND_MAX + 1.); }
}  // namespace ya


template<typename Pt, template<typename> class Solver>
void random_disk(float dist_to_nb, Solution<Pt, Solver>& points, unsigned int n_0 = 0,
    unsigned seed = 0)
{
    assert(n_0 < *points.h_n);
    ya::seed_rand(seed);
    // Radius based on hexagonal lattice
    const double r_max = pow((*points.h_n - n_0) / 0.9069, 1. / 2) * dist_to_nb / 2;
    for (unsigned i = n_0; i < (unsigned)*points.h_n; i++) {
        const double r = r_max * pow(ya::unit_rand(), 1. / 2);
        const double phi = ya::unit_rand() * 2 * M_PI;
        points.h_X[i].x = 0;
        points.h_X[i].y = r * sin(phi);
        points.h_X[i].z = r * cos(phi);
    }
    points.copy_to_device();
}

template<typename Pt, template<typename> class Solver>
void random_sphere(float dist_to_nb, Solution<Pt, Solver>& points, unsigned int n_0 = 0,
    unsigned seed = 0)
{
    assert(n_0 < *points.h_n);
    ya::seed_rand(seed);
    // Radius based on random sphere packing
    const double r_max = pow((*points.h_n - n_0) / 0.64, 1. / 3) * dist_to_nb / 2;
    for (unsigned i = n_0; i < (unsigned)*points.h_n; i++) {
        const double r = r_max * pow(ya::unit_rand(), 1. / 3);
        const double theta = acos(2. * rand() / (RAND_MAX + 1.) - 1);
        const double phi = ya::unit_rand() * 2 * M_PI;
        points.h_X[i].x = r * sin(theta) * cos(phi);
        points.h_X[i].y = r * sin(theta) * sin(phi);
        points.h_X[i].z = r * cos(theta);
    }
    points.copy_to_device();
}

template<typename Pt, template<typename> class Solver>
void random_cuboid(float dist_to_nb, float3 minimum, float3 maximum,
    Solution<Pt, Solver>& points, unsigned int n_0 = 0, unsigned seed = 0)
{
    assert(n_0 < *points.h_n);

    const float3 dimension = maximum - minimum;
    const auto cube_volume = dimension.x * dimension.y * dimension.z;
    const auto sphere_volume = 4. / 3 * M_PI * pow(dist_to_nb / 2, 3);
    const auto n = cube_volume / sphere_volume * 0.64;  // Sphere packing

    assert(n_0 + n < *points.h_n);
    *points.h_n = n_0 + n;

    ya::seed_rand(seed);
    for (unsigned i = n_0; i < (unsigned)*points.h_n; i++) {
        points.h_X[i].x = minimum.x + dimension.x * ya::unit_rand();
        points.h_X[i].y = minimum.y + dimension.y * ya::unit_rand();
        points.h_X[i].z = minimum.z + dimension.z * ya::unit_rand();
    }
    points.copy_to_device();
}


// Repulsion below 0.8, attraction up to 1 (inits.cuh:78-93).
template<typename Pt>
__device__ Pt relu_force(Pt Xi, Pt r, float dist, int i, int j)
{
    Pt dF{0};

    if (i == j) return dF;

    if (dist > 1.f) return dF;

    auto F = fmaxf(0.8f - dist, 0) * 2.f - fmaxf(dist - 0.8f, 0);
    dF.x = r.x * F / dist;
    dF.y = r.y * F / dist;
    dF.z = r.z * F / dist;

    return dF;
}

namespace ya {
// While one of these lives, a solver that has cooperative force kernels uses them (several lanes
// per cell: Tile_computer::lanes_per_cell = 64, Grid_computer::force_variant = 3; bit-identical
// results).  For steps whose functor is known to keep no per-cell state, like relu_force here.
template<typename S>
class Cooperative_kernels {
    S& solver;
    int saved_lanes = 0, saved_variant = 0;
    template<typename T>
    static auto lanes(T& s, int) -> decltype(s.lanes_per_cell)* { return &s.lanes_per_cell; }
    template<typename T>
    static int* lanes(T&, long) { return nullptr; }
    template<typename T>
    static auto variant(T& s, int) -> decltype(s.force_variant)* { return &s.force_variant; }
    template<typename T>
    static int* variant(T&, long) { return nullptr; }

public:
    explicit Cooperative_kernels(S& s) : solver{s}
    {
        if (int* l = lanes(solver, 0)) {
            saved_lanes = *l;
            if (*l <= 1) *l = 64;
        }
        if (int* v = variant(solver, 0)) {
            saved_variant = *v;
            if (*v == 2 || *v < 0) *v = 3;
        }
    }
    ~Cooperative_kernels()
    {
        if (int* l = lanes(solver, 0)) *l = saved_lanes;
        if (int* v = variant(solver, 0)) *v = saved_variant;
    }
};

template<typename Pt, template<typename> class Solver>
void relax_and_rescale(
    double scale, Solution<Pt, Solver>& points, int steps, int warn_above)
{
    if (*points.h_n > warn_above)
        std::cout << "Warning: The system is quite large, it may "
                  << "not be completely relaxed." << std::endl;
    {
        // 500-3000 steps of a few hundred to a few thousand cells: model set-up time is these
        Cooperative_kernels<Solution<Pt, Solver>> several_lanes_per_cell{points};
        for (int i = 0; i < steps; i++) points.template take_step<relu_force>(0.1f);
    }
    points.copy_to_host();
    for (int i = 0; i < *points.h_n; i++) {
        points.h_X[i].x *= scale;
        points.h_X[i].y *= scale;
        points.h_X[i].z *= scale;
    }
    points.copy_to_device();
}
}  // namespace ya

// Random sphere at spacing 0.6 relaxed with relu_force (equilibrium 0.8), then
// rescaled to dist_to_nb (inits.cuh:95-125).
template<typename Pt, template<typename> class Solver>
void relaxed_sphere(float dist_to_nb, Solution<Pt, Solver>& points, unsigned int n_0 = 0,
    unsigned seed = 0)
{
    random_sphere(0.6, points, n_0, seed);
    const int n = *points.h_n;
    const int steps = n <= 100 ? 500 : (n <= 1000 ? 1000 : (n <= 6000 ? 2000 : 3000));
    ya::relax_and_rescale(dist_to_nb / 0.8, points, steps, 10000);
}

// Same in a box; the box is shrunk by the final scale first (inits.cuh:127-155).
template<typename Pt, template<typename> class Solver>
void relaxed_cuboid(float dist_to_nb, float3 minimum, float3 maximum,
    Solution<Pt, Solver>& points, unsigned int n_0 = 0, unsigned seed = 0)
{
    const auto scale = dist_to_nb / 0.8;
    random_cuboid(0.8, minimum / scale, maximum / scale, points, n_0, seed);
    const int n = *points.h_n;
    const int steps = n <= 3000 ? 1000 : (n <= 12000 ? 2000 : 3000);
    ya::relax_and_rescale(scale, points, steps, 15000);
}


// Hexagonal patch in the z = 0 plane, filled ring by ring around a centre cell:
// ring i holds 6 corner cells at distance i * dist_to_nb (first one on the +y
// axis, going counter-clockwise) with i - 1 evenly spaced cells on each edge
// between consecutive corners (inits.cuh:158-215).  Stops when h_n cells exist.
template<typename Pt, template<typename> class Solver>
void regular_hexagon(float dist_to_nb, Solution<Pt, Solver>& points, unsigned int n_0 = 0)
{
    assert(n_0 < *points.h_n);

    const unsigned n = *points.h_n;
    unsigned placed = n_0;
    auto place = [&](float x, float y) {
        points.h_X[placed].x = x;
        points.h_X[placed].y = y;
        points.h_X[placed].z = 0.f;
        placed++;
        return placed == n;
    };
    const float beta = M_PI / 3.f;
    bool full = place(0.f, 0.f);
    for (int ring = 1; !full; ring++) {
        for (int corner = 0; corner < 6 && !full; corner++) {
            const float angle = beta * corner;
            const float3 p{-dist_to_nb * ring * sinf(angle), dist_to_nb * ring * cosf(angle), 0.f};
            full = place(p.x, p.y);
            const int n_between = ring - 1;
            if (full || n_between < 1) continue;
            const float next_angle = beta * (corner + 1);
            const float3 q{
                -dist_to_nb * ring * sinf(next_angle), dist_to_nb * ring * cosf(next_angle), 0.f};
            float3 v = q - p;
            const auto modulus = sqrt(pow(v.x, 2) + pow(v.y, 2));
            v = v * (1.f / modulus);
            for (int k = 1; k <= n_between && !full; k++) {
                const float3 u = v * modulus * (float(k) / float(n_between + 1));
                full = place(p.x + u.x, p.y + u.y);
            }
        }
    }
    points.copy_to_device();
}

// Rows of nx cells in the z = 0 plane on a triangular lattice: row spacing
// sqrt(3)/2 * dist_to_nb, odd rows shifted by half a spacing (inits.cuh:217-247).
template<typename Pt, template<typename> class Solver>
void regular_rectangle(
    float dist_to_nb, int nx, Solution<Pt, Solver>& points, unsigned int n_0 = 0)
{
    assert(n_0 < *points.h_n);

    const unsigned n = *points.h_n;
    unsigned placed = n_0;
    for (int row = 0; placed < n; row++) {
        const float y = row * sqrt(pow(dist_to_nb, 2) - pow(dist_to_nb / 2.f, 2));
        const float shift = row % 2 != 0 ? dist_to_nb / 2.f : 0.0f;
        for (int col = 0; col < nx && placed < n; col++) {
            points.h_X[placed].x = shift + col * dist_to_nb;
            points.h_X[placed].y = y;
            points.h_X[placed].z = 0.0f;
            placed++;
        }
    }
    points.copy_to_device();
}
