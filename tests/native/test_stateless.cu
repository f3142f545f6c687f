// YA_STATELESS (include/solvers.cuh): ONE line next to a model's pairwise functor lets the
// solvers share a cell among several lanes when the system is too small to fill the chip
// (ya::grid_force_coop / ya::tile_force_coop) -- same results bit for bit, several times faster
// steps.  Two functors with the same body, one declared stateless, from the same initial state.
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/solvers.cuh"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static int failures = 0;
#define EXPECT(cond)                                                  \
    do {                                                              \
        if (!(cond)) {                                                \
            printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
            failures++;                                               \
        }                                                             \
    } while (0)

// a two-type adhesion model in the spirit of examples/sorting.cu (types split at id 5000)
#define ADHESION_BODY                                                        \
    float3 dF{0.f, 0.f, 0.f};                                                \
    if (i == j) return dF;                                                   \
    if (dist > 1.f) return dF;                                               \
    const float strength = (1 + 2 * (j < 5000)) * (1 + 2 * (i < 5000));      \
    const float F = 2 * (0.5f - dist) * (1.f - dist) + (1.f - dist) * (1.f - dist); \
    dF = strength * r * F / dist;                                            \
    return dF;
__device__ float3 plain_adhesion(float3 Xi, float3 r, float dist, int i, int j) { ADHESION_BODY }
__device__ float3 declared_adhesion(float3 Xi, float3 r, float dist, int i, int j) { ADHESION_BODY }
YA_STATELESS(float3, declared_adhesion)   // <- the one line

template<template<typename> class Solver, Pairwise_interaction<float3> force, typename... Args>
double run(int n, int steps, float dt, std::vector<float3>& out, Args... args)
{
    Solution<float3, Solver> cells{n, args...};
    random_sphere(0.5f, cells, 0, 5);
    for (int s = 0; s < 5; s++) cells.template take_step<force>(dt);
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; s++) cells.template take_step<force>(dt);
    (void)hipDeviceSynchronize();
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    cells.copy_to_host();
    out.assign(cells.h_X, cells.h_X + n);
    return seconds / steps;
}

int main()
{
    static_assert(!ya::stateless_pair<float3, plain_adhesion, friction_w_neighbour<float3>>(), "");
    static_assert(ya::stateless_pair<float3, declared_adhesion, friction_w_neighbour<float3>>(), "");
    static_assert(ya::stateless_pair<float3, declared_adhesion, friction_on_background<float3>>(), "");
    {  // Grid_solver, 10 000 cells (BASELINE config 2's size)
        std::vector<float3> a, b;
        const double plain = run<Grid_solver, plain_adhesion>(10000, 200, 0.01f, a, 50, 1.f);
        const double declared = run<Grid_solver, declared_adhesion>(10000, 200, 0.01f, b, 50, 1.f);
        int different = 0;
        for (int i = 0; i < 10000; i++) different += memcmp(&a[i], &b[i], sizeof(float3)) != 0;
        EXPECT(different == 0);
        printf("Grid_solver, 10000 cells: %.1f us per step, declared stateless %.1f us (%.2f x), %d cells differ\n",
            plain * 1e6, declared * 1e6, plain / declared, different);
        EXPECT(declared * 1.4 < plain);
        printf("cell-updates/s with the one line: %.3g\n", 10000 / declared);
        EXPECT(10000 / declared >= 7e7);  // (host-launched C++ loop incl. 5 relaxing steps: 8.3e7; bench.py at dt 0.05: 1.17e8)
    }
    {  // Tile_solver, 800 cells (examples/springs.cu's size)
        std::vector<float3> a, b;
        const double plain = run<Tile_solver, plain_adhesion>(800, 100, 0.01f, a);
        const double declared = run<Tile_solver, declared_adhesion>(800, 100, 0.01f, b);
        int different = 0;
        for (int i = 0; i < 800; i++) different += memcmp(&a[i], &b[i], sizeof(float3)) != 0;
        EXPECT(different == 0);
        printf("Tile_solver, 800 cells: %.1f us per step, declared stateless %.1f us (%.2f x), %d cells differ\n",
            plain * 1e6, declared * 1e6, plain / declared, different);
        EXPECT(declared * 2 < plain);
    }
    if (failures == 0) printf("ALL STATELESS TESTS PASSED\n");
    return failures != 0;
}
