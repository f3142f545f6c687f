// The update kernels that fold the centre-of-mass reduction themselves (round 5) and leave the next stage's
// right-hand side zeroed for the generic forces: steps WITH and WITHOUT generic forces in turn, cells appended
// between steps, Grid_solver and Tile_solver -- bit for bit the states the plain pipeline gives (ya_reduce_mean's
// own second launch, a memset in front of every generic force).
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/solvers.cuh"

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

__device__ float3 clipped_spring(float3 Xi, float3 r, float dist, int i, int j)
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j || dist >= 1.f) return dF;
    return r * (0.5f - dist) / dist;
}

__global__ void push(int n, float3* d_dX)   // a generic force that ADDS to what it finds (solvers.cuh:235)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && i % 7 == 0) d_dX[i].x += 0.25f;
}

__global__ void append(int n, int extra, float3* d_X, float3* d_old_v, int* d_n)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) *d_n = n + extra;
    if (k >= extra) return;
    d_X[n + k] = float3{d_X[k].x + 0.3f, d_X[k].y, d_X[k].z - 0.2f};
    d_old_v[n + k] = d_old_v[k];
}

template<template<typename> class Solver>
struct Cells : public Solution<float3, Solver> {
    using Solution<float3, Solver>::Solution;
    void folding(bool on) { this->fold_in_update = on; }
};

template<template<typename> class Solver, typename... Args>
std::vector<float3> run(bool fold, int n0, Args... args)
{
    Cells<Solver> cells{2 * n0, args...};
    *cells.h_n = n0;
    random_sphere(0.6f, cells, 0, 31);
    cells.folding(fold);
    auto gen = [](const int n, const float3* __restrict__ d_X, float3* d_dX) { push<<<(n + 255) / 256, 256>>>(n, d_dX); };
    for (int step = 0; step < 9; step++) {
        if (step % 3 == 1)
            cells.template take_step<clipped_spring>(0.05f);        // no generic forces: the sorted-space step
        else
            cells.template take_step<clipped_spring>(0.05f, gen);   // with: zeroed rows matter
        if (step == 4) {   // the system grows between steps: rows beyond the old n were never zeroed
            const int n = cells.get_d_n();
            append<<<(n0 / 4 + 255) / 256, 256>>>(n, n0 / 4, cells.d_X, cells.d_old_v, cells.d_n);
        }
    }
    cells.copy_to_host();
    return std::vector<float3>(cells.h_X, cells.h_X + *cells.h_n);
}

int main()
{
    {
        const auto plain = run<Grid_solver>(false, 4000, 40, 1.f), folded = run<Grid_solver>(true, 4000, 40, 1.f);
        EXPECT(plain.size() == 5000 && folded.size() == 5000);
        EXPECT(memcmp(plain.data(), folded.data(), plain.size() * sizeof(float3)) == 0);
    }
    {
        const auto plain = run<Tile_solver>(false, 600), folded = run<Tile_solver>(true, 600);
        EXPECT(plain.size() == 750 && folded.size() == 750);
        EXPECT(memcmp(plain.data(), folded.data(), plain.size() * sizeof(float3)) == 0);
    }
    printf(failures ? "%d FAILURES\n" : "ALL FOLDING TESTS PASSED\n", failures);
    return failures != 0;
}
