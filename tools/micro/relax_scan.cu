// Scan seeds for a relaxation that misbehaves (cells leaving the grid, non-finite
// positions): the sequence of the reference's tests/test_inits.cu, seeded.
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/solvers.cuh"
#include <cmath>
#include <cstdio>
#include <cstdlib>

template<typename S>
float extent(S& points)
{
    float m = 0;
    for (int i = 0; i < *points.h_n; i++) {
        const float v = fmaxf(fabsf(points.h_X[i].x), fmaxf(fabsf(points.h_X[i].y), fabsf(points.h_X[i].z)));
        if (!(v <= 1e30f)) return INFINITY;
        m = fmaxf(m, v);
    }
    return m;
}

int main(int argc, char** argv)
{
    const int first = argc > 1 ? atoi(argv[1]) : 1, last = argc > 2 ? atoi(argv[2]) : 40;
    int bad = 0;
    for (int seed = first; seed <= last; seed++) {
        Solution<float3, Grid_solver> points{5000};
        if (getenv("FORCE_VARIANT")) points.force_variant = atoi(getenv("FORCE_VARIANT"));
        if (getenv("SORTED_PIPELINE")) points.sorted_pipeline = atoi(getenv("SORTED_PIPELINE"));
        relaxed_sphere(0.8, points, 0, seed);
        const float e1 = extent(points);
        const int n1 = *points.h_n;
        relaxed_cuboid(0.8, float3{0}, float3{9, 9, 9}, points, 0, seed + 1000);
        const float e2 = extent(points);
        const int n2 = *points.h_n;
        relaxed_cuboid(0.4, float3{0}, float3{4, 4, 4}, points, 0, seed + 2000);
        const float e3 = extent(points);
        const bool ok = e1 < 12 && e2 < 12 && e3 < 8;
        if (!ok) bad++;
        printf("seed %d: n %d extent %.3f | n %d extent %.3f | n %d extent %.3f %s\n", seed, n1, e1, n2, e2,
            *points.h_n, e3, ok ? "" : "  <-- BAD");
        fflush(stdout);
    }
    printf("%d bad of %d\n", bad, last - first + 1);
    return 0;
}
