// Solution_n<Pt, n_max, Solver> (include/solvers.cuh): the three-parameter spelling of older model files
// and of north_star -- same object as Solution<Pt, Solver>{n_max, ...}, same steps bit for bit.
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/solvers.cuh"

#include <cstdio>
#include <cstring>

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

template<typename A, typename B>
void same_steps(A& a, B& b)
{
    random_sphere(0.6f, a, 0, 11);
    random_sphere(0.6f, b, 0, 11);
    for (int k = 0; k < 5; k++) {
        a.template take_step<clipped_spring>(0.05f);
        b.template take_step<clipped_spring>(0.05f);
    }
    a.copy_to_host();
    b.copy_to_host();
    EXPECT(*a.h_n == *b.h_n);
    EXPECT(memcmp(a.h_X, b.h_X, (size_t)a.n_max * sizeof(float3)) == 0);
}

int main()
{
    {
        Solution_n<float3, 800, Tile_solver> bodies;   // default-constructed, as the old spelling was
        Solution<float3, Tile_solver> reference{800};
        EXPECT(bodies.n_max == 800 && decltype(bodies)::capacity == 800 && *bodies.h_n == 800);
        same_steps(bodies, reference);
    }
    {
        Solution_n<float3, 3000, Grid_solver> cells{40, 1.f};   // solver arguments follow
        Solution<float3, Grid_solver> reference{3000, 40, 1.f};
        EXPECT(cells.n_max == 3000 && cells.cube_size == 1.f);
        same_steps(cells, reference);
        Solution<float3, Grid_solver>& as_base = cells;   // a Solution_n IS a Solution
        EXPECT(as_base.get_d_n() == 3000);
    }
    printf(failures ? "%d FAILURES\n" : "ALL SOLUTION_N TESTS PASSED\n", failures);
    return failures != 0;
}
