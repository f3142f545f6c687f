// Solution<Pt, Slab_grid_solver> from a model program's point of view (include/slab.cuh): the
// header-level multi-GPU solver on ONE rank (the GPU boxes of this build have one GPU; RCCL needs
// one GPU per rank), where the decomposed step must reproduce Solution<Pt, Grid_solver>:
//   * communicator from the environment (no RANK / WORLD_SIZE: a world of one, no RCCL loaded),
//   * slab_init / slab_setup / slab_use_rccl, then take_step = exchanges (none), all-reduce
//     (none), updates and periodic migration,
//   * an id-indexed functor (the sorting model's `i < n / 2`) with the cells handed to the slab
//     in a shuffled order, so that local index != global id,
//   * the transport callbacks as the other way to move messages.
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/slab.cuh"
#include "../../include/solvers.cuh"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <vector>

static int failures = 0;
#define EXPECT(cond)                                                  \
    do {                                                              \
        if (!(cond)) {                                                \
            printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
            failures++;                                               \
        }                                                             \
    } while (0)

constexpr int N = 20000;

__device__ float3 spring(float3 Xi, float3 r, float dist, int i, int j)
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j) return dF;
    return r * (0.5f - dist) / dist;
}

// examples/sorting.cu:9-28 with n_cells = N: the first half of the IDS adheres more strongly
__device__ float3 differential_adhesion(float3 Xi, float3 r, float dist, int i, int j)
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j) return dF;
    if (dist > 1.f) return dF;
    auto strength = (1 + 2 * (j < N / 2)) * (1 + 2 * (i < N / 2));
    auto F = 2 * (0.5f - dist) * (1.f - dist) + powf(1.f - dist, 2);
    dF = strength * r * F / dist;
    return dF;
}

static int exchanges = 0, reductions = 0;
int count_exchange(void*, int, const void*, long, void*, long, const void*, long, void*, long)
{
    exchanges++;
    return 0;
}
int count_allreduce(void*, float*, int)
{
    reductions++;
    return 0;
}

// a generic force that needs no ids: every cell is pulled towards the origin
__global__ void pull_to_origin(const int n, const float3* __restrict__ d_X, float3* d_dX)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d_dX[i].x -= 0.3f * d_X[i].x;
    d_dX[i].y -= 0.3f * d_X[i].y;
    d_dX[i].z -= 0.3f * d_X[i].z;
}
void origin_forces(const int n, const float3* __restrict__ d_X, float3* d_dX)
{
    pull_to_origin<<<(n + 255) / 256, 256>>>(n, d_X, d_dX);
}
static bool with_generic_forces = false;

template<Pairwise_interaction<float3> force>
void compare(const char* name, float dt, bool callbacks)
{
    Solution<float3, Grid_solver> whole{N, 50, 1.f};
    random_sphere(0.5f, whole, 0, 11);
    std::vector<float3> X0(whole.h_X, whole.h_X + N);

    // the slab gets the same cells in another order: local index k holds global id order[k]
    std::vector<int> order(N);
    std::iota(order.begin(), order.end(), 0);
    for (int k = 0; k < N; k++) std::swap(order[k], order[(k * 7919 + 13) % N]);
    Solution<float3, Slab_grid_solver> slab{N + 64, 50, 1.f};
    for (int k = 0; k < N; k++) slab.h_X[k] = X0[order[k]];
    *slab.h_n = N;
    slab.copy_to_device();
    ya_comm* comm = nullptr;
    EXPECT(ya_comm_create_from_env(1, &comm) == 0);
    EXPECT(ya_comm_world(comm) == 1 && ya_comm_rank(comm) == 0);
    EXPECT(slab.slab_init(-INFINITY, INFINITY, 1.25f, order.data(), N) == 0);
    EXPECT(slab.slab_setup(0, 1, 1024, 1024) == 0);
    if (callbacks)
        EXPECT(slab.slab_set_transport(count_exchange, count_allreduce, nullptr) == 0);
    else
        EXPECT(slab.slab_use_rccl(comm) == 0);
    slab.migrate_every = 2;

    for (int s = 0; s < 5; s++) {
        if (with_generic_forces) {
            // (the decomposed step with generic forces goes through d_X1 and the plain update
            // kernels instead of the sorted-copy predictor and the raw corrector)
            whole.template take_step<force>(dt, origin_forces);
            slab.template take_step<force>(dt, origin_forces);
        } else {
            whole.template take_step<force>(dt);
            slab.template take_step<force>(dt);
        }
    }
    whole.copy_to_host();
    slab.copy_to_host();
    EXPECT(*slab.h_n == N && slab.slab.n_own == N);
    std::vector<float> Xs(3 * (size_t)(N + 64));
    std::vector<int> gid(N + 64);
    EXPECT(slab.get_own(Xs.data(), gid.data()) == N);
    double worst = 0, scale = 0;
    for (int k = 0; k < N; k++) {
        const float3 a = whole.h_X[gid[k]];
        worst = std::max({worst, (double)std::fabs(a.x - Xs[3 * k]), (double)std::fabs(a.y - Xs[3 * k + 1]),
            (double)std::fabs(a.z - Xs[3 * k + 2])});
        scale = std::max({scale, (double)std::fabs(a.x), (double)std::fabs(a.y), (double)std::fabs(a.z)});
    }
    printf("%s (%s): max |slab - undivided| = %.3g (scale %.3g)\n", name, callbacks ? "callbacks" : "ya_comm", worst, scale);
    // cells of a cube are summed in local instead of global id order: rounding-level differences
    EXPECT(worst <= 1e-5 * scale);
    ya_comm_destroy(comm);
}

int main()
{
    compare<spring>("springs", 0.001f, false);
    compare<differential_adhesion>("sorting (id-indexed functor)", 0.002f, false);
    compare<spring>("springs", 0.001f, true);
    with_generic_forces = true;
    compare<spring>("springs + a generic force", 0.001f, true);
    with_generic_forces = false;
    EXPECT(exchanges == 0 && reductions == 0);  // a world of one never calls its transport
    if (failures == 0) printf("ALL SLAB SOLVER TESTS PASSED\n");
    return failures != 0;
}
