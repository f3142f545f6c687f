// When and where every workgroup of one grid_force_bits launch ran (tools/ab/force_trace.cuh
// fills the kernel's probe hooks: s_memtime at entry and exit and the hardware ids):
//   force_trace [cells] [warm steps] [tail tiles: -1 the engine's choice, 0 whole tiles only] > stamps.csv
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "force_trace.cuh"  // before solvers.cuh: the probe hooks of grid_force_bits

#include "dtypes.cuh"
#include "inits.cuh"
#include "solvers.cuh"

#include "model_functors.h"

using Pt = float3;
struct Probe : public Solution<Pt, Grid_solver> {
    using Solution<Pt, Grid_solver>::Solution;
    void build(int n) { this->grid.build_sorted(n, this->d_X, this->d_old_v, this->cube_size, this->d_sorted, this->d_sorted_v); }
    void run(int n, Pt* out, Pt* out_sorted)
    {
        this->force_variant = 2;
        this->stage_v_max = 0;
        this->template forces<models::spring, friction_w_neighbour<Pt>>(n, this->d_sorted, this->d_sorted_v, out, false, n, out_sorted);
    }
};

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1000000;
    const int warm = argc > 2 ? atoi(argv[2]) : 10;
    const float dist = 0.5f;
    const float radius = powf(n / 0.64f, 1.f / 3) * dist / 2;
    const int gs = std::max(2 * ((int)radius + 3), 8);
    Probe cells{n, gs, 1.0f};
    random_sphere(dist, cells, 0, 42);
    for (int s = 0; s < warm; s++) cells.take_step<models::spring>(0.001f);
    (void)hipDeviceSynchronize();
    cells.build(n);
    Pt *d_out, *d_outs;
    (void)hipMalloc(&d_out, (size_t)n * sizeof(Pt));
    (void)hipMalloc(&d_outs, (size_t)n * sizeof(Pt));
    cells.force_tail_tiles = argc > 3 ? atoi(argv[3]) : -1;
    // half tiles exist under the opt-in summation order only (Grid_computer::sum_order)
    cells.sum_order = cells.force_tail_tiles != 0 ? YA_SUM_BY_PLANE : YA_SUM_REFERENCE;
    const int blocks = 2 * ((n + 63) / 64) + 64;  // room for a launch of half tiles; blocks that never ran stay zero
    unsigned long long* d_trace;
    (void)hipMalloc(&d_trace, (size_t)blocks * 4 * sizeof(unsigned long long));
    (void)hipMemset(d_trace, 0, (size_t)blocks * 4 * sizeof(unsigned long long));
    for (int k = 0; k < 3; k++) cells.run(n, d_out, d_outs);  // warm caches, no stamps
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ya_bits_trace), &d_trace, sizeof(d_trace));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    cells.run(n, d_out, d_outs);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), d_trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    int stamped = 0;
    for (int b = 0; b < blocks; b++) stamped += h[4 * b + 1] != 0;
    printf("# cells %d blocks %d launch_us %.1f tail_tiles %d\n", n, stamped, ms * 1e3f, cells.force_tail_tiles);
    printf("block,t0,t1,hw_id,xcc_id,tile\n");
    for (int b = 0; b < blocks; b++)
        if (h[4 * b + 1] != 0)
            printf("%d,%llu,%llu,%u,%u,%llu\n", b, h[4 * b], h[4 * b + 1], (unsigned)(h[4 * b + 2] & 0xffffffffu),
            (unsigned)(h[4 * b + 2] >> 32), h[4 * b + 3]);
    return 0;
}
