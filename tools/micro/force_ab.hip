// Same-process A/B of the grid force kernels on the bench workload (springs, random_sphere):
//   force_ab [cells] [warm take_steps] [rounds] [dist]
// Builds the grid once on a state that `warm` take_steps have relaxed, then times
// variant 1 (grid_force, byte FIFO) and variant 2 (grid_force_bits, compiled with this
// executable's -DYA_BITS_BLOCK / -DYA_BITS_POPS / -DYA_MASK_WORDS flags) in interleaved rounds with HIP events,
// (AB_BASE / AB_TEST select other pairs: 3 = grid_force_coop, 12 = grid_force_bits with old_v in LDS,
// 102 = grid_force_bits summing by plane, with the tail of half tiles)
// and compares their outputs (d_dX by id and d_dX in sorted order) bit for bit.
// One JSON line per run; tools/micro/force_ab.sh builds and runs a set of configurations.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#include "dtypes.cuh"
#include "inits.cuh"
#include "solvers.cuh"

#include "model_functors.h"

#ifndef AB_TAG
#define AB_TAG "default"
#endif
#ifndef AB_BASE
#define AB_BASE 1  /* force_variant timed as the reference */
#endif
#ifndef AB_TEST
#define AB_TEST 2  /* force_variant under test */
#endif

using Pt = float3;

struct Probe : public Solution<Pt, Grid_solver> {
    using Solution<Pt, Grid_solver>::Solution;
    void build(int n) { this->grid.build_sorted(n, this->d_X, this->d_old_v, this->cube_size, this->d_sorted, this->d_sorted_v); }
    void run(int n, int variant, Pt* out, Pt* out_sorted)
    {
        // variant % 100 >= 10: old_v staged in LDS whatever n is; < 10: never staged
        this->force_variant = variant % 10;
#ifdef YA_COOP_LANES
        this->coop_lanes = YA_COOP_LANES;  // variant 3: fixed instead of chosen from n
#endif
        this->stage_v_max = variant % 100 >= 10 ? 2000000000 : 0;
        // variants >= 100: Grid_computer::sum_order = YA_SUM_BY_PLANE (half tiles where the engine chooses them)
        this->sum_order = variant >= 100 ? YA_SUM_BY_PLANE : YA_SUM_REFERENCE;
#ifdef AB_TAIL_TILES
        this->force_tail_tiles = AB_TAIL_TILES;
#endif
        this->template forces<models::spring, friction_w_neighbour<Pt>>(
            n, this->d_sorted, this->d_sorted_v, out, false, n, out_sorted);
    }
};

static double median(std::vector<float> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1000000;
    const int warm = argc > 2 ? atoi(argv[2]) : 10;
    const int rounds = argc > 3 ? atoi(argv[3]) : 30;
    const float dist = argc > 4 ? atof(argv[4]) : 0.5f;
    const float radius = powf(n / 0.64f, 1.f / 3) * dist / 2;
    const int gs = std::max(2 * ((int)radius + 3), 8);

    Probe cells{n, gs, 1.0f};
    random_sphere(dist, cells, 0, 42);
    for (int s = 0; s < warm; s++) cells.take_step<models::spring>(0.001f);
    (void)hipDeviceSynchronize();
    cells.build(n);

    Pt *d_out[2], *d_outs[2];
    for (int k = 0; k < 2; k++) {
        (void)hipMalloc(&d_out[k], (size_t)n * sizeof(Pt));
        (void)hipMalloc(&d_outs[k], (size_t)n * sizeof(Pt));
        (void)hipMemset(d_out[k], 0xff, (size_t)n * sizeof(Pt));
        (void)hipMemset(d_outs[k], 0xff, (size_t)n * sizeof(Pt));
    }
    const int variants[2] = {AB_BASE, AB_TEST};
    for (int k = 0; k < 2; k++) cells.run(n, variants[k], d_out[k], d_outs[k]);
    (void)hipDeviceSynchronize();
    std::vector<Pt> a(n), b(n);
    long mismatches = 0;
    double max_abs = 0, max_rel = 0;  // kernels that sum in another order (by plane) differ by rounding
    for (int which = 0; which < 2; which++) {
        (void)hipMemcpy(a.data(), which ? d_outs[0] : d_out[0], (size_t)n * sizeof(Pt), hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), which ? d_outs[1] : d_out[1], (size_t)n * sizeof(Pt), hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) {
            mismatches += memcmp(&a[i], &b[i], sizeof(Pt)) != 0;
            const float* fa = (const float*)&a[i];
            const float* fb = (const float*)&b[i];
            double norm = 0;
            for (size_t k = 0; k < sizeof(Pt) / 4; k++) norm = std::max(norm, (double)fabsf(fa[k]));
            for (size_t k = 0; k < sizeof(Pt) / 4; k++) {
                const double d = fabs((double)fa[k] - fb[k]);
                max_abs = std::max(max_abs, d);
                if (norm > 0) max_rel = std::max(max_rel, d / norm);
            }
        }
    }

    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> us[2];
    for (int r = 0; r < rounds + 3; r++)
        for (int k = 0; k < 2; k++) {
            (void)hipEventRecord(e0, nullptr);
            cells.run(n, variants[k], d_out[k], d_outs[k]);
            (void)hipEventRecord(e1, nullptr);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 3) us[k].push_back(ms * 1e3f);
        }
    printf("{\"tag\": \"%s\", \"base\": %d, \"test\": %d, \"cells\": %d, \"gs\": %d, \"dist\": %g, \"warm\": %d, \"rounds\": %d, "
           "\"fifo_us_median\": %.1f, \"fifo_us_min\": %.1f, \"bits_us_median\": %.1f, \"bits_us_min\": %.1f, "
           "\"mismatches\": %ld, \"max_abs\": %.3g, \"max_rel\": %.3g, \"block\": %d, \"words\": %d, \"pops\": %d}\n",
        AB_TAG, AB_BASE, AB_TEST, n, gs, dist, warm, rounds, median(us[0]), *std::min_element(us[0].begin(), us[0].end()),
        median(us[1]), *std::min_element(us[1].begin(), us[1].end()), mismatches, max_abs, max_rel, ya::bits::BLOCK,
        ya::bits::WORDS, YA_BITS_POPS);
    return mismatches != 0 && (AB_TEST >= 100) == (AB_BASE >= 100);  // the two summation orders differ by rounding
}
