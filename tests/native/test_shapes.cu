// Geometry of the deterministic and seeded initial conditions of include/inits.cuh
// (reference inits.cuh:14-76,157-247): what the reference's own test_inits does not look at.
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/solvers.cuh"

#include <chrono>
#include <cmath>
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

template<typename S>
float nearest(const S& p, int i)
{
    float best = 1e30f;
    for (int j = 0; j < *p.h_n; j++) {
        if (j == i) continue;
        const float dx = p.h_X[i].x - p.h_X[j].x, dy = p.h_X[i].y - p.h_X[j].y,
                    dz = p.h_X[i].z - p.h_X[j].z;
        best = fminf(best, sqrtf(dx * dx + dy * dy + dz * dz));
    }
    return best;
}

int main()
{
    {  // hexagon: centre + rings of 6, 12, 18 ... cells, every nearest neighbour at dist_to_nb
        Solution<float3, Tile_solver> p{1 + 6 + 12 + 18};
        regular_hexagon(0.75f, p);
        for (int i = 0; i < *p.h_n; i++) {
            EXPECT(p.h_X[i].z == 0.f);
            EXPECT(std::fabs(nearest(p, i) - 0.75f) < 1e-5f);
        }
        EXPECT(p.h_X[0].x == 0.f && p.h_X[0].y == 0.f);
        // ring r lies between r * d * sqrt(3)/2 and r * d from the centre
        for (int i = 1 + 6 + 12; i < *p.h_n; i++) {
            const float r = std::hypot(p.h_X[i].x, p.h_X[i].y);
            EXPECT(r < 3 * 0.75f + 1e-5f && r > 3 * 0.75f * 0.8660254f - 1e-5f);
        }
        p.copy_to_host();  // regular_hexagon copied to the device: this reads the same back
        EXPECT(std::fabs(nearest(p, 5) - 0.75f) < 1e-5f);
    }
    {  // rectangle: rows of nx on a triangular lattice
        Solution<float3, Tile_solver> p{4 * 7};
        regular_rectangle(0.5f, 7, p);
        for (int i = 0; i < *p.h_n; i++) {
            EXPECT(p.h_X[i].z == 0.f);
            EXPECT(std::fabs(nearest(p, i) - 0.5f) < 1e-5f);
        }
        EXPECT(std::fabs(p.h_X[7].x - 0.25f) < 1e-6f);                   // odd rows shifted by d / 2
        EXPECT(std::fabs(p.h_X[7].y - 0.5f * 0.8660254f) < 1e-6f);       // row spacing sqrt(3)/2 d
        EXPECT(std::fabs(p.h_X[14].x) < 1e-6f);
    }
    {  // seeded random inits: reproducible, bounded, n_0 leaves earlier cells alone
        Solution<float3, Tile_solver> a{500}, b{500};
        random_sphere(0.5f, a, 0, 7);
        random_sphere(0.5f, b, 0, 7);
        const float r_max = std::pow(500 / 0.64, 1. / 3) * 0.5 / 2;
        bool same = true;
        for (int i = 0; i < 500; i++) {
            same = same && a.h_X[i].x == b.h_X[i].x && a.h_X[i].z == b.h_X[i].z;
            EXPECT(std::sqrt(a.h_X[i].x * a.h_X[i].x + a.h_X[i].y * a.h_X[i].y + a.h_X[i].z * a.h_X[i].z) <=
                   r_max * 1.00001f);
        }
        EXPECT(same);
        random_sphere(0.5f, b, 0, 8);
        EXPECT(b.h_X[3].x != a.h_X[3].x);
        const float3 kept = a.h_X[10];
        random_sphere(0.5f, a, 100, 9);
        EXPECT(a.h_X[10].x == kept.x && a.h_X[10].y == kept.y);

        random_disk(0.5f, a, 0, 7);
        const float disk_r = std::pow(500 / 0.9069, 1. / 2) * 0.5 / 2;
        for (int i = 0; i < 500; i++) {
            EXPECT(a.h_X[i].x == 0.f);  // the disk lies in the y-z plane (inits.cuh:26-28)
            EXPECT(std::hypot(a.h_X[i].y, a.h_X[i].z) <= disk_r * 1.00001f);
        }

        Solution<float3, Tile_solver> c{5000};
        random_cuboid(0.8f, float3{-1.f, 0.f, 2.f}, float3{3.f, 5.f, 4.f}, c, 0, 7);
        const int expected = (int)(4 * 5 * 2 / (4. / 3 * M_PI * std::pow(0.4, 3)) * 0.64);
        EXPECT(*c.h_n == expected);
        for (int i = 0; i < *c.h_n; i++) {
            EXPECT(c.h_X[i].x >= -1.f && c.h_X[i].x <= 3.f);
            EXPECT(c.h_X[i].y >= 0.f && c.h_X[i].y <= 5.f);
            EXPECT(c.h_X[i].z >= 2.f && c.h_X[i].z <= 4.f);
        }
        EXPECT(c.get_d_n() == expected);  // and the device knows
    }
    {  // relaxed_sphere / relaxed_cuboid (reference inits.cuh:95-155) relax with the cooperative
       // kernels (several lanes per cell; relu_force keeps no per-cell state): the result must be
       // bit for bit what one lane per cell gives, and nearest neighbours end at dist_to_nb
        for (int grid = 0; grid < 2; grid++) {
            const int n = grid ? 4000 : 800;
            std::vector<float3> results[2];
            double seconds[2];
            for (int coop = 0; coop < 2; coop++) {
                auto run = [&](auto& p) {
                    const auto t0 = std::chrono::steady_clock::now();
                    if (coop) {
                        relaxed_sphere(0.75f, p, 0, 11);
                    } else {  // the same steps on the default kernels
                        random_sphere(0.6f, p, 0, 11);
                        for (int i = 0; i < (n <= 1000 ? 1000 : 2000); i++) p.template take_step<relu_force>(0.1f);
                        p.copy_to_host();
                        for (int i = 0; i < n; i++) {
                            p.h_X[i].x *= 0.75 / 0.8;
                            p.h_X[i].y *= 0.75 / 0.8;
                            p.h_X[i].z *= 0.75 / 0.8;
                        }
                    }
                    seconds[coop] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    results[coop].assign(p.h_X, p.h_X + n);
                };
                if (grid) {
                    Solution<float3, Grid_solver> p{n, 50, 1.f};
                    run(p);
                    EXPECT(p.force_variant == -1);  // restored
                } else {
                    Solution<float3, Tile_solver> p{n};
                    run(p);
                    EXPECT(p.lanes_per_cell == 0);
                }
            }
            int different = 0;
            for (int i = 0; i < n; i++) different += memcmp(&results[0][i], &results[1][i], sizeof(float3)) != 0;
            EXPECT(different == 0);
            printf("relaxed_sphere, %d cells, %s: %.3f s (one lane per cell: %.3f s)\n", n,
                grid ? "Grid_solver" : "Tile_solver", seconds[1], seconds[0]);
        }
    }
    printf(failures ? "%d FAILURES\n" : "ALL SHAPE TESTS PASSED\n", failures);
    return failures != 0;
}
