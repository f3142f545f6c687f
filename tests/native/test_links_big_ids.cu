// Links::link_forces with cell ids at and above 2^24 (a 75 M-cell system has them): the
// atomics-free path sorts (cell, entry) pairs by cell and must tell apart ids that share their
// low 24 bits (round 2 sorted 24 key bits and used 0xFFFFFF as the dead key: such ids were
// merged or dropped).  Both paths against a host evaluation of links.cuh:98-111.
#include "../../include/dtypes.cuh"
#include "../../include/links.cuh"
#include "../../include/solvers.cuh"

#include <cmath>
#include <cstdio>
#include <vector>

static int failures = 0;
#define EXPECT(cond)                                                  \
    do {                                                              \
        if (!(cond)) {                                                \
            printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
            failures++;                                               \
        }                                                             \
    } while (0)

int main()
{
    const int N = (1 << 24) + 4096;
    // ids that collide in their low 24 bits, the old dead key, and ordinary ones
    const int ids[] = {5, (1 << 24) + 5, 0xFFFFFF, (1 << 24) - 2, (1 << 24) + 7, 7, 123456, (1 << 24) + 4095};
    const int n_ids = sizeof(ids) / sizeof(ids[0]);
    const Link pairs[] = {{5, (1 << 24) + 5}, {0xFFFFFF, 7}, {(1 << 24) + 7, 7}, {123456, (1 << 24) + 4095},
        {(1 << 24) - 2, (1 << 24) + 5}, {5, 5} /* inert */, {0xFFFFFF, (1 << 24) + 7}, {(1 << 24) + 5, 5}};
    const int n_links = sizeof(pairs) / sizeof(pairs[0]);

    float3 *d_X, *d_dX;
    EXPECT(hipMalloc(&d_X, (size_t)N * sizeof(float3)) == hipSuccess);
    EXPECT(hipMalloc(&d_dX, (size_t)N * sizeof(float3)) == hipSuccess);
    (void)hipMemset(d_X, 0, (size_t)N * sizeof(float3));
    std::vector<float3> X(n_ids);
    for (int k = 0; k < n_ids; k++) {
        X[k] = float3{0.37f * k - 1.f, 0.11f * k * k, 1.5f - 0.29f * k};
        (void)hipMemcpy(d_X + ids[k], &X[k], sizeof(float3), hipMemcpyHostToDevice);
    }
    auto pos = [&](int id) {
        for (int k = 0; k < n_ids; k++)
            if (ids[k] == id) return X[k];
        return float3{0.f, 0.f, 0.f};
    };
    // expected: dX[a] -= s r / |r|, dX[b] += s r / |r| (links.cuh:98-111), in double
    std::vector<double> want(3 * n_ids, 0.);
    const float strength = 0.2f;
    for (int l = 0; l < n_links; l++) {
        if (pairs[l].a == pairs[l].b) continue;
        const float3 a = pos(pairs[l].a), b = pos(pairs[l].b);
        const double r[3] = {(double)a.x - b.x, (double)a.y - b.y, (double)a.z - b.z};
        const double dist = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        for (int k = 0; k < n_ids; k++)
            for (int c = 0; c < 3; c++) {
                if (ids[k] == pairs[l].a) want[3 * k + c] -= strength * r[c] / dist;
                if (ids[k] == pairs[l].b) want[3 * k + c] += strength * r[c] / dist;
            }
    }

    Links links{16, strength};
    for (int l = 0; l < n_links; l++) links.h_link[l] = pairs[l];
    *links.h_n = n_links;
    links.copy_to_device();
    for (int path = 0; path < 2; path++) {
        ya::links_segmented_min() = path == 0 ? 1 : (1 << 30);  // segmented sum, then atomics
        (void)hipMemset(d_dX, 0, (size_t)N * sizeof(float3));
        link_forces<float3>(links, d_X, d_dX);
        (void)hipDeviceSynchronize();
        double worst = 0;
        for (int k = 0; k < n_ids; k++) {
            float3 got;
            (void)hipMemcpy(&got, d_dX + ids[k], sizeof(float3), hipMemcpyDeviceToHost);
            worst = std::fmax(worst, std::fabs(got.x - want[3 * k]));
            worst = std::fmax(worst, std::fabs(got.y - want[3 * k + 1]));
            worst = std::fmax(worst, std::fabs(got.z - want[3 * k + 2]));
        }
        // nothing may land on a cell that only shares low bits with a linked one
        float3 bystander;
        (void)hipMemcpy(&bystander, d_dX + ((1 << 24) + 123456 % 4096), sizeof(float3), hipMemcpyDeviceToHost);
        EXPECT(bystander.x == 0.f && bystander.y == 0.f && bystander.z == 0.f);
        printf("%s: max |dX - expected| = %.3g\n", path == 0 ? "segmented sum" : "atomics", worst);
        EXPECT(worst < 1e-6);
    }
    ya::links_segmented_min() = YA_LINKS_SEGMENTED_MIN;
    (void)hipFree(d_X);
    (void)hipFree(d_dX);
    if (failures == 0) printf("ALL BIG-ID LINK TESTS PASSED\n");
    return failures != 0;
}
