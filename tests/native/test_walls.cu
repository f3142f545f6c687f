// Known-answer test of the wall forces (include/links.cuh, reference links.cuh:142-228):
// xy_wall_relu_force pushes a cell within 1 of the wall plane by
// F = max(0.8 - d, 0) - max(d - 0.8, 0) along z, the wall node collects -F and is
// averaged over its interactions; link_wall_forces adds the link forces first.
#include "../../include/dtypes.cuh"
#include "../../include/links.cuh"
#include "../../include/solvers.cuh"

#include <cmath>
#include <cstdio>
#include <vector>

static int failures = 0;
#define EXPECT_NEAR(a, b)                                                                \
    do {                                                                                 \
        if (!(std::fabs((a) - (b)) <= 1e-6f)) {                                          \
            printf("FAIL %s:%d  %s = %.8g, expected %.8g\n", __FILE__, __LINE__, #a,    \
                (double)(a), (double)(b));                                               \
            failures++;                                                                  \
        }                                                                                \
    } while (0)

template<typename Pt>
std::vector<Pt> run(const std::vector<Pt>& X, int wall_idx, Links* links)
{
    const int n = (int)X.size();
    Pt *d_X, *d_dX;
    YA_CHECK(ya_malloc((void**)&d_X, n * sizeof(Pt)));
    YA_CHECK(ya_malloc((void**)&d_dX, n * sizeof(Pt)));
    YA_CHECK(ya_memcpy_h2d(d_X, X.data(), n * sizeof(Pt)));
    YA_CHECK(ya_memset_async(d_dX, 0, n * sizeof(Pt), nullptr));
    if (links)
        link_wall_forces<Pt, linear_force<Pt>, xy_wall_relu_force<Pt>>(*links, n, d_X, d_dX, wall_idx);
    else
        wall_forces<Pt, xy_wall_relu_force<Pt>>(n, d_X, d_dX, wall_idx);
    std::vector<Pt> dX(n);
    YA_CHECK(ya_memcpy_d2h(dX.data(), d_dX, n * sizeof(Pt)));
    ya_free(d_X);
    ya_free(d_dX);
    return dX;
}

int main()
{
    // cells 0..2, wall node 3 at z = 0
    std::vector<float3> X{{0.3f, 0.f, 0.5f}, {1.f, 2.f, -0.9f}, {0.f, 0.f, 3.f}, {5.f, 5.f, 0.f}};
    auto dX = run(X, 3, nullptr);
    EXPECT_NEAR(dX[0].z, 0.3f);   // d = 0.5: repelled
    EXPECT_NEAR(dX[1].z, -0.1f);  // d = 0.9: attracted branch of the relu
    EXPECT_NEAR(dX[2].z, 0.f);    // out of range
    EXPECT_NEAR(dX[0].x, 0.f);
    EXPECT_NEAR(dX[3].z, -(0.3f - 0.1f) / 2);  // opposite forces, averaged over 2 interactions
    EXPECT_NEAR(dX[3].x, 0.f);

    // the same call again reuses the counters: same answer
    dX = run(X, 3, nullptr);
    EXPECT_NEAR(dX[3].z, -(0.3f - 0.1f) / 2);

    // a wall node nobody touches keeps a zero right-hand side
    std::vector<float3> far{{0.f, 0.f, 4.f}, {0.f, 0.f, 0.f}};
    dX = run(far, 1, nullptr);
    EXPECT_NEAR(dX[0].z, 0.f);
    EXPECT_NEAR(dX[1].z, 0.f);

    // with links: cells 0 and 2 linked (strength 0.2, linear_force = s r / dist)
    Links links{4, 0.2f};
    links.h_link[0].a = 0;
    links.h_link[0].b = 2;
    *links.h_n = 1;
    links.copy_to_device();
    dX = run(X, 3, &links);
    const float rx = 0.3f, rz = 0.5f - 3.f, dist = std::sqrt(rx * rx + rz * rz);
    EXPECT_NEAR(dX[0].x, -0.2f * rx / dist);
    EXPECT_NEAR(dX[0].z, 0.3f - 0.2f * rz / dist);
    EXPECT_NEAR(dX[2].x, 0.2f * rx / dist);
    EXPECT_NEAR(dX[2].z, 0.2f * rz / dist);
    EXPECT_NEAR(dX[3].z, -(0.3f - 0.1f) / 2);

    // float4 points work the same (w untouched)
    std::vector<float4> X4{{0.f, 0.f, 0.25f, 7.f}, {0.f, 0.f, 0.f, 9.f}};
    auto dX4 = run(X4, 1, nullptr);
    EXPECT_NEAR(dX4[0].z, 0.55f);
    EXPECT_NEAR(dX4[0].w, 0.f);
    EXPECT_NEAR(dX4[1].z, -0.55f);

    printf(failures ? "%d FAILURES\n" : "ALL WALL TESTS PASSED\n", failures);
    return failures != 0;
}
