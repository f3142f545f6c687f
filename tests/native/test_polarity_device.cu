// The polarity force library ON THE DEVICE: the numeric known answers of the reference's
// tests/test_polarity.cu (:20-34 polarization force, :78-94 bending force, :175-193
// migration force; its second operand given as a Polarity) evaluated in a
// kernel (device libm), checked against those numbers (the reference's isclose, 1e-2
// relative) and against the host evaluation of the same __device__ __host__ code (1e-5).
#include "../../include/dtypes.cuh"
#include "../../include/polarity.cuh"
#include "../../include/solvers.cuh"

#include <cmath>
#include <cstdio>
#include <cstring>

struct Results {
    Po_cell polarization, polarization_whole_point, aligning_whole_point, bending, migration_i, migration_j, apical;
    float dot_whole_point, dot_polarity;
    Polarity inverse;
    float3 normal;
};

__host__ __device__ void evaluate(Results* out)
{
    {
        Po_cell i{0.601f, 0.305f, 0.320f, 0.209f, 0.295f};
        Polarity j{0.340f, 0.431f};
        out->polarization = bidirectional_polarization_force(i, j);
        // the partner as a whole point, as tests/test_polarity.cu:22-25 and examples/polarization.cu:29 write it
        Po_cell whole{0.762f, 0.403f, 0.121f, 0.340f, 0.431f};
        out->polarization_whole_point = bidirectional_polarization_force(i, whole);
        out->aligning_whole_point = unidirectional_polarization_force(i, whole) - unidirectional_polarization_force(i, j);
        out->dot_whole_point = pol_dot_product(j, whole);  // a Polarity against a point (test_polarity.cu:65)
        out->dot_polarity = pol_dot_product(whole, j);
    }
    Po_cell bi{0.935f, 0.675f, 0.649f, 0.793f, 0.073f}, bj{0.566f, 0.809f, 0.533f, 0.297f, 0.658f};
    {
        auto r = bi - bj;
        auto dist = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z);
        out->bending = bending_force(bi, r, dist);
        out->apical = apical_constriction_force(bi, r, dist, (float)(M_PI / 2));
    }
    {
        Po_cell Xi{0}, Xj{0};
        Xi.theta = M_PI / 2;
        Xj.x = 1;
        Xj.y = 1e-3;
        out->migration_i = migration_force(Xi, Xi - Xj, 1);
        out->migration_j = migration_force(Xj, Xj - Xi, 1);
    }
    {
        Polarity pol{1.234f, -2.1f};
        out->inverse = pt_to_pol(pol_to_float3(pol));
        float3 r{0.3f, 0.8f, 0.1f}, p{0.2f, 0.5f, 0.7f};
        p = p / sqrtf(dot_product(p, p));
        out->normal = orthonormal(r, p);
    }
}

__global__ void evaluate_on_device(Results* out) { evaluate(out); }

static int failures = 0;
static bool isclose(float a, float b) { return std::fabs(a - b) <= 1e-6 + 1e-2 * std::fabs(b); }
#define EXPECT(cond)                                                 \
    do {                                                             \
        if (!(cond)) {                                               \
            printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);  \
            failures++;                                              \
        }                                                            \
    } while (0)

int main()
{
    Results host, dev, *d_out;
    evaluate(&host);
    YA_CHECK(ya_malloc((void**)&d_out, sizeof(Results)));
    evaluate_on_device<<<1, 1>>>(d_out);
    YA_CHECK(ya_memcpy_d2h(&dev, d_out, sizeof(Results)));
    ya_free(d_out);

    // the reference's numbers, on the device
    EXPECT(isclose(dev.polarization.x, 0) && isclose(dev.polarization.y, 0) && isclose(dev.polarization.z, 0));
    EXPECT(isclose(dev.polarization.theta, 0.126f) && isclose(dev.polarization.phi, 0.215f));
    EXPECT(memcmp(&dev.polarization_whole_point, &dev.polarization, sizeof(Po_cell)) == 0);
    EXPECT(dev.aligning_whole_point.theta == 0 && dev.aligning_whole_point.phi == 0);
    EXPECT(isclose(dev.dot_whole_point, 1) && dev.dot_whole_point == dev.dot_polarity);
    EXPECT(isclose(dev.bending.x, 0.214f) && isclose(dev.bending.y, -0.971f) && isclose(dev.bending.z, -1.802f));
    EXPECT(isclose(dev.bending.theta, -0.339f) && isclose(dev.bending.phi, 0.453f));
    EXPECT(isclose(dev.migration_i.x, 0.6f) && isclose(dev.migration_i.y, -0.8f) &&
           std::fabs(dev.migration_i.z) < 5e-5f);
    EXPECT(isclose(dev.migration_i.x, -dev.migration_j.x) && isclose(dev.migration_i.y, -dev.migration_j.y));
    EXPECT(isclose(dev.inverse.theta, 1.234f) && isclose(dev.inverse.phi, -2.1f));
    EXPECT(std::fabs(dev.normal.x * dev.normal.x + dev.normal.y * dev.normal.y + dev.normal.z * dev.normal.z - 1) < 1e-5f);
    // pi/2 is the plain bending force
    EXPECT(isclose(dev.apical.x, dev.bending.x) && isclose(dev.apical.theta, dev.bending.theta));

    // device libm against the host's: every component within 1e-5 (absolute, values are O(1))
    const float* h = reinterpret_cast<const float*>(&host);
    const float* d = reinterpret_cast<const float*>(&dev);
    for (size_t k = 0; k < sizeof(Results) / sizeof(float); k++) EXPECT(std::fabs(h[k] - d[k]) <= 1e-5f);

    printf(failures ? "%d FAILURES\n" : "ALL DEVICE POLARITY TESTS PASSED\n", failures);
    return failures != 0;
}
