// include/mesh.cuh on the device and on disk: nearest-partner distances cut into slices and wavefront
// shares against a plain host loop (bit for bit: a minimum has no order), the reference-named kernel,
// shape comparison between two Solutions, and a mesh written, read back, copied and assigned.
// Argument: a closed mesh in legacy VTK (tests/mesh_fixtures.py: the torus).
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/mesh.cuh"
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

template<typename Pt1, typename Pt2>
std::vector<float> host_nearest(const int n1, const int n2, const Pt1* X1, const Pt2* X2)
{
    std::vector<float> out(n1);
    for (int i = 0; i < n1; i++) {
        float nearest = INFINITY;
        for (int j = 0; j < n2; j++) {
            const float x = X1[i].x - X2[j].x, y = X1[i].y - X2[j].y, z = X1[i].z - X2[j].z;
            nearest = fminf(nearest, sqrtf(fmaf(z, z, fmaf(y, y, x * x))));  // per pair, as the reference
        }
        out[i] = nearest;
    }
    return out;
}

template<typename Pt1, typename Pt2>
void nearest_case(const int n1, const int n2, const float spread)
{
    Solution<Pt1, Grid_solver> a{n1 + 1};
    Solution<Pt2, Grid_solver> b{n2 + 1};
    *a.h_n = n1;
    *b.h_n = n2;
    ya::seed_rand(n1 * 31 + n2);
    for (int i = 0; i < n1; i++) a.h_X[i].x = spread * ya::unit_rand(), a.h_X[i].y = ya::unit_rand(), a.h_X[i].z = -ya::unit_rand();
    for (int j = 0; j < n2; j++) b.h_X[j].x = spread * ya::unit_rand(), b.h_X[j].y = ya::unit_rand(), b.h_X[j].z = -ya::unit_rand();
    if (n1 > 3 && n2 > 3) b.h_X[2] = Pt2{a.h_X[3].x, a.h_X[3].y, a.h_X[3].z};  // a pair at distance zero
    a.copy_to_device();
    b.copy_to_device();
    const auto expect = host_nearest(n1, n2, a.h_X, b.h_X);

    float* d_dist;
    YA_CHECK(ya_malloc((void**)&d_dist, (size_t)(n1 + 1) * sizeof(float)));
    std::vector<float> got(n1 + 1);
    ya::nearest::distances(n1, n2, a.d_X, b.d_X, d_dist);
    YA_CHECK(ya_memcpy_d2h(got.data(), d_dist, n1 * sizeof(float)));
    EXPECT(memcmp(got.data(), expect.data(), n1 * sizeof(float)) == 0);
    if (n2 == 0)
        for (int i = 0; i < n1; i++) EXPECT(got[i] == INFINITY);

    // the reference's kernel by name, launched as the reference launches it (mesh.cuh:64)
    if (n1 > 0) {
        compute_minimum_distance<<<(n1 + TILE_SIZE - 1) / TILE_SIZE, TILE_SIZE>>>(n1, n2, a.d_X, b.d_X, d_dist);
        YA_CHECK(ya_memcpy_d2h(got.data(), d_dist, n1 * sizeof(float)));
        EXPECT(memcmp(got.data(), expect.data(), n1 * sizeof(float)) == 0);
    }
    ya_free(d_dist);

    if (n1 > 0 && n2 > 0) {
        double s12 = 0, s21 = 0;
        for (const float d : expect) s12 += d;
        for (const float d : host_nearest(n2, n1, b.h_X, a.h_X)) s21 += d;
        const double want = (s12 / n1 + s21 / n2) / 2;
        const float have = shape_comparison_points_to_points(a, b);
        EXPECT(fabs(have - want) <= 2e-6 * want + 1e-12);
        EXPECT(shape_comparison_points_to_points(a, b) == have);  // the same bits every time
        EXPECT(shape_comparison_points_to_points(a, a) == 0.f);
    }
    printf("nearest %d x %d ok\n", n1, n2);
}

static bool same(const float3& a, const float3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static bool same(const Triangle& a, const Triangle& b)
{
    return same(a.V0, b.V0) && same(a.V1, b.V1) && same(a.V2, b.V2) && same(a.C, b.C) && same(a.n, b.n);
}

int main(int argc, char** argv)
{
    nearest_case<float3, float3>(1, 1, 1.f);
    nearest_case<float3, float3>(70, 0, 1.f);        // nothing to be near to
    nearest_case<float3, float3>(63, 1025, 1.f);     // one workgroup, a chunk and a bit
    nearest_case<float3, Po_cell>(1000, 5000, 3.f);  // slices meet through atomicMin; two point types
    nearest_case<Po_cell, float3>(4100, 130, 0.1f);
    nearest_case<float3, float3>(300, 70001, 7.f);   // many slices

    if (argc > 1) {
        Mesh mesh{argv[1]};
        EXPECT(mesh.vertices.size() > 100 && mesh.facets.size() % 2 == 0);
        EXPECT(mesh.triangle_to_vertices.size() == mesh.facets.size());
        // a closed surface: every vertex sits in at least three triangles, three corners per triangle in all
        size_t corners = 0;
        for (const auto& list : mesh.vertex_to_triangles) {
            EXPECT(list.size() >= 3);
            corners += list.size();
        }
        EXPECT(corners == 3 * mesh.facets.size());
        for (const auto& f : mesh.facets) EXPECT(fabsf(sqrtf(dot_product(f.n, f.n)) - 1.f) < 1e-6f);

        // written with three points per facet, read back: the same triangles up to the text's six digits
        mesh.write_vtk("round_trip");
        Mesh back{"output/round_trip.mesh.vtk"};
        EXPECT(back.facets.size() == mesh.facets.size() && back.vertices.size() == 3 * mesh.facets.size());
        float worst = 0;
        for (size_t t = 0; t < mesh.facets.size(); t++)
            for (const float3 d : {back.facets[t].V0 - mesh.facets[t].V0, back.facets[t].V1 - mesh.facets[t].V1,
                     back.facets[t].V2 - mesh.facets[t].V2, back.facets[t].n - mesh.facets[t].n})
                worst = fmaxf(worst, fmaxf(fabsf(d.x), fmaxf(fabsf(d.y), fabsf(d.z))));
        EXPECT(worst < 2e-3f);  // normals of small triangles feel the rounding of the corners
        // and the inside stays the inside
        int differ = 0;
        ya::seed_rand(5);
        for (int k = 0; k < 2000; k++) {
            const float3 p{3.f * ya::unit_rand() - 1.5f, 3.f * ya::unit_rand() - 1.5f, ya::unit_rand() - 0.5f};
            differ += mesh.test_exclusion(p) != back.test_exclusion(p);
        }
        EXPECT(differ <= 4);  // points within the text's rounding of the surface

        // copies own their device room; assignment (declared but not defined in the reference) works
        Mesh copy{mesh};
        Mesh assigned;
        assigned = mesh;
        assigned = assigned;
        EXPECT(copy.d_vertices != mesh.d_vertices && assigned.d_vertices != mesh.d_vertices);
        EXPECT(copy.d_vertices != nullptr && assigned.d_vertices != nullptr);
        bool equal = copy.facets.size() == mesh.facets.size() && assigned.vertices.size() == mesh.vertices.size();
        for (size_t t = 0; equal && t < mesh.facets.size(); t++)
            equal = same(copy.facets[t], mesh.facets[t]) && same(assigned.facets[t], mesh.facets[t]);
        EXPECT(equal);

        // the mesh against a cloud made of its own vertices, float3 vertices against Po_cell points
        Solution<Po_cell, Tile_solver> cloud{(int)mesh.vertices.size()};
        for (size_t i = 0; i < mesh.vertices.size(); i++)
            cloud.h_X[i] = Po_cell{mesh.vertices[i].x, mesh.vertices[i].y, mesh.vertices[i].z, 0.f, 0.f};
        cloud.copy_to_device();
        assigned.copy_to_device();
        EXPECT(assigned.shape_comparison_mesh_to_points(cloud) == 0.f);
        assigned.translate(float3{0.f, 0.f, 10.f});
        assigned.copy_to_device();
        const float far = assigned.shape_comparison_mesh_to_points(cloud);
        EXPECT(far > 9.f && far < 10.5f);

        // turning there and back again lands where it started; a full turn about every axis too
        Mesh turned{mesh};
        turned.rotate(0.3f, -1.1f, 2.f);
        float moved = 0;
        for (size_t i = 0; i < mesh.vertices.size(); i++) moved = fmaxf(moved, fabsf(turned.vertices[i].x - mesh.vertices[i].x));
        EXPECT(moved > 0.1f);
        for (size_t t = 0; t < mesh.facets.size(); t += 97) {
            const auto& f = turned.facets[t];
            const float3 u = f.V1 - f.V0, v = f.V2 - f.V0;
            EXPECT(fabsf(dot_product(f.n, u)) < 1e-5f && fabsf(dot_product(f.n, v)) < 1e-5f);
            EXPECT(same(f.V0, turned.vertices[turned.triangle_to_vertices[t][0]]));
        }
        // the inverse of z-then-y-then-x is x, y, z with opposite signs: three calls
        turned.rotate(0.f, 0.f, -2.f);
        turned.rotate(0.f, 1.1f, 0.f);
        turned.rotate(-0.3f, 0.f, 0.f);
        float off = 0;
        for (size_t i = 0; i < mesh.vertices.size(); i++) {
            const float3 d = turned.vertices[i] - mesh.vertices[i];
            off = fmaxf(off, fmaxf(fabsf(d.x), fmaxf(fabsf(d.y), fabsf(d.z))));
        }
        EXPECT(off < 1e-5f);
    }
    printf(failures ? "%d FAILURES\n" : "ALL MESH TESTS PASSED\n", failures);
    return failures != 0;
}
