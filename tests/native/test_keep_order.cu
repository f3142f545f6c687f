// Solution::keep_in_cube_order (include/solvers.cuh): ONE line after a model has made its id-indexed arrays
// instead of a `cells.renumber(...)` call in its loop.  A model whose functor reads a Property by cell id,
// with Links and a plain device array in tow, a kernel that appends cells between steps: registered once
// against renumbering by hand every third step -- the same cells, properties, links and tags bit for bit.
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/links.cuh"
#include "../../include/property.cuh"
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

__device__ int* d_type;

// adhesion by type, read by ORIGINAL id from inside the pairwise functor (examples/passive_growth.cu:30-57)
__device__ float3 typed_spring(float3 Xi, float3 r, float dist, int i, int j)
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j || dist >= 1.f) return dF;
    const float k = d_type[i] == d_type[j] ? 2.f : 1.f;
    return r * (k * (0.6f - dist) / dist);
}

// a daughter for every 16th cell, appended at d_X[n] as the reference's proliferate kernels do
__global__ void divide(int n, float3* d_X, float3* d_old_v, int* d_n, int* type, float* tag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *d_n = n + (n + 12) / 16;   // cells 3, 19, 35, ... divide
    if (i >= n || i % 16 != 3) return;
    const int k = n + i / 16;               // (a fixed place: an atomicAdd's arrival order differs run to run)
    d_X[k] = float3{d_X[i].x + 0.05f, d_X[i].y - 0.03f, d_X[i].z + 0.02f};
    d_old_v[k] = d_old_v[i];
    type[k] = type[i];
    tag[k] = tag[i] + 0.5f;
}

struct Result {
    std::vector<float3> X;
    std::vector<int> type, links;
    std::vector<float> tag;
    int n;
};

Result run(bool registered)
{
    const int n0 = 3000, n_max = 6000;
    Solution<float3, Grid_solver> cells{n_max, 40, 1.f};
    *cells.h_n = n0;
    random_sphere(0.7f, cells, 0, 23);
    Property<int> type{n_max, "type"};
    for (int i = 0; i < n_max; i++) type.h_prop[i] = (i * 7) % 3;
    type.copy_to_device();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(d_type), &type.d_prop, sizeof(int*));
    float* d_tag;
    std::vector<float> tag(n_max);
    for (int i = 0; i < n_max; i++) tag[i] = (float)i;
    (void)hipMalloc(&d_tag, n_max * sizeof(float));
    (void)hipMemcpy(d_tag, tag.data(), n_max * sizeof(float), hipMemcpyHostToDevice);
    Links links{500, 0.f};
    for (int k = 0; k < 500; k++) links.h_link[k] = Link{(k * 5) % n0, (k * 11 + 1) % n0};
    links.copy_to_device();

    if (registered) cells.keep_in_cube_order(3, type, d_tag, links);   // <- the one line
    for (int step = 0; step < 10; step++) {
        if (!registered && step % 3 == 0) cells.renumber(type, d_tag, links);
        cells.take_step<typed_spring>(0.02f);
        if (step % 4 == 1) {   // cells appended between steps: the next renumbering places them
            const int n = cells.get_d_n();
            divide<<<(n + 255) / 256, 256>>>(n, cells.d_X, cells.d_old_v, cells.d_n, type.d_prop, d_tag);
        }
    }
    Result out;
    cells.copy_to_host();
    out.n = *cells.h_n;
    out.X.assign(cells.h_X, cells.h_X + out.n);
    type.copy_to_host();
    out.type.assign(type.h_prop, type.h_prop + out.n);
    links.copy_to_host();
    out.links.assign((int*)links.h_link, (int*)links.h_link + 2 * 500);
    out.tag.resize(out.n);
    (void)hipMemcpy(out.tag.data(), d_tag, out.n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_tag);
    return out;
}

int main()
{
    const Result by_hand = run(false), once = run(true);
    EXPECT(by_hand.n == once.n && by_hand.n > 3000);
    EXPECT(memcmp(by_hand.X.data(), once.X.data(), by_hand.n * sizeof(float3)) == 0);
    EXPECT(by_hand.type == once.type);
    EXPECT(by_hand.links == once.links);
    EXPECT(by_hand.tag == once.tag);
    // the ids really moved: a tag no longer sits at its own index
    int moved = 0;
    for (int i = 0; i < once.n; i++) moved += once.tag[i] != (float)i;
    EXPECT(moved > once.n / 2);
    printf(failures ? "%d FAILURES\n" : "ALL KEEP-ORDER TESTS PASSED\n", failures);
    return failures != 0;
}
