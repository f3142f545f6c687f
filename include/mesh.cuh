// Closed triangle meshes for image-based models: read a legacy-VTK surface, move it about, ask
// whether a point lies outside it, and compare the shape of a cloud of points with it.
// API parity with ya||a `include/mesh.cuh:1-462` (Ray, Triangle, Mesh, intersect, shape_comparison,
// shape_comparison_points_to_points, compute_minimum_distance); the arithmetic of every result a
// model can see follows the reference operation by operation (cited below), the way there does not.
#pragma once

#include <assert.h>
#include <math.h>
#include <sys/stat.h>

#include <array>
#include <fstream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "dtypes.cuh"
#include "solvers.cuh"
#include "utils.cuh"
#include "yalla_hip.h"


// ---- shape comparison: the mean distance from every point of A to the nearest point of B, and
// back (mesh.cuh:20-88) ----
//
// The reference finds a point's nearest partner with one thread walking all of B (32-thread
// blocks, one tile of 32 staged at a time).  A minimum does not care in which order it is taken,
// so here the walk is cut up as far as the chip likes: a workgroup owns 64 points of A (one per
// lane) and a SLICE of B; its four wavefronts each take a quarter of every staged chunk (all lanes
// of a wavefront read the same LDS entry: a broadcast), the quarters meet in LDS and the slices
// meet in memory through atomicMin on the bits of the squared distance -- for non-negative
// binary32 values the order of the bit patterns is the order of the numbers.  The square root is
// taken once per point at the end: sqrt is monotone, so sqrt(min d2) == min sqrt(d2) bit for bit,
// and d2 / sqrt are the engine's pair distance (ya::dist3: sqrtf(fmaf(z, z, fmaf(y, y, x * x)))).
namespace ya {
namespace nearest {
constexpr int BLOCK = 256, POINTS = 64, WAVES = BLOCK / POINTS;
constexpr int CHUNK = 1024;  // points of B staged per pass: 16 KB of LDS
constexpr unsigned FAR = 0x7f800000u;  // +inf: "no partner seen yet"

template<typename Pt1, typename Pt2>
__global__ __launch_bounds__(BLOCK) void squared(const int n1, const int n2, const Pt1* __restrict__ d_X1,
    const Pt2* __restrict__ d_X2, const int slice, unsigned* __restrict__ d_min_bits)
{
    __shared__ float4 sh_B[CHUNK];
    __shared__ float sh_min[WAVES][POINTS];
    const int lane = threadIdx.x % POINTS, wave = threadIdx.x / POINTS;
    const int i = blockIdx.x * POINTS + lane;
    float x = 0, y = 0, z = 0;
    if (i < n1) {
        const Pt1 Xi = d_X1[i];
        x = Xi.x, y = Xi.y, z = Xi.z;
    }
    const int j_begin = blockIdx.y * slice, j_end = min(j_begin + slice, n2);
    float nearest = INFINITY;
    for (int j0 = j_begin; j0 < j_end; j0 += CHUNK) {
        const int count = min(CHUNK, j_end - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < count; t += BLOCK) {
            const Pt2 Xj = d_X2[j0 + t];
            sh_B[t] = float4{Xj.x, Xj.y, Xj.z, 0.f};
        }
        __syncthreads();
        const int share = (count + WAVES - 1) / WAVES;
        const int k_end = min(wave * share + share, count);
        for (int k = wave * share; k < k_end; k++) {
            const float4 B = sh_B[k];
            const float rx = x - B.x, ry = y - B.y, rz = z - B.z;
            nearest = fminf(nearest, fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
        }
    }
    sh_min[wave][lane] = nearest;
    __syncthreads();
    if (wave == 0 && i < n1) {
#pragma unroll
        for (int w = 1; w < WAVES; w++) nearest = fminf(nearest, sh_min[w][lane]);
        atomicMin(&d_min_bits[i], __float_as_uint(nearest));
    }
}

__global__ void fill_far(const int n, unsigned* d_min_bits)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d_min_bits[i] = FAR;
}

__global__ void roots(const int n, float* d_min_dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d_min_dist[i] = ya::exact_sqrt(d_min_dist[i]);
}

// d_min_dist[i] = distance from d_X1[i] to the nearest of d_X2[0 .. n2) (+inf if there is none).
template<typename Pt1, typename Pt2>
void distances(const int n1, const int n2, const Pt1* d_X1, const Pt2* d_X2, float* d_min_dist)
{
    if (n1 <= 0) return;
    const int blocks = (n1 + POINTS - 1) / POINTS;
    // enough workgroups for 256 CUs, slices no shorter than one staged chunk
    const int chunks = (n2 + CHUNK - 1) / CHUNK;
    const int slices = std::max(1, std::min(chunks, (2048 + blocks - 1) / blocks));
    const int slice = std::max(1, (chunks + slices - 1) / slices) * CHUNK;
    const auto bits = reinterpret_cast<unsigned*>(d_min_dist);
    fill_far<<<(n1 + 255) / 256, 256>>>(n1, bits);
    if (n2 > 0)
        squared<<<dim3(blocks, (n2 + slice - 1) / slice), BLOCK>>>(n1, n2, d_X1, d_X2, slice, bits);
    roots<<<(n1 + 255) / 256, 256>>>(n1, d_min_dist);
    YA_CHECK((int)hipGetLastError());
}

// Sum of d_v[0 .. n) in ONE fixed order (lane t of a single 1024-lane workgroup adds elements t, t + 1024, ...
// in turn, the lanes' sums fold by halving): reproducible run to run, where the reference's thrust::reduce
// leaves the order open.
__global__ __launch_bounds__(1024) void total(const int n, const float* __restrict__ d_v, float* d_sum)
{
    __shared__ float sh[1024];
    float mine = 0;
    for (int k = threadIdx.x; k < n; k += 1024) mine += d_v[k];
    sh[threadIdx.x] = mine;
    for (int half = 512; half > 0; half >>= 1) {
        __syncthreads();
        if ((int)threadIdx.x < half) sh[threadIdx.x] += sh[threadIdx.x + half];
    }
    if (threadIdx.x == 0) *d_sum = sh[0];
}

inline float sum(const float* d_v, const int n)
{
    float* d_sum;
    YA_CHECK(ya_malloc((void**)&d_sum, sizeof(float)));
    total<<<1, 1024>>>(n, d_v, d_sum);
    YA_CHECK((int)hipGetLastError());
    float out;
    YA_CHECK(ya_memcpy_d2h(&out, d_sum, sizeof(float)));
    ya_free(d_sum);
    return out;
}
}  // namespace nearest
}  // namespace ya

// The reference's kernel under its own name and launch shape (one thread per point of the first
// set, any block size; mesh.cuh:27-56), for model code that launches it itself.
template<typename Pt1, typename Pt2>
__global__ void compute_minimum_distance(
    const int n1, const int n2, const Pt1* __restrict__ d_X1, const Pt2* __restrict__ d_X2, float* d_min_dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const Pt1 Xi = d_X1[i];
    float nearest = INFINITY;
    for (int j = 0; j < n2; j++) {
        const Pt2 Xj = d_X2[j];
        const float rx = Xi.x - Xj.x, ry = Xi.y - Xj.y, rz = Xi.z - Xj.z;
        nearest = fminf(nearest, fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
    }
    d_min_dist[i] = ya::exact_sqrt(nearest);
}

template<typename Pt1, typename Pt2>
float shape_comparison(const int n1, const int n2, const Pt1* __restrict__ d_X1, const Pt2* __restrict__ d_X2)
{
    float* d_dist;
    YA_CHECK(ya_malloc((void**)&d_dist, (size_t)std::max(std::max(n1, n2), 1) * sizeof(float)));
    ya::nearest::distances(n1, n2, d_X1, d_X2, d_dist);
    const float mean_12_dist = ya::nearest::sum(d_dist, n1) / n1;
    ya::nearest::distances(n2, n1, d_X2, d_X1, d_dist);
    const float mean_21_dist = ya::nearest::sum(d_dist, n2) / n2;
    ya_free(d_dist);
    return (mean_12_dist + mean_21_dist) / 2;
}

template<typename Pt1, typename Pt2, template<typename> class Solver1, template<typename> class Solver2>
float shape_comparison_points_to_points(Solution<Pt1, Solver1>& points1, Solution<Pt2, Solver2>& points2)
{
    return shape_comparison(points1.get_d_n(), points2.get_d_n(), points1.d_X, points2.d_X);
}


// ---- the mesh itself (host side; mesh.cuh:91-462) ----
struct Ray {
    float3 P0;
    float3 P1;
    Ray(float3 P0, float3 P1) : P0{P0}, P1{P1} {}
};

struct Triangle {
    float3 V0;
    float3 V1;
    float3 V2;
    float3 C;  // centroid
    float3 n;  // unit normal, (V1 - V0) x (V2 - V0)
    Triangle() : Triangle(float3{0, 0, 0}, float3{0, 0, 0}, float3{0, 0, 0}) {}
    Triangle(float3 V0, float3 V1, float3 V2) : V0{V0}, V1{V1}, V2{V2}
    {
        calculate_centroid();
        calculate_normal();
    }
    void calculate_centroid() { C = (V0 + V1 + V2) / 3.f; }
    void calculate_normal()
    {
        const float3 u = V1 - V0, v = V2 - V0;
        const float3 cross{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
        // the reference's `n /= length` is `n *= float(1. / length)` (dtypes.cuh:202-208), which is what
        // operator/ is here
        n = cross / sqrtf(cross.x * cross.x + cross.y * cross.y + cross.z * cross.z);
    }
};

class Mesh {
public:
    std::vector<float3> vertices;
    std::vector<Triangle> facets;
    float3* d_vertices = nullptr;
    std::vector<std::array<int, 3>> triangle_to_vertices;
    std::vector<std::vector<int>> vertex_to_triangles;
    Mesh() {}
    Mesh(std::string file_name);
    Mesh(const Mesh& copy);
    ~Mesh() { ya_free(d_vertices); }
    Mesh& operator=(const Mesh& other);
    float3 get_minimum();
    float3 get_maximum();
    void translate(float3 offset);
    void rotate(float around_z, float around_y, float around_x);
    void rescale(float factor);
    void grow_normally(float amount, bool boundary = false);
    template<typename Pt>
    bool test_exclusion(const Pt point);
    void write_vtk(std::string);
    void copy_to_device();
    template<typename Pt, template<typename> class Solver>
    float shape_comparison_mesh_to_points(Solution<Pt, Solver>& points);

private:
    // f(point) for every stored position: the vertex list, and the corners and centroid the facets keep
    // copies of
    template<typename F>
    void move_points(F f)
    {
        for (auto& vertex : vertices) f(vertex);
        for (auto& facet : facets) {
            f(facet.V0);
            f(facet.V1);
            f(facet.V2);
            f(facet.C);
        }
    }
    void device_room()
    {
        ya_free(d_vertices);
        d_vertices = nullptr;
        YA_CHECK(ya_malloc((void**)&d_vertices, std::max<size_t>(vertices.size(), 1) * sizeof(float3)));
    }
};

namespace ya {
namespace mesh_file {
// The next line of `in` whose first word is one of `keys`, as words; aborts at the end of the file.
inline std::vector<std::string> section(std::istream& in, std::initializer_list<const char*> keys)
{
    for (std::string line; std::getline(in, line);) {
        std::istringstream words_in(line);
        std::vector<std::string> words;
        for (std::string w; words_in >> w;) words.push_back(w);
        if (words.empty()) continue;
        for (const char* key : keys)
            if (words[0] == key) return words;
    }
    assert(!"section missing in mesh file");
    abort();
}
}  // namespace mesh_file
}  // namespace ya

// Legacy ASCII VTK (mesh.cuh:147-206): `POINTS n type` followed by 3 n numbers however they are
// broken into lines, then `POLYGONS m size` or `CELLS m size` followed by m rows `3 a b c`.
inline Mesh::Mesh(std::string file_name)
{
    std::ifstream in(file_name);
    assert(in.is_open());

    const int n_vertices = std::stoi(ya::mesh_file::section(in, {"POINTS"}).at(1));
    vertices.resize(n_vertices);
    for (auto& vertex : vertices) in >> vertex.x >> vertex.y >> vertex.z;
    assert(!in.fail());
    device_room();

    const int n_facets = std::stoi(ya::mesh_file::section(in, {"POLYGONS", "CELLS"}).at(1));
    assert(n_facets % 2 == 0);  // a closed surface of triangles has an even number of them
    vertex_to_triangles.assign(n_vertices, {});
    triangle_to_vertices.reserve(n_facets);
    facets.reserve(n_facets);
    for (int t = 0; t < n_facets; t++) {
        int corners;
        std::array<int, 3> v;
        in >> corners >> v[0] >> v[1] >> v[2];
        assert(!in.fail() && corners == 3);
        triangle_to_vertices.push_back(v);
        facets.emplace_back(vertices[v[0]], vertices[v[1]], vertices[v[2]]);
        for (const int corner : v) vertex_to_triangles[corner].push_back(t);
    }
}

// A copy has device room of its own; like the reference's it holds nothing until copy_to_device().
inline Mesh::Mesh(const Mesh& copy)
    : vertices{copy.vertices}, facets{copy.facets}, triangle_to_vertices{copy.triangle_to_vertices},
      vertex_to_triangles{copy.vertex_to_triangles}
{
    device_room();
}

inline Mesh& Mesh::operator=(const Mesh& other)  // declared, never defined in the reference (mesh.cuh:132)
{
    if (this == &other) return *this;
    vertices = other.vertices;
    facets = other.facets;
    triangle_to_vertices = other.triangle_to_vertices;
    vertex_to_triangles = other.vertex_to_triangles;
    device_room();
    return *this;
}

inline float3 Mesh::get_minimum()
{
    float3 low = vertices[0];
    for (const auto& v : vertices) low = float3{fminf(low.x, v.x), fminf(low.y, v.y), fminf(low.z, v.z)};
    return low;
}

inline float3 Mesh::get_maximum()
{
    float3 high = vertices[0];
    for (const auto& v : vertices) high = float3{fmaxf(high.x, v.x), fmaxf(high.y, v.y), fmaxf(high.z, v.z)};
    return high;
}

inline void Mesh::translate(float3 offset)
{
    move_points([offset](float3& p) { p = p + offset; });
}

inline void Mesh::rescale(float factor)
{
    move_points([factor](float3& p) { p = p * factor; });
}

// About z, then y, then x (mesh.cuh:257-333), each a plane rotation in binary32 with the angle's
// binary32 cosine and sine: (a, b) -> (a cos - b sin, a sin + b cos).
inline void Mesh::rotate(float around_z, float around_y, float around_x)
{
    struct Turn {
        float c, s;
        explicit Turn(float angle) : c{cosf(angle)}, s{sinf(angle)} {}
        void operator()(float& a, float& b) const
        {
            const float a_old = a, b_old = b;
            a = a_old * c - b_old * s;
            b = a_old * s + b_old * c;
        }
    };
    const Turn z{around_z}, y{around_y}, x{around_x};
    move_points([&](float3& p) {
        z(p.x, p.y);
        y(p.x, p.z);
        x(p.y, p.z);
    });
    // a normal is a function of its triangle's corners alone: once, after the last turn
    for (auto& facet : facets) facet.calculate_normal();
}

// Every vertex moves by `amount` along the normalised sum of its triangles' normals (mesh.cuh:349-377);
// with `boundary`, vertices in the plane x == 0 stay.
inline void Mesh::grow_normally(float amount, bool boundary)
{
    for (size_t i = 0; i < vertices.size(); i++) {
        if (boundary && vertices[i].x == 0.f) continue;
        float3 direction{0, 0, 0};
        for (const int t : vertex_to_triangles[i]) direction = direction + facets[t].n;
        // the length as the reference takes it: squares and sum in binary64, rounded once
        const double x = direction.x, y = direction.y, z = direction.z;
        const float length = sqrt(x * x + y * y + z * z);
        vertices[i] = vertices[i] + direction * (amount / length);
    }
    for (size_t t = 0; t < facets.size(); t++) {
        const auto& v = triangle_to_vertices[t];
        facets[t] = Triangle(vertices[v[0]], vertices[v[1]], vertices[v[2]]);
    }
}

// Does the ray from P0 through P1 (and on) pierce the triangle?  Plane hit, then barycentric
// coordinates (mesh.cuh:379-406; the comparisons are the reference's, so a ray parallel to the
// plane -- NaN everywhere -- answers as it does there).
inline bool intersect(Ray R, Triangle T)
{
    const float3 along = R.P1 - R.P0;
    const float r = dot_product(T.n, T.V0 - R.P0) / dot_product(T.n, along);
    if (r < 0) return false;  // the plane lies behind the ray
    const float3 hit = R.P0 + along * r;

    const float3 u = T.V1 - T.V0, v = T.V2 - T.V0, w = hit - T.V0;
    const float uu = dot_product(u, u), uv = dot_product(u, v), vv = dot_product(v, v);
    const float wu = dot_product(w, u), wv = dot_product(w, v);
    const float denom = uv * uv - uu * vv;
    const float s = (uv * wv - vv * wu) / denom;
    if (s < 0.0 or s > 1.0) return false;
    const float t = (uv * wu - uu * wv) / denom;
    if (t < 0.0 or (s + t) > 1.0) return false;
    return true;
}

// True if `point` is OUTSIDE the closed surface: a ray in a fixed skew direction crosses it an even
// number of times (mesh.cuh:408-419).
template<typename Pt>
bool Mesh::test_exclusion(const Pt point)
{
    const float3 from{point.x, point.y, point.z};
    const Ray ray(from, from + float3{0.22788, 0.38849, 0.81499});
    int crossings = 0;
    for (const auto& facet : facets) crossings += intersect(ray, facet);
    return crossings % 2 == 0;
}

// output/<tag>.mesh.vtk: every facet with three points of its own (mesh.cuh:421-449).
inline void Mesh::write_vtk(std::string output_tag)
{
    mkdir("output", 0755);
    std::ofstream out("output/" + output_tag + ".mesh.vtk");
    assert(out.is_open());
    out << "# vtk DataFile Version 3.0\n" << output_tag << ".mesh\nASCII\nDATASET POLYDATA\n";
    out << "\nPOINTS " << 3 * facets.size() << " float\n";
    for (const auto& facet : facets)
        for (const float3& p : {facet.V0, facet.V1, facet.V2}) out << p.x << " " << p.y << " " << p.z << "\n";
    out << "\nPOLYGONS " << facets.size() << " " << 4 * facets.size() << "\n";
    for (size_t t = 0; t < facets.size(); t++) out << "3 " << 3 * t << " " << 3 * t + 1 << " " << 3 * t + 2 << "\n";
}

inline void Mesh::copy_to_device()
{
    YA_CHECK(ya_memcpy_h2d(d_vertices, vertices.data(), vertices.size() * sizeof(float3)));
}

template<typename Pt, template<typename> class Solver>
float Mesh::shape_comparison_mesh_to_points(Solution<Pt, Solver>& points)
{
    return shape_comparison((int)vertices.size(), points.get_d_n(), d_vertices, points.d_X);
}
