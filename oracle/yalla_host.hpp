// oracle/yalla_host.hpp -- TEST INFRASTRUCTURE ONLY.  NOT A PRODUCT PATH.
//
// Host-serial, single-threaded, plain C++ restatement (built as C++17: the slab sequencing shared with the
// product, include/slab_logic.inc, uses `if constexpr`) of the ya||a hot path
//     Solution<Pt, Solver>::take_step<pw_int, pw_friction>(dt, gen_forces)
// with Tile_solver / Grid_solver, the two-stage Heun update and
// Links::link_forces.  Every function cites the reference file:line it
// follows (paths relative to the reference checkout).  Nothing here is
// shared with the HIP engine under include/ and yalla_amd/csrc/ except the
// *user-level* functor source (yalla_amd/csrc/model_functors.h), which is
// model code, not engine code.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// include, link or execute this file.  The product fails loudly without its
// HIP libraries and never falls back to this code.
//
// Parity pinning: the reference holds no golden position vectors for this
// path (its tests are analytic known-answer tests) and its CUDA sources cannot
// be built or run in this image (no nvcc, `<<< >>>` launch syntax, Thrust and
// cuRAND).  This oracle is therefore pinned against every analytic KAT of
// tests/test_solvers.cu and tests/test_links.cu (restated in
// tests/test_oracle_kats.py), and against test_dtypes.cu's operator algebra.
//
// Arithmetic contract (so that the HIP engine can match it bit for bit on
// functors that only use + - * / sqrt fma): every statement is evaluated in
// IEEE binary32 exactly as written, with NO floating-point contraction
// (compile with -ffp-contract=off), pair distance = sqrtf(fmaf(z,z,fmaf(y,y,
// x*x))), division of a Pt by a scalar = multiplication by float(1.0/b)
// (dtypes.cuh:202-208).
//
// Summation orders.  The DEFAULTS are the reference's: Grid_computer::pwints adds the terms of all 27 cubes to
// ONE running sum per cell (solvers.cuh:437-459; Grid_computer::sum_order = YA_SUM_REFERENCE), and the
// centre-of-mass sum runs left to right (reduce_order = YA_REDUCE_SERIAL).  Each has a second, documented
// order that restates what the HIP engine does when asked to: YA_SUM_BY_PLANE (the engine's opt-in order
// for half-tile workgroups: S[own z-plane] + S[planes below and above]) and YA_REDUCE_TREE (the engine's
// fixed tree).  The engine's DEFAULT grid sum is the reference's single sum as well.
#pragma once

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <type_traits>
#include <vector>

#ifndef __device__
#define __device__
#endif
#ifndef __host__
#define __host__
#endif

#define YA_ORACLE 1

// ---------------------------------------------------------------------------
// Point types: dtypes.cuh:1-217
// ---------------------------------------------------------------------------
struct float3 {
    float x, y, z;
};
struct float4 {
    float x, y, z, w;
};

template<typename Pt>
struct Is_vector : public std::false_type {};
template<>
struct Is_vector<float3> : public std::true_type {};
template<>
struct Is_vector<float4> : public std::true_type {};

// dtypes.cuh:10-47: component-wise += and *= for float3 / float4
inline float3 operator+=(float3& a, const float3& b)
{
    a.x += b.x; a.y += b.y; a.z += b.z;
    return a;
}
inline float3 operator*=(float3& a, const float b)
{
    a.x *= b; a.y *= b; a.z *= b;
    return a;
}
inline float4 operator+=(float4& a, const float4& b)
{
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    return a;
}
inline float4 operator*=(float4& a, const float b)
{
    a.x *= b; a.y *= b; a.z *= b; a.w *= b;
    return a;
}

// dtypes.cuh:58-75: MAKE_PT(Name, fields...) = struct {float x, y, z,
// fields...} with component-wise += and *=.  All members are float, so the
// oracle applies the operation to the struct viewed as sizeof/4 floats.
template<typename Pt>
inline void ya_oracle_add(Pt& a, const Pt& b)
{
    float* pa = reinterpret_cast<float*>(&a);
    const float* pb = reinterpret_cast<const float*>(&b);
    for (size_t k = 0; k < sizeof(Pt) / sizeof(float); k++) pa[k] += pb[k];
}
template<typename Pt>
inline void ya_oracle_scale(Pt& a, const float b)
{
    float* pa = reinterpret_cast<float*>(&a);
    for (size_t k = 0; k < sizeof(Pt) / sizeof(float); k++) pa[k] *= b;
}

#define MAKE_PT(Name, ...)                                   \
    struct Name {                                            \
        float x, y, z, __VA_ARGS__;                          \
        friend Name operator+=(Name& a, const Name& b)       \
        {                                                    \
            ya_oracle_add(a, b);                             \
            return a;                                        \
        }                                                    \
        friend Name operator*=(Name& a, const float b)       \
        {                                                    \
            ya_oracle_scale(a, b);                           \
            return a;                                        \
        }                                                    \
    };                                                       \
    template<>                                               \
    struct Is_vector<Name> : public std::true_type {}

MAKE_PT(Po_cell, theta, phi);  // dtypes.cuh:147

// dtypes.cuh:150-217: + -= - unary- * / /= generated from += and *=.
template<typename Pt>
typename std::enable_if<Is_vector<Pt>::value, Pt>::type operator*(
    const Pt& a, const float b)
{
    Pt p = a;
    p *= b;
    return p;
}
template<typename Pt>
typename std::enable_if<Is_vector<Pt>::value, Pt>::type operator*(
    const float b, const Pt& a)
{
    Pt p = a;
    p *= b;
    return p;
}
template<typename Pt>
typename std::enable_if<Is_vector<Pt>::value, Pt>::type operator+(
    const Pt& a, const Pt& b)
{
    Pt s = a;
    s += b;
    return s;
}
template<typename Pt>
typename std::enable_if<Is_vector<Pt>::value, Pt>::type operator-=(
    Pt& a, const Pt& b)
{
    a += -1 * b;  // dtypes.cuh:162-166
    return a;
}
template<typename Pt>
typename std::enable_if<Is_vector<Pt>::value, Pt>::type operator-(
    const Pt& a, const Pt& b)
{
    Pt d = a;
    d -= b;
    return d;
}
template<typename Pt>
typename std::enable_if<Is_vector<Pt>::value, Pt>::type operator-(const Pt& a)
{
    return -1 * a;
}
template<typename Pt>
typename std::enable_if<Is_vector<Pt>::value, Pt>::type operator/=(
    Pt& a, const float b)
{
    a *= 1. / b;  // dtypes.cuh:204-208: double reciprocal, rounded to float
    return a;
}
template<typename Pt>
typename std::enable_if<Is_vector<Pt>::value, Pt>::type operator/(
    const Pt& a, const float b)
{
    Pt q = a;
    q /= b;
    return q;
}

// utils.cuh:22-26
template<typename Pt_a, typename Pt_b>
float dot_product(Pt_a a, Pt_b b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}

// Device intrinsics a model functor may use, restated for the host.
inline float ya_dist3(float x, float y, float z)
{
    return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}
inline float norm3df(float x, float y, float z) { return ya_dist3(x, y, z); }
inline int atomicAdd(int* p, int v)
{
    int old = *p;
    *p += v;
    return old;
}
inline float atomicAdd(float* p, float v)
{
    float old = *p;
    *p += v;
    return old;
}

// ---------------------------------------------------------------------------
// Functor signatures and defaults: solvers.cuh:15-50
// ---------------------------------------------------------------------------
template<typename Pt>
using Pairwise_interaction = Pt(Pt Xi, Pt r, float dist, int i, int j);
template<typename Pt>
using Pairwise_friction = float(Pt Xi, Pt r, float dist, int i, int j);

template<typename Pt>
float friction_w_neighbour(Pt Xi, Pt r, float dist, int i, int j)
{
    if (i == j) return 0;
    if (dist < 1) return 1;
    return 0;
}
template<typename Pt>
float friction_on_background(Pt Xi, Pt r, float dist, int i, int j)
{
    return 0;
}

template<typename Pt>
using Generic_forces = std::function<void(const int n, const Pt* d_X, Pt* d_dX)>;
template<typename Pt>
void no_gen_forces(const int n, const Pt* d_X, Pt* d_dX)
{}

// How the centre-of-mass sum is ordered.  thrust::reduce (solvers.cuh:242,268)
// leaves the order unspecified, so both are valid restatements:
//   YA_REDUCE_SERIAL  left-to-right over i = 0..n-1 (the plain reading);
//   YA_REDUCE_TREE    the fixed two-level tree the HIP engine documents in
//                     DESIGN.md ("deterministic COM reduction"), which lets
//                     tests demand bit-identical positions.
enum Ya_reduce_order { YA_REDUCE_SERIAL = 0, YA_REDUCE_TREE = 1 };

template<typename Pt>
Pt ya_oracle_reduce(const Pt* v, int n, Ya_reduce_order order)
{
    const int NW = sizeof(Pt) / sizeof(float);
    Pt out;
    float* po = reinterpret_cast<float*>(&out);
    const float* pv = reinterpret_cast<const float*>(v);
    if (order == YA_REDUCE_SERIAL) {
        for (int k = 0; k < NW; k++) po[k] = 0;
        for (int i = 0; i < n; i++)
            for (int k = 0; k < NW; k++) po[k] = po[k] + pv[(size_t)i * NW + k];
        return out;
    }
    // Tree order: B = clamp(ceil(n/256), 1, 1024) blocks of 256 lanes; lane
    // (b, t) sums i = b*256 + t, += B*256 ... serially from 0; each block folds
    // its 256 lanes by halving (s = 128 .. 1: lane[t] += lane[t+s]); then 256
    // lanes each sum block totals p = t, t+256 ... serially from 0 and fold
    // the same way.
    int B = (n + 255) / 256;
    if (B < 1) B = 1;
    if (B > 1024) B = 1024;
    std::vector<float> part((size_t)B * NW), lane(256);
    for (int k = 0; k < NW; k++) {
        for (int b = 0; b < B; b++) {
            for (int t = 0; t < 256; t++) {
                float acc = 0;
                for (long i = (long)b * 256 + t; i < n; i += (long)B * 256)
                    acc = acc + pv[(size_t)i * NW + k];
                lane[t] = acc;
            }
            for (int s = 128; s >= 1; s >>= 1)
                for (int t = 0; t < s; t++) lane[t] = lane[t] + lane[t + s];
            part[(size_t)b * NW + k] = lane[0];
        }
        for (int t = 0; t < 256; t++) {
            float acc = 0;
            for (int p = t; p < B; p += 256) acc = acc + part[(size_t)p * NW + k];
            lane[t] = acc;
        }
        for (int s = 128; s >= 1; s >>= 1)
            for (int t = 0; t < s; t++) lane[t] = lane[t] + lane[t + s];
        po[k] = lane[0];
    }
    return out;
}

// ---------------------------------------------------------------------------
// Tile_computer: solvers.cuh:279-342.  All pairs, j ascending, functor called
// for every (i, j) including i == j and beyond any cut-off; dX is added to,
// the friction sums are assigned.
// ---------------------------------------------------------------------------
template<typename Pt>
class Tile_computer {
public:
    Tile_computer(int n_max) {}

protected:
    const int* cube_order(int, const Pt*) { return nullptr; }  // no grid: renumber() is a no-op
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void pwints(const int n, const Pt* d_X, const float3* d_old_v, Pt* d_dX,
        float3* d_sum_v, float* d_sum_friction, int n_active = -1)
    {
        for (int i = 0; i < n; i++) {           // one "thread" per point
            Pt Xi = d_X[i];                      // :290-291
            Pt F;
            memset(&F, 0, sizeof(Pt));
            float3 sum_v{0, 0, 0};
            float sum_friction = 0;
            for (int j = 0; j < n; j++) {        // tiles of 32 in order: :295-316
                Pt r = Xi - d_X[j];              // :306
                float dist = ya_dist3(r.x, r.y, r.z);  // :307
                F += pw_int(Xi, r, dist, i, j);  // :308
                float friction = pw_friction(Xi, r, dist, i, j);  // :309
                sum_friction += friction;        // :310
                sum_v += friction * d_old_v[j];  // :311
            }
            d_dX[i] += F;                        // :317-321
            d_sum_friction[i] = sum_friction;
            d_sum_v[i] = sum_v;
        }
    }
};

// ---------------------------------------------------------------------------
// Grid: solvers.cuh:345-425
// ---------------------------------------------------------------------------
template<typename Pt, template<typename> class Solver>
class Solution;

class Grid {
public:
    int *d_cube_id, *d_point_id, *d_cube_start, *d_cube_end;
    Grid* d_grid;
    const int n_max, grid_size, n_cubes;
    bool out_of_range = false;  // D_ASSERT at solvers.cuh:361-362 would fire
    Grid(int n_max, int gs = 50)
        : n_max{n_max}, grid_size{gs}, n_cubes{gs * gs * gs}
    {
        d_cube_id = (int*)malloc(n_max * sizeof(int));
        d_point_id = (int*)malloc(n_max * sizeof(int));
        d_cube_start = (int*)malloc(n_cubes * sizeof(int));
        d_cube_end = (int*)malloc(n_cubes * sizeof(int));
        d_grid = this;
    }
    ~Grid()
    {
        free(d_cube_id);
        free(d_point_id);
        free(d_cube_start);
        free(d_cube_end);
    }
    Grid(const Grid&) = delete;

    // compute_cube_id, solvers.cuh:349-365: evaluated in float, in this
    // association order, then truncated to int.
    static int cube_id_of(float x, float y, float z, float cube_size, int gs)
    {
        float fx = floorf(x / cube_size) + gs / 2;
        float fy = (floorf(y / cube_size) + gs / 2) * gs;
        float fz = (floorf(z / cube_size) + gs / 2) * gs * gs;
        return static_cast<int>(fx + fy + fz);
    }

    template<typename Pt>
    void build(const int n, const Pt* d_X, const float cube_size = 1)
    {
        for (int i = 0; i < n; i++) {  // :349-365
            int id = cube_id_of(d_X[i].x, d_X[i].y, d_X[i].z, cube_size, grid_size);
            if (id < 0 || id >= n_cubes) {
                out_of_range = true;
                id = id < 0 ? 0 : n_cubes - 1;
            }
            d_cube_id[i] = id;
            d_point_id[i] = i;
        }
        std::fill(d_cube_start, d_cube_start + n_cubes, -1);  // :411
        std::fill(d_cube_end, d_cube_end + n_cubes, -2);      // :412
        // :413-414 thrust::sort_by_key = radix sort = stable
        std::vector<int> perm(n);
        for (int i = 0; i < n; i++) perm[i] = i;
        std::stable_sort(perm.begin(), perm.end(),
            [this](int a, int b) { return d_cube_id[a] < d_cube_id[b]; });
        std::vector<int> keys(n);
        for (int i = 0; i < n; i++) keys[i] = d_cube_id[perm[i]];
        for (int i = 0; i < n; i++) {
            d_cube_id[i] = keys[i];
            d_point_id[i] = perm[i];
        }
        for (int i = 0; i < n; i++) {  // :367-378
            int cube = d_cube_id[i];
            int prev = i > 0 ? d_cube_id[i - 1] : -1;
            if (cube != prev) d_cube_start[cube] = i;
            int next = i < n - 1 ? d_cube_id[i + 1] : d_cube_id[i] + 1;
            if (cube != next) d_cube_end[cube] = i;
        }
    }
    template<typename Pt, template<typename> class Solver>
    void build(Solution<Pt, Solver>& points, const float cube_size = 1)
    {
        int n = points.get_d_n();
        assert(n <= n_max);
        build(n, points.d_X, cube_size);
    }
};

static int d_nhood[27];  // solvers.cuh:428

// ---- pair trace (diagnostics, not in the reference) ----------------------------------------------------
// For a handful of cells (by the id the functor sees: the global id in a z-slab decomposition) every
// candidate pair whose distance is below cube_size + margin is written down with the number of the
// pwints call it was tested in (a solver's calls count from 0: step k's two stages are calls 2k - 2 and
// 2k - 1) -- so that two runs of one system (undivided / in slabs) can be compared pair by pair where a
// cell's position first differs: tools/diag/slab_case.py.  Off unless armed (oracle_models.cpp:
// ya_oracle_trace_*); several slabs' host threads may record at once.
struct Pair_trace {
    struct Rec {
        int call, i, j;
        float dist;
    };
    std::mutex lock;
    std::vector<int> ids;  // sorted
    float margin = 0.f;
    std::vector<Rec> recs;
    std::atomic<bool> armed{false};
    bool wants(int id) const { return std::binary_search(ids.begin(), ids.end(), id); }
    void add(int call, int i, int j, float dist)
    {
        std::lock_guard<std::mutex> hold(lock);
        recs.push_back(Rec{call, i, j, dist});
    }
};
inline Pair_trace& pair_trace()
{
    static Pair_trace t;
    return t;
}

// The order in which Grid_computer::pwints adds a cell's pair terms:
//   YA_SUM_REFERENCE  one running sum over the 27 cubes in d_nhood order, cells of a cube in ascending
//                     slot order -- the reference's thread (solvers.cuh:437-459).  THE DEFAULT.
//   YA_SUM_BY_PLANE   the terms of the cell's own z-plane (stencil entries 0-8) and those of the planes
//                     below and above (9-26) summed separately, each in the reference's order from +0, and
//                     the two sums added: what the HIP engine computes when a model opts into
//                     Grid_computer::sum_order = YA_SUM_BY_PLANE so that a tile's planes may go to two
//                     wavefronts (include/solvers.cuh, "the tail").  ~1e-7 relative per step beside the
//                     reference's sum (tests/test_sum_order_gpu.py measures it).
enum Ya_sum_order { YA_SUM_REFERENCE = 0, YA_SUM_BY_PLANE = 1 };

// Grid_computer: solvers.cuh:430-502
template<typename Pt>
class Grid_computer {
public:
    float cube_size;
    Ya_sum_order sum_order = YA_SUM_REFERENCE;
    Grid_computer(int n_max, int grid_size = 50, float cube_size = 1)
        : cube_size{cube_size}, grid{n_max, grid_size}
    {
        int h[27];  // :472-483
        h[0] = -1;
        h[1] = 0;
        h[2] = 1;
        for (int i = 0; i < 3; i++) {
            h[i + 3] = h[i % 3] - grid_size;
            h[i + 6] = h[i % 3] + grid_size;
        }
        for (int i = 0; i < 9; i++) {
            h[i + 9] = h[i % 9] - grid_size * grid_size;
            h[i + 18] = h[i % 9] + grid_size * grid_size;
        }
        memcpy(d_nhood, h, sizeof(h));
        memcpy(nhood, h, sizeof(h));
    }
    Grid grid;  // public in the oracle so tests can read the four arrays
    // z-slab decomposition (not in the reference): local index -> global id for the functors
    const int* d_global_id = nullptr;
    int pwints_calls = 0;  // (pair trace: which call a pair was tested in)

protected:
    int nhood[27];
    // Heun_solver::renumber (not in the reference): the cells' ids in (cube, id) order
    const int* cube_order(int n, const Pt* d_X)
    {
        grid.build(n, d_X, cube_size);
        return grid.d_point_id;
    }
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void pwints(int n, const Pt* d_X, const float3* d_old_v, Pt* d_dX,
        float3* d_sum_v, float* d_sum_friction, int n_active = -1)
    {
        if (n_active < 0) n_active = n;
        grid.build(n, d_X, cube_size);  // :494
        Pair_trace& trace = pair_trace();
        const bool tracing = trace.armed.load();
        const int call = pwints_calls++;
        for (int i = 0; i < n; i++) {   // thread i owns sorted slot i: :430-463
            int pi = grid.d_point_id[i];
            if (pi >= n_active) continue;  // ghost cell of a slab decomposition (not in the reference)
            const bool traced = tracing && trace.wants(d_global_id ? d_global_id[pi] : pi);
            Pt Xi = d_X[pi];
            Pt F;
            memset(&F, 0, sizeof(Pt));
            float3 sum_v{0, 0, 0};
            float sum_friction = 0;
            // YA_SUM_BY_PLANE only (not the reference): the own plane's sums are set aside when the walk
            // reaches stencil entry 9, and added to the other planes' at the end.
            Pt F_own;
            memset(&F_own, 0, sizeof(Pt));
            float3 sum_v_own{0, 0, 0};
            float sum_friction_own = 0;
            for (int j = 0; j < 27; j++) {
                if (j == 9 && sum_order == YA_SUM_BY_PLANE) {
                    F_own = F, sum_v_own = sum_v, sum_friction_own = sum_friction;
                    memset(&F, 0, sizeof(Pt));
                    sum_v = float3{0, 0, 0};
                    sum_friction = 0;
                }
                int cube = grid.d_cube_id[i] + nhood[j];
                // The reference reads out of bounds here if a cell sits in the
                // grid's outermost layer; the oracle treats such cubes as empty.
                if (cube < 0 || cube >= grid.n_cubes) continue;
                for (int k = grid.d_cube_start[cube]; k <= grid.d_cube_end[cube]; k++) {
                    int pk = grid.d_point_id[k];
                    Pt r = Xi - d_X[pk];
                    float dist = ya_dist3(r.x, r.y, r.z);
                    if (traced && dist < cube_size + trace.margin)
                        trace.add(call, d_global_id ? d_global_id[pi] : pi, d_global_id ? d_global_id[pk] : pk, dist);
                    if (dist >= cube_size) continue;  // :450
                    const int gi = d_global_id ? d_global_id[pi] : pi;
                    const int gk = d_global_id ? d_global_id[pk] : pk;
                    F += pw_int(Xi, r, dist, gi, gk);
                    float friction = pw_friction(Xi, r, dist, gi, gk);
                    sum_friction += friction;
                    sum_v += friction * d_old_v[pk];
                }
            }
            if (sum_order == YA_SUM_BY_PLANE) {
                F = F_own + F;
                sum_v = sum_v_own + sum_v;
                sum_friction = sum_friction_own + sum_friction;
            }
            d_dX[pi] += F;              // :460
            d_sum_v[pi] = sum_v;        // :461
            d_sum_friction[pi] = sum_friction;  // :462
        }
    }
};

// ---------------------------------------------------------------------------
// Heun_solver: solvers.cuh:109-276
// ---------------------------------------------------------------------------
template<typename Pt, template<typename> class Computer>
class Heun_solver : public Computer<Pt> {
public:
    template<typename... Args>
    Heun_solver(int n_max, Args... args) : Computer<Pt>{n_max, args...}, n_max{n_max}
    {
        d_X = (Pt*)calloc(n_max, sizeof(Pt));
        d_dX = (Pt*)calloc(n_max, sizeof(Pt));
        d_X1 = (Pt*)calloc(n_max, sizeof(Pt));
        d_dX1 = (Pt*)calloc(n_max, sizeof(Pt));
        d_old_v = (float3*)calloc(n_max, sizeof(float3));  // zero-filled: :177
        d_sum_v = (float3*)calloc(n_max, sizeof(float3));
        d_sum_friction = (float*)calloc(n_max, sizeof(float));
        d_n = (int*)calloc(1, sizeof(int));
    }
    ~Heun_solver()
    {
        free(d_X); free(d_dX); free(d_X1); free(d_dX1);
        free(d_old_v); free(d_sum_v); free(d_sum_friction); free(d_n);
    }
    Heun_solver(const Heun_solver&) = delete;
    void set_fixed() { fix_com = true; }  // :196 (does not clear fix_com_z)
    void set_fixed(int point_id)          // :197-201
    {
        fix_com = false;
        fix_point = point_id;
    }
    void set_fixed_xy(int point_id)       // :203-208
    {
        fix_com = false;
        fix_com_z = true;
        fix_point = point_id;
    }
    Ya_reduce_order reduce_order = YA_REDUCE_SERIAL;

    // Not in the reference: the HIP engine's opt-in Solution::renumber (include/solvers.cuh,
    // "renumbering"), restated.  Cell s becomes what cell order[s] was, order = the point ids of a
    // fresh grid build (cube by cube, ascending id inside a cube); d_X, d_old_v and every array
    // handed over (Property: d_prop; Links: endpoints renamed; plain pointers) follow.
    template<typename... Arrays>
    void renumber(Arrays&... arrays)
    {
        const int n = get_d_n();
        if (n <= 0) return;
        const int* order = Computer<Pt>::cube_order(n, d_X);
        if (!order) return;
        const std::vector<int> fixed(order, order + n);  // (the grid's array is rebuilt by the next step)
        permute_array(fixed, d_X);
        permute_array(fixed, d_old_v);
        const int unused[] = {0, (renumber_one(fixed, arrays), 0)...};
        (void)unused;
    }

    // ... registered once (include/solvers.cuh, keep_in_cube_order): every `every`-th take_step, the next
    // one first, begins with renumber(arrays...)
    template<typename... Arrays>
    void keep_in_cube_order(const int every, Arrays&... arrays)
    {
        keep_order_every = every;
        keep_order_wait = 0;
        if (every > 0)
            keep_order = [this, &arrays...]() { this->renumber(arrays...); };
        else
            keep_order = nullptr;
    }
    std::function<void()> keep_order;
    int keep_order_every = 0, keep_order_wait = 0;

protected:
    template<typename T>
    static void permute_array(const std::vector<int>& order, T* array)
    {
        std::vector<T> tmp(order.size());
        for (size_t s = 0; s < order.size(); s++) tmp[s] = array[order[s]];
        for (size_t s = 0; s < order.size(); s++) array[s] = tmp[s];
    }
    template<typename T>
    static void renumber_one(const std::vector<int>& order, T*& array) { permute_array(order, array); }
    template<typename A>
    static auto renumber_one(const std::vector<int>& order, A& property) -> decltype((void)property.d_prop)
    {
        permute_array(order, property.d_prop);
    }
    template<typename L>
    static auto renumber_one(const std::vector<int>& order, L& links) -> decltype((void)links.d_link)
    {
        const int n = (int)order.size();
        std::vector<int> new_id(n);
        for (int s = 0; s < n; s++) new_id[order[s]] = s;
        for (int k = 0; k < *links.d_n && k < links.n_max; k++) {
            if (links.d_link[k].a >= 0 && links.d_link[k].a < n) links.d_link[k].a = new_id[links.d_link[k].a];
            if (links.d_link[k].b >= 0 && links.d_link[k].b < n) links.d_link[k].b = new_id[links.d_link[k].b];
        }
    }
    Pt *d_X, *d_dX, *d_X1, *d_dX1;
    float3 *d_old_v, *d_sum_v;
    float* d_sum_friction;
    int* d_n;
    bool fix_com = true;
    bool fix_com_z = false;
    int fix_point = 0;
    const int n_max;
    int get_d_n()
    {
        int n = *d_n;
        assert(n <= n_max);
        return n;
    }

    // add_rhs, :146-161
    void add_rhs(int n, const float3* sum_v, const float* sum_friction, Pt* dX)
    {
        for (int i = 0; i < n; i++) {
            if (sum_friction[i] > 0) {
                dX[i].x += sum_v[i].x / sum_friction[i];
                dX[i].y += sum_v[i].y / sum_friction[i];
                dX[i].z += sum_v[i].z / sum_friction[i];
            }
        }
    }

    // The pieces of one stage, for the z-slab decomposition (not in the reference,
    // which is single-GPU): n = own + ghost cells, n_active = own cells.  They
    // restate exactly the statements of take_step below, stage by stage.
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void stage_rhs(int stage, int n, int n_active, Generic_forces<Pt>& gen_forces)
    {
        Pt* in = stage == 1 ? d_X : d_X1;
        Pt* rhs = stage == 1 ? d_dX : d_dX1;
        memset(rhs, 0, (size_t)n * sizeof(Pt));
        memset(d_sum_friction, 0, (size_t)n * sizeof(float));
        memset(d_sum_v, 0, (size_t)n * sizeof(float3));
        gen_forces(n, in, rhs);
        Computer<Pt>::template pwints<pw_int, pw_friction>(
            n, in, d_old_v, rhs, d_sum_v, d_sum_friction, n_active);
        add_rhs(n_active, d_sum_v, d_sum_friction, rhs);
    }
    Pt stage_sum(int stage, int n)
    {
        return ya_oracle_reduce(stage == 1 ? d_dX : d_dX1, n, reduce_order);
    }
    void stage_update(int stage, int n, float dt, const float* fix)
    {
        for (int i = 0; i < n; i++) {
            if (stage == 1) {  // euler_step :113-125
                d_dX[i].x -= fix[0];
                d_dX[i].y -= fix[1];
                d_dX[i].z -= fix[2];
                d_X1[i] = d_X[i] + d_dX[i] * dt;
            } else {  // heun_step :127-144
                d_dX1[i].x -= fix[0];
                d_dX1[i].y -= fix[1];
                d_dX1[i].z -= fix[2];
                d_X[i] += (d_dX[i] + d_dX1[i]) * 0.5 * dt;
                d_old_v[i].x = (d_dX[i].x + d_dX1[i].x) * 0.5;
                d_old_v[i].y = (d_dX[i].y + d_dX1[i].y) * 0.5;
                d_old_v[i].z = (d_dX[i].z + d_dX1[i].z) * 0.5;
            }
        }
    }

    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void take_step(float dt, Generic_forces<Pt> gen_forces)
    {
        if (keep_order && keep_order_wait-- <= 0) {  // keep_in_cube_order (not in the reference)
            keep_order();
            keep_order_wait = keep_order_every - 1;
        }
        int n = get_d_n();  // :229

        // 1st stage, :231-255
        memset(d_dX, 0, (size_t)n * sizeof(Pt));
        memset(d_sum_friction, 0, (size_t)n * sizeof(float));
        memset(d_sum_v, 0, (size_t)n * sizeof(float3));
        gen_forces(n, d_X, d_dX);
        Computer<Pt>::template pwints<pw_int, pw_friction>(
            n, d_X, d_old_v, d_dX, d_sum_v, d_sum_friction);
        add_rhs(n, d_sum_v, d_sum_friction, d_dX);
        Pt fix_dX;
        if (fix_com or fix_com_z) {
            fix_dX = ya_oracle_reduce(d_dX, n, reduce_order) / n;  // :242
            if (fix_com_z) {
                Pt temp = d_dX[fix_point];
                fix_dX.x = temp.x;
                fix_dX.y = temp.y;
            }
        } else {
            fix_dX = d_dX[fix_point];
        }
        for (int i = 0; i < n; i++) {  // euler_step :113-125
            d_dX[i].x -= fix_dX.x;
            d_dX[i].y -= fix_dX.y;
            d_dX[i].z -= fix_dX.z;
            d_X1[i] = d_X[i] + d_dX[i] * dt;
        }

        // 2nd stage, :257-274
        memset(d_dX1, 0, (size_t)n * sizeof(Pt));
        memset(d_sum_friction, 0, (size_t)n * sizeof(float));
        memset(d_sum_v, 0, (size_t)n * sizeof(float3));
        gen_forces(n, d_X1, d_dX1);
        Computer<Pt>::template pwints<pw_int, pw_friction>(
            n, d_X1, d_old_v, d_dX1, d_sum_v, d_sum_friction);
        add_rhs(n, d_sum_v, d_sum_friction, d_dX1);
        Pt fix_dX1;
        if (fix_com) {
            fix_dX1 = ya_oracle_reduce(d_dX1, n, reduce_order) / n;  // :268
        } else {
            fix_dX1 = d_dX1[fix_point];  // also the set_fixed_xy case: :269-272
        }
        for (int i = 0; i < n; i++) {  // heun_step :127-144
            d_dX1[i].x -= fix_dX1.x;
            d_dX1[i].y -= fix_dX1.y;
            d_dX1[i].z -= fix_dX1.z;
            d_X[i] += (d_dX[i] + d_dX1[i]) * 0.5 * dt;
            d_old_v[i].x = (d_dX[i].x + d_dX1[i].x) * 0.5;
            d_old_v[i].y = (d_dX[i].y + d_dX1[i].y) * 0.5;
            d_old_v[i].z = (d_dX[i].z + d_dX1[i].z) * 0.5;
        }
    }
};

template<typename Pt>
using Tile_solver = Heun_solver<Pt, Tile_computer>;
template<typename Pt>
using Grid_solver = Heun_solver<Pt, Grid_computer>;

// ---------------------------------------------------------------------------
// Solution facade: solvers.cuh:56-106.  "Device" memory is host memory here.
// ---------------------------------------------------------------------------
template<typename Pt, template<typename> class Solver>
class Solution : public Solver<Pt> {
public:
    Pt* h_X;
    Pt* const d_X = Solver<Pt>::d_X;
    float3* const d_old_v = Solver<Pt>::d_old_v;
    int* const h_n = (int*)malloc(sizeof(int));
    int* const d_n = Solver<Pt>::d_n;
    const int n_max;
    template<typename... Args>
    Solution(int n_max, Args... args) : Solver<Pt>{n_max, args...}, n_max{n_max}
    {
        *h_n = n_max;
        h_X = (Pt*)calloc(n_max, sizeof(Pt));
    }
    ~Solution()
    {
        free(h_X);
        free(h_n);
    }
    void copy_to_device()  // :80-85: n_max elements, then n
    {
        assert(*h_n <= n_max);
        memcpy(d_X, h_X, (size_t)n_max * sizeof(Pt));
        *d_n = *h_n;
    }
    void copy_to_host()  // :86-91
    {
        memcpy(h_X, d_X, (size_t)n_max * sizeof(Pt));
        *h_n = *d_n;
        assert(*h_n <= n_max);
    }
    int get_d_n() { return Solver<Pt>::get_d_n(); }
    template<Pairwise_interaction<Pt> pw_int>
    void take_step(float dt, Generic_forces<Pt> gen_forces = no_gen_forces<Pt>)
    {
        return Solver<Pt>::template take_step<pw_int, friction_w_neighbour<Pt>>(
            dt, gen_forces);
    }
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void take_step(float dt, Generic_forces<Pt> gen_forces = no_gen_forces<Pt>)
    {
        return Solver<Pt>::template take_step<pw_int, pw_friction>(dt, gen_forces);
    }
};

// Solution<Pt, n_max, Solver> under the name the engine's headers give it (include/solvers.cuh).
template<typename Pt, int N_MAX, template<typename> class Solver>
class Solution_n : public Solution<Pt, Solver> {
public:
    static constexpr int capacity = N_MAX;
    template<typename... Args>
    Solution_n(Args... args) : Solution<Pt, Solver>{N_MAX, args...}
    {}
};

// ---------------------------------------------------------------------------
// Property: property.cuh:1-34 ("device" copy = a second host array)
// ---------------------------------------------------------------------------
template<typename Prop = int>
struct Property {
    Prop* h_prop;
    Prop* d_prop;
    const int n_max;
    Property(int n_max) : n_max{n_max}
    {
        h_prop = (Prop*)calloc(n_max, sizeof(Prop));
        d_prop = (Prop*)calloc(n_max, sizeof(Prop));
    }
    ~Property()
    {
        free(h_prop);
        free(d_prop);
    }
    Property(const Property&) = delete;
    void copy_to_device() { memcpy(d_prop, h_prop, (size_t)n_max * sizeof(Prop)); }
    void copy_to_host() { memcpy(h_prop, d_prop, (size_t)n_max * sizeof(Prop)); }
};

// ---------------------------------------------------------------------------
// Links and link_forces: links.cuh:16-140
// ---------------------------------------------------------------------------
struct Link {
    int a, b;
};

class Links {
public:
    Link* h_link;
    Link* d_link;
    int* h_n = (int*)malloc(sizeof(int));
    int* d_n;
    const int n_max;
    float strength;
    Links(int n_max, float strength = 1.f / 5) : n_max{n_max}, strength{strength}
    {
        h_link = (Link*)calloc(n_max, sizeof(Link));
        d_link = (Link*)calloc(n_max, sizeof(Link));  // reset(): all (0, 0)
        d_n = (int*)malloc(sizeof(int));
        *h_n = n_max;
        *d_n = n_max;
    }
    ~Links()
    {
        free(h_n); free(h_link); free(d_link); free(d_n);
    }
    Links(const Links&) = delete;
    void set_d_n(int n)
    {
        assert(n <= n_max);
        *d_n = n;
    }
    int get_d_n()
    {
        assert(*d_n <= n_max);
        return *d_n;
    }
    void copy_to_device()
    {
        assert(*h_n <= n_max);
        memcpy(d_link, h_link, (size_t)n_max * sizeof(Link));
        *d_n = *h_n;
    }
    void copy_to_host()
    {
        memcpy(h_link, d_link, (size_t)n_max * sizeof(Link));
        *h_n = *d_n;
    }
};

template<typename Pt>
using Link_force = void(
    const Pt* d_X, const int a, const int b, const float strength, Pt* d_dX);

// linear_force, links.cuh:98-111.  The six atomicAdds become plain adds in
// link order (the device order is unspecified).
template<typename Pt>
void linear_force(const Pt* d_X, const int a, const int b, const float strength, Pt* d_dX)
{
    Pt r = d_X[a] - d_X[b];
    float dist = ya_dist3(r.x, r.y, r.z);
    d_dX[a].x += -strength * r.x / dist;
    d_dX[a].y += -strength * r.y / dist;
    d_dX[a].z += -strength * r.z / dist;
    d_dX[b].x += strength * r.x / dist;
    d_dX[b].y += strength * r.y / dist;
    d_dX[b].z += strength * r.z / dist;
}

// link kernel + link_forces, links.cuh:113-140
template<typename Pt, Link_force<Pt> force>
void link_forces(Links& links, const Pt* d_X, Pt* d_dX)
{
    int n_links = links.get_d_n();
    for (int i = 0; i < n_links; i++) {
        int a = links.d_link[i].a;
        int b = links.d_link[i].b;
        if (a == b) continue;
        force(d_X, a, b, links.strength, d_dX);
    }
}
template<typename Pt>
void link_forces(Links& links, const Pt* d_X, Pt* d_dX)
{
    link_forces<Pt, linear_force<Pt>>(links, d_X, d_dX);
}

// ---------------------------------------------------------------------------
// random_sphere: inits.cuh:33-51, with the seed made a parameter (the
// reference seeds glibc rand() from std::random_device).  Per cell: three
// rand() draws in the order r, theta, phi; double arithmetic; stored as float.
// ---------------------------------------------------------------------------
template<typename Pt>
void ya_oracle_random_sphere(float dist_to_nb, Pt* h_X, int n, unsigned seed, unsigned n_0 = 0)
{
    srand(seed);
    auto r_max = pow((n - n_0) / 0.64, 1. / 3) * dist_to_nb / 2;
    for (auto i = n_0; i < (unsigned)n; i++) {
        auto r = r_max * pow(rand() / (RAND_MAX + 1.), 1. / 3);
        auto theta = acos(2. * rand() / (RAND_MAX + 1.) - 1);
        auto phi = rand() / (RAND_MAX + 1.) * 2 * M_PI;
        h_X[i].x = r * sin(theta) * cos(phi);
        h_X[i].y = r * sin(theta) * sin(phi);
        h_X[i].z = r * cos(theta);
    }
}

// relu_force: inits.cuh:78-93
template<typename Pt>
Pt relu_force(Pt Xi, Pt r, float dist, int i, int j)
{
    Pt dF;
    memset(&dF, 0, sizeof(Pt));
    if (i == j) return dF;
    if (dist > 1.f) return dF;
    auto F = fmaxf(0.8f - dist, 0) * 2.f - fmaxf(dist - 0.8f, 0);
    dF.x = r.x * F / dist;
    dF.y = r.y * F / dist;
    dF.z = r.z * F / dist;
    return dF;
}
