// Model code for the harness: the pairwise functors, frictions and generic
// forces of the BASELINE.json configurations and of the reference's own test
// cases, written once and compiled twice -- by hipcc against include/*.cuh
// (the HIP engine) and by g++ against oracle/yalla_host.hpp (the CPU
// restatement, which defines YA_ORACLE and empties __device__).  This is
// user-level model code in the sense of ya||a's examples/*.cu: it only uses
// the public header API.  Each functor cites the reference model it restates.
#pragma once

#include "polarity.cuh"

#ifdef YA_ORACLE
#define YA_MODEL_VAR static
#define YA_SET_VAR(var, value) ((var) = (value))
#else
#define YA_MODEL_VAR __device__
#define YA_SET_VAR(var, value)                                                      \
    do {                                                                            \
        auto ya_tmp_ = (value);                                                     \
        YA_CHECK((int)hipMemcpyToSymbol(HIP_SYMBOL(var), &ya_tmp_, sizeof(ya_tmp_))); \
    } while (0)
#endif

// examples/branching.cu:57 (MAKE_PT must sit at global scope)
MAKE_PT(Cell, theta, phi, u, v);

namespace models {

// --- springs: examples/springs.cu:7-21 (all-to-all springs of rest length L_0;
// under Grid_solver the cut-off at cube_size clips it, as
// tests/test_solvers.cu:44-53's clipped_spring does explicitly) ---------------
constexpr float L_0 = 0.5f;

__device__ inline float3 spring(float3 Xi, float3 r, float dist, int i, int j)
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j) return dF;
    dF = r * (L_0 - dist) / dist;
    return dF;
}

__device__ inline float3 clipped_spring(float3 Xi, float3 r, float dist, int i, int j)
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j) return dF;
    if (dist >= 1) return dF;
    dF = r * (L_0 - dist) / dist;
    return dF;
}

// A spring whose force fades to zero at the cut-off (NOT a reference model: the z-slab decomposition's
// tests use it with friction_on_background).  Nothing about such a pair changes by a jump when it crosses
// the cut-off, so a divided run has no pair "within rounding of the cut-off" to excuse a cell that differs
// from the undivided run: the comparison is strict (tests/test_slab.py, tests/fuzz_slab.py).
__device__ inline float3 fading_spring(float3 Xi, float3 r, float dist, int i, int j)
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j) return dF;
    if (dist >= 1) return dF;
    dF = r * ((L_0 - dist) * (1 - dist)) / dist;
    return dF;
}

// --- sorting: examples/sorting.cu:9-28 (differential adhesion between two cell
// types; the first half of the ids is the strongly adhering type).  n_cells is
// a compile-time constant in the example and a model parameter here. -----------
YA_MODEL_VAR unsigned sorting_n_cells = 100u;

__device__ inline float3 differential_adhesion(float3 Xi, float3 r, float dist, int i, int j)
{
    const float r_max = 1.f, r_min = 0.5f;
    float3 dF{0.f, 0.f, 0.f};
    if (i == j) return dF;
    if (dist > r_max) return dF;
    const unsigned half = sorting_n_cells / 2;
    auto strength = (1 + 2 * ((unsigned)j < half)) * (1 + 2 * ((unsigned)i < half));
    auto F = 2 * (r_min - dist) * (r_max - dist) + powf(r_max - dist, 2);
    dF = strength * r * F / dist;
    return dF;
}

// --- oscillator: tests/test_solvers.cu:8-16 (two points exchange their w) -----
__device__ inline float4 oscillator(float4 Xi, float4 r, float dist, int i, int j)
{
    float4 dF{0.f, 0.f, 0.f, 0.f};
    if (i == j) return dF;
    if (i == 0) return Xi - r;
    return -(Xi - r);
}

// --- no pairwise force: tests/test_solvers.cu:128-131, test_links.cu:7-12 -----
template<typename Pt>
__device__ inline Pt no_pw_int(Pt Xi, Pt r, float dist, int i, int j)
{
    Pt dF{0};
    return dF;
}

// --- generic force "push": tests/test_solvers.cu:133-144 sets d_dX[1] to
// (1, 0, 0) ----------------------------------------------------------------
#ifdef YA_ORACLE
template<typename Pt>
inline void push(const int n, const Pt* d_X, Pt* d_dX)
{
    d_dX[1].x = 1;
    d_dX[1].y = 0;
    d_dX[1].z = 0;
}
#else
template<typename Pt>
__global__ void push_cell(Pt* d_dX)
{
    if (blockIdx.x * blockDim.x + threadIdx.x != 0) return;
    d_dX[1].x = 1;
    d_dX[1].y = 0;
    d_dX[1].z = 0;
}
template<typename Pt>
inline void push(const int n, const Pt* d_X, Pt* d_dX)
{
    push_cell<<<1, 1>>>(d_dX);
}
#endif

// --- custom link force: tests/test_links.cu:53-59 -----------------------------
template<typename Pt>
__device__ inline void custom_force(
    const Pt* __restrict__ d_X, const int a, const int b, const float strength, Pt* d_dX)
{
    atomicAdd(&d_dX[a].w, -1.f);
    atomicAdd(&d_dX[b].w, 1.f);
}

// --- passive growth: examples/passive_growth.cu:16-91 (mesenchyme enveloped by a
// polarized epithelium; Po_cell; per-cell neighbour counters updated from inside
// the functor, one thread per cell i).  The example's compile-time constants are
// kept; its cuRAND draws are replaced by a counter-based hash so that the CPU
// oracle and the GPU make the same proliferation decisions. -----------------------
enum Cell_types { mesenchyme, epithelium };
#ifdef YA_AB_TYPE_AS_INT
YA_MODEL_VAR int* d_type;
#else
YA_MODEL_VAR Cell_types* d_type;  // (an enumeration, as in passive_growth.cu:26: loads of it do not alias the counters' stores)
#endif
YA_MODEL_VAR int* d_mes_nbs;  // number of mesenchymal neighbours
YA_MODEL_VAR int* d_epi_nbs;

__device__ inline Po_cell relu_w_epithelium(Po_cell Xi, Po_cell r, float dist, int i, int j)
{
    const auto r_max = 1;
    Po_cell dF{0};
    if (i == j) return dF;
    if (dist > r_max) return dF;

    float F;
    if (d_type[i] == d_type[j]) {
        F = fmaxf(0.7 - dist, 0) * 2 - fmaxf(dist - 0.8, 0);
    } else {
        F = fmaxf(0.8 - dist, 0) * 2 - fmaxf(dist - 0.9, 0);
    }
    dF.x = r.x * F / dist;
    dF.y = r.y * F / dist;
    dF.z = r.z * F / dist;

    if (d_type[j] == mesenchyme)
        d_mes_nbs[i] += 1;
    else
        d_epi_nbs[i] += 1;

    if (d_type[i] == mesenchyme or d_type[j] == mesenchyme) return dF;

    dF += bending_force(Xi, r, dist) * 0.15;
    return dF;
}

// Uniform in (0, 1] from (seed, step, cell, draw): lowbias32-style integer mix.
__device__ __host__ inline float hash_uniform(unsigned seed, unsigned step, unsigned i, unsigned k)
{
    unsigned x = seed ^ (step * 0x9E3779B9u) ^ (i * 0x85EBCA6Bu) ^ (k * 0xC2B2AE35u);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return ((x >> 8) + 1) * (1.0f / 16777216.0f);
}

// Division rules.  A rule says whether cell i of the n current cells divides
// this step and how daughter `d` is made from mother i; `rate` and `mean_dist`
// are model parameters.  Passive growth: examples/passive_growth.cu:60-91.
struct Passive_growth_rule {
    using Pt = Po_cell;
    __device__ __host__ static bool divides(float rate, unsigned seed, unsigned step, int i, int n,
        const Pt* X, const int* type, const int* mes_nbs, const int* epi_nbs)
    {
        if (type[i] == mesenchyme) return !(hash_uniform(seed, step, i, 0) > rate);
        return !(epi_nbs[i] > mes_nbs[i]);
    }
    __device__ __host__ static void divide(double mean_dist, unsigned seed, unsigned step, int i,
        int d, Pt* X, float3* old_v, int* type, int* mes_nbs, int* epi_nbs)
    {
        auto theta = acosf(2. * hash_uniform(seed, step, i, 1) - 1);
        auto phi = hash_uniform(seed, step, i, 2) * 2 * M_PI;
        X[d].x = X[i].x + mean_dist / 4 * sinf(theta) * cosf(phi);
        X[d].y = X[i].y + mean_dist / 4 * sinf(theta) * sinf(phi);
        X[d].z = X[i].z + mean_dist / 4 * cosf(theta);
        X[d].theta = X[i].theta;
        X[d].phi = X[i].phi;
        type[d] = type[i];
        mes_nbs[d] = 0;
        epi_nbs[d] = 0;
        old_v[d] = old_v[i];
    }
};

// --- branching: examples/branching.cu:14-170 (Turing pattern of two morphogens
// u, v on the epithelium drives mesenchymal proliferation; Cell = Po_cell + u, v;
// neighbour counters by atomicAdd).  Lineage tracing (:155-169) is bookkeeping
// outside the step path and is left out. ------------------------------------------
__device__ inline Cell epi_turing_mes_noturing(Cell Xi, Cell r, float dist, int i, int j)
{
    const auto r_max = 1.0f;
    const auto lambda = 0.0075;
    const auto D_u = 0.001;
    const auto D_v = 0.2;
    const auto f_v = 1.0;
    const auto f_u = 80.0;
    const auto g_u = 80.0;
    const auto m_u = 0.25;  // degradation rates
    const auto m_v = 0.75;
    const auto s_u = 0.05;
    Cell dF{0};

    // Meinhardt kinetics in the self-interaction
    if (i == j) {
        if (d_type[i] == epithelium) {
            dF.u = lambda * ((f_u * Xi.u * Xi.u) / (1 + f_v * Xi.v) - m_u * Xi.u + s_u);
            dF.v = lambda * (g_u * Xi.u * Xi.u - m_v * Xi.v);
            // Prevent negative values
            if (-dF.u > Xi.u) dF.u = 0.0f;
            if (-dF.v > Xi.v) dF.v = 0.0f;
        }
        return dF;
    }

    if (dist > r_max) return dF;

    float F;
    if (d_type[i] == d_type[j]) {
        F = fmaxf(0.7 - dist, 0) * 2 - fmaxf(dist - 0.8, 0);
    } else {
        F = fmaxf(0.8 - dist, 0) * 2 - fmaxf(dist - 0.9, 0);
    }
    dF.x = r.x * F / dist;
    dF.y = r.y * F / dist;
    dF.z = r.z * F / dist;

    // Diffusion
    if (d_type[i] == epithelium && d_type[j] == epithelium) {
        dF.u = -D_u * r.u;
        dF.v = -D_v * r.v;
        if (-dF.u > Xi.u) dF.u = 0.0f;
        if (-dF.v > Xi.v) dF.v = 0.0f;
        dF += bending_force(Xi, r, dist) * 0.2;
    } else {
        dF.v = -D_v * r.v;  // Diffuses into mesenchyme to induce proliferation
    }

    if (d_type[j] == epithelium)
        atomicAdd(&d_epi_nbs[i], 1);
    else
        atomicAdd(&d_mes_nbs[i], 1);

    return dF;
}

struct Branching_rule {  // examples/branching.cu:113-153
    using Pt = Cell;
    __device__ __host__ static bool divides(float, unsigned seed, unsigned step, int i, int n,
        const Pt* X, const int* type, const int* mes_nbs, const int* epi_nbs)
    {
        const auto epi_proliferation_rate = 0.2;
        const auto mes_proliferation_rate = 0.1;
        const auto prolif_threshold = 1150.0f;
        if (i >= n * (1 - epi_proliferation_rate)) return false;
        const auto rnd = hash_uniform(seed, step, i, 0);
        if (type[i] == mesenchyme) {
            if (X[i].v < prolif_threshold) return false;
            if (rnd > mes_proliferation_rate) return false;
        } else {
            if (epi_nbs[i] > 5) return false;
            if (mes_nbs[i] <= 0) return false;
            if (rnd > epi_proliferation_rate) return false;
        }
        return true;
    }
    __device__ __host__ static void divide(double mean_dist, unsigned seed, unsigned step, int i,
        int d, Pt* X, float3* old_v, int* type, int* mes_nbs, int* epi_nbs)
    {
        const float mean_distance = mean_dist;
        auto theta = acosf(2. * hash_uniform(seed, step, i, 1) - 1);
        auto phi = hash_uniform(seed, step, i, 2) * 2 * M_PI;
        X[d].x = X[i].x + mean_distance / 4 * sinf(theta) * cosf(phi);
        X[d].y = X[i].y + mean_distance / 4 * sinf(theta) * sinf(phi);
        X[d].u = X[i].u / 2;
        X[d].z = X[i].z + mean_distance / 4 * cosf(theta);
        X[i].u = X[i].u / 2;
        X[d].v = X[i].v / 2;
        X[i].v = X[i].v / 2;
        X[d].theta = X[i].theta;
        X[d].phi = X[i].phi;
        type[d] = type[i];
        old_v[d] = old_v[i];
    }
};

// proliferate, made reproducible: mothers are the cells i < n that the rule
// selects; daughters are appended in ascending mother order (the examples append
// in atomicAdd arrival order).
#ifdef YA_ORACLE
template<typename Rule>
inline void proliferate(float rate, double mean_dist, unsigned seed, unsigned step, int n,
    typename Rule::Pt* X, float3* old_v, int* d_n, int* type, int* mes_nbs, int* epi_nbs, int*,
    int n_max)
{
    std::vector<char> mother(n);
    for (int i = 0; i < n; i++)
        mother[i] = Rule::divides(rate, seed, step, i, n, X, type, mes_nbs, epi_nbs);
    int n_new = n;
    for (int i = 0; i < n; i++) {
        if (!mother[i]) continue;
        assert(n_new < n_max);
        Rule::divide(mean_dist, seed, step, i, n_new++, X, old_v, type, mes_nbs, epi_nbs);
    }
    *d_n = n_new;
}
#else
template<typename Rule>
__global__ void k_mothers(float rate, unsigned seed, unsigned step, int n,
    const typename Rule::Pt* X, const int* type, const int* mes_nbs, const int* epi_nbs, int* flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = Rule::divides(rate, seed, step, i, n, X, type, mes_nbs, epi_nbs);
}
// Exclusive prefix of the mothers' flags = the daughters' slots behind the n existing cells, in
// the order of the mothers' ids (the oracle's loop order): block sums, their prefix by one
// workgroup, then the prefix inside each block of 1024 flags.  (One workgroup walking all n
// flags took 0.21 ms per step at 10^6 cells, 6 % of config 4's step.)
__device__ __forceinline__ int block_inclusive_scan_1024(int value, int* sh)
{
    sh[threadIdx.x] = value;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    return sh[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_slot_counts(int n, const int* flag, int* block_sum)
{
    __shared__ int sh[1024];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int incl = block_inclusive_scan_1024(i < n ? flag[i] : 0, sh);
    if (threadIdx.x == 1023) block_sum[blockIdx.x] = incl;
}
__global__ __launch_bounds__(1024) void k_slot_block_offsets(int n_blocks, int* block_sum, int n, int* d_n, int n_max)
{
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int b = base + threadIdx.x;
        const int sum = b < n_blocks ? block_sum[b] : 0;
        const int incl = block_inclusive_scan_1024(sum, sh);
        if (b < n_blocks) block_sum[b] = carry + incl - sum;  // daughters of all blocks before b
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        D_ASSERT(n + carry <= n_max);
        *d_n = n + carry;
    }
}
__global__ __launch_bounds__(1024) void k_daughter_slots(int n, int* flag_to_offset, const int* block_offset)
{
    __shared__ int sh[1024];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int f = i < n ? flag_to_offset[i] : 0;
    const int incl = block_inclusive_scan_1024(f, sh);
    if (i < n) flag_to_offset[i] = f ? block_offset[blockIdx.x] + incl - f : -1;
}
template<typename Rule>
__global__ void k_daughters(double mean_dist, unsigned seed, unsigned step, int n, const int* offset,
    typename Rule::Pt* X, float3* old_v, int* type, int* mes_nbs, int* epi_nbs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || offset[i] < 0) return;
    Rule::divide(mean_dist, seed, step, i, n + offset[i], X, old_v, type, mes_nbs, epi_nbs);
}
template<typename Rule>
inline void proliferate(float rate, double mean_dist, unsigned seed, unsigned step, int n,
    typename Rule::Pt* X, float3* old_v, int* d_n, int* type, int* mes_nbs, int* epi_nbs,
    int* scratch, int n_max)
{
    if (n <= 0) return;
    const int blocks = (n + 255) / 256;
    k_mothers<Rule><<<blocks, 256>>>(rate, seed, step, n, X, type, mes_nbs, epi_nbs, scratch);
    const int slot_blocks = (n + 1023) / 1024;
    int* block_sum = scratch + n_max;  // behind the n_max flags (models_harness.inc allocates both)
    k_slot_counts<<<slot_blocks, 1024>>>(n, scratch, block_sum);
    k_slot_block_offsets<<<1, 1024>>>(slot_blocks, block_sum, n, d_n, n_max);
    k_daughter_slots<<<slot_blocks, 1024>>>(n, scratch, block_sum);
    k_daughters<Rule><<<blocks, 256>>>(
        mean_dist, seed, step, n, scratch, X, old_v, type, mes_nbs, epi_nbs);
}
#endif

}  // namespace models

// What the models say about their functors (include/solvers.cuh, YA_STATELESS; nothing on the
// oracle): pure functions of their arguments, or counting with atomicAdd (branching.cu:105-107).
// relu_w_epithelium is NOT listed: it counts with `d_mes_nbs[i] += 1` (passive_growth.cu:48-51).
#ifdef YA_STATELESS
YA_STATELESS(float3, models::spring)
YA_STATELESS(float3, models::clipped_spring)
YA_STATELESS(float3, models::fading_spring)
YA_STATELESS(float3, models::differential_adhesion)
YA_STATELESS(float3, relu_force<float3>)
YA_STATELESS(Po_cell, relu_force<Po_cell>)
YA_STATELESS(Cell, relu_force<Cell>)
YA_STATELESS(Cell, models::epi_turing_mes_noturing)
#endif
