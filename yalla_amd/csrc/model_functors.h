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
YA_MODEL_VAR int* d_type;
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

// Does cell i divide this step?  (passive_growth.cu:67-78)
__device__ __host__ inline bool pg_divides(float rate, unsigned seed, unsigned step, int i,
    const int* type, const int* mes_nbs, const int* epi_nbs)
{
    if (type[i] == mesenchyme) return !(hash_uniform(seed, step, i, 0) > rate);
    return !(epi_nbs[i] > mes_nbs[i]);
}

// Daughter n of mother i (passive_growth.cu:80-90).
__device__ __host__ inline void pg_divide(double mean_dist, unsigned seed, unsigned step, int i,
    int n, Po_cell* X, float3* old_v, int* type, int* mes_nbs, int* epi_nbs)
{
    auto theta = acosf(2. * hash_uniform(seed, step, i, 1) - 1);
    auto phi = hash_uniform(seed, step, i, 2) * 2 * M_PI;
    X[n].x = X[i].x + mean_dist / 4 * sinf(theta) * cosf(phi);
    X[n].y = X[i].y + mean_dist / 4 * sinf(theta) * sinf(phi);
    X[n].z = X[i].z + mean_dist / 4 * cosf(theta);
    X[n].theta = X[i].theta;
    X[n].phi = X[i].phi;
    type[n] = type[i];
    mes_nbs[n] = 0;
    epi_nbs[n] = 0;
    old_v[n] = old_v[i];
}

// proliferate (passive_growth.cu:60-91), made reproducible: mothers are the cells
// i < n that pg_divides() selects; daughters are appended in ascending mother
// order (the example appends in atomicAdd arrival order).
#ifdef YA_ORACLE
inline void pg_proliferate(float rate, double mean_dist, unsigned seed, unsigned step, int n,
    Po_cell* X, float3* old_v, int* d_n, int* type, int* mes_nbs, int* epi_nbs, int*, int n_max)
{
    int n_new = n;
    for (int i = 0; i < n; i++) {
        if (!pg_divides(rate, seed, step, i, type, mes_nbs, epi_nbs)) continue;
        assert(n_new < n_max);
        pg_divide(mean_dist, seed, step, i, n_new++, X, old_v, type, mes_nbs, epi_nbs);
    }
    *d_n = n_new;
}
#else
__global__ void pg_flag(float rate, unsigned seed, unsigned step, int n, const int* type,
    const int* mes_nbs, const int* epi_nbs, int* flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = pg_divides(rate, seed, step, i, type, mes_nbs, epi_nbs);
}
// exclusive prefix of the flags by ONE workgroup (model-side helper, not a hot path)
__global__ __launch_bounds__(1024) void pg_scan(int n, int* flag_to_offset, int* d_n, int n_max)
{
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int f = i < n ? flag_to_offset[i] : 0;
        sh[threadIdx.x] = f;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += v;
            __syncthreads();
        }
        const int incl = sh[threadIdx.x];
        if (i < n) flag_to_offset[i] = f ? carry + incl - f : -1;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        D_ASSERT(n + carry <= n_max);
        *d_n = n + carry;
    }
}
__global__ void pg_daughters(double mean_dist, unsigned seed, unsigned step, int n, const int* offset,
    Po_cell* X, float3* old_v, int* type, int* mes_nbs, int* epi_nbs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || offset[i] < 0) return;
    pg_divide(mean_dist, seed, step, i, n + offset[i], X, old_v, type, mes_nbs, epi_nbs);
}
inline void pg_proliferate(float rate, double mean_dist, unsigned seed, unsigned step, int n,
    Po_cell* X, float3* old_v, int* d_n, int* type, int* mes_nbs, int* epi_nbs, int* scratch,
    int n_max)
{
    const int blocks = (n + 255) / 256;
    pg_flag<<<blocks, 256>>>(rate, seed, step, n, type, mes_nbs, epi_nbs, scratch);
    pg_scan<<<1, 1024>>>(n, scratch, d_n, n_max);
    pg_daughters<<<blocks, 256>>>(mean_dist, seed, step, n, scratch, X, old_v, type, mes_nbs, epi_nbs);
}
#endif

}  // namespace models
