// Model code for the harness: the pairwise functors, frictions and generic
// forces of the BASELINE.json configurations and of the reference's own test
// cases, written once and compiled twice -- by hipcc against include/*.cuh
// (the HIP engine) and by g++ against oracle/yalla_host.hpp (the CPU
// restatement, which defines YA_ORACLE and empties __device__).  This is
// user-level model code in the sense of ya||a's examples/*.cu: it only uses
// the public header API.  Each functor cites the reference model it restates.
#pragma once

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

}  // namespace models
