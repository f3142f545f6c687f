// Links between points, to simulate protrusions and other long-range
// couplings.  API parity with ya||a `include/links.cuh:16-140`: Link, Links
// (h_link, d_link, h_n, d_n, n_max, d_state, strength, set_d_n, get_d_n, reset,
// copy_to_*), Link_force<Pt>, linear_force, link_forces (two overloads).
//
// MI355X notes: 256-thread workgroups (the reference launches half-empty
// 32-thread blocks), the link count is read from the device once per call
// instead of twice, and the default force issues hardware fp32 atomics
// (global_atomic_add_f32) rather than compare-and-swap loops.
#pragma once

#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>

#include <assert.h>
#include <stdlib.h>
#include <time.h>

#include <functional>

#include "cudebug.cuh"
#include "utils.cuh"
#include "yalla_hip.h"


struct Link {
    int a, b;
};

using Check_link = std::function<bool(int a, int b)>;

inline bool every_link(int a, int b) { return true; }

class Links {
public:
    Link* h_link;
    Link* d_link;
    int* h_n = (int*)malloc(sizeof(int));
    int* d_n;
    const int n_max;
    hiprandState* d_state;  // one generator per link slot, for model kernels
    float strength;
    Links(int n_max, float strength = 1.f / 5) : n_max{n_max}, strength{strength}
    {
        h_link = (Link*)calloc(n_max, sizeof(Link));
        YA_CHECK(ya_malloc((void**)&d_link, (size_t)n_max * sizeof(Link)));
        YA_CHECK(ya_malloc((void**)&d_n, sizeof(int)));
        YA_CHECK(ya_malloc((void**)&d_state, (size_t)n_max * sizeof(hiprandState)));
        *h_n = n_max;
        set_d_n(n_max);
        // reset(): every slot starts as the inert link (0, 0)  (links.cuh:42,66-76)
        YA_CHECK(ya_memset_async(d_link, 0, (size_t)n_max * sizeof(Link), nullptr));
        auto seed = time(NULL);
        setup_rand_states<<<(n_max + 255) / 256, 256>>>(n_max, seed, d_state);
    }
    ~Links()
    {
        free(h_n);
        free(h_link);
        ya_free(d_link);
        ya_free(d_n);
        ya_free(d_state);
    }
    Links(const Links&) = delete;
    void set_d_n(int n)
    {
        assert(n <= n_max);
        YA_CHECK(ya_memcpy_h2d(d_n, &n, sizeof(int)));
    }
    int get_d_n()
    {
        int n;
        YA_CHECK(ya_get_n(d_n, &n));
        assert(n <= n_max);
        return n;
    }
    void reset(Check_link check = every_link)
    {
        copy_to_host();
        for (auto i = 0; i < n_max; i++) {
            if (!check(h_link[i].a, h_link[i].b)) continue;
            h_link[i] = Link{0, 0};
        }
        copy_to_device();
    }
    void copy_to_device()
    {
        assert(*h_n <= n_max);
        YA_CHECK(ya_memcpy_h2d(d_link, h_link, (size_t)n_max * sizeof(Link)));
        YA_CHECK(ya_memcpy_h2d(d_n, h_n, sizeof(int)));
    }
    void copy_to_host()
    {
        YA_CHECK(ya_memcpy_d2h(h_link, d_link, (size_t)n_max * sizeof(Link)));
        YA_CHECK(ya_memcpy_d2h(h_n, d_n, sizeof(int)));
        assert(*h_n <= n_max);
    }
};


template<typename Pt>
using Link_force = void(const Pt* __restrict__ d_X, const int a, const int b,
    const float strength, Pt* d_dX);

// Constant-magnitude pull along the link: dX[a] -= s r/|r|, dX[b] += s r/|r|
// with r = X[a] - X[b]  (links.cuh:98-111).
template<typename Pt>
__device__ void linear_force(const Pt* __restrict__ d_X, const int a, const int b,
    const float strength, Pt* d_dX)
{
    const Pt r = d_X[a] - d_X[b];
    const float dist = sqrtf(fmaf(r.z, r.z, fmaf(r.y, r.y, r.x * r.x)));
    const float fx = strength * r.x / dist;
    const float fy = strength * r.y / dist;
    const float fz = strength * r.z / dist;
    unsafeAtomicAdd(&d_dX[a].x, -fx);
    unsafeAtomicAdd(&d_dX[a].y, -fy);
    unsafeAtomicAdd(&d_dX[a].z, -fz);
    unsafeAtomicAdd(&d_dX[b].x, fx);
    unsafeAtomicAdd(&d_dX[b].y, fy);
    unsafeAtomicAdd(&d_dX[b].z, fz);
}

// One thread per link; inert links (a == b) are skipped (links.cuh:113-125).
template<typename Pt, Link_force<Pt> force>
__global__ __launch_bounds__(256) void link(const Pt* __restrict__ d_X, Pt* d_dX,
    const Link* __restrict__ d_link, int n_links, float strength)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_links) return;

    const Link l = d_link[i];
    if (l.a == l.b) return;

    force(d_X, l.a, l.b, strength, d_dX);
}

template<typename Pt, Link_force<Pt> force>
void link_forces(Links& links, const Pt* __restrict__ d_X, Pt* d_dX)
{
    const int n_links = links.get_d_n();
    if (n_links <= 0) return;
    link<Pt, force><<<(n_links + 255) / 256, 256>>>(
        d_X, d_dX, links.d_link, n_links, links.strength);
}

template<typename Pt>
void link_forces(Links& links, const Pt* __restrict__ d_X, Pt* d_dX)
{
    link_forces<Pt, linear_force<Pt>>(links, d_X, d_dX);
}


// Solid wall normal to z whose position is the z of the "wall node" wall_idx
// (links.cuh:142-228).  Cells closer than 1 to the wall plane are pushed by a
// relu force; the opposite force and the number of interactions are accumulated
// on the wall node, whose velocity is then averaged over its interactions.
template<typename Pt>
using Wall_force =
    void(const Pt* __restrict__ d_X, const int i, const int wall_idx, Pt* d_dX, int* d_nints);

template<typename Pt>
__device__ void xy_wall_relu_force(
    const Pt* __restrict__ d_X, const int i, const int wall_idx, Pt* d_dX, int* d_nints)
{
    const auto Xwall = d_X[wall_idx].z;
    const auto dist_wall = fabs(d_X[i].z - Xwall);
    if (dist_wall < 1.0f) {
        const auto F = fmaxf(0.8 - dist_wall, 0) - fmaxf(dist_wall - 0.8, 0);
        d_dX[i].z += F;
        atomicAdd(&d_dX[wall_idx].z, -F);
        atomicAdd(&d_nints[wall_idx], 1);
    }
}

template<typename Pt, Wall_force<Pt> force>
__global__ __launch_bounds__(256) void wall(
    const Pt* __restrict__ d_X, Pt* d_dX, int n, int wall_idx, int* d_nints)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == wall_idx) return;

    force(d_X, i, wall_idx, d_dX, d_nints);
}

template<typename Pt>
__global__ void update_wall_node(Pt* d_dX, int wall_idx, int* d_nints)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (d_nints[wall_idx] > 0) {
        const float inv = 1 / float(d_nints[wall_idx]);
        d_dX[wall_idx].x *= inv;
        d_dX[wall_idx].y *= inv;
        d_dX[wall_idx].z *= inv;
    }
}

namespace ya {
// Interaction counters of the wall node(s), one allocation per host thread reused by
// every call (the reference allocates and leaks a buffer per call, links.cuh:201-209).
// Per thread because models step from a worker std::thread while the main thread
// prepares output (examples/branching.cu:263-280): two solvers stepping in two threads
// must not share the counters.  Calls of one thread are ordered by the stream.
inline int* wall_counters(int wall_idx)
{
    static thread_local int* d_nints = nullptr;
    static thread_local int capacity = 0;
    if (wall_idx + 1 > capacity) {
        if (d_nints) ya_free(d_nints);
        capacity = wall_idx + 1 > 16 ? wall_idx + 1 : 16;
        YA_CHECK(ya_malloc((void**)&d_nints, capacity * sizeof(int)));
    }
    YA_CHECK(ya_memset_async(d_nints, 0, capacity * sizeof(int), nullptr));
    return d_nints;
}
}  // namespace ya

// Use this when there is a wall node, but no links
template<typename Pt, Wall_force<Pt> force>
void wall_forces(const int n, const Pt* __restrict__ d_X, Pt* d_dX, const int wall_idx)
{
    int* d_nints = ya::wall_counters(wall_idx);
    wall<Pt, force><<<(n + 255) / 256, 256>>>(d_X, d_dX, n, wall_idx, d_nints);
    update_wall_node<<<1, 1>>>(d_dX, wall_idx, d_nints);
}

// Use this instead of "link_forces" when there is a wall node and links
template<typename Pt, Link_force<Pt> l_force, Wall_force<Pt> w_force>
void link_wall_forces(
    Links& links, const int n, const Pt* __restrict__ d_X, Pt* d_dX, const int wall_idx)
{
    link_forces<Pt, l_force>(links, d_X, d_dX);
    wall_forces<Pt, w_force>(n, d_X, d_dX, wall_idx);
}
