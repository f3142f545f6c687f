// Links between points, to simulate protrusions and other long-range
// couplings.  API parity with ya||a `include/links.cuh:16-140`: Link, Links
// (h_link, d_link, h_n, d_n, n_max, d_state, strength, set_d_n, get_d_n, reset,
// copy_to_*), Link_force<Pt>, linear_force, link_forces (two overloads).
//
// MI355X notes: 256-thread workgroups (the reference launches half-empty
// 32-thread blocks), the link count is read by the kernel from device memory
// (no host round trip per call), and the default force issues hardware fp32
// atomics (global_atomic_add_f32) rather than compare-and-swap loops.
#pragma once

#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>

#include <assert.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <cstring>
#include <functional>
#include <type_traits>

#include <rocprim/device/device_radix_sort.hpp>

#include "cudebug.cuh"
#include "utils.cuh"
#include "yalla_hip.h"


struct Link {
    int a, b;
};

namespace ya {
// Buffers of the atomics-free link_forces path (below), owned by a Links object and sized
// for its n_max slots on first use.
struct Link_scratch {
    float3* d_force = nullptr;                          // force of every link slot
    unsigned *d_key = nullptr, *d_key_sorted = nullptr;  // endpoint cell of entry 2i (a), 2i + 1 (b)
    unsigned *d_entry = nullptr, *d_entry_sorted = nullptr;
    void* d_temp = nullptr;
    size_t temp_bytes = 0;
    void release()
    {
        ya_free(d_force);
        ya_free(d_key);
        ya_free(d_key_sorted);
        ya_free(d_entry);
        ya_free(d_entry_sorted);
        ya_free(d_temp);
        d_force = nullptr;
        d_key = d_key_sorted = d_entry = d_entry_sorted = nullptr;
        d_temp = nullptr;
        temp_bytes = 0;
    }
};
}  // namespace ya

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
        scratch.release();
    }
    Links(const Links&) = delete;
    ya::Link_scratch scratch;  // engine-private (not in the reference)
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

// One thread per link slot; inert links (a == b) are skipped (links.cuh:113-125).  The
// number of links in use is read from links.d_n ON THE DEVICE (the launch covers n_max
// slots): the reference's two blocking 4-byte reads per call (links.cuh:130-133) cost more
// than the kernel itself, and model kernels may change *d_n right before.
template<typename Pt, Link_force<Pt> force>
__global__ __launch_bounds__(256) void link(const Pt* __restrict__ d_X, Pt* d_dX,
    const Link* __restrict__ d_link, const int* __restrict__ d_n_links, int n_max, float strength)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_links = *d_n_links;
    D_ASSERT(n_links <= n_max);
    if (i >= n_links || i >= n_max) return;

    const Link l = d_link[i];
    if (l.a == l.b) return;

    force(d_X, l.a, l.b, strength, d_dX);
}

// ---- linear_force without atomics --------------------------------------------------
// On MI355X a floating-point atomic on global memory is served by the memory side of the
// fabric, not by an XCD's L2 (eight L2s that are not coherent with each other): measured,
// the one-thread-per-link kernel above sustains ~2 x 10^10 atomics per second -- 91 us for
// 300 000 links, more than the whole grid force kernel of the 100 000 cells they connect.
// For the default force the engine therefore does a segmented sum instead:
//   1. link_linear_eval: one thread per link slot computes the link's force once and emits
//      two (cell, entry) pairs, entry 2i for endpoint a (gets -f) and 2i + 1 for b (gets +f);
//      inert or unused slots emit a key that sorts last;
//   2. a stable radix sort of the pairs by cell (rocPRIM, all 32 key bits);
//   3. link_linear_apply: the first entry of every cell's segment sums the segment in sorted
//      (= link slot) order and adds it to d_dX with a plain store.  Segments are cut every
//      256 entries so that a hub cell cannot serialise a thread; only pieces of such cut
//      segments use atomics.
// No host round trip and results that repeat run to run (the atomics' order does not).
// Used from YA_LINKS_SEGMENTED_MIN link slots up (see there).  Custom Link_force functors
// do their own atomics on d_dX and keep the one-thread-per-link kernel.
namespace ya {
// All 32 key bits are sorted: cell ids are ints, so every id a model can hold is below the dead
// key (a 24-bit sort was one radix pass cheaper, but ids from 2^24 - 1 up -- a 75 M-cell system
// has them -- would have been merged with other cells or dropped).
constexpr unsigned LINK_DEAD_KEY = 0xFFFFFFFFu;
constexpr int LINK_KEY_BITS = 32;
constexpr int LINK_SEGMENT_CUT = 256;

template<typename Pt>
__global__ __launch_bounds__(256) void link_linear_eval(const Pt* __restrict__ d_X,
    const Link* __restrict__ d_link, const int* __restrict__ d_n_links, const int n_max,
    const float strength, float3* __restrict__ force, unsigned* __restrict__ key,
    unsigned* __restrict__ entry)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_max) return;
    const int n_links = *d_n_links;
    D_ASSERT(n_links <= n_max);
    Link l{0, 0};
    if (i < n_links) l = d_link[i];
    unsigned key_a = LINK_DEAD_KEY, key_b = LINK_DEAD_KEY;
    if (l.a != l.b) {
        D_ASSERT(l.a >= 0 && l.b >= 0);
        const Pt r = d_X[l.a] - d_X[l.b];
        const float dist = sqrtf(fmaf(r.z, r.z, fmaf(r.y, r.y, r.x * r.x)));
        force[i] = float3{strength * r.x / dist, strength * r.y / dist, strength * r.z / dist};
        key_a = (unsigned)l.a;
        key_b = (unsigned)l.b;
    }
    key[2 * i] = key_a;
    key[2 * i + 1] = key_b;
    entry[2 * i] = 2u * i;
    entry[2 * i + 1] = 2u * i + 1u;
}

template<typename Pt>
__global__ __launch_bounds__(256) void link_linear_apply(const unsigned* __restrict__ key,
    const unsigned* __restrict__ entry, const float3* __restrict__ force, const int n_entries,
    Pt* d_dX)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_entries) return;
    const unsigned cell = key[t];
    if (cell == LINK_DEAD_KEY) return;
    const bool continues = t > 0 && key[t - 1] == cell;   // the segment began before t
    if (continues && t % LINK_SEGMENT_CUT != 0) return;   // neither a segment head nor a cut
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int u = t;
    do {
        const unsigned e = entry[u];
        const float3 f = force[e >> 1];
        const float sign = (e & 1u) ? 1.f : -1.f;  // endpoint a: dX -= f, endpoint b: dX += f
        sx += sign * f.x;
        sy += sign * f.y;
        sz += sign * f.z;
        u++;
    } while (u < n_entries && key[u] == cell && u % LINK_SEGMENT_CUT != 0);
    const bool cut_after = u < n_entries && key[u] == cell;
    if (continues || cut_after) {  // a piece of a segment longer than the cut
        unsafeAtomicAdd(&d_dX[cell].x, sx);
        unsafeAtomicAdd(&d_dX[cell].y, sy);
        unsafeAtomicAdd(&d_dX[cell].z, sz);
    } else {
        d_dX[cell].x += sx;
        d_dX[cell].y += sy;
        d_dX[cell].z += sz;
    }
}

template<typename Pt>
void link_forces_segmented(Links& links, const Pt* __restrict__ d_X, Pt* d_dX)
{
    Link_scratch& s = links.scratch;
    const int n_entries = 2 * links.n_max;
    if (!s.d_force) {
        const size_t words = (size_t)n_entries * sizeof(unsigned);
        YA_CHECK(ya_malloc((void**)&s.d_force, (size_t)links.n_max * sizeof(float3)));
        YA_CHECK(ya_malloc((void**)&s.d_key, words));
        YA_CHECK(ya_malloc((void**)&s.d_key_sorted, words));
        YA_CHECK(ya_malloc((void**)&s.d_entry, words));
        YA_CHECK(ya_malloc((void**)&s.d_entry_sorted, words));
        YA_CHECK((int)rocprim::radix_sort_pairs(nullptr, s.temp_bytes, s.d_key, s.d_key_sorted, s.d_entry,
            s.d_entry_sorted, (size_t)n_entries, 0, LINK_KEY_BITS, (hipStream_t) nullptr));
        YA_CHECK(ya_malloc(&s.d_temp, s.temp_bytes > 0 ? s.temp_bytes : 16));
    }
    const int blocks = (links.n_max + 255) / 256;
    link_linear_eval<Pt><<<blocks, 256>>>(
        d_X, links.d_link, links.d_n, links.n_max, links.strength, s.d_force, s.d_key, s.d_entry);
    YA_CHECK((int)rocprim::radix_sort_pairs(s.d_temp, s.temp_bytes, s.d_key, s.d_key_sorted, s.d_entry,
        s.d_entry_sorted, (size_t)n_entries, 0, LINK_KEY_BITS, (hipStream_t) nullptr));
    link_linear_apply<Pt><<<(n_entries + 255) / 256, 256>>>(
        s.d_key_sorted, s.d_entry_sorted, s.d_force, n_entries, d_dX);
}
}  // namespace ya

// The sort is ~10 launches of rocPRIM kernels: measured on MI355X the segmented path is
// slower than the atomics below ~10^6 link slots (0.53 vs 0.39 ms per step at 3 x 10^5
// links) and 1.65 x faster at 3 x 10^6 (1.51 vs 2.49 ms per step).
#ifndef YA_LINKS_SEGMENTED_MIN
#define YA_LINKS_SEGMENTED_MIN 1000000
#endif
namespace ya {
// the threshold at run time (tests force either path)
inline int& links_segmented_min()
{
    static int slots = YA_LINKS_SEGMENTED_MIN;
    return slots;
}
}  // namespace ya

template<typename Pt, Link_force<Pt> force>
void link_forces(Links& links, const Pt* __restrict__ d_X, Pt* d_dX)
{
    if (links.n_max <= 0) return;
    link<Pt, force><<<(links.n_max + 255) / 256, 256>>>(
        d_X, d_dX, links.d_link, links.d_n, links.n_max, links.strength);
}

// The default force (links.cuh:135-140): segmented sum for large link sets, atomics otherwise.
template<typename Pt>
void link_forces(Links& links, const Pt* __restrict__ d_X, Pt* d_dX)
{
    if (links.n_max >= ya::links_segmented_min()) {
        ya::link_forces_segmented<Pt>(links, d_X, d_dX);
        return;
    }
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
