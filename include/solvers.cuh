// Solvers for N-body problems of spheroid cells on MI355X (gfx950).
//
// Source-level API parity with ya||a `include/solvers.cuh` (reference lines
// cited per item): Pairwise_interaction / Pairwise_friction / Generic_forces
// signatures and the default frictions (:15-50), Solution<Pt, Solver> (:56-106),
// Heun_solver with set_fixed* (:164-276), Tile_solver (:279-342), Grid with its
// four public arrays and d_grid mirror (:380-425), __constant__ d_nhood[27]
// (:428), Grid_solver with public cube_size (:465-502).
//
// The implementation is new and CDNA4-first; what differs from the reference:
//
//  * One fused force kernel per stage.  The reference runs 3 fills, the force
//    kernel, add_rhs, a thrust::reduce with a blocking device->host copy and the
//    update kernel (:231-255).  Here the force kernel writes
//    dX = gen + F + sum_v / sum_friction directly (no fills, no add_rhs, no
//    d_sum_v / d_sum_friction arrays), the centre-of-mass mean is reduced on the
//    device into a buffer the update kernel reads, and nothing but the 4-byte
//    read of n at step entry (:229) touches the host.
//  * Grid force: cells are processed in cube-sorted order from a gathered,
//    16-byte-aligned copy {X, id} (+ old_v) so neighbour reads are contiguous:
//    the 27 stencil cubes are 9 x-rows of 3 consecutive cube ids, i.e. 9
//    contiguous slot ranges of the sorted array; they are walked in exactly the
//    reference's d_nhood order (:472-483) and ascending point id inside a cube.
//    THE SUMMATION ORDER (round 5; every grid kernel and the oracle): a cell's
//    terms from its own z-plane (d_nhood[0..8]) and those from the planes below
//    and above ([9..26]) are summed separately, each in that order from +0, and
//    the two sums are added -- where the reference's thread adds all 27 cubes'
//    terms to one sum (:437-459).  It lets grid_force_bits give a tile's planes
//    to two wavefronts where that shortens a launch ("the tail"); ~1e-7 relative
//    per step beside the reference's association.
//  * Tile force: 64-thread (one wavefront) workgroups, 256-point LDS tiles
//    holding X and old_v, two barriers per tile (the reference's single barrier
//    at :302 is only safe for a 32-thread block on a 32-wide warp).
//  * One thread owns one cell i for the whole stage, as in the reference:
//    functors may update per-i state non-atomically
//    (examples/passive_growth.cu:48-51) and see original point ids.
//
// Arithmetic contract: pair distance is sqrtf(fmaf(z,z,fmaf(y,y,x*x))); all
// other arithmetic is the reference's statement-by-statement binary32.  Build
// model files with -ffp-contract=off to reproduce the CPU oracle bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "cudebug.cuh"
#include "dtypes.cuh"
#include "yalla_hip.h"

// Model files use thrust::fill / reduce on their own arrays and rely on the
// reference's solvers.cuh for the includes (examples/branching.cu:189).  The
// engine itself does not use Thrust; define YALLA_NO_THRUST to skip them.
#ifndef YALLA_NO_THRUST
#include <thrust/execution_policy.h>
#include <thrust/fill.h>
#include <thrust/reduce.h>
#include <thrust/sort.h>
#endif


// Interactions are specified between two points Xi and Xj with r = Xi - Xj
// (solvers.cuh:15-19).
template<typename Pt>
using Pairwise_interaction = Pt(Pt Xi, Pt r, float dist, int i, int j);

// Pairwise friction coefficient (solvers.cuh:21-41).
template<typename Pt>
using Pairwise_friction = float(Pt Xi, Pt r, float dist, int i, int j);

template<typename Pt>
__device__ float friction_w_neighbour(Pt Xi, Pt r, float dist, int i, int j)
{
    if (i == j) return 0;
    return dist < 1 ? 1 : 0;
}

template<typename Pt>
__device__ float friction_on_background(Pt Xi, Pt r, float dist, int i, int j)
{
    return 0;
}

// Optional generic forces, run before the pairwise interactions
// (solvers.cuh:43-53).
template<typename Pt>
using Generic_forces =
    std::function<void(const int n, const Pt* __restrict__ d_X, Pt* d_dX)>;

template<typename Pt>
void no_gen_forces(const int n, const Pt* __restrict__ d_X, Pt* d_dX)
{}


// ---- stateless functors ----------------------------------------------------------------------
// One thread owns one cell for a whole stage because a functor may keep per-cell state without
// atomics (`d_mes_nbs[i] += 1`, examples/passive_growth.cu:48-51).  Most functors do not: they are
// pure functions of their arguments, or count with atomicAdd (examples/branching.cu:105-107).
// A model says so with ONE line next to the functor,
//
//     __device__ float3 my_force(float3 Xi, float3 r, float dist, int i, int j) { ... }
//     YA_STATELESS(float3, my_force)
//
// and the solvers then pick the kernels that share a cell among several lanes whenever the
// system is too small to fill the chip with one lane per cell (ya::tile_force_coop,
// ya::grid_force_coop: 2-8 x faster steps below ~10^5 cells, bit-identical results: every sum is
// still accumulated in the one-lane kernels' order) and split the last tiles of large launches between two
// wavefronts (grid_force_bits, "the tail").  Custom friction functors take
// YA_STATELESS_FRICTION; the two defaults are known to be stateless.
namespace ya {
template<typename Pt, Pairwise_interaction<Pt> pw_int>
struct Stateless_interaction : std::false_type {};
template<typename Pt, Pairwise_friction<Pt> pw_friction>
struct Stateless_friction : std::false_type {};
// the two default frictions, for any point type (compared as template arguments: class scope,
// where naming a __device__ function is allowed on the host side too)
template<typename Pt, Pairwise_friction<Pt> pw_friction>
struct Default_friction {
    static constexpr bool value =
        pw_friction == &friction_w_neighbour<Pt> || pw_friction == &friction_on_background<Pt>;
};
template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
constexpr bool stateless_pair()
{
    return Stateless_interaction<Pt, pw_int>::value &&
           (Default_friction<Pt, pw_friction>::value || Stateless_friction<Pt, pw_friction>::value);
}
}  // namespace ya
#define YA_STATELESS(Pt_, functor_)                                            \
    namespace ya {                                                            \
    template<>                                                                \
    struct Stateless_interaction<Pt_, functor_> : std::true_type {};          \
    }
#define YA_STATELESS_FRICTION(Pt_, functor_)                                   \
    namespace ya {                                                            \
    template<>                                                                \
    struct Stateless_friction<Pt_, functor_> : std::true_type {};             \
    }


namespace ya {

#ifndef YA_FORCE_BLOCK
#define YA_FORCE_BLOCK 256
#endif
constexpr int FORCE_BLOCK = YA_FORCE_BLOCK;  // grid force: cells (threads) per workgroup
constexpr int TILE_BLOCK = 64;     // tile force: one wavefront per workgroup
constexpr int TILE_POINTS = 256;   // points staged in LDS per tile
constexpr int UPDATE_BLOCK = 256;

// One cell in cube-sorted order: the point and its original id.  16-byte
// entries (float3) load as one dwordx4.
template<typename Pt>
struct alignas(alignof(Pt) > ((sizeof(Pt) + 4) % 16 == 0 ? 16 : ((sizeof(Pt) + 4) % 8 == 0 ? 8 : 4))
                   ? alignof(Pt)
                   : ((sizeof(Pt) + 4) % 16 == 0 ? 16 : ((sizeof(Pt) + 4) % 8 == 0 ? 8 : 4))) Entry {
    Pt X;
    int id;
};

// Correctly rounded square root for x = 0 or x >= 2^-96 (every squared distance
// between distinct binary32 positions of a model): the hardware estimate
// (v_sqrt_f32, <= 1 ulp) moved down or up by one ulp according to the sign of the
// exact residuals x - (s -+ 1ulp) * s, which is the core of the compiler's own
// expansion of sqrtf without its rescaling of tiny arguments (10 fewer VALU
// instructions per interacting pair).  Tiny non-zero arguments take the library
// path.  Verified against sqrtf for EVERY binary32 argument by
// tests/test_parity_gpu.py (ya::check_sqrt_all).
//
// YA_ARITH_FAST (a build switch for a model's translation unit, together with
// -ffp-contract=fast): the fast-arithmetic tier.  The pair distance is the bare v_sqrt_f32
// (<= 1 ulp, as CUDA's norm3df, which the reference calls at solvers.cuh:308,449), `Pt / float`
// multiplies by the bare v_rcp_f32 (<= 1 ulp), and multiply-adds are contracted as nvcc
// contracts them for the reference's CUDA build.  Results then agree with the exact tier (and
// the oracle) to rounding, i.e. far inside north_star's 1e-5 relative on positions
// (tests/test_fast_arith_gpu.py holds every configuration's functor to that in lock-step), but
// not bit for bit; cube ids, cell counts and the grid arrays stay bit-exact (they are computed in
// libyalla_hip.so, which is always built without contraction).  -18 % on the 1 M-cell force launch.
__device__ __forceinline__ float exact_sqrt(float x)
{
#ifdef YA_ARITH_FAST
    return __builtin_amdgcn_sqrtf(x);
#endif
    // The common path runs unconditionally; the rare one (a tiny non-zero argument)
    // replaces its result afterwards: one skipped branch per call instead of a two-sided one.
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_down = __int_as_float(__float_as_int(s) - 1);
    const float s_up = __int_as_float(__float_as_int(s) + 1);
    const float r_down = fmaf(-s_down, s, x);
    const float r_up = fmaf(-s_up, s, x);
    float out = r_down <= 0.f ? s_down : s;
    out = r_up > 0.f ? s_up : out;
    if (__builtin_expect(x < 0x1p-96f && x > 0.f, 0)) out = sqrtf(x);
    return out;
}

__device__ __forceinline__ float dist3(float x, float y, float z)
{
    return exact_sqrt(fmaf(z, z, fmaf(y, y, x * x)));
}

// Test hook: counts the binary32 bit patterns in [first, last] for which
// exact_sqrt differs from sqrtf (NaNs compare by class).
__global__ void check_sqrt_all(unsigned first, unsigned last, unsigned long long* mismatches)
{
    unsigned long long bad = 0;
    for (unsigned long long u = (unsigned long long)first + blockIdx.x * blockDim.x + threadIdx.x;
         u <= last; u += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __int_as_float((int)(unsigned)u);
        const float a = exact_sqrt(x), b = sqrtf(x);
        if (__float_as_int(a) != __float_as_int(b) && !(a != a && b != b)) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Workgroup -> tile mapping that gives each of the 8 XCDs one contiguous range of
// tiles.  The dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md, workgroup
// dispatch): with the identity mapping every XCD's 4 MiB L2 sees the whole sorted
// array; with this one it sees an eighth of it, which is about what it can hold at
// 10^6 cells, so the stencil rows a tile reads (the tiles before and after it, one
// grid row and one grid plane away) are L2 hits.  Bijective for any grid size; a
// wrong placement guess would cost speed, not correctness.
__device__ __forceinline__ int xcd_eighth_tile(const int block, const int n_blocks, const bool descending = false)
{
    constexpr int XCDS = 8;
    const int xcd = block % XCDS, turn = block / XCDS;
    const int q = n_blocks / XCDS, r = n_blocks % XCDS;
    const int mine = q + (xcd < r ? 1 : 0);  // tiles of this XCD's eighth
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (descending ? mine - 1 - turn : turn);
}
// Each XCD gets TWO contiguous sixteenths: one from the lower half of the tiles and the
// corresponding one from the upper half.  Tiles are in cube order (z-major), and a tile's cost
// follows the local density of neighbours: for a ball of cells it rises from the first tiles
// (a polar cap: many surface cells) to the middle and falls again, so eighths in order leave
// the XCDs of the caps idle at the end of a launch while those of the equator still work.
// Pairing sixteenth k of the lower half with sixteenth k of the upper half evens that out and
// keeps an XCD's L2 to two contiguous ranges.
// Round 4, OUTSIDE IN: the ranges are visited alternately from the front and from the back of the
// list -- 0, R - 1, 1, R - 2, ... -- and those from the back are walked downwards.  The list's two
// ends are the tips of the ball's caps, where a plane of the grid holds a few hundred cells: a tile
// of 64 cells spans several grid rows there, its nine stencil rows stage whole planes (several
// chunks each), and it lives two to three times as long as a tile of the interior.  With the ranges
// in storage order the top tip came LAST: a few dozen such wavefronts running on after every other
// had finished, 35 us at the end of a 290 us launch (the last slab of the 8-slab rehearsal, whose
// list ends with the tip: GRBM_GUI_ACTIVE + 10 % for the same SQ_WAVE_CYCLES as the first slab's,
// profiles/r04_slab_pmc_by_rank.json).  Now both tips are started within the first half of a
// launch, and it ends in the middle of the list, among tiles that are all alike.
#ifndef YA_XCD_RANGES
#define YA_XCD_RANGES 2  /* contiguous ranges of tiles per XCD (1 = plain eighths) */
#endif
#ifndef YA_XCD_OUTSIDE_IN
#define YA_XCD_OUTSIDE_IN 1  /* 0 = ranges in storage order (A/B) */
#endif
template<int RANGES = YA_XCD_RANGES>
__device__ __forceinline__ int xcd_contiguous_tile(const int block, const int n_blocks)
{
    // slots of a multiple of 8 blocks each (block and block - k * part then sit on the same XCD);
    // the last slot in time takes the remainder
    const int part = (n_blocks / RANGES) & ~7;
    if (RANGES == 1 || part == 0) return xcd_eighth_tile(block, n_blocks);
    const int p = min(block / part, RANGES - 1);
    const int first_block = p * part;
    const int len = p == RANGES - 1 ? n_blocks - first_block : part;
#if YA_XCD_OUTSIDE_IN
    // slot p works on range r of the list: 0, R - 1, 1, R - 2, ...; the last slot on range R / 2, which
    // is therefore the one that holds the remainder
    constexpr int LONG_RANGE = RANGES / 2;
    const bool from_back = p & 1;
    const int r = from_back ? RANGES - 1 - (p - 1) / 2 : p / 2;
    const int start = r * part + (r > LONG_RANGE ? n_blocks - RANGES * part : 0);
    return start + xcd_eighth_tile(block - first_block, len, from_back);
#else
    return first_block + xcd_eighth_tile(block - first_block, len);
#endif
}

// Test hook: the same for ya::reciprocal (dtypes.cuh) against 1.0f / x.
__global__ void check_reciprocal_all(
    unsigned first, unsigned last, unsigned long long* mismatches)
{
    unsigned long long bad = 0;
    for (unsigned long long u = (unsigned long long)first + blockIdx.x * blockDim.x + threadIdx.x;
         u <= last; u += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __int_as_float((int)(unsigned)u);
        const float a = reciprocal(x), b = 1.0f / x;
        if (__float_as_int(a) != __float_as_int(b) && !(a != a && b != b)) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Optional HIP-event timing of the force-kernel launches (bench.py's roofline
// leg).  The pair of events handed out by next() is attached to the kernel's own
// dispatch (hipExtLaunchKernelGGL), so that timing a launch puts no extra packet
// into the stream.  Even so a timed launch is a few microseconds slower end to end,
// so only every `stride`-th launch is timed (use an odd stride to see both Heun
// stages).  Disabled by default: next() then hands out null events.
class Profiler {
public:
    void enable(bool on, int every = 1)
    {
        enabled = on;
        stride = every > 0 ? every : 1;
        seen = 0;
    }
    bool active() const { return enabled; }
    void next(hipEvent_t* start, hipEvent_t* stop)
    {
        *start = *stop = nullptr;
        if (!enabled || seen++ % stride != 0) return;
        while (used + 2 > events.size()) {
            hipEvent_t e;
            YA_CHECK((int)hipEventCreate(&e));
            events.push_back(e);
        }
        *start = events[used++];
        *stop = events[used++];
    }
    // Sum of (after - before) over the recorded launches; resets the record.
    void read(double* total_ms, int* launches)
    {
        YA_CHECK((int)hipDeviceSynchronize());
        double ms = 0;
        for (size_t k = 0; k + 1 < used; k += 2) {
            float span = 0;
            YA_CHECK((int)hipEventElapsedTime(&span, events[k], events[k + 1]));
            ms += span;
        }
        if (total_ms) *total_ms = ms;
        if (launches) *launches = (int)(used / 2);
        used = 0;
    }
    ~Profiler()
    {
        for (auto e : events) (void)hipEventDestroy(e);
    }

private:
    bool enabled = false;
    int stride = 1;
    long seen = 0;
    size_t used = 0;
    std::vector<hipEvent_t> events;
};

template<typename Pt>
bool is_no_gen_forces(const Generic_forces<Pt>& f)
{
    using Fn = void (*)(const int, const Pt* __restrict__, Pt*);
    auto* target = f.template target<Fn>();
    return target && *target == static_cast<Fn>(&no_gen_forces<Pt>);
}

// dX = gen + F, then the friction term of add_rhs (solvers.cuh:146-161).
template<typename Pt>
__device__ __forceinline__ Pt store_rhs(
    Pt* __restrict__ d_dX, int i, bool has_gen, Pt F, float3 sum_v, float sum_friction)
{
    Pt dX;
    if (has_gen) {
        dX = d_dX[i];
        dX += F;
    } else {
        dX = F;
    }
    if (sum_friction > 0) {
        dX.x += sum_v.x / sum_friction;
        dX.y += sum_v.y / sum_friction;
        dX.z += sum_v.z / sum_friction;
    }
    d_dX[i] = dX;
    return dX;
}

// All-pairs force (replaces compute_tile, solvers.cuh:284-322): j ascending,
// functor called for every (i, j) including i == j.
template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
__global__ __launch_bounds__(TILE_BLOCK) void tile_force(const int n,
    const Pt* __restrict__ d_X, const float3* __restrict__ d_old_v, Pt* __restrict__ d_dX,
    const bool has_gen)
{
    __shared__ Pt sh_X[TILE_POINTS];
    __shared__ float3 sh_v[TILE_POINTS];

    const int i = blockIdx.x * TILE_BLOCK + threadIdx.x;
    Pt Xi = ya::zero<Pt>();
    if (i < n) Xi = d_X[i];
    Pt F = ya::zero<Pt>();
    float3 sum_v{0.f, 0.f, 0.f};
    float sum_friction = 0;
    for (int tile_start = 0; tile_start < n; tile_start += TILE_POINTS) {
        const int n_tile = min(TILE_POINTS, n - tile_start);
        __syncthreads();
        for (int k = threadIdx.x; k < n_tile; k += TILE_BLOCK) {
            sh_X[k] = d_X[tile_start + k];
            sh_v[k] = d_old_v[tile_start + k];
        }
        __syncthreads();
        if (i < n) {
            // unrolled: the pairs' distance / functor chains are independent and overlap,
            // the sums stay in j order
#ifndef YA_TILE_UNROLL
#define YA_TILE_UNROLL 4
#endif
#pragma unroll YA_TILE_UNROLL
            for (int k = 0; k < n_tile; k++) {
                const int j = tile_start + k;
                Pt r = Xi - sh_X[k];
                float dist = dist3(r.x, r.y, r.z);
                F += pw_int(Xi, r, dist, i, j);
                float friction = pw_friction(Xi, r, dist, i, j);
                sum_friction += friction;
                if (friction != 0) {
                    float3 v = sh_v[k];
                    sum_v.x += friction * v.x;
                    sum_v.y += friction * v.y;
                    sum_v.z += friction * v.z;
                }
            }
        }
    }
    if (i < n) store_rhs(d_dX, i, has_gen, F, sum_v, sum_friction);
}

// The same all-pairs force with 16 or 64 lanes per cell (opt-in: Tile_computer::lanes_per_cell).
// All pairs of 800 cells are only 13 wavefronts for tile_force, each lane walking its 800
// partners alone: 0.19 ms per launch on a 256-CU chip.  Here a 256-thread workgroup owns 16 or 4
// cells; for every tile of up to 512 partners each cell's lanes evaluate eight pairs apiece and
// leave {F, friction, friction * old_v} in LDS, then ONE lane per component runs that
// component's sum over the tile in ascending j -- every per-cell sum is still accumulated in
// exactly the reference's order, and results are bit-identical to tile_force.  What changes:
// the functor is called for one i from several lanes at once, so functors that update per-cell
// state non-atomically (d_mes_nbs[i] += 1, examples/passive_growth.cu:48-51) must keep the
// default of one lane per cell.

template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction, int COOP_LANES>
__global__ __launch_bounds__(256) void tile_force_coop(const int n,
    const Pt* __restrict__ d_X, const float3* __restrict__ d_old_v, Pt* __restrict__ d_dX,
    const bool has_gen)
{
    constexpr int COOP_CELLS = 256 / COOP_LANES;
    constexpr int NF = N_floats<Pt>::value;
    constexpr int NC = NF + 4;  // components summed per cell: F (NF), friction, friction * old_v (3)
    // partners per tile: a multiple of 64 whose terms fill at most 56 KiB of LDS
    constexpr int TILE = (57344 / (COOP_CELLS * NC * 4)) / 64 * 64 > 512 ? 512 : (57344 / (COOP_CELLS * NC * 4)) / 64 * 64;
    static_assert(TILE >= 64, "point type too large for tile_force_coop");
    constexpr int THREADS = COOP_CELLS * COOP_LANES;
    constexpr int LOADS = (TILE + THREADS - 1) / THREADS;  // partners a thread carries per tile
    constexpr int SLOTS = (NC + COOP_LANES - 1) / COOP_LANES;
    __shared__ __attribute__((aligned(16))) float sh_part[COOP_CELLS][NC][TILE];  // one tile's terms, [cell][component][j]
    __shared__ Pt sh_X[TILE];                        // the tile's partners
    __shared__ float3 sh_v[TILE];
    __shared__ float sh_sum[COOP_CELLS][NC];

    const int cell = threadIdx.x / COOP_LANES, lane = threadIdx.x % COOP_LANES;
    const int i = blockIdx.x * COOP_CELLS + cell;
    const bool active = i < n;
    Pt Xi = ya::zero<Pt>();
    if (active) Xi = d_X[i];
    float acc[SLOTS];  // this lane's component sums (components lane, lane + COOP_LANES, ...)
#pragma unroll
    for (int a = 0; a < SLOTS; a++) acc[a] = 0.f;

    // the next tile's partners travel from global memory while the current tile is worked on
    Pt x_next[LOADS];
    float3 v_next[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; k++) {
        const int j = threadIdx.x + k * THREADS;
        x_next[k] = ya::zero<Pt>();
        v_next[k] = float3{0.f, 0.f, 0.f};
        if (j < TILE && j < n) {
            x_next[k] = d_X[j];
            v_next[k] = d_old_v[j];
        }
    }
    for (int tile_start = 0; tile_start < n; tile_start += TILE) {
        const int n_tile = min(TILE, n - tile_start);
#pragma unroll
        for (int k = 0; k < LOADS; k++) {
            const int t = threadIdx.x + k * THREADS;
            if (t < TILE) {
                sh_X[t] = x_next[k];
                sh_v[t] = v_next[k];
                const int j_next = tile_start + TILE + t;
                if (j_next < n) {
                    x_next[k] = d_X[j_next];
                    v_next[k] = d_old_v[j_next];
                }
            }
        }
        __syncthreads();
        // (a) the tile's pair terms: lane l takes partners l, l + COOP_LANES, ...
        if (active) {
#pragma unroll 4
            for (int jj = lane; jj < n_tile; jj += COOP_LANES) {
                const int j = tile_start + jj;
                Pt r = Xi - sh_X[jj];
                float dist = dist3(r.x, r.y, r.z);
                const Pt f = pw_int(Xi, r, dist, i, j);
                const float friction = pw_friction(Xi, r, dist, i, j);
#pragma unroll
                for (int c = 0; c < NF; c++) sh_part[cell][c][jj] = field(f, c);
                sh_part[cell][NF][jj] = friction;
                // the old_v term only where the friction is not zero (solvers.cuh:312-316): a +0
                // term instead changes no bit of a sum that started at +0 (such a sum never is -0)
                const float3 v = sh_v[jj];
                sh_part[cell][NF + 1][jj] = friction != 0 ? friction * v.x : 0.f;
                sh_part[cell][NF + 2][jj] = friction != 0 ? friction * v.y : 0.f;
                sh_part[cell][NF + 3][jj] = friction != 0 ? friction * v.z : 0.f;
            }
        }
        __syncthreads();
        // (b) one lane per component adds the tile's terms in ascending j
#pragma unroll
        for (int a = 0; a < SLOTS; a++) {
            const int c = lane + COOP_LANES * a;
            if (c < NC && active) {
                // sixteen terms per trip: four 16-byte LDS reads in flight, then the adds in order
                float sum = acc[a];
                const float4* terms = reinterpret_cast<const float4*>(&sh_part[cell][c][0]);
                int jj = 0;
                for (; jj + 16 <= n_tile; jj += 16) {
                    float4 p[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) p[u] = terms[jj / 4 + u];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        sum += p[u].x;
                        sum += p[u].y;
                        sum += p[u].z;
                        sum += p[u].w;
                    }
                }
                for (; jj < n_tile; jj++) sum += sh_part[cell][c][jj];
                acc[a] = sum;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < SLOTS; a++) {
        const int c = lane + COOP_LANES * a;
        if (c < NC) sh_sum[cell][c] = acc[a];
    }
    __syncthreads();
    if (active && lane == 0) {
        Pt F;
#pragma unroll
        for (int c = 0; c < NF; c++) field(F, c) = sh_sum[cell][c];
        store_rhs(d_dX, i, has_gen, F,
            float3{sh_sum[cell][NF + 1], sh_sum[cell][NF + 2], sh_sum[cell][NF + 3]}, sh_sum[cell][NF]);
    }
}

// Row r of the 27-cube stencil in the reference's d_nhood order
// (solvers.cuh:472-483): rows of three consecutive cube ids centred on
// 0, -gs, +gs, then the same three shifted by -gs^2, then by +gs^2.
__device__ __forceinline__ int stencil_row_offset(int row, int gs)
{
    const int dy = row % 3, dz = row / 3;
    const int oy = dy == 0 ? 0 : (dy == 1 ? -gs : gs);
    const int oz = dz == 0 ? 0 : (dz == 1 ? -gs * gs : gs * gs);
    return oy + oz;
}

// Grid force (replaces compute_cube, solvers.cuh:430-463).  Thread s owns
// sorted slot s.  offs[c] = first slot of cube c, so the cubes c-1, c, c+1 of a
// stencil row are the contiguous slots [offs[c-1], offs[c+2]).

// The first 16 bytes of a staged cell (x, y, z and one more word).  Entries whose
// size is a multiple of 16 bytes are read as one 16-byte LDS access: 4 LDS cycles
// per wavefront, where the 12-byte form the compiler picks for x, y, z alone takes
// 8 (MI355X_MICROARCH.md, section LDS).  keep_wide() is what stops it narrowing.
template<typename Pt>
__device__ __forceinline__ float4 staged_words(const Entry<Pt>* e)
{
    if constexpr (sizeof(Entry<Pt>) % 16 == 0)
        return *reinterpret_cast<const float4*>(e);
    else
        return float4{e->X.x, e->X.y, e->X.z, 0.f};
}
__device__ __forceinline__ void keep_wide(const float4& w) { asm volatile("" ::"v"(w.w)); }

template<typename Pt>
__device__ __forceinline__ float dist2_to(const Pt& a, const float4 b)
{
    const float dx = a.x - b.x;
    const float dy = a.y - b.y;
    const float dz = a.z - b.z;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

template<typename Pt>
__device__ __forceinline__ float dist2(const Pt& a, const Pt& b)
{
    const float dx = a.x - b.x;
    const float dy = a.y - b.y;
    const float dz = a.z - b.z;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// Smallest binary32 t with sqrtf(t) >= cube_size.  sqrtf is monotonic and
// correctly rounded on host and device, so `d2 < t` is exactly the reference's
// `dist < cube_size` (solvers.cuh:450) without a square root per candidate.
inline float cutoff_squared(float cube_size)
{
    float t = cube_size * cube_size;
    while (t > 0 && sqrtf(t) >= cube_size) t = nextafterf(t, 0.f);
    while (sqrtf(t) < cube_size) t = nextafterf(t, INFINITY);
    return t;
}

// A model's pairwise functor is inlined where the grid kernels call it, whatever its size.  Left to
// the inliner, a functor with libm calls in it (bending_force's sinf / cosf) stays a function that is
// called per pair: it then re-reads the device symbols it uses (`d_type`, `d_mes_nbs` ...) and cell i's
// own entries for every pair, each of them a round trip in front of the arithmetic.
#ifndef YA_CALL_INLINED
#define YA_CALL_INLINED [[clang::always_inline]]
#endif

// Row bounds of a plane of the stencil for a workgroup that owns the cubes [c_lo, c_hi] and a lane in
// cube c: six workgroup-uniform and six per-lane reads of offs[].  Those of plane p + 1 are requested
// while plane p computes, so that a plane's staging loads do not queue behind a round trip to L2 for
// its bounds.  Expects next_lo / next_hi / next_begin / next_end [3], c_lo, c_hi, c, gs, n_cubes, offs.
#define YA_ROW_BOUNDS(plane_)                                                          \
    _Pragma("unroll") for (int r = 0; r < 3; r++)                                      \
    {                                                                                  \
        const int off = stencil_row_offset(3 * (plane_) + r, gs);                      \
        /* The reference indexes cube_start/end without bounds checks               */ \
        /* (solvers.cuh:444); out-of-grid cubes are treated as empty here.          */ \
        next_lo[r] = offs[min(max(c_lo + off - 1, 0), n_cubes)];                       \
        next_hi[r] = offs[min(max(c_hi + off + 2, 0), n_cubes)];                       \
        next_begin[r] = offs[min(max(c + off - 1, 0), n_cubes)];                       \
        next_end[r] = offs[min(max(c + off + 2, 0), n_cubes)];                         \
    }

}  // namespace ya
#ifdef YA_EXPERIMENTAL_FORCE_VARIANTS
#include "force_variants.cuh"  // tools/ab/: grid_force_direct, grid_force (byte FIFO) -- A/B baselines for tests and tools (-Itools/ab)
#endif
namespace ya {

// ---------------------------------------------------------------------------------
// grid_force_bits (the default): the same two-phase kernel with the hit list kept as a BIT
// STREAM (one bit per tested candidate, in the reference's order) instead of a byte FIFO,
// in one-wavefront workgroups.
//
//   phase 1  per candidate: one 16-byte LDS read, the squared distance and
//            `m = 2 m + (d2 < cut2)` -- a compare and an add-with-carry.  A word is
//            stored to LDS once per four candidates (one ds_write_b32 where the byte
//            FIFO issued four ds_write_b8) in [word][thread] order: no bank conflicts.
//            Rows are padded to a multiple of four bits, so there is no one-candidate
//            tail loop: the last group of a row is tested under a mask.
//   phase 2  pops the set bits highest first (= ascending candidate order), two per trip:
//            the second hit's LDS / global loads are in flight while the first one's
//            arithmetic runs (the sums stay in order).  The bit position gives the
//            staged cell through three per-lane row offsets; the next word of the stream
//            is requested one trip ahead.
//
// A pass covers 32 * YA_MASK_WORDS candidate bits per lane.  A plane's three rows fit
// into one pass at any density a model normally runs at (~90 candidates per plane at
// rho = 9.8); a wavefront in which some lane has more falls back to one pass per row
// stretch.  A workgroup is ONE wavefront (64 cells): nothing waits at a workgroup barrier
// for a slower wavefront, launches of 10^4..10^5 cells spread over four times as many
// CUs, and the workgroup needs 7 KiB of LDS.  MI355X, same box (tools/micro/force_ab.hip):
// 242 us per 1 M-cell launch against 260 us for round 1's byte-FIFO kernel, 62 against 75 us at 10^5 cells,
// 803 against 853 us at 4 M.  What bounds it is the VALU issue rate.
// Results are bit-identical to the earlier kernels' (tools/ab/force_variants.cuh): same candidates, same
// order, same arithmetic.
// The friction terms of one pair (solvers.cuh:309-313, :453-458): sum_friction += friction,
// sum_v += friction * old_v[j].  For the default functor friction_w_neighbour the coefficient f
// is 0 or 1, so f * v is exact and `fmaf(f, v, sum)` is `sum + f * v` with its one rounding: the
// reference's bits, unconditionally as the reference adds it (a non-finite old_v poisons the sum
// there and here), in one instruction per component where `sum += nb ? v : 0` took two (round 6:
// 242 -> 237 us per 1 M-cell launch, profiles/r06_force_ab.jsonl; round 3's selections had been
// -2.6 % against the multiply-and-add before them).
template<typename Pt, Pairwise_friction<Pt> pw_friction>
__device__ __forceinline__ void pair_friction(const Pt& Xi, const Pt& r, const float dist, const int i,
    const int j, const float4& v, float3& sum_v, float& sum_friction)
{
    if constexpr (pw_friction == &friction_w_neighbour<Pt>) {
        const bool nb = (i != j) & (dist < 1.f);
        const float f = nb ? 1.f : 0.f;
        sum_friction += f;
        sum_v.x = fmaf(f, v.x, sum_v.x);
        sum_v.y = fmaf(f, v.y, sum_v.y);
        sum_v.z = fmaf(f, v.z, sum_v.z);
    } else {
        const float friction = pw_friction(Xi, r, dist, i, j);
        sum_friction += friction;
        if (friction != 0) {
            sum_v.x += friction * v.x;
            sum_v.y += friction * v.y;
            sum_v.z += friction * v.z;
        }
    }
}

// Hooks of tools/micro/force_trace.hip (when and where every workgroup of a launch ran): empty
// unless tools/ab/force_trace.cuh was included first.
#ifndef YA_BITS_PROBE_BEGIN
#define YA_BITS_PROBE_BEGIN
#define YA_BITS_PROBE_END(tile_)
#endif
namespace bits {
#ifndef YA_BITS_BLOCK
#define YA_BITS_BLOCK 64
#endif
#ifndef YA_MASK_WORDS
#define YA_MASK_WORDS 4
#endif
#ifndef YA_BITS_POPS
#define YA_BITS_POPS 2  /* hits popped per trip of phase 2 (1 or 2), three-float points */
#endif
#ifndef YA_BITS_POPS_WIDE
#define YA_BITS_POPS_WIDE 2  /* the same for wider points (A/B knob: one pop per trip saves 10 VGPRs of config */
#endif                       /* 4's 146 and is 3 % slower there: two hits' memory accesses in flight matter more) */
// Wavefronts per SIMD the register allocator must leave room for (A/B knob).  Default 1 = the
// compiler's own choice.  Measured on config 4 (1.16 M Po_cell, relu_w_epithelium: 146 VGPRs, three
// wavefronts per SIMD, 78 % of wave cycles at s_waitcnt): forcing four (128 VGPRs + 8 spilled)
// made the launch SLOWER, 1431 -> 1615 us -- that kernel is bound by the address path of the
// functor's own global accesses (d_type[i], d_type[j], d_mes_nbs[i] += 1 by original id: 64 cache
// lines per wavefront instruction, TA 73 % busy), and more wavefronts only queue there.
// Round 4: once the MODEL keeps its ids in cube order (Solution::renumber) those accesses are
// neighbouring lines, and while the functor was still a called function four wavefronts per SIMD and
// one hit per trip were the faster build there, 652 -> 585 us per launch (profiles/r04_cfg4_renumber_ab.txt:
// a second build of the kernel for renumbered models).  With the functor inlined (YA_CALL_INLINED) the
// compiler's own choice and two hits per trip are the fastest again, renumbered or not: 559-566 us
// (profiles/r04_inline_ab.txt) -- a four-wavefront build then spills the force sums inside the loop
// (752 us; 1000 us with `int* d_type`) -- and the second build is gone.
#ifndef YA_BITS_MIN_WAVES_WIDE
#define YA_BITS_MIN_WAVES_WIDE 1
#endif
template<typename Pt>
struct Min_waves {
    static constexpr int value = sizeof(Pt) <= 16 ? 1 : YA_BITS_MIN_WAVES_WIDE;
};
template<typename Pt>
struct Pops {
    static constexpr int value = sizeof(Pt) <= 16 ? YA_BITS_POPS : YA_BITS_POPS_WIDE;
};
constexpr int BLOCK = YA_BITS_BLOCK;
constexpr int WORDS = YA_MASK_WORDS;
constexpr int PASS_BITS = 32 * WORDS;
using Lds_word = __attribute__((address_space(3))) unsigned;

template<typename Pt>
struct Stage {
#ifndef YA_BITS_STAGE_CELLS
#define YA_BITS_STAGE_CELLS (3 * YA_BITS_BLOCK + 160)
#endif
    // 16- and 24-byte entries: 352 cells (7 / 9.7 KiB per wavefront with the masks: 5 / 4 wavefronts
    // per SIMD); 32-byte entries (7-float points): 256 cells = 9.5 KiB, so that a fourth wavefront
    // fits the CU's LDS (352 were 12.6 KiB: three) -- denser planes are staged in chunks
    static constexpr int value = sizeof(Entry<Pt>) <= 24 ? YA_BITS_STAGE_CELLS
                                 : (sizeof(Entry<Pt>) <= 32 ? 256 : YA_BITS_STAGE_CELLS / 2);
};

// m = 2 m + (d2 < cut2)
__device__ __forceinline__ void shift_in(unsigned& m, const float d2, const float cut2)
{
    asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
        : "+v"(m)
        : "v"(d2), "s"(cut2)
        : "vcc");
}

// One pass: up to three candidate segments [b_r, e_r) of the staged cells (LDS indices;
// empty if b_r >= e_r), at most PASS_BITS bits after padding each to a multiple of four.
// shift_r turns an LDS index of segment r into a slot of the sorted arrays.
template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction, bool STAGE_V, bool GLOBAL_IDS>
__device__ __forceinline__ void pass(const Entry<Pt>* __restrict__ sh_e, const float4* __restrict__ sh_v,
    Lds_word* const words,
    const int b0, const int e0, const int b1, const int e1, const int b2, const int e2,
    const int shift0, const int shift1, const int shift2, const float4* __restrict__ sorted_v,
    const Pt& Xi, const int i, const float cut2, Pt& F, float3& sum_v, float& sum_friction,
    const int* __restrict__ global_id)
{
    constexpr int POPS = Pops<Pt>::value;
    // ---- phase 1 ----
    unsigned m = 0;
    int p = 0;  // bits emitted
    Lds_word* mp = words;
#define YA_BITS_GROUP(end_, masked_)                                                   \
    {                                                                                  \
        float4 w[4];                                                                   \
        float d2[4];                                                                   \
        _Pragma("unroll") for (int u = 0; u < 4; u++) w[u] = staged_words(&sh_e[t + u]); \
        _Pragma("unroll") for (int u = 0; u < 4; u++)                                  \
        {                                                                              \
            d2[u] = dist2_to(Xi, w[u]);                                                \
            keep_wide(w[u]);                                                           \
            if (masked_) d2[u] = t + u < (end_) ? d2[u] : INFINITY;                    \
        }                                                                              \
        _Pragma("unroll") for (int u = 0; u < 4; u++) shift_in(m, d2[u], cut2);        \
        p += 4;                                                                        \
        *mp = m; /* the word's last store, when it is full, is the one that counts */  \
        const bool full = (p & 31) == 0;                                               \
        mp += full ? BLOCK : 0;                                                        \
        m = full ? 0u : m;                                                             \
    }
#define YA_BITS_SEGMENT(begin_, end_)                                                  \
    {                                                                                  \
        int t = (begin_);                                                              \
        for (; t + 4 <= (end_); t += 4) YA_BITS_GROUP(end_, false)                     \
        if (t < (end_)) YA_BITS_GROUP(end_, true)                                      \
    }
    const int d0 = b0;  // LDS index = bit position + d_r inside segment r
    YA_BITS_SEGMENT(b0, e0)
    const int p1 = p, d1 = b1 - p1;
    YA_BITS_SEGMENT(b1, e1)
    const int p2 = p, d2_ = b2 - p2;
    YA_BITS_SEGMENT(b2, e2)
#undef YA_BITS_SEGMENT
#undef YA_BITS_GROUP
    if (p & 31) *mp = m << (32 - (p & 31));  // left-align the last, partial word

    // ---- phase 2 ----
    int left = ((p + 31) >> 5) - 1;  // words after the current one
    unsigned cur = p > 0 ? words[0] : 0u;
    Lds_word* rp = words + BLOCK;
    unsigned nxt = *rp;  // one spare row keeps this in bounds
    int wbase = 0;
#define YA_BITS_LOAD(q_, other_, v_)                                                   \
    {                                                                                  \
        const bool in2 = (q_) >= p2, in1 = (q_) >= p1;                                 \
        const int t = (q_) + (in2 ? d2_ : (in1 ? d1 : d0));                            \
        other_ = sh_e[t];                                                              \
        if (STAGE_V)                                                                   \
            v_ = sh_v[t];                                                              \
        else                                                                           \
            v_ = sorted_v[(unsigned)(t + (in2 ? shift2 : (in1 ? shift1 : shift0)))];   \
    }
#define YA_BITS_PAIR(other_, v_)                                                       \
    {                                                                                  \
        Pt r = Xi - other_.X;                                                          \
        float dist = dist3(r.x, r.y, r.z);                                             \
        const int j = GLOBAL_IDS ? global_id[other_.id] : other_.id;                   \
        YA_CALL_INLINED F += pw_int(Xi, r, dist, i, j);                                \
        pair_friction<Pt, pw_friction>(Xi, r, dist, i, j, v_, sum_v, sum_friction);    \
    }
#ifdef YA_BITS_MEASUREMENT_PROBES  // measurement builds only (tools/micro/force_ab.hip; -Itools/ab): the balance and phase-1 probes
#include "bits_probe.inc"
#endif
    while (cur != 0 || left > 0) {
        const bool refill = cur == 0;  // then left > 0
        cur = refill ? nxt : cur;
        wbase += refill ? 32 : 0;
        rp += refill ? BLOCK : 0;
        left -= refill ? 1 : 0;
        nxt = *rp;
        if (cur != 0) {
            // up to POPS hits of the word: the loads of the later ones are issued before
            // the first one's arithmetic (a hit that is not there repeats the one before it)
            int pos[POPS];
            bool have[POPS];
            Entry<Pt> other[POPS];
            float4 v[POPS];
#pragma unroll
            for (int u = 0; u < POPS; u++) {
                have[u] = cur != 0;
                pos[u] = have[u] ? __builtin_clz(cur) : pos[u > 0 ? u - 1 : 0];
                cur = have[u] ? cur & (0x7fffffffu >> pos[u]) : cur;
                YA_BITS_LOAD(wbase + pos[u], other[u], v[u])
            }
            YA_BITS_PAIR(other[0], v[0])
#pragma unroll
            for (int u = 1; u < POPS; u++)
                if (have[u]) YA_BITS_PAIR(other[u], v[u])
        }
    }
#undef YA_BITS_LOAD
#undef YA_BITS_PAIR
}
}  // namespace bits

// GLOBAL_IDS (z-slab decomposition): functors get global_id[local index]; a template parameter
// rather than a null test so that the single-GPU kernel carries neither the test nor the gather.
template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction, bool STAGE_V = false,
    bool GLOBAL_IDS = false>
__global__ __launch_bounds__(bits::BLOCK, bits::Min_waves<Pt>::value) void grid_force_bits(const int n,
    const Entry<Pt>* __restrict__ sorted, const float4* __restrict__ sorted_v,
    const int* __restrict__ cube_id, const int* __restrict__ offs, const int gs,
    const int n_cubes, const float cut2, Pt* __restrict__ d_dX, const bool has_gen,
    const int n_active, Pt* __restrict__ d_dX_sorted, const int* __restrict__ global_id, const int n_tiles,
    const int tail, float* tail_exchange, int* tail_tickets, const bool by_plane,
    const int part = 0, const int part_cube_lo = 0, const int part_cube_hi = 0, const int own_cube_lo = 0,
    const int own_cube_hi = 0x7fffffff)
{
    constexpr int FB = bits::BLOCK;
    constexpr int CAP = bits::Stage<Pt>::value;
    constexpr int NF = N_floats<Pt>::value;
    constexpr int NC = NF + 4;  // sums kept per cell: F (NF), friction * old_v (3), friction
    __shared__ __attribute__((aligned(16))) Entry<Pt> sh_e[CAP + 8];  // slack: whole groups are read
    __shared__ unsigned sh_m[(bits::WORDS + 1) * FB];                // [word][thread], one spare row
    // STAGE_V (small systems, Grid_computer::forces): old_v of the staged cells in LDS as well.
    // A launch that cannot fill the chip is one wavefront per SIMD and nothing hides the round
    // trip to L2 that every interacting pair's old_v otherwise costs.
    __shared__ float4 sh_v[STAGE_V ? CAP + 8 : 1];
    bits::Lds_word* const words = (bits::Lds_word*)sh_m + threadIdx.x;

    // z-slab decomposition: a stage's forces in two launches, so that the right-hand sides the
    // slab neighbours wait for are computed and sent first.  part 1 = the tiles that hold the
    // cells of the cubes below part_cube_lo or from part_cube_hi up (the z-planes next to the
    // slab's faces: cube ids are z-major, so these are the two ends of the sorted array), part 2
    // = the tiles in between, 0 = every tile.  Either launch spreads ITS tiles over all eight
    // XCDs (a launch that kept the whole array's mapping would occupy only the XCDs that own its
    // end of the array, and take as long as the full launch); blocks beyond its share exit.
    // Round 4: cubes below own_cube_lo and from own_cube_hi up hold mirrored cells only (the caller
    // knows how far own cells can have strayed): their tiles are not part of any launch.  And part 1
    // is dealt to the XCDs in SIXTEEN short ranges each instead of two: its list of tiles runs
    // mirrored cells -> own cells at the lower face and own -> mirrored at the upper one (a tile of
    // mirrored cells costs nothing), and with two ranges the XCDs in the middle of the list got
    // nearly twice the work of those at its ends -- all of them also carry an eighth of part 2.
    //
    // SUMMATION ORDER.  by_plane = false (Grid_computer::sum_order = YA_SUM_REFERENCE, the default): every
    // cell's terms go to ONE running sum over the nine rows in the reference's d_nhood order -- the
    // reference's arithmetic (solvers.cuh:437-459) -- and every tile is a whole tile (tail == 0).
    // by_plane = true (YA_SUM_BY_PLANE, a model's opt-in): S[dz = 0] + S[dz = -1, +1], which is what lets
    // THE TAIL exist (round 5): a launch ends with one wavefront lifetime at falling occupancy (16 % of a launch of
    // 10^6 cells).  The last `tail` tiles of a launch's list are then given to TWO one-wavefront
    // workgroups each -- half 0 the cells' own z-plane (~63 % of the pairs), half 1 the planes below and above
    // -- that live half as long and meet through memory: the half that finishes first leaves its sums in the
    // exchange area and draws the tile's ticket, the half that draws the second ticket adds the other's sums
    // to its own (a + b == b + a bit for bit) and stores the cells' right-hand sides.  A whole tile keeps the
    // own plane's sums aside and adds them at the end, so every cell's sums are S[dz = 0] + S[dz = -1, +1],
    // each in the reference's order from +0, whoever computed them: which tiles are split is a scheduling
    // decision without any effect on results (and the oracle sums in this order).  ONE launch, hardware
    // dispatch: blocks [0, whole) are whole tiles, blocks whole + 16 g + x and whole + 16 g + 8 + x the halves
    // of tile whole + 8 g + x, both on XCD x.  (Everything else that was tried to shorten the drain lost:
    // DESIGN.md section 6, round 5.)
    int tile, mine, t_first = 0, t_lo = 0, t_hi = 0;
    if (part == 0) {
        mine = n_tiles;
    } else {
        const int tiles = n_tiles;
        t_first = min(offs[min(max(own_cube_lo, 0), n_cubes)] / FB, tiles);
        const int t_end = max(min((offs[min(own_cube_hi, n_cubes)] + FB - 1) / FB, tiles), t_first);
        t_lo = max(min((offs[min(part_cube_lo, n_cubes)] + FB - 1) / FB, t_end), t_first);
        t_hi = max(min(offs[min(part_cube_hi, n_cubes)] / FB, t_end), t_lo);
        mine = part == 1 ? (t_lo - t_first) + (t_end - t_hi) : t_hi - t_lo;
    }
    // (tail < 0: EVERY tile as halves -- launches whose half tiles are all resident at once, where what
    // counts is not the drain but the lifetime itself)
    // (the two launches of a slab stage learn their list's length here, not on the host: a list of fewer than
    // four tails' tiles stays whole)
    const int whole = tail < 0 ? 0 : (tail > 0 && (part == 0 || mine >= 4 * tail) ? max(mine - tail, 0) & ~7 : mine);
    int half = -1, compact = blockIdx.x;
    if (compact >= whole) {
        const int b = compact - whole;
        half = (b >> 3) & 1;
        compact = whole + (b >> 4) * 8 + (b & 7);
    }
    if (compact >= mine) return;
    if (part == 0) {
        tile = xcd_contiguous_tile(compact, mine);
    } else {
        // (part 2 as well: in the cap of a ball -- the first and the last slab -- the tiles' cost falls
        // or rises all along the list, and two ranges per XCD, paired for a whole ball's rise and
        // fall, left the XCD with the two dearest ranges 10 % behind)
        const int t = xcd_contiguous_tile<16>(compact, mine);
        if (part == 1)
            tile = t < t_lo - t_first ? t_first + t : t_hi + (t - (t_lo - t_first));
        else
            tile = t_lo + t;
    }
    const int s0 = tile * FB;
    const int s = s0 + threadIdx.x;
    bool active = s < n;
    const int c_lo = cube_id[s0];
    const int c_hi = cube_id[min(s0 + FB, n) - 1];
    YA_BITS_PROBE_BEGIN

    Pt Xi = ya::zero<Pt>();
    int i = 0, c = c_lo;
    if (active) {
        const Entry<Pt> self = sorted[s];
        Xi = self.X;
        i = self.id;
        c = cube_id[s];
        active = i < n_active;  // ghost cells of a slab decomposition get no force
    }
    if (!__syncthreads_or(active)) return;  // a workgroup of ghosts only
    // functors see GLOBAL ids in a slab decomposition (they index per-cell model arrays)
    const int gi = GLOBAL_IDS && active ? global_id[i] : i;
    Pt F = ya::zero<Pt>();
    float3 sum_v{0.f, 0.f, 0.f};
    float sum_friction = 0;

    // a whole tile: the sums of the own plane while the other two planes are walked -- in registers for
    // three-float points (91 VGPRs: five wavefronts per SIMD as before), in a lane-private LDS column for
    // wider ones, whose functors leave no registers (relu_w_epithelium: 161 -> 172 VGPRs would cost the third
    // wavefront per SIMD)
    constexpr bool OWN_IN_LDS = sizeof(Pt) > 16;
    __shared__ float sh_own[OWN_IN_LDS ? NC * FB : 1];
    float own[OWN_IN_LDS ? 1 : NC];
#pragma unroll
    for (int k = 0; k < (OWN_IN_LDS ? 1 : NC); k++) own[k] = 0.f;

    int next_lo[3], next_hi[3], next_begin[3], next_end[3];
    const int plane_first = half == 1 ? 1 : 0, plane_end = half == 0 ? 1 : 3;
#pragma unroll 1
    for (int plane = plane_first; plane < plane_end; plane++) {
        if (by_plane && half < 0 && plane == 1) {
            float save[NC];
#pragma unroll
            for (int k = 0; k < NF; k++) save[k] = field(F, k);
            save[NF] = sum_v.x, save[NF + 1] = sum_v.y, save[NF + 2] = sum_v.z, save[NF + 3] = sum_friction;
#pragma unroll
            for (int k = 0; k < NC; k++) {
                if (OWN_IN_LDS)
                    sh_own[k * FB + threadIdx.x] = save[k];
                else
                    own[OWN_IN_LDS ? 0 : k] = save[k];
            }
            F = ya::zero<Pt>(), sum_v = float3{0.f, 0.f, 0.f}, sum_friction = 0;
        }
        int wg_begin[3], v0[4], k_begin[3], k_end[3];
        v0[0] = 0;
        YA_ROW_BOUNDS(plane)
#pragma unroll
        for (int r = 0; r < 3; r++) {
            wg_begin[r] = next_lo[r];
            v0[r + 1] = v0[r] + next_hi[r] - wg_begin[r];
            k_begin[r] = next_begin[r];
            k_end[r] = active ? next_end[r] : k_begin[r];
        }
        const int total = v0[3];

        for (int chunk = 0; chunk < total; chunk += CAP) {
            const int chunk_n = min(CAP, total - chunk);
            __syncthreads();
            for (int t = threadIdx.x; t < chunk_n; t += FB) {
                const int v = chunk + t;
                const int shift = v >= v0[2] ? wg_begin[2] - v0[2]
                                             : (v >= v0[1] ? wg_begin[1] - v0[1] : wg_begin[0]);
                sh_e[t] = sorted[v + shift];
                if (STAGE_V) sh_v[t] = sorted_v[v + shift];
            }
            __syncthreads();
            // this lane's candidates of the three rows, as LDS indices clipped to the chunk
            int sb[3], se[3], shift[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                sb[r] = max(k_begin[r] - wg_begin[r] + v0[r], chunk) - chunk;
                se[r] = min(k_end[r] - wg_begin[r] + v0[r], chunk + chunk_n) - chunk;
                shift[r] = wg_begin[r] - v0[r] + chunk;
            }
            const int bits_needed = (max(se[0] - sb[0], 0) + 3 & ~3) + (max(se[1] - sb[1], 0) + 3 & ~3) +
                                    (max(se[2] - sb[2], 0) + 3 & ~3);
            if (!__any(bits_needed > bits::PASS_BITS)) {
                bits::pass<Pt, pw_int, pw_friction, STAGE_V, GLOBAL_IDS>(sh_e, sh_v, words, sb[0], se[0], sb[1], se[1], sb[2],
                    se[2], shift[0], shift[1], shift[2], sorted_v, Xi, gi, cut2, F, sum_v, sum_friction,
                    global_id);
            } else {
                // dense rows: one pass per stretch of PASS_BITS candidates, rows in order
#pragma unroll 1
                for (int r = 0; r < 3; r++) {
                    const int rb = r == 0 ? sb[0] : (r == 1 ? sb[1] : sb[2]);
                    const int re = r == 0 ? se[0] : (r == 1 ? se[1] : se[2]);
                    const int rs = r == 0 ? shift[0] : (r == 1 ? shift[1] : shift[2]);
#pragma unroll 1
                    for (int b = rb; __any(b < re); b += bits::PASS_BITS)
                        bits::pass<Pt, pw_int, pw_friction, STAGE_V, GLOBAL_IDS>(sh_e, sh_v, words, b, min(re, b + bits::PASS_BITS),
                            0, 0, 0, 0, rs, 0, 0, sorted_v, Xi, gi, cut2, F, sum_v, sum_friction, global_id);
                }
            }
        }
    }
    float sums[NC];
#pragma unroll
    for (int k = 0; k < NF; k++) sums[k] = field(F, k);
    sums[NF] = sum_v.x, sums[NF + 1] = sum_v.y, sums[NF + 2] = sum_v.z, sums[NF + 3] = sum_friction;
    if (half < 0) {
        if (by_plane) {
#pragma unroll
            for (int k = 0; k < NC; k++)  // S[dz = 0] + S[dz = -1, +1]
                sums[k] = (OWN_IN_LDS ? sh_own[k * FB + threadIdx.x] : own[OWN_IN_LDS ? 0 : k]) + sums[k];
        }
    } else {
        // INVARIANT the hand-over below relies on (not promised by the HIP memory model; guarded by
        // tests/test_parity_gpu.py::test_tail_exchange_over_many_launches): both halves of a tile run on ONE
        // XCD (blocks whole + 16 g + x and whole + 16 g + 8 + x: same x = blockIdx % 8), the sums are
        // stored with device-scope (sc1) relaxed atomics that go past that XCD's L2, and the ticket is drawn
        // only after `s_waitcnt vmcnt(0)` has seen every one of those stores acknowledged.
        // Device-scope relaxed atomic stores and loads (sc1) are performed past the XCD's L2, so no fence is
        // needed for the DATA (a device-scope release fence would write back the whole L2 once per workgroup:
        // 65 us per 1024 workgroups, core.hip's one-launch reduction).  What orders the sums before the ticket
        // is the explicit wait: the ticket is drawn only after every store of this wavefront has been
        // acknowledged at device scope (a workgroup-scope fence and the barrier compile to nothing in a
        // one-wavefront workgroup, and the stores and the atomic travel to different channels).
        __shared__ int sh_second;
        const int slot = compact - whole;
        float* const mine_out = tail_exchange + ((size_t)slot * 2 + half) * NC * FB + threadIdx.x;
#pragma unroll
        for (int k = 0; k < NC; k++)
            __hip_atomic_store(mine_out + k * FB, sums[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            sh_second = __hip_atomic_fetch_add(&tail_tickets[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (sh_second == 0) {  // the other half will finish the tile
            YA_BITS_PROBE_END(tile)
            return;
        }
        if (threadIdx.x == 0) __hip_atomic_store(&tail_tickets[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float* const theirs = tail_exchange + ((size_t)slot * 2 + (1 - half)) * NC * FB + threadIdx.x;
#pragma unroll
        for (int k = 0; k < NC; k++)
            sums[k] = sums[k] + __hip_atomic_load(theirs + k * FB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < NF; k++) field(F, k) = sums[k];
        const Pt dX = store_rhs(d_dX, i, has_gen, F, float3{sums[NF], sums[NF + 1], sums[NF + 2]}, sums[NF + 3]);
        if (d_dX_sorted) d_dX_sorted[s] = dX;  // for the sorted-space Euler stage
    }
    YA_BITS_PROBE_END(tile)
}


// ---------------------------------------------------------------------------------
// grid_force_coop (the solver's own choice below ~7 * 10^4 cells for functors declared YA_STATELESS;
// forced by Grid_computer::force_variant = 3): the grid force with SEVERAL LANES
// PER CELL, for systems too small to fill the chip with one lane per cell.  A launch of
// <= 10^5 cells is at most one wavefront per SIMD for the kernels above, each lane working
// through its ~265 candidates and ~41 hits alone: 52-65 us whatever n is.  Here a 256-thread
// workgroup owns 256 / LANES consecutive sorted slots (LANES = 16, 8 or 4, chosen from n).  All
// nine stencil rows are staged in LDS at once when they fit (they do at the densities models
// run at; otherwise plane by plane, in chunks), then for every cell
//
//   A  row by row in the reference's order, each of its lanes tests a contiguous share of the
//      row's candidates into a bit mask (as grid_force_bits does); a prefix sum of the lanes'
//      hit counts ranks the hits, which are left as one compact list per cell in LDS, still in
//      the reference's order;
//   B  LANES hits at a time, lane l evaluates hit l of the round (distance, functor,
//      friction) and leaves the pair's terms {F, friction, friction * old_v} in LDS;
//   C  ONE lane per component adds the round's terms to that component's sum in list order.
//
// Every per-cell sum is therefore accumulated in exactly the order of the kernels above (the reference's one
// running sum, or own plane | other planes under YA_SUM_BY_PLANE) and the result is bit-identical to theirs.  What changes is the same as for tile_force_coop: the
// functor is called for one i from several lanes at once, so functors that update per-cell
// state non-atomically (d_mes_nbs[i] += 1, examples/passive_growth.cu:48-51) must keep one
// lane per cell.  A, B and C of a cell run inside one wavefront: the only workgroup barriers
// are those around staging.
namespace coop {
constexpr int BLOCK = 256;
#ifndef YA_COOP_MAX_HITS
#define YA_COOP_MAX_HITS 96
#endif
constexpr int MAX_HITS = YA_COOP_MAX_HITS;  // hit-list length per cell; more hits are worked off in parts
constexpr int STRETCH = 64;   // candidates of a row ranked at a time (<= MAX_HITS, <= 32 per lane)
// staged cells: nine rows of (the workgroup's cells + two cubes) at rho ~ 10 per cube; wider
// entries go plane by plane unless the workgroup is 16 cells (branching model, 32-byte entries,
// 10^4 cells: 200 -> 171 us per step with all rows at once, but 3 * 10^4 with 8 lanes 212 -> 225)
template<typename Pt, int LANES>
struct Stage {
    static constexpr bool all_rows = sizeof(Entry<Pt>) <= 16 || (sizeof(Entry<Pt>) <= 32 && LANES == 16);
    static constexpr int value = (all_rows ? 9 : 5) * (BLOCK / LANES + 34);
};
// Lanes per cell for a launch of n cells (MI355X, springs at rho ~ 10, tools/micro/force_ab.hip:
// 16 lanes 19 us at 10^4 cells, 8 lanes 24 us at 3 * 10^4, 4 lanes 39 us at 6.5 * 10^4, where one lane
// per cell takes 53, 64 and 61 us); 1 = one lane per cell is as fast or faster -- since round 5 that is
// grid_force_bits with EVERY tile as two half-tile workgroups while all of them are resident at once
// (Grid_computer::forces): 53 us at 10^5 cells where 4 lanes take 58 (their 1250 workgroups no longer fit
// the chip in one round from 8 * 10^4 cells on), 63 at 1.5 * 10^5 where they take 79 and whole tiles 71.
// Under the default summation order (no half tiles) four lanes per cell stay ahead of whole tiles up to
// ~1.2 * 10^5 cells (10^5: 58 against 62 us; 1.5 * 10^5: 79 against 71).
inline int lanes_for(const int n, const bool by_plane = false)
{
    return n <= 15000 ? 16 : (n <= 40000 ? 8 : (n <= (by_plane ? 70000 : 120000) ? 4 : 1));
}
// LDS traffic between the lanes of ONE wavefront: the hardware keeps a wavefront's LDS
// operations in order, the compiler must too.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace coop

template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction, int LANES>
__global__ __launch_bounds__(coop::BLOCK) void grid_force_coop(const int n,
    const Entry<Pt>* __restrict__ sorted, const float4* __restrict__ sorted_v,
    const int* __restrict__ cube_id, const int* __restrict__ offs, const int gs,
    const int n_cubes, const float cut2, Pt* __restrict__ d_dX, const bool has_gen,
    const int n_active, Pt* __restrict__ d_dX_sorted, const int* __restrict__ global_id, const bool by_plane)
{
    static_assert(LANES == 4 || LANES == 8 || LANES == 16, "lanes per cell");
    constexpr int CELLS = coop::BLOCK / LANES, MAX_HITS = coop::MAX_HITS;
    constexpr int CAP = coop::Stage<Pt, LANES>::value;
    static_assert(CAP < 4096, "a listed hit is row << 12 | LDS index");
    constexpr int NF = N_floats<Pt>::value;
    constexpr int NC = NF + 4;  // components summed per cell: F (NF), friction, friction * old_v (3)
    constexpr int SLOTS = (NC + LANES - 1) / LANES;
    constexpr bool STAGE_V = LANES >= 8;  // old_v in LDS too where the launch is too small to hide L2
    __shared__ __attribute__((aligned(16))) Entry<Pt> sh_e[CAP + 4];  // slack: whole groups are read
    __shared__ float4 sh_v[STAGE_V ? CAP : 1];
    __shared__ unsigned short sh_hit[CELLS][MAX_HITS];  // row << 12 | LDS index of the hit
    __shared__ __attribute__((aligned(16))) float sh_term[CELLS][NC][LANES];  // one round's terms
    __shared__ float sh_sum[CELLS][NC];
    // row r of the stencil: the workgroup's slot range is [sh_lo[r], sh_lo[r] + sh_len[r]);
    // a cell's own candidates are the slots [sh_kb[cell][r], sh_ke[cell][r])
    __shared__ int sh_lo[9], sh_len[9], sh_shift[9], sh_kb[CELLS][9], sh_ke[CELLS][9];

    const int cell = threadIdx.x / LANES, lane = threadIdx.x % LANES;
    const int s0 = xcd_contiguous_tile(blockIdx.x, gridDim.x) * CELLS;
    const int s = s0 + cell;
    bool active = s < n;

    Pt Xi = ya::zero<Pt>();
    int i = 0, c = 0;
    if (active) {
        const Entry<Pt> self = sorted[s];
        Xi = self.X;
        i = self.id;
        c = cube_id[s];
        active = i < n_active;  // ghost cells of a slab decomposition get no force
    }
    if (!__syncthreads_or(active)) return;  // a workgroup of ghosts only
    const int gi = global_id && active ? global_id[i] : i;
    // The reference indexes cube_start/end without bounds checks (solvers.cuh:444);
    // out-of-grid cubes are treated as empty here.
    for (int r = lane; r < 9; r += LANES) {
        const int off = stencil_row_offset(r, gs);
        const int begin = offs[min(max(c + off - 1, 0), n_cubes)];
        sh_kb[cell][r] = begin;
        sh_ke[cell][r] = active ? offs[min(max(c + off + 2, 0), n_cubes)] : begin;
    }
    if (threadIdx.x < 9) {
        const int off = stencil_row_offset(threadIdx.x, gs);
        const int c_lo = cube_id[s0], c_hi = cube_id[min(s0 + CELLS, n) - 1];
        const int lo = offs[min(max(c_lo + off - 1, 0), n_cubes)];
        sh_lo[threadIdx.x] = lo;
        sh_len[threadIdx.x] = offs[min(max(c_hi + off + 2, 0), n_cubes)] - lo;
    }
    __syncthreads();
    int v0[10];  // row r sits at [v0[r], v0[r + 1]) of the nine ranges laid end to end
    v0[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; r++) v0[r + 1] = v0[r] + sh_len[r];
    if (threadIdx.x < 9) {
        int mine = 0;
#pragma unroll
        for (int r = 0; r < 9; r++) mine = (int)threadIdx.x == r ? v0[r] : mine;
        sh_shift[threadIdx.x] = sh_lo[threadIdx.x] - mine;  // position in the nine ranges -> sorted slot
    }
    __syncthreads();

    float acc[SLOTS];  // this lane's component sums (components lane, lane + LANES, ...)
    // by_plane (Grid_computer::sum_order = YA_SUM_BY_PLANE; grid_force_bits, "the tail"): the own plane's sums
    // (rows 0-2) are kept aside and the other planes' summed from +0; the two are added at the end.  A cell's
    // hit list never mixes the two: listing stops before row 3 until the own plane's list has been worked off.
    // Otherwise (the default, the reference's order) one running sum: own_done from the start, acc_own stays +0
    // (and +0 + x is x for every x a sum that started at +0 can hold: never -0).
    float acc_own[SLOTS];
#pragma unroll
    for (int a = 0; a < SLOTS; a++) acc[a] = 0.f, acc_own[a] = 0.f;
    bool own_done = !by_plane;
#define YA_COOP_OWN_DONE                                          \
    {                                                             \
        _Pragma("unroll") for (int a = 0; a < SLOTS; a++) acc_own[a] = acc[a], acc[a] = 0.f; \
        own_done = true;                                          \
    }

    const int rows_at_once = v0[9] <= CAP ? 9 : 3;
#pragma unroll 1
    for (int g0 = 0; g0 < 9; g0 += rows_at_once) {
        const int g_end = g0 + rows_at_once;
        if (g0 == 3 && !own_done) YA_COOP_OWN_DONE  // plane by plane: every list is worked off at the end of a chunk
        const int g_base = g0 == 0 ? 0 : (g0 == 3 ? v0[3] : v0[6]);
        const int g_total = (g_end == 9 ? v0[9] : (g_end == 3 ? v0[3] : v0[6])) - g_base;
#pragma unroll 1
        for (int chunk = 0; chunk < g_total; chunk += CAP) {
            const int chunk_n = min(CAP, g_total - chunk);
            const int base = g_base + chunk;  // where sh_e[0] sits in the nine ranges laid end to end
            __syncthreads();
            for (int t = threadIdx.x; t < chunk_n; t += coop::BLOCK) {
                const int v = base + t;
                int r = 0;
#pragma unroll
                for (int q = 1; q < 9; q++) r += v >= v0[q];
                const int slot = v + sh_shift[r];
                sh_e[t] = sorted[slot];
                if (STAGE_V) sh_v[t] = sorted_v[slot];
            }
            __syncthreads();

            // the cell's candidates of row r inside this chunk: LDS indices [sb, sb + len);
            // the next row's are requested while this one is worked on
            int r = g0, sb, len, k0 = 0, sb_next, len_next;
#define YA_COOP_ROW(row_, sb_, len_)                                                   \
    {                                                                                  \
        const int rr = min(row_, 8);                                                   \
        const int shift = sh_shift[rr] + base;                                         \
        sb_ = max(sh_kb[cell][rr] - shift, 0);                                         \
        len_ = (row_) < g_end ? max(min(sh_ke[cell][rr] - shift, chunk_n) - sb_, 0) : 0; \
    }
            YA_COOP_ROW(g0, sb, len)
            YA_COOP_ROW(g0 + 1, sb_next, len_next)
            int listed = 0;  // hits in the cell's list (the same in all its lanes)
            while (true) {
                // ---- A: stretches of rows until every cell of the wavefront is through or
                // its list could overflow ----
                while (true) {
                    const int row_limit = own_done ? g_end : min(g_end, 3);
                    const int stretch = r < row_limit ? min(len - k0, coop::STRETCH) : 0;
                    const bool go = r < row_limit && listed + stretch <= MAX_HITS;
                    if (!__any(go)) break;
                    if (go) {
                        // this lane's share [pb, pe) of the stretch: bit j of the mask, counted
                        // from the top, is candidate pb + j
                        const int share = (stretch + LANES - 1) / LANES;
                        const int pb = sb + k0 + min(lane * share, stretch);
                        const int pe = sb + k0 + min(lane * share + share, stretch);
                        unsigned m = 0;
                        int bits = 0;
                        for (int t = pb; t < pe; t += 4) {
                            float4 w[4];
                            float d2[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) w[u] = staged_words(&sh_e[t + u]);
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                d2[u] = dist2_to(Xi, w[u]);
                                keep_wide(w[u]);
                                d2[u] = t + u < pe ? d2[u] : INFINITY;
                            }
#pragma unroll
                            for (int u = 0; u < 4; u++) bits::shift_in(m, d2[u], cut2);
                            bits += 4;
                        }
                        m = bits > 0 ? m << (32 - bits) : 0u;
                        const int mine = __builtin_popcount(m);
                        int upto = mine;  // hits of this lane and of the cell's lower lanes
#pragma unroll
                        for (int o = 1; o < LANES; o <<= 1) {
                            const int up = __shfl_up(upto, o, LANES);
                            upto += lane >= o ? up : 0;
                        }
                        int rank = listed + upto - mine;
                        listed += __shfl(upto, LANES - 1, LANES);
                        while (m != 0) {
                            const int pos = __builtin_clz(m);
                            m &= 0x7fffffffu >> pos;
                            sh_hit[cell][rank++] = (unsigned short)((r << 12) | (pb + pos));
                        }
                        k0 += stretch;
                        while (k0 >= len && r < g_end) {
                            r++;
                            k0 = 0;
                            sb = sb_next;
                            len = len_next;
                            YA_COOP_ROW(r + 1, sb_next, len_next)
                        }
                    }
                }
                if (!__any(listed > 0)) {
                    if (!__any(r < g_end)) break;  // every cell is through all rows
                    if (rows_at_once == 9 && !own_done && r >= 3) YA_COOP_OWN_DONE  // no hit in the own plane's last rows
                    continue;
                }
                coop::wave_sync();
                // ---- B and C, LANES hits per round.  Lanes past the end of the list leave zero
                // terms: a sum that starts at +0 never is -0, so adding +0 changes no bit ----
                for (int h0 = 0; __any(h0 < listed); h0 += LANES) {
                    const int h = h0 + lane;
                    Pt f = ya::zero<Pt>();
                    float friction = 0;
                    float4 v{0.f, 0.f, 0.f, 0.f};
                    if (h < listed) {
                        const int e = sh_hit[cell][h];
                        const int t = e & 0xfff;
                        const Entry<Pt> other = sh_e[t];
                        if (STAGE_V)
                            v = sh_v[t];
                        else
                            v = sorted_v[(unsigned)(t + base + sh_shift[e >> 12])];
                        Pt rr = Xi - other.X;
                        float dist = dist3(rr.x, rr.y, rr.z);
                        const int j = global_id ? global_id[other.id] : other.id;
                        YA_CALL_INLINED f = pw_int(Xi, rr, dist, gi, j);
                        friction = pw_friction(Xi, rr, dist, gi, j);
                    }
#pragma unroll
                    for (int q = 0; q < NF; q++) sh_term[cell][q][lane] = field(f, q);
                    sh_term[cell][NF][lane] = friction;
                    // the old_v term only where the friction is not zero (solvers.cuh:454-458)
                    sh_term[cell][NF + 1][lane] = friction != 0 ? friction * v.x : 0.f;
                    sh_term[cell][NF + 2][lane] = friction != 0 ? friction * v.y : 0.f;
                    sh_term[cell][NF + 3][lane] = friction != 0 ? friction * v.z : 0.f;
                    coop::wave_sync();
#pragma unroll
                    for (int a = 0; a < SLOTS; a++) {
                        const int q = lane + LANES * a;
                        if (q < NC) {
                            float sum = acc[a];
                            const float4* terms = reinterpret_cast<const float4*>(&sh_term[cell][q][0]);
#pragma unroll
                            for (int u = 0; u < LANES / 4; u++) {
                                const float4 p = terms[u];
                                sum += p.x;
                                sum += p.y;
                                sum += p.z;
                                sum += p.w;
                            }
                            acc[a] = sum;
                        }
                    }
                    coop::wave_sync();
                }
                listed = 0;
                if (rows_at_once == 9 && !own_done && r >= 3) YA_COOP_OWN_DONE
            }
#undef YA_COOP_ROW
        }
    }
#pragma unroll
    for (int a = 0; a < SLOTS; a++) {
        const int q = lane + LANES * a;
        if (q < NC) sh_sum[cell][q] = acc_own[a] + acc[a];
    }
#undef YA_COOP_OWN_DONE
    coop::wave_sync();
    if (active && lane == 0) {
        Pt F;
#pragma unroll
        for (int q = 0; q < NF; q++) field(F, q) = sh_sum[cell][q];
        const Pt dX = store_rhs(d_dX, i, has_gen, F,
            float3{sh_sum[cell][NF + 1], sh_sum[cell][NF + 2], sh_sum[cell][NF + 3]}, sh_sum[cell][NF]);
        if (d_dX_sorted) d_dX_sorted[s] = dX;  // for the sorted-space Euler stage
    }
}

#undef YA_ROW_BOUNDS

// Gabriel-graph force (replaces compute_cube_gabriel, solvers.cuh:509-602): the
// candidates inside the cut-off are collected per thread, ordered by distance,
// and the pair (i, j) only interacts if no closer candidate lies inside the
// sphere around the midpoint of i and j with radius coefficient * dist / 2.
// Kept straightforward (thread-private lists in scratch memory): it is off the
// benchmarked path and only used by models that ask for Gabriel_solver.
constexpr int GABRIEL_MAX_NEIGHBOURS = 100;  // the reference's fixed list size

template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
__global__ __launch_bounds__(64) void gabriel_force(const int n,
    const Entry<Pt>* __restrict__ sorted, const float4* __restrict__ sorted_v,
    const int* __restrict__ cube_id, const int* __restrict__ offs, const int gs,
    const int n_cubes, const float cube_size, const float gabriel_coefficient,
    Pt* __restrict__ d_dX, const bool has_gen)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;

    const Entry<Pt> self = sorted[s];
    const Pt Xi = self.X;
    const int i = self.id;
    const int c = cube_id[s];

    int slot[GABRIEL_MAX_NEIGHBOURS];
    float distance[GABRIEL_MAX_NEIGHBOURS];
    int n_neighs = 0;
    for (int row = 0; row < 9; row++) {
        const int mid = c + stencil_row_offset(row, gs);
        const int k_end = offs[min(max(mid + 2, 0), n_cubes)];
        for (int k = offs[min(max(mid - 1, 0), n_cubes)]; k < k_end; k++) {
            const Pt r = Xi - sorted[k].X;
            const float dist = dist3(r.x, r.y, r.z);
            if (dist >= cube_size) continue;
            D_ASSERT(n_neighs < GABRIEL_MAX_NEIGHBOURS);
            slot[n_neighs] = k;
            distance[n_neighs] = dist;
            n_neighs++;
        }
    }
    // selection sort by distance, closest first (solvers.cuh:550-566)
    for (int m = 0; m < n_neighs - 1; m++) {
        int closest = m;
        for (int q = m + 1; q < n_neighs; q++)
            if (distance[q] < distance[closest]) closest = q;
        if (closest != m) {
            const int ts = slot[closest];
            slot[closest] = slot[m];
            slot[m] = ts;
            const float td = distance[closest];
            distance[closest] = distance[m];
            distance[m] = td;
        }
    }
    // farthest first: keep (i, j) unless a closer candidate sits in its Gabriel sphere
    Pt F = ya::zero<Pt>();
    float3 sum_v{0.f, 0.f, 0.f};
    float sum_friction = 0;
    for (int m = n_neighs - 1; m >= 0; m--) {
        const Entry<Pt> other = sorted[slot[m]];
        const int j = other.id;
        const float dist = distance[m];
        bool keep = true;
        if (j != i) {
            const float radius = 0.5f * dist * gabriel_coefficient;
            const Pt mid_point = 0.5f * (Xi + other.X);
            for (int q = m - 1; q >= 0; q--) {
                const Pt r_mk = mid_point - sorted[slot[q]].X;
                if (dist3(r_mk.x, r_mk.y, r_mk.z) < radius) {
                    keep = false;
                    break;
                }
            }
        }
        if (!keep) continue;
        const Pt r = Xi - other.X;
        F += pw_int(Xi, r, dist, i, j);
        const float friction = pw_friction(Xi, r, dist, i, j);
        sum_friction += friction;
        if (friction != 0) {
            const float4 v = sorted_v[slot[m]];
            sum_v.x += friction * v.x;
            sum_v.y += friction * v.y;
            sum_v.z += friction * v.z;
        }
    }
    store_rhs(d_dX, i, has_gen, F, sum_v, sum_friction);
}

// fix = what is subtracted from dX.xyz: 0 = mean (already in d_mean), 1 = the
// fixed point's value, 2 = the point's x, y and the mean's z (set_fixed_xy,
// solvers.cuh:243-249).
template<typename Pt>
__global__ void make_fix(int mode, const float* __restrict__ d_mean,
    const Pt* __restrict__ d_point, float* __restrict__ d_fix)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (mode == 1) {
        d_fix[0] = d_point->x;
        d_fix[1] = d_point->y;
        d_fix[2] = d_point->z;
    } else {
        d_fix[0] = d_point->x;
        d_fix[1] = d_point->y;
        d_fix[2] = d_mean[2];
    }
}

}  // namespace ya


// 2nd order solver for the equation v = F + <v(t - dt)> for x, y, and z, where
// <v> is the mean velocity of the neighbours weighted by the friction
// coefficients; other variables in Pt follow dw/dt = F_w (solvers.cuh:109-144).
// The fixed velocity is read from device memory instead of a kernel argument.
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void euler_step(const int n, const float dt,
    const Pt* __restrict__ d_X0, const float* __restrict__ d_fix, Pt* __restrict__ d_dX,
    Pt* __restrict__ d_X)
{
    const int i = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (i >= n) return;

    Pt dX = d_dX[i];
    dX.x -= d_fix[0];
    dX.y -= d_fix[1];
    dX.z -= d_fix[2];
    d_dX[i] = dX;
    d_X[i] = d_X0[i] + dX * dt;
}

template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void heun_step(const int n, const float dt,
    const Pt* __restrict__ d_dX, const float* __restrict__ d_fix1, Pt* __restrict__ d_dX1,
    Pt* __restrict__ d_X, float3* __restrict__ d_old_v)
{
    const int i = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (i >= n) return;

    Pt dX1 = d_dX1[i];
    dX1.x -= d_fix1[0];
    dX1.y -= d_fix1[1];
    dX1.z -= d_fix1[2];
    d_dX1[i] = dX1;
    const Pt dX = d_dX[i];
    Pt X = d_X[i];
    X += (dX + dX1) * 0.5 * dt;
    d_X[i] = X;
    d_old_v[i] = float3{
        (dX.x + dX1.x) * 0.5f, (dX.y + dX1.y) * 0.5f, (dX.z + dX1.z) * 0.5f};
}


// The same two updates for the sorted-space pipeline of Grid_solver (no generic
// forces): the predictor X1 = X0 + (dX - fix) dt is applied in place to the
// cube-sorted copy of the cells, which then feeds the second grid build without
// any gather; d_dX keeps the raw right-hand side and the corrector subtracts both
// fixed velocities itself.  Statement for statement the arithmetic of
// euler_step / heun_step above.
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void euler_step_sorted(const int n, const float dt,
    const float* __restrict__ d_fix, const Pt* __restrict__ d_dX_sorted,
    ya::Entry<Pt>* __restrict__ d_sorted, const int n_active)
{
    const int s = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (s >= n) return;
    if (d_sorted[s].id >= n_active) return;  // a ghost cell: moved by its owner

    Pt dX = d_dX_sorted[s];
    dX.x -= d_fix[0];
    dX.y -= d_fix[1];
    dX.z -= d_fix[2];
    ya::Entry<Pt> e = d_sorted[s];
    e.X = e.X + dX * dt;
    d_sorted[s] = e;
}

// z-slab decomposition, no generic forces: own AND mirrored cells moved inside the sorted copy
// in one pass.  A mirrored cell (id >= n_active) has no right-hand side in sorted order -- it
// arrived by message as row id of d_dX -- and the sorted copy holds its X[id] bit for bit, so
// X[id] + (dX[id] - fix) dt is computed here exactly as euler_step computes it.  d_dX stays raw
// (heun_step_raw subtracts both fixed velocities), d_X1 is not written at all.
// The fixed velocity is taken from the stage's ALL-REDUCED totals (ya_slab_pack on every rank,
// summed): {sum[n_floats], count in two pieces, two votes, the fixed point's right-hand side}.
// fix_mode 0 (set_fixed()): fix = sum * float(1. / n), the reference's Pt / n arithmetic
// (dtypes.cuh:202-217; n through binary32 as there); 1 (set_fixed(i)): the fixed point's value,
// which only its owner put into the sum; 2 (set_fixed_xy(i), first stage): its x and y, the mean's z
// (solvers.cuh:241-253).  Computed by every thread alike; thread 0 leaves it in d_fix_out for the
// corrector.
namespace ya {
__device__ __forceinline__ float3 fix_from_total(const float* __restrict__ total, const int n_floats, const int fix_mode = 0)
{
    const double n = (double)total[n_floats] + 4096. * (double)total[n_floats + 1];
    const float inv = (float)(1. / (double)(float)n);
    const float3 mean{total[0] * inv, total[1] * inv, total[2] * inv};
    if (fix_mode == 0) return mean;
    const float* point = total + n_floats + 4;
    return float3{point[0], point[1], fix_mode == 1 ? point[2] : mean.z};
}
// The drift guard of a z-slab weighs a cell's movement by where it was when the mirrored cells were
// chosen: fully within `width` = halo + limit of a face of the slab (those cells are mirrored, or
// next in line), half elsewhere -- a cell further away has to cover `limit` more before it can
// matter, so the guard's bound `limit` on the weighted maximum allows it 2 limit
// (include/slab_logic.inc, guard_between_stages).
struct Guard_band {
    float lo_face = -INFINITY, hi_face = INFINITY, width = 0.f;
    __host__ __device__ float weight(const float z) const
    {
        return fabsf(z - lo_face) <= width || fabsf(z - hi_face) <= width ? 1.f : 0.5f;
    }
};
// The largest v of the workgroup (v >= 0, or NaN: counted as +inf) folded into one of
// GUARD_SLOTS slots of `partial` by a non-returning atomic max (non-negative binary32 values order
// like their bit patterns); every thread of an UPDATE_BLOCK-wide workgroup must call it.  The slots
// are read, and zeroed again, by the reduction that follows (ya_slab_pack).  For the drift guard of
// a z-slab: thousands of workgroups, a few dozen atomics per slot, one short list to fold.
constexpr int GUARD_SLOTS = 256;
__device__ __forceinline__ void block_max_to(float v, float* __restrict__ partial)
{
    __shared__ float sh_max[UPDATE_BLOCK / 64];
    v = v == v ? v : INFINITY;
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh_max[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < UPDATE_BLOCK / 64; w++) v = fmaxf(v, sh_max[w]);
        if (v > 0.f) atomicMax(reinterpret_cast<unsigned*>(partial) + (blockIdx.x % GUARD_SLOTS), __float_as_uint(v));
    }
}
}  // namespace ya

// (pred_partial, may be NULL: per workgroup the largest |z moved by this predictor| -- the drift
// guard of the decomposition, include/slab_logic.inc)
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void euler_step_sorted_mirrored(const int n, const float dt,
    const float* __restrict__ d_total, float* __restrict__ d_fix_out, const Pt* __restrict__ d_dX_sorted,
    const Pt* __restrict__ d_dX, ya::Entry<Pt>* __restrict__ d_sorted, const int n_active, const int fix_mode,
    float* __restrict__ pred_partial, const ya::Guard_band band)
{
    const int s = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    const float3 fix = ya::fix_from_total(d_total, sizeof(Pt) / sizeof(float), fix_mode);
    if (s == 0) {
        d_fix_out[0] = fix.x;
        d_fix_out[1] = fix.y;
        d_fix_out[2] = fix.z;
    }
    float moved = 0.f;
    if (s < n) {
        ya::Entry<Pt> e = d_sorted[s];
        Pt dX = e.id >= n_active ? d_dX[e.id] : d_dX_sorted[s];
        dX.x -= fix.x;
        dX.y -= fix.y;
        dX.z -= fix.z;
        const float z0 = e.X.z;
        e.X = e.X + dX * dt;
        moved = fabsf(e.X.z - z0) * band.weight(z0);
        d_sorted[s] = e;
    }
    if (pred_partial) ya::block_max_to(moved, pred_partial);
}

// z-slab decomposition: the ghost cells' predictor positions arrive from the slab
// neighbours in original order (d_X1[id], id >= n_active) and are put into the sorted
// copy here; the own cells were moved by euler_step_sorted.
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void ghosts_into_sorted(const int n,
    const int n_active, const Pt* __restrict__ d_X1, ya::Entry<Pt>* __restrict__ d_sorted)
{
    const int s = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (s >= n) return;
    const int id = d_sorted[s].id;
    if (id >= n_active) d_sorted[s].X = d_X1[id];
}

// Round 5: the update kernels of the sorted-space step fold the reduction's partial sums THEMSELVES.
// ya_reduce_mean was two launches -- B per-block partial sums, then one workgroup that folds them
// and scales by 1 / n -- and the second one is a whole launch (4-5 us, four times per step at any
// system size) for a few hundred additions.  Every workgroup of the kernel that needs the mean now
// repeats that fold from the partials (<= 1024 x 3 floats, L2 hits): the same additions in the same
// tree (lane t takes partials t, t + 256, ...; lanes folded by halving), the same `sum * float(1. / n)`
// (dtypes.cuh:202-217), hence the same bits as ya_reduce_mean leaves in d_mean -- only x, y, z, all the
// update kernels subtract (solvers.cuh:113-144).
namespace ya {
template<int NF>
__device__ __forceinline__ float3 fixed_velocity_from_partials(const float* __restrict__ partials, const int n_partials,
    const int n)
{
    static_assert(UPDATE_BLOCK == 256, "the fold is libyalla_hip.so's fold256");
    __shared__ float sh[3 * UPDATE_BLOCK];
    float acc[3] = {0.f, 0.f, 0.f};
    for (int p = threadIdx.x; p < n_partials; p += UPDATE_BLOCK) {
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] = acc[k] + partials[(size_t)p * NF + k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) sh[k * UPDATE_BLOCK + threadIdx.x] = acc[k];
    __syncthreads();
    if ((int)threadIdx.x < 128) {
#pragma unroll
        for (int k = 0; k < 3; k++)
            sh[k * UPDATE_BLOCK + threadIdx.x] = sh[k * UPDATE_BLOCK + threadIdx.x] + sh[k * UPDATE_BLOCK + threadIdx.x + 128];
    }
    __syncthreads();
    if ((int)threadIdx.x < 64) {
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = sh[k * UPDATE_BLOCK + threadIdx.x] + sh[k * UPDATE_BLOCK + threadIdx.x + 64];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; k++) v[k] = v[k] + __shfl_down(v[k], s, 64);
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) sh[k * UPDATE_BLOCK] = v[k];
        }
    }
    __syncthreads();
    const float inv = (float)(1. / (double)(float)n);  // Pt / n == Pt * float(1. / float(n))
    return float3{sh[0] * inv, sh[UPDATE_BLOCK] * inv, sh[2 * UPDATE_BLOCK] * inv};
}
}  // namespace ya

// euler_step_sorted with the fixed velocity = the mean of the stage's right-hand sides, folded here from
// the reduction's partial sums; workgroup 0 leaves it in d_fix_out for the corrector.
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void euler_step_sorted_folding(const int n, const float dt,
    const float* __restrict__ partials, const int n_partials, float* __restrict__ d_fix_out,
    const Pt* __restrict__ d_dX_sorted, ya::Entry<Pt>* __restrict__ d_sorted)
{
    const float3 fix = ya::fixed_velocity_from_partials<ya::N_floats<Pt>::value>(partials, n_partials, n);
    const int s = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (s == 0) {
        d_fix_out[0] = fix.x;
        d_fix_out[1] = fix.y;
        d_fix_out[2] = fix.z;
    }
    if (s >= n) return;
    Pt dX = d_dX_sorted[s];
    dX.x -= fix.x;
    dX.y -= fix.y;
    dX.z -= fix.z;
    ya::Entry<Pt> e = d_sorted[s];
    e.X = e.X + dX * dt;
    d_sorted[s] = e;
}

// euler_step / heun_step (the pipeline through d_X / d_X1: generic forces, Tile_solver, Gabriel_solver) with the
// fixed velocity folded here; euler_step_folding leaves it in d_fix_out.
// (d_sorted, may be NULL: the cube-sorted copy's predictor in the same launch, euler_step_sorted's statements for
// slot i; d_zero, may be NULL: the NEXT stage's right-hand side array, whose row i is dead by now and is left
// zeroed for the generic forces that will add to it -- a memset launch less)
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void euler_step_folding(const int n, const float dt,
    const Pt* __restrict__ d_X0, const float* __restrict__ partials, const int n_partials, float* __restrict__ d_fix_out,
    Pt* __restrict__ d_dX, Pt* __restrict__ d_X, const Pt* __restrict__ d_dX_sorted, ya::Entry<Pt>* __restrict__ d_sorted,
    Pt* __restrict__ d_zero)
{
    const float3 fix = ya::fixed_velocity_from_partials<ya::N_floats<Pt>::value>(partials, n_partials, n);
    const int i = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (i == 0) {
        d_fix_out[0] = fix.x;
        d_fix_out[1] = fix.y;
        d_fix_out[2] = fix.z;
    }
    if (i >= n) return;

    if (d_sorted) {
        Pt dXs = d_dX_sorted[i];
        dXs.x -= fix.x;
        dXs.y -= fix.y;
        dXs.z -= fix.z;
        ya::Entry<Pt> e = d_sorted[i];
        e.X = e.X + dXs * dt;
        d_sorted[i] = e;
    }
    Pt dX = d_dX[i];
    dX.x -= fix.x;
    dX.y -= fix.y;
    dX.z -= fix.z;
    d_dX[i] = dX;
    d_X[i] = d_X0[i] + dX * dt;
    if (d_zero) d_zero[i] = ya::zero<Pt>();
}

// (zero_dX: row i of d_dX, dead after this, is left zeroed for the next step's generic forces)
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void heun_step_folding(const int n, const float dt,
    Pt* __restrict__ d_dX, const float* __restrict__ partials1, const int n_partials1, Pt* __restrict__ d_dX1,
    Pt* __restrict__ d_X, float3* __restrict__ d_old_v, const bool zero_dX)
{
    const float3 fix1 = ya::fixed_velocity_from_partials<ya::N_floats<Pt>::value>(partials1, n_partials1, n);
    const int i = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (i >= n) return;

    Pt dX1 = d_dX1[i];
    dX1.x -= fix1.x;
    dX1.y -= fix1.y;
    dX1.z -= fix1.z;
    d_dX1[i] = dX1;
    const Pt dX = d_dX[i];
    Pt X = d_X[i];
    X += (dX + dX1) * 0.5 * dt;
    d_X[i] = X;
    d_old_v[i] = float3{
        (dX.x + dX1.x) * 0.5f, (dX.y + dX1.y) * 0.5f, (dX.z + dX1.z) * 0.5f};
    if (zero_dX) d_dX[i] = ya::zero<Pt>();
}

// heun_step_raw with the second stage's fixed velocity folded from its partial sums.
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void heun_step_raw_folding(const int n, const float dt,
    const Pt* __restrict__ d_dX, const float* __restrict__ d_fix, const Pt* __restrict__ d_dX1,
    const float* __restrict__ partials1, const int n_partials1, Pt* __restrict__ d_X, float3* __restrict__ d_old_v)
{
    const float3 fix1 = ya::fixed_velocity_from_partials<ya::N_floats<Pt>::value>(partials1, n_partials1, n);
    const int i = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (i >= n) return;

    Pt dX = d_dX[i];
    dX.x -= d_fix[0];
    dX.y -= d_fix[1];
    dX.z -= d_fix[2];
    Pt dX1 = d_dX1[i];
    dX1.x -= fix1.x;
    dX1.y -= fix1.y;
    dX1.z -= fix1.z;
    Pt X = d_X[i];
    X += (dX + dX1) * 0.5 * dt;
    d_X[i] = X;
    d_old_v[i] = float3{
        (dX.x + dX1.x) * 0.5f, (dX.y + dX1.y) * 0.5f, (dX.z + dX1.z) * 0.5f};
}

template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void heun_step_raw(const int n, const float dt,
    const Pt* __restrict__ d_dX, const float* __restrict__ d_fix, const Pt* __restrict__ d_dX1,
    const float* __restrict__ d_fix1, Pt* __restrict__ d_X, float3* __restrict__ d_old_v)
{
    const int i = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (i >= n) return;

    Pt dX = d_dX[i];
    dX.x -= d_fix[0];
    dX.y -= d_fix[1];
    dX.z -= d_fix[2];
    Pt dX1 = d_dX1[i];
    dX1.x -= d_fix1[0];
    dX1.y -= d_fix1[1];
    dX1.z -= d_fix1[2];
    Pt X = d_X[i];
    X += (dX + dX1) * 0.5 * dt;
    d_X[i] = X;
    d_old_v[i] = float3{
        (dX.x + dX1.x) * 0.5f, (dX.y + dX1.y) * 0.5f, (dX.z + dX1.z) * 0.5f};
}


// heun_step_raw with the second stage's fixed velocity taken from the stage's all-reduced totals
// (z-slab decomposition; see euler_step_sorted_mirrored).  z_selected / moved_partial (may be NULL):
// per workgroup the largest |z - z when the mirrored cells were chosen| after this update, weighted
// by the band.  votes_out (may be NULL; host memory the device can write): the all-reduced votes of
// this stage, d_total1[n_floats + 2 .. + 3], for the host to collect when the next step begins.
template<typename Pt>
__global__ __launch_bounds__(ya::UPDATE_BLOCK) void heun_step_raw_total(const int n, const float dt,
    const Pt* __restrict__ d_dX, const float* __restrict__ d_fix, const Pt* __restrict__ d_dX1,
    const float* __restrict__ d_total1, Pt* __restrict__ d_X, float3* __restrict__ d_old_v, const int fix_mode,
    const float* __restrict__ z_selected, float* __restrict__ moved_partial, const ya::Guard_band band,
    float* __restrict__ votes_out)
{
    const int i = blockIdx.x * ya::UPDATE_BLOCK + threadIdx.x;
    if (votes_out && i == 0) {
        votes_out[0] = d_total1[sizeof(Pt) / sizeof(float) + 2];
        votes_out[1] = d_total1[sizeof(Pt) / sizeof(float) + 3];
    }
    float moved = 0.f;
    if (i < n) {
        const float3 fix1 = ya::fix_from_total(d_total1, sizeof(Pt) / sizeof(float), fix_mode);

        Pt dX = d_dX[i];
        dX.x -= d_fix[0];
        dX.y -= d_fix[1];
        dX.z -= d_fix[2];
        Pt dX1 = d_dX1[i];
        dX1.x -= fix1.x;
        dX1.y -= fix1.y;
        dX1.z -= fix1.z;
        Pt X = d_X[i];
        X += (dX + dX1) * 0.5 * dt;
        d_X[i] = X;
        d_old_v[i] = float3{
            (dX.x + dX1.x) * 0.5f, (dX.y + dX1.y) * 0.5f, (dX.z + dX1.z) * 0.5f};
        if (z_selected) {
            const float z0 = z_selected[i];
            moved = fabsf(X.z - z0) * band.weight(z0);
        }
    }
    if (moved_partial) ya::block_max_to(moved, moved_partial);
}


// ---- renumbering (opt-in, not in the reference) ---------------------------------------------------
// Solution::renumber(properties ...) gives the cells new ids in cube order: the cell in slot s of
// a fresh grid build becomes cell s.  The reference's contract is that ids never change -- models
// index their own per-cell arrays by id (d_type[i], examples/passive_growth.cu:48-51) and append
// daughters at d_X[n] -- so a model that has grown by division holds its cells in birth order,
// spatially random, and every access of a pairwise functor to such an array (four per interacting
// pair in passive_growth's relu_w_epithelium) is 64 different cache lines per wavefront
// instruction: BASELINE configuration 4 moved 6.7 GB between L2 and the fabric per force launch
// where 70 MB are needed, 78 % of its wave cycles waiting (profiles/r03_pmc_cfg4_*).  Only the
// model knows all arrays that are indexed by cell id, so only the model can ask for a
// renumbering, handing over every such array:
//
//     if (step % 10 == 0) cells.renumber(type, n_mes_nbs, n_epi_nbs);   // Property<...>&, Links&, or T* d_array
//
// Afterwards cell s is what cell order[s] was (d_X, d_old_v and the arrays handed over are
// permuted alike, link endpoints are renamed), consecutive ids are neighbours in space, and so
// are the daughters a proliferation kernel appends later in the order of their mothers.  Host
// mirrors (h_X, h_prop) are stale until the next copy_to_host.  Results: every per-cell sum is
// still accumulated cube by cube in ascending id, and the stable sort keeps the order of ids
// inside a cube across the renumbering, so forces change only through cells that later move
// between cubes (rounding of reordered sums); a model that draws random numbers by cell id draws
// others.  The CPU oracle offers the same call (oracle/yalla_host.hpp) and both backends stay in
// lock-step through it (tests/test_growth.py).
namespace ya {
template<typename T>
__global__ __launch_bounds__(UPDATE_BLOCK) void permute_rows(const int n, const int* __restrict__ order,
    const T* __restrict__ src, T* __restrict__ dst)
{
    const int s = blockIdx.x * UPDATE_BLOCK + threadIdx.x;
    if (s < n) dst[s] = src[order[s]];
}
__global__ __launch_bounds__(UPDATE_BLOCK) void invert_order(const int n, const int* __restrict__ order,
    int* __restrict__ new_id)
{
    const int s = blockIdx.x * UPDATE_BLOCK + threadIdx.x;
    if (s < n) new_id[order[s]] = s;
}
// link endpoints (two ints per link: links.cuh Link{a, b}) renamed through new_id; ids >= n (none in a
// consistent model) are left alone
__global__ __launch_bounds__(UPDATE_BLOCK) void rename_ids(const int n_ids, const int* __restrict__ d_n_links,
    const int n_cells, const int* __restrict__ new_id, int* __restrict__ ids)
{
    const int k = blockIdx.x * UPDATE_BLOCK + threadIdx.x;
    if (k >= n_ids || k >= 2 * *d_n_links) return;
    const int i = ids[k];
    if (i >= 0 && i < n_cells) ids[k] = new_id[i];
}
}  // namespace ya


// Solution<Pt, Solver> combines a method, Solver, with a point type, Pt: host
// mirror of the variables plus access to the device arrays (solvers.cuh:56-106).
template<typename Pt, template<typename> class Solver>
class Solution : public Solver<Pt> {
public:
    Pt* h_X;                                      // Current variables on host
    Pt* const d_X = Solver<Pt>::d_X;              // Variables on device (GPU)
    float3* const d_old_v = Solver<Pt>::d_old_v;  // Velocities from previous step
    int* const h_n = (int*)malloc(sizeof(int));   // Number of points
    int* const d_n = Solver<Pt>::d_n;
    const int n_max;
    template<typename... Args>
    Solution(int n_max, Args... args) : Solver<Pt>{n_max, args...}, n_max{n_max}
    {
        *h_n = n_max;
        // page-locked if the runtime grants it: a frame's copy_to_host moves all n_max points
        // (26 MB at config 4); zeroed like calloc'ed memory
        const size_t bytes = (size_t)n_max * sizeof(Pt);
        h_X_locked = n_max > 0 && ya_host_alloc((void**)&h_X, bytes) == 0;
        if (h_X_locked)
            memset((void*)h_X, 0, bytes);
        else
            h_X = (Pt*)calloc(n_max, sizeof(Pt));
    }
    ~Solution()
    {
        if (h_X_locked)
            (void)ya_host_free(h_X);
        else
            free(h_X);
        free(h_n);
    }
    Solution(const Solution&) = delete;

private:
    bool h_X_locked = false;

public:
    void copy_to_device()
    {
        assert(*h_n <= n_max);
        YA_CHECK(ya_memcpy_h2d(d_X, h_X, (size_t)n_max * sizeof(Pt)));
        YA_CHECK(ya_memcpy_h2d(d_n, h_n, sizeof(int)));
    }
    void copy_to_host()
    {
        YA_CHECK(ya_memcpy_d2h(h_X, d_X, (size_t)n_max * sizeof(Pt)));
        YA_CHECK(ya_memcpy_d2h(h_n, d_n, sizeof(int)));
        assert(*h_n <= n_max);
        Solver<Pt>::check_status();
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
    // The former two-argument generic force `(const Pt* d_X, Pt* d_dX)`, which the
    // reference's own tests still use (tests/test_solvers.cu:141, test_links.cu:19).
    template<Pairwise_interaction<Pt> pw_int, typename Gen2,
        typename = decltype(std::declval<Gen2&>()((const Pt*)nullptr, (Pt*)nullptr))>
    void take_step(float dt, Gen2 gen_forces)
    {
        take_step<pw_int>(dt, Generic_forces<Pt>{[gen_forces](const int, const Pt* d_X,
                                                     Pt* d_dX) mutable { gen_forces(d_X, d_dX); }});
    }
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction, typename Gen2,
        typename = decltype(std::declval<Gen2&>()((const Pt*)nullptr, (Pt*)nullptr))>
    void take_step(float dt, Gen2 gen_forces)
    {
        take_step<pw_int, pw_friction>(dt,
            Generic_forces<Pt>{[gen_forces](const int, const Pt* d_X, Pt* d_dX) mutable {
                gen_forces(d_X, d_dX);
            }});
    }
};


// The three-parameter spelling `Solution<Pt, n_max, Solver>` of older ya||a model files (and of this
// build's north_star): the capacity as a template argument, objects default-constructed.  C++ cannot
// overload a class template on the KIND of its parameters, so beside the reference's two-parameter
// `Solution<Pt, Solver>{n_max, ...}` (every example at the surveyed commit) it has a name of its own:
//     Solution_n<float3, 800, Tile_solver> bodies;            // = Solution<float3, Tile_solver>{800}
//     Solution_n<float3, 100000, Grid_solver> cells{64, 1.f}; // solver arguments follow as usual
template<typename Pt, int N_MAX, template<typename> class Solver>
class Solution_n : public Solution<Pt, Solver> {
public:
    static constexpr int capacity = N_MAX;
    template<typename... Args>
    Solution_n(Args... args) : Solution<Pt, Solver>{N_MAX, args...}
    {}
};


// Two-stage Heun integrator (solvers.cuh:164-276).  Computer specifies how
// pairwise interactions are computed.
template<typename Pt, template<typename> class Computer>
class Heun_solver : public Computer<Pt> {
public:
    template<typename... Args>
    Heun_solver(int n_max, Args... args) : Computer<Pt>{n_max, args...}, n_max{n_max}
    {
        const size_t pts = (size_t)n_max * sizeof(Pt);
        YA_CHECK(ya_malloc((void**)&d_X, pts));
        YA_CHECK(ya_malloc((void**)&d_dX, pts));
        YA_CHECK(ya_malloc((void**)&d_X1, pts));
        YA_CHECK(ya_malloc((void**)&d_dX1, pts));
        YA_CHECK(ya_malloc((void**)&d_old_v, (size_t)n_max * sizeof(float3)));
        YA_CHECK(ya_memset_async(d_old_v, 0, (size_t)n_max * sizeof(float3), nullptr));
        YA_CHECK(ya_malloc((void**)&d_n, sizeof(int)));
        YA_CHECK(ya_malloc((void**)&d_mean, 2 * sizeof(Pt)));
        YA_CHECK(ya_malloc((void**)&d_mean_first, 2 * sizeof(Pt)));
        YA_CHECK(ya_n_reader_create(&n_reader));
        YA_CHECK(ya_malloc((void**)&d_fix, 4 * sizeof(float)));
        YA_CHECK(ya_malloc((void**)&d_fix_first, 4 * sizeof(float)));
        YA_CHECK(ya_malloc((void**)&d_workspace, ya_reduce_workspace_bytes(n_floats)));
    }
    ~Heun_solver()
    {
        ya_free(d_X);
        ya_free(d_dX);
        ya_free(d_X1);
        ya_free(d_dX1);
        ya_free(d_old_v);
        ya_free(d_n);
        ya_free(d_mean);
        ya_free(d_mean_first);
        ya_n_reader_destroy(n_reader);
        ya_free(d_fix);
        ya_free(d_fix_first);
        ya_free(d_workspace);
        if (renumber_scratch) ya_free(renumber_scratch);
        drop_graph();
        if (capture_stream) (void)hipStreamDestroy(capture_stream);
    }
    Heun_solver(const Heun_solver&) = delete;
    void set_fixed() { fix_com = true; }
    void set_fixed(int point_id)
    {
        fix_com = false;
        fix_point = point_id;
    }
    void set_fixed_xy(int point_id)
    {
        fix_com = false;
        fix_com_z = true;
        fix_point = point_id;
    }
    // Opt-in, not in the reference: new cell ids in cube order (see "renumbering" above).  Hand
    // over EVERY array the model indexes by cell id: Property<T>& (anything with a d_prop member),
    // Links& (anything with d_link / d_n: the endpoints are renamed), or a plain device pointer
    // T* of >= n elements.  No-op for solvers without a grid (Tile_solver).
    template<typename... Arrays>
    void renumber(Arrays&... arrays)
    {
        const int n = get_d_n();
        if (n <= 0) return;
        const int* order = Computer<Pt>::cube_order(n, d_X);  // order[s] = the id of the cell in slot s
        if (!order) return;
        permute_array(order, n, d_X);
        permute_array(order, n, d_old_v);
        int* new_id = nullptr;
        const int unused[] = {0, (renumber_one(order, n, arrays, new_id), 0)...};
        (void)unused;
        if (new_id) ya_free(new_id);
        rhs_zeroed[0] = rhs_zeroed[1] = 0;  // (nothing here writes d_dX / d_dX1; no promise is carried over a renumbering)
        Computer<Pt>::ids_changed();
        // a graph captured NOW would bake the forgotten visit order (n_prev = 0) into its first build:
        // capture again one step later, when the grid remembers an order again
        drop_graph();
        last_key = Step_key{};
    }

    // ... or ONE line after the model has made its arrays, instead of a call in its loop (round 5):
    //     cells.keep_in_cube_order(10, type, n_mes_nbs, n_epi_nbs, links, d_state);
    // -- every `every`-th take_step (the next one first) begins with renumber(arrays...).  The arrays are
    // taken by reference: they must outlive the solver's steps, as they must in a loop that calls
    // renumber itself.  every <= 0 switches it off again.  Cells a model kernel appended since the last
    // renumbering (proliferation) are put in their places by the next one.
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

    // Replaying the step as one hipGraph (Grid_solver without generic forces), opt-in:
    // 1 = whenever possible, -1 = for systems below YA_GRAPH_MAX_CELLS cells, 0 (default) =
    // never.  The graph is captured the second time the same step (functors, n, dt, fixed
    // point, cube size, force kernel) is asked for and replayed while that stays so; a
    // changing n (proliferation) simply keeps the plain launches.  Results are identical: the
    // same kernels with the same arguments.  Measured on MI355X it buys nothing when the
    // host queues launches from a C++ loop (a step of a 10^4-cell system is bound by the
    // ~18 dependent kernels' own latencies, 0.19 ms either way); it is for hosts that
    // cannot keep ~10^5 launches per second up.
    int graph_steps = 0;

protected:
#ifndef YA_GRAPH_MAX_CELLS
#define YA_GRAPH_MAX_CELLS 400000
#endif
    struct Step_key {
        const void* functors = nullptr;
        int n = -1;
        float dt = 0, cube_size = 0;
        int variant = 0, fix_mode = 0, fix_point = 0;
        bool operator==(const Step_key& o) const
        {
            return functors == o.functors && n == o.n && dt == o.dt && cube_size == o.cube_size &&
                   variant == o.variant && fix_mode == o.fix_mode && fix_point == o.fix_point;
        }
    };
    Step_key graph_key, last_key;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t capture_stream = nullptr;
    void drop_graph()
    {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        graph_exec = nullptr;
        graph_key = Step_key{};
    }
    static constexpr int n_floats = ya::N_floats<Pt>::value;
    Pt *d_X, *d_dX, *d_X1, *d_dX1;
    float3* d_old_v;
    int* d_n;
    float *d_mean, *d_fix, *d_mean_first, *d_fix_first, *d_workspace;
    ya_n_reader* n_reader = nullptr;
    int sorted_stage_cells = -1;  // stage API: cells in the sorted copy stage 2 may start from
    bool mirrored_in_sorted_copy = false;  // ... and the mirrored cells' predictor is in it already
    // rows of d_dX / d_dX1 an update kernel left zeroed (stage_update_folding).  INVARIANT: every writer of those two
    // arrays goes through stage_rhs / sorted_step (which reset the promise) -- the arrays are protected members and
    // no accessor hands them out; a new path that writes them must reset rhs_zeroed as well.
    int rhs_zeroed[2] = {0, 0};
    std::function<void()> keep_order;  // keep_in_cube_order: renumber(the registered arrays)
    int keep_order_every = 0, keep_order_wait = 0;
    bool fix_com = true;
    bool fix_com_z = false;
    // sorted-space step with set_fixed(): the update kernels fold the reductions' partial sums themselves
    // (ya::fixed_velocity_from_partials; false = ya_reduce_mean's second launch, A/B)
    bool fold_in_update = true;
    int fix_point = 0;
    const int n_max;
    int get_d_n()
    {
        int n;
        YA_CHECK(ya_get_n(d_n, &n));
        assert(n <= n_max);
        return n;
    }
    void check_status() { Computer<Pt>::check_status(); }
    // renumber(): array[s] = array[order[s]] for s < n, through a scratch buffer that is kept
    void* renumber_scratch = nullptr;
    size_t renumber_scratch_bytes = 0;
    template<typename T>
    void permute_array(const int* order, const int n, T* d_array)
    {
        const size_t bytes = (size_t)n * sizeof(T);
        if (renumber_scratch_bytes < bytes) {
            if (renumber_scratch) ya_free(renumber_scratch);
            renumber_scratch_bytes = 0;
            YA_CHECK(ya_malloc(&renumber_scratch, (size_t)n_max * sizeof(T) > bytes ? (size_t)n_max * sizeof(T) : bytes));
            renumber_scratch_bytes = (size_t)n_max * sizeof(T) > bytes ? (size_t)n_max * sizeof(T) : bytes;
        }
        ya::permute_rows<T><<<(n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK, ya::UPDATE_BLOCK>>>(
            n, order, d_array, (T*)renumber_scratch);
        YA_CHECK(ya_memcpy_d2d_async(d_array, renumber_scratch, bytes, nullptr));
    }
    template<typename T>
    void renumber_one(const int* order, const int n, T*& d_array, int*&)
    {
        permute_array(order, n, d_array);
    }
    template<typename A>
    auto renumber_one(const int* order, const int n, A& property, int*&) -> decltype((void)property.d_prop)
    {
        permute_array(order, n, property.d_prop);
    }
    template<typename L>
    auto renumber_one(const int* order, const int n, L& links, int*& new_id) -> decltype((void)links.d_link)
    {
        if (!new_id) {
            YA_CHECK(ya_malloc((void**)&new_id, (size_t)n * sizeof(int)));
            ya::invert_order<<<(n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK, ya::UPDATE_BLOCK>>>(n, order, new_id);
        }
        const int n_ids = 2 * links.n_max;
        ya::rename_ids<<<(n_ids + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK, ya::UPDATE_BLOCK>>>(
            n_ids, links.d_n, n, new_id, reinterpret_cast<int*>(links.d_link));
    }

    // The velocity subtracted from dX.xyz for this stage, left in device memory.
    const float* fix_velocity(int n, Pt* d_rhs, bool mean, bool point_xy, bool first = false)
    {
        float* mean_out = first ? d_mean_first : d_mean;
        float* fix_out = first ? d_fix_first : d_fix;
        if (mean) {  // solvers.cuh:241-249 / :266-268
            YA_CHECK(ya_reduce_mean(d_rhs, n_floats, n, mean_out, d_workspace, this->stream));
            if (!point_xy) return mean_out;
            ya::make_fix<Pt><<<1, 1, 0, this->stream>>>(2, mean_out, d_rhs + fix_point, fix_out);
            return fix_out;
        }
        ya::make_fix<Pt><<<1, 1, 0, this->stream>>>(1, mean_out, d_rhs + fix_point, fix_out);  // :250-253
        return fix_out;
    }

    // The three pieces of a stage (stage 1 works on d_X -> d_dX, stage 2 on
    // d_X1 -> d_dX1).  take_step composes them; a z-slab decomposition calls
    // them one by one with a ghost exchange and an all-reduce in between
    // (n = own + ghost cells, n_active = own cells).
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void stage_rhs(int stage, int n, int n_active, Generic_forces<Pt>& gen_forces)
    {
        Pt* d_in = stage == 1 ? d_X : d_X1;
        Pt* d_rhs = stage == 1 ? d_dX : d_dX1;
        const bool has_gen = !ya::is_no_gen_forces<Pt>(gen_forces);
        if (has_gen) {
            // (the update kernel before this stage may have left the rows zeroed already: stage_update_folding)
            if (rhs_zeroed[stage - 1] < n) YA_CHECK(ya_memset_async(d_rhs, 0, (size_t)n * sizeof(Pt), nullptr));
            gen_forces(n, d_in, d_rhs);
        }
        rhs_zeroed[stage - 1] = 0;  // the force kernel writes it next
        if (stage == 2 && sorted_stage_cells == n) {
            // sorted-space second stage (see take_step): the own cells were moved by
            // stage_update(1), the ghost cells' new positions are in d_X1 (which the
            // generic forces above were given, as the reference does)
            sorted_stage_cells = -1;
            if (!mirrored_in_sorted_copy) Computer<Pt>::ghosts_in_sorted_space(n, n_active, d_X1);
            mirrored_in_sorted_copy = false;
            Computer<Pt>::template pwints_from_sorted<pw_int, pw_friction>(
                n, d_dX1, n_active, has_gen);
            return;
        }
        sorted_stage_cells = -1;
        mirrored_in_sorted_copy = false;
        const bool keep_sorted = stage == 1 && Computer<Pt>::use_sorted_pipeline();
        Computer<Pt>::template pwints<pw_int, pw_friction>(
            n, d_in, d_old_v, d_rhs, has_gen, n_active, keep_sorted);
        if (keep_sorted) sorted_stage_cells = n;
    }
    // sum and mean of the stage's right-hand side over the first n points, left on
    // the device as {mean[n_floats], sum[n_floats]}
    const float* stage_sum(int stage, int n)
    {
        YA_CHECK(ya_reduce_mean(
            stage == 1 ? d_dX : d_dX1, n_floats, n, d_mean, d_workspace, nullptr));
        return d_mean;
    }
    // n_sorted_active: the cells moved inside the cube-sorted copy by their sorted right-hand
    // sides (a z-slab moves its own cells there; mirrored cells enter the copy from d_X1)
    void stage_update(int stage, int n, float dt, const float* d_fix_velocity, int n_sorted_active = -1)
    {
        const int blocks = (n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK;
        if (stage == 1) {
            // before euler_step: it replaces d_dX by d_dX - fix, the sorted copy is raw
            if (sorted_stage_cells >= 0)
                Computer<Pt>::predictor_in_sorted_space(
                    sorted_stage_cells, dt, d_fix_velocity, n_sorted_active >= 0 ? n_sorted_active : n);
            euler_step<<<blocks, ya::UPDATE_BLOCK>>>(n, dt, d_X, d_fix_velocity, d_dX, d_X1);
        } else
            heun_step<<<blocks, ya::UPDATE_BLOCK>>>(
                n, dt, d_dX, d_fix_velocity, d_dX1, d_X, d_old_v);
    }

    // stage_update for set_fixed() (the default) with the fixed velocity folded by the update kernels themselves
    // from the stage's partial sums (two launches fewer per step, the same bits: see
    // ya::fixed_velocity_from_partials).  The sorted copy's predictor runs first and leaves the velocity for
    // euler_step; without a sorted copy (Tile_solver, Gabriel_solver) euler_step_folding does both.
    // With generic forces (`zeroing`) the same kernels leave the right-hand side array of the NEXT stage zeroed --
    // d_dX1 after the predictor, d_dX after the corrector: both dead by then -- so that the memset in front of the
    // generic forces (solvers.cuh:232,258 `thrust::fill`) need not be launched (rhs_zeroed: rows known to be zero).
    void stage_update_folding(int stage, int n, float dt, bool zeroing)
    {
        const int blocks = (n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK;
        int n_partials = 0;
        YA_CHECK(ya_reduce_partials(stage == 1 ? d_dX : d_dX1, n_floats, n, d_workspace, &n_partials, nullptr));
        if (stage == 1) {
            const bool with_sorted = sorted_stage_cells == n;
            if (!with_sorted) sorted_stage_cells = -1;
            euler_step_folding<<<blocks, ya::UPDATE_BLOCK>>>(n, dt, d_X, d_workspace, n_partials, d_mean_first, d_dX, d_X1,
                with_sorted ? Computer<Pt>::sorted_rhs() : nullptr, with_sorted ? Computer<Pt>::sorted_cells() : nullptr,
                zeroing ? d_dX1 : nullptr);
            rhs_zeroed[1] = zeroing ? n : 0;
        } else {
            heun_step_folding<<<blocks, ya::UPDATE_BLOCK>>>(n, dt, d_dX, d_workspace, n_partials, d_dX1, d_X, d_old_v, zeroing);
            rhs_zeroed[0] = zeroing ? n : 0;
        }
    }

    // The two updates of a z-slab's step without generic forces: stage 1 entirely inside the
    // sorted copy (own and mirrored cells, one launch; d_dX stays raw, d_X1 is not written), stage 2
    // with both fixed velocities subtracted in the corrector.  Returns false if stage 1 left no
    // sorted copy to work in (a solver without the sorted pipeline): the caller then uses
    // stage_update.
    // (d_total: the stage's all-reduced {sum, count pieces}; stage 1 leaves its fixed velocity in
    // d_fix_out, stage 2 reads it from there)
    // (fix_mode: ya::fix_from_total; pred_partial / z_selected + moved_partial: the drift guard's
    // maxima, ya::GUARD_SLOTS floats each (zeroed by the reduction that reads them), or NULL;
    // votes_out: see heun_step_raw_total)
    bool stage1_update_in_sorted_copy(int n, float dt, const float* d_total, float* d_fix_out, int n_active,
        int fix_mode = 0, float* pred_partial = nullptr, const ya::Guard_band band = ya::Guard_band{})
    {
        if (sorted_stage_cells != n) return false;
        Computer<Pt>::predictor_in_sorted_space_mirrored(n, dt, d_total, d_fix_out, n_active, d_dX, fix_mode, pred_partial, band);
        mirrored_in_sorted_copy = true;
        return true;
    }
    void stage2_update_raw(int n, float dt, const float* d_fix_stage1, const float* d_total_stage2, int fix_mode = 0,
        const float* z_selected = nullptr, float* moved_partial = nullptr, const ya::Guard_band band = ya::Guard_band{},
        float* votes_out = nullptr)
    {
        heun_step_raw_total<<<(n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK, ya::UPDATE_BLOCK>>>(
            n, dt, d_dX, d_fix_stage1, d_dX1, d_total_stage2, d_X, d_old_v, fix_mode, z_selected, moved_partial, band, votes_out);
    }
    // what a rank puts into a stage's all-reduce (ya_slab_pack): the sum over its first n points and
    // their count, its two votes (the drift guard brought up to date in the same kernel if
    // fold_guard), the fixed point's right-hand side if *d_fix_index is one of its cells
    struct Guard_inputs {
        float* moved_partial = nullptr;
        int n_moved = 0;
        float* pred_partial = nullptr;
        int n_pred = 0;
        float limit = 0.f, lag_steps = 0.f;
        float* state = nullptr;
    };
    void stage_sum_packed(int stage, int n, float* d_out, const Guard_inputs& guard = Guard_inputs{}, int fold_guard = 0,
        int with_votes = 0, int host_error = 0, const int* d_fix_index = nullptr)
    {
        YA_CHECK(ya_slab_pack(stage == 1 ? d_dX : d_dX1, n_floats, n, d_out, d_workspace, guard.moved_partial, guard.n_moved,
            guard.pred_partial, guard.n_pred, guard.limit, guard.lag_steps, guard.state, fold_guard, with_votes, host_error,
            d_fix_index, nullptr));
    }

    // Sorted-space pipeline (Grid_solver without generic forces): the predictor lives in
    // the cube-sorted copy of the cells, so the second grid build gathers nothing and
    // d_X1 is never materialised.  Same arithmetic, same results.
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void sorted_step(const int n, const float dt)
    {
        const int blocks = (n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK;
        rhs_zeroed[0] = rhs_zeroed[1] = 0;  // both right-hand side arrays are written here
        Computer<Pt>::template pwints<pw_int, pw_friction>(n, d_X, d_old_v, d_dX, false, n, true);
        if (fix_com and !fix_com_z and fold_in_update) {
            // set_fixed() (the default): both fixed velocities are means, folded by the update kernels
            // themselves from the reductions' partial sums (two launches fewer; same bits)
            int n_partials = 0;
            YA_CHECK(ya_reduce_partials(d_dX, n_floats, n, d_workspace, &n_partials, this->stream));
            Computer<Pt>::predictor_in_sorted_space_folding(n, dt, d_workspace, n_partials, d_mean_first);
            Computer<Pt>::template pwints_from_sorted<pw_int, pw_friction>(n, d_dX1, n, false);
            YA_CHECK(ya_reduce_partials(d_dX1, n_floats, n, d_workspace, &n_partials, this->stream));
            heun_step_raw_folding<<<blocks, ya::UPDATE_BLOCK, 0, this->stream>>>(
                n, dt, d_dX, d_mean_first, d_dX1, d_workspace, n_partials, d_X, d_old_v);
            return;
        }
        // this stage's fixed velocity has to outlive the next reduction
        const float* fix = fix_velocity(n, d_dX, fix_com or fix_com_z, fix_com_z, true);
        Computer<Pt>::predictor_in_sorted_space(n, dt, fix, n);
        Computer<Pt>::template pwints_from_sorted<pw_int, pw_friction>(n, d_dX1, n, false);
        const float* fix1 = fix_velocity(n, d_dX1, fix_com, false);
        heun_step_raw<<<blocks, ya::UPDATE_BLOCK, 0, this->stream>>>(
            n, dt, d_dX, fix, d_dX1, fix1, d_X, d_old_v);
    }

    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    Step_key key_of(const int n, const float dt)
    {
        static const char functors_tag = 0;  // one per <pw_int, pw_friction>
        Step_key k;
        k.functors = &functors_tag;
        k.n = n;
        k.dt = dt;
        k.cube_size = Computer<Pt>::step_cube_size();
        k.variant = Computer<Pt>::step_variant();
        k.fix_mode = (fix_com ? 1 : 0) | (fix_com_z ? 2 : 0);
        k.fix_point = fix_point;
        return k;
    }

    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void take_step(float dt, Generic_forces<Pt> gen_forces)
    {
        if (keep_order && keep_order_wait-- <= 0) {  // keep_in_cube_order
            keep_order();
            keep_order_wait = keep_order_every - 1;
        }
        const bool sorted_path =
            Computer<Pt>::use_sorted_pipeline() && ya::is_no_gen_forces<Pt>(gen_forces);
        int n;
        if (sorted_path && graph_steps != 0 && last_key.n > 0 && !Computer<Pt>::profiler.active() &&
            (graph_steps > 0 || last_key.n < YA_GRAPH_MAX_CELLS)) {
            // Small system: read n first (solvers.cuh:229), then replay / capture / launch.
            n = get_d_n();
            if (n <= 0) return;
            const Step_key key = key_of<pw_int, pw_friction>(n, dt);
            if (graph_exec && key == graph_key) {
                YA_CHECK((int)hipGraphLaunch(graph_exec, nullptr));
                return;
            }
            if (key == last_key && (graph_steps > 0 || n < YA_GRAPH_MAX_CELLS)) {
                // the same step as last time: capture it (thread-local mode: other host
                // threads of the model, e.g. one writing output, may keep using HIP)
                drop_graph();
                if (!capture_stream)
                    YA_CHECK((int)hipStreamCreateWithFlags(&capture_stream, hipStreamNonBlocking));
                YA_CHECK((int)hipStreamBeginCapture(capture_stream, hipStreamCaptureModeThreadLocal));
                this->stream = capture_stream;
                sorted_step<pw_int, pw_friction>(n, dt);
                this->stream = nullptr;
                hipGraph_t graph = nullptr;
                YA_CHECK((int)hipStreamEndCapture(capture_stream, &graph));
                YA_CHECK((int)hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
                YA_CHECK((int)hipGraphDestroy(graph));
                graph_key = key;
                YA_CHECK((int)hipGraphLaunch(graph_exec, nullptr));
                return;
            }
            last_key = key;
            sorted_step<pw_int, pw_friction>(n, dt);
            return;
        }
        if (Computer<Pt>::use_sorted_pipeline()) {
            // The reference reads n first (solvers.cuh:229) and so must we, model kernels
            // change it between steps; but the round trip is hidden behind the binning
            // kernels of the first grid build, which read the count on the device.
            // (Round 5: with generic forces too.  They need n on the host -- gen_forces(n, d_X, d_dX) -- but
            // the first build reads d_X only and the generic forces write d_dX only, so the build's first
            // three kernels may run before them: the device works while the count travels, instead of
            // standing idle between two steps.)
            // (Round 6: the count travels in the first binning kernel itself, ya_grid_build_sorted_begin_publish,
            // not in a 4-byte copy queued ahead of it: one 4 us kernel fewer in the stream per step.)
            Computer<Pt>::begin_build(d_X, d_n, n_max, n_reader);
            YA_CHECK(ya_n_read_end(n_reader, &n));
            assert(n <= n_max);
        } else {
            n = get_d_n();
        }
        if (n <= 0) {
            Computer<Pt>::cancel_build();
            return;
        }

        if (sorted_path) {
            last_key = key_of<pw_int, pw_friction>(n, dt);
            sorted_step<pw_int, pw_friction>(n, dt);
            return;
        }
        last_key = Step_key{};

        const bool folding = fix_com and !fix_com_z and fold_in_update;
        // 1st stage
        stage_rhs<pw_int, pw_friction>(1, n, n, gen_forces);
        const bool zeroing = !ya::is_no_gen_forces<Pt>(gen_forces);
        if (folding)
            stage_update_folding(1, n, dt, zeroing);
        else
            stage_update(1, n, dt, fix_velocity(n, d_dX, fix_com or fix_com_z, fix_com_z));

        // 2nd stage
        stage_rhs<pw_int, pw_friction>(2, n, n, gen_forces);
        if (folding)
            stage_update_folding(2, n, dt, zeroing);
        else
            stage_update(2, n, dt, fix_velocity(n, d_dX1, fix_com, false));
    }
};


// All-pairs interactions, LDS-tiled (solvers.cuh:279-342).  TILE_SIZE is kept
// for source compatibility; the kernel's own tile is ya::TILE_POINTS.
const auto TILE_SIZE = 32;

template<typename Pt>
class Tile_computer {
public:
    Tile_computer(int n_max) {}
    ya::Profiler profiler;
    bool use_sorted_pipeline() const { return false; }
    // 0 (default) = one thread owns a cell for the whole stage, as in the reference, unless the
    // functors are declared stateless (YA_STATELESS): then 64 lanes per cell up to 4096 cells and
    // 16 above; 1 = always one thread per cell; 16 or 64 = ya::tile_force_coop with that many
    // lanes per cell whatever the functor says (bit-identical sums; force launch at 800 cells
    // 196 -> 32 us with 64 lanes, at 3000 cells 631 -> 135 us with 16).
    int lanes_per_cell = 0;

protected:
    hipStream_t stream = nullptr;  // every launch of a step goes here (null = the default stream)
    float step_cube_size() const { return 0; }
    int step_variant() const { return 0; }
    void check_status() {}
    void begin_build(const Pt*, const int*, int, ya_n_reader*) {}
    void cancel_build() {}
    const int* cube_order(int, const Pt*) { return nullptr; }  // no grid: renumber() is a no-op
    void ids_changed() {}
    void predictor_in_sorted_space(int, float, const float*, int) {}
    void predictor_in_sorted_space_folding(int, float, const float*, int, float*) {}
    const Pt* sorted_rhs() const { return nullptr; }
    ya::Entry<Pt>* sorted_cells() { return nullptr; }
    void predictor_in_sorted_space_mirrored(int, float, const float*, float*, int, const Pt*, int, float*, ya::Guard_band) {}
    void ghosts_in_sorted_space(int, int, const Pt*) {}
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void pwints_from_sorted(int, Pt*, int, bool) {}
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void pwints(const int n, const Pt* __restrict__ d_X, const float3* __restrict__ d_old_v,
        Pt* d_dX, const bool has_gen, const int n_active, const bool keep_sorted)
    {
        assert(n_active == n);  // Tile_solver is single-GPU only (all pairs)
        hipEvent_t start, stop;
        profiler.next(&start, &stop);
        int lanes = lanes_per_cell;
        if (lanes == 0) lanes = ya::stateless_pair<Pt, pw_int, pw_friction>() ? (n <= 4096 ? 64 : 16) : 1;
        if (lanes >= 64)
            hipExtLaunchKernelGGL((ya::tile_force_coop<Pt, pw_int, pw_friction, 64>), dim3((n + 3) / 4),
                dim3(256), 0, stream, start, stop, 0, n, d_X, d_old_v, d_dX, has_gen);
        else if (lanes > 1)
            hipExtLaunchKernelGGL((ya::tile_force_coop<Pt, pw_int, pw_friction, 16>), dim3((n + 15) / 16),
                dim3(256), 0, stream, start, stop, 0, n, d_X, d_old_v, d_dX, has_gen);
        else
            hipExtLaunchKernelGGL((ya::tile_force<Pt, pw_int, pw_friction>),
                dim3((n + ya::TILE_BLOCK - 1) / ya::TILE_BLOCK), dim3(ya::TILE_BLOCK), 0, stream, start,
                stop, 0, n, d_X, d_old_v, d_dX, has_gen);
    }
};

template<typename Pt>
using Tile_solver = Heun_solver<Pt, Tile_computer>;


// Uniform grid over space: cells sorted by cube id, ONLY points closer than
// cube_size interact (solvers.cuh:345-425).  The four arrays and the device
// mirror d_grid are public API used from model kernels, e.g.
// `d_grid->d_cube_start[cube]`.
class Grid {
public:
    int *d_cube_id, *d_point_id, *d_cube_start, *d_cube_end;
    Grid* d_grid;
    const int n_max, grid_size, n_cubes;
    Grid(int n_max, int gs = 50) : n_max{n_max}, grid_size{gs}, n_cubes{gs * gs * gs}
    {
        YA_CHECK(ya_grid_create(n_max, gs, &handle));
        YA_CHECK(ya_grid_arrays(handle, &d_cube_id, &d_point_id, &d_cube_start, &d_cube_end));
        YA_CHECK(ya_grid_offsets(handle, &d_offs));
        YA_CHECK(ya_malloc((void**)&d_grid, sizeof(Grid)));
        YA_CHECK(ya_memcpy_h2d(d_grid, this, sizeof(Grid)));
    }
    ~Grid()
    {
        ya_free(d_grid);
        ya_grid_destroy(handle);
    }
    Grid(const Grid&) = delete;
    template<typename Pt>
    void build(const int n, const Pt* __restrict__ d_X, const float cube_size = 1)
    {
        YA_CHECK(ya_grid_build(handle, d_X, sizeof(Pt), n, cube_size, nullptr));
    }
    template<typename Pt, template<typename> class Solver>
    void build(Solution<Pt, Solver>& points, const float cube_size = 1)
    {
        auto n = points.get_d_n();
        assert(n <= n_max);
        build(n, points.d_X, cube_size);
    }
    // Engine-side extras (not in the reference).
    template<typename Pt>
    void build_sorted(const int n, const Pt* d_X, const float3* d_old_v, const float cube_size,
        ya::Entry<Pt>* d_sorted, float4* d_sorted_v, hipStream_t stream = nullptr)
    {
        YA_CHECK(ya_grid_build_sorted(handle, d_X, sizeof(Pt), d_old_v, n, cube_size, d_sorted,
            sizeof(ya::Entry<Pt>), d_sorted_v, stream));
    }
    // build_sorted in two halves: the first one reads the count on the device and can be
    // queued before the host has it (ya_grid_build_sorted_begin / _finish).
    template<typename Pt>
    void build_sorted_begin(const Pt* d_X, const int* d_n, const int n_bound, const float cube_size, ya_n_reader* reader)
    {
        YA_CHECK(ya_grid_build_sorted_begin_publish(handle, d_X, sizeof(Pt), d_n, n_bound, cube_size, reader, nullptr));
    }
    template<typename Pt>
    void build_sorted_finish(const int n, const Pt* d_X, const float3* d_old_v,
        ya::Entry<Pt>* d_sorted, float4* d_sorted_v)
    {
        YA_CHECK(ya_grid_build_sorted_finish(handle, d_X, sizeof(Pt), d_old_v, n, d_sorted,
            sizeof(ya::Entry<Pt>), d_sorted_v, nullptr));
    }
    // Same result as build_sorted on the cells held (in any order) by d_prev, read
    // from those entries themselves: nothing is gathered through point ids.
    template<typename Pt>
    void rebuild_sorted(const int n, const ya::Entry<Pt>* d_prev, const float4* d_prev_v,
        const float cube_size, ya::Entry<Pt>* d_sorted, float4* d_sorted_v,
        hipStream_t stream = nullptr)
    {
        YA_CHECK(ya_grid_rebuild_sorted(handle, d_prev, sizeof(ya::Entry<Pt>), sizeof(Pt),
            d_prev_v, n, cube_size, d_sorted, d_sorted_v, stream));
    }
    const int* offsets() const { return d_offs; }
    // The ids the last build saw mean other cells now (Solution::renumber): the next build visits
    // the cells in storage order instead of the last build's.
    void forget_order() { YA_CHECK(ya_grid_forget_order(handle)); }
    // A promise that every cell's cube id lies in [cube_lo, cube_hi): builds then scan those cubes
    // only (ya_grid_set_cube_range; a z-slab holds cells in a fraction of the grid's planes).
    void set_cube_range(const int cube_lo, const int cube_hi) { YA_CHECK(ya_grid_set_cube_range(handle, cube_lo, cube_hi)); }
    void check_status()
    {
        int bits = 0;
        YA_CHECK(ya_grid_status(handle, &bits, 1));
        if (bits & YA_STATUS_OUT_OF_RANGE) {
            fprintf(stderr,
                "yalla-hip: a cell left the cube range its z-slab promised the grid (Grid::set_cube_range): it "
                "moved more than a cube between two selections of the mirrored cells.\n");
            abort();
        }
        if (bits & YA_STATUS_SCAN_STALLED) {
            fprintf(stderr,
                "yalla-hip: the grid build's prefix sum stalled (a workgroup waited for one of lower index that had "
                "not been started: the device does not dispatch workgroups in index order, or two builds of one "
                "Grid ran at once on different streams).  The grid arrays of that build are not valid.\n");
            abort();
        }
        if (bits & YA_STATUS_OUT_OF_GRID) {
            fprintf(stderr,
                "yalla-hip: a cell left the %d^3 grid (device assertion at "
                "ya||a solvers.cuh:361-362); enlarge grid_size or cube_size.\n",
                grid_size);
            abort();
        }
    }

private:
    ya_grid* handle = nullptr;
    const int* d_offs = nullptr;
};


__constant__ int d_nhood[27];  // stencil offsets for model kernels (solvers.cuh:428)

enum Ya_sum_order { YA_SUM_REFERENCE = 0, YA_SUM_BY_PLANE = 1 };  // Grid_computer::sum_order

template<typename Pt>
class Grid_computer {
public:
    float cube_size;
    ya::Profiler profiler;
    // 2 = grid_force_bits (bit stream) always; 1 = grid_force (byte FIFO), 0 = grid_force_direct: the A/B
    // baselines of tools/ab/force_variants.cuh, only with -DYA_EXPERIMENTAL_FORCE_VARIANTS;
    // 3 = grid_force_coop below ~7 * 10^4 cells (16, 8 or 4 lanes per cell, more the smaller the
    // system), grid_force_bits above: for models whose functors keep no per-cell state without
    // atomics (bit-identical results; see the kernel's comment)
    // -1 (default) = grid_force_bits, or -- for functors declared stateless (YA_STATELESS) --
    // grid_force_coop below ~7 * 10^4 cells
    int force_variant = -1;
    // The order in which a cell's pair terms are added (both restated in oracle/yalla_host.hpp under the same names):
    //   YA_SUM_REFERENCE (default)  one running sum over the 27 cubes in d_nhood order -- the reference's
    //                               thread (solvers.cuh:437-459), bit for bit for + - * / sqrt fma functors;
    //   YA_SUM_BY_PLANE  (opt-in)   S[own z-plane] + S[planes below and above], each in the reference's order:
    //                               lets grid_force_bits give a tile's planes to two wavefronts ("the tail";
    //                               launches of 7 * 10^4 .. 1.6 * 10^5 cells as halves altogether).  ~1e-7
    //                               relative per step beside the reference's sum, inside north_star's 1e-5.
    Ya_sum_order sum_order = YA_SUM_REFERENCE;
    int coop_lanes = 0;            // force_variant 3: 0 = from n (ya::coop::lanes_for), or 4 / 8 / 16
    int stage_v_max = 130000;      // grid_force_bits keeps old_v in LDS too up to this many cells
    Grid_computer(int n_max, int grid_size = 50, float cube_size = 1)
        : cube_size{cube_size}, grid{n_max, grid_size}
    {
        // Row-major 3x3x3 offsets in the order x, then {0,-gs,+gs}, then
        // {0,-gs^2,+gs^2} (solvers.cuh:472-483).
        int h_nhood[27];
        const int shift[3] = {0, -1, 1};
        for (int z = 0; z < 3; z++)
            for (int y = 0; y < 3; y++)
                for (int x = 0; x < 3; x++)
                    h_nhood[9 * z + 3 * y + x] =
                        (x - 1) + shift[y] * grid_size + shift[z] * grid_size * grid_size;
        YA_CHECK((int)hipMemcpyToSymbol(HIP_SYMBOL(d_nhood), h_nhood, sizeof(h_nhood)));
        YA_CHECK(ya_malloc((void**)&d_sorted, (size_t)n_max * sizeof(ya::Entry<Pt>)));
        YA_CHECK(ya_malloc((void**)&d_sorted_v, (size_t)n_max * sizeof(float4)));
        YA_CHECK(ya_malloc((void**)&d_resorted, (size_t)n_max * sizeof(ya::Entry<Pt>)));
        YA_CHECK(ya_malloc((void**)&d_resorted_v, (size_t)n_max * sizeof(float4)));
        YA_CHECK(ya_malloc((void**)&d_dX_sorted, (size_t)n_max * sizeof(Pt)));
    }
    ~Grid_computer()
    {
        ya_free(d_sorted);
        ya_free(d_sorted_v);
        ya_free(d_resorted);
        ya_free(d_resorted_v);
        ya_free(d_dX_sorted);
        for (int p = 0; p < 3; p++) ya_free(d_tail_exchange[p]), ya_free(d_tail_tickets[p]);
        if (interior_stream && interior_stream_owned) (void)hipStreamDestroy(interior_stream);
        if (grid_built) {
            (void)hipEventDestroy(grid_built);
            (void)hipEventDestroy(interior_done);
        }
    }
    Grid_computer(const Grid_computer&) = delete;
    // grid_force_bits' tail (the last tiles of a launch as half tiles): -1 = chosen from the launch's size,
    // 0 = none, or the number of tiles (A/B knob).  Results do not depend on it.
    int force_tail_tiles = -1;  // (a number of the launch's tiles or more: every tile as halves)
    float* d_tail_exchange[3] = {nullptr, nullptr, nullptr};  // per launch kind (part 0 / 1 / 2)
    int* d_tail_tickets[3] = {nullptr, nullptr, nullptr};
    int tail_room[3] = {0, 0, 0};
    // one-wavefront workgroups of `Kernel` the CURRENT device holds at once (occupancy x CUs), per device
    template<auto Kernel>
    static int resident_workgroups()
    {
        static std::mutex lock;
        static std::vector<int> by_device;
        int device = 0;
        YA_CHECK((int)hipGetDevice(&device));
        std::lock_guard<std::mutex> hold(lock);
        if ((int)by_device.size() <= device) by_device.resize(device + 1, 0);
        if (by_device[device] == 0) {
            int per_cu = 0, cus = 0;
            YA_CHECK((int)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, Kernel, ya::bits::BLOCK, 0));
            YA_CHECK((int)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
            by_device[device] = per_cu * cus;
        }
        return by_device[device];
    }
    bool sorted_pipeline = true;  // false = both stages through d_X / d_X1 (A/B)
    // z-slab decomposition (include/slab_logic.inc): 1 = the next forces() call launches the tiles
    // in the z-planes next to the slab's faces only (cube ids below force_part_cube_lo or from
    // force_part_cube_hi up) and remembers its arguments; forces_interior() then launches the
    // rest on a stream of its own, and join_interior() makes the step's stream wait for it.  The
    // exchange of the boundary cells' right-hand sides runs beside the interior launch.  Kernels
    // other than grid_force_bits compute everything in the first call.
    int force_part = 0;
    int force_part_cube_lo = 0, force_part_cube_hi = 0x7fffffff;
    // ... and the cubes that can hold own cells at all: [force_own_cube_lo, force_own_cube_hi); tiles of
    // cubes outside hold mirrored cells only and are left out of both launches
    int force_own_cube_lo = 0, force_own_cube_hi = 0x7fffffff;
    // z-slab decomposition: local cell index -> global id (own cells, then ghosts); pairwise
    // functors are then called with global (i, j).  NULL (default): local = global.
    const int* d_global_id = nullptr;
    bool use_sorted_pipeline() const { return sorted_pipeline and force_variant != 0; }

protected:
    hipStream_t stream = nullptr;  // every launch of a step goes here (null = the default stream)
    float step_cube_size() const { return cube_size; }
    int step_variant() const { return force_variant; }
    Grid grid;
    ya::Entry<Pt>*d_sorted, *d_resorted;
    float4 *d_sorted_v, *d_resorted_v;
    Pt* d_dX_sorted;
    void check_status() { grid.check_status(); }
    struct Forces_call {
        int n = 0, n_active = 0;
        const ya::Entry<Pt>* d_cells = nullptr;
        const float4* d_cells_v = nullptr;
        Pt *d_dX = nullptr, *d_dX_in_cell_order = nullptr;
        bool has_gen = false, split = false;
    } boundary_call;
    hipStream_t interior_stream = nullptr;
    bool interior_stream_owned = false;
    hipEvent_t grid_built = nullptr, interior_done = nullptr;
    // The stream of a stage's second force launch, supplied by the program (several solvers of
    // one process may share one: every stream beyond the device's few hardware queues shares a
    // queue with another, and two launches on one queue run one after the other).  Call before
    // the first decomposed step; the stream stays the caller's.
    void use_interior_stream(hipStream_t s)
    {
        if (interior_stream && interior_stream_owned) (void)hipStreamDestroy(interior_stream);
        interior_stream = s;
        interior_stream_owned = false;
    }
    // the second launch of a stage whose first one ran with force_part = 1
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void forces_interior()
    {
        if (!boundary_call.split) return;  // the first call computed every tile
        const Forces_call c = boundary_call;
        const int part = force_part;
        force_part = 2;
        forces<pw_int, pw_friction>(c.n, c.d_cells, c.d_cells_v, c.d_dX, c.has_gen, c.n_active, c.d_dX_in_cell_order);
        force_part = part;
        YA_CHECK((int)hipEventRecord(interior_done, interior_stream));
    }
    void join_interior()
    {
        if (!boundary_call.split) return;
        YA_CHECK((int)hipStreamWaitEvent(stream, interior_done, 0));
        boundary_call.split = false;
    }
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void forces(const int n, const ya::Entry<Pt>* d_cells, const float4* d_cells_v, Pt* d_dX,
        const bool has_gen, const int n_active, Pt* d_dX_in_cell_order)
    {
        const int blocks = (n + ya::FORCE_BLOCK - 1) / ya::FORCE_BLOCK;
        hipEvent_t start, stop;
        profiler.next(&start, &stop);
        const float cut2 = ya::cutoff_squared(cube_size);
        const bool by_plane = sum_order == YA_SUM_BY_PLANE;
        // a timed launch carries its events in the dispatch itself (ya::Profiler); an untimed
        // one is a plain launch, which is also what a stream capture records
#define YA_FORCE_LAUNCH(kernel_, grid_, block_, ...)                                          \
    if (start)                                                                                \
        hipExtLaunchKernelGGL((kernel_), dim3(grid_), dim3(block_), 0, stream, start, stop, 0, \
            __VA_ARGS__);                                                                     \
    else                                                                                      \
        hipLaunchKernelGGL((kernel_), dim3(grid_), dim3(block_), 0, stream, __VA_ARGS__)
        // the kernel this launch goes to: an explicit choice, or by what the model said about its functors
        const int force_variant = this->force_variant >= 0 ? this->force_variant
                                  : (ya::stateless_pair<Pt, pw_int, pw_friction>() ? 3 : 2);
        const int lanes = force_variant == 3 ? (coop_lanes ? coop_lanes : ya::coop::lanes_for(n, by_plane)) : 1;
        if (force_variant < 0 || force_variant > 3) {
            fprintf(stderr, "yalla-hip: Grid_computer::force_variant %d is not a kernel (-1, 0, 1, 2, 3)\n", force_variant);
            abort();
        }
        // two launches per stage (force_part) are what grid_force_bits offers; the other kernels
        // compute every tile in the first call
        const bool bits_kernel = lanes == 1 && force_variant >= 2;
        int part = 0;
        hipStream_t stream = this->stream;
        if (force_part == 1) {
            boundary_call = Forces_call{n, n_active, d_cells, d_cells_v, d_dX, d_dX_in_cell_order, has_gen, bits_kernel};
            if (bits_kernel) {
                part = 1;
                if (!interior_stream) {
                    // YA_INTERIOR_LOW_PRIORITY=1: the boundary launch, whose rows the neighbours
                    // wait for, gets the chip first.  Measured in the one-GPU rehearsal (10 M cells,
                    // 8 slabs): the boundary rows are ready after 173-210 instead of 255 us, but the
                    // interior launch then starts 110 us late and the stage takes 333 instead of
                    // 312 us -- off by default until a real neighbour has been seen waiting.
                    int least = 0, greatest = 0;
                    YA_CHECK((int)hipDeviceGetStreamPriorityRange(&least, &greatest));
                    YA_CHECK((int)hipStreamCreateWithPriority(&interior_stream, hipStreamNonBlocking,
                        getenv("YA_INTERIOR_LOW_PRIORITY") ? least : 0));
                    interior_stream_owned = true;
                }
                if (!grid_built) {
                    YA_CHECK((int)hipEventCreateWithFlags(&grid_built, hipEventDisableTiming));
                    YA_CHECK((int)hipEventCreateWithFlags(&interior_done, hipEventDisableTiming));
                }
                // the interior launch needs the grid, not the boundary launch
                YA_CHECK((int)hipEventRecord(grid_built, stream));
            }
        } else if (force_part == 2) {
            part = 2;
            stream = interior_stream;
            YA_CHECK((int)hipStreamWaitEvent(interior_stream, grid_built, 0));
        }
#define YA_COOP_LAUNCH(lanes_)                                                                 \
    YA_FORCE_LAUNCH((ya::grid_force_coop<Pt, pw_int, pw_friction, lanes_>),                    \
        (n + ya::coop::BLOCK / lanes_ - 1) / (ya::coop::BLOCK / lanes_), ya::coop::BLOCK, n,   \
        d_cells, d_cells_v, (const int*)grid.d_cube_id, grid.offsets(), grid.grid_size,        \
        grid.n_cubes, cut2, d_dX, has_gen, n_active, d_dX_in_cell_order, (const int*)d_global_id, by_plane)
        if (lanes == 16) {
            YA_COOP_LAUNCH(16);
        } else if (lanes == 8) {
            YA_COOP_LAUNCH(8);
        } else if (lanes == 4) {
            YA_COOP_LAUNCH(4);
        } else if (force_variant == 2 || force_variant == 3) {
#define YA_BITS_LAUNCH(stage_v_, gids_)                                                        \
    YA_FORCE_LAUNCH((ya::grid_force_bits<Pt, pw_int, pw_friction, stage_v_, gids_>),           \
        tail < 0 ? 16 * ((tiles + 7) / 8) : tiles + (tail > 0 ? std::min(tail, tiles) + 24 : 0), ya::bits::BLOCK, n, d_cells, \
        d_cells_v, (const int*)grid.d_cube_id, grid.offsets(), grid.grid_size, grid.n_cubes, cut2, d_dX,  \
        has_gen, n_active, d_dX_in_cell_order, (const int*)d_global_id, tiles, tail,           \
        d_tail_exchange[part], d_tail_tickets[part], by_plane, part, force_part_cube_lo,                 \
        force_part_cube_hi, force_own_cube_lo, force_own_cube_hi)
            const int tiles = (n + ya::bits::BLOCK - 1) / ya::bits::BLOCK;
            // Half tiles (grid_force_bits, "the tail"; > 0: the last so many tiles of the launch, < 0: all).
            // With R one-wavefront workgroups resident at once (5120 for springs): launches of up to R / 2 tiles
            // are made of halves altogether; up to R tiles, as many tiles are split as fill the chip in its one
            // round (R - tiles); beyond, the launch ends with 768 tiles as halves (measured at 2 * 10^5 ...
            // 10^7 cells, profiles/r05_tail_ab.txt).  Two wavefronts then call the functor for the same cell i
            // at once: only for functors declared stateless (YA_STATELESS; relu_w_epithelium's
            // `d_mes_nbs[i] += 1` would lose counts).
            int tail = 0;
            if (by_plane && ya::stateless_pair<Pt, pw_int, pw_friction>()) {
                if (force_tail_tiles >= 0) {
                    tail = force_tail_tiles >= tiles ? -1 : force_tail_tiles;
                } else {
                    const int resident = resident_workgroups<ya::grid_force_bits<Pt, pw_int, pw_friction, false, false>>();
                    if (part != 0)  // a slab stage's two launches, side by side: all halves only if both fit at once;
                        tail = 2 * tiles <= resident ? -1 : 768;  // the kernel leaves a list of < 4 tails' tiles whole
                    else
                        tail = 2 * tiles <= resident ? -1 : (tiles <= resident ? resident - tiles : 768);
                }
            }
            const int room = tail < 0 ? tiles : tail;
            if (room > 0 && (!d_tail_exchange[part] || tail_room[part] < room)) {
                // (a solver's launches are stream-ordered except parts 1 and 2 of a slab stage, which have
                // exchange areas of their own)
                constexpr int NC = ya::N_floats<Pt>::value + 4;
                YA_CHECK(ya_device_synchronize());
                if (d_tail_exchange[part]) ya_free(d_tail_exchange[part]), ya_free(d_tail_tickets[part]);
                // once per solver, not once per size: a system that grows (proliferation) asks for a tile more
                // every few steps while its launches are made of halves, which they are up to `resident` tiles
                const int most = std::min((grid.n_max + ya::bits::BLOCK - 1) / ya::bits::BLOCK,
                    std::max(resident_workgroups<ya::grid_force_bits<Pt, pw_int, pw_friction, false, false>>(), 768));
                const size_t slots = (size_t)std::max(room, most) + 8;
                YA_CHECK(ya_malloc((void**)&d_tail_exchange[part], slots * 2 * NC * ya::bits::BLOCK * sizeof(float)));
                YA_CHECK(ya_malloc((void**)&d_tail_tickets[part], slots * sizeof(int)));
                YA_CHECK(ya_memset_async(d_tail_tickets[part], 0, slots * sizeof(int), nullptr));
                YA_CHECK(ya_device_synchronize());
                tail_room[part] = (int)slots - 8;
            }
            // (old_v in LDS costs a launch of halves the residency it lives on: 13 KB per workgroup are 12 per CU)
            const bool stage_v = n <= stage_v_max && tail >= 0;
            if (d_global_id) {
                if (stage_v) {
                    YA_BITS_LAUNCH(true, true);
                } else {
                    YA_BITS_LAUNCH(false, true);
                }
            } else if (stage_v) {
                YA_BITS_LAUNCH(true, false);
            } else {
                YA_BITS_LAUNCH(false, false);
            }
#undef YA_BITS_LAUNCH
        }
#ifdef YA_EXPERIMENTAL_FORCE_VARIANTS
        else if (force_variant == 0) {
            YA_FORCE_LAUNCH((ya::grid_force_direct<Pt, pw_int, pw_friction>), blocks, ya::FORCE_BLOCK, n,
                d_cells, d_cells_v, (const int*)grid.d_cube_id, grid.offsets(), grid.grid_size,
                grid.n_cubes, cube_size, d_dX, has_gen, n_active, (const int*)d_global_id, by_plane);
        } else {
            YA_FORCE_LAUNCH((ya::grid_force<Pt, pw_int, pw_friction>), blocks, ya::FORCE_BLOCK, n, d_cells,
                d_cells_v, (const int*)grid.d_cube_id, grid.offsets(), grid.grid_size, grid.n_cubes,
                cut2, d_dX, has_gen, n_active, d_dX_in_cell_order, (const int*)d_global_id, by_plane);
        }
#else
        else {
            fprintf(stderr, "yalla-hip: Grid_computer::force_variant %d is an A/B baseline kept in "
                            "tools/ab/force_variants.cuh: compile with -DYA_EXPERIMENTAL_FORCE_VARIANTS -Itools/ab\n", force_variant);
            abort();
        }
#endif
#undef YA_COOP_LAUNCH
#undef YA_FORCE_LAUNCH
    }
    // Heun_solver::renumber: the cells' ids in (cube, id) order = the point ids of a fresh build
    const int* cube_order(const int n, const Pt* d_X)
    {
        grid.build(n, d_X, cube_size);
        return grid.d_point_id;
    }
    // ... after which cell s IS slot s: the next build visits the cells in storage order
    void ids_changed() { grid.forget_order(); }
    // The part of the first stage's grid build that can be queued before the host knows
    // n (Heun_solver::take_step); pwints then only finishes the build.
    void begin_build(const Pt* d_X, const int* d_n, const int n_bound, ya_n_reader* reader)
    {
        grid.build_sorted_begin(d_X, d_n, n_bound, cube_size, reader);
        build_begun = true;
    }
    void cancel_build() { build_begun = false; }
    bool build_begun = false;
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void pwints(const int n, const Pt* __restrict__ d_X, const float3* __restrict__ d_old_v,
        Pt* d_dX, const bool has_gen, const int n_active, const bool keep_sorted)
    {
        if (build_begun)
            grid.build_sorted_finish(n, d_X, d_old_v, d_sorted, d_sorted_v);
        else
            grid.build_sorted(n, d_X, d_old_v, cube_size, d_sorted, d_sorted_v, stream);
        build_begun = false;
        forces<pw_int, pw_friction>(n, d_sorted, d_sorted_v, d_dX, has_gen, n_active,
            keep_sorted ? d_dX_sorted : nullptr);
    }
    // The two halves of the second Heun stage when the first one kept its
    // right-hand side in cell order (Heun_solver::take_step).
    void predictor_in_sorted_space(
        const int n, const float dt, const float* d_fix, const int n_active)
    {
        euler_step_sorted<<<(n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK, ya::UPDATE_BLOCK, 0, stream>>>(
            n, dt, d_fix, d_dX_sorted, d_sorted, n_active);
    }
    const Pt* sorted_rhs() const { return d_dX_sorted; }
    ya::Entry<Pt>* sorted_cells() { return d_sorted; }
    void predictor_in_sorted_space_folding(const int n, const float dt, const float* d_partials, const int n_partials,
        float* d_fix_out)
    {
        euler_step_sorted_folding<<<(n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK, ya::UPDATE_BLOCK, 0, stream>>>(
            n, dt, d_partials, n_partials, d_fix_out, d_dX_sorted, d_sorted);
    }
    void predictor_in_sorted_space_mirrored(const int n, const float dt, const float* d_total, float* d_fix_out,
        const int n_active, const Pt* d_dX, const int fix_mode, float* pred_partial, const ya::Guard_band band)
    {
        euler_step_sorted_mirrored<<<(n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK, ya::UPDATE_BLOCK, 0, stream>>>(
            n, dt, d_total, d_fix_out, d_dX_sorted, d_dX, d_sorted, n_active, fix_mode, pred_partial, band);
    }
    void ghosts_in_sorted_space(const int n, const int n_active, const Pt* d_X1)
    {
        if (n_active >= n) return;
        ghosts_into_sorted<<<(n + ya::UPDATE_BLOCK - 1) / ya::UPDATE_BLOCK, ya::UPDATE_BLOCK, 0, stream>>>(
            n, n_active, d_X1, d_sorted);
    }
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void pwints_from_sorted(const int n, Pt* d_dX, const int n_active, const bool has_gen)
    {
        grid.rebuild_sorted(n, d_sorted, d_sorted_v, cube_size, d_resorted, d_resorted_v, stream);
        forces<pw_int, pw_friction>(n, d_resorted, d_resorted_v, d_dX, has_gen, n_active, nullptr);
    }
};

template<typename Pt>
using Grid_solver = Heun_solver<Pt, Grid_computer>;


// Pairwise interactions on the grid restricted to Gabriel-graph neighbours
// (Delile et al. 2017, Marin-Riera et al. 2016; solvers.cuh:505-644).
template<typename Pt>
class Gabriel_computer : public Grid_computer<Pt> {
public:
    float gabriel_coefficient;
    Gabriel_computer(
        int n_max, int grid_size = 50, float cube_size = 1, float gabriel_coefficient = 0.8)
        : Grid_computer<Pt>{n_max, grid_size, cube_size}, gabriel_coefficient{gabriel_coefficient}
    {}
    bool use_sorted_pipeline() const { return false; }

protected:
    template<Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
    void pwints(const int n, const Pt* __restrict__ d_X, const float3* __restrict__ d_old_v,
        Pt* d_dX, const bool has_gen, const int n_active, const bool keep_sorted)
    {
        assert(n_active == n);
        this->grid.build_sorted(n, d_X, d_old_v, this->cube_size, this->d_sorted, this->d_sorted_v);
        ya::gabriel_force<Pt, pw_int, pw_friction><<<(n + 63) / 64, 64>>>(n, this->d_sorted,
            this->d_sorted_v, this->grid.d_cube_id, this->grid.offsets(), this->grid.grid_size,
            this->grid.n_cubes, this->cube_size, gabriel_coefficient, d_dX, has_gen);
    }
};

template<typename Pt>
using Gabriel_solver = Heun_solver<Pt, Gabriel_computer>;
