// libyalla_hip.so -- Pt-agnostic kernels of the MI355X ya||a step path behind
// the C ABI of include/yalla_hip.h.  gfx950 only; wave = 64 lanes.
//
// Grid build (replaces solvers.cuh:349-378,406-417).  The reference bins, fills
// two gs^3 tables, radix-sorts (cube id, point id) pairs and detects segment
// bounds: >= 6 passes over the keys plus 16 B x gs^3 of fills per build.  Cube
// ids are small dense integers, so this build is a counting sort instead:
//
//   k_bin      cube id per cell (binary32, reference association order) and
//              an arrival rank from one returning atomic on count[cube].  Cells
//              are visited in the PREVIOUS build's sorted order (cells move
//              little between builds), so a wavefront's 64 atomics fall on a
//              few neighbouring counters instead of 64 random cache lines, and
//              the later scatter writes neighbouring slots
//   k_scan     per-tile totals of count[] (tile = 2048 cubes), published and collected inside the
//              one launch; exclusive prefix -> offs[], cube_start[], cube_end[] for EVERY
//              cube (so no fills are needed), and count[] re-zeroed in passing
//   k_scatter  slot = offs[cube] + rank (arrival order inside a cube)
//   k_order    restores ascending point id inside each cube (what a stable
//              sort yields) by rank-counting within the cube's segment, and
//              optionally gathers {X, id} and old_v into sorted order so the
//              force kernel streams them.
//
// (Round 5 built the four kernels as the four phases of ONE 1024-thread workgroup for systems of <= 16 k
// cells in <= 32 k cubes -- a step of such a system is bound by its ~14 dependent launches at ~5 us each --
// bit-identical and SLOWER: examples/sorting.cu at 10 k cells 79 -> 141 us per step.  One CU cannot keep
// enough loads in flight: ten cells per thread and phase, each a dependent round trip to the L2, where the
// kernels spread over 40 CUs.  profiles/r05_nonforce_ab.jsonl, tools/micro/r05_small_build.patch.)
//
// Everything is int32/fp32; compile with -ffp-contract=off so the cube id
// arithmetic is the plain IEEE evaluation the oracle restates.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "yalla_hip.h"

namespace {

constexpr int BLOCK = 256;
constexpr int SCAN_ITEMS = 8;                   // cubes per thread in the scan
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;   // cubes per block
constexpr int REDUCE_MAX_BLOCKS = 1024;

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// --- binning ---------------------------------------------------------------
// solvers.cuh:357-360 evaluated in float, left to right:
//   (floor(x/cs) + gs/2) + (floor(y/cs) + gs/2)*gs + (floor(z/cs) + gs/2)*gs*gs
__device__ __forceinline__ int cube_id_of(float x, float y, float z, float cs, int gs)
{
    const float half = (float)(gs / 2);
    const float fgs = (float)gs;
    float fx = floorf(x / cs) + half;
    float fy = (floorf(y / cs) + half) * fgs;
    float fz = ((floorf(z / cs) + half) * fgs) * fgs;
    return (int)((fx + fy) + fz);
}

// Visit order over max(n, n_prev) positions: position s of the previous build's
// sorted order, identity for cells added since (ids >= n_prev); -1 where the
// previous build held a cell that no longer exists (id >= n: the population
// shrank, e.g. a slab's ghost layer).  prev_pid is a permutation of [0, n_prev), so
// every id in [0, n) is visited exactly once.
__device__ __forceinline__ int visit(int s, const int* __restrict__ prev_pid, int n_prev, int n)
{
    const int i = s < n_prev ? prev_pid[s] : s;
    return i < n ? i : -1;
}

__global__ __launch_bounds__(BLOCK) void k_bin(const float* __restrict__ X,
    int stride_f, int n, float cs, int gs, int n_cubes, const int* __restrict__ prev_pid,
    int n_prev, int* __restrict__ cube_of, int* __restrict__ rank, int* __restrict__ count,
    int* __restrict__ status, float* __restrict__ stash, int stash_f,
    const int* __restrict__ d_n, int range_lo, int range_hi, int* publish = nullptr, int publish_seq = 0)
{
    // ya_grid_build_sorted_begin_publish: the count goes to the host from HERE -- {n, sequence number} stored
    // to host memory the device writes directly (fine-grained, system scope), the count first -- instead of by a
    // copy queued in front of this kernel (4 us of stream time per step that the build waited for)
    if (publish && blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(publish, *d_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(publish + 1, publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (d_n) n = min(*d_n, n);  // the count still on its way to the host; n = launch bound
    int s = blockIdx.x * BLOCK + threadIdx.x;
    const int n_visit = max(n, n_prev);
    int id = -1;
    const int i = s < n_visit ? visit(s, prev_pid, n_prev, n) : -1;
    if (i >= 0) {
        const float* p = X + (size_t)i * stride_f;
        const float x = p[0], y = p[1], z = p[2];
        // the point, kept in visit order for k_order (which then reads it nearly
        // in sequence instead of gathering it by id a second time)
        if (stash) {
            float* out = stash + (size_t)s * stash_f;
            out[0] = x;
            out[1] = y;
            out[2] = z;
            for (int k = 3; k < stash_f; k++) out[k] = p[k];
        }
        id = cube_id_of(x, y, z, cs, gs);
        if (id < 0 || id >= n_cubes) {
            atomicOr(status, YA_STATUS_OUT_OF_GRID);
            id = id < 0 ? 0 : n_cubes - 1;
        }
        // ya_grid_set_cube_range: the caller promised cube ids in [range_lo, range_hi) and the
        // prefix sum only covers those; a cell outside is reported and kept inside (memory-safe)
        if (id < range_lo || id >= range_hi) {
            atomicOr(status, YA_STATUS_OUT_OF_RANGE);
            id = id < range_lo ? range_lo : range_hi - 1;
        }
    }
    // Wave-aggregated counting: in visit order equal cube ids sit in adjacent
    // lanes, so each run of equal ids issues ONE atomic (by its first lane, for
    // the run length) instead of one per lane on the same counter.  Any
    // arrival order inside a cube is fine: k_order restores ascending ids.
    const int lane = threadIdx.x & 63;
    const int id_before = __shfl_up(id, 1, 64);
    const bool head = lane == 0 || id != id_before;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = heads & (~0ULL >> (63 - lane));          // heads in lanes <= lane
    const unsigned long long after = lane == 63 ? 0ULL : heads & (~0ULL << (lane + 1));
    const int head_lane = 63 - __builtin_clzll(upto);
    const int run_end = after ? __builtin_ctzll(after) : 64;
    int base = 0;
    if (head && id >= 0) base = atomicAdd(&count[id], run_end - lane);
    base = __shfl(base, head_lane, 64);
    if (s < n_visit) {
        cube_of[s] = id;  // -1: nothing to visit at this position
        rank[s] = base + (lane - head_lane);
    }
}

// --- scan over cubes ---------------------------------------------------------
__device__ __forceinline__ int block_sum(int v, int* sh)
{
    // wave reduce, then 4 waves through LDS
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// Round 5: the tile totals and the prefix in ONE launch (they were k_tile_sum and k_scan, two launches
// of a few microseconds of work each, twice per take_step).  Every block publishes its tile's total
// first -- one 64-bit word {1, total}, stored and read past the XCDs' L2s (device-scope atomics) -- and
// then collects the totals of the tiles before it, waiting for those not yet there.  A block only ever
// waits for blocks with LOWER indices, which the dispatcher started before it and which publish before
// they wait themselves: no cycle.  A word is valid if its upper half is the scan's EPOCH, a counter in device
// memory that the launch's last block moves on when it is through (by then every block has published, hence
// read the epoch): nothing is cleared, no ticket is drawn (1000 blocks drawing tickets from one address were
// 8 us of a 10 M-cell build's scan), and a captured graph replays correctly.
// (first_tile: the scan of a cube range, ya_grid_set_cube_range -- block b works on tile first_tile + b)
__global__ __launch_bounds__(BLOCK) void k_scan(int* __restrict__ count,
    unsigned long long* __restrict__ tile_state, unsigned* __restrict__ d_epoch, int n_cubes, int n,
    int* __restrict__ offs, int* __restrict__ cube_start, int* __restrict__ cube_end,
    const int* __restrict__ d_n, int first_tile, int all_tiles, int* __restrict__ status)
{
    if (d_n) n = min(*d_n, n);
    __shared__ int sh[4];
    __shared__ int sh_wave[4];
    const int tile = first_tile + blockIdx.x;

    size_t base = (size_t)tile * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    int4* c4 = reinterpret_cast<int4*>(count + base);
    int4 a = c4[0], b = c4[1];
    int c[SCAN_ITEMS] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    int total = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) total += c[k];
    // inclusive scan of the 256 thread totals inside each wavefront; the wavefronts' totals
    int incl = total;
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) sh_wave[w] = incl;
    // The epoch is read by thread 0 BEFORE the barrier behind which it publishes, and handed to the block through
    // LDS: the launch's last block moves the epoch on once it has seen every block's word, so no thread may be
    // left to read it from memory after its own block has published.
    __shared__ unsigned sh_epoch;
    if (threadIdx.x == 0) sh_epoch = *d_epoch;  // (written by the last scan: a kernel boundary lies between)
    __syncthreads();
    const unsigned long long epoch = sh_epoch;
    if (threadIdx.x == 0)
        __hip_atomic_store(&tile_state[tile],
            (epoch << 32) | (unsigned)(sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3]), __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
    // cells in all tiles before this one (a cube range: no cell lies below its first tile)
    // (eight words per thread asked for at once: the loads go past the L2s, ~2 us each, and a block of a
    // 1000-tile grid needs four or five of them per thread -- one after the other they were 10 us of the launch)
    int before = 0;
    for (int t0 = first_tile + threadIdx.x; t0 < tile; t0 += 8 * BLOCK) {
        unsigned long long v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = t0 + k * BLOCK < tile
                       ? __hip_atomic_load(&tile_state[t0 + k * BLOCK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : epoch << 32;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // REQUIREMENT (not promised by HIP, true of every dispatcher this was run on): the blocks of lower
            // index a block waits for have been STARTED before it.  Should that ever fail (another partition
            // mode, a runtime that dispatches out of order) the wait gives up after ~0.2 s instead of hanging
            // every grid build: YA_STATUS_SCAN_STALLED is raised (Grid::check_status aborts with the reason)
            // and the arrays of this build are wrong, as the message says.
            int spins = 0;
            while ((v[k] >> 32) != epoch) {  // not published yet
                __builtin_amdgcn_s_sleep(1);
                v[k] = __hip_atomic_load(&tile_state[t0 + k * BLOCK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > (1 << 22)) {
                    atomicOr(status, YA_STATUS_SCAN_STALLED);
                    break;
                }
            }
            before += (int)(unsigned)v[k];
        }
    }
    before = block_sum(before, sh);
    int wave_off = 0;
    for (int k = 0; k < w; k++) wave_off += sh_wave[k];
    int run = before + wave_off + incl - total;

    int o_[SCAN_ITEMS], s_[SCAN_ITEMS], e_[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        o_[k] = run;
        s_[k] = c[k] > 0 ? run : -1;             // solvers.cuh:411 sentinel
        e_[k] = c[k] > 0 ? run + c[k] - 1 : -2;  // solvers.cuh:412 sentinel
        run += c[k];
    }
    int4* o4 = reinterpret_cast<int4*>(offs + base);
    int4* s4 = reinterpret_cast<int4*>(cube_start + base);
    int4* e4 = reinterpret_cast<int4*>(cube_end + base);
    o4[0] = make_int4(o_[0], o_[1], o_[2], o_[3]);
    o4[1] = make_int4(o_[4], o_[5], o_[6], o_[7]);
    s4[0] = make_int4(s_[0], s_[1], s_[2], s_[3]);
    s4[1] = make_int4(s_[4], s_[5], s_[6], s_[7]);
    e4[0] = make_int4(e_[0], e_[1], e_[2], e_[3]);
    e4[1] = make_int4(e_[4], e_[5], e_[6], e_[7]);
    c4[0] = make_int4(0, 0, 0, 0);  // leave count[] zeroed for the next build
    c4[1] = make_int4(0, 0, 0, 0);
    if (tile == all_tiles - 1 && threadIdx.x == BLOCK - 1)
        offs[(size_t)all_tiles * SCAN_TILE] = n;
    // the launch's last block has seen every other block's word: all of them have read the epoch
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const unsigned next = (unsigned)epoch + 1u;
        __hip_atomic_store(d_epoch, next ? next : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(BLOCK) void k_scatter(const int* __restrict__ cube_of,
    const int* __restrict__ rank, const int* __restrict__ offs, int n,
    const int* __restrict__ prev_pid, int n_prev, int* __restrict__ arrival_pid,
    int* __restrict__ cube_id_sorted, int* __restrict__ arrival_src,
    const unsigned* __restrict__ entries, int entry_w, int id_word, const int* __restrict__ d_n)
{
    if (d_n) n = min(*d_n, n);
    int s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= max(n, n_prev)) return;
    int c = cube_of[s];
    if (c < 0) return;
    int slot = offs[c] + rank[s];
    // the id of the visited cell: position s of the visit order, or (rebuild from
    // sorted cells) the id stored in entry s
    arrival_pid[slot] = entries ? (int)entries[(size_t)s * entry_w + id_word]
                                : visit(s, prev_pid, n_prev, n);
    cube_id_sorted[slot] = c;
    if (arrival_src) arrival_src[slot] = s;
}

// Rank-count inside the cube's segment: slot of point p = segment start +
// #(points of the segment with a smaller id).  NW = floats per point (0 = no
// gather).
// Cost bound: a cell compares its id with the m cells of its own cube (m loads, m compares),
// so a cube costs m^2 -- the same order as, and 27 times less than, the 27 m candidate tests
// the force kernel then runs for each of those cells.  Dense or collapsed populations slow
// the force evaluation down long before they slow this kernel down; a plain Grid::build
// used on its own (model kernels that only need the cube lists) pays the m^2 as well.
template<int NW>
__global__ __launch_bounds__(BLOCK) void k_order(const int* __restrict__ arrival_pid,
    const int* __restrict__ cube_id_sorted, const int* __restrict__ offs, int n,
    int* __restrict__ point_id, int* __restrict__ next_prev_pid, const float* __restrict__ X,
    int stride_f, const float* __restrict__ old_v, float* __restrict__ sorted_X, int entry_f,
    float4* __restrict__ sorted_v, const float* __restrict__ stash,
    const int* __restrict__ arrival_src)
{
    int s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n) return;
    int c = cube_id_sorted[s];
    int a = offs[c], b = offs[c + 1];
    int p = arrival_pid[s];
    int smaller = 0;
    for (int t = a; t < b; t++) smaller += arrival_pid[t] < p;
    int dst = a + smaller;
    point_id[dst] = p;
    next_prev_pid[dst] = p;
    if (NW > 0) {
        const float* src =
            stash ? stash + (size_t)arrival_src[s] * NW : X + (size_t)p * stride_f;
        float* out = sorted_X + (size_t)dst * entry_f;
        float v[NW > 0 ? NW : 1];
#pragma unroll
        for (int k = 0; k < NW; k++) v[k] = src[k];
#pragma unroll
        for (int k = 0; k < NW; k++) out[k] = v[k];
        out[NW] = __int_as_float(p);
        const float* ov = old_v + (size_t)p * 3;
        sorted_v[dst] = make_float4(ov[0], ov[1], ov[2], 0.f);
    }
}

// Re-sort of cells that are ALREADY in (an earlier) cube-sorted order, e.g. the
// second Heun stage: positions moved a little, the previous sorted arrays are the
// input, so every read is coalesced or local and nothing is gathered from the
// original-order arrays.  arrival_src[slot] = index of the cell in the previous
// sorted arrays, arrival_pid[slot] = its id (k_scatter copied it from the entry).
template<int EW>
__global__ __launch_bounds__(BLOCK) void k_order_from(const int* __restrict__ arrival_pid,
    const int* __restrict__ arrival_src, const int* __restrict__ cube_id_sorted,
    const int* __restrict__ offs, int n,
    const unsigned* __restrict__ prev_entries, const float4* __restrict__ prev_v,
    int* __restrict__ point_id, int* __restrict__ next_prev_pid, unsigned* __restrict__ sorted_out,
    float4* __restrict__ sorted_v_out)
{
    int s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n) return;
    const int c = cube_id_sorted[s];
    const int a = offs[c], b = offs[c + 1];
    const int src = arrival_src[s];
    const unsigned* mine = prev_entries + (size_t)src * EW;
    const int p = arrival_pid[s];
    int smaller = 0;
    for (int t = a; t < b; t++) smaller += arrival_pid[t] < p;
    const int dst = a + smaller;
    point_id[dst] = p;
    next_prev_pid[dst] = p;
    unsigned w[EW];
#pragma unroll
    for (int k = 0; k < EW; k++) w[k] = mine[k];
    unsigned* out = sorted_out + (size_t)dst * EW;
#pragma unroll
    for (int k = 0; k < EW; k++) out[k] = w[k];
    sorted_v_out[dst] = prev_v[src];
}

// --- deterministic reduction -----------------------------------------------
// B = clamp(ceil(n/256), 1, 1024) blocks; lane (b, t) sums i = b*256 + t,
// += B*256 ... serially; block folds 256 lanes by halving through LDS; then one
// block sums the B partials the same way.  The oracle restates this order
// (oracle/yalla_host.hpp, YA_REDUCE_TREE).
template<int NW>
__device__ __forceinline__ void fold256(float (&acc)[NW], float* sh /* [NW][256] */)
{
    // lane[t] += lane[t + s] for s = 128 ... 1 (the documented order).  Round 5: from s = 32 down the
    // operands sit in ONE wavefront and travel by shuffle instead of through LDS and a workgroup
    // barrier per step -- the same additions of the same operands, so the same bits: a reduction
    // kernel is 8 barriers shorter (4.9 -> 3.4 us per launch at any size).
#pragma unroll
    for (int k = 0; k < NW; k++) sh[k * BLOCK + threadIdx.x] = acc[k];
    __syncthreads();
    if ((int)threadIdx.x < 128) {
#pragma unroll
        for (int k = 0; k < NW; k++)
            sh[k * BLOCK + threadIdx.x] = sh[k * BLOCK + threadIdx.x] + sh[k * BLOCK + threadIdx.x + 128];
    }
    __syncthreads();
    if ((int)threadIdx.x < 64) {
        float v[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) v[k] = sh[k * BLOCK + threadIdx.x] + sh[k * BLOCK + threadIdx.x + 64];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
            for (int k = 0; k < NW; k++) v[k] = v[k] + __shfl_down(v[k], s, 64);
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < NW; k++) sh[k * BLOCK] = v[k];
        }
    }
    __syncthreads();
}

// (Round 5 tried ONE launch -- the block that draws the last ticket folds the partials, same order --
// and it is slower, 13.6 us against 9.8 for the two launches at 1024 blocks: partial sums that cross
// XCDs must go past the L2s (device-scope stores, an acknowledged write, tickets drawn one after the
// other, uncached loads), 8-9 us of serial round trips where a launch of one block costs 4.9; with
// __threadfence() instead, 66 us: every block writes back and invalidates its XCD's L2.
// profiles/r05_nonforce_ab.jsonl)
template<int NW>
__global__ __launch_bounds__(BLOCK) void k_reduce_partial(
    const float* __restrict__ v, int n, float* __restrict__ partials)
{
    __shared__ float sh[NW * BLOCK];
    float acc[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) acc[k] = 0.f;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
        const float* p = v + (size_t)i * NW;
#pragma unroll
        for (int k = 0; k < NW; k++) acc[k] = acc[k] + p[k];
    }
    fold256<NW>(acc, sh);
    if (threadIdx.x < NW) partials[(size_t)blockIdx.x * NW + threadIdx.x] = sh[threadIdx.x * BLOCK];
}

// PACKED (z-slab decomposition): out = {sum[NW], n & 4095, n >> 12} -- what a rank puts into the
// all-reduce of a stage: the sum over its own cells and their count in two pieces that stay exact
// under a float sum.  Otherwise out = {mean[NW], sum[NW]}.
template<int NW, bool PACKED = false>
__global__ __launch_bounds__(BLOCK) void k_reduce_final(
    const float* __restrict__ partials, int n_partials, int n, float* __restrict__ out)
{
    __shared__ float sh[NW * BLOCK];
    float acc[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) acc[k] = 0.f;
    for (int p = threadIdx.x; p < n_partials; p += BLOCK) {
#pragma unroll
        for (int k = 0; k < NW; k++) acc[k] = acc[k] + partials[(size_t)p * NW + k];
    }
    fold256<NW>(acc, sh);
    if (threadIdx.x < NW) {
        float sum = sh[threadIdx.x * BLOCK];
        // Pt / n  ==  Pt * float(1. / float(n))   (dtypes.cuh:202-217)
        if (PACKED) {
            out[threadIdx.x] = sum;
            if (threadIdx.x == 0) {
                out[NW] = (float)(n & 4095);
                out[NW + 1] = (float)(n >> 12);
            }
        } else {
            float inv = (float)(1. / (double)(float)n);
            out[threadIdx.x] = sum * inv;
            out[NW + threadIdx.x] = sum;
        }
    }
}

// --- ordered selection ---------------------------------------------------------
constexpr int SEL_ITEMS = 8;
constexpr int SEL_TILE = BLOCK * SEL_ITEMS;

__device__ __forceinline__ bool in_range(const float* __restrict__ X, int stride_f, int i, int n,
    float z_min, float z_max)
{
    if (i >= n) return false;
    const float z = X[(size_t)i * stride_f + 2];
    return z >= z_min && z < z_max;
}

__global__ __launch_bounds__(BLOCK) void k_select_count(const float* __restrict__ X, int stride_f,
    int n, float z_min, float z_max, int* __restrict__ tile_counts)
{
    __shared__ int sh[4];
    const int base = blockIdx.x * SEL_TILE + threadIdx.x * SEL_ITEMS;
    int c = 0;
#pragma unroll
    for (int k = 0; k < SEL_ITEMS; k++) c += in_range(X, stride_f, base + k, n, z_min, z_max);
    const int total = block_sum(c, sh);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(BLOCK) void k_select_write(const float* __restrict__ X, int stride_f,
    int n, float z_min, float z_max, const int* __restrict__ tile_counts, int* __restrict__ idx,
    int* __restrict__ count)
{
    __shared__ int sh[4];
    __shared__ int sh_wave[4];
    int before = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += BLOCK) before += tile_counts[t];
    before = block_sum(before, sh);

    const int base = blockIdx.x * SEL_TILE + threadIdx.x * SEL_ITEMS;
    bool keep[SEL_ITEMS];
    int c = 0;
#pragma unroll
    for (int k = 0; k < SEL_ITEMS; k++) {
        keep[k] = in_range(X, stride_f, base + k, n, z_min, z_max);
        c += keep[k];
    }
    int incl = c;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) sh_wave[w] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int k = 0; k < w; k++) wave_off += sh_wave[k];
    int out = before + wave_off + incl - c;
#pragma unroll
    for (int k = 0; k < SEL_ITEMS; k++)
        if (keep[k]) idx[out++] = base + k;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == BLOCK - 1) *count = out;
}

__global__ __launch_bounds__(BLOCK) void k_gather_rows(const float* __restrict__ src, int row_f,
    const int* __restrict__ idx, const int* __restrict__ count, int cap, float* __restrict__ dst)
{
    const int m = min(*count, cap);
    // one thread per float of the output
    for (long e = (long)blockIdx.x * BLOCK + threadIdx.x; e < (long)m * row_f;
         e += (long)gridDim.x * BLOCK) {
        const int k = (int)(e / row_f), f = (int)(e % row_f);
        dst[e] = src[(size_t)idx[k] * row_f + f];
    }
}

// Two gathers from one array in one launch (a z-slab packs the rows its lower and its upper
// neighbour wait for): dst0[k] = src[idx0[k]], k < min(*count0, cap); dst1 likewise.  A null
// index list is an empty gather.
__global__ __launch_bounds__(BLOCK) void k_gather_rows_pair(const float* __restrict__ src, int row_f,
    const int* __restrict__ idx0, const int* __restrict__ count0, float* __restrict__ dst0,
    const int* __restrict__ idx1, const int* __restrict__ count1, float* __restrict__ dst1, int cap)
{
    const int m0 = idx0 ? min(max(*count0, 0), cap) : 0;
    const int m1 = idx1 ? min(max(*count1, 0), cap) : 0;
    const long f0 = (long)m0 * row_f, total = f0 + (long)m1 * row_f;
    for (long e = (long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long)gridDim.x * BLOCK) {
        const bool second = e >= f0;
        const long r = second ? e - f0 : e;
        const int k = (int)(r / row_f), f = (int)(r % row_f);
        const int* idx = second ? idx1 : idx0;
        (second ? dst1 : dst0)[r] = src[(size_t)idx[k] * row_f + f];
    }
}

// dst rows [n_own, n_own + c_lo) = src_lo rows [0, c_lo), then c_hi rows of src_hi; the
// counts are the first int of each message (clamped to [0, cap]); a missing message
// counts as empty.  One thread per float.
__global__ __launch_bounds__(BLOCK) void k_append_rows(float* __restrict__ dst, int row_f, int n_own,
    const float* __restrict__ src_lo, const int* __restrict__ count_lo,
    const float* __restrict__ src_hi, const int* __restrict__ count_hi, int cap,
    int* __restrict__ n_out)
{
    const int c_lo = count_lo ? min(max(*count_lo, 0), cap) : 0;
    const int c_hi = count_hi ? min(max(*count_hi, 0), cap) : 0;
    const long total = (long)(c_lo + c_hi) * row_f;
    for (long e = (long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long)gridDim.x * BLOCK) {
        const long lo_floats = (long)c_lo * row_f;
        dst[(size_t)n_own * row_f + e] = e < lo_floats ? src_lo[e] : src_hi[e - lo_floats];
    }
    if (n_out && blockIdx.x == 0 && threadIdx.x == 0) *n_out = n_own + c_lo + c_hi;
}


// --- the shader clock, measured --------------------------------------------------------------
// One wavefront watches s_memtime (shader cycles) against wall_clock64 (a constant 100 MHz counter)
// for `ticks` wall ticks: the clock the chip runs at NOW, beside whatever else it is running.
__global__ void k_shader_clock(unsigned long long ticks, unsigned long long* out)
{
    if (threadIdx.x != 0) return;
    unsigned long long c0, c1;
    const unsigned long long w0 = wall_clock64();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    unsigned long long w1 = w0;
    while (w1 - w0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        w1 = wall_clock64();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    out[0] = c1 - c0;
    out[1] = w1 - w0;
}

// --- z-slab decomposition: drift guard, fixed point, a stage's all-reduce payload -------------
__device__ __forceinline__ float block_max(float v, float* sh /* [4] */)
{
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__global__ __launch_bounds__(BLOCK) void k_copy_component(const float* __restrict__ src, int stride_f,
    int n, float* __restrict__ dst)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[(size_t)i * stride_f];
}

__global__ __launch_bounds__(BLOCK) void k_find_id(const int* __restrict__ ids, int n, int id, int* __restrict__ index)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n && ids[i] == id) *index = i;  // ids are unique
}

// partial[b] = max over block b's share of w_i |a[i] - b[i]|, w_i = 1 if b[i] lies within `width` of
// lo_face or hi_face, else 1/2 (grid-stride; NaN differences count as +inf)
__global__ __launch_bounds__(BLOCK) void k_max_abs_diff(const float* __restrict__ a, int a_stride_f,
    const float* __restrict__ b, int b_stride_f, int n, float lo_face, float hi_face, float width,
    float* __restrict__ partial)
{
    __shared__ float sh[4];
    float m = 0.f;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
        const float z = b[(size_t)i * b_stride_f];
        const float w = fabsf(z - lo_face) <= width || fabsf(z - hi_face) <= width ? 1.f : 0.5f;
        const float d = fabsf(a[(size_t)i * a_stride_f] - z) * w;
        m = d == d ? fmaxf(m, d) : INFINITY;
    }
    m = block_max(m, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = m;
}

// The drift guard of a z-slab (include/slab_logic.inc), between a step's two stages: from the
// per-block maxima of |z - z at selection| (own and mirrored cells, as the previous step left them)
// and of this step's predictor |dz|.  state = {moved, predicted, request, error}: `error` if the
// second stage is about to compute forces with a cell further than `limit` from where it was when
// the mirrored cells were chosen (moved + predicted; the first stage saw `moved`, which is less),
// `request` if the steps until a vote can act -- `lag_steps` of them at the present pace -- would
// get there.  Folded into the last kernel of the second stage's reduction (k_slab_pack_final), which
// puts the votes into the all-reduce's payload; ya_slab_guard_update runs it by itself.
struct Guard_args {
    float* moved_partial;
    int n_moved;
    float* pred_partial;
    int n_pred;
    float limit, lag_steps;
    float* state;  // NULL: no guard
};
__device__ __forceinline__ void guard_fold(const Guard_args& g, float* sh /* [4] */)
{
    // (several thousand partials, one workgroup: four independent loads in flight per thread and pass)
    float a4[4] = {0.f, 0.f, 0.f, 0.f}, p4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < g.n_moved; k += 4 * BLOCK) {
#pragma unroll
        for (int u = 0; u < 4; u++) a4[u] = fmaxf(a4[u], k + u * BLOCK < g.n_moved ? g.moved_partial[k + u * BLOCK] : 0.f);
    }
    for (int k = threadIdx.x; k < g.n_pred; k += 4 * BLOCK) {
#pragma unroll
        for (int u = 0; u < 4; u++) p4[u] = fmaxf(p4[u], k + u * BLOCK < g.n_pred ? g.pred_partial[k + u * BLOCK] : 0.f);
    }
    float a = fmaxf(fmaxf(a4[0], a4[1]), fmaxf(a4[2], a4[3])), p = fmaxf(fmaxf(p4[0], p4[1]), fmaxf(p4[2], p4[3]));
    a = block_max(a, sh);
    p = block_max(p, sh);
    // (the update kernels fold their maxima into these lists by atomic max: left zeroed for the next step)
    for (int k = threadIdx.x; k < g.n_moved; k += BLOCK) g.moved_partial[k] = 0.f;
    for (int k = threadIdx.x; k < g.n_pred; k += BLOCK) g.pred_partial[k] = 0.f;
    if (threadIdx.x == 0) {
        g.state[0] = a;
        g.state[1] = p;
        g.state[2] = !(a + g.lag_steps * p <= g.limit) ? 1.f : 0.f;
        g.state[3] = !(a + p <= g.limit) ? 1.f : g.state[3];  // sticky until the next selection
    }
    __syncthreads();
}
__global__ __launch_bounds__(BLOCK) void k_slab_guard(const Guard_args g)
{
    __shared__ float sh[4];
    guard_fold(g, sh);
}

// The last kernel of a rank's contribution to a stage's all-reduce (ya_slab_pack): k_reduce_final's
// packed sum {sum[NW], n & 4095, n >> 12}, then out[NW + 2] = this rank's vote for an early
// re-selection, out[NW + 3] = its error vote (the drift guard's, folded here first if `fold_guard`,
// plus 1 if host_error), out[NW + 4 .. NW + 6] = x, y, z of row *fix_index of v if this rank owns the
// fixed point (set_fixed(i), set_fixed_xy(i): every other rank adds zeros), out[NW + 7] = 0.
template<int NW>
__global__ __launch_bounds__(BLOCK) void k_slab_pack_final(const float* __restrict__ partials, int n_partials, int n,
    const float* __restrict__ v, const Guard_args guard, int fold_guard, int with_votes, int host_error,
    const int* __restrict__ fix_index, float* __restrict__ out)
{
    __shared__ float sh[NW * BLOCK];
    float acc[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) acc[k] = 0.f;
    for (int p = threadIdx.x; p < n_partials; p += BLOCK) {
#pragma unroll
        for (int k = 0; k < NW; k++) acc[k] = acc[k] + partials[(size_t)p * NW + k];
    }
    fold256<NW>(acc, sh);
    if (threadIdx.x < NW) out[threadIdx.x] = sh[threadIdx.x * BLOCK];
    __syncthreads();
    if (fold_guard && guard.state) guard_fold(guard, sh);
    if (threadIdx.x == 0) {
        out[NW] = (float)(n & 4095);
        out[NW + 1] = (float)(n >> 12);
        const bool votes = with_votes && guard.state;
        out[NW + 2] = votes ? guard.state[2] : 0.f;
        out[NW + 3] = (votes ? guard.state[3] : 0.f) + (host_error ? 1.f : 0.f);
        const int f = fix_index ? *fix_index : -1;
        for (int k = 0; k < 3; k++) out[NW + 4 + k] = f >= 0 ? v[(size_t)f * NW + k] : 0.f;
        out[NW + 7] = 0.f;
    }
}


// --- z-slab decomposition: a cell's fields {X, old_v, global id} moved together ------------------
// Up to three arrays with their own row widths (in floats) handled by ONE launch: a message is
// packed, two messages are appended, holes are filled.  One thread per float of a cell's record.
struct Fields3 {
    const float* src[3];
    float* dst[3];
    int row_f[3];
};
__device__ __forceinline__ void locate(const Fields3& f, long e, int total_f, int* cell, int* field, int* off)
{
    *cell = (int)(e / total_f);
    int r = (int)(e % total_f);
    int fl = 0;
    while (fl < 2 && r >= f.row_f[fl]) r -= f.row_f[fl++];
    *field = fl;
    *off = r;
}
// dst[f][k] = src[f][idx[k]] for k < min(*count, cap); header[0] = *count as it is (a receiver sees
// an overflow), header[1 .. 3] = 0
__global__ __launch_bounds__(BLOCK) void k_pack_cells(const Fields3 f, const int* __restrict__ idx,
    const int* __restrict__ count, int cap, int* __restrict__ header)
{
    const int total_f = f.row_f[0] + f.row_f[1] + f.row_f[2];
    const int m = min(max(*count, 0), cap);
    if (header && blockIdx.x == 0 && threadIdx.x < 4) header[threadIdx.x] = threadIdx.x == 0 ? *count : 0;
    for (long e = (long)blockIdx.x * BLOCK + threadIdx.x; e < (long)m * total_f; e += (long)gridDim.x * BLOCK) {
        int k, fl, off;
        locate(f, e, total_f, &k, &fl, &off);
        f.dst[fl][(size_t)k * f.row_f[fl] + off] = f.src[fl][(size_t)idx[k] * f.row_f[fl] + off];
    }
}
// rows [n_own, n_own + c_lo) of every field from the lower message, then c_hi rows from the upper one
// (counts = the messages' first ints, clamped to [0, cap]; a missing message counts as empty);
// *n_out = n_own + c_lo + c_hi, counts_out[0 .. 1] = the counts AS SENT
struct Append3 {
    const float* lo[3];
    const float* hi[3];
    float* dst[3];
    int row_f[3];
};
__global__ __launch_bounds__(BLOCK) void k_append_cells(const Append3 a, int n_own, const int* __restrict__ count_lo,
    const int* __restrict__ count_hi, int cap, int* __restrict__ n_out, int* __restrict__ counts_out)
{
    const int sent_lo = count_lo ? *count_lo : 0, sent_hi = count_hi ? *count_hi : 0;
    const int c_lo = min(max(sent_lo, 0), cap), c_hi = min(max(sent_hi, 0), cap);
    const int total_f = a.row_f[0] + a.row_f[1] + a.row_f[2];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n_out) *n_out = n_own + c_lo + c_hi;
        if (counts_out) {
            counts_out[0] = sent_lo;
            counts_out[1] = sent_hi;
        }
    }
    Fields3 f;
    for (int k = 0; k < 3; k++) f.row_f[k] = a.row_f[k];
    for (long e = (long)blockIdx.x * BLOCK + threadIdx.x; e < (long)(c_lo + c_hi) * total_f; e += (long)gridDim.x * BLOCK) {
        int k, fl, off;
        locate(f, e, total_f, &k, &fl, &off);
        const float* src = k < c_lo ? a.lo[fl] + (size_t)k * a.row_f[fl] : a.hi[fl] + (size_t)(k - c_lo) * a.row_f[fl];
        a.dst[fl][(size_t)(n_own + k) * a.row_f[fl] + off] = src[off];
    }
}
// Cells that left a slab leave holes in its arrays; the holes below n_new are filled with the cells
// that stay but sit at or above n_new (as many: both are what is missing from / surplus to the first
// n_new rows).  Holes = the entries < n_new of the two ascending lists of leavers (lower face first),
// movers[k] + n_new = the k-th staying cell of the tail, ascending.  Every other cell keeps its row:
// the grid's memory of the last order stays good, and nothing but the few movers is copied.
__device__ __forceinline__ int first_not_below(const int* __restrict__ list, int n, int bound)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (list[mid] < bound) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(BLOCK) void k_fill_holes(const Fields3 f, const int* __restrict__ leave_lo,
    const int* __restrict__ count_lo, const int* __restrict__ leave_hi, const int* __restrict__ count_hi,
    const int* __restrict__ movers, const int* __restrict__ count_movers, int n_new)
{
    const int a = leave_lo ? first_not_below(leave_lo, *count_lo, n_new) : 0;
    const int b = leave_hi ? first_not_below(leave_hi, *count_hi, n_new) : 0;
    const int holes = min(a + b, *count_movers);
    const int total_f = f.row_f[0] + f.row_f[1] + f.row_f[2];
    for (long e = (long)blockIdx.x * BLOCK + threadIdx.x; e < (long)holes * total_f; e += (long)gridDim.x * BLOCK) {
        int k, fl, off;
        locate(f, e, total_f, &k, &fl, &off);
        const int hole = k < a ? leave_lo[k] : leave_hi[k - a];
        const int mover = n_new + movers[k];
        f.dst[fl][(size_t)hole * f.row_f[fl] + off] = f.src[fl][(size_t)mover * f.row_f[fl] + off];
    }
}

template<int NW>
int launch_reduce(const float* v, int n, float* out, float* ws, hipStream_t st, bool packed = false)
{
    int B = ceil_div(n, BLOCK);
    if (B < 1) B = 1;
    if (B > REDUCE_MAX_BLOCKS) B = REDUCE_MAX_BLOCKS;
    k_reduce_partial<NW><<<B, BLOCK, 0, st>>>(v, n, ws);
    if (packed)
        k_reduce_final<NW, true><<<1, BLOCK, 0, st>>>(ws, B, n, out);
    else
        k_reduce_final<NW><<<1, BLOCK, 0, st>>>(ws, B, n, out);
    return (int)hipGetLastError();
}

template<int NW>
int launch_partials(const float* v, int n, float* ws, hipStream_t st)
{
    int B = ceil_div(n, BLOCK);
    if (B < 1) B = 1;
    if (B > REDUCE_MAX_BLOCKS) B = REDUCE_MAX_BLOCKS;
    k_reduce_partial<NW><<<B, BLOCK, 0, st>>>(v, n, ws);
    return B;
}

struct Pack_extra {
    Guard_args guard;
    int fold_guard, with_votes, host_error;
    const int* fix_index;
};
template<int NW>
int launch_pack(const float* v, int n, float* out, float* ws, hipStream_t st, const Pack_extra& x)
{
    int B = ceil_div(n, BLOCK);
    if (B < 1) B = 1;
    if (B > REDUCE_MAX_BLOCKS) B = REDUCE_MAX_BLOCKS;
    k_reduce_partial<NW><<<B, BLOCK, 0, st>>>(v, n, ws);
    k_slab_pack_final<NW><<<1, BLOCK, 0, st>>>(ws, B, n, v, x.guard, x.fold_guard, x.with_votes, x.host_error, x.fix_index, out);
    return (int)hipGetLastError();
}

}  // namespace

struct ya_grid {
    int n_max, grid_size, n_cubes, n_tiles;
    size_t padded;  // n_cubes rounded up to whole scan tiles
    int *d_cube_id, *d_point_id, *d_cube_start, *d_cube_end;  // public
    int *d_offs, *d_count;                                    // private
    unsigned long long* d_tile_state;  // k_scan: a tile's published total {epoch, total}, [n_tiles]
    unsigned* d_scan_epoch;            // ... valid if its upper half is this
    int *d_cube_of, *d_rank, *d_arrival;                      // private, [n_max]
    int *d_arrival_src;  // visit position of the cell that arrived in a slot
    float* d_stash;      // the points in visit order (ya_grid_build_sorted), lazily sized
    size_t stash_bytes;
    float* begun_stash;  // the stash the build in progress filled (nullptr: none)
    int *d_prev_pid;  // private copy of the last build's point ids (visit order)
    int n_prev;       // cells in that build; 0 = none / unusable
    int* d_status;
    // ya_grid_set_cube_range: the cubes that can hold cells, as whole scan tiles
    // [range_first_tile, range_end_tile) = cube ids [range_lo, range_hi); range_n = the cell count
    // of the last build that scanned EVERY tile since the range was set (-1: none yet): offs[] above
    // the range holds that count, so only builds of the same count may scan the range alone
    int range_first_tile, range_end_tile, range_lo, range_hi, range_n;
};

#define YA_TRY(expr)                      \
    do {                                  \
        hipError_t e_ = (expr);           \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

extern "C" {

int ya_abi_version(void) { return YA_ABI_VERSION; }

int ya_malloc(void** p, size_t bytes) { return (int)hipMalloc(p, bytes ? bytes : 4); }
int ya_free(void* p) { return (int)hipFree(p); }
int ya_memset_async(void* p, int value, size_t bytes, void* stream)
{
    if (bytes == 0) return 0;
    return (int)hipMemsetAsync(p, value, bytes, (hipStream_t)stream);
}
int ya_memcpy_h2d(void* d, const void* h, size_t bytes)
{
    return (int)hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
}
int ya_memcpy_d2h(void* h, const void* d, size_t bytes)
{
    return (int)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
}
int ya_host_alloc(void** h, size_t bytes)
{
    if (!h) return (int)hipErrorInvalidValue;
    return (int)hipHostMalloc(h, bytes ? bytes : 4, hipHostMallocDefault);
}
int ya_host_free(void* h) { return h ? (int)hipHostFree(h) : 0; }
int ya_memcpy_d2d_async(void* dst, const void* src, size_t bytes, void* stream)
{
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}
int ya_device_synchronize(void) { return (int)hipDeviceSynchronize(); }

int ya_get_n(const int* d_n, int* n_out)
{
    return (int)hipMemcpy(n_out, d_n, sizeof(int), hipMemcpyDeviceToHost);
}

struct ya_n_reader {
    int* h_n;  // pinned, fine-grained: {n, sequence number of the publishing kernel}
    int* d_view;  // the same two words as the device addresses them
    hipEvent_t done;
    int seq;        // last sequence number handed to a kernel
    bool published; // the pending read is a kernel's store (ya_grid_build_sorted_begin_publish), not a copy
    hipStream_t stream;
};

int ya_n_reader_create(ya_n_reader** out)
{
    if (!out) return (int)hipErrorInvalidValue;
    ya_n_reader* r = (ya_n_reader*)calloc(1, sizeof(ya_n_reader));
    if (!r) return (int)hipErrorOutOfMemory;
    hipError_t e = hipHostMalloc((void**)&r->h_n, 2 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) {
        r->h_n[0] = r->h_n[1] = 0;
        e = hipHostGetDevicePointer((void**)&r->d_view, r->h_n, 0);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (r->h_n) (void)hipHostFree(r->h_n);
        free(r);
        return (int)e;
    }
    *out = r;
    return 0;
}
int ya_n_reader_destroy(ya_n_reader* r)
{
    if (!r) return 0;
    (void)hipHostFree(r->h_n);
    (void)hipEventDestroy(r->done);
    free(r);
    return 0;
}
int ya_n_read_begin(ya_n_reader* r, const int* d_n, void* stream)
{
    if (!r || !d_n) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    r->published = false;
    YA_TRY(hipMemcpyAsync(r->h_n, d_n, sizeof(int), hipMemcpyDeviceToHost, st));
    return (int)hipEventRecord(r->done, st);
}
int ya_n_read_end(ya_n_reader* r, int* n_out)
{
    if (!r || !n_out) return (int)hipErrorInvalidValue;
    if (r->published) {
        // the binning kernel's first thread stores {n, seq}: spin on the sequence number (the kernel is at
        // most a few microseconds away); a stream that fails instead of running it ends the wait with its error
        r->published = false;
        for (long spins = 0;; spins++) {
            if (__atomic_load_n(&r->h_n[1], __ATOMIC_ACQUIRE) == r->seq) break;
            if (spins >= 2000000 && (spins & 0xfffff) == 0) {  // ~ms of spinning: is the stream still alive?
                const hipError_t e = hipStreamQuery(r->stream);
                if (e != hipSuccess && e != hipErrorNotReady) return (int)e;
                if (e == hipSuccess && __atomic_load_n(&r->h_n[1], __ATOMIC_ACQUIRE) != r->seq)
                    return (int)hipErrorUnknown;  // the stream drained and the kernel never published
            }
        }
        *n_out = __atomic_load_n(&r->h_n[0], __ATOMIC_RELAXED);
        return 0;
    }
    YA_TRY(hipEventSynchronize(r->done));
    *n_out = *r->h_n;
    return 0;
}

static int grid_allocate(ya_grid* g, int n_max)
{
    size_t nb = (size_t)(n_max > 0 ? n_max : 1) * sizeof(int);
    size_t cb = (g->padded + 4) * sizeof(int);
    YA_TRY(hipMalloc(&g->d_cube_id, nb));
    YA_TRY(hipMalloc(&g->d_point_id, nb));
    YA_TRY(hipMalloc(&g->d_cube_of, nb));
    YA_TRY(hipMalloc(&g->d_rank, nb));
    YA_TRY(hipMalloc(&g->d_arrival, nb));
    YA_TRY(hipMalloc(&g->d_arrival_src, nb));
    YA_TRY(hipMalloc(&g->d_prev_pid, nb));
    YA_TRY(hipMalloc(&g->d_cube_start, cb));
    YA_TRY(hipMalloc(&g->d_cube_end, cb));
    YA_TRY(hipMalloc(&g->d_offs, cb));
    YA_TRY(hipMalloc(&g->d_count, cb));
    YA_TRY(hipMalloc(&g->d_tile_state, (size_t)g->n_tiles * sizeof(unsigned long long)));
    YA_TRY(hipMalloc(&g->d_scan_epoch, sizeof(unsigned)));
    YA_TRY(hipMemset(g->d_tile_state, 0, (size_t)g->n_tiles * sizeof(unsigned long long)));
    {
        const unsigned one = 1;  // (the zeroed words are invalid)
        YA_TRY(hipMemcpy(g->d_scan_epoch, &one, sizeof(one), hipMemcpyHostToDevice));
    }
    YA_TRY(hipMalloc(&g->d_status, sizeof(int)));
    YA_TRY(hipMemset(g->d_count, 0, cb));
    YA_TRY(hipMemset(g->d_offs, 0, cb));
    YA_TRY(hipMemset(g->d_cube_start, 0xff, cb));
    YA_TRY(hipMemset(g->d_cube_end, 0xff, cb));
    YA_TRY(hipMemset(g->d_status, 0, sizeof(int)));
    return 0;
}

int ya_grid_create(int n_max, int grid_size, ya_grid** out)
{
    // cube ids are the reference's binary32 expression (solvers.cuh:357-360, cube_id_of
    // above): exact only while grid_size^3 <= 2^24.  Beyond 256 neighbouring cubes would
    // share an id and the stencil miss real neighbours, silently; refuse instead.
    if (!out || n_max < 0 || grid_size < 1) return (int)hipErrorInvalidValue;
    if (grid_size > YA_MAX_GRID_SIZE) {
        fprintf(stderr,
            "yalla-hip: grid_size %d > %d: cube ids are computed in binary32 like the reference's "
            "(solvers.cuh:357-360) and are exact only up to %d^3 cubes; use a larger cube_size or "
            "fewer cubes\n",
            grid_size, YA_MAX_GRID_SIZE, YA_MAX_GRID_SIZE);
        return (int)hipErrorInvalidValue;
    }
    ya_grid* g = (ya_grid*)calloc(1, sizeof(ya_grid));
    if (!g) return (int)hipErrorOutOfMemory;
    g->n_max = n_max;
    g->grid_size = grid_size;
    g->n_cubes = grid_size * grid_size * grid_size;
    g->n_tiles = ceil_div(g->n_cubes, SCAN_TILE);
    g->padded = (size_t)g->n_tiles * SCAN_TILE;
    g->range_first_tile = -1;  // no cube range: every build scans every tile
    g->range_end_tile = g->n_tiles;
    g->range_lo = 0;
    g->range_hi = g->n_cubes;
    g->range_n = -1;
    const int rc = grid_allocate(g, n_max);
    if (rc) {  // nothing half-built is handed out or leaked
        ya_grid_destroy(g);
        return rc;
    }
    *out = g;
    return 0;
}

int ya_grid_destroy(ya_grid* g)
{
    if (!g) return 0;
    (void)hipFree(g->d_cube_id);
    (void)hipFree(g->d_point_id);
    (void)hipFree(g->d_cube_of);
    (void)hipFree(g->d_rank);
    (void)hipFree(g->d_arrival);
    (void)hipFree(g->d_arrival_src);
    (void)hipFree(g->d_stash);
    (void)hipFree(g->d_prev_pid);
    (void)hipFree(g->d_cube_start);
    (void)hipFree(g->d_cube_end);
    (void)hipFree(g->d_offs);
    (void)hipFree(g->d_count);
    (void)hipFree(g->d_tile_state);
    (void)hipFree(g->d_scan_epoch);
    (void)hipFree(g->d_status);
    free(g);
    return 0;
}

int ya_grid_arrays(ya_grid* g, int** cube_id, int** point_id, int** cube_start, int** cube_end)
{
    if (!g) return (int)hipErrorInvalidValue;
    if (cube_id) *cube_id = g->d_cube_id;
    if (point_id) *point_id = g->d_point_id;
    if (cube_start) *cube_start = g->d_cube_start;
    if (cube_end) *cube_end = g->d_cube_end;
    return 0;
}

int ya_grid_offsets(ya_grid* g, const int** d_offs)
{
    if (!g || !d_offs) return (int)hipErrorInvalidValue;
    *d_offs = g->d_offs;
    return 0;
}

// The prefix sum over the cubes: every scan tile, or -- ya_grid_set_cube_range, and the count is the
// one the last full scan saw -- the tiles of the range alone.
static void launch_scan(ya_grid* g, int n, const int* d_n, hipStream_t st)
{
    const bool range_only = g->range_n >= 0 && g->range_n == n && !d_n;
    const int first = range_only ? g->range_first_tile : 0;
    const int tiles = range_only ? g->range_end_tile - g->range_first_tile : g->n_tiles;
    k_scan<<<tiles, BLOCK, 0, st>>>(g->d_count, g->d_tile_state, g->d_scan_epoch, g->n_cubes, n, g->d_offs, g->d_cube_start,
        g->d_cube_end, d_n, first, g->n_tiles, g->d_status);
    if (!range_only && g->range_first_tile >= 0) g->range_n = d_n ? -1 : n;
}

// First half of a build: binning, scan and scatter.  With d_n != nullptr the count is
// read on the device (n_bound only sizes the launches), so these kernels can be queued
// before the host knows n.
static int build_begin(ya_grid* g, const void* d_X, size_t stride_bytes, const int* d_n,
    int n_bound, float cube_size, bool with_stash, hipStream_t st, ya_n_reader* reader = nullptr)
{
    const int stride_f = (int)(stride_bytes / 4);
    // the previous order is worth visiting unless the population collapsed (judged by
    // the bound when the count itself is not known yet)
    const int n_prev = g->n_prev <= 2 * (long)n_bound ? g->n_prev : 0;
    const int n_visit = n_bound > n_prev ? n_bound : n_prev;
    const int nb_visit = ceil_div(n_visit, BLOCK);
    float* stash = nullptr;
    if (with_stash && n_bound > 0) {  // the stash grows with the first use (and with a wider point)
        const size_t need = (size_t)g->n_max * stride_bytes;
        if (g->stash_bytes < need) {
            (void)hipFree(g->d_stash);
            g->d_stash = nullptr;
            g->stash_bytes = 0;
            YA_TRY(hipMalloc(&g->d_stash, need));
            g->stash_bytes = need;
        }
        stash = g->d_stash;
    }
    if (reader) {
        if (n_bound > 0) {  // the count travels in the binning kernel itself
            reader->seq = reader->seq == 0x7fffffff ? 1 : reader->seq + 1;
            reader->published = true;
            reader->stream = st;
        } else {
            const int rc = ya_n_read_begin(reader, d_n, st);
            if (rc) return rc;
            reader = nullptr;
        }
    }
    if (n_bound > 0)
        k_bin<<<nb_visit, BLOCK, 0, st>>>((const float*)d_X, stride_f, n_bound, cube_size,
            g->grid_size, g->n_cubes, g->d_prev_pid, n_prev, g->d_cube_of, g->d_rank, g->d_count,
            g->d_status, stash, stride_f, d_n, g->range_lo, g->range_hi, reader ? reader->d_view : nullptr,
            reader ? reader->seq : 0);
    launch_scan(g, n_bound, d_n, st);
    if (n_bound > 0)
        k_scatter<<<nb_visit, BLOCK, 0, st>>>(g->d_cube_of, g->d_rank, g->d_offs, n_bound,
            g->d_prev_pid, n_prev, g->d_arrival, g->d_cube_id, stash ? g->d_arrival_src : nullptr,
            nullptr, 0, 0, d_n);
    g->begun_stash = stash;
    return (int)hipGetLastError();
}

// Second half: ascending ids inside each cube and (optionally) the sorted copies.
static int build_finish(ya_grid* g, const void* d_X, size_t stride_bytes, const void* d_old_v,
    int n, void* d_sorted_X, size_t entry_bytes, void* d_sorted_v, hipStream_t st)
{
    const int stride_f = (int)(stride_bytes / 4);
    const int nb = ceil_div(n, BLOCK);
    const bool gather = d_sorted_X != nullptr;
    const float* stash = g->begun_stash;
    if (n > 0) {
        const int entry_f = (int)(entry_bytes / 4);
#define YA_ORDER(NW)                                                                     \
    case NW:                                                                             \
        k_order<NW><<<nb, BLOCK, 0, st>>>(g->d_arrival, g->d_cube_id, g->d_offs, n,      \
            g->d_point_id, g->d_prev_pid, (const float*)d_X, stride_f, (const float*)d_old_v, \
            (float*)d_sorted_X, entry_f, (float4*)d_sorted_v, stash, g->d_arrival_src);  \
        break;
        switch (gather ? stride_f : 0) {
            YA_ORDER(0)
            YA_ORDER(3)
            YA_ORDER(4)
            YA_ORDER(5)
            YA_ORDER(6)
            YA_ORDER(7)
            YA_ORDER(8)
            YA_ORDER(9)
            YA_ORDER(10)
            YA_ORDER(11)
            YA_ORDER(12)
            YA_ORDER(13)
            YA_ORDER(14)
            YA_ORDER(15)
            YA_ORDER(16)
            default:
                return (int)hipErrorInvalidValue;  // points of > 16 floats: not built
        }
#undef YA_ORDER
    }
    g->n_prev = n;
    return (int)hipGetLastError();
}

static bool sorted_args_ok(ya_grid* g, int n, size_t stride_bytes, const void* d_old_v,
    const void* d_sorted_X, size_t entry_bytes, const void* d_sorted_v)
{
    if (!g || n < 0 || n > g->n_max || stride_bytes < 12 || stride_bytes % 4) return false;
    if (d_sorted_X && (entry_bytes < stride_bytes + 4 || entry_bytes % 4 || !d_sorted_v || !d_old_v))
        return false;
    return true;
}

int ya_grid_build_sorted(ya_grid* g, const void* d_X, size_t stride_bytes,
    const void* d_old_v, int n, float cube_size, void* d_sorted_X, size_t entry_bytes,
    void* d_sorted_v, void* stream)
{
    if (!sorted_args_ok(g, n, stride_bytes, d_old_v, d_sorted_X, entry_bytes, d_sorted_v))
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int rc = build_begin(g, d_X, stride_bytes, nullptr, n, cube_size, d_sorted_X != nullptr, st);
    if (rc) return rc;
    return build_finish(g, d_X, stride_bytes, d_old_v, n, d_sorted_X, entry_bytes, d_sorted_v, st);
}

int ya_grid_build_sorted_begin(ya_grid* g, const void* d_X, size_t stride_bytes, const int* d_n,
    int n_bound, float cube_size, void* stream)
{
    if (!g || !d_n || n_bound < 0 || n_bound > g->n_max || stride_bytes < 12 || stride_bytes % 4)
        return (int)hipErrorInvalidValue;
    return build_begin(g, d_X, stride_bytes, d_n, n_bound, cube_size, true, (hipStream_t)stream);
}

int ya_grid_build_sorted_begin_publish(ya_grid* g, const void* d_X, size_t stride_bytes, const int* d_n,
    int n_bound, float cube_size, ya_n_reader* reader, void* stream)
{
    if (!g || !d_n || !reader || n_bound < 0 || n_bound > g->n_max || stride_bytes < 12 || stride_bytes % 4)
        return (int)hipErrorInvalidValue;
    return build_begin(g, d_X, stride_bytes, d_n, n_bound, cube_size, true, (hipStream_t)stream, reader);
}

int ya_grid_build_sorted_finish(ya_grid* g, const void* d_X, size_t stride_bytes,
    const void* d_old_v, int n, void* d_sorted_X, size_t entry_bytes, void* d_sorted_v, void* stream)
{
    if (!d_sorted_X ||
        !sorted_args_ok(g, n, stride_bytes, d_old_v, d_sorted_X, entry_bytes, d_sorted_v))
        return (int)hipErrorInvalidValue;
    return build_finish(
        g, d_X, stride_bytes, d_old_v, n, d_sorted_X, entry_bytes, d_sorted_v, (hipStream_t)stream);
}

int ya_grid_rebuild_sorted(ya_grid* g, const void* d_prev_sorted, size_t entry_bytes,
    size_t point_bytes, const void* d_prev_sorted_v, int n, float cube_size, void* d_sorted_out,
    void* d_sorted_v_out, void* stream)
{
    if (!g || n < 0 || n > g->n_max || point_bytes < 12 || entry_bytes < point_bytes + 4 ||
        entry_bytes % 4 || point_bytes % 4 || !d_prev_sorted || !d_prev_sorted_v || !d_sorted_out ||
        !d_sorted_v_out)
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int entry_f = (int)(entry_bytes / 4);
    const int nb = ceil_div(n, BLOCK);
    // cells are read in the order they are stored: visit = identity (n_prev = 0)
    if (n > 0)
        k_bin<<<nb, BLOCK, 0, st>>>((const float*)d_prev_sorted, entry_f, n, cube_size, g->grid_size,
            g->n_cubes, g->d_prev_pid, 0, g->d_cube_of, g->d_rank, g->d_count, g->d_status,
            nullptr, 0, nullptr, g->range_lo, g->range_hi);
    launch_scan(g, n, nullptr, st);
    if (n > 0) {
        const int id_word = (int)(point_bytes / 4);
        k_scatter<<<nb, BLOCK, 0, st>>>(g->d_cube_of, g->d_rank, g->d_offs, n, g->d_prev_pid, 0,
            g->d_arrival, g->d_cube_id, g->d_arrival_src, (const unsigned*)d_prev_sorted, entry_f,
            id_word, nullptr);
#define YA_ORDER_FROM(EW)                                                                    \
    case EW:                                                                                 \
        k_order_from<EW><<<nb, BLOCK, 0, st>>>(g->d_arrival, g->d_arrival_src, g->d_cube_id, \
            g->d_offs, n, (const unsigned*)d_prev_sorted, (const float4*)d_prev_sorted_v,    \
            g->d_point_id, g->d_prev_pid, (unsigned*)d_sorted_out, (float4*)d_sorted_v_out); \
        break;
        switch (entry_f) {
            YA_ORDER_FROM(4)
            YA_ORDER_FROM(5)
            YA_ORDER_FROM(6)
            YA_ORDER_FROM(7)
            YA_ORDER_FROM(8)
            YA_ORDER_FROM(9)
            YA_ORDER_FROM(10)
            YA_ORDER_FROM(11)
            YA_ORDER_FROM(12)
            YA_ORDER_FROM(13)
            YA_ORDER_FROM(14)
            YA_ORDER_FROM(15)
            YA_ORDER_FROM(16)
            YA_ORDER_FROM(17)
            YA_ORDER_FROM(18)
            YA_ORDER_FROM(19)
            YA_ORDER_FROM(20)
            default:
                return (int)hipErrorInvalidValue;
        }
#undef YA_ORDER_FROM
    }
    g->n_prev = n;
    return (int)hipGetLastError();
}

int ya_grid_build(ya_grid* g, const void* d_X, size_t stride_bytes, int n, float cube_size,
    void* stream)
{
    return ya_grid_build_sorted(
        g, d_X, stride_bytes, nullptr, n, cube_size, nullptr, 0, nullptr, stream);
}

int ya_grid_forget_order(ya_grid* g)
{
    if (!g) return (int)hipErrorInvalidValue;
    g->n_prev = 0;  // visit = identity (k_bin, visit())
    return 0;
}

int ya_grid_set_cube_range(ya_grid* g, int cube_lo, int cube_hi)
{
    if (!g || cube_lo > cube_hi) return (int)hipErrorInvalidValue;
    if (cube_lo <= 0 && cube_hi >= g->n_cubes) {  // the whole grid: as created
        g->range_first_tile = -1;
        g->range_end_tile = g->n_tiles;
        g->range_lo = 0;
        g->range_hi = g->n_cubes;
        g->range_n = -1;
        return 0;
    }
    // whole scan tiles, outwards
    int first = (cube_lo < 0 ? 0 : cube_lo) / SCAN_TILE;
    int end = ceil_div(cube_hi > g->n_cubes ? g->n_cubes : cube_hi, SCAN_TILE);
    if (end > g->n_tiles) end = g->n_tiles;
    if (first >= end) first = end > 0 ? end - 1 : 0;
    if (first == g->range_first_tile && end == g->range_end_tile) return 0;  // unchanged: offs[] stay valid
    g->range_first_tile = first;
    g->range_end_tile = end;
    g->range_lo = first * SCAN_TILE;
    g->range_hi = end * SCAN_TILE < g->n_cubes ? end * SCAN_TILE : g->n_cubes;
    g->range_n = -1;  // the next build scans every tile once
    return 0;
}

int ya_grid_status(ya_grid* g, int* bits, int clear)
{
    if (!g || !bits) return (int)hipErrorInvalidValue;
    YA_TRY(hipMemcpy(bits, g->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (clear && *bits) YA_TRY(hipMemset(g->d_status, 0, sizeof(int)));
    return 0;
}

size_t ya_select_workspace_bytes(int n_max)
{
    return (size_t)(ceil_div(n_max > 0 ? n_max : 1, SEL_TILE) + 1) * sizeof(int);
}

int ya_select_z(const void* d_X, size_t stride_bytes, int n, float z_min, float z_max, int* d_idx,
    int* d_count, int* d_ws, void* stream)
{
    if (n < 0 || stride_bytes < 12 || stride_bytes % 4) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return (int)hipMemsetAsync(d_count, 0, sizeof(int), st);
    const int tiles = ceil_div(n, SEL_TILE);
    const int stride_f = (int)(stride_bytes / 4);
    k_select_count<<<tiles, BLOCK, 0, st>>>((const float*)d_X, stride_f, n, z_min, z_max, d_ws);
    k_select_write<<<tiles, BLOCK, 0, st>>>(
        (const float*)d_X, stride_f, n, z_min, z_max, d_ws, d_idx, d_count);
    return (int)hipGetLastError();
}

int ya_gather_rows(const void* d_src, size_t row_bytes, const int* d_idx, const int* d_count,
    int cap, void* d_dst, void* stream)
{
    if (row_bytes == 0 || row_bytes % 4 || cap < 0) return (int)hipErrorInvalidValue;
    if (cap == 0) return 0;
    const int row_f = (int)(row_bytes / 4);
    long floats = (long)cap * row_f;
    int blocks = (int)((floats + BLOCK - 1) / BLOCK);
    if (blocks > 4096) blocks = 4096;
    k_gather_rows<<<blocks, BLOCK, 0, (hipStream_t)stream>>>(
        (const float*)d_src, row_f, d_idx, d_count, cap, (float*)d_dst);
    return (int)hipGetLastError();
}

int ya_gather_rows_pair(const void* d_src, size_t row_bytes, const int* d_idx0, const int* d_count0,
    void* d_dst0, const int* d_idx1, const int* d_count1, void* d_dst1, int cap, void* stream)
{
    if (!d_src || row_bytes < 4 || row_bytes % 4 || cap < 0 || (d_idx0 && (!d_count0 || !d_dst0)) ||
        (d_idx1 && (!d_count1 || !d_dst1)))
        return (int)hipErrorInvalidValue;
    if (cap == 0 || (!d_idx0 && !d_idx1)) return 0;
    const int row_f = (int)(row_bytes / 4);
    long floats = ((d_idx0 ? 1L : 0L) + (d_idx1 ? 1L : 0L)) * cap * row_f;
    int blocks = (int)((floats + BLOCK - 1) / BLOCK);
    if (blocks > 4096) blocks = 4096;
    k_gather_rows_pair<<<blocks, BLOCK, 0, (hipStream_t)stream>>>((const float*)d_src, row_f, d_idx0, d_count0,
        (float*)d_dst0, d_idx1, d_count1, (float*)d_dst1, cap);
    return (int)hipGetLastError();
}

int ya_append_rows(void* d_dst, size_t row_bytes, int n_own, const void* d_src_lo,
    const int* d_count_lo, const void* d_src_hi, const int* d_count_hi, int cap, int* d_n_out,
    void* stream)
{
    if (!d_dst || row_bytes < 4 || row_bytes % 4 || n_own < 0 || cap < 0 ||
        (d_count_lo && !d_src_lo) || (d_count_hi && !d_src_hi))
        return (int)hipErrorInvalidValue;
    const int row_f = (int)(row_bytes / 4);
    long floats = 2L * cap * row_f;
    int blocks = (int)((floats + BLOCK - 1) / BLOCK);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    k_append_rows<<<blocks, BLOCK, 0, (hipStream_t)stream>>>((float*)d_dst, row_f, n_own,
        (const float*)d_src_lo, d_count_lo, (const float*)d_src_hi, d_count_hi, cap, d_n_out);
    return (int)hipGetLastError();
}

size_t ya_reduce_workspace_bytes(int n_floats)
{
    return (size_t)REDUCE_MAX_BLOCKS * (size_t)n_floats * sizeof(float);
}

static int reduce_any(const void* d_v, int n_floats, int n, float* d_out, float* d_ws, void* stream, bool packed)
{
    hipStream_t st = (hipStream_t)stream;
    const float* v = (const float*)d_v;
    switch (n_floats) {
#define YA_RED(NW) \
    case NW:       \
        return launch_reduce<NW>(v, n, d_out, d_ws, st, packed);
        YA_RED(3)
        YA_RED(4)
        YA_RED(5)
        YA_RED(6)
        YA_RED(7)
        YA_RED(8)
        YA_RED(9)
        YA_RED(10)
        YA_RED(11)
        YA_RED(12)
        YA_RED(13)
        YA_RED(14)
        YA_RED(15)
        YA_RED(16)
#undef YA_RED
        default:
            return (int)hipErrorInvalidValue;
    }
}

int ya_reduce_partials(const void* d_v, int n_floats, int n, float* d_ws, int* n_partials, void* stream)
{
    if (!d_v || !d_ws || !n_partials || n < 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const float* v = (const float*)d_v;
    switch (n_floats) {
#define YA_RED(NW)                                      \
    case NW:                                            \
        *n_partials = launch_partials<NW>(v, n, d_ws, st); \
        break;
        YA_RED(3)
        YA_RED(4)
        YA_RED(5)
        YA_RED(6)
        YA_RED(7)
        YA_RED(8)
        YA_RED(9)
        YA_RED(10)
        YA_RED(11)
        YA_RED(12)
        YA_RED(13)
        YA_RED(14)
        YA_RED(15)
        YA_RED(16)
#undef YA_RED
        default:
            return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

int ya_reduce_mean(const void* d_v, int n_floats, int n, float* d_out, float* d_ws, void* stream)
{
    return reduce_any(d_v, n_floats, n, d_out, d_ws, stream, false);
}

int ya_reduce_sum_packed(const void* d_v, int n_floats, int n, float* d_out, float* d_ws, void* stream)
{
    return reduce_any(d_v, n_floats, n, d_out, d_ws, stream, true);
}


int ya_slab_pack(const void* d_v, int n_floats, int n, float* d_out, float* d_ws, float* d_moved_partial,
    int n_moved, float* d_pred_partial, int n_pred, float limit, float lag_steps, float* d_guard_state,
    int fold_guard, int with_votes, int host_error, const int* d_fix_index, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    const float* v = (const float*)d_v;
    if (fold_guard && d_guard_state && ((n_moved && !d_moved_partial) || (n_pred && !d_pred_partial) || n_moved < 0 || n_pred < 0))
        return (int)hipErrorInvalidValue;
    const Pack_extra x{Guard_args{d_moved_partial, n_moved, d_pred_partial, n_pred, limit, lag_steps, d_guard_state},
        fold_guard, with_votes, host_error, d_fix_index};
    switch (n_floats) {
#define YA_PACK(NW) \
    case NW:        \
        return launch_pack<NW>(v, n, d_out, d_ws, st, x);
        YA_PACK(3)
        YA_PACK(4)
        YA_PACK(5)
        YA_PACK(6)
        YA_PACK(7)
        YA_PACK(8)
        YA_PACK(9)
        YA_PACK(10)
        YA_PACK(11)
        YA_PACK(12)
        YA_PACK(13)
        YA_PACK(14)
        YA_PACK(15)
        YA_PACK(16)
#undef YA_PACK
        default:
            return (int)hipErrorInvalidValue;
    }
}

static bool fields_ok(const size_t row_bytes[3])
{
    for (int k = 0; k < 3; k++)
        if (row_bytes[k] % 4) return false;
    return row_bytes[0] + row_bytes[1] + row_bytes[2] > 0;
}
static int blocks_for(long floats)
{
    long b = (floats + BLOCK - 1) / BLOCK;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int ya_pack_cells(const void* const d_src[3], const size_t row_bytes[3], const int* d_idx, const int* d_count, int cap,
    void* d_message, size_t header_bytes, void* stream)
{
    if (!d_src || !row_bytes || !d_idx || !d_count || !d_message || cap < 0 || header_bytes < 16 || header_bytes % 4 ||
        !fields_ok(row_bytes))
        return (int)hipErrorInvalidValue;
    Fields3 f;
    char* out = (char*)d_message + header_bytes;
    int total_f = 0;
    for (int k = 0; k < 3; k++) {
        f.src[k] = (const float*)d_src[k];
        f.dst[k] = (float*)out;
        f.row_f[k] = (int)(row_bytes[k] / 4);
        if (f.row_f[k] && !d_src[k]) return (int)hipErrorInvalidValue;
        out += (size_t)cap * row_bytes[k];
        total_f += f.row_f[k];
    }
    k_pack_cells<<<blocks_for((long)cap * total_f), BLOCK, 0, (hipStream_t)stream>>>(f, d_idx, d_count, cap, (int*)d_message);
    return (int)hipGetLastError();
}

int ya_append_cells(void* const d_dst[3], const size_t row_bytes[3], int n_own, const void* d_message_lo,
    const void* d_message_hi, int cap, size_t header_bytes, int* d_n_out, int* d_counts_out, void* stream)
{
    if (!d_dst || !row_bytes || n_own < 0 || cap < 0 || header_bytes < 16 || !fields_ok(row_bytes))
        return (int)hipErrorInvalidValue;
    Append3 a;
    size_t offset = header_bytes;
    int total_f = 0;
    for (int k = 0; k < 3; k++) {
        a.dst[k] = (float*)d_dst[k];
        a.row_f[k] = (int)(row_bytes[k] / 4);
        if (a.row_f[k] && !d_dst[k]) return (int)hipErrorInvalidValue;
        a.lo[k] = d_message_lo ? (const float*)((const char*)d_message_lo + offset) : nullptr;
        a.hi[k] = d_message_hi ? (const float*)((const char*)d_message_hi + offset) : nullptr;
        offset += (size_t)cap * row_bytes[k];
        total_f += a.row_f[k];
    }
    k_append_cells<<<blocks_for(2L * cap * total_f), BLOCK, 0, (hipStream_t)stream>>>(
        a, n_own, (const int*)d_message_lo, (const int*)d_message_hi, cap, d_n_out, d_counts_out);
    return (int)hipGetLastError();
}

int ya_fill_holes(void* const d_arrays[3], const size_t row_bytes[3], const int* d_leave_lo, const int* d_count_lo,
    const int* d_leave_hi, const int* d_count_hi, const int* d_movers, const int* d_count_movers, int n_new,
    int max_holes, void* stream)
{
    if (!d_arrays || !row_bytes || !d_movers || !d_count_movers || n_new < 0 || max_holes < 0 || !fields_ok(row_bytes) ||
        (d_leave_lo && !d_count_lo) || (d_leave_hi && !d_count_hi))
        return (int)hipErrorInvalidValue;
    if (max_holes == 0) return 0;
    Fields3 f;
    int total_f = 0;
    for (int k = 0; k < 3; k++) {
        f.src[k] = (const float*)d_arrays[k];
        f.dst[k] = (float*)d_arrays[k];
        f.row_f[k] = (int)(row_bytes[k] / 4);
        if (f.row_f[k] && !d_arrays[k]) return (int)hipErrorInvalidValue;
        total_f += f.row_f[k];
    }
    k_fill_holes<<<blocks_for((long)max_holes * total_f), BLOCK, 0, (hipStream_t)stream>>>(
        f, d_leave_lo, d_count_lo, d_leave_hi, d_count_hi, d_movers, d_count_movers, n_new);
    return (int)hipGetLastError();
}

int ya_copy_component(const void* d_src, size_t stride_bytes, int component, int n, float* d_dst, void* stream)
{
    if (!d_src || !d_dst || n < 0 || stride_bytes < 4 || stride_bytes % 4 || component < 0 ||
        (size_t)component * 4 >= stride_bytes)
        return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    k_copy_component<<<ceil_div(n, BLOCK), BLOCK, 0, (hipStream_t)stream>>>(
        (const float*)d_src + component, (int)(stride_bytes / 4), n, d_dst);
    return (int)hipGetLastError();
}

int ya_find_id(const int* d_ids, int n, int id, int* d_index, void* stream)
{
    if (!d_ids || !d_index || n < 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    YA_TRY(hipMemsetAsync(d_index, 0xff, sizeof(int), st));  // -1: not among these ids
    if (n > 0) k_find_id<<<ceil_div(n, BLOCK), BLOCK, 0, st>>>(d_ids, n, id, d_index);
    return (int)hipGetLastError();
}

int ya_max_abs_diff_partials(int n)
{
    int b = ceil_div(n > 0 ? n : 1, BLOCK);
    return b > REDUCE_MAX_BLOCKS ? REDUCE_MAX_BLOCKS : b;
}

int ya_max_abs_diff(const float* d_a, size_t a_stride_bytes, const float* d_b, size_t b_stride_bytes, int n,
    float lo_face, float hi_face, float width, float* d_partial, void* stream)
{
    if (!d_a || !d_b || !d_partial || n < 0 || a_stride_bytes % 4 || b_stride_bytes % 4 || !a_stride_bytes ||
        !b_stride_bytes)
        return (int)hipErrorInvalidValue;
    k_max_abs_diff<<<ya_max_abs_diff_partials(n), BLOCK, 0, (hipStream_t)stream>>>(
        d_a, (int)(a_stride_bytes / 4), d_b, (int)(b_stride_bytes / 4), n, lo_face, hi_face, width, d_partial);
    return (int)hipGetLastError();
}

int ya_slab_guard_update(float* d_moved_partial, int n_moved, float* d_pred_partial, int n_pred,
    float limit, float lag_steps, float* d_state, void* stream)
{
    if (!d_state || n_moved < 0 || n_pred < 0 || (n_moved && !d_moved_partial) || (n_pred && !d_pred_partial))
        return (int)hipErrorInvalidValue;
    k_slab_guard<<<1, BLOCK, 0, (hipStream_t)stream>>>(
        Guard_args{d_moved_partial, n_moved, d_pred_partial, n_pred, limit, lag_steps, d_state});
    return (int)hipGetLastError();
}

int ya_shader_clock_mhz(double microseconds, double* mhz_out)
{
    if (!mhz_out || !(microseconds > 0) || microseconds > 1e5) return (int)hipErrorInvalidValue;
    // a stream of its own (so that the probe runs BESIDE the work it watches) and a pinned word pair
    // written by the kernel itself, per device: the probe measures the device that is current for the
    // CALLING thread (a sampler thread starts on device 0: hipSetDevice first)
    struct Probe {
        hipStream_t stream = nullptr;
        unsigned long long* h_out = nullptr;
    };
    static std::mutex lock;
    static std::unordered_map<int, Probe> probes;
    int device = 0;
    YA_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> hold(lock);
    Probe& p = probes[device];
    if (!p.stream || !p.h_out) {
        if (!p.stream) YA_TRY(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
        if (!p.h_out) {
            const hipError_t e = hipHostMalloc((void**)&p.h_out, 2 * sizeof(unsigned long long), hipHostMallocDefault);
            if (e != hipSuccess) {  // nothing half-made is kept for the next call
                (void)hipStreamDestroy(p.stream);
                p = Probe{};
                return (int)e;
            }
        }
    }
    k_shader_clock<<<1, 64, 0, p.stream>>>((unsigned long long)(microseconds * 100.0), p.h_out);
    YA_TRY(hipStreamSynchronize(p.stream));
    *mhz_out = p.h_out[1] ? 100.0 * (double)p.h_out[0] / (double)p.h_out[1] : 0.0;
    return 0;
}

// A few bytes read back without stalling the stream they are produced on: queued behind the
// producer, collected later (ya_n_reader for any small record).
struct ya_async_read {
    void* h;  // pinned
    size_t bytes;
    hipEvent_t done;
    int pending;
};
int ya_async_read_create(size_t bytes, ya_async_read** out)
{
    if (!out || bytes == 0 || bytes > 4096) return (int)hipErrorInvalidValue;
    ya_async_read* r = (ya_async_read*)calloc(1, sizeof(ya_async_read));
    if (!r) return (int)hipErrorOutOfMemory;
    r->bytes = bytes;
    hipError_t e = hipHostMalloc(&r->h, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (r->h) (void)hipHostFree(r->h);
        free(r);
        return (int)e;
    }
    *out = r;
    return 0;
}
int ya_async_read_destroy(ya_async_read* r)
{
    if (!r) return 0;
    (void)hipHostFree(r->h);
    (void)hipEventDestroy(r->done);
    free(r);
    return 0;
}
int ya_async_read_begin(ya_async_read* r, const void* d_src, void* stream)
{
    if (!r || !d_src) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    YA_TRY(hipMemcpyAsync(r->h, d_src, r->bytes, hipMemcpyDeviceToHost, st));
    YA_TRY(hipEventRecord(r->done, st));
    r->pending = 1;
    return 0;
}
// The same record written by a kernel itself (the pinned buffer is device-visible at its own address):
// _target is where the kernel stores it, _mark notes "whatever is in `stream` now has written it".
void* ya_async_read_target(ya_async_read* r) { return r ? r->h : nullptr; }
int ya_async_read_mark(ya_async_read* r, void* stream)
{
    if (!r) return (int)hipErrorInvalidValue;
    YA_TRY(hipEventRecord(r->done, (hipStream_t)stream));
    r->pending = 1;
    return 0;
}
int ya_async_read_end(ya_async_read* r, void* h_out)
{
    if (!r || !h_out || !r->pending) return (int)hipErrorInvalidValue;
    YA_TRY(hipEventSynchronize(r->done));
    memcpy(h_out, r->h, r->bytes);
    r->pending = 0;
    return 0;
}

}  // extern "C"
