// EXPERIMENT (round 5, tools/micro/force_ab.hip variant 6): a tile's force evaluation as TWO independent
// one-wavefront workgroups -- plane dz = 0 | planes dz = -1, +1 -- that meet through memory instead of a
// workgroup barrier.  Each half accumulates its planes' terms in the reference's order into sums that
// start at +0; the half that finishes first leaves its sums {F, sum_v, sum_friction} in a per-tile exchange
// area (device-scope stores) and draws the tile's ticket; the half that draws the second ticket adds the
// other's sums to its own and stores the cell's right-hand side.  Two partial sums commute bit for bit
// (a + b == b + a), so the result does not depend on which half comes second: P[dz=0] + P[dz=-1,+1].
// Twice as many wavefronts that live half as long: the drain of a launch (one wavefront lifetime at falling
// occupancy) is halved without the shared LDS / barrier of a two-wavefront workgroup
// (profiles/r05_force_ab.jsonl: that form was 8-18 % slower).  Only for functors declared YA_STATELESS.
// As a WHOLE launch this is 25 % slower (the dispatcher delivers 128 workgroups per us); as the END of a launch
// of whole tiles it is the shipped kernel's tail (variant 8 below).
#pragma once
#ifndef YA_EXPERIMENTAL_FORCE_HALVES
#error "include/experimental/force_halves.cuh is pulled in by solvers.cuh under -DYA_EXPERIMENTAL_FORCE_HALVES"
#endif

namespace ya {

template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
__global__ __launch_bounds__(bits::BLOCK) void grid_force_halves(const int n,
    const Entry<Pt>* __restrict__ sorted, const float4* __restrict__ sorted_v,
    const int* __restrict__ cube_id, const int* __restrict__ offs, const int gs,
    const int n_cubes, const float cut2, Pt* __restrict__ d_dX, const bool has_gen,
    const int n_active, Pt* __restrict__ d_dX_sorted, float* exchange, int* tickets, const int n_tiles)
{
    constexpr int FB = bits::BLOCK;
    constexpr int CAP = bits::Stage<Pt>::value;
    constexpr int NF = N_floats<Pt>::value;
    constexpr int NC = NF + 4;
    __shared__ __attribute__((aligned(16))) Entry<Pt> sh_e[CAP + 8];
    __shared__ unsigned sh_m[(bits::WORDS + 1) * FB];
    __shared__ float4 sh_v[1];
    bits::Lds_word* const words = (bits::Lds_word*)sh_m + threadIdx.x;

    // blocks b and b + 8 are the two halves of one tile (both on XCD b % 8); 16 consecutive blocks hold 8 tiles
    const int half = (blockIdx.x >> 3) & 1;
    const int compact = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    if (compact >= n_tiles) return;
    const int tile = xcd_contiguous_tile(compact, n_tiles);
    const int s0 = tile * FB;
    const int s = s0 + threadIdx.x;
    bool active = s < n;
    const int c_lo = cube_id[s0];
    const int c_hi = cube_id[min(s0 + FB, n) - 1];

    Pt Xi = ya::zero<Pt>();
    int i = 0, c = c_lo;
    if (active) {
        const Entry<Pt> self = sorted[s];
        Xi = self.X;
        i = self.id;
        c = cube_id[s];
        active = i < n_active;
    }
    Pt F = ya::zero<Pt>();
    float3 sum_v{0.f, 0.f, 0.f};
    float sum_friction = 0;

    int next_lo[3], next_hi[3], next_begin[3], next_end[3];
    const int plane_first = half == 0 ? 0 : 1, plane_end = half == 0 ? 1 : 3;
#pragma unroll 1
    for (int plane = plane_first; plane < plane_end; plane++) {
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
            }
            __syncthreads();
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
                bits::pass<Pt, pw_int, pw_friction, false, false>(sh_e, sh_v, words, sb[0], se[0], sb[1], se[1], sb[2],
                    se[2], shift[0], shift[1], shift[2], sorted_v, Xi, i, cut2, F, sum_v, sum_friction, nullptr);
            } else {
#pragma unroll 1
                for (int r = 0; r < 3; r++) {
                    const int rb = r == 0 ? sb[0] : (r == 1 ? sb[1] : sb[2]);
                    const int re = r == 0 ? se[0] : (r == 1 ? se[1] : se[2]);
                    const int rs = r == 0 ? shift[0] : (r == 1 ? shift[1] : shift[2]);
#pragma unroll 1
                    for (int b = rb; __any(b < re); b += bits::PASS_BITS)
                        bits::pass<Pt, pw_int, pw_friction, false, false>(sh_e, sh_v, words, b, min(re, b + bits::PASS_BITS),
                            0, 0, 0, 0, rs, 0, 0, sorted_v, Xi, i, cut2, F, sum_v, sum_friction, nullptr);
                }
            }
        }
    }
    // ---- the halves meet ----
    float* const mine = exchange + ((size_t)tile * 2 + half) * NC * FB + threadIdx.x;
    float part[NC];
#pragma unroll
    for (int k = 0; k < NF; k++) part[k] = field(F, k);
    part[NF] = sum_v.x, part[NF + 1] = sum_v.y, part[NF + 2] = sum_v.z, part[NF + 3] = sum_friction;
#pragma unroll
    for (int k = 0; k < NC; k++) __hip_atomic_store(mine + k * FB, part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sums are acknowledged at device scope before the ticket is drawn  // s_waitcnt: the stores are acknowledged
    __syncthreads();
    __shared__ int sh_second;
    if (threadIdx.x == 0) sh_second = __hip_atomic_fetch_add(&tickets[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sh_second == 0) return;  // the other half will finish the tile
    if (threadIdx.x == 0) __hip_atomic_store(&tickets[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* const theirs = exchange + ((size_t)tile * 2 + (1 - half)) * NC * FB + threadIdx.x;
#pragma unroll
    for (int k = 0; k < NC; k++) part[k] = part[k] + __hip_atomic_load(theirs + k * FB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < NF; k++) field(F, k) = part[k];
    sum_v = float3{part[NF], part[NF + 1], part[NF + 2]};
    sum_friction = part[NF + 3];
    if (active) {
        const Pt dX = store_rhs(d_dX, i, has_gen, F, sum_v, sum_friction);
        if (d_dX_sorted) d_dX_sorted[s] = dX;
    }
}

// EXPERIMENT (variant 7): PERSISTENT wavefronts that fetch their own work.  `gridDim.x` one-wavefront
// workgroups (as many as the chip holds: no dispatcher in the loop) draw work items from one queue per XCD
// (workgroup b serves XCD b % 8, whose tiles are xcd_contiguous_tile(8 * turn + b % 8)): first the XCD's
// whole tiles, then -- for its last `tail_turns` tiles, the end of the launch -- HALF tiles that meet through
// memory as in grid_force_halves.  A whole tile's wavefront accumulates the two halves' sums separately and
// adds them itself, so every cell's sums are P[dz=0] + P[dz=-1,+1] whoever computed them: the split is a
// scheduling decision without any effect on results.
template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
__global__ __launch_bounds__(bits::BLOCK) void grid_force_persistent(const int n,
    const Entry<Pt>* __restrict__ sorted, const float4* __restrict__ sorted_v,
    const int* __restrict__ cube_id, const int* __restrict__ offs, const int gs,
    const int n_cubes, const float cut2, Pt* __restrict__ d_dX, const bool has_gen,
    const int n_active, Pt* __restrict__ d_dX_sorted, float* exchange, int* tickets, int* queue, int* queue_next,
    const int n_tiles, const int tail_turns)
{
    constexpr int FB = bits::BLOCK;
    constexpr int CAP = bits::Stage<Pt>::value;
    constexpr int NF = N_floats<Pt>::value;
    constexpr int NC = NF + 4;
    __shared__ __attribute__((aligned(16))) Entry<Pt> sh_e[CAP + 8];
    __shared__ unsigned sh_m[(bits::WORDS + 1) * FB];
    __shared__ float4 sh_v[1];
    __shared__ int sh_item, sh_second;
    bits::Lds_word* const words = (bits::Lds_word*)sh_m + threadIdx.x;

    // the XCD this wavefront really runs on: its queue is touched by this XCD's wavefronts only, so the queue's
    // atomics can stay in the XCD's own L2 (device-scope atomics on ONE address are served one after the
    // other past the L2s, ~0.1 us each: 2000 of them per queue and launch made this kernel 50 % slower)
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const int xcd = (int)(xcc_id & 7);
    queue += 32 * xcd;
    if (blockIdx.x < 8 && threadIdx.x == 0) queue_next[32 * (blockIdx.x & 7)] = 0;  // the next launch's queues
    const int turns = (n_tiles - xcd + 7) / 8;           // tiles of this XCD
    const int whole = max(turns - tail_turns, 0);        // ... taken as whole tiles; the rest as halves
    const int items = whole + 2 * (turns - whole);

    if (threadIdx.x == 0) sh_item = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    int item = sh_item;
#pragma unroll 1
    while (item < items) {
        // the next item is asked for now and looked at when this one is done
        int next_item = 0;
        if (threadIdx.x == 0) next_item = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int turn = item < whole ? item : whole + ((item - whole) >> 1);
        const int half = item < whole ? -1 : (item - whole) & 1;
        const int tile = xcd_contiguous_tile(8 * turn + xcd, n_tiles);
        const int s0 = tile * FB;
        const int s = s0 + threadIdx.x;
        bool active = s < n;
        const int c_lo = cube_id[s0];
        const int c_hi = cube_id[min(s0 + FB, n) - 1];
        Pt Xi = ya::zero<Pt>();
        int i = 0, c = c_lo;
        if (active) {
            const Entry<Pt> self = sorted[s];
            Xi = self.X;
            i = self.id;
            c = cube_id[s];
            active = i < n_active;
        }
        Pt F = ya::zero<Pt>();
        float3 sum_v{0.f, 0.f, 0.f};
        float sum_friction = 0;
        float first[NC];  // a whole tile: the sums of plane dz = 0 while the other two planes are walked
#pragma unroll
        for (int k = 0; k < NC; k++) first[k] = 0.f;

        int next_lo[3], next_hi[3], next_begin[3], next_end[3];
        const int plane_first = half == 1 ? 1 : 0, plane_end = half == 0 ? 1 : 3;
#pragma unroll 1
        for (int plane = plane_first; plane < plane_end; plane++) {
#ifdef YA_PERSISTENT_SETPRIO
            // a persistent wavefront never becomes "young" again: the issue arbiter (oldest first) would favour
            // the same wavefronts for the whole launch.  Priority by progress through the tile instead.
            if (plane == 0) __builtin_amdgcn_s_setprio(0);
            else if (plane == 1 && half < 0) __builtin_amdgcn_s_setprio(1);
            else if (plane == 1) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
#endif
            if (half < 0 && plane == 1) {
#pragma unroll
                for (int k = 0; k < NF; k++) first[k] = field(F, k);
                first[NF] = sum_v.x, first[NF + 1] = sum_v.y, first[NF + 2] = sum_v.z, first[NF + 3] = sum_friction;
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
                }
                __syncthreads();
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
                    bits::pass<Pt, pw_int, pw_friction, false, false>(sh_e, sh_v, words, sb[0], se[0], sb[1], se[1], sb[2],
                        se[2], shift[0], shift[1], shift[2], sorted_v, Xi, i, cut2, F, sum_v, sum_friction, nullptr);
                } else {
#pragma unroll 1
                    for (int r = 0; r < 3; r++) {
                        const int rb = r == 0 ? sb[0] : (r == 1 ? sb[1] : sb[2]);
                        const int re = r == 0 ? se[0] : (r == 1 ? se[1] : se[2]);
                        const int rs = r == 0 ? shift[0] : (r == 1 ? shift[1] : shift[2]);
#pragma unroll 1
                        for (int b = rb; __any(b < re); b += bits::PASS_BITS)
                            bits::pass<Pt, pw_int, pw_friction, false, false>(sh_e, sh_v, words, b, min(re, b + bits::PASS_BITS),
                                0, 0, 0, 0, rs, 0, 0, sorted_v, Xi, i, cut2, F, sum_v, sum_friction, nullptr);
                    }
                }
            }
        }
        float part[NC];
#pragma unroll
        for (int k = 0; k < NF; k++) part[k] = field(F, k);
        part[NF] = sum_v.x, part[NF + 1] = sum_v.y, part[NF + 2] = sum_v.z, part[NF + 3] = sum_friction;
        bool store = true;
        if (half < 0) {
#pragma unroll
            for (int k = 0; k < NC; k++) part[k] = first[k] + part[k];  // P[dz=0] + P[dz=-1,+1]
        } else {
            float* const mine = exchange + ((size_t)tile * 2 + half) * NC * FB + threadIdx.x;
#pragma unroll
            for (int k = 0; k < NC; k++) __hip_atomic_store(mine + k * FB, part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sums are acknowledged at device scope before the ticket is drawn
            __syncthreads();
            if (threadIdx.x == 0) sh_second = __hip_atomic_fetch_add(&tickets[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            store = sh_second != 0;
            if (store) {
                if (threadIdx.x == 0) __hip_atomic_store(&tickets[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float* const theirs = exchange + ((size_t)tile * 2 + (1 - half)) * NC * FB + threadIdx.x;
#pragma unroll
                for (int k = 0; k < NC; k++)
                    part[k] = part[k] + __hip_atomic_load(theirs + k * FB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (store && active) {
#pragma unroll
            for (int k = 0; k < NF; k++) field(F, k) = part[k];
            const Pt dX = store_rhs(d_dX, i, has_gen, F, float3{part[NF], part[NF + 1], part[NF + 2]}, part[NF + 3]);
            if (d_dX_sorted) d_dX_sorted[s] = dX;
        }
        __syncthreads();
        if (threadIdx.x == 0) sh_item = next_item;
        __syncthreads();
        item = sh_item;
    }
}

// EXPERIMENT (variant 8) -- THE FORM THAT WORKED: grid_force_bits (include/solvers.cuh, "the tail") now does this
// itself; kept as the kernel the A/B record (profiles/r05_force_ab.jsonl, boxes r05_mixed*) was taken with.
// ONE hardware-dispatched launch whose first `n_whole` workgroups (a multiple of 8) are
// whole tiles and whose LAST workgroups are half tiles that meet through memory as in grid_force_halves: only
// the end of the launch -- the drain, one wavefront lifetime at falling occupancy -- is made of wavefronts that
// live half as long; the dispatcher's rate (128 workgroups per us) is no limit for the few of them.  A whole
// tile adds P[dz=0] + P[dz=-1,+1] itself, so every cell's sums are the same whoever computed them.
template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
__global__ __launch_bounds__(bits::BLOCK) void grid_force_mixed(const int n,
    const Entry<Pt>* __restrict__ sorted, const float4* __restrict__ sorted_v,
    const int* __restrict__ cube_id, const int* __restrict__ offs, const int gs,
    const int n_cubes, const float cut2, Pt* __restrict__ d_dX, const bool has_gen,
    const int n_active, Pt* __restrict__ d_dX_sorted, float* exchange, int* tickets, const int n_tiles, const int n_whole)
{
    constexpr int FB = bits::BLOCK;
    constexpr int CAP = bits::Stage<Pt>::value;
    constexpr int NF = N_floats<Pt>::value;
    constexpr int NC = NF + 4;
    __shared__ __attribute__((aligned(16))) Entry<Pt> sh_e[CAP + 8];
    __shared__ unsigned sh_m[(bits::WORDS + 1) * FB];
    __shared__ float4 sh_v[1];
    __shared__ int sh_second;
    bits::Lds_word* const words = (bits::Lds_word*)sh_m + threadIdx.x;

    int half = -1, compact = blockIdx.x;
    if ((int)blockIdx.x >= n_whole) {
        const int b = blockIdx.x - n_whole;
#ifdef YA_MIXED_LONG_FIRST
        // all the longer halves (the cells' own plane: ~63 % of a tile's pairs) first, the shorter ones last
        const int tail8 = (n_tiles - n_whole + 7) & ~7;
        half = b >= tail8;
        compact = n_whole + (half ? b - tail8 : b);
#else
        // blocks b and b + 8 are the halves of one tile, both on XCD b % 8
        half = (b >> 3) & 1;
        compact = n_whole + (b >> 4) * 8 + (b & 7);
#endif
#ifdef YA_MIXED_PRIO
        __builtin_amdgcn_s_setprio(YA_MIXED_PRIO);
#endif
    }
    if (compact >= n_tiles) return;
    const int tile = xcd_contiguous_tile(compact, n_tiles);
    const int s0 = tile * FB;
    const int s = s0 + threadIdx.x;
    bool active = s < n;
    const int c_lo = cube_id[s0];
    const int c_hi = cube_id[min(s0 + FB, n) - 1];

    Pt Xi = ya::zero<Pt>();
    int i = 0, c = c_lo;
    if (active) {
        const Entry<Pt> self = sorted[s];
        Xi = self.X;
        i = self.id;
        c = cube_id[s];
        active = i < n_active;
    }
    Pt F = ya::zero<Pt>();
    float3 sum_v{0.f, 0.f, 0.f};
    float sum_friction = 0;
    float first[NC];  // a whole tile: the sums of plane dz = 0 while the other two planes are walked
#pragma unroll
    for (int k = 0; k < NC; k++) first[k] = 0.f;

    int next_lo[3], next_hi[3], next_begin[3], next_end[3];
    const int plane_first = half == 1 ? 1 : 0, plane_end = half == 0 ? 1 : 3;
#pragma unroll 1
    for (int plane = plane_first; plane < plane_end; plane++) {
        if (half < 0 && plane == 1) {
#pragma unroll
            for (int k = 0; k < NF; k++) first[k] = field(F, k);
            first[NF] = sum_v.x, first[NF + 1] = sum_v.y, first[NF + 2] = sum_v.z, first[NF + 3] = sum_friction;
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
            }
            __syncthreads();
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
                bits::pass<Pt, pw_int, pw_friction, false, false>(sh_e, sh_v, words, sb[0], se[0], sb[1], se[1], sb[2],
                    se[2], shift[0], shift[1], shift[2], sorted_v, Xi, i, cut2, F, sum_v, sum_friction, nullptr);
            } else {
#pragma unroll 1
                for (int r = 0; r < 3; r++) {
                    const int rb = r == 0 ? sb[0] : (r == 1 ? sb[1] : sb[2]);
                    const int re = r == 0 ? se[0] : (r == 1 ? se[1] : se[2]);
                    const int rs = r == 0 ? shift[0] : (r == 1 ? shift[1] : shift[2]);
#pragma unroll 1
                    for (int b = rb; __any(b < re); b += bits::PASS_BITS)
                        bits::pass<Pt, pw_int, pw_friction, false, false>(sh_e, sh_v, words, b, min(re, b + bits::PASS_BITS),
                            0, 0, 0, 0, rs, 0, 0, sorted_v, Xi, i, cut2, F, sum_v, sum_friction, nullptr);
                }
            }
        }
    }
    float part[NC];
#pragma unroll
    for (int k = 0; k < NF; k++) part[k] = field(F, k);
    part[NF] = sum_v.x, part[NF + 1] = sum_v.y, part[NF + 2] = sum_v.z, part[NF + 3] = sum_friction;
    if (half < 0) {
#pragma unroll
        for (int k = 0; k < NC; k++) part[k] = first[k] + part[k];  // P[dz=0] + P[dz=-1,+1]
    } else {
        float* const mine = exchange + ((size_t)tile * 2 + half) * NC * FB + threadIdx.x;
#pragma unroll
        for (int k = 0; k < NC; k++) __hip_atomic_store(mine + k * FB, part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sums are acknowledged at device scope before the ticket is drawn
        __syncthreads();
        if (threadIdx.x == 0) sh_second = __hip_atomic_fetch_add(&tickets[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (sh_second == 0) return;  // the other half will finish the tile
        if (threadIdx.x == 0) __hip_atomic_store(&tickets[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float* const theirs = exchange + ((size_t)tile * 2 + (1 - half)) * NC * FB + threadIdx.x;
#pragma unroll
        for (int k = 0; k < NC; k++)
            part[k] = part[k] + __hip_atomic_load(theirs + k * FB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < NF; k++) field(F, k) = part[k];
        const Pt dX = store_rhs(d_dX, i, has_gen, F, float3{part[NF], part[NF + 1], part[NF + 2]}, part[NF + 3]);
        if (d_dX_sorted) d_dX_sorted[s] = dX;
    }
}

}  // namespace ya
