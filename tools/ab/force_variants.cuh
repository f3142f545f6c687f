// Earlier grid-force kernels, kept as A/B baselines and as independent statements of the same sums:
//
//   ya::grid_force_direct   the reference's structure -- every cell walks its 27 cubes through L1 / L2
//   ya::grid_force          round 1's LDS-staged two-phase kernel with a BYTE FIFO of hits, 256-thread
//                           workgroups (grid_force_bits, the shipped kernel, keeps one BIT per candidate
//                           in one-wavefront workgroups: 242 against 260 us at 1 M cells)
//
// Both give grid_force_bits' results bit for bit (tests/test_parity_gpu.py::test_all_force_kernels_agree,
// the fuzz generator).  NOT part of the product path: include/solvers.cuh pulls this file in, and
// Grid_computer::force_variant 0 / 1 selects the kernels, only in translation units compiled with
// -DYA_EXPERIMENTAL_FORCE_VARIANTS (the model harness the tests and bench.py drive,
// tools/micro/force_ab.hip); a model's own translation unit neither compiles nor can select them
// (round 4: they were in solvers.cuh itself, instantiated for every functor of every model).
#pragma once
#ifndef YA_EXPERIMENTAL_FORCE_VARIANTS
#error "tools/ab/force_variants.cuh is pulled in by solvers.cuh under -DYA_EXPERIMENTAL_FORCE_VARIANTS"
#endif

namespace ya {

// grid_force_direct is the plain form (neighbours read straight from the
// sorted array through L1/L2); it is kept as the A/B baseline for grid_force.
template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
__global__ __launch_bounds__(FORCE_BLOCK) void grid_force_direct(const int n,
    const Entry<Pt>* __restrict__ sorted, const float4* __restrict__ sorted_v,
    const int* __restrict__ cube_id, const int* __restrict__ offs, const int gs,
    const int n_cubes, const float cube_size, Pt* __restrict__ d_dX, const bool has_gen,
    const int n_active, const int* __restrict__ global_id, const bool by_plane)
{
    const int s = blockIdx.x * FORCE_BLOCK + threadIdx.x;
    if (s >= n) return;

    const Entry<Pt> self = sorted[s];
    const Pt Xi = self.X;
    const int i = self.id;
    if (i >= n_active) return;  // ghost cell of a slab decomposition: no force needed
    // functors see GLOBAL ids in a slab decomposition (they index per-cell model arrays)
    const int gi = global_id ? global_id[i] : i;
    const int c = cube_id[s];
    Pt F = ya::zero<Pt>();
    float3 sum_v{0.f, 0.f, 0.f};
    float sum_friction = 0;
    // by_plane (Grid_computer::sum_order = YA_SUM_BY_PLANE): the own plane's sums (rows 0-2) kept aside, the
    // other planes' summed from +0, the two added at the end.  Otherwise the reference's one running sum:
    // the _own sums stay +0, and +0 + x == x for every x a sum that started at +0 can hold (never -0).
    Pt F_own = ya::zero<Pt>();
    float3 sum_v_own{0.f, 0.f, 0.f};
    float sum_friction_own = 0;
    for (int row = 0; row < 9; row++) {
        if (row == 3 && by_plane) {
            F_own = F, sum_v_own = sum_v, sum_friction_own = sum_friction;
            F = ya::zero<Pt>(), sum_v = float3{0.f, 0.f, 0.f}, sum_friction = 0;
        }
        const int mid = c + stencil_row_offset(row, gs);
        // The reference indexes cube_start/end without bounds checks
        // (solvers.cuh:444); out-of-grid cubes are treated as empty here.
        const int first = min(max(mid - 1, 0), n_cubes);
        const int last = min(max(mid + 2, 0), n_cubes);
        const int k_end = offs[last];
        for (int k = offs[first]; k < k_end; k++) {
            const Entry<Pt> other = sorted[k];
            Pt r = Xi - other.X;
            float dist = dist3(r.x, r.y, r.z);
            if (dist >= cube_size) continue;

            const int j = global_id ? global_id[other.id] : other.id;
            F += pw_int(Xi, r, dist, gi, j);
            float friction = pw_friction(Xi, r, dist, gi, j);
            sum_friction += friction;
            if (friction != 0) {
                float4 v = sorted_v[k];
                sum_v.x += friction * v.x;
                sum_v.y += friction * v.y;
                sum_v.z += friction * v.z;
            }
        }
    }
    store_rhs(d_dX, i, has_gen, F_own + F,
        float3{sum_v_own.x + sum_v.x, sum_v_own.y + sum_v.y, sum_v_own.z + sum_v.z}, sum_friction_own + sum_friction);
}

// Cells staged in LDS at a time (16 B per float3 cell) and the per-thread
// hit-queue depth (one byte per queued hit): 928 * 16 B + 44 * 256 B = 26 KiB per
// workgroup, i.e. six workgroups (24 wavefronts) per CU.  Swept on MI355X
// (DESIGN.md §6): workgroup size, staging capacity and queue depth all sit at a
// shallow optimum here.
template<typename Pt>
struct Stage_cells {
#ifndef YA_STAGE_CELLS
#define YA_STAGE_CELLS (3 * YA_FORCE_BLOCK + 160)
#endif
#ifndef YA_STAGE_CELLS_MID
#define YA_STAGE_CELLS_MID YA_STAGE_CELLS  /* 17..32-byte entries: a plane in one chunk beats a sixth workgroup (swept) */
#endif
    static constexpr int value =
        sizeof(Entry<Pt>) <= 16 ? YA_STAGE_CELLS : (sizeof(Entry<Pt>) <= 32 ? YA_STAGE_CELLS_MID : YA_STAGE_CELLS / 2);
};
#ifndef YA_QUEUE_DEPTH
#define YA_QUEUE_DEPTH 44
#endif
constexpr int QUEUE_DEPTH = YA_QUEUE_DEPTH;

// LDS-staged grid force.  A workgroup owns 256 consecutive sorted slots, i.e. a
// run of cubes [c_lo, c_hi] along x.  For stencil row r every neighbour of every
// cell of the workgroup lies in the contiguous slots
// [offs[c_lo + off_r - 1], offs[c_hi + off_r + 2]); each thread's own candidates
// are the sub-range [offs[c + off_r - 1], offs[c + off_r + 2]).  The nine rows are
// handled as three planes (dz = 0, -1, +1: rows 0-2, 3-5, 6-8 of the reference's
// d_nhood order).  Per plane the workgroup copies its three slot ranges into LDS
// (coalesced 16-byte loads of {X, id}), then every thread
//
//   phase 1  walks its candidates in the reference's order testing d2 < cut2
//            only, and appends one byte per hit (~15 % of the 27-cube volume lies
//            inside the cut-off sphere) to a per-thread FIFO in LDS;
//   phase 2  drains the FIFO: distance, functor, friction, old_v term.
//
// Both loops run until the slowest lane of the wavefront is done, so phase 2 is
// kept dense by draining only once per plane (or when a FIFO could overflow):
// a lane's hit count summed over a plane varies far less across the wavefront
// than its hit count within one 32-candidate stretch.  Order is preserved
// (FIFO), so every per-cell sum is accumulated in grid_force_bits' order (own plane | other planes).
template<typename Pt, Pairwise_interaction<Pt> pw_int, Pairwise_friction<Pt> pw_friction>
__global__ __launch_bounds__(FORCE_BLOCK) void grid_force(const int n,
    const Entry<Pt>* __restrict__ sorted, const float4* __restrict__ sorted_v,
    const int* __restrict__ cube_id, const int* __restrict__ offs, const int gs,
    const int n_cubes, const float cut2, Pt* __restrict__ d_dX, const bool has_gen,
    const int n_active, Pt* __restrict__ d_dX_sorted, const int* __restrict__ global_id, const bool by_plane)
{
    constexpr int CAP = Stage_cells<Pt>::value;
    __shared__ __attribute__((aligned(16))) Entry<Pt> sh_e[CAP + 8];  // slack: phase 1 reads whole groups
    // One byte per queued hit: (row of the plane) << 6 | offset of the candidate from
    // the lane's anchor in that row (0..63).
    __shared__ unsigned char sh_q[QUEUE_DEPTH * FORCE_BLOCK];

    // LDS (address space 3) FIFO pointers: 32-bit address arithmetic in the hot loops
    using Lds_byte = __attribute__((address_space(3))) unsigned char;
    Lds_byte* const q_base = (Lds_byte*)sh_q + threadIdx.x * QUEUE_DEPTH;  // this lane's FIFO
#ifndef YA_GROUP
#define YA_GROUP 4
#endif
    Lds_byte* const q_high = q_base + (QUEUE_DEPTH - YA_GROUP);  // "nearly full" mark

    const int s0 = xcd_contiguous_tile(blockIdx.x, gridDim.x) * FORCE_BLOCK;
    const int s = s0 + threadIdx.x;
    bool active = s < n;
    const int c_lo = cube_id[s0];
    const int c_hi = cube_id[min(s0 + FORCE_BLOCK, n) - 1];

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
    const int gi = global_id && active ? global_id[i] : i;  // what functors see (slab mode: global ids)
    Pt F = ya::zero<Pt>();
    float3 sum_v{0.f, 0.f, 0.f};
    float sum_friction = 0;
    Lds_byte* q_tail = q_base;
    asm volatile("" : "+v"(q_tail));

    // LDS index of a staged cell -> its slot in the sorted arrays (set per chunk):
    // old_v of an interacting neighbour is read from global memory (L1/L2 hits, the
    // neighbours of a workgroup are a few contiguous slot ranges) rather than staged,
    // which keeps the workgroup at 40 KiB of LDS = four workgroups per CU.
    int slot_shift0 = 0, slot_shift1 = 0, slot_shift2 = 0;
    int anchor0 = 0, anchor1 = 0, anchor2 = 0;  // LDS index a queued offset is relative to
    int slot0 = 0, slot1 = 0, slot2 = 0;        // the same anchors as slots of the sorted arrays

    int next_lo[3], next_hi[3], next_begin[3], next_end[3];
    Pt F_own = ya::zero<Pt>();  // by_plane only: own plane | other planes (grid_force_bits); else they stay +0
    float3 sum_v_own{0.f, 0.f, 0.f};
    float sum_friction_own = 0;
    YA_ROW_BOUNDS(0)
    for (int plane = 0; plane < 3; plane++) {
        if (plane == 1 && by_plane) {  // (the FIFOs are empty at the end of every plane)
            F_own = F, sum_v_own = sum_v, sum_friction_own = sum_friction;
            F = ya::zero<Pt>(), sum_v = float3{0.f, 0.f, 0.f}, sum_friction = 0;
        }
        // The plane's three rows, concatenated: row r occupies [v0[r], v0[r+1]).
        int wg_begin[3], v0[4], k_begin[3], k_end[3];
        v0[0] = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            wg_begin[r] = next_lo[r];
            v0[r + 1] = v0[r] + next_hi[r] - wg_begin[r];
            k_begin[r] = next_begin[r];
            k_end[r] = active ? next_end[r] : k_begin[r];
        }
        if (plane < 2) { YA_ROW_BOUNDS(plane + 1) }
        const int total = v0[3];

        for (int chunk = 0; chunk < total; chunk += CAP) {
            const int chunk_n = min(CAP, total - chunk);
            __syncthreads();
            for (int t = threadIdx.x; t < chunk_n; t += FORCE_BLOCK) {
                const int v = chunk + t;
                const int shift = v >= v0[2] ? wg_begin[2] - v0[2]
                                             : (v >= v0[1] ? wg_begin[1] - v0[1] : wg_begin[0]);
                sh_e[t] = sorted[v + shift];
            }
            __syncthreads();
            slot_shift0 = wg_begin[0] + chunk;
            slot_shift1 = wg_begin[1] - v0[1] + chunk;
            slot_shift2 = wg_begin[2] - v0[2] + chunk;

            // One wavefront-uniform loop over the plane's rows.  Phase 1: each lane
            // walks its candidates of the current row, four per trip (their LDS reads
            // in flight together), until done or its FIFO is nearly full; when every
            // lane is done the wavefront moves to the next row; phase 2 drains the
            // FIFOs when a lane is full or the plane is finished.
            Lds_byte* const q_last = q_base + (QUEUE_DEPTH - 1);
            int row = 0;
            int t = max(k_begin[0] - wg_begin[0], chunk) - chunk;
            int b = min(k_end[0] - wg_begin[0], chunk + chunk_n) - chunk;
            anchor0 = t;
            slot0 = t + slot_shift0;
            int off = 0;  // t - anchor of the row; a queued byte is (row << 6) | off
            while (true) {
                while (t + YA_GROUP <= b && q_tail <= q_high && off + YA_GROUP <= 64) {
                    float4 w[YA_GROUP];
                    float d2[YA_GROUP];

#pragma unroll
                    for (int u = 0; u < YA_GROUP; u++) w[u] = staged_words(&sh_e[t + u]);
#pragma unroll
                    for (int u = 0; u < YA_GROUP; u++) {
                        d2[u] = dist2_to(Xi, w[u]);
                        keep_wide(w[u]);
                    }
                    // the byte is always written and only kept (tail advanced) on a hit:
                    // no branch, no exec masking
#pragma unroll
                    for (int u = 0; u < YA_GROUP; u++) {
                        *q_tail = (unsigned char)((row << 6) + off + u);
                        q_tail += d2[u] < cut2;
                    }
                    t += YA_GROUP;
                    off += YA_GROUP;
                }
                while (t < b && q_tail <= q_last && off < 64 &&
                       (t + YA_GROUP > b || q_tail > q_high || off + YA_GROUP > 64)) {
                    const float4 w = staged_words(&sh_e[t]);
                    keep_wide(w);
                    *q_tail = (unsigned char)((row << 6) + off);
                    q_tail += dist2_to(Xi, w) < cut2;
                    t++;
                    off++;
                }
                const bool row_done = !__any(t < b);
                if (row_done && row < 2) {
                    row++;
                    const int kb = row == 1 ? k_begin[1] - wg_begin[1] + v0[1]
                                            : k_begin[2] - wg_begin[2] + v0[2];
                    const int ke = row == 1 ? k_end[1] - wg_begin[1] + v0[1]
                                            : k_end[2] - wg_begin[2] + v0[2];
                    t = max(kb, chunk) - chunk;
                    b = min(ke, chunk + chunk_n) - chunk;
                    off = 0;
                    if (row == 1) {
                        anchor1 = t;
                        slot1 = t + slot_shift1;
                    } else {
                        anchor2 = t;
                        slot2 = t + slot_shift2;
                    }
                    continue;
                }
                {  // phase 2: drain this lane's FIFO (no lambda: nothing may have its address
                   // taken here, the byte stores of phase 1 could alias it)
                const int count = (int)(q_tail - q_base);
                q_tail = q_base;
                asm volatile("" : "+v"(q_tail));  // keep the tail an address, not base + count
                int e_next = q_base[0];  // read one hit ahead: one LDS latency per trip, not two
                for (int q = 0; q < count; q++) {
                    const int e = e_next;
                    e_next = q_base[min(q + 1, QUEUE_DEPTH - 1)];
                    const int t = (e & 63) + (e >= 128 ? anchor2 : (e >= 64 ? anchor1 : anchor0));
                    const Entry<Pt> other = sh_e[t];
                    const unsigned slot = (e & 63) + (e >= 128 ? slot2 : (e >= 64 ? slot1 : slot0));
                    const float4 v = sorted_v[slot];
                    Pt r = Xi - other.X;
                    float dist = dist3(r.x, r.y, r.z);
                    const int j = global_id ? global_id[other.id] : other.id;
                    F += pw_int(Xi, r, dist, gi, j);
                    float friction = pw_friction(Xi, r, dist, gi, j);
                    sum_friction += friction;
                    if (friction != 0) {
                        sum_v.x += friction * v.x;
                        sum_v.y += friction * v.y;
                        sum_v.z += friction * v.z;
                    }
                }
                }
                if (row_done) break;
                // FIFOs are empty: lanes still inside this row re-anchor at their position
                off = 0;
                if (row == 0) {
                    anchor0 = t;
                    slot0 = t + slot_shift0;
                } else if (row == 1) {
                    anchor1 = t;
                    slot1 = t + slot_shift1;
                } else {
                    anchor2 = t;
                    slot2 = t + slot_shift2;
                }
            }
        }
    }
    if (active) {
        const Pt dX = store_rhs(d_dX, i, has_gen, F_own + F,
            float3{sum_v_own.x + sum_v.x, sum_v_own.y + sum_v.y, sum_v_own.z + sum_v.z}, sum_friction_own + sum_friction);
        if (d_dX_sorted) d_dX_sorted[s] = dX;  // for the sorted-space Euler stage
    }
}



}  // namespace ya
