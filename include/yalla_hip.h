/* yalla_hip.h -- C ABI of libyalla_hip.so, the Pt-agnostic half of the
 * MI355X-native ya||a step path.
 *
 * The reference (germannp/yalla) is a header-only CUDA template library and
 * has no FFI; its boundary for the hot path is the header API
 * `Solution<Pt, Solver>::take_step<pw_int, pw_friction>(dt, gen_forces)`
 * (include/solvers.cuh:60-106).  This repo keeps that header API
 * (include/solvers.cuh, links.cuh, dtypes.cuh ...) and puts every piece that
 * does not depend on the point type or on the user's functor behind the plain
 * C entry points below: plain pointers and sizes, no C++ or HIP types, every
 * function returns a hipError_t value as int (0 = success).  Each entry point
 * cites the reference interface it replaces.  `stream` is a hipStream_t passed
 * as void* (NULL = the default stream, which is what the reference uses
 * throughout).
 *
 * The functor-carrying kernels (tile force, grid force, link forces) and the
 * Pt-typed Heun updates are templates in the headers and are instantiated in
 * the model's translation unit by hipcc, exactly where the reference
 * instantiates compute_tile / compute_cube / link.
 */
#ifndef YALLA_HIP_H
#define YALLA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The libraries are built with -fvisibility=hidden; only this C ABI is exported. */
#pragma GCC visibility push(default)

#define YA_ABI_VERSION 10  /* 10: + ya_grid_build_sorted_begin_publish; 9: + ya_comm_info, ya_reduce_partials; 8: + ya_grid_set_cube_range, YA_STATUS_OUT_OF_RANGE, the slab guard / fixed point / payload entries, ya_async_read_* */

/* Status bits reported by ya_grid_status(). */
#define YA_STATUS_OUT_OF_GRID 1 /* a cell's cube id fell outside [0, n_cubes):
                                   the reference's D_ASSERT at
                                   solvers.cuh:361-362 */

#define YA_STATUS_OUT_OF_RANGE 2 /* a cell's cube id fell outside the range promised to
                                    ya_grid_set_cube_range */

#define YA_STATUS_SCAN_STALLED 4 /* the one-launch prefix sum over the cubes waited ~0.2 s for a block of lower
                                   index that never published its total: the scan relies on workgroups being
                                   started in index order; one grid per stream at a time */

int ya_abi_version(void);

/* Device memory owned by the solver classes.  Replaces the cudaMalloc /
 * cudaFree pairs of Heun_solver (solvers.cuh:171-195), Grid (:387-403),
 * Links (links.cuh:36-52) and Property (property.cuh:17-25). */
int ya_malloc(void** d_ptr, size_t bytes);
int ya_free(void* d_ptr);
int ya_memset_async(void* d_ptr, int value, size_t bytes, void* stream);
int ya_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int ya_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
/* Page-locked host memory (hipHostMalloc / hipHostFree) for a Solution's host mirror, so that
 * copy_to_host / copy_to_device (reference solvers.cuh:92-104: the whole n_max points per output
 * frame) run at PCIe speed instead of through a staging buffer.  The reference's h_X is pageable
 * malloc memory; models only index it.  (Page-locking malloc'ed memory in place, hipHostRegister,
 * was tried first: after some hundred register / unregister cycles of small arrays a later kernel
 * faulted -- tests/fuzz_parity.py found it.) */
int ya_host_alloc(void** h_ptr, size_t bytes);
int ya_host_free(void* h_ptr);
int ya_memcpy_d2d_async(void* d_dst, const void* d_src, size_t bytes, void* stream);
int ya_device_synchronize(void);

/* Blocking read of the device-side point count.  Replaces
 * Heun_solver::get_d_n (solvers.cuh:219-225) and Links::get_d_n
 * (links.cuh:58-64); the caller asserts n <= n_max as the reference does. */
int ya_get_n(const int* d_n, int* n_out);

/* The same read in two halves, so that work which does not need n on the host can be
 * queued in between: ya_n_read_begin queues the 4-byte copy (into pinned memory) behind
 * everything already in `stream`, ya_n_read_end waits for it. */
typedef struct ya_n_reader ya_n_reader;
int ya_n_reader_create(ya_n_reader** out);
int ya_n_reader_destroy(ya_n_reader* r);
int ya_n_read_begin(ya_n_reader* r, const int* d_n, void* stream);
int ya_n_read_end(ya_n_reader* r, int* n_out);

/* The shader clock in MHz as the chip runs it NOW: one wavefront on a stream of its own counts
 * s_memtime ticks (shader cycles) against wall_clock64 (constant 100 MHz) for `microseconds`,
 * beside whatever else is running.  Measurement aid (bench.py --sustained: MI355X drops from 2.4 to
 * ~1.7 GHz under dense VALU issue, and a throughput figure without the clock it was taken at says
 * little); blocks the calling thread for about that long.  No counterpart in the reference. */
int ya_shader_clock_mhz(double microseconds, double* mhz_out);

/* ---- Uniform grid (spatial hash) -------------------------------------- */

typedef struct ya_grid ya_grid;

/* Replaces Grid::Grid (solvers.cuh:384-395): allocates the four public
 * arrays d_cube_id[n_max], d_point_id[n_max], d_cube_start[n_cubes],
 * d_cube_end[n_cubes] (n_cubes = grid_size^3) plus private scratch.
 * grid_size > YA_MAX_GRID_SIZE is refused (hipErrorInvalidValue and a message):
 * cube ids follow the reference's binary32 expression (solvers.cuh:357-360), which
 * is exact only while grid_size^3 <= 2^24. */
#define YA_MAX_GRID_SIZE 256
int ya_grid_create(int n_max, int grid_size, ya_grid** out);
int ya_grid_destroy(ya_grid* g);

/* Device pointers of the four public arrays (stable for the grid's life). */
int ya_grid_arrays(ya_grid* g, int** d_cube_id, int** d_point_id,
    int** d_cube_start, int** d_cube_end);

/* Device pointer to the private exclusive prefix offs[0 .. n_cubes]:
 * offs[c] = number of cells in cubes < c, offs[n_cubes] = n.  Sorted slots of
 * cube c are [offs[c], offs[c + 1]). */
int ya_grid_offsets(ya_grid* g, const int** d_offs);

/* Replaces Grid::build(n, d_X, cube_size) (solvers.cuh:406-417), i.e.
 * compute_cube_id (:349-365) + 2x thrust::fill + thrust::sort_by_key +
 * compute_cube_start_and_end (:367-378).  d_X is an array of n points of
 * `stride_bytes` each whose first three floats are x, y, z.  Results are
 * bit-identical to the reference's: the cube id is evaluated in binary32 in
 * the reference's association order, cells are ordered by (cube id, point id)
 * (= stable sort by cube id), empty cubes get start -1 / end -2. */
int ya_grid_build(ya_grid* g, const void* d_X, size_t stride_bytes, int n,
    float cube_size, void* stream);

/* ya_grid_build + a gather of the cells into sorted order for the force
 * kernel: d_sorted_X[slot] = { point (stride_bytes), int point_id, padding up
 * to entry_bytes }, d_sorted_v[slot] = { old_v.x, old_v.y, old_v.z, 0 } (16 B).
 * d_old_v is the float3 array of solvers.cuh:65 (12-byte elements). */
int ya_grid_build_sorted(ya_grid* g, const void* d_X, size_t stride_bytes,
    const void* d_old_v, int n, float cube_size, void* d_sorted_X,
    size_t entry_bytes, void* d_sorted_v, void* stream);

/* ya_grid_build_sorted in two halves for a caller that has not read the point count
 * back yet (Heun_solver::take_step reads n at its start, solvers.cuh:229): _begin bins,
 * scans and scatters with the count read from *d_n ON THE DEVICE (n_bound >= n only
 * sizes the launches, n_max will do) and can be queued right after ya_n_read_begin;
 * _finish (ordering inside the cubes, the sorted copies) needs n on the host. */
int ya_grid_build_sorted_begin(ya_grid* g, const void* d_X, size_t stride_bytes, const int* d_n,
    int n_bound, float cube_size, void* stream);
int ya_grid_build_sorted_finish(ya_grid* g, const void* d_X, size_t stride_bytes,
    const void* d_old_v, int n, void* d_sorted_X, size_t entry_bytes, void* d_sorted_v,
    void* stream);
/* _begin that also STARTS the read of the count (instead of ya_n_read_begin in front of it): the binning
 * kernel's first thread stores *d_n into the reader's host memory itself -- no 4-byte copy, which is a 4 us
 * kernel of its own in the stream, ahead of the build (get_d_n, solvers.cuh:219-225, once per take_step).
 * ya_n_read_end(reader) then waits for that store.  n_bound == 0 falls back to the copy. */
int ya_grid_build_sorted_begin_publish(ya_grid* g, const void* d_X, size_t stride_bytes, const int* d_n,
    int n_bound, float cube_size, ya_n_reader* reader, void* stream);

/* The same result as ya_grid_build_sorted, but for cells that already sit in an
 * earlier build's sorted arrays and have moved a little since (the second Heun
 * stage): d_prev_sorted holds n entries {point (point_bytes), int id, padding to
 * entry_bytes} whose positions are the current ones, d_prev_sorted_v their old_v
 * (16 B each).  Nothing is gathered from the original-order arrays; the outputs
 * (distinct buffers) and the four public arrays are exactly what a build from
 * the original-order arrays would give. */
int ya_grid_rebuild_sorted(ya_grid* g, const void* d_prev_sorted, size_t entry_bytes,
    size_t point_bytes, const void* d_prev_sorted_v, int n, float cube_size, void* d_sorted_out,
    void* d_sorted_v_out, void* stream);

/* A build visits the cells in the previous build's sorted order (they move little between builds:
 * neighbouring lanes then bin into neighbouring counters).  After the caller has given the cells
 * new ids (Solution::renumber, include/solvers.cuh) that order means other cells: the next build
 * visits them in storage order.  Performance only; results never depend on the visit order. */
int ya_grid_forget_order(ya_grid* g);

/* A promise about where cells can be: every cell's cube id lies in [cube_lo, cube_hi).  Builds
 * then run their prefix sum (k_tile_sum, k_scan: 16 bytes written per cube of the grid, whatever
 * it holds) over the scan tiles of that range only -- once one build with the same cell count has
 * scanned the whole grid, which leaves offs[] = 0 / n and the -1 / -2 sentinels of cube_start /
 * cube_end below and above the range; a build with another count, or through
 * ya_grid_build_sorted_begin, scans everything again.  A z-slab of a decomposed system holds cells
 * in an eighth of the grid's planes (include/slab_logic.inc sets the range from its faces); the
 * full scan is 2 x 15 us of its 0.9 ms step at 10 M cells in 8 slabs.  A cell outside the range
 * raises YA_STATUS_OUT_OF_RANGE and is binned into the range's first or last cube (memory-safe,
 * like a cell outside the grid).  (0, n_cubes) or wider removes the promise.  New relative to the
 * reference (single GPU: its two thrust::fill of gs^3 ints per build, solvers.cuh:411-412). */
int ya_grid_set_cube_range(ya_grid* g, int cube_lo, int cube_hi);

/* Sticky status bits (YA_STATUS_*); blocking 4-byte read.  `clear` != 0
 * resets them. */
int ya_grid_status(ya_grid* g, int* bits, int clear);

/* ---- Centre-of-mass reduction ---------------------------------------- */

/* Replaces `thrust::reduce(d_dX, d_dX + n, Pt{0}) / n` (solvers.cuh:242,268)
 * without the device->host round trip.  d_v holds n points of n_floats packed
 * floats.  Writes to device memory d_out[0 .. n_floats) the mean (sum *
 * float(1.0 / n), the reference's operator/ arithmetic, dtypes.cuh:202-217)
 * and to d_out[n_floats .. 2 n_floats) the plain sum.  The summation order is
 * fixed (DESIGN.md "deterministic COM reduction"), so results are
 * reproducible run to run.  d_workspace must hold 1024 * n_floats floats. */
int ya_reduce_mean(const void* d_v, int n_floats, int n, float* d_out,
    float* d_workspace, void* stream);

/* The same sum in the same order, left as a rank of a z-slab decomposition puts it into a stage's
 * all-reduce: d_out[0 .. n_floats) = the sum, d_out[n_floats] = n & 4095, d_out[n_floats + 1] = n >> 12
 * (the cell count in two pieces that stay exact under a float sum).  d_out: n_floats + 2 floats. */
int ya_reduce_sum_packed(const void* d_v, int n_floats, int n, float* d_out, float* d_workspace,
    void* stream);

/* The first half of ya_reduce_mean alone: the B = clamp(ceil(n / 256), 1, 1024) per-block partial sums
 * {sum[n_floats]} left in d_workspace, *n_partials = B.  For update kernels that fold the partials
 * themselves (include/solvers.cuh, ya::fixed_velocity_from_partials: the same tree, the same bits as
 * ya_reduce_mean's second launch, which they replace). */
int ya_reduce_partials(const void* d_v, int n_floats, int n, float* d_workspace, int* n_partials, void* stream);
/* Size in bytes of the workspace ya_reduce_mean needs. */
size_t ya_reduce_workspace_bytes(int n_floats);

/* ---- Ordered selection and row gather (multi-GPU slab decomposition) ------ */

/* Indices i in [0, n), ascending, of the points whose third float (z) satisfies
 * z_min <= z < z_max, written to d_idx; their number to *d_count (device).
 * Deterministic (two-pass count + ordered write, no atomics).  New relative to
 * the reference, which is single-GPU; used to pick the ghost layer a z-slab
 * sends to its neighbour and the cells that migrate (DESIGN.md "Multi-GPU"). */
int ya_select_z(const void* d_X, size_t stride_bytes, int n, float z_min, float z_max,
    int* d_idx, int* d_count, int* d_workspace, void* stream);
size_t ya_select_workspace_bytes(int n_max);

/* d_dst[k] = d_src[d_idx[k]] for k < min(*d_count, cap); rows of row_bytes
 * (a multiple of 4) bytes.  The count stays on the device: no host round trip. */
int ya_gather_rows(const void* d_src, size_t row_bytes, const int* d_idx, const int* d_count,
    int cap, void* d_dst, void* stream);

/* Two such gathers from one array in one launch: d_dst0[k] = d_src[d_idx0[k]] for
 * k < min(*d_count0, cap), d_dst1 likewise from d_idx1 / d_count1.  A NULL index list is an empty
 * gather.  (A z-slab packs a stage's right-hand sides for its two neighbours with it.) */
int ya_gather_rows_pair(const void* d_src, size_t row_bytes, const int* d_idx0, const int* d_count0,
    void* d_dst0, const int* d_idx1, const int* d_count1, void* d_dst1, int cap, void* stream);

/* Appends the rows of up to two fixed-capacity messages behind the n_own rows already in
 * d_dst: first min(*d_count_lo, cap) rows of d_src_lo, then min(*d_count_hi, cap) rows of
 * d_src_hi (a NULL count = no message).  The counts are read on the device; the new total
 * is left in *d_n_out (may be NULL), so that a grid build can be begun on it
 * (ya_grid_build_sorted_begin) before the host has read the counts. */
int ya_append_rows(void* d_dst, size_t row_bytes, int n_own, const void* d_src_lo,
    const int* d_count_lo, const void* d_src_hi, const int* d_count_hi, int cap, int* d_n_out,
    void* stream);

/* ---- z-slab decomposition: drift guard, fixed point, a stage's all-reduce payload ---------- */

/* A cell's record in a slab is three arrays with their own row widths -- X (the point), old_v (12 bytes),
 * the global id (4) --; these move all three in ONE launch (a row width of 0 skips an array).
 * ya_pack_cells: a message = header_bytes (>= 16; the first int = *d_count as it is, so that a
 * receiver sees an overflow), then `cap` rows of array 0, of array 1, of array 2; rows k <
 * min(*d_count, cap) are d_src[f][d_idx[k]].
 * ya_append_cells: the rows of up to two such messages (NULL = none) go behind the n_own rows of
 * d_dst[f]: min(count, cap) of the lower one, then of the upper one; *d_n_out = the new total,
 * d_counts_out[0 .. 1] = the counts as sent (either may be NULL).
 * ya_fill_holes: cells whose indices are in the ascending lists d_leave_lo / d_leave_hi have left;
 * the first n_new rows are made whole again by moving the staying cells of the tail [n_new, ...) --
 * d_movers[k] + n_new, ascending, *d_count_movers of them -- into the holes below n_new (lower list's
 * holes first).  max_holes sizes the launch.  Every other cell keeps its row. */
int ya_pack_cells(const void* const d_src[3], const size_t row_bytes[3], const int* d_idx, const int* d_count, int cap,
    void* d_message, size_t header_bytes, void* stream);
int ya_append_cells(void* const d_dst[3], const size_t row_bytes[3], int n_own, const void* d_message_lo,
    const void* d_message_hi, int cap, size_t header_bytes, int* d_n_out, int* d_counts_out, void* stream);
int ya_fill_holes(void* const d_arrays[3], const size_t row_bytes[3], const int* d_leave_lo, const int* d_count_lo,
    const int* d_leave_hi, const int* d_count_hi, const int* d_movers, const int* d_count_movers, int n_new,
    int max_holes, void* stream);

/* d_dst[i] = float number `component` of row i of d_src (rows of stride_bytes): a slab keeps the z
 * of its own and mirrored cells at the moment the mirrored cells were chosen. */
int ya_copy_component(const void* d_src, size_t stride_bytes, int component, int n, float* d_dst, void* stream);
/* *d_index = the position of `id` in d_ids[0 .. n) (unique ids), or -1: which local cell is the
 * fixed point of set_fixed(i) / set_fixed_xy(i) (solvers.cuh:197-208), if this rank owns it. */
int ya_find_id(const int* d_ids, int n, int id, int* d_index, void* stream);
/* d_partial[b] = max over block b's share of w_i |d_a[i] - d_b[i]|, i < n (element strides in bytes; a NaN
 * counts as +inf), with w_i = 1 where d_b[i] lies within `width` of lo_face or hi_face (either may be
 * infinite) and 1/2 elsewhere: the drift guard weighs a cell by how near a face of its slab it was
 * (include/solvers.cuh, ya::Guard_band).  ya_max_abs_diff_partials(n) <= 1024 partials are written. */
int ya_max_abs_diff(const float* d_a, size_t a_stride_bytes, const float* d_b, size_t b_stride_bytes, int n,
    float lo_face, float hi_face, float width, float* d_partial, void* stream);
int ya_max_abs_diff_partials(int n);
/* The drift guard between a step's two stages.  d_state = {moved, predicted, request, error}: moved =
 * max of the n_moved partials (|z - z at selection| as the previous step left the cells), predicted =
 * max of the n_pred partials (this step's predictor |dz|); error (sticky) if moved + predicted exceeds
 * `limit` (the second stage would compute with it), request if moved + lag_steps * predicted does.
 * Both lists of partials are left zeroed (the update kernels of include/solvers.cuh fold their maxima
 * into 256 slots by atomic max).
 * (A z-slab's mirrored cells are valid only while no cell has moved further than
 * (halo - cube_size) / 2 since they were chosen; include/slab_logic.inc.) */
int ya_slab_guard_update(float* d_moved_partial, int n_moved, float* d_pred_partial, int n_pred,
    float limit, float lag_steps, float* d_state, void* stream);
/* ya_reduce_sum_packed plus what else a rank puts into a stage's all-reduce: d_out[n_floats + 2] = its
 * vote for an early re-selection and [n_floats + 3] = its error vote (d_guard_state[2], [3] if
 * with_votes, plus 1 if host_error), [n_floats + 4 .. + 6] = x, y, z of row *d_fix_index of d_v (the
 * fixed point's right-hand side, solvers.cuh:250-253,269-272) or zeros if that index is negative or
 * the pointer NULL, [n_floats + 7] = 0.  With fold_guard the drift guard is brought up to date first,
 * in the same kernel (ya_slab_guard_update's arguments).  d_out: n_floats + 8 floats. */
int ya_slab_pack(const void* d_v, int n_floats, int n, float* d_out, float* d_ws, float* d_moved_partial,
    int n_moved, float* d_pred_partial, int n_pred, float limit, float lag_steps, float* d_guard_state,
    int fold_guard, int with_votes, int host_error, const int* d_fix_index, void* stream);

/* `bytes` (<= 4096) read back without stalling the stream that produces them: _begin queues the copy
 * into pinned memory behind everything already in `stream`, _end waits for it (ya_n_reader for any
 * small record: the all-reduced votes of a z-slab step are collected one step later). */
typedef struct ya_async_read ya_async_read;
int ya_async_read_create(size_t bytes, ya_async_read** out);
int ya_async_read_destroy(ya_async_read* r);
int ya_async_read_begin(ya_async_read* r, const void* d_src, void* stream);
/* ... or written by a kernel itself: _target is the (device-visible) address the kernel stores the
 * record at, _mark notes that everything queued in `stream` so far includes that kernel. */
void* ya_async_read_target(ya_async_read* r);
int ya_async_read_mark(ya_async_read* r, void* stream);
int ya_async_read_end(ya_async_read* r, void* h_out);

/* ---- Slab neighbours over RCCL (multi-GPU, one process per GPU) ------------- */

/* The reference is single-GPU (no communication backend); these entry
 * points are what its Grid_solver needs to run as z-slabs on the GPUs of one node
 * (SURVEY.md section 8e): a communicator per process, the exchange of a slab's two ghost
 * messages with its two neighbours, and the sum of a few floats over all slabs.  RCCL is
 * loaded on first use (dlopen of librccl.so.1: a single-GPU process never loads it);
 * ncclSend / ncclRecv inside one group go point-to-point over the direct xGMI links.
 * All return 0 or a non-zero error code after printing the RCCL / socket error. */
typedef struct ya_comm ya_comm;
#define YA_COMM_ID_BYTES 128

/* Rank 0: a fresh unique id (ncclGetUniqueId) to hand to every rank. */
int ya_comm_unique_id(void* id_out_128_bytes);
/* Every rank, on its current device: ncclCommInitRank.  world == 1 needs no id and no RCCL
 * (id NULL); world == 1 WITH an id makes a real one-rank RCCL communicator. */
int ya_comm_create(const void* id_128_bytes, int rank, int world, ya_comm** out);
/* The same with the id passed from rank 0 over TCP: RANK, WORLD_SIZE, MASTER_ADDR and
 * MASTER_PORT from the environment (the variables torch.distributed.run sets); rank 0 listens
 * on MASTER_PORT + port_offset (and gives up after ten minutes without a connection).  For
 * model programs without any other launcher support.  With WORLD_SIZE > 1 the calling process
 * is put on GPU LOCAL_RANK (default RANK) modulo the visible devices first, unless it already
 * left device 0 itself or YALLA_KEEP_DEVICE=1: one process per GPU, RCCL refuses two ranks on
 * one.  Returns 997 if that device cannot be selected, 998 for rendezvous failures. */
int ya_comm_create_from_env(int port_offset, ya_comm** out);
/* `world` communicators for the slabs of ONE process on one GPU (out: an array of `world` handles): the
 * same entry points below, stream-ordered like RCCL's -- messages are device-to-device copies behind
 * the sender's stream, a send completes when its receiver has the data, the all-reduce is a kernel
 * that sums in rank order -- with every communicator driven by a host thread of its own.  For running
 * the decomposed step's asynchronous choreography against real peers where RCCL cannot (it refuses
 * two ranks on one GPU): tests/test_slab.py.  Destroy every handle. */
int ya_comm_create_loopback(int world, ya_comm** out);
int ya_comm_destroy(ya_comm* comm);
int ya_comm_rank(const ya_comm* comm);
int ya_comm_world(const ya_comm* comm);
/* Slab rank r sends `bytes` bytes of d_send_lo to rank r - 1 and of d_send_hi to rank r + 1 and
 * receives as many into d_recv_lo / d_recv_hi from them, all in one RCCL group on `stream`
 * (the first and last rank have one neighbour; their other buffers may be NULL). */
int ya_comm_exchange(ya_comm* comm, const void* d_send_lo, void* d_recv_lo, const void* d_send_hi,
    void* d_recv_hi, size_t bytes, void* stream);
/* The same with a size per message: what goes to a neighbour need not be as long as what comes
 * from it (ghost rows travel at their exact length).  A size of 0 skips that message. */
int ya_comm_exchange_v(ya_comm* comm, const void* d_send_lo, size_t send_lo_bytes, void* d_recv_lo,
    size_t recv_lo_bytes, const void* d_send_hi, size_t send_hi_bytes, void* d_recv_hi, size_t recv_hi_bytes,
    void* stream);
/* `bytes` bytes of d_send to this rank itself (one RCCL group of ncclSend + ncclRecv): what a
 * one-GPU machine can check of the binding ya_comm_exchange uses. */
int ya_comm_self_exchange(ya_comm* comm, const void* d_send, void* d_recv, size_t bytes, void* stream);
/* In-place sum of `count` floats over all ranks (ncclAllReduce) on `stream`. */
int ya_comm_allreduce_sum(ya_comm* comm, float* d_buf, int count, void* stream);
/* The same on host memory through the devices (a bounce buffer): for the few control values a
 * program needs once (timings, totals); blocking. */
int ya_comm_allreduce_host(ya_comm* comm, double* values, int count, int take_max);
/* What RCCL itself says about this communicator (so that a program can prove that N ranks sit on N
 * distinct GPUs instead of trusting its environment variables):
 *   info[0] = ncclCommCount, info[1] = ncclCommUserRank, info[2] = ncclCommCuDevice,
 *   info[3] = the calling thread's current HIP device, info[4] = 1 for a real RCCL communicator,
 *             0 for none (world 1 without an id) and 2 for a loopback one (then info[0..2] are its own
 *             world, rank and current device),
 *   info[5] = the device's PCI location as (domain << 16) | (bus << 8) | (device << 3) | function,
 *   info[6], info[7] = 0;
 * pci_bus_id (may be NULL): hipDeviceGetPCIBusId's string of info[2]'s device, at most 31 characters. */
int ya_comm_info(const ya_comm* comm, int info[8], char pci_bus_id[32]);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* YALLA_HIP_H */
