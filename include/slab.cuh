// Grid_solver on several GPUs of one node: z-slabs, one process per GPU (SURVEY.md section 8e,
// DESIGN.md section 7).  New relative to the reference (single-GPU); everything else about
// Solution<Pt, Slab_grid_solver> is Solution<Pt, Grid_solver>.
//
//     ya_comm* comm;
//     ya_comm_create_from_env(1, &comm);                       // RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun ...):
//                                                              // selects GPU LOCAL_RANK, then ncclCommInitRank -- call it
//                                                              // BEFORE constructing the Solution (its arrays live on the
//                                                              // device that is current then)
//     const ya::Slab_plan plan = ya::slab_plan(whole, n, ya_comm_world(comm), cube_size);   // whole[0 .. n): the system, on every rank
//     Solution<float3, Slab_grid_solver> cells{plan.n_max, grid_size, cube_size};
//     cells.slab_adopt(plan, ya_comm_rank(comm), whole, n, cells.h_X, cells.h_n);    // this rank's cells, slab_init, slab_setup
//     cells.slab_use_rccl(comm);
//     for (...) cells.take_step<my_force>(dt);                  // exchanges, all-reduces and migration inside
//     n_own = cells.slab.n_own;  cells.copy_to_host();          // own cells are h_X[0 .. n_own), their ids: get_own(X, ids)
//
// Pairwise functors are called with the cells' GLOBAL ids.  Generic forces see the local arrays
// (own cells first, then this stage's ghost cells).
#pragma once

#include "solvers.cuh"
#include "yalla_hip.h"

namespace ya {
__global__ void k_slab_mean_from_total(const float* total, int n_floats, float* fix, int fix_mode)
{
    // ya::fix_from_total (solvers.cuh): the mean as sum * float(1. / float(n)), the reference's Pt / n
    // arithmetic (dtypes.cuh:202-217; the cell count crosses the float all-reduce as two exact pieces),
    // or the fixed point's value
    const float3 f = fix_from_total(total, n_floats, fix_mode);
    if (threadIdx.x == 0) {
        fix[0] = f.x;
        fix[1] = f.y;
        fix[2] = f.z;
    }
}
// Backend operations of the slab logic on the device: thin wrappers over the C ABI.
struct Slab_device_ops {
    static constexpr bool device = true;
    using Stream = hipStream_t;
    using Guard_band = ya::Guard_band;
    static constexpr int guard_slots = ya::GUARD_SLOTS;
    template<typename Pt>
    static bool has_generic_forces(const Generic_forces<Pt>& gen) { return !ya::is_no_gen_forces<Pt>(gen); }
    // a stage's rows over RCCL: on a stream of their own between two events -- they leave once the packed
    // rows are complete (the step's stream has reached `packed`), the update waits for `landed` (rows_wait),
    // the interior launch runs meanwhile
    static int rows_over_rccl(void* rccl, void*& stream, void*& packed, void*& landed, const void* send_lo, size_t out_lo,
        void* recv_lo, size_t in_lo, const void* send_hi, size_t out_hi, void* recv_hi, size_t in_hi)
    {
        if (!stream) {
            hipStream_t st;
            hipEvent_t a, b;
            YA_CHECK((int)hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            YA_CHECK((int)hipEventCreateWithFlags(&a, hipEventDisableTiming));
            YA_CHECK((int)hipEventCreateWithFlags(&b, hipEventDisableTiming));
            stream = st;
            packed = a;
            landed = b;
        }
        YA_CHECK((int)hipEventRecord((hipEvent_t)packed, nullptr));
        YA_CHECK((int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)packed, 0));
        const int rc = ya_comm_exchange_v((ya_comm*)rccl, send_lo, out_lo, recv_lo, in_lo, send_hi, out_hi, recv_hi, in_hi,
            stream);
        YA_CHECK((int)hipEventRecord((hipEvent_t)landed, (hipStream_t)stream));
        return rc;
    }
    static void rows_wait(void* landed) { YA_CHECK((int)hipStreamWaitEvent(nullptr, (hipEvent_t)landed, 0)); }
    static void rows_path_destroy(void* stream, void* packed, void* landed)
    {
        if (!stream) return;
        (void)hipStreamDestroy((hipStream_t)stream);
        (void)hipEventDestroy((hipEvent_t)packed);
        (void)hipEventDestroy((hipEvent_t)landed);
    }
    static int rccl_is(void* comm, int rank, int world)
    {
        return comm && ya_comm_rank((ya_comm*)comm) == rank && ya_comm_world((ya_comm*)comm) == world ? 0 : -3;
    }
    static int rccl_exchange(void* rccl, const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes)
    {
        return ya_comm_exchange((ya_comm*)rccl, send_lo, recv_lo, send_hi, recv_hi, bytes, nullptr);
    }
    static int rccl_allreduce(void* rccl, float* buf, int count) { return ya_comm_allreduce_sum((ya_comm*)rccl, buf, count, nullptr); }
    static void* alloc(size_t bytes)
    {
        void* p = nullptr;
        YA_CHECK(ya_malloc(&p, bytes));
        return p;
    }
    static void zero(void* p, size_t bytes) { YA_CHECK(ya_memset_async(p, 0, bytes, nullptr)); }
    static void release(void* p) { ya_free(p); }
    static void sync() { YA_CHECK(ya_device_synchronize()); }
    static void d2h(void* h, const void* d, size_t bytes) { YA_CHECK(ya_memcpy_d2h(h, d, bytes)); }
    static void h2d(void* d, const void* h, size_t bytes) { YA_CHECK(ya_memcpy_h2d(d, h, bytes)); }
    static size_t select_workspace_bytes(int n_max) { return ya_select_workspace_bytes(n_max); }
    static void select_z(const void* X, size_t stride, int n, float z_min, float z_max, int* idx, int* count,
        int* ws)
    {
        YA_CHECK(ya_select_z(X, stride, n, z_min, z_max, idx, count, ws, nullptr));
    }
    static void gather_rows_pair(const void* src, size_t row_bytes, const int* idx0, const int* count0, void* dst0,
        const int* idx1, const int* count1, void* dst1, int cap)
    {
        YA_CHECK(ya_gather_rows_pair(src, row_bytes, idx0, count0, dst0, idx1, count1, dst1, cap, nullptr));
    }
    static void copy(void* dst, const void* src, size_t bytes)
    {
        if (bytes) YA_CHECK(ya_memcpy_d2d_async(dst, src, bytes, nullptr));
    }
    static int read_int(const void* d)
    {
        int v;
        YA_CHECK(ya_memcpy_d2h(&v, d, sizeof(int)));
        return v;
    }
    static void pack_cells(void* const arrays[3], const size_t row_bytes[3], const int* idx, const int* count, int cap,
        void* message, size_t header)
    {
        YA_CHECK(ya_pack_cells(arrays, row_bytes, idx, count, cap, message, header, nullptr));
    }
    static void append_cells(void* const arrays[3], const size_t row_bytes[3], int n_own, const void* lo, const void* hi,
        int cap, size_t header, int* n_out, int* counts_out)
    {
        YA_CHECK(ya_append_cells(arrays, row_bytes, n_own, lo, hi, cap, header, n_out, counts_out, nullptr));
    }
    static void fill_holes(void* const arrays[3], const size_t row_bytes[3], const int* leave_lo, const int* count_lo,
        const int* leave_hi, const int* count_hi, const int* movers, const int* count_movers, int n_new, int max_holes)
    {
        YA_CHECK(ya_fill_holes(arrays, row_bytes, leave_lo, count_lo, leave_hi, count_hi, movers, count_movers, n_new,
            max_holes, nullptr));
    }
    static void read_ints(const void* d, int k, int* out) { YA_CHECK(ya_memcpy_d2h(out, d, (size_t)k * sizeof(int))); }
    static void write_int(void* d, int v) { YA_CHECK(ya_memcpy_h2d(d, &v, sizeof(int))); }
    static void mean_from_total(const float* total, int n_floats, float* fix, int fix_mode)
    {
        k_slab_mean_from_total<<<1, 64>>>(total, n_floats, fix, fix_mode);
    }
    // drift guard and fixed point (include/yalla_hip.h)
    static void copy_z(const void* X, size_t stride, int n, float* z) { YA_CHECK(ya_copy_component(X, stride, 2, n, z, nullptr)); }
    static void find_id(const int* ids, int n, int id, int* index) { YA_CHECK(ya_find_id(ids, n, id, index, nullptr)); }
    static int max_abs_diff(const float* a, size_t a_stride, const float* b, size_t b_stride, int n, float lo_face,
        float hi_face, float width, float* partial)
    {
        YA_CHECK(ya_max_abs_diff(a, a_stride, b, b_stride, n, lo_face, hi_face, width, partial, nullptr));
        return ya_max_abs_diff_partials(n);
    }
    static void guard_update(float* moved, int n_moved, float* pred, int n_pred, float limit, float lag, float* state)
    {
        YA_CHECK(ya_slab_guard_update(moved, n_moved, pred, n_pred, limit, lag, state, nullptr));
    }
    // the all-reduced votes on their way to the host: queued behind the all-reduce, collected a step later
    static void* votes_create()
    {
        ya_async_read* r = nullptr;
        YA_CHECK(ya_async_read_create(2 * sizeof(float), &r));
        return r;
    }
    static void votes_destroy(void* r) { (void)ya_async_read_destroy((ya_async_read*)r); }
    static void votes_begin(void* r, const float* d_votes) { YA_CHECK(ya_async_read_begin((ya_async_read*)r, d_votes, nullptr)); }
    // ... or stored by the corrector kernel itself at votes_target(); votes_mark: that kernel is in the stream
    static float* votes_target(void* r) { return (float*)ya_async_read_target((ya_async_read*)r); }
    static void votes_mark(void* r) { YA_CHECK(ya_async_read_mark((ya_async_read*)r, nullptr)); }
    static void votes_end(void* r, float* votes) { YA_CHECK(ya_async_read_end((ya_async_read*)r, votes)); }
};
}  // namespace ya

#include "slab_logic.inc"

template<typename Pt>
using Slab_grid_solver = Slab_grid_solver_impl<Pt, ya::Slab_device_ops>;
