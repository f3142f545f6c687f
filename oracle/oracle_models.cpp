// oracle/oracle_models.cpp -- TEST INFRASTRUCTURE ONLY.
// The model harness (include/yalla_models.h) built on the host-serial CPU
// restatement oracle/yalla_host.hpp, from the same model source as the HIP
// build (yalla_amd/csrc/model_functors.h + models_harness.inc).  Loaded only
// by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
#include "yalla_host.hpp"

#include "model_functors.h"

#define YA_ZERO(d, bytes) memset((d), 0, (bytes))
#define YA_IS_DEVICE 0
#define YA_D2H(h, d, bytes) memcpy((h), (d), (bytes))
#define YA_H2D(d, h, bytes) memcpy((d), (h), (bytes))
#define YA_SYNC() ((void)0)

template<typename C>
void ya_harness_random_sphere(float dist_to_nb, C& cells, unsigned seed)
{
    ya_oracle_random_sphere(dist_to_nb, cells.h_X, *cells.h_n, seed);
    cells.copy_to_device();
}

// Backend operations of the z-slab decomposition, restated serially ("device"
// memory is host memory).  Same contracts as include/yalla_hip.h's ya_select_z
// and ya_gather_rows.
struct Oracle_slab_ops {
    // what only the device has (include/slab.cuh: streams, events, RCCL, the split force launch, the fused
    // update kernels) is refused or absent here
    static constexpr bool device = false;
    using Stream = void*;
    struct Guard_band {
        float lo_face, hi_face, width;
    };
    static constexpr int guard_slots = 0;
    template<typename Pt>
    static bool has_generic_forces(const Generic_forces<Pt>&) { return true; }
    static int rows_over_rccl(void*, void*&, void*&, void*&, const void*, size_t, void*, size_t, const void*, size_t, void*,
        size_t) { return -7; }
    static void rows_wait(void*) {}
    static void rows_path_destroy(void*, void*, void*) {}
    static int rccl_is(void*, int, int) { return -2; }
    static int rccl_exchange(void*, const void*, void*, const void*, void*, size_t) { return -7; }
    static int rccl_allreduce(void*, float*, int) { return -7; }
    static void* alloc(size_t bytes) { return calloc(1, bytes ? bytes : 4); }
    static void zero(void* p, size_t bytes) { memset(p, 0, bytes); }
    static void release(void* p) { free(p); }
    static size_t select_workspace_bytes(int) { return 4; }
    static void select_z(const void* X, size_t stride, int n, float z_min, float z_max, int* idx,
    int* count, int*)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        const float z = *(const float*)((const char*)X + (size_t)i * stride + 8);
        if (z >= z_min && z < z_max) idx[m++] = i;
    }
    *count = m;
}
    static void gather_rows(const void* src, size_t row_bytes, const int* idx, const int* count, int cap,
    void* dst)
{
    const int m = *count < cap ? *count : cap;
    for (int k = 0; k < m; k++)
        memcpy((char*)dst + (size_t)k * row_bytes, (const char*)src + (size_t)idx[k] * row_bytes,
            row_bytes);
}
    static void gather_rows_pair(const void* src, size_t row_bytes, const int* idx0, const int* count0,
    void* dst0, const int* idx1, const int* count1, void* dst1, int cap)
{
    if (idx0) gather_rows(src, row_bytes, idx0, count0, cap, dst0);
    if (idx1) gather_rows(src, row_bytes, idx1, count1, cap, dst1);
}
    static void copy(void* dst, const void* src, size_t bytes) { memmove(dst, src, bytes); }
    static int read_int(const void* d) { return *(const int*)d; }
// a cell's record -- three arrays with their own row widths -- moved together (ya_pack_cells,
// ya_append_cells, ya_fill_holes of include/yalla_hip.h, restated serially)
    static void pack_cells(void* const arrays[3], const size_t row_bytes[3], const int* idx, const int* count, int cap,
    void* message, size_t header)
{
    int* head = (int*)message;
    head[0] = *count;
    head[1] = head[2] = head[3] = 0;
    const int m = *count < 0 ? 0 : (*count < cap ? *count : cap);
    char* out = (char*)message + header;
    for (int f = 0; f < 3; f++) {
        for (int k = 0; k < m; k++)
            memcpy(out + (size_t)k * row_bytes[f], (const char*)arrays[f] + (size_t)idx[k] * row_bytes[f], row_bytes[f]);
        out += (size_t)cap * row_bytes[f];
    }
}
    static void append_cells(void* const arrays[3], const size_t row_bytes[3], int n_own, const void* lo, const void* hi,
    int cap, size_t header, int* n_out, int* counts_out)
{
    const void* messages[2] = {lo, hi};
    int n = n_own;
    for (int d = 0; d < 2; d++) {
        const int sent = messages[d] ? *(const int*)messages[d] : 0;
        if (counts_out) counts_out[d] = sent;
        const int c = sent < 0 ? 0 : (sent > cap ? cap : sent);
        size_t offset = header;
        for (int f = 0; f < 3; f++) {
            if (c > 0)
                memcpy((char*)arrays[f] + (size_t)n * row_bytes[f], (const char*)messages[d] + offset, (size_t)c * row_bytes[f]);
            offset += (size_t)cap * row_bytes[f];
        }
        n += c;
    }
    if (n_out) *n_out = n;
}
    static void fill_holes(void* const arrays[3], const size_t row_bytes[3], const int* leave_lo, const int* count_lo,
    const int* leave_hi, const int* count_hi, const int* movers, const int* count_movers, int n_new, int)
{
    int k = 0;
    const int* lists[2] = {leave_lo, leave_hi};
    const int* counts[2] = {count_lo, count_hi};
    for (int d = 0; d < 2; d++)
        for (int j = 0; lists[d] && j < *counts[d] && lists[d][j] < n_new && k < *count_movers; j++, k++)
            for (int f = 0; f < 3; f++)
                memcpy((char*)arrays[f] + (size_t)lists[d][j] * row_bytes[f],
                    (const char*)arrays[f] + (size_t)(n_new + movers[k]) * row_bytes[f], row_bytes[f]);
}
    static void read_ints(const void* d, int k, int* out) { memcpy(out, d, (size_t)k * sizeof(int)); }
    static void write_int(void* d, int v) { *(int*)d = v; }
// The cell count travels through the float all-reduce as two exact pieces (low 12 bits and
// the rest): exact for any total below 2^36 however many ranks add up.
// fix_mode 0: the mean, sum * float(1. / float(n)) (the reference's Pt / n, dtypes.cuh:202-217);
// 1: the fixed point's right-hand side (total[n_floats + 4 ..]), 2: its x and y, the mean's z
// (solvers.cuh:241-253)
    static void mean_from_total(const float* total, int n_floats, float* fix, int fix_mode)
{
    const double n = (double)total[n_floats] + 4096. * (double)total[n_floats + 1];
    const float inv = (float)(1. / (double)(float)n);
    for (int k = 0; k < 3; k++) fix[k] = total[k] * inv;
    if (fix_mode == 0) return;
    const float* point = total + n_floats + 4;
    fix[0] = point[0];
    fix[1] = point[1];
    if (fix_mode == 1) fix[2] = point[2];
}
// votes and the fixed point's right-hand side behind the packed sum (ya_slab_pack on the device)
    static void pack_extra(const float* rhs, int n_floats, const float* guard_state, int with_votes,
    int host_error, const int* fix_index, float* out)
{
    out[n_floats + 2] = with_votes ? guard_state[2] : 0.f;
    out[n_floats + 3] = (with_votes ? guard_state[3] : 0.f) + (host_error ? 1.f : 0.f);
    const int f = fix_index ? *fix_index : -1;
    for (int k = 0; k < 3; k++) out[n_floats + 4 + k] = f >= 0 ? rhs[(size_t)f * n_floats + k] : 0.f;
    out[n_floats + 7] = 0.f;
}
    static void copy_z(const void* X, size_t stride, int n, float* z)
{
    for (int i = 0; i < n; i++) z[i] = *(const float*)((const char*)X + (size_t)i * stride + 8);
}
    static void find_id(const int* ids, int n, int id, int* index)
{
    *index = -1;
    for (int i = 0; i < n; i++)
        if (ids[i] == id) *index = i;
}
// the drift guard's weighted maximum (ya_max_abs_diff): full weight for cells that were within `width`
// of a face of the slab, half elsewhere
    static int max_abs_diff(const float* a, size_t a_stride, const float* b, size_t b_stride, int n, float lo_face,
    float hi_face, float width, float* partial)
{
    float m = 0.f;
    for (int i = 0; i < n; i++) {
        const float z = *(const float*)((const char*)b + (size_t)i * b_stride);
        const float w = fabsf(z - lo_face) <= width || fabsf(z - hi_face) <= width ? 1.f : 0.5f;
        const float d = fabsf(*(const float*)((const char*)a + (size_t)i * a_stride) - z) * w;
        m = d == d ? fmaxf(m, d) : INFINITY;
    }
    partial[0] = m;
    return 1;
}
// the drift guard's bookkeeping (ya_slab_guard_update, yalla_amd/csrc/core.hip k_slab_guard)
    static void guard_update(const float* moved, int n_moved, const float* pred, int n_pred, float limit, float lag,
    float* state)
{
    float a = 0.f, p = 0.f;
    for (int k = 0; k < n_moved; k++) a = fmaxf(a, moved[k]);
    for (int k = 0; k < n_pred; k++) p = fmaxf(p, pred[k]);
    state[0] = a;
    state[1] = p;
    state[2] = !(a + lag * p <= limit) ? 1.f : 0.f;
    state[3] = !(a + p <= limit) ? 1.f : state[3];
}
    static void* votes_create() { return calloc(2, sizeof(float)); }
    static void votes_destroy(void* r) { free(r); }
    static void votes_begin(void* r, const float* votes) { memcpy(r, votes, 2 * sizeof(float)); }
    static void votes_end(void* r, float* votes) { memcpy(votes, r, 2 * sizeof(float)); }
    static float* votes_target(void* r) { return (float*)r; }
    static void votes_mark(void*) {}
    static void pack_sum(const float* sum, int n_floats, int n_own, float* out)
{
    for (int k = 0; k < n_floats; k++) out[k] = sum[k];
    out[n_floats] = (float)(n_own & 4095);
    out[n_floats + 1] = (float)(n_own >> 12);
}
    static void sync() {}
    static void d2h(void* h, const void* d, size_t bytes) { memcpy(h, d, bytes); }
    static void h2d(void* d, const void* h, size_t bytes) { memcpy(d, h, bytes); }
};
using harness_ops = Oracle_slab_ops;

#include "slab_logic.inc"  // the backend-independent slab logic the HIP engine ships (include/)


#include "models_harness.inc"


// ---- pair trace (oracle/yalla_host.hpp, Pair_trace): diagnostics for tools/diag/slab_case.py ----------------
// Only the oracle library exports these; nothing in include/ declares them.
extern "C" {
__attribute__((visibility("default"))) int ya_oracle_trace_begin(const int* ids, int n_ids, float margin)
{
    if (!ids || n_ids < 0) return -3;
    Pair_trace& t = pair_trace();
    std::lock_guard<std::mutex> hold(t.lock);
    t.ids.assign(ids, ids + n_ids);
    std::sort(t.ids.begin(), t.ids.end());
    t.margin = margin;
    t.recs.clear();
    t.armed.store(true);
    return 0;
}
// rows of {call, i, j, dist as bits}; returns how many records there are (read again with more room if > cap)
__attribute__((visibility("default"))) int ya_oracle_trace_read(int* rows, int cap)
{
    Pair_trace& t = pair_trace();
    std::lock_guard<std::mutex> hold(t.lock);
    const int n = (int)t.recs.size();
    for (int k = 0; k < n && k < cap && rows; k++) {
        rows[4 * k + 0] = t.recs[k].call;
        rows[4 * k + 1] = t.recs[k].i;
        rows[4 * k + 2] = t.recs[k].j;
        memcpy(&rows[4 * k + 3], &t.recs[k].dist, sizeof(float));
    }
    return n;
}
__attribute__((visibility("default"))) int ya_oracle_trace_end(void)
{
    Pair_trace& t = pair_trace();
    std::lock_guard<std::mutex> hold(t.lock);
    t.armed.store(false);
    t.recs.clear();
    t.ids.clear();
    return 0;
}
}
