// libyalla_models.so -- the model harness (include/yalla_models.h) built on
// the HIP engine: include/*.cuh templates instantiated for the named models of
// model_functors.h.  Links against libyalla_hip.so.  Product path: no CPU
// code from oracle/ is included, linked or called here.
#include <hip/hip_runtime.h>

#include "dtypes.cuh"
#include "inits.cuh"
#include "links.cuh"
#include "property.cuh"
#include "solvers.cuh"

#include "model_functors.h"

#define YA_ZERO(d, bytes) YA_CHECK(ya_memset_async((d), 0, (bytes), nullptr))
#define YA_IS_DEVICE 1
#define YA_D2H(h, d, bytes) YA_CHECK(ya_memcpy_d2h((h), (d), (bytes)))
#define YA_H2D(d, h, bytes) YA_CHECK(ya_memcpy_h2d((d), (h), (bytes)))
#define YA_SYNC() YA_CHECK(ya_device_synchronize())

template<typename C>
void ya_harness_random_sphere(float dist_to_nb, C& cells, unsigned seed)
{
    random_sphere(dist_to_nb, cells, 0, seed);
}

// Backend operations of the z-slab decomposition (device build): ordered
// selection and row gathers come from libyalla_hip.so; counts stay on the device.
namespace harness_ops {
inline void* alloc(size_t bytes)
{
    void* p = nullptr;
    YA_CHECK(ya_malloc(&p, bytes));
    return p;
}
inline void zero(void* p, size_t bytes) { YA_CHECK(ya_memset_async(p, 0, bytes, nullptr)); }
inline void release(void* p) { ya_free(p); }
inline size_t select_workspace_bytes(int n_max) { return ya_select_workspace_bytes(n_max); }
inline void select_z(const void* X, size_t stride, int n, float z_min, float z_max, int* idx,
    int* count, int* ws)
{
    YA_CHECK(ya_select_z(X, stride, n, z_min, z_max, idx, count, ws, nullptr));
}
inline void gather_rows(const void* src, size_t row_bytes, const int* idx, const int* count, int cap,
    void* dst)
{
    YA_CHECK(ya_gather_rows(src, row_bytes, idx, count, cap, dst, nullptr));
}
inline void copy(void* dst, const void* src, size_t bytes)
{
    if (bytes) YA_CHECK(ya_memcpy_d2d_async(dst, src, bytes, nullptr));
}
inline int read_int(const void* d)
{
    int v;
    YA_CHECK(ya_memcpy_d2h(&v, d, sizeof(int)));
    return v;
}
inline void append_rows(void* dst, size_t row_bytes, int n_own, const void* lo, const void* hi,
    int cap, size_t payload_offset, int* n_out)
{
    // a message = 16-byte header {int count}, then the rows
    YA_CHECK(ya_append_rows(dst, row_bytes, n_own, lo ? (const char*)lo + payload_offset : nullptr,
        (const int*)lo, hi ? (const char*)hi + payload_offset : nullptr, (const int*)hi, cap, n_out,
        nullptr));
}
inline void read_ints(const void* d, int k, int* out)
{
    YA_CHECK(ya_memcpy_d2h(out, d, (size_t)k * sizeof(int)));
}
inline void write_int(void* d, int v) { YA_CHECK(ya_memcpy_h2d(d, &v, sizeof(int))); }
// The cell count travels through the float all-reduce as two exact pieces (low 12 bits and
// the rest): exact for any total below 2^36 however many ranks add up.
__global__ void k_mean_from_total(const float* total, int n_floats, float* fix)
{
    // fix = sum * float(1. / n): the reference's Pt / n arithmetic (dtypes.cuh:202-217)
    const double n = (double)total[n_floats] + 4096. * (double)total[n_floats + 1];
    const float inv = (float)(1. / n);
    if (threadIdx.x < 3) fix[threadIdx.x] = total[threadIdx.x] * inv;
}
inline void mean_from_total(const float* total, int n_floats, float* fix)
{
    k_mean_from_total<<<1, 64>>>(total, n_floats, fix);
}
__global__ void k_pack_sum(const float* sum, int n_floats, int n_own, float* out)
{
    if ((int)threadIdx.x < n_floats) out[threadIdx.x] = sum[threadIdx.x];
    if ((int)threadIdx.x == n_floats) {
        out[n_floats] = (float)(n_own & 4095);
        out[n_floats + 1] = (float)(n_own >> 12);
    }
}
inline void pack_sum(const float* sum, int n_floats, int n_own, float* out)
{
    k_pack_sum<<<1, 64>>>(sum, n_floats, n_own, out);
}
}  // namespace harness_ops

#include "models_harness.inc"
