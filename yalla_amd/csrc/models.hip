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
#include "slab.cuh"

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

// Backend operations of the z-slab decomposition (device build): include/slab.cuh
using harness_ops = ya::Slab_device_ops;

#include "models_harness.inc"
