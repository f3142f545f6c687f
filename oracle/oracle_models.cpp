// oracle/oracle_models.cpp -- TEST INFRASTRUCTURE ONLY.
// The model harness (include/yalla_models.h) built on the host-serial CPU
// restatement oracle/yalla_host.hpp, from the same model source as the HIP
// build (yalla_amd/csrc/model_functors.h + models_harness.inc).  Loaded only
// by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
#include "yalla_host.hpp"

#include "model_functors.h"

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

#include "models_harness.inc"
