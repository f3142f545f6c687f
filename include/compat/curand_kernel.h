// Lets `#include <curand_kernel.h>` in unmodified ya||a model files resolve to
// hipRAND (see cuda_names.h; only on the include path when such files are built).
#pragma once
#include "cuda_names.h"
