// Per-cell properties that are not integrated (cell type, counters ...).
// API parity with ya||a `include/property.cuh:1-34`.
#pragma once

#include <stdlib.h>

#include <string>

#include "cudebug.cuh"
#include "yalla_hip.h"

template<typename Prop = int>
struct Property {
    Prop* h_prop;
    Prop* d_prop;
    std::string name;
    const int n_max;
    Property(int n_max, std::string name = "cell_type") : name{name}, n_max{n_max}
    {
        h_prop = (Prop*)calloc(n_max, sizeof(Prop));
        YA_CHECK(ya_malloc((void**)&d_prop, (size_t)n_max * sizeof(Prop)));
    }
    ~Property()
    {
        free(h_prop);
        ya_free(d_prop);
    }
    Property(const Property&) = delete;
    void copy_to_device()
    {
        YA_CHECK(ya_memcpy_h2d(d_prop, h_prop, (size_t)n_max * sizeof(Prop)));
    }
    void copy_to_host()
    {
        YA_CHECK(ya_memcpy_d2h(h_prop, d_prop, (size_t)n_max * sizeof(Prop)));
    }
};
