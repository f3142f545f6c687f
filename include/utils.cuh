// Small helpers shared by several headers.  API parity with ya||a
// `include/utils.cuh:1-33`: split(), dot_product(), setup_rand_states().
#pragma once

#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>

#include <sstream>
#include <string>
#include <vector>

inline std::vector<std::string> split(const std::string& s)
{
    std::vector<std::string> words;
    std::istringstream in(s);
    for (std::string w; std::getline(in, w, ' ');) words.push_back(w);
    return words;
}

template<typename Pt_a, typename Pt_b>
__device__ __host__ float dot_product(Pt_a a, Pt_b b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}

// One XORWOW stream per cell: sequence i of `seed`, offset 0 (utils.cuh:29-33).
__global__ void setup_rand_states(int n_states, int seed, hiprandState* d_state)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_states) hiprand_init(seed, i, 0, &d_state[i]);
}
