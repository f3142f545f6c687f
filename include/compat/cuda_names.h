// cuda_names.h -- OPTIONAL, for compiling UNMODIFIED ya||a model files with hipcc.
//
// The reference's own examples/*.cu and tests/*.cu spell runtime calls the CUDA
// way (cudaMalloc, cudaMemcpyToSymbol, curand_uniform ...).  Model files are
// user code, so this is the one place a name map is acceptable; the engine
// (include/*.cuh, yalla_amd/csrc) is written in HIP and never includes this
// file.  Use:  hipcc -x hip -std=c++17 -include compat/cuda_names.h -Iinclude/compat ...
#pragma once

#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>

#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
#define cudaMemcpyToSymbol(symbol, ...) hipMemcpyToSymbol(HIP_SYMBOL(symbol), __VA_ARGS__)
#define cudaMemcpyFromSymbol(dst, symbol, ...) hipMemcpyFromSymbol(dst, HIP_SYMBOL(symbol), __VA_ARGS__)
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess

#define curandState hiprandState
#define curand_init hiprand_init
#define curand_uniform hiprand_uniform
#define curand_normal hiprand_normal

// clang rejects min(int, unsigned) that nvcc accepts (examples/intercalation.cu:50)
__host__ __device__ inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }
__host__ __device__ inline int min(unsigned a, int b) { return (int)a < b ? (int)a : b; }
__host__ __device__ inline int max(int a, unsigned b) { return a > (int)b ? a : (int)b; }
__host__ __device__ inline int max(unsigned a, int b) { return (int)a > b ? (int)a : b; }

// CUDA offers min/max(float, float) and (double, double) in HOST code as well;
// HIP's host overloads are integer only, so `min(a.x, b.x)` in host code of a
// model file would silently truncate (tests/test_inits.cu:83-92).
__host__ inline float min(float a, float b) { return a < b ? a : b; }
__host__ inline float max(float a, float b) { return a > b ? a : b; }
__host__ inline double min(double a, double b) { return a < b ? a : b; }
__host__ inline double max(double a, double b) { return a > b ? a : b; }
