// Debugging helpers for model files.  API parity with ya||a
// `include/cudebug.cuh:1-35` (D_ASSERT, CHECK_CUDA); HIP spellings underneath.
#pragma once

#include <hip/hip_runtime.h>

#include <assert.h>
#include <stdio.h>
#include <stdlib.h>

// Device-side assertion (cudebug.cuh:6-14).
#define D_ASSERT(predicate) assert(predicate)

// Synchronise and abort on any pending HIP error (cudebug.cuh:19-35).
inline void ya_check_device(const char* file, int line)
{
    hipError_t launch_error = hipGetLastError();
    hipError_t run_error = hipDeviceSynchronize();
    if (launch_error != hipSuccess) {
        printf("Sync HIP error: %s, %s(%d).\n", hipGetErrorString(launch_error), file, line);
        exit(-1);
    }
    if (run_error != hipSuccess) {
        printf("Async HIP error: %s, %s(%d).\n", hipGetErrorString(run_error), file, line);
        exit(-1);
    }
}

#define CHECK_HIP ya_check_device(__FILE__, __LINE__)
#define CHECK_CUDA CHECK_HIP  // the name model files use

// The engine's own calls into libyalla_hip.so fail loudly: there is no CPU
// fallback behind them.
#define YA_CHECK(call)                                                          \
    do {                                                                        \
        int ya_err_ = (call);                                                   \
        if (ya_err_ != 0) {                                                     \
            fprintf(stderr, "yalla-hip: %s failed: %s (%s:%d)\n", #call,        \
                hipGetErrorString((hipError_t)ya_err_), __FILE__, __LINE__);    \
            abort();                                                            \
        }                                                                       \
    } while (0)
