// Per-workgroup time stamps of a grid_force_bits launch (tools/micro/force_trace.hip,
// tools/force_trace_summary.py): include BEFORE solvers.cuh.  Every workgroup records s_memtime at its
// entry and exit with the hardware id of the CU and XCD it ran on: 4 words per block in ya_bits_trace.
// A measurement build, not the product kernel (the hooks are empty macros otherwise).
#pragma once
#include <hip/hip_runtime.h>

__device__ unsigned long long* ya_bits_trace = nullptr;

#define YA_BITS_PROBE_BEGIN                                                                   \
    unsigned long long trace_t0;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trace_t0)::"memory");
#define YA_BITS_PROBE_END(tile_)                                                              \
    {                                                                                         \
        unsigned long long trace_t1;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trace_t1)::"memory");      \
        unsigned hw_id, xcc_id;                                                               \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));                   \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));                 \
        if (threadIdx.x == 0 && ya_bits_trace) {                                              \
            unsigned long long* out = ya_bits_trace + 4ull * blockIdx.x;                      \
            out[0] = trace_t0;                                                                \
            out[1] = trace_t1;                                                                \
            out[2] = ((unsigned long long)xcc_id << 32) | hw_id;                              \
            out[3] = (unsigned long long)(tile_);                                             \
        }                                                                                     \
    }
