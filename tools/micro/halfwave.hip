// Does gfx950 skip the half of a wave64 VALU instruction whose 32 lanes are all masked off?
// Times a VALU-bound loop with all lanes, lanes 0-31, lanes 32-63, lanes 0-15, odd lanes only,
// and the packed-vs-scalar fp32 rate.
#include <hip/hip_runtime.h>
#include <cstdio>

template<int MODE>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float a, float b)
{
    const int lane = threadIdx.x & 63;
    bool on = true;
    if (MODE == 1) on = lane < 32;
    if (MODE == 2) on = lane >= 32;
    if (MODE == 3) on = lane < 16;
    if (MODE == 4) on = lane & 1;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    if (on) {
        for (int k = 0; k < iters; k++) {
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void spin_pk(float* out, int iters, float a, float b)
{
    v2f x0{(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    const v2f va{a, a}, vb{b, b};
    for (int k = 0; k < iters; k++) {
        x0 = __builtin_elementwise_fma(x0, va, vb); x1 = __builtin_elementwise_fma(x1, va, vb);
        x2 = __builtin_elementwise_fma(x2, va, vb); x3 = __builtin_elementwise_fma(x3, va, vb);
    }
    const v2f s = x0 + x1 + x2 + x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template<typename K>
static float time_it(K launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, nullptr);
    launch();
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int blocks = 256 * 8, iters = 20000;
    float* d;
    (void)hipMalloc(&d, blocks * 256 * sizeof(float));
    const float a = 0.999f, b = 0.001f;
    const float t0 = time_it([&] { spin<0><<<blocks, 256>>>(d, iters, a, b); });
    const float t1 = time_it([&] { spin<1><<<blocks, 256>>>(d, iters, a, b); });
    const float t2 = time_it([&] { spin<2><<<blocks, 256>>>(d, iters, a, b); });
    const float t3 = time_it([&] { spin<3><<<blocks, 256>>>(d, iters, a, b); });
    const float t4 = time_it([&] { spin<4><<<blocks, 256>>>(d, iters, a, b); });
    const float tp = time_it([&] { spin_pk<<<blocks, 256>>>(d, iters, a, b); });
    const double fma_per_launch = (double)blocks * 256 * iters * 8;
    printf("{\"all_ms\": %.3f, \"lanes0_31_ms\": %.3f, \"lanes32_63_ms\": %.3f, \"lanes0_15_ms\": %.3f, \"odd_lanes_ms\": %.3f, "
           "\"packed_same_flops_ms\": %.3f, \"scalar_fma_TFLOPs\": %.1f, \"packed_fma_TFLOPs\": %.1f, "
           "\"cycles_per_wave_fma_at_2p4GHz\": %.2f}\n",
        t0, t1, t2, t3, t4, tp, 2 * fma_per_launch / t0 / 1e9, 2 * fma_per_launch / tp / 1e9,
        t0 * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * iters * 8));
    return 0;
}
