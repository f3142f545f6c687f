// Micro-benchmark: 19 small dependent kernels per "step", launched one by one on the
// null stream vs replayed as one hipGraph (stream capture).  Prints us per step.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(float* p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
int main()
{
    const int n = 10000, K = 19, STEPS = 2000;
    float* d;
    hipMalloc(&d, n * sizeof(float));
    hipMemset(d, 0, n * sizeof(float));
    int h_n = 0, *d_n;
    hipMalloc(&d_n, 4);
    hipMemset(d_n, 0, 4);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int sync = 0; sync < 2; sync++) {
        // direct launches
        auto t0 = now();
        for (int s = 0; s < STEPS; s++) {
            if (sync) hipMemcpy(&h_n, d_n, 4, hipMemcpyDeviceToHost);
            for (int k = 0; k < K; k++) touch<<<(n + 255) / 256, 256>>>(d, n);
        }
        hipDeviceSynchronize();
        printf("direct  sync_read=%d: %.1f us/step\n", sync, us(t0, now()) / STEPS);
        // graph
        hipStream_t st;
        hipStreamCreate(&st);
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < K; k++) touch<<<(n + 255) / 256, 256, 0, st>>>(d, n);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int where = 0; where < 2; where++) {
            hipStream_t ls = where ? st : nullptr;
            hipGraphLaunch(ge, ls);
            hipDeviceSynchronize();
            t0 = now();
            for (int s = 0; s < STEPS; s++) {
                if (sync) hipMemcpy(&h_n, d_n, 4, hipMemcpyDeviceToHost);
                hipGraphLaunch(ge, ls);
            }
            hipDeviceSynchronize();
            printf("graph   sync_read=%d stream=%s: %.1f us/step\n", sync, where ? "own" : "null", us(t0, now()) / STEPS);
        }
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
        hipStreamDestroy(st);
    }
    return 0;
}
