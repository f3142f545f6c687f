// What a gfx950 SIMD sustains on streams of independent VALU instructions, in SHADER CYCLES
// (s_memtime inside the kernel) and with the clock it ran at (s_memtime ticks against the
// constant 100 MHz wall_clock64) -- not derived from a wall time and an assumed 2.4 GHz.
//
//   valu_probe [iters]
//
// For every instruction kind and 1, 2, 4 and 8 wavefronts per SIMD (256-thread workgroups, one
// wavefront per SIMD each; 256 * W workgroups = every CU holds W of them at once) each
// wavefront times `iters` trips of a 32-instruction block (8 independent chains x 4) and
// the host reports
//   cyc_per_inst_wave   cycles one wavefront needs per instruction (its own issue interval),
//   cyc_per_inst_simd   the same divided by the wavefronts sharing the SIMD = the SIMD's
//                       issue interval per wave64 instruction: the guide's "2 cycles" figure,
//   ghz                 shader clock during the run.
// One JSON object per kind on stdout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

struct Stamp {
    unsigned long long cycles, wall;
};

__device__ __forceinline__ unsigned long long shader_clock()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// kind: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_fma_f16, 3 v_pk_add_f16, 4 v_cmp_gt_f32 + v_addc_co_u32 (counted as 2),
//       5 v_sqrt_f32, 6 v_rcp_f32, 7 v_add_u32, 8 v_cndmask_b32, 9 v_dot2_f32_f16, 10 v_pk_mul_f16,
//       11 v_sub_f32 / v_mul_f32 / v_fmac_f32 mix of the distance test, 12 v_mul_f32
template<int KIND>
__global__ __launch_bounds__(256) void probe(Stamp* out, float* sink, int iters, float a, float b)
{
    float x[8];
    float y[8];
    unsigned m[8];
    for (int k = 0; k < 8; k++) {
        x[k] = threadIdx.x * 0.001f + k;
        y[k] = x[k] * 0.5f;
        m[k] = threadIdx.x + k;
    }
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = shader_clock();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#define ONE(k)                                                                                              \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));                    \
    if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&x[k & 6]) : "v"(*(double*)&y[0]), "v"(*(double*)&y[2])); \
    if (KIND == 2) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));                 \
    if (KIND == 3) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x[k]) : "v"(a));                             \
    if (KIND == 4) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m[k]) : "v"(a), "v"(x[k]) : "vcc"); \
    if (KIND == 5) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[k]));                                            \
    if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k]));                                             \
    if (KIND == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(m[k]) : "v"(m[(k + 1) & 7]));                   \
    if (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(m[k]) : "v"(a) : "vcc");               \
    if (KIND == 9) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));               \
    if (KIND == 10) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x[k]) : "v"(a));                            \
    if (KIND == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
            if (KIND != 11) {
                REP8(ONE)
            } else {
                // the candidate test of grid_force_bits, two candidates: 3 sub, 1 mul, 2 fmac, cmp, addc each
#define TEST(k)                                                                                             \
    asm volatile("v_sub_f32 %0, %3, %0\n\tv_sub_f32 %1, %4, %1\n\tv_sub_f32 %2, %5, %2\n\t"                 \
                 "v_mul_f32 %0, %0, %0\n\tv_fmac_f32 %0, %1, %1\n\tv_fmac_f32 %0, %2, %2\n\t"                \
                 "v_cmp_gt_f32 vcc, %6, %0\n\tv_addc_co_u32 %7, vcc, %7, %7, vcc"                            \
                 : "+v"(x[k]), "+v"(x[k + 1]), "+v"(x[k + 2]), "+v"(y[k]), "+v"(y[k + 1]), "+v"(y[k + 2]),   \
                   "+v"(a), "+v"(m[k])                                                                      \
                 :                                                                                          \
                 : "vcc");
                TEST(0) TEST(3) TEST(0) TEST(3)
            }
        }
    }
    const unsigned long long t1 = shader_clock();
    const unsigned long long w1 = wall_clock64();
    float s = 0;
    for (int k = 0; k < 8; k++) s += x[k] + y[k] + (float)m[k];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = Stamp{t1 - t0, w1 - w0};
}

template<int KIND>
static void run(const char* name, int iters, int insts_per_trip, Stamp* d_out, float* d_sink)
{
    for (int waves = 1; waves <= 8; waves *= 2) {
        const int blocks = 256 * waves;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        probe<KIND><<<blocks, 256>>>(d_out, d_sink, iters / 10, 0.999f, 0.001f);  // warm-up
        (void)hipEventRecord(e0, nullptr);
        probe<KIND><<<blocks, 256>>>(d_out, d_sink, iters, 0.999f, 0.001f);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<Stamp> h(blocks * 4);
        (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (auto& s : h) {
            cyc.push_back((double)s.cycles);
            ghz.push_back((double)s.cycles / ((double)s.wall * 10.0));  // wall ticks are 10 ns
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(ghz.begin(), ghz.end());
        const double n_inst = (double)iters * insts_per_trip;
        const double med = cyc[cyc.size() / 2];
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"insts_per_wave\": %.0f, \"cycles_median\": %.0f, "
               "\"cycles_max\": %.0f, \"cyc_per_inst_wave\": %.3f, \"cyc_per_inst_simd\": %.3f, \"ghz_median\": %.3f, "
               "\"kernel_ms\": %.3f, \"ghz_from_events\": %.3f}\n",
            name, waves, n_inst, med, cyc.back(), med / n_inst, med / n_inst / waves, ghz[ghz.size() / 2], ms,
            cyc.back() / (ms * 1e6));
    }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    Stamp* d_out;
    float* d_sink;
    (void)hipMalloc(&d_out, 256 * 8 * 4 * sizeof(Stamp));
    (void)hipMalloc(&d_sink, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", iters, 32, d_out, d_sink);
    run<12>("v_mul_f32", iters, 32, d_out, d_sink);
    run<7>("v_add_u32", iters, 32, d_out, d_sink);
    run<8>("v_cndmask_b32", iters, 32, d_out, d_sink);
    run<4>("v_cmp_gt_f32+v_addc_co_u32", iters, 64, d_out, d_sink);
    run<11>("distance_test_mix(8 VALU per candidate)", iters, 4 * 4 * 8, d_out, d_sink);
    run<1>("v_pk_fma_f32", iters, 32, d_out, d_sink);
    run<2>("v_pk_fma_f16", iters, 32, d_out, d_sink);
    run<3>("v_pk_add_f16", iters, 32, d_out, d_sink);
    run<10>("v_pk_mul_f16", iters, 32, d_out, d_sink);
    run<9>("v_dot2_f32_f16", iters, 32, d_out, d_sink);
    run<5>("v_sqrt_f32", iters, 32, d_out, d_sink);
    run<6>("v_rcp_f32", iters, 32, d_out, d_sink);
    return 0;
}
