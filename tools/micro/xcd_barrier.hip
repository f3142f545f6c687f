// What does a grid barrier cost when every participant sits on ONE XCD (one L2)?  VERDICT r05 item 4: "sequence
// bin -> scan -> scatter -> order -> force -> reduce -> update with a grid barrier on atomics in that XCD's L2
// (agent scope not needed; ~1 us) ... if the L2-local barrier also costs > 3 us, record the number and close the item".
//
//   xcd_barrier [workgroups per CU] [rounds]
//
// Launches 8 * 32 * per_cu workgroups of 256 threads; workgroup b runs on XCD b % 8 (checked with XCC_ID), and
//   mode "one XCD"   only the workgroups of XCD 0 take part (the others exit at once): arrive = an atomic add
//                    WITHOUT sc1 (workgroup scope: executed in the XCD's own L2), wait = polling with an atomic
//                    add of 0 (read-modify-writes are performed in that L2; plain sc0 loads hit the L1), then
//                    `buffer_inv sc0` (the L1 alone) so that plain loads see what the other CUs of the XCD stored
//                    before the barrier;
//   mode "all XCDs"  every workgroup takes part: agent-scope atomics and loads (past the L2s), agent-scope fences
//                    (write back / invalidate the L2) -- what the one-launch reductions of core.hip paid for.
// Between barriers every workgroup stores a word and reads its left neighbour's (so that the barrier is used as
// one: a stale read is counted).  One JSON line.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            return 1;                                                            \
        }                                                                        \
    } while (0)

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template<bool LOCAL, bool INV_SC1>
__device__ __forceinline__ bool grid_barrier(unsigned* count, unsigned* generation, const unsigned members, unsigned& my_gen,
    unsigned long long* timeouts, const unsigned zero /* 0 the compiler cannot see: `add 0` would become a load */)
{
    __shared__ int gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        my_gen++;
        if (LOCAL) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores have reached the L2
            const unsigned arrived = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (arrived == members - 1) {
                __hip_atomic_exchange(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_exchange(generation, my_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                // (a workgroup-scope LOAD, sc0, may hit the CU's L1 for ever -- first version of this probe: every
                // waiter timed out; a read-modify-write is always performed in the L2)
                for (int spins = 0; __hip_atomic_fetch_add(generation, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != my_gen; spins++) {
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > (1 << 20)) {  // never hang the box: give up and say so
                        atomicAdd(timeouts, 1ULL);
                        gave_up = 1;
                        break;
                    }
                }
            }
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned arrived = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == members - 1) {
                __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(generation, my_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (int spins = 0; __hip_atomic_load(generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_gen; spins++) {
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > (1 << 20)) {
                        atomicAdd(timeouts, 1ULL);
                        gave_up = 1;
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    // what makes the other CUs' plain stores (written through to the shared L2) visible to plain loads here:
    // `buffer_inv sc0` does NOT (measured: every read after it stale -- at workgroup scope the L1 needs no
    // invalidation, so the instruction does nothing); `buffer_inv sc1` (agent scope: the L1, and the L2's
    // non-local lines) does
    if (LOCAL && !INV_SC1) asm volatile("buffer_inv sc0" ::: "memory");
    if (LOCAL && INV_SC1) asm volatile("buffer_inv sc1" ::: "memory");
    return gave_up != 0;
}

template<bool LOCAL, bool INV_SC1>
__global__ __launch_bounds__(256) void k_barriers(unsigned* count, unsigned* generation, unsigned* words, const int per_xcd,
    const int rounds, const unsigned zero, unsigned long long* out /* [0] cycles, [1] wall ticks, [2] stale reads, [3] wrong-XCD blocks, [4] barrier time-outs */)
{
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    if (LOCAL && xcd != 0) return;
    if (threadIdx.x == 0 && xcc_id() != (unsigned)xcd) atomicAdd(&out[3], 1ULL);
    const unsigned members = LOCAL ? per_xcd : 8 * per_xcd;
    const int me = LOCAL ? slot : blockIdx.x, left = (me + members - 1) % members;
    unsigned my_gen = 0;
    unsigned long long c0 = 0, w0 = 0, stale = 0;
    for (int r = 0; r < rounds + 10; r++) {
        if (r == 10 && threadIdx.x == 0) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
            w0 = wall_clock64();
        }
        // a plain store (write-through L1 -> this XCD's L2) before the barrier, a plain load of the neighbour's after it
        if (threadIdx.x == 0) words[me * 32] = (unsigned)r + 1;
        if (grid_barrier<LOCAL, INV_SC1>(count, generation, members, my_gen, &out[4], zero)) break;  // (a time-out: everybody gives up)
        if (threadIdx.x == 0) stale += words[left * 32] != (unsigned)r + 1;
        if (grid_barrier<LOCAL, INV_SC1>(count, generation, members, my_gen, &out[4], zero)) break;  // (a time-out: everybody gives up)
    }
    if (threadIdx.x == 0) {
        unsigned long long c1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
        if (me == 0) {
            out[0] = c1 - c0;
            out[1] = wall_clock64() - w0;
        }
        if (stale) atomicAdd(&out[2], stale);
    }
}

int main(int argc, char** argv)
{
    const int per_cu = argc > 1 ? atoi(argv[1]) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 2000;
    const int per_xcd = 32 * per_cu;
    unsigned *count, *generation, *words;
    unsigned long long* out;
    CHECK(hipMalloc(&count, 256));
    CHECK(hipMalloc(&generation, 256));
    CHECK(hipMalloc(&words, 8 * per_xcd * 32 * sizeof(unsigned)));
    CHECK(hipMalloc(&out, 8 * sizeof(unsigned long long)));
    for (int local = 2; local >= 0; local--) {
        CHECK(hipMemset(count, 0, 256));
        CHECK(hipMemset(generation, 0, 256));
        CHECK(hipMemset(words, 0, 8 * per_xcd * 32 * sizeof(unsigned)));
        CHECK(hipMemset(out, 0, 8 * sizeof(unsigned long long)));
        if (local == 2)
            k_barriers<true, true><<<8 * per_xcd, 256>>>(count, generation + 32, words, per_xcd, rounds, (unsigned)(argc > 5), out);
        else if (local == 1)
            k_barriers<true, false><<<8 * per_xcd, 256>>>(count, generation + 32, words, per_xcd, rounds, (unsigned)(argc > 5), out);
        else
            k_barriers<false, false><<<8 * per_xcd, 256>>>(count, generation + 32, words, per_xcd, rounds, (unsigned)(argc > 5), out);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[5];
        CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        printf("{\"mode\": \"%s\", \"workgroups_in_barrier\": %d, \"per_cu\": %d, \"barriers\": %d, \"us_per_barrier\": %.3f, "
               "\"cycles_per_barrier\": %.0f, \"stale_reads\": %llu, \"blocks_not_on_xcd_b_mod_8\": %llu, \"barrier_timeouts\": %llu}\n",
            local == 2 ? "one XCD: atomics in its L2 (no sc1), polling by atomic add of 0, buffer_inv sc1" :
            local == 1 ? "one XCD: atomics in its L2 (no sc1), polling by atomic add of 0, buffer_inv sc0" : "all XCDs: agent-scope atomics, loads and fences",
            local ? per_xcd : 8 * per_xcd, per_cu, 2 * rounds, (double)h[1] / 100.0 / (2.0 * rounds), (double)h[0] / (2.0 * rounds), h[2], h[3], h[4]);
    }
    return 0;
}
