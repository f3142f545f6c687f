// Rehearsal of north_star's multi-GPU configuration on ONE GPU: a 10 M-cell springs system cut
// into W z-slabs, every slab a Solution<float3, Slab_grid_solver> stepped by its own host thread
// through the NATIVE sequencing (Slab_grid_solver::take_step = slab_step) with a callback
// transport, exactly as W ranks would -- except that the slabs share one GPU (and that a
// callback transport is synchronous: with RCCL the stage's message travels on its own stream
// beside the interior tiles' forces; here the slab simply finishes its launches first).  So that each
// slab's device time is what it would be with a GPU of its own, the threads take turns: a
// thread holds the GPU from the moment a transport call returns until it enters the next one
// (where it drains the device and stops its clock), and the transport itself -- the copies
// between the slabs' message buffers and the sum of the all-reduce -- runs outside everyone's
// clock.  What the clocks hold is therefore the COMPUTE side of a rank's step, host launch
// latencies included, transport excluded.
//
//   slab_rehearsal [cells] [world] [steps] [warmup] [migrate_every]
//
// Output: one JSON object -- the undivided system's ms/step on the same GPU, per slab n_own /
// n_ghost / ms per step and per segment, the critical path (per segment the slowest slab: the
// exchange and the all-reduce are synchronisation points) and
// projected_speedup_compute_only = undivided / critical path.  A PROJECTION: RCCL latency and
// xGMI transfer times are not in it (tools/comm_probe measures what one rank can of those).
#include "../include/dtypes.cuh"
#include "../include/inits.cuh"
#include "../include/slab.cuh"
#include "../include/solvers.cuh"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

__device__ float3 spring(float3 Xi, float3 r, float dist, int i, int j)  // examples/springs.cu:14-21
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j) return dF;
    return r * (0.5f - dist) / dist;
}
YA_STATELESS(float3, spring)  // as the bench's model says of the same functor (yalla_amd/csrc/model_functors.h)

// One empty kernel per rank: launched whenever a slab takes the GPU, so that a kernel trace of
// this program (rocprofv3 --kernel-trace) can be cut into the slabs' segments
// (tools/slab_trace_summary.py: device-busy time per slab and step).
template<int RANK>
__global__ void slab_takes_the_gpu() {}
// ... and one when it hands the GPU back: what follows until the next slab's marker is the REHEARSAL's
// transport (messages as device-to-device copies, the all-reduce through the host), not the slab's work
__global__ void slab_hands_the_gpu_back() {}
static void mark(int rank)
{
    switch (rank & 7) {
        case 0: slab_takes_the_gpu<0><<<1, 1>>>(); break;
        case 1: slab_takes_the_gpu<1><<<1, 1>>>(); break;
        case 2: slab_takes_the_gpu<2><<<1, 1>>>(); break;
        case 3: slab_takes_the_gpu<3><<<1, 1>>>(); break;
        case 4: slab_takes_the_gpu<4><<<1, 1>>>(); break;
        case 5: slab_takes_the_gpu<5><<<1, 1>>>(); break;
        case 6: slab_takes_the_gpu<6><<<1, 1>>>(); break;
        default: slab_takes_the_gpu<7><<<1, 1>>>(); break;
    }
}

// YALLA_REHEARSAL_GENERIC=1: every take_step (undivided and slabs) gets a generic force as well -- each
// cell pulled towards the origin, no ids needed --, which sends the decomposed step through d_X1 and
// the plain update kernels instead of the sorted-copy predictor and the raw corrector (a parity
// check of that path over several slabs; not a timing configuration).
__global__ void pull_to_origin(const int n, const float3* __restrict__ d_X, float3* d_dX)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d_dX[i].x -= 0.3f * d_X[i].x;
    d_dX[i].y -= 0.3f * d_X[i].y;
    d_dX[i].z -= 0.3f * d_X[i].z;
}
void origin_forces(const int n, const float3* __restrict__ d_X, float3* d_dX)
{
    pull_to_origin<<<(n + 255) / 256, 256>>>(n, d_X, d_dX);
}
static bool generic_forces = false;
template<typename Cells>
void one_step(Cells& cells, float dt)
{
    if (generic_forces)
        cells.template take_step<spring>(dt, origin_forces);
    else
        cells.template take_step<spring>(dt);
}

using Clock = std::chrono::steady_clock;
using Slab = Solution<float3, Slab_grid_solver>;
// segments of a step (a new one begins whenever a transport call returns): [re-halo: pack |
// unpack +] build + forces 1 | sum 1 | update 1 + build + forces 2 | sum 2 | update 2 (+ migrate pack) | migrate unpack
constexpr int SEGMENTS = 10;

struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0, generation = 0, parties;
    explicit Barrier(int n) : parties(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lock(m);
        const int gen = generation;
        if (++waiting == parties) {
            waiting = 0;
            generation++;
            cv.notify_all();
        } else
            cv.wait(lock, [&] { return gen != generation; });
    }
};

struct Shared {
    int world;
    Barrier barrier;
    std::mutex gpu;  // whoever holds it has the device to itself
    // YALLA_REHEARSAL_ROTATE=1: the order in which the slabs take the GPU inside a round (a round = every
    // slab's stretch between two transport calls) rotates from round to round -- slab (k + r) % world is
    // the r-th of round k -- instead of being left to the mutex.  (Built to test whether the last slab's
    // dearer force launches came from running last on a throttled chip.  They did not: same times.)
    std::condition_variable turn_cv;
    long next_slot = 0;
    bool rotate = false;
    std::vector<const void*> send_lo, send_hi;
    std::vector<float*> sums;
    bool timing = false;
    bool markers = false;  // YALLA_REHEARSAL_MARKERS=1: a marker kernel per segment (for traces; costs a launch)
    explicit Shared(int w) : world(w), barrier(w), send_lo(w), send_hi(w), sums(w) {}
};

struct Rank {
    Shared* shared;
    int rank;
    Clock::time_point started;
    int segment = 0;
    double seconds[SEGMENTS] = {0};
    std::vector<double> per_step[SEGMENTS];  // the same, one entry per timed step (medians: the host's
                                             // eight threads share this box's cores with everything else)
    long message_bytes = 0;  // right-hand-side messages sent during the timed steps
    int step_index = 0;      // timed steps taken so far
    long starts = 0;  // stretches begun so far (the same on every slab at the same point of a step)
    void stop()  // the device is drained, the clock stopped, the GPU handed on
    {
        (void)hipDeviceSynchronize();
        if (shared->timing && segment < SEGMENTS) {
            const double span = std::chrono::duration<double>(Clock::now() - started).count();
            seconds[segment] += span;
            if ((int)per_step[segment].size() < step_index + 1) per_step[segment].resize(step_index + 1, 0.);
            per_step[segment][step_index] += span;
        }
        segment++;
        if (shared->markers) {
            slab_hands_the_gpu_back<<<1, 1>>>();
            (void)hipDeviceSynchronize();
        }
        if (shared->rotate) {
            {
                std::lock_guard<std::mutex> lock(shared->gpu);
                shared->next_slot++;
            }
            shared->turn_cv.notify_all();
        } else
            shared->gpu.unlock();
    }
    void start()
    {
        if (shared->rotate) {
            const int world = shared->world;
            const long round = starts++;
            const long slot = round * world + ((rank - round % world) % world + world) % world;
            std::unique_lock<std::mutex> lock(shared->gpu);
            shared->turn_cv.wait(lock, [&] { return shared->next_slot == slot; });
        } else
            shared->gpu.lock();
        if (shared->markers) mark(rank);
        started = Clock::now();
    }
};

static int exchange_cb(void* ctx, int kind, const void* send_lo, long send_lo_bytes, void* recv_lo, long recv_lo_bytes,
    const void* send_hi, long send_hi_bytes, void* recv_hi, long recv_hi_bytes)
{
    Rank& me = *(Rank*)ctx;
    Shared& sh = *me.shared;
    me.stop();
    sh.send_lo[me.rank] = send_lo;
    sh.send_hi[me.rank] = send_hi;
    if (sh.timing && kind == 2) me.message_bytes += send_lo_bytes + send_hi_bytes;
    sh.barrier.wait();
    // what RCCL send/recv would move over xGMI: device-to-device copies, outside the clocks
    if (recv_lo_bytes > 0 && me.rank > 0)
        (void)hipMemcpy(recv_lo, sh.send_hi[me.rank - 1], (size_t)recv_lo_bytes, hipMemcpyDeviceToDevice);
    if (recv_hi_bytes > 0 && me.rank + 1 < sh.world)
        (void)hipMemcpy(recv_hi, sh.send_lo[me.rank + 1], (size_t)recv_hi_bytes, hipMemcpyDeviceToDevice);
    (void)hipDeviceSynchronize();
    sh.barrier.wait();
    me.start();
    return 0;
}

static int allreduce_cb(void* ctx, float* buf, int count)
{
    Rank& me = *(Rank*)ctx;
    Shared& sh = *me.shared;
    me.stop();
    sh.sums[me.rank] = buf;
    sh.barrier.wait();
    if (me.rank == 0) {
        std::vector<float> total(count, 0.f), part(count);
        for (int r = 0; r < sh.world; r++) {
            (void)hipMemcpy(part.data(), sh.sums[r], count * sizeof(float), hipMemcpyDeviceToHost);
            for (int k = 0; k < count; k++) total[k] += part[k];
        }
        for (int r = 0; r < sh.world; r++)
            (void)hipMemcpy(sh.sums[r], total.data(), count * sizeof(float), hipMemcpyHostToDevice);
    }
    sh.barrier.wait();
    me.start();
    return 0;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000000;
    const int world = argc > 2 ? atoi(argv[2]) : 8;
    const int steps = argc > 3 ? atoi(argv[3]) : 16;
    const int warmup = argc > 4 ? atoi(argv[4]) : 3;
    const int migrate_every = argc > 5 ? atoi(argv[5]) : 16;
    const float dt = 0.001f, dist = 0.5f;
    generic_forces = getenv("YALLA_REHEARSAL_GENERIC") != nullptr;
    const float radius = powf(n / 0.64f, 1.f / 3) * dist / 2;
    const int gs = std::max(2 * ((int)radius + 3), 8);

    // the undivided system, same GPU, same step count
    std::vector<float3> X0((size_t)n), X_whole((size_t)n);
    double whole_ms;
    {
        Solution<float3, Grid_solver> whole{n, gs, 1.f};
        if (getenv("YALLA_SUM_ORDER") && atoi(getenv("YALLA_SUM_ORDER")) == 1) whole.sum_order = YA_SUM_BY_PLANE;  // (opt-in order: half tiles)
        random_sphere(dist, whole, 0, 42);
        std::copy(whole.h_X, whole.h_X + n, X0.begin());
        for (int s = 0; s < warmup; s++) one_step(whole, dt);
        (void)hipDeviceSynchronize();
        const auto t0 = Clock::now();
        for (int s = 0; s < steps; s++) one_step(whole, dt);
        (void)hipDeviceSynchronize();
        whole_ms = std::chrono::duration<double>(Clock::now() - t0).count() / steps * 1e3;
        whole.copy_to_host();  // after warmup + steps take_steps: what the slabs must reproduce
        std::copy(whole.h_X, whole.h_X + n, X_whole.begin());
    }

    const ya::Slab_plan plan = ya::slab_plan(X0.data(), n, world, 1.f);
    if (plan.error) {
        printf("{\"error\": \"a slab is thinner than the ghost layer\", \"cells\": %d, \"world\": %d}\n", n, world);
        return 1;
    }
    Shared shared{world};
    shared.markers = getenv("YALLA_REHEARSAL_MARKERS") != nullptr;
    shared.rotate = getenv("YALLA_REHEARSAL_ROTATE") != nullptr;
    std::vector<std::unique_ptr<Slab>> slabs;
    std::vector<Rank> ranks(world);
    // One stream for every slab's interior launch (they take turns on the GPU anyway): a rank with a
    // GPU to itself has three streams, but W slabs' own streams in this one process would share the
    // device's four hardware queues, and a slab whose interior stream lands on the default stream's
    // queue runs its two launches of a stage one after the other (YALLA_REHEARSAL_OWN_STREAMS=1: as
    // rounds' earlier runs did; one slab in eight then is 5-10 % slower).
    hipStream_t interior = nullptr;
    if (!getenv("YALLA_REHEARSAL_OWN_STREAMS")) YA_CHECK((int)hipStreamCreateWithFlags(&interior, hipStreamNonBlocking));
    for (int r = 0; r < world; r++) {
        slabs.emplace_back(new Slab{plan.n_max, gs, 1.f});
        Slab& s = *slabs.back();
        if (interior) s.slab_use_interior_stream(interior);
        if (s.slab_adopt(plan, r, X0.data(), n, s.h_X, s.h_n) != 0) return 2;
        s.d_global_id = nullptr;  // spring only compares i with j (as bench.py runs it)
        if (getenv("YALLA_SUM_ORDER") && atoi(getenv("YALLA_SUM_ORDER")) == 1) s.sum_order = YA_SUM_BY_PLANE;
        ranks[r].shared = &shared;
        ranks[r].rank = r;
        s.slab_set_transport(exchange_cb, allreduce_cb, &ranks[r]);
        s.migrate_every = migrate_every;
        if (const char* v = getenv("YALLA_SLAB_LOCAL_ORDER")) s.slab.local_order_every = atoi(v);  // A/B: 0 = never
    }
    X0.clear();
    X0.shrink_to_fit();

    std::vector<int> ghosts(world, 0), owns(world, 0);
    std::vector<long> selections(world, 0), guard_requests(world, 0);
    std::vector<float> guard_moved(world, 0.f), guard_predicted(world, 0.f);
    // every cell's position after the run, by global id, for the comparison with the undivided system
    std::vector<float3> X_slabs((size_t)n, float3{NAN, NAN, NAN});
    std::mutex collect;
    auto run = [&](int r) {
        Slab& s = *slabs[r];
        Rank& me = ranks[r];
        for (int k = 0; k < warmup + steps; k++) {
            if (k == warmup) {
                shared.barrier.wait();
                if (r == 0) shared.timing = true;
                shared.barrier.wait();
            }
            me.segment = 0;
            me.start();
            one_step(s, dt);
            me.stop();
            if (shared.timing) me.step_index++;
            shared.barrier.wait();
        }
        ghosts[r] = s.slab.ghosts[0] + s.slab.ghosts[1];
        owns[r] = s.slab.n_own;
        selections[r] = s.slab.rehalos;
        guard_requests[r] = s.slab.guard_requests;
        if (s.slab.d_guard) {
            float state[4] = {0, 0, 0, 0};
            (void)hipMemcpy(state, s.slab.d_guard, sizeof(state), hipMemcpyDeviceToHost);
            guard_moved[r] = state[0];
            guard_predicted[r] = state[1];
        }
        std::lock_guard<std::mutex> lock(collect);
        std::vector<float> X((size_t)3 * plan.n_max);
        std::vector<int> gid((size_t)plan.n_max);
        const int own = s.get_own(X.data(), gid.data());
        for (int k = 0; k < own; k++) X_slabs[gid[k]] = float3{X[3 * k], X[3 * k + 1], X[3 * k + 2]};
    };
    if (world == 1) {
        // a world of one never calls its transport: the whole step is one segment
        run(0);
    } else {
        std::vector<std::thread> threads;
        for (int r = 0; r < world; r++) threads.emplace_back(run, r);
        for (auto& t : threads) t.join();
    }

    // slabs against the undivided system: 1e-5 of the system's extent, except for the odd pair of
    // cells whose distance is within rounding of the cut-off (see tests/fuzz_slab.py)
    double scale = 0, max_diff = 0;
    long beyond = 0, missing = 0;
    for (int i = 0; i < n; i++)
        scale = std::max({scale, (double)std::fabs(X_whole[i].x), (double)std::fabs(X_whole[i].y), (double)std::fabs(X_whole[i].z)});
    for (int i = 0; i < n; i++) {
        if (X_slabs[i].x != X_slabs[i].x) {
            missing++;
            continue;
        }
        const double d = std::max({std::fabs((double)X_slabs[i].x - X_whole[i].x), std::fabs((double)X_slabs[i].y - X_whole[i].y),
            std::fabs((double)X_slabs[i].z - X_whole[i].z)});
        max_diff = std::max(max_diff, d);
        beyond += d > 1e-5 * scale;
    }

    // per slab and segment: mean over the timed steps and median (a step in which the host thread was
    // descheduled counts once, not with its full length); the critical path is built from the medians
    auto median_of = [](std::vector<double> v) {
        if (v.empty()) return 0.;
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    };
    double critical = 0, slowest_slab = 0, critical_mean = 0;
    double seg_max[SEGMENTS] = {0};
    std::vector<double> slab_median(world, 0.);
    for (int g = 0; g < SEGMENTS; g++) {
        double mean_max = 0;
        for (int r = 0; r < world; r++) {
            const double med = median_of(ranks[r].per_step[g]);
            seg_max[g] = std::max(seg_max[g], med * steps);
            slab_median[r] += med;
            mean_max = std::max(mean_max, ranks[r].seconds[g]);
        }
        critical += seg_max[g];
        critical_mean += mean_max;
    }
    printf("{\"cells\": %d, \"world\": %d, \"grid_size\": %d, \"steps\": %d, \"warmup\": %d, \"migrate_every\": %d, \"cuts\": \"%s\", "
           "\"sequencing\": \"native (Slab_grid_solver::take_step), one host thread per slab, slabs take turns on the GPU\", "
           "\"undivided_ms_per_step\": %.4f, \"halo_cap\": %d, \"slabs\": [",
        n, world, gs, steps, warmup, migrate_every,
        getenv("YALLA_SLAB_PLAN") && getenv("YALLA_SLAB_PLAN")[0] == 'q' ? "z-quantiles (equal own cells)" : "own + 0.26 mirrored cells balanced",
        whole_ms, plan.halo_cap);
    long total_own = 0;
    for (int r = 0; r < world; r++) {
        double sum = 0;
        for (int g = 0; g < SEGMENTS; g++) sum += ranks[r].seconds[g];
        slowest_slab = std::max(slowest_slab, slab_median[r] * steps);
        total_own += owns[r];
        printf("%s{\"rank\": %d, \"n_own\": %d, \"n_ghost\": %d, \"ms_per_step_median\": %.4f, \"ms_per_step_mean\": %.4f, "
               "\"rhs_message_bytes_per_stage\": %.0f, \"selections_of_mirrored_cells\": %ld, \"asked_for_by_drift_guard\": %ld, "
               "\"guard_moved\": %.4f, \"guard_predicted\": %.4f, \"segments_ms_mean\": [", r ? ", " : "", r, owns[r], ghosts[r],
            slab_median[r] * 1e3, sum / steps * 1e3, (double)ranks[r].message_bytes / (2.0 * steps), selections[r], guard_requests[r],
            guard_moved[r], guard_predicted[r]);
        for (int g = 0; g < SEGMENTS; g++) printf("%s%.4f", g ? ", " : "", ranks[r].seconds[g] / steps * 1e3);
        printf("]}");
    }
    printf("], \"segment_note\": \"a segment ends at every transport call: per stage 'build + both force launches + "
           "packing the neighbours\' rows', 'join + reduction', then 'update' joins the next stage\'s first segment; steps that "
           "begin with a re-halo or end with a migration have two more each\", \"segment_max_ms\": [");
    for (int g = 0; g < SEGMENTS; g++) printf("%s%.4f", g ? ", " : "", seg_max[g] / steps * 1e3);
    printf("], \"parity\": {\"take_steps\": %d, \"cells_missing\": %ld, \"cells_beyond_1e-5\": %ld, \"max_abs_diff\": %.3g, "
           "\"scale\": %.4g}", warmup + steps, missing, beyond, max_diff, scale);
    printf(", \"cells_after\": %ld, \"slowest_slab_ms_per_step\": %.4f, \"critical_path_ms_per_step\": %.4f, "
           "\"critical_path_ms_per_step_from_means\": %.4f, \"projected_speedup_compute_only\": %.3f, "
           "\"projected_speedup_compute_only_median_step\": %.3f, "
           "\"note\": \"projections: compute side only, no RCCL latency, no xGMI transfer time; the first from the means over the timed "
           "steps (re-selection of the mirrored cells and migration every migrate_every-th step included, and whatever the host "
           "threads lost to the box), the second from the per-segment medians (the common step)\"}\n",
        total_own, slowest_slab / steps * 1e3, critical / steps * 1e3, critical_mean / steps * 1e3,
        whole_ms / (critical_mean / steps * 1e3), whole_ms / (critical / steps * 1e3));
    return total_own == n ? 0 : 3;
}
