/* yalla_models.h -- C ABI of the model harness.
 *
 * ya||a models are C++ translation units that instantiate
 * Solution<Pt, Solver>::take_step<functor>() (reference examples/ directory).  To
 * drive that template API from Python (tests/, bench.py) without a compiler in
 * the loop, libyalla_models.so instantiates it for a fixed table of named
 * models -- the BASELINE.json configurations and the reference's own test
 * cases -- and exposes each through the handle-based entry points below.
 *
 * The same header is implemented twice from the same model source
 * (yalla_amd/csrc/model_functors.h + models_harness.inc):
 *   - yalla_amd/libyalla_models.so : hipcc, the .cuh headers -> HIP kernels on
 *     the GPU (the product path);
 *   - oracle/_build/liboracle_models.so : g++, oracle/yalla_host.hpp -> the
 *     serial CPU restatement (test infrastructure only).
 * so a parity test makes the identical sequence of calls on both.
 *
 * All functions return 0 on success, a negative value for a harness error
 * (unknown model, unsupported call for this model), or abort the process on a
 * HIP error (the engine fails loudly; there is no fallback).
 */
#ifndef YALLA_MODELS_H
#define YALLA_MODELS_H

#ifdef __cplusplus
extern "C" {
#endif
/* The libraries are built with -fvisibility=hidden; only this C ABI is exported. */
#pragma GCC visibility push(default)

typedef struct ya_sim ya_sim;

/* 1 = HIP engine, 0 = CPU oracle. */
int ya_models_is_device(void);
/* Arithmetic tier of this build: 0 = exact (every statement in IEEE binary32 as written, no
 * contraction, correctly rounded sqrt and reciprocal: bit-comparable with the oracle), 1 = fast
 * (libyalla_models_fast.so: -DYA_ARITH_FAST -ffp-contract=fast, see include/solvers.cuh
 * ya::exact_sqrt; within 1e-5 relative of the exact tier, not bit-identical). */
int ya_models_arith(void);
/* Number of models and their names ("springs_grid", "clipped_tile", ...). */
int ya_models_count(void);
const char* ya_models_name(int index);

/* Construct Solution<Pt, Solver>{n_max[, grid_size, cube_size]} for the named
 * model (solvers.cuh:60-74; grid args are ignored by Tile_solver models). */
int ya_sim_create(const char* model, int n_max, int grid_size, float cube_size, ya_sim** out);
void ya_sim_destroy(ya_sim* sim);

int ya_sim_n_floats(ya_sim* sim); /* floats per point: 3, 4, 5, 7 ... */
int ya_sim_n_max(ya_sim* sim);
float* ya_sim_h_X(ya_sim* sim);   /* host mirror h_X, n_max * n_floats floats */
int ya_sim_set_h_n(ya_sim* sim, int n);
int ya_sim_get_h_n(ya_sim* sim);
int ya_sim_copy_to_device(ya_sim* sim); /* solvers.cuh:80-85 */
int ya_sim_copy_to_host(ya_sim* sim);   /* solvers.cuh:86-91 */
int ya_sim_get_d_n(ya_sim* sim);        /* solvers.cuh:92 */

/* n_steps calls of take_step<pw_int, pw_friction>(dt, gen_forces) followed by
 * the model's post-step kernel if it has one (e.g. proliferation). */
int ya_sim_take_steps(ya_sim* sim, float dt, int n_steps);
/* Block until the device is idle (no-op on the oracle). */
int ya_sim_synchronize(ya_sim* sim);

/* mode 0 = set_fixed(), 1 = set_fixed(point), 2 = set_fixed_xy(point)
 * (solvers.cuh:196-208). */
int ya_sim_set_fixed(ya_sim* sim, int mode, int point);
int ya_sim_set_cube_size(ya_sim* sim, float cube_size); /* public member, :468 */

/* inits.cuh:33-51 with an explicit seed; fills h_X[n_0 .. h_n) and copies to
 * the device. */
int ya_sim_random_sphere(ya_sim* sim, float dist_to_nb, unsigned seed);

/* Copies of device-side state for checks: d_old_v (3 floats per point, n_max
 * points) and, for Grid_solver models, the four public Grid arrays as left by
 * the last build (sizes n_max, n_max, grid_size^3, grid_size^3). */
int ya_sim_get_old_v(ya_sim* sim, float* out);
int ya_sim_set_old_v(ya_sim* sim, const float* in); /* n_max * 3 floats -> d_old_v */
int ya_sim_get_grid(ya_sim* sim, int* cube_id, int* point_id, int* cube_start, int* cube_end);
/* Run Grid::build(points, cube_size) on a fresh public Grid{n_max, grid_size}
 * (tests/test_solvers.cu:247-315) and return its arrays. */
int ya_sim_build_grid(ya_sim* sim, int grid_size, float cube_size, int* cube_id, int* point_id,
    int* cube_start, int* cube_end);

/* Model parameters (e.g. "n_cells" for the sorting model) and the engine's A/B knobs, none of which changes a
 * result: "force_variant", "coop_lanes", "stage_v_max", "tail_tiles" (grid_force_bits: -1 the engine's choice,
 * 0 whole tiles only, k the last k tiles of a launch as half tiles, the launch's tile count or more: all),
 * "sorted_pipeline", "graph", "tile_lanes". */
int ya_sim_set_param(ya_sim* sim, const char* name, double value);
/* Integer per-cell properties of a model ("type", "mes_nbs", "epi_nbs"). */
int ya_sim_set_prop(ya_sim* sim, const char* name, const int* values, int n);
int ya_sim_get_prop(ya_sim* sim, const char* name, int* values, int n);
/* Links of a model that has them: n_links pairs (a, b), then copy_to_device. */
int ya_sim_set_links(ya_sim* sim, const int* ab, int n_links, float strength);

/* ---- z-slab decomposition of a Grid_solver model over ranks.
 * New relative to the reference (single-GPU).  A rank owns the cells with z in [z_lo, z_hi); its
 * local arrays hold the own cells [0, n_own) and then MIRRORED cells of the two slab neighbours
 * (their cells within halo_width of the shared face), which it advances itself with the same
 * update kernels.  One process per rank:
 *   ya_slab_init      this rank's share of the system (the cells in h_X[0 .. h_n) with their
 *                     global ids); pairwise functors are called with GLOBAL ids from then on
 *   ya_slab_setup     rank r of `world` slabs (neighbours r - 1, r + 1), message capacities in
 *                     cells (the same on every rank: ya::slab_plan in include/slab_logic.inc)
 *   a transport       ya_slab_use_rccl(sim, ya_comm*): RCCL send/recv and all-reduce on device
 *                     buffers (include/yalla_hip.h, device build only), or two callbacks (tests:
 *                     gloo, threads of one process, messages staged through the host)
 *   ya_slab_step      one take_step of the decomposed system, this rank's part: per Heun stage
 *                     the grid over own + mirrored cells, the forces of the tiles next to the
 *                     faces, ONE message per neighbour -- the right-hand sides dX of the cells it
 *                     mirrors, at their exact length, travelling beside the forces of all other
 *                     tiles --, the all-reduce of {sum dX, cell count, votes, the fixed point's
 *                     right-hand side} (n_floats + 8 floats: the count in two pieces that stay
 *                     exact under a float sum) and the update of own and mirrored cells; if
 *                     `migrate` -- or if the drift guard asked for it --, the cells that left the
 *                     slab are handed over afterwards and the next step chooses the mirrored
 *                     cells anew.  Between two selections no cell may drift further than
 *                     (halo_width - cube_size) / 2: the guard measures it every step, votes for an
 *                     early selection through the all-reduce, and stops ALL ranks in the same
 *                     step (-11 on the rank whose cell went too far, -12 on the others) if the
 *                     bound was broken all the same.  set_fixed(), set_fixed(i) and
 *                     set_fixed_xy(i) (i: a global id) all work: the owner of cell i contributes
 *                     its right-hand side to the all-reduce.
 *   ya_slab_info      what = 0: selections of the mirrored cells so far, 1: those the drift
 *                     guard asked for, 2: this rank's sticky failure code (0: none), 3 / 4: the guard's
 *                     moved / predicted distances of the last step in millionths of a length unit
 * exchange callback: `kind` 0 = mirrored cells' state and 1 = migrating cells (both fixed
 * capacity, 16-byte header {int count} + rows), 2 = a stage's right-hand sides (bare rows);
 * send_*_bytes from the send buffers to the lower / upper neighbour, recv_*_bytes from them into
 * the recv buffers (a size of 0: no such message); blocking.  allreduce: in-place sum of `count`
 * floats over all ranks.  Both return 0 on success. */
typedef int (*ya_slab_exchange_fn)(void* ctx, int kind, const void* send_lo, long send_lo_bytes, void* recv_lo,
    long recv_lo_bytes, const void* send_hi, long send_hi_bytes, void* recv_hi, long recv_hi_bytes);
typedef int (*ya_slab_allreduce_fn)(void* ctx, float* buf, int count);
int ya_slab_init(ya_sim* sim, float z_lo, float z_hi, float halo_width, const int* global_ids);
int ya_slab_setup(ya_sim* sim, int rank, int world, int halo_cap_cells, int migrate_cap_cells);
int ya_slab_set_transport(ya_sim* sim, ya_slab_exchange_fn exchange, ya_slab_allreduce_fn allreduce, void* ctx);
int ya_slab_use_rccl(ya_sim* sim, void* comm);
int ya_slab_step(ya_sim* sim, float dt, int migrate);
long ya_slab_info(ya_sim* sim, int what);
int ya_slab_n_own(ya_sim* sim);
/* own + mirrored cells as of the last step */
int ya_slab_n_local(ya_sim* sim);
int ya_slab_get_own(ya_sim* sim, float* X_host, int* global_ids_host);
/* ya::slab_plan (include/slab_logic.inc) for the n cells X (n_floats floats each, z the third):
 * cut planes bounds[world + 1], then capacities[4] = {halo_cap, mig_cap, n_max, fullest ghost
 * layer}.  Returns 0, or -9 if an interior slab is thinner than the ghost layer. */
int ya_slab_plan(const float* X, int n_floats, int n, int world, float cube_size, float* bounds, int* capacities);
/* Plan + this rank's share in one call (ya::slab_plan, Slab_grid_solver::slab_adopt): the sim (made
 * with n_max >= the plan's) takes the cells of X (n points of the sim's n_floats floats, the whole
 * system, the same on every rank) whose z lies in rank's slab, in ascending global id, and is
 * initialised and set up for stepping (ya_slab_init + ya_slab_setup).  Returns 0, -9 (a slab
 * thinner than the ghost layer) or -5 (n_max too small). */
int ya_slab_decompose(ya_sim* sim, const float* X, int n, int rank, int world, float cube_size);

/* Device only (test hook): number of binary32 bit patterns in [first, last] for
 * which the engine's correctly rounded square root (ya::exact_sqrt, used for every
 * pair distance) differs from sqrtf.  Returns -1 on the oracle. */
long ya_check_sqrt(unsigned first_bits, unsigned last_bits);
/* The same for the reciprocal behind `Pt / float` (ya::reciprocal) against 1.0f / x. */
long ya_check_reciprocal(unsigned first_bits, unsigned last_bits);

/* Oracle only: 0 = serial COM sum, 1 = the engine's documented tree order.
 * Returns -1 on the device build. */
int ya_sim_set_reduce_order(ya_sim* sim, int order);

/* Device only: accumulate HIP-event timings of the dominant kernel's launches
 * (the force kernel) while enabled -- every `enable`-th launch is timed, 1 = all,
 * 0 = off; read back as total milliseconds and the number of timed launches.
 * Returns -1 on the oracle. */
int ya_sim_profile(ya_sim* sim, int enable);
int ya_sim_profile_read(ya_sim* sim, double* total_ms, int* launches);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* YALLA_MODELS_H */
