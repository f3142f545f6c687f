"""A THIRD, independent statement of the config-3 and config-4 pairwise functors and of the
polarity forces they call, written in numpy float64 straight from the reference
(examples/passive_growth.cu:29-57, examples/branching.cu:21-110, include/polarity.cuh:13-94) -- not from yalla_amd/csrc/model_functors.h or include/polarity.cuh,
which the device build and the CPU oracle share.  One Heun step of a small all-pairs system
(Tile_solver, zero old velocities, so the friction term vanishes in both stages) must agree with
both backends: a transcription slip in the shared model source would show here and nowhere else.
Also the neighbour counters the functor keeps per cell (exact integers)."""
import numpy as np
import pytest

from yalla_amd.solution import Solution

MESENCHYME, EPITHELIUM = 0, 1
R_MAX = 1.0  # passive_growth.cu:14


def pol_to_vec(theta, phi):  # polarity.cuh:13-21
    return np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])


def unidirectional_polarization_force(theta_i, phi_i, p_theta, p_phi):  # polarity.cuh:50-60
    d_theta = np.cos(theta_i) * np.sin(p_theta) * np.cos(phi_i - p_phi) - np.sin(theta_i) * np.cos(p_theta)
    d_phi = 0.0
    if abs(np.sin(theta_i)) > 1e-10:
        d_phi = -np.sin(p_theta) * np.sin(phi_i - p_phi) / np.sin(theta_i)
    return d_theta, d_phi


def bending_force(Xi, r, dist):  # polarity.cuh:72-94, Xi and r = (x, y, z, theta, phi)
    pi = pol_to_vec(Xi[3], Xi[4])
    prodi = pi.dot(r[:3]) / dist
    r_hat = (np.arccos(r[2] / dist), np.arctan2(r[1], r[0]))  # pt_to_pol, :23-28
    u_theta, u_phi = unidirectional_polarization_force(Xi[3], Xi[4], *r_hat)
    dF = np.zeros(5)
    dF[3], dF[4] = -prodi * u_theta, -prodi * u_phi
    dF[:3] = -prodi / dist * pi + prodi ** 2 / dist ** 2 * r[:3]
    pj = pol_to_vec(Xi[3] - r[3], Xi[4] - r[4])  # the neighbour's polarity
    prodj = pj.dot(r[:3]) / dist
    dF[:3] += -prodj / dist * pj + prodj ** 2 / dist ** 2 * r[:3]
    return dF


def relu_w_epithelium(Xi, r, dist, i, j, types, mes_nbs, epi_nbs):  # passive_growth.cu:29-57
    dF = np.zeros(5)
    if i == j or dist > R_MAX:
        return dF
    if types[i] == types[j]:
        F = max(0.7 - dist, 0) * 2 - max(dist - 0.8, 0)
    else:
        F = max(0.8 - dist, 0) * 2 - max(dist - 0.9, 0)
    dF[:3] = r[:3] * F / dist
    if types[j] == MESENCHYME:
        mes_nbs[i] += 1
    else:
        epi_nbs[i] += 1
    if types[i] == MESENCHYME or types[j] == MESENCHYME:
        return dF
    return dF + bending_force(Xi, r, dist) * 0.15


# examples/branching.cu:21-30
LAMBDA, D_U, D_V, F_V, F_U, G_U, M_U, M_V, S_U = 0.0075, 0.001, 0.2, 1.0, 80.0, 80.0, 0.25, 0.75, 0.05


def epi_turing_mes_noturing(Xi, r, dist, i, j, types, mes_nbs, epi_nbs):  # branching.cu:60-110
    """Xi and r = (x, y, z, theta, phi, u, v)."""
    dF = np.zeros(7)
    if i == j:  # Meinhardt kinetics in the self-interaction
        if types[i] == EPITHELIUM:
            dF[5] = LAMBDA * ((F_U * Xi[5] * Xi[5]) / (1 + F_V * Xi[6]) - M_U * Xi[5] + S_U)
            dF[6] = LAMBDA * (G_U * Xi[5] * Xi[5] - M_V * Xi[6])
            if -dF[5] > Xi[5]:
                dF[5] = 0.0
            if -dF[6] > Xi[6]:
                dF[6] = 0.0
        return dF
    if dist > R_MAX:
        return dF
    if types[i] == types[j]:
        F = max(0.7 - dist, 0) * 2 - max(dist - 0.8, 0)
    else:
        F = max(0.8 - dist, 0) * 2 - max(dist - 0.9, 0)
    dF[:3] = r[:3] * F / dist
    if types[i] == EPITHELIUM and types[j] == EPITHELIUM:  # diffusion + bending
        dF[5] = -D_U * r[5]
        dF[6] = -D_V * r[6]
        if -dF[5] > Xi[5]:
            dF[5] = 0.0
        if -dF[6] > Xi[6]:
            dF[6] = 0.0
        dF[:5] += bending_force(Xi[:5], r[:5], dist) * 0.2
    else:
        dF[6] = -D_V * r[6]  # v diffuses into the mesenchyme
    if types[j] == EPITHELIUM:
        epi_nbs[i] += 1
    else:
        mes_nbs[i] += 1
    return dF


def rhs(functor, X, types, mes_nbs, epi_nbs, cut_off=None):
    """solvers.cuh:284-322 (all pairs; :450 `dist >= cube_size -> skip` on the grid) and :240-249
    (centre of mass held fixed), with old_v = 0."""
    n = len(X)
    dX = np.zeros_like(X)
    for i in range(n):
        for j in range(n):
            r = X[i] - X[j]
            dist = np.sqrt(r[0] ** 2 + r[1] ** 2 + r[2] ** 2)
            if cut_off is not None and dist >= cut_off:
                continue
            dX[i] += functor(X[i], r, dist, i, j, types, mes_nbs, epi_nbs)
    dX[:, :3] -= dX[:, :3].mean(axis=0)
    return dX


def heun_step(functor, X, types, dt, cut_off=None):
    mes, epi = np.zeros(len(X), int), np.zeros(len(X), int)  # reset_nbs zeroes them before every stage
    k1 = rhs(functor, X, types, mes, epi, cut_off)
    mes[:], epi[:] = 0, 0
    k2 = rhs(functor, X + dt * k1, types, mes, epi, cut_off)
    return X + 0.5 * dt * (k1 + k2), mes, epi


def case(n=70, seed=4):
    rng = np.random.default_rng(seed)
    X = np.zeros((n, 5))
    # a loose ball: most neighbours within r_max, a shell of polarized epithelium around a core
    X[:, :3] = rng.normal(size=(n, 3)) * 0.6
    radius = np.linalg.norm(X[:, :3], axis=1)
    types = (radius > np.median(radius)).astype(np.int32)
    outward = X[:, :3] / radius[:, None]
    X[:, 3] = np.where(types == EPITHELIUM, np.arccos(outward[:, 2]) + rng.normal(size=n) * 0.2, 0)
    X[:, 4] = np.where(types == EPITHELIUM, np.arctan2(outward[:, 1], outward[:, 0]) + rng.normal(size=n) * 0.2, 0)
    return X.astype(np.float32), types


def run_backend(lib, X0, types, dt, model="passive_growth_tile"):
    n = len(X0)
    with Solution(model, n, 50, 1.0, lib=lib) as s:
        if lib.ya_models_is_device() == 0:
            s.set_reduce_order(1)
        s.h_n = n
        s.h_X[:n] = X0
        s.copy_to_device()
        s.set_prop("type", types)
        s.set_prop("mes_nbs", np.zeros(n, np.int32))
        s.set_prop("epi_nbs", np.zeros(n, np.int32))
        s.take_step(dt)
        return s.positions(), s.get_prop("mes_nbs", n), s.get_prop("epi_nbs", n)


def check(lib):
    X0, types = case()
    dt = 0.05
    X_ref, mes_ref, epi_ref = heun_step(relu_w_epithelium, X0.astype(np.float64), types, dt)
    X, mes, epi = run_backend(lib, X0, types, dt)
    assert (mes_ref + epi_ref).sum() > 8 * len(X0), "test too sparse: hardly any neighbours"
    assert (types == EPITHELIUM).sum() > 10 and np.abs(X_ref[:, 3:] - X0[:, 3:]).max() > 1e-4, \
        "the polarity forces never acted"
    # the counters hold the second stage's counts (reset_nbs runs before every stage)
    assert np.array_equal(mes, mes_ref) and np.array_equal(epi, epi_ref)
    moved = np.abs(X_ref - X0).max()
    assert np.abs(X[:, :3] - X_ref[:, :3]).max() <= 2e-5 * max(np.abs(X_ref[:, :3]).max(), 1.0)
    assert np.abs(X[:, 3:] - X_ref[:, 3:]).max() <= 1e-4 * max(moved, 1e-3) + 2e-6


def check_branching(lib):
    """Config 3's functor on the grid (cube_size = r_max = 1): Turing kinetics in the self term,
    diffusion of u and v, bending between epithelial cells, integer counters by atomicAdd."""
    X5, types = case(n=80, seed=6)
    rng = np.random.default_rng(8)
    X0 = np.zeros((len(X5), 7), np.float32)
    X0[:, :5] = X5
    X0[:, 5] = np.where(types == EPITHELIUM, rng.random(len(X5)) * 0.2, 0)       # u: epithelium only
    X0[:, 6] = rng.random(len(X5)) * 0.3
    dt = 0.05
    X_ref, mes_ref, epi_ref = heun_step(epi_turing_mes_noturing, X0.astype(np.float64), types, dt, cut_off=1.0)
    X, mes, epi = run_backend(lib, X0, types, dt, model="branching_grid")
    assert (mes_ref + epi_ref).sum() > 8 * len(X0)
    assert np.abs(X_ref[:, 5:] - X0[:, 5:]).max() > 1e-5, "the kinetics never acted"
    assert np.array_equal(mes, mes_ref) and np.array_equal(epi, epi_ref)
    assert np.abs(X[:, :3] - X_ref[:, :3]).max() <= 2e-5 * max(np.abs(X_ref[:, :3]).max(), 1.0)
    for cols in (slice(3, 5), slice(5, 7)):
        moved = np.abs(X_ref[:, cols] - X0[:, cols]).max()
        assert np.abs(X[:, cols] - X_ref[:, cols]).max() <= 1e-4 * max(moved, 1e-3) + 2e-6, cols


def test_oracle_against_the_independent_statement(oracle):
    check(oracle)
    check_branching(oracle)


@pytest.mark.gpu
def test_device_against_the_independent_statement(device):
    check(device)
    check_branching(device)
