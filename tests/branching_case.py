"""BASELINE config 3 (examples/branching.cu) as a reusable test case: a 7-float
Cell {x, y, z, theta, phi, u, v}, Turing kinetics in the self-interaction,
diffusion + bending between epithelial cells, neighbour counters by atomicAdd,
division before every step.  The set-up itself lives in yalla_amd/cases.py."""
from yalla_amd.cases import EPITHELIUM, MESENCHYME, branching_setup as setup  # noqa: F401
