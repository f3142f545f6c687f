"""BASELINE config 4 (examples/passive_growth.cu) as a reusable test case:
mesenchyme enveloped by a polarized epithelium, growing by cell division.  The set-up itself lives
in yalla_amd/cases.py (bench.py and the tools use it too)."""
from yalla_amd.cases import EPITHELIUM, MESENCHYME, growth_grow as grow, growth_setup as setup  # noqa: F401
