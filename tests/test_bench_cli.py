"""bench.py contract pieces that can be checked without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_gpu():
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"],
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0
    assert "needs a GPU" in (proc.stderr + proc.stdout)


def test_grid_size_keeps_the_sphere_inside():
    sys.path.insert(0, ROOT)
    import bench
    for n, dist in ((1_000_000, 0.5), (8_000_000, 0.5), (1_000_000, 0.75)):
        gs = bench.grid_size_for(n, dist)
        radius = (n / 0.64) ** (1 / 3) * dist / 2
        assert gs // 2 - radius >= 2 and gs % 2 == 0
