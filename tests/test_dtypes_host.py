"""Operator algebra of the point types (reference tests/test_dtypes.cu, 4 cases)
compiled as HOST code twice: against include/dtypes.cuh with hipcc (the operators
are __host__ __device__, so this runs without a GPU) and against the oracle's
header with g++.  Both must print identical bits."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include <string.h>
MAKE_PT(My_float4, w);
static int fails = 0;
#define CHECK(msg, cond) do { if (!(cond)) { printf("FAIL %s\n", msg); fails++; } } while (0)
static bool isclose(float a, float b) { return fabs(a - b) <= 1e-6 + 1e-2 * fabs(b); }
static void bits(const char* tag, float v) { unsigned u; memcpy(&u, &v, 4); printf("%s %08x\n", tag, u); }
int main()
{
    { float3 x{1, 2, 3}; float3 y{5, 4, 3};            // test_float3
      CHECK("+= x", (x += y).x == 1 + 5); CHECK("+= y", (x += y).y == 2 + 4 + 4);
      CHECK("+= z", (x += y).z == 3 + 3 + 3 + 3);
      CHECK("*= x", (y *= 2).x == 5 * 2); CHECK("*= y", (y *= 2).y == 4 * 2 * 2);
      CHECK("*= z", (y *= 2).z == 3 * 2 * 2 * 2); }
    { float4 x{1, 2, 3, 4}; float4 y{5, 4, 3, 2};      // test_float4
      CHECK("4+= w", ((x += y), (x += y), (x += y), (x += y)).w == 4 + 2 + 2 + 2 + 2);
      CHECK("4*= w", ((y *= 2), (y *= 2), (y *= 2), (y *= 2)).w == 2 * 2 * 2 * 2 * 2); }
    { My_float4 x{1, 2, 3, 4}; My_float4 y{5, 4, 3, 2};  // test_make_pt
      CHECK("pt += x", (x += y).x == 1 + 5); CHECK("pt += w", (x += y).w == 4 + 2 + 2);
      CHECK("pt *= w", (y *= 2).w == 2 * 2); CHECK("pt *= x", (y *= 2).x == 5 * 2 * 2); }
    { float3 x{1, 2, 3}; float3 y{4, 3, 2};             // test_generalization
      CHECK("+", (x + y).x == 5 && (x + y).y == 5 && (x + y).z == 5);
      CHECK("-", (x - y).x == -3 && (x - y).y == -1 && (x - y).z == 1);
      CHECK("* r", (x * 3).x == 3 && (x * 3).y == 6 && (x * 3).z == 9);
      CHECK("* l", (3 * x).x == 3 && (3 * x).y == 6 && (3 * x).z == 9);
      CHECK("/", isclose((x / 3).x, 1. / 3) && isclose((x / 3).y, 2. / 3) && isclose((x / 3).z, 1));
      bits("x/3.x", (x / 3).x); bits("x/3.y", (x / 3).y); bits("x/7.z", (x / 7).z);
      bits("x/0.3.x", (x / 0.3f).x);
      CHECK("-= x", (x -= y).x == 1 - 4); CHECK("-= y", (x -= y).y == 2 - 3 - 3); }
    { Po_cell a{1, 2, 3, 0.5f, 0.25f}; Po_cell b{0.1f, 0.2f, 0.3f, 1.5f, 2.5f};
      Po_cell c = -(a - b * 3) / 7 + 2 * a; c -= b; c /= 3;
      bits("po.x", c.x); bits("po.y", c.y); bits("po.z", c.z); bits("po.theta", c.theta); bits("po.phi", c.phi); }
    printf("fails %d\n", fails);
    return fails != 0;
}
'''


def build_and_run(tmp_path, name, compiler, flags, prelude):
    src = tmp_path / (name + (".hip" if "hipcc" in compiler else ".cpp"))
    src.write_text(prelude + SRC)
    exe = tmp_path / name
    subprocess.run([compiler, *flags, str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "fails 0" in out.stdout
    return [l for l in out.stdout.splitlines() if not l.startswith("fails")]


def test_dtypes_algebra_engine_and_oracle_agree(tmp_path):
    engine = build_and_run(tmp_path, "dt_engine", "/opt/rocm/bin/hipcc",
                           ["--offload-arch=gfx950", "-std=c++17", "-O2", "-ffp-contract=off",
                            "-I" + os.path.join(ROOT, "include")],
                           '#include <math.h>\n#include "dtypes.cuh"\n')
    oracle = build_and_run(tmp_path, "dt_oracle", "g++",
                           ["-std=c++14", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "oracle")],
                           '#include "yalla_host.hpp"\n')
    assert engine == oracle
