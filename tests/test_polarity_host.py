"""Numeric known-answer tests of the polarity force library (reference
tests/test_polarity.cu:9-17 inverse, :20-34 polarization force, :78-94 bending
force, :160-173 orthonormal, :175-193 migration force; the reference file is
stale and passes a Po_cell where the headers want a Polarity -- SURVEY F3 -- so
the second operand is given as a Polarity).  include/polarity.cuh is
__device__ __host__, so it is checked on the host, compiled against the engine's
dtypes.cuh (hipcc) and against the oracle's header (g++)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include "polarity.cuh"
static bool isclose(float a, float b) { return fabs(a - b) <= 1e-6 + 1e-2 * fabs(b); }
int main() {
    int fails = 0;
    { Po_cell i{0.601, 0.305, 0.320, 0.209, 0.295}; Polarity j{0.340, 0.431};
      auto dF = bidirectional_polarization_force(i, j);
      if (!(isclose(dF.x,0)&&isclose(dF.y,0)&&isclose(dF.z,0)&&isclose(dF.theta,0.126)&&isclose(dF.phi,0.215))) { fails++; printf("polarization force %g %g\n", dF.theta, dF.phi);} }
    { Po_cell i{0.935, 0.675, 0.649, 0.793, 0.073}; Po_cell j{0.566, 0.809, 0.533, 0.297, 0.658};
      auto r = i - j; auto dist = sqrtf(r.x*r.x + r.y*r.y + r.z*r.z); auto dF = bending_force(i, r, dist);
      if (!(isclose(dF.x,0.214)&&isclose(dF.y,-0.971)&&isclose(dF.z,-1.802)&&isclose(dF.theta,-0.339)&&isclose(dF.phi,0.453))) { fails++; printf("bending %g %g %g %g %g\n", dF.x,dF.y,dF.z,dF.theta,dF.phi);} }
    { Po_cell Xi{0}, Xj{0}; Xi.theta = M_PI/2; Xj.x = 1; Xj.y = 1e-3;
      auto Fi = migration_force(Xi, Xi - Xj, 1);
      auto Fj = migration_force(Xj, Xj - Xi, 1);
      if (!(isclose(Fi.x,0.6)&&isclose(Fi.y,-0.8)&&fabs(Fi.z)<5e-5&&isclose(Fi.x,-Fj.x)&&isclose(Fi.y,-Fj.y)&&isclose(Fi.z,-Fj.z))) { fails++; printf("migration\n");} }
    { Polarity pol{1.234f, -2.1f}; auto inv = pt_to_pol(pol_to_float3(pol));
      if (!(isclose(pol.theta, inv.theta) && isclose(pol.phi, inv.phi))) { fails++; printf("inverse\n"); } }
    { float3 r{0.3f,0.8f,0.1f}; float3 p{0.2f,0.5f,0.7f}; p = p / sqrtf(dot_product(p,p)); auto n = orthonormal(r, p);
      if (!(isclose(dot_product(p,n),0) && isclose(dot_product(n,n),1))) { fails++; printf("orthonormal\n"); } }
    { Po_cell i{0.1f, 0.2f, 0.3f, 1.1f, 0.4f}; Po_cell j{0.6f, -0.1f, 0.2f, 0.9f, 0.7f};   // pi/2 == plain bending
      auto r = i - j; auto dist = sqrtf(r.x*r.x + r.y*r.y + r.z*r.z);
      auto a = apical_constriction_force(i, r, dist, (float)(M_PI/2)); auto b = bending_force(i, r, dist);
      if (!(isclose(a.x,b.x)&&isclose(a.y,b.y)&&isclose(a.z,b.z)&&isclose(a.theta,b.theta)&&isclose(a.phi,b.phi))) { fails++; printf("apical\n"); } }
    printf("fails %d\n", fails); return fails;
}
'''


def run(tmp_path, name, compiler, flags, prelude):
    src = tmp_path / (name + (".hip" if "hipcc" in compiler else ".cpp"))
    src.write_text(prelude + SRC)
    exe = tmp_path / name
    subprocess.run([compiler, *flags, str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "fails 0" in out.stdout, out.stdout


def test_polarity_kats_engine_headers(tmp_path):
    run(tmp_path, "pol_engine", "/opt/rocm/bin/hipcc",
        ["--offload-arch=gfx950", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include")],
        '#include "dtypes.cuh"\n')


def test_polarity_kats_oracle_headers(tmp_path):
    run(tmp_path, "pol_oracle", "g++",
        ["-std=c++14", "-O2", "-I" + os.path.join(ROOT, "oracle"), "-I" + os.path.join(ROOT, "include")],
        '#include "yalla_host.hpp"\n')
