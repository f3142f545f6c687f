"""The oracle restates the header API's round-5 additions too -- `Solution_n<Pt, n_max, Solver>` and
`keep_in_cube_order(every, arrays...)` -- so that model programs written against them can be held to it.  A small
model compiled with g++ against oracle/yalla_host.hpp (no GPU): registered once against renumbering by hand."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "yalla_host.hpp"
#include <stdio.h>
#include <string.h>
#include <vector>

static int* d_type;
static float3 typed_spring(float3 Xi, float3 r, float dist, int i, int j)
{
    float3 dF{0.f, 0.f, 0.f};
    if (i == j || dist >= 1.f) return dF;
    const float k = d_type[i] == d_type[j] ? 2.f : 1.f;
    return r * (k * (0.6f - dist) / dist);
}
struct Prop { int* d_prop; };   // anything with a d_prop member (Property<T>)

template<typename Cells>
std::vector<float3> run(Cells& cells, bool registered, std::vector<int>& type_out)
{
    const int n = cells.n_max;
    for (int i = 0; i < n; i++)   // a loose ball, deterministic
        cells.h_X[i] = float3{0.37f * (i % 11) - 2.f, 0.41f * ((i / 11) % 9) - 1.6f, 0.29f * (i / 99) - 1.f};
    cells.copy_to_device();
    std::vector<int> type(n);
    for (int i = 0; i < n; i++) type[i] = (i * 7) % 3;
    Prop prop{type.data()};
    d_type = prop.d_prop;
    if (registered) cells.keep_in_cube_order(2, prop);
    for (int step = 0; step < 7; step++) {
        if (!registered && step % 2 == 0) cells.renumber(prop);
        cells.template take_step<typed_spring>(0.02f);
    }
    cells.copy_to_host();
    type_out = type;
    return std::vector<float3>(cells.h_X, cells.h_X + n);
}

int main()
{
    Solution_n<float3, 500, Grid_solver> once{30, 1.f};          // the three-parameter spelling
    Solution<float3, Grid_solver> by_hand{500, 30, 1.f};
    std::vector<int> ta, tb;
    const auto a = run(once, true, ta), b = run(by_hand, false, tb);
    const bool same = memcmp(a.data(), b.data(), a.size() * sizeof(float3)) == 0 && ta == tb;
    int moved = 0;
    for (int i = 0; i < 500; i++) moved += ta[i] != (i * 7) % 3;
    printf("capacity %d same %d moved %d\n", decltype(once)::capacity, (int)same, moved > 50);
    return !(same && moved > 50 && decltype(once)::capacity == 500);
}
'''


def test_solution_n_and_keep_in_cube_order_on_the_oracle(tmp_path):
    src = tmp_path / "model.cpp"
    src.write_text(SRC)
    exe = tmp_path / "model"
    subprocess.run(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "oracle"),
                    str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "capacity 500 same 1 moved 1" in out.stdout, out.stdout + out.stderr
