// Vtk_output at BASELINE config 4's size (10^6 Po_cell cells; reference vtk.cuh:29-214): the
// bytes on disk must be what one `ostream << float` per number produces (the reference's way,
// restated here as the check), with and without a mask, and the frame must not take longer to
// write than the steps between two frames take to compute.  Prints the timings.
#include "../../include/dtypes.cuh"
#include "../../include/inits.cuh"
#include "../../include/property.cuh"
#include "../../include/solvers.cuh"
#include "../../include/vtk.cuh"

#include <chrono>
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static int failures = 0;
#define EXPECT(cond)                                                  \
    do {                                                              \
        if (!(cond)) {                                                \
            printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
            failures++;                                               \
        }                                                             \
    } while (0)

static double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

static std::string slurp(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    std::stringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

// One stream insertion per number: positions, vertices, polarity normals, an int property.
template<typename S>
static void write_with_streams(const std::string& path, const std::string& name, S& cells,
    Property<int>& type, const bool* mask)
{
    const int n = *cells.h_n;
    int n_write = 0;
    for (int i = 0; i < n; i++) n_write += !(mask && !mask[i]);
    std::ofstream f(path);
    f << "# vtk DataFile Version 3.0\n" << name << "\nASCII\nDATASET POLYDATA\n";
    f << "\nPOINTS " << n_write << " float\n";
    for (int i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        f << cells.h_X[i].x << " " << cells.h_X[i].y << " " << cells.h_X[i].z << "\n";
    }
    f << "\nVERTICES " << n_write << " " << 2 * n_write << "\n";
    for (int i = 0; i < n_write; i++) f << "1 " << i << "\n";
    f << "\nPOINT_DATA " << n_write << "\n";
    f << "NORMALS polarity float\n";
    for (int i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        float3 p = pol_to_float3(cells.h_X[i]);
        if (cells.h_X[i].theta == 0 && cells.h_X[i].phi == 0) p.z = 0;
        f << p.x << " " << p.y << " " << p.z << "\n";
    }
    f << "SCALARS " << type.name << " int\nLOOKUP_TABLE default\n";
    for (int i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        f << type.h_prop[i] << "\n";
    }
}

int main()
{
    const int n = 1000000;
    Solution<Po_cell, Tile_solver> cells{n};
    Property<int> type{n, "type"};
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> pos(-45.f, 45.f), angle(0.f, 3.1415927f);
    *cells.h_n = n;
    for (int i = 0; i < n; i++) {
        cells.h_X[i] = Po_cell{pos(rng), pos(rng), pos(rng), 0.f, 0.f};
        if (i % 3) {  // a third of the cells unpolarised: written as {0, 0, 0}
            cells.h_X[i].theta = angle(rng);
            cells.h_X[i].phi = 2 * angle(rng) - 3.1415927f;
        }
        if (i % 1000 == 0) cells.h_X[i].x = 1e-7f * i;  // exponents
        type.h_prop[i] = (int)(rng() % 3);
    }
    std::vector<char> keep(n);
    for (int i = 0; i < n; i++) keep[i] = (rng() % 4) != 0;
    bool* mask = reinterpret_cast<bool*>(keep.data());

    {  // the other half of an output frame: the host mirror's trip over PCIe (page-locked)
        cells.copy_to_device();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 10; k++) cells.copy_to_host();
        const double each = seconds_since(t0) / 10;
        printf("copy_to_host of %d Po_cell cells (%.0f MB): %.2f ms = %.1f GB/s\n", n, n * 20 / 1e6, each * 1e3,
            n * 20 / each / 1e9);
        EXPECT(cells.h_X[12345].x != 0.f && *cells.h_n == n);
    }
    const std::string dir = "/tmp/yalla_vtk_speed/";
    {
        Vtk_output out{"fast", dir, false};
        auto t0 = std::chrono::steady_clock::now();
        out.write_positions(cells);
        out.write_polarity(cells);
        out.write_property(type);
        const double fast = seconds_since(t0);
        t0 = std::chrono::steady_clock::now();
        write_with_streams(dir + "streams_0.vtk", "fast", cells, type, nullptr);
        const double streams = seconds_since(t0);
        const std::string a = slurp(dir + "fast_0.vtk"), b = slurp(dir + "streams_0.vtk");
        EXPECT(a.size() > 50u * n / 2 && a == b);
        printf("frame of %d Po_cell cells (positions, polarity, one property; %.1f MB): Vtk_output %.3f s, "
               "one stream insertion per number %.3f s\n", n, a.size() / 1e6, fast, streams);
        EXPECT(fast < streams);

        out.write_positions(cells, mask);  // frame 1: three quarters of the cells
        out.write_polarity(cells);
        out.write_property(type);
        write_with_streams(dir + "streams_1.vtk", "fast", cells, type, mask);
        EXPECT(slurp(dir + "fast_1.vtk") == slurp(dir + "streams_1.vtk"));
    }
    {  // and back: Vtk_input reads what was written (6 significant digits)
        auto t0 = std::chrono::steady_clock::now();
        Vtk_input in{dir + "fast_0.vtk"};
        EXPECT(in.n_points == n);
        Solution<Po_cell, Tile_solver> back{n};
        *back.h_n = n;
        in.read_positions(back);
        in.read_polarity(back);
        Property<int> type_back{n, "type"};
        in.read_property(type_back, "type");
        printf("Vtk_input of the same frame: %.3f s\n", seconds_since(t0));
        int wrong = 0;
        for (int i = 0; i < n; i++) wrong += type_back.h_prop[i] != type.h_prop[i];
        EXPECT(wrong == 0);
        double worst = 0;
        for (int i = 0; i < n; i += 97)
            worst = std::max(worst, (double)std::fabs(back.h_X[i].y - cells.h_X[i].y) /
                                        std::max(1e-3, (double)std::fabs(cells.h_X[i].y)));
        EXPECT(worst < 1e-5);
    }
    {  // a model's output loop (reference examples/branching.cu:263-280): the steps of the next
       // frame run in a worker thread while the main thread writes the frame it copied to the host
       // -- take_step must touch neither h_X nor h_n, and must work from any host thread
        const int m = 60000;
        auto simulate = [&](bool threaded, std::vector<float3>& final_state, std::string& frame) {
            Solution<float3, Grid_solver> s{m, 64, 1.f};
            random_sphere(0.5f, s, 0, 3);
            Vtk_output out{threaded ? "threaded" : "serial", dir, false};
            for (int f = 0; f < 3; f++) {
                s.copy_to_host();
                if (threaded) {
                    std::thread worker([&] { for (int k = 0; k < 5; k++) s.take_step<relu_force>(0.05f); });
                    out.write_positions(s);
                    worker.join();
                } else {
                    out.write_positions(s);
                    for (int k = 0; k < 5; k++) s.take_step<relu_force>(0.05f);
                }
            }
            s.copy_to_host();
            final_state.assign(s.h_X, s.h_X + m);
            frame = slurp(dir + (threaded ? "threaded_2.vtk" : "serial_2.vtk"));
        };
        std::vector<float3> a, b;
        std::string frame_a, frame_b;
        simulate(false, a, frame_a);
        simulate(true, b, frame_b);
        EXPECT(memcmp(a.data(), b.data(), m * sizeof(float3)) == 0);
        EXPECT(frame_a.size() > 1000000u && frame_a.substr(frame_a.find("ASCII")) == frame_b.substr(frame_b.find("ASCII")));
    }
    if (failures == 0) printf("ALL VTK SPEED TESTS PASSED\n");
    return failures != 0;
}
