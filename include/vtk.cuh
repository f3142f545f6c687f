// Legacy-VTK output/input of a Solution (ASCII POLYDATA), progress printing.
//
// API and on-disk format parity with ya||a `include/vtk.cuh:1-378`:
// Vtk_output{base_name, output_path = "output/", verbose = true} with
// write_positions (first), write_links (second, if any), write_field,
// write_polarity, write_property -- files are output/base_name_#.vtk --
// and Vtk_input{file_name} with n_points, find_entry, read_positions,
// read_polarity, read_field, read_property.  This is host code beside the step
// path (one frame per output interval).  Each section is formatted into one
// memory buffer and written with a single call instead of one stream insertion
// per number, which is what dominates wall time at 10^6 cells.
#pragma once

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <sys/stat.h>
#include <time.h>

#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <typeinfo>
#include <vector>

#include "links.cuh"
#include "polarity.cuh"
#include "utils.cuh"


template<typename Pt, template<typename> class Solver>
class Solution;

template<typename Prop>
struct Property;


namespace ya {
// Append printf-formatted text to a growing buffer.
class Text {
public:
    template<typename... Args>
    void add(const char* format, Args... args)
    {
        char piece[128];
        const int len = snprintf(piece, sizeof(piece), format, args...);
        if (len < 0) return;
        if ((size_t)len < sizeof(piece)) {
            buffer.append(piece, len);
            return;
        }
        // longer than the stack buffer (e.g. a long field name): format into the string itself
        const size_t old_size = buffer.size();
        buffer.resize(old_size + (size_t)len + 1);
        snprintf(&buffer[old_size], (size_t)len + 1, format, args...);
        buffer.resize(old_size + (size_t)len);
    }
    void add(const std::string& s) { buffer += s; }
    // "%g" is what `ostream << float` prints at the default precision of 6
    void number(float v) { add("%g", (double)v); }
    void number(double v) { add("%g", v); }
    void number(int v) { add("%d", v); }
    void number(unsigned v) { add("%u", v); }
    void write_to(const std::string& path, const char* mode)
    {
        FILE* f = fopen(path.c_str(), mode);
        assert(f != NULL);
        fwrite(buffer.data(), 1, buffer.size(), f);
        fclose(f);
    }

private:
    std::string buffer;
};
}  // namespace ya


class Vtk_output {
    int n_points;
    int n_to_write;
    bool* mask = NULL;
    int time_step{0};
    std::string base_name;
    std::string output_dir;
    std::string current_path;
    bool verbose;
    bool point_data_started;
    time_t t_0;

    bool skipped(int i) const { return mask != NULL and mask[i] == 0; }
    void start_point_data(ya::Text& out)
    {
        if (point_data_started) return;
        out.add("\nPOINT_DATA %d\n", n_to_write);
        point_data_started = true;
    }

public:
    // Files are stored as output/base_name_#.vtk
    Vtk_output(std::string base_name, std::string output_path = "output/", bool verbose = true)
        : base_name{base_name}, output_dir{output_path}, verbose{verbose}
    {
        if (output_dir.back() != '/') {
            output_dir.append("/");
            std::cout << output_dir << std::endl;
        }
        mkdir(output_dir.c_str(), 0755);
        time(&t_0);
    }
    ~Vtk_output(void)
    {
        if (!verbose) return;

        const long duration = time(NULL) - t_0;
        std::cout << "Integrating " << base_name << ", ";
        if (duration < 60)
            std::cout << duration << " seconds";
        else if (duration < 60 * 60)
            std::cout << duration / 60 << "m " << duration % 60 << "s";
        else
            std::cout << duration / (60 * 60) << "h " << duration % (60 * 60) << "m";
        std::cout << " taken (" << n_points << " points).        \n";  // Overwrite everything
    }

    // Write x, y, and z component of Pt; has to be written first
    template<typename Pt, template<typename> class Solver>
    void write_positions(Solution<Pt, Solver>& points, bool* input_mask = NULL)
    {
        n_points = *points.h_n;
        mask = input_mask;
        n_to_write = 0;
        for (int i = 0; i < n_points; i++) n_to_write += !skipped(i);

        current_path = output_dir + base_name + "_" + std::to_string(time_step) + ".vtk";
        ya::Text out;
        out.add("# vtk DataFile Version 3.0\n");
        out.add(base_name + "\n");
        out.add("ASCII\nDATASET POLYDATA\n");
        out.add("\nPOINTS %d float\n", n_to_write);
        for (int i = 0; i < n_points; i++) {
            if (skipped(i)) continue;
            out.add("%g %g %g\n", (double)points.h_X[i].x, (double)points.h_X[i].y,
                (double)points.h_X[i].z);
        }
        out.add("\nVERTICES %d %d\n", n_to_write, 2 * n_to_write);
        for (int i = 0; i < n_to_write; i++) out.add("1 %d\n", i);
        out.write_to(current_path, "w");

        point_data_started = false;
        time_step += 1;
        if (!verbose) return;

        std::cout << "Integrating " << base_name << ", ";
        std::cout << time_step << " steps done (" << n_points << " points)        \r";
        std::cout.flush();
    }

    // Write links, see links.cuh; if written has to be second
    void write_links(Links& links)
    {
        ya::Text out;
        out.add("\nLINES %d %d\n", *links.h_n, 3 * *links.h_n);
        for (int i = 0; i < *links.h_n; i++)
            out.add("2 %d %d\n", links.h_link[i].a, links.h_link[i].b);
        out.write_to(current_path, "a");
    }

    // Write further components of Pt
    template<typename Pt, template<typename> class Solver>
    void write_field(
        Solution<Pt, Solver>& points, const char* data_name = "w", float Pt::*field = &Pt::w)
    {
        ya::Text out;
        start_point_data(out);
        out.add("SCALARS %s float\nLOOKUP_TABLE default\n", data_name);
        for (int i = 0; i < n_points; i++) {
            if (skipped(i)) continue;
            out.add("%g\n", (double)(points.h_X[i].*field));
        }
        out.write_to(current_path, "a");
    }

    // Write a polarity vector of Pt (theta and phi by default, see polarity.cuh).
    // Writes {0, 0, 0} for the default theta = phi = 0.
    template<typename Pt, float Pt::*theta = &Pt::theta, float Pt::*phi = &Pt::phi,
        template<typename> class Solver>
    void write_polarity(Solution<Pt, Solver>& points, const char* data_name = "polarity")
    {
        ya::Text out;
        start_point_data(out);
        out.add("NORMALS %s float\n", data_name);
        for (int i = 0; i < n_points; i++) {
            if (skipped(i)) continue;
            float3 n = pol_to_float3<Pt, theta, phi>(points.h_X[i]);
            if ((points.h_X[i].*theta == 0) and (points.h_X[i].*phi == 0)) n.z = 0;
            out.add("%g %g %g\n", (double)n.x, (double)n.y, (double)n.z);
        }
        out.write_to(current_path, "a");
    }

    // Write not integrated property, see property.cuh
    template<typename Prop>
    void write_property(Property<Prop>& property)
    {
        assert(n_points <= property.n_max);
        ya::Text out;
        start_point_data(out);
        const char* ptype = typeid(Prop) == typeid(float) ? "float" : "int";
        out.add("SCALARS " + property.name + " " + ptype + "\nLOOKUP_TABLE default\n");
        for (int i = 0; i < n_points; i++) {
            if (skipped(i)) continue;
            out.number(property.h_prop[i]);
            out.add("\n");
        }
        out.write_to(current_path, "a");
    }
};


class Vtk_input {
    std::string file_name;

    // Stream positioned on the first data line of the section "key1 key2 ...".
    void open_at(std::ifstream& in, std::string keyword1, std::string keyword2)
    {
        in.open(file_name);
        assert(in.is_open());
        in.seekg(find_entry(keyword1, keyword2));
    }

public:
    int n_points;

    Vtk_input(std::string file_name) : file_name{file_name}
    {
        std::ifstream in(file_name);
        assert(in.is_open());
        std::string line;
        n_points = 0;
        for (int i = 0; i < 6 and getline(in, line); i++) {
            const auto items = split(line);
            if (items.size() > 1 and items[0] == "POINTS") {
                n_points = stoi(items[1]);
                break;
            }
        }
    }

    // Position after the line starting with the two keywords (header skipped)
    std::streampos find_entry(std::string keyword1, std::string keyword2)
    {
        std::ifstream in(file_name);
        assert(in.is_open());
        std::string line;
        for (int i = 0; i < 4; i++) getline(in, line);  // header: avoid false matches
        while (getline(in, line)) {
            const auto items = split(line);
            if (items.size() > 1 and items[0] == keyword1 and items[1] == keyword2)
                return in.tellg();
        }
        assert(false and "entry not found in vtk file");
        return in.tellg();
    }

    template<typename Pt, template<typename> class Solver>
    void read_positions(Solution<Pt, Solver>& points)
    {
        std::ifstream in;
        open_at(in, "POINTS", std::to_string(n_points));
        for (int i = 0; i < n_points; i++) in >> points.h_X[i].x >> points.h_X[i].y >> points.h_X[i].z;
    }

    // Read polarity of Pt, see polarity.cuh (the normals are unit vectors)
    template<typename Pt, template<typename> class Solver>
    void read_polarity(Solution<Pt, Solver>& points)
    {
        std::ifstream in;
        open_at(in, "NORMALS", "polarity");
        for (int i = 0; i < n_points; i++) {
            float x, y, z;
            in >> x >> y >> z;
            if (x == 0 and y == 0 and z == 0) {
                points.h_X[i].phi = 0.0f;
                points.h_X[i].theta = 0.0f;
            } else {
                points.h_X[i].phi = atan2(y, x);
                points.h_X[i].theta = acos(z);
            }
        }
    }

    // Read further field of Pt
    template<typename Pt, template<typename> class Solver>
    void read_field(
        Solution<Pt, Solver>& points, const char* data_name = "w", float Pt::*field = &Pt::w)
    {
        std::ifstream in;
        open_at(in, "SCALARS", data_name);
        std::string line;
        getline(in, line);  // LOOKUP_TABLE line
        for (int i = 0; i < n_points; i++) in >> points.h_X[i].*field;
    }

    // Read property, see property.cuh
    template<typename Prop>
    void read_property(Property<Prop>& property, std::string prop_name)
    {
        assert(n_points <= property.n_max);
        std::ifstream in;
        open_at(in, "SCALARS", prop_name);
        std::string line;
        getline(in, line);  // LOOKUP_TABLE line
        for (int i = 0; i < n_points; i++) in >> property.h_prop[i];
    }
};
