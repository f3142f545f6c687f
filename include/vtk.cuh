// Legacy-VTK output/input of a Solution (ASCII POLYDATA), progress printing.
//
// API and on-disk format parity with ya||a `include/vtk.cuh:1-378`:
// Vtk_output{base_name, output_path = "output/", verbose = true} with
// write_positions (first), write_links (second, if any), write_field,
// write_polarity, write_property -- files are output/base_name_#.vtk --
// and Vtk_input{file_name} with n_points, find_entry, read_positions,
// read_polarity, read_field, read_property.  This is host code beside the step
// path (one frame per output interval) -- but at 10^6 cells it is what a model run spends
// its wall time on: a frame of positions, polarities and one property is ~7 * 10^6 numbers,
// seconds of `ostream << float`, against 11 steps of ~3 ms between frames
// (examples/passive_growth.cu scaled to BASELINE config 4).  So each section is formatted
// into memory buffers -- numbers by std::to_chars (the digits of "%g", which is what
// `ostream << float` prints at its default precision of 6; checked against snprintf for
// 2 * 10^7 values), large sections by several threads over contiguous runs of points -- and
// written with one call per buffer.  The bytes on disk are the same as before.
#pragma once

#include <assert.h>
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include <algorithm>
#include <charconv>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
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
    void add(const char* text) { buffer += text; }  // no arguments: not a format
    void add(const std::string& s) { buffer += s; }
    void add(char c) { buffer.push_back(c); }
    // what `ostream << float` prints at the default precision of 6, i.e. "%g"
    void number(float v)
    {
        char piece[32];
        const auto end = std::to_chars(piece, piece + sizeof(piece), v, std::chars_format::general, 6);
        buffer.append(piece, end.ptr - piece);
    }
    void number(double v) { add("%g", v); }
    void number(int v)
    {
        char piece[16];
        const auto end = std::to_chars(piece, piece + sizeof(piece), v);
        buffer.append(piece, end.ptr - piece);
    }
    void number(unsigned v)
    {
        char piece[16];
        const auto end = std::to_chars(piece, piece + sizeof(piece), v);
        buffer.append(piece, end.ptr - piece);
    }
    // "x y z\n"
    void row(float x, float y, float z)
    {
        number(x);
        add(' ');
        number(y);
        add(' ');
        number(z);
        add('\n');
    }
    void reserve(size_t bytes) { buffer.reserve(bytes); }
    const std::string& str() const { return buffer; }
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

// The rows of points [0, n) formatted by `row(i, text)`, in order: one buffer for small n,
// otherwise one per thread over contiguous runs of points (at most 16 threads, at least
// 32768 points each).
template<typename Row>
std::vector<Text> format_rows(const int n, const Row& row)
{
    const int hardware = (int)std::thread::hardware_concurrency();
    const int threads = std::max(1, std::min({hardware > 0 ? hardware : 1, 16, n / 32768}));
    std::vector<Text> parts(threads);
    auto work = [&](const int t) {
        const long begin = (long)n * t / threads, end = (long)n * (t + 1) / threads;
        parts[t].reserve((size_t)(end - begin) * 24);
        for (long i = begin; i < end; i++) row((int)i, parts[t]);
    };
    if (threads == 1) {
        work(0);
        return parts;
    }
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    return parts;
}

// head, then the parts in order, then tail: one fwrite each
inline void write_sections(const std::string& path, const char* mode, const Text& head,
    const std::vector<Text>& parts, const Text& tail = Text{})
{
    FILE* f = fopen(path.c_str(), mode);
    assert(f != NULL);
    fwrite(head.str().data(), 1, head.str().size(), f);
    for (const auto& part : parts) fwrite(part.str().data(), 1, part.str().size(), f);
    fwrite(tail.str().data(), 1, tail.str().size(), f);
    fclose(f);
}
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
        const auto* X = points.h_X;
        ya::write_sections(current_path, "w", out, ya::format_rows(n_points, [&](int i, ya::Text& t) {
            if (!skipped(i)) t.row(X[i].x, X[i].y, X[i].z);
        }));
        ya::Text vertices;
        vertices.add("\nVERTICES %d %d\n", n_to_write, 2 * n_to_write);
        ya::write_sections(current_path, "a", vertices, ya::format_rows(n_to_write, [](int i, ya::Text& t) {
            t.add('1');
            t.add(' ');
            t.number(i);
            t.add('\n');
        }));

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
        const Link* link = links.h_link;
        ya::write_sections(current_path, "a", out, ya::format_rows(*links.h_n, [&](int i, ya::Text& t) {
            t.add('2');
            t.add(' ');
            t.number(link[i].a);
            t.add(' ');
            t.number(link[i].b);
            t.add('\n');
        }));
    }

    // Write further components of Pt
    template<typename Pt, template<typename> class Solver>
    void write_field(
        Solution<Pt, Solver>& points, const char* data_name = "w", float Pt::*field = &Pt::w)
    {
        ya::Text out;
        start_point_data(out);
        out.add("SCALARS %s float\nLOOKUP_TABLE default\n", data_name);
        const auto* X = points.h_X;
        ya::write_sections(current_path, "a", out, ya::format_rows(n_points, [&](int i, ya::Text& t) {
            if (skipped(i)) return;
            t.number(X[i].*field);
            t.add('\n');
        }));
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
        const auto* X = points.h_X;
        ya::write_sections(current_path, "a", out, ya::format_rows(n_points, [&](int i, ya::Text& t) {
            if (skipped(i)) return;
            float3 n = pol_to_float3<Pt, theta, phi>(X[i]);
            if ((X[i].*theta == 0) and (X[i].*phi == 0)) n.z = 0;
            t.row(n.x, n.y, n.z);
        }));
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
        const Prop* prop = property.h_prop;
        ya::write_sections(current_path, "a", out, ya::format_rows(n_points, [&](int i, ya::Text& t) {
            if (skipped(i)) return;
            t.number(prop[i]);
            t.add('\n');
        }));
    }
};


class Vtk_input {
    std::string file_name;
    std::string text;  // the whole file: sections are located and parsed in memory

    // (Re)load the file unless it still has the size and modification time of the copy held
    // in memory: sections may be appended to a file after this object was made, or the file
    // rewritten with other values of the same length (the reference opens the file anew for
    // every read, and its tests/test_vtk.cu:56-59 relies on that).
    long long loaded_sec = -1, loaded_nsec = -1;
    void load()
    {
        struct stat info;
        const bool there = stat(file_name.c_str(), &info) == 0;
        assert(there and "vtk file not found");
        if (!there) return;
        if ((size_t)info.st_size == text.size() and (long long)info.st_mtim.tv_sec == loaded_sec and
            (long long)info.st_mtim.tv_nsec == loaded_nsec)
            return;
        loaded_sec = (long long)info.st_mtim.tv_sec;
        loaded_nsec = (long long)info.st_mtim.tv_nsec;
        std::ifstream in(file_name, std::ios::binary);
        assert(in.is_open());
        text.resize((size_t)info.st_size);
        in.read(&text[0], (std::streamsize)text.size());
    }

    // First character after the line starting with the two keywords (header skipped)
    size_t entry_offset(const std::string& keyword1, const std::string& keyword2)
    {
        load();
        const char *p = text.data(), *end = p + text.size();
        for (int i = 0; i < 4 and p < end; i++) {  // header: avoid false matches
            const char* nl = (const char*)memchr(p, '\n', end - p);
            p = nl ? nl + 1 : end;
        }
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', end - p);
            const char* line_end = nl ? nl : end;
            // the first two whitespace-separated items of the line
            const char* q = p;
            while (q < line_end and isspace((unsigned char)*q)) q++;
            const char* a = q;
            while (q < line_end and !isspace((unsigned char)*q)) q++;
            const size_t a_len = q - a;
            while (q < line_end and isspace((unsigned char)*q)) q++;
            const char* b = q;
            while (q < line_end and !isspace((unsigned char)*q)) q++;
            const size_t b_len = q - b;
            if (b_len > 0 and a_len == keyword1.size() and b_len == keyword2.size() and
                memcmp(a, keyword1.data(), a_len) == 0 and memcmp(b, keyword2.data(), b_len) == 0)
                return (nl ? nl + 1 : end) - text.data();
            p = nl ? nl + 1 : end;
        }
        assert(false and "entry not found in vtk file");
        return text.size();
    }
    const char* skip_line(const char* p) const
    {
        const char* end = text.data() + text.size();
        const char* nl = (const char*)memchr(p, '\n', end - p);
        return nl ? nl + 1 : end;
    }
    // the next whitespace-separated number, as `istream >> value` reads it
    template<typename T>
    const char* parse(const char* p, T& value) const
    {
        const char* end = text.data() + text.size();
        while (p < end and isspace((unsigned char)*p)) p++;
        if (p < end and *p == '+') p++;
        const auto result = std::from_chars(p, end, value);
        assert(result.ec == std::errc() and "number expected in vtk file");
        return result.ptr;
    }
    const char* parse(const char* p, bool& value) const
    {
        int v = 0;
        p = parse(p, v);
        value = v != 0;
        return p;
    }

public:
    int n_points;

    Vtk_input(std::string file_name) : file_name{file_name}
    {
        load();
        n_points = 0;
        std::istringstream head(text.substr(0, std::min<size_t>(text.size(), 4096)));
        std::string line;
        for (int i = 0; i < 6 and getline(head, line); i++) {
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
        return (std::streampos)entry_offset(keyword1, keyword2);
    }

    template<typename Pt, template<typename> class Solver>
    void read_positions(Solution<Pt, Solver>& points)
    {
        const size_t at = entry_offset("POINTS", std::to_string(n_points));  // may reload `text`
        const char* p = text.data() + at;
        for (int i = 0; i < n_points; i++) {
            p = parse(p, points.h_X[i].x);
            p = parse(p, points.h_X[i].y);
            p = parse(p, points.h_X[i].z);
        }
    }

    // Read polarity of Pt, see polarity.cuh (the normals are unit vectors)
    template<typename Pt, template<typename> class Solver>
    void read_polarity(Solution<Pt, Solver>& points)
    {
        const size_t at = entry_offset("NORMALS", "polarity");
        const char* p = text.data() + at;
        for (int i = 0; i < n_points; i++) {
            float x, y, z;
            p = parse(p, x);
            p = parse(p, y);
            p = parse(p, z);
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
        const size_t at = entry_offset("SCALARS", data_name);
        const char* p = skip_line(text.data() + at);  // LOOKUP_TABLE line
        for (int i = 0; i < n_points; i++) p = parse(p, points.h_X[i].*field);
    }

    // Read property, see property.cuh
    template<typename Prop>
    void read_property(Property<Prop>& property, std::string prop_name)
    {
        assert(n_points <= property.n_max);
        const size_t at = entry_offset("SCALARS", prop_name);
        const char* p = skip_line(text.data() + at);  // LOOKUP_TABLE line
        for (int i = 0; i < n_points; i++) p = parse(p, property.h_prop[i]);
    }
};
