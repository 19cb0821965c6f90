// driver_common.h — shared host code of the two drop-in executables.
//
// They replace the reference's host drivers (run_ray_tracer + main of src/raytracer_3d.cu:3-227 and
// src/raytracer_3d_io.cu:3-236): same positional argument (photon count = 2^n), same stdout lines,
// same input and output files in the working directory. All device work goes through the C ABI of
// include/rt_photon.h. Optional --key=value arguments replace the reference's compile-time literals
// (its defaults are kept).
#pragma once

#include "../include/rt_photon.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

namespace drv
{
struct Config
{
    int itot, jtot, ktot;
    float dx, dy, dz;
    float albedo = 0.2f;                      // raytracer_3d.cu:24
    std::vector<double> zenith_deg{50.};      // raytracer_3d.cu:25 (a list = resident-grid sweep)
    std::vector<double> azimuth_deg{20.};     // raytracer_3d.cu:26 (likewise; every pair of the two lists is run)
    uint64_t seed = 0;
    int ngpus = 1;
    std::string indir = ".", outdir = ".";
    int exponent = -1;
    bool fixed_dot = false;
    // --launch=auto (default) | sobol | K | uniform (the library's "launch_log2k": -2, K >= 0, -1). sobol: photon ID n
    // enters at point n of the two-dimensional Sobol sequence, digitally shifted by the seed -- the reference's own
    // launch (scrambled Sobol32, raytracer_kernels.cu:239-253, 331) as a function of the ID, even for any photon
    // count on any grid. K: stratified, 2^K consecutive photon IDs per column; it samples the domain top evenly only
    // over whole passes of ncol * 2^K photons per GPU. "auto" takes the largest K <= 5 (<= 8 on grids beyond the
    // caches, where it also counts the columns in tiles of 32 x 32: --launch-tile=T, the library's
    // "launch_tile_log2") for which the photon count allows whole passes, else sobol -- e.g. always on the 240 x 240
    // columns of raytracer_3d_io.cu:15-17. uniform: Philox positions (Poisson counts per column). Sobol and uniform
    // launches are sorted by launch tile on the device.
    int launch_log2k = -1;
    bool launch_auto = true;
    int launch_tile_log2 = -1;      // stratified launch: columns counted in tiles of 2^T columns square; -1 = with auto
    // --resume: the output directory already holds the seven tally files of an earlier run and its
    // `checkpoint` file; this run adds photons [next_photon_id, next_photon_id + 2^n) to them (the restart the
    // reference prepares for by copying its tally arrays to the device, raytracer_3d.cu:89, 98-104)
    bool resume = false;
    bool input_f64 = false;                   // sidecar says the data files hold float64
    // --pipeline=1 (default) | 0 | K: a sweep traces job k+1 while job k's seven arrays come down over a copy stream into
    // pinned memory and host threads write its files (rt_download_tallies_async, rt_swap_tally_buffer); 0 = the
    // reference's order, one thing after another (raytracer_3d_io.cu:156-212). The files are the same bytes either way.
    // K >= 2: the files of up to K jobs are written side by side, each job from its own set of pinned arrays (a job's
    // files take several times as long as its trace and download); 1 = automatic: 4, fewer where four sets would
    // page-lock more than 4 GiB, never fewer than 2.
    bool pipeline = true;
    int pipeline_depth = 0;
};

inline void die(const std::string& msg)
{
    std::cerr << msg << std::endl;
    std::exit(1);
}

inline void check(int rc, const char* what)
{
    if (rc != RT_OK)
        die(std::string(what) + ": " + rt_last_error());
}

inline std::string trimmed(const std::string& t)
{
    const size_t b = t.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    return t.substr(b, t.find_last_not_of(" \t\r\n") - b + 1);
}

// The directory given by --indir= (default "."): needed before the other options are parsed.
inline std::string find_indir(int argc, char** argv)
{
    std::string dir = ".";
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        if (a.rfind("--indir=", 0) == 0) dir = a.substr(8);
    }
    return dir;
}

// Sidecar <indir>/grid.hdr (`key = value` lines, written by raytracer_amd.scenes.write_sidecar /
// python -m raytracer_amd.convert): dimensions, spacings and dtype of the headerless data files, which
// raytracer_3d_io.cu:11-17 compiles in. Absent file: the defaults stand. Command-line options parsed
// afterwards override it. Returns whether a sidecar was read.
inline bool read_sidecar(const std::string& dir, Config& c)
{
    std::ifstream f(dir + "/grid.hdr");
    if (!f) return false;
    std::string line;
    int lineno = 0;
    while (std::getline(f, line))
    {
        ++lineno;
        line = trimmed(line.substr(0, line.find('#')));
        if (line.empty()) continue;
        const size_t eq = line.find('=');
        if (eq == std::string::npos) die("grid.hdr:" + std::to_string(lineno) + ": expected `key = value`");
        const std::string key = trimmed(line.substr(0, eq)), val = trimmed(line.substr(eq + 1));
        try
        {
            if (key == "itot") c.itot = std::stoi(val);
            else if (key == "jtot") c.jtot = std::stoi(val);
            else if (key == "ktot") c.ktot = std::stoi(val);
            else if (key == "dx") c.dx = std::stof(val);
            else if (key == "dy") c.dy = std::stof(val);
            else if (key == "dz") c.dz = std::stof(val);
            else if (key == "albedo") c.albedo = std::stof(val);
            else if (key == "zenith_deg") c.zenith_deg = {std::stod(val)};
            else if (key == "azimuth_deg") c.azimuth_deg = {std::stod(val)};
            else if (key == "dtype")
            {
                if (val == "float64") c.input_f64 = true;
                else if (val == "float32") c.input_f64 = false;
                else die("grid.hdr:" + std::to_string(lineno) + ": dtype " + val + ": float32 or float64");
            }
            else if (key == "order")
            {
                if (val != "kji") die("grid.hdr:" + std::to_string(lineno) + ": only order = kji is supported");
            }
            else die("grid.hdr:" + std::to_string(lineno) + ": unknown key " + key);
        }
        catch (const std::logic_error&)
        {
            die("grid.hdr:" + std::to_string(lineno) + ": cannot parse value \"" + val + "\"");
        }
    }
    return true;
}

// argv: exactly one positional (the exponent n, raytracer_3d.cu:209-215) plus --key=value options.
inline void parse_args(int argc, char** argv, Config& c)
{
    int positional = 0;
    for (int i = 1; i < argc; ++i)
    {
        std::string a = argv[i];
        if (a.rfind("--", 0) == 0)
        {
            const size_t eq = a.find('=');
            const std::string key = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            const std::string val = eq == std::string::npos ? "1" : a.substr(eq + 1);
            if (key == "itot") c.itot = std::stoi(val);
            else if (key == "jtot") c.jtot = std::stoi(val);
            else if (key == "ktot") c.ktot = std::stoi(val);
            else if (key == "dx") c.dx = std::stof(val);
            else if (key == "dy") c.dy = std::stof(val);
            else if (key == "dz") c.dz = std::stof(val);
            else if (key == "albedo") c.albedo = std::stof(val);
            else if (key == "azimuth")
            {
                c.azimuth_deg.clear();
                std::stringstream ss(val);
                std::string item;
                while (std::getline(ss, item, ',')) c.azimuth_deg.push_back(std::stod(item));
                if (c.azimuth_deg.empty()) die("--azimuth needs at least one angle");
            }
            else if (key == "resume") c.resume = (val != "0");
            else if (key == "pipeline")
            {
                c.pipeline = (val != "0");
                c.pipeline_depth = std::stoi(val) >= 2 ? std::stoi(val) : 0;
                if (c.pipeline_depth > 16) die("--pipeline: at most 16 jobs in flight");
            }
            else if (key == "seed") c.seed = std::stoull(val);
            else if (key == "ngpus") c.ngpus = std::stoi(val);
            else if (key == "indir") c.indir = val;
            else if (key == "outdir") c.outdir = val;
            else if (key == "physics") c.fixed_dot = (val == "fixed");
            else if (key == "launch")
            {
                if (val == "uniform") { c.launch_log2k = -1; c.launch_auto = false; }
                else if (val == "sobol") { c.launch_log2k = -2; c.launch_auto = false; }
                else if (val == "auto") c.launch_auto = true;
                else { c.launch_log2k = std::stoi(val); c.launch_auto = false; }
            }
            else if (key == "launch-tile")
            {
                c.launch_tile_log2 = std::stoi(val);
                if (c.launch_tile_log2 < 0 || c.launch_tile_log2 > 8) die("--launch-tile must be 0..8 (log2 of the tile edge in columns)");
            }
            else if (key == "zenith")
            {
                c.zenith_deg.clear();
                std::stringstream ss(val);
                std::string item;
                while (std::getline(ss, item, ',')) c.zenith_deg.push_back(std::stod(item));
                if (c.zenith_deg.empty()) die("--zenith needs at least one angle");
            }
            else die("Unknown option --" + key);
        }
        else
        {
            ++positional;
            c.exponent = std::stoi(a);
        }
    }
    if (positional != 1)
    {
        // message and exit status of raytracer_3d.cu:209-213
        std::cout << "The number of photons is must be a power of two (2**n), please add the exponent n" << std::endl;
        std::exit(1);
    }
    if (c.exponent < 0 || c.exponent > 62) die("The exponent n must be in [0, 62]");
}

inline void save_binary(const std::string& dir, const std::string& name, const void* ptr, size_t count)
{
    // raytracer_3d.cu:184-195
    std::ofstream f(dir + "/" + name + ".bin", std::ios::out | std::ios::trunc | std::ios::binary);
    if (!f) throw std::runtime_error("Cannot write file \"" + name + ".bin\"");
    f.write(reinterpret_cast<const char*>(ptr), count * sizeof(uint64_t));
}

// float64 data files (sidecar dtype = float64) are narrowed to float32, the tracer's arithmetic type.
inline void load_binary_f64(const std::string& dir, const std::string& name, float* ptr, size_t count)
{
    std::ifstream f(dir + "/" + name + ".bin", std::ios::in | std::ios::binary);
    if (!f) throw std::runtime_error("Cannot read file \"" + name + ".bin\"");
    std::vector<double> buf(1 << 16);
    size_t done = 0;
    while (done < count)
    {
        const size_t n = std::min(buf.size(), count - done);
        f.read(reinterpret_cast<char*>(buf.data()), n * sizeof(double));
        if (static_cast<size_t>(f.gcount()) != n * sizeof(double))
            throw std::runtime_error("File \"" + name + ".bin\" is shorter than itot*jtot*ktot float64 values");
        for (size_t i = 0; i < n; ++i) ptr[done + i] = static_cast<float>(buf[i]);
        done += n;
    }
}

inline void load_binary(const std::string& dir, const std::string& name, float* ptr, size_t count)
{
    // raytracer_3d_io.cu:36-47, reading exactly count float32 (the reference asks for twice as
    // many bytes and relies on the short read, :41)
    std::ifstream f(dir + "/" + name + ".bin", std::ios::in | std::ios::binary);
    if (!f) throw std::runtime_error("Cannot read file \"" + name + ".bin\"");
    f.read(reinterpret_cast<char*>(ptr), count * sizeof(float));
    if (static_cast<size_t>(f.gcount()) != count * sizeof(float))
        throw std::runtime_error("File \"" + name + ".bin\" is shorter than itot*jtot*ktot float32 values");
}

static const char* const kTallyNames[RT_N_TALLIES] = {
    "toa_down", "toa_up", "surface_down_direct", "surface_down_diffuse", "surface_up", "atmos_direct", "atmos_diffuse"};

// `checkpoint` beside the seven tally files: what a later --resume needs to continue the same photon sequence
// (`key = value` lines). Every run writes it after its tallies.
struct Checkpoint
{
    uint64_t seed = 0, next_photon_id = 0;
    int itot = 0, jtot = 0, ktot = 0, launch_log2k = -1, launch_tile_log2 = 0;
    double zenith_deg = 0., azimuth_deg = 0.;
    float albedo = 0.f;
    bool fixed_dot = false;
};

inline void write_checkpoint(const std::string& dir, const Checkpoint& k)
{
    std::ofstream f(dir + "/checkpoint", std::ios::out | std::ios::trunc);
    if (!f) throw std::runtime_error("Cannot write file \"checkpoint\"");
    f << "# state of the tally files in this directory; continue with --resume\n"
      << "seed = " << k.seed << "\nnext_photon_id = " << k.next_photon_id << "\n"
      << "itot = " << k.itot << "\njtot = " << k.jtot << "\nktot = " << k.ktot << "\n"
      << "launch_log2k = " << k.launch_log2k << "\nlaunch_tile_log2 = " << k.launch_tile_log2 << "\n"
      << std::setprecision(17) << "zenith_deg = " << k.zenith_deg << "\nazimuth_deg = " << k.azimuth_deg << "\n"
      << std::setprecision(9) << "albedo = " << k.albedo << "\nfixed_dot = " << (k.fixed_dot ? 1 : 0) << "\n";
}

inline Checkpoint read_checkpoint(const std::string& dir)
{
    std::ifstream f(dir + "/checkpoint");
    if (!f) die("--resume: cannot read file \"" + dir + "/checkpoint\" (written by an earlier run into the same --outdir)");
    Checkpoint k;
    std::string line;
    while (std::getline(f, line))
    {
        line = trimmed(line.substr(0, line.find('#')));
        if (line.empty()) continue;
        const size_t eq = line.find('=');
        if (eq == std::string::npos) die("checkpoint: expected `key = value`");
        const std::string key = trimmed(line.substr(0, eq)), val = trimmed(line.substr(eq + 1));
        try
        {
            if (key == "seed") k.seed = std::stoull(val);
            else if (key == "next_photon_id") k.next_photon_id = std::stoull(val);
            else if (key == "itot") k.itot = std::stoi(val);
            else if (key == "jtot") k.jtot = std::stoi(val);
            else if (key == "ktot") k.ktot = std::stoi(val);
            else if (key == "launch_log2k") k.launch_log2k = std::stoi(val);
            else if (key == "launch_tile_log2") k.launch_tile_log2 = std::stoi(val);
            else if (key == "zenith_deg") k.zenith_deg = std::stod(val);
            else if (key == "azimuth_deg") k.azimuth_deg = std::stod(val);
            else if (key == "albedo") k.albedo = std::stof(val);
            else if (key == "fixed_dot") k.fixed_dot = std::stoi(val) != 0;
            else die("checkpoint: unknown key " + key);
        }
        catch (const std::logic_error&)
        {
            die("checkpoint: cannot parse value \"" + val + "\"");
        }
    }
    return k;
}

inline void load_tally(const std::string& dir, const std::string& name, uint64_t* ptr, size_t count)
{
    std::ifstream f(dir + "/" + name + ".bin", std::ios::in | std::ios::binary);
    if (!f) throw std::runtime_error("Cannot read file \"" + name + ".bin\"");
    f.read(reinterpret_cast<char*>(ptr), count * sizeof(uint64_t));
    if (static_cast<size_t>(f.gcount()) != count * sizeof(uint64_t) || f.peek() != std::ifstream::traits_type::eof())
        throw std::runtime_error("File \"" + name + ".bin\" does not hold the " + std::to_string(count) + " counts of this grid");
}

// The body of run_ray_tracer after the fields exist (raytracer_3d.cu:72-203).
inline void run(const Config& c, const std::vector<float>& k_gas, const std::vector<float>& k_cloud,
                const std::vector<float>& ssa, const std::vector<float>& asy)
{
    const uint64_t n_photons = 1ull << c.exponent;
    int ndev = 0;
    check(rt_device_count(&ndev), "rt_device_count");
    if (c.ngpus < 1 || c.ngpus > ndev) die("--ngpus must be between 1 and the number of visible GPUs");
    // first line of the reference (raytracer_3d.cu:7): its launch was 64 x 512 threads
    std::cout << "Shooting " << n_photons << " photons (" << n_photons / (64 * 512) << " per thread) " << std::endl;

    const size_t ncol = (size_t)c.itot * c.jtot, ncell = ncol * c.ktot;
    // Launch order. auto: stratified by column with the largest K that gives whole passes over the columns; on a grid
    // beyond the caches (packed cells > 256 MiB) whose sides are multiples of 32 the columns are counted in tiles of
    // 32 x 32 and K may go up to 8 (one tile's photons keep its cells in the XCD's L2: +6 % on 1024 x 1024 x 128),
    // smaller grids keep rows and K <= 5 (more photons per column at a time contend for the same tally words there).
    // No K with whole passes: the Sobol launch (-2), which the library sorts by launch tile on the device.
    const bool tiles_fit = c.itot % 32 == 0 && c.jtot % 32 == 0;
    const bool beyond_caches = 16ull * ncell > (256ull << 20);
    int launch_log2k = c.launch_log2k;
    int launch_tile_log2 = c.launch_tile_log2 >= 0 ? c.launch_tile_log2 : (beyond_caches && tiles_fit ? 5 : 0);
    if (c.launch_auto)
    {
        launch_log2k = -2;
        for (int k = (beyond_caches && launch_tile_log2 > 0 ? 8 : 5); k >= 0 && launch_log2k < 0; --k)
            if (n_photons % ((uint64_t)c.ngpus * (ncol << k)) == 0) launch_log2k = k;
    }
    else if (launch_log2k >= 0 && n_photons % ((uint64_t)c.ngpus * (ncol << launch_log2k)) != 0)
        std::cerr << "Warning: --launch=" << launch_log2k << " covers the domain top evenly only for whole passes of "
                  << (ncol << launch_log2k) << " photons per GPU" << std::endl;
    if (launch_log2k < 0) launch_tile_log2 = 0;
    if (launch_tile_log2 > 0 && (c.itot % (1 << launch_tile_log2) != 0 || c.jtot % (1 << launch_tile_log2) != 0))
        die("--launch-tile=" + std::to_string(launch_tile_log2) + " needs itot and jtot to be multiples of " +
            std::to_string(1 << launch_tile_log2));

    const bool sweep_z = c.zenith_deg.size() > 1, sweep_a = c.azimuth_deg.size() > 1;
    struct Job { double zen, azi; std::string dir; Checkpoint from; };
    std::vector<Job> jobs;
    for (const double zen : c.zenith_deg)
        for (const double azi : c.azimuth_deg)
        {
            std::ostringstream d;
            d << c.outdir;
            if (sweep_z) d << "/zenith_" << std::setw(2) << std::setfill('0') << (int)std::lround(zen);
            if (sweep_a) d << (sweep_z ? "_" : "/") << "azimuth_" << std::setw(3) << std::setfill('0') << (int)std::lround(azi);
            Job job{zen, azi, d.str(), Checkpoint()};
            if (c.resume)
            {
                // the earlier run's state: it fixes the photon sequence this run continues
                job.from = read_checkpoint(job.dir);
                const Checkpoint& k = job.from;
                if (k.itot != c.itot || k.jtot != c.jtot || k.ktot != c.ktot) die("--resume: " + job.dir + " holds another grid");
                if (k.seed != c.seed || k.zenith_deg != zen || k.azimuth_deg != azi || k.albedo != c.albedo || k.fixed_dot != c.fixed_dot)
                    die("--resume: " + job.dir + "/checkpoint was written with another seed, sun position, albedo or physics option");
                if (!c.launch_auto && k.launch_log2k != launch_log2k)
                    die("--resume: " + job.dir + " was launched with --launch=" + std::to_string(k.launch_log2k));
                if (c.launch_tile_log2 >= 0 && k.launch_tile_log2 != c.launch_tile_log2)
                    die("--resume: " + job.dir + " was launched with --launch-tile=" + std::to_string(k.launch_tile_log2));
            }
            jobs.push_back(job);
        }
    // a resumed run keeps the launch order of the run it continues (the passes over the columns go on)
    if (c.resume) { launch_log2k = jobs[0].from.launch_log2k; launch_tile_log2 = jobs[0].from.launch_tile_log2; }
    for (const Job& job : jobs)
        if (c.resume && (job.from.launch_log2k != launch_log2k || job.from.launch_tile_log2 != launch_tile_log2))
            die("--resume: the directories of this sweep disagree on the launch order");

    std::vector<rt_context*> ctx(c.ngpus, nullptr);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int g = 0; g < c.ngpus; ++g)
    {
        check(rt_create(g, c.itot, c.jtot, c.ktot, c.dx, c.dy, c.dz, &ctx[g]), "rt_create");
    }
    // the four fields to every GPU side by side (page-locked for the call, asynchronous copies)
    check(rt_upload_optics_multi(ctx.data(), c.ngpus, k_gas.data(), k_cloud.data(), ssa.data(), asy.data()), "rt_upload_optics");
    for (int g = 0; g < c.ngpus; ++g)
    {
        check(rt_set_option(ctx[g], "fixed_dot", c.fixed_dot), "rt_set_option");
        check(rt_set_option(ctx[g], "launch_log2k", launch_log2k), "rt_set_option");
        check(rt_set_option(ctx[g], "launch_tile_log2", launch_tile_log2), "rt_set_option");
        // majorant tables on every GPU now: the launches inside the timed region below are asynchronous
        check(rt_prepare(ctx[g]), "rt_prepare");
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    rt_slab_info slabs;
    check(rt_get_slabs(ctx[0], &slabs), "rt_get_slabs");
    std::cout << "Duration cloud-mask computation: " << std::setprecision(5)
              << std::chrono::duration<double>(t1 - t0).count() << " (s)" << std::endl;
    std::cout << slabs.cloud_min << " - " << slabs.cloud_max << std::endl;      // raytracer_3d.cu:130

    // Host side of the results: two sets of the seven arrays in pinned memory (the copy engine writes them while the
    // GPU traces the next job) -- one set when the pipeline is off.
    const size_t nwords = 5 * ncol + 2 * ncell;
    int depth = c.pipeline_depth;
    if (depth == 0)
    {
        depth = 4;
        while (depth > 2 && (size_t)depth * nwords * sizeof(uint64_t) > ((size_t)4 << 30)) --depth;
    }
    const int nsets = c.pipeline && jobs.size() > 1 ? std::min<int>(depth, (int)jobs.size()) : 1;
    std::vector<uint64_t*> pinned(nsets, nullptr);
    std::vector<std::array<uint64_t*, RT_N_TALLIES>> ptrs(nsets);
    for (int b = 0; b < nsets; ++b)
    {
        void* p = nullptr;
        check(rt_host_alloc(nwords * sizeof(uint64_t), &p), "rt_host_alloc");
        pinned[b] = static_cast<uint64_t*>(p);
        size_t off = 0;
        for (int t = 0; t < RT_N_TALLIES; ++t) { ptrs[b][t] = pinned[b] + off; off += t < RT_ATMOS_DIRECT ? ncol : ncell; }
    }
    const auto count_of = [&](int t) { return t < RT_ATMOS_DIRECT ? ncol : ncell; };

    // What is left of a job once its counts are on the host: files, checkpoint, the reference's stdout lines.
    struct Done { size_t job; int set; uint64_t first_id; double seconds; };
    std::string writer_error;
    std::mutex writer_mutex;
    const auto set_writer_error = [&](const std::string& what) { std::lock_guard<std::mutex> g(writer_mutex); if (writer_error.empty()) writer_error = what; };
    const auto get_writer_error = [&] { std::lock_guard<std::mutex> g(writer_mutex); return writer_error; };
    const auto finish = [&](const Done& d)
    {
        const Job& job = jobs[d.job];
        try
        {
            if (sweep_z || sweep_a) mkdir(job.dir.c_str(), 0777);
            // the two 3D absorption fields are all but 2 % of the bytes: one goes out on a thread of its own
            std::string other_error;
            std::thread other([&] {
                try { save_binary(job.dir, kTallyNames[RT_ATMOS_DIFFUSE], ptrs[d.set][RT_ATMOS_DIFFUSE], count_of(RT_ATMOS_DIFFUSE)); }
                catch (const std::exception& e) { other_error = e.what(); }
            });
            try
            {
                for (int t = 0; t < RT_ATMOS_DIFFUSE; ++t) save_binary(job.dir, kTallyNames[t], ptrs[d.set][t], count_of(t));
            }
            catch (...) { other.join(); throw; }
            other.join();
            if (!other_error.empty()) throw std::runtime_error(other_error);
            Checkpoint k;
            k.seed = c.seed; k.next_photon_id = d.first_id + n_photons;
            k.itot = c.itot; k.jtot = c.jtot; k.ktot = c.ktot; k.launch_log2k = launch_log2k; k.launch_tile_log2 = launch_tile_log2;
            k.zenith_deg = job.zen; k.azimuth_deg = job.azi; k.albedo = c.albedo; k.fixed_dot = c.fixed_dot;
            write_checkpoint(job.dir, k);
        }
        catch (const std::exception& e) { set_writer_error(e.what()); }
    };

    std::vector<std::thread> writers(nsets);       // writers[s]: the thread that writes the files held in set s
    const auto join_writers = [&] { for (std::thread& w : writers) if (w.joinable()) w.join(); };
    bool warned_host_sum = false;
    double trace_seconds = 0.;
    const auto sweep0 = std::chrono::high_resolution_clock::now();
    for (size_t jn = 0; jn < jobs.size(); ++jn)
    {
        const Job& job = jobs[jn];
        const int set = (int)(jn % (size_t)nsets);
        const float zen = static_cast<float>(job.zen * (M_PI / 180.));   // raytracer_3d.cu:25
        const float azi = static_cast<float>(job.azi * (M_PI / 180.));   // raytracer_3d.cu:26
        uint64_t first_id = 0;
        for (int g = 0; g < c.ngpus; ++g)
        {
            check(rt_set_scene(ctx[g], zen, azi, c.albedo), "rt_set_scene");
            check(rt_reset_tallies(ctx[g]), "rt_reset_tallies");
        }
        if (c.resume)
        {
            // the counts so far go to GPU 0 (raytracer_3d.cu:98-104 copies its host arrays up for this); the sum
            // over the GPUs below carries them into the total
            first_id = job.from.next_photon_id;
            std::vector<std::vector<uint64_t>> old(RT_N_TALLIES);
            const uint64_t* oldp[RT_N_TALLIES];
            for (int t = 0; t < RT_N_TALLIES; ++t)
            {
                old[t].resize(count_of(t));
                load_tally(job.dir, kTallyNames[t], old[t].data(), old[t].size());
                oldp[t] = old[t].data();
            }
            check(rt_upload_tallies(ctx[0], oldp), "rt_upload_tallies");
            std::cout << "Resuming " << job.dir << " at photon " << first_id << std::endl;
        }
        auto s0 = std::chrono::high_resolution_clock::now();
        // photon IDs [first, first + n) split into contiguous shards, one per GPU; launches are asynchronous
        const uint64_t base = n_photons / c.ngpus, rem = n_photons % c.ngpus;
        uint64_t begin = first_id;
        for (int g = 0; g < c.ngpus; ++g)
        {
            const uint64_t cnt = base + ((uint64_t)g < rem ? 1 : 0);
            check(rt_trace(ctx[g], c.seed, begin, cnt), "rt_trace");
            begin += cnt;
        }
        // the path's one exchange step: every GPU ends with the node total (RCCL over xGMI inside the library). Without
        // RCCL (the library cannot be loaded, or a communicator cannot be made) the counts are summed on the host.
        bool host_sum = false;
        if (c.ngpus > 1 && rt_allreduce(ctx.data(), c.ngpus) != RT_OK)
        {
            host_sum = true;
            if (!warned_host_sum)
                std::cerr << "Warning: " << rt_last_error() << "; summing the GPUs' counts on the host instead" << std::endl;
            warned_host_sum = true;
        }
        if (nsets == 1 || host_sum)
        {
            // the reference's order: wait for the kernel, then copy, then write (raytracer_3d.cu:165-203)
            for (int g = 0; g < c.ngpus; ++g) check(rt_synchronize(ctx[g]), "rt_synchronize");
            const double dur = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - s0).count();
            trace_seconds += dur;
            std::cout << "Duration: " << std::setprecision(5) << dur << " (s)" << std::endl;  // raytracer_3d.cu:169
            std::cout << "Photons per second: " << std::setprecision(6) << (double)n_photons / dur << std::endl;
            join_writers();
            check(rt_download_tallies(ctx[0], ptrs[set].data()), "rt_download_tallies");
            if (host_sum)
            {
                std::vector<uint64_t> other(nwords);
                uint64_t* op[RT_N_TALLIES];
                size_t off = 0;
                for (int t = 0; t < RT_N_TALLIES; ++t) { op[t] = other.data() + off; off += count_of(t); }
                for (int g = 1; g < c.ngpus; ++g)
                {
                    check(rt_download_tallies(ctx[g], op), "rt_download_tallies");
                    for (size_t w = 0; w < nwords; ++w) pinned[set][w] += other[w];
                }
            }
            if (sweep_z || sweep_a) std::cout << "Writing " << job.dir << std::endl;
            finish(Done{jn, set, first_id, dur});
        }
        else
        {
            // pipelined sweep: the copy of this job starts when its trace ends (an event, not a host wait), the context
            // moves on to its other tally buffer, and the host goes straight on to enqueue the next job; a thread waits
            // for the copy and writes the files. The set of host arrays must be free again: the files of the job that
            // used it, nsets jobs ago, are waited for first (the writers of the jobs in between go on side by side).
            if (writers[set].joinable()) writers[set].join();
            int ticket = 0;
            check(rt_download_tallies_async(ctx[0], ptrs[set].data(), &ticket), "rt_download_tallies_async");
            check(rt_swap_tally_buffer(ctx[0]), "rt_swap_tally_buffer");
            if (sweep_z || sweep_a) std::cout << "Writing " << job.dir << std::endl;
            writers[set] = std::thread([&, jn, set, first_id, ticket, s0]
            {
                if (rt_download_wait(ctx[0], ticket) != RT_OK) { set_writer_error(rt_last_error()); return; }
                finish(Done{jn, set, first_id, std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - s0).count()});
            });
        }
        if (!get_writer_error().empty()) break;
    }
    join_writers();
    if (!writer_error.empty()) die(writer_error);
    for (int g = 0; g < c.ngpus; ++g) check(rt_synchronize(ctx[g]), "rt_synchronize");
    const double sweep_s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - sweep0).count();
    if (nsets >= 2)
    {
        // one line per sweep in place of one per job (a job's end is not waited for on the host)
        std::cout << "Duration: " << std::setprecision(5) << sweep_s << " (s)" << std::endl;
        std::cout << "Photons per second: " << std::setprecision(6) << (double)n_photons * (double)jobs.size() / sweep_s << std::endl;
    }
    // machine-readable summary of the run ("metrics/logging"): one JSON object on the last line
    std::cout << "{\"photons_per_job\": " << n_photons << ", \"jobs\": " << jobs.size() << ", \"gpus\": " << c.ngpus
              << ", \"grid\": [" << c.itot << ", " << c.jtot << ", " << c.ktot << "], \"launch_log2k\": " << launch_log2k
              << ", \"launch_tile_log2\": " << launch_tile_log2 << ", \"seed\": " << c.seed
              << ", \"pipeline\": " << (nsets >= 2 ? "true" : "false") << ", \"jobs_in_flight\": " << nsets << std::setprecision(9)
              << ", \"sweep_seconds\": " << sweep_s << ", \"end_to_end_photons_per_s\": "
              << (double)n_photons * (double)jobs.size() / sweep_s;
    if (nsets == 1) std::cout << ", \"trace_seconds\": " << trace_seconds;
    std::cout << "}" << std::endl;
    for (int b = 0; b < nsets; ++b) rt_host_free(pinned[b]);
    for (int g = 0; g < c.ngpus; ++g) rt_destroy(ctx[g]);
}
} // namespace drv
