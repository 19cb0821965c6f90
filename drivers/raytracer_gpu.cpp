// raytracer_gpu — drop-in for the reference's 2D executable built from src/raytracer.cpp:
// one argument, the number of photons in multiples of 100 000 (raytracer.cpp:424-433); built-in scene of
// raytracer.cpp:126-173 (256 x 128 cells of 25 m, block cloud, albedo 0.2, zenith 50 deg); writes
// toa_down, toa_up, surface_down_direct, surface_down_diffuse, surface_up (itot uint64) and atmos.bin
// (ktot x itot uint64) into the working directory for python/plot_results.py:17-23.
// The library runs it as the degenerate jtot = 1 grid with option "planar" (the 2D scattering rule).
#include "driver_common.h"

int main(int argc, char* argv[])
{
    int multiple = -1;
    uint64_t seed = 0;
    std::string outdir = ".";
    double zenith_deg = 50., albedo = 0.2;                 // raytracer.cpp:137-138
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        if (a.rfind("--seed=", 0) == 0) seed = std::stoull(a.substr(7));
        else if (a.rfind("--outdir=", 0) == 0) outdir = a.substr(9);
        else if (a.rfind("--zenith=", 0) == 0) zenith_deg = std::stod(a.substr(9));
        else if (a.rfind("--albedo=", 0) == 0) albedo = std::stod(a.substr(9));
        else if (a.rfind("--", 0) == 0) drv::die("Unknown option " + a);
        else if (multiple < 0) multiple = std::stoi(a);
        else multiple = -2;
    }
    if (multiple < 0)
    {
        std::cout << "Add the multiple of 100,000 photons as an argument!" << std::endl;   // raytracer.cpp:428
        return 1;
    }
    const uint64_t n_photons = (uint64_t)multiple * 100000ull;

    // raytracer.cpp:126-173, evaluated in double as there, then narrowed to the tracer's float32
    const double dx_grid = 25.;
    const int itot = 256, ktot = 128;
    const double k_ext_gas = 1.e-4, ssa_gas = 0.5, asy_gas = 0.;
    const double k_ext_cloud = 5.e-3, ssa_cloud = 0.9, asy_cloud = 0.85;
    const size_t n = (size_t)itot * ktot;
    std::vector<float> kg(n, (float)k_ext_gas), kc(n, 0.f), ssa(n, (float)ssa_gas), asy(n, 0.f);
    for (int k = 0; k < ktot; ++k)
        for (int i = 0; i < itot; ++i)
            if (i*dx_grid > 4000 && i*dx_grid < 5000 && k*dx_grid > 1000 && k*dx_grid < 1500.)
            {
                const size_t c = (size_t)i + (size_t)k * itot;
                kc[c] = (float)k_ext_cloud;
                ssa[c] = (float)((ssa_gas*k_ext_gas + ssa_cloud*k_ext_cloud) / (k_ext_gas + k_ext_cloud));
                asy[c] = (float)((asy_gas*ssa_gas*k_ext_gas + asy_cloud*ssa_cloud*k_ext_cloud)
                                 / (ssa_gas*k_ext_gas + ssa_cloud*k_ext_cloud));
            }

    try
    {
        rt_context* ctx = nullptr;
        drv::check(rt_create(0, itot, 1, ktot, (float)dx_grid, (float)dx_grid, (float)dx_grid, &ctx), "rt_create");
        drv::check(rt_upload_optics(ctx, kg.data(), kc.data(), ssa.data(), asy.data()), "rt_upload_optics");
        drv::check(rt_set_option(ctx, "planar", 1), "rt_set_option");
        drv::check(rt_set_scene(ctx, (float)(zenith_deg * (M_PI / 180.)), 0.f, (float)albedo), "rt_set_scene");
        auto t0 = std::chrono::high_resolution_clock::now();
        drv::check(rt_trace(ctx, seed, 0, n_photons), "rt_trace");
        drv::check(rt_synchronize(ctx), "rt_synchronize");
        auto t1 = std::chrono::high_resolution_clock::now();
        std::cout << "Duration: " << std::setprecision(5) << std::chrono::duration<double>(t1 - t0).count()
                  << " (s)" << std::endl;                                             // raytracer.cpp:398
        std::vector<std::vector<uint64_t>> host(RT_N_TALLIES);
        uint64_t* ptrs[RT_N_TALLIES];
        for (int t = 0; t < RT_N_TALLIES; ++t)
        {
            host[t].resize(t < RT_ATMOS_DIRECT ? (size_t)itot : n);
            ptrs[t] = host[t].data();
        }
        drv::check(rt_download_tallies(ctx, ptrs), "rt_download_tallies");
        rt_destroy(ctx);
        for (size_t c = 0; c < n; ++c) host[RT_ATMOS_DIRECT][c] += host[RT_ATMOS_DIFFUSE][c];   // one atmos array in 2D
        for (int t = 0; t < RT_ATMOS_DIRECT; ++t)                                     // raytracer.cpp:416-420
            drv::save_binary(outdir, drv::kTallyNames[t], host[t].data(), host[t].size());
        drv::save_binary(outdir, "atmos", host[RT_ATMOS_DIRECT].data(), n);           // raytracer.cpp:421
    }
    catch (const std::exception& e)
    {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
