// oracle/cpu_ref_2d.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's 2D photon tracer (src/raytracer.cpp),
// generalised from its hard-coded block cloud to arbitrary x-z grids. Same arithmetic (double), same
// generator (xoroshiro128+ seeded by SplitMix64), same draw order and the same two-sweep 65 536-photon
// batch loop, so that with one thread and the built-in scene it reproduces the compiled reference
// (oracle/_ref/raytracer_2d_ref) bit for bit: tests/test_oracle_2d.py checks that against the
// reference built here and against the digests in tests/golden/cpu_ref_2d_100k.json.
//
// What differs from the 3D tracer (raytracer_3d.cpp) and is kept: photons live in the x-z plane;
// a scattering rotates the direction by +-acos(mu) IN that plane (raytracer.cpp:360-364, the sign is a
// random bit); the surface reflects to (mu*sign, sqrt(1-mu^2)) (:280-282); one global majorant; the
// level index divides z by dx (:339, the grid is square); no direct/diffuse split of the absorptions.
//
// Reference lines followed (relative to /root/reference/):
//   generator ............ src/raytracer.cpp:13-63      phase functions ... :88-103
//   free path ............ src/raytracer.cpp:106-109    launch ............ :112-122, 200-208
//   batch loop ........... src/raytracer.cpp:210-394
//
// Only tests/ may link or call this file. The product (raytracer_amd/) never does.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
struct Xoroshiro2
{
    uint64_t a, b;
    static uint64_t splitmix(uint64_t& s)
    {
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    // the reference constructor takes `const int seed` and widens it (sign-extending) to uint64
    explicit Xoroshiro2(int seed)
    {
        uint64_t s = static_cast<uint64_t>(static_cast<int64_t>(seed));
        a = splitmix(s);
        b = splitmix(s);
    }
    static uint64_t rot(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t s0 = a;
        uint64_t s1 = b;
        const uint64_t out = s0 + s1;
        s1 ^= s0;
        a = rot(s0, 24) ^ s1 ^ (s1 << 16);
        b = rot(s1, 37);
        return out;
    }
    double uniform() { return static_cast<double>(next() >> 11) * 0x1.0p-53; }
    double sign() { return static_cast<double>(-1 + 2*int(next() >> 63)); }       // raytracer.cpp:58
};

inline double sq(double v) { return v*v; }

inline double rayleigh_mu(double r)      // raytracer.cpp:88-94
{
    const double q = 4.*r - 2.;
    const double d = 1. + sq(q);
    const double u = std::pow(-q + std::sqrt(d), 1./3.);
    return u - 1./u;
}

inline double henyey_mu(double g, double r)      // raytracer.cpp:97-103
{
    const double a = sq(1. - sq(g));
    const double b = 2.*g*sq(2.*r*g + 1. - g);
    const double c = -g/2. - 1./(2.*g);
    return -1.*(a/b) - c;
}

inline double free_path_tau(double r)    // raytracer.cpp:106-109
{
    return -1.*std::log(-r + 1.) + std::numeric_limits<double>::epsilon();
}

struct Walker2
{
    double x, z, ux, uz;
    bool diffuse, counted;
};
}  // namespace

extern "C" {

// Fields [k][i] (i fastest). Event counters when `events` is non-null: [0] transport steps,
// [1] collision tests, [2] real collisions, [3] scatters, [4] cloud scatters, [5] surface hits,
// [6] toa exits, [7] absorptions.
int cpu_ref2d_trace(int itot, int ktot, double dx,
                    const double* k_gas, const double* k_cloud, const double* ssa, const double* asy,
                    double albedo, double zenith_rad,
                    uint64_t n_photons, int nthreads, int serial_seed_mode,
                    uint64_t* toa_down, uint64_t* toa_up, uint64_t* sfc_dir, uint64_t* sfc_dif, uint64_t* sfc_up,
                    uint64_t* atmos, uint64_t* events)
{
    if (itot <= 0 || ktot <= 0 || !k_gas || !k_cloud || !ssa || !asy)
        return 1;
    const double xs = itot*dx, zs = ktot*dx;
    const size_t ncell = static_cast<size_t>(itot)*ktot;
    std::vector<double> k_ext(ncell);
    for (size_t c = 0; c < ncell; ++c) k_ext[c] = k_gas[c] + k_cloud[c];
    const double k_null = *std::max_element(k_ext.begin(), k_ext.end());      // raytracer.cpp:176

    constexpr int batch = 1 << 16;                                            // raytracer.cpp:141
    std::vector<Walker2> pool(batch);
    uint64_t n_in = batch, n_out = 0;                                         // raytracer.cpp:193-194
    uint64_t ev_step = 0, ev_coll = 0, ev_real = 0, ev_scat = 0, ev_cscat = 0, ev_sfc = 0, ev_toa = 0, ev_abs = 0;

    auto launch = [&](Walker2& w, double r)                                   // raytracer.cpp:112-122
    {
        w.x = xs * r; w.z = zs;
        w.ux = -std::sin(zenith_rad); w.uz = -std::cos(zenith_rad);
        w.diffuse = false; w.counted = true;
    };
    auto bump = [](uint64_t* p)
    {
        #pragma omp atomic
        ++*p;
    };

#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    #pragma omp parallel reduction(+:ev_step,ev_coll,ev_real,ev_scat,ev_cscat,ev_sfc,ev_toa,ev_abs)
    {
#ifdef _OPENMP
        Xoroshiro2 g(serial_seed_mode ? static_cast<int>(3849284923u) : omp_get_thread_num());
#else
        Xoroshiro2 g(static_cast<int>(3849284923u));
#endif
        #pragma omp for
        for (int n = 0; n < batch; ++n)
        {
            launch(pool[n], g.uniform());
            bump(&toa_down[static_cast<int>(pool[n].x / dx)]);
        }

        while ((n_in < n_photons) || (n_in > n_out))
        {
            const bool done = n_in >= n_photons;

            #pragma omp for reduction(+:n_in) reduction(+:n_out)
            for (int n = 0; n < batch; ++n)
            {
                Walker2& w = pool[n];
                while (true)
                {
                    ++ev_step;
                    const double dn = free_path_tau(g.uniform()) / k_null;
                    double ddx = w.ux * dn, ddz = w.uz * dn;
                    bool at_surface = false, at_toa = false;
                    if ((w.z + ddz) <= 0.)
                    {
                        const double fac = std::abs(w.z / ddz);
                        ddx *= fac; ddz *= fac;
                        at_surface = true;
                    }
                    else if ((w.z + ddz) >= zs)
                    {
                        const double fac = std::abs((zs - w.z) / ddz);
                        ddx *= fac; ddz *= fac;
                        at_toa = true;
                    }
                    w.x += ddx; w.z += ddz;
                    w.x = std::fmod(w.x, xs);
                    if (w.x < 0.) w.x += xs;
                    const int i = static_cast<int>(w.x / dx);

                    auto relaunch = [&]()
                    {
                        launch(w, g.uniform());
                        if (done) w.counted = false;
                        if (w.counted)
                        {
                            ++n_in;
                            bump(&toa_down[static_cast<int>(w.x / dx)]);
                        }
                    };

                    if (at_surface)
                    {
                        ++ev_sfc;
                        if (w.counted)
                        {
                            ++n_out;
                            bump(w.diffuse ? &sfc_dif[i] : &sfc_dir[i]);
                        }
                        if (g.uniform() <= albedo)
                        {
                            if (w.counted)
                            {
                                --n_out;
                                bump(&sfc_up[i]);
                            }
                            const double mu = std::sqrt(g.uniform());
                            w.ux = mu * g.sign();
                            w.uz = std::sqrt(1. - mu*mu);
                            w.diffuse = true;
                        }
                        else
                            relaunch();
                    }
                    else if (at_toa)
                    {
                        ++ev_toa;
                        if (w.counted)
                        {
                            ++n_out;
                            bump(&toa_up[i]);
                        }
                        relaunch();
                    }
                    else
                        break;
                }
            }

            #pragma omp for reduction(+:n_in) reduction(+:n_out)
            for (int n = 0; n < batch; ++n)
            {
                Walker2& w = pool[n];
                const int i = static_cast<int>(w.x / dx);
                const int k = static_cast<int>(w.z / dx);
                const size_t c = static_cast<size_t>(i) + static_cast<size_t>(k)*itot;
                const double r = g.uniform();
                ++ev_coll;
                if (r >= (k_ext[c] / k_null))
                {
                }
                else if (r <= ssa[c] * k_ext[c] / k_null)
                {
                    ++ev_real; ++ev_scat;
                    const bool cloud = g.uniform() < (k_ext[c] - k_gas[c]) / k_ext[c];
                    if (cloud) ++ev_cscat;
                    const double mu = cloud ? henyey_mu(asy[c], g.uniform()) : rayleigh_mu(g.uniform());
                    const double angle = g.sign() * std::acos(mu) + std::atan2(w.ux, w.uz);
                    w.ux = std::sin(angle);
                    w.uz = std::cos(angle);
                    w.diffuse = true;
                }
                else
                {
                    ++ev_real; ++ev_abs;
                    if (w.counted)
                    {
                        ++n_out;
                        bump(&atmos[c]);
                    }
                    launch(w, g.uniform());
                    if (done) w.counted = false;
                    if (w.counted)
                    {
                        ++n_in;
                        bump(&toa_down[static_cast<int>(w.x / dx)]);
                    }
                }
            }
        }
    }
    if (events)
    {
        const uint64_t ev[8] = {ev_step, ev_coll, ev_real, ev_scat, ev_cscat, ev_sfc, ev_toa, ev_abs};
        for (int e = 0; e < 8; ++e) events[e] = ev[e];
    }
    return 0;
}

}  // extern "C"
