// oracle/cpu_ref.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's 3D CPU photon tracer (src/raytracer_3d.cpp),
// generalised from its hard-coded block cloud to arbitrary optical-property grids.
// Same arithmetic (double), same generator (xoroshiro128+ seeded by SplitMix64),
// same draw order and the same two-sweep 65 536-photon batch loop, so that with
// one thread and the block-cloud scene it reproduces the compiled reference
// (oracle/_ref/raytracer_3d_ref) bit for bit. That pin is checked by
// tests/test_oracle_cpu_ref.py against the reference built here and against the
// digests in tests/golden/cpu_ref_100k.json.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
// call this file. The product (raytracer_amd/) never does.
//
// Reference lines followed (relative to /root/reference/):
//   generator ............ src/raytracer_3d.cpp:14-65
//   vector helpers ....... src/raytracer_3d.cpp:76-104  (dot() keeps the v1.z*v1.z term of :87)
//   phase functions ...... src/raytracer_3d.cpp:123-138
//   free path ............ src/raytracer_3d.cpp:141-144
//   launch ............... src/raytracer_3d.cpp:147-161, 252-266
//   batch loop ........... src/raytracer_3d.cpp:268-508

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
// ---- xoroshiro128+ / SplitMix64 (raytracer_3d.cpp:14-65) -------------------------------
struct Xoroshiro
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

    // The reference constructor takes `const int seed` and widens it (sign-extending) to uint64.
    explicit Xoroshiro(int seed)
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
};

struct V3 { double x, y, z; };

// raytracer_3d.cpp:85-88. The third term squares v1.z instead of multiplying by v2.z. Both
// reference tracers (CPU and CUDA) do this, so the oracle has to as well.
inline double dot_as_reference(const V3& p, const V3& q) { return p.x*q.x + p.y*q.y + p.z*p.z; }

inline V3 cross(const V3& p, const V3& q)
{
    return V3{p.y*q.z - p.z*q.y, p.z*q.x - p.x*q.z, p.x*q.y - p.y*q.x};
}

inline double sq(double v) { return v*v; }

// raytracer_3d.cpp:123-129
inline double rayleigh_mu(double r)
{
    const double q = 4.*r - 2.;
    const double d = 1. + sq(q);
    const double u = std::pow(-q + std::sqrt(d), 1./3.);
    return u - 1./u;
}

// raytracer_3d.cpp:132-138 (g == 0 divides by zero there too; callers only reach it with g > 0
// in the scenes the tests use)
inline double henyey_mu(double g, double r)
{
    const double a = sq(1. - sq(g));
    const double b = 2.*g*sq(2.*r*g + 1. - g);
    const double c = -g/2. - 1./(2.*g);
    return -1.*(a/b) - c;
}

// raytracer_3d.cpp:141-144 — epsilon is added outside the log in the CPU tracer.
inline double free_path_tau(double r)
{
    return -1.*std::log(-r + 1.) + std::numeric_limits<double>::epsilon();
}

struct Walker
{
    V3 p, d;
    bool diffuse;   // Photon_kind
    bool counted;   // Photon_status::Enabled
};

struct Scene
{
    int itot, jtot, ktot;
    double dx, dy, dz, xs, ys, zs;
    const double* k_gas;
    const double* k_cloud;
    const double* ssa;
    const double* asy;
    double k_null, albedo, zen, azi;
};

struct Tallies
{
    uint64_t *toa_down, *toa_up, *sfc_dir, *sfc_dif, *sfc_up, *atm_dir, *atm_dif;
};

inline int clampi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// raytracer_3d.cpp:147-161
inline void to_toa(Walker& w, const Scene& s, double rx, double ry)
{
    w.p = V3{s.xs * rx, s.ys * ry, s.zs};
    w.d = V3{-std::sin(s.zen)*std::cos(s.azi), -std::sin(s.zen)*std::sin(s.azi), -std::cos(s.zen)};
    w.diffuse = false;
    w.counted = true;
}

inline int column_of(const Walker& w, const Scene& s)
{
    const int i = clampi(static_cast<int>(w.p.x / s.dx), s.itot);
    const int j = clampi(static_cast<int>(w.p.y / s.dy), s.jtot);
    return i + j*s.itot;
}

inline void bump(uint64_t* cell)
{
    #pragma omp atomic
    ++*cell;
}

// Relaunch after a sink (raytracer_3d.cpp:364-381, 394-412, 488-505).
inline void relaunch(Walker& w, const Scene& s, Xoroshiro& g, bool generation_done,
                     const Tallies& t, uint64_t& n_in)
{
    // The reference passes rg.fp64(), rg.fp64() as two call arguments (raytracer_3d.cpp:365);
    // g++ evaluates them right to left, so the y draw comes first. Kept: it fixes the stream.
    const double ry = g.uniform();
    const double rx = g.uniform();
    to_toa(w, s, rx, ry);
    if (generation_done)
        w.counted = false;
    if (w.counted)
    {
        ++n_in;
        bump(&t.toa_down[column_of(w, s)]);
    }
}
} // namespace


extern "C" {

// Event counters filled when `events` is non-null:
// [0] transport steps, [1] collision tests, [2] real collisions, [3] scatters, [4] cloud scatters,
// [5] surface hits, [6] toa exits, [7] absorptions  (counted photons and the untallied tail alike)
int cpu_ref_trace(
        int itot, int jtot, int ktot, double dx, double dy, double dz,
        const double* k_gas, const double* k_cloud, const double* ssa, const double* asy,
        double albedo, double zenith_rad, double azimuth_rad,
        uint64_t n_photons, int nthreads, int serial_seed_mode,
        uint64_t* toa_down, uint64_t* toa_up,
        uint64_t* sfc_dir, uint64_t* sfc_dif, uint64_t* sfc_up,
        uint64_t* atm_dir, uint64_t* atm_dif,
        uint64_t* events)
{
    if (itot <= 0 || jtot <= 0 || ktot <= 0 || !k_gas || !k_cloud || !ssa || !asy)
        return 1;

    Scene s;
    s.itot = itot; s.jtot = jtot; s.ktot = ktot;
    s.dx = dx; s.dy = dy; s.dz = dz;
    s.xs = itot*dx; s.ys = jtot*dy; s.zs = ktot*dz;
    s.k_gas = k_gas; s.k_cloud = k_cloud; s.ssa = ssa; s.asy = asy;
    s.albedo = albedo; s.zen = zenith_rad; s.azi = azimuth_rad;

    const size_t ncell = static_cast<size_t>(itot)*jtot*ktot;
    // raytracer_3d.cpp:223 — one global majorant.
    double kmax = 0.;
    for (size_t c = 0; c < ncell; ++c)
        kmax = std::max(kmax, k_gas[c] + k_cloud[c]);
    s.k_null = kmax;

    const Tallies t{toa_down, toa_up, sfc_dir, sfc_dif, sfc_up, atm_dir, atm_dif};

    constexpr int batch = 1 << 16;                      // raytracer_3d.cpp:184
    std::vector<Walker> pool(batch);

    uint64_t n_in = batch;                              // raytracer_3d.cpp:241-242
    uint64_t n_out = 0;
    uint64_t ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};

#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif

    #pragma omp parallel
    {
#ifdef _OPENMP
        Xoroshiro g(serial_seed_mode ? static_cast<int>(3849284923u) : omp_get_thread_num());
#else
        Xoroshiro g(serial_seed_mode ? static_cast<int>(3849284923u) : 0);
#endif
        uint64_t e_step = 0, e_coll = 0, e_real = 0, e_scat = 0, e_cscat = 0, e_sfc = 0, e_toa = 0, e_abs = 0;

        // initial fill, raytracer_3d.cpp:252-266
        #pragma omp for
        for (int n = 0; n < batch; ++n)
        {
            const double ry = g.uniform();   // right-to-left argument evaluation, see relaunch()
            const double rx = g.uniform();
            to_toa(pool[n], s, rx, ry);
            bump(&t.toa_down[column_of(pool[n], s)]);
        }

        while ((n_in < n_photons) || (n_in > n_out))
        {
            const bool generation_done = n_in >= n_photons;

            // transport sweep, raytracer_3d.cpp:273-419
            #pragma omp for reduction(+:n_in) reduction(+:n_out)
            for (int n = 0; n < batch; ++n)
            {
                Walker& w = pool[n];
                for (;;)
                {
                    ++e_step;
                    const double dn = free_path_tau(g.uniform()) / s.k_null;
                    double sx = w.d.x * dn;
                    double sy = w.d.y * dn;
                    double sz = w.d.z * dn;

                    bool hit_surface = false, hit_toa = false;
                    if ((w.p.z + sz) <= 0.)
                    {
                        const double f = std::abs(w.p.z / sz);
                        sx *= f; sy *= f; sz *= f;
                        hit_surface = true;
                    }
                    else if ((w.p.z + sz) >= s.zs)
                    {
                        const double f = std::abs((s.zs - w.p.z) / sz);
                        sx *= f; sy *= f; sz *= f;
                        hit_toa = true;
                    }

                    w.p.x += sx; w.p.y += sy; w.p.z += sz;

                    w.p.x = std::fmod(w.p.x, s.xs);
                    if (w.p.x < 0.) w.p.x += s.xs;
                    w.p.y = std::fmod(w.p.y, s.ys);
                    if (w.p.y < 0.) w.p.y += s.ys;

                    if (hit_surface)
                    {
                        ++e_sfc;
                        const int ij = column_of(w, s);
                        if (w.counted)
                        {
                            ++n_out;
                            bump(w.diffuse ? &t.sfc_dif[ij] : &t.sfc_dir[ij]);
                        }
                        if (g.uniform() <= s.albedo)
                        {
                            if (w.counted)
                            {
                                --n_out;
                                bump(&t.sfc_up[ij]);
                            }
                            const double mu = std::sqrt(g.uniform());
                            const double az = 2.*M_PI*g.uniform();
                            w.d.x = mu*std::sin(az);
                            w.d.y = mu*std::cos(az);
                            w.d.z = std::sqrt(1. - mu*mu);
                            w.diffuse = true;
                        }
                        else
                            relaunch(w, s, g, generation_done, t, n_in);
                    }
                    else if (hit_toa)
                    {
                        ++e_toa;
                        if (w.counted)
                        {
                            ++n_out;
                            bump(&t.toa_up[column_of(w, s)]);
                        }
                        relaunch(w, s, g, generation_done, t, n_in);
                    }
                    else
                        break;
                }
            }

            // collision sweep, raytracer_3d.cpp:422-507
            #pragma omp for reduction(+:n_in) reduction(+:n_out)
            for (int n = 0; n < batch; ++n)
            {
                Walker& w = pool[n];
                ++e_coll;
                const int i = clampi(static_cast<int>(w.p.x / s.dx), s.itot);
                const int j = clampi(static_cast<int>(w.p.y / s.dy), s.jtot);
                const int k = clampi(static_cast<int>(w.p.z / s.dz), s.ktot);
                const size_t c = static_cast<size_t>(i) + static_cast<size_t>(j)*s.itot
                               + static_cast<size_t>(k)*s.itot*s.jtot;

                const double kc = s.k_gas[c] + s.k_cloud[c];
                const double r = g.uniform();

                if (r >= (kc / s.k_null))
                {
                    // null collision
                }
                else if (r <= s.ssa[c] * kc / s.k_null)
                {
                    ++e_real; ++e_scat;
                    const bool from_cloud = g.uniform() < (kc - s.k_gas[c]) / kc;
                    if (from_cloud) ++e_cscat;
                    const double mu = from_cloud ? henyey_mu(s.asy[c], g.uniform()) : rayleigh_mu(g.uniform());
                    const double sn = std::sqrt(1. - mu*mu);

                    V3 t1{0., 0., 0.};
                    if (std::fabs(w.d.x) < std::fabs(w.d.y))
                    {
                        if (std::fabs(w.d.x) < std::fabs(w.d.z)) t1.x = 1; else t1.z = 1;
                    }
                    else
                    {
                        if (std::fabs(w.d.y) < std::fabs(w.d.z)) t1.y = 1; else t1.z = 1;
                    }
                    const double proj = dot_as_reference(t1, w.d);
                    V3 o{t1.x - proj*w.d.x, t1.y - proj*w.d.y, t1.z - proj*w.d.z};
                    const double len = std::sqrt(o.x*o.x + o.y*o.y + o.z*o.z);
                    t1 = V3{o.x/len, o.y/len, o.z/len};
                    const V3 t2 = cross(w.d, t1);

                    const double phi = 2.*M_PI*g.uniform();
                    const double sp = std::sin(phi), cp = std::cos(phi);
                    w.d = V3{mu*w.d.x + sn*(sp*t1.x + cp*t2.x),
                             mu*w.d.y + sn*(sp*t1.y + cp*t2.y),
                             mu*w.d.z + sn*(sp*t1.z + cp*t2.z)};
                    w.diffuse = true;
                }
                else
                {
                    ++e_real; ++e_abs;
                    if (w.counted)
                    {
                        ++n_out;
                        bump(w.diffuse ? &t.atm_dif[c] : &t.atm_dir[c]);
                    }
                    relaunch(w, s, g, generation_done, t, n_in);
                }
            }
        }

        #pragma omp critical
        {
            ev[0] += e_step; ev[1] += e_coll; ev[2] += e_real; ev[3] += e_scat;
            ev[4] += e_cscat; ev[5] += e_sfc; ev[6] += e_toa; ev[7] += e_abs;
        }
    }

    if (events)
        std::memcpy(events, ev, sizeof(ev));
    return 0;
}

} // extern "C"
