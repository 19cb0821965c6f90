/* oracle/photon_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement, in plain C, of the photon life cycle of the reference's CUDA kernel
 * (src/raytracer_kernels.cu:307-569) in the arithmetic the MI355X path computes in:
 * float32, a counter-based Philox4x32-10 stream keyed on the photon ID (north_star), layered
 * majorants for null-collision tracking. The HIP kernels in raytracer_amd/csrc must reproduce
 * the tallies of this file BIT FOR BIT on the same inputs (tests/test_gpu_parity.py); this file
 * itself is pinned to the reference statistically (tests/test_oracle_photon.py: global fractions
 * and per-cell z-scores against oracle/cpu_ref.cpp, which in turn is pinned bit-exact to the
 * compiled reference). The cuRAND streams of the CUDA path cannot be reproduced (cuRAND is not
 * in /root/reference and not in this image) — RNG-stream parity is unpinned by design, see
 * DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 *
 * Every transcendental is built from IEEE-754 correctly rounded operations only (+ - * / sqrt
 * fma), never libm, so CPU and GPU agree to the bit. The file must be compiled with
 * -ffp-contract=off; every fused multiply-add is written out as fmaf().
 *
 * Reference lines restated (relative to /root/reference/src/):
 *   launch at TOA ............ raytracer_kernels.cu:180-208
 *   free path ................ raytracer_kernels.cu:165-170, 367
 *   slab majorants/transition  raytracer_kernels.cu:360-415   (restated as explicit region state,
 *                              parametric distance to the next region face and an optical depth
 *                              carried across faces; the cloud slab optionally refined into a
 *                              grid of local majorants)
 *   surface / TOA ............ raytracer_kernels.cu:417-434, 455-499
 *   periodic wrap ............ raytracer_kernels.cu:440-448
 *   cell index ............... raytracer_kernels.cu:173-178
 *   collision classifier ..... raytracer_kernels.cu:507-520
 *   Rayleigh / HG ............ raytracer_kernels.cu:145-162
 *   scatter frame ............ raytracer_kernels.cu:523-548 with dot() of :70-74
 *   Lambertian surface ....... raytracer_kernels.cu:463-474
 *   majorants from the grids . raytracer_3d_io.cu:60-77
 *   cloud slab bounds ........ raytracer_kernels.cu:263-305  (its four indexing bugs not kept)
 */

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define RT_EPS 1.1920928955078125e-07f /* FLT_EPSILON, raytracer_kernels.cu:24 */
#define RT_TWO_PI 6.2831854820251465f   /* Float(2.*M_PI) */
#define RT_MAX_STEPS (1u << 24) /* guard against a photon that never ends: the reference has none; the kernels' guards are this
                                   one (lane-resident kernel) or looser (pool kernel, by rounds of the wave) */


/* ---------------------------------------------------------------- Philox4x32-10 ---------- */
/* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11). */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round)
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* one block per event: counter = (photon id low, photon id high, event index, 0), key = seed */
static inline void draw(uint64_t seed, uint64_t pid, uint32_t step, uint32_t tag, uint32_t w[4])
{
    const uint32_t ctr[4] = {(uint32_t)pid, (uint32_t)(pid >> 32), step, tag};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    oracle_philox4x32_10(ctr, key, w);
}

/* top 24 bits -> [0,1), every value exactly representable */
static inline float u01(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }

/* The two 32-bit words a photon's launch position is scaled from under the Sobol launch (what curand(&state_x),
 * curand(&state_y) return in raytracer_kernels.cu:248-249): point (pid mod 2^32), in natural index order, of the
 * two-dimensional Sobol sequence with the Joe-Kuo direction numbers of dimensions 0 and 1 (the tables the reference
 * copies from cuRAND, raytracer_3d.cu:135-140), by the textbook recurrence: dimension 0 has v_k = 2^(31-k) (the
 * radical inverse), dimension 1 v_0 = 2^31, v_k = v_(k-1) ^ (v_(k-1) >> 1). shifted != 0: XORed with the digital
 * shift of the photon's block of 2^32 IDs, a Philox draw keyed on the seed (cuRAND's scramble constants,
 * raytracer_3d.cu:141-145, enter its state by the same XOR). tests/test_oracle_photon.py pins the unshifted words
 * to scipy.stats.qmc.Sobol (same direction numbers, Gray-code order). */
void oracle_sobol_words(uint64_t seed, uint64_t pid, int shifted, uint32_t out[2])
{
    uint32_t s0 = 0u, s1 = 0u, v0 = 0x80000000u, v1 = 0x80000000u;
    if (shifted)
    {
        uint32_t w[4];
        draw(seed, pid >> 32, 0xFFFFFFFFu, 2u, w);
        s0 = w[0]; s1 = w[1];
    }
    for (uint32_t n = (uint32_t)pid; n; n >>= 1, v0 >>= 1, v1 ^= v1 >> 1)
        if (n & 1u) { s0 ^= v0; s1 ^= v1; }
    out[0] = s0; out[1] = s1;
}

/* ---------------------------------------------------------------- deterministic math ----- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* natural log for positive normal x; mantissa reduced to [sqrt(1/2), sqrt(2)), degree-9
 * polynomial (Cephes logf coefficients), Horner in fmaf. */
float oracle_logf(float x)
{
    uint32_t ix = f2u(x);
    ix += 0x3f800000u - 0x3f3504f3u;
    const int e = (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float f = u2f(ix) - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(fe, 0.693359375f, f + y);
}

/* sin and cos of 2*pi*u for u in [0,1) on the 2^-24 lattice: exact quadrant reduction, then
 * degree-7 / degree-8 polynomials on [-pi/4, pi/4] (Cephes sinf/cosf coefficients). */
void oracle_sincos2pi(float u, float* s_out, float* c_out)
{
    const int q = (int)fmaf(u, 4.0f, 0.5f);
    const float r = u - 0.25f * (float)q;          /* exact */
    const float x = r * RT_TWO_PI;
    const float x2 = x * x;
    float ps = -1.9515295891e-4f;
    ps = fmaf(ps, x2, 8.3321608736e-3f);
    ps = fmaf(ps, x2, -1.6666654611e-1f);
    const float s = fmaf(x * x2, ps, x);
    float pc = 2.443315711809948e-5f;
    pc = fmaf(pc, x2, -1.388731625493765e-3f);
    pc = fmaf(pc, x2, 4.166664568298827e-2f);
    const float c = fmaf(x2 * x2, pc, fmaf(-0.5f, x2, 1.0f));
    switch (q & 3)
    {
        case 0: *s_out = s;  *c_out = c;  break;
        case 1: *s_out = c;  *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}

/* x^(-1/3) of a positive normal float: bit-shift seed + three division-free Newton steps y <- y (4 - x y^3) / 3 */
float oracle_inv_cbrtf(float x)
{
    float y = u2f(0x54a2fa8cu - f2u(x) / 3u);
    const float x3 = x * 0.3333333432674408f;
    for (int it = 0; it < 3; ++it)
    {
        const float y3 = (y * y) * y;
        y = y * fmaf(-x3, y3, 1.3333333730697632f);
    }
    return y;
}

/* raytracer_kernels.cu:145-152: mu = u - 1/u, u = pow(v, 1/3), v = -q + sqrt(q^2 + 1), written with
 * y = v^(-1/3) (u = v y^2, 1/u = y): mu = y (v y - 1) */
static inline float rayleigh_mu(float r)
{
    const float q = fmaf(4.0f, r, -2.0f);
    const float d = fmaf(q, q, 1.0f);
    const float v = -q + sqrtf(d);
    const float y = oracle_inv_cbrtf(v);
    return y * fmaf(v, y, -1.0f);
}

/* raytracer_kernels.cu:155-162. The reference divides by g; |g| < 1e-3 takes the g -> 0 limit
 * (isotropic) instead of producing NaN (SURVEY.md §7 "float edge cases"). */
static inline float henyey_mu(float g, float r)
{
    if (fabsf(g) < 1.0e-3f)
        return fmaf(2.0f, r, -1.0f);
    const float one_m_g2 = 1.0f - g * g;
    const float a = one_m_g2 * one_m_g2;
    const float h = fmaf(2.0f * r, g, 1.0f - g);
    const float h2 = h * h;
    const float rb = 1.0f / ((2.0f * g) * h2);       /* 1/b; then 1/(2g) = h^2 / b: one reciprocal, no division */
    const float c = -g * 0.5f - h2 * rb;
    return -(a * rb) - c;
}

/* exported for the sampler-moment tests */
float oracle_rayleigh_mu(float r) { return rayleigh_mu(r); }
float oracle_henyey_mu(float g, float r) { return henyey_mu(g, r); }

/* raytracer_kernels.cu:165-170 */
static inline float free_path_tau(float r) { return -oracle_logf((1.0f - r) + RT_EPS); }

/* raytracer_kernels.cu:440-448, fmod restated as floor-multiply so the result is bit-defined */
static inline float wrap_periodic(float x, float size, float inv_size)
{
    const float q = floorf(x * inv_size);
    x = fmaf(-q, size, x);
    if (x < 0.0f) x += size;
    if (x >= size) x -= size;
    return x;
}

/* raytracer_kernels.cu:173-178 with s/ds as s*(1/ds), clamped on both sides */
static inline int cell_of(float s, float inv_ds, int n)
{
    int c = (int)(s * inv_ds);
    if (c > n - 1) c = n - 1;
    if (c < 0) c = 0;
    return c;
}

/* ---------------------------------------------------------------- scene preparation ------ */
typedef struct
{
    int itot, jtot, ktot;
    float dx, dy, dz;
    float zenith_rad, azimuth_rad, albedo;
    int fixed_dot; /* 0 = dot() as the reference writes it (v1.z*v1.z), 1 = v1.z*v2.z */
    int sx, sy, sz; /* supervoxel size in cells for the majorant grid of the cloud slab; 0 = none */
    int launch_log2k; /* -1: launch positions uniform over the domain (one Philox draw per axis); -2: the Sobol
                         sequence indexed by the photon ID, digitally shifted (the reference's launch);
                         >= 0: stratified by column, 2^launch_log2k consecutive photon IDs per column */
    int planar; /* 1: the scattering rule of the reference's 2D tracer — directions stay in the x-z plane:
                   a scattering turns the direction by +-acos(mu) in that plane (raytracer.cpp:360-364), the
                   surface reflects to (+-mu, 0, sqrt(1-mu^2)) (raytracer.cpp:280-282); the signs are the top
                   bit of the draw that is the azimuth in 3D. Meant for jtot = 1 and azimuth = 0. */
    int levels; /* every supervoxel gets a box of its own, up to 2^levels supervoxels per axis, that a photon standing in
                   it crosses as ONE majorant region (0 = the supervoxel alone) */
    float merge_tau; /* a box grows by a layer while (majorant - mean extinction) * 6 V / S <= merge_tau: the optical
                        depth of null collisions a photon crossing the box would meet. Clear air grows at any size,
                        homogeneous cloud too; cloud edges keep their fine supervoxels. */
    int launch_tile_log2; /* stratified launch: 0 = columns in row-major order; T > 0 = columns counted tile by tile
                             (tiles of 2^T x 2^T columns, row-major inside, tiles row by row; itot and jtot are
                             multiples of 2^T) */
} oracle_scene_t;

/* Majorant structure. Horizontal slabs as in the reference (below / in / above the cloud layer,
 * raytracer_kernels.cu:360-362), one majorant each; optionally the cloud slab is refined into a
 * grid of supervoxels (sx*sy*sz cells) with local majorants (north_star: "voxel-DDA traversal",
 * SURVEY.md §7 "majorant super-grid"). */
typedef struct
{
    int nslab;          /* 1..3 non-empty horizontal slabs, bottom to top */
    float zface[4];     /* nslab+1 faces, zface[0] = 0, zface[nslab] = z_size */
    float kmaj[3];      /* majorant extinction per slab */
    float inv_kmaj[3];
    int k_lo, k_hi;     /* cloud levels (asy > 0), -1 if none */
    float sun[3];       /* direction of the direct beam, raytracer_3d.cu:25-30 */
    int gslab;          /* index of the gridded slab, -1 if none */
    int nci, ncj, nck;  /* supervoxels per axis */
    float hx, hy, hz;   /* supervoxel edge lengths */
    float* maj;         /* nck*ncj*nci pairs {M, 1/M} of the BOX of a supervoxel; owned, freed by oracle_free_slabs */
    uint32_t* box;      /* nck*ncj*nci boxes: six extents of 5 bits, xm | xp << 5 | ym << 10 | yp << 15 | zm << 20 | zp << 25:
                           a photon in supervoxel (ci, cj, ck) crosses the supervoxels [ci - xm, ci + xp] x [cj - ym, cj + yp] x
                           [ck - zm, ck + zp] as one region */
} oracle_slabs_t;

void oracle_free_slabs(oracle_slabs_t* s) { free(s->maj); s->maj = NULL; free(s->box); s->box = NULL; }

/* 1/M as the tables hold it: finite. A region without any extinction has M = 0 (or, once the region level
 * is packed into M's low mantissa bits, a denormal) and 1/M = +inf; the traverse step computes tau * (1/M), which
 * is NaN for tau == 0 (one launch in 2^24 draws exactly 0). With FLT_MAX instead, tau == 0 stops the photon where
 * it is for a collision test that is null (k_tot = 0 there, and 0 * FLT_MAX = 0) and draws the next tau; any
 * tau > 0 still carries the photon through the whole region. */
static inline float majorant_reciprocal(float m)
{
    return fminf(1.0f / m, FLT_MAX);
}

/* 1/M of a box as the traverse steps take it from M (the product's region_reciprocal): the IEEE quotient for
 * M >= 2^-100, FLT_MAX below. */
static inline float region_reciprocal(float m)
{
    return m >= 0x1p-100f ? 1.0f / m : FLT_MAX;
}

/* Majorant regions over the supervoxel grid: every supervoxel gets a BOX OF ITS OWN -- the supervoxels around it, at
 * most max_extent per axis, clipped at the grid -- that a photon standing in it crosses as one region under the box's
 * majorant. Boxes of neighbours overlap: they are no partition, and need not be -- every step of the walk is made
 * under a majorant that bounds the extinction over the whole segment, and the optical depth carried over a face is
 * decremented with the majorant the segment was sampled with.
 * Growth: from the supervoxel alone, one layer at a time, trying the six faces in turn (x-, x+, y-, y+, z-, z+) until
 * six attempts in a row have failed; a layer is taken if (M - mean k_tot) * len <= tau for the grown box, M its
 * majorant, mean its volume mean, len = 6 V / S (the edge of a cube; three times the thickness of a wide slab): the
 * optical depth of null collisions a crossing photon meets on average, up to the factor 2/3 of the mean chord.
 * m0: majorant per supervoxel; ksum/kcnt: sum (double, cells in k, j, i order) and number of its cells. The layer's
 * supervoxels are visited in (k, j, i) order and the criterion is evaluated in double: the library's kernel
 * (rt_boxes_kernel) must decide identically. */
static void build_boxes(int nci, int ncj, int nck, int max_extent, const float* m0, const double* ksum, const int* kcnt,
                        float hx, float hy, float hz, float tau, float* maj, uint32_t* box)
{
    const long long n = (long long)nci * ncj * nck;
    const int dims[3] = {nci, ncj, nck};
    const double h[3] = {(double)hx, (double)hy, (double)hz};
#pragma omp parallel for schedule(dynamic, 256)
    for (long long e = 0; e < n; ++e)
    {
        const int ci = (int)(e % nci), cj = (int)((e / nci) % ncj), ck = (int)(e / ((long long)nci * ncj));
        int b[6] = {ci, ci + 1, cj, cj + 1, ck, ck + 1};
        const int c3[3] = {ci, cj, ck};
        float m = m0[e];
        double sum = ksum[e];
        long long cnt = kcnt[e];
        int fails = 0, dir = 0;
        while (fails < 6)
        {
            const int ax = dir >> 1, plus = dir & 1;
            int ok = plus ? b[2 * ax + 1] < dims[ax] : b[2 * ax] > 0;
            ok = ok && b[2 * ax + 1] - b[2 * ax] < max_extent && (plus ? b[2 * ax + 1] - 1 - c3[ax] : c3[ax] - b[2 * ax]) < 31;   /* 5-bit extents */
            if (ok)
            {
                int l[6] = {b[0], b[1], b[2], b[3], b[4], b[5]};        /* the new layer */
                if (plus) { l[2 * ax] = b[2 * ax + 1]; l[2 * ax + 1] = b[2 * ax + 1] + 1; }
                else { l[2 * ax] = b[2 * ax] - 1; l[2 * ax + 1] = b[2 * ax]; }
                float lm = 0.0f;
                double ls = 0.0;
                long long lc = 0;
                for (int kk = l[4]; kk < l[5]; ++kk)
                    for (int jj = l[2]; jj < l[3]; ++jj)
                        for (int ii = l[0]; ii < l[1]; ++ii)
                        {
                            const size_t q = ((size_t)kk * ncj + jj) * nci + ii;
                            if (m0[q] > lm) lm = m0[q];
                            ls += ksum[q];
                            lc += kcnt[q];
                        }
                const float nm = lm > m ? lm : m;
                const double nsum = sum + ls;
                const long long ncnt = cnt + lc;
                const double ex = (double)(b[1] - b[0] + (ax == 0 ? 1 : 0)) * h[0], ey = (double)(b[3] - b[2] + (ax == 1 ? 1 : 0)) * h[1],
                             ez = (double)(b[5] - b[4] + (ax == 2 ? 1 : 0)) * h[2];
                const double len = 3.0 * (ex * ey * ez) / (ex * ey + ey * ez + ex * ez);
                ok = ((double)nm - nsum / (double)ncnt) * len <= (double)tau;
                if (ok)
                {
                    if (plus) b[2 * ax + 1] += 1; else b[2 * ax] -= 1;
                    m = nm; sum = nsum; cnt = ncnt;
                }
            }
            fails = ok ? 0 : fails + 1;
            dir = dir == 5 ? 0 : dir + 1;
        }
        box[e] = (uint32_t)(ci - b[0]) | ((uint32_t)(b[1] - 1 - ci) << 5) | ((uint32_t)(cj - b[2]) << 10)
                 | ((uint32_t)(b[3] - 1 - cj) << 15) | ((uint32_t)(ck - b[4]) << 20) | ((uint32_t)(b[5] - 1 - ck) << 25);
        maj[2 * e] = m;
        maj[2 * e + 1] = region_reciprocal(m);
    }
}

/* Restates raytracer_kernels.cu:263-305 (cloud bounds: levels with any asy > 0) and
 * raytracer_3d_io.cu:60-77 (majorants), per slab: max over the slab's levels of gas + cloud. */
void oracle_build_slabs(const oracle_scene_t* sc, const float* k_gas, const float* k_cloud,
                        const float* asy, oracle_slabs_t* out)
{
    const size_t nlev = (size_t)sc->itot * sc->jtot;
    int k_lo = -1, k_hi = -1;
    float* lev_max = (float*)malloc(sizeof(float) * sc->ktot);
    for (int k = 0; k < sc->ktot; ++k)
    {
        float m = 0.0f;
        int cloudy = 0;
        for (size_t c = (size_t)k * nlev; c < (size_t)(k + 1) * nlev; ++c)
        {
            const float kt = k_gas[c] + k_cloud[c];
            if (kt > m) m = kt;
            if (asy[c] > 0.0f) cloudy = 1;
        }
        lev_max[k] = m;
        if (cloudy) { if (k_lo < 0) k_lo = k; k_hi = k; }
    }
    out->k_lo = k_lo; out->k_hi = k_hi;

    int lo[3], hi[3], n = 0, g = -1;
    if (k_lo < 0) { lo[0] = 0; hi[0] = sc->ktot; n = 1; }
    else
    {
        if (k_lo > 0) { lo[n] = 0; hi[n] = k_lo; ++n; }
        g = n;
        lo[n] = k_lo; hi[n] = k_hi + 1; ++n;
        if (k_hi + 1 < sc->ktot) { lo[n] = k_hi + 1; hi[n] = sc->ktot; ++n; }
    }
    out->nslab = n;
    for (int s = 0; s < 3; ++s) { out->kmaj[s] = 0.0f; out->inv_kmaj[s] = 0.0f; }
    for (int s = 0; s < n; ++s)
    {
        float m = 0.0f;
        for (int k = lo[s]; k < hi[s]; ++k) if (lev_max[k] > m) m = lev_max[k];
        out->kmaj[s] = m;
        out->inv_kmaj[s] = majorant_reciprocal(m);
        out->zface[s] = (float)lo[s] * sc->dz;
    }
    out->zface[n] = (float)sc->ktot * sc->dz;
    for (int s = n + 1; s < 4; ++s) out->zface[s] = out->zface[n];
    /* raytracer_3d.cu:25-30: angles are floats, cosines in float. Host libm is fine here: the
     * product computes the same three numbers on the host and hands them to the kernel, so they
     * are inputs of the traced path, not results of it. */
    out->sun[0] = -sinf(sc->zenith_rad) * cosf(sc->azimuth_rad);
    out->sun[1] = -sinf(sc->zenith_rad) * sinf(sc->azimuth_rad);
    out->sun[2] = -cosf(sc->zenith_rad);
    free(lev_max);

    /* majorant grid of the cloud slab */
    out->gslab = -1; out->nci = out->ncj = out->nck = 0; out->hx = out->hy = out->hz = 0.0f; out->maj = NULL;
    out->box = NULL;
    if (g >= 0 && sc->sx > 0 && sc->sy > 0 && sc->sz > 0)
    {
        const int nci = (sc->itot + sc->sx - 1) / sc->sx, ncj = (sc->jtot + sc->sy - 1) / sc->sy;
        const int nck = (k_hi + 1 - k_lo + sc->sz - 1) / sc->sz;
        out->gslab = g; out->nci = nci; out->ncj = ncj; out->nck = nck;
        out->hx = (float)sc->sx * sc->dx; out->hy = (float)sc->sy * sc->dy; out->hz = (float)sc->sz * sc->dz;
        out->maj = (float*)malloc(sizeof(float) * 2 * (size_t)nci * ncj * nck);
        out->box = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nci * ncj * nck);
        float* m0 = (float*)malloc(sizeof(float) * (size_t)nci * ncj * nck);
        double* ksum = (double*)malloc(sizeof(double) * (size_t)nci * ncj * nck);
        int* kcnt = (int*)malloc(sizeof(int) * (size_t)nci * ncj * nck);
        for (int ck = 0; ck < nck; ++ck)
            for (int cj = 0; cj < ncj; ++cj)
                for (int ci = 0; ci < nci; ++ci)
                {
                    float m = 0.0f;
                    double sum = 0.0;
                    int cnt = 0;
                    const int k1 = k_lo + (ck + 1) * sc->sz < k_hi + 1 ? k_lo + (ck + 1) * sc->sz : k_hi + 1;
                    const int j1 = (cj + 1) * sc->sy < sc->jtot ? (cj + 1) * sc->sy : sc->jtot;
                    const int i1 = (ci + 1) * sc->sx < sc->itot ? (ci + 1) * sc->sx : sc->itot;
                    for (int k = k_lo + ck * sc->sz; k < k1; ++k)
                        for (int j = cj * sc->sy; j < j1; ++j)
                            for (int i = ci * sc->sx; i < i1; ++i)
                            {
                                const size_t c = (size_t)i + (size_t)j * sc->itot + (size_t)k * nlev;
                                const float kt = k_gas[c] + k_cloud[c];
                                if (kt > m) m = kt;
                                sum += (double)kt;
                                ++cnt;
                            }
                    const size_t e = ((size_t)ck * ncj + cj) * nci + ci;
                    m0[e] = m;
                    ksum[e] = sum;
                    kcnt[e] = cnt;
                }
        build_boxes(nci, ncj, nck, 1 << (sc->levels > 0 ? sc->levels : 0), m0, ksum, kcnt, out->hx, out->hy, out->hz,
                    sc->merge_tau, out->maj, out->box);
        free(ksum); free(kcnt); free(m0);
    }
}

/* ---------------------------------------------------------------- the photon life cycle -- */
enum { EV_STEP = 0, EV_COLL, EV_REAL, EV_SCAT, EV_CSCAT, EV_SFC, EV_TOA, EV_ABS, EV_TRANS, EV_ABORT, EV_N };

typedef struct
{
    uint64_t *toa_down, *toa_up, *sfc_dir, *sfc_dif, *sfc_up, *atm_dir, *atm_dif;
} tallies_t;

static inline void bump(uint64_t* p)
{
#pragma omp atomic
    ++*p;
}

/* Reciprocal of a direction component as the traversal keeps it: finite (+-FLT_MAX for a component below
 * 2^-100 in size, zero included); its sign — not the component's — says which face of a region the photon heads
 * for, so (face - pos) * inv_u is never the 0 * inf or (-) * (+inf) of a zero component (a Lambert azimuth
 * draw of exactly 0 makes ux = +0). */
static inline float dir_reciprocal(float u)
{
    return fabsf(u) >= 0x1p-100f ? 1.0f / u : copysignf(FLT_MAX, u);
}

/* distance along the ray (in units of the direction vector) to the face the photon heads for; a
 * negative value (already past the face by rounding) -> 0 */
static inline float face_distance(float face, float pos, float inv_u)
{
    const float t = (face - pos) * inv_u;
    return t > 0.0f ? t : 0.0f;
}

/* Draw layout: ONE Philox block per event, counter = (pid_lo, pid_hi, event index, 0):
 *   launch:     w0 -> x, w1 -> y, w2 -> first optical depth
 *   collision:  w0 -> classifier r (and gas/cloud choice), w1 -> next optical depth,
 *               w2 -> scattering cosine, w3 -> scattering azimuth
 *   surface:    w0 -> albedo test, w1 -> Lambert cosine, w2 -> Lambert azimuth
 * The sampled optical depth is carried across majorant regions (decremented by M * t on every
 * face crossing), so crossings consume no random numbers. */
static void trace_one(const oracle_scene_t* sc, const oracle_slabs_t* sl,
                      const float* k_gas, const float* k_cloud, const float* ssa, const float* asy,
                      uint64_t seed, uint64_t pid, const tallies_t* t, uint64_t* ev)
{
    const int itot = sc->itot, jtot = sc->jtot, ktot = sc->ktot;
    const float xs = (float)itot * sc->dx, ys = (float)jtot * sc->dy;
    const float inv_xs = 1.0f / xs, inv_ys = 1.0f / ys;
    const float inv_dx = 1.0f / sc->dx, inv_dy = 1.0f / sc->dy, inv_dz = 1.0f / sc->dz;
    const int g = sl->gslab, nci = sl->nci, ncj = sl->ncj, nck = sl->nck;
    const float hx = sl->hx, hy = sl->hy, hz = sl->hz;
    const float inv_hx = g >= 0 ? 1.0f / hx : 0.0f, inv_hy = g >= 0 ? 1.0f / hy : 0.0f;
    const float zc0 = g >= 0 ? sl->zface[g] : 0.0f, zc1 = g >= 0 ? sl->zface[g + 1] : 0.0f;
    const float inv_hz = g >= 0 ? 1.0f / hz : 0.0f;

    uint32_t w[4];
    uint32_t nev = 0;

    /* launch, raytracer_kernels.cu:180-208 */
    draw(seed, pid, nev++, 0u, w);
    float x, y;
    if (sc->launch_log2k == -2)
    {
        /* Sobol launch: the reference's own (raytracer_kernels.cu:239-253, 331: scrambled Sobol32 dimensions 0 and
         * 1, tables raytracer_3d.cu:135-145, positions scaled as :195-196), as a function of the photon ID: point
         * (ID mod 2^32) of the sequence by the textbook recurrence -- dimension 0 the radical inverse (direction
         * numbers 2^(31-k)), dimension 1 v_0 = 2^31, v_k = v_(k-1) ^ (v_(k-1) >> 1) -- XORed with a digital shift
         * drawn per seed and per block of 2^32 IDs (counter words no photon's own stream uses). */
        uint32_t s[2];
        oracle_sobol_words(seed, pid, 1, s);
        x = xs * u01(s[0]);
        y = ys * u01(s[1]);
    }
    else if (sc->launch_log2k < 0)
    {
        x = xs * u01(w[0]);
        y = ys * u01(w[1]);
    }
    else
    {
        /* Stratified launch (stands in for the reference's scrambled-Sobol positions,
         * raytracer_kernels.cu:239-253, 331: low-discrepancy, here keyed on the photon ID).
         * Consecutive groups of K = 2^log2k IDs go to consecutive columns in row-major (or tile) order; one
         * pass over all columns takes ncol*K IDs and is rotated by a random offset drawn per
         * pass, so that any prefix of the ID sequence is unbiased; within the column the position
         * is uniform. */
        const uint64_t ncol = (uint64_t)itot * jtot;
        const uint64_t qt = pid >> sc->launch_log2k;
        const uint64_t pass = qt / ncol, q = qt % ncol;
        uint32_t wp[4];
        draw(seed, pass, 0xFFFFFFFFu, 1u, wp);
        const uint64_t off = ((uint64_t)wp[0] * ncol) >> 32;
        uint64_t c = q + off;
        if (c >= ncol) c -= ncol;
        const int T = sc->launch_tile_log2;
        const uint64_t tile = c >> (2 * T), within = c & ((1ull << (2 * T)) - 1ull), ntx = (uint64_t)(itot >> T);
        const uint32_t ic = (uint32_t)(((tile % ntx) << T) | (within & ((1ull << T) - 1ull)));
        const uint32_t jc = (uint32_t)(((tile / ntx) << T) | (within >> T));
        x = fmaf(u01(w[0]), sc->dx, (float)ic * sc->dx);
        y = fmaf(u01(w[1]), sc->dy, (float)jc * sc->dy);
    }
    float tau = free_path_tau(u01(w[2]));
    if (!(tau > 0.0f)) tau = 0.0f;
    float z = sl->zface[sl->nslab];
    float ux = sl->sun[0], uy = sl->sun[1], uz = sl->sun[2];
    float iux = dir_reciprocal(ux), iuy = dir_reciprocal(uy), iuz = dir_reciprocal(uz);
    int slab = sl->nslab - 1;
    int ci = 0, cj = 0, ck = 0;
    if (slab == g) { ci = cell_of(x, inv_hx, nci); cj = cell_of(y, inv_hy, ncj); ck = nck - 1; }
    int diffuse = 0;
    {
        const int i = cell_of(x, inv_dx, itot), j = cell_of(y, inv_dy, jtot);
        bump(&t->toa_down[(size_t)i + (size_t)j * itot]);
    }

    for (uint32_t step = 0;; ++step)
    {
        if (step >= RT_MAX_STEPS || !(z == z) || !(x == x) || !(y == y))
        {
            ++ev[EV_ABORT];
            return;
        }
        ++ev[EV_STEP];

        /* ---- traverse one majorant region ---- */
        const int gridded = slab == g;
        float M, invM, tx, ty, tz, fz;
        /* the region: the supervoxel's own box around (ci, cj, ck) */
        uint32_t eb = 0u;
        size_t e = 0;
        if (gridded)
        {
            e = ((size_t)ck * ncj + cj) * nci + ci;
            eb = sl->box[e];
        }
        const int bx0 = ci - (int)(eb & 31u), by0 = cj - (int)((eb >> 10) & 31u), bz0 = ck - (int)((eb >> 20) & 31u);
        const int bx1 = ci + 1 + (int)((eb >> 5) & 31u), by1 = cj + 1 + (int)((eb >> 15) & 31u), bz1 = ck + 1 + (int)(eb >> 25);
        const float xlo = (float)bx0 * hx, xhi = (bx1 == nci) ? xs : (float)bx1 * hx;
        const float ylo = (float)by0 * hy, yhi = (by1 == ncj) ? ys : (float)by1 * hy;
        const float fx = iux > 0.0f ? xhi : xlo, fy = iuy > 0.0f ? yhi : ylo;
        if (gridded)
        {
            M = sl->maj[2 * e]; invM = sl->maj[2 * e + 1];
            const float zlo = fmaf((float)bz0, hz, zc0), zhi = (bz1 == nck) ? zc1 : fmaf((float)bz1, hz, zc0);
            fz = iuz > 0.0f ? zhi : zlo;
            tx = face_distance(fx, x, iux); ty = face_distance(fy, y, iuy); tz = face_distance(fz, z, iuz);
        }
        else
        {
            M = sl->kmaj[slab]; invM = sl->inv_kmaj[slab];
            fz = iuz > 0.0f ? sl->zface[slab + 1] : sl->zface[slab];
            tx = INFINITY; ty = INFINITY; tz = face_distance(fz, z, iuz);
        }
        float t_exit = tx; int axis = gridded ? 0 : 2;           /* a clear-air slab has only its z faces */
        if (ty < t_exit) { t_exit = ty; axis = 1; }
        if (tz < t_exit) { t_exit = tz; axis = 2; }
        const float dn = tau * invM;                             /* raytracer_kernels.cu:367; invM is finite */
        const int hit = t_exit <= dn;
        const float tt = hit ? t_exit : dn;
        x = fmaf(ux, tt, x); y = fmaf(uy, tt, y); z = fmaf(uz, tt, z);
        if (!gridded) { x = wrap_periodic(x, xs, inv_xs); y = wrap_periodic(y, ys, inv_ys); }
        if (eb != 0u)
        {
            /* inside a box the supervoxel indices follow the position, within the box and never back along an axis:
             * the motion is monotone on each axis, a coordinate that sits on a face may round to either side, and
             * the box of the supervoxel behind a face reaches back over it */
            const int lx0 = iux > 0.0f ? ci : bx0, lx1 = iux > 0.0f ? bx1 - 1 : ci;
            const int ly0 = iuy > 0.0f ? cj : by0, ly1 = iuy > 0.0f ? by1 - 1 : cj;
            const int lz0 = iuz > 0.0f ? ck : bz0, lz1 = iuz > 0.0f ? bz1 - 1 : ck;
            ci = cell_of(x, inv_hx, nci); ci = ci < lx0 ? lx0 : ci; ci = ci > lx1 ? lx1 : ci;
            cj = cell_of(y, inv_hy, ncj); cj = cj < ly0 ? ly0 : cj; cj = cj > ly1 ? ly1 : cj;
            ck = (int)((z - zc0) * inv_hz); ck = ck < lz0 ? lz0 : ck; ck = ck > lz1 ? lz1 : ck;
        }

        if (hit)
        {
            ++ev[EV_TRANS];
            tau = fmaf(-tt, M, tau);
            if (!(tau > 0.0f)) tau = 0.0f;
            int leave = 0;                       /* +1: leaves the slab upwards, -1: downwards */
            if (axis == 0)
            {
                x = fx; ci = iux > 0.0f ? bx1 : bx0 - 1;
                if (ci == nci) { ci = 0; x = 0.0f; }
                if (ci < 0) { ci = nci - 1; x = xs; }
            }
            else if (axis == 1)
            {
                y = fy; cj = iuy > 0.0f ? by1 : by0 - 1;
                if (cj == ncj) { cj = 0; y = 0.0f; }
                if (cj < 0) { cj = ncj - 1; y = ys; }
            }
            else
            {
                z = fz;
                const int up = iuz > 0.0f;
                if (gridded)
                {
                    const int nk = up ? bz1 : bz0 - 1;
                    if (nk >= 0 && nk < nck) ck = nk; else leave = up ? 1 : -1;
                }
                else
                    leave = up ? 1 : -1;
            }
            if (leave != 0)
            {
                const int i = cell_of(x, inv_dx, itot), j = cell_of(y, inv_dy, jtot);
                const size_t ij = (size_t)i + (size_t)j * itot;
                if (leave > 0 && slab == sl->nslab - 1)
                {
                    /* top of atmosphere, raytracer_kernels.cu:487-499 */
                    ++ev[EV_TOA];
                    bump(&t->toa_up[ij]);
                    return;
                }
                else if (leave < 0 && slab == 0)
                {
                    /* surface, raytracer_kernels.cu:455-486 */
                    ++ev[EV_SFC];
                    bump(diffuse ? &t->sfc_dif[ij] : &t->sfc_dir[ij]);
                    draw(seed, pid, nev++, 0u, w);
                    if (u01(w[0]) <= sc->albedo)
                    {
                        bump(&t->sfc_up[ij]);
                        const float mu = sqrtf(u01(w[1]));
                        if (sc->planar)
                        {
                            ux = u01(w[2]) >= 0.5f ? mu : -mu;
                            uy = 0.0f;
                            uz = sqrtf(fmaf(-mu, mu, 1.0f));
                        }
                        else
                        {
                        float sa, ca;
                        oracle_sincos2pi(u01(w[2]), &sa, &ca);
                        ux = mu * sa;
                        uy = mu * ca;
                        uz = sqrtf(fmaf(-mu, mu, 1.0f) + RT_EPS);
                        }
                        iux = dir_reciprocal(ux); iuy = dir_reciprocal(uy); iuz = dir_reciprocal(uz);
                        diffuse = 1;
                    }
                    else
                        return;
                }
                else
                {
                    /* next slab, raytracer_kernels.cu:376-415 */
                    slab += leave;
                    if (slab == g)
                    {
                        ci = cell_of(x, inv_hx, nci); cj = cell_of(y, inv_hy, ncj);
                        ck = leave > 0 ? 0 : nck - 1;
                    }
                }
            }
            continue;
        }

        /* ---- collision test, raytracer_kernels.cu:507-566 ---- */
        ++ev[EV_COLL];
        const int i = cell_of(x, inv_dx, itot), j = cell_of(y, inv_dy, jtot), k = cell_of(z, inv_dz, ktot);
        const size_t c = (size_t)i + (size_t)j * itot + (size_t)k * itot * jtot;
        const float kt = k_gas[c] + k_cloud[c];
        draw(seed, pid, nev++, 0u, w);
        const float r = u01(w[0]);
        const float p_real = kt * invM;
        if (r >= p_real)
        {
            /* null collision: new optical depth, same direction */
            tau = free_path_tau(u01(w[1]));
            if (!(tau > 0.0f)) tau = 0.0f;
            continue;
        }
        const float s_scat = ssa[c] * p_real;
        if (r <= s_scat)
        {
            ++ev[EV_REAL]; ++ev[EV_SCAT];
            tau = free_path_tau(u01(w[1]));
            if (!(tau > 0.0f)) tau = 0.0f;
            /* gas or cloud: r is uniform on [0, s_scat] here, so r/s_scat < k_cloud/k_tot decides
             * (raytracer_kernels.cu:520 draws a fresh number for it) */
            const int from_cloud = r * kt < s_scat * k_cloud[c];
            if (from_cloud) ++ev[EV_CSCAT];
            const float mu = from_cloud ? henyey_mu(asy[c], u01(w[2])) : rayleigh_mu(u01(w[2]));
            if (sc->planar)
            {
                /* in-plane rotation by +-theta: sin/cos of (+-acos(mu) + atan2(ux, uz)) of raytracer.cpp:360-364
                 * expanded with the angle-sum formulas, so no inverse trigonometry is needed */
                const float sn2 = sqrtf(fmaxf(fmaf(-mu, mu, 1.0f), 0.0f));
                const float ssn = u01(w[3]) >= 0.5f ? sn2 : -sn2;
                const float nx = fmaf(mu, ux, ssn * uz);
                const float nz = fmaf(mu, uz, -(ssn * ux));
                ux = nx; uy = 0.0f; uz = nz;
                iux = dir_reciprocal(ux); iuy = dir_reciprocal(uy); iuz = dir_reciprocal(uz);
                diffuse = 1;
                continue;
            }
            /* raytracer_kernels.cu:523. In float the inverse CDFs can return |mu| a few ulp
             * above 1, which makes the radicand negative (NaN direction in the reference);
             * clamped at 0 here (SURVEY.md §7 "float edge cases"). */
            const float sn = sqrtf(fmaxf(fmaf(-mu, mu, 1.0f) + RT_EPS, 0.0f));

            /* frame, raytracer_kernels.cu:525-541 */
            float ax = 0.0f, ay = 0.0f, az = 0.0f;
            if (fabsf(ux) < fabsf(uy))
            {
                if (fabsf(ux) < fabsf(uz)) ax = 1.0f; else az = 1.0f;
            }
            else
            {
                if (fabsf(uy) < fabsf(uz)) ay = 1.0f; else az = 1.0f;
            }
            /* dot(t1, dir): the reference's third term is t1.z*t1.z (:73) */
            const float proj = sc->fixed_dot ? fmaf(az, uz, fmaf(ay, uy, ax * ux))
                                             : fmaf(az, az, fmaf(ay, uy, ax * ux));
            float ox = fmaf(-proj, ux, ax), oy = fmaf(-proj, uy, ay), oz = fmaf(-proj, uz, az);
            const float inv_len = 1.0f / sqrtf(fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
            ox *= inv_len; oy *= inv_len; oz *= inv_len;
            const float bx = fmaf(uy, oz, -(uz * oy));
            const float by = fmaf(uz, ox, -(ux * oz));
            const float bz = fmaf(ux, oy, -(uy * ox));

            float sp, cp;
            oracle_sincos2pi(u01(w[3]), &sp, &cp);
            /* raytracer_kernels.cu:545-546 */
            const float nx = fmaf(mu, ux, sn * fmaf(sp, ox, cp * bx));
            const float ny = fmaf(mu, uy, sn * fmaf(sp, oy, cp * by));
            const float nz = fmaf(mu, uz, sn * fmaf(sp, oz, cp * bz));
            ux = nx; uy = ny; uz = nz;
            iux = dir_reciprocal(ux); iuy = dir_reciprocal(uy); iuz = dir_reciprocal(uz);
            diffuse = 1;
        }
        else
        {
            ++ev[EV_REAL]; ++ev[EV_ABS];
            bump(diffuse ? &t->atm_dif[c] : &t->atm_dir[c]);
            return;
        }
    }
}

/* Traces photons [pid_begin, pid_begin + count). Tallies are ADDED to (never cleared).
 * events (may be NULL) receives EV_N counters. Returns 0 on success. */
int photon_oracle_trace(const oracle_scene_t* sc,
                        const float* k_gas, const float* k_cloud, const float* ssa, const float* asy,
                        uint64_t seed, uint64_t pid_begin, uint64_t count, int nthreads,
                        uint64_t* toa_down, uint64_t* toa_up,
                        uint64_t* sfc_dir, uint64_t* sfc_dif, uint64_t* sfc_up,
                        uint64_t* atm_dir, uint64_t* atm_dif,
                        uint64_t* events)
{
    if (!sc || sc->itot <= 0 || sc->jtot <= 0 || sc->ktot <= 0) return 1;
    oracle_slabs_t sl;
    oracle_build_slabs(sc, k_gas, k_cloud, asy, &sl);
    const tallies_t t = {toa_down, toa_up, sfc_dir, sfc_dif, sfc_up, atm_dir, atm_dif};
    uint64_t ev_total[EV_N];
    memset(ev_total, 0, sizeof(ev_total));

#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif

#pragma omp parallel
    {
        uint64_t ev[EV_N];
        memset(ev, 0, sizeof(ev));
#pragma omp for schedule(dynamic, 4096)
        for (uint64_t n = 0; n < count; ++n)
            trace_one(sc, &sl, k_gas, k_cloud, ssa, asy, seed, pid_begin + n, &t, ev);
#pragma omp critical
        for (int e = 0; e < EV_N; ++e) ev_total[e] += ev[e];
    }
    if (events) memcpy(events, ev_total, sizeof(ev_total));
    oracle_free_slabs(&sl);
    return 0;
}
