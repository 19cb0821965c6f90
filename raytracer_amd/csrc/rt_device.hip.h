// rt_device.hip.h — device-side building blocks of the MI355X photon tracer (gfx950).
//
// Everything here is written for 64-wide wavefronts and compiled with -ffp-contract=off: each
// fused multiply-add is an explicit __builtin_fmaf and every other float operation is a single
// IEEE-754 correctly rounded +, -, *, / or sqrt, so a photon's trajectory is a pure function of
// (seed, photon id, scene) and can be checked bit for bit against the CPU oracle.
//
// Reference code each piece stands in for (relative to /root/reference/src/):
//   philox4x32_10 ..... Random_number_generator (cuRAND XORWOW), raytracer_kernels.cu:211-236, and
//                       Quasi_random_number_generator_2d, :239-253 — replaced by a counter-based
//                       generator keyed on the photon ID (north_star)
//   rt_log ............ log() inside sample_tau, raytracer_kernels.cu:165-170
//   rt_sincos2pi ...... sin/cos of 2*pi*rng(), raytracer_kernels.cu:468-471, 543-546
//   rt_inv_cbrt ....... pow(., 1/3) inside rayleigh, raytracer_kernels.cu:145-152 (as its reciprocal)
//   rt_sqrt, rt_rcp ... sqrt() and 1/x of the scattering arithmetic, raytracer_kernels.cu:520-548, bit-identical
//                       to the IEEE operations on the ranges the tracer feeds them
//   rayleigh_mu / henyey_mu  raytracer_kernels.cu:145-162
//   wrap_periodic ..... fmod + add, raytracer_kernels.cu:440-448
//   cell_of ........... float_to_int, raytracer_kernels.cu:173-178
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_EPS 1.1920928955078125e-07f   // FLT_EPSILON (Float_epsilon, raytracer_kernels.cu:24)
#define RT_TWO_PI 6.2831854820251465f    // Float(2.*M_PI)
#define RT_MAX_STEPS (1u << 24)          // per-photon iteration guard (every wave must drain); the pool kernel packs a
                                         // photon's event index into 24 bits, and an event takes a step

namespace rt
{
enum : uint32_t { TAG_LAUNCH = 0, TAG_STEP = 1, TAG_SCATTER = 2 };

struct u32x4 { uint32_t x, y, z, w; };

// Philox4x32-10 (Salmon et al., SC'11). One v_mad_u64_u32 yields both halves of each product.
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round)
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input XOR in one instruction (v_bitop3_b32, truth table 0x96): half the XORs of a round
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// The same block with the ten round keys handed in, {k0 + r * 0x9E3779B9, k1 + r * 0xBB67AE85} for r = 0 .. 9 (the host
// computes them once per launch: they depend on the seed alone).
__device__ __forceinline__ u32x4 philox4x32_10_keys(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, const uint32_t (&rk)[20])
{
#pragma unroll
    for (int round = 0; round < 10; ++round)
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, rk[2 * round], 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, rk[2 * round + 1], 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
    }
    return u32x4{c0, c1, c2, c3};
}

// top 24 bits -> [0,1) on the 2^-24 lattice (exact)
__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }

// Point n of the two-dimensional Sobol sequence -- dimensions 0 and 1 of the Joe-Kuo direction numbers the
// reference hands to cuRAND (raytracer_3d.cu:135-145), a (0,2)-sequence in base 2 -- as a function of the index,
// no state and no table (the reference walks two curandStateScrambledSobol32 generators from offset
// tid * photons_per_thread, raytracer_kernels.cu:239-253, 331). Dimension 0 is the bit reversal of n (direction
// numbers 2^(31-k): v_bfrev_b32). Dimension 1 has the direction numbers v_k = v_(k-1) ^ (v_(k-1) >> 1), v_0 = 2^31,
// i.e. the Pascal matrix mod 2: output bit 31-j is the XOR of the index bits k whose binary digits contain those of
// j (Lucas), which on the bit-reversed index is a sum over subsets -- five shift-and-XOR steps. sx, sy: the digital
// shift (cuRAND's "scramble constants" are XORed into the state the same way); a digital shift keeps every
// elementary-interval property of the sequence.
__device__ __forceinline__ void sobol_2d(uint32_t n, uint32_t sx, uint32_t sy, uint32_t& x0, uint32_t& x1)
{
    const uint32_t r = __builtin_bitreverse32(n);
    uint32_t y = r;
    y ^= (y << 1) & 0xAAAAAAAAu;
    y ^= (y << 2) & 0xCCCCCCCCu;
    y ^= (y << 4) & 0xF0F0F0F0u;
    y ^= (y << 8) & 0xFF00FF00u;
    y ^= (y << 16) & 0xFFFF0000u;
    x0 = r ^ sx;
    x1 = y ^ sy;
}

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// natural log of a positive normal float: mantissa to [sqrt(1/2), sqrt(2)), degree-9 polynomial
__device__ __forceinline__ float rt_log(float x)
{
    uint32_t ix = __float_as_uint(x);
    ix += 0x3f800000u - 0x3f3504f3u;
    const int e = (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float f = __uint_as_float(ix) - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = fma_(p, f, -1.1514610310e-1f);
    p = fma_(p, f, 1.1676998740e-1f);
    p = fma_(p, f, -1.2420140846e-1f);
    p = fma_(p, f, 1.4249322787e-1f);
    p = fma_(p, f, -1.6668057665e-1f);
    p = fma_(p, f, 2.0000714765e-1f);
    p = fma_(p, f, -2.4999993993e-1f);
    p = fma_(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    const float fe = (float)e;
    y = fma_(fe, -2.12194440e-4f, y);
    y = fma_(-0.5f, z, y);
    return fma_(fe, 0.693359375f, f + y);
}

// sin(2*pi*u), cos(2*pi*u) for u in [0,1) on the 2^-24 lattice
__device__ __forceinline__ void rt_sincos2pi(float u, float& s_out, float& c_out)
{
    const int q = (int)fma_(u, 4.0f, 0.5f);
    const float r = u - 0.25f * (float)q;
    const float x = r * RT_TWO_PI;
    const float x2 = x * x;
    float ps = -1.9515295891e-4f;
    ps = fma_(ps, x2, 8.3321608736e-3f);
    ps = fma_(ps, x2, -1.6666654611e-1f);
    const float s = fma_(x * x2, ps, x);
    float pc = 2.443315711809948e-5f;
    pc = fma_(pc, x2, -1.388731625493765e-3f);
    pc = fma_(pc, x2, 4.166664568298827e-2f);
    const float c = fma_(x2 * x2, pc, fma_(-0.5f, x2, 1.0f));
    const bool swap = q & 1;
    const float a = swap ? c : s;        // |sin| source
    const float b = swap ? s : c;        // |cos| source
    // q&3: 0 -> (s, c); 1 -> (c, -s); 2 -> (-s, -c); 3 -> (-c, s)
    s_out = (q & 2) ? -a : a;
    c_out = (((q + 1) & 2) != 0) ? -b : b;
}

// sqrtf(x), bit for bit, for x == 0 or 2^-96 <= x <= 2^96: the hardware estimate (1 ulp) moved to the correctly
// rounded neighbour by the sign of two exact residuals. This is the core of the compiler's own IEEE sequence
// without its range scaling and special-value selects (16 instructions against 9). Every argument in the
// tracer is 0 or lies in [1e-14, 10]; tests/test_gpu_parity.py checks all 2^32 bit patterns of the domain.
__device__ __forceinline__ float rt_sqrt(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = fma_(-s_dn, s, x), r_up = fma_(-s_up, s, x);
    s = r_dn <= 0.0f ? s_dn : s;
    s = r_up > 0.0f ? s_up : s;
    return s;
}

// 1.0f / x, bit for bit, for 2^-100 <= |x| <= 2^100: the hardware estimate (1 ulp) and two Newton steps in
// fused arithmetic (the compiler's IEEE division spends 11 instructions on operand scaling and fix-ups that a
// reciprocal inside this range does not need). Checked over every bit pattern of the domain like rt_sqrt.
__device__ __forceinline__ float rt_rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    r = fma_(fma_(-x, r, 1.0f), r, r);
    r = fma_(fma_(-x, r, 1.0f), r, r);
    return r;
}

// 1/M of a majorant region as the traverse steps take it from M: the IEEE quotient for M >= 2^-100 (rt_rcp is exact
// up to 2^100; the pre-pass refuses larger extinctions), FLT_MAX below -- finite, so that tau * (1/M) is a number for
// tau == 0 too (a region without any extinction: tau == 0 then stops the photon for a collision test that is null,
// any tau > 0 carries it through)
__device__ __forceinline__ float region_reciprocal(float m)
{
    return m >= 0x1p-100f ? rt_rcp(m) : 3.4028234663852886e38f;
}

// x^(-1/3) of a positive normal float: bit-shift seed (4 % off), three division-free Newton steps
// y <- y (4 - x y^3) / 3; 2 ulp
__device__ __forceinline__ float rt_inv_cbrt(float x)
{
    float y = __uint_as_float(0x54a2fa8cu - __float_as_uint(x) / 3u);
    const float x3 = x * 0.3333333432674408f;
#pragma unroll
    for (int it = 0; it < 3; ++it)
    {
        const float y3 = (y * y) * y;
        y = y * fma_(-x3, y3, 1.3333333730697632f);
    }
    return y;
}

// mu = u - 1/u with u = cbrt(v), v = -q + sqrt(q^2 + 1), q = 4 r - 2 (raytracer_kernels.cu:145-152), written
// with y = v^(-1/3): u = v y^2 and 1/u = y, so mu = y (v y - 1) and no division is left
__device__ __forceinline__ float rayleigh_mu(float r)
{
    const float q = fma_(4.0f, r, -2.0f);
    const float d = fma_(q, q, 1.0f);
    const float v = -q + rt_sqrt(d);
    const float y = rt_inv_cbrt(v);
    return y * fma_(v, y, -1.0f);
}

// |g| < 1e-3 takes the isotropic limit: the reference formula divides by g (NaN at g == 0)
__device__ __forceinline__ float henyey_mu(float g, float r)
{
    if (__builtin_fabsf(g) < 1.0e-3f)
        return fma_(2.0f, r, -1.0f);
    const float one_m_g2 = 1.0f - g * g;
    const float a = one_m_g2 * one_m_g2;
    const float h = fma_(2.0f * r, g, 1.0f - g);
    const float h2 = h * h;
    // one reciprocal serves both quotients: 1/b, and 1/(2g) = h^2 / b; 7e-18 < |b| < 8
    const float rb = rt_rcp((2.0f * g) * h2);
    const float c = -g * 0.5f - h2 * rb;
    return -(a * rb) - c;
}

__device__ __forceinline__ float free_path_tau(float r) { return -rt_log((1.0f - r) + RT_EPS); }

__device__ __forceinline__ float wrap_periodic(float x, float size, float inv_size)
{
    const float q = __builtin_floorf(x * inv_size);
    x = fma_(-q, size, x);
    x = (x < 0.0f) ? x + size : x;
    x = (x >= size) ? x - size : x;
    return x;
}

__device__ __forceinline__ int cell_of(float s, float inv_ds, int n)
{
    int c = (int)(s * inv_ds);
    c = c > n - 1 ? n - 1 : c;
    c = c < 0 ? 0 : c;
    return c;
}
} // namespace rt
