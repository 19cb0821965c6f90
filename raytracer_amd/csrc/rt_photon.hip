// rt_photon.hip — MI355X (gfx950) photon tracer: kernels and the C ABI of include/rt_photon.h.
//
// Hot path replaced: ray_tracer_kernel of the reference (raytracer_kernels.cu:307-569) and the
// staging code around its launch (raytracer_3d.cu:81-180, raytracer_3d_io.cu:60-77).
//
// Design:
//  * photon state never touches HBM (the reference keeps it in global memory,
//    raytracer_kernels.cu:360-473): the default kernel holds 128 photons per wave (48-byte records
//    in LDS, ID words and step count in registers) and compacts those waiting for the same stage
//    (traverse / event / new direction / outside-the-grid) into full waves; a slot whose photon ended
//    takes the next photon ID of the wave's chunk (regeneration, :477-498);
//  * Philox4x32-10 keyed on (seed, photon id, event index): tallies are integers, so the sum
//    is independent of the launch shape, of the split over calls and of the GPU count;
//  * optical properties as one 16-byte record per cell {k_tot, ssa, asy, k_cloud}: a collision
//    costs one memory sector however it ends;
//  * null-collision tracking with one majorant per horizontal slab (below / in / above the cloud
//    layer, the reference's scheme, :360-415) and a grid of local majorants over the cloud slab,
//    traversed face to face with the sampled optical depth carried along.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC (see __graft_entry__.py).

#include "rt_device.hip.h"
#include "../../include/rt_photon.h"

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

// Scratch of rt_trace_noalloc (the one entry point that allocates nothing, see include/rt_photon.h): the
// photon-ID cursors and the cold block of a pool launch, 2 KiB per GPU, held until the process ends. In host code
// this name is only a stand-in for the symbol: the address on each device comes from hipGetSymbolAddress
// (scratch_of_device below) and only that pointer is ever handed to a kernel or a memset.
static __device__ __align__(256) unsigned long long scratch[256];

namespace
{
thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define RT_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// device allocation of a test hook: freed on every way out of the function, error returns included
template <typename T>
struct HookBuffer
{
    T* p = nullptr;
    HookBuffer() = default;
    HookBuffer(const HookBuffer&) = delete;
    HookBuffer& operator=(const HookBuffer&) = delete;
    ~HookBuffer() { if (p) (void)hipFree(p); }
};

constexpr int kBlock = 256;
// device words behind rt_context::d_events: RT_N_EVENTS counters, RT_N_SCHED scheduler statistics and,
// on a cache line of its own, the photon-ID cursor of the pool kernel
constexpr int kCursorWord = 32;
constexpr int kCursorStride = 16;     // words between the cursors of two XCDs: a 128-byte line each
constexpr int kXcds = 8;
constexpr int kEventWords = kCursorWord + kXcds * kCursorStride;
// words of the static scratch of rt_trace_noalloc: the eight cursors (a 128-byte line each), then the cold block
constexpr int kScratchWords = 256;
constexpr int kScratchColdWord = kXcds * kCursorStride;
constexpr int kMaxDevices = 64;
// values of "launch_log2k" below zero: where a photon enters the domain top as a function of its ID
constexpr int kLaunchUniform = -1;    // two Philox draws (the first block of the photon's stream)
constexpr int kLaunchSobol = -2;      // point (ID mod 2^32) of the 2-D Sobol sequence, digitally shifted per 2^32-ID block

// ------------------------------------------------------------------------------------------
// Layout of the packed cell records in HBM (internal: tallies and every interface keep [k][j][i]). RT_BRICK = 0
// (default): plain [k][j][i]; 1: bricks of 2 x 2 columns of one level, one 64-byte sector each; 2: bricks of
// 2 x 2 x 2 cells, one 128-byte cache line each. The bricked layouts were built to cut the fabric reads (a line
// that covers a cube instead of a stick of eight cells along x) and measured on C4 (2^28 photons, stratified
// launch): fabric reads 1.271e9 / 1.240e9 / 1.216e9 per launch (-2 %, -4 %), kernel 43.9 / 44.0 / 44.05 ms --
// the reuse the L2 sees is between photons over time, not between the cells of a line, and the longer index
// arithmetic costs what the few saved reads give. Kept as a build option; nbi = bricks per row, nbij = bricks per
// level (pair); odd grid sizes are padded (the padding is never read).
#ifndef RT_BRICK
#define RT_BRICK 0
#endif
__host__ __device__ __forceinline__ uint32_t brick_index(uint32_t i, uint32_t j, uint32_t k, uint32_t nbi, uint32_t nbij)
{
#if RT_BRICK == 2
    const uint32_t b = (k >> 1) * nbij + (j >> 1) * nbi + (i >> 1);
    return (b << 3) | ((k & 1u) << 2) | ((j & 1u) << 1) | (i & 1u);
#elif RT_BRICK == 1
    const uint32_t b = k * nbij + (j >> 1) * nbi + (i >> 1);
    return (b << 2) | ((j & 1u) << 1) | (i & 1u);
#else
    return k * nbij + j * nbi + i;
#endif
}

// ------------------------------------------------------------------------------------------
// Kernel arguments of the lane-resident tracer (by value: they live in SGPRs).
struct TraceArgs
{
    const float4* cells;      // {k_tot = gas + cloud, ssa, asy, k_cloud} per cell [k][j][i]: one 16-byte
                              // record, so a collision costs one memory sector however it ends
    const float2* maj;        // ncoarse supervoxels [ck][cj][ci]: {M of the supervoxel's own box, its six extents in
                              // 5 bits each} (rt_boxes_kernel; the step takes 1/M from M: region_reciprocal, finite,
                              // FLT_MAX where M is 0); then one entry {M, 1/M} per slab
    unsigned long long* tallies;  // concatenated, rt_tally order
    unsigned long long* events;   // RT_N_EVENTS counters
    uint64_t seed;
    uint64_t pid_begin;
    uint64_t pid_end;
    int itot, jtot, ktot;
    uint32_t ncol;
    uint32_t ncell;
    uint32_t nbi, nbij;       // brick_index
    float xs, ys, inv_xs, inv_ys, inv_dx, inv_dy, inv_dz;
    float albedo;
    float sun_x, sun_y, sun_z, inv_sun_x, inv_sun_y, inv_sun_z;
    int nslab;
    float zface[4];
    int gslab;                // index of the slab refined into supervoxels, -1 if none
    int nci, ncj, nck;
    uint32_t ncoarse;
    float hx, hy, hz, inv_hx, inv_hy, inv_hz, zc0, zc1;
    int th_event, th_scatter, low_traverse;   // wave scheduling thresholds (lanes)
    // stratified launch (launch_log2k >= 0): 2^k consecutive photon IDs per column, columns in
    // row-major order, each pass over the columns rotated by a per-pass random offset
    int launch_log2k;
    float dx, dy;
    uint32_t div_itot_mul, div_itot_shift;    // c / launch_div == (c * mul) >> shift for c < 2^31
    uint32_t launch_div, launch_tile_log2;    // launch_div = itot, or the tiles per row of a tiled launch order
    uint32_t sob_x, sob_y;                    // Sobol launch (launch_log2k == kLaunchSobol): digital shift of the launch's 2^32-ID block
};

// Kernel arguments of the pool tracer: only what the traverse step -- the traverse stage's, and the one at the end
// of the event stage -- uses in every round. The wave's loop is short of scalar registers (every spilled one costs a v_readlane per use), so
// everything the rare stages need -- launch, slab stage, surface event, the abort path, statistics -- lives in
// the ColdBlock in device memory and is fetched with scalar loads where it is used.
struct PoolArgs
{
    const float4* cells;          // as TraceArgs
    const float2* maj;
    unsigned long long* tallies;
    const struct ColdBlock* cold;
    // Photon IDs are handed out in chunks from device-wide cursors, so the launch needs no static partition
    // (waves that start late or run slower simply take fewer chunks). One cursor per XCD (kCursorStride words
    // apart), each over its own eighth of the ID range: with the stratified launch an XCD then sweeps its own
    // band of columns and its private L2 keeps that band's cells; a wave whose XCD has run dry takes chunks from
    // the next XCD's range.
    unsigned long long* cursor;
    uint64_t seed;
    int itot, jtot, ktot;
    uint32_t ncol;
    uint32_t nbi, nbij;           // brick_index
    int gslab, nci, ncj, nck;
    int th_launch;                // waiting launches that trigger a launch round
    int tally_match;              // RT_SLAB_TALLY: rounds of matching equal tally words across the wave (0: runs of neighbours only)
    float xs, ys, inv_dx, inv_dy, inv_dz;
    float hx, hy, hz, inv_hx, inv_hy, inv_hz, zc0, zc1;
    // Philox's ten round keys of this launch's seed, {k0 + r * 0x9E3779B9, k1 + r * 0xBB67AE85}: the event stage reads
    // them with its other arguments (scalar loads through the kernarg pointer) instead of deriving them round by round
    // -- twenty scalar additions per event round on a CU whose one scalar unit is as loaded as its vector units
    uint32_t round_keys[20];
};

// Read-only while the tracer runs (the cursors sit elsewhere: a line that takes atomics must not also serve
// these loads). Written by rt_launch_block_kernel on the stream before each pool launch.
struct ColdBlock
{
    uint64_t pid_begin, total;
    uint64_t sub_size;            // IDs per XCD range (a multiple of chunk)
    unsigned long long* events;   // RT_N_EVENTS counters, then the scheduler statistics
    uint32_t chunk;
    int launch_log2k;
    float sun_x, sun_y, sun_z, inv_sun_x, inv_sun_y, inv_sun_z;
    float ztop, dx, dy, albedo;
    uint32_t div_itot_mul, div_itot_shift;
    uint32_t launch_div, launch_tile_log2;    // as TraceArgs
    float zface[4];
    float inv_xs, inv_ys;
    uint32_t ncoarse;
    int nslab;
    const uint32_t* sorted;       // launch sort: the launch's IDs (relative to pid_begin) in the order they are taken; or null
    uint32_t toa_done;            // launch sort: toa_down of this launch is already counted (rt_sort_toa_kernel)
    uint32_t sob_x, sob_y;        // Sobol launch: digital shift of this launch's 2^32-ID block
    // Sobol launch in box order (sobol_position_to_id below; sob_a = 0: off): the launch's IDs are taken box by box of
    // a 2^sob_a x 2^sob_b partition of the domain top instead of in ascending order or through a sorted list
    uint32_t sob_a, sob_b, sob_tile_log2, sob_klog2;
    uint32_t sob_b0;              // (low word of the launch's first ID) >> (sob_a + sob_b): its first net of the block
    uint32_t sob_tinv[16];        // unknown ID bits a .. a+b-1 = XOR of sob_tinv[r] over the set bits r of the residual
    uint32_t* atm32;              // rt_bind_atmos_wire: absorption counts go here as 32-bit words (same word index), or null
};

// The cold block through the constant address space: its contents do not change while the tracer runs (the kernel
// before it on the stream wrote them), so wave-uniform reads of it are scalar loads, not vector memory traffic.
typedef const __attribute__((address_space(4))) ColdBlock* ColdPtr;
__device__ __forceinline__ ColdPtr cold_of(const PoolArgs& a)
{
    return (ColdPtr)(uintptr_t)a.cold;
}

// The kernel's own argument block through the constant address space, made opaque once per stage: what a stage
// reads of it are scalar loads issued inside that stage, not values held in scalar registers across the whole
// loop (which is short of them: held there, they were spilled to VGPR lanes and read back with v_readlane, 16 per
// event round). Used by the event stage for ITS arguments (cells, tallies, grid dimensions and reciprocal
// spacings) and by the rare launch and slab stages for all of theirs; the traverse step keeps
// its arguments in registers (`a.`): loading those per round as well costs more in exposed scalar-load latency
// than the spills did (measured on C4: all stages opaque 45.5 ms, none 44.3 ms, this split 43.6 ms).
typedef const __attribute__((address_space(4))) PoolArgs* ArgPtr;
__device__ __forceinline__ ArgPtr args_now()
{
    ArgPtr p = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

__device__ __forceinline__ ColdPtr cold_of_k(ArgPtr k)
{
    return (ColdPtr)(uintptr_t)k->cold;
}

__device__ __forceinline__ void tally_add(unsigned long long* p, unsigned long long v = 1ull)
{
    // no-return 64-bit global atomic (write_photon_out, raytracer_kernels.cu:256-261)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Reciprocal of a direction component as the traversal stores it: finite (+-FLT_MAX for a component below
// 2^-100 in size, zero included), and its sign — not the component's — says which face of a region the photon
// heads for, so that (face - pos) * inv_u is never the 0 * inf or (-) * (+inf) of a zero component. Branch-free:
// rt_rcp is 1.0f / u exactly on the range that takes it.
__device__ __forceinline__ float dir_reciprocal(float u)
{
    const float r = rt::rt_rcp(u);
    return __builtin_fabsf(u) >= 0x1p-100f ? r : __builtin_copysignf(3.402823466e+38f, u);
}

// distance along the ray to the face the photon heads for; negative (past the face by rounding) -> 0
__device__ __forceinline__ float face_distance(float face, float pos, float inv_u)
{
    const float t = (face - pos) * inv_u;
    return t > 0.0f ? t : 0.0f;
}

// Rotation of pass p of the stratified launch: a Philox draw keyed on the pass index.
__device__ __forceinline__ uint32_t launch_pass_offset(uint64_t pass, uint32_t ncol, uint32_t key0, uint32_t key1)
{
    const rt::u32x4 w = rt::philox4x32_10((uint32_t)pass, (uint32_t)(pass >> 32), 0xFFFFFFFFu, 1u, key0, key1);
    return (uint32_t)(((uint64_t)w.x * ncol) >> 32);
}

// Column (i, j) of rotated column index c. tile_log2 = 0: row-major (div = itot). tile_log2 = T > 0: the columns are
// counted tile by tile -- tiles of 2^T x 2^T columns, row-major inside the tile, tiles row by row (div = tiles per
// row; the grid is a whole number of tiles) -- so that the photons in flight together enter one neighbourhood in
// BOTH directions and an XCD's L2 keeps the tile's cells while its 2^(2T+k) photons come through. The division by
// `div` goes through the host's magic number.
__device__ __forceinline__ void launch_column(uint32_t div_mul, uint32_t div_shift, uint32_t div, uint32_t tile_log2,
                                              uint32_t c, uint32_t& i, uint32_t& j)
{
    const uint32_t t = c >> (2u * tile_log2), w = c & ((1u << (2u * tile_log2)) - 1u);
    const uint32_t tj = (uint32_t)(((uint64_t)t * div_mul) >> div_shift), ti = t - tj * div;
    i = (ti << tile_log2) | (w & ((1u << tile_log2) - 1u));
    j = (tj << tile_log2) | (w >> tile_log2);
}
__device__ __forceinline__ void launch_column(const TraceArgs& a, uint32_t c, uint32_t& i, uint32_t& j)
{
    launch_column(a.div_itot_mul, a.div_itot_shift, a.launch_div, a.launch_tile_log2, c, i, j);
}

// Stratified column of one photon ID: 64-bit division per photon.
__device__ __forceinline__ uint32_t launch_column_index_exact(int log2k, uint32_t ncol, uint64_t pid, uint32_t key0, uint32_t key1)
{
    const uint64_t qt = pid >> log2k;
    uint64_t pass;
    uint32_t q;
    if ((ncol & (ncol - 1u)) == 0u)        // (wave-uniform) a power-of-two number of columns: no 64-bit division
    {
        pass = qt >> (31 - __builtin_clz(ncol));
        q = (uint32_t)qt & (ncol - 1u);
    }
    else
    {
        pass = qt / ncol;
        q = (uint32_t)(qt - pass * ncol);
    }
    uint32_t c = q + launch_pass_offset(pass, ncol, key0, key1);      // < 2 * ncol < 2^32
    if (c >= ncol) c -= ncol;
    return c;
}
__device__ __forceinline__ uint32_t launch_column_index_exact(const TraceArgs& a, uint64_t pid, uint32_t key0, uint32_t key1)
{
    return launch_column_index_exact(a.launch_log2k, a.ncol, pid, key0, key1);
}

// Sobol launch in box order: which photon ID a launch takes at position p (< 2^31) of its ID range. Any 2^m
// consecutive IDs that start at a multiple of 2^m (m = a + b) are a (0,m,2)-net: exactly one of them falls into each
// box of the 2^a x 2^b partition of the domain top. So the ID in box (i, j) of net B is the solution of a linear
// system over GF(2) -- the low a bits of the ID are the reversed bits of i (dimension 0 is the radical inverse), and
// the next b bits follow from j through the inverse of a b x b block of the Pascal matrix, which the host hands over
// as sob_tinv (build_sobol_order). The launch walks the boxes tile by tile (tiles of 2^T x 2^T boxes, as the tiled
// stratified order counts its columns) and takes 2^k nets per box at a time (positions p = ((sweep * boxes + box)
// << k) + r, net = (sweep << k) + r): the photons in flight together share a neighbourhood, and runs of 2^k
// consecutive lanes enter one column -- the locality of the stratified order without its restriction to whole
// passes over a power-of-two number of columns, and without the launch sort's passes. Scheduling only: every ID of
// the range is taken exactly once whatever the grid's shape (the boxes need not be the columns).
__device__ __forceinline__ uint32_t sobol_position_to_id(ColdPtr cb, uint32_t p)
{
    const uint32_t a = cb->sob_a, b = cb->sob_b, T = cb->sob_tile_log2, k = cb->sob_klog2, m = a + b;
    const uint32_t q = p >> k, sweep = q >> m, c = q & ((1u << m) - 1u);
    const uint32_t net = cb->sob_b0 + (sweep << k) + (p & ((1u << k) - 1u));
    const uint32_t t = c >> (2u * T), w = c & ((1u << (2u * T)) - 1u);
    const uint32_t tj = t >> (a - T), ti = t & ((1u << (a - T)) - 1u);
    const uint32_t i = (ti << T) | (w & ((1u << T) - 1u)), j = (tj << T) | (w >> T);
    const uint32_t low = __builtin_bitreverse32(i ^ (cb->sob_x >> (32u - a))) >> (32u - a);
    const uint32_t known = (net << m) | low;
    uint32_t x0, x1;
    rt::sobol_2d(known, 0u, cb->sob_y, x0, x1);
    const uint32_t rhs = j ^ (x1 >> (32u - b));
    uint32_t u = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        u ^= (uint32_t)(-(int32_t)((rhs >> r) & 1u)) & cb->sob_tinv[r];
    return known | (u << a);
}

enum : int { ST_LAUNCH = 0, ST_TRAVERSE = 1, ST_COLLIDE = 2, ST_SURFACE = 3, ST_SCATTER = 4, ST_DONE = 5, ST_REFLECT = 6 };

// ------------------------------------------------------------------------------------------
// The photon life cycle. One lane = one photon at a time, all photon state in VGPRs.
//
// Every lane is a small state machine; the wave runs ONE of three stages per round, chosen from
// ballots so that the branch is scalar and a stage only costs issue slots when enough lanes need it:
//   T  traverse  — cheap, most frequent: cross one majorant region or reach a collision site
//   E  event     — one Philox block per lane: launch (regeneration), collision classification,
//                  surface reflection
//   S  scatter   — phase-function sampling and rotation of the direction
// Lanes waiting for E or S sit masked out until th_event / th_scatter lanes wait (or too few lanes
// are traversing). Because every random number is keyed on (photon id, event index), the order in
// which lanes are served does not change any photon's trajectory.
template <bool FIXED_DOT, bool STATS>
__global__ void __launch_bounds__(kBlock)
rt_trace_kernel(const TraceArgs a)
{
    using namespace rt;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint64_t next_pid = a.pid_begin + (uint64_t)blockIdx.x * kBlock + threadIdx.x;

    const uint32_t key0 = (uint32_t)a.seed, key1 = (uint32_t)(a.seed >> 32);
    const float ztop = a.zface[a.nslab];

    // photon state
    float x = 0.f, y = 0.f, z = 0.f, ux = 0.f, uy = 0.f, uz = -1.f, iux = 0.f, iuy = 0.f, iuz = -1.f;
    float tau = 0.f, inv_m = 0.f;
    uint32_t pid_lo = 0, pid_hi = 0, nev = 0, nstep = 0;
    int slab = 0, ci = 0, cj = 0, ck = 0;
    bool diffuse = false;
    int state = ST_LAUNCH;
    // collision data handed from stage E to stage S
    uint32_t cell = 0;
    float sc_asy = 0.f, sc_umu = 0.f, sc_uphi = 0.f;
    bool sc_cloud = false;

    uint32_t ev_step = 0, ev_coll = 0, ev_real = 0, ev_scat = 0, ev_cscat = 0, ev_sfc = 0, ev_toa = 0,
             ev_abs = 0, ev_trans = 0;
    // wave-level scheduling statistics (STATS builds): rounds and served lanes per stage
    uint32_t rounds_t = 0, rounds_e = 0, rounds_s = 0, lanes_t = 0, lanes_e = 0, lanes_s = 0;

    for (;;)
    {
        if (state == ST_LAUNCH && next_pid >= a.pid_end)
            state = ST_DONE;
        const int nT = __popcll(__ballot(state == ST_TRAVERSE));
        const int nE = __popcll(__ballot(state == ST_LAUNCH || state == ST_COLLIDE || state == ST_SURFACE));
        const int nS = __popcll(__ballot(state == ST_SCATTER || state == ST_REFLECT));
        if (nT + nE + nS == 0)
            break;
        const bool starving = nT < a.low_traverse;
        const bool runS = nS >= a.th_scatter || (starving && nS > 0 && nS >= nE);
        const bool runE = !runS && (nE >= a.th_event || (starving && nE > 0));
        if (STATS)
        {
            if (runS) { ++rounds_s; lanes_s += nS; }
            else if (runE) { ++rounds_e; lanes_e += nE; }
            else { ++rounds_t; lanes_t += nT; }
        }

        if (runS)
        {
            if (state == ST_SCATTER || state == ST_REFLECT)
            {
                float sp, cp;
                rt_sincos2pi(sc_uphi, sp, cp);          // azimuth of the new direction, both cases
                if (state == ST_SCATTER)
                {
                    // phase function + rotation, raytracer_kernels.cu:520-548
                    const float mu = sc_cloud ? henyey_mu(sc_asy, sc_umu) : rayleigh_mu(sc_umu);
                    // |mu| can come out a few ulp above 1 in float: clamp the radicand (the reference
                    // would produce a NaN direction here, raytracer_kernels.cu:523)
                    const float sn = __builtin_sqrtf(__builtin_fmaxf(fma_(-mu, mu, 1.0f) + RT_EPS, 0.0f));

                    float ax = 0.0f, ay = 0.0f, az = 0.0f;
                    const float fx = __builtin_fabsf(ux), fy = __builtin_fabsf(uy), fz = __builtin_fabsf(uz);
                    if (fx < fy)
                    {
                        if (fx < fz) ax = 1.0f; else az = 1.0f;
                    }
                    else
                    {
                        if (fy < fz) ay = 1.0f; else az = 1.0f;
                    }
                    // dot(t1, dir): the reference's third term is t1.z*t1.z (raytracer_kernels.cu:73)
                    const float proj = FIXED_DOT ? fma_(az, uz, fma_(ay, uy, ax * ux))
                                                 : fma_(az, az, fma_(ay, uy, ax * ux));
                    float ox = fma_(-proj, ux, ax), oy = fma_(-proj, uy, ay), oz = fma_(-proj, uz, az);
                    const float inv_len = 1.0f / __builtin_sqrtf(fma_(oz, oz, fma_(oy, oy, ox * ox)));
                    ox *= inv_len; oy *= inv_len; oz *= inv_len;
                    const float bx = fma_(uy, oz, -(uz * oy));
                    const float by = fma_(uz, ox, -(ux * oz));
                    const float bz = fma_(ux, oy, -(uy * ox));
                    const float nx = fma_(mu, ux, sn * fma_(sp, ox, cp * bx));
                    const float ny = fma_(mu, uy, sn * fma_(sp, oy, cp * by));
                    const float nz = fma_(mu, uz, sn * fma_(sp, oz, cp * bz));
                    ux = nx; uy = ny; uz = nz;
                }
                else
                {
                    // Lambertian reflection, raytracer_kernels.cu:467-473
                    const float mu = __builtin_sqrtf(sc_umu);
                    ux = mu * sp;
                    uy = mu * cp;
                    uz = __builtin_sqrtf(fma_(-mu, mu, 1.0f) + RT_EPS);
                }
                iux = dir_reciprocal(ux); iuy = dir_reciprocal(uy); iuz = dir_reciprocal(uz);
                diffuse = true;
                state = ST_TRAVERSE;
            }
        }
        else if (runE)
        {
            if (state == ST_LAUNCH || state == ST_COLLIDE || state == ST_SURFACE)
            {
                float4 cr = make_float4(0.f, 0.f, 0.f, 0.f);
                if (state == ST_LAUNCH)
                {
                    pid_lo = (uint32_t)next_pid;
                    pid_hi = (uint32_t)(next_pid >> 32);
                    next_pid += stride;
                    nev = 0;
                    nstep = 0;
                }
                else if (state == ST_COLLIDE)
                {
                    const int i = cell_of(x, a.inv_dx, a.itot), j = cell_of(y, a.inv_dy, a.jtot);
                    const int k = cell_of(z, a.inv_dz, a.ktot);
                    cell = (uint32_t)i + (uint32_t)j * (uint32_t)a.itot + (uint32_t)k * a.ncol;
                    cr = a.cells[brick_index((uint32_t)i, (uint32_t)j, (uint32_t)k, a.nbi, a.nbij)];
                }
                const float kt = cr.x;
                const u32x4 w = philox4x32_10(pid_lo, pid_hi, nev, 0u, key0, key1);
                ++nev;
                // the optical depth to the next collision: third word at launch, second otherwise
                float tau_new = free_path_tau(u01(state == ST_LAUNCH ? w.z : w.y));
                tau_new = tau_new > 0.0f ? tau_new : 0.0f;

                if (state == ST_LAUNCH)
                {
                    // launch at the top of the domain, raytracer_kernels.cu:180-208
                    if (a.launch_log2k == kLaunchSobol)
                    {
                        // point pid_lo of the Sobol sequence (raytracer_kernels.cu:195-196 scale the two 32-bit
                        // words by the domain size the same way)
                        uint32_t s0, s1;
                        sobol_2d(pid_lo, a.sob_x, a.sob_y, s0, s1);
                        x = a.xs * u01(s0);
                        y = a.ys * u01(s1);
                    }
                    else if (a.launch_log2k < 0)
                    {
                        x = a.xs * u01(w.x);
                        y = a.ys * u01(w.y);
                    }
                    else
                    {
                        uint32_t ic, jc;
                        launch_column(a, launch_column_index_exact(a, ((uint64_t)pid_hi << 32) | pid_lo, key0, key1), ic, jc);
                        x = fma_(u01(w.x), a.dx, (float)ic * a.dx);
                        y = fma_(u01(w.y), a.dy, (float)jc * a.dy);
                    }
                    tau = tau_new;
                    z = ztop;
                    ux = a.sun_x; uy = a.sun_y; uz = a.sun_z;
                    iux = a.inv_sun_x; iuy = a.inv_sun_y; iuz = a.inv_sun_z;
                    slab = a.nslab - 1;
                    ci = 0; cj = 0; ck = 0;
                    if (slab == a.gslab)
                    {
                        ci = cell_of(x, a.inv_hx, a.nci); cj = cell_of(y, a.inv_hy, a.ncj); ck = a.nck - 1;
                    }
                    diffuse = false;
                    const int i = cell_of(x, a.inv_dx, a.itot), j = cell_of(y, a.inv_dy, a.jtot);
                    tally_add(&a.tallies[(uint32_t)i + (uint32_t)j * (uint32_t)a.itot]);
                    state = ST_TRAVERSE;
                }
                else if (state == ST_COLLIDE)
                {
                    // collision classification, raytracer_kernels.cu:507-518
                    if (STATS) ++ev_coll;
                    const float r = u01(w.x);
                    const float p_real = kt * inv_m;
                    state = ST_TRAVERSE;
                    // null collision or scattering: the photon goes on with a fresh optical depth
                    tau = tau_new;
                    if (r < p_real)
                    {
                        if (STATS) ++ev_real;
                        const float s_scat = cr.y * p_real;
                        if (r <= s_scat)
                        {
                            // gas or cloud: r/s < k_cloud/k_tot (raytracer_kernels.cu:520)
                            sc_cloud = r * kt < s_scat * cr.w;
                            sc_asy = cr.z; sc_umu = u01(w.z); sc_uphi = u01(w.w);
                            if (STATS) { ++ev_scat; if (sc_cloud) ++ev_cscat; }
                            state = ST_SCATTER;
                        }
                        else
                        {
                            // absorption, raytracer_kernels.cu:551-566
                            if (STATS) ++ev_abs;
                            tally_add(&a.tallies[5u * a.ncol + (diffuse ? a.ncell : 0u) + cell]);
                            state = ST_LAUNCH;
                        }
                    }
                }
                else
                {
                    // surface: absorb or reflect, raytracer_kernels.cu:463-486 (cell = column here);
                    // the new direction is computed in stage S
                    if (u01(w.x) <= a.albedo)
                    {
                        tally_add(&a.tallies[4u * a.ncol + cell]);
                        sc_umu = u01(w.y); sc_uphi = u01(w.z);
                        state = ST_REFLECT;
                    }
                    else
                        state = ST_LAUNCH;
                }
            }
        }
        else
        {
            if (state == ST_TRAVERSE)
            {
                if (nstep >= RT_MAX_STEPS || !(z == z) || !(x == x) || !(y == y))
                {
                    tally_add(&a.events[RT_EV_ABORT]);
                    state = ST_LAUNCH;
                }
                else
                {
                    ++nstep;
                    if (STATS) ++ev_step;
                    // ---- cross one majorant region or stop at the collision site ----
                    const bool gridded = slab == a.gslab;
                    const uint32_t e = gridded ? (uint32_t)((ck * a.ncj + cj) * a.nci + ci) : a.ncoarse + (uint32_t)slab;
                    const float2 mm = a.maj[e];
                    // the region: the supervoxel's own box around (ci, cj, ck) (rt_boxes_kernel); a slab's entry holds 1/M
                    const uint32_t eb = gridded ? __float_as_uint(mm.y) : 0u;
                    const float inv_mm = gridded ? region_reciprocal(mm.x) : mm.y;
                    const int bx0 = ci - (int)(eb & 31u), by0 = cj - (int)((eb >> 10) & 31u), bz0 = ck - (int)((eb >> 20) & 31u);
                    const int bx1 = ci + 1 + (int)((eb >> 5) & 31u), by1 = cj + 1 + (int)((eb >> 15) & 31u), bz1 = ck + 1 + (int)(eb >> 25);
                    // the face ahead on each axis: index of the cell face, then its coordinate
                    const int cfx = iux > 0.0f ? bx1 : bx0, cfy = iuy > 0.0f ? by1 : by0;
                    const float fx = cfx == a.nci ? a.xs : (float)cfx * a.hx;
                    const float fy = cfy == a.ncj ? a.ys : (float)cfy * a.hy;
                    const bool up = iuz > 0.0f;
                    float fz;
                    if (gridded)
                    {
                        const int cfz = up ? bz1 : bz0;
                        fz = cfz == a.nck ? a.zc1 : fma_((float)cfz, a.hz, a.zc0);
                    }
                    else
                    {
                        const int fi = up ? slab + 1 : slab;
                        fz = fi == 0 ? a.zface[0] : (fi == 1 ? a.zface[1] : (fi == 2 ? a.zface[2] : a.zface[3]));
                    }
                    const float tx = gridded ? face_distance(fx, x, iux) : __builtin_inff();
                    const float ty = gridded ? face_distance(fy, y, iuy) : __builtin_inff();
                    const float tz = face_distance(fz, z, iuz);
                    float t_exit = tx; int axis = gridded ? 0 : 2;      // a clear-air slab has only its z faces
                    if (ty < t_exit) { t_exit = ty; axis = 1; }
                    if (tz < t_exit) { t_exit = tz; axis = 2; }
                    const float dn = tau * inv_mm;        // raytracer_kernels.cu:367; 1/M is finite
                    const bool hit = t_exit <= dn;
                    const float tt = hit ? t_exit : dn;
                    x = fma_(ux, tt, x); y = fma_(uy, tt, y); z = fma_(uz, tt, z);
                    if (!gridded) { x = wrap_periodic(x, a.xs, a.inv_xs); y = wrap_periodic(y, a.ys, a.inv_ys); }
                    if (eb != 0u)
                    {
                        // inside a box the supervoxel indices follow the position, within the box and never back along
                        // an axis (see RT_TRAVERSE_STEP)
                        ci = min(max(cell_of(x, a.inv_hx, a.nci), iux > 0.0f ? ci : bx0), iux > 0.0f ? bx1 - 1 : ci);
                        cj = min(max(cell_of(y, a.inv_hy, a.ncj), iuy > 0.0f ? cj : by0), iuy > 0.0f ? by1 - 1 : cj);
                        ck = min(max((int)((z - a.zc0) * a.inv_hz), up ? ck : bz0), up ? bz1 - 1 : ck);
                    }
                    inv_m = inv_mm;

                    if (!hit)
                        state = ST_COLLIDE;
                    else
                    {
                        if (STATS) ++ev_trans;
                        tau = fma_(-tt, mm.x, tau);
                        tau = tau > 0.0f ? tau : 0.0f;
                        int leave = 0;
                        if (axis == 0)
                        {
                            x = fx; ci = iux > 0.0f ? bx1 : bx0 - 1;
                            if (ci == a.nci) { ci = 0; x = 0.0f; }
                            if (ci < 0) { ci = a.nci - 1; x = a.xs; }
                        }
                        else if (axis == 1)
                        {
                            y = fy; cj = iuy > 0.0f ? by1 : by0 - 1;
                            if (cj == a.ncj) { cj = 0; y = 0.0f; }
                            if (cj < 0) { cj = a.ncj - 1; y = a.ys; }
                        }
                        else
                        {
                            z = fz;
                            if (gridded)
                            {
                                const int nk = up ? bz1 : bz0 - 1;
                                if (nk >= 0 && nk < a.nck) ck = nk; else leave = up ? 1 : -1;
                            }
                            else
                                leave = up ? 1 : -1;
                        }
                        if (leave != 0)
                        {
                            if (leave > 0 && slab == a.nslab - 1)
                            {
                                // top of the atmosphere, raytracer_kernels.cu:487-499
                                if (STATS) ++ev_toa;
                                const int i = cell_of(x, a.inv_dx, a.itot), j = cell_of(y, a.inv_dy, a.jtot);
                                tally_add(&a.tallies[a.ncol + (uint32_t)i + (uint32_t)j * (uint32_t)a.itot]);
                                state = ST_LAUNCH;
                            }
                            else if (leave < 0 && slab == 0)
                            {
                                // surface arrival, raytracer_kernels.cu:455-460; the albedo draw is stage E
                                if (STATS) ++ev_sfc;
                                const int i = cell_of(x, a.inv_dx, a.itot), j = cell_of(y, a.inv_dy, a.jtot);
                                cell = (uint32_t)i + (uint32_t)j * (uint32_t)a.itot;
                                tally_add(&a.tallies[(diffuse ? 3u : 2u) * a.ncol + cell]);
                                state = ST_SURFACE;
                            }
                            else
                            {
                                // next slab, raytracer_kernels.cu:376-415
                                slab += leave;
                                if (slab == a.gslab)
                                {
                                    ci = cell_of(x, a.inv_hx, a.nci); cj = cell_of(y, a.inv_hy, a.ncj);
                                    ck = leave > 0 ? 0 : a.nck - 1;
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    if (STATS)
    {
        // one atomic per counter per lane at kernel end (stats launches are not the timed ones)
        const uint32_t ev[9] = {ev_step, ev_coll, ev_real, ev_scat, ev_cscat, ev_sfc, ev_toa, ev_abs, ev_trans};
#pragma unroll
        for (int e = 0; e < 9; ++e)
            if (ev[e])
                __hip_atomic_fetch_add(&a.events[e], (unsigned long long)ev[e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        if ((threadIdx.x & 63) == 0)
        {
            const uint32_t sch[6] = {rounds_t, rounds_e, rounds_s, lanes_t, lanes_e, lanes_s};
#pragma unroll
            for (int e = 0; e < 6; ++e)
                __hip_atomic_fetch_add(&a.events[RT_N_EVENTS + e], (unsigned long long)sch[e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Pool variant of the tracer: every wave keeps NS (> 64) photons in LDS and, each round, gathers
// up to 64 of them that wait for the SAME stage into its lanes (ballot + rank compaction). The
// per-photon arithmetic is that of rt_trace_kernel above (its traverse stage split in two here: inside
// the supervoxel grid, and the slab stage for everything outside it); only where the photon state
// lives between rounds and the order of service differ, so results are bit-identical. With 2 photons per
// lane in the pool, a round almost always finds a full wave of work for one stage (the lane-
// resident kernel splits its 64 photons over three stages and runs each at ~1/3 to 1/2 occupancy).
// One wave per workgroup (no barriers anywhere), so LDS alone sets the residency.
//
// LDS per wave: NS records of three 16-byte groups (stored group-major) and a 64-entry selection list
// (bytes):
//   g0 {x, y, z, tau}          tau holds 1/M of the region while the photon waits for its collision event
//                              (the event stage draws the next tau before anything reads it)
//   g1 {ux, uy, uz, ci | cj << 16}
//   g2 {1/ux, 1/uy, 1/uz, ck | slab << 16 | diffuse << 24 | alt << 26 | age << 27}
//                              alt tells the two states of the event stage apart (surface event, not collision):
//                              set by the slab stage on arrival at the surface. slab is a signed byte: -1 (below the surface) and
//                              nslab (above the domain top) while a photon that has just left the atmosphere
//                              waits for the slab stage. age is the step guard (below)
// and, in three registers of the lane that owns the slot (slot h * 64 + lane; read with ds_bpermute by the lane
// that serves it, taken back by the owner at the end of the round), {pid_lo, pid_hi | nev << 8, state}:
// the selection of a round reads the states from registers, no memory. 6208 bytes per wave at NS = 128: 26 waves
// share a CU's 160 KB (19 with the full 64-byte record in LDS).
//
// Step guard. Every wave must drain, so a photon that never ends (a non-finite state, or a bug that makes it
// cross faces forever) has to be retired: the lane-resident kernel and the oracle count its steps up to
// RT_MAX_STEPS = 2^24. Here a count per photon would be one more word to carry between the owner and the serving
// lane every round, so the guard is kept in time instead: every kTickRounds-th round of the wave is a tick (wave-
// uniform), a photon served by a stepping stage in a tick round ages by one (5 bits of the record), and one that
// is still there at age 31 is abandoned (RT_EV_ABORT; rt_synchronize reports it) -- after at least 31 * 2^21 =
// 6.5 * 10^7 rounds of its wave. A photon is served at most once per round and takes at most two steps when it is,
// so this guard is always the LOOSER of the two: it must never retire a photon that the step limit would let
// finish. (It did once: with ticks every 2^17 rounds a conservative, optically very thick planar scene of the
// randomised sweep -- 39 000 scatterings per photon on average, the longest above a million -- lost 16 of 60 000
// photons that the oracle traced to their end, because a photon in a pool of 192 is served only every third round.)
// A photon of an ordinary run ends after tens of steps (a launch of 2^28 photons is about 10^4 rounds per wave).
// Non-finite positions are caught in the step itself, at once.
// (state >> 2 is the stage that serves it: 0 traverse, 1 event, 3 launch, 4 slab; class 2 was round 1's new-direction stage)
// PS_LAUNCH_TALLY: a free slot whose last photon was absorbed in the air; word 0 of its record holds the tally word
// that the next launch round adds to (see the event stage)
enum : uint32_t { PS_TRAVERSE = 0, PS_COLLIDE = 4, PS_LAUNCH = 12, PS_LAUNCH_TALLY = 13, PS_SLAB = 16, PS_SURFACE = 17, PS_DONE = 20 };
enum : int { PHYS_COMPAT = 0, PHYS_FIXED_DOT = 1, PHYS_PLANAR = 2 };
constexpr int kPoolRec = 12;
constexpr uint32_t kAgeOne = 1u << 27, kAgeMask = 0xF8000000u;
constexpr uint32_t kTickRounds = 1u << 21;
#ifndef RT_AUTO_LEVELS
#define RT_AUTO_LEVELS 5      // option "levels" = -1 on fine supervoxels: regions of up to 2^5 supervoxels per edge
#endif
// 2-D tallies of a slab / launch round (RT_SLAB_TALLY): words shared by many lanes are added first, for up to
// tally_match rounds (option; automatic: kTallyMatchAuto on grids of up to kTallyMatchCols columns, none on larger
// ones: measured +5 % at 128 x 128 columns of block cloud, +12 % at 64 x 64 clear-sky columns, -1.2 % at 256 x 256 and
// -0.4 % at 1024 x 1024 columns of cumulus) and for as long as a word has at least kTallyMatchMin takers
constexpr int kTallyMatchAuto = 6;
constexpr long long kTallyMatchCols = 1 << 15;
constexpr int kTallyMatchMin = 3;
#ifndef RT_FUSE_LAUNCH
#define RT_FUSE_LAUNCH 1
#endif
// Wave priority from the start of a round until its gather has been issued (s_setprio; back to 0 for the
// arithmetic that follows): a wave that is about to ask memory for something goes first, so more gathers are under
// way at any time (+1 %)
#ifndef RT_PRIO
#define RT_PRIO 1
#endif
#ifndef RT_TH_SLAB
#define RT_TH_SLAB 24       // photons waiting for the slab stage that get it served ahead of the larger stages
#endif
// traverse steps a lane may take in one round of the traverse stage (a lane that crossed a face inside the grid
// goes on into the next region at once instead of waiting for the next round)
// the default pool size: the one instantiated without the event counters as well (the others always count)
#ifndef RT_FAST_POOL
#define RT_FAST_POOL 128
#endif
#ifndef RT_T_STEPS
#define RT_T_STEPS 2
#endif
// the event stage takes Philox's round keys from the kernel arguments (1) or derives them from the seed each round (0)
#ifndef RT_ROUND_KEYS
#define RT_ROUND_KEYS 1
#endif
constexpr int kPoolBlock = 64;

// One traverse step of the pool kernel INSIDE THE GRIDDED SLAB on the photon held in the lane's locals (x, y,
// z, tau, ux, uy, uz, age, ci, cj, ck, slab, ns): cross one majorant region or stop at the collision site
// (tau then carries 1/M of the region to the event stage, which draws the next tau anyway);
// IX, IY, IZ are the direction reciprocals. Used by the traverse stage and, fused, at the end of the new-
// direction stage (a photon that has just been given a direction always traverses next: doing it in the same
// round saves the slot selection and the LDS round trip of one round per scattering). The region is the
// box of supervoxel (ci, cj, ck) -- its six extents ride beside M in the table entry; HIER = false (no box larger
// than its supervoxel anywhere) skips the box arithmetic. Inside a box the supervoxel indices follow the position
// (kept within the box and never back along an axis: a coordinate on a face may round to either side). A photon that leaves the slab through
// its top or bottom moves its slab index and waits for the slab stage (PS_SLAB), which owns everything
// outside the grid: the clear-air slabs, the domain top and the surface. 1/M is finite in the tables
// (majorant_reciprocal), so tau * (1/M) is a number whatever the region holds.
#define RT_TRAVERSE_STEP(IX, IY, IZ, MM) \
                { \
                    /* the step guard (a position that is not a number, or a photon of age 31 in a tick round): the step */ \
                    /* is taken regardless -- its indices stay inside the tables whatever the position holds -- and */ \
                    /* the photon is retired after it: no else-branch around the hot code */ \
                    /* (age holds the five top bits alone: adding a tick to age 31 wraps -- no wave-uniform tick flag to test) */ \
                    const uint32_t age_old = age; \
                    age += age_inc; \
                    const bool bad = !(z == z) || !(x == x) || !(y == y) || age < age_old; \
                    if (STATS) ++ev_step; \
                    const float2 mm = (MM); \
                    __builtin_amdgcn_s_setprio(0); \
                    int bx0 = ci, by0 = cj, bz0 = ck, bx1 = ci + 1, by1 = cj + 1, bz1 = ck + 1; \
                    const float inv_m = region_reciprocal(mm.x); \
                    if (HIER) \
                    { \
                        /* the supervoxel's own box (rt_boxes_kernel): six extents of 5 bits */ \
                        const uint32_t eb = __float_as_uint(mm.y); \
                        bx0 = ci - (int)(eb & 31u); bx1 = ci + 1 + (int)((eb >> 5) & 31u); \
                        by0 = cj - (int)((eb >> 10) & 31u); by1 = cj + 1 + (int)((eb >> 15) & 31u); \
                        bz0 = ck - (int)((eb >> 20) & 31u); bz1 = ck + 1 + (int)(eb >> 25); \
                    } \
                    const bool px = (IX) > 0.0f, py = (IY) > 0.0f, up = (IZ) > 0.0f; \
                    const int cfx = px ? bx1 : bx0, cfy = py ? by1 : by0, cfz = up ? bz1 : bz0; \
                    const float fx = cfx == a.nci ? a.xs : (float)cfx * a.hx; \
                    const float fy = cfy == a.ncj ? a.ys : (float)cfy * a.hy; \
                    const float fz = cfz == a.nck ? a.zc1 : fma_((float)cfz, a.hz, a.zc0); \
                    const float tx = face_distance(fx, x, (IX)); \
                    const float ty = face_distance(fy, y, (IY)); \
                    const float tz = face_distance(fz, z, (IZ)); \
                    const bool y_first = ty < tx; \
                    const float txy = y_first ? ty : tx; \
                    const bool z_first = tz < txy; \
                    const float t_exit = z_first ? tz : txy; \
                    const float dn = tau * inv_m; \
                    const bool hit = t_exit <= dn; \
                    const float tt = hit ? t_exit : dn; \
                    x = fma_(ux, tt, x); y = fma_(uy, tt, y); z = fma_(uz, tt, z); \
                    if (HIER) \
                    { \
                        /* inside a box the indices follow the position, within the box and never back along an axis */ \
                        /* (the motion is monotone on each axis; a coordinate that sits on a face may round to either */ \
                        /* side, and the box of the supervoxel behind the face reaches back over it); for a box of one */ \
                        /* supervoxel the clamp returns the index itself. (The rule is needed, not only tidy: clamped to the */ \
                        /* box alone -- six instructions per step fewer -- 1 424 of 2^28 photons of the LES scene went back */ \
                        /* and forth over a face at distance zero until the step guard retired them.) */ \
                        /* (no clamp to the grid first, as cell_of would: the box lies inside it; a position that is not a */ \
                        /* number converts to 0 and is clamped like any other) */ \
                        ci = min(max((int)(x * a.inv_hx), px ? ci : bx0), px ? bx1 - 1 : ci); \
                        cj = min(max((int)(y * a.inv_hy), py ? cj : by0), py ? by1 - 1 : cj); \
                        ck = min(max((int)((z - a.zc0) * a.inv_hz), up ? ck : bz0), up ? bz1 - 1 : ck); \
                    } \
                    /* a face: carry tau on, step over the face (periodic in x and y); no face: 1/M rides in tau */ \
                    const float tau_on = fma_(-tt, mm.x, tau); \
                    tau = hit ? (tau_on > 0.0f ? tau_on : 0.0f) : inv_m; \
                    const bool hit_x = hit && !y_first && !z_first, hit_y = hit && y_first && !z_first, hit_z = hit && z_first; \
                    int ni = px ? bx1 : bx0 - 1, nj = py ? by1 : by0 - 1; \
                    float nx = fx, ny = fy; \
                    if (ni == a.nci) { ni = 0; nx = 0.0f; } \
                    if (ni < 0) { ni = a.nci - 1; nx = a.xs; } \
                    if (nj == a.ncj) { nj = 0; ny = 0.0f; } \
                    if (nj < 0) { nj = a.ncj - 1; ny = a.ys; } \
                    ci = hit_x ? ni : ci; x = hit_x ? nx : x; \
                    cj = hit_y ? nj : cj; y = hit_y ? ny : y; \
                    const int nk = up ? bz1 : bz0 - 1; \
                    const bool inside = (uint32_t)nk < (uint32_t)a.nck; \
                    z = hit_z ? fz : z; \
                    ck = (hit_z && inside) ? nk : ck; \
                    const bool leave = hit_z && !inside; \
                    slab += leave ? (up ? 1 : -1) : 0; \
                    ns = hit ? (leave ? PS_SLAB : PS_TRAVERSE) : PS_COLLIDE; \
                    if (STATS) ev_trans += hit ? 1u : 0u; \
                    if (__builtin_expect(bad, 0)) \
                    { \
                        tally_add(&cold_of(a)->events[RT_EV_ABORT]); \
                        ns = PS_LAUNCH; \
                    } \
                }

// The new direction of a photon that scatters (phase function + rotation, raytracer_kernels.cu:520-548) or is
// reflected by the surface (Lambert, :467-473), in place on the lane's locals ux, uy, uz. REFLECT / FROM_CLOUD say
// which; ASY is the cell's asymmetry, UMU and UPHI the two uniform draws. PLANAR: the 2D tracer's rules
// (raytracer.cpp:360-364, 280-282): turn the direction by +-acos(mu) within the x-z plane (angle-sum formulas
// instead of acos/atan2/sin/cos), reflect to (+-mu, 0, sqrt(1 - mu^2)); the sign is the top bit of the azimuth draw.
#define RT_NEW_DIRECTION(REFLECT, FROM_CLOUD, ASY, UMU, UPHI) \
                if (PLANAR) \
                { \
                    const bool plus = (UPHI) >= 0.5f; \
                    if (!(REFLECT)) \
                    { \
                        const float mu = (FROM_CLOUD) ? henyey_mu((ASY), (UMU)) : rayleigh_mu((UMU)); \
                        const float sn2 = rt_sqrt(__builtin_fmaxf(fma_(-mu, mu, 1.0f), 0.0f)); \
                        const float ssn = plus ? sn2 : -sn2; \
                        const float nx = fma_(mu, ux, ssn * uz); \
                        const float nz = fma_(mu, uz, -(ssn * ux)); \
                        ux = nx; uz = nz; \
                    } \
                    else \
                    { \
                        const float mu = rt_sqrt((UMU)); \
                        ux = plus ? mu : -mu; \
                        uz = rt_sqrt(fma_(-mu, mu, 1.0f)); \
                    } \
                    uy = 0.0f; \
                } \
                else \
                { \
                    float sp, cp; \
                    rt_sincos2pi((UPHI), sp, cp); \
                    if (!(REFLECT)) \
                    { \
                        const float mu = (FROM_CLOUD) ? henyey_mu((ASY), (UMU)) : rayleigh_mu((UMU)); \
                        const float sn = rt_sqrt(__builtin_fmaxf(fma_(-mu, mu, 1.0f) + RT_EPS, 0.0f)); \
                        /* the axis of the smallest direction component (raytracer_kernels.cu:523-533), as selects */ \
                        const float fx = __builtin_fabsf(ux), fy = __builtin_fabsf(uy), fz = __builtin_fabsf(uz); \
                        const bool x_lt_y = fx < fy; \
                        const bool pick_x = x_lt_y && fx < fz, pick_y = !x_lt_y && fy < fz; \
                        const float ax = pick_x ? 1.0f : 0.0f, ay = pick_y ? 1.0f : 0.0f, az = (pick_x || pick_y) ? 0.0f : 1.0f; \
                        /* dot(t1, dir): the reference's third term is t1.z*t1.z (raytracer_kernels.cu:73) */ \
                        const float proj = FIXED_DOT ? fma_(az, uz, fma_(ay, uy, ax * ux)) \
                                                     : fma_(az, az, fma_(ay, uy, ax * ux)); \
                        float ox = fma_(-proj, ux, ax), oy = fma_(-proj, uy, ay), oz = fma_(-proj, uz, az); \
                        const float inv_len = rt_rcp(rt_sqrt(fma_(oz, oz, fma_(oy, oy, ox * ox)))); \
                        ox *= inv_len; oy *= inv_len; oz *= inv_len; \
                        const float bx = fma_(uy, oz, -(uz * oy)); \
                        const float by = fma_(uz, ox, -(ux * oz)); \
                        const float bz = fma_(ux, oy, -(uy * ox)); \
                        const float nx = fma_(mu, ux, sn * fma_(sp, ox, cp * bx)); \
                        const float ny = fma_(mu, uy, sn * fma_(sp, oy, cp * by)); \
                        const float nz = fma_(mu, uz, sn * fma_(sp, oz, cp * bz)); \
                        ux = nx; uy = ny; uz = nz; \
                    } \
                    else \
                    { \
                        const float mu = rt_sqrt((UMU)); \
                        ux = mu * sp; \
                        uy = mu * cp; \
                        uz = rt_sqrt(fma_(-mu, mu, 1.0f) + RT_EPS); \
                    } \
                }

// One visit of the slab stage on the photon held in the lane's locals (x, y, z, tau, age, slab, cicj,
// ck, ns; UX, UY, UZ the direction, IZ the reciprocal of its z component, up = IZ > 0; cb the cold block):
// everything outside the grid. A photon inside a clear-air slab takes its step there (to the slab's top or bottom
// face, or to a collision site under the slab's own majorant, raytracer_kernels.cu:366-415 with the slab as the
// only region); a photon that reaches a face -- this way or by leaving the grid in the traverse stage -- arrives
// at the domain top (raytracer_kernels.cu:487-499), at the surface (:455-460; the albedo draw is stage E) or in
// the next slab. Used by the slab stage and, fused, at the end of the launch stage (every photon starts with
// this step when clear air lies above the clouds). The surface and top-of-domain tallies of a round are
// aggregated over runs of neighbouring lanes that hit the same column (RT_SLAB_TALLY below).
#define RT_SLAB_STEP(UX, UY, UZ, IZ, DIFFUSE) \
                bool on_face = true; \
                ns = PS_SLAB; \
                const int nslab_ = cb->nslab; \
                if (slab >= 0 && slab < nslab_) \
                { \
                    if (__builtin_expect(!(z == z) || !(x == x) || !(y == y) || (uint32_t)(age + age_inc) < age, 0)) \
                    { \
                        tally_add(&cb->events[RT_EV_ABORT]); \
                        ns = PS_LAUNCH; \
                        on_face = false; \
                    } \
                    else \
                    { \
                        age += age_inc; \
                        if (STATS) ++ev_step; \
                        const float2 mm = k->maj[cb->ncoarse + (uint32_t)slab]; \
                        const int fi = up ? slab + 1 : slab; \
                        const float fz = fi == 0 ? cb->zface[0] : (fi == 1 ? cb->zface[1] : (fi == 2 ? cb->zface[2] : cb->zface[3])); \
                        const float tz = face_distance(fz, z, (IZ)); \
                        const float dn = tau * mm.y; \
                        const bool hit = tz <= dn; \
                        const float tt = hit ? tz : dn; \
                        x = fma_((UX), tt, x); y = fma_((UY), tt, y); z = fma_((UZ), tt, z); \
                        x = wrap_periodic(x, k->xs, cb->inv_xs); y = wrap_periodic(y, k->ys, cb->inv_ys); \
                        if (!hit) \
                        { \
                            ns = PS_COLLIDE; \
                            tau = mm.y; \
                            on_face = false; \
                        } \
                        else \
                        { \
                            if (STATS) ++ev_trans; \
                            tau = fma_(-tt, mm.x, tau); \
                            tau = tau > 0.0f ? tau : 0.0f; \
                            z = fz; \
                            slab += up ? 1 : -1; \
                        } \
                    } \
                } \
                if (on_face) \
                { \
                    if (slab < 0 || slab >= nslab_) \
                    { \
                        const int i = cell_of(x, k->inv_dx, k->itot), j = cell_of(y, k->inv_dy, k->jtot); \
                        const uint32_t ij = (uint32_t)i + (uint32_t)j * (uint32_t)k->itot; \
                        if (slab < 0) \
                        { \
                            if (STATS) ++ev_sfc; \
                            tally_word = ((DIFFUSE) ? 3u : 2u) * k->ncol + ij; \
                            slab = 0; \
                            ns = PS_SURFACE; \
                        } \
                        else \
                        { \
                            if (STATS) ++ev_toa; \
                            tally_word = k->ncol + ij; \
                            ns = PS_LAUNCH; \
                        } \
                    } \
                    else if (slab == k->gslab) \
                    { \
                        cicj = (uint32_t)cell_of(x, k->inv_hx, k->nci) | ((uint32_t)cell_of(y, k->inv_hy, k->ncj) << 16); \
                        ck = up ? 0u : (uint32_t)(k->nck - 1); \
                        ns = PS_TRAVERSE; \
                    } \
                }

// Per-wave aggregation of a round's 2-D tallies (north_star; the reference adds one by one,
// raytracer_kernels.cu:256-261): WORD is the tally word a lane wants to bump (~0u: none). Lanes are served in slot
// order and slots are filled in launch order, so photons that were launched into one column and came down (or
// went up) side by side sit in neighbouring lanes: the first lane of a run of equal words adds the run's length
// with one atomic. Executed by the whole wave (it votes).
#define RT_SLAB_TALLY(WORD) \
            { \
                bool wants = (WORD) != 0xffffffffu; \
                /* First the few words that many lanes share, wherever those lanes sit: the direct beam of one launch column */ \
                /* lands in two to four surface columns (its slant times the photon's place inside the column), interleaved */ \
                /* at random over the lanes, so runs of EQUAL NEIGHBOURS are short while each word has a dozen takers. Up to */ \
                /* tally_match rounds of: first waiting lane's word -> ballot of its takers -> one atomic of their number. */ \
                unsigned long long todo = __ballot(wants); \
                /* The host turns this on (k->tally_match rounds) where it was measured to pay: small grids, where the direct */ \
                /* beam is a large part of what comes down and the tally atomics are the ceiling; on the cumulus fields most */ \
                /* arrivals are diffuse and a fruitless matching round per slab and launch round costs 0.4-1.2 %. */ \
                _Pragma("unroll 1") \
                for (int it = 0; MATCH && it < k->tally_match && todo != 0ull; ++it) \
                { \
                    const int first = __builtin_ctzll(todo); \
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)(WORD), first); \
                    const unsigned long long same = __ballot(wants && (WORD) == v); \
                    if (lane == first) \
                        tally_add(&k->tallies[v], (unsigned long long)__popcll(same)); \
                    wants = wants && (WORD) != v; \
                    todo &= ~same; \
                    /* a word with few takers: the lanes left are diffuse photons, all over the place -- stop matching */ \
                    if (__popcll(same) < kTallyMatchMin) break; \
                } \
                /* what is left (diffuse photons, all over the place): runs of equal neighbours as before */ \
                if (todo != 0ull) \
                { \
                    const uint32_t prev_word = (uint32_t)__shfl_up((int)(WORD), 1); \
                    const bool leader = wants && (lane == 0 || prev_word != (WORD)); \
                    const unsigned long long lm = __ballot(leader || !wants); \
                    if (leader) \
                    { \
                        const unsigned long long above = lane == 63 ? 0ull : lm >> (lane + 1); \
                        const int run = above ? 1 + __builtin_ctzll(above) : 64 - lane; \
                        tally_add(&k->tallies[(WORD)], (unsigned long long)run); \
                    } \
                } \
            }

// MATCH: RT_SLAB_TALLY's matching rounds are in the code (their number is PoolArgs::tally_match). The tuned kernel
// (no event counters, default pool) also exists without them: their mere presence costs the large scenes 0.1-0.4 %.
template <int PHYS, bool STATS, int NS, bool HIER, bool MATCH>
__global__ void __launch_bounds__(kPoolBlock)
rt_trace_pool_kernel(const PoolArgs a)
{
    using namespace rt;
    static_assert(NS % 16 == 0 && NS >= 64 && NS <= 256, "pool size");
    constexpr int kHalves = (NS + 63) / 64;
    constexpr bool PLANAR = PHYS == PHYS_PLANAR;
    constexpr bool FIXED_DOT = PHYS == PHYS_FIXED_DOT;
    __shared__ __attribute__((aligned(16))) uint32_t lds[NS * kPoolRec + 16];

    const int lane = threadIdx.x;
    uint32_t* const rec = lds;                                              // NS records
    uint8_t* const sel = reinterpret_cast<uint8_t*>(rec + NS * kPoolRec);   // 64 selected slots
    sel[lane] = 0;          // every entry is a slot number from the start: a round reads its entry whether it serves or not

    // photon IDs of this wave: wave_left IDs from wave_next on, a chunk taken from the device-wide cursor
    uint64_t wave_next = 0;
    uint32_t wave_left = 0;
    // HW_REG_XCC_ID (id 20), bits [3:0]: the XCD this wave runs on; ranges of XCDs found empty so far
    uint32_t ranges_dry = 0;              // == kXcds: no photon ID is left anywhere
    uint32_t round_ctr = 0;               // rounds of this wave: every kTickRounds-th is a tick of the step guard

    const uint32_t key0 = (uint32_t)a.seed, key1 = (uint32_t)(a.seed >> 32);

    // state (PS_*) of the slots this lane owns: selection needs no memory at all
    uint32_t stv[kHalves];
#pragma unroll
    for (int h = 0; h < kHalves; ++h)
        stv[h] = (h * 64 + lane < NS) ? PS_LAUNCH : PS_DONE;
    int n_t = 0, n_e = 0, n_x = 0, n_l = NS;              // photons waiting per stage (wave-uniform)
    // photon ID words of the slots this lane owns (slot h * 64 + lane): {pid_lo, pid_hi | nev << 8}. Only the
    // event and launch stages touch them; keeping them out of LDS is what lets 26 waves share a CU
    uint32_t id_lo[kHalves], id_hn[kHalves];
#pragma unroll
    for (int h = 0; h < kHalves; ++h) { id_lo[h] = 0u; id_hn[h] = 0u; }

    uint32_t ev_step = 0, ev_coll = 0, ev_real = 0, ev_scat = 0, ev_cscat = 0, ev_sfc = 0, ev_toa = 0,
             ev_abs = 0, ev_trans = 0;
    uint32_t rounds_t = 0, rounds_e = 0, lanes_t = 0, lanes_e = 0, rounds_x = 0, lanes_x = 0;

    int th_launch = a.th_launch;          // held in a scalar register: reloaded from the kernel arguments at the top of
    asm volatile("" : "+s"(th_launch));   // every round, its load would sit in front of the choice of the round's stage
    while (n_t + n_e + n_x + n_l > 0)
    {
        // serve the stage with the most photons waiting (ties: traverse, then event); launches
        // (regeneration) once th_launch slots are free or nothing else is left
        // (fullest first is the optimum: asking the dearer stages -- new direction, event -- to be 1/8 or 1/4 fuller
        // than the others before they are served, or less full, loses 0.4 .. 1.8 % on C4 and C3)
        // (written out for the scalar unit: the compiler's own version of these three lines took a detour through a
        // vector register and twenty scalar instructions, every round)
        uint32_t cls;
        {
            const int mx = max(n_t, n_e), busy = n_t + n_e + n_x, th = th_launch;
            asm("s_cmp_lt_i32 %[nt], %[ne]\n\t"
                "s_cselect_b32 %[c], 1, 0\n\t"                 /* cls = n_t >= n_e ? 0 : 1 */
                "s_cmp_gt_i32 %[nx], %[mx]\n\t"
                "s_cselect_b32 %[c], 4, %[c]\n\t"
                "s_cmp_ge_i32 %[nx], %[ts]\n\t"
                "s_cselect_b32 %[c], 4, %[c]\n\t"              /* (n_x >= RT_TH_SLAB) | (n_x > max(n_t, n_e)) ? 4 */
                "s_cmp_ge_i32 %[nl], %[th]\n\t"
                "s_cselect_b32 %[c], 3, %[c]\n\t"
                "s_cmp_eq_u32 %[busy], 0\n\t"
                "s_cselect_b32 %[c], 3, %[c]"                   /* (n_l >= th_launch) | (nothing else waits) ? 3 */
                : [c] "=&s"(cls)
                : [nt] "s"(n_t), [ne] "s"(n_e), [nx] "s"(n_x), [mx] "s"(mx), [nl] "s"(n_l), [th] "s"(th), [busy] "s"(busy),
                  [ts] "i"(RT_TH_SLAB)
                : "scc");
        }
        // a tick of the step guard every kTickRounds-th round: the counter advances by 2^32 / kTickRounds and the
        // carry out of the addition is the tick (two scalar instructions)
        uint32_t age_inc;
        asm("s_add_u32 %[c], %[c], %[step]\n\t"
            "s_cselect_b32 %[inc], %[one], 0"
            : [c] "+s"(round_ctr), [inc] "=s"(age_inc)
            : [step] "i"((int)(0x100000000ull / kTickRounds)), [one] "i"((int)kAgeOne)
            : "scc");

        // ---- gather up to 64 waiting slots into the lanes ----
        __builtin_amdgcn_s_setprio(RT_PRIO);
        asm volatile("" ::: "memory");
        int found = 0;
        int srv[kHalves];                // the lane that serves slot h * 64 + lane this round, or -1
#pragma unroll
        for (int h = 0; h < kHalves; ++h)
        {
            const uint32_t sidx = h * 64 + lane;
            const bool match = (stv[h] >> 2) == cls;
            const unsigned long long m = __ballot(match);
            const int rank = found + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const bool taken = match && (h == 0 || rank < 64);      // (the first 64 slots always fit the 64 lanes)
            if (taken)
                sel[rank] = (uint8_t)sidx;
            srv[h] = taken ? rank : -1;
            found += __popcll(m);
        }
        asm volatile("" ::: "memory");
        const int n = found < 64 ? found : 64;
        const bool active = lane < n;
        const uint32_t slot = sel[lane];         // (a lane that serves nothing reads a stale slot number and touches nothing)
        // group-major layout: group g of slot s at 16-byte element g*NS + s, so lanes serving
        // ascending slots read ascending addresses (a slot-major record stride of 64 B would put
        // every lane on the same four banks)
        float4* const r4base = reinterpret_cast<float4*>(rec) + slot;
#define R4(g) r4base[(g) * NS]
#define RW(k) rec[(((k) >> 2) * NS + slot) * 4 + ((k) & 3)]
#define BPERM(LANE, V) (uint32_t)__builtin_amdgcn_ds_bpermute((int)((LANE) << 2), (int)(V))
        uint32_t ns = PS_DONE;           // new state of the photon this lane serves
        uint32_t out_lo = 0u, out_hn = 0u;   // its new ID words (event and launch stages)

        // (the served stage's count loses its n lanes here, where the stage is known, and every count gains its new
        // arrivals at the end of the round: four conditional subtractions on the scalar unit less)
        if (cls == 0u)
        {
            n_t -= n;
            if (STATS) { ++rounds_t; lanes_t += n; }
            if (active)
            {
                const float4 g0 = R4(0);
                const float4 g1 = R4(1);
                const float4 g2 = R4(2);
                const uint32_t cicj = __float_as_uint(g1.w), cks = __float_as_uint(g2.w);
                float x = g0.x, y = g0.y, z = g0.z, tau = g0.w;
                const float ux = g1.x, uy = g1.y, uz = g1.z;
                int ci = (int)(cicj & 0xffffu), cj = (int)(cicj >> 16), ck = (int)(cks & 0xffffu);
                int slab = (int)((cks >> 16) & 0xffu);
                uint32_t age = cks & kAgeMask;
                ns = PS_TRAVERSE;
                RT_TRAVERSE_STEP(g2.x, g2.y, g2.z, a.maj[(uint32_t)((ck * a.ncj + cj) * a.nci + ci)])
#if RT_T_STEPS > 1
                // a lane that crossed a face inside the grid goes on into the next region at once (about half the
                // lanes of a round do): no selection and no LDS round trip for that step. Measured: 2 steps per
                // round +0.7 % on C4, +5.5 % on C3; a third gains nothing; asking for a minimum of lanes neither
#pragma unroll
                for (int extra = 1; extra < RT_T_STEPS; ++extra)
                    if (ns == PS_TRAVERSE)
                    {
                        RT_TRAVERSE_STEP(g2.x, g2.y, g2.z, a.maj[(uint32_t)((ck * a.ncj + cj) * a.nci + ci)])
                    }
#endif
                R4(0) = make_float4(x, y, z, tau);
                RW(7) = (uint32_t)ci | ((uint32_t)cj << 16);
                RW(11) = (uint32_t)ck | (((uint32_t)slab & 0xffu) << 16) | (cks & (1u << 24)) | age;
            }
        }
        else if (cls == 1u)
        {
            n_e -= n;
            const ArgPtr k = args_now();
            if (STATS) { ++rounds_e; lanes_e += n; }
            // the photon's ID words live in the registers of the lane that owns its slot
            uint32_t pid_lo = 0u, pn = 0u;
#pragma unroll
            for (int h = 0; h < kHalves; ++h)
            {
                const uint32_t lo = BPERM(slot & 63u, id_lo[h]), hn = BPERM(slot & 63u, id_hn[h]);
                if ((slot >> 6) == (uint32_t)h) { pid_lo = lo; pn = hn; }
            }
            // Event, new direction and the photon's next traverse step in ONE round. A photon that scatters (two
            // thirds of the collision tests of the cumulus scenes) or is reflected gets its new direction here, from
            // the cell record and the draws that are in registers anyway, and every photon that lives on -- scattered,
            // reflected, or after a null collision -- takes its next traverse step at once (it stays in the region it
            // is in; that region's majorant is asked for together with the cell record). No new-direction stage: its
            // rounds (a quarter of all), their slot selection, record round trips and stage counts are gone, and
            // the pool's photons wait for two frequent stages instead of three, so rounds are fuller.
            if (active)
            {
#if RT_ROUND_KEYS
                // asked for here, with the stage's other scalar loads and ahead of the record's LDS reads: issued where
                // Philox starts, their latency would sit in front of it
                uint32_t rk[20];
#pragma unroll
                for (int r = 0; r < 20; ++r)
                    rk[r] = k->round_keys[r];
#endif
                uint32_t cell = 0;
                float inv_m = 0.0f;
                float4 cr = make_float4(0.f, 0.f, 0.f, 0.f);
                const uint32_t pid_hi = pn & 0xffu;
                uint32_t nev = pn >> 8;
                const float4 g0 = R4(0);
                const float4 g1 = R4(1);
                const float4 g2 = R4(2);
                const uint32_t cicj = __float_as_uint(g1.w), cks = __float_as_uint(g2.w);
                // (surface events -- the albedo test and the Lambert reflection -- belong to the slab stage since round 3: a
                // photon that arrives at the surface has its event in the round of its arrival, and this stage, which is
                // most of the kernel, lost their selects and the reflection's arithmetic: they ran here for one or two lanes
                // in most rounds)
                constexpr bool on_surface = false;
                int ci = (int)(cicj & 0xffffu), cj = (int)(cicj >> 16), ck = (int)(cks & 0xffffu);
                int slab = (int)((cks >> 16) & 0xffu);
                uint32_t age = cks & kAgeMask;
                float2 mm_pre = make_float2(0.0f, 0.0f);
                {
                    // the traverse and slab stages leave 1/M of the region in the tau slot (tau itself is
                    // dead until the draw below); a photon on the surface has its column from (x, y)
                    // (no branch around the gather: a photon on the surface reads the cell above it and ignores it, and
                    // the stage's scalar argument loads are issued in one batch instead of one per basic block)
                    const int i = cell_of(g0.x, k->inv_dx, k->itot), j = cell_of(g0.y, k->inv_dy, k->jtot);
                    const uint32_t kz = (uint32_t)cell_of(g0.z, k->inv_dz, k->ktot);
#if RT_BRICK == 0
                    // plain [k][j][i] records: the cell's record and its tally word share their index arithmetic
                    // (+0.25 % on C4; doing these multiplies and the majorant index's with 24-bit operands, where the grid
                    // allows it, would give another 0.2 %: not worth a second code path)
                    const uint32_t ij = (uint32_t)j * (uint32_t)k->itot + (uint32_t)i, kcol = kz * k->ncol;
                    cr = k->cells[ij + kcol];
                    cell = ij + (on_surface ? 0u : kcol);       // the [k][j][i] index, for the absorption tally
#else
                    cr = k->cells[brick_index((uint32_t)i, (uint32_t)j, kz, k->nbi, k->nbij)];
                    cell = (uint32_t)i + (uint32_t)j * (uint32_t)k->itot;
                    cell += (on_surface ? 0u : kz) * k->ncol;   // the [k][j][i] index, for the absorption tally
#endif
                    inv_m = g0.w;
                    mm_pre = a.maj[slab == a.gslab ? (uint32_t)((ck * a.ncj + cj) * a.nci + ci) : 0u];
                }
                __builtin_amdgcn_s_setprio(0);
                const float kt = cr.x;
                // the ten round keys are wave-uniform and loop-invariant: left alone the compiler hoists all
                // twenty out of the wave's loop, spills them to VGPR lanes and reads them back with twenty
                // v_readlane per round; redefining the key here keeps the schedule on the scalar unit
#if RT_ROUND_KEYS
                const u32x4 w = philox4x32_10_keys(pid_lo, pid_hi, nev, 0u, rk);
#else
                uint32_t rk0 = key0, rk1 = key1;
                asm volatile("" : "+s"(rk0), "+s"(rk1));
                const u32x4 w = philox4x32_10(pid_lo, pid_hi, nev, 0u, rk0, rk1);
#endif
                ++nev;
                out_hn = pid_hi | (nev << 8);
                float tau_new = free_path_tau(u01(w.y));
                tau_new = tau_new > 0.0f ? tau_new : 0.0f;

                // Collision classification (raytracer_kernels.cu:507-518) or the surface's albedo test (:463-486), as
                // predicates: the half-dozen small divergent blocks this was cost more in exec-mask bookkeeping on the
                // CU's one scalar unit than their arithmetic.
                const float rr = u01(w.x);
                const float p_real = kt * inv_m;
                const float s_scat = cr.y * p_real;
                // the event index has 24 bits beside the ID's high byte: a photon that has used them all (2^24 events:
                // a conservative scene over a white surface can keep one alive that long) is abandoned like one that
                // runs into the step guard, instead of starting its random stream over at event 0
                const bool worn = nev >= 0xffffffu;
                const bool real = !on_surface && rr < p_real;
                const bool scat = real && rr <= s_scat;
                const bool absorbed = real && !scat && !worn;           // absorption, raytracer_kernels.cu:551-566
                const bool from_cloud = rr * kt < s_scat * cr.w;        // r/s < k_cloud/k_tot
                const bool reflect = on_surface && rr <= cold_of(a)->albedo && !worn;
                const bool turn = scat || reflect;
                const bool alive = !(absorbed || (on_surface && !reflect) || worn);
                if (__builtin_expect(worn, 0))
                    tally_add(&cold_of(a)->events[RT_EV_ABORT]);
                const float u_mu = u01(on_surface ? w.y : w.z), u_phi = u01(on_surface ? w.z : w.w);
                if (STATS)
                {
                    if (!on_surface) ++ev_coll;
                    if (real) ++ev_real;
                    if (scat) { ++ev_scat; if (from_cloud) ++ev_cscat; }
                    if (absorbed) ++ev_abs;
                }
                if (absorbed)
                {
                    // The atomic itself is left to the launch round that refills this slot: issued here, in 36 M event
                    // rounds per 2^28 photons with six absorbing lanes each, every round's next wait for a gather (an
                    // L2 hit since the tiled launch order) also waits for a read-modify-write in HBM (2.2 of 37.6 ms on
                    // C4); a launch round (6.5 M) adds for all the slots it refills at once.
                    const bool diffuse = (cks >> 24) & 1u;
                    RW(0) = 5u * k->ncol + (diffuse ? k->ncol * (uint32_t)k->ktot : 0u) + cell;
                }
                if (reflect)
                    tally_add(&k->tallies[4u * k->ncol + cell]);
                ns = absorbed ? PS_LAUNCH_TALLY : PS_LAUNCH;            // (a photon that lives on gets its state below)
                if (alive)
                {
                    float x = g0.x, y = g0.y, z = g0.z;
                    float tau = on_surface ? g0.w : tau_new;        // a reflected photon keeps its optical depth
                    float ux = g1.x, uy = g1.y, uz = g1.z, ix = g2.x, iy = g2.y, iz = g2.z;
                    uint32_t diffuse_bit = cks & (1u << 24);
                    if (turn)
                    {
                        RT_NEW_DIRECTION(reflect, from_cloud, cr.z, u_mu, u_phi)
                        ix = dir_reciprocal(ux); iy = dir_reciprocal(uy); iz = dir_reciprocal(uz);
                        diffuse_bit = 1u << 24;
                    }
                    if (slab == a.gslab)
                    {
                        RT_TRAVERSE_STEP(ix, iy, iz, mm_pre)
                        // (a second traverse step here, for the lanes that crossed a face, costs 6 %: the round is
                        // long as it is, and its registers decide the kernel's count)
                    }
                    else
                        ns = PS_SLAB;
                    R4(0) = make_float4(x, y, z, tau);
                    R4(1) = make_float4(ux, uy, uz, __uint_as_float((uint32_t)ci | ((uint32_t)cj << 16)));
                    R4(2) = make_float4(ix, iy, iz, __uint_as_float((uint32_t)ck | (((uint32_t)slab & 0xffu) << 16) | diffuse_bit | age));
                }
            }
        }
        else if (__builtin_expect(cls == 4u, 0))
        {
            n_x -= n;
            const ArgPtr k = args_now();
            // ---- slab stage: everything outside the grid, and the surface event of a photon that arrives down there ----
            __builtin_amdgcn_s_setprio(0);
            if (STATS) { ++rounds_x; lanes_x += n; }
            const ColdPtr cb = cold_of_k(k);
            uint32_t tally_word = 0xffffffffu;
            // the photon's ID words, for the Philox block of a surface event (registers of the lane that owns the slot)
            uint32_t pid_lo = 0u, pn = 0u;
#pragma unroll
            for (int h = 0; h < kHalves; ++h)
            {
                const uint32_t lo = BPERM(slot & 63u, id_lo[h]), hn = BPERM(slot & 63u, id_hn[h]);
                if ((slot >> 6) == (uint32_t)h) { pid_lo = lo; pn = hn; }
            }
            out_hn = pn;
            if (active)
            {
                const float4 g0 = R4(0);
                const float4 g1 = R4(1);
                const float iz = __uint_as_float(RW(10));
                const uint32_t cks = RW(11);
                float x = g0.x, y = g0.y, z = g0.z, tau = g0.w;
                int slab = (int)(int8_t)((cks >> 16) & 0xffu);
                const bool up = iz > 0.0f;
                uint32_t cicj = 0u, ck = cks & 0xffffu, age = cks & kAgeMask;
                uint32_t diffuse_bit = cks & (1u << 24);
                // (bit 26: the photon reached the surface in the fused first step of its launch round, which has
                // counted its arrival; its event is still to come)
                if ((cks >> 26) & 1u)
                    ns = PS_SURFACE;
                else
                {
                    RT_SLAB_STEP(g1.x, g1.y, g1.z, iz, (cks >> 24) & 1u)
                }
                if (ns == PS_SURFACE)
                {
                    // the surface (raytracer_kernels.cu:463-486): reflected with probability albedo into a Lambertian
                    // direction (one Philox block: w0 the test, w1 and w2 cosine and azimuth; tau is kept), else the photon ends
                    const uint32_t pid_hi = pn & 0xffu;
                    uint32_t nev = pn >> 8;
                    uint32_t sk0 = key0, sk1 = key1;
                    asm volatile("" : "+s"(sk0), "+s"(sk1));
                    const u32x4 w = philox4x32_10(pid_lo, pid_hi, nev, 0u, sk0, sk1);
                    ++nev;
                    out_hn = pid_hi | (nev << 8);
                    const bool worn = nev >= 0xffffffu;             // (as in the event stage)
                    if (__builtin_expect(worn, 0))
                        tally_add(&cb->events[RT_EV_ABORT]);
                    ns = PS_LAUNCH;
                    if (u01(w.x) <= cb->albedo && !worn)
                    {
                        const int i = cell_of(x, k->inv_dx, k->itot), j = cell_of(y, k->inv_dy, k->jtot);
                        tally_add(&k->tallies[4u * k->ncol + (uint32_t)i + (uint32_t)j * (uint32_t)k->itot]);
                        float ux = g1.x, uy = g1.y, uz = g1.z;
                        RT_NEW_DIRECTION(true, false, 0.0f, u01(w.y), u01(w.z))
                        diffuse_bit = 1u << 24;
                        // upwards from the surface: through clear air, or -- where the grid reaches down to the ground -- back
                        // into the supervoxel it came out of (its indices are still in the record: the oracle keeps them too)
                        ns = slab == k->gslab ? PS_TRAVERSE : PS_SLAB;
                        R4(1) = make_float4(ux, uy, uz, g1.w);
                        RW(8) = __float_as_uint(dir_reciprocal(ux));
                        RW(9) = __float_as_uint(dir_reciprocal(uy));
                        RW(10) = __float_as_uint(dir_reciprocal(uz));
                    }
                }
                else if (ns == PS_TRAVERSE)
                    RW(7) = cicj;
                R4(0) = make_float4(x, y, z, tau);
                RW(11) = ck | (((uint32_t)slab & 0xffu) << 16) | diffuse_bit | age;
            }
            RT_SLAB_TALLY(tally_word)
        }
        else
        {
            n_l -= n;
            const ArgPtr k = args_now();
            // ---- launch (regeneration), raytracer_kernels.cu:180-208: next IDs of this wave's range ----
            __builtin_amdgcn_s_setprio(0);
            const ColdPtr cb = cold_of_k(k);
            while (__builtin_expect(wave_left == 0u && ranges_dry < (uint32_t)kXcds, 0))
            {
                // next chunk of IDs from this XCD's range (or the next range that has any left): one
                // returning atomic per chunk, broadcast from lane 0
                const uint32_t chunk = cb->chunk;
                const uint64_t total = cb->total, sub = cb->sub_size;
                const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
                const uint32_t x = (xcc + ranges_dry) & (kXcds - 1);
                const uint64_t sub_lo = (uint64_t)x * sub;
                const uint64_t sub_hi = sub_lo + sub < total ? sub_lo + sub : total;
                unsigned long long c0 = 0;
                if (lane == 0)
                    c0 = __hip_atomic_fetch_add(k->cursor + x * kCursorStride, (unsigned long long)chunk,
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t c0_lo = __builtin_amdgcn_readfirstlane((uint32_t)c0);
                const uint32_t c0_hi = __builtin_amdgcn_readfirstlane((uint32_t)(c0 >> 32));
                c0 = (((unsigned long long)c0_hi << 32) | c0_lo) + sub_lo;
                if (sub_lo >= total || c0 >= sub_hi)
                {
                    ++ranges_dry;
                }
                else
                {
                    wave_next = cb->pid_begin + c0;
                    wave_left = sub_hi - c0 < (uint64_t)chunk ? (uint32_t)(sub_hi - c0) : chunk;
                }
            }
            // absorption tallies the event stage left with the slots (PS_LAUNCH_TALLY): the states live with the owners
            // (adding at the end of the stage instead, behind its loads and its returning cursor atomic: no difference)
            {
                uint32_t st = 0u;
#pragma unroll
                for (int h = 0; h < kHalves; ++h)
                {
                    const uint32_t v = BPERM(slot & 63u, stv[h]);
                    if ((slot >> 6) == (uint32_t)h) st = v;
                }
                if (active && st == PS_LAUNCH_TALLY)
                {
                    // (multi-GPU jobs count absorptions straight into the 32-bit image that crosses xGMI: rt_bind_atmos_wire)
                    if (uint32_t* const a32 = cb->atm32)
                        __hip_atomic_fetch_add(a32 + RW(0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        tally_add(&k->tallies[RW(0)]);
                }
            }
            const bool go = active && (uint32_t)lane < wave_left;
            if (active && ranges_dry < (uint32_t)kXcds)
                ns = PS_LAUNCH;          // no ID left in this chunk: wait for the next one
            uint32_t toa_col = 0xffffffffu;
            uint32_t tally_word = 0xffffffffu;      // surface / domain-top tally of the fused first step
            if (go)
            {
                uint64_t my_pid = wave_next + (uint64_t)lane;
                if (const uint32_t* const order = cb->sorted)       // launch sort: position in the launch -> ID
                    my_pid = cb->pid_begin + (uint64_t)order[(uint32_t)(my_pid - cb->pid_begin)];
                else if (cb->sob_a != 0u)                           // Sobol launch in box order
                    my_pid = (cb->pid_begin & ~0xffffffffull) | (uint64_t)sobol_position_to_id(cb, (uint32_t)(my_pid - cb->pid_begin));
                const uint32_t pid_lo = (uint32_t)my_pid, pid_hi = (uint32_t)(my_pid >> 32);
                uint32_t lk0 = key0, lk1 = key1;            // see the event stage: no hoisted, spilled round keys
                asm volatile("" : "+s"(lk0), "+s"(lk1));
                const u32x4 w = philox4x32_10(pid_lo, pid_hi, 0u, 0u, lk0, lk1);
                float tau_new = free_path_tau(u01(w.z));
                tau_new = tau_new > 0.0f ? tau_new : 0.0f;
                const int log2k = cb->launch_log2k;
                float x, y;
                if (log2k == kLaunchSobol)
                {
                    uint32_t s0, s1;
                    sobol_2d(pid_lo, cb->sob_x, cb->sob_y, s0, s1);
                    x = k->xs * u01(s0);
                    y = k->ys * u01(s1);
                }
                else if (log2k < 0)
                {
                    x = k->xs * u01(w.x);
                    y = k->ys * u01(w.y);
                }
                else
                {
                    const uint32_t c = launch_column_index_exact(log2k, k->ncol, my_pid, lk0, lk1);
                    uint32_t ic, jc;
                    launch_column(cb->div_itot_mul, cb->div_itot_shift, cb->launch_div, cb->launch_tile_log2, c, ic, jc);
                    const float dx = cb->dx, dy = cb->dy;
                    x = fma_(u01(w.x), dx, (float)ic * dx);
                    y = fma_(u01(w.y), dy, (float)jc * dy);
                }
                const int i = cell_of(x, k->inv_dx, k->itot), j = cell_of(y, k->inv_dy, k->jtot);
                toa_col = (uint32_t)i + (uint32_t)j * (uint32_t)k->itot;
                int slab = cb->nslab - 1;
                float z = cb->ztop, tau = tau_new;
                const float sun_x = cb->sun_x, sun_y = cb->sun_y, sun_z = cb->sun_z, iz = cb->inv_sun_z;
                uint32_t cicj = 0u, ck = 0u, age = 0u;
                if (slab == k->gslab)
                {
                    cicj = (uint32_t)cell_of(x, k->inv_hx, k->nci) | ((uint32_t)cell_of(y, k->inv_hy, k->ncj) << 16);
                    ck = (uint32_t)(k->nck - 1);
                    ns = PS_TRAVERSE;
                }
                else
                {
                    // the first step, through the clear air above the clouds, at once
#if RT_FUSE_LAUNCH
                    const bool up = false;
                    RT_SLAB_STEP(sun_x, sun_y, sun_z, iz, false)
#else
                    ns = PS_SLAB;
#endif
                }
                R4(0) = make_float4(x, y, z, tau);
                R4(1) = make_float4(sun_x, sun_y, sun_z, __uint_as_float(cicj));
                R4(2) = make_float4(cb->inv_sun_x, cb->inv_sun_y, iz,
                                    __uint_as_float(ck | (((uint32_t)slab & 0xffu) << 16) | (ns == PS_SURFACE ? 1u << 26 : 0u) | age));
                out_lo = pid_lo;
                out_hn = pid_hi | (1u << 8);
            }
            // TOA-down tally. Stratified launch puts runs of consecutive lanes into one column: the
            // first lane of a run adds the run's length (one atomic per run instead of one per photon).
            // (A sorted launch has counted its columns tile by tile before the tracer started.)
            if (cb->toa_done == 0u)
            {
                const uint32_t prev_col = (uint32_t)__shfl_up((int)toa_col, 1);
                const bool leader = go && (lane == 0 || prev_col != toa_col);
                const unsigned long long lm = __ballot(leader);
                const int n_go = __popcll(__ballot(go));
                if (leader)
                {
                    const unsigned long long above = lane == 63 ? 0ull : lm >> (lane + 1);
                    const int next = above ? lane + 1 + __builtin_ctzll(above) : n_go;
                    tally_add(&k->tallies[toa_col], (unsigned long long)(next - lane));
                }
            }
#if RT_FUSE_LAUNCH
            RT_SLAB_TALLY(tally_word)
#endif
            // advance the wave's ID cursor by the lanes served
            const uint32_t adv = (uint32_t)n < wave_left ? (uint32_t)n : wave_left;
            wave_next += adv;
            wave_left -= adv;
        }

#undef R4
#undef RW
        // ---- the owners of the served slots take the new state and ID words from the lanes that served them ----
#pragma unroll
        for (int h = 0; h < kHalves; ++h)
        {
            const uint32_t v = BPERM((uint32_t)srv[h] & 63u, ns);
            if (srv[h] >= 0) stv[h] = v;
        }
        if (cls != 0u)
        {
#pragma unroll
            for (int h = 0; h < kHalves; ++h)
            {
                const uint32_t hn = BPERM((uint32_t)srv[h] & 63u, out_hn);
                if (srv[h] >= 0) id_hn[h] = hn;
            }
            if (cls == 3u)
            {
#pragma unroll
                for (int h = 0; h < kHalves; ++h)
                {
                    const uint32_t lo = BPERM((uint32_t)srv[h] & 63u, out_lo);
                    if (srv[h] >= 0) id_lo[h] = lo;
                }
            }
        }
#undef BPERM
        // ---- update the stage counts ----
        const uint32_t ncls = active ? (ns >> 2) : 5u;
        const int d_t = __popcll(__ballot(ncls == 0u)), d_e = __popcll(__ballot(ncls == 1u)),
                  d_l = __popcll(__ballot(ncls == 3u)), d_x = __popcll(__ballot(ncls == 4u));
        n_t += d_t; n_e += d_e; n_l += d_l; n_x += d_x;
    }

    if (STATS)
    {
        unsigned long long* const events = cold_of(a)->events;
        const uint32_t ev[9] = {ev_step, ev_coll, ev_real, ev_scat, ev_cscat, ev_sfc, ev_toa, ev_abs, ev_trans};
#pragma unroll
        for (int e = 0; e < 9; ++e)
            if (ev[e])
                __hip_atomic_fetch_add(&events[e], (unsigned long long)ev[e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0)
        {
            const uint32_t sch[8] = {rounds_t, rounds_e, 0u, lanes_t, lanes_e, 0u, rounds_x, lanes_x};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                __hip_atomic_fetch_add(&events[RT_N_EVENTS + e], (unsigned long long)sch[e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Majorant grid of the cloud slab: per supervoxel the max of k_tot over its cells, stored as {M, 1/M}
// (generalises k_ext_null of raytracer_3d_io.cu:73-74 from the domain to supervoxels), and the sum and
// number of its cells for the growth criterion of the boxes (rt_boxes_kernel).
// What the boxes' growth reads of a supervoxel, in one 16-byte record (one load per visit of rt_boxes_kernel, where
// three arrays cost three: the pre-pass is bound by the latency of those loads).
struct __attribute__((aligned(16))) SuperVoxel
{
    float m;        // majorant: max of k_tot over the supervoxel's cells
    int cnt;        // number of its cells
    double sum;     // sum of k_tot over them, cells in (k, j, i) order
};

__global__ void __launch_bounds__(64)
rt_majorant_kernel(const float4* __restrict__ cells, SuperVoxel* __restrict__ sv, uint32_t ncoarse,
                   int itot, int jtot, uint32_t nbi, uint32_t nbij, int k_lo, int k_end,
                   int sx, int sy, int sz, int nci, int ncj)
{
    // One THREAD per supervoxel, cells visited in (k, j, i) order: the double-precision sum of k_tot feeds the
    // growth criterion of rt_boxes_kernel, which must decide exactly as the oracle's does, so its rounding
    // must not depend on a reduction tree. Runs once per scene.
    const uint32_t e = blockIdx.x * 64u + threadIdx.x;
    if (e >= ncoarse)
        return;
    const int ci = e % nci, cj = (e / nci) % ncj, ck = e / (nci * ncj);
    const int i0 = ci * sx, j0 = cj * sy, k0 = k_lo + ck * sz;
    const int i1 = min(i0 + sx, itot), j1 = min(j0 + sy, jtot), k1 = min(k0 + sz, k_end);
    float m = 0.0f;
    double sum = 0.0;
    int cnt = 0;
    for (int k = k0; k < k1; ++k)
        for (int j = j0; j < j1; ++j)
            for (int i = i0; i < i1; ++i)
            {
                const float v = cells[brick_index((uint32_t)i, (uint32_t)j, (uint32_t)k, nbi, nbij)].x;
                m = v > m ? v : m;
                sum += (double)v;
                ++cnt;
            }
    SuperVoxel r;
    r.m = m; r.cnt = cnt; r.sum = sum;
    sv[e] = r;
}

// Majorant regions over the supervoxel grid: every supervoxel gets a BOX OF ITS OWN -- the supervoxels
// [ci - xm, ci + xp] x [cj - ym, cj + yp] x [ck - zm, ck + zp] around it, clipped at the grid, at most max_extent per
// axis -- that a photon standing in it crosses as ONE region under the box's majorant. The boxes of neighbouring
// supervoxels overlap; they are no partition of the grid and need not be: each step of the walk is made under a
// majorant that bounds the extinction over the whole segment, which is all that null-collision tracking asks for (the
// optical depth carried over a face is decremented with the same majorant the segment was sampled with). Round 3's
// first regions were the aligned boxes of a buddy merge (2^lx x 2^ly x 2^lz supervoxels): a photon in clear air then
// crosses the faces of a binary partition wherever it goes (8.2 crossings per photon on the LES grid); with a box grown
// AROUND the supervoxel it is in, it goes to the nearest cloud edge or box limit in one step (4.8).
// Growth: from the supervoxel alone, one layer at a time, trying the six faces in turn (x-, x+, y-, y+, z-, z+) until
// six attempts in a row have failed; a layer is taken if (M - mean k_tot) * len <= tau for the grown box, M its
// majorant, mean its volume mean and len = 6 V / S (the box's edge for a cube, three times the thickness of a wide
// slab): the optical depth of null collisions a crossing photon meets. One THREAD per supervoxel, the layer's cells
// in (k, j, i) order, sums in double: the oracle's copy of this routine must decide identically.
// out[e] = {M of the box, the six extents in 5 bits each: xm | xp << 5 | ym << 10 | yp << 15 | zm << 20 | zp << 25}.
#ifndef RT_BOX_TILE_X
#define RT_BOX_TILE_X 8
#endif
constexpr uint32_t kBoxTileX = RT_BOX_TILE_X, kBoxTileY = 64u / RT_BOX_TILE_X;      // a wave's tile of supervoxels

__global__ void __launch_bounds__(64)
rt_boxes_kernel(const SuperVoxel* __restrict__ sv,
                float2* __restrict__ out, unsigned int* __restrict__ any_box, uint32_t ncoarse, int nci, int ncj, int nck,
                float hx, float hy, float hz, float tau, int max_extent)
{
    // Which supervoxel a thread takes (its result does not depend on it): a wave takes a tile of 8 x 8 supervoxels of
    // one level -- neighbours grow similar boxes, so its lanes walk layers of similar size -- and the waves are dealt
    // out so that every XCD (workgroups go to the XCDs in turn) works its way through a band of its own, tile by tile
    // and level by level: what a tile's boxes read (a neighbourhood of at most 70 x 70 supervoxels over all levels,
    // 16 bytes each) stays in that XCD's L2. Dealt out in table order, 64 consecutive supervoxels per wave and
    // neighbouring waves on different XCDs, nearly every load went out to the Infinity Cache and the pre-pass was
    // bound by that latency (379 ms on the LES scene's 512 x 512 x 20 supervoxels).
    const uint32_t ntx = ((uint32_t)nci + kBoxTileX - 1u) / kBoxTileX, nty = ((uint32_t)ncj + kBoxTileY - 1u) / kBoxTileY, nblocks = ntx * nty * (uint32_t)nck;
    const uint32_t per_xcd = gridDim.x / (uint32_t)kXcds;          // the grid is kXcds * ceil(nblocks / kXcds)
    const uint32_t lb = (blockIdx.x % (uint32_t)kXcds) * per_xcd + blockIdx.x / (uint32_t)kXcds;
    if (lb >= nblocks)
        return;
    const uint32_t tile = lb / (uint32_t)nck;
    const int ti = (int)((tile % ntx) * kBoxTileX + (threadIdx.x % kBoxTileX)), tj = (int)((tile / ntx) * kBoxTileY + (threadIdx.x / kBoxTileX));
    if (ti >= nci || tj >= ncj)
        return;
    const uint32_t e = ((lb % (uint32_t)nck) * (uint32_t)ncj + (uint32_t)tj) * (uint32_t)nci + (uint32_t)ti;
    if (e >= ncoarse)
        return;
    const int dims[3] = {nci, ncj, nck};
    const double h[3] = {(double)hx, (double)hy, (double)hz};
    int b[6];
    b[0] = (int)(e % (uint32_t)nci); b[2] = (int)((e / (uint32_t)nci) % (uint32_t)ncj); b[4] = (int)(e / ((uint32_t)nci * (uint32_t)ncj));
    const int ci = b[0], cj = b[2], ck = b[4];
    const int c3[3] = {ci, cj, ck};
    b[1] = b[0] + 1; b[3] = b[2] + 1; b[5] = b[4] + 1;
    float m = sv[e].m;
    double sum = sv[e].sum;
    long long cnt = sv[e].cnt;
    int fails = 0, dir = 0;
    while (fails < 6)
    {
        const int ax = dir >> 1;
        const bool plus = dir & 1;
        bool ok = plus ? b[2 * ax + 1] < dims[ax] : b[2 * ax] > 0;
        ok = ok && b[2 * ax + 1] - b[2 * ax] < max_extent && (plus ? b[2 * ax + 1] - 1 - c3[ax] : c3[ax] - b[2 * ax]) < 31;   /* 5-bit extents */
        if (ok)
        {
            int l[6] = {b[0], b[1], b[2], b[3], b[4], b[5]};        // the new layer
            if (plus) { l[2 * ax] = b[2 * ax + 1]; l[2 * ax + 1] = b[2 * ax + 1] + 1; }
            else { l[2 * ax] = b[2 * ax] - 1; l[2 * ax + 1] = b[2 * ax]; }
            float lm = 0.0f;
            double ls = 0.0;
            long long lc = 0;
            for (int kk = l[4]; kk < l[5]; ++kk)
                for (int jj = l[2]; jj < l[3]; ++jj)
                    for (int ii = l[0]; ii < l[1]; ++ii)
                    {
                        const uint32_t q = (uint32_t)((kk * ncj + jj) * nci + ii);
                        const SuperVoxel r = sv[q];
                        lm = r.m > lm ? r.m : lm;
                        ls += r.sum;
                        lc += r.cnt;
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
    const uint32_t ext = (uint32_t)(ci - b[0]) | ((uint32_t)(b[1] - 1 - ci) << 5) | ((uint32_t)(cj - b[2]) << 10)
                         | ((uint32_t)(b[3] - 1 - cj) << 15) | ((uint32_t)(ck - b[4]) << 20) | ((uint32_t)(b[5] - 1 - ck) << 25);
    out[e] = make_float2(m, __uint_as_float(ext));
    if (ext != 0u)
        *any_box = 1u;
}

// ------------------------------------------------------------------------------------------
// Pre-pass: pack the four input fields for the tracer and reduce per level
//   lev_max[k]   = max over the level of gas + cloud          (raytracer_3d_io.cu:73-76)
//   lev_cloud[k] = 1 if any asy > 0 on the level              (raytracer_kernels.cu:274-284)
// One workgroup per (level, chunk); float max via integer atomicMax is exact for values >= 0.
__global__ void __launch_bounds__(kBlock)
rt_pack_kernel(const float* __restrict__ k_gas, const float* __restrict__ k_cloud,
               const float* __restrict__ asy,
               const float* __restrict__ ssa, float4* __restrict__ cells,
               unsigned int* __restrict__ lev_max_bits, unsigned int* __restrict__ lev_cloud,
               unsigned int* __restrict__ invalid,      // [0] cells with values no tracer can use, [1] the lowest such cell
               uint32_t ncol, uint32_t itot, uint32_t nbi, uint32_t nbij, int chunks_per_level)
{
    const int k = blockIdx.x / chunks_per_level;
    const int chunk = blockIdx.x % chunks_per_level;
    const size_t base = (size_t)k * ncol;
    float m = 0.0f;
    unsigned int cloudy = 0;
    for (uint32_t c = (uint32_t)chunk * kBlock + threadIdx.x; c < ncol; c += (uint32_t)chunks_per_level * kBlock)
    {
        const float g = k_gas[base + c], cl = k_cloud[base + c], as = asy[base + c];
        const float kt = g + cl;
        const uint32_t j = c / itot, i = c - j * itot;
        cells[brick_index(i, j, (uint32_t)k, nbi, nbij)] = make_float4(kt, ssa[base + c], as, cl);
        m = kt > m ? kt : m;
        cloudy |= as > 0.0f ? 1u : 0u;
        // What the tracer cannot work with (the reference reads anything, raytracer_3d_io.cu:36-57): an extinction that is
        // negative or not finite; where there is any extinction, a single-scattering albedo outside [0, 1] or an
        // asymmetry outside (-1, 1) (a cell without extinction never collides: its ssa and asy are not looked at).
        const float sa = ssa[base + c];
        const bool ok_k = g >= 0.0f && g < __builtin_inff() && cl >= 0.0f && cl < __builtin_inff();
        const bool ok_s = !(kt > 0.0f) || (sa >= 0.0f && sa <= 1.0f && __builtin_fabsf(as) < 1.0f);
        if (__builtin_expect(!(ok_k && ok_s), 0))
        {
            atomicAdd(&invalid[0], 1u);
            atomicMin(&invalid[1], (unsigned int)(base + c));
        }
    }
    // wave reduction (64 lanes), then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        const float o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    const unsigned long long any = __ballot(cloudy != 0);
    if ((threadIdx.x & 63) == 0)
    {
        atomicMax(&lev_max_bits[k], __float_as_uint(m));
        if (any) atomicOr(&lev_cloud[k], 1u);
    }
}

// ---- Launch sort ---------------------------------------------------------------------------------------
// With uniform launch positions (the reference's sampling, the library default) the photons a wave holds enter the
// grid anywhere: on a grid larger than the caches every gather and every majorant lookup misses (C4: 72.9 ms per
// 2^28 photons against 39.8 ms with the stratified launch order), and on any grid the 64 lanes of a launch round
// add to 64 different columns of toa_down. But a photon's launch position is a function of its ID alone (first
// Philox block), trajectories are keyed on the ID and tallies are integer sums -- so the ORDER in which a launch
// takes its IDs is free. These kernels put the IDs of a batch in the order of their launch tile (2^tile_shift
// columns square, tiles row by row) with a counting sort: (1) every workgroup counts the tiles of its share of
// the IDs, (2, 3) the counts become each workgroup's first output position per tile, (4) every workgroup walks
// its share again and writes each ID to its tile's next position (no global atomics; a workgroup's writes to one
// tile are consecutive words), (5) toa_down of the batch is counted per column in LDS, tile by tile. Three extra
// Philox blocks and a 4-byte write and two reads per photon: 2.0 ms per 2^28 photons. The tracer then takes
// list[position] instead of position, and an XCD's eighth of the list is a band of tile rows. Measured per 2^28
// photons, uniform launch: C4 72.9 -> 41.2 ms, C3 30.3 -> 28.7, C2 28.8 -> 20.9, C1 28.5 -> 17.8 (on the small grids
// the gain is toa_down: 2^28 atomics on 16 384 addresses). Order within a tile is whatever the LDS atomics of
// pass 4 give (it differs from run to run; the tallies cannot).
constexpr int kSortBlock = 1024;
constexpr uint32_t kSortMaxTiles = 4096;

struct SortArgs
{
    uint32_t* count;              // [workgroup][tile] IDs (count pass) -> first output position (offsets pass)
    uint32_t* tile_base;          // [tile] IDs of the tile (offsets pass) -> first output position of the tile (scan pass)
    uint32_t* list;               // n words out: IDs relative to pid_begin, tile by tile
    uint64_t seed, pid_begin;
    uint32_t n, per_block, nblocks;   // IDs of the batch; IDs per workgroup; workgroups of the count and scatter passes
    uint32_t tile_shift, ntx, ntiles;
    int itot, jtot;
    float xs, ys, inv_dx, inv_dy;
    unsigned long long* toa_down; // pass 5
    uint32_t nslices;             // pass 5: workgroups per tile
    int sobol;                    // launch positions from the Sobol sequence (no Philox block needed to find a tile)
    uint32_t sob_x, sob_y;
};

// launch column of photon `pid`: the launch stage's own arithmetic (uniform or Sobol positions)
__device__ __forceinline__ void launch_ij(const SortArgs& s, uint64_t pid, uint32_t key0, uint32_t key1, uint32_t& i, uint32_t& j)
{
    uint32_t wx, wy;
    if (s.sobol)
        rt::sobol_2d((uint32_t)pid, s.sob_x, s.sob_y, wx, wy);
    else
    {
        const rt::u32x4 w = rt::philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), 0u, 0u, key0, key1);
        wx = w.x; wy = w.y;
    }
    i = (uint32_t)rt::cell_of(s.xs * rt::u01(wx), s.inv_dx, s.itot);
    j = (uint32_t)rt::cell_of(s.ys * rt::u01(wy), s.inv_dy, s.jtot);
}

__device__ __forceinline__ uint32_t launch_tile(const SortArgs& s, uint64_t pid, uint32_t key0, uint32_t key1)
{
    uint32_t i, j;
    launch_ij(s, pid, key0, key1, i, j);
    return (j >> s.tile_shift) * s.ntx + (i >> s.tile_shift);
}

// pass 1: every workgroup counts the tiles of ITS contiguous share of the IDs
__global__ __launch_bounds__(kSortBlock) void rt_sort_count_kernel(const SortArgs s)
{
    __shared__ uint32_t hist[kSortMaxTiles];
    for (uint32_t t = threadIdx.x; t < s.ntiles; t += kSortBlock)
        hist[t] = 0u;
    __syncthreads();
    const uint32_t key0 = (uint32_t)s.seed, key1 = (uint32_t)(s.seed >> 32);
    const uint64_t lo = (uint64_t)blockIdx.x * s.per_block;
    const uint64_t hi = lo + s.per_block < (uint64_t)s.n ? lo + s.per_block : (uint64_t)s.n;
    for (uint64_t r = lo + threadIdx.x; r < hi; r += kSortBlock)
        atomicAdd(&hist[launch_tile(s, s.pid_begin + r, key0, key1)], 1u);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < s.ntiles; t += kSortBlock)
        s.count[(size_t)blockIdx.x * s.ntiles + t] = hist[t];
}

// pass 2 (one wave per tile): counts of the tile over the workgroups -> each workgroup's first position within
// the tile, and the tile's total
__global__ __launch_bounds__(64) void rt_sort_offsets_kernel(const SortArgs s)
{
    const uint32_t t = blockIdx.x, lane = threadIdx.x;
    const uint32_t per = (s.nblocks + 63u) / 64u;
    const uint32_t b0 = lane * per, b1 = b0 + per < s.nblocks ? b0 + per : s.nblocks;
    uint32_t sum = 0u;
    for (uint32_t b = b0; b < b1; ++b)
        sum += s.count[(size_t)b * s.ntiles + t];
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    uint32_t run = inc - sum;
    for (uint32_t b = b0; b < b1; ++b)
    {
        const uint32_t c = s.count[(size_t)b * s.ntiles + t];
        s.count[(size_t)b * s.ntiles + t] = run;
        run += c;
    }
    if (lane == 63u) s.tile_base[t] = inc;
}

// pass 3 (one workgroup): exclusive scan of the tile totals, in place
__global__ __launch_bounds__(kSortBlock) void rt_sort_scan_kernel(uint32_t* tile_base, uint32_t ntiles)
{
    constexpr uint32_t per = kSortMaxTiles / kSortBlock;
    __shared__ uint32_t wave_sum[kSortBlock / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t v[per], sum = 0u;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k)
    {
        const uint32_t t = threadIdx.x * per + k;
        v[k] = t < ntiles ? tile_base[t] : 0u;
        sum += v[k];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63u) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t w = 0; w < wave; ++w)
        run += wave_sum[w];
#pragma unroll
    for (uint32_t k = 0; k < per; ++k)
    {
        const uint32_t t = threadIdx.x * per + k;
        if (t < ntiles) tile_base[t] = run;
        run += v[k];
    }
}

// pass 4: every workgroup walks its share again, kSortSeg IDs at a time: the IDs of a segment are put in tile order
// in LDS (count, scan, place) and then written out, so that consecutive lanes store consecutive words of a tile's
// run instead of 64 unrelated words (the L2 takes a request per run, not per ID). A tile's runs follow each other at
// the positions passes 2 and 3 reserved for this workgroup: no global atomics.
#ifndef RT_SORT_SEG_PER
#define RT_SORT_SEG_PER 8
#endif
constexpr int kSortSegPer = RT_SORT_SEG_PER, kSortSeg = kSortBlock * kSortSegPer;
template <uint32_t MAXT>
__global__ __launch_bounds__(kSortBlock) void rt_sort_scatter_kernel(const SortArgs s)
{
    __shared__ uint32_t next[MAXT];               // the tile's next output position (of this workgroup's range)
    __shared__ uint32_t cnt[MAXT];                // IDs of the segment per tile, then the tile's first place in the segment
    __shared__ uint32_t sid[kSortSeg], sdst[kSortSeg];
    __shared__ uint32_t wave_sum[kSortBlock / 64];
    constexpr uint32_t per = MAXT / kSortBlock;   // tiles per thread in the scan
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t t = threadIdx.x; t < s.ntiles; t += kSortBlock)
        next[t] = s.tile_base[t] + s.count[(size_t)blockIdx.x * s.ntiles + t];
    const uint32_t key0 = (uint32_t)s.seed, key1 = (uint32_t)(s.seed >> 32);
    const uint64_t lo = (uint64_t)blockIdx.x * s.per_block;
    const uint64_t hi = lo + s.per_block < (uint64_t)s.n ? lo + s.per_block : (uint64_t)s.n;
    for (uint64_t seg = lo; seg < hi; seg += kSortSeg)
    {
        for (uint32_t t = threadIdx.x; t < s.ntiles; t += kSortBlock)
            cnt[t] = 0u;
        __syncthreads();
        uint32_t where[kSortSegPer];              // tile | rank among the segment's IDs of the tile << 12
#pragma unroll
        for (int k = 0; k < kSortSegPer; ++k)
        {
            const uint64_t r = seg + (uint64_t)k * kSortBlock + threadIdx.x;
            where[k] = 0xffffffffu;
            if (r < hi)
            {
                const uint32_t t = launch_tile(s, s.pid_begin + r, key0, key1);
                where[k] = t | (atomicAdd(&cnt[t], 1u) << 12);
            }
        }
        __syncthreads();
        uint32_t c[per], sum = 0u;
#pragma unroll
        for (uint32_t k = 0; k < per; ++k)
        {
            const uint32_t t = threadIdx.x * per + k;
            c[k] = t < s.ntiles ? cnt[t] : 0u;
            sum += c[k];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += o;
        }
        if (lane == 63u) wave_sum[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (uint32_t w = 0; w < wave; ++w)
            run += wave_sum[w];
#pragma unroll
        for (uint32_t k = 0; k < per; ++k)
        {
            const uint32_t t = threadIdx.x * per + k;
            if (t < s.ntiles) cnt[t] = run;
            run += c[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSortSegPer; ++k)
            if (where[k] != 0xffffffffu)
            {
                const uint32_t t = where[k] & 0xfffu, rank = where[k] >> 12, place = cnt[t] + rank;
                sid[place] = (uint32_t)(seg + (uint64_t)k * kSortBlock + threadIdx.x);
                sdst[place] = next[t] + rank;
            }
        __syncthreads();
        const uint32_t nseg = hi - seg < (uint64_t)kSortSeg ? (uint32_t)(hi - seg) : (uint32_t)kSortSeg;
        for (uint32_t p = threadIdx.x; p < nseg; p += kSortBlock)
            s.list[sdst[p]] = sid[p];
#pragma unroll
        for (uint32_t k = 0; k < per; ++k)
        {
            const uint32_t t = threadIdx.x * per + k;
            if (t < s.ntiles) next[t] += c[k];
        }
        // (the barrier after the next segment's zeroing of cnt orders these against its readers)
    }
}

// pass 5 (tiles of at most 64 x 64 columns; nslices workgroups per tile): toa_down of the batch, counted per
// column in LDS over the tile's stretch of the list -- one atomic per column and workgroup instead of one per
// photon in the tracer, whose sorted launches put 64 different columns into the lanes of a launch round
__global__ __launch_bounds__(kSortBlock) void rt_sort_toa_kernel(const SortArgs s)
{
    __shared__ uint32_t hist[kSortMaxTiles];
    const uint32_t t = blockIdx.x / s.nslices, slice = blockIdx.x - t * s.nslices;
    const uint32_t edge = 1u << s.tile_shift, mask = edge - 1u;
    for (uint32_t b = threadIdx.x; b < edge * edge; b += kSortBlock)
        hist[b] = 0u;
    __syncthreads();
    const uint32_t key0 = (uint32_t)s.seed, key1 = (uint32_t)(s.seed >> 32);
    const uint32_t first = s.tile_base[t], last = t + 1u < s.ntiles ? s.tile_base[t + 1u] : s.n;
    const uint32_t per = (last - first + s.nslices - 1u) / s.nslices;
    const uint32_t lo = first + slice * per, hi = lo + per < last ? lo + per : last;
    for (uint32_t p = lo + threadIdx.x; p < hi; p += kSortBlock)
    {
        uint32_t i, j;
        launch_ij(s, s.pid_begin + (uint64_t)s.list[p], key0, key1, i, j);
        atomicAdd(&hist[((j & mask) << s.tile_shift) | (i & mask)], 1u);
    }
    __syncthreads();
    const uint32_t tj = t / s.ntx, ti = t - tj * s.ntx;
    for (uint32_t b = threadIdx.x; b < edge * edge; b += kSortBlock)
        if (hist[b] != 0u)
        {
            const uint32_t i = (ti << s.tile_shift) | (b & mask), j = (tj << s.tile_shift) | (b >> s.tile_shift);
            tally_add(&s.toa_down[(size_t)j * (uint32_t)s.itot + i], (unsigned long long)hist[b]);
        }
}

// Stores the launch block of the next pool launch and rewinds the ID cursors, in stream order.
__global__ void rt_launch_block_kernel(ColdBlock* dst, const ColdBlock src, unsigned long long* cursor)
{
    if (blockIdx.x != 0)
        return;
    if (threadIdx.x == 0)
        *dst = src;
    if (threadIdx.x < kXcds)
        cursor[threadIdx.x * kCursorStride] = 0ull;
}

__global__ void rt_add_u64_kernel(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}

// The two kernels either side of a 32-bit all-reduce of the tallies (halves the bytes on
// xGMI). Narrow: wire[i] = low word of tallies[i]; any word at or above 2^guard_shift (whose sum over the ranks
// could pass 2^32) raises the flag in wire[n]. Widen (after the sum, flag included): if no rank raised the flag
// the sums replace the 64-bit tallies, otherwise those are left as they were for the caller's 64-bit fallback.
__global__ void rt_narrow_kernel(const unsigned long long* __restrict__ tallies, uint32_t* __restrict__ wire, size_t n,
                                 uint32_t* __restrict__ flag, int guard_shift)
{
    uint32_t big = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const unsigned long long t = tallies[i];
        big |= (uint32_t)((t >> guard_shift) != 0ull);
        wire[i] = (uint32_t)t;
    }
    if (__ballot(big != 0u) != 0ull && (threadIdx.x & 63) == 0)
        atomicOr(flag, 1u);
}

// native_from: words from this index on were COUNTED in the wire (rt_bind_atmos_wire: the absorption arrays), so they
// have no 64-bit original and are widened whatever the flag says; n for an ordinary wire
__global__ void rt_widen_kernel(unsigned long long* __restrict__ tallies, const uint32_t* __restrict__ wire, size_t n,
                                size_t native_from)
{
    const size_t first = wire[n] != 0u ? native_from : 0;
    for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        tallies[i] = (unsigned long long)wire[i];
}

__global__ void rt_selftest_philox_kernel(const uint32_t* ctr, size_t n, uint32_t k0, uint32_t k1, uint32_t* out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n)
    {
        const rt::u32x4 r = rt::philox4x32_10(ctr[4 * t], ctr[4 * t + 1], ctr[4 * t + 2], ctr[4 * t + 3], k0, k1);
        out[4 * t] = r.x; out[4 * t + 1] = r.y; out[4 * t + 2] = r.z; out[4 * t + 3] = r.w;
    }
}

__global__ void rt_selftest_math_kernel(int which, const float* in, size_t n, float* out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n)
    {
        const float v = in[t];
        float s, c;
        switch (which)
        {
            case 0: out[t] = rt::rt_log(v); break;
            case 1: out[t] = rt::rt_inv_cbrt(v); break;
            case 2: rt::rt_sincos2pi(v, s, c); out[t] = s; break;
            case 3: rt::rt_sincos2pi(v, s, c); out[t] = c; break;
            case 4: out[t] = rt::rayleigh_mu(v); break;
            default: out[t] = rt::henyey_mu(0.85f, v); break;
        }
    }
}
// launch positions of n consecutive photon IDs (uniform or Sobol), with the tracer's own arithmetic
__global__ void rt_selftest_launch_kernel(int sobol, uint32_t k0, uint32_t k1, uint32_t sob_x, uint32_t sob_y,
                                          unsigned long long pid_begin, size_t n, float xs, float ys, float* xy, uint32_t* words)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n)
        return;
    const unsigned long long pid = pid_begin + t;
    uint32_t wx, wy;
    if (sobol)
        rt::sobol_2d((uint32_t)pid, sob_x, sob_y, wx, wy);
    else
    {
        const rt::u32x4 w = rt::philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), 0u, 0u, k0, k1);
        wx = w.x; wy = w.y;
    }
    xy[2 * t] = xs * rt::u01(wx);
    xy[2 * t + 1] = ys * rt::u01(wy);
    if (words) { words[2 * t] = wx; words[2 * t + 1] = wy; }
}

// rt_sqrt / rt_rcp against the compiler's IEEE sqrtf / division over a range of bit patterns
__global__ void rt_selftest_exact_kernel(int which, uint32_t first, unsigned long long count, unsigned long long* out)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x)
    {
        const uint32_t bits = first + (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float want = which == 0 ? __builtin_sqrtf(x) : 1.0f / x;
        const float got = which == 0 ? rt::rt_sqrt(x) : rt::rt_rcp(x);
        if (__float_as_uint(want) != __float_as_uint(got))
        {
            atomicAdd(&out[0], 1ull);
            atomicMin(&out[1], (unsigned long long)bits);
        }
    }
}
} // namespace

// ------------------------------------------------------------------------------------------
struct rt_context
{
    int device = 0;
    int itot = 0, jtot = 0, ktot = 0;
    float dx = 0, dy = 0, dz = 0;
    size_t ncol = 0, ncell = 0, nwords = 0;
    uint32_t nbi = 0, nbij = 0;       // brick_index of the packed cells
    size_t ncell_packed = 0;          // cells of the packed array (odd grid sizes padded to whole bricks)
    int cu_count = 256;

    float4* d_cells = nullptr;        // {k_tot, ssa, asy, k_cloud} per cell
    float2* d_maj = nullptr;          // majorant table: supervoxels of the cloud slab, then the slabs
    int levels = -1;                  // largest region: boxes of up to 2^levels supervoxels per edge (0 = no merging,
                                      // -1 = automatic: RT_AUTO_LEVELS or 0)
    int merge_tau_milli = 750;        // a box grows while (M - mean k_tot) * 6 V / S <= merge_tau_milli / 1000
    bool hier = false;                // any region larger than a supervoxel (selects the kernel variant)
    size_t maj_capacity = 0, ncoarse = 0;
    bool maj_dirty = true;
    int sgx = -1, sgy = -1, sgz = -1; // supervoxel size in cells (0 = no grid, slabs only; -1 = automatic:
                                      // nearest to 50 x 50 x 50 m with a box each, or 8x8x4 without, see build_majorants)
    bool built_fine = false;          // what the automatic choices were made for (wants_fine below)
    int th_event = 20, th_scatter = 16, low_traverse = 20;
    int pool = RT_FAST_POOL;          // photons per wave kept in LDS by the pool kernel (0 = lane-resident kernel)
    int pool_waves_per_cu = 32;       // cap on resident waves per CU of the pool kernel
    int th_launch = 40;               // pool kernel: free slots that trigger a launch round
    bool planar = false;                // 2D tracer's scattering rule (pool kernel only)
    int chunk = 256;                    // pool kernel: photon IDs a wave takes from the cursor at a time
    int launch_log2k = -1;            // >= 0: stratified launch, 2^k consecutive photon IDs per column
    int launch_tile_log2 = 0;         // stratified launch: columns counted in tiles of 2^this columns square (0 = row-major)
    int tally_match = -1;             // pool kernel: matching rounds of RT_SLAB_TALLY; -1: automatic (kTallyMatchCols)
    bool sobol_boxes = true;          // Sobol launch, pool kernel: whole nets of IDs are taken box by box (no sort)
    int launch_sort = -1;             // uniform launch, pool kernel: take the IDs in the order of their launch tile
                                      // (1; 0 = never; -1 = automatic: calls of 2^19 photons or more)
    int sort_batch_log2 = 30;         // IDs sorted (and traced) at a time: 4 bytes of list each (every batch ends with a
                                      // drain of the pools: 2^30 photons in one batch 154.7 ms on C4, in four 158.4)
    int sort_tile_log2 = 5;           // launch tile: 2^this columns square (made larger while there are > 4096 tiles)
    uint32_t* d_sort = nullptr;       // the list, then the sort's tile tables (allocated on first use)
    size_t sort_capacity = 0;         // IDs the list holds
    unsigned long long* d_tallies = nullptr;
    bool owns_tallies = false;
    // second tally buffer of the context (rt_swap_tally_buffer) and the asynchronous download (rt_download_tallies_async):
    // a sweep traces job k+1 into one buffer while job k's counts leave the other over a copy stream
    unsigned long long* d_tallies_alt = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t dl_ready = nullptr;            // recorded on the context's stream: the counts to download are complete
    hipEvent_t dl_done[2] = {nullptr, nullptr};   // recorded on the copy stream per ticket: the download has left the buffer
    const unsigned long long* dl_src[2] = {nullptr, nullptr};   // the device buffer each ticket reads
    int dl_next = 0;
    unsigned long long* d_events = nullptr;
    uint32_t* d_wire = nullptr;       // nwords + 1 words: 32-bit image of the tallies for rt_allreduce (allocated on first use)
    uint32_t* atm_wire = nullptr;     // rt_bind_atmos_wire: caller's wire whose absorption words the tracer counts into
    uint64_t atm_photons = 0;         // photons traced into atm_wire since it was last zeroed (a word cannot pass this)
    bool wire32 = true;               // rt_allreduce sums 32-bit counts when every count allows it
    ColdBlock* d_cold = nullptr;

    bool have_optics = false, have_scene = false;
    std::vector<float> lev_max;
    std::vector<unsigned int> lev_cloud;
    rt_slab_info slabs{};
    float inv_kmaj[3] = {0, 0, 0};
    float albedo = 0.2f, zenith = 0, azimuth = 0;

    bool fixed_dot = false, stats = false;
    int blocks_per_cu = 6;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
};

namespace
{
// Slabs from the per-level reductions: cloud layer = levels k_lo..k_hi holding any asy > 0
// (cloud_mask_kernel, raytracer_kernels.cu:263-305, without its indexing bugs); majorant of a
// slab = max of gas + cloud over its levels (raytracer_3d_io.cu:60-77 takes it over the domain).
// 1/M as the tables hold it: finite (FLT_MAX where M is 0 or, with a region level packed into its low bits, a
// denormal), so that the traverse step's tau * (1/M) is a number for tau == 0 too; the CPU checker holds the same function
inline float majorant_reciprocal(float m)
{
    return std::fmin(1.0f / m, FLT_MAX);
}

void build_slabs(rt_context* c)
{
    int k_lo = -1, k_hi = -1;
    for (int k = 0; k < c->ktot; ++k)
        if (c->lev_cloud[k]) { if (k_lo < 0) k_lo = k; k_hi = k; }
    int lo[3], hi[3], n = 0;
    if (k_lo < 0) { lo[0] = 0; hi[0] = c->ktot; n = 1; }
    else
    {
        if (k_lo > 0) { lo[n] = 0; hi[n] = k_lo; ++n; }
        lo[n] = k_lo; hi[n] = k_hi + 1; ++n;
        if (k_hi + 1 < c->ktot) { lo[n] = k_hi + 1; hi[n] = c->ktot; ++n; }
    }
    rt_slab_info& s = c->slabs;
    s.nslab = n; s.k_lo = k_lo; s.k_hi = k_hi;
    s.cloud_min = k_lo < 0 ? 0.0f : (float)k_lo * c->dz;
    s.cloud_max = k_lo < 0 ? 0.0f : (float)(k_hi + 1) * c->dz;
    for (int i = 0; i < 3; ++i) { s.kmaj[i] = 0.0f; c->inv_kmaj[i] = 0.0f; }
    for (int i = 0; i < n; ++i)
    {
        float m = 0.0f;
        for (int k = lo[i]; k < hi[i]; ++k) m = c->lev_max[k] > m ? c->lev_max[k] : m;
        s.kmaj[i] = m;
        c->inv_kmaj[i] = majorant_reciprocal(m);
        s.zface[i] = (float)lo[i] * c->dz;
    }
    s.zface[n] = (float)c->ktot * c->dz;
    for (int i = n + 1; i < 4; ++i) s.zface[i] = s.zface[n];
}

// workgroups of the launch sort's count and scatter passes (two of 1024 threads fit a CU)
#ifndef RT_SORT_BLOCKS_X2
#define RT_SORT_BLOCKS_X2 4
#endif
size_t sort_blocks(const rt_context* c) { return (size_t)RT_SORT_BLOCKS_X2 * (size_t)c->cu_count / 2u; }

// packed cells that fit the Infinity Cache: uniform launch positions need no help there
bool small_grid(const rt_context* c) { return 16ull * (unsigned long long)c->ncell <= (256ull << 20); }

// Fine supervoxels with a box each need their table cache-resident: the stratified launch order provides that
// on any grid, the launch sort (pool kernel, uniform launch) too, and so does a grid that fits the Infinity Cache.
bool wants_fine(const rt_context* c)
{
    return c->launch_log2k >= 0 || small_grid(c) || (c->pool > 0 && c->launch_sort != 0);
}

int build_majorants(rt_context* c)
{
    rt_slab_info& s = c->slabs;
    // Automatic choices (options left at -1). Fine supervoxels with boxes need their 8-byte-per-supervoxel table
    // cache-resident: the stratified launch order provides that on any grid, a grid that fits the Infinity Cache
    // too. Uniform launch positions over a larger grid (every access a miss) do better with the coarse grid and no
    // boxes -- measured on 1024x1024x128: 3.60 vs 2.28 G photons/s.
    // The fine supervoxel is the whole number of cells nearest to 50 m x 50 m x 50 m (1..8 cells per edge): with a box
    // around every supervoxel a face costs a photon little (it is crossed once, not re-met as the boundary of an
    // aligned partition), so the supervoxel can follow the cloud edges more closely than the 80 x 80 x 50 m that was
    // the optimum for merged aligned regions -- 2x2x2 cells on the 25 m LES grid (C4: 29.4 against 30.7 ms with
    // 3x3x2), 1x1x2 on the 50 x 50 x 25 m grids (C3: 19.4 against 21.3 ms with 2x2x2).
    const bool fine = wants_fine(c);
    const auto nearest = [](double target, double d) { const int n = (int)(target / d + 0.5); return n < 1 ? 1 : (n > 8 ? 8 : n); };
    int sgx = c->sgx >= 0 ? c->sgx : (fine ? nearest(50.0, (double)c->dx) : 8),
        sgy = c->sgy >= 0 ? c->sgy : (fine ? nearest(50.0, (double)c->dy) : 8),
        sgz = c->sgz >= 0 ? c->sgz : (fine ? nearest(50.0, (double)c->dz) : 4);
    // (an automatic size is coarsened until the table's indices fit: 16 bits per axis in a photon's record)
    while (c->sgx < 0 && sgx > 0 && (c->itot + sgx - 1) / sgx > 32767) sgx *= 2;
    while (c->sgy < 0 && sgy > 0 && (c->jtot + sgy - 1) / sgy > 32767) sgy *= 2;
    if (s.k_lo >= 0 && sgx > 0 && sgy > 0 && sgz > 0)
        while ((c->sgx < 0 || c->sgy < 0) &&
               (unsigned long long)((c->itot + sgx - 1) / sgx) * (unsigned long long)((c->jtot + sgy - 1) / sgy) *
                   (unsigned long long)((s.k_hi + 1 - s.k_lo + sgz - 1) / sgz) >= (1ull << 31))
        {
            if (c->sgx < 0) sgx *= 2;
            if (c->sgy < 0) sgy *= 2;
        }
    const int levels = c->levels >= 0 ? c->levels : (fine ? RT_AUTO_LEVELS : 0);
    c->built_fine = fine;
    const int g = (s.k_lo >= 0 && sgx > 0 && sgy > 0 && sgz > 0) ? (s.k_lo > 0 ? 1 : 0) : -1;
    s.gslab = g; s.nci = s.ncj = s.nck = 0; s.hx = s.hy = s.hz = 0.0f;
    size_t ncoarse = 0;
    if (g >= 0)
    {
        s.nci = (c->itot + sgx - 1) / sgx;
        s.ncj = (c->jtot + sgy - 1) / sgy;
        s.nck = (s.k_hi + 1 - s.k_lo + sgz - 1) / sgz;
        s.hx = (float)sgx * c->dx; s.hy = (float)sgy * c->dy; s.hz = (float)sgz * c->dz;
        ncoarse = (size_t)s.nci * s.ncj * s.nck;
        if (s.nci > 32767 || s.ncj > 32767 || ncoarse >= (1ull << 31))
            return fail(RT_ERR_ARG, "majorant grid %dx%dx%d too large", s.nci, s.ncj, s.nck);
    }
    if (ncoarse + 3 > c->maj_capacity)
    {
        if (c->d_maj) RT_HIP(hipFree(c->d_maj));
        c->d_maj = nullptr;
        RT_HIP(hipMalloc(&c->d_maj, sizeof(float2) * (ncoarse + 3)));
        c->maj_capacity = ncoarse + 3;
    }
    c->hier = false;
    if (ncoarse)
    {
        // per supervoxel: majorant, sum and number of cells; then every supervoxel's box (rt_boxes_kernel)
        SuperVoxel* d_sv = nullptr;
        unsigned int* d_any = nullptr;
        unsigned int any = 0;
        hipError_t e = hipMalloc(&d_sv, sizeof(SuperVoxel) * ncoarse);
        if (e == hipSuccess) e = hipMalloc(&d_any, sizeof(unsigned int));
        if (e == hipSuccess) e = hipMemsetAsync(d_any, 0, sizeof(unsigned int), c->stream);
        if (e == hipSuccess)
        {
            const dim3 grid((unsigned)((ncoarse + 63) / 64)), block(64);
            hipLaunchKernelGGL(rt_majorant_kernel, grid, block, 0, c->stream,
                               c->d_cells, d_sv, (uint32_t)ncoarse, c->itot, c->jtot, c->nbi, c->nbij,
                               s.k_lo, s.k_hi + 1, sgx, sgy, sgz, s.nci, s.ncj);
            // (a wave per tile of 8 x 8 supervoxels and level, the grid padded to a whole number per XCD)
            const size_t btiles = (size_t)((s.nci + kBoxTileX - 1) / kBoxTileX) * (size_t)((s.ncj + kBoxTileY - 1) / kBoxTileY) * (size_t)s.nck;
            const dim3 bgrid((unsigned)(((btiles + kXcds - 1) / kXcds) * kXcds));
            hipLaunchKernelGGL(rt_boxes_kernel, bgrid, block, 0, c->stream, d_sv, c->d_maj, d_any,
                               (uint32_t)ncoarse, s.nci, s.ncj, s.nck, s.hx, s.hy, s.hz,
                               (float)((double)c->merge_tau_milli / 1000.0), 1 << levels);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&any, d_any, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (d_sv) (void)hipFree(d_sv);
        if (d_any) (void)hipFree(d_any);
        RT_HIP(e);
        c->hier = any != 0;
    }
    float2 slab_m[3];
    for (int i = 0; i < 3; ++i) slab_m[i] = make_float2(s.kmaj[i], c->inv_kmaj[i]);
    RT_HIP(hipMemcpyAsync(c->d_maj + ncoarse, slab_m, sizeof(slab_m), hipMemcpyHostToDevice, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    c->ncoarse = ncoarse;
    c->maj_dirty = false;
    return RT_OK;
}

int pack_from_device(rt_context* c, const float* d_gas, const float* d_cloud, const float* d_ssa, const float* d_asy)
{
    unsigned int* d_lev = nullptr;
    c->have_optics = false;
    RT_HIP(hipMalloc(&d_lev, sizeof(unsigned int) * (2 * c->ktot + 2)));
    RT_HIP(hipMemsetAsync(d_lev, 0, sizeof(unsigned int) * (2 * c->ktot + 1), c->stream));
    RT_HIP(hipMemsetAsync(d_lev + 2 * c->ktot + 1, 0xff, sizeof(unsigned int), c->stream));
    int chunks = (int)((c->ncol + kBlock * 8 - 1) / (kBlock * 8));
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(rt_pack_kernel, dim3(c->ktot * chunks), dim3(kBlock), 0, c->stream,
                       d_gas, d_cloud, d_asy, d_ssa, c->d_cells, d_lev, d_lev + c->ktot, d_lev + 2 * c->ktot,
                       (uint32_t)c->ncol, (uint32_t)c->itot, c->nbi, c->nbij, chunks);
    RT_HIP(hipGetLastError());
    std::vector<unsigned int> lev(2 * c->ktot + 2);
    RT_HIP(hipMemcpyAsync(lev.data(), d_lev, sizeof(unsigned int) * (2 * c->ktot + 2), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(hipStreamSynchronize(c->stream));
    RT_HIP(hipFree(d_lev));
    if (lev[2 * c->ktot] != 0u)
    {
        // refused: the context is left without optics (a photon in such a cell would spin until the step guard)
        const size_t cell = lev[2 * c->ktot + 1];
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* const src[4] = {d_gas, d_cloud, d_ssa, d_asy};
        for (int f = 0; f < 4; ++f) RT_HIP(hipMemcpy(&v[f], src[f] + cell, sizeof(float), hipMemcpyDeviceToHost));
        return fail(RT_ERR_ARG, "rt_upload_optics: %u cells hold optical properties the tracer cannot use (extinction negative "
                                "or not finite; ssa outside [0,1] or |asy| >= 1 where there is extinction); first: cell "
                                "i=%zu j=%zu k=%zu with k_ext_gas=%g k_ext_cloud=%g ssa=%g asy=%g", lev[2 * c->ktot],
                    cell % (size_t)c->itot, (cell / (size_t)c->itot) % (size_t)c->jtot, cell / c->ncol,
                    (double)v[0], (double)v[1], (double)v[2], (double)v[3]);
    }
    c->lev_max.resize(c->ktot);
    c->lev_cloud.resize(c->ktot);
    for (int k = 0; k < c->ktot; ++k)
    {
        memcpy(&c->lev_max[k], &lev[k], 4);
        c->lev_cloud[k] = lev[c->ktot + k];
    }
    build_slabs(c);
    c->have_optics = true;
    c->maj_dirty = true;
    return RT_OK;
}
} // namespace

namespace
{
// The device address of the static scratch on `device`, resolved once per device with that device current
// (each GPU holds its own copy of the array at its own address).
std::mutex g_scratch_mutex;
unsigned long long* g_scratch_ptr[kMaxDevices] = {};

int scratch_of_device(int device, unsigned long long** out)
{
    if (device < 0 || device >= kMaxDevices)
        return fail(RT_ERR_ARG, "device %d beyond the %d devices the static scratch table covers", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device])
    {
        static_assert(sizeof(scratch) == sizeof(unsigned long long) * kScratchWords, "scratch size");
        static_assert(sizeof(ColdBlock) <= sizeof(unsigned long long) * (kScratchWords - kScratchColdWord), "cold block fits");
        void* p = nullptr;
        RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_scratch_ptr[device] = static_cast<unsigned long long*>(p);
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

// ---- the path's one exchange step: a sum all-reduce of the tallies over the GPUs of the node (RCCL over xGMI) ----
// RCCL is bound at run time (dlopen), so the library loads on hosts without it and, inside a process that already
// carries an RCCL (PyTorch ships its own copy), uses that one instead of mapping a second.
struct RcclApi
{
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclSum = 0, kNcclUint32 = 3, kNcclUint64 = 5;      // rccl.h: ncclSum, ncclUint32, ncclUint64
std::mutex g_rccl_mutex;
RcclApi g_rccl;
std::vector<std::pair<std::vector<int>, std::vector<void*>>> g_comm_sets;   // device list -> communicators: kept until the
// process ends ON PURPOSE (ncclCommDestroy from a static destructor would run after the HIP runtime has begun to shut
// down; the driver reclaims them with the process)

int load_rccl()
{
    if (g_rccl.handle) return RT_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) return fail(RT_ERR_HIP, "rt_allreduce: cannot load RCCL (librccl.so.1): %s", dlerror());
    RcclApi a;
    a.handle = h;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.CommInitAll || !a.AllReduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString)
        return fail(RT_ERR_HIP, "rt_allreduce: the RCCL library lacks one of the five entry points used");
    g_rccl = a;
    return RT_OK;
}

#define RT_NCCL(call)                                                                              \
    do {                                                                                           \
        const int r_ = (call);                                                                     \
        if (r_ != 0)                                                                               \
            return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// one communicator per GPU of the set, created once per distinct device list and kept for the process
int comms_for(const std::vector<int>& devices, std::vector<void*>** out)
{
    for (auto& e : g_comm_sets)
        if (e.first == devices) { *out = &e.second; return RT_OK; }
    std::vector<void*> comms(devices.size(), nullptr);
    RT_NCCL(g_rccl.CommInitAll(comms.data(), (int)devices.size(), devices.data()));
    g_comm_sets.emplace_back(devices, comms);
    *out = &g_comm_sets.back().second;
    return RT_OK;
}

int launch_narrow(rt_context* c, const unsigned long long* tallies, uint32_t* wire, int world)
{
    int guard_shift = 32;
    for (int r = 1; r < world; r <<= 1) --guard_shift;       // R counts below 2^(32 - ceil(log2 R)) sum below 2^32
    RT_HIP(hipMemsetAsync(wire + c->nwords, 0, sizeof(uint32_t), c->stream));
    // a wire the tracer counts its absorptions into holds those words already: only the five 2-D arrays are narrowed
    // (a kernel over 40 MB instead of 3.3 GB on the 1024 x 1024 x 128 grid); the flag word sits at wire[nwords] either way
    const size_t n = wire == c->atm_wire ? 5 * c->ncol : c->nwords;
    hipLaunchKernelGGL(rt_narrow_kernel, dim3(n > (1u << 22) ? 2048 : 256), dim3(256), 0, c->stream, tallies, wire, n,
                       wire + c->nwords, guard_shift);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int allreduce_impl(rt_context* const ctx[], int n)
{
    if (!ctx || n < 1) return fail(RT_ERR_ARG, "rt_allreduce: needs at least one context");
    std::vector<int> devices(n);
    for (int g = 0; g < n; ++g)
    {
        if (!ctx[g]) return fail(RT_ERR_ARG, "rt_allreduce: context %d is null", g);
        if (ctx[g]->nwords != ctx[0]->nwords) return fail(RT_ERR_ARG, "rt_allreduce: context %d has another grid", g);
        devices[g] = ctx[g]->device;
        for (int h = 0; h < g; ++h)
            if (devices[h] == devices[g])
                return fail(RT_ERR_ARG, "rt_allreduce: contexts %d and %d share GPU %d (one context per GPU)", h, g, devices[g]);
    }
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    int rc = load_rccl();
    if (rc != RT_OK) return rc;
    std::vector<void*>* comms = nullptr;
    rc = comms_for(devices, &comms);
    if (rc != RT_OK) return rc;
    const size_t nwords = ctx[0]->nwords;
    bool narrow = true;
    for (int g = 0; g < n; ++g) narrow = narrow && ctx[g]->wire32;
    if (narrow)
    {
        for (int g = 0; g < n; ++g)
        {
            rt_context* c = ctx[g];
            RT_HIP(hipSetDevice(c->device));
            if (!c->d_wire) RT_HIP(hipMalloc(&c->d_wire, sizeof(uint32_t) * (nwords + 1)));
            rc = launch_narrow(c, c->d_tallies, c->d_wire, n);
            if (rc != RT_OK) return rc;
        }
        RT_NCCL(g_rccl.GroupStart());
        for (int g = 0; g < n; ++g)
        {
            const int r = g_rccl.AllReduce(ctx[g]->d_wire, ctx[g]->d_wire, nwords + 1, kNcclUint32, kNcclSum, (*comms)[g], ctx[g]->stream);
            if (r != 0)
            {
                (void)g_rccl.GroupEnd();        // never leave RCCL inside an open group
                return fail(RT_ERR_HIP, "ncclAllReduce (32-bit counts, GPU %d) failed: %s", ctx[g]->device, g_rccl.GetErrorString(r));
            }
        }
        RT_NCCL(g_rccl.GroupEnd());
        uint32_t flag = 0;
        for (int g = 0; g < n; ++g)
        {
            rt_context* c = ctx[g];
            RT_HIP(hipSetDevice(c->device));
            hipLaunchKernelGGL(rt_widen_kernel, dim3(2048), dim3(256), 0, c->stream, c->d_tallies, c->d_wire, nwords, nwords);
            RT_HIP(hipGetLastError());
        }
        for (int g = 0; g < n; ++g)
        {
            rt_context* c = ctx[g];
            RT_HIP(hipSetDevice(c->device));
            RT_HIP(hipStreamSynchronize(c->stream));
            if (g == 0) RT_HIP(hipMemcpy(&flag, c->d_wire + nwords, sizeof(flag), hipMemcpyDeviceToHost));
        }
        if (flag == 0) return RT_OK;
        // some count was too large for 32 bits on the wire: the 64-bit tallies are untouched, sum those
    }
    RT_NCCL(g_rccl.GroupStart());
    for (int g = 0; g < n; ++g)
    {
        const int r = g_rccl.AllReduce(ctx[g]->d_tallies, ctx[g]->d_tallies, nwords, kNcclUint64, kNcclSum, (*comms)[g], ctx[g]->stream);
        if (r != 0)
        {
            (void)g_rccl.GroupEnd();
            return fail(RT_ERR_HIP, "ncclAllReduce (64-bit counts, GPU %d) failed: %s", ctx[g]->device, g_rccl.GetErrorString(r));
        }
    }
    RT_NCCL(g_rccl.GroupEnd());
    for (int g = 0; g < n; ++g)
    {
        RT_HIP(hipSetDevice(ctx[g]->device));
        RT_HIP(hipStreamSynchronize(ctx[g]->stream));
    }
    return RT_OK;
}

// Philox4x32-10 on the host (the device's rt::philox4x32_10, word for word) for the few values a launch needs
// ahead of its kernels.
void philox_host(const uint32_t ctr[4], uint32_t k0, uint32_t k1, uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    for (int round = 0; round < 10; ++round)
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Digital shift of the Sobol launch for the 2^32 photon IDs of `block` (= ID >> 32): the scramble constants of
// raytracer_3d.cu:141-145, here a Philox draw keyed on the seed and the block (counter words 2 and 3 are values
// no photon's stream uses: event index 2^32 - 1, tag 2), so every block and every seed is an independent
// randomisation of the sequence. The CPU checker draws the same block.
void sobol_shift(uint64_t seed, uint64_t block, uint32_t& sx, uint32_t& sy)
{
    const uint32_t ctr[4] = {(uint32_t)block, (uint32_t)(block >> 32), 0xFFFFFFFFu, 2u};
    uint32_t w[4];
    philox_host(ctr, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    sx = w[0]; sy = w[1];
}

// Sobol launch in box order (sobol_position_to_id): the partition of the domain top into 2^a x 2^b boxes -- about one
// box per column, at most 2^15 per axis -- whose nets of 2^(a+b) IDs the launch walks.
void sobol_boxes(const rt_context* c, uint32_t& a, uint32_t& b)
{
    const auto ceil_log2 = [](int n) { uint32_t l = 0; while ((1 << l) < n && l < 15u) ++l; return l < 1u ? 1u : l; };
    a = ceil_log2(c->itot); b = ceil_log2(c->jtot);
}
uint32_t sobol_net_log2(const rt_context* c) { uint32_t a, b; sobol_boxes(c, a, b); return a + b; }

// The cold block's share of a launch in box order over IDs [b0, b0 + nb): whole nets inside one block of 2^32 IDs,
// nb <= 2^30. The b x b block T of the Pascal matrix that ties ID bits a .. a+b-1 to the top b bits of dimension 1
// (T[q][k] = 1 iff the binary digits of a + k contain those of q) is inverted over GF(2) here; sob_tinv[r] is the
// column of the inverse that belongs to bit r of the residual (bit b-1-q of the box index <-> output bit 31-q).
void build_sobol_order(const rt_context* c, uint64_t b0, uint64_t nb, ColdBlock& cb)
{
    uint32_t a, b;
    sobol_boxes(c, a, b);
    const uint32_t m = a + b;
    uint32_t rows[16], inv[16];
    for (uint32_t q = 0; q < b; ++q)
    {
        rows[q] = 0u; inv[q] = 1u << q;
        for (uint32_t k = 0; k < b; ++k)
            if (((a + k) & q) == q) rows[q] |= 1u << k;
    }
    for (uint32_t col = 0; col < b; ++col)          // Gauss-Jordan; T is regular because the sequence is a (0,2)-sequence
    {
        uint32_t piv = col;
        while (piv < b && !((rows[piv] >> col) & 1u)) ++piv;
        if (piv == b) { cb.sob_a = 0u; return; }    // (cannot happen; the launch then takes the ordinary path's order)
        std::swap(rows[col], rows[piv]); std::swap(inv[col], inv[piv]);
        for (uint32_t r = 0; r < b; ++r)
            if (r != col && ((rows[r] >> col) & 1u)) { rows[r] ^= rows[col]; inv[r] ^= inv[col]; }
    }
    // inv[k] bit q = (T^-1)[k][q]: unknown bit k = XOR over q of inv[k][q] * residual[q]
    for (uint32_t r = 0; r < 16; ++r)
    {
        cb.sob_tinv[r] = 0u;
        if (r < b)
            for (uint32_t k = 0; k < b; ++k)
                cb.sob_tinv[r] |= ((inv[k] >> (b - 1u - r)) & 1u) << k;
    }
    // tiles of boxes and nets per visit as the stratified order chooses them (choose_launch_order): beyond the caches
    // tiles of 32 x 32 and up to 2^8 nets at a time, else rows and up to 2^5 (more photons into one column at a time
    // contend for the same tally words on small grids)
    const bool beyond = !small_grid(c);
    const uint32_t npass = (uint32_t)(nb >> m);
    uint32_t tile = beyond ? 5u : 0u, kmax = beyond ? 8u : 5u, k = 0u;
    if (c->launch_tile_log2 > 0) tile = (uint32_t)c->launch_tile_log2;
    if (tile > a) tile = a;
    if (tile > b) tile = b;
    while (k < kmax && ((npass >> (k + 1u)) << (k + 1u)) == npass) ++k;
    cb.sob_a = a; cb.sob_b = b; cb.sob_tile_log2 = tile; cb.sob_klog2 = k;
    cb.sob_b0 = (uint32_t)b0 >> m;
}

// Launches the tracer for photon IDs [pid_begin, pid_begin + count). `cursor` / `cold`: the device words the pool
// kernel takes its photon-ID cursors and cold block from -- the context's own allocation (rt_trace) or the static
// scratch (rt_trace_noalloc, which must not allocate: it refuses instead of building majorants).
int trace_impl(rt_context* c, uint64_t seed, uint64_t pid_begin, uint64_t count, bool noalloc)
{
    const char* const who = noalloc ? "rt_trace_noalloc" : "rt_trace";
    if (!c) return fail(RT_ERR_ARG, "%s: null context", who);
    if (!c->have_optics || !c->have_scene)
        return fail(RT_ERR_STATE, "%s: call rt_upload_optics and rt_set_scene first", who);
    if (pid_begin + count < pid_begin) return fail(RT_ERR_ARG, "%s: photon id range overflows 64 bits", who);
    // every refusal comes before anything is enqueued or any state of the context changes
    if (c->planar && c->pool == 0)
        return fail(RT_ERR_ARG, "%s: option planar needs the pool kernel (pool > 0)", who);
    if (c->pool > 0 && count > 0 && pid_begin + count > (1ull << 40))
        return fail(RT_ERR_ARG, "%s: the pool kernel packs photon ids into 40 bits (ids up to 2^40; option pool = 0 "
                                "carries 64-bit ids)", who);
    if (c->launch_log2k >= 0 && c->launch_tile_log2 > 0 &&
        ((c->itot & ((1 << c->launch_tile_log2) - 1)) != 0 || (c->jtot & ((1 << c->launch_tile_log2) - 1)) != 0))
        return fail(RT_ERR_ARG, "%s: option launch_tile_log2 = %d needs itot and jtot to be multiples of %d (grid %d x %d)", who,
                    c->launch_tile_log2, 1 << c->launch_tile_log2, c->itot, c->jtot);
    if (c->atm_wire)
    {
        if (c->pool == 0)
            return fail(RT_ERR_ARG, "%s: a bound absorption wire (rt_bind_atmos_wire) needs the pool kernel (pool > 0)", who);
        if (count > 0xffffffffull - c->atm_photons)
            return fail(RT_ERR_ARG, "%s: %llu photons on top of the %llu already counted into the bound 32-bit absorption wire "
                                    "could overflow a word (reset the tallies, or unbind the wire)", who,
                        (unsigned long long)count, (unsigned long long)c->atm_photons);
    }
    if (noalloc)
    {
        if (c->pool == 0)
            return fail(RT_ERR_ARG, "rt_trace_noalloc: needs the pool kernel (option pool > 0); the lane-resident "
                                    "kernel has no scratch to place");
        if (c->maj_dirty)
            return fail(RT_ERR_STATE, "rt_trace_noalloc: the majorant tables are not built (they are allocated by "
                                      "rt_prepare / rt_get_slabs / rt_trace): call rt_prepare after the last "
                                      "rt_upload_optics or option change");
    }
    RT_HIP(hipSetDevice(c->device));
    if (count == 0) { c->timed = false; return RT_OK; }
    if (c->atm_wire) c->atm_photons += count;
    if (c->maj_dirty)
    {
        const int rc = build_majorants(c);
        if (rc != RT_OK) return rc;
    }

    const rt_slab_info& sl = c->slabs;
    const float xs = (float)c->itot * c->dx, ys = (float)c->jtot * c->dy;
    const auto rcp = [](float u) { return std::fabs(u) >= 0x1p-100f ? 1.0f / u : std::copysign(FLT_MAX, u); };   // dir_reciprocal
    // stratified launch: the divisor of launch_column (itot, or the tiles per row of a tiled order) and its magic
    // number, exact for c < 2^31 (Granlund-Montgomery, N = 31)
    const uint32_t tile_log2 = c->launch_log2k >= 0 ? (uint32_t)c->launch_tile_log2 : 0u;
    const uint32_t launch_div = (uint32_t)c->itot >> tile_log2;
    uint32_t div_shift, div_mul;
    {
        int l = 0;
        while ((1u << l) < launch_div) ++l;
        div_shift = 31u + (uint32_t)l;
        div_mul = (uint32_t)(((1ull << div_shift) / (uint64_t)launch_div) + 1ull);
    }
    const bool gridded = sl.gslab >= 0;
    const bool sobol = c->launch_log2k == kLaunchSobol;

    // Launch sort (rt_sort_count_kernel): uniform launch positions, pool kernel; by default from 2^19 photons per call
    // (below that the five small kernels cost what the order gains). The IDs go through in batches of
    // 2^sort_batch_log2, each sorted and then traced. rt_trace_noalloc sorts only into a list that an earlier
    // rt_trace left behind.
    bool sorted = c->pool > 0 && c->launch_log2k < 0 &&
                  (c->launch_sort == 1 || (c->launch_sort < 0 && count >= (1ull << 19)));
    const uint64_t batch = sorted ? (1ull << c->sort_batch_log2) : count;
    // (the list is allocated by the first batch that is sorted: a Sobol launch whose IDs go in box order never needs it)

    if (!noalloc) RT_HIP(hipEventRecord(c->ev_start, c->stream));
    if (c->pool > 0)
    {
        // pool kernel: one wave per workgroup; LDS bounds the resident waves (160 KiB per CU)
        const int ns = c->pool;
        const size_t lds_bytes = ((size_t)ns * kPoolRec + 16) * 4;
        int per_cu = (int)((160u * 1024u) / lds_bytes);
        if (per_cu > c->pool_waves_per_cu) per_cu = c->pool_waves_per_cu;
        if (per_cu < 1) per_cu = 1;
        const uint64_t pcap = (uint64_t)c->cu_count * per_cu;

        unsigned long long* cursor = c->d_events + kCursorWord;
        ColdBlock* d_cold = c->d_cold;
        if (noalloc)
        {
            unsigned long long* sp = nullptr;
            const int rc = scratch_of_device(c->device, &sp);
            if (rc != RT_OK) return rc;
            cursor = sp;
            d_cold = reinterpret_cast<ColdBlock*>(sp + kScratchColdWord);
        }
        ColdBlock cb;
        memset(&cb, 0, sizeof(cb));
        cb.events = c->d_events;
        cb.launch_log2k = c->launch_log2k;
        cb.sun_x = sl.sun[0]; cb.sun_y = sl.sun[1]; cb.sun_z = sl.sun[2];
        cb.inv_sun_x = rcp(cb.sun_x); cb.inv_sun_y = rcp(cb.sun_y); cb.inv_sun_z = rcp(cb.sun_z);
        cb.ztop = sl.zface[sl.nslab]; cb.dx = c->dx; cb.dy = c->dy; cb.albedo = c->albedo;
        cb.div_itot_mul = div_mul; cb.div_itot_shift = div_shift;
        cb.launch_div = launch_div; cb.launch_tile_log2 = tile_log2;
        for (int i = 0; i < 4; ++i) cb.zface[i] = sl.zface[i];
        cb.inv_xs = 1.0f / xs; cb.inv_ys = 1.0f / ys;
        cb.ncoarse = (uint32_t)c->ncoarse; cb.nslab = sl.nslab;
        cb.atm32 = c->atm_wire;

        PoolArgs a;
        a.cells = c->d_cells; a.maj = c->d_maj; a.tallies = c->d_tallies;
        a.cold = d_cold; a.cursor = cursor;
        a.seed = seed;
        for (int r = 0; r < 10; ++r)
        {
            a.round_keys[2 * r] = (uint32_t)seed + (uint32_t)r * 0x9E3779B9u;
            a.round_keys[2 * r + 1] = (uint32_t)(seed >> 32) + (uint32_t)r * 0xBB67AE85u;
        }
        a.itot = c->itot; a.jtot = c->jtot; a.ktot = c->ktot; a.ncol = (uint32_t)c->ncol;
        a.nbi = c->nbi; a.nbij = c->nbij;
        a.gslab = sl.gslab; a.nci = sl.nci; a.ncj = sl.ncj; a.nck = sl.nck;
        a.th_launch = c->th_launch;
        a.tally_match = c->tally_match >= 0 ? c->tally_match : (c->ncol <= kTallyMatchCols ? kTallyMatchAuto : 0);
        a.xs = xs; a.ys = ys;
        a.inv_dx = 1.0f / c->dx; a.inv_dy = 1.0f / c->dy; a.inv_dz = 1.0f / c->dz;
        a.hx = sl.hx; a.hy = sl.hy; a.hz = sl.hz;
        a.inv_hx = gridded ? 1.0f / a.hx : 0.0f; a.inv_hy = gridded ? 1.0f / a.hy : 0.0f;
        a.inv_hz = gridded ? 1.0f / a.hz : 0.0f;
        a.zc0 = gridded ? sl.zface[sl.gslab] : 0.0f; a.zc1 = gridded ? sl.zface[sl.gslab + 1] : 0.0f;

        SortArgs so;
        memset(&so, 0, sizeof(so));
        if (sorted)
        {
            so.seed = seed;             // (list and tables: per batch, once the list exists)
            so.tile_shift = (uint32_t)c->sort_tile_log2;
            const auto tiles_along = [&](int n) { return (uint32_t)((n + (1 << so.tile_shift) - 1) >> so.tile_shift); };
            while ((uint64_t)tiles_along(c->itot) * tiles_along(c->jtot) > kSortMaxTiles) ++so.tile_shift;
            so.ntx = tiles_along(c->itot); so.ntiles = so.ntx * tiles_along(c->jtot);
            so.itot = c->itot; so.jtot = c->jtot;
            so.xs = xs; so.ys = ys; so.inv_dx = a.inv_dx; so.inv_dy = a.inv_dy;
            so.toa_down = c->d_tallies;
            so.nslices = (uint32_t)((sort_blocks(c) + so.ntiles - 1) / so.ntiles);
        }
        const bool toa_pass = sorted && so.tile_shift <= 6u;      // the tile's columns fit the pass's LDS histogram
        // Sobol launch: a launch stays inside one block of 2^32 IDs (the sequence's index is the low word of the
        // ID, its digital shift belongs to the block), so a range that crosses a multiple of 2^32 is cut there
        for (uint64_t done = 0, nb = 0; done < count; done += nb)
        {
            const uint64_t b0 = pid_begin + done;
            nb = count - done < batch ? count - done : batch;
            bool in_boxes = false;          // Sobol launch in box order: needs neither a sort nor a list
            cb.sob_a = 0u;
            if (sobol)
            {
                const uint64_t to_block_end = (((b0 >> 32) + 1ull) << 32) - b0;
                uint64_t span = count - done < to_block_end ? count - done : to_block_end;
                sobol_shift(seed, b0 >> 32, cb.sob_x, cb.sob_y);
                so.sobol = 1; so.sob_x = cb.sob_x; so.sob_y = cb.sob_y;
                if (c->sobol_boxes)
                {
                    // the stretch of whole nets (2^m consecutive IDs from a multiple of 2^m) goes in box order; what lies
                    // ahead of the first net, and what is left behind the last, takes the ordinary path
                    const uint64_t net = 1ull << sobol_net_log2(c), head = (net - (b0 & (net - 1ull))) & (net - 1ull);
                    const uint64_t whole = (span < (1ull << 30) ? span : (1ull << 30)) & ~(net - 1ull);
                    if (head != 0ull) span = head < span ? head : span;
                    else if (whole != 0ull) { in_boxes = true; nb = whole; build_sobol_order(c, b0, nb, cb); }
                }
                if (!in_boxes) nb = span < batch ? span : batch;
            }
            // (what a Sobol launch leaves outside its whole nets is short: sorted only when that is worth five kernels)
            bool sorted_now = sorted && !in_boxes &&
                              !(sobol && c->sobol_boxes && c->launch_sort < 0 && nb < (1ull << 19));
            if (sorted_now && (size_t)nb > c->sort_capacity)
            {
                if (noalloc)
                    sorted_now = false;
                else
                {
                    // room for the largest batch this call can still bring
                    const size_t need = (size_t)(count - done < batch ? count - done : batch);
                    if (c->d_sort) RT_HIP(hipFree(c->d_sort));
                    c->d_sort = nullptr; c->sort_capacity = 0;
                    RT_HIP(hipMalloc(&c->d_sort, sizeof(uint32_t) * (need + kSortMaxTiles * (1 + sort_blocks(c)))));
                    c->sort_capacity = need;
                }
            }
            if (sorted_now)
            {
                so.list = c->d_sort; so.tile_base = c->d_sort + c->sort_capacity; so.count = so.tile_base + kSortMaxTiles;
            }
            // IDs go out in chunks from a device-wide cursor; small jobs get smaller chunks so that
            // every resident wave still finds work
            uint32_t chunk = (uint32_t)c->chunk;
            while (chunk > 64u && nb / chunk < 4u * pcap) chunk >>= 1;
            cb.pid_begin = b0; cb.total = nb;
            cb.sub_size = (((nb + kXcds - 1) / kXcds + chunk - 1) / chunk) * chunk;
            cb.chunk = chunk;
            cb.sorted = sorted_now ? c->d_sort : nullptr;
            cb.toa_done = (toa_pass && sorted_now) ? 1u : 0u;
            if (sorted_now)
            {
                so.pid_begin = b0; so.n = (uint32_t)nb;
                const uint64_t max_blocks = sort_blocks(c);
                const uint64_t want = (nb + kSortBlock - 1) / kSortBlock;
                so.nblocks = (uint32_t)(want < max_blocks ? want : max_blocks);
                so.per_block = (uint32_t)((nb + so.nblocks - 1) / so.nblocks);
                so.nblocks = (uint32_t)((nb + so.per_block - 1) / so.per_block);
                hipLaunchKernelGGL(rt_sort_count_kernel, dim3(so.nblocks), dim3(kSortBlock), 0, c->stream, so);
                hipLaunchKernelGGL(rt_sort_offsets_kernel, dim3(so.ntiles), dim3(64), 0, c->stream, so);
                hipLaunchKernelGGL(rt_sort_scan_kernel, dim3(1), dim3(kSortBlock), 0, c->stream, so.tile_base, so.ntiles);
                if (so.ntiles <= 1024u)
                    hipLaunchKernelGGL(rt_sort_scatter_kernel<1024u>, dim3(so.nblocks), dim3(kSortBlock), 0, c->stream, so);
                else
                    hipLaunchKernelGGL(rt_sort_scatter_kernel<kSortMaxTiles>, dim3(so.nblocks), dim3(kSortBlock), 0, c->stream, so);
                if (toa_pass)
                    hipLaunchKernelGGL(rt_sort_toa_kernel, dim3(so.ntiles * so.nslices), dim3(kSortBlock), 0, c->stream, so);
            }
            // the cold block and the rewound cursors, in stream order ahead of the tracer
            hipLaunchKernelGGL(rt_launch_block_kernel, dim3(1), dim3(64), 0, c->stream, d_cold, cb, cursor);
            uint64_t pblocks = (nb + chunk - 1) / chunk;
            if (pblocks > pcap) pblocks = pcap;
            const dim3 grid((unsigned)pblocks), pblock(kPoolBlock);
            const int phys = c->planar ? PHYS_PLANAR : (c->fixed_dot ? PHYS_FIXED_DOT : PHYS_COMPAT);
            // NS = 128 is the tuned default and comes with and without the event counters; the other pool sizes exist
            // for experiments and tests and always count
#define RT_LAUNCH_POOL_PS(P, S, NS, H, M) hipLaunchKernelGGL((rt_trace_pool_kernel<P, S, NS, H, M>), grid, pblock, 0, c->stream, a)
#define RT_LAUNCH_POOL_P(P, NS, H)                                                                             \
            do {                                                                                                   \
                if (NS != RT_FAST_POOL || c->stats) RT_LAUNCH_POOL_PS(P, true, NS, H, true);                       \
                else if (a.tally_match != 0) RT_LAUNCH_POOL_PS(P, (NS != RT_FAST_POOL), NS, H, true);              \
                else RT_LAUNCH_POOL_PS(P, (NS != RT_FAST_POOL), NS, H, (NS != RT_FAST_POOL));                      \
            } while (0)
#define RT_LAUNCH_POOL_H(NS, H)                                                                                \
            do {                                                                                                   \
                if (phys == PHYS_PLANAR) RT_LAUNCH_POOL_P(PHYS_PLANAR, NS, H);                                     \
                else if (phys == PHYS_FIXED_DOT) RT_LAUNCH_POOL_P(PHYS_FIXED_DOT, NS, H);                          \
                else RT_LAUNCH_POOL_P(PHYS_COMPAT, NS, H);                                                         \
            } while (0)
#define RT_LAUNCH_POOL(NS)                                                                                     \
            do {                                                                                                   \
                if (c->hier) RT_LAUNCH_POOL_H(NS, true); else RT_LAUNCH_POOL_H(NS, false);                         \
            } while (0)
            switch (ns)
            {
                case RT_FAST_POOL: RT_LAUNCH_POOL(RT_FAST_POOL); break;
#ifndef RT_ONLY_DEFAULT_POOL
#if RT_FAST_POOL != 64
                case 64: RT_LAUNCH_POOL(64); break;
#endif
#if RT_FAST_POOL != 96
                case 96: RT_LAUNCH_POOL(96); break;
#endif
#if RT_FAST_POOL != 128
                case 128: RT_LAUNCH_POOL(128); break;
#endif
                case 160: RT_LAUNCH_POOL(160); break;
                case 192: RT_LAUNCH_POOL(192); break;
                case 256: RT_LAUNCH_POOL(256); break;
#endif
                default: return fail(RT_ERR_ARG, "%s: pool size %d is not in this build", who, ns);
            }
        }   // batches
#undef RT_LAUNCH_POOL
#undef RT_LAUNCH_POOL_H
#undef RT_LAUNCH_POOL_P
#undef RT_LAUNCH_POOL_PS
    }
    else
    {
        TraceArgs a;
        a.cells = c->d_cells;
        a.tallies = c->d_tallies; a.events = c->d_events;
        a.seed = seed; a.pid_begin = pid_begin; a.pid_end = pid_begin + count;
        a.itot = c->itot; a.jtot = c->jtot; a.ktot = c->ktot;
        a.ncol = (uint32_t)c->ncol; a.ncell = (uint32_t)c->ncell;
        a.nbi = c->nbi; a.nbij = c->nbij;
        a.xs = xs; a.ys = ys;
        a.inv_xs = 1.0f / a.xs; a.inv_ys = 1.0f / a.ys;
        a.inv_dx = 1.0f / c->dx; a.inv_dy = 1.0f / c->dy; a.inv_dz = 1.0f / c->dz;
        a.albedo = c->albedo;
        a.sun_x = sl.sun[0]; a.sun_y = sl.sun[1]; a.sun_z = sl.sun[2];
        a.inv_sun_x = rcp(a.sun_x); a.inv_sun_y = rcp(a.sun_y); a.inv_sun_z = rcp(a.sun_z);
        a.nslab = sl.nslab;
        for (int i = 0; i < 4; ++i) a.zface[i] = sl.zface[i];
        a.maj = c->d_maj;
        a.gslab = sl.gslab;
        a.nci = sl.nci; a.ncj = sl.ncj; a.nck = sl.nck;
        a.ncoarse = (uint32_t)c->ncoarse;
        a.hx = sl.hx; a.hy = sl.hy; a.hz = sl.hz;
        a.inv_hx = gridded ? 1.0f / a.hx : 0.0f; a.inv_hy = gridded ? 1.0f / a.hy : 0.0f;
        a.inv_hz = gridded ? 1.0f / a.hz : 0.0f;
        a.zc0 = gridded ? a.zface[a.gslab] : 0.0f; a.zc1 = gridded ? a.zface[a.gslab + 1] : 0.0f;
        a.th_event = c->th_event; a.th_scatter = c->th_scatter; a.low_traverse = c->low_traverse;
        a.launch_log2k = c->launch_log2k;
        a.dx = c->dx; a.dy = c->dy;
        a.div_itot_mul = div_mul; a.div_itot_shift = div_shift;
        a.launch_div = launch_div; a.launch_tile_log2 = tile_log2;

        a.sob_x = 0u; a.sob_y = 0u;

        // (Sobol launch: one launch per block of 2^32 IDs, as in the pool path)
        for (uint64_t p0 = pid_begin, p1 = 0; p0 < pid_begin + count; p0 = p1)
        {
            p1 = pid_begin + count;
            if (sobol)
            {
                const uint64_t block_end = ((p0 >> 32) + 1ull) << 32;
                if (block_end != 0ull && block_end < p1) p1 = block_end;
                sobol_shift(seed, p0 >> 32, a.sob_x, a.sob_y);
            }
            a.pid_begin = p0; a.pid_end = p1;
            uint64_t blocks = (p1 - p0 + kBlock - 1) / kBlock;
            const uint64_t cap = (uint64_t)c->cu_count * c->blocks_per_cu;
            if (blocks > cap) blocks = cap;
            const dim3 grid((unsigned)blocks), block(kBlock);
            if (c->fixed_dot)
            {
                if (c->stats) hipLaunchKernelGGL((rt_trace_kernel<true, true>), grid, block, 0, c->stream, a);
                else hipLaunchKernelGGL((rt_trace_kernel<true, false>), grid, block, 0, c->stream, a);
            }
            else
            {
                if (c->stats) hipLaunchKernelGGL((rt_trace_kernel<false, true>), grid, block, 0, c->stream, a);
                else hipLaunchKernelGGL((rt_trace_kernel<false, false>), grid, block, 0, c->stream, a);
            }
        }
    }
    RT_HIP(hipGetLastError());
    if (!noalloc)
    {
        RT_HIP(hipEventRecord(c->ev_stop, c->stream));
        c->timed = true;
    }
    else
        c->timed = false;
    return RT_OK;
}
} // namespace

extern "C" {

const char* rt_last_error(void) { return g_last_error.c_str(); }

int rt_device_count(int* count)
{
    if (!count) return fail(RT_ERR_ARG, "rt_device_count: null pointer");
    RT_HIP(hipGetDeviceCount(count));
    return RT_OK;
}

int rt_create(int device_id, int itot, int jtot, int ktot, float dx, float dy, float dz, rt_context** out)
{
    if (!out) return fail(RT_ERR_ARG, "rt_create: null output pointer");
    *out = nullptr;
    if (itot <= 0 || jtot <= 0 || ktot <= 0 || !(dx > 0) || !(dy > 0) || !(dz > 0))
        return fail(RT_ERR_ARG, "rt_create: grid %dx%dx%d with spacing %g,%g,%g is not valid", itot, jtot, ktot, dx, dy, dz);
    const size_t ncol = (size_t)itot * jtot, ncell = ncol * ktot;
    if (5 * ncol + 2 * ncell >= (1ull << 32))
        return fail(RT_ERR_ARG, "rt_create: %zu cells exceed the 32-bit tally index of this build", ncell);
    int ndev = 0;
    RT_HIP(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev)
        return fail(RT_ERR_ARG, "rt_create: device %d not in [0,%d)", device_id, ndev);
    RT_HIP(hipSetDevice(device_id));

    rt_context* c = new rt_context;
    c->device = device_id;
    c->itot = itot; c->jtot = jtot; c->ktot = ktot;
    c->dx = dx; c->dy = dy; c->dz = dz;
    c->ncol = ncol; c->ncell = ncell; c->nwords = 5 * ncol + 2 * ncell;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
        c->cu_count = prop.multiProcessorCount;

    hipError_t e = hipSuccess;
#if RT_BRICK == 2
    c->nbi = (uint32_t)(itot + 1) / 2u; c->nbij = c->nbi * ((uint32_t)(jtot + 1) / 2u);
    c->ncell_packed = (size_t)c->nbij * ((size_t)(ktot + 1) / 2) * 8;
#elif RT_BRICK == 1
    c->nbi = (uint32_t)(itot + 1) / 2u; c->nbij = c->nbi * ((uint32_t)(jtot + 1) / 2u);
    c->ncell_packed = (size_t)c->nbij * (size_t)ktot * 4;
#else
    c->nbi = (uint32_t)itot; c->nbij = (uint32_t)ncol;
    c->ncell_packed = ncell;
#endif
    if (c->ncell_packed >= (1ull << 32))
    {
        const size_t packed = c->ncell_packed;
        delete c;
        return fail(RT_ERR_ARG, "rt_create: %zu packed cells exceed the 32-bit cell index of this build", packed);
    }
    if (e == hipSuccess) e = hipMalloc(&c->d_cells, sizeof(float4) * c->ncell_packed);
    if (e == hipSuccess) e = hipMemset(c->d_cells, 0, sizeof(float4) * c->ncell_packed);
    if (e == hipSuccess) e = hipMalloc(&c->d_tallies, sizeof(unsigned long long) * c->nwords);
    if (e == hipSuccess) e = hipMalloc(&c->d_events, sizeof(unsigned long long) * kEventWords);
    if (e == hipSuccess) e = hipMalloc(&c->d_cold, sizeof(ColdBlock));
    if (e == hipSuccess) e = hipMemset(c->d_tallies, 0, sizeof(unsigned long long) * c->nwords);
    if (e == hipSuccess) e = hipMemset(c->d_events, 0, sizeof(unsigned long long) * kEventWords);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_stop);
    if (e != hipSuccess)
    {
        c->owns_tallies = true;
        rt_destroy(c);
        return fail(RT_ERR_HIP, "rt_create: allocation failed: %s", hipGetErrorString(e));
    }
    c->owns_tallies = true;
    {
        // set-up of the static scratch of rt_trace_noalloc: its address on this device, once per device
        unsigned long long* sp = nullptr;
        const int rc = scratch_of_device(device_id, &sp);
        if (rc != RT_OK) { rt_destroy(c); return rc; }
    }
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_context* c)
{
    if (!c) return RT_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream); else (void)hipDeviceSynchronize();
    if (c->d_cells) (void)hipFree(c->d_cells);
    if (c->d_maj) (void)hipFree(c->d_maj);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->d_tallies && c->owns_tallies) (void)hipFree(c->d_tallies);
    if (c->d_tallies_alt) (void)hipFree(c->d_tallies_alt);
    if (c->dl_ready) (void)hipEventDestroy(c->dl_ready);
    for (int t = 0; t < 2; ++t) if (c->dl_done[t]) (void)hipEventDestroy(c->dl_done[t]);
    if (c->d_events) (void)hipFree(c->d_events);
    if (c->d_cold) (void)hipFree(c->d_cold);
    if (c->d_sort) (void)hipFree(c->d_sort);
    if (c->d_wire) (void)hipFree(c->d_wire);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    delete c;
    return RT_OK;
}

int rt_upload_optics(rt_context* c, const float* k_gas, const float* k_cloud, const float* ssa, const float* asy)
{
    if (!c || !k_gas || !k_cloud || !ssa || !asy) return fail(RT_ERR_ARG, "rt_upload_optics: null pointer");
    RT_HIP(hipSetDevice(c->device));
    const size_t bytes = sizeof(float) * c->ncell;
    float* tmp = nullptr;   // gas | cloud | asy | ssa staging for the pack kernel
    RT_HIP(hipMalloc(&tmp, 4 * bytes));
    hipError_t e = hipMemcpyAsync(tmp, k_gas, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp + c->ncell, k_cloud, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp + 2 * c->ncell, asy, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp + 3 * c->ncell, ssa, bytes, hipMemcpyHostToDevice, c->stream);
    int rc = RT_OK;
    if (e != hipSuccess) rc = fail(RT_ERR_HIP, "rt_upload_optics: copy failed: %s", hipGetErrorString(e));
    else rc = pack_from_device(c, tmp, tmp + c->ncell, tmp + 3 * c->ncell, tmp + 2 * c->ncell);
    (void)hipFree(tmp);
    return rc;
}

int rt_upload_optics_multi(rt_context* const ctx[], int n, const float* k_gas, const float* k_cloud, const float* ssa, const float* asy)
{
    if (!ctx || n < 1 || !k_gas || !k_cloud || !ssa || !asy) return fail(RT_ERR_ARG, "rt_upload_optics_multi: bad argument");
    for (int g = 0; g < n; ++g)
        if (!ctx[g] || ctx[g]->ncell != ctx[0]->ncell) return fail(RT_ERR_ARG, "rt_upload_optics_multi: context %d is null or has another grid", g);
    const size_t ncell = ctx[0]->ncell, bytes = sizeof(float) * ncell;
    const float* const src[4] = {k_gas, k_cloud, asy, ssa};       // staging order of the pack kernel: gas | cloud | asy | ssa
    // page-lock the caller's arrays for the duration of the call, so that the copies to all GPUs are true asynchronous
    // DMA and run side by side (each GPU has its own link to the host); arrays that cannot be locked are copied pageable
    bool locked[4] = {false, false, false, false};
    for (int f = 0; f < 4; ++f)
        locked[f] = hipHostRegister(const_cast<float*>(src[f]), bytes, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
    std::vector<float*> tmp(n, nullptr);
    int rc = RT_OK;
    for (int g = 0; g < n && rc == RT_OK; ++g)
    {
        rt_context* c = ctx[g];
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = hipMalloc(&tmp[g], 4 * bytes);
        for (int f = 0; f < 4 && e == hipSuccess; ++f)
            e = hipMemcpyAsync(tmp[g] + (size_t)f * ncell, src[f], bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, "rt_upload_optics_multi: GPU %d: %s", c->device, hipGetErrorString(e));
    }
    for (int g = 0; g < n; ++g)
    {
        rt_context* c = ctx[g];
        (void)hipSetDevice(c->device);
        if (rc == RT_OK && tmp[g])
            rc = pack_from_device(c, tmp[g], tmp[g] + ncell, tmp[g] + 3 * ncell, tmp[g] + 2 * ncell);
        else
            (void)hipStreamSynchronize(c->stream);
        if (tmp[g]) (void)hipFree(tmp[g]);
    }
    for (int f = 0; f < 4; ++f)
        if (locked[f]) (void)hipHostUnregister(const_cast<float*>(src[f]));
    return rc;
}

int rt_upload_optics_device(rt_context* c, const float* d_gas, const float* d_cloud, const float* d_ssa, const float* d_asy)
{
    if (!c || !d_gas || !d_cloud || !d_ssa || !d_asy) return fail(RT_ERR_ARG, "rt_upload_optics_device: null pointer");
    RT_HIP(hipSetDevice(c->device));
    return pack_from_device(c, d_gas, d_cloud, d_ssa, d_asy);
}

int rt_set_scene(rt_context* c, float zenith_rad, float azimuth_rad, float albedo)
{
    if (!c) return fail(RT_ERR_ARG, "rt_set_scene: null context");
    if (!(zenith_rad >= 0.0f) || !(zenith_rad < 1.5707963f))
        return fail(RT_ERR_ARG, "rt_set_scene: zenith %g rad outside [0, pi/2)", zenith_rad);
    if (!(albedo >= 0.0f) || !(albedo <= 1.0f))
        return fail(RT_ERR_ARG, "rt_set_scene: albedo %g outside [0,1]", albedo);
    c->zenith = zenith_rad; c->azimuth = azimuth_rad; c->albedo = albedo;
    // raytracer_3d.cu:28-30, float arithmetic
    c->slabs.sun[0] = -sinf(zenith_rad) * cosf(azimuth_rad);
    c->slabs.sun[1] = -sinf(zenith_rad) * sinf(azimuth_rad);
    c->slabs.sun[2] = -cosf(zenith_rad);
    c->have_scene = true;
    return RT_OK;
}

int rt_set_option(rt_context* c, const char* name, long long value)
{
    if (!c || !name) return fail(RT_ERR_ARG, "rt_set_option: null argument");
    if (!strcmp(name, "fixed_dot")) c->fixed_dot = value != 0;
    else if (!strcmp(name, "stats")) c->stats = value != 0;
    else if (!strcmp(name, "wire32")) c->wire32 = value != 0;
    else if (!strcmp(name, "planar")) c->planar = value != 0;
    else if (!strcmp(name, "blocks_per_cu"))
    {
        if (value < 1 || value > 16) return fail(RT_ERR_ARG, "rt_set_option: blocks_per_cu %lld outside [1,16]", value);
        c->blocks_per_cu = (int)value;
    }
    else if (!strcmp(name, "pool"))
    {
        if (value != 0 && value != RT_FAST_POOL && value != 64 && value != 96 && value != 128 && value != 160 && value != 192 && value != 256)
            return fail(RT_ERR_ARG, "rt_set_option: pool %lld not one of 0, 64, 96, 128, 160, 192, 256", value);
        c->pool = (int)value;
        if (wants_fine(c) != c->built_fine) c->maj_dirty = true;      // the automatic grid depends on it
    }
    else if (!strcmp(name, "launch_tile_log2"))
    {
        if (value < 0 || value > 8) return fail(RT_ERR_ARG, "rt_set_option: launch_tile_log2 %lld outside [0,8]", value);
        c->launch_tile_log2 = (int)value;
    }
    else if (!strcmp(name, "launch_sort"))
    {
        if (value < -1 || value > 1) return fail(RT_ERR_ARG, "rt_set_option: launch_sort %lld outside [-1,1]", value);
        c->launch_sort = (int)value;
        if (wants_fine(c) != c->built_fine) c->maj_dirty = true;
    }
    else if (!strcmp(name, "sobol_boxes")) c->sobol_boxes = value != 0;
    else if (!strcmp(name, "tally_match"))
    {
        if (value < -1 || value > 16) return fail(RT_ERR_ARG, "rt_set_option: tally_match %lld outside [-1,16]", value);
        c->tally_match = (int)value;
    }
    else if (!strcmp(name, "sort_tile_log2"))
    {
        if (value < 0 || value > 12) return fail(RT_ERR_ARG, "rt_set_option: sort_tile_log2 %lld outside [0,12]", value);
        c->sort_tile_log2 = (int)value;
    }
    else if (!strcmp(name, "sort_batch_log2"))
    {
        if (value < 16 || value > 31) return fail(RT_ERR_ARG, "rt_set_option: sort_batch_log2 %lld outside [16,31]", value);
        c->sort_batch_log2 = (int)value;
    }
    else if (!strcmp(name, "launch_log2k"))
    {
        if (value < -2 || value > 20) return fail(RT_ERR_ARG, "rt_set_option: launch_log2k %lld outside [-2,20]", value);
        c->launch_log2k = (int)value;
        if (wants_fine(c) != c->built_fine) c->maj_dirty = true;
    }
    else if (!strcmp(name, "launch_sobol"))
    {
        // by name: 1 = the Sobol launch (launch_log2k = -2), 0 = back to uniform positions if it was on
        if (value != 0 && value != 1) return fail(RT_ERR_ARG, "rt_set_option: launch_sobol %lld not 0 or 1", value);
        if (value == 1) c->launch_log2k = kLaunchSobol;
        else if (c->launch_log2k == kLaunchSobol) c->launch_log2k = kLaunchUniform;
        if (wants_fine(c) != c->built_fine) c->maj_dirty = true;
    }
    else if (!strcmp(name, "levels"))
    {
        if (value < -1 || value > 6) return fail(RT_ERR_ARG, "rt_set_option: levels %lld outside [-1,6]", value);
        c->levels = (int)value;
        c->maj_dirty = true;
    }
    else if (!strcmp(name, "merge_tau_milli"))
    {
        if (value < 0 || value > 1000000) return fail(RT_ERR_ARG, "rt_set_option: merge_tau_milli %lld outside [0,1000000]", value);
        c->merge_tau_milli = (int)value;
        c->maj_dirty = true;
    }
    else if (!strcmp(name, "chunk"))
    {
        if (value < 64 || value > (1 << 20) || (value & (value - 1)))
            return fail(RT_ERR_ARG, "rt_set_option: chunk %lld not a power of two in [64, 2^20]", value);
        c->chunk = (int)value;
    }
    else if (!strcmp(name, "th_launch"))
    {
        if (value < 1 || value > 256) return fail(RT_ERR_ARG, "rt_set_option: th_launch %lld outside [1,256]", value);
        c->th_launch = (int)value;
    }
    else if (!strcmp(name, "pool_waves_per_cu"))
    {
        if (value < 1 || value > 32) return fail(RT_ERR_ARG, "rt_set_option: pool_waves_per_cu %lld outside [1,32]", value);
        c->pool_waves_per_cu = (int)value;
    }
    else if (!strcmp(name, "supergrid_x") || !strcmp(name, "supergrid_y") || !strcmp(name, "supergrid_z"))
    {
        if (value < -1 || value > 4096) return fail(RT_ERR_ARG, "rt_set_option: %s %lld outside [-1,4096]", name, value);
        (name[10] == 'x' ? c->sgx : (name[10] == 'y' ? c->sgy : c->sgz)) = (int)value;
        c->maj_dirty = true;
    }
    else if (!strcmp(name, "th_event") || !strcmp(name, "th_scatter") || !strcmp(name, "low_traverse"))
    {
        if (value < 1 || value > 64) return fail(RT_ERR_ARG, "rt_set_option: %s %lld outside [1,64]", name, value);
        (!strcmp(name, "th_event") ? c->th_event : (!strcmp(name, "th_scatter") ? c->th_scatter : c->low_traverse)) = (int)value;
    }
    else return fail(RT_ERR_ARG, "rt_set_option: unknown option \"%s\"", name);
    return RT_OK;
}

int rt_set_stream(rt_context* c, void* hip_stream)
{
    if (!c) return fail(RT_ERR_ARG, "rt_set_stream: null context");
    c->stream = (hipStream_t)hip_stream;
    return RT_OK;
}

int rt_trace(rt_context* c, uint64_t seed, uint64_t pid_begin, uint64_t count)
{
    return trace_impl(c, seed, pid_begin, count, false);
}

int rt_trace_noalloc(rt_context* c, uint64_t seed, uint64_t pid_begin, uint64_t count)
{
    return trace_impl(c, seed, pid_begin, count, true);
}

int rt_prepare(rt_context* c)
{
    if (!c) return fail(RT_ERR_ARG, "rt_prepare: null context");
    if (!c->have_optics) return fail(RT_ERR_STATE, "rt_prepare: call rt_upload_optics first");
    RT_HIP(hipSetDevice(c->device));
    if (c->maj_dirty)
    {
        const int rc = build_majorants(c);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_synchronize(rt_context* c)
{
    if (!c) return fail(RT_ERR_ARG, "rt_synchronize: null context");
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipStreamSynchronize(c->stream));
    unsigned long long aborted = 0;
    RT_HIP(hipMemcpy(&aborted, c->d_events + RT_EV_ABORT, sizeof(aborted), hipMemcpyDeviceToHost));
    if (aborted)
    {
        RT_HIP(hipMemset(c->d_events + RT_EV_ABORT, 0, sizeof(aborted)));
        return fail(RT_ERR_ABORTED, "rt_synchronize: %llu photons abandoned by the step guard", aborted);
    }
    return RT_OK;
}

int rt_download_tallies(rt_context* c, uint64_t* const out[RT_N_TALLIES])
{
    if (!c || !out) return fail(RT_ERR_ARG, "rt_download_tallies: null argument");
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipStreamSynchronize(c->stream));
    size_t off = 0;
    for (int t = 0; t < RT_N_TALLIES; ++t)
    {
        const size_t n = t < RT_ATMOS_DIRECT ? c->ncol : c->ncell;
        if (!out[t]) return fail(RT_ERR_ARG, "rt_download_tallies: out[%d] is null", t);
        RT_HIP(hipMemcpy(out[t], c->d_tallies + off, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        off += n;
    }
    return RT_OK;
}

int rt_host_alloc(size_t bytes, void** out)
{
    if (!out) return fail(RT_ERR_ARG, "rt_host_alloc: null pointer");
    *out = nullptr;
    RT_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return RT_OK;
}

int rt_host_free(void* p)
{
    if (p) RT_HIP(hipHostFree(p));
    return RT_OK;
}

int rt_download_tallies_async(rt_context* c, uint64_t* const out[RT_N_TALLIES], int* ticket)
{
    if (!c || !out || !ticket) return fail(RT_ERR_ARG, "rt_download_tallies_async: null argument");
    for (int t = 0; t < RT_N_TALLIES; ++t)
        if (!out[t]) return fail(RT_ERR_ARG, "rt_download_tallies_async: out[%d] is null", t);
    RT_HIP(hipSetDevice(c->device));
    if (!c->copy_stream)
    {
        RT_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        RT_HIP(hipEventCreateWithFlags(&c->dl_ready, hipEventDisableTiming));
        for (int t = 0; t < 2; ++t) RT_HIP(hipEventCreateWithFlags(&c->dl_done[t], hipEventDisableTiming));
    }
    const int tk = c->dl_next;
    c->dl_next ^= 1;
    // the ticket's earlier download (two calls ago) must have left its host arrays: the caller has waited for it or
    // waits now (a host array may be the same one again)
    RT_HIP(hipEventSynchronize(c->dl_done[tk]));
    // behind everything enqueued on the context's stream so far, on a stream of its own: the next trace does not wait
    RT_HIP(hipEventRecord(c->dl_ready, c->stream));
    RT_HIP(hipStreamWaitEvent(c->copy_stream, c->dl_ready, 0));
    size_t off = 0;
    for (int t = 0; t < RT_N_TALLIES; ++t)
    {
        const size_t n = t < RT_ATMOS_DIRECT ? c->ncol : c->ncell;
        RT_HIP(hipMemcpyAsync(out[t], c->d_tallies + off, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->copy_stream));
        off += n;
    }
    RT_HIP(hipEventRecord(c->dl_done[tk], c->copy_stream));
    c->dl_src[tk] = c->d_tallies;
    *ticket = tk;
    return RT_OK;
}

int rt_download_wait(rt_context* c, int ticket)
{
    if (!c || ticket < 0 || ticket > 1) return fail(RT_ERR_ARG, "rt_download_wait: bad argument");
    if (!c->dl_done[ticket]) return fail(RT_ERR_STATE, "rt_download_wait: no download was started");
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipEventSynchronize(c->dl_done[ticket]));
    return RT_OK;
}

int rt_swap_tally_buffer(rt_context* c)
{
    if (!c) return fail(RT_ERR_ARG, "rt_swap_tally_buffer: null context");
    if (!c->owns_tallies)
        return fail(RT_ERR_STATE, "rt_swap_tally_buffer: the context accumulates into a caller-owned buffer (rt_bind_tally_buffer)");
    RT_HIP(hipSetDevice(c->device));
    if (!c->d_tallies_alt)
    {
        RT_HIP(hipMalloc(&c->d_tallies_alt, sizeof(unsigned long long) * c->nwords));
        RT_HIP(hipMemsetAsync(c->d_tallies_alt, 0, sizeof(unsigned long long) * c->nwords, c->stream));
    }
    std::swap(c->d_tallies, c->d_tallies_alt);
    // whatever is enqueued on the context's stream from now on (reset, trace) touches the new current buffer: it must
    // not run ahead of a download that still reads that buffer
    for (int t = 0; t < 2; ++t)
        if (c->dl_done[t] && c->dl_src[t] == c->d_tallies)
            RT_HIP(hipStreamWaitEvent(c->stream, c->dl_done[t], 0));
    return RT_OK;
}

int rt_upload_tallies(rt_context* c, const uint64_t* const in[RT_N_TALLIES])
{
    if (!c || !in) return fail(RT_ERR_ARG, "rt_upload_tallies: null argument");
    RT_HIP(hipSetDevice(c->device));
    unsigned long long* tmp = nullptr;
    RT_HIP(hipMalloc(&tmp, c->nwords * sizeof(uint64_t)));
    size_t off = 0;
    for (int t = 0; t < RT_N_TALLIES; ++t)
    {
        const size_t n = t < RT_ATMOS_DIRECT ? c->ncol : c->ncell;
        if (!in[t]) { (void)hipFree(tmp); return fail(RT_ERR_ARG, "rt_upload_tallies: in[%d] is null", t); }
        hipError_t e = hipMemcpyAsync(tmp + off, in[t], n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { (void)hipFree(tmp); return fail(RT_ERR_HIP, "rt_upload_tallies: %s", hipGetErrorString(e)); }
        off += n;
    }
    hipLaunchKernelGGL(rt_add_u64_kernel, dim3(1024), dim3(256), 0, c->stream, c->d_tallies, tmp, c->nwords);
    hipError_t e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_upload_tallies: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_reset_tallies(rt_context* c)
{
    if (!c) return fail(RT_ERR_ARG, "rt_reset_tallies: null context");
    RT_HIP(hipSetDevice(c->device));
    if (c->atm_wire)
    {
        // the absorption counts live in the bound wire: zero the five 2-D arrays here and the wire's absorption words
        // (half the bytes of the 64-bit arrays they stand for), and its flag
        RT_HIP(hipMemsetAsync(c->d_tallies, 0, 5 * c->ncol * sizeof(uint64_t), c->stream));
        RT_HIP(hipMemsetAsync(c->atm_wire + 5 * c->ncol, 0, (2 * c->ncell + 1) * sizeof(uint32_t), c->stream));
        c->atm_photons = 0;
    }
    else
        RT_HIP(hipMemsetAsync(c->d_tallies, 0, c->nwords * sizeof(uint64_t), c->stream));
    // the abort counter (last word) survives a reset: rt_synchronize reports and clears it
    RT_HIP(hipMemsetAsync(c->d_events, 0, RT_EV_ABORT * sizeof(uint64_t), c->stream));
    RT_HIP(hipMemsetAsync(c->d_events + RT_N_EVENTS, 0, RT_N_SCHED * sizeof(uint64_t), c->stream));
    return RT_OK;
}

int rt_tally_device_buffer(rt_context* c, void** ptr, size_t* n_words)
{
    if (!c || !ptr || !n_words) return fail(RT_ERR_ARG, "rt_tally_device_buffer: null argument");
    *ptr = c->d_tallies;
    *n_words = c->nwords;
    return RT_OK;
}

int rt_bind_tally_buffer(rt_context* c, void* ptr, size_t n_words)
{
    if (!c || !ptr) return fail(RT_ERR_ARG, "rt_bind_tally_buffer: null argument");
    if (n_words != c->nwords)
        return fail(RT_ERR_ARG, "rt_bind_tally_buffer: buffer has %zu words, the grid needs %zu", n_words, c->nwords);
    if (((uintptr_t)ptr & 7u) != 0) return fail(RT_ERR_ARG, "rt_bind_tally_buffer: buffer not 8-byte aligned");
    RT_HIP(hipSetDevice(c->device));
    // The buffer is a kernel argument, so the switch is in stream order by itself: launches already enqueued keep
    // the old pointer. Only a buffer the context owns has to be idle before it is freed.
    if (c->owns_tallies && c->d_tallies)
    {
        RT_HIP(hipStreamSynchronize(c->stream));
        if (c->copy_stream) RT_HIP(hipStreamSynchronize(c->copy_stream));
        RT_HIP(hipFree(c->d_tallies));
        if (c->d_tallies_alt) RT_HIP(hipFree(c->d_tallies_alt));
        c->d_tallies_alt = nullptr;
    }
    c->d_tallies = (unsigned long long*)ptr;
    c->owns_tallies = false;
    return RT_OK;
}

int rt_bind_atmos_wire(rt_context* c, void* wire, size_t n_words)
{
    if (!c) return fail(RT_ERR_ARG, "rt_bind_atmos_wire: null context");
    if (wire)
    {
        if (n_words != c->nwords + 1)
            return fail(RT_ERR_ARG, "rt_bind_atmos_wire: wire has %zu words, the grid needs %zu", n_words, c->nwords + 1);
        if (((uintptr_t)wire & 3u) != 0) return fail(RT_ERR_ARG, "rt_bind_atmos_wire: wire not 4-byte aligned");
    }
    c->atm_wire = static_cast<uint32_t*>(wire);
    c->atm_photons = 0;
    return RT_OK;
}

int rt_allreduce(rt_context* const ctx[], int n)
{
    return allreduce_impl(ctx, n);
}

int rt_tallies_narrow(rt_context* c, const void* tallies, void* wire, int world_size)
{
    if (!c || !wire) return fail(RT_ERR_ARG, "rt_tallies_narrow: null argument");
    if (world_size < 1 || world_size > 65536) return fail(RT_ERR_ARG, "rt_tallies_narrow: world size %d", world_size);
    if (((uintptr_t)wire & 3u) != 0 || ((uintptr_t)tallies & 7u) != 0) return fail(RT_ERR_ARG, "rt_tallies_narrow: misaligned buffer");
    RT_HIP(hipSetDevice(c->device));
    return launch_narrow(c, tallies ? (const unsigned long long*)tallies : c->d_tallies, (uint32_t*)wire, world_size);
}

int rt_tallies_widen(rt_context* c, void* tallies, const void* wire)
{
    if (!c || !wire) return fail(RT_ERR_ARG, "rt_tallies_widen: null argument");
    if (((uintptr_t)wire & 3u) != 0 || ((uintptr_t)tallies & 7u) != 0) return fail(RT_ERR_ARG, "rt_tallies_widen: misaligned buffer");
    RT_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(rt_widen_kernel, dim3(2048), dim3(256), 0, c->stream,
                       tallies ? (unsigned long long*)tallies : c->d_tallies, (const uint32_t*)wire, c->nwords,
                       wire == c->atm_wire ? 5 * c->ncol : c->nwords);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int rt_get_slabs(rt_context* c, rt_slab_info* out)
{
    if (!c || !out) return fail(RT_ERR_ARG, "rt_get_slabs: null argument");
    if (!c->have_optics) return fail(RT_ERR_STATE, "rt_get_slabs: call rt_upload_optics first");
    if (c->maj_dirty)
    {
        RT_HIP(hipSetDevice(c->device));
        const int rc = build_majorants(c);
        if (rc != RT_OK) return rc;
    }
    *out = c->slabs;
    return RT_OK;
}

int rt_get_majorants(rt_context* c, float* out_pairs, size_t n_pairs)
{
    if (!c || !out_pairs) return fail(RT_ERR_ARG, "rt_get_majorants: null argument");
    if (!c->have_optics) return fail(RT_ERR_STATE, "rt_get_majorants: call rt_upload_optics first");
    RT_HIP(hipSetDevice(c->device));
    if (c->maj_dirty)
    {
        const int rc = build_majorants(c);
        if (rc != RT_OK) return rc;
    }
    if (n_pairs != c->ncoarse)
        return fail(RT_ERR_ARG, "rt_get_majorants: the grid has %zu supervoxels, caller asked for %zu", c->ncoarse, n_pairs);
    if (n_pairs)
    {
        // the table holds {M, extents of the supervoxel's box}; the steps take 1/M from M (region_reciprocal)
        RT_HIP(hipMemcpy(out_pairs, c->d_maj, sizeof(float2) * n_pairs, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < n_pairs; ++e)
            out_pairs[2 * e + 1] = out_pairs[2 * e] >= 0x1p-100f ? 1.0f / out_pairs[2 * e] : FLT_MAX;
    }
    return RT_OK;
}

int rt_get_region_boxes(rt_context* c, uint32_t* out, size_t n)
{
    if (!c || !out) return fail(RT_ERR_ARG, "rt_get_region_boxes: null argument");
    if (!c->have_optics) return fail(RT_ERR_STATE, "rt_get_region_boxes: call rt_upload_optics first");
    RT_HIP(hipSetDevice(c->device));
    if (c->maj_dirty)
    {
        const int rc = build_majorants(c);
        if (rc != RT_OK) return rc;
    }
    if (n != c->ncoarse)
        return fail(RT_ERR_ARG, "rt_get_region_boxes: the grid has %zu supervoxels, caller asked for %zu", c->ncoarse, n);
    if (n)
    {
        std::vector<float2> h(n);
        RT_HIP(hipMemcpy(h.data(), c->d_maj, sizeof(float2) * n, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < n; ++e)
            memcpy(&out[e], &h[e].y, sizeof(uint32_t));
    }
    return RT_OK;
}

int rt_get_events(rt_context* c, uint64_t out[RT_N_EVENTS])
{
    if (!c || !out) return fail(RT_ERR_ARG, "rt_get_events: null argument");
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipStreamSynchronize(c->stream));
    RT_HIP(hipMemcpy(out, c->d_events, RT_N_EVENTS * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_get_sched_stats(rt_context* c, uint64_t out[RT_N_SCHED])
{
    if (!c || !out) return fail(RT_ERR_ARG, "rt_get_sched_stats: null argument");
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipStreamSynchronize(c->stream));
    RT_HIP(hipMemcpy(out, c->d_events + RT_N_EVENTS, RT_N_SCHED * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_last_trace_ms(rt_context* c, float* ms)
{
    if (!c || !ms) return fail(RT_ERR_ARG, "rt_last_trace_ms: null argument");
    if (!c->timed) return fail(RT_ERR_STATE, "rt_last_trace_ms: no trace launched yet");
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipEventSynchronize(c->ev_stop));
    RT_HIP(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return RT_OK;
}

int rt_selftest_philox(int device_id, const uint32_t* ctr, size_t n, const uint32_t key[2], uint32_t* out)
{
    if (!ctr || !key || !out) return fail(RT_ERR_ARG, "rt_selftest_philox: null argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(device_id));
    HookBuffer<uint32_t> d_in, d_out;
    RT_HIP(hipMalloc(&d_in.p, 16 * n));
    RT_HIP(hipMalloc(&d_out.p, 16 * n));
    RT_HIP(hipMemcpy(d_in.p, ctr, 16 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rt_selftest_philox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_in.p, n, key[0], key[1], d_out.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpy(out, d_out.p, 16 * n, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_selftest_launch(rt_context* c, uint64_t seed, uint64_t pid_begin, size_t n, float* xy, uint32_t* words)
{
    if (!c || !xy) return fail(RT_ERR_ARG, "rt_selftest_launch: null argument");
    if (c->launch_log2k >= 0) return fail(RT_ERR_ARG, "rt_selftest_launch: uniform or Sobol launch only");
    if (n == 0) return RT_OK;
    if (pid_begin + n < pid_begin || ((pid_begin + n - 1) >> 32) != (pid_begin >> 32))
        return fail(RT_ERR_ARG, "rt_selftest_launch: the range crosses a multiple of 2^32");
    RT_HIP(hipSetDevice(c->device));
    HookBuffer<float> d_xy;
    HookBuffer<uint32_t> d_w;
    RT_HIP(hipMalloc(&d_xy.p, 8 * n));
    if (words) RT_HIP(hipMalloc(&d_w.p, 8 * n));
    uint32_t sx = 0, sy = 0;
    const int sobol = c->launch_log2k == kLaunchSobol;
    if (sobol) sobol_shift(seed, pid_begin >> 32, sx, sy);
    hipLaunchKernelGGL(rt_selftest_launch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, sobol, (uint32_t)seed,
                       (uint32_t)(seed >> 32), sx, sy, (unsigned long long)pid_begin, n, (float)c->itot * c->dx,
                       (float)c->jtot * c->dy, d_xy.p, d_w.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpy(xy, d_xy.p, 8 * n, hipMemcpyDeviceToHost));
    if (words) RT_HIP(hipMemcpy(words, d_w.p, 8 * n, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_selftest_exact(int device_id, int which, uint32_t first_bits, uint64_t count, uint64_t out[2])
{
    if (!out || which < 0 || which > 1) return fail(RT_ERR_ARG, "rt_selftest_exact: bad argument");
    if (count > (1ull << 32) - first_bits) return fail(RT_ERR_ARG, "rt_selftest_exact: range beyond 2^32 bit patterns");
    RT_HIP(hipSetDevice(device_id));
    HookBuffer<unsigned long long> d_out;
    const unsigned long long init[2] = {0ull, ~0ull};
    RT_HIP(hipMalloc(&d_out.p, sizeof(init)));
    RT_HIP(hipMemcpy(d_out.p, init, sizeof(init), hipMemcpyHostToDevice));
    if (count > 0)
        hipLaunchKernelGGL(rt_selftest_exact_kernel, dim3(256 * 16), dim3(256), 0, 0, which, first_bits,
                           (unsigned long long)count, d_out.p);
    RT_HIP(hipGetLastError());
    unsigned long long res[2];
    RT_HIP(hipMemcpy(res, d_out.p, sizeof(res), hipMemcpyDeviceToHost));
    out[0] = res[0]; out[1] = res[1];
    return RT_OK;
}

int rt_selftest_math(int device_id, int which, const float* in, size_t n, float* out)
{
    if (!in || !out) return fail(RT_ERR_ARG, "rt_selftest_math: null argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(device_id));
    HookBuffer<float> d_in, d_out;
    RT_HIP(hipMalloc(&d_in.p, 4 * n));
    RT_HIP(hipMalloc(&d_out.p, 4 * n));
    RT_HIP(hipMemcpy(d_in.p, in, 4 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rt_selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, which, d_in.p, n, d_out.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpy(out, d_out.p, 4 * n, hipMemcpyDeviceToHost));
    return RT_OK;
}

} // extern "C"
