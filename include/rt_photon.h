/* rt_photon.h — C ABI of the MI355X photon-tracing library (librt_photon.so).
 *
 * The reference (Chiil/raytracer) has no plugin or FFI layer: its host drivers #include the CUDA
 * kernels and launch them directly. This header is the boundary a drop-in replacement for that
 * path exports: each entry point names the reference code it stands in for (file:line relative
 * to /root/reference/src/). Plain pointers and sizes only; every call returns 0 on success and a
 * non-zero rt_status otherwise, with a message available from rt_last_error() (the reference
 * prints and exit()s instead, raytracer_kernels.cu:101-110). No exceptions cross this boundary.
 *
 * Ownership: the context owns all device memory and frees it in rt_destroy (the reference never
 * frees, free_gpu of raytracer_kernels.cu:123-128 is uncalled). Host pointers handed to the
 * library stay the caller's. Threading: one context per GPU; a context is not thread-safe, calls
 * on different contexts are independent.
 */
#ifndef RT_PHOTON_H
#define RT_PHOTON_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;

enum rt_status
{
    RT_OK = 0,
    RT_ERR_ARG = 1,      /* bad argument */
    RT_ERR_HIP = 2,      /* a HIP runtime call failed */
    RT_ERR_STATE = 3,    /* call order: optics or scene not set */
    RT_ERR_ABORTED = 4   /* photons were abandoned by the step guard (non-finite state) */
};

/* Order of the seven tally arrays everywhere in this API (file names of
 * raytracer_3d.cu:197-203 without ".bin"). The first five hold itot*jtot counts laid out [j][i],
 * the last two itot*jtot*ktot counts laid out [k][j][i]; all uint64. */
enum rt_tally
{
    RT_TOA_DOWN = 0,
    RT_TOA_UP = 1,
    RT_SURFACE_DOWN_DIRECT = 2,
    RT_SURFACE_DOWN_DIFFUSE = 3,
    RT_SURFACE_UP = 4,
    RT_ATMOS_DIRECT = 5,
    RT_ATMOS_DIFFUSE = 6,
    RT_N_TALLIES = 7
};

/* Event counters of one or more rt_trace calls (rt_get_events). They feed the algorithmic-bytes
 * figure of the roofline. */
enum rt_event
{
    RT_EV_STEPS = 0,        /* traversal steps: one majorant region crossed or collision site reached */
    RT_EV_COLLISION_TESTS,  /* gathers of k_tot */
    RT_EV_REAL,             /* non-null collisions (gather of ssa) */
    RT_EV_SCATTER,          /* scatterings (gather of asy, k_cloud) */
    RT_EV_CLOUD_SCATTER,
    RT_EV_SURFACE,          /* surface arrivals */
    RT_EV_TOA,              /* escapes at the top */
    RT_EV_ABSORB,           /* absorptions in the atmosphere */
    RT_EV_TRANSITION,       /* faces of majorant regions crossed (domain top and bottom included) */
    RT_EV_ABORT,            /* photons abandoned by the step/NaN guard; must stay 0 */
    RT_N_EVENTS
};

/* What the pre-pass derives from the optical-property grids: the horizontal slabs used for
 * null-collision tracking. Replaces cloud_dims of cloud_mask_kernel (raytracer_kernels.cu:263-305)
 * and k_ext_null / k_ext_null_gas of raytracer_3d_io.cu:60-77. */
typedef struct rt_slab_info
{
    int nslab;          /* 1..3 non-empty slabs, bottom to top */
    float zface[4];     /* nslab+1 face heights (m) */
    float kmaj[3];      /* majorant extinction (1/m) per slab */
    int k_lo, k_hi;     /* lowest / highest level holding any asy > 0; -1 if none */
    float cloud_min;    /* = k_lo*dz,      cloud_dims[0] of the reference (0 if none) */
    float cloud_max;    /* = (k_hi+1)*dz,  cloud_dims[1] of the reference (0 if none) */
    float sun[3];       /* direction cosines of the direct beam */
    int gslab;          /* index of the slab refined into a grid of local majorants, -1 if none */
    int nci, ncj, nck;  /* supervoxels of that grid per axis */
    float hx, hy, hz;   /* supervoxel edge lengths (m) */
} rt_slab_info;

/* Message of the last failed call on this thread. */
const char* rt_last_error(void);

/* Number of visible GPUs (hipGetDeviceCount). */
int rt_device_count(int* count);

/* Creates a context on GPU `device_id` for an itot x jtot x ktot grid with spacings dx,dy,dz (m)
 * and allocates the seven tally arrays, zeroed. Stands in for the allocation block of
 * run_ray_tracer, raytracer_3d.cu:81-104 (allocate_gpu, raytracer_kernels.cu:113-120). */
int rt_create(int device_id, int itot, int jtot, int ktot, float dx, float dy, float dz,
              rt_context** out);

/* Frees everything the context owns. (free_gpu, raytracer_kernels.cu:123-128.) */
int rt_destroy(rt_context* ctx);

/* Uploads the four optical-property fields (host pointers, itot*jtot*ktot float32 each, [k][j][i],
 * the make_binary.py file layout) and runs the device pre-pass that packs them for the tracer and
 * derives majorants and cloud bounds. Stands in for copy_to_gpu of the two grids
 * (raytracer_3d.cu:83-87), the host loop of raytracer_3d_io.cu:60-77 and cloud_mask_kernel
 * (raytracer_kernels.cu:263-305, launched at raytracer_3d.cu:116-123). */
int rt_upload_optics(rt_context* ctx, const float* k_ext_gas, const float* k_ext_cloud,
                     const float* ssa, const float* asy);

/* The same fields to n contexts (one per GPU, same grid) at once: the host arrays are page-locked for the call and
 * the copies to all GPUs are enqueued before any is waited for, so the GPUs' links to the host work side by side
 * (the multi-GPU drivers; rt_upload_optics per GPU in turn moves 2 GiB of pageable memory eight times over, one after
 * another, on the 1024 x 1024 x 128 grid). The reference has one GPU. */
int rt_upload_optics_multi(rt_context* const ctx[], int n, const float* k_ext_gas, const float* k_ext_cloud,
                           const float* ssa, const float* asy);

/* Same, from device pointers already resident on the context's GPU (no PCIe in the path). */
int rt_upload_optics_device(rt_context* ctx, const float* d_k_ext_gas, const float* d_k_ext_cloud,
                            const float* d_ssa, const float* d_asy);

/* Solar geometry and surface albedo: zenith and azimuth in radians (as float, the way
 * raytracer_3d.cu:25-30 holds them) and the Lambertian albedo (raytracer_3d.cu:24). */
int rt_set_scene(rt_context* ctx, float zenith_rad, float azimuth_rad, float albedo);

/* Options by name:
 *   "fixed_dot"      0 (default): dot() as the reference writes it, raytracer_kernels.cu:70-74;
 *                    1: the mathematically intended v1.z*v2.z
 *   "planar"         1: the scattering rule of the reference's 2D tracer, src/raytracer.cpp:360-364 and
 *                    :280-282 — directions stay in the x-z plane (a scattering turns the direction by
 *                    +-acos(mu) in that plane, the surface reflects to (+-mu, 0, sqrt(1-mu^2))). With
 *                    jtot = 1 and azimuth = 0 the library is the 2D tracer (drivers/raytracer_gpu)
 *   "supergrid_x|y|z" supervoxel size in cells of the majorant grid laid over the cloud slab
 *                    (0 = slab majorants only, the reference's scheme; default -1 = automatic: the whole number of
 *                    cells nearest to 50 m x 50 m x 50 m, with a box per supervoxel, when the launch is stratified or sorted
 *                    or the grid fits the Infinity Cache, else 8, 8, 4 unmerged)
 *   "levels"         every supervoxel gets a box of its own -- the supervoxels around it, up to 2^levels per axis, that
 *                    a photon standing in it crosses as one majorant region (cubes where the field is homogeneous all
 *                    around, tall columns of clear air between clouds, flat layers under them; boxes of neighbours
 *                    overlap) -- grown a layer at a time for as long as (majorant - mean extinction) * 6 V / S of the
 *                    box <= "merge_tau_milli" / 1000 -- the optical depth of null collisions a crossing photon would
 *                    meet (levels = 0: one region per supervoxel; 0..6, an extent at most 31 to either side; defaults -1 = automatic, 5 or 0 as for the
 *                    supervoxel size, and 750)
 *   "th_event", "th_scatter", "low_traverse"  lanes of a wave that must wait for the event /
 *                    scatter stage before it runs, and the number of traversing lanes below which
 *                    waiting lanes are served regardless (scheduling only: results do not change; they steer the
 *                    lane-resident kernel, option "pool" = 0 -- the pool kernel schedules by slot counts and has no
 *                    scatter stage: its event round also turns the photon and takes its next step)
 *   "launch_log2k"   -1 (default): launch positions uniform over the domain top (two Philox draws);
 *                    -2 (also option "launch_sobol" = 1): photon ID n enters at point (n mod 2^32) of the
 *                    two-dimensional Sobol sequence, digitally shifted by a per-seed, per-2^32-ID-block draw -- the
 *                    reference's launch (scrambled Sobol32 dimensions 0 and 1 at offset tid * photons_per_thread,
 *                    raytracer_kernels.cu:239-253, 331, tables raytracer_3d.cu:135-145) as a pure function of the
 *                    photon ID: any aligned block of 2^m consecutive IDs puts exactly one photon into every
 *                    elementary interval 2^-a x 2^-(m-a) of the domain top, on any grid and for any photon count;
 *                    k >= 0: stratified by column — 2^k consecutive photon IDs per column, columns in row-major
 *                    order, every pass over the columns rotated by a per-pass random offset (even only over whole
 *                    passes of ncol * 2^k IDs)
 *   "launch_tile_log2" stratified launch: T > 0 counts the columns tile by tile (tiles of 2^T x 2^T columns, row-major
 *                    inside, tiles row by row; itot and jtot must be multiples of 2^T, else rt_trace refuses) instead of
 *                    row by row (0, default), so that the photons in flight together share a neighbourhood in both
 *                    directions: with T = 5 and k = 8 a tile's cells stay in the XCD's L2 while its 2^18 photons come
 *                    through (1024 x 1024 x 128 grid: +6 % photons/s, a third of the memory traffic)
 *   "launch_sort"    uniform launch positions, pool kernel: -1 (default) = calls of 2^19 photons or more, 1 = all,
 *                    0 = none take their photon IDs in the order of the launch tile (2^"sort_tile_log2" columns
 *                    square, default 5, enlarged until there are at most 4096 tiles): a counting sort on the
 *                    device ahead of the tracer, in batches of 2^"sort_batch_log2" IDs (default 30; 4 bytes of
 *                    list per ID, allocated by the first rt_trace that needs it and kept), which also counts
 *                    toa_down tile by tile. Which photon is traced when changes, no trajectory and no tally does;
 *                    it is what keeps the reference's own launch sampling within 5 % of the stratified order on
 *                    grids larger than the caches (the reference has no counterpart: its launch order is the
 *                    thread index, raytracer_kernels.cu:323-331)
 *   "th_launch"      free slots of a wave's pool that trigger a launch (regeneration) round
 *   "tally_match"    pool kernel: rounds in which a wave first adds the 2-D tally words that many of its lanes
 *                    share (the direct beam of one launch column) with one atomic each, before the runs of equal
 *                    neighbouring lanes; 0..16, -1 (default) = 6 on grids of up to 2^15 columns, 0 on larger ones
 *                    (scheduling only; the reference adds one by one, raytracer_kernels.cu:256-261)
 *   "chunk"          photon IDs a wave of the pool kernel takes at a time from the device-wide cursor
 *                    (power of two, default 256; halved for small jobs until every wave finds work)
 *   "pool"           photons each wave keeps in LDS (64, 96, 128, 160, 192 or 256) and compacts by
 *                    stage into full waves; 0 = photon state stays in the lane's registers
 *                    (scheduling only: results do not change). "pool_waves_per_cu" caps residency.
 *   "blocks_per_cu"  workgroups of 256 lanes per compute unit of the persistent launch (default 6)
 *   "stats"          1: rt_trace also counts events (rt_get_events); 0 (default): it does not
 *   "wire32"         1 (default): rt_allreduce sums 32-bit counts when every count allows it; 0: always 64-bit */
int rt_set_option(rt_context* ctx, const char* name, long long value);

/* Launch on this HIP stream (a hipStream_t) from now on; NULL = the null stream. */
int rt_set_stream(rt_context* ctx, void* hip_stream);

/* Traces photons with IDs [photon_id_begin, photon_id_begin + photon_count) and ADDS their
 * events to the device tallies (the restart-friendly accumulation raytracer_3d.cu:89 intends).
 * Every random number is Philox4x32-10(key = seed, counter = (photon id, step, purpose)), so the
 * tallies of a photon range do not depend on how it is split over calls, launches or GPUs.
 * Asynchronous on the context's stream. The default (pool) kernel packs photon ids into 40 bits: ranges that end
 * beyond 2^40 are refused (RT_ERR_ARG) unless option "pool" = 0 selects the lane-resident kernel, which carries
 * 64-bit ids. Stands in for the ray_tracer_kernel launch, raytracer_3d.cu:151-163 (kernel:
 * raytracer_kernels.cu:307-569). */
int rt_trace(rt_context* ctx, uint64_t seed, uint64_t photon_id_begin, uint64_t photon_count);

/* Runs everything rt_trace would otherwise do on its first call after rt_upload_optics or an option change:
 * the device pre-pass over supervoxels and the host-side merge into majorant regions (allocates the tables).
 * Calling it ahead of a timed or latency-critical section makes every rt_trace in it a pure asynchronous
 * launch -- except the first one that sorts its launches (option "launch_sort": uniform launch positions, 2^19
 * photons or more), which allocates the list for its batch size; a warm-up call of the largest size does that too.
 * Part of what cloud_mask_kernel's launch stands for (raytracer_3d.cu:116-130). */
int rt_prepare(rt_context* ctx);

/* rt_trace for callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc. It has NO counterpart in the reference. Same arguments, same results (bit for
 * bit) and the same asynchronous launch on the context's stream as rt_trace, but it calls no hipMalloc, hipFree
 * or any other allocating routine: the device words the pool kernel needs beyond its inputs and outputs -- the
 * eight photon-ID cursors and the launch's cold block -- are taken from a 2 KiB static device array of the
 * library instead of the context's allocation, and the call records no timing events. Limits, refused with
 * RT_ERR_ARG / RT_ERR_STATE and a message instead of allocating: the context must use the pool kernel (option
 * "pool" > 0), photon ids end at 2^40, and the majorant tables must already exist (rt_prepare after the last
 * rt_upload_optics / option change). A launch sort (option "launch_sort") is done only into a list that an earlier
 * rt_trace of at least this call's batch size has allocated; without one the IDs are taken in ascending order
 * (slower on large grids, same results). Every GPU holds ONE such array, shared by all contexts on it: calls of
 * rt_trace_noalloc that overlap on one GPU (two streams, two threads) race on it; calls on one stream, or
 * separated by rt_synchronize, do not. rt_trace itself is unaffected and may overlap with it. */
int rt_trace_noalloc(rt_context* ctx, uint64_t seed, uint64_t photon_id_begin, uint64_t photon_count);

/* Blocks until the context's stream is idle (cudaDeviceSynchronize at raytracer_3d.cu:165) and
 * reports RT_ERR_ABORTED (once) if any photon hit the step guard since the last call. */
int rt_synchronize(rt_context* ctx);

/* Copies the seven tally arrays to caller-allocated host arrays, out[rt_tally].
 * (copy_from_gpu, raytracer_3d.cu:174-180.) Synchronises first. */
int rt_download_tallies(rt_context* ctx, uint64_t* const out[RT_N_TALLIES]);

/* The same copy without blocking the host or the context's stream: the seven arrays leave the context's CURRENT tally
 * buffer over a copy stream of the library, behind everything enqueued on the context's stream so far, and *ticket
 * (0 or 1, alternating) names the copy for rt_download_wait. With host arrays from rt_host_alloc (pinned) the copy
 * runs beside whatever the context's stream does next -- with rt_swap_tally_buffer, the next job's trace: the zenith
 * sweep of the io flow (raytracer_3d_io.cu:174-212 copies seven arrays and writes seven files per run, strictly
 * after its kernel) then hides download and file writes behind the next run's trace. Pageable host arrays work too
 * (the copy is then staged by the runtime). At most two copies are in flight: a third call first waits for the one
 * that used its ticket. */
int rt_download_tallies_async(rt_context* ctx, uint64_t* const out[RT_N_TALLIES], int* ticket);

/* Blocks until the copy `ticket` has arrived in its host arrays. */
int rt_download_wait(rt_context* ctx, int ticket);

/* Switches the context to its second tally buffer (allocated and zeroed at the first call; the context must own its
 * tallies), in stream order: launches already enqueued keep the buffer they were given, later ones (rt_reset_tallies,
 * rt_trace, downloads) use the other -- and wait on the device, not on the host, for a download that still reads it.
 * No counterpart in the reference (one buffer, one run). */
int rt_swap_tally_buffer(rt_context* ctx);

/* Pinned host memory for rt_download_tallies_async / rt_upload_optics (hipHostMalloc / hipHostFree), so that a host
 * program without HIP headers -- the C++ drivers -- can give the library buffers that DMA runs to directly. */
int rt_host_alloc(size_t bytes, void** out);
int rt_host_free(void* p);

/* Adds caller-provided counts into the device tallies (restart: raytracer_3d.cu:98-104 copies
 * the host arrays up "to enable restarts later"). */
int rt_upload_tallies(rt_context* ctx, const uint64_t* const in[RT_N_TALLIES]);

/* Zeroes the device tallies and event counters (except the abort counter, which only
 * rt_synchronize reports and clears). */
int rt_reset_tallies(rt_context* ctx);

/* The concatenated device tally buffer: 5*itot*jtot + 2*itot*jtot*ktot uint64 words in rt_tally
 * order. For the one collective of the multi-GPU path (sum all-reduce over RCCL, done by the
 * host side on this pointer). No counterpart in the reference (single GPU only). */
int rt_tally_device_buffer(rt_context* ctx, void** device_ptr, size_t* n_words);

/* Makes the context accumulate into a caller-owned device buffer of the same layout (e.g. a
 * tensor the host framework allocated, so that it can hand it to its collective). The buffer must
 * outlive the context or a later rebind; its current contents are kept (not zeroed). The switch is in stream
 * order: launches already enqueued keep the buffer they were given, and the call does not wait for them (it
 * waits only when the buffer it replaces is the context's own, which it frees). */
int rt_bind_tally_buffer(rt_context* ctx, void* device_ptr, size_t n_words);

/* The path's one exchange step (the reference is single-GPU and has no counterpart): sums
 * the tallies of n contexts -- one per GPU, same grid, all in this process -- so that every context holds the
 * node total, with ONE RCCL all-reduce per GPU (ncclAllReduce, ncclSum) over xGMI inside one group call. RCCL is
 * loaded at run time (librccl.so.1; a copy already mapped into the process, e.g. PyTorch's, is reused) and the
 * communicators of a device set are created once and kept. Counts travel as 32-bit words (half the bytes: 1 GiB
 * instead of 2 GiB per GPU on the 1024x1024x128 grid) whenever every count of every GPU is below 2^32 / n, which
 * the narrowing kernel checks on the device; otherwise, or with option "wire32" = 0, as ncclUint64. Blocks until
 * the sums are in place on every GPU. Works for n = 1 too (RCCL is still called). */
int rt_allreduce(rt_context* const ctx[], int n);

/* The two halves of the 32-bit exchange for hosts that run the collective themselves, one process per GPU
 * (bench.py and raytracer_amd/distributed.py through torch.distributed = RCCL): rt_tallies_narrow writes the low
 * words of the tally buffer (`tallies`, or the context's current buffer if NULL) to wire[0 .. n_words) and raises
 * wire[n_words] to 1 if any count is at or above 2^32 / world_size (rounded down to a power of two); the host
 * all-reduces the n_words + 1 uint32 words with a sum; rt_tallies_widen then stores the sums back as 64-bit counts
 * unless wire[n_words] != 0 (some rank had a large count), in which case it leaves the tallies untouched and the
 * host must all-reduce the 64-bit buffer instead. Both are asynchronous on the context's stream. */
int rt_tallies_narrow(rt_context* ctx, const void* tallies, void* wire, int world_size);
int rt_tallies_widen(rt_context* ctx, void* tallies, const void* wire);

/* Multi-GPU jobs without the narrowing and widening passes over the absorption arrays (98 % of the tally words): while
 * a wire (nwords + 1 uint32 words, the layout rt_tallies_narrow writes) is bound, rt_trace counts ABSORPTIONS straight
 * into wire[word] as 32-bit words instead of into the 64-bit tally buffer (same word index; a photon is absorbed at
 * most once, so no word can pass the photons traced since the last reset: rt_trace refuses to go beyond 2^32 - 1 of
 * them, and the sum over ranks is safe while the job's photons over ALL ranks stay below 2^32 -- the caller's check).
 * The five 2-D arrays stay 64-bit (a photon can bounce off the surface any number of times: their words are the ones
 * the flag of rt_tallies_narrow guards). With the wire bound, rt_reset_tallies zeroes the 2-D arrays and the wire's
 * absorption words (half the bytes), rt_tallies_narrow narrows the 2-D arrays only (40 MB instead of 3.3 GB of traffic
 * on the 1024 x 1024 x 128 grid), and rt_tallies_widen -- needed only when the 64-bit result is read, not per job --
 * always widens the absorption words and the 2-D words unless the flag is raised. wire = NULL unbinds. Pool kernel
 * only. No counterpart in the reference (single GPU). */
int rt_bind_atmos_wire(rt_context* ctx, void* wire, size_t n_words);

/* Slabs, majorants and cloud bounds derived by rt_upload_optics + rt_set_scene. */
int rt_get_slabs(rt_context* ctx, rt_slab_info* out);

/* The majorant grid of the cloud slab as nck*ncj*nci pairs {M, 1/M}, [ck][cj][ci] (n_pairs must equal nck*ncj*nci of
 * rt_get_slabs): M of the supervoxel's box (rt_get_region_boxes) and the reciprocal the traverse steps take from it.
 * Test/diagnostic access; no counterpart in the reference. */
int rt_get_majorants(rt_context* ctx, float* out_pairs, size_t n_pairs);

/* The majorant box of every supervoxel, [ck][cj][ci] (n must equal nck*ncj*nci): six extents of 5 bits,
 * xm | xp << 5 | ym << 10 | yp << 15 | zm << 20 | zp << 25 -- a photon in supervoxel (ci, cj, ck) crosses the supervoxels
 * [ci - xm, ci + xp] x [cj - ym, cj + yp] x [ck - zm, ck + zp] as ONE region under the majorant rt_get_majorants returns for
 * it (boxes of neighbours overlap; option "levels": at most 2^levels supervoxels per axis, "merge_tau_milli": how much
 * null-collision optical depth a box may cost). Generalises the reference's three slabs with one majorant each
 * (raytracer_kernels.cu:360-362). Test/diagnostic access. */
int rt_get_region_boxes(rt_context* ctx, uint32_t* out, size_t n);

/* Event counters accumulated since the last rt_reset_tallies (needs option "stats" = 1 for all
 * but RT_EV_ABORT). Synchronises first. */
int rt_get_events(rt_context* ctx, uint64_t out[RT_N_EVENTS]);

/* Wave-scheduling statistics of "stats" launches since the last reset: rounds run and lanes
 * served per stage {T rounds, E rounds, S rounds, T lanes, E lanes, S lanes, slab rounds, slab lanes}
 * (T = traverse inside the grid, E = collision/surface event, S = new direction -- a stage of its own only in the
 * lane-resident kernel, 0 for the pool kernel --, slab = everything outside the grid). Diagnostic only. */
#define RT_N_SCHED 8
int rt_get_sched_stats(rt_context* ctx, uint64_t out[RT_N_SCHED]);

/* Device time of the most recent rt_trace launch in milliseconds (HIP events recorded on the
 * context's stream around the kernel; synchronises on the stop event). Mirrors the timer of
 * raytracer_3d.cu:149-169. */
int rt_last_trace_ms(rt_context* ctx, float* ms);

/* Test hook: Philox4x32-10 on the device, n counter blocks (4 words each) under one key. */
int rt_selftest_philox(int device_id, const uint32_t* ctr, size_t n, const uint32_t key[2], uint32_t* out);

/* Test hook: evaluates the device math (0 = log, 1 = x^(-1/3), 2 = sin(2*pi*x), 3 = cos(2*pi*x), 4 = the
 * Rayleigh cosine of a uniform draw, 5 = the Henyey-Greenstein cosine at g = 0.85) on n host floats. */
int rt_selftest_math(int device_id, int which, const float* in, size_t n, float* out);

/* Test hook: where photons pid_begin .. pid_begin + n - 1 enter the domain top under the context's launch option
 * (uniform or Sobol; the range must not cross a multiple of 2^32): xy[2n] positions in metres and, if words is not
 * NULL, the two 32-bit words they were scaled from (Philox draws, or the digitally shifted Sobol point -- what
 * curand(&state_x), curand(&state_y) return in raytracer_kernels.cu:248-249). */
int rt_selftest_launch(rt_context* ctx, uint64_t seed, uint64_t pid_begin, size_t n, float* xy, uint32_t* words);

/* Test hook: the device's short sqrt (which = 0) or reciprocal (which = 1) against the compiler's IEEE
 * sqrtf / division on the `count` consecutive float bit patterns from first_bits on; out[0] = number of
 * patterns whose results differ in any bit, out[1] = the lowest such pattern (~0 if none). */
int rt_selftest_exact(int device_id, int which, uint32_t first_bits, uint64_t count, uint64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif /* RT_PHOTON_H */
