// Microbenchmark: the random-access ("gather") roofline of one MI355X — how many independent 16-byte
// gathers and 64-bit no-return atomics per second HBM sustains when every access touches a different
// 64-byte sector of a multi-GiB array (the tracer's collision gathers and absorption tallies on the
// LES grid). Build: hipcc --offload-arch=gfx950 -O2 tools/gather_rate.hip -o tools/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: gathers only; 1: atomics only; 2: five gathers per atomic (the tracer's mix on C4)
template <int MODE, int ILP>
__global__ void __launch_bounds__(256) rate_kernel(const float4* __restrict__ cells, unsigned long long* tallies,
                                                    uint32_t cell_mask, uint32_t tally_mask, int iters, float* out)
{
    uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it)
    {
        uint32_t idx[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) { s = s * 747796405u + 2891336453u; idx[u] = mix(s); }
        if (MODE == 0 || MODE == 2)
        {
            float4 v[ILP];
#pragma unroll
            for (int u = 0; u < ILP; ++u) v[u] = cells[idx[u] & cell_mask];
#pragma unroll
            for (int u = 0; u < ILP; ++u) acc += v[u].x + v[u].w;
        }
        if (MODE == 1)
        {
#pragma unroll
            for (int u = 0; u < ILP; ++u)
                __hip_atomic_fetch_add(&tallies[idx[u] & tally_mask], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE == 2 && (it % 5) == 0)
        {
#pragma unroll
            for (int u = 0; u < ILP; ++u)
                __hip_atomic_fetch_add(&tallies[(idx[u] >> 3) & tally_mask], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE, int ILP>
int run(const char* name, int log2_bytes, int waves_per_cu, int cus, const float4* cells, unsigned long long* tallies, float* out)
{
    const uint32_t cell_mask = (uint32_t)((1ull << (log2_bytes - 4)) - 1), tally_mask = (uint32_t)((1ull << (log2_bytes - 3)) - 1);
    const int blocks = cus * waves_per_cu / 4;
    const int iters = 400;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, cells, tallies, cell_mask, tally_mask, 20, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, cells, tallies, cell_mask, tally_mask, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double threads = (double)blocks * 256;
    const double gathers = (MODE == 1 ? 0.0 : threads * iters * ILP);
    const double atomics = (MODE == 1 ? threads * iters * ILP : (MODE == 2 ? threads * (iters / 5) * ILP : 0.0));
    const double ops = gathers + atomics;
    printf("%-22s array 2^%d B  waves/CU=%2d ILP=%d  %.2f ms  %.1f G accesses/s  (= %.0f GB/s of 64-B sectors; gathers %.1f G/s, atomics %.1f G/s)\n",
           name, log2_bytes, waves_per_cu, ILP, ms, ops / ms * 1e-6, ops * 64 / ms * 1e-6, gathers / ms * 1e-6, atomics / ms * 1e-6);
    return 0;
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%d CUs\n", cus);
    float4* cells; unsigned long long* tallies; float* out;
    const size_t bytes = 1ull << 31;
    CHECK(hipMalloc(&cells, bytes)); CHECK(hipMalloc(&tallies, bytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(cells, 0, bytes)); CHECK(hipMemset(tallies, 0, bytes));
    for (int lb : {31, 27, 24})
    {
        if (run<0, 1>("gather16", lb, 32, cus, cells, tallies, out)) return 1;
        if (run<0, 4>("gather16", lb, 32, cus, cells, tallies, out)) return 1;
        if (run<0, 4>("gather16", lb, 16, cus, cells, tallies, out)) return 1;
        if (run<0, 8>("gather16", lb, 32, cus, cells, tallies, out)) return 1;
        if (run<1, 1>("atomic64", lb, 32, cus, cells, tallies, out)) return 1;
        if (run<1, 4>("atomic64", lb, 32, cus, cells, tallies, out)) return 1;
        if (run<2, 4>("5 gathers : 1 atomic", lb, 32, cus, cells, tallies, out)) return 1;
    }
    return 0;
}
