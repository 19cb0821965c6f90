// Probe: can another kernel that needs LDS (as RCCL's collectives do) run while the persistent tracer
// occupies the GPU? Exposes probe_launch(stream, lds_bytes, blocks, spin_us) -> launches a kernel of
// 256-thread blocks with `lds_bytes` of dynamic LDS that spins for spin_us.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/overlap_probe.hip -o tools/liboverlap_probe.so
#include <hip/hip_runtime.h>
#include <cstdint>
extern __shared__ uint32_t dyn[];
__global__ void __launch_bounds__(256) probe_kernel(uint32_t* out, long long spin_cycles, int words)
{
    const long long t0 = wall_clock64();
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < words; i += 256) dyn[i] = i;
    __syncthreads();
    while (wall_clock64() - t0 < spin_cycles) acc += dyn[(acc + threadIdx.x) % (words > 0 ? words : 1)];
    if (acc == 0xdeadbeefu) out[0] = acc;
}
// The same with the register footprint of RCCL's rcclGenericKernel (261-280 registers per lane,
// profiles/r2_rccl_kernel_resources.txt): 264 live accumulators, so a wave takes more than half of a SIMD's file.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
probe_kernel_regs(uint32_t* out, long long spin_cycles, int words)
{
    const long long t0 = wall_clock64();
    for (int i = threadIdx.x; i < words; i += 256) dyn[i] = i;
    __syncthreads();
    constexpr int N = 264;
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = (float)(threadIdx.x + i);
    while (wall_clock64() - t0 < spin_cycles)
    {
        const float m = __uint_as_float(dyn[threadIdx.x % (words > 0 ? words : 1)] | 0x3f800000u);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = fmaf(acc[i], m, acc[(i + 1) % N]);
    }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) sum += acc[i];
    if (sum == 12345.678f) out[0] = 1u;
}
extern "C" int probe_launch_regs(void* stream, int lds_bytes, int blocks, int spin_us)
{
    static uint32_t* out = nullptr;
    if (!out && hipMalloc(&out, 64) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe_kernel_regs, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, out, (long long)spin_us * 100, lds_bytes / 4);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
extern "C" int probe_launch(void* stream, int lds_bytes, int blocks, int spin_us)
{
    static uint32_t* out = nullptr;
    if (!out && hipMalloc(&out, 64) != hipSuccess) return 1;
    if (lds_bytes > 64 * 1024)
        if (hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 2;
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, out, (long long)spin_us * 100, lds_bytes / 4);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
