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
extern "C" int probe_launch(void* stream, int lds_bytes, int blocks, int spin_us)
{
    static uint32_t* out = nullptr;
    if (!out && hipMalloc(&out, 64) != hipSuccess) return 1;
    if (lds_bytes > 64 * 1024)
        if (hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 2;
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, out, (long long)spin_us * 100, lds_bytes / 4);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
