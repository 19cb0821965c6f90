// Probe: what SQ_LDS_BANK_CONFLICT counts for the LDS access shapes of the pool kernel. One wave per workgroup
// (as the tracer), each kernel repeats ONE access shape; run under
//   rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -- tools/lds_probe
// and divide the counters per kernel. Shapes: 128-bit reads of 64 consecutive 16-byte records (no two lanes share a
// bank in any pass), 128-bit reads through a gappy ascending slot list (what a compacted round reads), 32-bit reads
// of consecutive words, byte writes of 64 consecutive bytes (the selection list), ds_bpermute.
// Build: hipcc --offload-arch=gfx950 -O2 tools/lds_probe.hip -o tools/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NS = 128;

template <int SHAPE>
__global__ void __launch_bounds__(64) probe(const uint32_t* __restrict__ slots, float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[NS * 12 + 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < NS * 12 + 16; i += 64) lds[i] = (uint32_t)i;
    __syncthreads();
    const uint32_t slot = SHAPE == 1 ? slots[blockIdx.x % 16 * 64 + lane] : (uint32_t)lane;
    float acc = 0.0f;
    uint32_t v = (uint32_t)lane;
    for (int it = 0; it < iters; ++it)
    {
        if (SHAPE == 0 || SHAPE == 1)
        {
            const float4 r = reinterpret_cast<const float4*>(lds)[slot + (it & 1) * NS];
            acc += r.x + r.w;
        }
        else if (SHAPE == 2)
            acc += __uint_as_float(lds[lane + (it & 63)]);
        else if (SHAPE == 3)
            reinterpret_cast<uint8_t*>(lds + NS * 12)[lane] = (uint8_t)(it + lane);
        else
            v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((v + 7u) & 63u) << 2), (int)v);
        asm volatile("" ::: "memory");
    }
    if (acc == 123.456f || v == 0xffffffffu) out[0] = acc;
}

int main()
{
    // gappy ascending slot lists: 64 of 128 slots, drawn with a fixed generator (16 different lists)
    uint32_t h[16 * 64];
    uint32_t s = 12345u;
    for (int l = 0; l < 16; ++l)
    {
        int n = 0;
        for (int slot = 0; slot < NS && n < 64; ++slot)
        {
            s = s * 1664525u + 1013904223u;
            const int left = NS - slot, need = 64 - n;
            if ((s >> 8) % left < (uint32_t)need) h[l * 64 + n++] = slot;
        }
    }
    uint32_t* d; float* o;
    CHECK(hipMalloc(&d, sizeof(h))); CHECK(hipMalloc(&o, 64));
    CHECK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
    const int iters = 4096, blocks = 4096;
    hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64), 0, 0, d, o, iters);
    hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64), 0, 0, d, o, iters);
    hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(64), 0, 0, d, o, iters);
    hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(64), 0, 0, d, o, iters);
    hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(64), 0, 0, d, o, iters);
    CHECK(hipDeviceSynchronize());
    printf("lds_probe: 5 shapes x %d blocks x %d iterations done\n", blocks, iters);
    return 0;
}
