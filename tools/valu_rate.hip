// Microbenchmark: sustained issue rate of VALU instruction kinds on gfx950 (wave-instructions per
// cycle per SIMD), to price the tracer's VALU roofline. Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(64) rate_kernel(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    uint64_t w0 = u0, w1 = u1, w2 = u2, w3 = u3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f2 pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; ++i)
    {
        if (KIND == 0)
        {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
        else if (KIND == 1)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                             : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(u0), "v"(u1) : "vcc");
        }
        else if (KIND == 2)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(i));
        }
        else if (KIND == 3)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        }
        else if (KIND == 4)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(i | 3));
        }
        else if (KIND == 5)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(w0 + w1 + w2 + w3)
                                         + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
int run(const char* name, int waves_per_simd, int cus, float* d_out, double clk_ghz)
{
    const int iters = 4000;
    const int blocks = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 10);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = (double)iters * 64 * waves_per_simd;
    const double cycles = ms * 1e-3 * clk_ghz * 1e9;
    printf("%-14s waves/SIMD=%d  %.3f ms  cycles per wave-instruction per SIMD = %.2f\n", name, waves_per_simd, ms, cycles / insts_per_simd);
    return 0;
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const double clk = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, clk);
    float* d; CHECK(hipMalloc(&d, (size_t)p.multiProcessorCount * 32 * 64 * 4 * 2));
    for (int w : {1, 2, 4, 8})
    {
        if (run<0>("v_fma_f32", w, p.multiProcessorCount, d, clk)) return 1;
        if (run<1>("v_mad_u64_u32", w, p.multiProcessorCount, d, clk)) return 1;
        if (run<2>("v_xor_b32", w, p.multiProcessorCount, d, clk)) return 1;
        if (run<3>("v_pk_fma_f32", w, p.multiProcessorCount, d, clk)) return 1;
        if (run<4>("v_mul_hi_u32", w, p.multiProcessorCount, d, clk)) return 1;
        if (run<5>("v_rcp_f32", w, p.multiProcessorCount, d, clk)) return 1;
    }
    return 0;
}
