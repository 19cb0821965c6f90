// Probe: which XCD (HW_REG_XCC_ID) do the one-wave workgroups of a large launch land on?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* hist, unsigned* first)
{
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
    if (threadIdx.x == 0)
    {
        atomicAdd(&hist[xcc & 15u], 1u);
        if (blockIdx.x < 32) first[blockIdx.x] = xcc;
    }
}
int main()
{
    unsigned *h, *f; hipMalloc(&h, 64); hipMalloc(&f, 128); hipMemset(h, 0, 64); hipMemset(f, 0, 128);
    hipLaunchKernelGGL(probe, dim3(4608), dim3(64), 0, 0, h, f);
    unsigned hh[16], ff[32]; hipMemcpy(hh, h, 64, hipMemcpyDeviceToHost); hipMemcpy(ff, f, 128, hipMemcpyDeviceToHost);
    printf("XCC_ID histogram of 4608 blocks:"); for (int i = 0; i < 16; ++i) printf(" %u", hh[i]); printf("\n");
    printf("blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", ff[i]); printf("\n");
    return 0;
}
