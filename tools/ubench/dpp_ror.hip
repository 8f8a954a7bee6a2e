// What does DPP row_ror:n deliver?  Prints, for n = 1, 5, 15, which source lane each lane of the first
// 16-lane row (and lane 16, 48) reads.  Needed by the in-register skew reduction of detect_mfma_kernel.inc.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __global__ void k(int *out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, CTRL, 0xf, 0xf, false);
}
template <int CTRL> void run(const char *name, int *d)
{
    int h[64];
    hipLaunchKernelGGL(k<CTRL>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-10s:", name);
    for (int i = 0; i < 18; ++i) printf(" %d", h[i]);
    printf(" ... lane48<-%d\n", h[48]);
}
int main()
{
    int *d;
    if (hipMalloc((void **)&d, 256) != hipSuccess) return 1;
    run<0x121>("row_ror:1", d);
    run<0x125>("row_ror:5", d);
    run<0x12F>("row_ror:15", d);
    run<0x111>("row_shr:1", d);
    return 0;
}
