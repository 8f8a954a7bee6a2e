// LDS atomic cost vs the address pattern of one wave-instruction (gfx950): cycles per ds_add_u32
// wave-instruction per CU, 8 waves per workgroup, one WG per CU.
//   0: 64 distinct cells   1: 32x32 skew pattern (lane&31) + 4 (lane>>5): pairs of lanes share a cell
//   2: 16x16 skew pattern 4 (lane>>4) - (lane&15): up to four lanes share a cell   3: eight lanes per cell
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, unsigned* sink)
{
    __shared__ unsigned buf[8 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8 * 512; i += 512) buf[i] = 0;
    __syncthreads();
    unsigned* b = buf + wave * 512 + 64;
    int base;
    if (MODE == 0) base = lane;
    if (MODE == 1) base = (lane & 31) + 4 * (lane >> 5);
    if (MODE == 2) base = 4 * (lane >> 4) - (lane & 15);
    if (MODE == 3) base = lane & 7;
    int v = lane + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            atomicAdd(&b[base + ((r * 5 + it) & 63)], (unsigned)v);
            v = v * 3 + r;
        }
    }
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = buf[70] + buf[300];
}

template <int MODE> float run(int iters, unsigned* sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    unsigned* sink; CK(hipMalloc((void**)&sink, 256 * 4));
    const int iters = 2000;
    const char* names[] = {"distinct", "32x32 skew (2 lanes/cell)", "16x16 skew (<=4 lanes/cell)", "8 lanes/cell"};
    float ms[4] = {run<0>(iters, sink), run<1>(iters, sink), run<2>(iters, sink), run<3>(iters, sink)};
    for (int m = 0; m < 4; ++m)
        printf("%-28s %.3f ms -> %.1f cycles per wave-instruction per CU @2.4GHz\n", names[m], ms[m],
               ms[m] * 1e6 / (8.0 * iters * 16) * 2.4);
    return 0;
}
