// LDS atomic throughput microbenchmark (gfx950): cycles per wave-instruction per CU for
// ds_add_u32 / 2x ds_add_u32 / ds_add_u64 / ds_add_f64, 8 waves per workgroup, one WG per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* sink)
{
    __shared__ unsigned long long buf[8 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8 * 256; i += 512) buf[i] = 0;
    __syncthreads();
    unsigned int* b32 = reinterpret_cast<unsigned int*>(buf) + wave * 512;
    unsigned long long* b64 = buf + wave * 256;
    double* bf = reinterpret_cast<double*>(buf) + wave * 256;
    int v = lane + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = ((lane & 31) + 4 * (lane >> 5) + r * 3 + it) & 127;
            if (MODE == 0) atomicAdd(&b32[idx], (unsigned)v);
            if (MODE == 1) { atomicAdd(&b32[idx], (unsigned)v); atomicAdd(&b32[256 + idx], (unsigned)(v >> 3)); }
            if (MODE == 2) atomicAdd(&b64[idx], (unsigned long long)(long long)v);
            if (MODE == 3) atomicAdd(&bf[idx], (double)v);
            v = v * 3 + r;
        }
    }
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = buf[5] + buf[300];
}

template <int MODE> float run(int iters, unsigned long long* sink)
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
    unsigned long long* sink; CK(hipMalloc(&sink, 256 * 8));
    const int iters = 2000;
    const char* names[] = {"ds_add_u32", "2x ds_add_u32", "ds_add_u64", "ds_add_f64"};
    float ms[4] = {run<0>(iters, sink), run<1>(iters, sink), run<2>(iters, sink), run<3>(iters, sink)};
    for (int m = 0; m < 4; ++m) {
        double waveinstr_per_cu = 8.0 * iters * 16 * (m == 1 ? 2 : 1);
        printf("%-14s %.3f ms  -> %.2f ns per wave-instr per CU (%.1f cyc @2.4GHz); per element-emission %.2f ns\n",
               names[m], ms[m], ms[m] * 1e6 / waveinstr_per_cu, ms[m] * 1e6 / waveinstr_per_cu * 2.4,
               ms[m] * 1e6 / (8.0 * iters * 16));
    }
    return 0;
}
