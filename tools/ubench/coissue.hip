// How much non-matrix work fits beside v_mfma_i32_32x32x32_i8 on one SIMD of gfx950?
//   MODE 0: every wave runs {1 MFMA + K independent VALU adds} per step             (K = template F)
//   MODE 1: waves 0..W/2-1 run pure MFMAs, waves W/2.. run pure VALU (F adds per step)
//   MODE 2: as 0, fillers = F/2 x (v_lshl_add + ds_add_u32) pairs
//   MODE 3: as 1, partner runs (v_lshl_add + ds_add_u32) pairs
//   MODE 4: as 2, plus one ds_read_b128 per step whose result is the MFMAs' A operand two steps later (an operand queue)
// WAVES = 4 (one per SIMD) or 8 (two per SIMD: w and w+4 share a SIMD).  Prints cycles per step per wave class.
#include <hip/hip_runtime.h>
#include <cstdio>
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int F, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(int steps, unsigned long long *out, int *sink)
{
    __shared__ unsigned ring[8 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8 * 512; i += WAVES * 64) ring[i] = 0;
    __syncthreads();
    i32x4 a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, 1, 2, 3};
    i32x16 acc0 = {}, acc1 = {};
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    unsigned *cell = ring + wave * 512 + lane;
    const bool mfma_wave = (MODE == 0 || MODE == 2 || MODE == 4) ? true : wave < WAVES / 2;
    const bool fill_wave = (MODE == 0 || MODE == 2 || MODE == 4) ? true : wave >= WAVES / 2;
    i32x4 q0 = a, q1 = a;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int s = 0; s < steps; ++s) {
        if (MODE == 4) {  // operand queue: the fragment read two steps ago feeds this step
            a = q0;
            q0 = q1;
            q1 = *reinterpret_cast<const i32x4 *>(ring + ((s * 64 + lane * 4) & 2047));
        }
        if (mfma_wave) {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
        }
        if (fill_wave) {
            if (MODE == 0 || MODE == 1) {  // (MODE 4 takes the pairs below)
#pragma unroll
                for (int i = 0; i < F; ++i) v[i & 7] = v[i & 7] * 3 + s;  // independent chains (v_mad / lshl_add)
            } else {
#pragma unroll
                for (int i = 0; i < F / 2; ++i) {
                    v[i & 7] = (v[i & 7] << 8) + s;
                    __hip_atomic_fetch_add(cell + 64 * (i & 3), (unsigned)v[i & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (mfma_wave) {
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    int x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x += acc0[i] + acc1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) x += v[i];
    if (x == 0x12345678) sink[0] = x + ring[lane];
    if (lane == 0) out[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int MODE, int F, int WAVES> int run(const char *what, unsigned long long *d_out, int *d_sink)
{
    const int steps = 4000, grid = 256;
    static unsigned long long h[256 * 8];
    hipLaunchKernelGGL((k<MODE, F, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, steps, d_out, d_sink);
    hipLaunchKernelGGL((k<MODE, F, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, steps, d_out, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d_out, sizeof(unsigned long long) * grid * WAVES, hipMemcpyDeviceToHost));
    double lo = 0, hi = 0;
    for (int g = 0; g < grid; ++g)
        for (int w = 0; w < WAVES; ++w) (w < WAVES / 2 ? lo : hi) += (double)h[g * WAVES + w];
    lo /= grid * (WAVES / 2) * (double)steps;
    hi /= grid * (WAVES / 2) * (double)steps;
    printf("%-46s waves/SIMD %d  F=%2d  cycles/step: waves<half %.1f  waves>=half %.1f   (a step = 2 MFMAs = 64 matrix cycles)\n", what,
           WAVES / 4, F, lo, hi);
    return 0;
}

int main()
{
    unsigned long long *d_out;
    int *d_sink;
    CK(hipMalloc((void **)&d_out, 256 * 8 * 8));
    CK(hipMalloc((void **)&d_sink, 64));
    run<0, 0, 4>("MFMA only", d_out, d_sink);
    run<0, 4, 4>("2 MFMA + F valu, same wave", d_out, d_sink);
    run<0, 8, 4>("2 MFMA + F valu, same wave", d_out, d_sink);
    run<0, 12, 4>("2 MFMA + F valu, same wave", d_out, d_sink);
    run<0, 16, 4>("2 MFMA + F valu, same wave", d_out, d_sink);
    run<0, 0, 8>("MFMA only", d_out, d_sink);
    run<0, 4, 8>("2 MFMA + F valu, every wave", d_out, d_sink);
    run<0, 8, 8>("2 MFMA + F valu, every wave", d_out, d_sink);
    run<0, 12, 8>("2 MFMA + F valu, every wave", d_out, d_sink);
    run<0, 16, 8>("2 MFMA + F valu, every wave", d_out, d_sink);
    run<1, 4, 8>("w<4: 2 MFMA | w>=4: F valu", d_out, d_sink);
    run<1, 8, 8>("w<4: 2 MFMA | w>=4: F valu", d_out, d_sink);
    run<1, 16, 8>("w<4: 2 MFMA | w>=4: F valu", d_out, d_sink);
    run<2, 4, 4>("2 MFMA + F/2 (lshl_add, ds_add), same wave", d_out, d_sink);
    run<2, 8, 4>("2 MFMA + F/2 (lshl_add, ds_add), same wave", d_out, d_sink);
    run<2, 4, 8>("2 MFMA + F/2 (lshl_add, ds_add), every wave", d_out, d_sink);
    run<2, 8, 8>("2 MFMA + F/2 (lshl_add, ds_add), every wave", d_out, d_sink);
    run<4, 2, 4>("2 MFMA + ds_read_b128 + F/2 (lshl_add, ds_add), same wave", d_out, d_sink);
    run<4, 4, 4>("2 MFMA + ds_read_b128 + F/2 (lshl_add, ds_add), same wave", d_out, d_sink);
    run<4, 6, 4>("2 MFMA + ds_read_b128 + F/2 (lshl_add, ds_add), same wave", d_out, d_sink);
    run<4, 4, 8>("2 MFMA + ds_read_b128 + F/2 (lshl_add, ds_add), every wave", d_out, d_sink);
    run<3, 8, 8>("w<4: 2 MFMA | w>=4: F/2 (lshl_add, ds_add)", d_out, d_sink);
    run<3, 16, 8>("w<4: 2 MFMA | w>=4: F/2 (lshl_add, ds_add)", d_out, d_sink);
    return 0;
}
