// Issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 (fp4 x fp6) per wave and per SIMD on gfx950:
//   W waves per SIMD (1..4), scale operands as block scales (S = 1) or the constant 0 (S = 0: does the compiler pick the
//   unscaled encoding, and is that one faster?), accumulations dependent in threes (D = 1) or spread over 9 accumulators (D = 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W, int S, int D>
__global__ __launch_bounds__(W * 256, W) void rate(int steps, unsigned long long *out, float *sink, const i32x8 *ain, const i32x8 *bin)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    i32x8 A = ain[lane], B0 = bin[lane], B1 = bin[64 + lane], B2 = bin[128 + lane];
    f32x4 acc[9] = {};
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int s = 0; s < steps; ++s) {
        if (D) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B0, acc[j], 4, 2, 0, S ? 0x7f7f7f7f : 0, 0, S ? (int)0x82828282u : 0);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B1, acc[j], 4, 2, 0, S ? 0x7f7f7f7f : 0, 0, S ? (int)0x86868686u : 0);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B2, acc[j], 4, 2, 0, S ? 0x7f7f7f7f : 0, 0, S ? (int)0x8a8a8a8au : 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 9; ++j)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, j % 3 == 0 ? B0 : j % 3 == 1 ? B1 : B2, acc[j], 4, 2, 0, S ? 0x7f7f7f7f : 0, 0,
                                                                          S ? (int)0x82828282u : 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float x = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) x += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (x == 1.2345f) sink[0] = x;
    if (lane == 0) {
        out[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0;
    }
}

// the K loop of csrc/detect_mx.hip without its memory traffic: NT accumulators, three tiles x three limbs interleaved, the A
// operand from a ring of seven register sets, two B sets alternating per K-block
template <int W, int NT>
__global__ __launch_bounds__(W * 256, W) void loop_like(int steps, unsigned long long *out, float *sink, const i32x8 *ain, const i32x8 *bin)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    i32x8 A[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { A[i] = ain[(lane + i) & 63]; A[i][4] = A[i][5] = A[i][6] = A[i][7] = 0; }
    i32x8 B[3] = {bin[lane], bin[64 + lane], bin[128 + lane]};
    f32x4 acc[NT] = {};
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int j = 0; j < NT; j += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
                acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[(j + u) % 7], B[0], acc[j + u], 4, 2, 0, 0x7f7f7f7f, 0, (int)0x82828282u);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[(j + u) % 7], B[1], acc[j + u], 4, 2, 0, 0x7f7f7f7f, 0, (int)0x86868686u);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[(j + u) % 7], B[2], acc[j + u], 4, 2, 0, 0x7f7f7f7f, 0, (int)0x8a8a8a8au);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float x = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) x += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (x == 1.2345f) sink[0] = x;
    if (lane == 0) {
        out[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0;
    }
}

template <int W, int NT>
__global__ __launch_bounds__(W * 256, W) void loop_like_t(int steps, unsigned long long *out, float *sink, const i32x8 *ain, const i32x8 *bin)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    i32x8 A[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { A[i] = ain[(lane + i) & 63]; A[i][4] = A[i][5] = A[i][6] = A[i][7] = 0; }
    i32x8 B[3] = {bin[lane], bin[64 + lane], bin[128 + lane]};
    f32x4 acc[NT] = {};
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int j = 0; j < NT; j += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
                acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B[0], A[(j + u) % 7], acc[j + u], 2, 4, 0, (int)0x82828282u, 0, 0x7f7f7f7f);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B[1], A[(j + u) % 7], acc[j + u], 2, 4, 0, (int)0x86868686u, 0, 0x7f7f7f7f);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B[2], A[(j + u) % 7], acc[j + u], 2, 4, 0, (int)0x8a8a8a8au, 0, 0x7f7f7f7f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float x = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) x += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (x == 1.2345f) sink[0] = x;
    if (lane == 0) {
        out[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0;
    }
}

template <int W, int NT> int run_like(unsigned long long *dout, float *dsink, const i32x8 *da, const i32x8 *db)
{
    const int steps = 400;
    CK(hipMemset(dout, 0, 256 * 16 * 16));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop_like<W, NT>), dim3(256), dim3(W * 256), 0, 0, steps, dout, dsink, da, db);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 16 * 2);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    int n = 0;
    for (int i = 0; i < 256 * 16; ++i)
        if (h[2 * i]) { cyc += h[2 * i]; real += h[2 * i + 1]; ++n; }
    const double per = cyc / n / (steps * 3.0 * NT);
    double lo = 1e30, hi = 0;
    for (int i = 0; i < 256 * 16; ++i)
        if (h[2 * i]) { lo = lo < (double)h[2 * i] ? lo : (double)h[2 * i]; hi = hi > (double)h[2 * i] ? hi : (double)h[2 * i]; }
    printf("%d wave(s) per SIMD, loop-like, %d accumulators: %.2f cycles per instruction per wave (fastest wave %.2f, slowest %.2f: the job ends at %.2f per SIMD), clock %.3f GHz\n",
           W, NT, per, lo / (steps * 3.0 * NT), hi / (steps * 3.0 * NT), hi / (steps * 3.0 * NT) / W, cyc / real / 10.0);
    return 0;
}

template <int W, int NT> int run_like_t(unsigned long long *dout, float *dsink, const i32x8 *da, const i32x8 *db)
{
    const int steps = 400;
    CK(hipMemset(dout, 0, 256 * 16 * 16));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop_like_t<W, NT>), dim3(256), dim3(W * 256), 0, 0, steps, dout, dsink, da, db);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 16 * 2);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    int n = 0;
    for (int i = 0; i < 256 * 16; ++i)
        if (h[2 * i]) { cyc += h[2 * i]; real += h[2 * i + 1]; ++n; }
    const double per = cyc / n / (steps * 3.0 * NT);
    printf("%d wave(s) per SIMD, loop-like with swapped operand roles, %d accumulators: %.2f cycles per instruction per wave = %.2f per SIMD, clock %.3f GHz\n", W, NT, per, per / W,
           cyc / real / 10.0);
    return 0;
}


// The K loop of csrc/detect_mx.hip WITH its operand feeds, to price workgroup shapes before building them: W waves per SIMD,
// NT tiles per wave and K-block; per K-block 3 NT matrix instructions, NT ds_read_b128 (+ one address add each) into a ring
// seven deep, six global loads of the next K-block's weight fragments (an L2-resident buffer all workgroups share).
template <int W, int NT>
__global__ __launch_bounds__(W * 256, W) void loop_fed(int kblocks, unsigned long long *out, float *sink, const unsigned char *wbuf, int wblocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 40960 / 16; i += W * 256) reinterpret_cast<i32x4 *>(lds)[i] = i32x4{0x02002000, 0x00020200, 0x20000002, 0x00202000};
    __syncthreads();
    int tb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tb[j] = ((j * 16 + (lane & 15)) + 640 * (lane >> 4)) << 4;   // column-major image, 640 rows per array
    i32x4 A[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) A[j] = *reinterpret_cast<const i32x4 *>(lds + tb[j % NT]);
    const unsigned char *wp = wbuf + (size_t)((wave * 7) % wblocks) * 4608 + lane * 16;
    i32x4 c4[3], n4[3];
    i32x2 c2[3], n2[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        c4[l] = n4[l] = *reinterpret_cast<const i32x4 *>(wp + l * 1536);
        c2[l] = n2[l] = *reinterpret_cast<const i32x2 *>(wp + l * 1536 + 1024 - lane * 8);
    }
    f32x4 acc[NT] = {};
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int kb = 0; kb < kblocks; ++kb) {
        const int koff = (kb & 15) << 4;
        const unsigned char *nb = wbuf + (size_t)((kb * 8 + wave) % wblocks) * 4608 + lane * 16;
        i32x8 B[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) B[l] = i32x8{c4[l][0], c4[l][1], c4[l][2], c4[l][3], c2[l][0], c2[l][1], 0, 0};
        constexpr int NG = (NT + 2) / 3;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g < 6) {
                if (g & 1) n2[g / 2] = *reinterpret_cast<const i32x2 *>(nb + (g / 2) * 1536 + 1024 - lane * 8);
                else n4[g / 2] = *reinterpret_cast<const i32x4 *>(nb + (g / 2) * 1536);
            }
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (3 * g + u < NT) {
                        const i32x4 x = A[(3 * g + u) % 7];
                        acc[3 * g + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B[l], i32x8{x[0], x[1], x[2], x[3], 0, 0, 0, 0}, acc[3 * g + u], 2, 4, 0,
                                                                                          l == 0 ? (int)0x82828282u : l == 1 ? (int)0x86868686u : (int)0x8a8a8a8au, 0, 0x7f7f7f7f);
                    }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (3 * g + u < NT) A[(3 * g + u) % 7] = *reinterpret_cast<const i32x4 *>(lds + tb[(3 * g + u + 7) % NT] + koff);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) { c4[l] = n4[l]; c2[l] = n2[l]; }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float x = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) x += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (x == 1.2345f) sink[0] = x;
    if (lane == 0) {
        out[(blockIdx.x * 16 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0;
    }
}

template <int W, int NT> int run_fed(unsigned long long *dout, float *dsink, const unsigned char *wbuf, int kblocks)
{
    CK(hipMemset(dout, 0, 256 * 16 * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(loop_fed<W, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop_fed<W, NT>), dim3(256), dim3(W * 256), 40960, 0, kblocks, dout, dsink, wbuf, 93);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 16 * 2);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, hi = 0;
    int n = 0;
    for (int i = 0; i < 256 * 16; ++i)
        if (h[2 * i]) { cyc += h[2 * i]; hi = hi > (double)h[2 * i] ? hi : (double)h[2 * i]; ++n; }
    const double mf = kblocks * 3.0 * NT;
    printf("fed loop, %d wave(s) per SIMD x %d tiles: mean %.2f cycles per instruction per wave, the slowest wave %.2f => the job ends at %.2f per SIMD\n", W, NT,
           cyc / n / mf, hi / mf, hi / mf / W);
    return 0;
}

template <int W, int S, int D> int run(unsigned long long *dout, float *dsink, const i32x8 *da, const i32x8 *db)
{
    const int steps = 2000;
    CK(hipMemset(dout, 0, 256 * 16 * 16));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((rate<W, S, D>), dim3(256), dim3(W * 256), 0, 0, steps, dout, dsink, da, db);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 16 * 2);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    int n = 0;
    for (int i = 0; i < 256 * 16; ++i)
        if (h[2 * i]) { cyc += h[2 * i]; real += h[2 * i + 1]; ++n; }
    const double per = cyc / n / (steps * 9.0);
    printf("%d wave(s) per SIMD, %s, %s: %.2f cycles per instruction per wave = %.2f per SIMD, clock %.3f GHz\n", W, S ? "block scales" : "scale operands 0",
           D ? "dependent in threes" : "9 accumulators", per, per / W, cyc / real / 10.0);
    return 0;
}

int main()
{
    std::vector<unsigned> a(64 * 8), b(192 * 8);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (auto &v : a) v = rnd() & 0x22222222u & (rnd() << 8);
    for (size_t i = 0; i < b.size(); ++i) b[i] = (i % 8) < 6 ? ((rnd() << 8) ^ rnd()) & 0x9e79e79eu : 0;   // small digits
    unsigned *da, *db;
    unsigned long long *dout;
    float *dsink;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dout, 256 * 16 * 16)); CK(hipMalloc(&dsink, 16));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    {   // does the instruction's time depend on the operand values?  The same loop on three operand sets.
        std::vector<unsigned> a2(64 * 8, 0), b2(192 * 8, 0), b3(192 * 8, 0);
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 32; ++e)
                if (rnd() % 100 < 22) a2[l * 8 + (e >> 3)] |= 2u << ((e & 7) * 4);          // binary frames, 22 % set, as fp4 1.0
        for (int f = 0; f < 192; ++f)
            for (int e = 0; e < 32; ++e) {
                const int dgt = (int)(rnd() % 17) - 8;                                       // balanced digits -8 .. 8 (sign-magnitude fp6)
                const unsigned code = (unsigned)(dgt < 0 ? -dgt : dgt) | (dgt < 0 ? 0x20u : 0u);
                const int bit = e * 6;
                b2[f * 8 + (bit >> 5)] |= code << (bit & 31);
                if ((bit & 31) > 26) b2[f * 8 + (bit >> 5) + 1] |= code >> (32 - (bit & 31));
            }
        for (size_t i = 0; i < b3.size(); ++i) b3[i] = (i % 8) < 6 ? (rnd() << 8) ^ rnd() : 0;   // random bits in every field
        unsigned *da2, *db2, *db3;
        CK(hipMalloc(&da2, a2.size() * 4)); CK(hipMalloc(&db2, b2.size() * 4)); CK(hipMalloc(&db3, b3.size() * 4));
        CK(hipMemcpy(da2, a2.data(), a2.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db3, b3.data(), b3.size() * 4, hipMemcpyHostToDevice));
        printf("operand sets -- masked sparse (the rows below):\n");
        run_like<2, 21>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
        printf("more accumulators (more registers per wave):\n");
        run_like<2, 42>(dout, dsink, (i32x8 *)da2, (i32x8 *)db2);
        run_like<2, 48>(dout, dsink, (i32x8 *)da2, (i32x8 *)db2);
        printf("operand sets -- like the kernel's (22 %% of the fp4 frames set, balanced fp6 digits -8 .. 8):\n");
        run_like<2, 21>(dout, dsink, (i32x8 *)da2, (i32x8 *)db2);
        printf("operand sets -- random bits in every fp6 field, kernel-like frames:\n");
        run_like<2, 21>(dout, dsink, (i32x8 *)da2, (i32x8 *)db3);
        printf("operand roles swapped (A = fp6 weights, B = fp4 frames), kernel-like operands:\n");
        run_like_t<2, 21>(dout, dsink, (i32x8 *)db2, (i32x8 *)da2);
        run_like_t<1, 21>(dout, dsink, (i32x8 *)db2, (i32x8 *)da2);
        printf("operand sets -- all zero:\n");
        CK(hipMemset(da2, 0, a2.size() * 4)); CK(hipMemset(db2, 0, b2.size() * 4));
        run_like<2, 21>(dout, dsink, (i32x8 *)da2, (i32x8 *)db2);
    }
    {   // workgroup shapes with the operand feeds: total matrix work per SIMD is the same (W x NT x K-blocks constant)
        unsigned char *wbuf;
        CK(hipMalloc(&wbuf, 93 * 4608));
        CK(hipMemset(wbuf, 0x11, 93 * 4608));
        run_fed<2, 21>(dout, dsink, wbuf, 24);    // 8 waves: K split 8 ways, all 21 tiles (the committed kernel)
        run_fed<3, 11>(dout, dsink, wbuf, 31);    // 12 waves: K split 6 ways x 2 tile halves
        run_fed<3, 7>(dout, dsink, wbuf, 48);     // 12 waves: K split 4 ways x 3 tile thirds
        run_fed<4, 6>(dout, dsink, wbuf, 42);     // 16 waves: K split 4 ways x 4 tile quarters (5.25 tiles)
    }
    run<1, 1, 0>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run<1, 1, 1>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run<1, 0, 0>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run<2, 1, 0>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run<2, 1, 1>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run<2, 0, 0>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run_like<2, 9>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run_like<2, 21>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run_like<1, 21>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run<3, 1, 0>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    run<4, 1, 0>(dout, dsink, (i32x8 *)da, (i32x8 *)db);
    return 0;
}
