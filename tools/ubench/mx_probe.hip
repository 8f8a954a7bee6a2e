// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 / _32x32x64_ on gfx950 with A = fp4 (e2m1), B = fp6 (e2m3):
//   1. operand layout against a CPU product on exact integer data (assumed map: lane l holds row/col l%16 (l%32),
//      K-elements 32*(l/16) .. +31 (32*(l/32) .. +31), element j in bits [w*j, w*j + w) of the lane's register string);
//      a match matrix of one-hot A positions against one-hot B positions is printed when that product is wrong;
//   2. block scales: the E8M0 byte of a lane scales its own 32 K-elements;
//   3. cycles per instruction with two waves per SIMD on every CU, random operands, and the clock the chip holds;
//   4. the same loop with one ds_read_b128 per three matrix instructions feeding the A operand;
//   5. all workgroups streaming one shared L2-resident buffer into registers (24 B per lane and step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void one16(const i32x8 *a, const i32x8 *b, f32x4 *d, const int *sa, const int *sb)
{
    const int l = threadIdx.x;
    f32x4 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 4, 2, 0, sa[l], 0, sb[l]);
    d[l] = c;
}
__global__ void one32(const i32x8 *a, const i32x8 *b, f32x16 *d, const int *sa, const int *sb)
{
    const int l = threadIdx.x;
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 4, 2, 0, sa[l], 0, sb[l]);
    d[l] = c;
}
// match matrix: block = A position (quarter qa, element pa), thread-loop over B positions
__global__ void match16(float *out)
{
    const int l = threadIdx.x, ea = blockIdx.x;  // ea = qa*32 + pa
    const int qa = ea >> 5, pa = ea & 31;
    i32x8 A = {};
    if ((l >> 4) == qa) {  // fp4: nibble pa of the lane's 128-bit string = 1.0 (code 2)
        A[pa >> 3] = 2 << ((pa & 7) * 4);
    }
    for (int eb = 0; eb < 128; ++eb) {
        const int qb = eb >> 5, pb = eb & 31;
        unsigned long long lo = 0, hi = 0, top = 0;  // 192-bit string
        unsigned w[6] = {0, 0, 0, 0, 0, 0};
        if ((l >> 4) == qb) {
            const int bit = pb * 6;
            const unsigned long long v = 0x08ull;  // fp6 e2m3 1.0
            w[bit >> 5] |= (unsigned)(v << (bit & 31));
            if ((bit & 31) > 26) w[(bit >> 5) + 1] |= (unsigned)(v >> (32 - (bit & 31)));
        }
        (void)lo; (void)hi; (void)top;
        i32x8 B = {(int)w[0], (int)w[1], (int)w[2], (int)w[3], (int)w[4], (int)w[5], 0, 0};
        f32x4 c = {};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        if (l == 0) out[ea * 128 + eb] = c[0];
    }
}

template <int SHAPE, int LDSREAD>
__global__ __launch_bounds__(512, 2) void rate(int steps, unsigned long long *out, float *sink, const i32x8 *ain, const i32x8 *bin)
{
    __shared__ i32x4 img[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 512) img[i] = i32x4{ain[i & 63][0], ain[i & 63][1], ain[(i + 7) & 63][2], ain[i & 63][3]};
    __syncthreads();
    i32x8 A = ain[lane], B0 = bin[lane], B1 = bin[64 + lane], B2 = bin[128 + lane];
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    if (SHAPE == 16) {
        f32x4 acc[6] = {};
        i32x4 q = img[lane + 64 * wave];
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (LDSREAD) {
                    A[0] = q[0]; A[1] = q[1]; A[2] = q[2]; A[3] = q[3];
                    q = img[(lane + 64 * ((s * 6 + j + wave) & 31))];
                }
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B0, acc[j], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B1, acc[j], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B2, acc[j], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
        float x = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) x += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        if (x == 1.2345f) sink[0] = x;
    } else {
        f32x16 acc[3] = {};
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B0, acc[j], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B1, acc[j], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B2, acc[j], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
        float x = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) x += acc[j][i];
        if (x == 1.2345f) sink[0] = x;
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (lane == 0) {
        out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }
}

// every workgroup streams the same buffer: wave w takes chunks w, w+8, ... of 1536 B (16 B + 8 B per lane)
__global__ __launch_bounds__(512, 2) void l2stream(const char *buf, int chunks, int rounds, unsigned long long *out, int *sink)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long t0, t1;
    int x = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < rounds; ++r) {
        for (int c = wave; c < chunks; c += 8 * 4) {
            i32x4 v[4];
            i32x2 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = c + 8 * j;
                if (cc < chunks) {
                    v[j] = *reinterpret_cast<const i32x4 *>(buf + (size_t)cc * 1536 + lane * 16);
                    u[j] = *reinterpret_cast<const i32x2 *>(buf + (size_t)cc * 1536 + 1024 + lane * 8);
                } else {
                    v[j] = i32x4{0, 0, 0, 0};
                    u[j] = i32x2{0, 0};
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) x += v[j][0] ^ v[j][3] ^ u[j][1];
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (x == 0x1234567) sink[0] = x;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

static unsigned rnd_state = 12345;
static unsigned rnd() { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }

int main()
{
    // ---- 1. layout ----
    for (int shape = 16; shape <= 32; shape += 16) {
        const int R = shape, K = shape == 16 ? 128 : 64, KQ = 32;
        std::vector<int> Am(R * K), Bm(K * R);
        for (auto &v : Am) v = rnd() % 4 == 0;                  // {0,1}
        for (auto &v : Bm) v = (int)(rnd() % 31) - 15;          // -15..15 (units of 1/8)
        std::vector<unsigned> a(64 * 8, 0), b(64 * 8, 0);
        std::vector<int> sa(64), sb(64);
        for (int l = 0; l < 64; ++l) {
            const int r = l % R, q = l / R;
            for (int j = 0; j < KQ; ++j) {
                const int k = q * KQ + j;
                if (Am[r * K + k]) a[l * 8 + (j >> 3)] |= 2u << ((j & 7) * 4);
                const int dgt = Bm[k * R + r];
                const unsigned code = (unsigned)(dgt < 0 ? -dgt : dgt) | (dgt < 0 ? 0x20u : 0u);
                const int bit = j * 6;
                b[l * 8 + (bit >> 5)] |= code << (bit & 31);
                if ((bit & 31) > 26) b[l * 8 + (bit >> 5) + 1] |= code >> (32 - (bit & 31));
            }
            sa[l] = 0x7f7f7f7f;
            sb[l] = 0x7f7f7f7f + (l / R) * 0x01010101;  // quarter q of B scaled by 2^q
        }
        unsigned *da, *db;
        float *dd;
        int *dsa, *dsb;
        CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dd, 64 * 64));
        CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
        CK(hipMemcpy(da, a.data(), 64 * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), 64 * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        std::vector<float> d(64 * 16);
        if (shape == 16) hipLaunchKernelGGL(one16, dim3(1), dim3(64), 0, 0, (i32x8 *)da, (i32x8 *)db, (f32x4 *)dd, dsa, dsb);
        else hipLaunchKernelGGL(one32, dim3(1), dim3(64), 0, 0, (i32x8 *)da, (i32x8 *)db, (f32x16 *)dd, dsa, dsb);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(d.data(), dd, (shape == 16 ? 4 : 16) * 64 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        double worst = 0;
        for (int l = 0; l < 64; ++l)
            for (int g = 0; g < (shape == 16 ? 4 : 16); ++g) {
                const int col = l % R;
                const int row = shape == 16 ? (l >> 4) * 4 + g : (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += Am[row * K + k] * (Bm[k * R + col] / 8.0) * (double)(1 << (k / KQ));
                const double got = d[l * (shape == 16 ? 4 : 16) + g];
                if (got != ref) { ++bad; if (bad < 6) printf("  shape %d lane %d reg %d: got %g want %g\n", shape, l, g, got, ref); }
                worst = worst > (got > ref ? got - ref : ref - got) ? worst : (got > ref ? got - ref : ref - got);
            }
        printf("layout %dx%d: %d mismatches (assumed map, B quarter q scaled by 2^q), worst %g\n", shape, shape, bad, worst);
    }
    {
        float *dm;
        CK(hipMalloc(&dm, 128 * 128 * 4));
        hipLaunchKernelGGL(match16, dim3(128), dim3(64), 0, 0, dm);
        CK(hipDeviceSynchronize());
        std::vector<float> m(128 * 128);
        CK(hipMemcpy(m.data(), dm, 128 * 128 * 4, hipMemcpyDeviceToHost));
        int off = 0, diag = 0;
        for (int i = 0; i < 128; ++i)
            for (int j = 0; j < 128; ++j) {
                if (i == j) diag += m[i * 128 + j] == 1.0f;
                else if (m[i * 128 + j] != 0) { if (++off < 10) printf("  match A(%d,%d) B(%d,%d) = %g\n", i >> 5, i & 31, j >> 5, j & 31, m[i * 128 + j]); }
            }
        printf("match matrix 16x16x128: %d of 128 diagonal entries are 1.0, %d off-diagonal non-zero\n", diag, off);
    }
    // ---- 3/4. rates ----
    {
        std::vector<unsigned> a(64 * 8), b(192 * 8);
        for (auto &v : a) v = rnd() & 0x11111111u & (rnd() << 8);          // sparse 0/0.5 nibbles
        for (size_t i = 0; i < b.size(); ++i) b[i] = (i % 8) < 6 ? (rnd() << 8) ^ rnd() : 0;
        unsigned *da, *db;
        unsigned long long *dout;
        float *dsink;
        CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dout, 256 * 8 * 16)); CK(hipMalloc(&dsink, 16));
        CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        std::vector<unsigned long long> h(256 * 8 * 2);
        for (int variant = 0; variant < 3; ++variant) {
            const int steps = 3000;
            for (int rep = 0; rep < 3; ++rep) {
                if (variant == 0) hipLaunchKernelGGL((rate<16, 0>), dim3(256), dim3(512), 0, 0, steps, dout, dsink, (i32x8 *)da, (i32x8 *)db);
                if (variant == 1) hipLaunchKernelGGL((rate<16, 1>), dim3(256), dim3(512), 0, 0, steps, dout, dsink, (i32x8 *)da, (i32x8 *)db);
                if (variant == 2) hipLaunchKernelGGL((rate<32, 0>), dim3(256), dim3(512), 0, 0, steps, dout, dsink, (i32x8 *)da, (i32x8 *)db);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < 256 * 8; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
            const double per = variant == 2 ? 9.0 : 18.0;
            printf("%s: %.2f cycles per instruction per wave (two waves per SIMD => %.2f per SIMD), clock %.3f GHz\n",
                   variant == 0 ? "16x16x128 fp4 x fp6, operands in registers" : variant == 1 ? "16x16x128 fp4 x fp6, A by ds_read_b128 every 3rd" : "32x32x64 fp4 x fp6, operands in registers",
                   cyc / (256 * 8) / (steps * per), cyc / (256 * 8) / (steps * per) / 2, cyc / real / 10.0);
        }
    }
    // ---- 5. L2 stream ----
    {
        const int chunks = 297;  // 99 K-blocks x 3 limbs x 1536 B = 456 KB
        char *buf;
        unsigned long long *dout;
        int *dsink;
        CK(hipMalloc(&buf, (size_t)chunks * 1536)); CK(hipMemset(buf, 1, (size_t)chunks * 1536));
        CK(hipMalloc(&dout, 256 * 8 * 8)); CK(hipMalloc(&dsink, 16));
        for (int rounds = 1; rounds <= 4; rounds *= 4) {
            hipLaunchKernelGGL(l2stream, dim3(256), dim3(512), 0, 0, buf, chunks, rounds, dout, dsink);
            hipLaunchKernelGGL(l2stream, dim3(256), dim3(512), 0, 0, buf, chunks, rounds, dout, dsink);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(256 * 8);
            CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
            double cyc = 0, mx = 0;
            for (auto v : h) { cyc += v; mx = mx > v ? mx : v; }
            printf("L2 stream: %d x %d KB per workgroup in %.0f cycles (mean over waves; max %.0f) = %.1f B/clk per CU\n", rounds, chunks * 1536 / 1024,
                   cyc / h.size(), mx, rounds * chunks * 1536.0 / (cyc / h.size()));
        }
    }
    return 0;
}
