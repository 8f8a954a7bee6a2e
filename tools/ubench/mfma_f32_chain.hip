// Is v_mfma_f32_16x16x4_f32 an exact fmaf chain in ascending k?  D[i][j] = C[i][j] + sum_k A[i][k] B[k][j].
// Compare, bit for bit, with c = fmaf(a_k, b_k, c) for k = 0..3 (and with other orders) on random data,
// chained over many K-steps.  hipcc --offload-arch=gfx950 -O2 mfma_f32_chain.hip -o mfma_f32_chain.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, int KS, float *D)
{
    // A: [16][4*KS], B: [4*KS][16]; lane l: row/col = l % 16, k = l / 16
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < KS; ++s) {
        const float a = A[(l % 16) * 4 * KS + 4 * s + l / 16];
        const float b = B[(4 * s + l / 16) * 16 + l % 16];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    // C layout: lane l holds rows 4*(l/16) + r, column l % 16
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = acc[r];
}
int main()
{
    const int KS = 512;
    std::vector<float> A(16 * 4 * KS), B(4 * KS * 16), D(256);
    srand(1);
    for (auto &x : A) x = (rand() % 3 == 0) ? 0.0f : 1.0f;             // binary features
    for (auto &x : B) x = ((float)rand() / RAND_MAX - 0.5f) * 9.0f;     // log-odds-like weights
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, KS, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad_seq = 0, bad_rev = 0, bad_pair = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float c = 0, cr = 0, cp = 0;
            for (int s = 0; s < KS; ++s) {
                for (int q = 0; q < 4; ++q) c = fmaf(A[i * 4 * KS + 4 * s + q], B[(4 * s + q) * 16 + j], c);
                for (int q = 3; q >= 0; --q) cr = fmaf(A[i * 4 * KS + 4 * s + q], B[(4 * s + q) * 16 + j], cr);
                float t = 0;
                for (int q = 0; q < 4; ++q) t = fmaf(A[i * 4 * KS + 4 * s + q], B[(4 * s + q) * 16 + j], t);
                cp += t;
            }
            const float d = D[i * 16 + j];
            bad_seq += memcmp(&d, &c, 4) != 0;
            bad_rev += memcmp(&d, &cr, 4) != 0;
            bad_pair += memcmp(&d, &cp, 4) != 0;
        }
    printf("mismatches of 256: ascending-k fmaf chain %d, descending-k %d, block-sum-then-add %d\n", bad_seq, bad_rev, bad_pair);
    return 0;
}
