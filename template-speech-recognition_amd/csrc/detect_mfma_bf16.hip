// The bf16 variants of the bank pass (BASELINE configs[4] / SURVEY 8(d) "config 5": the bank as a bf16 MFMA GEMM,
// single-limb and hi + lo) -- kernels: alt_bank_kernel<..., BF = true> of detect_mfma_kernel.inc -- and the bf16
// copy of a resident corpus they read.  Comparison variants: TSR_FAST (int8 limbs, exact integer sums) is the
// product path.
#include "detect_mfma_kernel.inc"

namespace tsr_fast {

int launch_bf16(tsr_ctx *ctx, const FastArgs &args, int grid, size_t lds, int NL)
{
    const int mode = args.raw_frag ? 2 : (args.all ? 1 : 0);
    if (NL == 1) {
        if (mode == 2) return launch_alt<kBfSteps, 1, true, 2, true>(ctx, args, grid, lds);
        if (mode == 1) return launch_alt<kBfSteps, 1, true, 1, true>(ctx, args, grid, lds);
        return launch_alt<kBfSteps, 1, true, 0, true>(ctx, args, grid, lds);
    }
    if (mode == 2) return launch_alt<kBfSteps, 2, true, 2, true>(ctx, args, grid, lds);
    if (mode == 1) return launch_alt<kBfSteps, 2, true, 1, true>(ctx, args, grid, lds);
    return launch_alt<kBfSteps, 2, true, 0, true>(ctx, args, grid, lds);
}

}  // namespace tsr_fast

namespace {
// resident 0/1 bytes [P][Dp] -> bf16 rows of `stride` bytes (1.0 = 0x3F80), zero padded
__global__ __launch_bounds__(256) void corpus_to_bf16_kernel(const uint8_t *__restrict__ X, int64_t P, int Dp, int stride,
                                                             uint16_t *__restrict__ out)
{
    const int per_row = stride >> 4;  // 16-byte chunks = 8 features
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * per_row) return;
    const int64_t p = i / per_row;
    const int c = (int)(i - p * per_row) * 8;
    uint16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c + e < Dp && X[p * Dp + c + e]) ? 0x3F80 : 0;
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(out) + p * stride + (int64_t)c * 2) = *reinterpret_cast<const uint4 *>(v);
}
}  // namespace

int tsr_corpus_bf16_copy(tsr_ctx *ctx, tsr_corpus *cp)
{
    if (cp->d_feats_bf16) return TSR_OK;
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
    int stride = 2 * cp->Dp;
    if (((stride >> 4) & 1) == 0) stride += 16;  // odd number of 16-byte slots: conflict-free fragment reads
    const size_t body = (size_t)cp->P * stride, slack = (size_t)48 * stride + 8192;
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(&cp->d_feats_bf16), body + slack));
    TSR_HIP(hipMemsetAsync(reinterpret_cast<uint8_t *>(cp->d_feats_bf16) + body, 0, slack, ctx->stream));
    const int64_t n = cp->P * (stride >> 4);
    if (n > 0) {
        hipLaunchKernelGGL(corpus_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, cp->d_feats, cp->P,
                           cp->Dp, stride, cp->d_feats_bf16);
        TSR_HIP(hipGetLastError());
    }
    cp->bf16_stride = stride;
    cp->bf16_alloc = body + slack;
    return TSR_OK;
}
