// Instantiations of tap2_bank_kernel (detect_mfma_tap2.inc) for Dp / 16 = 1, 3, 5, 7 K-blocks.
#include "detect_mfma_tap2.inc"

namespace tsr_fast {
template int launch_tap2<1>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<3>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<5>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<7>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
