// Instantiations of tap2_bank_kernel (detect_mfma_tap2.inc) for Dp / 16 = 17, 19, 21, 23 K-blocks.
#include "detect_mfma_tap2.inc"

namespace tsr_fast {
template int launch_tap2<17>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<19>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<21>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<23>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
