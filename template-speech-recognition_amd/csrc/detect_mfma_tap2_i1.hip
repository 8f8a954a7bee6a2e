// Instantiations of tap2_bank_kernel (detect_mfma_tap2.inc) for Dp / 16 = 9, 11, 13, 15 K-blocks.
#include "detect_mfma_tap2.inc"

namespace tsr_fast {
template int launch_tap2<9>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<11>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<13>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_tap2<15>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
