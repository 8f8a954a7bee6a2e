// Instantiations of the MFMA bank kernel for NKB = 3, 4 (see detect_mfma_kernel.inc).
#include "detect_mfma_kernel.inc"

namespace tsr_fast {
template int launch<3>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch<4>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch_bits<3>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_bits<4>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
