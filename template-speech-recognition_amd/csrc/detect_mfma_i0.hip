// Instantiations of the MFMA bank kernel for NKB = 1, 2 (see detect_mfma_kernel.inc).
#include "detect_mfma_kernel.inc"

namespace tsr_fast {
template int launch<1>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch<2>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch_bits<1>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_bits<2>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
