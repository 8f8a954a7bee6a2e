// Instantiations of the MFMA bank kernel for NKB = 5, 6 (see detect_mfma_kernel.inc).
#include "detect_mfma_kernel.inc"

namespace tsr_fast {
template int launch<5>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch<6>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch_bits<5>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_bits<6>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
