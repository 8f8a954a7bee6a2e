// Instantiations of the MFMA bank kernel for NKB = 11, 12 (see detect_mfma_kernel.inc).
#include "detect_mfma_kernel.inc"

namespace tsr_fast {
template int launch<11>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch<12>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch_bits<11>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_bits<12>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
