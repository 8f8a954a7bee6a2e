// Instantiations of the MFMA bank kernel for NKB = 7, 8 (see detect_mfma_kernel.inc).
#include "detect_mfma_kernel.inc"

namespace tsr_fast {
template int launch<7>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch<8>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch_bits<7>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_bits<8>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
