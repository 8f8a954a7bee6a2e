// Instantiations of the MFMA bank kernel for NKB = 9, 10 (see detect_mfma_kernel.inc).
#include "detect_mfma_kernel.inc"

namespace tsr_fast {
template int launch<9>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch<10>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
template int launch_bits<9>(tsr_ctx *, const FastArgs &, int, size_t);
template int launch_bits<10>(tsr_ctx *, const FastArgs &, int, size_t);
}  // namespace tsr_fast
