// detect_mfma.hip -- MFMA (TSR_FAST) bank scorer.  Placeholder until the int8-limb
// kernel lands: requests fall through to TSR_ERR_UNSUPPORTED so nothing silently
// changes precision class.
#include "tsr_internal.h"

int tsr_fast_bank_build(tsr_ctx *, tsr_bank *)
{
    tsr_set_error("TSR_FAST: the MFMA path is not built into this library yet");
    return TSR_ERR_UNSUPPORTED;
}
void tsr_fast_bank_free(tsr_bank *) {}
int tsr_launch_fast_bank(tsr_ctx *, const tsr_bank *, const uint8_t *, int64_t, const int32_t *,
                         const int32_t *, int, float *, int32_t *, float *)
{
    return TSR_ERR_UNSUPPORTED;
}
