// attn_mfma.hip — placeholder until the MFMA kernel lands (next commit).
#include "common.h"
namespace million {
bool attn_mfma_supported(const AttnParams &) { return false; }
int launch_attn_mfma(const AttnParams &, hipStream_t) { set_error("mfma kernel not built"); return MILLION_ERR_SHAPE; }
}
