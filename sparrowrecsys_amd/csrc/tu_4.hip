// tu_4.hip -- kernel-family unit 4 of libsparrow_hip.so: DIN / DIEN tail, the two-launch attention, DIEN's recurrence: k_din_tail, k_din_attn_cols, k_dien_seq[_mfma].
// Nothing but the explicit instantiations tu_instances.h assigns to this family (scripts/gen_tu_instances.py); the kernels' source is in the
// k_*.h headers, the host side in sparrow_hip.hip.
#define SPRK_TU_FAMILY 4
#include "tu_kernels.h"
