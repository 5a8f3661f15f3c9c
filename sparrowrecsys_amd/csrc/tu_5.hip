// tu_5.hip -- kernel-family unit 5 of libsparrow_hip.so: DIN: k_din_attn (wave per sample) and k_din_fused (the whole forward in one launch).
// Nothing but the explicit instantiations tu_instances.h assigns to this family (scripts/gen_tu_instances.py); the kernels' source is in the
// k_*.h headers, the host side in sparrow_hip.hip.
#define SPRK_TU_FAMILY 5
#include "tu_kernels.h"
