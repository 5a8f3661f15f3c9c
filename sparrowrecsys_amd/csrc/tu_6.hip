// tu_6.hip -- kernel-family unit 6 of libsparrow_hip.so: EmbeddingMLP / Wide&Deep: k_mlp_rows.
// Nothing but the explicit instantiations tu_instances.h assigns to this family (scripts/gen_tu_instances.py); the kernels' source is in the
// k_*.h headers, the host side in sparrow_hip.hip.
#define SPRK_TU_FAMILY 6
#include "tu_kernels.h"
