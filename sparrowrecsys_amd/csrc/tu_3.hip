// tu_3.hip -- kernel-family unit 3 of libsparrow_hip.so: pair-dot DeepFM: k_deepfm_pairs / _many / 1.
// Nothing but the explicit instantiations tu_instances.h assigns to this family (scripts/gen_tu_instances.py); the kernels' source is in the
// k_*.h headers, the host side in sparrow_hip.hip.
#define SPRK_TU_FAMILY 3
#include "tu_kernels.h"
