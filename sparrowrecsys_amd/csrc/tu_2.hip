// tu_2.hip -- kernel-family unit 2 of libsparrow_hip.so: DeepFM_v2 one-task kernel, the all-large-fields chain, the rows kernels: k_deepfm_v2_joint1, k_deepfm_v2_chain, k_rows_chain / 1 / _many.
// Nothing but the explicit instantiations tu_instances.h assigns to this family (scripts/gen_tu_instances.py); the kernels' source is in the
// k_*.h headers, the host side in sparrow_hip.hip.
#define SPRK_TU_FAMILY 2
#include "tu_kernels.h"
