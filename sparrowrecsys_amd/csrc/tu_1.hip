// tu_1.hip -- kernel-family unit 1 of libsparrow_hip.so: DeepFM_v2, looped kernels: k_deepfm_v2_joint / _joint_many.
// Nothing but the explicit instantiations tu_instances.h assigns to this family (scripts/gen_tu_instances.py); the kernels' source is in the
// k_*.h headers, the host side in sparrow_hip.hip.
#define SPRK_TU_FAMILY 1
#include "tu_kernels.h"
