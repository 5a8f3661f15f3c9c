// tu_kernels.h -- what EVERY translation unit of libsparrow_hip.so starts with: the system headers, the C ABI, every kernel header (in the
// order their helpers depend on each other) and tu_instances.h.  sparrow_hip.hip goes on with the host side; a kernel-family unit
// (tu_<n>.hip: `#define SPRK_TU_FAMILY n` + this file) ends here -- its content is the explicit instantiations tu_instances.h turns into
// definitions for family n.  [r5] Until round 4 the library was ONE unit of 57-70 s; now seven build in parallel (sparrowrecsys_amd/_lib.py).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <thread>
#include <vector>

#include "sparrow_hip.h"

#include "host_common.h"             // error reporting, HIP_TRY, roctx ranges, SprkTuning (the environment's switches, read once per finalize)
#include "k_tile_forward.h"          // the device-side plan, the cross hash, the plan interpreter k_tile_forward and the generic DIN stage k_din_pool
// fused kernels, one header per graph family (each documents its own design)
#include "k_chain_v2.h"
#include "k_chain_v2j.h"
#include "k_chain_v2j1.h"
#include "k_rows_chain.h"
#include "k_din_attn.h"
#include "dyn_split.h"
#include "k_din_cols.h"
#include "k_din_tail.h"
#include "k_din_fused.h"
#include "k_chain_v1.h"
#include "k_mlp_rows.h"
#include "k_emb_rank.h"
#include "k_dien_seq.h"
#include "k_dien_mfma.h"
#include "k_dien_fused.h"
#include "k_peer_gather.h"
#include "k_csv_pack.h"
#include "k_operators.h"             // stand-alone operator kernels (bit-exact gather, cross hash) -- closes the kernels' anonymous namespace
#include "tu_instances.h"          // the heavy templates: defined in ONE family unit, `extern template` elsewhere
