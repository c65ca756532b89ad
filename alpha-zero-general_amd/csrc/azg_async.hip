// azg_async.hip -- third translation unit of libazg_hip.so: the asynchronous tree pipeline (azg_async.hip.h: persistent descent workgroups +
// persistent V80 net workgroups, device-side queues): its NET kernel and its C-ABI.  The descent kernel is azg_async_sel.hip: the two
// kernels want different code generation (build.py: machine LICM off takes the forward from 32.3 to 26.6 us and the descent from 23.2 to
// 30.6 us).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

// every workgroup of k_async_select runs 16 independent tree waves: wave_sync() must be a wavefront fence (azg_common.hip.h)
#define AZG_WAVE_LOCAL_SYNC 1
#define AZG_FUSED_DEVICE_ONLY 1
#define AZG_NN_KERNEL static
#define AZG_NN_OPAQUE_TID 1        /* nn_kernels.hip.h nn_tid(): nothing thread-derived is hoisted out of the persistent net kernel's loop */
#include "../../include/azg.h"
#include "../../include/azg_testaids.h"
#include "azg_host.h"
#include "azg_common.hip.h"
#include "nn_kernels.hip.h"
#include "nn_v80_h2.hip.h"
#include "nn_conv5x5.hip.h"
#include "nn_mb1d.hip.h"
#include "game_santorini.hip.h"
#include "game_azul.hip.h"

using namespace azg;

#define AZG_ASYNC_PART_NET 1
#include "azg_async.hip.h"
