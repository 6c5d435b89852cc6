// Hand-written gfx950 kernels for the GP edge-tracing hot path.
// Wavefront = 64 lanes everywhere.  blockIdx.y selects the edge of the batch.
#include "gpet_kernels.h"
#include "gpet_options.h"
#ifndef LB_FN
#define LB_FN inline __attribute__((always_inline))  // (see gpet_lbfgsb_dev.h; -DLB_FN="__attribute__((noinline))" builds round 5's form)
#endif
#include "gpet_lbfgsb_dev.h"

#include <atomic>
#include <type_traits>
#include <math.h>
#include <stdlib.h>

namespace gpet {

// The kernels of the hot path, one file per stage; they are compiled as ONE translation unit (the stages share inlined
// device helpers and launch-time constants; 25 s with hipcc), in this order:
#include "gpet_k_common.inc"
#include "gpet_k_conv.inc"
#include "gpet_k_fit.inc"
#include "gpet_k_factor.inc"
#include "gpet_k_rng.inc"
#include "gpet_k_sample_score.inc"
#include "gpet_k_kde_pix.inc"
#include "gpet_k_lml.inc"
#include "gpet_k_launch.inc"

}  // namespace gpet
