// mvx_plan.h - plan_call(): every decision run() (mvx_capi.hip) takes about HOW a call is executed - route, slab plan, channel
// chunks, molecule chunks, pacing, write-out path - as one pure host function of the call's shape, so that the decision table
// can be pinned by CPU tests (tests/test_plan.py through mvx_plan_call) next to the measurements that justify it
// (mvx_tuning.h, profiles/r03_odd_dimensions.txt).
#pragma once
#include <stdint.h>

#include "../../include/mvx.h"
#include "mvx_tuning.h"

namespace mvx {

// per-handle test / measurement switches (mvx_debug_set_option); the defaults are the production plan
struct PlanKnobs {
    int32_t force_nw = 0;     // waves per slab (0: the plan's)
    int32_t max_ct = 32;      // channels per workgroup, float32 grids
    int32_t max_ct64 = 32;    // float64 grids: 32 = matrix-core chunks where they apply
    int32_t direct_mode = -1; // -1: the rule; 0 / 1: never / always the one-launch route (where it applies)
    int32_t pipeline = 1;     // > 1: molecule chunks with the pre-pass on a side stream
    double mall_budget = MALL_BUDGET;
};

mvx_plan plan_call(const mvx_plan_query &q, const PlanKnobs &k);

} // namespace mvx
