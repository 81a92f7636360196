// params.h -- host-side handling of the reference's opt:: parameter block
// (ntedit.cpp:99-133) and its translation into the device parameter block.
#pragma once
#include "../../include/ntedit_hip.h"
#include "../csrc/nte_common.h"

#include <cstddef>

namespace nte_host {

void params_default(ntedit_hip_params* p);
void params_clamp(ntedit_hip_params* p, char* warn, size_t cap);

// Folds the float threshold comparisons of ntedit.cpp:1531-1535, 1659-1663,
// 1867-1872, 1992-1997 into integer "count >= thr" thresholds and fills the
// per-k constants.  Returns 0, or a negative NTEDIT_E_* code.
int make_dev_params(
    const ntedit_hip_params& hp,
    uint32_t k,
    uint32_t hash_num,
    bool secbf,
    nte::DevParams* out,
    bool counting = false);

} // namespace nte_host
