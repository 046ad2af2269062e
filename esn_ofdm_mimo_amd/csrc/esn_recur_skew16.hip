// Instantiations of the 16x16x32 skewed predict kernel (esn_recur_skew16_impl.h).
#include "esn_recur_skew16_impl.h"

namespace esn {

int launch_recur_skew16(int precision, const RecurParams& p, hipStream_t stream) {
    if (precision == ESN_F16) return launch_skew16<TraitsF16>(p, stream);
    if (precision == ESN_BF16) return launch_skew16<TraitsBF16>(p, stream);
    return -1;
}

}  // namespace esn
