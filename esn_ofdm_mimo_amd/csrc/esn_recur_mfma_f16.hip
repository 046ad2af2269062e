#include "esn_recur_mfma_impl.h"

namespace esn {

int launch_recur_mfma_f16(const RecurParams& p, hipStream_t stream) {
    const Geometry& g = p.g;
#define ESN_CASE(NWv, MTv, NTv) \
    if (g.NW == NWv && g.MT == MTv && g.NT == NTv) return launch_one<TraitsF16, NWv, MTv, NTv>(p, stream);
    ESN_CASE(4, 1, 2)
    ESN_CASE(8, 1, 4)
    ESN_CASE(8, 2, 4)
    ESN_CASE(8, 2, 3)
    ESN_CASE(8, 4, 2)
    ESN_CASE(16, 4, 1)
    ESN_CASE(8, 2, 1)
#undef ESN_CASE
    return -1;
}

}  // namespace esn
