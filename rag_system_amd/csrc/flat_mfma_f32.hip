// flat_mfma_f32.hip -- fp32-row instantiations of the MFMA batch kernel (v_mfma_f32_32x32x2_f32).
#include "flat_mfma_kernel.h"

namespace rag {
namespace mfma {
int dispatch_f32(const MfmaPlan& p, const Args& a) { return dispatch<false>(p, a); }
}  // namespace mfma
}  // namespace rag
