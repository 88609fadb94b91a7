// flat_mfma_bf16.hip -- bfloat16-row instantiations of the MFMA batch kernel
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
#include "flat_mfma_kernel.h"

namespace rag {
namespace mfma {
int dispatch_bf16(const MfmaPlan& p, const Args& a) { return dispatch<true>(p, a); }
}  // namespace mfma
}  // namespace rag
