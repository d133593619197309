// The split GEMM's kernel instances for bf16 activation storage ("bf16_activations"): the activations are the MFMA
// operand as loaded (no split), the weights stay the three exact bf16 planes of the fp32 tensor (NP = 3: products
// exact with respect to fp32 weights) or only the leading plane (NP = 1: bf16 weights, one MFMA per block).
// Kernels and dispatch: gemm_split_impl.h; tuner and launchers: gemm_split.hip.
#include "gemm_split_impl.h"

namespace dfd {

DFD_S6_INSTANTIATE(bf16_t, 3)
DFD_S6_INSTANTIATE(bf16_t, 1)

}  // namespace dfd
