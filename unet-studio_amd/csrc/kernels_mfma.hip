// MFMA implicit-GEMM conv family (bf16, gfx950).  Placeholder until the first kernel lands.
#include "device_util.h"

namespace unet {
bool mfma_conv_fwd_supported(int, const ConvGeom&, const SrcDesc*, int) { return false; }
}  // namespace unet
