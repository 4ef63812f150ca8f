// Fused tile kernels of the training step's backward pass (feature-head backward phases, grouped backward chains).
#include "pr_common.h"
#include "mlp_tile.h"

namespace pr {
}  // namespace pr
