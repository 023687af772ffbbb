// placeholder until the kernel lands (see git history): family with no configurations
#include "mdhip_internal.h"

namespace mdhip {
namespace MDHIP_ST {

static const ConvCfg g_none = {0, 0, 0, 0, 0, "v6:none"};
int conv6_num_cfgs() { return 0; }
const ConvCfg& conv6_cfg(int) { return g_none; }
hipError_t conv6_init() { return hipSuccess; }
bool conv6_supports(int, const ConvArgs&) { return false; }
hipError_t conv6_launch(int, const ConvArgs&, hipStream_t) { return hipErrorInvalidValue; }

}  // namespace MDHIP_ST
}  // namespace mdhip
