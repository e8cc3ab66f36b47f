// vk_depth_i6.hip -- optimize_depth_launch<6, false> and its kernels (vk_depth_impl.hpp): one translation unit per frame bound
#define VK_PHASE_UNIT 1
#include "vk_depth_impl.hpp"
namespace vk {
template int optimize_depth_launch<6, false>(Context* c, ImageSet& S, const OdParams& p, bool cost_only);
}
