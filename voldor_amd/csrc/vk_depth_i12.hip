// vk_depth_i12.hip -- optimize_depth_launch<12, false> and its kernels (vk_depth_impl.hpp): one translation unit per frame bound
#include "vk_depth_impl.hpp"
namespace vk {
template int optimize_depth_launch<12, false>(Context* c, ImageSet& S, const OdParams& p, bool cost_only);
}
