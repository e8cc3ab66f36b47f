// vk_depth_s8.hip -- optimize_depth_launch<8, true> and its kernels (vk_depth_impl.hpp): one translation unit per frame bound
#include "vk_depth_impl.hpp"
namespace vk {
template int optimize_depth_launch<8, true>(Context* c, ImageSet& S, const OdParams& p, bool cost_only);
}
