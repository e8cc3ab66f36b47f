/* stands in for gpu-kernels/fb_smooth.h inside optimize_depth.cu's device prefix: its kernel is compiled separately
 * (REF_FB_INC), its host function uses the <<< >>> launch syntax and is not needed */
#pragma once
