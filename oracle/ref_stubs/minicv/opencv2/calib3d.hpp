/* stands in for <opencv2/calib3d.hpp>: see ../minicv.hpp (TEST INFRASTRUCTURE ONLY) */
#pragma once
#include "../minicv.hpp"
