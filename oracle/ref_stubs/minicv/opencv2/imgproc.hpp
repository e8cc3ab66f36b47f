/* stands in for <opencv2/imgproc.hpp>: see ../minicv.hpp (TEST INFRASTRUCTURE ONLY) */
#pragma once
#include "../minicv.hpp"
