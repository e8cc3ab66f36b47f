/* stands in for <opencv2/highgui.hpp>: see ../minicv.hpp (TEST INFRASTRUCTURE ONLY) */
#pragma once
#include "../minicv.hpp"
