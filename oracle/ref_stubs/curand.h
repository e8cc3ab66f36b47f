#pragma once
#include "cuda_stub_common.h"
