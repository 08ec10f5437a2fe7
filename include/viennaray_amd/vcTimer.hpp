// vcTimer.hpp — stand-in for the ViennaCore header the examples include (Timer lives in viennaray.hpp).
#pragma once
#include "viennaray.hpp"
