// rayGeometry.hpp — forwarding header: lets a ViennaRay program keep its #include <rayGeometry.hpp>
// when its include path points at include/viennaray_amd/ (see INTEGRATION.md).
#pragma once
#include "viennaray.hpp"
