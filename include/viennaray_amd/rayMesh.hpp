// rayMesh.hpp — forwarding header: lets a ViennaRay program keep its #include <rayMesh.hpp>
// when its include path points at include/viennaray_amd/ (see INTEGRATION.md).
#pragma once
#include "viennaray.hpp"
