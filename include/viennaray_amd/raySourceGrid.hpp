// raySourceGrid.hpp — forwarding header: a ViennaRay program's `#include <raySourceGrid.hpp>` resolves to the façade
#pragma once
#include "viennaray.hpp"
