// raySource.hpp — forwarding header: a ViennaRay program's `#include <raySource.hpp>` resolves to the façade
#pragma once
#include "viennaray.hpp"
