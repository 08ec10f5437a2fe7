// rayReflection.hpp — forwarding header: a ViennaRay program's `#include <rayReflection.hpp>` resolves to the façade
#pragma once
#include "viennaray.hpp"
