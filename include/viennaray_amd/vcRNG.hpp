// vcRNG.hpp — forwarding header: a ViennaRay program's `#include <vcRNG.hpp>` resolves to the façade
#pragma once
#include "viennaray.hpp"
