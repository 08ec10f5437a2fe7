// ViennaCore vector types/helpers used by ViennaRay programs: forwarded to the drop-in facade
#pragma once
#include "viennaray.hpp"
