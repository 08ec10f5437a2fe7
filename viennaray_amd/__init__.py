"""viennaray_amd — MI355X-native flux ray-tracing core (ViennaRay-compatible hot path).

The product is the HIP library behind include/viennaray_amd.h; this package is
its Python host mirror (used by bench.py and the parity tests).
"""
from .capi import VrError, load, device_available, LIB_PATH  # noqa: F401
from .trace import (BoundaryCondition, TraceDirection, NormalizationType,  # noqa: F401
                    TracingDataMergeEnum, DiffuseParticle, SpecularParticle,
                    ConedCosineParticle, DiffuseCosineParticle, CoverageStickingParticle, UserModelParticle, SourceGrid,
                    TracingData, Trace, TraceDisk, TraceTriangle)
from . import io  # noqa: F401
