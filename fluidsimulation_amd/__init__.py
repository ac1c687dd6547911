"""MI355X-native Position-Based-Fluids sub-step behind the reference's Simulation step loop.

The product is the C-ABI library built from fluidsimulation_amd/csrc (include/pbf.h);
this package is the thin host-side plumbing around it.
"""
from .binding import (PbfContext, PbfError, PbfParams, PbfCollider, PbfStats, default_params,
                      plane, aabb, load_library)

__all__ = ["PbfContext", "PbfError", "PbfParams", "PbfCollider", "PbfStats", "default_params",
           "plane", "aabb", "load_library"]
