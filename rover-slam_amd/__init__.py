"""rover-slam_amd: MI355X-native learned front end (SuperPoint + LightGlue) for Rover-SLAM.

Host-side mirror of the reference's extractor / matcher interface
(include/Extractors/SPextractor.h, include/Matchers/SPmatcher.h in the reference) over the C ABI
of librover_fe.so (include/rover_fe.h).  The HIP library is the only compute path: there is no CPU
fallback, and importing `.capi` raises if the library is missing.
"""
from . import weights, synth  # noqa: F401

__all__ = ["weights", "synth"]
