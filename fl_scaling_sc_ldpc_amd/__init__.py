"""MI355X-native Monte-Carlo decoding of random SC-LDPC ensembles over the BEC.

Drop-in for the hot path of rsokolovskii/fl_scaling_sc_ldpc (simulators_sc_ldpc/{bp_decoding,
peeling_decoding}); all compute lives in libscldpc_hip.so (C-ABI: include/scldpc.h), built from
fl_scaling_sc_ldpc_amd/csrc for gfx950.  There is no CPU fallback.
"""
from ._lib import CodeParams, ScldpcError, LIB_PATH  # noqa: F401

__all__ = ["CodeParams", "ScldpcError", "LIB_PATH"]
