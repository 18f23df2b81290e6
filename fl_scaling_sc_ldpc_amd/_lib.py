"""ctypes binding of libscldpc_hip.so (C-ABI: include/scldpc.h).

The library is built in-tree by `make -C fl_scaling_sc_ldpc_amd/csrc` (or __graft_entry__.build()).
There is NO fallback: if the shared object is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SCLDPC_LIB_PATH: diagnostics only (tools/stamps.py loads the stamped build of the same sources)
LIB_PATH = os.environ.get("SCLDPC_LIB_PATH") or os.path.join(HERE, "libscldpc_hip.so")

NCOUNTERS = 8
NRUN = 9
NPEELRUN = 8
PEELRUN_NAMES = ("trials", "fuckups", "lost", "fuckups_exp", "lost_exp", "blocks_exp")     # SCLDPC_PR_*
COUNTER_NAMES = ("num_erasures", "num_blocks_err", "num_erasures_exp", "num_blocks_err_exp",
                 "num_erasures_p1", "iterations", "status", "channel_erasures")
RUN_NAMES = ("users_err", "frame_err", "frame_err_p1", "block_err", "users_err_exp", "frame_err_exp",
             "block_err_exp", "frames", "iterations")

# every symbol include/scldpc.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "scldpc_abi_version", "scldpc_last_error", "scldpc_device_count",
    "scldpc_sample_glibc_host", "scldpc_glibc_state_bytes", "scldpc_glibc_state_init",
    "scldpc_glibc_state_reset_perm", "scldpc_sample_glibc_next_host",
    "scldpc_sample_philox_device", "scldpc_full_bp_device", "scldpc_sw_bp_device",
    "scldpc_accumulate_run_device", "scldpc_full_bp_lds_bytes",
    "scldpc_sample_philox_device_adj16", "scldpc_full_bp_device_adj16", "scldpc_sw_bp_device_adj16",
    "scldpc_peel_sweep_device", "scldpc_peel_sweep_device_adj16",
    "scldpc_peel_pick_device", "scldpc_peel_pick_device_adj16", "scldpc_r1_moments_device",
    "scldpc_stream_state_bytes", "scldpc_stream_run_device",
    "scldpc_swc_bp_device", "scldpc_swc_bp_device_adj16", "scldpc_sample_philox_ensemble_device",
    "scldpc_full_bp_fixpoint_device", "scldpc_full_bp_fixpoint_device_adj16",
    "scldpc_sample_philox_cn16_supported", "scldpc_sample_philox_device_cn16",
    "scldpc_sample_philox_sock16_supported", "scldpc_sample_philox_device_sock16",
    "scldpc_full_bp_cn16_supported", "scldpc_full_bp_fixpoint_device_cn16", "scldpc_full_bp_device_cn16",
    "scldpc_stream_glibc_inputs_host", "scldpc_stream_run_device_inputs", "scldpc_workspace_bytes",
    "scldpc_sw_bp_ring_supported", "scldpc_cn_sockets_device", "scldpc_sw_bp_ring_device",
    "scldpc_accumulate_peel_device", "scldpc_clear_channel_range_device",
    "scldpc_stream_glibc_next_host", "scldpc_stream_run_device_inputs_at",
    "scldpc_full_bp_sock16_supported", "scldpc_full_bp_fixpoint_device_sock16", "scldpc_full_bp_device_sock16",
    "scldpc_full_bp_traj_device_cn16", "scldpc_full_bp_traj_device_sock16",
)


class ScldpcError(RuntimeError):
    pass


class CodeParams(C.Structure):
    """scldpc_code_params — (dv, dc, L, cns_pos, vns_pos); see include/scldpc.h for the naming trap."""
    _fields_ = [("dv", C.c_int32), ("dc", C.c_int32), ("L", C.c_int32), ("cns_pos", C.c_int32),
                ("vns_pos", C.c_int32)]

    @property
    def n(self):
        return self.vns_pos * self.L

    @property
    def nk(self):
        return (self.L + self.dv - 1) * self.cns_pos

    @property
    def nw(self):
        return (self.n + 31) // 32

    @property
    def edges(self):
        return self.n * self.dv

    def key(self):
        return (self.dv, self.dc, self.L, self.cns_pos, self.vns_pos)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScldpcError(f"{LIB_PATH} is missing — build it with `make -C fl_scaling_sc_ldpc_amd/csrc` "
                          "(hipcc, gfx950).  There is no CPU fallback.")
    # PyTorch's copy of the HIP runtime first: loaded after this library it would be a second runtime in the process, and
    # whichever of the two touches the GPU second finds no device
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp, i32, i64, u32, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
    pp = P(CodeParams)
    L.scldpc_abi_version.restype = C.c_int
    L.scldpc_last_error.restype = C.c_char_p
    L.scldpc_device_count.restype = C.c_int
    L.scldpc_sample_glibc_host.argtypes = [pp, u32, dbl, i32, vp, vp, vp]
    L.scldpc_glibc_state_bytes.argtypes = [pp]
    L.scldpc_glibc_state_bytes.restype = i64
    L.scldpc_glibc_state_init.argtypes = [pp, u32, vp]
    L.scldpc_glibc_state_reset_perm.argtypes = [pp, vp]
    L.scldpc_sample_glibc_next_host.argtypes = [pp, vp, dbl, i32, vp, i32, vp, vp]
    L.scldpc_workspace_bytes.argtypes = [i32, pp, i32, i32, i32]
    L.scldpc_workspace_bytes.restype = i64
    L.scldpc_sample_philox_device.argtypes = [pp, u64, u64, i32, dbl, i32, vp, vp, vp, vp, u64, vp]
    L.scldpc_sample_philox_ensemble_device.argtypes = [pp, i32, u64, u64, i32, dbl, i32, vp, vp, vp, vp]
    L.scldpc_full_bp_fixpoint_device.argtypes = [pp, i32, vp, vp, i32, vp, vp, vp, u64, vp]
    L.scldpc_full_bp_fixpoint_device_adj16.argtypes = L.scldpc_full_bp_fixpoint_device.argtypes
    L.scldpc_sample_philox_cn16_supported.argtypes = [pp]
    L.scldpc_full_bp_cn16_supported.argtypes = [pp]
    L.scldpc_sample_philox_device_cn16.argtypes = [pp, u64, u64, i32, dbl, i32, vp, vp, vp, vp, vp]
    L.scldpc_sample_philox_sock16_supported.argtypes = [pp]
    L.scldpc_sample_philox_device_sock16.argtypes = [pp, u64, u64, i32, dbl, i32, vp, vp, vp, vp, vp]
    L.scldpc_full_bp_fixpoint_device_cn16.argtypes = [pp, i32, vp, vp, vp, i32, vp, vp, vp]
    L.scldpc_full_bp_device_cn16.argtypes = [pp, i32, vp, vp, vp, i32, i32, vp, vp, vp]
    L.scldpc_full_bp_sock16_supported.argtypes = [pp]
    L.scldpc_full_bp_fixpoint_device_sock16.argtypes = L.scldpc_full_bp_fixpoint_device_cn16.argtypes
    L.scldpc_full_bp_device_sock16.argtypes = L.scldpc_full_bp_device_cn16.argtypes
    L.scldpc_full_bp_traj_device_cn16.argtypes = [pp, i32, vp, vp, vp, i32, i32, vp, vp, i32, vp, vp]
    L.scldpc_full_bp_traj_device_sock16.argtypes = L.scldpc_full_bp_traj_device_cn16.argtypes
    L.scldpc_full_bp_device.argtypes = [pp, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, u64, vp]
    L.scldpc_sw_bp_device.argtypes = [pp, i32, vp, vp, i32, i32, i32, vp, vp, vp, u64, vp]
    L.scldpc_sample_philox_device_adj16.argtypes = L.scldpc_sample_philox_device.argtypes
    L.scldpc_full_bp_device_adj16.argtypes = L.scldpc_full_bp_device.argtypes
    L.scldpc_sw_bp_device_adj16.argtypes = L.scldpc_sw_bp_device.argtypes
    L.scldpc_peel_sweep_device.argtypes = [pp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, u64, vp]
    L.scldpc_peel_sweep_device_adj16.argtypes = L.scldpc_peel_sweep_device.argtypes
    L.scldpc_peel_pick_device.argtypes = [pp, i32, vp, vp, i32, i32, vp, u64, u64, vp, vp, vp, vp, u64, vp]
    L.scldpc_peel_pick_device_adj16.argtypes = L.scldpc_peel_pick_device.argtypes
    L.scldpc_r1_moments_device.argtypes = [i32, i32, vp, vp, vp]
    L.scldpc_swc_bp_device.argtypes = [pp, i32, vp, vp, i32, i32, vp, vp, vp, u64, vp]
    L.scldpc_swc_bp_device_adj16.argtypes = L.scldpc_swc_bp_device.argtypes
    L.scldpc_stream_state_bytes.argtypes = [pp, i32]
    L.scldpc_stream_state_bytes.restype = i64
    L.scldpc_stream_run_device.argtypes = [pp, i32, u64, u64, dbl, i32, i32, vp, i32, vp, vp, vp, vp]
    L.scldpc_stream_glibc_inputs_host.argtypes = [pp, u32, dbl, i32, vp, i32, vp, vp]
    L.scldpc_stream_run_device_inputs.argtypes = [pp, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, i32, i64, vp]
    L.scldpc_stream_glibc_next_host.argtypes = [pp, vp, dbl, i32, vp, i32, i64, i32, vp, vp]
    L.scldpc_stream_run_device_inputs_at.argtypes = [pp, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, i64, i32, i64, vp]
    L.scldpc_sw_bp_ring_supported.argtypes = [pp, i32]
    L.scldpc_cn_sockets_device.argtypes = [pp, i32, vp, vp, vp]
    L.scldpc_sw_bp_ring_device.argtypes = [pp, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.scldpc_accumulate_run_device.argtypes = [i32, vp, i64, vp, vp]
    L.scldpc_accumulate_peel_device.argtypes = [i32, vp, i64, vp, vp]
    L.scldpc_clear_channel_range_device.argtypes = [pp, i32, i32, i32, vp, vp]
    L.scldpc_full_bp_lds_bytes.argtypes = [pp]
    L.scldpc_full_bp_lds_bytes.restype = i64
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise ScldpcError(f"libscldpc_hip error {rc}: {lib().scldpc_last_error().decode()}")


def doped_array(doped):
    arr = np.ascontiguousarray(doped, dtype=np.int32).reshape(-1)
    return arr, (arr.ctypes.data if arr.size else None)
