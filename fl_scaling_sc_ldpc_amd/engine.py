"""Thin host layer over the C-ABI: device buffers (torch, plumbing only) + kernel launches.

Everything that computes runs inside libscldpc_hip.so; this module only owns memory and streams.
Reference routines replaced (see include/scldpc.h for line citations): generate_code /
channel_doped (sampling), decodeBP / decodeBP_SW (decoding), plr_computation / willIstop (run
accumulation).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import (CodeParams, NCOUNTERS, NRUN, NPEELRUN, COUNTER_NAMES, RUN_NAMES, PEELRUN_NAMES, ScldpcError,  # noqa: F401
                   check, lib)


def make_params(dv=4, dc=8, L=50, N=1000):
    """(dv, dc, L, N) as in BASELINE.json: N = VNs per position (Def_VNsPos, Python's M);
    cns_pos = N*dv/dc (Def_M)."""
    if (N * dv) % dc:
        raise ValueError(f"N*dv must be divisible by dc (N={N}, dv={dv}, dc={dc})")
    return CodeParams(dv, dc, L, N * dv // dc, N)


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


WS_SAMPLE, WS_FULL_BP, WS_SW_BP, WS_PEEL_SWEEP, WS_PEEL_PICK = range(5)      # SCLDPC_WS_*


def _workspace(op, p, ntrials, device, arg0=0, arg1=0):
    """(device pointer, bytes) of a caller-owned workspace for one call, or (None, 0) when the ensemble needs none.
    A fresh torch buffer per call: the caching allocator hands memory back only to the stream that used it, so calls
    on different streams never share scratch (the library itself allocates nothing)."""
    nbytes = lib().scldpc_workspace_bytes(op, C.byref(p), int(ntrials), int(arg0), int(arg1))
    if nbytes < 0:
        check(int(nbytes))
    if nbytes == 0:
        return None, 0, None
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return buf.data_ptr(), int(nbytes), buf


def _require_gpu():
    if not torch.cuda.is_available():
        raise ScldpcError("no HIP device visible: the decoders run only on the GPU (no CPU fallback)")


# ------------------------------------------------------------------------------------------------
# sampling
# ------------------------------------------------------------------------------------------------
def sample_glibc_trials(p, seeds, eps, doped=()):
    """Self-contained trials exactly as the reference draws them after `inizio_sim(); srandom(seed)`.
    Returns host arrays vn_adj int32 [T,n,dv], chan_bits uint32 [T,nw]."""
    seeds = np.asarray(seeds, dtype=np.uint32).reshape(-1)
    T = seeds.size
    vn_adj = np.empty((T, p.n, p.dv), dtype=np.int32)
    chan = np.empty((T, p.nw), dtype=np.uint32)
    darr, dptr = _lib.doped_array(doped)
    for t in range(T):
        check(lib().scldpc_sample_glibc_host(C.byref(p), int(seeds[t]), float(eps), darr.size, dptr,
                                             vn_adj[t].ctypes.data, chan[t].ctypes.data))
    return vn_adj, chan


class GlibcRun:
    """One reference run: a single srandom(seed), frames drawn back to back with perm_code and the
    random() stream carried over (main_terminated, BPF:2057-2131)."""

    def __init__(self, p, seed):
        self.p = p
        nbytes = lib().scldpc_glibc_state_bytes(C.byref(p))
        if nbytes < 0:
            check(int(nbytes))
        self._state = np.zeros(nbytes, dtype=np.uint8)
        check(lib().scldpc_glibc_state_init(C.byref(p), int(seed) & 0xFFFFFFFF, self._state.ctypes.data))

    def new_point(self):
        """inizio_sim(): perm_code := identity at the start of every ε point (BPF:308-311)."""
        check(lib().scldpc_glibc_state_reset_perm(C.byref(self.p), self._state.ctypes.data))

    def snapshot(self):
        return self._state.copy()

    def restore(self, snap):
        self._state[:] = snap

    def next_frames(self, nframes, eps, doped=()):
        p = self.p
        vn_adj = np.empty((nframes, p.n, p.dv), dtype=np.int32)
        chan = np.empty((nframes, p.nw), dtype=np.uint32)
        darr, dptr = _lib.doped_array(doped)
        check(lib().scldpc_sample_glibc_next_host(C.byref(p), self._state.ctypes.data, float(eps), darr.size, dptr,
                                                  nframes, vn_adj.ctypes.data, chan.ctypes.data))
        return vn_adj, chan


def to_device(vn_adj, chan, device="cuda:0"):
    _require_gpu()
    vn_adj = np.ascontiguousarray(vn_adj)
    if vn_adj.dtype != np.int16:
        vn_adj = vn_adj.astype(np.int32, copy=False)
    d_adj = torch.from_numpy(vn_adj).to(device)
    d_ch = torch.from_numpy(np.ascontiguousarray(chan).view(np.int32)).to(device)
    return d_adj, d_ch


def _is_adj16(d_adj):
    """int16 tensors carry the compact position-local adjacency (uint16 bit patterns)."""
    return d_adj.dtype == torch.int16


ENSEMBLES = {"olmos": 0, "tail_biting": 1, "protograph": 2}      # SCLDPC_ENS_*


def sample_philox(p, seed, trial0, ntrials, eps, doped=(), device="cuda:0", out=None, adj16=False, ensemble="olmos"):
    """Throughput-mode sampling on the device (counter-based; see scldpc_sample_philox_device).
    adj16=True: compact adjacency, int16 tensor [T,n,dv] holding uint16 position-local CN ids (Olmos chain only).
    ensemble: "olmos" (generate_code / sc_ldpc.gen_slots), "tail_biting" or "protograph" (global CN ids)."""
    _require_gpu()
    if out is None:
        d_adj = torch.empty((ntrials, p.n, p.dv), dtype=torch.int16 if adj16 else torch.int32, device=device)
        d_ch = torch.empty((ntrials, p.nw), dtype=torch.int32, device=device)
    else:
        d_adj, d_ch = out
    darr, dptr = _lib.doped_array(doped)
    if ensemble != "olmos":
        if _is_adj16(d_adj):
            raise ValueError("the tail-biting / protograph samplers write global CN ids: use adj16=False")
        check(lib().scldpc_sample_philox_ensemble_device(C.byref(p), ENSEMBLES[ensemble], int(seed), int(trial0),
                                                         int(ntrials), float(eps), darr.size, dptr, d_adj.data_ptr(),
                                                         d_ch.data_ptr(), _stream_ptr(d_adj.device)))
        return d_adj, d_ch
    fn = lib().scldpc_sample_philox_device_adj16 if _is_adj16(d_adj) else lib().scldpc_sample_philox_device
    ws, wsb, _keep = _workspace(WS_SAMPLE, p, ntrials, d_adj.device)
    check(fn(C.byref(p), int(seed), int(trial0), int(ntrials), float(eps), darr.size, dptr, d_adj.data_ptr(),
             d_ch.data_ptr(), ws, wsb, _stream_ptr(d_adj.device)))
    return d_adj, d_ch


def local_device():
    """This rank's GPU: cuda:LOCAL_RANK, as torch.distributed.run numbers the processes of a node.  In the rehearsal mode of
    the tests (SCLDPC_DIST_BACKEND set to something other than nccl) ranks may share a GPU: LOCAL_RANK wraps around."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SCLDPC_DIST_BACKEND", "nccl") != "nccl":
        local %= max(1, torch.cuda.device_count())
    return torch.device("cuda", local)


def init_distributed():
    """Under torch.distributed.run (WORLD_SIZE > 1) join the job: one process per GPU, RCCL (backend "nccl") over xGMI.
    Returns True when this call created the process group (the caller then destroys it).  SCLDPC_DIST_BACKEND=gloo is the
    tests' rehearsal of the multi-rank drivers on a box with fewer GPUs than ranks."""
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or dist.is_initialized():
        return False
    _require_gpu()
    dev = local_device()
    torch.cuda.set_device(dev)
    backend = os.environ.get("SCLDPC_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    return True


def cn16_supported(p):
    """The (4,8) chain with N <= 2048 and fewer than 65535 VNs: the second-generation sampler + the 4-bits-per-CN decoder take it."""
    return bool(lib().scldpc_sample_philox_cn16_supported(C.byref(p))) and bool(lib().scldpc_full_bp_cn16_supported(C.byref(p)))


def sample_philox_cn16(p, seed, trial0, ntrials, eps, doped=(), device="cuda:0", out=None, want_cn=True):
    """scldpc_sample_philox_device_cn16: (vn_adj16 int16 [T,n,4], cn_adj16 int16 [T,nk,8] | None, chan int32 [T,nw]);
    the first and the last are bit for bit sample_philox(..., adj16=True)'s; cn_adj16 holds the VNs of every CN as
    uint16 bit patterns (0xFFFF: none), in unspecified order."""
    _require_gpu()
    if out is None:
        d_adj = torch.empty((ntrials, p.n, p.dv), dtype=torch.int16, device=device)
        d_cn = torch.empty((ntrials, p.nk, p.dc), dtype=torch.int16, device=device) if want_cn else None
        d_ch = torch.empty((ntrials, p.nw), dtype=torch.int32, device=device)
    else:
        d_adj, d_cn, d_ch = out
    darr, dptr = _lib.doped_array(doped)
    check(lib().scldpc_sample_philox_device_cn16(C.byref(p), int(seed), int(trial0), int(ntrials), float(eps), darr.size,
                                                 dptr, d_adj.data_ptr(), d_cn.data_ptr() if d_cn is not None else None,
                                                 d_ch.data_ptr(), _stream_ptr(d_adj.device)))
    return d_adj, d_cn, d_ch


def sw_ring_supported(p, W):
    """Whether the window-state-in-LDS decoder (sw_ring.hip) takes the square window W on this ensemble."""
    return bool(lib().scldpc_sw_bp_ring_supported(C.byref(p), int(W)))


def sock16_supported(p):
    """The (4,8) chain with N <= 2048: the second-generation sampler emits the ring window decoder's CN -> socket table."""
    return bool(lib().scldpc_sample_philox_sock16_supported(C.byref(p)))


def sample_philox_sock16(p, seed, trial0, ntrials, eps, doped=(), device="cuda:0", out=None):
    """scldpc_sample_philox_device_sock16: (vn_adj16 int16 [T,n,4], cn_sock16 int16 [T,nk,8], chan int32 [T,nw]); the first
    and the last are bit for bit sample_philox(..., adj16=True)'s, cn_sock16 is cn_sockets(p, vn_adj16) as a set per CN."""
    _require_gpu()
    if out is None:
        d_adj = torch.empty((ntrials, p.n, p.dv), dtype=torch.int16, device=device)
        d_cs = torch.empty((ntrials, p.nk, p.dc), dtype=torch.int16, device=device)
        d_ch = torch.empty((ntrials, p.nw), dtype=torch.int32, device=device)
    else:
        d_adj, d_cs, d_ch = out
    darr, dptr = _lib.doped_array(doped)
    check(lib().scldpc_sample_philox_device_sock16(C.byref(p), int(seed), int(trial0), int(ntrials), float(eps), darr.size,
                                                   dptr, d_adj.data_ptr(), d_cs.data_ptr(), d_ch.data_ptr(),
                                                   _stream_ptr(d_adj.device)))
    return d_adj, d_cs, d_ch


def cn_adj_from_vn_adj(p, adj16):
    """Host: the CN -> VN table (uint16 [.., nk, dc] as int16 bit patterns: the VNs of every CN in ascending order,
    0xFFFF where a chain-end CN has fewer than dc) from the position-local VN -> CN table — for feeding host-sampled
    codes (glibc replay, fixtures) to full_bp_fixpoint_cn16, and for checking the device table."""
    a = np.ascontiguousarray(adj16).view(np.uint16).astype(np.int64).reshape(-1, p.n, p.dv)
    if p.n >= 0xFFFF:
        raise ValueError("the CN -> VN table holds 16-bit VN indices: n must be below 65535")
    T = a.shape[0]
    out = np.full((T, p.nk, p.dc), 0xFFFF, dtype=np.uint16)
    pos = np.arange(p.n) // p.vns_pos
    for k in range(T):
        cn = ((pos[:, None] + np.arange(p.dv)[None, :]) * p.cns_pos + a[k]).ravel()      # CN of every edge, VN-major
        vn = np.repeat(np.arange(p.n), p.dv)
        order = np.lexsort((vn, cn))
        cn_s, vn_s = cn[order], vn[order]
        start = np.searchsorted(cn_s, np.arange(p.nk))
        slot = np.arange(cn_s.size) - start[cn_s]
        if slot.max() >= p.dc:
            raise ValueError("a CN has more than dc neighbours")
        out[k, cn_s, slot] = vn_s
    return out.view(np.int16)


def adj16_to_global(p, adj16):
    """Host: uint16 position-local ids [.., n, dv] → the int32 global CN ids of the reference's VNdegree."""
    a = np.ascontiguousarray(adj16).view(np.uint16).astype(np.int32)
    pos = (np.arange(p.n, dtype=np.int32) // p.vns_pos)[:, None] + np.arange(p.dv, dtype=np.int32)[None, :]
    return a + pos * p.cns_pos


def global_to_adj16(p, adj):
    """Host: int32 global CN ids → uint16 position-local ids (as int16 bit patterns for torch)."""
    adj = np.asarray(adj, dtype=np.int32)
    pos = (np.arange(p.n, dtype=np.int32) // p.vns_pos)[:, None] + np.arange(p.dv, dtype=np.int32)[None, :]
    loc = adj - pos * p.cns_pos
    if loc.min() < 0 or loc.max() >= p.cns_pos:
        raise ValueError("adjacency is not position-structured (edge i of a VN must land in CN position pos+i)")
    return loc.astype(np.uint16).view(np.int16)


# ------------------------------------------------------------------------------------------------
# decoding
# ------------------------------------------------------------------------------------------------
def full_bp(p, d_adj, d_chan, max_it=0, is_term=True, rows_cap=0, want_erased=False, counters=None):
    """decodeBP for a batch resident on the device.  Returns dict of device tensors:
    counters int32 [T,8] (+ rows int32 [T,rows_cap,3], erased int32 [T,nw] when asked)."""
    _require_gpu()
    T = d_adj.shape[0]
    assert d_adj.is_cuda and d_adj.dtype in (torch.int32, torch.int16) and d_adj.is_contiguous()
    assert d_chan.is_cuda and d_chan.dtype == torch.int32 and d_chan.is_contiguous()
    assert tuple(d_adj.shape[1:]) == (p.n, p.dv) and tuple(d_chan.shape) == (T, p.nw)
    dev = d_adj.device
    if counters is None:
        counters = torch.empty((T, NCOUNTERS), dtype=torch.int32, device=dev)
    rows = torch.zeros((T, rows_cap, 3), dtype=torch.int32, device=dev) if rows_cap > 0 else None
    erased = torch.empty((T, p.nw), dtype=torch.int32, device=dev) if want_erased else None
    fn = lib().scldpc_full_bp_device_adj16 if _is_adj16(d_adj) else lib().scldpc_full_bp_device
    ws, wsb, _keep = _workspace(WS_FULL_BP, p, T, dev, 1 if rows is not None else 0)
    check(fn(C.byref(p), T, d_adj.data_ptr(), d_chan.data_ptr(), int(max_it), 1 if is_term else 0,
             counters.data_ptr(), rows.data_ptr() if rows is not None else None, int(rows_cap),
             erased.data_ptr() if erased is not None else None, ws, wsb, _stream_ptr(dev)))
    return {"counters": counters, "rows": rows, "erased": erased}


def full_bp_fixpoint(p, d_adj, d_chan, is_term=True, want_erased=False, counters=None):
    """What unlimited decodeBP converges to, without its iteration count (scldpc_full_bp_fixpoint_device): counters as
    full_bp's except column 5 (barrier rounds of the kernel) — for runs with no iteration cap and no trajectory rows."""
    _require_gpu()
    T = d_adj.shape[0]
    assert d_adj.is_cuda and d_adj.dtype in (torch.int32, torch.int16) and d_adj.is_contiguous()
    assert d_chan.is_cuda and d_chan.dtype == torch.int32 and d_chan.is_contiguous()
    assert tuple(d_adj.shape[1:]) == (p.n, p.dv) and tuple(d_chan.shape) == (T, p.nw)
    dev = d_adj.device
    if counters is None:
        counters = torch.empty((T, NCOUNTERS), dtype=torch.int32, device=dev)
    erased = torch.empty((T, p.nw), dtype=torch.int32, device=dev) if want_erased else None
    fn = lib().scldpc_full_bp_fixpoint_device_adj16 if _is_adj16(d_adj) else lib().scldpc_full_bp_fixpoint_device
    ws, wsb, _keep = _workspace(WS_FULL_BP, p, T, dev, 0)
    check(fn(C.byref(p), T, d_adj.data_ptr(), d_chan.data_ptr(), 1 if is_term else 0, counters.data_ptr(),
             erased.data_ptr() if erased is not None else None, ws, wsb, _stream_ptr(dev)))
    return {"counters": counters, "rows": None, "erased": erased}


def full_bp_sock16_supported(p):
    """The (4,8) chain with at most 65536 CNs per trial, any number of VNs: sampler (CN -> socket table) + 4-bits-per-CN decoder."""
    return bool(lib().scldpc_sample_philox_sock16_supported(C.byref(p))) and bool(lib().scldpc_full_bp_sock16_supported(C.byref(p)))


def full_bp_fixpoint_cn16(p, d_adj16, d_cn16, d_chan, is_term=True, want_erased=False, counters=None, sockets=False):
    """scldpc_full_bp_fixpoint_device_cn16: full_bp_fixpoint's counters from the VN -> CN and CN -> VN tables
    (sockets=True: the CN -> socket table, scldpc_full_bp_fixpoint_device_sock16)."""
    _require_gpu()
    T = d_adj16.shape[0]
    assert d_adj16.is_cuda and d_adj16.dtype == torch.int16 and d_adj16.is_contiguous()
    assert d_cn16.is_cuda and d_cn16.dtype == torch.int16 and d_cn16.is_contiguous()
    assert d_chan.is_cuda and d_chan.dtype == torch.int32 and d_chan.is_contiguous()
    assert tuple(d_adj16.shape[1:]) == (p.n, p.dv) and tuple(d_cn16.shape) == (T, p.nk, p.dc) and tuple(d_chan.shape) == (T, p.nw)
    dev = d_adj16.device
    if counters is None:
        counters = torch.empty((T, NCOUNTERS), dtype=torch.int32, device=dev)
    erased = torch.empty((T, p.nw), dtype=torch.int32, device=dev) if want_erased else None
    fn = lib().scldpc_full_bp_fixpoint_device_sock16 if sockets else lib().scldpc_full_bp_fixpoint_device_cn16
    check(fn(C.byref(p), T, d_adj16.data_ptr(), d_cn16.data_ptr(), d_chan.data_ptr(),
             1 if is_term else 0, counters.data_ptr(), erased.data_ptr() if erased is not None else None, _stream_ptr(dev)))
    return {"counters": counters, "rows": None, "erased": erased}


def full_bp_cn16(p, d_adj16, d_cn16, d_chan, max_it=0, is_term=True, want_erased=False, counters=None, sockets=False,
                 rows_cap=0):
    """scldpc_full_bp_device_cn16: decodeBP with its iterations (count, cap, stop tests) from the VN -> CN and CN -> VN
    tables — every counter of full_bp (no trajectory rows)."""
    _require_gpu()
    T = d_adj16.shape[0]
    assert d_adj16.is_cuda and d_adj16.dtype == torch.int16 and d_adj16.is_contiguous()
    assert d_cn16.is_cuda and d_cn16.dtype == torch.int16 and d_cn16.is_contiguous()
    assert d_chan.is_cuda and d_chan.dtype == torch.int32 and d_chan.is_contiguous()
    assert tuple(d_adj16.shape[1:]) == (p.n, p.dv) and tuple(d_cn16.shape) == (T, p.nk, p.dc) and tuple(d_chan.shape) == (T, p.nw)
    dev = d_adj16.device
    if counters is None:
        counters = torch.empty((T, NCOUNTERS), dtype=torch.int32, device=dev)
    erased = torch.empty((T, p.nw), dtype=torch.int32, device=dev) if want_erased else None
    if rows_cap > 0:                                    # the trajectory build's rows (scldpc_full_bp_traj_device_*)
        rows = torch.empty((T, rows_cap, 3), dtype=torch.int32, device=dev)
        fn = lib().scldpc_full_bp_traj_device_sock16 if sockets else lib().scldpc_full_bp_traj_device_cn16
        check(fn(C.byref(p), T, d_adj16.data_ptr(), d_cn16.data_ptr(), d_chan.data_ptr(), int(max_it), 1 if is_term else 0,
                 counters.data_ptr(), rows.data_ptr(), int(rows_cap), erased.data_ptr() if erased is not None else None,
                 _stream_ptr(dev)))
        return {"counters": counters, "rows": rows, "erased": erased}
    fn = lib().scldpc_full_bp_device_sock16 if sockets else lib().scldpc_full_bp_device_cn16
    check(fn(C.byref(p), T, d_adj16.data_ptr(), d_cn16.data_ptr(), d_chan.data_ptr(), int(max_it), 1 if is_term else 0,
             counters.data_ptr(), erased.data_ptr() if erased is not None else None, _stream_ptr(dev)))
    return {"counters": counters, "rows": None, "erased": erased}


def cn_sockets(p, d_adj16, out=None):
    """scldpc_cn_sockets_device: the CN -> socket table int16 [T, nk, dc] (uint16 bit patterns) of a 2-byte VN -> CN table."""
    _require_gpu()
    T = d_adj16.shape[0]
    assert d_adj16.is_cuda and d_adj16.dtype == torch.int16 and d_adj16.is_contiguous()
    if out is None:
        out = torch.empty((T, p.nk, p.dc), dtype=torch.int16, device=d_adj16.device)
    check(lib().scldpc_cn_sockets_device(C.byref(p), T, d_adj16.data_ptr(), out.data_ptr(), _stream_ptr(d_adj16.device)))
    return out


def sw_bp(p, d_adj, d_chan, W, max_it, init_it=0, want_erased=False, counters=None, classical=False, ring=None, d_cn_sock=None):
    """decodeBP_SW for a batch resident on the device: square window (BPW:628-912) or, with classical=True, the
    classical window kept in BPF:627-897 (init_it unused).  ring: None = use the window-state-in-LDS kernel (sw_ring.hip)
    whenever it takes the ensemble (square window, 2-byte tables; the CN -> socket table is built on the fly unless
    d_cn_sock is given), False = the whole-chain kernel, True = insist on the ring kernel."""
    _require_gpu()
    T = d_adj.shape[0]
    use_ring = ring
    if ring is None or ring:
        ok = (not classical) and _is_adj16(d_adj) and bool(lib().scldpc_sw_bp_ring_supported(C.byref(p), int(W)))
        if ring and not ok:
            raise ScldpcError("the ring window kernel takes the square window on 2-byte tables of the (4,8) chain only")
        use_ring = ok
    if use_ring:
        assert d_chan.is_cuda and d_chan.dtype == torch.int32 and d_chan.is_contiguous()
        dev = d_adj.device
        if counters is None:
            counters = torch.empty((T, NCOUNTERS), dtype=torch.int32, device=dev)
        erased = torch.empty((T, p.nw), dtype=torch.int32, device=dev) if want_erased else None
        if d_cn_sock is None:
            d_cn_sock = cn_sockets(p, d_adj)
        check(lib().scldpc_sw_bp_ring_device(C.byref(p), T, d_adj.data_ptr(), d_cn_sock.data_ptr(), d_chan.data_ptr(), int(W),
                                             int(max_it), int(init_it), counters.data_ptr(),
                                             erased.data_ptr() if erased is not None else None, _stream_ptr(dev)))
        return {"counters": counters, "erased": erased}
    assert d_adj.is_cuda and d_adj.dtype in (torch.int32, torch.int16) and d_adj.is_contiguous()
    assert d_chan.is_cuda and d_chan.dtype == torch.int32 and d_chan.is_contiguous()
    dev = d_adj.device
    if counters is None:
        counters = torch.empty((T, NCOUNTERS), dtype=torch.int32, device=dev)
    erased = torch.empty((T, p.nw), dtype=torch.int32, device=dev) if want_erased else None
    ws, wsb, _keep = _workspace(WS_SW_BP, p, T, dev, int(W))
    if classical:
        fn = lib().scldpc_swc_bp_device_adj16 if _is_adj16(d_adj) else lib().scldpc_swc_bp_device
        check(fn(C.byref(p), T, d_adj.data_ptr(), d_chan.data_ptr(), int(W), int(max_it),
                 counters.data_ptr(), erased.data_ptr() if erased is not None else None, ws, wsb, _stream_ptr(dev)))
        return {"counters": counters, "erased": erased}
    fn = lib().scldpc_sw_bp_device_adj16 if _is_adj16(d_adj) else lib().scldpc_sw_bp_device
    check(fn(C.byref(p), T, d_adj.data_ptr(), d_chan.data_ptr(), int(W), int(max_it), int(init_it),
             counters.data_ptr(), erased.data_ptr() if erased is not None else None, ws, wsb, _stream_ptr(dev)))
    return {"counters": counters, "erased": erased}


def accumulate_run(d_counters, d_run, stop_frame_err=0):
    """plr_computation + willIstop in trial order; d_run int64 [NRUN] accumulated in place."""
    check(lib().scldpc_accumulate_run_device(d_counters.shape[0], d_counters.data_ptr(), int(stop_frame_err),
                                             d_run.data_ptr(), _stream_ptr(d_counters.device)))
    return d_run


def accumulate_peel(d_out, d_run, max_fuckups=0):
    """simulate_sc_ldpc's ordered bookkeeping + stop rule (PD:668-699) over a batch of peel_sweep rows; d_run int64
    [NPEELRUN] (order of _lib.PEELRUN_NAMES) accumulated in place."""
    check(lib().scldpc_accumulate_peel_device(d_out.shape[0], d_out.data_ptr(), int(max_fuckups), d_run.data_ptr(),
                                              _stream_ptr(d_out.device)))
    return d_run


def clear_channel_range(p, d_ch, vn_lo, vn_hi):
    """Soft doping (PD:176-183): VNs vn_lo .. vn_hi-1 of every trial become known."""
    check(lib().scldpc_clear_channel_range_device(C.byref(p), d_ch.shape[0], int(vn_lo), int(vn_hi), d_ch.data_ptr(),
                                                  _stream_ptr(d_ch.device)))
    return d_ch


def new_run(device="cuda:0"):
    return torch.zeros(NRUN, dtype=torch.int64, device=device)


def unpack_bits(words, n):
    """uint32/int32 words [.., nw] → uint8 [.., n] (host)."""
    w = np.ascontiguousarray(words).view(np.uint32)
    bits = np.unpackbits(w.view(np.uint8).reshape(*w.shape[:-1], -1), axis=-1, bitorder="little")
    return bits[..., :n]


def pack_bits(bits):
    """uint8 [.., n] → uint32 words [.., nw] (host)."""
    bits = np.asarray(bits, dtype=np.uint8)
    n = bits.shape[-1]
    pad = (-n) % 32
    if pad:
        bits = np.concatenate([bits, np.zeros(bits.shape[:-1] + (pad,), np.uint8)], axis=-1)
    return np.packbits(bits, axis=-1, bitorder="little").view(np.uint32)


# ------------------------------------------------------------------------------------------------
# Python peeling path (PD)
# ------------------------------------------------------------------------------------------------
def peel_sweep(p, d_adj, d_chan, total_size, sweep_start=0, lost_lo=0, lost_hi=None, want_lost=False):
    """One trial of simulate_sc_ldpc's loop body per batch entry (PD:650-691).  Returns dict with
    out int32 [T,8] ([0] #lost, [1] #lost_exp, [2] #blocks_failed_exp, [5] rounds, [7] #erased) and `lost` bits."""
    _require_gpu()
    T = d_adj.shape[0]
    assert d_adj.is_cuda and d_adj.dtype in (torch.int32, torch.int16) and d_adj.is_contiguous()
    assert d_chan.is_cuda and d_chan.dtype == torch.int32 and d_chan.is_contiguous()
    dev = d_adj.device
    out = torch.empty((T, 8), dtype=torch.int32, device=dev)
    lost = torch.empty((T, p.nw), dtype=torch.int32, device=dev) if want_lost else None
    fn = lib().scldpc_peel_sweep_device_adj16 if _is_adj16(d_adj) else lib().scldpc_peel_sweep_device
    ws, wsb, _keep = _workspace(WS_PEEL_SWEEP, p, T, dev, 1 if _is_adj16(d_adj) else 0)
    check(fn(C.byref(p), T, d_adj.data_ptr(), d_chan.data_ptr(), int(total_size), int(sweep_start), int(lost_lo),
             int(total_size if lost_hi is None else lost_hi), out.data_ptr(),
             lost.data_ptr() if lost is not None else None, ws, wsb, _stream_ptr(dev)))
    return {"out": out, "lost": lost}


def peel_pick(p, d_adj, d_chan, total_size, num_steps, mt_state=None, seed=0, trial0=0, want_r1=True, moments=None):
    """One trial of simulate_peeling_decoder_ldpc's loop body per batch entry (PD:750-785).
    mt_state: int32/uint32-as-int32 tensor [T,625] (CPython MT19937 state, updated in place) or None (Philox)."""
    _require_gpu()
    T = d_adj.shape[0]
    assert d_adj.is_cuda and d_adj.dtype in (torch.int32, torch.int16) and d_adj.is_contiguous()
    dev = d_adj.device
    out = torch.empty((T, 4), dtype=torch.int32, device=dev)
    r1 = torch.empty((T, num_steps + 1), dtype=torch.int32, device=dev) if want_r1 else None
    if mt_state is not None:
        assert mt_state.is_cuda and mt_state.dtype == torch.int32 and tuple(mt_state.shape) == (T, 625)
    fn = lib().scldpc_peel_pick_device_adj16 if _is_adj16(d_adj) else lib().scldpc_peel_pick_device
    ws, wsb, _keep = _workspace(WS_PEEL_PICK, p, T, dev, int(total_size), 1 if mt_state is not None else 0)
    check(fn(C.byref(p), T, d_adj.data_ptr(), d_chan.data_ptr(), int(total_size), int(num_steps),
             mt_state.data_ptr() if mt_state is not None else None, int(seed), int(trial0),
             r1.data_ptr() if r1 is not None else None, moments.data_ptr() if moments is not None else None,
             out.data_ptr(), ws, wsb, _stream_ptr(dev)))
    return {"out": out, "r1": r1, "moments": moments}


def r1_moments(d_r1, moments=None):
    """(cnt, Σr1, Σr1²) per step over a batch of trajectories, accumulated into int64 [3, steps+1]."""
    T, ncols = d_r1.shape
    if moments is None:
        moments = torch.zeros((3, ncols), dtype=torch.int64, device=d_r1.device)
    check(lib().scldpc_r1_moments_device(T, ncols, d_r1.data_ptr(), moments.data_ptr(), _stream_ptr(d_r1.device)))
    return moments


# ------------------------------------------------------------------------------------------------
# streaming mode (main_streaming, BPF:1934-2054)
# ------------------------------------------------------------------------------------------------
STREAM_COUNTERS = ("num_erasures", "num_blocks_err", "num_erasures_exp", "num_blocks_err_exp", "num_bits_generated",
                   "num_blocks_generated", "num_bits_generated_exp", "num_blocks_generated_exp", "positions", "generated")


class Streams:
    """nstreams independent doped SC-LDPC streams on the device, each a circular buffer of p.L positions."""

    def __init__(self, p, nstreams, seed, eps, W, doped=(), stream0=0, device="cuda:0"):
        _require_gpu()
        nbytes = lib().scldpc_stream_state_bytes(C.byref(p), int(W))
        if nbytes < 0:
            check(int(nbytes))
        self.p, self.n, self.seed, self.eps, self.W, self.stream0 = p, nstreams, seed, eps, W, stream0
        self.doped = tuple(doped)
        self.state = torch.zeros((nstreams, nbytes), dtype=torch.uint8, device=device)
        self.counters = torch.zeros((nstreams, 10), dtype=torch.int64, device=device)

    def run(self, npos, trace=False):
        """Decode npos further positions on every stream; returns (counters int64 [n,10], trace int32 [n,npos,10] | None)."""
        tr = torch.empty((self.n, npos, 10), dtype=torch.int32, device=self.state.device) if trace else None
        darr, dptr = _lib.doped_array(self.doped)
        check(lib().scldpc_stream_run_device(C.byref(self.p), self.n, int(self.seed), int(self.stream0), float(self.eps),
                                             int(self.W), darr.size, dptr, int(npos), self.state.data_ptr(),
                                             self.counters.data_ptr(), tr.data_ptr() if tr is not None else None,
                                             _stream_ptr(self.state.device)))
        return self.counters, tr


def stream_glibc_inputs(p, seed, eps, doped, npos_gen):
    """Host: main_streaming's draws for the first npos_gen generated positions of one stream after srandom(seed)
    (scldpc_stream_glibc_inputs_host): (inter uint16 [npos_gen + dv - 1, S], chan uint32 [npos_gen, wpp])."""
    S, wpp = p.cns_pos * p.dc, (p.vns_pos + 31) // 32
    inter = np.empty((npos_gen + p.dv - 1, S), dtype=np.uint16)
    chan = np.empty((npos_gen, wpp), dtype=np.uint32)
    darr, dptr = _lib.doped_array(doped)
    check(lib().scldpc_stream_glibc_inputs_host(C.byref(p), int(seed) & 0xFFFFFFFF, float(eps), darr.size, dptr, int(npos_gen),
                                                inter.ctypes.data, chan.ctypes.data))
    return inter, chan


class GlibcStreamRun:
    """main_streaming on the reference's OWN stream (BPF:1934-2054): ONE srandom(seed) for the whole run (BPF:1942-1945),
    the ε points back to back with random() carried from point to point; the draws are replayed on the host piece by piece
    (scldpc_stream_glibc_next_host) and decoded on the device (scldpc_stream_run_device_inputs_at).  One stream, as the
    reference runs it."""

    def __init__(self, p, seed, W, doped=(), device="cuda:0"):
        _require_gpu()
        self.p, self.W, self.doped, self.device = p, int(W), tuple(doped), device
        nbytes = lib().scldpc_glibc_state_bytes(C.byref(p))
        if nbytes < 0:
            check(int(nbytes))
        self._host = np.zeros(nbytes, dtype=np.uint8)
        check(lib().scldpc_glibc_state_init(C.byref(p), int(seed) & 0xFFFFFFFF, self._host.ctypes.data))     # srandom(seed)
        self.S, self.wpp = p.cns_pos * p.dc, (p.vns_pos + 31) // 32
        self.st = None

    def new_point(self, eps):
        """inizio_sim (perm_code := identity, BPF:308-311) + a fresh buffer (initialize_arrays_circular, BPF:1998)."""
        check(lib().scldpc_glibc_state_reset_perm(C.byref(self.p), self._host.ctypes.data))
        self.eps, self.done = float(eps), 0
        self.st = Streams(self.p, 1, 0, 0.0, self.W, self.doped, device=self.device)

    def _draw(self, ninit, gpos0, npos):
        inter = np.empty((ninit + npos, self.S), dtype=np.uint16)
        chan = np.empty((max(npos, 1), self.wpp), dtype=np.uint32)
        darr, dptr = _lib.doped_array(self.doped)
        check(lib().scldpc_stream_glibc_next_host(C.byref(self.p), self._host.ctypes.data, self.eps, darr.size, dptr, int(ninit),
                                                  int(gpos0), int(npos), inter.ctypes.data, chan.ctypes.data))
        return inter, chan[:npos]

    def run(self, npos, stop=None):
        """Decode npos further positions.  Returns the trace rows int64 [k, 10] (position, value of decodeBP_SW_circular, the
        eight running counters) of the positions that count: all npos, or — with stop(counters) -> bool — up to and including
        the first position at which main_streaming's rule trips (BPF:2033); the host stream is then left exactly where the
        reference leaves it (it does not generate beyond that position), ready for the next point."""
        p, dv, half = self.p, self.p.dv, self.p.L // 2
        snap = self._host.copy()
        first = self.done == 0
        g0 = 0 if first else half + self.done
        ng = (half if first else 0) + npos
        inter, chan = self._draw(dv - 1 if first else 0, g0, ng)
        d_inter = torch.zeros((1, ng + dv - 1, self.S), dtype=torch.int16, device=self.device)
        d_inter[0, (0 if first else dv - 1):] = torch.from_numpy(inter.view(np.int16)).to(self.device)
        d_chan = torch.from_numpy(np.ascontiguousarray(chan).view(np.int32)).to(self.device).reshape(1, ng, self.wpp)
        tr = torch.empty((1, npos, 10), dtype=torch.int32, device=self.device)
        darr, dptr = _lib.doped_array(self.doped)
        check(lib().scldpc_stream_run_device_inputs_at(C.byref(p), 1, self.W, darr.size, dptr, int(npos),
                                                       self.st.state.data_ptr(), self.st.counters.data_ptr(), tr.data_ptr(),
                                                       d_inter.data_ptr(), d_chan.data_ptr(), int(g0), int(ng), int(self.done),
                                                       _stream_ptr(self.st.state.device)))
        rows = tr[0].cpu().numpy().astype(np.int64)
        used = npos
        if stop is not None:
            for k in range(npos):
                if stop(rows[k, 2:]):
                    used = k + 1
                    break
        if used < npos:
            # the reference generated L/2 + (positions decoded before the tripping one) positions at this point: rewind and
            # draw exactly those
            self._host[:] = snap
            self._draw(dv - 1 if first else 0, g0, (half if first else 0) + used - 1)
        elif stop is not None and stop(rows[npos - 1, 2:]):
            self._host[:] = snap
            self._draw(dv - 1 if first else 0, g0, (half if first else 0) + used - 1)
        self.done += used
        return rows[:used]


class InputStreams(Streams):
    """Streams decoded from given per-position inputs (same-input mode): inter [nstreams, G + dv - 1, S] uint16 and
    chan [nstreams, G, wpp] uint32 as stream_glibc_inputs produces them, G = positions generated."""

    def __init__(self, p, inter, chan, W, doped=(), device="cuda:0"):
        inter, chan = np.ascontiguousarray(inter, dtype=np.uint16), np.ascontiguousarray(chan, dtype=np.uint32)
        super().__init__(p, inter.shape[0], 0, 0.0, W, doped, device=device)
        self.G = chan.shape[1]
        assert inter.shape[1] == self.G + p.dv - 1 and inter.shape[2] == p.cns_pos * p.dc
        self.d_inter = torch.from_numpy(inter.view(np.int16)).to(device)
        self.d_chan = torch.from_numpy(chan.view(np.int32)).to(device)
        self.done = 0

    def run(self, npos, trace=False):
        tr = torch.empty((self.n, npos, 10), dtype=torch.int32, device=self.state.device) if trace else None
        darr, dptr = _lib.doped_array(self.doped)
        check(lib().scldpc_stream_run_device_inputs(C.byref(self.p), self.n, int(self.W), darr.size, dptr, int(npos),
                                                    self.state.data_ptr(), self.counters.data_ptr(),
                                                    tr.data_ptr() if tr is not None else None, self.d_inter.data_ptr(),
                                                    self.d_chan.data_ptr(), int(self.G), int(self.done),
                                                    _stream_ptr(self.state.device)))
        self.done += npos
        return self.counters, tr
