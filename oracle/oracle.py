"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (the CPU restatement of the
reference's C simulators, see scldpc_oracle.h) plus a parser for the records printed by the
real-reference driver binaries in oracle/_ref/ (oracle/ref_driver_tail.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REFERENCE_ROOT = "/root/reference"

DEC_BP_LITERAL, DEC_BP_PEEL, DEC_SW_LITERAL, DEC_SW_PEEL, DEC_SW_CLASSICAL = 0, 1, 2, 3, 4


class Params(C.Structure):
    _fields_ = [("dv", C.c_int), ("dc", C.c_int), ("L", C.c_int), ("cns_pos", C.c_int), ("vns_pos", C.c_int)]

    @property
    def n(self):
        return self.vns_pos * self.L

    @property
    def nk(self):
        return (self.L + self.dv - 1) * self.cns_pos


class Result(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("num_erasures", "num_blocks_err", "num_erasures_exp",
                                         "num_blocks_err_exp", "num_erasures_p1", "iterations", "status")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Rng(C.Structure):
    _fields_ = [("r", C.c_int32 * 31), ("f", C.c_int), ("b", C.c_int)]


def build(with_reference=True):
    """Compile liboracle.so (always) and oracle/_ref (when /root/reference is present)."""
    target = "all" if (with_reference and os.path.isdir(REFERENCE_ROOT)) else "oracle"
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(with_reference=False)
        L = C.CDLL(LIB_PATH)
        P = C.POINTER
        i32p, u8p = P(C.c_int32), P(C.c_uint8)
        L.orc_srandom.argtypes = [P(Rng), C.c_uint]
        L.orc_random.argtypes = [P(Rng)]
        L.orc_random.restype = C.c_int32
        L.orc_fnv1a.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_fnv1a.restype = C.c_uint64
        L.orc_perm_identity.argtypes = [P(Params), i32p]
        L.orc_generate_code.argtypes = [P(Params), P(Rng), i32p, i32p, i32p, i32p]
        L.orc_channel.argtypes = [P(Params), P(Rng), C.c_double, C.c_int, P(C.c_int), u8p]
        common = [P(Params), i32p, i32p, i32p, u8p]
        L.orc_decode_bp_literal.argtypes = common + [C.c_int, C.c_int, u8p, C.c_void_p, C.c_int, P(Result)]
        L.orc_decode_bp_peel.argtypes = common + [C.c_int, C.c_int, u8p, C.c_void_p, C.c_int, P(Result)]
        L.orc_decode_sw_literal.argtypes = common + [C.c_int, C.c_int, C.c_int, C.c_int, u8p, P(Result)]
        L.orc_decode_sw_peel.argtypes = common + [C.c_int, C.c_int, C.c_int, u8p, P(Result)]
        L.orc_expurgate.argtypes = [P(Params), i32p, i32p, i32p, u8p, C.c_int, i32p, i32p, i32p]
        L.orc_trial.argtypes = [P(Params), C.c_uint, C.c_double, C.c_int, P(C.c_int), C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, P(Result), P(C.c_uint64), i32p, i32p, u8p, u8p,
                                C.c_void_p, C.c_int]
        L.orc_philox4x32_10.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
        L.orc_sample_philox.argtypes = [P(Params), C.c_uint64, C.c_uint64, C.c_double, C.c_int, P(C.c_int),
                                        i32p, P(C.c_uint32)]
        L.orc_sample_philox_ens.argtypes = [P(Params), C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_int, P(C.c_int),
                                            i32p, P(C.c_uint32)]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def glibc_random_stream(seed, count):
    g = Rng()
    lib().orc_srandom(C.byref(g), seed)
    return np.array([lib().orc_random(C.byref(g)) for _ in range(count)], dtype=np.int64)


class Graph:
    """One sampled code: VN-side table + CN-side CSR, as the oracle lays them out."""

    def __init__(self, params, vn_adj, cn_ptr, cn_adj):
        self.params, self.vn_adj, self.cn_ptr, self.cn_adj = params, vn_adj, cn_ptr, cn_adj

    @staticmethod
    def from_vn_adj(params, vn_adj):
        """Build the CN side in the reference's insertion order (VN-major)."""
        vn_adj = np.ascontiguousarray(vn_adj, dtype=np.int32).reshape(params.n, params.dv)
        nk = params.nk
        flat = vn_adj.reshape(-1)
        order = np.argsort(flat, kind="stable")
        cn_adj = (order // params.dv).astype(np.int32)
        cn_ptr = np.zeros(nk + 1, dtype=np.int32)
        np.cumsum(np.bincount(flat, minlength=nk), out=cn_ptr[1:])
        return Graph(params, vn_adj, cn_ptr, cn_adj)


def sample_trial_inputs(params, seed, eps, doped=()):
    """identity perm_code → srandom(seed) → generate_code → channel_doped (ref_driver_tail.c order)."""
    L = lib()
    n, nk, dv = params.n, params.nk, params.dv
    perm = np.empty(params.cns_pos * params.dc, dtype=np.int32)
    vn_adj = np.empty((n, dv), dtype=np.int32)
    cn_ptr = np.empty(nk + 1, dtype=np.int32)
    cn_adj = np.empty(n * dv, dtype=np.int32)
    chan = np.empty(n, dtype=np.uint8)
    g = Rng()
    L.orc_perm_identity(C.byref(params), _p(perm, C.c_int32))
    L.orc_srandom(C.byref(g), seed)
    L.orc_generate_code(C.byref(params), C.byref(g), _p(perm, C.c_int32), _p(vn_adj, C.c_int32),
                        _p(cn_ptr, C.c_int32), _p(cn_adj, C.c_int32))
    d = (C.c_int * max(1, len(doped)))(*doped)
    L.orc_channel(C.byref(params), C.byref(g), eps, len(doped), d, _p(chan, C.c_uint8))
    return Graph(params, vn_adj, cn_ptr, cn_adj), chan


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


ENSEMBLES = {"olmos": 0, "tail_biting": 1, "protograph": 2}


def sample_philox(params, seed, trial, eps, doped=(), ensemble="olmos"):
    """CPU twin of scldpc_sample_philox_device(_ensemble) for one trial: (vn_adj int32 [n,dv], chan_bits uint32 [nw])."""
    vn_adj = np.empty((params.n, params.dv), dtype=np.int32)
    chan = np.empty((params.n + 31) // 32, dtype=np.uint32)
    d = (C.c_int * max(1, len(doped)))(*doped)
    lib().orc_sample_philox_ens(C.byref(params), ENSEMBLES[ensemble], seed, trial, eps, len(doped), d,
                                _p(vn_adj, C.c_int32), _p(chan, C.c_uint32))
    return vn_adj, chan


ROW_DTYPE = np.dtype([("deg1", np.int32), ("recovered", np.int32), ("first_pos", np.int32)])


def decode_bp(graph, chan, max_it=0, is_term=1, literal=True, rows_cap=0):
    p = graph.params
    res = Result()
    erased = np.empty(p.n, dtype=np.uint8)
    rows = np.zeros(max(rows_cap, 1), dtype=ROW_DTYPE)
    fn = lib().orc_decode_bp_literal if literal else lib().orc_decode_bp_peel
    chan = np.ascontiguousarray(chan, dtype=np.uint8)
    fn(C.byref(p), _p(graph.vn_adj, C.c_int32), _p(graph.cn_ptr, C.c_int32), _p(graph.cn_adj, C.c_int32),
       _p(chan, C.c_uint8), max_it, is_term, _p(erased, C.c_uint8),
       rows.ctypes.data if rows_cap else None, rows_cap, C.byref(res))
    return res.as_dict(), erased, rows[:min(rows_cap, res.iterations)]


def decode_sw(graph, chan, W, max_it, init_it=0, literal=True, square=True):
    p = graph.params
    res = Result()
    erased = np.empty(p.n, dtype=np.uint8)
    chan = np.ascontiguousarray(chan, dtype=np.uint8)
    init_it = init_it if init_it else max_it
    args = [C.byref(p), _p(graph.vn_adj, C.c_int32), _p(graph.cn_ptr, C.c_int32), _p(graph.cn_adj, C.c_int32),
            _p(chan, C.c_uint8), W, max_it, init_it]
    if literal:
        lib().orc_decode_sw_literal(*args, 1 if square else 0, _p(erased, C.c_uint8), C.byref(res))
    else:
        assert square
        lib().orc_decode_sw_peel(*args, _p(erased, C.c_uint8), C.byref(res))
    return res.as_dict(), erased


def trial(params, seed, eps, decoder=DEC_BP_LITERAL, W=0, max_it=0, init_it=0, is_term=1, doped=(),
          rows_cap=0, want_arrays=False):
    """One self-contained trial; returns dict with counters, hashes, optional arrays and rows."""
    L = lib()
    res = Result()
    hashes = (C.c_uint64 * 3)()
    nch = C.c_int32()
    n, dv = params.n, params.dv
    vn_adj = np.empty((n, dv), dtype=np.int32) if want_arrays else None
    chan = np.empty(n, dtype=np.uint8) if want_arrays else None
    erased = np.empty(n, dtype=np.uint8) if want_arrays else None
    rows = np.zeros(max(rows_cap, 1), dtype=ROW_DTYPE)
    d = (C.c_int * max(1, len(doped)))(*doped)
    if decoder in (DEC_SW_LITERAL, DEC_SW_PEEL) and not init_it:
        init_it = max_it
    L.orc_trial(C.byref(params), seed, eps, len(doped), d, decoder, W, max_it, init_it, is_term,
                C.byref(res), hashes, C.byref(nch),
                _p(vn_adj, C.c_int32) if want_arrays else None,
                _p(chan, C.c_uint8) if want_arrays else None,
                _p(erased, C.c_uint8) if want_arrays else None,
                rows.ctypes.data if rows_cap else None, rows_cap)
    out = res.as_dict()
    out.update(nch=int(nch.value), hg=int(hashes[0]), hc=int(hashes[1]), he=int(hashes[2]))
    if want_arrays:
        out.update(vn_adj=vn_adj, chan=chan, erased=erased)
    if rows_cap:
        out["rows"] = rows[:min(rows_cap, res.iterations)]
    return out


# ---------------------------------------------------------------------------------------------
# Real reference (oracle/_ref) — only in the container that has /root/reference built.
# ---------------------------------------------------------------------------------------------
def ref_binary(variant, M, L):
    return os.path.join(REF_DIR, f"ref_{variant}_M{M}_L{L}")


def have_ref(variant, M, L):
    return os.path.exists(ref_binary(variant, M, L))


def run_ref(variant, M, L, T, seed0, eps, max_it=1000000, init_it=0, W=0, is_term=1, dump=False, doped=(),
            whole_run=False):
    """Run a reference driver; returns (header dict, list of per-trial dicts, run counters or None)."""
    cmd = [ref_binary(variant, M, L), str(-T if whole_run else T), str(seed0), repr(float(eps)), str(max_it),
           str(init_it), str(W), str(is_term), "1" if dump else "0", str(len(doped))] + [str(d) for d in doped]
    txt = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    return parse_ref_output(txt)


def _kv(tokens):
    out = {}
    for tok in tokens:
        k, v = tok.split("=", 1)
        if k in ("hg", "hc", "he"):
            out[k] = int(v, 16)
        elif k == "eps":
            out[k] = float(v)
        else:
            out[k] = int(v)
    return out


def parse_ref_output(txt):
    hdr, trials, run = None, [], None
    lines = txt.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("HDR "):
            hdr = _kv(ln.split()[1:])
        elif ln.startswith("TRIAL "):
            trials.append(_kv(ln.split()[1:]))
        elif ln.startswith("RUN "):
            run = _kv(ln.split()[1:])
        elif ln == "TRAJ_BEGIN":
            rows = []
            i += 1
            while lines[i] != "TRAJ_END":
                if lines[i].strip():
                    it, d1, rec, fp = (int(x) for x in lines[i].split("\t"))
                    assert it == len(rows)
                    rows.append((d1, rec, fp))
                i += 1
            trials[-1]["rows"] = np.array(rows, dtype=ROW_DTYPE)
        elif ln.startswith("VNADJ"):
            trials[-1]["vn_adj"] = np.array(ln.split()[1:], dtype=np.int32).reshape(hdr["n"], hdr["dv"])
        elif ln.startswith("CNDEG"):
            trials[-1]["cn_deg"] = np.array(ln.split()[1:], dtype=np.int32)
        elif ln.startswith("CHAN "):
            trials[-1]["chan"] = np.frombuffer(ln[5:].encode(), dtype=np.uint8) - ord("0")
        elif ln.startswith("ERASED "):
            trials[-1]["erased"] = np.frombuffer(ln[7:].encode(), dtype=np.uint8) - ord("0")
        i += 1
    return hdr, trials, run


# ---------------------------------------------------------------------------------------------
# streaming mode (oracle/scldpc_stream_oracle.c; reference: main_streaming, BPF:1934-2054)
# ---------------------------------------------------------------------------------------------
class Stream:
    """One stream: circular buffer of p.L positions.  rng_mode 0 = glibc (the reference's draws), 1 = Philox twin of
    the device kernel; decoder 0 = literal messages, 1 = node-level model."""
    FIELDS = ("pos", "nep", "ne", "be", "ee", "bee", "gb", "gbl", "gbe", "gble")

    def __init__(self, params, seed, eps, W, doped=(), rng_mode=0, decoder=0, sid=0):
        L = lib()
        L.orc_stream_new.restype = C.c_void_p
        L.orc_stream_new.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_int,
                                     C.c_int, C.POINTER(C.c_int)]
        L.orc_stream_step.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_stream_last_erased.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.orc_stream_free.argtypes = [C.c_void_p]
        d = (C.c_int * max(1, len(doped)))(*doped)
        self.params = params
        self._h = L.orc_stream_new(C.byref(params), rng_mode, decoder, seed, sid, eps, W, len(doped), d)

    def step(self):
        out = (C.c_int32 * 10)()
        lib().orc_stream_step(self._h, out)
        return dict(zip(self.FIELDS, (int(x) for x in out)))

    def last_erased(self):
        buf = np.zeros(self.params.vns_pos, dtype=np.uint8)
        ok = lib().orc_stream_last_erased(self._h, _p(buf, C.c_uint8))
        return buf if ok else None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_stream_free(self._h)
            self._h = None


def run_ref_stream(M, L, P, seed, eps, W, doped=(), dump=False):
    """Run oracle/_ref/ref_stream_M<M>_L<L>; returns list of per-position dicts (+ 'erased' when dumped)."""
    cmd = [os.path.join(REF_DIR, f"ref_stream_M{M}_L{L}"), str(P), str(seed), repr(float(eps)), str(W),
           "1" if dump else "0", str(len(doped))] + [str(d) for d in doped]
    txt = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    out = []
    for ln in txt.split("\n"):
        if ln.startswith("POS "):
            out.append({k: int(v) for k, v in (tok.split("=") for tok in ln.split()[1:])})
        elif ln.startswith("ERASED "):
            out[-1]["erased"] = np.frombuffer(ln[7:].encode(), dtype=np.uint8) - ord("0")
    return out
