"""Test doubles for the host-logic tests: a Simulator whose device work is replaced by a deterministic
function of the frame index, so that sharding / ordered-stop / file-format logic can run on CPU
(world_size 1 and 2 under gloo).  Lives in tests/: the product has no CPU path."""
import numpy as np
import torch

from fl_scaling_sc_ldpc_amd import bp_decoding as B
from fl_scaling_sc_ldpc_amd import engine as E


def fake_counters(sim, frame_idx, n, L):
    """Per-trial counter rows as a pure function of (point, frame index)."""
    rows = np.zeros((len(frame_idx), E.NCOUNTERS), dtype=np.int32)
    for k, f in enumerate(frame_idx):
        h = (int(f) * 2654435761 + sim * 40503 + 12345) & 0xFFFFFFFF
        fail = (h >> 7) % 3 != 0
        ne = 1 + h % (n // 2) if fail else 0
        rows[k] = [ne, (1 + h % L) if fail else 0, max(0, ne - 2 * ((h >> 3) % 2)) if fail else 0,
                   1 if fail and ne > 2 else 0, ne % 7 if fail else 0, 5 + h % 200, 0, n // 2]
    return rows


def numpy_accumulate(cnt, run, stop):
    """Literal plr_computation + willIstop loop (BPF:1503-1520, 2140-2144) in numpy."""
    run = np.array(run, dtype=np.int64)
    for c in cnt:
        if stop > 0 and run[1] >= stop:
            break
        ne, be, ee, bee, p1, it = (int(x) for x in c[:6])
        if ne > 0:
            run[0] += ne; run[1] += 1; run[3] += be
        if ee > 0:
            run[4] += ee; run[5] += 1; run[6] += bee
        if p1 > 0:
            run[2] += 1
        run[7] += 1; run[8] += it
    return run


class FakeSimulator(B.Simulator):
    def __init__(self, p, batch=8, **kw):
        self._frames = None
        self.frames_decoded = 0                 # frames this rank decoded (sharding tests)
        kw.pop("device", None)
        super().__init__(p, batch=batch, device="cpu", **kw)

    def _alloc(self):
        self.d_cnt = torch.zeros((self.batch, E.NCOUNTERS), dtype=torch.int32)

    bad_frames = ()                             # (sim, frame) pairs whose status word reports a broken invariant

    def fill_batch(self, sim, eps, frame0, nb):
        self._frames = (sim, np.arange(frame0, frame0 + nb))
        self.frames_decoded += nb

    def decode_batch(self, nb, want_rows=False):
        sim, idx = self._frames
        key = sim + 1000 * self.index            # the replica (INDEX) is part of the trial key
        cnt = fake_counters(key, idx, self.p.n, self.p.L)
        for k, f in enumerate(idx):
            if (sim, int(f)) in self.bad_frames:
                cnt[k, E.COUNTER_NAMES.index("status")] = -1
        self.d_cnt[:nb] = torch.from_numpy(cnt)
        rows = None
        if want_rows:                           # bp_traj: `iterations` rows of (deg1, recovered, first erased position)
            rows = torch.zeros((nb, self.rows_cap, 3), dtype=torch.int32)
            for k, f in enumerate(idx):
                it = int(cnt[k, 5])
                r = np.arange(it, dtype=np.int64)
                rows[k, :it] = torch.from_numpy(np.stack([(r * 7 + f + key) % 97, (r * 3 + f) % 89, r % self.p.L], 1).astype(np.int32))
        return {"counters": self.d_cnt[:nb], "rows": rows, "erased": None}

    def _new_run(self):
        return torch.zeros(E.NRUN, dtype=torch.int64)

    def _accumulate(self, allcnt, run, stop):
        run.copy_(torch.from_numpy(numpy_accumulate(allcnt.numpy(), run.numpy(), stop)))
        return run


class FakeStreams:
    """Stand-in for engine.Streams (streaming mode): cumulative per-stream counters as a pure function of the global
    stream id and the number of positions decoded so far."""

    def __init__(self, p, nstreams, seed, eps, W, doped, stream0=0, device=None):
        self.ids = np.arange(stream0, stream0 + nstreams, dtype=np.int64)
        self.done = 0
        self.vns = p.vns_pos
        self.cnt = np.zeros((nstreams, 10), dtype=np.int64)      # eight counters, positions decoded, positions generated

    def run(self, npos):
        for k in range(self.done, self.done + npos):
            h = (self.ids * 2654435761 + k * 40503 + 977) & 0xFFFFFFFF
            bad = (h >> 5) % 11 == 0
            ne = np.where(bad, 1 + h % 37, 0)
            ee = np.where(bad & (ne > 2), ne, 0)
            self.cnt[:, :8] += np.stack([ne, bad.astype(np.int64), ee, (ee > 0).astype(np.int64),
                                         np.full_like(ne, self.vns), np.ones_like(ne), np.full_like(ne, self.vns),
                                         np.ones_like(ne)], axis=1)
        self.done += npos
        return torch.from_numpy(self.cnt.copy()), None
