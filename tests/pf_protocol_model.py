"""Host-side MODEL of the sharded filter's DEFAULT exchange -- the device-side protocol of csrc/pf_auto.hip / pf_peers.hip / pf_device.h
-- so that a change of the protocol is caught in the build container, before a GPU box is spent on it.
Test infrastructure: nothing under slam.jl_amd/ imports it.

What is restated, with the source it follows:
  * the canonical statistics tree (pf_device.h: WRec, wrec_k, wrec_combine4, wrec_wave): leaf = 64 consecutive particles (largest
    log-weight m, k = ceil(m / ln 2), e_i = exp(logw_i - k ln 2), sums by the xor butterfly), node = four children rescaled by exact
    powers of two and added left to right, an absent child is the identity;
  * the 1024-particle RECORDS a rank writes into EVERY rank's inbox (pf_auto.hip: pf_auto_tail, `xpeers`): {m, s1, s2, tag} at index
    rank * ceil(n_local / 1024) + j of the step's parity, tag = wrec_hash(values) ^ key(step) ^ 0x5851F42D4C957F2D ^ index * 0xD6E8FEB86659FD93;
    a reader takes a record only when its tag fits (a stale, half written or torn record is polled again);
  * the reduction every rank makes of ALL records: passes of 1024 records -> the pass nodes -> the radix-4 tree over the pass nodes
    -> the root; shift = K ln 2 + log s1, Neff = s1^2 / s2, the decision Neff < frac * N (or forced);
  * the GATE of a resampling step (pf_peers.hip: pf_peer_gate_kernel): ready[r] = the last resampling step whose step kernel rank
    r has completed; a rank reads its peers' weights only when every ready word has reached the step;
  * the GONE word (pf_device.h: pf_peer_gone): a rank that destroys its handle says so in every peer's inbox; every kernel that
    would touch peer memory then stops with PF_ERR_PEER.
The transport of the model is whatever the test gives it (tests/test_pf_protocol_gloo.py: gloo all-gathers stand in for the posted
stores over xGMI); exp / log are NumPy's, so the model is compared with ITSELF across rank counts (bit for bit), not with the GPU.
"""
import math

import numpy as np

MASK = (1 << 64) - 1
LN2 = 0.693147180559945309417232121458
INV_LN2 = 1.442695040888963407359924681002
KEY_MUL, KEY_ADD = 0x9E3779B97F4A7C15, 0x632BE59BD9B4E019       # pf_device.h: part_key
XKEY = 0x5851F42D4C957F2D                                        # pf_auto.hip: the exchange's key differs from the lines' key
INDEX_MUL = 0xD6E8FEB86659FD93                                   # pf_auto.hip: the record's index enters its tag
PF_ERR_PEER = 3
NEG = -math.inf


def part_key(seq):
    return (seq * KEY_MUL + KEY_ADD) & MASK


def _bits(x):
    return int(np.float64(x).view(np.uint64))


def _rotl(b, r):
    return ((b << r) | (b >> (64 - r))) & MASK


def wrec_hash(rec):
    b0, b1, b2 = _bits(rec[0]), _bits(rec[1]), _bits(rec[2])
    return _rotl(b0, 7) ^ _rotl(b1, 23) ^ _rotl(b2, 41)


def record_tag(rec, seq, index):
    return wrec_hash(rec) ^ part_key(seq) ^ XKEY ^ ((index * INDEX_MUL) & MASK)


EMPTY = (NEG, 0.0, 0.0)


def wrec_k(m):
    return math.ceil(m * INV_LN2)


def combine4(a, b, c, d):
    """wrec_combine4: children rescaled to the node's exponent by exact powers of two, added left to right."""
    M = max(a[0], b[0], c[0], d[0])
    if not M > NEG:
        return (M, 0.0, 0.0)
    K = wrec_k(M)

    def sc(x):
        if not x[0] > NEG:
            return 0.0, 0.0
        dk = int(max(wrec_k(x[0]) - K, -4000.0))
        return math.ldexp(x[1], dk), math.ldexp(x[2], 2 * dk)
    (a1, a2), (b1, b2), (c1, c2), (d1, d2) = sc(a), sc(b), sc(c), sc(d)
    return (M, ((a1 + b1) + c1) + d1, ((a2 + b2) + c2) + d2)


def leaf(logw64):
    """wrec_wave: a wave's 64 particles (fewer: the rest are idle lanes).  Sums by the xor butterfly, as the lanes add them."""
    lw = np.full(64, NEG)
    lw[:len(logw64)] = np.asarray(logw64, dtype=np.float64)
    m = float(lw.max())
    if not m > NEG:
        return (m, 0.0, 0.0)
    e = np.where(lw > NEG, np.exp(lw - wrec_k(m) * LN2), 0.0)
    s1, s2 = e.copy(), e * e
    for off in (32, 16, 8, 4, 2, 1):
        idx = np.arange(64) ^ off
        s1 = s1 + s1[idx]
        s2 = s2 + s2[idx]
    return (m, float(s1[0]), float(s2[0]))


def climb(nodes, levels):
    """`levels` radix-4 levels over a list of consecutive nodes (absent children are the identity)."""
    for _ in range(levels):
        nodes = [combine4(*(nodes[4 * i + j] if 4 * i + j < len(nodes) else EMPTY for j in range(4))) for i in range((len(nodes) + 3) // 4)]
    return nodes


def records_of_slice(logw):
    """A rank's 1024-particle records: leaves -> 256-particle nodes -> 1024-particle records (what its step kernel's last workgroup
    forms from the lines)."""
    leaves = [leaf(logw[i:i + 64]) for i in range(0, len(logw), 64)]
    return climb(leaves, 2)


def root_from_records(records):
    """pf_auto_tail's reduction of all ranks' records: passes of 1024 records (each climbs five levels to one pass node), then the
    radix-4 tree over the pass nodes."""
    passes = [climb(records[i:i + 1024], 5)[0] for i in range(0, len(records), 1024)]
    while len(passes) > 1:
        passes = climb(passes, 1)
    return passes[0] if passes else EMPTY


def root_one_rank(logw):
    """The same root formed the way ONE rank forms it, organised differently on purpose: level by level over all leaves."""
    nodes = [leaf(logw[i:i + 64]) for i in range(0, len(logw), 64)]
    while len(nodes) > 1:
        nodes = climb(nodes, 1)
    return nodes[0]


def statistics(root, n_global, neff_frac, force):
    """shift, Neff and the decision, as the tail forms them from the root."""
    m, s1, s2 = root
    shift = wrec_k(m) * LN2 + math.log(s1)
    neff = s1 * s1 / s2
    want = bool(force) if force is not None else bool(neff < neff_frac * n_global)
    return shift, neff, want


class Inbox:
    """One rank's inbox as the peers see it: the hand-shake words and the ranks' records of a step, two parities."""

    def __init__(self, world, n_global):
        self.world = world
        self.rec_cap = (n_global + 1023) // 1024 + 8
        self.rec = np.zeros((2, self.rec_cap, 4), dtype=np.uint64)          # {m, s1, s2, tag} as raw words
        self.ready = np.zeros(world, dtype=np.int64)                        # [r]: last resampling step whose step kernel rank r has completed
        self.gone = np.zeros(world, dtype=np.int64)                         # [r] != 0: rank r is going away

    def write_record(self, seq, index, rec, tag=None):
        row = self.rec[seq & 1, index]
        row[0], row[1], row[2] = _bits(rec[0]), _bits(rec[1]), _bits(rec[2])
        row[3] = record_tag(rec, seq, index) if tag is None else tag

    def read_record(self, seq, index):
        """The record if its tag fits the step and the index, else None (the reader polls again)."""
        row = self.rec[seq & 1, index]
        rec = tuple(float(np.uint64(v).view(np.float64)) for v in row[:3])
        return rec if (wrec_hash(rec) ^ int(row[3])) == (part_key(seq) ^ XKEY ^ ((index * INDEX_MUL) & MASK)) else None


class RankModel:
    """One rank of the sharded filter, as far as the exchange goes."""

    def __init__(self, rank, world, n_global, neff_frac=0.75):
        assert n_global % world == 0
        self.rank, self.world, self.n_global, self.neff_frac = rank, world, n_global, neff_frac
        self.n = n_global // world
        self.nc_local = (self.n + 1023) // 1024
        self.inbox = Inbox(world, n_global)
        self.error = 0

    # -- what this rank's step kernel puts into every inbox: (index, raw words) of its records of step `seq`
    def outgoing(self, seq, logw_slice):
        out = np.zeros((self.nc_local, 5), dtype=np.uint64)
        for j, rec in enumerate(records_of_slice(logw_slice)):
            gi = self.rank * self.nc_local + j
            out[j] = (gi, _bits(rec[0]), _bits(rec[1]), _bits(rec[2]), record_tag(rec, seq, gi))
        return out

    def deliver(self, seq, rows):
        """Posted stores of some rank land in this rank's inbox (in any order, possibly late)."""
        for gi, b0, b1, b2, tag in rows:
            self.inbox.rec[seq & 1, int(gi)] = (b0, b1, b2, tag)

    def peer_gone(self):
        return bool(self.inbox.gone.any())

    def collect(self, seq):
        """All ranks' records of step `seq`, or None while one of them is not in (stale / torn / missing): the tail polls."""
        if self.peer_gone():
            self.error = PF_ERR_PEER
            return None
        recs = [self.inbox.read_record(seq, i) for i in range(self.world * self.nc_local)]
        return None if any(r is None for r in recs) else recs

    def decide(self, seq, force=None):
        recs = self.collect(seq)
        if recs is None:
            return None
        root = root_from_records(recs)
        return (root,) + statistics(root, self.n_global, self.neff_frac, force)

    # -- the gate of a resampling step
    def gate_open(self, seq):
        if self.peer_gone():
            self.error = PF_ERR_PEER
            return False
        return bool((self.inbox.ready >= seq).all())
