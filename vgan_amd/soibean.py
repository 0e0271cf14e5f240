"""Python plumbing around the soibean C-ABI (include/vgan_gpu.h): analyse_GAM's factorised tables on the device and
the per-iteration likelihood refresh of MCMC::run_tree_proportion.  No arithmetic here."""
import ctypes as C

import numpy as np

from . import _native as N
from .haplocart import _np_view

_SB_FIELDS = (("read_seg_off", np.uint32, "R1"), ("read_col_off", np.uint32, "R1"), ("read_qual_off", np.uint32, "R1"),
              ("read_gseq_len", np.uint16, "R"), ("read_rseq_len", np.uint16, "R"), ("read_rev", np.uint8, "R"),
              ("read_src", np.uint32, "R"), ("seg_node", np.uint32, "S"), ("seg_col", np.uint16, "S"),
              ("seg_len", np.uint16, "S"), ("seg_base_ix", np.uint16, "S"), ("graph_seq", np.uint8, "C"),
              ("read_seq", np.uint8, "C"), ("qual", np.uint8, "Q"))


class SbHostBatch:
    def __init__(self, graph, alns, r0=0, r1=None, n_threads=0):
        r1 = alns.n_reads if r1 is None else r1
        self._h = N.vp()
        self.stats = N.SbFlattenStats()
        N.check(N.lib().vgan_sb_flatten(graph._h, alns._h, r0, r1, n_threads, C.byref(self._h), C.byref(self.stats)))
        self.c = N.SbBatch()
        N.check(N.lib().vgan_sb_host_batch_get(self._h, C.byref(self.c)))
        self.n_reads, self.n_segments = self.c.n_reads, self.c.n_segments

    def arrays(self):
        c = self.c
        n = {"R1": c.n_reads + 1, "R": c.n_reads, "S": c.n_segments, "C": c.n_cols, "Q": c.n_qual}
        out = {name: _np_view(getattr(c, name), n[k], dt) for name, dt, k in _SB_FIELDS}
        out["_owner"] = self
        return out

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_sb_host_batch_free(self._h)
            self._h = None


def download_batch(c):
    """A device vgan_sb_batch's arrays as numpy arrays (vgan_sb_batch_download)."""
    n = {"R1": c.n_reads + 1, "R": c.n_reads, "S": c.n_segments, "C": c.n_cols, "Q": c.n_qual}
    out = {name: np.zeros(max(int(n[k]), 1), dt) for name, dt, k in _SB_FIELDS}
    h = N.SbBatch()
    for name, _, _ in _SB_FIELDS:
        setattr(h, name, out[name].ctypes.data)
    N.check(N.lib().vgan_sb_batch_download(C.byref(c), C.byref(h)))
    return {name: out[name][:int(n[k])] if c.n_reads else out[name][:0] for name, _, k in _SB_FIELDS}


class SbDeviceBatch:
    """What SbContext.precompute takes: a vgan_sb_batch whose arrays live on the device."""

    def __init__(self, c):
        self.c = c
        self.n_reads, self.n_segments = c.n_reads, c.n_segments


class SbDeviceFlatten:
    """soibean's front half on the device (vgan_sb_devflat): GamDevice parses -> rows appended to a vgan_sb_batch in HBM + the masks of the
    reads left to the host."""

    def __init__(self, ctx, graph):
        self._h = N.vp()
        self.ctx = ctx
        N.check(N.lib().vgan_sb_devflat_create(ctx._h, graph._h, C.byref(self._h)))

    def append_gamdev(self, gd, base=0):
        n = gd.sizes["reads"]
        mask = np.zeros(max(n, 1), np.uint8)
        self.stats = N.SbFlattenStats()
        N.check(N.lib().vgan_sb_devflat_append_gamdev(self._h, gd._h, base, mask.ctypes.data, C.byref(self.stats)))
        return mask[:n]

    def append_host(self, host_batch, src_map=None):
        m = None if src_map is None else np.ascontiguousarray(src_map, np.uint32)
        N.check(N.lib().vgan_sb_devflat_append_host(self._h, C.byref(host_batch.c), None if m is None else m.ctypes.data))

    def batch(self):
        c = N.SbBatch()
        N.check(N.lib().vgan_sb_devflat_batch(self._h, C.byref(c)))
        return SbDeviceBatch(c)

    def download(self):
        return download_batch(self.batch().c)

    def close(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_sb_devflat_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def gam_run(ctxs, graph, data, piece_bytes=0, slots=0, n_threads=0, batches=False):
    """vgan_sb_gam_*: a BGZF GAM's bytes through the device front end's pipeline into the contexts (piece i -> context i mod n) and
    analyse_GAM's tables made once per context.  Returns ({n_messages, n_mapped, n_reads, n_bad, n_dev_bad[, batches: per lane, downloaded]},
    pipeline statistics)."""
    buf = np.frombuffer(data, np.uint8)
    o = N.GamPipeOpts(int(piece_bytes), int(slots), 0, 0, int(n_threads), 0)
    dev = (C.c_int * len(ctxs))(*[c.device for c in ctxs])
    arr = (N.vp * len(ctxs))(*[c._h for c in ctxs])
    run = N.vp()
    N.check(N.lib().vgan_sb_gam_start(dev, len(ctxs), buf.ctypes.data, len(data), C.byref(o), C.byref(run)))
    res, ps = N.SbGamResult(), N.GamPipeStats()
    try:
        N.check(N.lib().vgan_sb_gam_attach(run, arr, len(ctxs), graph._h))
        N.check(N.lib().vgan_sb_gam_finish(run, C.byref(res), C.byref(ps)))
        out = {k: int(getattr(res, k)) for k in ("n_messages", "n_mapped", "n_reads", "n_bad", "n_dev_bad")}
        out["lane_reads"] = []
        for l, c in enumerate(ctxs):
            b = N.SbBatch()
            N.check(N.lib().vgan_sb_gam_batch(run, l, C.byref(b)))
            c.n_reads = int(b.n_reads)
            out["lane_reads"].append(int(b.n_reads))
            if batches:
                out.setdefault("batches", []).append(download_batch(b))
    finally:
        N.lib().vgan_sb_gam_free(run)
    return out, ps.as_dict()


class SbContext:
    def __init__(self, graph, damage, penalty=7, device=0):
        self.graph, self.damage = graph, damage
        self.n_paths = graph.n_paths
        self._h = N.vp()
        p = N.SbParams(penalty, 0)
        N.check(N.lib().vgan_sb_create(C.byref(graph.view), C.byref(damage.view), C.byref(p), device, C.byref(self._h)))
        self.device = device
        self.n_reads = 0

    def use_torch_stream(self):
        N.check(N.lib().vgan_sb_set_stream(self._h, N.torch_stream_ptr(self.device)))

    def precompute(self, batch):
        bad = C.c_int64(0)
        N.check(N.lib().vgan_sb_precompute(self._h, C.byref(batch.c), C.byref(bad)))
        self.n_reads = batch.n_reads
        return bad.value

    def read_tables(self, r0=0, r1=None):
        r1 = self.n_reads if r1 is None else r1
        n = r1 - r0
        pm = np.zeros((self.n_paths, n))
        cnt = np.zeros((self.n_paths, 25, n), np.uint16)
        ok = np.zeros(n, np.uint8)
        N.check(N.lib().vgan_sb_read_tables(self._h, r0, r1, pm.ctypes.data, cnt.ctypes.data, ok.ctypes.data))
        return pm, cnt, ok

    def loglike(self, states, con, freqs7, device_out=None):
        """states: list of lists of (child, parent, dist, pos, theta); returns (logLike[n_states], guard[n_states])."""
        k = len(states[0])
        assert all(len(s) == k for s in states)
        arr = (N.SbSource * (len(states) * k))()
        for e, st in enumerate(states):
            for y, (c, p, d, pos, th) in enumerate(st):
                arr[e * k + y] = N.SbSource(c, p, d, pos, th)
        f = np.ascontiguousarray(freqs7, np.float64)
        out = np.zeros(len(states))
        guard = np.zeros(len(states), np.uint64)
        N.check(N.lib().vgan_sb_loglike(self._h, len(states), k, arr, con, f.ctypes.data, out.ctypes.data,
                                        device_out.data_ptr() if device_out is not None else None, guard.ctypes.data))
        return out, guard

    def loglike_sums(self, states, con, freqs7):
        """vgan_sb_loglike_sums: per state the fixed-point sum itself as (hi, lo, nf) -- what a caller holding the reads in several
        contexts adds up (integers: the order does not matter) before converting once with sum_value()."""
        k = len(states[0])
        arr = (N.SbSource * (len(states) * k))()
        for e, st in enumerate(states):
            for y, (c, p, d, pos, th) in enumerate(st):
                arr[e * k + y] = N.SbSource(c, p, d, pos, th)
        f = np.ascontiguousarray(freqs7, np.float64)
        sums = (N.SbSum * len(states))()
        guard = np.zeros(len(states), np.uint64)
        N.check(N.lib().vgan_sb_loglike_sums(self._h, len(states), k, arr, con, f.ctypes.data, sums, guard.ctypes.data))
        return [(s.hi, s.lo, s.nf) for s in sums], guard

    def mixture_sums(self, paths, log_freq):
        p = np.ascontiguousarray(paths, np.int32)
        s = N.SbSum()
        N.check(N.lib().vgan_sb_mixture_sums(self._h, len(p), p.ctypes.data, log_freq, C.byref(s)))
        return (s.hi, s.lo, s.nf)

    def refresh(self, state, con, freqs7):
        """One state through the chain driver's engine (fused kernel + fold into pinned host memory): (logLike, guard).
        state: list of (child, parent, dist, pos, theta)."""
        if getattr(self, "_engine", None) is None:
            self._engine = N.SbEngine()
            N.check(N.lib().vgan_sb_engine_gpu(self._h, C.byref(self._engine)))
            self._f7 = (C.c_double * 7)()
            self._out, self._gd = C.c_double(0), C.c_uint64(0)
        k = len(state)
        arr = (N.SbSource * k)(*[N.SbSource(*s) for s in state])
        self._f7[:] = list(freqs7)
        N.check(self._engine.refresh(self._engine.user, k, C.cast(arr, C.c_void_p), con, self._f7, C.byref(self._out), C.byref(self._gd)))
        return self._out.value, self._gd.value

    def time_engine(self, on=True):
        N.check(N.lib().vgan_sb_time_engine(self._h, int(on)))

    def resident(self, on=None):
        """The engine's refresh through the kernel that stays on the device (vgan_sb_resident): set it, or ask with on=None."""
        rc = N.lib().vgan_sb_resident(self._h, -1 if on is None else int(on))
        N.check(rc)
        return bool(rc) if on is None else None

    def resident_launches(self):
        n = C.c_uint64(0)
        N.check(N.lib().vgan_sb_resident_launches(self._h, C.addressof(n)))
        return int(n.value)

    def best_paths(self):
        """analyse_GAM's mostProbPath: (best[n_reads] with -1 for ties / excluded reads, sig_count[n_paths], n_reads_ok)."""
        best = np.zeros(max(self.n_reads, 1), np.int32)
        sig = np.zeros(self.n_paths, np.int64)
        n = C.c_int64(0)
        N.check(N.lib().vgan_sb_best_paths(self._h, best.ctypes.data, sig.ctypes.data, C.addressof(n)))
        return best[:self.n_reads], sig, n.value

    def mixture_loglike(self, paths, log_freq):
        """soibean.cpp:737-756: sum over reads of the oplusInitnatl-fold of log_freq + pathMap[path]."""
        p = np.ascontiguousarray(paths, np.int32)
        out = C.c_double(0)
        N.check(N.lib().vgan_sb_mixture_loglike(self._h, len(p), p.ctypes.data, log_freq, C.addressof(out)))
        return out.value

    def kernel_ms(self):
        ms = np.zeros(2)
        n = np.zeros(2, np.uint64)
        N.check(N.lib().vgan_sb_kernel_ms(self._h, ms.ctypes.data, n.ctypes.data))
        return {"precompute": (float(ms[0]), int(n[0])), "refresh": (float(ms[1]), int(n[1]))}

    def close(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_sb_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def sum_value(parts):
    """The log-likelihood of fixed-point sums added over contexts: parts = [(hi, lo, nf), ...] (vgan_sb_sum_add / _value)."""
    acc = N.SbSum(0, 0, 0.0)
    for hi, lo, nf in parts:
        x = N.SbSum(hi, lo, nf)
        N.lib().vgan_sb_sum_add(C.byref(acc), C.byref(x))
    return N.lib().vgan_sb_sum_value(C.byref(acc))


class SbGroup:
    """Several contexts holding shares of one job's reads, as one likelihood engine (vgan_sb_group): MCMC.cpp:739's reduction
    over the reads, run over devices.  The contexts stay the caller's."""

    def __init__(self, ctxs):
        self.ctxs = list(ctxs)
        self._h = N.vp()
        arr = (N.vp * len(self.ctxs))(*[c._h for c in self.ctxs])
        N.check(N.lib().vgan_sb_group_create(arr, len(self.ctxs), C.byref(self._h)))
        self._engine = N.SbEngine()
        N.check(N.lib().vgan_sb_engine_group(self._h, C.byref(self._engine)))
        self.n_paths = self.ctxs[0].n_paths

    def engine(self):
        return self._engine

    def refresh(self, state, con, freqs7):
        k = len(state)
        arr = (N.SbSource * k)(*[N.SbSource(*s) for s in state])
        f7 = (C.c_double * 7)(*list(freqs7))
        out, gd = C.c_double(0), C.c_uint64(0)
        N.check(self._engine.refresh(self._engine.user, k, C.cast(arr, C.c_void_p), con, f7, C.byref(out), C.byref(gd)))
        return out.value, gd.value

    def mixture_loglike(self, paths, log_freq):
        p = np.ascontiguousarray(paths, np.int32)
        out = C.c_double(0)
        N.check(self._engine.mixture(self._engine.user, len(p), p.ctypes.data_as(C.POINTER(C.c_int32)), log_freq, C.byref(out)))
        return out.value

    def best_paths(self):
        sig = np.zeros(self.n_paths, np.int64)
        n = C.c_int64(0)
        N.check(N.lib().vgan_sb_group_best_paths(self._h, sig.ctypes.data, C.addressof(n)))
        return sig, n.value

    def close(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_sb_group_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def signature_paths(sig_count, n_reads, cutk=0):
    """Initial sources from the signature counts (soibean.cpp:669-712)."""
    sig = np.ascontiguousarray(sig_count, np.int64)
    paths = np.zeros(len(sig), np.int32)
    n = C.c_int32(0)
    N.check(N.lib().vgan_sb_signature_paths(sig.ctypes.data, len(sig), n_reads, cutk, paths.ctypes.data, C.addressof(n)))
    return paths[:n.value].copy()


class Tree:
    """The taxon tree (<dbprefix>.new.dnd, soibean.cpp:565-596); nodes numbered in pre-order of the Newick text."""

    def __init__(self, handle):
        self._h = handle
        self.view = N.TreeView()
        N.check(N.lib().vgan_tree_view_get(self._h, C.byref(self.view)))

    @classmethod
    def parse(cls, newick):
        h = N.vp()
        N.check(N.lib().vgan_tree_parse(newick.encode() if isinstance(newick, str) else newick, C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path):
        h = N.vp()
        N.check(N.lib().vgan_tree_load(path.encode(), C.byref(h)))
        return cls(h)

    @property
    def n_nodes(self):
        return self.view.n_nodes

    @property
    def n_leaves(self):
        return self.view.n_leaves

    @property
    def parent(self):
        return _np_view(self.view.parent, self.n_nodes, np.int32)

    @property
    def dist(self):
        return _np_view(self.view.dist, self.n_nodes, np.float64)

    @property
    def names(self):
        return (self.view.names or b"").decode().split("\n")[:self.n_nodes]

    def children(self, v):
        off = _np_view(self.view.child_off, self.n_nodes + 1, np.int32)
        ch = _np_view(self.view.children, max(int(off[-1]), 1), np.int32)
        return [int(x) for x in ch[off[v]:off[v + 1]]]

    def node_paths(self, path_names):
        """Graph path index of every tree node, by name (-1 when the graph has no such path)."""
        idx = {n: i for i, n in enumerate(path_names)}
        return np.array([idx.get(n, -1) for n in self.names], np.int32)

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_tree_free(self._h)
            self._h = None


def python_engine(refresh, mixture, batched=False):
    """A vgan_sb_engine from two Python callables (tests drive the chain logic with the oracle's likelihood this way):
    refresh(list of (child, parent, dist, pos, theta), con, freqs7) -> (loglike, guard); mixture(paths, log_freq) -> loglike.
    batched=True also provides refresh_many (all chains' states in one call), served by the same callable."""
    def _refresh(user, k, src, con, freqs7, out, guard):
        arr = C.cast(src, C.POINTER(N.SbSource))
        st = [(arr[i].child, arr[i].parent, arr[i].dist, arr[i].pos, arr[i].theta) for i in range(k)]
        try:
            ll, g = refresh(st, con, [freqs7[i] for i in range(7)])
        except Exception:  # noqa: BLE001 -- reported through the return code
            return -1
        out[0] = ll
        guard[0] = g
        return 0

    def _mixture(user, n, paths, log_freq, out):
        try:
            out[0] = mixture([paths[i] for i in range(n)], log_freq)
        except Exception:  # noqa: BLE001
            return -1
        return 0

    def _many(user, n_states, k, src, con, freqs7, out, guard):
        arr = C.cast(src, C.POINTER(N.SbSource))
        f7 = [freqs7[i] for i in range(7)]
        try:
            for e_ in range(n_states):
                st = [(arr[e_ * k + i].child, arr[e_ * k + i].parent, arr[e_ * k + i].dist, arr[e_ * k + i].pos, arr[e_ * k + i].theta)
                      for i in range(k)]
                out[e_], guard[e_] = refresh(st, con, f7)
        except Exception:  # noqa: BLE001
            return -1
        return 0

    fns = (N.SB_REFRESH_FN(_refresh), N.SB_MIXTURE_FN(_mixture), N.SB_REFRESH_MANY_FN(_many) if batched else N.SB_REFRESH_MANY_FN())
    e = N.SbEngine(None, *fns)
    e._keep = fns
    return e


def estimate(engine, tree, node_path, sig_nodes, prefix, n_paths, freqs7, con=0.01, iters=500000, burnin=75000, chains=4, seed=1,
             run_mcmc=True, quiet=True):
    """soibean.cpp:738-944 (vgan_sb_estimate).  engine: an SbContext (GPU), an SbGroup (several) or a python_engine()."""
    if isinstance(engine, SbContext):
        e = N.SbEngine()
        N.check(N.lib().vgan_sb_engine_gpu(engine._h, C.byref(e)))
    elif isinstance(engine, SbGroup):
        e = engine.engine()
    else:
        e = engine
    cfg = N.SbEstimateCfg(iters, burnin, chains, n_paths, seed, con, (C.c_double * 7)(*freqs7), int(run_mcmc), int(quiet))
    np_ = np.ascontiguousarray(node_path, np.int32)
    sg = np.ascontiguousarray(sig_nodes, np.int32)
    N.check(N.lib().vgan_sb_estimate(C.byref(e), tree._h, np_.ctypes.data, sg.ctypes.data, len(sg), C.byref(cfg), prefix.encode()))
