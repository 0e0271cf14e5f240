"""Python plumbing around the euka C-ABI (include/vgan_gpu.h): clade/bin tables, damage profiles, batches, the
device context.  Mirrors the call `readGAM3(...)` of the reference (src/Euka.cpp:534-537); no arithmetic here."""
import ctypes as C

import numpy as np

from . import _native as N
from .haplocart import AlnSet, Graph, _np_view


class EukaDb:
    def __init__(self, handle):
        self._h = handle
        self.view = N.EukaDbView()
        N.check(N.lib().vgan_euka_db_view_get(self._h, C.byref(self.view)))

    @classmethod
    def load(cls, clade_path, bins_path):
        h = N.vp()
        N.check(N.lib().vgan_euka_db_load(clade_path.encode(), bins_path.encode(), C.byref(h)))
        return cls(h)

    @property
    def n_clades(self):
        return self.view.n_clades

    @property
    def clade_dist(self):
        return _np_view(self.view.clade_dist, self.n_clades, np.float64)

    @property
    def clade_id(self):
        return _np_view(self.view.clade_id, self.n_clades, np.int32)

    @property
    def clade_names(self):
        return (self.view.clade_names or b"").decode().split()

    @property
    def bin_off(self):
        return _np_view(self.view.bin_off, self.n_clades + 1, np.uint32)

    @property
    def n_bins(self):
        return int(self.bin_off[-1])

    @property
    def bin_lo(self):
        return _np_view(self.view.bin_lo, self.n_bins, np.int32)

    @property
    def bin_hi(self):
        return _np_view(self.view.bin_hi, self.n_bins, np.int32)

    @property
    def bin_entropy(self):
        return _np_view(self.view.bin_entropy, self.n_bins, np.float64)

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_euka_db_free(self._h)
            self._h = None


class Damage:
    def __init__(self, handle):
        self._h = handle
        self.view = N.DamageView()
        N.check(N.lib().vgan_damage_view_get(self._h, C.byref(self.view)))

    @classmethod
    def from_text(cls, prof5="", prof3=""):
        h = N.vp()
        N.check(N.lib().vgan_damage_from_text(prof5.encode(), prof3.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path5=None, path3=None):
        h = N.vp()
        N.check(N.lib().vgan_damage_load(path5.encode() if path5 else None, path3.encode() if path3 else None, C.byref(h)))
        return cls(h)

    @property
    def sub5p(self):
        return _np_view(self.view.sub5p, self.view.n5 * 16, np.float64).reshape(-1, 4, 4)

    @property
    def sub3p(self):
        return _np_view(self.view.sub3p, self.view.n3 * 16, np.float64).reshape(-1, 4, 4)

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_damage_free(self._h)
            self._h = None


_EB_FIELDS = (("read_col_off", np.uint32, "R1"), ("read_qual_off", np.uint32, "R1"), ("read_map_off", np.uint32, "R1"),
              ("read_gseq_len", np.uint16, "R"), ("read_rseq_len", np.uint16, "R"), ("read_seq_len", np.uint16, "R"),
              ("read_mapq", np.int32, "R"), ("read_rev", np.uint8, "R"), ("read_src", np.uint32, "R"),
              ("map_node", np.uint32, "M"), ("graph_seq", np.uint8, "C"), ("read_seq", np.uint8, "C"), ("qual", np.uint8, "Q"))


class EukaHostBatch:
    def __init__(self, graph, alns, r0=0, r1=None, n_threads=0):
        r1 = alns.n_reads if r1 is None else r1
        self._h = N.vp()
        self.stats = N.EukaFlattenStats()
        N.check(N.lib().vgan_euka_flatten(graph._h, alns._h, r0, r1, n_threads, C.byref(self._h), C.byref(self.stats)))
        self.c = N.EukaBatch()
        N.check(N.lib().vgan_euka_host_batch_get(self._h, C.byref(self.c)))
        self.n_reads = self.c.n_reads

    def arrays(self):
        c = self.c
        n = {"R1": c.n_reads + 1, "R": c.n_reads, "M": c.n_maps, "C": c.n_cols, "Q": c.n_qual}
        out = {name: _np_view(getattr(c, name), n[k], dt) for name, dt, k in _EB_FIELDS}
        out["_owner"] = self
        return out

    def algorithmic_bytes(self):
        c = self.c
        return 2 * c.n_cols + c.n_qual + 4 * c.n_maps + 27 * c.n_reads + 37 * c.n_reads  # inputs + per-read outputs

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_euka_host_batch_free(self._h)
            self._h = None


class EukaDeviceBatch:
    """The batch and its per-read output arrays resident in HBM (torch tensors)."""

    def __init__(self, hb, device="cuda:0"):
        import torch
        self.t = {}
        arrs = hb.arrays()
        for name, dt, _ in _EB_FIELDS:
            a = arrs[name]
            if dt == np.uint32:
                a = a.view(np.int32)
            elif dt == np.uint16:
                a = a.view(np.int16)
            self.t[name] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        c = N.EukaBatch()
        c.n_reads, c.n_cols, c.n_qual, c.n_maps = hb.c.n_reads, hb.c.n_cols, hb.c.n_qual, hb.c.n_maps
        for name, _, _ in _EB_FIELDS:
            setattr(c, name, self.t[name].data_ptr() if self.t[name].numel() else None)
        c.on_device = 1
        self.c = c
        self.n_reads = c.n_reads
        R = c.n_reads
        self.out = {"clade": torch.zeros(R, dtype=torch.int32, device=device),
                    "in_lik": torch.zeros(R, dtype=torch.float64, device=device),
                    "out_lik": torch.zeros(R, dtype=torch.float64, device=device),
                    "like": torch.zeros(R, dtype=torch.float64, device=device),
                    "not_like": torch.zeros(R, dtype=torch.float64, device=device),
                    "pass": torch.zeros(R, dtype=torch.uint8, device=device)}
        self.out_c = N.EukaReadOut(*[self.out[k].data_ptr() for k in ("clade", "in_lik", "out_lik", "like", "not_like", "pass")])


class EukaContext:
    def __init__(self, db, damage, min_mapq=29, length_to_prof=5, device=0):
        self.db, self.damage = db, damage
        self.ltp = length_to_prof
        self._h = N.vp()
        p = N.EukaParams(min_mapq, length_to_prof)
        N.check(N.lib().vgan_euka_create(C.byref(db.view), C.byref(damage.view), C.byref(p), device, C.byref(self._h)))
        self.device = device

    def use_torch_stream(self):
        N.check(N.lib().vgan_euka_set_stream(self._h, N.torch_stream_ptr(self.device)))

    def reset(self):
        N.check(N.lib().vgan_euka_reset(self._h))

    def accumulate(self, batch):
        """Host batch -> dict of numpy per-read results; device batch -> results in batch.out (torch tensors)."""
        if isinstance(batch, EukaDeviceBatch):
            N.check(N.lib().vgan_euka_accumulate(self._h, C.byref(batch.c), C.byref(batch.out_c)))
            return batch.out
        R = batch.n_reads
        out = {"clade": np.zeros(R, np.int32), "in_lik": np.zeros(R), "out_lik": np.zeros(R), "like": np.zeros(R),
               "not_like": np.zeros(R), "pass": np.zeros(R, np.uint8)}
        oc = N.EukaReadOut(*[out[k].ctypes.data for k in ("clade", "in_lik", "out_lik", "like", "not_like", "pass")])
        N.check(N.lib().vgan_euka_accumulate(self._h, C.byref(batch.c), C.byref(oc)))
        return out

    def finalize(self):
        C_, nb = self.db.n_clades, self.db.n_bins
        count = np.zeros(C_, np.int32)
        shift = np.zeros((C_, 2 * self.ltp, 16), np.uint32)
        cov = np.zeros(nb)
        bad = C.c_int64(0)
        N.check(N.lib().vgan_euka_finalize(self._h, count.ctypes.data, shift.ctypes.data, cov.ctypes.data, C.byref(bad)))
        return {"clade_count": count, "baseshift": shift, "bin_cov": cov, "n_bad": bad.value}

    def like_sums(self):
        """Per clade (number of clade_like entries, sum of their logs); call after finalize()."""
        n = np.zeros(self.db.n_clades, np.int64)
        s = np.zeros(self.db.n_clades)
        N.check(N.lib().vgan_euka_like_sums(self._h, n.ctypes.data, s.ctypes.data))
        return n, s

    def kernel_ms(self):
        ms = C.c_double(0)
        n = C.c_uint64(0)
        N.check(N.lib().vgan_euka_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_euka_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class EukaDeviceFlatten:
    """euka's front half on the device (vgan_euka_devflat): a GamDevice parse -> a vgan_euka_batch in HBM + the mask of the reads left to
    the host."""

    def __init__(self, ctx, graph):
        self._h = N.vp()
        self.ctx = ctx
        N.check(N.lib().vgan_euka_devflat_create(ctx._h, graph._h, C.byref(self._h)))

    def run_gamdev(self, gd, base=0):
        n = gd.sizes["reads"]
        self.mask = np.zeros(max(n, 1), np.uint8)
        self.c = N.EukaBatch()
        self.stats = N.EukaFlattenStats()
        N.check(N.lib().vgan_euka_devflat_run_gamdev(self._h, gd._h, base, C.byref(self.c), self.mask.ctypes.data, C.byref(self.stats)))
        self.mask = self.mask[:n]
        return self

    def download(self):
        c = self.c
        n = {"R1": c.n_reads + 1, "R": c.n_reads, "M": c.n_maps, "C": c.n_cols, "Q": c.n_qual}
        out = {name: np.zeros(max(int(n[k]), 1), dt) for name, dt, k in _EB_FIELDS}
        h = N.EukaBatch()
        for name, _, _ in _EB_FIELDS:
            setattr(h, name, out[name].ctypes.data)
        N.check(N.lib().vgan_euka_batch_download(C.byref(c), C.byref(h)))
        return {name: out[name][:int(n[k])] for name, _, k in _EB_FIELDS}

    def accumulate(self):
        """The batch through the context's read kernel: per-read clade / pass as numpy arrays (in the batch's order)."""
        import torch
        R = self.c.n_reads
        dev = "cuda:%d" % self.ctx.device
        o = {"clade": torch.zeros(R, dtype=torch.int32, device=dev), "pass": torch.zeros(R, dtype=torch.uint8, device=dev)}
        d = torch.zeros(4 * R, dtype=torch.float64, device=dev)
        oc = N.EukaReadOut(o["clade"].data_ptr(), d.data_ptr(), d.data_ptr() + 8 * R, d.data_ptr() + 16 * R, d.data_ptr() + 24 * R, o["pass"].data_ptr())
        N.check(N.lib().vgan_euka_accumulate(self.ctx._h, C.byref(self.c), C.byref(oc)))
        N.check(N.lib().vgan_euka_synchronize(self.ctx._h))
        return {"clade": o["clade"].cpu().numpy(), "pass": o["pass"].cpu().numpy(), "in_lik": d[:R].cpu().numpy(), "out_lik": d[R:2 * R].cpu().numpy()}

    def close(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_euka_devflat_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def gam_run(ctxs, graph, data, piece_bytes=0, slots=0, n_threads=0):
    """vgan_euka_gam_*: a BGZF GAM's bytes through the device front end's pipeline into the contexts (piece i -> context i mod n).
    Returns ({n_messages, n_mapped, n_bad, read_index, read_clade, read_pass, read_seq_len}, pipeline statistics)."""
    buf = np.frombuffer(data, np.uint8)
    o = N.GamPipeOpts(int(piece_bytes), int(slots), 0, 0, int(n_threads), 0)
    dev = (C.c_int * len(ctxs))(*[c.device for c in ctxs])
    arr = (N.vp * len(ctxs))(*[c._h for c in ctxs])
    run = N.vp()
    N.check(N.lib().vgan_euka_gam_start(dev, len(ctxs), buf.ctypes.data, len(data), C.byref(o), C.byref(run)))
    res, ps = N.EukaGamResult(), N.GamPipeStats()
    try:
        N.check(N.lib().vgan_euka_gam_attach(run, arr, len(ctxs), graph._h))
        N.check(N.lib().vgan_euka_gam_finish(run, C.byref(res), C.byref(ps)))
        n = int(res.n_reads)
        out = {"n_messages": int(res.n_messages), "n_mapped": int(res.n_mapped), "n_bad": int(res.n_bad),
               "read_index": np.array(_np_view(res.read_index, n, np.uint32)), "read_clade": np.array(_np_view(res.read_clade, n, np.int32)),
               "read_pass": np.array(_np_view(res.read_pass, n, np.uint8)), "read_seq_len": np.array(_np_view(res.read_seq_len, n, np.uint16))}
    finally:
        N.lib().vgan_euka_gam_free(run)
    return out, ps.as_dict()


def detect(db, clade_count, bin_cov, min_bins=6, min_reads=10, max_zero_bins=0, entropy=1.17):
    """Detected clade ids (readGAM_Euka.h:582-630)."""
    p = N.EukaDetectParams(min_bins, min_reads, max_zero_bins, entropy)
    cc = np.ascontiguousarray(clade_count, np.int32)
    cov = np.ascontiguousarray(bin_cov, np.float64)
    ids = np.zeros(db.n_clades, np.int32)
    n = C.c_int32(0)
    N.check(N.lib().vgan_euka_detect(C.byref(db.view), cc.ctypes.data, cov.ctypes.data, C.byref(p), ids.ctypes.data, C.addressof(n)))
    return ids[:n.value].copy()


def abundance_mcmc(init, n_like, sum_log_like, iters=10000, burnin=100, seed=1):
    """MCMC::run (MCMC.cpp:1217-1366) on the per-clade sums; returns [n, 5] = median, 15 %, 85 %, 5 %, 95 %."""
    init = np.ascontiguousarray(init, np.float64)
    nl = np.ascontiguousarray(n_like, np.int64)
    sl = np.ascontiguousarray(sum_log_like, np.float64)
    est = np.zeros((len(init), 5))
    N.check(N.lib().vgan_euka_abundance_mcmc(len(init), init.ctypes.data, nl.ctypes.data, sl.ctypes.data, iters, burnin, seed,
                                             est.ctypes.data))
    return est


def report(db, fin, n_like, sum_log_like, read_clade, read_pass, read_seq_len, prefix, names=None, min_bins=6, min_reads=10,
           max_zero_bins=0, entropy=1.17, length_to_prof=5, run_mcmc=True, iters=10000, burnin=100, seed=1, out_frag=False,
           out_group=None, out_dir=None):
    """Euka::run after readGAM3 (Euka.cpp:540-1160): writes <prefix>_* and returns (detected ids, estimates[n, 5]).
    fin = EukaContext.finalize() result; names = list of bytes per read (needed for out_frag)."""
    keep = [np.ascontiguousarray(fin["clade_count"], np.int32), np.ascontiguousarray(fin["baseshift"], np.uint32),
            np.ascontiguousarray(fin["bin_cov"], np.float64), np.ascontiguousarray(n_like, np.int64),
            np.ascontiguousarray(sum_log_like, np.float64), np.ascontiguousarray(read_clade, np.int32),
            np.ascontiguousarray(read_pass, np.uint8), np.ascontiguousarray(read_seq_len, np.uint16)]
    R = len(keep[5])
    off = blob = None
    if names is not None:
        off = np.zeros(R + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in names])
        blob = C.create_string_buffer(b"".join(names))
    res = N.EukaResults(C.addressof(db.view), *[k.ctypes.data for k in keep[:5]], R, *[k.ctypes.data for k in keep[5:]],
                        off.ctypes.data if off is not None else None, C.addressof(blob) if blob is not None else None)
    cfg = N.EukaReportCfg(N.EukaDetectParams(min_bins, min_reads, max_zero_bins, entropy), length_to_prof, int(run_mcmc), iters, burnin,
                          seed, int(out_frag), 0, out_group.encode() if out_group else None, out_dir.encode() if out_dir else None)
    det = np.zeros(db.n_clades + 1, np.int32)
    est = np.zeros((db.n_clades + 1, 5))
    n = C.c_int32(0)
    N.check(N.lib().vgan_euka_report(C.byref(res), C.byref(cfg), prefix.encode(), det.ctypes.data, C.addressof(n), est.ctypes.data))
    return det[:n.value].copy(), est[:n.value].copy()


def synth_euka(n_reads, damage=None, seed=0x76676131, n_clades=335, nodes_per_clade=400, read_len_mean=75, read_seed=0):
    cfg = N.SynthEukaCfg(seed, n_clades, nodes_per_clade, n_reads, read_len_mean, 0, read_seed)
    g, d, a = N.vp(), N.vp(), N.vp()
    N.check(N.lib().vgan_synth_euka(C.byref(cfg), damage._h if damage is not None else None, C.byref(g), C.byref(d), C.byref(a)))
    return Graph(g), EukaDb(d), AlnSet(a)
