"""Python plumbing around the HaploCart C-ABI (include/vgan_gpu.h): owners for the native objects, numpy
views of their arrays, torch tensors for device-resident batches.  No arithmetic of the hot path lives here.

Reference call sites mirrored: Haplocart::update loop (src/HaploCart.cpp:408-421) -> HcContext.accumulate /
finalize; Haplocart::get_posterior (src/get_posterior.cpp:87-127) -> HcContext.posterior.
"""
import ctypes as C

import numpy as np

from . import _native as N

MODE_NODE_WEIGHTS, MODE_PER_READ, MODE_PER_READ_DENSE = 0, 1, 2


def _np_view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class Graph:
    """Owner of a native vgan_graph (GFA + hcfiles sidecars, or synthetic)."""

    def __init__(self, handle):
        self._h = handle
        self.view = N.GraphView()
        N.check(N.lib().vgan_graph_view_get(self._h, C.byref(self.view)))

    @classmethod
    def load(cls, gfa_path, hcfiles_dir=None):
        h = N.vp()
        N.check(N.lib().vgan_graph_load(gfa_path.encode(), hcfiles_dir.encode() if hcfiles_dir else None, C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, min_id, max_id, node_seq_off, node_seq, n_paths, mask, pangenome_base, mappability,
                    path_names="", parents_txt="", children_txt=""):
        keep = dict(
            node_seq_off=np.ascontiguousarray(node_seq_off, np.int64),
            node_seq=np.ascontiguousarray(np.frombuffer(bytes(node_seq) + b"\0", np.uint8)),
            mask=np.ascontiguousarray(mask, np.uint64),
            pangenome_base=np.ascontiguousarray(pangenome_base, np.int32),
            mappability=np.ascontiguousarray(mappability, np.float64))
        v = N.GraphView(min_id, max_id, keep["node_seq_off"].ctypes.data, keep["node_seq"].ctypes.data, n_paths,
                        (n_paths + 63) // 64, keep["mask"].ctypes.data, keep["pangenome_base"].ctypes.data,
                        keep["mappability"].ctypes.data, len(keep["mappability"]), path_names.encode(),
                        parents_txt.encode(), children_txt.encode())
        h = N.vp()
        N.check(N.lib().vgan_graph_from_arrays(C.byref(v), C.byref(h)))
        return cls(h)

    def write(self, directory):
        N.check(N.lib().vgan_graph_write(self._h, directory.encode()))

    # numpy views (valid while self is alive)
    @property
    def n_paths(self):
        return self.view.n_paths

    @property
    def max_id(self):
        return self.view.max_id

    @property
    def min_id(self):
        return self.view.min_id

    @property
    def mask(self):
        w = self.view.mask_words
        return _np_view(self.view.mask, (self.view.max_id + 1) * w, np.uint64).reshape(self.view.max_id + 1, w)

    @property
    def node_seq_off(self):
        return _np_view(self.view.node_seq_off, self.view.max_id + 2, np.int64)

    @property
    def node_seq(self):
        return _np_view(self.view.node_seq, int(self.node_seq_off[-1]), np.uint8)

    @property
    def pangenome_base(self):
        return _np_view(self.view.pangenome_base, self.view.max_id + 1, np.int32)

    @property
    def mappability(self):
        return _np_view(self.view.mappability, self.view.n_mappability, np.float64)

    @property
    def path_names(self):
        return (self.view.path_names or b"").decode().split()

    @property
    def parents_txt(self):
        return (self.view.parents_txt or b"").decode()

    @property
    def children_txt(self):
        return (self.view.children_txt or b"").decode()

    def pathsgo(self):
        """bool-per-byte [max_id+1, P] matrix as the reference's NodeInfo::pathsgo (for the oracle)."""
        m = self.mask
        bits = np.unpackbits(m.view(np.uint8).reshape(m.shape[0], -1), axis=1, bitorder="little")
        return np.ascontiguousarray(bits[:, : self.n_paths])

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_graph_free(self._h)
            self._h = None


class AlnSet:
    """Owner of a native vgan_alnset (decoded GAM)."""

    FIELDS = (("seq_off", np.int64), ("qual_off", np.int64), ("name_off", np.int64), ("map_off", np.int64))

    def __init__(self, handle):
        self._h = handle
        self.view = N.AlnSetView()
        N.check(N.lib().vgan_aln_view_get(self._h, C.byref(self.view)))

    @classmethod
    def read_gam(cls, path, keep_unmapped=False):
        h = N.vp()
        N.check(N.lib().vgan_aln_read_gam(path.encode(), int(keep_unmapped), C.byref(h)))
        return cls(h)

    @classmethod
    def parse_gam(cls, data, keep_unmapped=False):
        h = N.vp()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        N.check(N.lib().vgan_aln_parse_gam(C.addressof(buf), len(data), int(keep_unmapped), C.byref(h)))
        return cls(h)

    def write_gam(self, path, group_size=512):
        N.check(N.lib().vgan_aln_write_gam(self._h, path.encode(), group_size))

    @property
    def n_reads(self):
        return self.view.n_reads

    def mark_duplicates(self):
        """bool mask with the reference's keep-first rule on the first mapping's (node, offset) (rmdup.cpp:68-110)."""
        m = np.zeros(self.n_reads, np.uint8)
        N.check(N.lib().vgan_aln_mark_duplicates(self._h, m.ctypes.data, None))
        return m.astype(bool)

    def without(self, drop):
        d = np.ascontiguousarray(drop, np.uint8)
        h = N.vp()
        N.check(N.lib().vgan_aln_filter(self._h, d.ctypes.data, C.byref(h)))
        return AlnSet(h)

    def arrays(self):
        """dict of numpy views with the field names of vgan_alnset_view (and of the oracle's orc_alnset)."""
        v = self.view
        R = v.n_reads
        out = {"n_reads": R, "_owner": self}
        out["seq_off"] = _np_view(v.seq_off, R + 1, np.int64)
        out["qual_off"] = _np_view(v.qual_off, R + 1, np.int64)
        out["name_off"] = _np_view(v.name_off, R + 1, np.int64)
        out["map_off"] = _np_view(v.map_off, R + 1, np.int64)
        out["seq"] = _np_view(v.seq, int(out["seq_off"][-1]), np.uint8)
        out["qual"] = _np_view(v.qual, int(out["qual_off"][-1]), np.uint8)
        out["name"] = _np_view(v.name, int(out["name_off"][-1]), np.uint8)
        out["mapq"] = _np_view(v.mapq, R, np.int32)
        out["identity"] = _np_view(v.identity, R, np.float64)
        M = int(out["map_off"][-1])
        out["m_node"] = _np_view(v.m_node, M, np.int64)
        out["m_offset"] = _np_view(v.m_offset, M, np.int64)
        out["m_rev"] = _np_view(v.m_rev, M, np.uint8)
        out["edit_off"] = _np_view(v.edit_off, M + 1, np.int64)
        E = int(out["edit_off"][-1])
        out["e_from"] = _np_view(v.e_from, E, np.int32)
        out["e_to"] = _np_view(v.e_to, E, np.int32)
        out["e_seq_off"] = _np_view(v.e_seq_off, E + 1, np.int64)
        out["e_seq"] = _np_view(v.e_seq, int(out["e_seq_off"][-1]), np.uint8)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_aln_free(self._h)
            self._h = None


class GamStream:
    """A GAM decoded behind the caller and handed out in chunks of at least `min_reads` reads (vgan_gam_stream)."""

    def __init__(self, path, keep_unmapped=False):
        self._h = N.vp()
        N.check(N.lib().vgan_gam_stream_open(path.encode(), int(keep_unmapped), C.byref(self._h)))

    def chunks(self, min_reads=500000):
        while True:
            h = N.vp()
            N.check(N.lib().vgan_gam_stream_next(self._h, min_reads, C.byref(h)))
            if not h:
                return
            yield AlnParts(h)

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_gam_stream_close(self._h)
            self._h = None


class Dedup:
    """Keep-first duplicate marks across the chunks of a stream (vgan_dedup)."""

    def __init__(self):
        self._h = N.vp()
        N.check(N.lib().vgan_dedup_create(C.byref(self._h)))

    def mark(self, parts):
        m = np.zeros(parts.n_reads, np.uint8)
        N.check(N.lib().vgan_dedup_mark(self._h, parts._h, m.ctypes.data, None))
        return m.astype(bool)

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_dedup_free(self._h)
            self._h = None


class AlnParts:
    """A GAM kept as the slices its parser produced (vgan_alnparts): what a front end that only feeds the device uses."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def read_gam(cls, path, keep_unmapped=False):
        h = N.vp()
        N.check(N.lib().vgan_alnparts_read_gam(path.encode(), int(keep_unmapped), C.byref(h)))
        return cls(h)

    @property
    def n_reads(self):
        return N.lib().vgan_alnparts_n_reads(self._h)

    @property
    def n_parts(self):
        return N.lib().vgan_alnparts_count(self._h)

    @property
    def base(self):
        """Index in the whole input of this object's first read (chunks of a GamStream)."""
        return N.lib().vgan_alnparts_base(self._h)

    def first_read(self, i):
        return N.lib().vgan_alnparts_first_read(self._h, i)

    def mark_duplicates(self):
        m = np.zeros(self.n_reads, np.uint8)
        N.check(N.lib().vgan_alnparts_mark_duplicates(self._h, m.ctypes.data, None))
        return m.astype(bool)

    def merge(self):
        """The merged alignment set (consumes the slices)."""
        h = N.vp()
        N.check(N.lib().vgan_alnparts_merge(self._h, C.byref(h)))
        return AlnSet(h)

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_alnparts_free(self._h)
            self._h = None


def reconstruct(graph, alns, r, cap=1 << 16):
    """a1 of the product front end (for the reconstruction KATs)."""
    gs = C.create_string_buffer(cap)
    rs = C.create_string_buffer(cap)
    sizes = np.zeros(cap, np.int32)
    lens = np.zeros(3, np.int64)
    N.check(N.lib().vgan_reconstruct(graph._h, alns._h, r, gs, rs, sizes.ctypes.data, cap, lens.ctypes.data))
    return gs.raw[: lens[0]], rs.raw[: lens[1]], sizes[: lens[2]].tolist()


_BATCH_FIELDS = (("read_seg_off", np.uint32, "R1"), ("read_col_off", np.uint32, "R1"), ("read_qual_off", np.uint32, "R1"),
                 ("read_algn_len", np.uint16, "R"), ("read_mapq", np.uint8, "R"), ("seg_node", np.uint32, "S"),
                 ("seg_start", np.uint16, "S"), ("seg_len", np.uint16, "S"), ("graph_seq", np.uint8, "C"),
                 ("algnseq", np.uint8, "C"), ("qual", np.uint8, "Q"))


_PACKED_FIELDS = (("rhdr", np.uint32, lambda k: 4 * (k.n_reads + 1)), ("srec", np.uint32, lambda k: k.n_segments),
                  ("crec", np.uint32, lambda k: k.n_cols), ("qualp", np.uint8, lambda k: k.n_qual + 32))


class HostBatch:
    """Flattened HaploCart batch in host memory (vgan_hc_flatten).  packed=True (vgan_hc_flatten*_packed): the reads that
    satisfy the tile contract leave in the segment kernel's own layout (`pk`, a vgan_hc_packed_view; packed_arrays()), and
    `c` / arrays() hold the OTHER reads alone; n_reads / n_segments count both parts."""

    def __init__(self, graph, alns, r0=0, r1=None, n_threads=0, skip=None, packed=False):
        """skip: optional bool/uint8 mask over the alignment set (e.g. AlnSet.mark_duplicates()): reads left out.
        alns may be an AlnParts: r0 / r1 then count slices."""
        self._h = N.vp()
        self.stats = N.FlattenStats()
        self.pk = None
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        assert sk is None or len(sk) == alns.n_reads
        skp = None if sk is None else sk.ctypes.data
        L = N.lib()
        if isinstance(alns, AlnParts):
            r1 = alns.n_parts if r1 is None else r1
            fn = L.vgan_hc_flatten_parts_packed if packed else L.vgan_hc_flatten_parts
        else:
            r1 = alns.n_reads if r1 is None else r1
            fn = L.vgan_hc_flatten_packed if packed else L.vgan_hc_flatten_masked
        N.check(fn(graph._h, alns._h, r0, r1, skp, n_threads, C.byref(self._h), C.byref(self.stats)))
        self.c = N.HcBatch()
        N.check(L.vgan_hc_host_batch_get(self._h, C.byref(self.c)))
        if packed:
            self.pk = N.HcPackedView()
            N.check(L.vgan_hc_host_batch_get_packed(self._h, C.byref(self.pk)))

    @property
    def n_reads(self):
        return self.c.n_reads + (self.pk.n_reads if self.pk is not None else 0)

    @property
    def n_segments(self):
        return self.c.n_segments + (self.pk.n_segments if self.pk is not None else 0)

    def packed_arrays(self):
        """numpy views of the packed part (rhdr [4(R+1)], srec [S], crec [C], qualp [Q+32], read_src [R])."""
        k = self.pk
        out = {name: _np_view(getattr(k, name), size(k), dt) for name, dt, size in _PACKED_FIELDS}
        out["read_src"] = _np_view(k.read_src, k.n_reads, np.uint32)
        out["_owner"] = self
        return out

    def arrays(self):
        c = self.c
        n = {"R1": c.n_reads + 1, "R": c.n_reads, "S": c.n_segments, "C": c.n_cols, "Q": c.n_qual}
        out = {name: _np_view(getattr(c, name), n[k], dt) for name, dt, k in _BATCH_FIELDS}
        out["read_src"] = _np_view(c.read_src, c.n_reads, np.uint32)
        out["_owner"] = self
        return out

    @property
    def n_tileable(self):
        """Reads [0, n_tileable) take the LDS-tiled kernel (vgan_hc_flatten puts them first)."""
        return self.c.n_tileable

    @property
    def read_src(self):
        """Index in the alignment set of each batch read (the batch is not in input order): the packed reads, then the others."""
        rest = np.array(self.arrays()["read_src"])
        return rest if self.pk is None else np.concatenate([np.array(self.packed_arrays()["read_src"]), rest])

    def algorithmic_bytes(self, n_paths):
        """bases + quals + segment descriptors (+ one mask row per segment in PER_READ mode).  The same
        figure for both layouts: the input bytes of the SoA form (two byte streams per column, the quality string, an
        8-byte record per segment, 15 bytes of per-read header) -- the packed layout is fatter, which lowers the fraction."""
        c, k = self.c, self.pk
        cols, qual = c.n_cols + (k.n_cols if k is not None else 0), c.n_qual + (k.n_qual if k is not None else 0)
        io = 2 * cols + qual + 8 * self.n_segments + 15 * self.n_reads
        return {"node_weights": io, "per_read": io + self.n_segments * (8 * ((n_paths + 63) // 64) + 8)}

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_hc_host_batch_free(self._h)
            self._h = None


class ArrayBatch:
    """A vgan_hc_batch over caller-built numpy arrays (host memory): what a C caller of the ABI hands over.
    `arrays` uses the field names of vgan_hc_batch; n_tileable as in include/vgan_gpu.h (0 is always valid)."""

    def __init__(self, arrays, n_tileable=0):
        self._keep = {name: np.ascontiguousarray(arrays[name], dt) for name, dt, _ in _BATCH_FIELDS}
        c = N.HcBatch()
        c.n_reads = len(self._keep["read_algn_len"])
        c.n_segments = len(self._keep["seg_node"])
        c.n_cols = len(self._keep["graph_seq"])
        c.n_qual = len(self._keep["qual"])
        for name, _, _ in _BATCH_FIELDS:
            setattr(c, name, self._keep[name].ctypes.data if self._keep[name].size else None)
        c.on_device = 0
        c.n_tileable = n_tileable
        c.read_src = None
        self.c = c
        self.n_reads, self.n_segments = c.n_reads, c.n_segments


class DeviceBatch:
    """The same SoA resident in HBM as torch tensors (zero-copy hand-over to vgan_hc_accumulate).  With `ctx`, the
    tileable reads are also put -- once -- into the packed layout the segment kernel streams (vgan_hc_pack), so that
    every accumulate of the resident batch is the kernel alone; without it the batch takes the LDS-tiled kernel."""

    def __init__(self, host_batch, device="cuda:0", ctx=None):
        import torch
        self.t = {}
        self.pk = None
        if getattr(host_batch, "pk", None) is not None:
            # a packed batch: its four arrays go to HBM as they are (one copy, no layout pass on the device)
            pa, hk = host_batch.packed_arrays(), host_batch.pk
            k = N.HcPackedView()
            for f in ("n_reads", "n_segments", "n_cols", "n_qual", "max_read_segs", "max_read_qual", "max_read_cols", "max_read_node_span"):
                setattr(k, f, getattr(hk, f))
            for name, dt, _ in _PACKED_FIELDS:
                a = pa[name].view(np.int32) if dt == np.uint32 else pa[name]
                self.t["pk_" + name] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
                setattr(k, name, self.t["pk_" + name].data_ptr())
            k.on_device = 1
            k.read_src = None
            self.pk = k
            ctx = None  # (nothing left for the layout pass: the other reads take the general kernel)
        arrs = host_batch.arrays()
        for name, dt, _ in _BATCH_FIELDS:
            a = arrs[name]
            # torch has no uint16/uint32 arithmetic but can hold the bytes: ship as int16/int32 views
            if dt == np.uint32:
                a = a.view(np.int32)
            elif dt == np.uint16:
                a = a.view(np.int16)
            self.t[name] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        c = N.HcBatch()
        c.n_reads, c.n_segments = host_batch.c.n_reads, host_batch.c.n_segments
        c.n_cols, c.n_qual = host_batch.c.n_cols, host_batch.c.n_qual
        for name, _, _ in _BATCH_FIELDS:
            setattr(c, name, self.t[name].data_ptr() if self.t[name].numel() else None)
        c.on_device = 1
        c.n_tileable = host_batch.c.n_tileable
        c.read_src = None  # host-side bookkeeping only
        c.packed = None
        self.c = c
        self.n_reads, self.n_segments = host_batch.n_reads, host_batch.n_segments
        self._packed = N.vp()
        self.pack_ms = None
        if ctx is not None:
            self.pack(ctx)

    def pack(self, ctx):
        """The layout pass (vgan_hc_pack) on ctx's device; returns its wall time in ms (synchronous)."""
        import time
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N.check(N.lib().vgan_hc_pack(ctx._h, C.byref(self.c), C.byref(self._packed)))
        self.pack_ms = (time.perf_counter() - t0) * 1e3
        self.c.packed = self._packed
        return self.pack_ms

    def download_packed(self):
        """What the layout pass (vgan_hc_pack) wrote, as host arrays named as in vgan_hc_packed_view (test aid)."""
        n = np.zeros(4, np.uint64)
        N.check(N.lib().vgan_hc_packed_download(self._packed, n.ctypes.data, None, None, None, None))
        R, S, Cn, Q = (int(x) for x in n)
        out = {"rhdr": np.zeros(4 * (R + 1), np.uint32), "srec": np.zeros(S, np.uint32), "crec": np.zeros(Cn, np.uint32),
               "qualp": np.zeros(Q + 32, np.uint8)}
        N.check(N.lib().vgan_hc_packed_download(self._packed, n.ctypes.data, out["rhdr"].ctypes.data, out["srec"].ctypes.data,
                                                out["crec"].ctypes.data, out["qualp"].ctypes.data))
        return out

    def __del__(self):
        if getattr(self, "_packed", None) and N is not None:
            N.lib().vgan_hc_packed_free(self._packed)
            self._packed = None


class DeviceFlatten:
    """a1 on the device (vgan_hc_devflat): a chunk of parsed alignments (AlnParts) -> the packed batch of the reads the
    device can take, resident in HBM, and the mask of the reads left to the host flatten."""

    class Result:
        """pk: the device view (valid until the next run); host_mask[r] = 1: read r of the chunk is the host's; c: no SoA part."""

        def __init__(self, pk, host_mask, stats):
            self.pk, self.host_mask, self.stats = pk, host_mask, stats
            self.c = N.HcBatch()  # (an empty SoA part: HcContext.accumulate takes the packed view alone)
            self.n_reads, self.n_segments = pk.n_reads, pk.n_segments

        def download(self):
            k = self.pk
            out = {"rhdr": np.zeros(4 * (k.n_reads + 1), np.uint32), "srec": np.zeros(k.n_segments, np.uint32),
                   "crec": np.zeros(k.n_cols, np.uint32), "qualp": np.zeros(k.n_qual + 32, np.uint8)}
            if k.n_reads:
                N.check(N.lib().vgan_hc_packed_view_download(C.byref(k), out["rhdr"].ctypes.data, out["srec"].ctypes.data,
                                                             out["crec"].ctypes.data, out["qualp"].ctypes.data))
                out["read_src"] = np.array(_np_view(k.read_src, k.n_reads, np.uint32))
            else:
                out["read_src"] = np.zeros(0, np.uint32)
            return out

    def __init__(self, ctx, graph):
        self._h = N.vp()
        self._keep = (ctx, graph)
        N.check(N.lib().vgan_hc_devflat_create(ctx._h, graph._h, C.byref(self._h)))

    def run(self, parts, skip=None):
        assert isinstance(parts, AlnParts)
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        mask = np.zeros(max(parts.n_reads, 1), np.uint8)
        pk, st = N.HcPackedView(), N.FlattenStats()
        N.check(N.lib().vgan_hc_devflat_run(self._h, parts._h, None if sk is None else sk.ctypes.data, C.byref(pk), mask.ctypes.data, C.byref(st)))
        return DeviceFlatten.Result(pk, mask[:parts.n_reads], st)

    def run_gamdev(self, gd, skip=None, base=0, device_marks=False):
        """The same over the arrays a GamDevice.parse left on the device: nothing but the marks and the mask crosses the link
        (device_marks: the marks GamDevice.mark_duplicates left on the device)."""
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        n = gd.sizes["reads"]
        mask = np.zeros(max(n, 1), np.uint8)
        pk, st = N.HcPackedView(), N.FlattenStats()
        skp, on_dev = (None if sk is None else sk.ctypes.data), 0
        if device_marks:
            skp, on_dev = N.lib().vgan_gamdev_dup_marks(gd._h), 1
        N.check(N.lib().vgan_hc_devflat_run_gamdev(self._h, gd._h, skp, on_dev, base, C.byref(pk), mask.ctypes.data, C.byref(st)))
        return DeviceFlatten.Result(pk, mask[:n], st)

    def close(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_hc_devflat_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class GamDevice:
    """The GAM front end on the device (vgan_gamdev; csrc/gam_kernels.hip): a BGZF GAM file's bytes -> the parser's arrays in HBM, as
    kernels (inflate, framing, protobuf wire walk).  parse() raises NativeError (VGAN_EIO) for input the device cannot take: the caller
    then reads the file through AlnParts / GamStream."""

    _NAMES = ("inflated_bytes", "messages", "reads", "mappings", "edits", "edit_seq_bytes", "quality_bytes")

    def __init__(self, device=0):
        self._h = N.vp()
        N.check(N.lib().vgan_gamdev_create(device, None, C.byref(self._h)))
        self.sizes, self.ms = {k: 0 for k in self._NAMES}, {}

    @classmethod
    def open(cls, data, device=0, keep_unmapped=False):
        """vgan_gamdev_open: create + parse, the member index made before the first HIP call."""
        self = cls.__new__(cls)
        self._h = N.vp()
        buf = np.frombuffer(data, np.uint8)
        N.check(N.lib().vgan_gamdev_open(device, None, buf.ctypes.data, len(data), int(keep_unmapped), C.byref(self._h)))
        return self._sizes()

    def parse(self, data, keep_unmapped=False):
        buf = np.frombuffer(data, np.uint8)
        N.check(N.lib().vgan_gamdev_parse(self._h, buf.ctypes.data, len(data), int(keep_unmapped)))
        return self._sizes()

    def parse_piece(self, data, piece, carry=None, piece_bytes=0, keep_unmapped=False, tail_bytes=0):
        """vgan_gampipe_parse_piece: piece `piece` of the plan over `data` through this object, framed from `carry` (what the piece before
        left: None for piece 0).  Returns the state for the next piece (GamCarry)."""
        carry = carry or GamCarry()
        buf = np.frombuffer(data, np.uint8)
        o = N.GamPipeOpts(int(piece_bytes), 0, int(keep_unmapped), 0, 0, int(tail_bytes))
        N.check(N.lib().vgan_gampipe_parse_piece(self._h, buf.ctypes.data, len(data), C.byref(o), int(piece), C.byref(carry._h)))
        self._sizes()
        return carry

    def _sizes(self):
        sizes, ms = np.zeros(8, np.uint64), np.zeros(4)
        N.check(N.lib().vgan_gamdev_sizes(self._h, sizes.ctypes.data, ms.ctypes.data))
        self.sizes = dict(zip(self._NAMES, (int(x) for x in sizes[:7])))
        self.ms = dict(zip(("upload", "inflate", "frame", "parse"), (float(x) for x in ms)))
        return self

    def mark_duplicates(self):
        """src/rmdup.cpp's marks of the parse's reads, left on the device; returns their number."""
        nd = C.c_int64(0)
        N.check(N.lib().vgan_gamdev_mark_duplicates(self._h, C.byref(nd)))
        return int(nd.value)

    def picked_parts(self, read_mask, keep_unmapped=False):
        """The reads read_mask names (the device flatten's host_mask), parsed on the host from their messages: an AlnParts of one slice."""
        m = np.ascontiguousarray(read_mask, np.uint8)
        assert len(m) == self.sizes["reads"]
        nm, nb = C.c_uint64(0), C.c_uint64(0)
        N.check(N.lib().vgan_gamdev_pick(self._h, m.ctypes.data, C.byref(nm), C.byref(nb)))
        offs, byts = np.zeros(int(nm.value) + 1, np.uint64), np.zeros(max(int(nb.value), 1), np.uint8)
        N.check(N.lib().vgan_gamdev_picked(self._h, offs.ctypes.data, byts.ctypes.data))
        h = N.vp()
        N.check(N.lib().vgan_alnparts_from_messages(byts.ctypes.data, offs.ctypes.data, int(nm.value), int(keep_unmapped), 0, C.byref(h)))
        return AlnParts(h)

    def drop_bytes(self, inflated=False):
        """vgan_gamdev_drop_bytes: the file's bytes (and, inflated=True, their inflated form: picked_parts is over then) go back to the device."""
        N.check(N.lib().vgan_gamdev_drop_bytes(self._h, 2 if inflated else 1))

    def close(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_gamdev_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class GamCarry:
    """What a piece's framing leaves for the next piece (opaque: vgan_gampipe_carry)."""

    def __init__(self):
        self._h = N.vp()

    def __del__(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_gampipe_carry_free(self._h)
            self._h = None


def gampipe_plan(data, piece_bytes=0):
    """The pieces a BGZF buffer is cut into: [(offset in the file, compressed bytes, inflated bytes)]."""
    buf = np.frombuffer(data, np.uint8)
    o = N.GamPipeOpts(int(piece_bytes), 0, 0, 0, 0, 0)
    n = N.lib().vgan_gampipe_plan(buf.ctypes.data, len(data), C.byref(o), None, None, None, 0)
    if n < 0:
        N.check(int(n))
    a, b, c = (np.zeros(max(int(n), 1), np.uint64) for _ in range(3))
    N.lib().vgan_gampipe_plan(buf.ctypes.data, len(data), C.byref(o), a.ctypes.data, b.ctypes.data, c.ctypes.data, int(n))
    return [(int(a[i]), int(b[i]), int(c[i])) for i in range(int(n))]


def accumulate_gam_bytes(ctxs, graph, data, piece_bytes=0, slots=0, mark_duplicates=False, keep_unmapped=False, n_threads=0, tail_bytes=0):
    """vgan_hc_accumulate_gam_bytes: a BGZF GAM's bytes through the device front end's pipeline into the contexts (piece i -> context
    i mod n).  Returns (FlattenStats, pipeline statistics as a dict)."""
    buf = np.frombuffer(data, np.uint8)
    o = N.GamPipeOpts(int(piece_bytes), int(slots), int(keep_unmapped), int(mark_duplicates), int(n_threads), int(tail_bytes))
    arr = (N.vp * len(ctxs))(*[c._h for c in ctxs])
    st, ps = N.FlattenStats(), N.GamPipeStats()
    N.check(N.lib().vgan_hc_accumulate_gam_bytes(arr, len(ctxs), graph._h, buf.ctypes.data, len(data), C.byref(o), C.byref(st), C.byref(ps)))
    return st, ps.as_dict()


class HcContext:
    """One HaploCart device context (vgan_hc_ctx) on one GPU."""

    def __init__(self, graph, background_error_prob=0.0001, use_background_error_prob=False,
                 is_consensus_fasta=False, device=0):
        self.graph = graph
        self._h = N.vp()
        p = N.HcParams(background_error_prob, int(use_background_error_prob), int(is_consensus_fasta))
        N.check(N.lib().vgan_hc_create(C.byref(graph.view), C.byref(p), device, C.byref(self._h)))
        self.n_paths = graph.n_paths
        self.device = device

    def set_stream(self, stream_ptr):
        N.check(N.lib().vgan_hc_set_stream(self._h, stream_ptr))

    def use_torch_stream(self):
        self.set_stream(N.torch_stream_ptr(self.device))

    def set_mode(self, mode):
        N.check(N.lib().vgan_hc_set_mode(self._h, mode))

    def reset(self):
        N.check(N.lib().vgan_hc_reset(self._h))

    def accumulate(self, batch):
        """A packed batch (HostBatch(packed=True) or its DeviceBatch): the packed reads, then the others."""
        pk = getattr(batch, "pk", None)
        if pk is not None:
            N.check(N.lib().vgan_hc_accumulate_packed(self._h, C.byref(pk)))
            if batch.c.n_reads == 0:
                return
        N.check(N.lib().vgan_hc_accumulate(self._h, C.byref(batch.c)))

    def validate_packed(self, batch):
        N.check(N.lib().vgan_hc_packed_validate(self._h, C.byref(batch.pk)))

    def segment_weights_packed(self, batch):
        """D_m per segment of the packed part, in the packed batch's segment order."""
        D = np.zeros(batch.pk.n_segments)
        N.check(N.lib().vgan_hc_segment_weights_packed(self._h, C.byref(batch.pk), D.ctypes.data))
        return D

    def segment_scalars(self, batch):
        S = np.zeros(batch.n_segments)
        U = np.zeros(batch.n_segments)
        N.check(N.lib().vgan_hc_segment_scalars(self._h, C.byref(batch.c), S.ctypes.data, U.ctypes.data))
        return S, U

    def validate(self, batch):
        """Raises NativeError unless the (host) batch satisfies the batch and tile contracts for this context's graph."""
        N.check(N.lib().vgan_hc_batch_validate(self._h, C.byref(batch.c)))

    def segment_weights(self, batch):
        """D_m = S_m - U_m per segment through the routed kernels (the tiled one for the batch's tileable reads)."""
        D = np.zeros(batch.n_segments)
        N.check(N.lib().vgan_hc_segment_weights(self._h, C.byref(batch.c), D.ctypes.data))
        return D

    def read_loglik(self, batch):
        out = np.zeros((batch.n_reads, self.n_paths))
        N.check(N.lib().vgan_hc_read_loglik(self._h, C.byref(batch.c), out.ctypes.data))
        return out

    def finalize(self, device_out=None):
        """final_vec[P] as numpy (synchronises); device_out: optional torch float64 tensor filled on the stream."""
        out = np.zeros(self.n_paths)
        N.check(N.lib().vgan_hc_finalize(self._h, device_out.data_ptr() if device_out is not None else None,
                                         out.ctypes.data))
        return out

    def finalize_device(self, device_out):
        N.check(N.lib().vgan_hc_finalize(self._h, device_out.data_ptr(), None))

    def profile_enable(self, enable=True, segment_only=False):
        """HIP events around the context's kernels (segment_only: around the segment kernel alone -- a pair of events is ~8 us of stream time)."""
        N.check(N.lib().vgan_hc_profile_enable(self._h, 2 if (enable and segment_only) else int(bool(enable))))

    def profile_read(self):
        """{kernel: (summed device ms, launches)} measured with HIP events on the context's stream."""
        ms = np.zeros(5)
        n = np.zeros(5, np.uint64)
        N.check(N.lib().vgan_hc_profile_read(self._h, ms.ctypes.data, n.ctypes.data))
        names = ("segment", "sweep_segments", "sweep_nodes", "finish", "pack")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def synchronize(self):
        N.check(N.lib().vgan_hc_synchronize(self._h))

    def posterior(self, final_vec, predicted=None):
        fv = np.ascontiguousarray(final_vec, np.float64)
        if predicted is None:
            predicted = self.graph.path_names[self.argmax(fv)]
        buf = C.create_string_buffer(1 << 20)
        conf = np.zeros(4096)
        n = N.check(N.lib().vgan_hc_posterior(self._h, fv.ctypes.data, predicted.encode(), buf, 1 << 20,
                                              conf.ctypes.data, 4096))
        names = buf.value.decode().split("\n")[:n]
        return [(names[i], float(conf[i]), i) for i in range(n)]

    @staticmethod
    def argmax(final_vec):
        fv = np.ascontiguousarray(final_vec, np.float64)
        return N.check(N.lib().vgan_hc_argmax(fv.ctypes.data, len(fv)))

    def close(self):
        if getattr(self, "_h", None) and N is not None:
            N.lib().vgan_hc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def reduce_info():
    """(wall ms of the last RCCL communicator set-up of vgan_hc_reduce, number of set-ups in this process)."""
    ms, n = C.c_double(0.0), C.c_int(0)
    N.check(N.lib().vgan_hc_reduce_info(C.byref(ms), C.byref(n)))
    return ms.value, n.value


def reduce_contexts(ctxs):
    """Sum of the contexts' final_vec through vgan_hc_reduce (RCCL between distinct GPUs, through the host for contexts sharing
    a device or with VGAN_HC_REDUCE=host).  Returns (final_vec, used_rccl)."""
    arr = (N.vp * len(ctxs))(*[c._h for c in ctxs])
    out = np.zeros(ctxs[0].n_paths)
    used = C.c_int(0)
    N.check(N.lib().vgan_hc_reduce(arr, len(ctxs), out.ctypes.data, C.byref(used)))
    return out, bool(used.value)


def synth_graph(seed=0x76676131, genome_len=16569, n_nodes=11821, n_paths=5179):
    cfg = N.SynthGraphCfg(seed, genome_len, n_nodes, n_paths)
    h = N.vp()
    N.check(N.lib().vgan_synth_hc_graph(C.byref(cfg), C.byref(h)))
    return Graph(h)


def synth_reads(graph, n_reads, seed=0x76676131, read_len=150, indel_rate=0.005, softclip_rate=0.01,
                low_mapq_rate=0.1, errors=True, first_read=0):
    """Reads [first_read, first_read + n_reads) of the stream the seed defines (read i depends on (seed, i) only)."""
    cfg = N.SynthReadsCfg(seed, n_reads, read_len, indel_rate, softclip_rate, low_mapq_rate, int(errors), first_read)
    h = N.vp()
    N.check(N.lib().vgan_synth_hc_reads(graph._h, C.byref(cfg), C.byref(h)))
    return AlnSet(h)
