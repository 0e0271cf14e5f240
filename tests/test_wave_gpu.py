"""GPU tests of the wave-owned segment kernel and its packed batch layout (hc_wave_kernels.hip, vgan_hc_pack), through the
C-ABI.  The oracle holds the path end to end in test_hc_gpu.py (every accumulate of a tileable read goes through this
kernel); here the two data paths of the product are held against each other per segment, both variants of the kernel are
forced, and the resident (packed once) route is checked against the per-call one."""
import os

import numpy as np
import pytest

import orc
import util
from vgan_amd import _native as N
from vgan_amd import haplocart as hc

pytestmark = pytest.mark.gpu


@pytest.fixture
def kernel_switch():
    """VGAN_HC_KERNEL=tile keeps the tileable reads on the LDS-tiled kernel, =wave sends them to the wave kernel whatever
    their shape (read per call by the library)."""
    old = os.environ.get("VGAN_HC_KERNEL")

    def use(which):
        os.environ["VGAN_HC_KERNEL"] = which
    yield use
    if old is None:
        os.environ.pop("VGAN_HC_KERNEL", None)
    else:
        os.environ["VGAN_HC_KERNEL"] = old


def test_wave_kernel_against_tile_kernel_and_general_kernel_per_segment(kernel_switch):
    g = hc.synth_graph(seed=11, genome_len=4000, n_nodes=2600, n_paths=300)
    a = hc.synth_reads(g, 20000, seed=12, read_len=150, low_mapq_rate=0.3)
    b = hc.HostBatch(g, a)
    assert b.n_tileable == b.n_reads
    for kw in (dict(), dict(background_error_prob=0.02, use_background_error_prob=True),
               dict(background_error_prob=0.01, use_background_error_prob=True, is_consensus_fasta=True)):
        ctx = hc.HcContext(g, **kw)
        S, U = ctx.segment_scalars(b)  # the general kernel (held against the oracle's literal loops in test_hc_gpu.py)
        kernel_switch("wave")
        wave = ctx.segment_weights(b)
        kernel_switch("tile")
        tile = ctx.segment_weights(b)
        tol = 5e-12 + 1e-13 * np.maximum(np.abs(S), np.abs(U))
        assert np.all(np.abs(wave - (S - U)) <= tol)
        assert np.all(np.abs(wave - tile) <= tol)
        fin = {}
        for which in ("wave", "tile"):
            kernel_switch(which)
            ctx.reset()
            ctx.accumulate(b)
            fin[which] = ctx.finalize()
        assert util.rel_err(fin["wave"], fin["tile"]) < 1e-12


@pytest.mark.parametrize("read_len", [40, 75, 150, 300, 500])
def test_both_variants_of_the_wave_kernel_against_the_oracle(read_len, kernel_switch):
    """Reads of 40..300 columns take the small variant (several reads per tile), 500-column reads the large one.  (Left to
    itself the library sends the shapes the LDS-tiled kernel is faster on to that kernel: VGAN_HC_KERNEL=wave overrides.)"""
    kernel_switch("wave")
    g = hc.synth_graph(seed=21, genome_len=3000, n_nodes=2000, n_paths=200)
    a = hc.synth_reads(g, 1500, seed=22 + read_len, read_len=read_len, indel_rate=0.05, softclip_rate=0.1, low_mapq_rate=0.3)
    b = hc.HostBatch(g, a)
    ctx = hc.HcContext(g)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    _, ref, _ = orc.hc_run(og, oa, n_threads=8, faithful=False)
    ctx.accumulate(b)
    got = ctx.finalize()
    assert util.rel_err(got, ref) < 1e-9
    S, U = ctx.segment_scalars(b)
    D = ctx.segment_weights(b)
    assert np.all(np.abs(D - (S - U)) <= 5e-12 + 1e-13 * np.maximum(np.abs(S), np.abs(U)))


def test_resident_packed_batch_equals_the_per_call_route(kernel_switch):
    g = hc.synth_graph(seed=31, genome_len=5000, n_nodes=3300, n_paths=400)
    a = hc.synth_reads(g, 30000, seed=32, read_len=150)
    hb = hc.HostBatch(g, a)
    ctx = hc.HcContext(g)
    ctx.accumulate(hb)  # host arrays: staged, packed into the context's scratch, wave kernel
    want = ctx.finalize()
    db = hc.DeviceBatch(hb, ctx=ctx)  # resident: packed once
    assert db.pack_ms is not None and db.c.packed
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(db)
        ctx.accumulate(db)
        assert util.rel_err(ctx.finalize(), 2 * want) < 1e-10, mode
    ctx.set_mode(hc.MODE_NODE_WEIGHTS)
    plain = hc.DeviceBatch(hb)  # resident without the companion: the LDS-tiled kernel takes it
    ctx.reset()
    ctx.accumulate(plain)
    assert util.rel_err(ctx.finalize(), want) < 1e-12
    kernel_switch("tile")  # the switch also holds for a batch that carries a companion
    ctx.reset()
    ctx.accumulate(db)
    assert util.rel_err(ctx.finalize(), want) < 1e-12


def test_a_companion_of_another_batch_is_refused_and_an_empty_one_is_fine():
    g = hc.synth_graph(seed=41, genome_len=2000, n_nodes=1300, n_paths=100)
    ctx = hc.HcContext(g)
    b1 = hc.DeviceBatch(hc.HostBatch(g, hc.synth_reads(g, 500, seed=1, read_len=100)), ctx=ctx)
    b2 = hc.DeviceBatch(hc.HostBatch(g, hc.synth_reads(g, 700, seed=2, read_len=100)), ctx=ctx)
    keep = b2.c.packed
    b2.c.packed = b1.c.packed
    with pytest.raises(N.NativeError):
        ctx.accumulate(b2)
    b2.c.packed = keep
    ctx.reset()
    ctx.accumulate(b2)
    assert np.all(np.isfinite(ctx.finalize()))
    # nothing tileable: the companion is empty and the general kernel does the work
    hb = hc.HostBatch(g, hc.synth_reads(g, 300, seed=3, read_len=100))
    arrays = {k: v for k, v in hb.arrays().items() if not k.startswith("_") and k != "read_src"}
    none_tileable = hc.ArrayBatch(arrays, n_tileable=0)
    all_tileable = hc.ArrayBatch(arrays, n_tileable=hb.n_tileable)
    outs = []
    for b in (none_tileable, all_tileable):
        ctx.reset()
        ctx.accumulate(b)
        outs.append(ctx.finalize())
    assert util.rel_err(outs[0], outs[1]) < 1e-12


def test_unsorted_and_sparse_batches_through_the_wave_window():
    """The wave's W window is placed per wave and re-placed when a tile leaves it: any read order gives the same sums, and
    so does a batch whose reads lie far apart on the graph."""
    g = hc.synth_graph(seed=51, genome_len=16569, n_nodes=11821, n_paths=500)
    a = hc.synth_reads(g, 4000, seed=52, read_len=150)
    hb = hc.HostBatch(g, a)
    arr = hb.arrays()
    ctx = hc.HcContext(g)
    ctx.accumulate(hb)
    want = ctx.finalize()
    rng = np.random.default_rng(5)
    R = hb.n_reads
    order = rng.permutation(R)
    so, co, qo = arr["read_seg_off"], arr["read_col_off"], arr["read_qual_off"]
    new = {k: [] for k in ("seg_node", "seg_start", "seg_len", "graph_seq", "algnseq", "qual")}
    offs = {"read_seg_off": [0], "read_col_off": [0], "read_qual_off": [0]}
    for r in order:
        for k in ("seg_node", "seg_start", "seg_len"):
            new[k].append(arr[k][so[r]:so[r + 1]])
        for k in ("graph_seq", "algnseq"):
            new[k].append(arr[k][co[r]:co[r + 1]])
        new["qual"].append(arr["qual"][qo[r]:qo[r + 1]])
        offs["read_seg_off"].append(offs["read_seg_off"][-1] + so[r + 1] - so[r])
        offs["read_col_off"].append(offs["read_col_off"][-1] + co[r + 1] - co[r])
        offs["read_qual_off"].append(offs["read_qual_off"][-1] + qo[r + 1] - qo[r])
    arrays = {k: np.concatenate(v) for k, v in new.items()}
    arrays.update({k: np.array(v, np.uint32) for k, v in offs.items()})
    arrays["read_algn_len"] = arr["read_algn_len"][order]
    arrays["read_mapq"] = arr["read_mapq"][order]
    shuffled = hc.ArrayBatch(arrays, n_tileable=R)
    ctx.validate(shuffled)
    ctx.reset()
    ctx.accumulate(shuffled)
    assert util.rel_err(ctx.finalize(), want) < 1e-11
