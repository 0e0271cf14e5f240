"""vgan_amd -- MI355X-native per-read likelihood engine for vgan (HaploCart / euka / soibean hot path).

The product is the C-ABI library `vgan_amd/lib/libvgan_gpu.so` (HIP kernels for gfx950 + the C++ host front
end, see include/vgan_gpu.h).  This package is the thin Python plumbing around it used by tests and bench.py;
it has no CPU implementation of the hot path and raises if the native library is missing.
"""
from ._native import lib, load, NativeError  # noqa: F401
from .haplocart import (  # noqa: F401
    Graph, AlnSet, AlnParts, HostBatch, ArrayBatch, DeviceBatch, HcContext,
    MODE_NODE_WEIGHTS, MODE_PER_READ, MODE_PER_READ_DENSE,
    synth_graph, synth_reads,
)
from .euka import EukaDb, Damage, EukaHostBatch, EukaDeviceBatch, EukaContext, synth_euka  # noqa: F401
from .soibean import SbHostBatch, SbContext  # noqa: F401
