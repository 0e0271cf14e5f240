"""Test-side GAM (vg Alignment stream) decoder/encoder in pure Python.

Independent of the product's C++ reader (vgan_amd/csrc/host_gam.cpp) so the two can be
checked against each other.  Wire layout verified against the reference's fixture
test/reconstructInputSeq/test_reads.gam:

  file   = gzip/BGZF members; payload = groups
  group  = varint count, then `count` items, each varint length + bytes; the first
           item of each group is the type tag "GAM"
  Alignment{1 sequence, 2 path, 3 name, 4 quality(raw phred), 5 mapping_quality,
            6 score, 16 identity(double)}
  Path{1 name, 2 mapping}  Mapping{1 position, 2 edit, 5 rank}
  Position{1 node_id, 2 offset, 4 is_reverse}  Edit{1 from_length, 2 to_length, 3 sequence}

This is test infrastructure only.
"""
import gzip
import struct
import zlib


def _varint(buf, i):
    shift = 0
    val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _fields(buf):
    i = 0
    n = len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v = buf[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 5:
            v = buf[i:i + 4]
            i += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fno, wt, v


def _s64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def parse_edit(buf):
    e = {"from_length": 0, "to_length": 0, "sequence": b""}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            e["from_length"] = _s64(v)
        elif fno == 2:
            e["to_length"] = _s64(v)
        elif fno == 3:
            e["sequence"] = bytes(v)
    return e


def parse_position(buf):
    p = {"node_id": 0, "offset": 0, "is_reverse": False}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            p["node_id"] = _s64(v)
        elif fno == 2:
            p["offset"] = _s64(v)
        elif fno == 4:
            p["is_reverse"] = bool(v)
    return p


def parse_mapping(buf):
    m = {"position": {"node_id": 0, "offset": 0, "is_reverse": False}, "edit": [], "rank": 0}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            m["position"] = parse_position(v)
        elif fno == 2:
            m["edit"].append(parse_edit(v))
        elif fno == 5:
            m["rank"] = v
    return m


def parse_path(buf):
    p = {"name": b"", "mapping": []}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            p["name"] = bytes(v)
        elif fno == 2:
            p["mapping"].append(parse_mapping(v))
    return p


def parse_alignment(buf):
    a = {"sequence": b"", "path": {"name": b"", "mapping": []}, "name": b"", "quality": b"",
         "mapping_quality": 0, "score": 0, "identity": 0.0}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            a["sequence"] = bytes(v)
        elif fno == 2:
            a["path"] = parse_path(v)
        elif fno == 3:
            a["name"] = bytes(v)
        elif fno == 4:
            a["quality"] = bytes(v)
        elif fno == 5:
            a["mapping_quality"] = v
        elif fno == 6:
            a["score"] = v
        elif fno == 16 and wt == 1:
            a["identity"] = struct.unpack("<d", v)[0]
    return a


def gunzip_all(data):
    """Inflate a concatenation of gzip members (plain gzip or BGZF)."""
    out = []
    while data:
        d = zlib.decompressobj(16 + zlib.MAX_WBITS)
        out.append(d.decompress(data))
        data = d.unused_data
    return b"".join(out)


def read_gam(path_or_bytes):
    raw = path_or_bytes
    if not isinstance(raw, (bytes, bytearray)):
        with open(path_or_bytes, "rb") as f:
            raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gunzip_all(raw)
    buf = memoryview(raw)
    i = 0
    out = []
    while i < len(buf):
        count, i = _varint(buf, i)
        first = True
        for _ in range(count):
            ln, i = _varint(buf, i)
            item = buf[i:i + ln]
            i += ln
            if first:
                first = False
                if bytes(item) == b"GAM":
                    continue
            out.append(parse_alignment(item))
    return out


# ---------------------------------------------------------------- encoder
def _enc_varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fno, payload):
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(fno, v):
    return _enc_varint(fno << 3) + _enc_varint(v)


def enc_edit(e):
    out = b""
    if e.get("from_length", 0):
        out += _vi(1, e["from_length"])
    if e.get("to_length", 0):
        out += _vi(2, e["to_length"])
    if e.get("sequence", b""):
        out += _ld(3, e["sequence"])
    return out


def enc_mapping(m):
    pos = m["position"]
    p = b""
    if pos.get("node_id", 0):
        p += _vi(1, pos["node_id"])
    if pos.get("offset", 0):
        p += _vi(2, pos["offset"])
    if pos.get("is_reverse", False):
        p += _vi(4, 1)
    out = _ld(1, p)
    for e in m["edit"]:
        out += _ld(2, enc_edit(e))
    if m.get("rank", 0):
        out += _vi(5, m["rank"])
    return out


def enc_alignment(a):
    out = b""
    if a.get("sequence"):
        out += _ld(1, a["sequence"])
    path = b""
    if a["path"].get("name"):
        path += _ld(1, a["path"]["name"])
    for m in a["path"]["mapping"]:
        path += _ld(2, enc_mapping(m))
    out += _ld(2, path)
    if a.get("name"):
        out += _ld(3, a["name"])
    if a.get("quality"):
        out += _ld(4, a["quality"])
    if a.get("mapping_quality", 0):
        out += _vi(5, a["mapping_quality"])
    if a.get("score", 0):
        out += _vi(6, a["score"])
    if a.get("identity", 0.0) != 0.0:
        out += _enc_varint((16 << 3) | 1) + struct.pack("<d", a["identity"])
    return out


def write_gam(alns, group=512, compress=True):
    body = bytearray()
    for g in range(0, len(alns), group):
        chunk = alns[g:g + group]
        body += _enc_varint(len(chunk) + 1)
        body += _enc_varint(3) + b"GAM"
        for a in chunk:
            msg = enc_alignment(a)
            body += _enc_varint(len(msg)) + msg
    return gzip.compress(bytes(body)) if compress else bytes(body)
