"""CLI surface of `vgan haplocart` (reference src/HaploCart.cpp:87-261): validation errors and the duplicate
removal rule, on CPU."""
import os
import subprocess

import numpy as np
import pytest

from vgan_amd import _native as N
from vgan_amd import haplocart as hc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VGAN = os.path.join(ROOT, "vgan_amd", "bin", "vgan")


def run(*args):
    return subprocess.run([VGAN] + list(args), capture_output=True, text=True)


def test_cli_validation_errors(tmp_path):
    assert os.path.exists(VGAN), "build the CLI with __graft_entry__.build()"
    r = run("haplocart", "-e", "1.5")
    assert r.returncode != 0 and "option -e is not a valid probability" in r.stderr      # HaploCart.cpp:107-113
    r = run("haplocart", "-t", "0")
    assert r.returncode != 0 and "invalid number of threads" in r.stderr                # HaploCart.cpp:183-194
    r = run("haplocart", "-t", "-2")
    assert r.returncode != 0
    r = run("haplocart", "-g", str(tmp_path / "missing.gam"))
    assert r.returncode != 0 and "does not exist" in r.stderr
    r = run("haplocart", "-fq1", "x.fq")
    assert r.returncode != 0 and "giraffe" in r.stderr
    r = run("version")
    assert r.returncode == 0 and "ABI" in r.stdout
    # numeric options are whole-token parses that name the option (not an uncaught stoi / stod)
    for flag, val, word in (("-t", "abc", "integer"), ("-t", "4x", "integer"), ("-t", "99999999999999999999", "integer"),
                            ("-e", "xyz", "number"), ("-e", "1e999", "number"), ("-e", "nan", "number"),
                            ("--device", "one", "integer"), ("--device", "-3", "non-negative")):
        r = run("haplocart", flag, val, "-g", "/dev/null")
        assert r.returncode == 1 and flag in r.stderr and word in r.stderr, (flag, val, r.stderr)
    for flag in ("-t", "-e", "-g", "-o", "-pf", "-s", "--hc-files", "--device"):
        r = run("haplocart", flag)
        assert r.returncode == 1 and "needs a value" in r.stderr


def test_cli_garbage_arguments_are_errors_not_crashes(tmp_path):
    """Random argument vectors end with exit code 0 (help) or 1 (a message on stderr) -- never a signal."""
    import random
    rng = random.Random(3)
    vocab = ["haplocart", "version", "euka", "soibean", "--dbprefix", "--chains", "-k", "--randStart", "-P", "--iter", "--entropy", "--euka_dir", "--outGroup", "--minBins", "--no-mcmc", "-l", "-g", "-e", "-t", "-o", "-pf", "-s", "-np", "-q", "-d", "-w", "-z", "-i", "-j", "-f", "-fq1",
             "--hc-files", "--device", "--per-read", "--keep-duplicates", "-h", "", "0", "-1", "1e-3", "\xff\xfe", "a" * 5000,
             str(tmp_path), "/dev/null", "/nonexistent/x", "--", "-", "%s%n", "9" * 40]
    for _ in range(150):
        args = [rng.choice(vocab) for _ in range(rng.randrange(0, 7))]
        if rng.random() < 0.7:
            args.insert(0, "haplocart")
        r = run(*args)
        assert r.returncode in (0, 1), (args, r.returncode, r.stderr[-200:])
        assert r.returncode == 0 or r.stderr.strip(), args


def test_euka_cli_validation_errors(tmp_path):
    """`vgan euka` flag checks (reference src/Euka.cpp:195-340,373-395)."""
    for args, msg in ((("--entropy", "6"), "entropy thresold is too stringent"), (("--minBins", "21"), "minimum number of bins exceeds"),
                      (("--maxBins", "21"), "maximum number of bins exceeds"), (("-t", "0"), "invalid number of threads"),
                      (("-t", "-2"), "invalid number of threads"), (("-fq1", "x.fa"), "must be FASTQ, not FASTA"),
                      (("-fq1", "x.fq", "-fq2", "y.fasta.gz"), "must be FASTQ, not FASTA"),
                      (("-fq1", "x.fq", "-fq2", "y.fq", "-i"), "expects only one FASTQ file"), (("-fq1", "x.fq"), "giraffe"),
                      (("--minMQ", "61"), "0..60"), (("--iter", "-5"), "must not be negative"), (("--iter", "abc"), "needs an integer"),
                      (("--entropy", "x"), "needs a number"), (("--bogus",), "unrecognized option"), (("--iter",), "needs a value"),
                      (("--euka_dir", str(tmp_path)), "euka_db.og does not exist.")):
        r = run("euka", *args)
        assert r.returncode == 1 and msg in r.stderr, (args, r.stderr)
    assert run("euka", "-h").returncode == 0
    from vgan_amd import euka as ek
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    g, db, a = ek.synth_euka(300, None, seed=2, n_clades=5, nodes_per_clade=120)
    util.write_euka_db(db, g, tmp_path)
    r = run("euka", "--euka_dir", str(tmp_path))
    assert r.returncode == 1 and "no input file given" in r.stderr
    r = run("euka", "--euka_dir", str(tmp_path), "-g", str(tmp_path / "none.gam"))
    assert r.returncode == 1 and "does not exist" in r.stderr
    a.write_gam(str(tmp_path / "r.gam"))
    r = run("euka", "--euka_dir", str(tmp_path), "-g", str(tmp_path / "r.gam"), "--outGroup", "nope")
    assert r.returncode == 1 and "Outgroup not found in reference graph" in r.stderr
    r = run("euka", "--euka_dir", str(tmp_path), "-g", str(tmp_path / "r.gam"), "--iter", "10", "--burnin", "10")
    assert r.returncode == 1 and "--iter must exceed --burnin" in r.stderr
    if N.lib().vgan_device_count() <= 0:  # no CPU fallback for the per-read pass
        r = run("euka", "--euka_dir", str(tmp_path), "-g", str(tmp_path / "r.gam"), "-o", str(tmp_path / "out"))
        assert r.returncode == 1 and "no HIP device" in r.stderr and not list(tmp_path.glob("out_*"))


def test_soibean_cli_validation_errors(tmp_path):
    """`vgan soibean` flag checks (reference src/soibean.cpp:263-441)."""
    for args, msg in ((("-g", "x.gam"), "No database specified"), (("--dbprefix", "T", "-t", "0"), "invalid number of threads"),
                      (("--dbprefix", "T", "-fq1", "r.fasta"), "must be FASTQ, not FASTA"), (("--dbprefix", "T", "-fq1", "r.fq"), "giraffe"),
                      (("--dbprefix", "T", "--iter", "10", "--burnin", "20"), "must be higher than the burn-in"),
                      (("--dbprefix", "T", "--deam5p", "x.prof"), "damage profiles do not exist"),
                      (("--dbprefix", "T", "-P", "0"), "must be positive"), (("--dbprefix", "T", "--chains", "x"), "needs an integer"),
                      (("--dbprefix", "T", "--nope"), "unrecognized option"), (("--dbprefix", "T", "--alignment-detail"), "not kept on the GPU path"),
                      (("--dbprefix", "T", "--soibean_dir", str(tmp_path)), "T.og does not exist.")):
        r = run("soibean", *args)
        assert r.returncode == 1 and msg in r.stderr, (args, r.stderr)
    assert run("soibean", "-h").returncode == 0
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_sb_chain_cpu import _newick_of
    g = hc.synth_graph(seed=17, genome_len=900, n_nodes=600, n_paths=12)
    a = hc.synth_reads(g, 60, seed=1, read_len=60)
    g.write(str(tmp_path))
    os.rename(str(tmp_path / "graph.gfa"), str(tmp_path / "T.gfa"))
    r = run("soibean", "--dbprefix", "T", "--soibean_dir", str(tmp_path))
    assert r.returncode == 1 and "T.new.dnd does not exist." in r.stderr
    (tmp_path / "tree_dir").mkdir()
    (tmp_path / "tree_dir" / "T.new.dnd").write_text(_newick_of(g))
    a.write_gam(str(tmp_path / "r.gam"))
    r = run("soibean", "--dbprefix", "T", "--soibean_dir", str(tmp_path), "-g", str(tmp_path / "r.gam"))
    assert r.returncode == 1 and "Failed to open the base frequency file." in r.stderr
    (tmp_path / "soibean_db.baseFreq").write_text("T .3 .2 .2 .3\n")
    (tmp_path / "tree_dir" / "T.new.dnd").write_text("(a:1,b:2)c;")
    r = run("soibean", "--dbprefix", "T", "--soibean_dir", str(tmp_path), "-g", str(tmp_path / "r.gam"))
    assert r.returncode == 1 and "number of tree nodes and paths in the graph is unequal" in r.stderr
    (tmp_path / "tree_dir" / "T.new.dnd").write_text(_newick_of(g))
    if N.lib().vgan_device_count() <= 0:
        r = run("soibean", "--dbprefix", "T", "--soibean_dir", str(tmp_path), "-g", str(tmp_path / "r.gam"), "-o", str(tmp_path / "o_"))
        assert r.returncode == 1 and "no HIP device" in r.stderr and not list(tmp_path.glob("o_*"))


def test_cli_needs_a_gpu_not_a_fallback(tmp_path):
    if N.lib().vgan_device_count() > 0:
        pytest.skip("a GPU is visible")
    g = hc.synth_graph(seed=5, genome_len=800, n_nodes=560, n_paths=40)
    a = hc.synth_reads(g, 50, seed=1, read_len=100)
    g.write(str(tmp_path))
    a.write_gam(str(tmp_path / "r.gam"))
    r = run("haplocart", "-g", str(tmp_path / "r.gam"), "--hc-files", str(tmp_path), "-q")
    assert r.returncode != 0 and "no HIP device" in r.stderr and r.stdout == ""


def test_duplicate_marks_match_the_quadratic_rule():
    g = hc.synth_graph(seed=5, genome_len=300, n_nodes=210, n_paths=10)
    a = hc.synth_reads(g, 400, seed=3, read_len=40)
    arr = a.arrays()
    first = [(int(arr["m_node"][arr["map_off"][r]]), int(arr["m_offset"][arr["map_off"][r]])) for r in range(a.n_reads)]
    ref = np.zeros(a.n_reads, bool)
    for i in range(a.n_reads):          # rmdup.cpp:20-41,82-93 literally
        for j in range(i + 1, a.n_reads):
            if first[j] == first[i]:
                ref[j] = True
    got = a.mark_duplicates()
    assert got.sum() > 0 and np.array_equal(got, ref)
    kept = a.without(got)
    assert kept.n_reads == int((~ref).sum())
    k = kept.arrays()
    assert [bytes(k["name"][k["name_off"][i]:k["name_off"][i + 1]]) for i in range(3)] == \
        [bytes(arr["name"][arr["name_off"][i]:arr["name_off"][i + 1]]) for i in np.flatnonzero(~ref)[:3]]


def test_json_dump_of_the_alignments(tmp_path, golden_dir):
    """-j / -jf (HaploCart.cpp:146-152,231; readGAM.h:37-38): every Alignment as a line of JSON in protobuf's mapping -- held
    against the test-side GAM decoder on the reference's own fixture and on a re-encoded set with every edit kind."""
    import base64
    import ctypes as C
    import json

    import gamio
    from vgan_amd import _native as N

    r = run("haplocart", "-j", "-g", "/dev/null")
    assert r.returncode != 0 and "cannot invoke -j without -jf" in r.stderr
    src = os.path.join(golden_dir, "reconstruct", "test_reads.gam")
    alns = gamio.read_gam(src)
    # a second file from the test-side encoder: negative offsets cannot occur, but reverse strands, insertions, deletions,
    # substitutions, an empty quality string and a name that needs escaping do
    extra = [dict(a) for a in alns]
    extra[0]["name"] = b'q"uote\\back<tag>\ttab'
    extra[1]["quality"] = b""
    second = str(tmp_path / "second.gam")
    open(second, "wb").write(gamio.write_gam(extra, group=3))
    for path, want in ((src, alns), (second, extra)):
        out = str(tmp_path / "dump.json")
        n = C.c_int64(0)
        N.check(N.lib().vgan_gam_dump_json(path.encode(), out.encode(), C.byref(n)))
        lines = open(out).read().splitlines()
        assert n.value == len(want) == len(lines)
        for ln, a in zip(lines, want):
            d = json.loads(ln)
            assert list(d.keys()) == [k for k in ("sequence", "path", "name", "quality", "mapping_quality", "score", "identity", "time_used") if k in d]
            assert d.get("sequence", "").encode() == a["sequence"] and d.get("name", "").encode() == a["name"]
            assert base64.b64decode(d.get("quality", "")) == a["quality"]
            assert d.get("mapping_quality", 0) == a["mapping_quality"] and d.get("score", 0) == a["score"]
            assert float(d.get("identity", 0.0)) == a["identity"]  # the shortest text that reads back
            maps = d.get("path", {}).get("mapping", [])
            assert len(maps) == len(a["path"]["mapping"])
            for jm, m in zip(maps, a["path"]["mapping"]):
                pos = jm.get("position", {})
                assert int(pos.get("node_id", "0")) == m["position"]["node_id"] and isinstance(pos.get("node_id", "0"), str)
                assert int(pos.get("offset", "0")) == m["position"]["offset"] and bool(pos.get("is_reverse", False)) == bool(m["position"]["is_reverse"])
                assert [(e.get("from_length", 0), e.get("to_length", 0), e.get("sequence", "").encode()) for e in jm.get("edit", [])] == \
                       [(e["from_length"], e["to_length"], e["sequence"]) for e in m["edit"]]
    # a truncated file is an error, not a shorter dump
    bad = str(tmp_path / "bad.gam")
    raw = gamio.gunzip_all(open(src, "rb").read())
    open(bad, "wb").write(raw[:len(raw) - 7])
    assert N.lib().vgan_gam_dump_json(bad.encode(), str(tmp_path / "bad.json").encode(), None) < 0


def test_the_working_process_and_its_early_leaving_parent(tmp_path):
    """cli_util.h EarlyLeave (opt-in: VGAN_EARLY_LEAVE=1): the subcommand runs in a child, the parent leaves with the child's
    code as soon as the outputs are flushed.  Exit codes and messages are those of the run in one process (the default); a
    working process that is killed takes its parent with it by the same signal; nothing stays behind holding the caller's
    pipes; vgan as pid 1 of a pid namespace (`docker run image vgan ...`) works the same."""
    import shutil
    import signal
    import time
    early = dict(os.environ, VGAN_EARLY_LEAVE="1")
    for args in (["haplocart", "-e", "1.5"], ["haplocart", "-g", str(tmp_path / "missing.gam")], ["euka", "-t", "0"], ["soibean", "--iter", "x"]):
        a = subprocess.run([VGAN] + args, capture_output=True, text=True, env=early)
        b = subprocess.run([VGAN] + args, capture_output=True, text=True)
        assert a.returncode == b.returncode != 0 and a.stderr == b.stderr and a.stdout == b.stdout, args
    # as pid 1: the working child's parent legitimately has pid 1
    for ns in (["unshare", "-p", "-f"], ["unshare", "-p", "-f", "-r"]):  # (a new pid namespace needs privilege or a user namespace)
        if not shutil.which("unshare") or subprocess.run(ns + [VGAN, "haplocart", "-h"], capture_output=True).returncode in (1, 126, 127) and \
                subprocess.run(ns + ["true"], capture_output=True).returncode != 0:
            continue
        b = subprocess.run([VGAN, "haplocart", "-h"], capture_output=True, text=True)
        a = subprocess.run(ns + [VGAN, "haplocart", "-h"], capture_output=True, text=True, env=early)
        if a.returncode == 126:  # the binary cannot be executed from inside that namespace (path permissions)
            continue
        assert a.returncode == b.returncode and a.stdout == b.stdout and a.stderr == b.stderr
        break
    # a working process blocked on its input (a FIFO nobody writes to) is killed: the parent ends by the same signal
    fifo = str(tmp_path / "never.gam")
    os.mkfifo(fifo)
    g = hc.synth_graph(seed=5, genome_len=400, n_nodes=280, n_paths=8)
    g.write(str(tmp_path))
    p = subprocess.Popen([VGAN, "haplocart", "-g", fifo, "--hc-files", str(tmp_path), "-q"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=early)
    kids = []
    for _ in range(100):  # the child appears
        time.sleep(0.02)
        try:
            kids = [int(x) for x in open("/proc/%d/task/%d/children" % (p.pid, p.pid)).read().split()]
        except OSError:
            kids = []
        if kids:
            break
    assert len(kids) == 1
    os.kill(kids[0], signal.SIGTERM)
    p.communicate(timeout=20)
    assert p.returncode == -signal.SIGTERM
    # and the other way round: the parent is killed, the working process does not outlive it
    p = subprocess.Popen([VGAN, "haplocart", "-g", fifo, "--hc-files", str(tmp_path), "-q"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=early)
    for _ in range(100):
        time.sleep(0.02)
        try:
            kids = [int(x) for x in open("/proc/%d/task/%d/children" % (p.pid, p.pid)).read().split()]
        except OSError:
            kids = []
        if kids:
            break
    assert len(kids) == 1
    p.kill()
    p.communicate(timeout=20)
    for _ in range(100):
        if not os.path.exists("/proc/%d" % kids[0]) or open("/proc/%d/stat" % kids[0]).read().split()[2] == "Z":
            break
        time.sleep(0.05)
    else:
        raise AssertionError("the working process outlived its parent")
