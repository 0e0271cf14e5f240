"""AddressSanitizer + UndefinedBehaviorSanitizer run of the host front end (CPU build only: GPU ASan is unavailable)."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_sources_under_asan_ubsan(tmp_path):
    srcs = sorted(glob.glob(os.path.join(ROOT, "vgan_amd/csrc/host/*.cpp")))
    srcs = [s for s in srcs if not s.endswith("_main.cpp")]
    exe = str(tmp_path / "host_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "vgan_amd/csrc"),
           os.path.join(ROOT, "tests/native/host_sanitizer_driver.cpp")] + srcs + ["-o", exe, "-lz", "-lpthread", "-ldl"]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    # once with the decoder's own segment size, once with segments of two BGZF blocks (groups and messages across every boundary)
    for extra in ({}, {"VGAN_GAM_SEG_BLOCKS": "2", "VGAN_GAM_THREADS": "5"}):
        r = subprocess.run([exe, os.path.join(ROOT, "tests/golden"), str(tmp_path)], capture_output=True, text=True, env=dict(env, **extra))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "host sanitizer driver: ok" in r.stdout


def test_host_sources_under_tsan(tmp_path):
    """ThreadSanitizer over the threaded front end (block-parallel inflate feeding the framing pass feeding the parser
    pool, parallel merges and flattening) on an input large enough for several slices and inflate runs."""
    srcs = sorted(glob.glob(os.path.join(ROOT, "vgan_amd/csrc/host/*.cpp")))
    srcs = [s for s in srcs if not s.endswith("_main.cpp")]
    exe = str(tmp_path / "host_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "vgan_amd/csrc"),
           os.path.join(ROOT, "tests/native/host_sanitizer_driver.cpp")] + srcs + ["-o", exe, "-lz", "-lpthread", "-ldl"]
    subprocess.check_call(cmd)
    out = tmp_path / "o"
    out.mkdir()
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    for extra in ({}, {"VGAN_GAM_SEG_BLOCKS": "2", "VGAN_GAM_THREADS": "6"}):
        r = subprocess.run([exe, os.path.join(ROOT, "tests/golden"), str(out), "40000"], capture_output=True, text=True, env=dict(env, **extra))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
        assert "host sanitizer driver: ok" in r.stdout and "ThreadSanitizer" not in r.stderr
