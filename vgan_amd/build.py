"""Build the native library in-tree: HIP kernels for gfx950 + the C++ host front end -> vgan_amd/lib/libvgan_gpu.so.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
TAG = os.environ.get("VGAN_BUILD_TAG", "")  # developer builds beside the product's: lib/libvgan_gpu<TAG>.so, build<TAG>/
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build" + TAG)
LIB = os.path.join(LIBDIR, "libvgan_gpu%s.so" % TAG)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("VGAN_EXTRA_FLAGS", "").split()  # developer builds, e.g. -DVGAN_PHASE_TIMING


def sources():
    hip, cpp = [], []
    for d, _, files in os.walk(CSRC):
        for f in sorted(files):
            p = os.path.join(d, f)
            if f.endswith(".hip"):
                hip.append(p)
            elif f.endswith(".cpp") and not f.endswith("_main.cpp"):
                cpp.append(p)
    return hip, cpp


def headers():
    out = [os.path.join(ROOT, "include", "vgan_gpu.h")]
    for d, _, files in os.walk(CSRC):
        out += [os.path.join(d, f) for f in files if f.endswith(".h")]
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False, force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hip, cpp = sources()
    hdrs = headers() + [os.path.abspath(__file__)]
    objs, cmds = [], []
    for src in hip + cpp:
        obj = os.path.join(OBJDIR, os.path.relpath(src, CSRC).replace(os.sep, "_") + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + COMMON
            if src.endswith(".hip"):
                cmd += ["--offload-arch=" + ARCH, "-save-temps=obj"] if os.environ.get("VGAN_SAVE_TEMPS") else ["--offload-arch=" + ARCH]
            else:
                cmd += ["-x", "c++"]
            cmd += ["-c", src, "-o", obj]
            cmds.append(cmd)
    if cmds:  # (a from-scratch build is thirty translation units, up to a minute and ~1.5 GB each: a few at a time)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)

        jobs = max(1, min(int(os.environ.get("VGAN_BUILD_JOBS", "0")) or min(6, os.cpu_count() or 1), len(cmds)))
        with ThreadPoolExecutor(jobs) as ex:
            list(ex.map(run, cmds))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-lz", "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # the vgan CLI (C++ host driver keeping the reference's subcommand surface)
    main_src = os.path.join(CSRC, "host", "vgan_main.cpp")
    main_srcs = sorted(os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host")) if f.endswith("_main.cpp"))
    if os.path.exists(main_src) and not TAG:
        bindir = os.path.join(HERE, "bin")
        os.makedirs(bindir, exist_ok=True)
        exe = os.path.join(bindir, "vgan")
        if force or _stale(exe, main_srcs + [LIB] + hdrs):
            cmd = [HIPCC] + COMMON + ["-x", "c++"] + main_srcs + ["-x", "none", "-o", exe, "-L" + LIBDIR, "-lvgan_gpu",
                                      "-Wl,-rpath,$ORIGIN/../lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(verbose=True, force="--force" in sys.argv)
    print(LIB)
