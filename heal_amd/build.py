"""Build libheal_amd.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m heal_amd.build [--force] [--verbose]
    HEAL_BUILD_EXPERIMENTAL=1 python -m heal_amd.build     # + the measured-negative kernels (include/heal_amd_experimental.h)

Every csrc/*.hip is compiled to an object (in parallel; objects are cached per source + headers + flags, so an edit recompiles
one file) and linked into heal_amd/lib/libheal_amd.so.  csrc/experimental/*.hip -- kernels that are correct but lose against the
production path at every BASELINE shape -- are compiled only on request: the shipped library and its ABI
carry what runs.
The library is built with -ffp-contract=off: kernels that must reproduce the reference's fp32
arithmetic bit for bit (voxel indices, box decode) rely on it; FMAs are written explicitly
(fmaf) where they are wanted.
"""
import argparse
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libheal_amd.so")
OBJDIR = os.path.join(HERE, "build")
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
          f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(os.path.dirname(HERE), "include")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


EXPERIMENTAL = os.environ.get("HEAL_BUILD_EXPERIMENTAL", "0") == "1"
EXP_DIR = os.path.join(CSRC, "experimental")


def _flags():
    return CFLAGS + (["-DHEAL_BUILD_EXPERIMENTAL=1"] if EXPERIMENTAL else [])


def _sources():
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    if EXPERIMENTAL and os.path.isdir(EXP_DIR):
        srcs += sorted(os.path.join(EXP_DIR, f) for f in os.listdir(EXP_DIR) if f.endswith(".hip"))
    return srcs


def _stamp(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.relpath(p, HERE).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(f for f in _flags() if not f.startswith(os.sep)).encode())
    return h.hexdigest()


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(os.path.dirname(HERE), "include")
    hdrs.append(os.path.join(inc, "heal_amd.h"))
    if EXPERIMENTAL:
        hdrs.append(os.path.join(inc, "heal_amd_experimental.h"))
    return hdrs


def build(force=False, verbose=False):
    """Compile (if stale) and return the path of libheal_amd.so."""
    srcs = _sources()
    stamp = _stamp(srcs + _deps())
    stamp_file = os.path.join(LIBDIR, "libheal_amd.stamp")
    if (not force and os.path.exists(LIB) and os.path.exists(stamp_file)
            and open(stamp_file).read().strip() == stamp):
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)

    deps = _deps()

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        key = _stamp([src] + deps)                       # this object is current if its source, the headers and the flags are
        keyfile = obj + ".stamp"
        if (not force and os.path.exists(obj) and os.path.exists(keyfile) and open(keyfile).read().strip() == key):
            return obj
        cmd = [hipcc, "-c", src, "-o", obj] + _flags()
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(keyfile, "w") as f:
            f.write(key)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, verbose=a.verbose))
