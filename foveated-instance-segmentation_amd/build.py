"""Compile csrc/*.hip into csrc/libfovealseg_hip.so for gfx950 (hipcc cross-compiles without a GPU).

The library is stamped with a content hash of everything it was built from (every csrc/*.hip and *.h, the compiler flags):
`libfovealseg_hip.so.stamp` next to it.  `needs_build()` compares that stamp with the sources as they are now -- not mtimes -- and
`hip.load()` refuses a library whose stamp does not match the sources beside it, so a stale binary can neither ship nor load.
Objects are cached per translation unit by the same kind of hash, and compiled in parallel.

FS_BUILD_EXPERIMENTS=1 python build.py  ->  ab/libfovealseg_experiments.so with -DFS_EXPERIMENTS (the kernel A/B switches of
common.h: FS_ENV_INT); loaded through FS_HIP_LIB.  The shipped library has no such switches.
"""
import concurrent.futures
import glob
import hashlib
import json
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(CSRC, "libfovealseg_hip.so")
STAMP = LIB + ".stamp"
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the fovealseg HIP library cannot be built")


def _sha(paths, extra=()):
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))), sorted(glob.glob(os.path.join(CSRC, "*.h")))


def source_hash(flags=FLAGS) -> str:
    """Content hash of every source and header of the library plus the compiler flags."""
    hips, hdrs = sources()
    return _sha(hips + hdrs, extra=flags)


def built_hash(lib=LIB):
    try:
        with open(lib + ".stamp") as f:
            return json.load(f)["source_hash"]
    except (OSError, ValueError, KeyError):
        return None


def needs_build() -> bool:
    return not os.path.exists(LIB) or built_hash() != source_hash()


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    flags = FLAGS + (["-DFS_EXPERIMENTS"] + os.environ.get("FS_BUILD_DEFINES", "").split() if experiments else [])
    lib = LIB
    objdir = CSRC
    if experiments:
        objdir = os.path.join(os.path.dirname(PKG), "ab", "obj_experiments")
        os.makedirs(objdir, exist_ok=True)
        lib = os.path.join(os.path.dirname(PKG), "ab", "libfovealseg_experiments.so")
    want = source_hash(flags)
    if not force and os.path.exists(lib) and built_hash(lib) == want:
        return lib
    hipcc = _hipcc()
    hips, hdrs = sources()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        tag = _sha([src] + hdrs, extra=flags)
        tagfile = obj + ".stamp"
        try:
            if not force and os.path.exists(obj) and open(tagfile).read() == tag:
                return obj
        except OSError:
            pass
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(tagfile, "w") as f:
            f.write(tag)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, hips))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(lib + ".stamp", "w") as f:
        json.dump({"source_hash": want, "flags": flags, "sources": [os.path.basename(p) for p in hips + hdrs]}, f)
    return lib


if __name__ == "__main__":
    print(build(force=os.environ.get("FS_BUILD_FORCE", "0") == "1", verbose=True, experiments=os.environ.get("FS_BUILD_EXPERIMENTS", "0") == "1"))
