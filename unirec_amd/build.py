"""Build the HIP C-ABI library in-tree:  python -m unirec_amd.build  ->  unirec_amd/libunirec_amd.so

hipcc cross-compiles for gfx950 without a GPU.  Object files are cached in unirec_amd/csrc/_obj and
rebuilt only when a source or header changed.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libunirec_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


# the bounds-checked variant (common.h: UR_ROW): the same sources with -DUR_DEBUG_BOUNDS -> libunirec_amd_dbg.so, loaded by _lib.py when
# UR_DEBUG_BOUNDS=1.  Only the files that use the macro are recompiled; every other object is shared with the release library.
DBG_LIB = os.path.join(HERE, "libunirec_amd_dbg.so")
DBG_FILES = ("rowops.hip", "loss.hip", "rowchain.hip")


def _digest(paths, extra=()):
    h = hashlib.sha256(" ".join(FLAGS + list(extra)).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(verbose=True, force=False, debug_bounds=False):
    os.makedirs(OBJ, exist_ok=True)
    lib_path = DBG_LIB if debug_bounds else LIB
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "unirec_amd.h"))
    jobs, objs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        dbg = ["-DUR_DEBUG_BOUNDS"] if (debug_bounds and s in DBG_FILES) else []
        stem = os.path.splitext(s)[0] + ("_dbg" if dbg else "")
        tag = _digest([src] + hdrs, dbg)
        obj = os.path.join(OBJ, f"{stem}.{tag}.o")
        objs.append(obj)
        if force or not os.path.exists(obj):
            for old in os.listdir(OBJ):
                if old.startswith(stem + "."):
                    os.remove(os.path.join(OBJ, old))
            lang = ["-x", "hip"] if s.endswith(".hip") else []
            jobs.append((s, [HIPCC] + FLAGS + dbg + lang + ["-c", src, "-o", obj]))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return name, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name, rc, out in ex.map(run, jobs):
                if verbose and out.strip():
                    print(f"[{name}]\n{out}")
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {name}:\n{out}")
    # relink whenever the library on disk was not linked from exactly these objects (e.g. a .so copied in from another build)
    stamp = os.path.join(OBJ, "link_dbg.stamp" if debug_bounds else "link.stamp")
    want = hashlib.sha256(" ".join(os.path.basename(o) for o in objs).encode()).hexdigest()
    have = None
    if os.path.exists(stamp) and os.path.exists(lib_path):
        with open(stamp) as f:
            tag, _, mtime = f.read().partition(" ")
        if mtime == repr(os.path.getmtime(lib_path)):
            have = tag
    if jobs or force or have != want:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        with open(stamp, "w") as f:
            f.write(want + " " + repr(os.path.getmtime(lib_path)))
    if verbose:
        print(f"built {lib_path} ({len(jobs)} objects recompiled)")
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, debug_bounds="--debug-bounds" in sys.argv)
