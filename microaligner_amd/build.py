"""Build recipe for libmicroaligner_hip.so (hipcc, gfx950 only, in-tree).

    python -m microaligner_amd.build [--force]

-ffp-contract=off is load-bearing: the kernels follow OpenCV's operation order
(multiply and add as separate roundings) and must not be fused by the compiler;
fused multiply-adds are written explicitly where a mode asks for them.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmicroaligner_hip.so")
SOURCES = ["ma_api.hip", "farneback.hip", "remap.hip", "pyramid.hip", "dog.hip", "nmi.hip", "affine.hip", "knn.hip", "daisy.hip", "ransac.hip", "feature_round.hip", "register.hip", "probe.hip"]
HEADERS = [os.path.join(CSRC, "ma_internal.h"), os.path.join(HERE, "..", "include", "microaligner_hip.h")]
# -fno-slp-vectorize: the SLP vectoriser packs the sliding-window blur into v_pk_* ops with a storm of
# register-pair shuffles (measured 1.65x slower on blur_h_solve, profiles/r01_*); packed math is written by hand
# where it pays.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _flags():
    # MA_HIPCC_EXTRA: extra compiler flags for experiments (e.g. "-fno-slp-vectorize")
    return FLAGS + os.environ.get("MA_HIPCC_EXTRA", "").split()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash():
    """sha256 (first 16 hex digits) over every source file that holds kernels of the measured path, the internal header,
    the public header and the compiler flags: ma_version() carries it, the profile summaries under profiles/ record it, and bench.py only quotes PMC
    traffic from a summary whose hash is the loaded library's."""
    import hashlib
    h = hashlib.sha256()
    # register.hip decides which launches the measured path makes and on which stream, ma_api.hip how the buffer cache and
    # the streams behave: both are part of what a profile measures.  Left out: the clock probe and the feature stage
    # (FAST / DAISY / 2-NN / affine warp), which has its own tests and timings and no kernel on the measured path (cfg3:
    # pyramid, DOG, Farneback, warp, merge, NMI)
    off_path = {"probe.hip", "knn.hip", "daisy.hip", "ransac.hip", "feature_round.hip", "affine.hip"}
    for path in [os.path.join(CSRC, s) for s in SOURCES if s not in off_path] + HEADERS:
        h.update(open(path, "rb").read())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library in-tree."""
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    # the source hash reaches ma_version() through a generated header that is rewritten only when it changes
    hash_h = os.path.join(objdir, "ma_src_hash.h")
    text = f'#define MA_SRC_HASH "{source_hash()}"\n'
    if not os.path.exists(hash_h) or open(hash_h).read() != text:
        with open(hash_h, "w") as f:
            f.write(text)
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        deps = [src] + HEADERS + ([hash_h] if s == "ma_api.hip" else [])
        if force or _stale(obj, deps):
            jobs.append([hipcc] + _flags() + ["-I", objdir, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for err in ex.map(run, jobs):
            if verbose and err:
                print(err)
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
