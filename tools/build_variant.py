"""Build a VARIANT of libmicroaligner_hip.so beside the tree's own (for A/B timing on one box, tools/ab_libs.sh):

    python3 tools/build_variant.py <tag> [extra hipcc flags ...] [--src file.hip=path/to/replacement.hip ...]

writes variants/libma_<tag>.so (git-ignored, travels with gpurun).  Objects live in microaligner_amd/build_<tag>/.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microaligner_amd import build as B   # noqa: E402


def main():
    tag, rest = sys.argv[1], sys.argv[2:]
    repl, extra = {}, []
    it = iter(rest)
    for a in it:
        if a == "--src":
            k, v = next(it).split("=", 1)
            repl[k] = os.path.abspath(v)
        else:
            extra.append(a)
    objdir = os.path.join(ROOT, "microaligner_amd", f"build_{tag}")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    with open(os.path.join(objdir, "ma_src_hash.h"), "w") as f:
        f.write(f'#define MA_SRC_HASH "variant-{tag}"\n')
    hipcc = B._hipcc()
    objs, jobs = [], []
    for s in B.SOURCES:
        src = repl.get(s, os.path.join(B.CSRC, s))
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        jobs.append([hipcc] + B.FLAGS + extra + ["-I", objdir, "-I", B.CSRC, "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(" ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    lib = os.path.join(ROOT, "variants", f"libma_{tag}.so")
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
