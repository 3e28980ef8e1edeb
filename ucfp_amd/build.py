"""In-tree build of libucfp_hip.so for gfx950 (hipcc cross-compiles without a GPU).

Each translation unit is compiled to its own object (only when it or a header changed, in parallel), then linked."""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
SO = os.path.join(HERE, "libucfp_hip.so")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=off", "-Wall",
          "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-ldl", "-lpthread"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))


def build_hip(force: bool = False, verbose: bool = False) -> str:
    extra = os.environ.get("UCFP_HIPCC_EXTRA", "").split()   # e.g. -D switches of a tuning experiment
    force = force or bool(extra)
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = max(os.path.getmtime(h) for h in _headers())
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC] + CFLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(obj):
                os.remove(obj)
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n" + r.stdout + r.stderr)
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for warn in ex.map(compile_one, jobs):
                if verbose and warn.strip():
                    print(warn)
    if jobs or not os.path.exists(SO) or any(os.path.getmtime(o) > os.path.getmtime(SO) for o in objs):
        cmd = [HIPCC] + LDFLAGS + objs + ["-o", SO]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    return SO


if __name__ == "__main__":
    import sys
    print(build_hip(force="--force" in sys.argv, verbose=True))
