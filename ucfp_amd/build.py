"""In-tree build of libucfp_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libucfp_hip.so")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-fno-gpu-rdc", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(SO):
        m = os.path.getmtime(SO)
        if all(os.path.getmtime(d) <= m for d in _deps()):
            return SO
    extra = os.environ.get("UCFP_HIPCC_EXTRA", "").split()   # e.g. -D switches of a tuning experiment
    cmd = [HIPCC] + FLAGS + extra + sources() + ["-o", SO]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return SO


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
