"""The C ABI from plain C: include/ucfp_hip.h must compile as C11 and a gcc-built driver must link
against libucfp_hip.so (what a cgo / Rust `extern "C"` / JNI binding does)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")


def _build(tmp_path):
    from ucfp_amd import _lib
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(_lib.SO_PATH)
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC,
           "-o", exe, "-L", libdir, "-l:libucfp_hip.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_driver_links(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"   # BLAKE3("abc")
    assert lines[1].startswith("ctx_create rc=")


@pytest.mark.gpu
def test_c_driver_matches_oracle(tmp_path, oracle):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = r.stdout.strip().splitlines()
    yy, xx = np.mgrid[0:512, 0:512]
    frame = ((xx + yy) & 255).astype(np.uint8)
    ex = np.frombuffer(bytes.fromhex(lines[0]), np.uint8)[None]
    ref, _ = oracle.image_hash_batch(frame[None], 7, exact=ex)
    assert lines[1] == ref[0].tobytes().hex()
    mh, _ = oracle.text_minhash_batch([b"the quick brown fox jumps over the lazy dog"])
    assert lines[2] == mh[0, :16].tobytes().hex()
    # 0xf1 vs {0xff:3, 0xf0:1, 0x0f:7, 0x00:5} -> ids 20 (d=1), 10 (d=3)
    assert lines[3] == "knn 20:1 10:3 n=2"
