"""The coalescing core of the host micro-batchers (ucfp_amd/csrc/batch_core.h) is plain host code: stress it here,
without a GPU, under ThreadSanitizer.  Every submitter must get ITS result back, no flush may exceed max_batch /
max_units, and the protocol (claim by compare-and-swap, commit, generation futex, release) must be race-free."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "batch_core_stress.cpp")


def _build(tmp_path, *flags):
    exe = str(tmp_path / "batch_core_stress")
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    subprocess.run([cxx, "-std=c++17", *flags, SRC, "-lpthread", "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("threads,per_thread,delay_us", [(24, 300, 0), (40, 60, 200), (3, 500, 0)])
def test_batch_core_under_thread_sanitizer(tmp_path, threads, per_thread, delay_us):
    exe = _build(tmp_path, "-O1", "-g", "-fsanitize=thread")
    r = subprocess.run([exe, str(threads), str(per_thread), str(delay_us)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr


def test_batch_core_many_more_threads_than_slots(tmp_path):
    """200 request threads over sets of 16 slots: most of them wait for room most of the time."""
    exe = _build(tmp_path, "-O2")
    r = subprocess.run([exe, "200", "100", "50"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_rounding_division_magic_numbers(tmp_path):
    """The fused image kernel's floor((2 acc + D) / 2 D) is one multiply-high with a per-frame magic number
    (ucfp_amd/csrc/any_magic.h): exact at every quotient boundary for a grid of geometries and random divisors."""
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "any_magic_check")
    subprocess.run([cxx, "-std=c++17", "-O2", os.path.join(HERE, "native", "any_magic_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
