#!/usr/bin/env python3
"""The mixed-upload leg of bench.py alone: python tools/bench_uploads.py [n_images] [request_threads]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ucfp_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    th = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    print(json.dumps(bench.bench_upload_mix(torch.device("cuda", 0), _lib.default_context(0), n, th)), flush=True)
