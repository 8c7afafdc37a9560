"""Sidecar of a committed rocprofv3 kernel-trace summary of the bench command: which arguments the traced command had and the git
blob hash of every kernel source at that time.  bench.py (rocprof_avg_us) uses the summary as a cross-check of its live HIP-event
timing only when both still hold.
    python tools/trace_meta.py profiles/r06_bench_kernel_stats.txt [--arch dla_34 --batch 64 --size 512 --dtype bf16 --dcn-offsets init]"""
import argparse
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blob_sha(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


ap = argparse.ArgumentParser()
ap.add_argument("stats")
ap.add_argument("--arch", default="dla_34")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--dcn-offsets", default="init")
a = ap.parse_args()
stats = os.path.relpath(os.path.abspath(a.stats), ROOT)
srcs = sorted(glob.glob(os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc", "*.hip")) +
              glob.glob(os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc", "*.h")) + [os.path.join(ROOT, "include", "centernet_hip.h")])
meta = {"stats_file": stats,
        "cmd_args": {"arch": a.arch, "batch": a.batch, "size": a.size, "dtype": a.dtype, "dcn_offsets": a.dcn_offsets},
        "sources": {os.path.relpath(p, ROOT): blob_sha(p) for p in srcs}}
out = os.path.join(ROOT, os.path.dirname(stats), os.path.basename(stats).replace("_kernel_stats.txt", "_trace_meta.json"))
json.dump(meta, open(out, "w"), indent=1)
print("wrote", os.path.relpath(out, ROOT), f"({len(meta['sources'])} sources)")
