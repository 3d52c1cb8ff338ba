#!/usr/bin/env python3
"""End-to-end drop-in comparison on index files: kSpider::pairwise() through libkspider_amd.so vs the CPU
restatement of the reference (oracle), same index, byte-compared TSVs.   python tests/dropin_compare.py [N]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from kspider_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
threads = len(os.sched_getaffinity(0))
sk = synth.generate("C2", n_sources=n)
d = tempfile.mkdtemp(prefix="ksp_dropin_")
prefix = os.path.join(d, "idx")
t = time.time()
co, src, w = oracle.index_from_sketches(prefix, sk.keys, sk.offsets)
print(f"index: {n} sources, {len(w)} colours, {len(src)} colour-source entries, written in {time.time() - t:.1f} s", flush=True)
sizes = {f: os.path.getsize(prefix + f) for f in ("_color_to_sources.bin", "_color_count.bin", "_groupID_to_kmerCount.bin")}
print("files:", sizes, flush=True)
for rep in range(2):
    t = time.time()
    out = subprocess.run([os.path.join(os.path.dirname(engine.LIB_PATH), "pairwise"), prefix, str(threads)],
                         capture_output=True, text=True, check=True)
    t_gpu = time.time() - t
print(out.stdout.strip())
print(f"kspider_amd pairwise exe, wall {t_gpu:.2f} s ({threads} host threads for the TSV)", flush=True)
got = open(prefix + "_kSpider_pairwise.tsv", "rb").read()
t = time.time()
secs, ne, nu = oracle.ref_pairwise(prefix, threads)
t_cpu = time.time() - t
want = open(prefix + "_kSpider_pairwise.tsv", "rb").read()
print(f"reference restatement: accumulate {secs:.2f} s ({threads} threads), wall {t_cpu:.2f} s, {ne} rows")
print("TSV identical:", got == want, f"({len(got)} bytes)")
