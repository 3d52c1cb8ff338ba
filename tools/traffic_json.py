"""profiles/traffic.json from one tools/prof_bench.sh run: PMC HBM bytes per step and per kernel group.

FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE counts the 128-byte requests of a wide
streaming read as 64 bytes (MI355X_MICROARCH.md, HBM section), so fetch is doubled: exact for 16-byte-per-lane
streams, an upper estimate for narrower accesses (stated per group).  Kernels are mapped to the phase names
bench.py reports (ksp_engine_phase_times)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "kspider_amd", "csrc", "*.hip*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


GROUPS = [  # (substring of the kernel name, group)
    ("k_max_last", "key range + source sizes"), ("k_src_size", "key range + source sizes"), ("k_iota4", "key range + source sizes"),
    ("k_tag", "tags + source sizes"),
    ("k_zero_regions", "key range + source sizes"),
    ("k_seg_", "partition"), ("k_part", "partition"), ("k_hist2", "partition"), ("k_scan2", "partition"), ("k_scatter2", "partition"),
    ("radix_sort_onesweep", "partition"), ("onesweep_histograms", "partition"),
    ("k_match", "match records"),
    ("k_bucket", "bucket grouping"),
    ("k_label", "source labels + order"), ("k_perm", "source labels + order"), ("k_blk_bound", "source labels + order"),
    ("k_pack_blocks", "source labels + order"), ("k_place_sources", "source labels + order"),
    ("k_key_groups", "key groups"), ("k_group_totals", "key groups"),
    ("k_move_groups", "block lists"), ("k_blk_raw", "block lists"), ("k_blk_pos", "block lists"), ("k_pad", "block lists"),
    ("k_place_groups", "block lists"), ("k_cidx", "block lists"), ("k_ms_", "block lists"),
    ("k_tile_flags", "work list"), ("k_pack_flags", "work list"), ("k_list_pairs", "work list"), ("k_copy_regions", "work list"),
    ("k_join", "join"),
]


def group_of(name):
    for sub, g in GROUPS:
        if sub in name:
            return g
    return "other (scans, small sorts, fills, copies)"


def per_kernel(counter_dir, counter):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, counter_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"])
                disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return tot, disp


fetch, fd = per_kernel("pmc_fetch", "FETCH_SIZE")
write, wd = per_kernel("pmc_write", "WRITE_SIZE")
# steps of the profiled command = dispatches of the join kernel
joins = max([len(v) for k, v in fd.items() if "k_join" in k] or [1])
groups = collections.defaultdict(lambda: {"fetch_kb_raw": 0.0, "write_kb": 0.0})
for k, v in fetch.items():
    groups[group_of(k)]["fetch_kb_raw"] += v / joins
for k, v in write.items():
    groups[group_of(k)]["write_kb"] += v / joins
out = {"kernel_source_hash": kernel_source_hash(), "steps_profiled": joins,
       "source": "tools/prof_bench.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over "
                 "`python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --other-configs= --profile-steps 0`",
       "correction": "gfx950: FETCH_SIZE x 2 (128-byte requests counted as 64 bytes; exact for 16-byte-per-lane streams, an "
                     "upper estimate otherwise); WRITE_SIZE as read", "groups": {}}
for g, v in groups.items():
    out["groups"][g] = {"fetch_kb_raw_per_step": round(v["fetch_kb_raw"], 1), "write_kb_per_step": round(v["write_kb"], 1),
                        "hbm_bytes_per_step": int((2 * v["fetch_kb_raw"] + v["write_kb"]) * 1024)}
print(json.dumps(out, indent=1))
