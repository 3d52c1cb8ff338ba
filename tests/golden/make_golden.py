#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE's own test
oracle, unmodified, in this container.

  /root/reference/test/generate_golden_files.py   -> golden lengths / shared k-mers /
                                                      containments (brute-force sets)
  /root/reference/test/validate.py                -> the reference's validator, run
                                                      against the outputs of our CPU
                                                      restatement (oracle/) as a cross
                                                      check; its report is stored.

Both scripts hard-code the author's paths (generate_golden_files.py:17 globs
/home/mabuelanin/...; validate.py reads files from the CWD), so they are executed
with runpy from a scratch directory, with `glob.glob` redirected to the synthetic
.sig files written below.  Nothing of the reference is copied: only inputs (the
.sig files we synthesise) and outputs (numbers) are stored under tests/golden/.

Two fixture sets:
  setA  every pair of signatures shares >= 1 hash, which generate_golden_files.py
        needs to finish (its containment loop indexes shared_kmers[...] for every
        pair, :72) -> lengths, shared k-mers and containments.
  setB  clustered signatures with many disjoint pairs: the reference script writes
        lengths and shared k-mers (:26-53) and then stops with KeyError at :72, which
        we expect; containments are therefore not available for setB.

Run from the repo root (needs /root/reference, tqdm):  python tests/golden/make_golden.py
"""
import contextlib
import glob as globmod
import io
import json
import os
import pickle
import runpy
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_TEST = "/root/reference/test"

from kspider_amd import synth  # noqa: E402


def write_sigs(sk, out_dir, ksize=25):
    os.makedirs(out_dir, exist_ok=True)
    names = []
    for s in range(sk.n_sources):
        name = f"g{s + 1:03d}"
        names.append(name)
        doc = [{"class": "sourmash_signature", "email": "", "hash_function": "0.murmur64",
                "filename": name + ".fa", "license": "CC0", "version": 0.4,
                "signatures": [{"num": 0, "ksize": ksize, "seed": 42, "max_hash": int((1 << 64) // 1000),
                                "mins": [int(x) for x in sk.run(s)], "md5sum": "0" * 32, "molecule": "dna"}]}]
        with open(os.path.join(out_dir, name + ".sig"), "w") as f:
            json.dump(doc, f, separators=(",", ":"))
    return names


def run_reference_generator(sig_dir, work):
    real_glob = globmod.glob
    globmod.glob = lambda pattern, *a, **k: sorted(real_glob(os.path.join(sig_dir, "*sig")))
    cwd = os.getcwd()
    os.chdir(work)
    err = None
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            runpy.run_path(os.path.join(REF_TEST, "generate_golden_files.py"), run_name="__main__")
    except KeyError as e:  # expected for setB (generate_golden_files.py:72)
        err = e
    finally:
        os.chdir(cwd)
        globmod.glob = real_glob
    return err


def pickles_to_tsv(work, out_dir, have_containments):
    with open(os.path.join(work, "golden_sig_to_len.pickle"), "rb") as f:
        lens = pickle.load(f)
    with open(os.path.join(out_dir, "golden_sig_to_len.tsv"), "w") as f:
        for k in sorted(lens):
            f.write(f"{k}\t{lens[k]}\n")
    with open(os.path.join(work, "golden_pairwise.pickle"), "rb") as f:
        pw = pickle.load(f)
    cont = {}
    if have_containments:
        for nm in ("min", "avg", "max"):
            with open(os.path.join(work, f"golden_{nm}_containments.pickle"), "rb") as f:
                cont[nm] = pickle.load(f)
    with open(os.path.join(out_dir, "golden_pairwise.tsv"), "w") as f:
        f.write("sig1\tsig2\tshared_kmers" + ("\tmin_containment\tavg_containment\tmax_containment" if cont else "") + "\n")
        for (a, b) in sorted(pw):
            if a < b:
                assert pw[(a, b)] == pw[(b, a)]
                row = f"{a}\t{b}\t{pw[(a, b)]}"
                if cont:
                    row += f"\t{cont['min'][(a, b)]}\t{cont['avg'][(a, b)]}\t{cont['max'][(a, b)]}"
                f.write(row + "\n")
    return lens, pw


def run_reference_validator(work, names, sk):
    """Our CPU restatement -> TSVs -> the reference's validate.py (run unmodified)."""
    import oracle
    prefix = os.path.join(work, "sigs")
    oracle.index_from_sketches(prefix, sk.keys, sk.offsets)
    with contextlib.redirect_stdout(io.StringIO()):
        oracle.ref_pairwise(prefix, 2)
    with open(prefix + ".namesMap", "w") as f:   # "<groupID> <groupName>" (src/index.cpp:372-378)
        f.write(f"{len(names)}\n")
        for i, nm in enumerate(names):
            f.write(f"{i + 1} {nm}\n")
    out, errb = io.StringIO(), io.StringIO()
    cwd = os.getcwd()
    os.chdir(work)
    try:
        with contextlib.redirect_stdout(out), contextlib.redirect_stderr(errb):
            try:
                runpy.run_path(os.path.join(REF_TEST, "validate.py"), run_name="__main__")
            except SystemExit as e:
                errb.write(f"SystemExit({e.code})\n")
    finally:
        os.chdir(cwd)
    return out.getvalue(), errb.getvalue()


def make_set(tag, sk, have_containments):
    out_dir = os.path.join(HERE, tag)
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    names = write_sigs(sk, os.path.join(out_dir, "sigs"))
    work = tempfile.mkdtemp(prefix="kspider_golden_")
    err = run_reference_generator(os.path.join(out_dir, "sigs"), work)
    if have_containments and err is not None:
        raise RuntimeError(f"reference generator failed on {tag}: {err!r}")
    if not have_containments and err is None:
        raise RuntimeError(f"{tag}: expected the reference generator to stop at :72")
    lens, pw = pickles_to_tsv(work, out_dir, have_containments)
    report = f"# {tag}: {len(lens)} signatures, {sum(1 for k in pw if k[0] < k[1])} non-zero pairs\n"
    if have_containments:
        so, se = run_reference_validator(work, names, sk)
        report += "# stdout of /root/reference/test/validate.py on the oracle's TSVs\n" + so
        report += "# stderr (containment lines: validate.py compares a 5-char *truncation* with a\n"
        report += "# 3-decimal *rounding*, :76-78 vs generate_golden_files.py:80-82, so mismatches there are expected)\n" + se
    with open(os.path.join(out_dir, "reference_report.txt"), "w") as f:
        f.write(report)
    shutil.rmtree(work, ignore_errors=True)
    print(report)


def main():
    # setA: 36 signatures, 3 clusters, one hash common to all
    a = synth.generate("C2", n_sources=36, mean_size=90, cluster_cap=14, seed=777)
    runs = [np.concatenate([a.run(s), np.array([4242424242], dtype=np.uint64)]) for s in range(a.n_sources)]
    a = synth.from_runs(runs, "setA")
    make_set("setA", a, True)
    # setB: 40 signatures, small clusters, most pairs disjoint
    b = synth.generate("C2", n_sources=40, mean_size=70, cluster_cap=6, seed=778)
    make_set("setB", b, False)


if __name__ == "__main__":
    main()
