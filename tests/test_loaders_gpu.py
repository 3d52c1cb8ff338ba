"""SURVEY §8f rows N1/N3 on the GPU: sourmash .sig directories and phmap .bin sketch dumps go
straight to the engine; outputs must equal the golden vectors of the reference's own test oracle
and the CPU oracle's TSV for the same sketches."""
import gzip
import os
import shutil

import numpy as np
import pytest

from helpers import GOLDEN, load_golden_lens, load_golden_pairs, load_sig_set, read_pairwise_tsv
from kspider_amd import engine

pytestmark = pytest.mark.gpu


def _names_map(path):
    with open(path) as f:
        n = int(next(f))
        rows = [l.split() for l in f]
    assert len(rows) == n
    return {int(a): b for a, b in rows}


@pytest.mark.parametrize("tag", ["setA", "setB"])
def test_sig_directory_matches_golden_and_oracle(oracle_lib, tmp_path, tag):
    names, sk = load_sig_set(tag)
    golden = load_golden_pairs(tag)
    prefix = str(tmp_path / "out")
    engine.pairwise_sigs(os.path.join(GOLDEN, tag, "sigs"), 25, prefix, 2)
    nm = _names_map(prefix + ".namesMap")
    assert nm == {i + 1: n for i, n in enumerate(names)}                   # glob order -> IDs (sourmash_indexing.cpp:85-117)
    rows = read_pairwise_tsv(prefix + "_kSpider_pairwise.tsv")
    assert {(nm[r[0]], nm[r[1]]): r[2] for r in rows} == {k: v[0] for k, v in golden.items()}
    # identical bytes to the reference-algorithm restatement on the index built from the same sketches
    oprefix = str(tmp_path / "orc")
    oracle_lib.index_from_sketches(oprefix, sk.keys, sk.offsets)
    oracle_lib.ref_pairwise(oprefix, 1)
    assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == open(oprefix + "_kSpider_pairwise.tsv", "rb").read()
    lens = load_golden_lens(tag)
    with open(prefix + "_kSpider_seqToKmersNo.tsv") as f:
        next(f)
        assert {nm[int(l.split("\t")[1])]: int(l.split("\t")[2]) for l in f} == lens


def test_sig_directory_reference_quirks(tmp_path):
    """gzip content inside .sig is read; '.gz' files get an ID but are not read (sourmash_indexing.cpp:152);
    other extensions are ignored; a signature with another ksize is skipped; wrong ksize -> error."""
    src = os.path.join(GOLDEN, "setB", "sigs")
    d = tmp_path / "sigs"
    d.mkdir()
    files = sorted(os.listdir(src))[:12]
    for i, f in enumerate(files):
        data = open(os.path.join(src, f), "rb").read()
        if i % 3 == 0:
            with gzip.open(d / f, "wb") as g:          # gzip content, .sig name
                g.write(data)
        else:
            (d / f).write_bytes(data)
    (d / "zzz_extra.sig.gz").write_bytes(b"not read at all")      # gets ID 13, name "zzz_extra.sig"
    (d / "README.txt").write_text("ignored")
    prefix = str(tmp_path / "o")
    engine.pairwise_sigs(str(d), 25, prefix, 1)
    nm = _names_map(prefix + ".namesMap")
    assert len(nm) == 13 and nm[13] == "zzz_extra.sig"
    with open(prefix + "_kSpider_seqToKmersNo.tsv") as f:
        assert sum(1 for _ in f) == 1 + 12                         # the .gz entry has no k-mer count row
    with pytest.raises(engine.KspError):
        engine.pairwise_sigs(str(d), 31, str(tmp_path / "o2"), 1)  # no signature with ksize 31
    with pytest.raises(engine.KspError):
        engine.pairwise_sigs(str(tmp_path / "missing"), 25, str(tmp_path / "o3"), 1)


@pytest.mark.parametrize("kwidth,trailer", [(16, True), (16, False), (8, True)])
def test_bin_sketch_directory(oracle_lib, tmp_path, kwidth, trailer):
    names, sk = load_sig_set("setA")
    d = tmp_path / "bins"
    d.mkdir()
    rng = np.random.default_rng(3)
    for s, nm in enumerate(names):
        run = sk.run(s).copy()
        rng.shuffle(run)
        oracle_lib.write_bin_sketch(str(d / (nm + ".bin")), run, kwidth=kwidth, trailer=trailer, slot_seed=s)
    (d / "notes.txt").write_text("skipped: not .bin (bins_indexing.cpp:109-112)")
    prefix = str(tmp_path / "o")
    engine.pairwise_bins(str(d), prefix, 2)
    golden = load_golden_pairs("setA")
    nm = _names_map(prefix + ".namesMap")
    rows = read_pairwise_tsv(prefix + "_kSpider_pairwise.tsv")
    assert {(nm[r[0]], nm[r[1]]): r[2] for r in rows} == {k: v[0] for k, v in golden.items()}
