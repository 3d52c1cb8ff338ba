"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/kspider_amd.h declares, host logic (index reader, float text), and loud
failure without a GPU.  No compute calls."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from kspider_amd import engine, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "kspider_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ksp_[a-z0-9_]+|kspider_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = engine.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(engine.ABI_SYMBOLS) == declared


def test_cxx_symbol_of_reference_signature_is_exported():
    """kSpider::pairwise(std::string, int) with the Itanium name the reference's callers link against."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", engine.LIB_PATH], text=True)
    assert "_ZN7kSpider8pairwiseENSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEEi" in out


def test_struct_layouts():
    assert engine.EDGE_DTYPE.itemsize == 16
    assert ctypes.sizeof(engine.Stats) == 9 * 8 + 4 * 4 + 2 * 8 + 8 + 4 * 4 + 4 * 8 + 2 * 4  # (= sizeof(ksp_stats), include/kspider_amd.h)


def test_missing_index_fails_loudly(tmp_path):
    with pytest.raises(engine.KspError) as ei:
        engine.pairwise(str(tmp_path / "nope"), 1)
    assert ei.value.code == engine.KSP_E_IO and "cannot open" in str(ei.value)
    assert not os.path.exists(str(tmp_path / "nope_kSpider_pairwise.tsv"))


def test_python_module_has_reference_surface(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "kspider_amd", "lib"))
    try:
        import _kSpider_internal as ks   # name used by ks_pairwise.py:5
    finally:
        sys.path.pop(0)
    with pytest.raises(RuntimeError):
        ks.pairwise(index_prefix=str(tmp_path / "nope"), user_threads=1)   # kwargs: test/kspider_run.py:4
    with pytest.raises(TypeError):
        ks.pairwise(1, 2)
    with pytest.raises(NotImplementedError):
        ks.sourmash_sigs_indexing(sigs_dir="x", kSize=25)


@pytest.mark.parametrize("kwidth,trailer", [(16, True), (16, False), (8, True), (8, False)])
def test_index_reader_detects_every_dump_layout(oracle_lib, tmp_path, kwidth, trailer):
    sk = synth.generate("C2", n_sources=70, mean_size=120, cluster_cap=9, seed=31)
    prefix = str(tmp_path / "ix")
    co, src, w = oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets, kwidth=kwidth, trailer=trailer)
    info = engine.index_info(prefix)
    assert info["colors"] == len(w) and info["groups"] == 70 and info["sources"] == len(src)
    # (16, no trailer) and (8, trailer) have the same byte length; the cloned control bytes decide
    assert (info["kwidth"], info["trailer"]) == (kwidth, trailer)


def test_index_reader_rejects_truncated_and_corrupt_files(oracle_lib, tmp_path):
    sk = synth.generate("C2", n_sources=40, mean_size=80, cluster_cap=5, seed=32)
    prefix = str(tmp_path / "ix")
    oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets)
    path = prefix + "_color_to_sources.bin"
    blob = open(path, "rb").read()
    open(path, "wb").write(blob[:-5])
    with pytest.raises(engine.KspError):
        engine.index_info(prefix)
    # flip a bit in the `size` word of the first nested table (offset 16): ctrl bytes no longer agree
    open(path, "wb").write(blob[:16] + bytes([blob[16] ^ 0x01]) + blob[17:])
    with pytest.raises(engine.KspError):
        engine.index_info(prefix)
    open(path, "wb").write(blob)
    engine.index_info(prefix)


def test_float_text_matches_ostream_default():
    """`ostream << float` == printf("%g") with precision 6 (src/pairwise.cpp:266-273)."""
    rng = np.random.default_rng(5)
    vals = list(rng.random(2000).astype(np.float32)) + [np.float32(x) for x in
            (1.0, 0.5, 1 / 3, 2.5e-5, 1e-7, 0.999999, 0.9999995, 123456.7, 1234567.0, 0.0)]
    for v in vals:
        assert engine.format_float(float(v)) == "%g" % float(v)
    assert engine.format_float(float("inf")) == "inf"


def test_no_gpu_means_error_not_fallback():
    """There is no CPU path behind the C ABI: without a device the engine refuses to exist."""
    try:
        n = engine.device_count()
    except engine.KspError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(engine.KspError):
        engine.Engine(0)
    sk = synth.from_runs([[1, 2], [2, 3]])
    with pytest.raises(engine.KspError):
        engine.pairwise_host(sk.keys, sk.offsets)


def test_synth_is_deterministic():
    a = synth.generate("C2", n_sources=64, mean_size=100, seed=3)
    b = synth.generate("C2", n_sources=64, mean_size=100, seed=3)
    assert (a.keys == b.keys).all() and (a.offsets == b.offsets).all()
    for s in range(a.n_sources):
        r = a.run(s)
        assert (np.diff(r.astype(np.int64) if r.size and r.max() < 2**62 else r.astype(np.float64)) > 0).all()


def test_sig_loader_fails_loudly_without_inputs(tmp_path):
    with pytest.raises(engine.KspError) as ei:
        engine.pairwise_sigs(str(tmp_path), 31, str(tmp_path / "o"), 1)
    assert ei.value.code == engine.KSP_E_IO
    (tmp_path / "bad.sig").write_text('[{"signatures": [{"ksize": 31, "mins": [1, 2, }]}]')
    with pytest.raises(engine.KspError) as ei:
        engine.pairwise_sigs(str(tmp_path), 31, str(tmp_path / "o"), 1)
    assert "malformed JSON" in str(ei.value)
    with pytest.raises(engine.KspError):
        engine.pairwise_bins(str(tmp_path), str(tmp_path / "o"), 1)
