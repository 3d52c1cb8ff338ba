"""Host side of the drop-in under AddressSanitizer + UBSan (CPU build, `make -C kspider_amd/csrc asan`): the
parsers of untrusted files — phmap dumps, sourmash JSON / gzip, .bin sketches — are fed a malformed corpus.
Every case must come back as KSP_E_IO with a message, with no sanitizer report and no pairwise TSV (complete
or partial) left behind.  The reference reads garbage from a failed ifstream here (src/pairwise.cpp:97-101)
and asserts on empty maps (:117, :170)."""
import gzip
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from kspider_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kspider_amd", "lib", "host_asan_check")
KSP_E_HIP, KSP_E_IO = 2, 3
FILES = ("_color_to_sources.bin", "_color_count.bin", "_groupID_to_kmerCount.bin")


@pytest.fixture(scope="session")
def exe():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "kspider_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)
    return EXE


def _run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1:max_allocation_size_mb=4096",
               UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, *args], capture_output=True, text=True, env=env, timeout=120)
    report = p.stderr
    assert "AddressSanitizer" not in report and "runtime error" not in report and "LeakSanitizer" not in report, report[-3000:]
    assert p.returncode == 0, (p.returncode, report[-2000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("rc ")][-1]
    rc = int(line.split()[1])
    return rc, line, p.stdout


@pytest.fixture(scope="module")
def good_index(oracle_lib, tmp_path_factory):
    d = tmp_path_factory.mktemp("ix")
    sk = synth.generate("C2", n_sources=60, mean_size=80, cluster_cap=8, seed=77)
    prefix = str(d / "good")
    oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets)
    return prefix, {f: open(prefix + f, "rb").read() for f in FILES}


def _write(prefix, blobs):
    for f, b in blobs.items():
        with open(prefix + f, "wb") as fh:
            fh.write(b)


def test_valid_index_parses_clean_and_stops_at_the_missing_engine(exe, good_index, tmp_path):
    prefix, blobs = good_index
    rc, line, out = _run(exe, "info", prefix)
    assert rc == 0 and "info " in out
    p2 = str(tmp_path / "copy")
    _write(p2, blobs)
    rc, line, _ = _run(exe, "index", p2)          # loader fine, then: no HIP engine in this build, no CPU fallback
    assert rc == KSP_E_HIP, line
    assert not os.path.exists(p2 + "_kSpider_pairwise.tsv") and not os.path.exists(p2 + "_kSpider_pairwise.tsv.partial")


def _mutations(blobs):
    src, cnt, grp = (blobs[f] for f in FILES)
    cases = {}
    for name, f, data in (("sources", FILES[0], src), ("count", FILES[1], cnt), ("groups", FILES[2], grp)):
        for cut in (0, 7, 15, 17, len(data) // 3, len(data) // 2, len(data) - 9, len(data) - 1):
            cases[f"truncated_{name}_{cut}"] = {f: data[:cut]}
        cases[f"trailing_garbage_{name}"] = {f: data + b"\x00" * 13}
    # single-table files: header lies
    def hdr(data, size=None, cap=None):
        s, c = struct.unpack_from("<QQ", data, 0)
        return struct.pack("<QQ", s if size is None else size, c if cap is None else cap) + data[16:]
    s0, c0 = struct.unpack_from("<QQ", grp, 0)
    cases["capacity_not_pow2_minus_1"] = {FILES[2]: hdr(grp, cap=c0 - 1)}
    cases["size_above_capacity"] = {FILES[2]: hdr(grp, size=c0 + 5)}
    cases["size_absurd"] = {FILES[2]: hdr(grp, size=1 << 63)}
    cases["capacity_absurd"] = {FILES[2]: hdr(grp, cap=(1 << 62) - 1)}
    cases["capacity_huge_but_plausible"] = {FILES[1]: hdr(cnt, cap=(1 << 39) - 1)}
    cases["size_one_less"] = {FILES[2]: hdr(grp, size=s0 - 1)}
    cases["empty_groups_table"] = {FILES[2]: struct.pack("<QQ", 0, 0)}
    bad_sentinel = bytearray(grp); bad_sentinel[16 + c0] = 0x05
    cases["sentinel_overwritten"] = {FILES[2]: bytes(bad_sentinel)}
    # nested file: lying colour counts, broken inner tables
    C = struct.unpack_from("<Q", src, 0)[0]
    for name, v in (("plus1", C + 1), ("minus1", C - 1), ("huge", 1 << 60), ("max", (1 << 64) - 1), ("zero", 0)):
        cases[f"colour_count_{name}"] = {FILES[0]: struct.pack("<Q", v) + src[8:]}
    inner = bytearray(src)
    struct.pack_into("<Q", inner, 8 + 8, 1 << 50)          # size of the first inner table
    cases["inner_size_absurd"] = {FILES[0]: bytes(inner)}
    inner = bytearray(src)
    struct.pack_into("<Q", inner, 8 + 16, (1 << 33) - 1)   # capacity of the first inner table
    cases["inner_capacity_lies"] = {FILES[0]: bytes(inner)}
    rng = np.random.default_rng(3)
    for i in range(12):                                    # random byte damage in the structural parts
        b = bytearray(src)
        for pos in rng.integers(0, min(len(b), 4096), size=6):
            b[int(pos)] ^= int(rng.integers(1, 256))
        cases[f"random_damage_{i}"] = {FILES[0]: bytes(b)}
    return cases


def test_malformed_index_corpus(exe, good_index, tmp_path):
    _, blobs = good_index
    cases = _mutations(blobs)
    assert len(cases) > 40
    accepted = []
    for name, repl in cases.items():
        prefix = str(tmp_path / name)
        _write(prefix, {**blobs, **repl})
        rc, line, _ = _run(exe, "index", prefix)
        if rc == KSP_E_HIP:
            # random damage may land in bytes no table reads (unused slots): then the files are still a valid
            # index and the run ends at the missing engine — allowed only for the random cases
            assert name.startswith("random_damage"), (name, line)
            accepted.append(name)
        else:
            assert rc == KSP_E_IO, (name, line)
            assert len(line.split(None, 2)) == 3, (name, line)          # a message names the problem
        assert not os.path.exists(prefix + "_kSpider_pairwise.tsv"), name
        assert not os.path.exists(prefix + "_kSpider_pairwise.tsv.partial"), name
    assert len(accepted) < 12
    # a missing file
    prefix = str(tmp_path / "missing")
    _write(prefix, {FILES[0]: blobs[FILES[0]], FILES[1]: blobs[FILES[1]]})
    rc, line, _ = _run(exe, "index", prefix)
    assert rc == KSP_E_IO and "cannot open" in line


def _sig(mins, ksize=31):
    return json.dumps([{"class": "sourmash_signature", "signatures": [{"ksize": ksize, "mins": mins}]}]).encode()


def test_malformed_signature_and_bin_corpus(exe, oracle_lib, tmp_path):
    good = _sig([5, 7, 11, 13])
    gz = gzip.compress(_sig([5, 7, 99]))
    cases = {
        "unterminated_json": good[:-7],
        "unterminated_string": b'[{"signatures": [{"ksize": 31, "mins": [1, 2], "name": "abc',
        "bad_escape_at_eof": b'[{"name": "abc\\u12',
        "not_json": b"\x00\x01\x02\xff" * 50,
        "empty_file": b"",
        "top_level_object": b'{"signatures": []}',
        "mins_not_numbers": b'[{"signatures": [{"ksize": 31, "mins": ["a", "b"]}]}]',
        "hash_out_of_range": b'[{"signatures": [{"ksize": 31, "mins": [99999999999999999999999]}]}]',
        "no_mins": b'[{"signatures": [{"ksize": 31}]}]',
        "deep_nesting": b'[{"x": ' + b"[" * 200000 + b"]" * 200000 + b"}]",
        "truncated_gzip": gz[: len(gz) - 6],
        "gzip_with_damaged_body": gz[:12] + bytes(b ^ 0x5A for b in gz[12:20]) + gz[20:],
    }
    for name, data in cases.items():
        d = tmp_path / name
        d.mkdir()
        (d / "a.sig").write_bytes(good)
        (d / "b.sig").write_bytes(data)
        out = str(tmp_path / (name + "_out"))
        rc, line, _ = _run(exe, "sigs", str(d), "31", out)
        assert rc == KSP_E_IO, (name, line)
        assert not os.path.exists(out + "_kSpider_pairwise.tsv") and not os.path.exists(out + "_kSpider_pairwise.tsv.partial")
    # well-formed signatures reach the (absent) engine
    d = tmp_path / "fine"
    d.mkdir()
    (d / "a.sig").write_bytes(good)
    (d / "b.sig").write_bytes(gz)
    rc, line, _ = _run(exe, "sigs", str(d), "31", str(tmp_path / "fine_out"))
    assert rc == KSP_E_HIP, line
    # .bin sketches
    d = tmp_path / "bins"
    d.mkdir()
    oracle_lib.write_bin_sketch(str(d / "a.bin"), np.arange(1, 200, dtype=np.uint64))
    oracle_lib.write_bin_sketch(str(d / "b.bin"), np.arange(100, 300, dtype=np.uint64))
    rc, line, _ = _run(exe, "bins", str(d), str(tmp_path / "bins_out"))
    assert rc == KSP_E_HIP, line
    blob = (d / "b.bin").read_bytes()
    s0, c0 = struct.unpack_from("<QQ", blob, 0)
    for name, data in {"cut": blob[: len(blob) // 2], "size_gt_cap": struct.pack("<QQ", c0 + 1, c0) + blob[16:],
                       "cap_even": struct.pack("<QQ", s0, c0 + 1) + blob[16:], "tiny": blob[:5]}.items():
        (d / "b.bin").write_bytes(data)
        rc, line, _ = _run(exe, "bins", str(d), str(tmp_path / ("bins_" + name)))
        assert rc == KSP_E_IO, (name, line)


def test_multi_device_barrier_returns_on_an_injected_failure_at_every_stage(exe):
    """run_multi's host threads (one per device; kspider_amd/csrc/engine.hip) meet at ksp::FailBarrier sync points.  A
    failure injected into any worker at any of the 12 stages must make EVERY worker return at the sync point behind
    that stage — none left waiting at a later barrier (the round-2 race: a flag read after leaving a plain barrier).
    ThreadSanitizer build; a watchdog inside the program turns a hang into exit code 3."""
    check = os.path.join(os.path.dirname(exe), "host_sync_check")
    p = subprocess.run([check, "12"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    assert "host_sync ok" in p.stdout, p.stdout
