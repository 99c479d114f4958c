"""CPU: the committed counter passes are tied to a binary.  profiles/pmc_sq.json and
profiles/pmc_traffic.json carry, per "<workload>/sub<K>[_T<steps per launch>]" entry, the sha256 of the libauv_hip.so they were measured on;
bench.py hashes the library it runs and marks a leg `stale` when the two differ.  Here: the files have that shape, the
hash function hashes the in-tree library, and -- informational, printed, not asserted, so that a kernel change does not
turn the suite red before the next profile run -- whether the committed passes belong to the library as built now."""
import hashlib
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_counter_files_carry_the_library_hash_and_bench_hashes_the_library():
    from gym_auv_amd import _capi
    bench = _bench()
    sha = bench.library_sha256()
    assert sha == hashlib.sha256(open(_capi.LIB_PATH, "rb").read()).hexdigest() and len(sha) == 64
    for name, need in (("pmc_sq.json", ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "clock_ghz")),
                       ("pmc_traffic.json", ("bytes_raw", "bytes_reads_doubled"))):
        doc = json.load(open(os.path.join(ROOT, "profiles", name)))
        entries = {k: v for k, v in doc.items() if k != "_note"}
        assert "polygons50/sub4" in entries and "polygons50/sub1" in entries, name
        for key, e in entries.items():
            # "<workload>/sub<K>": K one-step launches per step; "<workload>/sub<K>_T<T>": K chains of T steps per launch
            k, _, t = key.split("/sub")[1].partition("_T")
            assert "/sub" in key and len(e["lib_sha256"]) == 64 and abs(e["launches_per_step"] - int(k) / float(t or 1)) < 1e-9, (name, key)
            assert all(k in e for k in need), (name, key)
            assert os.path.exists(os.path.join(ROOT, e["profile"])), e["profile"]
            print("%s [%s]: measured on %s... -> %s" % (name, key, e["lib_sha256"][:12], "current build" if e["lib_sha256"] == sha else "STALE for the current build"))
