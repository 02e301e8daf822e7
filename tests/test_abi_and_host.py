"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/eaz.h declares, the ctypes
binding covers the same set, host-only helpers behave, and the product never routes through the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "eaz.h")).read()
    return sorted(set(re.findall(r"EAZ_API\s+(?:const\s+char\*|int)\s+(eaz_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(eaz):
    names = _declared()
    assert len(names) >= 30
    lib = C.CDLL(eaz.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libeaz.so does not export {n}"
    out = subprocess.check_output(["nm", "-D", "--defined-only", eaz.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (eaz_\w+)", out)))
    assert exported == names, "exported symbols and include/eaz.h differ"


def test_ctypes_binding_matches_header(eaz):
    assert sorted(list(eaz.SIGNATURES) + ["eaz_last_error", "eaz_build_hash"]) == _declared()
    assert eaz.load_library().eaz_version() >= 100


def test_library_carries_the_hash_of_the_sources_it_was_built_from():
    """csrc/Makefile bakes the SHA-256 prefix of csrc/*.hip, *.hpp and include/eaz.h into the library; profiles/ files and bench.py's
    `traffic_source_status` compare against it, so a stale library (or a stale profile) is visible instead of silently attached."""
    from e_alphazero_amd import _lib

    assert re.fullmatch(r"[0-9a-f]{12}", _lib.source_hash())
    assert _lib.build_hash() == _lib.source_hash(), "libeaz.so was not rebuilt after the last source change (run __graft_entry__.build())"


def test_every_tuning_knob_of_the_header_is_bound_and_validated(eaz):
    """include/eaz.h EAZ_TUNE_*: ids are dense, engine.tune() has a keyword per knob, the library accepts each knob's default and refuses values
    outside its list and unknown knobs (eaz_tune_set only stores the value: no device needed)."""
    import inspect

    from e_alphazero_amd import engine

    text = open(os.path.join(ROOT, "include", "eaz.h")).read()
    knobs = {name: int(v) for name, v in re.findall(r"#define (EAZ_TUNE_\w+) (\d+)", text)}
    assert sorted(knobs.values()) == list(range(len(knobs))) and len(knobs) == 8
    kw = list(inspect.signature(engine.tune).parameters)
    assert [k.upper() for k in kw] == [n[len("EAZ_TUNE_"):] for n, _ in sorted(knobs.items(), key=lambda kv: kv[1])]
    lib = eaz.load_library()
    defaults = dict(fc_tile=0, fused_leaf=1, tree_block=64, tree_spread=0, persist=1, shards=0, subleq_wave=1, lazy_rows=1)
    bad = dict(fc_tile=7, tree_block=96, tree_spread=3, persist=5, shards=8, subleq_wave=2, lazy_rows=3)
    for name, value in bad.items():
        assert lib.eaz_tune_set(kw.index(name), value) != 0, f"{name}={value} must be refused"
    assert lib.eaz_tune_set(len(knobs), 0) != 0 and b"unknown knob" in lib.eaz_last_error()
    engine.tune(**defaults)


def test_state_struct_sizes_match_header():
    from helpers import DEVICE_SUBLEQ

    text = open(os.path.join(ROOT, "include", "eaz.h")).read()
    assert "eaz_deepsea_state; /* 16 bytes */" in text and "eaz_subleq_state; /* 336 bytes */" in text
    assert DEVICE_SUBLEQ.itemsize == 336


def test_no_gpu_means_loud_failure(eaz):
    """No CPU fallback: without a ROCm device the engine refuses to create a context."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from e_alphazero_amd import engine

    with pytest.raises(eaz.EazError):
        engine.EnvHandle.deepsea(8, None)
    out = C.c_void_p()
    assert eaz.load_library().eaz_ctx_create(0, C.byref(out)) != 0
    assert b"hip" in eaz.load_library().eaz_last_error().lower()


def test_bench_and_library_fail_loudly_without_a_gpu_or_library(tmp_path):
    """bench.py exits non-zero with a message instead of falling back to anything; a missing libeaz.so is a hard error."""
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "ROCm device" in (r.stderr + r.stdout)
        assert not any(line.startswith("{") for line in r.stdout.splitlines()), "no result line may be printed without a GPU"
    code = "import e_alphazero_amd as m\ntry:\n    m.load_library()\nexcept m.EazError as e:\n    print('LOUD', e)\n"
    env = dict(os.environ, EAZ_LIB=str(tmp_path / "missing_libeaz.so"), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "LOUD" in r.stdout and "no CPU fallback" in r.stdout.replace("There is no CPU fallback", "no CPU fallback")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "e-alphazero_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "eaz_oracle" not in text and "oracle/" not in text and "eazo_" not in text, f


def test_params_layout_and_npz_roundtrip(eaz, tmp_path):
    p = eaz.params.init_fc_params(64, 256, 2, seed=0)
    assert list(p)[0] == "fc_az_net/linear" and list(p)[-1] == "fc_az_net/linear_11"
    assert p["fc_az_net/linear"]["w"].shape == (64, 256) and p["fc_az_net/linear_2"]["w"].shape == (256, 1)
    assert p["fc_az_net/linear_8"]["w"].shape == (256, 2) and p["fc_az_net/linear_11"]["b"].shape == (2,)
    w = p["fc_az_net/linear"]["w"]
    assert np.abs(w).max() <= 2 / 8 + 1e-6 and abs(w.std() - 0.88 / 8) < 0.01  # trunc-normal(2 sigma), sigma = 1/sqrt(64)
    s = eaz.params.init_hash_state(fill=0.5, seed=1)
    bs = s["fc_az_net/xxhash32"]["binary_set"]
    assert bs.shape == (1 << 21,) and abs(np.unpackbits(bs[:4096]).mean() - 0.5) < 0.05
    path = str(tmp_path / "ckpt.npz")
    eaz.params.save_npz(path, p, s)
    p2, s2 = eaz.params.load_npz(path)
    assert all(np.array_equal(p[k]["w"], p2[k]["w"]) and np.array_equal(p[k]["b"], p2[k]["b"]) for k in p)
    assert np.array_equal(s2["fc_az_net/xxhash32"]["binary_set"], bs)
    # legacy (prefix-less) files: split by the documented heuristic; a weight leaf under a "/~/" module is ambiguous and refused
    legacy = str(tmp_path / "legacy.npz")
    np.savez(legacy, **{"fc_az_net/linear/w": p["fc_az_net/linear"]["w"], "fc_az_net/xxhash32/binary_set": bs,
                        "az_resnet/batch_norm/~/mean_ema/average": np.zeros(4, np.float32)})
    p3, s3 = eaz.params.load_npz(legacy)
    assert list(p3) == ["fc_az_net/linear"] and set(s3) == {"fc_az_net/xxhash32", "az_resnet/batch_norm/~/mean_ema"}
    bad = str(tmp_path / "bad.npz")
    np.savez(bad, **{"outer/~/inner/w": np.zeros((2, 2), np.float32)})
    with pytest.raises(ValueError, match="cannot tell"):
        eaz.params.load_npz(bad)


def test_subleq16_test_vector_table_matches_the_oracle_generator(eaz, oracle):
    """The one-wavefront Subleq interpreter reads its test vectors from a compile-time table (csrc/envs.hpp kSq16Cases) instead of
    running get_test_cases (subleq.py:398-501) per environment.  Most entries never decide a GPU parity test (cases 1 and 2 only run
    when case 0 passed), so the table is held here, entry by entry, against the oracle's generator -- every SubleqTask, the tasks beyond
    the last branch included (lax.switch clamps the index)."""
    import ctypes as C

    lib = eaz.load_library()
    for task in range(0, 16):
        tin = np.zeros((3, 8), np.int32)
        tout = np.zeros((3, 8), np.int32)
        oracle.lib().eazo_subleq_test_cases(task, 16, tin.ctypes.data, tout.ctypes.data)
        for c in range(3):
            a = np.zeros(8, np.uint16)
            b = np.zeros(8, np.uint16)
            assert lib.eaz_subleq16_test_vectors(task, c, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(a.astype(np.int32), tin[c]) and np.array_equal(b.astype(np.int32), tout[c]), (task, c)
    # the run-time generator the per-lane kernel evaluates for the other word sizes (csrc/envs.hpp subleq_test_case, compiled for the host too):
    # at 16 it must agree with the table, and at every other size with the oracle
    for W in (16, 17, 32, 100, 256):
        for task in range(0, 16):
            tin = np.zeros((3, 8), np.int32)
            tout = np.zeros((3, 8), np.int32)
            oracle.lib().eazo_subleq_test_cases(task, W, tin.ctypes.data, tout.ctypes.data)
            for c in range(3):
                a = np.zeros(8, np.uint16)
                b = np.zeros(8, np.uint16)
                assert lib.eaz_subleq_test_vectors(W, task, c, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
                assert np.array_equal(a.astype(np.int32), tin[c]) and np.array_equal(b.astype(np.int32), tout[c]), (W, task, c)
