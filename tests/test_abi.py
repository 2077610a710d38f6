"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/dismember_hip.h declares; with no GPU it refuses to run instead of falling back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "dismember_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def native():
    from dismember_amd import _native
    _native.build()
    return _native


def test_every_declared_symbol_is_exported(native):
    lib = C.CDLL(native.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    # the Python binding table covers exactly the declared ABI
    assert sorted(native.SIGNATURES) == declared


def test_no_oracle_or_cpu_fallback_linked(native):
    out = os.popen("ldd %s" % native.LIB_PATH).read()
    assert "libamdhip64" in out
    assert "oracle" not in out
    blob = open(native.LIB_PATH, "rb").read()
    assert b"orc_" not in blob


def test_fails_loudly_without_gpu(native):
    lib = native.lib()
    n = C.c_int(-1)
    lib.dm_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    rc = lib.dm_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"no CPU fallback" in lib.dm_last_error(None)
    from dismember_amd import DismemberError, Engine
    with pytest.raises(DismemberError):
        Engine(0)


def test_level_start_integer(native):
    lib = native.lib()
    s, l = C.c_int(), C.c_int()
    for n, exp in ((1, (0, 0)), (2, (1, 1)), (3, (1, 1)), (20, (15, 4)), (200, (127, 7)), (256, (255, 8))):
        assert lib.dm_level_start(n, C.byref(s), C.byref(l)) == 0
        assert (s.value, l.value) == exp
    assert lib.dm_level_start(0, C.byref(s), C.byref(l)) != 0


def test_cpp_host_facade_compiles_and_links(tmp_path):
    """include/dismember.hpp (the C++ mirror of the Scala facades) builds against the C ABI with a plain host compiler:
    no HIP headers, no torch (linking only; nothing runs without a GPU)."""
    import subprocess
    exe = str(tmp_path / "facade_test")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "dismember_amd"), "-ldismember_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "dismember_amd"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)


def test_null_handle_is_an_error_not_a_crash():
    """Every entry point that takes a handle rejects NULL with DM_ERR_INVALID (no device needed)."""
    import ctypes as C
    from dismember_amd import _native as N
    lib = N.lib()
    skip = {"dm_version", "dm_device_count", "dm_create", "dm_last_error", "dm_level_start", "dm_comm_last_error", "dm_last_beam_kernel"}
    checked = 0
    for name, (restype, argtypes) in N.SIGNATURES.items():
        if name in skip or not argtypes or argtypes[0] is not C.c_void_p:
            continue
        args = [None]
        for t in argtypes[1:]:
            if t in (C.c_int, C.c_int32, C.c_int64, C.c_size_t, C.c_uint64):
                args.append(0)
            elif t in (C.c_float, C.c_double):
                args.append(0.0)
            else:
                args.append(None)
        assert getattr(lib, name)(*args) == -1, name
        checked += 1
    assert checked >= 40


def test_allreduce_grads_and_clique_argument_validation():
    """dm_allreduce_grads / dm_comm_create_all (one process driving N GPUs) reject malformed calls before touching
    a device: NULL list, n < 1, NULL entries, no devices, a device listed twice (RCCL takes one rank per device)."""
    import ctypes as C
    from dismember_amd import _native as N
    lib = N.lib()
    assert lib.dm_allreduce_grads(None, 1) == -1
    hs = (C.c_void_p * 2)(None, None)
    assert lib.dm_allreduce_grads(hs, 0) == -1
    assert lib.dm_allreduce_grads(hs, 2) == -1                      # NULL handles inside the list
    out = (C.c_void_p * 2)()
    assert lib.dm_comm_create_all(0, (C.c_int * 1)(0), out) == -1
    assert lib.dm_comm_create_all(2, None, out) == -1
    assert lib.dm_comm_create_all(2, (C.c_int * 2)(0, 0), out) == -1
    assert b"twice" in lib.dm_comm_last_error(None)
    assert out[0] is None and out[1] is None
