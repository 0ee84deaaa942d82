"""CPU-side checks of the C-ABI boundary: the shared library loads and exports exactly what include/spmm_hip.h declares.
No compute call is made (there is no GPU in the dev container)."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    so = os.path.join(ROOT, "spmm_amd", "libspmm_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    return so


def test_library_exports_every_declared_symbol(built):
    from spmm_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 30
    cdll = ctypes.CDLL(built)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in include/spmm_hip.h but not exported"
    # and nothing spmm_* is exported that the header does not declare
    out = subprocess.run(["nm", "-D", "--defined-only", built], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("spmm_")}
    assert exported - {"spmm_set_error"} == set(protos), exported ^ set(protos)


def test_version_and_error_channel(built):
    from spmm_amd._lib import lib
    L = lib()
    assert L.cdll.spmm_version() == 100
    assert isinstance(L.cdll.spmm_last_error(), bytes)


def test_bad_shape_is_reported_without_touching_the_gpu(built):
    """Argument validation happens before any launch, so it can be exercised on a CPU-only box."""
    from spmm_amd._lib import lib
    L = lib()
    with pytest.raises(RuntimeError, match="multiple of 64"):
        L.call("spmm_gemm_nt", None, 8, None, 8, 16, 16, 100, 1, None, None, 1.0, None, 0, None, 0, None, 16, None, 0, 0, None, 0, None, None)
    with pytest.raises(RuntimeError, match=r"must be in \[1,256\]"):
        L.call("spmm_attn_fwd", None, 64, None, 64, None, 64, None, None, None, None, None, None, None, 64, None, 1, 1, 300, 54, 1, 0, 0.0, None, 0, 0, 0, None)
    with pytest.raises(RuntimeError, match="SPMM_models.py:279"):
        L.call("spmm_enqueue", None, 5, 64, None, 16, None, None, 64, 4, None, 1, None, None)


def test_product_refuses_to_run_without_the_extension(tmp_path):
    """No CPU / eager fallback: with the shared library missing the library handle and the ops that reach it raise and name the file
    they looked for (ops that take a stream fail even earlier on a box without a GPU: torch reports the missing device)."""
    code = (
        "import sys, torch\n"
        "from spmm_amd import _lib, ops\n"
        "for what, fn in (('lib', _lib.lib), ('op', ops.adam_scalars_bytes), ('query', lambda: ops.xattn_supported(768, 12, 54, 128))):\n"
        "    try:\n"
        "        fn()\n"
        "    except RuntimeError as e:\n"
        "        assert 'no CPU / eager fallback' in str(e) and 'missing_lib.so' in str(e), e\n"
        "    else:\n"
        "        sys.exit(what + ' did not raise')\n"
        "print('refused')\n")
    env = dict(os.environ, SPMM_HIP_LIB=str(tmp_path / "missing_lib.so"), PYTHONPATH=ROOT)
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and "refused" in r.stdout, r.stdout + r.stderr
