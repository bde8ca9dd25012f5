"""The C-ABI library loads on a CPU-only box, exports every symbol include/adrates.h declares, and the
pricing path refuses to run without a GPU (no silent fallback)."""
import os
import re

import pytest

from adrates_amd import _native
from adrates_amd.utils import LibError, RequestTypes

from . import _fixtures as F

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "adrates.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(adr_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(native_lib):
    names = _declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(native_lib, n), f"{n} declared in adrates.h but not exported"
    assert set(_native.EXPORTED_SYMBOLS) <= set(names)
    assert native_lib.adr_version() >= 100


def test_header_cites_reference_seam():
    text = open(HEADER).read()
    for needle in ("engine.py:2362-2412", "engine.py:2414-2448", ":2639-2728", "portfolio.py:39-66"):
        assert needle in text or needle.replace("engine.py", "") in text


def test_no_cpu_fallback(native_lib):
    import torch
    if torch.cuda.is_available() and native_lib.adr_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(LibError):
        _native.Context(0)
    vd = F.README_VALUE_DT
    model = F.readme_model()
    swap = F.make_swap(vd, "10Y", 0.045)
    with pytest.raises(LibError):
        swap.position(model).compute([RequestTypes.VALUE])


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dirpath, _, files in os.walk(os.path.join(root, "adrates_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "oracle/port" in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def _build_c_example(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_example")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_abi_example.c"), "-L", os.path.join(root, "adrates_amd"),
                           "-ladrates_hip", "-Wl,-rpath," + os.path.join(root, "adrates_amd"), "-o", exe])
    return exe


def test_c_example_builds_against_the_header_as_c99(native_lib, tmp_path):
    """include/adrates.h is plain C: examples/c_abi_example.c compiles with gcc -std=c99 -Werror and links against
    the library.  Without a GPU it must stop at adr_init with the no-fallback message (the run on a GPU is
    tests/test_gpu_parity_small.py::test_c_example_matches_the_python_path)."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    if not torch.cuda.is_available():
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 1 and "no CPU fallback" in out.stderr
