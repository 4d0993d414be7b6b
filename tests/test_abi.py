"""CPU checks of the drop-in boundary: liblvk_hip.so builds, loads, and exports every symbol include/lvk_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lvk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lvk_(?:hip|stab)_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_python_binds():
    from livevisionkit_amd import _native
    assert _declared_symbols() == _native.symbols()


def test_library_exports_every_declared_symbol():
    from livevisionkit_amd import _native
    lib = _native.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), f"liblvk_hip.so does not export {name}"


def test_no_device_fails_loudly_not_silently():
    """Without a GPU the product path must refuse to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from livevisionkit_amd import _native
    import livevisionkit_amd as lvk
    lib = _native.load()
    handle = ctypes.c_void_p()
    rc = lib.lvk_hip_ctx_create(0, ctypes.byref(handle))
    assert rc != 0 and handle.value is None
    assert b"device" in lib.lvk_hip_last_error(None).lower()
    with pytest.raises(Exception):
        lvk.Context(0)


def test_product_never_touches_the_oracle():
    """Nothing under livevisionkit_amd/ or include/ may reference oracle/ (it is test infrastructure)."""
    bad = []
    for base in ("livevisionkit_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"lvko_|liblvk_oracle|oracle/|oracle_lib", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md maps each C-ABI symbol to the reference interface it replaces."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "lvk_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = set(re.findall(r"\b(lvk_(?:hip|stab)_[a-z0-9_]+)\s*\(", header))
    assert not [n for n in sorted(names) if n not in doc]
