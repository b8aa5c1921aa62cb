"""CPU: the C-ABI library loads and exports every symbol include/recman_hip.h declares."""
import os
import re

from recman_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "recman_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "rm_embed_fwd" in syms and "rm_cross_bwd" in syms and len(syms) >= 10


def test_library_exports_every_declared_symbol(hip_lib):
    for name in declared_symbols():
        assert hasattr(hip_lib, name), f"{name} declared in recman_hip.h but not exported"


def test_binding_table_covers_the_header(hip_lib):
    bound = set(_lib.SIGNATURES) | set(_lib.SIGNATURES_I64) | {"rm_last_error"}
    assert set(declared_symbols()) == bound, set(declared_symbols()) ^ bound


def test_version_call_without_gpu(hip_lib):
    assert hip_lib.rm_version() >= 100
    assert isinstance(hip_lib.rm_last_error(), bytes)
