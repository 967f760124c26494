"""CPU: the C-ABI library builds, loads, and exports every symbol include/mpn.h declares."""
import os
import re

from multiposenet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mpn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert "mpn_heatmap_decode" in names and "mpn_version" in names
    l = _lib.lib()
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/mpn.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.SIGNATURES"
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound in _lib.py but not declared in include/mpn.h"


def test_library_exports_nothing_the_header_does_not_declare():
    """`nm -D` of the built library: every exported mpn_* symbol is declared in include/mpn.h (no hidden tuning or debug
    entry points; diagnostic stamps exist only in -DMPN_DIAG builds made by tools/build_variant.sh)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("mpn_")})
    assert exported, "nm found no mpn_* exports"
    declared = set(_declared())
    extra = [n for n in exported if n not in declared]
    assert not extra, f"exported but not declared in include/mpn.h: {extra}"


def test_no_environment_switches_in_launch_paths():
    """The library is stateless: no getenv in any kernel source (tuning constants are compile-time)."""
    csrc = os.path.join(ROOT, "multiposenet_amd", "csrc")
    bad = [f for f in os.listdir(csrc) if f.endswith((".hip", ".h")) and "getenv" in open(os.path.join(csrc, f)).read()]
    assert not bad, bad


def test_version_and_error_string():
    l = _lib.lib()
    import re
    hdr = open(os.path.join(ROOT, "include", "mpn.h")).read()
    declared = int(re.search(r"#define\s+MPN_VERSION\s+(\d+)", hdr).group(1))
    assert l.mpn_version() == declared == _lib.MPN_VERSION == 400
    assert isinstance(_lib.last_error(), str)


def test_host_side_validation_needs_no_gpu():
    # argument checks run before any HIP call
    import ctypes
    import pytest
    with pytest.raises(ValueError, match="C must be 17"):
        _lib.call("mpn_heatmap_decode", None, 0, 1, 4, 4, 16, None, 0.0, None, None, None, None, 0, None)
    assert _lib.lib().mpn_heatmap_decode_workspace_bytes(32) >= 32 * 17 * 8


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under multiposenet_amd/ may import it (bench.py's CPU-baseline legs and
    smoke() live at the repository root for that reason)."""
    import ast
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multiposenet_amd")
    bad = []
    for d, _, files in os.walk(root):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(d, f)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom) and node.level == 0:
                    names = [node.module or ""]
                if any(n == "oracle" or n.startswith("oracle.") for n in names):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
