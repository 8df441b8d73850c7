"""The C-ABI library loads without a GPU and exports every symbol include/drs_hip.h declares."""
import os
import re


def test_every_declared_symbol_is_exported_and_bound():
    from diffusionremotesensing_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "drs_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(drs_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in drs_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.drs_abi_version() == 7


def test_the_library_exports_nothing_but_the_declared_symbols():
    """-fvisibility=hidden + DRS_API: the dynamic symbol table's defined functions are exactly the header's entry points (the
    internal drs_launch_* / drs_*_supported C++ functions were all visible before round 5)."""
    import subprocess
    from diffusionremotesensing_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "drs_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(drs_[a-z0-9_]+)\s*\(", header))
    _lib.load()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.lib_path()], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in ("T", "t", "W")}
    exported = {n for n in exported if not n.startswith(("_init", "_fini", "__"))}
    assert exported == declared, sorted(exported ^ declared)


def test_argument_validation_without_gpu():
    from diffusionremotesensing_amd import _lib
    lib = _lib.load()
    assert lib.drs_noise_images(None, None, None, None, 10, None, 1, 1, None) == 1
    assert b"null pointer" in lib.drs_last_error()
    assert lib.drs_conv2d_workspace_bytes(1, 16, 8, 8, 16, 3, 3, 1, 1, 0, 0) > 2 * 16 * 64 * 4
