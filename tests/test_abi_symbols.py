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
    assert lib.drs_abi_version() == 6


def test_argument_validation_without_gpu():
    from diffusionremotesensing_amd import _lib
    lib = _lib.load()
    assert lib.drs_noise_images(None, None, None, None, 10, None, 1, 1, None) == 1
    assert b"null pointer" in lib.drs_last_error()
    assert lib.drs_conv2d_workspace_bytes(1, 16, 8, 8, 16, 3, 3, 1, 1, 0, 0) > 2 * 16 * 64 * 4
