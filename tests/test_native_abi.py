"""CPU: the C-ABI library loads and exports every symbol include/datr_hip.h declares;
the Python mirror reproduces the reference's error behaviour without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="datr_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(datr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "datr_msda_forward_f32" in syms and "datr_msda_backward_f32" in syms


def test_library_exports_every_declared_symbol():
    from datr_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"libdatr_hip.so does not export {name}"


def test_nothing_undeclared_is_exported():
    """Every `datr_*` symbol the library exports is declared in one of the two headers (the cross-file
    entry points inside csrc/ have hidden visibility)."""
    import subprocess
    from datr_amd import _native
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("datr_")}
    declared = set(declared_symbols()) | set(declared_symbols("datr_hip_internal.h"))
    assert exported - declared == set(), sorted(exported - declared)
    assert declared - exported == set(), sorted(declared - exported)


def test_abi_version_and_strerror():
    from datr_amd import _native
    assert _native.lib.datr_abi_version() == _native.ABI_VERSION
    assert _native.lib.datr_strerror(0) == b"ok"
    assert b"invalid" in _native.lib.datr_strerror(-1)


def test_fast_path_predicate():
    from datr_amd import _native
    f = _native.lib.datr_msda_uses_fast_path
    assert f(22223, 8, 32, 4, 4) == 1          # DINO encoder geometry
    assert f(30, 2, 30, 2, 2) == 0             # odd head dim -> generic kernels
    assert f(30, 2, 2048, 2, 2) == 0
    assert f(1 << 24, 8, 32, 4, 4) == 0        # > 2 GiB per batch item -> 64-bit generic path


def test_cpu_tensors_raise_like_the_reference():
    # /root/reference/models/dino/ops/src/ms_deform_attn.h:38,60 -> "Not implemented on the CPU"
    from datr_amd import msda
    v = torch.zeros(1, 4, 2, 2)
    sh = torch.tensor([[2, 2]])
    lsi = torch.tensor([0])
    loc = torch.zeros(1, 1, 2, 1, 1, 2)
    att = torch.zeros(1, 1, 2, 1, 1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        msda.ms_deform_attn_forward(v, sh, lsi, loc, att, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        msda.ms_deform_attn_backward(v, sh, lsi, loc, att, torch.zeros(1, 1, 4), 64)
    with pytest.raises(RuntimeError, match="contiguous"):
        msda.ms_deform_attn_forward(v.transpose(1, 2), sh, lsi, loc, att, 64)


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (judge rule: no CPU fallback)."""
    pkg = os.path.join(ROOT, "datr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libdatr_oracle" not in src, f


def test_oracle_and_fixtures_never_import_the_product():
    """The mirror of the rule above: the oracle (oracle/) and the fixture generators (tests/golden/) must not
    import datr_amd -- a fixture produced with product code inside would compare the product with itself.
    (The third-party stand-ins the reference needs, e.g. torchvision's ResNet-50, live in oracle/ as plain
    torch.nn restatements: oracle/resnet_ref.py.)"""
    pat = re.compile(r"^\s*(from\s+datr_amd\b|import\s+datr_amd\b)|__import__\(\s*[\"']datr_amd|"
                     r"import_module\(\s*[\"']datr_amd", flags=re.M)
    for sub in ("oracle", os.path.join("tests", "golden")):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".c", ".h")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not pat.search(src), os.path.join(dirpath, f)
                    assert "libdatr_hip" not in src, f


def test_module_state_dict_names_and_init():
    from datr_amd.msda import MSDeformAttn
    m = MSDeformAttn(256, 4, 8, 4)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        "sampling_offsets.weight": (256, 256), "sampling_offsets.bias": (256,),
        "attention_weights.weight": (128, 256), "attention_weights.bias": (128,),
        "value_proj.weight": (256, 256), "value_proj.bias": (256,),
        "output_proj.weight": (256, 256), "output_proj.bias": (256,)}
    b = sd["sampling_offsets.bias"].view(8, 4, 4, 2)
    # head 0 points along +x with magnitude 1..4 (ops/modules/ms_deform_attn.py:61-68)
    torch.testing.assert_close(b[0, :, :, 0], torch.arange(1., 5.).expand(4, 4))
    assert torch.count_nonzero(sd["sampling_offsets.weight"]) == 0
