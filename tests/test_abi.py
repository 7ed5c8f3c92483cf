"""CPU: the C-ABI shared library loads and exports every symbol declared in include/tgis_hip.h (no compute)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tgis_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgis_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tgis_amd import native

    lib = native.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in tgis_hip.h but not exported by libtgis_hip.so"
        assert name in native.SIGNATURES, f"{name} has no ctypes signature in tgis_amd.native"
    assert sorted(native.SIGNATURES) == declared, "native.SIGNATURES and tgis_hip.h disagree"


def test_info_calls_without_gpu():
    from tgis_amd import native

    lib = native.load_library()
    assert lib.tgis_arch() == b"gfx950"
    assert b"gfx950" in lib.tgis_version()
    # pure host-side size queries
    assert lib.tgis_gptq_prepared_bytes(4096, 4096, 32) > 4096 * 4096 // 2
    assert lib.tgis_dense_prepared_bytes(32000, 4096) == 1000 * 64 * 4096
    assert lib.tgis_attn_num_splits(32, 32, 32, 1, 1024) == 1
    assert lib.tgis_attn_num_splits(1, 8, 64, 1, 4096) > 1
    assert lib.tgis_attn_num_splits(4, 8, 64, 512, 4096) == 1  # prefill never splits


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: tensors that are not on the GPU are rejected by the binding, and FlashCausalLM refuses to
    construct when no GPU is present."""
    import pytest
    import torch

    from tgis_amd import native

    with pytest.raises(native.TgisHipError):
        native.rmsnorm_residual(torch.zeros(2, 64, dtype=torch.float16), None, torch.ones(64, dtype=torch.float16), 1e-5)
    if not torch.cuda.is_available():
        from tgis_amd.models.flash_causal_lm import FlashCausalLM

        with pytest.raises(NotImplementedError):
            FlashCausalLM("x", None, "synthetic", torch.float16, None, engine=object())


def test_argument_validation_needs_no_gpu():
    """Entry points validate their arguments before touching the device: bad calls return TGIS_EINVAL with a message."""
    from tgis_amd import native

    lib = native.load_library()
    none = [None] * 6
    rc = lib.tgis_warp_sample(None, 0, None, 0, 1, 10, *none, 0, 0, -1, None, -1, None, None, None, None, None, None)
    assert rc == -1 and b"tgis_warp_sample" in lib.tgis_last_error()
    rc = lib.tgis_argmax_logprob(None, 0, 1, 10, 1, 0, None, None, None, 0, None)
    assert rc == -1 and b"tgis_argmax_logprob" in lib.tgis_last_error()
    lib.tgis_clear_error()


def test_integration_doc_names_only_declared_entry_points():
    """INTEGRATION.md is what a maintainer binds from: outside its 'NOT in this ABI any more' section every tgis_* call it
    names must be declared in include/tgis_hip.h (round-4 review: three rows pointed at entry points that had moved out)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    kept = re.sub(r"## Entry points that are NOT in this ABI any more.*?(?=\n## )", "", doc, flags=re.S)
    assert kept != doc
    declared = set(_declared_symbols())
    called = set(re.findall(r"\b(tgis_[a-z0-9_]+)\s*\(", kept)) | set(re.findall(r"`(tgis_[a-z0-9_]+)`", kept))
    not_calls = {"tgis_hip", "tgis_amd", "tgis_native", "tgis_experiments"}
    missing = sorted(n for n in called - not_calls if n not in declared)
    assert not missing, missing
