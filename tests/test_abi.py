"""CPU: the C-ABI library loads and exports every symbol include/neat_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "neat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(neat_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = _declared_symbols()
    assert len(syms) >= 25
    for must in ("neat_ln_fwd", "neat_ln_bwd", "neat_gemm_bf16", "neat_attn_fwd", "neat_attn_bwd",
                 "neat_coarse_align", "neat_tta_loss_fwd_bwd", "neat_adamw_flat", "neat_topk_rows"):
        assert must in syms


def test_library_exports_all_declared_symbols():
    from neat_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        pytest.fail(f"{lib.LIB_PATH} missing: run __graft_entry__.build()")
    dll = ctypes.CDLL(lib.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(dll, s), f"symbol {s} not exported"
    assert set(_declared_symbols()) == set(lib.SIGNATURES), "ctypes SIGNATURES out of sync with the header"
    dll.neat_version.restype = ctypes.c_char_p
    assert b"gfx950" in dll.neat_version()


def test_pure_host_entry_points():
    """Entry points that do no device work can be called without a GPU."""
    from neat_amd import lib
    L = lib.load()
    assert L.neat_ln_bwd_blocks(78848) >= 1
    assert L.neat_coarse_ws_floats(256, 5000) == 2 * 256 * 79
    assert L.neat_loss_ws_floats(256, 256) == 256 * 256 * 2 + 256 + 3 * 256
    assert L.neat_topk_ws_bytes(5000, 25014, 10) > 0


def test_argument_validation_without_gpu():
    """Shape/pointer validation happens before any HIP call and reports through neat_last_error()."""
    from neat_amd import lib
    L = lib.load()
    rc = L.neat_gemm_bf16(None, None, None, None, None, None, None, 8, 8, 64, 0, None)
    assert rc == -2 and "null" in lib.last_error()
    rc = L.neat_ln_fwd(1, 1, 1, 1, None, None, None, 4, 1030, 1e-5, None)
    assert rc == -1 and "cols" in lib.last_error()
    rc = L.neat_topk_rows(1, 7, 3, 4, 4, 2, 1, None, None, None)
    assert rc == -1


def test_comm_entry_points_fail_cleanly_without_a_communicator():
    """neat_comm_*: no communicator -> NEAT_ERR_COMM (-4) with a message, before any RCCL or HIP call."""
    from neat_amd import lib
    L = lib.load()
    assert L.neat_comm_world() == 0 and L.neat_comm_rank() == -1
    assert L.neat_comm_allreduce_f32(1, 1, 4, None) == -4 and "no communicator" in lib.last_error()
    assert L.neat_comm_allgather(1, 1, 4, None) == -4
    assert L.neat_comm_init(None, 1, 0) == -2
    assert L.neat_comm_destroy() == 0
    assert L.neat_sim_topk_ws_bytes(25014, 5000, 10) > 64 * 1024 * 1024 // 2


def test_product_path_has_no_oracle_import():
    """The shipped package must never import the oracle (parity claims depend on it)."""
    pkg = os.path.join(ROOT, "neat_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_integration_md_stub_binds_against_the_library():
    """The ctypes stub printed in INTEGRATION.md is executed as written (only the library path is substituted) and every
    binding it makes must agree with neat_amd.lib.SIGNATURES."""
    import ctypes as C
    from neat_amd import lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"## ctypes stub.*?```python\n(.*?)```", md, flags=re.S).group(1)
    assert "libneat_hip.so" in block
    block = block.replace('"libneat_hip.so"', repr(lib.LIB_PATH))
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    bound = {v.__name__: v for v in ns.values() if isinstance(v, C._CFuncPtr) and hasattr(v, "__name__")}
    assert len(bound) >= 30
    for name, fn in bound.items():
        res, args = lib.SIGNATURES[name]
        assert list(fn.argtypes) == list(args), name
        assert fn.restype == res, name


def test_create_model_and_transforms_returns_transforms():
    import numpy as np
    import torch
    from neat_amd import open_clip as oc
    model, tr, val = oc.create_model_and_transforms("neat-tiny", device="cpu")
    assert tr.size == val.size == 32 and tr.is_train and not val.is_train
    img = np.random.default_rng(0).integers(0, 255, (20, 30, 3), dtype=np.uint8)
    out = val(img)
    assert out.dtype == torch.uint8 and tuple(out.shape) == (20, 30, 3) and out.is_contiguous()
    from PIL import Image
    assert torch.equal(val(Image.fromarray(img)), out)
    with pytest.raises(TypeError):
        val(torch.zeros(3, 20, 30))
