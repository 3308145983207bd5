"""CPU: the C-ABI library loads and exports every symbol include/rlcf_hip.h declares
(no compute calls: there is no GPU here), and the host-side bookkeeping is sane."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from rlcf_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "rlcf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rlcf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rlcf_stream"}
    h = lib.lib()
    for name in sorted(declared):
        assert hasattr(h, name), f"{name} declared in rlcf_hip.h but not exported"
        assert name in lib.SIGNATURES, f"{name} has no ctypes signature in rlcf_amd/_lib.py"
    assert set(lib.SIGNATURES) <= declared


def test_version_and_error_text(lib):
    h = lib.lib()
    assert h.rlcf_version() >= 1
    assert isinstance(h.rlcf_last_error(), bytes)


def test_engine_refuses_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rlcf_amd import synth
    from rlcf_amd.engine import Engine
    with pytest.raises(lib.RlcfError):
        Engine(synth.GEOMETRIES["tiny"], synth.GEOMETRIES["tiny-r"], 8, 16)


def test_struct_layouts(lib):
    import ctypes as C
    assert C.sizeof(lib.ClipCfg) == 56 and C.sizeof(lib.Seq) == 16
    assert C.sizeof(lib.TTAArgs) == 72 and C.sizeof(lib.TTAOut) == 18 * 8      # + n_sel, + step_skipped (ABI version 4)
    assert C.sizeof(lib.Crop) == 20 and C.sizeof(lib.AugmixOp) == 56          # rlcf_crop, rlcf_augmix_op {int, int, double[6]}


def test_bpe_tokenizer_matches_reference_fixture():
    """rlcf_amd.bpe.ClipBPE vs clip.tokenize of the reference (tests/golden/tokenizer.npz).  Needs OpenAI's merges file,
    which is data this repo does not ship: read from the reference checkout when present."""
    import sys
    import numpy as np
    vocab = os.environ.get("RLCF_BPE_VOCAB", "/root/reference/TPT/clip/bpe_simple_vocab_16e6.txt.gz")
    if not os.path.exists(vocab):
        pytest.skip("BPE merges file not available")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import TOKENIZER_STRINGS
    from rlcf_amd.bpe import ClipBPE
    tok = ClipBPE(vocab)
    got = tok.tokenize(TOKENIZER_STRINGS).numpy()
    ref = np.load(os.path.join(ROOT, "tests", "golden", "tokenizer.npz"))["tokens"]
    assert got.shape == ref.shape and (got == ref).all()
    assert tok.sot == 49406 and tok.eot == 49407
    with pytest.raises(RuntimeError):
        tok.tokenize("word " * 100)
    assert tok.tokenize("word " * 100, truncate=True)[0, -1] == tok.eot
