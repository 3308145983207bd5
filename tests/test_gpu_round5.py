"""GPU tests, round-5 additions: the dedicated single-pass f16 GEMM of the performance mode (rlcf_gemm_f16 -> gemm_nt_f16_p8_kernel,
rlcf_amd/csrc/gemm_f16.hip) against float64 products of the SAME f16 operands — what is checked is the kernel (tiling, DMA ring
ordering, phase schedule, epilogues), not the f16 rounding of the inputs, which is the mode's labelled arithmetic
(TPT/tpt_cls_rl.py:52: torch.cuda.amp.autocast)."""
import os

import pytest
import torch

from rlcf_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _st():
    return torch.cuda.current_stream().cuda_stream


def _gemm_f16(a16, w16, bias, res, epi, want_f32, want_f16, alpha=1.0):
    M, K = a16.shape
    N = w16.shape[0]
    c = res.clone() if res is not None else (torch.empty(M, N, device=DEV) if want_f32 else None)
    c16 = torch.empty(M, N, dtype=torch.float16, device=DEV) if want_f16 else None
    L.check(L.lib().rlcf_gemm_f16(a16.data_ptr(), K, w16.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                  c.data_ptr() if res is not None else None, N, c.data_ptr() if c is not None else None, N,
                                  c16.data_ptr() if c16 is not None else None, N, M, N, K, alpha, epi, _st()))
    torch.cuda.synchronize()
    return c, c16


def _ref(a16, w16, bias, res, epi, alpha=1.0):
    r = alpha * (a16.double() @ w16.double().t())
    if bias is not None:
        r = r + bias.double()
    if epi == L.EPI_QUICKGELU:
        r = r * torch.sigmoid(1.702 * r)
    if res is not None:
        r = r + res.double()
    return r


# (M, N, K): every shape runs >= 256 tiles of 256x128 (the regime the engine hands to the eight-phase kernel); ragged M and N, one K
# tile (prologue only), two, an odd count (the unrolled pair of K tiles ends on its first half), the layer shapes of ViT-B/16
# f16 outputs with K % 128 == 0 take the PERSISTENT kernel (one workgroup per CU walks several tiles: > 256 tiles = the cross-tile DMA
# hand-over; K = 128 = a tile that is nothing but the hand-over); everything else the one-workgroup-per-tile kernel
SHAPES = [(256 * 16 + 37, 256 * 8, 64), (256 * 16, 256 * 8 + 132, 128), (256 * 20 + 250, 1024 + 4, 192), (256 * 40 + 100, 2048 + 132, 128),
          (256 * 40 + 100, 2048 + 132, 256), (12608 * 2, 768, 768), (12608 * 2, 2304, 768), (12608, 3072, 768), (12608 * 2, 768, 3072)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("kind", ["f32", "f32_res", "f16", "gelu_f16"])
def test_gemm_f16_eight_phase_kernel_matches_f64(M, N, K, kind):
    torch.manual_seed(M + N + K)
    a16 = torch.randn(M, K, device=DEV).half()
    w16 = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV) if kind == "f32_res" else None
    epi = L.EPI_QUICKGELU if kind == "gelu_f16" else L.EPI_NONE
    c, c16 = _gemm_f16(a16, w16, bias, res, epi, kind in ("f32", "f32_res"), kind in ("f16", "gelu_f16"), alpha=0.5)
    ref = _ref(a16, w16, bias, res, epi, alpha=0.5)
    if c is not None:
        err = (c.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, K / 768), f"f32 out: {err:.3e}"            # f32 accumulation of exact f16 products
    if c16 is not None:
        # one f16 rounding of the f32 result
        err = ((c16.double() - ref).abs() / (ref.abs() + 1.0)).max().item()
        assert err < 6e-4, f"f16 out: {err:.3e}"


def test_gemm_f16_eight_phase_kernel_is_bit_reproducible():
    """No atomics, no split K: two runs on the same operands are identical to the last bit (the mode's top-1 report depends on it)."""
    import hashlib
    M, N, K = 12608 * 2, 2304, 768
    torch.manual_seed(5)
    a = torch.randn(M, K, device=DEV).half()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    digests = []
    for _ in range(3):
        c, _ = _gemm_f16(a, w, None, None, L.EPI_NONE, True, False)
        digests.append(hashlib.sha256(c.cpu().numpy().tobytes()).hexdigest())
    assert digests[0] == digests[1] == digests[2]
