"""Host logic of the arithmetic switches (no GPU): which layers / widths take the split-bf16 kernels, what the environment selects."""
import pytest

from models_amd import ops


def test_scorer_arith_default_and_overrides(monkeypatch):
    monkeypatch.delenv("MERLIN_HIP_SCORER_ARITH", raising=False)
    assert ops.scorer_arith() == "bf16x6"  # fp32-grade: the default
    for v in ("f32", "bf16x3", "bf16x6"):
        monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", v)
        assert ops.scorer_arith() == v
    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "fp8")  # unknown values do not silently change the arithmetic class
    assert ops.scorer_arith() == "bf16x6"


def test_gemm_arith_default_and_overrides(monkeypatch):
    monkeypatch.delenv("MERLIN_HIP_GEMM_ARITH", raising=False)
    assert ops.gemm_arith() == "bf16x6"
    for v in ("f32", "bf16x3"):
        monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", v)
        assert ops.gemm_arith() == v


@pytest.mark.parametrize("M,K,N,want", [
    (65536, 3341, 512, True),    # DCN-v2 deep tower, first layer: wide in both directions
    (65536, 1024, 768, True),
    (65536, 512, 256, False),    # two-tower / DCN-v2 512 -> 256: the operand preparation costs more than the GEMM saves (measured)
    (65536, 600, 300, False),
    (512, 3341, 512, False),     # too few rows
    (65536, 415, 128, False),    # the tower kernel's shape, not the split GEMM's
])
def test_wide_dense_layers_that_take_the_split_gemm(monkeypatch, M, K, N, want):
    monkeypatch.delenv("MERLIN_HIP_GEMM_ARITH", raising=False)
    assert ops._linear_split_ok(M, K, N) is want
    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", "f32")
    assert ops._linear_split_ok(M, K, N) is False  # the exact chains everywhere


def test_topk_split_widths():
    assert ops.TopKSplit.supported(128) and ops.TopKSplit.supported(64)
    assert not ops.TopKSplit.supported(32) and not ops.TopKSplit.supported(96)
