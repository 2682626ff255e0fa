"""CPU checks of csrc/ds_linear.hip's design (tools/linear_model.py): the 8-phase K-loop schedule has no read-after-write or
write-after-read hazard on its LDS half-tiles for any number of K-tiles, the swizzled LDS image is bank-conflict free for
the hardware's ds_read_b128 lane groups, and the whole index chain (DMA source swizzle -> LDS image -> fragment reads ->
MFMA 32x32x16 operand / accumulator layout -> permlane32_swap epilogue, shifted last row panel) reproduces x @ W.T."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import linear_model as lm  # noqa: E402


@pytest.mark.parametrize("nt", [2, 4, 6, 12, 16, 36, 64, 144])
def test_k_loop_schedule_has_no_hazard(nt):
    assert lm.check_schedule(nt) == []


def test_lds_image_is_bank_conflict_free():
    assert lm.bank_conflicts() == 0


def test_index_chain_reproduces_the_product():
    assert lm.check_indexing(M=300, N=256, K=128) < 1e-9


def test_convolution_gather_reproduces_a_direct_convolution():
    """border bits, zero line, tap-fastest K order, [out][tap][in] weight offsets, several images per tile, shifted last panel"""
    assert lm.check_conv_indexing(B=1, H=11, W=25, C=128, N=256) < 1e-9
    assert lm.check_conv_indexing(B=3, H=9, W=10, C=256, N=256, seed=2) < 1e-9
