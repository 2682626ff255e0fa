"""The tile-level numpy model of k_attention_fwd3's control flow (tools/emulate_attention_skew.py) against softmax attention.

Not a test of a kernel: generation 3 of the attention kernel (S of tile t + 1 beside the softmax of tile t, K staged one tile
ahead of V^T) exists only in -DDS_EXPERIMENTS builds (on hardware it reproduced generation 2 bit for bit at the benchmark shape
and was not faster: profiles/round4_attention_gen3_ab.txt).  The model keeps its bookkeeping honest
-- which tile sits in which LDS buffer when, the two alternating accumulator sets, the peeled masked iteration -- and flags a
buffer that is written in the barrier interval in which it is read (that check found the missing barrier behind the prologue)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("with_bias", [False, True])
def test_skewed_attention_schedule_matches_softmax_attention(with_bias):
    import emulate_attention_skew as eas
    # 1 .. 6 tiles, odd and even counts, full and ragged last tiles, a single valid key, the benchmark's 1025 keys
    for n_valid in (1, 17, 64, 65, 128, 129, 191, 192, 193, 256, 257, 320, 384, 1025):
        err = eas.check(n_valid, with_bias, seed=n_valid)
        assert err < (1.5e-3 if with_bias else 3e-4), (n_valid, with_bias, err)


def test_the_model_flags_a_missing_barrier():
    """the hazard check itself: without the barrier behind the prologue's S(0), iteration 0 overwrites K(0) while it is read"""
    import emulate_attention_skew as eas
    lds = eas.Lds()
    lds.stash("K", 0, 0, "k0")
    lds.barrier()
    assert lds.read("K", 0, 0) == "k0"
    with pytest.raises(AssertionError):
        lds.stash("K", 0, 2, "k2")
    lds.barrier()
    lds.stash("K", 0, 2, "k2")
    with pytest.raises(AssertionError):
        lds.read("K", 0, 0)                 # the buffer no longer holds the tile the reader means
