"""-m gpu, runs LAST (file order): with MH_ALLOC_CHECK=1 in the environment the device-allocation cache poisons every block
that goes back to it behind its last use and verifies it when it is handed out again (mimosa_amd/csrc/mh_internal.hpp); a word
that was overwritten in between was written by work that was not ordered in front of the free — the hand-over rule
("everything on the block was enqueued on the freeing stream, or waited for") broken somewhere.  After the whole suite:
blocks were verified, none was overwritten.  Without the variable the test is skipped (the product runs unchecked)."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("MH_ALLOC_CHECK", "0") in ("", "0"), reason="MH_ALLOC_CHECK is not set")
def test_allocation_cache_hand_over_rule_held_for_the_whole_run():
    from mimosa_amd import capi
    st = capi.alloc_check_stats()
    assert st is not None
    verified, overwritten = st
    print(f"MH_ALLOC_CHECK: {verified} cached blocks verified at hand-out, {overwritten} overwritten words")
    assert verified > 0
    assert overwritten == 0
