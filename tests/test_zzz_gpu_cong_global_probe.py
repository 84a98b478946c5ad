"""`-m gpu`: csrc/cong_global.hpp -- the GLOBAL CoNgram model's mixture arithmetic -- evaluated on the MI355X over the window sections in HBM
(kamd_debug_cong_global) against the host evaluation of the same header, which tests/test_cong_global.py pins to the real reference: 24 000 random
queries, bit for bit.  New in round 3 and not yet run on hardware when committed, hence last in the collection order."""
import pytest

from test_cong_global import probe_device_arithmetic


@pytest.mark.gpu
def test_device_arithmetic_equals_the_oracle():
    assert probe_device_arithmetic(None) == 24000
