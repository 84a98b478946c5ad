"""`-m gpu`: every line of the eval_data files on the model the REAL KiwiBuilder built (tests/test_built_model.py: combining rules, default.dict,
typo.dict of the reference itself; tests/golden/eval_built_*.json are the built Kiwi's own answers) through the MI355X kernels -- tokens, positions,
word / sentence / line numbers, fp32 scores, typo costs.  The file sorts last on purpose (it is the newest fixture)."""
import pytest

from test_built_model import FILES, check_device


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_device_equals_the_built_reference(name):
    assert check_device(None, name) >= 33


@pytest.mark.gpu
def test_device_bakes_the_dictionary_of_the_built_model():
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from test_built_model import built_model_path, mask_unused
    dev = KiwiAmd(built_model_path())
    prod = dev.dump_dict()
    dev.close()
    assert mask_unused(prod) == mask_unused(oraclelib.OracleKiwi(built_model_path()).dump_dict())
