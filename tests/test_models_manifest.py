"""The build's models.py must reproduce the reference model files' state-dict layout (keys, shapes,
parameter counts) so released checkpoints load (lib/utils.py:17-45).  Fixture: tests/golden/
models_manifest.json, captured from /root/reference/models via tests/golden/make_fixtures.py."""
import json
import os

import pytest

from helpers import Cfg
from languagegroundedsemseg_amd.models import load_model

MAN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "models_manifest.json")))


@pytest.mark.parametrize("name", sorted(MAN))
def test_state_dict_matches_reference(name):
    m = load_model(name)(3, 200, Cfg())
    sd = m.state_dict()
    ref = MAN[name]["state_dict"]
    assert [k for k, _ in ref] == list(sd.keys())
    for k, shape in ref:
        assert list(sd[k].shape) == shape, k
    assert sum(p.numel() for p in m.parameters()) == MAN[name]["num_parameters"]


def test_survey_parameter_counts():
    assert MAN["Res16UNet34C"]["num_parameters"] == 37864104
    assert MAN["Res16UNet14A"]["num_parameters"] == 8032552
    assert MAN["Res16UNet34D"]["num_parameters"] == 79429800
