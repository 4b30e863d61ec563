"""Drop-in check against the UNMODIFIED reference network code (only where /root/reference exists, i.e. in the
build container): with this repository ahead of the reference on sys.path, ``networks/ccnet.py`` (line 13:
``from cc_attention import CrissCrossAttention``) must pick up the MI355X module, RCCAModule must construct with
it, and the parameter names / shapes must equal those of the reference's own module so checkpoints interchange."""
import importlib
import os
import sys

import pytest
import torch

from conftest import ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")


@pytest.fixture()
def ref_networks():
    saved = list(sys.path)
    saved_mods = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("networks", "utils", "inplace_abn")}
    for k in saved_mods:
        del sys.modules[k]
    sys.path[:0] = [ROOT, REF]     # ours first (cc_attention AND the inplace_abn restatement), then the reference
    try:
        yield importlib.import_module("networks.ccnet")
    finally:
        sys.path[:] = saved
        for k in [k for k in sys.modules if k.split(".")[0] in ("networks", "utils", "inplace_abn")]:
            del sys.modules[k]
        sys.modules.update(saved_mods)


def test_unmodified_ccnet_imports_our_module(ref_networks):
    import ccnet_amd
    assert ref_networks.CrissCrossAttention is ccnet_amd.CrissCrossAttention
    head = ref_networks.RCCAModule(2048, 512, 19)            # networks/ccnet.py:99-123, as ResNet builds it (:147)
    assert isinstance(head.cca, ccnet_amd.CrissCrossAttention)
    assert head.cca.query_conv.out_channels == 64 and head.cca.value_conv.out_channels == 512


def test_state_dict_interchanges_with_reference_module(ref_networks):
    spec = importlib.util.spec_from_file_location("ref_functions", os.path.join(REF, "cc_attention", "functions.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    ours = ref_networks.RCCAModule(2048, 512, 19).cca
    theirs = ref.CrissCrossAttention(512)
    sd_ref = theirs.state_dict()
    assert {k: tuple(v.shape) for k, v in ours.state_dict().items()} == {k: tuple(v.shape) for k, v in sd_ref.items()}
    res = ours.load_state_dict(sd_ref, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in sd_ref.items():
        assert torch.equal(ours.state_dict()[k], v)
