"""Host-side logic of the drop-in modules that needs no GPU: state-dict compatibility with the reference's `net`
(network/net.py:24-32,94-102) and the argument checks of the facade."""
import pytest
import torch
import torch.nn as nn

from featurematching_amd.matcher import DEFAULT_CONFIG, Matcher


class _NoBackbone(nn.Module):            # the feature extractor is out of scope: a parameter-free stand-in
    def forward(self, x):
        raise RuntimeError("not used")


def _reference_keys():
    """The key set of the reference's net(...).state_dict() without its backbone (network/net.py:24-32;
    transformer.py:14-27: bias-free q/k/v/merge/mlp + two LayerNorms per layer; fine_preprocess.py:21-22;
    fine_matching_new.py:18-19; the sinusoidal tables are non-persistent buffers)."""
    keys = {}
    for name, d, n in (("coarse", 256, 8), ("fine", 64, 2)):
        for k in range(n):
            for lin, shape in (("q_proj", (d, d)), ("k_proj", (d, d)), ("v_proj", (d, d)), ("merge", (d, d)),
                               ("mlp.0", (2 * d, 2 * d)), ("mlp.2", (d, 2 * d))):
                keys[f"{name}.layers.{k}.{lin}.weight"] = shape
            for ln in ("norm1", "norm2"):
                keys[f"{name}.layers.{k}.{ln}.weight"] = (d,)
                keys[f"{name}.layers.{k}.{ln}.bias"] = (d,)
    keys.update({"fine_preprocess.down_proj.weight": (64, 256), "fine_preprocess.down_proj.bias": (64,),
                 "fine_preprocess.merge_feat.weight": (64, 128), "fine_preprocess.merge_feat.bias": (64,),
                 "fine_matching.mix_feat_0.weight": (1, 49), "fine_matching.mix_feat_0.bias": (1,),
                 "fine_matching.mix_feat_1.weight": (1, 49), "fine_matching.mix_feat_1.bias": (1,)})
    return keys


def test_matcher_has_the_reference_key_set():
    m = Matcher(backbone=_NoBackbone())
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert ours == _reference_keys()


@pytest.mark.parametrize("prefix", ["", "matcher.", "matcher.loftr_"])
def test_strict_load_of_a_reference_shaped_checkpoint(prefix):
    """`net.load_state_dict` strips the Lightning module's `matcher.` prefix (and `loftr_`), net.py:94-102; a
    checkpoint written by a build that kept the position tables persistent carries `*.pos_encoding.pe` as well."""
    g = torch.Generator().manual_seed(0)
    sd = {prefix + k: torch.randn(*shape, generator=g) for k, shape in _reference_keys().items()}
    sd[prefix + "pos_encoding.pe"] = torch.zeros(1, 256, 4, 4)
    sd[prefix + "fine_matching.pos_encoding.pe"] = torch.zeros(1, 64, 4, 4)
    m = Matcher(backbone=_NoBackbone())
    res = m.load_state_dict(dict(sd), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    got = m.state_dict()
    for k in _reference_keys():
        assert torch.equal(got[k], sd[prefix + k]), k
    with pytest.raises(RuntimeError):          # strict means strict: an unknown key is still an error
        m.load_state_dict(dict(sd, **{prefix + "coarse.layers.9.q_proj.weight": torch.zeros(256, 256)}), strict=True)


def test_default_config_is_the_reference_s():
    """config.py:11-13,22,29-36,40 / demo/net_config.py:28-37"""
    c = DEFAULT_CONFIG
    assert c['fine_window_size'] == 7 and c['fine_concat_coarse_feat'] is True and tuple(c['resolution']) == (8, 2)
    assert c['coarse']['d_model'] == 256 and c['fine']['d_model'] == 64
    assert c['match_coarse']['thr'] == 0.2 and c['match_coarse']['border_rm'] == 2
    assert c['match_coarse']['dsmax_temperature'] == 0.1
