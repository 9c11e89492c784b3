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


def test_context_layers_with_padding_masks_match_the_reference():
    """network/module/transformer.py:78-96 with mask0 / mask1 (attentions.py:35-40 zeroes the padded positions of
    phi(q), phi(k) and v): the module's torch layers - the route a masked call takes - against outputs of the
    reference's own LocalFeatureTransformer (tests/golden/make_golden.py:masked_transformer_case)."""
    import numpy as np
    import torch
    from featurematching_amd import synth
    from featurematching_amd.transformer import LocalFeatureTransformer
    from helpers import load_golden
    g = load_golden("tf_masked_small")
    seed, n, l, s, d = [int(v) for v in g['meta']]
    tf = LocalFeatureTransformer(dict(d_model=d, nhead=8, layer_names=['self', 'cross'], attention='linear')).eval()
    tf.load_state_dict({k: torch.as_tensor(v) for k, v in synth.transformer_weights(seed, d, 2).items()})
    x0, x1 = torch.as_tensor(synth.normal(seed, 1, (n, l, d))), torch.as_tensor(synth.normal(seed, 2, (n, s, d)))
    with torch.no_grad():
        y0, y1 = tf(x0, x1, torch.as_tensor(g['mask0']), torch.as_tensor(g['mask1']))
        z0, _ = tf(x0, x1)
    np.testing.assert_allclose(y0.numpy(), g['out0'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(y1.numpy(), g['out1'], rtol=0, atol=2e-5)
    assert (y0 - z0).abs().max() > 1e-2          # the masks matter


def test_masked_coarse_configuration_matches_reference_fixture():
    """The torch definition of the coarse configuration (d_model 256, four layers) with padding masks against the
    REFERENCE's outputs (tf_masked_coarse: the fixture the HIP kernels are checked against on the GPU)."""
    import numpy as np
    import torch
    from featurematching_amd import synth
    from featurematching_amd.transformer import LocalFeatureTransformer
    from helpers import load_golden
    g = load_golden("tf_masked_coarse")
    seed, n, l, s, d = [int(v) for v in g['meta']]
    layers = ['self', 'cross', 'self', 'cross']
    tf = LocalFeatureTransformer(dict(d_model=d, nhead=8, layer_names=layers, attention='linear')).eval()
    tf.load_state_dict({k: torch.as_tensor(v) for k, v in synth.transformer_weights(seed, d, len(layers)).items()})
    x0 = torch.as_tensor((2.0 * synth.normal(seed, 1, (n, l, d))).astype(np.float32))
    x1 = torch.as_tensor((2.0 * synth.normal(seed, 2, (n, s, d))).astype(np.float32))
    with torch.no_grad():
        y0, y1 = tf(x0, x1, torch.as_tensor(g['mask0']), torch.as_tensor(g['mask1']))
    np.testing.assert_allclose(y0.numpy(), g['out0'], rtol=0, atol=3e-5)
    np.testing.assert_allclose(y1.numpy(), g['out1'], rtol=0, atol=3e-5)


def test_full_attention_option_matches_reference_fixture():
    """attention='full' (network/module/attentions.py:54-79, transformer.py:22): the torch layers against the outputs of the
    REFERENCE's LocalFeatureTransformer(attention='full') on the same seeded inputs and weights, without masks and with
    padding masks (a fully padded query row is NaN in the reference - softmax over -inf - and here)."""
    import numpy as np
    import torch
    from featurematching_amd import synth
    from featurematching_amd.transformer import LocalFeatureTransformer
    from helpers import load_golden
    g = load_golden("tf_full_small")
    seed, n, l, s, d = [int(v) for v in g['meta']]
    tf = LocalFeatureTransformer(dict(d_model=d, nhead=8, layer_names=['self', 'cross'], attention='full')).eval()
    tf.load_state_dict({k: torch.as_tensor(v) for k, v in synth.transformer_weights(seed, d, 2).items()})
    x0, x1 = torch.as_tensor(synth.normal(seed, 1, (n, l, d))), torch.as_tensor(synth.normal(seed, 2, (n, s, d)))
    with torch.no_grad():
        y0, y1 = tf(x0, x1)
        z0, z1 = tf(x0, x1, torch.as_tensor(g['mask0']), torch.as_tensor(g['mask1']))
    np.testing.assert_allclose(y0.numpy(), g['out0'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(y1.numpy(), g['out1'], rtol=0, atol=2e-5)
    assert np.array_equal(np.isnan(z0.numpy()), np.isnan(g['mout0'])) and np.isnan(g['mout0']).any()
    np.testing.assert_allclose(np.nan_to_num(z0.numpy()), np.nan_to_num(g['mout0']), rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.nan_to_num(z1.numpy()), np.nan_to_num(g['mout1']), rtol=0, atol=2e-5)
    assert tf._hip_kind(x0, x1) is None              # never the linear-attention kernels
    import pytest
    with pytest.raises(ValueError):
        LocalFeatureTransformer(dict(d_model=d, nhead=8, layer_names=['self'], attention='softmax'))


def test_hint_memory_is_bounded_decays_and_forgets():
    """ops.HintMemory: the hint word of fm_coarse_match_auto per problem kind - LRU-bounded, passing no hint on every
    k-th call (a probe of the common path), forgetting a key whose probe succeeded, behind a lock (no GPU involved)."""
    from featurematching_amd import _lib
    from featurematching_amd.ops import HintMemory
    m = HintMemory(capacity=2, reprobe=3)
    assert m.start('a') == (0, False)
    dense, exact, flat, step = _lib.FM_MODE_DENSE, _lib.FM_MODE_EXACT_SCREENING, _lib.FM_MODE_FLAT, _lib.FM_MODE_EXACT_STEP
    m.finish('a', dense); m.finish('b', exact | dense | (3 << 24)); m.finish('c', dense)
    assert set(m.snapshot()) == {'b', 'c'}                       # 'a' was the least recently used
    assert m.snapshot()['b']['hint'] == exact | dense            # (the attempts byte is per call, not remembered)
    assert m.start('b') == (exact | dense, False) and m.start('b') == (exact | dense, False)
    assert m.start('b') == (0, True)                             # third remembered call: probe the common path
    m.finish('b', 0)                                             # ... which served it: forgotten
    assert m.start('b') == (0, False) and 'b' not in m.snapshot()
    m.finish('c', dense | flat | step | (16 << 8))
    d = m.snapshot()['c']
    assert d['dense'] and d['flat'] and d['step'] and d['wide'] and d['slots'] == 16 and not d['exact']
    m.finish('c', dense)
    d = m.snapshot()['c']
    assert d['dense'] and not d['flat'] and not d['wide']
    m.clear()
    assert not m.snapshot()
