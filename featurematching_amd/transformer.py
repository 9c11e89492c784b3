"""Linear-attention context layers on PyTorch-ROCm - the layers either side of the matching hot path
(network/net.py:74 coarse, :79-80 fine; network/module/transformer.py:34-57,78-96;
network/module/attentions.py:19-46).

The coarse layers (8 x [N,4800,256]) belong to the feature side of the boundary and run on PyTorch-ROCm (SURVEY.md:
the CNN/FPN feature stack hands over to the HIP kernels).  The FINE layers in the reference's default configuration
(d_model 64, 8 heads, ['self', 'cross'], windows of 25 or 49 tokens) take the fused HIP kernel fm_fine_transformer
in inference (one wave per match, ~11x the PyTorch module at 640x480); training and every
other configuration use the torch ops below.  Parameter names and shapes equal the reference's, so a reference
state dict loads unchanged:

    layers.<k>.{q_proj,k_proj,v_proj,merge}.weight [d,d]   layers.<k>.mlp.{0,2}.weight [2d,2d] / [d,2d]
    layers.<k>.{norm1,norm2}.{weight,bias} [d]
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F


def linear_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """"Transformers are RNNs" attention with the elu(x)+1 feature map (attentions.py:19-46, no masks).
    q [N,L,H,D], k,v [N,S,H,D] -> [N,L,H,D].  Per head: out_l = phi(q_l) (sum_s phi(k_s) v_s^T) / (phi(q_l).sum_s phi(k_s))."""
    q = F.elu(q) + 1
    k = F.elu(k) + 1
    s = v.shape[1]
    kv = torch.einsum("nshd,nshv->nhdv", k, v / s)            # values are pre-divided by S, as in the reference
    z = 1.0 / (torch.einsum("nlhd,nhd->nlh", q, k.sum(dim=1)) + eps)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", q, kv, z) * s


class EncoderLayer(nn.Module):
    """x <- x + LN2(MLP([x, LN1(merge(attn(q(x), k(src), v(src))))]))   (transformer.py:34-57)"""

    def __init__(self, d_model: int, nhead: int):
        super().__init__()
        self.nhead, self.dim = nhead, d_model // nhead
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(2 * d_model, 2 * d_model, bias=False), nn.ReLU(True),
                                 nn.Linear(2 * d_model, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, x: torch.Tensor, source: torch.Tensor) -> torch.Tensor:
        n, l, _ = x.shape
        heads = lambda t: t.view(n, -1, self.nhead, self.dim)
        msg = linear_attention(heads(self.q_proj(x)), heads(self.k_proj(source)), heads(self.v_proj(source)))
        msg = self.norm1(self.merge(msg.reshape(n, l, -1)))
        msg = self.norm2(self.mlp(torch.cat([x, msg], dim=2)))
        return x + msg


class LocalFeatureTransformer(nn.Module):
    """Alternating self / cross layers over the two images' token sets (transformer.py:78-96).  config =
    {'d_model', 'nhead', 'layer_names': ['self', 'cross', ...], 'attention': 'linear'}."""

    def __init__(self, config):
        super().__init__()
        if config.get('attention', 'linear') != 'linear':
            raise NotImplementedError("only the reference's default linear attention is provided")
        self.d_model, self.layer_names = config['d_model'], list(config['layer_names'])
        self.layers = nn.ModuleList(EncoderLayer(config['d_model'], config['nhead']) for _ in self.layer_names)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def _hip_ok(self, feat0, feat1) -> bool:
        """fm_fine_transformer serves inference on float32 GPU windows with the default fine configuration;
        FM_HIP_FINE_TF=0 forces the torch ops."""
        return not self.training and not torch.is_grad_enabled() and feat0.is_cuda and feat0.dtype == torch.float32 \
            and self.d_model == 64 and self.layer_names == ['self', 'cross'] and self.layers[0].nhead == 8 \
            and feat0.shape == feat1.shape and feat0.shape[1] in (25, 49) \
            and os.environ.get("FM_HIP_FINE_TF", "1") != "0"

    def _packed(self, device):
        """operand fragments of the two layers, re-packed when a parameter changes (in-place updates bump torch's
        version counters)"""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (str(device),)
        if getattr(self, '_pack_key', None) != key:
            from . import ops
            self._pack_cache = ops.pack_fine_transformer(self.state_dict(), device)
            self._pack_key = key
        return self._pack_cache

    def forward(self, feat0: torch.Tensor, feat1: torch.Tensor):
        assert feat0.shape[2] == self.d_model, "the feature number of src and transformer must be equal"
        if self._hip_ok(feat0, feat1):
            from . import ops
            return ops.fine_transformer(feat0, feat1, self._packed(feat0.device))
        for layer, name in zip(self.layers, self.layer_names):
            if name == 'self':
                feat0, feat1 = layer(feat0, feat0), layer(feat1, feat1)
            elif name == 'cross':
                feat0 = layer(feat0, feat1)
                feat1 = layer(feat1, feat0)          # sees the UPDATED feat0, as in the reference (:93-94)
            else:
                raise KeyError(name)
        return feat0, feat1
