"""Linear-attention context layers - the layers either side of the matching hot path
(network/net.py:74 coarse, :79-80 fine; network/module/transformer.py:34-57,78-96;
network/module/attentions.py:19-46).

In eval mode both of the reference's default configurations run as fused HIP kernels: the FINE
layers (d_model 64, 8 heads, ['self', 'cross'], windows of 25 or 49 tokens) as fm_fine_transformer - one wave per
match - and the COARSE layers (d_model 256, 8 heads, any self / cross sequence, 8 x [N,4800,256] in the reference)
as fm_coarse_transformer - three launches per encoder layer, hi/lo-split float16 products on the matrix cores (22
significant bits).  That holds under the reference's own inference call, `matcher.eval()(data)` with grad mode on
(demo/demo.py:105-108).  The torch ops below are the trainable definition: training mode, inputs that require grad,
other configurations (the reference's 'full' attention among them), padding masks, CPU tensors.  Parameter names and shapes equal the reference's, so a reference state dict
loads unchanged:

    layers.<k>.{q_proj,k_proj,v_proj,merge}.weight [d,d]   layers.<k>.mlp.{0,2}.weight [2d,2d] / [d,2d]
    layers.<k>.{norm1,norm2}.{weight,bias} [d]
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def linear_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, eps: float = 1e-6, q_mask=None,
                     kv_mask=None) -> torch.Tensor:
    """"Transformers are RNNs" attention with the elu(x)+1 feature map (attentions.py:19-46).
    q [N,L,H,D], k,v [N,S,H,D] -> [N,L,H,D].  Per head: out_l = phi(q_l) (sum_s phi(k_s) v_s^T) / (phi(q_l).sum_s phi(k_s)).
    q_mask [N,L] / kv_mask [N,S] zero the padded positions of phi(q) / of phi(k) and v (:35-40); the values are divided
    by the PADDED length S either way, as in the reference."""
    q = F.elu(q) + 1
    k = F.elu(k) + 1
    if q_mask is not None:
        q = q * q_mask[:, :, None, None]
    if kv_mask is not None:
        k = k * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    s = v.shape[1]
    kv = torch.einsum("nshd,nshv->nhdv", k, v / s)            # values are pre-divided by S, as in the reference
    z = 1.0 / (torch.einsum("nlhd,nhd->nlh", q, k.sum(dim=1)) + eps)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", q, kv, z) * s


def full_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_mask=None, kv_mask=None) -> torch.Tensor:
    """Scaled dot-product attention, the reference's other option (attentions.py:54-79, no dropout: its default).
    q [N,L,H,D], k,v [N,S,H,D] -> [N,L,H,D]: softmax over the source positions of q.k / sqrt(D); with a kv_mask the
    pairs outside q_mask x kv_mask are filled with -inf first (:71-72: a fully padded query row comes out as NaN there
    too).  Materialises the [N,L,S,H] scores as the reference does: an option for short sequences."""
    qk = torch.einsum("nlhd,nshd->nlsh", q, k)
    if kv_mask is not None:
        qm = torch.ones(q.shape[:2], dtype=torch.bool, device=q.device) if q_mask is None else q_mask.bool()
        qk = qk.masked_fill(~(qm[:, :, None, None] & kv_mask.bool()[:, None, :, None]), float('-inf'))
    a = torch.softmax(qk / q.shape[3] ** .5, dim=2)
    return torch.einsum("nlsh,nshd->nlhd", a, v)


class EncoderLayer(nn.Module):
    """x <- x + LN2(MLP([x, LN1(merge(attn(q(x), k(src), v(src))))]))   (transformer.py:34-57); attention = 'linear'
    (the default) or 'full' (:22)"""

    def __init__(self, d_model: int, nhead: int, attention: str = 'linear'):
        super().__init__()
        self.attention = attention
        self.nhead, self.dim = nhead, d_model // nhead
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(2 * d_model, 2 * d_model, bias=False), nn.ReLU(True),
                                 nn.Linear(2 * d_model, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, x: torch.Tensor, source: torch.Tensor, x_mask=None, source_mask=None) -> torch.Tensor:
        n, l, _ = x.shape
        heads = lambda t: t.view(n, -1, self.nhead, self.dim)
        attn = linear_attention if self.attention == 'linear' else full_attention
        msg = attn(heads(self.q_proj(x)), heads(self.k_proj(source)), heads(self.v_proj(source)),
                   q_mask=x_mask, kv_mask=source_mask)
        msg = self.norm1(self.merge(msg.reshape(n, l, -1)))
        msg = self.norm2(self.mlp(torch.cat([x, msg], dim=2)))
        return x + msg


class LocalFeatureTransformer(nn.Module):
    """Alternating self / cross layers over the two images' token sets (transformer.py:78-96).  config =
    {'d_model', 'nhead', 'layer_names': ['self', 'cross', ...], 'attention': 'linear'}.  use_hip=False keeps the torch ops
    in eval mode too (tools that time one against the other).  The fine kernel splits its operands into float16 halves
    at a per-match power-of-two scale that it lowers itself when a match needs it (see fmatch.h); what does not fit at
    its smallest scale either (|activation| ~ 1e6, NaN / Inf, a weight >= 16) it reports: with check_range (default) the
    module reads that report (one host sync per call) and redoes such a call with the float32 torch layers.  The read is
    skipped while a hipGraph is being captured (a capture cannot synchronise the host) and with check_range=False
    (callers that know their value ranges); the kernel's report then stays in `last_status` (an int32 device tensor the
    caller may read when it synchronises anyway: bit _lib.FM_DEV_RANGE = discard the outputs).  `range_fallbacks`
    counts the calls that were redone.
    inference_only=True (default) lets the eval-mode HIP path run even when the layers' own parameters require grad -
    the outputs then carry no graph through the parameters (a warning says so once); False sends such calls through
    the torch layers, as a fine-tuning run that freezes batch norm with .eval() needs."""

    _warned_detached = False

    def __init__(self, config, use_hip: bool = True, check_range: bool = True, inference_only: bool = True):
        super().__init__()
        self.use_hip, self.check_range, self.inference_only = use_hip, check_range, inference_only
        self.range_fallbacks = 0
        self.last_status = None
        self.attention = config.get('attention', 'linear')
        if self.attention not in ('linear', 'full'):
            raise ValueError(f"attention must be 'linear' or 'full' (transformer.py:22), got {self.attention!r}")
        self.d_model, self.layer_names = config['d_model'], list(config['layer_names'])
        self.layers = nn.ModuleList(EncoderLayer(config['d_model'], config['nhead'], self.attention) for _ in self.layer_names)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def _hip_kind(self, feat0, feat1):
        """Which fused HIP kernel serves this call: 'fine' = fm_fine_transformer (d_model 64, 8 heads, ['self',
        'cross'], windows of 25 or 49 tokens), 'coarse' = fm_coarse_transformer (d_model 256, 8 heads, any self /
        cross sequence), None = the torch ops below.  Eval mode on float32 GPU tensors that do not ask for a gradient;
        grad MODE alone does not matter (the reference's demo calls the eval-mode matcher without no_grad)."""
        wants_grad = torch.is_grad_enabled() and (feat0.requires_grad or feat1.requires_grad)
        if not wants_grad and not self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # eval mode, grad mode on, trainable parameters, inputs without grad (a frozen backbone under no_grad, or
            # .eval() used to freeze batch norm while fine-tuning): the reference would build a graph through the
            # layers' parameters here; the HIP kernels do not
            if not self.inference_only:
                wants_grad = True
            elif self.use_hip and feat0.is_cuda and not LocalFeatureTransformer._warned_detached:
                LocalFeatureTransformer._warned_detached = True      # (once per process)
                import warnings
                warnings.warn("LocalFeatureTransformer: eval-mode HIP kernels return tensors without a graph through the "
                              "layers' parameters; pass inference_only=False (or call .train()) to fine-tune them")
        if self.attention != 'linear':               # 'full' attention (attentions.py:54-79): the torch layers
            return None
        if not self.use_hip or self.training or wants_grad or not feat0.is_cuda or feat0.dtype != torch.float32 \
                or feat1.dtype != torch.float32 or self.layers[0].nhead != 8 or feat0.shape[0] != feat1.shape[0]:
            return None
        if self.d_model == 64 and self.layer_names == ['self', 'cross'] and feat0.shape == feat1.shape \
                and feat0.shape[1] in (25, 49):
            return 'fine'
        if self.d_model == 256 and all(k in ('self', 'cross') for k in self.layer_names) and feat0.shape[1] > 0 \
                and feat1.shape[1] > 0:
            return 'coarse'
        return None

    def _packed(self, device, kind):
        """operand fragments of the layers, re-packed when a parameter changes (in-place updates bump torch's
        version counters)"""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (str(device), kind)
        if getattr(self, '_pack_key', None) != key:
            from . import ops
            if kind == 'fine':
                self._pack_cache = ops.pack_fine_transformer(self.state_dict(), device)
            else:
                self._pack_cache = ops.pack_coarse_transformer(self.state_dict(), len(self.layer_names), device)
            self._pack_key = key
        return self._pack_cache

    def forward(self, feat0: torch.Tensor, feat1: torch.Tensor, mask0=None, mask1=None):
        """network/module/transformer.py:78 (`net.forward` passes mask0 = mask1 = None, net.py:73-74)"""
        assert feat0.shape[2] == self.d_model, "the feature number of src and transformer must be equal"
        self._pending_check = None
        kind = self._hip_kind(feat0, feat1)
        if (mask0 is not None or mask1 is not None) and kind != 'coarse':
            # padding masks (transformer.py:89-95, attentions.py:35-40; `net.forward` passes None, net.py:73-74): the coarse
            # kernels take them, every other configuration gets the torch layers
            return self._torch_layers(feat0, feat1, mask0, mask1)
        if kind is not None:
            from . import _lib, ops
            with torch.no_grad():
                packed = self._packed(feat0.device, kind)
                if kind == 'coarse':
                    return ops.coarse_transformer(feat0, feat1, packed, self.layer_names, mask0=mask0, mask1=mask1)
                # [0]: FM_DEV_RANGE report; [1]: by how much the matches went below the first attempt's activation scale
                status = torch.zeros(2, dtype=torch.int32, device=feat0.device)
                self._fine_calls = getattr(self, '_fine_calls', 0) + 1
                start = getattr(self, '_fine_start', 8)
                if self._fine_calls % 64 == 0:
                    start = 8                         # re-probe: the data may have calmed down
                out = ops.fine_transformer(feat0, feat1, packed, status=status, start_scale=start)
                self.last_status = status
                if not self.check_range or torch.cuda.is_current_stream_capturing():
                    return out
                if getattr(self, 'defer_range_check', False):
                    # the caller enqueues what consumes `out` first and asks afterwards (resolve_range_check): the host
                    # sync then waits behind work the GPU has, not in front of it
                    self._pending_check = (status, start, feat0, feat1)
                    return out
                flag, lowered = status.tolist()       # (the one host sync of this call)
                # matches that had to lower the scale repeat their passes inside the kernel: start the next call where
                # this one ended (0.74 -> 0.45 ms at 3769 windows of a network whose activations leave 2^8)
                self._fine_start = max(-4, start - int(lowered))
                if not (int(flag) & _lib.FM_DEV_RANGE):
                    return out
                self.range_fallbacks += 1          # values beyond the kernel's float16 operand scales: float32 layers
                return self._torch_layers(feat0, feat1)
        return self._torch_layers(feat0, feat1)

    def resolve_range_check(self):
        """With `defer_range_check` set: read the report of the last fine-kernel call now.  Returns None when its outputs
        stand, or the (feat0, feat1) of the float32 layers when the kernel could not hold the values (the caller redoes
        what consumed the outputs)."""
        pend, self._pending_check = getattr(self, '_pending_check', None), None
        if pend is None:
            return None
        from . import _lib
        status, start, feat0, feat1 = pend
        flag, lowered = status.tolist()
        self._fine_start = max(-4, start - int(lowered))
        if not (int(flag) & _lib.FM_DEV_RANGE):
            return None
        self.range_fallbacks += 1
        with torch.no_grad():
            return self._torch_layers(feat0, feat1)

    def _torch_layers(self, feat0, feat1, mask0=None, mask1=None):
        m0 = None if mask0 is None else mask0.to(feat0.dtype)
        m1 = None if mask1 is None else mask1.to(feat1.dtype)
        for layer, name in zip(self.layers, self.layer_names):
            if name == 'self':
                feat0, feat1 = layer(feat0, feat0, m0, m0), layer(feat1, feat1, m1, m1)
            elif name == 'cross':
                feat0 = layer(feat0, feat1, m0, m1)
                feat1 = layer(feat1, feat0, m1, m0)  # sees the UPDATED feat0, as in the reference (:93-94)
            else:
                raise KeyError(name)
        return feat0, feat1
