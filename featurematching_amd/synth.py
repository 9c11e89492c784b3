"""Portable synthetic inputs for the matcher hot path.

The same seeded tensors must be reproducible bit-for-bit on the build container
and on the GPU box, independent of torch's RNG implementation, so everything
here is a counter-based integer hash (splitmix64) -> uniform -> Box-Muller in
float64 numpy, cast to float32 at the end.  Distributions follow SURVEY.md
section 8(d):

* ``peaky``      F0 = 4*z0,  F1 = F0[perm] + 0.4*z1   (throughput; M ~ 0.79 L)
* ``borderline`` F0 = 1*z0,  F1 = F0[perm] + 1.0*z1   (parity stress around thr)
* fine maps      N(0,1) [N, Cf, H/2, W/2]
* mix weights    U(-1/W, 1/W)  (torch ``nn.Linear(WW, 1)`` default bound 1/sqrt(WW))
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + _GOLD).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream_key(seed: int, stream: int) -> np.uint64:
    k = _splitmix64(np.array([seed], dtype=np.uint64))
    with np.errstate(over="ignore"):
        k = _splitmix64(k + np.uint64(stream) * _M1)
    return k[0]


def hash_u64(seed: int, stream: int, n: int) -> np.ndarray:
    """n 64-bit hashes of the counters 0..n-1 under (seed, stream)."""
    key = _stream_key(seed, stream)
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64(ctr * _GOLD + key)


def uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """float64 uniforms in (0, 1)."""
    h = hash_u64(seed, stream, n)
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, shape) -> np.ndarray:
    """float32 standard normals of the given shape (Box-Muller on hashed uniforms)."""
    n = int(np.prod(shape))
    half = (n + 1) // 2
    u1 = uniform(seed, 2 * stream, half)
    u2 = uniform(seed, 2 * stream + 1, half)
    r = np.sqrt(-2.0 * np.log(u1))
    t = 2.0 * np.pi * u2
    z = np.concatenate([r * np.cos(t), r * np.sin(t)])[:n]
    return z.astype(np.float32).reshape(shape)


def permutation(seed: int, stream: int, n: int) -> np.ndarray:
    """A seeded permutation of 0..n-1 (stable argsort of hashes)."""
    return np.argsort(hash_u64(seed, stream, n), kind="stable").astype(np.int64)


# name -> (gain, noise); "mixed" = "peaky" with a fraction of near-zero cells (textureless regions: descriptors
# scaled by MIXED_SCALE) chosen independently in both images, so some partners of good cells are missing and the
# affected rows / columns have flat similarity
DISTRIBUTIONS = {"peaky": (4.0, 0.4), "borderline": (1.0, 1.0), "mixed": (4.0, 0.4)}
MIXED_FRACTION, MIXED_SCALE = 0.2, 1e-3


def coarse_descriptors(seed: int, n: int, l: int, c: int, dist: str = "peaky"):
    """(feat_c0, feat_c1) float32 [n, l, c] numpy arrays; sample b uses seed + b."""
    g, sigma = DISTRIBUTIONS[dist]
    f0 = np.empty((n, l, c), np.float32)
    f1 = np.empty((n, l, c), np.float32)
    for b in range(n):
        z0 = normal(seed + b, 1, (l, c))
        z1 = normal(seed + b, 2, (l, c))
        perm = permutation(seed + b, 3, l)
        f0[b] = g * z0
        f1[b] = f0[b][perm] + np.float32(sigma) * z1
        if dist == "mixed":
            f0[b][uniform(seed + b, 6, l) < MIXED_FRACTION] *= np.float32(MIXED_SCALE)
            f1[b][uniform(seed + b, 7, l) < MIXED_FRACTION] *= np.float32(MIXED_SCALE)
    return f0, f1


def fine_maps(seed: int, n: int, cf: int, hf: int, wf: int):
    """(feat_f0, feat_f1) float32 [n, cf, hf, wf] (NCHW)."""
    f0 = np.stack([normal(seed + b, 4, (cf, hf, wf)) for b in range(n)])
    f1 = np.stack([normal(seed + b, 5, (cf, hf, wf)) for b in range(n)])
    return f0, f1


def mix_weights(seed: int, ww: int):
    """(w0[ww], b0, w1[ww], b1) float32, U(-1/sqrt(ww), 1/sqrt(ww))."""
    bound = 1.0 / np.sqrt(float(ww))
    u = uniform(seed, 12, 2 * ww + 2)
    v = ((2.0 * u - 1.0) * bound).astype(np.float32)
    return v[:ww].copy(), np.float32(v[ww]), v[ww + 1:2 * ww + 1].copy(), np.float32(v[2 * ww + 1])


# The five BASELINE.json configurations, as (name -> dict) used by tests and bench.
def merge_weights(seed: int, c: int, cf: int):
    """Weights of FinePreprocess's two Linear layers (fine_preprocess.py:25-26), torch's default init range:
    (down_w [cf, c], down_b [cf], merge_w [cf, 2*cf], merge_b [cf]) float32, U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    def lin(stream, fan_out, fan_in):
        bound = 1.0 / np.sqrt(float(fan_in))
        u = uniform(seed, stream, fan_out * fan_in + fan_out)
        w = ((2.0 * u[:fan_out * fan_in] - 1.0) * bound).astype(np.float32).reshape(fan_out, fan_in)
        b = ((2.0 * u[fan_out * fan_in:] - 1.0) * bound).astype(np.float32)
        return w, b
    dw, db = lin(21, cf, c)
    mw, mb = lin(22, cf, 2 * cf)
    return dw, db, mw, mb


CONFIGS = {
    "cfg1": dict(n=1, h=128, w=128, c=64, cf=64, seed=0),
    "cfg2": dict(n=1, h=480, w=640, c=256, cf=64, seed=1),
    "cfg3": dict(n=64, h=480, w=640, c=256, cf=64, seed=2),
    "cfg5": dict(n=1, h=1024, w=1024, c=256, cf=64, seed=5),
    # the 9600 x 9600 cost volume BASELINE.json's config 5 mentions in passing: 640 x 960 -> 80 x 120 cells
    "l9600": dict(n=1, h=640, w=960, c=256, cf=64, seed=6),
}


def config_shapes(cfg: dict):
    """Derived sizes for an image of h x w at resolutions (8, 2)."""
    hc, wc = cfg["h"] // 8, cfg["w"] // 8
    hf, wf = cfg["h"] // 2, cfg["w"] // 2
    return dict(hc=hc, wc=wc, hf=hf, wf=wf, l=hc * wc)


def transformer_weights(seed: int, d_model: int, n_layers: int) -> dict:
    """Seeded weights of a LocalFeatureTransformer (network/module/transformer.py) keyed by its state-dict names,
    from the portable hash RNG: Linear weights uniform with the xavier bound sqrt(6 / (fan_in + fan_out)) (the
    reference's own initialisation), LayerNorm affine terms away from (1, 0) so that they are exercised."""
    out = {}
    stream = 100

    def lin(name, fan_out, fan_in):
        nonlocal stream
        bound = (6.0 / (fan_in + fan_out)) ** 0.5
        u = uniform(seed, stream, fan_out * fan_in).reshape(fan_out, fan_in)
        stream += 1
        out[name] = ((2.0 * u - 1.0) * bound).astype(np.float32)

    for k in range(n_layers):
        for nme in ("q_proj", "k_proj", "v_proj", "merge"):
            lin(f"layers.{k}.{nme}.weight", d_model, d_model)
        lin(f"layers.{k}.mlp.0.weight", 2 * d_model, 2 * d_model)
        lin(f"layers.{k}.mlp.2.weight", d_model, 2 * d_model)
        for nme in ("norm1", "norm2"):
            out[f"layers.{k}.{nme}.weight"] = (1.0 + 0.1 * normal(seed, stream, (d_model,))).astype(np.float32)
            out[f"layers.{k}.{nme}.bias"] = (0.1 * normal(seed, stream + 1, (d_model,))).astype(np.float32)
            stream += 2
    return out
