import sys, numpy as np, torch
sys.path.insert(0, '.')
from featurematching_amd import ops, synth
from oracle import matcher_ref as orc
DEV = 'cuda:0'
for name, args in (("borderline", (91, 2, 23 * 31, 256, "borderline", (23, 31), (184, 248))),
                   ("peaky47", (47, 1, 600, 128, "peaky", (20, 30), (160, 240)))):
    seed, n, l, c, dist, hw_c, hw_i = args
    f0, f1 = synth.coarse_descriptors(seed, n, l, c, dist)
    if name == "peaky47":
        f0[:, ::3] *= 1e-4; f1[:, 1::3] *= 1e-4
    ref = orc.coarse_match(f0, f1, hw_i, hw_c, hw_c, 0.2, 2, 0.1, return_conf=True)
    out = ops.coarse_match(torch.as_tensor(f0, device=DEV), torch.as_tensor(f1, device=DEV), hw_c, hw_c, 8.0, conf_matrix=True)
    got = out['conf_matrix'].cpu().numpy(); rc = np.asarray(ref['conf_matrix'])
    err = np.abs(got - rc)
    print(name, "max err", err.max())
    idx = np.argsort(err.ravel())[::-1][:8]
    for t in idx:
        b, i, j = np.unravel_index(t, err.shape)
        sim = float(f0[b, i].astype(np.float64) @ f1[b, j].astype(np.float64)) / (c * 0.1)
        print("  ", b, i, j, "err %.3g" % err[b, i, j], "ref %.6f got %.6f" % (rc[b, i, j], got[b, i, j]), "sim %.2f" % sim)
    for lo, hi in ((0.1, 2), (0.01, 0.1), (0, 0.01)):
        m = (rc > lo) & (rc <= hi)
        print("   conf in (%g,%g]: n=%d max err %.3g" % (lo, hi, m.sum(), err[m].max() if m.any() else 0))
