#!/usr/bin/env python3
"""Bring-up diagnostics for the coarse stage on a real GPU: runs fm_coarse_match on small
seeded cases and compares every intermediate statistic in the workspace (float16 planes,
pass-A maxima, stabilisers, pass-B sums, candidate lists) with a float64 numpy computation,
then the final matches with the CPU oracle.  Prints a report; exits non-zero on mismatch.

    python tools/gpu_bringup.py [--case small|cfg1|cfg2] [--dist peaky|borderline]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from featurematching_amd import _lib, ops, synth  # noqa: E402
from oracle import matcher_ref as orc  # noqa: E402  (checker only)

NAMES = ["cand_count", "ccand_count", "scalars", "blocktot", "hi0", "lo0", "hi1", "lo1", "q0", "q1", "sigimg",
         "l1_0", "rowS", "colS", "rowB", "colB", "nmr", "nmc", "rsum", "csum", "cand_j", "cand_x", "ccand_i",
         "umax", "dense_cnt", "rowmax_u", "colmax_u", "splits_s", "units_s", "total"]


def layout(n, l, s, c, slots):
    lib = _lib.load()
    arr = (C.c_int64 * 40)()
    _lib.check(lib.fm_debug_coarse_layout(n, l, s, c, slots, arr, 40), "layout")
    v = list(arr)
    d = dict(zip(["N", "L", "S", "C", "Lp", "Sp", "panels", "tiles", "splits", "slots"], v[:10]))
    d.update(dict(zip(NAMES, v[10:])))
    return d


def view(ws, base_off, off, count, dtype):
    nbytes = count * torch.empty(0, dtype=dtype).element_size()
    return ws[base_off + off: base_off + off + nbytes].view(dtype).cpu().numpy()


def run(f0, f1, hw_c0, hw_c1, thr=0.2, border=2, temp=0.1, label=""):
    ok = True
    n, l, c = f0.shape
    s = f1.shape[1]
    dev = torch.device("cuda:0")
    t0, t1 = torch.as_tensor(f0, device=dev), torch.as_tensor(f1, device=dev)
    buf = ops.coarse_match_async(t0, t1, hw_c0, hw_c1, 8.0, thr, border, temp, dense=True)
    torch.cuda.synchronize()
    slots = _lib.load().fm_default_cand_slots(thr)
    lay = layout(n, l, s, c, slots)
    ws = buf.workspace
    base = (-ws.data_ptr()) % 256
    Lp, Sp, splits, panels = lay["Lp"], lay["Sp"], lay["splits"], lay["panels"]
    print(f"== {label}: N={n} L={l} S={s} C={c} Lp={Lp} Sp={Sp} panels={panels} tiles={lay['tiles']} splits={splits}")
    cnt = buf.count.cpu().numpy()
    print("   d_count =", cnt)

    inv_ct = 1.0 / (c * temp)
    # planes
    cp = lay["C"]                       # padded channel count of the planes

    def unfrag(a, rows_pad):            # planes are fragment-major: [rowblock][ks][h][r][8]
        ksteps = cp // 16
        a = a.reshape(n * rows_pad // 32, ksteps, 2, 32, 8).transpose(0, 3, 2, 1, 4)      # rb, r, h, ks, e
        return a.reshape(n, rows_pad, cp)[:, :, :c]

    hi0 = unfrag(view(ws, base, lay["hi0"], n * Lp * cp, torch.float16), Lp)
    lo0 = unfrag(view(ws, base, lay["lo0"], n * Lp * cp, torch.float16), Lp)
    hi1 = unfrag(view(ws, base, lay["hi1"], n * Sp * cp, torch.float16), Sp)
    lo1 = unfrag(view(ws, base, lay["lo1"], n * Sp * cp, torch.float16), Sp)
    # k_prep_f16 writes the planes only for the samples the dense kernel redoes, scaled by a power of two per image
    dcnt_planes = view(ws, base, lay["dense_cnt"], n, torch.int32)
    for b in range(n):
        if dcnt_planes[b] == 0:
            continue
        sc0 = 2.0 ** (14 - np.frexp(np.abs(f0[b]).max())[1]) if np.abs(f0[b]).max() > 0 else 1.0
        sc1 = 2.0 ** (14 - np.frexp(np.abs(f1[b]).max())[1]) if np.abs(f1[b]).max() > 0 else 1.0
        x0, x1 = (f0[b] * np.float32(sc0)).astype(np.float32), (f1[b] * np.float32(sc1)).astype(np.float32)
        e_hi = np.abs(hi0[b, :l].astype(np.float32) - x0.astype(np.float16).astype(np.float32)).max()
        rec = np.abs(hi0[b, :l].astype(np.float64) + lo0[b, :l].astype(np.float64) - x0).max() / sc0
        rec1 = np.abs(hi1[b, :s].astype(np.float64) + lo1[b, :s].astype(np.float64) - x1).max() / sc1
        pad = max(np.abs(hi0[b, l:]).max() if Lp > l else 0, np.abs(hi1[b, s:]).max() if Sp > s else 0)
        print(f"   planes[{b}]: scales 2^{int(np.log2(sc0))}/2^{int(np.log2(sc1))} hi err {e_hi:.2e}  hi+lo recon err "
              f"{rec:.2e}/{rec1:.2e}  pad max {pad}")
        ok &= e_hi == 0 and rec < 1e-5 and pad == 0

    def unfrag8(a, rows_pad):           # int8 planes: [rowblock][ks][h][r][16]
        ks8 = cp // 32
        a = a.reshape(n * rows_pad // 32, ks8, 2, 32, 16).transpose(0, 3, 2, 1, 4)      # rb, r, h, ks, e
        return a.reshape(n, rows_pad, cp)[:, :, :c]

    q0 = unfrag8(view(ws, base, lay["q0"], n * Lp * cp, torch.int8), Lp).astype(np.float64)
    q1 = unfrag8(view(ws, base, lay["q1"], n * Sp * cp, torch.int8), Sp).astype(np.float64)
    sig = view(ws, base, lay["sigimg"], n * 2, torch.float32).reshape(n, 2).astype(np.float64)
    # one step per image; what the clamp at +-127 cut off is accounted for in the margins (fm_device.h)
    clip0 = np.maximum(np.abs(f0) - 127 * sig[:, 0, None, None], 0).sum(2)
    clip1 = np.maximum(np.abs(f1) - 127 * sig[:, 1, None, None], 0).sum(2)
    eq0 = ((np.abs(q0[:, :l] * sig[:, 0, None, None] - f0) - np.maximum(np.abs(f0) - 127 * sig[:, 0, None, None], 0)) / np.maximum(sig[:, 0, None, None], 1e-30)).max()
    eq1 = ((np.abs(q1[:, :s] * sig[:, 1, None, None] - f1) - np.maximum(np.abs(f1) - 127 * sig[:, 1, None, None], 0)) / np.maximum(sig[:, 1, None, None], 1e-30)).max()
    print(f"   int8 planes: steps {sig.tolist()}  (|x - sigma q| - clipped) / sigma max {eq0:.4f} / {eq1:.4f} (must be <= 0.5); "
          f"|q| max {np.abs(q0).max():.0f}/{np.abs(q1).max():.0f}; clipped L1 max {clip0.max():.3g}/{clip1.max():.3g}")
    ok &= eq0 <= 0.5001 and eq1 <= 0.5001

    def q_decode(u):                    # fm_device.h: biased integer code -> integer screening product
        return u.astype(np.int64) - 0x40000000

    rowA = q_decode(view(ws, base, lay["rowmax_u"], n * Lp, torch.int32)).reshape(n, Lp) * (sig[:, 0] * sig[:, 1])[:, None]
    colA = q_decode(view(ws, base, lay["colmax_u"], n * Sp, torch.int32)).reshape(n, Sp) * (sig[:, 0] * sig[:, 1])[:, None]
    nmr = view(ws, base, lay["nmr"], n * Lp, torch.float32).reshape(n, Lp)
    nmc = view(ws, base, lay["nmc"], n * Sp, torch.float32).reshape(n, Sp)
    # softmax denominators: the screening kernel hands over, per row and per column, the list of its significant entries
    # (index, exact dot product; x = -inf marks a reserved but empty place) and k_select sums the list's terms; the
    # dense sum kernel (samples with flagged units) hands over partial sums instead
    splits_s = lay["splits_s"]
    log2e_ = 1.4426950408889634
    k_ = inv_ct * log2e_

    def list_sums(cnt_key, x_key, rows_pad, nm):
        cntv = view(ws, base, lay[cnt_key], n * rows_pad, torch.int32).reshape(n, rows_pad)
        xv = view(ws, base, lay[x_key], n * rows_pad * slots, torch.float32).reshape(n, rows_pad, slots).astype(np.float64)
        live = np.arange(slots)[None, None, :] < np.minimum(cntv, slots)[:, :, None]
        with np.errstate(over="ignore", invalid="ignore"):
            term = np.where(live, np.exp2(xv * k_ + nm[:, :, None].astype(np.float64)), 0.0)
        return np.nan_to_num(term).sum(2)
    # (ccand_x sits right behind ccand_i in the workspace: one [cols, slots] int32 array further)
    lay = dict(lay, ccand_x=lay["ccand_i"] + ((n * Sp * slots * 4 + 255) // 256) * 256)
    rsum = list_sums("cand_count", "cand_x", Lp, nmr)
    csum = list_sums("ccand_count", "ccand_x", Sp, nmc)
    scal = view(ws, base, lay["scalars"], 3, torch.int32)
    dcnt = view(ws, base, lay["dense_cnt"], n, torch.int32)
    print(f"   scalars: flags={scal[0]} dense_units={scal[1]} per sample {dcnt.tolist()}; sparse splits {splits_s} x {lay['units_s']} units")
    rB = view(ws, base, lay["rowB"], n * splits * Lp, torch.float32).reshape(n, splits, Lp).sum(1)
    cB = view(ws, base, lay["colB"], n * panels * Sp, torch.float32).reshape(n, panels, Sp).sum(1)
    for bb in range(n):                 # the dense sum kernel redid the samples with flagged units
        if dcnt[bb] > 0:
            rsum[bb], csum[bb] = rB[bb], cB[bb]
    ccount = view(ws, base, lay["cand_count"], n * Lp, torch.int32).reshape(n, Lp).copy()
    cand_j = view(ws, base, lay["cand_j"], n * Lp * slots, torch.int32).reshape(n, Lp, slots).copy()
    # (the dense kernel's candidate set lives behind the common region; the superset check below looks at the sparse
    # kernel's set and skips the samples the dense kernel redid)
    log2e = 1.4426950408889634
    for b in range(n):
        dot_hi = (q0[b, :l] @ q1[b, :s].T) * (sig[b, 0] * sig[b, 1])      # the screening product
        dot = f0[b].astype(np.float64) @ f1[b].astype(np.float64).T
        sim = dot * inv_ct
        ea = np.abs(rowA[b, :l] - dot_hi.max(1)).max() / max(1.0, np.abs(dot_hi).max())
        ec = np.abs(colA[b, :s] - dot_hi.max(0)).max() / max(1.0, np.abs(dot_hi).max())
        print(f"   [b={b}] passA rel err rows {ea:.2e} cols {ec:.2e}")
        ok &= ea < 1e-5 and ec < 1e-5
        # unit maxima of the integer product (what the sparse kernel decides liveness from): valid entries only
        nunits = Sp // 32
        um = view(ws, base, lay["umax"], n * (Lp // 32) * nunits, torch.float32).reshape(n, Lp // 32, nunits)[b]
        qq = q0[b, :l] @ q1[b, :s].T
        eu = 0.0
        for rbk in range((l + 31) // 32):
            for u in range((s + 31) // 32):
                eu = max(eu, abs(float(um[rbk, u]) - float(qq[32 * rbk: 32 * rbk + 32, 32 * u: 32 * u + 32].max())))
        print(f"   [b={b}] unit maxima: max |difference| {eu:.1f} (integers: must be 0)")
        ok &= eu == 0.0
        mhat_r = -nmr[b, :l] / log2e
        mhat_c = -nmc[b, :s] / log2e
        gap_r = sim.max(1) - mhat_r
        gap_c = sim.max(0) - mhat_c
        print(f"   [b={b}] stabiliser gap rows [{gap_r.min():.3e}, {gap_r.max():.3e}] cols [{gap_c.min():.3e}, {gap_c.max():.3e}] (must be >= 0)")
        ok &= gap_r.min() >= 0 and gap_c.min() >= 0
        rs_ref = np.exp(sim - mhat_r[:, None]).sum(1)
        cs_ref = np.exp(sim - mhat_c[None, :]).sum(0)
        er = np.abs(rsum[b, :l] / rs_ref - 1).max()
        ec2 = np.abs(csum[b, :s] / cs_ref - 1).max()
        print(f"   [b={b}] passB sums rel err rows {er:.2e} cols {ec2:.2e}")
        tol = 1e-5 + 2e-6 * np.abs(sim).max()     # float32 ulp of the largest similarity
        ok &= er < tol and ec2 < tol
        # candidate superset: every (i,j) with conf > thr must be listed
        pr = np.exp(sim - sim.max(1, keepdims=True)); pr /= pr.sum(1, keepdims=True)
        pc = np.exp(sim - sim.max(0, keepdims=True)); pc /= pc.sum(0, keepdims=True)
        conf = pr * pc
        need = np.argwhere(conf > thr)
        missing = [] if dcnt[b] > 0 else [(i, j) for i, j in need if j not in cand_j[b, i, :min(ccount[b, i], slots)]]
        print(f"   [b={b}] candidates: {int(ccount[b, :l].sum())} listed, max/row {ccount[b, :l].max()}, "
              f"{len(need)} needed, {len(missing)} missing")
        ok &= not missing
    # final
    hw_i = (hw_c0[0] * 8, hw_c0[1] * 8)
    ref = orc.coarse_match(f0, f1, hw_i, hw_c0, hw_c1, thr, border, temp)
    m = int(cnt[0])
    got = {k: v.cpu().numpy() for k, v in buf.sliced(m).items()}
    same = (m == ref['i_ids'].shape[0] and np.array_equal(got['b_ids'], ref['b_ids'].numpy())
            and np.array_equal(got['i_ids'], ref['i_ids'].numpy()) and np.array_equal(got['j_ids'], ref['j_ids'].numpy()))
    print(f"   final: M={m} ref M={ref['i_ids'].shape[0]} ids identical: {same}")
    if same and m:
        print(f"   mconf max err {np.abs(got['mconf'] - ref['mconf'].numpy()).max():.2e}; "
              f"kpts equal: {np.array_equal(got['mkpts0_c'], ref['mkpts0_c'].numpy()) and np.array_equal(got['mkpts1_c'], ref['mkpts1_c'].numpy())}")
    ok &= same
    return ok


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="small")
    ap.add_argument("--dist", default="peaky")
    a = ap.parse_args()
    ok = True
    if a.case == "small":
        for (hc, wc, c) in [(8, 8, 64), (16, 16, 64), (12, 20, 128), (20, 30, 256)]:
            f0, f1 = synth.coarse_descriptors(7, 2, hc * wc, c, a.dist)
            ok &= run(f0, f1, (hc, wc), (hc, wc), label=f"{hc}x{wc} C={c}")
    else:
        cfg = synth.CONFIGS[a.case]
        sh = synth.config_shapes(cfg)
        f0, f1 = synth.coarse_descriptors(cfg['seed'], 1, sh['l'], cfg['c'], a.dist)
        ok &= run(f0, f1, (sh['hc'], sh['wc']), (sh['hc'], sh['wc']), label=a.case)
    print("BRINGUP", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
