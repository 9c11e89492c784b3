#!/usr/bin/env python3
"""Static check of a hipcc .s listing: no instruction may read or write a VGPR that is the destination of a
global_load still in flight (issued from inline asm, whose result hipcc believes is there at once - coarse_tf.hip's
weight ring).  Walks every kernel in program order (loops are walked once: the ring is refilled in the same order in
every iteration), keeps the queue of outstanding vector-memory operations and retires all but the newest N at every
s_waitcnt vmcnt(N).

    hipcc ... -save-temps=obj -c coarse_tf.hip && python tools/check_inflight_regs.py coarse_tf-hip-amdgcn-amd-amdhsa-gfx950.s k_ctx_layer
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def main():
    path, want = sys.argv[1], sys.argv[2]
    txt = open(path).read()
    bad = 0
    for m in re.finditer(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(want), txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        queue = []          # outstanding vmem ops: set of destination registers (empty for stores)
        n_ins = 0
        for ln in body.splitlines():
            ln = ln.split(";")[0].strip()
            if not ln or ln.startswith(".") or ln.endswith(":"):
                continue
            n_ins += 1
            op = ln.split()[0]
            args = ln[len(op):]
            pending = set().union(*queue) if queue else set()
            if op.startswith("s_waitcnt"):
                w = re.search(r"vmcnt\((\d+)\)", ln)
                if w:
                    keep = int(w.group(1))
                    queue = queue[len(queue) - keep:] if keep else []
                elif "vmcnt" not in ln and re.search(r"s_waitcnt\s+0x|s_waitcnt\s+\d", ln):
                    queue = []          # raw immediate: treat as a full wait (conservative for the check's purpose)
                continue
            touched = regs(args)
            hit = touched & pending
            if hit:
                bad += 1
                print(f"{name}: instruction #{n_ins} `{ln}` touches in-flight registers {sorted(hit)[:8]}")
            if op.startswith(("global_load", "buffer_load", "flat_load")):
                dst = regs(args.split(",")[0])
                queue.append(dst)
            elif op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic", "buffer_atomic")):
                queue.append(set())
        print(f"{name}: {n_ins} instructions walked, {bad} hazards")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
